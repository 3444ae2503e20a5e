// clip_vit.hip -- CLIP ViT image-tower forward for gfx950 (bf16 MFMA GEMMs, fp32 LayerNorm / softmax /
// residual stream).
//
// Replaces the call `clip_encoder.encode_image(image)` of the reference's offline annotator
// (CLIP_anno.py:166; model loaded at :16 from the un-vendored openai/CLIP package).  Architecture
// (openai/CLIP VisionTransformer == transformers.CLIPVisionModelWithProjection):
//   patch conv (stride = patch, no bias) -> [cls; patches] + pos -> ln_pre -> L x {x += attn(ln_1 x);
//   x += fc2(quick_gelu(fc1(ln_2 x)))} -> ln_post(x[:,0]) -> @ proj
// Parity: unpinned against the reference (third-party, no weights offline); checked against the
// transformers implementation with seeded random weights (tests/test_gpu_clip.py).
//
// Kernels:
//   patchify        image fp32 NCHW -> bf16 [B*np, 3*P*P]  (im2col of non-overlapping patches)
//   gemm_bf16<EPI>  C[M,N] = A[M,K] * W[N,K]^T + bias, 128x128x64 tiles, 4 waves x (2x2) 32x32x16 MFMA, operand tiles by
//                   LDS-DMA (global_load_lds_dwordx4) into two XOR-swizzled LDS stages, one raw barrier per K-step;
//                   epilogues: fp32 store / fp32 residual add / quick_gelu->bf16 / bf16.
//                   Measured round 2 (ViT-B/32, batch 256): 385 TFLOP/s whole tower, 480 TFLOP/s in the 12800-row GEMMs
//                   (19 % of the bf16 peak); the register-staged loop of round 1 with the same epilogue: 392 / 500.  At
//                   K = 768 a tile has 12 K-steps: prologue, epilogue and tile quantisation weigh as much as the K loop.
//   layernorm       fp32 row -> bf16 (GEMM input) or fp32, one wave per token
//   embed           [cls; patch tokens] + positional embedding -> fp32 residual stream
//   attention       MFMA: one workgroup per (image, head), K/Q operands straight from global, V^T in LDS, any token count
// Bound: the four GEMMs per layer -> bf16 MFMA (dense peak 2.5 PFLOP/s).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "gemm8p.hpp"

namespace sc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;

__device__ __forceinline__ bf16_t f2bf(float f) {      // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// 16-bit operand format of the tower: H = false bf16 (8-bit mantissa), H = true IEEE fp16 (11-bit mantissa: what openai/CLIP runs its
// weights and activations in on a GPU, CLIP_anno.py:16 -> clip.load(..., device="cuda")).  Same MFMA rate (2.5 PFLOP/s dense), same
// storage width: every kernel below is a template over H and differs only in the conversion and in the MFMA opcode.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool H>
__device__ __forceinline__ bf16_t cvt16(float f) {
    if (H) return __builtin_bit_cast(bf16_t, (_Float16)f);       // v_cvt_f16_f32, round to nearest even (saturates to +-inf past 65504)
    return f2bf(f);
}
template <bool H>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// out [B*np][Kpad]: columns >= C*P*P are zero (the GEMM wants K % 64 == 0; ViT-L/14 has 3*14*14 = 588 -> 640)
// One thread per 8 consecutive columns (one 16-byte store): the index arithmetic (divisions by run-time sizes) is paid once per chunk and
// the 8 values are a run along kx that wraps into the next patch row / channel at most a few times (P = 32: never, two float4 loads).
template <bool H16>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out,
                                                       int B, int C, int H, int W, int P, int Kpad) {
    const int gw = W / P, gh = H / P, np = gw * gh, K = C * P * P, kc8 = Kpad >> 3;          // Kpad % 64 == 0
    const unsigned total = (unsigned)B * np * kc8;                                           // chunks (host: < 2^31)
    const bool wide = (P & 7) == 0 && (W & 3) == 0;
    for (unsigned ch = blockIdx.x * 256u + threadIdx.x; ch < total; ch += gridDim.x * 256u) {
        const unsigned row = ch / kc8;
        const int k0 = (int)(ch - row * kc8) * 8;
        const int pidx = (int)(row % np), b = (int)(row / np);
        const int py = pidx / gw, px = pidx - py * gw;
        int c = k0 / (P * P), r = k0 - c * P * P, ky = r / P, kx = r - ky * P;
        alignas(16) bf16_t v[8];
        if (wide && k0 + 8 <= K) {
            const float* src = img + (((size_t)b * C + c) * H + py * P + ky) * W + px * P + kx;
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            v[0] = cvt16<H16>(lo.x); v[1] = cvt16<H16>(lo.y); v[2] = cvt16<H16>(lo.z); v[3] = cvt16<H16>(lo.w);
            v[4] = cvt16<H16>(hi.x); v[5] = cvt16<H16>(hi.y); v[6] = cvt16<H16>(hi.z); v[7] = cvt16<H16>(hi.w);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = k0 + e < K ? cvt16<H16>(img[(((size_t)b * C + c) * H + py * P + ky) * W + px * P + kx]) : (bf16_t)0;
                if (++kx == P) { kx = 0; if (++ky == P) { ky = 0; ++c; } }
            }
        }
        *reinterpret_cast<uint4*>(out + (size_t)ch * 8) = *reinterpret_cast<const uint4*>(v);
    }
}

// ---------------------------------------------------------------------------------------------------
enum { EPI_F32 = 0, EPI_RESID = 1, EPI_GELU_BF16 = 2, EPI_BF16 = 3 };

typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA wave instruction: every lane fetches 16 bytes from its own global address; the wave's 1 KiB lands at LDS byte
// address `lds_dst` (wave-uniform) + 16 x lane.  Inline asm on purpose: hipcc drains ANY outstanding LDS-DMA (vmcnt(0)) before
// the next ds_read it cannot prove disjoint, which serialises a double-buffered loop; issued from asm the DMA is invisible to
// its bookkeeping and completion is counted by hand (the s_waitcnt vmcnt(0) of SC_GEMM_SYNC).  M0 holds the LDS base.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lptr_t)p; }

// C[M,N] = A[M,K] W[N,K]^T (+ bias, epilogue).  128x128x64 tiles, 4 waves x (2x2) v_mfma_f32_32x32x16_bf16.
// Operand tiles go global -> LDS with the gfx950 LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass),
// double-buffered: the tile of K-step k+1 is in flight while the MFMAs of step k run, ONE raw s_barrier per K-step, the only
// vmcnt wait is the one for the tile about to be read.  The DMA writes lane-linear (base + 16 B x lane), so the XOR swizzle that
// makes the fragment ds_read_b128 conflict-free is applied to the SOURCE address: LDS chunk c = row*8 + slot holds the 8 bf16 of
// logical K-chunk slot ^ (row & 7) of that row (the read side applies the same involution).  Rows past M / N are clamped to the
// last valid row (their products are never stored).  The two buffers are separate __shared__ objects so that the compiler can
// prove the DMA of one does not alias the fragment reads of the other (otherwise it drains the DMA before the first ds_read).
// BT x BT output tiles (BT = 128: four waves x (2x2) MFMA tiles; BT = 64: four waves x one MFMA tile, for launches whose 128-wide
// grid would leave most CUs without a workgroup -- batch 32: 78 tiles for N = 768 -- instead of round 1's split-K with fp32
// atomics, which also made the summation order run-dependent).  Two LDS stages: the tile of K-step k+1 is in flight while k is
// consumed; large grids put two (BT = 128) or more workgroups on a CU and they hide the rest of the latency.
#ifndef SC_GEMM64_STAGES
#define SC_GEMM64_STAGES 3
#endif
template <int EPI, int BT, bool H16>
__global__ __launch_bounds__(256, BT == 128 ? 2 : 4) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                                           const float* __restrict__ bias, void* __restrict__ out,
                                                                           int M, int N, int K) {
    constexpr int MI = BT / 64;                // MFMA tiles per wave and dimension
    constexpr int CH = BT * 8;                 // 16-byte chunks per operand tile (BT rows x 64 K)
    constexpr int QN = CH / 256;               // DMA instructions per thread and operand
    // LDS stages.  The 64 x 64 kernel serves grids whose 128-wide form would leave CUs idle (batch 32): its K-step is 4 MFMAs per wave,
    // shorter than the DMA round trip, so it keeps SC_GEMM64_STAGES - 1 steps in flight (measured at batch 32: 3 stages 1.05 ms per
    // tower, 2 stages 1.09, 4 stages 1.08, 6 stages 1.58 -- these launches are mostly launch latency, 12-17 us whatever the shape).
    constexpr int NS = BT == 64 ? SC_GEMM64_STAGES : 2, PD = NS - 1;
    extern __shared__ uint4 Sbuf[];            // [NS stages][A: BT rows x 8 chunks | B: BT rows x 8 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2: with the plain (n, m) grid each
    // XCD touched every row tile of A, so the operand tiles were re-fetched from the Infinity Cache.  Here XCD k owns a
    // CONTIGUOUS range of the row-major tile list (all N tiles of a few M tiles): its A slice (1/8 of A) and W stay in L2.
    const int ntn = gridDim.x, tiles = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    const int tq = tiles >> 3, trem = tiles & 7;
    const int tile = (xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq) + idx;
    const int bm = (tile / ntn) * BT, bn = (tile % ntn) * BT;
    const int kt1 = K / 64;
    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-thread DMA sources: chunk c = q*256 + tid of each operand tile
    const bf16_t* pa[QN];
    const bf16_t* pb[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int c = q * 256 + tid, row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
        pa[q] = A + (size_t)min(bm + row, M - 1) * K + kc * 8;
        pb[q] = Wt + (size_t)min(bn + row, N - 1) * K + kc * 8;
    }
    const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(Sbuf)) + wave_off;
    auto issue = [&](int kt, int stage) {
        const unsigned base = lds0 + (unsigned)stage * (2u * CH) * 16u;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            glds16(pa[q] + (size_t)kt * 64, base + q * 256 * 16);
            glds16(pb[q] + (size_t)kt * 64, base + (CH + q * 256) * 16);
        }
    };
    // fragment reads: lanes 0-31 = 32 consecutive rows, lanes 32-63 the next 16-byte K chunk.  ds_read_b128 is serviced in the
    // lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with 128-byte rows the bank quad of a chunk is
    // 8 (row & 1) + slot, so the slot permutation must differ for the 8 rows of equal parity inside each group --
    // slot = chunk ^ ((row >> 1) & 7) does (round 1's chunk ^ (row & 7) left every read 2-way conflicted: PMC
    // SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).  The fragments of K sub-step kk+1 are requested before the MFMAs of kk.
    auto compute = [&](int stage) {
        const uint4* As = Sbuf + stage * (2 * CH);
        const uint4* Bs = As + CH;
        const int ra = (BT / 2) * wr + (lane & 31), rb = (BT / 2) * wc + (lane & 31);
        const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;          // rows +32 keep (row >> 1) & 7
        bf16x8 af[2][MI], bf[2][MI];
        auto frags = [&](int kk, bf16x8 (&a2)[MI], bf16x8 (&b2)[MI]) {
            const int kc = 2 * kk + (lane >> 5);
#pragma unroll
            for (int i = 0; i < MI; ++i) a2[i] = __builtin_bit_cast(bf16x8, As[(ra + 32 * i) * 8 + (kc ^ sa)]);
#pragma unroll
            for (int j = 0; j < MI; ++j) b2[j] = __builtin_bit_cast(bf16x8, Bs[(rb + 32 * j) * 8 + (kc ^ sb)]);
        };
        frags(0, af[0], bf[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = mfma16<H16>(af[kk & 1][i], bf[kk & 1][j], acc[i][j]);
        }
    };
    // Tile kt is complete for every wave once each wave has waited for its own DMA (issued from asm: counted by hand) and all
    // have met at the barrier; the same barrier says that everybody has finished reading the stage the next DMA overwrites.
#define SC_GEMM_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < kt1) issue(d, d);
    for (int kt = 0; kt < kt1; ++kt) {
        // step kt has landed once at most min(PD - 1, steps issued after it) DMA groups (2 QN instructions each) are outstanding
        const int later = min(PD - 1, kt1 - 1 - kt);
        if (later <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * QN) : "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * QN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(6 * QN) : "memory");
        if (kt + PD < kt1) issue(kt + PD, (kt + PD) % NS);       // into the stage read at step kt - 1: everybody is past it
        compute(kt % NS);
    }
    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Fast path (full-width tile, row pitch a multiple of 16 bytes): the tile goes through the (now free) LDS stages and leaves
    // as 16-byte row-contiguous stores -- BT/16 (bf16) / BT/8 (fp32) store instructions per lane instead of 16 MI^2 two- or
    // four-byte ones.  A K-sweep showed the element-wise epilogue costing as much as 12 K-steps (46 us of the 92 us of the
    // 12800 x 2304 x 768 qkv GEMM were spent at K = 64); the residual add reads its rows the same way, all of them before the
    // first store (`x[o] += v` element by element made every load wait for the previous store).
    constexpr bool OUT_BF16 = EPI == EPI_GELU_BF16 || EPI == EPI_BF16;
    if (bn + BT <= N && (N % 8) == 0) {
        SC_GEMM_SYNC();                                         // every wave is done with the stages
        float* Cf = reinterpret_cast<float*>(Sbuf);
        bf16_t* Ch = reinterpret_cast<bf16_t*>(Sbuf);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int cl = (BT / 2) * wc + 32 * j + (lane & 31);
                const float bv = bias ? bias[bn + cl] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (BT / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float v = acc[i][j][r] + bv;
                    if (EPI == EPI_GELU_BF16) Ch[rl * BT + cl] = cvt16<H16>(v / (1.f + __expf(-1.702f * v)));
                    else if (EPI == EPI_BF16) Ch[rl * BT + cl] = cvt16<H16>(v);
                    else Cf[rl * BT + cl] = v;
                }
            }
        __syncthreads();
        if (OUT_BF16) {
            constexpr int CPR = BT / 8;                          // 16-byte chunks per tile row
#pragma unroll
            for (int q = 0; q < BT * CPR / 256; ++q) {
                const int chunk = q * 256 + tid, rl = chunk / CPR, c8 = (chunk % CPR) * 8;
                if (bm + rl < M)
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + (size_t)(bm + rl) * N + bn + c8) =
                        *reinterpret_cast<const uint4*>(Ch + rl * BT + c8);
            }
        } else {
            constexpr int CPR = BT / 4, NQ = BT * CPR / 256;
            float4 res[NQ];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int chunk = q * 256 + tid, rl = chunk / CPR, c4 = (chunk % CPR) * 4;
                    res[q] = bm + rl < M ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(out) + (size_t)(bm + rl) * N + bn + c4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int chunk = q * 256 + tid, rl = chunk / CPR, c4 = (chunk % CPR) * 4;
                float4 v = *reinterpret_cast<const float4*>(Cf + rl * BT + c4);
                if (EPI == EPI_RESID) { v.x += res[q].x; v.y += res[q].y; v.z += res[q].z; v.w += res[q].w; }
                if (bm + rl < M) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)(bm + rl) * N + bn + c4) = v;
            }
        }
        return;
    }
#undef SC_GEMM_SYNC
    // general path: partial tiles in N or a row pitch that is not a multiple of 16 bytes
    float res[MI][MI][16];
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int col = bn + (BT / 2) * wc + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = bm + (BT / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    res[i][j][r] = (col < N && row < M) ? reinterpret_cast<const float*>(out)[(size_t)row * N + col] : 0.f;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int col = bn + (BT / 2) * wc + 32 * j + (lane & 31);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + (BT / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                const float v = acc[i][j][r] + bv;
                const size_t o = (size_t)row * N + col;
                if (EPI == EPI_F32) reinterpret_cast<float*>(out)[o] = v;
                else if (EPI == EPI_RESID) reinterpret_cast<float*>(out)[o] = res[i][j][r] + v;
                else if (EPI == EPI_GELU_BF16) reinterpret_cast<bf16_t*>(out)[o] = cvt16<H16>(v / (1.f + __expf(-1.702f * v)));
                else reinterpret_cast<bf16_t*>(out)[o] = cvt16<H16>(v);
            }
        }
}

// 256 x 256 x 64 tiles for the large shapes (M >= 2048 tokens, N a multiple of 256).  Why: the 128 x 128 kernels run at 600-700 TFLOP/s whatever
// the problem size (M = 51,200 included) with the matrix pipe ~30 % busy, the LDS read port 15 % busy and no bank conflicts (PMC) -- what
// they saturate is the L2 -> LDS path: a 128 x 128 tile pulls 32 KB per 512 MFMA cycles and SIMD, i.e. ~10 TB/s chip-wide at that rate.  A
// 256 x 256 tile needs half the operand bytes per FLOP.  Eight waves as 4 (M) x 2 (N), each 64 x 128 = 2 x 4 MFMA tiles (128 accumulator
// registers); per K sub-step a wave reads 2 + 4 fragments for 8 MFMAs.  128 KB of LDS stages, one workgroup per CU, the same LDS-DMA
// addressing and single barrier per K-step as above.  Measured on the way (M = 51,200, fp16): this kernel 700-820 TFLOP/s against 620-720 of
// the 128 x 128 persistent kernel; four 32 KB stages (K-step 32, three batches in flight) 5 % SLOWER than two 64 KB stages; the same kernel
// without its LDS-DMA instructions 890-1,150, without its MFMAs 860-970: operand delivery and matrix work cost about the same and overlap
// only partly -- an LDS-DMA piece occupies its wave for 60-185 issue cycles (MI355X_MICROARCH.md), 8 pieces per wave and K-step against 32
// MFMAs.  Operands through registers (global_load_dwordx4 into eight named uint4 + ds_write_b128, no LDS-DMA at all): 610-750, the same; with
// tile-major, pre-swizzled operands (every piece one contiguous KB): the same.  Neither the transport nor the layout is the limit.
// A producer / consumer form (8 computing waves that never issue a DMA, 4 loader waves three K-steps ahead through four 32 KB stages, one
// s_barrier per K-step) passed the same tests and ran 660-750: the computing waves were not what was short -- the DMA-only rate above is
// ~6.5 TB/s of L2 -> LDS traffic chip-wide in 16-byte-per-lane pieces that use 64-128 bytes of each cache line's row.  The next lever is the
// operand LAYOUT (tile-major weights and activations so that a piece is one contiguous KB), not the schedule.  The epilogue stages the tile through the free LDS: 16-bit outputs at
// once, fp32 outputs in two 128-row halves.
#ifndef SC_GEMM256_BK
#define SC_GEMM256_BK 64
#endif

template <int EPI, bool H16>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                         void* __restrict__ out, int M, int N, int K) {
    // K-step BK = 64: two 64 KB stages (one batch in flight); BK = 32: four 32 KB stages (three batches in flight -- what hides the DMA
    // round trip when there is ONE workgroup per CU).  Rows of BK / 8 sixteen-byte chunks; the XOR swizzle of the chunk slot is
    // (row >> 1) & 7 for 8-chunk rows (as above) and (row >> 2) & 3 for 4-chunk rows (same service groups, same argument).
    constexpr int BT = 256, BK = SC_GEMM256_BK, CPR = BK / 8, CH = BT * CPR, QN = CH / 512, NS = 128 * 1024 / (2 * CH * 16), PD = NS - 1, KK = BK / 16;
    static_assert(BK == 64 || BK == 32, "K-step");
    extern __shared__ uint4 Sbuf[];            // [NS stages][A: 256 rows x CPR chunks | B: 256 rows x CPR chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = gridDim.x, tiles = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    const int tq = tiles >> 3, trem = tiles & 7;
    const int tile = (xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq) + idx;
    const int bm = (tile / ntn) * BT, bn = (tile % ntn) * BT;
    const int kt1 = K / BK;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bf16_t* pa[QN];
    const bf16_t* pb[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int c = q * 512 + tid, row = c / CPR, kc = (c % CPR) ^ swz(row);
        pa[q] = A + (size_t)min(bm + row, M - 1) * K + kc * 8;
        pb[q] = Wt + (size_t)min(bn + row, N - 1) * K + kc * 8;
    }
    const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(Sbuf)) + wave_off;
    auto issue = [&](int kt, int stage) {
        const unsigned base = lds0 + (unsigned)stage * (2u * CH) * 16u;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            glds16(pa[q] + (size_t)kt * BK, base + q * 512 * 16);
            glds16(pb[q] + (size_t)kt * BK, base + (CH + q * 512) * 16);
        }
    };
    auto compute = [&](int stage) {
        const uint4* As = Sbuf + stage * (2 * CH);
        const uint4* Bs = As + CH;
        const int ra = 64 * wr + (lane & 31), rb = 128 * wc + (lane & 31);
        const int sa = swz(ra), sb = swz(rb);                      // rows + 32 keep the swizzle term
        bf16x8 af[2][2], bf[2][4];
        auto frags = [&](int kk, bf16x8 (&a2)[2], bf16x8 (&b2)[4]) {
            const int kc = 2 * kk + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) a2[i] = __builtin_bit_cast(bf16x8, As[(ra + 32 * i) * CPR + (kc ^ sa)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b2[j] = __builtin_bit_cast(bf16x8, Bs[(rb + 32 * j) * CPR + (kc ^ sb)]);
        };
        frags(0, af[0], bf[0]);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + 1 < KK) frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<H16>(af[kk & 1][i], bf[kk & 1][j], acc[i][j]);
        }
    };
#define SC_GEMM_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < kt1) issue(d, d);
    for (int kt = 0; kt < kt1; ++kt) {
        // step kt has landed once at most min(PD - 1, steps issued after it) DMA groups (2 QN instructions each) are outstanding
        const int later = min(PD - 1, kt1 - 1 - kt);
        if (later <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * QN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * QN) : "memory");
        if (kt + PD < kt1) issue(kt + PD, (kt + PD) % NS);       // into the stage read at step kt - 1: everybody is past it
        compute(kt % NS);
    }
    SC_GEMM_SYNC();                                             // every wave is done with the stages
    constexpr bool OUT_BF16 = EPI == EPI_GELU_BF16 || EPI == EPI_BF16;
    if (OUT_BF16) {
        bf16_t* Ch = reinterpret_cast<bf16_t*>(Sbuf);           // [256][256] 16-bit: 128 KB
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = 128 * wc + 32 * j + (lane & 31);
                const float bv = bias ? bias[bn + cl] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float v = acc[i][j][r] + bv;
                    Ch[rl * BT + cl] = cvt16<H16>(EPI == EPI_GELU_BF16 ? v / (1.f + __expf(-1.702f * v)) : v);
                }
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BT * 32 / 512; ++q) {
            const int chunk = q * 512 + tid, rl = chunk >> 5, c8 = (chunk & 31) * 8;
            if (bm + rl < M)
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + (size_t)(bm + rl) * N + bn + c8) =
                    *reinterpret_cast<const uint4*>(Ch + rl * BT + c8);
        }
    } else {
        float* Cf = reinterpret_cast<float*>(Sbuf);             // [128][256] fp32: 128 KB, one half of the tile at a time
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if ((wr >> 1) == hh) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cl = 128 * wc + 32 * j + (lane & 31);
                        const float bv = bias ? bias[bn + cl] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = 64 * (wr & 1) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            Cf[rl * BT + cl] = acc[i][j][r] + bv;
                        }
                    }
            }
            __syncthreads();
            constexpr int NQ = 128 * 64 / 512;
            float4 res[NQ];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int chunk = q * 512 + tid, rl = chunk >> 6, c4 = (chunk & 63) * 4, row = bm + 128 * hh + rl;
                    res[q] = row < M ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(out) + (size_t)row * N + bn + c4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int chunk = q * 512 + tid, rl = chunk >> 6, c4 = (chunk & 63) * 4, row = bm + 128 * hh + rl;
                float4 v = *reinterpret_cast<const float4*>(Cf + rl * BT + c4);
                if (EPI == EPI_RESID) { v.x += res[q].x; v.y += res[q].y; v.z += res[q].z; v.w += res[q].w; }
                if (row < M) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * N + bn + c4) = v;
            }
            __syncthreads();
        }
    }
#undef SC_GEMM_SYNC
}

// Persistent form of the 128x128 kernel for large grids: 2 workgroups per CU stay resident and walk the tile list of their XCD.
// What it buys over one workgroup per tile: the first K-step of the NEXT tile is requested during the last K-step of the current
// one, so its DMA round trip (~1.3 us under load) and the pointer set-up run under the epilogue instead of in front of an idle
// matrix pipe, and no workgroup is re-launched per tile.  The epilogue stages the tile through the LDS stage that was consumed
// last (the other one is receiving the next tile): 32 KiB = the whole bf16 tile, or the fp32 tile in two 64-row halves.
template <int EPI, bool H16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_persist_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt,
                                                                   const float* __restrict__ bias, void* __restrict__ out,
                                                                   int M, int N, int K, int ntn, int tiles) {
    constexpr int BT = 128, CH = 1024, QN = 4;
    extern __shared__ uint4 Sbuf[];            // [2 stages][A 1024 chunks | B 1024 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD k owns a contiguous range of the row-major tile list; its workgroups take the tiles of that range round-robin
    const int xcd = blockIdx.x & 7, wloc = blockIdx.x >> 3, wpx = gridDim.x >> 3;       // gridDim.x is a multiple of 8
    const int tq = tiles >> 3, trem = tiles & 7;
    const int t_lo = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
    const int t_hi = t_lo + tq + (xcd < trem ? 1 : 0);
    const int kt1 = K / 64;
    const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(Sbuf)) + wave_off;
    const int ra = 64 * wr + (lane & 31), rb = 64 * wc + (lane & 31);
    const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;

    const bf16_t* pa[QN];
    const bf16_t* pb[QN];
    auto pointers = [&](int tile) {
        const int bm = (tile / ntn) * BT, bn = (tile % ntn) * BT;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int c = q * 256 + tid, row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
            pa[q] = A + (size_t)min(bm + row, M - 1) * K + kc * 8;
            pb[q] = Wt + (size_t)min(bn + row, N - 1) * K + kc * 8;
        }
    };
    auto issue = [&](int kt, int stage) {
        const unsigned base = lds0 + (unsigned)stage * (2u * CH) * 16u;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            glds16(pa[q] + (size_t)kt * 64, base + q * 256 * 16);
            glds16(pb[q] + (size_t)kt * 64, base + (CH + q * 256) * 16);
        }
    };
#define SC_GEMM_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SC_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    int g = 0;                                  // running K-step count: stage = g & 1
    int tile = t_lo + wloc;
    if (tile < t_hi && kt1 > 0) { pointers(tile); issue(0, 0); }
    for (; tile < t_hi; tile += wpx) {
        const int bm = (tile / ntn) * BT, bn = (tile % ntn) * BT;
        const int next = tile + wpx;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < kt1; ++kt, ++g) {
            SC_GEMM_SYNC();                     // stage g & 1 has landed for everybody; stage (g + 1) & 1 is free
            if (kt + 1 < kt1) issue(kt + 1, (g + 1) & 1);
            else if (next < t_hi) { pointers(next); issue(0, (g + 1) & 1); }       // first K-step of the NEXT tile
            const uint4* As = Sbuf + (g & 1) * (2 * CH);
            const uint4* Bs = As + CH;
            bf16x8 af[2][2], bf[2][2];
            auto frags = [&](int kk, bf16x8 (&a2)[2], bf16x8 (&b2)[2]) {
                const int kc = 2 * kk + (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) a2[i] = __builtin_bit_cast(bf16x8, As[(ra + 32 * i) * 8 + (kc ^ sa)]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b2[j] = __builtin_bit_cast(bf16x8, Bs[(rb + 32 * j) * 8 + (kc ^ sb)]);
            };
            frags(0, af[0], bf[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma16<H16>(af[kk & 1][i], bf[kk & 1][j], acc[i][j]);
            }
        }
        // ---- epilogue through the stage consumed last, (g - 1) & 1; the other one may be receiving the next tile ----
        constexpr bool OUT_BF16 = EPI == EPI_GELU_BF16 || EPI == EPI_BF16;
        uint4* Cs = Sbuf + ((g - 1) & 1) * (2 * CH);
        if (OUT_BF16) {
            bf16_t* Ch = reinterpret_cast<bf16_t*>(Cs);
            SC_LDS_SYNC();                      // every wave is done reading that stage
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cl = 64 * wc + 32 * j + (lane & 31);
                    const float bv = bias ? bias[min(bn + cl, N - 1)] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float v = acc[i][j][r] + bv;
                        Ch[rl * BT + cl] = EPI == EPI_GELU_BF16 ? cvt16<H16>(v / (1.f + __expf(-1.702f * v))) : cvt16<H16>(v);
                    }
                }
            SC_LDS_SYNC();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int chunk = q * 256 + tid, rl = chunk >> 4, c8 = (chunk & 15) * 8;
                if (bm + rl < M && bn + c8 < N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + (size_t)(bm + rl) * N + bn + c8) =
                        *reinterpret_cast<const uint4*>(Ch + rl * BT + c8);
            }
        } else {
            float* Cf = reinterpret_cast<float*>(Cs);                       // [64 rows][128] per half
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                SC_LDS_SYNC();                  // stage free (half 0) / previous half stored (half 1)
                if (wr == half) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int cl = 64 * wc + 32 * j + (lane & 31);
                            const float bv = bias ? bias[min(bn + cl, N - 1)] : 0.f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int rl = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                Cf[rl * BT + cl] = acc[i][j][r] + bv;
                            }
                        }
                }
                SC_LDS_SYNC();
                float4 res[8];
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int chunk = q * 256 + tid, rl = chunk >> 5, c4 = (chunk & 31) * 4, row = bm + 64 * half + rl;
                        res[q] = (row < M && bn + c4 < N) ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(out) + (size_t)row * N + bn + c4)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int chunk = q * 256 + tid, rl = chunk >> 5, c4 = (chunk & 31) * 4, row = bm + 64 * half + rl;
                    float4 v = *reinterpret_cast<const float4*>(Cf + rl * BT + c4);
                    if (EPI == EPI_RESID) { v.x += res[q].x; v.y += res[q].y; v.z += res[q].z; v.w += res[q].w; }
                    if (row < M && bn + c4 < N) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * N + bn + c4) = v;
                }
            }
        }
    }
#undef SC_LDS_SYNC
#undef SC_GEMM_SYNC
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over D (multiple of 64, <= 1024 here) per row; x rows are `stride` floats apart.
template <bool OUT_BF16, bool H16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int stride, const float* __restrict__ g,
                                                        const float* __restrict__ b, void* __restrict__ out, int rows, int D,
                                                        float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * stride;
    if ((D & 255) == 0 && D <= 1024 && (stride & 3) == 0) {
        // the row lives in registers: D/256 float4 per lane, one pass (two-pass variance: mean first, then centred squares)
        float4 v[4];
        const int nv = D >> 8;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) { v[k] = reinterpret_cast<const float4*>(xr)[k * 64 + lane]; s += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / D;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) {
                v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
                ss += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
            }
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        const float inv = rsqrtf(ss / D + eps);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) {
                const int d = (k * 64 + lane) * 4;
                const float4 gg = *reinterpret_cast<const float4*>(g + d), bb = *reinterpret_cast<const float4*>(b + d);
                const float o0 = v[k].x * inv * gg.x + bb.x, o1 = v[k].y * inv * gg.y + bb.y;
                const float o2 = v[k].z * inv * gg.z + bb.z, o3 = v[k].w * inv * gg.w + bb.w;
                if (OUT_BF16) {
                    const uint2 pk = make_uint2((uint32_t)cvt16<H16>(o0) | ((uint32_t)cvt16<H16>(o1) << 16),
                                                (uint32_t)cvt16<H16>(o2) | ((uint32_t)cvt16<H16>(o3) << 16));
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)row * D + d) = pk;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * D + d) = make_float4(o0, o1, o2, o3);
                }
            }
        return;
    }
    float s = 0.f, ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = xr[d]; s += v; ss += v * v; }
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    const float mean = s / D;
    const float var = fmaxf(ss / D - mean * mean, 0.f);
    const float inv = rsqrtf(var + eps);
    for (int d = lane; d < D; d += 64) {
        const float v = (xr[d] - mean) * inv * g[d] + b[d];
        if (OUT_BF16) reinterpret_cast<bf16_t*>(out)[(size_t)row * D + d] = cvt16<H16>(v);
        else reinterpret_cast<float*>(out)[(size_t)row * D + d] = v;
    }
}

// tokens: x[b,0] = cls + pos[0]; x[b,1+i] = patch[b,i] + pos[1+i]
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                    const float* __restrict__ pos, float* __restrict__ x, int B, int T, int D) {
    const size_t total = (size_t)B * T * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int d = (int)(idx % D);
        const int t = (int)((idx / D) % T);
        const int b = (int)(idx / ((size_t)D * T));
        const float v = t == 0 ? cls[d] : patch[((size_t)b * (T - 1) + (t - 1)) * D + d];
        x[idx] = v + pos[(size_t)t * D + d];
    }
}

// Multi-head self-attention on the matrix cores.  qkv: bf16 [B*T, 3*D] (q | k | v), head dim 64, any T.  out bf16 [B*T, D].
// One workgroup (4 waves) per (image, head); a wave owns blocks of 32 queries and walks the keys 32 at a time.
//   S^T[key][query] = K Q^T      A = K rows, B = Q rows, both read straight from global memory: the contraction index
//                                (head dim) is contiguous in the qkv row, which is exactly the 32x32x16 operand layout
//   softmax over keys            two passes over the key chunks (maximum, then p = exp(s - max) with its sum): S is recomputed (4 MFMAs
//                                per chunk) instead of rescaling the output accumulators; O is divided by the sum at the end
//   O[query][d] += P V           A = P: the C/D registers of S^T ARE a legal A operand (lane = query, the 8 values of a
//                                K-step are 8 keys); B = V with keys contiguous per lane -> V is staged once per
//                                workgroup TRANSPOSED in LDS (Vt[d][key]), the K-step's key order follows the C/D layout
// fp32 scores / statistics / accumulation, bf16 probabilities (openai/CLIP on GPU keeps them in fp16).
// K of the (image, head) is staged in LDS as well when it fits next to V^T (k_lds: rows of 72 elements = 144 bytes, so that the 32 rows of
// a fragment read spread over the banks): both passes walk the keys with ds_read_b128 instead of one dependent global round trip per
// chunk and pass -- at T = 257 (ViT-L/14) that was 54 round trips in a row for the wave with three query blocks, 87 us per layer.  Eight
// waves: the nine query blocks of T = 257 are two deep instead of three.
constexpr int ATT_NW = 8, ATT_KLD = 72;
template <bool H16>
__global__ __launch_bounds__(64 * ATT_NW) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int D,
                                                                int heads, float scale, int k_lds) {
    extern __shared__ bf16_t Vt[];                   // [64][ldv], ldv = Tp + 4 (row stride = odd multiple of 2 words); then K [Tp][72]
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int Tp = (T + 31) & ~31, ldv = Tp + 4;
    bf16_t* Ks = Vt + 64 * ldv;                      // 64 ldv elements = 128 (Tp + 4) bytes: 16-byte aligned
    const size_t rs = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + (size_t)hd * 64;
    for (int e = threadIdx.x; e < Tp * 8; e += 64 * ATT_NW) {
        const int key = e >> 3, dc = (e & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0), k = make_uint4(0, 0, 0, 0);
        if (key < T) {
            v = *reinterpret_cast<const uint4*>(base + (size_t)key * rs + 2 * D + dc);
            if (k_lds) k = *reinterpret_cast<const uint4*>(base + (size_t)key * rs + D + dc);
        }
        const bf16_t* pv = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(dc + j) * ldv + key] = pv[j];
        if (k_lds) *reinterpret_cast<uint4*>(Ks + key * ATT_KLD + dc) = k;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    auto frag = [&](int row, int col_off) -> bf16x8 {       // 8 consecutive head-dim values of a qkv row (zero past T)
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < T) v = *reinterpret_cast<const uint4*>(base + (size_t)row * rs + col_off + 8 * h);
        return __builtin_bit_cast(bf16x8, v);
    };
    const float c2 = scale * 1.44269504088896341f;            // exp(scale s - m) = exp2(c2 s - c2 max s): one fma + v_exp_f32 per score
    const int Tfull = T & ~31;                                // keys below Tfull need no mask
    // Query blocks go round-robin over the eight waves.  T = 257 (ViT-L/14 at 224 x 224) is 8 full blocks + ONE query: left to wave 0 as a
    // ninth block it doubled the kernel (seven waves idle for a whole block time: 35 us per layer).  A single left-over block is instead
    // split over the KEYS: every wave takes the key chunks wave, wave + 8, ... for those queries, the partial (max, sum, O) of the eight
    // waves are combined in wave order through the (by then free) LDS -- round 4.
    const int nqb = (T + 31) >> 5;
    const bool ksplit = k_lds && nqb > ATT_NW && (nqb % ATT_NW) == 1;
    for (int qb = wave; qb < (ksplit ? nqb - 1 : nqb); qb += ATT_NW) {
        const int q = qb * 32 + n;
        bf16x8 qf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = frag(q, 16 * s4);
        auto scores = [&](int kc) -> f32x16 {                 // S[r] = raw score of key kc + (r&3) + 8(r>>2) + 4h, query q
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            if (k_lds) {                                      // rows past T were staged as zeros, like frag() returns them
                const bf16_t* kp = Ks + (kc + n) * ATT_KLD + 8 * h;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    S = mfma16<H16>(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + 16 * s4)), qf[s4], S);
            } else {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) S = mfma16<H16>(frag(kc + n, D + 16 * s4), qf[s4], S);
            }
            return S;
        };
        // pass 1: the maximum raw score of this lane's query (the two halves of the wave hold 16 keys of a chunk each); scale > 0
        float smax = -3.0e38f;
        for (int kc = 0; kc < Tfull; kc += 32) {
            const f32x16 S = scores(kc);
#pragma unroll
            for (int r = 0; r < 16; ++r) smax = fmaxf(smax, S[r]);
        }
        if (Tfull < T) {
            const f32x16 S = scores(Tfull);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (Tfull + (r & 3) + 8 * (r >> 2) + 4 * h < T) smax = fmaxf(smax, S[r]);
        }
        smax = fmaxf(smax, __shfl_xor(smax, 32));
        const float mc = smax * c2;
        // pass 2: p = exp(scale (s - max)) in (0, 1], O += p V, l += p; O is divided by l at the end (fp32)
        f32x16 O[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
        float l = 0.f;
        auto accumulate = [&](int kc, const float (&pr)[16]) {
            bf16x8 pf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16_t tmp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tmp[e] = cvt16<H16>(pr[8 * j + e]);
                pf[j] = __builtin_bit_cast(bf16x8, tmp);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // A operand: keys kc + 16j + 4h + {0..3} and {8..11} of head-dim row 32t + n (V^T); B operand: P of query n
                    const bf16_t* vp = Vt + (32 * t + n) * ldv + kc + 16 * j + 4 * h;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 8);
                    const uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    O[t] = mfma16<H16>(__builtin_bit_cast(bf16x8, pk), pf[j], O[t]);      // O^T[dim][query]: a lane holds ITS query's dims
                }
        };
        for (int kc = 0; kc < Tfull; kc += 32) {
            const f32x16 S = scores(kc);
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(fmaf(S[r], c2, -mc));
                l += pr[r];
            }
            accumulate(kc, pr);
        }
        if (Tfull < T) {
            const f32x16 S = scores(Tfull);
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = Tfull + (r & 3) + 8 * (r >> 2) + 4 * h < T ? __builtin_amdgcn_exp2f(fmaf(S[r], c2, -mc)) : 0.f;
                l += pr[r];
            }
            accumulate(Tfull, pr);
        }
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;                            // of query n: this lane's column of S^T AND of O^T
        // O^T[t][r] = dim 32 t + (r & 3) + 8 (r >> 2) + 4 h of query n: four consecutive dims per r >> 2 -> 8-byte stores (the 2-byte
        // stores of the [query][dim] form cost 4 of the kernel's 35 us at T = 257)
        if (q < T) {
            bf16_t* orow = out + ((size_t)b * T + q) * D + hd * 64 + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const uint32_t lo = (uint32_t)cvt16<H16>(O[t][4 * r4] * inv) | ((uint32_t)cvt16<H16>(O[t][4 * r4 + 1] * inv) << 16);
                    const uint32_t hi = (uint32_t)cvt16<H16>(O[t][4 * r4 + 2] * inv) | ((uint32_t)cvt16<H16>(O[t][4 * r4 + 3] * inv) << 16);
                    *reinterpret_cast<uint2*>(orow + 32 * t + 8 * r4) = make_uint2(lo, hi);
                }
        }
    }
    if (ksplit) {
        const int qb = nqb - 1, q = qb * 32 + n;
        bf16x8 qf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = frag(q, 16 * s4);
        auto scores = [&](int kc) -> f32x16 {
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const bf16_t* kp = Ks + (kc + n) * ATT_KLD + 8 * h;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                S = mfma16<H16>(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + 16 * s4)), qf[s4], S);
            return S;
        };
        float smax = -3.0e38f;
        for (int kc = 32 * wave; kc < T; kc += 32 * ATT_NW) {
            const f32x16 S = scores(kc);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kc + (r & 3) + 8 * (r >> 2) + 4 * h < T) smax = fmaxf(smax, S[r]);
        }
        smax = fmaxf(smax, __shfl_xor(smax, 32));
        const float mc = smax * c2;
        f32x16 O[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
        float l = 0.f;
        for (int kc = 32 * wave; kc < T; kc += 32 * ATT_NW) {
            const f32x16 S = scores(kc);
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = kc + (r & 3) + 8 * (r >> 2) + 4 * h < T ? __builtin_amdgcn_exp2f(fmaf(S[r], c2, -mc)) : 0.f;
                l += pr[r];
            }
            bf16x8 pf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16_t tmp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tmp[e] = cvt16<H16>(pr[8 * j + e]);
                pf[j] = __builtin_bit_cast(bf16x8, tmp);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16_t* vp = Vt + (32 * t + n) * ldv + kc + 16 * j + 4 * h;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 8);
                    O[t] = mfma16<H16>(__builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)), pf[j], O[t]);
                }
        }
        l += __shfl_xor(l, 32);
        __syncthreads();                                      // every wave is through with K and V^T: the LDS takes the partial results
        float* part = reinterpret_cast<float*>(Vt);           // [wave][query 0..31][66] = raw maximum, sum, O[64]: 67.6 KB <= V^T + K
        if (q < T) {                                          // only the left-over queries (ONE at T = 257) leave anything
            float* pq = part + (wave * 32 + n) * 66;
            if (h == 0) { pq[0] = smax; pq[1] = l; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) pq[2 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] = O[t][r];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * 64; idx += 64 * ATT_NW) {
            const int ql = idx >> 6, d = idx & 63, qq = qb * 32 + ql;
            if (qq >= T) continue;
            float M = -3.0e38f;
#pragma unroll
            for (int w = 0; w < ATT_NW; ++w) M = fmaxf(M, part[(w * 32 + ql) * 66]);
            float L = 0.f, acc = 0.f;
#pragma unroll
            for (int w = 0; w < ATT_NW; ++w) {                // wave order: a fixed summation order
                const float f = __builtin_amdgcn_exp2f((part[(w * 32 + ql) * 66] - M) * c2);
                L += part[(w * 32 + ql) * 66 + 1] * f;
                acc += part[(w * 32 + ql) * 66 + 2 + d] * f;
            }
            out[((size_t)b * T + qq) * D + hd * 64 + d] = cvt16<H16>(acc / L);
        }
    }
}

// T <= 64 (ViT-B/32: 50 tokens): the whole key range is two 32-key chunks, so nothing has to be walked twice.  Two waves per (image, head)
// -- one per 32-query block; the four-wave kernel above left two of them idle here -- request Q and ALL K fragments in one round
// trip, keep both score tiles in registers, take the softmax statistics from them and feed P straight into P V: one pass, 8 + 8 MFMAs per
// wave.  (The general kernel recomputes S in its second pass and re-reads K from global memory: 45 us per layer at batch 256.)
template <bool H16>
__global__ __launch_bounds__(128) void attention_small_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int D,
                                                              int heads, float scale) {
    constexpr int Tp = 64, ldv = Tp + 4;
    __shared__ bf16_t Vt[64 * ldv];                   // V transposed: [head dim][key]
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const size_t rs = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + (size_t)hd * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    auto frag = [&](int row, int col_off) -> bf16x8 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < T) v = *reinterpret_cast<const uint4*>(base + (size_t)row * rs + col_off + 8 * h);
        return __builtin_bit_cast(bf16x8, v);
    };
    const int q = wave * 32 + n;
    bf16x8 qf[4], kf[2][4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        qf[s4] = frag(q, 16 * s4);
        kf[0][s4] = frag(n, D + 16 * s4);
        kf[1][s4] = frag(32 + n, D + 16 * s4);
    }
    for (int e = threadIdx.x; e < Tp * 8; e += 128) {
        const int key = e >> 3, dc = (e & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < T) v = *reinterpret_cast<const uint4*>(base + (size_t)key * rs + 2 * D + dc);
        const bf16_t* pv = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(dc + j) * ldv + key] = pv[j];
    }
    const float NEG = -3.0e38f;
    float v[2][16];
    float m = NEG;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) S = mfma16<H16>(kf[c][s4], qf[s4], S);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
            v[c][r] = key < T ? S[r] * scale : NEG;
            m = fmaxf(m, v[c][r]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[c][r] = v[c][r] > NEG ? __expf(v[c][r] - m) : 0.f;
            l += v[c][r];
        }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    __syncthreads();                                   // Vt is complete
    f32x16 O[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16_t tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = cvt16<H16>(v[c][8 * j + e] * inv);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, tmp);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16_t* vp = Vt + (32 * t + n) * ldv + 32 * c + 16 * j + 4 * h;
                const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 8);
                const uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
                O[t] = mfma16<H16>(__builtin_bit_cast(bf16x8, pk), pf, O[t]);          // O^T[dim][query], as in attention_kernel
            }
        }
    if (q < T) {                                        // four consecutive dims per r >> 2: 8-byte stores
        bf16_t* orow = out + ((size_t)b * T + q) * D + hd * 64 + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const uint32_t lo = (uint32_t)cvt16<H16>(O[t][4 * r4]) | ((uint32_t)cvt16<H16>(O[t][4 * r4 + 1]) << 16);
                const uint32_t hi = (uint32_t)cvt16<H16>(O[t][4 * r4 + 2]) | ((uint32_t)cvt16<H16>(O[t][4 * r4 + 3]) << 16);
                *reinterpret_cast<uint2*>(orow + 32 * t + 8 * r4) = make_uint2(lo, hi);
            }
    }
}

}  // namespace sc
#include "clip_cluster.hpp"
namespace sc {

template <bool H16>
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = cvt16<H16>(x[i]);
}

// The last rows of a GEMM whose row count is not a multiple of the 128-row tile (ViT-L/14: 257 tokens x 32 images = 64 tiles + 32 rows).  Left
// to the tiled kernels they cost a full extra tile per 128 columns -- and with 8 / 24 / 32 such tiles on top of exactly 1 / 3 / 4 rounds of
// the 512 resident workgroups (proj + fc2 / qkv / fc1 of ViT-L/14 at batch 32) an almost empty extra ROUND: 520 tile slots of work on 512.
// Here: one workgroup per 32 x 32 block of the result, its sixteen waves split K, operands straight from global memory into the MFMA
// (no LDS staging: nothing is reused inside a wave), the sixteen partial blocks added in wave order through LDS (fixed order).
constexpr int THIN_NW = 16;
template <int EPI, bool H16>
__global__ __launch_bounds__(64 * THIN_NW) void gemm_thin_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                                 void* __restrict__ out, int M, int N, int K) {
    __shared__ float red[THIN_NW][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int bm = blockIdx.y * 32, bn = blockIdx.x * 32;
    const int kw = K / THIN_NW;                                      // K % 256 == 0: a multiple of 16 per wave
    const bf16_t* ap = A + (size_t)min(bm + n, M - 1) * K + wave * kw + 8 * h;
    const bf16_t* wp = Wt + (size_t)(bn + n) * K + wave * kw + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < kw; k += 128) {                              // eight K-steps per trip while they last: 16 loads in flight per lane
        uint4 a[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k + 16 * u < kw) {
                a[u] = *reinterpret_cast<const uint4*>(ap + k + 16 * u);
                w[u] = *reinterpret_cast<const uint4*>(wp + k + 16 * u);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k + 16 * u < kw) acc = mfma16<H16>(__builtin_bit_cast(bf16x8, a[u]), __builtin_bit_cast(bf16x8, w[u]), acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    {
        const int rr = wave;                                         // one accumulator row set per wave
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < THIN_NW; ++s) v += red[s][rr][lane];
        const int row = bm + (rr & 3) + 8 * (rr >> 2) + 4 * h, col = bn + n;
        if (row < M) {
            v += bias ? bias[col] : 0.f;
            const size_t o = (size_t)row * N + col;
            if (EPI == EPI_F32) reinterpret_cast<float*>(out)[o] = v;
            else if (EPI == EPI_RESID) reinterpret_cast<float*>(out)[o] += v;
            else if (EPI == EPI_GELU_BF16) reinterpret_cast<bf16_t*>(out)[o] = cvt16<H16>(v / (1.f + __expf(-1.702f * v)));
            else reinterpret_cast<bf16_t*>(out)[o] = cvt16<H16>(v);
        }
    }
}

template <bool H16>
static int launch_gemm(int epi, const bf16_t* A, const bf16_t* Wt, const float* bias, void* out, int M, int N, int K,
                       hipStream_t st) {
    if (K % 64) return (int)hipErrorInvalidValue;
    // 128-wide tiles when they give every CU work (>= 1.5 workgroups per CU), 64-wide ones otherwise
    const long long t128 = (long long)((N + 127) / 128) * ((M + 127) / 128);
    static const long long t128_min = [] { const char* e = getenv("SC_GEMM_T128_MIN"); return e ? atoll(e) : 384LL; }();   // tuning override
    const bool small = t128 < t128_min;
    // 256 x 256 tiles (half the L2 -> LDS bytes per FLOP) when there are enough of them to give most CUs one
    const long long t256 = (long long)(N / 256) * ((M + 255) / 256);
    static const long long t256_min = [] { const char* e = getenv("SC_GEMM_T256_MIN"); return e ? atoll(e) : 128LL; }();          // tuning override
    // ... and when the last round of the one-workgroup-per-CU grid is not mostly empty (qkv at 12,800 tokens: 450 tiles = 88 % of two rounds:
    // 66 -> 60 us; fc1: 600 tiles = 78 % of three rounds: slower than the persistent 128-wide kernel, which balances its tail)
    // (also tried: cutting whole rounds of 256 x 256 tiles off the top of fc1 -- 504 tiles = two rounds at 98 % -- and leaving 2,048 rows to the
    // persistent kernel: 4.17 instead of 4.09 ms per tower; and the 256 x 256 kernel for fc1 of ViT-L/14 with its 32 ragged rows peeled: no change)
    const long long rounds256 = (t256 + 255) / 256;
    static const long long fill_pct = [] { const char* e = getenv("SC_GEMM_FILL_PCT"); return e ? atoll(e) : 85LL; }();           // tuning override
    const bool big = (N % 256) == 0 && M >= 2048 && t256 >= t256_min && t256 * 100 >= rounds256 * 256 * fill_pct;
    // persistent kernel: a ragged last row tile that would open another round of the 512 resident workgroups goes to gemm_thin_kernel
    const int rem = M % 128;
    const long long t_full = (long long)((N + 127) / 128) * (M / 128);
    static const int peel_on = [] { const char* e = getenv("SC_GEMM_PEEL"); return e ? atoi(e) : 1; }();                          // tuning override
    const bool peel = peel_on && rem != 0 && t_full > 0 && (N % 32) == 0 && (K % 256) == 0 && (t128 + 511) / 512 > (t_full + 511) / 512;
    // ... and a last round that would be less than a quarter full (proj / fc2 at 12,800 tokens: 600 tiles = 512 + 88) is not opened either: the
    // row tiles it consists of go to the 64 x 64 kernel, whose 4 workgroups per CU spread them over the whole chip
    static const int tail_pct = [] { const char* e = getenv("SC_GEMM_TAIL_PCT"); return e ? atoi(e) : 25; }();                    // tuning override
    const int ntn128 = (N + 127) / 128;
    const long long main_mt = (t128 / 512) * 512 / ntn128;            // whole row tiles inside the full rounds
    const bool tail64 = !peel && rem == 0 && t128 > 512 && (t128 % 512) != 0 && (t128 % 512) * 100 < 512 * tail_pct && main_mt > 0;
    const int M0t = (int)main_mt * 128;
    // Round 4: the hand-scheduled 256 x 256 kernel of gemm8p.hpp (two wave groups a barrier apart, continuous LDS-DMA stream with counted
    // vmcnt, persistent tiles) where its tile count fills the chip: measured against the kernels below (tools/micro/gemm_lab) it wins from
    // ~200 tiles up (12,800 x 2,304 x 768: 58.7 vs 62.9 us; 12,800 x 3,072 x 768: 82.9 vs 105.9; ViT-L/14 qkv / fc1: 67.5 vs 79.7, 98.2 vs
    // 115.7) and loses below (N = 768 / 1,024 at these M: 150 / 132 tiles on 256 CUs).
    static const long long g8_min = [] { const char* e = getenv("SC_GEMM_G8_MIN"); return e ? atoll(e) : 200LL; }();              // tuning override
    if ((N % 256) == 0 && M >= 2048 && t256 >= g8_min && (unsigned long long)(M + 256) * N * 4ull < (1ull << 32) &&
        (unsigned long long)(M + 256) * K * 2ull < (1ull << 32) && (unsigned long long)N * K * 2ull < (1ull << 32)) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        static int cu_cache[64] = {0};
        if (dev >= 0 && dev < 64 && cu_cache[dev]) cus = cu_cache[dev];
        else {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipGetLastError();
            if (dev >= 0 && dev < 64) cu_cache[dev] = cus;
        }
        // A ragged last row tile that opens another ROUND of the one-workgroup-per-CU grid goes to gemm_thin_kernel instead (ViT-L/14 at batch
        // 32: 8,224 rows = 32 row tiles + 32 rows; fc1 = 33 x 16 = 528 tiles = three rounds of 256, 32 x 16 = 512 = exactly two)
        const int rem8 = M % 256, ntn8 = N / 256;
        const long long t_main = (long long)(M / 256) * ntn8, gridw = (cus / 8) * 8;
        if (rem8 != 0 && rem8 <= 64 && t_main > 0 && gridw > 0 && (N % 32) == 0 && (K % 256) == 0 &&
            (t256 + gridw - 1) / gridw > (t_main + gridw - 1) / gridw) {
            const int rc8 = g8::launch_gemm8p<H16>(epi, A, Wt, bias, out, M - rem8, N, K, cus, st);
            if (rc8) return rc8;
            const size_t esz = (epi == EPI_F32 || epi == EPI_RESID) ? 4 : 2;
            const bf16_t* A2 = A + (size_t)(M - rem8) * K;
            void* out2 = (char*)out + (size_t)(M - rem8) * N * esz;
            switch (epi) {
                case EPI_F32: hipLaunchKernelGGL((gemm_thin_kernel<EPI_F32, H16>), dim3(N / 32, (rem8 + 31) / 32), dim3(64 * THIN_NW), 0, st, A2, Wt, bias, out2, rem8, N, K); break;
                case EPI_RESID: hipLaunchKernelGGL((gemm_thin_kernel<EPI_RESID, H16>), dim3(N / 32, (rem8 + 31) / 32), dim3(64 * THIN_NW), 0, st, A2, Wt, bias, out2, rem8, N, K); break;
                case EPI_GELU_BF16: hipLaunchKernelGGL((gemm_thin_kernel<EPI_GELU_BF16, H16>), dim3(N / 32, (rem8 + 31) / 32), dim3(64 * THIN_NW), 0, st, A2, Wt, bias, out2, rem8, N, K); break;
                default: hipLaunchKernelGGL((gemm_thin_kernel<EPI_BF16, H16>), dim3(N / 32, (rem8 + 31) / 32), dim3(64 * THIN_NW), 0, st, A2, Wt, bias, out2, rem8, N, K); break;
            }
            return (int)hipGetLastError();
        }
        return g8::launch_gemm8p<H16>(epi, A, Wt, bias, out, M, N, K, cus, st);
    }
    // ... and its 256 x 192 form for the fp32-output GEMMs with N = 768 (proj / fc2 of ViT-B/32 at 12,800 tokens: 150 tiles of 256 x 256 on 256
    // CUs, 200 of 256 x 192: 32.4 vs 34.4 us and 78.6 vs 90.9 us against the kernels below)
    const long long t192 = (long long)(N / 192) * ((M + 255) / 256);
    if ((epi == EPI_F32 || epi == EPI_RESID) && (N % 192) == 0 && M >= 2048 && t192 >= 180 && t192 <= 256 && g8_min < (1LL << 40) &&
        (unsigned long long)(M + 256) * N * 4ull < (1ull << 32) && (unsigned long long)(M + 256) * K * 2ull < (1ull << 32) &&
        (unsigned long long)N * K * 2ull < (1ull << 32)) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipGetLastError();
        return g8::launch_gemm8p<H16, 1>(epi, A, Wt, bias, out, M, N, K, cus, st);
    }
#define SC_LAUNCH(E)                                                                                                          \
    if (big) {                                                                                                         \
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<E, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);        \
        hipLaunchKernelGGL((gemm256_kernel<E, H16>), dim3(N / 256, (M + 255) / 256), dim3(512), 131072, st, A, Wt, bias, out, M, N, K); \
    } else if (small) {                                                                                                       \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<E, 64, H16>, hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                   SC_GEMM64_STAGES * 2 * 64 * 8 * 16);                                                       \
        hipLaunchKernelGGL((gemm_bf16_kernel<E, 64, H16>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256),                          \
                           SC_GEMM64_STAGES * 2 * 64 * 8 * 16, st, A, Wt, bias, out, M, N, K);                                \
    } else if ((N % 8) == 0) {          /* persistent: 2 resident workgroups per CU walk the tile list */                     \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_persist_kernel<E, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
        hipLaunchKernelGGL((gemm_bf16_persist_kernel<E, H16>), dim3(512), dim3(256), 65536, st, A, Wt, bias, out,                  \
                           peel ? M - rem : tail64 ? M0t : M, N, K, (N + 127) / 128, (int)(peel ? t_full : tail64 ? main_mt * ntn128 : t128)); \
        if (tail64) {                                                                                                         \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<E, 64, H16>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                       SC_GEMM64_STAGES * 2 * 64 * 8 * 16);                                                   \
            hipLaunchKernelGGL((gemm_bf16_kernel<E, 64, H16>), dim3((N + 63) / 64, (M - M0t + 63) / 64), dim3(256),                \
                               SC_GEMM64_STAGES * 2 * 64 * 8 * 16, st, A + (size_t)M0t * K, Wt, bias,                         \
                               (char*)out + (size_t)M0t * N * ((E) == EPI_F32 || (E) == EPI_RESID ? 4 : 2), M - M0t, N, K);    \
        }                                                                                                                     \
        if (peel)                                                                                                             \
            hipLaunchKernelGGL((gemm_thin_kernel<E, H16>), dim3(N / 32, (rem + 31) / 32), dim3(64 * THIN_NW), 0, st, A + (size_t)(M - rem) * K, Wt, bias, \
                               (char*)out + (size_t)(M - rem) * N * ((E) == EPI_F32 || (E) == EPI_RESID ? 4 : 2), rem, N, K);  \
    } else {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<E, 128, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);  \
        hipLaunchKernelGGL((gemm_bf16_kernel<E, 128, H16>), dim3((N + 127) / 128, (M + 127) / 128), dim3(256), 65536, st,          \
                           A, Wt, bias, out, M, N, K);                                                                        \
    }
    switch (epi) {
        case EPI_F32: SC_LAUNCH(EPI_F32) break;
        case EPI_RESID: SC_LAUNCH(EPI_RESID) break;
        case EPI_GELU_BF16: SC_LAUNCH(EPI_GELU_BF16) break;
        default: SC_LAUNCH(EPI_BF16) break;
    }
#undef SC_LAUNCH
    return (int)hipGetLastError();
}

// batch range of the cluster form (sc_clip_cluster_set_batch_range; environment SC_CLIP_CLUSTER_MIN_B / _MAX_B at load time)
static std::atomic<int> g_cluster_min_b{[] { const char* e = getenv("SC_CLIP_CLUSTER_MIN_B"); return e ? atoi(e) : 26; }()};
static std::atomic<int> g_cluster_max_b{[] { const char* e = getenv("SC_CLIP_CLUSTER_MAX_B"); return e ? atoi(e) : 32; }()};

// Full image tower (H16: fp16 instead of bf16 operands).  See include/shapeclipper_hip.h for the weight image layout.
template <bool H16>
static int clip_vit_forward(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers, int heads,
                            int proj_dim, const uint16_t* w_bf16, const float* w_f32, const uint16_t* w_cluster, float ln_eps, float* out,
                            void* workspace, long long workspace_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (D % 64 || D / heads != 64 || mlp % 64) return (int)hipErrorInvalidValue;
    const int np = (H / patch) * (W / patch), T = np + 1, M = B * T;
    const int Kp = (C * patch * patch + 63) & ~63;      // patch-embedding K padded to the GEMM's K granularity (zeros)
    // workspace carve (all 256-byte aligned)
    char* ws = (char*)workspace;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* p = ws + off; off += (bytes + 255) & ~(size_t)255; return p; };
    bf16_t* a_patch = (bf16_t*)carve((size_t)B * np * Kp * 2);
    float* patch_out = (float*)carve((size_t)B * np * D * 4);
    float* x = (float*)carve((size_t)M * D * 4);
    bf16_t* xn = (bf16_t*)carve((size_t)M * D * 2);
    bf16_t* qkv = (bf16_t*)carve((size_t)M * 3 * D * 2);
    bf16_t* att = (bf16_t*)carve((size_t)M * D * 2);
    bf16_t* hbuf = (bf16_t*)carve((size_t)M * mlp * 2);
    bf16_t* pooled = (bf16_t*)carve((size_t)B * D * 2);
    unsigned* cl_sync = (unsigned*)carve(33 * 128);       // clip_cluster.hpp: error word + one counter line per cluster
    if ((long long)off > workspace_bytes) return (int)hipErrorInvalidValue;
    // weight images: bf16 matrices then fp32 vectors, in this fixed order
    const bf16_t* wb = w_bf16;
    const float* wf = w_f32;
    const bf16_t* w_patch = wb; wb += (size_t)D * Kp;
    const float* cls = wf; wf += D;
    const float* pos = wf; wf += (size_t)T * D;
    const float* lnpre_g = wf; wf += D;
    const float* lnpre_b = wf; wf += D;

    if ((long long)B * np * (Kp / 8) >= (1LL << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(patchify_kernel<H16>, dim3(4096), dim3(256), 0, st, image, a_patch, B, C, H, W, patch, Kp);
    const int Tp32 = (T + 31) & ~31, att_v = 64 * (Tp32 + 4) * (int)sizeof(bf16_t), att_k = Tp32 * ATT_KLD * (int)sizeof(bf16_t);
    if (att_v > 160 * 1024) return (int)hipErrorInvalidValue;
    const int k_lds = att_v + att_k <= 160 * 1024 ? 1 : 0;          // T = 257 (ViT-L/14 at 224 x 224): 77 KB, two workgroups per CU
    const int att_lds = att_v + (k_lds ? att_k : 0);
    if (att_lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<H16>), hipFuncAttributeMaxDynamicSharedMemorySize, att_lds);
    int rc = launch_gemm<H16>(EPI_F32, a_patch, w_patch, nullptr, patch_out, B * np, D, Kp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(embed_kernel, dim3(1024), dim3(256), 0, st, patch_out, cls, pos, x, B, T, D);
    hipLaunchKernelGGL((layernorm_kernel<false, H16>), dim3((M + 3) / 4), dim3(256), 0, st, x, D, lnpre_g, lnpre_b, (void*)x, M, D, ln_eps);
    // One batch range of the ViT-B geometry runs all its layers in one launch, an image per cluster of 8 CUs (clip_cluster.hpp).  Measured
    // (profiles/r05_clip_cluster_ab.txt): below ~26 images the launch-per-operation form is faster (its GEMMs are small), above 32 the
    // cluster form needs a second round of the chip.
    const int cl_min_b = g_cluster_min_b.load(), cl_max_b = g_cluster_max_b.load();
    int layers_left = layers;
    if (w_cluster != nullptr && D == cl::CD && mlp == cl::CMLP && heads == cl::CHEADS && T <= 64 && B >= cl_min_b && B <= cl_max_b) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipGetLastError();
        if (cus >= 256) {                                   // 32 clusters resident at once: one workgroup per CU
            if ((rc = cl::launch_layers_cluster<H16>(x, qkv, att, hbuf, w_cluster, wf, layers, B, T, ln_eps, cl_sync, st))) return rc;
            wb += (size_t)layers * 12 * D * D;
            wf += (size_t)layers * (9 * D + mlp);
            layers_left = 0;
        }
    }
    for (int l = 0; l < layers_left; ++l) {
        const bf16_t* w_qkv = wb; wb += (size_t)3 * D * D;
        const bf16_t* w_o = wb; wb += (size_t)D * D;
        const bf16_t* w_fc1 = wb; wb += (size_t)mlp * D;
        const bf16_t* w_fc2 = wb; wb += (size_t)D * mlp;
        const float* ln1_g = wf; wf += D;
        const float* ln1_b = wf; wf += D;
        const float* b_qkv = wf; wf += 3 * D;
        const float* b_o = wf; wf += D;
        const float* ln2_g = wf; wf += D;
        const float* ln2_b = wf; wf += D;
        const float* b_fc1 = wf; wf += mlp;
        const float* b_fc2 = wf; wf += D;
        hipLaunchKernelGGL((layernorm_kernel<true, H16>), dim3((M + 3) / 4), dim3(256), 0, st, x, D, ln1_g, ln1_b, (void*)xn, M, D, ln_eps);
        if ((rc = launch_gemm<H16>(EPI_BF16, xn, w_qkv, b_qkv, qkv, M, 3 * D, D, st))) return rc;
        if (T <= 64 && T > 32) hipLaunchKernelGGL(attention_small_kernel<H16>, dim3(B * heads), dim3(128), 0, st, qkv, att, T, D, heads, 0.125f);
        else hipLaunchKernelGGL(attention_kernel<H16>, dim3(B * heads), dim3(64 * ATT_NW), att_lds, st, qkv, att, T, D, heads, 0.125f, k_lds);
        if ((rc = launch_gemm<H16>(EPI_RESID, att, w_o, b_o, x, M, D, D, st))) return rc;
        hipLaunchKernelGGL((layernorm_kernel<true, H16>), dim3((M + 3) / 4), dim3(256), 0, st, x, D, ln2_g, ln2_b, (void*)xn, M, D, ln_eps);
        if ((rc = launch_gemm<H16>(EPI_GELU_BF16, xn, w_fc1, b_fc1, hbuf, M, mlp, D, st))) return rc;
        if ((rc = launch_gemm<H16>(EPI_RESID, hbuf, w_fc2, b_fc2, x, M, D, mlp, st))) return rc;
    }
    const bf16_t* w_proj = wb;
    const float* lnpost_g = wf; wf += D;
    const float* lnpost_b = wf; wf += D;
    // ln_post on the class token of every image (row stride T*D), then the projection (no bias)
    hipLaunchKernelGGL((layernorm_kernel<true, H16>), dim3((B + 3) / 4), dim3(256), 0, st, x, T * D, lnpost_g, lnpost_b, (void*)pooled, B, D, ln_eps);
    return launch_gemm<H16>(EPI_F32, pooled, w_proj, nullptr, out, B, proj_dim, D, st);
}

}  // namespace sc

extern "C" {

// y[n] = bf16(x[n])  (weight preparation, once per model)
int sc_f32_to_bf16(const float* x, uint16_t* y, long long n, void* stream_) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sc::f32_to_bf16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, y, (size_t)n);
    return (int)hipGetLastError();
}
// y[n] = fp16(x[n]), round to nearest even
int sc_f32_to_f16(const float* x, uint16_t* y, long long n, void* stream_) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sc::f32_to_bf16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, y, (size_t)n);
    return (int)hipGetLastError();
}

// Generic bf16 GEMM used by the tower (exported for tests): out[M,N] (epi 0 fp32 / 1 fp32 += / 2 quick_gelu bf16 / 3 bf16)
int sc_gemm_bf16(int epi, const uint16_t* A, const uint16_t* Wt, const float* bias, void* out, int M, int N, int K, void* stream_) {
    return sc::launch_gemm<false>(epi, A, Wt, bias, out, M, N, K, (hipStream_t)stream_);
}
// The same GEMM with IEEE fp16 operands (A, Wt and the 16-bit outputs of epilogues 2 / 3 are fp16 bit patterns)
int sc_gemm_f16(int epi, const uint16_t* A, const uint16_t* Wt, const float* bias, void* out, int M, int N, int K, void* stream_) {
    return sc::launch_gemm<true>(epi, A, Wt, bias, out, M, N, K, (hipStream_t)stream_);
}

#if SC_CL_PROF
// profiling build only: the phase stamps of workgroup 0 (12 per layer, 16 layers x 16 slots of 100 MHz ticks)
int sc_clip_cluster_prof_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sc::cl::g_prof), sizeof(long long) * 256);
}
#endif

// Bytes of device workspace sc_clip_vit_forward carves for this geometry (same carve order and 256-byte alignment).
long long sc_clip_vit_workspace_bytes(int B, int C, int H, int W, int patch, int D, int mlp) {
    const long long np = (long long)(H / patch) * (W / patch), T = np + 1, M = (long long)B * T;
    const long long Kp = ((long long)C * patch * patch + 63) & ~63LL;
    const long long sizes[9] = {B * np * Kp * 2, B * np * D * 4, M * D * 4, M * D * 2, M * 3 * D * 2, M * D * 2, M * mlp * 2,
                                (long long)B * D * 2, 33 * 128};
    long long total = 0;
    for (long long s : sizes) total += (s + 255) & ~255LL;
    return total;
}

// Full image tower, bf16 operands.  See include/shapeclipper_hip.h for the weight image layout.
int sc_clip_vit_forward(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers, int heads,
                        int proj_dim, const uint16_t* w_bf16, const float* w_f32, float ln_eps, float* out,
                        void* workspace, long long workspace_bytes, void* stream_) {
    return sc::clip_vit_forward<false>(image, B, C, H, W, patch, D, mlp, layers, heads, proj_dim, w_bf16, w_f32, nullptr, ln_eps, out, workspace,
                                       workspace_bytes, stream_);
}
// The same tower with IEEE fp16 operands (`w_f16`: the 16-bit weight image as fp16 bit patterns, same order) -- the arithmetic
// openai/CLIP uses on a GPU (CLIP_anno.py:16: fp16 weights and activations, fp32 LayerNorm); fp32 accumulation in the MFMAs.
int sc_clip_vit_forward_f16(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers, int heads,
                            int proj_dim, const uint16_t* w_f16, const float* w_f32, float ln_eps, float* out,
                            void* workspace, long long workspace_bytes, void* stream_) {
    return sc::clip_vit_forward<true>(image, B, C, H, W, patch, D, mlp, layers, heads, proj_dim, w_f16, w_f32, nullptr, ln_eps, out, workspace,
                                      workspace_bytes, stream_);
}


// ---- small-batch form of the ViT-B layers (clip_cluster.hpp): one launch for all layers, an image per cluster of 8 CUs ----------------
// 1 if the cluster form takes this geometry (width 768, MLP 3072, 12 heads, at most 64 tokens per image), else 0
int sc_clip_cluster_supported(int D, int mlp, int heads, int tokens) {
    return D == sc::cl::CD && mlp == sc::cl::CMLP && heads == sc::cl::CHEADS && tokens >= 1 && tokens <= 64 ? 1 : 0;
}
// Batches of min_b <= B <= max_b images take the cluster form (max_b < min_b: never).  Returns 0.
int sc_clip_cluster_set_batch_range(int min_b, int max_b) {
    sc::g_cluster_min_b.store(min_b);
    sc::g_cluster_max_b.store(max_b);
    return 0;
}
// 16-bit values of the re-packed layer image: layers x 12 x 768 x 768 (the same values as the row-major layer matrices, other order)
long long sc_clip_cluster_pack_elems(int layers) { return (long long)layers * (long long)sc::cl::LAYER_ELEMS; }
// w16: the tower's 16-bit weight image (layout of sc_clip_vit_forward); Kp = patch-embedding K padded to 64 (the layers start at D * Kp)
int sc_clip_cluster_pack(const uint16_t* w16, int Kp, int layers, uint16_t* out, void* stream_) {
    if (layers <= 0 || Kp <= 0) return (int)hipErrorInvalidValue;
    return sc::cl::launch_cluster_pack(w16 + (size_t)sc::cl::CD * Kp, out, layers, (hipStream_t)stream_);
}
// sc_clip_vit_forward / _f16 with the re-packed layer image beside the row-major one (`fp16` != 0: IEEE fp16 operands, else bf16): batches
// of SC_CLIP_CLUSTER_MIN_B..SC_CLIP_CLUSTER_MAX_B (default 26..32) images run their layers in the cluster form, larger ones exactly as sc_clip_vit_forward.
int sc_clip_vit_forward_packed(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers, int heads, int proj_dim,
                               const uint16_t* w16, const float* w_f32, const uint16_t* w_cluster, int fp16, float ln_eps, float* out,
                               void* workspace, long long workspace_bytes, void* stream_) {
    if (fp16)
        return sc::clip_vit_forward<true>(image, B, C, H, W, patch, D, mlp, layers, heads, proj_dim, w16, w_f32, w_cluster, ln_eps, out, workspace,
                                          workspace_bytes, stream_);
    return sc::clip_vit_forward<false>(image, B, C, H, W, patch, D, mlp, layers, heads, proj_dim, w16, w_f32, w_cluster, ln_eps, out, workspace,
                                       workspace_bytes, stream_);
}

}  // extern "C"
