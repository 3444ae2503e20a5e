// gemm_rect.hpp -- EXPERIMENT (round 4, not part of the library; `tools/micro/gemm_lab.bin rect`): rectangular tiles with one workgroup per CU
// and deep LDS-DMA staging for the small-M GEMMs of the CLIP tower (ViT-B/32 at batch 32: 1,600 token rows).  Result: correct and NOT faster than
// the 64 x 64 three-stage kernel with four workgroups per CU (qkv 13.1 vs 14.3 us with 128 x 128; fc1 23.4 vs 19.4 with 128 x 256; fc2 28.2 vs 21.0
// with 64 x 128; proj 10.4 vs 8.0): with four waves per CU nothing overlaps the ds_read -> MFMA chain of a K-step, which costs more than the
// operand bytes saved.  Kept for the record (profiles/r04_gemm_rect_experiment.txt).  Original header:
// the SMALL-M 16-bit GEMM of the CLIP tower (ViT-B/32 at batch 32: 1,600 token rows; CLIP_anno.py:166), round 4.
//
//   C[M,N] = A[M,K] W[N,K]^T (+ bias, epilogue)          16-bit operands (bf16 or IEEE fp16), fp32 accumulate
//
// Why another kernel.  At 1,600 rows the 64 x 64 tiles of gemm_bf16_kernel (clip_vit.hip) give every CU work, but they pull 2 x 64 x 64 K-slices
// per 64 x 64 x 64 products: 173-236 MB of L2 -> LDS traffic per GEMM, i.e. ~0.7-0.9 MB per CU at the ~100 GB/s a CU streams from L2
// (tools/micro/l2_stream) = 7-9 us of a 14-20 us launch; the MFMA work is a fifth of that.  The lever is bytes per FLOP -- larger, RECTANGULAR
// tiles chosen per GEMM so that the tile count still fills most of the chip once (128 x 128 for N = 2,304: 234 tiles; 128 x 256 for N = 3,072: 156;
// 64 x 128 for N = 768: 150) -- together with enough bytes in flight to cover the L2 round trip with ONE workgroup per CU: NS LDS stages
// (all of the 160 KB), NS - 1 K-steps of LDS-DMA outstanding, counted vmcnt, one raw barrier per K-step.
// Structure otherwise as gemm_bf16_kernel: 4 waves = 2 x 2, each wave (BM/2) x (BN/2) as 32 x 32 x 16 MFMA tiles, operand tiles by LDS-DMA into
// XOR-swizzled stages (slot = chunk ^ ((row >> 1) & 7): conflict-free ds_read_b128), epilogue through the free stages as row-contiguous 16-byte
// stores (the residual epilogue reads its rows before it writes them).  Requires K % 64 == 0 and N % BN == 0; rows past M are clamped / masked.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {
namespace gr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t h16_t;
typedef __attribute__((address_space(3))) void* lptr_t;
enum { EPI_F32 = 0, EPI_RESID = 1, EPI_GELU_BF16 = 2, EPI_BF16 = 3 };

template <bool H>
__device__ __forceinline__ h16_t cvt16(float f) {
    if (H) return __builtin_bit_cast(h16_t, (_Float16)f);
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (h16_t)(u >> 16);
}
template <bool H>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
    if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void glds(const void* sbase, unsigned voff, unsigned lds_dst) {      // see gemm8p.hpp
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int EPI, int BM, int BN, int NS, bool H16>
__global__ __launch_bounds__(256, 1) void gemm_rect_kernel(const h16_t* __restrict__ A, const h16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                           void* __restrict__ out, int M, int N, int K) {
    constexpr int MIM = BM / 64, MIN = BN / 64;            // 32 x 32 MFMA tiles per wave and dimension
    constexpr int CHA = BM * 8, CHB = BN * 8;              // 16-byte chunks of one A / B stage (rows x 64 K)
    constexpr int QA = CHA / 256, QB = CHB / 256;          // LDS-DMA instructions per thread, operand and K-step
    constexpr int STAGE = (CHA + CHB) * 16, PD = NS - 1;   // bytes per stage; K-steps in flight
    static_assert(BM % 64 == 0 && BN % 64 == 0 && NS >= 2 && NS * STAGE <= 160 * 1024 && PD * (QA + QB) < 64, "tile / stage geometry");
    static_assert(BM * BN * 4 <= NS * STAGE, "the fp32 tile is staged through the LDS stages");
    extern __shared__ uint4 Sbuf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD k owns a contiguous range of the row-major (m tile, n tile) list (workgroup b runs on XCD b % 8)
    const int ntn = gridDim.x, tiles = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    const int tq = tiles >> 3, trem = tiles & 7;
    const int tile = (xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq) + idx;
    const int bm = (tile / ntn) * BM, bn = (tile % ntn) * BN;
    const int kt1 = K / 64;
    f32x16 acc[MIM][MIN];
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // per-thread DMA sources (32-bit byte offsets from the operand bases; host: operands < 4 GB): chunk c = q * 256 + tid of the operand tile
    unsigned pa[QA], pb[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int c = q * 256 + tid, row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
        pa[q] = (unsigned)min(bm + row, M - 1) * (unsigned)K * 2u + kc * 16;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int c = q * 256 + tid, row = c >> 3, kc = (c & 7) ^ ((row >> 1) & 7);
        pb[q] = (unsigned)min(bn + row, N - 1) * (unsigned)K * 2u + kc * 16;
    }
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)Sbuf) + (unsigned)wave * 1024u;
    auto issue = [&](int kt, int stage) {
        const unsigned base = lds0 + (unsigned)stage * STAGE;
#pragma unroll
        for (int q = 0; q < QA; ++q) glds(A, pa[q] + (unsigned)kt * 128u, base + q * 4096);
#pragma unroll
        for (int q = 0; q < QB; ++q) glds(Wt, pb[q] + (unsigned)kt * 128u, base + CHA * 16 + q * 4096);
    };
    const int ra = (BM / 2) * wr + (lane & 31), rb = (BN / 2) * wc + (lane & 31);
    const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;              // rows + 32 keep (row >> 1) & 7
    auto compute = [&](int stage) {
        const uint4* As = Sbuf + stage * (CHA + CHB);
        const uint4* Bs = As + CHA;
        uint4 af[2][MIM], bf[2][MIN];
        auto frags = [&](int kk, uint4 (&a2)[MIM], uint4 (&b2)[MIN]) {
            const int kc = 2 * kk + (lane >> 5);
#pragma unroll
            for (int i = 0; i < MIM; ++i) a2[i] = As[(ra + 32 * i) * 8 + (kc ^ sa)];
#pragma unroll
            for (int j = 0; j < MIN; ++j) b2[j] = Bs[(rb + 32 * j) * 8 + (kc ^ sb)];
        };
        frags(0, af[0], bf[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MIM; ++i)
#pragma unroll
                for (int j = 0; j < MIN; ++j) acc[i][j] = mfma16<H16>(af[kk & 1][i], bf[kk & 1][j], acc[i][j]);
        }
    };
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < kt1) issue(d, d);
    int st_c = 0, st_i = PD % NS;                                   // stage consumed at this step / stage the next issue goes to
    for (int kt = 0; kt < kt1; ++kt) {
        // step kt has landed once at most `later` younger DMA groups ((QA + QB) instructions each) of this wave are outstanding
        const int later = min(PD - 1, kt1 - 1 - kt);
        if (later <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(QA + QB) : "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (QA + QB)) : "memory");
        else if (later == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(3 * (QA + QB) > 63 ? 63 : 3 * (QA + QB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * (QA + QB) > 63 ? 63 : 4 * (QA + QB)) : "memory");
        if (kt + PD < kt1) issue(kt + PD, st_i);                   // into the stage read at step kt - 1: everybody is past it (the barrier)
        compute(st_c);
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
        st_i = st_i + 1 == NS ? 0 : st_i + 1;
    }
    // ---- epilogue through the (now free) stages: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    constexpr bool OUT16 = EPI == EPI_GELU_BF16 || EPI == EPI_BF16;
    float* Cf = reinterpret_cast<float*>(Sbuf);
    h16_t* Ch = reinterpret_cast<h16_t*>(Sbuf);
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j) {
            const int cl = (BN / 2) * wc + 32 * j + (lane & 31);
            const float bv = bias ? bias[bn + cl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (BM / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[i][j][r] + bv;
                if (EPI == EPI_GELU_BF16) Ch[rl * BN + cl] = cvt16<H16>(v / (1.f + __expf(-1.702f * v)));
                else if (EPI == EPI_BF16) Ch[rl * BN + cl] = cvt16<H16>(v);
                else Cf[rl * BN + cl] = v;
            }
        }
    __syncthreads();
    if (OUT16) {
        constexpr int CPR = BN / 8, NQ = BM * CPR / 256;            // 16-byte chunks per tile row / per thread
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int chunk = q * 256 + tid, rl = chunk / CPR, c8 = (chunk % CPR) * 8;
            if (bm + rl < M)
                *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(out) + (size_t)(bm + rl) * N + bn + c8) = *reinterpret_cast<const uint4*>(Ch + rl * BN + c8);
        }
    } else {
        constexpr int CPR = BN / 4, NQ = BM * CPR / 256, HQ = NQ > 16 ? 16 : NQ;       // at most 16 float4 of residual in registers at a time
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += HQ) {
            float4 res[HQ];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const int chunk = (q0 + q) * 256 + tid, rl = chunk / CPR, c4 = (chunk % CPR) * 4;
                    res[q] = bm + rl < M ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(out) + (size_t)(bm + rl) * N + bn + c4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int chunk = (q0 + q) * 256 + tid, rl = chunk / CPR, c4 = (chunk % CPR) * 4;
                float4 v = *reinterpret_cast<const float4*>(Cf + rl * BN + c4);
                if (EPI == EPI_RESID) { v.x += res[q].x; v.y += res[q].y; v.z += res[q].z; v.w += res[q].w; }
                if (bm + rl < M) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)(bm + rl) * N + bn + c4) = v;
            }
        }
    }
}

template <int BM, int BN, int NS, bool H16>
static inline int launch_gemm_rect(int epi, const h16_t* A, const h16_t* Wt, const float* bias, void* out, int M, int N, int K, hipStream_t st) {
    if ((K % 64) || (N % BN) || M <= 0 || (unsigned long long)M * K * 2ull >= (1ull << 32) || (unsigned long long)N * K * 2ull >= (1ull << 32))
        return (int)hipErrorInvalidValue;
    constexpr int LDS = NS * (BM + BN) * 128;
#define SC_GR_LAUNCH(E)                                                                                                                \
    do {                                                                                                                               \
        (void)hipFuncSetAttribute((const void*)gemm_rect_kernel<E, BM, BN, NS, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
        hipLaunchKernelGGL((gemm_rect_kernel<E, BM, BN, NS, H16>), dim3(N / BN, (M + BM - 1) / BM), dim3(256), LDS, st, A, Wt, bias, out, M, N, K); \
    } while (0)
    switch (epi) {
        case EPI_F32: SC_GR_LAUNCH(EPI_F32); break;
        case EPI_RESID: SC_GR_LAUNCH(EPI_RESID); break;
        case EPI_GELU_BF16: SC_GR_LAUNCH(EPI_GELU_BF16); break;
        default: SC_GR_LAUNCH(EPI_BF16); break;
    }
#undef SC_GR_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace gr
}  // namespace sc
