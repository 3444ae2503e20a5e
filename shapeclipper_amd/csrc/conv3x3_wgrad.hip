// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions of the ResNet trunks (autograd of the torchvision BasicBlock
// convolutions behind model/graph.py:50-54 and model/view_estimator.py:40-42), fp32, NCHW, on the fp32 matrix pipe:
//     dW[co][ci][ky][kx] = sum over images b and pixels (y, x) of  gy[b][co][y][x] * x[b][ci][y + ky - 1][x + kx - 1]
// as nine GEMMs  D_tap[co][ci] = sum_p A[co][p] B_tap[p][ci]  that share A (= gy) and read B from ONE zero-padded patch of the input.
// The reduction index is the pixel: a workgroup owns a 64 x 64 block of (co, ci) and a contiguous range of K-steps (K-step = 112 /
// 112 / 196 / 98 pixels = 2 rows / 4 rows / 1 image / 2 images of a 56 / 28 / 14 / 7 wide map); its 8 waves are 2 x 2 MFMA tiles x 2
// halves of the K-step's rows, each wave holding the 9 tap accumulators (144 registers) of its 32 x 32 tile.  MIOpen's fp32 weight
// gradient kernels are NHWC-only (three layout transposes per call); this one reads NCHW as it lies.
//
// The inner loop follows the issue-cycle rule of DESIGN.md section 4.1 (every vector / LDS instruction costs ~4.5 cycles the matrix pipe
// cannot hide): the k pair of an MFMA is (pixel x, pixel x + GS) of one row, so a lane needs GS consecutive pixels of gy -- one
// ds_read_b128 / b64 / b32 -- and, per filter row ky, the GS + 2 consecutive patch values that serve all three kx: GS + 2 ds_read_b32
// for 3 GS MFMAs, every address an immediate offset from two per-lane bases.
// Staging: 56 consecutive floats of a channel are 1 / 2 / 4 rows (49 = one 7 x 7 image): every global load is one coalesced row
// chunk per wave with scalar addressing, prefetched into registers during the MFMAs of the previous K-step; LDS is single-buffered
// (two barriers per K-step of >= 16,000 MFMA cycles).
//
// Parallelism comes from the reduction: grid = (Cout / 64) (Cin / 64) blocks x S pixel ranges with S = 256 / blocks; every workgroup
// writes its partial block and conv3x3_wgrad_reduce_kernel adds the S partials of an element in range order (fixed summation order).
// Roofline: 2 * 9 * Cin * Cout FLOP per pixel on the 157.3 TFLOP/s fp32 matrix pipe; one read of gy and of x from HBM.
#include <hip/hip_runtime.h>

#include "grid_cus.hpp"
#include "shapeclipper_hip.h"

namespace sc {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

#ifdef SC_WGRAD_PROFILE          // profile build (tools/prof_wgrad_phases.py; tools/build_variants.sh conv3x3_wgrad.hip SC_WGRAD_PROFILE 1):
                                 // s_memrealtime stamps (100 MHz) of thread 0 of every workgroup of the split kernel
__device__ unsigned long long* wgrad_prof_buf = nullptr;
#define WG_STAMP(ID) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (wgrad_prof_buf && tid == 0) wgrad_prof_buf[(size_t)blockIdx.x * 16 + (ID)] = t_; }
#define WG_ACCUM(ID, T0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (wgrad_prof_buf && tid == 0) wgrad_prof_buf[(size_t)blockIdx.x * 16 + (ID)] += t_ - (T0); }
#else
#define WG_STAMP(ID)
#endif

// S = 2: the weight gradient of the three 3x3 / stride-2 / pad-1 convolutions per trunk (BasicBlock.conv1 of layer2-4): gy lives on the
// W x W OUTPUT map, x on the 2W x 2W input map; output pixel (y, x) meets input (2y + ky - 1, 2x + kx - 1), so the patch of NR output
// rows holds 2 NR + 1 input rows (only a top halo) of 2W + 1 positions (only a left pad), and the 3 kx of GS consecutive output pixels
// are 2 GS + 1 consecutive patch values.
template <int W_, int NR_, int NIMG_, int GS_, int S_ = 1>
struct WgCfg {
    static constexpr int W = W_, H = W_, NR = NR_, NIMG = NIMG_, GS = GS_, S = S_;
    static constexpr int HW = W * W;
    static constexpr int XW = S * W, XH = S * H, XHW = XW * XH;             // the input map
    static constexpr int Wp = S == 1 ? W + 2 : XW + 1;                       // patch row: left pad (+ right pad for stride 1)
    static constexpr bool WHOLE = NR == H;                                   // K-step = whole image(s): the halo rows are always zero
    static constexpr int SLOTW = (W + 2 * GS - 1) / (2 * GS) * (2 * GS);     // k slots per row (7 -> 8: one zero slot)
    static constexpr int ROWS = NIMG * NR;                                   // rows per K-step
    static constexpr int NSLOT = ROWS * SLOTW;
    static constexpr int GST = NSLOT + (((NSLOT / GS) & 1) ? 0 : GS);        // gy row stride: GST / GS odd -> conflict-free operand reads
    static constexpr int PR = S == 1 ? NR + 2 : 2 * NR + 1;                  // patch rows per image
    static constexpr int PATCH = NIMG * PR * Wp;
    static constexpr int LQ = (S == 1 ? PATCH : PATCH + 2 * GS + 2) | 1;     // odd channel stride; a few positions past PATCH may be read (x 0), kept zero
    static constexpr int LDS_FLOATS = 64 * GST + 64 * LQ > 4 * 48 * 64 ? 64 * GST + 64 * LQ : 4 * 48 * 64;   // (>= the final reduction's scratch)
    static constexpr int CHUNK = W == 7 ? 49 : 56;                           // floats per staging load: whole rows, contiguous in memory
    static constexpr int RPC = CHUNK / W;                                    // rows per chunk
    static constexpr int XCHUNK = XW == 7 ? 49 : 56, XRPC = XCHUNK / XW;     // the same for the input map's rows
    // x: rows S y0 - 1 .. (halo rows may be real) unless the K-step is a whole image; gy: the K-step's rows
    static constexpr int XROWS = WHOLE ? XH : PR;
    static constexpr int XCH = (XROWS + XRPC - 1) / XRPC, GCH = (NR + RPC - 1) / RPC;        // chunks per channel and image
    static constexpr int HR = ROWS / 2;                                      // rows per K-step half
    static constexpr int GPR = SLOTW / (2 * GS);                             // MFMA groups per row
    static constexpr int KH_A = HR * SLOTW;                                  // second half: offset in the gy tile ...
    static constexpr int KH_B = NIMG == 2 ? PR * Wp : S * HR * Wp;           // ... and in the patch
    static constexpr int NBV = S == 1 ? GS + 2 : 2 * GS + 1;                 // patch values per filter row that serve GS pixels x 3 kx
    static_assert(ROWS % 2 == 0 && H % NR == 0 && SLOTW % (2 * GS) == 0 && CHUNK % W == 0 && XCHUNK % XW == 0 && (S == 1 || S == 2), "K-step shape");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024 && 4 * 48 * 64 <= LDS_FLOATS, "LDS");
};

template <class C>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                               float* __restrict__ partial, int batch, int cin, int cout, int S) {
    constexpr int W = C::W, H = C::H, HW = C::HW, Wp = C::Wp, GS = C::GS, NIMG = C::NIMG, NR = C::NR, ST = C::S;
    constexpr int XW = C::XW, XH = C::XH, XHW = C::XHW;
    extern __shared__ float4 wg_smem[];
    float* As = reinterpret_cast<float*>(wg_smem);                           // gy tile  [64][GST]
    float* Xs = As + 64 * C::GST;                                            // x patch  [64][LQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
    const int wt = wave & 3, kh = wave >> 2;                                 // MFMA tile (co half, ci half) and K-step half
    const int nti = cin / 64;
    const int blk = blockIdx.x / S, s = blockIdx.x - blk * S;
    const int co0 = (blk / nti) * 64, ci0 = (blk % nti) * 64;
    const int nimg_steps = NIMG == 2 ? (batch + 1) / 2 : batch * (H / NR);  // K-steps of the whole batch
    const int k_lo = (int)((long long)s * nimg_steps / S), k_hi = (int)((long long)(s + 1) * nimg_steps / S);

    // zero what staging never writes: pad columns, always-zero halo rows, the extra position, the zero slots of 7-wide rows
    for (int i = tid; i < C::LDS_FLOATS; i += 512) As[i] = 0.f;

    // staging: wave w copies channels 8 w .. 8 w + 7 of both operands; lane = position inside a CHUNK of whole rows
    const bool lane_on = lane < C::CHUNK, xlane_on = lane < C::XCHUNK;
    const int lrow = lane / W, lx = lane - lrow * W;
    const int xlrow = lane / XW, xlx = lane - xlrow * XW;
    float xv[8][C::XCH * NIMG], gv[8][C::GCH * NIMG];
    auto load = [&](int kstep) {
        int b, y0;
        if (NIMG == 2) { b = 2 * kstep; y0 = 0; } else { b = kstep / (H / NR); y0 = (kstep - b * (H / NR)) * NR; }
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            const bool img_ok = b + im < batch;
            const float* xb = x + ((size_t)(b + im) * cin + ci0 + 8 * wave) * XHW;
            const float* gb = gy + ((size_t)(b + im) * cout + co0 + 8 * wave) * HW;
#pragma unroll
            for (int c = 0; c < C::XCH; ++c) {
                const int yr = (C::WHOLE ? 0 : ST * y0 - 1) + c * C::XRPC + xlrow;    // input-map row of this lane's element
                const bool ok = xlane_on && img_ok && (unsigned)yr < (unsigned)XH && c * C::XRPC + xlrow < C::XROWS;
                const int off = yr * XW + xlx;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) xv[ch][im * C::XCH + c] = ok ? xb[ch * XHW + off] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < C::GCH; ++c) {
                const int yr = y0 + c * C::RPC + lrow;
                const bool ok = lane_on && img_ok && c * C::RPC + lrow < NR;
                const int off = yr * W + lx;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) gv[ch][im * C::GCH + c] = ok ? gb[ch * HW + off] : 0.f;
            }
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
#pragma unroll
            for (int c = 0; c < C::XCH; ++c) {
                const int pr = (C::WHOLE ? 1 : 0) + c * C::XRPC + xlrow;             // patch row
                if (xlane_on && c * C::XRPC + xlrow < C::XROWS) {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) Xs[(8 * wave + ch) * C::LQ + (im * C::PR + pr) * Wp + 1 + xlx] = xv[ch][im * C::XCH + c];
                }
            }
#pragma unroll
            for (int c = 0; c < C::GCH; ++c) {
                if (lane_on && c * C::RPC + lrow < NR) {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) As[(8 * wave + ch) * C::GST + (im * NR + c * C::RPC + lrow) * C::SLOTW + lx] = gv[ch][im * C::GCH + c];
                }
            }
        }
    };

    wg_f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // operand bases: lane = (row of the 32 x 32 tile, k half): the k pair of an MFMA is (pixel, pixel + GS)
    const float* Ab = As + ((wt >> 1) * 32 + (lane & 31)) * C::GST + half * GS + kh * C::KH_A;
    const float* Bb = Xs + ((wt & 1) * 32 + (lane & 31)) * C::LQ + half * ST * GS + kh * C::KH_B;

    if (k_lo < k_hi) load(k_lo);
    __syncthreads();                                                          // zero fill done
    for (int kstep = k_lo; kstep < k_hi; ++kstep) {
        store();
        __syncthreads();
        if (kstep + 1 < k_hi) load(kstep + 1);
#pragma unroll
        for (int yr = 0; yr < C::HR; ++yr) {
            // the second image of a 2-image K-step starts PR patch rows (not NR) after the first: kh already carries that offset
#pragma unroll
            for (int gx = 0; gx < C::GPR; ++gx) {
                float a[GS];
                if constexpr (GS == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(Ab + yr * C::SLOTW + gx * 8);
                    a[0] = t.x, a[1] = t.y, a[2] = t.z, a[3] = t.w;
                } else if constexpr (GS == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(Ab + yr * C::SLOTW + gx * 4);
                    a[0] = t.x, a[1] = t.y;
                } else {
                    a[0] = Ab[yr * C::SLOTW + gx * 2];
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float bv[C::NBV];
#pragma unroll
                    for (int t = 0; t < C::NBV; ++t) bv[t] = Bb[(ST * yr + ky) * Wp + ST * gx * 2 * GS + t];
#pragma unroll
                    for (int ss = 0; ss < GS; ++ss)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ss], bv[ST * ss + kx], acc[ky * 3 + kx], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                      // every wave is done reading this K-step's tiles
    }

    // add the second K-step half onto the first through LDS (three passes of 3 taps), then write the partial block in register order
    float* R = As;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[((wt * 3 + t) * 16 + r) * 64 + lane] = acc[pass * 3 + t][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pass * 3 + t][r] += R[((wt * 3 + t) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (kh == 0) {
        float* dst = partial + (size_t)blockIdx.x * (64 * 64 * 9);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((wt * 9 + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
}

// dW[co][ci][tap] = sum_s partial[block][s][...].  A workgroup handles 64 consecutive elements (register order: one coalesced 256-byte
// row per partial); its four waves take the ranges s = w, w + 4, w + 8, ... and the four sums are added as (0 + 1) + (2 + 3): a
// fixed summation order, independent of timing.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int cin,
                                                                  int cout, int S) {
    __shared__ float part[4][64];
    const int nti = cin / 64, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int blk = blockIdx.x / 576, e = (blockIdx.x - blk * 576) * 64 + lane;         // 576 * 64 = 36864 elements per block
    const float* src = partial + (size_t)blk * S * 36864 + e;
    float s0 = 0.f, s1 = 0.f;
    int s = w;
    for (; s + 4 < S; s += 8) {
        s0 += src[(size_t)s * 36864];
        s1 += src[(size_t)(s + 4) * 36864];
    }
    if (s < S) s0 += src[(size_t)s * 36864];
    part[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0) {
        const float sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        const int r = (e >> 6) & 15, t = (e >> 10) % 9, wt = e / (9 * 1024);
        const int co = (blk / nti) * 64 + (wt >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int ci = (blk % nti) * 64 + (wt & 1) * 32 + (lane & 31);
        dw[((size_t)co * cin + ci) * 9 + t] = sum;
    }
}

using Wg56 = WgCfg<56, 2, 1, 4>;
using Wg28 = WgCfg<28, 4, 1, 2>;
using Wg14 = WgCfg<14, 14, 1, 1>;
using Wg7 = WgCfg<7, 7, 2, 1>;
// stride 2 (template arguments: side of the OUTPUT map = side of gy): 56 -> 28, 28 -> 14, 14 -> 7
using Wg28S2 = WgCfg<28, 2, 1, 2, 2>;
using Wg14S2 = WgCfg<14, 2, 1, 1, 2>;
using Wg7S2 = WgCfg<7, 7, 2, 1, 2>;

// ---------------------------------------------------------------------------------------------------------------------------
// The same weight gradient with fp32-accurate products on the bf16 matrix pipe (the arithmetic of sc_conv3x3_forward_split): every
// operand is staged as its exact three-way bf16 split v = p0 + p1 + p2 (round to nearest even, residuals exact), the six piece
// products a_p b_q with p + q <= 2 go through v_mfma_f32_32x32x16_bf16, smallest terms first, fp32 accumulate.
//
// The reduction index of an MFMA is 16 consecutive pixel SLOTS of one row (rows are padded to a multiple of 16 slots with zeros of
// gy; a 7-wide map packs two rows of 7 + 1 slots): lane half h owns slots 8h .. 8h + 7, so the gy operand is one ds_read_b128 per piece
// and the three kx taps of the patch are the 8-element windows at +0, +1, +2 of ten consecutive bf16 patch values (5 dwords): kx = 0
// and kx = 2 are register renames, kx = 1 is four v_alignbit_b32 -- no per-tap copies of the patch.  A workgroup owns a 64 x 64
// (co, ci) block as 2 x 2 tiles x 3 filter ROWS: its 12 waves hold the three kx accumulators of one (tile, ky), every wave sees the
// whole K-step, and no cross-wave sum is needed at the end.  Per 16 slots and wave: 3 + 15 LDS reads, 12 v_alignbit, 18 MFMAs of 32
// cycles (the fp32 kernel: 21 MFMAs of 64 cycles per 14 pixels and tap row).  K-steps are sized by LDS (6 bytes per element instead
// of 4): 2 rows of a 56-wide map, 4 of 28, 7 of 14, two 7 x 7 images.  Partial blocks and the fixed-order reduction are those above.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
#ifndef SC_WGRAD_S16
#define SC_WGRAD_S16 0          // 1: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (round-6 A/B, tools/build_variants.sh conv3x3_wgrad.hip SC_WGRAD_S16 0 1)
#endif

template <int W_, int NR_, int NIMG_>
struct WsCfg {
    static constexpr int W = W_, H = W_, NR = NR_, NIMG = NIMG_, HW = W * W;
    static constexpr int SLOTS = W == 7 ? 8 : (W + 15) / 16 * 16;             // slots per map row in the gy tile
    static constexpr int ROWS = NIMG * NR, NSLOT = ROWS * SLOTS, G = NSLOT / 16;
    static constexpr int ASTR = NSLOT + 8;                                    // bf16 per channel; ASTR / 8 odd: conflict-free b128 reads
    static constexpr int Wp = SLOTS + 2, PR = NR + 2, PATCH = NIMG * PR * Wp;
    static constexpr int BSTR = (PATCH + 3) / 4 * 4 + 2;                      // bf16 per channel; an odd number of dwords
    static constexpr bool WHOLE = NR == H;
    static constexpr int LDS_BYTES = 3 * 64 * (ASTR + BSTR) * 2;
    static constexpr int CHUNK = W == 7 ? 49 : 56, RPC = CHUNK / W;           // floats per staging load: whole rows, contiguous in memory
    static constexpr int XROWS = WHOLE ? H : PR;
    static constexpr int XCH = (XROWS + RPC - 1) / RPC, GCH = (NR + RPC - 1) / RPC;
    static_assert(NSLOT % 16 == 0 && H % NR == 0 && LDS_BYTES <= 160 * 1024 && (W >= 14 || (NIMG == 2 && NR == 7)), "K-step shape");
    // patch offset (bf16 units, ky = kx = 0) of slot 0 of lane half h in group g
    static constexpr int boff(int g, int h) {
        if (W == 7) { const int r = 2 * g + h; return ((r / 7) * PR + r % 7) * Wp; }
        const int r = g / (SLOTS / 16), gx = g % (SLOTS / 16);
        return ((r / NR) * PR + r % NR) * Wp + 16 * gx + 8 * h;
    }
};

template <class C>
__global__ __launch_bounds__(768, 1) void conv3x3_wgrad_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                     float* __restrict__ partial, int batch, int cin, int cout, int S) {
    constexpr int W = C::W, H = C::H, HW = C::HW, Wp = C::Wp, NIMG = C::NIMG, NR = C::NR;
    extern __shared__ float4 wg_smem[];
    __bf16* As = reinterpret_cast<__bf16*>(wg_smem);                          // gy pieces [3][64][ASTR]
    __bf16* Xs = As + 3 * 64 * C::ASTR;                                       // patch pieces [3][64][BSTR]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
    const int wt = wave & 3, ky = wave >> 2;                                  // MFMA tile (co half, ci half) and filter row
    const int nti = cin / 64;
    const int blk = blockIdx.x / S, s = blockIdx.x - blk * S;
    const int co0 = (blk / nti) * 64, ci0 = (blk % nti) * 64;
    const int nimg_steps = NIMG == 2 ? (batch + 1) / 2 : batch * (H / NR);
    const int k_lo = (int)((long long)s * nimg_steps / S), k_hi = (int)((long long)(s + 1) * nimg_steps / S);

    for (int i = tid; i < C::LDS_BYTES / 16; i += 768) wg_smem[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // pads, zero slots, absent halo rows

    // staging: waves 0-7 copy 8 patch channels each, waves 8-11 copy 16 gy channels each (every SIMD hosts two of the former and one of
    // the latter); lane = position inside a chunk of whole rows; prefetched into registers during the MFMAs of the previous K-step
    const bool lane_on = lane < C::CHUNK, xrole = wave < 8;
    const int lrow = lane / W, lx = lane - lrow * W;
    constexpr int NSV = 8 * C::XCH * NIMG > 16 * C::GCH * NIMG ? 8 * C::XCH * NIMG : 16 * C::GCH * NIMG;
    float sv[NSV];
    auto load = [&](int kstep) {
        int b, y0;
        if (NIMG == 2) { b = 2 * kstep; y0 = 0; } else { b = kstep / (H / NR); y0 = (kstep - b * (H / NR)) * NR; }
        if (xrole) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) {
                const float* xb = x + ((size_t)(b + im) * cin + ci0 + 8 * wave) * HW;
#pragma unroll
                for (int c = 0; c < C::XCH; ++c) {
                    const int rr = c * C::RPC + lrow, yr = (C::WHOLE ? 0 : y0 - 1) + rr;
                    const bool ok = lane_on && b + im < batch && (unsigned)yr < (unsigned)H && rr < C::XROWS;
                    const int off = yr * W + lx;
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) sv[(im * C::XCH + c) * 8 + ch] = ok ? xb[ch * HW + off] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) {
                const float* gb = gy + ((size_t)(b + im) * cout + co0 + 16 * (wave - 8)) * HW;
#pragma unroll
                for (int c = 0; c < C::GCH; ++c) {
                    const int rr = c * C::RPC + lrow;
                    const bool ok = lane_on && b + im < batch && rr < NR;
                    const int off = (y0 + rr) * W + lx;
#pragma unroll
                    for (int ch = 0; ch < 16; ++ch) sv[(im * C::GCH + c) * 16 + ch] = ok ? gb[ch * HW + off] : 0.f;
                }
            }
        }
    };
    auto put = [&](__bf16* d, int stride, float v) {      // v = p0 + p1 + p2 exactly, each piece a bf16 (round to nearest even, residuals exact)
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        d[0] = h0; d[stride] = h1; d[2 * stride] = (__bf16)(r1 - (float)h1);
    };
    auto store = [&]() {
        if (xrole) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im)
#pragma unroll
                for (int c = 0; c < C::XCH; ++c) {
                    const int rr = c * C::RPC + lrow;
                    if (lane_on && rr < C::XROWS) {
                        __bf16* d = Xs + 8 * wave * C::BSTR + (im * C::PR + (C::WHOLE ? 1 : 0) + rr) * Wp + 1 + lx;
#pragma unroll
                        for (int ch = 0; ch < 8; ++ch) put(d + ch * C::BSTR, 64 * C::BSTR, sv[(im * C::XCH + c) * 8 + ch]);
                    }
                }
        } else {
#pragma unroll
            for (int im = 0; im < NIMG; ++im)
#pragma unroll
                for (int c = 0; c < C::GCH; ++c) {
                    const int rr = c * C::RPC + lrow;
                    if (lane_on && rr < NR) {
                        __bf16* d = As + 16 * (wave - 8) * C::ASTR + (im * NR + rr) * C::SLOTS + lx;
#pragma unroll
                        for (int ch = 0; ch < 16; ++ch) put(d + ch * C::ASTR, 64 * C::ASTR, sv[(im * C::GCH + c) * 16 + ch]);
                    }
                }
        }
    };

#if SC_WGRAD_S16
    // Round 6: the same products on v_mfma_f32_16x16x32_bf16 -- the chip sustains 0.84 of the nominal bf16 rate on that shape against 0.73 on
    // 32x32x16 under random operands (profiles/r04_mfma_sustained.txt).  A K = 32 step covers TWO adjacent 16-slot groups: lane quarter q
    // owns slots 8 q .. 8 q + 7 of the pair, i.e. half q & 1 of group 2 G + (q >> 1): the gy fragment is still one ds_read_b128 per piece at
    // 32 G + 8 q, the patch window the same five dwords at boff(group, half).  The wave's 32 x 32 (co, ci) tile is 2 x 2 tiles of 16 x 16.
    // An odd group count (14 x 14, 7 x 7) ends with a half-empty pair: quarters 2, 3 multiply zeros.
    wg_f32x4 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u >> 1][u & 1] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int q4 = lane >> 4, l16 = lane & 15;
    const __bf16* Ab = As + ((wt >> 1) * 32 + l16) * C::ASTR + 8 * q4;                       // + 16 sa rows
    const unsigned* Bb = reinterpret_cast<const unsigned*>(Xs + ((wt & 1) * 32 + l16) * C::BSTR + ky * Wp);      // + 16 sb rows
#else
    wg_f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const __bf16* Ab = As + ((wt >> 1) * 32 + (lane & 31)) * C::ASTR + 8 * half;
    const unsigned* Bb = reinterpret_cast<const unsigned*>(Xs + ((wt & 1) * 32 + (lane & 31)) * C::BSTR + ky * Wp);
#endif

    WG_STAMP(0)
    if (k_lo < k_hi) load(k_lo);
    __syncthreads();                                                          // zero fill done
    WG_STAMP(1)
    for (int kstep = k_lo; kstep < k_hi; ++kstep) {
#ifdef SC_WGRAD_PROFILE
        const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
#endif
        store();
        __syncthreads();
#ifdef SC_WGRAD_PROFILE
        WG_ACCUM(4, ts0)                                                      // staging: wait for the loads, split, LDS writes, barrier
        const unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
#endif
        if (kstep + 1 < k_hi) load(kstep + 1);
#if SC_WGRAD_S16
#pragma unroll
        for (int G2 = 0; G2 < (C::G + 1) / 2; ++G2) {
            constexpr int NG = C::G;
            const bool tail = 2 * G2 + 1 >= NG;                                // (compile-time per unrolled iteration) the pair's second group does not exist
            const bool dead = tail && q4 >= 2;
            const int g1 = tail ? 2 * G2 : 2 * G2 + 1;
            const int bo = (q4 == 0 ? C::boff(2 * G2, 0) : q4 == 1 ? C::boff(2 * G2, 1) : q4 == 2 ? C::boff(g1, 0) : C::boff(g1, 1)) >> 1;
            float4 a[3][2];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 av = *reinterpret_cast<const float4*>(Ab + pc * 64 * C::ASTR + u * 16 * C::ASTR + (dead ? 0 : 32 * G2));
                    a[pc][u] = dead ? make_float4(0.f, 0.f, 0.f, 0.f) : av;
                }
            // one ci half at a time: 15 patch dwords live beside the 24 of the two gy fragments (both halves at once spilled 177 registers:
            // three waves per SIMD leave 168)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                unsigned d[3][5];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int q = 0; q < 5; ++q) d[pc][q] = Bb[pc * 32 * C::BSTR + sb * 8 * C::BSTR + bo + q];
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
                    const unsigned* q = d[PB[term]];
                    const uint4 b0 = {q[0], q[1], q[2], q[3]}, b2 = {q[1], q[2], q[3], q[4]};
                    const uint4 b1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                      __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(q[4], q[3], 16)};
#pragma unroll
                    for (int sa = 0; sa < 2; ++sa) {
                        const wg_bf16x8 av = __builtin_bit_cast(wg_bf16x8, a[PA[term]][sa]);
                        acc[0][sa][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(wg_bf16x8, b0), acc[0][sa][sb], 0, 0, 0);
                        acc[1][sa][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(wg_bf16x8, b1), acc[1][sa][sb], 0, 0, 0);
                        acc[2][sa][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(wg_bf16x8, b2), acc[2][sa][sb], 0, 0, 0);
                    }
                }
                // the operands stay live until the last MFMA that reads them has issued: the K = 32 shape reads its operands after the first
                // pass and hipcc may otherwise place a destination on a dying operand
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    asm volatile("" : "+v"(acc[0][0][sb]) : "v"(d[pc][0]), "v"(d[pc][1]), "v"(d[pc][2]), "v"(d[pc][3]), "v"(d[pc][4]));
            }
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                asm volatile("" : "+v"(acc[0][0][0]) : "v"(__builtin_bit_cast(wg_f32x4, a[pc][0])), "v"(__builtin_bit_cast(wg_f32x4, a[pc][1])));
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int g = 0; g < C::G; ++g) {
            float4 a[3];
            unsigned d[3][5];
            const int bo = (half ? C::boff(g, 1) : C::boff(g, 0)) >> 1;       // dword offset: every term of boff is even
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a[pc] = *reinterpret_cast<const float4*>(Ab + pc * 64 * C::ASTR + 16 * g);
#pragma unroll
                for (int q = 0; q < 5; ++q) d[pc][q] = Bb[pc * 32 * C::BSTR + bo + q];
            }
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
                const unsigned* q = d[PB[term]];
                const wg_bf16x8 av = __builtin_bit_cast(wg_bf16x8, a[PA[term]]);
                uint4 b0 = {q[0], q[1], q[2], q[3]}, b2 = {q[1], q[2], q[3], q[4]};
                uint4 b1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                            __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(q[4], q[3], 16)};
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b0), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b1), acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b2), acc[2], 0, 0, 0);
            }
        }
#endif
        __syncthreads();                                                      // every wave is done reading this K-step's tiles
#ifdef SC_WGRAD_PROFILE
        WG_ACCUM(5, ts1)                                                      // load issue + MFMA loop + barrier
#endif
    }
    WG_STAMP(2)
    float* dst = partial + (size_t)blockIdx.x * (64 * 64 * 9);
#if SC_WGRAD_S16
    // into the partial layout of the 32 x 32 form (what conv3x3_wgrad_reduce_kernel reads): element (co, ci) of the wave's tile sits at
    // register (co & 3) + 4 (co >> 3) of lane ci + 32 ((co >> 2) & 1); here co = 16 sa + 4 q4 + r, ci = 16 sb + l16
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sa = 0; sa < 2; ++sa)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[((wt * 9 + ky * 3 + t) * 16 + r + 4 * (2 * sa + (q4 >> 1))) * 64 + 16 * sb + l16 + 32 * (q4 & 1)] = acc[t][sa][sb][r];
#else
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((wt * 9 + ky * 3 + t) * 16 + r) * 64 + lane] = acc[t][r];
#endif
#ifdef SC_WGRAD_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(3)
    if (wgrad_prof_buf && tid == 0) wgrad_prof_buf[(size_t)blockIdx.x * 16 + 6] = (unsigned long long)(k_hi - k_lo);
#endif
}

using Ws56 = WsCfg<56, 2, 1>;
using Ws28 = WsCfg<28, 4, 1>;
using Ws14 = WsCfg<14, 7, 1>;
using Ws7 = WsCfg<7, 7, 2>;

static int wg_cus() { return grid_cus(); }
static int wg_splits(int cin, int cout) {
    const int nblk = (cin / 64) * (cout / 64), s = wg_cus() / nblk;
    return s > 0 ? s : 1;
}

template <class C>
static int launch_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, hipStream_t st) {
    if (cin % 64 || cout % 64 || batch <= 0) return (int)hipErrorInvalidValue;
    const int nblk = (cin / 64) * (cout / 64), S = wg_splits(cin, cout);
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_FLOATS * 4);
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<C>), dim3(nblk * S), dim3(512), C::LDS_FLOATS * 4, st, gy, x, workspace, batch, cin, cout, S);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(nblk * 576), dim3(256), 0, st, workspace, dw, cin, cout, S);
    return (int)hipGetLastError();
}

template <class C>
static int launch_wgrad_split(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, hipStream_t st) {
    if (cin % 64 || cout % 64 || batch <= 0) return (int)hipErrorInvalidValue;
    const int nblk = (cin / 64) * (cout / 64), S = wg_splits(cin, cout);
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_split_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<C>), dim3(nblk * S), dim3(768), C::LDS_BYTES, st, gy, x, workspace, batch, cin, cout, S);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(nblk * 576), dim3(256), 0, st, workspace, dw, cin, cout, S);
    return (int)hipGetLastError();
}

}  // namespace sc

#ifdef SC_WGRAD_PROFILE
extern "C" int sc_wgrad_debug_set_prof(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sc::wgrad_prof_buf), &buf, sizeof(buf));
}
#endif
// sc_conv3x3_wgrad with fp32-accurate products on the bf16 matrix pipe (exact three-way bf16 split of both operands, six piece
// products, fp32 accumulate): same arguments, workspace and summation order over the pixel ranges.
extern "C" int sc_conv3x3_wgrad_split(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream) {
    switch (hw) {
        case 56: return sc::launch_wgrad_split<sc::Ws56>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 28: return sc::launch_wgrad_split<sc::Ws28>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 14: return sc::launch_wgrad_split<sc::Ws14>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 7: return sc::launch_wgrad_split<sc::Ws7>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        default: return -1;
    }
}

extern "C" long long sc_conv3x3_wgrad_workspace_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 64 || cout % 64) return -1;
    return (long long)(cin / 64) * (cout / 64) * sc::wg_splits(cin, cout) * 36864;
}

// dL/dw [cout][cin][3][3] of F.conv2d(x, w, None, 2, 1): gy [batch][cout][hw/2][hw/2], x [batch][cin][hw][hw], hw = 56 / 28 / 14 (side of the
// INPUT map); workspace: sc_conv3x3_wgrad_workspace_floats(cin, cout) floats.  Same kernel, stride-2 patch addressing.
extern "C" int sc_conv3x3s2_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream) {
    switch (hw) {
        case 56: return sc::launch_wgrad<sc::Wg28S2>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 28: return sc::launch_wgrad<sc::Wg14S2>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 14: return sc::launch_wgrad<sc::Wg7S2>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        default: return -1;
    }
}

extern "C" int sc_conv3x3_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream) {
    switch (hw) {
        case 56: return sc::launch_wgrad<sc::Wg56>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 28: return sc::launch_wgrad<sc::Wg28>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 14: return sc::launch_wgrad<sc::Wg14>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        case 7: return sc::launch_wgrad<sc::Wg7>(gy, x, dw, workspace, batch, cin, cout, (hipStream_t)stream);
        default: return -1;
    }
}
