// 3x3 / stride 1 / pad 1 convolutions of the ResNet-18/34 trunks (model/graph.py:50-54 encoder,
// model/view_estimator.py:40-42 estimator; torchvision BasicBlock conv1/conv2), fp32 in / fp32 accumulate, NCHW.
//
// Direct convolution on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), no im2col, no Winograd:
//     out[co][p] = sum_{ci, tap} Wt[co][ci, tap] * x[ci][p + shift(tap)]          p = flat pixel (image, y, x)
// as the GEMM  D[co][p] = A[co][k] B[k][p], k = (tap, ci) -- computed "transposed" so that the 32 lanes of a C/D column group are 32
// CONSECUTIVE PIXELS: the stores are 128-byte contiguous runs of the NCHW output and the B operand is a plain shifted read of the
// input.  One workgroup (8 waves, two per SIMD) per CU; a tile is CT output channels x PT consecutive flat pixels, every wave owns
// WM x WN 32x32 accumulators of it.  Per K-step (8 input channels x 9 taps):
//   * the input is staged ONCE as a zero-padded patch in "padded flat" coordinates  q = b (W+2)^2 + (y+1)(W+2) + (x+1): in that space
//     tap (ky, kx) is the constant offset ky (W+2) + kx, so the nine B operands of a pixel are nine reads of the same patch with
//     different IMMEDIATE offsets (each input element is fetched once per tile and used 9 CT times from LDS); lanes find their
//     pixel through one per-lane patch offset computed once per tile;
//   * the patch is channel-interleaved, [half][position][4 channels], and the weight image is [tap][half][channel out][4 channels in]:
//     ONE ds_read_b128 per operand tile feeds FOUR MFMAs (the MFMA's two k of a lane half are k = (tap, 4 half + s), s = 0..3 over the
//     four MFMAs).  This is what the kernel is built around: on this chip a vector or LDS instruction costs ~4.5 issue cycles that the
//     matrix pipe cannot hide, so the loop carries WM + WN reads per 4 WM WN MFMAs and nothing else;
//   * the weight image arrives by LDS-DMA as a verbatim copy of what sc_conv3x3_pack wrote (for the backward-data pass that is the
//     transposed + flipped filter: the same kernel computes dL/dx from dL/dy).
// Two LDS stages, one raw barrier per K-step.
//
// Load balance ("stream-K"): the feature maps hold 49 * 2^n pixels, so no tile size divides the work evenly over 256 CUs.  The tiles
// that fill whole rounds of the grid are computed one per workgroup; the K-steps of the remaining tiles are cut into gridDim.x equal
// contiguous spans, every workgroup writes the (at most two) partial tiles of its span to a workspace and conv3x3_fixup_kernel adds
// the partials of a tile in K order -- a fixed summation order, results do not depend on timing.
//
// Roofline: 2 * 9 * Cin FLOP per output element on the 157.3 TFLOP/s fp32 matrix pipe; HBM traffic is one read of x and one write of
// out (<= 0.1 byte per FLOP), i.e. compute bound by two orders of magnitude.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "grid_cus.hpp"
#include "shapeclipper_hip.h"

#ifndef SC_CONV_SPANS
#define SC_CONV_SPANS 1
#endif
#ifndef SC_CONV_ABLATE          // timing experiments only (wrong results): 1 no MFMAs, 2 no operand reads, 4 no patch staging, 8 no filter DMA
#define SC_CONV_ABLATE 0
#endif

namespace sc {

#ifdef SC_CONV_PROFILE           // profile build (tools/prof_conv_phases.py; tools/build_variants.sh conv3x3.hip SC_CONV_PROFILE 1): s_memrealtime
                                 // stamps (100 MHz) of thread 0 of every workgroup
__device__ unsigned long long* conv_prof_buf = nullptr;
#define CV_STAMP(ID) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (conv_prof_buf && tid == 0) conv_prof_buf[(size_t)g * 32 + (ID)] = t_; }
#else
#define CV_STAMP(ID)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cv_bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_cv_t;

__device__ __forceinline__ void conv_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// BD2: backward-data of the 3x3 / stride-2 / pad-1 convolutions (BasicBlock.conv1 of layer2-4) as FOUR stride-1 sub-convolutions over the
// W x W gradient map, one per parity (py, px) of the 2W x 2W result:  gx[ci][2y + py][2x + px] = sum_co sum_taps w[co][ci][ky][kx]
// gy[co][y + dy][x + dx]  with the taps whose (2y + py + 1 - ky, 2x + px + 1 - kx) are even -- 1, 2, 2 and 4 of the nine, at (dy, dx) in
// {0, 1}^2: every product of the transposed convolution is computed exactly once (a zero-stuffed stride-1 convolution would do 4x the
// work).  A tile is (phase, CT channels, PT pixels of the gradient map); the tap-pair list and the filter image depend on the phase.
template <int W_, int CT_, int PT_, int WGM_, int WGN_, bool SPLIT_ = false, int S_ = 1, bool BD2_ = false>
struct ConvCfg {
    static constexpr int W = W_, CT = CT_, PT = PT_, WGM = WGM_, WGN = WGN_, CB = 8;
    static constexpr bool SPLIT = SPLIT_;            // operands as three bf16 pieces on the bf16 matrix pipe (see conv3x3 SPLIT below)
    static constexpr bool BD2 = BD2_;
    static constexpr int PSZ = 3 * 2 * CT * 4;       // floats of one tap-pair image (SPLIT): [piece][half][CT][8 bf16]
    // PAIRED (the split stride-1 instance): no zero tap.  Taps 0..7 of a K-step are four tap pairs; tap 8 of an EVEN step shares its
    // MFMAs with tap 8 of the following odd step (lane half 0: the even step's channels, half 1: the odd step's), computed in the odd
    // step from the two patch stages: 9 tap-pair products per two K-steps instead of 10.  K-steps come in pairs: an odd channel-block
    // count is rounded up with a step of zero weights over zero inputs.
    static constexpr bool PAIRED = SPLIT && !BD2_;
    static_assert(!BD2 || (SPLIT && S_ == 1), "the stride-2 backward-data instance exists in the split arithmetic only");
    static constexpr int Wp = W + 2, HW = W * W, Sp = Wp * Wp;             // geometry of the INPUT map (W x W, one pad ring)
    static constexpr int S = S_, WO = W / S, HWO = WO * WO;                  // stride and output map (stride 2: the three conv1 of layer2-4)
    static constexpr int NT = 64 * WGM * WGN;
    static constexpr int WM = CT / (32 * WGM), WN = PT / (32 * WGN);        // 32x32 MFMA tiles per wave
    // floats (4-byte units) of one weight stage: fp32 [tap][half][CT][4]; SPLIT up to five tap-pair images [piece (3)][half][CT][8 bf16]
    static constexpr int WIMG = BD2 ? 2 * PSZ : (SPLIT ? 5 * PSZ : 9 * CB * CT);
    // longest padded-flat span of PT consecutive (output) pixels plus the halo.  Stride 1: 2 pad columns per row crossed, 2 pad rows
    // per image crossed.  Stride S: S positions per pixel, S (Wp - WO) extra per row crossed, Sp - S (WO - 1)(Wp + 1) per image crossed.
    static constexpr int LMAX = S == 1 ? PT + 2 * (PT / W + 2) + 2 * Wp * (PT / HW + 1) + 2 * (Wp + 1)
                                       : S * PT + (PT / WO + 2) * S * (Wp - WO) + (PT / HWO + 1) * (Sp - S * (WO - 1) * (Wp + 1)) + 2 * (Wp + 1);
    static constexpr int LX = LMAX;                                          // positions per channel half
    static constexpr int NXE = (LMAX + NT - 1) / NT;                         // patch positions per thread
    static constexpr int STAGE = WIMG + (SPLIT ? 3 : 2) * LX * 4;            // floats: patch [half][LX][4 fp32] or [piece][LX][8 bf16]
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static constexpr int TILE = CT * PT;                                     // floats of one (partial) output tile
    static constexpr int WGS_PER_CU = NT == 512 ? 1 : 2;                     // 8 waves per CU either way
    static_assert(WM >= 1 && WN >= 1 && (NT == 512 || NT == 256) && LDS_BYTES * WGS_PER_CU <= 160 * 1024, "tile shape");
};

template <class C>
__device__ __forceinline__ int padded_q(int p) {
    const int b = p / C::HWO, r = p - b * C::HWO, y = r / C::WO, x = r - y * C::WO;          // output pixel -> centre tap in the input
    return b * C::Sp + (C::S * y + 1) * C::Wp + (C::S * x + 1);
}

// ---- BD2 tables: phase = 2 py + px; tap pairs per phase 1, 1, 1, 2; lane half h of pair t multiplies patch position (ky', kx') (offset
// ky' Wp + kx' from the patch origin, centre = (1, 1)) with forward-filter tap (ky, kx) -- none for the second half of phase 0.
__host__ __device__ constexpr int bd2_pairs(int phase) { return phase == 3 ? 2 : 1; }
__host__ __device__ constexpr int bd2_pair_base(int phase) { return phase; }                      // pairs before this phase: 0, 1, 2, 3
// forward-filter tap ky * 3 + kx of (phase, pair, half), -1 = none
__host__ __device__ constexpr int bd2_filter_tap(int phase, int pair, int h) {
    return phase == 0 ? (h == 0 ? 4 : -1)
         : phase == 1 ? (h == 0 ? 3 : 5)
         : phase == 2 ? (h == 0 ? 1 : 7)
         : pair == 0 ? (h == 0 ? 0 : 2) : (h == 0 ? 6 : 8);
}
// patch position ky' * 3 + kx' the same lane half reads (the second half of phase 0 re-reads the centre: finite values x zero weights)
__host__ __device__ constexpr int bd2_patch_tap(int phase, int pair, int h) {
    return phase == 0 ? 4
         : phase == 1 ? (h == 0 ? 5 : 4)
         : phase == 2 ? (h == 0 ? 7 : 4)
         : pair == 0 ? (h == 0 ? 8 : 7) : (h == 0 ? 5 : 4);
}

// K-steps of a tile: channel blocks of CB, rounded up to a pair for the PAIRED instances
template <class C>
__host__ __device__ constexpr int conv_nk(int cin) { return C::PAIRED ? ((cin / C::CB + 1) & ~1) : cin / C::CB; }

// The tiles of the last, incomplete round of the grid are cut into spans of K-steps ("units"), one span per workgroup: the units [U[g],
// U[g + 1]) of the tile-major unit sequence, from a table the host builds once per (tiles, K-steps, grid) (conv_spans below; a null
// table = equal spans of per_wg units).  A span is at most one tile's worth of units long, i.e. it touches one or two tiles.
struct ConvSplit {
    int rounds;          // whole rounds: workgroup g computes tiles r * G + g, r < rounds, completely
    int tail_tiles;      // tiles rounds * G .. rounds * G + tail_tiles - 1 are shared
    int per_wg;          // units (K-steps) of the shared tiles per workgroup
};
__host__ __device__ inline ConvSplit conv_split(int tiles, int nk, int G) {
    ConvSplit s;
    s.rounds = tiles / G;
    s.tail_tiles = tiles - s.rounds * G;
    s.per_wg = (int)(((long long)s.tail_tiles * nk + G - 1) / G);
    return s;
}

// first unit of workgroup g's span (g == G: the end of the sequence)
__device__ __forceinline__ long long conv_span_at(const int* __restrict__ spans, const ConvSplit& sp, int g, int nk) {
    return spans ? (long long)spans[g] : min((long long)g * sp.per_wg, (long long)sp.tail_tiles * nk);
}
// the workgroup whose span holds unit u
__device__ __forceinline__ int conv_span_owner(const int* __restrict__ spans, const ConvSplit& sp, long long u, int G) {
    if (!spans) return (int)(u / sp.per_wg);
    int lo = 0, hi = G - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (spans[mid] <= u) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// acc[r] of a 32x32 C/D tile is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
// tile -> (phase, pixel tile, channel tile); address of output pixel p (of that phase) in channel 0 of its image, and the channel stride
template <class C>
__device__ __forceinline__ void conv_tile_decode(int tile, int nct, int& phase, int& pt, int& ct) {
    phase = C::BD2 ? tile & 3 : 0;
    const int t = C::BD2 ? tile >> 2 : tile;
    pt = t / nct, ct = t - pt * nct;
}
template <class C>
__device__ __forceinline__ float* conv_out_pixel(float* __restrict__ out, int p, int phase, int cout, int& cstride) {
    const int b = p / C::HWO, rem = p - b * C::HWO;
    if constexpr (C::BD2) {                         // the result lives on the 2W x 2W map: pixel (2y + py, 2x + px)
        const int y = rem / C::W, x = rem - y * C::W;
        cstride = 4 * C::HW;
        return out + (size_t)b * cout * cstride + (2 * y + (phase >> 1)) * (2 * C::W) + 2 * x + (phase & 1);
    } else {
        cstride = C::HWO;
        return out + (size_t)b * cout * C::HWO + rem;
    }
}

template <class C>
__device__ __forceinline__ void conv_store_tile(float* __restrict__ out, const f32x16 (&acc)[C::WM][C::WN], int tile, int nct, int npix,
                                                int cout, int wm, int wn, int lane, const float* __restrict__ addend) {
    int phase, pt, ct;
    conv_tile_decode<C>(tile, nct, phase, pt, ct);
#pragma unroll
    for (int j = 0; j < C::WN; ++j) {
        const int p = pt * C::PT + (wn * C::WN + j) * 32 + (lane & 31);
        if (p >= npix) continue;
        int cs;
        float* ob = conv_out_pixel<C>(out, p, phase, cout, cs);
#pragma unroll
        for (int i = 0; i < C::WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * C::CT + (wm * C::WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                // addend (round 5): a tensor of the output's shape added in the epilogue -- the residual-branch gradient joining the
                // backward-data result of a BasicBlock's conv1 (was a separate read-read-write launch per block and step)
                if (co < cout) ob[(size_t)co * cs] = addend ? acc[i][j][r] + addend[(ob - out) + (size_t)co * cs] : acc[i][j][r];
            }
    }
}

// six exact piece products of one tap pair (SPLIT): a_p b_q with p + q <= 2, small terms first
template <class C>
__device__ __forceinline__ void conv_split_terms(const float4 (&a)[C::WM][3], const float4 (&b)[C::WN][3], f32x16 (&acc)[C::WM][C::WN]) {
#pragma unroll
    for (int term = 0; term < 6; ++term) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int i = 0; i < C::WM; ++i)
#pragma unroll
            for (int j = 0; j < C::WN; ++j)
#if SC_CONV_ABLATE & 1
                acc[i][j][term] += a[i][PA[term]].x * b[j][PB[term]].x;
#else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cv_bf16x8, a[i][PA[term]]),
                                                                    __builtin_bit_cast(cv_bf16x8, b[j][PB[term]]), acc[i][j], 0, 0, 0);
#endif
    }
}
// the tap pairs of one K-step of a BD2 tile of phase PHASE
template <class C, int PHASE>
__device__ __forceinline__ void conv_bd2_step(const float* Ws, const float* Xs, const int (&aoff)[C::WM], const int (&boff)[C::WN], int half,
                                              f32x16 (&acc)[C::WM][C::WN]) {
    constexpr int Wp = C::Wp, LX = C::LX, CT = C::CT;
#pragma unroll
    for (int tp = 0; tp < bd2_pairs(PHASE); ++tp) {
        constexpr int dummy = 0;
        (void)dummy;
        const int t0 = bd2_patch_tap(PHASE, tp, 0), t1 = bd2_patch_tap(PHASE, tp, 1);
        const int o0 = (t0 / 3) * Wp + t0 % 3, o1 = (t1 / 3) * Wp + t1 % 3;
        float4 a[C::WM][3], b[C::WN][3];
#pragma unroll
        for (int i = 0; i < C::WM; ++i)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) a[i][pc] = *reinterpret_cast<const float4*>(Ws + aoff[i] + (tp * 3 + pc) * 2 * CT * 4);
#pragma unroll
        for (int j = 0; j < C::WN; ++j)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                b[j][pc] = *reinterpret_cast<const float4*>(Xs + boff[j] + (pc * LX + o0) * 4 + half * (o1 - o0) * 4);
        conv_split_terms<C>(a, b, acc);
    }
}

template <class C>
__global__ __launch_bounds__(C::NT, C::WGS_PER_CU) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wpack,
                                                             float* __restrict__ out, float* __restrict__ partial, int batch, int cin,
                                                             int cout, const int* __restrict__ spans, const float* __restrict__ addend) {
    constexpr int W = C::W, Wp = C::Wp, HW = C::HW, Sp = C::Sp, CT = C::CT, PT = C::PT, CB = C::CB, NT = C::NT;
    constexpr int WM = C::WM, WN = C::WN, LX = C::LX, NXE = C::NXE;
    extern __shared__ float4 conv_smem[];
    float* S = reinterpret_cast<float*>(conv_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int npix = batch * C::HWO, nct = (cout + CT - 1) / CT, tiles = ((npix + PT - 1) / PT) * nct * (C::BD2 ? 4 : 1);
    const int nk = conv_nk<C>(cin), nk_real = cin / CB, G = gridDim.x, g = blockIdx.x;
    const ConvSplit sp = conv_split(tiles, nk, G);
    // workgroup b runs on XCD b % 8: give every XCD a contiguous range of each round's tiles (shared patches / weights stay in its L2)
    const int xq = G >> 3, xr = G & 7, xcd = g & 7;
    const int gperm = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (g >> 3);
    const long long unit_lo = conv_span_at(spans, sp, g, nk), unit_hi = conv_span_at(spans, sp, g + 1, nk);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_cv_t)S);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    CV_STAMP(0)
    for (int it = 0; it < sp.rounds + 2; ++it) {
        // the second workgroup of a CU starts with its shared tiles: the two never store / restart at the same time
        const int item = (C::WGS_PER_CU == 2 && g >= (G >> 1)) ? (it + sp.rounds) % (sp.rounds + 2) : it;
        int tile, kb0, kb1;
        if (item < sp.rounds) {
            tile = item * G + gperm, kb0 = 0, kb1 = nk;
        } else if (item == sp.rounds) {                    // first shared tile of this workgroup's span
            if (unit_lo >= unit_hi) continue;
            const int t = (int)(unit_lo / nk);
            tile = sp.rounds * G + t, kb0 = (int)(unit_lo - (long long)t * nk), kb1 = (int)min((long long)nk, unit_hi - (long long)t * nk);
        } else {                                            // second one (a span is at most nk units long)
            const int t = (int)(unit_lo / nk) + 1;
            if (unit_lo >= unit_hi || (long long)t * nk >= unit_hi) continue;
            tile = sp.rounds * G + t, kb0 = 0, kb1 = (int)(unit_hi - (long long)t * nk);
        }
        int phase, pt, ct;
        conv_tile_decode<C>(tile, nct, phase, pt, ct);
        const int p0 = pt * PT;
        const int q0 = padded_q<C>(p0), q_lo = q0 - Wp - 1;

        // patch staging plan: position e of the patch is padded-flat position q_lo + e
        int xoff[NXE];
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int e = tid + i * NT, q = q_lo + e;
            const int b = q / Sp, r = q - b * Sp, yp = r / Wp, xp = r - yp * Wp;
            const bool valid = e < C::LMAX && b < batch && yp >= 1 && yp <= W && xp >= 1 && xp <= W;
            xoff[i] = valid ? (b * cin * HW + (yp - 1) * W + (xp - 1)) : -1;
        }
        // operand offsets (floats): lane half h takes input channels 4 h .. 4 h + 3 of the K-step
        int boff[WN], aoff[WM];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int p = min(p0 + (wn * WN + j) * 32 + (lane & 31), npix - 1);
            boff[j] = (half * LX + padded_q<C>(p) - q0) * 4;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) aoff[i] = (half * CT + (wm * WM + i) * 32 + (lane & 31)) * 4;
        if constexpr (C::SPLIT) {       // the patch holds one 16-byte chunk (8 channels of one piece) per position: no half term
#pragma unroll
            for (int j = 0; j < WN; ++j) boff[j] -= half * LX * 4;
        }

        // filter image of one K-step: [ct][kb][tap][half][CT][4]; BD2: [phase][ct][kb][pair][piece][half][CT][8 bf16], 1 or 2 pairs per phase
        // PAIRED: [ct][K-step pair][9 images]: images 0..3 the even step's tap pairs, 4..7 the odd step's, 8 the shared tap-8 pair -- an even
        // step stages 4 images, an odd step 5 (its own four and the shared one)
        const int wimg = C::BD2 ? bd2_pairs(phase) * C::PSZ : C::WIMG;
        const float* wsrc = wpack + (C::BD2 ? (size_t)bd2_pair_base(phase) * nct * nk * C::PSZ : 0) +
                            (C::PAIRED ? (size_t)ct * (nk >> 1) * 9 * C::PSZ : (size_t)ct * nk * wimg);
        auto issue_w = [&](int kb, int stage) {
            const int CHUNKS = C::PAIRED ? ((kb & 1) ? 5 : 4) * C::PSZ / 4 : wimg / 4;
            const char* src = reinterpret_cast<const char*>(wsrc + (C::PAIRED ? ((size_t)(kb >> 1) * 9 + (kb & 1) * 4) * C::PSZ : (size_t)kb * wimg));
            const unsigned dst = lds0 + (unsigned)stage * (C::STAGE * 4);
#pragma unroll
            for (int c = 0; c < (C::WIMG / 4 + NT - 1) / NT; ++c) {
                const int chunk0 = c * NT + wave_u * 64;                     // wave-uniform
                if (chunk0 + lane < CHUNKS) conv_glds16(src + (size_t)(chunk0 + lane) * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + chunk0 * 16)));
            }
        };
        float xv[NXE][CB];
        unsigned xkeep = 0xffffffffu;
        // SPLIT: x through a buffer resource -- a position outside the images carries an out-of-range offset (the load returns 0, no
        // branch), the channel plane is the scalar offset.  The zero step that completes a pair (kb >= nk_real, odd cin / 8 only) reads
        // whatever follows its block (0 past the end of x) and store_x masks the values: zero weights would not silence an Inf.
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x), 0, (int)min((long long)batch * cin * HW * 4, 0x7fffffffLL), 0x00020000);
        int xoffb[NXE];
#pragma unroll
        for (int i = 0; i < NXE; ++i) xoffb[i] = xoff[i] >= 0 ? xoff[i] * 4 : (int)0x80000000;
        auto load_x = [&](int kb) {
            if constexpr (C::SPLIT) {
                xkeep = kb < nk_real ? 0xffffffffu : 0u;
#pragma unroll
                for (int i = 0; i < NXE; ++i)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        xv[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoffb[i], (kb * CB + c) * HW * 4, 0));
            } else {
                const float* xb = x + (size_t)kb * CB * HW;
#pragma unroll
                for (int i = 0; i < NXE; ++i)
#pragma unroll
                    for (int c = 0; c < CB; ++c) xv[i][c] = xoff[i] >= 0 ? xb[xoff[i] + c * HW] : 0.f;
            }
        };
        auto store_x = [&](int stage) {
            float4* Xs = reinterpret_cast<float4*>(S + stage * C::STAGE + C::WIMG);
#pragma unroll
            for (int i = 0; i < NXE; ++i) {
                const int e = tid + i * NT;
                if (e < C::LMAX) {
                    if constexpr (C::SPLIT) {       // x = p0 + p1 + p2 exactly, each piece a bf16 (round to nearest even, residuals exact)
                        cv_bf16x8 p0, p1, p2;
#pragma unroll
                        for (int c = 0; c < CB; ++c) {
                            const float v = C::PAIRED ? __uint_as_float(__float_as_uint(xv[i][c]) & xkeep) : xv[i][c];
                            const __bf16 h0 = (__bf16)v;
                            const float r1 = v - (float)h0;
                            const __bf16 h1 = (__bf16)r1;
                            const float r2 = r1 - (float)h1;
                            p0[c] = h0, p1[c] = h1, p2[c] = (__bf16)r2;
                        }
                        Xs[e] = __builtin_bit_cast(float4, p0);
                        Xs[LX + e] = __builtin_bit_cast(float4, p1);
                        Xs[2 * LX + e] = __builtin_bit_cast(float4, p2);
                    } else {
                        Xs[e] = make_float4(xv[i][0], xv[i][1], xv[i][2], xv[i][3]);
                        Xs[LX + e] = make_float4(xv[i][4], xv[i][5], xv[i][6], xv[i][7]);
                    }
                }
            }
        };

        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        [[maybe_unused]] const int sid = 1 + 8 * (item < sp.rounds ? 0 : item - sp.rounds + 1);      // profile stamps of this item (whole tile | first | second shared tile)
        CV_STAMP(sid)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done with the previous item's stages
        // A span that starts on an odd step needs the even partner's patch in the other stage (the shared tap-8 pair reads it).  Staged
        // unconditionally (an even start re-stages its own block there: nobody reads it) and before the filter DMA: straight-line code --
        // with a conditional staging hipcc's wait-count pass put `s_waitcnt vmcnt(0)` between the loads, one round trip each.
        if constexpr (C::PAIRED) {
            load_x(kb0 - (kb0 & 1));
            store_x(1);
        }
        issue_w(kb0, 0);
        load_x(kb0);
        store_x(0);
        CV_STAMP(sid + 1)
        for (int kb = kb0; kb < kb1; ++kb) {
            // (builtins, not inline asm: hipcc's wait-count pass must SEE that nothing is in flight here.  Behind an asm barrier it assumed
            //  the previous step's patch loads could still be pending and put `s_waitcnt vmcnt(0)` in front of the first register they
            //  had targeted -- which then waited for the filter DMA issued two instructions earlier: 2.7 % of the kernel.)
            __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0) lgkmcnt(0): this step's stage complete ...
            __builtin_amdgcn_s_barrier();                      // ... for every wave, the other stage free
            const int st = (kb - kb0) & 1;
#ifdef SC_CONV_PROFILE
            if (kb == kb0 + 1) CV_STAMP(sid + 5)
#endif
#if SC_CONV_ABLATE & 12
            if (kb + 1 < kb1) {
                if (!(SC_CONV_ABLATE & 8)) issue_w(kb + 1, st ^ 1);
                if (!(SC_CONV_ABLATE & 4)) load_x(kb + 1);
            }
#else
            if (kb + 1 < kb1) { issue_w(kb + 1, st ^ 1); load_x(kb + 1); }
#endif
            const float* Ws = S + st * C::STAGE;
            const float* Xs = Ws + C::WIMG;
            if constexpr (C::BD2) {
                switch (phase) {
                    case 0: conv_bd2_step<C, 0>(Ws, Xs, aoff, boff, half, acc); break;
                    case 1: conv_bd2_step<C, 1>(Ws, Xs, aoff, boff, half, acc); break;
                    case 2: conv_bd2_step<C, 2>(Ws, Xs, aoff, boff, half, acc); break;
                    default: conv_bd2_step<C, 3>(Ws, Xs, aoff, boff, half, acc); break;
                }
            } else if constexpr (C::SPLIT) {
                // k block of an MFMA (32x32x16): lane half h holds the 8 channels of tap 2 tp + h.  Per tap pair: WM + WN operand tiles x 3
                // pieces, six products a_p b_q with p + q <= 2 (what is dropped is < 2^-23 of |a||b|)
                float4 a[WM][3], b[WN][3];
                if (kb & 1) {
                    // the shared tap-8 pair of K-steps kb - 1 (lane half 0: its patch is still in the OTHER stage) and kb (half 1).  It
                    // goes first, and a barrier separates its reads from this step's store_x into that stage: the workgroup's waves were
                    // released together a few instructions ago, the barrier costs no skew here.
                    constexpr int o8 = 2 * Wp + 2;
                    const int xother = half ? 0 : (st ? -C::STAGE : C::STAGE);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) a[i][pc] = *reinterpret_cast<const float4*>(Ws + aoff[i] + (4 * 3 + pc) * 2 * CT * 4);
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            b[j][pc] = *reinterpret_cast<const float4*>(Xs + xother + boff[j] + (pc * LX + o8) * 4);
                    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0) (vmcnt, expcnt: no wait): the reads of the other stage are done ...
                    __builtin_amdgcn_s_barrier();                  // ... in every wave
                    conv_split_terms<C>(a, b, acc);
                }
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int t0 = 2 * tp, t1 = 2 * tp + 1;
                    const int o0 = (t0 / 3) * Wp + t0 % 3, o1 = (t1 / 3) * Wp + t1 % 3;
#if SC_CONV_ABLATE & 2
                    if (tp == 0 && kb == kb0)
#endif
                    {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) a[i][pc] = *reinterpret_cast<const float4*>(Ws + aoff[i] + (tp * 3 + pc) * 2 * CT * 4);
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            b[j][pc] = *reinterpret_cast<const float4*>(Xs + boff[j] + (pc * LX + o0) * 4 + half * (o1 - o0) * 4);
                    }
                    conv_split_terms<C>(a, b, acc);
                }
            } else {
            float4 a[2][WM], b[2][WN];
            auto frags = [&](int tap, float4 (&a2)[WM], float4 (&b2)[WN]) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a2[i] = *reinterpret_cast<const float4*>(Ws + aoff[i] + tap * 2 * CT * 4);
#pragma unroll
                for (int j = 0; j < WN; ++j) b2[j] = *reinterpret_cast<const float4*>(Xs + boff[j] + ((tap / 3) * Wp + (tap % 3)) * 4);
            };
            frags(0, a[0], b[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) frags(tap + 1, a[(tap + 1) & 1], b[(tap + 1) & 1]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            const float av = s == 0 ? a[tap & 1][i].x : s == 1 ? a[tap & 1][i].y : s == 2 ? a[tap & 1][i].z : a[tap & 1][i].w;
                            const float bv = s == 0 ? b[tap & 1][j].x : s == 1 ? b[tap & 1][j].y : s == 2 ? b[tap & 1][j].z : b[tap & 1][j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
            }
            if (kb + 1 < kb1 && !(SC_CONV_ABLATE & 4)) store_x(st ^ 1);
        }

        CV_STAMP(sid + 2)
        if (kb0 == 0 && kb1 == nk) {
            conv_store_tile<C>(out, acc, tile, nct, npix, cout, wm, wn, lane, addend);
        } else {                                            // partial tile, in register order: [wave][i][j][r / 4][lane][4], 16-byte stores
            float4* dst = reinterpret_cast<float4*>(partial + ((size_t)g * 2 + (item - sp.rounds)) * C::TILE);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        float4* d = dst + (((wave * WM + i) * WN + j) * 4 + q) * 64 + lane;
                        *d = v;        // (plain stores: write-through `sc1` stores, meant to shorten the dirty-L2 write-back at the kernel boundary, were 3 % slower)
                    }
        }
#ifdef SC_CONV_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_STAMP(sid + 3)
        if (conv_prof_buf && tid == 0) conv_prof_buf[(size_t)g * 32 + sid + 4] = (unsigned long long)(kb1 - kb0);
#endif
    }
    CV_STAMP(25)
}

// Four workgroups per shared tile (each takes 4 of the 16 accumulator rows of every 32x32 block): add the partial tiles of the
// workgroups whose spans cover it, in K order, and store the result.
template <class C>
__global__ __launch_bounds__(C::NT) void conv3x3_fixup_kernel(const float* __restrict__ partial, float* __restrict__ out, int batch,
                                                              int cin, int cout, int G, const int* __restrict__ spans,
                                                              const float* __restrict__ addend) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int npix = batch * C::HWO, nct = (cout + C::CT - 1) / C::CT, tiles = ((npix + C::PT - 1) / C::PT) * nct * (C::BD2 ? 4 : 1);
    const int nk = conv_nk<C>(cin);
    const ConvSplit sp = conv_split(tiles, nk, G);
#ifdef SC_CONV_PROFILE
    if (conv_prof_buf && tid == 0) conv_prof_buf[(size_t)(8192 + blockIdx.x * 4 + blockIdx.y) * 2] = __builtin_amdgcn_s_memrealtime();
#endif
    const int t = blockIdx.x, r0 = 4 * blockIdx.y;
    const long long u0 = (long long)t * nk, u1 = u0 + nk;
    const int g_first = conv_span_owner(spans, sp, u0, G), g_last = conv_span_owner(spans, sp, u1 - 1, G);
    if (g_first == g_last && conv_span_at(spans, sp, g_first, nk) == u0 && conv_span_at(spans, sp, g_first + 1, nk) == u1) return;   // computed whole, already stored
    float acc[C::WM][C::WN][4];
#pragma unroll
    for (int i = 0; i < C::WM; ++i)
#pragma unroll
        for (int j = 0; j < C::WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int g = g_first; g <= g_last; ++g) {
        const int which = (int)(conv_span_at(spans, sp, g, nk) / nk) == t ? 0 : 1;  // first or second tile of that workgroup's span
        const float4* src = reinterpret_cast<const float4*>(partial + ((size_t)g * 2 + which) * C::TILE);
#pragma unroll
        for (int i = 0; i < C::WM; ++i)
#pragma unroll
            for (int j = 0; j < C::WN; ++j) {
                const float4 v = src[(((wave * C::WM + i) * C::WN + j) * 4 + (r0 >> 2)) * 64 + lane];
                acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
            }
    }
    const int tile = sp.rounds * G + t;
    int phase, pt, ct;
    conv_tile_decode<C>(tile, nct, phase, pt, ct);
#pragma unroll
    for (int j = 0; j < C::WN; ++j) {
        const int p = pt * C::PT + (wn * C::WN + j) * 32 + (lane & 31);
        if (p >= npix) continue;
        int cs;
        float* ob = conv_out_pixel<C>(out, p, phase, cout, cs);
#pragma unroll
        for (int i = 0; i < C::WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = ct * C::CT + (wm * C::WM + i) * 32 + r + 2 * r0 + 4 * (lane >> 5);      // row of acc[r0 + r]: r + 8 (r0 / 4)
                if (co < cout) ob[(size_t)co * cs] = addend ? acc[i][j][r] + addend[(ob - out) + (size_t)co * cs] : acc[i][j][r];
            }
    }
#ifdef SC_CONV_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (conv_prof_buf && tid == 0) conv_prof_buf[(size_t)(8192 + blockIdx.x * 4 + blockIdx.y) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// w [cout][cin][3][3] -> w_pack [ct][kb][tap][half][CT][4]: element (co = ct CT + cl, ci = 8 kb + 4 half + s).  transpose_flip: the filter
// of the backward-data pass, w'[ci][co][ky][kx] = w[co][ci][2 - ky][2 - kx] (`cin` / `cout` are the channel counts of THAT convolution).
// value of filter element (co, ci, tap) in the orientation of this pass
__device__ __forceinline__ float conv_w_at(const float* __restrict__ w, int cin, int cout, int co, int ci, int tap, int transpose_flip) {
    return transpose_flip ? w[((size_t)ci * cout + co) * 9 + (8 - tap)] : w[((size_t)co * cin + ci) * 9 + tap];
}
// piece pc (0, 1, 2) of the exact three-way bf16 split v = p0 + p1 + p2
__device__ __forceinline__ unsigned conv_bf16_piece(float v, int pc) {
    const __bf16 h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const __bf16 h2 = (__bf16)(r1 - (float)h1);
    const __bf16 h = pc == 0 ? h0 : pc == 1 ? h1 : h2;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
// one 4-byte word of a packed filter image.  fp32: [tap][half][CT][4]: element (co = ct CT + cl, ci = 8 kb + 4 half + s).
// split: per pair of K-steps nine images [piece][half][CT][8 bf16]: pairs (0,1) .. (6,7) of the even step, of the odd step, then the
// tap-8 pair (lane half 0: channels of the even step, half 1: of the odd step); a missing odd step (odd cin / 8) is zeros.
// (t is a 32-bit offset inside ONE filter's image: at most 512 x 512 x 10 x 3 / 2 words -- 64-bit divisions by the run-time CT / nk
// made the all-filters pack kernel compute-bound: 284 us per network)
__device__ __forceinline__ float conv_pack_word(const float* __restrict__ w, unsigned t, int CT, int nk, int cin, int cout, int flip,
                                                bool split) {
    if (!split) {
        const int s4 = (int)(t & 3);
        t >>= 2;
        const int cl = (int)(t % CT);
        t /= CT;
        const int h = (int)(t & 1);
        t >>= 1;
        const int tap = (int)(t % 9);
        t /= 9;
        const int kb = (int)(t % nk), ct = (int)(t / nk);
        const int ci = kb * 8 + 4 * h + s4, co = ct * CT + cl;
        return co < cout ? conv_w_at(w, cin, cout, co, ci, tap, flip) : 0.f;
    }
    const int wi = (int)(t & 3);
    t >>= 2;
    const int cl = (int)(t % CT);
    t /= CT;
    const int h = (int)(t & 1);
    t >>= 1;
    const int pc = (int)(t % 3);
    t /= 3;
    // nine tap-pair images per PAIR of K-steps: pairs 0..3 of the even step, pairs 0..3 of the odd step, the shared tap-8 pair
    const int slot = (int)(t % 9);
    t /= 9;
    const int nkp = (nk + 1) >> 1, kp = (int)(t % nkp), ct = (int)(t / nkp);
    const int kb = 2 * kp + (slot < 4 ? 0 : (slot < 8 ? 1 : h)), tap = slot < 8 ? 2 * (slot & 3) + h : 8;
    const int co = ct * CT + cl, ci = kb * 8 + 2 * wi;
    unsigned lo = 0, hi = 0;
    if (kb < nk && co < cout) {
        lo = conv_bf16_piece(conv_w_at(w, cin, cout, co, ci, tap, flip), pc);
        hi = conv_bf16_piece(conv_w_at(w, cin, cout, co, ci + 1, tap, flip), pc);
    }
    return __uint_as_float(lo | (hi << 16));
}

// Filter image of the stride-2 backward-data instance (BD2): [pair region 0..4][ct][kb][...] as read by conv3x3_kernel -- phases 0, 1, 2
// own one tap pair per K-step, phase 3 two.  w is the FORWARD filter [cfwd_out][cfwd_in][3][3]; this convolution's reduction channels are
// the forward's output channels (kb), its output channels the forward's input channels (ct, cl).
template <class C>
__global__ void conv3x3_pack_bd2_kernel(const float* __restrict__ w, float* __restrict__ wpack, int cfwd_in, int cfwd_out) {
    const int nk = cfwd_out / C::CB, nct = (cfwd_in + C::CT - 1) / C::CT;
    const long long region = (long long)nct * nk * C::PSZ, total = 5 * region;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / region), phase = r < 3 ? r : 3;
        long long u = i - (long long)bd2_pair_base(phase) * region;
        const int blk = bd2_pairs(phase) * C::PSZ;
        const int ctkb = (int)(u / blk);
        u -= (long long)ctkb * blk;
        const int ct = ctkb / nk, kb = ctkb - ct * nk;
        const int pair = (int)(u / C::PSZ);
        int v = (int)(u - (long long)pair * C::PSZ);
        const int wi = v & 3;
        v >>= 2;
        const int cl = v % C::CT;
        v /= C::CT;
        const int h = v & 1, pc = v >> 1;
        const int tap = bd2_filter_tap(phase, pair, h), co_o = ct * C::CT + cl, ck = kb * 8 + 2 * wi;
        unsigned lo = 0, hi = 0;
        if (tap >= 0 && co_o < cfwd_in) {
            lo = conv_bf16_piece(w[((size_t)ck * cfwd_in + co_o) * 9 + tap], pc);
            hi = conv_bf16_piece(w[((size_t)(ck + 1) * cfwd_in + co_o) * 9 + tap], pc);
        }
        wpack[i] = __uint_as_float(lo | (hi << 16));
    }
}

template <class C>
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ wpack, int cin, int cout, int transpose_flip) {
    const int nk = cin / C::CB, nct = (cout + C::CT - 1) / C::CT;
    const long long total = C::PAIRED ? (long long)nct * ((nk + 1) / 2) * 9 * C::PSZ : (long long)nct * nk * C::WIMG;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        wpack[i] = conv_pack_word(w, (unsigned)i, C::CT, nk, cin, cout, transpose_flip, C::SPLIT);
}

// Tile shapes (measured, tools/perf_conv.py): 8 waves of 64 x 64 each.  Four-wave workgroups, two per CU (NT = 256: 64 x 256 tiles) reach
// the same rates alone and in the training step; a plain one-tile-per-workgroup grid is 10-40 % slower (incomplete last round);
// storing a finished tile under the next tile's first K-step (second accumulator set) spills and is 4-6 % slower.
using Conv56 = ConvCfg<56, 64, 512, 1, 8>;       // 64 channels: the tile spans all of them
using Conv28 = ConvCfg<28, 128, 256, 2, 4>;
using Conv14 = ConvCfg<14, 128, 256, 2, 4>;
using Conv7 = ConvCfg<7, 128, 128, 2, 4>;
// stride 2 (BasicBlock.conv1 of layer2-4, forward only): 64 channels x 256 output pixels, 8 waves of 64 x 32
using Conv56S2 = ConvCfg<56, 64, 256, 1, 8, false, 2>;
using Conv28S2 = ConvCfg<28, 64, 256, 1, 8, false, 2>;
using Conv14S2 = ConvCfg<14, 64, 256, 1, 8, false, 2>;
// SPLIT (fp32-accurate products on the bf16 matrix pipe): 64 channels x 512 pixels, 8 waves of 64 x 64
using Conv56S = ConvCfg<56, 64, 512, 1, 8, true>;
using Conv28S = ConvCfg<28, 64, 512, 1, 8, true>;
using Conv14S = ConvCfg<14, 64, 512, 1, 8, true>;
using Conv7S = ConvCfg<7, 64, 256, 1, 8, true>;
// backward-data of the stride-2 layers (first argument: side of the GRADIENT map = half the side of the result), split arithmetic
using Conv28BD = ConvCfg<28, 64, 512, 1, 8, true, 1, true>;
using Conv14BD = ConvCfg<14, 64, 512, 1, 8, true, 1, true>;
using Conv7BD = ConvCfg<7, 64, 256, 1, 8, true, 1, true>;

template <class C>
static long long pack_floats(int cin, int cout) {
    if (cin % C::CB) return -1;
    const long long nct = (cout + C::CT - 1) / C::CT, nk = cin / C::CB;
    return C::PAIRED ? nct * ((nk + 1) / 2) * 9 * C::PSZ : nct * nk * C::WIMG;
}

// All filters of a network in ONE launch (they change once per optimizer step): table row e = {address of w, first float of its image
// in dst, cin, cout, CT of the map side's tile shape, transpose_flip}; rows sorted by their first float, `total` = end of the last one.
__global__ void conv3x3_pack_multi_kernel(const long long* __restrict__ table_g, int n, float* __restrict__ dst, long long total) {
    __shared__ long long table[128 * 6];                                    // n <= 128 rows (checked by the host)
    for (int i = threadIdx.x; i < n * 6; i += blockDim.x) table[i] = table_g[i];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid * 6 + 1] <= i) lo = mid; else hi = mid - 1;
        }
        const long long* row = table + lo * 6;
        const float* w = reinterpret_cast<const float*>(row[0]);
        const int cin = (int)row[2], cout = (int)row[3], CT = (int)row[4], flags = (int)row[5];     // flags: 1 transpose_flip, 2 split
        dst[i] = conv_pack_word(w, (unsigned)(i - row[1]), CT, cin / 8, cin, cout, flags & 1, (flags & 2) != 0);
    }
}

// The same for tables whose rows are all SPLIT images with 64-channel tiles (the default arithmetic): one workgroup per (filter, ct,
// pair of K-steps) unit = 64 output channels x 16 reduction channels x 9 taps.  The 9216 weights of a unit arrive in LDS through
// contiguous segments (144 floats per output channel; 576 per reduction channel in the transposed orientation) and leave as the unit's
// 13824 contiguous words [9 images][piece][half][64][4]: the word-driven kernel above gathers two weights per word from 64 different
// cache lines per wave.
__global__ __launch_bounds__(256) void conv3x3_pack_multi_split_kernel(const long long* __restrict__ table_g, int n, float* __restrict__ dst) {
    constexpr int CT = 64, UNIT = 9 * 3 * 2 * CT * 4, LST = 145;
    __shared__ float wl[CT * LST];
    __shared__ long long row_s[6];
    const int tid = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * UNIT;
    if (tid == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table_g[mid * 6 + 1] <= i0) lo = mid; else hi = mid - 1;
        }
        for (int k = 0; k < 6; ++k) row_s[k] = table_g[lo * 6 + k];
    }
    __syncthreads();
    const float* w = reinterpret_cast<const float*>(row_s[0]);
    const int cin = (int)row_s[2], cout = (int)row_s[3], flip = (int)row_s[5] & 1, nkp = (cin / 8 + 1) / 2;
    const int unit = (int)((i0 - row_s[1]) / UNIT), ct = unit / nkp, kp = unit - ct * nkp;
    for (int e = tid; e < 16 * CT * 9; e += 256) {
        int cl, sc, tap;
        float v = 0.f;
        if (flip) {      // w[ci][co][8 - tap]: per reduction channel 64 x 9 contiguous floats
            sc = e / (CT * 9);
            const int r = e - sc * (CT * 9);
            cl = r / 9; tap = 8 - (r - cl * 9);
            if (ct * CT + cl < cout && kp * 16 + sc < cin) v = w[((size_t)(kp * 16 + sc) * cout + ct * CT) * 9 + r];
        } else {         // w[co][ci][tap]: per output channel 16 x 9 contiguous floats
            cl = e / 144;
            const int r = e - cl * 144;
            sc = r / 9; tap = r - sc * 9;
            if (ct * CT + cl < cout && kp * 16 + sc < cin) v = w[((size_t)(ct * CT + cl) * cin + kp * 16) * 9 + r];
        }
        wl[cl * LST + sc * 9 + tap] = v;
    }
    __syncthreads();
    for (int v = tid; v < UNIT; v += 256) {
        const int wi = v & 3, cl = (v >> 2) & 63, h = (v >> 8) & 1, q = v >> 9, pc = q % 3, slot = q / 3;
        const int kl = slot < 4 ? 0 : (slot < 8 ? 1 : h), tap = slot < 8 ? 2 * (slot & 3) + h : 8;       // K-step of the pair, tap
        const unsigned lo = conv_bf16_piece(wl[cl * LST + (8 * kl + 2 * wi) * 9 + tap], pc);
        const unsigned hi = conv_bf16_piece(wl[cl * LST + (8 * kl + 2 * wi + 1) * 9 + tap], pc);
        dst[i0 + v] = __uint_as_float(lo | (hi << 16));
    }
}

static int conv_grid() { return grid_cus(); }

template <class C>
static long long workspace_floats() { return (long long)conv_grid() * C::WGS_PER_CU * 2 * C::TILE; }

template <class C>
static int launch_pack(const float* w, float* wpack, int cin, int cout, int tf, hipStream_t st) {
    if (cin % C::CB) return (int)hipErrorInvalidValue;
    const long long total = pack_floats<C>(cin, cout);
    hipLaunchKernelGGL((conv3x3_pack_kernel<C>), dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st,
                       w, wpack, cin, cout, tf);
    return (int)hipGetLastError();
}

// Span table of the shared tiles.  Every tile a workgroup touches costs it a fixed overhead besides the K-steps -- staging the first
// stage, a slower first K-step, the partial-tile store: ~5 us measured (tools/prof_conv_phases.py), the time of `ov2` HALF K-steps -- so
// equal spans leave the workgroups whose span crosses a tile boundary (40 % of them at 14 x 14) one overhead behind the others and the
// launch waits for them.  Here a span's cost is 2 * units + ov2 * tiles touched, and the spans are the greedy cut at the smallest cost
// bound that still covers all units with G workgroups.  Built once per (device, tail tiles, K-steps, grid, ov2) and kept on the device.
//  The device copy is made ONCE per key, stream-ordered (hipMemcpyAsync on the launch stream from a host table the cache keeps
// alive), never while that stream is being captured into a graph (hipMalloc is not capturable: a launch under capture that meets an
// unseen shape runs with equal spans -- same values up to the summation order of the shared tiles' partials, which is fixed either way),
// and sc_conv3x3_release_tables() frees every table.
struct SpanTable {
    int* dev = nullptr;
    std::vector<int> host;
};
static std::mutex span_mu;
static std::map<std::tuple<int, int, int, int, int>, SpanTable> span_cache;

static const int* conv_spans(int tail_tiles, int nk, int G, int ov2, hipStream_t st) {
    std::map<std::tuple<int, int, int, int, int>, SpanTable>& cache = span_cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(span_mu);
    const auto key = std::make_tuple(dev, tail_tiles, nk, G, ov2);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.dev;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;                                               // not cached: built by the first launch outside a capture
    }
    const int total = tail_tiles * nk;
    std::vector<int> U(G + 1), best;
    auto build = [&](int bound) {
        int pos = 0;
        for (int g = 0; g < G; ++g) {
            U[g] = pos;
            int end = pos;
            while (end < total && end - pos < nk) {
                const int e1 = end + 1, items = (e1 - 1) / nk - pos / nk + 1;
                if (end > pos && 2 * (e1 - pos) + ov2 * items > bound) break;     // (a span holds at least one unit)
                end = e1;
            }
            pos = end;
        }
        U[G] = total;
        return pos >= total;
    };
    int lo = 2 * ((total + G - 1) / G), hi = lo + 4 * ov2 + 4;       // hi: the equal cut fits (<= per_wg units, <= 2 tiles)
    while (!build(hi)) hi += 2 * nk;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (build(mid)) hi = mid; else lo = mid + 1;
    }
    (void)build(hi);
    SpanTable& t = cache[key];
    t.host = U;                                                       // the source of the asynchronous copy outlives it
    if (hipMalloc(&t.dev, (size_t)(G + 1) * sizeof(int)) != hipSuccess ||
        hipMemcpyAsync(t.dev, t.host.data(), (size_t)(G + 1) * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
        // the table is cached and handed to launches on ANY stream (the two trunks run on two streams with equal shapes): the one-time
        // upload must be complete, not merely ordered on the first launch's stream, before the pointer is published
        hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        if (t.dev) (void)hipFree(t.dev);
        t.dev = nullptr;                                              // (null: the kernels fall back to equal spans)
    }
    return t.dev;
}

template <class C>
static int launch_conv(const float* x, const float* wpack, float* out, float* workspace, int batch, int cin, int cout, hipStream_t st,
                       const float* addend = nullptr) {
    if (cin % C::CB || batch <= 0) return (int)hipErrorInvalidValue;
    const int tiles = ((batch * C::HWO + C::PT - 1) / C::PT) * ((cout + C::CT - 1) / C::CT) * (C::BD2 ? 4 : 1), nk = conv_nk<C>(cin);
    const int G = conv_grid() * C::WGS_PER_CU;
    // SPLIT instances read x through a buffer resource whose byte range is a 31-bit count: a larger input must be refused, not silently
    // read as zeros past the clamp; 2 GiB of fp32 is 2,674 images of 64 x 56 x 56 -- far outside any batch of this path
    if (C::SPLIT && (long long)batch * cin * C::HW * 4 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    (void)hipFuncSetAttribute((const void*)conv3x3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    const ConvSplit sp = conv_split(tiles, nk, G);
    // overhead of a touched tile in half K-steps: a K-step of the 512-pixel tiles takes ~5.3 us, of the 256-pixel tiles ~3.2
    const int* spans = (SC_CONV_SPANS && sp.tail_tiles > 0 && sp.per_wg <= nk) ? conv_spans(sp.tail_tiles, nk, G, C::PT >= 512 ? 2 : 3, st) : nullptr;
    hipLaunchKernelGGL((conv3x3_kernel<C>), dim3(G), dim3(C::NT), C::LDS_BYTES, st, x, wpack, out, workspace, batch, cin, cout, spans, addend);
    if (sp.tail_tiles > 0)
        hipLaunchKernelGGL((conv3x3_fixup_kernel<C>), dim3(sp.tail_tiles, 4), dim3(C::NT), 0, st, workspace, out, batch, cin, cout, G, spans, addend);
    return (int)hipGetLastError();
}

}  // namespace sc

// Frees the span tables of every device (they are rebuilt on demand).  The caller makes sure no launch that uses them is in flight.
extern "C" int sc_conv3x3_release_tables(void) {
    std::lock_guard<std::mutex> lock(sc::span_mu);
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    for (auto& kv : sc::span_cache)
        if (kv.second.dev) {
            (void)hipSetDevice(std::get<0>(kv.first));
            (void)hipFree(kv.second.dev);
        }
    sc::span_cache.clear();
    (void)hipSetDevice(dev0);
    return 0;
}

#define SC_CONV_DISPATCH(hw, CALL)                   \
    switch (hw) {                                    \
        case 56: return CALL(sc::Conv56);            \
        case 28: return CALL(sc::Conv28);            \
        case 14: return CALL(sc::Conv14);            \
        case 7: return CALL(sc::Conv7);              \
        default: return -1;                          \
    }

#define SC_CONV_DISPATCH_SPLIT(hw, CALL)              \
    switch (hw) {                                    \
        case 56: return CALL(sc::Conv56S);           \
        case 28: return CALL(sc::Conv28S);           \
        case 14: return CALL(sc::Conv14S);           \
        case 7: return CALL(sc::Conv7S);             \
        default: return -1;                          \
    }

#define SC_CONV_DISPATCH_S2(hw, CALL)                 \
    switch (hw) {                                    \
        case 56: return CALL(sc::Conv56S2);          \
        case 28: return CALL(sc::Conv28S2);          \
        case 14: return CALL(sc::Conv14S2);          \
        default: return -1;                          \
    }

// ---- backward-data of the stride-2 convolutions (hw = side of the forward INPUT map = side of the result: 56 / 28 / 14) ----
#define SC_CONV_DISPATCH_BD(hw, CALL)                 \
    switch (hw) {                                    \
        case 56: return CALL(sc::Conv28BD);          \
        case 28: return CALL(sc::Conv14BD);          \
        case 14: return CALL(sc::Conv7BD);           \
        default: return -1;                          \
    }
// floats of the filter image sc_conv3x3s2_bd_pack writes for a forward filter [cout][cin][3][3]
#ifdef SC_CONV_PROFILE
extern "C" int sc_conv_debug_set_prof(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sc::conv_prof_buf), &buf, sizeof(buf));
}
#endif
extern "C" long long sc_conv3x3s2_bd_pack_floats(int cin, int cout, int hw) {
#define CALL(C) (cout % C::CB ? -1 : 5LL * ((cin + C::CT - 1) / C::CT) * (cout / C::CB) * C::PSZ)
    SC_CONV_DISPATCH_BD(hw, CALL)
#undef CALL
}
extern "C" long long sc_conv3x3s2_bd_workspace_floats(int hw) {
#define CALL(C) sc::workspace_floats<C>()
    SC_CONV_DISPATCH_BD(hw, CALL)
#undef CALL
}
namespace sc {
template <class C>
static int launch_pack_bd2(const float* w, float* wpack, int cin, int cout, hipStream_t st) {
    if (cout % C::CB) return (int)hipErrorInvalidValue;
    const long long total = 5LL * ((cin + C::CT - 1) / C::CT) * (cout / C::CB) * C::PSZ;
    hipLaunchKernelGGL((conv3x3_pack_bd2_kernel<C>), dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st,
                       w, wpack, cin, cout);
    return (int)hipGetLastError();
}
}  // namespace sc
// w: the forward filter [cout][cin][3][3]
extern "C" int sc_conv3x3s2_bd_pack(const float* w, float* w_pack, int cin, int cout, int hw, void* stream) {
#define CALL(C) sc::launch_pack_bd2<C>(w, w_pack, cin, cout, (hipStream_t)stream)
    SC_CONV_DISPATCH_BD(hw, CALL)
#undef CALL
}
// gx [batch][cin][hw][hw] = dL/dx of F.conv2d(x, w, None, 2, 1) from gy [batch][cout][hw/2][hw/2] (cin / cout: the FORWARD convolution's)
extern "C" int sc_conv3x3s2_backward_data(const float* gy, const float* w_pack, float* gx, float* workspace, int batch, int cin, int cout,
                                          int hw, void* stream) {
#define CALL(C) sc::launch_conv<C>(gy, w_pack, gx, workspace, batch, cout, cin, (hipStream_t)stream)
    SC_CONV_DISPATCH_BD(hw, CALL)
#undef CALL
}

extern "C" long long sc_conv3x3s2_pack_floats(int cin, int cout, int hw) {
#define CALL(C) sc::pack_floats<C>(cin, cout)
    SC_CONV_DISPATCH_S2(hw, CALL)
#undef CALL
}
extern "C" long long sc_conv3x3s2_workspace_floats(int hw) {
#define CALL(C) sc::workspace_floats<C>()
    SC_CONV_DISPATCH_S2(hw, CALL)
#undef CALL
}
extern "C" int sc_conv3x3s2_forward(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout, int hw,
                                    void* stream) {
#define CALL(C) sc::launch_conv<C>(x, w_pack, out, workspace, batch, cin, cout, (hipStream_t)stream)
    SC_CONV_DISPATCH_S2(hw, CALL)
#undef CALL
}

extern "C" long long sc_conv3x3_pack_floats_split(int cin, int cout, int hw) {
#define CALL(C) sc::pack_floats<C>(cin, cout)
    SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
}
extern "C" long long sc_conv3x3_workspace_floats_split(int hw) {
#define CALL(C) sc::workspace_floats<C>()
    SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
}
extern "C" int sc_conv3x3_tile_channels_split(int hw) {
#define CALL(C) C::CT
    SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
}
extern "C" int sc_conv3x3_forward_split(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout,
                                        int hw, void* stream) {
#define CALL(C) sc::launch_conv<C>(x, w_pack, out, workspace, batch, cin, cout, (hipStream_t)stream)
    SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
}

extern "C" long long sc_conv3x3_pack_floats(int cin, int cout, int hw) {
#define CALL(C) sc::pack_floats<C>(cin, cout)
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}

extern "C" long long sc_conv3x3_workspace_floats(int hw) {
#define CALL(C) sc::workspace_floats<C>()
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}

extern "C" int sc_conv3x3_tile_channels(int hw) {
#define CALL(C) C::CT
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}

// `unit_form` != 0: the caller guarantees that every row is a split image with 64-channel tiles (flags & 2, CT = 64): the unit kernel.
extern "C" int sc_conv3x3_pack_multi_units(const long long* table, int n, float* dst, long long total, void* stream) {
    if (n <= 0 || total <= 0) return 0;
    if (n > 128 || total % 13824) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sc::conv3x3_pack_multi_split_kernel, dim3((unsigned)(total / 13824)), dim3(256), 0, (hipStream_t)stream, table, n, dst);
    return (int)hipGetLastError();
}

extern "C" int sc_conv3x3_pack_multi(const long long* table, int n, float* dst, long long total, void* stream) {
    if (n <= 0 || total <= 0) return 0;
    if (n > 128) return (int)hipErrorInvalidValue;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(sc::conv3x3_pack_multi_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, table, n, dst,
                       total);
    return (int)hipGetLastError();
}

extern "C" int sc_conv3x3_pack(const float* w, float* w_pack, int cin, int cout, int hw, int transpose_flip, void* stream) {
    if (transpose_flip & 4) {           /* bit 2: the image of sc_conv3x3s2_forward (hw = side of the INPUT map) */
#define CALL(C) sc::launch_pack<C>(w, w_pack, cin, cout, transpose_flip & 1, (hipStream_t)stream)
        SC_CONV_DISPATCH_S2(hw, CALL)
#undef CALL
    }
    if (transpose_flip & 2) {           /* bit 1: the three-piece bf16 image of sc_conv3x3_forward_split */
#define CALL(C) sc::launch_pack<C>(w, w_pack, cin, cout, transpose_flip & 1, (hipStream_t)stream)
        SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
    }
#define CALL(C) sc::launch_pack<C>(w, w_pack, cin, cout, transpose_flip & 1, (hipStream_t)stream)
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}

// out = conv(x, w) + addend (addend [batch][cout][hw][hw], may alias nothing else): the sum is formed in the store epilogue
extern "C" int sc_conv3x3_forward_add(const float* x, const float* w_pack, const float* addend, float* out, float* workspace, int batch, int cin,
                                      int cout, int hw, int split, void* stream) {
    if (split) {
#define CALL(C) sc::launch_conv<C>(x, w_pack, out, workspace, batch, cin, cout, (hipStream_t)stream, addend)
        SC_CONV_DISPATCH_SPLIT(hw, CALL)
#undef CALL
    }
#define CALL(C) sc::launch_conv<C>(x, w_pack, out, workspace, batch, cin, cout, (hipStream_t)stream, addend)
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}

extern "C" int sc_conv3x3_forward(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout, int hw,
                                  void* stream) {
#define CALL(C) sc::launch_conv<C>(x, w_pack, out, workspace, batch, cin, cout, (hipStream_t)stream)
    SC_CONV_DISPATCH(hw, CALL)
#undef CALL
}
