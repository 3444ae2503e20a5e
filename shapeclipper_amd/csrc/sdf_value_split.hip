// sdf_value_split.hip -- value-only SDF MLP (SDFNetwork.forward, model/implicit.py:138-161, without the feature rows and without
// d sdf/dx) in the exact three-piece bf16 split arithmetic, weights PRE-SPLIT in LDS.  What compute_level_grid needs
// (utils/eval_3D.py:9-38: the SDF on the (N+1)^3 evaluation grid), round 6.
//
// Why this kernel exists: the chain kernels of mlp_tile.hpp run fp32 MFMAs
// (v_mfma_f32_16x16x4_f32, 32 cycles for 2,048 FLOP) and sit at ~0.67 of that pipe; the same products from exact three-piece bf16 splits
// (x = p0 + p1 + p2, six piece products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the arithmetic of the trunk convolutions) need
// 3.1x fewer matrix cycles.  Round 5 built that arithmetic INTO the existing kernels (SC_MLP_SPLIT=1) and it was slower: every wave split
// the weights again for every 16-point tile (80 % of the added vector instructions), and pre-split weights (6 bytes per value) of the
// whole network do not fit 160 KiB.  The value-only chain drops the 64 feature rows of the output layer and every transposed product, and
// then they do fit: 150 KiB of fragment-ordered bf16 pieces, built once per workgroup.  Per 16-point tile: 264 K=32 + 72 K=16 MFMAs
// (4.8 k matrix cycles against 14.8 k) and only the ACTIVATIONS are split on the fly (16 values per lane and layer).  A wave walks two
// tiles at a time so that a weight fragment read from LDS (3 x ds_read_b128) feeds 12 MFMAs.
//
// Operand layouts (v_mfma_f32_16x16x32_bf16: A lane l = row l & 15, K values 8 (l >> 4) .. + 7; B lane l = column l & 15, the same K
// values; D lane l, register r = row 4 (l >> 4) + r, column l & 15) -- with the weights as A and the 16 points as B, register r of
// channel tile T of lane group g is channel 16 T + 4 g + r of point l & 15, so a lane's eight registers h[8 ks .. 8 ks + 7] ARE its B
// fragment of K-step ks when K index 8 g + j stands for channel 16 (2 ks + j / 4) + 4 g + j % 4: the fragments are packed in that order.
// The positional encoding arrives in the slot order of mlp_tile.hpp (e[s], s = 4 c + j <-> packed column 4 s + g): one K = 32 step
// (e[0..7]) and one K = 16 step (e[8..11]).
//
// LDS image (bytes): per (K-step, channel tile) a FRAGMENT = [piece 0..2][lane 0..63][8 bf16] (3 KiB; K = 16: [piece][lane][4 bf16],
// 1.5 KiB), every piece one lane-linear KiB: conflict-free ds_read_b128.
#include "mlp_tile.hpp"

namespace sc {
namespace vs {

constexpr int WAVES = 8;                         // 2 per SIMD
constexpr int F32B = 3 * 64 * 16, F16B = 3 * 64 * 8;
constexpr int PE_FRAG = F32B + F16B;             // the encoding's two steps of one channel tile
constexpr int OFF_L0 = 0;                        // [mt 4] PE_FRAG
constexpr int OFF_L1 = OFF_L0 + 4 * PE_FRAG;     // [ks 2][mt 4] F32B, then [mt 4] PE_FRAG
constexpr int OFF_L2 = OFF_L1 + 8 * F32B + 4 * PE_FRAG;
constexpr int OFF_L3 = OFF_L2 + 8 * F32B + 4 * PE_FRAG;
constexpr int OFF_L4 = OFF_L3 + 8 * F32B;
constexpr int OFF_W5 = OFF_L4 + 8 * F32B;        // 64 floats: the sdf row of the output layer, + b5[0], fp32
constexpr int LDS_BYTES = OFF_W5 + 80 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "the pre-split value chain must fit one CU's LDS");

__device__ __forceinline__ void split3(float v, __bf16& h0, __bf16& h1, __bf16& h2) {
    h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    h1 = (__bf16)r1;
    h2 = (__bf16)(r1 - (float)h1);
}

// hidden-input part of a layer: W[64][ld], columns c0 .. c0 + 63 -> [ks][mt] K = 32 fragments
__device__ __forceinline__ void stage_hidden(char* dst, const float* __restrict__ W, int ld, int c0, int tid, int nthreads) {
    for (int idx = tid; idx < 2 * 4 * 64 * 8; idx += nthreads) {
        const int j = idx & 7, lane = (idx >> 3) & 63, mt = (idx >> 9) & 3, ks = idx >> 11;
        const int row = 16 * mt + (lane & 15), col = c0 + 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
        __bf16 h[3];
        split3(W[row * ld + col], h[0], h[1], h[2]);
        char* f = dst + (ks * 4 + mt) * F32B + lane * 16 + j * 2;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<__bf16*>(f + p * 1024) = h[p];
    }
}
// encoding part: packed columns c0 .. c0 + 47 -> [mt] (K = 32 fragment of slots 0..7, K = 16 fragment of slots 8..11)
__device__ __forceinline__ void stage_pe(char* dst, const float* __restrict__ W, int ld, int c0, int tid, int nthreads) {
    for (int idx = tid; idx < 4 * 64 * 12; idx += nthreads) {
        const int s = idx % 12, lane = (idx / 12) & 63, mt = idx / (12 * 64);
        const int row = 16 * mt + (lane & 15), col = c0 + 4 * s + (lane >> 4);
        __bf16 h[3];
        split3(W[row * ld + col], h[0], h[1], h[2]);
        char* f = dst + mt * PE_FRAG;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (s < 8) *reinterpret_cast<__bf16*>(f + p * 1024 + lane * 16 + s * 2) = h[p];
            else *reinterpret_cast<__bf16*>(f + F32B + p * 512 + lane * 8 + (s - 8) * 2) = h[p];
        }
    }
}

__device__ __forceinline__ MlpPieces<8> frag32(const char* f, int lane) {
    MlpPieces<8> a;
#pragma unroll
    for (int p = 0; p < 3; ++p) a.p[p] = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(f + p * 1024 + lane * 16));
    return a;
}
__device__ __forceinline__ MlpPieces<4> frag16(const char* f, int lane) {
    MlpPieces<4> a;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const uint2 v = *reinterpret_cast<const uint2*>(f + p * 512 + lane * 8);
        a.p[p] = __builtin_bit_cast(mlp_bf16x8, make_uint4(v.x, v.y, 0u, 0u));
    }
    return a;
}

struct Args {
    const float* points;   // [n_points][3]
    const float* w;        // SdfPack image (fp32, global)
    const float* cbias;    // [n_images][5][64]
    int n_points, n_per_image, n_images, symmetric;
    float* sdf;            // [n_points]
};

constexpr int TPW = 2;     // tiles a wave walks together (one weight-fragment read feeds both)

__global__ __launch_bounds__(64 * WAVES) void sdf_value_split_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int tid = threadIdx.x, nt = 64 * WAVES;
        stage_pe(lds + OFF_L0, a.w + SdfPack::W0, 48, 0, tid, nt);
        stage_hidden(lds + OFF_L1, a.w + SdfPack::W1, 112, 0, tid, nt);
        stage_pe(lds + OFF_L1 + 8 * F32B, a.w + SdfPack::W1, 112, 64, tid, nt);
        stage_hidden(lds + OFF_L2, a.w + SdfPack::W2, 112, 0, tid, nt);
        stage_pe(lds + OFF_L2 + 8 * F32B, a.w + SdfPack::W2, 112, 64, tid, nt);
        stage_hidden(lds + OFF_L3, a.w + SdfPack::W3, 64, 0, tid, nt);
        stage_hidden(lds + OFF_L4, a.w + SdfPack::W4, 64, 0, tid, nt);
        float* w5 = reinterpret_cast<float*>(lds + OFF_W5);
        if (tid < 64) w5[tid] = a.w[SdfPack::W5 + tid];
        if (tid == 64) w5[64] = a.w[SdfPack::B5];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int ntiles = (a.n_points + TP - 1) / TP, ngroups = (ntiles + TPW - 1) / TPW;
    const float* w5s = reinterpret_cast<const float*>(lds + OFF_W5) + 4 * g;
    const float b5 = reinterpret_cast<const float*>(lds + OFF_W5)[64];

    for (int grp = blockIdx.x * WAVES + wave; grp < ngroups; grp += gridDim.x * WAVES) {
        MlpPieces<8> e32[TPW];
        MlpPieces<4> e16[TPW];
        const float* cb[TPW];
        int pt[TPW];
        bool valid[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            pt[u] = (grp * TPW + u) * TP + p;
            valid[u] = pt[u] < a.n_points;
            const int ptc = valid[u] ? pt[u] : a.n_points - 1;
            const float x0 = a.points[(size_t)ptc * 3 + 0], x1 = a.points[(size_t)ptc * 3 + 1], x2 = a.points[(size_t)ptc * 3 + 2];
            cb[u] = a.cbias + (size_t)min(ptc / a.n_per_image, a.n_images - 1) * 320 + 4 * g;
            float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
            pe_slots<false, false>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
            const float ea[8] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]}, eb[4] = {e[8], e[9], e[10], e[11]};
            mlp_split<8>(ea, e32[u]);
            mlp_split<4>(eb, e16[u]);
        }
        f32x4 acc[TPW][NT];
        MlpPieces<8> hp[TPW][2];
        float h[TPW][ACT_STEPS];

        // acc[u] += W_e e   (the encoding's part of a layer: fragments at `base`)
        // (all K = 32 products first, then the K = 16 ones: a 16x16x16 MFMA issued DIRECTLY behind the 16x16x32 MFMA that writes its
        //  accumulator read the accumulator before that write had landed -- values off by the main piece product, 0.4 on the level grid;
        //  hipcc 7.2 under-counts the passes of the gfx950 K = 32 shape, DESIGN.md 4.1.1.  Here 7 independent products lie between.)
        auto pe_part = [&](const char* base) {
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const MlpPieces<8> w32 = frag32(base + mt * PE_FRAG, lane);
#pragma unroll
                for (int u = 0; u < TPW; ++u) acc[u][mt] = mlp_six<8>(w32, e32[u], acc[u][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const MlpPieces<4> w16 = frag16(base + mt * PE_FRAG + F32B, lane);
#pragma unroll
                for (int u = 0; u < TPW; ++u) acc[u][mt] = mlp_six<4>(w16, e16[u], acc[u][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // acc[u] += W_h h   (hidden part: fragments [ks][mt] at `base`)
        auto hidden_part = [&](const char* base) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    const MlpPieces<8> w = frag32(base + (ks * 4 + mt) * F32B, lane);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) acc[u][mt] = mlp_six<8>(w, hp[u][ks], acc[u][mt]);
                }
        };
        // h = softplus(acc); split for the next layer
        auto activate = [&](bool split) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
#pragma unroll
                for (int s = 0; s < ACT_STEPS; ++s) {
                    const float av = acc[u][s >> 2][s & 3];
                    float t, r;
                    softplus_parts(av, t, r);
                    h[u][s] = softplus_val(av, t);
                }
                if (split) {
                    const float ha[8] = {h[u][0], h[u][1], h[u][2], h[u][3], h[u][4], h[u][5], h[u][6], h[u][7]};
                    const float hb[8] = {h[u][8], h[u][9], h[u][10], h[u][11], h[u][12], h[u][13], h[u][14], h[u][15]};
                    mlp_split<8>(ha, hp[u][0]);
                    mlp_split<8>(hb, hp[u][1]);
                }
            }
        };
        auto bias = [&](int L) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) acc_init(acc[u], cb[u] + L * 64);
        };

        bias(0);
        pe_part(lds + OFF_L0);
        activate(true);
        bias(1);
        hidden_part(lds + OFF_L1);
        pe_part(lds + OFF_L1 + 8 * F32B);
        activate(true);
        bias(2);
        hidden_part(lds + OFF_L2);
        pe_part(lds + OFF_L2 + 8 * F32B);
        activate(true);
        bias(3);
        hidden_part(lds + OFF_L3);
        activate(true);
        bias(4);
        hidden_part(lds + OFF_L4);
        activate(false);

        // output layer, sdf row only: the fp32 dot of sdf_fwd.hip (same order of operations)
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            float sp = 0.f;
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) sp = __builtin_fmaf(w5s[kp(s)], h[u][s], sp);
            const float sdf = group_sum(sp) + b5;
            if (valid[u] && g == 0) a.sdf[pt[u]] = sdf;
        }
    }
}

}  // namespace vs
}  // namespace sc

// Value-only SDF of n_points points (image-major, n_per_image each) in the split arithmetic.  Same operands as sc_sdf_forward
// (include/shapeclipper_hip.h); sdf [n_points] is the only output.
extern "C" int sc_sdf_value_forward_split(const float* points, const float* w_pack, const float* cbias, int n_points, int n_per_image,
                                          int n_images, int symmetric, float* sdf, void* stream_) {
    if (n_points <= 0) return 0;
    if (!points || !w_pack || !cbias || !sdf || n_per_image <= 0 || n_images <= 0) return (int)hipErrorInvalidValue;
    sc::vs::Args a{points, w_pack, cbias, n_points, n_per_image, n_images, symmetric, sdf};
    const int ntiles = (n_points + sc::TP - 1) / sc::TP, ngroups = (ntiles + sc::vs::TPW - 1) / sc::vs::TPW;
    int blocks = (ngroups + sc::vs::WAVES - 1) / sc::vs::WAVES;
    if (blocks > 256) blocks = 256;      // one persistent 8-wave workgroup per CU (LDS-resident pre-split weights)
    (void)hipFuncSetAttribute((const void*)sc::vs::sdf_value_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, sc::vs::LDS_BYTES);
    hipLaunchKernelGGL(sc::vs::sdf_value_split_kernel, dim3(blocks), dim3(64 * sc::vs::WAVES), sc::vs::LDS_BYTES, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
