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
#include "mlp_presplit.hpp"

namespace sc {
namespace vs {

constexpr int WAVES = 8;                         // 2 per SIMD
using namespace ps;
constexpr int OFF_L0 = 0;                        // [mt 4] PE_FRAG
constexpr int OFF_L1 = OFF_L0 + 4 * PE_FRAG;     // [ks 2][mt 4] F32B, then [mt 4] PE_FRAG
constexpr int OFF_L2 = OFF_L1 + 8 * F32B + 4 * PE_FRAG;
constexpr int OFF_L3 = OFF_L2 + 8 * F32B + 4 * PE_FRAG;
constexpr int OFF_L4 = OFF_L3 + 8 * F32B;
constexpr int OFF_W5 = OFF_L4 + 8 * F32B;        // 64 floats: the sdf row of the output layer, + b5[0], fp32
constexpr int LDS_BYTES = OFF_W5 + 80 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "the pre-split value chain must fit one CU's LDS");

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
            split_pe(e, e32[u], e16[u]);
        }
        f32x4 acc[TPW][NT];
        MlpPieces<8> hp[TPW][2];
        float h[TPW][ACT_STEPS];

        // acc[u] += W_e e   (the encoding's part of a layer: fragments at `base`)
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
                if (split) split_act(h[u], hp[u]);
            }
        };
        auto bias = [&](int L) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) acc_init(acc[u], cb[u] + L * 64);
        };

        bias(0);
        pe_part<TPW>(lds + OFF_L0, lane, e32, e16, acc);
        activate(true);
        bias(1);
        hidden_part<TPW>(lds + OFF_L1, lane, hp, acc);
        pe_part<TPW>(lds + OFF_L1 + 8 * F32B, lane, e32, e16, acc);
        activate(true);
        bias(2);
        hidden_part<TPW>(lds + OFF_L2, lane, hp, acc);
        pe_part<TPW>(lds + OFF_L2 + 8 * F32B, lane, e32, e16, acc);
        activate(true);
        bias(3);
        hidden_part<TPW>(lds + OFF_L3, lane, hp, acc);
        activate(true);
        bias(4);
        hidden_part<TPW>(lds + OFF_L4, lane, hp, acc);
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
