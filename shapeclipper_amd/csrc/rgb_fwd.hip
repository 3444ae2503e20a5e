// rgb_fwd.hip -- RGB MLP + Laplace density + VolSDF alpha compositing, one wavefront per ray.
//
// Replaces, for one render call: RGBNetwork.forward (model/implicit.py:220-239),
// LaplaceDensity.density_func (:65-83), Renderer.volume_rendering (model/renderer.py:187-209) and the
// per-ray reductions of Renderer.forward (:117-152).  Consumes what sdf_fwd.hip left in HBM
// (sdf, d sdf/dx, 64-channel feature in TBL64 layout).
//
// A wave owns one ray = 64 samples = four 16-point MFMA tiles.  After the RGB chain of tile k the
// per-point results (density, colour, unit normal) are kept by lane group g==k, so that at the end
// lane i holds sample i of the ray and the compositing is a 64-lane scan + eight wave reductions.
#include "rgb_common.hpp"
#include "mlp_presplit.hpp"

namespace sc {

struct RgbFwdArgs {
    const float* points;     // [n_rays*64][3]
    const float* z_vals;     // [n_rays][64]
    const float* depth_fac;  // [n_rays]
    const float* sdf;        // [n_rays*64]
    const float* grad;       // [n_rays*64][3]   d sdf / d point
    const float* feat;       // TBL64
    const float* v;          // RgbPack image
    const float* dbias;      // [n_images][3][64]
    const float* beta_param; // scalar parameter (renderer.density.beta), device memory
    int n_rays, rays_per_image, n_images, symmetric;
    float beta_min, bgcolor, normal_pow;
    float* rgb;        // [n_rays][3]
    float* mask;       // [n_rays]
    float* mask_hard;  // [n_rays]   {0,1}
    float* depth;      // [n_rays]
    float* normal;     // [n_rays][3]
    float* weights;    // [n_rays][64] or null (kept for tests / visualisation: alpha in `alpha`)
    float* alpha;      // [n_rays][64] or null
    float* rgb_flat;   // [n_rays*64][3] or null (visualisation path, renderer.py:176)
    float* rr;         // null, or 3 x TBL64 (layer-major): the post-ReLU activations r0, r1, r2 of the hidden layers, parked for
                       // sc_rgb_composite_backward_fused_stash (which then does not recompute the forward chain)
};

// STASH: rr and rgb_flat are given (the training call): compile-time, so that the stores of the parked activations sit in no branch
template <bool STASH>
__global__ __launch_bounds__(256) void rgb_composite_fwd_kernel(RgbFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_rgb_weights(lds, a.v, threadIdx.x, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    RgbLanePtrs L(lds, p, g);
    const float beta = fabsf(a.beta_param[0]) + a.beta_min;

    for (int ray = blockIdx.x * 4 + wave; ray < a.n_rays; ray += gridDim.x * 4) {
        const int img = min(ray / a.rays_per_image, a.n_images - 1);
        const float* db = a.dbias + (size_t)img * 192 + 4 * g;
        float sigma = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int tile = ray * 4 + k;
            const size_t pt = (size_t)tile * TP + p;
            const float x0 = a.points[pt * 3 + 0], x1 = a.points[pt * 3 + 1], x2 = a.points[pt * 3 + 2];
            float f[ACT_STEPS];
            tbl_load(a.feat, tile, p, g, f);
            // every input of the tile is requested before the chain (the parked activations are stored behind it: a load below those stores
            // could not be moved above them by the compiler and would wait with nothing to hide behind)
            const float s = a.sdf[pt];
            const float gx = a.grad[pt * 3 + 0], gy = a.grad[pt * 3 + 1], gz = a.grad[pt * 3 + 2];
            float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
            pe_slots<false, false>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
            float y[3][ACT_STEPS];
            float col[3];
            rgb_chain(L, db, e, f, y, col);
            if (STASH || a.rr) {
                const size_t tbl = (size_t)a.n_rays * 4 * 1024;
                tbl_store(a.rr + 0 * tbl, tile, p, g, y[0]);
                tbl_store(a.rr + 1 * tbl, tile, p, g, y[1]);
                tbl_store(a.rr + 2 * tbl, tile, p, g, y[2]);
            }
            const float ex = expf(-fabsf(s) / beta);
            const float sg = (1.f / beta) * (s >= 0.f ? 0.5f * ex : 1.f - 0.5f * ex);
            // normal_flat = -d(density)/dx = (0.5/beta^2) exp(-|s|/beta) * g; then F.normalize (eps 1e-12)
            const float kk = (0.5f / (beta * beta)) * ex;
            const float vx = kk * gx, vy = kk * gy, vz = kk * gz;
            const float inv = 1.f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
            if (g == k) {
                sigma = sg; c0 = col[0]; c1 = col[1]; c2 = col[2];
                n0 = vx * inv; n1 = vy * inv; n2 = vz * inv;
            }
        }
        // ---- compositing over the 64 samples of the ray (lane == sample index) ----
        const float z = a.z_vals[(size_t)ray * 64 + lane];
        const float znext = __shfl_down(z, 1);
        const float delta = lane == 63 ? 0.f : znext - z;
        const float E = delta * sigma;
        const float alpha = 1.f - expf(-E);
        const float T = expf(-(wave_inclusive_scan(E) - E));
        const float w = alpha * T;
        const float wn = a.normal_pow == 1.f ? w : powf(w, a.normal_pow);
        const float dfac = a.depth_fac[ray];
        const float acc = wave_sum(w);
        const float dep = wave_sum(w * (z * dfac));
        const float r0 = wave_sum(w * c0), r1 = wave_sum(w * c1), r2 = wave_sum(w * c2);
        const float m0 = wave_sum(wn * n0), m1 = wave_sum(wn * n1), m2 = wave_sum(wn * n2);
        if (a.weights) a.weights[(size_t)ray * 64 + lane] = w;
        if (a.alpha) a.alpha[(size_t)ray * 64 + lane] = alpha;
        if (STASH || a.rgb_flat) {
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 0] = c0;
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 1] = c1;
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 2] = c2;
        }
        if (lane == 0) {
            const float bg = (1.f - acc) * a.bgcolor;
            a.rgb[(size_t)ray * 3 + 0] = r0 + bg;
            a.rgb[(size_t)ray * 3 + 1] = r1 + bg;
            a.rgb[(size_t)ray * 3 + 2] = r2 + bg;
            a.mask[ray] = acc;
            a.mask_hard[ray] = acc > 0.5f ? 1.f : 0.f;
            a.depth[ray] = dep;
            const float inv = 1.f / fmaxf(sqrtf(m0 * m0 + m1 * m1 + m2 * m2), 1e-12f);
            a.normal[(size_t)ray * 3 + 0] = m0 * inv;
            a.normal[(size_t)ray * 3 + 1] = m1 * inv;
            a.normal[(size_t)ray * 3 + 2] = m2 * inv;
        }
    }
}

// ---- round 6: the same pass with the RGB network in the exact three-piece bf16 split arithmetic, weights pre-split in LDS ------------------
// (mlp_presplit.hpp; the arithmetic of the trunk convolutions: error against float64 that of the fp32 chain).  90 KiB of fragments
// ([V0 feature | V0 encoding | V1 | V2]), so ONE 8-wave workgroup per CU instead of two 4-wave ones; a wave still owns a ray and walks its
// four tiles two at a time (a weight fragment read feeds both).  Per tile 144 K = 32 + 24 K = 16 MFMAs + 24 = 2.9 k matrix cycles against
// 7.7 k of the fp32 form.  Everything behind the chain (density, normal, compositing, the parked activations) is the code above.
namespace rs {
using namespace ps;
constexpr int WAVES = 8;
constexpr int OFF_V0F = 0;                         // [ks][mt]: feature columns 48..111 of V0
constexpr int OFF_V0E = OFF_V0F + HID_BYTES;       // [mt]: encoding columns 0..47 of V0
constexpr int OFF_V1 = OFF_V0E + PE_BYTES;
constexpr int OFF_V2 = OFF_V1 + HID_BYTES;
constexpr int OFF_V3 = OFF_V2 + HID_BYTES;         // fp32: [3][64] + b3[3] (+1 pad)
constexpr int LDS_BYTES = OFF_V3 + (3 * 64 + 4) * 4;
}  // namespace rs

template <bool STASH>
__global__ __launch_bounds__(64 * rs::WAVES) void rgb_composite_fwd_split_kernel(RgbFwdArgs a) {
    using namespace rs;
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    {
        const int tid = threadIdx.x, nt = 64 * WAVES;
        stage_hidden(lds_c + OFF_V0F, a.v + RgbPack::V0, 112, 48, tid, nt);
        stage_pe(lds_c + OFF_V0E, a.v + RgbPack::V0, 112, 0, tid, nt);
        stage_hidden(lds_c + OFF_V1, a.v + RgbPack::V1, 64, 0, tid, nt);
        stage_hidden(lds_c + OFF_V2, a.v + RgbPack::V2, 64, 0, tid, nt);
        float* v3 = reinterpret_cast<float*>(lds_c + OFF_V3);
        if (tid < 3 * 64 + 3) v3[tid] = a.v[RgbPack::V3 + tid];       // V3 [3][64] and b3 [3] are contiguous in the pack
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, g = lane >> 4;
    const float* v3 = reinterpret_cast<const float*>(lds_c + OFF_V3) + 4 * g;
    const float* b3 = reinterpret_cast<const float*>(lds_c + OFF_V3) + 3 * 64;
    const float beta = fabsf(a.beta_param[0]) + a.beta_min;
    const size_t tbl = (size_t)a.n_rays * 4 * 1024;

    for (int ray = blockIdx.x * WAVES + wave; ray < a.n_rays; ray += gridDim.x * WAVES) {
        const int img = min(ray / a.rays_per_image, a.n_images - 1);
        const float* db = a.dbias + (size_t)img * 192 + 4 * g;
        float sigma = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
#pragma unroll 1
        for (int kk = 0; kk < 2; ++kk) {
            MlpPieces<8> e32[2], fp[2][2], hp[2][2];
            MlpPieces<4> e16[2];
            float s[2], gx[2], gy[2], gz[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tile = ray * 4 + 2 * kk + u;
                const size_t pt = (size_t)tile * TP + p;
                const float x0 = a.points[pt * 3 + 0], x1 = a.points[pt * 3 + 1], x2 = a.points[pt * 3 + 2];
                float f[ACT_STEPS];
                tbl_load(a.feat, tile, p, g, f);
                s[u] = a.sdf[pt];
                gx[u] = a.grad[pt * 3 + 0], gy[u] = a.grad[pt * 3 + 1], gz[u] = a.grad[pt * 3 + 2];
                float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
                pe_slots<false, false>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
                split_pe(e, e32[u], e16[u]);
                split_act(f, fp[u]);
            }
            f32x4 acc[2][NT];
            float r[2][ACT_STEPS];
            // layer l: acc = bias; products; r = relu(acc); parked when asked for; split for the next layer
#define SC_RGB_LAYER_END(L, NEXT)                                                                           \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                  \
                relu_from_acc(acc[u], r[u]);                                                                 \
                if (STASH || a.rr) tbl_store_pinned(a.rr + (size_t)(L) * tbl, ray * 4 + 2 * kk + u, p, g, r[u]);   \
                if (NEXT) split_act(r[u], hp[u]);                                                            \
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) acc_init(acc[u], db);
            hidden_part<2>(lds_c + OFF_V0F, lane, fp, acc);
            pe_part<2>(lds_c + OFF_V0E, lane, e32, e16, acc);
            SC_RGB_LAYER_END(0, true)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc_init(acc[u], db + 64);
            hidden_part<2>(lds_c + OFF_V1, lane, hp, acc);
            SC_RGB_LAYER_END(1, true)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc_init(acc[u], db + 128);
            hidden_part<2>(lds_c + OFF_V2, lane, hp, acc);
            SC_RGB_LAYER_END(2, false)
#undef SC_RGB_LAYER_END
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float col[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float part = 0.f;
#pragma unroll
                    for (int q = 0; q < ACT_STEPS; ++q) part = __builtin_fmaf(v3[j * 64 + kp(q)], r[u][q], part);
                    const float yv = group_sum(part) + b3[j];
                    col[j] = 1.f / (1.f + expf(-yv));
                }
                const float ex = expf(-fabsf(s[u]) / beta);
                const float sg = (1.f / beta) * (s[u] >= 0.f ? 0.5f * ex : 1.f - 0.5f * ex);
                const float kq = (0.5f / (beta * beta)) * ex;
                const float vx = kq * gx[u], vy = kq * gy[u], vz = kq * gz[u];
                const float inv = 1.f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
                if (g == 2 * kk + u) {
                    sigma = sg; c0 = col[0]; c1 = col[1]; c2 = col[2];
                    n0 = vx * inv; n1 = vy * inv; n2 = vz * inv;
                }
            }
        }
        // ---- compositing over the 64 samples of the ray (lane == sample index): the code of rgb_composite_fwd_kernel ----
        const float z = a.z_vals[(size_t)ray * 64 + lane];
        const float znext = __shfl_down(z, 1);
        const float delta = lane == 63 ? 0.f : znext - z;
        const float E = delta * sigma;
        const float alpha = 1.f - expf(-E);
        const float T = expf(-(wave_inclusive_scan(E) - E));
        const float w = alpha * T;
        const float wn = a.normal_pow == 1.f ? w : powf(w, a.normal_pow);
        const float dfac = a.depth_fac[ray];
        const float acc_w = wave_sum(w);
        const float dep = wave_sum(w * (z * dfac));
        const float r0 = wave_sum(w * c0), r1 = wave_sum(w * c1), r2 = wave_sum(w * c2);
        const float m0 = wave_sum(wn * n0), m1 = wave_sum(wn * n1), m2 = wave_sum(wn * n2);
        if (a.weights) a.weights[(size_t)ray * 64 + lane] = w;
        if (a.alpha) a.alpha[(size_t)ray * 64 + lane] = alpha;
        if (STASH || a.rgb_flat) {
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 0] = c0;
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 1] = c1;
            a.rgb_flat[((size_t)ray * 64 + lane) * 3 + 2] = c2;
        }
        if (lane == 0) {
            const float bg = (1.f - acc_w) * a.bgcolor;
            a.rgb[(size_t)ray * 3 + 0] = r0 + bg;
            a.rgb[(size_t)ray * 3 + 1] = r1 + bg;
            a.rgb[(size_t)ray * 3 + 2] = r2 + bg;
            a.mask[ray] = acc_w;
            a.mask_hard[ray] = acc_w > 0.5f ? 1.f : 0.f;
            a.depth[ray] = dep;
            const float inv = 1.f / fmaxf(sqrtf(m0 * m0 + m1 * m1 + m2 * m2), 1e-12f);
            a.normal[(size_t)ray * 3 + 0] = m0 * inv;
            a.normal[(size_t)ray * 3 + 1] = m1 * inv;
            a.normal[(size_t)ray * 3 + 2] = m2 * inv;
        }
    }
}

}  // namespace sc

// sc_rgb_composite_forward_stash with the RGB network in the exact bf16x3 split arithmetic (pre-split weights in LDS); same operands,
// same outputs (the colours differ from the fp32-MFMA form by fp32 rounding only; mask / mask_hard / depth / normal do not depend on the
// RGB network and are bit-identical).
extern "C" int sc_rgb_composite_forward_split(const float* points, const float* z_vals, const float* depth_fac,
                                              const float* sdf, const float* grad, const float* feat,
                                              const float* v_pack, const float* dbias, const float* beta_param,
                                              int n_rays, int rays_per_image, int n_images, int symmetric,
                                              float beta_min, float bgcolor, float normal_pow,
                                              float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                                              float* weights, float* alpha, float* rgb_flat, float* rr, void* stream_) {
    if (n_rays <= 0) return 0;
    sc::RgbFwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, n_rays, rays_per_image,
                     n_images, symmetric, beta_min, bgcolor, normal_pow, rgb, mask, mask_hard, depth, normal,
                     weights, alpha, rgb_flat, rr};
    int blocks = (n_rays + sc::rs::WAVES - 1) / sc::rs::WAVES;
    if (blocks > 256) blocks = 256;   // one 8-wave workgroup per CU (90 KiB of pre-split fragments)
    if (rr && rgb_flat) {
        (void)hipFuncSetAttribute((const void*)sc::rgb_composite_fwd_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, sc::rs::LDS_BYTES);
        hipLaunchKernelGGL(sc::rgb_composite_fwd_split_kernel<true>, dim3(blocks), dim3(64 * sc::rs::WAVES), sc::rs::LDS_BYTES, (hipStream_t)stream_, a);
    } else {
        (void)hipFuncSetAttribute((const void*)sc::rgb_composite_fwd_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, sc::rs::LDS_BYTES);
        hipLaunchKernelGGL(sc::rgb_composite_fwd_split_kernel<false>, dim3(blocks), dim3(64 * sc::rs::WAVES), sc::rs::LDS_BYTES, (hipStream_t)stream_, a);
    }
    return (int)hipGetLastError();
}

// sc_rgb_composite_forward that also parks the hidden activations r0, r1, r2 (rr: 3 x TBL64 = 3 x n_rays * 4 * 1024 floats, or null).
extern "C" int sc_rgb_composite_forward_stash(const float* points, const float* z_vals, const float* depth_fac,
                                              const float* sdf, const float* grad, const float* feat,
                                              const float* v_pack, const float* dbias, const float* beta_param,
                                              int n_rays, int rays_per_image, int n_images, int symmetric,
                                              float beta_min, float bgcolor, float normal_pow,
                                              float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                                              float* weights, float* alpha, float* rgb_flat, float* rr, void* stream_) {
    if (n_rays <= 0) return 0;
    sc::RgbFwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, n_rays, rays_per_image,
                     n_images, symmetric, beta_min, bgcolor, normal_pow, rgb, mask, mask_hard, depth, normal,
                     weights, alpha, rgb_flat, rr};
    int blocks = (n_rays + 3) / 4;
    if (blocks > 512) blocks = 512;   // two 4-wave workgroups per CU (63 KiB LDS each)
    const size_t lds_bytes = sc::RgbLds::TOTAL * sizeof(float);
    if (rr && rgb_flat) hipLaunchKernelGGL(sc::rgb_composite_fwd_kernel<true>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(sc::rgb_composite_fwd_kernel<false>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

extern "C" int sc_rgb_composite_forward(const float* points, const float* z_vals, const float* depth_fac,
                                        const float* sdf, const float* grad, const float* feat,
                                        const float* v_pack, const float* dbias, const float* beta_param,
                                        int n_rays, int rays_per_image, int n_images, int symmetric,
                                        float beta_min, float bgcolor, float normal_pow,
                                        float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                                        float* weights, float* alpha, float* rgb_flat, void* stream_) {
    return sc_rgb_composite_forward_stash(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, n_rays, rays_per_image, n_images,
                                          symmetric, beta_min, bgcolor, normal_pow, rgb, mask, mask_hard, depth, normal, weights, alpha, rgb_flat,
                                          nullptr, stream_);
}
