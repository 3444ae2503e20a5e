// rgb_bwd.hip -- reverse pass of rgb_fwd.hip: compositing, density, per-sample normals, RGB MLP.
//
// Hand-derived counterpart of what autograd does for Renderer.forward's tail
// (model/renderer.py:110-152 and volume_rendering :187-209) plus RGBNetwork (model/implicit.py:220-239)
// and LaplaceDensity (:65-83, incl. d/d beta).  One wavefront per ray.
//
// Phase 1 (lane == sample): recompute delta, E, alpha, T, w from z / sdf and the per-sample colours
//   the forward stored; turn the per-ray upstream gradients into per-sample gradients
//   (Gc, G sigma, G n_hat, Gz) with one 64-lane scan + one suffix scan; then
//   G sdf, G(d sdf/dx), G beta, Gz, G depth_fac by the closed-form density / normal derivatives.
// Phase 2 (four 16-point MFMA tiles): recompute the RGB chain, run it in reverse
//   (Gy2 -> V2^T -> Gy1 -> V1^T -> Gy0 -> V0f^T -> G feature, PE Jacobian -> G point) and leave
//   Gy_l / r_l in HBM (TBL64) for the weight-gradient GEMMs (wgrad.hip).
#include "rgb_common.hpp"

namespace sc {

struct RgbBwdArgs {
    const float* points; const float* z_vals; const float* depth_fac; const float* sdf; const float* grad;
    const float* feat; const float* v; const float* dbias; const float* beta_param; const float* rgb_flat;
    int n_rays, rays_per_image, n_images, symmetric;
    float beta_min, bgcolor, normal_pow;
    const float* G_rgb;     // [n_rays][3] or null
    const float* G_mask;    // [n_rays] or null
    const float* G_depth;   // [n_rays] or null
    const float* G_normal;  // [n_rays][3] or null
    float* g_sdf;        // [P]
    float* g_grad;       // [P][3]
    float* g_feat;       // TBL64
    float* g_points;     // [P][3]   (RGB net's own dependence on the point; sdf_bwd adds the rest)
    float* g_z;          // [n_rays][64]
    float* g_depth_fac;  // [n_rays]
    float* g_beta;       // [SC_RGB_BWD_BETA_PARTS = 2048]: one partial per wave of the grid (gradient wrt the raw parameter), fully written;
                         // the gradient is their sum in index order (sc_partial_reduce) -- it used to be one float atomicAdd per wave
    float* gy;           // 3 x TBL64: pre-activation gradients Gy0, Gy1, Gy2
    float* rr;           // 3 x TBL64: post-ReLU activations r0, r1, r2
    float* gy3;          // [P][3]: gradient at the pre-sigmoid output
    float* v3_part;      // null, or [SC_RGB_BWD_BETA_PARTS][196]: per wave of the grid the sums over its points of gy3_j * r2[ch] (= dV3 [3][64]),
                         // of gy3_j (= db3 [3]) and one zero -- fully written; their sum in index order (sc_partial_reduce) is the gradient of
                         // the output layer.  With it rr[2] and gy3 are NOT written (nobody else reads them): 280 MB less per launch.
};

__global__ __launch_bounds__(256, 2) void rgb_composite_bwd_kernel(RgbBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_rgb_weights(lds, a.v, threadIdx.x, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    RgbLanePtrs L(lds, p, g);
    const float bp = a.beta_param[0];
    const float beta = fabsf(bp) + a.beta_min;
    const float dbeta_dbp = bp > 0.f ? 1.f : (bp < 0.f ? -1.f : 0.f);
    const size_t tbl = (size_t)a.n_rays * 4 * 1024;
    float gbeta_acc = 0.f;
    float v3acc[3][ACT_STEPS], b3acc[3] = {0.f, 0.f, 0.f};     // this lane's share of dV3 (channels kp(s) + 4 g of its point column) and db3
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s2 = 0; s2 < ACT_STEPS; ++s2) v3acc[j][s2] = 0.f;

    for (int ray = blockIdx.x * 4 + wave; ray < a.n_rays; ray += gridDim.x * 4) {
        const int img = min(ray / a.rays_per_image, a.n_images - 1);
        const float* db = a.dbias + (size_t)img * 192 + 4 * g;
        const size_t sp = (size_t)ray * 64 + lane;   // this lane's sample
        // ---------------- phase 1: compositing backward, lane == sample ----------------
        const float z = a.z_vals[sp];
        const float s = a.sdf[sp];
        const float gx = a.grad[sp * 3 + 0], gy_ = a.grad[sp * 3 + 1], gz = a.grad[sp * 3 + 2];
        const float c0 = a.rgb_flat[sp * 3 + 0], c1 = a.rgb_flat[sp * 3 + 1], c2 = a.rgb_flat[sp * 3 + 2];
        const float dfac = a.depth_fac[ray];
        const float ex = expf(-fabsf(s) / beta);
        const float psi = s >= 0.f ? 0.5f * ex : 1.f - 0.5f * ex;
        const float sigma = psi / beta;
        const float kk = (0.5f / (beta * beta)) * ex;
        const float vx = kk * gx, vy = kk * gy_, vz = kk * gz;
        const float vnorm = sqrtf(vx * vx + vy * vy + vz * vz);
        const float vden = fmaxf(vnorm, 1e-12f);
        const float n0 = vx / vden, n1 = vy / vden, n2 = vz / vden;
        const float znext = __shfl_down(z, 1);
        const float delta = lane == 63 ? 0.f : znext - z;
        const float E = delta * sigma;
        const float emE = expf(-E);
        const float alpha = 1.f - emE;
        const float T = expf(-(wave_inclusive_scan(E) - E));
        const float w = alpha * T;
        const float pw = a.normal_pow;
        const float wn = pw == 1.f ? w : powf(w, pw);
        const float N0 = wave_sum(wn * n0), N1 = wave_sum(wn * n1), N2 = wave_sum(wn * n2);

        const float Gr0 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 0] : 0.f;
        const float Gr1 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 1] : 0.f;
        const float Gr2 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 2] : 0.f;
        const float Gm = a.G_mask ? a.G_mask[ray] : 0.f;
        const float Gd = a.G_depth ? a.G_depth[ray] : 0.f;
        float GN0 = 0.f, GN1 = 0.f, GN2 = 0.f;
        if (a.G_normal) {
            const float q0 = a.G_normal[(size_t)ray * 3 + 0], q1 = a.G_normal[(size_t)ray * 3 + 1], q2 = a.G_normal[(size_t)ray * 3 + 2];
            const float nn = sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
            if (nn > 1e-12f) {   // F.normalize backward: (G - (G.n) n) / |N|
                const float o0 = N0 / nn, o1 = N1 / nn, o2 = N2 / nn;
                const float dt = q0 * o0 + q1 * o1 + q2 * o2;
                GN0 = (q0 - dt * o0) / nn; GN1 = (q1 - dt * o1) / nn; GN2 = (q2 - dt * o2) / nn;
            } else {             // clamped branch: output = N / eps
                GN0 = q0 / 1e-12f; GN1 = q1 / 1e-12f; GN2 = q2 / 1e-12f;
            }
        }
        const float Gc0 = w * Gr0, Gc1 = w * Gr1, Gc2 = w * Gr2;
        const float gn_dot = GN0 * n0 + GN1 * n1 + GN2 * n2;
        float Gw = (Gr0 * c0 + Gr1 * c1 + Gr2 * c2) - a.bgcolor * (Gr0 + Gr1 + Gr2) + Gm + Gd * z * dfac;
        Gw += pw == 1.f ? gn_dot : (w > 0.f ? pw * powf(w, pw - 1.f) * gn_dot : 0.f);
        const float Gnh0 = wn * GN0, Gnh1 = wn * GN1, Gnh2 = wn * GN2;
        const float Galpha = Gw * T;
        const float GC = -(Gw * alpha) * T;                      // dL/dC_i, C_i = sum_{j<i} E_j
        const float GE = wave_exclusive_suffix(GC) + Galpha * emE;
        const float Gsigma = GE * delta;
        const float Gdelta = GE * sigma;
        const float Gdelta_prev = __shfl_up(Gdelta, 1);
        float Gz = Gd * w * dfac - (lane == 63 ? 0.f : Gdelta) + (lane == 0 ? 0.f : Gdelta_prev);
        const float Gdfac = wave_sum(Gd * w * z);
        // n_hat = v / max(|v|, eps), v = kk * g
        float Gv0, Gv1, Gv2;
        if (vnorm > 1e-12f) {
            const float dt = Gnh0 * n0 + Gnh1 * n1 + Gnh2 * n2;
            Gv0 = (Gnh0 - dt * n0) / vnorm; Gv1 = (Gnh1 - dt * n1) / vnorm; Gv2 = (Gnh2 - dt * n2) / vnorm;
        } else {
            Gv0 = Gnh0 / 1e-12f; Gv1 = Gnh1 / 1e-12f; Gv2 = Gnh2 / 1e-12f;
        }
        const float Gkk = Gv0 * gx + Gv1 * gy_ + Gv2 * gz;
        const float sgn = s >= 0.f ? 1.f : -1.f;
        const float Gs = -Gsigma * kk - Gkk * sgn * kk / beta;
        const float Gbeta = Gsigma * (-sigma / beta + 0.5f * ex * s / (beta * beta * beta))
                          + Gkk * kk * (-2.f / beta + fabsf(s) / (beta * beta));
        gbeta_acc += Gbeta;
        a.g_sdf[sp] = Gs;
        a.g_grad[sp * 3 + 0] = kk * Gv0;
        a.g_grad[sp * 3 + 1] = kk * Gv1;
        a.g_grad[sp * 3 + 2] = kk * Gv2;
        a.g_z[sp] = Gz;
        if (lane == 0) a.g_depth_fac[ray] = Gdfac;

        // ---------------- phase 2: RGB MLP reverse, four 16-point tiles ----------------
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int tile = ray * 4 + k;
            const size_t pt = (size_t)tile * TP + p;
            const int src = 16 * k + p;
            const float gc0 = __shfl(Gc0, src), gc1 = __shfl(Gc1, src), gc2 = __shfl(Gc2, src);
            const float x0 = a.points[pt * 3 + 0], x1 = a.points[pt * 3 + 1], x2 = a.points[pt * 3 + 2];
            float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
            pe_slots<true, false, true>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
            float f[ACT_STEPS];
            tbl_load(a.feat, tile, p, g, f);
            float r[3][ACT_STEPS];
            float col[3];
            rgb_chain(L, db, e, f, r, col);
            tbl_store(a.rr + 0 * tbl, tile, p, g, r[0]);
            tbl_store(a.rr + 1 * tbl, tile, p, g, r[1]);
            const float y0 = gc0 * col[0] * (1.f - col[0]);
            const float y1 = gc1 * col[1] * (1.f - col[1]);
            const float y2 = gc2 * col[2] * (1.f - col[2]);
            if (a.v3_part) {
#pragma unroll
                for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                    v3acc[0][s2] = __builtin_fmaf(y0, r[2][s2], v3acc[0][s2]);
                    v3acc[1][s2] = __builtin_fmaf(y1, r[2][s2], v3acc[1][s2]);
                    v3acc[2][s2] = __builtin_fmaf(y2, r[2][s2], v3acc[2][s2]);
                }
                b3acc[0] += y0; b3acc[1] += y1; b3acc[2] += y2;       // (every lane group of a point holds the same y: group 0 is taken below)
            } else {
                tbl_store(a.rr + 2 * tbl, tile, p, g, r[2]);
                if (g == 0) { a.gy3[pt * 3 + 0] = y0; a.gy3[pt * 3 + 1] = y1; a.gy3[pt * 3 + 2] = y2; }
            }
            float gyv[ACT_STEPS];
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                const float gr = L.v3[kp(s2)] * y0 + L.v3[64 + kp(s2)] * y1 + L.v3[128 + kp(s2)] * y2;
                gyv[s2] = r[2][s2] > 0.f ? gr : 0.f;
            }
            tbl_store(a.gy + 2 * tbl, tile, p, g, gyv);
            f32x4 acc[NT];
            acc_zero(acc);
            mm_act_t<RgbLds::LD1, NT>(L.v2t, gyv, acc);
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) gyv[s2] = r[1][s2] > 0.f ? acc[s2 >> 2][s2 & 3] : 0.f;
            tbl_store(a.gy + 1 * tbl, tile, p, g, gyv);
            acc_zero(acc);
            mm_act_t<RgbLds::LD1, NT>(L.v1t, gyv, acc);
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) gyv[s2] = r[0][s2] > 0.f ? acc[s2 >> 2][s2 & 3] : 0.f;
            tbl_store(a.gy + 0 * tbl, tile, p, g, gyv);
            acc_zero(acc);
            mm_act_t<RgbLds::LD0, NT>(L.v0ft, gyv, acc);
            float gf[ACT_STEPS];
            acc_to_regs(acc, gf);
            tbl_store(a.g_feat, tile, p, g, gf);
            float gxs[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 tacc[NT];
                acc_zero(tacc);
                if (c == 0) mm_pe<RgbLds::LD0, NT, 0, 4>(L.v0e, d1 + 0, tacc);
                if (c == 1) mm_pe<RgbLds::LD0, NT, 4, 4>(L.v0e, d1 + 4, tacc);
                if (c == 2) mm_pe<RgbLds::LD0, NT, 8, 4>(L.v0e, d1 + 8, tacc);
                float dsum = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < ACT_STEPS; ++s2) dsum = __builtin_fmaf(gyv[s2], tacc[s2 >> 2][s2 & 3], dsum);
                gxs[c] = group_sum(dsum);
            }
            if (g == 0) { a.g_points[pt * 3 + 0] = gxs[0]; a.g_points[pt * 3 + 1] = gxs[1]; a.g_points[pt * 3 + 2] = gxs[2]; }
        }
    }
    const float gb = wave_sum(gbeta_acc);
    if (lane == 0) a.g_beta[blockIdx.x * 4 + wave] = gb * dbeta_dbp;
    if (blockIdx.x == 0 && wave == 0)               // the waves of the blocks that were not launched (small renders)
        for (int e = gridDim.x * 4 + lane; e < 2048; e += 64) a.g_beta[e] = 0.f;
    if (a.v3_part) {
        // sum over the 16 point columns of a lane group (xor shuffles stay inside the group of 16), then lane p == 0 of group g writes
        // its 16 channels of every row: a fixed order, per wave
        float* dst = a.v3_part + (size_t)(blockIdx.x * 4 + wave) * 196;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                float v = v3acc[j][s2];
                for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                if (p == 0) dst[j * 64 + kp(s2) + 4 * g] = v;
            }
            float bsum = b3acc[j];
            for (int d = 8; d >= 1; d >>= 1) bsum += __shfl_xor(bsum, d);
            if (lane == 0) dst[192 + j] = bsum;
        }
        if (lane == 0) dst[195] = 0.f;
        if (blockIdx.x == 0 && wave == 0)
            for (int e = gridDim.x * 4 * 196 + lane; e < 2048 * 196; e += 64) a.v3_part[e] = 0.f;
    }
}

}  // namespace sc

extern "C" int sc_rgb_composite_backward_v3(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, float* v3_part, void* stream_) {
    if (n_rays <= 0) return 0;
    sc::RgbBwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat,
                     n_rays, rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow,
                     G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat, g_points, g_z, g_depth_fac, g_beta,
                     gy, rr, gy3, v3_part};
    int blocks = (n_rays + 3) / 4;
    if (blocks > 512) blocks = 512;
    const size_t lds_bytes = sc::RgbLds::TOTAL * sizeof(float);
    hipLaunchKernelGGL(sc::rgb_composite_bwd_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

extern "C" int sc_rgb_composite_backward(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, void* stream_) {
    return sc_rgb_composite_backward_v3(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat, n_rays, rays_per_image,
                                        n_images, symmetric, beta_min, bgcolor, normal_pow, G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat,
                                        g_points, g_z, g_depth_fac, g_beta, gy, rr, gy3, nullptr, stream_);
}
