// render_bwd.hip -- the whole reverse pass of one training render behind ONE C entry point (SURVEY 8b:
// sc_render_backward), the counterpart of sc_render_forward.
//
// What autograd does in the reference for Renderer.forward (model/renderer.py:57-185) and the two MLPs it calls
// (model/implicit.py:138-239), including the double backward through d(sdf)/dx, is here the fixed sequence
//   sc_rgb_composite_backward                    per-ray upstream gradients -> per-point gradients + RGB-net operands
//   sc_wgrad x4, sc_partial_reduce, sc_tbl_sum   RGB weight / per-image bias gradients
//   sc_sdf_backward_fused, sc_partial_reduce     first- and second-order input gradients + SDF weight / per-image bias
//                                                gradients in one workgroup-cooperative launch (csrc/sdf_bwdw.hip)
//   sc_ray_sample_backward                       d/d camera centre, ray direction, scale_dist (per ray)
// on caller-provided workspace.  The host-side Python of this build issues the same sequence step by step
// (shapeclipper_amd/ops.py); this entry point is the same thing for a non-Python host.
#include "mlp_tile.hpp"
#include "shapeclipper_hip.h"

namespace sc {

enum { W_OP_NONE = 0, W_OP_PLAIN = 1, W_OP_SP = 2, W_OP_Q = 3, W_OP_Q4 = 4, W_OP_PE = 5, W_OP_EPS = 6 };
constexpr int RB_PARTS = 512;
constexpr int RB_MAX_IMAGES = 64;      // the fixed-order partial images of the per-image sums are carved for up to this many images

__global__ __launch_bounds__(256) void rb_add_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += b[i];
}

// part[block][k] = sum over the block's rows of x[i*stride + k], k < K <= 4 (every block writes its 4 floats; sc_partial_reduce adds the
// blocks in index order: fixed summation order); used for the column sums of gy3 (K = 3)
__global__ __launch_bounds__(256) void rb_colsum_kernel(const float* __restrict__ x, size_t n, int stride, int K,
                                                        float* __restrict__ part) {
    __shared__ float red[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        for (int k = 0; k < K; ++k) s[k] += x[i * stride + k];
    for (int k = 0; k < K; ++k) {
        float v = s[k];
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[blockIdx.x * 4 + threadIdx.x] = threadIdx.x < K ? (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]) : 0.f;
}

// [n_images][5][64] -> [5][n_images][64]
__global__ __launch_bounds__(256) void rb_transpose_cbias_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_images) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 5 * n_images * 64) return;
    const int ch = idx & 63, l = (idx >> 6) % 5, img = idx / 320;
    dst[((size_t)l * n_images + img) * 64 + ch] = src[idx];
}

struct Carver {
    char* base; size_t off;
    float* f(size_t n_floats) { char* p = base + off; off += (n_floats * 4 + 255) & ~(size_t)255; return reinterpret_cast<float*>(p); }
};

static size_t rb_workspace_floats(int n_rays, int* T_out) {
    const size_t P = (size_t)n_rays * 64, T = (size_t)((P + TP - 1) / TP) * 1024;
    if (T_out) *T_out = (int)0;
    size_t n = 0;
    auto add = [&](size_t k) { n += (k + 63) & ~(size_t)63; };
    add(P); add(3 * P); add(T); add(3 * P); add(3 * P);            // g_sdf, g_grad, g_feat, g_points (rgb), g_points (sdf)
    add((size_t)n_rays * 64);                                       // g_z
    add(3 * T); add(3 * T); add(3 * P);                            // gy, rr, gy3
    add(5 * T > (size_t)256 * 4 * 4 * 1024 ? 5 * T : (size_t)256 * 4 * 4 * 1024);   // park scratch of the fused SDF backward (was Ga)
    add(4 * T > (size_t)5 * 4096 * 64 ? 4 * T : (size_t)5 * 4096 * 64);             // per-image bias-gradient staging (was Gp)
    add(T);                                                                          // (unused, kept for layout stability)
    add((size_t)RB_PARTS * SdfPack::TOTAL);                        // partial images of the RGB weight gradients
    add(2 * 64);                                                   // (unused, kept for layout stability)
    add((size_t)RB_PARTS * 4 * RB_MAX_IMAGES * 64);                // per-wave row-sum partials of sc_wgrad
    add((size_t)256 * (SdfPack::TOTAL + RB_MAX_IMAGES * 320));     // partial images of the fused SDF backward (weights + per-image biases)
    add(SC_RGB_BWD_BETA_PARTS); add(512 * 4); add((size_t)1024 * 3 * 64);    // beta partials, column-sum partials, tbl_sum partials
    return n;
}

static int wgrad1(const float* a0, const float* b0, int bop0, const float* points, int n, int symmetric, int nb0, float* partial,
                  int stride, int off, int ld, float* rowsum, int npi, int nimg, void* st) {
    return sc_wgrad(1, a0, nullptr, W_OP_PLAIN, b0, bop0, nullptr, W_OP_NONE, nullptr, nullptr, W_OP_NONE, nullptr, W_OP_NONE, nullptr,
                    W_OP_NONE, points, nullptr, nullptr, n, symmetric, nb0, 0, partial, RB_PARTS, stride, off, ld, rowsum, npi, nimg, st);
}

}  // namespace sc

extern "C" long long sc_render_backward_workspace_bytes(int n_rays) {
    return (long long)(sc::rb_workspace_floats(n_rays, nullptr) * 4 + 64 * 256);
}

extern "C" int sc_render_backward(
    const float* ray_dirs, const float* depth_fac, const float* sdf_pack, const float* rgb_pack, const float* rgb_dbias,
    const float* beta_param, const float* z_vals, const float* points, const float* sdf, const float* grad, const float* feat,
    const float* stash_a, const float* stash_p, const float* rgb_flat, int n_rays, int rays_per_image, int n_images,
    int symmetric, float cam_dist, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal, const float* G_z_extra,
    float* g_sdf_pack, float* g_cbias, float* g_rgb_pack, float* g_dbias, float* g_beta,
    float* g_cam_loc, float* g_ray_dirs, float* g_scale_dist, float* g_depth_fac,
    void* workspace, long long workspace_bytes, void* stream_) {
    using namespace sc;
    if (n_rays <= 0) return 0;
    if (workspace_bytes < sc_render_backward_workspace_bytes(n_rays)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    const int P = n_rays * 64, npi = rays_per_image * 64;
    const size_t T = (size_t)((P + TP - 1) / TP) * 1024;
    Carver ws{(char*)workspace, 0};
    float* g_sdf = ws.f(P); float* g_grad = ws.f(3 * (size_t)P); float* g_feat = ws.f(T);
    float* gpts_rgb = ws.f(3 * (size_t)P); float* gpts_sdf = ws.f(3 * (size_t)P); float* g_z = ws.f((size_t)n_rays * 64);
    float* gy = ws.f(3 * T); float* rr = ws.f(3 * T); float* gy3 = ws.f(3 * (size_t)P);
    float* ga = ws.f(5 * T > (size_t)256 * 4 * 4 * 1024 ? 5 * T : (size_t)256 * 4 * 4 * 1024);
    float* gp = ws.f(4 * T > (size_t)5 * 4096 * 64 ? 4 * T : (size_t)5 * 4096 * 64);
    float* r0 = ws.f(T);
    (void)r0;
    if (n_images > RB_MAX_IMAGES) return (int)hipErrorInvalidValue;
    float* partial = ws.f((size_t)RB_PARTS * SdfPack::TOTAL);
    float* tot = ws.f(2 * 64);
    (void)tot;
    float* rs_part = ws.f((size_t)RB_PARTS * 4 * RB_MAX_IMAGES * 64);
    float* sdf_partial = ws.f((size_t)256 * (SdfPack::TOTAL + RB_MAX_IMAGES * 320));
    float* beta_part = ws.f(SC_RGB_BWD_BETA_PARTS);
    float* col_part = ws.f(512 * 4);
    float* tbl_part = ws.f((size_t)1024 * 3 * 64);
    const int E = n_images * 64;                        // floats of one per-image row-sum image
    int rc;
#define SC_TRY(x) if ((rc = (x))) return rc
    // ---------------- RGB network + compositing ----------------
    SC_TRY(sc_rgb_composite_backward(points, z_vals, depth_fac, sdf, grad, feat, rgb_pack, rgb_dbias, beta_param, rgb_flat, n_rays,
                                     rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow, G_rgb, G_mask, G_depth,
                                     G_normal, g_sdf, g_grad, g_feat, gpts_rgb, g_z, g_depth_fac, beta_part, gy, rr, gy3, stream_));
    SC_TRY(sc_partial_reduce(beta_part, SC_RGB_BWD_BETA_PARTS, 1, 1, g_beta, stream_));
    hipMemsetAsync(g_dbias, 0, (size_t)3 * n_images * 64 * 4, st);       // [3][n_images][64]
    hipMemsetAsync(g_rgb_pack, 0, (size_t)RgbPack::TOTAL * 4, st);
    const int rs = RgbPack::TOTAL;
    // V0 = [PE 48 | sdf feature 64] in one pass over Gy0
    SC_TRY(sc_wgrad(1, gy, nullptr, W_OP_PLAIN, nullptr, W_OP_PE, feat, W_OP_PLAIN, nullptr, nullptr, W_OP_NONE, nullptr, W_OP_NONE, nullptr,
                    W_OP_NONE, points, nullptr, nullptr, P, symmetric, 48, 64, partial, RB_PARTS, rs, RgbPack::V0, 112, rs_part, npi, n_images,
                    stream_));
    SC_TRY(sc_partial_reduce(rs_part, RB_PARTS * 4, E, E, g_dbias, stream_));                    // per-wave row sums, index order
    SC_TRY(wgrad1(gy + T, rr, W_OP_PLAIN, points, P, symmetric, 64, partial, rs, RgbPack::V1, 64, rs_part, npi, n_images, stream_));
    SC_TRY(sc_partial_reduce(rs_part, RB_PARTS * 4, E, E, g_dbias + (size_t)E, stream_));
    SC_TRY(wgrad1(gy + 2 * T, rr + T, W_OP_PLAIN, points, P, symmetric, 64, partial, rs, RgbPack::V2, 64, rs_part, npi, n_images, stream_));
    SC_TRY(sc_partial_reduce(rs_part, RB_PARTS * 4, E, E, g_dbias + (size_t)2 * E, stream_));
    // the 3-row output layer: V3 via the coefficient form of tbl_sum, its bias via column sums (both accumulate into zeros);
    // the partial images only cover V0..V2 -- reduce exactly that prefix so V3/B3 are not overwritten with garbage
    SC_TRY(sc_partial_reduce(partial, RB_PARTS, rs, RgbPack::V3, g_rgb_pack, stream_));
    {
        const float* xs[1] = {rr + 2 * T};
        float* outs[1] = {g_rgb_pack + RgbPack::V3};
        SC_TRY(sc_tbl_sum(xs, 1, gy3, P, P, 1, outs, tbl_part, stream_));
    }
    hipLaunchKernelGGL(rb_colsum_kernel, dim3(512), dim3(256), 0, st, gy3, (size_t)P, 3, 3, col_part);
    SC_TRY(sc_partial_reduce(col_part, 512, 4, 3, g_rgb_pack + RgbPack::B3, stream_));
    // ---------------- SDF network (first and second order): one workgroup-cooperative launch ----------------
    // (csrc/sdf_bwdw.hip: input gradients + all weight / bias gradients, no Ga/Gp/r0 hand-off tensors; g_cbias comes out as
    //  [n_images][5][64] and is transposed into this entry point's [5][n_images][64] below)
    if (npi % TP != 0) return (int)hipErrorInvalidValue;
    const int parts = sc_sdf_backward_fused_parts(P);
    float* park = ga;                                   // 256 x 4 x 4 KiB x ... : carved from the (now unused) Ga region
    float* g_c_img = gp;                                // [n_images][5][64] staging
    const int S = sc_sdf_backward_fused_partial_floats(n_images);       // weight image + [n_images][5][64] per workgroup (n_images <= 256)
    SC_TRY(sc_sdf_backward_fused(points, sdf_pack, P, npi, n_images, symmetric, stash_a, stash_p, g_sdf, g_grad, g_feat, gpts_sdf,
                                 park, sdf_partial, nullptr, stream_));
    SC_TRY(sc_partial_reduce(sdf_partial, parts, S, SdfPack::TOTAL, g_sdf_pack, stream_));
    SC_TRY(sc_partial_reduce(sdf_partial + SdfPack::TOTAL, parts, S, 5 * E, g_c_img, stream_));
    hipLaunchKernelGGL(rb_transpose_cbias_kernel, dim3((5 * n_images * 64 + 255) / 256), dim3(256), 0, st, g_c_img, g_cbias, n_images);
    // ---------------- points -> camera ----------------
    hipLaunchKernelGGL(rb_add_kernel, dim3(2048), dim3(256), 0, st, gpts_sdf, gpts_rgb, (size_t)3 * P);
    if (G_z_extra) hipLaunchKernelGGL(rb_add_kernel, dim3(1024), dim3(256), 0, st, g_z, G_z_extra, (size_t)n_rays * 64);
    SC_TRY(sc_ray_sample_backward(ray_dirs, z_vals, gpts_sdf, g_z, n_rays, rays_per_image, n_images, cam_dist, g_cam_loc, g_ray_dirs,
                                  g_scale_dist, stream_));
#undef SC_TRY
    return (int)hipGetLastError();
}
