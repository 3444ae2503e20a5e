// render.hip -- ray sampling kernels and single-call orchestration of one render (C ABI level).
//
//   sc_ray_sample_forward / _backward : UniformSampler.get_z_vals + point generation
//                                       (model/renderer.py:13-37, :84-86) and their adjoints
//   sc_render_forward                 : sample -> sdf_fwd -> rgb_composite_fwd, i.e. Renderer.forward
//                                       (model/renderer.py:57-152) for a host that does not want to
//                                       sequence the kernels itself (evaluation / visualisation path)
//   sc_sdf_grid_forward               : compute_level_grid (utils/eval_3D.py:9-38), grid generated on the fly
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/shapeclipper_hip.h"

namespace sc {

// torch.linspace(0, 1, 64) in fp32 as ATen evaluates it: symmetric from both ends, the upper half as one fused
// multiply-add end - step*(steps-1-i) (verified bit-for-bit against torch.linspace on the host)
__device__ __forceinline__ float linspace01(int i, int steps) {
    const float step = 1.0f / (float)(steps - 1);
    return i < steps / 2 ? __fmul_rn(step, (float)i) : __fmaf_rn(-step, (float)(steps - 1 - i), 1.0f);
}

// lane == sample (64 samples per ray, one wave per ray)
__global__ __launch_bounds__(256) void ray_sample_kernel(const float* __restrict__ cam_loc, const float* __restrict__ ray_dirs,
                                                         const float* __restrict__ scale_dist, const float* __restrict__ u,
                                                         int n_rays, int rays_per_image, int n_images, float cam_dist,
                                                         float* __restrict__ z_vals, float* __restrict__ points,
                                                         const long long* __restrict__ eik_idx, const float* __restrict__ eik_uniform,
                                                         float* __restrict__ eik_points) {
#pragma clang fp contract(off)   // keep torch's separate mul / add roundings (HIP's __fmul_rn is a plain '*' and would be fused)
    const int lane = threadIdx.x & 63;
    for (int ray = blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += gridDim.x * 4) {
        const int img = min(ray / rays_per_image, n_images - 1);
        const float c = __fmul_rn(cam_dist, scale_dist[img]);
        const float nearv = __fsub_rn(c, 0.7f), farv = __fadd_rn(c, 0.7f);
        const float t = linspace01(lane, 64);
        float z = __fadd_rn(__fmul_rn(nearv, __fsub_rn(1.0f, t)), __fmul_rn(farv, t));
        if (u) {   // stratified jitter inside [lower, upper] (renderer.py:24-30)
            const float zn = __shfl_down(z, 1), zp = __shfl_up(z, 1);
            const float upper = lane == 63 ? z : __fmul_rn(0.5f, __fadd_rn(zn, z));
            const float lower = lane == 0 ? z : __fmul_rn(0.5f, __fadd_rn(z, zp));
            z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u[(size_t)ray * 64 + lane]));
        }
        z_vals[(size_t)ray * 64 + lane] = z;
        const size_t p = (size_t)ray * 64 + lane;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = __fadd_rn(cam_loc[(size_t)ray * 3 + k], __fmul_rn(z, ray_dirs[(size_t)ray * 3 + k]));
            points[p * 3 + k] = v;
            // eikonal points of the image (renderer.py:154-165): [uniform block R | near-surface block R]; the near point of a ray IS its
            // sample eik_idx -- cam_loc + z_eik * ray_dir, the same two roundings -- so it is copied, not re-derived from a gather of z
            if (eik_points && lane == (int)eik_idx[ray]) eik_points[((size_t)(img * 2 + 1) * rays_per_image + (ray - img * rays_per_image)) * 3 + k] = v;
        }
        if (eik_points && lane < 3)
            eik_points[((size_t)(img * 2) * rays_per_image + (ray - img * rays_per_image)) * 3 + lane] = eik_uniform[(size_t)ray * 3 + lane];
    }
}

// adjoint: g_points [P][3], g_z_extra [n_rays][64] (from compositing, may be null)
//   -> g_cam_loc [n_rays][3], g_ray_dirs [n_rays][3], g_scale_dist_ray [n_rays] (sum over the rays of an image = d/d scale_dist)
__global__ __launch_bounds__(256) void ray_sample_bwd_kernel(const float* __restrict__ ray_dirs, const float* __restrict__ z_vals,
                                                             const float* __restrict__ g_points, const float* __restrict__ g_z_extra,
                                                             int n_rays, int rays_per_image, int n_images, float cam_dist,
                                                             float* __restrict__ g_cam_loc, float* __restrict__ g_ray_dirs,
                                                             float* __restrict__ g_scale_dist, const long long* __restrict__ eik_idx,
                                                             const float* __restrict__ g_eik_points) {
    const int lane = threadIdx.x & 63;
    for (int ray = blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += gridDim.x * 4) {
        const size_t p = (size_t)ray * 64 + lane;
        const float z = z_vals[p];
        float g0 = g_points ? g_points[p * 3] : 0.f, g1 = g_points ? g_points[p * 3 + 1] : 0.f, g2 = g_points ? g_points[p * 3 + 2] : 0.f;
        if (g_eik_points && lane == (int)eik_idx[ray]) {        // the near-surface eikonal point of this ray is this sample
            const int img = min(ray / rays_per_image, n_images - 1);
            const float* ge = g_eik_points + ((size_t)(img * 2 + 1) * rays_per_image + (ray - img * rays_per_image)) * 3;
            g0 += ge[0]; g1 += ge[1]; g2 += ge[2];
        }
        const float d0 = ray_dirs[(size_t)ray * 3], d1 = ray_dirs[(size_t)ray * 3 + 1], d2 = ray_dirs[(size_t)ray * 3 + 2];
        float gz = g0 * d0 + g1 * d1 + g2 * d2 + (g_z_extra ? g_z_extra[p] : 0.f);
        float s0 = g0, s1 = g1, s2 = g2, t0 = z * g0, t1 = z * g1, t2 = z * g2;
        for (int o = 32; o >= 1; o >>= 1) {
            s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
            t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o);
            gz += __shfl_xor(gz, o);
        }
        if (lane == 0) {
            g_cam_loc[(size_t)ray * 3] = s0; g_cam_loc[(size_t)ray * 3 + 1] = s1; g_cam_loc[(size_t)ray * 3 + 2] = s2;
            g_ray_dirs[(size_t)ray * 3] = t0; g_ray_dirs[(size_t)ray * 3 + 1] = t1; g_ray_dirs[(size_t)ray * 3 + 2] = t2;
            // every z of the ray shifts by cam_dist * d(scale_dist): near and far move together.  Per-ray value;
            // the caller sums the rays of an image (32 addresses would serialise 16K atomics).
            g_scale_dist[ray] = cam_dist * gz;
        }
    }
}

__global__ __launch_bounds__(256) void grid_points_kernel(float lo, float hi, int n_axis, int n_images, float* __restrict__ points) {
    // torch.linspace(lo, hi, n_axis) with 'ij' meshgrid, repeated per image (eval_3D.py:9-18)
    const size_t per = (size_t)n_axis * n_axis * n_axis, total = per * n_images;
    const float step = (hi - lo) / (float)(n_axis - 1);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t r = idx % per;
        const int k = (int)(r % n_axis), j = (int)((r / n_axis) % n_axis), i = (int)(r / ((size_t)n_axis * n_axis));
        auto lin = [&](int a) { return a < n_axis / 2 ? __fmaf_rn(step, (float)a, lo) : __fmaf_rn(-step, (float)(n_axis - 1 - a), hi); };
        points[idx * 3] = lin(i); points[idx * 3 + 1] = lin(j); points[idx * 3 + 2] = lin(k);
    }
}

__global__ __launch_bounds__(256) void scale4_kernel(const float* __restrict__ G, float* g_rgb, size_t n_rgb, float* g_mask, size_t n_mask,
                                                     float* g_normal, size_t n_normal, float* g_eik, size_t n_eik, float* g_normal_t) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g_normal_t && i < n_normal) g_normal_t[i] *= G[2];
    if (i < n_rgb) g_rgb[i] *= G[0];
    if (i < n_mask) g_mask[i] *= G[1];
    if (i < n_normal) g_normal[i] *= G[2];
    if (g_eik && i < n_eik) g_eik[i] *= G[3];
}

}  // namespace sc

extern "C" {

int sc_ray_sample_forward(const float* cam_loc, const float* ray_dirs, const float* scale_dist, const float* u, int n_rays,
                          int rays_per_image, int n_images, float cam_dist, float* z_vals, float* points, void* stream_) {
    if (n_rays <= 0) return 0;
    int blocks = (n_rays + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sc::ray_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, cam_loc, ray_dirs, scale_dist, u,
                       n_rays, rays_per_image, n_images, cam_dist, z_vals, points, (const long long*)nullptr, (const float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

int sc_ray_sample_forward_eik(const float* cam_loc, const float* ray_dirs, const float* scale_dist, const float* u, const long long* eik_idx,
                              const float* eik_uniform, int n_rays, int rays_per_image, int n_images, float cam_dist, float* z_vals,
                              float* points, float* eik_points, void* stream_) {
    if (n_rays <= 0) return 0;
    if (!eik_idx || !eik_uniform || !eik_points || n_rays != rays_per_image * n_images) return (int)hipErrorInvalidValue;
    int blocks = (n_rays + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sc::ray_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, cam_loc, ray_dirs, scale_dist, u,
                       n_rays, rays_per_image, n_images, cam_dist, z_vals, points, eik_idx, eik_uniform, eik_points);
    return (int)hipGetLastError();
}

int sc_ray_sample_backward_eik(const float* ray_dirs, const float* z_vals, const float* g_points, const float* g_z_extra, const long long* eik_idx,
                               const float* g_eik_points, int n_rays, int rays_per_image, int n_images, float cam_dist, float* g_cam_loc,
                               float* g_ray_dirs, float* g_scale_dist, void* stream_) {
    if (n_rays <= 0) return 0;
    if (g_eik_points && (!eik_idx || n_rays != rays_per_image * n_images)) return (int)hipErrorInvalidValue;
    int blocks = (n_rays + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sc::ray_sample_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, ray_dirs, z_vals, g_points,
                       g_z_extra, n_rays, rays_per_image, n_images, cam_dist, g_cam_loc, g_ray_dirs, g_scale_dist, eik_idx, g_eik_points);
    return (int)hipGetLastError();
}

int sc_ray_sample_backward(const float* ray_dirs, const float* z_vals, const float* g_points, const float* g_z_extra, int n_rays,
                           int rays_per_image, int n_images, float cam_dist, float* g_cam_loc, float* g_ray_dirs,
                           float* g_scale_dist, void* stream_) {
    if (n_rays <= 0) return 0;
    int blocks = (n_rays + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sc::ray_sample_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, ray_dirs, z_vals, g_points,
                       g_z_extra, n_rays, rays_per_image, n_images, cam_dist, g_cam_loc, g_ray_dirs, g_scale_dist, (const long long*)nullptr,
                       (const float*)nullptr);
    return (int)hipGetLastError();
}

int sc_render_forward(const float* cam_loc, const float* ray_dirs, const float* depth_fac, const float* scale_dist,
                      const float* u, const float* sdf_pack, const float* sdf_cbias, const float* rgb_pack,
                      const float* rgb_dbias, const float* beta_param, int n_rays, int rays_per_image, int n_images,
                      int symmetric, float cam_dist, float beta_min, float bgcolor, float normal_pow,
                      float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                      float* z_vals, float* points, float* sdf, float* grad, float* feat, float* scratch,
                      float* stash_a, float* stash_p, float* rgb_flat, void* stream) {
    int rc = sc_ray_sample_forward(cam_loc, ray_dirs, scale_dist, u, n_rays, rays_per_image, n_images, cam_dist, z_vals, points, stream);
    if (rc) return rc;
    rc = sc_sdf_forward(points, sdf_pack, sdf_cbias, n_rays * 64, rays_per_image * 64, n_images, symmetric, sdf, grad, feat,
                        stash_a, stash_p, scratch, stream);
    if (rc) return rc;
    return sc_rgb_composite_forward(points, z_vals, depth_fac, sdf, grad, feat, rgb_pack, rgb_dbias, beta_param, n_rays,
                                    rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow, rgb, mask, mask_hard,
                                    depth, normal, nullptr, nullptr, rgb_flat, stream);
}

int sc_sdf_grid_forward(const float* sdf_pack, const float* sdf_cbias, float lo, float hi, int n_axis, int n_images,
                        int symmetric, float* points_ws, float* level, void* stream_) {
    const size_t total = (size_t)n_axis * n_axis * n_axis * n_images;
    if (total == 0) return 0;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sc::grid_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, lo, hi, n_axis, n_images, points_ws);
    return sc_sdf_forward(points_ws, sdf_pack, sdf_cbias, (int)total, n_axis * n_axis * n_axis, n_images, symmetric, level,
                          nullptr, nullptr, nullptr, nullptr, nullptr, stream_);
}

// compute_level_grid in the exact three-piece bf16 split arithmetic with pre-split weights (csrc/sdf_value_split.hip, round 6): the same
// grid points (grid_points_kernel), the value-only chain.  ops.sdf_grid_forward's default since the round-6 A/B.
int sc_sdf_grid_forward_split(const float* sdf_pack, const float* sdf_cbias, float lo, float hi, int n_axis, int n_images,
                              int symmetric, float* points_ws, float* level, void* stream_) {
    const size_t total = (size_t)n_axis * n_axis * n_axis * n_images;
    if (total == 0) return 0;
    if (total > 0x7fffffffULL) return (int)hipErrorInvalidValue;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sc::grid_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, lo, hi, n_axis, n_images, points_ws);
    return sc_sdf_value_forward_split(points_ws, sdf_pack, sdf_cbias, (int)total, n_axis * n_axis * n_axis, n_images, symmetric, level, stream_);
}

int sc_loss_fused_backward(const float* G4, float* g_rgb, long long n_rgb, float* g_mask, long long n_mask, float* g_normal,
                           long long n_normal, float* g_eik, long long n_eik, float* g_normal_t, void* stream_) {
    long long n = n_rgb > n_normal ? n_rgb : n_normal;
    if (n_mask > n) n = n_mask;
    if (n_eik > n) n = n_eik;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sc::scale4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, G4, g_rgb,
                       (size_t)n_rgb, g_mask, (size_t)n_mask, g_normal, (size_t)n_normal, g_eik, (size_t)n_eik, g_normal_t);
    return (int)hipGetLastError();
}

}  // extern "C"
