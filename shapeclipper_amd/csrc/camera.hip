// camera.hip -- camera algebra of one render in one launch each way.
//
// camera_rays: reference utils/camera.py:157-196 (get_camera_grid, img2cam, cam2world, get_center_and_ray) plus the
// ray normalisation and depth factor of model/renderer.py:69-76.  The reference builds all H*W rays with ~25 torch
// operators on [B,H*W,3] tensors and gathers 512 of them; the torch form of this build still needed 65 tiny launches
// forward and ~190 backward per render.  Here: one thread per rendered pixel,
//     p = (x+.5, y+.5, 1),  g = K^-1 p,  grid_w = R^T g + t_inv,  centre_w = t_inv = -R^T t,  ray = grid_w - centre_w,
//     dir = ray / max(|ray|, 1e-12),  depth_fac = |dir| / |ray|
// (same operation order as the reference), and a hand-derived adjoint that returns d/d pose [B,3,4] and d/d intr [B,3,3]
// with the per-image sums reduced inside the block (one block per image).
//
// pose_from_trig: reference model/graph.py:272-293 + utils/camera.py:105-155,198-211: the viewpoint estimator's
// (cos, sin) pairs -> R = Rz(theta) Rx(elev) Ry(azim) P, pose = [R | (0, 0, dist*scale_dist)], K = diag(f W, f H) with
// principal point (W/2, H/2), f = focal * scale_focal.  One thread per image, adjoint in closed form.
// Bound: latency (KBs of data); what is bought is ~600 fewer launches per training step.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

struct Mat3 {
    float m[3][3];
};

__device__ __forceinline__ Mat3 inverse3(const float* K, float& det) {
    const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const float A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    det = a * A + b * Bc + c * C;
    Mat3 r;
    r.m[0][0] = A / det;  r.m[0][1] = -(b * i - c * h) / det; r.m[0][2] = (b * f - c * e) / det;
    r.m[1][0] = Bc / det; r.m[1][1] = (a * i - c * g) / det;  r.m[1][2] = -(a * f - c * d) / det;
    r.m[2][0] = C / det;  r.m[2][1] = -(a * h - b * g) / det; r.m[2][2] = (a * e - b * d) / det;
    return r;
}

struct RayGeom {       // everything the forward and the adjoint share for one pixel
    float p[3], gc[3], ray[3], n;
};

__device__ __forceinline__ RayGeom ray_geom(const float* pose, const Mat3& Kinv, const float* tinv, long long pix, int W) {
    RayGeom q;
    q.p[0] = (float)(pix % W) + 0.5f;
    q.p[1] = (float)(pix / W) + 0.5f;
    q.p[2] = 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) q.gc[i] = Kinv.m[i][0] * q.p[0] + Kinv.m[i][1] * q.p[1] + Kinv.m[i][2] * q.p[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {          // grid_w_j = sum_i R[i][j] g_i + t_inv_j ;  ray = grid_w - centre_w
        const float gw = pose[0 * 4 + j] * q.gc[0] + pose[1 * 4 + j] * q.gc[1] + pose[2 * 4 + j] * q.gc[2] + tinv[j];
        q.ray[j] = gw - tinv[j];
    }
    q.n = sqrtf(q.ray[0] * q.ray[0] + q.ray[1] * q.ray[1] + q.ray[2] * q.ray[2]);
    return q;
}

__global__ __launch_bounds__(256) void camera_rays_fwd_kernel(const float* __restrict__ pose_all, const float* __restrict__ intr_all,
                                                              const long long* __restrict__ ray_idx, int R, int W,
                                                              float* __restrict__ cam_loc, float* __restrict__ ray_dirs,
                                                              float* __restrict__ depth_fac) {
    const int b = blockIdx.x;
    const float* pose = pose_all + (size_t)b * 12;
    float det;
    const Mat3 Kinv = inverse3(intr_all + (size_t)b * 9, det);
    float tinv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) tinv[j] = -(pose[0 * 4 + j] * pose[3] + pose[1 * 4 + j] * pose[7] + pose[2 * 4 + j] * pose[11]);
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const long long pix = ray_idx ? ray_idx[(size_t)b * R + r] : r;
        const RayGeom q = ray_geom(pose, Kinv, tinv, pix, W);
        const float inv = 1.f / fmaxf(q.n, 1e-12f);
        const float d0 = q.ray[0] * inv, d1 = q.ray[1] * inv, d2 = q.ray[2] * inv;
        const size_t o = ((size_t)b * R + r) * 3;
        ray_dirs[o] = d0; ray_dirs[o + 1] = d1; ray_dirs[o + 2] = d2;
        cam_loc[o] = tinv[0]; cam_loc[o + 1] = tinv[1]; cam_loc[o + 2] = tinv[2];
        depth_fac[(size_t)b * R + r] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2) / q.n;
    }
}

__global__ __launch_bounds__(256) void camera_rays_bwd_kernel(const float* __restrict__ pose_all, const float* __restrict__ intr_all,
                                                              const long long* __restrict__ ray_idx, int R, int W,
                                                              const float* __restrict__ g_cam_loc, const float* __restrict__ g_dirs,
                                                              const float* __restrict__ g_depth_fac,
                                                              float* __restrict__ g_pose, float* __restrict__ g_intr) {
    __shared__ float red[21][4];
    const int b = blockIdx.x;
    const float* pose = pose_all + (size_t)b * 12;
    float det;
    const Mat3 Kinv = inverse3(intr_all + (size_t)b * 9, det);
    float tinv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) tinv[j] = -(pose[0 * 4 + j] * pose[3] + pose[1 * 4 + j] * pose[7] + pose[2 * 4 + j] * pose[11]);
    // per-thread partial sums: GR[i][j] (9), GKinv[i][j] (9), Gtinv[j] (3)
    float acc[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const long long pix = ray_idx ? ray_idx[(size_t)b * R + r] : r;
        const RayGeom q = ray_geom(pose, Kinv, tinv, pix, W);
        const size_t o = ((size_t)b * R + r) * 3;
        const float inv = 1.f / fmaxf(q.n, 1e-12f);
        const float d[3] = {q.ray[0] * inv, q.ray[1] * inv, q.ray[2] * inv};
        float gd[3] = {0.f, 0.f, 0.f};
        if (g_dirs) { gd[0] = g_dirs[o]; gd[1] = g_dirs[o + 1]; gd[2] = g_dirs[o + 2]; }
        const float gdf = g_depth_fac ? g_depth_fac[(size_t)b * R + r] : 0.f;
        const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
        float gray[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gray[j] = (gd[j] - d[j] * dot) * inv - gdf * d[j] * inv * inv;   // d(1/n)/d ray = -d/n^2
        float ggc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) ggc[i] = pose[i * 4 + 0] * gray[0] + pose[i * 4 + 1] * gray[1] + pose[i * 4 + 2] * gray[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[i * 3 + j] += q.gc[i] * gray[j];          // ray_j = sum_i R_ij g_i
                acc[9 + i * 3 + j] += ggc[i] * q.p[j];        // g_i = sum_j Kinv_ij p_j
            }
        if (g_cam_loc) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[18 + j] += g_cam_loc[o + j];
        }
    }
    // block reduction of the 21 sums (fixed order: lanes, then the 4 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        float v = acc[k];
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S[21];
        for (int k = 0; k < 21; ++k) S[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        const float* GR = S;            // from the rays
        const float* GKi = S + 9;
        const float* Gt = S + 18;       // d/d t_inv
        float* gp = g_pose + (size_t)b * 12;
        // t_inv_j = -sum_i R_ij t_i  ->  dR_ij += -t_i Gt_j ;  dt_i = -sum_j R_ij Gt_j
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) gp[i * 4 + j] = GR[i * 3 + j] - pose[i * 4 + 3] * Gt[j];
            gp[i * 4 + 3] = -(pose[i * 4 + 0] * Gt[0] + pose[i * 4 + 1] * Gt[1] + pose[i * 4 + 2] * Gt[2]);
        }
        // d(K^-1) -> dK = -K^-T G K^-T
        float T[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T[i][j] = Kinv.m[0][i] * GKi[0 * 3 + j] + Kinv.m[1][i] * GKi[1 * 3 + j] + Kinv.m[2][i] * GKi[2 * 3 + j];
        float* gk = g_intr + (size_t)b * 9;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) gk[i * 3 + j] = -(T[i][0] * Kinv.m[j][0] + T[i][1] * Kinv.m[j][1] + T[i][2] * Kinv.m[j][2]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// R = Rz Rx Ry P with (utils/camera.py:105-155)
//   Ry = [[ca,0,sa],[0,1,0],[-sa,0,ca]],  Rx = [[1,0,0],[0,ce,-se],[0,se,ce]],  Rz = [[ct,st,0],[-st,ct,0],[0,0,1]],
//   P  = [[-1,0,0],[0,0,-1],[0,-1,0]]  (graph.py:278-283)
__device__ __forceinline__ void mul3(const float (&A)[3][3], const float (&B)[3][3], float (&C)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}

struct TrigIn {
    const float* azim; const float* elev; const float* theta;    // [B][2] (cos, sin)
    const float* scale_focal; const float* scale_dist;            // [B]
    int B; float cam_dist, focal, W, H;
};

__device__ __forceinline__ void trig_mats(const TrigIn& a, int b, float (&Ry)[3][3], float (&Rx)[3][3], float (&Rz)[3][3]) {
    const float ca = a.azim[2 * b], sa = a.azim[2 * b + 1], ce = a.elev[2 * b], se = a.elev[2 * b + 1];
    const float ct = a.theta[2 * b], st = a.theta[2 * b + 1];
    const float ry[3][3] = {{ca, 0.f, sa}, {0.f, 1.f, 0.f}, {-sa, 0.f, ca}};
    const float rx[3][3] = {{1.f, 0.f, 0.f}, {0.f, ce, -se}, {0.f, se, ce}};
    const float rz[3][3] = {{ct, st, 0.f}, {-st, ct, 0.f}, {0.f, 0.f, 1.f}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Ry[i][j] = ry[i][j]; Rx[i][j] = rx[i][j]; Rz[i][j] = rz[i][j]; }
}

__global__ void pose_from_trig_fwd_kernel(TrigIn a, float* __restrict__ pose, float* __restrict__ intr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float Ry[3][3], Rx[3][3], Rz[3][3], A[3][3], M[3][3];
    trig_mats(a, b, Ry, Rx, Rz);
    mul3(Rz, Rx, A);
    mul3(A, Ry, M);
    // (M P): columns of M permuted / negated: P = [[-1,0,0],[0,0,-1],[0,-1,0]] -> col0 = -M col0, col1 = -M col2, col2 = -M col1
    float* p = pose + (size_t)b * 12;
    for (int i = 0; i < 3; ++i) {
        p[i * 4 + 0] = -M[i][0];
        p[i * 4 + 1] = -M[i][2];
        p[i * 4 + 2] = -M[i][1];
        p[i * 4 + 3] = i == 2 ? a.scale_dist[b] * a.cam_dist : 0.f;
    }
    const float f = a.focal * a.scale_focal[b];
    float* k = intr + (size_t)b * 9;
    k[0] = f * a.W; k[1] = 0.f; k[2] = a.W / 2;
    k[3] = 0.f; k[4] = f * a.H; k[5] = a.H / 2;
    k[6] = 0.f; k[7] = 0.f; k[8] = 1.f;
}

__global__ void pose_from_trig_bwd_kernel(TrigIn a, const float* __restrict__ g_pose, const float* __restrict__ g_intr,
                                          float* __restrict__ g_azim, float* __restrict__ g_elev, float* __restrict__ g_theta,
                                          float* __restrict__ g_scale_focal, float* __restrict__ g_scale_dist) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float Ry[3][3], Rx[3][3], Rz[3][3];
    trig_mats(a, b, Ry, Rx, Rz);
    // G_M = G_R P^T : undo the column permutation
    float GM[3][3];
    const float* gp = g_pose + (size_t)b * 12;
    for (int i = 0; i < 3; ++i) {
        GM[i][0] = -gp[i * 4 + 0];
        GM[i][2] = -gp[i * 4 + 1];
        GM[i][1] = -gp[i * 4 + 2];
    }
    // M = Rz Rx Ry:  G_Ry = (Rz Rx)^T G_M ;  G_Rx = Rz^T G_M Ry^T ;  G_Rz = G_M (Rx Ry)^T
    float A[3][3], Bm[3][3], GRy[3][3], GRx[3][3], GRz[3][3], T[3][3];
    mul3(Rz, Rx, A);
    mul3(Rx, Ry, Bm);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            GRy[i][j] = A[0][i] * GM[0][j] + A[1][i] * GM[1][j] + A[2][i] * GM[2][j];
            T[i][j] = Rz[0][i] * GM[0][j] + Rz[1][i] * GM[1][j] + Rz[2][i] * GM[2][j];
            GRz[i][j] = GM[i][0] * Bm[j][0] + GM[i][1] * Bm[j][1] + GM[i][2] * Bm[j][2];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) GRx[i][j] = T[i][0] * Ry[j][0] + T[i][1] * Ry[j][1] + T[i][2] * Ry[j][2];
    g_azim[2 * b] = GRy[0][0] + GRy[2][2];
    g_azim[2 * b + 1] = GRy[0][2] - GRy[2][0];
    g_elev[2 * b] = GRx[1][1] + GRx[2][2];
    g_elev[2 * b + 1] = GRx[2][1] - GRx[1][2];
    g_theta[2 * b] = GRz[0][0] + GRz[1][1];
    g_theta[2 * b + 1] = GRz[0][1] - GRz[1][0];
    g_scale_dist[b] = gp[2 * 4 + 3] * a.cam_dist;
    const float* gk = g_intr + (size_t)b * 9;
    g_scale_focal[b] = a.focal * (gk[0] * a.W + gk[4] * a.H);
}

}  // namespace sc

extern "C" int sc_camera_rays_forward(const float* pose, const float* intr, const long long* ray_idx, int n_images,
                                      int rays_per_image, int image_width, float* cam_loc, float* ray_dirs,
                                      float* depth_fac, void* stream_) {
    if (n_images <= 0 || rays_per_image <= 0) return 0;
    hipLaunchKernelGGL(sc::camera_rays_fwd_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream_, pose, intr, ray_idx,
                       rays_per_image, image_width, cam_loc, ray_dirs, depth_fac);
    return (int)hipGetLastError();
}

extern "C" int sc_camera_rays_backward(const float* pose, const float* intr, const long long* ray_idx, int n_images,
                                       int rays_per_image, int image_width, const float* g_cam_loc, const float* g_ray_dirs,
                                       const float* g_depth_fac, float* g_pose, float* g_intr, void* stream_) {
    if (n_images <= 0 || rays_per_image <= 0) return 0;
    hipLaunchKernelGGL(sc::camera_rays_bwd_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream_, pose, intr, ray_idx,
                       rays_per_image, image_width, g_cam_loc, g_ray_dirs, g_depth_fac, g_pose, g_intr);
    return (int)hipGetLastError();
}

extern "C" int sc_pose_from_trig_forward(const float* azim, const float* elev, const float* theta, const float* scale_focal,
                                         const float* scale_dist, int n_images, float cam_dist, float focal, int image_width,
                                         int image_height, float* pose, float* intr, void* stream_) {
    if (n_images <= 0) return 0;
    sc::TrigIn a{azim, elev, theta, scale_focal, scale_dist, n_images, cam_dist, focal, (float)image_width, (float)image_height};
    hipLaunchKernelGGL(sc::pose_from_trig_fwd_kernel, dim3((n_images + 63) / 64), dim3(64), 0, (hipStream_t)stream_, a, pose, intr);
    return (int)hipGetLastError();
}

extern "C" int sc_pose_from_trig_backward(const float* azim, const float* elev, const float* theta, const float* scale_focal,
                                          const float* scale_dist, int n_images, float cam_dist, float focal, int image_width,
                                          int image_height, const float* g_pose, const float* g_intr, float* g_azim,
                                          float* g_elev, float* g_theta, float* g_scale_focal, float* g_scale_dist,
                                          void* stream_) {
    if (n_images <= 0) return 0;
    sc::TrigIn a{azim, elev, theta, scale_focal, scale_dist, n_images, cam_dist, focal, (float)image_width, (float)image_height};
    hipLaunchKernelGGL(sc::pose_from_trig_bwd_kernel, dim3((n_images + 63) / 64), dim3(64), 0, (hipStream_t)stream_, a, g_pose,
                       g_intr, g_azim, g_elev, g_theta, g_scale_focal, g_scale_dist);
    return (int)hipGetLastError();
}
