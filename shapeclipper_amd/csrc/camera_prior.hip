// camera_prior.hip -- the [B]-sized arithmetic around the view estimator, one launch each way per block of reference code.
//
//   estimator head   : F.normalize of the three (cos, sin) pairs, tanh / range scaling of the two scale outputs
//                      (reference model/view_estimator.py:62-75)
//   camera priors    : cam_margin_loss + cam_uniform_loss + cam_sym_loss (reference model/loss.py:99-167)
//   transform_normal : camera-frame normals -> canonical frame with the predicted rotation (utils/camera.py:98-103)
//   loss total       : loss.all = sum_k float(w_k) * loss_k, and the NaN/Inf flag of every term (model/runner.py:294-305)
//
// Each of these is a few dozen [B]- or [B,2]-shaped torch operators in the reference (and as many again in its backward);
// at B = 32 they are pure launch latency and host time (~600 of the ~1600 launches of a step).  Nothing here is bound by
// HBM or a pipe: the metric is launches per step.  Sums run in a fixed order (no atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace sc {

// ---------------------------------------------------------------------------------------------------------
// estimator head
struct HeadArgs {
    const float* trig;        // [N][6]   extr_fc output
    const float* size_lin;    // [N]      size_fc output
    const float* persp_lin;   // [N]      perspect_fc output
    int N;
    float size_range, persp_range;
};

__device__ __forceinline__ float l2_clamped(float x, float y) { return fmaxf(sqrtf(x * x + y * y), 1.e-12f); }   // F.normalize eps

__global__ void estimator_head_fwd_kernel(HeadArgs a, float* __restrict__ azim, float* __restrict__ elev, float* __restrict__ theta,
                                          float* __restrict__ scale_focal, float* __restrict__ scale_dist) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    float* outs[3] = {azim, elev, theta};
    for (int k = 0; k < 3; ++k) {
        const float x = a.trig[n * 6 + 2 * k], y = a.trig[n * 6 + 2 * k + 1];
        const float d = l2_clamped(x, y);
        outs[k][2 * n] = x / d;
        outs[k][2 * n + 1] = y / d;
    }
    const float size = 1.f + tanhf(a.size_lin[n]) * a.size_range;
    const float persp = 1.f + tanhf(a.persp_lin[n]) * a.persp_range;
    scale_focal[n] = persp;
    scale_dist[n] = size * persp;
}

struct HeadGrads {            // upstream gradients, one set per group of `rows` consecutive rows (null: not differentiated)
    const float* g[8][5];     // azim [rows][2], elev, theta, scale_focal [rows], scale_dist [rows]
    int rows;
};

__global__ void estimator_head_bwd_kernel(HeadArgs a, HeadGrads G, float* __restrict__ g_trig, float* __restrict__ g_size,
                                          float* __restrict__ g_persp) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const int grp = n / G.rows, r = n - grp * G.rows;
    for (int k = 0; k < 3; ++k) {
        const float x = a.trig[n * 6 + 2 * k], y = a.trig[n * 6 + 2 * k + 1];
        const float* g = G.g[grp][k];
        const float gx = g ? g[2 * r] : 0.f, gy = g ? g[2 * r + 1] : 0.f;
        const float nr = sqrtf(x * x + y * y);
        float ox, oy;
        if (nr > 1.e-12f) {           // y = x / |x| :  dx = (g - y (y . g)) / |x|
            const float ux = x / nr, uy = y / nr, dot = ux * gx + uy * gy;
            ox = (gx - ux * dot) / nr;
            oy = (gy - uy * dot) / nr;
        } else {                      // clamp_min passes no gradient to the norm below eps
            ox = gx / 1.e-12f;
            oy = gy / 1.e-12f;
        }
        g_trig[n * 6 + 2 * k] = ox;
        g_trig[n * 6 + 2 * k + 1] = oy;
    }
    const float ts = tanhf(a.size_lin[n]), tp = tanhf(a.persp_lin[n]);
    const float size = 1.f + ts * a.size_range, persp = 1.f + tp * a.persp_range;
    const float gf = G.g[grp][3] ? G.g[grp][3][r] : 0.f, gd = G.g[grp][4] ? G.g[grp][4][r] : 0.f;
    g_size[n] = gd * persp * a.size_range * (1.f - ts * ts);
    g_persp[n] = (gf + gd * size) * a.persp_range * (1.f - tp * tp);
}

// ---------------------------------------------------------------------------------------------------------
// camera priors
struct PriorArgs {
    const float* azim; const float* elev; const float* theta;          // [B][2] (cos, sin)
    const float* f_azim; const float* f_elev; const float* f_theta;    // the estimator's outputs on the mirrored images
    int B, n_pow2, emd_p;
    float elev_lo, elev_hi, theta_lo, theta_hi, margin_eps;
    float* out;        // [3]: cam_margin, cam_uniform, cam_sym
    float* grads;      // [6][B][2]: d margin/d elev, d margin/d theta, d uniform/d azim, d sym/d azim, d sym/d elev, d sym/d theta
};

constexpr int PRIOR_MAX = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float* red) {      // fixed order: lanes by xor tree, waves in index order
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}

// margin term of one angle: L1(relu(-angle + lo - eps)) + L1(relu(angle - hi - eps)), angle in degrees (loss.py:100-104)
__device__ __forceinline__ void margin_term(float c, float s, float lo, float hi, float eps, float& v, float& gc, float& gs) {
    const float angle = atan2f(s, c) * 180.f / 3.14159274f;
    const float below = -angle + lo - eps, above = angle - hi - eps;
    v = fmaxf(below, 0.f) + fmaxf(above, 0.f);
    const float ga = (above > 0.f ? 1.f : 0.f) - (below > 0.f ? 1.f : 0.f);
    const float k = ga * (180.f / 3.14159274f) / (c * c + s * s);
    gc = -k * s;
    gs = k * c;
}

__global__ __launch_bounds__(PRIOR_MAX) void camera_prior_kernel(PriorArgs a) {
    __shared__ float key[6][PRIOR_MAX];       // 0..2: empirical cos, sin, cos*sin of the azimuth;  3..5: the uniform prior's
    __shared__ short idx[3][PRIOR_MAX];       // original row of every empirical key
    __shared__ float ge[3][PRIOR_MAX];        // gradient of the uniform loss w.r.t. empirical key, by original row
    __shared__ float red[16];
    const int t = threadIdx.x, B = a.B, n = a.n_pow2;
    const bool live = t < B;
    const float invB = 1.f / (float)B;
    float ca = 0.f, sa = 0.f;
    if (live) { ca = a.azim[2 * t]; sa = a.azim[2 * t + 1]; }
    if (t < n) {
        const float inf = __builtin_inff();
        // grid = arange(1, 2B, 2) * pi / B  (float32, loss.py:143)
        const float grid = (float)(2 * t + 1) * 3.14159274f / (float)B;
        const float pc = cosf(grid), ps = sinf(grid);
        key[0][t] = live ? ca : inf; key[1][t] = live ? sa : inf; key[2][t] = live ? ca * sa : inf;
        key[3][t] = live ? pc : inf; key[4][t] = live ? ps : inf; key[5][t] = live ? pc * ps : inf;
        idx[0][t] = idx[1][t] = idx[2][t] = (short)t;
    }
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1)                     // bitonic sort, ascending, six arrays side by side
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = t ^ j;
            if (t < n && p > t) {
                const bool up = (t & k) == 0;
                for (int q = 0; q < 6; ++q) {
                    const float x = key[q][t], y = key[q][p];
                    if ((x > y) == up) {
                        key[q][t] = y; key[q][p] = x;
                        if (q < 3) { const short i0 = idx[q][t]; idx[q][t] = idx[q][p]; idx[q][p] = i0; }
                    }
                }
            }
            __syncthreads();
        }
    // ---- cam_uniform (loss.py:140-167): d = sort(prior) - sort(empirical) per quantity
    float d[3] = {0.f, 0.f, 0.f}, usum = 0.f;
    float nrm[3] = {0.f, 0.f, 0.f};
    for (int q = 0; q < 3; ++q) {
        if (live) d[q] = key[3 + q][t] - key[q][t];
        const float s = block_sum_1024(a.emd_p == 1 ? fabsf(d[q]) : d[q] * d[q], red);
        if (a.emd_p == 1) usum += s * invB;                       // d.abs().mean()
        else { nrm[q] = sqrtf(s); usum += nrm[q]; }               // torch.norm(d, p=2)
    }
    const float uniform = a.emd_p == 1 ? usum / 3.f : usum / (3.f * (float)B);
    for (int q = 0; q < 3; ++q)
        if (live) {
            float g;        // d uniform / d empirical key = -(d uniform / d d)
            if (a.emd_p == 1) g = -((d[q] > 0.f) - (d[q] < 0.f)) * invB / 3.f;
            else g = nrm[q] > 0.f ? -(d[q] / nrm[q]) / (3.f * (float)B) : 0.f;
            ge[q][idx[q][t]] = g;
        }
    __syncthreads();
    // ---- cam_margin and cam_sym, per image
    float mv = 0.f, sv = 0.f;
    if (live) {
        float* gm_e = a.grads + (size_t)0 * B * 2, *gm_t = a.grads + (size_t)1 * B * 2, *gu_a = a.grads + (size_t)2 * B * 2;
        float* gs3[3] = {a.grads + (size_t)3 * B * 2, a.grads + (size_t)4 * B * 2, a.grads + (size_t)5 * B * 2};
        gu_a[2 * t] = ge[0][t] + ge[2][t] * sa;
        gu_a[2 * t + 1] = ge[1][t] + ge[2][t] * ca;
        const float ce = a.elev[2 * t], se = a.elev[2 * t + 1], ct = a.theta[2 * t], st = a.theta[2 * t + 1];
        float v, gc, gs;
        margin_term(ce, se, a.elev_lo, a.elev_hi, a.margin_eps, v, gc, gs);
        mv += v; gm_e[2 * t] = gc * invB; gm_e[2 * t + 1] = gs * invB;
        margin_term(ct, st, a.theta_lo, a.theta_hi, a.margin_eps, v, gc, gs);
        mv += v; gm_t[2 * t] = gc * invB; gm_t[2 * t + 1] = gs * invB;
        // mirrored image: azimuth and roll change sign (sin flips), elevation is unchanged (loss.py:121-137)
        const float* tr[3] = {a.azim, a.elev, a.theta};
        const float* fl[3] = {a.f_azim, a.f_elev, a.f_theta};
        const float sign[3] = {-1.f, 1.f, -1.f};
        for (int q = 0; q < 3; ++q) {
            const float d0 = tr[q][2 * t] - fl[q][2 * t], d1 = sign[q] * tr[q][2 * t + 1] - fl[q][2 * t + 1];
            sv += d0 * d0 + d1 * d1;
            gs3[q][2 * t] = 2.f * d0 * invB;
            gs3[q][2 * t + 1] = 2.f * d1 * sign[q] * invB;
        }
    }
    const float margin = block_sum_1024(mv, red) * invB;
    const float sym = block_sum_1024(sv, red) * invB;
    if (t == 0) { a.out[0] = margin; a.out[1] = uniform; a.out[2] = sym; }
}

// upstream gradients of the three losses (device scalars, null: not differentiated) -> gradients of the six estimator outputs
__global__ void camera_prior_bwd_kernel(const float* __restrict__ grads, int B, const float* G_margin, const float* G_uniform,
                                        const float* G_sym, float* __restrict__ g_azim, float* __restrict__ g_elev,
                                        float* __restrict__ g_theta, float* __restrict__ g_fa, float* __restrict__ g_fe,
                                        float* __restrict__ g_ft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // over [B][2]
    if (i >= 2 * B) return;
    const float Gm = G_margin ? *G_margin : 0.f, Gu = G_uniform ? *G_uniform : 0.f, Gs = G_sym ? *G_sym : 0.f;
    const size_t P = (size_t)B * 2;
    const float sa = grads[3 * P + i], se = grads[4 * P + i], st = grads[5 * P + i];
    g_azim[i] = Gu * grads[2 * P + i] + Gs * sa;
    g_elev[i] = Gm * grads[0 * P + i] + Gs * se;
    g_theta[i] = Gm * grads[1 * P + i] + Gs * st;
    // d sym / d flipped: -(t0 - f0) for the cosine, -(sign t1 - f1) = -sign * (d sym / d t1) for the sine
    const bool sine = i & 1;
    g_fa[i] = Gs * (sine ? sa : -sa);        // sign = -1
    g_fe[i] = Gs * -se;                      // sign = +1
    g_ft[i] = Gs * (sine ? st : -st);        // sign = -1
}

// ---------------------------------------------------------------------------------------------------------
// transform_normal: out[b][r][:] = normals[b][r][:] @ R_b  (to_hom(n) @ invert([R|0])^T, utils/camera.py:85-103)
__global__ void transform_normal_fwd_kernel(const float* __restrict__ normals, const float* __restrict__ pose, int R,
                                            float* __restrict__ out) {
    const int b = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* p = pose + (size_t)b * 12;
    const size_t o = ((size_t)b * R + r) * 3;
    const float n0 = normals[o], n1 = normals[o + 1], n2 = normals[o + 2];
    for (int j = 0; j < 3; ++j) out[o + j] = n0 * p[0 * 4 + j] + n1 * p[1 * 4 + j] + n2 * p[2 * 4 + j];
}

// g_pose[b][i][j] = sum_r normals[b][r][i] g_out[b][r][j]  (translation column: zero)
__global__ __launch_bounds__(256) void transform_normal_bwd_kernel(const float* __restrict__ normals, const float* __restrict__ g_out,
                                                                   int R, float* __restrict__ g_pose) {
    __shared__ float red[9][4];
    const int b = blockIdx.x;
    float acc[9] = {0.f};
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const size_t o = ((size_t)b * R + r) * 3;
        const float n[3] = {normals[o], normals[o + 1], normals[o + 2]}, g[3] = {g_out[o], g_out[o + 1], g_out[o + 2]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) acc[i * 3 + j] += n[i] * g[j];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 9; ++k) {
        float v = acc[k];
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        g_pose[(size_t)b * 12 + threadIdx.x] = j < 3 ? red[i * 3 + j][0] + red[i * 3 + j][1] + red[i * 3 + j][2] + red[i * 3 + j][3] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// loss.all
struct TotalArgs {
    const float* v[16];     // device scalars
    float w[16];
    int K;
};

__global__ void loss_total_kernel(TotalArgs a, float* __restrict__ total, unsigned char* __restrict__ bad) {
    if (threadIdx.x != 0) return;
    float s = 0.f;
    bool b = false;
    for (int k = 0; k < a.K; ++k) {          // the reference's order: all += float(w_k) * loss_k, key by key
        const float v = *a.v[k];
        b |= !isfinite(v);
        s = s + a.w[k] * v;
    }
    *total = s;
    *bad = b ? 1 : 0;
}

__global__ void loss_total_bwd_kernel(TotalArgs a, const float* __restrict__ G, float* __restrict__ g) {
    const int k = threadIdx.x;
    if (k < a.K) g[k] = a.w[k] * *G;
}

}  // namespace sc

extern "C" int sc_estimator_head_forward(const float* trig, const float* size_lin, const float* persp_lin, int n_rows,
                                         float size_range, float persp_range, float* azim, float* elev, float* theta,
                                         float* scale_focal, float* scale_dist, void* stream_) {
    if (n_rows <= 0) return 0;
    sc::HeadArgs a{trig, size_lin, persp_lin, n_rows, size_range, persp_range};
    hipLaunchKernelGGL(sc::estimator_head_fwd_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, (hipStream_t)stream_, a, azim, elev, theta,
                       scale_focal, scale_dist);
    return (int)hipGetLastError();
}

extern "C" int sc_estimator_head_backward(const float* trig, const float* size_lin, const float* persp_lin, int n_rows,
                                          float size_range, float persp_range, const float* const* grads, int n_groups,
                                          float* g_trig, float* g_size_lin, float* g_persp_lin, void* stream_) {
    if (n_rows <= 0) return 0;
    if (n_groups < 1 || n_groups > 8 || n_rows % n_groups != 0) return -1;
    sc::HeadArgs a{trig, size_lin, persp_lin, n_rows, size_range, persp_range};
    sc::HeadGrads G;
    G.rows = n_rows / n_groups;
    for (int g = 0; g < 8; ++g)
        for (int k = 0; k < 5; ++k) G.g[g][k] = g < n_groups ? grads[g * 5 + k] : nullptr;
    hipLaunchKernelGGL(sc::estimator_head_bwd_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, (hipStream_t)stream_, a, G, g_trig,
                       g_size_lin, g_persp_lin);
    return (int)hipGetLastError();
}

extern "C" int sc_camera_prior_max_images(void) { return sc::PRIOR_MAX; }

extern "C" int sc_camera_prior_forward(const float* azim, const float* elev, const float* theta, const float* flip_azim,
                                       const float* flip_elev, const float* flip_theta, int n_images, float elev_lo, float elev_hi,
                                       float theta_lo, float theta_hi, float margin_eps, int emd_p, float* out, float* grads,
                                       void* stream_) {
    if (n_images <= 0 || n_images > sc::PRIOR_MAX || (emd_p != 1 && emd_p != 2)) return -1;
    int n = 2;
    while (n < n_images) n <<= 1;
    sc::PriorArgs a{azim, elev, theta, flip_azim, flip_elev, flip_theta, n_images, n, emd_p, elev_lo, elev_hi, theta_lo, theta_hi,
                    margin_eps, out, grads};
    const int threads = ((n > n_images ? n : n_images) + 63) / 64 * 64;
    hipLaunchKernelGGL(sc::camera_prior_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

extern "C" int sc_camera_prior_backward(const float* grads, int n_images, const float* G_margin, const float* G_uniform,
                                        const float* G_sym, float* g_azim, float* g_elev, float* g_theta, float* g_flip_azim,
                                        float* g_flip_elev, float* g_flip_theta, void* stream_) {
    if (n_images <= 0) return 0;
    hipLaunchKernelGGL(sc::camera_prior_bwd_kernel, dim3((2 * n_images + 255) / 256), dim3(256), 0, (hipStream_t)stream_, grads, n_images,
                       G_margin, G_uniform, G_sym, g_azim, g_elev, g_theta, g_flip_azim, g_flip_elev, g_flip_theta);
    return (int)hipGetLastError();
}

extern "C" int sc_transform_normal_forward(const float* normals, const float* pose, int n_images, int n_per_image, float* out,
                                           void* stream_) {
    if (n_images <= 0 || n_per_image <= 0) return 0;
    hipLaunchKernelGGL(sc::transform_normal_fwd_kernel, dim3((n_per_image + 255) / 256, n_images), dim3(256), 0, (hipStream_t)stream_,
                       normals, pose, n_per_image, out);
    return (int)hipGetLastError();
}

extern "C" int sc_transform_normal_backward(const float* normals, const float* g_out, int n_images, int n_per_image, float* g_pose,
                                            void* stream_) {
    if (n_images <= 0) return 0;
    hipLaunchKernelGGL(sc::transform_normal_bwd_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream_, normals, g_out, n_per_image,
                       g_pose);
    return (int)hipGetLastError();
}

extern "C" int sc_loss_total_forward(const float* const* values, const float* weights, int n_terms, float* total, unsigned char* bad,
                                     void* stream_) {
    if (n_terms < 0 || n_terms > 16) return -1;
    sc::TotalArgs a;
    a.K = n_terms;
    for (int k = 0; k < 16; ++k) { a.v[k] = k < n_terms ? values[k] : nullptr; a.w[k] = k < n_terms ? weights[k] : 0.f; }
    hipLaunchKernelGGL(sc::loss_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a, total, bad);
    return (int)hipGetLastError();
}

extern "C" int sc_loss_total_backward(const float* weights, int n_terms, const float* G, float* g_values, void* stream_) {
    if (n_terms < 0 || n_terms > 16) return -1;
    if (n_terms == 0) return 0;
    sc::TotalArgs a;
    a.K = n_terms;
    for (int k = 0; k < 16; ++k) { a.v[k] = nullptr; a.w[k] = k < n_terms ? weights[k] : 0.f; }
    hipLaunchKernelGGL(sc::loss_total_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a, G, g_values);
    return (int)hipGetLastError();
}
