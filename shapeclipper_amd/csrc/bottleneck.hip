// bottleneck.hip -- the 1x1 "Bottleneck_Linear" blocks of the estimator heads and the latent projectors (reference
// model/view_estimator.py:6-33, model/graph.py:16-44) as fused launches:
//     a1  = relu(bn1(x W1^T))            out = relu(bn2(a1 W2^T) + x)
// On a 1x1 map a block is two [N, C] x [C, C] products with N = 32 .. 96 rows, each followed by a BatchNorm over the N rows.  As stock
// operators one block was ~14 launches per step (rocBLAS mm x 6, BatchNorm x 4, adds, copies), 9 blocks per step, every launch ~5-8 us for
// ~0.1 us of work (tools/aten_census.py: 83 mm + their satellites).  Here a workgroup OWNS 16 output channels for all N rows, so the
// BatchNorm statistics of its channels are local to it:
//   sc_linear_bn_forward   product (fp32 MFMA 16x16x4) + batch statistics + affine + residual + ReLU             1 launch per linear
//   sc_linear_bn_backward  [gradient = next layer's  gy W  (+ skip gradient) | given] -> ReLU mask -> BatchNorm backward -> gy,
//                          dW = gy^T x, dgamma, dbeta                                                             1 launch per linear
//   sc_linear_backward_data  dx = gy W (+ skip gradient): the head of a chain                                     1 launch per block
// Every sum has a fixed order (lane groups by shuffle, waves in index order through LDS): bit-reproducible.  nn.BatchNorm2d semantics
// as csrc/bn_act.hip: `groups` stacked sub-batches with their own statistics, running statistics updated once per group in order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shapeclipper_hip.h"

namespace sc {
namespace bl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MAXG = 4;        // sub-batches
constexpr int MAXT = 8;        // 16-row tiles (N <= 128)
constexpr int WAVES = 4;
constexpr int TPW = MAXT / WAVES;      // row tiles per wave

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4e(const float4& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// acc[u][r] (lane: channel n = lane & 15, row 16 (wave + 4 u) + 4 kg + r) = sum_k A[row][k] * B[k][n], A = a [N][K] row-major,
// B[k][n] = BT ? b[k * ldb + n0 + n] : b[(n0 + n) * ldb + k].  K % 64 == 0.
// The four waves split K: wave w multiplies ALL row tiles by its quarter of K (operands double-buffered in registers: the loads of the
// next 16 K values are in flight under the MFMAs of the current ones -- these products are latency-bound, 32 workgroups on 256 CUs),
// the quarters are added in wave order through LDS (fixed order), and wave w leaves with the tiles w and w + 4 it owns in the epilogue.
// The K index a lane group contributes to an MFMA is arbitrary as long as both operands agree: element r of the float4 at
// k = 16 t + 4 kg is K index 16 t + 4 kg + r.
template <bool BT>
__device__ __forceinline__ void gemm_rows(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int n0, int K, int N,
                                          int wave, int lane, float* __restrict__ red, f32x4 (&acc)[TPW]) {
    const int i = lane & 15, kg = lane >> 4, ntile = (N + 15) >> 4;
    const int kq = K / WAVES, k0 = wave * kq + 4 * kg;
    const float* ap[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) ap[t] = a + (size_t)min(16 * t + i, N - 1) * lda + k0;
    const float* bp = BT ? b + (size_t)k0 * ldb + n0 + i : b + (size_t)(n0 + i) * ldb + k0;
    auto load_b = [&](int kk) {
        float4 v;
        if (BT) {
            v.x = bp[(size_t)(kk + 0) * ldb]; v.y = bp[(size_t)(kk + 1) * ldb]; v.z = bp[(size_t)(kk + 2) * ldb]; v.w = bp[(size_t)(kk + 3) * ldb];
        } else {
            v = *reinterpret_cast<const float4*>(bp + kk);
        }
        return v;
    };
    f32x4 part[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 av[MAXT], an[MAXT], bv = load_b(0), bn = bv;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        av[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < ntile) av[t] = *reinterpret_cast<const float4*>(ap[t]);
        an[t] = av[t];
    }
#pragma unroll 1
    for (int kk = 0; kk < kq; kk += 16) {
        if (kk + 16 < kq) {
            bn = load_b(kk + 16);
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (t < ntile) an[t] = *reinterpret_cast<const float4*>(ap[t] + kk + 16);
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[t] = mfma(f4e(av[t], r), f4e(bv, r), part[t]);
            }
        bv = bn;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) av[t] = an[t];
    }
    __syncthreads();                                               // (`red` may still be read by the previous phase)
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (t < ntile) *reinterpret_cast<f32x4*>(red + ((wave * MAXT + t) * 64 + lane) * 4) = part[t];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int t = wave + WAVES * u;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < ntile) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += *reinterpret_cast<const f32x4*>(red + ((w * MAXT + t) * 64 + lane) * 4);
        }
        acc[u] = s;
    }
    __syncthreads();
}

// per-group sums of v[u][r] over the rows of the workgroup, for this lane's channel: lane groups by shuffle, waves through LDS in index
// order.  out[g] is valid in every thread.  red: [WAVES][MAXG][16] floats.
__device__ __forceinline__ void group_sums(const float (&v)[TPW][4], const int (&grp)[TPW][4], int G, float* red, int wave, int lane,
                                           float (&out)[MAXG]) {
    float s[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) s[g] = 0.f;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
                if (grp[u][r] == g) s[g] += v[u][r];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        s[g] += __shfl_xor(s[g], 16);
        s[g] += __shfl_xor(s[g], 32);
    }
    __syncthreads();                                              // (the previous use of `red` is over)
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < MAXG; ++g) red[(wave * MAXG + g) * 16 + lane] = s[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red[(w * MAXG + g) * 16 + (lane & 15)];
        out[g] = t;
    }
}

struct FwdArgs {
    const float *x, *w, *gamma, *beta, *res;
    float *y, *out, *save_mean, *save_rstd, *run_mean, *run_var;
    int64_t* n_tracked;
    int N, Cin, Cout, G, training, relu;
    float eps, momentum;
};

__global__ __launch_bounds__(64 * WAVES) void linear_bn_fwd_kernel(FwdArgs a) {
    __shared__ float red[WAVES * MAXG * 16];
    __shared__ __attribute__((aligned(16))) float gred[WAVES * MAXT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
    const int c0 = blockIdx.x * 16, c = c0 + i, N = a.N, G = a.G, Ng = N / G;
    f32x4 acc[TPW];
    gemm_rows<false>(a.x, a.Cin, a.w, a.Cin, c0, a.Cin, N, wave, lane, gred, acc);
    float v[TPW][4];
    int grp[TPW][4], rowi[TPW][4];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave + WAVES * u) + 4 * kg + r;
            rowi[u][r] = row;
            grp[u][r] = row < N ? row / Ng : -1;
            v[u][r] = acc[u][r];
            if (row < N) a.y[(size_t)row * a.Cout + c] = acc[u][r];
        }
    float mean[MAXG], rstd[MAXG];
    const float n = (float)Ng;
    if (a.training) {
        float s[MAXG], q[MAXG];
        group_sums(v, grp, G, red, wave, lane, s);
        float d2[TPW][4];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) mean[g] = s[g] / n;
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = 0.f;
#pragma unroll
                for (int g = 0; g < MAXG; ++g)
                    if (grp[u][r] == g) m = mean[g];
                const float d = v[u][r] - m;
                d2[u][r] = d * d;
            }
        group_sums(d2, grp, G, red, wave, lane, q);
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const float var = q[g] / n;                            // biased: the normaliser (two-pass: no cancellation)
            rstd[g] = rsqrtf(var + a.eps);
            if (g < G && threadIdx.x < 16 && a.run_mean) {         // the G momentum updates of this thread's channel, in group order
                a.run_mean[c] = (1.f - a.momentum) * a.run_mean[c] + a.momentum * mean[g];
                a.run_var[c] = (1.f - a.momentum) * a.run_var[c] + a.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_tracked) *a.n_tracked += G;
    } else {
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            mean[g] = a.run_mean[c];
            rstd[g] = rsqrtf(a.run_var[c] + a.eps);
        }
    }
    if (threadIdx.x < 16)
        for (int g = 0; g < G; ++g) {
            a.save_mean[(size_t)g * a.Cout + c] = mean[g < MAXG ? g : 0];
            a.save_rstd[(size_t)g * a.Cout + c] = rstd[g < MAXG ? g : 0];
        }
    const float gamma = a.gamma[c], beta = a.beta[c];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (grp[u][r] < 0) continue;
            float m = mean[0], rs = rstd[0];
#pragma unroll
            for (int g = 1; g < MAXG; ++g)
                if (grp[u][r] == g) { m = mean[g]; rs = rstd[g]; }
            const float scale = gamma * rs, shift = beta - m * scale;
            float o = fmaf(v[u][r], scale, shift);
            if (a.res) o += a.res[(size_t)rowi[u][r] * a.Cout + c];
            if (a.relu) o = o < 0.f ? 0.f : o;                     // relu(NaN) = NaN, as torch
            a.out[(size_t)rowi[u][r] * a.Cout + c] = o;
        }
}

struct BwdArgs {
    const float *g_out;                       // [N][Cout] gradient of this layer's output, or null: formed from the next layer's
    const float *gy_next, *w_next;            //   gy_next [N][Cnext] x w_next [Cnext][Cout]
    const float *g_add;                       //   (+ g_add [N][Cout], e.g. the skip gradient of the next block), either may be null
    int Cnext;
    const float *out, *y, *save_mean, *save_rstd, *gamma, *x;
    float *gy, *g_res, *dw, *dgamma, *dbeta;
    int N, Cin, Cout, G, training, relu;
};

__global__ __launch_bounds__(64 * WAVES) void linear_bn_bwd_kernel(BwdArgs a) {
    __shared__ float red[WAVES * MAXG * 16];
    __shared__ float gys[MAXT * 16 * 16];                          // gy of this workgroup's 16 channels: [row][channel]
    __shared__ __attribute__((aligned(16))) float gred[WAVES * MAXT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
    const int c0 = blockIdx.x * 16, c = c0 + i, N = a.N, G = a.G, Ng = N / G;
    f32x4 acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!a.g_out) gemm_rows<true>(a.gy_next, a.Cnext, a.w_next, a.Cout, c0, a.Cnext, N, wave, lane, gred, acc);
    float g[TPW][4], xh[TPW][4], gx[TPW][4];
    int grp[TPW][4], rowi[TPW][4];
    float mean[MAXG], rstd[MAXG];
#pragma unroll
    for (int q = 0; q < MAXG; ++q) {
        mean[q] = a.save_mean[(size_t)(q < G ? q : 0) * a.Cout + c];
        rstd[q] = a.save_rstd[(size_t)(q < G ? q : 0) * a.Cout + c];
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave + WAVES * u) + 4 * kg + r;
            rowi[u][r] = row;
            grp[u][r] = row < N ? row / Ng : -1;
            float gv = 0.f, xv = 0.f;
            if (row < N) {
                const size_t o = (size_t)row * a.Cout + c;
                gv = a.g_out ? a.g_out[o] : acc[u][r];
                if (a.g_add) gv += a.g_add[o];
                if (a.relu && !(a.out[o] > 0.f)) gv = 0.f;
                if (a.g_res) a.g_res[o] = gv;
                float m = mean[0], rs = rstd[0];
#pragma unroll
                for (int q = 1; q < MAXG; ++q)
                    if (grp[u][r] == q) { m = mean[q]; rs = rstd[q]; }
                xv = (a.y[o] - m) * rs;
            }
            g[u][r] = gv;
            xh[u][r] = xv;
            gx[u][r] = gv * xv;
        }
    float sb[MAXG], sg[MAXG];
    group_sums(g, grp, G, red, wave, lane, sb);
    group_sums(gx, grp, G, red, wave, lane, sg);
    const float gamma = a.gamma[c], n = (float)Ng;
    if (threadIdx.x < 16) {
        float tg = 0.f, tb = 0.f;
        for (int q = 0; q < G; ++q) { tg += sg[q]; tb += sb[q]; }
        a.dgamma[c] = tg;
        a.dbeta[c] = tb;
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gyv = 0.f;
            if (grp[u][r] >= 0) {
                float rs = rstd[0], b_ = sb[0], g_ = sg[0];
#pragma unroll
                for (int q = 1; q < MAXG; ++q)
                    if (grp[u][r] == q) { rs = rstd[q]; b_ = sb[q]; g_ = sg[q]; }
                gyv = a.training ? gamma * rs * (g[u][r] - b_ / n - xh[u][r] * g_ / n) : g[u][r] * gamma * rs;
                a.gy[(size_t)rowi[u][r] * a.Cout + c] = gyv;
            }
            gys[rowi[u][r] * 16 + i] = gyv;                        // (rows past N: zeros)
        }
    __syncthreads();
    // dW[c0 + m][:] = sum_rows gy[row][m] x[row][:]: wave w takes the 64-column groups w, w + 4, ...; a lane's float4 of x (columns
    // 4 n .. 4 n + 3 of the group) feeds four MFMAs whose accumulators are those four columns
    const int ntile = (N + 15) / 16, nstep = 4 * ntile;            // K steps of the row sum: step s = (tile s >> 2, r = s & 3) <-> rows 16 t + 4 kg + r
    for (int cg = wave; cg < a.Cin / 64; cg += WAVES) {
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* xc = a.x + 64 * cg + 4 * i;
        auto load_x = [&](int st) {
            const int row = 16 * (st >> 2) + 4 * kg + (st & 3);
            return *reinterpret_cast<const float4*>(xc + (size_t)min(row, N - 1) * a.Cin);
        };
        constexpr int PF = 8;                                      // loads in flight per lane (the products are latency-bound)
        float4 xb[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) xb[q] = load_x(min(q, nstep - 1));
#pragma unroll 1
        for (int s0 = 0; s0 < nstep; s0 += PF) {
            float4 xc_[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) xc_[q] = xb[q];
            if (s0 + PF < nstep) {
#pragma unroll
                for (int q = 0; q < PF; ++q) xb[q] = load_x(min(s0 + PF + q, nstep - 1));
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int st = s0 + q;
                if (st < nstep) {
                    const float av = gys[(16 * (st >> 2) + 4 * kg + (st & 3)) * 16 + i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = mfma(av, f4e(xc_[q], j), d[j]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(a.dw + (size_t)(c0 + 4 * kg + r) * a.Cin + 64 * cg + 4 * i) = make_float4(d[0][r], d[1][r], d[2][r], d[3][r]);
    }
}

// dx [N][Cin] = gy [N][Cout] w [Cout][Cin] (+ g_add)
__global__ __launch_bounds__(64 * WAVES) void linear_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                      const float* __restrict__ g_add, float* __restrict__ dx, int N, int Cin,
                                                                      int Cout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
    __shared__ __attribute__((aligned(16))) float gred[WAVES * MAXT * 256];
    const int c0 = blockIdx.x * 16;
    f32x4 acc[TPW];
    gemm_rows<true>(gy, Cout, w, Cin, c0, Cout, N, wave, lane, gred, acc);
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave + WAVES * u) + 4 * kg + r;
            if (row < N) {
                const size_t o = (size_t)row * Cin + c0 + i;
                dx[o] = acc[u][r] + (g_add ? g_add[o] : 0.f);
            }
        }
}

static bool ok(int N, int Cin, int Cout, int G) {
    return N > 0 && N <= 16 * MAXT && G >= 1 && G <= MAXG && N % G == 0 && Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 16 == 0;
}

}  // namespace bl
}  // namespace sc

extern "C" int sc_linear_bn_supported(int N, int Cin, int Cout, int groups) { return sc::bl::ok(N, Cin, Cout, groups) ? 1 : 0; }

extern "C" int sc_linear_bn_forward(const float* x, const float* w, const float* gamma, const float* beta, const float* res, float* y, float* out,
                                    float* save_mean, float* save_rstd, float* run_mean, float* run_var, int64_t* n_tracked, int N, int Cin,
                                    int Cout, int groups, int training, int relu, float eps, float momentum, void* stream) {
    if (!sc::bl::ok(N, Cin, Cout, groups) || (!training && (!run_mean || !run_var))) return (int)hipErrorInvalidValue;
    sc::bl::FwdArgs a{x, w, gamma, beta, res, y, out, save_mean, save_rstd, run_mean, run_var, n_tracked, N, Cin, Cout, groups, training, relu, eps, momentum};
    hipLaunchKernelGGL(sc::bl::linear_bn_fwd_kernel, dim3(Cout / 16), dim3(64 * sc::bl::WAVES), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int sc_linear_bn_backward(const float* g_out, const float* gy_next, const float* w_next, const float* g_add, int Cnext, const float* out,
                                     const float* y, const float* save_mean, const float* save_rstd, const float* gamma, const float* x, float* gy,
                                     float* g_res, float* dw, float* dgamma, float* dbeta, int N, int Cin, int Cout, int groups, int training, int relu,
                                     void* stream) {
    if (!sc::bl::ok(N, Cin, Cout, groups) || (!g_out && (!gy_next || !w_next || Cnext <= 0 || Cnext % 16))) return (int)hipErrorInvalidValue;
    sc::bl::BwdArgs a{g_out, gy_next, w_next, g_add, Cnext, out, y, save_mean, save_rstd, gamma, x, gy, g_res, dw, dgamma, dbeta, N, Cin, Cout, groups,
                      training, relu};
    hipLaunchKernelGGL(sc::bl::linear_bn_bwd_kernel, dim3(Cout / 16), dim3(64 * sc::bl::WAVES), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int sc_linear_backward_data(const float* gy, const float* w, const float* g_add, float* dx, int N, int Cin, int Cout, void* stream) {
    if (N <= 0 || N > 16 * sc::bl::MAXT || Cin % 16 || Cout % 16 || Cin <= 0 || Cout <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sc::bl::linear_bwd_data_kernel, dim3(Cin / 16), dim3(64 * sc::bl::WAVES), 0, (hipStream_t)stream, gy, w, g_add, dx, N, Cin, Cout);
    return (int)hipGetLastError();
}
