// bn_act.hip -- BatchNorm2d fused with the residual add and ReLU that surround it in a ResNet basic block, and with
// the 3x3/2 max-pool of the stem (NCHW fp32, forward + hand-written backward).
//
// after the renderer kernels the encoders dominate the step (reference model/graph.py:16-65,
// model/view_estimator.py:35-103: torchvision ResNet-18/34).  Convolutions stay on MIOpen; everything between them is
// HBM-bound streaming that the stock path spreads over BN (read x twice, write), add (2R+1W), ReLU (1R+1W) forward
// and threshold_backward + BN backward on the way back.  Here per BN:
//   forward  : stats pass (1R) + apply pass  y = relu(gamma*(x-mean)*rstd + beta [+ res])   (1-2R, 1W)
//   backward : stats pass  g = dy*[y>0]; sum g, sum g*xhat [; dres = g]                     (2-3R, 0-1W)
//              apply pass  dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))                (2R, 1W)
// The ReLU mask of the no-residual form is recomputed from x (same fma as the forward, bit-identical), so y is not read.
// Statistics: one block per (channel, image subset); partial sums of (x-K), (x-K)^2 with K = first element of the
// channel (shifted-data variance), combined in a fixed order -> deterministic.  Running statistics (momentum update,
// unbiased variance) and num_batches_tracked are updated in the apply kernel, as nn.BatchNorm2d does in training.
// Groups: the batch may hold G independent sub-batches of N/G images (the passes the reference runs one after the
// other through the same network: input view, CLIP neighbour, mirrored image).  Statistics, normalisation and the
// backward means are per group; the running statistics receive the G momentum updates in group order, exactly as G
// sequential forward calls would apply them; dgamma/dbeta are summed over the groups (shared parameters).
// Bound: HBM (8 TB/s).  Algorithmic bytes per element: forward 12-16 B, backward 20-28 B.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

namespace sc {

constexpr int BN_T = 256;

template <int W> struct Vec;
template <> struct Vec<4> {
    float v[4];
    __device__ __forceinline__ static Vec ld(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        return Vec{{t.x, t.y, t.z, t.w}};
    }
    __device__ __forceinline__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<1> {
    float v[1];
    __device__ __forceinline__ static Vec ld(const float* p) { return Vec{{*p}}; }
    __device__ __forceinline__ void st(float* p) const { *p = v[0]; }
};

// This block's share of the batch: group g (images n0 .. n0+Ng-1), split s of S inside the group.
struct BnShare {
    int g, s, S, n0, Ng;
};
__device__ __forceinline__ BnShare bn_share(int N, int G) {
    BnShare sh;
    sh.S = gridDim.y / G;
    sh.g = blockIdx.y / sh.S;
    sh.s = blockIdx.y - sh.g * sh.S;
    sh.Ng = N / G;
    sh.n0 = sh.g * sh.Ng;
    return sh;
}

// Visit the elements of channel c in images n0+s, n0+s+S, ... of the group, float4-wide when HW % 4 == 0.
template <class F>
__device__ __forceinline__ void bn_iterate(const BnShare& sh, int C, int HW, int c, F&& f) {
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S;
    if ((HW & 3) == 0) {
        const int hw4 = HW >> 2, total = cnt * hw4;
#pragma unroll 2
        for (int idx = threadIdx.x; idx < total; idx += BN_T) {
            const int nl = idx / hw4, i = idx - nl * hw4;
            f(((size_t)(sh.n0 + sh.s + nl * sh.S) * C + c) * HW + 4 * i, std::integral_constant<int, 4>{});
        }
    } else {
        const int total = cnt * HW;
        for (int idx = threadIdx.x; idx < total; idx += BN_T) {
            const int nl = idx / HW, i = idx - nl * HW;
            f(((size_t)(sh.n0 + sh.s + nl * sh.S) * C + c) * HW + i, std::integral_constant<int, 1>{});
        }
    }
}

__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {   // red: 2 * (BN_T / 64) floats
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[w] = a;
        red[BN_T / 64 + w] = b;
    }
    __syncthreads();
    a = 0.f;
    b = 0.f;
    for (int k = 0; k < BN_T / 64; ++k) {
        a += red[k];
        b += red[BN_T / 64 + k];
    }
}

// partial layout: [C][G*S][2].  Sum of the S pairs of (channel c, group g), fixed order; every thread gets the totals.
__device__ __forceinline__ void combine_partials(const float* partial, int c, int g, int S, int G, float& a, float& b) {
    a = 0.f;
    b = 0.f;
    const float* p = partial + ((size_t)c * G * S + (size_t)g * S) * 2;
    for (int k = 0; k < S; ++k) {          // S <= 32: a short uniform loop
        a += p[2 * k];
        b += p[2 * k + 1];
    }
}

__device__ __forceinline__ void store_partial(float* partial, int c, const BnShare& sh, int G, float a, float b) {
    float* p = partial + ((size_t)c * G * sh.S + (size_t)sh.g * sh.S + sh.s) * 2;
    p[0] = a;
    p[1] = b;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float* __restrict__ x, int N, int C, int HW, int G,
                                                        float* __restrict__ partial) {
    __shared__ float red[2 * BN_T / 64];
    const int c = blockIdx.x;
    const BnShare sh = bn_share(N, G);
    const float K = x[((size_t)sh.n0 * C + c) * HW];
    float s1 = 0.f, s2 = 0.f;
    bn_iterate(sh, C, HW, c, [&](size_t off, auto w) {
        const auto v = Vec<decltype(w)::value>::ld(x + off);
#pragma unroll
        for (int k = 0; k < decltype(w)::value; ++k) {
            const float d = v.v[k] - K;
            s1 += d;
            s2 += d * d;
        }
    });
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) store_partial(partial, c, sh, G, s1, s2);
}

struct BnStatArgs {      // what the forward needs to turn partial sums into (mean, rstd) and running statistics
    const float* x; const float* partial;
    float* save_mean; float* save_rstd;      // [G][C]
    float* run_mean; float* run_var; int64_t* n_tracked;
    int N, C, HW, G, training;
    float eps, momentum;
};

// (mean, rstd) of channel c for this block's group; block (blockIdx.y == 0) also records the statistics of every
// group for the backward and applies the G running-statistics updates in group order.
__device__ __forceinline__ void bn_forward_stats(const BnStatArgs& a, int c, const BnShare& sh, float& mean, float& rstd) {
    const float n = (float)sh.Ng * (float)a.HW;
    auto group_stats = [&](int g, float& m, float& var) {
        float s1, s2;
        combine_partials(a.partial, c, g, sh.S, a.G, s1, s2);
        const float dm = s1 / n;
        const float v0 = s2 / n - dm * dm;
        var = v0 < 0.f ? 0.f : v0;                 // (not fmaxf: a NaN variance stays NaN, as in torch)
        m = a.x[((size_t)g * sh.Ng * a.C + c) * a.HW] + dm;
    };
    if (a.training) {
        float var;
        group_stats(sh.g, mean, var);
        rstd = rsqrtf(var + a.eps);
    } else {
        mean = a.run_mean[c];
        rstd = rsqrtf(a.run_var[c] + a.eps);
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        for (int g = 0; g < a.G; ++g) {
            float m = mean, r = rstd;
            if (a.training) {
                float var;
                group_stats(g, m, var);
                r = rsqrtf(var + a.eps);
                if (a.run_mean) {
                    a.run_mean[c] = (1.f - a.momentum) * a.run_mean[c] + a.momentum * m;
                    a.run_var[c] = (1.f - a.momentum) * a.run_var[c] + a.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
                }
            }
            a.save_mean[(size_t)g * a.C + c] = m;
            a.save_rstd[(size_t)g * a.C + c] = r;
        }
        if (a.training && a.n_tracked && c == 0) *a.n_tracked += a.G;
    }
}

struct BnFwdArgs {
    BnStatArgs st;
    const float* res; const float* gamma; const float* beta; float* y;
    int relu;
};

__global__ __launch_bounds__(BN_T) void bn_apply_fwd_kernel(BnFwdArgs a) {
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.st.N, a.st.G);
    float mean, rstd;
    bn_forward_stats(a.st, c, sh, mean, rstd);
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    const float* x = a.st.x;
    bn_iterate(sh, a.st.C, a.st.HW, c, [&](size_t off, auto w) {
        constexpr int W = decltype(w)::value;
        auto v = Vec<W>::ld(x + off);
        if (a.res) {
            const auto r = Vec<W>::ld(a.res + off);
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = fmaf(v.v[k], scale, shift) + r.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = fmaf(v.v[k], scale, shift);
        }
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = v.v[k] < 0.f ? 0.f : v.v[k];      // relu(NaN) = NaN, as torch (fmaxf would return 0)
        }
        v.st(a.y + off);
    });
}

struct BnBwdArgs {
    const float* dy; const float* x; const float* y;      // y: forward output, needed for the mask only when res was added
    const float* gamma; const float* beta; const float* mean; const float* rstd;     // mean, rstd: [G][C]
    float* partial; float* dx; float* dres; float* dgamma; float* dbeta;
    int N, C, HW, G, relu, training, has_res;
};

// g = dy * [output > 0]; the no-residual mask is recomputed from x with the forward's own fma.
template <int W>
__device__ __forceinline__ Vec<W> bn_masked_grad(const BnBwdArgs& a, size_t off, const Vec<W>& xv, float scale, float shift) {
    auto g = Vec<W>::ld(a.dy + off);
    if (a.relu) {
        if (a.has_res) {
            const auto yv = Vec<W>::ld(a.y + off);
#pragma unroll
            for (int k = 0; k < W; ++k) g.v[k] = yv.v[k] <= 0.f ? 0.f : g.v[k];        // threshold_backward: a NaN output passes its gradient
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) g.v[k] = fmaf(xv.v[k], scale, shift) <= 0.f ? 0.f : g.v[k];
        }
    }
    return g;
}

__global__ __launch_bounds__(BN_T) void bn_bwd_stats_kernel(BnBwdArgs a) {
    __shared__ float red[2 * BN_T / 64];
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.N, a.G);
    const float mean = a.mean[(size_t)sh.g * a.C + c], rstd = a.rstd[(size_t)sh.g * a.C + c];
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    float s1 = 0.f, s2 = 0.f;
    bn_iterate(sh, a.C, a.HW, c, [&](size_t off, auto w) {
        constexpr int W = decltype(w)::value;
        const auto xv = Vec<W>::ld(a.x + off);
        const auto g = bn_masked_grad<W>(a, off, xv, scale, shift);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            s1 += g.v[k];
            s2 += g.v[k] * ((xv.v[k] - mean) * rstd);
        }
        if (a.dres) g.st(a.dres + off);
    });
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) store_partial(a.partial, c, sh, a.G, s1, s2);
}

__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(BnBwdArgs a) {
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.N, a.G);
    const float mean = a.mean[(size_t)sh.g * a.C + c], rstd = a.rstd[(size_t)sh.g * a.C + c], gamma = a.gamma[c];
    const float scale = gamma * rstd, shift = a.beta[c] - mean * scale;
    float sg, sgx;
    combine_partials(a.partial, c, sh.g, sh.S, a.G, sg, sgx);
    if (blockIdx.y == 0 && threadIdx.x == 0) {      // parameters are shared by the groups: sum over all of them
        float tg = 0.f, tgx = 0.f;
        for (int g = 0; g < a.G; ++g) {
            float u, v;
            combine_partials(a.partial, c, g, sh.S, a.G, u, v);
            tg += u;
            tgx += v;
        }
        a.dgamma[c] = tgx;
        a.dbeta[c] = tg;
    }
    if (!a.dx) return;
    const float n = (float)sh.Ng * (float)a.HW;
    const float mg = a.training ? sg / n : 0.f, mgx = a.training ? sgx / n : 0.f;
    bn_iterate(sh, a.C, a.HW, c, [&](size_t off, auto w) {
        constexpr int W = decltype(w)::value;
        const auto xv = Vec<W>::ld(a.x + off);
        // with a residual the masked gradient was already written to dres by the stats pass
        const auto g = a.dres ? Vec<W>::ld(a.dres + off) : bn_masked_grad<W>(a, off, xv, scale, shift);
        Vec<W> o;
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = scale * (g.v[k] - mg - (xv.v[k] - mean) * rstd * mgx);
        o.st(a.dx + off);
    });
}

// ---------------------------------------------------------------------------------------------------------
// Stem: y = maxpool3x3/s2/p1( relu( bn(x) ) ), x [N,C,H,W] -> y [N,C,Ho,Wo]; idx = argmax position (h*W+w, first max
// in scan order like torch) kept as int32 for the backward.  BN+ReLU is monotone per channel when scale >= 0, but the
// general form is evaluated (scale may be negative).
struct PoolFwdArgs {
    BnStatArgs st;
    const float* gamma; const float* beta; float* y; int* idx;
    int H, W, Ho, Wo;
};

__global__ __launch_bounds__(BN_T) void bn_relu_pool_fwd_kernel(PoolFwdArgs a) {
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.st.N, a.st.G);
    const int HW = a.H * a.W, C = a.st.C;
    float mean, rstd;
    bn_forward_stats(a.st, c, sh, mean, rstd);
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    const int HoWo = a.Ho * a.Wo;
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S, total = cnt * HoWo;
    for (int t = threadIdx.x; t < total; t += BN_T) {
        const int nl = t / HoWo, o = t - nl * HoWo;
        const int ho = o / a.Wo, wo = o - ho * a.Wo;
        const size_t plane = ((size_t)(sh.n0 + sh.s + nl * sh.S) * C + c);
        const float* xp = a.st.x + plane * HW;
        float best = -__builtin_inff();
        int bi = -1;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int h = 2 * ho - 1 + dh;
            if (h < 0 || h >= a.H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int w = 2 * wo - 1 + dw;
                if (w < 0 || w >= a.W) continue;
                const float u = fmaf(xp[h * a.W + w], scale, shift), v = u < 0.f ? 0.f : u;       // relu(NaN) = NaN
                if (v > best || bi < 0 || v != v) {          // strict >: first maximum in scan order; a NaN wins like in torch's max_pool2d
                    best = v;
                    bi = h * a.W + w;
                }
            }
        }
        a.y[plane * HoWo + o] = best;
        a.idx[plane * HoWo + o] = bi;
    }
}

// Even W (the 112 x 112 stem maps): a lane loads the column pair (2 wo, 2 wo + 1) of its three input rows as one 8-byte vector
// (consecutive lanes = consecutive pairs: fully coalesced, every input element of a row loaded once) and takes column 2 wo - 1 -- already
// normalised and rectified -- from the lane to its left; only a wave's first lane fetches it itself.  Same fma, same comparisons in the
// same scan order as the kernel above: identical values and indices.  (The scalar form reads nine 4-byte values per output at a lane
// stride of two floats: 1.7 TB/s on a pass that moves 307 MB.)
__global__ __launch_bounds__(BN_T) void bn_relu_pool_fwd2_kernel(PoolFwdArgs a) {
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.st.N, a.st.G);
    const int HW = a.H * a.W, C = a.st.C, W = a.W;
    float mean, rstd;
    bn_forward_stats(a.st, c, sh, mean, rstd);
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    const int HoWo = a.Ho * a.Wo;
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S, total = cnt * HoWo;
    const int lane = threadIdx.x & 63;
    for (int t0 = 0; t0 < total; t0 += BN_T) {
        const int t = t0 + threadIdx.x;
        const bool on = t < total;
        const int tt = on ? t : total - 1;
        const int nl = tt / HoWo, o = tt - nl * HoWo;
        const int ho = o / a.Wo, wo = o - ho * a.Wo;
        const size_t plane = ((size_t)(sh.n0 + sh.s + nl * sh.S) * C + c);
        const float* xp = a.st.x + plane * HW;
        float best = -__builtin_inff();
        int bi = -1;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int h = 2 * ho - 1 + dh;
            const bool hv = h >= 0 && h < a.H;
            float2 v = make_float2(0.f, 0.f);
            if (hv) v = *reinterpret_cast<const float2*>(xp + h * W + 2 * wo);
            const float u0 = fmaf(v.x, scale, shift), u1 = fmaf(v.y, scale, shift);
            const float r0 = u0 < 0.f ? 0.f : u0, r1 = u1 < 0.f ? 0.f : u1;               // relu(NaN) = NaN
            float left = __shfl_up(r1, 1);                                                // the left neighbour's column 2 wo - 1 (same row when wo > 0)
            if (lane == 0 && wo > 0 && hv) {
                const float ul = fmaf(xp[h * W + 2 * wo - 1], scale, shift);
                left = ul < 0.f ? 0.f : ul;
            }
            if (!hv) continue;
            if (wo > 0 && (left > best || bi < 0 || left != left)) { best = left; bi = h * W + 2 * wo - 1; }
            if (r0 > best || bi < 0 || r0 != r0) { best = r0; bi = h * W + 2 * wo; }      // strict >: first maximum in scan order; a NaN wins
            if (r1 > best || bi < 0 || r1 != r1) { best = r1; bi = h * W + 2 * wo + 1; }
        }
        if (on) {
            a.y[plane * HoWo + o] = best;
            a.idx[plane * HoWo + o] = bi;
        }
    }
}

// Round 5: the same pass with R consecutive output rows per thread.  The 2 R + 1 input rows of the strip are requested together
// (2 R + 1 eight-byte loads in flight per lane instead of 3), a shared row (2 ho + 1 is the last row of window ho and the first of ho + 1) is
// loaded, normalised and rectified ONCE, and the index arithmetic is paid per strip.  Same fma, same comparisons in the same scan order per
// output: identical values and indices.
template <int R>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_fwd_rows_kernel(PoolFwdArgs a) {
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.st.N, a.st.G);
    const int HW = a.H * a.W, C = a.st.C, W = a.W;
    float mean, rstd;
    bn_forward_stats(a.st, c, sh, mean, rstd);
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    const int HoWo = a.Ho * a.Wo;
    const int RG = (a.Ho + R - 1) / R, per = RG * a.Wo;            // strips per plane
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S, total = cnt * per;
    const int lane = threadIdx.x & 63;
    for (int t0 = 0; t0 < total; t0 += BN_T) {
        const int t = t0 + threadIdx.x;
        const bool on = t < total;
        const int tt = on ? t : total - 1;
        const int nl = tt / per, o = tt - nl * per;
        const int rg = o / a.Wo, wo = o - rg * a.Wo;
        const int ho0 = rg * R;
        const size_t plane = ((size_t)(sh.n0 + sh.s + nl * sh.S) * C + c);
        const float* xp = a.st.x + plane * HW;
        float2 v[2 * R + 1];
        float lf[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) {
            const int h = 2 * ho0 - 1 + k;
            const bool hv = h >= 0 && h < a.H;
            v[k] = make_float2(0.f, 0.f);
            lf[k] = 0.f;
            if (hv) v[k] = *reinterpret_cast<const float2*>(xp + h * W + 2 * wo);
            if (lane == 0 && wo > 0 && hv) lf[k] = xp[h * W + 2 * wo - 1];
        }
        float r0[2 * R + 1], r1[2 * R + 1], left[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) {
            const float u0 = fmaf(v[k].x, scale, shift), u1 = fmaf(v[k].y, scale, shift);
            r0[k] = u0 < 0.f ? 0.f : u0;                                                  // relu(NaN) = NaN
            r1[k] = u1 < 0.f ? 0.f : u1;
            left[k] = __shfl_up(r1[k], 1);                                                // the left neighbour's column 2 wo - 1 (same strip when wo > 0)
            if (lane == 0) {
                const float ul = fmaf(lf[k], scale, shift);
                left[k] = ul < 0.f ? 0.f : ul;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ho = ho0 + r;
            float best = -__builtin_inff();
            int bi = -1;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const int k = 2 * r + dh, h = 2 * ho - 1 + dh;
                if (h < 0 || h >= a.H) continue;
                if (wo > 0 && (left[k] > best || bi < 0 || left[k] != left[k])) { best = left[k]; bi = h * W + 2 * wo - 1; }
                if (r0[k] > best || bi < 0 || r0[k] != r0[k]) { best = r0[k]; bi = h * W + 2 * wo; }      // strict >: first maximum in scan order; a NaN wins
                if (r1[k] > best || bi < 0 || r1[k] != r1[k]) { best = r1[k]; bi = h * W + 2 * wo + 1; }
            }
            if (on && ho < a.Ho) {
                a.y[plane * HoWo + ho * a.Wo + wo] = best;
                a.idx[plane * HoWo + ho * a.Wo + wo] = bi;
            }
        }
    }
}

struct PoolBwdArgs {
    const float* dy; const int* idx; const float* x; const float* gamma; const float* beta; const float* mean;
    const float* rstd; float* partial; float* dx;
    int N, C, H, W, Ho, Wo, G;
    float* dgamma; float* dbeta;
    int training;
};

// Pass 1 of the stem backward.  One thread owns a 2x2 block of BN-output positions (rows 2i,2i+1, cols 2j,2j+1): the
// only pooling windows that can have their argmax there are (i..i+1) x (j..j+1), so 4 idx/dy reads serve 4 inputs
// (gather form, no atomics).  The gradient is gated by the ReLU (recomputed from x), written to dx as a temporary and
// reduced into the per-channel sums; pass 2 (bn_bwd_apply_kernel with dres == dx) finishes dx in place.
// V2 (even W): the two columns of the thread's 2 x 2 block are one 8-byte load of x and one 8-byte store of dx per row.
// MODE 0: as described (g to dx + sums; bn_bwd_apply_kernel follows).  MODE 1 / 2 (even W): the gather runs twice instead -- 1: sums only,
// 2: g again and dx = scale * (g - mean(g) - xhat * mean(g xhat)) directly (the same expressions: identical values) -- so that the
// full-resolution g is neither written nor read back: 819 MB instead of 1,127 MB per stem backward at 64 images.
template <bool V2, int MODE>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_bwd_gather_kernel(PoolBwdArgs a) {
    __shared__ float red[2 * BN_T / 64];
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.N, a.G);
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int Hq = (a.H + 1) >> 1, Wq = (a.W + 1) >> 1, Q = Hq * Wq;
    const float mean = a.mean[(size_t)sh.g * a.C + c], rstd = a.rstd[(size_t)sh.g * a.C + c];
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    float sg = 0.f, sgx = 0.f;
    float mg = 0.f, mgx = 0.f;
    if (MODE == 2) {
        float tsg, tsgx;
        combine_partials(a.partial, c, sh.g, sh.S, a.G, tsg, tsgx);
        if (blockIdx.y == 0 && threadIdx.x == 0) {      // parameters are shared by the groups: sum over all of them
            float tg = 0.f, tgx = 0.f;
            for (int g = 0; g < a.G; ++g) {
                float u, v;
                combine_partials(a.partial, c, g, sh.S, a.G, u, v);
                tg += u;
                tgx += v;
            }
            a.dgamma[c] = tgx;
            a.dbeta[c] = tg;
        }
        const float n = (float)sh.Ng * (float)HW;
        mg = a.training ? tsg / n : 0.f;
        mgx = a.training ? tsgx / n : 0.f;
    }
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S, total = cnt * Q;
    for (int t = threadIdx.x; t < total; t += BN_T) {
        const int nl = t / Q, q = t - nl * Q;
        const int i = q / Wq, j = q - i * Wq;
        const size_t plane = ((size_t)(sh.n0 + sh.s + nl * sh.S) * a.C + c);
        int wi[2][2];
        float wd[2][2];
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const bool ok = (i + di) < a.Ho && (j + dj) < a.Wo;
                const size_t o = plane * HoWo + (size_t)(i + di) * a.Wo + (j + dj);
                wi[di][dj] = ok ? a.idx[o] : -1;
                wd[di][dj] = ok ? a.dy[o] : 0.f;
            }
#pragma unroll
        for (int eh = 0; eh < 2; ++eh) {
            const int h = 2 * i + eh;
            if (h >= a.H) continue;
            float2 xrow = make_float2(0.f, 0.f), grow;
            if (V2) xrow = *reinterpret_cast<const float2*>(a.x + plane * HW + h * a.W + 2 * j);
#pragma unroll
            for (int ew = 0; ew < 2; ++ew) {
                const int w = 2 * j + ew;
                if (w >= a.W) continue;
                const int pos = h * a.W + w;
                const float xv = V2 ? (ew ? xrow.y : xrow.x) : a.x[plane * HW + pos];
                float g = 0.f;
                // even row/col: covered only by window i (j); odd: by windows i and i+1 (j and j+1)
#pragma unroll
                for (int di = 0; di <= eh; ++di)
#pragma unroll
                    for (int dj = 0; dj <= ew; ++dj)
                        if (wi[di][dj] == pos) g += wd[di][dj];
                if (!(fmaf(xv, scale, shift) > 0.f)) g = 0.f;        // ReLU gate (a max of 0 carries no gradient)
                const float o = MODE == 2 ? scale * (g - mg - (xv - mean) * rstd * mgx) : g;
                if (V2) { if (ew) grow.y = o; else grow.x = o; }
                else if (MODE != 1) a.dx[plane * HW + pos] = o;
                sg += g;
                sgx += g * ((xv - mean) * rstd);
            }
            if (V2 && MODE != 1) *reinterpret_cast<float2*>(a.dx + plane * HW + h * a.W + 2 * j) = grow;
        }
    }
    if (MODE == 2) return;
    block_sum2(sg, sgx, red);
    if (threadIdx.x == 0) store_partial(a.partial, c, sh, a.G, sg, sgx);
}

// Round 5, even W: the same two gather passes with a strip of R vertically adjacent 2 x 2 blocks per thread.  The windows of block row i are
// rows i and i + 1, so a strip needs R + 1 window rows instead of 2 R; window column j + 1 comes from the lane to the right (consecutive
// lanes = consecutive j of one strip; only a wave's last lane loads it itself): 2 (R + 1) four-byte loads of idx / dy per strip instead of
// 8 R, all of them and the 2 R eight-byte loads of x requested before the first use.  Per position the same expressions in the same order
// as the kernel above.  R = 1 (the default) keeps every thread's set of positions, i.e. results identical to that kernel bit for bit
// (206 -> 188 us at 64 images, 328 -> 276 at 96); with R = 2 / 4 (180 / 272, 178 / 270 us) a thread's share of the per-channel sums is a
// different set of positions and dgamma / dbeta / dx agree to rounding only (tools/perf_stem_pool_ab.py, profiles/r05_stem_pool_ab.txt).
template <int MODE, int R>
__global__ __launch_bounds__(BN_T) void bn_relu_pool_bwd_rows_kernel(PoolBwdArgs a) {
    __shared__ float red[2 * BN_T / 64];
    const int c = blockIdx.x;
    const BnShare sh = bn_share(a.N, a.G);
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int Hq = (a.H + 1) >> 1, Wq = a.W >> 1, RG = (Hq + R - 1) / R, Q = RG * Wq;
    const float mean = a.mean[(size_t)sh.g * a.C + c], rstd = a.rstd[(size_t)sh.g * a.C + c];
    const float scale = a.gamma[c] * rstd, shift = a.beta[c] - mean * scale;
    float sg = 0.f, sgx = 0.f;
    float mg = 0.f, mgx = 0.f;
    if (MODE == 2) {
        float tsg, tsgx;
        combine_partials(a.partial, c, sh.g, sh.S, a.G, tsg, tsgx);
        if (blockIdx.y == 0 && threadIdx.x == 0) {      // parameters are shared by the groups: sum over all of them
            float tg = 0.f, tgx = 0.f;
            for (int g = 0; g < a.G; ++g) {
                float u, v;
                combine_partials(a.partial, c, g, sh.S, a.G, u, v);
                tg += u;
                tgx += v;
            }
            a.dgamma[c] = tgx;
            a.dbeta[c] = tg;
        }
        const float n = (float)sh.Ng * (float)HW;
        mg = a.training ? tsg / n : 0.f;
        mgx = a.training ? tsgx / n : 0.f;
    }
    const int cnt = (sh.Ng - sh.s + sh.S - 1) / sh.S, total = cnt * Q;
    const int lane = threadIdx.x & 63;
    for (int t0 = 0; t0 < total; t0 += BN_T) {
        const int t = t0 + threadIdx.x;
        const bool on = t < total;
        const int tt = on ? t : total - 1;
        const int nl = tt / Q, q = tt - nl * Q;
        const int rg = q / Wq, j = q - rg * Wq, i0 = rg * R;
        const size_t plane = ((size_t)(sh.n0 + sh.s + nl * sh.S) * a.C + c);
        int wi[R + 1][2];
        float wd[R + 1][2];
        float2 xrow[2 * R];
        const bool own_right = lane == 63 && (j + 1) < a.Wo;            // nobody to the right in this wave
#pragma unroll
        for (int di = 0; di <= R; ++di) {
            const bool okr = (i0 + di) < a.Ho;
            const size_t o = plane * HoWo + (size_t)(i0 + di) * a.Wo + j;
            wi[di][0] = okr && j < a.Wo ? a.idx[o] : -1;
            wd[di][0] = okr && j < a.Wo ? a.dy[o] : 0.f;
            wi[di][1] = okr && own_right ? a.idx[o + 1] : -1;
            wd[di][1] = okr && own_right ? a.dy[o + 1] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) {
            const int h = 2 * i0 + k;
            xrow[k] = h < a.H ? *reinterpret_cast<const float2*>(a.x + plane * HW + h * a.W + 2 * j) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int di = 0; di <= R; ++di) {
            const int ri = __shfl_down(wi[di][0], 1);
            const float rd = __shfl_down(wd[di][0], 1);
            if (lane != 63) {                                           // the right neighbour holds (strip, j + 1) when j + 1 < Wq
                const bool okc = (j + 1) < a.Wo;
                wi[di][1] = okc ? ri : -1;
                wd[di][1] = okc ? rd : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {
                const int h = 2 * (i0 + r) + eh;
                if (h >= a.H) continue;
                float2 grow;
#pragma unroll
                for (int ew = 0; ew < 2; ++ew) {
                    const int pos = h * a.W + 2 * j + ew;
                    const float xv = ew ? xrow[2 * r + eh].y : xrow[2 * r + eh].x;
                    float g = 0.f;
                    // even row/col: covered only by window i (j); odd: by windows i and i+1 (j and j+1)
#pragma unroll
                    for (int di = 0; di <= eh; ++di)
#pragma unroll
                        for (int dj = 0; dj <= ew; ++dj)
                            if (wi[r + di][dj] == pos) g += wd[r + di][dj];
                    if (!(fmaf(xv, scale, shift) > 0.f)) g = 0.f;        // ReLU gate (a max of 0 carries no gradient)
                    const float o = MODE == 2 ? scale * (g - mg - (xv - mean) * rstd * mgx) : g;
                    if (ew) grow.y = o; else grow.x = o;
                    if (on) {
                        sg += g;
                        sgx += g * ((xv - mean) * rstd);
                    }
                }
                if (MODE == 2 && on) *reinterpret_cast<float2*>(a.dx + plane * HW + h * a.W + 2 * j) = grow;
            }
        }
    }
    if (MODE == 2) return;
    block_sum2(sg, sgx, red);
    if (threadIdx.x == 0) store_partial(a.partial, c, sh, a.G, sg, sgx);
}

// ---------------------------------------------------------------------------------------------------------
// Small maps with many channels (14 x 14 x 256, 7 x 7 x 512, the 1 x 1 maps of the heads): ONE launch per pass.  A channel of such a
// layer is 12-75 KB for the whole batch, so the two-launch form above spends more time on its second launch (grid fill / drain plus the
// dependent-kernel boundary) than on the bytes: 13-16 us per BatchNorm where the data moves in 3-5.  Here one 512-thread block owns a
// channel and keeps ALL of it in registers (at most BNF_V vectors per thread): one load phase for every group at once, the per-group
// sums (block reductions of 2 G values), the apply pass from registers, one store phase.  The block applies the G running-statistics
// updates in group order and sums dgamma / dbeta over the groups itself: same results contract as the two-launch form (fixed summation
// order; values differ from it in the last bits only because the partial sums are grouped differently).
constexpr int BNF_T = 512, BNF_V = 10, BNF_G = 4;

// sums of v[0 .. NV-1] over the block; every thread gets the totals (red: NV * (BNF_T / 64) floats)
template <int NV>
__device__ __forceinline__ void block_sum_f(float (&v)[NV], float* red) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k * (BNF_T / 64) + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float t = 0.f;
        for (int j = 0; j < BNF_T / 64; ++j) t += red[k * (BNF_T / 64) + j];
        v[k] = t;
    }
}

// vector j of this thread: element offset and group (-1: past the end).  W = 4: HW % 4 == 0, a vector never straddles two images.
template <int W>
__device__ __forceinline__ bool bnf_slot(int j, int N, int C, int HW, int Ng, int c, size_t& off, int& g) {
    const int per = HW / W, idx = threadIdx.x + j * BNF_T;
    if (idx >= N * per) return false;
    const int n = idx / per, i = idx - n * per;
    off = ((size_t)n * C + c) * HW + (size_t)W * i;
    g = n / Ng;
    return true;
}

template <int W>
__global__ __launch_bounds__(BNF_T) void bn_fused_fwd_kernel(BnFwdArgs a) {
    __shared__ float red[2 * BNF_G * BNF_T / 64];
    const int c = blockIdx.x, C = a.st.C, HW = a.st.HW, G = a.st.G, N = a.st.N, Ng = N / G;
    const float n = (float)Ng * (float)HW;
    const float* x = a.st.x;
    const float gamma = a.gamma[c], beta = a.beta[c];
    Vec<W> xv[BNF_V];
    size_t off[BNF_V];
    int grp[BNF_V];
#pragma unroll
    for (int j = 0; j < BNF_V; ++j) {
        grp[j] = -1;
        if (bnf_slot<W>(j, N, C, HW, Ng, c, off[j], grp[j])) xv[j] = Vec<W>::ld(x + off[j]);
    }
    float mean[BNF_G], rstd[BNF_G];
    if (a.st.training) {
        float K[BNF_G], s[2 * BNF_G];
#pragma unroll
        for (int g = 0; g < BNF_G; ++g) {
            K[g] = g < G ? x[((size_t)g * Ng * C + c) * HW] : 0.f;
            s[2 * g] = 0.f;
            s[2 * g + 1] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < BNF_V; ++j)
#pragma unroll
            for (int g = 0; g < BNF_G; ++g)
                if (grp[j] == g) {
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const float d = xv[j].v[k] - K[g];
                        s[2 * g] += d;
                        s[2 * g + 1] += d * d;
                    }
                }
        block_sum_f(s, red);
#pragma unroll
        for (int g = 0; g < BNF_G; ++g) {
            const float dm = s[2 * g] / n, v0 = s[2 * g + 1] / n - dm * dm;
            const float var = v0 < 0.f ? 0.f : v0;                 // (not fmaxf: a NaN variance stays NaN, as in torch)
            mean[g] = K[g] + dm;
            rstd[g] = rsqrtf(var + a.st.eps);
            if (g < G && threadIdx.x == 0 && a.st.run_mean) {      // the G momentum updates, in group order
                a.st.run_mean[c] = (1.f - a.st.momentum) * a.st.run_mean[c] + a.st.momentum * mean[g];
                a.st.run_var[c] = (1.f - a.st.momentum) * a.st.run_var[c] + a.st.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < BNF_G; ++g) {
            mean[g] = a.st.run_mean[c];
            rstd[g] = rsqrtf(a.st.run_var[c] + a.st.eps);
        }
    }
    if (threadIdx.x == 0) {
        for (int g = 0; g < G; ++g) {
            a.st.save_mean[(size_t)g * C + c] = mean[g < BNF_G ? g : 0];
            a.st.save_rstd[(size_t)g * C + c] = rstd[g < BNF_G ? g : 0];
        }
        if (a.st.training && a.st.n_tracked && c == 0) *a.st.n_tracked += G;
    }
#pragma unroll
    for (int j = 0; j < BNF_V; ++j) {
        if (grp[j] < 0) continue;
        float m = mean[0], r = rstd[0];
#pragma unroll
        for (int g = 1; g < BNF_G; ++g)
            if (grp[j] == g) { m = mean[g]; r = rstd[g]; }
        const float scale = gamma * r, shift = beta - m * scale;
        Vec<W> v = xv[j];
        if (a.res) {
            const auto rv = Vec<W>::ld(a.res + off[j]);
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = fmaf(v.v[k], scale, shift) + rv.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = fmaf(v.v[k], scale, shift);
        }
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < W; ++k) v.v[k] = v.v[k] < 0.f ? 0.f : v.v[k];      // relu(NaN) = NaN, as torch
        }
        v.st(a.y + off[j]);
    }
}

template <int W>
__global__ __launch_bounds__(BNF_T) void bn_fused_bwd_kernel(BnBwdArgs a) {
    __shared__ float red[2 * BNF_G * BNF_T / 64];
    const int c = blockIdx.x, C = a.C, HW = a.HW, G = a.G, N = a.N, Ng = N / G;
    const float n = (float)Ng * (float)HW;
    const float gamma = a.gamma[c], beta = a.beta[c];
    float mean[BNF_G], rstd[BNF_G];
#pragma unroll
    for (int g = 0; g < BNF_G; ++g) {
        mean[g] = a.mean[(size_t)(g < G ? g : 0) * C + c];
        rstd[g] = a.rstd[(size_t)(g < G ? g : 0) * C + c];
    }
    Vec<W> xh[BNF_V], gr[BNF_V];                                  // xhat and the masked gradient of this thread's elements
    size_t off[BNF_V];
    int grp[BNF_V];
    float s[2 * BNF_G];
#pragma unroll
    for (int g = 0; g < 2 * BNF_G; ++g) s[g] = 0.f;
#pragma unroll
    for (int j = 0; j < BNF_V; ++j) {
        grp[j] = -1;
        if (!bnf_slot<W>(j, N, C, HW, Ng, c, off[j], grp[j])) continue;
        float m = mean[0], r = rstd[0];
#pragma unroll
        for (int g = 1; g < BNF_G; ++g)
            if (grp[j] == g) { m = mean[g]; r = rstd[g]; }
        const float scale = gamma * r, shift = beta - m * scale;
        const auto xv = Vec<W>::ld(a.x + off[j]);
        gr[j] = bn_masked_grad<W>(a, off[j], xv, scale, shift);
        if (a.dres) gr[j].st(a.dres + off[j]);
#pragma unroll
        for (int k = 0; k < W; ++k) xh[j].v[k] = (xv.v[k] - m) * r;
#pragma unroll
        for (int g = 0; g < BNF_G; ++g)
            if (grp[j] == g) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    s[2 * g] += gr[j].v[k];
                    s[2 * g + 1] += gr[j].v[k] * xh[j].v[k];
                }
            }
    }
    block_sum_f(s, red);
    if (threadIdx.x == 0) {                                        // parameters are shared by the groups: sums in group order
        float tg = 0.f, tgx = 0.f;
#pragma unroll
        for (int g = 0; g < BNF_G; ++g)
            if (g < G) { tg += s[2 * g]; tgx += s[2 * g + 1]; }
        a.dgamma[c] = tgx;
        a.dbeta[c] = tg;
    }
    if (!a.dx) return;
#pragma unroll
    for (int j = 0; j < BNF_V; ++j) {
        if (grp[j] < 0) continue;
        float r = rstd[0], sg = s[0], sgx = s[1];
#pragma unroll
        for (int g = 1; g < BNF_G; ++g)
            if (grp[j] == g) { r = rstd[g]; sg = s[2 * g]; sgx = s[2 * g + 1]; }
        const float scale = gamma * r, mg = a.training ? sg / n : 0.f, mgx = a.training ? sgx / n : 0.f;
        Vec<W> o;
#pragma unroll
        for (int k = 0; k < W; ++k) o.v[k] = scale * (gr[j].v[k] - mg - xh[j].v[k] * mgx);
        o.st(a.dx + off[j]);
    }
}

// one launch when a whole channel of the batch fits the block's registers and there are enough channels to fill the chip
static inline bool bn_fused_takes(int N, int C, int HW, int G) {
#ifdef SC_BN_NO_FUSED
    return false;
#endif
    const long long vecs = (HW & 3) ? (long long)N * HW : (long long)N * (HW >> 2);
    return C >= 256 && G <= BNF_G && vecs <= (long long)BNF_T * BNF_V;
}

// Blocks per (channel, group); C * G * S ~ 2048 blocks (8 resident blocks of 256 threads per CU).  Round 5: a rule that minimises
// rounds(C G S) x ceil(Ng / S) instead (one image per block for the three-group passes of the view estimator: exactly three rounds) was
// measured and is NOT faster -- 44 -> 46 us, 24 -> 28 us per forward at 96 images: these passes already move 5-6 TB/s, the smaller blocks
// only add prologues.
static inline int bn_splits(int Ng, int C, int G) {
    int S = (2048 + C * G - 1) / (C * G);
    if (S > Ng) S = Ng;
    if (S > 32) S = 32;
    return S < 1 ? 1 : S;
}

}  // namespace sc

static inline bool bn_bad(int N, int C, int HW, int G) { return G < 1 || N % G != 0; }

extern "C" int sc_bn_splits(int N, int C) { return sc::bn_splits(N, C, 1); }

extern "C" int sc_bn_act_forward(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                 float* save_mean, float* save_rstd, float* run_mean, float* run_var,
                                 int64_t* n_tracked, float* partial, int N, int C, int HW, int relu, int training,
                                 int groups, float eps, float momentum, void* stream_) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if (bn_bad(N, C, HW, groups)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    sc::BnFwdArgs a{{x, partial, save_mean, save_rstd, run_mean, run_var, n_tracked, N, C, HW, groups, training, eps, momentum},
                    res, gamma, beta, y, relu};
    if (sc::bn_fused_takes(N, C, HW, groups)) {
        if (HW & 3) hipLaunchKernelGGL(sc::bn_fused_fwd_kernel<1>, dim3(C), dim3(sc::BNF_T), 0, st, a);
        else hipLaunchKernelGGL(sc::bn_fused_fwd_kernel<4>, dim3(C), dim3(sc::BNF_T), 0, st, a);
        return (int)hipGetLastError();
    }
    const dim3 grid(C, groups * sc::bn_splits(N / groups, C, groups));
    if (training) hipLaunchKernelGGL(sc::bn_stats_kernel, grid, dim3(sc::BN_T), 0, st, x, N, C, HW, groups, partial);
    hipLaunchKernelGGL(sc::bn_apply_fwd_kernel, grid, dim3(sc::BN_T), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int sc_bn_act_backward(const float* dy, const float* x, const float* y, const float* gamma,
                                  const float* beta, const float* mean, const float* rstd, float* partial, float* dx,
                                  float* dres, float* dgamma, float* dbeta, int N, int C, int HW, int relu,
                                  int training, int groups, void* stream_) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if (bn_bad(N, C, HW, groups)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    sc::BnBwdArgs a{dy, x, y, gamma, beta, mean, rstd, partial, dx, dres, dgamma, dbeta, N, C, HW, groups, relu, training,
                    (y != nullptr) ? 1 : 0};
    if (sc::bn_fused_takes(N, C, HW, groups)) {
        if (HW & 3) hipLaunchKernelGGL(sc::bn_fused_bwd_kernel<1>, dim3(C), dim3(sc::BNF_T), 0, st, a);
        else hipLaunchKernelGGL(sc::bn_fused_bwd_kernel<4>, dim3(C), dim3(sc::BNF_T), 0, st, a);
        return (int)hipGetLastError();
    }
    const dim3 grid(C, groups * sc::bn_splits(N / groups, C, groups));
    hipLaunchKernelGGL(sc::bn_bwd_stats_kernel, grid, dim3(sc::BN_T), 0, st, a);
    hipLaunchKernelGGL(sc::bn_bwd_apply_kernel, grid, dim3(sc::BN_T), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int sc_bn_relu_pool_forward(const float* x, const float* gamma, const float* beta, float* y, int* idx,
                                       float* save_mean, float* save_rstd, float* run_mean, float* run_var,
                                       int64_t* n_tracked, float* partial, int N, int C, int H, int W, int training,
                                       int groups, float eps, float momentum, void* stream_) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (bn_bad(N, C, H * W, groups)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    const dim3 grid(C, groups * sc::bn_splits(N / groups, C, groups));
    if (training) hipLaunchKernelGGL(sc::bn_stats_kernel, grid, dim3(sc::BN_T), 0, st, x, N, C, H * W, groups, partial);
    sc::PoolFwdArgs a{{x, partial, save_mean, save_rstd, run_mean, run_var, n_tracked, N, C, H * W, groups, training, eps, momentum},
                      gamma, beta, y, idx, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
#ifndef SC_POOL_ROWS
#define SC_POOL_ROWS 4
#endif
    if ((W & 1) == 0 && SC_POOL_ROWS > 0) hipLaunchKernelGGL(sc::bn_relu_pool_fwd_rows_kernel<(SC_POOL_ROWS > 0 ? SC_POOL_ROWS : 1)>, grid, dim3(sc::BN_T), 0, st, a);
    else if ((W & 1) == 0) hipLaunchKernelGGL(sc::bn_relu_pool_fwd2_kernel, grid, dim3(sc::BN_T), 0, st, a);
    else hipLaunchKernelGGL(sc::bn_relu_pool_fwd_kernel, grid, dim3(sc::BN_T), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int sc_bn_relu_pool_backward(const float* dy, const int* idx, const float* x, const float* gamma,
                                        const float* beta, const float* mean, const float* rstd, float* partial,
                                        float* dx, float* dgamma, float* dbeta, int N, int C, int H, int W,
                                        int training, int groups, void* stream_) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (bn_bad(N, C, H * W, groups)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    const dim3 grid(C, groups * sc::bn_splits(N / groups, C, groups));
    sc::PoolBwdArgs a{dy, idx, x, gamma, beta, mean, rstd, partial, dx, N, C, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, groups, dgamma, dbeta, training};
#ifndef SC_POOL_BWD_ROWS
#define SC_POOL_BWD_ROWS 1
#endif
    if ((W & 1) == 0 && SC_POOL_BWD_ROWS > 0) {      // two gather passes over strips of 2 x 2 blocks
        hipLaunchKernelGGL((sc::bn_relu_pool_bwd_rows_kernel<1, (SC_POOL_BWD_ROWS > 0 ? SC_POOL_BWD_ROWS : 1)>), grid, dim3(sc::BN_T), 0, st, a);
        hipLaunchKernelGGL((sc::bn_relu_pool_bwd_rows_kernel<2, (SC_POOL_BWD_ROWS > 0 ? SC_POOL_BWD_ROWS : 1)>), grid, dim3(sc::BN_T), 0, st, a);
        return (int)hipGetLastError();
    }
    if ((W & 1) == 0) {      // two gather passes, no full-resolution temporary
        hipLaunchKernelGGL((sc::bn_relu_pool_bwd_gather_kernel<true, 1>), grid, dim3(sc::BN_T), 0, st, a);
        hipLaunchKernelGGL((sc::bn_relu_pool_bwd_gather_kernel<true, 2>), grid, dim3(sc::BN_T), 0, st, a);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((sc::bn_relu_pool_bwd_gather_kernel<false, 0>), grid, dim3(sc::BN_T), 0, st, a);
    // pass 2: dx = scale * (g - mean(g) - xhat * mean(g xhat)) in place (g was left in dx)
    sc::BnBwdArgs b{nullptr, x, nullptr, gamma, beta, mean, rstd, partial, dx, dx, dgamma, dbeta, N, C, H * W, groups, 0, training, 0};
    hipLaunchKernelGGL(sc::bn_bwd_apply_kernel, grid, dim3(sc::BN_T), 0, st, b);
    return (int)hipGetLastError();
}
