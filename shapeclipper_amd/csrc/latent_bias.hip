// latent_bias.hip -- per-image biases of the conditioned MLP layers (SURVEY section 7 hard part 2: the latent columns of a layer's weight
// act on the [B, Z] latent once per image instead of on a latent repeated per point, reference model/implicit.py:166, renderer.py:89):
//     c[b][l][ch] = bias[l][ch] + (l < L ? post[l] * sum_k z[b][k] * lat[l * 64 + ch][k] : 0)
// and its reverse pass, one launch each.  Replaces one rocBLAS product + 5 element-wise launches forward and ~10 backward per use (6 uses
// per training step), and makes the result independent of the batch size: every output element is one fixed-order sum over k (rocBLAS
// picked another kernel / summation order per batch size).
#include <hip/hip_runtime.h>

#include "shapeclipper_hip.h"

namespace sc {

__global__ __launch_bounds__(64) void latent_bias_fwd_kernel(const float* __restrict__ z, const float* __restrict__ lat,
                                                              const float* __restrict__ bias, const float* __restrict__ post, float* __restrict__ out,
                                                              int Z, int L, int NL) {
    const int b = blockIdx.x, l = blockIdx.y, ch = threadIdx.x;
    float v = 0.f;
    if (l < L) {
        const float* zr = z + (size_t)b * Z;
        const float* wr = lat + (size_t)(l * 64 + ch) * Z;
        for (int k = 0; k < Z; ++k) v = __builtin_fmaf(zr[k], wr[k], v);
        v *= post ? post[l] : 1.f;
    }
    out[((size_t)b * NL + l) * 64 + ch] = bias[l * 64 + ch] + v;
}

// blocks [0, B): g_z[b][:];  blocks [B, B + 4 NL): layer l = (block - B) / 4, channels 16 q .. 16 q + 15 (q = (block - B) % 4):
// g_bias[l][those] and (l < L) g_lat[l * 64 + those][:].  Loops over the batch / the channels are unrolled with independent loads in flight
// (these kernels are latency-bound: a few KB of work on one workgroup each); the summation ORDER stays the index order.
__global__ __launch_bounds__(256) void latent_bias_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ lat,
                                                               const float* __restrict__ post, float* __restrict__ g_z, float* __restrict__ g_lat,
                                                               float* __restrict__ g_bias, int B, int Z, int L, int NL) {
    const int t = threadIdx.x;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        if (!g_z) return;
        __shared__ float gs[8 * 64];                               // this image's gradient rows of the conditioned layers (L <= 8)
        for (int e = t; e < L * 64; e += 256) gs[e] = g[(size_t)b * NL * 64 + e];
        __syncthreads();
        for (int k = t; k < Z; k += 256) {
            float s = 0.f;
            for (int l = 0; l < L; ++l) {
                const float p = post ? post[l] : 1.f;
                const float* lp = lat + (size_t)l * 64 * Z + k;
                float sl = 0.f;
#pragma unroll 16
                for (int ch = 0; ch < 64; ++ch) sl = __builtin_fmaf(gs[l * 64 + ch], lp[(size_t)ch * Z], sl);
                s += p * sl;
            }
            g_z[(size_t)b * Z + k] = s;
        }
        return;
    }
    const int l = (blockIdx.x - B) >> 2, q = (blockIdx.x - B) & 3;
    if (t < 16) {
        const int ch = 16 * q + t;
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s += g[((size_t)b * NL + l) * 64 + ch];
        g_bias[l * 64 + ch] = s;
    }
    if (l < L) {
        const float p = post ? post[l] : 1.f;
        for (int e = t; e < 16 * Z; e += 256) {
            const int ch = 16 * q + e / Z, k = e % Z;
            float s = 0.f;
#pragma unroll 8
            for (int b = 0; b < B; ++b) s = __builtin_fmaf(g[((size_t)b * NL + l) * 64 + ch], z[(size_t)b * Z + k], s);
            g_lat[(size_t)(l * 64 + ch) * Z + k] = p * s;
        }
    }
}

}  // namespace sc

extern "C" int sc_latent_bias_forward(const float* z, const float* lat, const float* bias, const float* post, float* out, int B, int Z, int L, int NL,
                                      void* stream) {
    if (B <= 0) return 0;
    if (Z <= 0 || L < 0 || NL < L || NL <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sc::latent_bias_fwd_kernel, dim3(B, NL), dim3(64), 0, (hipStream_t)stream, z, lat, bias, post, out, Z, L, NL);
    return (int)hipGetLastError();
}

extern "C" int sc_latent_bias_backward(const float* g, const float* z, const float* lat, const float* post, float* g_z, float* g_lat, float* g_bias,
                                       int B, int Z, int L, int NL, void* stream) {
    if (B <= 0 || Z <= 0 || L < 0 || L > 8 || NL < L || NL <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sc::latent_bias_bwd_kernel, dim3(B + 4 * NL), dim3(256), 0, (hipStream_t)stream, g, z, lat, post, g_z, g_lat, g_bias, B, Z, L, NL);
    return (int)hipGetLastError();
}
