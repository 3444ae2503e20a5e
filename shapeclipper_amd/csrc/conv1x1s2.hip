// The 1x1 / stride 2 shortcut convolutions of the ResNet trunks (torchvision BasicBlock.downsample[0] behind model/graph.py:50-54 and
// model/view_estimator.py:40-42): 64 -> 128 at 56 x 56, 128 -> 256 at 28 x 28, 256 -> 512 at 14 x 14.  1.2 GFLOP each: HBM- and
// latency-bound, MIOpen spends 45-90 us on each of forward / backward-data / backward-weight (12-25 TFLOP/s,
// profiles/r02_conv_layers.txt).  Three small fp32 kernels on v_mfma_f32_32x32x2_f32, NCHW:
//   forward        out[b][co][p]      = sum_ci w[co][ci] x[b][ci][2y][2x]                       D[co][pixel], K = ci
//   backward-data  gx[b][ci][2y][2x]  = sum_co w[co][ci] gy[b][co][p], zero at odd rows / columns (written here: no memset pass)
//   backward-weight dw[co][ci]        = sum_{b,p} gy[b][co][p] x[b][ci][2y][2x]                 D[co][ci], K = pixel, split over workgroups
// The two GEMMs with pixel columns share one kernel (64 x 128 tiles, K-steps of 16 staged through LDS, lanes = consecutive pixels so
// that loads and stores are contiguous runs); the weight gradient keeps its operands pixel-contiguous in LDS, one ds_read_b128 per
// operand and 4 MFMAs (k pair = (pixel, pixel + 4)), and sums its partial blocks in range order (fixed summation order).
#include <hip/hip_runtime.h>

#include "grid_cus.hpp"
#include "shapeclipper_hip.h"

namespace sc {

typedef float ds_f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: forward (M = cout, K = cin, B operand = x at even positions); MODE 1: backward-data (M = cin, K = cout, B operand = gy)
template <int MODE>
__global__ __launch_bounds__(256) void conv1x1s2_gemm_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             float* __restrict__ out, int batch, int cin, int cout, int hin) {
    constexpr int KS = 16, ALD = 96, BLD = 128;
    __shared__ float As[KS * ALD];          // [k][m]: k and k + 1 on different bank halves
    __shared__ float Bs[KS * BLD];          // [k][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int hout = hin >> 1, hwo = hout * hout, hwi = hin * hin, npix = batch * hwo;
    const int M = MODE == 0 ? cout : cin, K = MODE == 0 ? cin : cout;
    const int ntm = M / 64;
    const int m0 = (blockIdx.x % ntm) * 64, p0 = (blockIdx.x / ntm) * 128;

    // this thread's staging slots.  A: element (k, m): forward w[(m0 + m) cin + k] (4 consecutive k), backward w[k cin + m0 + m]
    const int am = tid & 63, ak = tid >> 6;
    // B: pixel column tid & 127, rows (tid >> 7) + 2 i
    const int bp = min(p0 + (tid & 127), npix - 1), bb = bp / hwo, bpo = bp - bb * hwo, byo = bpo / hout, bxo = bpo - byo * hout;
    const float* bsrc = MODE == 0 ? in + (size_t)bb * cin * hwi + 2 * byo * hin + 2 * bxo : in + (size_t)bb * cout * hwo + bpo;
    const int bstride = MODE == 0 ? hwi : hwo;

    ds_f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;

    // the operands of K-step k0 + KS are fetched into registers while the MFMAs of step k0 run (round 4: each of the 4-16 K-steps used
    // to expose a global-load round trip between two barriers -- these launches are latency-bound, not bandwidth-bound)
    float av[4], bv[8];
    auto fetch = [&](int k0) {
        if (MODE == 0) {
            const float4 t = *reinterpret_cast<const float4*>(w + (size_t)(m0 + am) * cin + k0 + 4 * ak);
            av[0] = t.x, av[1] = t.y, av[2] = t.z, av[3] = t.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = w[(size_t)(k0 + ak + 4 * i) * cin + m0 + am];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = bsrc[(size_t)(k0 + (tid >> 7) + 2 * i) * bstride];
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KS) {
        __syncthreads();                                      // every wave is done with the previous K-step's tiles
#pragma unroll
        for (int i = 0; i < 4; ++i) As[(MODE == 0 ? 4 * ak + i : ak + 4 * i) * ALD + am] = av[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) Bs[((tid >> 7) + 2 * i) * BLD + (tid & 127)] = bv[i];
        __syncthreads();
        if (k0 + KS < K) fetch(k0 + KS);
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) {
            const float b = Bs[(2 * j + half) * BLD + 32 * wave + (lane & 31)];
            const float a0 = As[(2 * j + half) * ALD + (lane & 31)], a1 = As[(2 * j + half) * ALD + 32 + (lane & 31)];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        }
    }

    // acc[r] is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    const int p = p0 + 32 * wave + (lane & 31);
    if (p >= npix) return;
    const int b = p / hwo, po = p - b * hwo, yo = po / hout, xo = po - yo * hout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (MODE == 0) {
                out[((size_t)b * cout + m) * hwo + po] = acc[i][r];
            } else {                                          // the 2 x 2 input block of this output pixel: value, 0 / 0, 0
                float* o = out + ((size_t)b * cin + m) * hwi + 2 * yo * hin + 2 * xo;
                *reinterpret_cast<float2*>(o) = make_float2(acc[i][r], 0.f);
                *reinterpret_cast<float2*>(o + hin) = make_float2(0.f, 0.f);
            }
        }
}

// dw block [64 co][64 ci] of workgroup (tile, range s): sum over the pixels of its range
__global__ __launch_bounds__(256) void conv1x1s2_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              float* __restrict__ partial, int batch, int cin, int cout, int hin, int S) {
    constexpr int PK = 64, LD = PK + 4;                       // LD / 4 odd: conflict-free ds_read_b128 across 32 rows
    __shared__ __attribute__((aligned(16))) float Gs[64 * LD];
    __shared__ __attribute__((aligned(16))) float Xs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int hout = hin >> 1, hwo = hout * hout, hwi = hin * hin, npix = batch * hwo;
    const int nti = cin / 64;
    const int tile = blockIdx.x / S, s = blockIdx.x - tile * S;
    const int co0 = (tile / nti) * 64, ci0 = (tile % nti) * 64;
    const int ksteps = (npix + PK - 1) / PK;
    const int k_lo = (int)((long long)s * ksteps / S), k_hi = (int)((long long)(s + 1) * ksteps / S);

    ds_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* Ab = Gs + ((wave >> 1) * 32 + (lane & 31)) * LD + 4 * half;
    const float* Bb = Xs + ((wave & 1) * 32 + (lane & 31)) * LD + 4 * half;

    // (the operands of K-step ks + 1 are fetched while the MFMAs of step ks run, as in the GEMM kernel above)
    float gv[16], xv[16];
    auto fetch = [&](int ks) {
        // staging: pixel column tid & 63, channel rows (tid >> 6) + 4 i
        const int p = ks * PK + (tid & 63);
        const bool ok = p < npix;
        const int pc = ok ? p : npix - 1, b = pc / hwo, po = pc - b * hwo, yo = po / hout, xo = po - yo * hout;
        const float* gsrc = gy + ((size_t)b * cout + co0) * hwo + po;
        const float* xsrc = x + ((size_t)b * cin + ci0) * hwi + 2 * yo * hin + 2 * xo;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (tid >> 6) + 4 * i;
            gv[i] = ok ? gsrc[(size_t)c * hwo] : 0.f;
            xv[i] = ok ? xsrc[(size_t)c * hwi] : 0.f;
        }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int ks = k_lo; ks < k_hi; ++ks) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (tid >> 6) + 4 * i;
            Gs[c * LD + (tid & 63)] = gv[i];
            Xs[c * LD + (tid & 63)] = xv[i];
        }
        __syncthreads();
        if (ks + 1 < k_hi) fetch(ks + 1);
#pragma unroll
        for (int g8 = 0; g8 < PK / 8; ++g8) {                 // k pair of an MFMA: (pixel, pixel + 4)
            const float4 a = *reinterpret_cast<const float4*>(Ab + 8 * g8), bq = *reinterpret_cast<const float4*>(Bb + 8 * g8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc, 0, 0, 0);
        }
    }
    float* dst = partial + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(wave * 16 + r) * 64 + lane] = acc[r];
}

__global__ void conv1x1s2_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int cin, int cout, int S) {
    const int nti = cin / 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)(cout / 64) * nti * 4096) return;
    const int tile = (int)(i >> 12), e = (int)(i & 4095);
    const float* src = partial + (size_t)tile * S * 4096 + e;
    float s0 = 0.f, s1 = 0.f;
    int s = 0;
    for (; s + 1 < S; s += 2) {
        s0 += src[(size_t)s * 4096];
        s1 += src[(size_t)(s + 1) * 4096];
    }
    if (s < S) s0 += src[(size_t)s * 4096];
    const int lane = e & 63, r = (e >> 6) & 15, wave = e >> 10;
    const int co = (tile / nti) * 64 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int ci = (tile % nti) * 64 + (wave & 1) * 32 + (lane & 31);
    dw[(size_t)co * cin + ci] = s0 + s1;
}

static int ds_splits(int cin, int cout) {
    const int s = grid_cus() / ((cin / 64) * (cout / 64));
    return s > 0 ? s : 1;
}
static bool ds_ok(int batch, int cin, int cout, int hin) {
    return batch > 0 && cin > 0 && cout > 0 && cin % 64 == 0 && cout % 64 == 0 && hin > 0 && hin % 2 == 0;
}

}  // namespace sc

extern "C" long long sc_conv1x1s2_wgrad_workspace_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 64 || cout % 64) return -1;
    return (long long)(cin / 64) * (cout / 64) * sc::ds_splits(cin, cout) * 4096;
}

extern "C" int sc_conv1x1s2_forward(const float* x, const float* w, float* out, int batch, int cin, int cout, int hin, void* stream) {
    if (!sc::ds_ok(batch, cin, cout, hin)) return (int)hipErrorInvalidValue;
    const int npix = batch * (hin / 2) * (hin / 2);
    hipLaunchKernelGGL(sc::conv1x1s2_gemm_kernel<0>, dim3((cout / 64) * ((npix + 127) / 128)), dim3(256), 0, (hipStream_t)stream, x, w, out, batch,
                       cin, cout, hin);
    return (int)hipGetLastError();
}

extern "C" int sc_conv1x1s2_backward_data(const float* gy, const float* w, float* gx, int batch, int cin, int cout, int hin, void* stream) {
    if (!sc::ds_ok(batch, cin, cout, hin)) return (int)hipErrorInvalidValue;
    const int npix = batch * (hin / 2) * (hin / 2);
    hipLaunchKernelGGL(sc::conv1x1s2_gemm_kernel<1>, dim3((cin / 64) * ((npix + 127) / 128)), dim3(256), 0, (hipStream_t)stream, gy, w, gx, batch,
                       cin, cout, hin);
    return (int)hipGetLastError();
}

extern "C" int sc_conv1x1s2_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hin,
                                  void* stream) {
    if (!sc::ds_ok(batch, cin, cout, hin)) return (int)hipErrorInvalidValue;
    const int tiles = (cin / 64) * (cout / 64), S = sc::ds_splits(cin, cout);
    hipLaunchKernelGGL(sc::conv1x1s2_wgrad_kernel, dim3(tiles * S), dim3(256), 0, (hipStream_t)stream, gy, x, workspace, batch, cin, cout, hin, S);
    const long long n = (long long)tiles * 4096;
    hipLaunchKernelGGL(sc::conv1x1s2_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, dw, cin,
                       cout, S);
    return (int)hipGetLastError();
}
