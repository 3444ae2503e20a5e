// Go / no-go micro-benchmark (VERDICT r04 next #2 ii): the 64 -> 64 layers of the MLP chain kernels (csrc/mlp_tile.hpp) on
// v_mfma_f32_16x16x32_bf16 with the exact three-piece operand split that won 1.7x in conv3x3.hip, against the fp32 form
// (v_mfma_f32_16x16x4_f32) the kernels use.  Same structure as chain_micro.hip: one wave owns a 16-point tile, L dependent layers,
// weights read from LDS, activations never leave registers (the C/D register r of lane group g is channel 16 T + 4 g + r of point
// lane & 15; with the K order of the weights permuted at pack time -- k = 8 g + j of K-step ks <-> channel 16 (2 ks + j / 4) + 4 g + j % 4 --
// a lane's own 8 registers h[8 ks .. 8 ks + 7] ARE its B fragment of K-step ks, so the split needs no data movement).
//   fp32 layer : 64 MFMAs 16x16x4  + 16 ds_read_b128
//   split layer: 16 elements / lane split into 3 bf16 pieces (v_cvt + shift + sub), 4 x 2 x 6 = 48 MFMAs 16x16x32, 24 ds_read_b128 of
//                PRE-SPLIT weights ([piece][mt][ks][lane][8 bf16]: 6 bytes per weight instead of 4 -- the LDS question)
//   split_w    : the same with the weights split ON THE FLY from an fp32 image (what an unchanged 4-byte LDS image would need)
// Checks the split result against the fp32 result (identical up to fp32 accumulation order) before timing.
//   hipcc --offload-arch=gfx950 -O3 -I shapeclipper_amd/csrc -I include tools/micro/chain_split_micro.hip -o chain_split_micro.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "mlp_tile.hpp"
using namespace sc;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8_t& p0, bf16x8_t& p1, bf16x8_t& p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h0 = (__bf16)v[j];
        const float r1 = v[j] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        p0[j] = h0, p1[j] = h1, p2[j] = (__bf16)r2;
    }
}

// six exact piece products a_p b_q, p + q <= 2, small terms first
__device__ __forceinline__ f32x4 six(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x4 acc) {
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[t]], b[PB[t]], acc, 0, 0, 0);
    return acc;
}

// weight element of output row `row`, K-step ks, k index k (0..31) in the permuted order: input channel
__host__ __device__ inline int chan_of(int ks, int k) { const int g = k >> 3, j = k & 7; return 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3); }

// MODE 0 fp32; 1 split, pre-split weights; 2 split, weights split on the fly from the fp32 image
template <int MODE, bool SP>
__global__ __launch_bounds__(512) void chain_loop(const float* __restrict__ wsrc, float* out, int iters, int store_h) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = 68;
    // fp32 image [row 64][LD]; pre-split image [piece 3][mt 4][ks 2][lane 64][8 bf16] = 3 * 4 * 2 * 64 * 16 B = 24 KiB per layer
    if (MODE == 1) {
        unsigned short* wl = reinterpret_cast<unsigned short*>(lds);
        for (int e = threadIdx.x; e < 2 * 3 * 4 * 2 * 64 * 8; e += blockDim.x) {
            const int j = e & 7, ln = (e >> 3) & 63, ks = (e >> 9) & 1, mt = (e >> 10) & 3, pc = (e >> 12) % 3, layer = e / (3 * 4096);
            const int row = 16 * mt + (ln & 15), k = 8 * (ln >> 4) + j;
            const float v = wsrc[layer * 4096 + row * 64 + chan_of(ks, k)];
            const __bf16 h0 = (__bf16)v; const float r1 = v - (float)h0; const __bf16 h1 = (__bf16)r1; const __bf16 h2 = (__bf16)(r1 - (float)h1);
            wl[e] = __builtin_bit_cast(unsigned short, pc == 0 ? h0 : (pc == 1 ? h1 : h2));
        }
    } else {
        for (int e = threadIdx.x; e < 2 * 64 * 64; e += blockDim.x) lds[(e >> 12) * 64 * LD + ((e >> 6) & 63) * LD + (e & 63)] = wsrc[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    float h[ACT_STEPS];
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s) h[s] = 0.01f * ((lane * 7 + s * 3) % 23) - 0.1f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            f32x4 acc[NT];
            acc_zero(acc);
            if (MODE == 0) {
                const float* wl = lds + (l & 1) * 64 * LD + p * LD + 4 * g;
                // the kernels' form: k = 4 T' + g' ... (mlp_tile.hpp mm_act): one ds_read_b32 per MFMA
                mm_act<LD, NT>(wl, h, acc);
            } else {
                bf16x8_t b[2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = h[8 * ks + j];
                    split3(v, b[ks][0], b[ks][1], b[ks][2]);
                }
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        bf16x8_t a[3];
                        if (MODE == 1) {
                            const float4* wl = reinterpret_cast<const float4*>(lds) + (l & 1) * 3 * 512;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) a[pc] = __builtin_bit_cast(bf16x8_t, wl[(pc * 4 + mt) * 128 + ks * 64 + lane]);
                        } else {       // fp32 image: this lane's 8 weights of row 16 mt + p sit at channels chan_of(ks, 8 g + j): two float4 reads
                            const float* wr = lds + (l & 1) * 64 * LD + (16 * mt + p) * LD + 4 * g;
                            const float4 w0 = *reinterpret_cast<const float4*>(wr + 16 * (2 * ks));
                            const float4 w1 = *reinterpret_cast<const float4*>(wr + 16 * (2 * ks + 1));
                            const float v[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                            split3(v, a[0], a[1], a[2]);
                        }
                        acc[mt] = six(a, b[ks], acc[mt]);
                    }
            }
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) {
                if (SP) { float t, r; softplus_parts(acc[s >> 2][s & 3], t, r); h[s] = softplus_val(acc[s >> 2][s & 3], t) * softplus_d1(acc[s >> 2][s & 3], t, r); }
                else h[s] = acc[s >> 2][s & 3];
            }
        }
    }
    if (store_h) {
#pragma unroll
        for (int s = 0; s < ACT_STEPS; ++s) out[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * ACT_STEPS + s] = h[s];
    } else {
        float s = 0.f;
        for (int k = 0; k < ACT_STEPS; ++k) s += h[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

static float* d_w;

template <int MODE, bool SP>
double run(int waves_per_simd, std::vector<float>* result) {
    const int cus = 256, threads = 64 * 4 * waves_per_simd, iters = 500;
    float* out;
    hipMalloc(&out, (size_t)cus * threads * ACT_STEPS * 4);
    const int ldsb = MODE == 1 ? 2 * 3 * 4 * 2 * 64 * 16 : 2 * 64 * 68 * 4;
    hipFuncSetAttribute((const void*)chain_loop<MODE, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    if (result) {
        hipLaunchKernelGGL((chain_loop<MODE, SP>), dim3(1), dim3(64), ldsb, 0, d_w, out, 1, 1);
        result->resize(64 * ACT_STEPS);
        hipMemcpy(result->data(), out, 64 * ACT_STEPS * 4, hipMemcpyDeviceToHost);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain_loop<MODE, SP>), dim3(cus), dim3(threads), ldsb, 0, d_w, out, 10, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_loop<MODE, SP>), dim3(cus), dim3(threads), ldsb, 0, d_w, out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0);
    const double flops = (double)cus * (threads / 64) * iters * 4 * 64 * 2048.0;
    printf("%-34s softplus=%d waves/SIMD=%d: %7.1f fp32-equivalent TFLOP/s, %5.0f cycles (at 2.4 GHz) per 64x64 layer and wave\n",
           MODE == 0 ? "fp32 MFMA 16x16x4" : (MODE == 1 ? "bf16x3 split, pre-split weights" : "bf16x3 split, weights split on the fly"), (int)SP,
           waves_per_simd, flops / (ms * 1e-3) / 1e12, cyc);
    hipFree(out);
    return cyc;
}

int main() {
    std::vector<float> w(2 * 4096);
    unsigned s = 12345u;
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 1.5e-4f; }
    hipMalloc(&d_w, w.size() * 4);
    hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> r0, r1, r2;
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, false>(wps, wps == 1 ? &r0 : nullptr);
        run<1, false>(wps, wps == 1 ? &r1 : nullptr);
        run<2, false>(wps, wps == 1 ? &r2 : nullptr);
        run<0, true>(wps, nullptr);
        run<1, true>(wps, nullptr);
        run<2, true>(wps, nullptr);
    }
    double e1 = 0, e2 = 0, m = 0;
    for (size_t i = 0; i < r0.size(); ++i) { e1 = fmax(e1, fabs(r1[i] - r0[i])); e2 = fmax(e2, fabs(r2[i] - r0[i])); m = fmax(m, fabs(r0[i])); }
    printf("4 chained layers, no activation: max |split - fp32| = %.3g (pre-split), %.3g (on the fly); max |fp32| = %.3g -> relative %.2g\n", e1, e2, m, e1 / m);
    return 0;
}
