// What the matrix pipes SUSTAIN on this chip on random operands, from registers (no memory traffic): v_mfma_f32_32x32x16_bf16 (the split
// convolutions, the CLIP GEMMs) and v_mfma_f32_16x16x4_f32 (the MLP chain kernels), every SIMD busy (2 waves each), ~30 ms per measurement.
// The nominal peaks (2.5 PFLOP/s, 157.3 TFLOP/s) assume 2.4 GHz; under these loads the chip holds less.   tools/micro/mfma_sustained.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(unsigned s) { return (float)(hash(s) & 0xffff) / 32768.f - 1.f; }

template <bool RANDOM>
__global__ __launch_bounds__(512) void k_bf16(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x[4], y[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) {
        x[q][e] = (__bf16)(RANDOM ? rnd(threadIdx.x * 64 + q * 8 + e) : 1.f);
        y[q][e] = (__bf16)(RANDOM ? rnd(blockIdx.x * 977 + threadIdx.x * 64 + 32 + q * 8 + e) * 0.01f : 0.f);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[q], y[(q + a) & 3], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][7];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// 16x16x32 (a quarter of the C registers per FLOP) and the f16 flavour of 32x32x16 (the CLIP GEMMs), random operands
template <int KIND>
__global__ __launch_bounds__(512) void k_alt(float* out, int iters) {
    f32x4 acc4[8];
    f32x16 acc[4];
    for (int a = 0; a < 8; ++a) acc4[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x[4], y[4];
    f16x8 xh[4], yh[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) {
        const float u = rnd(threadIdx.x * 64 + q * 8 + e), v = rnd(blockIdx.x * 977 + threadIdx.x * 64 + 32 + q * 8 + e) * 0.01f;
        x[q][e] = (__bf16)u; y[q][e] = (__bf16)v; xh[q][e] = (_Float16)u; yh[q][e] = (_Float16)v;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (KIND == 0) {
#pragma unroll
                for (int a = 0; a < 8; ++a) acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[q], y[(q + a) & 3], acc4[a], 0, 0, 0);
            } else if (KIND == 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[q], yh[(q + a) & 3], acc[a], 0, 0, 0);
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[q], yh[(q + a) & 3], acc4[a], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) s += acc4[a][0];
    for (int a = 0; a < 4; ++a) s += acc[a][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <bool RANDOM>
__global__ __launch_bounds__(512) void k_f32(float* out, int iters) {
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x[8], y[8];
    for (int q = 0; q < 8; ++q) { x[q] = RANDOM ? rnd(threadIdx.x * 16 + q) : 1.f; y[q] = RANDOM ? rnd(blockIdx.x * 977 + threadIdx.x * 16 + 8 + q) * 0.01f : 0.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[(q + a) & 7], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <class K>
static void run(const char* name, K kern, double flop_per_wave_iter, double nominal_tf, int iters) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double tf = flop_per_wave_iter * iters * 256 * 8 / (ms * 1e-3) / 1e12;
        if (rep) printf("%-46s %7.2f ms  %8.1f TFLOP/s = %.2f of the nominal %.1f\n", name, ms, tf, tf / nominal_tf, nominal_tf);
    }
    hipFree(out);
}
int main() {
    run("v_mfma_f32_32x32x16_bf16, constant operands", k_bf16<false>, 16.0 * 32 * 32 * 16 * 2, 2500.0, 60000);
    run("v_mfma_f32_32x32x16_bf16, random operands", k_bf16<true>, 16.0 * 32 * 32 * 16 * 2, 2500.0, 60000);
    run("v_mfma_f32_16x16x32_bf16, random operands", k_alt<0>, 32.0 * 16 * 16 * 32 * 2, 2500.0, 60000);
    run("v_mfma_f32_32x32x16_f16, random operands", k_alt<1>, 16.0 * 32 * 32 * 16 * 2, 2500.0, 60000);
    run("v_mfma_f32_16x16x32_f16, random operands", k_alt<2>, 32.0 * 16 * 16 * 32 * 2, 2500.0, 60000);
    run("v_mfma_f32_16x16x4_f32, constant operands", k_f32<false>, 64.0 * 16 * 16 * 4 * 2, 157.3, 15000);
    run("v_mfma_f32_16x16x4_f32, random operands", k_f32<true>, 64.0 * 16 * 16 * 4 * 2, 157.3, 15000);
    return 0;
}
