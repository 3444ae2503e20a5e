// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 (the instruction of the MLP chain kernels) with NACC independent
// accumulators per wave and W waves per SIMD, operands in registers (no LDS, no memory).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
    const int cus = 256, threads = 64 * 4 * waves_per_simd, iters = 2000;
    float* out;
    hipMalloc(&out, (size_t)cus * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, out, 10, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * (threads / 64) * iters * 16 * NACC * 2048.0;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s\n", NACC, waves_per_simd, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<8>(1); run<8>(2);
    return 0;
}
