// Chip-wide sustained rate of v_mfma_f32_32x32x16_bf16 / f16 from registers (no memory traffic): what the matrix pipe delivers at the clock
// the chip holds under that load.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool H, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 7); y[e] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
            acc[a] = H ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc[a], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool H, int NACC>
static void run(const char* name, int wgs_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000, blocks = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<H, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flop = 2.0 * 32 * 32 * 16 * (double)NACC * iters * 4.0 * blocks;
        printf("%s, %d accumulators, %d waves/SIMD, run %d: %.2f ms  %.0f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", name, NACC, wgs_per_cu, rep, ms,
               flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wgs_per_cu));
    }
    hipFree(out);
}
int main() {
    run<false, 8>("bf16 32x32x16", 1);
    run<false, 8>("bf16 32x32x16", 2);
    run<true, 8>("f16  32x32x16", 2);
    run<false, 4>("bf16 32x32x16", 2);
    return 0;
}
