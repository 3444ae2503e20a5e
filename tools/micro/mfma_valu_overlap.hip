// Do fp32-input MFMAs (v_mfma_f32_16x16x4_f32) and plain VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// Workgroup = 8 waves (2 per SIMD): waves 0-3 run an MFMA loop, waves 4-7 a VALU (v_fma_f32 / v_exp_f32) loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit0: even waves do MFMA; bit1: odd waves do VALU; KIND 0: f32 MFMA, 1: bf16 MFMA 16x16x32; VK 0: fma, 1: exp
template <int KIND, int VK>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {            // waves 0-3 land on SIMDs 0,2,1,3; waves 4-7 on the same four again: one MFMA + one VALU wave per SIMD
        if (mode & 1) {
            f32x4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            float a = 1.f + threadIdx.x * 1e-6f, b = 0.5f;
            bf16x8 ab, bb;
            for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
                    }
            }
            for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
        }
    } else if (mode & 2) {
        if (mode & 4) __builtin_amdgcn_s_setprio(3);      // VALU wave above the MFMA wave in the issue arbitration
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
        const float c = 1.0001f, d = 0.0003f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (VK == 0) v[i] = __builtin_fmaf(v[i], c, d);
                    else v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.25f;
                }
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND, int VK>
float run(int mode, int iters) {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, VK>), dim3(256), dim3(512), 0, 0, out, 10, mode);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, VK>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}

template <int KIND, int VK>
void report(const char* name) {
    const int iters = 2000;
    const float m = run<KIND, VK>(1, iters), v = run<KIND, VK>(2, iters), b = run<KIND, VK>(3, iters), bp = run<KIND, VK>(7, iters);
    printf("%s: MFMA wave alone %.3f ms (%.1f cycles/MFMA), VALU wave alone %.3f ms (%.2f cycles/VALU instr), both on one SIMD %.3f ms (VALU wave at s_setprio 3: %.3f ms) -> %s\n", name, m,
           m * 1e-3 * 2.4e9 / (iters * 64.0), v, v * 1e-3 * 2.4e9 / (iters * 256.0), b, bp, b > 0.9f * (m + v) ? "SERIALISED (shared pipe)" : (b < 1.15f * (m > v ? m : v) ? "overlapped" : "partial overlap"));
}

int main() {
    report<0, 0>("f32 MFMA 16x16x4  + v_fma_f32");
    report<0, 1>("f32 MFMA 16x16x4  + v_exp_f32");
    report<1, 0>("bf16 MFMA 16x16x32 + v_fma_f32");
    report<1, 1>("bf16 MFMA 16x16x32 + v_exp_f32");
    return 0;
}
