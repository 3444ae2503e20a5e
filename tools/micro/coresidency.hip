// Can the workgroups of a SECOND kernel (another stream) run on a CU next to one resident persistent workgroup of a first kernel?
// Kernel A: 256 workgroups x 512 threads (one per CU), dynamic LDS of X KB, REGS live vector registers, a bf16-MFMA loop of ~T us.
// Kernel B: 2048 x 256 threads, HBM streaming (read + write of a 512 MB buffer).  Times: A alone, B alone, both on two streams.
//   tools/micro/coresidency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC>      // NACC 32x32 accumulators = 16 NACC registers
__global__ __launch_bounds__(512, 1) void kernel_a(float* out, int iters) {
    extern __shared__ float lds[];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.f + threadIdx.x * 1e-3f); b[i] = (__bf16)0.5f; }
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float r = lds[(threadIdx.x + 1) & 511];
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) r += acc[i][k];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void kernel_b(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = in[i];
        v.x += 1.f;
        out[i] = v;
    }
}

template <int NACC>
static int run(int lds_kb, int iters, int b_blocks) {
    float* oa; float4 *bi, *bo;
    const size_t n4 = (size_t)(256 << 20) / 16;
    CK(hipMalloc(&oa, 256 * 512 * 4)); CK(hipMalloc(&bi, n4 * 16)); CK(hipMalloc(&bo, n4 * 16));
    CK(hipMemset(bi, 0, n4 * 16));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)kernel_a<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](bool doa, bool dob, int order) -> float {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int k = 0; k < 5; ++k) {
                if (order == 0) { if (doa) hipLaunchKernelGGL(kernel_a<NACC>, dim3(256), dim3(512), lds_kb * 1024, s1, oa, iters);
                                  if (dob) hipLaunchKernelGGL(kernel_b, dim3(b_blocks), dim3(256), 0, s2, bi, bo, n4); }
                else            { if (dob) hipLaunchKernelGGL(kernel_b, dim3(b_blocks), dim3(256), 0, s2, bi, bo, n4);
                                  if (doa) hipLaunchKernelGGL(kernel_a<NACC>, dim3(256), dim3(512), lds_kb * 1024, s1, oa, iters); }
            }
            CK(hipDeviceSynchronize());
            float ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() / 5;
            if (ms < best) best = ms;
        }
        return best;
    };
    const float ta = timeit(true, false, 0), tb = timeit(false, true, 0), tab = timeit(true, true, 0), tba = timeit(true, true, 1);
    printf("A: %3d regs of acc, LDS %3d KB, B grid %4d | A alone %.3f ms, B alone %.3f ms (%.2f TB/s) | A then B enqueued %.3f, B then A %.3f (sum %.3f)\n",
           NACC * 16, lds_kb, b_blocks, ta, tb, 2.0 * n4 * 16 / tb / 1e9, tab, tba, ta + tb);
    hipFree(oa); hipFree(bi); hipFree(bo);
    return 0;
}
#include <chrono>
int main() {
    const int iters = 1500;
    for (int lds : {1, 64, 100, 135, 159}) if (run<4>(lds, iters, 2048)) return 1;
    for (int lds : {1, 135}) if (run<8>(lds, iters / 2, 2048)) return 1;
    for (int lds : {1, 135}) if (run<4>(lds, iters, 512)) return 1;
    for (int lds : {1, 135}) if (run<4>(lds, iters, 256)) return 1;
    return 0;
}
