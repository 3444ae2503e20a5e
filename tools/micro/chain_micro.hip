// Micro-benchmark of the MLP chain pattern of csrc/mlp_tile.hpp: L dependent 64->64 layers per 16-point tile, weights read
// from LDS (one ds_read per MFMA), optional softplus between layers.  Variants: weight read width, waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -I shapeclipper_amd/csrc -I include tools/micro/chain_micro.hip -o /tmp/chain_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include "mlp_tile.hpp"
using namespace sc;

// b128 variant: LD multiple of 4, one ds_read_b128 gives the weights of 4 consecutive K-steps (4T .. 4T+3)
template <int LD>
__device__ __forceinline__ void mm_act128(const float* wl, const float (&in)[ACT_STEPS], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        float4 w[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) w[mt] = *reinterpret_cast<const float4*>(wl + mt * 16 * LD + 16 * T);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const float wv = r == 0 ? w[mt].x : (r == 1 ? w[mt].y : (r == 2 ? w[mt].z : w[mt].w));
                acc[mt] = mfma16(wv, in[4 * T + r], acc[mt]);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// MODE 2: like MODE 0 plus a sched_group_barrier pipeline {1 MFMA, 1 DS read, NV VALU} x 64 over the region
// [element-wise part of layer l | MFMAs of layer l+1]
template <int LD, int NV>
__device__ __forceinline__ void mm_act_il(const float* wl, const float (&in)[ACT_STEPS], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) acc[mt] = mfma16(wl[mt * 16 * LD + kp(s)], in[s], acc[mt]);
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, bool SP>
__global__ void chain_loop(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = MODE == 1 ? 68 : 65;
    for (int e = threadIdx.x; e < 64 * LD * 2; e += blockDim.x) lds[e] = 0.001f * (e % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    const float* wA = lds + p * LD + 4 * g;
    const float* wB = lds + 64 * LD + p * LD + 4 * g;
    float h[ACT_STEPS];
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s) h[s] = 0.01f * (lane + s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            f32x4 acc[NT];
            acc_zero(acc);
            if (MODE == 0) mm_act<LD, NT>((l & 1) ? wB : wA, h, acc);
            else if (MODE == 1) mm_act128<LD>((l & 1) ? wB : wA, h, acc);
            else if (MODE == 2) mm_act_il<LD, 7>((l & 1) ? wB : wA, h, acc);
            else mm_act_il<LD, 5>((l & 1) ? wB : wA, h, acc);
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) {
                if (SP) { float t, r; softplus_parts(acc[s >> 2][s & 3], t, r); h[s] = softplus_val(acc[s >> 2][s & 3], t) * softplus_d1(acc[s >> 2][s & 3], t, r); }
                else h[s] = acc[s >> 2][s & 3];
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < ACT_STEPS; ++k) s += h[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool SP>
void run(int waves_per_simd) {
    const int cus = 256, threads = 64 * 4 * waves_per_simd, iters = 500;
    float* out;
    hipMalloc(&out, (size_t)cus * threads * 4);
    const int ldsb = 64 * 68 * 2 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain_loop<MODE, SP>), dim3(cus), dim3(threads), ldsb, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_loop<MODE, SP>), dim3(cus), dim3(threads), ldsb, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * (threads / 64) * iters * 4 * 64 * 2048.0;
    printf("weights %s softplus=%d waves/SIMD=%d: %.1f TFLOP/s (%.0f cycles per 64-MFMA layer per wave)\n", MODE == 0 ? "ds_read_b32 " : (MODE == 1 ? "ds_read_b128" : (MODE == 2 ? "b32 interleave7" : "b32 interleave5")),
           (int)SP, waves_per_simd, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 4.0));
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) { run<0, false>(w); run<1, false>(w); run<0, true>(w); run<1, true>(w); run<2, true>(w); run<3, true>(w); }
    return 0;
}
