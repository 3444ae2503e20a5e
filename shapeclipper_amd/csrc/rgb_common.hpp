// rgb_common.hpp -- RGB MLP chain + wave-level scan/reduce helpers shared by rgb_fwd.hip / rgb_bwd.hip.
#pragma once
#include "mlp_tile.hpp"

namespace sc {

// per-lane LDS base pointers into the RgbLds image (normal and transposed views)
struct RgbLanePtrs {
    const float *v0e, *v0f, *v1, *v2, *v3, *b3;   // A[i=out][k=in]
    const float *v0ft, *v1t, *v2t;                 // A[i=in][k=out]   (backward)
    __device__ __forceinline__ RgbLanePtrs(const float* lds, int p, int g) {
        v0e = lds + RgbLds::V0 + p * RgbLds::LD0 + g;
        v0f = lds + RgbLds::V0 + p * RgbLds::LD0 + 48 + 4 * g;
        v1 = lds + RgbLds::V1 + p * RgbLds::LD1 + 4 * g;
        v2 = lds + RgbLds::V2 + p * RgbLds::LD1 + 4 * g;
        v3 = lds + RgbLds::V3 + 4 * g;
        b3 = lds + RgbLds::B3;
        v0ft = lds + RgbLds::V0 + 4 * g * RgbLds::LD0 + 48 + p;
        v1t = lds + RgbLds::V1 + 4 * g * RgbLds::LD1 + p;
        v2t = lds + RgbLds::V2 + 4 * g * RgbLds::LD1 + p;
    }
};

__device__ __forceinline__ void relu_from_acc(const f32x4 (&acc)[NT], float (&r)[ACT_STEPS]) {
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s) r[s] = relu_f(acc[s >> 2][s & 3]);
}

// RGBNetwork.forward (model/implicit.py:220-239) for one 16-point tile.
//   r[l] : post-ReLU activations of the three hidden layers (the ReLU mask is r > 0)
//   col  : sigmoid output (3), identical in all four lane groups of a point
__device__ __forceinline__ void rgb_chain(const RgbLanePtrs& L, const float* db, const float (&e)[PE_STEPS],
                                          const float (&f)[ACT_STEPS], float (&r)[3][ACT_STEPS], float (&col)[3]) {
    f32x4 acc[NT];
    acc_init(acc, db);
    mm_pe<RgbLds::LD0, NT, 0, PE_STEPS>(L.v0e, e, acc);
    mm_act<RgbLds::LD0, NT>(L.v0f, f, acc);
    relu_from_acc(acc, r[0]);
    acc_init(acc, db + 64);
    mm_act<RgbLds::LD1, NT>(L.v1, r[0], acc);
    relu_from_acc(acc, r[1]);
    acc_init(acc, db + 128);
    mm_act<RgbLds::LD1, NT>(L.v2, r[1], acc);
    relu_from_acc(acc, r[2]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < ACT_STEPS; ++s) part = __builtin_fmaf(L.v3[j * 64 + kp(s)], r[2][s], part);
        const float yv = group_sum(part) + L.b3[j];
        col[j] = 1.f / (1.f + expf(-yv));
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ float wave_inclusive_scan(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
// sum over lanes with a HIGHER index (exclusive suffix sum)
__device__ __forceinline__ float wave_exclusive_suffix(float v) {
    const int lane = threadIdx.x & 63;
    float s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_down(s, d);
        if (lane + d < 64) s += t;
    }
    return s - v;
}

}  // namespace sc
