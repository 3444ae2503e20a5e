// chamfer_common.hpp -- the distance expression and the LDS target streaming shared by the brute-force search (chamfer.hip)
// and the exact grid search's fallback (chamfer_grid.hip).  Everything that includes this file is compiled with floating-point
// contraction OFF from here on: the distance must be the reference's fmaf(dy, dy, dx*dx) + dz*dz, bit for bit (see chamfer.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

constexpr int CH_THREADS = 256;
constexpr int CH_Q = 4;          // queries per lane
constexpr int CH_TCHUNK = 2048;  // targets per LDS chunk (32 KiB as float4)
constexpr int CH_SUB = 16;       // targets per min-only sub-block
constexpr float CH_FAR = 1.0e18f;  // padding coordinate: d ~ 3e36, finite, never wins

typedef float ch_f2 __attribute__((ext_vector_type(2)));

#pragma clang fp contract(off)
__device__ __forceinline__ float dist2(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, zz = dz * dz;
    return __builtin_fmaf(dy, dy, xx) + zz;
}
// the same arithmetic for two targets at once (packed fp32: IEEE results per component, identical to dist2)
__device__ __forceinline__ ch_f2 dist2_pk(ch_f2 tx, ch_f2 ty, ch_f2 tz, float qx, float qy, float qz) {
    const ch_f2 dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const ch_f2 xx = dx * dx, zz = dz * dz;
    return __builtin_elementwise_fma(dy, dy, xx) + zz;
}
// LDS image of a target pair (a, b): {xa, xb, ya, yb} {za, zb, -, -}
struct ChPair { float4 xy, z; };
__device__ __forceinline__ void stage_targets(float4* tgt, const float* t_ptr, int k0, int cnt, int cnt_pad, int tid) {
    for (int j = tid; j < cnt_pad; j += CH_THREADS) {
        float x = CH_FAR, y = CH_FAR, z = CH_FAR;
        if (j < cnt) {
            const float* p = t_ptr + (size_t)(k0 + j) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        float* e = reinterpret_cast<float*>(tgt + (j >> 1) * 2) + (j & 1);
        e[0] = x; e[2] = y; e[4] = z;
    }
}
// running minimum of one query over the CH_SUB targets of sub-block sb (pair images sb .. sb + CH_SUB - 1)
#define CH_MIN_SUBBLOCK(mn)                                                                              \
    _Pragma("unroll") for (int t = 0; t < CH_SUB; t += 2) {                                              \
        const float4 Txy = tgt[sb + t], Tz = tgt[sb + t + 1];                                            \
        const ch_f2 tx = {Txy.x, Txy.y}, ty = {Txy.z, Txy.w}, tz = {Tz.x, Tz.y};                         \
        _Pragma("unroll") for (int q = 0; q < CH_Q; ++q) {                                               \
            const ch_f2 d = dist2_pk(tx, ty, tz, qx[q], qy[q], qz[q]);                                   \
            mn[q] = __builtin_fminf(__builtin_fminf(mn[q], d.x), d.y);   /* v_min3_f32: inputs are finite */ \
        }                                                                                                \
    }

}  // namespace sc
