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

// How many target slices for `wgs` workgroups' worth of queries (b * ceil(n / 1024)) against m targets?  A workgroup holds a 32 KiB chunk
// of targets in LDS, so `slots` = 5 per CU run at a time, all of the same length: the launch runs in rounds of `slots` workgroups and takes
// about rounds x (slice length + a fixed cost per workgroup).  100,000 x 100,000 at batch 1 (98 workgroups of queries): 13 slices = 1,274
// workgroups = ONE full round of 7,696 targets, where 20 slices (the old "about 2,048 workgroups" rule) ran two rounds of 5,008 with the
// second one a third full.  Slices stay at least one LDS chunk long and at most 64 per cloud.
__host__ __device__ inline int ch_auto_split(long long wgs, int m, int slots) {
    int best = 1;
    long long best_cost = -1;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && m / s < CH_TCHUNK) break;
        const long long rounds = (wgs * s + slots - 1) / slots;
        const long long cost = rounds * ((m + s - 1) / s + 384);        // 384: query loads, index resolution and the key atomics, in targets
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

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
int chamfer_slots();      // chamfer.hip

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
