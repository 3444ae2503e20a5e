// mlp_xch.hpp -- hand-over of 64-channel operands from "chain" waves (MFMA C/D register layout: one lane = one point, 16 channels) to
// "weight-gradient" waves (MFMA operands with K = point) through 4 KiB LDS slots, shared by sdf_bwdw.hip and rgb_bwd.hip's fused kernel.
#pragma once
#include "mlp_tile.hpp"

namespace sc {

// LDS-only barrier: does NOT drain the global loads in flight (the stash prefetches must survive it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= the 16 points of lane group g); every lane ends up with the total
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return v;
}

// ---- chain side: drop a 64-channel operand (C/D register layout, v[4T+r] = channel 16T+4g+r of point p) into a slot ----
// slot layout: float4 chunk index = (p>>2)*64 + (channel ^ (p>>2)), element p&3
__device__ __forceinline__ void xch_write(float* slot, const int (&wr)[4], const float (&v)[ACT_STEPS], float mask) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[wr[r] + 64 * t] = v[4 * t + r] * mask;
}
__device__ __forceinline__ void xch_zero(float* slot, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(slot)[lane + 64 * k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- wgrad side ---------------------------------------------------------------------------------------------------
// fragment of channel tile m: x..w = K-steps 0..3 = points 4kg..4kg+3 of channel 16m + i
__device__ __forceinline__ float4 xch_frag(const float* slot, int rd, int m) {
    return *reinterpret_cast<const float4*>(slot + rd + 64 * m);
}
__device__ __forceinline__ float f4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// positional-encoding operand of this lane (PE column 16c + i <-> step = i>>2, owner group = i&3) at the 4 points of
// its K slot, from the point stash: MODE 1 = E, MODE 2 = eps = Gg_c * dE/dx_c.  A lane's column is sin OR cos of one
// frequency, and d/dx sin = f cos, d/dx cos = -f sin are again a sine with a quarter-turn phase: every entry is
//   amp * sin(2 pi (x * f/2pi + phase))   ->  one v_fma + one v_sin_f32 (argument in revolutions) + one v_mul;
// the raw-coordinate lanes (group 3) select x / 1 / 0 instead.  (VALU instructions are not hidden behind MFMAs on this
// chip, see mlp_tile.hpp: this operand is re-evaluated for 6 of the 11 steps, so it is kept to ~50 instructions.)
template <int MODE>
__device__ __forceinline__ void pe_frags(const float* pts, int i, int kg, bool symmetric, float4 (&out)[3]) {
    const int step = i >> 2, gq = i & 3;
    const bool raw = gq == 3, first = step == 0, iscos = step & 1;
    const float f = raw ? 0.f : (float)(1 << (2 * gq + (step >> 1)));
    const float frev = f * 0.15915494309189535f;                         // f / 2 pi
    // E: sin -> phase 0, cos -> 0.25;  dE/dx: f cos -> 0.25, -f sin -> 0.5
    const float phase = MODE == 1 ? (iscos ? 0.25f : 0.f) : (iscos ? 0.5f : 0.25f);
    const float amp = MODE == 1 ? 1.f : f;
    float o[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 xa = *reinterpret_cast<const float4*>(pts + (4 * kg + s) * 8);        // x0 x1 x2 gam0
        const float4 xb = *reinterpret_cast<const float4*>(pts + (4 * kg + s) * 8 + 4);    // gam1 gam2 valid -
        float x[3] = {xa.x, xa.y, xa.z};
        const float sg0 = symmetric ? (x[0] > 0.f ? 1.f : (x[0] < 0.f ? -1.f : 0.f)) : 1.f;
        if (symmetric) x[0] = fabsf(x[0]);
        // per-point factor: validity (E) or the upstream gradient of d sdf/dx_c times the |x0| chain-rule sign (eps)
        const float k[3] = {MODE == 1 ? xb.z : xa.w * sg0, MODE == 1 ? xb.z : xb.x, MODE == 1 ? xb.z : xb.y};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float trig = amp * __builtin_amdgcn_sinf(__builtin_fmaf(x[c], frev, phase));
            const float rawv = first ? (MODE == 1 ? x[c] : 1.f) : 0.f;
            o[c][s] = k[c] * (raw ? rawv : trig);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
}

// acc[n] += A (this wave's 16 rows) x B[n]^T over the 16 points of one chain tile
template <int N>
__device__ __forceinline__ void outer16(const float4& af, const float4 (&bf)[N], f32x4 (&acc)[N]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = mfma16(f4(af, s), f4(bf[n], s), acc[n]);
}


}  // namespace sc
