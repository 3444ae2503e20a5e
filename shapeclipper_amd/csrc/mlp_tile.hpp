// mlp_tile.hpp -- per-wavefront MLP tile primitives for gfx950 (CDNA4), fp32 MFMA.
//
// One wave64 owns a tile of 16 sample points and walks the whole MLP chain in registers with
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chain; runs on the matrix pipe at the fp32 peak rate, so
// the VALU stays free for softplus / sin / cos / exp).  Convention ("transposed GEMM"):
//        Y^T[ch, pt] = W[ch, k] * X^T[k, pt]
//   A operand  = weights    : lane l holds W[i = l&15][k = l>>4]         (one float per lane)
//   B operand  = activations: lane l holds X[k = l>>4][pt = l&15]
//   C/D        = 16 ch x 16 pt: lane l, reg r holds ch = 4*(l>>4) + r, pt = l&15
//
// The point of this orientation: C/D register r of lane group g=(l>>4) in channel tile T is
// channel 16T+4g+r of point l&15 -- exactly a legal B operand for K-step 4T+r of the next layer
// when the weight columns are visited in that (permuted) order.  Layer outputs feed the next layer
// with NO cross-lane movement, no LDS round trip, no transposes; only the weight reads (LDS,
// plain row-major, odd row stride) know about the permutation.  Transposed products (W^T q, for
// d(sdf)/dx and all backward passes) read the same LDS image column-wise.
// (A 32x32x2 variant of the same idea needs 2x the registers per lane: the d(sdf)/dx sweep keeps
// sp'(a_l) of five layers live, 160 VGPRs at 32 points vs 80 at 16 -- it spilled; this one fits.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 16;         // points per wave tile
constexpr int HID = 64;        // hidden width (arch.impl_sdf.n_channels / impl_rgb.n_channels)
constexpr int NT = 4;          // 16-channel tiles per 64-channel activation
constexpr int ACT_STEPS = 16;  // K-steps for a 64-channel activation (4 channels per step)
constexpr int PE_STEPS = 12;   // K-steps for the positional encoding: 3 coords x 4 steps
constexpr int PE_COLS = 48;    // 39 real PE columns + 9 zero pads, in "slot order" (see pe_slot_col)

// channel visited by lane group 0 at K-step s of a 64-channel activation (group g visits kp(s)+4g)
__host__ __device__ constexpr int kp(int s) { return 16 * (s >> 2) + (s & 3); }

// Slot order of the positional encoding.  Packed column = 4*(4*c + j) + g  (c coord, j step in 0..3,
// g lane group):
//   g = 0..2 : frequency index m = 2*g + (j>>1), sin if (j&1)==0 else cos
//              -> reference column 3 + 6*m + 3*(j&1) + c           (model/implicit.py:12-34 order)
//   g = 3    : j==0 -> raw coordinate, reference column c;  j>0 -> zero pad (column -1)
// Lane group g of a point therefore evaluates frequencies {2^(2g), 2^(2g+1)} of all three
// coordinates, sin/cos of one argument live in the same lane, and the 13 slots that depend on
// coordinate c are exactly K-steps 4c..4c+3 (the forward-mode PE Jacobian uses that).
__host__ __device__ constexpr int pe_slot_col(int col) {
    const int g = col & 3, cj = col >> 2, c = cj >> 2, j = cj & 3;
    return g < 3 ? 3 + 6 * (2 * g + (j >> 1)) + 3 * (j & 1) + c : (j == 0 ? c : -1);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- products ------------------------------------------------------------------------------------
// Two arithmetics with the same operand layouts, selected at compile time by SC_MLP_SPLIT.  The product is built with 0.  1 is the round-5
// go / no-go experiment (VERDICT r04 next #2 ii): correct at every existing parity bar (44 render / SDF / full-step tests), but SLOWER in
// the real kernels -- training render 6.4 -> 7.6 ms, evaluation render 11.9 -> 14.9 ms, sdf_bwdw 2.58 -> 3.25 ms on one box
// (profiles/r05_mlp_split_ab.txt; tools/build_mlp_variant.sh split -DSC_MLP_SPLIT=1) although the isolated layer is 1.6-1.9x faster
// (profiles/r05_chain_split_micro.txt): per 16-point tile the split adds ~5,900 vector instructions (weights AND activations, every
// product) to kernels whose vector work is already on the critical path, about what the 18 k saved matrix cycles are worth, plus 21-38
// spilled registers.  Pre-split weights would remove 80 % of the added instructions but need 6 instead of 4 bytes per weight in LDS:
// 177 KiB for the SDF image, over the 160 KiB of a CU (sdf_bwdw has 7 KiB to spare).  No-go in this form; a go needs weights streamed
// through LDS per layer, i.e. a different kernel structure.
//   0  v_mfma_f32_16x16x4_f32: an exact fp32 fma chain at the fp32 peak rate (157 TFLOP/s);
//   1  (round 5) every product evaluated from EXACT three-piece bf16 splits of both operands, x = p0 + p1 + p2 (round to nearest,
//      residuals exact), as the six piece products a_p b_q with p + q <= 2 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the
//      arithmetic of the trunk convolutions (conv3x3.hip, VERDICT r02 ruling): what is dropped is below 2^-24 of a product, the error
//      against float64 is that of the fp32 form.  A lane's own eight registers in[8 ks .. 8 ks + 7] ARE its B fragment of K-step ks
//      when K index 8 g + j of that step stands for channel 16 (2 ks + j / 4) + 4 g + j % 4 -- the weights are read from the SAME fp32
//      LDS image in that order and split in registers (tools/micro/chain_split_micro.hip: 1.6-1.9x per 64 x 64 layer, and the split of
//      the weights costs nothing measurable beside pre-split ones: these vector instructions run in the shadow of the matrix pipe).
#ifndef SC_MLP_SPLIT
#define SC_MLP_SPLIT 0
#endif
typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
typedef short mlp_s16x4 __attribute__((ext_vector_type(4)));
typedef short mlp_s16x8 __attribute__((ext_vector_type(8)));

template <int N>
struct MlpPieces {          // three bf16 pieces of N (4 or 8) fp32 values, packed as MFMA fragments
    mlp_bf16x8 p[3];        // (N == 4: the low halves)
};
template <int N>
__device__ __forceinline__ void mlp_split(const float (&v)[N], MlpPieces<N>& o) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const __bf16 h0 = (__bf16)v[j];
        const float r1 = v[j] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        o.p[0][j] = h0, o.p[1][j] = h1, o.p[2][j] = (__bf16)r2;
    }
}
// acc += sum over the 32 (N = 8) or 16 (N = 4) K indices of the step of a * b: six exact piece products, small terms first
template <int N>
__device__ __forceinline__ f32x4 mlp_six(const MlpPieces<N>& a, const MlpPieces<N>& b, f32x4 acc) {
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    // Every piece stays live until the last of the six MFMAs has issued (the empty asm statements below).  Without them hipcc 7.2 lets the
    // register allocator put the DESTINATION of an MFMA on the registers of a piece whose last use it is (seen in rgb_composite_bwd:
    // `v_mfma_f32_16x16x32_bf16 v[18:21], v[48:51], v[18:21], v[22:25]`, D on B): legal for the 16x16 shapes LLVM knows, but the gfx950
    // K = 32 shape still reads its operands after the first pass -- the product came out wrong and different from run to run (round 5,
    // tools/dbg_render_ops.py).  The statements take the accumulator as an in / out operand so that they cannot move above the MFMAs.
    if constexpr (N == 8) {
#pragma unroll
        for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[PA[t]], b.p[PB[t]], acc, 0, 0, 0);
        asm volatile("" : "+v"(acc) : "v"(__builtin_bit_cast(f32x4, a.p[0])), "v"(__builtin_bit_cast(f32x4, a.p[1])), "v"(__builtin_bit_cast(f32x4, a.p[2])),
                     "v"(__builtin_bit_cast(f32x4, b.p[0])), "v"(__builtin_bit_cast(f32x4, b.p[1])), "v"(__builtin_bit_cast(f32x4, b.p[2])));
    } else {
        mlp_s16x4 a4[3], b4[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const mlp_s16x8 a8 = __builtin_bit_cast(mlp_s16x8, a.p[q]), b8 = __builtin_bit_cast(mlp_s16x8, b.p[q]);
            a4[q] = mlp_s16x4{a8[0], a8[1], a8[2], a8[3]};
            b4[q] = mlp_s16x4{b8[0], b8[1], b8[2], b8[3]};
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4[PA[t]], b4[PB[t]], acc, 0, 0, 0);
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        asm volatile("" : "+v"(acc) : "v"(__builtin_bit_cast(f32x2_, a4[0])), "v"(__builtin_bit_cast(f32x2_, a4[1])), "v"(__builtin_bit_cast(f32x2_, a4[2])),
                     "v"(__builtin_bit_cast(f32x2_, b4[0])), "v"(__builtin_bit_cast(f32x2_, b4[1])), "v"(__builtin_bit_cast(f32x2_, b4[2])));
    }
    return acc;
}
// hipcc 7.2 under-counts the wait states between v_mfma_f32_16x16x32_bf16 and a VMEM instruction that READS its result (measured round 5:
// rgb_composite_bwd stored d L / d feature straight out of the last MFMA of a product -- one s_nop 0 and seven unrelated instructions
// after it -- and HBM received the accumulator's previous contents, different from run to run).  Every split product therefore ends
// with explicit wait states; 2 x s_nop 7 = 16 states cover the longest pass count the ISA documents for this hazard class.
#ifndef SC_MLP_SPLIT_NOPS
#define SC_MLP_SPLIT_NOPS 0
#endif
__device__ __forceinline__ void mlp_split_settle() {
#pragma unroll
    for (int i = 0; i < SC_MLP_SPLIT_NOPS; ++i) asm volatile("s_nop 7");
}

// acc[mt] += W[row0 + 16*mt + i][col0 + kp(s) + 4*g] * in[s]       wl = W + (row0+i)*LD + col0 + 4*g
// Row strides are multiples of 4 floats: the weights of the four K-steps 4T..4T+3 of a lane (columns 16T+4g .. +3 of its row)
// are one aligned ds_read_b128 -- a quarter of the LDS instructions (and of their address adds) of the per-MFMA ds_read_b32.
template <int LD, int MT>
__device__ __forceinline__ void mm_act(const float* wl, const float (&in)[ACT_STEPS], f32x4 (&acc)[MT]) {
    static_assert(LD % 4 == 0, "row stride must keep the float4 weight reads aligned");
#if SC_MLP_SPLIT
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        MlpPieces<8> b;
        const float bv[8] = {in[8 * ks], in[8 * ks + 1], in[8 * ks + 2], in[8 * ks + 3], in[8 * ks + 4], in[8 * ks + 5], in[8 * ks + 6], in[8 * ks + 7]};
        mlp_split<8>(bv, b);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 w0 = *reinterpret_cast<const float4*>(wl + mt * 16 * LD + 32 * ks);
            const float4 w1 = *reinterpret_cast<const float4*>(wl + mt * 16 * LD + 32 * ks + 16);
            const float av[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            MlpPieces<8> a;
            mlp_split<8>(av, a);
            acc[mt] = mlp_six<8>(a, b, acc[mt]);
        }
    }
    mlp_split_settle();
#else
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        float4 w[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) w[mt] = *reinterpret_cast<const float4*>(wl + mt * 16 * LD + 16 * T);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float wv = r == 0 ? w[mt].x : (r == 1 ? w[mt].y : (r == 2 ? w[mt].z : w[mt].w));
                acc[mt] = mfma16(wv, in[4 * T + r], acc[mt]);
            }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// acc[mt] += W[row0 + kp(s) + 4*g][col0 + 16*mt + i] * in[s]       wl = W + (row0+4*g)*LD + col0 + i
template <int LD, int MT>
__device__ __forceinline__ void mm_act_t(const float* wl, const float (&in)[ACT_STEPS], f32x4 (&acc)[MT]) {
#if SC_MLP_SPLIT
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        MlpPieces<8> b;
        const float bv[8] = {in[8 * ks], in[8 * ks + 1], in[8 * ks + 2], in[8 * ks + 3], in[8 * ks + 4], in[8 * ks + 5], in[8 * ks + 6], in[8 * ks + 7]};
        mlp_split<8>(bv, b);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = wl[kp(8 * ks + j) * LD + 16 * mt];
            MlpPieces<8> a;
            mlp_split<8>(av, a);
            acc[mt] = mlp_six<8>(a, b, acc[mt]);
        }
    }
    mlp_split_settle();
#else
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(wl[kp(s) * LD + 16 * mt], in[s], acc[mt]);
#endif
    __builtin_amdgcn_sched_barrier(0);
}
// The same product with the column-wise weight reads software-pipelined one block of 4 K-steps (16 ds_read_b32, merged into 8
// ds_read2_b32) ahead of the MFMAs that consume them, the order pinned by scheduling barriers.  For a wave that has its SIMD to
// itself while it runs a transposed layer (the chain waves of sdf_bwdw in the V sweep: the wgrad partner is parked at a barrier)
// hipcc's own schedule -- one ds_read2_b32, s_waitcnt lgkmcnt(0), two MFMAs, 32 times -- exposes an LDS round trip per MFMA pair
// (measured 5.5 k cycles for the 64 MFMAs = 2 k of a layer).  Kernels with two chain waves per SIMD keep mm_act_t: there the partner
// hides the latency and the 32 extra live registers cost more.
template <int LD, int MT>
__device__ __forceinline__ void mm_act_t_pipe(const float* wl, const float (&in)[ACT_STEPS], f32x4 (&acc)[MT]) {
#if SC_MLP_SPLIT
    // split arithmetic: the eight column-wise reads of the NEXT (K-step, channel tile) are issued before the split + MFMAs of this one
    float cur[8], nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = wl[kp(j) * LD];
    MlpPieces<8> b;
#pragma unroll
    for (int u = 0; u < 2 * MT; ++u) {
        const int ks = u / MT, mt = u % MT;
        if (mt == 0) {
            const float bv[8] = {in[8 * ks], in[8 * ks + 1], in[8 * ks + 2], in[8 * ks + 3], in[8 * ks + 4], in[8 * ks + 5], in[8 * ks + 6], in[8 * ks + 7]};
            mlp_split<8>(bv, b);
        }
        if (u + 1 < 2 * MT) {
            const int ks1 = (u + 1) / MT, mt1 = (u + 1) % MT;
#pragma unroll
            for (int j = 0; j < 8; ++j) nxt[j] = wl[kp(8 * ks1 + j) * LD + 16 * mt1];
        }
        __builtin_amdgcn_sched_barrier(0);
        MlpPieces<8> a;
        mlp_split<8>(cur, a);
        acc[mt] = mlp_six<8>(a, b, acc[mt]);
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    mlp_split_settle();
#else
    float wa[4 * MT], wb[4 * MT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wa[r * MT + mt] = wl[kp(r) * LD + 16 * mt];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        float (&cur)[4 * MT] = (T & 1) ? wb : wa;
        float (&nxt)[4 * MT] = (T & 1) ? wa : wb;
        if (T + 1 < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) nxt[r * MT + mt] = wl[kp(4 * (T + 1) + r) * LD + 16 * mt];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(cur[r * MT + mt], in[4 * T + r], acc[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

// acc[mt] += W[row0 + 16*mt + i][col0 + 4*(S0+s) + g] * in[s]      wl = W + (row0+i)*LD + col0 + g
template <int LD, int MT, int S0, int NS>
__device__ __forceinline__ void mm_pe(const float* wl, const float* in, f32x4 (&acc)[MT]) {
#if SC_MLP_SPLIT
    static_assert(NS == 4 || NS == 12, "one coordinate (16 slots: one 16x16x16 step) or the whole encoding (32 + 16 slots)");
    if constexpr (NS == 12) {
        MlpPieces<8> b;
        const float bv[8] = {in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]};
        mlp_split<8>(bv, b);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = wl[mt * 16 * LD + 4 * (S0 + j)];
            MlpPieces<8> a;
            mlp_split<8>(av, a);
            acc[mt] = mlp_six<8>(a, b, acc[mt]);
        }
    }
    {
        constexpr int S1 = NS == 12 ? 8 : 0;
        MlpPieces<4> b;
        const float bv[4] = {in[S1], in[S1 + 1], in[S1 + 2], in[S1 + 3]};
        mlp_split<4>(bv, b);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = wl[mt * 16 * LD + 4 * (S0 + S1 + j)];
            MlpPieces<4> a;
            mlp_split<4>(av, a);
            acc[mt] = mlp_six<4>(a, b, acc[mt]);
        }
    }
    mlp_split_settle();
#else
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(wl[mt * 16 * LD + 4 * (S0 + s)], in[s], acc[mt]);
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// acc[t][r] = b[kp(4t+r)]   (b already offset by 4*g) -- the bias seeds the accumulator
template <int MT>
__device__ __forceinline__ void acc_init(f32x4 (&acc)[MT], const float* b) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = b[kp(4 * t + r)];
}
template <int MT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT]) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
}

// ---- activations -----------------------------------------------------------------------------------
// softplus(beta=100) and derivatives (model/implicit.py:136; torch threshold=20 is reproduced to
// well below fp32 resolution: beyond it log1p(t) < 2.1e-9*0.01).  t = exp(-|100a|), r = 1/(1+t):
//   sp = max(a,0) + log(1+t)/100,  sp' = (a>=0 ? 1 : t) * r,  sp'' = 100 * t * r^2
// (one multiply by -100 log2(e) with the |.| source modifier + v_exp_f32; max(a, 0) as ONE instruction (relu_f below): fmaxf() costs a second
//  v_max for sNaN quieting -- every vector instruction of these kernels is on the critical path, see below.  NOT inline asm:
//  the hazard recogniser does not see an asm statement reading an MFMA result and omits the wait states -- measured: wrong
//  colours at 1e-2.)
// Round 5: as ONE v_max_i32 on the bit pattern (negative floats are negative integers; -0.0 -> +0.0).  hipcc 7.2 expands the
// v_med3_f32 form into a canonicalising v_max_f32 x, x + v_max_f32 0, x (read from the ISA: two instructions per element).
__device__ __forceinline__ float relu_f(float a) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), 0)); }
__device__ __forceinline__ void softplus_parts(float a, float& t, float& r) {
    t = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(a));
    r = __builtin_amdgcn_rcpf(1.f + t);
}
// log(1+t)/100 with the hardware log2 (v_log_f32, ~1 ulp on [1,2]: absolute error < 1e-9 after the 0.0069 scale); the
// library logf spends ~12 more VALU instructions per element on denormal / accuracy handling this argument range never
// needs -- and on gfx950 VALU instructions do NOT overlap with MFMAs of the same SIMD (tools/micro/mfma_valu_overlap.hip:
// time = MFMA time + VALU time), so every VALU instruction of the chain kernels is on the critical path.
__device__ __forceinline__ float softplus_val(float a, float t) {
    return __builtin_fmaf(0.0069314718056f, __builtin_amdgcn_logf(1.f + t), relu_f(a));
}
__device__ __forceinline__ float softplus_d1(float a, float t, float r) { return (a >= 0.f ? 1.f : t) * r; }
__device__ __forceinline__ float softplus_d2(float t, float r) { return 100.f * t * r * r; }

// ---- what the kernels park between their sweeps (training stash `stash_a`, per-wave scratch) -----------------------------------------------
// SC_STASH_H = 0: the pre-activation a_l; every reader re-evaluates softplus and its derivatives from it (exp2 + rcp, + log for the value).
// SC_STASH_H = 1 (round 5): the ACTIVATION h_l = softplus(a_l).  With u = exp(-100 h):  exp(100 h) = 1 + exp(100 a), so
//     sp'(a) = sigmoid(100 a) = 1 - u,      sp''(a) = 100 sp' (1 - sp') = 100 (1 - u) u,      sp(a) = h (no evaluation at all):
// ONE transcendental per element and sweep instead of two or three.  1 - u carries an absolute error of one fp32 ulp of 1 (6e-8): for
// strongly negative a (sp' < 1e-4) that is a large RELATIVE error of a factor that multiplies a negligible term -- the parity bars
// (2e-4 of the largest gradient) do not move (tests/test_gpu_parity_large.py, test_gpu_full_step_parity.py).
#ifndef SC_STASH_H
#define SC_STASH_H 1
#endif
__device__ __forceinline__ float stash_of(float a, float h) { return SC_STASH_H ? h : a; }
__device__ __forceinline__ void stash_parts(float v, float& t, float& r) {
#if SC_STASH_H
    t = __builtin_amdgcn_exp2f(-144.26950408889634f * v);          // u = exp(-100 h)  (h >= 0)
    r = 1.f - t;                                                   // sp'(a)
#else
    softplus_parts(v, t, r);
#endif
}
__device__ __forceinline__ float stash_d1(float v, float t, float r) { return SC_STASH_H ? r : softplus_d1(v, t, r); }
__device__ __forceinline__ float stash_d2(float t, float r) { return SC_STASH_H ? 100.f * t * r : softplus_d2(t, r); }
__device__ __forceinline__ float stash_val(float v, float t) { return SC_STASH_H ? v : softplus_val(v, t); }

// ---- positional encoding in slot order ---------------------------------------------------------------
// e[4c+j]: value, d1[4c+j]: d/dx_c, d2[4c+j]: d2/dx_c^2 of the slot this lane (group g) owns.
// Symmetry (implicit.py:139-145): coordinate 0 enters as |x0|; chain-rule factor sign(x0) with
// sign(0)=0 exactly as torch.abs' backward; the second derivative carries sign^2.
// FAST: hardware sin/cos (|error| ~ 1e-6 for these |arguments| <= 64) -- used by the backward kernels, whose outputs are
// gradients accumulated over many points; the forward value path keeps the accurate sincosf.
template <bool D1, bool D2, bool FAST = false>
__device__ __forceinline__ void pe_slots(float x0, float x1, float x2, int g, bool symmetric,
                                         float (&e)[PE_STEPS], float (&d1)[PE_STEPS], float (&d2)[PE_STEPS]) {
    const float fbase = g == 0 ? 1.f : (g == 1 ? 4.f : 16.f);
    const float xs[3] = {symmetric ? fabsf(x0) : x0, x1, x2};
    const float sg0 = symmetric ? (x0 > 0.f ? 1.f : (x0 < 0.f ? -1.f : 0.f)) : 1.f;
    const bool raw = g == 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sg = c == 0 ? sg0 : 1.f;
        const float sg2 = sg * sg;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float f = fbase * (float)(1 << m);
            float sn, cs;
#ifdef SC_NO_FAST_TRIG                       // experiment switch: accurate sincosf everywhere (tools/ab_fast_trig.sh)
            sincosf(xs[c] * f, &sn, &cs);
#else
#ifdef SC_FORCE_FAST_TRIG                    // timing experiment: what the accurate sincosf of the forward kernels costs
            __sincosf(xs[c] * f, &sn, &cs);
#else
            if (FAST) __sincosf(xs[c] * f, &sn, &cs);
            else sincosf(xs[c] * f, &sn, &cs);
#endif
#endif
            e[4 * c + 2 * m] = raw ? (m == 0 ? xs[c] : 0.f) : sn;
            e[4 * c + 2 * m + 1] = raw ? 0.f : cs;
            if (D1) {
                d1[4 * c + 2 * m] = raw ? (m == 0 ? sg : 0.f) : f * cs * sg;
                d1[4 * c + 2 * m + 1] = raw ? 0.f : -f * sn * sg;
            }
            if (D2) {
                d2[4 * c + 2 * m] = raw ? 0.f : -f * f * sn * sg2;
                d2[4 * c + 2 * m + 1] = raw ? 0.f : -f * f * cs * sg2;
            }
        }
    }
}

// ---- tile-blocked activation layout (TBL64) ----------------------------------------------------------
// A 64-channel per-point tensor is stored as float4 blocks  [tile][ch/4 = 16][pt = 16][4]
// (1024 floats per 16-point tile).  In C/D layout a lane holds 4 consecutive channels per
// accumulator, so a tile is written/read with 4 fully coalesced dwordx4 accesses per lane, and the
// weight-gradient kernel restages the same image through LDS.  v[4T+r] <-> channel 16T + 4g + r.
// Addressing: a wave owns one tile, so `tile` is wave-uniform.  Saying so (readfirstlane) and going through a buffer resource
// (base = the 4 KiB tile in 4 SGPRs built by a handful of scalar instructions, ONE per-lane 32-bit offset shared by every tensor,
// the quarter of the tile as the instruction's immediate) keeps all of the addressing out of the vector registers.  With plain
// pointers hipcc held a 64-bit VGPR pointer per tensor and layer across the tile loop: 75 spilled registers in sdf_bwdw, and
// every scratch reload is followed by `s_waitcnt vmcnt(0)`, which drains the stash prefetches (measured: tools/prof_bwdw.py).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tbl_rsrc(const float* base, int tile) {
    const char* tb = reinterpret_cast<const char*>(base) + (size_t)__builtin_amdgcn_readfirstlane(tile) * 4096;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb), 0, 4096, 0x00020000);       // raw buffer, 4096 bytes, gfx9 dword 3
}
__device__ __forceinline__ void tbl_store(float* base, int tile, int p, int g, const float (&v)[ACT_STEPS]) {
    const __amdgpu_buffer_rsrc_t rs = tbl_rsrc(base, tile);
    const int lo = (g * 16 + p) * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 q = {v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q), rs, lo, 1024 * t, 0);
    }
    // gfx950 hazard that hipcc 7.2 does not cover (measured: tools/micro/store_war_hazard.hip, profiles/r05_store_hazard.txt): a VALU
    // instruction issued DIRECTLY behind a buffer_store_dwordx4 may overwrite the store's data registers before the store has read them --
    // 3e-4 of the stored values with an SGPR in the soffset field (as here), 6e-2 with an immediate; ONE instruction in between is enough
    // for the SGPR form.  LLVM's hazard recogniser inserts wait states for the immediate form only (GCNHazardRecognizer::createsVALUHazard:
    // "this hazard only exists if the instruction is not using a register in the soffset field").  Found when the training instance of
    // sdf_fwd stored a wrong p0: `buffer_store_dwordx4 v[90:93], ..., s74 offen` / `v_mul_f32 v90, v91, v1`.  The last store of a group is
    // therefore followed by one pinned s_nop; tools/scan_store_hazard.py checks the compiler's output of every kernel file for the pattern
    // (tests/test_store_hazard_scan.py).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// tbl_store with the four stores pinned back to back: in the round-6 split RGB forward the scheduler moved the bf16 split of the SAME
// values (v_and / v_sub written in place) between the stores of a group -- the hazard above on stores 2 and 3 of 4 (parked activations wrong
// in register 1 of a quad, different from run to run; tools/scan_store_hazard.py reported the four places).  Nothing may sit between the
// stores, and the pinned s_nop follows the last one.
__device__ __forceinline__ void tbl_store_pinned(float* base, int tile, int p, int g, const float (&v)[ACT_STEPS]) {
    const __amdgpu_buffer_rsrc_t rs = tbl_rsrc(base, tile);
    const int lo = (g * 16 + p) * 16;
    f32x4 q[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) q[t] = f32x4{v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
    // (the values must EXIST before the barrier: without the empty asm statements LLVM sinks the pure instructions that produce them -- the
    //  ReLU's v_max -- down to their use, i.e. between the stores, and the scheduling barrier has nothing to hold back)
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(q[t]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q[t]), rs, lo, 1024 * t, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void tbl_load(const float* base, int tile, int p, int g, float (&v)[ACT_STEPS]) {
    const __amdgpu_buffer_rsrc_t rs = tbl_rsrc(base, tile);
    const int lo = (g * 16 + p) * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, 1024 * t, 0));
        v[4 * t] = q[0]; v[4 * t + 1] = q[1]; v[4 * t + 2] = q[2]; v[4 * t + 3] = q[3];
    }
}
__device__ __forceinline__ void acc_to_regs(const f32x4 (&acc)[NT], float (&v)[ACT_STEPS]) {
#pragma unroll
    for (int s = 0; s < ACT_STEPS; ++s) v[s] = acc[s >> 2][s & 3];
}

// sum over the four lane groups (the groups of a point hold disjoint channel sets)
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// ---- packed weight images ------------------------------------------------------------------------------
// Global (unpadded, row-major) layout handed over by the host, see include/shapeclipper_hip.h.
struct SdfPack {
    static constexpr int W0 = 0;                      // [64][48]   PE slots
    static constexpr int W1 = W0 + 64 * 48;           // [64][112]  cols 0..63 hidden, 64..111 PE slots (pre-scaled 1/sqrt2)
    static constexpr int W2 = W1 + 64 * 112;          // [64][112]
    static constexpr int W3 = W2 + 64 * 112;          // [64][64]
    static constexpr int W4 = W3 + 64 * 64;           // [64][64]
    static constexpr int W5 = W4 + 64 * 64;           // [65][64]   row 0 = sdf, rows 1..64 = feature
    static constexpr int B5 = W5 + 65 * 64;           // [65]
    static constexpr int TOTAL = B5 + 65;             // floats
};
// LDS image: same matrices with row strides = 4 mod 64 floats (16 rows of a ds_read_b128 fall on 16 distinct bank quads;
// column-wise ds_read_b32 is conflict-free for any stride).
struct SdfLds {
    static constexpr int LD0 = 52, LD1 = 116, LD3 = 68;
    static constexpr int W0 = 0;
    static constexpr int W1 = W0 + 64 * LD0;
    static constexpr int W2 = W1 + 64 * LD1;
    static constexpr int W3 = W2 + 64 * LD1;
    static constexpr int W4 = W3 + 64 * LD3;
    static constexpr int W5 = W4 + 64 * LD3;
    static constexpr int B5 = W5 + 65 * LD3;
    static constexpr int TOTAL = B5 + 68;             // floats
};

struct RgbPack {
    static constexpr int V0 = 0;                      // [64][112]  cols 0..47 PE slots, 48..111 sdf feature
    static constexpr int V1 = V0 + 64 * 112;          // [64][64]
    static constexpr int V2 = V1 + 64 * 64;           // [64][64]
    static constexpr int V3 = V2 + 64 * 64;           // [3][64]
    static constexpr int B3 = V3 + 3 * 64;            // [3] (+1 pad)
    static constexpr int TOTAL = B3 + 4;
};
struct RgbLds {
    static constexpr int LD0 = 116, LD1 = 68;
    static constexpr int V0 = 0;
    static constexpr int V1 = V0 + 64 * LD0;
    static constexpr int V2 = V1 + 64 * LD1;
    static constexpr int V3 = V2 + 64 * LD1;          // [3][64] stride 64 (broadcast reads only)
    static constexpr int B3 = V3 + 3 * 64;
    static constexpr int TOTAL = B3 + 4;
};

// copy a [rows][cols] matrix from global (row stride cols) into LDS (row stride ld)
__device__ __forceinline__ void stage_matrix(float* lds, const float* g, int rows, int cols, int ld, int tid, int nthreads) {
    for (int idx = tid; idx < rows * cols; idx += nthreads) {
        const int r = idx / cols, c = idx - r * cols;
        lds[r * ld + c] = g[idx];
    }
}

__device__ __forceinline__ void stage_sdf_weights(float* lds, const float* w, int tid, int nthreads) {
    stage_matrix(lds + SdfLds::W0, w + SdfPack::W0, 64, 48, SdfLds::LD0, tid, nthreads);
    stage_matrix(lds + SdfLds::W1, w + SdfPack::W1, 64, 112, SdfLds::LD1, tid, nthreads);
    stage_matrix(lds + SdfLds::W2, w + SdfPack::W2, 64, 112, SdfLds::LD1, tid, nthreads);
    stage_matrix(lds + SdfLds::W3, w + SdfPack::W3, 64, 64, SdfLds::LD3, tid, nthreads);
    stage_matrix(lds + SdfLds::W4, w + SdfPack::W4, 64, 64, SdfLds::LD3, tid, nthreads);
    stage_matrix(lds + SdfLds::W5, w + SdfPack::W5, 65, 64, SdfLds::LD3, tid, nthreads);
    stage_matrix(lds + SdfLds::B5, w + SdfPack::B5, 1, 65, 68, tid, nthreads);
}

__device__ __forceinline__ void stage_rgb_weights(float* lds, const float* w, int tid, int nthreads) {
    stage_matrix(lds + RgbLds::V0, w + RgbPack::V0, 64, 112, RgbLds::LD0, tid, nthreads);
    stage_matrix(lds + RgbLds::V1, w + RgbPack::V1, 64, 64, RgbLds::LD1, tid, nthreads);
    stage_matrix(lds + RgbLds::V2, w + RgbPack::V2, 64, 64, RgbLds::LD1, tid, nthreads);
    stage_matrix(lds + RgbLds::V3, w + RgbPack::V3, 1, 3 * 64 + 4, 3 * 64 + 4, tid, nthreads);
}

}  // namespace sc
