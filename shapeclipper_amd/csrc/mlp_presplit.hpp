// mlp_presplit.hpp -- PRE-SPLIT weight fragments for the exact three-piece bf16 arithmetic of the MLP chain kernels (round 6).
//
// A product W x of the chain kernels (mlp_tile.hpp conventions: weights = MFMA operand A, the 16 points of a tile = operand B) is evaluated
// as six piece products on v_mfma_f32_16x16x32_bf16 / 16x16x16 (mlp_six).  Splitting the WEIGHTS on the fly costs four times the vector
// instructions of splitting the activations (round 5: slower than fp32 MFMAs); here a workgroup splits them ONCE into LDS, in the order the
// MFMAs consume them:
//   FRAGMENT of (K-step, channel tile) = [piece 0..2][lane 0..63][8 bf16]  (3 KiB; K = 16: [piece][lane][4 bf16], 1.5 KiB)
//   lane l holds row 16 mt + (l & 15); its K index 8 (l >> 4) + j of step ks stands for input channel 16 (2 ks + j / 4) + 4 (l >> 4) + j % 4,
//   i.e. a lane's own registers h[8 ks .. 8 ks + 7] of the previous layer's output ARE its B fragment (no cross-lane movement).
// Every piece is one lane-linear KiB: conflict-free ds_read_b128.  The positional encoding (slot order of mlp_tile.hpp: e[s] <-> packed
// column 4 s + g) is one K = 32 step (e[0..7]) and one K = 16 step (e[8..11]).
//
// gfx950 / hipcc 7.2 hazard met here (tools/scan_mfma_shape_hazard.py, tests/test_store_hazard_scan.py): a 16x16x16 MFMA issued directly
// behind the 16x16x32 MFMA that writes its accumulator reads the accumulator too early.  Callers issue ALL K = 32 products of a layer
// before its K = 16 products (several independent MFMAs lie between the two that share an accumulator).
#pragma once
#include "mlp_tile.hpp"

namespace sc {
namespace ps {

constexpr int F32B = 3 * 64 * 16, F16B = 3 * 64 * 8;
constexpr int PE_FRAG = F32B + F16B;             // the encoding's two steps of one channel tile
constexpr int HID_BYTES = 8 * F32B;              // a 64 x 64 block: [ks 2][mt 4] fragments (24 KiB)
constexpr int PE_BYTES = 4 * PE_FRAG;            // a 64 x 48 encoding block: [mt 4] (18 KiB)

__device__ __forceinline__ void split3(float v, __bf16& h0, __bf16& h1, __bf16& h2) {
    h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    h1 = (__bf16)r1;
    h2 = (__bf16)(r1 - (float)h1);
}

// hidden-input part of a layer: W[64][ld], columns c0 .. c0 + 63 -> [ks][mt] K = 32 fragments
__device__ __forceinline__ void stage_hidden(char* dst, const float* __restrict__ W, int ld, int c0, int tid, int nthreads) {
    for (int idx = tid; idx < 2 * 4 * 64 * 8; idx += nthreads) {
        const int j = idx & 7, lane = (idx >> 3) & 63, mt = (idx >> 9) & 3, ks = idx >> 11;
        const int row = 16 * mt + (lane & 15), col = c0 + 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
        __bf16 h[3];
        split3(W[row * ld + col], h[0], h[1], h[2]);
        char* f = dst + (ks * 4 + mt) * F32B + lane * 16 + j * 2;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<__bf16*>(f + p * 1024) = h[p];
    }
}
// encoding part: packed columns c0 .. c0 + 47 -> [mt] (K = 32 fragment of slots 0..7, K = 16 fragment of slots 8..11)
__device__ __forceinline__ void stage_pe(char* dst, const float* __restrict__ W, int ld, int c0, int tid, int nthreads) {
    for (int idx = tid; idx < 4 * 64 * 12; idx += nthreads) {
        const int s = idx % 12, lane = (idx / 12) & 63, mt = idx / (12 * 64);
        const int row = 16 * mt + (lane & 15), col = c0 + 4 * s + (lane >> 4);
        __bf16 h[3];
        split3(W[row * ld + col], h[0], h[1], h[2]);
        char* f = dst + mt * PE_FRAG;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (s < 8) *reinterpret_cast<__bf16*>(f + p * 1024 + lane * 16 + s * 2) = h[p];
            else *reinterpret_cast<__bf16*>(f + F32B + p * 512 + lane * 8 + (s - 8) * 2) = h[p];
        }
    }
}

__device__ __forceinline__ MlpPieces<8> frag32(const char* f, int lane) {
    MlpPieces<8> a;
#pragma unroll
    for (int p = 0; p < 3; ++p) a.p[p] = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(f + p * 1024 + lane * 16));
    return a;
}
__device__ __forceinline__ MlpPieces<4> frag16(const char* f, int lane) {
    MlpPieces<4> a;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const uint2 v = *reinterpret_cast<const uint2*>(f + p * 512 + lane * 8);
        a.p[p] = __builtin_bit_cast(mlp_bf16x8, make_uint4(v.x, v.y, 0u, 0u));
    }
    return a;
}


// The consumers of a layer's accumulators (vector instructions, stores) must not issue before the LAST MFMA chain has written them:
// hipcc 7.2 under-counts the passes of the gfx950 K = 32 shape when it inserts the wait states between an MFMA and a vector / memory
// instruction that reads its result (mlp_tile.hpp, round 5: a store received the accumulator's previous contents).  Seen here in round 6
// as parked ReLU activations that were NEGATIVE and differed from run to run: v_max read the accumulator early and wrote its result in
// place, then the MFMA's write landed on top of it.  Every product part therefore ends with explicit wait states (2 x s_nop 7 = 16,
// the longest documented requirement of this hazard class); the cost is 16 cycles per part against ~1,000 cycles of MFMAs.
__device__ __forceinline__ void mfma_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// acc[u][mt] += W_e e for TPW tiles (fragments [mt] at `base`): all K = 32 products, then all K = 16 products (see the hazard above).
// The fragment reads are software-pipelined ONE fragment ahead and pinned there (scheduling barriers): left alone, hipcc hoists the reads
// of all the fragments of a part in front of its first MFMA -- 96 live registers in the streamed forward kernel, 91 of them spilled.
template <int TPW>
__device__ __forceinline__ void pe_part(const char* base, int lane, const MlpPieces<8> (&e32)[TPW], const MlpPieces<4> (&e16)[TPW], f32x4 (&acc)[TPW][NT]) {
    MlpPieces<8> w32 = frag32(base, lane);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        MlpPieces<8> n32 = w32;
        if (mt + 1 < NT) n32 = frag32(base + (mt + 1) * PE_FRAG, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u][mt] = mlp_six<8>(w32, e32[u], acc[u][mt]);
        __builtin_amdgcn_sched_barrier(0);
        w32 = n32;
    }
    MlpPieces<4> w16 = frag16(base + F32B, lane);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        MlpPieces<4> n16 = w16;
        if (mt + 1 < NT) n16 = frag16(base + (mt + 1) * PE_FRAG + F32B, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u][mt] = mlp_six<4>(w16, e16[u], acc[u][mt]);
        __builtin_amdgcn_sched_barrier(0);
        w16 = n16;
    }
    mfma_settle();
}
// acc[u][mt] += W_h h (fragments [ks][mt] at `base`; hp[u][ks] = the split activations)
template <int TPW>
__device__ __forceinline__ void hidden_part(const char* base, int lane, const MlpPieces<8> (&hp)[TPW][2], f32x4 (&acc)[TPW][NT]) {
    MlpPieces<8> w = frag32(base, lane);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        MlpPieces<8> wn = w;
        if (f + 1 < 8) wn = frag32(base + (f + 1) * F32B, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u][f & 3] = mlp_six<8>(w, hp[u][f >> 2], acc[u][f & 3]);
        __builtin_amdgcn_sched_barrier(0);
        w = wn;
    }
    mfma_settle();
}
// the 16 activations of a lane -> its two B fragments
__device__ __forceinline__ void split_act(const float (&h)[ACT_STEPS], MlpPieces<8> (&hp)[2]) {
    const float ha[8] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    const float hb[8] = {h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]};
    mlp_split<8>(ha, hp[0]);
    mlp_split<8>(hb, hp[1]);
}
__device__ __forceinline__ void split_pe(const float (&e)[PE_STEPS], MlpPieces<8>& e32, MlpPieces<4>& e16) {
    const float ea[8] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]}, eb[4] = {e[8], e[9], e[10], e[11]};
    mlp_split<8>(ea, e32);
    mlp_split<4>(eb, e16);
}

// acc[mt] += W_e[:, slots of coordinate c] dE_c with the forward encoding fragments: c = 0 / 1 the low / high half of the K = 32 step
// (dj = the 4 Jacobian slots of the coordinate in that half, zeros in the other), c = 2 the K = 16 step
__device__ __forceinline__ void jac_part(const char* base, int lane, int c, const float* dj, f32x4 (&t)[NT]) {
    if (c < 2) {
        const float bv[8] = {c == 0 ? dj[0] : 0.f, c == 0 ? dj[1] : 0.f, c == 0 ? dj[2] : 0.f, c == 0 ? dj[3] : 0.f,
                             c == 1 ? dj[0] : 0.f, c == 1 ? dj[1] : 0.f, c == 1 ? dj[2] : 0.f, c == 1 ? dj[3] : 0.f};
        MlpPieces<8> b;
        mlp_split<8>(bv, b);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) t[mt] = mlp_six<8>(frag32(base + mt * PE_FRAG, lane), b, t[mt]);
    } else {
        const float bv[4] = {dj[0], dj[1], dj[2], dj[3]};
        MlpPieces<4> b;
        mlp_split<4>(bv, b);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) t[mt] = mlp_six<4>(frag16(base + mt * PE_FRAG + F32B, lane), b, t[mt]);
    }
    mfma_settle();
}

// TRANSPOSED hidden block (a reverse sweep's W^T g: contraction over the layer's OUTPUT channels): fragment rows are the layer's INPUT
// channels c0 + a, K index k runs over the rows of W
__device__ __forceinline__ void stage_hidden_t(char* dst, const float* __restrict__ W, int ld, int c0, int tid, int nthreads) {
    for (int idx = tid; idx < 2 * 4 * 64 * 8; idx += nthreads) {
        const int j = idx & 7, lane = (idx >> 3) & 63, mt = (idx >> 9) & 3, ks = idx >> 11;
        const int a = 16 * mt + (lane & 15), k = 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
        __bf16 h[3];
        split3(W[k * ld + c0 + a], h[0], h[1], h[2]);
        char* f = dst + (ks * 4 + mt) * F32B + lane * 16 + j * 2;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<__bf16*>(f + p * 1024) = h[p];
    }
}

}  // namespace ps
}  // namespace sc
