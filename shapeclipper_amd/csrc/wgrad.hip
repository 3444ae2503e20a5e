// wgrad.hip -- weight-gradient GEMMs and bias/latent reductions of the implicit MLPs (gfx950).
//
// dW[out][in] = sum over sample points of  A(point)[out] * B(point)[in]  -- a GEMM with tiny M,N
// (64 x 48..112) and a huge K (= millions of points).  In the reference this is what autograd's
// addmm backward does layer by layer for SDFNetwork / RGBNetwork (model/implicit.py:138-161,220-239),
// twice for the SDF net because d(sdf)/dx is itself differentiated (renderer.py:101-107).
//
// Operands live in HBM in the tile-blocked layout the chain kernels write (TBL64, lane<->point).
// MFMA needs K<->point across lanes, i.e. a transpose; it is done in registers (4x4 quad transposes
// on coalesced float4 loads, see quad_transpose) -- no LDS staging, no barriers, every wave streams
// its own 16-point tiles independently.  Cheap element-wise operand transforms are applied on the
// way (softplus(a), p*softplus'(a), positional encoding from the raw point, ...) so those tensors are
// never materialised in HBM.  A wave keeps the whole [64 x N] accumulator in registers over its
// grid-stride loop; the four waves of a workgroup are combined in LDS and every workgroup writes one
// partial image; a second tiny kernel sums the partials in a fixed order.
// Bound: HBM (every operand byte is read once per GEMM it takes part in).
#include "mlp_tile.hpp"

namespace sc {

enum { OP_NONE = 0, OP_PLAIN = 1, OP_SP = 2, OP_Q = 3, OP_Q4 = 4, OP_PE = 5, OP_EPS = 6 };

struct WgradTerm {
    const float* a0;   // A operand, TBL64 (OP_PLAIN), or p (OP_Q: q = p * sp'(a1)), unused for OP_Q4
    const float* a1;   // pre-activation a_l for OP_Q / OP_Q4
    int aop;
    const float* b0;   // B segment 0: TBL64 (OP_PLAIN / OP_SP) or unused (OP_PE / OP_EPS)
    int bop0;
    const float* b1;   // B segment 1 (optional)
    int bop1;
};

struct WgradArgs {
    WgradTerm t[2];
    int nterms;
    const float* points;   // [n_points][3]   (OP_PE / OP_EPS)
    const float* g_grad;   // [n_points][3]   (OP_EPS: eps_j = g_grad[c(j)] * dE_j/dx)
    const float* w5row;    // 64 floats: W5[0,:] (OP_Q4)
    int n_points, symmetric;
    int nb0, nb1;          // widths of the two B segments (48 or 64; nb1 may be 0)
    float* partial;        // [gridDim.x][partial_stride]
    int partial_stride, out_offset, out_ld;
    float* rowsum;         // optional [nparts * 4][n_images][64]: per wave, per-image sum over points of term 0's A operand (fully
                           // written here; the caller adds the nparts * 4 partial images in index order: sc_partial_reduce)
    int n_per_image, n_images;   // (= the bias / latent gradient of the layer); needs n_per_image % 16 == 0
};

constexpr int WG_MAXNB = 112;

__device__ __forceinline__ float4 wg_load4(const float* base, int tile, int grp, int pt) {
    return reinterpret_cast<const float4*>(base)[((size_t)tile * 16 + grp) * 16 + pt];
}
__device__ __forceinline__ float sp_d1(float a) { float t, r; stash_parts(a, t, r); return stash_d1(a, t, r); }       // (a: the stashed value, mlp_tile.hpp SC_STASH_H)
__device__ __forceinline__ float sp_val(float a) {
#if SC_STASH_H
    return a;
#else
    float t, r; softplus_parts(a, t, r); return softplus_val(a, t);
#endif
}

// ---- fragment loads straight from the TBL64 image --------------------------------------------------
// MFMA 16x16x4 wants  A[i = channel][k = point]: lane (i = l&15, g = l>>4) supplies channel i of point
// "slot g" at each of the 4 K-steps of a 16-point tile.  K order is free, so slot g / step s := point
// 4g + s.  A lane therefore needs ONE channel for FOUR consecutive points, while TBL64 stores FOUR
// consecutive channels of ONE point per float4.  Let lane (i,g) load the float4 of channel group
// (i>>2) at point 4g + (i&3): the four lanes of a quad now hold a 4x4 (channel x point) block, and a
// 4x4 transpose inside the quad (two DPP-able xor-shuffle stages) leaves every lane with its channel
// at points 4g..4g+3.  The wave's 64 loads are one fully coalesced 1 KiB request; no LDS, no barrier.
// lane ^ 1 / lane ^ 2 inside a quad as DPP quad_perm moves (VALU, no LDS crossbar round trip)
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
}

__device__ __forceinline__ void quad_transpose(float4& v, int j) {
    const bool odd = j & 1;
    float s0 = odd ? v.x : v.y, s1 = odd ? v.z : v.w;
    float t0 = quad_xor1(s0), t1 = quad_xor1(s1);
    if (odd) { v.x = t0; v.z = t1; } else { v.y = t0; v.w = t1; }
    const bool hi = j & 2;
    s0 = hi ? v.x : v.z; s1 = hi ? v.y : v.w;
    t0 = quad_xor2(s0); t1 = quad_xor2(s1);
    if (hi) { v.x = t0; v.y = t1; } else { v.z = t0; v.w = t1; }
}

// channel tile T (16 channels) of a TBL operand for this lane: x..w = K-steps 0..3
// ---- operand fragments: raw loads (issued one tile ahead) and the transform + transpose applied at use ----
// channel tile T (16 channels) of a TBL operand, raw float4 of this lane: 4 consecutive channels of ONE point
template <int op>
__device__ __forceinline__ void wg_raw(const float* x0, const float* x1, int tile, int T, int i, int g, bool valid,
                                       float4& r0, float4& r1) {
    r0 = make_float4(0.f, 0.f, 0.f, 0.f);
    r1 = r0;
    const int grp = 4 * T + (i >> 2), pt = 4 * g + (i & 3);
    // TBL64 tensors are allocated in whole 16-point tiles, so the load is always in bounds; points past n_points are
    // zeroed after the transform (wg_cook).  Unconditional loads keep the loop body one basic block (a per-lane branch
    // around each of the ~40 loads of a tile cost more than the loads).
    (void)valid;
    if constexpr (op == OP_PLAIN || op == OP_SP) r0 = wg_load4(x0, tile, grp, pt);
    if constexpr (op == OP_Q) { r0 = wg_load4(x0, tile, grp, pt); r1 = wg_load4(x1, tile, grp, pt); }
    if constexpr (op == OP_Q4) r1 = wg_load4(x1, tile, grp, pt);
}
// transform + quad transpose: x..w = K-steps 0..3 (points 4g..4g+3) of channel 16T + i
template <int op>
__device__ __forceinline__ float4 wg_cook(float4 r0, float4 r1, const float* w5row, int T, int i, bool valid) {
    float4 v = r0;
    if constexpr (op == OP_SP) v = make_float4(sp_val(r0.x), sp_val(r0.y), sp_val(r0.z), sp_val(r0.w));
    if constexpr (op == OP_Q) v = make_float4(r0.x * sp_d1(r1.x), r0.y * sp_d1(r1.y), r0.z * sp_d1(r1.z), r0.w * sp_d1(r1.w));
    if constexpr (op == OP_Q4) {
        const float4 w = reinterpret_cast<const float4*>(w5row)[4 * T + (i >> 2)];
        v = make_float4(w.x * sp_d1(r1.x), w.y * sp_d1(r1.y), w.z * sp_d1(r1.z), w.w * sp_d1(r1.w));
    }
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);     // softplus(0) != 0: re-mask out-of-range points
    quad_transpose(v, i & 3);
    return v;
}

// sin/cos of this lane's PE column for the 4 points of its slot and the 3 coordinates (shared by the
// OP_PE and OP_EPS operands of one tile).  Column j of coordinate tile c: step = j>>2, owner gq = j&3.
struct PeLane {
    float pe[3][4];   // E column value at the 4 points of this lane's slot, per coordinate tile
    float ep[3][4];   // g_grad[c] * dE/dx_c   (OP_EPS)
};

__device__ __forceinline__ void pe_lane_setup(PeLane& P, const float* points, const float* g_grad, bool want_eps,
                                              int tile, int j, int g, int n_points, bool symmetric, bool full) {
    const int step = j >> 2, gq = j & 3;
    const bool raw = gq == 3, first = step == 0, iscos = step & 1;
    const float f = raw ? 0.f : (float)(1 << (2 * gq + (step >> 1)));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int gp = tile * TP + 4 * g + s;
        const bool valid = full || gp < n_points;
        float x[3] = {0.f, 0.f, 0.f}, gm[3] = {0.f, 0.f, 0.f};
        if (valid) {
            x[0] = points[(size_t)gp * 3]; x[1] = points[(size_t)gp * 3 + 1]; x[2] = points[(size_t)gp * 3 + 2];
            if (want_eps) { gm[0] = g_grad[(size_t)gp * 3]; gm[1] = g_grad[(size_t)gp * 3 + 1]; gm[2] = g_grad[(size_t)gp * 3 + 2]; }
        }
        const float sg0 = symmetric ? (x[0] > 0.f ? 1.f : (x[0] < 0.f ? -1.f : 0.f)) : 1.f;
        if (symmetric) x[0] = fabsf(x[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // hardware sin/cos (|error| ~ 1e-6 for these |arguments| <= 64): the weight gradient is a sum over ~1e6
            // points whose fp32 accumulation noise is larger; the accurate sincosf cost more than the MFMAs of this launch
            float sn, cs;
            __sincosf(x[c] * f, &sn, &cs);
            const float val = raw ? (first ? x[c] : 0.f) : (iscos ? cs : sn);
            const float der = raw ? (first ? 1.f : 0.f) : (iscos ? -f * sn : f * cs);
            P.pe[c][s] = valid ? val : 0.f;
            P.ep[c][s] = gm[c] * der * (c == 0 ? sg0 : 1.f);   // gm == 0 for out-of-range points
        }
    }
}

// B fragment of PE coordinate tile c (x..w = K-steps = points 4g..4g+3)
template <int op>
__device__ __forceinline__ float4 pe_frag(const PeLane& P, int c) {
    return op == OP_PE ? make_float4(P.pe[c][0], P.pe[c][1], P.pe[c][2], P.pe[c][3])
                       : make_float4(P.ep[c][0], P.ep[c][1], P.ep[c][2], P.ep[c][3]);
}

template <int A, int B0, int B1, int NT0, int NNT>
struct TermRaw {
    float4 a0[NT], a1[NT], b[NNT];
};

template <int A, int B0, int B1, int NT0, int NNT>
__device__ __forceinline__ void term_load(const WgradTerm& T, int tile, int i, int g, bool valid, TermRaw<A, B0, B1, NT0, NNT>& R) {
    float4 dummy;
#pragma unroll
    for (int m = 0; m < NT; ++m) wg_raw<A>(T.a0, T.a1, tile, m, i, g, valid, R.a0[m], R.a1[m]);
#pragma unroll
    for (int n = 0; n < NNT; ++n) {
        if (n < NT0) {
            if constexpr (B0 == OP_PLAIN || B0 == OP_SP) wg_raw<B0>(T.b0, nullptr, tile, n, i, g, valid, R.b[n], dummy);
        } else {
            if constexpr (B1 == OP_PLAIN || B1 == OP_SP) wg_raw<B1>(T.b1, nullptr, tile, n - NT0, i, g, valid, R.b[n], dummy);
        }
    }
}

template <int A, int B0, int B1, int NT0, int NNT>
__device__ __forceinline__ void term_compute(const TermRaw<A, B0, B1, NT0, NNT>& R, const PeLane& P, const float* w5row, int i,
                                             bool valid, f32x4 (&acc)[NT][NNT], float (&rowsum)[NT], bool want_rowsum) {
    float4 af[NT], bf[NNT];
#pragma unroll
    for (int m = 0; m < NT; ++m) af[m] = wg_cook<A>(R.a0[m], R.a1[m], w5row, m, i, valid);
    if (want_rowsum) {      // channel 16m + i of this lane, its 4 points of the tile (array by reference: stays in registers)
#pragma unroll
        for (int m = 0; m < NT; ++m) rowsum[m] += (af[m].x + af[m].y) + (af[m].z + af[m].w);
    }
#pragma unroll
    for (int n = 0; n < NNT; ++n) {
        if (n < NT0) {
            if constexpr (B0 == OP_PE || B0 == OP_EPS) bf[n] = pe_frag<B0>(P, n);
            else bf[n] = wg_cook<B0>(R.b[n], R.b[n], nullptr, n, i, valid);
        } else {
            if constexpr (B1 == OP_PE || B1 == OP_EPS) bf[n] = pe_frag<B1>(P, n - NT0);
            else if constexpr (B1 == OP_NONE) bf[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            else bf[n] = wg_cook<B1>(R.b[n], R.b[n], nullptr, n - NT0, i, valid);
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < NNT; ++n) {
                const float av = s == 0 ? af[m].x : (s == 1 ? af[m].y : (s == 2 ? af[m].z : af[m].w));
                const float bv = s == 0 ? bf[n].x : (s == 1 ? bf[n].y : (s == 2 ? bf[n].z : bf[n].w));
                acc[m][n] = mfma16(av, bv, acc[m][n]);
            }
}

// NT0: N tiles of B segment 0.  Term 0 = (A0; B00 | B10), optional term 1 = (A1; B01 | B11) (A1 == OP_NONE: absent).
// FULL: n_points is a multiple of 16 (always the case for renders: 64 samples per ray) -- no per-point validity masks.
template <int NNT, int NT0, int WPS, int A0, int B00, int B10, int A1, int B01, int B11, bool FULL>
__global__ __launch_bounds__(256, WPS) void wgrad_kernel(WgradArgs a) {
    __shared__ float red[64 * 16 * NNT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.n_points + TP - 1) / TP;
    constexpr int nnt = NNT;
    for (int e = tid; e < 64 * 16 * NNT; e += 256) red[e] = 0.f;
    f32x4 acc[NT][NNT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int n = 0; n < NNT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool need_pe = B00 == OP_PE || B10 == OP_PE || B01 == OP_EPS || B11 == OP_EPS || B00 == OP_EPS || B10 == OP_EPS;
    constexpr bool need_eps = B01 == OP_EPS || B11 == OP_EPS || B00 == OP_EPS || B10 == OP_EPS;

    // software pipeline: the raw operand loads of the NEXT tile are issued before the transforms + MFMAs of
    // the current one, so every wave always has ~20 KB of HBM reads in flight
    TermRaw<A0, B00, B10, NT0, NNT> c0, n0;
    TermRaw<A1, B01, B11, NT0, NNT> c1, n1;
    // every wave streams one CONTIGUOUS range of tiles (so that it stays inside one image most of the time and the
    // per-image row sums below need one flush per image change, not one per tile)
    const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    const bool want_rs = a.rowsum != nullptr;
    // with row sums: every WORKGROUP streams one contiguous range of tiles (its 4 waves interleaved), so a wave changes
    // image rarely and flushes its row sums once per change; without: grid-stride (all waves sweep the tensor together)
    const int per_wg = ((ntiles + gridDim.x - 1) / gridDim.x + 3) & ~3;
    const int step = want_rs ? 4 : nw;
    int tile = want_rs ? blockIdx.x * per_wg + wave : wid;
    const int tile_end = want_rs ? min(ntiles, (int)(blockIdx.x + 1) * per_wg) : ntiles;
    bool nvalid = tile * TP + 4 * g + (i & 3) < a.n_points;   // validity of the point this lane LOADS
    if (tile < tile_end) {
        term_load(a.t[0], tile, i, g, nvalid, n0);
        if constexpr (A1 != OP_NONE) term_load(a.t[1], tile, i, g, nvalid, n1);
    }
    float rsum[NT] = {0.f, 0.f, 0.f, 0.f};
    const int tiles_per_image = want_rs ? a.n_per_image / TP : 1;
    int cur_img = -1;
    // Row sums leave through a partial image PER WAVE ([n_images][64], zero-filled here): a wave visits the images in ascending
    // order, so every (image, channel) is stored at most once -- no atomics, and the caller's fixed-order sum over the waves makes the
    // result independent of timing (it used to be one float atomicAdd per wave, image and channel).
    float* rs_part = want_rs ? a.rowsum + (size_t)(blockIdx.x * 4 + wave) * a.n_images * 64 : nullptr;
    if (want_rs)
        for (int e = lane; e < a.n_images * 64; e += 64) rs_part[e] = 0.f;
    auto flush_rowsum = [&]() {          // wave-uniform: cur_img is the same in every lane
        if (cur_img >= 0) {
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                float v = rsum[m];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (g == 0) rs_part[(size_t)cur_img * 64 + 16 * m + i] = v;
                rsum[m] = 0.f;
            }
        }
    };
    for (; tile < tile_end; tile += step) {
        c0 = n0;
        if constexpr (A1 != OP_NONE) c1 = n1;
        const bool valid = FULL ? true : nvalid;
        const int next = tile + step;
        if (next < tile_end) {
            nvalid = next * TP + 4 * g + (i & 3) < a.n_points;
            term_load(a.t[0], next, i, g, nvalid, n0);
            if constexpr (A1 != OP_NONE) term_load(a.t[1], next, i, g, nvalid, n1);
        }
        if (want_rs) {
            const int img = min(tile / tiles_per_image, a.n_images - 1);
            if (img != cur_img) {
                flush_rowsum();
                cur_img = img;
            }
        }
        PeLane P;
        if constexpr (need_pe) pe_lane_setup(P, a.points, a.g_grad, need_eps, tile, i, g, a.n_points, a.symmetric != 0, FULL);
        term_compute(c0, P, a.w5row, i, valid, acc, rsum, want_rs);
        if constexpr (A1 != OP_NONE) term_compute(c1, P, a.w5row, i, valid, acc, rsum, false);
    }
    if (want_rs) flush_rowsum();
    // combine the four waves of the workgroup in LDS in wave order (a fixed summation order), then one partial image per workgroup
    const int ld = 16 * nnt;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int n = 0; n < NNT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * m + 4 * g + r) * ld + 16 * n + i] += acc[m][n][r];
        }
    }
    __syncthreads();
    float* out = a.partial + (size_t)blockIdx.x * a.partial_stride + a.out_offset;
    for (int e = tid; e < 64 * ld; e += 256) out[(e / ld) * a.out_ld + (e % ld)] = red[e];
}

// out[i] = sum_b partial[b][i] in a FIXED order (no atomics: results do not depend on timing).  A workgroup of 1024 threads handles
// 32 consecutive elements; its 32 thread rows take the parts b = row, row + 32, ... (four interleaved sequential sums each, so that four
// loads are in flight) and the 32 row sums are added by a fixed binary tree.  `out` is assigned (it need not be zero-filled).
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* __restrict__ partial, int nparts, int stride,
                                                              int n, float* __restrict__ out) {
    __shared__ float rows[32][33];
    const int e = threadIdx.x & 31, row = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < n) {
        const float* src = partial + idx;
        int b = row;
        for (; b + 96 < nparts; b += 128) {
            s0 += src[(size_t)b * stride];
            s1 += src[(size_t)(b + 32) * stride];
            s2 += src[(size_t)(b + 64) * stride];
            s3 += src[(size_t)(b + 96) * stride];
        }
        for (; b < nparts; b += 32) s0 += src[(size_t)b * stride];
    }
    rows[row][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int half = 16; half >= 1; half >>= 1) {
        if (row < half) rows[row][e] += rows[row + half][e];
        __syncthreads();
    }
    if (row == 0 && idx < n) out[idx] = rows[0][e];
}

// out[img][k][ch] += sum_{p in img} coef_k(p) * X[ch][p]     (K = 1: coef = 1;  K = 3: coef = cw[p][k])
struct TblSumArgs {
    const float* xs[8];  // up to 8 TBL64 tensors, one per blockIdx.y
    const float* coef;   // [n_points][3] or null
    int n_points, n_per_image, n_images;
    float* outs[8];      // each [n_images][K][64]: legacy mode (part == null): pre-zeroed, atomicAdd
    float* part;         // fixed-order mode: [gridDim.x][n_tensors][n_images][K][64] partial images, one per block, fully written here
};

template <int K>
__global__ __launch_bounds__(256) void tbl_sum_kernel(TblSumArgs a) {
    const int tid = threadIdx.x, grp = tid >> 4, pt = tid & 15;
    const float* __restrict__ x = a.xs[blockIdx.y];
    // Fixed-order mode (n_per_image % 16 == 0: a tile never straddles two images): the block owns a partial image, visits the images in
    // ascending order and stores every (image, k, channel) at most once; the caller adds the blocks' images in block order.
    const bool fixed = a.part != nullptr;
    const size_t E = (size_t)a.n_images * K * 64;
    float* __restrict__ out = fixed ? a.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * E : a.outs[blockIdx.y];
    if (fixed) {
        for (size_t e = tid; e < E; e += 256) out[e] = 0.f;
        __syncthreads();
    }
    const int ntiles = (a.n_points + TP - 1) / TP;
    const int tiles_per_block = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
    float acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
    int cur = -1;
    // Wave-uniform control flow: whenever ANY lane moves on to another image every lane flushes what it
    // has (flushing early is harmless, the accumulators restart at zero).
    auto flush = [&]() {
        const bool have = cur >= 0;
        const int row_img = __shfl(cur, (tid & 63) & ~15);
        const bool uniform = __all(cur == row_img);   // every 16-point row agrees on its image
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = have ? acc[k][r] : 0.f;
                if (uniform) {
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    if (pt == 0 && have) {
                        if (fixed) out[((size_t)cur * K + k) * 64 + 4 * grp + r] += v;       // (this lane is the only writer of the address)
                        else atomicAdd(&out[((size_t)cur * K + k) * 64 + 4 * grp + r], v);
                    }
                } else if (have) {
                    atomicAdd(&out[((size_t)cur * K + k) * 64 + 4 * grp + r], v);
                }
                acc[k][r] = 0.f;
            }
    };
    for (int tile = t0; tile < t1; ++tile) {
        const int gp = tile * TP + pt;
        const bool valid = gp < a.n_points;
        const int img = valid ? min(gp / a.n_per_image, a.n_images - 1) : cur;
        if (__any(img != cur)) {
            flush();
            cur = img;
        }
        if (valid) {
            const float4 v = wg_load4(x, tile, grp, pt);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float c = K == 1 ? 1.f : a.coef[(size_t)gp * 3 + k];
                acc[k][0] = __builtin_fmaf(c, v.x, acc[k][0]);
                acc[k][1] = __builtin_fmaf(c, v.y, acc[k][1]);
                acc[k][2] = __builtin_fmaf(c, v.z, acc[k][2]);
                acc[k][3] = __builtin_fmaf(c, v.w, acc[k][3]);
            }
        }
    }
    flush();
}

}  // namespace sc

extern "C" {

// Generic two-term weight-gradient GEMM; see WgradArgs.  ops: 1 plain TBL, 2 softplus(a), 3 p*sp'(a),
// 4 w5*sp'(a4), 5 positional encoding of the point, 6 g_grad-weighted PE Jacobian.
int sc_wgrad(int nterms,
             const float* a0_0, const float* a1_0, int aop_0, const float* b0_0, int bop0_0, const float* b1_0, int bop1_0,
             const float* a0_1, const float* a1_1, int aop_1, const float* b0_1, int bop0_1, const float* b1_1, int bop1_1,
             const float* points, const float* g_grad, const float* w5row, int n_points, int symmetric,
             int nb0, int nb1, float* partial, int nparts, int partial_stride, int out_offset, int out_ld,
             float* rowsum, int n_per_image, int n_images, void* stream_) {
    if (n_points <= 0) return 0;
    if (rowsum && (n_per_image <= 0 || n_per_image % sc::TP != 0 || n_images <= 0)) return (int)hipErrorInvalidValue;
    sc::WgradArgs a;
    a.t[0] = sc::WgradTerm{a0_0, a1_0, aop_0, b0_0, bop0_0, b1_0, bop1_0};
    a.t[1] = sc::WgradTerm{a0_1, a1_1, aop_1, b0_1, bop0_1, b1_1, bop1_1};
    a.nterms = nterms; a.points = points; a.g_grad = g_grad; a.w5row = w5row; a.n_points = n_points;
    a.symmetric = symmetric; a.nb0 = nb0; a.nb1 = nb1; a.partial = partial; a.partial_stride = partial_stride;
    a.out_offset = out_offset; a.out_ld = out_ld;
    a.rowsum = rowsum; a.n_per_image = n_per_image; a.n_images = n_images;
    hipStream_t st = (hipStream_t)stream_;
    using namespace sc;
    const int key0 = aop_0 * 100 + bop0_0 * 10 + bop1_0;
    const int key1 = nterms > 1 ? aop_1 * 100 + bop0_1 * 10 + bop1_1 : 0;
#define SC_WG(NNT, NT0, WPS, A0, B00, B10, A1, B01, B11)                                                         \
    if (nb0 == 16 * NT0 && nb0 + nb1 == 16 * NNT && key0 == A0 * 100 + B00 * 10 + B10 &&                         \
        key1 == (A1 == OP_NONE ? 0 : A1 * 100 + B01 * 10 + B11)) {                                               \
        if (n_points % TP == 0)                                                                                  \
            hipLaunchKernelGGL((wgrad_kernel<NNT, NT0, WPS, A0, B00, B10, A1, B01, B11, true>), dim3(nparts), dim3(256), 0, st, a); \
        else                                                                                                     \
            hipLaunchKernelGGL((wgrad_kernel<NNT, NT0, WPS, A0, B00, B10, A1, B01, B11, false>), dim3(nparts), dim3(256), 0, st, a); \
        return (int)hipGetLastError();                                                                           \
    }
    SC_WG(3, 3, 2, OP_PLAIN, OP_PE, OP_NONE, OP_Q, OP_EPS, OP_NONE)         // dW0e, dW1e, dW2e
    SC_WG(3, 3, 2, OP_PLAIN, OP_PE, OP_NONE, OP_NONE, OP_NONE, OP_NONE)     // ... without d/d(grad); dV0e
    SC_WG(4, 4, 2, OP_PLAIN, OP_SP, OP_NONE, OP_Q, OP_PLAIN, OP_NONE)       // dW1h, dW2h, dW3
    SC_WG(4, 4, 2, OP_PLAIN, OP_SP, OP_NONE, OP_Q4, OP_PLAIN, OP_NONE)      // dW4
    SC_WG(4, 4, 2, OP_PLAIN, OP_SP, OP_NONE, OP_NONE, OP_NONE, OP_NONE)     // dW5 feature rows, no-Gg variants
    SC_WG(4, 4, 2, OP_PLAIN, OP_PLAIN, OP_NONE, OP_NONE, OP_NONE, OP_NONE)  // dV0f, dV1, dV2
    SC_WG(7, 3, 2, OP_PLAIN, OP_PE, OP_PLAIN, OP_NONE, OP_NONE, OP_NONE)    // dV0 = [PE 48 | feature 64] in one pass over Gy0
#undef SC_WG
    return (int)hipErrorInvalidValue;   // operand combination not instantiated
}

int sc_partial_reduce(const float* partial, int nparts, int stride, int n, float* out, void* stream_) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sc::partial_reduce_kernel, dim3((n + 31) / 32), dim3(1024), 0, (hipStream_t)stream_,
                       partial, nparts, stride, n, out);
    return (int)hipGetLastError();
}

// Number of blocks (= partial images per tensor in fixed-order mode) sc_tbl_sum launches for n_points.
int sc_tbl_sum_blocks(int n_points) {
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + 63) / 64;
    return blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
}

// xs / outs: HOST arrays of n_tensors (<= 8) device pointers; K = 3 when coef != NULL else 1.  One launch for all tensors (blockIdx.y).
// part != NULL (needs n_per_image % 16 == 0): fixed summation order -- `part` is sc_tbl_sum_blocks(n_points) * n_tensors * n_images * K * 64
// floats of workspace, outs[0] must be ONE buffer holding all tensors back to back ([n_tensors][n_images][K][64], fully written);
// part == NULL: legacy mode, every out [n_images][K][64] zero-filled by the caller, float atomicAdd (order depends on timing).
int sc_tbl_sum(const float* const* xs, int n_tensors, const float* coef, int n_points, int n_per_image, int n_images,
               float* const* outs, float* part, void* stream_) {
    if (n_points <= 0 || n_tensors <= 0) return 0;
    if (n_tensors > 8) return (int)hipErrorInvalidValue;
    if (part && (n_per_image <= 0 || n_per_image % sc::TP != 0)) return (int)hipErrorInvalidValue;
    sc::TblSumArgs a;
    for (int t = 0; t < 8; ++t) { a.xs[t] = t < n_tensors ? xs[t] : nullptr; a.outs[t] = t < n_tensors ? outs[t] : nullptr; }
    a.coef = coef; a.n_points = n_points; a.n_per_image = n_per_image; a.n_images = n_images; a.part = part;
    const int blocks = sc_tbl_sum_blocks(n_points);
    if (coef) hipLaunchKernelGGL(sc::tbl_sum_kernel<3>, dim3(blocks, n_tensors), dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(sc::tbl_sum_kernel<1>, dim3(blocks, n_tensors), dim3(256), 0, (hipStream_t)stream_, a);
    if (part) {
        const int E = n_tensors * n_images * (coef ? 3 : 1) * 64;
        hipLaunchKernelGGL(sc::partial_reduce_kernel, dim3((E + 31) / 32), dim3(1024), 0, (hipStream_t)stream_, part, blocks, E, E, outs[0]);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
