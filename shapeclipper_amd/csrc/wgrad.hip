// wgrad.hip -- weight-gradient GEMMs and bias/latent reductions of the implicit MLPs (gfx950).
//
// dW[out][in] = sum over sample points of  A(point)[out] * B(point)[in]  -- a GEMM with tiny M,N
// (64 x 48..112) and a huge K (= millions of points).  In the reference this is what autograd's
// addmm backward does layer by layer for SDFNetwork / RGBNetwork (model/implicit.py:138-161,220-239),
// twice for the SDF net because d(sdf)/dx is itself differentiated (renderer.py:101-107).
//
// Operands live in HBM in the tile-blocked layout the chain kernels write (TBL64, lane<->point).
// MFMA needs K<->point across lanes, i.e. a transpose: each workgroup restages 32 points per round
// through LDS as [channel][point] (row stride 34 => conflict-free fragment reads), applying cheap
// element-wise operand transforms on the way (softplus(a), p*softplus'(a), positional encoding
// from the raw point, ...) so that those tensors never have to be materialised in HBM.
// Wave w of a workgroup owns output rows 16w..16w+15 and all N tiles; accumulators stay in
// registers over the whole grid-stride loop; every workgroup writes one partial image, a second
// tiny kernel sums the partials in a fixed order (deterministic, no atomics).
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
};

constexpr int WG_PT = 32;      // points per round
constexpr int WG_LDP = 34;     // LDS row stride (floats): (2*i + g) mod 32 distinct for i<16, g<2
constexpr int WG_MAXNB = 112;

__device__ __forceinline__ float4 wg_load4(const float* base, int tile, int grp, int pt) {
    return reinterpret_cast<const float4*>(base)[((size_t)tile * 16 + grp) * 16 + pt];
}

__device__ __forceinline__ void wg_store_col(float* dst, int ch, int col, float4 v) {
    dst[(ch + 0) * WG_LDP + col] = v.x;
    dst[(ch + 1) * WG_LDP + col] = v.y;
    dst[(ch + 2) * WG_LDP + col] = v.z;
    dst[(ch + 3) * WG_LDP + col] = v.w;
}

__device__ __forceinline__ float sp_d1(float a) { float t, r; softplus_parts(a, t, r); return softplus_d1(a, t, r); }
__device__ __forceinline__ float sp_val(float a) { float t, r; softplus_parts(a, t, r); return softplus_val(a, t); }

// stage one 64-channel TBL operand (two 16-point tiles) into dst[64][WG_LDP]
__device__ __forceinline__ void wg_stage_tbl(float* dst, int op, const float* x0, const float* x1, const float* w5row,
                                             int tile0, int ntiles, int n_points, int tid) {
    for (int e = tid; e < 512; e += 256) {
        const int k = e >> 8, grp = (e >> 4) & 15, pt = e & 15;
        const int tile = tile0 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tile < ntiles && tile * TP + pt < n_points) {
            if (op == OP_PLAIN) {
                v = wg_load4(x0, tile, grp, pt);
            } else if (op == OP_SP) {
                const float4 a = wg_load4(x0, tile, grp, pt);
                v = make_float4(sp_val(a.x), sp_val(a.y), sp_val(a.z), sp_val(a.w));
            } else if (op == OP_Q) {
                const float4 pz = wg_load4(x0, tile, grp, pt);
                const float4 a = wg_load4(x1, tile, grp, pt);
                v = make_float4(pz.x * sp_d1(a.x), pz.y * sp_d1(a.y), pz.z * sp_d1(a.z), pz.w * sp_d1(a.w));
            } else if (op == OP_Q4) {
                const float4 a = wg_load4(x1, tile, grp, pt);
                const float4 w = reinterpret_cast<const float4*>(w5row)[grp];
                v = make_float4(w.x * sp_d1(a.x), w.y * sp_d1(a.y), w.z * sp_d1(a.z), w.w * sp_d1(a.w));
            }
        }
        wg_store_col(dst, 4 * grp, 16 * k + pt, v);
    }
}

// stage the 48-column positional-encoding operand (OP_PE: E, OP_EPS: g_grad[c] * dE/dx) for 32 points
__device__ __forceinline__ void wg_stage_pe(float* dst, int op, const float* points, const float* g_grad,
                                            int pt0, int n_points, bool symmetric, int tid) {
    const int pt = tid & 31, role = tid >> 5;   // role 0..5: frequency 2^role; role 6: raw + pads; role 7 idle
    if (role > 6) return;
    const int gp = pt0 + pt;
    const bool valid = gp < n_points;
    float x[3] = {0.f, 0.f, 0.f}, gm[3] = {0.f, 0.f, 0.f};
    if (valid) {
        x[0] = points[(size_t)gp * 3]; x[1] = points[(size_t)gp * 3 + 1]; x[2] = points[(size_t)gp * 3 + 2];
        if (op == OP_EPS) { gm[0] = g_grad[(size_t)gp * 3]; gm[1] = g_grad[(size_t)gp * 3 + 1]; gm[2] = g_grad[(size_t)gp * 3 + 2]; }
    }
    const float sg0 = symmetric ? (x[0] > 0.f ? 1.f : (x[0] < 0.f ? -1.f : 0.f)) : 1.f;
    if (symmetric) x[0] = fabsf(x[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sg = c == 0 ? sg0 : 1.f;
        if (role < 6) {
            const int gq = role >> 1, j0 = 2 * (role & 1);          // packed col = 4*(4c+j) + gq
            const float f = (float)(1 << role);
            float sn, cs;
            sincosf(x[c] * f, &sn, &cs);
            float v0, v1;
            if (op == OP_PE) { v0 = sn; v1 = cs; } else { v0 = gm[c] * f * cs * sg; v1 = -gm[c] * f * sn * sg; }
            if (!valid) { v0 = 0.f; v1 = 0.f; }
            dst[(4 * (4 * c + j0) + gq) * WG_LDP + pt] = v0;
            dst[(4 * (4 * c + j0 + 1) + gq) * WG_LDP + pt] = v1;
        } else {
            float v0 = op == OP_PE ? x[c] : gm[c] * sg;
            if (!valid) v0 = 0.f;
            dst[(4 * (4 * c + 0) + 3) * WG_LDP + pt] = v0;
            dst[(4 * (4 * c + 1) + 3) * WG_LDP + pt] = 0.f;
            dst[(4 * (4 * c + 2) + 3) * WG_LDP + pt] = 0.f;
            dst[(4 * (4 * c + 3) + 3) * WG_LDP + pt] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    __shared__ float Al[2][64 * WG_LDP];
    __shared__ float Bl[2][WG_MAXNB * WG_LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.n_points + TP - 1) / TP;
    const int nrounds = (ntiles + 1) / 2;
    const int nnt = (a.nb0 + a.nb1) / 16;
    f32x4 acc[7];
#pragma unroll
    for (int n = 0; n < 7; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int round = blockIdx.x; round < nrounds; round += gridDim.x) {
        const int tile0 = round * 2;
        __syncthreads();
        for (int t = 0; t < a.nterms; ++t) {
            const WgradTerm& T = a.t[t];
            wg_stage_tbl(Al[t], T.aop, T.a0, T.a1, a.w5row, tile0, ntiles, a.n_points, tid);
            if (T.bop0 == OP_PE || T.bop0 == OP_EPS)
                wg_stage_pe(Bl[t], T.bop0, a.points, a.g_grad, tile0 * TP, a.n_points, a.symmetric != 0, tid);
            else
                wg_stage_tbl(Bl[t], T.bop0, T.b0, nullptr, nullptr, tile0, ntiles, a.n_points, tid);
            if (a.nb1 > 0) {
                float* dst = Bl[t] + a.nb0 * WG_LDP;
                if (T.bop1 == OP_PE || T.bop1 == OP_EPS)
                    wg_stage_pe(dst, T.bop1, a.points, a.g_grad, tile0 * TP, a.n_points, a.symmetric != 0, tid);
                else
                    wg_stage_tbl(dst, T.bop1, T.b1, nullptr, nullptr, tile0, ntiles, a.n_points, tid);
            }
        }
        __syncthreads();
        for (int t = 0; t < a.nterms; ++t) {
            const float* ap = Al[t] + (16 * wave + i) * WG_LDP + g;
            const float* bp = Bl[t] + i * WG_LDP + g;
#pragma unroll
            for (int s = 0; s < WG_PT / 4; ++s) {
                const float av = ap[4 * s];
#pragma unroll
                for (int n = 0; n < 7; ++n)
                    if (n < nnt) acc[n] = mfma16(av, bp[n * 16 * WG_LDP + 4 * s], acc[n]);
            }
        }
    }
    float* out = a.partial + (size_t)blockIdx.x * a.partial_stride + a.out_offset;
#pragma unroll
    for (int n = 0; n < 7; ++n)
        if (n < nnt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * wave + 4 * g + r) * a.out_ld + 16 * n + i] = acc[n][r];
        }
}

// out[i] = sum_b partial[b][i], fixed order
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int nparts, int stride,
                                                             int n, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int b = 0; b < nparts; ++b) s += partial[(size_t)b * stride + idx];
    out[idx] = s;
}

// out[img][k][ch] += sum_{p in img} coef_k(p) * X[ch][p]     (K = 1: coef = 1;  K = 3: coef = cw[p][k])
struct TblSumArgs {
    const float* x;      // TBL64
    const float* coef;   // [n_points][3] or null
    int n_points, n_per_image, n_images;
    float* out;          // [n_images][K][64], pre-zeroed, atomicAdd
};

template <int K>
__global__ __launch_bounds__(256) void tbl_sum_kernel(TblSumArgs a) {
    const int tid = threadIdx.x, grp = tid >> 4, pt = tid & 15;
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
                    if (pt == 0 && have) atomicAdd(&a.out[((size_t)cur * K + k) * 64 + 4 * grp + r], v);
                } else if (have) {
                    atomicAdd(&a.out[((size_t)cur * K + k) * 64 + 4 * grp + r], v);
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
            const float4 v = wg_load4(a.x, tile, grp, pt);
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
             int nb0, int nb1, float* partial, int nparts, int partial_stride, int out_offset, int out_ld, void* stream_) {
    if (n_points <= 0) return 0;
    sc::WgradArgs a;
    a.t[0] = sc::WgradTerm{a0_0, a1_0, aop_0, b0_0, bop0_0, b1_0, bop1_0};
    a.t[1] = sc::WgradTerm{a0_1, a1_1, aop_1, b0_1, bop0_1, b1_1, bop1_1};
    a.nterms = nterms; a.points = points; a.g_grad = g_grad; a.w5row = w5row; a.n_points = n_points;
    a.symmetric = symmetric; a.nb0 = nb0; a.nb1 = nb1; a.partial = partial; a.partial_stride = partial_stride;
    a.out_offset = out_offset; a.out_ld = out_ld;
    hipLaunchKernelGGL(sc::wgrad_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

int sc_partial_reduce(const float* partial, int nparts, int stride, int n, float* out, void* stream_) {
    hipLaunchKernelGGL(sc::partial_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_,
                       partial, nparts, stride, n, out);
    return (int)hipGetLastError();
}

// out [n_images][K][64] must be zero-filled by the caller; K = 3 when coef != NULL else 1.
int sc_tbl_sum(const float* x, const float* coef, int n_points, int n_per_image, int n_images, float* out, void* stream_) {
    if (n_points <= 0) return 0;
    sc::TblSumArgs a{x, coef, n_points, n_per_image, n_images, out};
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + 63) / 64;
    if (blocks > 1024) blocks = 1024;
    if (coef) hipLaunchKernelGGL(sc::tbl_sum_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(sc::tbl_sum_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

}  // extern "C"
