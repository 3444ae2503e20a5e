// sdf_fwd_stream.hip -- conditional SDF MLP forward (value, feature, d sdf/dx; training stashes) in the exact three-piece bf16 split
// arithmetic with the PRE-SPLIT weights STREAMED through LDS one layer at a time (round 6).
//
// Same contract as sdf_fwd.hip (SDFNetwork.forward / get_conditional_output, model/implicit.py:138-189, incl. the autograd.grad call at
// :180-186 and the renderer's d density / dx, model/renderer.py:94-107).  What is different is where the weights live:
//   * sdf_fwd.hip keeps the whole network in LDS as fp32 (119 KiB) and runs fp32 MFMAs (27.6 k matrix cycles per 16 points);
//   * pre-split fragments (6 bytes per value, and a second, transposed set for the adjoint sweep) are 324 KiB -- two CUs' worth of LDS.
// Here the 8 waves of a workgroup walk the chain in LOCK STEP, one PHASE (a layer of the value chain, or a step of the adjoint sweep) at a
// time; the encoding blocks and two hidden blocks are LDS-resident, the other seven hidden blocks of a round alternate through two LDS
// slots by LDS-DMA (issued at the start of the phase after the slot's last reader, awaited at that phase's end; one barrier per phase).  Per 16 points 11.6 k matrix cycles; a workgroup
// streams 168 KiB per 128 points from L2.
//
// Phases (fragment layouts: mlp_presplit.hpp; T = transposed: contraction over the layer's OUTPUT channels):
//   0 W0 e | 1 W1 [h0; e] | 2 W2 [h1; e] | 3 W3 h2 | 4 W4 h3 | 5 W5[1:] h4 (feature rows; the sdf row is an fp32 dot)
//   6 W4^T q4 | 7 W3^T q3 | 8 W2e dE (Jacobian of the encoding) + W2h^T q2 | 9 W1e dE + W1h^T q1 | 10 W0 dE
// The encoding's Jacobian dE/dx_c has 4 non-zero slots per coordinate: coordinate 0 / 1 are the low / high half of the encoding's K = 32
// fragment (the other half of the B operand is zero), coordinate 2 is its K = 16 fragment -- no second copy of W_e.
// Parked between the sweeps: the ACTIVATIONS h_l (SC_STASH_H form: sp' = 1 - exp(-100 h)), in the training stash or in the per-wave L2
// scratch of sdf_fwd.hip.
#include "mlp_presplit.hpp"

namespace sc {
namespace st {
using namespace ps;

#ifndef SC_STREAM_ABLATE
#define SC_STREAM_ABLATE 0      // timing experiments (wrong values): 1 = no DMA, no wait; 2 = also no phase barriers; 3 = DMA but no wait; 4 = wait but no DMA
#endif
constexpr int WAVES = 8;
constexpr int NPH = 11;
constexpr int BUF_BYTES = HID_BYTES + PE_BYTES;            // the largest phase (43,008 bytes)
__host__ __device__ constexpr int ph_bytes(int ph) {
    return (ph == 0 || ph == 10) ? PE_BYTES : (ph == 1 || ph == 2 || ph == 8 || ph == 9) ? HID_BYTES + PE_BYTES : HID_BYTES;
}
__host__ __device__ constexpr int ph_off(int ph) {
    int o = 0;
    for (int i = 0; i < ph; ++i) o += ph_bytes(i);
    return o;
}
constexpr int IMG_BYTES = ph_off(NPH);                     // 331,776
// LDS: the three encoding blocks (W0, W1e, W2e: used twice per round) and two hidden blocks (W3, W4) stay RESIDENT; the other seven hidden
// blocks of a round (W1h, W2h, W5f, W4^T, W3^T, W2h^T, W1h^T: 168 KiB) alternate through two slots.  (First form of this kernel: every
// phase's fragments streamed, 324 KiB per round -- the DMA traffic alone cost 1.2 of 7.3 ms, profiles/r06_sdf_stream_ablation.txt.)
constexpr int OFF_PE0 = 0, OFF_PE1 = PE_BYTES, OFF_PE2 = 2 * PE_BYTES;
constexpr int OFF_R0 = 3 * PE_BYTES, OFF_R1 = OFF_R0 + HID_BYTES;
constexpr int OFF_S0 = OFF_R1 + HID_BYTES, OFF_S1 = OFF_S0 + HID_BYTES;
constexpr int OFF_W5 = OFF_S1 + HID_BYTES;                 // fp32: sdf row of W5 [64], b5 [65]
constexpr int LDS_BYTES = OFF_W5 + (64 + 68) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "resident + streamed fragments must fit one CU's LDS");

// ---- the streamed image: one launch per weight update ----------------------------------------------------------------------------------
// element e of a hidden block: (ks, mt, lane, j) -> W[row][col]; T: the fragment's rows are the layer's INPUT channels
__device__ __forceinline__ void pack_value(char* img, int off, bool k16, int lane, int j, float v) {
    __bf16 h[3];
    split3(v, h[0], h[1], h[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (k16) *reinterpret_cast<__bf16*>(img + off + p * 512 + lane * 8 + j * 2) = h[p];
        else *reinterpret_cast<__bf16*>(img + off + p * 1024 + lane * 16 + j * 2) = h[p];
    }
}
__global__ __launch_bounds__(256) void sdf_stream_pack_kernel(const float* __restrict__ w, char* __restrict__ img) {
    // hidden blocks: {phase, byte offset inside the phase, matrix offset in the SdfPack image, row stride, first column, transposed}
    struct Hid { int ph, off, w, ld, c0, t; };
    const Hid hid[9] = {{1, 0, SdfPack::W1, 112, 0, 0}, {2, 0, SdfPack::W2, 112, 0, 0}, {3, 0, SdfPack::W3, 64, 0, 0}, {4, 0, SdfPack::W4, 64, 0, 0},
                        {5, 0, SdfPack::W5 + 64, 64, 0, 0},                 // rows 1..64 of W5
                        {6, 0, SdfPack::W4, 64, 0, 1}, {7, 0, SdfPack::W3, 64, 0, 1}, {8, 0, SdfPack::W2, 112, 0, 1}, {9, 0, SdfPack::W1, 112, 0, 1}};
    struct Pe { int ph, off, w, ld, c0; };
    const Pe pe[5] = {{0, 0, SdfPack::W0, 48, 0}, {1, HID_BYTES, SdfPack::W1, 112, 64}, {2, HID_BYTES, SdfPack::W2, 112, 64},
                      {8, HID_BYTES, SdfPack::W2, 112, 64}, {9, HID_BYTES, SdfPack::W1, 112, 64}};
    const int NH = 9 * 4096, NP = 6 * 3072;      // (W0 appears in phases 0 and 10)
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < NH + NP; idx += gridDim.x * 256) {
        if (idx < NH) {
            const Hid b = hid[idx >> 12];
            const int e = idx & 4095, j = e & 7, lane = (e >> 3) & 63, mt = (e >> 9) & 3, ks = e >> 11;
            const int a = 16 * mt + (lane & 15), k = 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
            const float v = b.t ? w[b.w + k * b.ld + b.c0 + a] : w[b.w + a * b.ld + b.c0 + k];
            pack_value(img, ph_off(b.ph) + b.off + (ks * 4 + mt) * F32B, false, lane, j, v);
        } else {
            const int q = idx - NH, blk = q / 3072;
            const Pe b = blk < 5 ? pe[blk] : Pe{10, 0, SdfPack::W0, 48, 0};
            const int e = q - blk * 3072, s = e % 12, lane = (e / 12) & 63, mt = e / 768;
            const float v = w[b.w + (16 * mt + (lane & 15)) * b.ld + b.c0 + 4 * s + (lane >> 4)];
            const int base = ph_off(b.ph) + b.off + mt * PE_FRAG;
            if (s < 8) pack_value(img, base, false, lane, s, v);
            else pack_value(img, base + F32B, true, lane, s - 8, v);
        }
    }
}

// One LDS-DMA wave instruction (gfx950 global_load_lds_dwordx4, the idiom of gemm8p.hpp / conv3x3.hip): lane i's 16 bytes of
// `sbase + 16 i` land at LDS byte `lds_dst + 16 i` (both wave-uniform).  No staging registers, no ds_write pass.  Issued from asm: hipcc
// does not count it, so every wait it inserts for its own loads over-waits at worst (safe); completion is awaited by hand
// (s_waitcnt vmcnt(0) in front of the phase barrier).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void glds1k(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

struct Args {
    const float* points;   // [n_points][3]
    const char* img;       // the streamed image (sc_sdf_stream_pack)
    const float* w;        // SdfPack image (fp32): the sdf row of W5 and b5
    const float* cbias;    // [n_images][5][64]
    int n_points, n_per_image, n_images, symmetric;
    float* sdf;            // [n_points] or null
    float* grad;           // [n_points][3] (GRAD)
    float* feat;           // TBL64 or null
    float* stash_a;        // [5] x TBL64 or null: the parked activations h_l (training)
    float* stash_p;        // [4] x TBL64 or null: adjoint p_0..p_3 (training)
    float* scratch;        // GRAD without stash_a: [gridDim.x * WAVES][5][1024] floats of per-wave scratch
};

template <bool STASH>
__global__ __launch_bounds__(64 * WAVES) void sdf_fwd_stream_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int ntiles = (a.n_points + TP - 1) / TP;
    const size_t tbl = (size_t)ntiles * 1024;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_ptr_t)lds);
    // a 24 KiB hidden block global -> LDS by DMA: 24 one-KiB wave instructions, three per wave
#define SC_DMA(PH, SLOT)                                                                                             \
    if (SC_STREAM_ABLATE == 0 || SC_STREAM_ABLATE == 3) {                                                            \
        _Pragma("unroll") for (int i = 0; i < HID_BYTES / 1024 / WAVES; ++i)                                         \
            glds1k(a.img + ph_off(PH) + (wave + WAVES * i) * 1024, lane * 16, lds0 + (SLOT) + (wave + WAVES * i) * 1024); \
    }
    {   // resident blocks and the first two streamed ones
        float* w5 = reinterpret_cast<float*>(lds + OFF_W5);
        if (tid < 64) w5[tid] = a.w[SdfPack::W5 + tid];
        if (tid < 65) w5[64 + tid] = a.w[SdfPack::B5 + tid];
        auto copy = [&](int dst, int src, int bytes) {
            const uint4* sp = reinterpret_cast<const uint4*>(a.img + src);
            for (int i = tid; i < bytes / 16; i += 64 * WAVES) reinterpret_cast<uint4*>(lds + dst)[i] = sp[i];
        };
        copy(OFF_PE0, ph_off(0), PE_BYTES);
        copy(OFF_PE1, ph_off(1) + HID_BYTES, PE_BYTES);
        copy(OFF_PE2, ph_off(2) + HID_BYTES, PE_BYTES);
        copy(OFF_R0, ph_off(3), HID_BYTES);
        copy(OFF_R1, ph_off(4), HID_BYTES);
        copy(OFF_S0, ph_off(1), HID_BYTES);
        copy(OFF_S1, ph_off(2), HID_BYTES);
    }
    __syncthreads();
    const float* w5s = reinterpret_cast<const float*>(lds + OFF_W5) + 4 * g;
    const float* b5 = reinterpret_cast<const float*>(lds + OFF_W5) + 64;

    float* park = a.stash_a;
    size_t park_stride = tbl;
    if (!STASH && !park) { park = a.scratch + (size_t)(blockIdx.x * WAVES + wave) * 5 * 1024; park_stride = 1024; }

    // every wave runs the same number of rounds (the phases end in workgroup barriers); a wave past the last tile computes on the last
    // tile again and stores nothing but (identical) parked activations
    const int rounds = (ntiles + gridDim.x * WAVES - 1) / (gridDim.x * WAVES);
    for (int rd = 0; rd < rounds; ++rd) {
        const int tile_raw = (rd * gridDim.x + blockIdx.x) * WAVES + wave;
        const bool live = tile_raw < ntiles;
        const int tile = live ? tile_raw : ntiles - 1;
        const int ptile = (STASH || a.stash_a) ? tile : 0;
        const int pt = tile * TP + p;
        const bool valid = live && pt < a.n_points;
        const int ptc = pt < a.n_points ? pt : a.n_points - 1;
        const float x0 = a.points[(size_t)ptc * 3 + 0], x1 = a.points[(size_t)ptc * 3 + 1], x2 = a.points[(size_t)ptc * 3 + 2];
        const float* cb = a.cbias + (size_t)min(ptc / a.n_per_image, a.n_images - 1) * 320 + 4 * g;

        float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
        pe_slots<true, false>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
        MlpPieces<8> e32[1], hp[1][2];
        MlpPieces<4> e16[1];
        split_pe(e, e32[0], e16[0]);
        f32x4 acc[1][NT];
        float h[ACT_STEPS], q[ACT_STEPS];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;

        // A phase ends by awaiting the DMA it issued (and its stores) and a workgroup barrier: the block it fetched is complete for
        // everybody, and the slot it read may be overwritten by the next phase's DMA.  A phase's own loads (biases, parked activations)
        // are issued BEFORE its DMA: the memory counter retires in order, a load behind the DMA would wait for all of it.
#define SC_PH_END()                                                                                                  \
        {                                                                                                            \
            if (SC_STREAM_ABLATE == 0 || SC_STREAM_ABLATE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
            if (SC_STREAM_ABLATE != 2) __syncthreads();                                                              \
        }
        // h = softplus(acc); parked; split for the next layer
#define SC_ACTIVATE(L)                                                                                               \
        {                                                                                                            \
            _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                                                  \
                const float av = acc[0][s >> 2][s & 3];                                                              \
                float t, r;                                                                                          \
                softplus_parts(av, t, r);                                                                            \
                h[s] = softplus_val(av, t);                                                                          \
            }                                                                                                        \
            tbl_store_pinned(park + (size_t)(L) * park_stride, ptile, p, g, h);                                      \
            split_act(h, hp[0]);                                                                                     \
        }

        // ---- value chain ----
        acc_init(acc[0], cb);                                       // phase 0: W0 e
        pe_part<1>(lds + OFF_PE0, lane, e32, e16, acc);
        SC_ACTIVATE(0)
        SC_PH_END()

        acc_init(acc[0], cb + 64);                                  // phase 1: W1 [h0; e]   (W1h in slot 0)
        hidden_part<1>(lds + OFF_S0, lane, hp, acc);
        pe_part<1>(lds + OFF_PE1, lane, e32, e16, acc);
        SC_ACTIVATE(1)
        SC_PH_END()

        acc_init(acc[0], cb + 128);                                 // phase 2: W2 [h1; e]   (W2h in slot 1); W5f -> slot 0
        SC_DMA(5, OFF_S0)
        hidden_part<1>(lds + OFF_S1, lane, hp, acc);
        pe_part<1>(lds + OFF_PE2, lane, e32, e16, acc);
        SC_ACTIVATE(2)
        SC_PH_END()

        acc_init(acc[0], cb + 192);                                 // phase 3: W3 h2 (resident); W4^T -> slot 1
        SC_DMA(6, OFF_S1)
        hidden_part<1>(lds + OFF_R0, lane, hp, acc);
        SC_ACTIVATE(3)
        SC_PH_END()

        acc_init(acc[0], cb + 256);                                 // phase 4: W4 h3 (resident)
        hidden_part<1>(lds + OFF_R1, lane, hp, acc);
        SC_ACTIVATE(4)
        SC_PH_END()

        // ---- phase 5, output layer: sdf by the fp32 dot of sdf_fwd.hip, feature rows by MFMA (W5f in slot 0) ----
        {
            float sp = 0.f;
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) sp = __builtin_fmaf(w5s[kp(s)], h[s], sp);
            const float sdf = group_sum(sp) + b5[0];
            if (a.sdf && valid && g == 0) a.sdf[pt] = sdf;
        }
        if (STASH || a.feat) {
            acc_init(acc[0], b5 + 1 + 4 * g);
            hidden_part<1>(lds + OFF_S0, lane, hp, acc);
            float fv[ACT_STEPS];
            acc_to_regs(acc[0], fv);
            if (live) tbl_store_pinned(a.feat, tile, p, g, fv);
        }
        // q4 = W5[0, :] sp'(a4), from the activation still in registers
#pragma unroll
        for (int s = 0; s < ACT_STEPS; ++s) {
            float t, r;
            stash_parts(h[s], t, r);
            q[s] = w5s[kp(s)] * stash_d1(h[s], t, r);
        }
        SC_PH_END()

        // ---- adjoint sweep ----
        // p_l = W_{l+1,h}^T q_{l+1} (transposed fragments in SLOT);  q_l = p_l sp'(a_l), sp' from the parked activation
#define SC_ADJOINT_LOAD(L)                                                                                           \
        float hl_##L[ACT_STEPS];                                                                                     \
        tbl_load(park + (size_t)(L) * park_stride, ptile, p, g, hl_##L);                                             \
        __builtin_amdgcn_sched_barrier(0);
#define SC_ADJOINT(SLOT, L)                                                                                          \
        {                                                                                                            \
            split_act(q, hp[0]);                                                                                     \
            acc_zero(acc[0]);                                                                                        \
            hidden_part<1>(lds + (SLOT), lane, hp, acc);                                                             \
            float pv_[ACT_STEPS];                                                                                    \
            acc_to_regs(acc[0], pv_);                                                                                \
            if (STASH || a.stash_p) { if (live) tbl_store_pinned(a.stash_p + (size_t)(L) * tbl, tile, p, g, pv_); }  \
            _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                                                  \
                float t, r;                                                                                          \
                stash_parts(hl_##L[s], t, r);                                                                        \
                q[s] = pv_[s] * stash_d1(hl_##L[s], t, r);                                                           \
            }                                                                                                        \
        }
        // g_c += sum_s q[s] (W_e dE_c)[s], one coordinate at a time (encoding fragments at BASE)
#define SC_PE_JAC(BASE)                                                                                              \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                              \
            f32x4 t_[NT];                                                                                            \
            acc_zero(t_);                                                                                            \
            jac_part(lds + (BASE), lane, c, d1 + 4 * c, t_);                                                         \
            float gs_ = 0.f;                                                                                         \
            _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) gs_ = __builtin_fmaf(q[s], t_[s >> 2][s & 3], gs_); \
            if (c == 0) g0 += gs_; else if (c == 1) g1 += gs_; else g2 += gs_;                                       \
        }
        SC_ADJOINT_LOAD(3)                                          // phase 6: p3 = W4^T q4 (slot 1); W3^T -> slot 0
        SC_DMA(7, OFF_S0)
        SC_ADJOINT(OFF_S1, 3)
        SC_PH_END()
        SC_ADJOINT_LOAD(2)                                          // phase 7: p2 = W3^T q3 (slot 0); W2h^T -> slot 1
        SC_DMA(8, OFF_S1)
        SC_ADJOINT(OFF_S0, 2)
        SC_PH_END()
        SC_ADJOINT_LOAD(1)                                          // phase 8: Jacobian with W2e, p1 = W2h^T q2 (slot 1); W1h^T -> slot 0
        SC_DMA(9, OFF_S0)
        SC_PE_JAC(OFF_PE2)
        SC_ADJOINT(OFF_S1, 1)
        SC_PH_END()
        SC_ADJOINT_LOAD(0)                                          // phase 9: Jacobian with W1e, p0 = W1h^T q1 (slot 0); next W2h -> slot 1
        SC_DMA(2, OFF_S1)
        SC_PE_JAC(OFF_PE1)
        SC_ADJOINT(OFF_S0, 0)
        SC_PH_END()
        SC_DMA(1, OFF_S0)                                           // phase 10: Jacobian with W0; next W1h -> slot 0
        SC_PE_JAC(OFF_PE0)
        g0 = group_sum(g0); g1 = group_sum(g1); g2 = group_sum(g2);
        if (a.grad && valid && g == 0) {
            a.grad[(size_t)pt * 3 + 0] = g0;
            a.grad[(size_t)pt * 3 + 1] = g1;
            a.grad[(size_t)pt * 3 + 2] = g2;
        }
        SC_PH_END()
#undef SC_ADJOINT
#undef SC_ADJOINT_LOAD
#undef SC_PE_JAC
#undef SC_PH_END
#undef SC_ACTIVATE
    }
#undef SC_DMA
}

}  // namespace st
}  // namespace sc

extern "C" long long sc_sdf_stream_pack_bytes(void) { return sc::st::IMG_BYTES; }

// SdfPack image (fp32) -> the streamed image of sc_sdf_forward_stream (sc_sdf_stream_pack_bytes() bytes); once per weight update.
extern "C" int sc_sdf_stream_pack(const float* w_pack, void* img, void* stream_) {
    if (!w_pack || !img) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sc::st::sdf_stream_pack_kernel, dim3(108), dim3(256), 0, (hipStream_t)stream_, w_pack, (char*)img);
    return (int)hipGetLastError();
}

// sc_sdf_forward (same operands, same outputs) with the weights streamed as pre-split fragments; w_stream = sc_sdf_stream_pack(w_pack).
// grad is required (the value-only chain is sc_sdf_value_forward_split); stash_a / stash_p / feat all given = the training render.
extern "C" int sc_sdf_forward_stream(const float* points, const void* w_stream, const float* w_pack, const float* cbias, int n_points,
                                     int n_per_image, int n_images, int symmetric, float* sdf, float* grad, float* feat,
                                     float* stash_a, float* stash_p, float* scratch, void* stream_) {
    if (n_points <= 0) return 0;
    if (!points || !w_stream || !w_pack || !cbias || !grad || (!stash_a && !scratch)) return (int)hipErrorInvalidValue;
    if ((stash_a != nullptr) != (stash_p != nullptr) || (stash_a && !feat)) return (int)hipErrorInvalidValue;      // the two compiled forms
    sc::st::Args a{points, (const char*)w_stream, w_pack, cbias, n_points, n_per_image, n_images, symmetric, sdf, grad, feat, stash_a, stash_p, scratch};
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + sc::st::WAVES - 1) / sc::st::WAVES;
    if (blocks > 256) blocks = 256;
    hipStream_t stream = (hipStream_t)stream_;
    if (stash_a) {
        (void)hipFuncSetAttribute((const void*)sc::st::sdf_fwd_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, sc::st::LDS_BYTES);
        hipLaunchKernelGGL((sc::st::sdf_fwd_stream_kernel<true>), dim3(blocks), dim3(64 * sc::st::WAVES), sc::st::LDS_BYTES, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)sc::st::sdf_fwd_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, sc::st::LDS_BYTES);
        hipLaunchKernelGGL((sc::st::sdf_fwd_stream_kernel<false>), dim3(blocks), dim3(64 * sc::st::WAVES), sc::st::LDS_BYTES, stream, a);
    }
    return (int)hipGetLastError();
}
