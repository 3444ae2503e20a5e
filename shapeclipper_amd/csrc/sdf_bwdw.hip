// sdf_bwdw.hip -- workgroup-cooperative reverse pass of the conditional SDF MLP: input gradients (first and
// second order) AND all weight / bias gradients in ONE launch, with no hand-off tensors in HBM.
//
// Replaces, for a render whose d(sdf)/dx output is differentiated (every training render: the reference reaches this
// through loss.backward() over the create_graph=True graph of model/renderer.py:101-107 / model/implicit.py:180-186),
// the pair  sdf_bwd.hip (chain) + 8 launches of wgrad.hip (addmm-backward GEMMs) + tbl_sum: that pair moved 8.2 + 9.5 KB
// per sample point through HBM (Ga_l, Gp_l, r0 written by one kernel and read back by the other, a_l / p_l read three
// times) for ~0 algorithmic bytes.  Here a 512-thread workgroup splits into two roles:
//
//   waves 0-3  "chain"  one 16-point tile each: the R sweep (reverse of the adjoint chain) and the V sweep (reverse of
//                       the value chain) of sdf_bwd.hip, in registers, weights read from the LDS image;
//   waves 4-7  "wgrad"  wave w owns rows 16w..16w+15 of EVERY weight-gradient matrix (116 accumulator registers) and
//                       adds  A(point)[row] * B(point)[col]  over the points of all four chain tiles with fp32 MFMAs.
//
// After each layer of a sweep the chain waves drop the operand pair of that layer (A = q_l | Ga_l | Gf, B = Gp_{l-1} |
// h_{l-1}; 64 channels x 16 points each) into a 4 KiB LDS slot in the layout the MFMA wants with K = point
// ([point/4][channel ^ point/4][point%4]: conflict-free ds_write_b32 on one side, conflict-free ds_read_b128 that
// yields the four K-steps of a channel on the other -- this IS the transpose wgrad.hip did with quad shuffles), two
// workgroup barriers bracket the write, and the wgrad waves consume the pair while the chain waves run the next layer.
// One chain wave and one wgrad wave share every SIMD: the matrix pipe alternates between the dependent MFMA chain of
// one and the independent outer-product MFMAs of the other (1008 + 864 per 16 points).  The positional-encoding
// operands (e, eps = Gg * dE/dx) are recomputed by the wgrad waves from the points (hardware sin/cos, as wgrad.hip).
// Per-image bias gradients (= latent gradients) are row sums of the A operands; W5 row 0 / b5 sums go through LDS.
// Every workgroup writes one partial image of the packed gradient; sc_partial_reduce sums them in a fixed order.
//
// HBM traffic: a_0..4, p_0..3, Gf read once per point (2.6 KB), g_points written (12 B); the parked second-order terms
// (pend_0..3, 1 KiB per layer and wave) live in a per-wave scratch that never leaves L2.
// LDS: weight image 118 KiB + 4 x 2 x 4 KiB exchange slots + point stash = 153 KiB of the 160 KiB.
// Bound: fp32 MFMA (1872 v_mfma_f32_16x16x4 per 16 points; 157.3 TFLOP/s dense peak).
#include "mlp_xch.hpp"

namespace sc {

struct SdfBwdwArgs {
    const float* points;   // [n_points][3]
    const float* w;        // SdfPack image
    int n_points, n_per_image, n_images, symmetric;
    const float* stash_a;  // 5 x TBL64
    const float* stash_p;  // 4 x TBL64
    const float* g_sdf;    // [n_points] or null
    const float* g_grad;   // [n_points][3]
    const float* g_feat;   // TBL64 or null
    float* g_points;       // [n_points][3] or null
    float* park;           // [gridDim.x * 4][4][1024] floats of per-wave scratch (L2-resident)
    float* partial;        // [gridDim.x][partial_stride]: one partial gradient image per workgroup (fully written): SdfPack::TOTAL floats
                           // of d/d(w_pack), then -- when cb_dense -- [n_images][5][64] of d/d(per-image biases)
    float* g_cbias;        // !cb_dense only: [n_images][5][64], zero-filled by the caller (float atomicAdd: order depends on timing)
    int partial_stride, cb_dense;
#ifdef SC_BWDW_PROFILE
    unsigned long long* prof;   // [8 waves][64] s_memtime stamps of one iteration of one workgroup (tools/prof_bwdw.py)
    int prof_block, prof_iter;
#endif
};
#ifdef SC_BWDW_PROFILE
#define BW_STAMP(ID) if (prof_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) a.prof[wave * 64 + (ID)] = t_; }
#else
#define BW_STAMP(ID)
#endif

constexpr int BW_CHAIN = 4;                          // chain waves (= wgrad waves) per workgroup
constexpr int BW_WLDS = (SdfLds::TOTAL + 3) & ~3;    // weight image, floats
constexpr int BW_XCH = BW_WLDS;                      // exchange slots: [chain wave][A|B][1024]
constexpr int BW_PTS = BW_XCH + BW_CHAIN * 2 * 1024; // point stash: [chain wave][16 points][8] = x0 x1 x2 gam0 gam1 gam2 valid -
constexpr int BW_RED = BW_PTS + BW_CHAIN * 16 * 8;   // [0..63] sum r0 (dW5 row 0), [64] sum Gs (db5[0])
constexpr int BW_LDS_FLOATS = BW_RED + 68;
static_assert(BW_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(512, 2) void sdf_bwdw_kernel(SdfBwdwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_sdf_weights(lds, a.w, tid, 512);
    for (int e = tid; e < BW_LDS_FLOATS - BW_XCH; e += 512) lds[BW_XCH + e] = 0.f;      // slots, point stash, sums
    // Per-image bias gradients: with cb_dense every workgroup owns a zero-filled [n_images][5][64] block behind its weight-gradient
    // image and ONE lane (wave w, lane i, kg == 0) is the only writer of an address, so plain read-modify-writes replace the float
    // atomics; sc_partial_reduce then adds the workgroups' blocks in index order (results independent of timing).
    float* cbp = a.partial + (size_t)blockIdx.x * a.partial_stride + SdfPack::TOTAL;
    if (a.cb_dense)
        for (int e = tid; e < a.n_images * 320; e += 512) cbp[e] = 0.f;
    __syncthreads();

    const int ntiles = (a.n_points + TP - 1) / TP;
    const size_t tbl = (size_t)ntiles * 1024;
    // every workgroup owns one contiguous range of tiles (a wave changes image rarely: one flush of the bias sums per change)
    const int per_wg = ((ntiles + (int)gridDim.x - 1) / (int)gridDim.x + BW_CHAIN - 1) & ~(BW_CHAIN - 1);
    const int t_begin = blockIdx.x * per_wg, t_end = min(ntiles, t_begin + per_wg);
    const int tiles_per_image = a.n_per_image / TP;
    const bool symmetric = a.symmetric != 0;

    if (wave < BW_CHAIN) {
        // =====================================================================================================
        // chain role
        // =====================================================================================================
        const int cw = __builtin_amdgcn_readfirstlane(wave), p = lane & 15, g = lane >> 4;     // cw in an SGPR: scalar tile addressing
        float* slotA = lds + BW_XCH + (cw * 2 + 0) * 1024;
        float* slotB = lds + BW_XCH + (cw * 2 + 1) * 1024;
        float* ptsw = lds + BW_PTS + cw * 16 * 8;
        int wr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r] = (((p >> 2) * 64 + 4 * g + (r ^ (p >> 2))) << 2) + (p & 3);
        float* park = a.park + (size_t)(blockIdx.x * BW_CHAIN + cw) * 4 * 1024;

        // Seven lane-dependent LDS offsets (made opaque so that hipcc keeps ONE register each and puts the matrix / column constants
        // into the 16-bit immediate of the ds_read -- it otherwise hoists a full address per matrix and orientation out of the tile
        // loop, 13 registers that end up in scratch; every group of matrices below spans less than 64 KiB from its first one).
        int o52 = p * SdfLds::LD0 + g, o116h = p * SdfLds::LD1 + 4 * g, o116e = p * SdfLds::LD1 + g, o68 = p * SdfLds::LD3 + 4 * g,
            o4g = 4 * g, c68 = 4 * g * SdfLds::LD3 + p, c116 = 4 * g * SdfLds::LD1 + p;
        asm volatile("" : "+v"(o52), "+v"(o116h), "+v"(o116e), "+v"(o68), "+v"(o4g), "+v"(c68), "+v"(c116));
        const float* w0 = lds + SdfLds::W0 + o52;
        const float* w1h = lds + SdfLds::W1 + o116h;
        const float* w1e = lds + SdfLds::W1 + 64 + o116e;
        const float* w2h = lds + SdfLds::W2 + o116h;
        const float* w2e = lds + SdfLds::W2 + 64 + o116e;
        const float* w3 = lds + SdfLds::W3 + o68;
        const float* w4 = lds + SdfLds::W4 + o68;
        const float* w5s = lds + SdfLds::W5 + o4g;
        const float* w5ft = lds + SdfLds::W5 + SdfLds::LD3 + c68;
        const float* w4t = lds + SdfLds::W4 + c68;
        const float* w3t = lds + SdfLds::W3 + c68;
        const float* w2t = lds + SdfLds::W2 + c116;
        const float* w1t = lds + SdfLds::W1 + c116;

// the write phase of one step: B1 (the wgrad waves have finished reading the previous pair), write, B2 (visible).
// VM is the validity mask of the lane's point: a compile-time 1 for full tiles (all but the last tile of a launch)
// (Round 5 experiment, commit c7425e8: the six point-gradient terms -- 288 of the chain waves' 1,008 MFMAs per tile, all of d L / d point --
// moved to the weight-gradient waves, which wait half of the time and already receive the terms' V operands (q_l, Ga_l) in slot A.  Correct,
// 17 % SLOWER: 3.03 vs 2.59 ms, profiles/r05_bwdw_gxw_experiment.txt -- with the PE derivatives of four tiles and the extra accumulators the
// weight-gradient role spills 89 registers inside its loops; the role is register-bound, not time-bound.)
// (Round 5 experiment, commit 2093f09: the two barriers replaced by per-slot LDS counters -- each chain wave hands over at its own pace, the
// wgrad waves walk the four slots as a queue.  Correct, 4 % SLOWER: 2.69 vs 2.59 ms; profiles/r05_bwdw_phase_profile_flags_experiment.txt:
// the wgrad waves then wait for each slot's flag and fragments in front of its MFMAs instead of reading ahead across the four tiles.)
#define BW_EXCHANGE(K, WRITES)                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        BW_STAMP(4 * (K) + 1)                                                                \
        lds_barrier();                                                                       \
        BW_STAMP(4 * (K) + 2)                                                                \
        if (full) { constexpr float VM = 1.f; WRITES } else { const float VM = vmask; WRITES } \
        lds_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        BW_STAMP(4 * (K) + 3)

        // The per-tile inputs -- point, upstream gradients, layer-0 stash -- are fetched ONE TILE AHEAD (issued before the last two
        // exchanges of the previous tile, whose waits hide the HBM round trip).  Buffer loads: the 16 points of a tile are one
        // bounds-checked resource (lanes past n_points read 0 and are masked below).
        float nx[3] = {0.f, 0.f, 0.f}, ngam[3] = {0.f, 0.f, 0.f}, nGs = 0.f;
#define BW_FETCH(TILE)                                                                      \
        {                                                                                    \
            const int ft_ = __builtin_amdgcn_readfirstlane(TILE);                             \
            const int left_ = min(TP, a.n_points - ft_ * TP);                                 \
            const __amdgpu_buffer_rsrc_t rp_ = __builtin_amdgcn_make_buffer_rsrc(             \
                const_cast<float*>(a.points) + (size_t)ft_ * TP * 3, 0, left_ * 12, 0x00020000); \
            const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(             \
                const_cast<float*>(a.g_grad) + (size_t)ft_ * TP * 3, 0, left_ * 12, 0x00020000); \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                   \
                nx[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp_, p * 12, 4 * c, 0));   \
                ngam[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg_, p * 12, 4 * c, 0)); \
            }                                                                                \
            nGs = 0.f;                                                                       \
            if (a.g_sdf) {                                                                   \
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(         \
                    const_cast<float*>(a.g_sdf) + (size_t)ft_ * TP, 0, left_ * 4, 0x00020000); \
                nGs = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, p * 4, 0, 0)); \
            }                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                               \
        }
        if (t_begin + cw < t_end) BW_FETCH(t_begin + cw)
#ifdef SC_BWDW_PROFILE
        int prof_it = 0;
#endif
#pragma unroll 1
        for (int base = t_begin; base < t_end; base += BW_CHAIN) {
#ifdef SC_BWDW_PROFILE
            const bool prof_on = (int)blockIdx.x == a.prof_block && prof_it++ == a.prof_iter;
#endif
            BW_STAMP(60)
            const int tile = base + cw;
            if (tile >= t_end) {            // tail: nothing to do but keep the barrier count (11 steps) and feed zeros
                for (int k = 0; k < 11; ++k) {
                    lds_barrier();
                    if (k == 0) {
                        xch_zero(slotA, lane); xch_zero(slotB, lane);
                        if (lane < 32) reinterpret_cast<float4*>(ptsw)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    lds_barrier();
                }
                continue;
            }
            const int pt = tile * TP + p;
            const bool valid = pt < a.n_points;
            const float vmask = valid ? 1.f : 0.f;
            const bool full = tile * TP + TP <= a.n_points;                // wave-uniform
            const float x0 = nx[0], x1 = nx[1], x2 = nx[2], Gs = nGs;       // (zeros for the lanes past n_points)
            const float gam[3] = {ngam[0], ngam[1], ngam[2]};
            float j_av[ACT_STEPS], j_pend[ACT_STEPS], j_u[ACT_STEPS];      // R -> V junction registers
            float r0v[ACT_STEPS];                                          // dW5 row-0 operand: leaves with step 10
            float gx[3] = {0.f, 0.f, 0.f};                                 // d L / d point
// Point-gradient terms that need only one operand of a step and the PE derivatives of the point:
//   gx_c += sum_ch V[ch] * (W_le * DV[4c..4c+3])[ch]     V = q_l, DV = Gg_c d2E/dx_c^2 (steps 0-2);  V = Ga_l, DV = dE/dx_c (layers 2-0)
// They sit where this wave would otherwise wait for the wgrad waves (which carry the PE outer products of the same steps).
#define BW_PE_DOT2(WE, LD, V)                                                               \
            if (a.g_points) {                                                                \
                _Pragma("unroll") for (int c = 0; c < 3; ++c) {                               \
                    f32x4 tacc[NT];                                                          \
                    acc_zero(tacc);                                                          \
                    BW_D2(c)                                                                 \
                    if (c == 0) mm_pe<LD, NT, 0, 4>(WE, dv, tacc);                           \
                    if (c == 1) mm_pe<LD, NT, 4, 4>(WE, dv, tacc);                           \
                    if (c == 2) mm_pe<LD, NT, 8, 4>(WE, dv, tacc);                           \
                    float dsum = 0.f;                                                        \
                    _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s)                     \
                        dsum = __builtin_fmaf(V[s], tacc[s >> 2][s & 3], dsum);              \
                    gx[c] += dsum;                                                           \
                }                                                                            \
            }
#define BW_PE_DOT(WE, LD, V, DV)                                                            \
            if (a.g_points) {                                                                \
                _Pragma("unroll") for (int c = 0; c < 3; ++c) {                               \
                    f32x4 tacc[NT];                                                          \
                    acc_zero(tacc);                                                          \
                    if (c == 0) mm_pe<LD, NT, 0, 4>(WE, DV + 0, tacc);                       \
                    if (c == 1) mm_pe<LD, NT, 4, 4>(WE, DV + 4, tacc);                       \
                    if (c == 2) mm_pe<LD, NT, 8, 4>(WE, DV + 8, tacc);                       \
                    float dsum = 0.f;                                                        \
                    _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s)                     \
                        dsum = __builtin_fmaf(V[s], tacc[s >> 2][s & 3], dsum);              \
                    gx[c] += dsum;                                                           \
                }                                                                            \
            }
            // ================= R sweep =================
            {
                float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS], eps[PE_STEPS];
                pe_slots<true, false, true>(x0, x1, x2, g, symmetric, e, d1, d2);
#pragma unroll
                for (int j = 0; j < PE_STEPS; ++j) eps[j] = gam[j >> 2] * d1[j];
                // Gg_c d2E/dx_c^2 of the point terms is rebuilt from eps where it is used (12 registers less across the sweep): a
                // slot pair is (sin, cos) of one argument f |x_c|, so  d2 sin = -f^2 sin sg^2 = +f sg * (d1 cos)  and
                // d2 cos = -f sg * (d1 sin);  raw-coordinate lanes (g == 3) have no second derivative.
                const float fb = g == 3 ? 0.f : (g == 0 ? 1.f : (g == 1 ? 4.f : 16.f));
                const float fb0 = (symmetric ? (x0 > 0.f ? 1.f : (x0 < 0.f ? -1.f : 0.f)) : 1.f) * fb;
#define BW_D2(C)                                                                            \
                const float fs_ = (C) == 0 ? fb0 : fb;                                       \
                const float dv[4] = {fs_ * eps[4 * (C) + 1], -fs_ * eps[4 * (C)], 2.f * fs_ * eps[4 * (C) + 3], -2.f * fs_ * eps[4 * (C) + 2]};
                f32x4 acc[NT];
                float avA[ACT_STEPS], pvA[ACT_STEPS], avB[ACT_STEPS], pvB[ACT_STEPS], gpA[ACT_STEPS], gpB[ACT_STEPS];
#define BW_R_LOAD(L, av, pv)                                                                \
                tbl_load(a.stash_a + (size_t)(L) * tbl, tile, p, g, av);                     \
                tbl_load(a.stash_p + (size_t)(L) * tbl, tile, p, g, pv);                     \
                __builtin_amdgcn_sched_barrier(0);
// element-wise part of R layer L: acc = Gq_L -> gpn = Gp_L, pend_L parked, pv <- q_L
#define BW_R_ELEM(L, av, pv, gpn)                                                           \
                {                                                                            \
                    float pn[ACT_STEPS];                                                     \
                    _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                   \
                        float t, r;                                                          \
                        stash_parts(av[s], t, r);                                         \
                        const float ds = stash_d1(av[s], t, r);                           \
                        const float gq = acc[s >> 2][s & 3];                                 \
                        gpn[s] = gq * ds;                                                    \
                        pn[s] = gq * pv[s] * stash_d2(t, r);                              \
                        pv[s] = pv[s] * ds;                                                  \
                    }                                                                        \
                    tbl_store(park + (size_t)(L) * 1024, 0, p, g, pn);                       \
                }
                acc_zero(acc);
                BW_R_LOAD(0, avA, pvA)
                BW_R_LOAD(1, avB, pvB)
                mm_pe<SdfLds::LD0, NT, 0, PE_STEPS>(w0, eps, acc);                 // Gq0
                BW_STAMP(0)
                BW_R_ELEM(0, avA, pvA, gpA)
                BW_PE_DOT2(w0, SdfLds::LD0, pvA)
                BW_EXCHANGE(0,                                                         // step 0: A = q0 (pairs with eps)
                    xch_write(slotA, wr, pvA, VM);
                    if (g == 0) {
                        *reinterpret_cast<float4*>(ptsw + p * 8) = make_float4(x0, x1, x2, gam[0]);
                        *reinterpret_cast<float4*>(ptsw + p * 8 + 4) = make_float4(gam[1], gam[2], VM, Gs);
                    })
                acc_zero(acc);
                BW_R_LOAD(2, avA, pvA)
                mm_act<SdfLds::LD1, NT>(w1h, gpA, acc);
                mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w1e, eps, acc);                // Gq1
                BW_STAMP(4)
                BW_R_ELEM(1, avB, pvB, gpB)
                BW_PE_DOT2(w1e, SdfLds::LD1, pvB)
                BW_EXCHANGE(1, xch_write(slotA, wr, pvB, VM); xch_write(slotB, wr, gpA, VM);)      // step 1: (q1, Gp0)
                acc_zero(acc);
                BW_R_LOAD(3, avB, pvB)
                mm_act<SdfLds::LD1, NT>(w2h, gpB, acc);
                mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w2e, eps, acc);                // Gq2
                BW_STAMP(8)
                BW_R_ELEM(2, avA, pvA, gpA)
                BW_PE_DOT2(w2e, SdfLds::LD1, pvA)
                BW_EXCHANGE(2, xch_write(slotA, wr, pvA, VM); xch_write(slotB, wr, gpB, VM);)      // step 2: (q2, Gp1)
                acc_zero(acc);
                tbl_load(a.stash_a + 4 * tbl, tile, p, g, j_av);
                __builtin_amdgcn_sched_barrier(0);
                mm_act<SdfLds::LD3, NT>(w3, gpA, acc);                             // Gq3
                BW_STAMP(12)
                BW_R_ELEM(3, avB, pvB, gpB)
                BW_EXCHANGE(3, xch_write(slotA, wr, pvB, VM); xch_write(slotB, wr, gpA, VM);)      // step 3: (q3, Gp2)
                acc_zero(acc);
                mm_act<SdfLds::LD3, NT>(w4, gpB, acc);                             // Gq4
                BW_STAMP(16)
                {
                    float q4[ACT_STEPS];
#pragma unroll
                    for (int s = 0; s < ACT_STEPS; ++s) {
                        float t, r;
                        stash_parts(j_av[s], t, r);
                        const float ds = stash_d1(j_av[s], t, r), gq = acc[s >> 2][s & 3], w5 = w5s[kp(s)];
                        j_pend[s] = gq * w5 * stash_d2(t, r);
                        j_u[s] = gq * ds;
                        q4[s] = w5 * ds;
                    }
                    BW_EXCHANGE(4, xch_write(slotA, wr, q4, VM); xch_write(slotB, wr, gpB, VM);)   // step 4: (q4, Gp3)
                }
#undef BW_R_ELEM
#undef BW_R_LOAD
            }
            __builtin_amdgcn_sched_barrier(0);
            // ================= V sweep =================
            {
                f32x4 acc[NT];
                float av[ACT_STEPS], pv[ACT_STEPS], avB[ACT_STEPS], pvB[ACT_STEPS], gaA[ACT_STEPS], gaB[ACT_STEPS], hv[ACT_STEPS];
#define BW_V_LOAD(L, av, pv)                                                                \
                tbl_load(a.stash_a + (size_t)(L) * tbl, tile, p, g, av);                     \
                tbl_load(park + (size_t)(L) * 1024, 0, p, g, pv);                            \
                __builtin_amdgcn_sched_barrier(0);
// V layer L: acc = W_{L+1}^T Ga_{L+1} -> gan = Ga_L = acc * sp'(a_L) + pend_L,  hv = h_L = sp(a_L)
#define BW_V_ELEM(av, pv, gan)                                                              \
                _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                       \
                    float t, r;                                                              \
                    stash_parts(av[s], t, r);                                             \
                    gan[s] = acc[s >> 2][s & 3] * stash_d1(av[s], t, r) + pv[s];          \
                    hv[s] = stash_val(av[s], t);                                          \
                }
                acc_zero(acc);
                float gf[ACT_STEPS];
                if (a.g_feat) {
                    tbl_load(a.g_feat, tile, p, g, gf);
                    mm_act_t_pipe<SdfLds::LD3, NT>(w5ft, gf, acc);
                } else {
#pragma unroll
                    for (int s = 0; s < ACT_STEPS; ++s) gf[s] = 0.f;
                }
                BW_V_LOAD(3, av, pv)
                BW_STAMP(20)
                {
#pragma unroll
                    for (int s = 0; s < ACT_STEPS; ++s) {
                        float t, r;
                        stash_parts(j_av[s], t, r);
                        const float gh = acc[s >> 2][s & 3] + w5s[kp(s)] * Gs;
                        hv[s] = stash_val(j_av[s], t);
                        r0v[s] = (Gs * hv[s] + j_u[s]) * vmask;
                        gaA[s] = gh * stash_d1(j_av[s], t, r) + j_pend[s];
                    }
                    // dW5 row 0 = sum over points of r0 and db5[0] = sum Gs are row sums the wgrad waves take on the way: r0 rides in
                    // the idle slot B of step 10, Gs in the point stash.  (They used to be reduced here with DPP adds and 16 LDS
                    // atomics whose spilled addresses came back through `s_waitcnt vmcnt(0)`: 13 k cycles per tile, measured.)
                }
                BW_EXCHANGE(5, xch_write(slotA, wr, gf, VM); xch_write(slotB, wr, hv, VM);)        // step 5: (Gf, h4)
                acc_zero(acc);
                mm_act_t_pipe<SdfLds::LD3, NT>(w4t, gaA, acc);
                BW_V_LOAD(2, avB, pvB)          // (after the MFMAs: a scratch reload in front of them would drain the fresh loads)
                BW_STAMP(24)
                BW_V_ELEM(av, pv, gaB)                                                              // Ga3, h3
                BW_EXCHANGE(6, xch_write(slotA, wr, gaA, VM); xch_write(slotB, wr, hv, VM);)       // step 6: (Ga4, h3)
                acc_zero(acc);
                mm_act_t_pipe<SdfLds::LD3, NT>(w3t, gaB, acc);
                BW_V_LOAD(1, av, pv)
                BW_STAMP(28)
                BW_V_ELEM(avB, pvB, gaA)                                                            // Ga2, h2
                BW_EXCHANGE(7, xch_write(slotA, wr, gaB, VM); xch_write(slotB, wr, hv, VM);)       // step 7: (Ga3, h2)
                acc_zero(acc);
                mm_act_t_pipe<SdfLds::LD1, NT>(w2t, gaA, acc);
                BW_V_LOAD(0, avB, pvB)
                BW_STAMP(32)
                BW_V_ELEM(av, pv, gaB)                                                              // Ga1, h1
                BW_EXCHANGE(8, xch_write(slotA, wr, gaA, VM); xch_write(slotB, wr, hv, VM);)       // step 8: (Ga2, h1)
                float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
                pe_slots<true, false, true>(x0, x1, x2, g, symmetric, e, d1, d2);
                BW_PE_DOT(w2e, SdfLds::LD1, gaA, d1)                                                // Ga2 (gaA is overwritten below)
                acc_zero(acc);
                mm_act_t_pipe<SdfLds::LD1, NT>(w1t, gaB, acc);
                BW_STAMP(36)
                BW_V_ELEM(avB, pvB, gaA)                                                            // Ga0, h0
                if (tile + BW_CHAIN < t_end) BW_FETCH(tile + BW_CHAIN)                              // next tile's inputs
                BW_EXCHANGE(9, xch_write(slotA, wr, gaB, VM); xch_write(slotB, wr, hv, VM);)       // step 9: (Ga1, h0)
                BW_PE_DOT(w1e, SdfLds::LD1, gaB, d1)                                                // Ga1
                BW_PE_DOT(w0, SdfLds::LD0, gaA, d1)                                                 // Ga0
                if (a.g_points) {
                    const float o0 = group_sum(gx[0]), o1 = group_sum(gx[1]), o2 = group_sum(gx[2]);
                    if (valid && g == 0) {
                        a.g_points[(size_t)pt * 3 + 0] = o0;
                        a.g_points[(size_t)pt * 3 + 1] = o1;
                        a.g_points[(size_t)pt * 3 + 2] = o2;
                    }
                }
                BW_EXCHANGE(10, xch_write(slotA, wr, gaA, VM); xch_write(slotB, wr, r0v, 1.f);)      // step 10: Ga0 (pairs with e) | r0
#undef BW_V_ELEM
#undef BW_V_LOAD
            }
        }
#undef BW_EXCHANGE
#undef BW_FETCH
#undef BW_PE_DOT
#undef BW_PE_DOT2
#undef BW_D2
        __syncthreads();        // the wgrad waves' sum of Gs is in LDS
        if (tid == 0)       // db5[0] = sum of Gs: the four wgrad waves' sums, added in wave order
            a.partial[(size_t)blockIdx.x * a.partial_stride + SdfPack::B5] =
                (lds[BW_RED + 64] + lds[BW_RED + 65]) + (lds[BW_RED + 66] + lds[BW_RED + 67]);
    } else {
        // =====================================================================================================
        // wgrad role: wave w owns rows 16w..16w+15 of every matrix
        // =====================================================================================================
        const int w = wave - BW_CHAIN, i = lane & 15, kg = lane >> 4;
        const int rd = (kg * 64 + (i ^ kg)) << 2;            // this lane's float4 chunk of channel tile 0 (+ 64 floats per tile)
        const float* ptsw = lds + BW_PTS + w * 16 * 8;
        f32x4 d0e[3], d1h[4], d1e[3], d2h[4], d2e[3], d3[4], d4[4], d5[4];
        acc_zero(d0e); acc_zero(d1h); acc_zero(d1e); acc_zero(d2h); acc_zero(d2e); acc_zero(d3); acc_zero(d4); acc_zero(d5);
        float rs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};            // per-image bias-gradient partials (this lane's 4 points of every tile)
        float rsf = 0.f;                                     // sum over all points of Gf (db5 feature rows)
        float rs0 = 0.f, gss = 0.f;                          // sum over all points of r0 (dW5 row 0, channel 16w + i) and of Gs (db5[0], tile w)
        // The positional-encoding operand of a tile is the same in steps 0-2 (eps) and in steps 8-10 (E): it is evaluated once per run of
        // three steps and kept in 48 registers (4 tiles x 3 fragments) instead of six evaluations per tile (round 5: 2.77 -> 2.59 ms per
        // launch on one box, profiles/r05_bwdw_variants_ab.txt -- these ~50 vector instructions per tile and step were issue time of the
        // SIMD the chain wave shares, and they sat in exactly the steps where the chain waves waited for this role).
        float4 pec[BW_CHAIN][3];
        int cur_img = -1;                                    // >= 0: rs[] belongs to this image; -2: mixed iteration (direct atomics)
        auto flush = [&]() {
            if (cur_img >= 0) {
#pragma unroll
                for (int l = 0; l < 5; ++l) {
                    float v = rs[l];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (kg == 0) {
                        const size_t o = ((size_t)cur_img * 5 + l) * 64 + 16 * w + i;
                        if (a.cb_dense) cbp[o] += v; else atomicAdd(&a.g_cbias[o], v);
                    }
                    rs[l] = 0.f;
                }
            }
        };
#ifdef SC_BWDW_PROFILE
#define BW_STEP_BEGIN BW_STAMP(prof_k) lds_barrier(); lds_barrier(); BW_STAMP(prof_k + 1) prof_k += 2;
#else
#define BW_STEP_BEGIN lds_barrier(); lds_barrier();
#endif
// one step over the four chain tiles.  HP: 64-wide B operand in slot B; PEM: 0 none, 1 E, 2 eps; RS: bias layer (-1 none, 5 = Gf)
#define BW_CONSUME(ACCH, ACCE, HP, PEM, RS)                                                 \
        _Pragma("unroll") for (int c = 0; c < BW_CHAIN; ++c) {                               \
            const float* sA = lds + BW_XCH + (c * 2 + 0) * 1024;                             \
            const float* sB = lds + BW_XCH + (c * 2 + 1) * 1024;                             \
            const float4 af = xch_frag(sA, rd, w);                                           \
            if (HP) {                                                                        \
                float4 bf[4];                                                                \
                _Pragma("unroll") for (int n = 0; n < 4; ++n) bf[n] = xch_frag(sB, rd, n);   \
                outer16<4>(af, bf, ACCH);                                                    \
            }                                                                                \
            if (PEM) {       /* PEM > 0: evaluate the operand (1 = E, 2 = eps); PEM < 0: the one of the previous step again */ \
                if (PEM > 0) pe_frags<((PEM) > 0 ? (PEM) : 1)>(lds + BW_PTS + c * 16 * 8, i, kg, symmetric, pec[c]); \
                outer16<3>(af, pec[c], ACCE);                                                \
            }                                                                                \
            if (RS >= 0) {                                                                   \
                const float v = (af.x + af.y) + (af.z + af.w);                               \
                if (RS == 5) rsf += v;                                                       \
                else if (cur_img >= 0) rs[(RS >= 0 && RS < 5) ? RS : 0] += v;                             \
                else {                                                                       \
                    float t = v;                                                             \
                    t += __shfl_xor(t, 16);                                                  \
                    t += __shfl_xor(t, 32);                                                  \
                    const int img = min((base + c) / tiles_per_image, a.n_images - 1);       \
                    if (kg == 0 && base + c < t_end) {                                       \
                        const size_t o = ((size_t)img * 5 + ((RS >= 0 && RS < 5) ? RS : 0)) * 64 + 16 * w + i; \
                        if (a.cb_dense) cbp[o] += t; else atomicAdd(&a.g_cbias[o], t);       \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
        }
#ifdef SC_BWDW_PROFILE
        int prof_it = 0, prof_k = 0;
#endif
#pragma unroll 1
        for (int base = t_begin; base < t_end; base += BW_CHAIN) {
#ifdef SC_BWDW_PROFILE
            const bool prof_on = (int)blockIdx.x == a.prof_block && prof_it++ == a.prof_iter;
            prof_k = 0;
#endif
            BW_STAMP(60)
            const int img0 = min(base / tiles_per_image, a.n_images - 1);
            const int img3 = min(min(base + BW_CHAIN - 1, t_end - 1) / tiles_per_image, a.n_images - 1);
            if (img0 != cur_img || img3 != img0) {
                flush();
                cur_img = img0 == img3 ? img0 : -2;
            }
            BW_STEP_BEGIN
            gss += kg == 0 ? ptsw[i * 8 + 7] : 0.f;                 // db5[0] = sum of Gs (point i of chain tile w; stash written in step 0)
            BW_CONSUME(d3, d0e, false, 2, -1)                   // 0: q0 x eps
            BW_STEP_BEGIN BW_CONSUME(d1h, d1e, true, -2, -1)       // 1: q1 x (Gp0 | eps)     (PEM < 0: the operand of the previous step again)
            BW_STEP_BEGIN BW_CONSUME(d2h, d2e, true, -2, -1)       // 2: q2 x (Gp1 | eps)
            BW_STEP_BEGIN BW_CONSUME(d3, d0e, true, 0, -1)         // 3: q3 x Gp2
            BW_STEP_BEGIN BW_CONSUME(d4, d0e, true, 0, -1)         // 4: q4 x Gp3
            BW_STEP_BEGIN BW_CONSUME(d5, d0e, true, 0, 5)          // 5: Gf x h4
            BW_STEP_BEGIN BW_CONSUME(d4, d0e, true, 0, 4)          // 6: Ga4 x h3
            BW_STEP_BEGIN BW_CONSUME(d3, d0e, true, 0, 3)          // 7: Ga3 x h2
            BW_STEP_BEGIN BW_CONSUME(d2h, d2e, true, 1, 2)         // 8: Ga2 x (h1 | E)
            BW_STEP_BEGIN BW_CONSUME(d1h, d1e, true, -1, 1)        // 9: Ga1 x (h0 | E)
            BW_STEP_BEGIN BW_CONSUME(d3, d0e, false, -1, 0)        // 10: Ga0 x E
#pragma unroll
            for (int c = 0; c < BW_CHAIN; ++c) {                    //     slot B of step 10 carries r0: its row sums are dW5 row 0
                const float4 rf = xch_frag(lds + BW_XCH + (c * 2 + 1) * 1024, rd, w);
                rs0 += (rf.x + rf.y) + (rf.z + rf.w);
            }
            BW_STAMP(prof_k)
        }
#undef BW_CONSUME
#undef BW_STEP_BEGIN
        flush();
        {
            const float gs = row_sum16(gss);
            if (lane == 0) lds[BW_RED + 64 + w] = gs;
        }
        __syncthreads();
        // ---- this workgroup's partial image: rows 16w + 4kg + r, columns 16n + i of every matrix ----
        float* out = a.partial + (size_t)blockIdx.x * a.partial_stride;
        {
            float v0 = rs0;
            v0 += __shfl_xor(v0, 16);
            v0 += __shfl_xor(v0, 32);
            if (kg == 0) out[SdfPack::W5 + 16 * w + i] = v0;
        }
        auto put = [&](const f32x4* acc, int ntile, int off, int ld, int col0) {
            for (int n = 0; n < ntile; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[off + (16 * w + 4 * kg + r) * ld + col0 + 16 * n + i] = acc[n][r];
        };
        put(d0e, 3, SdfPack::W0, 48, 0);
        put(d1h, 4, SdfPack::W1, 112, 0);
        put(d1e, 3, SdfPack::W1, 112, 64);
        put(d2h, 4, SdfPack::W2, 112, 0);
        put(d2e, 3, SdfPack::W2, 112, 64);
        put(d3, 4, SdfPack::W3, 64, 0);
        put(d4, 4, SdfPack::W4, 64, 0);
        put(d5, 4, SdfPack::W5 + 64, 64, 0);
        float v = rsf;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kg == 0) out[SdfPack::B5 + 1 + 16 * w + i] = v;
    }
}


}  // namespace sc

extern "C" {

#ifdef SC_BWDW_PROFILE
static unsigned long long* sc_bwdw_prof_buf = nullptr;
static int sc_bwdw_prof_block = -1, sc_bwdw_prof_iter = -1;
void sc_bwdw_set_prof(unsigned long long* buf, int block, int iter) { sc_bwdw_prof_buf = buf; sc_bwdw_prof_block = block; sc_bwdw_prof_iter = iter; }
#endif

// Number of workgroups (= partial images, = 4-wave park slots / 4) sc_sdf_backward_fused launches for n_points.
int sc_sdf_backward_fused_parts(int n_points) {
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + sc::BW_CHAIN - 1) / sc::BW_CHAIN;
    return blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks);
}

// Floats per partial image of sc_sdf_backward_fused: the packed weight-gradient image, followed -- for up to 256 images -- by the
// [n_images][5][64] per-image bias gradients (fixed summation order; beyond that they go through float atomics into g_cbias).
int sc_sdf_backward_fused_partial_floats(int n_images) {
    return sc::SdfPack::TOTAL + (n_images > 0 && n_images <= 256 ? n_images * 320 : 0);
}

int sc_sdf_backward_fused(const float* points, const float* w_pack, int n_points, int n_per_image, int n_images, int symmetric,
                          const float* stash_a, const float* stash_p, const float* g_sdf, const float* g_grad,
                          const float* g_feat, float* g_points, float* park, float* partial, float* g_cbias, void* stream_) {
    if (n_points <= 0) return 0;
    if (!g_grad || !stash_p || n_per_image <= 0 || n_per_image % sc::TP != 0 || n_images <= 0) return (int)hipErrorInvalidValue;
    const int stride = sc_sdf_backward_fused_partial_floats(n_images), dense = stride > sc::SdfPack::TOTAL;
    if (!dense && !g_cbias) return (int)hipErrorInvalidValue;
    sc::SdfBwdwArgs a{points, w_pack, n_points, n_per_image, n_images, symmetric, stash_a, stash_p, g_sdf, g_grad, g_feat,
                      g_points, park, partial, g_cbias, stride, dense};
#ifdef SC_BWDW_PROFILE
    a.prof = sc_bwdw_prof_buf; a.prof_block = sc_bwdw_prof_block; a.prof_iter = sc_bwdw_prof_iter;
#endif
    const int blocks = sc_sdf_backward_fused_parts(n_points);
    const size_t lds_bytes = (size_t)sc::BW_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)sc::sdf_bwdw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(sc::sdf_bwdw_kernel, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

}  // extern "C"
