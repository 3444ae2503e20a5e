// sdf_bwd.hip -- reverse pass of sdf_fwd.hip, including the second-order terms.
//
// Replaces what autograd does for SDFNetwork.get_conditional_output (model/implicit.py:163-189) when the
// outputs (sdf, feature, d sdf/dx) are all differentiated: the reference reaches this through
// loss.backward() over a graph built with create_graph=True (model/renderer.py:101-107,
// implicit.py:180-186) -- a double backward through five softplus layers.  Here it is one
// hand-derived reverse sweep over the value chain *and* the adjoint chain of sdf_fwd.hip:
//
//   upstream:  Gs = dL/dsdf, Gg = dL/d(dsdf/dx) (3), Gf = dL/dfeat (64)
//   R pass (reverse of the adjoint chain, only when Gg is given), eps_j = Gg_c(j) * dE_j/dx:
//     Gq0 = W0e eps;                    Gp0 = Gq0*sp'(a0);  pend0 = Gq0*p0*sp''(a0)
//     Gq1 = W1h Gp0 + W1e eps;          Gp1 = ...           pend1 = Gq1*p1*sp''(a1)   (same for 2)
//     Gq3 = W3 Gp2; Gq4 = W4 Gp3;       pend4 = Gq4*w5*sp''(a4);  u4 = Gq4*sp'(a4)
//     Gx_c += Gg_c * sum_l q_l . (W_le d2E/dx_c^2)                  (l = 0,1,2; q_l = p_l*sp'(a_l))
//   V pass (reverse of the value chain):
//     Gh4 = W5f^T Gf + w5*Gs;  Ga4 = Gh4*sp'(a4) + pend4;  Gh3 = W4^T Ga4; ... ; Ga0
//     Gx_c += sum_l Ga_l . (W_le dE/dx_c)
//   written for the weight-gradient GEMMs (wgrad.hip): Ga_0..4, Gp_0..3, r0 = Gs*h4 + u4.
// a_l and p_l come from the forward stash (TBL64).  1008 v_mfma_f32_16x16x4 per 16 points.
#include "mlp_tile.hpp"

namespace sc {

struct SdfBwdArgs {
    const float* points;   // [n_points][3]
    const float* w;        // SdfPack image
    int n_points, symmetric;
    const float* stash_a;  // 5 x TBL64
    const float* stash_p;  // 4 x TBL64 (required when g_grad)
    const float* g_sdf;    // [n_points] or null
    const float* g_grad;   // [n_points][3] or null
    const float* g_feat;   // TBL64 or null
    float* g_points;       // [n_points][3] or null
    float* ga;             // 5 x TBL64 out
    float* gp;             // 4 x TBL64 out (written only when g_grad)
    float* r0;             // TBL64 out
};

constexpr int SDFB_WAVES = 8;   // 2 waves per SIMD: this kernel is dominated by stash loads (64 % wave-cycles in s_waitcnt at 1 wave/SIMD)

template <bool HAS_GG>
__global__ __launch_bounds__(64 * SDFB_WAVES) void sdf_bwd_kernel(SdfBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_sdf_weights(lds, a.w, threadIdx.x, 64 * SDFB_WAVES);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int ntiles = (a.n_points + TP - 1) / TP;
    const size_t tbl = (size_t)ntiles * 1024;

    const float* w0 = lds + SdfLds::W0 + p * SdfLds::LD0 + g;
    const float* w1h = lds + SdfLds::W1 + p * SdfLds::LD1 + 4 * g;
    const float* w1e = lds + SdfLds::W1 + p * SdfLds::LD1 + 64 + g;
    const float* w2h = lds + SdfLds::W2 + p * SdfLds::LD1 + 4 * g;
    const float* w2e = lds + SdfLds::W2 + p * SdfLds::LD1 + 64 + g;
    const float* w3 = lds + SdfLds::W3 + p * SdfLds::LD3 + 4 * g;
    const float* w4 = lds + SdfLds::W4 + p * SdfLds::LD3 + 4 * g;
    const float* w5s = lds + SdfLds::W5 + 4 * g;
    const float* w5ft = lds + SdfLds::W5 + (1 + 4 * g) * SdfLds::LD3 + p;
    const float* w4t = lds + SdfLds::W4 + 4 * g * SdfLds::LD3 + p;
    const float* w3t = lds + SdfLds::W3 + 4 * g * SdfLds::LD3 + p;
    const float* w2t = lds + SdfLds::W2 + 4 * g * SdfLds::LD1 + p;
    const float* w1t = lds + SdfLds::W1 + 4 * g * SdfLds::LD1 + p;

// gx_c += scale_c * sum_s V[s] * (W_le * DV[4c..4c+3])[s]
#define SC_PE_DOT(WE, LD, V, DV, SCALE)                                                     \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                      \
            f32x4 tacc[NT];                                                                  \
            acc_zero(tacc);                                                                  \
            if (c == 0) mm_pe<LD, NT, 0, 4>(WE, DV + 0, tacc);                               \
            if (c == 1) mm_pe<LD, NT, 4, 4>(WE, DV + 4, tacc);                               \
            if (c == 2) mm_pe<LD, NT, 8, 4>(WE, DV + 8, tacc);                               \
            float dsum = 0.f;                                                                \
            _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s)                             \
                dsum = __builtin_fmaf(V[s], tacc[s >> 2][s & 3], dsum);                      \
            gx[c] = __builtin_fmaf(SCALE, dsum, gx[c]);                                      \
        }

    // ================= R sweep (only when d/d(grad) is given), then V sweep, per chunk of CH tiles =================
    // R parks pend_0..3 in Ga_0..3, pend_4 in Ga_4, u4 in r0 and the second-order part of G point in g_points; the V
    // sweep of the same chunk picks them up a few microseconds later (same wave, still in L2 / Infinity Cache, so the
    // round trip and the second read of a_l cost no HBM traffic).  Two loops instead of one fused body keep each body
    // under 256 VGPRs (2 waves per SIMD) -- a single fused body needed ~500 and spilled.
    // One tile per iteration: R sweep, then V sweep of the same tile.  What the V sweep needs first (a_4, pend_4, u_4) and
    // the running point gradient stay in registers across the junction; pend_0..3 are parked in Ga_0..3.
    const int wave_gid = blockIdx.x * SDFB_WAVES + wave, nwaves = gridDim.x * SDFB_WAVES;
#pragma unroll 1
    for (int tile = wave_gid; tile < ntiles; tile += nwaves) {
    float gx[3] = {0.f, 0.f, 0.f};
    float j_av[ACT_STEPS], j_pend[ACT_STEPS], j_u[ACT_STEPS];      // junction registers (HAS_GG only)
    float d1[PE_STEPS];                                            // dE/dx of the tile, shared by both sweeps
    if (HAS_GG) {
        {
            const int pt = tile * TP + p;
            const bool valid = pt < a.n_points;
            const int ptc = valid ? pt : a.n_points - 1;
            const float x0 = a.points[(size_t)ptc * 3 + 0], x1 = a.points[(size_t)ptc * 3 + 1], x2 = a.points[(size_t)ptc * 3 + 2];
            float e[PE_STEPS], d2[PE_STEPS];
            pe_slots<true, true, true>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
            float gam[3] = {0.f, 0.f, 0.f};
            if (valid) { gam[0] = a.g_grad[(size_t)pt * 3]; gam[1] = a.g_grad[(size_t)pt * 3 + 1]; gam[2] = a.g_grad[(size_t)pt * 3 + 2]; }
            f32x4 acc[NT];
            float avA[ACT_STEPS], pvA[ACT_STEPS], avB[ACT_STEPS], pvB[ACT_STEPS], gq[ACT_STEPS], gpv[ACT_STEPS];
            float eps[PE_STEPS];
#pragma unroll
            for (int j = 0; j < PE_STEPS; ++j) eps[j] = gam[j >> 2] * d1[j];
// the stash loads of layer L are issued one layer ahead (before the element-wise step of layer L-1 and the MFMAs of
// layer L; two register buffers A/B) and consumed after the MFMAs of layer L
#define SC_R_LOAD(L, av, pv)                                                                \
            tbl_load(a.stash_a + (size_t)(L) * tbl, tile, p, g, av);                         \
            tbl_load(a.stash_p + (size_t)(L) * tbl, tile, p, g, pv);                         \
            __builtin_amdgcn_sched_barrier(0);
#define SC_R_STEP(L, WE, LD, HAS_PE, av, pv)                                                \
            acc_to_regs(acc, gq);                                                            \
            {                                                                                \
                float pn[ACT_STEPS];                                                         \
                _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                       \
                    float t, r;                                                              \
                    stash_parts(av[s], t, r);                                             \
                    const float ds = stash_d1(av[s], t, r);                               \
                    gpv[s] = gq[s] * ds;                                                     \
                    pn[s] = gq[s] * pv[s] * stash_d2(t, r);                               \
                    pv[s] = pv[s] * ds;                      /* q_l, for the second-order point term */ \
                }                                                                            \
                tbl_store(a.gp + (size_t)(L) * tbl, tile, p, g, gpv);                        \
                tbl_store(a.ga + (size_t)(L) * tbl, tile, p, g, pn);                         \
            }                                                                                \
            if (HAS_PE && a.g_points) { SC_PE_DOT(WE, LD, pv, d2, gam[c]) }
            acc_zero(acc);
            SC_R_LOAD(0, avA, pvA)
            SC_R_LOAD(1, avB, pvB)
            mm_pe<SdfLds::LD0, NT, 0, PE_STEPS>(w0, eps, acc);                 // Gq0
            SC_R_STEP(0, w0, SdfLds::LD0, true, avA, pvA)
            acc_zero(acc);
            SC_R_LOAD(2, avA, pvA)
            mm_act<SdfLds::LD1, NT>(w1h, gpv, acc);
            mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w1e, eps, acc);                // Gq1
            SC_R_STEP(1, w1e, SdfLds::LD1, true, avB, pvB)
            acc_zero(acc);
            SC_R_LOAD(3, avB, pvB)
            mm_act<SdfLds::LD1, NT>(w2h, gpv, acc);
            mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w2e, eps, acc);                // Gq2
            SC_R_STEP(2, w2e, SdfLds::LD1, true, avA, pvA)
            acc_zero(acc);
            tbl_load(a.stash_a + 4 * tbl, tile, p, g, j_av);
            __builtin_amdgcn_sched_barrier(0);
            mm_act<SdfLds::LD3, NT>(w3, gpv, acc);                             // Gq3
            SC_R_STEP(3, w3, SdfLds::LD3, false, avB, pvB)
#undef SC_R_STEP
#undef SC_R_LOAD
            acc_zero(acc);
            mm_act<SdfLds::LD3, NT>(w4, gpv, acc);                             // Gq4
            acc_to_regs(acc, gq);
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) {
                float t, r;
                stash_parts(j_av[s], t, r);
                j_pend[s] = gq[s] * w5s[kp(s)] * stash_d2(t, r);
                j_u[s] = gq[s] * stash_d1(j_av[s], t, r);
            }
            // gx now holds the second-order part of G point (Gx_c = Gg_c * sum_l q_l . (W_le d2E/dx_c^2)); the V sweep adds the rest
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ================= V pass =================
    {
        const int pt = tile * TP + p;
        const bool valid = pt < a.n_points;
        const int ptc = valid ? pt : a.n_points - 1;
        if (!HAS_GG) {
            const float x0 = a.points[(size_t)ptc * 3 + 0], x1 = a.points[(size_t)ptc * 3 + 1], x2 = a.points[(size_t)ptc * 3 + 2];
            float e[PE_STEPS], d2[PE_STEPS];
            pe_slots<true, false, true>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
        }
        const float Gs = (valid && a.g_sdf) ? a.g_sdf[pt] : 0.f;
        f32x4 acc[NT];
        float av[ACT_STEPS], pv[ACT_STEPS], avB[ACT_STEPS], pvB[ACT_STEPS], gav[ACT_STEPS];
// a_L and the parked pend_L are requested one layer ahead of their use (two register buffers)
#define SC_V_LOAD(L, av, pv)                                                                \
        tbl_load(a.stash_a + (size_t)(L) * tbl, tile, p, g, av);                             \
        if (HAS_GG) tbl_load(a.ga + (size_t)(L) * tbl, tile, p, g, pv);                      \
        __builtin_amdgcn_sched_barrier(0);
        acc_zero(acc);
        if (a.g_feat) {
            float gf[ACT_STEPS];
            tbl_load(a.g_feat, tile, p, g, gf);
            mm_act_t<SdfLds::LD3, NT>(w5ft, gf, acc);
        }
        if (!HAS_GG) tbl_load(a.stash_a + 4 * tbl, tile, p, g, av);
        {
            float r0v[ACT_STEPS];
#pragma unroll
            for (int s = 0; s < ACT_STEPS; ++s) {
                const float a4 = HAS_GG ? j_av[s] : av[s];      /* (av is free again after this block) */
                float t, r;
                stash_parts(a4, t, r);
                const float gh = acc[s >> 2][s & 3] + w5s[kp(s)] * Gs;
                r0v[s] = Gs * stash_val(a4, t) + (HAS_GG ? j_u[s] : 0.f);
                gav[s] = gh * stash_d1(a4, t, r) + (HAS_GG ? j_pend[s] : 0.f);
            }
            tbl_store(a.r0, tile, p, g, r0v);
            tbl_store(a.ga + 4 * tbl, tile, p, g, gav);
        }
#define SC_V_STEP(L, WT, LD, av, pv)                                                        \
        acc_zero(acc);                                                                       \
        mm_act_t<LD, NT>(WT, gav, acc);                                                      \
        _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                               \
            float t, r;                                                                      \
            stash_parts(av[s], t, r);                                                     \
            gav[s] = acc[s >> 2][s & 3] * stash_d1(av[s], t, r) + (HAS_GG ? pv[s] : 0.f); \
        }                                                                                    \
        tbl_store(a.ga + (size_t)(L) * tbl, tile, p, g, gav);
        SC_V_LOAD(3, av, pv)
        SC_V_LOAD(2, avB, pvB)
        SC_V_STEP(3, w4t, SdfLds::LD3, av, pv)
        SC_V_LOAD(1, av, pv)
        SC_V_STEP(2, w3t, SdfLds::LD3, avB, pvB)
        SC_V_LOAD(0, avB, pvB)
        SC_PE_DOT(w2e, SdfLds::LD1, gav, d1, 1.f)
        SC_V_STEP(1, w2t, SdfLds::LD1, av, pv)
        SC_PE_DOT(w1e, SdfLds::LD1, gav, d1, 1.f)
        SC_V_STEP(0, w1t, SdfLds::LD1, avB, pvB)
        SC_PE_DOT(w0, SdfLds::LD0, gav, d1, 1.f)
#undef SC_V_STEP
#undef SC_V_LOAD
#undef SC_PE_DOT
        if (a.g_points) {
            const float o0 = group_sum(gx[0]), o1 = group_sum(gx[1]), o2 = group_sum(gx[2]);
            if (valid && g == 0) {
                a.g_points[(size_t)pt * 3 + 0] = o0;
                a.g_points[(size_t)pt * 3 + 1] = o1;
                a.g_points[(size_t)pt * 3 + 2] = o2;
            }
        }
    }
    }   // tile
}

}  // namespace sc

extern "C" int sc_sdf_backward(const float* points, const float* w_pack, int n_points, int symmetric,
                               const float* stash_a, const float* stash_p, const float* g_sdf,
                               const float* g_grad, const float* g_feat, float* g_points, float* ga,
                               float* gp, float* r0, void* stream_) {
    if (n_points <= 0) return 0;
    sc::SdfBwdArgs a{points, w_pack, n_points, symmetric, stash_a, stash_p, g_sdf, g_grad, g_feat, g_points, ga, gp, r0};
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + sc::SDFB_WAVES - 1) / sc::SDFB_WAVES;
    if (blocks > 256) blocks = 256;
    const size_t lds_bytes = sc::SdfLds::TOTAL * sizeof(float);
    hipStream_t stream = (hipStream_t)stream_;
    if (g_grad) {
        (void)hipFuncSetAttribute((const void*)sc::sdf_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   // per launch: the attribute is per device, no process-wide state
        hipLaunchKernelGGL(sc::sdf_bwd_kernel<true>, dim3(blocks), dim3(64 * sc::SDFB_WAVES), lds_bytes, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)sc::sdf_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   // per launch: the attribute is per device, no process-wide state
        hipLaunchKernelGGL(sc::sdf_bwd_kernel<false>, dim3(blocks), dim3(64 * sc::SDFB_WAVES), lds_bytes, stream, a);
    }
    return (int)hipGetLastError();
}
