// sdf_fwd.hip -- conditional SDF MLP forward on gfx950: value, feature and d(sdf)/dx in one pass.
//
// Replaces SDFNetwork.forward / get_conditional_output (model/implicit.py:138-189) including the
// autograd.grad(create_graph) call at :180-186 and the Renderer's d(density)/dx at
// model/renderer.py:94-107 (density' is a per-point scalar applied by the caller).
//
// Per point (latent already folded into per-image biases c_l by the host, skip-scaling 1/sqrt2
// pre-applied to W1/W2, PE columns in slot order):
//   a0 = W0 e + c0;  h0 = sp(a0)
//   a1 = W1 [h0; e] + c1; h1 = sp(a1);   a2 = W2 [h1; e] + c2; h2 = sp(a2)
//   a3 = W3 h2 + c3; h3;  a4 = W4 h3 + c4; h4;   out = W5 h4 + b5  -> sdf = out[0], feat = out[1:]
// d(sdf)/dx by the reverse ("adjoint") sweep, all in registers:
//   q4 = W5[0,:] * sp'(a4);  p_l = W_{l+1,h}^T q_{l+1};  q_l = p_l * sp'(a_l)
//   g_c = sum_l q_l . (W_{l,e} dE/dx_c)          (the PE Jacobian is applied in forward mode:
//                                                 13 non-zero slots per coordinate)
// Work: 464 (value) + 400 (gradient) v_mfma_f32_16x16x4 per 16 points.
#include "mlp_tile.hpp"

namespace sc {

struct SdfFwdArgs {
    const float* points;   // [n_points][3]
    const float* w;        // SdfPack image
    const float* cbias;    // [n_images][5][64]
    int n_points;
    int n_per_image;       // points are image-major; image = point / n_per_image
    int n_images;
    int symmetric;
    float* sdf;            // [n_points] or null
    float* grad;           // [n_points][3] or null (required when GRAD)
    float* feat;           // TBL64 [ntiles] or null
    float* stash_a;        // [5] x TBL64 (layer-major) or null: pre-activations a_l  (training)
    float* stash_p;        // [4] x TBL64 or null: adjoint p_0..p_3               (training)
    float* scratch;        // GRAD without stash_a: [gridDim.x * WAVES][5][1024] floats of L2-resident per-wave scratch
};

constexpr int SDF_WAVES = 8;   // 2 waves per SIMD: the VALU phases of one wave overlap the MFMA phases of the other

// STASH: both training stashes AND the feature output are given (a compile-time fact for the training instance: every `if (a.stash_p)` around a store was a
// branch, and at its join the compiler's wait-count pass has to assume the stores in flight and waits for them before the NEXT load can be
// used -- on gfx9 loads and stores retire through one in-order counter).  STASH = false keeps the run-time checks (evaluation, odd callers).
template <bool GRAD, bool STASH = false>
__global__ __launch_bounds__(64 * SDF_WAVES) void sdf_fwd_kernel(SdfFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_sdf_weights(lds, a.w, threadIdx.x, 64 * SDF_WAVES);
    __syncthreads();

    // wave index as a SCALAR: the per-wave scratch base below depends on it, and with a vector-register wave index hipcc wrapped each
    // of the 40 park stores / loads of a tile in a waterfall loop (4 v_readfirstlane + 2 v_cmp + exec save / restore + branch).
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const float* w5f = lds + SdfLds::W5 + (1 + p) * SdfLds::LD3 + 4 * g;   // feature rows 1..64
    const float* w5s = lds + SdfLds::W5 + 4 * g;                            // sdf row 0 (broadcast reads)
    // transposed views: rows indexed by K (out channel), columns by M (in channel)
    const float* w4t = lds + SdfLds::W4 + 4 * g * SdfLds::LD3 + p;
    const float* w3t = lds + SdfLds::W3 + 4 * g * SdfLds::LD3 + p;
    const float* w2t = lds + SdfLds::W2 + 4 * g * SdfLds::LD1 + p;
    const float* w1t = lds + SdfLds::W1 + 4 * g * SdfLds::LD1 + p;
    const float* b5 = lds + SdfLds::B5;

    // pre-activations a_l are parked in memory between the value chain and the gradient sweep (keeping
    // sp'(a_l) of five layers in registers costs 80 VGPRs and forced 1 wave/SIMD): the training stash if
    // there is one, otherwise a per-wave scratch slot that never leaves L2.
    float* park = a.stash_a;
    size_t park_stride = tbl;
    if (!STASH && GRAD && !park) { park = a.scratch + (size_t)(blockIdx.x * SDF_WAVES + wave) * 5 * 1024; park_stride = 1024; }
    for (int tile = blockIdx.x * SDF_WAVES + wave; tile < ntiles; tile += gridDim.x * SDF_WAVES) {
        const int ptile = (STASH || a.stash_a) ? tile : 0;
        const int pt = tile * TP + p;
        const bool valid = pt < a.n_points;
        const int ptc = valid ? pt : a.n_points - 1;
        const float x0 = a.points[(size_t)ptc * 3 + 0], x1 = a.points[(size_t)ptc * 3 + 1], x2 = a.points[(size_t)ptc * 3 + 2];
        const int img = min(ptc / a.n_per_image, a.n_images - 1);
        const float* cb = a.cbias + (size_t)img * 320 + 4 * g;

        float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
        pe_slots<GRAD, false>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);

        float h[ACT_STEPS];
        f32x4 acc[NT];

#define SC_ACTIVATE(L)                                                                    \
        {                                                                                  \
            float av[ACT_STEPS];                                                           \
            acc_to_regs(acc, av);                                                          \
            if (!SC_STASH_H && (GRAD || park)) tbl_store(park + (size_t)(L) * park_stride, ptile, p, g, av); \
            _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                         \
                float t, r;                                                                \
                softplus_parts(av[s], t, r);                                               \
                h[s] = softplus_val(av[s], t);                                             \
            }                                                                              \
            /* the ACTIVATION is parked (mlp_tile.hpp).  GRAD: park is never null (the launch refuses it) -- saying so removes a branch whose  \
               join made the compiler's wait-count pass wait for the four stores before the next layer's bias loads could be used */          \
            if (SC_STASH_H && (GRAD || park)) tbl_store(park + (size_t)(L) * park_stride, ptile, p, g, h); \
        }

        // ---- value chain ----
        // The per-image biases of layer l + 1 are requested BEFORE the park stores of layer l: on gfx9 loads and stores share one in-order
        // counter, so a bias load issued behind the four stores waits for their acknowledgement as well (read from the ISA: `s_waitcnt
        // vmcnt(3)` right behind [4 stores, 4 loads], five times per tile).
        f32x4 bnext[NT];
#define SC_NEXT_BIAS(L) acc_init(bnext, cb + (L) * 64); __builtin_amdgcn_sched_barrier(0);
#define SC_TAKE_BIAS_(L) SC_TAKE_BIAS()
#define SC_TAKE_BIAS() _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = bnext[t];
        acc_init(acc, cb + 0 * 64);
        mm_pe<SdfLds::LD0, NT, 0, PE_STEPS>(w0, e, acc);
        SC_NEXT_BIAS(1)
        SC_ACTIVATE(0)
        SC_TAKE_BIAS_(1)
        mm_act<SdfLds::LD1, NT>(w1h, h, acc);
        mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w1e, e, acc);
        SC_NEXT_BIAS(2)
        SC_ACTIVATE(1)
        SC_TAKE_BIAS_(2)
        mm_act<SdfLds::LD1, NT>(w2h, h, acc);
        mm_pe<SdfLds::LD1, NT, 0, PE_STEPS>(w2e, e, acc);
        SC_NEXT_BIAS(3)
        SC_ACTIVATE(2)
        SC_TAKE_BIAS_(3)
        mm_act<SdfLds::LD3, NT>(w3, h, acc);
        SC_NEXT_BIAS(4)
        SC_ACTIVATE(3)
        SC_TAKE_BIAS_(4)
        mm_act<SdfLds::LD3, NT>(w4, h, acc);
        SC_ACTIVATE(4)
#undef SC_ACTIVATE
#undef SC_NEXT_BIAS
#undef SC_TAKE_BIAS
#undef SC_TAKE_BIAS_

        // ---- output layer: sdf by VALU dot (row 0), feature rows by MFMA ----
        float sp = 0.f;
#pragma unroll
        for (int s = 0; s < ACT_STEPS; ++s) sp = __builtin_fmaf(w5s[kp(s)], h[s], sp);
        const float sdf = group_sum(sp) + b5[0];
        if (a.sdf && valid && g == 0) a.sdf[pt] = sdf;
        if (STASH || a.feat) {
            acc_init(acc, b5 + 1 + 4 * g);
            mm_act<SdfLds::LD3, NT>(w5f, h, acc);
            float fv[ACT_STEPS];
            acc_to_regs(acc, fv);
            tbl_store(a.feat, tile, p, g, fv);
        }

        if (GRAD) {
            // ---- adjoint sweep ----
            float q[ACT_STEPS], pv[ACT_STEPS];
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#define SC_PE_JAC(WE, LD)                                                                  \
            {                                                                              \
                f32x4 t0[NT], t1[NT], t2[NT];                                              \
                acc_zero(t0); acc_zero(t1); acc_zero(t2);                                  \
                mm_pe<LD, NT, 0, 4>(WE, d1 + 0, t0);                                       \
                mm_pe<LD, NT, 4, 4>(WE, d1 + 4, t1);                                       \
                mm_pe<LD, NT, 8, 4>(WE, d1 + 8, t2);                                       \
                _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                     \
                    g0 = __builtin_fmaf(q[s], t0[s >> 2][s & 3], g0);                      \
                    g1 = __builtin_fmaf(q[s], t1[s >> 2][s & 3], g1);                      \
                    g2 = __builtin_fmaf(q[s], t2[s >> 2][s & 3], g2);                      \
                }                                                                          \
            }
// The parked pre-activation of layer L is requested before the MFMAs that precede its use (it does not depend on
// them), so the reload latency hides behind the matrix pipe.
#define SC_DSP_LOAD(L)                                                                      \
            tbl_load(park + (size_t)(L) * park_stride, ptile, p, g, av);                   \
            __builtin_amdgcn_sched_barrier(0);
#define SC_DSP(L, EXPR)                                                                     \
            {                                                                              \
                _Pragma("unroll") for (int s = 0; s < ACT_STEPS; ++s) {                     \
                    float t, r;                                                            \
                    stash_parts(av[s], t, r);                                              \
                    const float ds = stash_d1(av[s], t, r);                                \
                    q[s] = (EXPR) * ds;                                                    \
                }                                                                          \
            }
            float av[ACT_STEPS];
            SC_DSP_LOAD(4)
            SC_DSP(4, w5s[kp(s)])
            acc_zero(acc);
            SC_DSP_LOAD(3)
            mm_act_t<SdfLds::LD3, NT>(w4t, q, acc);                 // p3 = W4^T q4
            acc_to_regs(acc, pv);
            if (STASH || a.stash_p) tbl_store(a.stash_p + 3 * tbl, tile, p, g, pv);
            SC_DSP(3, pv[s])
            acc_zero(acc);
            SC_DSP_LOAD(2)
            mm_act_t<SdfLds::LD3, NT>(w3t, q, acc);                 // p2 = W3^T q3
            acc_to_regs(acc, pv);
            if (STASH || a.stash_p) tbl_store(a.stash_p + 2 * tbl, tile, p, g, pv);
            SC_DSP(2, pv[s])
            SC_DSP_LOAD(1)
            SC_PE_JAC(w2e, SdfLds::LD1)
            acc_zero(acc);
            mm_act_t<SdfLds::LD1, NT>(w2t, q, acc);                 // p1 = W2h^T q2
            acc_to_regs(acc, pv);
            if (STASH || a.stash_p) tbl_store(a.stash_p + 1 * tbl, tile, p, g, pv);
            SC_DSP(1, pv[s])
            SC_DSP_LOAD(0)
            SC_PE_JAC(w1e, SdfLds::LD1)
            acc_zero(acc);
            mm_act_t<SdfLds::LD1, NT>(w1t, q, acc);                 // p0 = W1h^T q1
            acc_to_regs(acc, pv);
            if (STASH || a.stash_p) tbl_store(a.stash_p + 0 * tbl, tile, p, g, pv);
            SC_DSP(0, pv[s])
#undef SC_DSP_LOAD
#undef SC_DSP
            SC_PE_JAC(w0, SdfLds::LD0)
#undef SC_PE_JAC
            g0 = group_sum(g0); g1 = group_sum(g1); g2 = group_sum(g2);
            if (a.grad && valid && g == 0) {
                a.grad[(size_t)pt * 3 + 0] = g0;
                a.grad[(size_t)pt * 3 + 1] = g1;
                a.grad[(size_t)pt * 3 + 2] = g2;
            }
        }
    }
}

}  // namespace sc

extern "C" int sc_sdf_forward(const float* points, const float* w_pack, const float* cbias, int n_points,
                              int n_per_image, int n_images, int symmetric, float* sdf, float* grad,
                              float* feat, float* stash_a, float* stash_p, float* scratch, void* stream_) {
    if (n_points <= 0) return 0;
    sc::SdfFwdArgs a{points, w_pack, cbias, n_points, n_per_image, n_images, symmetric, sdf, grad, feat, stash_a, stash_p, scratch};
    const int ntiles = (n_points + sc::TP - 1) / sc::TP;
    int blocks = (ntiles + sc::SDF_WAVES - 1) / sc::SDF_WAVES;
    if (blocks > 256) blocks = 256;   // one persistent 8-wave workgroup per CU (LDS-resident weights)
    if (grad && !stash_a && !scratch) return (int)hipErrorInvalidValue;
    const size_t lds_bytes = sc::SdfLds::TOTAL * sizeof(float);
    hipStream_t stream = (hipStream_t)stream_;
    if (grad && stash_a && stash_p && feat) {      // the training render: both stashes and the feature output (what STASH stands for)
        (void)hipFuncSetAttribute((const void*)sc::sdf_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((sc::sdf_fwd_kernel<true, true>), dim3(blocks), dim3(64 * sc::SDF_WAVES), lds_bytes, stream, a);
    } else if (grad) {
        (void)hipFuncSetAttribute((const void*)sc::sdf_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   // per launch: the attribute is per device, no process-wide state
        hipLaunchKernelGGL(sc::sdf_fwd_kernel<true>, dim3(blocks), dim3(64 * sc::SDF_WAVES), lds_bytes, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)sc::sdf_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   // per launch: the attribute is per device, no process-wide state
        hipLaunchKernelGGL(sc::sdf_fwd_kernel<false>, dim3(blocks), dim3(64 * sc::SDF_WAVES), lds_bytes, stream, a);
    }
    return (int)hipGetLastError();
}
