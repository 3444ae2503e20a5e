// rgb_bwd.hip -- reverse pass of rgb_fwd.hip: compositing, density, per-sample normals, RGB MLP.
//
// Hand-derived counterpart of what autograd does for Renderer.forward's tail
// (model/renderer.py:110-152 and volume_rendering :187-209) plus RGBNetwork (model/implicit.py:220-239)
// and LaplaceDensity (:65-83, incl. d/d beta).  One wavefront per ray.
//
// Phase 1 (lane == sample): recompute delta, E, alpha, T, w from z / sdf and the per-sample colours
//   the forward stored; turn the per-ray upstream gradients into per-sample gradients
//   (Gc, G sigma, G n_hat, Gz) with one 64-lane scan + one suffix scan; then
//   G sdf, G(d sdf/dx), G beta, Gz, G depth_fac by the closed-form density / normal derivatives.
// Phase 2 (four 16-point MFMA tiles): recompute the RGB chain, run it in reverse
//   (Gy2 -> V2^T -> Gy1 -> V1^T -> Gy0 -> V0f^T -> G feature, PE Jacobian -> G point) and leave
//   Gy_l / r_l in HBM (TBL64) for the weight-gradient GEMMs (wgrad.hip).
#include "rgb_common.hpp"
#include "mlp_xch.hpp"
#include "mlp_presplit.hpp"

namespace sc {

struct RgbBwdArgs {
    const float* points; const float* z_vals; const float* depth_fac; const float* sdf; const float* grad;
    const float* feat; const float* v; const float* dbias; const float* beta_param; const float* rgb_flat;
    int n_rays, rays_per_image, n_images, symmetric;
    float beta_min, bgcolor, normal_pow;
    const float* G_rgb;     // [n_rays][3] or null
    const float* G_mask;    // [n_rays] or null
    const float* G_depth;   // [n_rays] or null
    const float* G_normal;  // [n_rays][3] or null
    float* g_sdf;        // [P]
    float* g_grad;       // [P][3]
    float* g_feat;       // TBL64
    float* g_points;     // [P][3]   (RGB net's own dependence on the point; sdf_bwd adds the rest)
    float* g_z;          // [n_rays][64]
    float* g_depth_fac;  // [n_rays]
    float* g_beta;       // [SC_RGB_BWD_BETA_PARTS = 2048]: one partial per wave of the grid (gradient wrt the raw parameter), fully written;
                         // the gradient is their sum in index order (sc_partial_reduce) -- it used to be one float atomicAdd per wave
    float* gy;           // 3 x TBL64: pre-activation gradients Gy0, Gy1, Gy2
    float* rr;           // 3 x TBL64: post-ReLU activations r0, r1, r2
    float* gy3;          // [P][3]: gradient at the pre-sigmoid output
    float* partial;      // FUSED kernel only: [gridDim.x][partial_stride]: per workgroup the gradient of V0 | V1 | V2 (RgbPack::V3 floats) followed
                         // by the per-image bias gradients [n_images][3][64]; fully written; summed in index order by sc_partial_reduce
    int partial_stride;
    float* v3_part;      // null, or [SC_RGB_BWD_BETA_PARTS][196]: per wave of the grid the sums over its points of gy3_j * r2[ch] (= dV3 [3][64]),
                         // of gy3_j (= db3 [3]) and one zero -- fully written; their sum in index order (sc_partial_reduce) is the gradient of
                         // the output layer.  With it rr[2] and gy3 are NOT written (nobody else reads them): 280 MB less per launch.
};

// FUSED (round 5): the weight gradients of V0, V1, V2 and the per-image bias gradients are formed INSIDE this kernel by
// four more waves (the scheme of sdf_bwdw.hip: the chain waves drop each layer's operand pair -- Gy_l and the layer's input -- into a
// 4 KiB LDS slot pair, two workgroup barriers bracket the write, wave 4 + w accumulates rows 16 w .. 16 w + 15 of every matrix with fp32
// MFMAs whose K index is the point) instead of writing Gy_0..2 and r_0..1 to HBM (1.6 GB per bs32 render) for three sc_wgrad launches.
// SC_RGBB_DB = 1 (experiment of round 5, correct at every parity bar, NOT faster: 5.52 against 5.54 ms per bs32 training render on one
// box, profiles/r05_rgb_stash_ab.txt): TWO sets of slots, used alternately by consecutive hand-overs, and ONE barrier per hand-over
// ("written") instead of two: a chain wave may write set n & 1 as soon as it has passed the barrier of hand-over n - 1, because the
// weight-gradient waves reach that barrier only after they have consumed hand-over n - 2 -- the last reader of that set.  The point stash
// alternates by tile for the same reason (written at the first hand-over of a tile, read at its third).  LDS: 63 + 64 + 4 KB.  The
// product keeps one set and two barriers (63 + 32 + 2 KB): the barriers are not what the chain waves wait for.
#ifndef SC_RGBB_DB
#define SC_RGBB_DB 0
#endif
constexpr int RB_NBUF = SC_RGBB_DB ? 2 : 1;
constexpr int RB_XCH = (RgbLds::TOTAL + 3) & ~3;          // exchange slots [set][chain wave][A | B][1024]
constexpr int RB_XSET = 4 * 2 * 1024;
constexpr int RB_PTS = RB_XCH + RB_NBUF * RB_XSET;        // point stash [set][chain wave][16 points][8]: x0 x1 x2 - | - - valid -
constexpr int RB_PSET = 4 * 16 * 8;
constexpr int RB_LDS_FLOATS = RB_PTS + RB_NBUF * RB_PSET;
// SPLIT (round 6; FUSED + STASH only): the three transposed products of the reverse chain (V2^T, V1^T, V0f^T) and the encoding's Jacobian
// (V0e) in the exact bf16x3 split arithmetic from PRE-SPLIT fragments (mlp_presplit.hpp) -- the parked activations make the forward
// matrices unnecessary here, so the fp32 weight image (63 KiB) gives way to [V2^T | V1^T | V0f^T | V0e] = 90 KiB of fragments + V3 / b3 in
// fp32; the exchange slots and the point stash follow.  144 K = 32 MFMAs (2.3 k matrix cycles) instead of 192 fp32 ones (6.1 k) per tile
// on the chain waves, whose SIMD's matrix pipe is shared with a weight-gradient wave.
constexpr int RBS_V2T = 0, RBS_V1T = ps::HID_BYTES, RBS_V0FT = 2 * ps::HID_BYTES, RBS_V0E = 3 * ps::HID_BYTES;       // bytes
constexpr int RBS_V3 = (3 * ps::HID_BYTES + ps::PE_BYTES) / 4;                                                       // floats: V3 [3][64], b3 [3]
constexpr int RBS_XCH = (RBS_V3 + 196 + 3) & ~3;
constexpr int RBS_LDS_FLOATS = RBS_XCH + RB_NBUF * RB_XSET + RB_NBUF * RB_PSET;
static_assert(RBS_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget of the split reverse kernel");

// Phase profile (tuning tool, tools/prof_rgb_bwd.py; compiled only with -DSC_RGBB_PROFILE): s_memtime stamps of ONE iteration of one
// workgroup, chain wave 0 and weight-gradient wave 4, into rgbb_prof_buf [8 waves][64].
#ifdef SC_RGBB_PROFILE
__device__ unsigned long long* rgbb_prof_buf = nullptr;
__device__ int rgbb_prof_block = -1, rgbb_prof_iter = -1;
#define RB_STAMP(ID) if (prof_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) rgbb_prof_buf[wave * 64 + (ID)] = t_; }
#else
#define RB_STAMP(ID)
#endif

// STASH (round 5, second half; FUSED only): the forward pass parked r0, r1, r2 (sc_rgb_composite_forward_stash) and they are LOADED here
// instead of recomputed -- the phase profile (tools/prof_rgb_bwd.py) charges the recomputed forward chain 48 k of the chain wave's 155 k
// cycles per ray; the colours come from rgb_flat, which phase 1 loads anyway.
template <bool FUSED, bool STASH = false, bool SPLIT = false>
__global__ __launch_bounds__(FUSED ? 512 : 256, FUSED ? 1 : 2) void rgb_composite_bwd_kernel(RgbBwdArgs a) {
    static_assert(!SPLIT || (FUSED && STASH), "the split reverse chain exists for the fused form that reads the parked activations");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int XCH = SPLIT ? RBS_XCH : RB_XCH, PTS = XCH + RB_NBUF * RB_XSET, LDS_FLOATS = PTS + RB_NBUF * RB_PSET;
    [[maybe_unused]] char* const ldsc = reinterpret_cast<char*>(lds);
    if (SPLIT) {
        ps::stage_hidden_t(ldsc + RBS_V2T, a.v + RgbPack::V2, 64, 0, threadIdx.x, 512);
        ps::stage_hidden_t(ldsc + RBS_V1T, a.v + RgbPack::V1, 64, 0, threadIdx.x, 512);
        ps::stage_hidden_t(ldsc + RBS_V0FT, a.v + RgbPack::V0, 112, 48, threadIdx.x, 512);
        ps::stage_pe(ldsc + RBS_V0E, a.v + RgbPack::V0, 112, 0, threadIdx.x, 512);
        if (threadIdx.x < 196) lds[RBS_V3 + threadIdx.x] = threadIdx.x < 195 ? a.v[RgbPack::V3 + threadIdx.x] : 0.f;
    } else {
        stage_rgb_weights(lds, a.v, threadIdx.x, FUSED ? 512 : 256);
    }
    if (FUSED) {
        for (int e = threadIdx.x; e < LDS_FLOATS - XCH; e += 512) lds[XCH + e] = 0.f;
        float* cbz = a.partial + (size_t)blockIdx.x * a.partial_stride + RgbPack::V3;
        for (int e = threadIdx.x; e < a.n_images * 192; e += 512) cbz[e] = 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_iter = (a.n_rays + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);       // ray of (iteration, wave): (it * grid + block) * 4 + wave
    if (FUSED && wave >= 4) {
        // ================= weight-gradient role: wave 4 + w owns rows 16 w .. 16 w + 15 of dV0 [64][48 | 64], dV1, dV2 [64][64] =================
        const int w = wave - 4, i = lane & 15, kg = lane >> 4;
        const int rd = (kg * 64 + (i ^ kg)) << 2;
        const bool symmetric = a.symmetric != 0;
        f32x4 d0e[3], d0f[4], d1[4], d2[4];
        acc_zero(d0e); acc_zero(d0f); acc_zero(d1); acc_zero(d2);
        float* cbp = a.partial + (size_t)blockIdx.x * a.partial_stride + RgbPack::V3;
        float rs[3] = {0.f, 0.f, 0.f};
        int cur_img = -1;                                    // >= 0: rs[] belongs to this image; -2: the four rays of the iteration straddle images
        auto flush = [&]() {
            if (cur_img >= 0) {
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    float v = rs[l];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (kg == 0) cbp[((size_t)cur_img * 3 + l) * 64 + 16 * w + i] += v;      // one writer per address: plain read-modify-write
                    rs[l] = 0.f;
                }
            }
        };
#define RB_CONSUME(ACC, PEACC, WITH_PE, RS)                                                 \
        lds_barrier();                                                                       \
        if (!SC_RGBB_DB) lds_barrier();                                                      \
        RB_STAMP(1 + 2 * (k * 3 + (2 - RS)))                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                      \
            const int set_ = SC_RGBB_DB ? ((k + (2 - RS)) & 1) : 0;                          \
            const float* sA = lds + XCH + set_ * RB_XSET + (c * 2 + 0) * 1024;            \
            const float* sB = lds + XCH + set_ * RB_XSET + (c * 2 + 1) * 1024;            \
            const float4 af = xch_frag(sA, rd, w);                                           \
            float4 bf[4];                                                                    \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) bf[n] = xch_frag(sB, rd, n);       \
            outer16<4>(af, bf, ACC);                                                         \
            if (WITH_PE) {                                                                   \
                float4 pf[3];                                                                \
                pe_frags<1>(lds + PTS + (SC_RGBB_DB ? (k & 1) : 0) * RB_PSET + c * 16 * 8, i, kg, symmetric, pf); \
                outer16<3>(af, pf, PEACC);                                                   \
            }                                                                                \
            const float v = (af.x + af.y) + (af.z + af.w);                                   \
            if (cur_img >= 0) rs[RS] += v;                                                   \
            else {                                                                           \
                float t = v;                                                                 \
                t += __shfl_xor(t, 16);                                                      \
                t += __shfl_xor(t, 32);                                                      \
                const int ray_c = (it * (int)gridDim.x + (int)blockIdx.x) * 4 + c;           \
                if (kg == 0 && ray_c < a.n_rays)                                             \
                    cbp[((size_t)min(ray_c / a.rays_per_image, a.n_images - 1) * 3 + RS) * 64 + 16 * w + i] += t; \
            }                                                                                \
        }                                                                                    \
        RB_STAMP(2 + 2 * (k * 3 + (2 - RS)))
#pragma unroll 1
        for (int it = 0; it < n_iter; ++it) {
#ifdef SC_RGBB_PROFILE
            const bool prof_on = (int)blockIdx.x == rgbb_prof_block && it == rgbb_prof_iter && rgbb_prof_buf;
#endif
            RB_STAMP(0)
            const int ray0 = (it * (int)gridDim.x + (int)blockIdx.x) * 4, ray3 = min(ray0 + 3, a.n_rays - 1);
            const int img0 = min(min(ray0, a.n_rays - 1) / a.rays_per_image, a.n_images - 1), img3 = min(ray3 / a.rays_per_image, a.n_images - 1);
            if (img0 != cur_img || img3 != img0) {
                flush();
                cur_img = img0 == img3 ? img0 : -2;
            }
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                RB_CONSUME(d2, d0e, false, 2)               // Gy2 x r1
                RB_CONSUME(d1, d0e, false, 1)               // Gy1 x r0
                RB_CONSUME(d0f, d0e, true, 0)               // Gy0 x (feature | E)
            }
        }
#undef RB_CONSUME
        flush();
        float* out = a.partial + (size_t)blockIdx.x * a.partial_stride;
        auto put = [&](const f32x4* acc, int ntile, int off, int ld, int col0) {
            for (int n = 0; n < ntile; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[off + (16 * w + 4 * kg + r) * ld + col0 + 16 * n + i] = acc[n][r];
        };
        put(d0e, 3, RgbPack::V0, 112, 0);
        put(d0f, 4, RgbPack::V0, 112, 48);
        put(d1, 4, RgbPack::V1, 64, 0);
        put(d2, 4, RgbPack::V2, 64, 0);
        return;
    }
    const int p = lane & 15, g = lane >> 4;
    RgbLanePtrs L(lds, p, g);
    const float* const v3p = SPLIT ? lds + RBS_V3 + 4 * g : L.v3;            // V3 [3][64] (+ 4 g), fp32 in both layouts
    [[maybe_unused]] float* const slot0 = lds + XCH + (wave * 2 + 0) * 1024;       // set 0; set 1 is RB_XSET floats further
    [[maybe_unused]] float* const pts0 = lds + PTS + wave * 16 * 8;
    [[maybe_unused]] int wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = (((p >> 2) * 64 + 4 * g + (r ^ (p >> 2))) << 2) + (p & 3);
// hand operand pair J (0, 1, 2) of tile k to the weight-gradient waves.  One set of slots: B1 (they have finished the previous pair),
// write, B2 (visible).  Two sets: write into set (k + J) & 1, one barrier (see SC_RGBB_DB above).
#define RB_SLOTS(J)                                                                          \
        [[maybe_unused]] float* slotA = slot0 + (SC_RGBB_DB ? ((k + (J)) & 1) : 0) * RB_XSET; \
        [[maybe_unused]] float* slotB = slotA + 1024;                                        \
        [[maybe_unused]] float* ptsw = pts0 + (SC_RGBB_DB ? (k & 1) : 0) * RB_PSET;
#ifdef SC_RGBB_PROFILE
#define RB_EXCHANGE(J, WRITES) if (FUSED) { RB_SLOTS(J) if (!SC_RGBB_DB) lds_barrier(); RB_STAMP(sid++) WRITES RB_STAMP(sid++) lds_barrier(); RB_STAMP(sid++) }
#else
#define RB_EXCHANGE(J, WRITES) if (FUSED) { RB_SLOTS(J) if (!SC_RGBB_DB) lds_barrier(); WRITES lds_barrier(); }
#endif
    const float bp = a.beta_param[0];
    const float beta = fabsf(bp) + a.beta_min;
    const float dbeta_dbp = bp > 0.f ? 1.f : (bp < 0.f ? -1.f : 0.f);
    const size_t tbl = (size_t)a.n_rays * 4 * 1024;
    float gbeta_acc = 0.f;
    float v3acc[3][ACT_STEPS], b3acc[3] = {0.f, 0.f, 0.f};     // this lane's share of dV3 (channels kp(s) + 4 g of its point column) and db3
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s2 = 0; s2 < ACT_STEPS; ++s2) v3acc[j][s2] = 0.f;

#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
#ifdef SC_RGBB_PROFILE
        const bool prof_on = (int)blockIdx.x == rgbb_prof_block && it == rgbb_prof_iter && rgbb_prof_buf;
        int sid = 2;
#endif
        RB_STAMP(0)
        const int ray = (it * (int)gridDim.x + (int)blockIdx.x) * 4 + wave;
        if (ray >= a.n_rays) {
            if (!FUSED) break;
            for (int st = 0; st < 12; ++st) {               // tail: keep the barrier count of an iteration (4 tiles x 3 steps), feed zeros
                if (!SC_RGBB_DB) lds_barrier();
                if (st < RB_NBUF) {                         // every set is cleared at the hand-over that would write it (st = 3 k + J: set st & 1)
                    xch_zero(slot0 + st * RB_XSET, lane); xch_zero(slot0 + st * RB_XSET + 1024, lane);
                }
                if (st == 0 || (SC_RGBB_DB && st == 3)) {   // ... and so is every point stash (tile k = st / 3: set k & 1)
                    if (lane < 32) reinterpret_cast<float4*>(pts0 + (st ? RB_PSET : 0))[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                lds_barrier();
            }
            continue;
        }
        const int img = min(ray / a.rays_per_image, a.n_images - 1);
        const float* db = a.dbias + (size_t)img * 192 + 4 * g;
        const size_t sp = (size_t)ray * 64 + lane;   // this lane's sample
        // ---------------- phase 1: compositing backward, lane == sample ----------------
        const float z = a.z_vals[sp];
        const float s = a.sdf[sp];
        const float gx = a.grad[sp * 3 + 0], gy_ = a.grad[sp * 3 + 1], gz = a.grad[sp * 3 + 2];
        const float c0 = a.rgb_flat[sp * 3 + 0], c1 = a.rgb_flat[sp * 3 + 1], c2 = a.rgb_flat[sp * 3 + 2];
        const float dfac = a.depth_fac[ray];
        const float ex = expf(-fabsf(s) / beta);
        const float psi = s >= 0.f ? 0.5f * ex : 1.f - 0.5f * ex;
        const float sigma = psi / beta;
        const float kk = (0.5f / (beta * beta)) * ex;
        const float vx = kk * gx, vy = kk * gy_, vz = kk * gz;
        const float vnorm = sqrtf(vx * vx + vy * vy + vz * vz);
        const float vden = fmaxf(vnorm, 1e-12f);
        const float n0 = vx / vden, n1 = vy / vden, n2 = vz / vden;
        const float znext = __shfl_down(z, 1);
        const float delta = lane == 63 ? 0.f : znext - z;
        const float E = delta * sigma;
        const float emE = expf(-E);
        const float alpha = 1.f - emE;
        const float T = expf(-(wave_inclusive_scan(E) - E));
        const float w = alpha * T;
        const float pw = a.normal_pow;
        const float wn = pw == 1.f ? w : powf(w, pw);
        const float N0 = wave_sum(wn * n0), N1 = wave_sum(wn * n1), N2 = wave_sum(wn * n2);

        const float Gr0 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 0] : 0.f;
        const float Gr1 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 1] : 0.f;
        const float Gr2 = a.G_rgb ? a.G_rgb[(size_t)ray * 3 + 2] : 0.f;
        const float Gm = a.G_mask ? a.G_mask[ray] : 0.f;
        const float Gd = a.G_depth ? a.G_depth[ray] : 0.f;
        float GN0 = 0.f, GN1 = 0.f, GN2 = 0.f;
        if (a.G_normal) {
            const float q0 = a.G_normal[(size_t)ray * 3 + 0], q1 = a.G_normal[(size_t)ray * 3 + 1], q2 = a.G_normal[(size_t)ray * 3 + 2];
            const float nn = sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
            if (nn > 1e-12f) {   // F.normalize backward: (G - (G.n) n) / |N|
                const float o0 = N0 / nn, o1 = N1 / nn, o2 = N2 / nn;
                const float dt = q0 * o0 + q1 * o1 + q2 * o2;
                GN0 = (q0 - dt * o0) / nn; GN1 = (q1 - dt * o1) / nn; GN2 = (q2 - dt * o2) / nn;
            } else {             // clamped branch: output = N / eps
                GN0 = q0 / 1e-12f; GN1 = q1 / 1e-12f; GN2 = q2 / 1e-12f;
            }
        }
        const float Gc0 = w * Gr0, Gc1 = w * Gr1, Gc2 = w * Gr2;
        const float gn_dot = GN0 * n0 + GN1 * n1 + GN2 * n2;
        float Gw = (Gr0 * c0 + Gr1 * c1 + Gr2 * c2) - a.bgcolor * (Gr0 + Gr1 + Gr2) + Gm + Gd * z * dfac;
        Gw += pw == 1.f ? gn_dot : (w > 0.f ? pw * powf(w, pw - 1.f) * gn_dot : 0.f);
        const float Gnh0 = wn * GN0, Gnh1 = wn * GN1, Gnh2 = wn * GN2;
        const float Galpha = Gw * T;
        const float GC = -(Gw * alpha) * T;                      // dL/dC_i, C_i = sum_{j<i} E_j
        const float GE = wave_exclusive_suffix(GC) + Galpha * emE;
        const float Gsigma = GE * delta;
        const float Gdelta = GE * sigma;
        const float Gdelta_prev = __shfl_up(Gdelta, 1);
        float Gz = Gd * w * dfac - (lane == 63 ? 0.f : Gdelta) + (lane == 0 ? 0.f : Gdelta_prev);
        const float Gdfac = wave_sum(Gd * w * z);
        // n_hat = v / max(|v|, eps), v = kk * g
        float Gv0, Gv1, Gv2;
        if (vnorm > 1e-12f) {
            const float dt = Gnh0 * n0 + Gnh1 * n1 + Gnh2 * n2;
            Gv0 = (Gnh0 - dt * n0) / vnorm; Gv1 = (Gnh1 - dt * n1) / vnorm; Gv2 = (Gnh2 - dt * n2) / vnorm;
        } else {
            Gv0 = Gnh0 / 1e-12f; Gv1 = Gnh1 / 1e-12f; Gv2 = Gnh2 / 1e-12f;
        }
        const float Gkk = Gv0 * gx + Gv1 * gy_ + Gv2 * gz;
        const float sgn = s >= 0.f ? 1.f : -1.f;
        const float Gs = -Gsigma * kk - Gkk * sgn * kk / beta;
        const float Gbeta = Gsigma * (-sigma / beta + 0.5f * ex * s / (beta * beta * beta))
                          + Gkk * kk * (-2.f / beta + fabsf(s) / (beta * beta));
        gbeta_acc += Gbeta;
        a.g_sdf[sp] = Gs;
        a.g_grad[sp * 3 + 0] = kk * Gv0;
        a.g_grad[sp * 3 + 1] = kk * Gv1;
        a.g_grad[sp * 3 + 2] = kk * Gv2;
        a.g_z[sp] = Gz;
        if (lane == 0) a.g_depth_fac[ray] = Gdfac;

        // ---------------- phase 2: RGB MLP reverse, four 16-point tiles ----------------
        RB_STAMP(1)
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
#ifdef SC_RGBB_PROFILE
            sid = 2 + 15 * k;
#endif
            const int tile = ray * 4 + k;
            const size_t pt = (size_t)tile * TP + p;
            const int src = 16 * k + p;
            const float gc0 = __shfl(Gc0, src), gc1 = __shfl(Gc1, src), gc2 = __shfl(Gc2, src);
            const float x0 = a.points[pt * 3 + 0], x1 = a.points[pt * 3 + 1], x2 = a.points[pt * 3 + 2];
            float f[ACT_STEPS];
            float r[3][ACT_STEPS];
            float col[3];
            if (STASH) {                                     // requested behind the point and in the order of their first use (the counter retires
                tbl_load(a.rr + 2 * tbl, tile, p, g, r[2]);  // in order): r2 (output layer, mask of Gy2), r1 (hand-over 1), r0 (hand-over 2), the
                tbl_load(a.rr + 1 * tbl, tile, p, g, r[1]);  // feature (hand-over 3); the positional encoding is evaluated while they travel
                tbl_load(a.rr + 0 * tbl, tile, p, g, r[0]);
            }
            tbl_load(a.feat, tile, p, g, f);
            __builtin_amdgcn_sched_barrier(0);
            float e[PE_STEPS], d1[PE_STEPS], d2[PE_STEPS];
            pe_slots<true, false, true>(x0, x1, x2, g, a.symmetric != 0, e, d1, d2);
            RB_STAMP(sid++)                                  /* +0: inputs requested, PE evaluated */
            if (STASH) {
                col[0] = __shfl(c0, src); col[1] = __shfl(c1, src); col[2] = __shfl(c2, src);
            } else {
                rgb_chain(L, db, e, f, r, col);
            }
            RB_STAMP(sid++)                                  /* +1: forward chain recomputed */
            if (!FUSED) {
                tbl_store(a.rr + 0 * tbl, tile, p, g, r[0]);
                tbl_store(a.rr + 1 * tbl, tile, p, g, r[1]);
            }
            const float y0 = gc0 * col[0] * (1.f - col[0]);
            const float y1 = gc1 * col[1] * (1.f - col[1]);
            const float y2 = gc2 * col[2] * (1.f - col[2]);
            if (FUSED || a.v3_part) {
#pragma unroll
                for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                    v3acc[0][s2] = __builtin_fmaf(y0, r[2][s2], v3acc[0][s2]);
                    v3acc[1][s2] = __builtin_fmaf(y1, r[2][s2], v3acc[1][s2]);
                    v3acc[2][s2] = __builtin_fmaf(y2, r[2][s2], v3acc[2][s2]);
                }
                b3acc[0] += y0; b3acc[1] += y1; b3acc[2] += y2;       // (every lane group of a point holds the same y: group 0 is taken below)
            } else {
                tbl_store(a.rr + 2 * tbl, tile, p, g, r[2]);
                if (g == 0) { a.gy3[pt * 3 + 0] = y0; a.gy3[pt * 3 + 1] = y1; a.gy3[pt * 3 + 2] = y2; }
            }
            float gyv[ACT_STEPS];
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                const float gr = v3p[kp(s2)] * y0 + v3p[64 + kp(s2)] * y1 + v3p[128 + kp(s2)] * y2;
                gyv[s2] = r[2][s2] > 0.f ? gr : 0.f;
            }
            if (!FUSED) tbl_store(a.gy + 2 * tbl, tile, p, g, gyv);
            RB_STAMP(sid++)                                  /* +2: output layer, dV3 sums, Gy2; then exchange 1: +3 B1 passed, +4 written, +5 B2 passed */
            RB_EXCHANGE(0, xch_write(slotA, wr, gyv, 1.f); xch_write(slotB, wr, r[1], 1.f);
                        if (g == 0) {
                            *reinterpret_cast<float4*>(ptsw + p * 8) = make_float4(x0, x1, x2, 0.f);
                            *reinterpret_cast<float4*>(ptsw + p * 8 + 4) = make_float4(0.f, 0.f, 1.f, 0.f);
                        })
            f32x4 acc1[1][NT];
            f32x4 (&acc)[NT] = acc1[0];
            [[maybe_unused]] MlpPieces<8> gp[1][2];
            acc_zero(acc);
            if constexpr (SPLIT) { ps::split_act(gyv, gp[0]); ps::hidden_part<1>(ldsc + RBS_V2T, lane, gp, acc1); }
            else mm_act_t<RgbLds::LD1, NT>(L.v2t, gyv, acc);
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) gyv[s2] = r[1][s2] > 0.f ? acc[s2 >> 2][s2 & 3] : 0.f;
            if (!FUSED) tbl_store(a.gy + 1 * tbl, tile, p, g, gyv);
            RB_STAMP(sid++)                                  /* +6: V2^T product + mask; exchange 2: +7, +8, +9 */
            RB_EXCHANGE(1, xch_write(slotA, wr, gyv, 1.f); xch_write(slotB, wr, r[0], 1.f);)
            acc_zero(acc);
            if constexpr (SPLIT) { ps::split_act(gyv, gp[0]); ps::hidden_part<1>(ldsc + RBS_V1T, lane, gp, acc1); }
            else mm_act_t<RgbLds::LD1, NT>(L.v1t, gyv, acc);
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) gyv[s2] = r[0][s2] > 0.f ? acc[s2 >> 2][s2 & 3] : 0.f;
            if (!FUSED) tbl_store(a.gy + 0 * tbl, tile, p, g, gyv);
            RB_STAMP(sid++)                                  /* +10: V1^T product + mask; exchange 3: +11, +12, +13 */
            RB_EXCHANGE(2, xch_write(slotA, wr, gyv, 1.f); xch_write(slotB, wr, f, 1.f);)
            acc_zero(acc);
            if constexpr (SPLIT) { ps::split_act(gyv, gp[0]); ps::hidden_part<1>(ldsc + RBS_V0FT, lane, gp, acc1); }
            else mm_act_t<RgbLds::LD0, NT>(L.v0ft, gyv, acc);
            float gf[ACT_STEPS];
            acc_to_regs(acc, gf);
            tbl_store(a.g_feat, tile, p, g, gf);
            RB_STAMP(sid++)                                  /* +14: V0f^T product, G feature stored; the tile ends at the next tile's +0 (or stamp 62) */
            float gxs[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 tacc[NT];
                acc_zero(tacc);
                if constexpr (SPLIT) ps::jac_part(ldsc + RBS_V0E, lane, c, d1 + 4 * c, tacc);
                else {
                    if (c == 0) mm_pe<RgbLds::LD0, NT, 0, 4>(L.v0e, d1 + 0, tacc);
                    if (c == 1) mm_pe<RgbLds::LD0, NT, 4, 4>(L.v0e, d1 + 4, tacc);
                    if (c == 2) mm_pe<RgbLds::LD0, NT, 8, 4>(L.v0e, d1 + 8, tacc);
                }
                float dsum = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < ACT_STEPS; ++s2) dsum = __builtin_fmaf(gyv[s2], tacc[s2 >> 2][s2 & 3], dsum);
                gxs[c] = group_sum(dsum);
            }
            if (g == 0) { a.g_points[pt * 3 + 0] = gxs[0]; a.g_points[pt * 3 + 1] = gxs[1]; a.g_points[pt * 3 + 2] = gxs[2]; }
        }
        RB_STAMP(62)
    }
    const float gb = wave_sum(gbeta_acc);
    if (lane == 0) a.g_beta[blockIdx.x * 4 + wave] = gb * dbeta_dbp;
    if (blockIdx.x == 0 && wave == 0)               // the waves of the blocks that were not launched (small renders)
        for (int e = gridDim.x * 4 + lane; e < 2048; e += 64) a.g_beta[e] = 0.f;
    if (FUSED || a.v3_part) {
        // sum over the 16 point columns of a lane group (xor shuffles stay inside the group of 16), then lane p == 0 of group g writes
        // its 16 channels of every row: a fixed order, per wave
        float* dst = a.v3_part + (size_t)(blockIdx.x * 4 + wave) * 196;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < ACT_STEPS; ++s2) {
                float v = v3acc[j][s2];
                for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                if (p == 0) dst[j * 64 + kp(s2) + 4 * g] = v;
            }
            float bsum = b3acc[j];
            for (int d = 8; d >= 1; d >>= 1) bsum += __shfl_xor(bsum, d);
            if (lane == 0) dst[192 + j] = bsum;
        }
        if (lane == 0) dst[195] = 0.f;
        if (blockIdx.x == 0 && wave == 0)
            for (int e = gridDim.x * 4 * 196 + lane; e < 2048 * 196; e += 64) a.v3_part[e] = 0.f;
    }
}

}  // namespace sc

extern "C" int sc_rgb_composite_backward_v3(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, float* v3_part, void* stream_) {
    if (n_rays <= 0) return 0;
    sc::RgbBwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat,
                     n_rays, rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow,
                     G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat, g_points, g_z, g_depth_fac, g_beta,
                     gy, rr, gy3, nullptr, 0, v3_part};
    int blocks = (n_rays + 3) / 4;
    if (blocks > 512) blocks = 512;
    const size_t lds_bytes = sc::RgbLds::TOTAL * sizeof(float);
    hipLaunchKernelGGL(sc::rgb_composite_bwd_kernel<false>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

// Workgroups (= partial images) of sc_rgb_composite_backward_fused for n_rays, and floats per partial image.
extern "C" int sc_rgb_composite_backward_fused_parts(int n_rays) {
    const int blocks = (n_rays + 3) / 4;
    return blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks);
}
extern "C" int sc_rgb_composite_backward_fused_partial_floats(int n_images) { return sc::RgbPack::V3 + n_images * 192; }

// The same reverse pass with the gradients of V0, V1, V2 and of the per-image biases formed in the kernel (no gy / rr hand-off tensors):
// partial [parts][partial_floats(n_images)] receives one image per workgroup, fully written; v3_part as in the _v3 form (required).
extern "C" int sc_rgb_composite_backward_fused(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, void* stream_) {
    if (n_rays <= 0) return 0;
    if (!partial || !v3_part || n_images <= 0 || n_images > 256) return (int)hipErrorInvalidValue;
    sc::RgbBwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat,
                     n_rays, rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow,
                     G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat, g_points, g_z, g_depth_fac, g_beta,
                     nullptr, nullptr, nullptr, partial, sc_rgb_composite_backward_fused_partial_floats(n_images), v3_part};
    const int blocks = sc_rgb_composite_backward_fused_parts(n_rays);
    const size_t lds_bytes = (size_t)sc::RB_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)sc::rgb_composite_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(sc::rgb_composite_bwd_kernel<true>, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

// sc_rgb_composite_backward_fused with the hidden activations of the forward pass (rr: what sc_rgb_composite_forward_stash wrote) instead of
// their recomputation.  Everything else as above.
extern "C" int sc_rgb_composite_backward_fused_stash(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, const float* rr, void* stream_) {
    if (n_rays <= 0) return 0;
    if (!partial || !v3_part || !rr || n_images <= 0 || n_images > 256) return (int)hipErrorInvalidValue;
    sc::RgbBwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat,
                     n_rays, rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow,
                     G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat, g_points, g_z, g_depth_fac, g_beta,
                     nullptr, const_cast<float*>(rr), nullptr, partial, sc_rgb_composite_backward_fused_partial_floats(n_images), v3_part};
    const int blocks = sc_rgb_composite_backward_fused_parts(n_rays);
    const size_t lds_bytes = (size_t)sc::RB_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)sc::rgb_composite_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((sc::rgb_composite_bwd_kernel<true, true>), dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

// sc_rgb_composite_backward_fused_stash with the reverse chain's transposed products and the encoding's Jacobian in the exact bf16x3 split
// arithmetic from pre-split fragments (round 6; same operands and outputs, gradients equal up to fp32 rounding).
extern "C" int sc_rgb_composite_backward_fused_split(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, const float* rr, void* stream_) {
    if (n_rays <= 0) return 0;
    if (!partial || !v3_part || !rr || n_images <= 0 || n_images > 256) return (int)hipErrorInvalidValue;
    sc::RgbBwdArgs a{points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat,
                     n_rays, rays_per_image, n_images, symmetric, beta_min, bgcolor, normal_pow,
                     G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat, g_points, g_z, g_depth_fac, g_beta,
                     nullptr, const_cast<float*>(rr), nullptr, partial, sc_rgb_composite_backward_fused_partial_floats(n_images), v3_part};
    const int blocks = sc_rgb_composite_backward_fused_parts(n_rays);
    const size_t lds_bytes = (size_t)sc::RBS_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)sc::rgb_composite_bwd_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((sc::rgb_composite_bwd_kernel<true, true, true>), dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

extern "C" int sc_rgb_composite_backward(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, void* stream_) {
    return sc_rgb_composite_backward_v3(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param, rgb_flat, n_rays, rays_per_image,
                                        n_images, symmetric, beta_min, bgcolor, normal_pow, G_rgb, G_mask, G_depth, G_normal, g_sdf, g_grad, g_feat,
                                        g_points, g_z, g_depth_fac, g_beta, gy, rr, gy3, nullptr, stream_);
}

#ifdef SC_RGBB_PROFILE
// tuning tool only: where the stamps go (device buffer of 8 x 64 uint64) and which (workgroup, iteration) writes them
extern "C" int sc_rgbb_set_prof(unsigned long long* buf, int block, int iter) {
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(sc::rgbb_prof_buf), &buf, sizeof(buf));
    if (!rc) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(sc::rgbb_prof_block), &block, sizeof(block));
    if (!rc) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(sc::rgbb_prof_iter), &iter, sizeof(iter));
    return rc;
}
#endif
