// loss.hip -- the per-step loss reductions of one render in ONE launch (values + input gradients).
//
// Replaces, for the [B,R,*] tensors a render produces (reference model/graph.py:220-265 -> model/loss.py):
//   render  : Loss.MSE_loss(rgb, rgb_target)                                   loss.py:19-32
//   mask    : Loss.mask_loss = iou_loss + reg.mask_mse * MSE_loss              loss.py:75-97
//   normal  : Loss.normal_loss (mask compaction, 5*L1 + angular, keep the int(n*(1-tol)) smallest
//             angular errors, mean)                                            loss.py:52-67
//   eikonal : Loss.MSE_loss(|grad sdf|, 1)                                     loss.py:19-32
// The reference spends ~60 launch-bound torch kernels (+ a full sort and two boolean-index syncs)
// on this per render; here blocks 0..B-1 do the per-image sums (IoU needs per-image numerators) and
// block B does the robust normal selection with a 4-pass radix select instead of a sort.
// Loss values: every image block publishes its three partial sums, the block that arrives last adds the B partials in image
// order (a fixed summation order: results do not depend on timing -- they used to be float atomicAdds into out[]); the gradients
// of each loss w.r.t. its prediction are written alongside (the backward is then 4 scalings).
// Bound: latency (tensors are tens of KB); the metric is microseconds and launch count.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

struct LossArgs {
    const float* rgb; const float* rgb_t;        // [B][R][3]
    const float* mask; const float* mask_t;      // [B][R]
    const float* normal; const float* normal_t;  // [B][R][3]
    const float* eik;                            // [B][E] or null
    int B, R, E;
    float normal_l1, mask_mse;
    double keep_frac;                            // 1 - reg.normal_tol
    float* out;                                  // [8]: render, mask, normal, eikonal | arrival counter (one word, ZERO at launch) | pad
    float* part;                                 // [B][4] workspace: per-image partial sums
    float* g_rgb; float* g_mask; float* g_normal; float* g_eik;
    float* g_normal_t;                           // d normal loss / d normal_t (the target is differentiable in the pose) or null
    float* ang_ws;                               // [B*R] workspace (angular error of masked rays)
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < nw; ++k) s += red[k];
    return s;
}

__device__ __forceinline__ uint32_t order_key(float f) {   // monotone float -> uint map
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void loss_fused_kernel(LossArgs a) {
    __shared__ float red[16];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sh_prefix, sh_remaining, sh_n;
    const int tid = threadIdx.x, nt = blockDim.x;
    if ((int)blockIdx.x < a.B) {
        // ---------------- per-image sums: MSE partials + IoU ----------------
        const int b = blockIdx.x;
        const float inv_rgb = 1.f / ((float)a.B * a.R * 3), inv_m = 1.f / ((float)a.B * a.R);
        float s_rgb = 0.f, s_mm = 0.f, s_i = 0.f, s_u = 0.f, s_e = 0.f;
        for (int i = tid; i < a.R * 3; i += nt) {
            const float d = a.rgb[(size_t)b * a.R * 3 + i] - a.rgb_t[(size_t)b * a.R * 3 + i];
            s_rgb += d * d;
            a.g_rgb[(size_t)b * a.R * 3 + i] = 2.f * d * inv_rgb;
        }
        for (int i = tid; i < a.R; i += nt) {
            const float p = a.mask[(size_t)b * a.R + i], t = a.mask_t[(size_t)b * a.R + i];
            s_mm += (p - t) * (p - t);
            s_i += p * t;
            s_u += p + t - p * t + 1.e-8f;
        }
        if (a.eik) {
            const float inv_e = 1.f / ((float)a.B * a.E);
            for (int i = tid; i < a.E; i += nt) {
                const float d = a.eik[(size_t)b * a.E + i] - 1.f;
                s_e += d * d;
                a.g_eik[(size_t)b * a.E + i] = 2.f * d * inv_e;
            }
            s_e = block_sum(s_e, red);
        }
        s_rgb = block_sum(s_rgb, red);
        s_mm = block_sum(s_mm, red);
        const float I = block_sum(s_i, red), U = block_sum(s_u, red);
        // d/dp [1 - I/U] = -(t*U - I*(1-t)) / U^2 ; mean over the B images
        for (int i = tid; i < a.R; i += nt) {
            const float p = a.mask[(size_t)b * a.R + i], t = a.mask_t[(size_t)b * a.R + i];
            a.g_mask[(size_t)b * a.R + i] = -(t * U - I * (1.f - t)) / (U * U) / (float)a.B
                                            + a.mask_mse * 2.f * (p - t) * inv_m;
        }
        if (tid == 0) {
            a.part[b * 4 + 0] = s_rgb * inv_rgb;
            a.part[b * 4 + 1] = (1.f - I / U) / (float)a.B + a.mask_mse * s_mm * inv_m;
            a.part[b * 4 + 2] = a.eik ? s_e / ((float)a.B * a.E) : 0.f;
            // publish (agent-scope release; the explicit wait keeps the counter from overtaking the write-back), take a ticket
            __threadfence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int ticket = atomicAdd(reinterpret_cast<unsigned int*>(a.out + 4), 1u);
            if (ticket == (unsigned int)a.B - 1u) {         // last image block: all partials are published
                __threadfence();                             // acquire
                float t0 = 0.f, t1 = 0.f, t3 = 0.f;
                for (int k = 0; k < a.B; ++k) {
                    t0 += __builtin_nontemporal_load(&a.part[k * 4 + 0]);
                    t1 += __builtin_nontemporal_load(&a.part[k * 4 + 1]);
                    t3 += __builtin_nontemporal_load(&a.part[k * 4 + 2]);
                }
                a.out[0] = t0;
                a.out[1] = t1;
                if (a.eik) a.out[3] = t3;
            }
        }
        return;
    }
    // ---------------- block B: robust masked normal loss over the whole batch ----------------
    // (round 5) The first LOSS_VR elements of a thread (i = tid + j nt: all of them at the training size, 16,384 rays on 1,024 threads) stay in
    // registers through the four selection passes and the final pass -- they used to be re-read from the workspace in each -- and the
    // 256-bin scan of a pass is a wave-wide prefix sum instead of one thread walking the bins (100 -> ~35 us per launch, two launches per step).
    const int N = a.B * a.R;
    constexpr int LOSS_VR = 16;
    float vreg[LOSS_VR];
    unsigned int n_local = 0;
    auto angle = [&](int i) {
        const bool m = a.mask_t[i] > 0.5f && a.mask[i] > 0.5f;
        float ang = __builtin_nanf("");                 // NaN marks "not in the mask"
        if (m) {
            ang = 1.f - (a.normal[i * 3] * a.normal_t[i * 3] + a.normal[i * 3 + 1] * a.normal_t[i * 3 + 1]
                         + a.normal[i * 3 + 2] * a.normal_t[i * 3 + 2]);
            ++n_local;
        }
        return ang;
    };
#pragma unroll
    for (int j = 0; j < LOSS_VR; ++j) {
        const int i = tid + j * nt;
        vreg[j] = i < N ? angle(i) : __builtin_nanf("");
    }
    for (int i = tid + LOSS_VR * nt; i < N; i += nt) a.ang_ws[i] = angle(i);
    if (tid == 0) sh_n = 0;
    __syncthreads();
    atomicAdd(&sh_n, n_local);
    __syncthreads();
    const unsigned int n = sh_n;
    const unsigned int n_keep = (unsigned int)((double)n * a.keep_frac);
    // radix select: key of the n_keep-th smallest angular error (1-based rank n_keep)
    uint32_t prefix = 0, kth_key = 0;
    unsigned int remaining = n_keep;      // rank (1-based) inside the current candidate set
    if (n_keep > 0) {
        for (int pass = 3; pass >= 0; --pass) {
            for (int k = tid; k < 256; k += nt) hist[k] = 0;
            __syncthreads();
            const uint32_t hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
            auto count = [&](float v) {
                if (v == v) {
                    const uint32_t key = order_key(v);
                    if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
                }
            };
#pragma unroll
            for (int j = 0; j < LOSS_VR; ++j) count(vreg[j]);
            for (int i = tid + LOSS_VR * nt; i < N; i += nt) count(a.ang_ws[i]);
            __syncthreads();
            if (tid < 64) {       // first bin d with (bins before d) + hist[d] >= remaining: lane = 4 bins, wave-wide inclusive prefix sum
                const unsigned int r = remaining;
                unsigned int h[4], sum = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { h[q] = hist[4 * tid + q]; sum += h[q]; }
                unsigned int incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned int t = __shfl_up(incl, d);
                    if (tid >= d) incl += t;
                }
                unsigned int acc = incl - sum;
                if (acc < r && r <= incl) {             // exactly one lane (the total is >= remaining)
                    int d = 4 * tid;
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (d == 4 * tid + q && acc + h[q] < r) { acc += h[q]; ++d; }
                    sh_prefix = prefix | ((uint32_t)d << (8 * pass));
                    sh_remaining = r - acc;
                }
            }
            __syncthreads();
            prefix = sh_prefix;
            remaining = sh_remaining;
            __syncthreads();
        }
        kth_key = prefix;       // `remaining` = how many elements equal to the k-th value are kept (by index order)
    }
    // keep: key < kth, plus the first `remaining` ties in index order (ties are exact float duplicates)
    __shared__ unsigned int tie_taken;
    if (tid == 0) tie_taken = 0;
    __syncthreads();
    float s_loss = 0.f;
    const float inv_keep = n_keep > 0 ? 1.f / (float)n_keep : 0.f;
    // ties: process in index order with a serialised counter only when needed (rare)
#pragma unroll 1
    for (int base = 0, jj = 0; base < N; base += nt, ++jj) {
        const int i = base + tid;
        bool keep = false, tie = false;
        float v = 0.f;
        if (i < N) {
            v = 0.f;
            if (jj < LOSS_VR) {
#pragma unroll
                for (int j = 0; j < LOSS_VR; ++j)
                    if (j == jj) v = vreg[j];
            } else {
                v = a.ang_ws[i];
            }
            if (v == v && n_keep > 0) {
                const uint32_t key = order_key(v);
                keep = key < kth_key;
                tie = key == kth_key;
            }
        }
        if (__syncthreads_or(tie)) {
            // rank ties inside this chunk by thread index (chunks are visited in index order)
            for (int w = 0; w < nt; w += 64) {
                if ((tid & ~63) == w) {
                    const unsigned long long bal = __ballot(tie);
                    if (tie) {
                        const unsigned int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                        if (tie_taken + before < remaining) keep = true;
                    }
                    if ((tid & 63) == 0 && bal) tie_taken += __popcll(bal);
                }
                __syncthreads();
            }
        }
        if (i < N) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f;
            if (keep) {
                const float p0 = a.normal[i * 3], p1 = a.normal[i * 3 + 1], p2 = a.normal[i * 3 + 2];
                const float t0 = a.normal_t[i * 3], t1 = a.normal_t[i * 3 + 1], t2 = a.normal_t[i * 3 + 2];
                const float l1 = fabsf(p0 - t0) + fabsf(p1 - t1) + fabsf(p2 - t2);
                s_loss += a.normal_l1 * l1 + v;
                auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
                g0 = (a.normal_l1 * sgn(p0 - t0) - t0) * inv_keep;
                g1 = (a.normal_l1 * sgn(p1 - t1) - t1) * inv_keep;
                g2 = (a.normal_l1 * sgn(p2 - t2) - t2) * inv_keep;
                // the target normal is transform_normal(input normal, predicted pose) (graph.py:85,260): it carries a
                // gradient into the view estimator: d/dt [l1 |p - t| + 1 - p.t] = -l1 sgn(p - t) - p
                h0 = (-a.normal_l1 * sgn(p0 - t0) - p0) * inv_keep;
                h1 = (-a.normal_l1 * sgn(p1 - t1) - p1) * inv_keep;
                h2 = (-a.normal_l1 * sgn(p2 - t2) - p2) * inv_keep;
            }
            a.g_normal[i * 3] = g0; a.g_normal[i * 3 + 1] = g1; a.g_normal[i * 3 + 2] = g2;
            if (a.g_normal_t) { a.g_normal_t[i * 3] = h0; a.g_normal_t[i * 3 + 1] = h1; a.g_normal_t[i * 3 + 2] = h2; }
        }
    }
    s_loss = block_sum(s_loss, red);
    if (tid == 0) a.out[2] = n_keep > 0 ? s_loss * inv_keep : __builtin_nanf("");   // mean of an empty set is NaN (torch)
}

}  // namespace sc

extern "C" int sc_loss_fused_forward(const float* rgb, const float* rgb_t, const float* mask, const float* mask_t,
                                     const float* normal, const float* normal_t, const float* eik, int B, int R, int E,
                                     float normal_l1, float mask_mse, double keep_frac, float* out4, float* g_rgb,
                                     float* g_mask, float* g_normal, float* g_eik, float* g_normal_t, float* ang_ws, void* stream_) {
    if (B <= 0 || R <= 0) return 0;
    // out4: 8 floats, zero-filled by the caller ([4] is the arrival counter); ang_ws: B*R + 4*B floats of workspace
    sc::LossArgs a{rgb, rgb_t, mask, mask_t, normal, normal_t, eik, B, R, E, normal_l1, mask_mse, keep_frac,
                   out4, ang_ws + (size_t)B * R, g_rgb, g_mask, g_normal, g_eik, g_normal_t, ang_ws};
    hipLaunchKernelGGL(sc::loss_fused_kernel, dim3(B + 1), dim3(1024), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
