/* oracle/chamfer_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Scalar C restatement of the reference's only native kernel,
 * /root/reference/external/chamfer3D/chamfer3D.cu.
 *
 *   sc_ref_chamfer_forward  <- NmDistanceKernel           (chamfer3D.cu:12-134, launches :142-143)
 *   sc_ref_chamfer_backward <- NmDistanceGradKernel       (chamfer3D.cu:155-174, launches :184-185)
 *
 * PINNED against the reference itself: oracle/build_chamfer_ref.py builds the reference's own extension (chamfer_cuda.cpp +
 * chamfer3D.cu, untouched, where they lie under /root/reference) for gfx950 into oracle/_ref/chamfer_3D_ref.so, and
 * tests/test_gpu_chamfer_ref.py runs it on the GPU beside this restatement and the HIP kernels: dist and idx agree bit for bit
 * (uniform clouds up to 100,000 x 100,000, lattice clouds with thousands of exact ties, ragged sizes), the gradients to 1e-6
 * (float atomics).  Also pinned: known-answer vectors in tests/golden/g9_chamfer.npz (float64 brute force within 1e-6,
 * engineered exact ties).  Arithmetic:
 *   - fp32 throughout; the source expression is  d = x2*x2 + y2*y2 + z2*z2  (chamfer3D.cu:35, x2 = target - query,
 *     :32-34), parsed as (x2*x2 + y2*y2) + z2*z2 and contracted by the compiler.  The reference build for this GPU computes
 *         d = fmaf(y2, y2, x2*x2) + z2*z2
 *     (x2*x2 rounded and fused into the y product, z2*z2 rounded, one rounded add: v_pk_mul / v_fma / v_add in its disassembly;
 *     of the eight candidate contractions only this one reproduces its output, tools/probe_chamfer_ref.py).  What nvcc emits
 *     for NVIDIA hardware may differ in the last bit of d (<= 1 ulp; idx can then differ only between targets whose distances
 *     differ by < 2 ulp) -- on THIS platform the reference's results are the ones above.
 *   - strict '<' inside a 512-target tile, first element unconditionally (k==0), so the lowest
 *     index among equal minima wins inside a tile (chamfer3D.cu:36, :46 ...);
 *   - across tiles: overwrite only if previous best > tile best (chamfer3D.cu:126) -> again the
 *     lowest index wins globally.  The two rules together equal "global first minimum", which is
 *     what the loop below computes in one pass.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (explicit fmaf keeps the contraction exact).
 */
#include <math.h>
#include <stdint.h>

static void nm_distance(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int32_t *result_i) {
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz[(i * (int64_t)n + j) * 3 + 0];
            const float y1 = xyz[(i * (int64_t)n + j) * 3 + 1];
            const float z1 = xyz[(i * (int64_t)n + j) * 3 + 2];
            float best = 0.0f;
            int32_t best_i = 0;
            for (int k = 0; k < m; ++k) {
                const float x2 = xyz2[(i * (int64_t)m + k) * 3 + 0] - x1;
                const float y2 = xyz2[(i * (int64_t)m + k) * 3 + 1] - y1;
                const float z2 = xyz2[(i * (int64_t)m + k) * 3 + 2] - z1;
                const float xx = x2 * x2, zz = z2 * z2;
                const float d = fmaf(y2, y2, xx) + zz;
                if (k == 0 || d < best) { best = d; best_i = k; }
            }
            /* m == 0: the reference leaves the caller's zero-filled outputs untouched */
            if (m > 0) { result[i * (int64_t)n + j] = best; result_i[i * (int64_t)n + j] = best_i; }
        }
    }
}

int sc_ref_chamfer_forward(const float *xyz1, const float *xyz2, float *dist1, float *dist2,
                           int32_t *idx1, int32_t *idx2, int b, int n, int m) {
    nm_distance(b, n, xyz1, m, xyz2, dist1, idx1);
    nm_distance(b, m, xyz2, n, xyz1, dist2, idx2);
    return 1; /* reference returns 1 on success (chamfer3D.cu:151) */
}

static void nm_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                             const float *grad_dist1, const int32_t *idx1,
                             float *grad_xyz1, float *grad_xyz2) {
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz1[(i * (int64_t)n + j) * 3 + 0];
            const float y1 = xyz1[(i * (int64_t)n + j) * 3 + 1];
            const float z1 = xyz1[(i * (int64_t)n + j) * 3 + 2];
            const int j2 = idx1[i * (int64_t)n + j];
            const float x2 = xyz2[(i * (int64_t)m + j2) * 3 + 0];
            const float y2 = xyz2[(i * (int64_t)m + j2) * 3 + 1];
            const float z2 = xyz2[(i * (int64_t)m + j2) * 3 + 2];
            const float g = grad_dist1[i * (int64_t)n + j] * 2;
            grad_xyz1[(i * (int64_t)n + j) * 3 + 0] += g * (x1 - x2);
            grad_xyz1[(i * (int64_t)n + j) * 3 + 1] += g * (y1 - y2);
            grad_xyz1[(i * (int64_t)n + j) * 3 + 2] += g * (z1 - z2);
            grad_xyz2[(i * (int64_t)m + j2) * 3 + 0] += -(g * (x1 - x2));
            grad_xyz2[(i * (int64_t)m + j2) * 3 + 1] += -(g * (y1 - y2));
            grad_xyz2[(i * (int64_t)m + j2) * 3 + 2] += -(g * (z1 - z2));
        }
    }
}

/* Caller zero-fills gradxyz1/gradxyz2 (the reference's memset is commented out, chamfer3D.cu:177-178). */
int sc_ref_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                            const float *graddist1, const float *graddist2,
                            const int32_t *idx1, const int32_t *idx2, int b, int n, int m) {
    nm_distance_grad(b, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
    nm_distance_grad(b, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
    return 1;
}
