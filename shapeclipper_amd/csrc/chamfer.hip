// chamfer.hip -- Chamfer3D nearest-neighbour distance for gfx950 (MI355X).
//
// Replaces the reference's only native code, external/chamfer3D/chamfer3D.cu:
//   NmDistanceKernel      (chamfer3D.cu:12-134)  -> chamfer_nn_kernel + chamfer_nn_index_kernel
//   NmDistanceGradKernel  (chamfer3D.cu:155-174) -> chamfer_grad_kernel
//
// Roofline: FP32 VALU bound (8 algorithmic FLOP per ordered pair: 3 sub, 2 mul, 1 fma, 1 add; no exact
// MFMA form exists because |a|^2+|b|^2-2ab changes rounding and therefore argmin ties).
// The inner loop works on TWO targets per instruction with the packed fp32 VALU forms (v_pk_add / v_pk_mul / v_pk_fma: 7 packed
// instructions + one v_min3 per two pairs = 4 vector instructions per pair; the scalar form of round 1 needed 7).
//
// Design (wave64, 256 CUs):
//   * a lane owns Q query points in registers; targets stream through LDS as float4 and are read
//     with broadcast ds_read_b128 (every lane reads the same address -> one bank access);
//   * the inner loop tracks only the running MIN over a 16-target sub-block (1 VALU op per pair
//     instead of compare + two selects); after each sub-block one strict '<' test records the
//     sub-block id.  A second tiny pass re-evaluates the 16 candidates with the *same* fma chain
//     and takes the first exact match.  "First sub-block whose min is strictly smaller" + "first
//     index inside it" == the reference's "lowest index among equal minima" rule
//     (chamfer3D.cu:36,46,126), bit for bit.
//   * d = fmaf(dy, dy, dx*dx) + dz*dz with dx = target - query: what the reference's own extension computes for
//     chamfer3D.cu:35 `x2*x2+y2*y2+z2*z2` when it is built for this GPU (oracle/build_chamfer_ref.py -> oracle/_ref: the
//     first product rounded, fused into the second, the third product rounded and added; identified instruction by
//     instruction in its disassembly and bit for bit on 100,000 x 100,000 points, tests/test_gpu_chamfer_ref.py), written
//     with explicit fma and contraction switched off so that host oracle, packed and scalar device code agree exactly.
#include "chamfer_common.hpp"
#include <stdlib.h>

namespace sc {

// One direction: for every query j of cloud `xyz` [b,n,3], min_k d(q_j, t_k) over `xyz2` [b,m,3].
// Writes the squared distance and the 16-target sub-block id that first reached it.
__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_kernel(
    int n, const float* __restrict__ xyz, int m, const float* __restrict__ xyz2,
    float* __restrict__ result, int* __restrict__ result_blk) {
    __shared__ float4 tgt[CH_TCHUNK];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int qbase = blockIdx.x * (CH_THREADS * CH_Q);
    const float* q_ptr = xyz + (size_t)b * n * 3;
    const float* t_ptr = xyz2 + (size_t)b * m * 3;

    float qx[CH_Q], qy[CH_Q], qz[CH_Q], best[CH_Q];
    int bblk[CH_Q];
#pragma unroll
    for (int q = 0; q < CH_Q; ++q) {
        int j = qbase + q * CH_THREADS + tid;
        j = j < n ? j : n - 1;
        qx[q] = q_ptr[j * 3 + 0];
        qy[q] = q_ptr[j * 3 + 1];
        qz[q] = q_ptr[j * 3 + 2];
        best[q] = __builtin_inff();
        bblk[q] = 0;
    }

    for (int k0 = 0; k0 < m; k0 += CH_TCHUNK) {
        const int cnt = min(CH_TCHUNK, m - k0);
        const int cnt_pad = (cnt + CH_SUB - 1) & ~(CH_SUB - 1);
        __syncthreads();
        stage_targets(tgt, t_ptr, k0, cnt, cnt_pad, tid);
        __syncthreads();
        for (int sb = 0; sb < cnt_pad; sb += CH_SUB) {
            float mn[CH_Q];
#pragma unroll
            for (int q = 0; q < CH_Q; ++q) mn[q] = __builtin_inff();
            CH_MIN_SUBBLOCK(mn)
            const int blk = (k0 + sb) / CH_SUB;
#pragma unroll
            for (int q = 0; q < CH_Q; ++q) {
                const bool better = mn[q] < best[q];
                best[q] = better ? mn[q] : best[q];
                bblk[q] = better ? blk : bblk[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CH_Q; ++q) {
        const int j = qbase + q * CH_THREADS + tid;
        if (j < n) {
            result[(size_t)b * n + j] = best[q];
            result_blk[(size_t)b * n + j] = bblk[q];
        }
    }
}

// Target-split variant for small batches (B*N/1024 workgroups cannot fill 256 CUs): blockIdx.z walks a slice
// of the targets; every (query, slice) resolves its exact first-minimum index inside the slice and publishes
// key = (float bits of d) << 32 | index with a 64-bit atomicMin.  d >= 0, so unsigned order of the float bits is
// numeric order, and for equal d the smaller index wins -- exactly the reference's global lowest-index tie rule.
__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_split_kernel(
    int n, const float* __restrict__ xyz, int m, const float* __restrict__ xyz2, int slice_len,
    unsigned long long* __restrict__ keys) {
    __shared__ float4 tgt[CH_TCHUNK];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int qbase = blockIdx.x * (CH_THREADS * CH_Q);
    const int t_begin = blockIdx.z * slice_len, t_end = min(m, t_begin + slice_len);
    const float* q_ptr = xyz + (size_t)b * n * 3;
    const float* t_ptr = xyz2 + (size_t)b * m * 3;
    float qx[CH_Q], qy[CH_Q], qz[CH_Q], best[CH_Q];
    int bblk[CH_Q];
#pragma unroll
    for (int q = 0; q < CH_Q; ++q) {
        int j = qbase + q * CH_THREADS + tid;
        j = j < n ? j : n - 1;
        qx[q] = q_ptr[j * 3 + 0]; qy[q] = q_ptr[j * 3 + 1]; qz[q] = q_ptr[j * 3 + 2];
        best[q] = __builtin_inff();
        bblk[q] = t_begin / CH_SUB;
    }
    for (int k0 = t_begin; k0 < t_end; k0 += CH_TCHUNK) {
        const int cnt = min(CH_TCHUNK, t_end - k0);
        const int cnt_pad = (cnt + CH_SUB - 1) & ~(CH_SUB - 1);
        __syncthreads();
        stage_targets(tgt, t_ptr, k0, cnt, cnt_pad, tid);
        __syncthreads();
        for (int sb = 0; sb < cnt_pad; sb += CH_SUB) {
            float mn[CH_Q];
#pragma unroll
            for (int q = 0; q < CH_Q; ++q) mn[q] = __builtin_inff();
            CH_MIN_SUBBLOCK(mn)
            const int blk = (k0 + sb) / CH_SUB;      // slices start at multiples of CH_SUB
#pragma unroll
            for (int q = 0; q < CH_Q; ++q) {
                const bool better = mn[q] < best[q];
                best[q] = better ? mn[q] : best[q];
                bblk[q] = better ? blk : bblk[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CH_Q; ++q) {
        const int j = qbase + q * CH_THREADS + tid;
        if (j < n && t_begin < t_end) {
            const int kb = bblk[q] * CH_SUB;
            int idx = kb;
            for (int t = CH_SUB - 1; t >= 0; --t) {
                const int k = kb + t;
                if (k < t_end && dist2(t_ptr[k * 3 + 0], t_ptr[k * 3 + 1], t_ptr[k * 3 + 2], qx[q], qy[q], qz[q]) == best[q]) idx = k;
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(best[q]) << 32) | (unsigned int)idx;
            atomicMin(&keys[(size_t)b * n + j], key);
        }
    }
}

__global__ __launch_bounds__(CH_THREADS) void chamfer_unpack_kernel(size_t total, const unsigned long long* __restrict__ keys,
                                                                   float* __restrict__ result, int* __restrict__ result_i) {
    const size_t gid = (size_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (gid >= total) return;
    const unsigned long long k = keys[gid];
    result[gid] = __uint_as_float((unsigned int)(k >> 32));
    result_i[gid] = (int)(unsigned int)(k & 0xFFFFFFFFull);
}

// Second pass: turn the winning sub-block id into the exact first index (same arithmetic).
__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_index_kernel(
    int b_total, int n, const float* __restrict__ xyz, int m, const float* __restrict__ xyz2,
    const float* __restrict__ result, int* __restrict__ result_i) {
    const size_t gid = (size_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (gid >= (size_t)b_total * n) return;
    const int b = (int)(gid / n);
    const float* q = xyz + gid * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float best = result[gid];
    const int k0 = result_i[gid] * CH_SUB;
    const float* t_ptr = xyz2 + (size_t)b * m * 3;
    int idx = k0;  // NaN inputs: nothing compares equal; keep the sub-block start (reference keeps 0)
    for (int t = CH_SUB - 1; t >= 0; --t) {
        const int k = k0 + t;
        if (k < m) {
            const float d = dist2(t_ptr[k * 3 + 0], t_ptr[k * 3 + 1], t_ptr[k * 3 + 2], qx, qy, qz);
            if (d == best) idx = k;
        }
    }
    result_i[gid] = idx;
}

// Backward, one direction (chamfer3D.cu:155-174): g = 2 * grad_dist; scatter-add +-g * (p1 - p2).
__global__ __launch_bounds__(CH_THREADS) void chamfer_grad_kernel(
    int b_total, int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
    const float* __restrict__ grad_dist1, const int* __restrict__ idx1,
    float* __restrict__ grad_xyz1, float* __restrict__ grad_xyz2) {
    const size_t gid = (size_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (gid >= (size_t)b_total * n) return;
    const int b = (int)(gid / n);
    const float x1 = xyz1[gid * 3 + 0], y1 = xyz1[gid * 3 + 1], z1 = xyz1[gid * 3 + 2];
    const int j2 = idx1[gid];
    const size_t o2 = ((size_t)b * m + j2) * 3;
    const float x2 = xyz2[o2 + 0], y2 = xyz2[o2 + 1], z2 = xyz2[o2 + 2];
    const float g = grad_dist1[gid] * 2;
    atomicAdd(&grad_xyz1[gid * 3 + 0], g * (x1 - x2));
    atomicAdd(&grad_xyz1[gid * 3 + 1], g * (y1 - y2));
    atomicAdd(&grad_xyz1[gid * 3 + 2], g * (z1 - z2));
    atomicAdd(&grad_xyz2[o2 + 0], -(g * (x1 - x2)));
    atomicAdd(&grad_xyz2[o2 + 1], -(g * (y1 - y2)));
    atomicAdd(&grad_xyz2[o2 + 2], -(g * (z1 - z2)));
}

// workgroups of the all-pairs kernels the chip runs at a time: 32 KiB of LDS each, 160 KiB per CU (SC_CHAMFER_SLOTS_PER_CU: tuning override)
int chamfer_slots() {
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* e = getenv("SC_CHAMFER_SLOTS_PER_CU");
        const int per_cu = e && atoi(e) > 0 ? atoi(e) : 5;
        return (cus > 0 ? cus : 256) * per_cu;
    }();
    return slots;
}

}  // namespace sc

extern "C" {

// See include/shapeclipper_hip.h for the contract.
int sc_chamfer3d_forward(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                         int32_t* idx1, int32_t* idx2, int b, int n, int m, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (b <= 0) return 0;
    if (n > 0 && m > 0) {
        dim3 g1((n + sc::CH_THREADS * sc::CH_Q - 1) / (sc::CH_THREADS * sc::CH_Q), b);
        hipLaunchKernelGGL(sc::chamfer_nn_kernel, g1, dim3(sc::CH_THREADS), 0, stream, n, xyz1, m, xyz2, dist1, idx1);
        size_t t1 = (size_t)b * n;
        hipLaunchKernelGGL(sc::chamfer_nn_index_kernel, dim3((unsigned)((t1 + sc::CH_THREADS - 1) / sc::CH_THREADS)),
                           dim3(sc::CH_THREADS), 0, stream, b, n, xyz1, m, xyz2, dist1, idx1);
        dim3 g2((m + sc::CH_THREADS * sc::CH_Q - 1) / (sc::CH_THREADS * sc::CH_Q), b);
        hipLaunchKernelGGL(sc::chamfer_nn_kernel, g2, dim3(sc::CH_THREADS), 0, stream, m, xyz2, n, xyz1, dist2, idx2);
        size_t t2 = (size_t)b * m;
        hipLaunchKernelGGL(sc::chamfer_nn_index_kernel, dim3((unsigned)((t2 + sc::CH_THREADS - 1) / sc::CH_THREADS)),
                           dim3(sc::CH_THREADS), 0, stream, b, m, xyz2, n, xyz1, dist2, idx2);
    }
    return (int)hipGetLastError();
}

// Same contract as sc_chamfer3d_forward, plus `workspace`: (b*n + b*m) uint64 of scratch.  Splits the target cloud over `nsplit`
// workgroup slices; nsplit <= 0: chosen per direction so that the launch is a whole number of full rounds of the chip (ch_auto_split).
int sc_chamfer3d_forward_split(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1,
                               int32_t* idx2, int b, int n, int m, int nsplit, void* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    unsigned long long* k1 = (unsigned long long*)workspace;
    unsigned long long* k2 = k1 + (size_t)b * n;
    (void)hipMemsetAsync(workspace, 0xFF, ((size_t)b * n + (size_t)b * m) * sizeof(unsigned long long), stream);
    const int per = sc::CH_THREADS * sc::CH_Q;
    const int slots = sc::chamfer_slots();
    auto one = [&](const float* q, int nq, const float* t, int mt, unsigned long long* keys, float* dist, int32_t* idx) {
        const int groups = (nq + per - 1) / per;
        const int ns = nsplit > 0 ? nsplit : sc::ch_auto_split((long long)groups * b, mt, slots);
        int sl = (mt + ns - 1) / ns;
        sl = (sl + sc::CH_SUB - 1) / sc::CH_SUB * sc::CH_SUB;
        dim3 g(groups, b, (mt + sl - 1) / sl);
        hipLaunchKernelGGL(sc::chamfer_nn_split_kernel, g, dim3(sc::CH_THREADS), 0, stream, nq, q, mt, t, sl, keys);
        const size_t tot = (size_t)b * nq;
        hipLaunchKernelGGL(sc::chamfer_unpack_kernel, dim3((unsigned)((tot + sc::CH_THREADS - 1) / sc::CH_THREADS)), dim3(sc::CH_THREADS), 0, stream, tot,
                           keys, dist, idx);
    };
    one(xyz1, n, xyz2, m, k1, dist1, idx1);
    one(xyz2, m, xyz1, n, k2, dist2, idx2);
    return (int)hipGetLastError();
}

int sc_chamfer3d_backward(const float* xyz1, const float* xyz2, float* gradxyz1, float* gradxyz2,
                          const float* graddist1, const float* graddist2, const int32_t* idx1,
                          const int32_t* idx2, int b, int n, int m, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    size_t t1 = (size_t)b * n, t2 = (size_t)b * m;
    hipLaunchKernelGGL(sc::chamfer_grad_kernel, dim3((unsigned)((t1 + sc::CH_THREADS - 1) / sc::CH_THREADS)),
                       dim3(sc::CH_THREADS), 0, stream, b, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
    hipLaunchKernelGGL(sc::chamfer_grad_kernel, dim3((unsigned)((t2 + sc::CH_THREADS - 1) / sc::CH_THREADS)),
                       dim3(sc::CH_THREADS), 0, stream, b, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
    return (int)hipGetLastError();
}

}  // extern "C"
