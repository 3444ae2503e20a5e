// chamfer_grid.hip -- EXACT accelerated nearest-neighbour search for Chamfer3D (same results as chamfer.hip, bit for bit).
//
// The reference (external/chamfer3D/chamfer3D.cu:12-134) and chamfer.hip evaluate all N x M ordered pairs: 8 FLOP per pair at the
// packed-fp32 issue floor of 4 vector instructions per pair (0.39 of the VALU peak, 82 ms for B=32 x 100k x 100k).  The result of
// the search -- for every query the smallest d = fmaf(dy, dy, dx*dx) + dz*dz over the other cloud and the LOWEST index attaining
// it -- does not need all pairs.  Here the target cloud is binned into a uniform grid (about CG_TPC targets per cell) and every
// query walks Chebyshev rings of cells around its own cell:
//   * a candidate is evaluated with the SAME expression (dist2 of chamfer_common.hpp, contraction off) and accepted on
//     d < best || (d == best && index < best_index): the order candidates are met in does not matter, the result is the
//     reference's "first strict minimum in index order" (chamfer3D.cu:36,46,126);
//   * after ring r every target not yet seen lies outside the (2r+1)^3 block of cells, at least `lb` away along one axis (block
//     faces on the grid boundary bound nothing: there are no targets beyond).  The walk stops when best < (lb - slack)^2 * (1 - 1e-4):
//     slack (16 ulp of the cloud's coordinate scale) covers the rounding of the binning and of the face coordinates, the factor
//     covers the rounding of d itself, so a target that could tie or beat `best` is never skipped;
//   * queries that do not terminate within the ring / candidate limits (far from the other cloud, one huge cell, non-finite input)
//     go on a list and are answered by the brute-force scan (the chamfer.hip inner loop with an index list, split over target
//     slices and merged with 64-bit atomicMin keys so that a short list still fills the chip), so the worst case is about the old
//     cost and the answer is the same.
// Two walks, chosen per image from the targets' occupancy (meta.dense, set while the histogram is scanned):
//   * volume-like targets (about CG_TPC per cell everywhere): one THREAD per query (cg_query_kernel), ~54 candidates each;
//   * surface-like targets (>= CG_DENSE per OCCUPIED cell -- what the evaluation compares): hundreds to thousands of candidates per
//     query as soon as the two surfaces are a few cells apart, which the thread walk fetches through divergent gathers at ~1 % of the
//     all-pairs kernels' pair rate (it was 2-3x SLOWER than all pairs at a mean distance of 0.05-0.1).  There the queries are binned
//     into the same grid by 2 x 2 x 2 tile of cells and one WAVE walks for up to 64 queries of a tile (cg_query_wave_kernel): uniform
//     control flow, 64 candidate ranges looked up per round trip, 64 candidates per coalesced fetch, each broadcast to the 64 queries.
// Measured at batch 1, 100,000 x 100,000 (tools/perf_chamfer_surface.py, profiles/r04_chamfer_regimes.txt): uniform volumes 0.27 ms,
// coinciding surfaces 0.30 ms, surfaces a mean 0.02 / 0.05 / 0.09 apart 0.38 / 0.77 / 1.37 ms, all pairs 2.6-2.9 ms; at 0.15-0.25 the walk
// answers a fifth of the queries and the scan the rest: 2.63-2.68 ms, what all pairs cost (it was 3.1-3.2 ms while the scan's slicing was
// fixed on the host for the longest possible list; batch 8: 13.5-15.6 ms against 19.6).  Tried and dropped: sending query tiles with no
// target within 2-4 cells straight to the scan -- the walk is not where those cases spend their time, and at 0.05-0.09 it made the search
// 1.3-2x slower (profiles/r04_chamfer_far_tile_rule_experiment.txt).
// Bound: latency / L2 gathers; the brute-force line stays in bench.py's workloads.
#include "chamfer_common.hpp"
#include <limits.h>
#include <stdlib.h>

namespace sc {

constexpr int CG_TPC = 2;          // aimed-at targets per cell
constexpr int CG_GMAX = 128;       // cells per axis
constexpr int CG_RMAX = 5;         // rings before a query is handed to the brute-force scan
constexpr int CG_BUDGET = 3072;    // candidates before a query is handed to the brute-force scan
constexpr int CG_REMPTY = 2;       // rings without any candidate before a query is handed to the scan
constexpr int CG_DENSE = 8;         // targets per occupied cell from which a cloud counts as surface-like
constexpr int CG_WAVE_RMAX = 6;     // rings of the wave walk (its candidates cost ~1/50 of the thread walk's per query)
constexpr int CG_WAVE_REMPTY = 4;   // rings without any candidate before the wave gives up
constexpr int CG_WAVE_BUDGET = 16384;   // candidates per WAVE (64 queries of one cell) before its open queries are handed to the scan

struct GridMeta {                  // one per batch element
    float lo[3], h[3], inv_h[3];
    int g[3];
    float slack;
    int valid;
    int dense;                     // >= CG_DENSE targets per OCCUPIED cell (a surface, not a volume): the queries walk in waves (5b)
};

__host__ __device__ inline int cg_cells_capacity(int m) { return m / CG_TPC * 2 + 64; }

// ---- 1. bounding box of the target cloud -> grid geometry ---------------------------------------------------------------------
// Many workgroups per image: per-wave minima / maxima go into six order-preserving integer keys with atomicMax (the minima as the
// complement of their key, so that the all-zero state the workspace memset leaves means "nothing yet" for all six); box[6] collects
// the "not a plain finite coordinate" flag.  One thread per image turns the box into the grid geometry.  (One 1024-thread workgroup
// per image took 40 us for 100,000 points -- a tenth of a whole batch-1 search.)
__device__ __forceinline__ unsigned cg_key(float f) {
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float cg_unkey(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__global__ __launch_bounds__(256) void cg_bbox_kernel(int m, const float* __restrict__ tgt, unsigned* __restrict__ box_all) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* t = tgt + (size_t)b * m * 3;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int bad = 0, any = 0;
    for (int k = blockIdx.x * 256 + tid; k < m; k += gridDim.x * 256) {
        any = 1;
        for (int a = 0; a < 3; ++a) {
            const float v = t[(size_t)k * 3 + a];
            bad |= !(fabsf(v) < 1.0e15f);          // NaN, Inf and magnitudes the padding / slack arithmetic is not made for
            mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v);
        }
    }
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d >= 1; d >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], d)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d)); }
    any = __any(any);
    bad = __any(bad);
    unsigned* box = box_all + (size_t)b * 8;
    if ((tid & 63) == 0 && any) {
        for (int a = 0; a < 3; ++a) { atomicMax(&box[a], ~cg_key(mn[a])); atomicMax(&box[3 + a], cg_key(mx[a])); }
        if (bad) atomicOr(&box[6], 1u);
    }
}

__global__ void cg_meta_kernel(int m, int n_images, const unsigned* __restrict__ box_all, GridMeta* __restrict__ meta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_images) return;
    const unsigned* box = box_all + (size_t)b * 8;
    {
        GridMeta g;
        float ext[3], scale = 0.f, emax = 0.f;
        for (int a = 0; a < 3; ++a) {
            const float lo = cg_unkey(~box[a]), hi = cg_unkey(box[3 + a]);
            g.lo[a] = lo;
            ext[a] = hi - lo;
            emax = fmaxf(emax, ext[a]);
            scale = fmaxf(scale, fmaxf(fabsf(lo), fabsf(hi)));
        }
        g.valid = (!box[6] && m > 0 && emax > 0.f) ? 1 : 0;
        // a flat or thin cloud still gets cells of a sensible size along its thin axes
        float vol = 1.f;
        for (int a = 0; a < 3; ++a) { ext[a] = fmaxf(ext[a], emax * 1.0e-3f); vol *= ext[a]; }
        float h = cbrtf(vol * (float)CG_TPC / (float)(m > 0 ? m : 1));
        const int cap = cg_cells_capacity(m);
        for (int it = 0; it < 8; ++it) {           // the per-axis ceil can overshoot the cell budget: grow h until it fits
            long long cells = 1;
            for (int a = 0; a < 3; ++a) {
                int n = (int)ceilf(ext[a] / h);
                n = n < 1 ? 1 : (n > CG_GMAX ? CG_GMAX : n);
                g.g[a] = n;
                cells *= n;
            }
            if (cells <= cap) break;
            h *= 1.26f;
        }
        if ((long long)g.g[0] * g.g[1] * g.g[2] > cap) g.valid = 0;
        for (int a = 0; a < 3; ++a) {
            g.h[a] = ext[a] / (float)g.g[a];       // cells tile the extent exactly (the last cell also takes x == hi by clamping)
            g.inv_h[a] = (float)g.g[a] / ext[a];
        }
        g.slack = 16.f * 1.1920929e-7f * (scale + emax);
        if (!g.valid) { g.g[0] = g.g[1] = g.g[2] = 1; }
        g.dense = 0;                                   // set by the scan of the target histogram
        meta[b] = g;
    }
}

__device__ __forceinline__ int cg_axis_cell(float v, float lo, float inv_h, int g) {
    const int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// ---- 2. cell of every target + histogram -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cg_count_kernel(int m, const float* __restrict__ tgt, const GridMeta* __restrict__ meta, int cap,
                                                       int* __restrict__ cell_of, int* __restrict__ counts, int only_dense) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const GridMeta& g = meta[b];
    if (only_dense && !g.dense) return;            // the query-side binning is for the wave walk only
    const float* t = tgt + ((size_t)b * m + k) * 3;
    int c = 0;
    if (g.valid) {
        const int cx = cg_axis_cell(t[0], g.lo[0], g.inv_h[0], g.g[0]), cy = cg_axis_cell(t[1], g.lo[1], g.inv_h[1], g.g[1]);
        const int cz = cg_axis_cell(t[2], g.lo[2], g.inv_h[2], g.g[2]);
        c = (cz * g.g[1] + cy) * g.g[0] + cx;       // x fastest: a run of cells along x is one contiguous run of sorted targets
        if (only_dense)                             // the QUERY side is binned by 2 x 2 x 2 TILE of cells (cg_query_wave_kernel)
            c = ((cz >> 1) * ((g.g[1] + 1) >> 1) + (cy >> 1)) * ((g.g[0] + 1) >> 1) + (cx >> 1);
    }
    cell_of[(size_t)b * m + k] = c;
    atomicAdd(&counts[(size_t)b * (cap + 1) + c], 1);
}

// ---- 3. exclusive scan of the histogram, in place; start[cells] = total -----------------------------------------------------
// Two launches over 1,024-cell blocks (one 1024-thread workgroup per image took 77 us for 50,000 cells, and the search runs it four
// times): a local exclusive scan per block + the block totals, then every block adds the totals in front of it.  The range scanned is
// [0, cells] INCLUSIVE: the entry behind the last cell is zero on entry and ends up holding the grand total.
// targets != 0 (the target histogram): also counts the occupied cells and (second launch) sets meta.dense; targets == 0 (the query
// histogram): runs for dense images only.
__global__ __launch_bounds__(1024) void cg_scan_local_kernel(const GridMeta* __restrict__ meta, int cap, int* __restrict__ counts,
                                                             int* __restrict__ block_tot, int nblk, int* __restrict__ occupied, int targets) {
    __shared__ int wtot[16];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GridMeta& g = meta[b];
    if (!targets && !g.dense) return;
    const int cells = g.g[0] * g.g[1] * g.g[2];
    if (blk * 1024 > cells) return;
    int* c = counts + (size_t)b * (cap + 1);
    const int i = blk * 1024 + tid;
    const int v = i <= cells ? c[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wtot[w];
        if (w < wave) before += t;
        total += t;
    }
    if (i <= cells) c[i] = before + incl - v;
    if (tid == 0) block_tot[(size_t)b * nblk + blk] = total;
    if (targets) {
        const unsigned long long nz = __ballot(v > 0);
        if (lane == 0 && nz) atomicAdd(&occupied[b], __builtin_popcountll(nz));
    }
}

__global__ __launch_bounds__(1024) void cg_scan_add_kernel(GridMeta* __restrict__ meta, int cap, int* __restrict__ counts,
                                                           const int* __restrict__ block_tot, int nblk, const int* __restrict__ occupied,
                                                           int targets, int dense_min) {
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const GridMeta& g = meta[b];
    if (!targets && !g.dense) return;
    const int cells = g.g[0] * g.g[1] * g.g[2];
    if (blk * 1024 > cells) return;
    __shared__ int off_s;
    if (tid < 64) {                                                                // <= 100 values in front of this block
        int part = 0;
        for (int k = tid; k < blk; k += 64) part += block_tot[(size_t)b * nblk + k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
        if (tid == 0) off_s = part;
    }
    __syncthreads();
    const int off = off_s;
    const int i = blk * 1024 + tid;
    if (i <= cells && off) counts[(size_t)b * (cap + 1) + i] += off;
    if (targets && blk == 0 && tid == 0) meta[b].dense = (g.valid && (long long)targets >= (long long)occupied[b] * dense_min) ? 1 : 0;
}

// ---- 4. targets into cell order as {x, y, z, original index}; order inside a cell is arbitrary (see the acceptance rule) ----
__global__ __launch_bounds__(256) void cg_scatter_kernel(int m, const float* __restrict__ tgt, int cap, const int* __restrict__ cell_of,
                                                         const int* __restrict__ start, int* __restrict__ cursor,
                                                         float4* __restrict__ sorted, const GridMeta* __restrict__ only_dense) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    if (only_dense && !only_dense[b].dense) return;
    const int c = cell_of[(size_t)b * m + k];
    const int pos = start[(size_t)b * (cap + 1) + c] + atomicAdd(&cursor[(size_t)b * cap + c], 1);
    const float* t = tgt + ((size_t)b * m + k) * 3;
    sorted[(size_t)b * m + pos] = make_float4(t[0], t[1], t[2], __int_as_float(k));
}

// ---- 5. the ring walk --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cg_query_kernel(int n, const float* __restrict__ qry, int m, const GridMeta* __restrict__ meta, int cap,
                                                       const int* __restrict__ start_all, const float4* __restrict__ sorted_all,
                                                       float* __restrict__ dist, int* __restrict__ idx, int* __restrict__ todo,
                                                       int* __restrict__ todo_count, unsigned long long* __restrict__ keys,
                                                       int wave_walk) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const GridMeta g = meta[b];
    if (wave_walk && g.dense) return;              // surface-like targets: the queries of this image walk in waves (cg_query_wave_kernel)
    const float* qp = qry + ((size_t)b * n + j) * 3;
    const float q[3] = {qp[0], qp[1], qp[2]};
    const int* start = start_all + (size_t)b * (cap + 1);
    const float4* sorted = sorted_all + (size_t)b * m;
    bool done = false;
    float best = __builtin_inff();
    int bidx = INT_MAX;
    if (g.valid && fabsf(q[0]) < 1.0e15f && fabsf(q[1]) < 1.0e15f && fabsf(q[2]) < 1.0e15f) {
        int c[3];
        bool outside = false;          // more than CG_REMPTY cells off the targets' bounding box: the walk could only confirm a candidate
        for (int a = 0; a < 3; ++a) {  // after many rings (its bound grows by one cell per ring, the lateral offsets do not shrink)
            c[a] = cg_axis_cell(q[a], g.lo[a], g.inv_h[a], g.g[a]);
            const float off = fmaxf(g.lo[a] - q[a], q[a] - (g.lo[a] + (float)g.g[a] * g.h[a]));
            outside |= off > (float)CG_REMPTY * g.h[a];
        }
        int seen = 0;
        for (int r = 1; r <= CG_RMAX && !done && !outside && seen <= CG_BUDGET; ++r) {
            const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.g[2] - 1), y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.g[1] - 1);
            const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.g[0] - 1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    const int row = (z * g.g[1] + y) * g.g[0];
                    // ring 1 takes the whole 3x3x3 block (ring 0 included); from ring 2 on only the shell
                    const bool full = r == 1 || z == c[2] - r || z == c[2] + r || y == c[1] - r || y == c[1] + r;
                    for (int side = 0; side < (full ? 1 : 2); ++side) {
                        int xa, xb;
                        if (full) { xa = x0; xb = x1; }
                        else {
                            xa = xb = side == 0 ? c[0] - r : c[0] + r;
                            if (xa < 0 || xa >= g.g[0]) continue;
                        }
                        const int s = start[row + xa], e = start[row + xb + 1];
                        seen += e - s;
                        for (int k = s; k < e; ++k) {
                            const float4 t = sorted[k];
                            const float d = dist2(t.x, t.y, t.z, q[0], q[1], q[2]);
                            const int ti = __float_as_int(t.w);
                            if (d < best || (d == best && ti < bidx)) { best = d; bidx = ti; }
                        }
                    }
                }
            // everything not seen yet lies beyond a face of the block that is not a face of the grid
            float lb = __builtin_inff();
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) lb = fminf(lb, q[a] - (g.lo[a] + (float)(c[a] - r) * g.h[a]));
                if (c[a] + r < g.g[a] - 1) lb = fminf(lb, (g.lo[a] + (float)(c[a] + r + 1) * g.h[a]) - q[a]);
            }
            const float safe = lb - g.slack;
            done = lb == __builtin_inff() ? bidx != INT_MAX : (safe > 0.f && best < safe * safe * 0.9999f);
            // two rings (125 cells) without a single target: the query is far from the other cloud.  Rings 3-5 are 374 more row look-ups,
            // each a dependent load -- as much chip time as the scan this query is most likely headed for anyway (mismatched clouds at
            // evaluation: a 100k x 100k pair took 2.3 ms per direction in this kernel).  Exactness is the scan's.
            if (r >= CG_REMPTY && bidx == INT_MAX) break;
        }
    }
    if (done) {
        dist[(size_t)b * n + j] = best;
        idx[(size_t)b * n + j] = bidx;
    } else {
        const int pos = atomicAdd(&todo_count[b], 1);
        todo[(size_t)b * n + pos] = j;
        keys[(size_t)b * n + pos] = ~0ull;           // the scan below publishes (distance bits, index) with atomicMin
    }
}

// ---- 5b. the ring walk, one WAVE per (2 x 2 x 2 tile of cells, 64 queries of that tile) --------------------------------------------------------
// The queries are binned into the TARGET grid as well (steps 2-4 on the query cloud, by tile of 2 x 2 x 2 cells) and a wave takes up to 64
// queries of one tile: the walk around that tile -- rows, candidate ranges, candidates -- is the same for all of them, so the control flow is wave-uniform and a
// candidate is fetched ONCE per wave (uniform address: scalar load) and tested against 64 queries.  The thread-per-query kernel above
// fetches every candidate once per query through divergent 16-byte gathers: ~70 G pairs/s against 7.7 T pairs/s of the all-pairs
// kernels, so that on SURFACE clouds a few cells apart (1,500-3,000 candidates per query: the evaluation's predicted surface against the
// ground truth, mean distance 0.05-0.1) the grid search took 6-9 ms where all pairs take 2.9.  Same acceptance rule, same stopping
// rule per lane; the wave stops when all its lanes have.
__global__ __launch_bounds__(256) void cg_items_kernel(const GridMeta* __restrict__ meta, int cap, const int* __restrict__ qstart_all,
                                                       int* __restrict__ items_all, int* __restrict__ item_count, int max_items) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    const GridMeta& g = meta[b];
    if (!g.dense || c >= g.g[0] * g.g[1] * g.g[2]) return;
    const int* qs = qstart_all + (size_t)b * (cap + 1);
    const int cnt = qs[c + 1] - qs[c];
    if (cnt <= 0) return;
    const int chunks = (cnt + 63) >> 6;
    const int at = atomicAdd(&item_count[b], chunks);
    int* items = items_all + (size_t)b * max_items * 2;
    for (int k = 0; k < chunks; ++k) { items[2 * (at + k)] = c; items[2 * (at + k) + 1] = k; }
}

__global__ __launch_bounds__(256) void cg_query_wave_kernel(int n, int m, const GridMeta* __restrict__ meta, int cap,
                                                            const int* __restrict__ start_all, const float4* __restrict__ sorted_all,
                                                            const int* __restrict__ qstart_all, const float4* __restrict__ qsorted_all,
                                                            const int* __restrict__ items_all, const int* __restrict__ item_count,
                                                            int max_items, float* __restrict__ dist, int* __restrict__ idx,
                                                            int* __restrict__ todo, int* __restrict__ todo_count,
                                                            unsigned long long* __restrict__ keys) {
#define CG_TEST(t)                                                                              \
    {                                                                                           \
        const float d = dist2((t).x, (t).y, (t).z, q[0], q[1], q[2]);                           \
        const int ti = __float_as_int((t).w);                                                   \
        if (d < best || (d == best && ti < bidx)) { best = d; bidx = ti; }                      \
    }
    __shared__ float4 cand_all[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    float4* cand = cand_all[threadIdx.x >> 6];
    const int n_items = item_count[b];
    const GridMeta g = meta[b];
    for (int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))); item < n_items; item += (int)gridDim.x * 4) {
    const int* start = start_all + (size_t)b * (cap + 1);
    const float4* sorted = sorted_all + (size_t)b * m;
    const int* qstart = qstart_all + (size_t)b * (cap + 1);
    const int cell = __builtin_amdgcn_readfirstlane(items_all[((size_t)b * max_items + item) * 2]);      // a 2 x 2 x 2 TILE of cells
    const int chunk = __builtin_amdgcn_readfirstlane(items_all[((size_t)b * max_items + item) * 2 + 1]);
    const int q0 = qstart[cell] + chunk * 64, q1 = qstart[cell + 1];
    const bool live = q0 + lane < q1;
    const float4 qv = qsorted_all[(size_t)b * n + (live ? q0 + lane : q0)];
    const float q[3] = {qv.x, qv.y, qv.z};
    const int j = __float_as_int(qv.w);
    const int tgx = (g.g[0] + 1) >> 1, tgy = (g.g[1] + 1) >> 1;
    int lo[3], hi[3];                  // the tile's cells along every axis (one cell where the grid side is odd and this is its last tile)
    lo[0] = 2 * (cell % tgx);
    lo[1] = 2 * ((cell / tgx) % tgy);
    lo[2] = 2 * (cell / (tgx * tgy));
    for (int a = 0; a < 3; ++a) hi[a] = min(lo[a] + 1, g.g[a] - 1);
    // (no "outside the bounding box" shortcut here: a wave's candidates are cheap, and a cloud that encloses the other one -- every query
    // of one direction outside the other's box -- is the common case at evaluation)
    const bool skip = !(g.valid && fabsf(q[0]) < 1.0e15f && fabsf(q[1]) < 1.0e15f && fabsf(q[2]) < 1.0e15f);
    bool done = false, hopeless = false;
    float best = __builtin_inff();
    int bidx = INT_MAX, seen = 0;
    const float hinv = fmaxf(g.inv_h[0], fmaxf(g.inv_h[1], g.inv_h[2]));       // 1 / smallest cell side
    for (int r = 1; r <= CG_WAVE_RMAX && seen <= CG_WAVE_BUDGET; ++r) {
        if (__builtin_amdgcn_readfirstlane((int)__all((int)(done || skip || hopeless || !live)))) break;
        const int x0 = max(lo[0] - r, 0), x1 = min(hi[0] + r, g.g[0] - 1);
        // The shell of ring r around the tile as a list of candidate RANGES: rows (z, y) of the block tile +- r, each one run of cells
        // along x (rows on the shell's z / y faces; ring 1 takes the whole block) or its two end cells (inner rows).  64 ranges at a
        // time: every lane looks up the two `start` entries of ITS range (one round trip for 64 ranges instead of one per row), then
        // the wave goes through the non-empty ones.
        const int sy = hi[1] - lo[1] + 1 + 2 * r, sz = hi[2] - lo[2] + 1 + 2 * r, nranges = 2 * sy * sz;
        for (int p0 = 0; p0 < nranges; p0 += 64) {
            const int p = p0 + lane;
            int rs = 0, re = 0;
            if (p < nranges) {
                const int pair = p >> 1, sd = p & 1;
                const int zc = lo[2] - r + pair / sy, yc = lo[1] - r + pair % sy;
                if (zc >= 0 && zc < g.g[2] && yc >= 0 && yc < g.g[1]) {
                    const int row = (zc * g.g[1] + yc) * g.g[0];
                    const bool full = r == 1 || zc == lo[2] - r || zc == hi[2] + r || yc == lo[1] - r || yc == hi[1] + r;
                    int xa = -1, xb = -1;
                    if (full) { if (sd == 0) { xa = x0; xb = x1; } }
                    else {
                        const int x = sd == 0 ? lo[0] - r : hi[0] + r;
                        if (x >= 0 && x < g.g[0]) xa = xb = x;
                    }
                    if (xa >= 0) { rs = start[row + xa]; re = start[row + xb + 1]; }
                }
            }
            unsigned long long nonempty = __ballot(re > rs);
            while (nonempty) {
                const int i = __builtin_ctzll(nonempty);
                nonempty &= nonempty - 1;
                const int s = __builtin_amdgcn_readlane(rs, i), e = __builtin_amdgcn_readlane(re, i);
                seen += e - s;
                // 64 candidates per trip: one coalesced 1 KB fetch (a lane each) into the wave's LDS slot, then every candidate is read
                // back at a wave-uniform address (a broadcast ds_read_b128) and tested against the 64 queries -- no dependent global load per
                // candidate (v_readlane broadcasts were tried first: the SGPR hazards behind them cost more than the LDS round trip)
                for (int base = s; base < e; base += 64) {
                    const int cnt = min(64, e - base);
                    cand[lane] = sorted[base + min(lane, cnt - 1)];                                   // this wave's 1 KB slot
                    int k = 0;
                    for (; k + 4 <= cnt; k += 4) {                                                    // four broadcast reads in flight
                        const float4 t0 = cand[k], t1 = cand[k + 1], t2 = cand[k + 2], t3 = cand[k + 3];
                        CG_TEST(t0) CG_TEST(t1) CG_TEST(t2) CG_TEST(t3)
                    }
                    for (; k < cnt; ++k) {
                        const float4 t0 = cand[k];
                        CG_TEST(t0)
                    }
                }
            }
        }
        float lb = __builtin_inff();
        for (int a = 0; a < 3; ++a) {
            if (lo[a] - r > 0) lb = fminf(lb, q[a] - (g.lo[a] + (float)(lo[a] - r) * g.h[a]));
            if (hi[a] + r < g.g[a] - 1) lb = fminf(lb, (g.lo[a] + (float)(hi[a] + r + 1) * g.h[a]) - q[a]);
        }
        const float safe = lb - g.slack;
        // a lane that is done stays done: what it meets later is strictly farther (that is what the test established)
        done = done || (lb == __builtin_inff() ? bidx != INT_MAX : (safe > 0.f && best < safe * safe * 0.9999f));
        // the bound grows by one cell per ring: a lane whose best candidate is more rings away than the walk goes stops holding the
        // wave (the scan answers it; best can still shrink, so this is a cost heuristic, not a correctness condition)
        if (!done && bidx != INT_MAX && lb != __builtin_inff())
            hopeless |= (float)r + (sqrtf(best) - safe) * hinv > (float)CG_WAVE_RMAX;
        if (r >= CG_WAVE_REMPTY && seen == 0) break;                    // empty rings: the scan is the cheaper way (uniform)
    }
    const bool answered = done && !skip;
    if (live && answered) {
        dist[(size_t)b * n + j] = best;
        idx[(size_t)b * n + j] = bidx;
    }
    // the others go on the scan's list: one atomic per wave
    const unsigned long long listed = __ballot(live && !answered);
    if (listed) {
        const int leader = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(listed));
        int base = 0;
        if (lane == leader) base = atomicAdd(&todo_count[b], __builtin_popcountll(listed));
        base = __builtin_amdgcn_readlane(base, leader);
        if (live && !answered) {
            const int pos = base + __builtin_popcountll(listed & ((1ull << lane) - 1ull));
            todo[(size_t)b * n + pos] = j;
            keys[(size_t)b * n + pos] = ~0ull;
        }
    }
    }
#undef CG_TEST
}

// ---- 6. brute-force scan of the queries on the list (chamfer.hip's inner loop; exact index inside the winning sub-block) -----
// blockIdx.z walks a slice of the targets (as chamfer_nn_split_kernel does): a list of a few thousand queries -- surface samples far from
// every target, e.g. an untrained network's prediction against the ground truth at evaluation batch size 1 -- used to be scanned by a
// handful of workgroups that each walked ALL targets (3 ms per direction whatever the list length).  Every (query, slice) publishes
// key = (float bits of d) << 32 | index with a 64-bit atomicMin: d >= 0, so the unsigned order of the bits is the numeric order and equal
// distances resolve to the lowest index -- the all-pairs kernels' rule; cg_fallback_unpack_kernel writes the winners out.
__global__ __launch_bounds__(CH_THREADS) void cg_fallback_kernel(int n, const float* __restrict__ qry, int m, const float* __restrict__ tgt_all,
                                                                 const int* __restrict__ todo, const int* __restrict__ todo_count,
                                                                 int slots, unsigned long long* __restrict__ keys) {
    __shared__ float4 tgt[CH_TCHUNK];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cnt_q = todo_count[b];
    // The list length is only known here: the slicing is chosen on the device, by the rule of the all-pairs entry point (whole rounds of
    // the chip; the other images of the batch are taken to have lists of about this length).  blockIdx.x = slice * groups + group.
    const int groups = (cnt_q + CH_THREADS * CH_Q - 1) / (CH_THREADS * CH_Q);
    if (groups == 0) return;
    const int nsl = ch_auto_split((long long)groups * gridDim.y, m, slots);
    if ((int)blockIdx.x >= groups * nsl) return;
    const int slice_len = (((m + nsl - 1) / nsl) + CH_SUB - 1) / CH_SUB * CH_SUB;
    const int qbase = ((int)blockIdx.x % groups) * (CH_THREADS * CH_Q);
    const int t_begin = ((int)blockIdx.x / groups) * slice_len, t_end = min(m, t_begin + slice_len);
    if (t_begin >= t_end) return;
    const float* q_ptr = qry + (size_t)b * n * 3;
    const float* t_ptr = tgt_all + (size_t)b * m * 3;
    float qx[CH_Q], qy[CH_Q], qz[CH_Q], best[CH_Q];
    int bblk[CH_Q];
#pragma unroll
    for (int q = 0; q < CH_Q; ++q) {
        int e = qbase + q * CH_THREADS + tid;
        e = e < cnt_q ? e : cnt_q - 1;
        const int jq = todo[(size_t)b * n + e];
        qx[q] = q_ptr[jq * 3 + 0]; qy[q] = q_ptr[jq * 3 + 1]; qz[q] = q_ptr[jq * 3 + 2];
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
            const int blk = (k0 + sb) / CH_SUB;          // slices start at multiples of CH_SUB
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
        const int e = qbase + q * CH_THREADS + tid;
        if (e >= cnt_q) continue;
        const int kb = bblk[q] * CH_SUB;
        int id = kb;          // NaN inputs: nothing compares equal; keep the sub-block start, as chamfer_nn_index_kernel does
        for (int t = CH_SUB - 1; t >= 0; --t) {
            const int k = kb + t;
            if (k < t_end && dist2(t_ptr[k * 3 + 0], t_ptr[k * 3 + 1], t_ptr[k * 3 + 2], qx[q], qy[q], qz[q]) == best[q]) id = k;
        }
        atomicMin(&keys[(size_t)b * n + e], ((unsigned long long)__float_as_uint(best[q]) << 32) | (unsigned int)id);
    }
}

__global__ __launch_bounds__(256) void cg_fallback_unpack_kernel(int n, const int* __restrict__ todo, const int* __restrict__ todo_count,
                                                                 const unsigned long long* __restrict__ keys, float* __restrict__ dist,
                                                                 int* __restrict__ idx) {
    const int b = blockIdx.y, cnt_q = todo_count[b];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt_q; e += gridDim.x * 256) {
        const unsigned long long k = keys[(size_t)b * n + e];
        const int j = todo[(size_t)b * n + e];
        dist[(size_t)b * n + j] = __uint_as_float((unsigned int)(k >> 32));
        idx[(size_t)b * n + j] = (int)(unsigned int)(k & 0xFFFFFFFFull);
    }
}

// workspace of one direction (queries [b][n], targets [b][m]), in 4-byte words
struct CgCarve {
    size_t meta, counts, cursor, cell_of, todo, todo_count, sorted, keys, qcounts, qcursor, item_count, qcell_of, qsorted, items, max_items, occupied, box, block_tot, nblk, total;
};
inline CgCarve cg_carve(int b, int n, int m) {
    CgCarve c;
    const size_t cap = (size_t)cg_cells_capacity(m);
    size_t o = 0;
    auto take = [&](size_t words) { const size_t at = o; o += (words + 3) & ~(size_t)3; return at; };
    c.counts = take((size_t)b * (cap + 1));      // counts, cursors and the two list counters are cleared by ONE memset: first and adjacent
    c.cursor = take((size_t)b * cap);
    c.todo_count = take((size_t)b);
    c.qcounts = take((size_t)b * (cap + 1));     // the query cloud binned into the same grid (histogram -> starts)
    c.qcursor = take((size_t)b * cap);
    c.item_count = take((size_t)b);
    c.occupied = take((size_t)b);
    c.box = take((size_t)b * 8);                 // bounding-box keys (zero = nothing yet)
    c.meta = take((size_t)b * (sizeof(GridMeta) / 4));
    c.cell_of = take((size_t)b * m);
    c.todo = take((size_t)b * n);
    c.sorted = take((size_t)b * m * 4);
    c.keys = take((size_t)b * n * 2);            // 64-bit (distance bits, index) of the listed queries (offsets are multiples of 4 words)
    c.qcell_of = take((size_t)b * n);
    c.qsorted = take((size_t)b * n * 4);
    c.max_items = (size_t)n / 64 + 1 + ((size_t)n < cap ? (size_t)n : cap);      // sum over cells of ceil(count / 64)
    c.items = take((size_t)b * c.max_items * 2);
    c.nblk = cap / 1024 + 1;
    c.block_tot = take((size_t)b * c.nblk);
    c.total = o;
    return c;
}

int cg_one_direction(const float* qry, int n, const float* tgt, int m, int b, float* dist, int32_t* idx, int* ws, hipStream_t stream) {
    const CgCarve c = cg_carve(b, n, m);
    const int cap = cg_cells_capacity(m);
    GridMeta* meta = reinterpret_cast<GridMeta*>(ws + c.meta);
    (void)hipMemsetAsync(ws, 0, c.meta * sizeof(int), stream);
    const int bbox_blocks = (m + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(cg_bbox_kernel, dim3(bbox_blocks < 64 ? bbox_blocks : 64, b), dim3(256), 0, stream, m, tgt, reinterpret_cast<unsigned*>(ws + c.box));
    hipLaunchKernelGGL(cg_meta_kernel, dim3((b + 63) / 64), dim3(64), 0, stream, m, b, reinterpret_cast<const unsigned*>(ws + c.box), meta);
    hipLaunchKernelGGL(cg_count_kernel, dim3((m + 255) / 256, b), dim3(256), 0, stream, m, tgt, meta, cap, ws + c.cell_of, ws + c.counts, 0);
    hipLaunchKernelGGL(cg_scan_local_kernel, dim3((unsigned)c.nblk, b), dim3(1024), 0, stream, meta, cap, ws + c.counts, ws + c.block_tot, (int)c.nblk,
                       ws + c.occupied, m);
    static const int dense_min = [] { const char* e = getenv("SC_CHAMFER_GRID_DENSE"); return e ? atoi(e) : CG_DENSE; }();      // tuning override
    hipLaunchKernelGGL(cg_scan_add_kernel, dim3((unsigned)c.nblk, b), dim3(1024), 0, stream, meta, cap, ws + c.counts, ws + c.block_tot, (int)c.nblk,
                       ws + c.occupied, m, dense_min);
    hipLaunchKernelGGL(cg_scatter_kernel, dim3((m + 255) / 256, b), dim3(256), 0, stream, m, tgt, cap, ws + c.cell_of, ws + c.counts,
                       ws + c.cursor, reinterpret_cast<float4*>(ws + c.sorted), nullptr);
    static const int wave_walk = [] { const char* e = getenv("SC_CHAMFER_GRID_WAVE"); return e ? atoi(e) : 1; }();      // A/B: 0 = thread per query
    if (wave_walk) {       // images whose targets are surface-like (meta.dense): queries binned into the same grid, one wave per (cell, 64 queries)
        hipLaunchKernelGGL(cg_count_kernel, dim3((n + 255) / 256, b), dim3(256), 0, stream, n, qry, meta, cap, ws + c.qcell_of, ws + c.qcounts, 1);
        hipLaunchKernelGGL(cg_scan_local_kernel, dim3((unsigned)c.nblk, b), dim3(1024), 0, stream, meta, cap, ws + c.qcounts, ws + c.block_tot,
                           (int)c.nblk, ws + c.occupied, 0);
        hipLaunchKernelGGL(cg_scan_add_kernel, dim3((unsigned)c.nblk, b), dim3(1024), 0, stream, meta, cap, ws + c.qcounts, ws + c.block_tot,
                           (int)c.nblk, ws + c.occupied, 0, 0);
        hipLaunchKernelGGL(cg_scatter_kernel, dim3((n + 255) / 256, b), dim3(256), 0, stream, n, qry, cap, ws + c.qcell_of, ws + c.qcounts,
                           ws + c.qcursor, reinterpret_cast<float4*>(ws + c.qsorted), meta);
        hipLaunchKernelGGL(cg_items_kernel, dim3((cap + 255) / 256, b), dim3(256), 0, stream, meta, cap, ws + c.qcounts, ws + c.items,
                           ws + c.item_count, (int)c.max_items);
        hipLaunchKernelGGL(cg_query_wave_kernel, dim3((unsigned)((c.max_items + 3) / 4 < 4096 ? (c.max_items + 3) / 4 : 4096), b), dim3(256), 0, stream, n, m, meta, cap, ws + c.counts,
                           reinterpret_cast<const float4*>(ws + c.sorted), ws + c.qcounts, reinterpret_cast<const float4*>(ws + c.qsorted),
                           ws + c.items, ws + c.item_count, (int)c.max_items, dist, idx, ws + c.todo, ws + c.todo_count,
                           reinterpret_cast<unsigned long long*>(ws + c.keys));
    }
    hipLaunchKernelGGL(cg_query_kernel, dim3((n + 255) / 256, b), dim3(256), 0, stream, n, qry, m, meta, cap, ws + c.counts,
                       reinterpret_cast<const float4*>(ws + c.sorted), dist, idx, ws + c.todo, ws + c.todo_count,
                       reinterpret_cast<unsigned long long*>(ws + c.keys), wave_walk);
    // the scan of the listed queries: as many workgroups as the longest possible list needs under the slicing rule (the kernel picks the
    // slicing from the actual list length; the rest exit at once)
    const int groups = (n + CH_THREADS * CH_Q - 1) / (CH_THREADS * CH_Q);
    const int slots = chamfer_slots();
    long long max_wgs = 1;
    for (int gq = 1; gq <= groups; ++gq) {
        const long long w = (long long)gq * ch_auto_split((long long)gq * b, m, slots);
        max_wgs = w > max_wgs ? w : max_wgs;
    }
    hipLaunchKernelGGL(cg_fallback_kernel, dim3((unsigned)max_wgs, b), dim3(CH_THREADS), 0, stream, n, qry, m, tgt, ws + c.todo,
                       ws + c.todo_count, slots, reinterpret_cast<unsigned long long*>(ws + c.keys));
    hipLaunchKernelGGL(cg_fallback_unpack_kernel, dim3(groups < 64 ? groups : 64, b), dim3(256), 0, stream, n, ws + c.todo, ws + c.todo_count,
                       reinterpret_cast<const unsigned long long*>(ws + c.keys), dist, idx);
    return (int)hipGetLastError();
}

}  // namespace sc

extern "C" {

// See include/shapeclipper_hip.h for the contract.
long long sc_chamfer3d_grid_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    const size_t a = sc::cg_carve(b, n, m).total, c = sc::cg_carve(b, m, n).total;
    return (long long)((a > c ? a : c) * sizeof(int));
}

int sc_chamfer3d_forward_grid(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, int b,
                              int n, int m, void* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    int code = sc::cg_one_direction(xyz1, n, xyz2, m, b, dist1, idx1, (int*)workspace, stream);
    if (code) return code;
    return sc::cg_one_direction(xyz2, m, xyz1, n, b, dist2, idx2, (int*)workspace, stream);
}

}  // extern "C"
