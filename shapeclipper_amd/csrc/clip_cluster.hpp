// clip_cluster.hpp -- the transformer layers of the CLIP ViT-B image tower at SMALL batch as ONE launch (included by clip_vit.hip).
//
// Replaces, for 26 <= B <= 32 images of T <= 64 tokens (ViT-B/32 at 224 x 224: T = 50; the annotator's batch), the 7 dependent launches per
// layer of clip_vit_forward (LayerNorm, qkv GEMM, attention, out-projection, LayerNorm, fc1, fc2: 84 launches per tower, ~4.6 us of cold
// start each whatever the shape, profiles/r04_clip_gaps_32_B_32.txt) behind `clip_encoder.encode_image(image)`
// (/root/reference/CLIP_anno.py:166).  DESIGN.md 4.3.1 has the measurements (0.99 -> 0.86 ms per batch-32 tower) and what bounds it.
//
// Decomposition BY IMAGE.  A transformer layer is row-local except attention, and attention is image-local: nothing in the tower needs a
// chip-wide synchronisation.  An image's 50 rows (one padded 64-row MFMA band) are owned by a CLUSTER of 8 workgroups = 8 CUs of ONE XCD
// (workgroup b is dispatched to XCD b % 8; 256 CUs = 32 clusters = 32 images in flight); member c of a cluster computes, for its image,
//   qkv      the 192 columns (q | k | v) of head c and, for c < 4, of head c + 8     -> attention of those heads inside the same CU
//   out-proj columns [96 c, 96 c + 96) of x +=                                        (A = the attention rows of all 8 members)
//   fc1      columns [384 c, 384 c + 384) of quick_gelu(.)
//   fc2      columns [96 c, 96 c + 96) of x +=                                        (A = the hidden rows of all 8 members)
// and the members meet at FOUR cluster barriers per layer (one counter per cluster; XCD-local when the members share an XCD -- checked from
// the hardware XCC id -- and at agent scope otherwise: correct wherever the workgroups land).  Every weight byte is read by exactly one
// member of a cluster; the four clusters of an XCD walk the same weight stream (L2 hits for three of them).
//
// GEMM core: the activation band is SMALL (64 x 768 16-bit = 96 KB) and every wave needs all of it, the weight slice is LARGE and each
// element is needed once -- so neither goes through LDS: the four waves split K (wave q owns K-quarter q: its 64 x 192 piece of the band
// lives in 96 registers for the whole phase), weight fragments go global -> registers from a RE-PACKED image (a fragment = one contiguous
// KB; a ring of chunks per wave) and feed v_mfma_f32_16x16x32 directly (W fragment as the first operand: a lane ends up with 4 consecutive
// output columns of one row).  The four K-partials of a 64 x 32 block are added in a fixed order through LDS (wave w finishes row block w)
// and leave through the epilogue (bias, 16-bit / quick_gelu / x +=) under the next step's MFMAs.  LayerNorm is recomputed by every member
// (two-pass, fp32) into an LDS band the waves read their operand registers from.
//
// A lost workgroup (a device that cannot hold the whole grid at once) would leave the others spinning: every wait is bounded (0.2 s), the
// first time-out raises the error word, every later wait returns at once and the images of the launch are poisoned with NaN -- loud, no hang.
#pragma once

namespace sc {
namespace cl {

typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef SC_CL_ABLATE
#define SC_CL_ABLATE 0                // experiments of tools/prof_clip_cluster.py (never set in the product build): 1 = no weight loads
#endif
#ifndef SC_CL_PROF
#define SC_CL_PROF 0                  // 1 (tools/build_variants.sh clip_vit.hip SC_CL_PROF 1): workgroup 0 stamps the 100 MHz clock at every phase boundary
#endif
#if SC_CL_PROF
__device__ long long g_prof[16 * 16];
#define SC_CL_STAMP(I) if (blockIdx.x == 0 && threadIdx.x == 0 && l < 16) g_prof[l * 16 + (I)] = wall_clock64();
#else
#define SC_CL_STAMP(I)
#endif

constexpr int CL = 8;                 // workgroups (CUs) per image
constexpr int CD = 768, CMLP = 3072, CHEADS = 12;
constexpr int KQ = 192, KS = KQ / 32; // K-quarter of a 768-wide operand band per wave, 32-wide MFMA sub-steps in it
#ifndef SC_CL_LN_ROWS
#define SC_CL_LN_ROWS 7          // rows of a wave normalised per round trip (7: two round trips at 50 tokens; 13 / 16: one)
#endif
#ifndef SC_CL_RING
#define SC_CL_RING 3
#endif
constexpr int NR = SC_CL_RING;        // weight chunks of a wave's ring (NR - 1 in flight)
constexpr int pad_to_ring(int n) { return (n + NR - 1) / NR * NR; }
constexpr int P3 = pad_to_ring(3), P12 = pad_to_ring(12);
constexpr int VT_LD = 64 + 4;
constexpr int BAND_BYTES = 64 * (CD * 2 + 16);            // LayerNorm operand band (97 KB); the two K-partial buffers (64 KB) alias its start
constexpr int LDS_BYTES = BAND_BYTES + 2 * 64 * VT_LD * 2 + 384 * 4;      // 97 KB + 17 KB + 1.5 KB: one workgroup per CU
constexpr long long SPIN_TICKS = 20000000LL;             // 0.2 s of the 100 MHz wall clock

template <bool H>
__device__ __forceinline__ f32x4 mma(const uint4& w, const uint4& a, const f32x4& c) {
    if (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}
template <bool H>
__device__ __forceinline__ uint32_t pk2(float lo, float hi) { return (uint32_t)cvt16<H>(lo) | ((uint32_t)cvt16<H>(hi) << 16); }

struct WChunk { uint4 v[2][KS]; };          // 32 weight rows (two 16-row MFMA blocks) x this wave's K-quarter

// The fp32 vectors of one layer inside the tower's weight image (clip_vit_forward: same order)
struct Layer { const float *ln1_g, *ln1_b, *b_qkv, *b_o, *ln2_g, *ln2_b, *b_fc1, *b_fc2; };
__device__ __forceinline__ Layer layer_at(const float* wf, int l) {
    Layer L;
    wf += (size_t)l * (9 * CD + CMLP);
    L.ln1_g = wf; L.ln1_b = wf + CD; L.b_qkv = wf + 2 * CD; L.b_o = wf + 5 * CD; L.ln2_g = wf + 6 * CD; L.ln2_b = wf + 7 * CD;
    L.b_fc1 = wf + 8 * CD; L.b_fc2 = wf + 8 * CD + CMLP;
    return L;
}

// The weight stream of a member: per layer n1 = 6 (one head) or 12 (two heads) chunks of qkv, 3 of the out-projection, 12 of fc1, 12 of fc2
// (4 K-passes x 3 column pairs), the layers one after the other.  A chunk = 32 weight rows x one 768-wide K band; wave q needs its K-quarter
// of it as 12 MFMA fragments (2 row blocks x 6 sub-steps; lane = 16 (k / 8 % 4) + row % 16 holds 8 consecutive k of one row).  Read from the
// row-major matrices that is 64 scattered 16-byte accesses per wave instruction (every lane another row, 1.5 KB apart): measured 1.5 us per
// chunk step, 30 GB/s per CU.  The layers are therefore RE-PACKED once per model (sc_clip_cluster_pack) in consumption order
//     [layer][member][chunk][wave q][row block][sub-step][lane][8 values]
// so that a fragment is one contiguous KB, a wave's chunk 12 consecutive KB and a member's layer one contiguous stream.
constexpr int FRAG_ELEMS = 512, WAVE_CHUNK_ELEMS = 12 * FRAG_ELEMS, CHUNK_ELEMS = 4 * WAVE_CHUNK_ELEMS;     // 1 KB, 12 KB, 48 KB
constexpr int LAYER_CHUNKS = 288;                                  // 12 * 768 * 768 values = 4 x 39 + 4 x 33 chunks
constexpr size_t LAYER_ELEMS = (size_t)LAYER_CHUNKS * CHUNK_ELEMS;
__host__ __device__ __forceinline__ int member_chunks(int c) { return (c + 8 < CHEADS ? 12 : 6) + 27; }
__host__ __device__ __forceinline__ int member_first_chunk(int c) { return c <= 4 ? 39 * c : 156 + 33 * (c - 4); }
// source of chunk `idx` of member c: matrix offset inside the layer (row-major image of clip_vit_forward), first row, row pitch, K offset
__host__ __device__ __forceinline__ void chunk_source(int c, int idx, size_t& mat, int& n0, int& ld, int& koff) {
    const int n1 = c + 8 < CHEADS ? 12 : 6;
    koff = 0;
    if (idx < n1) {
        const int hs = idx >= 6, j = idx - 6 * hs, head = hs ? c + 8 : c;
        mat = 0; n0 = (j >> 1) * CD + head * 64 + (j & 1) * 32; ld = CD;
    } else if (idx < n1 + 3) {
        mat = (size_t)3 * CD * CD; n0 = 96 * c + 32 * (idx - n1); ld = CD;
    } else if (idx < n1 + 15) {
        mat = (size_t)4 * CD * CD; n0 = 384 * c + 32 * (idx - n1 - 3); ld = CD;
    } else {
        const int j = idx - n1 - 15, pass = j / 3, pair = j - 3 * pass;
        mat = (size_t)4 * CD * CD + (size_t)CMLP * CD; n0 = 96 * c + 32 * pair; ld = CMLP; koff = pass * CD;
    }
}
// one thread per 16-byte piece of the packed image
__global__ __launch_bounds__(256) void cluster_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int layers) {
    const size_t pieces = (size_t)layers * LAYER_ELEMS / 8;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pieces; p += (size_t)gridDim.x * 256) {
        const int lane = (int)(p & 63);
        size_t r = p >> 6;
        const int s = (int)(r % 6); r /= 6;
        const int cb = (int)(r & 1); r >>= 1;
        const int kq = (int)(r & 3); r >>= 2;
        const int chunk = (int)(r % LAYER_CHUNKS), l = (int)(r / LAYER_CHUNKS);
        int c = 0;
        while (c < 7 && chunk >= member_first_chunk(c + 1)) ++c;
        size_t mat; int n0, ld, koff;
        chunk_source(c, chunk - member_first_chunk(c), mat, n0, ld, koff);
        const bf16_t* src = w + (size_t)l * (12 * CD * CD) + mat + (size_t)(n0 + 16 * cb + (lane & 15)) * ld + koff + kq * KQ + 32 * s + 8 * (lane >> 4);
        reinterpret_cast<uint4*>(out)[p] = *reinterpret_cast<const uint4*>(src);
    }
}

// Entries of a member's layer as the ring sees them: every phase (n1 qkv chunks, 3 out-projection, 12 fc1, 12 fc2) is padded with BUBBLES to a
// multiple of the ring length (fc2: each of its four K-passes of 3 chunks), so that each phase starts at ring slot 0 and every slot index is a compile-time constant.  A bubble is a
// prefetch and nothing else.
struct Stream {
    const bf16_t* wc; int layers, first, n1, n1p, kq, lane;
    __device__ __forceinline__ void load(WChunk& q, int l, int e) const {
        if (SC_CL_ABLATE & 1) { if (l >= 0) return; }             // experiment: no weight loads (the registers keep what they held)
        const int per = n1p + P3 + P12 + 4 * P3;
        if (e >= per) { e -= per; ++l; }
        if (l >= layers) { l = layers - 1; e = per - 1; }            // past the end: a valid chunk again (never consumed) -- the prefetch
                                                                     // stays unconditional, so the memory counter arithmetic never forks
        int chunk, r;
        bool real;
        if (e < n1p) { real = e < n1; chunk = min(e, n1 - 1); }
        else if ((r = e - n1p) < P3) { real = r < 3; chunk = n1 + min(r, 2); }
        else if ((r -= P3) < P12) { real = r < 12; chunk = n1 + 3 + min(r, 11); }
        else { r -= P12; const int pass = r / P3, k = r - pass * P3; real = k < 3; chunk = n1 + 15 + 3 * pass + min(k, 2); }
        // a bubble asks for ONE fragment of its phase's last chunk twelve times (a first-level cache hit after the first): the number of
        // loads per step stays the same, the traffic does not grow
        const unsigned stride = real ? 1024u : 0u;
        const char* base = reinterpret_cast<const char*>(wc + (size_t)l * LAYER_ELEMS + (size_t)(first + chunk) * CHUNK_ELEMS + kq * WAVE_CHUNK_ELEMS);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < KS; ++s) q.v[cb][s] = *reinterpret_cast<const uint4*>(base + (size_t)((cb * KS + s) * stride) + (unsigned)(lane * 16));
        // (non-temporal loads here -- global_load ... nt -- were 12 % slower and, once in four processes, not reproducible run to run: dropped)
    }
};

// Row blocks are ROTATED per wave: operand / accumulator index i of wave w is row block (w + i) & 3, so that the block a wave finishes
// (row block w) is always its index 0 -- a compile-time register, not a run-time choice.
template <bool H16>
__device__ __forceinline__ void mma_half(f32x4 (&acc)[4], const uint4 (&w)[KS], const uint4 (&act)[4][KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mma<H16>(w[s], act[i][s], acc[i]);
}

// K-partials of a 64 x 32 block.  Wave p parks the three row blocks it does not finish (index i = 1..3 = row block (p + i) & 3; lane-linear
// float4: conflict-free) and keeps its own; after the barrier wave w adds, to its own partial of row block w, those of waves w + 1, w + 2,
// w + 3 (mod 4) -- a fixed order per row block.  24 KB written + 24 KB read per step and CU (the first form parked and re-read all four: the
// LDS time of a step, 0.2 us, was as long as two thirds of its MFMAs and did not run under them).
constexpr int RED_FLOATS = 4 * 2 * 3 * 64 * 4;          // [wave][column block][i - 1][lane] float4 = 24 KB per buffer
__device__ __forceinline__ void put_half(float* red, int cb, const f32x4 (&acc)[4], int wave, int lane) {
    f32x4* dst = reinterpret_cast<f32x4*>(red) + ((wave * 2 + cb) * 3) * 64 + lane;
#pragma unroll
    for (int i = 1; i < 4; ++i) dst[(i - 1) * 64] = acc[i];
}
__device__ __forceinline__ void get_sums(const float* red, const f32x4 (&own)[2], int wave, int lane, f32x4 (&out)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        f32x4 t = own[cb];
#pragma unroll
        for (int d = 1; d < 4; ++d) {            // source wave p = (wave + d) & 3 holds row block `wave` at its index (wave - p) & 3 = 4 - d
            const int p = (wave + d) & 3;
            t = t + reinterpret_cast<const f32x4*>(red)[((p * 2 + cb) * 3 + (3 - d)) * 64 + lane];
        }
        out[cb] = t;
    }
}

// LayerNorm of the image's T rows -> the 16-bit operand band in LDS (row pitch BAND_PITCH bytes).  layernorm_kernel's arithmetic and
// two-pass form (a row in the registers of one wave: three float4 per lane, mean first, then the centred squares; sums by wave_sum).
// Wave w takes rows w, w + 4, ..., seven at a time: their loads leave in one round trip.  (First form of this phase: every wave normalised
// its own MFMA-layout piece of x straight into the operand registers -- 48 float4 loads per lane with their arithmetic on 96 + 96 registers;
// under that pressure the compiler issued the loads two at a time, ~24 dependent round trips per LayerNorm.)
constexpr int BAND_PITCH = CD * 2 + 16;
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes, every lane gets it: four DPP steps inside each row of 16 lanes, the four row totals through scalar registers (no
// LDS traffic; the 2 x 6 ds_bpermute butterflies per row of the first form cost 5-8 us per LayerNorm at 13 rows per wave)
__device__ __forceinline__ float wave_sum(float v) {
    if (SC_CL_ABLATE & 2) return v;
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
template <bool H16>
__device__ __forceinline__ void ln_to_band(char* band, const float* xi, int T, float eps, const float* g, const float* b, int wave, int lane) {
    constexpr int RB = SC_CL_LN_ROWS;
    float4 gg[3], bb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gg[k] = reinterpret_cast<const float4*>(g)[k * 64 + lane];
        bb[k] = reinterpret_cast<const float4*>(b)[k * 64 + lane];
    }
    for (int r0 = wave; r0 < T; r0 += 4 * RB) {
        float4 v[RB][3];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float4* xr = reinterpret_cast<const float4*>(xi + (size_t)min(r0 + 4 * i, T - 1) * CD);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[i][k] = (SC_CL_ABLATE & 4) ? make_float4(lane, i, k, 1.f) : xr[k * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the seven rows side by side: their butterflies are independent chains (one row after the other, the 2 x 6 dependent cross-lane
        // steps per row -- 13 rows per wave -- were most of this phase: 9.7 us)
        float sm[RB], ss[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            sm[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) sm[i] += (v[i][k].x + v[i][k].y) + (v[i][k].z + v[i][k].w);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) sm[i] = wave_sum(sm[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float mean = sm[i] / CD;
            ss[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[i][k].x -= mean; v[i][k].y -= mean; v[i][k].z -= mean; v[i][k].w -= mean;
                ss[i] += (v[i][k].x * v[i][k].x + v[i][k].y * v[i][k].y) + (v[i][k].z * v[i][k].z + v[i][k].w * v[i][k].w);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) ss[i] = wave_sum(ss[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float inv = rsqrtf(ss[i] / CD + eps);
            const int row = r0 + 4 * i;
            if (row < T && !((SC_CL_ABLATE & 8) && lane != 0)) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float o0 = v[i][k].x * inv * gg[k].x + bb[k].x, o1 = v[i][k].y * inv * gg[k].y + bb[k].y;
                    const float o2 = v[i][k].z * inv * gg[k].z + bb[k].z, o3 = v[i][k].w * inv * gg[k].w + bb[k].w;
                    *reinterpret_cast<uint2*>(band + row * BAND_PITCH + (k * 64 + lane) * 8) = make_uint2(pk2<H16>(o0, o1), pk2<H16>(o2, o3));
                }
            }
        }
    }
}
// Operand registers of this wave = its K-quarter of the band: index i = rows 16 ((w + i) & 3) + (lane & 15) (the wave index IS the K-quarter),
// 8 consecutive K values per lane and sub-step.  Rows
// past T - 1 hold whatever the LDS held: their products are never stored and never enter another row's result.
__device__ __forceinline__ void act_from_band(uint4 (&act)[4][KS], const char* band, int kq, int lane) {
    const char* p = band + (lane & 15) * BAND_PITCH + (kq * KQ + 8 * (lane >> 4)) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < KS; ++s) act[i][s] = *reinterpret_cast<const uint4*>(p + ((kq + i) & 3) * 16 * BAND_PITCH + 64 * s);
}
// ... = 16-bit rows src[row][koff + ...] (row pitch ld); rows past T - 1 repeat row T - 1
__device__ __forceinline__ void act_from_rows(uint4 (&act)[4][KS], const bf16_t* src, int ld, int koff, int T, int kq, int lane) {
    const int m16 = lane & 15, k8 = koff + kq * KQ + 8 * (lane >> 4);
    const bf16_t* p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = src + (size_t)min(16 * ((kq + i) & 3) + m16, T - 1) * ld + k8;
    // requested in the order the MFMAs consume them (sub-step major): loads retire in issue order, so the first products wait for 4 of
    // the 24 fragments instead of 19, and the rest of the transfer (24 KB per wave) runs under them
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) act[i][s] = *reinterpret_cast<const uint4*>(p[i] + 32 * s);
}

// Bounded wait of a cluster: all CL members have arrived `target / CL` times.  Thread 0 publishes and acquires for the workgroup; the two
// workgroup barriers order the other waves' accesses around it (their stores are acknowledged by L2 before the first: the explicit
// s_waitcnt vmcnt(0) every thread executes on entry).
//   local = false  release / acquire fences at agent scope (L2 write-back + invalidate: buffer_wbl2 sc1 / buffer_inv sc1) -- correct wherever
//                  the members run; measured ~4 us per barrier inside this kernel
//   local = true   the members share one XCD, i.e. one L2 (checked at kernel start from the hardware XCC id): what a member wrote is in that
//                  L2 once its stores are acknowledged, so publishing is the counter increment alone and acquiring is the invalidation of
//                  this CU's vector L1 (buffer_inv sc0: the workgroup-scope invalidate of the threadgroup-split memory model)
#ifndef SC_CL_BARRIER
#define SC_CL_BARRIER 1               // 0: never take the local form (A/B)
#endif
__device__ __forceinline__ void cluster_barrier(unsigned* cnt, unsigned target, unsigned* err, bool local) {
    // Every wave's global stores must be acknowledged by L2 BEFORE thread 0 publishes the arrival: the workgroup barrier below does not
    // wait on vmcnt by itself (the compiler emits only lgkmcnt(0) in front of s_barrier outside threadgroup-split mode -- ADVICE r05,
    // read from the ISA), and the local form has no release fence that would.  tools/scan_store_hazard.py checks the instruction stays.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!local) __threadfence();
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > SPIN_TICKS) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        if (local) asm volatile("buffer_inv sc0" ::: "memory");
        else __threadfence();
    }
    __syncthreads();
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// Attention of one (image, head) by the two waves of a pair -- attention_small_kernel's arithmetic (one pass, both 32-key score tiles in
// registers, V^T in LDS).  `active` = false: the pair only keeps the workgroup barrier company.
template <bool H16>
__device__ __forceinline__ void attention_pair(const bf16_t* qkv_i, bf16_t* att_i, int T, int hd, bool active, bf16_t* Vt, int ptid) {
    constexpr int D = CD, ldv = VT_LD;
    const size_t rs = (size_t)3 * D;
    const bf16_t* base = qkv_i + (size_t)hd * 64;
    const int lane = ptid & 63, pw = ptid >> 6, n = lane & 31, h = lane >> 5;
    auto frag = [&](int row, int col_off) -> bf16x8 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (active && row < T) v = *reinterpret_cast<const uint4*>(base + (size_t)row * rs + col_off + 8 * h);
        return __builtin_bit_cast(bf16x8, v);
    };
    const int q = pw * 32 + n;
    bf16x8 qf[4], kf[2][4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        qf[s4] = frag(q, 16 * s4);
        kf[0][s4] = frag(n, D + 16 * s4);
        kf[1][s4] = frag(32 + n, D + 16 * s4);
    }
    if (active)
        for (int e = ptid; e < 64 * 8; e += 128) {
            const int key = e >> 3, dc = (e & 7) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < T) v = *reinterpret_cast<const uint4*>(base + (size_t)key * rs + 2 * D + dc);
            const bf16_t* pv = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[(dc + j) * ldv + key] = pv[j];
        }
    const float NEG = -3.0e38f, scale = 0.125f;
    float v[2][16];
    float m = NEG;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) S = mfma16<H16>(kf[c][s4], qf[s4], S);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
            v[c][r] = key < T ? S[r] * scale : NEG;
            m = fmaxf(m, v[c][r]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[c][r] = v[c][r] > NEG ? __expf(v[c][r] - m) : 0.f;
            l += v[c][r];
        }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    __syncthreads();                                   // both V^T images are complete
    if (!active) return;
    f32x16 O[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16_t tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = cvt16<H16>(v[c][8 * j + e] * inv);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, tmp);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16_t* vp = Vt + (32 * t + n) * ldv + 32 * c + 16 * j + 4 * h;
                const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 8);
                const uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
                O[t] = mfma16<H16>(__builtin_bit_cast(bf16x8, pk), pf, O[t]);
            }
        }
    if (q < T) {
        bf16_t* orow = att_i + (size_t)q * D + hd * 64 + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<uint2*>(orow + 32 * t + 8 * r4) =
                    make_uint2(pk2<H16>(O[t][4 * r4], O[t][4 * r4 + 1]), pk2<H16>(O[t][4 * r4 + 2], O[t][4 * r4 + 3]));
    }
}

enum { E_QKV = 0, E_RESID = 1, E_GELU = 2, E_ACC = 3 };

// Epilogue of a 64 x 32 block [rows of this image] x [n0, n0 + 32): wave w owns rows 16 w + (lane & 15), a lane 4 consecutive columns per
// column block.  It issues NO vector-memory load: the memory counter retires in issue order, so a load issued behind the weight prefetch makes
// its wait drain the whole ring (the first builds did exactly that -- the compiler sinks such loads to their use).  The phase's bias slice is
// staged in LDS when the phase starts (`bias_l`: the chunk's 32 floats), the residual values of x += are requested with the phase's operand
// loads.  `t_store` = T, or 0 for a block that is not to be stored (the step in front of a phase's first one; an fc2 pass that is not the last).
//   E_QKV   16-bit(sum + bias)                 E_GELU  16-bit(quick_gelu(sum + bias))
//   E_RESID x = res + (sum + bias)             E_ACC   res += sum (valid blocks); x = res + bias where stored (fc2: four K-passes)
template <bool H16, int EPI>
__device__ __forceinline__ void epilogue(const f32x4 (&sum)[2], const float* bias_l, float4 (&res)[2], int n0, void* out_i, int ldo,
                                         int t_store, bool valid, int wave, int lane) {
    const int row = 16 * wave + (lane & 15);
    float4 bv[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) bv[cb] = *reinterpret_cast<const float4*>(bias_l + 16 * cb + 4 * (lane >> 4));
    if (EPI == E_ACC) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            res[cb].x += valid ? sum[cb][0] : 0.f; res[cb].y += valid ? sum[cb][1] : 0.f;
            res[cb].z += valid ? sum[cb][2] : 0.f; res[cb].w += valid ? sum[cb][3] : 0.f;
        }
    }
    if (row >= t_store) return;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const unsigned n = (unsigned)(n0 + 16 * cb + 4 * (lane >> 4)), o = (unsigned)row * (unsigned)ldo + n;
        float v0 = sum[cb][0] + bv[cb].x, v1 = sum[cb][1] + bv[cb].y, v2 = sum[cb][2] + bv[cb].z, v3 = sum[cb][3] + bv[cb].w;
        if (EPI == E_ACC) {
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(out_i) + o * 4u) =
                make_float4(res[cb].x + bv[cb].x, res[cb].y + bv[cb].y, res[cb].z + bv[cb].z, res[cb].w + bv[cb].w);
        } else if (EPI == E_RESID) {
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(out_i) + o * 4u) =
                make_float4(v0 + res[cb].x, v1 + res[cb].y, v2 + res[cb].z, v3 + res[cb].w);
        } else {
            if (EPI == E_GELU) {          // x * sigmoid(1.702 x); the quotient by v_rcp_f32 (1 ulp) -- the result is rounded to 16 bits
                v0 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v0)); v1 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v1));
                v2 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v2)); v3 *= __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v3));
            }
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(out_i) + o * 2u) = make_uint2(pk2<H16>(v0, v1), pk2<H16>(v2, v3));
        }
    }
}
// this lane's residual values of the member's 96 columns of x (three chunks): only this member ever writes them
__device__ __forceinline__ void resid_request(float4 (&res)[3][2], const float* xi, int c, int T, int wave, int lane) {
    const unsigned row = (unsigned)min(16 * wave + (lane & 15), T - 1);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            res[p][cb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xi) +
                                                          (row * (unsigned)CD + (unsigned)(96 * c + 32 * p + 16 * cb + 4 * (lane >> 4))) * 4u);
}

// One step of a phase (entry E of the layer, ring slot J): the ring NR - 1 entries ahead; the K-partials of the PREVIOUS step are fetched
// from LDS, added and sent through its epilogue while this step's 48 MFMAs run (one after the other the two cost 0.7 us per step with the
// matrix pipe idle for more than half of it); this step's partials go to the other LDS buffer; one workgroup barrier.
#define SC_CL_STEP(J, E, EPI, BIASPREV, RESPREV, N0PREV, OUT, LDO, TPREV, VALIDPREV)             \
    {                                                                                          \
        st.load(ring[((J) + NR - 1) % NR], l, (E) + NR - 1);                                   \
        f32x4 psum[2];                                                                         \
        if (!(SC_CL_ABLATE & 32)) get_sums(red + (par ^ 1) * RED_FLOATS, own, wave, lane, psum); else { psum[0] = own[0]; psum[1] = own[1]; } \
        float* rbuf = red + par * RED_FLOATS;                                                  \
        f32x4 a0[4], a1[4];                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { a0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[i] = a0[i]; } \
        if (!(SC_CL_ABLATE & 16)) mma_half<H16>(a0, ring[J].v[0], act);                        \
        else a0[0][0] = __uint_as_float(ring[J].v[0][0].x ^ ring[J].v[0][5].w ^ act[0][0].x ^ act[3][5].w);  \
        if (!(SC_CL_ABLATE & 32)) put_half(rbuf, 0, a0, wave, lane);                           \
        if (!(SC_CL_ABLATE & 16)) mma_half<H16>(a1, ring[J].v[1], act);                        \
        else a1[0][0] = __uint_as_float(ring[J].v[1][0].x ^ ring[J].v[1][5].w ^ act[1][0].x ^ act[2][5].w);  \
        epilogue<H16, EPI>(psum, BIASPREV, RESPREV, N0PREV, OUT, LDO, TPREV, VALIDPREV, wave, lane); \
        if (!(SC_CL_ABLATE & 32)) { put_half(rbuf, 1, a1, wave, lane); __syncthreads(); }      \
        own[0] = a0[0]; own[1] = a1[0];                                                        \
        par ^= 1;                                                                              \
        __builtin_amdgcn_sched_barrier(0);     /* nothing of the next step moves up here (hoisted operand loads cost registers) */ \
    }
// the last step of a phase has nobody behind it
#define SC_CL_TAIL(E, EPI, BIAS, RES, N0, OUT, LDO, TST)                                        \
    {                                                                                          \
        f32x4 psum[2];                                                                         \
        get_sums(red + (par ^ 1) * RED_FLOATS, own, wave, lane, psum);                         \
        epilogue<H16, EPI>(psum, BIAS, RES, N0, OUT, LDO, TST, true, wave, lane);              \
    }

template <bool H16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void clip_layers_cluster_kernel(
    float* x, bf16_t* qkv, bf16_t* att, bf16_t* hbuf, const bf16_t* __restrict__ wc, const float* __restrict__ wf, int layers, int img0, int B,
    int T, float eps, unsigned* sync) {
    extern __shared__ float lds_f[];
    float* red = lds_f;
    char* band = reinterpret_cast<char*>(lds_f);
    bf16_t* Vt = reinterpret_cast<bf16_t*>(band + BAND_BYTES);
    float* bias_l = reinterpret_cast<float*>(Vt + 2 * 64 * VT_LD);      // 384 floats: the running phase's bias slice, chunk-major
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: a scalar for the compiler too
    const int b = blockIdx.x, xcd = b & 7, r8 = b >> 3, c = r8 & 7, grp = r8 >> 3;
    const int slot = grp * 8 + xcd, img = img0 + slot;
    if (img >= B) return;                                  // the whole cluster leaves
    unsigned* cnt = sync + 32 * (1 + slot);                // one 128-byte line per cluster; sync[0] = error word
    unsigned* err = sync;
    unsigned arrivals = 0;
    float* xi = x + (size_t)img * T * CD;
    bf16_t* qkv_i = qkv + (size_t)img * T * 3 * CD;
    bf16_t* att_i = att + (size_t)img * T * CD;
    bf16_t* h_i = hbuf + (size_t)img * T * CMLP;
    const int nh = c + 8 < CHEADS ? 2 : 1, n1 = 6 * nh;
    float4 no_res[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};

    // do the members of this cluster share an XCD?  (every member sets the bit of its XCC id; one agent-scope barrier; one bit = yes)
    if (tid == 0) __hip_atomic_fetch_or(cnt + 1, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrivals += CL;
    cluster_barrier(cnt, arrivals, err, false);
    const bool local = SC_CL_BARRIER != 0 && __popc(__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;

    const int n1p = pad_to_ring(n1);
    Stream st{wc, layers, member_first_chunk(c), n1, n1p, wave, lane0};
    WChunk ring[NR];
#pragma unroll
    for (int j = 0; j < NR - 1; ++j) st.load(ring[j], 0, j);
    uint4 act[4][KS];
    float4 res[3][2];
    f32x4 own[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // this wave's partial of the row block it finishes, kept across the barrier
    int par = 0;                                          // K-partial buffer of the running step (the step behind it used the other)
#define SC_CL_BUBBLE(J, E) st.load(ring[((J) + NR - 1) % NR], l, (E) + NR - 1);
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        const Layer L = layer_at(wf, l);
        // the lane index, opaque once per layer: everything addressed through it is recomputed here instead of being hoisted out of the layer
        // loop as dozens of loop-invariant 64-bit per-lane addresses -- which the register allocator then spilled, and a scratch reload in a
        // step is a vector-memory load behind the weight prefetch (its wait drains the ring)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        SC_CL_STAMP(0)
#define SC_CL_QKV_N0(CI) ((((CI) % 6) >> 1) * CD + ((CI) >= 6 ? c + 8 : c) * 64 + (((CI) % 6) & 1) * 32)
        // ---- ln_1 -> q | k | v of this member's heads -------------------------------------------------------------------------------
        if (tid < n1 * 8) reinterpret_cast<float4*>(bias_l)[tid] = *reinterpret_cast<const float4*>(L.b_qkv + SC_CL_QKV_N0(tid >> 3) + 4 * (tid & 7));
        ln_to_band<H16>(band, xi, T, eps, L.ln1_g, L.ln1_b, wave, lane);
        __syncthreads();
        act_from_band(act, band, wave, lane);
        __syncthreads();                                   // the K-partial buffers alias the band
        SC_CL_STAMP(1)
#pragma unroll 1
        for (int ci = 0; ci < n1p; ci += NR) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int e = ci + j;
                if (e < n1) SC_CL_STEP(j, e, E_QKV, bias_l + 32 * (e - 1), no_res, SC_CL_QKV_N0(e - 1), qkv_i, 3 * CD, e > 0 ? T : 0, true)
                else SC_CL_BUBBLE(j, e)
            }
        }
        SC_CL_TAIL(n1 - 1, E_QKV, bias_l + 32 * (n1 - 1), no_res, SC_CL_QKV_N0(n1 - 1), qkv_i, 3 * CD, T)
#undef SC_CL_QKV_N0
        __syncthreads();                                   // this member's q | k | v rows are in memory for all four waves
        SC_CL_STAMP(2)
        // ---- attention of those heads (waves 0, 1: head c; waves 2, 3: head c + 8) ---------------------------------------------------
        attention_pair<H16>(qkv_i, att_i, T, wave < 2 ? c : c + 8, wave < 2 || nh == 2, Vt + (wave >> 1) * 64 * VT_LD, tid & 127);
        if (tid < 24) reinterpret_cast<float4*>(bias_l)[tid] = *reinterpret_cast<const float4*>(L.b_o + 96 * c + 4 * tid);
        resid_request(res, xi, c, T, wave, lane);
        SC_CL_STAMP(3)
        arrivals += CL;
        cluster_barrier(cnt, arrivals, err, local);
        SC_CL_STAMP(4)
        // ---- x += out-projection: three steps (+ bubbles) ------------------------------------------------------------------------------
        act_from_rows(act, att_i, CD, 0, T, wave, lane);
        {
            const int e0 = n1p;
            static_assert(NR >= 3, "the out-projection is written for a ring of at least three chunks");
            SC_CL_STEP(0, e0, E_RESID, bias_l, no_res, 0, xi, CD, 0, false)
            SC_CL_STEP(1, e0 + 1, E_RESID, bias_l, res[0], 96 * c, xi, CD, T, true)
            SC_CL_STEP(2, e0 + 2, E_RESID, bias_l + 32, res[1], 96 * c + 32, xi, CD, T, true)
#pragma unroll
            for (int j = 3; j < P3; ++j) SC_CL_BUBBLE(j % NR, e0 + j)
            SC_CL_TAIL(e0 + 2, E_RESID, bias_l + 64, res[2], 96 * c + 64, xi, CD, T)
        }
        SC_CL_STAMP(5)
        arrivals += CL;
        cluster_barrier(cnt, arrivals, err, local);
        SC_CL_STAMP(6)
        // ---- ln_2 -> quick_gelu(fc1) -------------------------------------------------------------------------------------------------
        if (tid < 96) reinterpret_cast<float4*>(bias_l)[tid] = *reinterpret_cast<const float4*>(L.b_fc1 + 384 * c + 4 * tid);
        ln_to_band<H16>(band, xi, T, eps, L.ln2_g, L.ln2_b, wave, lane);
        __syncthreads();
        act_from_band(act, band, wave, lane);
        __syncthreads();
        SC_CL_STAMP(7)
#pragma unroll 1
        for (int ci = 0; ci < P12; ci += NR) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int k = ci + j, e = n1p + P3 + k;
                if (P12 == 12 || k < 12) SC_CL_STEP(j, e, E_GELU, bias_l + 32 * (k - 1), no_res, 384 * c + 32 * (k - 1), h_i, CMLP, k > 0 ? T : 0, true)
                else SC_CL_BUBBLE(j, e)
            }
        }
        SC_CL_TAIL(n1p + P3 + 11, E_GELU, bias_l + 32 * 11, no_res, 384 * c + 32 * 11, h_i, CMLP, T)
        SC_CL_STAMP(8)
        arrivals += CL;
        cluster_barrier(cnt, arrivals, err, local);
        SC_CL_STAMP(9)
        // ---- x += fc2: four K-passes of 768 over the hidden rows; the member's three column pairs accumulate in `res` across the passes --
        if (tid < 24) reinterpret_cast<float4*>(bias_l)[tid] = *reinterpret_cast<const float4*>(L.b_fc2 + 96 * c + 4 * tid);
        resid_request(res, xi, c, T, wave, lane);
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            act_from_rows(act, h_i, CMLP, pass * CD, T, wave, lane);
#pragma unroll
            for (int kk = 0; kk < P3; ++kk) {
                const int e = n1p + P3 + P12 + pass * P3 + kk;
                // the step behind: column pair (kk + 2) % 3 (of the pass before for kk = 0); its block leaves after the last pass
                if (kk < 3) SC_CL_STEP(kk % NR, e, E_ACC, bias_l + 32 * ((kk + 2) % 3), res[(kk + 2) % 3], 96 * c + 32 * ((kk + 2) % 3), xi, CD,
                                       pass == 3 && kk > 0 ? T : 0, pass > 0 || kk > 0)
                else SC_CL_BUBBLE(kk % NR, e)
            }
        }
        SC_CL_TAIL(0, E_ACC, bias_l + 64, res[2], 96 * c + 64, xi, CD, T)
        SC_CL_STAMP(10)
        arrivals += CL;
        cluster_barrier(cnt, arrivals, err, local);
        SC_CL_STAMP(11)
    }
#undef SC_CL_BUBBLE
    // a time-out anywhere in the launch: poison this image's class-token row (ln_post reads it), the caller sees NaN
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u && c == 0)
        for (int d = tid; d < CD; d += 256) xi[d] = __builtin_nanf("");
}
#undef SC_CL_STEP
#undef SC_CL_TAIL

static int launch_cluster_pack(const bf16_t* w, bf16_t* out, int layers, hipStream_t st) {
    hipLaunchKernelGGL(cluster_pack_kernel, dim3(4096), dim3(256), 0, st, w, out, layers);
    return (int)hipGetLastError();
}

// grid of one launch: 64 workgroups per 8 images (workgroup b -> XCD b % 8, member (b / 8) % 8, image group b / 64)
template <bool H16>
static int launch_layers_cluster(float* x, bf16_t* qkv, bf16_t* att, bf16_t* hbuf, const bf16_t* wc, const float* wf, int layers, int B, int T,
                                 float eps, unsigned* sync, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)clip_layers_cluster_kernel<H16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int img0 = 0; img0 < B; img0 += 32) {
        const int nb = B - img0 < 32 ? B - img0 : 32;
        if (hipMemsetAsync(sync, 0, 33 * 128, st) != hipSuccess) return (int)hipGetLastError();
        hipLaunchKernelGGL(clip_layers_cluster_kernel<H16>, dim3(64 * ((nb + 7) / 8)), dim3(256), LDS_BYTES, st, x, qkv, att, hbuf, wc, wf, layers,
                           img0, B, T, eps, sync);
    }
    return (int)hipGetLastError();
}

}  // namespace cl
}  // namespace sc
