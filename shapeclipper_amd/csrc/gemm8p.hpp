// gemm8p.hpp -- the large-shape 16-bit GEMM of the CLIP tower (CLIP_anno.py:166 -> encode_image; the four GEMMs of every transformer layer),
// round 4: a hand-scheduled K loop in the form MI355X's guide documents for this chip (cdna_hip_programming.md section 5, "8-phase").
//
//   C[M,N] = A[M,K] W[N,K]^T (+ bias, epilogue)          16-bit operands (bf16 or IEEE fp16), fp32 accumulate
//
// What was wrong with the round-3 kernels (gemm256_kernel / gemm_bf16_persist_kernel in clip_vit.hip: 600-820 TFLOP/s at any size, matrix pipe
// 29-44 % busy): one barrier per K-step behind an `s_waitcnt vmcnt(0)`, every wave issuing its 8 LDS-DMA pieces, then its fragment reads, then
// its MFMAs -- all waves in the same stage at the same time, so operand delivery and matrix work took turns instead of overlapping.
//
// Structure here.  256 x 256 x 64 tiles, 8 waves = 2 (M) x 4 (N), each wave 128 x 64 of the tile = 8 x 4 v_mfma_f32_16x16x32 blocks (128
// accumulator registers).  The two waves that share a SIMD (wave w and w + 4: the two M halves) run ONE barrier apart: a K-tile is four
// phases, a phase is a LOAD segment (fragment ds_reads + 2 LDS-DMA pieces + a counted vmcnt) and an MFMA segment (16 MFMAs under s_setprio 1),
// each closed by a raw s_barrier -- so while one wave of a SIMD runs its 16 MFMAs the other one does its LDS / DMA work, and the matrix pipe
// sees MFMA segments back to back.
//   phase   reads (ds_read_b128)            MFMAs                       stages (one 16 KB part per phase, 2 pieces per wave)
//   1       A0 (8) + B0 (4)                 A0 x B0                     B1 of step g+1
//   2       B1 (4)                          A0 x B1                     A1 of step g+1
//   3       A1 (8)                          A1 x B1                     A0 of step g+2
//   4       --  (B0 stays in registers)     A1 x B0                     B0 of step g+2
// A K-tile (64 KB) is four 16 KB PARTS ordered by when they are consumed: A0 / A1 = the first / second 64 rows of each wave's 128, B0 / B1 = the
// first / second 32 columns of each wave's 64.  LDS holds two K-tiles (128 KB); a part's slot is re-staged as soon as its last reader is through
// (A0, B0: phase 3 / 4 of the same step; B1, A1: phase 1 / 2 of the next), which keeps FOUR parts (64 KB per CU) in flight all the time: every
// part has >= 5 phases (~0.7 us) to land, the only waits are `s_waitcnt vmcnt(8)` -- never 0 -- at the end of load segments 1, 2 and 4.
// RAW: a part is waited for (by every wave, for its own pieces) in the load segment ONE PHASE BEFORE the one that reads it: with the two groups
// a barrier apart that is what puts a barrier between the last wave's wait and the first wave's read.  WAR: argued per slot in DESIGN.md section 4.3.
// The stream of K-tiles is continuous ACROSS output tiles (persistent workgroups, one per CU, XCD-contiguous tile lists): the first parts of the
// next tile are in flight during the epilogue, and the vmcnt arithmetic never changes (past the last tile the stream re-reads valid addresses
// into slots nobody reads).
// LDS image of a part: [row block of 16][k half of 32][16 rows x 64 bytes] = 1 KB sub-tiles, one LDS-DMA wave instruction each (lane-linear),
// bit 5 of the byte offset XORed with bit 9 (rows 8-15 swap their 32-byte halves): the 16 lanes of a ds_read_b128 service group then hit 16
// different 16-byte bank groups (conflict-free); the swizzle is applied to the SOURCE address of the DMA and to the read address.
// MFMA operands are passed (W fragment, activation fragment): D[i][j] with i = output column, j = output row, so a lane holds FOUR CONSECUTIVE
// COLUMNS of one output row -- the epilogue stores 16 bytes per lane straight from the accumulators (16-bit outputs: two lanes 32 apart
// pair their packed halves first, see the fragment read addresses below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sc {
namespace g8 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef uint16_t h16_t;
typedef __attribute__((address_space(3))) void* lptr_t;

enum { EPI_F32 = 0, EPI_RESID = 1, EPI_GELU_BF16 = 2, EPI_BF16 = 3 };

// Ablation switches of tools/micro/gemm_lab (never set in the product build): 1 = no LDS-DMA pieces, 2 = no MFMAs, 4 = no epilogue stores
#ifndef SC_G8_ABLATE
#define SC_G8_ABLATE 0
#endif

template <bool H>
__device__ __forceinline__ f32x4 mfma32(const uint4& a, const uint4& b, const f32x4& c) {
    if (SC_G8_ABLATE & 2) { f32x4 r = c; r[0] += __uint_as_float(a.x ^ b.x); return r; }      // keeps the fragment reads alive
    if (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool H>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if (H) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(uint32_t, v);
    }
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}

// One LDS-DMA wave instruction (see clip_vit.hip glds16): lane i's 16 bytes land at LDS byte `lds_dst` (wave-uniform) + 16 i.  Issued from
// asm so that hipcc neither counts it nor drains it; completion is counted by hand (SC_G8_VMCNT).
__device__ __forceinline__ void glds(const void* sbase, unsigned voff, unsigned lds_dst) {      // sbase: wave-uniform (SGPR pair)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// lane id without keeping a register live across the K loop (2 VALU operations where it is needed: epilogue, tile set-up)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

#define SC_G8_BARRIER()                           \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define SC_G8_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) > 63 ? 63 : (n)) : "memory")
#define SC_G8_LGKM0()                                         \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

// Register loads the compiler must not count either (its own `s_waitcnt vmcnt(N)` for them would be computed without the LDS-DMA pieces in the
// queue and, for the last of them, be a vmcnt(0): a drain of the whole stream).  Form (ii) of the guide's section 5.7: "=v" load, later a wait
// statement that names every destination "+v" so that no consumer can be scheduled above it.
#define SC_G8_LD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
template <int N>
__device__ __forceinline__ void wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait3(f32x4& a, f32x4& b, f32x4& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait8(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f, f32x4& g, f32x4& h) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N) : "memory");
}

constexpr int PART = 16384, KTILE = 65536, LDS_BYTES = 2 * KTILE;       // bytes: one part, one K-tile (A0 | A1 | B0 | B1), the two buffers

// Persistent grid: gridDim.x workgroups (a multiple of 8, <= one per CU), workgroup b on XCD b % 8 walks the tiles
// t_lo(xcd) + (b >> 3) + i * (gridDim.x >> 3) of its XCD's contiguous range of the row-major (m tile, n tile) list.
// Requires K % 64 == 0, N % 256 == 0 (host checks); rows past M are clamped on load and masked on store.
// NB1 = 16-column blocks in a wave's B1 half: 2 -> 256-column tiles (the layout described above); 1 -> 192-column tiles (a wave owns 128 x 48:
// B1 is an 8 KB part of ONE piece per wave, 7 pieces per wave and K-step instead of 8) for N = 768: 12,800 x 768 is 150 tiles of 256 x 256 on 256
// CUs but 200 of 256 x 192.  fp32 outputs only (the paired 16-bit stores want an even number of column blocks).
template <int EPI, bool H16, int NB1 = 2>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8p_kernel(
    const h16_t* __restrict__ A, const h16_t* __restrict__ Wt, const float* __restrict__ bias, void* __restrict__ out, int M, int N, int K,
    int ntn, int tiles) {
    static_assert(NB1 == 2 || (NB1 == 1 && (EPI == EPI_F32 || EPI == EPI_RESID)), "192-column tiles: fp32 epilogues only");
    constexpr int WCOLS = 32 + 16 * NB1, BN = 4 * WCOLS, NBLK = 2 + NB1;      // columns per wave / per tile, 16-column blocks per wave
    constexpr int NPC = 6 + NB1;                                              // LDS-DMA pieces per wave and K-step = pieces in flight at every wait
    extern __shared__ uint4 Sbuf[];                 // 128 KB: [2 K-tiles][A0 | A1 | B0 | B1] x 16 KB
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    const int xcd = blockIdx.x & 7, wloc = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int tq = tiles >> 3, trem = tiles & 7;
    const int t_lo = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
    const int t_hi = t_lo + tq + (xcd < trem ? 1 : 0);
    const int kt1 = K >> 6;
    // Tile order: panels of 4 row tiles, column-major inside a panel, so that the ~32 consecutive tiles an XCD works on at a time form a
    // 4 x 8 block (4 A bands + 8 W bands per K-step in its L2: 5.3x reuse) instead of 1 x 32 (1.9x: at 8192^3 the K loop was memory-side bound,
    // tools/micro/l2_stream).  The CLIP shapes (9-16 column tiles) are indifferent to it.
    const int ntm = tiles / ntn;
    auto tile_mn = [&](int tile, int& bm, int& bn) {
        const int per = 4 * ntn, panel = tile / per, within = tile - panel * per;
        const int rows = min(4, ntm - 4 * panel);
        const int nt = within / rows;
        bm = (4 * panel + (within - nt * rows)) * 256;
        bn = nt * BN;
    };
    const int n_my = t_lo + wloc < t_hi ? (t_hi - t_lo - wloc + wpx - 1) / wpx : 0;
    if (n_my == 0 || kt1 == 0) return;

    // ---- staging state: the part rows this lane fetches (wave w stages row block w of every part: 16 rows x two k halves) ----
    // 32-bit byte offsets from the (wave-uniform) operand bases: the host checks that A and W are smaller than 4 GB
    unsigned sA0, sA1, sB0, sB1;                                         // this lane's source rows of the stage step's tile, at k = 0
    int s_i = 0, s_kt = 0;                                               // stage step = (s_i-th tile of this workgroup, K-tile s_kt)
    auto stage_tile = [&](int i) {
        const int ln = lane_id();
        const int pr = 16 * wave + (ln >> 2);                            // part row 0..127
        const int kch = (ln & 3) ^ (((ln >> 5) & 1) << 1);               // logical 16-byte chunk whose data belongs at this lane's LDS position
        const int ra0 = pr + 64 * (pr >> 6);                             // tile row of part row pr in A0 (A1: + 64)
        const int rb0 = WCOLS * (pr >> 5) + (pr & 31);                   // tile column in B0 (256-column tiles: B1 = + 32)
        const int tile = t_lo + wloc + min(i, n_my - 1) * wpx;           // past the end: the last tile again (valid addresses, unread slots)
        int bm, bn;
        tile_mn(tile, bm, bn);
        const unsigned rowb = (unsigned)K * 2u;
        sA0 = (unsigned)min(bm + ra0, M - 1) * rowb + kch * 16;
        sA1 = (unsigned)min(bm + ra0 + 64, M - 1) * rowb + kch * 16;
        sB0 = (unsigned)(bn + rb0) * rowb + kch * 16;
        if (NB1 == 2) sB1 = (unsigned)(bn + rb0 + 32) * rowb + kch * 16;
        else sB1 = (unsigned)(bn + WCOLS * (wave >> 1) + 32 + (ln >> 2)) * rowb + (wave & 1) * 64 + kch * 16;   // ONE piece: row block wave >> 1, k half wave & 1
    };
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)Sbuf);
    const unsigned stage_w = lds0 + (unsigned)wave * 2048u;
    // part P (0 A0, 1 A1, 2 B0, 3 B1) of the CURRENT stage step: two 1 KB pieces (k halves) into row block `wave` of the part's slot
    // part P (0 A0, 1 A1, 2 B0, 3 B1) of the CURRENT stage step: two 1 KB pieces (k halves) into row block `wave` of the part's slot
    auto issue = [&](unsigned src, int P) {
        const unsigned dst = stage_w + (unsigned)((s_i * kt1 + s_kt) & 1) * KTILE + (unsigned)P * PART;
        const unsigned v = src + (unsigned)s_kt * 128u;
        const void* base = P < 2 ? (const void*)A : (const void*)Wt;
        if (!(SC_G8_ABLATE & 1)) {
            if (NB1 == 1 && P == 3) {
                glds(base, v, dst - (unsigned)wave * 2048u + (unsigned)(wave >> 1) * 2048u + (unsigned)(wave & 1) * 1024u);
            } else {
                glds(base, v, dst);
                glds(base, v + 64u, dst + 1024u);
            }
        }
    };
    auto advance = [&]() {
        if (++s_kt == kt1) { s_kt = 0; ++s_i; stage_tile(s_i); }
    };

    // ---- fragment read addresses (bytes from the K-tile buffer's base) ----
    // D layout: lane (fr = lane & 15, fc = lane >> 4) holds output row fr, columns 4 fc' .. + 3 of each 16 x 16 block.
    // W fragments are read with the rows of a 16-row block permuted: fragment row i is block row pi(i) = i with bits 2 and 3 swapped, so that
    // lane group fc (= i >> 2) of the D layout holds block COLUMNS 4 * {0, 2, 1, 3}[fc] .. + 3 -- what lets two lanes 32 apart pair their
    // packed halves into 8 consecutive 16-bit columns (v_permlane32_swap) and store 16 bytes per lane.  Same 16 rows per instruction, same banks.
    unsigned offA, offB, offB1;                                          // + part * PART + i * 2048 + kh * 1024
    {
        const int ln = lane_id(), fr = ln & 15, fc = ln >> 4;           // MFMA fragment row / 16-byte k chunk
        const int frp = (fr & 3) | ((fr & 4) << 1) | ((fr & 8) >> 1);
        offA = (unsigned)(4 * wr) * 2048u + (unsigned)(fr * 64 + ((fc ^ ((fr >> 3) << 1)) * 16));
        offB = 2u * PART + (unsigned)(2 * wc) * 2048u + (unsigned)(frp * 64 + ((fc ^ ((frp >> 3) << 1)) * 16));
        offB1 = 3u * PART + (unsigned)(NB1 * wc) * 2048u + (unsigned)(frp * 64 + ((fc ^ ((frp >> 3) << 1)) * 16));
    }
    const char* S = reinterpret_cast<const char*>(Sbuf);
    auto rd = [&](unsigned byte) -> uint4 { return *reinterpret_cast<const uint4*>(S + byte); };

    uint4 af[4][2], b0f[2][2], b1f[NB1][2];
    f32x4 acc[8][NBLK];
    f32x4 bv[NBLK];

    // ---- per-tile register loads and the epilogue ---------------------------------------------------------------------------------------
    // Loads hipcc must not count (asm; waited for by hand with counted vmcnt), issued right after the PREVIOUS tile's stores (before the
    // prologue for the first tile) so that their round trips run under the K loop:
    //   bias      4 x 16 bytes per lane (this lane's 4 columns of each of the wave's 4 column blocks), pinned at the epilogue;
    //   residual  EPI_RESID (x += A W^T + b on the fp32 residual stream): the accumulators of a tile START as the residual values -- 32 loads
    //             of 16 bytes per lane in the order the phases need them; the first K-step's MFMA segments wait for their quadrant
    //             (26 / 20 / 14 / 8 = the younger operations in the queue at that point).  Rows past M are clamped.
    // Stores go through a buffer resource whose range ends at row M: rows past M are dropped by the address check while the instruction is
    // issued and counted all the same, so the number of operations the epilogue puts into the queue is a compile-time constant (NE) and the
    // three counted waits of the step behind a tile boundary are raised by it: nothing waits for the stores (vmcnt retires in issue order).
    constexpr bool OUT32 = EPI == EPI_F32 || EPI == EPI_RESID;
    constexpr unsigned ESZ = OUT32 ? 4u : 2u;
    constexpr int NE = (OUT32 ? 8 * NBLK : 4 * NBLK) + (EPI == EPI_RESID ? 8 * NBLK : 0);   // stores (+ residual loads); the 4 bias loads are not counted (stricter)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)M * N * ESZ), 0x00020000);
    auto tile_init = [&](int i) {
        const int tile = t_lo + wloc + i * wpx;
        int bm, bn;
        tile_mn(tile, bm, bn);
        const int ln = lane_id(), fr = ln & 15, fc = ln >> 4;
        const int cq = 4 * (((fc & 1) << 1) | (fc >> 1));               // this lane's first column inside a 16-column block
        const int col0 = bn + WCOLS * wc + cq;
        if (bias) {
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) { const float* bp = bias + col0 + 16 * nb; SC_G8_LD4(bv[nb], bp); }
        }
        if (EPI == EPI_RESID) {
            const float* base = reinterpret_cast<const float*>(out) + col0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                // quadrants in phase order: (A0,B0) (A0,B1) (A1,B1) (A1,B0)
                const int mh = q >> 1, nh = (q == 1 || q == 2) ? 1 : 0;
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int mb = 4 * mh + i4;
                    const float* rp = base + (size_t)min(bm + 128 * wr + 16 * mb + fr, M - 1) * N;
#pragma unroll
                    for (int j = 0; j < (nh ? NB1 : 2); ++j) { const float* p = rp + 16 * (2 * nh + j); SC_G8_LD4(acc[mb][2 * nh + j], p); }
                }
            }
        }
    };
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) bv[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_init(0);

    // ---- prologue: step 0 complete + A0, B0 of step 1 (the steady-state issue order) ----
    stage_tile(0);
    issue(sA0, 0); issue(sB0, 2); issue(sB1, 3); issue(sA1, 1);
    advance();
    issue(sA0, 0); issue(sB0, 2);
    SC_G8_VM(NPC);                                  // A0, B0 of step 0 have landed (this wave's pieces)
    SC_G8_BARRIER();
    if (wr == 1) SC_G8_BARRIER();                   // the second M half runs one barrier behind the first

    int c_i = 0, c_kt = 0;                          // compute step
    const int nsteps = n_my * kt1;
    for (int g = 0; g < nsteps; ++g) {
        const unsigned buf = (unsigned)(g & 1) * KTILE;
        const bool first = EPI != EPI_RESID && c_kt == 0;     // EPI_RESID: the accumulators were initialised with the residual tile
        const bool rfirst = EPI == EPI_RESID && c_kt == 0;
        const bool p1 = c_kt == 0 && c_i > 0;                 // the step behind a tile boundary: NE epilogue operations are in the queue
        // ---------------- phase 1 ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[i][kh] = rd(buf + offA + i * 2048 + kh * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) b0f[j][kh] = rd(buf + offB + j * 2048 + kh * 1024);
        issue(sB1, 3);
        if (p1) SC_G8_VM(NPC + NE); else SC_G8_VM(NPC);
        SC_G8_BARRIER();
        SC_G8_LGKM0();
        if (rfirst) wait8<9 * NB1 + 8>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
        __builtin_amdgcn_s_setprio(1);
        if (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<H16>(b0f[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<H16>(b0f[j][0], af[i][0], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<H16>(b0f[j][1], af[i][1], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        SC_G8_BARRIER();
        // ---------------- phase 2 ----------------
#pragma unroll
        for (int j = 0; j < NB1; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) b1f[j][kh] = rd(buf + offB1 + j * 2048 + kh * 1024);
        issue(sA1, 1);
        if (p1) SC_G8_VM(NPC + NE); else SC_G8_VM(NPC);
        advance();
        SC_G8_BARRIER();
        SC_G8_LGKM0();
        if (rfirst) {
            if constexpr (NB1 == 2) wait8<5 * NB1 + 10>(acc[0][2], acc[0][3], acc[1][2], acc[1][3], acc[2][2], acc[2][3], acc[3][2], acc[3][3]);
            else wait4<5 * NB1 + 10>(acc[0][2], acc[1][2], acc[2][2], acc[3][2]);
        }
        __builtin_amdgcn_s_setprio(1);
        if (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB1; ++j) acc[i][2 + j] = mfma32<H16>(b1f[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB1; ++j) acc[i][2 + j] = mfma32<H16>(b1f[j][0], af[i][0], acc[i][2 + j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB1; ++j) acc[i][2 + j] = mfma32<H16>(b1f[j][1], af[i][1], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        SC_G8_BARRIER();
        // ---------------- phase 3 ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[i][kh] = rd(buf + offA + PART + i * 2048 + kh * 1024);
        issue(sA0, 0);
        SC_G8_BARRIER();
        SC_G8_LGKM0();
        if (rfirst) {
            if constexpr (NB1 == 2) wait8<12 + NB1>(acc[4][2], acc[4][3], acc[5][2], acc[5][3], acc[6][2], acc[6][3], acc[7][2], acc[7][3]);
            else wait4<12 + NB1>(acc[4][2], acc[5][2], acc[6][2], acc[7][2]);
        }
        __builtin_amdgcn_s_setprio(1);
        if (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB1; ++j) acc[4 + i][2 + j] = mfma32<H16>(b1f[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB1; ++j) acc[4 + i][2 + j] = mfma32<H16>(b1f[j][0], af[i][0], acc[4 + i][2 + j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB1; ++j) acc[4 + i][2 + j] = mfma32<H16>(b1f[j][1], af[i][1], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        SC_G8_BARRIER();
        // ---------------- phase 4 ----------------
        issue(sB0, 2);
        if (p1) SC_G8_VM(NPC + NE); else SC_G8_VM(NPC);
        SC_G8_BARRIER();
        if (rfirst) wait8<NPC>(acc[4][0], acc[4][1], acc[5][0], acc[5][1], acc[6][0], acc[6][1], acc[7][0], acc[7][1]);
        __builtin_amdgcn_s_setprio(1);
        if (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma32<H16>(b0f[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma32<H16>(b0f[j][0], af[i][0], acc[4 + i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma32<H16>(b0f[j][1], af[i][1], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (++c_kt == kt1) {
            // ---- epilogue of tile c_i: straight from the accumulators (a lane holds 4 consecutive columns of one row per block) ----
            const int tile = t_lo + wloc + c_i * wpx;
            int bm, bn;
            tile_mn(tile, bm, bn);
            const int ln = lane_id(), fr = ln & 15, fc = ln >> 4;
            const int cq = 4 * (((fc & 1) << 1) | (fc >> 1));
            if constexpr (NB1 == 2) wait4<NPC>(bv[0], bv[1], bv[2], bv[3]);  // requested a whole tile ago; NPC = the stream pieces in flight
            else wait3<NPC>(bv[0], bv[1], bv[2]);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const unsigned row = (unsigned)(bm + 128 * wr + 16 * mb + fr);
                const unsigned rowoff = row * (unsigned)N + (unsigned)(bn + WCOLS * wc);
                if constexpr (OUT32) {
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) {
                        const f32x4 v = acc[mb][nb] + bv[nb];
                        if (SC_G8_ABLATE & 4) { if (v[0] == 1.2345f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1]), orsrc, rowoff * 4u, 0, 0); continue; }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, (rowoff + 16u * nb + (unsigned)cq) * 4u, 0, 0);
                    }
                } else if constexpr (NB1 == 2) {
#pragma unroll
                    for (int np = 0; np < 2; ++np) {                     // column blocks 2 np, 2 np + 1: one 16-byte store per lane
                        u32x2 pk[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4 v = acc[mb][2 * np + j] + bv[2 * np + j];
                            if (EPI == EPI_GELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v[r]));
                            }
                            pk[j] = u32x2{pack2<H16>(v[0], v[1]), pack2<H16>(v[2], v[3])};
                        }
                        // lanes 0-31 (fc 0, 1) keep their block-2np half and take the partner's (lane + 32: fc + 2) block-2np half = 8 consecutive
                        // columns of block 2 np; lanes 32-63 take the partner's block-(2np+1) half in front of their own
                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 q = {s0[0], s1[0], s0[1], s1[1]};
                        if (SC_G8_ABLATE & 4) { if (q[0] == 0x12345u) __builtin_amdgcn_raw_buffer_store_b32(q[1], orsrc, rowoff * 2u, 0, 0); continue; }
                        // columns: block 2 np + (fc >> 1), offset 8 (fc & 1)
                        __builtin_amdgcn_raw_buffer_store_b128(q, orsrc, (rowoff + 32u * np + 16u * (unsigned)(fc >> 1) + 8u * (unsigned)(fc & 1)) * 2u, 0, 0);
                    }
                }
            }
            c_kt = 0;
            ++c_i;
            if (c_i < n_my) tile_init(c_i);
        }
        SC_G8_BARRIER();
    }
    if (wr == 0) SC_G8_BARRIER();                   // matches the second half's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing stream pieces land before the LDS allocation is released
}

#undef SC_G8_BARRIER
#undef SC_G8_VM
#undef SC_G8_LGKM0

// Host side.  `cus`: compute units of the device (the grid is one workgroup per CU, rounded down to a multiple of 8).
template <bool H16, int NB1 = 2>
static inline int launch_gemm8p(int epi, const h16_t* A, const h16_t* Wt, const float* bias, void* out, int M, int N, int K, int cus,
                                hipStream_t st) {
    // the buffer resource addresses the output with 32 bits, the LDS-DMA the operands with 32-bit offsets
    if ((unsigned long long)(M + 256) * K * 2ull >= (1ull << 32) || (unsigned long long)N * K * 2ull >= (1ull << 32)) return (int)hipErrorInvalidValue;
    constexpr int BN = 4 * (32 + 16 * NB1);
    if ((K % 64) || (N % BN) || M <= 0 || (unsigned long long)(M + 256) * N * 4ull >= (1ull << 32)) return (int)hipErrorInvalidValue;
    const int ntn = N / BN, ntm = (M + 255) / 256, tiles = ntn * ntm;
    int grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    if (grid > ((tiles + 7) / 8) * 8) grid = ((tiles + 7) / 8) * 8;
#define SC_G8_LAUNCH(E)                                                                                                         \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<E, H16, NB1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);   \
        hipLaunchKernelGGL((gemm8p_kernel<E, H16, NB1>), dim3(grid), dim3(512), LDS_BYTES, st, A, Wt, bias, out, M, N, K, ntn, tiles); \
    } while (0)
    if constexpr (NB1 == 2) {
        switch (epi) {
            case EPI_F32: SC_G8_LAUNCH(EPI_F32); break;
            case EPI_RESID: SC_G8_LAUNCH(EPI_RESID); break;
            case EPI_GELU_BF16: SC_G8_LAUNCH(EPI_GELU_BF16); break;
            default: SC_G8_LAUNCH(EPI_BF16); break;
        }
    } else {
        switch (epi) {
            case EPI_F32: SC_G8_LAUNCH(EPI_F32); break;
            case EPI_RESID: SC_G8_LAUNCH(EPI_RESID); break;
            default: return (int)hipErrorInvalidValue;
        }
    }
#undef SC_G8_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace g8
}  // namespace sc
