// l2_stream: how fast can ONE CU pull L2-resident operand tiles, by transport?  (round 4: csrc/gemm8p.hpp is bound by its 64 KB per K-step
// and CU arriving at ~43 GB/s per CU -- is that the LDS-DMA path or the CU's load path as such?)
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction, lane-linear)          -- what the GEMM does
//   mode 1  global_load_dwordx4 into registers (same addresses, 16 B per lane), values XOR-reduced
//   mode 2  half of the instructions each way
// Every workgroup (8 waves, one per CU) re-reads its own 256 KB window of a buffer that stays in L2 / MALL, `depth` 1-KB pieces per wave in
// flight (counted vmcnt).   hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream.bin && ./l2_stream.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 1) void stream_kernel(const char* __restrict__ buf, size_t window, int iters, unsigned* __restrict__ sink) {
    extern __shared__ uint4 S[];                       // 128 KB of DMA landing area (16 x 1 KB per wave)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const char* base = buf + (size_t)blockIdx.x * window;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)S) + (unsigned)wave * 16384u;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH];
    const unsigned pieces = (unsigned)(window / 8192);            // 1 KB pieces per wave in the window (8 waves interleaved)
    // piece p of this wave: bytes [(p * 8 + wave) * 1024, +1024), lane i -> + 16 i; as GEMM pieces do, 16 rows x 64 B would be the same lines
    for (int it = 0; it < iters; ++it) {
        for (unsigned p0 = 0; p0 < pieces; p0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const unsigned off = (((p0 + d) % pieces) * 8u + (unsigned)wave) * 1024u + (unsigned)lane * 16u;
                const bool dma = MODE == 0 || (MODE == 2 && (d & 1) == 0);
                if (dma) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds0 + (unsigned)(d & 15) * 1024u) : "memory");
                } else {
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[d]) : "v"(off), "s"(base) : "memory");
                }
            }
            // wait for all of this batch (keeps DEPTH pieces per wave in flight at the issue point of the next batch's first piece)
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const bool dma = MODE == 2 && (d & 1) == 0;
                    if (!dma) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[d]) : "n"(DEPTH - 1 - d) : "memory"); acc ^= r[d]; }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = S[lane].x;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, size_t window, int grid, unsigned* sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int iters = 40;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(grid), dim3(512), 131072, 0, buf, window, iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    const double bytes = (double)window * iters * grid;
    printf("%-34s depth %2d  grid %3d: %7.3f ms  %6.2f TB/s chip  %6.1f GB/s per CU (%.1f B/clk/CU at 2.1 GHz)\n", name, DEPTH, grid, best,
           bytes / best / 1e9, bytes / best / 1e6 / grid, bytes / best / 1e6 / grid / 2.1);
}

int main() {
    const size_t window = 256 * 1024;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, window * 256)); CK(hipMemset(buf, 1, window * 256)); CK(hipMalloc(&sink, 4096));
    for (int grid : {256, 128, 32}) {
        if (grid == 256) {
            run<0, 4>("LDS-DMA", buf, window, grid, sink);
            run<0, 8>("LDS-DMA", buf, window, grid, sink);
            run<0, 16>("LDS-DMA", buf, window, grid, sink);
            run<1, 4>("registers", buf, window, grid, sink);
            run<1, 8>("registers", buf, window, grid, sink);
            run<1, 16>("registers", buf, window, grid, sink);
            run<2, 8>("half DMA, half registers", buf, window, grid, sink);
            run<2, 16>("half DMA, half registers", buf, window, grid, sink);
        } else {
            run<0, 8>("LDS-DMA", buf, window, grid, sink);
            run<1, 8>("registers", buf, window, grid, sink);
        }
    }
    return 0;
}
