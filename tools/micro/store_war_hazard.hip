// store_war_hazard.hip -- does a VALU write to the data registers of a buffer_store_dwordx4 issued right before it corrupt the stored data on
// gfx950?  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) inserts a wait state for this write-after-read only when the
// store has NO register in its soffset field; csrc/mlp_tile.hpp's tbl_store uses an SGPR soffset, and sdf_fwd's training instance stored a
// wrong p0 once hipcc scheduled `v_mul_f32 v90, ...` directly behind `buffer_store_dwordx4 v[90:93], ..., s74 offen`.
//   make -C tools/micro store_war_hazard.bin && tools/micro/store_war_hazard.bin     (the %.bin rule; *.bin is git-ignored)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define RUN(NAME, STORE, GAP)                                                                                  \
    __global__ void NAME(float* out, int n_iter) {                                                            \
        const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                \
        float* base = out + (size_t)wave * 64 * 4 * 2;                                                        \
        const unsigned long long b_ = (unsigned long long)base;                                               \
        const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(b_ & 0xffffffffu)),                             \
                          __builtin_amdgcn_readfirstlane((int)((b_ >> 32) & 0xffffu)),                         \
                          __builtin_amdgcn_readfirstlane(64 * 4 * 2 * 4), __builtin_amdgcn_readfirstlane(0x00020000)}; \
        const int voff = lane * 16;                                                                            \
        const int soff = __builtin_amdgcn_readfirstlane(n_iter > 0 ? 1024 : 0);                                \
        const float good = 1.0f, bad = -7.0f;                                                                  \
        for (int it = 0; it < n_iter; ++it) {                                                                  \
            asm volatile("v_mov_b32 v40, %[a]\n v_mov_b32 v41, %[a]\n v_mov_b32 v42, %[a]\n v_mov_b32 v43, %[a]\n s_nop 7\n" \
                         STORE "\n" GAP                                                                        \
                         "v_mov_b32 v40, %[b]\n v_mov_b32 v41, %[b]\n v_mov_b32 v42, %[b]\n v_mov_b32 v43, %[b]\n" \
                         "s_waitcnt vmcnt(0)\n"                                                                \
                         :: [a] "v"(good), [b] "v"(bad), [voff] "v"(voff), [rs] "s"(rs), [soff] "s"(soff)      \
                         : "v40", "v41", "v42", "v43", "memory");                                              \
        }                                                                                                      \
    }

RUN(k_soff_gap0, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], %[soff] offen", "")
RUN(k_soff_gap1, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], %[soff] offen", "s_nop 0\n")
RUN(k_soff_gap2, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], %[soff] offen", "s_nop 1\n")
RUN(k_imm_gap0, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], 0 offen offset:1024", "")
RUN(k_imm_gap1, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], 0 offen offset:1024", "s_nop 0\n")
RUN(k_imm_gap2, "buffer_store_dwordx4 v[40:43], %[voff], %[rs], 0 offen offset:1024", "s_nop 1\n")

// the same question for a 128-bit LDS store (no hazard is listed for it; the convolution kernels stage with ds_write_b128)
#define RUN_LDS(NAME, GAP)                                                                                     \
    __global__ void NAME(float* out, int n_iter) {                                                            \
        __shared__ float4 buf[256];                                                                            \
        const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                \
        const unsigned addr = (unsigned)(size_t)(buf + threadIdx.x);                                           \
        const float good = 1.0f, bad = -7.0f;                                                                  \
        for (int it = 0; it < n_iter; ++it) {                                                                  \
            asm volatile("v_mov_b32 v40, %[a]\n v_mov_b32 v41, %[a]\n v_mov_b32 v42, %[a]\n v_mov_b32 v43, %[a]\n s_nop 7\n" \
                         "ds_write_b128 %[addr], v[40:43]\n" GAP                                              \
                         "v_mov_b32 v40, %[b]\n v_mov_b32 v41, %[b]\n v_mov_b32 v42, %[b]\n v_mov_b32 v43, %[b]\n" \
                         "s_waitcnt lgkmcnt(0)\n"                                                             \
                         :: [a] "v"(good), [b] "v"(bad), [addr] "v"(addr) : "v40", "v41", "v42", "v43", "memory"); \
        }                                                                                                      \
        __syncthreads();                                                                                       \
        const float4 v = buf[threadIdx.x];                                                                     \
        float* dst = out + (size_t)wave * 512 + 256 + lane * 4;                                                \
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;                                                \
    }
RUN_LDS(k_lds_gap0, "")
RUN_LDS(k_lds_gap1, "s_nop 0\n")

template <class K>
static void run(const char* name, K kern) {
    const int blocks = 1024, threads = 256, waves = blocks * threads / 64;
    const size_t n = (size_t)waves * 64 * 4 * 2;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    long long bad = 0, total = 0;
    std::vector<float> h(n);
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(d, 0, n * sizeof(float));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 4);
        hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
        for (int w = 0; w < waves; ++w)
            for (int e = 0; e < 256; ++e) {
                const float v = h[(size_t)w * 512 + 256 + e];      // the store lands 1024 bytes into the wave's region
                ++total;
                if (v != 1.0f) ++bad;
            }
    }
    printf("%-12s: %lld of %lld stored values are not the value the registers held when the store was issued\n", name, bad, total);
    hipFree(d);
}

int main() {
    run("soff, gap 0", k_soff_gap0);
    run("soff, gap 1", k_soff_gap1);
    run("soff, gap 2", k_soff_gap2);
    run("imm,  gap 0", k_imm_gap0);
    run("imm,  gap 1", k_imm_gap1);
    run("imm,  gap 2", k_imm_gap2);
    run("lds,  gap 0", k_lds_gap0);
    run("lds,  gap 1", k_lds_gap1);
    return 0;
}
