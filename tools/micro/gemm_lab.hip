// gemm_lab: validates and times the CLIP tower's GEMM kernels on an MI355X without the Python stack (seconds per run instead of minutes).
//   make -C tools/micro gemm_lab.bin && tools/micro/gemm_lab.bin [check]
// Every shape: the round-4 kernel (csrc/gemm8p.hpp) beside the library's dispatcher (sc_gemm_f16: the kernels of clip_vit.hip), both checked
// against a plain one-thread-per-element fp32 reference on the same random fp16 operands (all elements, transpose-detecting: A and W differ).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "gemm8p.hpp"
#include "gemm_rect.hpp"      // tools/micro: an experiment, see its header
extern "C" int sc_gemm_f16(int epi, const uint16_t* A, const uint16_t* Wt, const float* bias, void* out, int M, int N, int K, void* stream);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_f16(uint16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((h & 0xFFFF) / 32768.f - 1.f) * scale;
        p[i] = __builtin_bit_cast(uint16_t, (_Float16)v);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xFFFF) / 32768.f - 1.f) * scale;
    }
}
// reference: out_ref[m][n] (fp32) = epilogue(sum_k A[m][k] W[n][k] + bias[n] (+ resid[m][n]))
__global__ void ref_gemm(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, float* o, int M, int N, int K, int epi) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += (float)__builtin_bit_cast(_Float16, A[(size_t)m * K + k]) * (float)__builtin_bit_cast(_Float16, W[(size_t)n * K + k]);
    s += bias ? bias[n] : 0.f;
    if (epi == 1) s += resid[(size_t)m * N + n];
    if (epi == 2) s = s / (1.f + expf(-1.702f * s));
    o[(size_t)m * N + n] = s;
}
__global__ void cmp(const void* got, const float* ref, size_t n, int is16, float* maxerr, float* maxref) {
    float e = 0.f, r = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = is16 ? (float)__builtin_bit_cast(_Float16, ((const uint16_t*)got)[i]) : ((const float*)got)[i];
        const float d = fabsf(g - ref[i]);
        e = fmaxf(e, d == d ? d : 3.0e38f); r = fmaxf(r, fabsf(ref[i]));
    }
    atomicMax((int*)maxerr, __float_as_int(e)); atomicMax((int*)maxref, __float_as_int(r));
}

struct Shape { const char* name; int M, N, K, epi; int nb1 = 2; int rect = 0; };   // rect: 1 = 128x128 NS4, 2 = 128x256 NS3, 3 = 64x128 NS6, 4 = 64x256 NS4, 5 = 128x128 NS5
int main(int argc, char** argv) {
    const bool check = argc > 1 && !strcmp(argv[1], "check");
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    std::vector<Shape> shapes = {
        {"B256 qkv ", 12800, 2304, 768, 3}, {"B256 proj", 12800, 768, 768, 1}, {"B256 fc1 ", 12800, 3072, 768, 2}, {"B256 fc2 ", 12800, 768, 3072, 1},
        {"L14  qkv ", 8224, 3072, 1024, 3}, {"L14  proj", 8224, 1024, 1024, 1}, {"L14  fc1 ", 8224, 4096, 1024, 2}, {"L14  fc2 ", 8224, 1024, 4096, 1},
        {"B32  qkv ", 1600, 2304, 768, 3},  {"B32  proj", 1600, 768, 768, 1},   {"B32  fc1 ", 1600, 3072, 768, 2},  {"B32  fc2 ", 1600, 768, 3072, 1},
        {"B256 proj 192", 12800, 768, 768, 1, 1}, {"B256 fc2 192 ", 12800, 768, 3072, 1, 1}, {"ragged 192   ", 777, 576, 192, 1, 1}, {"f32 192      ", 3000, 960, 448, 0, 1},
        {"sq 4096  ", 4096, 4096, 4096, 3}, {"sq 8192  ", 8192, 8192, 8192, 3}, {"ragged   ", 777, 512, 192, 1}, {"one tile ", 256, 256, 64, 0},
    };
    const bool stress = argc > 1 && !strcmp(argv[1], "stress");
    if (argc > 1 && !strcmp(argv[1], "rect")) {
        shapes.clear();
        const Shape base[4] = {{"B32 qkv ", 1600, 2304, 768, 3}, {"B32 proj", 1600, 768, 768, 1}, {"B32 fc1 ", 1600, 3072, 768, 2}, {"B32 fc2 ", 1600, 768, 3072, 1}};
        for (const Shape& b : base)
            for (int r = 1; r <= 5; ++r) { Shape t = b; t.rect = r; shapes.push_back(t); }
        shapes.push_back({"ragged r1", 777, 512, 192, 1, 2, 1}); shapes.push_back({"ragged r3", 777, 384, 192, 0, 2, 3}); shapes.push_back({"ragged r2", 300, 512, 64, 2, 2, 2});
    }
    if (stress) shapes = {{"B256 qkv ", 12800, 2304, 768, 3}, {"B256 fc2 ", 12800, 768, 3072, 1}, {"L14  fc1 ", 8224, 4096, 1024, 2}, {"ragged   ", 777, 512, 192, 1},
                          {"f32 out  ", 3000, 1024, 448, 0}, {"one step ", 1500, 2304, 64, 3}, {"two steps", 1500, 768, 128, 1},
                          {"proj 192 ", 12800, 768, 768, 1, 1}, {"fc2 192  ", 12800, 768, 3072, 1, 1}, {"ragged192", 777, 576, 192, 1, 1}, {"f32 192  ", 3000, 960, 64, 0, 1}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* d_err; CK(hipMalloc(&d_err, 8));
    for (const Shape& s : shapes) {
        const size_t nA = (size_t)s.M * s.K, nW = (size_t)s.N * s.K, nO = (size_t)s.M * s.N;
        const bool o16 = s.epi >= 2;
        uint16_t *A, *W; float *bias, *resid, *ref; void *o_new, *o_old;
        CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&W, nW * 2)); CK(hipMalloc(&bias, s.N * 4)); CK(hipMalloc(&resid, nO * 4)); CK(hipMalloc(&ref, nO * 4));
        CK(hipMalloc(&o_new, nO * 4)); CK(hipMalloc(&o_old, nO * 4));
        const float sc = 1.f / sqrtf((float)s.K) * 4.f;
        fill_f16<<<1024, 256, 0, st>>>(A, nA, 17u, 1.f); fill_f16<<<1024, 256, 0, st>>>(W, nW, 99u, sc);
        fill_f32<<<64, 256, 0, st>>>(bias, s.N, 5u, 1.f); fill_f32<<<1024, 256, 0, st>>>(resid, nO, 7u, 2.f);
        auto reset = [&](void* o) { if (s.epi == 1) CK(hipMemcpyAsync(o, resid, nO * 4, hipMemcpyDeviceToDevice, st)); else CK(hipMemsetAsync(o, 0xFF, nO * (o16 ? 2 : 4), st)); };
        auto run_rect = [&](void* o) {
            switch (s.rect) {
                case 1: return sc::gr::launch_gemm_rect<128, 128, 4, true>(s.epi, A, W, bias, o, s.M, s.N, s.K, st);
                case 2: return sc::gr::launch_gemm_rect<128, 256, 3, true>(s.epi, A, W, bias, o, s.M, s.N, s.K, st);
                case 3: return sc::gr::launch_gemm_rect<64, 128, 6, true>(s.epi, A, W, bias, o, s.M, s.N, s.K, st);
                case 4: return sc::gr::launch_gemm_rect<64, 256, 4, true>(s.epi, A, W, bias, o, s.M, s.N, s.K, st);
                default: return sc::gr::launch_gemm_rect<128, 128, 5, true>(s.epi, A, W, bias, o, s.M, s.N, s.K, st);
            }
        };
        auto run_new = [&](void* o) { return s.rect ? run_rect(o) : s.nb1 == 1 ? sc::g8::launch_gemm8p<true, 1>(s.epi, A, W, bias, o, s.M, s.N, s.K, cus, st)
                                                        : sc::g8::launch_gemm8p<true>(s.epi, A, W, bias, o, s.M, s.N, s.K, cus, st); };
        auto run_old = [&](void* o) { return sc_gemm_f16(s.epi, A, W, bias, o, s.M, s.N, s.K, (void*)st); };
        if (stress) {
            // race screen of the hand-scheduled kernel: 40 launches per shape on fresh operands (a second, unrelated GEMM keeps the memory system
            // busy on another stream for half of them), EVERY element of every launch checked against the plain reference
            hipStream_t st2; CK(hipStreamCreate(&st2));
            float worst = 0.f; int bad = 0;
            for (int it = 0; it < 40; ++it) {
                fill_f16<<<1024, 256, 0, st>>>(A, nA, 17u + 977u * it, 1.f); fill_f16<<<1024, 256, 0, st>>>(W, nW, 99u + 31u * it, sc);
                ref_gemm<<<dim3((s.N + 63) / 64, (s.M + 3) / 4), 256, 0, st>>>(A, W, bias, resid, ref, s.M, s.N, s.K, s.epi);
                reset(o_new);
                CK(hipStreamSynchronize(st));
                if (it & 1) for (int k = 0; k < 4; ++k) sc_gemm_f16(3, A, W, nullptr, o_old, s.M > 4096 ? 4096 : s.M, s.N, s.K, (void*)st2);
                if (run_new(o_new)) { printf("launch refused\n"); break; }
                CK(hipMemsetAsync(d_err, 0, 8, st));
                cmp<<<1024, 256, 0, st>>>(o_new, ref, nO, o16, d_err, d_err + 1);
                float h[2]; CK(hipMemcpyAsync(h, d_err, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
                worst = h[0] > worst ? h[0] : worst;
                if (h[0] > (o16 ? 2e-2f : 2e-4f)) ++bad;
            }
            printf("%s M=%5d N=%4d K=%4d epi=%d | 40 launches: worst max|err| %.3e, launches over tolerance %d\n", s.name, s.M, s.N, s.K, s.epi, worst, bad);
            fflush(stdout);
            hipFree(A); hipFree(W); hipFree(bias); hipFree(resid); hipFree(ref); hipFree(o_new); hipFree(o_old);
            continue;
        }
        const bool big = (size_t)s.M * s.N * s.K > (size_t)8192 * 8192 * 4096;
        float err_new = -1.f, err_old = -1.f, mref = 0.f;
#ifdef LAB_NEW_ONLY
        if (false) {
#else
        if (check || !big) {
#endif
            ref_gemm<<<dim3((s.N + 63) / 64, (s.M + 3) / 4), 256, 0, st>>>(A, W, bias, resid, ref, s.M, s.N, s.K, s.epi);
            for (int which = 0; which < 2; ++which) {
                void* o = which ? o_old : o_new;
                reset(o);
                const int rc = which ? run_old(o) : ((!s.rect && (s.N % (s.nb1 == 1 ? 192 : 256))) ? 1 : run_new(o));
                if (rc) { if (which == 0) err_new = -2.f; continue; }
                CK(hipMemsetAsync(d_err, 0, 8, st));
                cmp<<<1024, 256, 0, st>>>(o, ref, nO, o16, d_err, d_err + 1);
                float h[2]; CK(hipMemcpyAsync(h, d_err, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                (which ? err_old : err_new) = h[0]; mref = h[1];
            }
        }
        float ms_new = -1.f, ms_old = -1.f;
        for (int which = 0; which < 2; ++which) {
            if (which == 0 && !s.rect && (s.N % (s.nb1 == 1 ? 192 : 256))) continue;
            if (which == 0 && s.rect && run_new(o_new)) { (void)hipGetLastError(); continue; }
#ifdef LAB_NEW_ONLY
            if (which == 1) continue;
#endif
            void* o = which ? o_old : o_new;
            for (int i = 0; i < 3; ++i) which ? run_old(o) : run_new(o);
            CK(hipStreamSynchronize(st));
            float best = 1e30f, tot = 0.f; const int reps = 5, inner = 10;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < inner; ++i) which ? run_old(o) : run_new(o);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= inner; tot += ms; best = ms < best ? ms : best;
            }
            (which ? ms_old : ms_new) = tot / reps;
        }
        const double fl = 2.0 * s.M * s.N * s.K;
        if (s.rect) printf("rect %d ", s.rect);
        printf("%s M=%5d N=%4d K=%4d epi=%d | new %7.1f us %6.0f TF  err %.2e | old %7.1f us %6.0f TF  err %.2e | ref max %.2f\n", s.name, s.M, s.N, s.K, s.epi,
               ms_new * 1e3, ms_new > 0 ? fl / (ms_new * 1e-3) / 1e12 : 0.0, err_new, ms_old * 1e3, fl / (ms_old * 1e-3) / 1e12, err_old, mref);
        fflush(stdout);
        hipFree(A); hipFree(W); hipFree(bias); hipFree(resid); hipFree(ref); hipFree(o_new); hipFree(o_old);
    }
    return 0;
}
