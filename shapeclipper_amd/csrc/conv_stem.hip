// The stem of the ResNet trunks: 7x7 / stride 2 / pad 3 convolution, 3 -> 64 channels, 224 x 224 -> 112 x 112 (torchvision
// ResNet.conv1 behind model/graph.py:50-54 and model/view_estimator.py:40-42), forward and weight gradient (the input is data: there
// is no backward-data), fp32 on v_mfma_f32_32x32x2_f32, NCHW.  MIOpen runs these two at 51 and 38 TFLOP/s (profiles/r02_conv_layers.txt).
//
// Forward: D[co][pixel] = W[co][k] X[k][pixel], k = (ci, ky, kx) = 147 (+ 1 zero), one K pass.  A workgroup keeps the whole filter in LDS
// ([k][96]: the two k of an MFMA on different bank halves) and walks tiles of 256 consecutive output pixels (49 per image exactly, so
// a tile never straddles two images); per tile the 13 x 230 x 3 zero-padded input patch is staged once (prefetched through registers
// during the previous tile's MFMAs) and every operand read is an immediate offset from one per-lane base: lanes are consecutive output
// pixels = stride-2 floats in the patch = 32 distinct banks.  8 waves = 8 column tiles x both channel halves.
//
// Weight gradient: D[co][k] = gy[co][pixel] X[k][pixel], reduction over all B x 112 x 112 pixels: a K-step is two output rows of one
// image; a wave owns one channel half (32 co) x all five 32-wide k tiles (160 >= 147 columns; 80 accumulator registers) for a quarter
// of the K-step's pixels; gy is one ds_read_b128 per 4 MFMA k-steps (the k pair is (pixel, pixel + 4)), the patch values are
// ds_read_b32 with immediate offsets from a per-lane tap base.  256 workgroups x equal ranges of K-steps, one partial [64][160] block
// each, summed in range order by conv_stem_wgrad_reduce_kernel (fixed summation order).
#include <hip/hip_runtime.h>

#include "grid_cus.hpp"
#include "shapeclipper_hip.h"

namespace sc {

typedef float st_f32x16 __attribute__((ext_vector_type(16)));

constexpr int ST_WI = 224, ST_WO = 112, ST_WP = 230, ST_K = 147, ST_KP = 148, ST_CO = 64;
constexpr int ST_HWI = ST_WI * ST_WI, ST_HWO = ST_WO * ST_WO;
// ---- forward
constexpr int SF_PT = 256;                       // output pixels per tile; ST_HWO = 49 * 256
constexpr int SF_ROWS = 13;                      // input rows of a tile's patch: at most 4 output rows -> 2 * 3 + 7
constexpr int SF_CH = SF_ROWS * ST_WP;           // floats per input channel of the patch
constexpr int SF_PATCH = 3 * SF_CH;
constexpr int SF_WLD = 96;                       // filter row stride: k and k + 1 land on different bank halves
constexpr int SF_LDS = ST_KP * SF_WLD + SF_PATCH;
constexpr int SF_NLOAD = (3 * SF_ROWS + 7) / 8;  // (channel, row) lines per wave

__global__ __launch_bounds__(512, 1) void conv_stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ out, int batch) {
    extern __shared__ float4 st_smem[];
    float* Ws = reinterpret_cast<float*>(st_smem);           // [148][96]
    float* Xs = Ws + ST_KP * SF_WLD;                         // [3][13][230]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
    const int tiles = batch * (ST_HWO / SF_PT), G = gridDim.x, g = blockIdx.x;
    const int t_lo = (int)((long long)g * tiles / G), t_hi = (int)((long long)(g + 1) * tiles / G);

    for (int i = tid; i < ST_KP * SF_WLD; i += 512) {       // filter, transposed to [k][co]; row 147 and the pad columns are zero
        const int k = i / SF_WLD, co = i - k * SF_WLD;
        Ws[i] = (k < ST_K && co < ST_CO) ? w[co * ST_K + k] : 0.f;
    }

    // patch staging: wave w copies (channel, row) lines w, w + 8, ...; lanes cover columns lane + 64 c
    float xv[SF_NLOAD][4];
    auto load = [&](int tile) {
        const int b = tile / (ST_HWO / SF_PT), p0 = (tile - b * (ST_HWO / SF_PT)) * SF_PT;
        const int r0 = 2 * (p0 / ST_WO) - 3;                                   // first input row of the patch
        const float* xb = x + (size_t)b * 3 * ST_HWI;
#pragma unroll
        for (int l = 0; l < SF_NLOAD; ++l) {
            const int line = wave + 8 * l, ci = line / SF_ROWS, row = r0 + line - ci * SF_ROWS;
            const bool row_ok = line < 3 * SF_ROWS && row >= 0 && row < ST_WI;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = lane + 64 * c - 3;
                xv[l][c] = (row_ok && col >= 0 && col < ST_WI) ? xb[ci * ST_HWI + row * ST_WI + col] : 0.f;
            }
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int l = 0; l < SF_NLOAD; ++l) {
            const int line = wave + 8 * l;
            if (line < 3 * SF_ROWS) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (lane + 64 * c < ST_WP) Xs[line * ST_WP + lane + 64 * c] = xv[l][c];
            }
        }
    };

    if (t_lo < t_hi) load(t_lo);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        __syncthreads();                                     // filter staged / every wave is done with the previous patch
        store();
        __syncthreads();
        if (tile + 1 < t_hi) load(tile + 1);
        const int b = tile / (ST_HWO / SF_PT), p0 = (tile - b * (ST_HWO / SF_PT)) * SF_PT;
        const int p = p0 + wave * 32 + (lane & 31), y = p / ST_WO, xo = p - y * ST_WO;
        const float* Bb = Xs + (y - p0 / ST_WO) * 2 * ST_WP + 2 * xo;          // + ci * SF_CH + ky * ST_WP + kx
        const float* Ab = Ws + half * SF_WLD + (lane & 31);
        st_f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
        for (int j = 0; j < ST_KP / 2; ++j) {
            // lane half h multiplies k = 2 j + h; k = 147 has a zero filter row and reads the patch value of k = 146
            const int k0 = 2 * j, k1 = 2 * j + 1 < ST_K ? 2 * j + 1 : ST_K - 1;
            const int o0 = (k0 / 49) * SF_CH + ((k0 % 49) / 7) * ST_WP + k0 % 7, o1 = (k1 / 49) * SF_CH + ((k1 % 49) / 7) * ST_WP + k1 % 7;
            const float bv = Bb[o0 + half * (o1 - o0)];
            const float a0 = Ab[2 * j * SF_WLD], a1 = Ab[2 * j * SF_WLD + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
        }
        // acc[r] is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
        float* ob = out + (size_t)b * ST_CO * ST_HWO + p;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(size_t)(32 * i + (r & 3) + 8 * (r >> 2) + 4 * half) * ST_HWO] = acc[i][r];
    }
}

// ---- weight gradient
constexpr int SW_ROWS = 9;                       // input rows of a K-step (two output rows): 2 * 1 + 7
constexpr int SW_CH = SW_ROWS * ST_WP;
constexpr int SW_PATCH = 3 * SW_CH;              // 6210
constexpr int SW_ZERO = 1026;                    // zero floats behind the patch: what the 13 padding columns (k >= 147) read
static_assert((SW_PATCH + SW_ZERO) % 4 == 0, "the gy tile behind it is read with ds_read_b128");
constexpr int SW_GST = 2 * ST_WO + 4;            // gy row stride: 228 = 4 * 57 (odd): conflict-free ds_read_b128 across 32 channels
constexpr int SW_LDS = SW_PATCH + SW_ZERO + ST_CO * SW_GST;
constexpr int SW_PART = 2 * 5 * 16 * 64;         // floats of one partial block: [co half][k tile][acc register][lane]

__global__ __launch_bounds__(512, 1) void conv_stem_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                 float* __restrict__ partial, int batch) {
    extern __shared__ float4 st_smem[];
    float* Xs = reinterpret_cast<float*>(st_smem);           // [3][9][230] + zeros
    float* Gs = Xs + SW_PATCH + SW_ZERO;                     // [64][228]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
    const int cot = wave & 1, kq = wave >> 1;                // channel half, quarter of the K-step's pixel groups
    const int steps = batch * (ST_WO / 2), G = gridDim.x, g = blockIdx.x;
    const int s_lo = (int)((long long)g * steps / G), s_hi = (int)((long long)(g + 1) * steps / G);

    for (int i = tid; i < SW_LDS; i += 512) Xs[i] = 0.f;

    float xv[4][4], gv[8][4];
    auto load = [&](int step) {
        const int b = step / (ST_WO / 2), y0 = 2 * (step - b * (ST_WO / 2));   // output rows y0, y0 + 1
        const float* xb = x + (size_t)b * 3 * ST_HWI;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int line = wave + 8 * l, ci = line / SW_ROWS, row = 2 * y0 - 3 + line - ci * SW_ROWS;
            const bool row_ok = line < 3 * SW_ROWS && row >= 0 && row < ST_WI;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = lane + 64 * c - 3;
                xv[l][c] = (row_ok && col >= 0 && col < ST_WI) ? xb[ci * ST_HWI + row * ST_WI + col] : 0.f;
            }
        }
        const float* gb = gy + ((size_t)b * ST_CO + 8 * wave) * ST_HWO + y0 * ST_WO;      // 224 contiguous floats per channel
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[ch][c] = lane + 64 * c < 2 * ST_WO ? gb[ch * ST_HWO + lane + 64 * c] : 0.f;
    };
    auto store = [&]() {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int line = wave + 8 * l;
            if (line < 3 * SW_ROWS) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (lane + 64 * c < ST_WP) Xs[line * ST_WP + lane + 64 * c] = xv[l][c];
            }
        }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (lane + 64 * c < 2 * ST_WO) Gs[(8 * wave + ch) * SW_GST + lane + 64 * c] = gv[ch][c];
    };

    // per-lane operand bases.  B: lane = filter element k = 32 nt + (lane & 31): patch offset of (ci, ky, kx), or the zero block
    int tapb[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int k = 32 * nt + (lane & 31);
        tapb[nt] = k < ST_K ? (k / 49) * SW_CH + ((k % 49) / 7) * ST_WP + k % 7 + 8 * half : SW_PATCH;
    }
    const float* Ab = Gs + (32 * cot + (lane & 31)) * SW_GST + 4 * half;

    st_f32x16 acc[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    if (s_lo < s_hi) load(s_lo);
    __syncthreads();
    for (int step = s_lo; step < s_hi; ++step) {
        store();
        __syncthreads();
        if (step + 1 < s_hi) load(step + 1);
        // 28 groups of 8 pixels (2 rows x 14); this wave takes groups 7 kq .. 7 kq + 6.  k pair of an MFMA: (pixel, pixel + 4)
#pragma unroll
        for (int gi = 0; gi < 7; ++gi) {
            const int grp = 7 * kq + gi;                                        // wave-uniform
            const int r = grp / 14, x0 = (grp - 14 * r) * 8;
            const float4 a = *reinterpret_cast<const float4*>(Ab + r * ST_WO + x0);
            const int po = r * 2 * ST_WP + 2 * x0;                              // patch offset of (output row r, column x0), scalar
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const float* bp = Xs + tapb[nt] + po;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * s], acc[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // add the four pixel quarters through LDS (one quarter at a time: 2 waves x 20 KB), then write the partial block
    float* R = Gs;
#pragma unroll 1
    for (int q = 1; q < 4; ++q) {
        if (kq == q) {
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[((cot * 5 + nt) * 16 + r) * 64 + lane] = acc[nt][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] += R[((cot * 5 + nt) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (kq == 0) {
        float* dst = partial + (size_t)g * SW_PART;
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((cot * 5 + nt) * 16 + r) * 64 + lane] = acc[nt][r];
    }
}

// dw[co][k] = sum over workgroups (range order) of their partial blocks
__global__ void conv_stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= SW_PART) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int gi = 0;
    for (; gi + 3 < G; gi += 4) {
        s0 += partial[(size_t)gi * SW_PART + e];
        s1 += partial[(size_t)(gi + 1) * SW_PART + e];
        s2 += partial[(size_t)(gi + 2) * SW_PART + e];
        s3 += partial[(size_t)(gi + 3) * SW_PART + e];
    }
    for (; gi < G; ++gi) s0 += partial[(size_t)gi * SW_PART + e];
    const int lane = e & 63, r = (e >> 6) & 15, nt = (e >> 10) % 5, cot = e / (5 * 1024);
    const int co = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), k = 32 * nt + (lane & 31);
    if (k < ST_K) dw[co * ST_K + k] = (s0 + s1) + (s2 + s3);
}

static int stem_cus() { return grid_cus(); }

}  // namespace sc

extern "C" long long sc_conv_stem_wgrad_workspace_floats(void) { return (long long)sc::stem_cus() * sc::SW_PART; }

extern "C" int sc_conv_stem_forward(const float* x, const float* w, float* out, int batch, void* stream) {
    if (batch <= 0) return (int)hipErrorInvalidValue;
    (void)hipFuncSetAttribute((const void*)sc::conv_stem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, sc::SF_LDS * 4);
    hipLaunchKernelGGL(sc::conv_stem_fwd_kernel, dim3(sc::stem_cus()), dim3(512), sc::SF_LDS * 4, (hipStream_t)stream, x, w, out, batch);
    return (int)hipGetLastError();
}

extern "C" int sc_conv_stem_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, void* stream) {
    if (batch <= 0) return (int)hipErrorInvalidValue;
    const int G = sc::stem_cus();
    (void)hipFuncSetAttribute((const void*)sc::conv_stem_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, sc::SW_LDS * 4);
    hipLaunchKernelGGL(sc::conv_stem_wgrad_kernel, dim3(G), dim3(512), sc::SW_LDS * 4, (hipStream_t)stream, gy, x, workspace, batch);
    hipLaunchKernelGGL(sc::conv_stem_wgrad_reduce_kernel, dim3((sc::SW_PART + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, dw, G);
    return (int)hipGetLastError();
}
