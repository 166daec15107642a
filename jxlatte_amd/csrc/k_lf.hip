// LF stage on gfx950 (row f1 of the scope table): LF dequantisation + LF chroma-from-luma + adaptive LF smoothing
// of one LF group, written straight into the frame-level LF planes the IDCT stage reads.
//
// Replaces J/frame/vardct/LFCoefficients.java:65-75 (dequant), :78-95 (chroma from luma), :113-180 (adaptiveSmooth).
// lfIndex (:105-111,182-…) feeds the entropy decoder's context model and stays on the host.
//
// One lane per 8x8 cell. The 3x3 neighbourhood is re-dequantised from the integer LF image (9 x 3 int loads, L1/L2
// hits) instead of staging a float image: 1/64 of the frame's pixels, the kernel is a few microseconds.
#include "jxl_internal.h"

namespace jxl {

struct LfArgs {
    const int32_t* q[3];   // lfQuant in X,Y,B order, [H][W] of the LF group
    float* out[3];         // frame-level LF planes
    int H, W;              // LF group size in cells
    int64_t out_off;       // offset of the LF group's (0,0) cell in the frame-level plane
    int out_stride;        // frame-level plane row stride (cells)
    float sd[3];           // scaledDequant[i] / (1 << extraPrecision)
    float sd_gap[3];       // scaledDequant[i]
    float kX, kB;
    int smooth;
};

// dequantised + CfL'd LF sample of channel i at (y, x) (LFCoefficients.java:66-95)
__device__ __forceinline__ void lf_sample(const LfArgs& a, int y, int x, float v[3]) {
    const int k = y * a.W + x;
    const float yv = (float)a.q[1][k] * a.sd[1];
    v[1] = yv;
    v[0] = (float)a.q[0][k] * a.sd[0] + a.kX * yv;
    v[2] = (float)a.q[2][k] * a.sd[2] + a.kB * yv;
}

__global__ __launch_bounds__(256) void k_lf_dequant(const LfArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    float c[3];
    lf_sample(a, y, x, c);
    float o[3] = {c[0], c[1], c[2]};
    if (a.smooth && y > 0 && y + 1 < a.H && x > 0 && x + 1 < a.W) {  // adaptiveSmooth (:113-180); border cells are copied
        float nb[3][3][3];  // [dy][dx][channel]
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) lf_sample(a, y + dy - 1, x + dx - 1, nb[dy][dx]);
        float gap = 0.5f;
        float w[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float sample = nb[1][1][i];
            const float adjacent = nb[1][0][i] + nb[1][2][i] + nb[0][1][i] + nb[2][1][i];
            const float diag = nb[0][0][i] + nb[0][2][i] + nb[2][0][i] + nb[2][2][i];
            w[i] = 0.05226273532324128f * sample + 0.20345139757231578f * adjacent + 0.0334829185968739f * diag;
            const float g = fabsf(sample - w[i]) * a.sd_gap[i];
            if (g > gap) gap = g;
        }
        const float t = 3.0f - 4.0f * gap;
        gap = t > 0.0f ? t : 0.0f;  // Math.max(0f, 3f - 4f * g)
#pragma unroll
        for (int i = 0; i < 3; i++) o[i] = (c[i] - w[i]) * gap + w[i];
    }
    const int64_t d = a.out_off + (int64_t)y * a.out_stride + x;
#pragma unroll
    for (int i = 0; i < 3; i++) a.out[i][d] = o[i];
}

__global__ __launch_bounds__(256) void k_lf_dequant_plain(const int32_t* __restrict__ q, float* __restrict__ out, int H, int W,
                                                          int64_t out_off, int out_stride, float sd) {
    const int n = H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        out[out_off + (int64_t)y * out_stride + x] = (float)q[i] * sd;  // LFCoefficients.java:66-75
    }
}

void launch_lf_dequant_plain(const int32_t* q, float* out, int H, int W, int64_t out_off, int out_stride, float scaled_dequant,
                             int extra_precision, hipStream_t s) {
    if (H <= 0 || W <= 0) return;
    const float sd = scaled_dequant / (float)(1 << extra_precision);
    const int grid = std::min(4096, (H * W + 255) / 256);
    hipLaunchKernelGGL(k_lf_dequant_plain, dim3(grid), dim3(256), 0, s, q, out, H, W, out_off, out_stride, sd);
}

void launch_lf_dequant(const int32_t* const q[3], float* const out[3], int H, int W, int64_t out_off, int out_stride,
                       const float scaled_dequant[3], int extra_precision, float base_corr_x, float base_corr_b,
                       int color_factor, int x_factor_lf, int b_factor_lf, int smooth, hipStream_t s) {
    if (H <= 0 || W <= 0) return;
    LfArgs a;
    for (int i = 0; i < 3; i++) {
        a.q[i] = q[i];
        a.out[i] = out[i];
        a.sd[i] = scaled_dequant[i] / (float)(1 << extra_precision);  // LFCoefficients.java:69
        a.sd_gap[i] = scaled_dequant[i];
    }
    a.H = H;
    a.W = W;
    a.out_off = out_off;
    a.out_stride = out_stride;
    // SPEC: -128, not -127 (LFCoefficients.java:80-82)
    a.kX = base_corr_x + ((float)x_factor_lf - 128.0f) / (float)color_factor;
    a.kB = base_corr_b + ((float)b_factor_lf - 128.0f) / (float)color_factor;
    a.smooth = smooth;
    hipLaunchKernelGGL(k_lf_dequant, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace jxl
