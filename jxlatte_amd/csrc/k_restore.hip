// Restoration filters + colour on gfx950: Gaborish, edge-preserving filter, XYB -> linear,
// transfer + integer quantisation.
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/Frame.java:505-542   performGabConvolution      (3x3, clamped edges)
//   J/frame/Frame.java:544-679   performEdgePreservingFilter (mirrored edges, 13/5-tap, 3 iterations)
//   J/color/OpsinInverseMatrix.java:105-142 invertXYB
//   J/JXLImage.java:244-258 + J/color/TransferFunction.java:39-44,83-87 + J/util/ImageBuffer.java:129-147
//
// This file holds the stage-per-kernel forms (one reference function = one kernel, global memory,
// every tap addressed exactly like the reference). They are the general path (any size, any
// iteration count) and the cross-check for the fused tile kernel in k_restore_fused.hip.
// Strict f32: sums in reference order, no FMA contraction, correctly rounded division.
#include "jxl_internal.h"
#include "jxl_fastpow.h"

namespace jxl {

// MathHelper.mirrorCoordinate (MathHelper.java:323-329)
__device__ __forceinline__ int mirror(int c, int size) {
    while (c < 0 || c >= size) {
        const int tc = ~c;
        c = tc >= 0 ? tc : (size << 1) + tc;
    }
    return c;
}

struct GabW {
    float base[3], adj[3], diag[3];
};

__global__ __launch_bounds__(256) void k_gab(const float* i0, const float* i1, const float* i2, float* o0, float* o1, float* o2,
                                             int h, int w, GabW g) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int north = y == 0 ? 0 : y - 1;
    const int south = y + 1 == h ? h - 1 : y + 1;
    const int west = x == 0 ? 0 : x - 1;
    const int east = x + 1 == w ? w - 1 : x + 1;
    const float* in[3] = {i0, i1, i2};
    float* out[3] = {o0, o1, o2};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float* R = in[c] + (int64_t)y * w;
        const float* N = in[c] + (int64_t)north * w;
        const float* S = in[c] + (int64_t)south * w;
        const float adj = R[west] + R[east] + N[x] + S[x];
        const float diag = N[west] + N[east] + S[west] + S[east];
        out[c][(int64_t)y * w + x] = g.base[c] * R[x] + g.adj[c] * adj + g.diag[c] * diag;
    }
}

void launch_gab(const float* const in[3], float* const out[3], int h, int w, const float w1[3], const float w2[3],
                hipStream_t s) {
    GabW g;
    for (int c = 0; c < 3; c++) {  // Frame.java:510-517 (host-side scalar prologue, f32)
        const float mult = 1.0f / (1.0f + 4.0f * (w1[c] + w2[c]));
        g.base[c] = mult;
        g.adj[c] = w1[c] * mult;
        g.diag[c] = w2[c] * mult;
    }
    hipLaunchKernelGGL(k_gab, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, in[0], in[1], in[2], out[0], out[1], out[2], h,
                       w, g);
}

struct SharpLut {
    float v[8];
};

// Frame.java:552-571
__global__ void k_epf_sigma(const int32_t* hf_mul, const int32_t* sharpness, int n, float global_scale_f, SharpLut lut,
                            float* inv_sigma, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sharp = sharpness[i];
    if (sharp < 0 || sharp > 7) {
        *bad = 1;
        inv_sigma[i] = 0.0f;
        return;
    }
    const float sigma = global_scale_f * lut.v[sharp] / (float)hf_mul[i];
    inv_sigma[i] = 1.0f / sigma;
}

void launch_epf_sigma(const int32_t* hf_mul, const int32_t* sharpness, int bh, int bw, float global_scale_f,
                      const float sharp_lut[8], float* inv_sigma, int* bad_flag, hipStream_t s) {
    SharpLut l;
    for (int i = 0; i < 8; i++) l.v[i] = sharp_lut[i];
    const int n = bh * bw;
    hipLaunchKernelGGL(k_epf_sigma, dim3((n + 255) / 256), dim3(256), 0, s, hf_mul, sharpness, n, global_scale_f, l, inv_sigma,
                       bad_flag);
}

__constant__ int8_t kCross[5][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}};  // Frame.java:44-48 (y, x)
__constant__ int8_t kDCross[13][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1},
                                      {1, -1}, {-1, -1}, {0, -2}, {0, 2}, {2, 0}, {-2, 0}};  // :50-55

template <int ITER>
__global__ __launch_bounds__(256) void k_epf(const float* i0, const float* i1, const float* i2, float* o0, float* o1, float* o2,
                                             int h, int w, const float* inv_sigma, EpfParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* in[3] = {i0, i1, i2};
    float* out[3] = {o0, o1, o2};
    const int bw = (w + 7) >> 3;
    const float s = inv_sigma ? inv_sigma[(y >> 3) * bw + (x >> 3)] : p.inv_sigma_modular;
    const int64_t idx = (int64_t)y * w + x;
    if (s != s || s > (1.0f / 0.3f)) {  // Frame.java:608-612
#pragma unroll
        for (int c = 0; c < 3; c++) out[c][idx] = in[c][idx];
        return;
    }
    constexpr int NT = ITER == 0 ? 13 : 5;
    float sumWeights = 0.0f;
    float sumC[3] = {0.0f, 0.0f, 0.0f};
    const int modY = y & 7, modX = x & 7;
    const bool border = modY == 0 || modY == 7 || modX == 0 || modX == 7;
    for (int t = 0; t < NT; t++) {
        const int dy = ITER == 0 ? kDCross[t][0] : kCross[t][0];
        const int dx = ITER == 0 ? kDCross[t][1] : kCross[t][1];
        float dist = 0.0f;
        if (ITER == 2) {  // epfDistance2 (:657-669)
            const int dY = mirror(y + dy, h), dX = mirror(x + dx, w);
#pragma unroll
            for (int c = 0; c < 3; c++)
                dist = dist + fabsf(in[c][idx] - in[c][(int64_t)dY * w + dX]) * p.channel_scale[c];
        } else {  // epfDistance1 (:638-655)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                for (int q = 0; q < 5; q++) {
                    const int pY = mirror(y + kCross[q][0], h), pX = mirror(x + kCross[q][1], w);
                    const int dY = mirror(y + dy + kCross[q][0], h), dX = mirror(x + dx + kCross[q][1], w);
                    dist = dist + fabsf(in[c][(int64_t)pY * w + pX] - in[c][(int64_t)dY * w + dX]) * p.channel_scale[c];
                }
            }
        }
        if (border) dist = dist * p.border_sad_mul;  // epfWeight (:671-679)
        const float v = 1.0f - dist * p.sigma_scale * s;
        const float weight = v < 0.0f ? 0.0f : v;
        sumWeights = sumWeights + weight;
        const int mY = mirror(y + dy, h), mX = mirror(x + dx, w);
#pragma unroll
        for (int c = 0; c < 3; c++) sumC[c] = sumC[c] + in[c][(int64_t)mY * w + mX] * weight;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) out[c][idx] = sumC[c] / sumWeights;
}

void launch_epf_iter(const float* const in[3], float* const out[3], int h, int w, int iter, const float* inv_sigma,
                     const EpfParams& p, hipStream_t s) {
    const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
    if (iter == 0)
        hipLaunchKernelGGL(k_epf<0>, grid, block, 0, s, in[0], in[1], in[2], out[0], out[1], out[2], h, w, inv_sigma, p);
    else if (iter == 1)
        hipLaunchKernelGGL(k_epf<1>, grid, block, 0, s, in[0], in[1], in[2], out[0], out[1], out[2], h, w, inv_sigma, p);
    else
        hipLaunchKernelGGL(k_epf<2>, grid, block, 0, s, in[0], in[1], in[2], out[0], out[1], out[2], h, w, inv_sigma, p);
}

// OpsinInverseMatrix.invertXYB (OpsinInverseMatrix.java:124-139)
__device__ __forceinline__ void xyb_px(const XybParams& p, float& X, float& Y, float& B) {
    const float gammaL = Y + X + p.cob[0];
    const float gammaM = Y - X + p.cob[1];
    const float gammaS = B + p.cob[2];
    const float mixL = (gammaL * gammaL) * gammaL + p.ob[0];
    const float mixM = (gammaM * gammaM) * gammaM + p.ob[1];
    const float mixS = (gammaS * gammaS) * gammaS + p.ob[2];
    X = p.sm[0] * mixL + p.sm[1] * mixM + p.sm[2] * mixS;
    Y = p.sm[3] * mixL + p.sm[4] * mixM + p.sm[5] * mixS;
    B = p.sm[6] * mixL + p.sm[7] * mixM + p.sm[8] * mixS;
}

__global__ __launch_bounds__(256) void k_xyb(float* p0, float* p1, float* p2, int64_t n, XybParams p) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float X = p0[i], Y = p1[i], B = p2[i];
        xyb_px(p, X, Y, B);
        p0[i] = X;
        p1[i] = Y;
        p2[i] = B;
    }
}

void launch_xyb(float* const planes[3], int64_t n, const XybParams& p, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_xyb, dim3(grid), dim3(256), 0, s, planes[0], planes[1], planes[2], n, p);
}

// JXLCodestreamDecoder.java:270-281
__global__ __launch_bounds__(256) void k_ycbcr(float* p0, float* p1, float* p2, int64_t n) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float cb = p0[i];
        const float yh = p1[i] + 0.50196078431372549019f;
        const float cr = p2[i];
        p0[i] = yh + 1.402f * cr;
        p1[i] = yh - 0.34413628620102214650f * cb - 0.71413628620102214650f * cr;
        p2[i] = yh + 1.772f * cb;
    }
}

void launch_ycbcr(float* const planes[3], int64_t n, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_ycbcr, dim3(grid), dim3(256), 0, s, planes[0], planes[1], planes[2], n);
}

// TransferFunction.TF_PQ.fromLinear through the default fromLinearF (TransferFunction.java:83-87,104-106):
// double pow, result cast to float. Java's Math.pow is specified to 1 ulp (double); parity bar <= 1 ulp float.
// PQ / sRGB through jxl_fastpow.h (~110 instead of 463 instructions for the PQ curve, float results identical on all
// sampled inputs; JXL_EXACT_POW builds the ocml pow() form for comparison)
#ifdef JXL_EXACT_POW
__device__ __forceinline__ float tf_pq(float f) {
    const double d = pow((double)f, 0.159423828125);
    return (float)pow((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375);
}
__device__ __forceinline__ float tf_srgb(float f) {
    if (f < 0.00313066844250063f) return f * 12.92f;
    return 1.055f * (float)pow((double)f, 0.4166666666666667) + -0.055f;
}
#else
__device__ __forceinline__ float tf_pq(float f) { return fp_tf_pq(f); }
__device__ __forceinline__ float tf_srgb(float f) { return fp_tf_srgb(f); }
#endif
// Java (int)float: NaN -> 0, saturating
__device__ __forceinline__ int32_t java_f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

__device__ __forceinline__ float apply_transfer(float v, int transfer, const float4* pq_tab) {
#ifndef JXL_EXACT_POW
    if (transfer == JXL_TRANSFER_PQ && pq_tab) return fp_tf_pq_tab(v, pq_tab);
#endif
    if (transfer == JXL_TRANSFER_PQ) return tf_pq(v);
    if (transfer == JXL_TRANSFER_SRGB) return tf_srgb(v);
    return v;
}

// ImageBuffer.castToInt0 (ImageBuffer.java:129-147)
__device__ __forceinline__ int32_t quantise(float v, int max_value) {
    const int32_t q = java_f2i(v * (float)max_value + 0.5f);
    return q < 0 ? 0 : q > max_value ? max_value : q;
}

__global__ __launch_bounds__(256) void k_transfer(const float* in, int64_t n, int transfer, int max_value, void* out,
                                                  int out_elem, int out_pitch, int out_off, const float4* pq_tab, const float4* srgb8_tab,
                                                  const float* pq16_thr, const float* srgb16_tab) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t o = i * out_pitch + out_off;
#ifndef JXL_EXACT_POW
        if (transfer == JXL_TRANSFER_PQ && max_value == 65535 && pq_tab && pq16_thr) {  // PQ + 16-bit quantisation, exact
            const int32_t q = fp_pq16(in[i], pq_tab, pq16_thr);
            if (out_elem == 4) ((int32_t*)out)[o] = q;
            else ((uint16_t*)out)[o] = (uint16_t)q;
            continue;
        }
        if (transfer == JXL_TRANSFER_PQ && max_value == 255 && pq16_thr) {  // PQ + 8-bit quantisation, exact
            const int32_t q = fp_pq8(in[i], pq16_thr + 65537);
            if (out_elem == 4) ((int32_t*)out)[o] = q;
            else if (out_elem == 2) ((uint16_t*)out)[o] = (uint16_t)q;
            else ((uint8_t*)out)[o] = (uint8_t)q;
            continue;
        }
        if (transfer == JXL_TRANSFER_SRGB && max_value == 65535 && srgb16_tab) {  // sRGB + 16-bit quantisation, exact
            const int32_t q = fp_srgb16(in[i], reinterpret_cast<const float4*>(srgb16_tab), srgb16_tab + kSrgb8TableFloats);
            if (out_elem == 4) ((int32_t*)out)[o] = q;
            else ((uint16_t*)out)[o] = (uint16_t)q;
            continue;
        }
        if (transfer == JXL_TRANSFER_SRGB && max_value == 255 && srgb8_tab) {  // transfer + quantisation as one threshold look-up
            const int32_t q = fp_srgb8(in[i], srgb8_tab);
            if (out_elem == 4) ((int32_t*)out)[o] = q;
            else if (out_elem == 2) ((uint16_t*)out)[o] = (uint16_t)q;
            else ((uint8_t*)out)[o] = (uint8_t)q;
            continue;
        }
#endif
        const float v = apply_transfer(in[i], transfer, pq_tab);
        if (max_value > 0) {
            const int32_t q = quantise(v, max_value);
            if (out_elem == 4) ((int32_t*)out)[o] = q;
            else if (out_elem == 2) ((uint16_t*)out)[o] = (uint16_t)q;
            else ((uint8_t*)out)[o] = (uint8_t)q;
        } else {
            ((float*)out)[o] = v;
        }
    }
}

void launch_transfer(const float* in, int64_t n, int transfer, int max_value, void* out, int out_elem, hipStream_t s,
                     int out_pitch, int out_off, const float* pq_tab, const float* srgb8_tab, const float* pq16_thr,
                     const float* srgb16_tab) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_transfer, dim3(grid), dim3(256), 0, s, in, n, transfer, max_value, out, out_elem, out_pitch, out_off,
                       reinterpret_cast<const float4*>(pq_tab), reinterpret_cast<const float4*>(srgb8_tab), pq16_thr, srgb16_tab);
}

}  // namespace jxl
