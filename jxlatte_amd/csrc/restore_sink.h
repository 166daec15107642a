// Output sink shared by the fused restoration kernels (k_restore_fused.hip: LDS tiles, k_restore_stream.hip: register
// streaming): OpsinInverseMatrix.invertXYB (OpsinInverseMatrix.java:105-142) + JXLImage.transferInPlace
// (JXLImage.java:244-258, TransferFunction.java:39-44,83-87) + ImageBuffer.castToInt0 (ImageBuffer.java:129-147) + the
// global store of one pixel.
#pragma once
#include "jxl_internal.h"
#include "jxl_fastpow.h"

namespace jxl {

// Java (int)float
__device__ __forceinline__ int32_t sink_f2i_java(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
// PQ / sRGB through jxl_fastpow.h (JXL_EXACT_POW builds the ocml pow() form for comparison)
#ifdef JXL_EXACT_POW
__device__ __forceinline__ float sink_tf_pq(float f) {
    const double d = pow((double)f, 0.159423828125);
    return (float)pow((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375);
}
__device__ __forceinline__ float sink_tf_srgb(float f) {
    if (f < 0.00313066844250063f) return f * 12.92f;
    return 1.055f * (float)pow((double)f, 0.4166666666666667) + -0.055f;
}
#else
__device__ __forceinline__ float sink_tf_pq(float f) { return fp_tf_pq(f); }
__device__ __forceinline__ float sink_tf_srgb(float f) { return fp_tf_srgb(f); }
#endif

// OpsinInverseMatrix.invertXYB of one pixel
__device__ __forceinline__ void sink_colour(const XybParams& xp, float& v0, float& v1, float& v2) {
    const float gammaL = v1 + v0 + xp.cob[0];
    const float gammaM = v1 - v0 + xp.cob[1];
    const float gammaS = v2 + xp.cob[2];
    const float mixL = (gammaL * gammaL) * gammaL + xp.ob[0];
    const float mixM = (gammaM * gammaM) * gammaM + xp.ob[1];
    const float mixS = (gammaS * gammaS) * gammaS + xp.ob[2];
    v0 = xp.sm[0] * mixL + xp.sm[1] * mixM + xp.sm[2] * mixS;
    v1 = xp.sm[3] * mixL + xp.sm[4] * mixM + xp.sm[5] * mixS;
    v2 = xp.sm[6] * mixL + xp.sm[7] * mixM + xp.sm[8] * mixS;
}

// transfer function + ImageBuffer.castToInt0 of one sample (max_value > 0): the threshold-table forms of jxl_fastpow.h where the
// output format has one (the reference's integer for every input), else the float transfer and the Java cast
__device__ __forceinline__ int32_t sink_quant(const FusedArgs& a, float t) {
#ifndef JXL_EXACT_POW
    if (a.p.transfer == JXL_TRANSFER_PQ && a.p.max_value == 65535 && a.p.pq_tab && a.p.pq16_thr)
        return fp_pq16(t, reinterpret_cast<const float4*>(a.p.pq_tab), a.p.pq16_thr);
    if (a.p.transfer == JXL_TRANSFER_PQ && a.p.max_value == 255 && a.p.pq16_thr) return fp_pq8(t, a.p.pq16_thr + 65537);
    if (a.p.transfer == JXL_TRANSFER_SRGB && a.p.max_value == 65535 && a.p.srgb16_tab)
        return fp_srgb16(t, reinterpret_cast<const float4*>(a.p.srgb16_tab), a.p.srgb16_tab + kSrgb8TableFloats);
    if (a.p.transfer == JXL_TRANSFER_SRGB && a.p.max_value == 255 && a.p.srgb8_tab)
        return fp_srgb8(t, reinterpret_cast<const float4*>(a.p.srgb8_tab));
    if (a.p.transfer == JXL_TRANSFER_PQ && a.p.pq_tab) t = fp_tf_pq_tab(t, reinterpret_cast<const float4*>(a.p.pq_tab));
    else
#endif
    if (a.p.transfer == JXL_TRANSFER_PQ) t = sink_tf_pq(t);
    else if (a.p.transfer == JXL_TRANSFER_SRGB) t = sink_tf_srgb(t);
    const int32_t q = sink_f2i_java(t * (float)a.p.max_value + 0.5f);
    return q < 0 ? 0 : q > a.p.max_value ? a.p.max_value : q;
}

// pixel g (= y * W + x) of the frame; PLAIN: float planes, no transfer function
template <bool PLAIN>
__device__ __forceinline__ void sink_store(const FusedArgs& a, uint32_t g, float v0, float v1, float v2) {
    float v[3] = {v0, v1, v2};
    if (PLAIN) {
#pragma unroll
        for (int c = 0; c < 3; c++) ((float*)a.out[c])[g] = v[c];
        return;
    }
    if (a.p.max_value > 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int32_t q = sink_quant(a, v[c]);
            if (a.p.interleaved) {  // R,G,B per pixel in out[0] (PNGWriter.writeIDAT order)
                if (a.p.out_elem == 2) ((uint16_t*)a.out[0])[3 * g + c] = (uint16_t)q;
                else ((uint8_t*)a.out[0])[3 * g + c] = (uint8_t)q;
            } else if (a.p.out_elem == 2) ((uint16_t*)a.out[c])[g] = (uint16_t)q;
            else if (a.p.out_elem == 1) ((uint8_t*)a.out[c])[g] = (uint8_t)q;
            else ((int32_t*)a.out[c])[g] = q;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float t = v[c];
#ifndef JXL_EXACT_POW
        if (a.p.transfer == JXL_TRANSFER_PQ && a.p.pq_tab) t = fp_tf_pq_tab(t, reinterpret_cast<const float4*>(a.p.pq_tab));
        else
#endif
        if (a.p.transfer == JXL_TRANSFER_PQ) t = sink_tf_pq(t);
        else if (a.p.transfer == JXL_TRANSFER_SRGB) t = sink_tf_srgb(t);
        ((float*)a.out[c])[g] = t;
    }
}

// four consecutive pixels of a row into an interleaved 8-bit buffer (JXL_OUT_RGB8), g even (tile origins are multiples of 62, patch
// columns of 4; the frame width is a multiple of 8): the 12 bytes leave as three dwords when 3 * g is a multiple of 4 bytes, as
// 2 + 4 + 4 + 2 bytes otherwise, instead of twelve byte stores
__device__ __forceinline__ void sink_store_rgb8x4(const FusedArgs& a, uint32_t g, const float o[3][4]) {
    uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t q = (uint32_t)sink_quant(a, o[c][i]);
            const int k = 3 * i + c;
            w[k >> 2] |= q << (8 * (k & 3));
        }
    uint8_t* d8 = (uint8_t*)a.out[0] + 3 * (size_t)g;
    if ((g & 3u) == 0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(d8);
        d[0] = w[0];
        d[1] = w[1];
        d[2] = w[2];
    } else {  // g = 2 (mod 4): the byte offset is 2 (mod 4)
        *reinterpret_cast<uint16_t*>(d8) = (uint16_t)(w[0] & 0xffffu);
        uint32_t* d = reinterpret_cast<uint32_t*>(d8 + 2);
        d[0] = (w[0] >> 16) | (w[1] << 16);
        d[1] = (w[1] >> 16) | (w[2] << 16);
        *reinterpret_cast<uint16_t*>(d8 + 10) = (uint16_t)(w[2] >> 16);
    }
}

}  // namespace jxl
