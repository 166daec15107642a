// Output sink of the fused restoration kernels (k_restore_fused*.hip): OpsinInverseMatrix.invertXYB (OpsinInverseMatrix.java:105-142) + JXLImage.transferInPlace
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

// ---- sink kinds: what the fused restoration kernels are instantiated for (r4). SK_GENERIC picks transfer function and output
// format per sample at run time (uniform compares and branches around every store: SALU 4.3x the float-plane variant, r3
// counters); the others fix both at compile time -- the formats the PNG writer and the u16 HDR path ask for
// (PNGWriter.java:65,105-111: tf = hdr ? PQ : sRGB, 8 / 16 bit; ImageBuffer.java:129-147).
enum SinkKind {
    SK_PLAIN = 0,       // float planes, no transfer function
    SK_GENERIC = 1,     // anything (run-time)
    SK_PQ_U16 = 2,      // TF_PQ -> castToIntWithMax(65535), planar u16 (JXL_OUT_U16; BASELINE config C4)
    SK_PQ_RGB16 = 3,    // the same, R,G,B interleaved (JXL_OUT_RGB16: the HDR PNG)
    SK_SRGB_RGB8 = 4,   // TF_SRGB -> 255, interleaved bytes (JXL_OUT_RGB8: the 8-bit PNG)
    SK_SRGB_RGB16 = 5,  // TF_SRGB -> 65535, interleaved u16
    SK_COUNT = 6
};
// the kind a frame's parameters select (host side): a specialised kind only when its tables are there
inline int sink_kind_of(const RestoreParams& p) {
    if (p.transfer == JXL_TRANSFER_NONE && p.max_value == 0) return SK_PLAIN;
#ifndef JXL_EXACT_POW
    if (p.transfer == JXL_TRANSFER_PQ && p.max_value == 65535 && p.out_elem == 2 && p.pq_tab && p.pq16_thr) return p.interleaved ? SK_PQ_RGB16 : SK_PQ_U16;
    if (p.transfer == JXL_TRANSFER_SRGB && p.max_value == 255 && p.out_elem == 1 && p.interleaved && p.srgb8_tab) return SK_SRGB_RGB8;
    if (p.transfer == JXL_TRANSFER_SRGB && p.max_value == 65535 && p.out_elem == 2 && p.interleaved && p.srgb16_tab) return SK_SRGB_RGB16;
#endif
    return SK_GENERIC;
}

// transfer function + ImageBuffer.castToInt0 of one sample (max_value > 0): the threshold-table forms of jxl_fastpow.h where the
// output format has one (the oracle's integer for every input), else the float transfer and the Java cast
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
// the same for a sink kind known at compile time
template <int SK>
__device__ __forceinline__ int32_t sink_quant_k(const FusedArgs& a, float t) {
    if constexpr (SK == SK_PQ_U16 || SK == SK_PQ_RGB16) return fp_pq16(t, reinterpret_cast<const float4*>(a.p.pq_tab), a.p.pq16_thr);
    else if constexpr (SK == SK_SRGB_RGB8) return fp_srgb8(t, reinterpret_cast<const float4*>(a.p.srgb8_tab));
    else if constexpr (SK == SK_SRGB_RGB16)
        return fp_srgb16(t, reinterpret_cast<const float4*>(a.p.srgb16_tab), a.p.srgb16_tab + kSrgb8TableFloats);
    else return sink_quant(a, t);
}

// pixel g (= y * W + x) of the frame
template <int SK>
__device__ __forceinline__ void sink_store_k(const FusedArgs& a, uint32_t g, float v0, float v1, float v2) {
    float v[3] = {v0, v1, v2};
    if constexpr (SK == SK_PLAIN) {
#pragma unroll
        for (int c = 0; c < 3; c++) ((float*)a.out[c])[g] = v[c];
        return;
    } else if constexpr (SK == SK_PQ_U16) {
#pragma unroll
        for (int c = 0; c < 3; c++) ((uint16_t*)a.out[c])[g] = (uint16_t)sink_quant_k<SK>(a, v[c]);
        return;
    } else if constexpr (SK == SK_PQ_RGB16 || SK == SK_SRGB_RGB16) {
#pragma unroll
        for (int c = 0; c < 3; c++) ((uint16_t*)a.out[0])[3 * g + c] = (uint16_t)sink_quant_k<SK>(a, v[c]);
        return;
    } else if constexpr (SK == SK_SRGB_RGB8) {
#pragma unroll
        for (int c = 0; c < 3; c++) ((uint8_t*)a.out[0])[3 * g + c] = (uint8_t)sink_quant_k<SK>(a, v[c]);
        return;
    } else {
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
}
// (the register-streaming kernel's two forms)
template <bool PLAIN>
__device__ __forceinline__ void sink_store(const FusedArgs& a, uint32_t g, float v0, float v1, float v2) {
    sink_store_k<PLAIN ? SK_PLAIN : SK_GENERIC>(a, g, v0, v1, v2);
}

struct __attribute__((packed, aligned(8))) sink_f4a8 {
    float x, y, z, w;
};
struct __attribute__((packed, aligned(4))) sink_u2a4 {
    uint32_t x, y;
};
struct __attribute__((packed, aligned(4))) sink_u4a4 {
    uint32_t x, y, z, w;
};
// pixels g .. g+3 of a row, g even, as the kind's wide stores; false: the kind has none (the caller stores pixel by pixel)
// NT (r6): the planes are the frame's final output -- read next by a copy to the host or a collective, not by a kernel of the path --:
// non-temporal stores (kernel 94.8-99.8 -> 93.6 us, batch +0.6...2.4 %). Not for the first half of a two-launch EPF x 3 run, whose
// planes the second launch reads at once.
template <int SK, bool NT = false>
__device__ __forceinline__ bool sink_store4_k(const FusedArgs& a, uint32_t g, const float o[3][4]) {
    if constexpr (SK == SK_PLAIN) {
        // one 16-byte store per lane and channel: a wave instruction then covers whole rows of 256 contiguous bytes instead of
        // every other 8 bytes (g even: 8-byte aligned, so the store is declared 8-byte aligned; gfx950 global stores do not need
        // 16-byte alignment)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if constexpr (NT) {
                typedef float sink_v4 __attribute__((ext_vector_type(4), aligned(8)));
                __builtin_nontemporal_store(sink_v4{o[c][0], o[c][1], o[c][2], o[c][3]}, reinterpret_cast<sink_v4*>((float*)a.out[c] + g));
            } else {
                *reinterpret_cast<sink_f4a8*>((float*)a.out[c] + g) = sink_f4a8{o[c][0], o[c][1], o[c][2], o[c][3]};
            }
        }
        return true;
    } else if constexpr (SK == SK_PQ_U16) {
        // four u16 of a plane = 8 bytes at a 4-byte-aligned address
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = (uint32_t)sink_quant_k<SK>(a, o[c][i]);
            *reinterpret_cast<sink_u2a4*>((uint16_t*)a.out[c] + g) = sink_u2a4{q[0] | q[1] << 16, q[2] | q[3] << 16};
        }
        return true;
    } else if constexpr (SK == SK_PQ_RGB16 || SK == SK_SRGB_RGB16) {
        // twelve u16 = 24 contiguous bytes at 6 g: a multiple of 12, 4-byte aligned
        uint32_t q[12];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) q[3 * i + c] = (uint32_t)sink_quant_k<SK>(a, o[c][i]);
        uint16_t* d = (uint16_t*)a.out[0] + 3 * (size_t)g;
        *reinterpret_cast<sink_u4a4*>(d) = sink_u4a4{q[0] | q[1] << 16, q[2] | q[3] << 16, q[4] | q[5] << 16, q[6] | q[7] << 16};
        *reinterpret_cast<sink_u2a4*>(d + 8) = sink_u2a4{q[8] | q[9] << 16, q[10] | q[11] << 16};
        return true;
    } else {
        return false;  // RGB8: twelve byte stores measured faster than three packed dwords (r3, JXL_RGB8_PACKED); generic: per pixel
    }
}

}  // namespace jxl
