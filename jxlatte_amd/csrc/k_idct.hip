// VarDCT stage 1 on gfx950: dequantisation + chroma-from-luma + LLF insertion + inverse
// transform of every varblock, written straight into the frame planes.
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/vardct/HFCoefficients.java:140-319  bakeDequantizedCoeffs (dequant, CfL, finalizeLLF)
//   J/frame/group/PassGroup.java:83-331         invertVarDCT and the special 8x8 transforms
//   J/util/MathHelper.java:68-145               inverse/forward DCT (direct O(N^2) form, cosine LUT)
//
// Bit-exactness rules: every sum keeps the reference's order
// (dest[k] = src[0]; dest[k] += src[n] * lut[n-1][k], n ascending), multiplies and adds are
// separate IEEE f32 operations (-ffp-contract=off), no MFMA (fused, different order).
//
// Three kernel families, all type-uniform per workgroup so there is no divergence:
//   special : Hornuss, DCT2, DCT4, DCT4x8, DCT8x4, AFV0-3. One LANE per (varblock, channel), the whole
//             8x8 block lives in VGPRs, every LUT / AFV-basis factor is an instruction literal.
//   wave    : DCT8, DCT16 and the 16x8 rectangles. One WAVE = 64/max(H,W) blocks, their three channels one after
//             the other (luma first: its dequantised column stays in registers for chroma-from-luma):
//             lane = one column, coefficient rows streamed from HBM in order (dequantised on the fly),
//             accumulators in VGPRs, LUT rows wave-uniform (scalar loads); transpose through a
//             wave-private LDS image; lane = one row for the row pass.
//   wg      : the 32- and 64-point families. One 256-thread workgroup per 64/min(H,W) blocks of one
//             channel: dequantised coefficients staged in LDS, every 1-D transform split over the
//             waves (8 or 16 outputs per lane) so that no lane runs a 2x63x64-instruction chain.
//   large   : 128/256-edge blocks do not fit LDS: dequant -> column pass -> row pass through a
//             scratch plane, 64x64 register tiles per wave.
// Launches: one per register class (k_idct_multi<CLASS>: every type of the class in one grid) + the special kernel;
// k_idct_multi_batch / k_idct_special_batch are the same bodies over a batch of frames (blockIdx.y = frame).
#include "jxl_internal.h"
#include "../../include/jxl_tables.h"
#include <cstdlib>

#ifndef JXL_WG_PATH_MIN
#define JXL_WG_PATH_MIN 32
#endif

namespace jxl {

#include "idct_small.h"

typedef float v2f __attribute__((ext_vector_type(2)));
// Read-only tables (the cosine LUT) are addressed through the constant address space: loads from it are invariant by
// definition, so a wave-uniform address always becomes a scalar load (s_load_dwordx8 feeding v_pk_mul straight from
// SGPRs). With plain global pointers the compiler has to prove that no store can clobber the table; in the merged
// kernels that proof runs out of budget after the first body and the LUT rows arrive as per-lane vector loads instead
// (measured: 64x64 body 173 -> 330 us).
typedef const __attribute__((address_space(4))) float* cfloatp;
typedef const __attribute__((address_space(4))) v2f* cv2fp;
// Ordering of a wave's own LDS traffic. The wave-level paths transpose through a wave-PRIVATE LDS image: the lanes that read
// it belong to the wave that wrote it, and the LDS unit executes one wave's instructions in issue order, so no workgroup
// barrier is needed -- only that the compiler keeps the writes before the reads. A wavefront-scope fence does exactly that
// and emits no s_barrier: the four waves of a workgroup stop waiting for each other six times per work item.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ __forceinline__ cfloatp as_const(const float* p) { return (cfloatp)p; }
// one block record (16 bytes, 16-byte aligned) through the constant address space: s_load_dwordx4 when the index is
// wave-uniform, global_load_dwordx4 when it is per lane
__device__ __forceinline__ DevBlock load_block(const DevBlock* blocks, int i) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = ((const __attribute__((address_space(4))) v4i*)blocks)[i];
    DevBlock b;
    b.cy = (uint16_t)((uint32_t)r.x & 0xffffu);
    b.cx = (uint16_t)((uint32_t)r.x >> 16);
    b.type = (uint32_t)r.y;
    b.cfl_zero = (uint32_t)r.z;
    b.hf_mul = r.w;
    return b;
}

// Diagnostic build only (-DJXL_STAMPS, tools/idct_stamps.py): lane 0 of every workgroup writes s_memtime at the phase
// boundaries of its work item to a buffer of its own; no product build contains a stamp.
#ifdef JXL_STAMPS
__device__ unsigned long long* g_stamps = nullptr;  // [workgroup][8]
#define JXL_STAMP(i)                                                                                              \
    do {                                                                                                          \
        if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define JXL_STAMP_VAL(i, v)                                                                       \
    do {                                                                                          \
        if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)(v); \
    } while (0)
#else
#define JXL_STAMP(i)
#define JXL_STAMP_VAL(i, v)
#endif

__constant__ float kLlfScale[32] = JXL_LLF_SCALE_INIT;

// all L outputs of one 1-D IDCT held by a lane, as packed pairs: lo[j] = outputs (2j, 2j+1), hi[j] = outputs
// (L-1-2j, L-2-2j) -- the mirror images of lo[j], so the same packed product updates both (see idct1d_reg)
template <int L>
struct MirrorAcc {
    v2f lo[L / 4], hi[L / 4];
    __device__ __forceinline__ void init(float s0) {
#pragma unroll
        for (int j = 0; j < L / 4; j++) lo[j] = hi[j] = v2f{s0, s0};
    }
    // lut_lo: the first L/2 entries of table row n-1 (or the lane's slice of them)
    template <bool ODD>
    __device__ __forceinline__ void step(float s, cfloatp lut_lo) {
        const v2f s2 = {s, s};
        const cv2fp lr = (cv2fp)lut_lo;
#pragma unroll
        for (int j = 0; j < L / 4; j++) {
            const v2f p = s2 * lr[j];
            lo[j] = lo[j] + p;
            hi[j] = ODD ? hi[j] - p : hi[j] + p;
        }
    }
    // output i of the L this lane holds: i < L/2 counts up from the lane's first low output, i >= L/2 continues through
    // the mirrored ones so that get(L-1-i) is the mirror of get(i)
    __device__ __forceinline__ float get(int i) const {
        if (i < L / 2) return (i & 1) ? lo[i / 2].y : lo[i / 2].x;
        const int m = L - 1 - i;
        return (m & 1) ? hi[m / 2].y : hi[m / 2].x;
    }
};

// HFCoefficients.dequantizeHFCoefficients inner expression (HFCoefficients.java:309-315)
__device__ __forceinline__ float dequant1(int32_t q, float qb, float qbn, float sfc, float w) {
    const float quant = (q > -2 && q < 2) ? (q == 0 ? 0.0f : (q > 0 ? qb : -qb)) : (float)q - qbn / (float)q;
    return quant * sfc * w;
}

// Same value as dequant1, but qbn/(float)q comes from a 64-entry table of qbn/(float)|q| (built once per wave with
// the same correctly rounded division; x/(-y) == -(x/y) exactly in IEEE arithmetic), so the ~12-instruction
// division sequence runs only for |q| >= 64.
__device__ __forceinline__ float dequant1_tab(int32_t q, float qb, float qbn, float sfc, float w, const float* __restrict__ tab) {
    const int aq = q < 0 ? -q : q;
    float t;
    if (aq < 64) t = tab[aq];
    else t = qbn / (float)aq;
    const float quant = aq < 2 ? (q == 0 ? 0.0f : (q > 0 ? qb : -qb)) : (float)q - (q > 0 ? t : -t);
    return quant * sfc * w;
}

// CfL factors of one tile for one block (HFCoefficients.java:177-181) honouring the reference's
// per-group cache order (DevBlock::cfl_zero)
__device__ __forceinline__ void cfl_factors(const DevFrame& f, int ty, int tx, bool zero, float& kX, float& kB) {
    if (zero) {
        kX = 0.0f;
        kB = 0.0f;
    } else {
        kX = f.kx_tab[ty * f.tw + tx];
        kB = f.kb_tab[ty * f.tw + tx];
    }
}

// ---- small: one lane = one 8x8 varblock -------------------------------------------------------------
// dequantised coefficient row `y` of channel c of an 8x8-footprint block: co[x], x = 0..7, with
// chroma-from-luma applied (the luma row is dequantised again for each chroma channel instead of being
// kept in 64 registers) and the LLF sample inserted at (0,0).
template <int PI, bool FLIP>
__device__ __forceinline__ void small_row(const DevFrame& f, int c, int y, int64_t base, float hfm, float kc, float lf_c,
                                          float co[8]) {
    const float qbn = f.quant_bias_numerator;
    auto load8 = [&](int ch, int q[8]) {
        const int4 a = *reinterpret_cast<const int4*>(f.coeff[ch] + base + y * 8);  // (cell-tiled plane: coeff_off)
        const int4 b = *reinterpret_cast<const int4*>(f.coeff[ch] + base + y * 8 + 4);
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    };
    int q[8];
    load8(c, q);
    const float* w = f.weights + f.woffs[PI * 3 + c];
    const float sfc = f.scale_factor[c] / hfm;
    const float qb = f.quant_bias[c];
#pragma unroll
    for (int x = 0; x < 8; x++) co[x] = dequant1(q[x], qb, qbn, sfc, w[FLIP ? x * 8 + y : y * 8 + x]);
    if (c != 1 && !f.no_cfl) {
        int qy[8];
        load8(1, qy);
        const float* wy = f.weights + f.woffs[PI * 3 + 1];
        const float sfy = f.scale_factor[1] / hfm;
        const float qby = f.quant_bias[1];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            float dy = dequant1(qy[x], qby, qbn, sfy, wy[FLIP ? x * 8 + y : y * 8 + x]);
            if (y == 0 && x == 0) dy = 0.0f;  // the LLF corner is skipped by the dequantiser (:305-306)
            co[x] = co[x] + kc * dy;          // chromaFromLuma (:186-188)
        }
    }
    // finalizeLLF for a 1x1 dctSelect: forwardDCT2D of one sample is sample * (1f/1) twice and llfScale is
    // 1f*1f: multiplications by exactly 1.0f, i.e. the LF sample itself
    if (y == 0) co[0] = lf_c;
}

// the other nine 8x8-footprint types: whole block of one channel in registers
template <int TYPE>
__device__ __forceinline__ void special_block(const DevFrame& f, const DevBlock b, const int c, float* __restrict__ o0,
                                              float* __restrict__ o1, float* __restrict__ o2) {
    constexpr int PI = TYPE == 1 ? 1 : TYPE == 2 ? 2 : TYPE == 3 ? 3 : (TYPE == 12 || TYPE == 13) ? 9 : 10;
    const int W = f.width;
    const int py0 = b.cy * 8, px0 = b.cx * 8;
    const int64_t base = (int64_t)py0 * W + px0;
    const float hfm = (float)b.hf_mul;
    float kX, kB;
    cfl_factors(f, py0 >> 6, px0 >> 6, b.cfl_zero & 1u, kX, kB);
    {
        const float kc = c == 0 ? kX : kB;
        const float lf_c = f.lf[c][b.cy * f.bw + b.cx];
        float co[64], px[64];
#pragma unroll
        for (int y = 0; y < 8; y++) small_row<PI, false>(f, c, y, coeff_off(W, py0, px0), hfm, kc, lf_c, co + y * 8);
        invert_small<TYPE>(co, px);
        float* o = (c == 0 ? o0 : c == 1 ? o1 : o2) + base;
#pragma unroll
        for (int y = 0; y < 8; y++) {
            *reinterpret_cast<float4*>(o + (int64_t)y * W) = make_float4(px[y * 8], px[y * 8 + 1], px[y * 8 + 2], px[y * 8 + 3]);
            *reinterpret_cast<float4*>(o + (int64_t)y * W + 4) =
                make_float4(px[y * 8 + 4], px[y * 8 + 5], px[y * 8 + 6], px[y * 8 + 7]);
        }
    }
}

// ---- shared pieces of medium / large ---------------------------------------------------------------
__device__ __forceinline__ int ceil_log2_dev(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}

// dequant + CfL of one sample position of a block (all three channels).
// (y, x) are relative to the block; returns dq[3]. LLF corner samples return 0 (filled later).
struct BlockCtx {
    int py0, px0, H, W;       // footprint
    int dsh, dsw;             // dctSelect size
    int ty0, tx0;             // first tile
    uint32_t cfl_zero;
    float sfc[3];
    const float* w[3];
    int mw;                   // weight matrix row stride (matrixWidth)
    bool flip;
};

__device__ __forceinline__ void make_block_ctx(const DevFrame& f, const DevBlock b, int H, int W, int PI, BlockCtx& k) {
    k.py0 = b.cy * 8;
    k.px0 = b.cx * 8;
    k.H = H;
    k.W = W;
    k.dsh = H >> 3;
    k.dsw = W >> 3;
    k.ty0 = k.py0 >> 6;
    k.tx0 = k.px0 >> 6;
    k.cfl_zero = b.cfl_zero;
    const float hfm = (float)b.hf_mul;
    for (int c = 0; c < 3; c++) {
        k.sfc[c] = f.scale_factor[c] / hfm;
        k.w[c] = f.weights + f.woffs[PI * 3 + c];
    }
    k.mw = H < W ? W : H;
    k.flip = H >= W;  // METHOD_DCT: tall or square
}

__device__ __forceinline__ void dequant_sample(const DevFrame& f, const BlockCtx& k, int y, int x, float dq[3]) {
    if (y < k.dsh && x < k.dsw) {
        dq[0] = dq[1] = dq[2] = 0.0f;
        return;
    }
    const int64_t off = coeff_off(f.width, k.py0 + y, k.px0 + x);
    const int wy = k.flip ? x : y, wx = k.flip ? y : x;
    const int wi = wy * k.mw + wx;
    const float dy = dequant1(f.coeff[1][off], f.quant_bias[1], f.quant_bias_numerator, k.sfc[1], k.w[1][wi]);
    const float dx = dequant1(f.coeff[0][off], f.quant_bias[0], f.quant_bias_numerator, k.sfc[0], k.w[0][wi]);
    const float db = dequant1(f.coeff[2][off], f.quant_bias[2], f.quant_bias_numerator, k.sfc[2], k.w[2][wi]);
    const int ty = (k.py0 + y) >> 6, tx = (k.px0 + x) >> 6;
    const int bit = (ty - k.ty0) * 5 + (tx - k.tx0);
    float kX, kB;
    cfl_factors(f, ty, tx, (k.cfl_zero >> bit) & 1u, kX, kB);
    dq[0] = dx + kX * dy;
    dq[1] = dy;
    dq[2] = db + kB * dy;
}

// one channel of dequant_sample (luma is dequantised again for the chroma channels)
__device__ __forceinline__ float dequant_sample_c(const DevFrame& f, const BlockCtx& k, int c, int y, int x) {
    if (y < k.dsh && x < k.dsw) return 0.0f;
    const int64_t off = coeff_off(f.width, k.py0 + y, k.px0 + x);
    const int wy = k.flip ? x : y, wx = k.flip ? y : x;
    const int wi = wy * k.mw + wx;
    const float dy = dequant1(f.coeff[1][off], f.quant_bias[1], f.quant_bias_numerator, k.sfc[1], k.w[1][wi]);
    if (c == 1) return dy;
    const float dc = dequant1(f.coeff[c][off], f.quant_bias[c], f.quant_bias_numerator, k.sfc[c], k.w[c][wi]);
    const int ty = (k.py0 + y) >> 6, tx = (k.px0 + x) >> 6;
    const int bit = (ty - k.ty0) * 5 + (tx - k.tx0);
    float kX, kB;
    cfl_factors(f, ty, tx, (k.cfl_zero >> bit) & 1u, kX, kB);
    return dc + (c == 0 ? kX : kB) * dy;
}

// finalizeLLF (HFCoefficients.java:194-229) for one block and channel, executed by `nthr` threads
// (thread ids tid = 0..nthr-1) of a workgroup. s0/s1: LDS scratch of dsh*dsw floats each. Writes
// the dsh x dsw corner through put(k_row, k_col, value). Contains __syncthreads: call uniformly.
template <typename Put>
__device__ __forceinline__ void llf_block(const DevFrame& f, int cy, int cx, int c, int dsh, int dsw, float* s0, float* s1,
                                          int tid, int nthr, Put put) {
    const int lw = ceil_log2_dev(dsw), lh = ceil_log2_dev(dsh);
    const float* lutw = f.lut + lut_off(lw);
    const float* luth = f.lut + lut_off(lh);
    const float* src = f.lf[c] + (int64_t)cy * f.bw + cx;
    // rows: forwardDCTHorizontal of length dsw on each of the dsh rows -> s0[y][k]
    for (int i = tid; i < dsh * dsw; i += nthr) {
        const int y = i / dsw, k = i % dsw;
        const float inv = 1.0f / (float)dsw;
        const float* r = src + (int64_t)y * f.bw;
        float d2;
        if (k == 0) {
            d2 = r[0];
            for (int x = 1; x < dsw; ++x) d2 = d2 + r[x];
        } else {
            const float* lut = lutw + (k - 1) * dsw;
            d2 = r[0] * lut[0];
            for (int n = 1; n < dsw; ++n) d2 = d2 + r[n] * lut[n];
        }
        s0[y * dsw + k] = d2 * inv;
    }
    __syncthreads();
    // transpose (s1[x][y] = s0[y][x]) folded into indexing; columns: length dsh on each of the dsw rows of s1
    for (int i = tid; i < dsh * dsw; i += nthr) {
        const int x = i / dsh, k = i % dsh;
        const float inv = 1.0f / (float)dsh;
        float d2;
        if (k == 0) {
            d2 = s0[0 * dsw + x];
            for (int y = 1; y < dsh; ++y) d2 = d2 + s0[y * dsw + x];
        } else {
            const float* lut = luth + (k - 1) * dsh;
            d2 = s0[0 * dsw + x] * lut[0];
            for (int n = 1; n < dsh; ++n) d2 = d2 + s0[n * dsw + x] * lut[n];
        }
        s1[x * dsh + k] = d2 * inv;
    }
    __syncthreads();
    // transposeMatrixInto(scratch0, dest, ...): dest[k][x] = s1[x][k]; then *= llfScale[k][x]
    const int yll = ceil_log2_dev(dsh), xll = ceil_log2_dev(dsw);
    for (int i = tid; i < dsh * dsw; i += nthr) {
        const int k = i / dsw, x = i % dsw;
        const float sc = kLlfScale[k << (5 - yll)] * kLlfScale[x << (5 - xll)];
        put(k, x, s1[x * dsh + k] * sc);
    }
    __syncthreads();
}

// One LLF coefficient (ky, kx) of a DSH x DSW LF patch: forwardDCT2D (MathHelper.java:124-136) row pass for
// column kx of every row, then the column pass for ky, times llfScale (HFCoefficients.java:194-229). Each lane
// recomputes the row-pass values it needs; the values are deterministic, so this equals the reference's
// shared scratch arrays bit for bit.
template <int DSH, int DSW>
__device__ __forceinline__ float llf_coeff(const DevFrame& f, const float* __restrict__ lfp /* patch origin, stride f.bw */, int ky,
                                           int kx) {
    const cfloatp lutw = as_const(f.lut + lut_off(ceil_log2_dev(DSW)));
    const cfloatp luth = as_const(f.lut + lut_off(ceil_log2_dev(DSH)));
    const float invw = 1.0f / (float)DSW, invh = 1.0f / (float)DSH;
    float r[DSH];
#pragma unroll
    for (int y = 0; y < DSH; y++) {
        const float* row = lfp + (int64_t)y * f.bw;
        float d2;
        if (kx == 0) {
            d2 = row[0];
#pragma unroll
            for (int x = 1; x < DSW; ++x) d2 = d2 + row[x];
        } else {
            const cfloatp lut = lutw + (kx - 1) * DSW;
            d2 = row[0] * lut[0];
#pragma unroll
            for (int n = 1; n < DSW; ++n) d2 = d2 + row[n] * lut[n];
        }
        r[y] = d2 * invw;
    }
    float d2;
    if (ky == 0) {
        d2 = r[0];
#pragma unroll
        for (int y = 1; y < DSH; ++y) d2 = d2 + r[y];
    } else {
        const cfloatp lut = luth + (ky - 1) * DSH;
        d2 = r[0] * lut[0];
#pragma unroll
        for (int n = 1; n < DSH; ++n) d2 = d2 + r[n] * lut[n];
    }
    constexpr int yll = DSH <= 1 ? 0 : DSH <= 2 ? 1 : DSH <= 4 ? 2 : 3;
    constexpr int xll = DSW <= 1 ? 0 : DSW <= 2 ? 1 : DSW <= 4 ? 2 : 3;
    return (d2 * invh) * (kLlfScale[ky << (5 - yll)] * kLlfScale[kx << (5 - xll)]);
}

// ---- medium: DCT16..DCT64 and rectangles: one WAVE = BPW blocks of ONE channel, no workgroup barriers ----
// Column pass: lane = one column of a block; the coefficient rows are streamed from global memory in row
// order n = 0,1,.. (dequantised on the fly), which is exactly the order of the reference's sum, so the H
// partial sums of the column live in VGPRs and every LUT row is wave-uniform (scalar loads). The column
// results are transposed through a wave-private LDS image (stride W+1, conflict-free both ways); row pass:
// lane = one row, W partial sums in VGPRs, the finished row leaves as 16-byte stores.
template <int H, int W>
struct MediumCfg {
    static constexpr int MAXD = H > W ? H : W;
    static constexpr int BPW = 64 / MAXD;  // blocks per wave
    static constexpr int LD = W + 1;       // padded row stride
    static constexpr int IMG = H * LD;     // floats per block image
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(4 * BPW * IMG + 4 * 64);  // 4 waves, each with its qbn/|q| table
};

template <int H, int W, int TYPE>
__device__ __forceinline__ void medium_item(const DevFrame& f, const DevBlock* __restrict__ blocks, const WorkItem it,
                                            float* __restrict__ lds_wg, float* __restrict__ o0, float* __restrict__ o1,
                                            float* __restrict__ o2) {
    using Cfg = MediumCfg<H, W>;
    constexpr int LD = Cfg::LD, IMG = Cfg::IMG;
    constexpr int PI = JXL_TT[TYPE].param_index;
    constexpr bool FLIP = H >= W;  // TransformType.flip() for METHOD_DCT
    constexpr int DSH = H / 8, DSW = W / 8;
    // rows prefetched per chunk of the column pass: 4 for the 16-row types keeps them within 64 VGPRs (8 waves per SIMD
    // and a seat in the merged class-0 launch), 8 for the 8-row ones
    constexpr int MAXRC = 8;
    // the 4 waves of the workgroup are independent: wave w owns blocks [w*BPW, (w+1)*BPW) of the item
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wfirst = (int)it.first + wave * Cfg::BPW;
    const int nb = min(max((int)it.count - wave * Cfg::BPW, 0), Cfg::BPW);
    const int c = (int)(it.type >> 8);  // channel of this item
    float* lds = lds_wg + wave * (Cfg::BPW * IMG);
    const int FW = f.width;
    float* out = c == 0 ? o0 : c == 1 ? o1 : o2;
    const int* qc_plane = f.coeff[c];
    const int* qy_plane = f.coeff[1];
    // reciprocal weights, stored so that consecutive x are consecutive addresses for both orientations
    const float* wc = (FLIP ? f.weights_t : f.weights) + f.woffs[PI * 3 + c];
    const float* wy = (FLIP ? f.weights_t : f.weights) + f.woffs[PI * 3 + 1];
    const float qbn = f.quant_bias_numerator;
    float* qtab = lds_wg + 4 * Cfg::BPW * IMG;
    // the block record first: its load is in flight while the qbn/|q| table is built (the barrier would pin it behind)
    const int bi_col = lane / W;
    DevBlock b_col{};
    if (bi_col < nb) b_col = load_block(blocks, wfirst + bi_col);
    if (wave == 0) qtab[lane] = lane > 0 ? qbn / (float)lane : 0.0f;
    __syncthreads();

    // ---- column pass
    {
        const int bi = bi_col, x = lane % W;
        if (bi < nb) {
            const DevBlock b = b_col;
            const int py0 = b.cy * 8, px0 = b.cx * 8;
            const float hfm = (float)b.hf_mul;
            const float sfc = f.scale_factor[c] / hfm, sfy = f.scale_factor[1] / hfm;
            const float qbc = f.quant_bias[c], qby = f.quant_bias[1];
            const int ty0 = py0 >> 6, tx0 = px0 >> 6;
            const int tx = (px0 + x) >> 6;
            const float* lfp = f.lf[c] + (int64_t)b.cy * f.bw + b.cx;
            const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(H)));
            float kcfl = 0.0f;
            MirrorAcc<H> acc;
            // rows in chunks of RC: all global loads of a chunk are issued before its arithmetic, so RC (x2-4)
            // loads per lane are in flight instead of one dependent load per row
            constexpr int RC = MAXRC;
#pragma unroll 1
            for (int n0 = 0; n0 < H; n0 += RC) {
                int qcv[RC], qyv[RC];
                float wcv[RC], wyv[RC];
                static_assert(RC == 4 || RC == 8, "a chunk of rows stays inside one cell row of the tiled plane");
                const int64_t off0 = coeff_off(FW, py0 + n0, px0 + x);
#pragma unroll
                for (int r = 0; r < RC; r++) {
                    qcv[r] = qc_plane[off0 + r * 8];
                    wcv[r] = wc[(n0 + r) * W + x];  // (FLIP ? transposed table : table)[n][x]
                    if (c != 1 && !f.no_cfl) {
                        qyv[r] = qy_plane[off0 + r * 8];
                        wyv[r] = wy[(n0 + r) * W + x];
                    }
                }
#pragma unroll
                for (int r = 0; r < RC; r++) {
                    const int n = n0 + r;
                    float co;
                    if (c != 1 && !f.no_cfl && (n == 0 || ((py0 + n) & 63) == 0)) {  // entering a new CfL tile row
                        const int ty = (py0 + n) >> 6;
                        float kX, kB;
                        cfl_factors(f, ty, tx, (b.cfl_zero >> ((ty - ty0) * 5 + (tx - tx0))) & 1u, kX, kB);
                        kcfl = c == 0 ? kX : kB;
                    }
                    if (n < DSH && x < DSW) {
                        co = llf_coeff<DSH, DSW>(f, lfp, n, x);  // finalizeLLF (:194-229)
                    } else {
                        co = dequant1_tab(qcv[r], qbc, qbn, sfc, wcv[r], qtab);
                        if (c != 1 && !f.no_cfl) {
                            const float dy = dequant1_tab(qyv[r], qby, qbn, sfy, wyv[r], qtab);
                            co = co + kcfl * dy;  // chromaFromLuma (:186-188)
                        }
                    }
                    // packed f32 (v_pk_mul_f32 + v_pk_add_f32: one rounding per multiply and per add, same order per
                    // output); n0 is a multiple of RC = 8, so the parity of n is the parity of r
                    if (n == 0) acc.init(co);
                    else if (r & 1) acc.template step<true>(co, lut + (n - 1) * H);
                    else acc.template step<false>(co, lut + (n - 1) * H);
                }
            }
            float* dcol = lds + bi * IMG + x;
#pragma unroll
            for (int k = 0; k < H; k++) dcol[k * LD] = acc.get(k);
        }
    }
    wave_lds_fence();  // the wave's image is complete before its own row pass reads it
    // ---- row pass
    {
        const int bi = lane / H, y = lane % H;
        if (bi < nb) {
            const DevBlock b = load_block(blocks, wfirst + bi);
            const float* row = lds + bi * IMG + y * LD;
            const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(W)));
            MirrorAcc<W> acc;
            acc.init(row[0]);
#pragma unroll 4
            for (int n = 1; n < W; n += 2) {  // (odd, even) pairs; W is even, so the last odd n stands alone
                acc.template step<true>(row[n], lut + (n - 1) * W);
                if (n + 1 < W) acc.template step<false>(row[n + 1], lut + n * W);
            }
            float* o = out + (int64_t)(b.cy * 8 + y) * FW + b.cx * 8;
#pragma unroll
            for (int k = 0; k < W; k += 4) *reinterpret_cast<float4*>(o + k) = make_float4(acc.get(k), acc.get(k + 1), acc.get(k + 2), acc.get(k + 3));
        }
    }
}

// The same transform for all three channels of the wave's blocks, luma first: the dequantised luma column stays in
// registers and feeds chromaFromLuma of X and B, instead of being loaded and dequantised again by separate chroma work
// items (5 coefficient + 5 weight loads and 5 dequantisations per pixel become 3 + 3 + 3). Used for frames without
// chroma subsampling; the channel loop is rolled, so the code size is that of medium_item.
template <int H, int W, int TYPE>
__device__ __forceinline__ void medium_item3(const DevFrame& f, const DevBlock* __restrict__ blocks, const WorkItem it,
                                             float* __restrict__ lds_wg, float* __restrict__ o0, float* __restrict__ o1,
                                             float* __restrict__ o2) {
    using Cfg = MediumCfg<H, W>;
    constexpr int LD = Cfg::LD, IMG = Cfg::IMG;
    constexpr int PI = JXL_TT[TYPE].param_index;
    constexpr bool FLIP = H >= W;  // TransformType.flip() for METHOD_DCT
    constexpr int DSH = H / 8, DSW = W / 8;
    constexpr int RC = 8;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wfirst = (int)it.first + wave * Cfg::BPW;
    const int nb = min(max((int)it.count - wave * Cfg::BPW, 0), Cfg::BPW);
    float* lds = lds_wg + wave * (Cfg::BPW * IMG);
    const int FW = f.width;
    const float qbn = f.quant_bias_numerator;
    // qbn / |q| table, one per WAVE (64 entries, one division per lane): with the wave-private transposes this leaves the
    // wave path without a single workgroup barrier -- the four waves of a workgroup never wait for each other
    float* qtab = lds_wg + 4 * Cfg::BPW * IMG + wave * 64;
    const int bi_col = lane / W, x = lane % W;
    const int bi_row = lane / H, y = lane % H;
    JXL_STAMP(0);
    JXL_STAMP_VAL(7, TYPE);
    DevBlock b_col{}, b_row{};
    if (bi_col < nb) b_col = load_block(blocks, wfirst + bi_col);
    if (bi_row < nb) b_row = load_block(blocks, wfirst + bi_row);
    qtab[lane] = lane > 0 ? qbn / (float)lane : 0.0f;
    wave_lds_fence();
    const int py0 = b_col.cy * 8, px0 = b_col.cx * 8;
    const float hfm = (float)b_col.hf_mul;
    const int ty0 = py0 >> 6, tx0 = px0 >> 6;
    const int tx = (px0 + x) >> 6;
    const cfloatp lut_h = as_const(f.lut + lut_off(ceil_log2_dev(H)));
    const cfloatp lut_w = as_const(f.lut + lut_off(ceil_log2_dev(W)));
    float dyv[H];  // dequantised luma of this lane's column (HFCoefficients.java:186-188 reads it for X and B)
#pragma unroll
    for (int n = 0; n < H; n++) dyv[n] = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < 3; pass++) {
        const int c = pass == 0 ? 1 : pass == 1 ? 0 : 2;
        // ---- column pass
        if (bi_col < nb) {
            const int* qc_plane = f.coeff[c];
            const float* wc = (FLIP ? f.weights_t : f.weights) + f.woffs[PI * 3 + c];
            const float sfc = f.scale_factor[c] / hfm;
            const float qbc = f.quant_bias[c];
            const float* lfp = f.lf[c] + (int64_t)b_col.cy * f.bw + b_col.cx;
            float kcfl = 0.0f;
            MirrorAcc<H> acc;
#pragma unroll
            for (int n0 = 0; n0 < H; n0 += RC) {
                int qcv[RC];
                float wcv[RC];
                static_assert(RC == 4 || RC == 8, "a chunk of rows stays inside one cell row of the tiled plane");
                const int64_t off0 = coeff_off(FW, py0 + n0, px0 + x);
#pragma unroll
                for (int r = 0; r < RC; r++) {
                    qcv[r] = qc_plane[off0 + r * 8];
                    wcv[r] = wc[(n0 + r) * W + x];  // (FLIP ? transposed table : table)[n][x]
                }
#pragma unroll
                for (int r = 0; r < RC; r++) {
                    const int n = n0 + r;
                    if (c != 1 && (n == 0 || ((py0 + n) & 63) == 0)) {  // entering a new CfL tile row
                        const int ty = (py0 + n) >> 6;
                        float kX, kB;
                        cfl_factors(f, ty, tx, (b_col.cfl_zero >> ((ty - ty0) * 5 + (tx - tx0))) & 1u, kX, kB);
                        kcfl = c == 0 ? kX : kB;
                    }
                    float co = dequant1_tab(qcv[r], qbc, qbn, sfc, wcv[r], qtab);
                    if (c == 1) dyv[n] = co;
                    else co = co + kcfl * dyv[n];  // chromaFromLuma (:186-188)
                    if (n < DSH && x < DSW) co = llf_coeff<DSH, DSW>(f, lfp, n, x);  // finalizeLLF (:194-229)
                    if (n == 0) acc.init(co);
                    else if (r & 1) acc.template step<true>(co, lut_h + (n - 1) * H);
                    else acc.template step<false>(co, lut_h + (n - 1) * H);
                }
            }
            float* dcol = lds + bi_col * IMG + x;
#pragma unroll
            for (int k = 0; k < H; k++) dcol[k * LD] = acc.get(k);
        }
        wave_lds_fence();  // the wave's image is complete before its own row pass reads it
        // ---- row pass
        if (bi_row < nb) {
            const float* row = lds + bi_row * IMG + y * LD;
            MirrorAcc<W> acc;
            acc.init(row[0]);
#pragma unroll 4
            for (int n = 1; n < W; n += 2) {  // (odd, even) pairs; W is even, so the last odd n stands alone
                acc.template step<true>(row[n], lut_w + (n - 1) * W);
                if (n + 1 < W) acc.template step<false>(row[n + 1], lut_w + n * W);
            }
            float* o = (c == 0 ? o0 : c == 1 ? o1 : o2) + (int64_t)(b_row.cy * 8 + y) * FW + b_row.cx * 8;
#pragma unroll
            for (int k = 0; k < W; k += 4) *reinterpret_cast<float4*>(o + k) = make_float4(acc.get(k), acc.get(k + 1), acc.get(k + 2), acc.get(k + 3));
        }
        wave_lds_fence();  // the row pass has read the image before the next channel's column pass overwrites it
        JXL_STAMP(1 + pass);
    }
    JXL_STAMP(5);
}

// finalizeLLF (HFCoefficients.java:194-229) of every block larger than 8x8, written over the block's own cells
// of the llf planes (a block covers exactly dctSelectHeight x dctSelectWidth cells). grid = (3, nblocks).
__global__ __launch_bounds__(256) void k_llf(const DevFrame f, const DevBlock* __restrict__ blocks, int first, float* l0, float* l1,
                                             float* l2) {
    __shared__ float s0[1024], s1[1024];
    const DevBlock b = blocks[first + blockIdx.y];
    const jxl_tt_info tt = JXL_TT[b.type];
    const int c = blockIdx.x;
    float* o = (c == 0 ? l0 : c == 1 ? l1 : l2) + (int64_t)b.cy * f.bw + b.cx;
    const int bw = f.bw;
    llf_block(f, b.cy, b.cx, c, tt.ph >> 3, tt.pw >> 3, s0, s1, threadIdx.x, 256, [&](int ky, int kx, float v) { o[ky * bw + kx] = v; });
}

void launch_llf(const DevFrame& f, const DevBlock* blocks, int first, int count, float* const llf[3], hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_llf, dim3(3, count), dim3(256), 0, s, f, blocks, first, llf[0], llf[1], llf[2]);
}

// ---- 64-point family (64x64, 64x32, 32x64): one 256-thread workgroup per NBLK blocks of one channel ---------
// A 64-point column/row has 63 x 64 multiply-adds; one lane per column would serialise 8k dependent-issue
// instructions per pass. Here the dequantised coefficients are staged once in LDS and every 1-D transform is
// split over `chunks` waves, each lane producing KC = 16 outputs of its column / row; a wave's lanes share the
// output chunk, so the LUT slice stays wave-uniform (scalar loads). Two LDS images (ping-pong).
template <int H, int W>
struct Wg64Cfg {
    static constexpr int NBLK = 64 / (H < W ? H : W);  // blocks per workgroup
    static constexpr int LD = W + 1;
    static constexpr int IMG = H * LD;
    static constexpr int CH_COL = 256 / (NBLK * W), KC_COL = H / CH_COL;  // column pass: chunks, outputs per lane
    static constexpr int CH_ROW = 256 / (NBLK * H), KC_ROW = W / CH_ROW;
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(2 * NBLK * IMG + 64);
};

template <int H, int W, int TYPE>
__device__ __forceinline__ void wg64_item(const DevFrame& f, const DevBlock* __restrict__ blocks, const WorkItem it,
                                          float* __restrict__ lds, float* __restrict__ o0, float* __restrict__ o1,
                                          float* __restrict__ o2) {
    using Cfg = Wg64Cfg<H, W>;
    constexpr int NBLK = Cfg::NBLK, LD = Cfg::LD, IMG = Cfg::IMG;
    constexpr int PI = JXL_TT[TYPE].param_index;
    constexpr bool FLIP = H >= W;
    constexpr int DSH = H / 8, DSW = W / 8;
    const int nb = (int)it.count;
    const int c = (int)(it.type >> 8);
    const int tid = threadIdx.x;
    const int FW = f.width;
    float* out = c == 0 ? o0 : c == 1 ? o1 : o2;
    float* img0 = lds;               // dequantised coefficients, later the row-pass input
    float* img1 = lds + NBLK * IMG;  // column-pass output
    float* qtab = lds + 2 * NBLK * IMG;
    const float qbn = f.quant_bias_numerator;
    JXL_STAMP(0);
    JXL_STAMP_VAL(7, TYPE);
    if (tid < 64) qtab[tid] = tid > 0 ? qbn / (float)tid : 0.0f;
    __syncthreads();
    // 1. dequant + CfL + LLF -> img0. Sample s = j*256 + tid of the NBLK blocks (x fastest: coalesced rows); all
    //    global loads of the workgroup's NS sweeps are issued before the arithmetic.
    {
        const float* wc = (FLIP ? f.weights_t : f.weights) + f.woffs[PI * 3 + c];
        const float* wy = (FLIP ? f.weights_t : f.weights) + f.woffs[PI * 3 + 1];
        constexpr int RSTEP = 256 / W;          // rows covered per sweep
        constexpr int BSTEP = (H * W) / 256;    // sweeps per block
        constexpr int NS = NBLK * BSTEP;        // sweeps per workgroup
        const int x = tid % W, r0 = tid / W;
        int qcv[NS], qyv[NS];
        float wcv[NS], wyv[NS];
        // per-block scale factors once per lane (scaleFactor[c] / hfMul, HFCoefficients.java:299): NBLK divisions instead
        // of one per sample
        float sfc_b[NBLK], sfy_b[NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; bi++) {
            sfc_b[bi] = sfy_b[bi] = 0.0f;
            if (bi < nb) {
                const DevBlock b = load_block(blocks, (int)it.first + bi);
                const float hfm = (float)b.hf_mul;
                sfc_b[bi] = f.scale_factor[c] / hfm;
                sfy_b[bi] = f.scale_factor[1] / hfm;
            }
        }
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int bi = j / BSTEP, n = (j % BSTEP) * RSTEP + r0;
            qcv[j] = qyv[j] = 0;
            wcv[j] = wyv[j] = 0.0f;
            if (bi < nb) {
                const DevBlock b = load_block(blocks, (int)it.first + bi);
                const int64_t off = coeff_off(FW, b.cy * 8 + n, b.cx * 8 + x);
                qcv[j] = f.coeff[c][off];
                wcv[j] = wc[n * W + x];
                if (c != 1 && !f.no_cfl) {
                    qyv[j] = f.coeff[1][off];
                    wyv[j] = wy[n * W + x];
                }
            }
        }
        JXL_STAMP(1);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int bi = j / BSTEP, n = (j % BSTEP) * RSTEP + r0;
            if (bi < nb) {
                const DevBlock b = load_block(blocks, (int)it.first + bi);
                const int py0 = b.cy * 8, px0 = b.cx * 8;
                float co;
                if (n < DSH && x < DSW) {
                    co = llf_coeff<DSH, DSW>(f, f.lf[c] + (int64_t)b.cy * f.bw + b.cx, n, x);
                } else {
                    co = dequant1_tab(qcv[j], f.quant_bias[c], qbn, sfc_b[bi], wcv[j], qtab);
                    if (c != 1 && !f.no_cfl) {
                        const int ty = (py0 + n) >> 6, tx = (px0 + x) >> 6;
                        float kX, kB;
                        cfl_factors(f, ty, tx, (b.cfl_zero >> ((ty - (py0 >> 6)) * 5 + (tx - (px0 >> 6)))) & 1u, kX, kB);
                        const float dy = dequant1_tab(qyv[j], f.quant_bias[1], qbn, sfy_b[bi], wyv[j], qtab);
                        co = co + (c == 0 ? kX : kB) * dy;
                    }
                }
                img0[bi * IMG + n * LD + x] = co;
            }
        }
    }
    __syncthreads();
    JXL_STAMP(2);
    // 2. column pass: lanes enumerate (chunk, block, column), x fastest
    {
        constexpr int KC = Cfg::KC_COL;
        const int col = tid % (NBLK * W), kc = __builtin_amdgcn_readfirstlane(tid / (NBLK * W));
        const int bi = col / W, x = col % W;
        if (bi < nb) {
            // chunk kc = outputs [kc*KC/2, (kc+1)*KC/2) and their mirror images [H-(kc+1)*KC/2, H-kc*KC/2): one product
            // serves output k and output H-1-k (see idct1d_reg), so a lane keeps both halves of its KC outputs
            const float* src = img0 + bi * IMG + x;
            const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(H)) + kc * (KC / 2));
            MirrorAcc<KC> acc;
            acc.init(src[0]);
#pragma unroll 4
            for (int n = 1; n < H; n += 2) {
                acc.template step<true>(src[n * LD], lut + (n - 1) * H);
                if (n + 1 < H) acc.template step<false>(src[(n + 1) * LD], lut + n * H);
            }
            float* d = img1 + bi * IMG + x;
#pragma unroll
            for (int i = 0; i < KC / 2; i++) {
                d[(kc * (KC / 2) + i) * LD] = acc.get(i);
                d[(H - 1 - kc * (KC / 2) - i) * LD] = acc.get(KC - 1 - i);
            }
        }
    }
    __syncthreads();
    JXL_STAMP(3);
    // 3. row pass: lanes enumerate (chunk, block, row), y fastest; each lane stores KC consecutive outputs
    {
        constexpr int KC = Cfg::KC_ROW;
        const int rr = tid % (NBLK * H), kc = __builtin_amdgcn_readfirstlane(tid / (NBLK * H));
        const int bi = rr / H, y = rr % H;
        if (bi < nb) {
            const DevBlock b = load_block(blocks, (int)it.first + bi);
            const float* row = img1 + bi * IMG + y * LD;
            const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(W)) + kc * (KC / 2));
            MirrorAcc<KC> acc;
            acc.init(row[0]);
#pragma unroll 4
            for (int n = 1; n < W; n += 2) {
                acc.template step<true>(row[n], lut + (n - 1) * W);
                if (n + 1 < W) acc.template step<false>(row[n + 1], lut + n * W);
            }
            JXL_STAMP(4);
            // two runs of KC/2 consecutive outputs: the low one ascending, the mirrored one ending at W-1-kc*KC/2
            float* o = out + (int64_t)(b.cy * 8 + y) * FW + b.cx * 8;
            float* olo = o + kc * (KC / 2);
            float* ohi = o + W - (kc + 1) * (KC / 2);
#pragma unroll
            for (int k = 0; k < KC / 2; k += 4) {
                *reinterpret_cast<float4*>(olo + k) = make_float4(acc.get(k), acc.get(k + 1), acc.get(k + 2), acc.get(k + 3));
                // ohi[t] is output W-(kc+1)*KC/2+t = mirror of low output i = KC/2-1-t, i.e. get(KC-1-i) = get(KC/2+t)
                *reinterpret_cast<float4*>(ohi + k) = make_float4(acc.get(KC / 2 + k), acc.get(KC / 2 + k + 1), acc.get(KC / 2 + k + 2), acc.get(KC / 2 + k + 3));
            }
        }
    }
    JXL_STAMP(5);
}

// which types take the workgroup-level (LDS-staged, k-split) path instead of the wave-level streamed one
__host__ __device__ constexpr bool use_wg_path(int h, int w) { return (h > w ? h : w) >= JXL_WG_PATH_MIN; }

// ---- launches: ONE kernel per register class, every transform type of the class in it -------------------------------
// Per-type launches are individually too small to fill 256 CUs (a 4K frame has ~1000 waves of most types) and the
// device runs at most 4 queues side by side, so a frame's 11 type kernels left most of the chip idle most of the time
// (measured: GPU_MAX_HW_QUEUES 2 -> 4 is +35 %, beyond 4 nothing). A merged kernel is a type switch around the same
// bodies; a workgroup is type-uniform, so nothing diverges. What makes it work:
//  (1) classes by register need: class 0 = the types whose body fits 64 VGPRs (8 waves per SIMD; the minimum is part of
//      __launch_bounds__ so the allocator is held to it -- without it the merged kernel came out at 256 VGPRs + scratch),
//      class 1 = the 64-point family together with DCT16 / DCT16x8 (all ~100 VGPRs, 4 waves per SIMD; the 64-point
//      workgroups are LDS-bound at 4 per CU anyway, and their few hundred long-running waves now share a grid with the
//      16-point types' thousands instead of holding a queue of their own);
//  (2) read-only tables through the constant address space (cfloatp, load_block): in a function this large the
//      compiler no longer proves that the LUT / block records are never clobbered, and uniform loads silently turn
//      into per-lane vector loads (the 64x64 body went from 173 to 330 us until this was done);
//  (3) segments ordered longest-running type first, so the tail of the launch is made of the cheapest workgroups, with
//      the XCD-aware order applied inside each segment;
//  (4) work items are implicit (segment descriptors in the kernel arguments) and the block record carries hfMultiplier,
//      so a workgroup's first dependent load is already its coefficients' address.
// 4K mixed frame: 12 launches -> 3 (+ the restoration kernel).
// ALLCH (frames without chroma subsampling): a wave-path item is a block group with all three channels (medium_item3);
// everything else -- the LDS-staged types, and every type of a per-channel launch -- is (block group, channel)
template <int H, int W, int TYPE, bool ALLCH>
__device__ __forceinline__ void type_body(const MultiArgs& a, int k, int li, float* lds) {
    constexpr int NB = use_wg_path(H, W) ? 64 / (H < W ? H : W) : 4 * (64 / (H > W ? H : W));
    constexpr bool ITEM3 = ALLCH && !use_wg_path(H, W);
    const int g = ITEM3 ? li : ALLCH ? li / 3 : li;
    const int ch = ITEM3 ? 0 : ALLCH ? li - 3 * g : a.ch0;
    WorkItem it;
    it.type = (uint32_t)TYPE | ((uint32_t)ch << 8);
    it.first = (uint32_t)(a.seg_first[k] + g * NB);
    it.count = (uint32_t)min(NB, a.seg_nblocks[k] - g * NB);
    if constexpr (use_wg_path(H, W)) wg64_item<H, W, TYPE>(a.f, a.blocks, it, lds, a.o0, a.o1, a.o2);
    else if constexpr (ALLCH) medium_item3<H, W, TYPE>(a.f, a.blocks, it, lds, a.o0, a.o1, a.o2);
    else medium_item<H, W, TYPE>(a.f, a.blocks, it, lds, a.o0, a.o1, a.o2);
}

template <int CLASS, bool ALLCH>
__device__ __forceinline__ void idct_multi_body(const MultiArgs& a, float* lds) {
    const int b = (int)blockIdx.x;
    if (b >= a.seg_b0[a.n_seg]) return;  // batched launch: the grid is sized for the frame with the most items
    int k = 0;
    while (k + 1 < a.n_seg && b >= a.seg_b0[k + 1]) k++;
    // XCD-aware item order inside the segment (see k_restore_fused): consecutive workgroup ids go to different XCDs, so
    // XCD x takes the x-th contiguous run of the segment's spatially ordered items (chroma items re-read luma: L2 hits)
    const int local = b - a.seg_b0[k], n = a.seg_n[k];
    const int per_xcd = (n + 7) >> 3;
    const int li = (local & 7) * per_xcd + (local >> 3);
    if (li >= n) return;
    const int t = a.seg_type[k];
    if (CLASS == 0) {
        switch (t) {
        case 0: type_body<8, 8, 0, ALLCH>(a, k, li, lds); break;
        case 5: type_body<32, 32, 5, ALLCH>(a, k, li, lds); break;
        case 7: type_body<8, 16, 7, ALLCH>(a, k, li, lds); break;
        case 8: type_body<32, 8, 8, ALLCH>(a, k, li, lds); break;
        case 9: type_body<8, 32, 9, ALLCH>(a, k, li, lds); break;
        case 10: type_body<32, 16, 10, ALLCH>(a, k, li, lds); break;
        case 11: type_body<16, 32, 11, ALLCH>(a, k, li, lds); break;
        default: break;
        }
    } else {
        switch (t) {
        case 18: type_body<64, 64, 18, ALLCH>(a, k, li, lds); break;
        case 19: type_body<64, 32, 19, ALLCH>(a, k, li, lds); break;
        case 20: type_body<32, 64, 20, ALLCH>(a, k, li, lds); break;
        case 4: type_body<16, 16, 4, ALLCH>(a, k, li, lds); break;
        case 6: type_body<16, 8, 6, ALLCH>(a, k, li, lds); break;
        default: break;
        }
    }
}

template <int CLASS, bool ALLCH>
__global__ __launch_bounds__(256, CLASS == 0 ? 8 : 4) void k_idct_multi(const MultiArgs a) {
    extern __shared__ float lds[];
    idct_multi_body<CLASS, ALLCH>(a, lds);
}

// The same launch for a BATCH of independent frames: blockIdx.y selects the frame's argument block in device memory
// (read through the constant address space: uniform, invariant, scalar loads -- exactly what the kernel-argument
// segment gives the single-frame form). One frame's share of a class is a few hundred to a few thousand waves; eight
// frames' shares in one grid keep all 256 CUs busy through the whole launch instead of 4 queues' worth of small kernels.
template <int CLASS>
__global__ __launch_bounds__(256, CLASS == 0 ? 8 : 4) void k_idct_multi_batch(const MultiArgs* __restrict__ args) {
    extern __shared__ float lds[];
    typedef const __attribute__((address_space(4))) MultiArgs* cargs;
    const MultiArgs& a = *(const MultiArgs*)((cargs)args + blockIdx.y);
    idct_multi_body<CLASS, true>(a, lds);
}

size_t medium_lds_bytes(int type);

#ifdef JXL_STAMPS
extern "C" int jxl_debug_set_stamps(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p);
}
#endif

int idct_class_of(int type) { return (type == 18 || type == 19 || type == 20 || type == 4 || type == 6) ? 1 : 0; }

// segs: the class's types in launch order. Returns the grid size (0: nothing to do).
int build_idct_multi_args(const DevFrame& f, const DevBlock* blocks, const IdctSegment* segs, int n_seg, int nch, int ch0,
                          float* const out[3], MultiArgs& a, size_t* lds_bytes_out) {
    a.f = f;
    a.blocks = blocks;
    a.items = nullptr;
    a.o0 = out[0]; a.o1 = out[1]; a.o2 = out[2];
    a.nch = nch;
    a.ch0 = ch0;
    a.n_seg = 0;
    int b0 = 0;
    size_t lds_bytes = 0;
    for (int i = 0; i < n_seg; i++) {
        if (segs[i].n_blocks <= 0) continue;
        if (a.n_seg >= MultiArgs::kMaxSeg) return -1;  // never drop blocks silently (finalize_tables checks the lists it builds)
        const int k = a.n_seg++;
        const int nb = medium_blocks_per_wg(segs[i].type);
        const int tt_h = JXL_TT[segs[i].type].ph, tt_w = JXL_TT[segs[i].type].pw;
        const bool item3 = nch == 3 && !use_wg_path(tt_h, tt_w);  // one item = all three channels of a block group
        const int n_items = ((segs[i].n_blocks + nb - 1) / nb) * (item3 ? 1 : nch);
        a.seg_b0[k] = b0;
        a.seg_n[k] = n_items;
        a.seg_type[k] = segs[i].type;
        a.seg_first[k] = segs[i].first_block;
        a.seg_nblocks[k] = segs[i].n_blocks;
        b0 += ((n_items + 7) / 8) * 8;
        lds_bytes = std::max(lds_bytes, medium_lds_bytes(segs[i].type));
    }
    for (int k = a.n_seg; k <= MultiArgs::kMaxSeg; k++) a.seg_b0[k] = b0;
    if (lds_bytes_out) *lds_bytes_out = lds_bytes;
    return a.n_seg == 0 ? 0 : b0;
}

void launch_idct_multi(const DevFrame& f, const DevBlock* blocks, int cls, const IdctSegment* segs, int n_seg, int nch, int ch0,
                       float* const out[3], hipStream_t s) {
    if (n_seg <= 0) return;
    MultiArgs a;
    size_t lds_bytes = 0;
    const int b0 = build_idct_multi_args(f, blocks, segs, n_seg, nch, ch0, out, a, &lds_bytes);
    if (b0 <= 0) return;
    const dim3 grid(b0), wg(256);
    if (nch == 3) {
        if (cls == 0) hipLaunchKernelGGL((k_idct_multi<0, true>), grid, wg, lds_bytes, s, a);
        else hipLaunchKernelGGL((k_idct_multi<1, true>), grid, wg, lds_bytes, s, a);
    } else {
        if (cls == 0) hipLaunchKernelGGL((k_idct_multi<0, false>), grid, wg, lds_bytes, s, a);
        else hipLaunchKernelGGL((k_idct_multi<1, false>), grid, wg, lds_bytes, s, a);
    }
}

// n_frames argument blocks at dev_args (device memory), grid_x = the largest frame's grid
void launch_idct_multi_batch(const MultiArgs* dev_args, int n_frames, int grid_x, size_t lds_bytes, int cls, hipStream_t s) {
    if (n_frames <= 0 || grid_x <= 0) return;
    const dim3 grid(grid_x, n_frames), wg(256);
    if (cls == 0) hipLaunchKernelGGL((k_idct_multi_batch<0>), grid, wg, lds_bytes, s, dev_args);
    else hipLaunchKernelGGL((k_idct_multi_batch<1>), grid, wg, lds_bytes, s, dev_args);
}

// blocks of one channel that one workgroup (work item) handles
int medium_blocks_per_wg(int type) {
    const int h = JXL_TT[type].ph, w = JXL_TT[type].pw;
    if (use_wg_path(h, w)) return 64 / (h < w ? h : w);
    return 4 * (64 / (h > w ? h : w));
}

size_t medium_lds_bytes(int type) {
    const int h = JXL_TT[type].ph, w = JXL_TT[type].pw;
    if (use_wg_path(h, w)) return sizeof(float) * (size_t)(2 * medium_blocks_per_wg(type) * h * (w + 1) + 64);
    return sizeof(float) * (size_t)(medium_blocks_per_wg(type) * h * (w + 1) + 4 * 64);
}

// the nine special 8x8-footprint types (Hornuss, DCT2, DCT4, DCT4x8, DCT8x4, AFV0-3): whole block in
// registers, so they get their own launch and register budget
__device__ __forceinline__ void idct_special_body(const DevFrame& f, const DevBlock* __restrict__ blocks, const WorkItem* __restrict__ items,
                                                  float* o0, float* o1, float* o2) {
    const WorkItem it = items[blockIdx.x];
    if (threadIdx.x >= it.count) return;
    const DevBlock b = load_block(blocks, (int)it.first + (int)threadIdx.x);
    const int c = (int)(it.type >> 8);  // one channel per item: 3x the waves, a third of the latency
    switch (it.type & 0xffu) {
    case 1: special_block<1>(f, b, c, o0, o1, o2); break;
    case 2: special_block<2>(f, b, c, o0, o1, o2); break;
    case 3: special_block<3>(f, b, c, o0, o1, o2); break;
    case 12: special_block<12>(f, b, c, o0, o1, o2); break;
    case 13: special_block<13>(f, b, c, o0, o1, o2); break;
    case 14: special_block<14>(f, b, c, o0, o1, o2); break;
    case 15: special_block<15>(f, b, c, o0, o1, o2); break;
    case 16: special_block<16>(f, b, c, o0, o1, o2); break;
    case 17: special_block<17>(f, b, c, o0, o1, o2); break;
    default: break;
    }
}

__global__ __launch_bounds__(64) void k_idct_special(const DevFrame f, const DevBlock* __restrict__ blocks,
                                                     const WorkItem* __restrict__ items, float* o0, float* o1, float* o2) {
    idct_special_body(f, blocks, items, o0, o1, o2);
}

// ---- the nine special 8x8 types, workgroup form (frames without chroma subsampling) -----------------------------------
// The lane-per-block kernel above makes every lane fetch its own 8 (+ 8 luma) rows one dependent load after the other,
// dequantises luma again for both chroma channels and divides once per sample: 20-27 us for a few hundred waves, the most
// expensive kernel of a real 720p frame. Here a 256-thread workgroup takes 64 blocks of one type with all three channels:
//   A. every lane fetches four 16-byte coefficient groups per channel -- all loads of the item are in flight at once --
//      dequantises them through the 3 x 64 sign / bias / (a - qbn / a) table (HFCoefficients.java:309-315, as in
//      k_idct_wg3), applies chroma-from-luma from the luma value it holds, and parks the block images in LDS;
//   B. lane (channel, block) lifts its 64 samples out of LDS (row stride 65: bank = lane + i), runs the type's transform
//      (invert_small: PassGroup.java:83-168, 234-325) in registers and puts the pixels back;
//   C. the lanes of A write the pixels out, 16 bytes each.
// Same operations per sample as the lane-per-block form, so the same bits.
#define JXL_SPECIAL_WG_LDS ((192 * 65 + 192) * sizeof(float))
__device__ __forceinline__ void special_wg_body(const DevFrame& f, const DevBlock* __restrict__ blocks, const WorkItem it, float* o0, float* o1,
                                                float* o2) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    __shared__ float img[192 * 65];
    __shared__ float qtab[192];
    const int tid = threadIdx.x;
    const int type = (int)(it.type & 0xffu), n = (int)it.count;
    const float qbn = f.quant_bias_numerator;
    if (tid < 192) {
        const int c = tid >> 6, a = tid & 63;
        qtab[tid] = a == 0 ? 0.0f : a == 1 ? f.quant_bias[c] : (float)a - qbn / (float)a;
    }
    const int PI = JXL_TT[type].param_index;
    const float* wt[3] = {f.weights + f.woffs[PI * 3], f.weights + f.woffs[PI * 3 + 1], f.weights + f.woffs[PI * 3 + 2]};
    const int W = f.width;
    // ---- A: loads of the four groups of this lane
    v4i_t q[4][3];
    v4f_t wv[3];
    int gx[4];
    float hfm[4], kx[4], kb[4], lfv[4][3];
    const int r = tid & 15, row = r >> 1, half = r & 1;  // the same (row, half) for all four groups: blocks tid/16 + 16 k
#pragma unroll
    for (int c = 0; c < 3; c++) wv[c] = *reinterpret_cast<const v4f_t*>(wt[c] + row * 8 + half * 4);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int blk = (tid >> 4) + 16 * k;
        gx[k] = -1;
        hfm[k] = 1.0f;
        kx[k] = kb[k] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            q[k][c] = v4i_t{0, 0, 0, 0};
            lfv[k][c] = 0.0f;
        }
        if (blk < n) {
            const DevBlock b = load_block(blocks, (int)it.first + blk);
            gx[k] = (int)b.cy | ((int)b.cx << 16);
            hfm[k] = (float)b.hf_mul;
            const int64_t off = coeff_off(W, b.cy * 8 + row, b.cx * 8 + half * 4);  // (the block's 16 lanes: 256 consecutive bytes)
#pragma unroll
            for (int c = 0; c < 3; c++) q[k][c] = *reinterpret_cast<const v4i_t*>(f.coeff[c] + off);
            cfl_factors(f, (b.cy * 8) >> 6, (b.cx * 8) >> 6, b.cfl_zero & 1u, kx[k], kb[k]);
            if (r == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) lfv[k][c] = f.lf[c][(int64_t)b.cy * f.bw + b.cx];
            }
        }
    }
    __syncthreads();  // qtab
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int blk = (tid >> 4) + 16 * k;
        if (gx[k] < 0) continue;
        const float sf[3] = {f.scale_factor[0] / hfm[k], f.scale_factor[1] / hfm[k], f.scale_factor[2] / hfm[k]};
        float dq[3][4];
        int big = 0;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qv = q[k][c][i];
                const int aq = qv < 0 ? -qv : qv;
                big |= aq;
                const float m = qtab[c * 64 + (aq & 63)];
                dq[c][i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, m) ^ ((uint32_t)qv & 0x80000000u));
            }
        if ((uint32_t)big >= 64u) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int qv = q[k][c][i];
                    const int aq = qv < 0 ? -qv : qv;
                    if (aq >= 64) dq[c][i] = (float)qv - qbn / (float)qv;
                }
        }
        float* d0 = img + (0 * 64 + blk) * 65 + row * 8 + half * 4;
        float* d1 = img + (1 * 64 + blk) * 65 + row * 8 + half * 4;
        float* d2 = img + (2 * 64 + blk) * 65 + row * 8 + half * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float dy = dq[1][i] * sf[1] * wv[1][i];
            float dx = dq[0][i] * sf[0] * wv[0][i] + kx[k] * dy;  // chromaFromLuma (HFCoefficients.java:186-188)
            float db = dq[2][i] * sf[2] * wv[2][i] + kb[k] * dy;
            float yy = dy;
            if (r == 0 && i == 0) {  // finalizeLLF of a 1x1 dctSelect: the LF sample itself (see small_row)
                dx = lfv[k][0];
                yy = lfv[k][1];
                db = lfv[k][2];
            }
            d0[i] = dx;
            d1[i] = yy;
            d2[i] = db;
        }
    }
    __syncthreads();
    // ---- B: one lane per (channel, block)
    if (tid < 192 && (tid & 63) < n) {
        float co[64], px[64];
        float* im = img + tid * 65;
#pragma unroll
        for (int i = 0; i < 64; i++) co[i] = im[i];
        switch (type) {
        case 1: invert_small<1>(co, px); break;
        case 2: invert_small<2>(co, px); break;
        case 3: invert_small<3>(co, px); break;
        case 12: invert_small<12>(co, px); break;
        case 13: invert_small<13>(co, px); break;
        case 14: invert_small<14>(co, px); break;
        case 15: invert_small<15>(co, px); break;
        case 16: invert_small<16>(co, px); break;
        case 17: invert_small<17>(co, px); break;
        default: break;
        }
#pragma unroll
        for (int i = 0; i < 64; i++) im[i] = px[i];
    }
    __syncthreads();
    // ---- C: pixels out
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (gx[k] < 0) continue;
        const int blk = (tid >> 4) + 16 * k;
        const int cy = gx[k] & 0xffff, cx = (int)((uint32_t)gx[k] >> 16);
        const int64_t off = (int64_t)(cy * 8 + row) * W + cx * 8 + half * 4;
        float* o3[3] = {o0, o1, o2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float* sp = img + (c * 64 + blk) * 65 + row * 8 + half * 4;
            *reinterpret_cast<v4f_t*>(o3[c] + off) = v4f_t{sp[0], sp[1], sp[2], sp[3]};
        }
    }
}

__global__ __launch_bounds__(256) void k_idct_special_wg(const DevFrame f, const DevBlock* __restrict__ blocks,
                                                         const WorkItem* __restrict__ items, float* o0, float* o1, float* o2) {
#ifdef JXL_IDCT_PRIO
    __builtin_amdgcn_s_setprio(JXL_IDCT_PRIO);
#endif
    special_wg_body(f, blocks, items[blockIdx.x], o0, o1, o2);
}

// batch form: MultiArgs block per frame (f, blocks, items, o0..2; seg_n[0] = number of items); items in the workgroup form
__global__ __launch_bounds__(256) void k_idct_special_batch(const MultiArgs* __restrict__ args) {
    typedef const __attribute__((address_space(4))) MultiArgs* cargs;
    const MultiArgs& a = *(const MultiArgs*)((cargs)args + blockIdx.y);
    if ((int)blockIdx.x >= a.seg_n[0]) return;
    special_wg_body(a.f, a.blocks, a.items[blockIdx.x], a.o0, a.o1, a.o2);
}

void launch_idct_special_batch(const MultiArgs* dev_args, int n_frames, int max_items, hipStream_t s) {
    if (n_frames <= 0 || max_items <= 0) return;
    hipLaunchKernelGGL(k_idct_special_batch, dim3(max_items, n_frames), dim3(256), 0, s, dev_args);
}

// wg_items: items name 64 blocks of all three channels (k_idct_special_wg); else one channel each (the lane-per-block kernel:
// chroma-subsampled frames, whose channels have their own geometry and no chroma-from-luma)
void launch_idct_special(const DevFrame& f, const DevBlock* blocks, const WorkItem* items, int n_items, float* const out[3],
                         hipStream_t s, bool wg_items) {
    if (n_items <= 0) return;
    if (wg_items) hipLaunchKernelGGL(k_idct_special_wg, dim3(n_items), dim3(256), 0, s, f, blocks, items, out[0], out[1], out[2]);
    else hipLaunchKernelGGL(k_idct_special, dim3(n_items), dim3(64), 0, s, f, blocks, items, out[0], out[1], out[2]);
}


// ---- large: 128/256-edge blocks through a scratch plane ---------------------------------------------
// phase A: dequant + CfL of every sample -> out planes (used as the coefficient store); grid.y = block
__global__ __launch_bounds__(256) void k_large_dequant(const DevFrame f, const DevBlock* __restrict__ blocks, int first,
                                                       float* o0, float* o1, float* o2) {
    const DevBlock b = blocks[first + blockIdx.y];
    const jxl_tt_info tt = JXL_TT[b.type];
    BlockCtx k;
    make_block_ctx(f, b, tt.ph, tt.pw, tt.param_index, k);
    const int n = tt.ph * tt.pw;
    const int lw = __builtin_ctz((unsigned)tt.pw);  // block edges are powers of two
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i >> lw, x = i & (tt.pw - 1);
        float dq[3];
        dequant_sample(f, k, y, x, dq);
        if (y < k.dsh && x < k.dsw) {  // finalizeLLF result (k_llf)
            const int64_t lo = (int64_t)(b.cy + y) * f.bw + b.cx + x;
            dq[0] = f.llf[0][lo];
            dq[1] = f.llf[1][lo];
            dq[2] = f.llf[2][lo];
        }
        const int64_t off = (int64_t)(k.py0 + y) * f.width + k.px0 + x;
        o0[off] = dq[0];
        o1[off] = dq[1];
        o2[off] = dq[2];
    }
}

// phase B: column pass. unit = (block, channel, 64-column strip, output chunk); one wave per unit, lane = column.
// dst[k][x] = idct over n of src[n][x]. A unit produces the 32 outputs [32 chunk, 32 chunk + 32) and their 32 mirror images
// (MirrorAcc: every product src[n] * lut[n-1][k] serves outputs k and N-1-k), so a table row costs 32 scalar registers instead
// of 64 -- room for the next row to be in flight while this one is used -- and 96 instead of 128 operations per input sample.
// The table goes through the constant address space: as plain global memory the 64 table values of a step were fetched by
// every lane (r1: 215 us for the default large mix's column passes).
__global__ __launch_bounds__(64) void k_large_colpass(const DevFrame f, const DevBlock* __restrict__ blocks, int first,
                                                      const float* s0p, const float* s1p, const float* s2p, float* d0, float* d1,
                                                      float* d2) {
    // grid = (block, channel, unit): the units a block does not have (4 of 16 for 128 x 128) are the LAST workgroups of the launch.
    // With the unit in x, those empty workgroups sat at fixed residues of the linear id, and the workgroups with work all
    // landed on half of the XCDs (id % 8): a 128 x 128 frame took 1.6x the time of a 256 x 256 one
    const DevBlock b = load_block(blocks, first + (int)blockIdx.x);
    const jxl_tt_info tt = JXL_TT[b.type];
    const int H = tt.ph, W = tt.pw;
    const int c = blockIdx.y;
    const int strips = W / 64, chunks = H / 64;
    const int u = blockIdx.z;
    if (u >= strips * chunks) return;
    const int strip = u % strips, chunk = u / strips;
    const float* src = c == 0 ? s0p : c == 1 ? s1p : s2p;
    float* dst = c == 0 ? d0 : c == 1 ? d1 : d2;
    const int FW = f.width;
    const int64_t base = (int64_t)(b.cy * 8) * FW + b.cx * 8 + strip * 64 + threadIdx.x;
    const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(H)) + chunk * 32);
    MirrorAcc<64> acc;
    acc.init(src[base]);
    // (odd, even) pairs of steps; H is even, so the last odd n stands alone
    float s1 = src[base + FW], s2 = src[base + 2 * (int64_t)FW];
    for (int n = 1; n < H; n += 2) {
        const float c1 = s1, c2 = s2;
        if (n + 2 < H) s1 = src[base + (int64_t)(n + 2) * FW];
        if (n + 3 < H) s2 = src[base + (int64_t)(n + 3) * FW];
        acc.template step<true>(c1, lut + (n - 1) * H);
        if (n + 1 < H) acc.template step<false>(c2, lut + n * H);
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        dst[base + (int64_t)(chunk * 32 + i) * FW] = acc.get(i);
        dst[base + (int64_t)(H - 1 - chunk * 32 - i) * FW] = acc.get(63 - i);
    }
}

// phase C: row pass. unit = (block, channel, 64-row strip, output chunk); lane = row, 64 x 64 tiles of the input staged through
// LDS so that global loads and stores stay row-contiguous; the outputs of a unit are the columns [32 chunk, 32 chunk + 32) and
// their mirror images [W - 32 chunk - 32, W - 32 chunk): two 128-byte runs per row.
__global__ __launch_bounds__(64) void k_large_rowpass(const DevFrame f, const DevBlock* __restrict__ blocks, int first,
                                                      const float* s0p, const float* s1p, const float* s2p, float* d0, float* d1,
                                                      float* d2) {
    __shared__ float tile[64 * 65];
    const DevBlock b = load_block(blocks, first + (int)blockIdx.x);
    const jxl_tt_info tt = JXL_TT[b.type];
    const int H = tt.ph, W = tt.pw;
    const int c = blockIdx.y;
    const int strips = H / 64, chunks = W / 64;
    const int u = blockIdx.z;
    if (u >= strips * chunks) return;
    const int strip = u % strips, chunk = u / strips;
    const float* src = c == 0 ? s0p : c == 1 ? s1p : s2p;
    float* dst = c == 0 ? d0 : c == 1 ? d1 : d2;
    const int FW = f.width;
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)(b.cy * 8 + strip * 64) * FW + b.cx * 8;
    const cfloatp lut = as_const(f.lut + lut_off(ceil_log2_dev(W)) + chunk * 32);
    MirrorAcc<64> acc;
    for (int n0 = 0; n0 < W; n0 += 64) {
        __syncthreads();
        float t[64];
#pragma unroll
        for (int r = 0; r < 64; r++) t[r] = src[base + (int64_t)r * FW + n0 + lane];  // all 64 row loads of the tile in flight
#pragma unroll
        for (int r = 0; r < 64; r++) tile[r * 65 + lane] = t[r];
        __syncthreads();
        const float* row = tile + lane * 65;
        if (n0 == 0) acc.init(row[0]);
        // n = n0 + nn: nn and n have the same parity (n0 is a multiple of 64)
        for (int nn = n0 == 0 ? 1 : 0; nn < 64; nn++) {
            const int n = n0 + nn;
            if (nn & 1) acc.template step<true>(row[nn], lut + (n - 1) * W);
            else acc.template step<false>(row[nn], lut + (n - 1) * W);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64; i++) tile[lane * 65 + i] = acc.get(i);  // i < 32: column 32 chunk + i; i >= 32: W - 32 chunk - 64 + i
    __syncthreads();
    const int col = lane < 32 ? chunk * 32 + lane : W - 32 * chunk - 64 + lane;
    for (int r = 0; r < 64; r++) dst[base + (int64_t)r * FW + col] = tile[r * 65 + lane];
}

void launch_idct_large(const DevFrame& f, const DevBlock* blocks, const DevBlock* host_blocks, int first, int count,
                       float* const out[3], float* const scratch[3], hipStream_t s, int* n_launches) {
    (void)host_blocks;
    if (count <= 0) return;
    // every large type has <= 256x256 samples: 32 workgroups of 256 threads x 8 samples
    hipLaunchKernelGGL(k_large_dequant, dim3(32, count), dim3(256), 0, s, f, blocks, first, out[0], out[1], out[2]);
    // at most (256/64)*(256/64) = 16 units per (block, channel)
    hipLaunchKernelGGL(k_large_colpass, dim3(count, 3, 16), dim3(64), 0, s, f, blocks, first, out[0], out[1], out[2],
                       scratch[0], scratch[1], scratch[2]);
    hipLaunchKernelGGL(k_large_rowpass, dim3(count, 3, 16), dim3(64), 0, s, f, blocks, first, scratch[0], scratch[1],
                       scratch[2], out[0], out[1], out[2]);
    if (n_launches) *n_launches += 3;
}

__global__ void k_accumulate(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (int32_t)((uint32_t)dst[i] + (uint32_t)src[i]);  // PassGroup.java:174-200, Java int wrap
}

void launch_accumulate(int32_t* dst, const int32_t* src, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_accumulate, dim3(grid), dim3(256), 0, s, dst, src, n);
}

// ---- single-block 2-D transforms for the stage-level entry points ------------------------------------
// One workgroup, everything in global memory, one thread per 1-D transform: slow, exact, any size.
__global__ __launch_bounds__(256) void k_idct2d_single(const float* src, float* dst, float* tmp, int h, int w, int transposed,
                                                       const float* lut_all) {
    const int lh = ceil_log2_dev(h), lw = ceil_log2_dev(w);
    const float* luth = lut_all + lut_off(lh);
    const float* lutw = lut_all + lut_off(lw);
    if (transposed) {
        for (int y = threadIdx.x; y < h; y += 256)
            for (int k = 0; k < w; k++) {
                float d = src[y * w];
                for (int n = 1; n < w; n++) d = d + src[y * w + n] * lutw[(n - 1) * w + k];
                tmp[y * w + k] = d;
            }
        __syncthreads();
        for (int x = threadIdx.x; x < w; x += 256)
            for (int k = 0; k < h; k++) {
                float d = tmp[x];
                for (int n = 1; n < h; n++) d = d + tmp[n * w + x] * luth[(n - 1) * h + k];
                dst[x * h + k] = d;
            }
    } else {
        for (int x = threadIdx.x; x < w; x += 256)
            for (int k = 0; k < h; k++) {
                float d = src[x];
                for (int n = 1; n < h; n++) d = d + src[n * w + x] * luth[(n - 1) * h + k];
                tmp[k * w + x] = d;
            }
        __syncthreads();
        for (int y = threadIdx.x; y < h; y += 256)
            for (int k = 0; k < w; k++) {
                float d = tmp[y * w];
                for (int n = 1; n < w; n++) d = d + tmp[y * w + n] * lutw[(n - 1) * w + k];
                dst[y * w + k] = d;
            }
    }
}

// MathHelper.forwardDCT2D (MathHelper.java:124-136)
__global__ __launch_bounds__(256) void k_fdct2d_single(const float* src, float* dst, float* tmp, int h, int w,
                                                       const float* lut_all) {
    const int lh = ceil_log2_dev(h), lw = ceil_log2_dev(w);
    const float* luth = lut_all + lut_off(lh);
    const float* lutw = lut_all + lut_off(lw);
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const int y = i / w, k = i % w;
        const float inv = 1.0f / (float)w;
        float d2;
        if (k == 0) {
            d2 = src[y * w];
            for (int x = 1; x < w; x++) d2 = d2 + src[y * w + x];
        } else {
            d2 = src[y * w] * lutw[(k - 1) * w];
            for (int n = 1; n < w; n++) d2 = d2 + src[y * w + n] * lutw[(k - 1) * w + n];
        }
        tmp[y * w + k] = d2 * inv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const int x = i / h, k = i % h;
        const float inv = 1.0f / (float)h;
        float d2;
        if (k == 0) {
            d2 = tmp[x];
            for (int y = 1; y < h; y++) d2 = d2 + tmp[y * w + x];
        } else {
            d2 = tmp[x] * luth[(k - 1) * h];
            for (int n = 1; n < h; n++) d2 = d2 + tmp[n * w + x] * luth[(k - 1) * h + n];
        }
        dst[k * w + x] = d2 * inv;
    }
}

static float* g_single_tmp = nullptr;

void launch_idct2d_single(const float* src, float* dst, int h, int w, int transposed, const float* lut, hipStream_t s) {
    if (!g_single_tmp) (void)hipMalloc(&g_single_tmp, sizeof(float) * 256 * 256);
    hipLaunchKernelGGL(k_idct2d_single, dim3(1), dim3(256), 0, s, src, dst, g_single_tmp, h, w, transposed, lut);
}

void launch_fdct2d_single(const float* src, float* dst, int h, int w, const float* lut, hipStream_t s) {
    if (!g_single_tmp) (void)hipMalloc(&g_single_tmp, sizeof(float) * 256 * 256);
    hipLaunchKernelGGL(k_fdct2d_single, dim3(1), dim3(256), 0, s, src, dst, g_single_tmp, h, w, lut);
}

}  // namespace jxl
