// VarDCT stage 1 for the small METHOD_DCT varblocks (8x8, 16x16, 16x8, 8x16 -- two thirds of the pixels of a photographic
// frame): dequantisation + chroma-from-luma + finalizeLLF + inverse DCT, one WAVE per work item, no workgroup barrier,
// every 1-D transform whole inside one lane (round 3).
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/vardct/HFCoefficients.java:140-319  bakeDequantizedCoeffs (dequant, CfL, finalizeLLF)
//   J/frame/group/PassGroup.java:229-233        METHOD_DCT branch of invertVarDCT
//   J/util/MathHelper.java:68-136               inverseDCT2D (columns, then rows), forwardDCT2D for the LLF corner
//
// Why another IDCT kernel (k_idct_wg3.hip stays for the 32- and 64-point types): measured in round 2, the persistent
// workgroup kernel spends a 4K frame of DCT8 blocks in 55 us with 9 us worth of multiply-adds in it; what it pays for is
// four workgroup barriers per item, lane groups that share a column (so the cosine table reaches the multiplier through
// scalar registers: an SGPR operand HALVES the v_mul_f32 issue rate on gfx950, tools/ubench/op_rate.hip) and a separate
// LLF launch that the whole stage waits for. Here
//   * a work item = NB varblocks of one type (64 columns and >= 64 rows: 8 DCT8 blocks, 4 DCT16 blocks, 8 16x8 / 8x16
//     blocks) and belongs to ONE wave: the LDS image is wave-private, LDS operations of a wave execute in order, so the
//     phases need no s_barrier at all and waves of a workgroup drift freely (16 waves per CU hide each other's latencies);
//   * a lane owns a whole column (then a whole row): out[k] += in[n] * lut[n-1][k] runs with all N accumulators in
//     registers, the input sample read once from LDS per step and the table entries as INSTRUCTION LITERALS (full-rate
//     v_mul_f32; lut_wave.inc), each product used for output k and its mirror image N-1-k (MirrorAcc of k_idct.hip);
//   * finalizeLLF is computed inside the item (a 2x2 / 2x1 / 1x2 forward DCT of the LF patch by the first lanes): no LLF
//     launch, no dependency of this launch on anything but the frame's inputs;
//   * channels run Y, X, B through one single-channel image (5.4 KB per wave); Y's dequantised samples stay in registers for
//     the chroma-from-luma of X and B (luma is dequantised once).
// Bit-exactness: every sum keeps the reference's order, multiplies and adds are separate IEEE f32 operations.
#include "jxl_internal.h"
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "../../include/jxl_tables.h"

namespace jxl {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) int* cintp;

template <int N>
struct WLut;
#include "lut_wave.inc"

constexpr float kLlfScaleW[32] = JXL_LLF_SCALE_INIT;

// orders the LDS traffic of the wave's phases for the COMPILER only (the hardware executes a wave's DS operations in order)
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int H_, int W_>
struct WCfg {
    static constexpr int H = H_, W = W_;
    static constexpr int NB = 64 / (H < W ? H : W);  // blocks per item: 64 columns and 64 rows at least
    static constexpr int P = NB * H * W;              // sample positions per channel
    static constexpr int NG = P / 256;                // 4-sample groups per lane and channel
    static constexpr int GPB = H * W / 4;             // groups per block
    static constexpr int LD = W + 1;                  // odd row stride: column and row walks hit 32 different banks
    static constexpr int IMG0 = H * LD;
    static constexpr int IMG = W >= 32 ? IMG0 : IMG0 + ((W - IMG0 % 32) + 32) % 32;  // blocks land W banks apart
    static constexpr int CL = NB * W / 64;            // columns per lane
    static constexpr int RL = NB * H / 64;            // rows per lane
    static constexpr int DSH = H / 8, DSW = W / 8, DS = DSH * DSW;
    static constexpr int NLLF = NB * 3 * DS;          // LLF coefficients per item (<= 48)
    static_assert(64 % GPB == 0 || GPB % 64 == 0, "a lane's groups sit at the same place of their blocks");
    static_assert(NLLF <= 64, "one LLF coefficient per lane");
};
constexpr int kWaveRecFloats = 8 * 8 + 48;  // block records + LLF coefficients (small types) / pointer table (32-point types)

// MathHelper.inverseDCTHorizontal (MathHelper.java:68-78): dest = src[0], then n = 1 .. N-1 in order, all N outputs of the
// line in this lane's registers
template <int N>
__device__ __forceinline__ void idct1d_lane(float (&acc)[N], const float* p, int stride) {
    float s[N];
#pragma unroll
    for (int n = 0; n < N; n++) s[n] = p[n * stride];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = s[0];
#pragma unroll
    for (int n = 1; n < N; n++) {
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            const float pr = s[n] * WLut<N>::v[n - 1][k];
            acc[k] = acc[k] + pr;
            acc[N - 1 - k] = (n & 1) ? acc[N - 1 - k] - pr : acc[N - 1 - k] + pr;  // lut[n-1][N-1-k] == (-1)^n lut[n-1][k]
        }
    }
}

// One LLF coefficient (ky, kx) of a DSH x DSW LF patch p (row-major): forwardDCT2D (MathHelper.java:80-95,124-136: rows,
// then columns) times llfScale (HFCoefficients.java:194-229). The 2-point table row is {1, -1}.
template <int DSH, int DSW>
__device__ __forceinline__ float llf_coeff_w(const float* p, int ky, int kx) {
    float r[DSH];
#pragma unroll
    for (int y = 0; y < DSH; y++) {
        float d2;
        if (DSW == 1) d2 = p[y];
        else d2 = kx == 0 ? p[y * 2] + p[y * 2 + 1] : p[y * 2] * 1.0f + p[y * 2 + 1] * -1.0f;
        r[y] = d2 * (1.0f / (float)DSW);
    }
    float d2;
    if (DSH == 1) d2 = r[0];
    else d2 = ky == 0 ? r[0] + r[1] : r[0] * 1.0f + r[1] * -1.0f;
    const float sy = DSH == 1 ? kLlfScaleW[0] : (ky == 0 ? kLlfScaleW[0] : kLlfScaleW[16]);
    const float sx = DSW == 1 ? kLlfScaleW[0] : (kx == 0 ? kLlfScaleW[0] : kLlfScaleW[16]);
    return (d2 * (1.0f / (float)DSH)) * (sy * sx);
}

// what a lane holds of an item between its requests and its arithmetic
template <int NG, int DS>
struct WRaw {
    v4i q[NG][3];          // quantised coefficients of 4 consecutive x; slots in processing order Y, X, B
    float kx[NG], kb[NG];  // CfL factors of the group's 64x64 tile (0 where the reference's cache reads 0)
    int gb[NG];            // block of group j inside the item, or -1
    float lfp[DS];         // lanes < NLLF: the LF patch of the lane's LLF coefficient
};

// block record of lane < NB (DevBlock words), or zeros
__device__ __forceinline__ v4i wave_load_rec(const WaveArgs& a, int first, int nb, int lane, int NB) {
    v4i rec = v4i{0, 0, 0, 1};
    if (lane < NB && lane < nb) rec = reinterpret_cast<const v4i*>(a.blocks)[first + lane];
    return rec;
}

// lanes < NB publish {cy, cx, cfl_zero, scaleFactor[c] / hfMultiplier} of their block (HFCoefficients.java:299)
template <int H, int W>
__device__ __forceinline__ void wave_publish(const WaveArgs& a, const v4i rec, int lane, float* __restrict__ brec) {
    using C = WCfg<H, W>;
    int* breci = reinterpret_cast<int*>(brec);
    if (lane < C::NB) {
        const float hf = (float)rec.w;
        breci[lane * 8 + 0] = (int)((uint32_t)rec.x & 0xffffu);
        breci[lane * 8 + 1] = (int)((uint32_t)rec.x >> 16);
        breci[lane * 8 + 2] = rec.z;
        brec[lane * 8 + 3] = a.f.scale_factor[0] / hf;
        brec[lane * 8 + 4] = a.f.scale_factor[1] / hf;
        brec[lane * 8 + 5] = a.f.scale_factor[2] / hf;
    }
    wave_fence();
}

// every load the item's arithmetic will need: quantised coefficients of all three channels, CfL factors, LF patch
template <int H, int W>
__device__ __forceinline__ void wave_request(const WaveArgs& a, int nb, int lane, const float* __restrict__ brec, WRaw<WCfg<H, W>::NG, WCfg<H, W>::DS>& raw) {
    using C = WCfg<H, W>;
    const DevFrame& f = a.f;
    const int* breci = reinterpret_cast<const int*>(brec);
    const int r = lane % C::GPB;  // (lane + 64 j) % GPB: the same for every group of the lane
    const int n = r / (W / 4), x4 = (r % (W / 4)) * 4;
#pragma unroll
    for (int j = 0; j < C::NG; j++) {
        const int b = (lane + 64 * j) / C::GPB;
        raw.gb[j] = b < nb ? b : -1;
        raw.kx[j] = raw.kb[j] = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 3; ci++) raw.q[j][ci] = v4i{0, 0, 0, 0};
        if (b < nb) {
            const int cy = breci[b * 8], cx = breci[b * 8 + 1];
            const uint32_t cfl_zero = (uint32_t)breci[b * 8 + 2];
            const int py = cy * 8 + n, px = cx * 8 + x4;
            const int64_t goff = coeff_off(f.width, py, px);
            raw.q[j][0] = *reinterpret_cast<const v4i*>(f.coeff[1] + goff);
            raw.q[j][1] = *reinterpret_cast<const v4i*>(f.coeff[0] + goff);
            raw.q[j][2] = *reinterpret_cast<const v4i*>(f.coeff[2] + goff);
            // chromaFromLuma factors of the 64x64 tile the group lies in, honouring the reference's per-group cache order
            // (DevBlock::cfl_zero; HFCoefficients.java:159-181)
            const int ty = py >> 6, tx = px >> 6;
            const int bit = (ty - ((cy * 8) >> 6)) * 5 + (tx - ((cx * 8) >> 6));
            if (!((cfl_zero >> bit) & 1u)) {
                raw.kx[j] = f.kx_tab[ty * f.tw + tx];
                raw.kb[j] = f.kb_tab[ty * f.tw + tx];
            }
        }
    }
    // finalizeLLF (HFCoefficients.java:194-229): lane t < NLLF owns coefficient (c, ky, kx) = t % (3 DS) of block t / (3 DS)
#pragma unroll
    for (int k = 0; k < C::DS; k++) raw.lfp[k] = 0.0f;
    if (lane < C::NLLF) {
        const int b = lane / (3 * C::DS), c = (lane % (3 * C::DS)) / C::DS;
        if (b < nb) {
            const int cy = breci[b * 8], cx = breci[b * 8 + 1];
            // (selects between opaque copies, not f.lf[c]: a lane-dependent index into the kernel-argument block -- which is also
            // what the optimiser makes of a select between three argument loads -- has the whole block copied to scratch and
            // every argument read from there)
            const float *l0 = f.lf[0], *l1 = f.lf[1], *l2 = f.lf[2];
            asm volatile("" : "+s"(l0), "+s"(l1), "+s"(l2));
            const float* lp = (c == 0 ? l0 : c == 1 ? l1 : l2) + (int64_t)cy * f.bw + cx;
#pragma unroll
            for (int y = 0; y < C::DSH; y++)
#pragma unroll
                for (int x = 0; x < C::DSW; x++) raw.lfp[y * C::DSW + x] = lp[(int64_t)y * f.bw + x];
        }
    }
}

// the weight rows of the lane's group position, processing order Y, X, B (TransformType.flip() for METHOD_DCT: tall or square)
template <int H, int W>
__device__ __forceinline__ void wave_weights(const WaveArgs& a, int type, int lane, v4f (&wt)[3]) {
    using C = WCfg<H, W>;
    const DevFrame& f = a.f;
    const int r = lane % C::GPB;
    const int PI = JXL_TT[type].param_index;
    const float* wtab = (H >= W ? f.weights_t : f.weights);
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        wt[ci] = *reinterpret_cast<const v4f*>(wtab + f.woffs[PI * 3 + c] + r * 4);
    }
}

template <int H, int W>
__device__ __forceinline__ void wave_compute(const WaveArgs& a, int nb, int lane, float* __restrict__ img, float* __restrict__ brec,
                                             const float* __restrict__ qtab, const v4f (&wt_in)[3], WRaw<WCfg<H, W>::NG, WCfg<H, W>::DS>& raw) {
    using C = WCfg<H, W>;
    const DevFrame& f = a.f;
    const int* breci = reinterpret_cast<const int*>(brec);
    float* llfb = brec + 64;
    const int r = lane % C::GPB;
    const int n = r / (W / 4), x4 = (r % (W / 4)) * 4;
    if (lane < C::NLLF) {
        const int k = lane % C::DS;
        llfb[lane] = llf_coeff_w<C::DSH, C::DSW>(raw.lfp, k / C::DSW, k % C::DSW);
    }
    v4f wt[3] = {wt_in[0], wt_in[1], wt_in[2]};
    // ---- channels Y, X, B through the single-channel image (slot 0 = the channel being processed; the slots rotate after each
    // channel so that the loop body exists once: a runtime channel index into register arrays would put them in scratch)
    float dy[C::NG][4];
    const float qbn = f.quant_bias_numerator;
#pragma unroll 1
    for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        // A. dequantise (+ chroma-from-luma) -> image. dequantizeHFCoefficients (:309-315) through the table: tab[a] = 0,
        // quantBias, (float)a - qbn / (float)a for a = 0, 1, 2..63, applied with the sign of q; one rare branch for |q| >= 64
        const float* qt = qtab + c * 64;
#pragma unroll
        for (int j = 0; j < C::NG; j++) {
            if (raw.gb[j] < 0) continue;
            float dq[4];
            int big = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qv = raw.q[j][0][i];
                const int aq = qv < 0 ? -qv : qv;
                big |= aq;
                const float m = qt[aq & 63];
                dq[i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, m) ^ ((uint32_t)qv & 0x80000000u));
            }
            if ((uint32_t)big >= 64u) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int qv = raw.q[j][0][i];
                    const int aq = qv < 0 ? -qv : qv;
                    if (aq >= 64) dq[i] = (float)qv - qbn / (float)qv;
                }
            }
            const float sf = brec[raw.gb[j] * 8 + 3 + c];
            const float kc = ci == 1 ? raw.kx[j] : raw.kb[j];
            float* d = img + raw.gb[j] * C::IMG + n * C::LD + x4;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float v = dq[i] * sf * wt[0][i];
                if (ci == 0) dy[j][i] = v;
                else v = v + kc * dy[j][i];  // chromaFromLuma (:186-188)
                d[i] = v;
            }
        }
        // the LLF corner is skipped by the dequantiser (:305-306) and overwritten by finalizeLLF
        if (lane < nb * C::DS) {
            const int b = lane / C::DS, k = lane % C::DS;
            img[b * C::IMG + (k / C::DSW) * C::LD + (k % C::DSW)] = llfb[(b * 3 + c) * C::DS + k];
        }
        wave_fence();
        // B. column pass, in place: lane = column
#pragma unroll
        for (int l = 0; l < C::CL; l++) {
            const int cidx = lane + 64 * l;
            float* p = img + (cidx / W) * C::IMG + (cidx % W);
            float acc[H];
            idct1d_lane<H>(acc, p, C::LD);
#pragma unroll
            for (int k = 0; k < H; k++) p[k * C::LD] = acc[k];
        }
        wave_fence();
        // C. row pass, in place: lane = row
#pragma unroll
        for (int l = 0; l < C::RL; l++) {
            const int ridx = lane + 64 * l;
            float* p = img + (ridx / H) * C::IMG + (ridx % H) * C::LD;
            float acc[W];
            idct1d_lane<W>(acc, p, 1);
#pragma unroll
            for (int k = 0; k < W; k++) p[k] = acc[k];
        }
        wave_fence();
        // D. image -> frame plane in the layout of the coefficient loads: a lane stores 16 bytes, consecutive lanes consecutive
        // pieces of a block row, then the next row -- a store instruction writes whole 128-byte lines wherever the item's blocks
        // are neighbours. (Stored straight from the row pass, lane = row, every instruction wrote 64 pieces of 16 bytes with
        // 16-byte gaps: measured 16 us of a 51 us DCT8 frame, against 5 us for the loads of the same bytes in this layout.)
        float* oc = c == 0 ? a.o0 : c == 1 ? a.o1 : a.o2;
#pragma unroll
        for (int j = 0; j < C::NG; j++) {
            if (raw.gb[j] < 0) continue;
            const int b = raw.gb[j];
            const float* sp = img + b * C::IMG + n * C::LD + x4;
            const float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
            const int cy = breci[b * 8], cx = breci[b * 8 + 1];
            *reinterpret_cast<float4*>(oc + (int64_t)(cy * 8 + n) * f.width + cx * 8 + x4) = v;
        }
        wave_fence();
        // next channel into slot 0
#pragma unroll
        for (int j = 0; j < C::NG; j++) {
            raw.q[j][0] = raw.q[j][1];
            raw.q[j][1] = raw.q[j][2];
        }
        wt[0] = wt[1];
        wt[1] = wt[2];
    }
}

// one item, nothing in flight across items (the 16-point class: 48 coefficient registers per item leave no room for a second set)
template <int H, int W>
__device__ __forceinline__ void wave_item(const WaveArgs& a, int type, int first, int nb, int lane, float* __restrict__ img,
                                          float* __restrict__ brec, const float* __restrict__ qtab) {
    using C = WCfg<H, W>;
    const v4i rec = wave_load_rec(a, first, nb, lane, C::NB);
    wave_publish<H, W>(a, rec, lane, brec);
    WRaw<C::NG, C::DS> raw;
    wave_request<H, W>(a, nb, lane, brec, raw);
    v4f wt[3];
    wave_weights<H, W>(a, type, lane, wt);
    wave_compute<H, W>(a, nb, lane, img, brec, qtab, wt, raw);
}

// ---- the 32-point types (32x32, 32x8, 8x32, 32x16, 16x32): one type-generic body (run-time geometry) so that the 32-point
// transform (1488 instructions with literal table entries) exists ONCE in the code object -- as a pass over lines of the image
// that serves the column pass and the row pass of all five types -- and the kernel stays inside the instruction cache.
// 2048 sample positions per item and channel = 8 groups per lane: the channels are loaded one at a time.
struct BigGeo {
    int H, W, lgH, lgW, NB, LD, IMG, DSH, DSW;
};
__device__ __forceinline__ BigGeo big_geo(int type) {
    switch (type) {
    case 5: return BigGeo{32, 32, 5, 5, 2, 33, 1056, 4, 4};
    case 8: return BigGeo{32, 8, 5, 3, 8, 9, 296, 4, 1};
    case 9: return BigGeo{8, 32, 3, 5, 8, 33, 264, 1, 4};
    case 10: return BigGeo{32, 16, 5, 4, 4, 17, 560, 4, 2};
    default: return BigGeo{16, 32, 4, 5, 4, 33, 528, 2, 4};  // 11
    }
}
constexpr float kL4[3][4] = {
    {0x1.4e7aea0000000p+0f, 0x1.1517a80000000p-1f, -0x1.1517a80000000p-1f, -0x1.4e7aea0000000p+0f},
    {0x1.0000000000000p+0f, -0x1.0000000000000p+0f, -0x1.0000000000000p+0f, 0x1.0000000000000p+0f},
    {0x1.1517a80000000p-1f, -0x1.4e7aea0000000p+0f, 0x1.4e7aea0000000p+0f, -0x1.1517a80000000p-1f},
};
// coefficient k of MathHelper.forwardDCTHorizontal (MathHelper.java:80-95) of v[0..len) BEFORE the division by len; len = 1, 2, 4
__device__ __forceinline__ float fdct_coeff(const float v[4], int len, int k) {
    if (len == 1) return v[0];
    float d;
    if (k == 0) {
        d = v[0] + v[1];
        if (len > 2) {
            d = d + v[2];
            d = d + v[3];
        }
    } else if (len == 2) {
        d = v[0] * 1.0f + v[1] * -1.0f;
    } else {
        float l[4];
#pragma unroll
        for (int n = 0; n < 4; n++) l[n] = k == 1 ? kL4[0][n] : k == 2 ? kL4[1][n] : kL4[2][n];
        d = v[0] * l[0];
        d = d + v[1] * l[1];
        d = d + v[2] * l[2];
        d = d + v[3] * l[3];
    }
    return d;
}
// LLFScale entry k << (5 - log2(len)) for len = 1, 2, 4 (HFCoefficients.java:207-214): indices 0, 8, 16, 24
__device__ __forceinline__ float llf_scale_of(int len, int k) {
    const int idx = len == 4 ? k * 8 : len == 2 ? k * 16 : 0;
    return idx == 0 ? kLlfScaleW[0] : idx == 8 ? kLlfScaleW[8] : idx == 16 ? kLlfScaleW[16] : kLlfScaleW[24];
}

__device__ __forceinline__ void wave_big(const WaveArgs& a, int type, int first, int nb, int lane, float* __restrict__ img,
                                         float* __restrict__ brec, const float* __restrict__ qtab) {
    const DevFrame& f = a.f;
    const BigGeo g = big_geo(type);
    const int lgGPB = g.lgH + g.lgW - 2, lgW4 = g.lgW - 2;
    const int GPBm = (1 << lgGPB) - 1, W4m = (1 << lgW4) - 1;
    const int DS = g.DSH * g.DSW;
    int* breci = reinterpret_cast<int*>(brec);
    // block records (as wave_publish, run-time NB)
    {
        const v4i rec = wave_load_rec(a, first, nb, lane, g.NB);
        if (lane < g.NB) {
            const float hf = (float)rec.w;
            breci[lane * 8 + 0] = (int)((uint32_t)rec.x & 0xffffu);
            breci[lane * 8 + 1] = (int)((uint32_t)rec.x >> 16);
            breci[lane * 8 + 2] = rec.z;
            brec[lane * 8 + 3] = f.scale_factor[0] / hf;
            brec[lane * 8 + 4] = f.scale_factor[1] / hf;
            brec[lane * 8 + 5] = f.scale_factor[2] / hf;
        }
        wave_fence();
    }
    // The per-channel pointers go through a small table in LDS, read with the run-time channel index: an index into the
    // kernel-argument block (also what the optimiser makes of selects between argument loads) would send the whole block to
    // scratch, and nine opaque scalar copies kept 24 SGPRs alive through the whole item (77 spilled).
    unsigned long long* ptab = reinterpret_cast<unsigned long long*>(brec + 64);  // 12 entries (the LLF slots of the small types)
    if (lane == 0) {
        const int PI = JXL_TT[type].param_index;
        const float* wtab = (g.H >= g.W ? f.weights_t : f.weights);
        ptab[0] = (unsigned long long)f.coeff[0]; ptab[1] = (unsigned long long)f.coeff[1]; ptab[2] = (unsigned long long)f.coeff[2];
        ptab[3] = (unsigned long long)f.lf[0]; ptab[4] = (unsigned long long)f.lf[1]; ptab[5] = (unsigned long long)f.lf[2];
        ptab[6] = (unsigned long long)a.o0; ptab[7] = (unsigned long long)a.o1; ptab[8] = (unsigned long long)a.o2;
        ptab[9] = (unsigned long long)(wtab + f.woffs[PI * 3]);
        ptab[10] = (unsigned long long)(wtab + f.woffs[PI * 3 + 1]);
        ptab[11] = (unsigned long long)(wtab + f.woffs[PI * 3 + 2]);
    }
    wave_fence();
    auto tab_ptr = [&](int i) {
        const unsigned long long v = ptab[i];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    const float qbn = f.quant_bias_numerator;
    const int n_ws = lgGPB > 6 ? 1 << (lgGPB - 6) : 1;  // distinct weight rows among a lane's groups (1, 2 or 4)
    float dy[8][4];
#pragma unroll 1
    for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        const int32_t* qp = reinterpret_cast<const int32_t*>(tab_ptr(c));
        const float* wp = reinterpret_cast<const float*>(tab_ptr(9 + c));
        // ---- requests of this channel
        v4i q[8];
        v4f wt[4];
        float kc[8];
        int gbm = 0;  // bit j: group j belongs to a block of the item
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < n_ws) wt[j] = *reinterpret_cast<const v4f*>(wp + ((lane + 64 * j) & GPBm) * 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int gi = lane + 64 * j;
            const int b = gi >> lgGPB, r = gi & GPBm;
            const int n = r >> lgW4, x4 = (r & W4m) << 2;
            q[j] = v4i{0, 0, 0, 0};
            kc[j] = 0.0f;
            if (b < nb) {
                gbm |= 1 << j;
                const int cy = breci[b * 8], cx = breci[b * 8 + 1];
                const int py = cy * 8 + n, px = cx * 8 + x4;
                q[j] = *reinterpret_cast<const v4i*>(qp + coeff_off(f.width, py, px));
                if (ci > 0) {
                    const uint32_t cfl_zero = (uint32_t)breci[b * 8 + 2];
                    const int ty = py >> 6, tx = px >> 6;
                    const int bit = (ty - ((cy * 8) >> 6)) * 5 + (tx - ((cx * 8) >> 6));
                    if (!((cfl_zero >> bit) & 1u)) kc[j] = (ci == 1 ? f.kx_tab : f.kb_tab)[ty * f.tw + tx];
                }
            }
        }
        // ---- A. dequantise (+ chroma-from-luma) -> image (as in wave_compute)
        const float* qt = qtab + c * 64;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!((gbm >> j) & 1)) continue;
            const int gi = lane + 64 * j;
            const int b = gi >> lgGPB, r = gi & GPBm;
            const int n = r >> lgW4, x4 = (r & W4m) << 2;
            float dq[4];
            int big = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qv = q[j][i];
                const int aq = qv < 0 ? -qv : qv;
                big |= aq;
                const float m = qt[aq & 63];
                dq[i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, m) ^ ((uint32_t)qv & 0x80000000u));
            }
            if ((uint32_t)big >= 64u) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int qv = q[j][i];
                    const int aq = qv < 0 ? -qv : qv;
                    if (aq >= 64) dq[i] = (float)qv - qbn / (float)qv;
                }
            }
            const float sf = brec[b * 8 + 3 + c];
            const v4f w = n_ws == 1 ? wt[0] : n_ws == 2 ? wt[j & 1] : wt[j & 3];
            float* d = img + b * g.IMG + n * g.LD + x4;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float v = dq[i] * sf * w[i];
                if (ci == 0) dy[j][i] = v;
                else v = v + kc[j] * dy[j][i];  // chromaFromLuma (:186-188)
                d[i] = v;
            }
        }
        // finalizeLLF of this channel: lane t < NB * DS (= 32) owns coefficient (ky, kx) = t % DS of block t / DS. The LF patch is
        // requested here, behind the dequantisation (its registers are free now; the LF planes are L2-resident)
        const int lb = lane / DS, lk = lane % DS, lky = lk / g.DSW, lkx = lk % g.DSW;
        const bool llf_lane = lane < g.NB * DS && lb < nb;
        if (llf_lane) {
            float patch[4][4];
            const float* lp = reinterpret_cast<const float*>(tab_ptr(3 + c)) + (int64_t)breci[lb * 8] * f.bw + breci[lb * 8 + 1];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) patch[y][x] = (y < g.DSH && x < g.DSW) ? lp[(int64_t)y * f.bw + x] : 0.0f;
            // forwardDCT2D (rows, then columns; MathHelper.java:124-136) of the patch, coefficient (lky, lkx), times llfScale
            float rr[4];
#pragma unroll
            for (int y = 0; y < 4; y++) rr[y] = fdct_coeff(patch[y], g.DSW, lkx) * (1.0f / (float)g.DSW);
            const float d2 = fdct_coeff(rr, g.DSH, lky);
            const float v = (d2 * (1.0f / (float)g.DSH)) * (llf_scale_of(g.DSH, lky) * llf_scale_of(g.DSW, lkx));
            img[lb * g.IMG + lky * g.LD + lkx] = v;
        }
        wave_fence();
        // ---- B, C. column pass, then row pass, both in place: one loop so that the transforms are instantiated once
#pragma unroll 1
        for (int ph = 0; ph < 2; ph++) {
            const int N = ph == 0 ? g.H : g.W;
            const int lgL = ph == 0 ? g.lgW : g.lgH;  // lines per block (columns in the column pass, rows in the row pass)
            const int nl = (g.NB << lgL) >> 6;
            const int pos_stride = ph == 0 ? 1 : g.LD, stride = ph == 0 ? g.LD : 1;
#pragma unroll 1
            for (int l = 0; l < nl; l++) {
                const int idx = lane + 64 * l;
                float* p = img + (idx >> lgL) * g.IMG + (idx & ((1 << lgL) - 1)) * pos_stride;
                if (N == 32) {
                    float acc[32];
                    idct1d_lane<32>(acc, p, stride);
#pragma unroll
                    for (int k = 0; k < 32; k++) p[k * stride] = acc[k];
                } else if (N == 16) {
                    float acc[16];
                    idct1d_lane<16>(acc, p, stride);
#pragma unroll
                    for (int k = 0; k < 16; k++) p[k * stride] = acc[k];
                } else {
                    float acc[8];
                    idct1d_lane<8>(acc, p, stride);
#pragma unroll
                    for (int k = 0; k < 8; k++) p[k * stride] = acc[k];
                }
            }
            wave_fence();
        }
        // ---- D. image -> frame plane, in the layout of the loads
        float* oc = reinterpret_cast<float*>(tab_ptr(6 + c));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!((gbm >> j) & 1)) continue;
            const int gi = lane + 64 * j;
            const int b = gi >> lgGPB, r = gi & GPBm;
            const int n = r >> lgW4, x4 = (r & W4m) << 2;
            const float* sp = img + b * g.IMG + n * g.LD + x4;
            const int cy = breci[b * 8], cx = breci[b * 8 + 1];
            *reinterpret_cast<float4*>(oc + (int64_t)(cy * 8 + n) * f.width + cx * 8 + x4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
        wave_fence();
    }
}

}  // namespace

constexpr int kWaveSmallImgFloats = 1344;  // the largest single-channel image of the 8/16-point types (16x8: 8 x 168)
constexpr int kWaveBigImgFloats = 2368;    // 32x8: 8 x 296

// One work item per 64-thread workgroup (= one wave): the hardware dispatcher hands the items out as wave slots come free, so
// these launches share the machine fairly with the kernels of the other types that run beside them (a persistent launch sized
// to fill the machine held every slot until its end and the others ran after it: measured). Two kernels by register class:
// the 8/16-point types (128 VGPRs, 4 waves per SIMD) and the 32-point types (the whole-line accumulators, eight coefficient
// groups and luma's samples for chroma-from-luma need ~210 VGPRs: 2 waves per SIMD, which is enough for a body of 3000
// multiply-adds per channel).
__device__ __forceinline__ void wave_qtab(const WaveArgs& a, int lane, float* qtab) {
    // dequantisation table [3][64]: 0, quantBias[c], (float)a - qbn / (float)a
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float qb = c == 0 ? a.f.quant_bias[0] : c == 1 ? a.f.quant_bias[1] : a.f.quant_bias[2];
        qtab[c * 64 + lane] = lane == 0 ? 0.0f : lane == 1 ? qb : (float)lane - a.f.quant_bias_numerator / (float)lane;
    }
    wave_fence();
}

#ifndef JXL_WAVE_OCC
#define JXL_WAVE_OCC 4
#endif
__global__ __launch_bounds__(64, JXL_WAVE_OCC) void k_idct_wave_small(const WaveArgs a) {
    __shared__ float lds[kWaveSmallImgFloats + kWaveRecFloats + 192];
    float* img = lds;
    float* brec = lds + kWaveSmallImgFloats;
    float* qtab = brec + kWaveRecFloats;
    const int lane = threadIdx.x;
    const cintp w = (cintp)a.items + 4 * (int)blockIdx.x;
    const int type = w[0], first = w[1], nb = w[2];
    if (type < 0) return;
    wave_qtab(a, lane, qtab);
    switch (type) {
    case 0: wave_item<8, 8>(a, type, first, nb, lane, img, brec, qtab); break;
    case 4: wave_item<16, 16>(a, type, first, nb, lane, img, brec, qtab); break;
    case 6: wave_item<16, 8>(a, type, first, nb, lane, img, brec, qtab); break;
    case 7: wave_item<8, 16>(a, type, first, nb, lane, img, brec, qtab); break;
    default: break;
    }
}

#ifndef JXL_WAVE_BIG_OCC
#define JXL_WAVE_BIG_OCC 2
#endif
__global__ __launch_bounds__(64, JXL_WAVE_BIG_OCC) void k_idct_wave_big(const WaveArgs a) {
    __shared__ float lds[kWaveBigImgFloats + kWaveRecFloats + 192];
    float* img = lds;
    float* brec = lds + kWaveBigImgFloats;
    float* qtab = brec + kWaveRecFloats;
    const int lane = threadIdx.x;
    const cintp w = (cintp)a.items + 4 * (int)blockIdx.x;
    const int type = w[0], first = w[1], nb = w[2];
    if (type < 0) return;
    wave_qtab(a, lane, qtab);
    wave_big(a, type, first, nb, lane, img, brec, qtab);
}

bool wave_handles(int type) {
    // r3 status: bit-exact on every parity test; alone on the device the 8/16-point class beats k_idct_wg3 per type (4K frame of
    // one type: DCT8 49-51 against 55 us, DCT16 58-61 / 76, 16x8 55-58 / 67, 8x16 54-56 / 67), the 32-point class loses (DCT32
    // 152 / 100: two waves per SIMD do not cover its three load phases per item), and in the batch of eight frames -- where only
    // machine time counts -- every combination loses to the software-pipelined persistent kernel (35.1-36.7 against 38.5-39.8
    // Gpx/s, same box, tools/r3_bench_ab.sh). Off unless JXL_IDCT_WAVE=1.
    static const int on = getenv("JXL_IDCT_WAVE") ? atoi(getenv("JXL_IDCT_WAVE")) : 0;
    static const unsigned mask = getenv("JXL_IDCT_WAVE_TYPES") ? (unsigned)strtoul(getenv("JXL_IDCT_WAVE_TYPES"), nullptr, 0) : 0xFF1u;  // types 0, 4..11
    if (!on || type < 0 || type >= 32) return false;
    return ((mask & 0xFF1u) >> type) & 1u;
}

int wave_blocks_per_item(int type) {
    const int h = JXL_TT[type].ph, w = JXL_TT[type].pw;
    return 64 / (h < w ? h : w);
}

// The item list {type, first block, blocks, 0} of launch class cls (0: 8/16-point types, 1: 32-point types): items in spatial order (sorted by the 256 x 256 group of their first
// block; the block lists are group-major, so an item's blocks are neighbours), dealt to the XCDs in runs: workgroup b (= item b)
// runs on XCD b % 8, so a run of neighbouring items is put on positions of ONE residue -- the items that share 128-byte lines
// meet in one L2.
int wave_class_of(int type) { return (type == 0 || type == 4 || type == 6 || type == 7) ? 0 : 1; }

void wave_item_table(const DevBlock* hb, int frame_bw, const IdctSegment* segs, int n_seg, int cls, std::vector<int>& out) {
    struct Rec { uint32_t key; int type, first, nb; };
    std::vector<Rec> recs;
    const int grs = std::max(1, (frame_bw + 31) >> 5);
    for (int i = 0; i < n_seg; i++) {
        if (segs[i].n_blocks <= 0 || !wave_handles(segs[i].type) || wave_class_of(segs[i].type) != cls) continue;
        const int nb = wave_blocks_per_item(segs[i].type);
        for (int o = 0; o < segs[i].n_blocks; o += nb) {
            const DevBlock& b0 = hb[segs[i].first_block + o];
            recs.push_back(Rec{(uint32_t)((b0.cy >> 5) * grs + (b0.cx >> 5)), segs[i].type, segs[i].first_block + o, std::min(nb, segs[i].n_blocks - o)});
        }
    }
    out.clear();
    if (recs.empty()) return;
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.key < y.key; });
    static const int run = getenv("JXL_WAVE_RUN") ? std::max(1, atoi(getenv("JXL_WAVE_RUN"))) : 48;
    std::vector<const Rec*> q[8];
    for (size_t i = 0; i < recs.size(); i++) q[(i / (size_t)run) % 8].push_back(&recs[i]);
    size_t longest = 0;
    for (auto& v : q) longest = std::max(longest, v.size());
    out.reserve(longest * 32);
    for (size_t i = 0; i < longest; i++)
        for (int x = 0; x < 8; x++) {
            if (i < q[x].size()) {
                const Rec& r = *q[x][i];
                out.push_back(r.type); out.push_back(r.first); out.push_back(r.nb); out.push_back(0);
            } else {
                out.push_back(-1); out.push_back(0); out.push_back(0); out.push_back(0);  // hole: the workgroup exits
            }
        }
}

void launch_idct_wave(const WaveArgs& a, int cls, hipStream_t s) {
    if (a.n_items <= 0) return;
    if (cls == 0) hipLaunchKernelGGL(k_idct_wave_small, dim3(a.n_items), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_idct_wave_big, dim3(a.n_items), dim3(64), 0, s, a);
}

}  // namespace jxl
