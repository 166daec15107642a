// VarDCT stage 1, METHOD_DCT varblocks of 16..64 points per side (and the 8-point rectangles): dequantisation +
// chroma-from-luma + LLF + inverse DCT of all THREE channels of a group of varblocks per workgroup, as a persistent,
// software-pipelined kernel.
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/vardct/HFCoefficients.java:140-319  bakeDequantizedCoeffs (dequant, CfL, finalizeLLF)
//   J/frame/group/PassGroup.java:229-233        METHOD_DCT branch of invertVarDCT
//   J/util/MathHelper.java:68-136               inverseDCT2D (columns, then rows), forwardDCT2D for the LLF corner
//
// Why this shape (measured with per-phase s_memtime stamps, round 2): in the per-channel kernels of k_idct.hip a 32x32 work
// item spent 73 % of its life loading and dequantising (25 k of 34 k cycles) and 21 % in the two transform passes; chroma
// items loaded and dequantised luma again; every sample fetched its weight and its CfL factor on its own. Here
//   * one work item = NB varblocks x 3 channels (2048 sample positions per channel, 4096 for the 64-point family): luma is
//     dequantised once and feeds chroma-from-luma of X and B in registers; coefficient rows, weights and block records
//     move as 16-byte loads (4 consecutive x per lane);
//   * the workgroup is persistent: it walks the frame's items (all types of its register class, costliest type first)
//     with stride gridDim, and everything item i+1 needs from memory -- block records one item earlier still, then
//     coefficients, weights, CfL factors, LLF coefficients -- is requested while item i is in its transform passes: the
//     HBM latency that dominated every item overlaps with arithmetic of the same wave. The prefetch is type-generic
//     (run-time geometry), so the pipeline runs across type boundaries; only the arithmetic is specialised per type;
//   * workgroup barriers order LDS traffic only (lds_barrier): __syncthreads() would wait for the prefetch;
//   * both passes work on ONE LDS image per channel in place (read column -> barrier -> write column): 3 channels of 2048
//     positions fit 27 KB;
//   * a lane produces 8 (16 for the 64-point family) outputs of its column / row with the mirrored-product trick of
//     k_idct.hip (lut[n-1][N-1-k] == (-1)^n lut[n-1][k] bit for bit), the three channels interleaved so that one scalar
//     load of the LUT slice feeds 18 (36) packed operations, four steps per loop trip with the next trip's LUT slices and
//     LDS samples requested ahead;
//   * finalizeLLF runs inside the item (r3): the lane that holds LLF coefficient (c, ky, kx) of a block prefetches LF sample
//     (c, y, x) of its dctSelect-sized patch, the lanes exchange the patches through a small LDS table and dequant() forms
//     forwardDCT2D x llfScale there (llf_coeff3). Until r3 a kernel of its own in front of the launch (k_llf_wg3, still
//     selectable with JXL_WG3_LLF_IN_ITEM=0): 9-13 us and a dependent launch on every frame's critical path.
// Bit-exactness: every sum keeps the reference's order, multiplies and adds are separate IEEE f32 operations.
#include "jxl_internal.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../../include/jxl_tables.h"

namespace jxl {

#include "idct_small.h"

namespace {

// Two adjacent outputs of a lane. JXL_WG3_SCALAR_MAC: a plain pair of floats instead of the 2-vector the packed instructions
// take (A/B build, r4): v_pk_mul_f32 reads its broadcast sample as a 64-bit register pair whose upper half is undefined, the
// register allocator parks any live value there -- e.g. a register of the next item's prefetch -- and the wait-count pass then
// makes the first multiply of the column pass wait for that load (s_waitcnt vmcnt(0)); the scalar form has no such false
// dependency but needs 15-20 % more cycles in the MAC loops.
#ifndef JXL_WG3_SCALAR_MAC
typedef float v2f __attribute__((ext_vector_type(2)));
// LUT pair j of a wave-uniform slice: ONE 8-byte scalar load into an aligned SGPR pair the packed multiply takes as it is
// (built from two scalar floats the pair goes through two v_mov per use: +700 instructions in the kernel, measured)
__device__ __forceinline__ v2f lut_pair(const __attribute__((address_space(4))) float* lr, int j) {
    return ((const __attribute__((address_space(4))) v2f*)lr)[j];
}
#else
struct v2f {
    float x, y;
};
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f lut_pair(const __attribute__((address_space(4))) float* lr, int j) { return v2f{lr[2 * j], lr[2 * j + 1]}; }
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cfloatp;
typedef const __attribute__((address_space(4))) v4i* cv4ip;

__constant__ float kLlfScale3[32] = JXL_LLF_SCALE_INIT;
// LDS behind the block images: [3][128] dequantisation table | [104] finalizeLLF tables | [192] LF patches | [3][256] scaleFactor[c] / m
constexpr int kWg3QTab = 3 * 128;  // dequantisation table: [3][128], entry q + 64 for q = -64 .. 63
constexpr int kWg3AuxFloats = kWg3QTab + 104 + 192;
constexpr int kWg3SfEntries = 256;

// Diagnostic build only (-DJXL_STAMPS): lane 0 of every workgroup records s_memtime at the phase boundaries of its first
// two items (rows 2*wg and 2*wg+1 of the stamp buffer)
#ifdef JXL_STAMPS
__device__ unsigned long long* g_stamps3 = nullptr;
// (r4) the stamps of items kStampItem and kStampItem + 1 (steady state, not the cold first items) are collected in LDS and written
// out when the workgroup ends: a stamp that stores to memory reads the buffer pointer with a vector load and waits
// vmcnt(0) -- for every load in flight, i.e. it MEASURED the prefetch latency into whichever phase it closed (r2-r3 stamps did)
constexpr int kStampItem = 3;
__shared__ unsigned long long g_stamp_lds[24];
#define STAMP3(i)                                                                                                        \
    do {                                                                                                                 \
        if (threadIdx.x == 0 && (unsigned)(it_no - kStampItem) < 2u) g_stamp_lds[(it_no - kStampItem) * 12 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// workgroup lifetime: entry time, exit time and items done in column 11 of the workgroup's two rows
#define STAMP3_LIFE(row, val)                                                                                            \
    do {                                                                                                                 \
        if (threadIdx.x == 0) g_stamp_lds[(row) * 12 + 11] = (val);                                                      \
    } while (0)
#define STAMP3_FLUSH()                                                                                                   \
    do {                                                                                                                 \
        if (g_stamps3 && threadIdx.x == 0)                                                                               \
            for (int q_ = 0; q_ < 24; q_++) g_stamps3[(size_t)blockIdx.x * 24 + q_] = g_stamp_lds[q_];                   \
    } while (0)
#else
#define STAMP3(i)
#define STAMP3_LIFE(row, val)
#define STAMP3_FLUSH()
#endif

// Workgroup barrier that orders LDS traffic only. __syncthreads() carries a workgroup-scope release fence over ALL address
// spaces, which the compiler lowers to s_waitcnt vmcnt(0): every global load in flight -- the prefetch of the next item --
// would be waited for at the next barrier and the software pipeline would collapse (measured: the first version of this
// kernel was slower than the unpipelined one). Nothing a barrier of this kernel separates goes through global memory.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int clog2(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}

// the two launch classes: everything up to 32 points (2048 sample positions per item and channel, 256 threads) and the
// 64-point family (4096 positions; 512 threads, so that a lane still holds 2 coefficient groups per channel and 8 outputs
// per pass: the 256-thread form needed 228 VGPRs -- 2 waves per SIMD and no room beside the other class's workgroups -- and
// its single items, 100 per 4K frame of the default mix, ran 33 us each)
#ifndef WG3_BIG_T
#define WG3_BIG_T 512
#endif
template <bool BIG>
struct Cls {
    static constexpr int T = BIG ? WG3_BIG_T : 256;   // threads per workgroup
    static constexpr int P = BIG ? 4096 : 2048;       // sample positions per item and channel
    static constexpr int NG = P / 4 / T;              // 4-sample groups per lane and channel
    // weight-row slots per lane: the groups of a lane sit at the same place of different blocks (one slot) unless a block
    // has more groups than the workgroup has lanes
    static constexpr int WS = BIG ? NG : 1;
};

// LD_ / IMG_ (r6): row stride and size of a block image when they are not the transform passes' own (0: LD = W + 1 and the bank-staggered
// IMG below) -- the special 8x8 types keep their blocks as 64 consecutive floats at an odd pitch of 65, so that the lane that transforms
// block-channel k in registers reads image k without bank conflicts.
template <int H, int W, int LD_ = 0, int IMG_ = 0>
struct Cfg {
    static constexpr int MAXD = H > W ? H : W;
    static constexpr bool BIG = MAXD > 32;
    static constexpr int T = Cls<BIG>::T;
    static constexpr int P = MAXD <= 32 ? 2048 : 4096;  // sample positions per channel and work item
    static constexpr int NB = P / (H * W);               // varblocks per work item
    static constexpr int LD = LD_ ? LD_ : W + 1;         // row stride of a block image (odd: rows hit different banks)
    static constexpr int IMG0 = H * LD;
    // consecutive blocks of a column-pass wave land on consecutive bank ranges: image size == W (mod 32) for W < 32
    static constexpr int IMG = IMG_ ? IMG_ : W >= 32 ? IMG0 : IMG0 + ((W - IMG0 % 32) + 32) % 32;
    static constexpr int GPB = H * W / 4;                // 4-sample groups per block
    static constexpr int NG = P / 4 / T;                 // groups per lane and channel
    static constexpr int WS = Cls<BIG>::WS;
    static constexpr int CH_COL = T / (NB * W), KC_COL = H / CH_COL;  // column pass: lane groups per column, outputs per lane
    static constexpr int CH_ROW = T / (NB * H), KC_ROW = W / CH_ROW;
    static constexpr int DSH = H / 8, DSW = W / 8;       // dctSelect size = LLF corner
    static constexpr int PER_B = 3 * DSH * DSW;          // LLF coefficients per block
    static constexpr int NLLF = NB * PER_B;              // ... per item (<= 192)
    static_assert(NB >= 1 && NB * W >= 64 && NB * H >= 64, "lane groups of a pass are whole waves");
    static_assert(KC_COL == 8 || KC_COL == 16, "8 or 16 outputs per lane");
    static_assert(KC_ROW == 8 || KC_ROW == 16, "8 or 16 outputs per lane");
    static_assert(NLLF <= T, "one LLF coefficient per lane");
};

// KC outputs of one 1-D IDCT for three channels at once: lo[j] = outputs (2j, 2j+1) of the lane's low run, hi[j] = their
// mirror images (see MirrorAcc in k_idct.hip)
template <int KC>
struct Acc3 {
    v2f lo[3][KC / 4], hi[3][KC / 4];
    __device__ __forceinline__ void init(float a, float b, float c) {
#pragma unroll
        for (int j = 0; j < KC / 4; j++) {
            lo[0][j] = hi[0][j] = v2f{a, a};
            lo[1][j] = hi[1][j] = v2f{b, b};
            lo[2][j] = hi[2][j] = v2f{c, c};
        }
    }
    // output i of channel ch: i < KC/2 counts up through the low run, i >= KC/2 continues through the mirrored run so
    // that get(KC-1-i) is the mirror of get(i)
    __device__ __forceinline__ float get(int ch, int i) const {
        if (i < KC / 2) return (i & 1) ? lo[ch][i / 2].y : lo[ch][i / 2].x;
        const int m = KC - 1 - i;
        return (m & 1) ? hi[ch][m / 2].y : hi[ch][m / 2].x;
    }
};

// U = 4 steps of the three-channel MAC loop: their LUT slices (scalar registers) and sample triples
template <int KC>
struct Blk {
    v2f l[4][KC / 4];
    float s[4][3];
};
template <int KC>
__device__ __forceinline__ void load_blk(Blk<KC>& b, cfloatp lut /* row n0-1, the lane group's slice */, int N, const float* p0,
                                         const float* p1, const float* p2, int stride) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const cfloatp lr = lut + u * N;  // (scalar loads: the slice is wave-uniform)
#pragma unroll
        for (int j = 0; j < KC / 4; j++) b.l[u][j] = lut_pair(lr, j);
        b.s[u][0] = p0[u * stride];
        b.s[u][1] = p1[u * stride];
        b.s[u][2] = p2[u * stride];
    }
}
// steps n0 .. n0+3 (n0 even): even steps add to the mirrored half, odd steps subtract
template <int KC>
__device__ __forceinline__ void mac_blk(Acc3<KC>& acc, const Blk<KC>& b) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const v2f s[3] = {v2f{b.s[u][0], b.s[u][0]}, v2f{b.s[u][1], b.s[u][1]}, v2f{b.s[u][2], b.s[u][2]}};
#pragma unroll
        for (int j = 0; j < KC / 4; j++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const v2f p = s[ch] * b.l[u][j];
                acc.lo[ch][j] = acc.lo[ch][j] + p;
                acc.hi[ch][j] = (u & 1) ? acc.hi[ch][j] - p : acc.hi[ch][j] + p;
            }
    }
}
// MathHelper.inverseDCTHorizontal (MathHelper.java:68-78) for the lane's KC outputs of three channels: dest = src[0], then
// n = 1 .. N-1 in order. Steps 1..3 first, then whole blocks of 4 with the NEXT block's LUT slices and samples requested
// before the current block's arithmetic: the scalar loads and the LDS reads share one counter (lgkmcnt) and scalar loads
// return out of order, so a wait for either is a wait for everything outstanding -- with one block in flight behind 72
// (144) packed operations that wait is already satisfied.
template <int KC, int N>
__device__ __forceinline__ void idct1d3(Acc3<KC>& acc, cfloatp lut /* row 0, lane group's slice */, const float* p0, const float* p1,
                                        const float* p2, int stride) {
    acc.init(p0[0], p1[0], p2[0]);
    {
        float s[3][3];
        v2f l[3][KC / 4];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const cfloatp lr = lut + u * N;
#pragma unroll
            for (int j = 0; j < KC / 4; j++) l[u][j] = lut_pair(lr, j);
            s[u][0] = p0[(u + 1) * stride];
            s[u][1] = p1[(u + 1) * stride];
            s[u][2] = p2[(u + 1) * stride];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const v2f sv[3] = {v2f{s[u][0], s[u][0]}, v2f{s[u][1], s[u][1]}, v2f{s[u][2], s[u][2]}};
#pragma unroll
            for (int j = 0; j < KC / 4; j++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const v2f p = sv[ch] * l[u][j];
                    acc.lo[ch][j] = acc.lo[ch][j] + p;
                    acc.hi[ch][j] = (u & 1) ? acc.hi[ch][j] + p : acc.hi[ch][j] - p;  // n = u + 1: odd n subtracts
                }
        }
    }
    if (N > 4) {
        Blk<KC> cur;
        load_blk<KC>(cur, lut + 3 * N, N, p0 + 4 * stride, p1 + 4 * stride, p2 + 4 * stride, stride);
#pragma unroll 1
        for (int n0 = 4; n0 < N; n0 += 4) {
            Blk<KC> nxt;
            if (n0 + 4 < N)
                load_blk<KC>(nxt, lut + (n0 + 3) * N, N, p0 + (n0 + 4) * stride, p1 + (n0 + 4) * stride, p2 + (n0 + 4) * stride, stride);
            mac_blk<KC>(acc, cur);
            cur = nxt;
        }
    }
}

// The same for ONE channel (r6: a 64x64 block as an item of the 256-thread launch is worked channel by channel, Item64 below): KC
// outputs per lane, the LUT slice of a step feeds KC / 4 packed multiplies instead of 3 KC / 4.
template <int KC>
struct Acc1 {
    v2f lo[KC / 4], hi[KC / 4];
    __device__ __forceinline__ void init(float a) {
#pragma unroll
        for (int j = 0; j < KC / 4; j++) lo[j] = hi[j] = v2f{a, a};
    }
    __device__ __forceinline__ float get(int i) const {
        if (i < KC / 2) return (i & 1) ? lo[i / 2].y : lo[i / 2].x;
        const int m = KC - 1 - i;
        return (m & 1) ? hi[m / 2].y : hi[m / 2].x;
    }
};
template <int KC>
struct Blk1 {
    v2f l[4][KC / 4];
    float s[4];
};
template <int KC>
__device__ __forceinline__ void load_blk1(Blk1<KC>& b, cfloatp lut /* row n0-1, the lane group's slice */, int N, const float* p0, int stride) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const cfloatp lr = lut + u * N;
#pragma unroll
        for (int j = 0; j < KC / 4; j++) b.l[u][j] = lut_pair(lr, j);
        b.s[u] = p0[u * stride];
    }
}
template <int KC>
__device__ __forceinline__ void mac_blk1(Acc1<KC>& acc, const Blk1<KC>& b) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const v2f s = v2f{b.s[u], b.s[u]};
#pragma unroll
        for (int j = 0; j < KC / 4; j++) {
            const v2f p = s * b.l[u][j];
            acc.lo[j] = acc.lo[j] + p;
            acc.hi[j] = (u & 1) ? acc.hi[j] - p : acc.hi[j] + p;  // steps n0 .. n0 + 3, n0 even: odd n subtracts
        }
    }
}
// MathHelper.inverseDCTHorizontal (MathHelper.java:68-78), the lane's KC outputs of one channel: dest = src[0], then n = 1 .. N-1 in order
template <int KC, int N>
__device__ __forceinline__ void idct1d1(Acc1<KC>& acc, cfloatp lut /* row 0, lane group's slice */, const float* p0, int stride) {
    acc.init(p0[0]);
    {
        float s[3];
        v2f l[3][KC / 4];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const cfloatp lr = lut + u * N;
#pragma unroll
            for (int j = 0; j < KC / 4; j++) l[u][j] = lut_pair(lr, j);
            s[u] = p0[(u + 1) * stride];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const v2f sv = v2f{s[u], s[u]};
#pragma unroll
            for (int j = 0; j < KC / 4; j++) {
                const v2f p = sv * l[u][j];
                acc.lo[j] = acc.lo[j] + p;
                acc.hi[j] = (u & 1) ? acc.hi[j] + p : acc.hi[j] - p;  // n = u + 1: odd n subtracts
            }
        }
    }
    Blk1<KC> cur;
    load_blk1<KC>(cur, lut + 3 * N, N, p0 + 4 * stride, stride);
#pragma unroll 1
    for (int n0 = 4; n0 < N; n0 += 4) {
        Blk1<KC> nxt;
        if (n0 + 4 < N) load_blk1<KC>(nxt, lut + (n0 + 3) * N, N, p0 + (n0 + 4) * stride, stride);
        mac_blk1<KC>(acc, cur);
        cur = nxt;
    }
}

// One LLF coefficient (ky, kx) of channel plane lfp (patch origin, stride bw): forwardDCT2D of the DSH x DSW LF patch
// (MathHelper.java:124-136: rows, then columns) times llfScale (HFCoefficients.java:194-229). Every lane recomputes the
// row-pass values it needs: same operations in the same order as the reference's shared scratch arrays.
template <int DSH, int DSW, typename ScaleTab>
__device__ __forceinline__ float llf_coeff3(const float* __restrict__ lut_all, const float* __restrict__ lfp, int bw, int ky, int kx, ScaleTab llf_scale) {
    const float* lutw = lut_all + lut_off(clog2(DSW));
    const float* luth = lut_all + lut_off(clog2(DSH));
    const float invw = 1.0f / (float)DSW, invh = 1.0f / (float)DSH;
    float r[DSH];
#pragma unroll
    for (int y = 0; y < DSH; y++) {
        const float* row = lfp + (int64_t)y * bw;
        float d2;
        if (kx == 0) {
            d2 = row[0];
#pragma unroll
            for (int x = 1; x < DSW; ++x) d2 = d2 + row[x];
        } else {
            const float* lut = lutw + (kx - 1) * DSW;
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
        const float* lut = luth + (ky - 1) * DSH;
        d2 = r[0] * lut[0];
#pragma unroll
        for (int n = 1; n < DSH; ++n) d2 = d2 + r[n] * lut[n];
    }
    constexpr int yll = DSH <= 1 ? 0 : DSH <= 2 ? 1 : DSH <= 4 ? 2 : 3;
    constexpr int xll = DSW <= 1 ? 0 : DSW <= 2 ? 1 : DSW <= 4 ? 2 : 3;
    return (d2 * invh) * (llf_scale[ky << (5 - yll)] * llf_scale[kx << (5 - xll)]);
}

// ---- work items ---------------------------------------------------------------------------------------------------
// uniform description of one item (all scalar)
struct Item {
    int type;      // TransformType.type; -1: none
    int first;     // first block record
    int nb;        // blocks of the item
    uint32_t geo;  // run-time geometry of the type for the type-generic prefetch (wg3_geo)
    int gi;        // its position in the launch's item list (the prefetch reads the weight offsets from the record itself)
};

// geometry word of a type: bits 0-2 log2(W / 4), 4-7 log2(H * W / 4) (4-sample groups per block), 8-10 log2 of the LLF
// coefficients per block and channel (H / 8 * W / 8), 12-13 log2(W / 8), 15 TransformType.flip() (tall or square), 16-23 the
// parameter index. The item list carries it (r3), so that the prefetch of an item starts from ONE scalar load instead of a
// chain of dependent table look-ups and five run-time integer divisions.
__host__ __device__ inline uint32_t wg3_geo(int type) {
    const int H = JXL_TT[type].ph, W = JXL_TT[type].pw;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
    // (TransformType.flip(): tall, or square AND METHOD_DCT -- the special 8x8 types are square and not flipped)
    return (uint32_t)lg(W / 4) | (uint32_t)lg(H * W / 4) << 4 | (uint32_t)lg((H / 8) * (W / 8)) << 8 | (uint32_t)lg(W / 8) << 12 |
           ((H > W || (H == W && JXL_TT[type].method == JXL_METHOD_DCT)) ? 1u << 15 : 0u) | (uint32_t)JXL_TT[type].param_index << 16;
}

// Item gi of the launch: ONE 32-byte record {type, first block, blocks, geometry word, weight offsets of the three channels, 0}
// of the list finalize_tables built (wg3_item_table) -- one s_load_dwordx8, nothing looked up behind it (r1-r3: a 16-byte
// record, the weight offsets through DevFrame::woffs behind it, and a walk over the launch's segments when no list was given).
// r6: the list may hold HOLES (type -2: wg3_item_table gives a workgroup that ends on a 64x64 block fewer items than the others, and a
// workgroup's list is the positions w, w + G, ... whatever G is): the first record at or behind position gi, in steps of `step`, that
// is not one; gi is left on it. type -1: none.
template <int P>
__device__ __forceinline__ Item item_of(const Wg3Args& a, int& gi, int step) {
    Item it{-1, 0, 0, 0u, 0};
    while (gi < a.total_items) {
        const auto* w = (const __attribute__((address_space(4))) int*)a.items + 8 * gi;
        const int type = w[0];
        if (type != -2) {
            it.type = type;
            it.first = w[1];
            it.nb = w[2];
            it.geo = (uint32_t)w[3];
            it.gi = gi;
            break;
        }
        gi += step;
    }
    return it;
}

// lane t's LLF coefficient of an item: block t / (3 A), channel and position from the remainder (A = LLF coefficients per
// block and channel = 1 << lgA, a power of two: only the division by 3 is one, by multiplication)
__device__ __forceinline__ int llf_lane_block(int tid, int lgA) { return (int)(((uint32_t)tid * 0xAAABu) >> 17) >> lgA; }

// what a lane holds of an item between its prefetch and its dequantisation
template <int NG, int WS>
struct Raw {
    v4i q[NG][3];          // quantised coefficients of 4 consecutive x, channels X, Y, B
    v4f w[WS][3];          // their reciprocal weights ((flip ? transposed : plain) table row); with 2048 positions per
                           // item a block has at most 256 groups, so both groups of a lane sit at the same place of their blocks
    float kx[NG], kb[NG];  // CfL factors of the group's 64x64 tile (0 where the reference's cache reads 0)
    float hfm[NG];         // (float)hfMultiplier of the group's block
    float llf;             // lanes < NLLF: one LF sample of the item's blocks (llf_in_item; else the LLF coefficient from the llf planes)
    int rowx;              // DevBlock word 0 (cy | cx << 16) of the block whose row this lane stores in the row pass
    uint32_t ok;           // bit j: group j belongs to a block of the item
};

// block records the prefetch of an item will need: its groups' blocks and (lanes < NLLF) the block of its LLF coefficient
template <int NG>
struct Recs {
    int gx[NG], gz[NG], gw[NG];  // DevBlock words 0 (cy | cx << 16), 2 (cfl_zero), 3 (hf_mul) of the groups' blocks
    int lx;                      // word 0 of the LLF lane's block
};

template <int T, int NG>
__device__ __forceinline__ void load_recs(const Wg3Args& a, const Item& it, int tid, Recs<NG>& rc) {
    const int lgGPB = (int)((it.geo >> 4) & 15u), lgA = (int)((it.geo >> 8) & 7u);
#pragma unroll
    for (int j = 0; j < NG; j++) {
        rc.gx[j] = rc.gz[j] = 0;
        rc.gw[j] = 1;
        const int b = (tid + T * j) >> lgGPB;
        if (it.type >= 0 && b < it.nb) {
            // three dword loads straight into the three loop-carried registers: one 16-byte load would land in a temporary
            // and have to be COPIED into them, i.e. waited for right here (a full L2 round trip per item, measured)
            const auto* w = (const __attribute__((address_space(4))) int*)a.blocks + 4 * (it.first + b);
            rc.gx[j] = w[0];
            rc.gz[j] = w[2];
            rc.gw[j] = w[3];
        }
    }
    rc.lx = 0;
    const int bl = llf_lane_block(tid, lgA);
    if (it.type >= 0 && bl < it.nb) rc.lx = ((const __attribute__((address_space(4))) int*)a.blocks)[4 * (it.first + bl)];
}

// a wave-uniform pointer to global memory, held in SGPRs and opaque to the optimiser from here on (the address space is
// kept: a generic pointer would turn the loads into flat_load, which also counts on lgkmcnt -- the LDS waits of the passes
// would then wait for the prefetch)
template <typename P>
__device__ __forceinline__ const __attribute__((address_space(1))) P* sgpr_ptr(const P* p) {
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (const __attribute__((address_space(1))) P*)u;
}

// issue every load of item `it` this lane will need at dequantisation time (type-generic: run-time geometry).
// r4: nothing in here may WAIT for memory. The r3 form indexed the kernel-argument block with the lane's channel to pick the
// LF plane of its LLF sample -- a per-lane pointer load followed by s_waitcnt vmcnt(0). The plane pointers are now scalar
// arguments selected with v_cndmask, and the argument words the requests need are pinned in SGPRs.
template <int T, int NG, int WS>
__device__ __forceinline__ void prefetch(const Wg3Args& a, const Item& it, int tid, const Recs<NG>& rc, Raw<NG, WS>& raw) {
    const DevFrame& f = a.f;
    raw.ok = 0;
    raw.llf = 0.0f;
    const int lgW4 = (int)(it.geo & 7u), lgGPB = (int)((it.geo >> 4) & 15u);
    const float* wtab = ((it.geo >> 15) & 1u) ? f.weights_t : f.weights;  // TransformType.flip() for METHOD_DCT: tall or square
    // (words 4-6 of the item's record: read here, where they are used, instead of carried in SGPRs through two items)
    const auto* irec = (const __attribute__((address_space(4))) int*)a.items + 8 * it.gi;
    const float* wt[3] = {wtab + irec[4], wtab + irec[5], wtab + irec[6]};
    const __attribute__((address_space(1))) int32_t* cp[3] = {sgpr_ptr(f.coeff[0]), sgpr_ptr(f.coeff[1]), sgpr_ptr(f.coeff[2])};
    const auto* kxt = sgpr_ptr(f.kx_tab);
    const auto* kbt = sgpr_ptr(f.kb_tab);
    const __attribute__((address_space(1))) float* lfp[3] = {sgpr_ptr(a.llf_in_item ? f.lf[0] : f.llf[0]), sgpr_ptr(a.llf_in_item ? f.lf[1] : f.llf[1]),
                                                            sgpr_ptr(a.llf_in_item ? f.lf[2] : f.llf[2])};
    const int cells_w = f.width >> 3, tw = f.tw, bw = f.bw;
#pragma unroll
    for (int j = 0; j < NG; j++) {
        const int g = tid + T * j;
        const int b = g >> lgGPB, r = g & ((1 << lgGPB) - 1);
        const int n = r >> lgW4, x4 = (r & ((1 << lgW4) - 1)) << 2;
        // (256-thread class: raw.q / raw.hfm of a group outside the item keep whatever they held -- dequant() skips such groups,
        // Raw::ok -- which saves 26 v_mov per item; in the 512-thread class the re-definition ends their live ranges, without it
        // the kernel spills 12 registers instead of 3)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (WS > 1) raw.q[j][c] = v4i{0, 0, 0, 0};
            if (WS > 1 || j == 0) raw.w[WS == 1 ? 0 : j][c] = *reinterpret_cast<const v4f*>(wt[c] + r * 4);  // row-major [n][x]: n * W + x4 = 4 r
        }
        raw.kx[j] = raw.kb[j] = 0.0f;
        if (WS > 1) raw.hfm[j] = 1.0f;
        if (it.type >= 0 && b < it.nb && !((it.geo >> 14) & 1u)) {  // (bit 14: the item fetches its coefficients itself -- Item64)
            const int cy = (int)((uint32_t)rc.gx[j] & 0xffffu), cx = (int)((uint32_t)rc.gx[j] >> 16);  // DevBlock
            const uint32_t cfl_zero = (uint32_t)rc.gz[j];
            raw.hfm[j] = (float)rc.gw[j];
            const int py = cy * 8 + n, px = cx * 8 + x4;
            // cell-tiled int32 planes (coeff_off): cell (py >> 3, px >> 3), 64 samples each
            const int64_t off = (((int64_t)(py >> 3) * cells_w + (px >> 3)) << 6) + (((py & 7) << 3) | (px & 7));
#pragma unroll
            // (r6: non-temporal -- a frame's coefficients are read exactly once; leaving them out of the caches' keep-lists is worth 1-2 %
            // of the saturated stage and 1.5 % of the batch. The same frame's stage re-run alone, whose 99.5 MB of coefficients would
            // otherwise come back out of the 256 MB memory-side cache, is 8 % slower for it: profiles/experiments/README.md)
            for (int c = 0; c < 3; c++) raw.q[j][c] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) v4i*>(cp[c] + off));
            // chromaFromLuma factors of the tile this group lies in (4 consecutive x from a multiple of 4 never cross a
            // 64-px boundary), honouring the reference's per-group cache order (DevBlock::cfl_zero)
            const int ty = py >> 6, tx = px >> 6;
            const int bit = (ty - ((cy * 8) >> 6)) * 5 + (tx - ((cx * 8) >> 6));
            if (!((cfl_zero >> bit) & 1u)) {
                raw.kx[j] = kxt[ty * tw + tx];
                raw.kb[j] = kbt[ty * tw + tx];
            }
            raw.ok |= 1u << j;
        }
    }
    // finalizeLLF input: lane t < nb * PER_B holds LF sample (c, y, x) = t % PER_B of block t / PER_B (for an 8x8 block that IS
    // its LLF coefficient); the lanes exchange them through the LDS patch table and dequant() transforms them (r3: this was a
    // kernel of its own in front of the launch, k_llf_wg3 -- 9 us + a dependent launch on every frame's critical path)
    const int lgA = (int)((it.geo >> 8) & 7u), lgDSW = (int)((it.geo >> 12) & 3u);
    const int bl = llf_lane_block(tid, lgA);
    if (it.type >= 0 && bl < it.nb) {
        const int rr = tid - ((bl * 3) << lgA), c = rr >> lgA, k = rr & ((1 << lgA) - 1);
        const int ky = k >> lgDSW, kx = k & ((1 << lgDSW) - 1);
        const int cy = (int)((uint32_t)rc.lx & 0xffffu), cx = (int)((uint32_t)rc.lx >> 16);
        const auto* lp = c == 0 ? lfp[0] : c == 1 ? lfp[1] : lfp[2];  // (selects, not a per-lane load of the pointer)
        raw.llf = lp[(int64_t)(cy + ky) * bw + cx + kx];
    }
    // the block of the row this lane will store in the row pass (lane -> row ridx = tid % (NB * H), block ridx / H): r1-r3 loaded
    // the record at the start of the passes, and the compiler waited for it -- vmcnt(0), i.e. for this whole prefetch -- before
    // the column pass had begun. It travels with the prefetch now and is covered by the same wait in front of the stores.
    {
        constexpr int P = T * NG * 4;
        const int lgW = lgW4 + 2, lgH = lgGPB - lgW4;
        const int rb = (tid & ((P >> lgW) - 1)) >> lgH;
        raw.rowx = 0;
        if (it.type >= 0 && rb < it.nb) raw.rowx = sgpr_ptr(reinterpret_cast<const int*>(a.blocks))[4 * (it.first + rb)];
    }
}

template <int H, int W, int TYPE, int LD_ = 0, int IMG_ = 0>
struct Body {
    using C = Cfg<H, W, LD_, IMG_>;

    // ---- A. dequantise + chroma-from-luma -> LDS
    static __device__ __forceinline__ void dequant(const Wg3Args& a, const Item& it, int tid, const Raw<C::NG, C::WS>& raw, float* __restrict__ img,
                                                   const float* __restrict__ qtab) {
        const DevFrame& f = a.f;
        const float qbn = f.quant_bias_numerator;
#pragma unroll
        for (int j = 0; j < C::NG; j++) {
            if (!((raw.ok >> j) & 1u)) continue;
            const int g = tid + C::T * j;
            const int b = g / C::GPB, r = g % C::GPB;
            const int n = r / (W / 4), x4 = (r % (W / 4)) * 4;
            // scaleFactor[c] / hfMultiplier (HFCoefficients.java:299): the same IEEE quotient, formed once per workgroup and
            // multiplier value (sf_tab, r4) instead of three divisions per group of four samples
            float sf[3];
            {
                const int hi = (int)raw.hfm[j];
                const float* sft = qtab + kWg3AuxFloats;
#ifdef JXL_WG3_SF_DIV  // A/B build: the three divisions per group (r1-r3)
                if (false) {
#else
                if (__builtin_expect((unsigned)hi < (unsigned)kWg3SfEntries, 1)) {
#endif
                    sf[0] = sft[hi];
                    sf[1] = sft[kWg3SfEntries + hi];
                    sf[2] = sft[2 * kWg3SfEntries + hi];
                } else {
                    sf[0] = f.scale_factor[0] / raw.hfm[j];
                    sf[1] = f.scale_factor[1] / raw.hfm[j];
                    sf[2] = f.scale_factor[2] / raw.hfm[j];
                }
            }
            // (element access by constant index after unrolling: copying the vectors into local arrays made the compiler load
            // them as one <12 x float> and keep the whole prefetch state in scratch memory)
#define QV(c, i) (raw.q[j][c][i])
#define WV(c, i) (raw.w[C::WS == 1 ? 0 : j][c][i])
            // HFCoefficients.dequantizeHFCoefficients inner expression (:309-315) through a table indexed by the SIGNED value
            // (r4): tab[q + 64] = 0, +-quantBias, (float)q - qbn / (float)q for q = -64 .. 63 -- the reference's own expression
            // per entry, so neither |q| nor the sign has to be formed per sample (r3: tab[|q|] and the sign by XOR: two more
            // instructions per sample, 48 per lane and item). One branch per group for the rare q outside the table instead
            // of one per sample.
            float dq[3][4];
            uint32_t big = 0;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t t = (uint32_t)QV(c, i) + 64u;
                    big |= t;
                    dq[c][i] = qtab[c * 128 + (t & 127u)];
                }
            if (big >= 128u) {
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int qv = QV(c, i);
                        if ((uint32_t)qv + 64u >= 128u) dq[c][i] = (float)qv - qbn / (float)qv;
                    }
            }
            float* d0 = img + (0 * C::NB + b) * C::IMG + n * C::LD + x4;
            float* d1 = img + (1 * C::NB + b) * C::IMG + n * C::LD + x4;
            float* d2 = img + (2 * C::NB + b) * C::IMG + n * C::LD + x4;
            // the LLF corner (dctSelect size) is skipped by the dequantiser (:305-306), reads 0 in chromaFromLuma and is
            // overwritten by finalizeLLF (:194-229): those samples are written by the LLF lanes below
            const bool corner_row = n < C::DSH;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float dy = dq[1][i] * sf[1] * WV(1, i);
                const float dx = dq[0][i] * sf[0] * WV(0, i) + raw.kx[j] * dy;  // chromaFromLuma (:186-188)
                const float db = dq[2][i] * sf[2] * WV(2, i) + raw.kb[j] * dy;
                if (i < C::DSW && corner_row && x4 + i < C::DSW) continue;
                d0[i] = dx;
                d1[i] = dy;
                d2[i] = db;
            }
#undef QV
#undef WV
        }
        if (tid < C::NLLF) {
            const int b = tid / C::PER_B, rr = tid % C::PER_B;
            if (b < it.nb) {
                const int c = rr / (C::DSH * C::DSW), k = rr % (C::DSH * C::DSW);
                float v = raw.llf;
                if (C::DSH * C::DSW > 1 && a.llf_in_item) {
                    // forwardDCT2D of the block's LF patch, which the item's lanes published in lf_patch[] one barrier ago
                    const float* aux = qtab + kWg3QTab;  // [70] cosine LUT of 2, 4, 8 points | [2] | [32] LLF scale | [192] LF patches
                    v = llf_coeff3<C::DSH, C::DSW>(aux, aux + 104 + b * C::PER_B + c * (C::DSH * C::DSW), C::DSW, k / C::DSW, k % C::DSW, aux + 72);
                }
                img[(c * C::NB + b) * C::IMG + (k / C::DSW) * C::LD + (k % C::DSW)] = v;
            }
        }
    }

    // ---- B + C. column pass (in place), row pass -> frame planes
    template <typename PreStore>
    static __device__ __forceinline__ void passes(const Wg3Args& a, const Item& it, int tid, float* __restrict__ img, int it_no, int rowx,
                                                  PreStore pre_store) {
        (void)it_no;
        const DevFrame& f = a.f;
        const cfloatp lut_h = (cfloatp)(f.lut + lut_off(clog2(H)));
        const cfloatp lut_w = (cfloatp)(f.lut + lut_off(clog2(W)));
        const int cidx = tid % (C::NB * W), kc_col = __builtin_amdgcn_readfirstlane(tid / (C::NB * W));
        const int cb = cidx / W, cxx = cidx % W;
        const int ridx = tid % (C::NB * H), kc_row = __builtin_amdgcn_readfirstlane(tid / (C::NB * H));
        const int rb = ridx / H, ry = ridx % H;
        {
            constexpr int KC = C::KC_COL;
            Acc3<KC> acc;
            float* d0 = img + (0 * C::NB + cb) * C::IMG + cxx;
            float* d1 = img + (1 * C::NB + cb) * C::IMG + cxx;
            float* d2 = img + (2 * C::NB + cb) * C::IMG + cxx;
            idct1d3<KC, H>(acc, lut_h + kc_col * (KC / 2), d0, d1, d2, C::LD);
            STAMP3(4);
            lds_barrier();
            STAMP3(5);
            float* d[3] = {d0, d1, d2};
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int i = 0; i < KC / 2; i++) {
                    d[ch][(kc_col * (KC / 2) + i) * C::LD] = acc.get(ch, i);
                    d[ch][(H - 1 - kc_col * (KC / 2) - i) * C::LD] = acc.get(ch, KC - 1 - i);
                }
        }
        lds_barrier();
        STAMP3(6);
        {
            constexpr int KC = C::KC_ROW;
            Acc3<KC> acc;
            const float* r0 = img + (0 * C::NB + rb) * C::IMG + ry * C::LD;
            const float* r1 = img + (1 * C::NB + rb) * C::IMG + ry * C::LD;
            const float* r2 = img + (2 * C::NB + rb) * C::IMG + ry * C::LD;
            idct1d3<KC, W>(acc, lut_w + kc_row * (KC / 2), r0, r1, r2, 1);
            pre_store();
            STAMP3(7);
            if (rb < it.nb) {
                float* o3[3] = {a.o0, a.o1, a.o2};
                const int cy = (int)((uint32_t)rowx & 0xffffu), cx = (int)((uint32_t)rowx >> 16);  // (prefetched with the item: Raw::rowx)
                const int64_t off = (int64_t)(cy * 8 + ry) * f.width + cx * 8;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    // two runs of KC/2 consecutive outputs: the low one ascending, the mirrored one ending at W-1-kc*KC/2
                    float* olo = o3[ch] + off + kc_row * (KC / 2);
                    float* ohi = o3[ch] + off + W - (kc_row + 1) * (KC / 2);
#pragma unroll
                    for (int kk = 0; kk < KC / 2; kk += 4) {
                        *reinterpret_cast<float4*>(olo + kk) = make_float4(acc.get(ch, kk), acc.get(ch, kk + 1), acc.get(ch, kk + 2), acc.get(ch, kk + 3));
                        *reinterpret_cast<float4*>(ohi + kk) = make_float4(acc.get(ch, KC / 2 + kk), acc.get(ch, KC / 2 + kk + 1),
                                                                           acc.get(ch, KC / 2 + kk + 2), acc.get(ch, KC / 2 + kk + 3));
                    }
                }
            }
        }
        STAMP3(8);
    }
};

// ---- r6: the 8x8-footprint types that are not METHOD_DCT -- Hornuss, DCT2, DCT4, DCT4x8, DCT8x4, AFV0-3 (PassGroup.java:88-168, 234-325)
// -- as items of this launch (32 blocks x 3 channels, the geometry of an 8x8 DCT item: same requests, same dequantisation, same store
// mapping; only the weights are the untransposed ones, TransformType.flip() being false for them). Until r6 a launch of their own on a
// side stream (k_idct_special_wg): 17 us alone for 12 % of the pixels of the default mix, and in the saturated regime -- eight frames in
// flight, where `value` is measured -- a second launch per frame costs the stage 11-19 us per frame however the streams are mapped
// (profiles/experiments/r6_idct_saturated_mixes_r5kernels.txt: 88 % DCT8 + 12 % AFV0 61 us per frame, either type alone 40-42).
// Transform: lane (channel, block) holds the block's 64 pixels in registers and reads its coefficients from the image as it goes
// (idct_small.h); 96 of 256 lanes work in that phase.
using SpecialBody = Body<8, 8, 0, 8, 65>;
__host__ __device__ constexpr bool wg3_is_special(int type) { return (type >= 1 && type <= 3) || (type >= 12 && type <= 17); }

// The transform of one block of one channel, image at lds[off .. off + 64), in place. NOT inlined on purpose: its 64 + 64 registers sit
// on top of whatever the item loop keeps alive; inlined, the allocator answered with 112 spilled registers whose reloads landed in
// EVERY type's passes (9 scratch accesses per body); as a call, the registers the callee has to preserve are saved and restored in its
// own prologue and epilogue -- paid by the special items only. `off` instead of a pointer keeps the LDS address space (a generic
// pointer would turn the accesses into flat_* ones).
__device__ __attribute__((noinline)) void special_transform(int off, int type) {
    extern __shared__ float lds[];
    float* im = lds + off;
    float px[64];
    switch (type) {
    case 1: invert_small<1>(im, px); break;
    case 2: invert_small<2>(im, px); break;
    case 3: invert_small<3>(im, px); break;
    case 12: invert_small<12>(im, px); break;
    case 13: invert_small<13>(im, px); break;
    case 14: invert_small<14>(im, px); break;
    case 15: invert_small<15>(im, px); break;
    case 16: invert_small<16>(im, px); break;
    case 17: invert_small<17>(im, px); break;
    default: break;
    }
#pragma unroll
    for (int i = 0; i < 64; i++) im[i] = px[i];
}

// the transform phase of a special item: between the item's dequantisation and the requests of the next item (wg3_body)
__device__ __forceinline__ void special_phase(const Item& it, int tid) {
    using C = SpecialBody::C;
    if (tid < 3 * C::NB && (tid & (C::NB - 1)) < it.nb) special_transform(tid * C::IMG, it.type);  // image index (channel * NB + block) == lane
    lds_barrier();
}

// ... and its stores: lane -> (block, row), as in the row pass of an 8x8 DCT item (Raw::rowx is that block's record)
template <typename PreStore>
__device__ __forceinline__ void special_store(const Wg3Args& a, const Item& it, int tid, const float* __restrict__ img, int rowx, PreStore pre_store) {
    using C = SpecialBody::C;
    const int rb = tid >> 3, ry = tid & 7;
    // The next item's requests went out between the transform and here (not before the transform: their 45 registers would sit on
    // top of the block's 64 + 64), i.e. ahead of these stores: the wait for them below (vmcnt counts in order) does not wait for the
    // stores, and what is exposed of their latency is paid once per special item. wg3_item_table puts the special items last.
    if (rb < it.nb) {
        const DevFrame& f = a.f;
        float* o3[3] = {a.o0, a.o1, a.o2};
        const int cy = (int)((uint32_t)rowx & 0xffffu), cx = (int)((uint32_t)rowx >> 16);
        const int64_t off = (int64_t)(cy * 8 + ry) * f.width + cx * 8;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float* sp = img + (ch * C::NB + rb) * C::IMG + ry * 8;
            *reinterpret_cast<float4*>(o3[ch] + off) = make_float4(sp[0], sp[1], sp[2], sp[3]);
            *reinterpret_cast<float4*>(o3[ch] + off + 4) = make_float4(sp[4], sp[5], sp[6], sp[7]);
        }
    }
    pre_store();  // the next item's loads have landed; its LF patches are published (the barrier at the end of the item loop follows)
}

// ---- r6: a 64x64 DCT block (type 18) as an item of the 256-thread launch -- ONE block, worked channel by channel (Y, X, B) on one
// 64 x 65 LDS image: 16 samples per lane and channel, column pass and row pass with 16 outputs per lane (idct1d1). Until r6 such blocks
// were a launch of their own (512 threads, 50 KB of LDS, one round of single items = one item's latency: 29 us alone), and in the
// regime `value` is measured in a second launch per frame costs the stage ~10 us per frame for 5 % of the pixels
// (profiles/experiments/r6_wg3_special_items_and_dynamic_ab.txt). Nothing of it is prefetched (the item list puts these items last, behind
// the special 8x8 ones: wg3_item_table); what the generic prefetch delivers for it is the LF patch of its LLF corner (lf_patch) and
// its block record word (Raw::rowx). Luma is dequantised first and kept in registers for the chroma-from-luma of X and B. Same
// operations in the same order as Body<64, 64, 18>::dequant / passes of the 512-thread class (HFCoefficients.java:267-319, 146-229;
// MathHelper.java:96-122).
struct Item64 {
    static constexpr int LD = 65, KC = 16;

    // column pass of the channel in the image -> registers -> back in place
    static __device__ __forceinline__ void column_pass(const DevFrame& f, int tid, float* __restrict__ img) {
        const cfloatp lut = (cfloatp)(f.lut + lut_off(6));
        const int col = tid & 63, kc = __builtin_amdgcn_readfirstlane(tid >> 6);
        Acc1<KC> acc;
        idct1d1<KC, 64>(acc, lut + kc * (KC / 2), img + col, LD);
        lds_barrier();
#pragma unroll
        for (int i = 0; i < KC / 2; i++) {
            img[(kc * (KC / 2) + i) * LD + col] = acc.get(i);
            img[(63 - kc * (KC / 2) - i) * LD + col] = acc.get(KC - 1 - i);
        }
        lds_barrier();
    }
    // row pass of the channel in the image -> plane `out`
    template <typename PreStore>
    static __device__ __forceinline__ void row_pass(const DevFrame& f, int tid, const float* __restrict__ img, float* __restrict__ out, int cy, int cx,
                                                    PreStore pre_store) {
        const cfloatp lut = (cfloatp)(f.lut + lut_off(6));
        const int row = tid & 63, kc = __builtin_amdgcn_readfirstlane(tid >> 6);
        Acc1<KC> acc;
        idct1d1<KC, 64>(acc, lut + kc * (KC / 2), img + row * LD, 1);
        pre_store();
        float* olo = out + (int64_t)(cy * 8 + row) * f.width + cx * 8 + kc * (KC / 2);
        float* ohi = out + (int64_t)(cy * 8 + row) * f.width + cx * 8 + 64 - (kc + 1) * (KC / 2);
#pragma unroll
        for (int kk = 0; kk < KC / 2; kk += 4) {
            *reinterpret_cast<float4*>(olo + kk) = make_float4(acc.get(kk), acc.get(kk + 1), acc.get(kk + 2), acc.get(kk + 3));
            *reinterpret_cast<float4*>(ohi + kk) = make_float4(acc.get(KC / 2 + kk), acc.get(KC / 2 + kk + 1), acc.get(KC / 2 + kk + 2), acc.get(KC / 2 + kk + 3));
        }
    }

    // Y and X completely, B up to and including its column pass (its row pass follows the next item's requests: back())
    static __device__ __forceinline__ void front(const Wg3Args& a, const Item& it, int tid, float* __restrict__ img, const float* __restrict__ qtab, int rowx) {
        const DevFrame& f = a.f;
        const float qbn = f.quant_bias_numerator;
        const auto* brec = (const __attribute__((address_space(4))) int*)a.blocks + 4 * it.first;  // DevBlock {cy | cx << 16, .., cfl_zero, hf_mul}
        const auto* irec = (const __attribute__((address_space(4))) int*)a.items + 8 * it.gi;
        const int cy = (int)((uint32_t)brec[0] & 0xffffu), cx = (int)((uint32_t)brec[0] >> 16);
        (void)rowx;
        const uint32_t cfl_zero = (uint32_t)brec[2];
        const int hfmul = brec[3];
        float sf[3];
        {
            const float* sft = qtab + kWg3AuxFloats;
            if ((unsigned)hfmul < (unsigned)kWg3SfEntries) {
                sf[0] = sft[hfmul];
                sf[1] = sft[kWg3SfEntries + hfmul];
                sf[2] = sft[2 * kWg3SfEntries + hfmul];
            } else {
                sf[0] = f.scale_factor[0] / (float)hfmul;
                sf[1] = f.scale_factor[1] / (float)hfmul;
                sf[2] = f.scale_factor[2] / (float)hfmul;
            }
        }
        const float* aux = qtab + kWg3QTab;  // [70] cosine LUT of 2, 4, 8 points | [2] | [32] LLF scale | [192] LF patches
        const int cells_w = f.width >> 3, tw = f.tw;
        float dyk[16];  // this lane's 16 dequantised luma samples
#pragma unroll 1
        for (int ci = 0; ci < 3; ci++) {
            const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;  // Y first
            const float* wt = f.weights_t + irec[4 + c];  // TransformType.flip(): square METHOD_DCT
            const int32_t* cp = f.coeff[c];
            const float* kt = c == 0 ? f.kx_tab : f.kb_tab;
            v4i q[4];
            v4f w[4];
            float kf[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = tid + 256 * j;  // 4-sample group g of the block: row g / 16, columns 4 (g % 16) ..
                const int n = g >> 4, x4 = (g & 15) << 2;
                const int py = cy * 8 + n, px = cx * 8 + x4;
                const int64_t off = (((int64_t)(py >> 3) * cells_w + (px >> 3)) << 6) + (((py & 7) << 3) | (px & 7));  // coeff_off
                q[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(cp + off));  // (read once: see prefetch())
                w[j] = *reinterpret_cast<const v4f*>(wt + g * 4);
                kf[j] = 0.0f;
                if (c != 1) {  // chromaFromLuma factor of the group's 64x64 tile, honouring the cache order (DevBlock::cfl_zero)
                    const int ty = py >> 6, tx = px >> 6;
                    const int bit = (ty - ((cy * 8) >> 6)) * 5 + (tx - ((cx * 8) >> 6));
                    if (!((cfl_zero >> bit) & 1u)) kf[j] = kt[ty * tw + tx];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = tid + 256 * j;
                const int n = g >> 4, x4 = (g & 15) << 2;
                float dq[4];
                uint32_t big = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t t = (uint32_t)q[j][i] + 64u;
                    big |= t;
                    dq[i] = qtab[c * 128 + (t & 127u)];
                }
                if (big >= 128u) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int qv = q[j][i];
                        if ((uint32_t)qv + 64u >= 128u) dq[i] = (float)qv - qbn / (float)qv;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float v;
                    if (c == 1) {
                        v = dq[i] * sf[1] * w[j][i];
                        dyk[4 * j + i] = v;
                    } else {
                        v = dq[i] * sf[c] * w[j][i] + kf[j] * dyk[4 * j + i];  // chromaFromLuma (:186-188)
                    }
                    if (n < 8 && x4 + i < 8) continue;  // the LLF corner: written below
                    img[n * LD + x4 + i] = v;
                }
            }
            if (tid < 64) {
                const int ky = tid >> 3, kx = tid & 7;
                float v = aux[104 + c * 64 + tid];  // (llf_in_item = 0: the LLF coefficient itself, from the llf planes)
                if (a.llf_in_item) v = llf_coeff3<8, 8>(aux, aux + 104 + c * 64, 8, ky, kx, aux + 72);
                img[ky * LD + kx] = v;
            }
            lds_barrier();
            column_pass(f, tid, img);
            if (ci == 2) return;
            row_pass(f, tid, img, c == 1 ? a.o1 : a.o0, cy, cx, []() {});
            lds_barrier();  // every lane has read the image before the next channel's samples overwrite it
        }
    }
    template <typename PreStore>
    static __device__ __forceinline__ void back(const Wg3Args& a, const Item& it, int tid, const float* __restrict__ img, PreStore pre_store) {
        const auto* brec = (const __attribute__((address_space(4))) int*)a.blocks + 4 * it.first;
        const int cy = (int)((uint32_t)brec[0] & 0xffffu), cx = (int)((uint32_t)brec[0] >> 16);
        row_pass(a.f, tid, img, a.o2, cy, cx, pre_store);
    }
};

// type dispatch of the two specialised phases (workgroup-uniform scalar branch)
template <bool BIG>
__device__ __forceinline__ void do_dequant(const Wg3Args& a, const Item& it, int tid, const Raw<Cls<BIG>::NG, Cls<BIG>::WS>& raw, float* img,
                                           const float* qtab) {
    if constexpr (BIG) {
        switch (it.type) {
        case 18: Body<64, 64, 18>::dequant(a, it, tid, raw, img, qtab); break;
        case 19: Body<64, 32, 19>::dequant(a, it, tid, raw, img, qtab); break;
        case 20: Body<32, 64, 20>::dequant(a, it, tid, raw, img, qtab); break;
        default: __builtin_unreachable();  // item_of hands out the types of this class only
        }
    } else {
        switch (it.type) {
        case 0: Body<8, 8, 0>::dequant(a, it, tid, raw, img, qtab); break;
        case 4: Body<16, 16, 4>::dequant(a, it, tid, raw, img, qtab); break;
        case 5: Body<32, 32, 5>::dequant(a, it, tid, raw, img, qtab); break;
        case 6: Body<16, 8, 6>::dequant(a, it, tid, raw, img, qtab); break;
        case 7: Body<8, 16, 7>::dequant(a, it, tid, raw, img, qtab); break;
        case 8: Body<32, 8, 8>::dequant(a, it, tid, raw, img, qtab); break;
        case 9: Body<8, 32, 9>::dequant(a, it, tid, raw, img, qtab); break;
        case 10: Body<32, 16, 10>::dequant(a, it, tid, raw, img, qtab); break;
        case 11: Body<16, 32, 11>::dequant(a, it, tid, raw, img, qtab); break;
        case 18: break;  // (Item64::front, behind the barrier)
        default: SpecialBody::dequant(a, it, tid, raw, img, qtab); break;  // (item_of hands out the types of this class only)
        }
    }
}
template <bool BIG, typename PreStore>
__device__ __forceinline__ void do_passes(const Wg3Args& a, const Item& it, int tid, float* img, int it_no, int rowx, PreStore pre_store) {
    if constexpr (BIG) {
        switch (it.type) {
        case 18: Body<64, 64, 18>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 19: Body<64, 32, 19>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 20: Body<32, 64, 20>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        default: __builtin_unreachable();  // item_of hands out the types of this class only
        }
    } else {
        switch (it.type) {
        case 0: Body<8, 8, 0>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 4: Body<16, 16, 4>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 5: Body<32, 32, 5>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 6: Body<16, 8, 6>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 7: Body<8, 16, 7>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 8: Body<32, 8, 8>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 9: Body<8, 32, 9>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 10: Body<32, 16, 10>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 11: Body<16, 32, 11>::passes(a, it, tid, img, it_no, rowx, pre_store); break;
        case 18: Item64::back(a, it, tid, img, pre_store); break;
        default: special_store(a, it, tid, img, rowx, pre_store); break;  // (item_of hands out the types of this class only)
        }
    }
}

}  // namespace

// Persistent kernel: workgroup w handles the items w, w + G, w + 2G, ... (G = grid) of the launch's item list (segments in
// launch order, costliest type first: consecutive items go to different workgroups and every workgroup gets an equal share
// of every type). Dynamic hand-out through an atomic ticket counter was built and measured, and lost: one ticket per item on
// one address runs at the chip-wide ~90 atomics per microsecond (a 4000-item launch twice as slow), four tickets per atomic
// leave up to four items of imbalance at the tail (same loss). The grid is sized to be resident at once instead.
// Two instantiations by register / LDS class: BIG = the 64-point family (4096 positions per item: 50 KB of LDS, 4 groups
// per lane), !BIG = everything up to 32 points (2048 positions: <= 34 KB).
#ifndef WG3_SMALL_OCC
#define WG3_SMALL_OCC 4
#endif
#ifndef WG3_BIG_OCC
#define WG3_BIG_OCC (WG3_BIG_T == 256 ? 2 : 4)
#endif
template <bool BIG>
__device__ __forceinline__ void wg3_body(const Wg3Args& a) {
    constexpr int P = Cls<BIG>::P, T = Cls<BIG>::T, NG = Cls<BIG>::NG, WS = Cls<BIG>::WS;
    extern __shared__ float lds[];
    const int tid0 = threadIdx.x;
    const int G = (int)gridDim.x;
#ifdef JXL_IDCT_PRIO
    __builtin_amdgcn_s_setprio(JXL_IDCT_PRIO);
#endif
    int gi = (int)blockIdx.x;  // list position of the item requested last (cur, then nxt, then nn)
    Item cur = item_of<P>(a, gi, G);
    if (cur.type < 0) return;
    STAMP3_LIFE(0, __builtin_amdgcn_s_memtime());
    float* img = lds;
    float* qtab = lds + a.img_floats;  // [3][128]: entry q + 64 = 0, +-quantBias[c], (float)q - qbn / (float)q (HFCoefficients.java:309-315)
    for (int i = tid0; i < kWg3QTab; i += T) {
        const int c = i >> 7, q = (i & 127) - 64;
        qtab[i] = q == 0 ? 0.0f : q == 1 ? a.f.quant_bias[c] : q == -1 ? -a.f.quant_bias[c] : (float)q - a.f.quant_bias_numerator / (float)q;
    }
    // finalizeLLF inside the item: the cosine tables of 2, 4 and 8 points (the first 70 floats of the LUT), the LLF scale
    // table and one LF sample per LLF coefficient of the coming item (lf_patch, written when the item's prefetch has landed)
    float* aux = qtab + kWg3QTab;
    float* lf_patch = aux + 104;
    if (tid0 < 70) aux[tid0] = a.f.lut[tid0];
    else if (tid0 >= 72 && tid0 < 104) aux[tid0] = kLlfScale3[tid0 - 72];
    for (int i = tid0; i < 3 * kWg3SfEntries; i += T) {  // hfMultiplier values 1 .. 255 (larger ones divide in place)
        const int c = i / kWg3SfEntries, m = i % kWg3SfEntries;
        qtab[kWg3AuxFloats + i] = a.f.scale_factor[c] / (float)(m > 0 ? m : 1);
    }
    Raw<NG, WS> raw{};  // (zeroed once: the prefetch leaves the groups outside an item alone)
    Recs<NG> rc;
    load_recs<T, NG>(a, cur, tid0, rc);
    prefetch<T, NG, WS>(a, cur, tid0, rc, raw);
    gi += G;
    Item nxt = item_of<P>(a, gi, G);
    load_recs<T, NG>(a, nxt, tid0, rc);
    // "every load issued so far has landed": an empty asm that reads the destination registers makes the compiler place the
    // wait HERE. Used (i) once before the loop -- the first item needs its data anyway, and the loop header then has no load
    // in flight on either incoming edge, so no conservative vmcnt wait is generated inside the loop -- and (ii) in every
    // item right before its stores are issued: the prefetch and the records have had both passes to arrive, so the wait is
    // free there, whereas any wait for them AFTER the stores (vmcnt counts loads and stores in one in-order queue) would last
    // until the stores have drained.
    auto loads_landed = [&]() {
#pragma unroll
        for (int j = 0; j < NG; j++) {
            asm volatile("" ::"v"(raw.q[j][0]), "v"(raw.q[j][1]), "v"(raw.q[j][2]), "v"(raw.kx[j]), "v"(raw.kb[j]), "v"(rc.gx[j]), "v"(rc.gz[j]),
                         "v"(rc.gw[j]));
        }
#pragma unroll
        for (int j = 0; j < WS; j++) asm volatile("" ::"v"(raw.w[j][0]), "v"(raw.w[j][1]), "v"(raw.w[j][2]));
        asm volatile("" ::"v"(raw.llf), "v"(rc.lx), "v"(raw.rowx));
    };
    loads_landed();
    if (tid0 < 192) lf_patch[tid0] = raw.llf;
    lds_barrier();  // qtab, aux, lf_patch
    // the next item's patches: published when its prefetch has landed (right before this item's stores), read by the next
    // dequant() -- the barrier at the end of the loop body lies between
    auto landed_and_published = [&]() {
        loads_landed();
        if (tid0 < 192) lf_patch[tid0] = raw.llf;
    };
    int it_no = 0;
#pragma unroll 1
    for (;;) {
        STAMP3(0);
        // the lane index is made opaque once per item: otherwise the optimiser hoists the lane-dependent address arithmetic
        // of ALL type cases out of the item loop (some 80 loop-invariant VGPRs) and spills it
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        do_dequant<BIG>(a, cur, tid, raw, img, qtab);  // the loads were issued one item ago
        const int rowx = raw.rowx;                     // (the prefetch below overwrites raw)
        STAMP3(1);
        lds_barrier();
        STAMP3(2);
        // everything the next item needs from memory: in flight during both passes of this one. Its block records were
        // requested one item earlier still, so no load here waits for another. (A special 8x8 item issues them between its transform
        // and its stores instead: special_passes.)
        // a special 8x8 item transforms its blocks first (lane = block of one channel, in registers: special_transform); the registers
        // of the item's own requests are dead by now, and those of the next item's are not in use yet
        if (!BIG && __builtin_expect(wg3_is_special(cur.type), 0)) {
            // (the 256-thread class's prefetch leaves the registers of groups outside an item alone, i.e. the old values stay live
            // across the call below: ended here -- 45 v_mov per special item)
            raw = Raw<NG, WS>{};
            special_phase(cur, tid);
        }
        if (!BIG && __builtin_expect(cur.type == 18, 0)) {  // a 64x64 block: everything up to its last channel's row pass (Item64)
            raw = Raw<NG, WS>{};
            Item64::front(a, cur, tid, img, qtab, rowx);
        }
        prefetch<T, NG, WS>(a, nxt, tid, rc, raw);
        gi += G;
        const Item nn = item_of<P>(a, gi, G);
        load_recs<T, NG>(a, nn, tid, rc);
        STAMP3(3);
        do_passes<BIG>(a, cur, tid, img, it_no, rowx, landed_and_published);
        if (nxt.type < 0) break;
        cur = nxt;
        nxt = nn;
        STAMP3(9);
        lds_barrier();  // every lane has read the image before the next item's samples overwrite it
        STAMP3(10);
        it_no++;
    }
    STAMP3_LIFE(1, __builtin_amdgcn_s_memtime() | ((unsigned long long)(it_no + 1) << 56));
    STAMP3_FLUSH();
}

template <bool BIG>
__global__ __launch_bounds__(Cls<BIG>::T, BIG ? WG3_BIG_OCC : WG3_SMALL_OCC) void k_idct_wg3(const Wg3Args a) {
    wg3_body<BIG>(a);
}

// a batch of frames in one launch (jxl_vardct_run_batch): blockIdx.y = frame, each frame's workgroups stride over that frame's
// items; the argument blocks live in device memory and are read through the constant address space (scalar loads)
template <bool BIG>
__global__ __launch_bounds__(Cls<BIG>::T, BIG ? WG3_BIG_OCC : WG3_SMALL_OCC) void k_idct_wg3_batch(const Wg3Args* __restrict__ args) {
    typedef const __attribute__((address_space(4))) Wg3Args* cargs;
    wg3_body<BIG>(*(const Wg3Args*)((cargs)args + blockIdx.y));
}

// finalizeLLF (HFCoefficients.java:194-229) of every block of the launch's segments, one lane per coefficient, written over
// the block's own cells of the llf planes (a block covers exactly dctSelectHeight x dctSelectWidth cells)
template <int H, int W>
__device__ __forceinline__ void llf_one(const Wg3Args& a, const Wg3Seg& sg, int t, float* l0, float* l1, float* l2) {
    constexpr int DSH = H / 8, DSW = W / 8, PER_B = 3 * DSH * DSW;
    const int b = t / PER_B, rr = t % PER_B;
    if (b >= sg.n_blocks) return;
    const int c = rr / (DSH * DSW), k = rr % (DSH * DSW);
    const v4i rec = ((cv4ip)a.blocks)[sg.first_block + b];
    const int cy = (int)((uint32_t)rec.x & 0xffffu), cx = (int)((uint32_t)rec.x >> 16);
    const float v = llf_coeff3<DSH, DSW>(a.f.lut, a.f.lf[c] + (int64_t)cy * a.f.bw + cx, a.f.bw, k / DSW, k % DSW, kLlfScale3);
    (c == 0 ? l0 : c == 1 ? l1 : l2)[(int64_t)(cy + k / DSW) * a.f.bw + cx + k % DSW] = v;
}

__device__ __forceinline__ void llf_wg3_body(const Wg3Args& a, float* l0, float* l1, float* l2) {
    int t = (int)(blockIdx.x * 256 + threadIdx.x);
    for (int k = 0; k < a.n_seg; k++) {
        const Wg3Seg sg = a.seg[k];
        // (the LLF of an 8x8 block is its LF sample: the llf planes start as a copy of the lf planes)
        const int n = sg.type == 0 ? 0 : sg.n_blocks * 3 * (JXL_TT[sg.type].ph / 8) * (JXL_TT[sg.type].pw / 8);
        if (t < n) {
            switch (sg.type) {
            case 4: llf_one<16, 16>(a, sg, t, l0, l1, l2); break;
            case 5: llf_one<32, 32>(a, sg, t, l0, l1, l2); break;
            case 6: llf_one<16, 8>(a, sg, t, l0, l1, l2); break;
            case 7: llf_one<8, 16>(a, sg, t, l0, l1, l2); break;
            case 8: llf_one<32, 8>(a, sg, t, l0, l1, l2); break;
            case 9: llf_one<8, 32>(a, sg, t, l0, l1, l2); break;
            case 10: llf_one<32, 16>(a, sg, t, l0, l1, l2); break;
            case 11: llf_one<16, 32>(a, sg, t, l0, l1, l2); break;
            case 18: llf_one<64, 64>(a, sg, t, l0, l1, l2); break;
            case 19: llf_one<64, 32>(a, sg, t, l0, l1, l2); break;
            case 20: llf_one<32, 64>(a, sg, t, l0, l1, l2); break;
            default: break;
            }
            return;
        }
        t -= n;
    }
}

__global__ __launch_bounds__(256) void k_llf_wg3(const Wg3Args a, float* l0, float* l1, float* l2) {
    llf_wg3_body(a, l0, l1, l2);
}

// batch form: the llf planes are the ones the frame's own argument block names (DevFrame::llf)
__global__ __launch_bounds__(256) void k_llf_wg3_batch(const Wg3Args* __restrict__ args) {
    typedef const __attribute__((address_space(4))) Wg3Args* cargs;
    const Wg3Args& a = *(const Wg3Args*)((cargs)args + blockIdx.y);
    llf_wg3_body(a, const_cast<float*>(a.f.llf[0]), const_cast<float*>(a.f.llf[1]), const_cast<float*>(a.f.llf[2]));
}

#ifdef JXL_STAMPS
extern "C" int jxl_debug_set_stamps3(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps3), &p, sizeof p);
}
#endif

bool wg3_llf_in_item() {
    static const bool v = !(getenv("JXL_WG3_LLF_IN_ITEM") && atoi(getenv("JXL_WG3_LLF_IN_ITEM")) == 0);
    return v;
}

bool wg3_handles(int type) {
    // experiment knob: JXL_WG3_SKIP=<bit mask of types> leaves those types to the per-channel kernels of k_idct.hip
    static const unsigned skip = getenv("JXL_WG3_SKIP") ? (unsigned)strtoul(getenv("JXL_WG3_SKIP"), nullptr, 0) : 0u;
    if (type < 32 && ((skip >> type) & 1u)) return false;
    switch (type) {
    case 0: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 18: case 19: case 20: return true;
    default: return wg3_special_items() && wg3_is_special(type);
    }
}
// r6: the special 8x8 types as items of the persistent launch (JXL_WG3_SPECIAL=0: their own launch on the side stream, as until r5)
bool wg3_special_items() {
    static const bool v = !(getenv("JXL_WG3_SPECIAL") && atoi(getenv("JXL_WG3_SPECIAL")) == 0);
    return v;
}
// r6: 64x64 blocks as items of the 256-thread launch (Item64; JXL_WG3_FOLD64=0: in the 512-thread class's own launch, as until r5).
// The 64x32 / 32x64 types stay in the 512-thread class (a single block of them has fewer columns / rows than a wave has lanes).
bool wg3_fold64() {
    static const bool v = !(getenv("JXL_WG3_FOLD64") && atoi(getenv("JXL_WG3_FOLD64")) == 0);
    return v;
}
bool wg3_big(int type) { return (type == 18 && !wg3_fold64()) || type == 19 || type == 20; }

int wg3_grid_cap(bool big) {
    // r6: 768 = three 256-thread workgroups per CU. Until r5 512, so that the 64-point and the special launches fitted beside this one;
    // with those blocks as items of this launch (one launch per frame) the batch is 1.5-2 % faster on 640-1024 workgroups and a frame alone
    // 3 % (profiles/experiments/r6_wg3_grid_single_launch.txt)
    static const int g_small = getenv("JXL_WG3_GRID") ? atoi(getenv("JXL_WG3_GRID")) : 768;
    static const int g_big = getenv("JXL_WG3_GRID_BIG") ? atoi(getenv("JXL_WG3_GRID_BIG")) : 512;
    return big ? g_big : g_small;
}

int wg3_blocks_per_item(int type) {
    const int h = JXL_TT[type].ph, w = JXL_TT[type].pw;
    if (type == 18 && !wg3_big(type)) return 1;  // Item64: one block, channel by channel
    return ((h > w ? h : w) <= 32 ? 2048 : 4096) / (h * w);
}

// floats of the three-channel LDS image of a type
static int wg3_img_floats(int type) {
    const int h = JXL_TT[type].ph, w = JXL_TT[type].pw;
    const int nb = wg3_blocks_per_item(type);
    if (wg3_is_special(type)) return 3 * nb * 65;  // SpecialBody
    if (type == 18 && !wg3_big(type)) return 64 * 65;  // Item64: one channel at a time
    const int img0 = h * (w + 1);
    const int img = w >= 32 ? img0 : img0 + ((w - img0 % 32) + 32) % 32;
    return 3 * nb * img;
}

// Fills the argument block for the frame's types of one register class (which = 0: up to 32 points, 1: the 64-point family;
// 2: both, for the LLF launch), in the given launch order. Returns the number of items (0: nothing to launch; -1: more
// segments than the argument block holds).
int build_wg3_args(const DevFrame& f, const DevBlock* blocks, const IdctSegment* segs, int n_seg, int which, float* const out[3],
                   Wg3Args& a) {
    a.f = f;
    a.blocks = blocks;
    a.o0 = out[0]; a.o1 = out[1]; a.o2 = out[2];
    a.n_seg = 0;
    a.total_items = 0;
    a.img_floats = 0;
    a.items = nullptr;
    a.llf_in_item = wg3_llf_in_item() ? 1 : 0;
    static_assert(Wg3Args::kMaxSeg >= 21, "one segment per type wg3_handles() accepts (21 types: the LLF launch passes both classes)");
    for (int i = 0; i < n_seg; i++) {
        if (segs[i].n_blocks <= 0 || !wg3_handles(segs[i].type) || (which != 2 && wg3_big(segs[i].type) != (which == 1))) continue;
        // a type twice in the list, or a new type without a larger kMaxSeg: never drop blocks silently, never kill the host
        // process (finalize_tables checks the lists it builds and reports JXL_ERR_STATE)
        if (a.n_seg >= Wg3Args::kMaxSeg) return -1;
        Wg3Seg& sg = a.seg[a.n_seg++];
        const int nb = wg3_blocks_per_item(segs[i].type);
        sg.type = segs[i].type;
        sg.first_block = segs[i].first_block;
        sg.n_blocks = segs[i].n_blocks;
        sg.item_base = a.total_items;
        a.total_items += (segs[i].n_blocks + nb - 1) / nb;
        a.img_floats = std::max(a.img_floats, wg3_img_floats(segs[i].type));
    }
    return a.total_items;
}

// Spatial item order. Launched type after type, a line of the coefficient / output planes (128 bytes = 4 cells wide) is
// touched once per type that owns one of its cells, megabytes apart in time: mixed frames cost 5-16 % over the sum of their
// types at 4K (the lines come back from the memory-side cache) and more once the frame outgrows it (8K: 95 against 106 Gpx/s
// for the same mix; batches likewise). So the items are sorted by the 256 x 256 group of their first block (the block
// lists are group-major already, so an item's blocks are neighbours) and dealt to the XCDs in runs of about one group's
// items: workgroup w takes items w, w + G, ...; with G a multiple of 8 item i runs on XCD i % 8, so a run's items -- the
// ones that share lines -- are in flight together behind ONE L2.
//
// Balance (r3). The persistent grid takes list positions w, w + G, ...: a workgroup gets ~13 items of a 4K frame, their types as
// the spatial order happens to deal them, and an item of 32-point blocks costs 1.8x one of 8-point blocks: the slowest of 512
// workgroups ran 25 % longer than the average one (SQ_WAVE_CYCLES / waves = 57 us of a 73 us launch, default mix; 52 of 60 for a
// frame of DCT8 only). With `grid` given, the items of every round k (positions [kG, (k+1)G)) are permuted among the positions of
// the same XCD (same position mod 8: the run / L2 pairing stays, and so does the time at which a region is touched): the costliest
// item of the round goes to the workgroup with the least work so far. Every position keeps an item, so a launch with another
// grid (the batch path) is still correct, only unbalanced.
static float wg3_item_cost(int type) {  // us per 4K frame tiled with the type, i.e. relative cost of 2048 positions
    switch (type) {
    case 0: return 55.f;
    case 1: case 2: case 3: case 12: case 13: case 14: case 15: case 16: case 17: return 70.f;
    case 4: return 76.f;
    case 5: return 100.f;
    case 6: case 7: return 67.f;
    case 8: case 9: return 83.f;
    case 10: case 11: return 90.f;
    case 18: return 330.f;  // (in the 256-thread class: ONE 64x64 block = 4096 positions, channel by channel, nothing prefetched)
    default: return 100.f;
    }
}

void wg3_item_table(const DevBlock* hb, int frame_bw, const IdctSegment* segs, int n_seg, int which, const int32_t* woffs, bool spatial,
                    std::vector<int>& out, int grid) {
    struct Rec { uint32_t key; int type, first, nb; };
    std::vector<Rec> recs;
    static const int rsh = getenv("JXL_WG3_REGION_SHIFT") ? std::min(12, std::max(0, atoi(getenv("JXL_WG3_REGION_SHIFT")))) : 5;
    const int grs = std::max(1, (frame_bw + (1 << rsh) - 1) >> rsh);
    for (int i = 0; i < n_seg; i++) {
        if (segs[i].n_blocks <= 0 || !wg3_handles(segs[i].type) || (which != 2 && wg3_big(segs[i].type) != (which == 1))) continue;
        const int nb = wg3_blocks_per_item(segs[i].type);
        for (int o = 0; o < segs[i].n_blocks; o += nb) {
            const DevBlock& b0 = hb[segs[i].first_block + o];
            recs.push_back(Rec{(uint32_t)((b0.cy >> rsh) * grs + (b0.cx >> rsh)), segs[i].type, segs[i].first_block + o, std::min(nb, segs[i].n_blocks - o)});
        }
    }
    auto emit = [&](const Rec& r) {
        const int pi3 = (int)JXL_TT[r.type].param_index * 3;
        // (geometry bit 14: the item fetches its coefficients itself -- a 64x64 block in the 256-thread class, Item64)
        const int rec[8] = {r.type, r.first, r.nb, (int)(wg3_geo(r.type) | (r.type == 18 && !wg3_big(18) ? 1u << 14 : 0u)), woffs[pi3], woffs[pi3 + 1], woffs[pi3 + 2], 0};
        out.insert(out.end(), rec, rec + 8);
    };
    if (!spatial) {  // JXL_WG3_SPATIAL=0: the segments' own order, type after type
        out.clear();
        for (const Rec& r : recs) emit(r);
        return;
    }
    // r6: the special 8x8 items go BEHIND the others (each part in its own spatial order): such an item issues the next item's requests
    // only after its transform (wg3_body), i.e. the item behind it waits for memory -- at the end of a workgroup's list that item is
    // another special one or none (a 4K frame of the default mix has about one special item per workgroup) --, and the 64x64 blocks of
    // the 256-thread class (Item64: nothing of them is prefetched either) behind those.
    auto tail_rank = [](const Rec& r) { return wg3_is_special(r.type) ? 1 : (r.type == 18 && !wg3_big(18)) ? 2 : 0; };
    std::stable_sort(recs.begin(), recs.end(), [&](const Rec& x, const Rec& y) { return tail_rank(x) < tail_rank(y); });
    const size_t n_normal = (size_t)(std::find_if(recs.begin(), recs.end(), [&](const Rec& r) { return tail_rank(r) > 0; }) - recs.begin());
    const size_t n_special_end = (size_t)(std::find_if(recs.begin(), recs.end(), [&](const Rec& r) { return tail_rank(r) > 1; }) - recs.begin());
    static const int run = getenv("JXL_WG3_RUN") ? std::max(1, atoi(getenv("JXL_WG3_RUN"))) : 24;
    out.clear();
    // per part: the eight queues (one per XCD: workgroup w runs on XCD w % 8) in spatial order, dealt in runs of about one group's items
    std::vector<const Rec*> q[3][8];
    for (int part = 0; part < 3; part++) {
        const size_t r0 = part == 0 ? 0 : part == 1 ? n_normal : n_special_end, r1 = part == 0 ? n_normal : part == 1 ? n_special_end : recs.size();
        std::stable_sort(recs.begin() + (ptrdiff_t)r0, recs.begin() + (ptrdiff_t)r1, [](const Rec& x, const Rec& y) { return x.key < y.key; });
        for (size_t i = r0; i < r1; i++) q[part][((i - r0) / (size_t)(part == 2 ? 1 : run)) % 8].push_back(&recs[i]);
    }
    static const bool balance = !(getenv("JXL_WG3_BALANCE") && atoi(getenv("JXL_WG3_BALANCE")) == 0);
    const size_t G = (size_t)std::max(0, grid);
    if (!(balance && G >= 8 && G % 8 == 0 && recs.size() > G)) {
        // no grid to balance for: the queues interleaved (position p belongs to queue p % 8 while all eight last), part after part
        for (int part = 0; part < 3; part++) {
            size_t longest = 0;
            for (auto& v : q[part]) longest = std::max(longest, v.size());
            for (size_t i = 0; i < longest; i++)
                for (int x = 0; x < 8; x++)
                    if (i < q[part][x].size()) emit(*q[part][x][i]);
        }
        return;
    }
    // Balance (r3; r6: explicit lists). A workgroup's list is the positions w, w + G, ...; an item of 32-point blocks costs 1.8 x one
    // of 8-point blocks and a 64x64 block six times as much. The 64x64 blocks are dealt first (they are LAST in the lists, but on the
    // account from the start). Then every queue is dealt round by round to the workgroups of its XCD: a round gives one item to every
    // workgroup that is not more than one 32-point item ahead of the least loaded one -- the costliest item of the round to the one
    // with the least work so far (r3's rule) --, so that neighbours in the spatial order run at about the same time behind ONE L2,
    // the workgroups without a 64x64 block differ by at most an item, and the ones with such a block sit rounds out. A list shorter
    // than the longest ends in holes (type -2: wg3_body's item walk steps over them, whatever grid it is launched on).
    const size_t Gq = G / 8;
    std::vector<float> load(G, 0.0f);
    std::vector<std::vector<const Rec*>> lists[3];
    for (auto& l : lists) l.assign(G, {});
    for (int x = 0; x < 8; x++)
        for (size_t i = 0; i < q[2][x].size(); i++) {
            const size_t w = (size_t)x + 8 * (i % Gq);
            lists[2][w].push_back(q[2][x][i]);
            load[w] += wg3_item_cost(q[2][x][i]->type);
        }
    const float slack = wg3_item_cost(5);
    std::vector<size_t> elig;
    std::vector<const Rec*> its;
    for (int x = 0; x < 8; x++)
        for (int part = 0; part < 2; part++) {
            const auto& qq = q[part][x];
            for (size_t pos = 0; pos < qq.size();) {
                float lo = load[(size_t)x];
                for (size_t k = 1; k < Gq; k++) lo = std::min(lo, load[(size_t)x + 8 * k]);
                elig.clear();
                for (size_t k = 0; k < Gq; k++)
                    if (load[(size_t)x + 8 * k] <= lo + slack) elig.push_back((size_t)x + 8 * k);
                std::stable_sort(elig.begin(), elig.end(), [&](size_t u, size_t v) { return load[u] < load[v]; });
                const size_t m = std::min(elig.size(), qq.size() - pos);
                its.assign(qq.begin() + (ptrdiff_t)pos, qq.begin() + (ptrdiff_t)(pos + m));
                std::stable_sort(its.begin(), its.end(), [](const Rec* u, const Rec* v) { return wg3_item_cost(u->type) > wg3_item_cost(v->type); });
                for (size_t i = 0; i < m; i++) {
                    lists[part][elig[i]].push_back(its[i]);
                    load[elig[i]] += wg3_item_cost(its[i]->type);
                }
                pos += m;
            }
        }
    size_t rounds = 0;
    for (size_t w = 0; w < G; w++) rounds = std::max(rounds, lists[0][w].size() + lists[1][w].size() + lists[2][w].size());
    out.assign(rounds * G * 8, 0);
    for (size_t i = 0; i < rounds * G; i++) out[i * 8] = -2;  // holes
    for (size_t w = 0; w < G; w++) {
        size_t k = 0;
        for (int part = 0; part < 3; part++)
            for (const Rec* r : lists[part][w]) {
                const int pi3 = (int)JXL_TT[r->type].param_index * 3;
                const int rec[8] = {r->type, r->first, r->nb, (int)(wg3_geo(r->type) | (r->type == 18 && !wg3_big(18) ? 1u << 14 : 0u)), woffs[pi3], woffs[pi3 + 1], woffs[pi3 + 2], 0};
                std::copy(rec, rec + 8, out.begin() + (ptrdiff_t)((k * G + w) * 8));
                k++;
            }
    }
}


// LLF coefficients of the class's blocks into the llf planes (must precede launch_idct_wg3 on the same stream)
void launch_llf_wg3(const Wg3Args& a, float* const llf[3], hipStream_t s) {
    int64_t n = 0;
    for (int k = 0; k < a.n_seg; k++)
        if (a.seg[k].type != 0) n += (int64_t)a.seg[k].n_blocks * 3 * (JXL_TT[a.seg[k].type].ph / 8) * (JXL_TT[a.seg[k].type].pw / 8);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_llf_wg3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, llf[0], llf[1], llf[2]);
}

int64_t wg3_llf_count(const Wg3Args& a) {
    int64_t n = 0;
    for (int k = 0; k < a.n_seg; k++)
        if (a.seg[k].type != 0) n += (int64_t)a.seg[k].n_blocks * 3 * (JXL_TT[a.seg[k].type].ph / 8) * (JXL_TT[a.seg[k].type].pw / 8);
    return n;
}

// block images + dequantisation table [3][64] + finalizeLLF tables (cosine LUT 70 + 2, LLF scale 32, LF patches 192) + scale table [3][256]
size_t wg3_lds_bytes(const Wg3Args& a) { return sizeof(float) * ((size_t)a.img_floats + kWg3AuxFloats + 3 * kWg3SfEntries); }

// batch forms: dev_args[0..n_frames) in device memory; max_llf = the largest wg3_llf_count, grid_x workgroups per frame,
// lds = the largest wg3_lds_bytes among the frames
void launch_llf_wg3_batch(const Wg3Args* dev_args, int n_frames, int64_t max_llf, hipStream_t s) {
    if (n_frames <= 0 || max_llf <= 0) return;
    hipLaunchKernelGGL(k_llf_wg3_batch, dim3((unsigned)((max_llf + 255) / 256), n_frames), dim3(256), 0, s, dev_args);
}

void launch_idct_wg3_batch(const Wg3Args* dev_args, int n_frames, bool big, int grid_x, size_t lds, hipStream_t s) {
    if (n_frames <= 0 || grid_x <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idct_wg3_batch<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idct_wg3_batch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    if (big) hipLaunchKernelGGL(k_idct_wg3_batch<true>, dim3(grid_x, n_frames), dim3(WG3_BIG_T), lds, s, dev_args);
    else hipLaunchKernelGGL(k_idct_wg3_batch<false>, dim3(grid_x, n_frames), dim3(256), lds, s, dev_args);
}

// grid_cap: workgroups to launch at most (persistent; never more than there are items)
void launch_idct_wg3(const Wg3Args& a, bool big, int grid_cap, hipStream_t s) {
    if (a.total_items <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idct_wg3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idct_wg3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    const size_t lds = wg3_lds_bytes(a);
    const int grid = std::max(1, std::min(a.total_items, grid_cap));
    if (big) hipLaunchKernelGGL(k_idct_wg3<true>, dim3(grid), dim3(WG3_BIG_T), lds, s, a);
    else hipLaunchKernelGGL(k_idct_wg3<false>, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace jxl
