// Fused restoration + colour tile kernel for gfx950:
//   Gaborish -> EPF iteration(s) -> XYB->linear -> (transfer + quantise) in ONE launch, one read and
//   one write of the frame planes; all intermediates live in LDS.
//
// Replaces the same reference functions as k_restore.hip (Frame.java:505-679,
// OpsinInverseMatrix.java:105-142, JXLImage.java:244-258, ImageBuffer.java:129-147) and is checked
// bit-for-bit against them through the oracle AND against the stage kernels.
//
// Geometry (template <GAB, ITERS, SK, PH>): the most expensive stage -- the first EPF iteration that runs -- is computed on a
// 64x32 window = 512 threads x (4 wide x 1 tall) register patches (PH = 2: 256 threads x 4x2 patches, measured slower). Later
// iterations shrink the window by their radius (iteration 1: 2 px of input halo, iteration 2: 1, iteration 0: 3), Gaborish
// adds 1. E.g. the default (Gab + iterations 1,2): input tile 70x38 -> Gab 68x36 -> EPF1 64x32 -> EPF2 62x30 output tile.
// r5: a frame with three iterations runs as TWO launches (host.hip run_frame): ITERS = 4 -- Gaborish + the 13-tap iteration alone on a
// 64x64 tile, float planes out -- and ITERS = 2 without Gaborish from there; ITERS = 3 (all in one launch) stays for JXL_EPF3_SPLIT=0.
// ONE LDS image of 3 planes (~34 KB) that every stage updates IN PLACE: a thread computes the patch it owns into registers, the
// workgroup meets at a barrier (all reads of the old values done), then the patches are written back over the input
// (4 workgroups per CU = 8 waves per SIMD at 63 VGPRs).
//
// Exactness: per pixel the EPF distance is the reference's strictly sequential sum
//   dist = (((0 + |a-b|*s0) + ...)            channel-major, cross order (0,0),(0,-1),(0,1),(-1,0),(1,0)
// Every |P(u) - P(v)| * scale term is written with canonically ordered operands (|x-y| == |y-x| exactly)
// so the compiler's value numbering shares the terms between the pixels and taps of a patch instead of
// recomputing them. The centre tap has distance exactly 0 and weight exactly 1 for finite samples (the
// IDCT output is finite), so it is folded: sumWeights starts at 1, sumChannels at the centre sample.
// Frame edges: Gab reads clamped coordinates, EPF reads mirrored ones (MathHelper.mirrorCoordinate);
// edge tiles re-create that by copying mirrored positions inside LDS after each stage.
//
// This header holds the kernel template (restore_fused_body<GAB, ITERS, SK, PH>, SK = the output sink kind of restore_sink.h) and its
// launch templates; it is included by k_restore_fused.hip (float planes: the hot variants, and the public entry points),
// k_restore_fused_gen.hip (the run-time-generic sink) and k_restore_fused_q.hip (the quantised PNG / u16 sinks), so that the three
// groups of instantiations compile in parallel.
#pragma once
#include <hip/hip_ext.h>  // hipExtLaunchKernelGGL (a launch that records the kernel's own start / stop events)
#include "jxl_internal.h"
#include <algorithm>
#include "restore_sink.h"
#include <cstdlib>

namespace jxl {

#ifndef JXL_EPF3_PH2_WAVES
#define JXL_EPF3_PH2_WAVES 2
#endif

namespace {

__device__ __forceinline__ int mirror_c(int c, int size) {
    while (c < 0 || c >= size) {
        const int tc = ~c;
        c = tc >= 0 ? tc : (size << 1) + tc;
    }
    return c;
}

template <bool GAB, int ITERS>
struct Geo {
    static constexpr int RG = GAB ? 1 : 0;
    // ITERS = 4: the 13-tap iteration ALONE (the first half of a three-iteration frame run as two launches, k_restore_fused.hip)
    static constexpr int R0 = (ITERS == 3 || ITERS == 4) ? 3 : 0;
    static constexpr int R1 = (ITERS >= 1 && ITERS <= 3) ? 2 : 0;
    static constexpr int R2 = (ITERS == 2 || ITERS == 3) ? 1 : 0;
    static constexpr int SHR = ITERS == 3 ? (R1 + R2) : ITERS == 2 ? R2 : 0;  // window -> output tile
#ifndef JXL_EPF0_WH
#define JXL_EPF0_WH 64
#endif
    // the 13-tap iteration alone has no later stage to shrink for, and its 118 registers hold it to 2 workgroups per CU whatever the
    // tile costs in LDS: a 64 x 64 window (66 KB) loads and Gab-filters 1.27 x / 1.20 x its pixels instead of 1.41 x / 1.30 x
    static constexpr int WH = ITERS == 4 ? JXL_EPF0_WH : 32;                  // window height
    static constexpr int OW = 64 - 2 * SHR, OH = WH - 2 * SHR;               // output tile
    static constexpr int RE = R0 + R1 + R2;
    static constexpr int RT = RE + RG;                                        // input halo
    static constexpr int IW = OW + 2 * RT, IH = OH + 2 * RT;                  // input tile
    static constexpr int SW = IW + 3;                                         // LDS row stride (patch over-read <= 3)
    static constexpr int SH = IH + 1;
    static constexpr int PLANE = SW * SH;
    static constexpr size_t LDS_BYTES = sizeof(float) * (3 * PLANE + 16 * 16);
};

// canonical |P[u] - P[v]| * s: operands ordered by index so equal terms are literally the same expression
template <int PS>
__device__ __forceinline__ float adiff(const float* p, int u, int v, float s) {
    return u < v ? fabsf(p[u] - p[v]) * s : fabsf(p[v] - p[u]) * s;
}

// One EPF iteration (Frame.java:583-635) on a 4x2 patch whose top-left sample is (ry, rx) in region
// coordinates of src (3 planes, stride SW). Results for the 8 pixels go to res[c][py*4+px].
// ITER: 0 = 13 taps with cross distances, 1 = 5 taps with cross distances, 2 = 5 taps single-pixel.
template <int ITER, int SW, int PLANE, int PH, bool CHAINS = true>
__device__ __forceinline__ void epf_patch(const float* __restrict__ src, int ry, int rx, const float* s_inv /*[NP]*/,
                                          const float* bmul /*[NP]*/, const EpfParams& ep, float res[3][4 * PH]) {
    constexpr int NP = 4 * PH;  // pixels per patch
    constexpr int R = ITER == 0 ? 3 : ITER == 1 ? 2 : 1;  // neighbourhood radius
    constexpr int NW = 4 + 2 * R, NH = PH + 2 * R;
    constexpr int NT = ITER == 0 ? 12 : 4;  // non-centre taps
    // (dy, dx) in the reference's order, centre tap folded (Frame.java:44-55)
    constexpr int TY[12] = {0, 0, -1, 1, -1, 1, 1, -1, 0, 0, 2, -2};
    constexpr int TX[12] = {-1, 1, 0, 0, 1, 1, -1, -1, -2, 2, 0, 0};
    constexpr int QY[5] = {0, 0, 0, -1, 1};
    constexpr int QX[5] = {0, -1, 1, 0, 0};

    float dist[NP][NT];
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) dist[i][t] = 0.0f;

    if (ITER != 0 && CHAINS) {
        // The distance of pixel p to its east neighbour IS the distance of p+1 to its west neighbour: the same
        // terms |P(p+k) - P(p+1+k)| * s in the same channel-major, cross-minor order (likewise south/north). So a
        // patch needs one chain per horizontally / vertically adjacent pixel PAIR, not one per (pixel, tap):
        //   hc[py][j] = distance between (py, j-1) and (py, j), j = 0..4;  vc[i][px] = between (i-1, px) and (i, px), i = 0..PH
        float hc[PH][5], vc[PH + 1][4];
#pragma unroll
        for (int py = 0; py < PH; py++)
#pragma unroll
            for (int j = 0; j < 5; j++) hc[py][j] = 0.0f;
#pragma unroll
        for (int i = 0; i <= PH; i++)
#pragma unroll
            for (int px = 0; px < 4; px++) vc[i][px] = 0.0f;
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            float nb[NH * NW];
            const float* pc = src + c * PLANE + (ry - R) * SW + (rx - R);
#pragma unroll
            for (int y = 0; y < NH; y++)
#pragma unroll
                for (int x = 0; x < NW; x++) nb[y * NW + x] = pc[y * SW + x];
            const float sc = ep.channel_scale[c];
#pragma unroll
            for (int py = 0; py < PH; py++)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int cy = py + R, cx = j - 1 + R;
                    if (ITER == 2) hc[py][j] = hc[py][j] + adiff<NW>(nb, cy * NW + cx, cy * NW + cx + 1, sc);
                    else
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            const int u = (cy + QY[q]) * NW + cx + QX[q];
                            hc[py][j] = hc[py][j] + adiff<NW>(nb, u, u + 1, sc);
                        }
                }
            constexpr int I0 = 0;
#pragma unroll
            for (int i = I0; i <= PH; i++)
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    const int cy = i - 1 + R, cx = px + R;
                    if (ITER == 2) vc[i][px] = vc[i][px] + adiff<NW>(nb, cy * NW + cx, (cy + 1) * NW + cx, sc);
                    else
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            const int u = (cy + QY[q]) * NW + cx + QX[q];
                            vc[i][px] = vc[i][px] + adiff<NW>(nb, u, u + NW, sc);
                        }
                }
        }
#pragma unroll
        for (int py = 0; py < PH; py++)
#pragma unroll
            for (int px = 0; px < 4; px++) {
                dist[py * 4 + px][0] = hc[py][px];      // tap (0,-1)
                dist[py * 4 + px][1] = hc[py][px + 1];  // tap (0,+1)
                dist[py * 4 + px][2] = vc[py][px];      // tap (-1,0)
                dist[py * 4 + px][3] = vc[py + 1][px];  // tap (+1,0)
            }
    } else {
    // 13-tap iteration: the four taps on the patch's own row -- (0,-1), (0,1), (0,-2), (0,2) -- are pair distances that two pixels of
    // the 4x1 patch share (dist(p, p + t) and dist(p + t, p) are the same terms in the same order): 5 + 6 chains instead of 16.
    // The other eight taps pair a pixel with one of another row, i.e. of another thread's patch.
    constexpr bool HROW = ITER == 0 && CHAINS && PH == 1;
    float h1[5], h2[6];
#pragma unroll
    for (int j = 0; j < 5; j++) h1[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; j++) h2[j] = 0.0f;
    // channels one after the other (not interleaved by the scheduler): keeps the live set under 128 VGPRs
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        // neighbourhood of this channel in registers; only the diamond of radius R around the patch is
        // ever referenced, the compiler drops the unused corner loads
        float nb[NH * NW];
        const float* pc = src + c * PLANE + (ry - R) * SW + (rx - R);
#pragma unroll
        for (int y = 0; y < NH; y++)
#pragma unroll
            for (int x = 0; x < NW; x++) nb[y * NW + x] = pc[y * SW + x];
        const float sc = ep.channel_scale[c];
#pragma unroll
        for (int py = 0; py < PH; py++)
#pragma unroll
            for (int px = 0; px < 4; px++) {
                const int cy = py + R, cx = px + R;  // centre inside nb
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (HROW && TY[t] == 0) continue;  // through h1 / h2 below
                    if (ITER == 2) {  // epfDistance2 (:657-669): single pixel
                        dist[py * 4 + px][t] =
                            dist[py * 4 + px][t] + adiff<NW>(nb, cy * NW + cx, (cy + TY[t]) * NW + cx + TX[t], sc);
                    } else {  // epfDistance1 (:638-655): 5-point cross
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            const int u = (cy + QY[q]) * NW + cx + QX[q];
                            const int v = (cy + TY[t] + QY[q]) * NW + cx + TX[t] + QX[q];
                            dist[py * 4 + px][t] = dist[py * 4 + px][t] + adiff<NW>(nb, u, v, sc);
                        }
                    }
                }
            }
        if (HROW) {
            constexpr int cy = R;
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const int u = (cy + QY[q]) * NW + (j - 1 + R) + QX[q];
                    h1[j] = h1[j] + adiff<NW>(nb, u, u + 1, sc);
                }
#pragma unroll
            for (int j = 0; j < 6; j++)
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const int u = (cy + QY[q]) * NW + (j - 2 + R) + QX[q];
                    h2[j] = h2[j] + adiff<NW>(nb, u, u + 2, sc);
                }
        }
    }
    if (HROW) {
#pragma unroll
        for (int px = 0; px < 4; px++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (TY[t] != 0) continue;
                dist[px][t] = TX[t] == -1 ? h1[px] : TX[t] == 1 ? h1[px + 1] : TX[t] == -2 ? h2[px] : h2[px + 2];
            }
    }
    }
    // weights (epfWeight, :671-679), in place of the distances
    float sumW[NP];
    bool skip[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const float s = s_inv[i];
        skip[i] = (s != s) || (s > (1.0f / 0.3f));  // :608-612
        float sw = 0.0f + 1.0f;                     // centre tap: dist 0 -> weight 1 (finite samples)
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float d = dist[i][t] * bmul[i];
            const float v = 1.0f - d * ep.sigma_scale * s;
            // max(v, 0): the reference's `v < 0 ? 0 : v` (Frame.java:676-678) for every v that is not NaN or -0. v = 1 - x is -0 for no x,
            // and NaN only where inv sigma is not finite or the samples are not -- pixels the `skip` select below replaces anyway
            const float w = __builtin_fmaxf(v, 0.0f);
            dist[i][t] = w;
            sw = sw + w;
        }
        sumW[i] = sw;
    }
    // weighted sums per channel (:615-626): tap radius RT2 only
    constexpr int RT2 = ITER == 0 ? 2 : 1;
    constexpr int MW = 4 + 2 * RT2, MH = PH + 2 * RT2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float nb[MH * MW];
        const float* pc = src + c * PLANE + (ry - RT2) * SW + (rx - RT2);
#pragma unroll
        for (int y = 0; y < MH; y++)
#pragma unroll
            for (int x = 0; x < MW; x++) nb[y * MW + x] = pc[y * SW + x];
#pragma unroll
        for (int py = 0; py < PH; py++)
#pragma unroll
            for (int px = 0; px < 4; px++) {
                const int i = py * 4 + px;
                const int cy = py + RT2, cx = px + RT2;
                float sc = 0.0f + nb[cy * MW + cx] * 1.0f;
#pragma unroll
                for (int t = 0; t < NT; t++) sc = sc + nb[(cy + TY[t]) * MW + cx + TX[t]] * dist[i][t];
#ifdef JXL_EPF_DIV_BRANCH
                res[c][i] = skip[i] ? nb[cy * MW + cx] : sc / sumW[i];
#else
                // the quotient is formed unconditionally (the empty asm keeps the optimiser from sinking the division into
                // a branch on `skip`): twelve exec-masked blocks per stage become straight-line code, and the schedulers
                // interleave the twelve dependent rcp / fma chains
                float qv = sc / sumW[i];
                asm volatile("" : "+v"(qv));
                res[c][i] = skip[i] ? nb[cy * MW + cx] : qv;
#endif
            }
    }
}

struct TileCtx {
    int ix0, iy0;  // frame coordinates of region (0,0)
    int W, H;
    bool edge;
};

// s_inv / border factor of the 4 x PH pixels of a patch
template <int PH>
__device__ __forceinline__ void patch_sigma(int ry, int rx, const TileCtx& tc, const float* __restrict__ sig, int scy0, int scx0,
                                            const EpfParams& ep, float s_inv[4 * PH], float bmul[4 * PH]) {
    const int gx0 = tc.ix0 + rx;
    // a 4-pixel run touches at most two cells: look up the first and the last, pick per pixel. Out-of-frame
    // positions (only on edge tiles: a uniform branch) get some in-range cell; they are recomputed by the mirror fix-up.
    int cxa = (gx0 >> 3) - scx0, cxb = ((gx0 + 3) >> 3) - scx0;
    if (tc.edge) {
        cxa = (min(max(gx0, 0), tc.W - 1) >> 3) - scx0;
        cxb = (min(max(gx0 + 3, 0), tc.W - 1) >> 3) - scx0;
    }
    // pixel i of the run lies in the first cell while i < 8 - (gx0 & 7); it is on an 8x8-border column when (gx0 + i) & 7 is 0 or 7:
    // bit i of 0x8181 >> (gx0 & 7)
    const int first_n = 8 - (gx0 & 7);
    const uint32_t colb = 0x8181u >> (gx0 & 7);
#pragma unroll
    for (int py = 0; py < PH; py++) {
        const int gy = tc.iy0 + ry + py;
        const bool rowb = ((gy + 1) & 7) < 2;  // gy & 7 is 7 or 0
        int crow = ((gy >> 3) - scy0) * 16;
        if (tc.edge) crow = ((min(max(gy, 0), tc.H - 1) >> 3) - scy0) * 16;
        const float sa = sig[crow + cxa], sb = sig[crow + cxb];
        const uint32_t bm = rowb ? 0xfu : colb;
#pragma unroll
        for (int px = 0; px < 4; px++) {
            // epfWeight's border factor (:672-675): border_sad_mul on 8x8-border rows/columns, else 1 (d * 1 == d)
            bmul[py * 4 + px] = (bm & (1u << px)) ? ep.border_sad_mul : 1.0f;
            s_inv[py * 4 + px] = px < first_n ? sa : sb;
        }
    }
}

// run one EPF iteration over the output region [m, IH-m) x [m, IW-m) of the tile.
// LAST: results go to the sink (colour + global store). Otherwise IN PLACE: every thread computes the one patch it owns
// into registers, the workgroup meets at a barrier (all reads of the old values done), then the patches are written
// back over the input. One LDS buffer instead of two doubles the workgroups a CU can hold.
template <int ITER, typename G, bool LAST, int PH, typename Sink>
__device__ __forceinline__ void epf_stage(float* buf, int m, const TileCtx& tc, const float* __restrict__ sig, int scy0, int scx0,
                                          const EpfParams& ep, Sink sink) {
    constexpr int SW = G::SW, PLANE = G::PLANE;
    const int rw = G::IW - 2 * m, rh = G::IH - 2 * m;
    constexpr int NTHR = 512 / PH;
    const int pcols = (rw + 3) >> 2, prows = (rh + PH - 1) / PH;
    // thread -> patch. LDS banks are (address / 4) mod 32 for ds_read_b32 / ds_read2 / ds_write, and lanes conflict within a
    // 32-lane half. A patch is 4 floats wide, so 16 patches of one row put their first words on only 8 banks: the row-major
    // assignment (lane -> 16 patches x 2 rows) made every tap read and every write-back a 2-way conflict (44 % of all LDS
    // cycles, profiles/r1). With 8 patches x 4 rows per half the rows add 0, SW, 2 SW, 3 SW: SW is odd, so they land in the
    // four residue classes mod 4 and the 32 lanes cover the 32 banks exactly once. Any constant tap offset keeps that.
    const int cblocks = (pcols + 7) >> 3, n_groups = cblocks * ((prows + 3) >> 2);
    const int q = threadIdx.x & 31;
    auto patch_of = [&](int g, int& prow, int& pcol) {
        pcol = (g % cblocks) * 8 + (q & 7);
        prow = (g / cblocks) * 4 + (q >> 3);
        return prow < prows && pcol < pcols;
    };
    if (LAST) {
        for (int g = threadIdx.x >> 5; g < n_groups; g += NTHR / 32) {
            int prow, pcol;
            if (!patch_of(g, prow, pcol)) continue;
            const int ry = m + prow * PH, rx = m + pcol * 4;
            float s_inv[4 * PH], bmul[4 * PH];
            patch_sigma<PH>(ry, rx, tc, sig, scy0, scx0, ep, s_inv, bmul);
            float res[3][4 * PH];
            epf_patch<ITER, SW, PLANE, PH>(buf, ry, rx, s_inv, bmul, ep, res);
#pragma unroll
            for (int py = 0; py < PH; py++) {
                const int y = ry + py;
                if (y < G::IH - m) sink.row4(y, rx, &res[0][py * 4], &res[1][py * 4], &res[2][py * 4], min(4, G::IW - m - rx));
            }
        }
    } else {
        // the largest region any stage sees is the 64x32 window: one patch per thread
        static_assert((64 / 4) * ((32 + PH - 1) / PH) <= NTHR, "one patch per thread");
        int prow, pcol;
        const bool act = patch_of(threadIdx.x >> 5, prow, pcol);  // 16 groups of 8 x 4 patches cover the 16 x 32 window
        const int ry = m + prow * PH, rx = m + pcol * 4;
        float res[3][4 * PH];
        if (act) {
            float s_inv[4 * PH], bmul[4 * PH];
            patch_sigma<PH>(ry, rx, tc, sig, scy0, scx0, ep, s_inv, bmul);
            epf_patch<ITER, SW, PLANE, PH>(buf, ry, rx, s_inv, bmul, ep, res);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int py = 0; py < PH; py++) {
                const int y = ry + py;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    const int x = rx + px;
                    if (y < G::IH - m && x < G::IW - m) {
#pragma unroll
                        for (int c = 0; c < 3; c++) buf[c * PLANE + y * SW + x] = res[c][py * 4 + px];
                    }
                }
            }
        }
    }
}

// after a non-final stage on an edge tile: positions of the region outside the frame take the value of
// their mirrored in-frame position (what a mirrored read of the full-frame plane would return)
template <typename G, int NTHR>
__device__ __forceinline__ void mirror_fixup(float* __restrict__ buf, int m, int rem, const TileCtx& tc) {
    const int rw = G::IW - 2 * m, rh = G::IH - 2 * m;
    __syncthreads();
    for (int i = threadIdx.x; i < rw * rh; i += NTHR) {
        const int y = m + i / rw, x = m + i % rw;
        const int gy = tc.iy0 + y, gx = tc.ix0 + x;
        const bool outside = gy < 0 || gy >= tc.H || gx < 0 || gx >= tc.W;
        // only positions within the radius the remaining stages can reach from an in-frame pixel matter
        const bool near = gy >= -rem && gy < tc.H + rem && gx >= -rem && gx < tc.W + rem;
        if (outside && near) {
            const int my = mirror_c(gy, tc.H) - tc.iy0, mx = mirror_c(gx, tc.W) - tc.ix0;
#pragma unroll
            for (int c = 0; c < 3; c++) buf[c * G::PLANE + y * G::SW + x] = buf[c * G::PLANE + my * G::SW + mx];
        }
    }
}


typedef float v2f_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f2a4 {
    float x, y;
};

// OpsinInverseMatrix.invertXYB + JXLImage.transferInPlace + ImageBuffer.castToInt0 + the global store; SK = sink kind (restore_sink.h)
template <int SK, bool NT = false>
struct OutSink {
    const FusedArgs& a;
    const TileCtx& tc;
    __device__ __forceinline__ void colour(float& v0, float& v1, float& v2) const {
        if (a.p.xyb) sink_colour(a.p.xybp, v0, v1, v2);
    }
    // one pixel at region position (y, x)
    __device__ __forceinline__ void operator()(int y, int x, float v0, float v1, float v2) const {
        const int gy = tc.iy0 + y, gx = tc.ix0 + x;
        if (gy >= tc.H || gx >= tc.W) return;
        colour(v0, v1, v2);
        sink_store_k<SK>(a, (uint32_t)(gy * tc.W + gx), v0, v1, v2);
    }
    // up to 4 consecutive pixels of a row (a patch row)
    __device__ __forceinline__ void row4(int y, int x, const float* r0, const float* r1, const float* r2, int nvalid) const {
        const int gy = tc.iy0 + y, gx = tc.ix0 + x;
        if (gy >= tc.H) return;
        const int n = min(nvalid, tc.W - gx);
        if (n <= 0) return;
        float o[3][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o[0][i] = r0[i];
            o[1][i] = r1[i];
            o[2][i] = r2[i];
            colour(o[0][i], o[1][i], o[2][i]);
        }
        const uint32_t g = (uint32_t)(gy * tc.W + gx);
        // a whole run of 4 from an even pixel index (tile origins are multiples of 62 px, patch columns of 4, frame widths of 8:
        // g is even for every full run): the kind's wide stores
        if (n == 4 && (g & 1u) == 0 && sink_store4_k<SK, NT>(a, g, o)) return;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < n) sink_store_k<SK>(a, g + i, o[0][i], o[1][i], o[2][i]);
    }
};

// SK = SK_PLAIN: float planes out, no transfer function (keeps the transfer code out of the hot variant); the other sink kinds
// fix transfer function and output format at compile time, SK_GENERIC picks them per sample at run time
template <bool GAB, int ITERS, int SK, int PH>
__device__ __forceinline__ void restore_fused_body(const FusedArgs& a) {
    using G = Geo<GAB, ITERS>;
    constexpr int NTHR = 512 / PH;
    extern __shared__ float lds[];
    float* A = lds;                   // the tile: 3 planes, every stage works in place
    float* sig = lds + 3 * G::PLANE;  // [16][16] inverse sigma of the cells under the tile
    const int W = a.W, H = a.H;
    TileCtx tc;
    tc.W = W;
    tc.H = H;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b % 8 share an L2), so give every
    // XCD one contiguous run of tiles in raster order: neighbouring tiles (shared halo reads, shared 128-byte
    // lines of the 62-wide output rows) then meet in the same L2. Speed only, never correctness.
    const int tiles_x = (W + G::OW - 1) / G::OW, tiles_y = (H + G::OH - 1) / G::OH;
    int ox, oy;
    {
        const int n_tiles = tiles_x * tiles_y;
        const int per_xcd = (n_tiles + 7) >> 3;
        const int tile = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
        if (tile >= n_tiles) return;  // uniform per workgroup, before any barrier
        ox = (tile % tiles_x) * G::OW;
        oy = (tile / tiles_x) * G::OH;
    }
    tc.ix0 = ox - G::RT;
    tc.iy0 = oy - G::RT;
    tc.edge = tc.ix0 < 0 || tc.iy0 < 0 || tc.ix0 + G::IW > W || tc.iy0 + G::IH > H;

    // inverse sigma per cell (Frame.java:552-571)
    const int scy0 = max(tc.iy0, 0) >> 3, scx0 = max(tc.ix0, 0) >> 3;
    if (ITERS > 0) {
        const int cy = scy0 + ((threadIdx.x & 255) >> 4), cx = scx0 + (threadIdx.x & 15);
        float v = 0.0f;
        if (cy < ((H + 7) >> 3) && cx < a.bw) {
            const int sharp = a.sharpness[cy * a.bw + cx] & 7;
            const float sigma = a.p.global_scale_f * a.p.sharp_lut[sharp] / (float)a.hf_mul[cy * a.bw + cx];
            v = 1.0f / sigma;
        }
        sig[threadIdx.x & 255] = v;
    }
    // load the input tile: clamped coordinates feed Gab, mirrored ones feed EPF directly. Flat row-major
    // walk with incrementally updated (y, x) and 32-bit plane offsets (scalar base + 32-bit lane offset).
    if (!tc.edge) {
        // interior tile: no coordinate fix-ups; two samples per lane and load (the tile origin is only 4-byte aligned)
        static_assert(G::IW % 2 == 0, "pairs");
        constexpr int PAIRS = G::IW / 2, TOTAL = PAIRS * G::IH;
        const uint32_t base = (uint32_t)(tc.iy0 * W + tc.ix0);
#pragma unroll
        for (int k = 0; k < (TOTAL + NTHR - 1) / NTHR; k++) {
            const int idx = (int)threadIdx.x + k * NTHR;
            if (idx < TOTAL) {
                const int y = idx / PAIRS, x = (idx - y * PAIRS) * 2;
                const uint32_t g = base + (uint32_t)(y * W + x);
                const f2a4 v0 = *reinterpret_cast<const f2a4*>(a.in[0] + g);
                const f2a4 v1 = *reinterpret_cast<const f2a4*>(a.in[1] + g);
                const f2a4 v2 = *reinterpret_cast<const f2a4*>(a.in[2] + g);
                float* d = A + y * G::SW + x;
                d[0] = v0.x;
                d[1] = v0.y;
                d[G::PLANE] = v1.x;
                d[G::PLANE + 1] = v1.y;
                d[2 * G::PLANE] = v2.x;
                d[2 * G::PLANE + 1] = v2.y;
            }
        }
    } else {
        constexpr int STEP_Y = NTHR / G::IW, STEP_X = NTHR % G::IW;
        int y = threadIdx.x / G::IW, x = threadIdx.x % G::IW;
        const float* __restrict__ in0 = a.in[0];
        const float* __restrict__ in1 = a.in[1];
        const float* __restrict__ in2 = a.in[2];
        while (y < G::IH) {
            int gy = tc.iy0 + y, gx = tc.ix0 + x;
            if (GAB) {
                gy = min(max(gy, 0), H - 1);
                gx = min(max(gx, 0), W - 1);
            } else {
                gy = mirror_c(gy, H);
                gx = mirror_c(gx, W);
            }
            const uint32_t g = (uint32_t)(gy * W + gx);
            const float v0 = in0[g], v1 = in1[g], v2 = in2[g];
            float* d = A + y * G::SW + x;
            d[0] = v0;
            d[G::PLANE] = v1;
            d[2 * G::PLANE] = v2;
            x += STEP_X;
            y += STEP_Y;
            if (x >= G::IW) {
                x -= G::IW;
                y++;
            }
        }
    }
    __syncthreads();
    float* cur = A;
    int m = 0;
    if (GAB) {  // Frame.performGabConvolution (:505-542). Lane = column (consecutive lanes = consecutive x: conflict-free
                // LDS rows), walking down a segment of rows with the 3x3 neighbourhood sliding through registers: three
                // new samples and 10 flops per output. (W + E) of a row is the first partial sum of both its own `adj`
                // and of the `diag` of the row below -- same operands, same order -- so it is formed once.
        m = 1;
        constexpr int rw = G::IW - 2, rh = G::IH - 2;
        constexpr int NCOLSEG = NTHR / rw;                         // row segments worked on at once
        constexpr int SEG = (rh + NCOLSEG - 1) / NCOLSEG;          // rows per segment
        constexpr int NSEG = (rh + SEG - 1) / SEG;
        static_assert(NSEG * rw <= NTHR, "one pass");
        const int seg = threadIdx.x / rw, x = threadIdx.x - seg * rw;
        const int y0 = seg * SEG;
        float go[3][SEG];
        if (seg < NSEG) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float* p = cur + c * G::PLANE + y0 * G::SW + x;
                float a0 = p[0], a1 = p[1], a2 = p[2];
                float b0 = p[G::SW], b1 = p[G::SW + 1], b2 = p[G::SW + 2];
                float hs_a = a0 + a2;
                const float wb = a.p.gab_base[c], wa = a.p.gab_adj[c], wd = a.p.gab_diag[c];
#pragma unroll
                for (int k = 0; k < SEG; k++) {
                    if (y0 + k < rh) {
                        const float c0 = p[(k + 2) * G::SW], c1 = p[(k + 2) * G::SW + 1], c2 = p[(k + 2) * G::SW + 2];
                        const float hs_b = b0 + b2;
                        const float adj = hs_b + a1 + c1;    // p[-1] + p[1] + p[-SW] + p[SW]
                        const float diag = hs_a + c0 + c2;   // p[-SW-1] + p[-SW+1] + p[SW-1] + p[SW+1]
                        go[c][k] = wb * b1 + wa * adj + wd * diag;
                        a1 = b1;
                        hs_a = hs_b;
                        b0 = c0; b1 = c1; b2 = c2;
                    }
                }
            }
        }
        __syncthreads();  // every read of the un-filtered tile is done: write the results over it
        if (seg < NSEG) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float* o = cur + c * G::PLANE + (y0 + 1) * G::SW + x + 1;
#pragma unroll
                for (int k = 0; k < SEG; k++)
                    if (y0 + k < rh) o[k * G::SW] = go[c][k];
            }
        }
        if (tc.edge && ITERS > 0) mirror_fixup<G, NTHR>(cur, m, G::RE, tc);
        __syncthreads();
    }

    // final sink: XYB + transfer/quantise + global store
    const OutSink<SK, (ITERS != 4)> sink{a, tc};  // (ITERS == 4: the first half of a two-launch run -- its planes are read again at once)

    if (ITERS == 0) {
        const int mm = m;
        for (int i = threadIdx.x; i < G::OW * G::OH; i += NTHR) {
            const int y = mm + i / G::OW, x = mm + i % G::OW;
            sink(y, x, cur[y * G::SW + x], cur[G::PLANE + y * G::SW + x], cur[2 * G::PLANE + y * G::SW + x]);
        }
        return;
    }
    if (ITERS == 4) {  // iteration 0 on the whole 64x32 window, straight to the sink
        m += 3;
        epf_stage<0, G, true, PH>(cur, m, tc, sig, scy0, scx0, a.p.epf[0], sink);
        return;
    }
    if (ITERS == 3) {
        m += 3;
        epf_stage<0, G, false, PH>(cur, m, tc, sig, scy0, scx0, a.p.epf[0], sink);
        if (tc.edge) mirror_fixup<G, NTHR>(cur, m, G::R1 + G::R2, tc);
        __syncthreads();
    }
    m += 2;
    if (ITERS >= 2) {
        epf_stage<1, G, false, PH>(cur, m, tc, sig, scy0, scx0, a.p.epf[1], sink);
        if (tc.edge) mirror_fixup<G, NTHR>(cur, m, G::R2, tc);
        __syncthreads();
        m += 1;
        epf_stage<2, G, true, PH>(cur, m, tc, sig, scy0, scx0, a.p.epf[2], sink);
    } else {
        epf_stage<1, G, true, PH>(cur, m, tc, sig, scy0, scx0, a.p.epf[1], sink);
    }
}

#define JXL_RESTORE_BOUNDS(ITERS, PH) __launch_bounds__(512 / PH, ITERS >= 3 ? (PH == 1 ? 4 : JXL_EPF3_PH2_WAVES) : PH == 1 ? 8 : 4)
// occupancy floor: 8 waves per SIMD (64 VGPRs); the 3-iteration variant holds 48 tap distances per patch and spilled 130
// VGPRs at that bound, so it is allowed 128 registers (4 waves per SIMD; its 43 KB tile allows 3 workgroups per CU anyway)
template <bool GAB, int ITERS, int SK, int PH>
__global__ JXL_RESTORE_BOUNDS(ITERS, PH) void k_restore_fused(const FusedArgs a) {
    restore_fused_body<GAB, ITERS, SK, PH>(a);
}

// a batch of frames in one launch (jxl_vardct_run_batch): blockIdx.y = frame, argument blocks in device memory read
// through the constant address space (uniform scalar loads, as from the kernel-argument segment)
template <bool GAB, int ITERS, int SK>
__global__ JXL_RESTORE_BOUNDS(ITERS, 1) void k_restore_fused_batch(const FusedArgs* __restrict__ args) {
    typedef const __attribute__((address_space(4))) FusedArgs* cargs;
    restore_fused_body<GAB, ITERS, SK, 1>(*(const FusedArgs*)((cargs)args + blockIdx.y));
}

template <bool GAB, int ITERS, int SK, int PH>
void launch_tph(const FusedArgs& a, hipStream_t s) {
    using G = Geo<GAB, ITERS>;
    // experiment knob: extra dynamic LDS per workgroup caps the workgroups per CU (160 KiB / size), leaving wave slots to
    // the latency-bound IDCT kernels of other frames in a batch
    static const size_t pad = getenv("JXL_RESTORE_LDS_PAD") ? (size_t)atoi(getenv("JXL_RESTORE_LDS_PAD")) : 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_restore_fused<GAB, ITERS, SK, PH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::LDS_BYTES + pad));
        attr_set = true;
    }
    const int tiles_x = (a.W + G::OW - 1) / G::OW, tiles_y = (a.H + G::OH - 1) / G::OH;
    const int n_tiles = tiles_x * tiles_y;
    const dim3 grid(((n_tiles + 7) / 8) * 8);
    // stage timing (jxl_vardct_enable_stage_timing): the KERNEL's own start and stop -- what rocprofv3 reports for it -- besides the
    // stage's events on the stream, which also hold the boundary to the launch in front (r6)
    if (g_restore_kernel_ev[0] && g_restore_kernel_ev[1])
        hipExtLaunchKernelGGL((k_restore_fused<GAB, ITERS, SK, PH>), grid, dim3(512 / PH), (uint32_t)(G::LDS_BYTES + pad), s, g_restore_kernel_ev[0],
                              g_restore_kernel_ev[1], 0u, a);
    else
        hipLaunchKernelGGL((k_restore_fused<GAB, ITERS, SK, PH>), grid, dim3(512 / PH), G::LDS_BYTES + pad, s, a);
}

template <bool GAB, int ITERS, int SK>
void launch_batch_t(const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s) {
    using G = Geo<GAB, ITERS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_restore_fused_batch<GAB, ITERS, SK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        attr_set = true;
    }
    int max_tiles = 0;
    for (int i = 0; i < n; i++)
        max_tiles = std::max(max_tiles, ((host_args[i].W + G::OW - 1) / G::OW) * ((host_args[i].H + G::OH - 1) / G::OH));
    const dim3 grid(((max_tiles + 7) / 8) * 8, n);
    hipLaunchKernelGGL((k_restore_fused_batch<GAB, ITERS, SK>), grid, dim3(512), G::LDS_BYTES, s, dev_args);
}

// one sink kind, every (Gaborish, EPF iterations) combination, 4x1 patches
template <int SK>
void launch_fused_sk(const FusedArgs& a, hipStream_t s) {
    const int it = a.p.epf_iters;
    if (a.p.gab) {
        if (it == 0) launch_tph<true, 0, SK, 1>(a, s);
        else if (it == 1) launch_tph<true, 1, SK, 1>(a, s);
        else if (it == 2) launch_tph<true, 2, SK, 1>(a, s);
        else if (it == 4) {
            if constexpr (SK == SK_PLAIN) launch_tph<true, 4, SK_PLAIN, 1>(a, s);  // (float planes only: the first half of a split run)
        } else launch_tph<true, 3, SK, 1>(a, s);
    } else {
        if (it == 0) launch_tph<false, 0, SK, 1>(a, s);
        else if (it == 1) launch_tph<false, 1, SK, 1>(a, s);
        else if (it == 2) launch_tph<false, 2, SK, 1>(a, s);
        else if (it == 4) {
            if constexpr (SK == SK_PLAIN) launch_tph<false, 4, SK_PLAIN, 1>(a, s);
        } else launch_tph<false, 3, SK, 1>(a, s);
    }
}
template <int SK>
void launch_fused_batch_sk(const FusedArgs* h, const FusedArgs* d, int n, hipStream_t s) {
    const int it = h[0].p.epf_iters;
    if (h[0].p.gab) {
        if (it == 0) launch_batch_t<true, 0, SK>(h, d, n, s);
        else if (it == 1) launch_batch_t<true, 1, SK>(h, d, n, s);
        else if (it == 2) launch_batch_t<true, 2, SK>(h, d, n, s);
        else launch_batch_t<true, 3, SK>(h, d, n, s);
    } else {
        if (it == 0) launch_batch_t<false, 0, SK>(h, d, n, s);
        else if (it == 1) launch_batch_t<false, 1, SK>(h, d, n, s);
        else if (it == 2) launch_batch_t<false, 2, SK>(h, d, n, s);
        else launch_batch_t<false, 3, SK>(h, d, n, s);
    }
}

}  // namespace
}  // namespace jxl
