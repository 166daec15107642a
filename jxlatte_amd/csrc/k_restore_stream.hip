// Register-streaming restoration + colour kernel for gfx950 (round 3):
//   Gaborish -> EPF iteration 1 -> EPF iteration 2 -> XYB -> (transfer + quantise), one read and one write of the planes,
//   NO LDS tile, NO workgroup barrier in the row loop.
//
// Replaces the same reference functions as k_restore_fused.hip (Frame.java:505-679, OpsinInverseMatrix.java:105-142,
// JXLImage.java:244-258) for the INTERIOR of a frame: every output pixel whose dependency cone (radius RT = 1 + 2 + 1)
// lies inside the frame. The ring of RT pixels along the frame edges (clamped / mirrored coordinates) stays with the
// tile kernel (launch_restore_fused_ring), which computes the same bits.
//
// Decomposition: lane = one pixel COLUMN; a wave owns a strip of 64 columns (4 halo + 56 output + 4 halo) and walks down
// a band of rows. Everything a stage needs from the rows above / below sits in rolling register windows (rings of 4 rows
// with compile-time slot numbers: the row loop is unrolled by 4); everything it needs from the columns left / right comes
// from the neighbouring lanes through DPP wave shifts (operands of the consuming add / sub / mul where the compiler can
// fold them). Why: the tile kernel recomputes the vertical halo of every stage per tile (Gab 1.32x, EPF1 1.10x), shares the
// |P(u) - P(v)| * scale terms only inside a 4x1 patch and spends a third of its instructions on LDS addressing, sigma / border
// look-ups per pixel and edge fix-ups. Here
//   * each difference term D_E(y,x) = |P(y,x) - P(y,x+1)| * s_c and D_S(y,x) = |P(y,x) - P(y+1,x)| * s_c is formed ONCE and
//     used by the ten distance sums it belongs to (5 cross positions x 2 pixels of a pair): the distance of p to its east
//     neighbour IS the distance of p+1 to its west neighbour, bit for bit (same terms, same channel-major cross-minor order);
//   * a band of R rows pays the vertical halo once (RT rows of warm-up per stage), not per 30-row tile;
//   * the per-pixel inverse sigma and border factor are per-LANE constants (the lane's x never changes) and per-row scalars.
// Exactness: the reference's strictly sequential float sums, no FMA, correctly rounded division (as k_restore_fused.hip).
//
// Measured instruction prices on gfx950 that shaped this (tools/ubench/op_rate.hip, dpp_rate.hip): v_add/sub/mul_f32 with
// VGPR or literal operands issue every 2.4-2.8 cycles per SIMD; an SGPR operand, a DPP modifier, v_cmp, v_cndmask,
// v_max_f32, v_div_scale/fmas/fixup cost 4.3-4.6; v_rcp_f32 8.2. So frame constants are copied into VGPRs once per wave.
#include "restore_sink.h"
#include <algorithm>
#include <cstdlib>

namespace jxl {
namespace {

// value held by the lane to the left (x - 1) / right (x + 1); out-of-wave lanes read 0 (halo lanes only)
__device__ __forceinline__ float lft(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float rgt(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
// a frame constant as an opaque VGPR value (an SGPR operand halves the issue rate of v_mul / v_add)
__device__ __forceinline__ float vconst(float s) {
    float v = s;
    asm volatile("" : "+v"(v));
    return v;
}

// Plane accesses go through buffer descriptors: (i) a lane that must not store gets an offset beyond the plane and the
// hardware drops the store -- no exec-masked block, no branch, so the compiler knows how many vector-memory operations are in
// flight at every wait and can leave the youngest ones outstanding (vmcnt counts loads and stores in one in-order queue: with
// the stores inside conditional blocks it waited for everything at the top of every row); (ii) 32-bit offsets.
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int rsrc_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ int bload_i(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0); }
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)off, 0, 0);
}
constexpr uint32_t kOob = 0x80000000u;  // beyond any plane (<= 2 GiB): such a store is dropped, such a load returns 0

constexpr int kHalo = 4;             // halo lanes on each side of a strip (>= RT of every variant)
constexpr int kStripOut = 64 - 2 * kHalo;  // output columns per wave
constexpr int kWavesPerWg = 4;
constexpr int kWgOut = kStripOut * kWavesPerWg;  // 224 px = 896 B = 7 x 128-byte lines per row and channel

template <bool GAB, int ITERS>
struct SGeo {
    static constexpr int RG = GAB ? 1 : 0;
    static constexpr int R1 = ITERS >= 1 ? 2 : 0;
    static constexpr int R2 = ITERS >= 2 ? 1 : 0;
    static constexpr int RT = RG + R1 + R2;
    static_assert(RT <= kHalo, "halo");
};

struct StreamGeo {
    int n_wgx;      // workgroup columns
    int n_bands;    // bands of rows
    int band_rows;  // output rows per band
};

// per-wave constants
struct Lane {
    int x;          // frame column of this lane (may be outside [0, W))
    int xl;         // clamped column used for loads
    int cxc;        // cell column (xl >> 3)
    bool store;     // lane owns output pixels
    uint32_t lo;    // byte offset of the lane's (clamped) column inside a plane row: loads
    uint32_t so;    // byte offset of the lane's column for stores, or 2^31 (out of every plane's range: the store is dropped
                    // by the buffer range check) for lanes that own no output pixel
    float fx;       // border factor of the column: border_sad_mul if x & 7 is 0 or 7, else 1
};

template <bool GAB, int ITERS>
struct State {
    float I[4][3];    // input rows (ring)
    float hs[4][3];   // Gab: W + E of an input row
    float G[4][3];    // Gab output rows
    float DE[4][3];   // EPF1: |G(y,x) - G(y,x+1)| * s_c
    float DS[4][3];   // EPF1: |G(y,x) - G(y+1,x)| * s_c
    float E1[4][3];   // EPF1 output rows
    float DE2[4][3];  // EPF2: |E1(y,x) - E1(y,x+1)| * s_c
    float prevS1, prevS2;  // south chain of the previous output row = north distance of this one
};

// 1 / sigma of cell (cy, lane's cell column) (Frame.java:552-571)
__device__ __forceinline__ float inv_sigma_cell(const FusedArgs& a, const float* slut, int cy, int cxc) {
    const int idx = cy * a.bw + cxc;
    const int sharp = a.sharpness[idx] & 7;
    const float sigma = a.p.global_scale_f * slut[sharp] / (float)a.hf_mul[idx];
    return 1.0f / sigma;
}

struct EpfK {
    float cs[3];  // channel scales (VGPR copies)
    float ss;     // sigma scale
};

// weights of the four non-centre taps (epfWeight, Frame.java:671-679) and the normalised sums (:615-630) of one pixel.
// dW..dS: distances in tap order (0,-1),(0,1),(-1,0),(1,0); C/Wv/Ev/Nv/Sv: the centre and tap samples per channel.
__device__ __forceinline__ void epf_combine(float dW, float dE, float dN, float dS, float bm, float ss, float s_inv, bool skip,
                                            const float C[3], const float Wv[3], const float Ev[3], const float Nv[3],
                                            const float Sv[3], float out[3]) {
    float w[4];
    const float d[4] = {dW, dE, dN, dS};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float v = 1.0f - ((d[t] * bm) * ss) * s_inv;
        w[t] = v < 0.0f ? 0.0f : v;
    }
    const float sw = ((((0.0f + 1.0f) + w[0]) + w[1]) + w[2]) + w[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float sc = 0.0f + C[c];  // (0f + sample * 1f): the 0f + stays, it turns a -0 sample into +0 as the reference does
        sc = sc + Wv[c] * w[0];
        sc = sc + Ev[c] * w[1];
        sc = sc + Nv[c] * w[2];
        sc = sc + Sv[c] * w[3];
        float q = sc / sw;
        asm volatile("" : "+v"(q));  // keep the division unconditional (no exec-masked blocks)
        out[c] = skip ? C[c] : q;
    }
}

struct RowCtx {
    // per-step scalars of the EPF stages
    float s1, s2;      // inverse sigma of the lane's cell at the EPF1 / EPF2 output row
    bool skip1, skip2;
    float bm1, bm2;    // border factor at the EPF1 / EPF2 output row
};

struct Planes {
    __amdgpu_buffer_rsrc_t in[3], out[3];
};

template <int P, bool GAB, int ITERS, bool PLAIN>
__device__ __forceinline__ void stream_step(State<GAB, ITERS>& st, const FusedArgs& a, const Planes& pl, const Lane& ln,
                                            const float gw[3][3], const EpfK& k1, const EpfK& k2, const float xk[15],
                                            const RowCtx& rc, int r, int out_row, bool do_store) {
    constexpr int p0 = P, p1 = (P + 3) & 3, p2 = (P + 2) & 3, p3 = (P + 1) & 3;
    // ---- Gaborish (Frame.java:505-542): input row r arrives, output row g = r - 1
    float Gn[3];
    if (GAB) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float Ic = st.I[p0][c];
            const float Il = lft(Ic), Ir = rgt(Ic);
            const float h = Il + Ir;
            st.hs[p0][c] = h;
            const float adj = (st.hs[p1][c] + st.I[p2][c]) + Ic;  // ((W + E) + N) + S
            const float diag = (st.hs[p2][c] + Il) + Ir;          // ((NW + NE) + SW) + SE
            Gn[c] = (gw[c][0] * st.I[p1][c] + gw[c][1] * adj) + gw[c][2] * diag;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) Gn[c] = st.I[p0][c];
    }
    // the slot of row r - 2 is free now: fetch row r + 2 into it
    {
        const uint32_t off = (uint32_t)min(r + 2, a.H - 1) * (uint32_t)(a.W * 4) + ln.lo;
#pragma unroll
        for (int c = 0; c < 3; c++) st.I[p2][c] = bload(pl.in[c], off);
    }
    float res[3];
    if (ITERS == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) res[c] = Gn[c];
    } else {
        // ---- EPF iteration 1 (Frame.java:583-655): row g arrives, output row e = g - 2
#pragma unroll
        for (int c = 0; c < 3; c++) {
            st.G[p0][c] = Gn[c];
            st.DE[p0][c] = fabsf(Gn[c] - rgt(Gn[c])) * k1.cs[c];
            st.DS[p0][c] = fabsf(st.G[p1][c] - Gn[c]) * k1.cs[c];  // rows g-1 | g
        }
        // distance chains of row e: channel-major, cross order (0,0),(0,-1),(0,1),(-1,0),(1,0) (Frame.java:44-48,638-655)
        float dE = 0.0f, dS = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float de = st.DE[p2][c];
            dE = c == 0 ? de : dE + de;  // 0f + term == term for what follows: dist only enters 1 - dist * k, and 1 - (+-0) == 1
            dE = dE + lft(de);
            dE = dE + rgt(de);
            dE = dE + st.DE[p3][c];
            dE = dE + st.DE[p1][c];
            const float ds = st.DS[p1][c];  // rows e | e+1
            dS = c == 0 ? ds : dS + ds;
            dS = dS + lft(ds);
            dS = dS + rgt(ds);
            dS = dS + st.DS[p2][c];  // rows e-1 | e
            dS = dS + st.DS[p0][c];  // rows e+1 | e+2
        }
        float Cc[3], Wv[3], Ev[3], Nv[3], Sv[3], e1[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Cc[c] = st.G[p2][c];
            Wv[c] = lft(Cc[c]);
            Ev[c] = rgt(Cc[c]);
            Nv[c] = st.G[p3][c];
            Sv[c] = st.G[p1][c];
        }
        epf_combine(lft(dE), dE, st.prevS1, dS, rc.bm1, k1.ss, rc.s1, rc.skip1, Cc, Wv, Ev, Nv, Sv, e1);
        st.prevS1 = dS;
        if (ITERS == 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) res[c] = e1[c];
        } else {
            // ---- EPF iteration 2 (Frame.java:657-669): row e arrives, output row f = e - 1
            float dE2 = 0.0f, dS2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                st.E1[p0][c] = e1[c];
                st.DE2[p0][c] = fabsf(e1[c] - rgt(e1[c])) * k2.cs[c];
                const float ds = fabsf(st.E1[p1][c] - e1[c]) * k2.cs[c];  // rows f | f+1
                const float de = st.DE2[p1][c];
                dE2 = c == 0 ? de : dE2 + de;
                dS2 = c == 0 ? ds : dS2 + ds;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                Cc[c] = st.E1[p1][c];
                Wv[c] = lft(Cc[c]);
                Ev[c] = rgt(Cc[c]);
                Nv[c] = st.E1[p2][c];
                Sv[c] = e1[c];
            }
            epf_combine(lft(dE2), dE2, st.prevS2, dS2, rc.bm2, k2.ss, rc.s2, rc.skip2, Cc, Wv, Ev, Nv, Sv, res);
            st.prevS2 = dS2;
        }
    }
    // ---- colour + store of row out_row
    if (PLAIN) {
        // straight-line: rows of the warm-up get an out-of-range offset like the lanes that own no pixel
        float v0 = res[0], v1 = res[1], v2 = res[2];
        if (a.p.xyb) {  // OpsinInverseMatrix.invertXYB (:131-139), constants in VGPRs
            const float gammaL = v1 + v0 + xk[12];
            const float gammaM = v1 - v0 + xk[13];
            const float gammaS = v2 + xk[14];
            const float mixL = (gammaL * gammaL) * gammaL + xk[9];
            const float mixM = (gammaM * gammaM) * gammaM + xk[10];
            const float mixS = (gammaS * gammaS) * gammaS + xk[11];
            v0 = xk[0] * mixL + xk[1] * mixM + xk[2] * mixS;
            v1 = xk[3] * mixL + xk[4] * mixM + xk[5] * mixS;
            v2 = xk[6] * mixL + xk[7] * mixM + xk[8] * mixS;
        }
        const uint32_t off = (do_store ? (uint32_t)out_row * (uint32_t)(a.W * 4) : kOob) + ln.so;
        bstore(pl.out[0], off, v0);
        bstore(pl.out[1], off, v1);
        bstore(pl.out[2], off, v2);
    } else if (do_store) {
        float v0 = res[0], v1 = res[1], v2 = res[2];
        if (a.p.xyb) sink_colour(a.p.xybp, v0, v1, v2);
        if (ln.store) sink_store<PLAIN>(a, (uint32_t)(out_row * a.W + ln.x), v0, v1, v2);
    }
}

#ifndef JXL_STREAM_WAVES
#define JXL_STREAM_WAVES 4
#endif

// the transfer / quantise variants carry the PQ / sRGB code: 3 waves per SIMD (168 registers) instead of spilling at 128
template <bool GAB, int ITERS, bool PLAIN>
__global__ __launch_bounds__(256, PLAIN ? JXL_STREAM_WAVES : 3) void k_restore_stream(const FusedArgs a, const StreamGeo sg) {
    using SG = SGeo<GAB, ITERS>;
    constexpr int RT = SG::RT;
    __shared__ float slut[8];
    if (threadIdx.x < 8) slut[threadIdx.x] = a.p.sharp_lut[threadIdx.x];
    __syncthreads();
    const int wgx = (int)blockIdx.x % sg.n_wgx, band = (int)blockIdx.x / sg.n_wgx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sx0 = wgx * kWgOut + wave * kStripOut;  // first output column of the strip
    if (sx0 >= a.W - RT) return;                      // strip without interior columns (after the only barrier)
    const int y0 = RT + band * sg.band_rows, y1 = min(y0 + sg.band_rows, a.H - RT);
    if (y0 >= y1) return;

    Lane ln;
    ln.x = sx0 + lane - kHalo;
    ln.xl = min(max(ln.x, 0), a.W - 1);
    ln.cxc = ln.xl >> 3;
    ln.store = lane >= kHalo && lane < 64 - kHalo && ln.x >= RT && ln.x < a.W - RT;
    ln.lo = (uint32_t)ln.xl * 4u;
    ln.so = ln.store ? (uint32_t)ln.x * 4u : kOob;
    const EpfParams& ep1 = a.p.epf[1];
    const EpfParams& ep2 = a.p.epf[2];
    const bool xborder = ((ln.x + 1) & 7) < 2;  // x & 7 is 7 or 0
    ln.fx = xborder ? ep1.border_sad_mul : 1.0f;  // border_sad_mul is the same for every iteration

    const uint32_t plane_bytes = (uint32_t)a.W * (uint32_t)a.H * 4u;
    Planes pl;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        pl.in[c] = make_rsrc(a.in[c], plane_bytes);
        pl.out[c] = make_rsrc(a.out[c], plane_bytes);
    }
    const int bh = (a.H + 7) >> 3;
    const __amdgpu_buffer_rsrc_t r_hf = make_rsrc(a.hf_mul, (uint32_t)(bh * a.bw) * 4u);
    const __amdgpu_buffer_rsrc_t r_sh = make_rsrc(a.sharpness, (uint32_t)(bh * a.bw) * 4u);

    float gw[3][3];
    EpfK k1, k2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        gw[c][0] = vconst(a.p.gab_base[c]);
        gw[c][1] = vconst(a.p.gab_adj[c]);
        gw[c][2] = vconst(a.p.gab_diag[c]);
        k1.cs[c] = vconst(ep1.channel_scale[c]);
        k2.cs[c] = vconst(ep2.channel_scale[c]);
    }
    k1.ss = vconst(ep1.sigma_scale);
    k2.ss = vconst(ep2.sigma_scale);
    float xk[15];  // XYB constants: matrix, opsin bias, -cbrt opsin bias
#pragma unroll
    for (int i = 0; i < 9; i++) xk[i] = PLAIN ? vconst(a.p.xybp.sm[i]) : 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        xk[9 + i] = PLAIN ? vconst(a.p.xybp.ob[i]) : 0.0f;
        xk[12 + i] = PLAIN ? vconst(a.p.xybp.cob[i]) : 0.0f;
    }

    State<GAB, ITERS> st;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            st.I[i][c] = 0.0f; st.hs[i][c] = 0.0f; st.G[i][c] = 0.0f; st.DE[i][c] = 0.0f; st.DS[i][c] = 0.0f;
            st.E1[i][c] = 0.0f; st.DE2[i][c] = 0.0f;
        }
    st.prevS1 = 0.0f;
    st.prevS2 = 0.0f;

    const int r_begin = y0 - RT, r_end = y1 + RT;  // input rows fed: [r_begin, r_end), all inside the frame
    // rows r_begin, r_begin + 1 into slots 0, 1 (step P expects row r in slot P, row r + 1 in slot P + 1)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t off = (uint32_t)min(r_begin + i, a.H - 1) * (uint32_t)(a.W * 4) + ln.lo;
#pragma unroll
        for (int c = 0; c < 3; c++) st.I[i][c] = bload(pl.in[c], off);
    }
    // Inverse sigma of the lane's cell column (Frame.java:552-571): sA for the cell row of the EPF1 output row e, sB for the
    // cell row above (the EPF2 output row e - 1 may still be in it). The raw hfMultiplier / sharpness of the cell row that
    // e + 4 lies in are requested at EVERY step (two loads of an L1-resident line, no branch around a memory operation: the
    // wait counters stay exact) and turned into 1 / sigma only at the step that enters the cell row.
    const float bsm = ep1.border_sad_mul;
    auto sigma_of = [&](int hf, int sh) {
        const float sigma = a.p.global_scale_f * slut[sh & 7] / (float)hf;
        return 1.0f / sigma;
    };
    auto cell_off = [&](int cy) { return (uint32_t)(min(max(cy, 0), bh - 1) * a.bw + ln.cxc) * 4u; };
    float sA = 0.0f, sB = 0.0f;
    int cyA = 0, hfN = 1, shN = 0;
    if (ITERS >= 1) {
        const int e0 = max(r_begin - SG::RG - 2, 0);
        cyA = e0 >> 3;
        sA = sigma_of(bload_i(r_hf, cell_off(cyA)), bload_i(r_sh, cell_off(cyA)));
        sB = sigma_of(bload_i(r_hf, cell_off(cyA - 1)), bload_i(r_sh, cell_off(cyA - 1)));
    }

    int r = r_begin;
    auto ctx_of = [&](int rr) {
        // EPF1 output row e = rr - RG - 2, EPF2 output row f = e - 1
        RowCtx rc;
        rc.s1 = rc.s2 = 0.0f;
        rc.skip1 = rc.skip2 = false;
        rc.bm1 = rc.bm2 = 1.0f;
        if (ITERS >= 1) {
            const int ec = max(rr - SG::RG - 2, 0);
            const int cy = ec >> 3;
            if (cy != cyA) {  // uniform: the step enters a new cell row; hfN / shN were requested for it one step ago
                sB = sA;
                sA = sigma_of(hfN, shN);
                cyA = cy;
            }
            hfN = bload_i(r_hf, cell_off((ec + 4) >> 3));
            shN = bload_i(r_sh, cell_off((ec + 4) >> 3));
            rc.s1 = sA;
            rc.skip1 = (sA != sA) || (sA > (1.0f / 0.3f));
            rc.bm1 = (((ec + 1) & 7) < 2) ? bsm : ln.fx;
            if (ITERS >= 2) {
                const int fc = max(ec - 1, 0);
                rc.s2 = (fc >> 3) == cy ? sA : sB;
                rc.skip2 = (rc.s2 != rc.s2) || (rc.s2 > (1.0f / 0.3f));
                rc.bm2 = (((fc + 1) & 7) < 2) ? bsm : ln.fx;
            }
        }
        return rc;
    };
#define JXL_STREAM_STEP(P)                                                                                         \
    {                                                                                                              \
        const RowCtx rc = ctx_of(r);                                                                               \
        const int orow = r - RT;                                                                                   \
        stream_step<P, GAB, ITERS, PLAIN>(st, a, pl, ln, gw, k1, k2, xk, rc, r, orow, orow >= y0);                 \
        if (++r >= r_end) break;                                                                                   \
    }
    for (;;) {
        JXL_STREAM_STEP(0)
        JXL_STREAM_STEP(1)
        JXL_STREAM_STEP(2)
        JXL_STREAM_STEP(3)
    }
#undef JXL_STREAM_STEP
}

template <bool GAB, int ITERS, bool PLAIN>
void launch_stream_t(const FusedArgs& a, hipStream_t s) {
    constexpr int RT = SGeo<GAB, ITERS>::RT;
    StreamGeo sg;
    sg.n_wgx = (a.W - RT - 1) / kWgOut + 1;  // columns RT .. W - RT - 1 live in workgroup columns 0 .. (W - RT - 1) / 224
    // one resident round: 256 CUs x (JXL_STREAM_WAVES workgroups of 4 waves) slots
    static const int slots_env = getenv("JXL_STREAM_SLOTS") ? atoi(getenv("JXL_STREAM_SLOTS")) : 0;
    const int slots = slots_env > 0 ? slots_env : 256 * (PLAIN ? JXL_STREAM_WAVES : 3);
    const int rows = a.H - 2 * RT;
    int bands = std::max(1, slots / sg.n_wgx);
    int band_rows = (rows + bands - 1) / bands;
    static const int min_rows = getenv("JXL_STREAM_MIN_ROWS") ? atoi(getenv("JXL_STREAM_MIN_ROWS")) : 16;
    band_rows = std::max(band_rows, min_rows);
    bands = (rows + band_rows - 1) / band_rows;
    sg.n_bands = bands;
    sg.band_rows = band_rows;
    hipLaunchKernelGGL((k_restore_stream<GAB, ITERS, PLAIN>), dim3(sg.n_wgx * bands), dim3(256), 0, s, a, sg);
}

template <bool GAB, int ITERS>
void launch_stream_i(const FusedArgs& a, hipStream_t s) {
    if (a.p.transfer == JXL_TRANSFER_NONE && a.p.max_value == 0) launch_stream_t<GAB, ITERS, true>(a, s);
    else launch_stream_t<GAB, ITERS, false>(a, s);
}

}  // namespace

// radius of the frame-edge ring the streaming kernel leaves to the tile kernel
int restore_stream_ring(const RestoreParams& p) {
    return (p.gab ? 1 : 0) + (p.epf_iters >= 1 ? 2 : 0) + (p.epf_iters >= 2 ? 1 : 0);
}

// true if the configuration is covered (EPF with 0..2 iterations, frame large enough to have an interior)
bool restore_stream_covers(const FusedArgs& a) {
    // r3 status: bit-exact on every parity test, but 115 us + 19 us (ring tiles) per 4K frame against the tile kernel's 95 us:
    // both kernels are bound by VALU issue (~3.4-3.8 cycles per wave-instruction at their occupancies) and this one executes
    // only 9 % fewer instructions (489 against 534 per output pixel: the 64-lane strip has 8 halo lanes, a 39-row band 8 halo
    // rows, and hipcc leaves 36 of the 38 DPP neighbour reads as separate v_mov_dpp). Off unless JXL_RESTORE_STREAM=1.
    static const int on = getenv("JXL_RESTORE_STREAM") ? atoi(getenv("JXL_RESTORE_STREAM")) : 0;
    if (!on) return false;
    if (a.p.epf_iters > 2) return false;
    const int rt = restore_stream_ring(a.p);
    if (rt == 0) return false;  // nothing but colour / transfer: the tile kernel's element-wise form is as good
    return a.W >= 64 && a.H >= 2 * rt + 8;
}

void launch_restore_stream(const FusedArgs& a, hipStream_t s) {
    const int it = a.p.epf_iters;
    if (a.p.gab) {
        if (it == 0) launch_stream_i<true, 0>(a, s);
        else if (it == 1) launch_stream_i<true, 1>(a, s);
        else launch_stream_i<true, 2>(a, s);
    } else {
        if (it == 1) launch_stream_i<false, 1>(a, s);
        else launch_stream_i<false, 2>(a, s);
    }
}

}  // namespace jxl
