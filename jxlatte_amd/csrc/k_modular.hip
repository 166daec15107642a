// Modular inverse transforms on gfx950: inverse Squeeze lifting steps, RCT, int -> float.
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/modular/ModularChannel.java:23-47    tendency
//   J/frame/modular/ModularChannel.java:361-413  inverseHorizontalSqueeze / inverseVerticalSqueeze
//   J/frame/modular/ModularStream.java:255-326   RCT
//   J/frame/Frame.java:430-455                   modular ints -> frame buffer
//
// int32 arithmetic wraps (Java): all adds/muls are done in uint32; `/` truncates toward zero.
// The squeeze recurrence is serial along the squeeze axis (left = previously OUTPUT odd sample feeds
// the non-linear tendency()), so a serial walk has only rows x channels (H) or columns x channels (V) lanes.
// The axis is therefore cut into 64-pair segments that start 16 pairs early from a guessed state and are
// verified afterwards (k_squeeze_verify; the scheme and why it is exact: jxl_internal.h, kSqueezeSeg).
//   V step: k_inv_squeeze_walk<false>, lane = column, rows walked in order: every load/store is a coalesced row segment.
//   H step: lane = row. Small steps: k_inv_squeeze_walk<true>, the same register-only walk with strided access
//           (cache-resident planes). Large steps: k_inv_hsqueeze, a wave owns 64 rows and stages 16-pair chunks of
//           avg/res through LDS so that global traffic stays row-contiguous while each lane walks its own row.
#include "jxl_internal.h"
#include "modular_tend.h"
#include <cstdlib>

namespace jxl {

// One squeeze step is ONE launch over all of its channels: blockIdx.y selects the channel descriptor.
// (SqueezeBatch is declared in jxl_internal.h.)

// inverseVerticalSqueeze (ModularChannel.java:389-413): lane = column. The avg/res rows do not depend on the
// recurrence, so they are fetched RV rows ahead (all loads of a chunk in flight) and only the short
// left -> tendency -> diff -> first/second chain is serial.
// HZ = false: inverseVerticalSqueeze, lane = column, element (k, lane) at [k * w + lane] (every access a coalesced row segment).
// HZ = true: inverseHorizontalSqueeze walked the same way, lane = row, element (lane, k) at [lane * pitch + k]: a wave's
// access touches 64 lines, but every lane then reads its own line to the end (L1 / L2 hits), no LDS transposition, no
// workgroup barriers, nothing but registers -- the form k_inv_hsqueeze (LDS-staged) is compared against in DESIGN.md 4.3.
// The fused check (SqueezeBatch::chk): this workgroup's share of an earlier step's segment boundaries. side[s][x] (the state
// segment s started its own pairs from, after its warm-up) must equal tail[s - 1][x] (the last output of segment s - 1); both
// arrays are compact [segment][lane], so boundary (s, x), s >= 1, is element i = (s - 1) * n + x of tail and n + i of side.
__device__ __forceinline__ void squeeze_fused_check(const SqueezeBatch& bt) {
    if (bt.n_chk <= 0) return;
    const int64_t wg = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
    const int64_t stride = (int64_t)gridDim.x * gridDim.y * gridDim.z * 64;
    bool bad = false;
    for (int q = 0; q < bt.n_chk; q++) {
        const SqueezeCheck ck = bt.chk[q];
        const int64_t total = (int64_t)(ck.nseg - 1) * ck.n;
        for (int64_t i = wg * 64 + threadIdx.x; i < total; i += stride) bad = bad || ck.side[ck.n + i] != ck.tail[i];
    }
    if (__builtin_expect(__any(bad), 0) && threadIdx.x == 0) atomicOr(bt.flag, 1);
}

template <bool HZ>
__global__ __launch_bounds__(64) void k_inv_squeeze_walk(const SqueezeBatch bt) {
    squeeze_fused_check(bt);
    const SqueezeDesc d = bt.d[blockIdx.y];
    const int w = d.other, ah = d.adim, rh = d.rdim;  // w: number of lanes (columns for V, rows for H)
    // strides of the walked index k and of the lane index, per array
    const int64_t ak = HZ ? 1 : w, al = HZ ? ah : 1, rk = HZ ? 1 : w, rl = HZ ? rh : 1, ok = HZ ? 1 : w, ol = HZ ? ah + rh : 1;
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    // segment of pair rows [yb, ye) stored by this wave; the walk starts kSqueezeWarm pairs earlier (jxl_internal.h)
    const int nseg = squeeze_segments(d);
    const int s = blockIdx.z;
    if (s >= nseg) return;
    const int yb = s * squeeze_seg(d);
    const int ye = nseg == 1 ? rh : min(rh, yb + squeeze_seg(d));
    const int ys = s > 0 ? yb - squeeze_warm(d) : 0;
    const int32_t* __restrict__ avg = d.a + (int64_t)x * al;
    const int32_t* __restrict__ res = d.b + (int64_t)x * rl;
    int32_t* __restrict__ out = d.o + (int64_t)x * ol;
    constexpr int RV = 8;
    static_assert(kSqueezeWarm % RV == 0 && kSqueezeSeg % RV == 0, "segment and warm-up boundaries fall on chunk boundaries");
    int32_t top = 0;
    int32_t a = rh > 0 ? avg[(int64_t)ys * ak] : 0;
    // software pipeline: the loads of chunk k+1 are issued before the serial chain of chunk k runs, so a lone wave
    // does not sit in s_waitcnt for a full memory round trip per chunk
    int32_t rr_n[RV], na_n[RV];
    auto fetch = [&](int yb_, int32_t* r_, int32_t* n_) {
        if (HZ && yb_ + RV < ah && yb_ + RV <= rh) {
            // the lane's own run of 8 + 8 consecutive samples as four 16-byte loads (4-byte aligned: rows start anywhere):
            // a quarter of the requests of eight dword loads per array, and each touches the lane's cache line once
            struct __attribute__((packed, aligned(4))) i4 { int32_t v[4]; };
            const i4 r0 = *reinterpret_cast<const i4*>(res + yb_), r1 = *reinterpret_cast<const i4*>(res + yb_ + 4);
            const i4 a0 = *reinterpret_cast<const i4*>(avg + yb_ + 1), a1 = *reinterpret_cast<const i4*>(avg + yb_ + 5);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r_[i] = r0.v[i]; r_[4 + i] = r1.v[i];
                n_[i] = a0.v[i]; n_[4 + i] = a1.v[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < RV; i++) {
            const int y = yb_ + i;
            r_[i] = y < rh ? res[(int64_t)y * rk] : 0;
            n_[i] = (y < rh && y + 1 < ah) ? avg[(int64_t)(y + 1) * ak] : 0;
        }
    };
    fetch(ys, rr_n, na_n);
    for (int y0 = ys; y0 < ye; y0 += RV) {
        if (y0 == yb && s > 0) d.side[(int64_t)s * w + x] = top;  // the state this wave starts its own segment from
        const bool keep = y0 >= yb;                                // warm-up chunks store nothing
        int32_t rr[RV], na[RV];
#pragma unroll
        for (int i = 0; i < RV; i++) {
            rr[i] = rr_n[i];
            na[i] = na_n[i];
        }
        if (y0 + RV < ye) fetch(y0 + RV, rr_n, na_n);
        // everything that does not depend on the recurrence first
        int32_t av[RV + 1];
        av[0] = a;
#pragma unroll
        for (int i = 0; i < RV; i++) {
            const int y = y0 + i;
            av[i + 1] = y + 1 < ah ? na[i] : av[i];
        }
        if (y0 + RV <= ye) {  // full chunk: no guards on the serial chain, stores after it
            int32_t o1[RV], o2[RV];
            // the first pair of a column uses its own average as `left`; a warm-up start guesses the same
            const int32_t top0 = y0 > ys ? top : av[0];
            top = top0;
            SqueezeRange rg;
            rg.init(av[0]);
#pragma unroll
            for (int i = 0; i < RV; i++) rg.add(av[i + 1], rr[i]);
            if (__builtin_expect(__all(rg.ok(top0)), 1)) {  // sign-normalised short form (modular_tend.h)
                TendN tf[RV];
#pragma unroll
                for (int i = 0; i < RV; i++) tf[i] = tend_n_pre(av[i], av[i + 1], rr[i]);
#pragma unroll
                for (int i = 0; i < RV; i++) top = squeeze_pair_n(top, tf[i], o1[i], o2[i]);
            } else {  // operands near the int32 limits: exact long form for this chunk
#pragma unroll
                for (int i = 0; i < RV; i++) top = squeeze_pair_exact(top, av[i], av[i + 1], rr[i], o1[i], o2[i]);
            }
            if (keep && HZ) {  // 16 consecutive outputs of the lane's row: four 16-byte stores
                struct __attribute__((packed, aligned(4))) i4 { int32_t v[4]; };
#pragma unroll
                for (int i = 0; i < RV; i += 2) {
                    i4 t;
                    t.v[0] = o1[i]; t.v[1] = o2[i]; t.v[2] = o1[i + 1]; t.v[3] = o2[i + 1];
                    *reinterpret_cast<i4*>(out + 2 * (y0 + i)) = t;
                }
            } else if (keep) {
#pragma unroll
                for (int i = 0; i < RV; i++) {
                    out[(int64_t)(2 * (y0 + i)) * ok] = o1[i];
                    out[(int64_t)(2 * (y0 + i) + 1) * ok] = o2[i];
                }
            }
            a = av[RV];
        } else {
#pragma unroll
            for (int i = 0; i < RV; i++) {
                const int y = y0 + i;
                if (y < ye) {
                    const int32_t t = y > ys ? top : av[i];
                    const int32_t diff = wadd(rr[i], tend_apply(t, tend_pre(av[i], av[i + 1])));
                    const int32_t first = wadd(av[i], diff / 2);
                    const int32_t second = wsub(first, diff);
                    if (keep) {
                        out[(int64_t)(2 * y) * ok] = first;
                        out[(int64_t)(2 * y + 1) * ok] = second;
                    }
                    top = second;
                    a = av[i + 1];
                }
            }
        }
    }
    if (d.tail && s + 1 < nseg) d.tail[(int64_t)s * w + x] = top;  // last output of this segment: what segment s + 1 is checked against
    if (s == 0 && ah > rh) out[(int64_t)(2 * rh) * ok] = avg[(int64_t)rh * ak];
}

// Check of the segmented walk (jxl_internal.h): lane = column (V) or row (H). A boundary is good when the state the
// segment's wave reached after its warm-up equals the last output of the previous segment; the lane redoes its
// column / row serially from the first bad boundary (everything before it is exact by induction).
__global__ __launch_bounds__(64) void k_squeeze_verify(const SqueezeBatch bt) {
    const SqueezeDesc d = bt.d[blockIdx.y];
    const int nseg = squeeze_segments(d);
    if (nseg <= 1) return;
    const int i = blockIdx.x * 64 + threadIdx.x;  // column (V) or row (H)
    if (i >= d.other) return;
    const int n = d.other, adim = d.adim, rdim = d.rdim;
    const int32_t* __restrict__ avg = d.a;
    const int32_t* __restrict__ res = d.b;
    int32_t* out = d.o;
    const bool hz = bt.horizontal != 0;
    const int ow = adim + rdim;
    // element k of the squeezed axis of lane i
    auto at_a = [&](int k) { return hz ? avg[(int64_t)i * adim + k] : avg[(int64_t)k * n + i]; };
    auto at_r = [&](int k) { return hz ? res[(int64_t)i * rdim + k] : res[(int64_t)k * n + i]; };
    auto po = [&](int k) { return hz ? out + (int64_t)i * ow + k : out + (int64_t)k * n + i; };
    int bad = 0;
    for (int s = nseg - 1; s >= 1; s--)
        if (d.side[(int64_t)s * n + i] != (d.tail ? d.tail[(int64_t)(s - 1) * n + i] : *po(2 * s * squeeze_seg(d) - 1))) bad = s;
    if (__builtin_expect(bad == 0, 1)) return;
    if (bt.flag) {  // report only: the following steps have already consumed this step's output
        atomicOr(bt.flag, 1);
        return;
    }
    int32_t left = *po(2 * bad * squeeze_seg(d) - 1);
    for (int k = bad * squeeze_seg(d); k < rdim; k++) {
        const int32_t a = at_a(k);
        const int32_t nx = k + 1 < adim ? at_a(k + 1) : a;
        const int32_t diff = wadd(at_r(k), tendency(left, a, nx));
        const int32_t first = wadd(a, diff / 2);
        const int32_t second = wsub(first, diff);
        *po(2 * k) = first;
        *po(2 * k + 1) = second;
        left = second;
    }
}

// inverseHorizontalSqueeze (ModularChannel.java:361-387): lane = row. 64-row x 64-column chunks of avg/res are
// staged through LDS (global traffic stays row-contiguous, the whole chunk's loads are issued before the first
// LDS write); each lane then walks its own row, four columns per step with the LDS reads hoisted ahead of the
// serial chain; outputs overwrite the consumed inputs in LDS and leave as contiguous rows.
__global__ __launch_bounds__(64) void k_inv_hsqueeze(const SqueezeBatch bt) {
    squeeze_fused_check(bt);
    const SqueezeDesc d = bt.d[blockIdx.y];
    const int aw = d.adim, rw = d.rdim, h = d.other;
    const int y0 = blockIdx.x * 64;
    if (y0 >= h) return;
    // segment of pairs [xb, xe) stored by this wave; the walk starts kSqueezeWarm pairs earlier (jxl_internal.h)
    const int nseg = squeeze_segments(d);
    const int s = blockIdx.z;
    if (s >= nseg) return;
    const int xb = s * squeeze_seg(d);
    const int xe = nseg == 1 ? rw : min(rw, xb + squeeze_seg(d));
    const int xs = s > 0 ? xb - squeeze_warm(d) : 0;
    const int32_t* __restrict__ avg = d.a;
    const int32_t* __restrict__ res = d.b;
    int32_t* __restrict__ out = d.o;
    // 64 rows x CW pairs per LDS chunk. CW = 16 (8.7 KB per wave, 18 waves per CU) instead of 64 (33 KB, 4 waves per CU):
    // the segmented walk has thousands of waves per step, and the LDS chunk was what kept all but 4 per CU waiting
    // (8K image: 1.47 ms with 64, 0.97 ms with 32, 0.84 ms with 16)
    constexpr int CW = 16, LDW = CW + 1, RPI = 64 / CW;  // row stride; rows covered by one wave-wide load
    __shared__ int32_t sA[64 * LDW];  // avg chunk [row][col]; overwritten in place by the even outputs
    __shared__ int32_t sR[64 * LDW];  // res chunk [row][col]; overwritten in place by the odd outputs
    const int lc = threadIdx.x % CW, lr = threadIdx.x / CW;
    const int lane = threadIdx.x;
    const int rows = min(64, h - y0);
    const int ow = aw + rw;
    int32_t left = 0;
    for (int x0 = xs; x0 < xe;) {
        const bool warm = x0 < xb;  // the warm-up chunk: walked, not stored
        const int cols = warm ? xb - x0 : min(CW, xe - x0);
        __syncthreads();
        if (rows == 64 && cols == CW) {  // fast path: all row loads in flight, then the LDS writes
            int32_t ta[CW], tr[CW];
#pragma unroll
            for (int k = 0; k < CW; k++) {
                ta[k] = avg[(int64_t)(y0 + k * RPI + lr) * aw + x0 + lc];
                tr[k] = res[(int64_t)(y0 + k * RPI + lr) * rw + x0 + lc];
            }
#pragma unroll
            for (int k = 0; k < CW; k++) {
                sA[(k * RPI + lr) * LDW + lc] = ta[k];
                sR[(k * RPI + lr) * LDW + lc] = tr[k];
            }
        } else {
            for (int r = lr; r < rows; r += RPI) {
                if (lc < cols) {
                    sA[r * LDW + lc] = avg[(int64_t)(y0 + r) * aw + x0 + lc];
                    sR[r * LDW + lc] = res[(int64_t)(y0 + r) * rw + x0 + lc];
                }
            }
        }
        __syncthreads();
        if (lane < rows) {
            const int64_t rowA = (int64_t)(y0 + lane) * aw;
            const bool has_next = x0 + cols < aw;  // avg[x0 + cols]: first avg of the next chunk or the odd tail
            const int32_t a_next_chunk = has_next ? avg[rowA + x0 + cols] : 0;
            int32_t* pa = sA + lane * LDW;
            int32_t* pr = sR + lane * LDW;
            constexpr int U = 8;  // columns per step: LDS reads and the a-independent part of tendency() run ahead
            if ((cols & (U - 1)) == 0) {
                // whole steps (full chunks and the warm-up chunk): straight-line code, no per-column guards on the chain
                for (int i0 = 0; i0 < cols; i0 += U) {
                    int32_t va[U + 1], vr[U];
#pragma unroll
                    for (int j = 0; j <= U; j++) va[j] = pa[i0 + j];  // column `cols` is stale or padding: replaced below
#pragma unroll
                    for (int j = 0; j < U; j++) vr[j] = pr[i0 + j];
                    if (i0 + U == cols) va[U] = has_next ? a_next_chunk : va[U - 1];  // x + 1 < orig.width ? orig[x+1] : avg
                    int32_t o1[U], o2[U];
                    const int32_t left0 = (x0 + i0) > xs ? left : va[0];  // the first pair of a row uses its own average; a warm-up start guesses the same
                    left = left0;
                    SqueezeRange rg;
                    rg.init(va[0]);
#pragma unroll
                    for (int j = 0; j < U; j++) rg.add(va[j + 1], vr[j]);
                    if (__builtin_expect(__all(rg.ok(left0)), 1)) {  // sign-normalised short form (modular_tend.h)
                        TendN tf[U];
#pragma unroll
                        for (int j = 0; j < U; j++) tf[j] = tend_n_pre(va[j], va[j + 1], vr[j]);
#pragma unroll
                        for (int j = 0; j < U; j++) left = squeeze_pair_n(left, tf[j], o1[j], o2[j]);
                    } else {  // operands near the int32 limits: exact long form
#pragma unroll
                        for (int j = 0; j < U; j++) left = squeeze_pair_exact(left, va[j], va[j + 1], vr[j], o1[j], o2[j]);
                    }
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        pa[i0 + j] = o1[j];  // a, residu of these columns are consumed; the next ones are read afterwards
                        pr[i0 + j] = o2[j];
                    }
                }
            } else {
                for (int i0 = 0; i0 < cols; i0 += U) {
                    int32_t va[U + 1], vr[U];
                    TendPre tp[U];
#pragma unroll
                    for (int j = 0; j <= U; j++) va[j] = pa[min(i0 + j, CW)];
#pragma unroll
                    for (int j = 0; j < U; j++) vr[j] = pr[min(i0 + j, CW - 1)];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int i = i0 + j;
                        if (i + 1 >= cols) va[j + 1] = has_next ? a_next_chunk : va[j];
                        tp[j] = tend_pre(va[j], va[j + 1]);
                    }
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int i = i0 + j;
                        const int32_t l = (x0 + i) > xs ? left : va[j];
                        const int32_t diff = wadd(vr[j], tend_apply(l, tp[j]));
                        const int32_t first = wadd(va[j], diff / 2);
                        const int32_t second = wsub(first, diff);
                        if (i < cols) {
                            pa[i] = first;
                            pr[i] = second;
                            left = second;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (warm) {
            if (lane < rows) d.side[(int64_t)s * h + y0 + lane] = left;  // the state this wave starts its own segment from
        } else {
            constexpr int OPR = 2 * CW, RPW = 64 / OPR;  // outputs per row of the chunk; rows one wave-wide store covers
            const int j0 = lane % OPR;
            for (int r = lane / OPR; r < rows; r += RPW) {
                const int64_t ro = (int64_t)(y0 + r) * ow + 2 * x0;
                if (j0 < 2 * cols) out[ro + j0] = (j0 & 1) ? sR[r * LDW + (j0 >> 1)] : sA[r * LDW + (j0 >> 1)];
            }
        }
        x0 += cols;
    }
    if (d.tail && s + 1 < nseg && lane < rows) d.tail[(int64_t)s * h + y0 + lane] = left;  // last output of this segment
    if (s == 0 && aw > rw && lane < rows) out[(int64_t)(y0 + lane) * ow + 2 * rw] = avg[(int64_t)(y0 + lane) * aw + rw];
}

// ---- the coarse levels as ONE launch ------------------------------------------------------------------------------------
// The first inverse steps of a squeeze plan work on a few thousand samples each (the default plan of a 1080p image starts
// at 8 x 5 and doubles one axis per step): as launches of their own they cost a launch boundary each and nothing else
// (measured round 1: ~30 dependent launches of 5-20 us = the whole 0.22 ms of a 1080p image). In the in-place part of a plan
// channel i of step k+1 takes the output of channel i of step k as its averages and an untouched input channel as its
// residuals, so a workgroup can run one channel through a whole run of steps by itself: workgroup = channel slot, lanes =
// rows (H) / columns (V), every lane walking its row / column serially from the true start (no segments, nothing to
// verify), a workgroup barrier between steps (the planes are a few KB: they stay in the CU's L1 / the XCD's L2).
__global__ __launch_bounds__(256) void k_squeeze_chain(const SqueezeBatch* __restrict__ steps, int n_steps) {
    const int slot = blockIdx.x;
    for (int k = 0; k < n_steps; k++) {
        const SqueezeBatch& bt = steps[k];
        if (slot < bt.n) {
            const SqueezeDesc d = bt.d[slot];
            const bool hz = bt.horizontal != 0;
            const int n = d.other, adim = d.adim, rdim = d.rdim;
            const int ow = adim + rdim;
            for (int i = threadIdx.x; i < n; i += 256) {
                // element j of the squeezed axis of lane i
                const int64_t ab = hz ? (int64_t)i * adim : i, as = hz ? 1 : n;
                const int64_t rb = hz ? (int64_t)i * rdim : i, rs = hz ? 1 : n;
                const int64_t ob = hz ? (int64_t)i * ow : i, os = hz ? 1 : n;
                int32_t a = rdim > 0 || adim > 0 ? d.a[ab] : 0;
                int32_t left = a;  // the first pair uses its own average (ModularChannel.java:370, :399)
                // chunks of 8 pairs: the averages and residuals of a chunk do not depend on the recurrence, so they are all
                // requested before its serial chain runs, and the next chunk's before that (a dependent L2 round trip per
                // pair made this kernel slower than the eleven launches it replaces)
                constexpr int RV = 8;
                int32_t rn[RV], an[RV];
                auto fetch = [&](int j0, int32_t* r_, int32_t* a_) {
#pragma unroll
                    for (int q = 0; q < RV; q++) {
                        const int j = j0 + q;
                        r_[q] = j < rdim ? d.b[rb + (int64_t)j * rs] : 0;
                        a_[q] = j + 1 < adim ? d.a[ab + (int64_t)(j + 1) * as] : 0;
                    }
                };
                fetch(0, rn, an);
                for (int j0 = 0; j0 < rdim; j0 += RV) {
                    int32_t rr[RV], na[RV];
#pragma unroll
                    for (int q = 0; q < RV; q++) {
                        rr[q] = rn[q];
                        na[q] = an[q];
                    }
                    if (j0 + RV < rdim) fetch(j0 + RV, rn, an);
                    // r5: whole chunks whose samples pass the range guard walk with the sign-normalised short form (modular_tend.h:
                    // 13 instead of ~25 dependent instructions per pair); the guard is per lane (nothing here is shared between
                    // lanes), a chunk that fails it and the ragged last chunk take the reference's form
                    SqueezeRange rg;
                    rg.init(a);
#pragma unroll
                    for (int q = 0; q < RV; q++) rg.add(na[q], rr[q]);
                    if (j0 + RV < adim && j0 + RV <= rdim && rg.ok(left)) {
#pragma unroll
                        for (int q = 0; q < RV; q++) {
                            const int j = j0 + q;
                            int32_t first, second;
                            left = squeeze_pair_n(left, tend_n_pre(a, na[q], rr[q]), first, second);
                            d.o[ob + (int64_t)(2 * j) * os] = first;
                            d.o[ob + (int64_t)(2 * j + 1) * os] = second;
                            a = na[q];
                        }
                        continue;
                    }
#pragma unroll
                    for (int q = 0; q < RV; q++) {
                        const int j = j0 + q;
                        if (j < rdim) {
                            const int32_t nx = j + 1 < adim ? na[q] : a;
                            const int32_t diff = wadd(rr[q], tendency(left, a, nx));
                            const int32_t first = wadd(a, diff / 2);
                            const int32_t second = wsub(first, diff);
                            d.o[ob + (int64_t)(2 * j) * os] = first;
                            d.o[ob + (int64_t)(2 * j + 1) * os] = second;
                            left = second;
                            a = nx;
                        }
                    }
                }
                if (adim > rdim) d.o[ob + (int64_t)(2 * rdim) * os] = d.a[ab + (int64_t)rdim * as];
            }
        }
        __syncthreads();  // (workgroup-scope release / acquire of the global stores above: one CU, one L1)
    }
}

// (r4: the same chain with the channel's image kept in LDS from step to step -- residual plane in one coalesced sweep, walk from
// LDS, image out in one sweep -- was built and measured: 0.1777 ms per 1080p image against 0.1784, and with the two following
// steps of 60 / 67 pairs taken into the chain 0.1995. The chain is bound by the ~100 dependent instructions per pair that ONE wave
// issues at ~6 cycles each, not by its memory round trips. Not kept.)
void launch_squeeze_chain(const SqueezeBatch* dev_steps, int n_steps, int n_slots, hipStream_t s) {
    if (n_steps <= 0 || n_slots <= 0) return;
    hipLaunchKernelGGL(k_squeeze_chain, dim3(n_slots), dim3(256), 0, s, dev_steps, n_steps);
}

static bool squeeze_grid(const SqueezeBatch& bt, int& maxdim, int& nseg) {
    maxdim = 0;
    nseg = 1;
    for (int i = 0; i < bt.n; i++) {
        maxdim = bt.d[i].other > maxdim ? bt.d[i].other : maxdim;
        nseg = squeeze_segments(bt.d[i]) > nseg ? squeeze_segments(bt.d[i]) : nseg;
    }
    return bt.n > 0 && maxdim > 0;
}

void launch_squeeze_walk(const SqueezeBatch& bt, hipStream_t s) {
    int maxdim, nseg;
    if (!squeeze_grid(bt, maxdim, nseg)) return;
    const dim3 grid((maxdim + 63) / 64, bt.n, nseg);
    // H steps: the register walk (lane = row, strided access) while the step's planes stay cache-resident, the LDS-staged
    // kernel (coalesced rows) beyond that. JXL_HSQUEEZE_WALK_MAX overrides the threshold (output samples of the step).
    const char* wm = getenv("JXL_HSQUEEZE_WALK_MAX");  // read per launch: the parity tests force either kernel
    const int64_t walk_max = wm ? atoll(wm) : (int64_t)8 << 20;  // 8 Mi samples: measured crossover (1080p steps below, 8K steps above)
    int64_t samples = 0;
    for (int i = 0; i < bt.n; i++) samples += (int64_t)bt.d[i].other * (bt.d[i].adim + bt.d[i].rdim);
    const bool h_lds = samples > walk_max;
    if (bt.horizontal && h_lds) hipLaunchKernelGGL(k_inv_hsqueeze, grid, dim3(64), 0, s, bt);
    else if (bt.horizontal) hipLaunchKernelGGL(k_inv_squeeze_walk<true>, grid, dim3(64), 0, s, bt);
    else hipLaunchKernelGGL(k_inv_squeeze_walk<false>, grid, dim3(64), 0, s, bt);
}

void launch_squeeze_verify(const SqueezeBatch& bt, hipStream_t s) {
    int maxdim, nseg;
    if (!squeeze_grid(bt, maxdim, nseg) || nseg <= 1) return;
    hipLaunchKernelGGL(k_squeeze_verify, dim3((maxdim + 63) / 64, bt.n), dim3(64), 0, s, bt);
}

bool squeeze_can_fuse_check(const SqueezeBatch& bt) {
    bool any = false;
    for (int i = 0; i < bt.n; i++) {
        if (squeeze_segments(bt.d[i]) <= 1) continue;
        if (!bt.d[i].tail || !bt.d[i].side) return false;
        any = true;
    }
    return any;
}

void launch_squeeze_batch(const SqueezeBatch& bt, hipStream_t s, hipStream_t check_stream, hipEvent_t ev) {
    int maxdim, nseg;
    if (!squeeze_grid(bt, maxdim, nseg)) return;
    launch_squeeze_walk(bt, s);
    if (nseg > 1) {
        hipStream_t vs = s;
        if (bt.flag && check_stream && ev) {
            (void)hipEventRecord(ev, s);
            (void)hipStreamWaitEvent(check_stream, ev, 0);
            vs = check_stream;
        }
        launch_squeeze_verify(bt, vs);
    }
}

void launch_inv_hsqueeze(const int32_t* avg, int aw, const int32_t* res, int rw, int h, int32_t* out, hipStream_t s) {
    if (h <= 0 || aw + rw <= 0) return;
    SqueezeBatch bt{};
    bt.n = 1;
    bt.horizontal = 1;
    bt.d[0] = SqueezeDesc{avg, res, out, aw, rw, h, nullptr, 0, 0};
    launch_squeeze_batch(bt, s);
}

void launch_inv_vsqueeze(const int32_t* avg, int ah, const int32_t* res, int rh, int w, int32_t* out, hipStream_t s) {
    if (w <= 0 || ah + rh <= 0) return;
    SqueezeBatch bt{};
    bt.n = 1;
    bt.horizontal = 0;
    bt.d[0] = SqueezeDesc{avg, res, out, ah, rh, w, nullptr, 0, 0};
    launch_squeeze_batch(bt, s);
}

// ModularStream.java:270-324; the channel permutation (:325-326) is applied by the host as pointer shuffling
__global__ __launch_bounds__(256) void k_rct(int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int type) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        switch (type) {
        case 1: v2[i] = wadd(v2[i], v0[i]); break;
        case 2: v1[i] = wadd(v1[i], v0[i]); break;
        case 3: { const int32_t a = v0[i]; v2[i] = wadd(v2[i], a); v1[i] = wadd(v1[i], a); break; }
        case 4: v1[i] = wadd(v1[i], wadd(v0[i], v2[i]) >> 1); break;
        case 5: { const int32_t a = v0[i]; const int32_t ac = wadd(a, v2[i]); v1[i] = wadd(v1[i], wadd(a, ac) >> 1); v2[i] = ac; break; }
        case 6: {
            const int32_t b = v1[i], c = v2[i];
            const int32_t tmp = wsub(v0[i], c >> 1);
            const int32_t f = wsub(tmp, b >> 1);
            v0[i] = wadd(f, b);
            v1[i] = wadd(c, tmp);
            v2[i] = f;
            break;
        }
        default: break;
        }
    }
}

void launch_rct(int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int type, hipStream_t s) {
    if (n <= 0 || type == 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_rct, dim3(grid), dim3(256), 0, s, v0, v1, v2, n, type);
}

// Frame.java:437-448: out = scale * (a [+ b]) with the int sum wrapping, int -> float conversion first
__global__ __launch_bounds__(256) void k_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale,
                                                          float* out) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = b ? scale * (float)wadd(a[i], b[i]) : scale * (float)a[i];
}

void launch_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale, float* out, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_modular_to_float, dim3(grid), dim3(256), 0, s, a, b, n, scale, out);
}

}  // namespace jxl
