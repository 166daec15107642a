// Fused inverse squeeze pair on gfx950: inverseVerticalSqueeze followed by inverseHorizontalSqueeze of the same channels
// (J/frame/modular/ModularChannel.java:361-413, applied back to back by J/frame/modular/ModularStream.java:229-254) in ONE
// launch. The V output -- half of all samples the two steps move -- never reaches HBM: per level of the squeeze pyramid the
// pair reads avg (1/4) + V residuals (1/4) + H residuals (1/2) and writes the level (1) = 2 x its samples, against 3 x for two
// launches.
//
// One wave = one tile = 64 output rows (32 V pairs: a "stripe") x one H segment, walked left to right in chunks of CW H pairs
// (CW = 16: 32 output columns = one 128-byte line per row; CW = 32: two). Per chunk:
//   V pass   lane (q, j) = (part of the stripe, column): CW V-output columns x 64 / CW parts of 32 CW / 64 pairs, each part
//            started 8 pairs early from a guessed state (the recurrence forgets its start within a few pairs: jxl_internal.h).
//            Inputs straight from global memory (row pieces of 4 CW bytes), outputs into the LDS image A[row][column].
//   H pass   lane = row: reads its row of A and of the H residuals R (staged through LDS by coalesced 16-byte loads), walks CW
//            pairs serially from the chain state it carries in a register from chunk to chunk, and leaves every pair's two
//            outputs in the two LDS slots the pair has just consumed.
//   output   whole row pieces (8 CW bytes) leave LDS as 8-byte stores (lane = (row, pair)).
// CW = 32 (r5, the big steps): 128-byte input pieces and 256-byte output pieces -- the access shape of CW = 16 tops out at
// ~3.8 TB/s of HBM traffic on this part, CW = 32 at ~4.7 (tools/ubench/vh_pattern.hip) -- and 1.5 x instead of 2 x redundant V
// pairs (two halves of 16 + 8 instead of four quarters of 8 + 8), for 16.6 KB of LDS per wave instead of 8.4.
// The V pass of a chunk produces columns c0+1 .. c0+CW (the H pair at column c needs column c + 1 as its `next average`); column
// c0 itself is the last column of the previous chunk and is handed on in a register. The chunks start at pairs 1, 1 + CW, ... and
// every plane has a leading pad (kVhPad), so that V-input pieces, H-residual pieces and output pieces all start on a line.
//
// Guessed states and their verification (results are bit-identical to the serial walk whenever nothing is reported):
//   * H chain at a segment start: the segment walks one chunk (15 pairs) early and stores nothing for it; side_h / tail_h;
//   * V chain at the top of a stripe: side_v / tail_v (only the columns of non-warm-up chunks are recorded: the V columns a
//     warm-up chunk computes feed nothing but the H warm-up, which has its own check);
//   * V chain at the quarter boundaries inside a stripe: compared inside the wave (lane - 16 holds the previous quarter).
// The compact side / tail arrays are compared in the prologue of a LATER launch (SqueezeCheck) or by k_squeeze_check; any
// mismatch sets VHBatch::flag and the host runs the plan again with the one-step kernels in order (host.hip, mod_settle).
//
// LDS: 64 rows x (2 CW + 1) dwords per wave (A: slots 0..CW, R: CW+1..2CW; pair i leaves its outputs in A[i] and R[i]): 8448 B
// (19 waves per CU) for CW = 16, 16640 B (9) for CW = 32. The row stride is odd: lane = row accesses are conflict-free.
#include "jxl_internal.h"
#include "modular_tend.h"
#include <cstdlib>
#include <type_traits>

namespace jxl {
namespace {

#ifdef JXL_VH_ABL  // timing-only ablations (experiment builds, tools/archive/r5_vh_abl.sh): 1 = no loads, 2 = no stores (zero-size descriptors:
                   // the range check drops the accesses, the instruction stream stays), set per launch from the environment
__device__ int g_vh_abl;
#define VH_ABL(bit) (g_vh_abl & (bit))
#else
#define VH_ABL(bit) 0
#endif

constexpr int VH_ROWS = 64;  // output rows per tile
constexpr int VH_WP = 8;     // warm-up pairs in front of every part of a V pass

// geometry of the two chunk widths
template <int CW>
struct VHG {
    static constexpr int NQ = 64 / CW;      // parts of the stripe in a V pass (lane = (part, column))
    static constexpr int KP = 32 / NQ;      // pairs a part keeps
#ifdef JXL_VH_LDPAD
    static constexpr int LD = 2 * CW + 1 + JXL_VH_LDPAD;
#else
    static constexpr int LD = 2 * CW + 1;   // LDS row stride in dwords
#endif
    static constexpr int R0 = CW + 1;       // first H-residual slot of an LDS row
    static constexpr int HL = CW / 4;       // lanes per row of a 16-byte H-residual load
    static constexpr int HN = 64 / (64 / HL);  // such loads per chunk ( = HL )
};

// Orders this wave's LDS traffic: what other LANES wrote before it is visible to every read after it. One wave per workgroup, so no
// s_barrier is needed -- the LDS queue of a wave is in order -- but the COMPILER must not move an LDS access across it either, and a
// workgroup-scope fence does not promise that here: with a flat workgroup size of 64 the compiler may drop it (one wave: nothing
// to synchronise with, it reasons), after which a lane's reads of slots other lanes wrote are free to float above those writes.
// The first r5 build did exactly that after a refactoring changed the schedule: a handful of samples per image read stale
// LDS, differently from run to run. An asm statement that clobbers memory is a barrier the optimiser cannot reason away.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Plane accesses go through buffer descriptors: 32-bit offsets (a uniform row term in an SGPR, the lane's term in one VGPR that
// is the same for every row of a pass) and the hardware range check -- a row above the first or below the last reads as 0 and a
// store below the last row is dropped, so the edge tiles need no clamped addresses.
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct VHRsrc {
    __amdgpu_buffer_rsrc_t va, vb, hb, o;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vh_rsrc(const void* p, int64_t samples) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(uint32_t)(4 * samples), 0x00020000);
}
constexpr uint32_t kVhOob = 0xFFFFFFF0u;  // beyond every plane (< 2^30 samples): such a store is dropped

// The inputs of one V pass as they are LOADED: every lane fetches the rows of its own part only (KP averages, KP residuals).
// The 8 warm-up pairs in front of a part are the previous part's last own rows and come from lane - CW by ds_bpermute; only
// part 0 loads its warm-up rows itself (they lie above the stripe), and the last part the one average row below it. (r5, first
// form: every lane loaded all rows of its walk, so a wave requested each row piece twice within a few instructions.)
template <int CW>
struct VLoad {
    int32_t a[VHG<CW>::KP], r[VHG<CW>::KP];  // own rows: pairs 32 st + KP q + i
    int32_t wa[VH_WP], wr[VH_WP];            // part 0: pairs 32 st - 8 + i
    int32_t nx;                              // last part: average row 32 st + 32
    __device__ __forceinline__ void clear_edges() {
        nx = 0;
#pragma unroll
        for (int i = 0; i < VH_WP; i++) wa[i] = wr[i] = 0;
    }
};
// inside = every row and column the pass touches exists (st >= 1, 32 st + 33 <= ah, 32 st + 32 <= rh, cfirst + CW <= w): the
// row term rides in the SGPR offset. Otherwise the whole offset sits in the VGPR (an SGPR offset is not range-checked): rows
// above the plane wrap to huge offsets and read 0 like the rows below it.
template <int CW>
__device__ __forceinline__ void vh_v_load(const VHDesc& d, const VHRsrc& rs, int st, int cfirst, bool inside, VLoad<CW>& in) {
    using G = VHG<CW>;
    const int lane = threadIdx.x, q = lane / CW, j = lane % CW;
    const int w = d.w;
    // the row pitch as a value the compiler cannot see through: it would otherwise keep the row offsets i * pitch of a pass in
    // registers across the whole chunk loop (loop-invariant); formed by one add per load they need one
    int w4 = w * 4;
    asm volatile("" : "+s"(w4));
    if (inside) {
        const int voff = (G::KP * q * w + j) * 4;
        const int soff = (32 * st * w + cfirst) * 4;
        int so = soff;
#pragma unroll
        for (int i = 0; i < G::KP; i++, so += w4) {
            in.a[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.va, voff, so, 0);
            in.r[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.vb, voff, so, 0);
        }
        if (q == G::NQ - 1) in.nx = __builtin_amdgcn_raw_buffer_load_b32(rs.va, voff, so, 0);
        if (q == 0) {
            so = soff - VH_WP * w4;
#pragma unroll
            for (int i = 0; i < VH_WP; i++, so += w4) {
                in.wa[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.va, voff, so, 0);
                in.wr[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.vb, voff, so, 0);
            }
        }
    } else {
        int vo = ((32 * st + G::KP * q) * w + min(max(cfirst + j, 0), w - 1)) * 4;
        const int vo0 = vo - VH_WP * w4;
#pragma unroll
        for (int i = 0; i < G::KP; i++, vo += w4) {
            in.a[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.va, vo, 0, 0);
            in.r[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.vb, vo, 0, 0);
        }
        if (q == G::NQ - 1) in.nx = __builtin_amdgcn_raw_buffer_load_b32(rs.va, vo, 0, 0);
        if (q == 0) {
            vo = vo0;
#pragma unroll
            for (int i = 0; i < VH_WP; i++, vo += w4) {
                in.wa[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.va, vo, 0, 0);
                in.wr[i] = __builtin_amdgcn_raw_buffer_load_b32(rs.vb, vo, 0, 0);
            }
        }
    }
}
// One V pass: V-output columns cfirst + j (j = lane % CW, j < nc) of stripe st into LDS slot slot0 + j. The walk of part q covers
// pairs ybase .. ybase + 8 + KP - 1 (ybase = 32 st + KP q - 8): its first 8 pairs -- the warm-up -- take their inputs from lane -
// CW (the previous part's own rows) as they go, the others are the lane's own loads.
// EDGE = false: st >= 1, every row the pass touches exists (32 st + 33 <= ah, 32 st + 32 <= rh), all CW columns exist.
// Only the short form of tendency() lives in this kernel (modular_tend.h): samples outside its range guard are REPORTED like a
// mismatching boundary (the guard rides along the walk; what a violating walk stored is overwritten when the plan runs again
// with the one-step kernels, which carry the reference's long form). Both forms inlined per pass doubled the loop's code and
// spilled registers -- and a scratch reload makes the compiler wait for every outstanding load and store.
template <int CW, bool EDGE>
__device__ __forceinline__ void vh_v_pass(const VHDesc& d, const VLoad<CW>& ld, int st, int cfirst, int nc, int slot0, bool store_state,
                                          int32_t* lds, bool& bad) {
    using G = VHG<CW>;
    constexpr int KP = G::KP, NS = VH_WP + KP;
    const int lane = threadIdx.x, q = lane / CW, j = lane % CW;
    const int w = d.w;
    const int ybase = 32 * st + KP * q - VH_WP;
    const int col = min(max(cfirst + j, 0), w - 1);
    const bool colok = !EDGE || (j < nc && cfirst + j >= 0 && cfirst + j < w);
    const bool wr = !EDGE || j < nc;
    int32_t* p = lds + (2 * KP * q) * G::LD + slot0 + j;
    auto warm_a = [&](int i) { const int32_t t = __shfl_up(ld.a[KP - VH_WP + i], CW); return q == 0 ? ld.wa[i] : t; };
    auto warm_r = [&](int i) { const int32_t t = __shfl_up(ld.r[KP - VH_WP + i], CW); return q == 0 ? ld.wr[i] : t; };
    int32_t a = warm_a(0);
    int32_t top = a, side = 0;
    SqueezeRange rg;
    rg.init(a);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        int32_t nx, r;
        if (i < VH_WP - 1) nx = warm_a(i + 1);
        else if (i < NS - 1) nx = ld.a[i - (VH_WP - 1)];
        else {
            const int32_t t = __shfl_down(ld.a[0], CW);
            nx = q == G::NQ - 1 ? ld.nx : t;
        }
        r = i < VH_WP ? warm_r(i) : ld.r[i - VH_WP];
        rg.add(nx, r);
        int32_t f, s;
        if (!EDGE) {
            squeeze_pair_n(top, tend_n_pre(a, nx, r), f, s);
            top = s;
        } else {
            const int y = ybase + i;
            const int32_t nxe = y + 1 < d.ah ? nx : a;  // ModularChannel.java:398
            if (y == 0) top = a;                        // the column's first pair uses its own average (:399)
            squeeze_pair_n(top, tend_n_pre(a, nxe, r), f, s);
            if (y >= 0 && y < d.rh) top = s;
        }
        if (i == VH_WP - 1) side = top;
        if (i >= VH_WP && wr) {
            p[(2 * (i - VH_WP)) * G::LD] = f;
            p[(2 * (i - VH_WP) + 1) * G::LD] = s;
        }
        a = nx;
    }
    if (!rg.ok(0)) bad = true;
    if (EDGE && d.ah > d.rh) {  // odd height: the last row is the last average row (ModularChannel.java:409-411)
        const int yl = 2 * d.rh - VH_ROWS * st;
        if (yl >= 0 && yl < VH_ROWS && q == 0 && colok) lds[yl * G::LD + slot0 + j] = d.va[(int64_t)d.rh * w + col];
    }
    // the previous part's last output is what this part's warm-up should have arrived at
    const int32_t prev_top = __shfl_up(top, CW);
    const bool qlive = !EDGE || (colok && 32 * st + KP * q < d.rh);
    if (q >= 1 && qlive && side != prev_top) bad = true;
    if (store_state && colok) {
        const int nsv = (d.rh + 31) >> 5;  // stripes that hold V pairs
        if (q == 0 && st >= 1 && st < nsv) d.side_v[(int64_t)st * w + col] = side;
        if (q == G::NQ - 1 && st + 1 < nsv) d.tail_v[(int64_t)st * w + col] = top;
    }
}

// H residuals of pairs c0 .. c0 + CW - 1, rows of the stripe -> registers (coalesced 16-byte loads: lane = (row, piece of the row
// piece)) -> LDS slots R0 .. R0 + CW - 1. Rows below the plane read as 0; columns beyond the row's end read the next row's
// samples, which no pair that exists ever uses.
template <int CW>
struct HIn {
    i32x4 v[VHG<CW>::HN];
};
template <int CW>
__device__ __forceinline__ void vh_h_load(const VHDesc& d, const VHRsrc& rs, int st, int c0, bool inside, HIn<CW>& in) {
    using G = VHG<CW>;
    const int lane = threadIdx.x, pc = lane % G::HL;
    const int voff = ((lane / G::HL) * d.rw + 4 * pc) * 4;
    int rstep = (64 / G::HL) * d.rw * 4;
    asm volatile("" : "+s"(rstep));  // (as in vh_v_load: one add per load instead of the offsets held across the chunk loop)
    if (inside) {
        int so = (VH_ROWS * st * d.rw + c0) * 4;
#pragma unroll
        for (int ps = 0; ps < G::HN; ps++, so += rstep) in.v[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs.hb, voff, so, 0);
    } else {  // rows below the plane: the whole offset in the VGPR, where the range check sees it
        int vo = voff + (VH_ROWS * st * d.rw + c0) * 4;
#pragma unroll
        for (int ps = 0; ps < G::HN; ps++, vo += rstep) in.v[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs.hb, vo, 0, 0);
    }
}
template <int CW>
__device__ __forceinline__ void vh_h_to_lds(const HIn<CW>& in, int32_t* lds) {
    using G = VHG<CW>;
    const int lane = threadIdx.x, pc = lane % G::HL;
#pragma unroll
    for (int ps = 0; ps < G::HN; ps++) {
        const int rr = (64 / G::HL) * ps + lane / G::HL;
#pragma unroll
        for (int e = 0; e < 4; e++) lds[rr * G::LD + G::R0 + 4 * pc + e] = in.v[ps][e];
    }
}

// H pass of one chunk: lane = row. a0: the average of the chunk's first pair (column c0). npairs: pairs of this chunk that exist
// (EDGE only); extra: the odd last output column (a copy of V-output column rw) falls into this chunk at pair index npairs.
// The row's samples are read from LDS as the walk reaches them; pair i leaves its two outputs in the two slots it has just
// consumed (its average A[i] -- slot 0 is spare -- and its residual R[i]): nothing of the row has to sit in registers.
template <int CW, bool EDGE>
__device__ __forceinline__ void vh_h_pass(const VHDesc& d, int c0, bool warm, int npairs, bool extra, int32_t* lds,
                                          int32_t a0, int32_t& left, int32_t& carry, bool& bad) {
    using G = VHG<CW>;
    int32_t* row = lds + threadIdx.x * G::LD;
    int32_t a = warm ? row[1] : a0;
    SqueezeRange rg;
    rg.init(a);
    if (!rg.ok(left)) bad = true;
#pragma unroll
    for (int i = 0; i < CW; i++) {
        const int32_t nx = row[i + 1], r = row[G::R0 + i];
        rg.add(nx, r);
        if (i == 1 && warm) left = a;  // the warm-up starts at pair c0 + 1 from the same guess
        int32_t f, s;
        if (!EDGE) {
            squeeze_pair_n(left, tend_n_pre(a, nx, r), f, s);
            left = s;
        } else {
            const int32_t nxe = c0 + i + 1 < d.w ? nx : a;  // ModularChannel.java:368
            squeeze_pair_n(left, tend_n_pre(a, nxe, r), f, s);
            if (i < npairs) left = s;
            if (extra && i == npairs) f = a;  // :383-385
        }
        if (!warm) {
            row[i] = f;
            row[G::R0 + i] = s;
        }
        a = nx;
    }
    carry = a;
    if (!rg.ok(0)) bad = true;
}

// The chunk's 64 x 2 CW outputs: LDS rows -> global. Lane = (row, pair): the pair's two outputs are its slots A[i] and R[i] (one
// ds_read2) and leave as ONE 8-byte store, a wave-instruction covering 64 / CW whole row pieces of 8 CW bytes.
// Why not 16 bytes per lane: gfx950 store-data hazard. A 16-byte store still reads its data registers for two issue cycles after
// it has issued, and a VALU write to one of them in that window reaches memory in some lanes (observed: the last four lanes of
// every row of 16, one of the four dwords, differently from run to run -- the register allocator recycles v[2..3] of one store as
// v[0..1] of the next). hipcc pads the hazard with s_nop itself, but not for buffer stores with a register in the soffset field
// (its rule for older parts), and it counts an s_waitcnt as a wait state, which an already satisfied one is not: the first r5
// builds got a few dozen samples per image wrong. Stores of at most 8 bytes have no such window; the 16-byte form measured no
// faster (the stores of this access shape run at 6.8 TB/s either way, tools/ubench/vh_pattern.hip). tools/scan_store_hazard.py
// (tests/test_isa_hazards.py) checks the ISA of every kernel of the library for the pattern.
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int CW, bool EDGE>
__device__ __forceinline__ void vh_store(const VHDesc& d, const VHRsrc& rs, int st, int c0, int ncols, const int32_t* lds) {
    using G = VHG<CW>;
    constexpr int RPI = 64 / CW;  // rows per wave-instruction
    const int lane = threadIdx.x, pi = lane % CW;
    const int ow = d.w + d.rw;
    const int voff = ((lane / CW) * ow + 2 * pi) * 4;
    int rstep = RPI * ow * 4;
    asm volatile("" : "+s"(rstep));
    int so = (VH_ROWS * st * ow + 2 * c0) * 4;
    const int32_t* src = lds + (lane / CW) * G::LD + pi;
#pragma unroll
    for (int ps = 0; ps < CW; ps++, so += rstep) {
        const int32_t* sp = src + RPI * ps * G::LD;
        i32x2 v;
        v[0] = sp[0];
        v[1] = sp[G::R0];
        if (!EDGE) {
            __builtin_amdgcn_raw_buffer_store_b64(v, rs.o, voff, so, 0);
        } else {  // rows below the plane are dropped by the range check (whole offset in the VGPR); columns beyond the chunk's last one must not be written
#pragma unroll
            for (int e = 0; e < 2; e++)
                __builtin_amdgcn_raw_buffer_store_b32(v[e], rs.o, 2 * pi + e < ncols ? voff + so + 4 * e : (int)kVhOob, 0, 0);
        }
    }
}

}  // namespace

// the segment-boundary arrays of an earlier step: every workgroup compares a slice (the r4 scheme of k_modular.hip)
__device__ __forceinline__ void squeeze_check_slice(const SqueezeCheck* chk, int n_chk, int32_t* flag) {
    if (n_chk <= 0) return;
    const int64_t wg = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
    const int64_t stride = (int64_t)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
    bool mism = false;
    for (int q = 0; q < n_chk; q++) {
        const SqueezeCheck ck = chk[q];
        const int64_t total = (int64_t)(ck.nseg - 1) * ck.n;
        for (int64_t i = wg * blockDim.x + threadIdx.x; i < total; i += stride) mism = mism || ck.side[ck.n + i] != ck.tail[i];
    }
    if (__builtin_expect(__any(mism), 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <int CW>
__device__ __forceinline__ void vh_body(const VHBatch& bt, int32_t* lds) {
    using G = VHG<CW>;
    squeeze_check_slice(bt.chk, bt.n_chk, bt.flag);
    // consecutive workgroup ids go to different XCDs: give every XCD a contiguous run of tiles (neighbouring tiles share their
    // warm-up rows / columns, and an XCD has its own L2)
    const int per = (bt.n_tiles + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= bt.n_tiles) return;
    int ci = 0;
    for (int i = 1; i < bt.n; i++)
        if (t >= bt.d[i].tile0) ci = i;
    const VHDesc d = bt.d[ci];
    const int local = t - d.tile0;
    const int st = local / d.nseg, sg = local - st * d.nseg;
    const int htot = d.ah + d.rh;
    // The chunks of a row start at pairs 1, 1 + CW, 1 + 2 CW, ... (pair 0 is done on its own in front of them): with the planes'
    // leading pads (kVhPad / kVhPadH, host.hip) the V-input pieces (from column c0 + 1), the H-residual pieces (from pair c0) and the
    // output pieces (from column 2 c0) then all start on 128-byte lines. r5, first form: chunks from pair 0 -- the V pieces started
    // one sample late, every line of V input was requested by two consecutive chunks ~10 us apart and fetched twice
    // (TCC_EA0_RDREQ x 128 B = 1.77 x the input bytes).
    const int xb = sg == 0 ? 0 : 1 + sg * d.seg, xe = min(d.rw, 1 + (sg + 1) * d.seg);
    const int xs = sg > 0 ? xb - CW : 1;                // first chunk: the warm-up chunk, or the one at pair 1
    const bool extra = sg == d.nseg - 1 && d.w > d.rw;  // the odd last output column belongs to the last segment
    const int nck = max(0, (xe + (extra ? 1 : 0) - xs + CW - 1) / CW);
    const bool stripe_inside = st >= 1 && 32 * st + 33 <= d.ah && 32 * st + 32 <= d.rh && VH_ROWS * st + VH_ROWS <= htot;
    VHRsrc rs;
    rs.va = vh_rsrc(d.va, VH_ABL(1) ? 0 : (int64_t)d.ah * d.w);
    rs.vb = vh_rsrc(d.vb, VH_ABL(1) ? 0 : (int64_t)d.rh * d.w);
    rs.hb = vh_rsrc(d.hb, VH_ABL(1) ? 0 : (int64_t)htot * d.rw);
    rs.o = vh_rsrc(d.o, VH_ABL(2) ? 0 : (int64_t)htot * (d.w + d.rw));
    int32_t left = 0, carry = 0;
    bool bad = false;
    const int grow = VH_ROWS * st + threadIdx.x;
    // software pipeline: the loads of chunk k + 1 are issued between the V pass and the H pass of chunk k and land while the
    // H pass and the output stores run
    auto inside = [&](int c0) { return stripe_inside && c0 + CW < d.w && c0 + CW <= xe; };
    VLoad<CW> vld;
    HIn<CW> hin;
    vld.clear_edges();
    vh_v_load<CW>(d, rs, st, xs + 1, inside(xs), vld);
    vh_h_load<CW>(d, rs, st, xs, inside(xs), hin);
    if (sg == 0) {
        // pair 0 of every row, in front of the chunk grid: V-output columns 0 and 1 (slots 0 and 1), then the pair itself -- its
        // left neighbour is its own average (ModularChannel.java:370) -- and column 1 is what the chunk at pair 1 starts from
        VLoad<CW> l0;
        l0.clear_edges();
        vh_v_load<CW>(d, rs, st, 0, false, l0);
        vh_v_pass<CW, true>(d, l0, st, 0, 2, 0, true, lds, bad);
        lds_fence();
        if (grow < htot) {
            const int32_t* row = lds + threadIdx.x * G::LD;
            const int32_t a = row[0], nx = 1 < d.w ? row[1] : a;
            const int32_t r = d.hb[(int64_t)grow * d.rw];
            SqueezeRange rg;
            rg.init(a);
            rg.add(nx, r);
            if (!rg.ok(a)) bad = true;
            int32_t f, s2;
            squeeze_pair_n(a, tend_n_pre(a, nx, r), f, s2);
            int32_t* o = d.o + (int64_t)grow * (d.w + d.rw);
            o[0] = f;
            o[1] = s2;
            left = s2;
            carry = row[1];
        }
        lds_fence();
    }
    // Inside a stripe whose rows all exist the chunks of a segment are "inside" (every column and pair exists) up to the last
    // one or two: two loops in sequence, so that the edge forms stay out of the hot loop's code and registers.
    auto run = [&](auto edge_tag, int k0, int k1) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        for (int k = k0; k < k1; k++) {
            const int c0 = xs + CW * k;
            const bool warm = sg > 0 && k == 0;
            vh_h_to_lds<CW>(hin, lds);
            __builtin_amdgcn_sched_barrier(0);  // (phase boundaries: the scheduler must not stretch the prefetch registers' lives)
            vh_v_pass<CW, EDGE>(d, vld, st, c0 + 1, CW, 1, !warm, lds, bad);
            lds_fence();
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nck) {  // (issued before the V pass -- a whole chunk ahead -- they cost 20 more registers and bought nothing)
                vh_v_load<CW>(d, rs, st, c0 + CW + 1, inside(c0 + CW), vld);
                vh_h_load<CW>(d, rs, st, c0 + CW, inside(c0 + CW), hin);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int npairs = min(CW, xe - c0);
            const bool xcol = extra && npairs < CW;
            vh_h_pass<CW, EDGE>(d, c0, warm, npairs, xcol, lds, carry, left, carry, bad);
            lds_fence();
            if (!warm) vh_store<CW, EDGE>(d, rs, st, c0, 2 * npairs + (xcol ? 1 : 0), lds);
            lds_fence();
            if (warm && grow < htot) d.side_h[(int64_t)sg * htot + grow] = left;  // the state this wave starts its own segment from
        }
    };
    int n_in = 0;
    if (stripe_inside) {
        const int lim = min(d.w - 1, xe) - xs;  // inside(c0) <=> c0 + CW <= min(w - 1, xe)
        n_in = lim >= 0 ? min(nck, lim / CW) : 0;
    }
    run(std::false_type{}, 0, n_in);
    run(std::true_type{}, n_in, nck);
    if (sg + 1 < d.nseg && grow < htot) d.tail_h[(int64_t)sg * htot + grow] = left;  // last output of this segment
    if (__builtin_expect(__any(bad), 0) && threadIdx.x == 0) atomicOr(bt.flag, 1);
}

__global__ __launch_bounds__(64, 3) void k_inv_vh(const VHBatch bt) {
    __shared__ int32_t lds[VH_ROWS * VHG<16>::LD];
    vh_body<16>(bt, lds);
}
// the wide form: 9 waves per CU by LDS, so the register budget of two waves per SIMD is free
__global__ __launch_bounds__(64, 2) void k_inv_vh32(const VHBatch bt) {
    __shared__ int32_t lds[VH_ROWS * VHG<32>::LD];
    vh_body<32>(bt, lds);
}

__device__ __forceinline__ bool squeeze_check_one(const SqueezeCheck ck) {
    const int64_t total = (int64_t)(ck.nseg - 1) * ck.n;
    bool mism = false;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) mism = mism || ck.side[ck.n + i] != ck.tail[i];
    return mism;
}

// the arrays of up to four channels per launch, by value (no table in memory, no indexed copy in scratch)
__global__ __launch_bounds__(256) void k_squeeze_check(int n_chk, int32_t* flag, SqueezeCheck c0, SqueezeCheck c1, SqueezeCheck c2, SqueezeCheck c3) {
    bool mism = squeeze_check_one(c0);
    if (n_chk > 1) mism = mism || squeeze_check_one(c1);
    if (n_chk > 2) mism = mism || squeeze_check_one(c2);
    if (n_chk > 3) mism = mism || squeeze_check_one(c3);
    if (__builtin_expect(__any(mism), 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

void launch_squeeze_vh(const VHBatch& bt, hipStream_t s) {
    if (bt.n <= 0 || bt.n_tiles <= 0) return;
    const int per = (bt.n_tiles + 7) / 8;
#ifdef JXL_VH_ABL
    const int abl = getenv("JXL_VH_ABL") ? atoi(getenv("JXL_VH_ABL")) : 0;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_vh_abl), &abl, sizeof(int), 0, hipMemcpyHostToDevice, s);
#endif
    if (bt.cw == 32) hipLaunchKernelGGL(k_inv_vh32, dim3(8 * per), dim3(64), 0, s, bt);
    else hipLaunchKernelGGL(k_inv_vh, dim3(8 * per), dim3(64), 0, s, bt);
}

void launch_squeeze_check(const SqueezeCheck* chk, int n_chk, int32_t* flag, hipStream_t s) {
    for (int i0 = 0; i0 < n_chk; i0 += 4) {
        SqueezeCheck c[4] = {};
        const int n = n_chk - i0 < 4 ? n_chk - i0 : 4;
        int64_t total = 0;
        for (int i = 0; i < n; i++) {
            c[i] = chk[i0 + i];
            total += (int64_t)(c[i].nseg - 1) * c[i].n;
        }
        int grid = (int)((total / n + 255) / 256);  // ~ one element of every array per thread
        grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
        hipLaunchKernelGGL(k_squeeze_check, dim3(grid), dim3(256), 0, s, n, flag, c[0], c[1], c[2], c[3]);
    }
}

}  // namespace jxl
