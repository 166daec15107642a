// Host side of libjxlatte_amd.so: the C-ABI of include/jxlatte_amd.h, the per-frame device arena,
// varblock binning and the launch sequence. gfx950 only; there is no CPU fallback -- every entry
// point needs a HIP device and fails with JXL_ERR_DEVICE otherwise.
#include "jxl_internal.h"
#include "modular_tend.h"  // kSqueezeSafeIn (jxl_modular_begin picks the plan by it)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include <mutex>
#include <limits>

using namespace jxl;
namespace jxl {
thread_local hipEvent_t g_restore_kernel_ev[2] = {nullptr, nullptr};  // (stage timing: restore_fused_body.h, launch_tph)
}

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool own = true;  // false: a view into another allocation (the per-frame table arena)
    bool ensure(size_t bytes) {
        if (bytes <= cap && p && own) return true;
        if (p && own) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        own = true;
        if (bytes == 0) bytes = 16;
        if (hipMalloc(&p, bytes) != hipSuccess) return false;
        cap = bytes;
        return true;
    }
    void release() {
        if (p && own) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        own = true;
    }
    void view(void* q, size_t bytes) {
        if (p && own) (void)hipFree(p);
        p = q;
        cap = bytes;
        own = false;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct ModOp {
    int kind;  // 0 hsqueeze, 1 vsqueeze, 2 rct, 3 copy, 4 batched squeeze step (bt), 5 chain of small steps (chain_*),
               // 6 a V step and the H step behind it in one launch (vh; r5, only in jxl_ctx::mod_ops_fused)
    VHBatch vh;
    const SqueezeBatch* chain_dev = nullptr;
    int chain_steps = 0, chain_slots = 0;
    SqueezeBatch bt;
    const int32_t* a;
    const int32_t* b;
    int32_t* o;
    int32_t *v0, *v1, *v2;
    int adim, rdim, other;  // hsq: aw, rw, h; vsq: ah, rh, w
    int64_t n;
    int type;
};

struct ModChan {
    int w, h;
    int32_t* d;
    bool original;
};

}  // namespace

// a typed window of the table staging buffer (jxl_ctx::h_tab)
template <class T>
struct HSpan {
    T* p = nullptr;
    size_t n = 0;
    T& operator[](size_t i) const { return p[i]; }
    T* data() const { return p; }
    T* begin() const { return p; }
    T* end() const { return p + n; }
    size_t size() const { return n; }
};

struct jxl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    DevBuf lut;

    // ---- VarDCT frame state
    bool frame_open = false;
    bool tables_dirty = true;
    bool have_weights = false;
    jxl_vardct_params p{};
    int W = 0, H = 0, bw = 0, bh = 0, tw = 0, th = 0;
    DevBuf coeff[3], lf[3], llf[3], weights_t, hf_mul, sharp, xfy, bfy, weights, planeA[3], planeB[3], outbuf[3], inv_sigma, blocks, items,
        bad_flag;
    int32_t woffs[51]{};
    // Per-frame tables (r4): ONE page-locked staging buffer laid out like ONE device arena, ONE transfer per prepare. The
    // fixed-size grids (hfMultiplier, sharpness, CfL factors per tile, LF planes: sizes known at begin_frame) come first and the
    // host writes them in place (set_lfgroup); the block records and item lists follow (finalize_tables). The transfer is queued
    // and an event is recorded; the host waits for it only before it writes the staging buffer again. Until r3: nine transfers
    // (two of them synchronous) from nine buffers and a stream synchronisation, 0.41 ms of a 1.2 ms prepare.
    char* h_tab = nullptr;
    size_t h_tab_cap = 0;
    bool h_tab_pinned = false;
    DevBuf tab;
    hipEvent_t tab_ev = nullptr;
    bool tab_inflight = false;
    // r4: device work begin_frame used to queue for every frame, now done only where something depends on it (six memsets of
    // 33 MB and three plane copies per 4K frame: 0.05 ms of device time, but nine runtime calls -- with a dozen contexts driving
    // frames from a dozen threads the calls, not the bytes, were what the streaming rate paid for)
    bool coeff_zero_pending = false;   // the coefficient planes have not been zeroed for this frame yet (put_group / run do it; a
                                       // commit of the mapped planes overwrites every sample and needs none)
    bool out_zero_pending = false;     // likewise the transform output planes: only needed when the varblocks do not tile the frame
    bool blocks_cover = false;         // (finalize_tables) every 8x8 cell belongs to exactly one varblock
    std::vector<uint8_t> cell_mark;    // its proof: one mark per cell
    bool llf_alias = false;            // no kernel writes the llf planes for this frame: they ARE the lf planes (no copy)
    size_t tab_fixed = 0, off_hfm = 0, off_sharp = 0, off_kx = 0, off_kb = 0, off_lf[3] = {0, 0, 0};
    HSpan<int32_t> h_hf_mul, h_sharp;
    std::vector<int32_t> h_xfy, h_bfy;
    HSpan<float> h_kx, h_kb;
    std::vector<uint8_t> h_sel;
    HSpan<float> h_lf[3];
    int32_t sharp_bad = 0;       // first EPF sharpness outside 0..7 seen by finalize_tables (Frame.java:565-566), or 0
    bool sharp_is_bad = false;
    std::vector<std::vector<DevBlock>> lfg_blocks;  // per LF group, reference order, frame coordinates
    std::vector<uint8_t> lfg_set;
    struct LfJob { jxl_lfquant_desc d; std::vector<int32_t> q[3]; };
    std::vector<LfJob> lf_jobs;  // integer LF images to dequantise + smooth on the device (row f1)
    DevBuf lfq_tmp[3];
    DevBuf pq_tab;  // PQ segment table (jxl_fastpow.h), uploaded at context creation
    DevBuf srgb8_tab;  // sRGB -> 8-bit threshold table (fp_srgb8), uploaded at context creation
    DevBuf srgb16_tab; // sRGB -> 16 bit: segments + thresholds (fp_srgb16), built and uploaded on first use
    DevBuf pq16_thr;   // PQ -> 16-bit thresholds (fp_pq16), built and uploaded on the first PQ frame / stage with 16-bit output
    // resident colour planes between decodeFrame and the colour transform (jxl_planes_*): dense [rp_h][rp_w] floats
    DevBuf rp[3], rp_tmp[3], rp_noise[3];
    int rp_h = 0, rp_w = 0;
    // binned work
    // one merged launch: the segments (types) of one register class; channel >= 0: chroma-subsampled frame, one channel per launch
    struct TypeLaunch { int cls, channel; std::vector<IdctSegment> segs; };
    bool sub = false;      // any jpeg_upsampling shift non-zero
    int sy[3] = {0, 0, 0}, sx[3] = {0, 0, 0};
    DevBuf hfm_sub[3];     // hfMultiplier resampled onto each channel's cell grid
    struct SpecialLaunch { int items_off, n_items, channel; bool wg_items; };
    std::vector<SpecialLaunch> special_launches;
    std::vector<TypeLaunch> type_launches;
    int large_first = 0, large_count = 0;
    int llf_first = 0, llf_count = 0;  // blocks larger than 8x8 (contiguous in h_blocks)
    std::vector<DevBlock> h_blocks;
    // results
    void* result[3] = {nullptr, nullptr, nullptr};
    int result_elem = 4;
    bool result_interleaved = false;
    int last_launches = 0;
    uint64_t tables_gen = 0;  // bumped whenever finalize_tables rebuilds the binned work (batch argument cache key)
    // jxl_vardct_run_batch state (kept by the first context of a batch)
    DevBuf batch_args, batch_restore_args;
    std::vector<FusedArgs> batch_restore_host;
    bool batch_restore_valid = false;
    std::vector<std::pair<const jxl_ctx*, uint64_t>> batch_key;
    // cls 0 / 1 = k_idct_multi classes, 3 = the special 8x8 kernel (MultiArgs blocks in batch_args);
    // 10 = k_llf_wg3, 11 / 12 = k_idct_wg3<false / true> (Wg3Args blocks in batch_wg3_args; grid_x of 10 = lanes)
    struct BatchLaunch { int cls, n_frames, grid_x; size_t lds_bytes, offset; };
    DevBuf batch_wg3_args;
    void* h_map16 = nullptr;   // page-locked frame-sized int16 planes handed to the caller (jxl_vardct_map_coeffs_i16)
    bool map16_nofill = false;       // mapped without zero-fill: commit must be told which groups were written
    bool is_feeder = false;          // counted in g_feeders (bus_grid): this context has mapped coefficient planes
    hipEvent_t map16_ev = nullptr;   // "the commit's transfers have read h_map16": what the next map waits for (not the whole stream)
    bool map16_inflight = false;
    hipEvent_t out_ev = nullptr;     // jxl_vardct_read_output_begin's copies
    bool out_inflight = false;
    size_t h_map16_bytes = 0;
    bool map16_valid = false;
    DevBuf stage16;            // the committed int16 planes on the device (the staged form of jxl_vardct_commit_coeffs_i16)
    // page-locked staging ring of jxl_vardct_put_group*: one slot = the three rectangles of a group
    static constexpr int kGrpSlots = 8;
    void* h_grp = nullptr;
    const void* h_grp_dev = nullptr;
    hipEvent_t grp_ev[kGrpSlots] = {};
    bool grp_inflight[kGrpSlots] = {};
    int grp_slot = 0;
    std::vector<float> h_weights_in;  // the last weight set handed over (set_weights skips an identical one)
    int32_t h_woffs_in[51] = {};
    DevBuf wg3_items[2];       // spatially ordered item lists of the two k_idct_wg3 classes (wg3_item_table)
    int wg3_item_count[2] = {0, 0};
    std::vector<BatchLaunch> batch_launches;
    hipEvent_t batch_ev = nullptr;
    bool timing = false;
    static constexpr int kEvSlots = 32;
    hipEvent_t ev[kEvSlots][3] = {};  // ring of (start, after IDCT stage, end) per run
    hipEvent_t kev[kEvSlots][2] = {}; // ... and the restoration kernel's own start / stop (r6)
    bool kev_valid = false;           // the last timed run took the single-launch fused restoration kernel
    int ev_runs = 0;                  // runs recorded since timing was enabled
    bool owns_stream = true;
    // fork/join side streams: the per-type IDCT kernels are independent and individually too small to fill
    // 256 CUs, so they run concurrently
    static constexpr int kAux = 12;
    // side streams in use (JXL_AUX_STREAMS overrides). HIP maps streams onto 4 hardware queues round-robin in creation order,
    // process-wide, and kernels of streams that share a queue run one after the other: with main + ONE side stream per context the
    // main streams of a batch of contexts alternate between two queues and the side streams between the other two. Measured
    // (r3, 8 contexts x 4K frames, same box): streams per context 2 -> 48.9 Gpx/s, 3 -> 40.8, 4 -> 41.4 (every main stream on ONE
    // queue), 5 -> 45.3, 6 -> 46.7, 7 -> 40.0, 8 -> 40.7. So: no stream is created that is not used.
    int n_aux = 1;
    hipStream_t aux[kAux] = {};
    hipEvent_t fork_ev = nullptr, llf_ev = nullptr, join_ev[kAux] = {};

    // ---- Modular state
    std::vector<DevBuf> mod_bufs;
    std::vector<ModOp> mod_ops;        // the plan, one launch per squeeze step (what a reported mismatch falls back to)
    std::vector<ModOp> mod_ops_fused;  // the same plan with every (V, H) pair of steps as one launch (r5): what jxl_modular_run runs
    std::vector<ModChan> mod_out;
    // speculative verification of the segmented squeeze walks (jxl_modular_run): report flag (device), its page-locked host
    // copy, the events that order the check stream, and whether a run's flag has not been looked at yet
    DevBuf mod_flag;
    int32_t* mod_flag_host = nullptr;
    hipEvent_t mod_ev = nullptr, mod_join = nullptr;
    bool mod_pending = false;
    int mod_redos = 0;
    int mod_launches = 0;
};

namespace {

thread_local std::string g_err;

jxl_status fail(jxl_ctx* ctx, jxl_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return st;
}

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(ctx, e_ == hipErrorOutOfMemory ? JXL_ERR_OOM : JXL_ERR_DEVICE, "%s: %s", #expr,     \
                        hipGetErrorString(e_));                                                             \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// cosineLut of MathHelper.java:17-30, all sizes back to back (size s = 1<<l at lut_off(l))
void build_lut(std::vector<float>& t) {
    t.assign(kLutTotal, 0.0f);
    const double root2 = std::sqrt(2.0);
    const double pi = 3.14159265358979323846;  // Math.PI
    for (int l = 0; l < 9; l++) {
        const int s = 1 << l;
        float* o = t.data() + lut_off(l);
        for (int n = 0; n < s - 1; n++)
            for (int k = 0; k < s; k++) o[n * s + k] = (float)(root2 * std::cos(pi * (n + 1) * (k + 0.5) / s));
    }
}

jxl_status bind(jxl_ctx* ctx) {
    if (!ctx) return fail(nullptr, JXL_ERR_INVALID_ARGUMENT, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return JXL_OK;
}

// The writers of the coefficient planes. The C ABI hands over raster planes; the device planes are tiled by 8x8 cell
// (coeff_off, jxl_internal.h), so every writer places sample (y0 + y, x0 + x) of a plane W wide through coeff_off.
// One (pass, group) of all three channels in one launch, read straight from page-locked HOST memory (the caller's buffers when
// they are page-locked and 16-byte aligned, else the context's staging ring): a lane moves 8 samples of one row -- 16 (int16) or
// 32 (int32) bytes over PCIe, one 32-byte cell row into the tiled plane. acc: the passes after the first add
// (PassGroup.java:174-200, Java int wrap). Replaces a 2-D copy + a kernel (+ a host wait) per CHANNEL (r1-r3).
struct PutGroupArgs {
    int32_t* plane[3];
    const void* src[3];
    int32_t sstride[3];  // elements
    int32_t W[3], y0[3], x0[3], gw[3], gh[3];
    int32_t acc;
};
template <typename T>
__global__ __launch_bounds__(256) void k_put_group(const PutGroupArgs a) {
    const int ch = blockIdx.z;
    const int r = threadIdx.x >> 5, bx = blockIdx.x * 32 + (threadIdx.x & 31), by = blockIdx.y;
    if (bx * 8 >= a.gw[ch] || by * 8 >= a.gh[ch]) return;
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const T* sp = static_cast<const T*>(a.src[ch]) + (int64_t)(by * 8 + r) * a.sstride[ch] + bx * 8;
    v4i_ lo, hi;
    if constexpr (sizeof(T) == 2) {
        const v4i_ pk = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sp));
        lo = v4i_{(pk.x << 16) >> 16, pk.x >> 16, (pk.y << 16) >> 16, pk.y >> 16};
        hi = v4i_{(pk.z << 16) >> 16, pk.z >> 16, (pk.w << 16) >> 16, pk.w >> 16};
    } else {
        lo = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sp));
        hi = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sp) + 1);
    }
    int32_t* d = a.plane[ch] + coeff_off(a.W[ch], a.y0[ch] + by * 8 + r, a.x0[ch] + bx * 8);
    if (a.acc) {
        const v4i_ l0 = *reinterpret_cast<v4i_*>(d), h0 = *reinterpret_cast<v4i_*>(d + 4);
        typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
        lo = (v4i_)((v4u_)lo + (v4u_)l0);
        hi = (v4i_)((v4u_)hi + (v4u_)h0);
    }
    *reinterpret_cast<v4i_*>(d) = lo;
    *reinterpret_cast<v4i_*>(d + 4) = hi;
}

// int16 wire format -> the int32 coefficient planes
__global__ void k_widen2d(int32_t* plane, int W, int y0, int x0, const int16_t* src, int gw, int gh, int acc) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= gw || y >= gh) return;
    int32_t* d = plane + coeff_off(W, y0 + y, x0 + x);
    const int32_t v = (int32_t)src[(int64_t)y * gw + x];
    *d = acc ? (int32_t)((uint32_t)*d + (uint32_t)v) : v;
}

// The same for a whole plane from page-locked HOST memory (r4: jxl_vardct_commit_coeffs_i16 without the staging copy): every lane
// moves 8 samples -- one 16-byte read over PCIe, two 16-byte stores (one cell row) -- so that the transfer is made of full-size
// read requests and needs no SDMA transfer, no device staging buffer and no host API call per plane besides this launch.
// gw % 8 == 0, gh % 8 == 0. A workgroup takes 32 cells of a cell row: 32 consecutive lanes read 512 consecutive bytes of one
// sample row (the PCIe side keeps its long runs), the 8 rows of the workgroup complete every cell it touches.
// r5: a bounded grid walks the tiles (JXL_WIDEN_GRID workgroups, default 128): the transfer runs at the bus's rate with a few
// hundred requests in flight, and a launch that parks a wave on every slot of the chip while it waits for the bus starves the
// other contexts' IDCT / restoration kernels of their slots (streaming: those ran 3-5 x longer beside a widening launch).
__global__ __launch_bounds__(256) void k_widen2d_host8(int32_t* __restrict__ plane, const int16_t* __restrict__ src, int gw, int gh, int gx, int tiles) {
    // a tile = one cell row x 64 cells: a wave reads 1 KB of consecutive samples of a row (two rows per wave, both loads in flight)
    const int wv = threadIdx.x >> 6, lx = threadIdx.x & 63;
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int by = t / gx, bx = (t - by * gx) * 64 + lx;
        if (bx * 8 >= gw) continue;
        const int16_t* sp = src + (int64_t)(by * 8 + wv * 2) * gw + bx * 8;
        const v4i_ p0 = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sp));
        const v4i_ p1 = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sp + gw));
        // 8-byte stores (kept apart by compiler barriers, or the back end merges them again): the kernel waits on the bus, not on its stores, and a 16-byte store followed at once by the next unpack
        // into its data registers is the gfx950 store-data hazard of DESIGN 4.3 (tools/scan_store_hazard.py flags the 16-byte form)
        typedef int v2i_ __attribute__((ext_vector_type(2)));
        v2i_* d = reinterpret_cast<v2i_*>(plane + (((int64_t)by * (gw >> 3) + bx) << 6) + wv * 16);
        d[0] = v2i_{(p0.x << 16) >> 16, p0.x >> 16}; asm volatile("" ::: "memory");
        d[1] = v2i_{(p0.y << 16) >> 16, p0.y >> 16}; asm volatile("" ::: "memory");
        d[2] = v2i_{(p0.z << 16) >> 16, p0.z >> 16}; asm volatile("" ::: "memory");
        d[3] = v2i_{(p0.w << 16) >> 16, p0.w >> 16}; asm volatile("" ::: "memory");
        d[4] = v2i_{(p1.x << 16) >> 16, p1.x >> 16}; asm volatile("" ::: "memory");
        d[5] = v2i_{(p1.y << 16) >> 16, p1.y >> 16}; asm volatile("" ::: "memory");
        d[6] = v2i_{(p1.z << 16) >> 16, p1.z >> 16}; asm volatile("" ::: "memory");
        d[7] = v2i_{(p1.w << 16) >> 16, p1.w >> 16};
    }
}

// r5: transfers between page-locked host memory and the device as KERNELS that move 16 bytes per lane through the host buffer's
// device alias, on a bounded grid. With several contexts streaming frames from several host threads, every hipMemcpyAsync of a frame
// (the side tables in, the pixels out) held its calling thread for 1.6-2.6 ms -- the runtime's copy path waits on the host for the
// stream's earlier work -- while a launch returns in microseconds (tools/archive/r5_stream_sections.sh, profiles/r5_stream_*.txt).
typedef int v4i_cp __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const v4i_cp* __restrict__ src, v4i_cp* __restrict__ dst, size_t n16, size_t tail_bytes) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += step) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    if (blockIdx.x == 0 && threadIdx.x < tail_bytes)
        reinterpret_cast<uint8_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t*>(src + n16)[threadIdx.x];
}

// queue dst[0, bytes) = src[0, bytes) on the context's stream as a launch of k_copy16; the host side (dst if dst_is_host, else src)
// must be page-locked and both sides 16-byte aligned. Returns 1: queued; 0: not possible (pageable memory, unaligned, or the extent
// of a caller's buffer cannot be established) -- the caller takes the runtime's copy, which checks the range itself; -1: the
// page-locked allocation is SHORTER than the transfer (r6: a kernel writing through the device alias past the end of a registration is
// a GPU page fault that kills the process, where hipMemcpyAsync returned an error -- the library's own staging buffers skip the
// look-up, `caller_buffer` = false).
int copy_zero(jxl_ctx* c, void* dst, const void* src, size_t bytes, bool dst_is_host, int grid, bool caller_buffer) {
    if (!bytes || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return 0;
    void* alias = nullptr;
    if (hipHostGetDevicePointer(&alias, const_cast<void*>(dst_is_host ? dst : src), 0) != hipSuccess || !alias) {
        (void)hipGetLastError();
        return 0;
    }
    if (caller_buffer) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)alias) != hipSuccess || !base || !size) {
            (void)hipGetLastError();
            return 0;
        }
        const uintptr_t lo = (uintptr_t)base, hi = lo + size, a0 = (uintptr_t)alias;
        if (a0 < lo || a0 + bytes > hi) return -1;
    }
    const size_t n16 = bytes >> 4;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::min<size_t>((size_t)grid, (n16 + 255) / 256 + 1)), dim3(256), 0, c->stream,
                       static_cast<const v4i_cp*>(dst_is_host ? src : alias), static_cast<v4i_cp*>(dst_is_host ? alias : dst), n16, bytes & 15);
    return 1;
}

// How many workgroups a bus transfer of this context gets. One bounded launch per direction moves 45-57 GB/s each way; many
// waves of BOTH kinds in flight -- what several contexts streaming frames produce -- drop the link to 55 GB/s for the two
// directions together (tools/ubench/pcie_duplex: 128 + 128 workgroups 90 GB/s, 1024 + 1024 55 GB/s). So the grid shrinks
// with the number of live contexts that feed frames through the mapped planes (g_feeders): alone 128 workgroups, eight
// contexts 32 (in) / 16 (out) each -- streaming 4K frames: 1.37 ms per frame against 1.46 with 128 everywhere.
std::atomic<int> g_feeders[64];
int bus_grid(const jxl_ctx* c, bool out) {
    const int n = std::max(1, c->device >= 0 && c->device < 64 ? g_feeders[c->device].load(std::memory_order_relaxed) : 1);
    return n <= 2 ? 128 : std::max(out ? 16 : 32, (out ? 128 : 256) / n);
}

// three EPF iterations as two launches (run_frame; JXL_EPF3_SPLIT=0: one launch)
bool epf3_split_on() {
    static const bool v = !(getenv("JXL_EPF3_SPLIT") && atoi(getenv("JXL_EPF3_SPLIT")) == 0);
    return v;
}
bool is_small(int t) { return JXL_TT[t].ph == 8 && JXL_TT[t].pw == 8; }
bool is_large(int t) { return JXL_TT[t].ph >= 128 || JXL_TT[t].pw >= 128; }

// ---- table staging buffer (jxl_ctx::h_tab) ----
constexpr size_t kTabAlign = 256;
inline size_t tab_up(size_t v) { return (v + kTabAlign - 1) & ~(kTabAlign - 1); }
// the queued transfer of the staging buffer has finished reading it
void tab_wait(jxl_ctx* c) {
    if (c->tab_inflight && c->tab_ev) (void)hipEventSynchronize(c->tab_ev);
    c->tab_inflight = false;
}
// at least `bytes` of staging, the first `keep` bytes preserved; grows geometrically and never shrinks (a page-locked
// allocation costs milliseconds and hipHostFree synchronises the device)
bool tab_reserve(jxl_ctx* c, size_t bytes, size_t keep) {
    if (bytes <= c->h_tab_cap) return true;
    tab_wait(c);
    const size_t cap = std::max(bytes, c->h_tab_cap + c->h_tab_cap / 2);
    void* q = nullptr;
    bool pinned = true;
    if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess || !q) {
        (void)hipGetLastError();
        static bool said = false;
        if (!said && getenv("JXL_PREPARE_TIMING")) fprintf(stderr, "[prepare] page-locking %zu bytes failed: pageable staging\n", cap);
        said = true;
        q = malloc(cap);
        pinned = false;
        if (!q) return false;
    }
    if (keep && c->h_tab) memcpy(q, c->h_tab, std::min(keep, c->h_tab_cap));
    if (c->h_tab) {
        if (c->h_tab_pinned) (void)hipHostFree(c->h_tab);
        else free(c->h_tab);
    }
    c->h_tab = static_cast<char*>(q);
    c->h_tab_cap = cap;
    c->h_tab_pinned = pinned;
    return true;
}
// point the fixed-size grids at their sections
void tab_bind_fixed(jxl_ctx* c, size_t nc, size_t nt) {
    c->h_hf_mul = {reinterpret_cast<int32_t*>(c->h_tab + c->off_hfm), nc};
    c->h_sharp = {reinterpret_cast<int32_t*>(c->h_tab + c->off_sharp), nc};
    c->h_kx = {reinterpret_cast<float*>(c->h_tab + c->off_kx), nt};
    c->h_kb = {reinterpret_cast<float*>(c->h_tab + c->off_kb), nt};
    for (int i = 0; i < 3; i++) c->h_lf[i] = {reinterpret_cast<float*>(c->h_tab + c->off_lf[i]), nc};
}
// diagnostics (JXL_PREPARE_TIMING): host time of a call's sections, to stderr
struct SectTimer {
    const char* who;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit SectTimer(const char* w) : who(w) {
        static const bool e = getenv("JXL_PREPARE_TIMING") != nullptr;
        on = e;
        if (on) t = std::chrono::steady_clock::now();
    }
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[%s] %-28s %.3f ms\n", who, what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// begin_frame: lay the fixed sections out, make room (plus a first guess for the block records), fill the defaults
bool tab_begin_frame(jxl_ctx* c, size_t nc, size_t nt) {
    SectTimer tm("begin");
    size_t o = 0;
    c->off_hfm = o; o = tab_up(o + 4 * nc);
    c->off_sharp = o; o = tab_up(o + 4 * nc);
    c->off_kx = o; o = tab_up(o + 4 * nt);
    c->off_kb = o; o = tab_up(o + 4 * nt);
    for (int i = 0; i < 3; i++) { c->off_lf[i] = o; o = tab_up(o + 4 * nc); }
    c->tab_fixed = o;
    tab_wait(c);  // the previous frame's transfer may still be reading the buffer
    tm.mark("tab_wait");
    if (!tab_reserve(c, o + sizeof(DevBlock) * nc / 2 + 65536, 0)) return false;
    tab_bind_fixed(c, nc, nt);
    tm.mark("reserve");
    std::fill(c->h_hf_mul.begin(), c->h_hf_mul.end(), 1);
    memset(c->h_sharp.data(), 0, 4 * nc);
    memset(c->h_kx.data(), 0, 4 * nt);
    memset(c->h_kb.data(), 0, 4 * nt);
    for (int i = 0; i < 3; i++) memset(c->h_lf[i].data(), 0, 4 * nc);
    tm.mark("defaults");
    return true;
}

// Bin the varblocks and compute the CfL cache-order masks; upload side tables.
jxl_status finalize_tables(jxl_ctx* c) {
    if (!c->tables_dirty) return JXL_OK;
    static const bool ptime = getenv("JXL_PREPARE_TIMING") != nullptr;  // diagnostics: host time of the sections, to stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!ptime) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[prepare] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const int lrs = ceil_div(c->W, 2048), lcs = ceil_div(c->H, 2048);
    for (int i = 0; i < lrs * lcs; i++)
        if (!c->lfg_set[i]) return fail(c, JXL_ERR_STATE, "LF group %d was never set", i);
    if (!c->have_weights) return fail(c, JXL_ERR_STATE, "quant weights were never set");
    const int grs = ceil_div(c->W, 256), gcs = ceil_div(c->H, 256);
    // Reference visiting order: groups in raster order (Frame.java:367-373), inside a group the LF group's blockList order
    // filtered by the group (HFCoefficients.java:76-85). The per-type block lists are kept in that order (group-major: an item's
    // blocks are neighbours in the frame). r4: two passes over the block lists instead of four (count by group, scatter by group,
    // bin with push_back, concatenate) --
    //   1. count per (group, type); the counts become the cursors of every (group, type) run inside the FINAL layout of h_blocks;
    //   2. every block gets its CfL mask and hfMultiplier and goes straight to its slot.
    // Pass 2 walks the LF groups' lists as they are. That is enough for the masks: a 64x64 tile lies in one group, only ONE block
    // contains its origin, and the mask bit of (block, tile) asks whether that block was visited earlier IN THE SAME GROUP --
    // the filtered list order, which the LF group's list order preserves.
    const int n_groups = grs * gcs;
    constexpr int NTY = JXL_NUM_TRANSFORM_TYPES;
    size_t t_count[NTY] = {};
    size_t n_all = 0;
    std::vector<uint32_t> cur((size_t)n_groups * NTY, 0);
    for (int li = 0; li < lrs * lcs; li++) {
        n_all += c->lfg_blocks[li].size();
        for (const DevBlock& b : c->lfg_blocks[li]) cur[(size_t)((b.cy >> 5) * grs + (b.cx >> 5)) * NTY + b.type]++;
    }
    for (int g = 0; g < n_groups; g++)
        for (int t = 0; t < NTY; t++) t_count[t] += cur[(size_t)g * NTY + t];
    // EPF sharpness range (Frame.java:565-566): found here, once per frame description; reported by jxl_vardct_run when the frame
    // runs the filter (it used to walk all cells on EVERY run)
    c->sharp_is_bad = false;
    for (int32_t v : c->h_sharp)
        if (v < 0 || v > 7) {
            c->sharp_is_bad = true;
            c->sharp_bad = v;
            break;
        }
    mark("counts");
    // final layout of h_blocks: [8x8-footprint types..., medium types..., large types...] (what lay_out / the large-block path expect)
    uint32_t first_of_all[NTY] = {};
    {
        uint32_t o = 0;
        for (int pass = 0; pass < 3; pass++)
            for (int t = 0; t < NTY; t++) {
                const int cls = is_large(t) ? 2 : is_small(t) ? 0 : 1;
                if (cls != pass) continue;
                first_of_all[t] = o;
                o += (uint32_t)t_count[t];
            }
    }
    for (int t = 0; t < NTY; t++) {
        uint32_t o = first_of_all[t];
        for (int g = 0; g < n_groups; g++) {
            const uint32_t n = cur[(size_t)g * NTY + t];
            cur[(size_t)g * NTY + t] = o;
            o += n;
        }
    }
    c->h_blocks.resize(n_all);
    {
        // do the varblocks tile the frame? (they do in every valid stream: HFMetadata places a block on every cell; blocks cannot
        // overlap -- the front-end's dctSelect grid has one owner per cell). Then the transforms write every output sample.
        size_t cells = 0;
        for (int t = 0; t < NTY; t++) cells += t_count[t] * (size_t)(JXL_TT[t].ph / 8) * (size_t)(JXL_TT[t].pw / 8);
        c->blocks_cover = !c->sub && cells == (size_t)c->bh * c->bw;
    }
    // ... if none of them overlap: the C ABI accepts any block list, so the claim is checked with one mark per cell (a double mark
    // plus holes of the same total area would leave cells that are neither written nor zero-filled: stale samples of the
    // previous frame on this context where the reference's ImageBuffer starts zeroed)
    if (c->blocks_cover) c->cell_mark.assign((size_t)c->bh * c->bw, 0);
    {
        std::vector<int32_t> stamp((size_t)c->th * c->tw, -1);
        DevBlock* out = c->h_blocks.data();
        for (int li = 0; li < lrs * lcs; li++)
            for (const DevBlock& b0 : c->lfg_blocks[li]) {
                DevBlock b = b0;
                const int g = (b.cy >> 5) * grs + (b.cx >> 5);
                const int ph = JXL_TT[b.type].ph, pw = JXL_TT[b.type].pw;
                const int py0 = b.cy * 8, px0 = b.cx * 8;
                if (py0 + ph > c->H || px0 + pw > c->W)
                    return fail(c, JXL_ERR_INVALID_BITSTREAM, "varblock (%d,%d) type %u leaves the frame", b.cy, b.cx, b.type);
                const int ty0 = py0 >> 6, tx0 = px0 >> 6, ty1 = (py0 + ph - 1) >> 6, tx1 = (px0 + pw - 1) >> 6;
                if (ty1 - ty0 > 4 || tx1 - tx0 > 4) return fail(c, JXL_ERR_INVALID_BITSTREAM, "varblock spans too many tiles");
                if (c->blocks_cover) {
                    uint8_t* m = c->cell_mark.data() + (size_t)b.cy * c->bw + b.cx;
                    uint8_t seen = 0;
                    for (int y = 0; y < ph / 8; y++, m += c->bw)
                        for (int x = 0; x < pw / 8; x++) {
                            seen |= m[x];
                            m[x] = 1;
                        }
                    if (seen) c->blocks_cover = false;  // overlapping varblocks: some cell is uncovered, the planes are zero-filled
                }
                uint32_t mask = 0;
                if (ty0 == ty1 && tx0 == tx1) {  // one tile (every 8x8 block: five blocks in six of a photographic frame)
                    int32_t& sp = stamp[(size_t)ty0 * c->tw + tx0];
                    if (!((py0 | px0) & 63)) sp = g;
                    else if (sp != g) mask = 1u;
                } else {
                    for (int ty = ty0; ty <= ty1; ty++)
                        for (int tx = tx0; tx <= tx1; tx++) {
                            const bool origin_inside = ty * 64 >= py0 && tx * 64 >= px0;  // (< py0+ph, px0+pw by the loop bounds)
                            if (origin_inside) stamp[(size_t)ty * c->tw + tx] = g;
                            else if (stamp[(size_t)ty * c->tw + tx] != g) mask |= 1u << ((ty - ty0) * 5 + (tx - tx0));
                        }
                }
                b.cfl_zero = mask;
                b.hf_mul = c->h_hf_mul[(size_t)b.cy * c->bw + b.cx];
                out[cur[(size_t)g * NTY + b.type]++] = b;
            }
    }
    // the per-type lists as windows of h_blocks
    struct BlockList {
        const DevBlock* p = nullptr;
        size_t n = 0;
        bool empty() const { return n == 0; }
        size_t size() const { return n; }
        const DevBlock& operator[](size_t i) const { return p[i]; }
        const DevBlock* begin() const { return p; }
        const DevBlock* end() const { return p + n; }
    };
    BlockList sm[NTY];
    for (int t = 0; t < NTY; t++) sm[t] = BlockList{c->h_blocks.data() + first_of_all[t], t_count[t]};
    mark("CfL masks + bins by type");
    // layout: [8x8-footprint types..., medium types..., large types...]; expensive items first so that the
    // tail of the single launch is made of cheap workgroups. Chroma-subsampled frames (c->sub) get one such layout per
    // channel: only the blocks aligned to the channel's grid, in the channel's own cell coordinates
    // (HFCoefficients.java:292-297, PassGroup.java:215-221), and one launch set per channel.
    std::vector<WorkItem> items;
    c->type_launches.clear();
    c->special_launches.clear();
    static const int kOrder[] = {18, 19, 20, 5, 10, 11, 4, 8, 9, 6, 7, 0};  // longest-running kernels first
    static const int kSpecial[] = {14, 15, 16, 17, 1, 2, 3, 12, 13};
    bool seg_overflow = false;
    // preplaced: the lists ARE windows of h_blocks at these offsets (frames without chroma subsampling); else they are appended
    auto lay_out = [&](const auto* lists, int channel, const uint32_t* preplaced) {
        std::vector<uint32_t> first_of(JXL_NUM_TRANSFORM_TYPES, 0);
        for (int pass = 0; pass < 2; pass++)
            for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++) {
                if (is_large(t) || lists[t].empty() || is_small(t) != (pass == 0)) continue;
                if (preplaced) {
                    first_of[t] = preplaced[t];
                    continue;
                }
                first_of[t] = (uint32_t)c->h_blocks.size();
                c->h_blocks.insert(c->h_blocks.end(), lists[t].begin(), lists[t].end());
            }
        const uint32_t ch0 = channel < 0 ? 0 : (uint32_t)channel, ch1 = channel < 0 ? 3 : (uint32_t)channel + 1;
        // classes 2 / 3 = the persistent three-channel kernel (k_idct_wg3.hip; 3: the 64-point family) for every METHOD_DCT type
        // above 8x8 of a frame without chroma subsampling; classes 1 / 0 = the per-channel kernels of k_idct.hip (8x8 always; everything for
        // subsampled frames, whose channels have their own geometry)
        static const bool use_wg3 = !(getenv("JXL_IDCT_WG3") && atoi(getenv("JXL_IDCT_WG3")) == 0);
        jxl_ctx::TypeLaunch cl[4] = {{3, channel, {}}, {2, channel, {}}, {1, channel, {}}, {0, channel, {}}};  // launch order: heaviest class first
        // r6: the special 8x8 types of a frame without chroma subsampling are items of the persistent launch too (wg3_special_items):
        // they join class 2's segments and get no launch of their own below
        const bool special_in_wg3 = use_wg3 && channel < 0 && wg3_special_items();
        for (int t : kOrder) {
            if (lists[t].empty()) continue;
            const IdctSegment sg{t, (int)first_of[t], (int)lists[t].size()};
            const int cls = (use_wg3 && channel < 0 && wg3_handles(t)) ? (wg3_big(t) ? 3 : 2) : idct_class_of(t);
            for (auto& l : cl)
                if (l.cls == cls) l.segs.push_back(sg);
        }
        if (special_in_wg3)
            for (int t : kSpecial) {
                if (lists[t].empty() || !wg3_handles(t)) continue;
                for (auto& l : cl)
                    if (l.cls == 2) l.segs.push_back(IdctSegment{t, (int)first_of[t], (int)lists[t].size()});
            }
        for (auto& l : cl) {
            // the launch argument blocks hold kMaxSeg segments (one per type of a class): checked HERE, where the lists are made
            if (l.segs.size() > (size_t)(l.cls >= 2 ? Wg3Args::kMaxSeg : MultiArgs::kMaxSeg)) seg_overflow = true;
            if (!l.segs.empty()) c->type_launches.push_back(std::move(l));
        }
        jxl_ctx::SpecialLaunch sl{(int)items.size(), 0, channel, false};
        // One wave per item = 64 consecutive blocks of one type and channel. The block lists are group-major (256 x 256 px
        // groups in raster order), so item k of every type covers about the same few groups. An 8 x 8 block's rows are 32-byte
        // pieces of 128-byte lines whose other pieces belong to blocks of other types: launched type after type, every line is
        // fetched once per type that touches it (measured: a frame of all nine special types 113 us against 60 us for any one
        // of them; 8K: 46 against 150 Gpx/s). So the items are launched in spatial order instead -- sorted by the group of
        // their first block -- and dealt to the XCDs in runs (workgroup b runs on XCD b % 8, each with its own L2): the items
        // that share lines are in flight together on one L2.
        struct Ord { uint32_t key; WorkItem w; };
        std::vector<Ord> ord;
        const int grs_c = std::max(1, (((channel < 0 ? c->bw : (c->bw >> c->sx[channel])) + 31) >> 5));
        // frames without chroma subsampling: one item = 64 blocks with all three channels (k_idct_special_wg);
        // per-channel launches: one item per channel (the lane-per-block kernel)
        static const bool special_wg = !(getenv("JXL_SPECIAL_WG") && atoi(getenv("JXL_SPECIAL_WG")) == 0);
        const bool wg_items = channel < 0 && special_wg;
        for (int t : kSpecial)
            for (uint32_t o = 0; o < (special_in_wg3 && wg3_handles(t) ? 0u : (uint32_t)lists[t].size()); o += 64)
                for (uint32_t ch = ch0; ch < (wg_items ? ch0 + 1 : ch1); ch++) {
                    const DevBlock& b0 = lists[t][o];
                    ord.push_back(Ord{(uint32_t)((b0.cy >> 5) * grs_c + (b0.cx >> 5)),
                                      WorkItem{(uint32_t)t | (ch << 8), first_of[t] + o, (uint32_t)std::min<size_t>(64, lists[t].size() - o)}});
                }
        static const bool spatial = !(getenv("JXL_SPECIAL_SPATIAL") && atoi(getenv("JXL_SPECIAL_SPATIAL")) == 0);
        if (spatial && ord.size() > 8) {
            std::stable_sort(ord.begin(), ord.end(), [](const Ord& x, const Ord& y) { return x.key < y.key; });
            static const int run_env = getenv("JXL_SPECIAL_RUN") ? std::max(1, atoi(getenv("JXL_SPECIAL_RUN"))) : 0;
            const int run = run_env ? run_env : wg_items ? 12 : 32;  // about the items of the few groups one item spans
            std::vector<WorkItem> q[8];
            for (size_t i = 0; i < ord.size(); i++) q[(i / (size_t)run) % 8].push_back(ord[i].w);
            size_t longest = 0;
            for (auto& v : q) longest = std::max(longest, v.size());
            for (size_t i = 0; i < longest; i++)
                for (int x = 0; x < 8; x++) items.push_back(i < q[x].size() ? q[x][i] : WorkItem{0u, 0u, 0u});  // count 0: the wave exits
        } else {
            for (const Ord& o : ord) items.push_back(o.w);
        }
        sl.wg_items = wg_items;
        sl.n_items = (int)items.size() - sl.items_off;
        if (sl.n_items > 0) c->special_launches.push_back(sl);
    };
    std::vector<int32_t> h_hfm_sub[3];
    if (!c->sub) {
        lay_out(sm, -1, first_of_all);
    } else {
        for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++)
            if (is_large(t) && !sm[t].empty())
                return fail(c, JXL_ERR_UNSUPPORTED, "128/256-edge varblocks in a chroma-subsampled frame");
        // per channel: its own block lists in its own cell coordinates; h_blocks is rebuilt from them
        std::vector<DevBlock> full[JXL_NUM_TRANSFORM_TYPES];
        for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++) full[t].assign(sm[t].begin(), sm[t].end());
        c->h_blocks.clear();
        for (int ch = 0; ch < 3; ch++) {
            const int sy = c->sy[ch], sx = c->sx[ch], bwc = c->bw >> sx, bhc = c->bh >> sy;
            std::vector<DevBlock> sub_lists[JXL_NUM_TRANSFORM_TYPES];
            h_hfm_sub[ch].assign((size_t)bwc * bhc, 1);
            for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++)
                for (const DevBlock& b : full[t]) {
                    const int cy2 = b.cy >> sy, cx2 = b.cx >> sx;
                    if ((cy2 << sy) != b.cy || (cx2 << sx) != b.cx) continue;  // subsampled away
                    if (cy2 * 8 + JXL_TT[t].ph > (c->H >> sy) || cx2 * 8 + JXL_TT[t].pw > (c->W >> sx))
                        return fail(c, JXL_ERR_INVALID_BITSTREAM, "varblock (%d,%d) leaves the subsampled channel %d", b.cy, b.cx, ch);
                    sub_lists[t].push_back(DevBlock{(uint16_t)cy2, (uint16_t)cx2, b.type, 0u, b.hf_mul});
                    h_hfm_sub[ch][(size_t)cy2 * bwc + cx2] = c->h_hf_mul[(size_t)b.cy * c->bw + b.cx];
                }
            lay_out(sub_lists, ch, nullptr);
        }
    }
    if (seg_overflow) return fail(c, JXL_ERR_STATE, "more transform types in one launch class than an argument block holds");
    mark("launch layout");
    if (!c->sub) {  // the large types already sit behind the others (first_of_all)
        size_t n_large = 0;
        for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++)
            if (is_large(t)) n_large += sm[t].size();
        c->large_first = (int)(c->h_blocks.size() - n_large);
    } else {
        c->large_first = (int)c->h_blocks.size();  // (none: refused above)
    }
    c->large_count = (int)c->h_blocks.size() - c->large_first;
    c->llf_first = c->large_first;  // only the 128/256-edge blocks take their LLF from the llf planes (k_llf)
    c->llf_count = c->large_count;
    if (c->sub) {
        // kernels of the previous frame on this (non-blocking) stream may still read hfm_sub, and the source is pageable: wait for
        // them, then copy on the same stream so that this frame's kernels are ordered behind the upload
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int ch = 0; ch < 3; ch++) {
            if (!c->hfm_sub[ch].ensure(4 * std::max<size_t>(1, h_hfm_sub[ch].size()))) return fail(c, JXL_ERR_OOM, "device allocation failed");
            HIP_TRY(c, hipMemcpyAsync(c->hfm_sub[ch].p, h_hfm_sub[ch].data(), 4 * h_hfm_sub[ch].size(), hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // h_hfm_sub is a local
    }
    const size_t nc = (size_t)c->bh * c->bw, nt = (size_t)c->th * c->tw;
    // item lists of the persistent / wave kernels
    std::vector<int> wg3_tab[2];
    {
        static const bool spatial = !(getenv("JXL_WG3_SPATIAL") && atoi(getenv("JXL_WG3_SPATIAL")) == 0);
        for (int k = 0; k < 2; k++) {
            c->wg3_item_count[k] = 0;
            for (const auto& tl : c->type_launches) {
                if (tl.cls != 2 + k) continue;
                wg3_item_table(c->h_blocks.data(), c->bw, tl.segs.data(), (int)tl.segs.size(), k, c->woffs, spatial, wg3_tab[k], wg3_grid_cap(k == 1));
                c->wg3_item_count[k] = (int)(wg3_tab[k].size() / 8);
            }
        }
    }
    mark("item tables");
    // chroma-from-luma factor per 64x64 tile, HFCoefficients.java:177-181: base + factor / colorFactor in float (one IEEE
    // division and one addition, the same two operations the per-sample device code used to perform)
    tab_wait(c);  // first write into the staging buffer: a transfer queued by an earlier finalize of this frame has finished reading it
    {
        const volatile float cf = (float)c->p.color_factor;
        for (size_t i = 0; i < nt; i++) {
            const volatile float qx = (float)c->h_xfy[i] / cf, qb = (float)c->h_bfy[i] / cf;
            c->h_kx[i] = c->p.base_corr_x + qx;
            c->h_kb[i] = c->p.base_corr_b + qb;
        }
    }
    // the variable sections behind the fixed ones: block records, special-kernel items, item lists
    size_t o = c->tab_fixed;
    const size_t off_blocks = o, n_blk = sizeof(DevBlock) * c->h_blocks.size();
    o = tab_up(o + std::max<size_t>(16, n_blk));
    const size_t off_items = o, n_items = sizeof(WorkItem) * items.size();
    o = tab_up(o + std::max<size_t>(16, n_items));
    size_t off_wg3[2];
    for (int k = 0; k < 2; k++) {
        off_wg3[k] = o;
        o = tab_up(o + std::max<size_t>(16, sizeof(int) * wg3_tab[k].size()));
    }
    const size_t total = o;
    if (!tab_reserve(c, total, c->tab_fixed)) return fail(c, JXL_ERR_OOM, "host allocation failed (table staging)");
    tab_bind_fixed(c, nc, nt);  // (the buffer may have moved)
    if (n_blk) memcpy(c->h_tab + off_blocks, c->h_blocks.data(), n_blk);
    if (n_items) memcpy(c->h_tab + off_items, items.data(), n_items);
    for (int k = 0; k < 2; k++) {
        if (!wg3_tab[k].empty()) memcpy(c->h_tab + off_wg3[k], wg3_tab[k].data(), sizeof(int) * wg3_tab[k].size());
    }
    if (total > c->tab.cap || !c->tab.p) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // kernels of an earlier frame may still read the old arena
        if (!c->tab.ensure(total + total / 4)) return fail(c, JXL_ERR_OOM, "device allocation failed (frame tables)");
    }
    mark("staging");
    if (!c->tab_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->tab_ev, hipEventDisableTiming));
    static const bool tab_zero = !(getenv("JXL_TABLE_ZEROCOPY") && atoi(getenv("JXL_TABLE_ZEROCOPY")) == 0);
    if (!(tab_zero && c->h_tab_pinned && copy_zero(c, c->tab.p, c->h_tab, total, false, 64, false) == 1))
        HIP_TRY(c, hipMemcpyAsync(c->tab.p, c->h_tab, total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->tab_ev, c->stream));
    c->tab_inflight = true;
    {
        char* d = static_cast<char*>(c->tab.p);
        c->hf_mul.view(d + c->off_hfm, 4 * nc);
        c->sharp.view(d + c->off_sharp, 4 * nc);
        c->xfy.view(d + c->off_kx, 4 * nt);
        c->bfy.view(d + c->off_kb, 4 * nt);
        for (int i = 0; i < 3; i++) c->lf[i].view(d + c->off_lf[i], 4 * nc);
        c->blocks.view(d + off_blocks, std::max<size_t>(16, n_blk));
        c->items.view(d + off_items, std::max<size_t>(16, n_items));
        for (int k = 0; k < 2; k++) c->wg3_items[k].view(d + off_wg3[k], sizeof(int) * wg3_tab[k].size());
    }
    // row f1: LF groups handed over as integers are dequantised + smoothed on the device, over the uploaded planes
    for (const auto& job : c->lf_jobs) {
        const jxl_lfquant_desc& d = job.d;
        if (c->sub) {  // per channel, own geometry, no CfL, no smoothing (LFCoefficients.java:66-75 only)
            for (int ch = 0; ch < 3; ch++) {
                const int h2 = d.cells_h >> c->sy[ch], w2 = d.cells_w >> c->sx[ch], bw2 = c->bw >> c->sx[ch];
                const size_t n2 = (size_t)h2 * w2;
                if (!c->lfq_tmp[ch].ensure(4 * std::max<size_t>(1, n2))) return fail(c, JXL_ERR_OOM, "device allocation failed (LF image)");
                HIP_TRY(c, hipMemcpyAsync(c->lfq_tmp[ch].p, job.q[ch].data(), 4 * n2, hipMemcpyHostToDevice, c->stream));
                launch_lf_dequant_plain(c->lfq_tmp[ch].as<int32_t>(), c->lf[ch].as<float>(), h2, w2,
                                        (int64_t)((d.lfg_y * 256) >> c->sy[ch]) * bw2 + ((d.lfg_x * 256) >> c->sx[ch]), bw2,
                                        d.scaled_dequant[ch], d.extra_precision, c->stream);
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            continue;
        }
        const size_t n = (size_t)d.cells_h * d.cells_w;
        const int32_t* dq[3];
        for (int ch = 0; ch < 3; ch++) {
            if (!c->lfq_tmp[ch].ensure(4 * std::max<size_t>(1, n))) return fail(c, JXL_ERR_OOM, "device allocation failed (LF image)");
            HIP_TRY(c, hipMemcpyAsync(c->lfq_tmp[ch].p, job.q[ch].data(), 4 * n, hipMemcpyHostToDevice, c->stream));
            dq[ch] = c->lfq_tmp[ch].as<int32_t>();
        }
        float* lfp[3] = {c->lf[0].as<float>(), c->lf[1].as<float>(), c->lf[2].as<float>()};
        launch_lf_dequant(dq, lfp, d.cells_h, d.cells_w, (int64_t)(d.lfg_y * 256) * c->bw + d.lfg_x * 256, c->bw, d.scaled_dequant,
                          d.extra_precision, c->p.base_corr_x, c->p.base_corr_b, c->p.color_factor, d.x_factor_lf, d.b_factor_lf,
                          d.adaptive_smoothing, c->stream);
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // lfq_tmp is reused by the next job
    }
    // llf starts as a copy of lf (the LLF of an 8x8 block is its LF sample); k_llf overwrites the cells of the 128 / 256-edge
    // blocks and (JXL_WG3_LLF_IN_ITEM=0) k_llf_wg3 those of every block above 8x8. With neither, nothing ever writes the llf
    // planes: the kernels are handed the lf planes under both names and the three plane copies are not made (r4)
    c->llf_alias = c->large_count == 0 && wg3_llf_in_item();
    if (!c->llf_alias)
        for (int ch = 0; ch < 3; ch++)
            HIP_TRY(c, hipMemcpyAsync(c->llf[ch].p, c->lf[ch].p, 4 * nc, hipMemcpyDeviceToDevice, c->stream));
    // no synchronisation here: the staging buffer is only written again behind tab_wait(), the LF jobs have waited for their own
    // sources, and everything that reads the tables is queued on this stream behind the transfer
    if (!c->h_tab_pinned) HIP_TRY(c, hipStreamSynchronize(c->stream));
    mark("transfer queued + LF");
    c->tables_dirty = false;
    static std::atomic<uint64_t> g_tables_gen{0};  // process-wide: a new context at a recycled address never matches an old key
    c->tables_gen = ++g_tables_gen;
    return JXL_OK;
}

void fill_dev_frame(const jxl_ctx* c, DevFrame& f) {
    f.no_cfl = 0;
    f.width = c->W; f.height = c->H; f.bw = c->bw; f.bh = c->bh; f.tw = c->tw; f.th = c->th;
    for (int ch = 0; ch < 3; ch++) {
        f.coeff[ch] = c->coeff[ch].as<int32_t>();
        f.lf[ch] = c->lf[ch].as<float>();
        f.llf[ch] = c->llf_alias ? c->lf[ch].as<float>() : c->llf[ch].as<float>();
        f.scale_factor[ch] = c->p.scale_factor[ch];
        f.quant_bias[ch] = c->p.quant_bias[ch];
    }
    f.hf_mul = c->hf_mul.as<int32_t>();
    f.sharpness = c->sharp.as<int32_t>();
    f.kx_tab = c->xfy.as<float>();
    f.kb_tab = c->bfy.as<float>();
    f.weights = c->weights.as<float>();
    f.weights_t = c->weights_t.as<float>();
    memcpy(f.woffs, c->woffs, sizeof f.woffs);
    f.lut = c->lut.as<float>();
    f.quant_bias_numerator = c->p.quant_bias_numerator;
    f.base_corr_x = c->p.base_corr_x;
    f.base_corr_b = c->p.base_corr_b;
    f.color_factor_f = (float)c->p.color_factor;
}

const float kStepMultiplier = 1.65f * 4.0f * (1.0f - (float)std::sqrt(0.5));  // Frame.java:545

// TF_PQ.fromLinear (TransferFunction.java:83-87) in long double, with the reference's constants
long double pq_ld(long double x) {
    const long double d = powl(x, 0.159423828125L);
    return powl((0.8359375L + 18.8515625L * d) / (1.0L + 18.6875L * d), 78.84375L);
}

EpfParams make_epf(const float cs[3], float pass0, float pass2, float border, int iter, float inv_sigma_modular) {
    EpfParams e;
    for (int i = 0; i < 3; i++) e.channel_scale[i] = cs[i];
    e.sigma_scale = iter == 0 ? kStepMultiplier * pass0 : iter == 2 ? kStepMultiplier * pass2 : kStepMultiplier;  // :592-598
    e.border_sad_mul = border;
    e.inv_sigma_modular = inv_sigma_modular;
    return e;
}

XybParams make_xyb(const float m[9], const float ob[3], const float cob[3], float intensity_target) {
    XybParams x;
    const float itScale = 255.0f / intensity_target;  // OpsinInverseMatrix.java:108
    for (int i = 0; i < 9; i++) x.sm[i] = m[i] * itScale;
    for (int i = 0; i < 3; i++) {
        x.ob[i] = ob[i];
        x.cob[i] = -cob[i];
    }
    return x;
}

}  // namespace

namespace jxl {
// The PQ segment table of jxl_fastpow.h (fp_tf_pq_tab): segment idx covers the floats whose bits >> 16 == idx + (87 << 7),
// i.e. [x0, x0 + w) with x0 = 2^(e-127) (1 + m / 128), w = 2^(e-127) / 128; quadratic through the Chebyshev nodes of the segment
// around its midpoint, a0 as a float pair.
void build_pq_table(float* out) {
    const long double cn = 0.86602540378443864676L;  // cos(pi / 6)
    for (int i = 0; i < (129 - 87) * 128; i++) {
        const uint32_t b0 = ((uint32_t)i + (87u << 7)) << 16;
        float x0f, xmf;
        const uint32_t bm = b0 | 0x00008000u;
        memcpy(&x0f, &b0, 4);
        memcpy(&xmf, &bm, 4);
        const long double xm = xmf, h = (long double)xmf - (long double)x0f, tn = h * cn;
        const long double y0 = pq_ld(xm), yp = pq_ld(xm + tn), ym = pq_ld(xm - tn);
        const long double a1 = (yp - ym) / (2.0L * tn), a2 = (yp + ym - 2.0L * y0) / (2.0L * tn * tn);
        const float a0hi = (float)y0;
        out[4 * i + 0] = a0hi;
        out[4 * i + 1] = (float)(y0 - (long double)a0hi);
        out[4 * i + 2] = (float)a1;
        out[4 * i + 3] = (float)a2;
    }
}
}  // namespace jxl

namespace jxl {
// The reference's composite for an 8-bit sRGB sample: TF_SRGB.fromLinearF (TransferFunction.java:39-44) then
// ImageBuffer.castToIntWithMax(255) (ImageBuffer.java:129-147), in the reference's own operations (this file is built with
// -ffp-contract=off; the double pow is the host libm's, as in the oracle)
static int srgb8_ref(float f) {
    const volatile float t = f < 0.00313066844250063f ? f * 12.92f : 1.055f * (float)std::pow((double)f, 0.4166666666666667) + -0.055f;
    const volatile float v = t * 255.0f + 0.5f;
    if (v != v) return 0;
    if (v >= 255.0f) return 255;
    if (v <= 0.0f) return 0;
    return (int)v;
}
// fp_srgb8's table (jxl_fastpow.h): per segment {base level, up to three thresholds (+inf: none)}
bool build_srgb8_table(float* out) {
    const float inf = std::numeric_limits<float>::infinity();
    for (int i = 0; i < (127 - 118) * 128; i++) {
        const uint32_t b0 = ((uint32_t)i + (118u << 7)) << 16, b1 = b0 + 0x10000u;  // the segment's floats: bits in [b0, b1)
        auto at = [](uint32_t b) { float f; memcpy(&f, &b, 4); return srgb8_ref(f); };
        const int base = at(b0), last = at(b1 - 1);
        if (last < base || last - base > 3) return false;
        out[4 * i] = (float)base;
        for (int k = 1; k <= 3; k++) {
            float thr = inf;
            if (base + k <= last) {  // smallest bit pattern in the segment whose level is >= base + k (levels do not decrease)
                uint32_t lo = b0, hi = b1 - 1;  // at(lo) < base + k <= at(hi)
                while (hi - lo > 1) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (at(mid) >= base + k) hi = mid; else lo = mid;
                }
                memcpy(&thr, &hi, 4);
            }
            out[4 * i + k] = thr;
        }
    }
    return true;
}
}  // namespace jxl

namespace jxl {
// TF_PQ.fromLinear (TransferFunction.java:83-87) then ImageBuffer.castToIntWithMax(65535), in the reference's own operations
static int pq16_ref(float f) {
    const double d = std::pow((double)f, 0.159423828125);
    const volatile float t = (float)std::pow((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375);
    const volatile float v = t * 65535.0f + 0.5f;
    if (v != v) return 0;
    if (v >= 65535.0f) return 65535;
    if (v <= 0.0f) return 0;
    return (int)v;
}
// thr[k], k = 1..65535: the smallest float whose level is >= k (bisection over the bit patterns of [+0, 1.0f]; the level does
// not decrease with the input); thr[0] = -inf, thr[65536] = +inf. ~4 M pow calls: on 8 threads, once per process, on first use.
void build_pq16_thresholds(float* out) {
    const float inf = std::numeric_limits<float>::infinity();
    out[0] = -inf;
    out[65536] = inf;
    const uint32_t one = 0x3F800000u;
    auto at = [](uint32_t b) { float f; memcpy(&f, &b, 4); return pq16_ref(f); };
    auto work = [&](int t, int nt) {
        for (int k = 1 + t; k <= 65535; k += nt) {
            uint32_t lo = 0u, hi = one;  // at(lo) = 0 < k <= 65535 = at(hi)
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (at(mid) >= k) hi = mid; else lo = mid;
            }
            memcpy(&out[k], &hi, 4);
        }
    };
    const int nt = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t, nt);
    work(0, nt);
    for (auto& x : th) x.join();
}
// PQ -> 8 bit: thr[1..255] appended to the 16-bit thresholds' buffer (offset 65537: thr8[0] = -inf, thr8[256] = +inf)
static int pq8_ref(float f) {
    const double d = std::pow((double)f, 0.159423828125);
    const volatile float t = (float)std::pow((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375);
    const volatile float v = t * 255.0f + 0.5f;
    if (v != v) return 0;
    if (v >= 255.0f) return 255;
    if (v <= 0.0f) return 0;
    return (int)v;
}
void build_pq8_thresholds(float* out /* [257] */) {
    const float inf = std::numeric_limits<float>::infinity();
    out[0] = -inf;
    out[256] = inf;
    auto at = [](uint32_t b) { float f; memcpy(&f, &b, 4); return pq8_ref(f); };
    for (int k = 1; k <= 255; k++) {
        uint32_t lo = 0u, hi = 0x3F800000u;
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (at(mid) >= k) hi = mid; else lo = mid;
        }
        memcpy(&out[k], &hi, 4);
    }
}

// TF_SRGB.fromLinearF then ImageBuffer.castToIntWithMax(65535), in the reference's own operations
static int srgb16_ref(float f) {
    const volatile float t = f < 0.00313066844250063f ? f * 12.92f : 1.055f * (float)std::pow((double)f, 0.4166666666666667) + -0.055f;
    const volatile float v = t * 65535.0f + 0.5f;
    if (v != v) return 0;
    if (v >= 65535.0f) return 65535;
    if (v <= 0.0f) return 0;
    return (int)v;
}
// fp_srgb16's tables: quadratic segments of c1 x^(1/2.4) - c0 over [2^-9, 1) (as build_pq_table), then the 65 537 thresholds
void build_srgb16_table(float* out) {
    const long double cn = 0.86602540378443864676L, c1 = (long double)1.055f, c0 = (long double)0.055f;
    auto T = [&](long double x) { return c1 * powl(x, 1.0L / 2.4L) - c0; };
    for (int i = 0; i < (127 - 118) * 128; i++) {
        const uint32_t b0 = ((uint32_t)i + (118u << 7)) << 16, bm = b0 | 0x00008000u;
        float x0f, xmf;
        memcpy(&x0f, &b0, 4);
        memcpy(&xmf, &bm, 4);
        const long double xm = xmf, h = (long double)xmf - (long double)x0f, tn = h * cn;
        const long double y0 = T(xm), yp = T(xm + tn), ym = T(xm - tn);
        const float a0hi = (float)y0;
        out[4 * i + 0] = a0hi;
        out[4 * i + 1] = (float)(y0 - (long double)a0hi);
        out[4 * i + 2] = (float)((yp - ym) / (2.0L * tn));
        out[4 * i + 3] = (float)((yp + ym - 2.0L * y0) / (2.0L * tn * tn));
    }
    float* thr = out + kSrgb8TableFloats;
    const float inf = std::numeric_limits<float>::infinity();
    thr[0] = -inf;
    thr[65536] = inf;
    auto at = [](uint32_t b) { float f; memcpy(&f, &b, 4); return srgb16_ref(f); };
    auto work = [&](int t, int nt) {
        for (int k = 1 + t; k <= 65535; k += nt) {
            uint32_t lo = 0u, hi = 0x3F800000u;  // at(+0) = 0 < k <= 65535 = at(1.0f)
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (at(mid) >= k) hi = mid; else lo = mid;
            }
            memcpy(&thr[k], &hi, 4);
        }
    };
    const int nt = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t, nt);
    work(0, nt);
    for (auto& x : th) x.join();
}
}  // namespace jxl

// Diagnostic (tools/clock_probe.py, r6): the shader clock the chip actually holds while other work runs. One wave on a stream of its
// own spins for `us` microseconds of s_memrealtime (a constant 100 MHz counter) and reports how many shader cycles (s_memtime) went by:
// clock = cycles / ticks x 100 MHz (guide, DVFS give-back (6)). Not part of the C-ABI of include/jxlatte_amd.h.
__global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}
// CPU test hook (tests/test_item_table_cpu.py, no device): the item table wg3_item_table builds for a frame whose varblocks are
// n_blocks[k] blocks of type types[k], laid out type after type in raster order of a frame frame_bw cells wide (positions only feed the
// spatial sort). out: 8 ints per table position (type, first block, blocks, geometry, 3 weight offsets, 0; type -2 = hole). Returns the
// number of positions, or -1 if `cap` positions do not hold them.
extern "C" int jxl_debug_wg3_item_table(const int32_t* types, const int32_t* n_blocks, int n_types, int frame_bw, int grid, int32_t* out, int cap) {
    std::vector<DevBlock> blocks;
    std::vector<IdctSegment> segs;
    int cell = 0;
    for (int k = 0; k < n_types; k++) {
        const int t = types[k];
        if (t < 0 || t >= JXL_NUM_TRANSFORM_TYPES || n_blocks[k] <= 0) return -1;
        segs.push_back(IdctSegment{t, (int)blocks.size(), n_blocks[k]});
        const int cw = JXL_TT[t].pw / 8, chh = JXL_TT[t].ph / 8;
        for (int i = 0; i < n_blocks[k]; i++, cell += cw * chh)
            blocks.push_back(DevBlock{(uint16_t)((cell / std::max(1, frame_bw)) & 0xffff), (uint16_t)(cell % std::max(1, frame_bw)), (uint32_t)t, 0u, 1});
    }
    int32_t woffs[3 * 17] = {};
    std::vector<int> tab;
    wg3_item_table(blocks.data(), frame_bw, segs.data(), (int)segs.size(), 0, woffs, true, tab, grid);
    const int n = (int)(tab.size() / 8);
    if (n > cap) return -1;
    std::copy(tab.begin(), tab.end(), out);
    return n;
}
extern "C" int jxl_debug_clock_probe(int device, double us, double* mhz) {
    static hipStream_t s = nullptr;
    static unsigned long long* d = nullptr;
    if (hipSetDevice(device) != hipSuccess) return -1;
    if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!d && hipMalloc(&d, 16) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, (unsigned long long)(us * 100.0), d);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    *mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return 0;
}
extern "C" void jxl_debug_pq_table(float* out) { jxl::build_pq_table(out); }  // CPU tests: the table without a device
extern "C" void jxl_debug_pq16_thresholds(float* out) { jxl::build_pq16_thresholds(out); }
extern "C" void jxl_debug_srgb16_table(float* out) { jxl::build_srgb16_table(out); }
extern "C" void jxl_debug_pq8_thresholds(float* out) { jxl::build_pq8_thresholds(out); }
extern "C" int jxl_debug_srgb8_table(float* out) { return jxl::build_srgb8_table(out) ? 0 : -1; }
extern "C" int jxl_debug_srgb8_ref(float f) { return jxl::srgb8_ref(f); }

namespace {
int out_elem_size(int fmt) { return (fmt == JXL_OUT_U16 || fmt == JXL_OUT_RGB16) ? 2 : (fmt == JXL_OUT_U8 || fmt == JXL_OUT_RGB8) ? 1 : 4; }
int out_max_value(int fmt) { return (fmt == JXL_OUT_U16 || fmt == JXL_OUT_RGB16) ? 65535 : (fmt == JXL_OUT_U8 || fmt == JXL_OUT_RGB8) ? 255 : 0; }
bool out_interleaved(int fmt) { return fmt == JXL_OUT_RGB8 || fmt == JXL_OUT_RGB16; }

// device thresholds of PQ -> 16 bit for this context, or null (other transfer / range, JXL_PQ16_F64=1, allocation failure: the
// caller then quantises the float result). Built once per process on first use (~4 M pow calls on 8 host threads).
const float* pq16_thresholds_for(jxl_ctx* c, int transfer, int max_value) {
    if (transfer != JXL_TRANSFER_PQ || (max_value != 65535 && max_value != 255) || !c->pq_tab.p) return nullptr;
    static const bool off = getenv("JXL_PQ16_F64") != nullptr;
    if (off) return nullptr;
    if (c->pq16_thr.p) return c->pq16_thr.as<float>();
    static std::once_flag once;
    static std::vector<float> thr;
    std::call_once(once, [] {
        thr.resize(65537 + 257);
        build_pq16_thresholds(thr.data());
        build_pq8_thresholds(thr.data() + 65537);
    });
    if (!c->pq16_thr.ensure(sizeof(float) * thr.size()) ||
        hipMemcpy(c->pq16_thr.p, thr.data(), sizeof(float) * thr.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        c->pq16_thr.release();
        return nullptr;
    }
    return c->pq16_thr.as<float>();
}

// the same for sRGB -> 16 bit (fp_srgb16): segments + thresholds, or null (JXL_SRGB16_F64=1: the double-precision form)
const float* srgb16_table_for(jxl_ctx* c, int transfer, int max_value) {
    if (transfer != JXL_TRANSFER_SRGB || max_value != 65535) return nullptr;
    static const bool off = getenv("JXL_SRGB16_F64") != nullptr;
    if (off) return nullptr;
    if (c->srgb16_tab.p) return c->srgb16_tab.as<float>();
    static std::once_flag once;
    static std::vector<float> tab;
    std::call_once(once, [] {
        tab.resize((size_t)kSrgb8TableFloats + 65537);
        build_srgb16_table(tab.data());
    });
    if (!c->srgb16_tab.ensure(sizeof(float) * tab.size()) ||
        hipMemcpy(c->srgb16_tab.p, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        c->srgb16_tab.release();
        return nullptr;
    }
    return c->srgb16_tab.as<float>();
}

}  // namespace

namespace {
struct Tmp {
    std::vector<void*> v;
    ~Tmp() { for (void* p : v) (void)hipFree(p); }
    template <typename T>
    T* up(const T* host, size_t n) {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(16, n * sizeof(T))) != hipSuccess) return nullptr;
        v.push_back(d);
        if (host && n && hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return (T*)d;
    }
};
jxl_status finish(jxl_ctx* c) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "device error: %s", hipGetErrorString(e));
    return JXL_OK;
}
}  // namespace


// ================================================================================================
extern "C" {

const char* jxl_version(void) { return "jxlatte_amd 0.1 (gfx950)"; }

const char* jxl_last_error(const jxl_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

jxl_status jxl_ctx_create(int32_t device, jxl_ctx** out) {
    if (!out) return fail(nullptr, JXL_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, JXL_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, JXL_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, n);
    jxl_ctx* c = new jxl_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, JXL_ERR_DEVICE, "cannot initialise device %d", device);
    }
    std::vector<float> lut;
    build_lut(lut);
    if (!c->lut.ensure(sizeof(float) * lut.size()) ||
        hipMemcpy(c->lut.p, lut.data(), sizeof(float) * lut.size(), hipMemcpyHostToDevice) != hipSuccess ||
        !c->bad_flag.ensure(sizeof(int))) {
        jxl_ctx_destroy(c);
        return fail(nullptr, JXL_ERR_DEVICE, "cannot upload the cosine LUT");
    }
    {
        static std::vector<float> pq;  // the same for every context of the process (contexts may be created from several threads)
        static std::once_flag pq_once;
        std::call_once(pq_once, [] {
            pq.resize(kPqTableFloats);
            build_pq_table(pq.data());
        });
        if (!getenv("JXL_PQ_F64")) {  // experiment knob: keep the double-precision PQ
            if (!c->pq_tab.ensure(sizeof(float) * pq.size()) ||
                hipMemcpy(c->pq_tab.p, pq.data(), sizeof(float) * pq.size(), hipMemcpyHostToDevice) != hipSuccess) {
                jxl_ctx_destroy(c);
                return fail(nullptr, JXL_ERR_DEVICE, "cannot upload the PQ table");
            }
        }
    }
    {
        static std::vector<float> st;  // the same for every context of the process
        static bool st_ok = false;
        static std::once_flag st_once;
        std::call_once(st_once, [] {
            st.resize(kSrgb8TableFloats);
            st_ok = build_srgb8_table(st.data());
        });
        if (st_ok && !getenv("JXL_SRGB8_F64")) {  // experiment knob: keep the double-precision form for 8-bit sRGB output too
            if (!c->srgb8_tab.ensure(sizeof(float) * st.size()) ||
                hipMemcpy(c->srgb8_tab.p, st.data(), sizeof(float) * st.size(), hipMemcpyHostToDevice) != hipSuccess) {
                jxl_ctx_destroy(c);
                return fail(nullptr, JXL_ERR_DEVICE, "cannot upload the sRGB table");
            }
        }
    }
    for (int i = 0; i < jxl_ctx::kEvSlots; i++)
        for (int j = 0; j < 3; j++) (void)hipEventCreate(&c->ev[i][j]);
    for (int i = 0; i < jxl_ctx::kEvSlots; i++)
        for (int j = 0; j < 2; j++) (void)hipEventCreate(&c->kev[i][j]);
    (void)hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->llf_ev, hipEventDisableTiming);
    if (const char* e = getenv("JXL_AUX_STREAMS")) c->n_aux = std::max(0, std::min((int)jxl_ctx::kAux, atoi(e)));
    // (r3: side streams created with hipStreamCreateWithPriority at the highest priority -- meant to keep the few long-running
    // workgroups of the 64-point and special launches from queueing behind the machine-filling main launch -- started those
    // launches ~50 us LATER and ran them one after the other: single frame 310 against 233 us, batch 35.5 against 39.9 Gpx/s.
    // Default-priority streams stay.)
    for (int i = 0; i < c->n_aux; i++) {
        (void)hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming);
    }
    *out = c;
    return JXL_OK;
}

void jxl_ctx_destroy(jxl_ctx* c) {
    if (!c) return;
    if (c->is_feeder) g_feeders[c->device].fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* all[] = {&c->lut, &c->hf_mul, &c->sharp, &c->xfy, &c->bfy, &c->weights, &c->weights_t, &c->inv_sigma, &c->blocks, &c->items,
                     &c->bad_flag};
    for (DevBuf* b : all) b->release();
    for (int i = 0; i < 3; i++) {
        c->coeff[i].release(); c->lf[i].release(); c->llf[i].release(); c->lfq_tmp[i].release(); c->planeA[i].release(); c->planeB[i].release(); c->outbuf[i].release();
    }
    for (int i = 0; i < 3; i++) c->hfm_sub[i].release();
    c->pq_tab.release();
    c->srgb8_tab.release();
    c->pq16_thr.release();
    c->srgb16_tab.release();
    c->tab.release();
    if (c->tab_ev) (void)hipEventDestroy(c->tab_ev);
    for (int i = 0; i < jxl_ctx::kGrpSlots; i++)
        if (c->grp_ev[i]) (void)hipEventDestroy(c->grp_ev[i]);
    if (c->h_grp) (void)hipHostFree(c->h_grp);
    if (c->h_tab) {
        if (c->h_tab_pinned) (void)hipHostFree(c->h_tab);
        else free(c->h_tab);
    }
    for (int i = 0; i < 3; i++) { c->rp[i].release(); c->rp_tmp[i].release(); c->rp_noise[i].release(); }
    for (auto& b : c->mod_bufs) b.release();
    for (int i = 0; i < jxl_ctx::kEvSlots; i++)
        for (int j = 0; j < 3; j++)
            if (c->ev[i][j]) (void)hipEventDestroy(c->ev[i][j]);
    for (int i = 0; i < jxl_ctx::kEvSlots; i++)
        for (int j = 0; j < 2; j++)
            if (c->kev[i][j]) (void)hipEventDestroy(c->kev[i][j]);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->llf_ev) (void)hipEventDestroy(c->llf_ev);
    if (c->batch_ev) (void)hipEventDestroy(c->batch_ev);
    c->batch_args.release();
    c->batch_wg3_args.release();
    c->mod_flag.release();
    if (c->mod_flag_host) (void)hipHostFree(c->mod_flag_host);
    c->mod_flag_host = nullptr;
    if (c->mod_ev) (void)hipEventDestroy(c->mod_ev);
    if (c->mod_join) (void)hipEventDestroy(c->mod_join);
    c->mod_ev = c->mod_join = nullptr;
    c->stage16.release();
    if (c->map16_ev) (void)hipEventDestroy(c->map16_ev);
    if (c->out_ev) (void)hipEventDestroy(c->out_ev);
    if (c->h_map16) (void)hipHostFree(c->h_map16);
    c->h_map16 = nullptr;
    c->wg3_items[0].release();
    c->wg3_items[1].release();
    c->batch_restore_args.release();
    for (int i = 0; i < jxl_ctx::kAux; i++) {
        if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
        if (c->join_ev[i]) (void)hipEventDestroy(c->join_ev[i]);
    }
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

jxl_status jxl_ctx_synchronize(jxl_ctx* c) {
    jxl_status st = bind(c);
    if (st) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JXL_OK;
}

void* jxl_ctx_stream(jxl_ctx* c) { return c ? (void*)c->stream : nullptr; }

jxl_status jxl_ctx_set_stream(jxl_ctx* c, void* stream) {
    jxl_status st = bind(c);
    if (st) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)stream;
    c->owns_stream = false;
    return JXL_OK;
}

// ---- VarDCT frame ---------------------------------------------------------------------------------
jxl_status jxl_vardct_begin_frame(jxl_ctx* c, const jxl_vardct_params* p) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!p) return fail(c, JXL_ERR_INVALID_ARGUMENT, "params is null");
    if (p->width <= 0 || p->height <= 0 || (p->width & 7) || (p->height & 7))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "padded frame size %dx%d must be positive multiples of 8", p->width, p->height);
    if (p->epf_iters < 0 || p->epf_iters > 3) return fail(c, JXL_ERR_INVALID_BITSTREAM, "epfIterations %d", p->epf_iters);
    if (p->out_format < 0 || p->out_format > JXL_OUT_RGB16 || p->transfer < 0 || p->transfer > JXL_TRANSFER_PQ_EXACT)
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad output stage selector");
    c->sub = false;
    for (int i = 0; i < 3; i++) {
        c->sy[i] = p->jpeg_upsampling_y[i];
        c->sx[i] = p->jpeg_upsampling_x[i];
        if (c->sy[i] < 0 || c->sy[i] > 1 || c->sx[i] < 0 || c->sx[i] > 1)
            return fail(c, JXL_ERR_INVALID_ARGUMENT, "jpeg upsampling shift of channel %d out of range", i);
        c->sub = c->sub || c->sy[i] || c->sx[i];
    }
    if (c->sub && ((p->width & 15) || (p->height & 15)))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "a chroma-subsampled frame is padded to multiples of 16 (Frame.getPaddedFrameSize)");
    c->p = *p;
    c->W = p->width; c->H = p->height;
    c->bw = c->W / 8; c->bh = c->H / 8;
    c->tw = ceil_div(c->bw, 8); c->th = ceil_div(c->bh, 8);
    const size_t npx = (size_t)c->W * c->H, nc = (size_t)c->bw * c->bh, nt = (size_t)c->tw * c->th;
    SectTimer tm("begin2");
    bool ok = true;
    for (int i = 0; i < 3; i++) {
        ok = ok && c->coeff[i].ensure(4 * npx) && c->planeA[i].ensure(4 * npx) && c->planeB[i].ensure(4 * npx) &&
             c->llf[i].ensure(4 * nc);  // (lf: a window of the table arena, finalize_tables)
        if (out_interleaved(p->out_format)) ok = ok && (i > 0 || c->outbuf[0].ensure(3 * (size_t)out_elem_size(p->out_format) * npx));
        else if (p->out_format != JXL_OUT_F32 || p->transfer != JXL_TRANSFER_NONE) ok = ok && c->outbuf[i].ensure(4 * npx);
    }
    ok = ok && c->inv_sigma.ensure(4 * nc);
    tm.mark("ensure");
    ok = ok && tab_begin_frame(c, nc, nt);
    if (!ok) return fail(c, JXL_ERR_OOM, "device allocation failed for a %dx%d frame", c->W, c->H);
    tm.mark("tab_begin_frame");
    c->coeff_zero_pending = true;  // new int[sY][sX] (HFCoefficients.java:68): zeroed when a group is put / the frame runs
    c->out_zero_pending = true;    // frame buffer starts zeroed (ImageBuffer ctor): zeroed at run time if a cell has no varblock
    c->h_sel.assign(nc, 255);
    c->h_xfy.assign(nt, 0);
    c->h_bfy.assign(nt, 0);
    const int nl = ceil_div(c->W, 2048) * ceil_div(c->H, 2048);
    c->lfg_blocks.assign(nl, {});
    c->lfg_set.assign(nl, 0);
    c->lf_jobs.clear();
    c->tables_dirty = true;
    c->frame_open = true;
    c->map16_valid = false;
    c->ev_runs = 0;
    c->result[0] = c->result[1] = c->result[2] = nullptr;
    tm.mark("host vectors");
    return JXL_OK;
}

jxl_status jxl_vardct_set_weights(jxl_ctx* c, const float* w, size_t n_floats, const int32_t* offs) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!w || !offs || n_floats == 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null weights");
    for (int pi = 0; pi < JXL_NUM_WEIGHT_SETS; pi++) {
        int mh = 0, mw = 0;
        for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++)
            if (JXL_TT[t].param_index == pi && !(JXL_TT[t].ph > JXL_TT[t].pw)) { mh = jxl_tt_mh(&JXL_TT[t]); mw = jxl_tt_mw(&JXL_TT[t]); break; }
        for (int ch = 0; ch < 3; ch++) {
            const int32_t o = offs[pi * 3 + ch];
            if (o < 0 || (size_t)o + (size_t)mh * mw > n_floats) return fail(c, JXL_ERR_INVALID_ARGUMENT, "weight offset %d out of range", o);
        }
    }
    // the same quant tables as the previous frame (the usual case inside one codestream; tables are ~1.5 MB): nothing to do
    if (c->have_weights && c->h_weights_in.size() == n_floats && memcmp(c->h_woffs_in, offs, sizeof c->h_woffs_in) == 0 &&
        memcmp(c->h_weights_in.data(), w, sizeof(float) * n_floats) == 0)
        return JXL_OK;
    c->h_weights_in.assign(w, w + n_floats);
    memcpy(c->h_woffs_in, offs, sizeof c->h_woffs_in);
    // device layout: every matrix at a 16-byte aligned offset of the library's own (the kernels move weight rows as float4),
    // plus a transposed copy of every matrix for flip() blocks, which index w3[x][y] (HFCoefficients.java:312-314)
    int32_t doffs[51];
    size_t total = 0;
    int mhs[JXL_NUM_WEIGHT_SETS], mws[JXL_NUM_WEIGHT_SETS];
    for (int pi = 0; pi < JXL_NUM_WEIGHT_SETS; pi++) {
        int mh = 0, mw = 0;
        for (int t = 0; t < JXL_NUM_TRANSFORM_TYPES; t++)
            if (JXL_TT[t].param_index == pi && !(JXL_TT[t].ph > JXL_TT[t].pw)) { mh = jxl_tt_mh(&JXL_TT[t]); mw = jxl_tt_mw(&JXL_TT[t]); break; }
        mhs[pi] = mh; mws[pi] = mw;
        for (int ch = 0; ch < 3; ch++) {
            doffs[pi * 3 + ch] = (int32_t)total;
            total += ((size_t)mh * mw + 3) & ~(size_t)3;
        }
    }
    std::vector<float> wd(total, 0.0f), wt(total, 0.0f);
    for (int pi = 0; pi < JXL_NUM_WEIGHT_SETS; pi++)
        for (int ch = 0; ch < 3; ch++) {
            const float* src = w + offs[pi * 3 + ch];
            float* dst = wd.data() + doffs[pi * 3 + ch];
            float* dstt = wt.data() + doffs[pi * 3 + ch];
            const int mh = mhs[pi], mw = mws[pi];
            memcpy(dst, src, sizeof(float) * (size_t)mh * mw);
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++) dstt[(size_t)x * mh + y] = src[(size_t)y * mw + x];
        }
    if (!c->weights.ensure(sizeof(float) * total) || !c->weights_t.ensure(sizeof(float) * total))
        return fail(c, JXL_ERR_OOM, "device allocation failed (weights)");
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // an earlier run may still be reading the old tables
    HIP_TRY(c, hipMemcpy(c->weights.p, wd.data(), sizeof(float) * total, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->weights_t.p, wt.data(), sizeof(float) * total, hipMemcpyHostToDevice));
    offs = doffs;
    memcpy(c->woffs, offs, sizeof c->woffs);
    c->have_weights = true;
    c->tables_dirty = true;  // woffs and the weight buffers are captured by fill_dev_frame: cached batch argument blocks
                             // (keyed on tables_gen) must not outlive them
    return JXL_OK;
}

jxl_status jxl_vardct_set_lfgroup(jxl_ctx* c, const jxl_lfgroup_desc* g) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if (!g || !g->dct_select || !g->hf_mul || !g->sharpness || !g->x_from_y || !g->b_from_y ||
        (g->n_blocks > 0 && !g->block_yx))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "null pointer in LF group descriptor");
    const int lrs = ceil_div(c->W, 2048), lcs = ceil_div(c->H, 2048);
    if (g->lfg_x < 0 || g->lfg_x >= lrs || g->lfg_y < 0 || g->lfg_y >= lcs) return fail(c, JXL_ERR_INVALID_ARGUMENT, "LF group position out of range");
    const int y0 = g->lfg_y * 256, x0 = g->lfg_x * 256;
    const int eh = std::min(256, c->bh - y0), ew = std::min(256, c->bw - x0);
    if (g->cells_h != eh || g->cells_w != ew)
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "LF group (%d,%d) must be %dx%d cells, got %dx%d", g->lfg_y, g->lfg_x, eh, ew, g->cells_h, g->cells_w);
    const int gth = ceil_div(eh, 8), gtw = ceil_div(ew, 8);
    tab_wait(c);  // prepare / run, then another LF group of the same frame: the queued transfer may still be reading the staging buffer
    for (int y = 0; y < eh; y++) {  // row copies into the frame-level grids (element-wise over four grids it was 1.1 ms per 4K frame)
        const size_t d = (size_t)(y0 + y) * c->bw + x0, s = (size_t)y * ew;
        memcpy(&c->h_hf_mul[d], g->hf_mul + s, sizeof(int32_t) * ew);
        memcpy(&c->h_sharp[d], g->sharpness + s, sizeof(int32_t) * ew);
        memcpy(&c->h_sel[d], g->dct_select + s, ew);
        if (!c->sub)
            for (int ch = 0; ch < 3; ch++)
                if (g->lf[ch]) memcpy(&c->h_lf[ch][d], g->lf[ch] + s, sizeof(float) * ew);
    }
    if (c->sub)  // lf[ch] is (cells_h >> sy) x (cells_w >> sx), placed on the channel's own cell grid (LFCoefficients.java:38-44)
        for (int ch = 0; ch < 3; ch++) {
            if (!g->lf[ch]) continue;
            const int h2 = eh >> c->sy[ch], w2 = ew >> c->sx[ch], bw2 = c->bw >> c->sx[ch];
            for (int y = 0; y < h2; y++)
                for (int x = 0; x < w2; x++)
                    c->h_lf[ch][(size_t)((y0 >> c->sy[ch]) + y) * bw2 + (x0 >> c->sx[ch]) + x] = g->lf[ch][(size_t)y * w2 + x];
        }
    for (int y = 0; y < gth; y++)
        for (int x = 0; x < gtw; x++) {
            const size_t d = (size_t)(g->lfg_y * 32 + y) * c->tw + g->lfg_x * 32 + x;
            c->h_xfy[d] = g->x_from_y[y * gtw + x];
            c->h_bfy[d] = g->b_from_y[y * gtw + x];
        }
    std::vector<DevBlock>& bl = c->lfg_blocks[g->lfg_y * lrs + g->lfg_x];
    bl.clear();
    bl.reserve(g->n_blocks);
    for (int i = 0; i < g->n_blocks; i++) {
        const int by = g->block_yx[2 * i], bx = g->block_yx[2 * i + 1];
        if (by < 0 || bx < 0 || by >= eh || bx >= ew) return fail(c, JXL_ERR_INVALID_BITSTREAM, "block %d at (%d,%d) outside its LF group", i, by, bx);
        const int t = g->dct_select[(size_t)by * ew + bx];
        if (t > 26) return fail(c, JXL_ERR_INVALID_BITSTREAM, "Invalid Transform Type: %d", t);  // HFMetadata.java:46-47
        bl.push_back(DevBlock{(uint16_t)(y0 + by), (uint16_t)(x0 + bx), (uint32_t)t, 0u, 1});
    }
    c->lfg_set[g->lfg_y * lrs + g->lfg_x] = 1;
    c->tables_dirty = true;
    return JXL_OK;
}

static jxl_status check_lfquant(jxl_ctx* c, const jxl_lfquant_desc* d) {
    if (!d || !d->lf_quant[0] || !d->lf_quant[1] || !d->lf_quant[2]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null LF image");
    if (d->cells_h <= 0 || d->cells_w <= 0 || d->cells_h > 256 || d->cells_w > 256) return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad LF group size");
    if (d->extra_precision < 0 || d->extra_precision > 3) return fail(c, JXL_ERR_INVALID_BITSTREAM, "extraPrecision %d", d->extra_precision);
    return JXL_OK;
}

jxl_status jxl_vardct_set_lfgroup_lfquant(jxl_ctx* c, const jxl_lfquant_desc* d) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if ((st = check_lfquant(c, d))) return st;
    const int lrs = ceil_div(c->W, 2048), lcs = ceil_div(c->H, 2048);
    if (d->lfg_x < 0 || d->lfg_x >= lrs || d->lfg_y < 0 || d->lfg_y >= lcs) return fail(c, JXL_ERR_INVALID_ARGUMENT, "LF group position out of range");
    const int eh = std::min(256, c->bh - d->lfg_y * 256), ew = std::min(256, c->bw - d->lfg_x * 256);
    if (d->cells_h != eh || d->cells_w != ew) return fail(c, JXL_ERR_INVALID_ARGUMENT, "LF group (%d,%d) must be %dx%d cells", d->lfg_y, d->lfg_x, eh, ew);
    if (c->sub && d->adaptive_smoothing)
        return fail(c, JXL_ERR_INVALID_BITSTREAM, "Adaptive Smoothing is incompatible with subsampling");  // LFCoefficients.java:36-37
    jxl_ctx::LfJob job;
    job.d = *d;
    for (int ch = 0; ch < 3; ch++) {
        const size_t n = (size_t)(eh >> c->sy[ch]) * (ew >> c->sx[ch]);
        job.q[ch].assign(d->lf_quant[ch], d->lf_quant[ch] + n);
        job.d.lf_quant[ch] = nullptr;
    }
    for (auto& j : c->lf_jobs)
        if (j.d.lfg_x == d->lfg_x && j.d.lfg_y == d->lfg_y) {
            j = std::move(job);
            c->tables_dirty = true;
            return JXL_OK;
        }
    c->lf_jobs.push_back(std::move(job));
    c->tables_dirty = true;
    return JXL_OK;
}

jxl_status jxl_stage_lf_dequant(jxl_ctx* c, const jxl_lfquant_desc* d, float base_corr_x, float base_corr_b, int32_t color_factor,
                                float* const out[3]) {
    jxl_status st = bind(c);
    if (st) return st;
    if ((st = check_lfquant(c, d))) return st;
    if (!out || !out[0] || !out[1] || !out[2] || color_factor == 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "lf_dequant: bad arguments");
    Tmp t;
    const size_t n = (size_t)d->cells_h * d->cells_w;
    const int32_t* dq[3];
    float* dout[3];
    for (int ch = 0; ch < 3; ch++) {
        dq[ch] = t.up(d->lf_quant[ch], n);
        dout[ch] = t.up<float>(nullptr, n);
        if (!dq[ch] || !dout[ch]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_lf_dequant(dq, dout, d->cells_h, d->cells_w, 0, d->cells_w, d->scaled_dequant, d->extra_precision, base_corr_x, base_corr_b,
                      color_factor, d->x_factor_lf, d->b_factor_lf, d->adaptive_smoothing, c->stream);
    if ((st = finish(c))) return st;
    for (int ch = 0; ch < 3; ch++) HIP_TRY(c, hipMemcpy(out[ch], dout[ch], 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

// the deferred zero-fills of begin_frame
static jxl_status zero_coeff_planes(jxl_ctx* c) {
    if (!c->coeff_zero_pending) return JXL_OK;
    const size_t npx = (size_t)c->W * c->H;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemsetAsync(c->coeff[i].p, 0, 4 * npx, c->stream));
    c->coeff_zero_pending = false;
    return JXL_OK;
}
static jxl_status zero_output_planes(jxl_ctx* c) {
    if (!c->out_zero_pending) return JXL_OK;
    const size_t npx = (size_t)c->W * c->H;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemsetAsync(c->planeA[i].p, 0, 4 * npx, c->stream));
    c->out_zero_pending = false;
    return JXL_OK;
}

// before the transforms of a frame run (finalize_tables has run): whatever of begin_frame's zero-fills is still owed
static jxl_status pre_run_zero(jxl_ctx* c) {
    jxl_status st = zero_coeff_planes(c);
    if (st) return st;
    if (c->out_zero_pending && c->blocks_cover && (c->p.stages & JXL_STAGE_IDCT)) c->out_zero_pending = false;  // every sample is written
    return zero_output_planes(c);
}

// device-visible address of page-locked host memory (null: pageable)
static const void* pinned_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc pointer: not an error of ours
        return nullptr;
    }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

// jxl_vardct_put_group / _i16. The group's three rectangles go to the device in ONE launch that reads host memory itself
// (k_put_group): page-locked, 16-byte aligned caller buffers are read in place (the caller keeps them until the stream has
// passed this point -- jxl_ctx_synchronize or a finished frame -- as with the queued copies of r2-r3); anything else is copied
// by the host into a small page-locked ring first, after which the caller's buffer is free at once and nothing waits
// until the ring wraps onto a slot whose launch has not finished.
extern "C++" {
template <typename T>
static jxl_status put_group_t(jxl_ctx* c, int32_t pass, int32_t group, const T* const q[3], const int32_t stride[3]) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    const int grs = ceil_div(c->W, 256), gcs = ceil_div(c->H, 256);
    if (group < 0 || group >= grs * gcs || pass < 0 || !q || !stride) return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad group/pass");
    const int gy = group / grs, gx = group % grs;  // Frame.getGroupLocation (Frame.java:883)
    const int gh = std::min(256, c->H - gy * 256), gw = std::min(256, c->W - gx * 256);  // getGroupSize (:905)
    PutGroupArgs a;
    a.acc = pass > 0 ? 1 : 0;
    bool in_place = true;
    for (int ch = 0; ch < 3; ch++) {
        // channel geometry (HFCoefficients.java:64-69, PassGroup.java:223-226): all shifts are zero for ordinary frames
        a.gw[ch] = gw >> c->sx[ch];
        a.gh[ch] = gh >> c->sy[ch];
        a.W[ch] = c->W >> c->sx[ch];
        a.y0[ch] = (gy * 256) >> c->sy[ch];
        a.x0[ch] = (gx * 256) >> c->sx[ch];
        a.plane[ch] = c->coeff[ch].as<int32_t>();
        if (!q[ch] || stride[ch] < a.gw[ch]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad plane %d", ch);
        a.sstride[ch] = stride[ch];
        a.src[ch] = nullptr;
        if (in_place && (((uintptr_t)q[ch] | ((uintptr_t)stride[ch] * sizeof(T))) & 15) == 0) a.src[ch] = pinned_device_ptr(q[ch]);
        in_place = in_place && a.src[ch];
    }
    if ((st = zero_coeff_planes(c))) return st;
    if (!in_place) {
        constexpr size_t kSlot = 3 * (size_t)256 * 256 * sizeof(int32_t);
        if (!c->h_grp) {
            // all or nothing: the ring is published (h_grp set) only when its device address and every event exist, so a
            // failure here can never leave a later call with a null device address or null events
            void *hp = nullptr, *dp = nullptr;
            hipEvent_t ev[jxl_ctx::kGrpSlots] = {};
            bool ok = hipHostMalloc(&hp, kSlot * jxl_ctx::kGrpSlots, hipHostMallocDefault) == hipSuccess && hp;
            const bool have_mem = ok;
            ok = ok && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp;
            for (int i = 0; i < jxl_ctx::kGrpSlots && ok; i++) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                for (int i = 0; i < jxl_ctx::kGrpSlots; i++)
                    if (ev[i]) (void)hipEventDestroy(ev[i]);
                if (hp && have_mem) (void)hipHostFree(hp);
                return fail(c, have_mem ? JXL_ERR_DEVICE : JXL_ERR_OOM, "set-up of the page-locked group staging ring failed");
            }
            for (int i = 0; i < jxl_ctx::kGrpSlots; i++) c->grp_ev[i] = ev[i];
            c->h_grp_dev = dp;
            c->h_grp = hp;
        }
        const int slot = c->grp_slot;
        c->grp_slot = (slot + 1) % jxl_ctx::kGrpSlots;
        if (c->grp_inflight[slot]) HIP_TRY(c, hipEventSynchronize(c->grp_ev[slot]));
        c->grp_inflight[slot] = false;
        for (int ch = 0; ch < 3; ch++) {
            const size_t o = kSlot * slot + (size_t)ch * 256 * 256 * sizeof(int32_t);
            T* d = reinterpret_cast<T*>(static_cast<char*>(c->h_grp) + o);
            for (int y = 0; y < a.gh[ch]; y++) memcpy(d + (size_t)y * a.gw[ch], q[ch] + (size_t)y * stride[ch], sizeof(T) * (size_t)a.gw[ch]);
            a.src[ch] = static_cast<const char*>(c->h_grp_dev) + o;
            a.sstride[ch] = a.gw[ch];
        }
        hipLaunchKernelGGL(k_put_group<T>, dim3(ceil_div(gw / 8, 32), gh / 8, 3), dim3(256), 0, c->stream, a);
        HIP_TRY(c, hipEventRecord(c->grp_ev[slot], c->stream));
        c->grp_inflight[slot] = true;
    } else {
        hipLaunchKernelGGL(k_put_group<T>, dim3(ceil_div(gw / 8, 32), gh / 8, 3), dim3(256), 0, c->stream, a);
    }
    HIP_TRY(c, hipGetLastError());
    return JXL_OK;
}
}  // extern "C++"

jxl_status jxl_vardct_put_group(jxl_ctx* c, int32_t pass, int32_t group, const int32_t* const q[3], const int32_t stride[3]) {
    return put_group_t<int32_t>(c, pass, group, q, stride);
}

jxl_status jxl_vardct_put_group_i16(jxl_ctx* c, int32_t pass, int32_t group, const int16_t* const q[3], const int32_t stride[3]) {
    return put_group_t<int16_t>(c, pass, group, q, stride);
}

jxl_status jxl_vardct_map_coeffs_i16_ex(jxl_ctx* c, int16_t* planes[3], int32_t strides[3], int32_t flags) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if (!planes || !strides) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null argument");
    if (flags & ~JXL_MAP_NO_FILL) return fail(c, JXL_ERR_INVALID_ARGUMENT, "map: unknown flags");
    size_t off[4] = {0, 0, 0, 0};
    for (int ch = 0; ch < 3; ch++) off[ch + 1] = off[ch] + (((size_t)(c->W >> c->sx[ch]) * (c->H >> c->sy[ch]) * sizeof(int16_t) + 255) & ~(size_t)255);
    // the buffer may be written again once the last commit's transfers have READ it (r4: an event, not the whole stream -- the
    // previous frame's kernels and its output copy no longer hold the host back)
    if (c->map16_inflight && c->map16_ev) HIP_TRY(c, hipEventSynchronize(c->map16_ev));
    c->map16_inflight = false;
    if (!c->is_feeder && c->device >= 0 && c->device < 64) {
        c->is_feeder = true;
        g_feeders[c->device].fetch_add(1, std::memory_order_relaxed);
    }
    if (c->h_map16_bytes < off[3]) {
        if (c->h_map16) (void)hipHostFree(c->h_map16);
        c->h_map16 = nullptr;
        c->h_map16_bytes = 0;
        if (hipHostMalloc(&c->h_map16, off[3], hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_map16 = nullptr;
            return fail(c, JXL_ERR_OOM, "page-locked allocation of %zu bytes failed", off[3]);
        }
        c->h_map16_bytes = off[3];
    }
    if (!(flags & JXL_MAP_NO_FILL)) memset(c->h_map16, 0, off[3]);
    c->map16_nofill = (flags & JXL_MAP_NO_FILL) != 0;
    for (int ch = 0; ch < 3; ch++) {
        planes[ch] = reinterpret_cast<int16_t*>(static_cast<char*>(c->h_map16) + off[ch]);
        strides[ch] = c->W >> c->sx[ch];
    }
    c->map16_valid = true;
    return JXL_OK;
}

jxl_status jxl_vardct_map_coeffs_i16(jxl_ctx* c, int16_t* planes[3], int32_t strides[3]) {
    return jxl_vardct_map_coeffs_i16_ex(c, planes, strides, 0);
}

jxl_status jxl_vardct_coeff_plane_rows(jxl_ctx* c, int32_t rows[3]) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if (!rows) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null argument");
    for (int ch = 0; ch < 3; ch++) rows[ch] = c->H >> c->sy[ch];
    return JXL_OK;
}

jxl_status jxl_vardct_geometry(jxl_ctx* c, int32_t info[13]) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if (!info) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null argument");
    for (int ch = 0; ch < 3; ch++) {
        info[ch] = c->W >> c->sx[ch];
        info[3 + ch] = c->H >> c->sy[ch];
        info[6 + 2 * ch] = c->sx[ch];
        info[7 + 2 * ch] = c->sy[ch];
    }
    info[12] = jxl_vardct_out_elem_size(c);
    return JXL_OK;
}

jxl_status jxl_vardct_output_geometry(jxl_ctx* c, int32_t info[5]) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    if (!info) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null argument");
    // what run_frame will decide (do_out) -- the same expression, so that the numbers hold before and after jxl_vardct_run
    const jxl_vardct_params& p = c->p;
    const bool do_out = (p.stages & JXL_STAGE_OUT) && (p.transfer != JXL_TRANSFER_NONE || p.out_format != JXL_OUT_F32);
    const bool il = do_out && out_interleaved(p.out_format);
    info[0] = c->W;
    info[1] = c->H;
    info[2] = do_out ? out_elem_size(p.out_format) : 4;
    info[3] = il ? 1 : 0;
    info[4] = il ? 1 : 3;
    return JXL_OK;
}

jxl_status jxl_vardct_group_size(jxl_ctx* c, int32_t group, int32_t gw[3], int32_t gh[3]) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    const int grs = ceil_div(c->W, 256), gcs = ceil_div(c->H, 256);
    if (!gw || !gh || group < 0 || group >= grs * gcs) return fail(c, JXL_ERR_INVALID_ARGUMENT, "group index out of range");
    const int gy = group / grs, gx = group % grs;
    const int w = std::min(256, c->W - gx * 256), h = std::min(256, c->H - gy * 256);
    for (int ch = 0; ch < 3; ch++) {
        gw[ch] = w >> c->sx[ch];
        gh[ch] = h >> c->sy[ch];
    }
    return JXL_OK;
}

static jxl_status commit_i16(jxl_ctx* c, const uint8_t* written, int32_t n_groups) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open || !c->map16_valid) return fail(c, JXL_ERR_STATE, "map_coeffs_i16 first");
    const int grs = ceil_div(c->W, 256), gcs = ceil_div(c->H, 256);
    if (written && n_groups != grs * gcs) return fail(c, JXL_ERR_INVALID_ARGUMENT, "commit: %d group flags for a frame of %d groups", n_groups, grs * gcs);
    if (!written && c->map16_nofill) return fail(c, JXL_ERR_STATE, "planes mapped with JXL_MAP_NO_FILL: commit with the list of written groups");
    size_t off = 0;
    static const bool zero_copy = !(getenv("JXL_COMMIT_ZEROCOPY") && atoi(getenv("JXL_COMMIT_ZEROCOPY")) == 0);
    void* hdev = nullptr;
    bool all8 = true;
    // (the zero-copy widening kernel walks whole 8x8 cells: plane width AND height -- a subsampled plane of a frame whose padded height
    // is 8 has 4 rows -- must be multiples of 8 and not empty; otherwise the staged path, which handles partial cells)
    for (int ch = 0; ch < 3; ch++) all8 = all8 && (((c->W >> c->sx[ch]) & 7) == 0) && (((c->H >> c->sy[ch]) & 7) == 0) && (c->H >> c->sy[ch]) >= 8;
    const bool zc = zero_copy && all8 && hipHostGetDevicePointer(&hdev, c->h_map16, 0) == hipSuccess && hdev;
    if (!zc) (void)hipGetLastError();
    for (int ch = 0; ch < 3; ch++) {
        const int Wc = c->W >> c->sx[ch], Hc = c->H >> c->sy[ch];
        const size_t bytes = (size_t)Wc * Hc * sizeof(int16_t);
        if (written && c->map16_nofill) {
            // groups the caller did not write read as zero (HFCoefficients.java:68: a fresh int[][]): zero-fill THEIR rectangles
            // only -- a decoder writes every group of a frame, so this is normally nothing (the unconditional zero-fill of map
            // was 50 MB of host stores per 4K frame, 1.0-1.4 ms)
            int16_t* pl = reinterpret_cast<int16_t*>(static_cast<char*>(c->h_map16) + off);
            const int gh = 256 >> c->sy[ch], gw = 256 >> c->sx[ch];
            for (int g = 0; g < n_groups; g++) {
                if (written[g]) continue;
                const int y0 = (g / grs) * gh, x0 = (g % grs) * gw;
                const int y1 = std::min(Hc, y0 + gh), x1 = std::min(Wc, x0 + gw);
                for (int y = y0; y < y1; y++) memset(pl + (size_t)y * Wc + x0, 0, sizeof(int16_t) * (size_t)(x1 - x0));
            }
        }
        // r4: the widening kernel reads the page-locked planes over PCIe itself -- no SDMA transfer, no device staging copy, one
        // runtime call per plane instead of two. With a dozen contexts committing from a dozen threads the hipMemcpyAsync calls
        // had become the slowest part of a frame (commit 2-3 ms per frame and thread against 0.06; tools/archive/r4_zerocopy_ab.sh).
        // JXL_COMMIT_ZEROCOPY=0: the staged form
        if (zc) {
            static const int wgrid_env = getenv("JXL_WIDEN_GRID") ? std::max(1, atoi(getenv("JXL_WIDEN_GRID"))) : 0;
            const int wgrid = wgrid_env ? wgrid_env : bus_grid(c, false);
            const int gx = ceil_div(Wc / 8, 64), tiles = gx * (Hc / 8);
            hipLaunchKernelGGL(k_widen2d_host8, dim3(std::min(wgrid, tiles)), dim3(256), 0, c->stream, c->coeff[ch].as<int32_t>(),
                               reinterpret_cast<const int16_t*>(static_cast<char*>(hdev) + off), Wc, Hc, gx, tiles);
            off += (bytes + 255) & ~(size_t)255;
            continue;
        }
        if (!c->stage16.ensure(c->h_map16_bytes)) return fail(c, JXL_ERR_OOM, "device allocation failed (int16 staging)");
        int16_t* stg = reinterpret_cast<int16_t*>(static_cast<char*>(c->stage16.p) + off);
        HIP_TRY(c, hipMemcpyAsync(stg, static_cast<char*>(c->h_map16) + off, bytes, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_widen2d, dim3(ceil_div(Wc, 64), ceil_div(Hc, 4)), dim3(256), 0, c->stream, c->coeff[ch].as<int32_t>(), Wc, 0, 0, stg, Wc,
                           Hc, 0);
        off += (bytes + 255) & ~(size_t)255;
    }
    c->coeff_zero_pending = false;  // every sample of the three planes has just been written
    if (!c->map16_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->map16_ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->map16_ev, c->stream));
    c->map16_inflight = true;
    return JXL_OK;
}

jxl_status jxl_vardct_commit_coeffs_i16(jxl_ctx* c) { return commit_i16(c, nullptr, 0); }

jxl_status jxl_vardct_commit_coeffs_i16_groups(jxl_ctx* c, const uint8_t* group_written, int32_t n_groups) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    if (!group_written) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null group flags");
    return commit_i16(c, group_written, n_groups);
}

void* jxl_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void jxl_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

jxl_status jxl_vardct_enable_stage_timing(jxl_ctx* c, int32_t on) {
    if (!c) return JXL_ERR_INVALID_ARGUMENT;
    c->timing = on != 0;
    c->ev_runs = 0;
    return JXL_OK;
}

}  // extern "C"

namespace {
// the frame pipeline; idct_done: the IDCT stage of this frame has already been enqueued (batched launch); collect: the
// fused restoration launch is not enqueued here, its argument block is handed back instead (*collected = true if it was)
jxl_status run_frame(jxl_ctx* c, bool idct_done, FusedArgs* collect = nullptr, bool* collected = nullptr) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    st = finalize_tables(c);
    if (st) return st;
    if ((st = pre_run_zero(c))) return st;
    const jxl_vardct_params& p = c->p;
    hipStream_t s = c->stream;
    int launches = 0;
    float* A[3] = {c->planeA[0].as<float>(), c->planeA[1].as<float>(), c->planeA[2].as<float>()};
    float* B[3] = {c->planeB[0].as<float>(), c->planeB[1].as<float>(), c->planeB[2].as<float>()};
    hipEvent_t* evs = c->ev[c->ev_runs % jxl_ctx::kEvSlots];
    if (c->timing) (void)hipEventRecord(evs[0], s);
    if ((p.stages & JXL_STAGE_IDCT) && !idct_done) {
        DevFrame f;
        fill_dev_frame(c, f);
        const DevBlock* blocks = c->blocks.as<DevBlock>();
        const WorkItem* items = c->items.as<WorkItem>();
        if (c->llf_count > 0) {
            float* L[3] = {c->llf[0].as<float>(), c->llf[1].as<float>(), c->llf[2].as<float>()};
            launch_llf(f, blocks, c->llf_first, c->llf_count, L, s);
            launches++;
        }
        // chroma-subsampled frames: the launch of channel ch sees that channel's geometry
        auto frame_of = [&](int channel) {
            DevFrame fc = f;
            if (channel >= 0) {
                fc.no_cfl = 1;
                fc.width = c->W >> c->sx[channel];
                fc.height = c->H >> c->sy[channel];
                fc.bw = c->bw >> c->sx[channel];
                fc.bh = c->bh >> c->sy[channel];
                fc.hf_mul = c->hfm_sub[channel].as<int32_t>();
            }
            return fc;
        };
        // classes 2 / 3 (k_idct_wg3.hip): argument blocks of the two persistent launches and of the LLF launch in front
        // 512 = two 256-thread workgroups per CU (r3; was 768): at 128 VGPRs three of them leave one wave slot per SIMD, and a
        // workgroup of the 64-point launch (512 threads: two waves per SIMD) then fits on no CU until a persistent workgroup of
        // this launch retires -- at its end. With two per CU all 110 workgroups of the 64-point launch are resident at once
        // beside it: single 4K frame 231 -> 214 us, IDCT stage 134 -> 117 us, batch unchanged (49.0 / 48.9 Gpx/s, same box)
        static const int wg3_grid = wg3_grid_cap(false);
        static const int wg3_grid_big = wg3_grid_cap(true);
        Wg3Args wa[2], wl;
        int wn[2] = {0, 0};
        bool any_llf = false;
        {
            std::vector<IdctSegment> all;
            for (const auto& tl : c->type_launches)
                if (tl.cls >= 2) {
                    wn[tl.cls - 2] = build_wg3_args(f, blocks, tl.segs.data(), (int)tl.segs.size(), tl.cls - 2, A, wa[tl.cls - 2]);
                    if (wn[tl.cls - 2] < 0) return fail(c, JXL_ERR_STATE, "IDCT launch: too many segments");
                    // (r6: the list may hold holes -- wg3_item_table --: at least one record per item, and the list's length is what the walk is bounded by)
                    if (c->wg3_item_count[tl.cls - 2] < wn[tl.cls - 2]) return fail(c, JXL_ERR_STATE, "IDCT launch: item list out of date");
                    if (wn[tl.cls - 2] > 0) wa[tl.cls - 2].total_items = wn[tl.cls - 2] = c->wg3_item_count[tl.cls - 2];
                    wa[tl.cls - 2].items = c->wg3_items[tl.cls - 2].as<int>();
                    all.insert(all.end(), tl.segs.begin(), tl.segs.end());
                }
            if (!all.empty()) {
                if (build_wg3_args(f, blocks, all.data(), (int)all.size(), 2, A, wl) < 0) return fail(c, JXL_ERR_STATE, "IDCT launch: too many segments");
                for (int q = 0; q < wl.n_seg; q++) any_llf = any_llf || (wl.seg[q].type != 0 && !wl.llf_in_item);
            }
        }
        const int n_k = (int)c->type_launches.size() + (int)c->special_launches.size();
        const bool fork = n_k > 1 && c->n_aux > 0;
        if (wn[0] > 0 || wn[1] > 0) {
            // Frame without chroma subsampling. The long pole is the persistent launch of everything up to 32 points (with the
            // 8x8 DCTs: ~80 % of the pixels): it goes on the main stream right behind the LLF launch it depends on. The special
            // 8x8 transforms need no LLF and start at once on the side stream; the 64-point family follows them there, behind an
            // event that says the LLF planes are written. (Measured on the 4K default mix: LLF as two launches in front of
            // everything, the special kernel queued behind the 64-point launch: 112 us; this order: see profiles/.)
            hipStream_t side = fork ? c->aux[0] : s;
            if (fork) {
                (void)hipEventRecord(c->fork_ev, s);
                (void)hipStreamWaitEvent(side, c->fork_ev, 0);
            }
            static const int plan = getenv("JXL_WG3_PLAN") ? atoi(getenv("JXL_WG3_PLAN")) : 0;
            // With finalizeLLF inside the items no IDCT launch depends on another. The 64-point launch is ONE round of items (a
            // 4K frame of the default mix has ~100 of them for 110 workgroups), i.e. its duration is the latency of a single
            // item -- 28 us alone on the device, 58-72 us beside the other class's workgroups -- so it goes first on the side
            // stream and the 8x8 special kernel (24 us) behind it, not in front of it
            static const bool big_first = !(getenv("JXL_WG3_BIG_FIRST") && atoi(getenv("JXL_WG3_BIG_FIRST")) == 0);
            // experiment (JXL_WG3_BIG_AFTER=1): the 64-point launch on the MAIN stream behind the <= 32-point launch, so that no CU
            // ever holds two workgroups of each (4 x 128 registers per lane: no room for a restoration workgroup of another frame)
            static const bool big_after = getenv("JXL_WG3_BIG_AFTER") && atoi(getenv("JXL_WG3_BIG_AFTER")) != 0;
            const bool big_early = big_first && !big_after && plan != 1 && !any_llf && wn[1] > 0;
            if (big_early) {
                launch_idct_wg3(wa[1], true, wg3_grid_big, side);
                launches++;
            }
            for (const auto& sl : c->special_launches) {
                launch_idct_special(frame_of(sl.channel), blocks, items + sl.items_off, sl.n_items, A, side, sl.wg_items);
                launches++;
            }
            if (any_llf) {
                float* L[3] = {c->llf[0].as<float>(), c->llf[1].as<float>(), c->llf[2].as<float>()};
                launch_llf_wg3(wl, L, s);
                launches++;
                if (fork && wn[1] > 0) {
                    (void)hipEventRecord(c->llf_ev, s);  // "LLF planes written"
                    (void)hipStreamWaitEvent(side, c->llf_ev, 0);
                }
            }
            hipStream_t s_small = plan == 1 ? side : s, s_big = plan == 1 ? s : side;
            if (plan == 1 && fork && any_llf && wn[0] > 0 && wn[1] <= 0) {
                (void)hipEventRecord(c->llf_ev, s);
                (void)hipStreamWaitEvent(side, c->llf_ev, 0);
            }
            if (plan == 1 && wn[1] > 0) {
                launch_idct_wg3(wa[1], true, wg3_grid_big, s_big);
                launches++;
            }
            if (wn[0] > 0) {
                launch_idct_wg3(wa[0], false, wg3_grid, s_small);
                launches++;
            }
            if (plan != 1 && wn[1] > 0 && !big_early) {
                launch_idct_wg3(wa[1], true, wg3_grid_big, big_after ? s : s_big);
                launches++;
            }
            for (const auto& tl : c->type_launches)  // nothing today: every class 0 / 1 type of such a frame is handled above
                if (tl.cls < 2) {
                    launch_idct_multi(frame_of(tl.channel), blocks, tl.cls, tl.segs.data(), (int)tl.segs.size(), 3, 0, A, s);
                    launches++;
                }
            if (fork) {
                (void)hipEventRecord(c->join_ev[0], side);
                (void)hipStreamWaitEvent(s, c->join_ev[0], 0);
            }
        } else {
            // fork: every type kernel writes a disjoint set of varblocks
            int used = 0;
            if (fork) {
                (void)hipEventRecord(c->fork_ev, s);
                used = std::min(n_k - 1, c->n_aux);
                for (int i = 0; i < used; i++) (void)hipStreamWaitEvent(c->aux[i], c->fork_ev, 0);
            }
            int k = 0;
            auto pick = [&]() { const int i = k++; return (!fork || i % (used + 1) == 0) ? s : c->aux[i % (used + 1) - 1]; };
            for (const auto& tl : c->type_launches) {
                launch_idct_multi(frame_of(tl.channel), blocks, tl.cls, tl.segs.data(), (int)tl.segs.size(), tl.channel < 0 ? 3 : 1, tl.channel < 0 ? 0 : tl.channel, A, pick());
                launches++;
            }
            for (const auto& sl : c->special_launches) {
                launch_idct_special(frame_of(sl.channel), blocks, items + sl.items_off, sl.n_items, A, pick(), sl.wg_items);
                launches++;
            }
            if (fork)
                for (int i = 0; i < used; i++) {
                    (void)hipEventRecord(c->join_ev[i], c->aux[i]);
                    (void)hipStreamWaitEvent(s, c->join_ev[i], 0);
                }
        }
        if (c->large_count > 0) launch_idct_large(f, blocks, c->h_blocks.data(), c->large_first, c->large_count, A, B, s, &launches);
    }
    // Frame.invertSubsampling (Frame.java:457, 681-723): horizontal doublings, then vertical ones, per channel
    float* curp[3] = {A[0], A[1], A[2]};
    float* othp[3] = {B[0], B[1], B[2]};
    if (c->sub && (p.stages & JXL_STAGE_IDCT))
        for (int ch = 0; ch < 3; ch++) {
            int h2 = c->H >> c->sy[ch], w2 = c->W >> c->sx[ch];
            for (int i = 0; i < c->sx[ch]; i++) {
                launch_chroma_upsample_h(curp[ch], h2, w2, othp[ch], s);
                w2 *= 2;
                std::swap(curp[ch], othp[ch]);
                launches++;
            }
            for (int i = 0; i < c->sy[ch]; i++) {
                launch_chroma_upsample_v(curp[ch], h2, w2, othp[ch], s);
                h2 *= 2;
                std::swap(curp[ch], othp[ch]);
                launches++;
            }
        }
    if (c->timing) (void)hipEventRecord(evs[1], s);
    float** cur = curp;
    float** oth = othp;
    const bool do_gab = (p.stages & JXL_STAGE_GAB) && p.gab;
    const bool do_epf = (p.stages & JXL_STAGE_EPF) && p.epf_iters > 0;
    const bool do_xyb = (p.stages & JXL_STAGE_XYB) && p.xyb;
    const bool do_out = (p.stages & JXL_STAGE_OUT) && (p.transfer != JXL_TRANSFER_NONE || p.out_format != JXL_OUT_F32);
    // sharpness range check of Frame.java:565-566 (host side: the maps came through the host; scanned by finalize_tables)
    if (do_epf && c->sharp_is_bad) return fail(c, JXL_ERR_INVALID_BITSTREAM, "Invalid EPF Sharpness: %d", c->sharp_bad);
    bool fused = false;
    if (do_gab || do_epf || do_xyb || do_out) {
        RestoreParams rp{};
        rp.gab = do_gab; rp.epf_iters = do_epf ? p.epf_iters : 0; rp.xyb = do_xyb;
        rp.transfer = !do_out ? JXL_TRANSFER_NONE : p.transfer == JXL_TRANSFER_PQ_EXACT ? JXL_TRANSFER_PQ : p.transfer;
        rp.max_value = do_out ? out_max_value(p.out_format) : 0;
        rp.out_elem = do_out ? out_elem_size(p.out_format) : 4;
        rp.interleaved = do_out && out_interleaved(p.out_format);
        for (int i = 0; i < 3; i++) {
            const float mult = 1.0f / (1.0f + 4.0f * (p.gab_w1[i] + p.gab_w2[i]));  // Frame.java:510-517
            rp.gab_base[i] = mult; rp.gab_adj[i] = p.gab_w1[i] * mult; rp.gab_diag[i] = p.gab_w2[i] * mult;
            rp.epf[i] = make_epf(p.epf_channel_scale, p.epf_pass0_sigma_scale, p.epf_pass2_sigma_scale, p.epf_border_sad_mul, i, 0.0f);
        }
        rp.xybp = make_xyb(p.opsin_matrix, p.opsin_bias, p.cbrt_opsin_bias, p.intensity_target);
        rp.global_scale_f = p.global_scale_f;
        rp.pq_tab = p.transfer == JXL_TRANSFER_PQ_EXACT ? nullptr : c->pq_tab.as<float>();  // _EXACT: the double-precision form
        rp.srgb8_tab = c->srgb8_tab.as<float>();
        rp.pq16_thr = do_out ? pq16_thresholds_for(c, p.transfer, out_max_value(p.out_format)) : nullptr;
        rp.srgb16_tab = do_out ? srgb16_table_for(c, p.transfer, out_max_value(p.out_format)) : nullptr;
        memcpy(rp.sharp_lut, p.epf_sharp_lut, sizeof rp.sharp_lut);
        void* dst[3];
        for (int i = 0; i < 3; i++) dst[i] = do_out ? c->outbuf[i].p : (void*)oth[i];
        const float* src[3] = {cur[0], cur[1], cur[2]};
        // r5: three EPF iterations as TWO launches -- Gaborish + the 13-tap iteration on a 64x32 tile into the other plane set, then
        // the two-iteration kernel (no Gaborish) from there. Fused in one launch, the 13-tap iteration runs on the whole 64x32
        // window of a 58x26 output tile (1.36 x its pixels) and its 118 registers hold ALL stages of the kernel to 2 workgroups
        // per CU; split, it runs once per pixel, and the second launch is the 63-register kernel at 4 workgroups per CU. The
        // planes cross HBM once more (200 MB per 4K frame, under the launches' arithmetic). JXL_EPF3_SPLIT=0: one launch.
        const bool split = epf3_split_on() && rp.epf_iters == 3 && c->W >= 8 && c->H >= 8;
        if (collect && split) {
            if (collected) *collected = false;  // (jxl_vardct_run_batch: this frame launches its own pair)
            return JXL_OK;
        }
        if (collect) {
            fused = fill_restore_fused_args(src, dst, c->H, c->W, c->hf_mul.as<int32_t>(), c->sharp.as<int32_t>(), rp, *collect);
            if (collected) *collected = fused;
        } else if (split) {
            RestoreParams ra = rp, rb = rp;
            ra.epf_iters = 4;  // restore_fused_body.h: the 13-tap iteration alone
            ra.xyb = 0; ra.transfer = JXL_TRANSFER_NONE; ra.max_value = 0; ra.out_elem = 4; ra.interleaved = 0;
            rb.gab = 0; rb.epf_iters = 2;
            void* mid[3] = {oth[0], oth[1], oth[2]};
            if (!do_out)
                for (int i = 0; i < 3; i++) dst[i] = cur[i];  // the input planes are dead once the first launch has read them
            const float* mids[3] = {oth[0], oth[1], oth[2]};
            fused = launch_restore_fused(src, mid, c->H, c->W, c->hf_mul.as<int32_t>(), c->sharp.as<int32_t>(), ra, s) &&
                    launch_restore_fused(mids, dst, c->H, c->W, c->hf_mul.as<int32_t>(), c->sharp.as<int32_t>(), rb, s);
            if (fused) launches++;
        } else {
            if (c->timing) {  // the kernel records its own start / stop (jxl_vardct_last_stage_ms, which = 3)
                g_restore_kernel_ev[0] = c->kev[c->ev_runs % jxl_ctx::kEvSlots][0];
                g_restore_kernel_ev[1] = c->kev[c->ev_runs % jxl_ctx::kEvSlots][1];
            }
            fused = launch_restore_fused(src, dst, c->H, c->W, c->hf_mul.as<int32_t>(), c->sharp.as<int32_t>(), rp, s);
            g_restore_kernel_ev[0] = g_restore_kernel_ev[1] = nullptr;
            c->kev_valid = c->timing && fused;
        }
        if (fused) {
            launches++;
            for (int i = 0; i < 3; i++) c->result[i] = dst[i];
            c->result_elem = rp.out_elem;
            c->result_interleaved = rp.interleaved != 0;
        }
    }
    if (!fused) {
        if (do_gab) {
            const float* src[3] = {cur[0], cur[1], cur[2]};
            launch_gab(src, oth, c->H, c->W, p.gab_w1, p.gab_w2, s);
            std::swap(cur, oth);
            launches++;
        }
        if (do_epf) {
            launch_epf_sigma(c->hf_mul.as<int32_t>(), c->sharp.as<int32_t>(), c->bh, c->bw, p.global_scale_f, p.epf_sharp_lut,
                             c->inv_sigma.as<float>(), c->bad_flag.as<int>(), s);
            launches++;
            for (int i = 0; i < 3; i++) {  // Frame.java:583-587
                if (i == 0 && p.epf_iters < 3) continue;
                if (i == 2 && p.epf_iters < 2) break;
                const float* src[3] = {cur[0], cur[1], cur[2]};
                launch_epf_iter(src, oth, c->H, c->W, i, c->inv_sigma.as<float>(),
                                make_epf(p.epf_channel_scale, p.epf_pass0_sigma_scale, p.epf_pass2_sigma_scale, p.epf_border_sad_mul, i, 0.0f), s);
                std::swap(cur, oth);
                launches++;
            }
        }
        if (do_xyb) {
            launch_xyb(cur, (int64_t)c->W * c->H, make_xyb(p.opsin_matrix, p.opsin_bias, p.cbrt_opsin_bias, p.intensity_target), s);
            launches++;
        }
        if (do_out) {
            const int maxv = out_max_value(p.out_format);
            const int es = out_elem_size(p.out_format);
            const bool il = out_interleaved(p.out_format);
            for (int i = 0; i < 3; i++) {
                launch_transfer(cur[i], (int64_t)c->W * c->H, p.transfer == JXL_TRANSFER_PQ_EXACT ? JXL_TRANSFER_PQ : p.transfer, maxv,
                                c->outbuf[il ? 0 : i].p, es, s, il ? 3 : 1, il ? i : 0, p.transfer == JXL_TRANSFER_PQ_EXACT ? nullptr : c->pq_tab.as<float>(), c->srgb8_tab.as<float>(),
                                pq16_thresholds_for(c, p.transfer, maxv), srgb16_table_for(c, p.transfer, maxv));
                c->result[i] = c->outbuf[i].p;
                launches++;
            }
            c->result_elem = es;
            c->result_interleaved = il;
        } else {
            for (int i = 0; i < 3; i++) c->result[i] = cur[i];
            c->result_elem = 4;
            c->result_interleaved = false;
        }
    }
    if (c->timing) {
        (void)hipEventRecord(evs[2], s);
        c->ev_runs++;
    }
    c->last_launches = launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return JXL_OK;
}

// frames whose IDCT stage can share launches: plain 4:4:4 frames made of the merged-launch types
bool batchable(const jxl_ctx* c) {
    for (const auto& sl : c->special_launches)
        if (!sl.wg_items) return false;  // k_idct_special_batch takes workgroup items
    return c->frame_open && !c->sub && c->large_count == 0 && c->llf_count == 0 && (c->p.stages & JXL_STAGE_IDCT);
}
}  // namespace

extern "C" {

jxl_status jxl_vardct_prepare(jxl_ctx* c) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
    return finalize_tables(c);
}

jxl_status jxl_vardct_run(jxl_ctx* c) { return run_frame(c, false); }

// A batch of independent frames (one context each, all on one device): the IDCT stage of the whole batch runs as ONE
// launch per register class (k_idct_multi_batch, blockIdx.y = frame), on the first context's stream; every frame's
// restoration kernel then runs on its own stream as in jxl_vardct_run. Per-frame argument blocks live in device memory
// and are rebuilt only when a frame's binned work changed. Frames the batched kernels do not cover (chroma subsampling,
// 128/256-edge varblocks) make the call fall back to n plain runs. Results are those of n jxl_vardct_run calls.
jxl_status jxl_vardct_run_batch(jxl_ctx* const* ctxs, int32_t n) {
    if (!ctxs || n <= 0) return fail(nullptr, JXL_ERR_INVALID_ARGUMENT, "run_batch: no contexts");
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return fail(nullptr, JXL_ERR_INVALID_ARGUMENT, "run_batch: null context %d", i);
    jxl_ctx* c0 = ctxs[0];
    bool ok = n > 1 && !getenv("JXL_NO_BATCH");
    for (int i = 0; i < n && ok; i++) {
        jxl_ctx* c = ctxs[i];
        if (c->device != c0->device) return fail(c, JXL_ERR_INVALID_ARGUMENT, "run_batch: contexts on different devices");
        for (int j = 0; j < i; j++)
            if (ctxs[j] == c) return fail(c, JXL_ERR_INVALID_ARGUMENT, "run_batch: context %d listed twice", i);
        jxl_status st = bind(c);
        if (st) return st;
        if (!c->frame_open) return fail(c, JXL_ERR_STATE, "begin_frame first");
        st = finalize_tables(c);
        if (st) return st;
        if ((st = pre_run_zero(c))) return st;  // (on the frame's own stream: the shared launches are ordered behind it below)
        ok = batchable(c);
    }
    if (!ok) {
        for (int i = 0; i < n; i++) {
            const jxl_status st = run_frame(ctxs[i], false);
            if (st) return st;
        }
        return JXL_OK;
    }
    // ---- argument blocks (cached per batch composition)
    bool same = (int)c0->batch_key.size() == n;
    for (int i = 0; i < n && same; i++) same = c0->batch_key[i].first == ctxs[i] && c0->batch_key[i].second == ctxs[i]->tables_gen;
    if (!same) {
        std::vector<MultiArgs> host_args;
        c0->batch_launches.clear();
        for (int cls : {1, 0, 3}) {  // launch order of jxl_vardct_run: heaviest class first, the special kernel last
            jxl_ctx::BatchLaunch bl{cls, 0, 0, 0, host_args.size() * sizeof(MultiArgs)};
            for (int i = 0; i < n; i++) {
                jxl_ctx* c = ctxs[i];
                DevFrame f;
                fill_dev_frame(c, f);
                float* A[3] = {c->planeA[0].as<float>(), c->planeA[1].as<float>(), c->planeA[2].as<float>()};
                MultiArgs a{};
                if (cls == 3) {
                    if (c->special_launches.empty()) continue;
                    const auto& sl = c->special_launches[0];
                    a.f = f;
                    a.blocks = c->blocks.as<DevBlock>();
                    a.items = c->items.as<WorkItem>() + sl.items_off;
                    a.seg_n[0] = sl.n_items;
                    a.o0 = A[0]; a.o1 = A[1]; a.o2 = A[2];
                    bl.grid_x = std::max(bl.grid_x, sl.n_items);
                } else {
                    const jxl_ctx::TypeLaunch* tl = nullptr;
                    for (const auto& t : c->type_launches)
                        if (t.cls == cls) tl = &t;
                    if (!tl) continue;
                    size_t lds = 0;
                    const int g = build_idct_multi_args(f, c->blocks.as<DevBlock>(), tl->segs.data(), (int)tl->segs.size(), 3, 0, A, a, &lds);
                    if (g < 0) return fail(c0, JXL_ERR_STATE, "IDCT launch: too many segments");
                    if (g <= 0) continue;
                    bl.grid_x = std::max(bl.grid_x, g);
                    bl.lds_bytes = std::max(bl.lds_bytes, lds);
                }
                host_args.push_back(a);
                bl.n_frames++;
            }
            if (bl.n_frames > 0) c0->batch_launches.push_back(bl);
        }
        // the persistent three-channel kernels (k_idct_wg3.hip): LLF pre-pass of both classes, then the two classes
        std::vector<Wg3Args> wg3_args;
        static const int wg3_cap = getenv("JXL_WG3_GRID") ? atoi(getenv("JXL_WG3_GRID")) : 768;
        static const int wg3_cap_big = getenv("JXL_WG3_GRID_BIG") ? atoi(getenv("JXL_WG3_GRID_BIG")) : 512;
        for (int cls : {10, 11, 12}) {
            jxl_ctx::BatchLaunch bl{cls, 0, 0, 0, wg3_args.size() * sizeof(Wg3Args)};
            int64_t max_n = 0;
            for (int i = 0; i < n; i++) {
                jxl_ctx* c = ctxs[i];
                DevFrame f;
                fill_dev_frame(c, f);
                float* A[3] = {c->planeA[0].as<float>(), c->planeA[1].as<float>(), c->planeA[2].as<float>()};
                std::vector<IdctSegment> segs;
                for (const auto& tl : c->type_launches)
                    if (tl.cls >= 2 && (cls == 10 || tl.cls == cls - 9)) segs.insert(segs.end(), tl.segs.begin(), tl.segs.end());
                if (segs.empty()) continue;
                Wg3Args a;
                const int items = build_wg3_args(f, c->blocks.as<DevBlock>(), segs.data(), (int)segs.size(), cls == 10 ? 2 : cls - 11, A, a);
                if (items < 0) return fail(c0, JXL_ERR_STATE, "IDCT launch: too many segments");
                if (items <= 0) continue;
                if (cls != 10) {
                    if (c->wg3_item_count[cls - 11] < items) return fail(c0, JXL_ERR_STATE, "IDCT launch: item list out of date");
                    a.total_items = c->wg3_item_count[cls - 11];
                    a.items = c->wg3_items[cls - 11].as<int>();
                }
                if (cls == 10) {
                    const int64_t nl = a.llf_in_item ? 0 : wg3_llf_count(a);  // the items do finalizeLLF themselves: no launch
                    if (nl <= 0) continue;
                    max_n = std::max(max_n, nl);
                } else {
                    max_n = std::max<int64_t>(max_n, items);
                    bl.lds_bytes = std::max(bl.lds_bytes, wg3_lds_bytes(a));
                }
                wg3_args.push_back(a);
                bl.n_frames++;
            }
            if (bl.n_frames <= 0) continue;
            // persistent grids: the chip-wide cap shared between the frames of the launch
            const int cap = cls == 11 ? wg3_cap : wg3_cap_big;
            bl.grid_x = cls == 10 ? (int)std::min<int64_t>(max_n, INT32_MAX) : (int)std::min<int64_t>(max_n, std::max(1, (cap + bl.n_frames - 1) / bl.n_frames));
            c0->batch_launches.push_back(bl);
        }
        HIP_TRY(c0, hipSetDevice(c0->device));
        if (!c0->batch_args.ensure(std::max<size_t>(sizeof(MultiArgs), host_args.size() * sizeof(MultiArgs))) ||
            !c0->batch_wg3_args.ensure(std::max<size_t>(sizeof(Wg3Args), wg3_args.size() * sizeof(Wg3Args))))
            return fail(c0, JXL_ERR_OOM, "device allocation failed (batch arguments)");
        HIP_TRY(c0, hipStreamSynchronize(c0->stream));  // an earlier batch may still be reading the old blocks
        if (c0->n_aux > 0) HIP_TRY(c0, hipStreamSynchronize(c0->aux[0]));
        if (!host_args.empty())
            HIP_TRY(c0, hipMemcpy(c0->batch_args.p, host_args.data(), host_args.size() * sizeof(MultiArgs), hipMemcpyHostToDevice));
        if (!wg3_args.empty())
            HIP_TRY(c0, hipMemcpy(c0->batch_wg3_args.p, wg3_args.data(), wg3_args.size() * sizeof(Wg3Args), hipMemcpyHostToDevice));
        c0->batch_key.clear();
        for (int i = 0; i < n; i++) c0->batch_key.emplace_back(ctxs[i], ctxs[i]->tables_gen);
        c0->batch_restore_valid = false;
        if (!c0->batch_ev) HIP_TRY(c0, hipEventCreateWithFlags(&c0->batch_ev, hipEventDisableTiming));
    }
    // ---- the batched IDCT stage on the first context's streams, ordered after whatever the frames' own streams still run
    hipStream_t s0 = c0->stream;
    for (int i = 1; i < n; i++) {
        (void)hipEventRecord(ctxs[i]->fork_ev, ctxs[i]->stream);
        (void)hipStreamWaitEvent(s0, ctxs[i]->fork_ev, 0);
    }
    const bool fork = c0->n_aux > 0 && c0->batch_launches.size() > 1;
    if (fork) {
        (void)hipEventRecord(c0->fork_ev, s0);
        (void)hipStreamWaitEvent(c0->aux[0], c0->fork_ev, 0);
    }
    // as in run_frame: the special kernel (no LLF needed) and the 64-point family go to the side stream, the LLF pre-pass and
    // the launch of everything up to 32 points to the main one; the r1 classes alternate
    int k = 0;
    hipStream_t side = fork ? c0->aux[0] : s0;
    for (const auto& bl : c0->batch_launches) {
        if (bl.cls >= 10) {
            const Wg3Args* da = reinterpret_cast<const Wg3Args*>(static_cast<const char*>(c0->batch_wg3_args.p) + bl.offset);
            if (bl.cls == 10) {
                launch_llf_wg3_batch(da, bl.n_frames, bl.grid_x, s0);
                if (fork) {
                    (void)hipEventRecord(c0->llf_ev, s0);  // "LLF planes written"
                    (void)hipStreamWaitEvent(side, c0->llf_ev, 0);
                }
            } else {
                launch_idct_wg3_batch(da, bl.n_frames, bl.cls == 12, bl.grid_x, bl.lds_bytes, bl.cls == 12 ? side : s0);
            }
            continue;
        }
        const MultiArgs* da = reinterpret_cast<const MultiArgs*>(static_cast<const char*>(c0->batch_args.p) + bl.offset);
        if (bl.cls == 3) launch_idct_special_batch(da, bl.n_frames, bl.grid_x, side);
        else launch_idct_multi_batch(da, bl.n_frames, bl.grid_x, bl.lds_bytes, bl.cls, (fork && (k++ & 1)) ? side : s0);
    }
    if (fork) {
        (void)hipEventRecord(c0->join_ev[0], c0->aux[0]);
        (void)hipStreamWaitEvent(s0, c0->join_ev[0], 0);
    }
    // ---- the restoration stage: one launch for the whole batch when every frame takes the fused kernel in the same
    //      variant (Gaborish, EPF iterations, output kind); otherwise per frame, on the frame's own stream
    static const bool batch_restore = !getenv("JXL_NO_BATCH_RESTORE");
    std::vector<FusedArgs> fa((size_t)n);
    bool all = batch_restore;
    for (int i = 0; i < n; i++) all = all && ctxs[i]->W >= 8 && ctxs[i]->H >= 8;  // what the fused kernel covers: in collect
                                                                                   // mode such a frame enqueues nothing
    // r6: a frame with three EPF iterations launches its own pair of kernels (epf3_split_on): decided HERE, before any
    // frame's collect pass -- run_frame(collect) used to return early for such a frame after its bookkeeping, and the collect passes
    // of the frames in front of it were thrown away
    for (int i = 0; i < n && all; i++) {
        const jxl_vardct_params& q = ctxs[i]->p;
        if (epf3_split_on() && (q.stages & JXL_STAGE_EPF) && q.epf_iters == 3) all = false;
    }
    if (all) {
        for (int i = 0; i < n && all; i++) {
            bool got = false;
            const jxl_status st = run_frame(ctxs[i], true, &fa[(size_t)i], &got);  // bookkeeping + argument block, no launch
            if (st) return st;
            all = got && restore_fused_variant(fa[(size_t)i]) == restore_fused_variant(fa[0]);  // !got: no restoration stage enabled
        }
    }
    if (all) {
        if (!c0->batch_restore_valid || c0->batch_restore_host.size() != (size_t)n ||
            memcmp(c0->batch_restore_host.data(), fa.data(), sizeof(FusedArgs) * (size_t)n) != 0) {
            if (!c0->batch_restore_args.ensure(sizeof(FusedArgs) * (size_t)n)) return fail(c0, JXL_ERR_OOM, "device allocation failed (batch arguments)");
            HIP_TRY(c0, hipStreamSynchronize(s0));  // the previous batch may still be reading the old blocks
            HIP_TRY(c0, hipMemcpy(c0->batch_restore_args.p, fa.data(), sizeof(FusedArgs) * (size_t)n, hipMemcpyHostToDevice));
            c0->batch_restore_host = fa;
            c0->batch_restore_valid = true;
        }
        launch_restore_fused_batch(fa.data(), c0->batch_restore_args.as<FusedArgs>(), n, s0);
        (void)hipEventRecord(c0->batch_ev, s0);
        for (int i = 1; i < n; i++) (void)hipStreamWaitEvent(ctxs[i]->stream, c0->batch_ev, 0);
        // run_frame in collect mode has already counted the fused launch for every frame; the shared IDCT launches are
        // attributed to the first context
        c0->last_launches += (int)c0->batch_launches.size();
        return JXL_OK;
    }
    (void)hipEventRecord(c0->batch_ev, s0);
    for (int i = 1; i < n; i++) (void)hipStreamWaitEvent(ctxs[i]->stream, c0->batch_ev, 0);
    for (int i = 0; i < n; i++) {
        const jxl_status st = run_frame(ctxs[i], true);
        if (st) return st;
        ctxs[i]->last_launches += (i == 0 ? (int)c0->batch_launches.size() : 0);
    }
    return JXL_OK;
}

jxl_status jxl_vardct_last_stage_ms(jxl_ctx* c, int32_t which, float* ms) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!ms || c->ev_runs <= 0) return fail(c, JXL_ERR_STATE, "enable stage timing and run first");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n = std::min(c->ev_runs, (int)jxl_ctx::kEvSlots);
    const int a = which == 2 ? 1 : 0, b = which == 1 ? 1 : 2;
    if (which == 3 && !c->kev_valid) return fail(c, JXL_ERR_STATE, "the timed runs did not take the single-launch fused restoration kernel");
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        float t = 0.0f;
        if (which == 3) HIP_TRY(c, hipEventElapsedTime(&t, c->kev[i][0], c->kev[i][1]));  // the kernel's own start -> stop
        else HIP_TRY(c, hipEventElapsedTime(&t, c->ev[i][a], c->ev[i][b]));
        sum += t;
    }
    *ms = (float)(sum / n);
    return JXL_OK;
}

int32_t jxl_vardct_out_elem_size(const jxl_ctx* c) { return c ? c->result_elem : 0; }
int32_t jxl_vardct_last_launch_count(const jxl_ctx* c) { return c ? c->last_launches : 0; }

// r5: a result buffer in page-locked memory is written by a kernel over PCIe (copy_zero, above) instead of through hipMemcpyAsync.
// JXL_OUTPUT_ZEROCOPY=0: the runtime's copy.
static int copy_out_zero(jxl_ctx* c, void* dst, const void* src, size_t bytes) {
    static const bool on = !(getenv("JXL_OUTPUT_ZEROCOPY") && atoi(getenv("JXL_OUTPUT_ZEROCOPY")) == 0);
    static const int grid_env = getenv("JXL_OUTPUT_GRID") ? std::max(1, atoi(getenv("JXL_OUTPUT_GRID"))) : 0;
    return on ? copy_zero(c, dst, src, bytes, true, grid_env ? grid_env : bus_grid(c, true), true) : 0;
}

// the copies of the last run's result planes to the host, queued on the context's stream
static jxl_status enqueue_output(jxl_ctx* c, void* const out[3], int64_t out_stride) {
    if (!c->result[0]) return fail(c, JXL_ERR_STATE, "nothing has been run");
    if (!out || out_stride < c->W) return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad output planes");
    const size_t es = (size_t)c->result_elem;
    // dense destination rows (the usual case): one linear copy per buffer -- hipMemcpy2D into pageable memory goes row by row
    // (measured: 31 ms for a 25 MB RGB8 4K frame against 3 ms linear)
    const bool dense = out_stride == c->W;
    if (c->result_interleaved) {  // one buffer, rows of 3*W samples; out_stride counts pixels
        if (!out[0]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null output buffer");
        if (dense) {
            const int z = copy_out_zero(c, out[0], c->result[0], (size_t)c->W * 3 * es * c->H);
            if (z < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "page-locked output buffer is shorter than the frame (%zu bytes)", (size_t)c->W * 3 * es * c->H);
            if (!z) HIP_TRY(c, hipMemcpyAsync(out[0], c->result[0], (size_t)c->W * 3 * es * c->H, hipMemcpyDeviceToHost, c->stream));
        } else HIP_TRY(c, hipMemcpy2DAsync(out[0], (size_t)out_stride * 3 * es, c->result[0], (size_t)c->W * 3 * es, (size_t)c->W * 3 * es, c->H,
                                         hipMemcpyDeviceToHost, c->stream));
        return JXL_OK;
    }
    for (int i = 0; i < 3; i++) {
        if (!out[i]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null output plane %d", i);
        if (dense) {
            const int z = copy_out_zero(c, out[i], c->result[i], (size_t)c->W * es * c->H);
            if (z < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "page-locked output plane %d is shorter than the frame (%zu bytes)", i, (size_t)c->W * es * c->H);
            if (!z) HIP_TRY(c, hipMemcpyAsync(out[i], c->result[i], (size_t)c->W * es * c->H, hipMemcpyDeviceToHost, c->stream));
        } else HIP_TRY(c, hipMemcpy2DAsync(out[i], (size_t)out_stride * es, c->result[i], (size_t)c->W * es, (size_t)c->W * es, c->H,
                                         hipMemcpyDeviceToHost, c->stream));
    }
    return JXL_OK;
}

jxl_status jxl_vardct_read_output(jxl_ctx* c, void* const out[3], int64_t out_stride) {
    jxl_status st = bind(c);
    if (st) return st;
    if ((st = enqueue_output(c, out, out_stride))) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "device error: %s", hipGetErrorString(e));
    return JXL_OK;
}

jxl_status jxl_vardct_read_output_begin(jxl_ctx* c, void* const out[3], int64_t out_stride) {
    jxl_status st = bind(c);
    if (st) return st;
    if ((st = enqueue_output(c, out, out_stride))) return st;
    if (!c->out_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->out_ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->out_ev, c->stream));
    c->out_inflight = true;
    return JXL_OK;
}

jxl_status jxl_vardct_read_output_wait(jxl_ctx* c) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->out_inflight) return fail(c, JXL_ERR_STATE, "read_output_begin first");
    HIP_TRY(c, hipEventSynchronize(c->out_ev));
    c->out_inflight = false;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "device error: %s", hipGetErrorString(e));
    return JXL_OK;
}

jxl_status jxl_vardct_finish_frame(jxl_ctx* c, void* const out[3], int64_t out_stride) {
    jxl_status st = jxl_vardct_run(c);
    if (st) return st;
    return jxl_vardct_read_output(c, out, out_stride);
}

jxl_status jxl_vardct_copy_output_device(jxl_ctx* c, void* dst_device) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->result[0] || !dst_device) return fail(c, JXL_ERR_STATE, "nothing has been run");
    const size_t bytes = (size_t)c->W * c->H * c->result_elem;
    if (c->result_interleaved) {
        HIP_TRY(c, hipMemcpyAsync(dst_device, c->result[0], 3 * bytes, hipMemcpyDeviceToDevice, c->stream));
        return JXL_OK;
    }
    for (int i = 0; i < 3; i++)
        HIP_TRY(c, hipMemcpyAsync((char*)dst_device + i * bytes, c->result[i], bytes, hipMemcpyDeviceToDevice, c->stream));
    return JXL_OK;
}

// ---- resident colour planes (the stages of JXLCodestreamDecoder.java:628-637 chained on the device) -----------------------
jxl_status jxl_planes_from_frame(jxl_ctx* c, int32_t height, int32_t width) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!c->result[0]) return fail(c, JXL_ERR_STATE, "nothing has been run");
    if (c->result_elem != 4 || c->result_interleaved || ((c->p.stages & JXL_STAGE_OUT) && (c->p.transfer != JXL_TRANSFER_NONE || c->p.out_format != JXL_OUT_F32)))
        return fail(c, JXL_ERR_STATE, "the frame's result is not a set of float planes");
    if (height <= 0 || width <= 0 || height > c->H || width > c->W) return fail(c, JXL_ERR_INVALID_ARGUMENT, "planes: window outside the frame");
    for (int i = 0; i < 3; i++) {
        if (!c->rp[i].ensure(sizeof(float) * (size_t)height * width)) return fail(c, JXL_ERR_OOM, "device allocation failed (resident planes)");
        HIP_TRY(c, hipMemcpy2DAsync(c->rp[i].p, sizeof(float) * (size_t)width, c->result[i], sizeof(float) * (size_t)c->W,
                                    sizeof(float) * (size_t)width, (size_t)height, hipMemcpyDeviceToDevice, c->stream));
    }
    c->rp_h = height;
    c->rp_w = width;
    return JXL_OK;
}

jxl_status jxl_planes_shape(const jxl_ctx* c, int32_t* height, int32_t* width) {
    if (!c || !height || !width) return JXL_ERR_INVALID_ARGUMENT;
    *height = c->rp_h;
    *width = c->rp_w;
    return JXL_OK;
}

jxl_status jxl_planes_upsample(jxl_ctx* c, int32_t k, const float* weights) {
    jxl_status st = bind(c);
    if (st) return st;
    if (c->rp_h <= 0) return fail(c, JXL_ERR_STATE, "no resident planes");
    if (!weights || (k != 2 && k != 4 && k != 8)) return fail(c, JXL_ERR_INVALID_ARGUMENT, "upsample: bad arguments");
    Tmp t;
    float* dw = t.up(weights, (size_t)k * k * 25);
    if (!dw) return fail(c, JXL_ERR_OOM, "device allocation failed");
    const size_t n = (size_t)c->rp_h * c->rp_w;
    for (int i = 0; i < 3; i++) {
        if (!c->rp_tmp[i].ensure(sizeof(float) * n * k * k)) return fail(c, JXL_ERR_OOM, "device allocation failed (resident planes)");
        launch_upsample(c->rp[i].as<float>(), c->rp_h, c->rp_w, k, dw, c->rp_tmp[i].as<float>(), c->stream);
        std::swap(c->rp[i], c->rp_tmp[i]);
    }
    c->rp_h *= k;
    c->rp_w *= k;
    return finish(c);  // the weights are freed on return
}

jxl_status jxl_planes_noise(jxl_ctx* c, int32_t group_dim, uint64_t seed0, const float lut[8], float base_corr_x, float base_corr_b) {
    jxl_status st = bind(c);
    if (st) return st;
    if (c->rp_h <= 0) return fail(c, JXL_ERR_STATE, "no resident planes");
    if (!lut || group_dim < 16 || (group_dim & (group_dim - 1))) return fail(c, JXL_ERR_INVALID_ARGUMENT, "noise: bad arguments");
    const size_t n = (size_t)c->rp_h * c->rp_w;
    float *tmp[3], *nz[3], *pl[3];
    for (int i = 0; i < 3; i++) {
        if (!c->rp_tmp[i].ensure(sizeof(float) * n) || !c->rp_noise[i].ensure(sizeof(float) * n))
            return fail(c, JXL_ERR_OOM, "device allocation failed (noise planes)");
        tmp[i] = c->rp_tmp[i].as<float>();
        nz[i] = c->rp_noise[i].as<float>();
        pl[i] = c->rp[i].as<float>();
    }
    launch_noise_init(c->rp_h, c->rp_w, group_dim, seed0, 3, tmp, nz, c->stream);
    const float* cn[3] = {nz[0], nz[1], nz[2]};
    launch_noise_add(pl, cn, (int64_t)n, lut, base_corr_x, base_corr_b, c->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return JXL_OK;
}

jxl_status jxl_planes_xyb(jxl_ctx* c, const float matrix[9], const float opsin_bias[3], const float cbrt_opsin_bias[3], float intensity_target) {
    jxl_status st = bind(c);
    if (st) return st;
    if (c->rp_h <= 0) return fail(c, JXL_ERR_STATE, "no resident planes");
    if (!matrix || !opsin_bias || !cbrt_opsin_bias) return fail(c, JXL_ERR_INVALID_ARGUMENT, "xyb: bad arguments");
    float* pl[3] = {c->rp[0].as<float>(), c->rp[1].as<float>(), c->rp[2].as<float>()};
    launch_xyb(pl, (int64_t)c->rp_h * c->rp_w, make_xyb(matrix, opsin_bias, cbrt_opsin_bias, intensity_target), c->stream);
    return JXL_OK;
}

jxl_status jxl_planes_ycbcr(jxl_ctx* c) {
    jxl_status st = bind(c);
    if (st) return st;
    if (c->rp_h <= 0) return fail(c, JXL_ERR_STATE, "no resident planes");
    float* pl[3] = {c->rp[0].as<float>(), c->rp[1].as<float>(), c->rp[2].as<float>()};
    launch_ycbcr(pl, (int64_t)c->rp_h * c->rp_w, c->stream);
    return JXL_OK;
}

jxl_status jxl_planes_download(jxl_ctx* c, float* const out[3]) {
    jxl_status st = bind(c);
    if (st) return st;
    if (c->rp_h <= 0) return fail(c, JXL_ERR_STATE, "no resident planes");
    if (!out || !out[0] || !out[1] || !out[2]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "null output plane");
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(out[i], c->rp[i].p, sizeof(float) * (size_t)c->rp_h * c->rp_w, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_planes_upload(jxl_ctx* c, const float* const in[3], int32_t height, int32_t width) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || !in[0] || !in[1] || !in[2] || height <= 0 || width <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "planes: bad arguments");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) {
        if (!c->rp[i].ensure(sizeof(float) * (size_t)height * width)) return fail(c, JXL_ERR_OOM, "device allocation failed (resident planes)");
        HIP_TRY(c, hipMemcpy(c->rp[i].p, in[i], sizeof(float) * (size_t)height * width, hipMemcpyHostToDevice));
    }
    c->rp_h = height;
    c->rp_w = width;
    return JXL_OK;
}

// ---- stage-level entries -----------------------------------------------------------------------------
jxl_status jxl_stage_idct2d(jxl_ctx* c, const float* src, float* dst, int32_t h, int32_t w, int32_t transposed) {
    jxl_status st = bind(c);
    if (st) return st;
    auto pow2 = [](int v) { return v >= 1 && v <= 256 && (v & (v - 1)) == 0; };
    if (!src || !dst || !pow2(h) || !pow2(w)) return fail(c, JXL_ERR_INVALID_ARGUMENT, "idct2d: sizes must be powers of two <= 256");
    Tmp t;
    float* ds = t.up(src, (size_t)h * w);
    float* dd = t.up<float>(nullptr, (size_t)h * w);
    if (!ds || !dd) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_idct2d_single(ds, dd, h, w, transposed, c->lut.as<float>(), c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(dst, dd, sizeof(float) * h * w, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_fdct2d(jxl_ctx* c, const float* src, float* dst, int32_t h, int32_t w) {
    jxl_status st = bind(c);
    if (st) return st;
    auto pow2 = [](int v) { return v >= 1 && v <= 256 && (v & (v - 1)) == 0; };
    if (!src || !dst || !pow2(h) || !pow2(w)) return fail(c, JXL_ERR_INVALID_ARGUMENT, "fdct2d: sizes must be powers of two <= 256");
    Tmp t;
    float* ds = t.up(src, (size_t)h * w);
    float* dd = t.up<float>(nullptr, (size_t)h * w);
    if (!ds || !dd) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_fdct2d_single(ds, dd, h, w, c->lut.as<float>(), c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(dst, dd, sizeof(float) * h * w, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_gab(jxl_ctx* c, const float* const in[3], float* const out[3], int32_t height, int32_t width,
                         const float w1[3], const float w2[3]) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || !out || height <= 0 || width <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "gab: bad arguments");
    Tmp t;
    const size_t n = (size_t)height * width;
    const float* di[3];
    float* dd[3];
    for (int i = 0; i < 3; i++) {
        di[i] = t.up(in[i], n);
        dd[i] = t.up<float>(nullptr, n);
        if (!di[i] || !dd[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_gab(di, dd, height, width, w1, w2, c->stream);
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(out[i], dd[i], 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_epf_sigma(jxl_ctx* c, const int32_t* hf_mul, const int32_t* sharpness, int32_t bh, int32_t bw,
                               float global_scale_f, const float sharp_lut[8], float* inv_sigma) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!hf_mul || !sharpness || !inv_sigma || bh <= 0 || bw <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "epf_sigma: bad arguments");
    Tmp t;
    const size_t n = (size_t)bh * bw;
    int32_t* dh = t.up(hf_mul, n);
    int32_t* ds = t.up(sharpness, n);
    float* dv = t.up<float>(nullptr, n);
    if (!dh || !ds || !dv) return fail(c, JXL_ERR_OOM, "device allocation failed");
    HIP_TRY(c, hipMemsetAsync(c->bad_flag.p, 0, sizeof(int), c->stream));
    launch_epf_sigma(dh, ds, bh, bw, global_scale_f, sharp_lut, dv, c->bad_flag.as<int>(), c->stream);
    if ((st = finish(c))) return st;
    int bad = 0;
    HIP_TRY(c, hipMemcpy(&bad, c->bad_flag.p, sizeof(int), hipMemcpyDeviceToHost));
    if (bad) return fail(c, JXL_ERR_INVALID_BITSTREAM, "Invalid EPF Sharpness");  // Frame.java:565-566
    HIP_TRY(c, hipMemcpy(inv_sigma, dv, 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_epf(jxl_ctx* c, const float* const in[3], float* const out[3], int32_t height, int32_t width,
                         int32_t iterations, const float* inv_sigma, float inv_sigma_modular, const float channel_scale[3],
                         float pass0_sigma_scale, float pass2_sigma_scale, float border_sad_mul) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || !out || height <= 0 || width <= 0 || iterations < 0 || iterations > 3) return fail(c, JXL_ERR_INVALID_ARGUMENT, "epf: bad arguments");
    Tmp t;
    const size_t n = (size_t)height * width;
    float* a[3];
    float* b[3];
    for (int i = 0; i < 3; i++) {
        a[i] = t.up(in[i], n);
        b[i] = t.up<float>(nullptr, n);
        if (!a[i] || !b[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    float* dsig = nullptr;
    if (inv_sigma) {
        dsig = t.up(inv_sigma, (size_t)((height + 7) >> 3) * ((width + 7) >> 3));
        if (!dsig) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    float** cur = a;
    float** oth = b;
    for (int i = 0; i < 3 && iterations > 0; i++) {
        if (i == 0 && iterations < 3) continue;
        if (i == 2 && iterations < 2) break;
        const float* src[3] = {cur[0], cur[1], cur[2]};
        launch_epf_iter(src, oth, height, width, i, dsig,
                        make_epf(channel_scale, pass0_sigma_scale, pass2_sigma_scale, border_sad_mul, i, inv_sigma_modular), c->stream);
        std::swap(cur, oth);
    }
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(out[i], cur[i], 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_xyb(jxl_ctx* c, float* const planes[3], int64_t n, const float matrix[9], const float opsin_bias[3],
                         const float cbrt_opsin_bias[3], float intensity_target) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!planes || n < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "xyb: bad arguments");  // "Can only XYB on 3 channels"
    Tmp t;
    float* d[3];
    for (int i = 0; i < 3; i++) {
        d[i] = t.up(planes[i], (size_t)n);
        if (!d[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_xyb(d, n, make_xyb(matrix, opsin_bias, cbrt_opsin_bias, intensity_target), c->stream);
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(planes[i], d[i], 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_ycbcr(jxl_ctx* c, float* const planes[3], int64_t n) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!planes || n < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "ycbcr: bad arguments");
    Tmp t;
    float* d[3];
    for (int i = 0; i < 3; i++) {
        d[i] = t.up(planes[i], (size_t)n);
        if (!d[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_ycbcr(d, n, c->stream);
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(planes[i], d[i], 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_transfer(jxl_ctx* c, const float* in, int64_t n, int32_t transfer, int32_t max_value, float* out_f,
                              int32_t* out_i) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || n < 0 || transfer < 0 || transfer > JXL_TRANSFER_PQ_EXACT || max_value < 0 || (max_value > 0 ? !out_i : !out_f))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "transfer: bad arguments");
    Tmp t;
    float* di = t.up(in, (size_t)n);
    int32_t* dout = t.up<int32_t>(nullptr, (size_t)n);
    if (!di || !dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_transfer(di, n, transfer == JXL_TRANSFER_PQ_EXACT ? JXL_TRANSFER_PQ : transfer, max_value, dout, 4, c->stream, 1, 0,
                    transfer == JXL_TRANSFER_PQ_EXACT ? nullptr : c->pq_tab.as<float>(), c->srgb8_tab.as<float>(),
                    pq16_thresholds_for(c, transfer, max_value), srgb16_table_for(c, transfer, max_value));
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(max_value > 0 ? (void*)out_i : (void*)out_f, dout, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_inv_hsqueeze(jxl_ctx* c, const int32_t* avg, int32_t aw, const int32_t* res, int32_t rw, int32_t h, int32_t* out) {
    jxl_status st = bind(c);
    if (st) return st;
    // shape checks of ModularChannel.java:363-366 -> IllegalArgumentException
    if (aw < 0 || rw < 0 || h < 0 || (aw != rw && aw != rw + 1) || !out) return fail(c, JXL_ERR_INVALID_ARGUMENT, "Corrupted squeeze transform");
    Tmp t;
    int32_t* da = t.up(avg, (size_t)aw * h);
    int32_t* dr = t.up(res, (size_t)rw * h);
    int32_t* dout = t.up<int32_t>(nullptr, (size_t)(aw + rw) * h);
    if (!da || !dr || !dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_inv_hsqueeze(da, aw, dr, rw, h, dout, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dout, 4 * (size_t)(aw + rw) * h, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_inv_vsqueeze(jxl_ctx* c, const int32_t* avg, int32_t ah, const int32_t* res, int32_t rh, int32_t w, int32_t* out) {
    jxl_status st = bind(c);
    if (st) return st;
    // ModularChannel.java:391-394 -> IllegalStateException
    if (ah < 0 || rh < 0 || w < 0 || (ah != rh && ah != rh + 1) || !out) return fail(c, JXL_ERR_STATE, "Corrupted squeeze transform");
    Tmp t;
    int32_t* da = t.up(avg, (size_t)ah * w);
    int32_t* dr = t.up(res, (size_t)rh * w);
    int32_t* dout = t.up<int32_t>(nullptr, (size_t)(ah + rh) * w);
    if (!da || !dr || !dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_inv_vsqueeze(da, ah, dr, rh, w, dout, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dout, 4 * (size_t)(ah + rh) * w, hipMemcpyDeviceToHost));
    return JXL_OK;
}

static const int kPermutationLut[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};  // ModularStream.java:35-38

jxl_status jxl_stage_rct(jxl_ctx* c, int32_t* const v[3], int64_t n, int32_t rct_type) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!v || n < 0 || rct_type < 0 || rct_type >= 42) return fail(c, JXL_ERR_INVALID_ARGUMENT, "rct: bad arguments");
    Tmp t;
    int32_t* d[3];
    for (int i = 0; i < 3; i++) {
        d[i] = t.up(v[i], (size_t)n);
        if (!d[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_rct(d[0], d[1], d[2], n, rct_type % 7, c->stream);
    if ((st = finish(c))) return st;
    const int perm = rct_type / 7;
    for (int j = 0; j < 3; j++)  // channels.set(start + permutationLut[permutation][j], v[j])
        HIP_TRY(c, hipMemcpy(v[kPermutationLut[perm][j]], d[j], 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_modular_to_float(jxl_ctx* c, const int32_t* a, const int32_t* b, int64_t n, float scale, float* out) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!a || !out || n < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "modular_to_float: bad arguments");
    Tmp t;
    int32_t* da = t.up(a, (size_t)n);
    int32_t* db = b ? t.up(b, (size_t)n) : nullptr;
    float* dd = t.up<float>(nullptr, (size_t)n);
    if (!da || (b && !db) || !dd) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_modular_to_float(da, db, n, scale, dd, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dd, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}


// ---- rows f4 / f3: stage-level entries ----------------------------------------------------------------
jxl_status jxl_stage_chroma_upsample(jxl_ctx* c, const float* in, int32_t h, int32_t w, int32_t x_shift, int32_t y_shift,
                                     float* out) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || !out || h <= 0 || w <= 0 || x_shift < 0 || y_shift < 0 || x_shift > 2 || y_shift > 2)
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "chroma_upsample: bad arguments");
    Tmp t;
    const size_t cap = ((size_t)h << y_shift) * ((size_t)w << x_shift);
    float* cur = t.up<float>(nullptr, cap);
    float* nxt = t.up<float>(nullptr, cap);
    if (!cur || !nxt) return fail(c, JXL_ERR_OOM, "device allocation failed");
    HIP_TRY(c, hipMemcpy(cur, in, sizeof(float) * (size_t)h * w, hipMemcpyHostToDevice));
    int ch = h, cw = w;
    for (int i = 0; i < x_shift; i++) {  // Frame.java:684-699
        launch_chroma_upsample_h(cur, ch, cw, nxt, c->stream);
        cw *= 2;
        std::swap(cur, nxt);
    }
    for (int i = 0; i < y_shift; i++) {  // :701-720
        launch_chroma_upsample_v(cur, ch, cw, nxt, c->stream);
        ch *= 2;
        std::swap(cur, nxt);
    }
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, cur, sizeof(float) * cap, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_upsampling_weights(int32_t k, const float* packed, float* out) {
    if ((k != 2 && k != 4 && k != 8) || !packed || !out) return JXL_ERR_INVALID_ARGUMENT;
    // ImageHeader.java:454-466: the k*k 5x5 kernels are mirror images of a symmetric (5k/2)x(5k/2) table stored as its
    // upper triangle
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++)
            for (int iy = 0; iy < 5; iy++)
                for (int ix = 0; ix < 5; ix++) {
                    const int j = ky < k / 2 ? iy + 5 * ky : (4 - iy) + 5 * (k - 1 - ky);
                    const int i = kx < k / 2 ? ix + 5 * kx : (4 - ix) + 5 * (k - 1 - kx);
                    const int hi = std::max(i, j), lo = std::min(i, j);
                    out[((ky * k + kx) * 5 + iy) * 5 + ix] = packed[5 * k * lo / 2 - lo * (lo - 1) / 2 + hi - lo];
                }
    return JXL_OK;
}

jxl_status jxl_stage_upsample(jxl_ctx* c, const float* in, int32_t h, int32_t w, int32_t k, const float* weights, float* out) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!in || !out || !weights || h <= 0 || w <= 0 || (k != 2 && k != 4 && k != 8))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "upsample: bad arguments");
    Tmp t;
    const size_t n = (size_t)h * w;
    float* di = t.up(in, n);
    float* dw = t.up(weights, (size_t)k * k * 25);
    float* dout = t.up<float>(nullptr, n * k * k);
    if (!di || !dw || !dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_upsample(di, h, w, k, dw, dout, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dout, sizeof(float) * n * k * k, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_noise_init(jxl_ctx* c, int32_t h, int32_t w, int32_t group_dim, uint64_t seed0, int32_t colors,
                                float* const out[3]) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!out || h <= 0 || w <= 0 || group_dim < 16 || (group_dim & (group_dim - 1)) || colors < 1 || colors > 3)
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "noise_init: bad arguments");
    Tmp t;
    const size_t n = (size_t)h * w;
    float* tmp[3] = {nullptr, nullptr, nullptr};
    float* dout[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < colors; i++) {
        tmp[i] = t.up<float>(nullptr, n);
        dout[i] = t.up<float>(nullptr, n);
        if (!tmp[i] || !dout[i] || !out[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    launch_noise_init(h, w, group_dim, seed0, colors, tmp, dout, c->stream);
    if ((st = finish(c))) return st;
    for (int i = 0; i < colors; i++) HIP_TRY(c, hipMemcpy(out[i], dout[i], 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_noise_add(jxl_ctx* c, float* const planes[3], const float* const noise[3], int64_t n, const float lut[8],
                               float base_corr_x, float base_corr_b) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!planes || !noise || !lut || n < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "noise_add: bad arguments");
    Tmp t;
    float* d[3];
    const float* dn[3];
    for (int i = 0; i < 3; i++) {
        d[i] = t.up(planes[i], (size_t)n);
        dn[i] = t.up(noise[i], (size_t)n);
        if (!d[i] || !dn[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    if (n > 0) launch_noise_add(d, dn, n, lut, base_corr_x, base_corr_b, c->stream);
    if ((st = finish(c))) return st;
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpy(planes[i], d[i], 4 * (size_t)n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_blend(jxl_ctx* c, int32_t mode, uint32_t flags, int32_t is_int, void* canvas, int32_t ch, int32_t cw,
                           const void* frame, int32_t fh, int32_t fw, const void* ref, int32_t rh, int32_t rw,
                           const float* frame_alpha, const float* ref_alpha, const jxl_blend_rect* r) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!canvas || !r || ch <= 0 || cw <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "blend: bad arguments");
    const int op = blend_op(mode, flags, is_int);
    if (op == -1) return fail(c, JXL_ERR_INVALID_BITSTREAM, "Illegal blend mode");  // JXLCodestreamDecoder.java:506
    if (op == -2) return fail(c, JXL_ERR_INVALID_ARGUMENT, "blend: this mode works on float samples");
    bool nf, nr, nfa, nra;
    blend_needs(op, &nf, &nr, &nfa, &nra, (flags & JXL_BLEND_FLAG_IS_ALPHA) != 0);
    if ((nf && !frame) || (nr && !ref) || (nfa && !frame_alpha) || (nra && !ref_alpha))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "blend: a plane this mode reads is NULL");
    // the rectangle must lie inside every plane that is touched (Java would throw ArrayIndexOutOfBounds)
    auto inside = [&](int y, int x, int H, int W) { return r->h >= 0 && r->w >= 0 && y >= 0 && x >= 0 && y + r->h <= H && x + r->w <= W; };
    const bool copy_ref = nr && !nf;  // blendMulAdd's alpha case indexes ref with frameOffset
    if (!inside(r->canvas_y, r->canvas_x, ch, cw) || ((nf || nfa) && !inside(r->frame_y, r->frame_x, fh, fw)) ||
        ((nr || nra) && !inside(copy_ref ? r->frame_y : r->ref_y, copy_ref ? r->frame_x : r->ref_x, rh, rw)))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "blend: rectangle outside a plane");
    Tmp t;
    uint32_t* dc = t.up((const uint32_t*)canvas, (size_t)ch * cw);
    uint32_t* df = nf ? t.up((const uint32_t*)frame, (size_t)fh * fw) : nullptr;
    uint32_t* dr = nr ? t.up((const uint32_t*)ref, (size_t)rh * rw) : nullptr;
    float* dfa = nfa ? t.up(frame_alpha, (size_t)fh * fw) : nullptr;
    float* dra = nra ? t.up(ref_alpha, (size_t)rh * rw) : nullptr;
    if (!dc || (nf && !df) || (nr && !dr) || (nfa && !dfa) || (nra && !dra)) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_blend(op, flags, dc, cw, df, fw, dr, rw, dfa, dra, *r, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(canvas, dc, 4 * (size_t)ch * cw, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_orient(jxl_ctx* c, const void* in, int32_t h, int32_t w, int32_t orientation, void* out) {
    jxl_status st = bind(c);
    if (st) return st;
    if (orientation < 1 || orientation > 8) return fail(c, JXL_ERR_STATE, "orientation %d", orientation);  // IllegalStateException, :107
    if (!in || !out || h <= 0 || w <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "orient: bad arguments");
    Tmp t;
    const size_t n = (size_t)h * w;
    uint32_t* di = t.up((const uint32_t*)in, n);
    uint32_t* dout = t.up<uint32_t>(nullptr, n);
    if (!di || !dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_orient(di, h, w, orientation, dout, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dout, 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_stage_pack(jxl_ctx* c, const void* const planes[4], const jxl_pack_params* p, void* out) {
    jxl_status st = bind(c);
    if (st) return st;
    if (!planes || !p || !out || p->height <= 0 || p->width <= 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "pack: bad arguments");
    if (p->bit_depth != 8 && p->bit_depth != 16) return fail(c, JXL_ERR_INVALID_ARGUMENT, "PNG only supports 8 and 16");  // PNGWriter.java:57-58
    if ((p->n_color != 1 && p->n_color != 3) || (p->premultiplied && !p->has_alpha))
        return fail(c, JXL_ERR_INVALID_ARGUMENT, "pack: bad channel layout");
    const int nch = p->n_color + (p->has_alpha ? 1 : 0);
    bool coerce = p->premultiplied != 0;  // PNGWriter.java:79-88
    for (int i = 0; i < nch && !coerce; i++) coerce = p->is_int[i] && p->tagged_depth[i] != p->bit_depth;
    for (int i = 0; i < nch; i++) {
        if (!planes[i]) return fail(c, JXL_ERR_INVALID_ARGUMENT, "pack: null plane %d", i);
        if (coerce && p->is_int[i] && (p->tagged_depth[i] < 1 || p->tagged_depth[i] > 31))
            return fail(c, JXL_ERR_INVALID_ARGUMENT, "invalid Max Value");  // ImageBuffer.java:115-116
    }
    Tmp t;
    const size_t n = (size_t)p->height * p->width;
    const void* d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < nch; i++) {
        d[i] = t.up((const uint32_t*)planes[i], n);
        if (!d[i]) return fail(c, JXL_ERR_OOM, "device allocation failed");
    }
    const size_t ob = n * nch * (p->bit_depth / 8);
    uint8_t* dout = t.up<uint8_t>(nullptr, ob);
    if (!dout) return fail(c, JXL_ERR_OOM, "device allocation failed");
    launch_pack(d, *p, coerce, dout, c->stream);
    if ((st = finish(c))) return st;
    HIP_TRY(c, hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost));
    return JXL_OK;
}

}  // extern "C"

// ---- Modular: pairing of squeeze steps for the fused kernel (r5) ---------------------------------------------------------------
namespace {
// a V step and the H step that takes its outputs as averages, channel by channel (ModularStream.java:229-254: V_k, H_k of one level)
bool squeeze_pair_fusable(const ModOp& v, const ModOp& h) {
    if (v.kind != 4 || h.kind != 4 || v.bt.horizontal || !h.bt.horizontal || v.bt.n != h.bt.n || v.bt.n <= 0) return false;
    for (int i = 0; i < v.bt.n; i++) {
        const SqueezeDesc &a = v.bt.d[i], &b = h.bt.d[i];
        if (b.a != a.o || b.adim != a.other || b.other != a.adim + a.rdim || a.rdim < 1 || b.rdim < 1) return false;
        if ((int64_t)(a.adim + a.rdim) * (b.adim + b.rdim) >= ((int64_t)1 << 29)) return false;  // 32-bit byte offsets in the kernel
    }
    return true;
}

template <class Alloc>
bool make_vh(jxl_ctx* c, const SqueezeBatch& v, const SqueezeBatch& h, VHBatch& out, Alloc&& alloc) {
    out = VHBatch{};
    out.n = v.n;
    // H segment length: long segments amortise the warm-up chunk (16 pairs per segment); short ones give a small step enough waves.
    // The big steps run at the memory system's rate for this access shape from ~1500 waves on (measured: 2040 waves of 17 chunks
    // beat 4080 of 9 and 8160 of 5), so: the longest segment that still leaves `want` tiles. A step that has only a few hundred
    // tiles even at 32 pairs is bound by how long ONE tile takes, not by throughput: 16-pair segments (a warm-up chunk and one chunk).
    const int seg_env = getenv("JXL_VH_SEG") ? atoi(getenv("JXL_VH_SEG")) : 0;  // (read per plan: the parity tests sweep it)
    const int want = getenv("JXL_VH_TILES") ? atoi(getenv("JXL_VH_TILES")) : 1500;
    const int small = getenv("JXL_VH_SMALL") ? atoi(getenv("JXL_VH_SMALL")) : 2000;
    int seg = 256;
    auto tiles_with = [&](int sg) {
        int64_t t = 0;
        for (int i = 0; i < v.n; i++) t += (int64_t)((v.d[i].adim + v.d[i].rdim + 63) / 64) * std::max(1, (h.d[i].rdim - 1 + sg - 1) / sg);
        return t;
    };
    while (seg > 32 && tiles_with(seg) < want) seg >>= 1;
    if (seg == 32 && tiles_with(32) < small) seg = 16;
    if (seg_env >= 16) seg = (seg_env + 15) & ~15;
    // chunk width: the wide form (128-byte input pieces, 256-byte output pieces, 1.5 x instead of 2 x redundant V pairs, 9 waves
    // per CU) where the step is bound by the memory system, i.e. where it got long segments; JXL_VH_CW forces one
    const int cw_env = getenv("JXL_VH_CW") ? atoi(getenv("JXL_VH_CW")) : 0;
    const int cw32_min = getenv("JXL_VH_CW32_MINSEG") ? atoi(getenv("JXL_VH_CW32_MINSEG")) : 128;
    int cw = seg >= cw32_min ? 32 : 16;
    if (cw_env == 16 || cw_env == 32) cw = cw_env;
    if (cw == 32) seg = (seg + 31) & ~31;
    out.cw = cw;
    int tile0 = 0;
    for (int i = 0; i < v.n; i++) {
        VHDesc& d = out.d[i];
        d.va = v.d[i].a;
        d.vb = v.d[i].b;
        d.hb = h.d[i].b;
        d.o = h.d[i].o;
        d.w = v.d[i].other;
        d.ah = v.d[i].adim;
        d.rh = v.d[i].rdim;
        d.rw = h.d[i].rdim;
        d.seg = seg;
        d.nseg = std::max(1, (d.rw - 1 + seg - 1) / seg);  // segment s covers the pairs [s == 0 ? 0 : 1 + s seg, 1 + (s + 1) seg)
        d.nstripe = (d.ah + d.rh + 63) / 64;
        d.tile0 = tile0;
        tile0 += d.nseg * d.nstripe;
        const int nsv = (d.rh + 31) / 32;
        d.side_h = alloc((size_t)d.nseg * (d.ah + d.rh));
        d.tail_h = alloc((size_t)d.nseg * (d.ah + d.rh));
        d.side_v = alloc((size_t)nsv * d.w);
        d.tail_v = alloc((size_t)nsv * d.w);
        if (!d.side_h || !d.tail_h || !d.side_v || !d.tail_v) return false;
    }
    out.n_tiles = tile0;
    (void)c;
    return true;
}

// the segment-boundary arrays a launch leaves behind (checked in the prologue of the next launch, or by k_squeeze_check)
void checks_of(const SqueezeBatch& bt, std::vector<SqueezeCheck>& out) {
    out.clear();
    for (int i = 0; i < bt.n; i++) {
        const int nseg = squeeze_segments(bt.d[i]);
        if (nseg > 1) out.push_back(SqueezeCheck{bt.d[i].side, bt.d[i].tail, bt.d[i].other, nseg});
    }
}
void checks_of(const VHBatch& bt, std::vector<SqueezeCheck>& out) {
    out.clear();
    for (int i = 0; i < bt.n; i++) {
        const VHDesc& d = bt.d[i];
        if (d.nseg > 1) out.push_back(SqueezeCheck{d.side_h, d.tail_h, d.ah + d.rh, d.nseg});
        const int nsv = (d.rh + 31) / 32;
        if (nsv > 1) out.push_back(SqueezeCheck{d.side_v, d.tail_v, d.w, nsv});
    }
}
}  // namespace

extern "C" {

// ---- Modular ----------------------------------------------------------------------------------------
int32_t jxl_modular_default_squeeze_params(const int32_t* widths, const int32_t* heights, int32_t n_channels, int32_t nb_meta,
                                           jxl_squeeze_param* out, int32_t cap) {
    // ModularStream.java:110-131
    if (!widths || !heights || !out || n_channels < 0 || nb_meta < 0) return JXL_ERR_INVALID_ARGUMENT;
    int n = 0;
    const int first = nb_meta, count = n_channels - first;
    if (count <= 0) return 0;
    auto push = [&](int h, int ip, int b, int num) {
        if (n >= cap) return false;
        out[n++] = jxl_squeeze_param{h, ip, b, num};
        return true;
    };
    int sw = widths[0], sh = heights[0];
    if (count > 2 && sw == widths[first + 1] && sh == heights[first + 1]) {
        if (!push(1, 0, first + 1, 2) || !push(0, 0, first + 1, 2)) return JXL_ERR_INVALID_ARGUMENT;
    }
    if (sh >= sw && sh > 8) {
        if (!push(0, 1, first, count)) return JXL_ERR_INVALID_ARGUMENT;
        sh = (sh + 1) / 2;
    }
    while (sw > 8 || sh > 8) {
        if (sw > 8) {
            if (!push(1, 1, first, count)) return JXL_ERR_INVALID_ARGUMENT;
            sw = (sw + 1) / 2;
        }
        if (sh > 8) {
            if (!push(0, 1, first, count)) return JXL_ERR_INVALID_ARGUMENT;
            sh = (sh + 1) / 2;
        }
    }
    return n;
}

int32_t jxl_modular_squeezed_shapes(const int32_t* widths, const int32_t* heights, int32_t n_channels, const jxl_squeeze_param* sp,
                                    int32_t n_sp, int32_t* out_w, int32_t* out_h, int32_t cap) {
    // ModularStream.java:134-167
    if (!widths || !heights || !out_w || !out_h || n_channels < 0 || n_channels > cap || (n_sp > 0 && !sp)) return JXL_ERR_INVALID_ARGUMENT;
    std::vector<std::pair<int, int>> ch;  // (w, h)
    for (int i = 0; i < n_channels; i++) ch.emplace_back(widths[i], heights[i]);
    for (int j = 0; j < n_sp; j++) {
        const int begin = sp[j].begin_c, end = begin + sp[j].num_c - 1;
        if (begin < 0 || end >= (int)ch.size()) return JXL_ERR_INVALID_BITSTREAM;
        const int offset = sp[j].in_place ? end + 1 : (int)ch.size();
        for (int k = begin; k <= end; k++) {
            const int r = offset + k - begin;
            std::pair<int, int> res;
            if (sp[j].horizontal) {
                const int w = ch[k].first;
                ch[k].first = (w + 1) / 2;
                res = {w / 2, ch[k].second};
            } else {
                const int h = ch[k].second;
                ch[k].second = (h + 1) / 2;
                res = {ch[k].first, h / 2};
            }
            ch.insert(ch.begin() + r, res);
        }
    }
    if ((int)ch.size() > cap) return JXL_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < ch.size(); i++) {
        out_w[i] = ch[i].first;
        out_h[i] = ch[i].second;
    }
    return (int32_t)ch.size();
}

jxl_status jxl_modular_begin(jxl_ctx* c, const jxl_channel* chans, int32_t n_chans, const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin) {
    jxl_status st = bind(c);
    if (st) return st;
    if (n_chans < 0 || (n_chans > 0 && !chans) || (n_sp > 0 && !sp) || rct_type >= 42) return fail(c, JXL_ERR_INVALID_ARGUMENT, "modular: bad arguments");
    if (c->mod_flag_host) {  // reports of an earlier plan's speculative runs are void
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->n_aux > 0) HIP_TRY(c, hipStreamSynchronize(c->aux[0]));
        HIP_TRY(c, hipMemset(c->mod_flag.p, 0, sizeof(int32_t)));
        *c->mod_flag_host = 0;
    }
    c->mod_pending = false;
    for (auto& b : c->mod_bufs) b.release();
    c->mod_bufs.clear();
    c->mod_ops.clear();
    c->mod_ops_fused.clear();
    c->mod_out.clear();
    // Every plane starts kVhPad samples into its allocation (kVhPadH for a plane that is the residual of a horizontal step): the
    // fused squeeze kernel (k_modular_vh.hip) tiles the H pairs in chunks that start at pairs 1, 1 + CW, 1 + 2 CW, ..., and with
    // these leads the pieces it loads (V inputs from column c0 + 1, H residuals from pair c0) and the pieces it stores (outputs
    // from column 2 c0) all begin on 128-byte lines wherever a row's pitch is a multiple of 128 bytes -- the big levels.
    auto alloc_pad = [&](size_t n_elems, int pad) -> int32_t* {
        c->mod_bufs.emplace_back();
        if (!c->mod_bufs.back().ensure(4 * (std::max<size_t>(1, n_elems) + 64))) return nullptr;
        return c->mod_bufs.back().as<int32_t>() + pad;
    };
    auto alloc = [&](size_t n_elems) -> int32_t* { return alloc_pad(n_elems, kVhPad); };
    // which input channels are residuals of horizontal steps: a dry walk over the steps, as the loop below
    std::vector<uint8_t> is_hres((size_t)std::max(n_chans, 0), 0);
    {
        std::vector<int> idx((size_t)std::max(n_chans, 0));
        for (int i = 0; i < n_chans; i++) idx[i] = i;
        for (int j = n_sp - 1; j >= 0; j--) {
            const int begin = sp[j].begin_c, end = begin + sp[j].num_c - 1, n = (int)idx.size();
            const int offset = sp[j].in_place ? end + 1 : n + begin - end - 1;
            if (begin < 0 || end < begin || end >= n || offset < 0 || offset + (end - begin) >= n) break;  // (reported below)
            for (int k = begin; k <= end; k++) {
                const int r = offset + k - begin;
                if (sp[j].horizontal && idx[r] >= 0) is_hres[idx[r]] = 1;
                idx[k] = -1;  // the step's output
            }
            idx.erase(idx.begin() + offset, idx.begin() + offset + (end - begin + 1));
        }
    }
    std::vector<ModChan> ch;
    // r6: the fused V + H kernel carries only the short tendency() form (exact while |sample| < 2^23, modular_tend.h); a
    // plan whose inputs are outside that range -- high-bit-depth or float-as-int channels -- would run fused once, report, and be run
    // again step by step (mod_settle). A sparse scan of the uploaded channels (every 16th row: ~2 ms of host time for an 8K image)
    // picks the step-by-step plan up front for such inputs. Only a choice of plan: the device check is what guarantees the result.
    bool wide_samples = false;
    for (int i = 0; i < n_chans; i++) {
        if (chans[i].width < 0 || chans[i].height < 0) return fail(c, JXL_ERR_INVALID_ARGUMENT, "negative channel size");
        const size_t n = (size_t)chans[i].width * chans[i].height;
        if (n && !chans[i].data) return fail(c, JXL_ERR_INVALID_ARGUMENT, "channel %d has no data", i);
        for (int y = 0; y < chans[i].height && !wide_samples; y += 16) {
            const int32_t* row = chans[i].data + (size_t)y * chans[i].width;
            int32_t lo = 0, hi = 0;
            for (int x = 0; x < chans[i].width; x++) {
                lo = std::min(lo, row[x]);
                hi = std::max(hi, row[x]);
            }
            wide_samples = lo <= -kSqueezeSafeIn || hi >= kSqueezeSafeIn;
        }
        int32_t* d = alloc_pad(n, is_hres[i] ? kVhPadH : kVhPad);
        if (!d) return fail(c, JXL_ERR_OOM, "device allocation failed (modular channel)");
        if (n) HIP_TRY(c, hipMemcpy(d, chans[i].data, 4 * n, hipMemcpyHostToDevice));
        ch.push_back(ModChan{chans[i].width, chans[i].height, d, true});
    }
    // ModularStream.applyTransforms, SQUEEZE branch (ModularStream.java:229-254)
    for (int j = n_sp - 1; j >= 0; j--) {
        const int begin = sp[j].begin_c, end = begin + sp[j].num_c - 1;
        const int n = (int)ch.size();
        const int offset = sp[j].in_place ? end + 1 : n + begin - end - 1;
        if (begin < 0 || end < begin || end >= n || offset < 0 || offset + (end - begin) >= n)
            return fail(c, JXL_ERR_INVALID_BITSTREAM, "squeeze step %d addresses channels outside the list", j);
        ModOp batch{};
        batch.kind = 4;
        batch.bt.n = 0;
        batch.bt.horizontal = sp[j].horizontal ? 1 : 0;
        for (int k = begin; k <= end; k++) {
            const int r = offset + k - begin;
            const ModChan a = ch[k], re = ch[r];
            ModOp op{};
            ModChan o{};
            if (sp[j].horizontal) {
                if ((a.w != re.w && a.w != 1 + re.w) || re.h != a.h) return fail(c, JXL_ERR_INVALID_ARGUMENT, "Corrupted squeeze transform");
                o.w = a.w + re.w; o.h = a.h;
                op.kind = 0; op.adim = a.w; op.rdim = re.w; op.other = a.h;
            } else {
                if ((a.h != re.h && a.h != 1 + re.h) || re.w != a.w) return fail(c, JXL_ERR_STATE, "Corrupted squeeze transform");
                o.w = a.w; o.h = a.h + re.h;
                op.kind = 1; op.adim = a.h; op.rdim = re.h; op.other = a.w;
            }
            o.d = alloc((size_t)o.w * o.h);
            if (!o.d) return fail(c, JXL_ERR_OOM, "device allocation failed (squeeze output)");
            o.original = false;
            op.a = a.d; op.b = re.d; op.o = o.d;
            SqueezeDesc sd{a.d, re.d, o.d, op.adim, op.rdim, op.other, nullptr, 0, 0};
            // segment geometry by the size of the step (jxl_internal.h, squeeze_seg): short segments while one wave's walk,
            // not bandwidth, bounds the step
            static const int64_t short_max = getenv("JXL_SQUEEZE_SHORT_MAX") ? atoll(getenv("JXL_SQUEEZE_SHORT_MAX")) : ((int64_t)8 << 20);
            if ((int64_t)(end - begin + 1) * o.w * o.h <= short_max) {
                sd.seg = 32;
                sd.warm = 8;
            }
            if (op.rdim > squeeze_seg(sd) && !getenv("JXL_SQUEEZE_SERIAL")) {  // segmented walk: chain state per segment start
                sd.side = alloc((size_t)((op.rdim + squeeze_seg(sd) - 1) / squeeze_seg(sd)) * op.other);
                if (!sd.side) return fail(c, JXL_ERR_OOM, "device allocation failed (squeeze segment states)");
                if (!getenv("JXL_SQUEEZE_NO_TAIL")) {
                    sd.tail = alloc((size_t)((op.rdim + squeeze_seg(sd) - 1) / squeeze_seg(sd)) * op.other);
                    if (!sd.tail) return fail(c, JXL_ERR_OOM, "device allocation failed (squeeze segment states)");
                }
            }
            batch.bt.d[batch.bt.n++] = sd;
            if (batch.bt.n == 8) {
                c->mod_ops.push_back(batch);
                batch.bt.n = 0;
            }
            ch[k] = o;
        }
        if (batch.bt.n > 0) c->mod_ops.push_back(batch);
        ch.erase(ch.begin() + offset, ch.begin() + offset + (end - begin + 1));
    }
    // The leading run of small squeeze steps becomes one launch (k_squeeze_chain). A step joins the run while every channel of
    // it is small (<= 256 lanes, <= 40 pairs along the squeezed axis) and slot i takes its averages from slot i of the step before (or
    // from an input channel) and its residuals from an input channel: then one workgroup per slot needs no other workgroup.
    if (!getenv("JXL_SQUEEZE_NO_CHAIN")) {
        size_t run = 0;
        int slots = 0;
        while (run < c->mod_ops.size()) {
            const ModOp& op = c->mod_ops[run];
            if (op.kind != 4 || op.bt.n > 8) break;
            bool ok = true;
            for (int i = 0; i < op.bt.n && ok; i++) {
                const SqueezeDesc& d = op.bt.d[i];
                // (one workgroup walks a channel's rows / columns serially, ~100 cycles per pair: worth a launch boundary (~7 us)
                // only while the axis is short. Measured on the 1080p plan: 11 steps in the chain 0.277 ms, none 0.247 ms)
                static const int chain_max = getenv("JXL_SQUEEZE_CHAIN_MAX") ? atoi(getenv("JXL_SQUEEZE_CHAIN_MAX")) : 40;
                ok = d.other <= 256 && d.rdim <= chain_max;
                // averages: slot i of the previous step, or not an output of any earlier step of the run
                for (size_t q = 0; q < run && ok; q++)
                    for (int j = 0; j < c->mod_ops[q].bt.n && ok; j++) {
                        const int32_t* o = c->mod_ops[q].bt.d[j].o;
                        if (d.b == o) ok = false;
                        if (d.a == o && !(q + 1 == run && j == i)) ok = false;
                    }
            }
            if (!ok) break;
            slots = std::max(slots, op.bt.n);
            run++;
        }
        // r5: a V step whose H partner would stay outside the run pairs with it in the fused kernel instead
        if (run >= 1 && run < c->mod_ops.size() && !getenv("JXL_SQUEEZE_NO_VH") && squeeze_pair_fusable(c->mod_ops[run - 1], c->mod_ops[run])) run--;
        if (run >= 2) {
            std::vector<SqueezeBatch> steps;
            for (size_t q = 0; q < run; q++) {
                SqueezeBatch bt = c->mod_ops[q].bt;
                for (int i = 0; i < bt.n; i++) bt.d[i].side = bt.d[i].tail = nullptr;  // full serial walks: no segment states
                steps.push_back(bt);
            }
            c->mod_bufs.emplace_back();
            if (!c->mod_bufs.back().ensure(sizeof(SqueezeBatch) * steps.size())) return fail(c, JXL_ERR_OOM, "device allocation failed (squeeze chain)");
            HIP_TRY(c, hipMemcpy(c->mod_bufs.back().p, steps.data(), sizeof(SqueezeBatch) * steps.size(), hipMemcpyHostToDevice));
            ModOp chain{};
            chain.kind = 5;
            chain.chain_dev = c->mod_bufs.back().as<SqueezeBatch>();
            chain.chain_steps = (int)run;
            chain.chain_slots = slots;
            c->mod_ops.erase(c->mod_ops.begin(), c->mod_ops.begin() + run);
            c->mod_ops.insert(c->mod_ops.begin(), chain);
        }
    }
    if (rct_type >= 0) {  // RCT branch (:255-326)
        if (rct_begin < 0 || rct_begin + 2 >= (int)ch.size()) return fail(c, JXL_ERR_INVALID_ARGUMENT, "rct channels out of range");
        ModChan* v = &ch[rct_begin];
        if (v[1].w != v[0].w || v[1].h != v[0].h || v[2].w != v[1].w || v[2].h != v[1].h)
            return fail(c, JXL_ERR_INVALID_BITSTREAM, "RCT must be performed on three equal size channels");
        const int64_t n = (int64_t)v[0].w * v[0].h;
        for (int j = 0; j < 3; j++) {
            if (v[j].original) {  // keep the uploaded input intact so that run() is repeatable
                int32_t* d = alloc((size_t)n);
                if (!d) return fail(c, JXL_ERR_OOM, "device allocation failed (rct copy)");
                ModOp cp{};
                cp.kind = 3; cp.a = v[j].d; cp.o = d; cp.n = n;
                c->mod_ops.push_back(cp);
                v[j].d = d;
                v[j].original = false;
            }
        }
        ModOp op{};
        op.kind = 2; op.v0 = v[0].d; op.v1 = v[1].d; op.v2 = v[2].d; op.n = n; op.type = rct_type % 7;
        c->mod_ops.push_back(op);
        const int perm = rct_type / 7;
        ModChan t[3] = {v[0], v[1], v[2]};
        for (int j = 0; j < 3; j++) v[kPermutationLut[perm][j]] = t[j];
    }
    c->mod_out = ch;
    // r5: every V step followed by the H step of the same channels becomes one launch (k_modular_vh.hip)
    c->mod_ops_fused.clear();
    if (!getenv("JXL_SQUEEZE_NO_VH") && !wide_samples) {
        bool any = false;
        for (size_t i = 0; i < c->mod_ops.size();) {
            if (i + 1 < c->mod_ops.size() && squeeze_pair_fusable(c->mod_ops[i], c->mod_ops[i + 1])) {
                ModOp f{};
                f.kind = 6;
                if (!make_vh(c, c->mod_ops[i].bt, c->mod_ops[i + 1].bt, f.vh, alloc)) return fail(c, JXL_ERR_OOM, "device allocation failed (fused squeeze states)");
                c->mod_ops_fused.push_back(f);
                any = true;
                i += 2;
            } else {
                c->mod_ops_fused.push_back(c->mod_ops[i++]);
            }
        }
        if (!any) c->mod_ops_fused.clear();
    }
    return JXL_OK;
}

}  // extern "C"

namespace {
// The plan of jxl_modular_begin. speculative: the verification of every segmented step runs on the side stream beside the
// steps that follow and only REPORTS a mismatch (SqueezeBatch::flag); the steps themselves stay back to back on the main
// stream -- a step is one launch on the critical path instead of two (1080p: 11 of 23 launches were verifications, 29 % of
// the time). A mismatch is rare (the recurrence forgets its start within a few pairs, jxl_internal.h) and costs a second,
// in-order run (mod_settle). RCT works in place on squeeze outputs the checks still read: it waits for them.
// mode 0: every segmented step is checked (and repaired) by its own launch before the next one starts;
// mode 1: the checks run on the side stream and only report (the r2 experiment);
// mode 2 (r4, the default): the check of a step rides in the prologue of the NEXT step's walk launch and only reports.
jxl_status run_modular_plan(jxl_ctx* c, int mode) {
    int launches = 0;
    const bool speculative = mode == 1;
    hipStream_t s = c->stream, vs = speculative ? c->aux[0] : nullptr;
    bool checks_out = false;
    auto join = [&]() {
        if (!checks_out) return;
        (void)hipEventRecord(c->mod_join, vs);
        (void)hipStreamWaitEvent(s, c->mod_join, 0);
        checks_out = false;
    };
    // mode 2: the segment-boundary arrays of the last launch, not compared yet
    std::vector<SqueezeCheck> pending, mine;
    int32_t* flag = mode == 2 ? c->mod_flag.as<int32_t>() : nullptr;
    auto flush_pending = [&]() {  // nothing follows that could carry the check: a (tiny) report-only launch of its own
        if (pending.empty()) return;
        launch_squeeze_check(pending.data(), (int)pending.size(), flag, s);
        launches += ((int)pending.size() + 3) / 4;
        pending.clear();
    };
    auto take_pending = [&](int& n_chk, SqueezeCheck* chk) {
        if ((int)pending.size() > kSqueezeMaxChecks) flush_pending();
        n_chk = (int)pending.size();
        for (int i = 0; i < n_chk; i++) chk[i] = pending[i];
        pending.clear();
    };
    const std::vector<ModOp>& ops = (mode == 2 && !c->mod_ops_fused.empty()) ? c->mod_ops_fused : c->mod_ops;
    for (const ModOp& op : ops) {
        if (mode == 2 && op.kind != 4 && op.kind != 6) flush_pending();
        switch (op.kind) {
        case 0: launch_inv_hsqueeze(op.a, op.adim, op.b, op.rdim, op.other, op.o, s); break;
        case 1: launch_inv_vsqueeze(op.a, op.adim, op.b, op.rdim, op.other, op.o, s); break;
        case 2: join(); launch_rct(op.v0, op.v1, op.v2, op.n, op.type, s); break;
        case 3: (void)hipMemcpyAsync(op.o, op.a, 4 * (size_t)op.n, hipMemcpyDeviceToDevice, s); break;
        case 4: {
            SqueezeBatch bt = op.bt;
            if (mode == 2) {
                bt.flag = flag;
                take_pending(bt.n_chk, bt.chk);
                launch_squeeze_walk(bt, s);
                if (squeeze_can_fuse_check(bt)) {
                    checks_of(bt, pending);
                } else {
                    bt.flag = nullptr;  // (no compact tails: JXL_SQUEEZE_NO_TAIL) check and repair in place, as mode 0
                    bt.n_chk = 0;
                    launch_squeeze_verify(bt, s);
                }
                break;
            }
            bt.flag = speculative ? c->mod_flag.as<int32_t>() : nullptr;
            launch_squeeze_batch(bt, s, vs, c->mod_ev);
            checks_out = checks_out || speculative;
            break;
        }
        case 5: launch_squeeze_chain(op.chain_dev, op.chain_steps, op.chain_slots, s); break;
        case 6: {  // (mode 2 only)
            VHBatch bt = op.vh;
            bt.flag = flag;
            take_pending(bt.n_chk, bt.chk);
            launch_squeeze_vh(bt, s);
            checks_of(bt, pending);
            break;
        }
        }
        launches++;
    }
    if (mode == 2) flush_pending();
    if (mode != 0) {
        join();
        (void)hipMemcpyAsync(c->mod_flag_host, c->mod_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        c->mod_pending = true;
    }
    c->mod_launches = launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, JXL_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return JXL_OK;
}

// before anything reads the plan's outputs: wait, look at the report of the speculative run(s), redo in order if needed
jxl_status mod_settle(jxl_ctx* c) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->mod_pending) return JXL_OK;
    c->mod_pending = false;
    if (*c->mod_flag_host == 0) return JXL_OK;
    *c->mod_flag_host = 0;
    HIP_TRY(c, hipMemset(c->mod_flag.p, 0, sizeof(int32_t)));
    c->mod_redos++;
    // a plan whose data trip the fused kernel (chains that do not forget their start, or samples outside the short tendency form's
    // range guard) would pay for two runs every time: from here on this plan runs its one-step launches (still checked in flight)
    c->mod_ops_fused.clear();
    const jxl_status st = run_modular_plan(c, 0);
    if (st) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JXL_OK;
}
}  // namespace

extern "C" {

jxl_status jxl_modular_run(jxl_ctx* c) {
    jxl_status st = bind(c);
    if (st) return st;
    // JXL_SQUEEZE_SPECULATE (read per call): unset / 2 = the check of every segmented step inside the next step's walk launch
    // (r4: 13 launches per 1080p image instead of 23); 0 = a check-and-repair launch behind every step (the r1 form, and what a
    // reported mismatch falls back to); 1 = the checks on the side stream (r2: measured slower, 0.235 against 0.205 ms -- the
    // event pair per step costs more than the launch it moves away).
    const char* se = getenv("JXL_SQUEEZE_SPECULATE");
    int mode = se ? atoi(se) : 2;
    if (mode < 0 || mode > 2) mode = 2;
    if (mode == 1 && c->n_aux <= 0) mode = 0;
    if (mode != 0 && !c->mod_flag_host) {
        if (!c->mod_flag.ensure(sizeof(int32_t))) return fail(c, JXL_ERR_OOM, "device allocation failed");
        HIP_TRY(c, hipMemset(c->mod_flag.p, 0, sizeof(int32_t)));
        HIP_TRY(c, hipHostMalloc((void**)&c->mod_flag_host, sizeof(int32_t), hipHostMallocDefault));
        *c->mod_flag_host = 0;
        HIP_TRY(c, hipEventCreateWithFlags(&c->mod_ev, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->mod_join, hipEventDisableTiming));
    }
    return run_modular_plan(c, mode);
}

int32_t jxl_modular_redo_count(const jxl_ctx* c) { return c ? c->mod_redos : 0; }

int32_t jxl_modular_out_count(const jxl_ctx* c) { return c ? (int32_t)c->mod_out.size() : 0; }
int32_t jxl_modular_last_launch_count(const jxl_ctx* c) { return c ? c->mod_launches : 0; }

jxl_status jxl_modular_out_shape(const jxl_ctx* c, int32_t idx, int32_t* w, int32_t* h) {
    if (!c || idx < 0 || idx >= (int)c->mod_out.size() || !w || !h) return JXL_ERR_INVALID_ARGUMENT;
    *w = c->mod_out[idx].w;
    *h = c->mod_out[idx].h;
    return JXL_OK;
}

jxl_status jxl_modular_read_channel(jxl_ctx* c, int32_t idx, int32_t* dst) {
    jxl_status st = bind(c);
    if (st) return st;
    if (idx < 0 || idx >= (int)c->mod_out.size() || !dst) return fail(c, JXL_ERR_INVALID_ARGUMENT, "bad channel index");
    if ((st = mod_settle(c))) return st;
    if ((st = finish(c))) return st;
    const size_t n = (size_t)c->mod_out[idx].w * c->mod_out[idx].h;
    if (n) HIP_TRY(c, hipMemcpy(dst, c->mod_out[idx].d, 4 * n, hipMemcpyDeviceToHost));
    return JXL_OK;
}

jxl_status jxl_modular_apply(jxl_ctx* c, const jxl_channel* chans, int32_t n_chans, const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin, jxl_channel* out, int32_t n_out) {
    jxl_status st = jxl_modular_begin(c, chans, n_chans, sp, n_sp, rct_type, rct_begin);
    if (st) return st;
    if (n_out != (int)c->mod_out.size() || (n_out > 0 && !out)) return fail(c, JXL_ERR_INVALID_ARGUMENT, "expected %d output channels", (int)c->mod_out.size());
    for (int i = 0; i < n_out; i++)
        if (out[i].width != c->mod_out[i].w || out[i].height != c->mod_out[i].h)
            return fail(c, JXL_ERR_INVALID_ARGUMENT, "output channel %d must be %dx%d", i, c->mod_out[i].w, c->mod_out[i].h);
    if ((st = jxl_modular_run(c))) return st;
    for (int i = 0; i < n_out; i++)
        if ((st = jxl_modular_read_channel(c, i, out[i].data))) return st;
    return JXL_OK;
}

}  // extern "C"
