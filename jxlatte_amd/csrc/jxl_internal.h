// Internal declarations shared by the HIP translation units of libjxlatte_amd.so.
// gfx950 (MI355X) only. All device arithmetic is strict IEEE f32: the library is built with
// -ffp-contract=off (no FMA contraction), default correctly-rounded f32 division, f32
// denormals preserved -- required for bit-exact parity with the Java reference.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include "../../include/jxlatte_amd.h"
#include "../../include/jxl_transform_types.h"

namespace jxl {

// total floats of the cosine LUT for sizes 1..256: sum (s-1)*s
constexpr int kLutTotal = 86870;

// One varblock, frame coordinates. 16 bytes (one dwordx4 load).
struct DevBlock {
    uint16_t cy, cx;    // top-left cell (8x8 px units) in the frame
    uint32_t type;      // TransformType.type
    uint32_t cfl_zero;  // bit (ty*5+tx): CfL factors of that 64x64 tile (relative to the block's first
                        // tile) read as 0 for this block -- the reference's per-group xFactors cache has
                        // not been filled yet when this block is visited (HFCoefficients.java:159-181)
    int32_t hf_mul;     // HFMetadata.hfMultiplier of the block's first cell: carried here so that the kernels do not
                        // chain a second dependent load behind the block record
};

// One workgroup's share of a type-uniform launch.
static_assert(sizeof(DevBlock) == 16, "one dwordx4 per block record");

// One workgroup's share of a type-uniform launch.
struct WorkItem {
    uint32_t type;
    uint32_t first;  // index into the binned DevBlock array
    uint32_t count;  // blocks handled by this workgroup
};

// Device-resident view of one VarDCT frame (all pointers are device pointers).
struct DevFrame {
    int32_t no_cfl;         // chroma-subsampled frame: chroma-from-luma is skipped (HFCoefficients.java:149-151) and the
                            // geometry below is the one of the channel the launch handles
    int32_t width, height;  // padded px
    int32_t bw, bh;         // cells
    int32_t tw, th;         // 64x64 tiles
    const int32_t* coeff[3];
    const float* lf[3];     // [bh][bw]
    const float* llf[3];    // [bh][bw]: lf with the LLF coefficients of blocks larger than 8x8 (k_llf)
    const int32_t* hf_mul;  // [bh][bw]
    const int32_t* sharpness;
    const float* kx_tab;  // [th][tw]: baseCorrelationX + xFromY / colorFactor per 64x64 tile (HFCoefficients.java:177-181),
    const float* kb_tab;  //           computed once per frame on the host (same IEEE float division and addition)
    const float* weights;     // flat, reciprocal
    const float* weights_t;   // same offsets, every matrix transposed (for flip() blocks: coalesced reads)
    int32_t woffs[51];
    const float* lut;         // cosine LUT, all sizes; size s=1<<l starts at lut_off(l)
    float scale_factor[3];
    float quant_bias[3];
    float quant_bias_numerator;
    float base_corr_x, base_corr_b;
    float color_factor_f;     // (float)colorFactor
};

// Layout of the quantised-coefficient planes on the device (r4): tiled by 8x8 cell. The 64 samples of cell (by, bx) of a
// plane W samples wide are consecutive, row-major, at ((by * (W / 8) + bx) * 64 -- a varblock's rows then share cache lines
// (an 8x8 block is two whole 128-byte lines instead of eight quarter lines), which is what the launches that handle one
// transform type at a time need: in a raster plane they fetched 1.5-3.7 times the bytes they used (profiles/r4_rocprof_summary).
// The C ABI still speaks raster planes; the writers (k_store2d_tiled / k_widen2d* in host.hip) do the re-tiling. Runs of up to
// 8 samples that start at a multiple of 4 (8) stay inside one cell row, so 16- and 32-byte loads are unaffected.
__host__ __device__ __forceinline__ int64_t coeff_off(int W, int py, int px) {
    return (((int64_t)(py >> 3) * (W >> 3) + (px >> 3)) << 6) + (((py & 7) << 3) | (px & 7));
}

__host__ __device__ inline int lut_off(int l) {
    // sum_{j<l} (2^j - 1) * 2^j
    int o = 0;
    for (int j = 0; j < l; j++) o += ((1 << j) - 1) << j;
    return o;
}

struct EpfParams {
    float channel_scale[3];
    float sigma_scale;   // stepMultiplier * pass scale for this iteration
    float border_sad_mul;
    float inv_sigma_modular;
};

struct XybParams {
    float sm[9];         // matrix * itScale
    float ob[3];         // opsinBias
    float cob[3];        // -cbrtOpsinBias
};

// Argument block of a merged IDCT launch (k_idct_multi): a few type-uniform segments of work items, in launch order.
struct MultiArgs {
    static constexpr int kMaxSeg = 12;
    DevFrame f;
    const DevBlock* blocks;
    const WorkItem* items;
    int n_seg;
    int seg_b0[kMaxSeg + 1];  // first workgroup of segment k (a multiple of 8); seg_b0[n_seg] = grid size
    int seg_n[kMaxSeg];       // items in segment k
    int seg_type[kMaxSeg];    // TransformType.type of segment k
    int seg_first[kMaxSeg];   // first block of the type in `blocks`
    int seg_nblocks[kMaxSeg]; // blocks of the type
    int nch, ch0;             // channels per block group (3, or 1 for a chroma-subsampled frame's per-channel launch), first one
    float *o0, *o1, *o2;
};
// work items are implicit: item i of a segment = channel ch0 + i % nch of block group i / nch, a group being
// medium_blocks_per_wg(type) consecutive blocks (no item table to load before the block records)
struct IdctSegment { int type, first_block, n_blocks; };

// Argument block of the persistent three-channel IDCT launch (k_idct_wg3.hip): type-uniform segments in launch order
struct Wg3Seg { int type, first_block, n_blocks, item_base; };  // item_base: index of the segment's first work item in the launch
struct Wg3Args {
    static constexpr int kMaxSeg = 21;
    DevFrame f;
    const DevBlock* blocks;
    float *o0, *o1, *o2;
    int n_seg, total_items;
    int img_floats;  // floats of the largest three-channel LDS image among the segments' types (the tables follow it)
    int llf_in_item;  // 1: the items transform their blocks' LF patches themselves (finalizeLLF); 0: k_llf_wg3 ran before (llf planes)
    // the item list: 32-byte records {type, first_block, n_blocks, geometry word (wg3_geo), weight offsets of the three channels, 0},
    // total_items of them, in the order the workgroups take them (wg3_item_table: spatial, dealt to the XCDs, cost-balanced)
    const int* items;
    Wg3Seg seg[kMaxSeg];
};

// the item list of one class's segments in spatial order, dealt to the XCDs in runs (host side)
void wg3_item_table(const DevBlock* host_blocks, int frame_bw, const IdctSegment* segs, int n_seg, int which, const int32_t* woffs, bool spatial,
                    std::vector<int>& out, int grid);
int wg3_grid_cap(bool big);  // workgroups of a single-frame launch (JXL_WG3_GRID / JXL_WG3_GRID_BIG)
bool wg3_handles(int type);
bool wg3_special_items();  // r6: the special 8x8 types are items of the persistent launch
bool wg3_big(int type);  // the 64-point family: its own launch (register / LDS class)
bool wg3_llf_in_item();  // finalizeLLF inside the k_idct_wg3 items (default) or as a launch of its own writing the llf planes
int build_wg3_args(const DevFrame& f, const DevBlock* blocks, const IdctSegment* segs, int n_seg, int which, float* const out[3],
                   Wg3Args& a);
void launch_llf_wg3(const Wg3Args& a, float* const llf[3], hipStream_t s);
void launch_idct_wg3(const Wg3Args& a, bool big, int grid_cap, hipStream_t s);
int64_t wg3_llf_count(const Wg3Args& a);
size_t wg3_lds_bytes(const Wg3Args& a);
void launch_llf_wg3_batch(const Wg3Args* dev_args, int n_frames, int64_t max_llf, hipStream_t s);
void launch_idct_wg3_batch(const Wg3Args* dev_args, int n_frames, bool big, int grid_x, size_t lds, hipStream_t s);


// ---- launchers (defined in the kernel TUs) ---------------------------------------------------
// One launch per register class (0: every type up to 32x32, 1: the 64-point family), 256-thread workgroups.
// WorkItem.type = TransformType.type | channel << 8; an item covers up to medium_blocks_per_wg(type) blocks of that channel.
int idct_class_of(int type);
void launch_idct_multi(const DevFrame& f, const DevBlock* blocks, int cls, const IdctSegment* segs, int n_seg, int nch, int ch0,
                       float* const out[3], hipStream_t s);
int build_idct_multi_args(const DevFrame& f, const DevBlock* blocks, const IdctSegment* segs, int n_seg, int nch, int ch0,
                          float* const out[3], MultiArgs& a, size_t* lds_bytes_out);
// batch of frames: one MultiArgs block per frame in device memory (blockIdx.y); see k_idct_multi_batch
void launch_idct_multi_batch(const MultiArgs* dev_args, int n_frames, int grid_x, size_t lds_bytes, int cls, hipStream_t s);
void launch_idct_special_batch(const MultiArgs* dev_args, int n_frames, int max_items, hipStream_t s);
// the special 8x8-footprint types: items of up to 64 blocks (one per lane)
void launch_idct_special(const DevFrame& f, const DevBlock* blocks, const WorkItem* items, int n_items, float* const out[3],
                         hipStream_t s, bool wg_items);
int medium_blocks_per_wg(int type);
// finalizeLLF of blocks[first..first+count) (all larger than 8x8) into the llf planes
void launch_llf(const DevFrame& f, const DevBlock* blocks, int first, int count, float* const llf[3], hipStream_t s);
// large (128/256-edge) blocks: `first..first+count` of `blocks`; scratch planes same shape as out
void launch_idct_large(const DevFrame& f, const DevBlock* blocks, const DevBlock* host_blocks, int first, int count,
                       float* const out[3], float* const scratch[3], hipStream_t s, int* n_launches);
void launch_accumulate(int32_t* dst, const int32_t* src, int64_t n, hipStream_t s);

void launch_gab(const float* const in[3], float* const out[3], int h, int w, const float w1[3], const float w2[3],
                hipStream_t s);
void launch_epf_sigma(const int32_t* hf_mul, const int32_t* sharpness, int bh, int bw, float global_scale_f,
                      const float sharp_lut[8], float* inv_sigma, int* bad_flag, hipStream_t s);
void launch_epf_iter(const float* const in[3], float* const out[3], int h, int w, int iter, const float* inv_sigma,
                     const EpfParams& p, hipStream_t s);
void launch_xyb(float* const planes[3], int64_t n, const XybParams& p, hipStream_t s);
void launch_ycbcr(float* const planes[3], int64_t n, hipStream_t s);
// transfer + quantise: out elem size 4 (float or int32), 2 (u16), 1 (u8)
// out index = i * out_pitch + out_off (pitch 1 = planar; pitch 3, off c = pixel-interleaved)
// pq_tab: the PQ segment table of build_pq_table (device; null: the double-precision form)
void launch_transfer(const float* in, int64_t n, int transfer, int max_value, void* out, int out_elem, hipStream_t s,
                     int out_pitch = 1, int out_off = 0, const float* pq_tab = nullptr, const float* srgb8_tab = nullptr,
                     const float* pq16_thr = nullptr, const float* srgb16_tab = nullptr);
// PQ as a table of quadratic segments (jxl_fastpow.h): kPqTableFloats floats = float4 {a0 hi, a0 lo, a1, a2} per segment
constexpr int kPqTableFloats = (129 - 87) * 128 * 4;
void build_pq_table(float* out /* [kPqTableFloats] */);
constexpr int kSrgb8TableFloats = (127 - 118) * 128 * 4;
bool build_srgb8_table(float* out /* [kSrgb8TableFloats] */);
void build_pq16_thresholds(float* out /* [65537] */);
void build_pq8_thresholds(float* out /* [257] */);
void build_srgb16_table(float* out /* [kSrgb8TableFloats + 65537]: segments, then thresholds */);  // false: a segment with more than three thresholds (never)
// fused restoration + colour tile kernel (Gab -> EPF iters -> XYB -> optional transfer/quantise)
struct RestoreParams {
    int gab, epf_iters, xyb, transfer, max_value, out_elem;
    int interleaved;  // JXL_OUT_RGB8 / RGB16: out[0] holds R,G,B per pixel
    float gab_base[3], gab_adj[3], gab_diag[3];
    EpfParams epf[3];  // per iteration index 0..2
    XybParams xybp;
    float global_scale_f;
    float sharp_lut[8];
    const float* pq_tab;  // device: PQ segment table (null: double-precision PQ)
    const float* srgb8_tab;  // device: sRGB -> 8-bit threshold table (fp_srgb8; null: double-precision pow + quantise)
    const float* pq16_thr;   // device: the 65537 thresholds of PQ -> 16 bit (fp_pq16), then the 257 of PQ -> 8 bit (fp_pq8);
                             // null: table / f64 float result, then quantise
    const float* srgb16_tab;  // device: quadratic segments of the sRGB curve over [2^-9, 1) + (at float offset kSrgb8TableFloats) the
                              // 65537 thresholds of sRGB -> 16 bit (fp_srgb16; null: double-precision pow + quantise)
};
// argument block of the fused kernel (one per frame; an array of them for the batched launch)
struct FusedArgs {
    const float* in[3];
    void* out[3];
    const int32_t* hf_mul;
    const int32_t* sharpness;
    int W, H, bw;
    RestoreParams p;
};
// false if the configuration is not covered by the fused kernel
bool fill_restore_fused_args(const float* const in[3], void* const out[3], int h, int w, const int32_t* hf_mul,
                             const int32_t* sharpness, const RestoreParams& p, FusedArgs& a);
// (stage timing) the events the next single-frame launch of the fused restoration kernel records its own start / stop into; else null
extern thread_local hipEvent_t g_restore_kernel_ev[2];
int restore_fused_variant(const FusedArgs& a);  // which kernel instantiation the arguments select
// the non-float sinks of the fused kernel (k_restore_fused_gen.hip, k_restore_fused_q.hip): one frame (`single`) or a batch
void launch_restore_fused_gen(const FusedArgs* single, const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s);
void launch_restore_fused_q(int sink_kind, const FusedArgs* single, const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s);
void launch_restore_fused_batch(const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s);
// returns false if the configuration is not covered by the fused kernel (caller falls back to stage kernels)
bool launch_restore_fused(const float* const in[3], void* const out[3], int h, int w, const int32_t* hf_mul,
                          const int32_t* sharpness, const RestoreParams& p, hipStream_t s);

// one inverse squeeze step over up to 8 channels in a single launch
struct SqueezeDesc {
    const int32_t* a;  // averages
    const int32_t* b;  // residuals
    int32_t* o;        // output
    int adim, rdim;    // squeezed-axis length of a and b (widths for H, heights for V)
    int other;         // the other dimension (rows for H, columns for V)
    int32_t* side;     // [nseg][other] chain state at every segment start (segmented mode), or null: one segment
    int seg, warm;     // pairs per segment and warm-up pairs in front of it (multiples of 8); 0: kSqueezeSeg / kSqueezeWarm
    int32_t* tail;     // [nseg][other] last output of every segment but the last (what side[s + 1] is checked against), or null:
                       // the check reads it from the output plane (for H steps that is one cache line per row and segment)
};
__host__ __device__ inline int squeeze_seg(const SqueezeDesc& d);
__host__ __device__ inline int squeeze_warm(const SqueezeDesc& d);
// The recurrence (left = previously output odd sample) is serial along the squeezed axis, but it forgets its start
// within a few pairs (tendency() is clamped to [0, 2(avg - nextAvg)], so every step maps all inputs to a handful of
// outputs: measured 1-2 pairs on average, 6 at most on photographic, synthetic and white-noise rows). So the axis is cut
// into segments of kSqueezeSeg pairs; the wave of segment s > 0 starts kSqueezeWarm pairs early from a guessed state,
// stores nothing for those, and records the state it has reached at its segment start in `side`. A second, tiny
// kernel compares that state with the true one (the last output of segment s-1); every segment of a row / column whose
// boundaries all match is exact by induction, and a row / column with a mismatch (constructible, not observed) is redone
// serially from its first bad boundary. Results are bit-identical to the serial walk in every case.
#ifndef JXL_SQUEEZE_SEG
#define JXL_SQUEEZE_SEG 64
#endif
#ifndef JXL_SQUEEZE_WARM
#define JXL_SQUEEZE_WARM 16
#endif
constexpr int kSqueezeSeg = JXL_SQUEEZE_SEG;
constexpr int kSqueezeWarm = JXL_SQUEEZE_WARM;
// the segment-boundary arrays of an EARLIER step, checked in the prologue of a later step's walk kernel (r4)
struct SqueezeCheck {
    const int32_t* side;  // [nseg][n]
    const int32_t* tail;  // [nseg - 1][n] (rows 0 .. nseg-2 used)
    int n, nseg;
};
struct SqueezeBatch {
    int n;
    int horizontal;
    SqueezeDesc d[8];
    // k_squeeze_verify: nullptr = repair a mismatching row / column in place (the step is then exact before the next one
    // starts); else = only report (atomicOr 1): the check runs beside / behind the following steps and the host redoes the plan
    int32_t* flag;
    // r4: the verification of the PREVIOUS segmented step rides in this step's walk launch -- every workgroup compares a slice
    // of that step's compact side / tail arrays before it starts its own walk and reports a mismatch in `flag`; a 1080p plan is
    // then 13 launches instead of 23 (the check launches were 6 us each on the chain of dependent launches that bounds it)
    int n_chk;
    SqueezeCheck chk[16];
};
constexpr int kSqueezeMaxChecks = 16;

// r5: a vertical step and the horizontal step that follows it on the same channels (ModularStream.java:229-254 applies them back
// to back: V_k, H_k produce level k of the squeeze pyramid from level k + 1) as ONE launch -- the V output lives in LDS only.
// One wave = one tile: 64 output rows (a "stripe": 32 V pairs) x one H segment of `seg` pairs, walked left to right in chunks of
// 16 H pairs (k_modular_vh.hip). Guessed states, all verified like the segmented walk's (SqueezeCheck): the H chain at a segment
// start (side_h / tail_h, [nseg][rows]), the V chain at a stripe start (side_v / tail_v, [nstripe][columns]), and the V chain at
// the three quarter boundaries inside a stripe (compared inside the wave, reported in VHBatch::flag).
constexpr int kVhPad = 30, kVhPadH = 31;  // samples in front of a plane / of a horizontal step's residual plane (host.hip, jxl_modular_begin)
struct VHDesc {
    const int32_t* va;  // V averages  [ah][w]
    const int32_t* vb;  // V residuals [rh][w]
    const int32_t* hb;  // H residuals [ah + rh][rw]
    int32_t* o;         // output      [ah + rh][w + rw]
    int w, ah, rh, rw;  // rh >= 1, rw >= 1; ah - rh and w - rw are 0 or 1
    int seg;            // H pairs per segment (a multiple of VHBatch::cw)
    int nseg, nstripe;  // max(1, ceil((rw - 1) / seg)), ceil((ah + rh) / 64)
    int tile0;          // index of this channel's first tile in the launch (tiles: [stripe][segment])
    int32_t *side_h, *tail_h;  // [nseg][ah + rh]
    int32_t *side_v, *tail_v;  // [ceil(rh / 32)][w]
};
struct VHBatch {
    int n, n_tiles;
    int cw;         // H pairs per chunk: 16 or 32 (the kernel instantiation; VHDesc::seg is a multiple of it)
    VHDesc d[8];
    int32_t* flag;  // report a mismatch (atomicOr 1); never null
    int n_chk;      // segment-boundary arrays of an EARLIER step, checked in this launch's prologue
    SqueezeCheck chk[kSqueezeMaxChecks];
};
void launch_squeeze_vh(const VHBatch& bt, hipStream_t s);
// parallel report-only comparison of segment-boundary arrays (the last step of a plan has no later launch to ride in)
void launch_squeeze_check(const SqueezeCheck* chk, int n_chk, int32_t* flag, hipStream_t s);
// Small steps are bound by the time ONE wave needs for its segment (~130 cycles per pair), large ones by bandwidth: the host
// gives steps of up to 8 Mi samples 32-pair segments with an 8-pair warm-up (measured: 1080p image 0.245 -> 0.195 ms) and
// keeps 64 + 16 beyond (8K image: 0.78 ms against 0.80 ms with 32 + 8 everywhere).
__host__ __device__ inline int squeeze_seg(const SqueezeDesc& d) { return d.seg > 0 ? d.seg : kSqueezeSeg; }
__host__ __device__ inline int squeeze_warm(const SqueezeDesc& d) { return d.seg > 0 ? d.warm : kSqueezeWarm; }
__host__ __device__ inline int squeeze_segments(const SqueezeDesc& d) {
    return d.side && d.rdim > squeeze_seg(d) ? (d.rdim + squeeze_seg(d) - 1) / squeeze_seg(d) : 1;
}
// check_stream / ev: with bt.flag set, the verification launch goes to check_stream behind an event recorded on s
void launch_squeeze_batch(const SqueezeBatch& bt, hipStream_t s, hipStream_t check_stream = nullptr, hipEvent_t ev = nullptr);
// the two halves on their own: the walk (with the fused check of an earlier step, SqueezeBatch::chk) and the verification;
// squeeze_can_fuse_check: the step is segmented and all its channels keep compact tail arrays
void launch_squeeze_walk(const SqueezeBatch& bt, hipStream_t s);
void launch_squeeze_verify(const SqueezeBatch& bt, hipStream_t s);
bool squeeze_can_fuse_check(const SqueezeBatch& bt);
// a run of small steps in one launch: dev_steps = the steps' SqueezeBatch blocks in device memory, slot i of every step by
// workgroup i (the caller has checked that slot i of a step depends on slot i of the step before only)
void launch_squeeze_chain(const SqueezeBatch* dev_steps, int n_steps, int n_slots, hipStream_t s);
void launch_inv_hsqueeze(const int32_t* avg, int aw, const int32_t* res, int rw, int h, int32_t* out, hipStream_t s);
void launch_inv_vsqueeze(const int32_t* avg, int ah, const int32_t* res, int rh, int w, int32_t* out, hipStream_t s);
void launch_rct(int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int type, hipStream_t s);
void launch_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale, float* out, hipStream_t s);

// LF stage (row f1): q/out device pointers; out written at out_off + y*out_stride + x
// one channel of a chroma-subsampled frame: out = q * (scaledDequant / (1 << extraPrecision)), no CfL, no smoothing
void launch_lf_dequant_plain(const int32_t* q, float* out, int H, int W, int64_t out_off, int out_stride, float scaled_dequant,
                             int extra_precision, hipStream_t s);
void launch_lf_dequant(const int32_t* const q[3], float* const out[3], int H, int W, int64_t out_off, int out_stride,
                       const float scaled_dequant[3], int extra_precision, float base_corr_x, float base_corr_b,
                       int color_factor, int x_factor_lf, int b_factor_lf, int smooth, hipStream_t s);

// rows f4 / f3 (k_post.hip); all pointers are device pointers
void launch_chroma_upsample_h(const float* in, int h, int w, float* out, hipStream_t s);
void launch_chroma_upsample_v(const float* in, int h, int w, float* out, hipStream_t s);
void launch_upsample(const float* in, int h, int w, int k, const float* weights, float* out, hipStream_t s);
void launch_noise_init(int h, int w, int group_dim, uint64_t seed0, int colors, float* const tmp[3], float* const out[3],
                       hipStream_t s);
void launch_noise_add(float* const planes[3], const float* const noise[3], int64_t n, const float lut[8], float bcx, float bcb,
                      hipStream_t s);
// maps (mode, flags, is_int) to the inner blend function; -1 illegal mode, -2 int samples on a float-only function
int blend_op(int mode, unsigned flags, int is_int);
bool blend_needs(int op, bool* frame, bool* ref, bool* frame_alpha, bool* ref_alpha, bool is_alpha);
void launch_blend(int op, unsigned flags, void* canvas, int cw, const void* frame, int fw, const void* ref, int rw,
                  const float* frame_alpha, const float* ref_alpha, const jxl_blend_rect& r, hipStream_t s);
void launch_orient(const void* in, int h, int w, int orientation, void* out, hipStream_t s);
void launch_pack(const void* const planes[4], const jxl_pack_params& p, bool coerce, void* out, hipStream_t s);

void launch_idct2d_single(const float* src, float* dst, int h, int w, int transposed, const float* lut, hipStream_t s);
void launch_fdct2d_single(const float* src, float* dst, int h, int w, const float* lut, hipStream_t s);

}  // namespace jxl
