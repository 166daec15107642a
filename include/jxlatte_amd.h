/*
 * jxlatte_amd.h -- C-ABI of the MI355X (gfx950) transform back-end for jxlatte.
 *
 * The reference (Traneptora/jxlatte, pure Java) has no FFI; this header is the
 * boundary *cut* at the call sites of its per-frame transform stage
 * (J/ = java/com/traneptora/jxlatte/ in the reference tree):
 *
 *   J/frame/Frame.java:361-374      decodePassGroups VarDCT tail -> PassGroup.invertVarDCT
 *   J/frame/Frame.java:427          globalModular.applyTransforms()
 *   J/frame/Frame.java:430-461      modular->buffer, Gab, EPF
 *   J/JXLCodestreamDecoder.java:637 performColorTransforms (invertXYB)
 *   J/io/PNGWriter.java:65,105-111  transfer (PQ/sRGB) + castToIntWithMax
 *
 * Plain pointers and sizes only. Every function returns a jxl_status
 * (0 = OK, negative = error); no exception crosses the ABI. The two reference
 * exception families map 1:1: InvalidBitstreamException -> JXL_ERR_INVALID_BITSTREAM,
 * UnsupportedOperationException -> JXL_ERR_UNSUPPORTED.
 *
 * Layout everywhere: planar, row-major, 32-bit elements, channel order X,Y,B
 * (buffer index 0,1,2 as in Frame.buffer[]). "cell" = 8x8 px, "tile" = 64x64 px
 * (chroma-from-luma granularity), "group" = 256x256 px, "LF group" = 2048x2048 px.
 */
#ifndef JXLATTE_AMD_H
#define JXLATTE_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t jxl_status;
#define JXL_OK                      0
#define JXL_ERR_INVALID_ARGUMENT  (-1)
#define JXL_ERR_INVALID_BITSTREAM (-2) /* J/io/InvalidBitstreamException.java:5 */
#define JXL_ERR_UNSUPPORTED       (-3) /* UnsupportedOperationException, PassGroup.java:326-327 */
#define JXL_ERR_DEVICE            (-4)
#define JXL_ERR_OOM               (-5)
#define JXL_ERR_STATE             (-6) /* IllegalStateException (call order) */

/* TransformType.type values, J/frame/vardct/TransformType.java:10-36 */
#define JXL_NUM_TRANSFORM_TYPES 27
/* number of quant-weight parameter sets, HFGlobal.weights[17][3][][] (HFGlobal.java:191) */
#define JXL_NUM_WEIGHT_SETS 17

/* output stage selector (PNGWriter ctor: tf = hdr ? PQ : sRGB; bit depth 8/16) */
#define JXL_TRANSFER_NONE 0 /* stop after invertXYB: linear float */
#define JXL_TRANSFER_PQ   1 /* TransferFunction.TF_PQ.fromLinear, TransferFunction.java:83-87 */
#define JXL_TRANSFER_SRGB 2 /* TransferFunction.TF_SRGB.fromLinearF, :39-44. With 8-bit or 16-bit output (max 255 / 65535) transfer and
                            * quantisation go through tables of the composite's thresholds: the ORACLE's integer for every float
                            * input (all 2^32 checked for both). The thresholds are bisected on the host with glibc's pow(), the
                            * oracle's form; Java's Math.pow is a HotSpot intrinsic specified to 1 ulp, so a threshold that sits on
                            * such a disagreement may move one code value against a JVM (unpinned until a JVM runs tests/test_jvm_pin.py).
                            * Float output evaluates the double pow on the device */
/* Tolerance of the PQ entries against the reference's (float)Math.pow(double) form. JXL_TRANSFER_PQ evaluates a table of
 * quadratic segments (jxl_fastpow.h): as FLOAT output over ALL 2^32 inputs 99.96 % identical, the rest off by exactly 1 ulp, none
 * worse (profiles/r3_pq_sweep.txt). With 16-bit output (JXL_OUT_U16 / RGB16, max 65535) the quantised sample is the oracle's
 * integer (glibc pow) for EVERY input (r3: the table value is settled against the composite's 65 535 thresholds, profiles/r3_pq16_sweep.txt);
 * an 8-bit PQ sample likewise (binary search in its 255 thresholds, profiles/r3_pq8_sweep.txt). JXL_TRANSFER_PQ_EXACT evaluates the two
 * pow() in double precision on the device (3x the instructions; the pre-round-2 form). Both are accepted by
 * jxl_vardct_params.transfer and jxl_stage_transfer. */
#define JXL_TRANSFER_PQ_EXACT 3
#define JXL_OUT_F32 0       /* float planes */
#define JXL_OUT_U16 1       /* ImageBuffer.castToIntWithMax(65535), ImageBuffer.java:129-147 */
#define JXL_OUT_U8  2       /* ImageBuffer.castToIntWithMax(255) */
/* row f3: the same quantised samples, pixel-interleaved R,G,B in the order PNGWriter.writeIDAT emits them
 * (PNGWriter.java:191-203); one buffer of height*width*3 elements (out[0]; out[1], out[2] unused). u16 is host order. */
#define JXL_OUT_RGB8  3
#define JXL_OUT_RGB16 4

/* stage mask bits for jxl_vardct_params.stages */
#define JXL_STAGE_IDCT 1u  /* dequant + CfL + LLF + inverse transforms (PassGroup.invertVarDCT) */
#define JXL_STAGE_GAB  2u  /* Frame.performGabConvolution */
#define JXL_STAGE_EPF  4u  /* Frame.performEdgePreservingFilter */
#define JXL_STAGE_XYB  8u  /* OpsinInverseMatrix.invertXYB */
#define JXL_STAGE_OUT 16u  /* transfer + int quantisation */

typedef struct jxl_ctx jxl_ctx;

/* Frame-constant parameters of one VarDCT frame. All float fields are produced by
 * the host exactly as the reference computes them (file:line given per field). */
typedef struct jxl_vardct_params {
    int32_t width;   /* Frame.getPaddedFrameSize().width  (Frame.java:924-941), multiple of 8 */
    int32_t height;  /* Frame.getPaddedFrameSize().height */
    uint32_t stages; /* JXL_STAGE_* mask; gab/epf/xyb bits are ANDed with the flags below */

    /* HFCoefficients.dequantizeHFCoefficients (HFCoefficients.java:267-275) */
    float scale_factor[3];      /* {gs*(float)pow(0.8,xqm-2), gs, gs*(float)pow(0.8,bqm-2)}, gs = 65536f/globalScale */
    float quant_bias[3];        /* OpsinInverseMatrix.quantBias */
    float quant_bias_numerator; /* OpsinInverseMatrix.quantBiasNumerator */

    /* HFCoefficients.chromaFromLuma (:146-192), LFChannelCorrelation */
    float base_corr_x;
    float base_corr_b;
    int32_t color_factor;

    /* RestorationFilter (RestorationFilter.java:12-79) */
    int32_t gab;         /* restorationFilter.gab */
    float gab_w1[3];     /* gab1Weights */
    float gab_w2[3];     /* gab2Weights */
    int32_t epf_iters;   /* epfIterations 0..3 */
    float global_scale_f;    /* 65536f / lfGlobal.globalScale (Frame.java:554) */
    float epf_sharp_lut[8];  /* epfSharpLut, already multiplied by epfQuantMul (:78) */
    float epf_channel_scale[3];
    float epf_pass0_sigma_scale;
    float epf_pass2_sigma_scale;
    float epf_border_sad_mul;

    /* OpsinInverseMatrix (OpsinInverseMatrix.java:105-142) */
    int32_t xyb;             /* matrix != null in performColorTransforms */
    float opsin_matrix[9];   /* adapted matrix (getMatrix), NOT yet scaled by 255/intensityTarget */
    float opsin_bias[3];
    float cbrt_opsin_bias[3];/* (float)Math.cbrt(opsinBias[c]) (:83) */
    float intensity_target;

    int32_t transfer;   /* JXL_TRANSFER_* */
    int32_t out_format; /* JXL_OUT_* */

    /* row a16, JPEG-recompressed frames: FrameHeader.jpegUpsamplingY/X[c] AFTER the header's normalisation
     * (FrameHeader.java:190-195: max - own), i.e. the shift by which channel c (X/Cb, Y, B/Cr buffer order) is smaller
     * than the padded frame. All zero for ordinary frames. With a non-zero shift: put_group planes, lf[] planes and the
     * coefficient positions of channel c are in its own subsampled geometry (PassGroup.java:213-226), chroma-from-luma
     * is skipped (HFCoefficients.java:149-151), and Frame.invertSubsampling (Frame.java:681-723) runs right after the
     * inverse transforms, before Gab / EPF. */
    int32_t jpeg_upsampling_y[3];
    int32_t jpeg_upsampling_x[3];
} jxl_vardct_params;

/* One LF group's side information, in the reference's own per-LF-group shape
 * (HFMetadata.java:16-52, LFCoefficients.dequantLFCoeff). Host pointers. */
typedef struct jxl_lfgroup_desc {
    int32_t lfg_y, lfg_x;       /* Frame.getLFGroupLocation: position in LF-group units */
    int32_t cells_h, cells_w;   /* LFGroup.size in 8x8 cells (<= 256) */
    const uint8_t* dct_select;  /* [cells_h][cells_w] TransformType.type of the covering varblock */
    const int32_t* hf_mul;      /* [cells_h][cells_w] hfMultiplier */
    const int32_t* sharpness;   /* [cells_h][cells_w] hfStreamBuffer[3] */
    const int32_t* x_from_y;    /* [ceil(cells_h/8)][ceil(cells_w/8)] hfStreamBuffer[0] */
    const int32_t* b_from_y;    /* same shape, hfStreamBuffer[1] */
    const int32_t* block_yx;    /* blockList: n_blocks x {y,x} in cells, placement order */
    int32_t n_blocks;
    const float* lf[3];         /* dequantLFCoeff[c] [cells_h][cells_w] */
} jxl_lfgroup_desc;

/* SqueezeParam (J/frame/modular/SqueezeParam.java) */
typedef struct jxl_squeeze_param {
    int32_t horizontal;
    int32_t in_place;
    int32_t begin_c;
    int32_t num_c;
} jxl_squeeze_param;

/* A modular channel plane (ModularChannel.buffer + size). Host pointer. */
typedef struct jxl_channel {
    int32_t width;
    int32_t height;
    int32_t* data; /* [height][width], may be NULL when width*height == 0 */
} jxl_channel;

/* ---- context ------------------------------------------------------------------ */
/* One ctx = one HIP device + one stream + a device arena reused across frames.
 * Single-threaded like a JXLDecoder instance; distinct ctxs are independent. */
jxl_status jxl_ctx_create(int32_t device, jxl_ctx** out);
void       jxl_ctx_destroy(jxl_ctx* ctx);
const char* jxl_last_error(const jxl_ctx* ctx);
const char* jxl_version(void);
jxl_status jxl_ctx_synchronize(jxl_ctx* ctx);
/* HIP stream handle (hipStream_t) the ctx launches on; for event timing by callers. */
void*      jxl_ctx_stream(jxl_ctx* ctx);
/* Launch on a caller-owned HIP stream instead (e.g. several frame contexts sharing one stream, or
 * torch's current stream). The ctx no longer owns a stream after this call. */
jxl_status jxl_ctx_set_stream(jxl_ctx* ctx, void* hip_stream);

/* ---- VarDCT frame path: replaces Frame.decodePassGroups tail .. performColorTransforms */
/* call order: begin_frame, set_weights, set_lfgroup* , put_group*, (run | finish_frame) */
jxl_status jxl_vardct_begin_frame(jxl_ctx* ctx, const jxl_vardct_params* params);
/* HFGlobal.weights (already reciprocal, HFGlobal.java:421-431): 17 sets x 3 channels, each
 * matrixHeight x matrixWidth row-major; offs[p*3+c] = element offset of set p channel c in w. */
jxl_status jxl_vardct_set_weights(jxl_ctx* ctx, const float* w, size_t n_floats, const int32_t* offs /*[51]*/);
jxl_status jxl_vardct_set_lfgroup(jxl_ctx* ctx, const jxl_lfgroup_desc* lfg);
/* Row f1 (LF stage on device): instead of lfg->lf[] (already dequantised), hand over the INTEGER LF image of the LF
 * group and let the device run LFCoefficients.java:65-75 (dequant), :78-95 (LF chroma-from-luma) and :113-180
 * (adaptiveSmooth). Call after jxl_vardct_set_lfgroup for the same LF group (its lf[] pointers may then be NULL).
 * lf_quant[i]: lfQuant[cMap[i]] i.e. already in X,Y,B buffer order, [cells_h][cells_w] int32;
 * scaled_dequant = LFGlobal.scaledDequant (X,Y,B); x/b_factor_lf = LFChannelCorrelation.xFactorLF/bFactorLF. */
typedef struct jxl_lfquant_desc {
    int32_t lfg_y, lfg_x;
    int32_t cells_h, cells_w;
    const int32_t* lf_quant[3];
    int32_t extra_precision;     /* reader.readBits(2), LFCoefficients.java:61 */
    float scaled_dequant[3];
    int32_t x_factor_lf, b_factor_lf;
    int32_t adaptive_smoothing;  /* (flags & (SKIP_ADAPTIVE_LF_SMOOTHING | USE_LF_FRAME)) == 0 */
} jxl_lfquant_desc;
jxl_status jxl_vardct_set_lfgroup_lfquant(jxl_ctx* ctx, const jxl_lfquant_desc* d);

/* quantizedCoeffs of one (pass, group) (HFCoefficients.java:43,68): q[c] is [gh][gw] with row
 * stride[c] elements; gh,gw = Frame.getGroupSize(group). pass > 0 accumulates
 * (PassGroup.java:174-200).
 * Buffer lifetime -- NOT "retains nothing after the call" for every source: pageable sources (and page-locked ones that are
 * not 16-byte aligned in address and row stride) are copied before the call returns and may be reused at once. Aligned
 * page-locked sources (jxl_host_alloc, or memory the caller registered) are read by the DEVICE in place, asynchronously, by a
 * kernel queued on the context's stream: keep them unchanged until a call that waits for that stream has returned --
 * jxl_vardct_finish_frame / jxl_vardct_read_output / jxl_vardct_read_output_wait of this frame, or jxl_ctx_synchronize
 * (jxl_vardct_run only queues work). A caller that wants the copy semantics with page-locked memory passes a pointer that is
 * not 16-byte aligned, or copies itself. The JNI shim's callers (integration/jni/GpuFrameBridge.java) hand over fresh pageable
 * direct ByteBuffers: always the copying path. The call itself never waits for the device except when more than 8 puts are
 * still in flight. */
jxl_status jxl_vardct_put_group(jxl_ctx* ctx, int32_t pass, int32_t group,
                                const int32_t* const q[3], const int32_t stride[3]);
/* The same with 16-bit samples -- the wire format for the PCIe leg: quantised HF coefficients of photographic content fit
 * int16 (|q| < 32768; the Java host checks while it fills the buffer and falls back to jxl_vardct_put_group for a group that
 * does not), which halves the bytes of the dominant transfer. The device widens into the same int32 planes; everything
 * downstream is identical. */
jxl_status jxl_vardct_put_group_i16(jxl_ctx* ctx, int32_t pass, int32_t group,
                                    const int16_t* const q[3], const int32_t stride[3]);
/* The whole frame's coefficients in ONE page-locked buffer the library owns: the entropy decoder writes its groups in place
 * (planes[c] is [H >> sy][W >> sx] int16 with row stride strides[c]; group g occupies the rectangle Frame.getGroupLocation /
 * getGroupSize give it, HFCoefficients.java:64-69) and jxl_vardct_commit_coeffs_i16 moves the three planes with three DMA
 * transfers instead of three per group (405 per 4K frame: their fixed cost, not their bytes, is what the per-group entry
 * pays). map zero-fills the planes (the reference's `new int[..]`, HFCoefficients.java:68: only non-zero coefficients are
 * ever written) and is valid until the next begin_frame; commit is asynchronous (jxl_vardct_run is ordered behind it) and
 * may be followed by jxl_vardct_put_group for groups whose samples did not fit 16 bits, or for later passes. */
jxl_status jxl_vardct_map_coeffs_i16(jxl_ctx* ctx, int16_t* planes[3], int32_t strides[3]);
/* Rows of the three mapped planes ((paddedHeight >> jpegUpsamplingY[c]), HFCoefficients.java:64-69): plane c holds
 * rows[c] * strides[c] samples. The JNI shim sizes its direct ByteBuffers from this, never from a caller-supplied count. */
jxl_status jxl_vardct_coeff_plane_rows(jxl_ctx* ctx, int32_t rows[3]);
/* Geometry of the open frame, for callers that must size buffers from the library's own numbers (the JNI shim checks every
 * direct buffer it is handed against these): info[0..2] = plane width of channel c (paddedWidth >> jpegUpsamplingX[c]),
 * info[3..5] = plane height, info[6..7] = the two shifts (x, y) of channel 0, [8..9] of channel 1, [10..11] of channel 2,
 * info[12] = bytes per output sample. Group g of a frame covers Frame.getGroupLocation / getGroupSize (J/frame/Frame.java:767-786):
 * jxl_vardct_group_size gives its width and height per channel. */
jxl_status jxl_vardct_geometry(jxl_ctx* ctx, int32_t info[13]);
/* Geometry of the OUTPUT of the open frame (valid from begin_frame on; what jxl_vardct_read_output* / finish_frame write): info[0],
 * info[1] = width and height of every output plane -- always the full padded frame, also for chroma-subsampled frames (the planes of
 * jxl_vardct_geometry are the COEFFICIENT planes) --, info[2] = bytes per sample, info[3] = 1 if the three colours are interleaved
 * into out[0] (JXL_OUT_RGB8 / JXL_OUT_RGB16: rows of 3 * width samples; out[1], out[2] are ignored), info[4] = number of buffers
 * that must be non-null (1 or 3). A buffer with row stride `s` pixels (>= width) holds
 *     info[2] * (info[3] ? 3 : 1) * ((info[1] - 1) * s + info[0])   bytes.
 * Stands for the sizes the reference states implicitly: Frame.buffer[c] = new float[paddedHeight][paddedWidth]
 * (J/frame/Frame.java:331-340) and PNGWriter's interleaved rows (J/io/PNGWriter.java:191-212). */
jxl_status jxl_vardct_output_geometry(jxl_ctx* ctx, int32_t info[5]);
jxl_status jxl_vardct_group_size(jxl_ctx* ctx, int32_t group, int32_t gw[3], int32_t gh[3]);
jxl_status jxl_vardct_commit_coeffs_i16(jxl_ctx* ctx);
/* The same pair without the zero-fill (r4): a decoder writes EVERY sample of every group it decodes (HFCoefficients.java:76-138
 * leaves the untouched samples of its fresh int[][] at zero -- the caller of this form stores those zeros itself, or keeps its
 * own cleared scratch and copies whole groups), so map's 50 MB of host stores per 4K frame are wasted on it. With
 * JXL_MAP_NO_FILL the planes come back as they are; commit_..._groups names the groups whose rectangles the caller has fully
 * written (group_written[g] != 0, g in Frame group order, n_groups = all groups of the frame); every other group reads as zero
 * (its rectangle is cleared in the staging buffer before the transfer). map also no longer waits for the context's whole
 * stream, only for the previous commit's transfers to have read the buffer. */
#define JXL_MAP_NO_FILL 1
jxl_status jxl_vardct_map_coeffs_i16_ex(jxl_ctx* ctx, int16_t* planes[3], int32_t strides[3], int32_t flags);
jxl_status jxl_vardct_commit_coeffs_i16_groups(jxl_ctx* ctx, const uint8_t* group_written, int32_t n_groups);
/* Page-locked host memory for the buffers that cross the bus (coefficient planes in, pixel planes out; a JNI caller wraps it
 * with NewDirectByteBuffer). put_group / put_group_i16 / read_output recognise such pointers: the device reads / writes them
 * in place at bus speed instead of through a staged copy out of pageable memory, and put_group returns without waiting
 * (the buffer must stay untouched until jxl_vardct_run or jxl_ctx_synchronize). NULL on failure. */
void* jxl_host_alloc(size_t bytes);
void jxl_host_free(void* p);
/* Host-side preparation a run needs and would otherwise do on first use: varblock binning by transform type
 * (the device counterpart of walking HFMetadata.blockList, HFCoefficients.java:76-85), the chroma-from-luma
 * cache-order masks (HFCoefficients.java:159-181), upload of the side tables, LF dequantisation jobs. Synchronous.
 * Idempotent until the frame's inputs change; jxl_vardct_run calls it implicitly. bench.py times it as
 * `host_prepare_ms`.
 * Deliberate refusal (JXL_ERR_UNSUPPORTED): a varblock with a 128- or 256-sample edge in a CHROMA-SUBSAMPLED frame. The
 * reference transforms every channel's copy of such a block at the channel's own geometry (PassGroup.java:203-233,
 * TransformType.java:10-36), where the copies of neighbouring blocks overlap in the subsampled planes and the later one
 * overwrites the earlier: its output depends on its visiting order, no encoder emits such frames (libjxl's only subsampled
 * frames are JPEG recompressions, DCT8 throughout), and a device that transforms blocks concurrently has no such order to
 * reproduce. Frames without subsampling take these blocks through the three-launch path of k_idct.hip (dequantise, column
 * pass, row pass through a scratch plane). */
jxl_status jxl_vardct_prepare(jxl_ctx* ctx);
/* Launch every enabled stage on the ctx stream; inputs are resident after put_group.
 * Asynchronous: returns after enqueue. Re-runnable (inputs are not consumed). */
jxl_status jxl_vardct_run(jxl_ctx* ctx);
/* Run n independent frames (one context each, all on one device; every context prepared exactly as for jxl_vardct_run).
 * The reference decodes its frames one after the other (JXLCodestreamDecoder.decode, :506-720); this entry is what a
 * batched caller (BASELINE config 5: 8 frames per GPU) uses instead of n jxl_vardct_run calls: the inverse-transform stage
 * of all frames is enqueued as one launch per kernel class, the remaining stages per frame on the frames' own streams.
 * Same results, same completion rule (synchronise / read each context as usual). Frames the shared launches do not cover
 * are run one by one. Measured on MI355X: the faster form for small, launch-bound frames (8 x 1280x720: +16 %, 8 x 512x512:
 * +40..70 %); for 4K frames n jxl_vardct_run calls on n contexts are 5 % faster. */
jxl_status jxl_vardct_run_batch(jxl_ctx* const* ctxs, int32_t n);
/* run + synchronize + copy result planes to the host. out[c]: width*height elements of
 * float (JXL_OUT_F32) / uint16 / uint8, row stride = out_stride elements. */
jxl_status jxl_vardct_finish_frame(jxl_ctx* ctx, void* const out[3], int64_t out_stride);
/* copy the last run's result planes (device) to host without re-running */
jxl_status jxl_vardct_read_output(jxl_ctx* ctx, void* const out[3], int64_t out_stride);
/* read_output in two halves (r4): _begin queues the device-to-host copies behind the frame's kernels and returns, _wait blocks
 * until they have landed. In between the host may drive the NEXT frame of this context (begin_frame ... commit ... run: its
 * device work queues behind the copies), which is how one context overlaps the host's share of frame n+1 with the device's share
 * of frame n -- what the reference's one-frame-at-a-time loop (JXLCodestreamDecoder.decode, :506-720) leaves on the table. The
 * destination must stay valid until _wait and should be page-locked (jxl_host_alloc): a page-locked, 16-byte aligned destination
 * with dense rows is written by a kernel through its device alias (r5: no runtime copy call -- with several contexts streaming
 * frames each hipMemcpyAsync held its caller for 1.6-2.5 ms), anything else goes through the runtime's copy. A host that runs
 * several decoder contexts should start with GPU_MAX_HW_QUEUES=16 in its environment (the runtime's default of 4 hardware
 * queues makes one context's launches wait behind the others' bus transfers). */
jxl_status jxl_vardct_read_output_begin(jxl_ctx* ctx, void* const out[3], int64_t out_stride);
jxl_status jxl_vardct_read_output_wait(jxl_ctx* ctx);
/* ---- the frame's colour planes kept on the device between the stages that follow decodeFrame --------------------------
 * JXLCodestreamDecoder.decode runs, on the frame's own buffers and in this order (JXLCodestreamDecoder.java:628-637):
 * Frame.upsample, Frame.initializeNoise, computePatches, Frame.renderSplines, Frame.synthesizeNoise,
 * performColorTransforms. The jxl_stage_* entries take and return host planes; these entries run the same kernels on a set
 * of three float planes that STAYS in device memory, so a frame with upsampling / noise costs one transfer in (its
 * coefficients) and one out (its pixels). Patches and splines stay host code (as in the reference): jxl_planes_download /
 * jxl_planes_upload bracket them, only for frames that have them. */
/* adopt the top-left height x width window (Frame bounds; the restoration filters worked on the padded size) of the last
 * jxl_vardct_run's result as the resident planes. The run must have produced float planes (no transfer / integer output;
 * XYB stage off if the later stages need XYB samples). */
jxl_status jxl_planes_from_frame(jxl_ctx* ctx, int32_t height, int32_t width);
/* Frame.performUpsampling (Frame.java:217-260) of the three planes, k = 2, 4, 8; weights as for jxl_stage_upsample */
jxl_status jxl_planes_upsample(jxl_ctx* ctx, int32_t k, const float* weights);
/* Frame.initializeNoise (Frame.java:748-788) + Frame.synthesizeNoise (:790-831) on the resident planes */
jxl_status jxl_planes_noise(jxl_ctx* ctx, int32_t group_dim, uint64_t seed0, const float lut[8], float base_corr_x, float base_corr_b);
/* OpsinInverseMatrix.invertXYB / the YCbCr branch of performColorTransforms on the resident planes */
jxl_status jxl_planes_xyb(jxl_ctx* ctx, const float matrix[9], const float opsin_bias[3], const float cbrt_opsin_bias[3],
                          float intensity_target);
jxl_status jxl_planes_ycbcr(jxl_ctx* ctx);
/* current size of the resident planes */
jxl_status jxl_planes_shape(const jxl_ctx* ctx, int32_t* height, int32_t* width);
/* the host hook (patches, splines, saveBeforeCT references) and the way out: dense height x width float planes */
jxl_status jxl_planes_download(jxl_ctx* ctx, float* const out[3]);
jxl_status jxl_planes_upload(jxl_ctx* ctx, const float* const in[3], int32_t height, int32_t width);

/* device-to-device copy of the result into caller-owned device memory (e.g. an RCCL send
 * buffer): dst = 3 planes back to back, width*height elements each. Async on the ctx stream. */
jxl_status jxl_vardct_copy_output_device(jxl_ctx* ctx, void* dst_device);
/* bytes of one output element for the configured out_format */
int32_t    jxl_vardct_out_elem_size(const jxl_ctx* ctx);
/* number of kernel launches the last jxl_vardct_run enqueued (diagnostics) */
int32_t    jxl_vardct_last_launch_count(const jxl_ctx* ctx);
/* HIP-event timing on the ctx stream, averaged over the runs recorded since timing was enabled
 * (ring of the 32 most recent): which = 0 whole run, 1 IDCT stage, 2 restoration+colour stage (from the end of the IDCT stage's last
 * launch: the boundary between the two launches is inside), 3 the fused restoration kernel's own start -> stop (what a profiler reports
 * for that launch; JXL_ERR_STATE if the timed runs did not take that kernel). */
jxl_status jxl_vardct_last_stage_ms(jxl_ctx* ctx, int32_t which, float* ms);
jxl_status jxl_vardct_enable_stage_timing(jxl_ctx* ctx, int32_t on);

/* ---- stage-level entry points (host planes in, host planes out; synchronous) --- */
/* One per reference function on the path; used by the parity tests. */

/* MathHelper.inverseDCT2D (MathHelper.java:96-122) on one h x w block. */
jxl_status jxl_stage_idct2d(jxl_ctx* ctx, const float* src, float* dst, int32_t h, int32_t w, int32_t transposed);
/* MathHelper.forwardDCT2D (MathHelper.java:124-136). */
jxl_status jxl_stage_fdct2d(jxl_ctx* ctx, const float* src, float* dst, int32_t h, int32_t w);
/* Frame.performGabConvolution (Frame.java:505-542). */
jxl_status jxl_stage_gab(jxl_ctx* ctx, const float* const in[3], float* const out[3],
                         int32_t height, int32_t width, const float w1[3], const float w2[3]);
/* Frame.performEdgePreservingFilter (Frame.java:544-636). inv_sigma: [ceil(h/8)][ceil(w/8)]
 * map for VarDCT, or NULL to use the constant inv_sigma_modular (Frame.java:573-575). */
jxl_status jxl_stage_epf(jxl_ctx* ctx, const float* const in[3], float* const out[3],
                         int32_t height, int32_t width, int32_t iterations,
                         const float* inv_sigma, float inv_sigma_modular,
                         const float channel_scale[3], float pass0_sigma_scale,
                         float pass2_sigma_scale, float border_sad_mul);
/* inverse-sigma map of Frame.java:552-571 from hfMul + sharpness cell maps. */
jxl_status jxl_stage_epf_sigma(jxl_ctx* ctx, const int32_t* hf_mul, const int32_t* sharpness,
                               int32_t bh, int32_t bw, float global_scale_f,
                               const float sharp_lut[8], float* inv_sigma);
/* LFCoefficients dequant + LF CfL + adaptiveSmooth (LFCoefficients.java:65-180) of one LF group: out[c] [cells_h][cells_w].
 * base_corr_x/b and color_factor as in jxl_vardct_params. */
jxl_status jxl_stage_lf_dequant(jxl_ctx* ctx, const jxl_lfquant_desc* d, float base_corr_x, float base_corr_b,
                                int32_t color_factor, float* const out[3]);
/* OpsinInverseMatrix.invertXYB (OpsinInverseMatrix.java:105-142), in place on planes[3]. */
jxl_status jxl_stage_xyb(jxl_ctx* ctx, float* const planes[3], int64_t n,
                         const float matrix[9], const float opsin_bias[3],
                         const float cbrt_opsin_bias[3], float intensity_target);
/* YCbCr branch of performColorTransforms (JXLCodestreamDecoder.java:270-281), in place. */
jxl_status jxl_stage_ycbcr(jxl_ctx* ctx, float* const planes[3], int64_t n);
/* JXLImage.transferInPlace + ImageBuffer.castToInt0: transfer = JXL_TRANSFER_*,
 * max_value = 0 keeps float output in out_f, else writes clamped ints to out_i. */
jxl_status jxl_stage_transfer(jxl_ctx* ctx, const float* in, int64_t n, int32_t transfer,
                              int32_t max_value, float* out_f, int32_t* out_i);
/* ModularChannel.inverseHorizontalSqueeze / inverseVerticalSqueeze
 * (ModularChannel.java:361-413). out is (h) x (aw+rw) resp. (ah+rh) x (w). */
jxl_status jxl_stage_inv_hsqueeze(jxl_ctx* ctx, const int32_t* avg, int32_t aw, const int32_t* res, int32_t rw,
                                  int32_t h, int32_t* out);
jxl_status jxl_stage_inv_vsqueeze(jxl_ctx* ctx, const int32_t* avg, int32_t ah, const int32_t* res, int32_t rh,
                                  int32_t w, int32_t* out);
/* RCT branch of ModularStream.applyTransforms (ModularStream.java:255-326): in place on
 * v[3] of n samples; rct_type = permutation*7 + type. On return v[] holds the planes in
 * output channel order (the permutation is applied). */
jxl_status jxl_stage_rct(jxl_ctx* ctx, int32_t* const v[3], int64_t n, int32_t rct_type);
/* Frame.decodeFrame modular->buffer (Frame.java:430-455) for one output channel:
 * out = scale * (a + b) (b may be NULL) as float. */
jxl_status jxl_stage_modular_to_float(jxl_ctx* ctx, const int32_t* a, const int32_t* b, int64_t n,
                                      float scale, float* out);

/* ---- row f4: pixel-domain stencils that run between EPF and the colour transform ---- */
/* Frame.invertSubsampling (Frame.java:681-723) for one channel: x_shift horizontal doublings then y_shift
 * vertical doublings (3/4, 1/4 triangle, replicated edges). out is (h << y_shift) x (w << x_shift). */
jxl_status jxl_stage_chroma_upsample(jxl_ctx* ctx, const float* in, int32_t h, int32_t w, int32_t x_shift,
                                     int32_t y_shift, float* out);
/* ImageHeader.getUpWeights index expansion (ImageHeader.java:441-470): packed = the k==2: 15, k==4: 55,
 * k==8: 210 coefficient list of the image header; out = [k][k][5][5]. Host-only helper, no device work. */
jxl_status jxl_upsampling_weights(int32_t k, const float* packed, float* out);
/* Frame.performUpsampling (Frame.java:217-260): k in {2,4,8}, weights [k][k][5][5], mirrored edges,
 * result clamped to the reference's [min, max] window (max starts at Float.MIN_VALUE, :237). out is (h*k) x (w*k). */
jxl_status jxl_stage_upsample(jxl_ctx* ctx, const float* in, int32_t h, int32_t w, int32_t k, const float* weights,
                              float* out);
/* Frame.initializeNoise (Frame.java:748-788): per-group XorShiro streams (features/XorShiro.java) turned into
 * floats in [1,2), then the 5x5 "laplacian" high-pass with mirrored edges. seed0 = (visibleFrames << 32) |
 * invisibleFrames (JXLCodestreamDecoder.java:629). out[c]: h x w, c < colors. */
jxl_status jxl_stage_noise_init(jxl_ctx* ctx, int32_t h, int32_t w, int32_t group_dim, uint64_t seed0, int32_t colors,
                                float* const out[3]);
/* Frame.synthesizeNoise (Frame.java:790-831), in place on the XYB planes[3] (X, Y, B); lut = LFGlobal.noiseParameters[8]. */
jxl_status jxl_stage_noise_add(jxl_ctx* ctx, float* const planes[3], const float* const noise[3], int64_t n,
                               const float lut[8], float base_corr_x, float base_corr_b);

/* ---- row f3: output stage (blending, orientation, sample packing) ---- */
#define JXL_BLEND_REPLACE 0 /* FrameFlags.java:18-22 */
#define JXL_BLEND_ADD     1
#define JXL_BLEND_BLEND   2
#define JXL_BLEND_MULADD  3
#define JXL_BLEND_MULT    4
#define JXL_BLEND_FLAG_IS_ALPHA  1u /* this channel is the alpha channel itself */
#define JXL_BLEND_FLAG_HAS_EXTRA 2u /* the image has extra channels (else BLEND / MULADD degrade to ADD) */
#define JXL_BLEND_FLAG_CLAMP     4u /* BlendingInfo.clamp */
#define JXL_BLEND_FLAG_PREMULT   8u /* alpha is associated */
typedef struct jxl_blend_rect {
    int32_t h, w;               /* blendSize */
    int32_t canvas_y, canvas_x; /* patchStart: where the rectangle lands on the canvas */
    int32_t frame_y, frame_x;   /* frameOffset: its origin inside the frame buffers */
    int32_t ref_y, ref_x;       /* refOffset: its origin inside the reference buffers */
} jxl_blend_rect;
/* One channel of JXLCodestreamDecoder.blendBuffers' inner switch (JXLCodestreamDecoder.java:26-40 copyToCanvas,
 * :285-318 blendAdd, :320-340 blendMult, :342-386 blendBlend, :388-422 blendMulAdd). "frame" and "ref" are the
 * arguments those functions receive under these names. canvas is ch x cw and updated in place inside the rectangle;
 * frame / frame_alpha are fh x fw; ref / ref_alpha are rh x rw. is_int: samples are int32 (REPLACE and the ADD
 * cases only), else float. Unused planes may be NULL. */
jxl_status jxl_stage_blend(jxl_ctx* ctx, int32_t mode, uint32_t flags, int32_t is_int,
                           void* canvas, int32_t ch, int32_t cw, const void* frame, int32_t fh, int32_t fw,
                           const void* ref, int32_t rh, int32_t rw, const float* frame_alpha, const float* ref_alpha,
                           const jxl_blend_rect* rect);
/* JXLCodestreamDecoder.transposeBufferFloat / transposeBufferInt (:43-177): EXIF orientation 1..8 of one plane of
 * 4-byte samples. out is h x w for orientation <= 4, else w x h. */
jxl_status jxl_stage_orient(jxl_ctx* ctx, const void* in, int32_t h, int32_t w, int32_t orientation, void* out);
/* PNGWriter ctor tail + writeIDAT sample order (PNGWriter.java:79-111, 191-203): coerce to float when needed,
 * un-premultiply, quantise / clamp to bit_depth, and interleave colour channels then alpha. */
typedef struct jxl_pack_params {
    int32_t height, width;
    int32_t n_color;         /* 1 (gray) or 3 */
    int32_t has_alpha;       /* planes[n_color] is the alpha plane */
    int32_t premultiplied;   /* image.isAlphaPremultiplied() */
    int32_t bit_depth;       /* 8 or 16 */
    int32_t big_endian;      /* 16-bit samples as DataOutput.writeShort emits them (PNG), else host order */
    int32_t is_int[4];       /* plane holds int32 samples (else float) */
    int32_t tagged_depth[4]; /* image.getTaggedBitDepth(c) */
} jxl_pack_params;
/* out: height * width * (n_color + has_alpha) samples of 1 or 2 bytes */
jxl_status jxl_stage_pack(jxl_ctx* ctx, const void* const planes[4], const jxl_pack_params* p, void* out);

/* ---- Modular path: replaces ModularStream.applyTransforms squeeze/RCT branches ---- */
/* Default squeeze parameter list of ModularStream.java:110-131 for a channel list whose
 * first nb_meta channels are meta channels. Returns the count (<= cap) or a negative status. */
int32_t    jxl_modular_default_squeeze_params(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                                              int32_t nb_meta, jxl_squeeze_param* out, int32_t cap);
/* Forward shape replay of ModularStream.java:137-167: given the n_channels image channels,
 * produce the encoded channel list's shapes (count returned; <= cap). */
int32_t    jxl_modular_squeezed_shapes(const int32_t* widths, const int32_t* heights, int32_t n_channels,
                                       const jxl_squeeze_param* sp, int32_t n_sp,
                                       int32_t* out_w, int32_t* out_h, int32_t cap);
/* Upload the encoded channel list (averages + residuals as decoded by the host) and the
 * transform to undo. n_out = number of channels after the inverse. rct_type < 0 = no RCT,
 * otherwise applied on channels rct_begin..+2 after the squeeze. */
jxl_status jxl_modular_begin(jxl_ctx* ctx, const jxl_channel* chans, int32_t n_chans,
                             const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin);
/* enqueue the inverse steps (asynchronous, re-runnable) */
jxl_status jxl_modular_run(jxl_ctx* ctx);
/* number / shape of result channels */
int32_t    jxl_modular_out_count(const jxl_ctx* ctx);
jxl_status jxl_modular_out_shape(const jxl_ctx* ctx, int32_t idx, int32_t* w, int32_t* h);
/* synchronize + copy result channel idx to host */
jxl_status jxl_modular_read_channel(jxl_ctx* ctx, int32_t idx, int32_t* dst);
/* begin + run + read of all channels: out[i].data must hold out w*h elements */
jxl_status jxl_modular_apply(jxl_ctx* ctx, const jxl_channel* chans, int32_t n_chans,
                             const jxl_squeeze_param* sp, int32_t n_sp,
                             int32_t rct_type, int32_t rct_begin,
                             jxl_channel* out, int32_t n_out);
int32_t    jxl_modular_last_launch_count(const jxl_ctx* ctx);
/* how often a plan had to be run again with in-order verification because a segment boundary of the speculative run did not
 * match (diagnostics; the result is exact either way) */
int32_t    jxl_modular_redo_count(const jxl_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* JXLATTE_AMD_H */
