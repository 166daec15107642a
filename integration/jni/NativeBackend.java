package com.traneptora.jxlatte.gpu;

import java.nio.ByteBuffer;

/**
 * JNI mirror of include/jxlatte_amd.h for the jxlatte Java host (row f4 of the scope table).
 *
 * NOT COMPILED OR TESTED IN THIS REPOSITORY: the build image has no JDK (no javac, no jni.h). It is the binding a jxlatte
 * maintainer adds; INTEGRATION.md lists the five call sites of the reference it is used from. All bulk data crosses as
 * DIRECT ByteBuffers in native byte order because Java float[][] / int[][] rows are separate heap arrays.
 */
public final class NativeBackend implements AutoCloseable {
    static {
        System.loadLibrary("jxlatte_amd_jni"); // links libjxlatte_amd.so
    }

    private long ctx; // jxl_ctx*

    public NativeBackend(int device) {
        ctx = create(device);
    }

    @Override
    public void close() {
        destroy(ctx);
        ctx = 0;
    }

    private static native long create(int device);                 // jxl_ctx_create
    private static native void destroy(long ctx);                  // jxl_ctx_destroy

    /** params: struct jxl_vardct_params packed by the caller (4-byte fields, C layout). */
    public native void beginFrame(ByteBuffer params);              // jxl_vardct_begin_frame
    public native void setWeights(ByteBuffer weights, int[] offs); // jxl_vardct_set_weights
    public native void setLFGroup(int lfgY, int lfgX, int cellsH, int cellsW, ByteBuffer dctSelect, ByteBuffer hfMul,
        ByteBuffer sharpness, ByteBuffer xFromY, ByteBuffer bFromY, ByteBuffer blockYX, int nBlocks,
        ByteBuffer lfX, ByteBuffer lfY, ByteBuffer lfB);           // jxl_vardct_set_lfgroup
    public native void setLFGroupQuant(int lfgY, int lfgX, int cellsH, int cellsW, ByteBuffer qX, ByteBuffer qY, ByteBuffer qB,
        int extraPrecision, float[] scaledDequant, int xFactorLF, int bFactorLF, boolean adaptiveSmoothing); // ..._lfquant
    public native void putGroup(int pass, int group, ByteBuffer qX, ByteBuffer qY, ByteBuffer qB, int strideX, int strideY,
        int strideB);                                              // jxl_vardct_put_group
    /** int16 wire format: the caller has checked that every coefficient of the group fits (else putGroup). */
    public native void putGroupI16(int pass, int group, ByteBuffer qX, ByteBuffer qY, ByteBuffer qB, int strideX, int strideY,
        int strideB);                                              // jxl_vardct_put_group_i16
    /** The frame's three int16 coefficient planes in page-locked memory the library owns (zero-filled): the entropy
     *  decoder stores its non-zero coefficients in place (row stride = plane width), then commitCoeffsI16(). */
    public native ByteBuffer[] mapCoeffsI16();                     // jxl_vardct_map_coeffs_i16 + jxl_vardct_coeff_plane_rows + NewDirectByteBuffer
    public native int[] coeffPlaneRows();                          // jxl_vardct_coeff_plane_rows -> rows of the three mapped planes
    public native void commitCoeffsI16();                          // jxl_vardct_commit_coeffs_i16
    /** the same planes without the zero-fill: the decoder writes every sample of the groups it then names in commitCoeffsI16Groups */
    public native ByteBuffer[] mapCoeffsI16NoFill();               // jxl_vardct_map_coeffs_i16_ex(JXL_MAP_NO_FILL)
    public native void commitCoeffsI16Groups(byte[] groupWritten); // jxl_vardct_commit_coeffs_i16_groups
    /** Page-locked direct buffers for planes that cross the bus (coefficients in, pixels out). */
    public static native ByteBuffer hostAlloc(long bytes);         // jxl_host_alloc + NewDirectByteBuffer
    public static native void hostFree(ByteBuffer b);              // jxl_host_free(GetDirectBufferAddress(b))
    public native void finishFrame(ByteBuffer outX, ByteBuffer outY, ByteBuffer outB, long stride); // jxl_vardct_finish_frame
    /** Asynchronous form for independent frames decoded side by side (one NativeBackend each): run() enqueues the frame
     *  on its context's stream, readOutput() waits for it. runBatch hands several prepared frames to one call. */
    public native void run();                                      // jxl_vardct_run
    public native void readOutput(ByteBuffer outX, ByteBuffer outY, ByteBuffer outB, long stride); // jxl_vardct_read_output
    /** readOutput in two halves: between them the host may drive the next frame of this context (buffers from hostAlloc) */
    public native void readOutputBegin(ByteBuffer outX, ByteBuffer outY, ByteBuffer outB, long stride); // jxl_vardct_read_output_begin
    public native void readOutputWait();                                                           // jxl_vardct_read_output_wait
    public static void runBatch(NativeBackend[] frames) {          // jxl_vardct_run_batch
        long[] h = new long[frames.length];
        for (int i = 0; i < frames.length; i++) h[i] = frames[i].ctx;
        runBatch0(h);
    }
    private static native void runBatch0(long[] ctxs);
    /** Binning, chroma-from-luma masks and side-table upload of the frame that was just described (setLFGroup calls done);
     *  run() does the same on first use. Lets the host overlap this with the entropy decoding of the HF groups. */
    public native void prepare();                                  // jxl_vardct_prepare

    // ---- the stages between decodeFrame and the blend on planes that stay on the device (JXLCodestreamDecoder.java:628-637)
    public native void planesFromFrame(int height, int width);     // jxl_planes_from_frame (after run(); header.bounds size)
    public native void planesUpload(ByteBuffer p0, ByteBuffer p1, ByteBuffer p2, int height, int width); // jxl_planes_upload
    public native void planesUpsample(int k, float[] weights);     // jxl_planes_upsample   (Frame.upsample)
    public native void planesNoise(int groupDim, long seed0, float[] lut, float baseCorrX, float baseCorrB); // jxl_planes_noise
    public native void planesXYB(float[] matrix, float[] opsinBias, float[] cbrtOpsinBias, float intensityTarget); // jxl_planes_xyb
    public native void planesYCbCr();                              // jxl_planes_ycbcr
    public native int[] planesShape();                             // jxl_planes_shape -> {height, width}
    public native void planesDownload(ByteBuffer p0, ByteBuffer p1, ByteBuffer p2); // jxl_planes_download
    // ---- context, diagnostics
    public static native String version();                         // jxl_version
    public native void synchronize();                              // jxl_ctx_synchronize (JXLCodestreamDecoder.java:637: before the host reads)
    public native long stream();                                   // jxl_ctx_stream (hipStream_t as a long, for callers that own a HIP runtime)
    public native void setStream(long hipStream);                  // jxl_ctx_set_stream
    public native void copyOutputDevice(long dstDevice);           // jxl_vardct_copy_output_device (D2D into an RCCL send buffer)
    public native int outElemSize();                               // jxl_vardct_out_elem_size
    public native int lastLaunchCount();                           // jxl_vardct_last_launch_count
    public native void enableStageTiming(boolean on);              // jxl_vardct_enable_stage_timing
    public native float lastStageMs(int which);                    // jxl_vardct_last_stage_ms (0 frame, 1 inverse transforms, 2 restoration)

    // ---- stage entries: one reference function each, host planes in / out (what the parity tests drive)
    public native void stageIdct2d(ByteBuffer src, ByteBuffer dst, int h, int w, boolean transposed);  // jxl_stage_idct2d (MathHelper.inverseDCT2D)
    public native void stageFdct2d(ByteBuffer src, ByteBuffer dst, int h, int w);                      // jxl_stage_fdct2d (MathHelper.forwardDCT2D)
    public native void stageGab(ByteBuffer i0, ByteBuffer i1, ByteBuffer i2, ByteBuffer o0, ByteBuffer o1, ByteBuffer o2, int h, int w,
        float[] w1, float[] w2);                                   // jxl_stage_gab (Frame.performGabConvolution)
    public native void stageEpf(ByteBuffer i0, ByteBuffer i1, ByteBuffer i2, ByteBuffer o0, ByteBuffer o1, ByteBuffer o2, int h, int w,
        int iterations, ByteBuffer invSigma, float invSigmaModular, float[] channelScale, float pass0SigmaScale, float pass2SigmaScale,
        float borderSadMul);                                       // jxl_stage_epf (Frame.performEdgePreservingFilter)
    public native void stageEpfSigma(ByteBuffer hfMul, ByteBuffer sharpness, int bh, int bw, float globalScale, float[] sharpLut,
        ByteBuffer invSigma);                                      // jxl_stage_epf_sigma (Frame.java:552-571)
    public native void stageLfDequant(int lfgY, int lfgX, int cellsH, int cellsW, ByteBuffer qX, ByteBuffer qY, ByteBuffer qB,
        int extraPrecision, float[] scaledDequant, int xFactorLF, int bFactorLF, boolean adaptiveSmoothing, float baseCorrX,
        float baseCorrB, int colorFactor, ByteBuffer o0, ByteBuffer o1, ByteBuffer o2); // jxl_stage_lf_dequant (LFCoefficients)
    public native void stageXyb(ByteBuffer p0, ByteBuffer p1, ByteBuffer p2, long n, float[] matrix, float[] opsinBias, float[] cbrtOpsinBias,
        float intensityTarget);                                    // jxl_stage_xyb (OpsinInverseMatrix.invertXYB)
    public native void stageYcbcr(ByteBuffer p0, ByteBuffer p1, ByteBuffer p2, long n);               // jxl_stage_ycbcr
    public native void stageTransfer(ByteBuffer in, long n, int transfer, int maxValue, ByteBuffer outF, ByteBuffer outI); // jxl_stage_transfer
    public native void stageInvHSqueeze(ByteBuffer avg, int aw, ByteBuffer res, int rw, int h, ByteBuffer out); // jxl_stage_inv_hsqueeze
    public native void stageInvVSqueeze(ByteBuffer avg, int ah, ByteBuffer res, int rh, int w, ByteBuffer out); // jxl_stage_inv_vsqueeze
    public native void stageRct(ByteBuffer v0, ByteBuffer v1, ByteBuffer v2, long n, int rctType);    // jxl_stage_rct
    public native void stageModularToFloat(ByteBuffer a, ByteBuffer b, long n, float scale, ByteBuffer out); // jxl_stage_modular_to_float
    public native void stageChromaUpsample(ByteBuffer in, int h, int w, int xShift, int yShift, ByteBuffer out); // jxl_stage_chroma_upsample
    public static native float[] upsamplingWeights(int k, float[] packed);                            // jxl_upsampling_weights
    public native void stageUpsample(ByteBuffer in, int h, int w, int k, float[] weights, ByteBuffer out); // jxl_stage_upsample
    public native void stageNoiseInit(int h, int w, int groupDim, long seed0, int colors, ByteBuffer o0, ByteBuffer o1, ByteBuffer o2); // jxl_stage_noise_init
    public native void stageNoiseAdd(ByteBuffer p0, ByteBuffer p1, ByteBuffer p2, ByteBuffer n0, ByteBuffer n1, ByteBuffer n2, long n,
        float[] lut, float baseCorrX, float baseCorrB);            // jxl_stage_noise_add
    /** rect: {h, w, canvasY, canvasX, frameY, frameX, refY, refX}. */
    public native void stageBlend(int mode, int flags, boolean isInt, ByteBuffer canvas, int ch, int cw, ByteBuffer frame, int fh, int fw,
        ByteBuffer ref, int rh, int rw, ByteBuffer frameAlpha, ByteBuffer refAlpha, int[] rect); // jxl_stage_blend
    public native void stageOrient(ByteBuffer in, int h, int w, int orientation, ByteBuffer out);     // jxl_stage_orient
    /** params: {height, width, nColor, hasAlpha, premultiplied, bitDepth, bigEndian, isInt[4], taggedDepth[4]}. */
    public native void stagePack(ByteBuffer[] planes, int[] params, ByteBuffer out);                  // jxl_stage_pack

    // ---- Modular: plan once (begin), run, read channel by channel
    public static native int[] modularDefaultSqueezeParams(int[] widths, int[] heights, int nbMeta);  // jxl_modular_default_squeeze_params
    public static native int[] modularSqueezedShapes(int[] widths, int[] heights, int[] squeezeParams); // jxl_modular_squeezed_shapes
    public native void modularBegin(ByteBuffer[] chans, int[] widths, int[] heights, int[] squeezeParams, int rctType, int rctBegin); // jxl_modular_begin
    public native void modularRun();                               // jxl_modular_run
    public native int modularOutCount();                           // jxl_modular_out_count
    public native int[] modularOutShape(int idx);                  // jxl_modular_out_shape -> {width, height}
    public native void modularReadChannel(int idx, ByteBuffer dst); // jxl_modular_read_channel
    public native int modularLastLaunchCount();                    // jxl_modular_last_launch_count
    public native int modularRedoCount();                          // jxl_modular_redo_count
    /** channels: one direct buffer per encoded channel; squeezeParams: 4 ints per step. */
    public native void modularApply(ByteBuffer[] chans, int[] widths, int[] heights, int[] squeezeParams, int rctType,
        int rctBegin, ByteBuffer[] out, int[] outWidths, int[] outHeights); // jxl_modular_apply
}
