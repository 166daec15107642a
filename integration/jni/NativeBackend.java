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
    public native ByteBuffer[] mapCoeffsI16(int rowsX, int rowsY, int rowsB); // jxl_vardct_map_coeffs_i16 + NewDirectByteBuffer (rows = paddedHeight >> jpegUpsamplingY[c])
    public native void commitCoeffsI16();                          // jxl_vardct_commit_coeffs_i16
    /** Page-locked direct buffers for planes that cross the bus (coefficients in, pixels out). */
    public static native ByteBuffer hostAlloc(long bytes);         // jxl_host_alloc + NewDirectByteBuffer
    public static native void hostFree(ByteBuffer b);              // jxl_host_free(GetDirectBufferAddress(b))
    public native void finishFrame(ByteBuffer outX, ByteBuffer outY, ByteBuffer outB, long stride); // jxl_vardct_finish_frame
    /** Asynchronous form for independent frames decoded side by side (one NativeBackend each): run() enqueues the frame
     *  on its context's stream, readOutput() waits for it. runBatch hands several prepared frames to one call. */
    public native void run();                                      // jxl_vardct_run
    public native void readOutput(ByteBuffer outX, ByteBuffer outY, ByteBuffer outB, long stride); // jxl_vardct_read_output
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
    /** channels: one direct buffer per encoded channel; squeezeParams: 4 ints per step. */
    public native void modularApply(ByteBuffer[] chans, int[] widths, int[] heights, int[] squeezeParams, int rctType,
        int rctBegin, ByteBuffer[] out, int[] outWidths, int[] outHeights); // jxl_modular_apply
}
