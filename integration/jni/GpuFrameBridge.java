package com.traneptora.jxlatte.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import com.traneptora.jxlatte.color.OpsinInverseMatrix;
import com.traneptora.jxlatte.frame.Frame;
import com.traneptora.jxlatte.color.ColorEncodingBundle;
import com.traneptora.jxlatte.frame.FrameFlags;
import com.traneptora.jxlatte.frame.FrameHeader;
import com.traneptora.jxlatte.frame.LFGlobal;
import com.traneptora.jxlatte.frame.features.RestorationFilter;
import com.traneptora.jxlatte.frame.group.LFGroup;
import com.traneptora.jxlatte.frame.group.PassGroup;
import com.traneptora.jxlatte.frame.vardct.HFMetadata;
import com.traneptora.jxlatte.frame.vardct.TransformType;
import com.traneptora.jxlatte.util.Dimension;
import com.traneptora.jxlatte.util.Point;

/**
 * The hook tools/patch_reference_for_gpu.sh inserts into Frame.decodePassGroups (Frame.java:361-374): the loop
 * "passGroup.invertVarDCT(buffers, prev)" over passes and groups becomes ONE call of {@link #invertVarDCT}, which hands the
 * frame's quantised coefficients and side information to libjxlatte_amd.so through {@link NativeBackend} and fills the
 * frame's three float planes with the result. Two forms, chosen with -Djxlatte.gpu=N (or JXLATTE_GPU=N in the environment):
 *
 *   1  the inverse transforms only (stage mask JXL_STAGE_IDCT): Gaborish, EPF and the colour transform stay where the reference
 *      has them, so every later line of the reference runs unchanged on the same values; coefficients go group by group through
 *      putGroup (int32, heap-backed staging). The smallest cut -- and the slowest form of the boundary.
 *   2  the fused cut, the path bench.py measures: inverse transforms + Gaborish + EPF (+ the inverse XYB where nothing sits
 *      between decodeFrame and performColorTransforms: no upsampling, patches, splines, noise, saveBeforeCT, LF level) in one
 *      finishFrame; coefficients written in place into the library's page-locked int16 planes (mapCoeffsI16NoFill +
 *      commitCoeffsI16Groups: three DMA transfers per frame; a group with a sample outside int16 falls back to putGroup). The
 *      script guards the reference's own performGabConvolution / performEdgePreservingFilter (Frame.java:457-461) and invertXYB
 *      (JXLCodestreamDecoder.java:266-267) with the flags this class sets on the frame (gpuRestored, gpuXYB).
 *
 * Otherwise the patched reference behaves as before.
 *
 * NOT COMPILED OR TESTED IN THIS REPOSITORY (no JDK in the build image). Our source, not reference code. Frames with chroma
 * subsampling are passed on to the reference's own path (the boundary supports them, this hook does not map them).
 */
public final class GpuFrameBridge {
    private GpuFrameBridge() {}

    private static final int MODE = mode();
    private static NativeBackend backend; // one context, reused for every frame of the process

    private static int mode() {
        String v = System.getProperty("jxlatte.gpu");
        if (v == null)
            v = System.getenv("JXLATTE_GPU");
        if ("1".equals(v))
            return 1;
        if ("2".equals(v))
            return 2;
        return 0;
    }

    public static boolean enabled(Frame frame) {
        if (MODE == 0)
            return false;
        FrameHeader header = frame.getFrameHeader();
        for (int c = 0; c < 3; c++) {
            if (header.jpegUpsamplingY[c] != 0 || header.jpegUpsamplingX[c] != 0)
                return false;
        }
        return true;
    }

    /** mode 2: may the inverse XYB run inside the fused launch? Only if nothing sits between decodeFrame and performColorTransforms
     *  (JXLCodestreamDecoder.java:615-637) and the frame's planes are not stored as an LF frame / reference before the transform. */
    private static boolean fuseXYB(Frame frame) {
        FrameHeader header = frame.getFrameHeader();
        if (MODE != 2 || !frame.globalMetadata.isXYBEncoded())
            return false;
        if (header.upsampling != 1 || header.lfLevel != 0 || header.type == FrameFlags.LF_FRAME)
            return false;
        if ((header.flags & (FrameFlags.NOISE | FrameFlags.PATCHES | FrameFlags.SPLINES)) != 0)
            return false;
        return !(header.saveBeforeCT && !header.isLast);
    }

    private static ByteBuffer direct(int bytes) {
        return ByteBuffer.allocateDirect(bytes).order(ByteOrder.nativeOrder());
    }

    private static ByteBuffer ints(int[][] a, int h, int w) {
        ByteBuffer b = direct(4 * h * w);
        java.nio.IntBuffer v = b.asIntBuffer();
        for (int y = 0; y < h; y++)
            v.put(a[y], 0, w);
        return b;
    }

    private static ByteBuffer floats(float[][] a, int h, int w) {
        ByteBuffer b = direct(4 * h * w);
        java.nio.FloatBuffer v = b.asFloatBuffer();
        for (int y = 0; y < h; y++)
            v.put(a[y], 0, w);
        return b;
    }

    /** struct jxl_vardct_params (include/jxlatte_amd.h), field by field in declaration order: all members are 4 bytes wide. */
    private static ByteBuffer packParams(Frame frame, boolean xyb) {
        FrameHeader header = frame.getFrameHeader();
        LFGlobal lfGlobal = frame.getLFGlobal();
        RestorationFilter rf = header.restorationFilter;
        OpsinInverseMatrix matrix = frame.globalMetadata.getOpsinInverseMatrix();
        Dimension padded = frame.getPaddedFrameSize();
        ByteBuffer p = direct(4 * 64);
        p.putInt(padded.width).putInt(padded.height);
        p.putInt(MODE == 2 ? (1 | 2 | 4 | (xyb ? 8 : 0)) : 1); // stages: JXL_STAGE_IDCT (| GAB | EPF | XYB: the fused cut)
        float globalScale = 65536.0f / lfGlobal.globalScale; // HFCoefficients.java:270-275
        p.putFloat(globalScale * (float)Math.pow(0.8D, header.xqmScale - 2D));
        p.putFloat(globalScale);
        p.putFloat(globalScale * (float)Math.pow(0.8D, header.bqmScale - 2D));
        for (int c = 0; c < 3; c++)
            p.putFloat(matrix.quantBias[c]);
        p.putFloat(matrix.quantBiasNumerator);
        p.putFloat(lfGlobal.lfChanCorr.baseCorrelationX).putFloat(lfGlobal.lfChanCorr.baseCorrelationB);
        p.putInt(lfGlobal.lfChanCorr.colorFactor);
        // restoration filter fields: carried for completeness, unused under the IDCT-only stage mask
        p.putInt(rf.gab ? 1 : 0);
        for (int c = 0; c < 3; c++)
            p.putFloat(rf.gab1Weights[c]);
        for (int c = 0; c < 3; c++)
            p.putFloat(rf.gab2Weights[c]);
        p.putInt(rf.epfIterations);
        p.putFloat(globalScale);
        for (int i = 0; i < 8; i++)
            p.putFloat(rf.epfSharpLut[i]); // already multiplied by epfQuantMul (RestorationFilter.java:78)
        for (int c = 0; c < 3; c++)
            p.putFloat(rf.epfChannelScale[c]);
        p.putFloat(rf.epfPass0SigmaScale).putFloat(rf.epfPass2SigmaScale).putFloat(rf.epfBorderSadMul);
        p.putInt(xyb ? 1 : 0); // xyb: 0 = the colour transform stays in Java
        // opsin_matrix (adapted to the image's primaries / white point as JXLCodestreamDecoder.java:592-595 does, not yet scaled
        // by 255 / intensityTarget), opsin_bias, cbrt_opsin_bias: read only with JXL_STAGE_XYB. The three fields are private in
        // the reference; tools/patch_reference_for_gpu.sh makes them public.
        ColorEncodingBundle bundle = frame.globalMetadata.getColorEncoding();
        OpsinInverseMatrix adapted = xyb ? matrix.getMatrix(bundle.prim, bundle.white) : matrix;
        for (int i = 0; i < 9; i++)
            p.putFloat(xyb ? adapted.matrix[i / 3][i % 3] : 0f);
        for (int i = 0; i < 3; i++)
            p.putFloat(xyb ? adapted.opsinBias[i] : 0f);
        for (int i = 0; i < 3; i++)
            p.putFloat(xyb ? adapted.cbrtOpsinBias[i] : 0f);
        p.putFloat(xyb ? frame.globalMetadata.getToneMapping().intensityTarget : 255f); // intensity_target
        p.putInt(0).putInt(0); // transfer = JXL_TRANSFER_NONE, out_format = JXL_OUT_F32
        for (int i = 0; i < 6; i++)
            p.putInt(0); // jpeg_upsampling_y / x
        p.flip();
        return p;
    }

    /** HFGlobal.weights as one float array + the 51 element offsets jxl_vardct_set_weights asks for. */
    private static void setWeights(NativeBackend nb, float[][][][] weights) {
        int[] offs = new int[51];
        int total = 0;
        for (int pi = 0; pi < 17; pi++) {
            for (int c = 0; c < 3; c++) {
                offs[pi * 3 + c] = total;
                total += weights[pi][c].length * weights[pi][c][0].length;
            }
        }
        ByteBuffer w = direct(4 * total);
        java.nio.FloatBuffer fb = w.asFloatBuffer();
        for (int pi = 0; pi < 17; pi++) {
            for (int c = 0; c < 3; c++) {
                for (float[] row : weights[pi][c])
                    fb.put(row);
            }
        }
        nb.setWeights(w, offs);
    }

    /** mode 2: the groups written in place into the library's page-locked int16 planes (one pass; every sample of every group,
     *  zeros included: the planes are not zero-filled). A group with a sample outside int16 is left out of the commit and sent
     *  again as int32 (jxl_vardct_put_group overrides its rectangle). */
    private static void putCoefficientsMapped(NativeBackend nb, Frame frame, PassGroup[] groups, int numGroups) {
        ByteBuffer[] planes = nb.mapCoeffsI16NoFill();
        Dimension padded = frame.getPaddedFrameSize();
        java.nio.ShortBuffer[] sb = new java.nio.ShortBuffer[3];
        for (int c = 0; c < 3; c++)
            sb[c] = planes[c].order(ByteOrder.nativeOrder()).asShortBuffer();
        byte[] written = new byte[numGroups];
        short[] row = new short[256];
        for (int group = 0; group < numGroups; group++) {
            int[][][] q = groups[group].hfCoefficients.quantizedCoeffs;
            Point loc = frame.getGroupLocation(group);
            boolean fits = true;
            for (int c = 0; c < 3 && fits; c++) {
                int gh = q[c].length, gw = gh == 0 ? 0 : q[c][0].length;
                for (int y = 0; y < gh && fits; y++) {
                    int[] src = q[c][y];
                    for (int x = 0; x < gw; x++) {
                        int v = src[x];
                        if (v < -32768 || v > 32767) {
                            fits = false;
                            break;
                        }
                        row[x] = (short)v;
                    }
                    if (fits) {
                        sb[c].position(((loc.y << 8) + y) * padded.width + (loc.x << 8));
                        sb[c].put(row, 0, gw);
                    }
                }
            }
            written[group] = (byte)(fits ? 1 : 0);
        }
        nb.commitCoeffsI16Groups(written);
        for (int group = 0; group < numGroups; group++) {
            if (written[group] != 0)
                continue;
            int[][][] q = groups[group].hfCoefficients.quantizedCoeffs;
            int gh = q[0].length, gw = gh == 0 ? 0 : q[0][0].length;
            nb.putGroup(0, group, ints(q[0], gh, gw), ints(q[1], gh, gw), ints(q[2], gh, gw), gw, gw, gw);
        }
    }

    public static synchronized void invertVarDCT(Frame frame, float[][][] buffers, PassGroup[][] passGroups, LFGroup[] lfGroups,
            int numPasses, int numGroups) {
        if (backend == null)
            backend = new NativeBackend(Integer.getInteger("jxlatte.gpu.device", 0));
        NativeBackend nb = backend;
        final boolean xyb = fuseXYB(frame);
        nb.beginFrame(packParams(frame, xyb));
        setWeights(nb, frame.getHFGlobal().weights);
        for (LFGroup lfg : lfGroups) {
            HFMetadata m = lfg.hfMetadata;
            int ch = lfg.size.height, cw = lfg.size.width;
            ByteBuffer sel = direct(ch * cw);
            for (int y = 0; y < ch; y++) {
                for (int x = 0; x < cw; x++) {
                    TransformType tt = m.dctSelect[y][x];
                    sel.put(y * cw + x, (byte)(tt == null ? 0 : tt.type));
                }
            }
            ByteBuffer blocks = direct(8 * m.blockList.length);
            for (Point b : m.blockList)
                blocks.putInt(b.y).putInt(b.x);
            blocks.flip();
            int kh = (ch + 7) / 8, kw = (cw + 7) / 8;
            Point loc = frame.getLFGroupLocation(lfg.lfGroupID);
            float[][][] lf = lfg.lfCoeff.dequantLFCoeff;
            nb.setLFGroup(loc.y, loc.x, ch, cw, sel, ints(m.hfMultiplier, ch, cw), ints(m.hfStreamBuffer[3], ch, cw),
                ints(m.hfStreamBuffer[0], kh, kw), ints(m.hfStreamBuffer[1], kh, kw), blocks, m.blockList.length,
                floats(lf[0], ch, cw), floats(lf[1], ch, cw), floats(lf[2], ch, cw));
        }
        if (MODE == 2 && numPasses == 1) {
            putCoefficientsMapped(nb, frame, passGroups[0], numGroups);
        } else {
            for (int pass = 0; pass < numPasses; pass++) {
                for (int group = 0; group < numGroups; group++) {
                    int[][][] q = passGroups[pass][group].hfCoefficients.quantizedCoeffs;
                    int gh = q[0].length, gw = gh == 0 ? 0 : q[0][0].length;
                    // (pass > 0 accumulates on the device, PassGroup.java:174-200)
                    nb.putGroup(pass, group, ints(q[0], gh, gw), ints(q[1], gh, gw), ints(q[2], gh, gw), gw, gw, gw);
                }
            }
        }
        Dimension padded = frame.getPaddedFrameSize();
        int h = buffers[0].length, w = buffers[0][0].length;
        ByteBuffer[] out = new ByteBuffer[3];
        for (int c = 0; c < 3; c++)
            out[c] = direct(4 * padded.height * padded.width);
        nb.finishFrame(out[0], out[1], out[2], padded.width);
        for (int c = 0; c < 3; c++) {
            java.nio.FloatBuffer fb = out[c].asFloatBuffer();
            for (int y = 0; y < Math.min(h, padded.height); y++) {
                fb.position(y * padded.width);
                fb.get(buffers[c][y], 0, Math.min(w, padded.width));
            }
        }
        // what the reference must NOT do again for this frame (fields added by tools/patch_reference_for_gpu.sh)
        frame.gpuRestored = MODE == 2;
        frame.gpuXYB = xyb;
    }
}
