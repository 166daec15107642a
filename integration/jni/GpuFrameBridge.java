package com.traneptora.jxlatte.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import com.traneptora.jxlatte.color.OpsinInverseMatrix;
import com.traneptora.jxlatte.frame.Frame;
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
 * frame's three float planes with the inverse transforms' output (stage mask JXL_STAGE_IDCT only: Gaborish, EPF and the colour
 * transform stay where the reference has them, so every later line of the reference runs unchanged on the same values).
 *
 * Enabled with -Djxlatte.gpu=1 (or JXLATTE_GPU=1 in the environment); otherwise the patched reference behaves as before.
 *
 * NOT COMPILED OR TESTED IN THIS REPOSITORY (no JDK in the build image). Our source, not reference code. Frames with chroma
 * subsampling are passed on to the reference's own path (the boundary supports them, this first hook does not map them).
 */
public final class GpuFrameBridge {
    private GpuFrameBridge() {}

    private static final boolean ENABLED =
        "1".equals(System.getProperty("jxlatte.gpu")) || "1".equals(System.getenv("JXLATTE_GPU"));
    private static NativeBackend backend; // one context, reused for every frame of the process

    public static boolean enabled(Frame frame) {
        if (!ENABLED)
            return false;
        FrameHeader header = frame.getFrameHeader();
        for (int c = 0; c < 3; c++) {
            if (header.jpegUpsamplingY[c] != 0 || header.jpegUpsamplingX[c] != 0)
                return false;
        }
        return true;
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
    private static ByteBuffer packParams(Frame frame) {
        FrameHeader header = frame.getFrameHeader();
        LFGlobal lfGlobal = frame.getLFGlobal();
        RestorationFilter rf = header.restorationFilter;
        OpsinInverseMatrix matrix = frame.globalMetadata.getOpsinInverseMatrix();
        Dimension padded = frame.getPaddedFrameSize();
        ByteBuffer p = direct(4 * 64);
        p.putInt(padded.width).putInt(padded.height);
        p.putInt(1); // stages = JXL_STAGE_IDCT
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
        p.putInt(0); // xyb: the colour transform stays in Java
        for (int i = 0; i < 9 + 3 + 3; i++)
            p.putFloat(0f); // opsin_matrix, opsin_bias, cbrt_opsin_bias: unused without JXL_STAGE_XYB
        p.putFloat(255f); // intensity_target (unused)
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

    public static synchronized void invertVarDCT(Frame frame, float[][][] buffers, PassGroup[][] passGroups, LFGroup[] lfGroups,
            int numPasses, int numGroups) {
        if (backend == null)
            backend = new NativeBackend(Integer.getInteger("jxlatte.gpu.device", 0));
        NativeBackend nb = backend;
        nb.beginFrame(packParams(frame));
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
        for (int pass = 0; pass < numPasses; pass++) {
            for (int group = 0; group < numGroups; group++) {
                int[][][] q = passGroups[pass][group].hfCoefficients.quantizedCoeffs;
                int gh = q[0].length, gw = gh == 0 ? 0 : q[0][0].length;
                // (the page-locked int16 form -- mapCoeffsI16 / commitCoeffsI16, INTEGRATION.md "The PCIe leg" -- is the fast
                //  path; this first hook keeps the reference's int[3][][] groups as they are)
                nb.putGroup(pass, group, ints(q[0], gh, gw), ints(q[1], gh, gw), ints(q[2], gh, gw), gw, gw, gw);
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
    }
}
