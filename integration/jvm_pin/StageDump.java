// Pin-on-arrival harness for the oracle (SURVEY.md section 8(c): the reference ships no golden vectors, and no JVM exists in the
// build image, so the oracle of this repository is pinned by nothing the reference holds). This class is NEW code, not part
// of jxlatte: tools/pin_patch_reference.sh drops it into a scratch copy of the reference tree next to
// com/traneptora/jxlatte/util/ImageBuffer.java and inserts one-line calls to it at the cut points of the hot path:
//   Frame.decodeFrame          "mod"  the modular stream's channels after applyTransforms (inverse squeeze / RCT / palette)
//                              "idct" the frame buffers after the modular -> buffer loop: for a VarDCT frame the output of the
//                                     inverse transforms (+ its modular extra channels), for a Modular frame the converted planes
//                              "sub"  after invertSubsampling, "gab" after Gaborish, "epf" after the edge-preserving filter
//   JXLCodestreamDecoder       "xyb"  the frame's buffers after performColorTransforms
//   PNGWriter (constructor)    "tf"   the image after JXLImage.transform (transfer function applied, float)
//                              "int"  the planes the IDAT writer reads (after castToIntWithMax / clamp)
// Run on the sample bitstreams it writes the reference's own intermediate planes, which tests/test_jvm_pin.py then compares
// bit for bit with the oracle's (every stage of VarDCT AND Modular frames, and the PNG output stage).
//
// File format (little endian): int32 magic 0x3144584A ("JXD1"), int32 type (0 int, 1 float), int32 height, int32 width,
// then height*width 4-byte samples, row major. One file per frame, stage and channel:
//   $JXLATTE_DUMP_PREFIX.f<frame>.<stage>.c<channel>.bin        frame stages ("mod" opens a new frame: every decodeFrame runs it)
//   $JXLATTE_DUMP_PREFIX.png.<stage>.c<channel>.bin             PNG stages
package com.traneptora.jxlatte.util;

import java.io.FileOutputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.channels.FileChannel;

public final class StageDump {
    private static final String PREFIX = System.getenv("JXLATTE_DUMP_PREFIX");
    private static int frame = -1;

    private StageDump() {}

    private interface Row {
        void put(ByteBuffer bb, int y);
    }

    private static void write(String name, boolean isInt, int height, int width, Row row) {
        try (FileOutputStream out = new FileOutputStream(name); FileChannel ch = out.getChannel()) {
            ByteBuffer bb = ByteBuffer.allocate(16 + 4 * Math.max(width, 1)).order(ByteOrder.LITTLE_ENDIAN);
            bb.putInt(0x3144584A).putInt(isInt ? 0 : 1).putInt(height).putInt(width);
            bb.flip();
            ch.write(bb);
            for (int y = 0; y < height; y++) {
                bb.clear();
                row.put(bb, y);
                bb.flip();
                ch.write(bb);
            }
        } catch (IOException e) {
            throw new RuntimeException("StageDump: cannot write " + name, e);
        }
    }

    private static void writeBuffers(String stem, ImageBuffer[] buffers) {
        for (int c = 0; c < buffers.length; c++) {
            final ImageBuffer b = buffers[c];
            if (b == null)
                continue;
            String name = String.format("%s.c%d.bin", stem, c);
            if (b.isInt()) {
                final int[][] p = b.getIntBuffer();
                write(name, true, b.height, b.width, (bb, y) -> { for (int x = 0; x < b.width; x++) bb.putInt(p[y][x]); });
            } else {
                final float[][] p = b.getFloatBuffer();
                write(name, false, b.height, b.width, (bb, y) -> { for (int x = 0; x < b.width; x++) bb.putInt(Float.floatToRawIntBits(p[y][x])); });
            }
        }
    }

    /** the modular stream's decoded channels (ModularStream.getDecodedBuffer): opens a new frame */
    public static synchronized void dumpInt(String stage, int[][][] planes) {
        if (PREFIX == null)
            return;
        if (stage.equals("mod"))
            frame++;
        if (planes == null)
            return;
        for (int c = 0; c < planes.length; c++) {
            final int[][] p = planes[c];
            if (p == null)
                continue;
            final int h = p.length, w = h > 0 ? p[0].length : 0;
            write(String.format("%s.f%d.%s.c%d.bin", PREFIX, Math.max(frame, 0), stage, c), true, h, w,
                (bb, y) -> { for (int x = 0; x < w; x++) bb.putInt(p[y][x]); });
        }
    }

    /** a frame stage: the frame's buffers as they stand */
    public static synchronized void dump(String stage, ImageBuffer[] buffers) {
        if (PREFIX == null || buffers == null)
            return;
        writeBuffers(String.format("%s.f%d.%s", PREFIX, Math.max(frame, 0), stage), buffers);
    }

    /** a PNG-writer stage */
    public static synchronized void dumpImage(String stage, ImageBuffer[] buffers) {
        if (PREFIX == null || buffers == null)
            return;
        writeBuffers(String.format("%s.png.%s", PREFIX, stage), buffers);
    }
}
