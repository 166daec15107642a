// Pin-on-arrival harness for the oracle (SURVEY.md section 8(c): the reference ships no golden vectors, and no JVM exists in the
// build image, so the oracle of this repository is pinned by nothing the reference holds). This class is NEW code, not part
// of jxlatte: tools/pin_oracle_with_jvm.sh drops it into a scratch copy of the reference tree next to
// com/traneptora/jxlatte/util/ImageBuffer.java and inserts five one-line calls to it at the cut points of the hot path
// (Frame.decodeFrame: after the inverse transforms, after invertSubsampling, after Gaborish, after the EPF; and
// JXLCodestreamDecoder.decode: after performColorTransforms). Run on the sample bitstreams it writes the reference's own
// intermediate planes, which tests/test_jvm_pin.py then compares bit for bit with the oracle's.
//
// File format (little endian): int32 magic 0x3144584A ("JXD1"), int32 type (0 int, 1 float), int32 height, int32 width,
// then height*width 4-byte samples, row major. One file per frame, stage and channel:
//   $JXLATTE_DUMP_PREFIX.f<frame>.<stage>.c<channel>.bin
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

    /** stage "idct" opens a new frame; every later stage of the same frame reuses its index */
    public static synchronized void dump(String stage, ImageBuffer[] buffers) {
        if (PREFIX == null || buffers == null)
            return;
        if (stage.equals("idct"))
            frame++;
        for (int c = 0; c < buffers.length; c++) {
            ImageBuffer b = buffers[c];
            if (b == null)
                continue;
            String name = String.format("%s.f%d.%s.c%d.bin", PREFIX, Math.max(frame, 0), stage, c);
            try (FileOutputStream out = new FileOutputStream(name); FileChannel ch = out.getChannel()) {
                ByteBuffer bb = ByteBuffer.allocate(16 + 4 * b.width).order(ByteOrder.LITTLE_ENDIAN);
                bb.putInt(0x3144584A).putInt(b.isInt() ? 0 : 1).putInt(b.height).putInt(b.width);
                bb.flip();
                ch.write(bb);
                for (int y = 0; y < b.height; y++) {
                    bb.clear();
                    if (b.isInt()) {
                        int[] row = b.getIntBuffer()[y];
                        for (int x = 0; x < b.width; x++)
                            bb.putInt(row[x]);
                    } else {
                        float[] row = b.getFloatBuffer()[y];
                        for (int x = 0; x < b.width; x++)
                            bb.putInt(Float.floatToRawIntBits(row[x]));
                    }
                    bb.flip();
                    ch.write(bb);
                }
            } catch (IOException e) {
                throw new RuntimeException("StageDump: cannot write " + name, e);
            }
        }
    }
}
