#!/bin/bash
# Text patches of the pin-on-arrival harness: tools/pin_patch_reference.sh <java-root>   (a SCRATCH COPY of the reference's java/ tree)
# Adds integration/jvm_pin/StageDump.java and one-line calls to it at the cut points of the hot path (see StageDump.java).
# Every anchor is checked: exit 3 names the one that no longer matches the reference. Needs only sed / grep (tests/test_jvm_pin.py
# runs it against /root/reference where that exists; the build + run on a JDK box is tools/pin_oracle_with_jvm.sh).
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
JROOT=${1:?usage: pin_patch_reference.sh <java-root>}
J=$JROOT/com/traneptora/jxlatte
cp "$ROOT/integration/jvm_pin/StageDump.java" "$J/util/StageDump.java"
F=$J/frame/Frame.java
D=$J/JXLCodestreamDecoder.java
P=$J/io/PNGWriter.java
IMP='import com.traneptora.jxlatte.util.StageDump;'
for f in "$F" "$D" "$P"; do sed -i "0,/^import /s//$IMP\nimport /" "$f"; done
# Frame.decodeFrame (:427-461): the modular channels, then the four buffer stages, each anchored on the statement that follows it
sed -i '/^        int\[\]\[\]\[\] modularBuffer = lfGlobal.globalModular.getDecodedBuffer();$/a\        StageDump.dumpInt("mod", modularBuffer);' "$F"
sed -i '/^        invertSubsampling();$/i\        StageDump.dump("idct", buffer);' "$F"
sed -i '/^        if (header.restorationFilter.gab)$/i\        StageDump.dump("sub", buffer);' "$F"
sed -i '/^        if (header.restorationFilter.epfIterations > 0)$/i\        StageDump.dump("gab", buffer);' "$F"
sed -i '/^            performEdgePreservingFilter();$/a\        StageDump.dump("epf", buffer);' "$F"
# JXLCodestreamDecoder.decode (:637): after the colour transform of a frame
sed -i '/^            performColorTransforms(matrix, frame);$/a\            StageDump.dump("xyb", frame.getBuffer());' "$D"
# PNGWriter constructor (:65, :105-111): after JXLImage.transform, and after the cast / clamp loop (anchored on its closing lines)
sed -i '/^        image = iccProfile != null ? image : image.transform(primaries, whitePoint, tf, peakDetect);$/a\        StageDump.dumpImage("tf", image.getBuffer(false));' "$P"
sed -i '/^                buffer\[c\].castToIntWithMax(maxValue);$/{n;n;a\        StageDump.dumpImage("int", buffer);
}' "$P"
chk() { [ "$(grep -c -- "$2" "$1")" = "1" ] || { echo "pin_patch_reference: anchor for $2 not found exactly once in $(basename "$1") (reference changed?)"; exit 3; }; }
chk "$F" 'StageDump.dumpInt("mod", modularBuffer);'
chk "$F" 'StageDump.dump("idct", buffer);'
chk "$F" 'StageDump.dump("sub", buffer);'
chk "$F" 'StageDump.dump("gab", buffer);'
chk "$F" 'StageDump.dump("epf", buffer);'
chk "$D" 'StageDump.dump("xyb", frame.getBuffer());'
chk "$P" 'StageDump.dumpImage("tf", image.getBuffer(false));'
chk "$P" 'StageDump.dumpImage("int", buffer);'
echo "pin_patch_reference: 8 hooks placed"
