#!/bin/bash
# Pin the oracle against the REAL reference, on a machine that has a JDK (this image has none: `java`, `javac`, `jni.h` absent).
#
#   JXLATTE_SRC=/path/to/jxlatte  tools/pin_oracle_with_jvm.sh  [sample.jxl ...]
#
# 1. copies the reference's java/ tree to a scratch directory (the reference checkout is not touched);
# 2. adds integration/jvm_pin/StageDump.java and inserts five one-line calls to it at the cut points of the hot path
#    (anchored on the statements that call the stages, Frame.java:457-461 and JXLCodestreamDecoder.java:637);
# 3. compiles everything with javac (no Meson needed), decodes every sample with JXLATTE_DUMP_PREFIX set, and also writes
#    the reference's own PFM output (PFMWriter) per sample;
# 4. leaves the dumps under tests/golden/jvm/ -- `python -m pytest tests/test_jvm_pin.py` then compares the oracle with them
#    bit for bit (the test skips while the directory is empty). Commit the dumps you want as fixtures: they are data.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${JXLATTE_SRC:?set JXLATTE_SRC to a checkout of Traneptora/jxlatte}"
command -v javac >/dev/null || { echo "pin_oracle_with_jvm: no javac on PATH (a JDK >= 11 is needed)"; exit 2; }
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$JXLATTE_SRC/java" "$WORK/java"
J=$WORK/java/com/traneptora/jxlatte
cp "$ROOT/integration/jvm_pin/StageDump.java" "$J/util/StageDump.java"
F=$J/frame/Frame.java
D=$J/JXLCodestreamDecoder.java
IMP='import com.traneptora.jxlatte.util.StageDump;'
# Frame.decodeFrame: the four cut points, each anchored on the statement that starts the next stage
sed -i "0,/^import /s//$IMP\nimport /" "$F"
sed -i '/^        invertSubsampling();$/i\        StageDump.dump("idct", buffer);' "$F"
sed -i '/^        if (header.restorationFilter.gab)$/i\        StageDump.dump("sub", buffer);' "$F"
sed -i '/^        if (header.restorationFilter.epfIterations > 0)$/i\        StageDump.dump("gab", buffer);' "$F"
sed -i '/^            performEdgePreservingFilter();$/a\        StageDump.dump("epf", buffer);' "$F"
# JXLCodestreamDecoder.decode: after the colour transform of a frame
sed -i "0,/^import /s//$IMP\nimport /" "$D"
sed -i '/^            performColorTransforms(matrix, frame);$/a\            StageDump.dump("xyb", frame.getBuffer());' "$D"
for pat in 'StageDump.dump("idct"' 'StageDump.dump("sub"' 'StageDump.dump("gab"' 'StageDump.dump("epf"'; do
  grep -q "$pat" "$F" || { echo "pin_oracle_with_jvm: anchor for $pat not found in Frame.java (reference changed?)"; exit 3; }
done
grep -q 'StageDump.dump("xyb"' "$D" || { echo "pin_oracle_with_jvm: anchor not found in JXLCodestreamDecoder.java"; exit 3; }
mkdir -p "$WORK/classes"
find "$WORK/java" -name '*.java' ! -name 'ChebyschevApproximation.java' > "$WORK/sources.txt"   # (not in java/meson.build)
javac --release 11 -d "$WORK/classes" @"$WORK/sources.txt"
cp -r "$JXLATTE_SRC/java/resources/." "$WORK/classes/" 2>/dev/null || true
OUT=$ROOT/tests/golden/jvm
mkdir -p "$OUT"
SAMPLES=("$@")
[ ${#SAMPLES[@]} -gt 0 ] || SAMPLES=("$ROOT"/tests/golden/samples/*.jxl)
java -version 2>&1 | head -1 > "$OUT/JVM_VERSION.txt"
for s in "${SAMPLES[@]}"; do
  b=$(basename "$s" .jxl)
  echo "== $b"
  JXLATTE_DUMP_PREFIX="$OUT/$b" java -cp "$WORK/classes" com.traneptora.jxlatte.JXLatte "$s" "$OUT/$b.pfm" || echo "   (reference failed on $b)"
done
ls "$OUT" | head -40
echo "dumps under $OUT; now: python -m pytest tests/test_jvm_pin.py -q"
