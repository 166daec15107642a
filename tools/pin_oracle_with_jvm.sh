#!/bin/bash
# Pin the oracle against the REAL reference, on a machine that has a JDK (this image has none: `java`, `javac`, `jni.h` absent).
#
#   JXLATTE_SRC=/path/to/jxlatte  tools/pin_oracle_with_jvm.sh  [sample.jxl ...]
#
# 1. copies the reference's java/ tree to a scratch directory (the reference checkout is not touched);
# 2. tools/pin_patch_reference.sh adds integration/jvm_pin/StageDump.java and eight one-line calls to it at the cut points of the
#    hot path: the modular stream after applyTransforms, the frame buffers after the modular -> buffer loop / invertSubsampling /
#    Gaborish / EPF (Frame.java:427-461), after performColorTransforms (JXLCodestreamDecoder.java:637), and in the PNG writer
#    after JXLImage.transform and after the integer cast (PNGWriter.java:65,105-111);
# 3. compiles everything with javac (no Meson needed) and decodes every sample to PNG with JXLATTE_DUMP_PREFIX set;
# 4. leaves the dumps under tests/golden/jvm/ -- `python -m pytest tests/test_jvm_pin.py` then compares the oracle with them
#    bit for bit, VarDCT and Modular frames and the PNG stage alike (the test skips while the directory is empty). Commit the
#    dumps you want as fixtures: they are data.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${JXLATTE_SRC:?set JXLATTE_SRC to a checkout of Traneptora/jxlatte}"
command -v javac >/dev/null || { echo "pin_oracle_with_jvm: no javac on PATH (a JDK >= 11 is needed)"; exit 2; }
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$JXLATTE_SRC/java" "$WORK/java"
bash "$ROOT/tools/pin_patch_reference.sh" "$WORK/java"
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
  JXLATTE_DUMP_PREFIX="$OUT/$b" java -cp "$WORK/classes" com.traneptora.jxlatte.JXLatte "$s" "$OUT/$b.png" || echo "   (reference failed on $b)"
done
ls "$OUT" | head -40
echo "dumps under $OUT; now: python -m pytest tests/test_jvm_pin.py -q"
