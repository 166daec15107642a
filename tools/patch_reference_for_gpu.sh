#!/bin/bash
# Build jxlatte WITH the MI355X back-end hooked in, on a machine that has a JDK (this image has none: `javac`, `jni.h` absent --
# the script has never run; it is the appliable form of INTEGRATION.md section 2, in the style of tools/pin_oracle_with_jvm.sh).
#
#   JXLATTE_SRC=/path/to/jxlatte  [JAVA_HOME=...]  tools/patch_reference_for_gpu.sh  [out-dir]
#
# 1. copies the reference's java/ tree to <out-dir>/java (default: a scratch directory; the checkout is not touched);
# 2. adds integration/jni/NativeBackend.java + GpuFrameBridge.java as package com.traneptora.jxlatte.gpu;
# 3. patches ONE call site: the loop `passGroup.invertVarDCT(buffers, prev)` at the end of Frame.decodePassGroups
#    (Frame.java:361-374) runs only when GpuFrameBridge.enabled(this) is false; otherwise GpuFrameBridge.invertVarDCT(...)
#    sends the frame through libjxlatte_amd.so (stage mask IDCT: Gab / EPF / colour stay in Java, so the StageDump hooks of the
#    pin script see the same cut points);
# 4. javac for the classes, cc for libjxlatte_amd_jni.so (integration/jni/jxlatte_amd_jni.c against include/ and
#    jxlatte_amd/libjxlatte_amd.so);
# 5. prints the command line that decodes with the GPU: java -Djxlatte.gpu=1 -Djava.library.path=... -cp ... JXLatte in.jxl out.png
# With PIN=1 the five StageDump hooks of tools/pin_oracle_with_jvm.sh are added as well, so that ONE build gives both the
# reference's own dumps (run without -Djxlatte.gpu) and the GPU path's (run with it) for tests/test_jvm_pin.py to compare.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${JXLATTE_SRC:?set JXLATTE_SRC to a checkout of Traneptora/jxlatte}"
command -v javac >/dev/null || { echo "patch_reference_for_gpu: no javac on PATH (a JDK >= 11 is needed)"; exit 2; }
JH=${JAVA_HOME:-$(dirname "$(dirname "$(readlink -f "$(command -v javac)")")")}
[ -f "$JH/include/jni.h" ] || { echo "patch_reference_for_gpu: $JH/include/jni.h not found (set JAVA_HOME)"; exit 2; }
[ -f "$ROOT/jxlatte_amd/libjxlatte_amd.so" ] || { echo "patch_reference_for_gpu: build the library first: python -m jxlatte_amd.build"; exit 2; }
OUT=${1:-$(mktemp -d)}
mkdir -p "$OUT"
rm -rf "$OUT/java" "$OUT/classes"
cp -r "$JXLATTE_SRC/java" "$OUT/java"
J=$OUT/java/com/traneptora/jxlatte
mkdir -p "$J/gpu"
cp "$ROOT/integration/jni/NativeBackend.java" "$ROOT/integration/jni/GpuFrameBridge.java" "$J/gpu/"
F=$J/frame/Frame.java
# the call site: guard the reference's statement, then add the bridge call in front of the pass loop that contains it.
# Anchors are the statements themselves (unique in Frame.java); the script stops if the reference has changed.
grep -q '^                    passGroup.invertVarDCT(buffers, prev);$' "$F" || { echo "patch_reference_for_gpu: anchor invertVarDCT not found in Frame.java"; exit 3; }
sed -i 's/^                    passGroup.invertVarDCT(buffers, prev);$/                    if (!gpuFrame) passGroup.invertVarDCT(buffers, prev);/' "$F"
# `buffers[c] = buffer[c].getFloatBuffer();` + the closing brace of its loop precede the pass loop: the bridge call goes behind them
perl -0pi -e 's/(                buffers\[c\] = buffer\[c\]\.getFloatBuffer\(\);\n            \}\n)/$1            final boolean gpuFrame = com.traneptora.jxlatte.gpu.GpuFrameBridge.enabled(this);\n            if (gpuFrame)\n                com.traneptora.jxlatte.gpu.GpuFrameBridge.invertVarDCT(this, buffers, passGroups, lfGroups, numPasses, numGroups);\n/' "$F"
grep -q 'GpuFrameBridge.invertVarDCT(this, buffers, passGroups, lfGroups, numPasses, numGroups);' "$F" || { echo "patch_reference_for_gpu: anchor getFloatBuffer loop not found in Frame.java"; exit 3; }
if [ "${PIN:-0}" = 1 ]; then
  cp "$ROOT/integration/jvm_pin/StageDump.java" "$J/util/StageDump.java"
  D=$J/JXLCodestreamDecoder.java
  IMP='import com.traneptora.jxlatte.util.StageDump;'
  sed -i "0,/^import /s//$IMP\nimport /" "$F"
  sed -i '/^        invertSubsampling();$/i\        StageDump.dump("idct", buffer);' "$F"
  sed -i '/^        if (header.restorationFilter.gab)$/i\        StageDump.dump("sub", buffer);' "$F"
  sed -i '/^        if (header.restorationFilter.epfIterations > 0)$/i\        StageDump.dump("gab", buffer);' "$F"
  sed -i '/^            performEdgePreservingFilter();$/a\        StageDump.dump("epf", buffer);' "$F"
  sed -i "0,/^import /s//$IMP\nimport /" "$D"
  sed -i '/^            performColorTransforms(matrix, frame);$/a\            StageDump.dump("xyb", frame.getBuffer());' "$D"
fi
mkdir -p "$OUT/classes"
find "$OUT/java" -name '*.java' ! -name 'ChebyschevApproximation.java' > "$OUT/sources.txt"   # (not in java/meson.build)
javac --release 11 -d "$OUT/classes" @"$OUT/sources.txt"
cp -r "$JXLATTE_SRC/java/resources/." "$OUT/classes/" 2>/dev/null || true
cc -O2 -fPIC -shared -I"$JH/include" -I"$JH/include/linux" -I"$ROOT/include" "$ROOT/integration/jni/jxlatte_amd_jni.c" \
   -L"$ROOT/jxlatte_amd" -ljxlatte_amd -Wl,-rpath,"$ROOT/jxlatte_amd" -o "$OUT/libjxlatte_amd_jni.so"
echo "built: $OUT/classes, $OUT/libjxlatte_amd_jni.so"
echo "decode on the GPU : java -Djxlatte.gpu=1 -Djava.library.path=$OUT -cp $OUT/classes com.traneptora.jxlatte.JXLatte in.jxl out.png"
echo "decode in Java    : java -cp $OUT/classes com.traneptora.jxlatte.JXLatte in.jxl out.png"
echo "pin both (PIN=1)  : JXLATTE_DUMP_PREFIX=tests/golden/jvm/<name> with and without -Djxlatte.gpu=1, then python -m pytest tests/test_jvm_pin.py"
