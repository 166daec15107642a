#!/bin/bash
# Build jxlatte WITH the MI355X back-end hooked in, on a machine that has a JDK (this image has none: `javac`, `jni.h` absent --
# the script has never run; it is the appliable form of INTEGRATION.md section 2, in the style of tools/pin_oracle_with_jvm.sh).
#
#   JXLATTE_SRC=/path/to/jxlatte  [JAVA_HOME=...]  tools/patch_reference_for_gpu.sh  [out-dir]
#
# 1. copies the reference's java/ tree to <out-dir>/java (default: a scratch directory; the checkout is not touched);
# 2. adds integration/jni/NativeBackend.java + GpuFrameBridge.java as package com.traneptora.jxlatte.gpu;
# 3. patches the call sites (text patches, every anchor checked; `--patch-only <java-root>` applies just these to a scratch copy --
#    tests/test_abi.py runs that against /root/reference):
#      a. Frame.decodePassGroups (Frame.java:361-374): the loop `passGroup.invertVarDCT(buffers, prev)` runs only when
#         GpuFrameBridge.enabled(this) is false; otherwise GpuFrameBridge.invertVarDCT(...) sends the frame through
#         libjxlatte_amd.so;
#      b. two flags on Frame (gpuRestored, gpuXYB) that the bridge sets, and guards on the reference's own
#         performGabConvolution / performEdgePreservingFilter (Frame.java:457-461) and OpsinInverseMatrix.invertXYB
#         (JXLCodestreamDecoder.java:266-267): with -Djxlatte.gpu=2 those stages have already run on the device, fused with the
#         inverse transforms (the path bench.py measures); with -Djxlatte.gpu=1 only the inverse transforms move and the guards
#         never fire (so the StageDump hooks of the pin script see the same cut points);
#      c. OpsinInverseMatrix's matrix / opsinBias / cbrtOpsinBias become public (the bridge packs them into jxl_vardct_params);
# 4. javac for the classes, cc for libjxlatte_amd_jni.so (integration/jni/jxlatte_amd_jni.c against include/ and
#    jxlatte_amd/libjxlatte_amd.so);
# 5. prints the command lines that decode with the GPU.
# With PIN=1 the StageDump hooks of tools/pin_patch_reference.sh are added as well, so that ONE build gives both the
# reference's own dumps (run without -Djxlatte.gpu) and the GPU path's (run with it) for tests/test_jvm_pin.py to compare.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)

apply_patches() {  # $1 = java root of a scratch copy
  local J=$1/com/traneptora/jxlatte
  mkdir -p "$J/gpu"
  cp "$ROOT/integration/jni/NativeBackend.java" "$ROOT/integration/jni/GpuFrameBridge.java" "$J/gpu/"
  local F=$J/frame/Frame.java D=$J/JXLCodestreamDecoder.java O=$J/color/OpsinInverseMatrix.java
  once() { [ "$(grep -c -- "$2" "$1")" = "1" ] || { echo "patch_reference_for_gpu: expected exactly one '$2' in $(basename "$1") (reference changed?)"; exit 3; }; }
  # a. the call site: guard the reference's statement, then add the bridge call in front of the pass loop that contains it
  once "$F" '^                    passGroup.invertVarDCT(buffers, prev);$'
  sed -i 's/^                    passGroup.invertVarDCT(buffers, prev);$/                    if (!gpuFrame) passGroup.invertVarDCT(buffers, prev);/' "$F"
  # `buffers[c] = buffer[c].getFloatBuffer();` + the closing brace of its loop precede the pass loop: the bridge call goes behind them
  perl -0pi -e 's/(                buffers\[c\] = buffer\[c\]\.getFloatBuffer\(\);\n            \}\n)/$1            final boolean gpuFrame = com.traneptora.jxlatte.gpu.GpuFrameBridge.enabled(this);\n            if (gpuFrame)\n                com.traneptora.jxlatte.gpu.GpuFrameBridge.invertVarDCT(this, buffers, passGroups, lfGroups, numPasses, numGroups);\n/' "$F"
  once "$F" 'GpuFrameBridge.invertVarDCT(this, buffers, passGroups, lfGroups, numPasses, numGroups);'
  # b. the flags and the guards
  once "$F" '^    private boolean decoded = false;$'
  sed -i 's/^    private boolean decoded = false;$/&\n    public boolean gpuRestored = false, gpuXYB = false; \/\/ set by GpuFrameBridge: stages already run on the device/' "$F"
  once "$F" '^        if (header.restorationFilter.gab)$'
  sed -i 's/^        if (header.restorationFilter.gab)$/        if (header.restorationFilter.gab \&\& !gpuRestored)/' "$F"
  once "$F" '^        if (header.restorationFilter.epfIterations > 0)$'
  sed -i 's/^        if (header.restorationFilter.epfIterations > 0)$/        if (header.restorationFilter.epfIterations > 0 \&\& !gpuRestored)/' "$F"
  once "$D" '^        if (matrix != null)$'
  sed -i 's/^        if (matrix != null)$/        if (matrix != null \&\& !frame.gpuXYB)/' "$D"
  # c. the three opsin fields the bridge reads
  for f in 'float\[\]\[\] matrix' 'float\[\] opsinBias' 'float\[\] cbrtOpsinBias'; do
    once "$O" "^    private final $f;\$"
    sed -i "s/^    private final \($f;\)\$/    public final \1/" "$O"
  done
  once "$F" 'gpuRestored = false, gpuXYB = false;'
  once "$F" 'if (header.restorationFilter.gab && !gpuRestored)'
  once "$F" 'if (header.restorationFilter.epfIterations > 0 && !gpuRestored)'
  once "$D" 'if (matrix != null && !frame.gpuXYB)'
  [ "$(grep -c '^    public final float' "$O")" -ge 5 ] || { echo "patch_reference_for_gpu: opsin fields not made public"; exit 3; }
  echo "patch_reference_for_gpu: call site, 2 flags, 3 guards, 3 fields patched"
}

if [ "${1:-}" = "--patch-only" ]; then
  apply_patches "${2:?usage: --patch-only <java-root>}"
  exit 0
fi
: "${JXLATTE_SRC:?set JXLATTE_SRC to a checkout of Traneptora/jxlatte}"
command -v javac >/dev/null || { echo "patch_reference_for_gpu: no javac on PATH (a JDK >= 11 is needed)"; exit 2; }
JH=${JAVA_HOME:-$(dirname "$(dirname "$(readlink -f "$(command -v javac)")")")}
[ -f "$JH/include/jni.h" ] || { echo "patch_reference_for_gpu: $JH/include/jni.h not found (set JAVA_HOME)"; exit 2; }
[ -f "$ROOT/jxlatte_amd/libjxlatte_amd.so" ] || { echo "patch_reference_for_gpu: build the library first: python -m jxlatte_amd.build"; exit 2; }
OUT=${1:-$(mktemp -d)}
mkdir -p "$OUT"
rm -rf "$OUT/java" "$OUT/classes"
cp -r "$JXLATTE_SRC/java" "$OUT/java"
if [ "${PIN:-0}" = 1 ]; then  # (first: its anchors are the unguarded statements)
  bash "$ROOT/tools/pin_patch_reference.sh" "$OUT/java"
fi
apply_patches "$OUT/java"
mkdir -p "$OUT/classes"
find "$OUT/java" -name '*.java' ! -name 'ChebyschevApproximation.java' > "$OUT/sources.txt"   # (not in java/meson.build)
javac --release 11 -d "$OUT/classes" @"$OUT/sources.txt"
cp -r "$JXLATTE_SRC/java/resources/." "$OUT/classes/" 2>/dev/null || true
cc -O2 -fPIC -shared -I"$JH/include" -I"$JH/include/linux" -I"$ROOT/include" "$ROOT/integration/jni/jxlatte_amd_jni.c" \
   -L"$ROOT/jxlatte_amd" -ljxlatte_amd -Wl,-rpath,"$ROOT/jxlatte_amd" -o "$OUT/libjxlatte_amd_jni.so"
echo "built: $OUT/classes, $OUT/libjxlatte_amd_jni.so"
echo "decode on the GPU : java -Djxlatte.gpu=2 -Djava.library.path=$OUT -cp $OUT/classes com.traneptora.jxlatte.JXLatte in.jxl out.png   (fused: IDCT + Gab + EPF + XYB on the device, int16 mapped planes)"
echo "inverse transforms only: -Djxlatte.gpu=1"
echo "decode in Java    : java -cp $OUT/classes com.traneptora.jxlatte.JXLatte in.jxl out.png"
echo "pin both (PIN=1)  : JXLATTE_DUMP_PREFIX=tests/golden/jvm/<name> with and without -Djxlatte.gpu=1, then python -m pytest tests/test_jvm_pin.py"
