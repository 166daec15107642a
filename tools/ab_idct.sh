#!/bin/bash
# same-box A/B of the IDCT stage (saturated, 8 frames in flight): tools/ab_idct.sh "<specs>" VARIANT...   VARIANT = name:ENV=V,ENV=V (no env: name:)
# e.g. tools/ab_idct.sh "default DCT8" base:JXL_AMD_LIB=jxlatte_amd/libjxlatte_amd_base.so,JXL_AMD_LIB_OLD=1 new:
SPECS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name ($envs) rep $rep"
  env $(echo $envs | tr ',' ' ') python3 tools/idct_saturated.py --also-single $SPECS | grep -v '^#\|^type'
done
done
