#!/bin/bash
# Build libm3p2i_hip.so variants with extra hipcc flags (here, CPU container) into
# gpurun_variants/<tag>.so; load one on the GPU box with M3P2I_HIP_LIB=.../gpurun_variants/<tag>.so
# usage: tools/flag_variants.sh tag1 "flags1" tag2 "flags2" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_variants
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  python - "$tag" $flags <<'PY' || echo "FAILED $tag"
import sys
from m3p2i_aip_amd import build as b
b.build(force=True, extra_flags=sys.argv[2:], out="gpurun_variants/%s.so" % sys.argv[1])
PY
done
ls -la gpurun_variants
