#!/bin/bash
# Build libm3p2i_hip.so variants with extra hipcc flags (here, CPU container) into
# gpurun_variants/<tag>.so; tools/time_variants.sh times them on the GPU box.
# usage: tools/flag_variants.sh tag1 "flags1" tag2 "flags2" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_variants
C=m3p2i_aip_amd/csrc
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
     $flags $C/rollout_point.hip $C/rollout_panda.hip $C/update.hip $C/sampler.hip $C/m3_api.hip -o gpurun_variants/$tag.so || echo "FAILED $tag"
done
ls -la gpurun_variants
