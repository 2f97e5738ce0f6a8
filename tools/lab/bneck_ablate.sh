#!/bin/bash
# CPU (build container): side libraries of the fused bottleneck tail with one memory stream switched off each (bneck_x3.hpp: BNX_ABLATE),
# mcgaze_amd/lab_bnx_<mask>.so = the product objects with engine.hip recompiled.  They travel to the GPU box with the snapshot (*.so is
# git-ignored, not gpurun-ignored); tools/lab/bneck_ablate.py loops each under rocm-smi sampling.  usage: tools/lab/bneck_ablate.sh [masks...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/mcgaze_amd/csrc
make -C $C -j8 > /dev/null
for m in ${@:-1 2 4 8 15}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DBNX_ABLATE=$m -c $C/engine.hip -o /tmp/engine_abl_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mcgaze_amd/lab_bnx_$m.so $C/igemm.o $C/roi_align.o $C/decoder.o /tmp/engine_abl_$m.o $C/preprocess.o $C/build_id.o &&
    echo "built lab_bnx_$m.so" ) &
done
wait
