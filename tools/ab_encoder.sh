#!/bin/bash
# A/B of csrc/am_encoder.hip against a git revision of it: builds tools/probes/_build/lib_old.so from REV's source HERE
# (hipcc cross-compiles), the GPU box then runs tools/enc_layers.py on both libraries, alternating.
#   tools/ab_encoder.sh build <rev>      (build container)
#   gpurun -- 'tools/ab_encoder.sh run'  (GPU box)
set -e
R=$(cd $(dirname $0)/.. && pwd)
W=$R/tools/probes/_build
mkdir -p $W
if [ "$1" = "build" ]; then
  git -C $R show $2:rl4co_amd/csrc/am_encoder.hip | sed 's#"common.h"#"'$R'/rl4co_amd/csrc/common.h"#' > $W/am_encoder_old.hip
  python -c "import sys; sys.path.insert(0,'$R'); from rl4co_amd import build; build.build_library()"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/rl4co_amd/csrc -c $W/am_encoder_old.hip -o $W/am_encoder_old.o
  OBJS=$(ls $R/rl4co_amd/lib/obj/*.o | grep -v "am_encoder.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_old.so $W/am_encoder_old.o $OBJS
  echo built $W/lib_old.so
else
  for i in 1 2 3; do
    echo "--- old"; RL4CO_AMD_LIB=$W/lib_old.so python $R/tools/enc_layers.py | tr '\n' ' '; echo
    echo "--- new"; python $R/tools/enc_layers.py | tr '\n' ' '; echo
  done
fi
