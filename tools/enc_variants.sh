#!/bin/bash
# A/B of compile-time variants of csrc/am_encoder.hip (RL4CO_ENC_* macros) against a git revision of it.
#   tools/enc_variants.sh build [rev]     build container: tools/probes/_build/lib_<name>.so per variant (+ lib_old.so from <rev>)
#   gpurun -- 'tools/enc_variants.sh run' GPU box: tools/enc_bench.py on every library, three rounds, alternating
# Variant libraries carry ONE instantiation of the fused kernel (-DRL4CO_ENC_PROBE: TSP-100, bf16), so each builds in seconds.
set -e
R=$(cd $(dirname $0)/.. && pwd)
W=$R/tools/probes/_build
mkdir -p $W
# name:compile definitions[:environment at run time]
VARIANTS=${RL4CO_ENC_VARIANTS:-"base:|nogctx:-DRL4CO_ENC_SKIP_GCTX|nofold:-DRL4CO_ENC_SKIP_FOLD|noinit:-DRL4CO_ENC_SKIP_INIT"}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/rl4co_amd/csrc"
if [ "$1" = "build" ]; then
  python -c "import sys; sys.path.insert(0,'$R'); from rl4co_amd import build; build.build_library()"
  OBJS=$(ls $R/rl4co_amd/lib/obj/*.o | grep -v "am_encoder.hip.o")
  if [ -n "$2" ]; then
    git -C $R show $2:rl4co_amd/csrc/am_encoder.hip | sed 's#"common.h"#"'$R'/rl4co_amd/csrc/common.h"#' > $W/am_encoder_old.hip
    hipcc $FLAGS -c $W/am_encoder_old.hip -o $W/am_encoder_old.o &
  fi
  IFS='|' read -ra VS <<< "$VARIANTS"
  for v in "${VS[@]}"; do
    name=${v%%:*}; defs=${v#*:}
    ( hipcc $FLAGS -DRL4CO_ENC_PROBE $defs -c $R/rl4co_amd/csrc/am_encoder.hip -o $W/enc_$name.o && hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_$name.so $W/enc_$name.o $OBJS && echo built lib_$name.so ) &
  done
  wait
  if [ -n "$2" ]; then hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_old.so $W/am_encoder_old.o $OBJS; echo built lib_old.so; fi
else
  LIBS=$(ls $W/lib_*.so)
  for i in 1 2 3; do
    for l in $LIBS; do
      printf "%-22s " $(basename $l); RL4CO_AMD_LIB=$l python $R/tools/enc_bench.py | tail -1
    done
  done
fi
