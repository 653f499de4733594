#!/bin/bash
# GPU box: the multistart rollout (TSP-100, 4096 x 8, sampling) with probe definitions compiled in; every variant is built
# with -DRL4CO_MS_PROBE_ONLY (ONE instantiation: TSP, 7 node tiles, sampling) so a build takes seconds, not minutes
#   bash tools/ms_variants.sh "name:-DDEF ..." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
W=$R/tools/probes/_build; mkdir -p $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/rl4co_amd/csrc"
OBJS=""
for o in $R/rl4co_amd/lib/obj/*.o; do
  case $(basename $o) in am_decode_ms.hip.o) ;; *) OBJS="$OBJS $o" ;; esac
done
if [ $# -eq 0 ]; then set -- "base:"; fi
for V in "$@"; do
  NAME=${V%%:*}; DEFS=${V#*:}
  hipcc $FLAGS -DRL4CO_MS_PROBE_ONLY $DEFS -c rl4co_amd/csrc/am_decode_ms.hip -o $W/v_ms_$NAME.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_ms_$NAME.so $W/v_ms_$NAME.o $OBJS
  echo "== $NAME ($DEFS)"
  RL4CO_AMD_LIB=$W/lib_ms_$NAME.so python tools/ms_bench.py 4096 8 ms 2>&1 | tail -1
done
if [ -n "${MS_OLD:-}" ]; then
  hipcc $FLAGS -c tools/probes/old_decode_ms.hip -o $W/v_ms_old.o && hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_ms_old.so $W/v_ms_old.o $OBJS
  echo "== old (r05 kernel)"
  RL4CO_AMD_LIB=$W/lib_ms_old.so python tools/ms_bench.py 4096 8 ms 2>&1 | tail -1
fi
