#!/bin/bash
# GPU box: the r05 teacher kernel (tools/probes/old_teacher_mma.hip) in the same micro-benchmark as tools/teacher_variants.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
W=$R/tools/probes/_build; mkdir -p $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/rl4co_amd/csrc"
OBJS=""
for o in $R/rl4co_amd/lib/obj/*.o; do
  case $(basename $o) in am_teacher_mma.hip.o) ;; *) OBJS="$OBJS $o" ;; esac
done
hipcc $FLAGS -c tools/probes/old_teacher_mma.hip -o $W/v_old.o && hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_tm_old.so $W/v_old.o $OBJS
echo "== old (r05 kernel)"
RL4CO_AMD_LIB=$W/lib_tm_old.so python tools/teacher_bench.py 4096 8 100 mma 2>&1 | tail -1
RL4CO_AMD_LIB=$W/lib_tm_old.so python tools/teacher_bench.py 4096 8 100 mma 2>&1 | tail -1
echo "== new"
python tools/teacher_bench.py 4096 8 100 mma 2>&1 | tail -1
