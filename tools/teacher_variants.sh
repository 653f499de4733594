#!/bin/bash
# GPU box: the teacher backward (TSP-100, 4096 x 8) as built, and with the r06 probes compiled in (tools/kernel_variant.sh)
#   bash tools/teacher_variants.sh "name:-DDEF ..." ...      (default: the product build and the no-packing probe)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
if [ $# -eq 0 ]; then set -- "base:" "nopack:-DRL4CO_TEACHER_NOPACK"; fi
for V in "$@"; do
  NAME=${V%%:*}; DEFS=${V#*:}
  bash tools/kernel_variant.sh am_teacher_mma.hip tm_$NAME "$DEFS" > /dev/null 2>&1
  echo "== $NAME ($DEFS)"
  RL4CO_AMD_LIB=$R/tools/probes/_build/lib_tm_$NAME.so python tools/teacher_bench.py 4096 8 100 mma 2>&1 | tail -1
done
