#!/bin/bash
# GPU box: LDS row-stride probe for the two training latency chains (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
# = 0.44 - 0.50). Variant libraries with another row padding (made by sed from the product source into /tmp, never
# committed) are timed with tools/teacher_bench.py / tools/ms_bench.py.
# Reasoning (guide: ds_read_b64 = 2 lane groups of 32, bank = (addr / 4) mod 64): with rows of 68 dwords (pad 8) the
# natural operand reads (row = lane & 15, 8 bytes at 4 g) are conflict-free but the transpose reads (rows 4 g + (lane >> 2) & 3,
# chunk lane & 3) fold 32 lanes onto 18 bank pairs; rows of 72 dwords (pad 16) swap the two patterns, and the transpose
# reads are two thirds of the teacher kernel's LDS reads.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/padprobe; mkdir -p $W
SRC=$R/rl4co_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$R/include -I$SRC"
ALL=$(cd $SRC && ls *.hip)
build() { # name, file, sed-script
  OTHERS=$(for f in $ALL; do [ "$f" != "$2" ] && [ "$f" != "${2%.hip}_f16.hip" ] && echo $f; done)
  cp $SRC/$2 $W/v_$1.hip
  sed -i -E "$3" $W/v_$1.hip
  sed -i 's#"common.h"#"'$SRC'/common.h"#; s#"elem16.h"#"'$SRC'/elem16.h"#; s#"rl4co_math.h"#"'$SRC'/rl4co_math.h"#' $W/v_$1.hip
  # the _f16 wrapper includes its bf16 namesake: point it at the variant
  F16=${2%.hip}_f16.hip
  sed -E 's#"'$2'"#"'$W/v_$1.hip'"#; s#"elem16.h"#"'$SRC'/elem16.h"#' $SRC/$F16 > $W/v_$1_f16.hip
  ( cd $SRC && hipcc $FLAGS -o $W/lib_$1.so $OTHERS $W/v_$1.hip $W/v_$1_f16.hip ) 2> $W/build_$1.log || { echo "build $1 failed"; tail -5 $W/build_$1.log; }
}
build t8 am_teacher_mma.hip 's/^constexpr int kRS = kD \+ 8;/constexpr int kRS = kD + 8;/'
build t16 am_teacher_mma.hip 's/^constexpr int kRS = kD \+ 8;/constexpr int kRS = kD + 16;/'
build m16 am_decode_ms.hip 's/^constexpr int kRS = kD \+ 8;/constexpr int kRS = kD + 16;/'
for v in t8 t16 t8 t16; do RL4CO_AMD_LIB=$W/lib_$v.so python $R/tools/teacher_bench.py 4096 8 100 2>&1 | grep "\[mma\]" | sed "s/^/$v: /"; done
for v in t8 m16 t8 m16; do RL4CO_AMD_LIB=$W/lib_$v.so python $R/tools/ms_bench.py 4096 8 ms 2>&1 | tail -1 | sed "s/^/$v: /"; done
