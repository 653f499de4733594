#!/bin/bash
# GPU box: WHERE does the fused encoder's time go? Semantics-breaking variants of csrc/am_encoder.hip (made by sed from
# the product source into /tmp, never committed) are timed with tools/enc_bench.py; each removes one suspected cost.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/encprobe; mkdir -p $W
SRC=$R/rl4co_amd/csrc
OTHERS="api.hip env_step.hip tour_length.hip am_decode.hip am_decode_ms.hip am_teacher.hip am_teacher_mma.hip am_train_ops.hip am_train_attn.hip am_attn_flash.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$R/include -I$SRC"
build() { # name, sed-script
  cp $SRC/am_encoder.hip $W/enc_$1.hip
  [ -n "$2" ] && sed -i -E "$2" $W/enc_$1.hip
  sed -i 's#"common.h"#"'$SRC'/common.h"#' $W/enc_$1.hip
  ( cd $SRC && hipcc $FLAGS -o $W/lib_$1.so $OTHERS $W/enc_$1.hip ) 2> $W/build_$1.log || { echo "build $1 failed"; tail -3 $W/build_$1.log; }
}
run() { RL4CO_AMD_LIB=$W/lib_$1.so python $R/tools/enc_bench.py 2>&1 | tail -1 | sed "s/^/$1: /"; }
build base ""
build nobarrier 's/^(\s*)__syncthreads\(\);/\1__builtin_amdgcn_sched_barrier(0);/'
build noexp 's/__builtin_amdgcn_exp2f\(sk\[r\]\)/sk[r]/'
build w0 's/return \*reinterpret_cast<const bf16x8\*>\(packed \+ \(\(\(int64_t\)tile \* ksteps \+ ks\) \* 64 \+ lane\) \* 8\);/return *reinterpret_cast<const bf16x8*>(packed + (int64_t)lane * 8);/'
build x0 's/return \*reinterpret_cast<const bf16x8\*>\(xs \+ \(32 \* tt \+ l31\) \* kRS \+ 16 \* ks \+ 8 \* hi\);/return *reinterpret_cast<const bf16x8*>(xs + l31 * kRS + 8 * hi);/'
build nostore 's/^(\s*)\*reinterpret_cast<uint4\*>\(out \+ \(int64_t\)row \* kD \+ 8 \* c16\) = .*/\1;/; s/^(\s*)\*reinterpret_cast<float4\*>\(out \+ \(int64_t\)\(64 \* p \+ row\) \* kD \+ 4 \* c4\) =/\1float4 sink_unused =/'
for v in base nobarrier noexp w0 x0 nostore base; do run $v; done
