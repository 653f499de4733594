#!/bin/bash
# Where do the 1.7 ms go that the TRAINING forward (am_encoder_kernel<.., TRAIN>) takes over the inference kernel's six
# layers? Variants of csrc/am_encoder.hip with one group of saves removed (semantics-breaking, made by sed into
# tools/probes/_build, never committed), built HERE (hipcc cross-compiles) against the product's other objects; the GPU box
# times the training encoder's forward with each (tools/train_encoder_bench.py prints GPU forward ms).
#   tools/probes/train_fwd_probe.sh build     (build container, ~3 min per variant, in parallel)
#   gpurun -- 'tools/probes/train_fwd_probe.sh run'
set -e
R=$(cd $(dirname $0)/../.. && pwd)
W=$R/tools/probes/_build
mkdir -p $W
SRC=$R/rl4co_amd/csrc
variant() { # name, sed script
  sed -E "$2" $SRC/am_encoder.hip | sed 's#"common.h"#"'$SRC'/common.h"#' > $W/enc_$1.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$SRC -c $W/enc_$1.hip -o $W/enc_$1.o
  OBJS=$(ls $R/rl4co_amd/lib/obj/*.o | grep -v "am_encoder.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_$1.so $W/enc_$1.o $OBJS
  echo built $1
}
if [ "$1" = "build" ]; then
  python -c "import sys; sys.path.insert(0,'$R'); from rl4co_amd import build; build.build_library()"
  variant nov 's/if \(tok < N\) vdst\[\(int64_t\)tok \* 3 \* kD\] = \(E\)acc\[tt\]\[r\];/if (tok < 0) vdst[(int64_t)tok * 3 * kD] = (E)acc[tt][r];/' &
  variant noqk 's/if constexpr \(TRAIN\) save_t<TT>\(ts\.qkv/if constexpr (false) save_t<TT>(ts.qkv/' &
  variant noh 's/if constexpr \(TRAIN\) rows_out\(ys, ts\.h/if constexpr (false) rows_out(ys, ts.h/' &
  variant norows 's/if constexpr \(TRAIN\) rows_out\(/if constexpr (false) rows_out(/' &
  variant noy 's/if \(32 \* tt \+ l31 < N\) \*reinterpret_cast<vec4<E>\*>\(y_out/if (32 * tt + l31 < 0) *reinterpret_cast<vec4<E>*>(y_out/' &
  wait
else
  for v in base nov noqk noh norows noy; do
    if [ $v = base ]; then L=""; else L=$W/lib_$v.so; fi
    RL4CO_AMD_LIB=$L python $R/tools/train_encoder_bench.py 2>&1 | grep "fused_stack=True" | head -1 | sed "s/^/$v: /"
  done
fi
