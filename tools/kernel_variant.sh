#!/bin/bash
# Build a variant of ONE translation unit of the library with extra compile definitions, linked against the product's other
# objects:   tools/kernel_variant.sh am_train_ops.hip wls144 "-DRL4CO_WLS=144"   ->  tools/probes/_build/lib_wls144.so
# (*_f16.hip wrappers of the same source are rebuilt with the same definitions). Run on the GPU box with RL4CO_AMD_LIB=<that file>.
set -e
R=$(cd $(dirname $0)/.. && pwd)
W=$R/tools/probes/_build
mkdir -p $W
SRC=$1; NAME=$2; DEFS=$3
python -c "import sys; sys.path.insert(0,'$R'); from rl4co_amd import build; build.build_library()"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/rl4co_amd/csrc"
STEM=${SRC%.hip}
OBJS=""
for o in $R/rl4co_amd/lib/obj/*.o; do
  case $(basename $o) in
    $STEM.hip.o|${STEM}_f16.hip.o) ;;
    *) OBJS="$OBJS $o" ;;
  esac
done
hipcc $FLAGS $DEFS -c $R/rl4co_amd/csrc/$SRC -o $W/v_$NAME.o &
if [ -f $R/rl4co_amd/csrc/${STEM}_f16.hip ]; then hipcc $FLAGS $DEFS -c $R/rl4co_amd/csrc/${STEM}_f16.hip -o $W/v_${NAME}_f16.o & fi
wait
VO="$W/v_$NAME.o"; [ -f $W/v_${NAME}_f16.o ] && VO="$VO $W/v_${NAME}_f16.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o $W/lib_$NAME.so $VO $OBJS
rm -f $W/v_$NAME.o $W/v_${NAME}_f16.o
echo built $W/lib_$NAME.so
