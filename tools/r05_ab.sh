#!/bin/bash
# GPU box: A/B of two builds of the library on one box:  tools/r05_ab.sh <lib A> <lib B> -- <command ...>   (each run twice, alternating)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
A=$1; B=$2; shift 3
for i in 1 2; do
  for l in $A $B; do echo "== $l"; RL4CO_AMD_LIB=$R/$l "$@" 2>&1 | grep -v amdgpu.ids | tail -6; done
done
