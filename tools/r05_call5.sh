#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo new; python tools/attn_bwd_err.py
echo old; RL4CO_AMD_LIB=tools/probes/_build/lib_attnold.so python tools/attn_bwd_err.py old
echo new; python tools/attn_bwd_err.py
