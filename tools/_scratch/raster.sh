#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "4 8" "2 12" "2 16" "8 4" "1 32" "16 2" "3 11" "4 6"; do set -- $cfg; echo "== BR=$1 BC=$2"; UA_SPLIT_BR=$1 UA_SPLIT_BC=$2 python tools/bench_kernels.py gemm_split 2>/dev/null | grep -E "qkv Bc=32|fc1|conv4|square" ; done
