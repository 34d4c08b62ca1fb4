#!/bin/bash
# XCD tile blocking of the split GEMM at the bench's 248 clouds (UA_SPLIT_BR x UA_SPLIT_BC), one box, two interleaved repetitions
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; }
for rep in 1 2; do
  for cfg in "4 8" "2 12" "8 4" "2 8" "4 12" "1 32"; do set -- $cfg
    echo -n "BR=$1 BC=$2 rep=$rep: "; UA_SPLIT_BR=$1 UA_SPLIT_BC=$2 $B 2>/dev/null | val
  done
done
echo -n "row-major order: "; UA_SPLIT_TILE_ORDER=0 $B 2>/dev/null | val
