#!/bin/bash
# Clouds per mini-PointNet slice (UA_PE_CLOUDS): the per-point activations of a slice (h3: 64 MB per cloud in split form) fit the 256 MB Infinity Cache only for small slices
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; }
for rep in 1 2; do
  for c in 8 2 4 1 16; do echo -n "pe_clouds=$c rep=$rep: "; UA_PE_CLOUDS=$c $B 2>/dev/null | val; done
done
