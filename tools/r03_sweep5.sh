#!/bin/bash
# Chunk sizes whose GEMM tile counts ((4*chunk + 1) row tiles x {4, 12, 22} column tiles) land just under a multiple of 256 CUs: 63, 127 against 62, 124, 128
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; }
for rep in 1 2; do
  for c in 124 127 128 63 62; do echo -n "chunk=$c rep=$rep: "; $B --chunk $c 2>/dev/null | val; done
done
