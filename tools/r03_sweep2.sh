#!/bin/bash
# Same-box A/B of the alignment forward's queries per workgroup (UA_ALIGN_FWD_Q = 4 | 10 | 20): replay latency alone and the bench at chunk 124 / 31
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; }
for q in 4 10 20; do echo -n "Q=$q "; UA_ALIGN_FWD_Q=$q python tools/bench_kernels.py replay 2>/dev/null | grep replay; done
for rep in 1 2; do
  for q in 4 10 20; do
    for c in 124 31; do echo -n "Q=$q chunk=$c rep=$rep: "; UA_ALIGN_FWD_Q=$q $B --chunk $c 2>/dev/null | val; done
  done
done
