#!/bin/bash
# Same-box A/B of the last block for the CLS rows only (UA_LAST_BLOCK_CLS_ONLY = 1 default | 0): bench at chunk 124 and 31, two interleaved repetitions
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'], d['roofline']['launches'])"; }
for rep in 1 2; do
  for v in 1 0; do
    for c in 124 31; do echo -n "cls_only_last=$v chunk=$c rep=$rep: "; UA_LAST_BLOCK_CLS_ONLY=$v $B --chunk $c 2>/dev/null | val; done
  done
done
