#!/bin/bash
# Same-box sweeps for round 3: chunk size, and the lean (three-launch) residual step against the one-launch step (interleaved).
# usage: bash tools/r03_sweep.sh > gpurun_out/r03_sweep.txt
B="python bench.py --no-cpu-baseline --no-similarity --no-comparator --no-both-modes --steps 8 --warmup 2"
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; }
for rep in 1 2; do
  for lean in 0 1; do
    for c in 31 124; do
      echo -n "lean=$lean chunk=$c rep=$rep: "; UA_REPLAY_LEAN=$lean $B --chunk $c 2>/dev/null | val
    done
  done
done
for c in 62 248; do echo -n "lean=default chunk=$c: "; $B --chunk $c 2>/dev/null | val; done
echo -n "no residual learning chunk=124: "; $B --chunk 124 --no-res-learning 2>/dev/null | val
echo -n "no overlap chunk=124: "; $B --chunk 124 --no-overlap 2>/dev/null | val
