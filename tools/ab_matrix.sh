#!/bin/bash
# Interleaved A/B on one box (headline leg only): UA_FOLD_LN=1 (default) vs 0, three pairs of 24 timed steps
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-similarity --no-both-modes"
run() { tag=$1; shift; "$@" > $R/gpurun_out/ab_$tag.log 2>&1; tail -1 $R/gpurun_out/ab_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], 'pc/s', d['ms_per_step'], 'ms/step')"; }
for i in 1 2 3; do
  run fold_$i $B
  UA_FOLD_LN=0 run nofold_$i $B
done
