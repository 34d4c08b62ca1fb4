#!/bin/bash
# Interleaved A/B on one box (headline leg only): VAR=1 vs VAR=0 for the environment knob named in $KNOB, three pairs of 24 timed steps
R=$GRAFT_REPO_ROOT
KNOB=${KNOB:-UA_FOLD_LN}
B="python $R/bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-similarity --no-both-modes"
run() { tag=$1; shift; "$@" > $R/gpurun_out/ab_$tag.log 2>&1; tail -1 $R/gpurun_out/ab_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], 'pc/s', d['ms_per_step'], 'ms/step', 'gemm ms/launch', d['roofline']['ms_per_launch_avg'])"; }
for i in 1 2 3; do
  env $KNOB=1 $B > $R/gpurun_out/ab_on_$i.log 2>&1; tail -1 $R/gpurun_out/ab_on_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KNOB=1', d['value'], 'pc/s', d['ms_per_step'], 'ms/step', 'gemm ms/launch', d['roofline']['ms_per_launch_avg'])"
  env $KNOB=0 $B > $R/gpurun_out/ab_off_$i.log 2>&1; tail -1 $R/gpurun_out/ab_off_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KNOB=0', d['value'], 'pc/s', d['ms_per_step'], 'ms/step', 'gemm ms/launch', d['roofline']['ms_per_launch_avg'])"
done
