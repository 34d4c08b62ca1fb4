#!/bin/bash
# Interleaved A/B on one box (headline leg only): $KNOB=$ON (default 1) vs $KNOB=$OFF (default 0), three pairs of 24 timed steps
R=$GRAFT_REPO_ROOT
KNOB=${KNOB:-UA_FOLD_LN}; ON=${ON:-1}; OFF=${OFF:-0}
B="python $R/bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-similarity --no-both-modes"
show() { tail -1 $2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'pc/s', d['ms_per_step'], 'ms/step', 'gemm ms/launch', d['roofline']['ms_per_launch_avg'])"; }
for i in 1 2 3; do
  env $KNOB=$ON $B > $R/gpurun_out/ab_on_$i.log 2>&1; show "$KNOB=$ON" $R/gpurun_out/ab_on_$i.log
  env $KNOB=$OFF $B > $R/gpurun_out/ab_off_$i.log 2>&1; show "$KNOB=$OFF" $R/gpurun_out/ab_off_$i.log
done
