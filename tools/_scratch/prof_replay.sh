#!/bin/bash
# rocprofv3 kernel stats of the replay micro-benchmark (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_replay -o replay -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py replay > $GRAFT_REPO_ROOT/gpurun_out/prof_replay.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_replay -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/r02_replay_kernel_stats.csv
