#!/bin/bash
# rocprofv3 kernel stats of the default bench (headline legs only); TAG names the output, extra env via the caller
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02b}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_$TAG -o bench -- python3 $R/bench.py --no-cpu-baseline --no-similarity --no-both-modes > $R/gpurun_out/${TAG}_bench_under_rocprof_stdout.json 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err
cp $R/gpurun_out/prof_bench_$TAG/bench_kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
echo done $TAG
