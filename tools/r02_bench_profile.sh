#!/bin/bash
# rocprofv3 kernel stats of the default bench (headline legs only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_r2b -o bench -- python3 $R/bench.py --no-cpu-baseline --no-similarity --no-both-modes > $R/gpurun_out/r02b_bench_under_rocprof_stdout.json 2> $R/gpurun_out/r02b_bench_under_rocprof.err
cp $R/gpurun_out/prof_bench_r2b/bench_kernel_stats.csv $R/gpurun_out/r02b_bench_kernel_stats.csv
echo done
