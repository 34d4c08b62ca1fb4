#!/bin/bash
# round-2 profile collection on the GPU box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=r2 bash $R/tools/pmc_passes.sh > $R/gpurun_out/pmc_r2_run.log 2>&1
UA_PMC_PART=sim TAG=r2s bash $R/tools/pmc_passes.sh > $R/gpurun_out/pmc_r2s_run.log 2>&1
cd $R && python tools/pmc_collect.py gpurun_out/pmc_r2_? > gpurun_out/r02_pmc_main.json 2>> gpurun_out/pmc_r2_run.log
python tools/pmc_collect.py gpurun_out/pmc_r2s_? > gpurun_out/r02_pmc_sim.json 2>> gpurun_out/pmc_r2s_run.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_r2 -o bench -- python3 $R/bench.py --no-cpu-baseline --no-similarity --no-both-modes > $R/gpurun_out/r02_bench_under_rocprof_stdout.json 2> $R/gpurun_out/r02_bench_under_rocprof.err
cp $R/gpurun_out/prof_bench_r2/bench_kernel_stats.csv $R/gpurun_out/r02_bench_kernel_stats.csv
echo done
