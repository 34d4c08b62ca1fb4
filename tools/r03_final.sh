#!/bin/bash
# round-3 evidence on one box: default bench line, the same under rocprofv3 (kernel stats), PMC passes on the round's code,
# smoke, and the plumbing rehearsal of `bench.py --gpus 2` (two ranks sharing ONE GPU over gloo; not a scaling number)
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r03_bench_final_stdout.json 2> gpurun_out/r03_bench_final.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-similarity --no-comparator --no-both-modes > $R/gpurun_out/r03_bench_prof_stdout.json 2> $R/gpurun_out/r03_bench_prof.err
f=$(find $R/gpurun_out/prof_r03 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/r03_bench_kernel_stats.csv
rm -rf $R/gpurun_out/prof_r03
TAG=r03 bash $R/tools/pmc_passes.sh > $R/gpurun_out/pmc_r03_run.log 2>&1
cd $R && python tools/pmc_collect.py gpurun_out/pmc_r03_? > gpurun_out/r03_pmc.json 2>> gpurun_out/pmc_r03_run.log
rm -rf gpurun_out/pmc_r03_?
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1
cd $R && UA_DIST_BACKEND=gloo HIP_VISIBLE_DEVICES=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-similarity --no-comparator --no-both-modes > gpurun_out/r03_bench_2ranks_one_gpu_gloo_stdout.json 2> gpurun_out/r03_bench_2ranks.err
echo done
