#!/bin/bash
# round-2 final evidence on one box: default bench line, the same under rocprofv3 (kernel stats), smoke, and the
# plumbing rehearsal of `bench.py --gpus 2` (two ranks sharing ONE GPU over gloo; not a scaling number)
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r02_bench_final_stdout.json 2> gpurun_out/r02_bench_final.err
TAG=r02 bash tools/r02_bench_profile.sh
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.log 2>&1
cd $R && UA_DIST_BACKEND=gloo HIP_VISIBLE_DEVICES=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-similarity --no-both-modes > gpurun_out/r02_bench_2ranks_one_gpu_gloo_stdout.json 2> gpurun_out/r02_bench_2ranks.err
echo done
