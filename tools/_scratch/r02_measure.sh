#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_harness.py -m gpu -q -x -k "align or residual or long_stream" > gpurun_out/r02_t9.log 2>&1; tail -2 gpurun_out/r02_t9.log
python tools/bench_kernels.py replay 2>/dev/null | grep replay
for c in 1 8 16 31 32; do python bench.py --chunk $c --no-cpu-baseline --no-similarity --no-both-modes --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', d['config']['samples_per_rank_per_step'], d['value'], 'pc/s', d['ms_per_step'], 'ms/step', 'gemm frac', d['roofline']['executed_frac'])"; done
HSA_ENABLE_IPC_MODE_LEGACY=0 UA_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --no-similarity --no-both-modes > gpurun_out/r02_bench2_owner.json 2> gpurun_out/r02_bench2_owner.err; grep -E "ratio|timed" gpurun_out/r02_bench2_owner.err; cut -c1-200 gpurun_out/r02_bench2_owner.json
