#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_replay_valu -o replay -- python3 $R/tools/bench_kernels.py replay > $R/gpurun_out/prof_replay_valu.log 2>&1
cp $R/gpurun_out/prof_replay_valu/replay_kernel_stats.csv $R/gpurun_out/r02_replay_kernel_stats_valu.csv
export UA_ALIGN_PATH=mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_replay_mfma -o replay -- python3 $R/tools/bench_kernels.py replay > $R/gpurun_out/prof_replay_mfma.log 2>&1
cp $R/gpurun_out/prof_replay_mfma/replay_kernel_stats.csv $R/gpurun_out/r02_replay_kernel_stats_mfma.csv
grep "cache replay" $R/gpurun_out/prof_replay_valu.log $R/gpurun_out/prof_replay_mfma.log
cd $R && python -m pytest tests/test_gpu_harness.py -m gpu -q -s -k "long_stream_against" 2>&1 | grep -E "reference run|HIP" > gpurun_out/r02_g12_spread.txt; cat gpurun_out/r02_g12_spread.txt | cut -c1-200
