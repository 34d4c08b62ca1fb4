#!/bin/bash
# round-2 final evidence on one box: GPU test-suite, default bench line, the same under rocprofv3 (kernel stats), smoke
R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests.log 2>&1; tail -2 gpurun_out/r02_gpu_tests.log
cd $R && python bench.py > gpurun_out/r02_bench_final_stdout.json 2> gpurun_out/r02_bench_final.err
TAG=r02 bash tools/r02_bench_profile.sh
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.log 2>&1
echo done
