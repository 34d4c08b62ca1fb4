#!/bin/bash
# bench.py over chunk sizes on one box (headline leg only)
R=$GRAFT_REPO_ROOT
echo "python bench.py --chunk C --no-cpu-baseline --no-similarity --no-both-modes --steps 12 --warmup 3   (MI355X, split GEMM mode, last round-2 code)"
echo "chunk  point-clouds/s  ms/step  executed fraction of the f16 MFMA peak (split GEMM)"
for c in 1 8 16 31 32 62 124; do
  python $R/bench.py --chunk $c --steps 12 --warmup 3 --no-cpu-baseline --no-similarity --no-both-modes > $R/gpurun_out/chunk_$c.log 2>&1
  tail -1 $R/gpurun_out/chunk_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6d %8.1f %12.2f %8.3f' % ($c, d['value'], d['ms_per_step'], d['roofline']['executed_frac']))"
done
