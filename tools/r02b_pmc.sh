#!/bin/bash
# PMC passes on the last round-2 code (folded-LayerNorm GEMM epilogues, XCD-aware attention order)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=r2b bash $R/tools/pmc_passes.sh > $R/gpurun_out/pmc_r2b_run.log 2>&1
cd $R && python tools/pmc_collect.py gpurun_out/pmc_r2b_? > gpurun_out/r02b_pmc.json 2>> gpurun_out/pmc_r2b_run.log
echo done
