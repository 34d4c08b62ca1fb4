#!/bin/bash
# Emulated owner mode (one GPU standing in for rank 0 and for an encoder rank, wire time excluded): sweep of the rank-0 share parameter rho at chunk 124
for rho in 0.04 0.08 0.12 0.18 0.25; do
  python tools/emulate_scaling.py --mode owner --rho $rho --chunk 124 --steps 3 --world 2,4,8 2>/dev/null | grep "W="
done
