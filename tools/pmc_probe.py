#!/usr/bin/env python3
"""Tiny driver for PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): launches the similarity
kernel (ua_gmm_loglik) and the fused cache step at BASELINE.json configs[4] size a few times, and
one GEMM of the dominant shape, so per-kernel HBM traffic can be read from the counter CSV.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_probe.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uni_adapter_amd  # noqa: E402,F401
from uni_adapter_amd import lib  # noqa: E402
from uni_adapter_amd.lib import GemmEpilogue  # noqa: E402

PART = os.environ.get('UA_PMC_PART', 'main')
if PART == 'sim':
    # batched-query MFMA similarity at BASELINE configs[4] size: operand build, 512 queries in split and plain-f16 form
    from uni_adapter_amd.dota_mixture import DOTA_mix
    from uni_adapter_amd import synthetic
    K, M, D, B = 4096, 16, 1024, 512
    text = synthetic.text_bank(K, D, 1)
    mdl = DOTA_mix({'epsilon': 1e-4, 'sigma': 1e-4}, D, K, text.t(), num_modes=M)
    xq = torch.nn.functional.normalize(torch.randn(B, D, device='cuda'), dim=1)
    for _ in range(3):
        mdl.mfma_prepare(B)
    for prec in ('split', 'f16'):
        for _ in range(3):
            mdl.predict_mfma(xq, precision=prec, reuse_operand=True)
    torch.cuda.synchronize()
    print('pmc probe (sim) done: operand bytes read+written', 16 * K * M * D, 'GEMM flops (algorithmic)', 4 * B * K * M * D)
    raise SystemExit
K, M, D = 4096, 16, 1024
mu = torch.randn(K, M, D, device='cuda') * 0.03
var = 1e-4 * (1 + torch.rand(K, M, D, device='cuda'))
pi = torch.full((K, M), 1.0 / M, device='cuda'); c = pi.clone(); cc = torch.zeros(K, device='cuda')
x = torch.nn.functional.normalize(torch.randn(1, D, device='cuda'), dim=1)
g = torch.softmax(torch.randn(1, K, device='cuda'), 1); sc = torch.empty(1, K, device='cuda')
st = lib.stream_ptr()
for _ in range(3):
    lib.call('ua_gmm_loglik', mu.data_ptr(), var.data_ptr(), pi.data_ptr(), x.data_ptr(), 1, K, M, D, 1e-4, sc.data_ptr(), None, None, st)
for _ in range(3):
    lib.call('ua_gmm_step_fused', mu.data_ptr(), var.data_ptr(), pi.data_ptr(), c.data_ptr(), cc.data_ptr(), x.data_ptr(), 1, x.data_ptr(),
             x.data_ptr(), g.data_ptr(), sc.data_ptr(), K, M, D, 1e-4, st)
Mr, N, Kk = 16416, 3072, 1024
A = torch.randn(Mr, Kk, device='cuda'); W = torch.randn(N, Kk, device='cuda') / 32; Cc = torch.empty(Mr, N, device='cuda')
e = GemmEpilogue()
for _ in range(3):
    lib.call('ua_gemm', A.data_ptr(), Kk, W.data_ptr(), Kk, Cc.data_ptr(), N, Mr, N, Kk, C.byref(e), st)
# split GEMM of the same shape + split attention
As = torch.empty(Mr, Kk, device='cuda'); Ws = torch.empty(N, Kk, device='cuda')
lib.call('ua_split_f16', A.data_ptr(), Kk, As.data_ptr(), Kk, Mr, Kk, Kk, 1.0, st)
lib.call('ua_split_f16', W.data_ptr(), Kk, Ws.data_ptr(), Kk, N, Kk, Kk, 1024.0, st)
for _ in range(3):
    lib.call('ua_gemm_split', As.data_ptr(), Kk, 1.0, Ws.data_ptr(), Kk, 1024.0, Cc.data_ptr(), N, Mr, N, Kk, C.byref(e), st)
qs = torch.empty(Mr, N, device='cuda'); lib.call('ua_split_f16', Cc.data_ptr(), N, qs.data_ptr(), N, Mr, N, N, 1.0, st)
ao = torch.empty(Mr, 1024, device='cuda')
for _ in range(3):
    lib.call('ua_attention_split', qs.data_ptr(), N, 1.0, None, 0, ao.data_ptr(), 1024, 1.0, 32, 512, 16, 64, st)
torch.cuda.synchronize()
print('pmc probe done: similarity algorithmic bytes', 2 * K * M * D * 4, 'fused', 4 * K * M * D * 4, 'gemm A+W+C bytes', 4 * (Mr * Kk + N * Kk + Mr * N))
