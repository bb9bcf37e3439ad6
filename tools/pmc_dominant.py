#!/usr/bin/env python3
"""Launch the dominant kernel (video bi-LSTM input projection, M=B*T N=1024 K=2048) a few times on its own so that
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter set per pass) stay cheap.  bench.py's roofline.traffic
is filled from the summary of these passes (profiles/*_pmc_dominant.json)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops

B, T, V, N = 2048, 64, 2048, 1024
M = B * T
dev = 'cuda:0'
x = torch.randn(M, V, device=dev)
w = torch.randn(N, V, device=dev) * 0.02
b = torch.zeros(N, device=dev)
out = torch.empty(M, 2 * N, device=dev)
for _ in range(6):
    ops.gemm_grouped(x, V, None, w, b, out, 2 * N, None, M, 1, N, V, lda=V, ldc=2 * N)
torch.cuda.synchronize()
print('algorithmic bytes per launch: A %d + W %d + C %d = %d' % (M * V * 4, N * V * 4, M * N * 4, (M * V + N * V + M * N) * 4))
