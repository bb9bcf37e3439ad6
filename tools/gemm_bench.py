#!/usr/bin/env python3
"""Micro-benchmark of the NT (forward / dX) and TN (dW) GEMMs in both arithmetic modes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = 'cuda:0'
shapes = [(131072, 1024, 2048), (131072, 1024, 256), (131072, 512, 512), (65536, 512, 512), (32768, 512, 512), (8192, 512, 512), (2048, 512, 1536), (2048, 1024, 1024), (34000, 1024, 300)]
if len(sys.argv) > 1 and sys.argv[1] == 'tn':
    MODES = ('bf16x3',)
    if len(sys.argv) > 2:       # tn M,N,K M,N,K ...
        shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]]
else:
    MODES = ('f32', 'bf16x3')
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dz = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev); dw = torch.zeros(N, K, device=dev)
    for mode in MODES:
        ops.set_matmul_mode(mode)
        t_nt = timeit(lambda: ops.gemm_grouped(x, K, None, w, None, out, N, None, M, 1, N, K))
        t_tn = timeit(lambda: ops.gemm_tn(dz, x, dw, M, N, K))
        fl = 2.0 * M * N * K / 1e9
        print('M=%6d N=%4d K=%4d %-7s NT %8.3f ms (%6.1f TF)   TN %8.3f ms (%6.1f TF)' % (M, N, K, mode, t_nt, fl / t_nt, t_tn, fl / t_tn))
ops.set_matmul_mode('bf16x3')
