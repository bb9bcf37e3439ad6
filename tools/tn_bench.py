#!/usr/bin/env python3
"""Weight-gradient products of the module level at the shapes of a 2048-question step (instances x 64 rows, N = K = 512): the
atomic register-transposing kernel (stair_gemm_tn_f32) against the slab-reduced transposed-read kernel (stair_gemm_tn_slabs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops
dev = 'cuda:0'
N = K = 512; R = 64
for inst in (13, 87, 273, 531, 754, 1414, 1704):
    M = inst * R
    g = torch.Generator(device=dev).manual_seed(0)
    dZ = torch.randn(M, N, device=dev, generator=g); X = torch.randn(inst + 5, R, K, device=dev, generator=g)
    idx = torch.randperm(inst + 5, device=dev, generator=g)[:inst].to(torch.int32)
    Cm = torch.zeros(N, K, device=dev); b = torch.zeros(N, device=dev)
    res = []
    for det in (False, True):
        for _ in range(3):
            ops.gemm_tn(dZ, X, Cm, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=idx, colsum=b, deterministic=det)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_tn(dZ, X, Cm, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=idx, colsum=b, deterministic=det)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * N * K
    print('M = %6d (%4d instances): atomic %.1f us = %.0f TFLOP/s, slabs (+ reduction, + scratch alloc) %.1f us = %.0f TFLOP/s' %
          (M, inst, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6), flush=True)
