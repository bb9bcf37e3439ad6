#!/usr/bin/env python3
"""Micro-benchmark of the plane GEMM (csrc/gemm_planes.hip) against the register-staged split kernel on the dominant
shape (LSTM input projection, M = B*T, N = 1024, K = 2048) and the K = 512 module shapes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = 'cuda:0'
shapes = [(131072, 1024, 2048), (131072, 512, 512), (32768, 512, 512), (34048, 1024, 320)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
print('STAIR_PLANES_ABLATE =', os.environ.get('STAIR_PLANES_ABLATE', '0'))
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    out = torch.empty(M, N, device=dev)
    xh, xl = ops.split_planes(x); wh, wl = ops.split_planes(w); th, tl = ops.split_planes_tiled(w)
    fl = 2.0 * M * N * K / 1e9
    t0 = timeit(lambda: ops.gemm_grouped(x, K, None, w, None, out, N, None, M, 1, N, K))
    t1 = timeit(lambda: ops.gemm_planes(xh, None, wh, wl, out=out))
    t2 = timeit(lambda: ops.gemm_planes(xh, xl, wh, wl, out=out))
    t3 = timeit(lambda: ops.split_planes(x))
    t4 = timeit(lambda: ops.gemm_planes(xh, None, th, tl, out=out))
    t5 = timeit(lambda: ops.gemm_planes(xh, xl, th, tl, out=out))
    print('M=%6d N=%4d K=%4d  fp32-staged x3 %7.3f ms (%6.1f TF) | planes, W row-major: 2 products %7.3f ms (%6.1f TF)  3 products %7.3f ms (%6.1f TF) | W tiled: 2 products %7.3f ms (%6.1f TF)  3 products %7.3f ms (%6.1f TF) | split pass %7.3f ms'
          % (M, N, K, t0, fl / t0, t1, fl / t1, t2, fl / t2, t4, fl / t4, t5, fl / t5, t3), flush=True)
