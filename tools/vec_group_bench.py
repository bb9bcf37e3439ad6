#!/usr/bin/env python3
"""Launch times of the grouped vector-level operator (stair_vec_group) on the shapes a training step gives it: one Exists-shaped
first layer (K = 1536), a level's worth of mixed problems, the decoder's two layers; HIP events over 50 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops                                          # noqa: E402

DEV = 'cuda:0'
H = 512
g = torch.Generator(device=DEV).manual_seed(0)


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def fwd(rows, pack, nseg, N=512, arena=None):
    arena = arena if arena is not None else torch.randn(4096, H, device=DEV, generator=g)
    ia = torch.randint(0, 4096, (rows,), device=DEV, generator=g, dtype=torch.int32)
    ib = torch.randint(0, 4096, (rows,), device=DEV, generator=g, dtype=torch.int32)
    W = torch.randn(N, nseg * H, device=DEV, generator=g) * 0.02
    return dict(kind='fwd', rows=rows, a=arena, b=arena, ia=ia, ib=ib, pack=pack, W=W, bias=torch.zeros(N, device=DEV), N=N, act='relu',
                out=torch.empty(rows, N, device=DEV))


for rows in (13, 66, 130, 1044):
    for pack, nseg in (('a', 1), ('cat2', 2), ('exists', 3)):
        p = fwd(rows, pack, nseg)
        print('fwd rows %5d %-7s K=%4d : %7.1f us' % (rows, pack, nseg * H, timed(lambda: ops.vec_group([p]))))
level = [fwd(40, 'exists', 3), fwd(66, 'cat2', 2), fwd(30, 'xor', 3), fwd(50, 'a', 1), fwd(20, 'a', 1)]
print('a level of five problems (20 .. 66 rows)  : %7.1f us' % timed(lambda: ops.vec_group(level)))
for n in (128, 2048):
    d0 = fwd(n, 'cat2', 2, N=1024)
    d3 = fwd(n, 'cat2', 2, N=172)
    print('decoder n = %4d: layer 1 %7.1f us, layer 2 %7.1f us' % (n, timed(lambda: ops.vec_group([d0])), timed(lambda: ops.vec_group([d3]))))
