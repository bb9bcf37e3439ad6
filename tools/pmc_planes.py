#!/usr/bin/env python3
"""Launch the dominant kernel of the bf16-feature configuration -- the video bi-LSTM input projection as a plane GEMM
(M = B*T, N = 1024, K = 2048; A = stored bf16 clip features, W = hi/lo planes) -- a few times on its own, so that
rocprofv3 --pmc passes stay cheap.  bench.py's roofline.traffic is read from the summary of these passes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops

B, T, V = 2048, 64, 2048
N = int(os.environ.get('STAIR_PLANES_N', '2048'))          # 2048 = 8 * Hh: both directions of the bi-LSTM in one launch; 256: ONE column
                                                            # tile per A panel, so A is read exactly once (the FETCH_SIZE calibration)
M = B * T
dev = 'cuda:0'
x = torch.randn(M, V, device=dev).to(torch.bfloat16)
w = torch.randn(N, V, device=dev) * 0.02
wh, wl = ops.split_planes_tiled(w)
b = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(6):
    ops.gemm_planes(x, None, wh, wl, b, out=out)
torch.cuda.synchronize()
print('shape M N K = %d %d %d' % (M, N, V))
print('algorithmic bytes per launch: A %d + W %d + C %d = %d' % (M * V * 2, N * V * 4, M * N * 4, M * V * 2 + N * V * 4 + M * N * 4))
