#!/usr/bin/env python3
"""Micro-benchmark of stair_lstm_bidir_fwd at the video-encoder shape (n sequences x T x V -> Hh)."""
import argparse, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=2048)
ap.add_argument('--T', type=int, default=64)
ap.add_argument('--V', type=int, default=2048)
ap.add_argument('--Hh', type=int, default=256)
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(a.n * a.T, a.V, device=dev, generator=g)
off = (torch.arange(a.n + 1, device=dev) * a.T).to(torch.int32)
b = 1.0 / a.Hh ** 0.5
ws = []
for d in range(2):
    ws += [torch.empty(4 * a.Hh, a.V, device=dev).uniform_(-b, b, generator=g), torch.empty(4 * a.Hh, a.Hh, device=dev).uniform_(-b, b, generator=g),
           torch.empty(4 * a.Hh, device=dev).uniform_(-b, b, generator=g), torch.empty(4 * a.Hh, device=dev).uniform_(-b, b, generator=g)]
for _ in range(2):
    ops.lstm_bidir(x, off, a.T, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.lstm_bidir(x, off, a.T, ws)
e1.record(); torch.cuda.synchronize()
print('lstm_bidir n=%d T=%d V=%d Hh=%d: %.3f ms per call (input proj + recurrence)' % (a.n, a.T, a.V, a.Hh, e0.elapsed_time(e1) / a.iters))
