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

# BPTT (recurrence + hprev + the four weight-gradient GEMMs), cooperative vs one-workgroup reverse recurrence
out, h_n, gates, cbuf = ops.lstm_bidir(x, off, a.T, ws, save=True)
d_out = torch.randn(a.n * a.T, 2 * a.Hh, device=dev, generator=g)
saved = gates.clone()
for coop in (True, False):
    for it in range(2 + a.iters):
        if it == 2:
            torch.cuda.synchronize(); tot = 0.0
        gates.copy_(saved)
        e0.record()
        ops.lstm_bidir_bwd(x, off, a.T, ws, out, gates, cbuf, d_out, None, coop=coop)
        e1.record(); torch.cuda.synchronize()
        if it >= 2:
            tot += e0.elapsed_time(e1)
    print('lstm_bidir_bwd coop=%d: %.3f ms per call (BPTT + dW GEMMs)' % (coop, tot / a.iters))

if os.environ.get('STAIR_LSTM_COOP_PROF') == '1':
    # diagnostic build: re-run once with our own scratch so the phase sums (cycles of wave 0, workgroup 0) can be read back
    from stair_amd._lib import LstmArgs, lib, check
    import ctypes as C
    rows, I = x.shape
    out = torch.empty(rows, 2 * a.Hh, device=dev); h_n = torch.empty(a.n, 2 * a.Hh, device=dev)
    xproj = torch.empty(rows, 8 * a.Hh, device=dev); bias_ws = torch.empty(8 * a.Hh, device=dev); pack = torch.empty(8 * a.Hh * a.Hh, device=dev)
    nb = int(lib.stair_lstm_coop_ws_bytes(a.n)); coop = torch.zeros(nb, device=dev, dtype=torch.uint8)
    A = LstmArgs()
    A.x, A.ldx, A.rows, A.n, A.max_len, A.I, A.Hh = x.data_ptr(), I, rows, a.n, a.T, I, a.Hh
    A.seq_off = off.data_ptr()
    for d in range(2):
        A.w_ih[d], A.w_hh[d], A.b_ih[d], A.b_hh[d] = (ws[4 * d + i].data_ptr() for i in range(4))
    A.xproj_ws, A.bias_ws, A.whh_pack_ws = xproj.data_ptr(), bias_ws.data_ptr(), pack.data_ptr()
    A.out, A.ldo, A.h_n = out.data_ptr(), 2 * a.Hh, h_n.data_ptr()
    A.coop_ws, A.coop_ws_bytes = coop.data_ptr(), nb
    check(lib.stair_lstm_bidir_fwd(C.byref(A), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    import numpy as np
    raw = coop.cpu().numpy()
    # the flag block follows the slabs; err word + 8 words in, 8 uint64 sums.  Geometry mirrors coop_geometry().
    nt = int(os.environ.get('STAIR_LSTM_COOP_TILES', '0')) or (3 if a.n > 2048 else (2 if a.n > 32 else 1))
    gpd = max(1, min((a.n + 32 * nt - 1) // (32 * nt), 32)); G = 2 * gpd
    base = 2 * G * nt * 32768 + (G * nt * 4 * 16) * 4
    sums = raw[base + 32: base + 32 + 64].view(np.uint64)
    names = ['xproj issue + MFMA chain', 'vmcnt(0)', 'mid barrier', 'flag/peek/cell/out stores', 'publish', 'slab wait + ds_write', 'land barrier', 'poll + slab issue']
    tot = float(sums.sum())
    print('phase cycles of wave 0 / workgroup 0 (err word %d):' % int(raw[base: base + 4].view(np.uint32)[0]))
    for nme, v in zip(names, sums):
        print('  %-28s %12d  %5.1f %%' % (nme, int(v), 100.0 * float(v) / max(tot, 1.0)))
