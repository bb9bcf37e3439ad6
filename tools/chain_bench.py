#!/usr/bin/env python3
"""Backward chains of the fused tile operator (stair_tile_mlp_fwd with act 3 / ACCUMULATE, and Temporal's chain: ln_bwd +
ROWSCALE_ADJ) alone, by instance count: microseconds per launch and per round of 256 tiles."""
import os, sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops
from stair_amd._lib import TileMlpArgs, lib, check

dev = 'cuda:0'
H, T = 512, 64
g = torch.Generator(device=dev).manual_seed(0)
w1 = torch.randn(H, H, device=dev, generator=g) * 0.04
w2 = torch.randn(H, H, device=dev, generator=g) * 0.04
p1t, p2t = ops.pack_wfrag(w1, transpose=True), ops.pack_wfrag(w2, transpose=True)
gamma = torch.ones(H, device=dev)


def ev_time(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


for cnt in (256, 1024):
    dy = torch.randn(cnt, T, H, device=dev, generator=g)
    a1 = torch.randn(cnt, T, H, device=dev, generator=g)
    a2 = torch.randn(cnt, T, H, device=dev, generator=g)
    feat = torch.randn(cnt, T, H, device=dev, generator=g)
    rs = torch.rand(cnt, T, device=dev, generator=g)
    dfeat = torch.zeros(cnt, T, H, device=dev)
    drs = torch.zeros(cnt, T, device=dev)
    dz1 = torch.empty(cnt, T, H, device=dev); dz2 = torch.empty(cnt, T, H, device=dev)
    dgam = torch.zeros(H, device=dev); dbet = torch.zeros(H, device=dev)

    def chain(nl, excl, masked_in):
        a = TileMlpArgs()
        a.X, a.x_gstride = dy.data_ptr(), T * H
        if masked_in:
            a.in_mask, a.in_mask_gstride, a.in_scale, a.save_in = a2.data_ptr(), T * H, 1.0, dz2.data_ptr()
        a.n_layers = nl
        a.act_scale = 1.0
        if nl == 2:
            a.W[0], a.act[0] = p2t.data_ptr(), 3
            a.act_mask[0] = a1.data_ptr(); a.save[0] = dz1.data_ptr()
            a.W[1], a.act[1] = p1t.data_ptr(), 0
        else:
            a.W[0], a.act[0] = p1t.data_ptr(), 0
        a.tail = 6
        a.out, a.out_gstride = dfeat.data_ptr(), T * H
        a.acc_exclusive = excl
        a.cnt, a.T, a.H = cnt, T, H
        check(lib.stair_tile_mlp_fwd(C.byref(a), stream()))

    def temporal(excl):
        a = TileMlpArgs()
        a.X, a.x_gstride = dy.data_ptr(), T * H
        a.ln_bwd, a.in_mask, a.in_mask_gstride, a.in_scale, a.save_in = 1, a1.data_ptr(), T * H, 1.0, dz1.data_ptr()
        a.gamma, a.dgamma, a.dbeta, a.ln_eps = gamma.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), 1e-5
        a.W[0], a.act[0], a.n_layers = p1t.data_ptr(), 0, 1
        a.tail = 8
        a.out, a.out_gstride = dfeat.data_ptr(), T * H
        a.adj_feat, a.adj_feat_gstride = feat.data_ptr(), T * H
        a.adj_rs, a.adj_drs = rs.data_ptr(), drs.data_ptr()
        a.acc_exclusive = excl
        a.cnt, a.T, a.H = cnt, T, H
        check(lib.stair_tile_mlp_fwd(C.byref(a), stream()))

    rounds = cnt / 256
    for name, fn in (('2-layer chain, atomics', lambda: chain(2, 0, True)), ('2-layer chain, read-add-write', lambda: chain(2, 1, True)),
                     ('1-layer chain (HasItem), atomics', lambda: chain(1, 0, True)), ('1-layer chain, read-add-write', lambda: chain(1, 1, True)),
                     ('1-layer chain, no input mask', lambda: chain(1, 0, False)),
                     ('Temporal chain, atomics', lambda: temporal(0)), ('Temporal chain, read-add-write', lambda: temporal(1))):
        t = ev_time(fn)
        print('cnt %5d  %-36s %7.1f us  = %5.1f us per round of 256 tiles' % (cnt, name, t, t / rounds), flush=True)
