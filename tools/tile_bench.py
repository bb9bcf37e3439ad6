#!/usr/bin/env python3
"""Fused per-clip tile operator (stair_tile_mlp_fwd) against the launch sequence it replaces, by instance count.
Localize-shaped: Lin . ReLU . Lin on [cnt, 64, 512] tiles + cosine against one keyword row per instance."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops

dev = 'cuda:0'
H, T = 512, 64
g = torch.Generator(device=dev).manual_seed(0)
w1 = torch.randn(H, H, device=dev, generator=g) * 0.04; b1 = torch.zeros(H, device=dev)
w2 = torch.randn(H, H, device=dev, generator=g) * 0.04; b2 = torch.zeros(H, device=dev)
p1, p2 = ops.pack_wfrag(w1), ops.pack_wfrag(w2)
from stair_amd._lib import TileMlpArgs, lib, check
import ctypes as C


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


for cnt in (64, 256, 273, 512, 1024, 1056):
    x = torch.randn(cnt, T, H, device=dev, generator=g)
    kb = torch.randn(cnt, H, device=dev, generator=g)
    att = torch.empty(cnt, T, device=dev)
    first = torch.arange(cnt, dtype=torch.int32, device=dev); one = torch.ones(cnt, dtype=torch.int32, device=dev)
    sv1 = torch.empty(cnt, T, H, device=dev); sv2 = torch.empty(cnt, T, H, device=dev)

    def fused(train):
        a = TileMlpArgs()
        a.X, a.x_gstride = x.data_ptr(), T * H
        a.W[0], a.bias[0], a.act[0] = p1.data_ptr(), b1.data_ptr(), 1
        a.W[1], a.bias[1], a.act[1] = p2.data_ptr(), b2.data_ptr(), 0
        a.n_layers = 2
        if train:
            a.save[0], a.save[1] = sv1.data_ptr(), sv2.data_ptr()
        a.tail = 3
        a.kb, a.pair_first, a.pair_cnt, a.att_idx, a.att = kb.data_ptr(), first.data_ptr(), one.data_ptr(), first.data_ptr(), att.data_ptr()
        a.cnt, a.T, a.H = cnt, T, H
        check(lib.stair_tile_mlp_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def sequence():
        ops.gemm_grouped(x.view(-1, H), H, None, w1, b1, sv1.view(-1, H), H, None, cnt * T, 1, H, H, lda=H, ldc=H, act='relu')
        ops.gemm_grouped(sv1.view(-1, H), H, None, w2, b2, sv2.view(-1, H), H, None, cnt * T, 1, H, H, lda=H, ldc=H)
        ops.cosine_attn(sv2, first, kb, first, cnt, T, H)
    t_inf, t_tr = ev_time(lambda: fused(False)), ev_time(lambda: fused(True))
    try:
        t_seq = ev_time(sequence)
    except Exception as e:
        t_seq = float('nan')
    fl = 2.0 * cnt * T * H * H * 2
    print('cnt %5d: fused inference %7.1f us (%6.1f TFLOP/s), fused + saves %7.1f us, GEMM-GEMM-cosine sequence %7.1f us' % (
        cnt, t_inf, fl / t_inf / 1e6, t_tr, t_seq), flush=True)

# per-layer slope and fixed cost: 1, 2, 3 layers, tail NONE (tail = 0) and STORE
p3 = ops.pack_wfrag(w2)
out = torch.empty(1056, T, H, device=dev)
for cnt in (64, 256):
    x = torch.randn(cnt, T, H, device=dev, generator=g)
    for tail in (0, 1):
        ts = []
        for nl in (1, 2, 3):
            def run():
                a = TileMlpArgs()
                a.X, a.x_gstride = x.data_ptr(), T * H
                for l, pl_ in enumerate((p1, p2, p3)[:nl]):
                    a.W[l], a.bias[l], a.act[l] = pl_.data_ptr(), b1.data_ptr(), 1
                a.n_layers = nl
                a.tail = tail
                a.out, a.out_gstride = out.data_ptr(), T * H
                a.cnt, a.T, a.H = cnt, T, H
                check(lib.stair_tile_mlp_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ts.append(ev_time(run))
        print('cnt %4d tail %d: 1 / 2 / 3 layers %6.1f %6.1f %6.1f us -> per layer %5.1f us, fixed %5.1f us' % (
            cnt, tail, ts[0], ts[1], ts[2], (ts[2] - ts[0]) / 2, ts[0] - (ts[2] - ts[0]) / 2), flush=True)
