#!/usr/bin/env python3
"""Launch the fused tile operator on its own (Localize-shaped: 1024 instances, [64, 512] tiles, two layers + cosine tail;
inference form, then the training form that also writes both saved activations) so that rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE, one counter per pass) stay cheap:
    rocprofv3 --pmc FETCH_SIZE -d DIR/fetch -o run --output-format csv -- python3 tools/pmc_tile.py
    rocprofv3 --pmc WRITE_SIZE -d DIR/write -o run --output-format csv -- python3 tools/pmc_tile.py
    python3 tools/summarize_prof.py pmc DIR/fetch/run_counter_collection.csv DIR/write/run_counter_collection.csv profiles/rNN_pmc_tile.json"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import ops
from stair_amd._lib import TileMlpArgs, check, lib

dev = 'cuda:0'
H, T, cnt = 512, 64, 1024
g = torch.Generator(device=dev).manual_seed(0)
w1 = torch.randn(H, H, device=dev, generator=g) * 0.04; w2 = torch.randn(H, H, device=dev, generator=g) * 0.04
b1 = torch.zeros(H, device=dev)
p1, p2 = ops.pack_wfrag(w1), ops.pack_wfrag(w2)
x = torch.randn(cnt, T, H, device=dev, generator=g)
kb = torch.randn(cnt, H, device=dev, generator=g)
att = torch.empty(cnt, T, device=dev)
first = torch.arange(cnt, dtype=torch.int32, device=dev); one = torch.ones(cnt, dtype=torch.int32, device=dev)
sv1 = torch.empty(cnt, T, H, device=dev); sv2 = torch.empty(cnt, T, H, device=dev)
for train in (False, True):
    for _ in range(4):
        a = TileMlpArgs()
        a.X, a.x_gstride = x.data_ptr(), T * H
        a.W[0], a.bias[0], a.act[0] = p1.data_ptr(), b1.data_ptr(), 1
        a.W[1], a.bias[1], a.act[1] = p2.data_ptr(), b1.data_ptr(), 0
        a.n_layers = 2
        if train:
            a.save[0], a.save[1] = sv1.data_ptr(), sv2.data_ptr()
        a.tail = 3
        a.kb, a.pair_first, a.pair_cnt, a.att_idx, a.att = kb.data_ptr(), first.data_ptr(), one.data_ptr(), first.data_ptr(), att.data_ptr()
        a.cnt, a.T, a.H = cnt, T, H
        check(lib.stair_tile_mlp_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
tile = cnt * T * H * 4
print('instances %d; algorithmic bytes per launch: inference = tiles in %d + weight planes %d + keyword rows %d + scores out %d = %d; '
      'training adds two saved activations: %d' % (cnt, tile, 2 * H * H * 4, cnt * H * 4, cnt * T * 4,
                                                  tile + 2 * H * H * 4 + cnt * H * 4 + cnt * T * 4, 2 * tile))
