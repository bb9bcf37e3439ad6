#!/usr/bin/env python3
"""Fused vs launch-per-layer runner under the same dropout seed: per-parameter gradient differences."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stair_amd import spec, synth
from stair_amd.module_net import VideoNMN
from stair_amd._lib import lib
DEV = 'cuda:0'
config = dict(spec.DEFAULT_CONFIG)
NQ = int(os.environ.get('NQ', '24'))
qs = [synth.make_question(config, 4, i, form=synth.ALL_FORMS[i % len(synth.ALL_FORMS)], T=(64 if os.environ.get('RAGGED') != '1' else [64, 17, 40, 33, 8, 51][i % 6])) for i in range(NQ)]
out = {}
for mode in (1, 0):
    lib.stair_set_tile_mlp(mode)
    m = VideoNMN(config); w = synth.make_weights(config, 3)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)}); m = m.to(DEV)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    drop = (0.25, int(os.environ.get("DSEED", "7"))) if len(sys.argv) < 2 or sys.argv[1] != 'nodrop' else None
    res = m.forward_batch(qs, train=True, dropout=drop)
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    res.backward(answers, 1.0 / len(qs))
    out[mode] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
lib.stair_set_tile_mlp(-1)
gmax = max(float(g.abs().max()) for g in out[0].values())
worst = 0.0
for n, g in out[0].items():
    worst = max(worst, float((out[1][n] - g).norm()) / max(float(g.norm()), 1e-3 * gmax))
    d = float((out[1][n] - g).abs().max()); mx = float(g.abs().max())
    flag = '  <<<' if d > 4e-4 * max(mx, 1e-3 * gmax) else ''
    if flag or d > 1e-5 * max(mx, 1e-3 * gmax):
        print('%-55s max|g| %.3e  max diff %.3e  rel %.2e%s' % (n, mx, d, d / max(mx, 1e-12), flag))

print('worst relative L2 error over all parameters: %.3e' % worst)
