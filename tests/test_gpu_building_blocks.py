"""The backward halves of the exported building-block families (include/stair_hip.h: stair_cosine_attn_bwd,
stair_temporal_relate_bwd) against torch autograd of the oracle's restatement of the same operators
(oracle/nmn_oracle.py: cos_rows / op_existsframe for modules.py:162-217, temporal_relate for modules.py:255-277,317-323)."""
import numpy as np
import pytest
import torch

from helpers import oracle_weights
from oracle import nmn_oracle as O
from stair_amd import spec

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('G,T,H,npairs', [(3, 8, 64, 5), (4, 64, 512, 9), (1, 33, 128, 1)])
def test_cosine_attn_backward_matches_autograd(G, T, H, npairs):
    from stair_amd import ops
    g = torch.Generator().manual_seed(G + T + H)
    F = torch.randn(G, T, H, generator=g, dtype=torch.float64, requires_grad=True)
    K = torch.randn(npairs + 2, H, generator=g, dtype=torch.float64, requires_grad=True)
    f_idx = torch.randint(0, G, (npairs,), generator=g, dtype=torch.int32)
    k_idx = torch.randint(0, npairs + 2, (npairs,), generator=g, dtype=torch.int32)       # repeats: tiles and rows are shared by pairs
    d_att = torch.randn(npairs, T, generator=g, dtype=torch.float64)
    att = torch.stack([O.op_existsframe(K[int(k_idx[p])], F[int(f_idx[p])]).reshape(T) for p in range(npairs)])
    (att * d_att).sum().backward()
    d = lambda t: t.detach().float().to(DEV)
    got = ops.cosine_attn(d(F), f_idx.to(DEV), d(K), k_idx.to(DEV), npairs, T, H)
    assert float((got.cpu().double() - att.detach()).abs().max()) < 1e-5
    dF, dK = ops.cosine_attn_bwd(d(F), f_idx.to(DEV), d(K), k_idx.to(DEV), d(d_att), npairs, T, H)
    for a, ref in ((dF, F.grad), (dK, K.grad)):
        scale = max(1.0, float(ref.abs().max()))
        assert float((a.cpu().double() - ref).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize('mode', ['while', 'before', 'after', 'between'])
@pytest.mark.parametrize('T,conv', [(64, True), (24, True), (8, False)])
def test_temporal_relate_backward_matches_autograd(mode, T, conv):
    from stair_amd import ops
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, max_video_length=T if not conv else 64)
    w = {k: (v.double().requires_grad_(True) if 'Temporal.relate' in k else v) for k, v in oracle_weights(config, seed=3).items()}
    assert (w['submodules.Temporal.relate.before.0.weight'].dim() == 3) == conv
    g = torch.Generator().manual_seed(T)
    n, Ks = 5, [1, 2, 1, 2, 2]
    rows = sum(Ks)
    att = torch.rand(rows, T, generator=g, dtype=torch.float64, requires_grad=True)
    d_out = torch.randn(n, T, generator=g, dtype=torch.float64)
    starts = np.concatenate([[0], np.cumsum(Ks)])[:-1]
    outs = [O.temporal_relate(w, mode, att[int(starts[i]): int(starts[i]) + Ks[i]].mean(dim=0)) for i in range(n)]
    (torch.stack(outs) * d_out).sum().backward()
    modes = {'while': 0, 'before': 1, 'after': 2, 'between': 3}
    names = ['submodules.Temporal.relate.%s.%d.%s' % (mode, l, p) for l in (0, 2, 4) for p in ('weight', 'bias')] if mode != 'while' else None
    w6 = [w[nm].detach().float().reshape(-1 if conv and nm.endswith('weight') else w[nm].shape).contiguous().to(DEV) for nm in names] if names else [None] * 6
    d = lambda t: t.detach().float().to(DEV)
    att_idx = torch.tensor(starts, dtype=torch.int32, device=DEV)
    att_k = torch.tensor(Ks, dtype=torch.int32, device=DEV)
    ksize = int(w['submodules.Temporal.relate.before.0.weight'].numel()) if conv else 0
    got = ops.temporal_relate(d(att), att_idx, att_k, n, T, modes[mode], conv, ksize, w6)
    assert float((got.cpu().double() - torch.stack(outs).detach()).abs().max()) < 1e-5
    d_att, dws = ops.temporal_relate_bwd(d(att), att_idx, att_k, d(d_out), n, T, modes[mode], conv, ksize, w6)
    assert float((d_att.cpu().double() - att.grad).abs().max()) < 2e-5 * max(1.0, float(att.grad.abs().max()))
    if names:
        for nm, dw in zip(names, dws):
            ref = w[nm].grad.reshape(dw.shape)
            assert float((dw.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), nm
