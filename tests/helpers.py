"""Shared helpers for the parity tests (oracle is test infrastructure; see oracle/nmn_oracle.py)."""
import json
import os

import numpy as np
import torch

from stair_amd import synth
from oracle import nmn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PRETRAIN_MODULES = frozenset({'Exists', 'Xor', 'Equals', 'Filter', 'ToAction', 'FilterFrame', 'ExistsFrame',
                              'Superlative', 'Localize', 'Temporal', 'decoder'})


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    return z, meta


def oracle_weights(config, seed=0):
    return O.to_torch(synth.make_weights(config, seed))


def question_for(meta, q):
    return synth.make_question(meta['config'], meta['seed'], q['qid'], form=q['form'], T=meta['T'])


def compare_with_reference_grads(fixture, get_grad, tol_rel=2e-4):
    """Check gradients against tests/golden/<fixture>.npz (the reference's own loss.backward()).
    get_grad(name) -> flat float tensor/ndarray of the full gradient (or None)."""
    z, meta = load_golden(fixture)
    stride, thr = meta['stride_large'], meta['large_threshold']
    worst = (0.0, '')
    n_checked = 0
    for key in z.files:
        if not key.startswith('grad/'):
            continue
        name = key[5:]
        if name.startswith('submodules.Superlative.localize_module.'):
            continue                                           # alias of Localize.* (same tensor, module_net.py:31-32)
        ref = np.asarray(z[key], dtype=np.float64)
        g = get_grad(name)
        assert g is not None, name
        g = np.asarray(torch.as_tensor(g).detach().cpu().reshape(-1), dtype=np.float64)
        if g.size > thr:
            g = g[::stride]
        assert g.shape == ref.shape, (name, g.shape, ref.shape)
        tol = tol_rel * max(float(np.abs(ref).max()), 1e-3)
        err = float(np.abs(g - ref).max()) if ref.size else 0.0
        worst = max(worst, (err / tol, name))
        assert err < tol, (name, err, tol)
        n_checked += 1
    assert n_checked > 80
    return worst, z, meta
