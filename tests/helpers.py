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


def window_fixture():
    """tests/golden/window.npz (the reference's own train_module.main() run for two 32-question windows): returns
    (z, meta, questions with numpy golds, per-window reference records).  A record = {'module': [...], 'decoder':
    [...], 'contrastive': [...]} -- the criterion values in the reference's call order, split by kind (the contrastive
    modules are only ever scored in the pooled pass at the end of a window, train_module.py:360-366,388-406)."""
    z, meta = load_golden('window')
    config, T = meta['config'], meta['T']
    qs = []
    for i, form in enumerate(meta['forms']):
        q = synth.make_question(config, meta['question_seed'], i, form=form, T=T)
        q['sg_res_by_step'] = synth.make_gold(config, meta['gold_seed'], q, T=T, keep=meta['gold_keep'])
        qs.append(q)
    records, cur, n_dec = [], {'module': [], 'decoder': [], 'contrastive': []}, 0
    for module, value in zip(meta['loss_modules'], z['loss_values']):
        if module in ('Filter', 'Superlative', 'ToAction'):
            cur['contrastive'].append((module, float(value)))
            continue
        if n_dec == meta['window']:                            # first non-contrastive call after a full window
            records.append(cur)
            cur, n_dec = {'module': [], 'decoder': [], 'contrastive': []}, 0
        if module == 'decoder':
            cur['decoder'].append(float(value))
            n_dec += 1
        else:
            cur['module'].append((module, float(value)))
    records.append(cur)
    assert len(records) * meta['window'] == len(qs)
    return z, meta, qs, records


def window_weights(z, meta, window_no, name, tensor):
    """The reference's value of parameter `name` after `window_no` optimizer steps, and `tensor` subsampled the same way."""
    ref = np.asarray(z['w%d/%s' % (window_no, name)], dtype=np.float64)
    t = np.asarray(torch.as_tensor(tensor).detach().cpu().reshape(-1), dtype=np.float64)
    if t.size > meta['large_threshold']:
        t = t[::meta['stride_large']]
    assert t.shape == ref.shape, (name, t.shape, ref.shape)
    return ref, t
