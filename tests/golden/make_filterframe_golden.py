#!/usr/bin/env python3
"""Regenerates tests/golden/criteria_filterframe.npz: the REFERENCE's CriterionByModule on FilterFrame predictions
(/root/reference/train_module.py:141-155: BCELoss(Softmax(dim=1)(pred [T,O]), row-normalised interval masks)), loss
value and gradient w.r.t. the prediction, for gold dicts with one / several / overlapping / out-of-range intervals
and a word->id file in which two words share an id (:49-54).

    python tests/golden/make_filterframe_golden.py        (build container only)
"""
import contextlib
import importlib.machinery
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402

WORD2ID = {'cup': 'o1', 'glass': 'o1', 'dish': 'o2', 'door': 'o5', 'phone': 'o3', 'sofa': 'o9', 'blanket': 'o4',
           'window': 'o7', 'food': 'o8', 'bag': 'o6'}


def main():
    G.import_reference()
    tbx = types.ModuleType('tensorboardX'); tbx.SummaryWriter = object
    tbx.__spec__ = importlib.machinery.ModuleSpec('tensorboardX', None); sys.modules['tensorboardX'] = tbx
    import train_module
    train_module.device = 'cpu'
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump(WORD2ID, f)
    with contextlib.redirect_stdout(io.StringIO()):
        crit = train_module.CriterionByModule(types.SimpleNamespace(word2id_filename=f.name))
    O = len(crit.id2index)
    g = torch.Generator().manual_seed(11)
    out, cases = {}, []
    golds = [
        (40, {'cup': (3.2, 17.9)}),
        (40, {'cup': (3.2, 17.9), 'dish': (10.0, 30.5), 'door': (0.0, 40.0)}),
        (40, {'glass': (5.0, 9.0), 'cup': (20.0, 22.5)}),                 # two words, one id: the later one overwrites
        (24, {'phone': (12.4, 12.9), 'sofa': (23.5, 25.0), 'bag': (-1.0, 0.4)}),
        (8, {'food': (0.0, 8.0), 'window': (2.0, 2.0)}),
        (40, {}),                                                         # no gold entity: all-zero target
    ]
    for T, gold in golds:
        pred = (torch.randn(T, O, generator=g) * 2).requires_grad_(True)
        loss = crit('FilterFrame', pred, gold)
        loss.backward()
        i = len(cases)
        out['c%d/pred' % i], out['c%d/loss' % i], out['c%d/dpred' % i] = pred.detach().numpy(), loss.detach().numpy(), pred.grad.numpy()
        cases.append({'T': T, 'gold': {k: list(v) for k, v in gold.items()}})
    meta = {'cases': cases, 'word2id': WORD2ID, 'word2index': crit.word2id, 'O': O}
    out['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'criteria_filterframe.npz'), **out)
    print('wrote', len(cases), 'cases, O =', O, [float(out['c%d/loss' % i]) for i in range(len(cases))])


if __name__ == '__main__':
    main()
