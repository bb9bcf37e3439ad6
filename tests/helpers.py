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
