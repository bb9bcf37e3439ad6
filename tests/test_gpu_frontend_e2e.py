"""End to end from TEXT: AGQA program string + question sentence -> frontend.parse_program ->
frontend.match_spans -> ProgramCache / plan -> HIP forward, against the oracle fed the SAME question dict.  The parse and
the spans are pinned to the reference by tests/golden/frontend.json and tests/golden/spans.json (CPU tests); here the whole
chain runs on the GPU path and must give the oracle's logits (1e-4) and its top-1 answer."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import frontend as F, spec, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PROGRAMS = json.load(open(os.path.join(GOLD, 'programs.json')))
SPANS = json.load(open(os.path.join(GOLD, 'spans.json')))
KEYS = ['P0', 'P1', 'P2', 'P3', 'P4', 'P5', 'P6', 'P7', 'C0', 'C1', 'C2']


def _question(config, key, seed):
    case = SPANS[key]
    if key.startswith('P'):                                   # from the AGQA program STRING
        nmn, more = F.parse_program(PROGRAMS[key]['string'])
        idx = more['idx_list']
        assert nmn == case['nmn']
    else:                                                     # coverage forms exist as NMN lists only
        nmn, idx = PROGRAMS[key]['nmn'], PROGRAMS[key]['idx']
    nz = F.Normaliser()
    by_word, _ = F.match_spans(nmn, case['question'], nz)
    assert {str(k): list(v) for k, v in by_word.items()} == case['by_word']          # == the reference matcher's spans
    Q = len(nz.tokenize(case['question']))
    T = config['max_video_length']
    return {'question': synth.normal(seed, key + '/question', (Q, config['text_size'])),
            'video_features': synth.normal(seed, key + '/video', (T, config['video_size'])),
            'prog_str_to_question_tokens': by_word, 'nmn_program_list': nmn, 'nmn_program_idx': idx,
            'answer': 0, 'qa_id': key}


@pytest.mark.parametrize('size', ['tiny', 'full'])
def test_program_strings_and_question_text_to_logits(size):
    from stair_amd.module_net import VideoNMN
    if size == 'tiny':
        config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    else:
        config = dict(spec.DEFAULT_CONFIG)
    weights = synth.make_weights(config, 6)
    model = VideoNMN(config)
    model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
    model = model.to(DEV)
    qs = [_question(config, k, 6) for k in KEYS]
    res = model.forward_batch(qs)
    logits, pred = res.logits.cpu(), res.pred.cpu()
    w = O.to_torch(weights)
    for i, q in enumerate(qs):
        ref = O.forward(w, config, q, return_res_by_step=False)['logits']
        err = float((logits[i] - ref).abs().max())
        assert err < 1e-4, (q['qa_id'], err)
        assert int(pred[i]) == int(torch.argmax(ref)), q['qa_id']


def test_unmatched_phrase_fails_like_the_reference():
    """A phrase the matcher cannot place has span (None, None); the reference then averages token_feature[None:None]
    silently (module_net.py:128-129); here it is refused when the program is packed (KeyError naming the token)."""
    from stair_amd.module_net import VideoNMN
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    weights = synth.make_weights(config, 6)
    model = VideoNMN(config)
    model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
    model = model.to(DEV)
    case = SPANS['X2']
    by_word, _ = F.match_spans(case['nmn'], case['question'], F.Normaliser())
    q = {'question': synth.normal(6, 'x2/q', (8, config['text_size'])), 'video_features': synth.normal(6, 'x2/v', (40, config['video_size'])),
         'prog_str_to_question_tokens': by_word, 'nmn_program_list': case['nmn'], 'nmn_program_idx': [None] * len(case['nmn']), 'answer': 0}
    with pytest.raises(KeyError):
        model.forward_batch([q])
