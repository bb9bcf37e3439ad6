#!/usr/bin/env python3
"""Regenerates tests/golden/validation.json by running the REFERENCE's evaluate_by_module
(/root/reference/train_module.py:219-270) on CPU: the 12 golden questions of tests/golden/tiny_conv.npz with the
synthetic gold intermediates of stair_amd.synth.make_gold, module_loss_weight = 1.  Stored: accuracy, the mean
validation loss per module (contrastive modules switch to the 'cont-valid' cosine metric, :224-227), predictions.

    python tests/golden/make_validation_golden.py        (build container only: needs /root/reference)
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
from stair_amd import synth        # noqa: E402


def main():
    VideoNMN, _ = G.import_reference()
    tbx = types.ModuleType('tensorboardX'); tbx.SummaryWriter = object
    tbx.__spec__ = importlib.machinery.ModuleSpec('tensorboardX', None); sys.modules['tensorboardX'] = tbx
    import train_module
    train_module.device = 'cpu'
    name = 'tiny_conv'
    config = G.CONFIGS[name]
    meta = json.loads(bytes(np.load(os.path.join(HERE, name + '.npz'))['meta']).decode())
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump({'cup': 'o1', 'dish': 'o2'}, f)
    with contextlib.redirect_stdout(io.StringIO()):
        crit = train_module.CriterionByModule(types.SimpleNamespace(word2id_filename=f.name))
    model = G.build_model(VideoNMN, config)              # pretrain_modules = the criterion keys, eval mode

    batches = []
    for q in meta['questions']:
        d = synth.make_question(config, G.SEED, q['qid'], form=q['form'], T=meta['T'])
        gold = synth.make_gold(config, G.SEED, d, T=meta['T'], keep=1.0)
        sg = {k: ([(n, torch.from_numpy(e)) for n, e in v] if isinstance(v, list) else v) for k, v in gold.items()}
        batches.append({'question': torch.from_numpy(d['question']), 'video_features': torch.from_numpy(d['video_features']),
                        'prog_str_to_question_tokens': d['prog_str_to_question_tokens'],
                        'nmn_program_list': d['nmn_program_list'], 'nmn_program_idx': d['nmn_program_idx'],
                        'sg_res_by_step': sg, 'answer': torch.tensor(int(d['answer'])), 'qa_id': 'q%d' % q['qid']})

    A = config['answer_vocab_length']
    vocab = {'word2id': {('w%d' % i): i for i in range(A)}, 'id2word': {i: 'w%d' % i for i in range(A)}}
    vocab['word2id']['<UNK>'] = A - 1
    vocab['id2word'][A - 1] = '<UNK>'

    class _Loader:
        dataset = types.SimpleNamespace(answer_vocab=vocab)

        def __iter__(self):
            return iter(batches)

    args = types.SimpleNamespace(module_loss_weight=1.0, gradient_accumulation=32)
    with contextlib.redirect_stdout(io.StringIO()):
        acc, valid_losses = train_module.evaluate_by_module(args, _Loader(), model, crit, preds_file=None)
    out = {'config': name, 'accuracy': acc, 'unk_token_id': A - 1,
           'valid_losses': {k: (None if v == float('inf') else v) for k, v in valid_losses.items()}}
    json.dump(out, open(os.path.join(HERE, 'validation.json'), 'w'), indent=0, sort_keys=True)
    print(out)


if __name__ == '__main__':
    main()
