#!/usr/bin/env python3
"""Regenerates tests/golden/window.npz by running the REFERENCE's own training loop
(/root/reference/train_module.py:273-439 `main(args)`) on CPU for TWO optimizer windows of 32 questions:
per-question intermediate losses (:351-373), decoder CE (:376-380), the 32-question contrastive pooling
(:388-406), one backward per window (:408), Adam (:326) and LambdaLR (:328-332, :412) -- exactly as main() runs them.

What is replaced around main() (none of it is part of the loop under test):
    AGQADataset / DataLoader   -> 64 synthetic questions of stair_amd.synth (make_question + make_gold), fixed order
    VideoNMN                   -> the reference's VideoNMN, constructed by main() as usual, then loaded with the
                                  deterministic weights of stair_amd.synth (main() itself would draw random ones)
    CriterionByModule          -> the reference's class, subclassed only to RECORD every (module, loss) it returns
    SummaryWriter              -> records the scalars main() logs (the learning rate after each window)
    evaluate_by_module         -> called by main() every `evaluate_interval` = 32 questions: snapshots the weights
Dropout is 0 (torch's Philox masks cannot be reproduced).  Stored: the questions' forms, every criterion
value in call order, the learning rates, and all weights after window 1 and after window 2 (large tensors
subsampled) -- outputs only.

    python tests/golden/make_window_golden.py        (build container only: needs /root/reference)
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
from stair_amd import spec, synth  # noqa: E402

NAME = 'tiny_conv'
QSEED, GSEED = 11, 4               # question / gold seeds (the model weights use make_golden.SEED)
WINDOW, N_WINDOWS = 32, 2
LR, TOTAL_ITERS = 2e-4, 10         # a short LambdaLR ramp so that the two windows see different learning rates
STRIDE, THRESHOLD = 3, 4096
WORD2ID = {'w%d' % i: 'o%d' % (i % 10) for i in range(12)}      # 10 distinct ids = tiny_conv's object_types


def window_forms(n):
    """Program forms of the n questions: every form appears in each window, in a different order per window."""
    forms = []
    for w in range(n // WINDOW):
        base = [synth.ALL_FORMS[(3 * w + i) % len(synth.ALL_FORMS)] for i in range(WINDOW)]
        forms += base
    return forms


def make_batches(config, T):
    forms = window_forms(WINDOW * N_WINDOWS)
    batches = []
    for i, form in enumerate(forms):
        d = synth.make_question(config, QSEED, i, form=form, T=T)
        gold = synth.make_gold(config, GSEED, d, T=T, keep=0.9)
        sg = {k: ([(n, torch.from_numpy(e)) for n, e in v] if isinstance(v, list) else v) for k, v in gold.items()}
        batches.append({'question': torch.from_numpy(d['question']), 'video_features': torch.from_numpy(d['video_features']),
                        'prog_str_to_question_tokens': d['prog_str_to_question_tokens'],
                        'nmn_program_list': d['nmn_program_list'], 'nmn_program_idx': d['nmn_program_idx'],
                        'sg_res_by_step': sg, 'answer': torch.tensor(int(d['answer'])), 'qa_id': d['qa_id']})
    return forms, batches


def main():
    VideoNMN, _ = G.import_reference()
    tbx = types.ModuleType('tensorboardX'); tbx.SummaryWriter = object
    tbx.__spec__ = importlib.machinery.ModuleSpec('tensorboardX', None); sys.modules['tensorboardX'] = tbx
    import train_module
    train_module.device = 'cpu'
    torch.set_num_threads(4)
    config = G.CONFIGS[NAME]
    T = config['max_video_length']
    forms, batches = make_batches(config, T)
    weights = synth.make_weights(config, G.SEED)

    calls, scalars, snapshots = [], [], []

    class Dataset:
        def __init__(self, args, split):
            pass

        def answer_vocab_length(self):
            return config['answer_vocab_length']

    class Loader:
        def __init__(self, dataset, batch_size, shuffle, num_workers, collate_fn):
            assert batch_size == 1

        def __iter__(self):
            return iter(batches)

    class RecordingCriterion(train_module.CriterionByModule):
        def __call__(self, module_name, pred, gold):
            loss = super().__call__(module_name, pred, gold)
            calls.append((module_name, float(loss.detach())))
            return loss

    class Writer:
        def __init__(self, logdir):
            pass

        def add_scalar(self, tag, value, step):
            scalars.append((tag, float(value), int(step)))

    def model_factory(model_config, debug=False, pretrain_modules=()):
        assert model_config['object_types'] == config['object_types'] and model_config['have_pretrain_head']
        model = VideoNMN(model_config, debug=debug, pretrain_modules=pretrain_modules)
        sd = {k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)}
        assert list(model.state_dict().keys()) == list(sd.keys())
        model.load_state_dict(sd)
        return model

    def snapshot(args, loader, model, criterions, preds_file=None):
        assert model.training is False                       # main() switches to eval around the validation pass
        snapshots.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        return 0.0, {'decoder': 0.0}                         # valid_acc 0 -> no checkpoint is written

    train_module.AGQADataset, train_module.DataLoader = Dataset, Loader
    train_module.CriterionByModule, train_module.SummaryWriter = RecordingCriterion, Writer
    train_module.VideoNMN, train_module.evaluate_by_module = model_factory, snapshot

    out_dir = tempfile.mkdtemp()
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump(WORD2ID, f)
    args = types.SimpleNamespace(
        dataset='AGQA', debug=True, num_workers=0, model_ckpt=None, config_filename=None, output=out_dir, result_filename=None,
        hidden_size=config['hidden_size'], video_size=config['video_size'], text_size=config['text_size'], dropout=0.0,
        max_video_length=config['max_video_length'], init_method='default', layer_norm=1, word2id_filename=f.name,
        module_loss_weight=1.0, decoder_loss_weight=1.0, train_module_before_iters=1e10, train_decoder_after_iters=0,
        modules_no_intermediate_train=['FilterFrame'], lr=LR, weight_decay=0, scheduler_start_factor=1.0,
        scheduler_end_factor=0.1, scheduler_total_iters=TOTAL_ITERS, num_epochs=1, gradient_accumulation=WINDOW,
        report_interval=WINDOW, evaluate_interval=WINDOW)
    with contextlib.redirect_stdout(io.StringIO()):
        train_module.main(args)
    assert len(snapshots) == N_WINDOWS

    out = {}
    for wi, snap in enumerate(snapshots):
        for k, v in snap.items():
            if k.startswith('submodules.Superlative.localize_module.'):
                continue                                      # the same tensors as submodules.Localize.* (module_net.py:31-32)
            flat = v.reshape(-1).numpy()
            out['w%d/%s' % (wi + 1, k)] = flat if flat.size <= THRESHOLD else flat[::STRIDE]
    lrs = [v for tag, v, _ in scalars if tag == 'lr/lr']
    assert len(lrs) == N_WINDOWS
    out['loss_values'] = np.asarray([v for _, v in calls], dtype=np.float64)
    meta = {'config': dict(config, dropout=0.0), 'T': T, 'seed': G.SEED, 'question_seed': QSEED, 'gold_seed': GSEED,
            'gold_keep': 0.9, 'forms': forms, 'window': WINDOW, 'lr': LR, 'scheduler_total_iters': TOTAL_ITERS,
            'lr_after_window': lrs, 'loss_modules': [m for m, _ in calls], 'stride_large': STRIDE,
            'large_threshold': THRESHOLD, 'word2id': WORD2ID}
    out['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'window.npz'), **out)
    print('window.npz: %d criterion calls, %d arrays, lr after each window %s' % (len(calls), len(out), lrs))


if __name__ == '__main__':
    main()
