#!/usr/bin/env python3
"""Regenerates tests/golden/agqa_mini/: a miniature AGQA data directory in the reference's on-disk formats and the
items the REFERENCE's AGQADataset (/root/reference/video_nmn/dataset.py:31-233) builds from it.

    clips/<video_id>.npy   3 clips, I3D-style [frames, 16] float32 (one of them longer than max_video_length)
    records.json           7 question records in the layout of utils/agqa_lite.py:122-143 (the reference reads the same
                           list from a pickle; this script pickles it into a temp dir for the reference run)
    glove.txt, video_secs.json                       the side tables
    expected.npz / expected.json                     every tensor / every other field of dataset[i], train and test split
    vocab.json                                       the answer vocabulary the reference creates on first use

nltk is absent here, so the reference runs with word_tokenize = str.split (make_golden.import_reference) and the test
passes the same tokenizer to stair_amd.data.AGQAQuestions.  Every question word is in glove.txt (the reference draws
np.random.rand for unknown words, which no fixture can pin).

    python tests/golden/make_dataset_golden.py            (build container only)
"""
import argparse
import contextlib
import io
import json
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'agqa_mini')
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402
from stair_amd import synth, frontend        # noqa: E402

MAXLEN = 10
WORDS = ('the was person holding a dish before opening door did they touch blanket after eating sandwich what which '
         'longer ? object while sitting on sofa phone').split()


def records():
    P = json.load(open(os.path.join(HERE, 'frontend.json')))['cases']
    def rec(i, vid, question, answer, case, spans, gold, novel=0, more=0):
        return {'question': question, 'answer': answer, 'video_id': vid, 'program': P[case]['string'], 'qa_id': 'qa-%d' % i,
                'novel_comp': novel, 'more_steps': more, 'nmn_program': P[case]['nmn'], 'nmn_program_idx': P[case]['idx'],
                'sg_program': ['unused'], 'sg_program_idx': [0], 'sg_res_by_step': gold,
                'nmn_program_span_by_word': spans, 'nmn_program_span_by_char': {k: (0, 1) for k in spans}}
    return [
        rec(0, 'AAA11', 'was the person holding a dish ?', 'yes', 'P1', {1: (5, 6)}, {2: True, 3: 'dish'}),
        rec(1, 'AAA11', 'did they touch the blanket before opening the door ?', 'no', 'F0', {6: (7, 10), 7: (2, 3)},
            {5: ((3.0, 9.0),), 4: (0.0, 3.0), 3: ['blanket', 'the door']}, novel=1),
        rec(2, 'BBB22', 'which object did they touch after eating a sandwich ?', 'blanket', 'F0', {6: (6, 9), 7: (4, 5)}, None),
        rec(3, 'BBB22', 'what did the person touch while sitting on the sofa ?', 'phone', 'F3', {1: (0, 1), 8: (6, 10), 9: (4, 5)},
            {8: ((1.5, 4.5), (10.0, 12.0)), 2: {'sofa': (2.0, 8.0), 'phone': (0.0, 1.0)}}, more=1),
        rec(4, 'CCC33', 'was the person holding a phone ?', 'zebra', 'P1', {1: (5, 6)}, {2: False}),
        rec(5, 'CCC33', 'was the person eating a sandwich ?', 'yes', 'P1', {1: (None, None)}, {}),      # dropped by train/valid
        rec(6, 'CCC33', 'did they touch the door ?', 'blanket', 'P1', {1: (4, 5)}, {3: 'door'}),
    ]


def dump_item(item, prefix, tensors, plain):
    out = {}
    for k, v in item.items():
        if isinstance(v, torch.Tensor):
            tensors[prefix + k] = v.numpy()
            out[k] = '@tensor'
        elif k == 'sg_res_by_step':
            g = {}
            for key, val in v.items():
                if isinstance(val, list) and val and isinstance(val[0], tuple) and isinstance(val[0][1], torch.Tensor):
                    for n, (name, emb) in enumerate(val):
                        tensors['%ssg/%s/%d' % (prefix, key, n)] = emb.numpy()
                    g[str(key)] = {'classes': [name for name, _ in val]}
                else:
                    g[str(key)] = {'value': val}
            out[k] = g
        elif k == 'prog_str_to_question_tokens':
            out[k] = {str(a): list(b) for a, b in v.items()}
        else:
            out[k] = v
    plain[prefix.rstrip('/')] = out


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, 'clips'))
    rng = np.random.default_rng(0)
    for vid, frames in (('AAA11', 16), ('BBB22', 26), ('CCC33', 9)):
        np.save(os.path.join(OUT, 'clips', vid + '.npy'), rng.standard_normal((frames, 16)).astype(np.float32))
    np.save(os.path.join(OUT, 'clips', 'ZZZ99.npy'), np.zeros((4, 16), np.float32))        # a clip no question uses
    with open(os.path.join(OUT, 'glove.txt'), 'w') as f:
        f.write('%d 8\n' % len(WORDS))
        for w in WORDS:
            f.write(w + ' ' + ' '.join('%.4f' % x for x in rng.standard_normal(8)) + '\n')
    json.dump({'AAA11': 4.0, 'BBB22': 10.5, 'CCC33': 3.0}, open(os.path.join(OUT, 'video_secs.json'), 'w'))
    recs = records()
    json.dump(recs, open(os.path.join(OUT, 'records.json'), 'w'), indent=0)

    G.import_reference()
    from video_nmn.dataset import AGQADataset
    tensors, plain = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        pk = os.path.join(tmp, 'records.pkl')
        pickle.dump(recs, open(pk, 'wb'))
        vocab = os.path.join(tmp, 'vocab.json')
        args = argparse.Namespace(debug=False, rgb_path=os.path.join(OUT, 'clips'), flow_path=None,
                                  video_secs_path=os.path.join(OUT, 'video_secs.json'), str2num_path=None,
                                  train_filename=pk, valid_filename=pk, test_filename=pk, novel_comp=None, more_steps=None,
                                  vocab_filename=vocab, max_video_length=MAXLEN, glove_filename=os.path.join(OUT, 'glove.txt'),
                                  shuffle_video=False)
        for split in ('train', 'test'):
            with contextlib.redirect_stdout(io.StringIO()):
                ds = AGQADataset(args, split)
            plain[split + '/len'] = len(ds)
            for i in range(len(ds)):
                dump_item(ds[i], '%s/%d/' % (split, i), tensors, plain)
        shutil.copy(vocab, os.path.join(OUT, 'vocab.json'))
        args.novel_comp = 1
        with contextlib.redirect_stdout(io.StringIO()):
            plain['train_novel1/qa_ids'] = [d['qa_id'] for d in AGQADataset(args, 'train').data]
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **tensors)
    json.dump(plain, open(os.path.join(OUT, 'expected.json'), 'w'), indent=0, sort_keys=True)
    print('wrote', OUT, '-', len(tensors), 'tensors;', plain['train/len'], 'train items,', plain['test/len'], 'test items')


if __name__ == '__main__':
    main()
