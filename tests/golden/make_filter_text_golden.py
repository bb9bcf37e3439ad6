#!/usr/bin/env python3
"""Regenerates tests/golden/filter_text.json by running the REFERENCE's evaluate.get_filter_text_results
(/root/reference/evaluate.py:65-117) on CPU over the 12 golden questions of tests/golden/tiny_conv.npz with a
30-phrase synthetic vocabulary (stair_amd.synth.class_embedding stands in for the dataset's GloVe lookup).
The reference function iterates a dataloader, reads the vocabulary from a json file and pickles its result;
this script hands it an in-memory list of batches and temporary files, then stores the result as JSON together
with the cosine similarities of every ranked phrase (so a test can tell a real mismatch from a near-tie).

    python tests/golden/make_filter_text_golden.py        (build container only: needs /root/reference)
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402  (shares import_reference / build_model / configs)
from stair_amd import synth        # noqa: E402

N_PHRASES = 30


def main():
    VideoNMN, _ = G.import_reference()
    import evaluate as ref_eval     # /root/reference/evaluate.py
    name = 'tiny_conv'
    config = G.CONFIGS[name]
    import numpy as np
    meta = json.loads(bytes(np.load(os.path.join(HERE, name + '.npz'))['meta']).decode())
    model = G.build_model(VideoNMN, config)
    vocab = ['class %d' % c for c in range(N_PHRASES)]
    emb = {v: torch.from_numpy(synth.class_embedding(config, G.SEED, c)) for c, v in enumerate(vocab)}

    batches = []
    for q in meta['questions']:
        d = synth.make_question(config, G.SEED, q['qid'], form=q['form'], T=meta['T'])
        batches.append({'question': torch.from_numpy(d['question']), 'video_features': torch.from_numpy(d['video_features']),
                        'prog_str_to_question_tokens': d['prog_str_to_question_tokens'],
                        'nmn_program_list': d['nmn_program_list'], 'nmn_program_idx': d['nmn_program_idx'],
                        'qa_id': 'q%d' % q['qid']})

    class _Dataset(list):
        def embed_sent(self, sent):
            return emb[sent]

    class _Loader:
        dataset = _Dataset(batches)

        def __iter__(self):
            return iter(batches)

    with tempfile.TemporaryDirectory() as tmp:
        vf, rf = os.path.join(tmp, 'vocab.json'), os.path.join(tmp, 'res.pkl')
        json.dump(vocab, open(vf, 'w'))
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            ref_eval.get_filter_text_results(_Loader(), model, filter_vocab_filename=vf, result_filename=rf)
        res = pickle.load(open(rf, 'rb'))           # a file this script wrote a moment ago

    # similarities of the ranked phrases, from the same reference model (for near-tie diagnosis only)
    reps = torch.stack([model.contrastive_head(model.encode_question_no_grad(emb[v])[1].squeeze()) for v in vocab])
    out = {'vocab': vocab, 'config': name, 'results': {}}
    n_nodes = 0
    for b in batches:
        with torch.no_grad():
            steps = model(b, return_res_by_step=False, return_result_of_each_step=True, test_mode=True)['result_of_each_step']
        entry = {}
        for pidx, (level, kw, top) in res[b['qa_id']].items():
            i = b['nmn_program_idx'].index(pidx)
            sims = torch.nn.CosineSimilarity()(steps[i][1].unsqueeze(0), reps)
            entry[str(pidx)] = {'level': level, 'keyword': kw, 'top': top, 'sims': [float(sims[vocab.index(t)]) for t in top]}
            n_nodes += 1
        out['results'][b['qa_id']] = entry
    json.dump(out, open(os.path.join(HERE, 'filter_text.json'), 'w'), indent=0, sort_keys=True)
    print('wrote filter_text.json: %d questions, %d Filter nodes' % (len(out['results']), n_nodes))


if __name__ == '__main__':
    main()
