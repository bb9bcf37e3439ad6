"""CPU tests of stair_amd/data.py against tests/golden/agqa_mini/: a miniature AGQA data directory plus the items the
REFERENCE's AGQADataset built from it (tests/golden/make_dataset_golden.py).  Tokenizer = str.split on both sides
(nltk is absent); every question word is in the GloVe fixture, so no random out-of-vocabulary vectors are involved."""
import json
import os

import numpy as np
import pytest
import torch

from stair_amd import data as D, spec

MINI = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'agqa_mini')
EXP = json.load(open(os.path.join(MINI, 'expected.json')))
TEN = np.load(os.path.join(MINI, 'expected.npz'))
MAXLEN = 10


def _dataset(split, **filt):
    recs = D.filter_records(D.load_question_records(os.path.join(MINI, 'records.json')), split, **filt)
    clips = D.load_clip_features(os.path.join(MINI, 'clips'), {r['video_id'] for r in recs}, MAXLEN)
    return D.AGQAQuestions(recs, clips, D.load_glove(os.path.join(MINI, 'glove.txt')),
                           D.load_answer_vocab(os.path.join(MINI, 'vocab.json')), split=split,
                           video_secs=json.load(open(os.path.join(MINI, 'video_secs.json'))), tokenize=str.split)


def _same(a, b):
    """compare a python value with its JSON round trip (tuples became lists)"""
    if isinstance(a, (tuple, list)):
        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return sorted(map(str, a)) == sorted(map(str, b)) and all(_same(v, b[k] if k in b else b[str(k)]) for k, v in a.items())
    if isinstance(a, float):
        return b == pytest.approx(a, rel=1e-12, abs=0)
    return a == b


@pytest.mark.parametrize('split', ['train', 'test'])
def test_items_equal_the_reference_datasets(split):
    ds = _dataset(split)
    assert len(ds) == EXP[split + '/len']
    for i in range(len(ds)):
        item, exp = ds[i], EXP['%s/%d' % (split, i)]
        extra = set(item) - set(exp)
        assert extra <= {'video_id'} and set(exp) <= set(item)         # video_id is this loader's addition
        for k, v in exp.items():
            got = item[k]
            if v == '@tensor':
                ref = TEN['%s/%d/%s' % (split, i, k)]
                assert tuple(got.shape) == ref.shape and got.dtype == torch.from_numpy(ref).dtype, k
                assert np.array_equal(got.numpy(), ref), k             # bit-exact: same arithmetic, same order
            elif k == 'sg_res_by_step':
                assert sorted(map(str, got)) == sorted(v)
                for key, val in got.items():
                    e = v[str(key)]
                    if 'classes' in e:
                        assert [name for name, _ in val] == e['classes']
                        for n, (_, emb) in enumerate(val):
                            assert np.array_equal(emb.numpy(), TEN['%s/%d/sg/%s/%d' % (split, i, key, n)])
                    else:
                        assert _same(val, e['value']), (key, val, e['value'])
            elif k == 'prog_str_to_question_tokens':
                assert {str(a): list(b) for a, b in got.items()} == v
            else:
                assert got == v, k


def test_clip_loading_rules():
    clips = D.load_clip_features(os.path.join(MINI, 'clips'), {'AAA11', 'BBB22'}, MAXLEN)
    assert sorted(clips) == ['AAA11', 'BBB22']                          # unused clips are not read
    raw = np.load(os.path.join(MINI, 'clips', 'BBB22.npy'))
    assert raw.shape == (26, 16) and tuple(clips['BBB22'].shape) == (MAXLEN, 16)
    assert np.array_equal(clips['BBB22'].numpy(), raw[0:26:2][:MAXLEN])  # every second frame, then the cap (dataset.py:139-142)
    assert tuple(clips['AAA11'].shape) == (8, 16)
    ds = _dataset('test')
    assert ds[0]['video_features'] is ds[1]['video_features']           # one tensor per clip: what clip sharing keys on
    with pytest.raises(ValueError):
        D.load_clip_features(os.path.join(MINI, 'nowhere'), {'AAA11'}, MAXLEN)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='resnet_features'):
            D.load_clip_features(os.path.join(MINI, 'glove.txt'), {'AAA11'}, MAXLEN, str2num={'AAA11': 1})


def test_record_filters_vocab_and_formats(tmp_path):
    recs = D.load_question_records(os.path.join(MINI, 'records.json'))
    assert len(recs) == 7 and recs[5]['nmn_program_span_by_word'] == {1: (None, None)}
    train = D.filter_records(D.load_question_records(os.path.join(MINI, 'records.json')), 'train')
    assert [r['qa_id'] for r in train] == ['qa-0', 'qa-1', 'qa-2', 'qa-3', 'qa-4', 'qa-6'] and train[2]['sg_res_by_step'] == {}
    novel = D.filter_records(D.load_question_records(os.path.join(MINI, 'records.json')), 'train', novel_comp=1)
    assert [r['qa_id'] for r in novel] == EXP['train_novel1/qa_ids']
    # the vocabulary the reference created from the train split (dataset.py:71-85)
    assert D.build_answer_vocab(train) == D.load_answer_vocab(os.path.join(MINI, 'vocab.json'))
    # pkl / jsonl / json round trips of the record list
    for name in ('r.pkl', 'r.jsonl', 'r.json'):
        D.save_question_records(recs, str(tmp_path / name))
        back = D.load_question_records(str(tmp_path / name))
        assert back == recs, name
    # unknown words: uniform noise like np.random.rand (dataset.py:253), repeatable by default, caller's rng otherwise
    ds = _dataset('test')
    a, b = ds.embed_sent('the zebra'), ds.embed_sent('the zebra')
    assert torch.equal(a, b) and 0.0 <= float(a[1].min()) and float(a[1].max()) < 1.0 and tuple(a.shape) == (2, 8)
    ds.rng = np.random.default_rng(1)
    assert not torch.equal(ds.embed_sent('zebra'), ds.embed_sent('zebra'))


def test_pack_questions_stages_each_clip_once():
    ds = _dataset('test')
    items = [ds[i] for i in (4, 5, 6)]                                  # three questions, one clip (CCC33, 5 frames)
    b = D.pack_questions(items, 'cpu')
    assert b.n_clips == 1 and tuple(b.video.shape) == (1, 5, 16) and b.video_index == [0, 0, 0]
    assert b.q_lens == [int(i['question'].shape[0]) for i in items] and b.question.shape[0] == sum(b.q_lens)
    assert b.answers.tolist() == [int(i['answer']) for i in items]
    assert b.h2d_bytes == 4 * (5 * 16 + sum(b.q_lens) * 8 + 3)
    flat = D.pack_questions(items, 'cpu', share_clips=False)
    assert flat.n_clips == 3 and flat.video_index is None and torch.equal(flat.video[2], b.video[0])
    mixed = D.pack_questions([ds[0], ds[4]], 'cpu')                     # 8 vs 5 frames: padded to 8, lengths kept
    assert mixed.video_len == [8, 5] and tuple(mixed.video.shape) == (2, 8, 16)
    assert torch.equal(mixed.video[1, :5], b.video[0]) and float(mixed.video[1, 5:].abs().max()) == 0.0
    assert b.video_len is None


def test_checkpoint_round_trip_and_refusal_of_pickled_modules(tmp_path):
    from stair_amd import synth
    from stair_amd.module_net import VideoNMN
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    m = VideoNMN(config)
    w = synth.make_weights(config, 3)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    D.save_checkpoint(str(tmp_path / 'ck'), m, config)
    assert sorted(os.listdir(tmp_path / 'ck')) == ['config.json', 'pytorch_model.bin']     # train_module.py:212-216 layout
    m2, cfg2 = D.load_checkpoint(str(tmp_path / 'ck'))
    assert cfg2 == config and list(m2.state_dict()) == list(m.state_dict())
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # a pickled module object (what train_module.py:214 writes) is refused, not executed
    os.makedirs(tmp_path / 'bad')
    json.dump(config, open(tmp_path / 'bad' / 'config.json', 'w'))
    torch.save(torch.nn.Linear(2, 2), str(tmp_path / 'bad' / 'pytorch_model.bin'))
    with pytest.raises(ValueError, match='state_dict'):
        D.load_checkpoint(str(tmp_path / 'bad'))


def test_bf16_clip_storage_and_packing(tmp_path):
    """load_clip_features(dtype='bf16') rounds once at staging (BASELINE configs[1]); pack_questions keeps such clips bf16."""
    rng = np.random.default_rng(0)
    d = tmp_path / 'clips'
    d.mkdir()
    for vid, frames in (('A1', 20), ('B2', 14)):
        np.save(d / (vid + '.npy'), rng.standard_normal((frames, 64)).astype(np.float32))
    f32 = D.load_clip_features(str(d), ['A1', 'B2'], 8)
    b16 = D.load_clip_features(str(d), ['A1', 'B2'], 8, dtype='bf16')
    assert b16['A1'].dtype == torch.bfloat16 and tuple(b16['A1'].shape) == (8, 64) and tuple(b16['B2'].shape) == (7, 64)
    assert torch.equal(b16['A1'], f32['A1'].to(torch.bfloat16))
    with pytest.raises(ValueError):
        D.load_clip_features(str(d), ['A1'], 8, dtype='fp8')
    items = [{'video_features': b16[v], 'question': torch.zeros(3, 8), 'nmn_program_list': ['x'], 'prog_str_to_question_tokens': {},
              'answer': torch.tensor(1), 'video_id': v} for v in ('A1', 'B2', 'A1')]
    b = D.pack_questions(items, 'cpu')
    assert b.video.dtype == torch.bfloat16 and tuple(b.video.shape) == (2, 8, 64) and b.video_len == [8, 7]
    assert b.h2d_bytes == 2 * 8 * 64 * 2 + 9 * 8 * 4 + 3 * 4
    mixed = [dict(items[0]), dict(items[1], video_features=f32['B2'])]
    assert D.pack_questions(mixed, 'cpu').video.dtype == torch.float32


def test_dataset_pickles_cannot_run_code(tmp_path):
    """The reference's .pkl formats hold plain data; a file that names any other global is refused, not imported."""
    import pickle
    recs = [{'question': 'q', 'answer': 'yes', 'nmn_program_span_by_word': {0: (1, 2)}, 'emb': np.arange(6, dtype=np.float32).reshape(2, 3)}]
    p = tmp_path / 'ok.pkl'
    D.save_question_records(recs, str(p))
    back = D.load_question_records(str(p))
    assert back[0]['nmn_program_span_by_word'] == {0: (1, 2)} and np.array_equal(back[0]['emb'], recs[0]['emb'])

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo pwned > %s' % (tmp_path / 'pwned'),))
    bad = tmp_path / 'bad.pkl'
    with open(bad, 'wb') as f:
        pickle.dump([Evil()], f)
    with pytest.raises(pickle.UnpicklingError, match='refusing'):
        D.load_question_records(str(bad))
    with pytest.raises(pickle.UnpicklingError, match='refusing'):
        D.load_glove(str(bad))
    assert not (tmp_path / 'pwned').exists()
