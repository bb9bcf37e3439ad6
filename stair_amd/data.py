"""On-disk formats of the AGQA pipeline -> the executor's packed device layout.

What the reference reads, and where (all /root/reference/video_nmn/dataset.py unless noted):
  * question records      pickled list of dicts written by utils/agqa_lite.py:122-143 (`question`, `answer`,
                          `video_id`, `nmn_program`, `nmn_program_idx`, `nmn_program_span_by_word`, `sg_res_by_step`,
                          `qa_id`, `novel_comp`, `more_steps`, ...), filtered as :41-62;
  * clip features         a directory of `<video_id>.npy` (I3D; every second frame, :134-143) or an h5 file with
                          datasets `ids` + `resnet_features` (per clip [frames, crops, 2048], mean over crops, :145-154),
                          optionally a second h5 `resnext_features` concatenated on the feature axis (:163-172);
  * GloVe table           text (`count dim` header line, then `word v1 .. v300`) or a pickled dict (:235-246);
  * answer vocabulary     json {word2id, id2word}, `yes no before after` first, `<UNK>` last (:71-99);
  * video_secs.json       clip length in seconds, for rescaling gold intervals to T frames (:200-221, :258-261);
  * checkpoints           `config.json` + `pytorch_model.bin` (train_module.py:212-216, evaluate.py:136-139).

`AGQAQuestions` reproduces `AGQADataset.__getitem__` (:174-233) item for item; `pack_questions` turns a list of
items into the tensors `VideoNMN.run_programs` takes (clips deduplicated, one pinned staging buffer per tensor, one
H2D copy each).

Not available in this image, and gated rather than stubbed: h5py (the h5 branch raises ImportError with the
dataset names it would have read) and nltk's word_tokenize (questions are split by `frontend.Normaliser.tokenize`
unless a tokenizer is passed in).  Pickle is only ever applied to files the caller names (their own data).
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch

from . import frontend


# ----------------------------------------------------------------------------------------------
# question records
# ----------------------------------------------------------------------------------------------
class _DataUnpickler(pickle.Unpickler):
    """The reference stores question records and GloVe tables as pickles of plain containers, strings, numbers and numpy
    arrays (utils/agqa_lite.py:122-143, dataset.py:235-246).  Only those may be rebuilt: any other global a file names
    (os.system, a class with __reduce__, ...) is refused instead of imported, so a dataset file cannot run code here."""
    _ALLOWED = {('builtins', n) for n in ('dict', 'list', 'tuple', 'set', 'frozenset', 'str', 'bytes', 'bytearray', 'int', 'float',
                                          'bool', 'complex', 'slice', 'range')} | {
        ('collections', 'OrderedDict'), ('collections', 'defaultdict'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy', 'float32'), ('numpy', 'float64'), ('numpy', 'int32'), ('numpy', 'int64'),
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer')}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError('refusing to load %s.%s from a dataset pickle: only containers, numbers, strings and numpy '
                                         'arrays are data' % (module, name))
        return super().find_class(module, name)


def _load_data_pickle(path):
    with open(path, 'rb') as f:
        return _DataUnpickler(f).load()


def load_question_records(path):
    """List of question dicts from `.pkl` (the reference's format), `.json` or `.jsonl`.  JSON turns the integer
    keys of the span / gold dicts into strings and tuples into lists; both are restored.  Pickles go through a restricted
    unpickler (containers, numbers, strings, numpy arrays only)."""
    if path.endswith('.pkl'):
        return _load_data_pickle(path)
    if path.endswith('.jsonl'):
        with open(path) as f:
            recs = [json.loads(line) for line in f if line.strip()]
    else:
        with open(path) as f:
            recs = json.load(f)
    for r in recs:
        for key in ('nmn_program_span_by_word', 'nmn_program_span_by_char'):
            if isinstance(r.get(key), dict):
                r[key] = {int(k): tuple(v) for k, v in r[key].items()}
        if isinstance(r.get('sg_res_by_step'), dict):
            r['sg_res_by_step'] = {int(k): _untuple(v) for k, v in r['sg_res_by_step'].items()}
    return recs


def _untuple(v):
    """JSON lists back to the tuples the gold-intermediate code tells intervals by (dataset.py:203-209)."""
    if isinstance(v, list) and v and all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v) and len(v) == 2:
        return (float(v[0]), float(v[1]))
    if isinstance(v, list) and v and all(isinstance(x, list) and len(x) == 2 and
                                         all(isinstance(y, (int, float)) for y in x) for x in v):
        return tuple((float(a), float(b)) for a, b in v)
    if isinstance(v, dict):
        return {k: _untuple(x) for k, x in v.items()}
    return v


def save_question_records(records, path):
    if path.endswith('.pkl'):
        with open(path, 'wb') as f:
            pickle.dump(records, f)
    elif path.endswith('.jsonl'):
        with open(path, 'w') as f:
            for r in records:
                f.write(json.dumps(r) + '\n')
    else:
        with open(path, 'w') as f:
            json.dump(records, f)


def filter_records(records, split, novel_comp=None, more_steps=None):
    """dataset.py:41-62: train/valid drop questions with an unmatched program phrase and give missing gold
    intermediates an empty dict; the generalisation splits filter on novel_comp / more_steps."""
    out = []
    for r in records:
        if split in ('train', 'valid'):
            if r.get('sg_res_by_step') is None:
                r['sg_res_by_step'] = {}
            if (None, None) in r['nmn_program_span_by_word'].values():
                continue
        out.append(r)
    if novel_comp is not None:
        out = [r for r in out if r['novel_comp'] == novel_comp]
    if more_steps is not None:
        out = [r for r in out if r['more_steps'] == more_steps]
    return out


# ----------------------------------------------------------------------------------------------
# clip features
# ----------------------------------------------------------------------------------------------
def load_clip_features(appearance_path, video_ids, max_video_length, motion_path=None, str2num=None, dtype='f32'):
    """{video_id: tensor [T, V]} as dataset.py:131-172 builds `self.video_feats`.
    dtype 'f32': float32, the reference's in-memory format.  dtype 'bf16': the clips are rounded ONCE to bfloat16 here, where
    they are staged (BASELINE.json configs[1]: "[T=64,2048] feats, bf16"): pack_questions / VideoNMN.forward_batch keep such
    clips in bf16 on the device and the video encoder's input projection then runs on the stored rows directly (plane GEMM,
    two MFMA products per operand pair).  Parity for bf16 storage is defined against an fp32 computation on the same rounded
    values (tests/test_gpu_planes.py, tests/test_gpu_bench_path.py)."""
    if dtype not in ('f32', 'bf16'):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    feats = _load_clip_features_f32(appearance_path, video_ids, max_video_length, motion_path, str2num)
    if dtype == 'bf16':
        feats = {k: v.to(torch.float32).to(torch.bfloat16) for k, v in feats.items()}
    return feats


def _load_clip_features_f32(appearance_path, video_ids, max_video_length, motion_path=None, str2num=None):
    wanted = set(video_ids)
    feats = {}
    if os.path.isdir(appearance_path):
        for fname in sorted(os.listdir(appearance_path)):
            vid = fname.split('.')[0]
            if vid not in wanted:
                continue
            a = np.load(os.path.join(appearance_path, fname))          # allow_pickle stays False
            a = a[0:a.shape[0]:2, :]                                    # every second frame (:139-140)
            a = a[:max_video_length]
            feats[vid] = torch.tensor(a).squeeze()
    elif os.path.isfile(appearance_path):
        h5 = _h5(appearance_path, 'ids, resnet_features')
        if str2num is None:
            raise ValueError('an h5 feature file needs the str2num mapping (strID2numID.json, dataset.py:37)')
        row = {i: n for n, i in enumerate(h5['ids'][()])}
        for vid, num in str2num.items():
            if vid in wanted:
                a = h5['resnet_features'][row[num]][:max_video_length]
                feats[vid] = torch.tensor(a).mean(dim=1)                # average the crops (:153)
    else:
        raise ValueError('appearance path not given!')
    if motion_path is not None and os.path.isfile(motion_path):        # a motion DIRECTORY is ignored, as :161-162
        h5 = _h5(motion_path, 'ids, resnext_features')
        row = {i: n for n, i in enumerate(h5['ids'][()])}
        for vid, num in (str2num or {}).items():
            if vid in wanted:
                a = torch.tensor(h5['resnext_features'][row[num]][:max_video_length])
                feats[vid] = torch.cat([feats[vid], a], dim=-1)
    return feats


def _h5(path, what):
    try:
        import h5py
    except ImportError as e:
        raise ImportError('reading %s (%s) needs h5py, which is not installed here; export the clips as a directory '
                          'of <video_id>.npy files instead' % (path, what)) from e
    return h5py.File(path, 'r')


# ----------------------------------------------------------------------------------------------
# GloVe, answer vocabulary
# ----------------------------------------------------------------------------------------------
def load_glove(path):
    """{word: float64 ndarray} from the text format (first line `count dim`) or a pickled dict (:235-246)."""
    if path.endswith('.pkl'):
        return _load_data_pickle(path)
    table = {}
    with open(path) as f:
        for n, line in enumerate(f):
            if n == 0:
                continue
            parts = line.rstrip('\n').split(' ')
            table[parts[0]] = np.array([float(x) for x in parts[1:]])
    return table


def build_answer_vocab(records):
    """dataset.py:73-85: the four fixed answers, the rest by descending frequency, `<UNK>` last."""
    counts = {}
    for r in records:
        counts[r['answer']] = counts.get(r['answer'], 0) + 1
    words = ['yes', 'no', 'before', 'after']
    fixed = set(words)
    for w, _ in sorted(counts.items(), key=lambda kv: -kv[1]):          # stable: first seen first among equals
        if w not in fixed:
            words.append(w)
    words.append('<UNK>')
    return {'word2id': {w: i for i, w in enumerate(words)}, 'id2word': {i: w for i, w in enumerate(words)}}


def load_answer_vocab(path):
    with open(path) as f:
        v = json.load(f)
    v['id2word'] = {int(k): w for k, w in v['id2word'].items()}        # json made the keys strings (:90-93)
    if len(v['id2word']) != len(v['word2id']) or [v['id2word'][i] for i in range(4)] != ['yes', 'no', 'before', 'after']:
        raise ValueError('not an AGQA answer vocabulary: %s' % path)
    return v


def rescale_interval(interval, src_length, tgt_length):
    """dataset.py:258-261 frame_interval_change_fps."""
    return (interval[0] / src_length * tgt_length, interval[1] / src_length * tgt_length)


# ----------------------------------------------------------------------------------------------
# the dataset
# ----------------------------------------------------------------------------------------------
class AGQAQuestions:
    """Indexable collection whose items are the dicts `AGQADataset.__getitem__` returns (:174-233).

    records        list of question dicts (load_question_records + filter_records)
    clips          {video_id: [T,V] tensor} (load_clip_features)
    glove          {word: vector}; unknown words get uniform [0,1) noise like the reference's np.random.rand (:253),
                   drawn from `rng` (a numpy Generator; default: seeded per word so that runs repeat)
    answer_vocab   {word2id, id2word}
    video_secs     {video_id: seconds} -- needed only for train/valid items with gold intervals
    tokenize       callable str -> [str]; default is the regex tokenizer of frontend.Normaliser (nltk is absent)
    """

    def __init__(self, records, clips, glove, answer_vocab, split='test', video_secs=None, tokenize=None, rng=None):
        self.records, self.clips, self.glove, self.answer_vocab = records, clips, glove, answer_vocab
        self.split, self.video_secs = split, video_secs or {}
        self.tokenize = tokenize or frontend.Normaliser().tokenize
        self.rng = rng
        self.dim = int(next(iter(glove.values())).size)

    def __len__(self):
        return len(self.records)

    def embed_sent(self, sent):
        words = self.tokenize(sent.lower()) if isinstance(sent, str) else [s.lower() for s in sent]
        rows = []
        for w in words:
            v = self.glove.get(w)
            if v is None:
                if self.rng is not None:
                    v = self.rng.random(self.dim)
                else:
                    seed = int.from_bytes(w.encode()[:8].ljust(8, b'\0'), 'little') & 0x7fffffff
                    v = np.random.default_rng(seed).random(self.dim)
            rows.append(v)
        return torch.tensor(np.asarray(rows), dtype=torch.float32)

    def answer_vocab_length(self):
        return len(self.answer_vocab['word2id'])

    def __getitem__(self, i):
        r = self.records[i]
        w2i = self.answer_vocab['word2id']
        clip = self.clips[r['video_id']]
        item = {'question': self.embed_sent(r['question']),
                'answer': torch.tensor(w2i.get(r['answer'], w2i.get('<UNK>'))),
                'video_features': clip,                                  # the SAME tensor for every question of the clip
                'prog_str_to_question_tokens': r['nmn_program_span_by_word'],
                'nmn_program_list': r['nmn_program'], 'nmn_program_idx': r['nmn_program_idx'],
                'qa_id': r['qa_id'], 'question_raw': r['question'], 'video_id': r['video_id']}
        if self.split == 'test':
            return item
        T = clip.size(0)
        src = self.video_secs[r['video_id']] * 3                       # gold intervals are in 3 fps frames (:200)
        gold = {}
        for key, value in (r.get('sg_res_by_step') or {}).items():
            if isinstance(value, (tuple, list)) and len(value) >= 1:
                if isinstance(value[0], float):
                    value = rescale_interval(value, src, T)
                if isinstance(value[0], tuple) and isinstance(value[0][0], float):
                    value = tuple(rescale_interval(v, src, T) for v in value)
            if isinstance(value, dict) and value:
                first = next(iter(value.values()))
                if isinstance(first, tuple) and isinstance(first[0], float):
                    value = {k: rescale_interval(v, src, T) for k, v in value.items()}
            if isinstance(value, str):                                   # class names -> (name, word embeddings) (:212-218)
                value = [(value, self.embed_sent(value))]
            elif isinstance(value, list) and len(value) and isinstance(value[0], str):
                value = [(v, self.embed_sent(v)) for v in value]
            gold[key] = value
        item['sg_program_list'] = r.get('sg_program')
        item['sg_res_by_step'] = gold
        return item


# ----------------------------------------------------------------------------------------------
# items -> executor tensors
# ----------------------------------------------------------------------------------------------
class PackedBatch:
    __slots__ = ('programs', 'spans', 'video', 'video_index', 'question', 'q_lens', 'answers', 'n_clips', 'h2d_bytes', 'video_len')


def pack_questions(items, device, share_clips=True, pin=True):
    """Question dicts -> the arguments of VideoNMN.run_programs.  Clips are staged once each (items of one clip carry
    the same feature tensor); clips of different frame counts are padded to the longest and their lengths kept in
    `video_len` (None when all agree); every tensor goes through one pinned host buffer and one asynchronous H2D copy on
    the current stream."""
    from .evaluate import clip_key
    n = len(items)
    keys, order, index = {}, [], []
    for d in items:
        k = clip_key(d) if share_clips else len(order)
        if k not in keys:
            keys[k] = len(order)
            order.append(d['video_features'])
        index.append(keys[k])
    frames = [int(c.shape[0]) for c in order]
    Tm = max(frames)
    order = [torch.as_tensor(c) for c in order]
    # clips stored as bf16 (load_clip_features(..., dtype='bf16')) stay bf16: half the H2D bytes and the plane-GEMM input path
    vdtype = torch.bfloat16 if all(c.dtype == torch.bfloat16 for c in order) else torch.float32
    video = torch.stack([torch.nn.functional.pad(c.to(vdtype), (0, 0, 0, Tm - int(c.shape[0]))) for c in order])
    qs = [torch.as_tensor(d['question'], dtype=torch.float32) for d in items]
    question = torch.cat(qs)
    answers = torch.tensor([int(d['answer']) for d in items], dtype=torch.int32) if 'answer' in items[0] else None
    cuda = torch.device(device).type == 'cuda'

    def up(t):
        if t is None:
            return None
        if cuda and pin:
            t = t.pin_memory()
        return t.to(device, non_blocking=True)

    b = PackedBatch()
    b.programs = [d['nmn_program_list'] for d in items]
    b.spans = [d['prog_str_to_question_tokens'] for d in items]
    b.h2d_bytes = video.numel() * video.element_size() + question.numel() * 4 + (answers.numel() * 4 if answers is not None else 0)
    b.video, b.question, b.answers = up(video), up(question), up(answers)
    b.video_index = index if len(order) < n else None
    b.q_lens = [int(q.shape[0]) for q in qs]
    b.n_clips = len(order)
    b.video_len = frames if len(set(frames)) > 1 else None
    return b


# ----------------------------------------------------------------------------------------------
# checkpoints
# ----------------------------------------------------------------------------------------------
def save_checkpoint(output_dir, model, config):
    """`config.json` + `pytorch_model.bin` holding the state_dict -- the layout evaluate.py:136-139 loads.
    (train_module.py:212-216 pickles the whole module object instead; a state_dict is what its own evaluate.py and
    this loader expect, and it loads without executing anything.)"""
    os.makedirs(output_dir, exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(output_dir, 'pytorch_model.bin'))
    with open(os.path.join(output_dir, 'config.json'), 'w') as f:
        json.dump(config, f)


def load_checkpoint(ckpt_dir, device=None, pretrain_modules=()):
    """VideoNMN from `config.json` + `pytorch_model.bin` (tensors only: torch.load(weights_only=True))."""
    from .module_net import VideoNMN
    with open(os.path.join(ckpt_dir, 'config.json')) as f:
        config = json.load(f)
    path = os.path.join(ckpt_dir, 'pytorch_model.bin')
    try:
        sd = torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        raise ValueError('%s is not a plain state_dict (train_module.py:214 pickles the whole module); re-save it with '
                         'torch.save(model.state_dict(), ...) in the environment that wrote it' % path) from e
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError('%s does not hold a state_dict' % path)
    model = VideoNMN(config, pretrain_modules=set(pretrain_modules))
    model.load_state_dict(sd)
    return (model.to(device) if device is not None else model), config
