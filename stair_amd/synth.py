"""Deterministic synthetic weights / questions for parity tests, golden fixtures and the bench.

Real AGQA data, GloVe vectors and checkpoints are download-only assets of the reference
(/root/reference/README.md:8-30) and are not available offline, so every test and benchmark in this
repository runs on synthetic data *of the same layout* as ``AGQADataset.__getitem__`` returns
(/root/reference/video_nmn/dataset.py:174-233).

The generator is counter based (splitmix64 of (seed, stream, index)) so that any tensor can be
re-created bit-for-bit on another machine from its name alone -- golden fixtures therefore store
only *outputs* for the full-size configuration.
"""
from __future__ import annotations

import zlib
import numpy as np

from . import spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream_id(name: str) -> int:
    return zlib.crc32(name.encode('utf-8')) & 0xFFFFFFFF


def uniform01(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (seed, name, lane, index)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.array([(seed << 34) ^ (_stream_id(name) << 2) ^ lane], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, name, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * uniform01(seed, name, n)).astype(np.float32).reshape(shape)


def normal(seed, name, shape) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name, n, lane=1)
    u2 = uniform01(seed, name, n, lane=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def randint(seed, name, n, lo, hi) -> np.ndarray:
    """n integers uniform in [lo, hi)."""
    return (lo + np.floor(uniform01(seed, name, n) * (hi - lo))).astype(np.int64)


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------
def make_weights(config, seed=0):
    """{state_dict_key: float32 ndarray} for every canonical weight (aliases share the array).

    Magnitudes follow torch's default initialisers (U(+-1/sqrt(fan_in)) for Linear/Conv/LSTM) so
    activations live in the range a trained checkpoint would produce; LayerNorm affine and
    Relate.beta are randomised too so that no term is trivially 0 or 1 in a parity test.
    """
    H = config['hidden_size']
    out = {}
    for name, shape in spec.weight_table(config):
        if name.endswith('layer_norm.weight'):
            w = 1.0 + 0.25 * uniform(seed, name, shape)
        elif name.endswith('layer_norm.bias'):
            w = 0.25 * uniform(seed, name, shape)
        elif name.endswith('Relate.beta'):
            w = uniform(seed, name, shape, 0.0, 1.0)
        elif '_encoder.' in name:
            w = uniform(seed, name, shape) / np.float32(np.sqrt(H // 2))
        else:
            if name.endswith('.bias'):
                # bound of the matching weight: look it up through its fan_in
                wshape = dict(spec.weight_table(config))[name[:-5] + '.weight']
                fan_in = int(np.prod(wshape[1:]))
            else:
                fan_in = int(np.prod(shape[1:]))
            w = uniform(seed, name, shape) / np.float32(np.sqrt(fan_in))
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    for alias, canon in spec.WEIGHT_ALIASES.items():
        out[alias] = out[canon]
    return out


# --------------------------------------------------------------------------------------------
# program corpus
# --------------------------------------------------------------------------------------------
# P0..P7: nmn program lists and idx lists exactly as produced by the reference's parse_program on
# the eight AGQA-grammar strings of SURVEY.md Appendix B (P0 is utils/scene_graphs.py:586); they are
# reference *outputs* (data), regenerated and cross-checked by tests/golden/make_golden.py.
# C0.. : hand-written prefix programs (valid per program_is_valid) that cover the operator
# branches P0-P7 never reach (ExistsFrame, XorFrame, frame-level And, Relate backward,
# Superlative min, FilterFrame 'relations', Temporal before with K=2, ...).
CORPUS = {
    'P0': (["Xor", "Exists", "food", "Filter", "Temporal", "between", "video", "Localize", "video", "Array2", "grasping_onto_a_doorknob", "drinking_from_a_cup", "holding", "Exists", "Filter", "video", "opening", "Filter", "Temporal", "between", "video", "Localize", "video", "Array2", "grasping_onto_a_doorknob", "drinking_from_a_cup", "holding"],
           [0, 1, 2, 3, 4, None, None, 5, None, 6, 7, 8, 13, 15, 19, 20, 25, 27, 28, None, None, 29, None, 30, 31, 32, 37]),
    'P1': (["Exists", "dish", "Filter", "video", "objects"], [0, 1, 2, 3, 7]),
    'P2': (["Filter", "AttnVideo", "Temporal", "after", "video", "Localize", "video", "eating_a_sandwich", "Relate", "forward", "HasItem", "FilterFrame", "video", "holding", "holding"],
           [3, None, 5, None, None, 6, None, 7, None, 4, 8, 9, 10, 15, 23]),
    'P3': (["Superlative", "max", "FilterFrame", "video", "actions", "video"], [0, 1, 2, 3, 5, None]),
    'P4': (["Equals", "Filter", "video", "holding", "Filter", "video", "touching"], [0, 4, 5, 10, 15, 16, 21]),
    'P5': (["Compare", "Exists", "eating_a_sandwich", "Filter", "Temporal", "before", "video", "Localize", "video", "opening_a_door", "actions", "Exists", "eating_a_sandwich", "Filter", "Temporal", "after", "video", "Localize", "video", "opening_a_door", "actions"],
           [0, 4, 5, 6, 7, None, None, 8, None, 9, 13, 4, 5, 6, 7, None, None, 8, None, 9, 13]),
    'P6': (["Choose", "dish", "blanket", "Filter", "Temporal", "while", "video", "Localize", "video", "holding_a_dish", "holding"],
           [0, 1, 2, 3, 4, None, None, 5, None, 6, 11]),
    'P7': (["And", "Exists", "ToAction", "holding", "dish", "Filter", "video", "actions", "Exists", "door", "Filter", "video", "objects"],
           [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 16]),
    # --- coverage programs (written for this repository) ---
    'C0': (["Filter", "AttnVideo", "video", "Relate", "backward", "And", "ExistsFrame", "cup", "FilterFrame", "video", "relations", "XorFrame", "HasItem", "FilterFrame", "video", "holding", "ExistsFrame", "dish", "video", "cup"],
           [0, 1, None, 2, None, 3, 4, 5, 6, None, 7, 8, 9, 10, None, 11, 12, 13, None, 14]),
    'C1': (["Superlative", "min", "FilterFrame", "Temporal", "before", "video", "Localize", "video", "Array2", "running", "jumping", "actions", "video"],
           [0, None, 1, 2, None, None, 3, None, 4, 5, 6, 7, None]),
    'C2': (["Xor", "Exists", "phone", "Filter", "video", "relations", "Equals", "Filter", "video", "objects", "ToAction", "holding", "phone"],
           [0, 1, 2, 3, None, 4, 5, 6, None, 7, 8, 9, 10]),
    'C3': (["Exists", "Choose", "cup", "dish", "Filter", "video", "holding", "Filter", "Temporal", "after", "video", "Localize", "video", "sitting_down", "objects"],
           [0, 1, 2, 3, 4, None, 5, 6, 7, None, None, 8, None, 9, 10]),
}
PAPER_FORMS = ['P0', 'P1', 'P2', 'P3', 'P4', 'P5', 'P6', 'P7']
ALL_FORMS = PAPER_FORMS + ['C0', 'C1', 'C2', 'C3']


def make_question(config, seed, qid, form=None, T=None, forms=PAPER_FORMS, with_video=True):
    """One synthetic question dict in the layout of dataset.py:191-233 (numpy arrays, no torch).

    Spans follow the survey harness: program position i -> (1 + i % (Q-2), 2 + i % (Q-2)); tokens
    that are modules/keywords simply never look their span up (module_net.py:126-128).
    """
    T = T or config['max_video_length']
    name = 'q%d' % qid
    if form is None:
        form = forms[int(randint(seed, name + '/form', 1, 0, len(forms))[0])]
    prog, idx = CORPUS[form]
    Q = int(randint(seed, name + '/Q', 1, 8, 26)[0])
    # span widths 1..3 so the span-mean really averages (clipped to the question length)
    widths = randint(seed, name + '/w', len(prog), 1, 4)
    spans = {}
    for i in range(len(prog)):
        s = 1 + i % (Q - 2)
        spans[i] = (s, min(Q, s + int(widths[i])))
    d = {
        'question': normal(seed, name + '/question', (Q, config['text_size'])),
        'prog_str_to_question_tokens': spans,
        'nmn_program_list': list(prog),
        'nmn_program_idx': list(idx),
        'answer': int(randint(seed, name + '/answer', 1, 0, config['answer_vocab_length'] - 1)[0]),
        'qa_id': '%s-%s' % (form, name),
        'form': form,
    }
    if with_video:
        d['video_features'] = normal(seed, name + '/video', (T, config['video_size']))
    return d


def make_questions(config, seed, n, T=None, forms=PAPER_FORMS, start=0):
    return [make_question(config, seed, start + i, T=T, forms=forms) for i in range(n)]


# --------------------------------------------------------------------------------------------
# synthetic gold intermediates (the layout of dataset.py:200-221 after rescaling to T frames)
# --------------------------------------------------------------------------------------------
N_CLASSES = 214            # size of data/AGQA/filter_answers.json in the reference


def class_embedding(config, seed, cid):
    L = 1 + cid % 3
    return normal(seed, 'class%d' % cid, (L, config['text_size']))


def make_gold(config, seed, q, T=None, keep=0.85):
    """sg_res_by_step for a question of make_question: one gold per supervised module node, dropped with
    probability 1-keep (the scene-graph executor does not always produce one, agqa_lite.py:54-57)."""
    T = T or config['max_video_length']
    prog, idx = q['nmn_program_list'], q['nmn_program_idx']
    name = q['qa_id']
    u = uniform01(seed, name + '/gold', 8 * len(prog))
    sg = {}
    for i, tok in enumerate(prog):
        if i == 0 or idx[i] is None or tok not in ARITY_FOR_GOLD:
            continue
        r = u[8 * i: 8 * i + 8]
        if r[0] > keep:
            continue

        def interval(a, b):
            lo, hi = sorted((a * T, b * T))
            return (float(lo), float(hi))
        if tok == 'Localize':
            sg[idx[i]] = (interval(r[1], r[2]), interval(r[3], r[4]))       # two intervals; a K=1 node reads only the first
        elif tok in ('Temporal', 'ExistsFrame'):
            sg[idx[i]] = interval(r[1], r[2])
        elif tok in ('Exists', 'Xor', 'Equals'):
            sg[idx[i]] = bool(r[1] > 0.5)
        else:
            cids = sorted({int(r[1] * N_CLASSES), int(r[2] * N_CLASSES)} if r[3] > 0.5 else {int(r[1] * N_CLASSES)})
            sg[idx[i]] = [('class_%d' % c, class_embedding(config, seed, c)) for c in cids]
    return sg


ARITY_FOR_GOLD = {'Localize', 'Temporal', 'ExistsFrame', 'Exists', 'Xor', 'Equals', 'Filter', 'ToAction', 'Superlative'}
