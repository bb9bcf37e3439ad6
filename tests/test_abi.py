"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, its weight table equals the reference's state_dict keys, the VideoNMN mirror has the
reference's state_dict layout, and the plan builder (host-only code) levels programs exactly like
utils/program_parser.py::stat_module_levels.  No kernel is launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from stair_amd import spec, synth
from stair_amd import _lib
from stair_amd._lib import lib, check, StairConfig, PlanInfo, StairError
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    header = open(os.path.join(ROOT, 'include', 'stair_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(stair_[a-z0-9_]+)\s*\(', header))
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.stair_abi_version() == 6


def _ctx(config):
    cfg = StairConfig(config['hidden_size'], config['video_size'], config['text_size'], config['answer_vocab_length'],
                      config['max_video_length'], config['object_types'], 1 if config['have_pretrain_head'] else 0)
    h = C.c_void_p()
    check(lib.stair_ctx_create(C.byref(cfg), C.byref(h)))
    return h


CONFIGS = [
    dict(spec.DEFAULT_CONFIG),
    dict(spec.DEFAULT_CONFIG, have_pretrain_head=False),
    dict(spec.DEFAULT_CONFIG, max_video_length=8, video_size=4096),
    dict(spec.DEFAULT_CONFIG, max_video_length=150),          # args.py:29 default -> k = round(37.5) = 38
    dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10),
]


@pytest.mark.parametrize('config', CONFIGS)
def test_weight_table_matches_spec(config):
    h = _ctx(config)
    try:
        table = spec.weight_table(config)
        assert lib.stair_weight_count(h) == len(table)
        for i, (name, shape) in enumerate(table):
            assert lib.stair_weight_name(h, i).decode() == name
            assert lib.stair_weight_numel(h, i) == int(np.prod(shape)), name
    finally:
        lib.stair_ctx_destroy(h)


def test_bad_config_is_rejected():
    cfg = StairConfig(100, 2048, 300, 172, 64, 36, 1)
    h = C.c_void_p()
    assert lib.stair_ctx_create(C.byref(cfg), C.byref(h)) != 0
    assert b'hidden_size' in lib.stair_last_error()


def test_context_options_are_per_context():
    """stair_ctx_set_option / stair_ctx_get_option: overrides live in the context (host state only, no GPU needed), -1 = inherit."""
    cfg = StairConfig(512, 2048, 300, 172, 64, 36, 1)
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.stair_ctx_create(C.byref(cfg), C.byref(a)) == 0 and lib.stair_ctx_create(C.byref(cfg), C.byref(b)) == 0
    v = C.c_int32(7)
    for opt in range(5):
        assert lib.stair_ctx_get_option(a, opt, C.byref(v)) == 0 and v.value == -1
    assert lib.stair_ctx_set_option(a, 0, 0) == 0 and lib.stair_ctx_set_option(a, 1, 0) == 0
    assert lib.stair_ctx_get_option(a, 0, C.byref(v)) == 0 and v.value == 0
    assert lib.stair_ctx_get_option(b, 0, C.byref(v)) == 0 and v.value == -1            # the other context is untouched
    assert lib.stair_ctx_set_option(a, 0, -5) == 0 and lib.stair_ctx_get_option(a, 0, C.byref(v)) == 0 and v.value == -1
    assert lib.stair_ctx_set_option(a, 0, 9) != 0 and b'matmul mode' in lib.stair_last_error()
    assert lib.stair_ctx_set_option(a, 5, 1) != 0 and b'unknown option' in lib.stair_last_error()
    lib.stair_ctx_destroy(a); lib.stair_ctx_destroy(b)


@pytest.mark.parametrize('heads', [True, False])
def test_videonmn_state_dict_layout(heads):
    from stair_amd.module_net import VideoNMN
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40,
                  object_types=10, have_pretrain_head=heads)
    m = VideoNMN(config)
    sd = m.state_dict()
    assert list(sd.keys()) == spec.state_dict_keys(config)
    shapes = dict(spec.weight_table(config))
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[spec.WEIGHT_ALIASES.get(k, k)]), k
    # Superlative.localize_module IS Localize (module_net.py:31-32)
    assert sd['submodules.Superlative.localize_module.video_linear.0.weight'].data_ptr() == \
        sd['submodules.Localize.video_linear.0.weight'].data_ptr()
    # loading a reference-layout checkpoint works
    w = synth.make_weights(config, 3)
    import torch
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    assert np.array_equal(m.state_dict()['submodules.decoder.3.bias'].numpy(), w['submodules.decoder.3.bias'])


def _build(config, programs, spans, q_lens, T):
    h = _ctx(config)
    enc = [np.asarray(spec.encode_program(p), dtype=np.int32) for p in programs]
    n = len(programs)
    prog_off = np.zeros(n + 1, np.int32); np.cumsum([len(e) for e in enc], out=prog_off[1:])
    tokens = np.concatenate(enc)
    lo = np.zeros(len(tokens), np.int32); hi = np.zeros(len(tokens), np.int32)
    for q in range(n):
        for i, c in enumerate(enc[q]):
            if c == spec.TOK_SPAN:
                lo[prog_off[q] + i], hi[prog_off[q] + i] = spans[q][i]
    q_off = np.zeros(n + 1, np.int32); np.cumsum(q_lens, out=q_off[1:])
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    plan = C.c_void_p()
    rc = lib.stair_plan_build(h, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), T, 0, C.byref(plan))
    return h, plan, rc, prog_off


def test_plan_levels_match_reference_stat_module_levels():
    progs = json.load(open(os.path.join(GOLDEN, 'programs.json')))
    progs.pop('_nary_mappings')
    config = dict(spec.DEFAULT_CONFIG)
    names = sorted(progs)
    qs = [synth.make_question(config, 0, i, form=name, with_video=False) for i, name in enumerate(names)]
    h, plan, rc, prog_off = _build(config, [q['nmn_program_list'] for q in qs],
                                   [q['prog_str_to_question_tokens'] for q in qs], [q['question'].shape[0] for q in qs], 64)
    assert rc == 0, lib.stair_last_error()
    try:
        for qi, name in enumerate(names):
            lv = []
            for i in range(len(progs[name]['nmn'])):
                k, s, a, l, r = (C.c_int32() for _ in range(5))
                check(lib.stair_plan_node(plan, int(prog_off[qi] + i), C.byref(k), C.byref(s), C.byref(a), C.byref(l), C.byref(r)))
                lv.append(l.value)
            assert lv == progs[name]['levels'], name
        info = PlanInfo()
        check(lib.stair_plan_get_info(plan, C.byref(info)))
        assert info.n_questions == len(names) and info.n_levels == 1 + max(max(p['levels']) for p in progs.values())
        assert info.workspace_bytes > 0 and info.vec_off % config['hidden_size'] == 0 and info.map_off % config['hidden_size'] == 0
    finally:
        lib.stair_plan_destroy(plan)
        lib.stair_ctx_destroy(h)


@pytest.mark.parametrize('program,msg', [
    (['Filter', 'video'], 'stack underflow'),                              # invalid: missing operand
    (['video', 'video'], 'stack holds 2'),                                 # assert len(stack)==1, module_net.py:135
    (['Temporal', 'while', 'video', 'Localize', 'video', 'x'], 'root must produce'),   # root is a [T,H] map
    (['Filter', 'dish', 'objects'], 'kind mismatch'),                      # feat is not a map
    (['Exists', 'dish', 'Filter', 'video', 'max'], 'Filter keyword'),      # KeyError 'max' in the reference
])
def test_invalid_programs_are_rejected(program, msg):
    config = dict(spec.DEFAULT_CONFIG)
    spans = {i: (1, 2) for i in range(len(program))}
    h, plan, rc, _ = _build(config, [program], [spans], [8], 64)
    try:
        assert rc != 0
        assert msg in lib.stair_last_error().decode(), lib.stair_last_error()
    finally:
        lib.stair_ctx_destroy(h)


def test_linear_temporal_requires_full_length():
    config = dict(spec.DEFAULT_CONFIG, max_video_length=8)
    h, plan, rc, _ = _build(config, [['Filter', 'video', 'objects']], [{}], [8], 6)
    assert rc != 0 and b'max_video_length' in lib.stair_last_error()
    lib.stair_ctx_destroy(h)


@pytest.mark.parametrize('train', [False, True])
def test_workspace_regions_are_disjoint(train):
    """Every region of the plan's workspace layout (arenas, per-bucket saves, gradient arenas, scratch) must be
    disjoint from every other one, for all 12 program forms in one batch."""
    config = dict(spec.DEFAULT_CONFIG)
    names = sorted(synth.CORPUS)
    qs = [synth.make_question(config, 0, i, form=name, with_video=False) for i, name in enumerate(names)]
    h = _ctx(config)
    enc = [np.asarray(spec.encode_program(q['nmn_program_list']), dtype=np.int32) for q in qs]
    n = len(qs)
    prog_off = np.zeros(n + 1, np.int32); np.cumsum([len(e) for e in enc], out=prog_off[1:])
    tokens = np.concatenate(enc)
    lo = np.zeros(len(tokens), np.int32); hi = np.zeros(len(tokens), np.int32)
    for q in range(n):
        for i, c in enumerate(enc[q]):
            if c == spec.TOK_SPAN:
                lo[prog_off[q] + i], hi[prog_off[q] + i] = qs[q]['prog_str_to_question_tokens'][i]
    q_off = np.zeros(n + 1, np.int32); np.cumsum([q['question'].shape[0] for q in qs], out=q_off[1:])
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    plan = C.c_void_p()
    check(lib.stair_plan_build(h, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), 64, 1 if train else 0, C.byref(plan)))
    cap = 8192
    nm = (C.c_char_p * cap)(); beg = (C.c_int64 * cap)(); end = (C.c_int64 * cap)()
    k = lib.stair_plan_regions(plan, h, nm, beg, end, cap)
    assert 0 < k < cap
    regs = sorted((beg[i], end[i], nm[i].decode()) for i in range(k))
    for a, b in zip(regs, regs[1:]):
        assert a[1] <= b[0], ('overlap', a, b)
    info = PlanInfo()
    check(lib.stair_plan_get_info(plan, C.byref(info)))
    assert regs[-1][2] == 'END' and regs[-1][0] * 4 == info.workspace_bytes
    lib.stair_plan_destroy(plan)
    lib.stair_ctx_destroy(h)


@pytest.mark.parametrize('train', [False, True])
def test_external_projection_plan_leaves_the_projection_regions_out(train):
    """STAIR_PLAN_EXT_PROJECTION (ABI 6): the encoders' input-projection regions move to a caller-owned buffer whose size depends on the
    batch shape alone (stair_projection_floats) -- the workspace shrinks by exactly that much, the remaining regions stay disjoint, and a
    buffer that is too small, or a plan built without the flag, is refused."""
    config = dict(spec.DEFAULT_CONFIG)
    qs = [synth.make_question(config, 0, i, form=name, with_video=False) for i, name in enumerate(sorted(synth.CORPUS))]
    programs = [q['nmn_program_list'] for q in qs]
    h = _ctx(config)
    enc = [np.asarray(spec.encode_program(p), dtype=np.int32) for p in programs]
    n = len(qs)
    prog_off = np.zeros(n + 1, np.int32); np.cumsum([len(e) for e in enc], out=prog_off[1:])
    tokens = np.concatenate(enc)
    lo = np.zeros(len(tokens), np.int32); hi = np.zeros(len(tokens), np.int32)
    for q in range(n):
        for i, c in enumerate(enc[q]):
            if c == spec.TOK_SPAN:
                lo[prog_off[q] + i], hi[prog_off[q] + i] = qs[q]['prog_str_to_question_tokens'][i]
    q_off = np.zeros(n + 1, np.int32); np.cumsum([q['question'].shape[0] for q in qs], out=q_off[1:])
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    sizes = {}
    for ext in (0, 4):
        plan = C.c_void_p()
        check(lib.stair_plan_build(h, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), 64, (1 if train else 0) | ext, C.byref(plan)))
        info = PlanInfo()
        check(lib.stair_plan_get_info(plan, C.byref(info)))
        sizes[ext] = info.workspace_bytes
        cap = 8192
        nm = (C.c_char_p * cap)(); beg = (C.c_int64 * cap)(); end = (C.c_int64 * cap)()
        k = lib.stair_plan_regions(plan, h, nm, beg, end, cap)
        regs = sorted((beg[i], end[i], nm[i].decode()) for i in range(k))
        for a, b in zip(regs, regs[1:]):
            assert a[1] <= b[0], ('overlap', a, b)
        assert ('xpv' in {r[2] for r in regs}) == (ext == 0)
        nf = lib.stair_projection_floats(h, n, 64, int(q_off[-1]))
        assert nf > n * 64 * 4 * config['hidden_size']
        dummy = C.c_void_p(256)               # an aligned non-null address: the setter only records it (nothing is launched here)
        if ext:
            with pytest.raises(StairError, match='too small'):
                check(lib.stair_plan_set_projection(h, plan, dummy, nf - 1))
            check(lib.stair_plan_set_projection(h, plan, dummy, nf))
        else:
            with pytest.raises(StairError, match='EXT_PROJECTION'):
                check(lib.stair_plan_set_projection(h, plan, dummy, nf))
        lib.stair_plan_destroy(plan)
    saved = sizes[0] - sizes[4] - 4 * lib.stair_projection_floats(h, n, 64, int(q_off[-1]))
    assert 0 <= saved < 16384            # (alignment of the regions laid out behind them)
    assert lib.stair_projection_floats(h, -1, 64, 10) == -1
    lib.stair_ctx_destroy(h)


def test_common_subexpressions_are_aliased_not_recomputed():
    """STAIR_PLAN_NO_CSE off (default): nodes that depend only on the clip / keyword strings / identical spans alias the first
    occurrence -- across questions about one clip and inside one program (P0 evaluates the same Localize / Temporal / Filter
    chain twice); with the flag every node gets a slot of its own, as module_net.py:100-106 computes it."""
    from stair_amd._lib import PlanInfo
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    h = _ctx(config)
    forms = ['P1', 'P1', 'P3', 'P3', 'P0', 'P7']
    progs = [synth.CORPUS[f][0] for f in forms]
    enc = [np.asarray(spec.encode_program(p), dtype=np.int32) for p in progs]
    n = len(progs)
    prog_off = np.zeros(n + 1, np.int32); np.cumsum([len(e) for e in enc], out=prog_off[1:])
    tokens = np.concatenate(enc)
    # every span token of a program gets the span (1 + its phrase id): equal phrases -> equal spans
    lo = np.zeros(len(tokens), np.int32); hi = np.zeros(len(tokens), np.int32)
    for q in range(n):
        phrase = {}
        for i, c in enumerate(enc[q]):
            if c == spec.TOK_SPAN:
                k = phrase.setdefault(progs[q][i], len(phrase))
                lo[prog_off[q] + i], hi[prog_off[q] + i] = 1 + k, 2 + k
    q_off = np.zeros(n + 1, np.int32); np.cumsum([9] * n, out=q_off[1:])
    clip = np.asarray([0, 0, 1, 1, 2, 0], dtype=np.int32)       # questions 0, 1, 5 ask about clip 0; 2, 3 about clip 1
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    info = {}
    slots = {}
    for flag in (0, 2):
        plan = C.c_void_p()
        check(lib.stair_plan_build_shared(h, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), 3, ip(clip), 40, flag, C.byref(plan)))
        inf = PlanInfo()
        check(lib.stair_plan_get_info(plan, C.byref(inf)))
        if flag == 2:
            # dropout on a TRAINING plan: every occurrence draws its own mask in the reference (module_net.py:100-106 under
            # model.train()), so a plan that shares nodes refuses p > 0 and an unshared one takes it
            for fl, ok in ((1, False), (1 | 2, True)):
                tplan = C.c_void_p()
                check(lib.stair_plan_build_shared(h, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), 3, ip(clip), 40, fl, C.byref(tplan)))
                rc = lib.stair_plan_set_dropout(tplan, C.c_float(0.25), C.c_uint64(7))
                assert (rc == 0) == ok, (fl, rc)
                if not ok:
                    assert b'STAIR_PLAN_NO_CSE' in lib.stair_last_error()
                    assert lib.stair_plan_set_dropout(tplan, C.c_float(0.0), C.c_uint64(0)) == 0
                lib.stair_plan_destroy(tplan)
        info[flag] = inf
        tab = [np.empty(inf.n_nodes, np.int32) for _ in range(5)]
        check(lib.stair_plan_nodes(plan, *[ip(t) for t in tab], inf.n_nodes))
        slots[flag] = tab
        lib.stair_plan_destroy(plan)
    assert info[2].n_aliased == 0 and info[0].n_aliased > 10
    assert info[0].n_map < info[2].n_map and info[0].n_vec < info[2].n_vec and info[0].workspace_bytes < info[2].workspace_bytes
    kind, slot = slots[0][0], slots[0][1]
    # question 1 is question 0 again on the same clip, but 'dish' is its own words: Filter(video, objects) is shared, Exists is not
    p0, p1 = prog_off[0], prog_off[1]
    assert slot[p0 + 2] == slot[p1 + 2] and slot[p0 + 0] != slot[p1 + 0]
    # P7 on clip 0 reuses that same Filter(video, objects) node (its token 10)
    p5 = prog_off[5]
    assert progs[5][10] == 'Filter' and slot[p5 + 10] == slot[p0 + 2]
    # P3 = Superlative(max, FilterFrame(video, actions), video): the whole program depends on the clip only
    p2, p3 = prog_off[2], prog_off[3]
    assert slot[p2 + 0] == slot[p3 + 0]
    # inside P0 the two identical Filter(Temporal(between, video, Localize(video, Array2(..))), holding) chains are one
    p4 = prog_off[4]
    assert progs[4][3] == 'Filter' and progs[4][17] == 'Filter' and slot[p4 + 3] == slot[p4 + 17]
    assert np.array_equal(slots[0][3], slots[2][3])              # levels do not change
    # without sharing every module node owns its slot
    k2, s2 = slots[2][0], slots[2][1]
    mods = [(k2[i], s2[i]) for q in range(n) for i in range(prog_off[q], prog_off[q + 1]) if progs[q][i - prog_off[q]] in spec.ARITY and progs[q][i - prog_off[q]] != 'Array2']
    assert len(set(mods)) == len(mods)
