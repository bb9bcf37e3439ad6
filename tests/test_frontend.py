"""CPU tests of the program front-end (stair_amd/frontend.py) against tests/golden/frontend.json, which holds the
REFERENCE's parse_program / get_childrens_and_parents / stat_module_levels / program_is_valid outputs for 28
AGQA-grammar strings (made by tests/golden/make_frontend_golden.py).  Span matching has no reference fixture (nltk
is absent from this image): its matching logic is tested with an injected normaliser, its fallback normaliser is
parity-unpinned and says so."""
import json
import os

import numpy as np
import pytest

from stair_amd import frontend as F, spec

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'frontend.json')))
CASES = sorted(GOLD['cases'])


def test_arity_tables_are_the_references():
    assert F.RAW_ARITY == GOLD['_raw_arity']
    assert F.NMN_ARITY == GOLD['_nmn_arity']
    # the executor's table (spec.ARITY) is the registered subset, module_net.py:14-38
    for name, k in spec.ARITY.items():
        assert F.NMN_ARITY[name] == k


@pytest.mark.parametrize('key', CASES)
def test_parse_program_matches_reference(key):
    case = GOLD['cases'][key]
    if 'error' in case:                       # the reference raises on this string: so does the front-end, same type
        with pytest.raises(BaseException) as ei:
            F.parse_program(case['string'])
        assert type(ei.value).__name__ == case['error']
        return
    nmn, more = F.parse_program(case['string'])
    assert nmn == case['nmn']
    assert more['idx_list'] == case['idx']
    assert more['common_list'] == case['common']
    got_map = more['existsframe_filterframe_idx_mapping']
    assert (None if got_map is None else {str(k): v for k, v in got_map.items()}) == case['mapping']
    assert F.program_is_valid(nmn) == case['valid']
    if not case['valid']:
        return
    children, parents = F.children_and_parents(nmn)
    assert children == case['children'] and parents == case['parents']
    assert F.module_levels(nmn) == case['levels']
    # every parsed program only uses modules the executor registers, so it compiles
    F.CompiledProgram(nmn, {i: (0, 1) for i in range(len(nmn))})


def test_validity_probes_match_reference():
    assert len(GOLD['validity_probes']) > 100
    for probe in GOLD['validity_probes']:
        assert F.program_is_valid(probe['program']) == probe['valid'], probe['program']


def test_fixture_covers_the_references_failure_modes():
    errors = {GOLD['cases'][k]['error'] for k in CASES if 'error' in GOLD['cases'][k]}
    assert {'IndexError', 'AssertionError'} <= errors       # truncated IterateUntil; block that changes length (program_parser.py:259-263)


class _Plain(F.Normaliser):
    """whitespace words, nothing lemmatised: isolates the matching logic"""

    def tokenize(self, text):
        return text.replace('?', ' ?').split()

    def pos_tag(self, words):
        return [(w, 'X') for w in words]

    def lemmatize(self, word, pos):
        return word


def test_span_matching_logic():
    prog = ['Exists', 'dish', 'Filter', 'Temporal', 'before', 'video', 'Localize', 'video', 'washing_a_door', 'holding', 'zebra']
    q = 'Was the person holding a dish before washing a door?'
    by_word, by_char = F.match_spans(prog, q, _Plain())
    assert sorted(by_word) == [1, 8, 9, 10]                       # module names and keywords are never looked up
    assert by_word[1] == (5, 6) and q[slice(*by_char[1])] == 'dish'
    assert by_word[8] == (7, 10) and q[slice(*by_char[8])] == 'washing a door'
    assert by_word[9] == (3, 4)
    assert by_word[10] == (None, None) and by_char[10] == (None, None)
    # agqa_lite.py:75: the scan stops one start position early -- a phrase that ENDS the token list is not found
    by_word, _ = F.match_spans(['door'], 'open the door', _Plain())
    assert by_word[0] == (None, None)
    by_word, _ = F.match_spans(['door'], 'open the door ?', _Plain())
    assert by_word[0] == (2, 3)
    assert F.match_spans(None, 'x') == (None, None)


def test_span_word_rules_and_fallback_normaliser():
    # question-side and program-side rewrite tables (agqa_lite.py:25-26) meet in the middle: 'drank' ~ 'drinking'
    by_word, _ = F.match_spans(['drinking_from_a_cup'], 'What did they touch after they drank from a cup ?', _Plain())
    assert by_word[0] == (6, 10)                                   # per-word rules: drank -> drink <- drinking
    by_word, _ = F.match_spans(['sitting_on'], 'Were they sitting on it ?', _Plain())
    assert by_word[0] == (None, None)                              # whole-phrase rule 'sitting on' -> 'sit' has no question-side twin
    # built-in normaliser (NOT nltk, parity unpinned): lemmatises both sides the same way, so templated phrases match
    by_word, by_char = F.match_spans(['holding_a_dish', 'blankets'], 'Were they holding a dish or some blanket first?')
    assert by_word[0] == (2, 5) and by_word[1] == (7, 8)


def test_nltk_normaliser_is_gated():
    try:
        import nltk  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            F.nltk_normaliser()


def test_compiled_program_cache_and_batch_packing():
    cache = F.ProgramCache(capacity=2)
    p1 = ['Exists', 'dish', 'Filter', 'video', 'objects']
    c1 = cache.get(p1, {1: (3, 4)})
    assert cache.get(list(p1), {1: (3, 4)}) is c1 and cache.hits == 1
    assert cache.get(p1, {1: (2, 4)}) is not c1                   # same program, other span: another shape
    assert c1.codes.tolist() == spec.encode_program(p1)
    assert c1.lo.tolist() == [0, 3, 0, 0, 0] and c1.hi.tolist() == [0, 4, 0, 0, 0]
    cache.get(['Exists', 'cup', 'Filter', 'video', 'objects'], {1: (0, 1)})
    assert cache.size <= 2 + 1 and len(cache._d) <= 2              # bounded: oldest template dropped past capacity
    with pytest.raises(KeyError):
        F.CompiledProgram(p1, {})                                 # module_net.py:127 fails the same way
    with pytest.raises(KeyError):
        F.CompiledProgram(p1, {1: (None, None)})                  # an unmatched phrase must not average the whole question
    c2 = F.CompiledProgram(['HasItem', 'video'], {})
    prog_off, tokens, lo, hi, q_off = F.pack_batch([c1, c2, c1], [9, 4, 7])
    assert prog_off.tolist() == [0, 5, 7, 12] and q_off.tolist() == [0, 9, 13, 20]
    assert tokens.dtype == np.int32 and tokens.tolist() == c1.codes.tolist() + c2.codes.tolist() + c1.codes.tolist()
    assert lo.tolist()[5:7] == [0, 0] and hi.tolist()[8 - 2 + 0] == 0 and hi[1] == 4 and hi[8] == 4


SPANS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'spans.json')))


@pytest.mark.parametrize('key', sorted(SPANS))
def test_match_spans_equals_reference_matcher_under_the_injected_normaliser(key):
    """tests/golden/spans.json holds the outputs of the REFERENCE's get_program_list_string_index (utils/agqa_lite.py:62-119)
    driven with frontend.Normaliser as its three nltk tools (make_spans_golden.py): rewrite tables, 'ing' override, the scan
    that stops one position early (X0), repeated words (X1), unmatched phrases (C3, X2), word and character spans."""
    case = SPANS[key]
    by_word, by_char = F.match_spans(case['nmn'], case['question'], F.Normaliser())
    assert {str(k): list(v) for k, v in by_word.items()} == case['by_word']
    assert {str(k): list(v) for k, v in by_char.items()} == case['by_char']
