"""Program front-end: AGQA program strings -> the executor's prefix programs, question-word spans, and
compiled (packed int32) programs cached per distinct shape.

Mirrors, as host code in the reference's own language (Python):
  * /root/reference/utils/program_parser.py:28-170   parse_program (+ 173-201 tree helpers, 226-268 the IterateUntil
    rewrite, 307-333 levels / validity), pinned by tests/golden/frontend.json (the reference's own outputs);
  * /root/reference/utils/agqa_lite.py:62-119        get_program_list_string_index (span matching).  Its word
    normalisation needs nltk (tokenizer, POS tagger, WordNet), which this image does not have: the matching logic is
    implemented and tested here with the normaliser injected; with the built-in fallback normaliser the spans are
    "parity unpinned" (no reference output exists to compare with).

The reference rewrites the flat token list in place; this implementation scans the tokens once into a new list
(`_Scan`) and expresses the block rewrites on an explicit tree (`_tree`), but it keeps the reference's observable
behaviour, including the parts that look accidental:
  * `Localize(mode, x)` becomes `Temporal mode video Localize video x` where the NEW Localize takes the `mode`
    token's index and `mode` itself loses its index (program_parser.py:81-91);
  * the element kept from `[actions, x]` / the `mode` token are emitted as they are (no further rewriting);
  * `existsframe_filterframe_idx_mapping` is that of the LAST IterateUntil block rewritten (blocks go smallest first).
"""
from __future__ import annotations

import re
from collections import OrderedDict

import numpy as np

from . import spec

# arities while parsing the raw AGQA grammar (program_parser.py:9-15) ...
RAW_ARITY = {'Array1': 1, 'HasItem': 1, 'OnlyItem': 1,
             'Array2': 2, 'AND': 2, 'XOR': 2, 'And': 2, 'Xor': 2, 'Compare': 2, 'Equals': 2, 'Exists': 2, 'Filter': 2,
             'Iterate': 2, 'Localize': 2, 'ToAction': 2, 'Query': 2, 'Subtract': 2,
             'Array3': 3, 'Superlative': 3, 'Choose': 3, 'IterateUntil': 4}
# ... and of the rewritten (executor) vocabulary (program_parser.py:17-25): Query drops to 1, Subtract disappears,
# the frame-level modules and Temporal appear.
NMN_ARITY = {'Array1': 1, 'HasItem': 1, 'OnlyItem': 1, 'Query': 1,
             'Array2': 2, 'AND': 2, 'XOR': 2, 'And': 2, 'Xor': 2, 'Compare': 2, 'Equals': 2, 'Exists': 2, 'Filter': 2,
             'Iterate': 2, 'Localize': 2, 'ToAction': 2, 'Relate': 2, 'AttnVideo': 2, 'FilterFrame': 2,
             'ExistsFrame': 2, 'XorFrame': 2,
             'Array3': 3, 'Superlative': 3, 'Choose': 3, 'Temporal': 3, 'IterateUntil': 4}
KEYWORD_NAMES = {'forward', 'backward', 'while', 'temporal_tag', 'between', 'before', 'after', 'max', 'min', 'start',
                 'end', 'video', 'frame', 'relations', 'objects', 'class', 'actions'}
# words that are never looked up in the question (agqa_lite.py:22-23)
WORDS_TO_KEEP = {'forward', 'backward', 'while', 'between', 'before', 'after', 'max', 'min', 'start', 'end', 'video',
                 'relations', 'objects', 'actions'}
NOT_A_SPAN = WORDS_TO_KEEP | set(NMN_ARITY)


# ----------------------------------------------------------------------------------------------
# tree helpers over a prefix token list
# ----------------------------------------------------------------------------------------------
def children_and_parents(program_list, arity=NMN_ARITY):
    """children[i] = operand positions of token i in pop order (first positional argument first), parents[c] = i
    (0 for the root and for tokens nobody consumed) -- program_parser.py:182-201."""
    n = len(program_list)
    children = [[] for _ in range(n)]
    parents = [0] * n
    pending = []
    for i in range(n - 1, -1, -1):
        k = arity.get(program_list[i])
        if k is not None:
            if len(pending) < k:
                raise IndexError('program underflows at token %d (%s): %r' % (i, program_list[i], program_list))
            for _ in range(k):
                c = pending.pop()
                children[i].append(c)
                parents[c] = i
        pending.append(i)
    return children, parents


def subtree(children, i):
    """Sorted positions of token i and everything below it."""
    out, todo = [], [i]
    while todo:
        j = todo.pop()
        out.append(j)
        todo.extend(children[j])
    return sorted(out)


def module_levels(program_list, arity=NMN_ARITY):
    """Level of every token: operands 0, a module 1 + max over its operands (program_parser.py:307-321)."""
    levels = [0] * len(program_list)
    pending = []
    for i in range(len(program_list) - 1, -1, -1):
        k = arity.get(program_list[i])
        if k is not None:
            args = pending[len(pending) - k:] if k else []
            del pending[len(pending) - k:]
            levels[i] = 1 + max(levels[j] for j in args)
        pending.append(i)
    return levels


def program_is_valid(program_list, arity=NMN_ARITY):
    """Scanning from the last token, the value stack never underflows and ends with exactly one value
    (program_parser.py:324-333)."""
    depth = 0
    for tok in reversed(program_list):
        depth += 1 - arity.get(tok, 0)
        if depth < 0:
            return False
    return depth == 1


# ----------------------------------------------------------------------------------------------
# parse_program
# ----------------------------------------------------------------------------------------------
_SPLIT = re.compile(r';')


def _flatten(string):
    """'Exists(dish, Iterate(video, Filter(frame, [objects])))' -> prefix tokens with ArrayN heads."""
    s = string.replace(', ', ';').replace(' ', '_').replace('(', ';').replace(')', '')
    s = s.replace('[', '[;').replace(']', ';]')
    toks = _SPLIT.split(s)
    out, open_at = [], []
    for t in toks:
        if t == '[':
            open_at.append(len(out))
            out.append(t)
        elif t == ']':
            lo = open_at.pop()
            inside = out[lo + 1:]
            # elements at the top of the bracket = tokens inside minus what the modules inside consume
            n_elem = len(inside) - sum(RAW_ARITY.get(x, 0) for x in inside)
            out[lo] = 'Array%d' % n_elem
        else:
            out.append(t)
    return out


class _Scan:
    """One forward pass over (token, idx) pairs producing the rewritten pairs."""

    def __init__(self, tokens):
        self.src = [[t, i] for i, t in enumerate(tokens)]
        self.out = []
        self.iterates = []          # output positions of Iterate tokens

    def run(self):
        src, out = self.src, self.out
        p = 0
        while p < len(src):
            tok, idx = src[p]
            nxt = src[p + 1][0] if p + 1 < len(src) else None
            if tok in ('OnlyItem', 'Array1'):                       # OnlyItem(x) / [x]  ->  x
                p += 1
            elif tok == 'XOR' or tok == 'AND':
                out.append([tok.capitalize(), idx])
                p += 1
            elif tok == 'Query' and nxt == 'class':                 # Query(class, x)  ->  x
                p += 2
            elif tok == 'relation':
                out.append(['relations', idx])
                p += 1
            elif tok == 'Subtract':                                 # Subtract(Query(end, action), Query(start, action)) -> video
                out.append(['video', None])
                p += 7
            elif tok == 'Iterate':
                self.iterates.append(len(out))
                out.append([tok, idx])
                p += 1
            elif tok == 'Localize':                                 # Localize(mode, x) -> Temporal(mode, video, Localize(video, x))
                mode_tok, mode_idx = src[p + 1]
                out.extend([['Temporal', idx], [mode_tok, None], ['video', None], ['Localize', mode_idx], ['video', None]])
                p += 2
            elif tok == 'Array3':                                   # [relations, x, objects] -> x (x is rewritten in turn)
                keep = src[p + 2]
                del src[p:p + 4]
                src.insert(p, keep)
            elif tok == 'Array2' and nxt == 'actions':              # [actions, x] -> x, taken as it is
                out.append(src[p + 2])
                p += 3
            else:
                if tok == 'Superlative' and src[p + 2][0] == 'Filter':
                    src[p + 2][0] = 'FilterFrame'
                out.append([tok, idx])
                p += 1
        return out


def _drop_iterates(pairs, positions):
    """Iterate(x, Filter(frame, kw)) -> Filter(x, kw): the inner Filter and its first operand go away."""
    children, _ = children_and_parents([t for t, _ in pairs])
    gone = set()
    for pos in positions:
        pairs[pos][0] = 'Filter'
        inner = children[pos][1]
        gone.update((inner, inner + 1))
    return [pr for i, pr in enumerate(pairs) if i not in gone]


def _rewrite_iterate_until(pairs, lo, hi):
    """IterateUntil(direction, x, cond, Filter(frame, kw)) occupying pairs[lo:hi]  ->
    Filter(AttnVideo(x, Relate(direction, cond')), kw) with the condition moved to frame level
    (program_parser.py:226-268).  Returns (new pairs, {ExistsFrame idx: FilterFrame idx})."""
    names = [t for t, _ in pairs]
    children, parents = children_and_parents(names)
    x_tokens = subtree(children, children[lo][1])
    seg = [['Filter', pairs[lo][1]], ['AttnVideo', None]]
    seg.extend(pairs[lo + 2: lo + 2 + len(x_tokens)])
    seg.extend([['Relate', None], pairs[lo + 1]])
    mapping = {}
    for j in subtree(children, children[lo][2]):
        tok, idx = pairs[j]
        if tok == 'frame':
            seg.append(['video', idx])
        elif tok == 'Filter' and pairs[j + 1][0] == 'frame':
            up = parents[j]
            if pairs[up][0] == 'Exists':
                seg[up - j][0] = 'ExistsFrame'          # the parent sits (j - up) places before the end of seg
            seg.append(['FilterFrame', idx])
            mapping[pairs[up][1]] = idx
        elif tok == 'Xor':
            seg.append(['XorFrame', idx])
        else:
            seg.append(pairs[j])
    last_filter = children[lo][3]
    seg.extend(pairs[j] for j in subtree(children, children[last_filter][1]))
    if len(seg) != hi - lo:
        raise AssertionError('IterateUntil block of %d tokens rewrote to %d: %r -> %r'
                             % (hi - lo, len(seg), pairs[lo:hi], seg))
    return pairs[:lo] + seg + pairs[hi:], mapping


def parse_program(string):
    """AGQA program string -> (nmn_program_list, more_data) as program_parser.py:28-170.
    more_data: idx_list (position of each token in common_list, None for inserted tokens),
    existsframe_filterframe_idx_mapping (or None), common_list (the flattened input)."""
    common = _flatten(string)
    scan = _Scan(common)
    pairs = scan.run()
    if scan.iterates:
        pairs = _drop_iterates(pairs, scan.iterates)

    mapping = None
    if any(t == 'IterateUntil' for t, _ in pairs):
        children, _ = children_and_parents([t for t, _ in pairs])
        blocks = []
        for i, (t, _) in enumerate(pairs):
            if t == 'IterateUntil':
                span = subtree(children, i)
                blocks.append((span[0], span[-1] + 1))
        blocks.sort(key=lambda b: b[1] - b[0])             # inner blocks first; a rewrite keeps every length
        for lo, hi in blocks:
            pairs, mapping = _rewrite_iterate_until(pairs, lo, hi)

    if pairs[0][0] == 'Compare':                           # Compare([before, after], f(temporal_tag)) -> Compare f(before) f(after)
        del pairs[1:4]
        body = pairs[1:]
        tag = [t for t, _ in pairs].index('temporal_tag')
        first = [list(pr) for pr in pairs]
        second = [list(pr) for pr in body]
        first[tag][0] = 'before'
        second[tag - 1][0] = 'after'
        pairs = first + second

    more = {'idx_list': [i for _, i in pairs], 'existsframe_filterframe_idx_mapping': mapping, 'common_list': common}
    return [t for t, _ in pairs], more


# ----------------------------------------------------------------------------------------------
# span matching (agqa_lite.py:62-119)
# ----------------------------------------------------------------------------------------------
QUESTION_WORD_RULES = {'consume': 'eat', 'consuming': 'eat', 'ate': 'eat', 'taking': 'take', 'sneezing': 'sneeze',
                       'drank': 'drink', 'wiping': 'wipe', 'drinking': 'drink', 'closing': 'close', 'lay': 'lie'}
PROGRAM_WORD_RULES = {'opening': 'open', 'closing': 'close', 'sitting on': 'sit', 'playing on': 'play',
                      'drinking': 'drink', 'putting down': 'put', 'consuming': 'eat'}


class Normaliser:
    """The three language tools the matcher needs.  `nltk_normaliser()` gives the reference's (word_tokenize,
    pos_tag, WordNetLemmatizer) when nltk and its data are installed; the default here is a plain regex tokenizer
    with a suffix-stripping lemmatiser, good enough for AGQA's templated questions but NOT the reference's tools."""

    _tok = re.compile(r"[A-Za-z0-9_]+(?:'[a-z]+)?|[^\sA-Za-z0-9_]")

    def tokenize(self, text):
        return self._tok.findall(text)

    def pos_tag(self, words):
        return [(w, 'V' if w.endswith('ing') or w.endswith('ed') else 'N') for w in words]

    def lemmatize(self, word, pos):
        for suf, rep in (('ies', 'y'), ('sses', 'ss'), ('ing', ''), ('ed', ''), ('s', '')):
            if word.endswith(suf) and len(word) - len(suf) >= 3 and not word.endswith('ss'):
                return word[:len(word) - len(suf)] + rep
        return word


def nltk_normaliser():
    import nltk                                            # raises ImportError where nltk is absent (this image)
    from nltk.stem import WordNetLemmatizer
    wnl = WordNetLemmatizer()

    class _N(Normaliser):
        def tokenize(self, text):
            return nltk.tokenize.word_tokenize(text)

        def pos_tag(self, words):
            return nltk.pos_tag(words)

        def lemmatize(self, word, pos):
            return wnl.lemmatize(word, pos)
    return _N()


def _find(big, small):
    """First start of `small` inside `big`.  As in agqa_lite.py:75-78 the scan stops one position early, so a phrase
    that ends the question is NOT found (the last question token is normally '?', which hides this)."""
    for s in range(len(big) - len(small)):
        if big[s:s + len(small)] == small:
            return s
    return None


def match_spans(program_list, question, normaliser=None):
    """{token position: (first word, one past last word)} and the same in characters for every program token that
    is not a module name or keyword; (None, None) where the phrase is not found."""
    if program_list is None:
        return None, None
    nz = normaliser or Normaliser()
    words = nz.tokenize(question)
    chars, at = [], 0
    for w in words:
        at = question.index(w, at)             # the search restarts at the START of the previous word, as the reference's does
        chars.append((at, at + len(w)))
    words = [QUESTION_WORD_RULES.get(w, w) for w in words]
    tagged = [(w, 'V') if w.endswith('ing') else (w, p) for w, p in nz.pos_tag(words)]
    words = [nz.lemmatize(w, p[0].lower()) if p[0].lower() in ('v', 'n') and w != 'clothes' else w for w, p in tagged]

    by_word, by_char = {}, {}
    for i, tok in enumerate(program_list):
        if tok in NOT_A_SPAN:
            continue
        phrase = tok.replace('_', ' ')
        phrase = PROGRAM_WORD_RULES.get(phrase, phrase)
        pw = [PROGRAM_WORD_RULES.get(w, w) for w in nz.tokenize(phrase)]
        pw = [nz.lemmatize(w, p[0].lower()) if p[0] in ('V', 'N') else w for w, p in nz.pos_tag(pw)]
        s = _find(words, pw)
        if s is None:
            by_word[i] = by_char[i] = (None, None)
        else:
            by_word[i] = (s, s + len(pw))
            by_char[i] = (chars[s][0], chars[s + len(pw) - 1][1])
    return by_word, by_char


# ----------------------------------------------------------------------------------------------
# compiled programs: what stair_plan_build consumes, cached per distinct (program, spans) shape
# ----------------------------------------------------------------------------------------------
class CompiledProgram:
    """int32 token codes (enum stair_token) and the [lo, hi) question-word span of every span token."""
    __slots__ = ('codes', 'lo', 'hi', 'n_tokens', 'packed')

    def __init__(self, program_list, spans, _shape=None):
        codes, pos = _shape if _shape is not None else _encode(program_list)
        self.n_tokens = int(codes.shape[0])
        self.packed = np.zeros((3, self.n_tokens), dtype=np.int32)     # rows: codes, lo, hi -- one concatenation per batch (pack_batch)
        self.packed[0] = codes
        self.codes, self.lo, self.hi = self.packed[0], self.packed[1], self.packed[2]
        for i in pos:
            s, e = _span(spans, i)
            self.lo[i], self.hi[i] = s, e


def _encode(program_list):
    codes = np.asarray(spec.encode_program(program_list), dtype=np.int32)
    return codes, [int(i) for i in np.nonzero(codes == spec.TOK_SPAN)[0]]


def _span(spans, i):
    try:
        s, e = spans[i]
    except KeyError:
        raise KeyError(i)                  # the reference fails the same way, module_net.py:127
    if s is None or e is None:
        raise KeyError(i)                  # an unmatched phrase: tensor[None:None] would average the whole question
    return s, e


class ProgramCache:
    """Two-level cache: program tokens -> (codes, span positions) -> spans used -> CompiledProgram.  AGQA has a few
    hundred program templates but millions of questions, and every question repeats each epoch, so the per-question
    Python work of packing (string -> enum, dict lookups) is paid once.  Bounded: past `capacity` compiled programs
    the oldest template's entries are dropped."""

    def __init__(self, capacity=1 << 20):
        self.capacity = capacity
        self._d = OrderedDict()             # tuple(program) -> [shape, {span tuple: CompiledProgram}]
        self.size = self.hits = self.misses = 0

    def get(self, program_list, spans):
        key = tuple(program_list)
        ent = self._d.get(key)
        if ent is None:
            ent = self._d[key] = [_encode(program_list), {}]
        pos = ent[0][1]
        try:
            sub = tuple([spans[i] for i in pos])
        except KeyError as e:
            raise KeyError(e.args[0])
        hit = ent[1].get(sub)
        if hit is not None:
            self.hits += 1
            return hit
        self.misses += 1
        cp = ent[1][sub] = CompiledProgram(program_list, spans, ent[0])
        self.size += 1
        while self.size > self.capacity and len(self._d) > 1:
            _, old = self._d.popitem(last=False)
            self.size -= len(old[1])
        return cp


def pack_batch(compiled, q_lens):
    """Concatenate compiled programs into the host arrays of stair_plan_build: (prog_off, tokens, lo, hi, q_off)."""
    n = len(compiled)
    prog_off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum([c.n_tokens for c in compiled], out=prog_off[1:])
    allp = np.concatenate([c.packed for c in compiled], axis=1) if n else np.zeros((3, 0), dtype=np.int32)
    tokens, lo, hi = allp[0], allp[1], allp[2]          # rows of a C-contiguous [3, total] array: each contiguous
    q_off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.asarray(q_lens, dtype=np.int64), out=q_off[1:])
    return prog_off, tokens, lo, hi, q_off
