#!/usr/bin/env python3
"""Regenerates tests/golden/spans.json: the REFERENCE's span matcher, utils/agqa_lite.py:62-119
(get_program_list_string_index), run on (program, question text) pairs.

nltk (word_tokenize, pos_tag, WordNetLemmatizer) and tkinter are not installed in the build image, so the three language
tools the function calls are INJECTED: stand-in modules named nltk / nltk.stem / nltk.tokenize / tkinter are registered
before the import and route to stair_amd.frontend.Normaliser (a regex tokenizer, an 'ing'/'ed' tagger, a suffix
stripper).  What the fixture pins is therefore the reference's matching LOGIC around those tools -- the rewrite tables
of questions and programs, the 'ing' -> verb override, the 'clothes' exception, the scan that stops one position early,
the character spans -- under exactly the normaliser the test injects into frontend.match_spans; the behaviour of the real
nltk tools stays parity-unpinned.  Only runs where /root/reference exists; data only is committed.

    python tests/golden/make_spans_golden.py
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, '/root/reference/utils')

from stair_amd.frontend import Normaliser            # noqa: E402

NZ = Normaliser()
nltk = types.ModuleType('nltk')
nltk.pos_tag = lambda words: NZ.pos_tag(list(words))
stem = types.ModuleType('nltk.stem')


class WordNetLemmatizer:
    def lemmatize(self, word, pos='n'):
        return NZ.lemmatize(word, pos)


stem.WordNetLemmatizer = WordNetLemmatizer
tok = types.ModuleType('nltk.tokenize')
tok.word_tokenize = lambda text: NZ.tokenize(text)
corpus = types.ModuleType('nltk.corpus')
corpus.stopwords = type('S', (), {'words': lambda self, l: []})()
nltk.stem, nltk.tokenize, nltk.corpus = stem, tok, corpus
for name, mod in (('nltk', nltk), ('nltk.stem', stem), ('nltk.tokenize', tok), ('nltk.corpus', corpus), ('tkinter', types.ModuleType('tkinter'))):
    sys.modules[name] = mod
sys.modules['tkinter'].Frame = object          # scene_graphs.py:7 imports it and never uses it

import agqa_lite            # noqa: E402  (the reference)
import program_parser as pp  # noqa: E402

# question texts written for this fixture in AGQA's template style; each holds the phrases of its program somewhere
QUESTIONS = {
    'P0': 'Did they interact with food or open something between grasping onto a doorknob and drinking from a cup while holding it ?',
    'P1': 'Does someone touch a dish in the video ?',
    'P2': 'What did the person hold first after eating a sandwich and holding on ?',
    'P3': 'Which action was the longest ?',
    'P4': 'Was the thing they were holding the same as the thing they were touching then ?',
    'P5': 'Were they eating a sandwich before or after opening a door today ?',
    'P6': 'Was it the dish or the blanket they were holding while holding a dish up ?',
    'P7': 'Did they hold a dish and close the door after that ?',
    'C0': 'Did they hold the cup or the dish while holding on ?',
    'C1': 'Were they running or jumping for the least time ?',
    'C2': 'Was the phone something they were holding up ?',
    'C3': 'Did they consume food while they lay on the clothes closing a door ?',
    # matcher corner cases: a phrase that ends the question (missed by the early-stopping scan), a rewritten verb,
    # a repeated word (character spans restart at the previous word's start), an unmatched phrase
    'X0': 'Were they drinking from a cup',
    'X1': 'Did the person who ate the sandwich open the door door ?',
    'X2': 'Is the blanket on the sofa ?',
}
EXTRA_PROGRAMS = {
    'X0': ['Exists', 'drinking_from_a_cup', 'Filter', 'video', 'actions'],
    'X1': ['Exists', 'eating_a_sandwich', 'Filter', 'Temporal', 'while', 'video', 'Localize', 'video', 'opening_a_door', 'actions'],
    'X2': ['Exists', 'phone', 'Filter', 'video', 'objects'],
}


def main():
    progs = json.load(open(os.path.join(HERE, 'programs.json')))
    out = {}
    for key, question in QUESTIONS.items():
        if key in progs:
            nmn, _ = pp.parse_program(progs[key]['string']) if 'string' in progs[key] and key.startswith('P') else (progs[key]['nmn'], None)
        else:
            nmn = EXTRA_PROGRAMS[key]
        by_word, by_char = agqa_lite.get_program_list_string_index(list(nmn), question)
        out[key] = {'question': question, 'nmn': list(nmn),
                    'by_word': {str(k): list(v) for k, v in by_word.items()}, 'by_char': {str(k): list(v) for k, v in by_char.items()}}
        print(key, {k: v for k, v in by_word.items()})
    json.dump(out, open(os.path.join(HERE, 'spans.json'), 'w'), indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
