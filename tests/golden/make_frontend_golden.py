#!/usr/bin/env python3
"""Regenerates tests/golden/frontend.json: the REFERENCE's parse_program / tree helpers run on AGQA-grammar program
strings (P* are the eight strings of tests/golden/programs.json; F* were written for this fixture to reach the
remaining rewrite rules: nested IterateUntil, backward, Array forms, Superlative/Subtract, Compare, Choose, AND/XOR).
Only runs where /root/reference exists; the fixture (data: inputs and the reference's outputs) is committed.

    python tests/golden/make_frontend_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference/utils')
import program_parser as pp            # noqa: E402  (pure python, no third-party imports)

HERE = os.path.dirname(os.path.abspath(__file__))

REL = lambda r: '[relations, %s, objects]' % r
FRAME_REL = lambda r: 'Iterate(frame, Filter(frame, %s))' % REL(r)

STRINGS = {
    'F0': 'Query(class, OnlyItem(Iterate(Localize(before, putting a phone somewhere), Filter(frame, [relations, holding, objects]))))',
    'F1': 'Query(class, OnlyItem(IterateUntil(backward, Localize(before, opening a door), HasItem(%s), %s)))' % (FRAME_REL('touching'), FRAME_REL('touching')),
    'F2': 'Exists(phone, IterateUntil(forward, video, Exists(blanket, Filter(frame, [objects])), Filter(frame, [objects])))',
    'F3': 'Equals(dish, Query(class, OnlyItem(Iterate(Localize(while, sitting on a sofa), Filter(frame, [relations, holding, objects])))))',
    'F4': 'Superlative(min, Filter(video, [actions]), Subtract(Query(end, action), Query(start, action)))',
    'F5': 'Superlative(max, Filter(Localize(after, washing a window), [actions]), Subtract(Query(end, action), Query(start, action)))',
    'F6': 'Compare([before, after], Exists(holding a dish, Iterate(Localize(temporal tag, taking a phone from somewhere), Filter(frame, [actions]))))',
    'F7': 'Choose(holding, touching, Iterate(Localize(between, [opening a door, closing a door]), Filter(frame, [relations])))',
    'F8': 'AND(Exists(ToAction(holding, dish), Filter(Localize(before, eating a sandwich), [actions])), Exists(ToAction(washing, dish), Filter(Localize(after, eating a sandwich), [actions])))',
    'F9': 'XOR(Exists(door, Iterate(video, Filter(frame, [objects]))), Exists(window, Iterate(video, Filter(frame, [objects]))))',
    'F10': 'Exists(ToAction(Query(class, OnlyItem(Iterate(video, Filter(frame, [relations])))), dish), Filter(video, [actions]))',
    'F11': 'Query(class, OnlyItem(IterateUntil(forward, Localize(after, putting a dish somewhere), XOR(HasItem(%s), HasItem(%s)), %s)))' % (FRAME_REL('holding'), FRAME_REL('touching'), FRAME_REL('holding')),
    'F12': 'Equals(Query(class, OnlyItem(Iterate(Localize(before, opening a door), Filter(frame, [relations, holding, objects])))), Query(class, OnlyItem(Iterate(Localize(after, opening a door), Filter(frame, [relations, holding, objects])))))',
    'F13': 'Exists(dish, Iterate(Localize(between, [opening a door, closing a door]), Filter(frame, [objects])))',
    'F14': 'Exists(holding, Iterate(video, Filter(frame, [relations])))',
    'F15': 'Choose(before, after, Exists(eating a sandwich, Iterate(Localize(while, holding a dish), Filter(frame, [actions]))))',
    'F16': 'Query(class, OnlyItem(IterateUntil(backward, video, Exists(dish, Filter(frame, [objects])), Iterate(frame, Filter(frame, [relations, holding, objects])))))',
    'F17': 'AND(XOR(Exists(door, Iterate(video, Filter(frame, [objects]))), Exists(dish, Iterate(video, Filter(frame, [objects])))), Exists(phone, Iterate(Localize(while, sitting on a sofa), Filter(frame, [objects]))))',
    'F18': 'Exists(food, Iterate(Localize(before, [grasping onto a doorknob, drinking from a cup]), Filter(frame, [relation, holding, objects])))',
    # strings the reference rejects or handles oddly (its failure type / odd output is the expected behaviour)
    'E0': 'Query(class, OnlyItem(IterateUntil(forward, video, Exists(dish, Filter(frame, [objects])), video)))',
    'E1': 'Query(class, OnlyItem(IterateUntil(forward, video, Exists(dish, Filter(frame, [objects])), Iterate(Localize(before, opening a door), Filter(frame, [objects])))))',
    'E2': 'Exists(dish, Iterate(video, Filter(frame, [a, b, c, d])))',
    'E3': 'Superlative(max, video, video)',
    'E4': 'Query(class, OnlyItem(IterateUntil(forward, video, Exists(dish, Filter(frame, [objects])), Filter(video, [relations, holding, objects]))))',
    'F19': 'Query(class, OnlyItem(IterateUntil(forward, Localize(after, eating a sandwich), HasItem(%s), Iterate(frame, Filter(frame, [relations, wiping, objects])))))' % FRAME_REL('holding'),
}


def main():
    old = json.load(open(os.path.join(HERE, 'programs.json')))
    strings = {k: old[k]['string'] for k in old if k.startswith('P')}
    strings.update(STRINGS)
    out = {'_raw_arity': pp.parse_nary_mappings, '_nmn_arity': pp.nary_mappings, 'cases': {}}
    for key, s in strings.items():
        case = {'string': s}
        try:
            nmn, more = pp.parse_program(s)
            case.update(nmn=nmn, idx=more['idx_list'], common=more['common_list'],
                        mapping=(None if more['existsframe_filterframe_idx_mapping'] is None else
                                 {str(k): v for k, v in more['existsframe_filterframe_idx_mapping'].items()}),
                        valid=pp.program_is_valid(nmn))
            if case['valid']:
                ch, pa = pp.get_childrens_and_parents(nmn)
                case.update(levels=pp.stat_module_levels(nmn), children=ch, parents=pa)
        except Exception as e:                 # a string the reference itself cannot parse: record how it fails
            case['error'] = type(e).__name__
        out['cases'][key] = case
    # validity of damaged programs (drop / duplicate a token of every parsed program)
    probes = []
    for key, case in out['cases'].items():
        if 'nmn' not in case:
            continue
        nmn = case['nmn']
        for variant in (nmn[1:], nmn[:-1], nmn + nmn[-1:], nmn[:2] + nmn[3:]):
            probes.append({'program': variant, 'valid': pp.program_is_valid(variant)})
    out['validity_probes'] = probes
    json.dump(out, open(os.path.join(HERE, 'frontend.json'), 'w'), indent=0, sort_keys=True)
    ok = sum('nmn' in c for c in out['cases'].values())
    print('wrote %d cases (%d parsed, %d raise), %d validity probes' % (len(out['cases']), ok, len(out['cases']) - ok, len(probes)))
    for k, c in out['cases'].items():
        print(k, c.get('error') or ('valid' if c['valid'] else 'INVALID'), ' '.join(c.get('nmn', [])))


if __name__ == '__main__':
    main()
