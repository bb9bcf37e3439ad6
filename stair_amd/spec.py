"""Static description of the STAIR NMN hot path: weight table, program vocabulary, arities.

Everything here is *data about the reference's interface* that both the host code and the
C-ABI library (include/stair_hip.h) must agree on:

* ``weight_table(config)`` -- canonical (name, shape) list whose names are the reference's
  ``state_dict`` keys (/root/reference/video_nmn/module_net.py:27-53, modules.py:446-465).
  ``Superlative.localize_module.*`` aliases ``Localize.*`` (module_net.py:31-32) and is listed in
  ``WEIGHT_ALIASES`` rather than as separate storage.
* ``ARITY`` -- arity of each interpreter token (/root/reference/utils/program_parser.py:16-23).
* ``OP_*`` / ``KW_*`` integer codes -- the encoding the plan builder (csrc/plan.cpp) consumes.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------------
# program vocabulary
# --------------------------------------------------------------------------------------------
# module tokens that VideoNMN registers (modules.py:446-465); order fixes the integer op code.
MODULE_NAMES = [
    'And', 'AttnVideo', 'Choose', 'Compare', 'Equals', 'Exists', 'ExistsFrame', 'Filter',
    'FilterFrame', 'HasItem', 'Localize', 'Relate', 'Superlative', 'Temporal', 'ToAction', 'Xor',
    'XorFrame', 'Array2',
]
# arity per module token, program_parser.py:16-23 (nary_1 / nary_2 / nary_3 after the rewrites)
ARITY = {
    'HasItem': 1,
    'And': 2, 'Xor': 2, 'Compare': 2, 'Equals': 2, 'Exists': 2, 'Filter': 2, 'Localize': 2,
    'ToAction': 2, 'Relate': 2, 'AttnVideo': 2, 'FilterFrame': 2, 'ExistsFrame': 2, 'XorFrame': 2,
    'Array2': 2,
    'Superlative': 3, 'Choose': 3, 'Temporal': 3,
}
# tokens that stay python strings on the interpreter stack: dataset.py:23 WORDS_TO_KEEP plus the
# three type keywords added in module_net.py:25-26.
KEYWORDS = ['forward', 'backward', 'while', 'between', 'before', 'after', 'max', 'min', 'start',
            'end', 'video', 'actions', 'objects', 'relations']

OP_CODE = {name: i for i, name in enumerate(MODULE_NAMES)}            # 0..17
KW_BASE = 100
KW_CODE = {name: KW_BASE + i for i, name in enumerate(KEYWORDS)}       # 100..113
TOK_SPAN = 200                                                        # any other token: span mean

# modules whose reference class defines a pretrain_head when have_pretrain_head (modules.py)
HEAD_KIND = {
    'Exists': 'linear', 'Xor': 'linear', 'Equals': 'linear', 'FilterFrame': 'linear',
    'Filter': 'l2norm', 'Superlative': 'l2norm', 'ToAction': 'l2norm',
    'Localize': 'identity', 'HasItem': 'identity', 'ExistsFrame': 'identity',
    'Temporal': 'related_attn',
}

DEFAULT_CONFIG = {
    'hidden_size': 512, 'video_size': 2048, 'text_size': 300, 'dropout': 0.25,
    'answer_vocab_length': 172, 'max_video_length': 64, 'init_method': 'default', 'layer_norm': 1,
    'have_pretrain_head': True, 'object_types': 36,
}


def temporal_mode(config) -> str:
    """modules.py:255-277: Conv1d relate nets when max_video_length > 32, else Linear(T,T)."""
    return 'conv' if config['max_video_length'] > 32 else 'linear'


def temporal_kernel_size(config) -> int:
    """modules.py:258: python round() (banker's rounding) of max_video_length / 4."""
    return round(config['max_video_length'] / 4)


def weight_table(config):
    """Canonical weights in reference state_dict order (aliases excluded).

    Returns a list of (state_dict_key, shape). The position in this list is the integer weight id
    used by ``stair_ctx_set_weight`` (include/stair_hip.h, enum stair_weight_id order is generated
    from the same table by tools/gen_header.py and checked by tests/test_abi.py).
    """
    H, V, E = config['hidden_size'], config['video_size'], config['text_size']
    A, L, O = config['answer_vocab_length'], config['max_video_length'], config['object_types']
    heads = bool(config['have_pretrain_head'])
    Hh = H // 2
    t = []

    def lin(prefix, n_out, n_in):
        t.append((prefix + '.weight', (n_out, n_in)))
        t.append((prefix + '.bias', (n_out,)))

    p = 'submodules.'
    lin(p + 'Compare.param.0', H, 2 * H)
    lin(p + 'Equals.param.0', H, 2 * H)
    if heads:
        lin(p + 'Equals.pretrain_head', 1, H)
    lin(p + 'Exists.param.0', H, 3 * H)
    lin(p + 'Exists.param.3', H, H)
    if heads:
        lin(p + 'Exists.pretrain_head', 2, H)
    for kw in ('representation', 'actions', 'objects', 'relations'):
        lin(p + 'Filter.param.%s.0' % kw, H, H)
        lin(p + 'Filter.param.%s.3' % kw, H, H)
    lin(p + 'Filter.attention.0', 1, 2 * H)
    lin(p + 'Filter.dense.0', H, H)
    for kw in ('representation', 'relations', 'actions'):
        lin(p + 'FilterFrame.param.%s.0' % kw, H, H)
        lin(p + 'FilterFrame.param.%s.3' % kw, H, H)
    lin(p + 'FilterFrame.attention.0', 1, 2 * H)
    lin(p + 'FilterFrame.dense.0', H, H)
    if heads:
        lin(p + 'FilterFrame.pretrain_head', O, H)
    lin(p + 'HasItem.param.0', H, H)
    lin(p + 'HasItem.param.3', 1, H)
    lin(p + 'Localize.video_linear.0', H, H)
    lin(p + 'Localize.video_linear.3', H, H)
    lin(p + 'Localize.keyword_linear.0', H, H)
    t.append((p + 'Relate.beta', (L,)))
    lin(p + 'Superlative.dense.0', H, H)
    if temporal_mode(config) == 'conv':
        k = temporal_kernel_size(config)
        for mode in ('before', 'after', 'between'):
            for layer, ks in ((0, k), (2, k), (4, 2 * k + 1)):
                t.append((p + 'Temporal.relate.%s.%d.weight' % (mode, layer), (1, 1, ks)))
                t.append((p + 'Temporal.relate.%s.%d.bias' % (mode, layer), (1,)))
    else:
        for mode in ('before', 'after', 'between'):
            for layer in (0, 2, 4):
                lin(p + 'Temporal.relate.%s.%d' % (mode, layer), L, L)
    lin(p + 'Temporal.dense.0', H, H)
    t.append((p + 'Temporal.layer_norm.weight', (H,)))
    t.append((p + 'Temporal.layer_norm.bias', (H,)))
    lin(p + 'ToAction.param.0', H, 2 * H)
    lin(p + 'ToAction.param.3', H, H)
    lin(p + 'Xor.param.0', H, 3 * H)
    if heads:
        lin(p + 'Xor.pretrain_head', 2, H)
    for enc, n_in in (('video_encoder', V), ('text_encoder', E)):
        for sfx in ('', '_reverse'):
            t.append((p + '%s.weight_ih_l0%s' % (enc, sfx), (4 * Hh, n_in)))
            t.append((p + '%s.weight_hh_l0%s' % (enc, sfx), (4 * Hh, Hh)))
            t.append((p + '%s.bias_ih_l0%s' % (enc, sfx), (4 * Hh,)))
            t.append((p + '%s.bias_hh_l0%s' % (enc, sfx), (4 * Hh,)))
    lin(p + 'decoder.0', 2 * H, 2 * H)
    lin(p + 'decoder.3', A, 2 * H)
    return t


# alias -> canonical (module_net.py:31-32: Superlative is constructed with the Localize instance)
WEIGHT_ALIASES = {
    'submodules.Superlative.localize_module.' + s: 'submodules.Localize.' + s
    for s in ('video_linear.0.weight', 'video_linear.0.bias', 'video_linear.3.weight',
              'video_linear.3.bias', 'keyword_linear.0.weight', 'keyword_linear.0.bias')
}


def state_dict_keys(config):
    """All keys of the reference state_dict, in its order (119 with heads / 111 without)."""
    keys = []
    for name, _ in weight_table(config):
        keys.append(name)
        if name == 'submodules.Relate.beta':
            keys.extend(WEIGHT_ALIASES.keys())
    # reference order puts Superlative.localize_module.* before Superlative.dense.* and after
    # Relate.beta, which is exactly where the loop above inserts them.
    return keys


def encode_program(program_list):
    """token strings -> int32 codes for the plan builder (module code, keyword code or TOK_SPAN)."""
    out = []
    for tok in program_list:
        if tok in OP_CODE:
            out.append(OP_CODE[tok])
        elif tok in KW_CODE:
            out.append(KW_CODE[tok])
        else:
            out.append(TOK_SPAN)
    return out
