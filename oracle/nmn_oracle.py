"""ORACLE -- CPU restatement of STAIR's NMN hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (stair_amd/) never does and fails loudly when the HIP library is missing.

What it restates: the stack interpreter of /root/reference/video_nmn/module_net.py:65-145 and the
eighteen operators of /root/reference/video_nmn/modules.py, as plain functions over a
{state_dict_key: tensor} weight dict, batch-1, fp32, same operation order as the reference, dropout
in eval mode (identity).  Each function cites the reference lines it follows.

Pinning: tests/golden/*.npz hold outputs of the reference itself (imported on CPU in the build
container by tests/golden/make_golden.py, which is committed); tests/test_oracle_golden.py checks
every tensor of this oracle against them.  The reference has no tests or golden vectors of its
own, so those generated fixtures are the pin.
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

ARITY = {
    'HasItem': 1,
    'And': 2, 'Xor': 2, 'Compare': 2, 'Equals': 2, 'Exists': 2, 'Filter': 2, 'Localize': 2,
    'ToAction': 2, 'Relate': 2, 'AttnVideo': 2, 'FilterFrame': 2, 'ExistsFrame': 2, 'XorFrame': 2,
    'Array2': 2,
    'Superlative': 3, 'Choose': 3, 'Temporal': 3,
}                                   # utils/program_parser.py:16-23
WORDS_TO_KEEP = {'forward', 'backward', 'while', 'between', 'before', 'after', 'max', 'min',
                 'start', 'end', 'video', 'actions', 'objects', 'relations'}
                                    # video_nmn/dataset.py:23 + module_net.py:25-26
P = 'submodules.'


def to_torch(weights):
    return {k: torch.as_tensor(v) for k, v in weights.items()}


def _lin(w, prefix, x):
    """nn.Linear: y = x W^T + b."""
    return F.linear(x, w[prefix + '.weight'], w[prefix + '.bias'])


# ------------------------------------------------------------------------------------------
# encoders  (module_net.py:39-47, 151-163)
# ------------------------------------------------------------------------------------------
def lstm_bidir_explicit(w, enc, x):
    """Single-layer bidirectional LSTM written out.  x [L, I] -> (out [L, 2*Hh], h_n [2, Hh]).

    Gate order i, f, g, o; zero initial state; both bias vectors added; out[t] = [h_fwd_t ; h_bwd_t]
    (torch.nn.LSTM semantics as instantiated at module_net.py:39-47).
    """
    L = x.shape[0]
    outs, finals = [], []
    for sfx, order in (('', range(L)), ('_reverse', range(L - 1, -1, -1))):
        w_ih, w_hh = w[P + enc + '.weight_ih_l0' + sfx], w[P + enc + '.weight_hh_l0' + sfx]
        b = w[P + enc + '.bias_ih_l0' + sfx] + w[P + enc + '.bias_hh_l0' + sfx]
        Hh = w_hh.shape[1]
        h = torch.zeros(Hh, dtype=x.dtype)
        c = torch.zeros(Hh, dtype=x.dtype)
        xp = x @ w_ih.t() + b
        hs = [None] * L
        for t in order:
            g = xp[t] + w_hh @ h
            i, f, gg, o = g[:Hh], g[Hh:2 * Hh], g[2 * Hh:3 * Hh], g[3 * Hh:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs))
        finals.append(h)
    return torch.cat(outs, dim=1), torch.stack(finals)


def lstm_bidir(w, enc, x):
    """Same result through ATen's fused LSTM (the library op the reference itself runs on CPU); used
    for the timed cpu_baseline so the baseline is not handicapped by a python time loop."""
    flat = []
    for sfx in ('', '_reverse'):
        flat += [w[P + enc + '.weight_ih_l0' + sfx], w[P + enc + '.weight_hh_l0' + sfx],
                 w[P + enc + '.bias_ih_l0' + sfx], w[P + enc + '.bias_hh_l0' + sfx]]
    Hh = flat[1].shape[1]
    z = torch.zeros(2, 1, Hh, dtype=x.dtype)
    out, h_n, _ = torch.lstm(x.unsqueeze(0), (z, z), flat, True, 1, 0.0, False, True, True)
    return out[0], h_n[:, 0, :]


def encode_video(w, video, explicit=False):
    """module_net.py:160-163."""
    out, _ = (lstm_bidir_explicit if explicit else lstm_bidir)(w, 'video_encoder', video)
    return out


def encode_question(w, question, explicit=False):
    """module_net.py:151-158: token_feature [Q,H]; sentence = [h_fwd after last token ; h_bwd after token 0]."""
    out, h_n = (lstm_bidir_explicit if explicit else lstm_bidir)(w, 'text_encoder', question)
    return out, h_n.reshape(-1)


def l2normalize(x):
    """module_net.py:211-216 (F.normalize, dim 0, eps 1e-12)."""
    return x / x.norm().clamp_min(1e-12)


# ------------------------------------------------------------------------------------------
# operators  (modules.py)
# ------------------------------------------------------------------------------------------
def cos_rows(a, b, eps=1e-8):
    """nn.CosineSimilarity(dim=-1) as ATen >= 2.0 evaluates it: normalise each side by
    max(||.||, eps), then sum the products."""
    an = a / a.norm(dim=-1, keepdim=True).clamp_min(eps)
    bn = b / b.norm(dim=-1, keepdim=True).clamp_min(eps)
    return (an * bn).sum(-1)


def _mlp2(w, prefix, x, relu_last=True):
    """Lin . ReLU . (Dropout) . Lin [. ReLU . (Dropout)] -- the '.0' / '.3' Sequential pattern."""
    y = _lin(w, prefix + '.3', torch.relu(_lin(w, prefix + '.0', x)))
    return torch.relu(y) if relu_last else y


def op_and(a, b):                      # modules.py:7-12
    return torch.minimum(a, b)


def op_xorframe(a, b):                 # modules.py:75-80
    return (a - b).abs()


def op_compare(w, a, b):               # modules.py:15-21
    return torch.relu(_lin(w, P + 'Compare.param.0', torch.cat([a, b])))


def op_equals(w, a, b):                # modules.py:24-37
    return torch.relu(_lin(w, P + 'Equals.param.0', torch.cat([a, b])))


def op_choose(kw1, kw2, query):        # modules.py:40-56
    c1 = cos_rows(kw1.unsqueeze(0), query.unsqueeze(0))
    c2 = cos_rows(kw2.unsqueeze(0), query.unsqueeze(0))
    return kw1 if bool(c1 > c2) else kw2


def op_xor(w, a, b):                   # modules.py:59-72
    return torch.relu(_lin(w, P + 'Xor.param.0', torch.cat([(a - b).abs(), a, b])))


def op_toaction(w, action, kw):        # modules.py:102-120
    return _mlp2(w, P + 'ToAction.param', torch.cat([action, kw]))


def op_hasitem(w, feat):               # modules.py:123-138
    y = _lin(w, P + 'HasItem.param.3', torch.relu(_lin(w, P + 'HasItem.param.0', feat)))
    return torch.sigmoid(y).squeeze()


def op_exists(w, kw, feat):            # modules.py:141-159
    return _mlp2(w, P + 'Exists.param', torch.cat([feat, kw, feat * kw]))


def op_existsframe(kw, feat):          # modules.py:162-178
    return (cos_rows(feat, kw.unsqueeze(0)) + 1) * 0.49


def op_localize(w, feat, kw):          # modules.py:181-217
    f = _mlp2(w, P + 'Localize.video_linear', feat, relu_last=False)          # [T,H]
    if kw.dim() == 1:
        kw = kw.unsqueeze(0)
    k = _lin(w, P + 'Localize.keyword_linear.0', kw)                          # [K,H]
    return (cos_rows(f.unsqueeze(0), k.unsqueeze(1)) + 1) * 0.49              # [K,T]


def op_superlative(w, mode, actions, feat):        # modules.py:220-248
    s = op_localize(w, feat, actions)                                         # [Ka,T]
    wt = torch.softmax(s.sum(dim=1), dim=0)
    if mode == 'min':
        wt = 1 - wt
    return torch.relu(_lin(w, P + 'Superlative.dense.0', (wt.unsqueeze(1) * actions).sum(0)))


def conv1d_same(x, weight, bias):
    """nn.Conv1d(1,1,k,padding='same', zeros): left pad (k-1)//2, right pad the rest (torch puts the
    odd element on the right), cross-correlation."""
    k = weight.numel()
    left = (k - 1) // 2
    xp = F.pad(x, (left, k - 1 - left))
    return F.conv1d(xp.view(1, 1, -1), weight.view(1, 1, -1), bias.view(1)).view(-1)


def temporal_relate(w, mode, a):
    """modules.py:255-277 + :317-323: the learned 'relate' net on the mean attention a [T]."""
    if mode == 'while':
        return a
    pre = P + 'Temporal.relate.%s.' % mode
    if w[pre + '0.weight'].dim() == 3:                       # conv variant
        y = torch.relu(conv1d_same(a, w[pre + '0.weight'], w[pre + '0.bias']))
        y = torch.relu(conv1d_same(y, w[pre + '2.weight'], w[pre + '2.bias']))
        return torch.sigmoid(conv1d_same(y, w[pre + '4.weight'], w[pre + '4.bias']))
    y = torch.relu(_lin(w, pre + '0', a))
    y = torch.relu(_lin(w, pre + '2', y))
    return torch.sigmoid(_lin(w, pre + '4', y))


def op_temporal(w, mode, feat, attn):              # modules.py:310-327
    r = temporal_relate(w, mode, attn.mean(dim=0))
    y = torch.relu(_lin(w, P + 'Temporal.dense.0', r.unsqueeze(-1) * feat))
    out = F.layer_norm(y, (y.shape[-1],), w[P + 'Temporal.layer_norm.weight'],
                       w[P + 'Temporal.layer_norm.bias'], 1e-5)
    return out, r


def op_attnvideo(feat, attn):          # modules.py:330-340
    return attn.unsqueeze(1) * feat


def op_filter(w, feat, kw):            # modules.py:343-378
    if isinstance(kw, torch.Tensor):
        f = _mlp2(w, P + 'Filter.param.representation', feat)
        fk = torch.cat([f, kw.unsqueeze(0).expand(f.shape[0], -1)], dim=1)
        # nn.Softmax() with no dim on a [T,1] tensor resolves to dim=1 -> every score is exactly 1
        a = torch.softmax(_lin(w, P + 'Filter.attention.0', fk), dim=1)
        agg = (a * f).sum(0)
    else:
        agg = _mlp2(w, P + 'Filter.param.' + kw, feat).sum(0)
    return torch.relu(_lin(w, P + 'Filter.dense.0', agg))


def op_filterframe(w, feat, kw):       # modules.py:381-414
    if isinstance(kw, torch.Tensor):
        f = _mlp2(w, P + 'FilterFrame.param.representation', feat)
        fk = torch.cat([f, kw.unsqueeze(0).expand(f.shape[0], -1)], dim=1)
        agg = torch.sigmoid(_lin(w, P + 'FilterFrame.attention.0', fk)) * f
    else:
        agg = _mlp2(w, P + 'FilterFrame.param.' + kw, feat)
    return torch.relu(_lin(w, P + 'FilterFrame.dense.0', agg))


def op_relate(w, mode, attn):          # modules.py:417-435 (nn.Softmax() on 1-D -> dim 0)
    beta = w[P + 'Relate.beta'][:attn.shape[0]]
    return torch.softmax(attn + beta if mode == 'forward' else attn - beta, dim=0)


def pretrain_head(w, prog, result, related_attn=None):
    """The per-module pretrain_head (modules.py: Equals:29, Xor:63, Exists:149, FilterFrame:396 are
    Linear; Filter:363 / Superlative:234 / ToAction:110 share module_net.py:21 L2Normalize;
    Localize/HasItem/ExistsFrame are Identity; Temporal:285-288 returns the last related_attn)."""
    if prog in ('Exists', 'Xor', 'Equals', 'FilterFrame'):
        return _lin(w, P + prog + '.pretrain_head', result)
    if prog in ('Filter', 'Superlative', 'ToAction'):
        return l2normalize(result)
    if prog == 'Temporal':
        return related_attn
    if prog in ('Localize', 'HasItem', 'ExistsFrame'):
        return result
    raise AttributeError('%s has no pretrain_head' % prog)     # torch raises AttributeError too


def run_module(w, prog, params):
    """Dispatch one module call; returns (result, related_attn or None)."""
    if prog == 'And':
        return op_and(*params), None
    if prog == 'XorFrame':
        return op_xorframe(*params), None
    if prog == 'Compare':
        return op_compare(w, *params), None
    if prog == 'Equals':
        return op_equals(w, *params), None
    if prog == 'Choose':
        return op_choose(*params), None
    if prog == 'Xor':
        return op_xor(w, *params), None
    if prog == 'ToAction':
        return op_toaction(w, *params), None
    if prog == 'HasItem':
        return op_hasitem(w, *params), None
    if prog == 'Exists':
        return op_exists(w, *params), None
    if prog == 'ExistsFrame':
        return op_existsframe(*params), None
    if prog == 'Localize':
        return op_localize(w, *params), None
    if prog == 'Superlative':
        return op_superlative(w, *params), None
    if prog == 'Temporal':
        return op_temporal(w, *params)
    if prog == 'AttnVideo':
        return op_attnvideo(*params), None
    if prog == 'Filter':
        return op_filter(w, *params), None
    if prog == 'FilterFrame':
        return op_filterframe(w, *params), None
    if prog == 'Relate':
        return op_relate(w, *params), None
    if prog == 'Array2':
        return torch.stack(list(params)), None              # modules.py:438-443
    raise KeyError(prog)


# ------------------------------------------------------------------------------------------
# interpreter  (module_net.py:65-145)
# ------------------------------------------------------------------------------------------
def forward(w, config, data, return_res_by_step=True, return_result_of_each_step=False,
            pretrain_modules=frozenset(), explicit_lstm=False):
    """VideoNMN.forward restated.  ``data`` as in dataset.py:191-233 (tensors or ndarrays)."""
    question = torch.as_tensor(data['question'])
    video = torch.as_tensor(data['video_features'])
    spans = data['prog_str_to_question_tokens']
    program, program_idx = data['nmn_program_list'], data['nmn_program_idx']

    video_feat = encode_video(w, video, explicit_lstm)
    token_feature, question_feature = encode_question(w, question, explicit_lstm)

    stack, res_by_step, each = [], {}, []
    for i in range(len(program) - 1, -1, -1):                       # module_net.py:97
        prog = program[i]
        params = []
        if prog in ARITY:
            for _ in range(ARITY[prog]):
                p = stack.pop()
                if isinstance(p, str) and p == 'video':
                    p = video_feat
                params.append(p)
            result, rel = run_module(w, prog, params)
            want_head = config['have_pretrain_head'] and prog in pretrain_modules
            if return_res_by_step and program_idx[i] is not None and prog in pretrain_modules and i != 0:
                res_by_step[program_idx[i]] = (prog, pretrain_head(w, prog, result, rel) if config['have_pretrain_head'] else result)
            if return_result_of_each_step:
                each.append((params, pretrain_head(w, prog, result, rel) if want_head else result))
        elif prog in WORDS_TO_KEEP:
            result = prog
            if return_result_of_each_step:
                each.append((params, result))
        else:
            s, e = spans[i]
            result = token_feature[s:e, :].mean(dim=0)              # module_net.py:128-129
            if return_result_of_each_step:
                each.append((params, result))
        stack.append(result)

    assert len(stack) == 1                                          # module_net.py:135
    hidden = stack[0]
    hq = torch.cat([hidden, question_feature])
    logits = _lin(w, P + 'decoder.3', torch.relu(_lin(w, P + 'decoder.0', hq)))
    ret = {'logits': logits, 'res_by_step': res_by_step,
           'video_feat': video_feat, 'token_feature': token_feature, 'question_feature': question_feature}
    if return_result_of_each_step:
        ret['result_of_each_step'] = list(reversed(each))
    return ret


# ------------------------------------------------------------------------------------------
# program utilities (utils/program_parser.py:307-333) -- used to check the plan builder's levels
# ------------------------------------------------------------------------------------------
def program_is_valid(program):
    n = 0
    for tok in reversed(program):
        n = n - ARITY[tok] + 1 if tok in ARITY else n + 1
        if n < 0:
            return False
    return n == 1


def module_levels(program):
    """stat_module_levels: leaves 0, a module 1 + max(children)."""
    levels, stack = [], []
    for tok in reversed(program):
        if tok not in ARITY:
            stack.append(0)
            levels.append(0)
        else:
            k = ARITY[tok]
            lvl = max(stack[-k:]) + 1
            del stack[-k:]
            stack.append(lvl)
            levels.append(lvl)
    return levels[::-1]


# ------------------------------------------------------------------------------------------
# evaluate.py:65-117 get_filter_text_results -- pinned by tests/golden/filter_text.json (reference output)
# ------------------------------------------------------------------------------------------
def children_of(program):
    """utils/program_parser.py:182-201: operand positions of every token, first positional argument first."""
    children, stack = [[] for _ in program], []
    for i in range(len(program) - 1, -1, -1):
        if program[i] in ARITY:
            children[i] = [stack.pop() for _ in range(ARITY[program[i]])]
        stack.append(i)
    return children


def filter_text_results(w, config, questions, vocab, phrase_embeddings, pretrain_modules=frozenset(), top=10):
    """{qa_id: {program_idx: (level, keyword text, top phrases)}} for every Filter node, question by question."""
    reps = torch.stack([l2normalize(encode_question(w, torch.as_tensor(e))[1]) for e in phrase_embeddings])   # :66-76
    out = {}
    for n, d in enumerate(questions):
        prog = d['nmn_program_list']
        pidx = d.get('nmn_program_idx') or list(range(len(prog)))
        steps = forward(w, config, d, return_res_by_step=False, return_result_of_each_step=True,
                        pretrain_modules=pretrain_modules)['result_of_each_step']
        levels, children = module_levels(prog), children_of(prog)
        entry = {}
        for i, tok in enumerate(prog):
            if tok != 'Filter':
                continue
            sims = torch.nn.functional.cosine_similarity(steps[i][1].unsqueeze(0), reps)                      # :95
            order = torch.argsort(sims, descending=True)[:top]
            entry[pidx[i]] = (levels[i], prog[children[i][1]].replace('_', ' '), [vocab[int(c)] for c in order])
        out[d.get('qa_id', n)] = entry
    return out
