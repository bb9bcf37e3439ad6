"""GPU tests of the training path (`-m gpu`): the HIP backward pass against torch autograd of the oracle
(the oracle is differentiable torch code, itself pinned to the reference's forward by tests/golden), and
the Adam / LambdaLR step against torch.optim on the oracle's weights."""
import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import spec, synth
from helpers import load_golden, question_for

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(params=['f32', 'bf16x3'])
def matmul(request):
    """exact fp32 MFMA vs the default split-precision kernels in every GEMM of forward and backward"""
    from stair_amd import ops
    ops.set_matmul_mode(request.param)
    yield request.param
    ops.set_matmul_mode('bf16x3')


def _model(config, seed=0):
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(config)
    w = synth.make_weights(config, seed)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    return m.to(DEV)


def _oracle_params(config, seed):
    names = [n for n, _ in spec.weight_table(config)]
    w = synth.make_weights(config, seed)
    return names, {k: torch.from_numpy(w[k].copy()).requires_grad_(True) for k in names}


def _oracle_loss(w, config, qs, scale):
    total, per_q = 0.0, []
    for q in qs:
        logits = O.forward(w, config, q, return_res_by_step=False, explicit_lstm=True)['logits']
        ce = torch.nn.functional.cross_entropy(logits.unsqueeze(0), torch.tensor([q['answer']]))   # train_module.py:193-194
        per_q.append(float(ce))
        total = total + ce * scale
    return total, per_q


def _pack(model, qs):
    video = torch.stack([torch.as_tensor(q['video_features']) for q in qs]).to(DEV)
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    return ([q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs], video, question,
            [q['question'].shape[0] for q in qs], torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV))


@pytest.mark.parametrize('name', ['tiny_conv', 'tiny_linear', 'tiny_conv_t24'])
def test_backward_matches_autograd_of_oracle(name, matmul):
    """All 12 program forms in one batch: every parameter gradient of the decoder CE loss."""
    z, meta = load_golden(name)
    config = meta['config']
    qs = [question_for(meta, q) for q in meta['questions']]
    scale = 1.0 / len(qs)
    names, w = _oracle_params(config, meta['seed'])
    loss, per_q = _oracle_loss(w, config, qs, scale)
    loss.backward()

    model = _model(config, meta['seed'])
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    res = model.run_programs(progs, spans, video, question, q_lens, train=True)
    # training-mode forward == inference forward == reference
    for qi, q in enumerate(meta['questions']):
        assert float((res.logits[qi].cpu() - torch.as_tensor(z['q%d/logits' % q['qid']])).abs().max()) < 2e-5
    losses = res.backward(answers, scale)
    assert np.allclose(losses.cpu().numpy(), per_q, rtol=1e-5, atol=1e-5)
    got = dict(model.named_parameters())
    touched = dict(zip(model._weight_names, res.touched()))
    worst = (0.0, '')
    for n in names:
        ref = w[n].grad
        g = got[n].grad.cpu()
        if ref is None:
            assert not touched[n] or 'Filter.attention' in n, n
            assert float(g.abs().max()) == 0.0, n
            continue
        assert touched[n], n
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3)
        err = float((g - ref).abs().max())
        worst = max(worst, (err / tol, n))
        assert err < tol, (n, err, float(ref.abs().max()))
    print('worst gradient error / tolerance:', worst)


@pytest.mark.parametrize('H,L,T', [(64, 40, 7), (128, 40, 33), (128, 2, 2), (512, 8, 8), (512, 40, 33), (64, 100, 100)])
def test_backward_on_odd_shapes(H, L, T):
    """Every parameter gradient vs autograd of the oracle for odd frame counts / hidden sizes, exact-fp32 products (in
    split mode single ReLU masks flip at H = 512, see test_full_size_backward_sample)."""
    from stair_amd import ops
    ops.set_matmul_mode('f32')
    try:
        config = dict(spec.DEFAULT_CONFIG, hidden_size=H, video_size=128, answer_vocab_length=16, max_video_length=L, object_types=10)
        qs = synth.make_questions(config, 5, 12, forms=synth.ALL_FORMS, T=T)
        names, w = _oracle_params(config, 1)
        loss, _ = _oracle_loss(w, config, qs, 1.0 / len(qs))
        loss.backward()
        model = _model(config, 1)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        res = model.forward_batch(qs, train=True)
        res.backward(torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV), 1.0 / len(qs))
        got = dict(model.named_parameters())
        for n in names:
            ref = w[n].grad
            if ref is None:
                continue
            assert float((got[n].grad.cpu() - ref).abs().max()) < 2e-4 * max(float(ref.abs().max()), 1e-3), n
    finally:
        ops.set_matmul_mode('bf16x3')


def test_shared_clip_gradients_equal_expanded_batch(matmul):
    """Training plan over questions that share clips (stair_plan_build_shared): every consumer's gradient
    accumulates into the one encoded map, so parameter gradients equal those of the expanded batch (up to the
    order of the float atomics) -- including the video encoder's, which sees each clip once instead of 4 times."""
    z, meta = load_golden('tiny_conv')
    config = meta['config']
    qs = [question_for(meta, q) for q in meta['questions']]
    clips = [torch.as_tensor(qs[c]['video_features']) for c in range(3)]
    for i, q in enumerate(qs):
        q['video_features'] = clips[i % 3]
    model = _model(config, meta['seed'])
    grads = []
    for share in (False, True):
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        res = model.forward_batch(qs, train=True, share_videos=share)
        assert res._video.shape[0] == (3 if share else len(qs))
        answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
        losses = res.backward(answers, 1.0 / len(qs))
        grads.append(({n: p.grad.clone() for n, p in model.named_parameters()}, losses.clone()))
    (ge, le), (gs, ls) = grads
    assert torch.equal(le, ls)
    for n in ge:
        scale = max(float(ge[n].abs().max()), 1e-3)
        assert float((ge[n] - gs[n]).abs().max()) < (2e-5 if matmul == 'f32' else 1e-4) * scale, n


def test_trainer_steps_match_torch_adam(matmul):
    """Three optimizer steps (different program mixes per window, so some modules are untouched at first)
    against torch.optim.Adam + LambdaLR on the oracle's weights, zero_grad(set_to_none=False) = torch 1.13."""
    from stair_amd.train import Trainer
    z, meta = load_golden('tiny_conv')
    config = meta['config']
    windows = [['P1', 'P4', 'P1', 'P4'], ['P0', 'P2', 'P3', 'P5'], ['P6', 'P7', 'C0', 'C1']]
    names, w = _oracle_params(config, 0)
    opt = torch.optim.Adam([w[n] for n in names], lr=2e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0 + (0.1 - 1.0) / 10 * it if it <= 10 else 0.1)
    model = _model(config, 0)
    tr = Trainer(model, lr=2e-4, scheduler_total_iters=10, skip_untouched='ever', dropout=0.0)
    qid = 100
    for forms in windows:
        qs = []
        for f in forms:
            qs.append(synth.make_question(config, 5, qid, form=f, T=40))
            qid += 1
        loss, _ = _oracle_loss(w, config, qs, 1.0 / len(qs))
        loss.backward()
        opt.step(); opt.zero_grad(set_to_none=False); sched.step()
        progs, spans, video, question, q_lens, answers = _pack(model, qs)
        tr.step(progs, spans, video, question, q_lens, answers)
    got = dict(model.named_parameters())
    for n in names:
        ref = w[n].detach()
        diff = (got[n].detach().cpu() - ref).abs()
        # Adam's first steps move every touched weight by ~lr = 2e-4 whatever the gradient's scale (m / sqrt(v) ~ +-1),
        # so entries whose gradient is ~1e5 times smaller than the tensor's typical entry see the split kernels'
        # ~1e-5 relative noise as an O(1) change of m / sqrt(v).  Exact mode: every entry within 2e-5.  Split mode:
        # 99.5 % of the entries within 2e-5 and none further than one step (3 steps x lr = 6e-4 is the hard bound).
        if matmul == 'f32':
            assert float(diff.max()) < 2e-5, (n, float(diff.max()))
        else:
            assert float((diff < 2e-5).float().mean()) > 0.995, (n, float((diff < 2e-5).float().mean()))
            assert float(diff.max()) < 2.5e-4, (n, float(diff.max()))
    # an untouched-so-far tensor must be bit-identical to its initial value
    init = synth.make_weights(config, 0)
    # (Filter 'relations' weights are only used by form C2, which is in none of the windows)
    n = 'submodules.Filter.param.relations.0.weight'
    assert torch.equal(got[n].detach().cpu(), torch.from_numpy(init[n]))
    assert not torch.equal(got['submodules.Compare.param.0.weight'].detach().cpu(), torch.from_numpy(init['submodules.Compare.param.0.weight']))


@pytest.mark.parametrize('fixture', ['tiny_conv_grads', 'tiny_linear_grads'])
def test_backward_matches_reference_backward(fixture, matmul):
    """HIP backward vs the REFERENCE's own loss.backward() (fixtures generated by tests/golden/make_golden.py)."""
    from helpers import compare_with_reference_grads
    z, meta = load_golden(fixture)
    config = meta['config']
    qs = [synth.make_question(config, meta['seed'], qid, form=form, T=meta['T']) for qid, form in enumerate(meta['forms'])]
    model = _model(config, meta['seed'])
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    res = model.run_programs(progs, spans, video, question, q_lens, train=True)
    losses = res.backward(answers, 1.0 / len(qs))
    for qid in range(len(qs)):
        assert abs(float(losses[qid]) - float(z['ce/q%d' % qid])) < 2e-5
    got = dict(model.named_parameters())
    worst, _, _ = compare_with_reference_grads(fixture, lambda n: got[n].grad)
    print('worst gradient error / tolerance vs reference:', worst)


# measured: 2.1 % (f32, LSTM biases: every flip of a module mask reaches the encoder through dX) and 20 % (bf16x3, a 512-entry
# bias) with six questions in the window; a real window of thousands averages the flips out (rel. L2 above)
FRAC_OUTSIDE_STRICT = {'f32': 5e-2, 'bf16x3': 0.3}


def test_full_size_backward_sample(matmul):
    """Full-size shapes (H=512, V=2048, T=64): 6 questions against autograd of the oracle."""
    config = dict(spec.DEFAULT_CONFIG)
    qs = [synth.make_question(config, 21, i, form=f) for i, f in enumerate(['P0', 'P2', 'P3', 'P5', 'C0', 'C1'])]
    names, w = _oracle_params(config, 3)
    loss, per_q = _oracle_loss(w, config, qs, 1.0 / len(qs))
    loss.backward()
    model = _model(config, 3)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    res = model.run_programs(progs, spans, video, question, q_lens, train=True)
    losses = res.backward(answers, 1.0 / len(qs))
    assert np.allclose(losses.cpu().numpy(), per_q, rtol=1e-5, atol=2e-5)
    got = dict(model.named_parameters())
    # At this size (32768 pre-activations per [T,H] tile) a few ReLU inputs land within fp32 rounding of zero, and
    # the HIP GEMM and ATen round them to different signs; the gradient is discontinuous there, so single rows of a
    # weight gradient can differ by O(1/rows) while everything else agrees to 1e-6 (verified by dumping the saved
    # activations: 1 mask flip in 65536 elements, no other difference).  The criterion is therefore the relative
    # L2 error per tensor plus a loose max-abs bound; the tiny-config tests above use the strict elementwise bound.
    worst_frac = (0.0, '')
    for n in names:
        ref = w[n].grad
        if ref is None:
            continue
        g = got[n].grad.cpu()
        rel_l2 = float((g - ref).norm() / ref.norm().clamp_min(1e-12))
        lim = 3e-3 if matmul == 'f32' else 1e-2      # split kernels: ~100x more inputs fall inside the rounding band of a kink
        assert rel_l2 < (lim if ref.numel() >= 64 else 2e-2), (n, rel_l2)     # scalars cannot average a flip out
        assert float((g - ref).abs().max()) < 0.05 * float(ref.abs().max()) + 3e-6, n
        # ... and ELEMENTWISE at the strict bound of the tiny-config tests (2e-4 max|g|) for all but the few rows a flipped
        # mask touches: the fraction of entries outside the strict bound is itself bounded
        frac = float(((g - ref).abs() > 2e-4 * float(ref.abs().max()) + 1e-9).float().mean())
        if ref.numel() >= 512:
            worst_frac = max(worst_frac, (frac, n))
    print('largest fraction of gradient entries outside 2e-4 max|g| (%s): %.3g in %s' % (matmul, worst_frac[0], worst_frac[1]))
    assert worst_frac[0] < FRAC_OUTSIDE_STRICT[matmul], worst_frac


def _with_gold(config, seed, qs, T):
    out = []
    for q in qs:
        q = dict(q)
        q['sg_res_by_step'] = synth.make_gold(config, seed, q, T=T)
        out.append(q)
    return out


def _oracle_view(q):
    """class-name golds as torch tensors, as the reference's dataset hands them over"""
    q = dict(q)
    q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(e)) for n, e in v] if isinstance(v, list) else v)
                           for k, v in q['sg_res_by_step'].items()}
    return q


@pytest.mark.parametrize('name,window', [('tiny_conv', 32), ('tiny_conv', 5), ('tiny_linear', 32)])
def test_intermediate_supervision_losses_and_gradients(name, window, matmul):
    """configs[4]: decoder CE + every per-module loss (attention BCE, Exists/Xor CE, Equals MSE, windowed contrastive
    CE through L2Normalize) -- loss values and all parameter gradients vs autograd of the oracle's restatement of
    train_module.py:341-406 (whose criteria are pinned to the reference's CriterionByModule)."""
    from oracle import nmn_losses as OL
    from stair_amd import losses as L
    z, meta = load_golden(name)
    config, T = meta['config'], meta['T']
    qs = _with_gold(config, 3, [question_for(meta, q) for q in meta['questions']] +
                    [synth.make_question(config, 8, 50 + i, form=f, T=T) for i, f in enumerate(synth.ALL_FORMS)], T)
    n_gold = sum(len(q['sg_res_by_step']) for q in qs)
    assert n_gold > 40
    Gw = len(qs)
    names, w = _oracle_params(config, meta['seed'])
    # the oracle handles ONE contrastive window per call: split the batch the way the trainer windows it
    total, det_all = 0.0, {'module': [], 'decoder': [], 'contrastive': []}
    for s in range(0, len(qs), window):
        t, det = OL.window_loss(w, config, [_oracle_view(q) for q in qs[s:s + window]], L.CRITERION_MODULES,
                                gradient_accumulation=Gw, explicit_lstm=True)
        total = total + t
        for k in det_all:
            det_all[k] += det[k]
    total.backward()

    model = _model(config, meta['seed'])
    model.pretrain_modules = set(L.CRITERION_MODULES)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    res = model.run_programs(progs, spans, video, question, q_lens, train=True)
    res.zero_grad_arenas()
    losses, extra = L.apply_module_losses(model, res, qs, 1.0 / Gw, window=window)
    dec = res.backward(answers, 1.0 / Gw, keep_arenas=True)
    assert np.allclose(dec.cpu().numpy(), det_all['decoder'], rtol=1e-5, atol=2e-5)
    ref_mod = sorted(x[3] for x in det_all['module'])
    got_mod = sorted(torch.cat([v for k, v in losses.items() if k != 'contrastive']).cpu().tolist())
    assert len(ref_mod) == len(got_mod) and np.allclose(got_mod, ref_mod, rtol=2e-5, atol=2e-6)
    ref_c = sorted(x[3] for x in det_all['contrastive'])
    got_c = sorted(losses['contrastive'].cpu().tolist())
    assert len(ref_c) == len(got_c) > 5 and np.allclose(got_c, ref_c, rtol=2e-5, atol=2e-6)
    got = dict(model.named_parameters())
    worst = (0.0, '')
    for n in names:
        ref = w[n].grad
        if ref is None:
            continue
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3)
        err = float((got[n].grad.cpu() - ref).abs().max())
        worst = max(worst, (err / tol, n))
        assert err < tol, (n, err, float(ref.abs().max()))
    assert w['submodules.Exists.pretrain_head.weight'].grad is not None and 'submodules.Exists.pretrain_head.weight' in extra
    print('worst gradient error / tolerance with intermediate supervision:', worst)


def test_trainer_loss_gates_follow_the_global_step():
    """train_module.py:350,376: intermediate losses only while global_steps < train_module_before_iters, decoder loss
    only once global_steps > train_decoder_after_iters, with one global step per QUESTION.  A 12-question window that
    straddles both thresholds: questions 1..4 carry no decoder loss, questions 9.. no intermediate loss."""
    from stair_amd.train import Trainer
    from stair_amd import losses as L
    z, meta = load_golden('tiny_conv')
    config, T = meta['config'], meta['T']
    qs = _with_gold(config, 3, [question_for(meta, q) for q in meta['questions']], T)
    model = _model(config, meta['seed'])
    model.pretrain_modules = set(L.CRITERION_MODULES)
    tr = Trainer(model, train_module_before_iters=9, train_decoder_after_iters=4, dropout=0.0)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    dec, res = tr.step(progs, spans, video, question, q_lens, answers, questions=qs)
    dec = dec.cpu()
    assert float(dec[:4].abs().max()) == 0.0 and float(dec[4:].min()) > 0.0          # global steps 1..4 vs 5..12
    supervised = [i for i, q in enumerate(qs) if q['sg_res_by_step']]
    assert any(i >= 8 for i in supervised) and any(i < 8 for i in supervised)
    n_items = sum(int(v.numel()) for v in tr.module_losses.values())
    model2 = _model(config, meta['seed'])
    model2.pretrain_modules = set(L.CRITERION_MODULES)
    tr2 = Trainer(model2, dropout=0.0)
    tr2.step(progs, spans, video, question, q_lens, answers, questions=qs)
    n_all = sum(int(v.numel()) for v in tr2.module_losses.values())
    assert 0 < n_items < n_all
    assert tr.questions_seen == 12 and tr2.questions_seen == 12
    # second window: every question is past both thresholds -> decoder loss everywhere, no intermediate loss at all
    dec, _ = tr.step(progs, spans, video, question, q_lens, answers, questions=qs)
    assert float(dec.min()) > 0.0 and sum(int(v.numel()) for v in tr.module_losses.values()) == 0


def test_dropout_mask_generator():
    """stair_dropout_fwd: kept fraction ~ 1 - p, kept values scaled by 1/(1-p), dropped ones exactly 0, the mask a pure
    function of (seed, site, element) -- same call twice gives the same tensor, another seed or site another mask --
    and row gathering through gidx touches only the listed rows."""
    import ctypes as C
    from stair_amd._lib import lib, check
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(300, 512, device=DEV) + 3.0                       # no zeros in the input
    def run(p, seed, site, idx=None, groups=300):
        y = x.clone()
        check(lib.stair_dropout_fwd(C.c_void_p(y.data_ptr()), 512, C.c_void_p(idx.data_ptr()) if idx is not None else None, groups, 512,
                                    C.c_float(p), C.c_uint64(seed), C.c_uint32(site), stream))
        return y
    for p in (0.25, 0.5, 0.9):
        y = run(p, 7, 3)
        kept = y != 0
        assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
        assert torch.allclose(y[kept], x[kept] / (1 - p), rtol=1e-6)
        assert abs(float(y.mean()) / float(x.mean()) - 1.0) < 0.03      # expectation preserved
    a, b = run(0.25, 7, 3), run(0.25, 7, 3)
    assert torch.equal(a, b)
    assert not torch.equal(a != 0, run(0.25, 8, 3) != 0) and not torch.equal(a != 0, run(0.25, 7, 4) != 0)
    assert torch.equal(run(0.0, 7, 3), x)
    idx = torch.tensor([5, 299, 17], dtype=torch.int32, device=DEV)
    g = run(0.5, 1, 0, idx, 3)
    untouched = [i for i in range(300) if i not in (5, 299, 17)]
    assert torch.equal(g[untouched], x[untouched]) and float((g[[5, 299, 17]] == 0).float().mean()) > 0.4


def test_dropout_draws_one_mask_per_question_even_when_questions_share_everything():
    """module_net.py:100-106 evaluates every node of every question, so under model.train() two questions with the same
    program about the same clip still see different dropout masks.  A plan that shared their nodes (the default for
    dropout-free plans) would drop them identically: with dropout on, the plan is built without sharing."""
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    model = _model(config, 2)
    base = synth.make_question(config, 3, 0, form='P3')          # Superlative(max, FilterFrame(video, actions), video): clip-only
    qs = [dict(base, qa_id='a'), dict(base, qa_id='b')]
    for q in qs:
        q['video_features'] = base['video_features']             # the same clip object: one encoder pass
    plain = model.forward_batch(qs, train=True)
    assert plain.info.n_aliased > 0 and torch.equal(plain.logits[0], plain.logits[1])
    dropped = model.forward_batch(qs, train=True, dropout=(0.25, 11))
    assert dropped.info.n_aliased == 0
    assert not torch.equal(dropped.logits[0], dropped.logits[1])
    again = model.forward_batch(qs, train=True, dropout=(0.25, 11))
    assert torch.equal(again.logits, dropped.logits)


def test_dropout_training_forward_and_gradients():
    """Training-mode dropout at the reference's `D` positions (stair_plan_set_dropout).  torch's masks cannot be matched,
    so what is checked: (a) p = 0 and inference are the pinned arithmetic, bit for bit; (b) a seed fixes the step
    (replay is bit-identical), another seed changes it; (c) with the masks held fixed the loss is an ordinary function
    of the weights, and stair_plan_backward is its gradient: directional derivative by central differences along a
    random direction over ALL parameters, exact-fp32 products, every program form in the batch."""
    from stair_amd import ops
    ops.set_matmul_mode('f32')
    try:
        z, meta = load_golden('tiny_conv')
        config = meta['config']
        qs = [question_for(meta, q) for q in meta['questions']]
        model = _model(config, meta['seed'])
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        progs, spans, video, question, q_lens, answers = _pack(model, qs)
        plain = model.run_programs(progs, spans, video, question, q_lens, train=True).logits.clone()
        assert torch.equal(model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=(0.0, 5)).logits, plain)
        d1 = model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=(0.25, 5)).logits.clone()
        d2 = model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=(0.25, 5)).logits.clone()
        d3 = model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=(0.25, 6)).logits.clone()
        assert torch.equal(d1, d2) and not torch.equal(d1, d3) and not torch.equal(d1, plain)
        assert float((d1 - plain).abs().max()) < 5.0                   # a perturbation, not garbage
        with pytest.raises(ValueError):
            model.run_programs(progs, spans, video, question, q_lens, dropout=(0.25, 5))

        ans = answers.long()
        for drop in ((0.25, 11), None):                                 # None calibrates the method on the pinned path
            def loss_at():
                lg = model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=drop).logits
                return float(torch.nn.functional.cross_entropy(lg.double(), ans, reduction='mean'))
            for p in model.parameters():
                p.grad.zero_()
            res = model.run_programs(progs, spans, video, question, q_lens, train=True, dropout=drop)
            res.backward(answers, 1.0 / len(qs))
            # one check per module: step along THAT module's gradient (largest signal over fp32 noise), so that a wrong
            # factor at any single dropout site shows up instead of drowning in the other modules' gradient
            groups = {}
            for n, p in model.named_parameters():
                groups.setdefault(n.split('.')[1], []).append((n, p))
            total = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters()) ** 0.5
            checked = 0
            for gname, plist in sorted(groups.items()):
                gn = sum(float((p.grad.double() ** 2).sum()) for _, p in plist) ** 0.5
                if gn < 0.02 * total:
                    continue                                             # too little signal for a finite difference in fp32
                wn = sum(float((p.detach().double() ** 2).sum()) for _, p in plist) ** 0.5
                direction = {n: p.grad.clone() / gn for n, p in plist}
                eps = 5e-4 * wn                                          # 2e-3 already bends the loss by up to 8 %
                with torch.no_grad():
                    for n, p in plist:
                        p.add_(direction[n], alpha=eps)
                    up = loss_at()
                    for n, p in plist:
                        p.add_(direction[n], alpha=-2 * eps)
                    down = loss_at()
                    for n, p in plist:
                        p.add_(direction[n], alpha=eps)
                numeric = (up - down) / (2 * eps)
                assert abs(numeric - gn) < 0.03 * gn, (drop, gname, numeric, gn)
                checked += 1
            assert checked >= 6, checked
    finally:
        ops.set_matmul_mode('bf16x3')


WORD2ID = {'cup': 'o1', 'glass': 'o1', 'dish': 'o2', 'door': 'o5', 'phone': 'o3', 'sofa': 'o9', 'blanket': 'o4',
           'window': 'o7', 'food': 'o8', 'bag': 'o6'}


def test_filterframe_loss_kernel_matches_reference_fixture():
    """stair_loss_filterframe against tests/golden/criteria_filterframe.npz (the reference's CriterionByModule):
    with head W = [I | 0], b = 0 the head's logits ARE the fixture's predictions, so the kernel's loss and the first O
    columns of d_map must equal the reference's loss and d loss / d pred; then a random head vs autograd of the oracle."""
    import ctypes as C, json, os
    from oracle import nmn_losses as OL
    from stair_amd import losses as L
    from stair_amd._lib import lib, check
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'criteria_filterframe.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    O, H = meta['O'], 64
    index = L.object_index(meta['word2id'])
    assert index == meta['word2index']
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, case in enumerate(meta['cases']):
        T = case['T']
        pred = torch.from_numpy(z['c%d/pred' % i])
        gold = {k: tuple(v) for k, v in case['gold'].items()}
        tgt = torch.from_numpy(L.filterframe_target(gold, T, O, index)).unsqueeze(0).to(DEV)
        x = torch.zeros(3, T, H); x[1, :, :O] = pred                      # the item reads map slot 1
        W = torch.zeros(O, H); W[:, :O] = torch.eye(O)
        xd, Wd, bd = x.to(DEV), W.to(DEV), torch.zeros(O, device=DEV)
        dmap, dW, db = torch.zeros_like(xd), torch.zeros_like(Wd), torch.zeros_like(bd)
        slot = torch.tensor([1], dtype=torch.int32, device=DEV)
        loss = torch.empty(1, device=DEV)
        check(lib.stair_loss_filterframe(p(xd), p(dmap), p(slot), p(tgt), p(Wd), p(bd), p(dW), p(db), 1, T, H, O,
                                         C.c_float(1.0), p(loss), stream))
        assert float(loss) == pytest.approx(float(z['c%d/loss' % i]), abs=2e-6)
        assert float((dmap[1, :, :O].cpu() - torch.from_numpy(z['c%d/dpred' % i])).abs().max()) < 2e-6
        assert float(dmap[0].abs().max()) == 0.0 and float(dmap[2].abs().max()) == 0.0
        # values only (NULL gradients), as the validation loop calls it
        loss2 = torch.empty(1, device=DEV)
        check(lib.stair_loss_filterframe(p(xd), None, p(slot), p(tgt), p(Wd), p(bd), None, None, 1, T, H, O,
                                         C.c_float(0.0), p(loss2), stream))
        assert float(loss2) == float(loss)
        # the same clip stored at a larger stride (a batch that mixes clip lengths): len = the clip's own frames; garbage rows behind it
        Tp = T + 7
        xp = torch.full((3, Tp, H), 5.0); xp[1, :T] = x[1]
        tp = torch.full((1, Tp, O), 0.3); tp[0, :T] = tgt[0].cpu()
        xpd, tpd = xp.to(DEV), tp.to(DEV)
        dmap_p, dW_p, db_p = torch.zeros_like(xpd), torch.zeros_like(Wd), torch.zeros_like(bd)
        loss3 = torch.empty(1, device=DEV)
        len_d = torch.tensor([T], dtype=torch.int32, device=DEV)
        check(lib.stair_loss_filterframe_len(p(xpd), p(dmap_p), p(slot), p(tpd), p(Wd), p(bd), p(dW_p), p(db_p), p(len_d), 1, Tp, H, O,
                                             C.c_float(1.0), p(loss3), stream))
        assert float(loss3) == float(loss)
        assert torch.equal(dmap_p[1, :T], dmap[1]) and float(dmap_p[1, T:].abs().max()) == 0.0
        assert float((dW_p - dW).abs().max()) < 1e-6 and float((db_p - db).abs().max()) < 1e-6
    # random head, two items sharing a slot, scale != 1: vs torch autograd of the oracle criterion
    g = torch.Generator().manual_seed(3)
    T = 40
    x = torch.randn(4, T, H, generator=g).requires_grad_(True)
    W = (torch.randn(O, H, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(O, generator=g).requires_grad_(True)
    golds = [{'cup': (3.2, 17.9), 'door': (10.0, 30.5)}, {'phone': (0.0, 40.0)}, {'sofa': (5.5, 6.5), 'bag': (20.0, 21.0)}]
    slots = [2, 0, 2]
    total = sum(OL.criterion_filterframe(x[s] @ W.t() + b, gd, index) for s, gd in zip(slots, golds)) * 0.25
    total.backward()
    tgt = torch.from_numpy(np.stack([L.filterframe_target(gd, T, O, index) for gd in golds])).to(DEV)
    xd, Wd, bd = x.detach().to(DEV), W.detach().to(DEV), b.detach().to(DEV)
    dmap, dW, db = torch.zeros_like(xd), torch.zeros_like(Wd), torch.zeros_like(bd)
    loss = torch.empty(3, device=DEV)
    check(lib.stair_loss_filterframe(p(xd), p(dmap), p(torch.tensor(slots, dtype=torch.int32, device=DEV)), p(tgt), p(Wd), p(bd),
                                     p(dW), p(db), 3, T, H, O, C.c_float(0.25), p(loss), stream))
    for got, ref in ((dmap, x.grad), (dW, W.grad), (db, b.grad)):
        assert float((got.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_training_with_filterframe_supervision(matmul):
    """The FilterFrame criterion switched ON (no_intermediate=()): window loss and every parameter gradient vs
    autograd of the oracle, whose criterion is pinned to the reference by criteria_filterframe.npz."""
    from oracle import nmn_losses as OL
    from stair_amd import losses as L
    z, meta = load_golden('tiny_conv')
    config, T = meta['config'], meta['T']
    qs = _with_gold(config, 3, [question_for(meta, q) for q in meta['questions']], T)
    index = L.object_index(WORD2ID)
    words = sorted(WORD2ID)
    n_ff = 0
    for qi, q in enumerate(qs):
        for i, tok in enumerate(q['nmn_program_list']):
            if tok == 'FilterFrame' and i != 0 and q['nmn_program_idx'][i] is not None:
                a, b = words[(qi + i) % len(words)], words[(qi + 2 * i + 3) % len(words)]
                q['sg_res_by_step'][q['nmn_program_idx'][i]] = {a: (0.1 * T, 0.5 * T), b: (0.3 * T + qi, 0.9 * T)}
                n_ff += 1
    assert n_ff >= 3
    Gw = len(qs)
    names, w = _oracle_params(config, meta['seed'])
    total, det = OL.window_loss(w, config, [_oracle_view(q) for q in qs], L.CRITERION_MODULES, gradient_accumulation=Gw,
                                no_intermediate=(), explicit_lstm=True, word2index=index)
    total.backward()
    model = _model(config, meta['seed'])
    model.pretrain_modules = set(L.CRITERION_MODULES)
    model.object_index = index
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    progs, spans, video, question, q_lens, answers = _pack(model, qs)
    res = model.run_programs(progs, spans, video, question, q_lens, train=True)
    res.zero_grad_arenas()
    losses, extra = L.apply_module_losses(model, res, qs, 1.0 / Gw, no_intermediate=())
    res.backward(answers, 1.0 / Gw, keep_arenas=True)
    ref_ff = sorted(x[3] for x in det['module'] if x[2] == 'FilterFrame')
    assert len(ref_ff) == n_ff and np.allclose(sorted(losses['FilterFrame'].cpu().tolist()), ref_ff, rtol=2e-5, atol=2e-6)
    assert 'submodules.FilterFrame.pretrain_head.weight' in extra
    got = dict(model.named_parameters())
    for n in names:
        ref = w[n].grad
        if ref is None:
            continue
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3)
        assert float((got[n].grad.cpu() - ref).abs().max()) < tol, n
    assert float(w['submodules.FilterFrame.pretrain_head.weight'].grad.abs().max()) > 0
    model.object_index = None
    with pytest.raises(RuntimeError, match='object_index'):
        L.apply_module_losses(model, res, qs, 1.0 / Gw, no_intermediate=())


def test_validation_loop_matches_reference(matmul):
    """stair_amd.evaluate.evaluate_by_module on the HIP path == the reference's evaluate_by_module output
    (tests/golden/validation.json): accuracy, every module's mean validation loss ('cont-valid' cosine for the
    contrastive modules, attention criteria, head criteria, decoder CE); modules with no scored node report inf."""
    import json, os
    from stair_amd import evaluate as E
    from stair_amd.module_net import VideoNMN
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'validation.json')))
    z, meta = load_golden(gold['config'])
    config = meta['config']
    from helpers import PRETRAIN_MODULES
    m = VideoNMN(config, pretrain_modules=set(PRETRAIN_MODULES))
    w = synth.make_weights(config, meta['seed'])
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    m = m.to(DEV)
    qs = []
    for q in meta['questions']:
        d = question_for(meta, q)
        d['sg_res_by_step'] = synth.make_gold(config, meta['seed'], d, T=meta['T'], keep=1.0)
        qs.append(d)
    acc, valid = E.evaluate_by_module(m, qs, gold['unk_token_id'], batch_size=5)
    assert acc == pytest.approx(gold['accuracy'])
    for module, ref in gold['valid_losses'].items():
        if ref is None:
            assert valid[module] == float('inf'), module
        else:
            assert valid[module] == pytest.approx(ref, rel=1e-4, abs=2e-5), module


@pytest.mark.parametrize('ragged,supervised', [(False, True), (True, False)])
def test_training_step_reads_no_uninitialised_workspace(ragged, supervised):
    """Forward saves, gradient arenas, scratch: everything the backward pass reads was written in the same step.  With the
    (reused) workspace filled with NaN / 1e30 before the step, the gradients carry no NaN and differ from the clean step only by
    the order of the fp32 atomics (relative L2 < 1e-5)."""
    from stair_amd import losses as L
    from stair_amd.module_net import VideoNMN
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG)
    w = synth.make_weights(config, 4)

    def grads(poison):
        model = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
        model.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
        model = model.to(DEV)
        tr = Trainer(model, dropout=0.0, lr=0.0)
        qs = synth.make_questions(config, 31, 48, forms=synth.ALL_FORMS)
        for q in qs:
            sg = synth.make_gold(config, 0, q, T=64)
            q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v) for k, v in sg.items()}
        video = torch.stack([torch.as_tensor(q['video_features']) for q in qs]).to(DEV)
        vl = None
        if ragged:
            vl = [64 - (i * 7) % 40 for i in range(len(qs))]
            for i, l in enumerate(vl):
                video[i, l:] = 0
        q_lens = [q['question'].shape[0] for q in qs]
        question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
        answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
        progs = [q['nmn_program_list'] for q in qs]
        spans = [q['prog_str_to_question_tokens'] for q in qs]
        kw = dict(questions=qs if supervised else None, video_len=vl)
        tr.step(progs, spans, video, question, q_lens, answers, **kw)        # allocates the workspace
        if poison is not None:
            model._ws.fill_(poison)
        tr.step(progs, spans, video, question, q_lens, answers, **kw)
        torch.cuda.synchronize()
        return tr.flat_g.clone()

    ref = grads(None)
    for poison in (float('nan'), 1e30):
        g = grads(poison)
        assert int(g.isnan().sum()) == 0, poison
        assert float((g - ref).norm() / ref.norm()) < 1e-5, poison


def test_trainer_windows_match_reference_main_loop(matmul):
    """L2 pinned to the reference itself: tests/golden/window.npz holds what /root/reference/train_module.py main()
    produced on CPU for TWO windows of 32 questions (per-module criteria :351-373, decoder CE :376-380, the pooled
    contrastive pass :388-406, one backward :408, Adam :326/:410, LambdaLR :328-332/:412).  Trainer.step on the same
    questions must give the same criterion values and the same weights after each optimizer step."""
    from helpers import window_fixture, window_weights
    from stair_amd import losses as L
    from stair_amd.train import Trainer
    z, meta, qs, records = window_fixture()
    config, W = meta['config'], meta['window']
    model = _model(config, meta['seed'])
    model.pretrain_modules = set(L.CRITERION_MODULES)
    tr = Trainer(model, lr=meta['lr'], scheduler_total_iters=meta['scheduler_total_iters'], contrastive_window=W,
                 skip_untouched='window', dropout=0.0)          # 'window': the fixture was made with torch >= 2 (grads set to None)
    init = synth.make_weights(config, meta['seed'])
    names = [n for n, _ in spec.weight_table(config)]
    for wi, rec in enumerate(records):
        window = qs[wi * W:(wi + 1) * W]
        progs, spans, video, question, q_lens, answers = _pack(model, window)
        dec, _ = tr.step(progs, spans, video, question, q_lens, answers, questions=window)
        assert np.allclose(dec.cpu().numpy(), rec['decoder'], rtol=1e-5, atol=2e-5)
        got_mod = sorted(torch.cat([v for k, v in tr.module_losses.items() if k != 'contrastive']).cpu().tolist())
        ref_mod = sorted(v for _, v in rec['module'])
        assert len(got_mod) == len(ref_mod) and np.allclose(got_mod, ref_mod, rtol=2e-5, atol=2e-6)
        got_c, ref_c = sorted(tr.module_losses['contrastive'].cpu().tolist()), sorted(v for _, v in rec['contrastive'])
        assert len(got_c) == len(ref_c) > 10 and np.allclose(got_c, ref_c, rtol=2e-5, atol=2e-6)
        assert abs(tr.lr * tr.lr_factor() - meta['lr_after_window'][wi]) < 1e-12
        got = dict(model.named_parameters())
        worst, moved = (0.0, ''), 0
        for n in names:
            ref, g = window_weights(z, meta, wi + 1, n, got[n])
            diff = np.abs(g - ref)
            # Adam moves every touched weight by ~lr = 2e-4 per step whatever the gradient's scale; see
            # test_trainer_steps_match_torch_adam for why the split-product mode is stated as a fraction
            if matmul == 'f32':
                assert diff.max() < 2e-5, (wi, n, diff.max())
            else:
                assert (diff < 2e-5).mean() > 0.995, (wi, n, (diff < 2e-5).mean())
                assert diff.max() < 2.5e-4 * (wi + 1), (wi, n, diff.max())
            worst = max(worst, (float(diff.max()), n))
            i0 = np.asarray(init[n], dtype=np.float64).reshape(-1)
            i0 = i0 if i0.size <= meta['large_threshold'] else i0[::meta['stride_large']]
            if np.abs(ref - i0).max() == 0:                          # never reached by a loss (FilterFrame's head, args.py:62):
                assert np.array_equal(g, i0), n                      # Adam must skip it, bit for bit
            else:
                moved += 1
        assert moved > 90
        print('window %d (%s): largest weight difference to the reference %.3g in %s' % (wi + 1, matmul, worst[0], worst[1]))


def test_contrastive_pools_from_a_class_table_equal_the_pooled_lists(matmul):
    """stair_loss_contrastive_table (pools = rows of a [windows, classes] presence matrix over a table of all classes) against
    stair_loss_contrastive on the explicitly pooled class lists (train_module.py:388-406): same loss per item, same gradients."""
    from stair_amd import losses as L
    z, meta = load_golden('tiny_conv')
    config, T = meta['config'], meta['T']
    qs = _with_gold(config, 3, [question_for(meta, q) for q in meta['questions']] +
                    [synth.make_question(config, 8, 50 + i, form=f, T=T) for i, f in enumerate(synth.ALL_FORMS)], T)
    table = L.ClassTable.from_questions(qs + [{'sg_res_by_step': {0: [('zz_unused', np.ones((2, config['text_size']), np.float32))]}}])
    out = {}
    for mode in ('lists', 'table'):
        model = _model(config, meta['seed'])
        model.pretrain_modules = set(L.CRITERION_MODULES)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        progs, spans, video, question, q_lens, answers = _pack(model, qs)
        res = model.run_programs(progs, spans, video, question, q_lens, train=True)
        res.zero_grad_arenas()
        losses, _ = L.apply_module_losses(model, res, qs, 1.0 / len(qs), window=7, class_table=table if mode == 'table' else None)
        res.backward(answers, 1.0 / len(qs), keep_arenas=True)
        out[mode] = (losses['contrastive'].cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
    assert out['lists'][0].numel() > 10
    assert torch.allclose(out['lists'][0], out['table'][0], rtol=1e-5, atol=1e-6)
    for n, g in out['lists'][1].items():
        assert float((g - out['table'][1][n]).abs().max()) <= 2e-5 * max(float(g.abs().max()), 1e-3), n


@pytest.mark.parametrize('n_q,clips', [(64, 32), (300, 300)])
def test_training_step_is_bit_reproducible(n_q, clips):
    """The reference's CPU loop is deterministic (train_module.py:341-412); so is this step: two fresh trainers, the same two
    windows -- all twelve program forms, full hidden size, stored-bf16 clips, two questions per clip in the first case (shared
    clips and common subexpressions: several readers per gradient slot) -- must end with bit-identical gradient buckets and
    bit-identical weights after Adam.  What makes it so: gradient fan-in through staging slots added in a fixed order
    (stair_plan_info.n_*_stage), weight-gradient partials stored and reduced in slab order, and 64-bit fixed-point shadows
    under every remaining many-to-one float atomic (csrc/common.h det_shadow)."""
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG)
    qs = [synth.make_question(config, 21, i, T=64, forms=synth.ALL_FORMS, with_video=False) for i in range(n_q)]
    g = torch.Generator().manual_seed(9)
    video = torch.randn(clips, 64, config['video_size'], generator=g).to(torch.bfloat16).to(DEV)
    vidx = [i % clips for i in range(n_q)] if clips != n_q else None
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    q_lens = [q['question'].shape[0] for q in qs]
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    progs, spans = [q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs]
    runs = []
    for rep in range(2):
        tr = Trainer(_model(config, 5), dropout=0.0, lr=1e-3)
        grads = []
        for it in range(2):
            _, res = tr.step(progs, spans, video, question, q_lens, answers, video_index=vidx)
            grads.append(tr.flat_g.clone())
        tr.check()
        runs.append((grads, tr.flat_p.clone(), res.info))
    inf = runs[0][2]
    if vidx is not None:
        assert inf.n_aliased > 0 and inf.n_vec_stage > 0 and inf.n_map_stage > 0       # the case is not trivially free of fan-in
    for it in range(2):
        assert torch.equal(runs[0][0][it], runs[1][0][it]), 'gradient bucket of step %d differs between two runs' % it
    assert torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][0][0].abs().max()) > 0


def test_dropout_step_is_bit_reproducible_and_runs_fused():
    """The recipe the reference trains with (nn.Dropout(0.25), args.py:31): the masks are a pure function of (seed, site, element) drawn
    inside the fused tile operators and the grouped vector-level launches, so the step replays bit for bit on two fresh trainers -- and it
    must really be running fused (the launch-per-layer forms were the only ones a dropout plan could take before ABI 6)."""
    from stair_amd import ops
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG)
    n_q = 96
    qs = [synth.make_question(config, 22, i, T=64, forms=synth.ALL_FORMS, with_video=False) for i in range(n_q)]
    g = torch.Generator().manual_seed(11)
    video = torch.randn(n_q, 64, config['video_size'], generator=g).to(torch.bfloat16).to(DEV)
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    q_lens = [q['question'].shape[0] for q in qs]
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    progs, spans = [q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs]
    runs = []
    for rep in range(2):
        tr = Trainer(_model(config, 5), dropout=0.25, lr=1e-3)
        with ops.kernel_accounting() as acct:
            for it in range(2):
                tr.step(progs, spans, video, question, q_lens, answers)
        tr.check()
        runs.append((tr.flat_g.clone(), tr.flat_p.clone(), dict(acct.table)))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][0].abs().max()) > 0
    assert 'tile_mlp' in runs[0][2] and 'vec_group' in runs[0][2], sorted(runs[0][2])
    # another step draws other masks: the second step's gradients differ from a trainer that repeats the first seed
    plain = Trainer(_model(config, 5), dropout=0.0, lr=1e-3)
    plain.step(progs, spans, video, question, q_lens, answers)
    plain.step(progs, spans, video, question, q_lens, answers)
    assert not torch.equal(plain.flat_g, runs[0][0])


@pytest.mark.parametrize('dropout', [0.0, 0.25])
def test_supervised_step_is_bit_reproducible(dropout):
    """BASELINE configs[4]: the step with every per-module criterion (train_module.py:33-194, 351-406) is bit-identical from run to
    run as well -- 64 questions on 32 shared clips (aliased supervised nodes: several criteria items per gradient slot).  The criteria
    add into the arenas group by group (stair_loss_groups) and into the head weights through the fixed-point shadows
    (stair_grad_shadows_begin), the LDS sums of the criteria run in wave order.  dropout 0.25: the reference's whole recipe (args.py:31);
    its plan runs the grouped vector-level launches too since ABI 6 -- on the launch-per-layer forms this step was not reproducible."""
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG)
    n_q, clips = 64, 32
    qs = [synth.make_question(config, 23, i, T=64, forms=synth.ALL_FORMS, with_video=False) for i in range(n_q)]
    qs = _with_gold(config, 23, qs, 64)
    g = torch.Generator().manual_seed(10)
    video = torch.randn(clips, 64, config['video_size'], generator=g).to(torch.bfloat16).to(DEV)
    vidx = [i % clips for i in range(n_q)]
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    q_lens = [q['question'].shape[0] for q in qs]
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    progs, spans = [q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs]
    runs = []
    for rep in range(2):
        tr = Trainer(_model(config, 5), dropout=dropout, lr=1e-3)
        grads = []
        for it in range(2):
            tr.step(progs, spans, video, question, q_lens, answers, questions=qs, video_index=vidx)
            grads.append(tr.flat_g.clone())
        tr.check()
        runs.append((grads, tr.flat_p.clone(), {k: v.clone() for k, v in tr.module_losses.items()}))
    for it in range(2):
        assert torch.equal(runs[0][0][it], runs[1][0][it]), 'gradient bucket of supervised step %d differs between two runs' % it
    assert torch.equal(runs[0][1], runs[1][1])
    assert set(runs[0][2]) >= {'attention', 'contrastive'} and any(k in runs[0][2] for k in ('Exists', 'Xor', 'Equals'))    # every criterion family ran
    for k, v in runs[0][2].items():
        assert torch.equal(v, runs[1][2][k]), k


def test_collated_gold_batch_gives_the_step_of_the_question_dicts():
    """losses.collate_gold (the loader's collate step: the batch's gold intermediates as flat arrays) against the list of question
    dicts: the same index / target arrays reach the same loss kernels -- every criterion value and every parameter gradient bit for
    bit --, with pooled class lists and with a class table; GoldBatch.select drops the gold of the masked questions."""
    from stair_amd import losses as L
    z, meta = load_golden('tiny_conv')
    config, T = meta['config'], meta['T']
    qs = _with_gold(config, 3, [question_for(meta, q) for q in meta['questions']] +
                    [synth.make_question(config, 8, 50 + i, form=f, T=T) for i, f in enumerate(synth.ALL_FORMS)], T)
    table = L.ClassTable.from_questions(qs)
    for tab in (None, table):
        out = {}
        for mode in ('dicts', 'collated'):
            model = _model(config, meta['seed'])
            model.pretrain_modules = set(L.CRITERION_MODULES)
            for p in model.parameters():
                p.grad = torch.zeros_like(p)
            progs, spans, video, question, q_lens, answers = _pack(model, qs)
            res = model.run_programs(progs, spans, video, question, q_lens, train=True)
            res.zero_grad_arenas()
            gold = qs if mode == 'dicts' else L.collate_gold(qs, class_table=tab)
            losses, _ = L.apply_module_losses(model, res, gold, 1.0 / len(qs), window=7, class_table=tab)
            out[mode] = ({k: v.cpu().clone() for k, v in losses.items()},
                         [res.grad_arena(k).clone() for k in ('vec', 'att')])
        assert set(out['dicts'][0]) == set(out['collated'][0]) and len(out['dicts'][0]) >= 3
        for k, v in out['dicts'][0].items():
            assert torch.equal(v, out['collated'][0][k]), k
        # (the loss kernels add into the arenas with float atomics: equal up to the order of the sums)
        for a, b in zip(out['dicts'][1], out['collated'][1]):
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    gb = L.collate_gold(qs, class_table=table)
    keep = [i % 2 == 0 for i in range(len(qs))]
    half = gb.select(keep)
    ref = L.collate_gold([q if k else dict(q, sg_res_by_step={}) for q, k in zip(qs, keep)], class_table=table)
    assert np.array_equal(half.att_q, ref.att_q) and np.array_equal(half.att_pos, ref.att_pos) and np.array_equal(half.att_iv, ref.att_iv)
    assert half.cg_name == ref.cg_name and np.array_equal(half.cg_cls, ref.cg_cls) and set(half.head) == set(ref.head)
    for m in half.head:
        assert all(np.array_equal(a, b) for a, b in zip(half.head[m], ref.head[m]))


@pytest.mark.parametrize('name', ['tiny_conv', 'full'])
def test_common_subexpression_sharing_keeps_values_and_gradients(name, matmul):
    """Clip-level common subexpressions computed once (stair_plan_build: a node of clip-only operands is aliased by every later
    occurrence -- other questions about the clip, or the same sub-program twice in one question, as P0 has it) against the
    plan that computes every node (STAIR_PLAN_NO_CSE, i.e. what module_net.py:100-106 does): same logits, same per-question
    loss, same parameter gradients -- the users' gradients add up in the shared slot -- incl. intermediate supervision."""
    from stair_amd import losses as L
    if name == 'full':
        config, T = dict(spec.DEFAULT_CONFIG), 64
        if matmul == 'f32':
            pytest.skip('one mode is enough at full size')
    else:
        z, meta = load_golden(name)
        config, T = meta['config'], meta['T']
    forms = synth.ALL_FORMS * 2 + ['P1', 'P3', 'P4', 'P7', 'P3', 'P1']
    qs = [synth.make_question(config, 12, i, form=f, T=T) for i, f in enumerate(forms)]
    clips = [torch.as_tensor(qs[c]['video_features']) for c in range(4)]
    for i, q in enumerate(qs):
        q['video_features'] = clips[i % 4]
    qs = _with_gold(config, 5, qs, T)
    out = []
    for cse in (False, True):
        model = _model(config, 7)
        model.pretrain_modules = set(L.CRITERION_MODULES)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        res = model.forward_batch(qs, train=True, cse=cse)
        assert (res.info.n_aliased > 20) == cse
        res.zero_grad_arenas()
        losses, _ = L.apply_module_losses(model, res, qs, 1.0 / len(qs), window=8)
        dec = res.backward(torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV), 1.0 / len(qs), keep_arenas=True)
        out.append((res.logits.cpu().clone(), dec.cpu().clone(), {k: v.cpu().clone() for k, v in losses.items()},
                    {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}, res.info.n_map, res.info.n_vec))
    (l0, d0, m0, g0, nm0, nv0), (l1, d1, m1, g1, nm1, nv1) = out
    assert nm1 < nm0 and nv1 < nv0
    assert float((l0 - l1).abs().max()) < 2e-6 and torch.allclose(d0, d1, rtol=1e-6, atol=1e-6)
    for k in m0:
        assert torch.allclose(torch.sort(m0[k]).values, torch.sort(m1[k]).values, rtol=1e-5, atol=1e-6), k
    gmax = max(float(g.abs().max()) for g in g0.values())
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) < (2e-5 if matmul == 'f32' else 2e-4) * max(float(g0[n].abs().max()), 1e-3 * gmax), n
