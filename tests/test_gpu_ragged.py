"""Clips of different frame counts in ONE launch batch (stair_plan_build_ragged): real I3D .npy clips keep their own
length and are only truncated above max_video_length (/root/reference/video_nmn/dataset.py:137-143), and the reference runs
each question at its clip's length.  Every question of a mixed batch must equal the oracle run on that question alone --
logits at 1e-4 with the same top-1, every parameter gradient of a training step -- for T in {3, 17, 40, 64} and all
program forms (Conv1d-Temporal configuration; Linear(T,T) nets fix T like the reference)."""
import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import spec, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LENS = (3, 17, 40, 64)


def _maxerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def _setup(config, seed, n, bf16=False):
    from stair_amd.module_net import VideoNMN
    weights = synth.make_weights(config, seed)
    model = VideoNMN(config)
    model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
    qs = [synth.make_question(config, seed, i, form=synth.ALL_FORMS[i % len(synth.ALL_FORMS)], T=LENS[(i // 3) % len(LENS)]) for i in range(n)]
    if bf16:
        for q in qs:
            q['video_features'] = torch.as_tensor(q['video_features']).to(torch.bfloat16)
    return model.to(DEV), weights, qs


def _oracle_q(q):
    v = q['video_features']
    return dict(q, video_features=v.float() if isinstance(v, torch.Tensor) else v)


@pytest.mark.parametrize('size,bf16', [('tiny', False), ('tiny', True), ('full', True)])
def test_mixed_clip_lengths_forward_matches_oracle_per_question(size, bf16):
    if size == 'tiny':
        config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=64, object_types=10)
        n = 36
    else:
        config, n = dict(spec.DEFAULT_CONFIG), 24
    model, weights, qs = _setup(config, 4, n, bf16)
    res = model.forward_batch(qs)
    assert res.question_frames is not None and sorted(set(int(x) for x in res.question_frames)) == sorted(LENS)
    logits, pred = res.logits.cpu(), res.pred.cpu()
    w = O.to_torch(weights)
    for i, q in enumerate(qs):
        ref = O.forward(w, config, _oracle_q(q), return_res_by_step=False)['logits']
        assert _maxerr(logits[i], ref) < 1e-4, (q['form'], q['video_features'].shape[0], _maxerr(logits[i], ref))
        assert int(pred[i]) == int(torch.argmax(ref)), q['form']
    # the same questions one length at a time give the same logits (padding changes nothing)
    for L in LENS:
        idx = [i for i, q in enumerate(qs) if q['video_features'].shape[0] == L]
        alone = model.forward_batch([qs[i] for i in idx]).logits.cpu()
        assert _maxerr(alone, logits[idx]) < 2e-6, L


def test_mixed_clip_lengths_shared_clips():
    """Several questions per clip AND several clip lengths."""
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=64, object_types=10)
    model, weights, qs = _setup(config, 9, 24)
    for i in range(0, 24, 3):                       # questions 3k, 3k+1, 3k+2 share the clip of 3k (same length by construction)
        qs[i + 1]['video_features'] = qs[i + 2]['video_features'] = qs[i]['video_features']
    res = model.forward_batch(qs)
    assert res._video.shape[0] == 8
    w = O.to_torch(weights)
    for i, q in enumerate(qs):
        ref = O.forward(w, config, q, return_res_by_step=False)['logits']
        assert _maxerr(res.logits[i], ref) < 1e-4


GRAD_SEED = 1      # (a seed at which no ReLU pre-activation of the batch falls inside the split products' rounding band of zero:
#                     a flipped mask changes a whole row of that layer's weight gradient and everything upstream of it, in ANY
#                     batch, ragged or not -- tools/ and DESIGN.md section 4; the exact-f32 mode has no such band)


@pytest.mark.parametrize('bf16,mode', [(False, 'f32'), (False, 'bf16x3'), (True, 'bf16x3')])
def test_mixed_clip_lengths_training_gradients(bf16, mode):
    from stair_amd import ops
    ops.set_matmul_mode(mode)
    try:
        _training_gradients(bf16, GRAD_SEED)
    finally:
        ops.set_matmul_mode('bf16x3')


def _training_gradients(bf16, seed):
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=64, object_types=10)
    model, weights, qs = _setup(config, seed, 24, bf16)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    res = model.forward_batch(qs, train=True)
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    losses = res.backward(answers, 1.0 / len(qs))
    names = [n for n, _ in spec.weight_table(config)]
    wt = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    loss, per_q = 0, []
    for q in qs:
        lg = O.forward(wt, config, _oracle_q(q), return_res_by_step=False, explicit_lstm=True)['logits']
        ce = torch.nn.functional.cross_entropy(lg.unsqueeze(0), torch.tensor([q['answer']]))
        per_q.append(float(ce.detach()))
        loss = loss + ce / len(qs)
    loss.backward()
    assert np.allclose(losses.cpu().numpy(), per_q, rtol=1e-5, atol=1e-5)
    sd = dict(model.named_parameters())
    bad = []
    for n in names:
        ref = wt[n].grad if wt[n].grad is not None else torch.zeros_like(wt[n])
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3)
        if not _maxerr(sd[n].grad, ref) < tol:
            bad.append((n, _maxerr(sd[n].grad, ref), float(ref.abs().max())))
    print('BAD', [(b[0], round(b[1] / max(b[2], 1e-9), 3)) for b in bad])
    assert not bad, len(bad)


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_mixed_clip_lengths_intermediate_losses(mode):
    """Attention criteria on a mixed batch: the gold interval masks and the means run over each clip's own frames.
    Loss values in both arithmetic modes; every parameter gradient elementwise in the exact-f32 mode, by relative L2 in the
    split mode (ReLU masks at the split products' rounding band, see GRAD_SEED)."""
    from stair_amd import ops
    ops.set_matmul_mode(mode)
    try:
        _intermediate_losses(mode)
    finally:
        ops.set_matmul_mode('bf16x3')


def _intermediate_losses(mode):
    from oracle import nmn_losses as OL
    from stair_amd import losses as L
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=64, object_types=10)
    model, weights, qs = _setup(config, 11, 24)
    model.pretrain_modules = set(L.CRITERION_MODULES)
    for q in qs:
        Tq = q['video_features'].shape[0]
        q['sg_res_by_step'] = synth.make_gold(config, 5, q, T=Tq, keep=1.0)
    view = [dict(q, sg_res_by_step={k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v)
                                     for k, v in q['sg_res_by_step'].items()}) for q in qs]
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    names = [n for n, _ in spec.weight_table(config)]
    wt = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    total, det = OL.window_loss(wt, config, view, L.CRITERION_MODULES, gradient_accumulation=len(qs), explicit_lstm=True)
    total.backward()
    res = model.forward_batch(view, train=True)
    res.zero_grad_arenas()
    losses, _ = L.apply_module_losses(model, res, view, 1.0 / len(qs), window=32)
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    res.backward(answers, 1.0 / len(qs), keep_arenas=True)
    ref_mod = sorted(x[3] for x in det['module'])
    got_mod = sorted(torch.cat([v for k, v in losses.items() if k != 'contrastive']).cpu().tolist())
    assert len(ref_mod) == len(got_mod) and np.allclose(got_mod, ref_mod, rtol=2e-5, atol=2e-6)
    sd = dict(model.named_parameters())
    for n in names:
        ref = wt[n].grad
        if ref is None:
            continue
        if mode == 'f32':
            tol = 2e-4 * max(float(ref.abs().max()), 1e-3)
            assert _maxerr(sd[n].grad, ref) < tol, (n, _maxerr(sd[n].grad, ref))
        elif ref.numel() >= 64:
            rel = float((sd[n].grad.cpu() - ref).norm() / ref.norm().clamp_min(1e-12))
            assert rel < 3e-2, (n, rel)
