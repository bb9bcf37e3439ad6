"""GPU tests of the plane GEMMs (csrc/gemm_planes.hip): operands that arrive as bf16 planes, staged by LDS-DMA.
Reference = fp64 matmul of the SAME (already rounded) operands, so the only error left is the dropped lo*lo term and the
fp32 accumulation: tolerance 2e-5 of the row scale, against the 1e-4 logit budget of BASELINE.json."""
import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import spec, synth
from helpers import oracle_weights

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _maxerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def test_split_planes_reconstructs_fp32():
    from stair_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4096 * 8, generator=g) * torch.logspace(-6, 6, 4096 * 8)).to(DEV)
    hi, lo = ops.split_planes(x)
    assert torch.equal(hi, x.to(torch.bfloat16))                       # round to nearest even, like torch
    assert torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    rel = ((hi.float() + lo.float() - x).abs() / x.abs().clamp_min(1e-30)).max()
    assert float(rel) < 2.0 ** -16
    only_hi = ops.split_planes(x, lo=False)
    assert torch.equal(only_hi, hi)


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (300, 256, 96), (1000, 1024, 2048), (256 * 5 + 7, 512, 512), (4096, 300, 320),
                                   (777, 1000, 32 * 9)])
@pytest.mark.parametrize('act', [None, 'relu', 'sigmoid'])
@pytest.mark.parametrize('a_lo', [False, True])
@pytest.mark.parametrize('tiled', [False, True])
def test_gemm_planes_matches_fp64(M, N, K, act, a_lo, tiled):
    from stair_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    w_hi, w_lo = ops.split_planes_tiled(w) if tiled else ops.split_planes(w)
    if a_lo:
        x_hi, x_lo = ops.split_planes(x)
        xe = x.double()
    else:
        x_hi, x_lo = x.to(torch.bfloat16), None
        xe = x_hi.double()                                              # A is exactly the stored bf16 values
    y = ops.gemm_planes(x_hi, x_lo, w_hi, w_lo, b, act)
    ref = xe @ w.double().T + b.double()
    if act == 'relu':
        ref = ref.clamp_min(0)
    elif act == 'sigmoid':
        ref = torch.sigmoid(ref)
    err = float((y.double() - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (300, 256, 128), (1000, 1024, 2048), (256 * 5 + 7, 512, 512), (4096, 320, 320 + 64),
                                   (70000, 2048, 256)])
def test_gemm_planes_with_w_in_fragment_order_matches_fp64(M, N, K):
    """stair_gemm_planes_args.w_tiled == 2 (gemm_planes_wr_kernel): W as ONE fragment-order image (stair_pack_wfrag) that never enters
    LDS -- each wave loads its own fragments global -> VGPR with hand-counted waits, LDS-DMA stages A alone.  Ragged M and N tails,
    several tiles per workgroup (the stage stream crosses tile boundaries), bias; against fp64 of the stored bf16 A."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M * 5 + N + K)
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    wf = ops.pack_wfrag(w)
    y = ops.gemm_planes(x, None, wf, wf, b, None, w_frag_rows=N)
    for lo, hi in ((0, min(M, 600)), (max(0, M - 600), M)):
        ref = x[lo:hi].double() @ w.double().T + b.double()
        err = float((y[lo:hi].double() - ref).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), (M, N, K, lo, err)


def test_split_planes_tiled_is_a_relayout_of_the_plain_split():
    from stair_amd import ops
    w = torch.randn(300, 96, generator=torch.Generator().manual_seed(5)).to(DEV)
    hi, lo = ops.split_planes(w)
    thi, tlo = ops.split_planes_tiled(w)
    assert torch.equal(thi, hi.view(300, 3, 32).permute(1, 0, 2).contiguous())
    assert torch.equal(tlo, lo.view(300, 3, 32).permute(1, 0, 2).contiguous())


@pytest.mark.parametrize('tiled', [False, True])
def test_gemm_planes_many_tiles_per_workgroup(tiled):
    """More tiles than CUs: every workgroup walks several tiles and the LDS-DMA stream crosses tile boundaries."""
    from stair_amd import ops
    M, N, K = 256 * 70 + 33, 1024 + 256, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g).to(DEV)
    w_hi, w_lo = ops.split_planes_tiled(w) if tiled else ops.split_planes(w)
    y = ops.gemm_planes(x, None, w_hi, w_lo)
    ref = x.double() @ w.double().T
    assert float((y.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_gemm_planes_asymmetric_layout():
    """A = I against an asymmetric W: catches a transposed C write or a permuted k order inside the LDS image."""
    from stair_amd import ops
    M = N = K = 256
    eye = torch.eye(M, dtype=torch.bfloat16, device=DEV)
    w = (torch.arange(N * K, dtype=torch.float32, device=DEV).reshape(N, K) % 251) - 125.0
    w_hi, w_lo = ops.split_planes(w)
    y = ops.gemm_planes(eye, None, w_hi, w_lo)
    assert torch.equal(y, w.T.contiguous())


def test_gemm_planes_rejects_bad_shapes():
    from stair_amd import ops
    from stair_amd._lib import StairError
    x = torch.zeros(256, 40, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(256, 40, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(StairError):
        ops.gemm_planes(x, None, w, w)


# ---------------------------------------------------------------------------------------------
# stored bf16 clip features (BASELINE.json configs[1]): the oracle is fed the SAME rounded values (exact in fp32), so the
# 1e-4 logit bar of north_star applies unchanged
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(64, 16, 64), (4096, 512, 512), (1001, 1024, 320), (333, 36, 512), (16384, 1024, 2048)])
def test_gemm_tn_with_exact_bf16_rows(M, N, K):
    """dW = dZ^T X with X stored as bf16 (two products per pair, no lo image of X), plus the bias gradient riding along."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M * 3 + N + K)
    X = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    dZ = torch.randn(M, N, generator=g).to(DEV)
    C0 = torch.randn(N, K, generator=g).to(DEV)
    Cm, b1 = C0.clone(), torch.zeros(N, device=DEV)
    ops.gemm_tn(dZ, X, Cm, M, N, K, rows_per_group=1, colsum=b1)
    ref = C0.double() + dZ.double().t() @ X.double()
    assert _maxerr(Cm, ref) < 4e-4 * max(1.0, (M / 1000) ** 0.5 * 3)
    assert _maxerr(b1, dZ.double().sum(0)) < 4e-5 * max(1.0, (M / 100) ** 0.5)


@pytest.mark.parametrize('M,N,K,wide', [(2048, 256, 256, False), (8192, 1024, 2048, True), (16384, 256, 256, False), (16384, 1024, 2048, True), (32768, 512, 256, True), (17408, 256, 512, False),
                                        (131072, 1024, 2048, True)])       # the last one is the bench shape: 512 stages per slab
def test_gemm_tn_transposed_read_kernel(M, N, K, wide):
    """The dW_ih shapes (M a multiple of 512 and >= 2048, N and K multiples of 256, no bias sum riding along) go to
    csrc/gemm_tn_tr.hip: row-major LDS tiles, fragments by ds_read_b64_tr_b16, X by LDS-DMA.  `wide`: dZ is a column block
    of a wider matrix, as the gate gradients of one direction are."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    lda = 2 * N if wide else N
    full = torch.randn(M, lda, generator=g).to(DEV)
    dZ = full[:, lda - N:]
    C0 = torch.randn(N, K, generator=g).to(DEV)
    Cm = C0.clone()
    ops.gemm_tn(dZ, X, Cm, M, N, K, rows_per_group=1, lda=lda)
    ref = C0.double() + dZ.double().t() @ X.double()
    assert _maxerr(Cm, ref) < 4e-4 * max(1.0, (M / 1000) ** 0.5 * 3)
    rel = float((Cm.double() - ref).norm() / ref.norm())
    assert rel < 2e-5, rel          # split arithmetic: ~4e-6 per product, averaged over the long reduction


@pytest.mark.parametrize('Hh,I,lens', [(32, 128, [5, 1, 9, 9, 3]), (256, 2048, [64] * 5), (64, 64, [4, 6]), (128, 96, [10, 3, 7]),
                                       (256, 512, [64] * 40)])
def test_lstm_on_bf16_rows_matches_oracle_fed_the_rounded_rows(Hh, I, lens):
    from stair_amd import ops
    cfg = dict(spec.DEFAULT_CONFIG, hidden_size=2 * Hh, video_size=I, max_video_length=64)
    names = ['submodules.video_encoder.' + n + sfx for sfx in ('', '_reverse')
             for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    w = {k: v.clone().requires_grad_(k in names) for k, v in oracle_weights(cfg, seed=4).items()}
    g = torch.Generator().manual_seed(Hh + I)
    xs = [torch.randn(L, I, generator=g).to(torch.bfloat16) for L in lens]
    d_outs = [torch.randn(L, 2 * Hh, generator=g) for L in lens]
    d_hn = torch.randn(len(lens), 2 * Hh, generator=g)
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    X = torch.cat(xs).to(DEV)
    ws = [w[n].detach().to(DEV) for n in names]
    out, h_n, gates, cbuf = ops.lstm_bidir(X, off.to(DEV), max(lens), ws, save=True)
    loss = 0
    check = range(len(xs)) if len(xs) <= 8 else (0, 7, len(xs) - 1)
    for s in check:
        ro, rh = O.lstm_bidir_explicit(w, 'video_encoder', xs[s].float())
        assert _maxerr(out[off[s]:off[s + 1]], ro) < 1e-4, s
        assert _maxerr(h_n[s], rh.reshape(-1)) < 1e-4, s
    if len(xs) > 8:
        return
    for s, x in enumerate(xs):
        ro, rh = O.lstm_bidir_explicit(w, 'video_encoder', x.float())
        loss = loss + (ro * d_outs[s]).sum() + (rh.reshape(-1) * d_hn[s]).sum()
    loss.backward()
    grads = ops.lstm_bidir_bwd(X, off.to(DEV), max(lens), ws, out, gates, cbuf, torch.cat(d_outs).to(DEV), d_hn.to(DEV))
    for n, gr in zip(names, grads):
        ref = w[n].grad
        scale = max(1.0, float(ref.abs().max()))
        assert _maxerr(gr, ref) < 2e-4 * scale, (n, _maxerr(gr, ref), scale)


def _model(config, seed=0):
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(config)
    w = synth.make_weights(config, seed)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    return m.to(DEV), w


@pytest.mark.parametrize('size', ['tiny', 'full'])
def test_bf16_clip_features_logits_match_oracle(size):
    """Whole path on stored bf16 features vs the oracle fed the same rounded features: 1e-4 on the logits, identical
    top-1 answers; and the fp32 path fed the rounded features agrees with the bf16 path to the split-product error."""
    if size == 'tiny':
        config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
        forms, n = synth.ALL_FORMS, 24
    else:
        config, forms, n = dict(spec.DEFAULT_CONFIG), synth.PAPER_FORMS, 16
    model, weights = _model(config, 0)
    qs = [synth.make_question(config, 0, i, form=forms[i % len(forms)]) for i in range(n)]
    for q in qs:
        q['video_features'] = torch.as_tensor(q['video_features']).to(torch.bfloat16)
    res = model.forward_batch(qs)
    assert res._video.dtype == torch.bfloat16
    logits, pred = res.logits.clone(), res.pred.clone()
    wt = O.to_torch(weights)
    for i, q in enumerate(qs):
        ref = O.forward(wt, config, dict(q, video_features=q['video_features'].float()), return_res_by_step=False)['logits']
        assert _maxerr(logits[i], ref) < 1e-4, (q['form'], _maxerr(logits[i], ref))
        assert int(pred[i]) == int(torch.argmax(ref)), q['form']
    res32 = model.forward_batch([dict(q, video_features=q['video_features'].float()) for q in qs])
    assert _maxerr(res32.logits, logits) < 2e-5


def test_bf16_clip_features_training_step_gradients():
    """One backward pass on bf16 features: every parameter gradient vs the oracle's autograd on the rounded features."""
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    model, weights = _model(config, 1)
    qs = [synth.make_question(config, 1, i, form=synth.ALL_FORMS[i % len(synth.ALL_FORMS)]) for i in range(12)]
    for q in qs:
        q['video_features'] = torch.as_tensor(q['video_features']).to(torch.bfloat16)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    res = model.forward_batch(qs, train=True)
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
    res.backward(answers, 1.0 / len(qs))
    names = [n for n, _ in spec.weight_table(config)]
    wt = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    loss = 0
    for q in qs:
        lg = O.forward(wt, config, dict(q, video_features=q['video_features'].float()), return_res_by_step=False)['logits']
        loss = loss + torch.nn.functional.cross_entropy(lg.unsqueeze(0), torch.tensor([q['answer']])) / len(qs)
    loss.backward()
    sd = dict(model.named_parameters())
    gmax = max(float(wt[n].grad.abs().max()) for n in names if wt[n].grad is not None)
    for n in names:
        ref = wt[n].grad if wt[n].grad is not None else torch.zeros_like(wt[n])
        assert _maxerr(sd[n].grad, ref) < 2e-4 * gmax, (n, _maxerr(sd[n].grad, ref), gmax)


def test_bf16_stored_clips_from_disk_to_logits(tmp_path):
    """The product data pipeline reaches the benched input path: .npy clips -> data.load_clip_features(dtype='bf16') ->
    data.pack_questions (bf16 on the device) -> run_programs selects the plane GEMM -> logits vs the oracle fed the same
    rounded clip values (1e-4, identical top-1)."""
    from stair_amd import data as D, ops
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=64, answer_vocab_length=16, max_video_length=40, object_types=10)
    rng = np.random.default_rng(3)
    clip_dir = tmp_path / 'clips'
    clip_dir.mkdir()
    vids = ['V%02d' % i for i in range(8)]
    for v in vids:
        np.save(clip_dir / (v + '.npy'), rng.standard_normal((80, 64)).astype(np.float32))      # every 2nd frame is kept: 40
    clips = D.load_clip_features(str(clip_dir), vids, 40, dtype='bf16')
    model, weights = _model(config, 2)
    items = []
    for i in range(20):
        q = synth.make_question(config, 6, i, form=synth.ALL_FORMS[i % len(synth.ALL_FORMS)], with_video=False)
        q['video_features'], q['video_id'] = clips[vids[i % 8]], vids[i % 8]
        q['question'] = torch.as_tensor(q['question'])
        items.append(q)
    b = D.pack_questions(items, DEV)
    assert b.video.dtype == torch.bfloat16 and b.n_clips == 8
    with ops.kernel_accounting() as acct:
        res = model.run_programs(b.programs, b.spans, b.video, b.question, b.q_lens, video_index=b.video_index)
    assert 'gemm_planes' in acct.table, sorted(acct.table)
    wt = O.to_torch(weights)
    for i, q in enumerate(items):
        ref = O.forward(wt, config, dict(q, video_features=q['video_features'].float()), return_res_by_step=False)['logits']
        assert _maxerr(res.logits[i], ref) < 1e-4 and int(res.pred[i]) == int(torch.argmax(ref)), q['form']
