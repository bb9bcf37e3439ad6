"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI,
against (a) the oracle on the same seeded inputs and (b) the committed golden fixtures that hold the
REFERENCE's own outputs (tests/golden/*.npz).  Tolerances are written per test; the contract is
logits within 1e-4 (fp32) and identical top-1 answers (BASELINE.json north_star).

/root/reference is never read here."""
import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import spec, synth
from helpers import PRETRAIN_MODULES, load_golden, oracle_weights, question_for

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(params=['f32', 'bf16x3'])
def matmul(request):
    """Every parity test below runs under the exact fp32 MFMA and under the default split-precision kernels."""
    from stair_amd import ops
    ops.set_matmul_mode(request.param)
    yield request.param
    ops.set_matmul_mode('bf16x3')


def _tol(matmul, exact, split):
    return exact if matmul == 'f32' else split


def _model(config, seed=0, pretrain_modules=frozenset()):
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(config, pretrain_modules=set(pretrain_modules))
    w = synth.make_weights(config, seed)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    return m.to(DEV)


def _maxerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(1, 16, 64), (7, 172, 1024), (130, 512, 300), (256, 128, 128), (1000, 512, 1536),
                                   (64 * 5, 512, 512), (129, 129, 36)])
@pytest.mark.parametrize('act', [None, 'relu', 'sigmoid'])
def test_gemm_matches_fp64(M, N, K, act, matmul):
    from stair_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act)
    ref = x.double() @ w.double().t() + b.double()
    ref = {None: ref, 'relu': ref.relu(), 'sigmoid': ref.sigmoid()}[act]
    assert _maxerr(y, ref) < _tol(matmul, 2e-5, 1e-4)          # fp32 accumulation over K <= 1536 terms of O(1)


def test_single_product_bf16_mode():
    """STAIR_MATMUL_BF16 (BASELINE configs[1]'s "bf16"): one bf16 product per operand pair.  GEMMs land where bf16
    rounding of both operands puts them (relative error ~3e-3 of the row scale, far outside the 1e-4 budget, which is why
    the mode is opt-in); on the full-size model the logits move by ~1e-3 and the top-1 answers of 64 questions stay those
    of the oracle."""
    from stair_amd import ops
    ops.set_matmul_mode('bf16')
    try:
        assert ops.get_matmul_mode() == 'bf16'
        g = torch.Generator().manual_seed(2)
        for (M, N, K) in ((1000, 512, 1536), (70000, 512, 128), (4096, 1024, 2048)):
            x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
            y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), 'relu').cpu().double()
            ref = (x.double() @ w.double().t() + b.double()).relu()
            rounded = (x.bfloat16().double() @ w.bfloat16().double().t() + b.double()).relu()
            assert float((y - rounded).abs().max()) < 2e-4                 # exactly the bf16-operand product, fp32 accumulation
            assert 1e-4 < float((y - ref).abs().max()) < 5e-2              # and measurably not the fp32 one
            dz = torch.randn(M, N, generator=g)
            dw = torch.zeros(N, K, device=DEV)
            ops.gemm_tn(dz.to(DEV), x.to(DEV), dw, M, N, K)
            refw = dz.bfloat16().double().t() @ x.bfloat16().double()
            assert float((dw.cpu().double() - refw).abs().max()) < 2e-3 * max(1.0, (M / 1000) ** 0.5)
        config = dict(spec.DEFAULT_CONFIG)
        model = _model(config, 2)
        w = oracle_weights(config, 2)
        qs = synth.make_questions(config, 11, 64, forms=synth.ALL_FORMS)
        res = model.forward_batch(qs)
        worst = 0.0
        for qi in range(0, 64, 4):
            r = O.forward(w, config, qs[qi])
            worst = max(worst, _maxerr(res.logits[qi], r['logits']))
            assert int(res.pred[qi]) == int(torch.argmax(r['logits']))
        assert 1e-5 < worst < 2e-2
        print('bf16 mode: max |logit - oracle| =', worst)
    finally:
        ops.set_matmul_mode('bf16x3')


@pytest.mark.parametrize('M,N,K', [(1, 512, 512), (7, 172, 1024), (300, 512, 1536), (1000, 512, 1536), (2048, 1024, 1024), (130, 512, 300)])
@pytest.mark.parametrize('act', [None, 'relu'])
def test_small_launches_split_k_deterministically(M, N, K, act, matmul):
    """The vector-level layers (<= 64 output tiles) split their K loop into 128-wide pieces when the caller provides
    scratch (a plan does): result == fp64 reference to the mode's tolerance, identical from run to run, and a row's
    output does not depend on which other rows are in the launch (same pieces, same order)."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    d = lambda t: t.to(DEV)
    y1 = ops.linear(d(x), d(w), d(b), act, splitk=True)
    y2 = ops.linear(d(x), d(w), d(b), act, splitk=True)
    assert torch.equal(y1, y2)
    ref = x.double() @ w.double().t() + b.double()
    ref = ref.relu() if act else ref
    assert _maxerr(y1, ref) < _tol(matmul, 2e-5, 1e-4)
    if ((M + 127) // 128) * ((N + 127) // 128) <= 64:           # both launches split (the 2048 x 1024 case is not small)
        solo = ops.linear(d(x[:1]), d(w), d(b), act, splitk=True)
        assert torch.equal(solo[0], y1[0])
    plain = ops.linear(d(x), d(w), d(b), act)                    # the unsplit kernel: same numbers up to summation order
    assert _maxerr(plain, y1.cpu()) < _tol(matmul, 2e-5, 1e-4)


def test_gemm_group_gather_scatter_rowscale(matmul):
    """The packed-launch form: tiles gathered/scattered by slot index, rows scaled before the product."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(5)
    T, H, slots, G = 24, 64, 11, 7
    arena = torch.randn(slots, T, H, generator=g)
    w = torch.randn(H, H, generator=g) / 8
    b = torch.randn(H, generator=g)
    rs = torch.rand(slots, T, generator=g)
    a_idx = torch.tensor([3, 0, 10, 5, 5, 1, 9], dtype=torch.int32)
    c_idx = torch.tensor([2, 4, 6, 8, 7, 0, 1], dtype=torch.int32)
    r_idx = torch.tensor([1, 2, 3, 4, 5, 6, 7], dtype=torch.int32)
    out = torch.full((slots, T, H), -7.0)
    d = lambda t: t.to(DEV)
    A, Cm, R = d(arena), d(out), d(rs)
    ops.gemm_grouped(A, T * H, d(a_idx), d(w), d(b), Cm, T * H, d(c_idx), G, T, H, H, act='relu', lda=H, ldc=H,
                     row_scale=R, rs_gstride=T, rs_gidx=d(r_idx))
    ref = out.clone().double()
    for gi in range(G):
        x = arena[a_idx[gi]].double() * rs[r_idx[gi]].double().unsqueeze(1)
        ref[c_idx[gi]] = (x @ w.double().t() + b.double()).relu()
    assert _maxerr(Cm, ref) < _tol(matmul, 1e-5, 1e-4)
    assert float(Cm.cpu()[3].min()) == -7.0 and float(Cm.cpu()[5].max()) == -7.0     # untouched slots stay untouched


@pytest.mark.parametrize('G,N,gather', [(1100, 512, True), (1030, 256, False), (2048, 128, True)])
def test_gemm_bench_sized_launch_paths(G, N, gather, matmul):
    """Launches as large as the benchmark's (>= 65 536 rows), which take the 8-wave 128 x 128 kernel (G=1100, 1030:
    ragged last round) or the 256 x 256 tiles (G=2048: whole rounds of CUs), gathered by slot index or contiguous,
    bias + ReLU epilogue -- every output row against fp64."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(G + N)
    T, K = 64, 128
    slots = G + 5
    arena = torch.randn(slots, T, K, generator=g)
    w = torch.randn(N, K, generator=g) / 8
    b = torch.randn(N, generator=g)
    a_idx = torch.randperm(slots, generator=g)[:G].to(torch.int32)
    c_idx = torch.randperm(slots, generator=g)[:G].to(torch.int32)
    out = torch.full((slots, T, N), -7.0, device=DEV)
    d = lambda t: t.to(DEV)
    if gather:
        ops.gemm_grouped(d(arena), T * K, d(a_idx), d(w), d(b), out, T * N, d(c_idx), G, T, N, K, act='relu', lda=K, ldc=N)
        ref = (arena[a_idx.long()].double() @ w.double().t() + b.double()).relu()
        got = out.cpu()[c_idx.long()]
        untouched = sorted(set(range(slots)) - set(c_idx.tolist()))
        assert float(out.cpu()[untouched].min()) == -7.0 and float(out.cpu()[untouched].max()) == -7.0
    else:
        ops.gemm_grouped(d(arena), T * K, None, d(w), d(b), out, T * N, None, G, T, N, K, act='relu', lda=K, ldc=N)
        ref = (arena[:G].double() @ w.double().t() + b.double()).relu()
        got = out.cpu()[:G]
        assert float(out.cpu()[G:].min()) == -7.0
    assert _maxerr(got, ref) < _tol(matmul, 1e-5, 1e-4)


@pytest.mark.parametrize('Hh,I,lens', [(32, 128, [5, 1, 9, 9, 3]), (32, 300, [17] * 33), (256, 2048, [64] * 3),
                                       (256, 300, [8, 25, 12, 19, 25, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20]),
                                       (64, 64, [4, 6]), (128, 64, [10, 3, 7])])
def test_lstm_matches_explicit_oracle(Hh, I, lens, matmul):
    from stair_amd import ops
    cfg = dict(spec.DEFAULT_CONFIG, hidden_size=2 * Hh, video_size=I, max_video_length=64)
    w = oracle_weights(cfg, seed=4)
    names = ['submodules.video_encoder.' + n + sfx for sfx in ('', '_reverse')
             for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    g = torch.Generator().manual_seed(Hh + I)
    xs = [torch.randn(L, I, generator=g) for L in lens]
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    out, h_n = ops.lstm_bidir(torch.cat(xs).to(DEV), off.to(DEV), max(lens), [w[n].to(DEV) for n in names])
    for s, x in enumerate(xs):
        ro, rh = O.lstm_bidir_explicit(w, 'video_encoder', x)
        assert _maxerr(out[off[s]:off[s + 1]], ro) < _tol(matmul, 2e-5, 1e-4), s
        assert _maxerr(h_n[s], rh.reshape(-1)) < _tol(matmul, 2e-5, 1e-4), s


@pytest.mark.parametrize('I,lens', [(300, [8, 25, 12, 19, 25, 8, 9, 10] * 5), (300, [11] * 47), (64, [30] * 9)])
def test_lstm_input_projection_on_padded_planes_matches_oracle(I, lens):
    """The text encoder's form (stair_lstm_args.x_planes_ws): fp32 token rows and W_ih split once into zero-padded bf16 hi / lo
    planes (E = 300 -> 320 columns), ONE plane GEMM for both directions (/root/reference/video_nmn/module_net.py:44-47,151-158);
    >= 256 rows take it, against the written-out oracle LSTM, ragged row count."""
    from stair_amd import ops
    Hh = 256
    cfg = dict(spec.DEFAULT_CONFIG, hidden_size=2 * Hh, video_size=I, max_video_length=64)
    w = oracle_weights(cfg, seed=6)
    names = ['submodules.video_encoder.' + n + sfx for sfx in ('', '_reverse')
             for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    g = torch.Generator().manual_seed(I + len(lens))
    xs = [torch.randn(L, I, generator=g) for L in lens]
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    assert int(off[-1]) >= 256
    with ops.kernel_accounting() as acct:
        out, h_n = ops.lstm_bidir(torch.cat(xs).to(DEV), off.to(DEV), max(lens), [w[n].to(DEV) for n in names], x_planes=True)
    assert 'gemm_planes' in acct.table, sorted(acct.table)
    for s in range(0, len(xs), 3):
        ro, rh = O.lstm_bidir_explicit(w, 'video_encoder', xs[s])
        assert _maxerr(out[off[s]:off[s + 1]], ro) < 2e-5, s
        assert _maxerr(h_n[s], rh.reshape(-1)) < 2e-5, s


def test_two_contexts_keep_their_own_options():
    """Two models (contexts) in one process with different per-context options, used alternately: each computes exactly what it
    computes when its setting is the process-wide one (stair_ctx_set_option: the exact-f32 model next to the split-product one, the
    unfused one next to the fused one)."""
    from stair_amd._lib import lib
    config = dict(spec.DEFAULT_CONFIG)
    qs = [synth.make_question(config, 3, i, form=f) for i, f in enumerate(synth.ALL_FORMS)]
    ref = {}
    for mode in (0, 1):
        assert lib.stair_set_matmul_mode(mode) == 0
        ref[mode] = _model(config, 5).forward_batch(qs).logits.cpu().clone()
    lib.stair_set_tile_mlp(0)
    ref['unfused'] = _model(config, 5).forward_batch(qs).logits.cpu().clone()
    lib.stair_set_tile_mlp(-1)
    assert not torch.equal(ref[0], ref[1]) and not torch.equal(ref[1], ref['unfused'])
    a, b, c = _model(config, 5), _model(config, 5), _model(config, 5)
    a.set_option('matmul_mode', 'f32'); c.set_option('tile_mlp', 0)
    for _ in range(2):                            # interleaved: no model inherits another's setting
        assert torch.equal(a.forward_batch(qs).logits.cpu(), ref[0])
        assert torch.equal(b.forward_batch(qs).logits.cpu(), ref[1])
        assert torch.equal(c.forward_batch(qs).logits.cpu(), ref['unfused'])
    a.set_option('matmul_mode', None)
    assert torch.equal(a.forward_batch(qs).logits.cpu(), ref[1])            # (the process-wide mode is 1 = bf16x3 again: the loop's last value)


def test_l2normalize_and_zero_vector():
    from stair_amd import ops
    x = torch.randn(9, 64)
    x[4] = 0
    y = ops.l2normalize(x.to(DEV))
    ref = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    assert _maxerr(y, ref) < 1e-6


# ---------------------------------------------------------------------------------------------
# whole path, tiny configurations: every intermediate value vs the reference's outputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['tiny_conv', 'tiny_conv_t24', 'tiny_linear'])
def test_every_program_node_matches_reference(name, matmul):
    z, meta = load_golden(name)
    config = meta['config']
    model = _model(config, meta['seed'])
    batch = [question_for(meta, q) for q in meta['questions']]
    res = model.forward_batch(batch)          # all forms in ONE batched pass
    TOL = _tol(matmul, 2e-5, 1e-4)
    for qi, q in enumerate(meta['questions']):
        key = 'q%d/' % q['qid']
        assert _maxerr(res.logits[qi], z[key + 'logits']) < TOL, key
        assert int(res.pred[qi]) == int(np.argmax(z[key + 'logits']))
        assert _maxerr(res.question_feature[qi], z[key + 'question_feature']) < TOL
        prog = batch[qi]['nmn_program_list']
        checked = 0
        for i, tok in enumerate(prog):
            k = key + 'step%d' % i
            if k in z.files:
                assert _maxerr(res.node(qi, i), z[k]) < TOL, (key, i, tok)
                checked += 1
        assert checked == sum(1 for f in z.files if f.startswith(key + 'step'))
        assert res.levels(qi) == O.module_levels(prog)


def test_single_question_forward_api_and_heads():
    """forward(data) returns the reference's dict: logits [A], res_by_step with pretrain heads."""
    z, meta = load_golden('tiny_conv')
    config = meta['config']
    model = _model(config, meta['seed'], PRETRAIN_MODULES)
    for q in meta['questions']:
        d = question_for(meta, q)
        out = model(d, return_res_by_step=True, return_result_of_each_step=True, test_mode=True)
        key = 'q%d/' % q['qid']
        assert out['logits'].shape == (config['answer_vocab_length'],)
        assert _maxerr(out['logits'], z[key + 'logits']) < 2e-5
        heads = {k for k in z.files if k.startswith(key + 'head')}
        assert {key + 'head%d' % i for i in out['res_by_step']} == heads
        for idx, (prog, val) in out['res_by_step'].items():
            assert _maxerr(val, z[key + 'head%d' % idx]) < 2e-5, (key, idx, prog)
        assert len(out['result_of_each_step']) == len(d['nmn_program_list'])
        assert 'sg_res_by_step' not in out


# ---------------------------------------------------------------------------------------------
# whole path, full-size configuration (BASELINE.json configs[1] shape: T=64, V=2048, H=512, A=172)
# ---------------------------------------------------------------------------------------------
def test_full_size_logits_match_reference(matmul):
    z, meta = load_golden('full')
    config = meta['config']
    model = _model(config, meta['seed'], PRETRAIN_MODULES)
    batch = [question_for(meta, q) for q in meta['questions']]
    res = model.forward_batch(batch)
    worst = 0.0
    for qi, q in enumerate(meta['questions']):
        key = 'q%d/' % q['qid']
        e = _maxerr(res.logits[qi], z[key + 'logits'])
        worst = max(worst, e)
        assert e < 1e-4, (key, e)                                   # the north_star bar
        assert int(res.pred[qi]) == int(np.argmax(z[key + 'logits'])), key
        assert _maxerr(res.question_feature[qi], z[key + 'question_feature']) < 1e-4
    print('full-size max |logit diff| vs reference: %.3g' % worst)
    # heads through the single-question API on a few questions
    for q in meta['questions'][:4]:
        out = model(question_for(meta, q), test_mode=True)
        key = 'q%d/' % q['qid']
        for idx, (prog, val) in out['res_by_step'].items():
            assert _maxerr(val, z[key + 'head%d' % idx]) < 1e-4, (key, idx, prog)


def test_batch_composition_does_not_change_results(matmul):
    """Size-independent property: a question's logits do not depend on what else is in the batch
    (packing order, bucket sizes) -- bit-exact, since every kernel treats groups independently."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 1)
    qs = synth.make_questions(config, 7, 48, forms=synth.ALL_FORMS)
    full = model.forward_batch(qs).logits.cpu()
    perm = np.random.RandomState(0).permutation(len(qs))
    shuffled = model.forward_batch([qs[i] for i in perm]).logits.cpu()
    assert torch.equal(shuffled, full[perm])
    solo = model.forward_batch([qs[5]]).logits.cpu()
    assert torch.equal(solo[0], full[5])


def _questions_sharing_clips(config, seed, n, n_clips, T=None):
    """n questions over n_clips clips: question q asks about clip q % n_clips and carries that clip's
    feature tensor OBJECT, as AGQADataset does (dataset.py:183)."""
    qs = synth.make_questions(config, seed, n, forms=synth.ALL_FORMS, T=T) if T else synth.make_questions(config, seed, n, forms=synth.ALL_FORMS)
    clips = [torch.as_tensor(qs[c]['video_features']) for c in range(n_clips)]
    for i, q in enumerate(qs):
        q['video_features'] = clips[i % n_clips]
    return qs


def test_questions_sharing_a_clip_encode_it_once(matmul):
    """the per-video encoder cache.  40 questions over 5 clips: the shared plan encodes 5 clips,
    the expanded plan 40; logits are bit-identical, match the oracle run question by question, and the plan
    really holds 35 fewer [T,H] maps."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 3)
    w = oracle_weights(config, 3)
    qs = _questions_sharing_clips(config, 21, 40, 5)
    shared = model.forward_batch(qs, cse=False)              # clip sharing alone: every module node still has a slot of its own
    expanded = model.forward_batch(qs, share_videos=False)
    assert shared._video.shape[0] == 5 and expanded._video.shape[0] == 40
    assert expanded.info.n_map - shared.info.n_map == 35
    with_cse = model.forward_batch(qs)                       # ... and with clip-level common subexpressions computed once per clip
    assert with_cse.info.n_aliased > 0 and with_cse.info.n_map <= shared.info.n_map and shared.info.n_aliased == 0
    assert torch.equal(with_cse.logits, shared.logits)
    assert torch.equal(shared.logits.cpu(), expanded.logits.cpu())
    assert torch.equal(shared.pred.cpu(), expanded.pred.cpu())
    for qi in range(0, 40, 3):
        r = O.forward(w, config, qs[qi])
        assert _maxerr(shared.logits[qi], r['logits']) < 1e-4
        assert int(shared.pred[qi]) == int(torch.argmax(r['logits']))
    # explicit index form of the same call, clips in another order
    video = torch.stack([qs[c]['video_features'] for c in (4, 3, 2, 1, 0)]).to(DEV)
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    res = model.run_programs([q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs], video,
                             question, [q['question'].shape[0] for q in qs], video_index=[4 - (i % 5) for i in range(40)])
    assert torch.equal(res.logits.cpu(), shared.logits.cpu())
    with pytest.raises(ValueError):
        model.run_programs([qs[0]['nmn_program_list']], [qs[0]['prog_str_to_question_tokens']], video, question[:qs[0]['question'].shape[0]],
                           [qs[0]['question'].shape[0]], video_index=[5])


def test_appearance_feature_config_against_oracle(matmul):
    """The README's ResNet/ResNeXt setting (README.md:180-183, BASELINE configs[0]): 8 frames of 4096 features,
    max_video_length 8 -> Linear(T,T) Temporal nets at full hidden size.  24 questions of all forms vs the oracle."""
    config = dict(spec.DEFAULT_CONFIG, video_size=4096, max_video_length=8)
    model = _model(config, 6)
    w = oracle_weights(config, 6)
    qs = synth.make_questions(config, 13, 24, forms=synth.ALL_FORMS)
    assert qs[0]['video_features'].shape == (8, 4096)
    res = model.forward_batch(qs)
    for qi in range(0, 24, 2):
        r = O.forward(w, config, qs[qi])
        assert _maxerr(res.logits[qi], r['logits']) < 1e-4
        assert int(res.pred[qi]) == int(torch.argmax(r['logits']))


def test_configs0_thousand_questions_against_oracle():
    """BASELINE configs[0] at the survey's size (Config 1): 1 000 questions, 8 frames of 2048 appearance
    features (the h5 path averages the clip-frames axis, dataset.py:145-154), max_video_length 8 -> Linear(T,T) Temporal nets.
    One batched pass on the GPU; every question's logits against the CPU oracle (batch-1, as the reference runs) at 1e-4
    with identical top-1."""
    config = dict(spec.DEFAULT_CONFIG, video_size=2048, max_video_length=8)
    model = _model(config, 8)
    w = oracle_weights(config, 8)
    qs = synth.make_questions(config, 21, 1000, T=8)
    assert qs[0]['video_features'].shape == (8, 2048)
    res = model.forward_batch(qs)
    logits, pred = res.logits.cpu(), res.pred.cpu()
    worst = 0.0
    for qi, q in enumerate(qs):
        r = O.forward(w, config, q)
        worst = max(worst, float((logits[qi] - r['logits']).abs().max()))
        assert int(pred[qi]) == int(torch.argmax(r['logits'])), qi
    assert worst < 1e-4, worst


@pytest.mark.parametrize('H,V,L,T', [(64, 128, 40, 2), (64, 260, 40, 33), (128, 128, 2, 2), (128, 260, 64, 63), (256, 128, 100, 100),
                                     (256, 260, 40, 7), (512, 128, 8, 8), (512, 260, 40, 33), (512, 128, 100, 100)])
def test_odd_shapes_against_oracle(H, V, L, T):
    """Hidden sizes 64..512, feature sizes that are not multiples of 64, frame counts 2..100 incl. odd ones, T below and
    at max_video_length, Conv1d and Linear(T,T) Temporal nets: all 12 program forms in one batch, every other question
    against the oracle."""
    config = dict(spec.DEFAULT_CONFIG, hidden_size=H, video_size=V, answer_vocab_length=16, max_video_length=L, object_types=10)
    model = _model(config, 1)
    w = oracle_weights(config, 1)
    qs = synth.make_questions(config, 5, 12, forms=synth.ALL_FORMS, T=T)
    res = model.forward_batch(qs)
    for qi in range(0, 12, 2):
        r = O.forward(w, config, qs[qi])
        assert _maxerr(res.logits[qi], r['logits']) < 1e-4
        assert int(res.pred[qi]) == int(torch.argmax(r['logits']))


def test_larger_batch_against_oracle(matmul):
    """128 random questions of all 12 forms vs the oracle run question by question."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 2)
    w = oracle_weights(config, 2)
    qs = synth.make_questions(config, 11, 128, forms=synth.ALL_FORMS)
    res = model.forward_batch(qs)
    logits = res.logits.cpu()
    for qi in range(0, 128, 4):               # oracle on every 4th keeps the test to seconds
        r = O.forward(w, config, qs[qi])
        assert _maxerr(logits[qi], r['logits']) < 1e-4
        assert int(res.pred[qi]) == int(torch.argmax(r['logits']))


@pytest.mark.parametrize('n,C,H,k', [(1, 5, 64, 5), (37, 214, 512, 10), (300, 1024, 256, 3), (9, 70, 96, 10)])
def test_cosine_topk_matches_torch(n, C, H, k):
    """stair_cosine_topk == argsort(CosineSimilarity) (evaluate.py:95-97): indices exact, similarities to 2e-6;
    row gather through q_idx; exact ties keep the lower index."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(n + C)
    rows = torch.randn(n + 3, H, generator=g)
    keys = torch.randn(C, H, generator=g)
    keys[C // 2] = keys[0]                                        # an exact tie between candidates 0 and C//2
    keys[-1] = 0.0                                                # zero vector: cosine 0 through the eps clamp
    pick = torch.randperm(n + 3, generator=g)[:n].to(torch.int32)
    idx, sim = ops.cosine_topk(rows.to(DEV), keys.to(DEV), k, q_idx=pick.to(DEV))
    ref = torch.nn.functional.cosine_similarity(rows[pick.long()].double().unsqueeze(1), keys.double().unsqueeze(0), dim=2)
    idx, sim = idx.cpu().long(), sim.cpu().double()
    assert float((sim - ref.gather(1, idx)).abs().max()) < 2e-6   # the reported similarity is that of the reported index
    top = torch.sort(ref, dim=1, descending=True).values[:, :k]
    assert float((sim - top).abs().max()) < 2e-6                  # ... and they are the k largest, best first
    assert bool((sim[:, :-1] >= sim[:, 1:]).all()) if k > 1 else True
    for r in range(n):
        assert len(set(idx[r].tolist())) == k
        row = idx[r].tolist()
        if 0 in row and C // 2 in row:
            assert row.index(0) + 1 == row.index(C // 2)          # tie: lower index first, adjacent
    full, _ = ops.cosine_topk(rows.to(DEV), keys.to(DEV), k)     # without the gather
    assert full.shape == (n + 3, k)
    with pytest.raises(Exception):
        ops.cosine_topk(rows.to(DEV), keys.to(DEV), C + 1)


def test_filter_text_results_match_reference():
    """stair_amd.evaluate.filter_text_results on the HIP path == the reference's get_filter_text_results output
    (tests/golden/filter_text.json).  A rank may differ from the fixture only between phrases whose reference
    similarities are closer than 1e-5 (none are, in this fixture)."""
    import json, os
    from stair_amd import evaluate as E
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'filter_text.json')))
    z, meta = load_golden(gold['config'])
    config = meta['config']
    model = _model(config, meta['seed'], PRETRAIN_MODULES)
    qs = []
    for q in meta['questions']:
        d = question_for(meta, q)
        d['qa_id'] = 'q%d' % q['qid']
        qs.append(d)
    embs = [synth.class_embedding(config, meta['seed'], c) for c in range(len(gold['vocab']))]
    got = E.filter_text_results(model, qs, gold['vocab'], embs, batch_size=5)
    assert sorted(got) == sorted(gold['results'])
    n = 0
    for qa, entry in gold['results'].items():
        assert sorted(str(k) for k in got[qa]) == sorted(entry)
        for pidx, ref in entry.items():
            level, kw, top = got[qa][int(pidx)]
            assert (level, kw) == (ref['level'], ref['keyword'])
            for a, b, in zip(top, ref['top']):
                if a != b:
                    sa, sb = ref['sims'][ref['top'].index(a)], ref['sims'][ref['top'].index(b)]
                    assert abs(sa - sb) < 1e-5, (qa, pidx, a, b)
            n += 1
    assert n == 16
    # phrase representations: batched ragged text-encoder pass == the reference's one-by-one loop (evaluate.py:66-76)
    w = oracle_weights(config, meta['seed'])
    reps = model.encode_phrases(embs).cpu()
    for c in (0, 7, 29):
        ref = O.l2normalize(O.encode_question(w, torch.as_tensor(embs[c]))[1])
        assert _maxerr(reps[c], ref) < 1e-5          # unit vectors, split-precision GEMM in the input projection


def test_on_disk_dataset_to_logits():
    """tests/golden/agqa_mini (npy clips, question records, GloVe text, vocab) -> stair_amd.data -> packed batch ->
    HIP path, against the oracle question by question; the packed path, forward_batch and evaluate.predict agree."""
    import json, os
    from stair_amd import data as D, evaluate as E
    mini = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'agqa_mini')
    recs = D.filter_records(D.load_question_records(os.path.join(mini, 'records.json')), 'valid')
    clips = D.load_clip_features(os.path.join(mini, 'clips'), {r['video_id'] for r in recs}, 10)
    vocab = D.load_answer_vocab(os.path.join(mini, 'vocab.json'))
    ds = D.AGQAQuestions(recs, clips, D.load_glove(os.path.join(mini, 'glove.txt')), vocab, split='valid',
                         video_secs=json.load(open(os.path.join(mini, 'video_secs.json'))), tokenize=str.split)
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=16, text_size=8, answer_vocab_length=ds.answer_vocab_length(),
                  max_video_length=40, object_types=10)
    model = _model(config, 5)
    w = oracle_weights(config, 5)
    items = [ds[i] for i in range(len(ds))]
    preds = E.predict(model, items, batch_size=4)
    ball = D.pack_questions(items, DEV)            # every clip length in ONE launch batch
    rall = model.run_programs(ball.programs, ball.spans, ball.video, ball.question, ball.q_lens, video_index=ball.video_index,
                              video_len=ball.video_len)
    for i in range(len(items)):
        assert _maxerr(rall.logits[i], O.forward(w, config, items[i])['logits']) < 1e-4
    for T, idxs in E.group_by_frames(items, ragged=False).items():
        group = [items[i] for i in idxs]
        b = D.pack_questions(group, DEV)
        res = model.run_programs(b.programs, b.spans, b.video, b.question, b.q_lens, video_index=b.video_index)
        assert b.n_clips == 1 and tuple(b.video.shape) == (1, T, 16)
        assert torch.equal(res.logits.cpu(), model.forward_batch(group).logits.cpu())
        for j, i in enumerate(idxs):
            r = O.forward(w, config, items[i])
            assert _maxerr(res.logits[j], r['logits']) < 1e-4
            assert preds[i] == int(torch.argmax(r['logits']))
    acc, _ = E.evaluate(model, items, unk_token_id=vocab['word2id']['<UNK>'], batch_size=3, shard='clip')
    assert acc == E.accuracy(preds, [int(it['answer']) for it in items], vocab['word2id']['<UNK>'])


@pytest.mark.parametrize('queue', [1, 0])
def test_captured_plan_replays_bit_exactly(matmul, queue):
    """BASELINE configs[3]: a plan's forward pass recorded into a hipGraph.  Replay == eager run, bit for bit; new
    inputs written into the static tensors (same programs and lengths) give the eager result for those inputs -- on the
    third and fourth replay too, with other batches run eagerly in between.  Both tile schedules: the self-resetting work
    queue (default; round 3 fell back to the static schedule under capture) and the static round robin."""
    from stair_amd._lib import lib
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 4)
    qs = synth.make_questions(config, 31, 24, forms=synth.ALL_FORMS)
    lib.stair_set_tile_queue(queue)
    try:
        eager = model.forward_batch(qs)
        want = eager.logits.clone()
        cap = model.forward_batch(qs).capture_graph()
        cap.logits.zero_()
        logits, pred = cap.replay()
        torch.cuda.synchronize()
        if not torch.equal(logits, want):      # say where: a replay that differs is a determinism bug somewhere in the forward pass
            d = (logits - want).abs().amax(1)
            rows = d.nonzero().flatten().tolist()
            raise AssertionError('replay differs from the eager run in %d rows (max %.3g): %s' % (
                len(rows), float(d.max()), [(i, qs[i]['nmn_program_list'][0]) for i in rows[:8]]))
        assert torch.equal(pred, eager.pred)
        # other clips and word embeddings under the same programs, spans and lengths (= the same plan)
        for rep in range(3):
            g = torch.Generator().manual_seed(5 + rep)
            other = []
            for q in qs:
                o = dict(q)
                o['video_features'] = torch.randn(q['video_features'].shape, generator=g)
                o['question'] = torch.randn(q['question'].shape, generator=g)
                other.append(o)
            fresh = model.forward_batch(other).logits.clone()
            cap.result._video.copy_(torch.stack([torch.as_tensor(q['video_features']) for q in other]).to(DEV))
            cap.result._question.copy_(torch.cat([torch.as_tensor(q['question']) for q in other]).to(DEV))
            cap.logits.zero_()
            logits, _ = cap.replay()
            torch.cuda.synchronize()
            assert torch.equal(logits, fresh), rep
            # the queue words are back at zero after every pass
            off = cap.result.info.status_off
            assert int(cap._ws[off + 16: off + 20].view(torch.int32).abs().sum()) == 0
        with pytest.raises(Exception):
            model.forward_batch(qs, train=True).capture_graph()
    finally:
        lib.stair_set_tile_queue(-1)


def test_program_deeper_than_the_old_queue_head_block(matmul):
    """A program nested 40 levels deep (round 3 reserved one queue head per fused launch, 24 forward launches at most, and
    refused deeper plans): forward against the oracle, and a training step runs through its 40 backward chain launches."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 3)
    w = oracle_weights(config, 3)
    depth = 39
    qs = synth.make_questions(config, 13, 6, forms=['P1'])
    for q in qs:
        q['nmn_program_list'] = ['Exists', 'dish', 'Filter'] + ['FilterFrame'] * depth + ['video'] + ['holding'] * depth + ['objects']
        q['nmn_program_idx'] = list(range(len(q['nmn_program_list'])))
        Q = q['question'].shape[0]
        q['prog_str_to_question_tokens'] = {i: (1 + i % (Q - 2), 2 + i % (Q - 2)) for i in range(len(q['nmn_program_list']))}
    res = model.forward_batch(qs)
    assert res.info.n_levels >= depth + 2
    for qi in (0, 5):
        r = O.forward(w, config, qs[qi])
        assert _maxerr(res.logits[qi], r['logits']) < 1e-4
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    tr = model.forward_batch(qs, train=True)
    loss = tr.backward(torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()
    tr.check()


def test_forward_pass_is_bit_reproducible(matmul):
    """The inference pass has no atomics (split-K partials are reduced in a fixed order, the cooperative recurrence adds its
    exchanged pieces in a fixed order): 20 runs of one batch, every form, full hidden size, must agree bit for bit -- a data race
    in one of the cross-workgroup hand-offs would show up here as a rare difference."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 4)
    qs = synth.make_questions(config, 77, 40, forms=synth.ALL_FORMS)
    first = model.forward_batch(qs).logits.clone()
    for it in range(20):
        again = model.forward_batch(qs).logits
        if not torch.equal(again, first):
            d = (again - first).abs().amax(1)
            rows = d.nonzero().flatten().tolist()
            raise AssertionError('run %d differs in %d rows (max %.3g): %s' % (it, len(rows), float(d.max()),
                                                                           [(i, qs[i]['nmn_program_list'][0]) for i in rows[:8]]))


def test_forward_reads_no_uninitialised_workspace():
    """Every workspace float a kernel reads was written earlier in the same pass: with the (reused) workspace filled with
    NaN, 1e30 or -7 before the pass, the logits do not change by a bit -- uniform and mixed clip lengths."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 4)
    batches = [synth.make_questions(config, 31, 24, forms=synth.ALL_FORMS), synth.make_questions(config, 32, 33, forms=synth.ALL_FORMS, T=40)]
    ragged = synth.make_questions(config, 5, 24, forms=synth.ALL_FORMS)
    for i, q in enumerate(ragged):
        q['video_features'] = torch.as_tensor(q['video_features'])[:64 - (i * 7) % 40].clone()
    batches.append(ragged)
    for qs in batches:
        ref = model.forward_batch(qs).logits.clone()
        for poison in (float('nan'), 1e30, -7.0):
            model._ws.fill_(poison)
            got = model.forward_batch(qs).logits
            assert torch.equal(got, ref), (poison, int(got.isnan().sum()))


def test_missing_gpu_tensor_fails_loudly():
    from stair_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(4, 8), torch.randn(4, 8))


def test_evaluation_driver_buckets_by_frame_count():
    """stair_amd.evaluate.predict on a mixed-T question list (T=40 and T=24) reproduces the reference's
    top-1 answers; accuracy follows train_module.py:252-253."""
    from stair_amd import evaluate as E
    z40, m40 = load_golden('tiny_conv')
    z24, m24 = load_golden('tiny_conv_t24')
    model = _model(m40['config'], m40['seed'])
    qs, gold_pred = [], []
    for z, meta in ((z40, m40), (z24, m24)):
        for q in meta['questions']:
            qs.append(question_for(meta, q))
            gold_pred.append(int(np.argmax(z['q%d/logits' % q['qid']])))
    order = np.random.RandomState(1).permutation(len(qs))
    qs = [qs[i] for i in order]
    gold_pred = [gold_pred[i] for i in order]
    preds = E.predict(model, qs, batch_size=5)
    assert preds == gold_pred
    acc, preds2 = E.evaluate(model, qs, unk_token_id=15, batch_size=7)
    assert preds2 == gold_pred
    assert acc == E.accuracy(gold_pred, [q['answer'] for q in qs], 15)


# ---------------------------------------------------------------------------------------------
# backward building blocks (training path)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,R', [(64, 16, 64, 1), (5000, 512, 512, 1), (24 * 7, 64, 128, 24), (1000, 1024, 300, 1),
                                     (333, 36, 512, 1)])
def test_gemm_tn_weight_gradient(M, N, K, R, matmul):
    """dW = dZ^T (rs * X) with X gathered in groups, accumulated on top of existing contents."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    G = M // R
    slots = G + 3
    X = torch.randn(slots, R, K, generator=g)
    idx = torch.randperm(slots, generator=g)[:G].to(torch.int32)
    dZ = torch.randn(M, N, generator=g)
    rs = torch.rand(slots, R, generator=g)
    C0 = torch.randn(N, K, generator=g)
    d = lambda t: t.to(DEV)
    Cm = d(C0.clone())
    ops.gemm_tn(d(dZ), d(X), Cm, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=d(idx), row_scale=d(rs),
                rs_gstride=R, rs_gidx=d(idx))
    Xg = (X[idx.long()] * rs[idx.long()].unsqueeze(-1)).reshape(M, K).double()
    ref = C0.double() + dZ.double().t() @ Xg
    assert _maxerr(Cm, ref) < _tol(matmul, 1e-4, 4e-4) * max(1.0, (M / 1000) ** 0.5 * 3)


@pytest.mark.parametrize('M,N,K', [(64, 16, 64), (4096, 512, 512), (1001, 1024, 300), (333, 36, 512), (8192 + 64, 128, 2048),
                                   (16384, 1024, 2048)])     # the last one takes the 256 x 256 tile kernel (32 output tiles, long M)
def test_gemm_tn_bias_gradient_rides_along(M, N, K, matmul):
    """stair_gemm_tn_args.colsum / colsum2: db += colsum(dZ) out of the same launch as dW (fused into the staging of
    the split-precision kernel, a second kernel in f32 mode), on top of existing contents, ragged M included."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(M * 3 + N + K)
    X = torch.randn(M, K, generator=g)
    dZ = torch.randn(M, N, generator=g)
    b0 = torch.randn(N, generator=g)
    d = lambda t: t.to(DEV)
    Cm, b1, b2 = torch.zeros(N, K, device=DEV), d(b0.clone()), d(b0.clone() * 2)
    ops.gemm_tn(d(dZ), d(X), Cm, M, N, K, rows_per_group=1, colsum=b1, colsum2=b2)
    ref = dZ.double().sum(0)
    tol = 4e-5 * max(1.0, (M / 100) ** 0.5)          # fp32 accumulation of M terms of unit variance
    assert _maxerr(b1, b0.double() + ref) < tol and _maxerr(b2, 2 * b0.double() + ref) < tol
    refw = (d(dZ).double().t() @ d(X).double()).cpu()             # fp64 on the device: the largest case is 69 GFLOP
    assert _maxerr(Cm, refw) < _tol(matmul, 1e-4, 4e-4) * max(1.0, (M / 1000) ** 0.5 * 3)


@pytest.mark.parametrize('M', [6700, 6704])
def test_gemm_tn_is_stable_with_two_workgroups_per_cu(M):
    """Regression: a ragged product whose grid puts two WORKING workgroups on every CU (16 tiles x 32 slabs), launched back to back.  The
    form of the kernel that kept its row masks in registers spilled 12 bytes per lane, and with the spill a third of such launches returned
    sums that were wrong by O(|dW|) (dW_hh of the text encoder in tests/test_gpu_lstm_coop.py was where it showed); see
    tests/test_kernel_resources.py for the static side of the same rule."""
    from stair_amd import ops
    N, K = 1024, 256
    g = torch.Generator(device=DEV).manual_seed(M)
    dZ = torch.randn(M, N, device=DEV, generator=g)
    X = torch.randn(M, K, device=DEV, generator=g)
    ref = (dZ.double().t() @ X.double()).float()
    tol = 2e-5 * float(ref.abs().max())
    outs = []
    for _ in range(60):
        Cm = torch.zeros(N, K, device=DEV)
        ops.gemm_tn(dZ, X, Cm, M, N, K)
        outs.append(Cm)
    torch.cuda.synchronize()
    worst = max(float((c - ref).abs().max()) for c in outs)
    assert worst < tol, (worst, tol)


def test_gemm_accumulate_scatter_add(matmul):
    """dX products: two groups writing the same output slot must add up (atomic epilogue)."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(9)
    T, H = 8, 64
    dY = torch.randn(3, T, H, generator=g)
    Wt = torch.randn(H, H, generator=g) / 8
    out0 = torch.randn(4, T, H, generator=g)
    c_idx = torch.tensor([2, 2, 0], dtype=torch.int32)
    d = lambda t: t.to(DEV)
    Cm = d(out0.clone())
    ops.gemm_grouped(d(dY), T * H, None, d(Wt), None, Cm, T * H, d(c_idx), 3, T, H, H, lda=H, ldc=H, accumulate=True)
    ref = out0.clone().double()
    for gi in range(3):
        ref[c_idx[gi]] += dY[gi].double() @ Wt.double().t()
    assert _maxerr(Cm, ref) < _tol(matmul, 1e-5, 1e-4)


@pytest.mark.parametrize('Hh,I,lens', [(32, 128, [5, 1, 9, 9, 3]), (256, 300, [8, 25, 12, 19, 25, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20]),
                                       (256, 512, [64] * 5), (64, 64, [4, 6]), (128, 64, [10, 3, 7])])
def test_lstm_backward_matches_autograd_of_oracle(Hh, I, lens, matmul):
    from stair_amd import ops
    cfg = dict(spec.DEFAULT_CONFIG, hidden_size=2 * Hh, video_size=I, max_video_length=64)
    names = ['submodules.video_encoder.' + n + sfx for sfx in ('', '_reverse')
             for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    w = {k: v.clone().requires_grad_(k in names) for k, v in oracle_weights(cfg, seed=4).items()}
    g = torch.Generator().manual_seed(Hh + I)
    xs = [torch.randn(L, I, generator=g) for L in lens]
    d_outs = [torch.randn(L, 2 * Hh, generator=g) for L in lens]
    d_hn = torch.randn(len(lens), 2 * Hh, generator=g)
    loss = 0
    for s, x in enumerate(xs):
        ro, rh = O.lstm_bidir_explicit(w, 'video_encoder', x)
        loss = loss + (ro * d_outs[s]).sum() + (rh.reshape(-1) * d_hn[s]).sum()
    loss.backward()
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32).to(DEV)
    X = torch.cat(xs).to(DEV)
    ws = [w[n].detach().to(DEV) for n in names]
    out, h_n, gates, cbuf = ops.lstm_bidir(X, off, max(lens), ws, save=True)
    grads = ops.lstm_bidir_bwd(X, off, max(lens), ws, out, gates, cbuf, torch.cat(d_outs).to(DEV), d_hn.to(DEV))
    for n, gr in zip(names, grads):
        ref = w[n].grad
        scale = max(1.0, float(ref.abs().max()))
        assert _maxerr(gr, ref) < 2e-4 * scale, (n, _maxerr(gr, ref), scale)
