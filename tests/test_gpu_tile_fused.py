"""The fused per-clip tile operators inside the plan runner (csrc/tile_mlp.hip; DEFAULT_CONFIG: H = 512, T <= 64) against
(1) the oracle, node by node, on every program form, and (2) the GEMM -> row-kernel launch sequences they replace
(stair_set_tile_mlp(0)), forward values and every parameter gradient.  `-m gpu`."""
import numpy as np
import pytest
import torch

from oracle import nmn_oracle as O
from stair_amd import spec, synth
from helpers import oracle_weights

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(config, seed=0):
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(config)
    w = synth.make_weights(config, seed)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    return m.to(DEV)


@pytest.fixture
def unfused():
    """switches the plan runner back to the kernel sequences for the duration of a `with`-free block"""
    from stair_amd._lib import lib

    def set_(on):
        lib.stair_set_tile_mlp(on)
    yield set_
    lib.stair_set_tile_mlp(-1)


@pytest.fixture
def vec_tiles():
    """vector-level modules on the tile operator whatever the bucket size (the default policy wants 128 instances per bucket)"""
    import os
    old = os.environ.get('STAIR_TILE_VEC')
    os.environ['STAIR_TILE_VEC'] = '1'
    yield
    if old is None:
        del os.environ['STAIR_TILE_VEC']
    else:
        os.environ['STAIR_TILE_VEC'] = old


@pytest.mark.parametrize('T', [64, 40])
def test_every_node_at_full_width_matches_the_oracle(T, vec_tiles):
    """Every intermediate value of all 12 program forms at H = 512 (the tiny-config node tests never reach the fused
    operators): vec / map / attention values against the oracle's interpreter, 2e-5 relative to the value's scale."""
    from stair_amd import ops
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 2)
    w = oracle_weights(config, 2)
    qs = [synth.make_question(config, 9, i, form=f, T=T) for i, f in enumerate(synth.ALL_FORMS)]
    with ops.kernel_accounting() as acct:
        res = model.forward_batch(qs)
    assert 'tile_mlp' in acct.table and acct.table['tile_mlp'][0] >= 8, sorted(acct.table)
    checked = 0
    for qi, q in enumerate(qs):
        r = O.forward(w, config, q, return_res_by_step=False, return_result_of_each_step=True, pretrain_modules=frozenset())
        assert float((res.logits[qi].cpu() - r['logits']).abs().max()) < 1e-4
        for i, (params, ref) in enumerate(r['result_of_each_step']):
            if not isinstance(ref, torch.Tensor):
                continue
            got = res.node(qi, i)
            got = got.cpu() if isinstance(got, torch.Tensor) else got
            err = float((got.reshape(ref.shape) - ref).abs().max())
            assert err < 2e-5 * max(1.0, float(ref.abs().max())), (q['form'], i, q['nmn_program_list'][i], err)
            checked += 1
    assert checked > 100


def test_fused_and_sequenced_paths_agree_forward_and_backward(unfused, vec_tiles):
    """Same batch through both runners: logits, every node, per-question loss and every parameter gradient.  The two compute the
    same split-bf16 products in another order, so only rounding separates them."""
    config = dict(spec.DEFAULT_CONFIG)
    qs = [synth.make_question(config, 4, i, form=f) for i, f in enumerate(synth.ALL_FORMS * 2)]
    out = {}
    for mode in (1, 0):
        unfused(mode)
        model = _model(config, 3)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        res = model.forward_batch(qs, train=True)
        nodes = []
        for qi, q in enumerate(qs):
            for i in range(len(q['nmn_program_list'])):
                v = res.node(qi, i)
                if isinstance(v, torch.Tensor):
                    nodes.append(v.detach().cpu().clone())
        answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
        loss = res.backward(answers, 1.0 / len(qs))
        out[mode] = (res.logits.cpu().clone(), nodes, loss.cpu().clone(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
    assert float((out[1][0] - out[0][0]).abs().max()) < 2e-5
    for a, b in zip(out[1][1], out[0][1]):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    assert torch.allclose(out[1][2], out[0][2], rtol=1e-5, atol=1e-5)
    gmax = max(float(g.abs().max()) for g in out[0][3].values())
    for n, g in out[0][3].items():
        assert float((out[1][3][n] - g).abs().max()) < 2e-4 * max(float(g.abs().max()), 1e-3 * gmax), n


@pytest.mark.parametrize('ragged', [False, True])
def test_fused_and_sequenced_paths_agree_under_dropout(unfused, ragged):
    """nn.Dropout(0.25) at the reference's `D` positions (modules.py; args.py:31 -- the recipe the reference trains with).  The fused tile
    operator draws the bits stair_dropout_fwd draws for the same (site, element) in its epilogues, so with one seed both runners drop
    the SAME elements and only the order of the split-bf16 sums separates them: logits, every node, the loss and every parameter
    gradient (the chains take their relu' masks from the saved post-dropout activations and the factor 1 / (1 - p))."""
    config = dict(spec.DEFAULT_CONFIG)
    lens = [64, 17, 40, 64, 33, 8, 51, 64, 29, 12, 64, 45]
    forms = synth.ALL_FORMS * 2
    qs = [synth.make_question(config, 4, i, form=f, T=(lens[i % 12] if ragged else 64)) for i, f in enumerate(forms)]
    out = {}
    for mode in (1, 0):
        unfused(mode)
        model = _model(config, 3)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        res = model.forward_batch(qs, train=True, dropout=(0.25, 7))
        nodes = []
        for qi, q in enumerate(qs):
            for i in range(len(q['nmn_program_list'])):
                v = res.node(qi, i)
                if isinstance(v, torch.Tensor):
                    nodes.append(v.detach().cpu().clone())
        answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV)
        loss = res.backward(answers, 1.0 / len(qs))
        out[mode] = (res.logits.cpu().clone(), nodes, loss.cpu().clone(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
        if mode == 1:
            plain = model.forward_batch(qs, train=True).logits.cpu()
            assert float((plain - out[1][0]).abs().max()) > 1e-3          # the masks are really on
            again = model.forward_batch(qs, train=True, dropout=(0.25, 7)).logits.cpu()
            other = model.forward_batch(qs, train=True, dropout=(0.25, 8)).logits.cpu()
            assert torch.equal(again, out[1][0]) and not torch.equal(other, out[1][0])      # a seed fixes the step
    assert float((out[1][0] - out[0][0]).abs().max()) < 5e-5
    zeros = 0
    for a, b in zip(out[1][1], out[0][1]):
        assert float((a - b).abs().max()) < 5e-5 * max(1.0, float(b.abs().max()))
        assert torch.equal(a == 0, b == 0) or float(((a == 0) != (b == 0)).float().mean()) < 1e-4     # the same elements dropped
        zeros += int((b == 0).sum())
    assert zeros > 1000
    assert torch.allclose(out[1][2], out[0][2], rtol=2e-5, atol=2e-5)
    # Gradients: a pre-activation within rounding of zero may take different sides of its ReLU in the two runners (their sums differ in
    # order); that switches ONE sample's contribution to one row of the layer's weight gradient on or off -- seeds 10 and 11 of this
    # batch show it in Filter's `objects` layers, 7 in `actions`, 8 and 9 nowhere (tools/scratch/drop_cmp.py); the layer below sees a
    # rank-one change of its weight gradient.  So: every tensor within 3 % in L2 and 10 % of its largest entry anywhere -- a missing
    # 1 / (1 - p) at any site would be 25-33 % -- and most tensors within the elementwise bound of the dropout-free comparison.
    gmax = max(float(g.abs().max()) for g in out[0][3].values())
    tight = 0
    for n, g in out[0][3].items():
        d = (out[1][3][n] - g).abs()
        scale = max(float(g.abs().max()), 1e-3 * gmax)
        assert float(d.norm()) <= 3e-2 * max(float(g.norm()), 1e-3 * gmax), n
        assert float(d.max()) <= 0.1 * scale, n
        tight += float(d.max()) < 4e-4 * scale
    assert tight >= 0.8 * len(out[0][3]), tight


def test_ragged_batch_through_the_fused_operators():
    """Clips of different lengths in one launch batch (padded to the longest): each question equals its solo run."""
    config = dict(spec.DEFAULT_CONFIG)
    model = _model(config, 5)
    lens = [64, 17, 40, 64, 33, 8, 51, 64, 29, 12, 64, 45]
    qs = [synth.make_question(config, 6, i, form=f, T=lens[i]) for i, f in enumerate(synth.ALL_FORMS)]
    batch = model.forward_batch(qs).logits.cpu().clone()
    for i, q in enumerate(qs):
        solo = model.forward_batch([q]).logits.cpu()
        assert float((batch[i] - solo[0]).abs().max()) < 2e-5, (q['form'], lens[i])


@pytest.mark.parametrize('kind', ['cat2', 'xor', 'exists'])
@pytest.mark.parametrize('n,two_layers', [(1, True), (63, False), (64, True), (65, True), (200, False), (1044, True)])
def test_vector_level_module_as_one_tile_launch(kind, n, two_layers):
    """Compare / Equals / ToAction ('cat2'), Xor, Exists (/root/reference/video_nmn/modules.py:15-37,59-72,102-120,141-159):
    pack + Linear + ReLU (+ Linear + ReLU) on 64 instances per tile, the concatenation never materialised for the GEMM; against
    fp64 of the same formula.  Rows of `out` that no instance writes stay untouched; the saved concatenation and activations are
    what the backward pass reads."""
    from stair_amd import ops
    H = 512
    g = torch.Generator().manual_seed(n * 7 + len(kind))
    slots = n + 9
    vec = torch.randn(slots, H, generator=g)
    ia = torch.randint(0, slots, (n,), generator=g, dtype=torch.int32)
    ib = torch.randint(0, slots, (n,), generator=g, dtype=torch.int32)
    io = torch.randperm(slots, generator=g)[:n].to(torch.int32)
    nseg = 2 if kind == 'cat2' else 3
    w1 = torch.randn(H, nseg * H, generator=g) / (nseg * H) ** 0.5; b1 = torch.randn(H, generator=g) * 0.1
    w2 = torch.randn(H, H, generator=g) / H ** 0.5; b2 = torch.randn(H, generator=g) * 0.1
    d = lambda t: t.to(DEV)
    out0 = torch.randn(slots, H, generator=g)
    out = d(out0.clone())
    layers = [(d(w1), d(b1), 'relu')] + ([(d(w2), d(b2), 'relu')] if two_layers else [])
    cat, saves = ops.vec_mlp(kind, d(vec), d(ia), d(vec), d(ib), layers, out, d(io), save=True)
    a, b = vec[ia.long()].double(), vec[ib.long()].double()
    ref_cat = {'cat2': torch.cat([a, b], 1), 'xor': torch.cat([(a - b).abs(), a, b], 1), 'exists': torch.cat([a, b, a * b], 1)}[kind]
    h1 = torch.relu(ref_cat @ w1.double().t() + b1.double())
    ref = torch.relu(h1 @ w2.double().t() + b2.double()) if two_layers else h1
    assert float((cat.cpu().double() - ref_cat).abs().max()) == 0.0 or kind != 'cat2'
    assert float((cat.cpu().double() - ref_cat).abs().max()) < 1e-6
    assert float((saves[0].cpu().double() - h1).abs().max()) < 3e-5
    got = out.cpu().double()
    assert float((got[io.long()] - ref).abs().max()) < 3e-5
    untouched = torch.ones(slots, dtype=torch.bool); untouched[io.long()] = False
    assert torch.equal(out.cpu()[untouched], out0[untouched])
    out2 = d(out0.clone())
    ops.vec_mlp(kind, d(vec), d(ia), d(vec), d(ib), layers, out2, d(io), save=False)      # inference form: no saves, direct layer boundary
    assert float((out2.cpu().double()[io.long()] - ref).abs().max()) < 3e-5


@pytest.mark.parametrize('n,T,exclusive', [(3, 64, False), (300, 64, False), (70, 40, False), (300, 64, True), (9, 33, True)])
def test_temporal_backward_chain_matches_autograd(n, T, exclusive):
    """Temporal's backward as one chain per tile (stair_tile_mlp_args.ln_bwd + ROWSCALE_ADJ; /root/reference/video_nmn/modules.py:310-327
    under autograd: y = LayerNorm(ReLU(Lin(r_t feat_t)))) against fp64 autograd of the same formula: dZ of the dense layer, the
    gradient of the (shared) feature tiles, of the per-frame scales, of the LayerNorm parameters.  300 tiles: several tiles per
    workgroup (the fixed-point accumulators), shared input tiles (atomic accumulation)."""
    from stair_amd import ops
    H = 512
    g = torch.Generator().manual_seed(11 * n + T)
    nf = n + 1 if exclusive else max(2, n // 3)             # feature tiles are shared by several instances (atomic accumulation)
    feat = torch.randn(nf, T, H, generator=g)               # ... or every instance has its own (read - add - write)
    fidx = torch.randperm(nf, generator=g)[:n].to(torch.int32) if exclusive else torch.randint(0, nf, (n,), generator=g, dtype=torch.int32)
    rs = torch.rand(n + 4, T, generator=g)
    ridx = torch.randperm(n + 4, generator=g)[:n].to(torch.int32)
    w = torch.randn(H, H, generator=g) / H ** 0.5; b = torch.randn(H, generator=g) * 0.1
    gamma = 1.0 + 0.1 * torch.randn(H, generator=g); beta = 0.1 * torch.randn(H, generator=g)
    dy = torch.randn(n + 2, T, H, generator=g)
    yidx = torch.randperm(n + 2, generator=g)[:n].to(torch.int32)
    F = feat.double().requires_grad_(True); R = rs.double().requires_grad_(True)
    G = gamma.double().requires_grad_(True); Bt = beta.double().requires_grad_(True)
    z = (R[ridx.long()].unsqueeze(-1) * F[fidx.long()]) @ w.double().t() + b.double()
    z.retain_grad()
    a = torch.relu(z)
    y = torch.nn.functional.layer_norm(a, (H,), G, Bt, 1e-5)
    (y * dy[yidx.long()].double()).sum().backward()
    d = lambda t: t.to(DEV)
    dfeat0 = torch.randn(nf, T, H, generator=g); drs0 = torch.randn(n + 4, T, generator=g)
    dgamma0 = torch.randn(H, generator=g); dbeta0 = torch.randn(H, generator=g)
    dfeat, drs, dgamma, dbeta = d(dfeat0.clone()), d(drs0.clone()), d(dgamma0.clone()), d(dbeta0.clone())
    dz = ops.tile_temporal_bwd(d(dy), d(a.detach().float()), d(gamma), d(w), d(feat), d(rs), dfeat, drs, dgamma, dbeta,
                               dy_idx=d(yidx), feat_idx=d(fidx), rs_idx=d(ridx), dfeat_idx=d(fidx), exclusive=exclusive)
    def close(got, ref, what):
        err = float((got.cpu().double() - ref).abs().max())
        assert err < 3e-5 * max(1.0, float(ref.abs().max())), (what, err)
    close(dz, z.grad, 'dZ')
    close(dfeat - d(dfeat0), F.grad, 'dfeat')
    close(drs - d(drs0), R.grad, 'drs')
    close(dgamma - d(dgamma0), G.grad, 'dgamma')
    close(dbeta - d(dbeta0), Bt.grad, 'dbeta')
