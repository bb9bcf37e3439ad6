"""csrc/vec_group.hip (stair_vec_group): the row-wise Linear layers of a program level -- Compare / Equals / Xor / ToAction / Exists
(/root/reference/video_nmn/modules.py:15-37, 59-72, 102-120, 141-159), Filter's dense layer, Localize's keyword projection, the
decoder (module_net.py:49-53) -- as problems of ONE launch.  Checked here through the C ABI against fp64 (forward forms, ragged row
counts, column and reduction tails, gather / scatter) and against torch autograd of the same layers (adjoint forms); the whole-path
tests (test_gpu_parity / test_gpu_train) run the plan with the operator in place against the oracle and the reference fixtures."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H = 512


def _planes(W, nin):
    """[N, nin * 512] fp32 -> fragment-order planes, one [512 x 512] image per (block of 512 output rows, input segment)"""
    from stair_amd import ops
    N = W.shape[0]
    imgs = [ops.pack_wfrag(W[jb * 512:(jb + 1) * 512, s * 512:(s + 1) * 512].contiguous()) for jb in range(N // 512) for s in range(nin)]
    return torch.cat([i.reshape(-1) for i in imgs])


def _cat(kind, a, b):
    return {'a': a, 'cat2': torch.cat([a, b], 1), 'xor': torch.cat([(a - b).abs(), a, b], 1), 'exists': torch.cat([a, b, a * b], 1)}[kind]


@pytest.mark.parametrize('rows', [1, 13, 64, 130])
def test_forward_forms_match_fp64_in_one_launch(rows):
    from stair_amd import ops
    g = torch.Generator().manual_seed(rows)
    arena = torch.randn(300, H, generator=g)
    ia = torch.randint(0, 300, (rows,), generator=g, dtype=torch.int32)
    ib = torch.randint(0, 300, (rows,), generator=g, dtype=torch.int32)
    io = torch.randperm(300, generator=g)[:rows].to(torch.int32)
    d = lambda t: t.to(DEV)
    A, IA, IB, IO = d(arena), d(ia), d(ib), d(io)
    probs, want = [], []
    for kind, nseg, N, act in (('cat2', 2, 512, 'relu'), ('xor', 3, 512, 'relu'), ('exists', 3, 512, 'relu'), ('a', 1, 512, None),
                               ('cat2', 2, 1024, 'relu'), ('cat2', 2, 172, None), ('a', 1, 172, 'relu')):
        W = torch.randn(N, nseg * H, generator=g) / (nseg * H) ** 0.5
        bias = torch.randn(N, generator=g)
        out = torch.full((300, N), -7.0, device=DEV)
        save = torch.zeros(rows, nseg * H, device=DEV)
        planes = _planes(d(W), nseg) if N % 512 == 0 else None         # (the fp32-row path serves the column tail: N = 172)
        probs.append(dict(kind='fwd', rows=rows, a=A, b=A, ia=IA, ib=IB, pack=kind, W=d(W), planes=planes, bias=d(bias), N=N, act=act, out=out, io=IO,
                          in_save=save))
        x = _cat(kind, arena[ia.long()].double(), arena[ib.long()].double())
        y = x @ W.double().t() + bias.double()
        want.append((x, y.relu() if act else y, out, save))
    ops.vec_group(probs)
    torch.cuda.synchronize()
    for x, y, out, save in want:
        got = out.cpu().double()
        assert float((got[io.long()] - y).abs().max()) < 1e-4
        untouched = torch.ones(300, dtype=torch.bool); untouched[io.long()] = False
        assert float(got[untouched].min()) == -7.0 and float(got[untouched].max()) == -7.0
        assert torch.equal(save.cpu().double(), x.float().double())          # the kept input rows are the formed values, exactly
    # a row's result does not depend on the other rows of the launch, and a launch is bit-reproducible
    first = [w[2].clone() for w in want]
    ops.vec_group(probs)
    torch.cuda.synchronize()
    assert all(torch.equal(a_, w[2]) for a_, w in zip(first, want))
    if rows > 1:
        solo = dict(probs[2], rows=1, out=torch.zeros(300, 512, device=DEV), in_save=None)
        ops.vec_group([solo])
        torch.cuda.synchronize()
        assert torch.equal(solo['out'][io[0].long()], first[2][io[0].long()])


def test_short_reduction_rows_mask_epilogue_and_accumulate():
    """The decoder's last layer seen from behind: d(hidden) = (dlogits [n, 172] W3) * relu'(hidden) -- a reduction length that is
    neither a multiple of 16 nor of 8 -- and an accumulating epilogue (float atomics into rows that already hold values)."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(3)
    n, A = 77, 172
    dl = torch.randn(n, A, generator=g)
    Wt = torch.randn(1024, A, generator=g) / A ** 0.5            # the transposed image of the [172, 1024] weight
    hid = torch.randn(n, 1024, generator=g)
    out = torch.ones(n, 1024, device=DEV)
    d = lambda t: t.to(DEV)
    ops.vec_group([dict(kind='fwd', rows=n, a=d(dl), pack='a', kred=A, W=d(Wt), N=1024, act=('mask', d(hid), 1.25), out=out, accumulate=True)])
    want = 1.0 + (dl.double() @ Wt.double().t()) * (hid > 0).double() * 1.25
    assert float((out.cpu().double() - want).abs().max()) < 1e-4


@pytest.mark.parametrize('kind,nseg', [('cat2', 2), ('xor', 3), ('exists', 3)])
def test_adjoint_forms_match_autograd(kind, nseg):
    """backward of out_i = relu(cat(a_i, b_i) W^T + bias) given the gradient rows of out: relu' applied on load (kept as dZ), the
    product with the transposed weight, the adjoint of the concatenation added into the operands' gradient rows (shared rows)."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(11 + nseg)
    rows = 93
    arena = torch.randn(40, H, generator=g, dtype=torch.float64).requires_grad_(True)      # 93 instances over 40 rows: fan-in
    ia = torch.randint(0, 40, (rows,), generator=g)
    ib = torch.randint(0, 40, (rows,), generator=g)
    W = (torch.randn(H, nseg * H, generator=g, dtype=torch.float64) / (nseg * H) ** 0.5)
    bias = torch.randn(H, generator=g, dtype=torch.float64)
    y = (_cat(kind, arena[ia], arena[ib]) @ W.t() + bias).relu()
    gy = torch.randn(rows, H, generator=g, dtype=torch.float64)
    y.backward(gy)
    d = lambda t: t.float().to(DEV)
    garena = torch.zeros(40, H, device=DEV)
    dz = torch.zeros(rows, H, device=DEV)
    A = d(arena.detach())
    Wt = d(W.t().contiguous())
    ops.vec_group([dict(kind='adj', rows=rows, a=d(gy), b=d(y.detach()), pack='mask', in_scale=1.0, in_save=dz, W=Wt, planes=_planes(Wt, 1), N=nseg * H,
                        adj=kind, fa=A, fb=A, fia=ia.to(torch.int32).to(DEV), fib=ib.to(torch.int32).to(DEV), ga=garena, gb=garena)])
    assert float((dz.cpu().double() - gy * (y.detach() > 0)).abs().max()) < 1e-6
    ref = arena.grad
    assert float((garena.cpu().double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))


def test_decoder_shaped_adjoint_with_two_input_segments():
    """d(cat[root, qfeat]) = dHidden [n, 1024] W0 through the [1024, 1024] transposed image: two input segments, two output blocks,
    the CAT2 adjoint into two different gradient buffers."""
    from stair_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 70
    gh = torch.randn(n, 1024, generator=g)
    W0 = torch.randn(1024, 1024, generator=g) / 32.0
    roots = torch.randperm(200, generator=g)[:n].to(torch.int32)
    gvec = torch.zeros(200, H, device=DEV)
    gq = torch.zeros(n, H, device=DEV)
    d = lambda t: t.to(DEV)
    GH = d(gh)
    W0t = d(W0.t().contiguous())
    ops.vec_group([dict(kind='adj', rows=n, a=GH, b=GH[:, 512:], lda=1024, ldb=1024, pack='cat2', W=W0t, planes=_planes(W0t, 2), N=1024, adj='cat2',
                        fia=d(roots), ga=gvec, gb=gq)])
    want = gh.double() @ W0.double()
    assert float((gvec.cpu().double()[roots.long()] - want[:, :512]).abs().max()) < 2e-4
    assert float((gq.cpu().double() - want[:, 512:]).abs().max()) < 2e-4
