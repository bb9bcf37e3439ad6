"""GPU tests of the slab-reduced weight-gradient product (csrc/gemm_tn_x3tr.hip, stair_gemm_tn_slabs): dW += dZ^T (rs * X) with X
gathered per group of rows, db += colsum(dZ) -- the autograd of the module-level nn.Linear layers
(/root/reference/video_nmn/modules.py, train_module.py:408).  Reference = fp64 of the same operands; the split products' error is
~4e-6 of a term, accumulated in fp32.  The result must not depend on the run (no atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(M, N, K, R, seed, gather=True, scale=True):
    g = torch.Generator().manual_seed(seed)
    G = M // R
    slots = G + 3
    X = torch.randn(slots, R, K, generator=g)
    idx = torch.randperm(slots, generator=g)[:G].to(torch.int32) if gather else None
    dZ = torch.randn(M, N, generator=g)
    rs = torch.rand(slots, R, generator=g) if scale else None
    C0 = torch.randn(N, K, generator=g)
    b0 = torch.randn(N, generator=g)
    return X, idx, dZ, rs, C0, b0


@pytest.mark.parametrize('M,N,K,R', [(64, 256, 128, 32), (64 * 13, 512, 512, 64), (64 * 273, 512, 512, 64), (32 * 1001, 256, 384, 32),
                                     (64 * 1704, 512, 512, 64), (96 * 40, 512, 256, 96)])
@pytest.mark.parametrize('gather,scale', [(True, True), (True, False), (False, False), (False, True)])
def test_slab_product_matches_fp64(M, N, K, R, gather, scale):
    from stair_amd import ops
    X, idx, dZ, rs, C0, b0 = _case(M, N, K, R, M + N + K, gather, scale)
    d = lambda t: t.to(DEV) if t is not None else None
    Cm, b1 = d(C0.clone()), d(b0.clone())
    G = M // R
    Xd = d(X) if gather else d(X[:G].contiguous())
    ops.gemm_tn(d(dZ), Xd, Cm, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=d(idx), row_scale=d(rs) if gather else d(rs[:G].contiguous()) if scale else None,
                rs_gstride=R, rs_gidx=d(idx) if scale else None, colsum=b1, deterministic=True)
    sel = idx.long() if gather else torch.arange(G)
    Xg = X[sel]
    if scale:
        Xg = Xg * rs[sel].unsqueeze(-1)
    ref = (C0.double().to(DEV) + d(dZ).double().t() @ d(Xg.reshape(M, K)).double()).cpu()
    tol = 4e-4 * max(1.0, (M / 1000) ** 0.5 * 3)          # the bound of test_gemm_tn_weight_gradient in the split mode
    assert float((Cm.cpu().double() - ref).abs().max()) < tol
    refb = b0.double() + dZ.double().sum(0)
    assert float((b1.cpu().double() - refb).abs().max()) < 4e-5 * max(1.0, (M / 100) ** 0.5)


def test_slab_product_is_bit_reproducible_and_equals_the_atomic_kernel_closely():
    from stair_amd import ops
    M, N, K, R = 64 * 754, 512, 512, 64
    X, idx, dZ, rs, C0, b0 = _case(M, N, K, R, 5)
    d = lambda t: t.to(DEV)
    dZd, Xd, idxd = d(dZ), d(X), d(idx)
    outs = []
    for _ in range(3):
        Cm, b1 = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ops.gemm_tn(dZd, Xd, Cm, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=idxd, colsum=b1, deterministic=True)
        outs.append((Cm.cpu(), b1.cpu()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    Ca, ba = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    ops.gemm_tn(dZd, Xd, Ca, M, N, K, rows_per_group=R, b_gstride=R * K, b_gidx=idxd, colsum=ba)
    assert float((Ca.cpu() - outs[0][0]).abs().max()) < 2e-3            # same products, different summation order


def test_unsupported_shapes_are_refused():
    from stair_amd import ops
    A = torch.zeros(64, 100, device=DEV); B = torch.zeros(64, 128, device=DEV); Cm = torch.zeros(100, 128, device=DEV)
    with pytest.raises(RuntimeError, match='shape not supported'):
        ops.gemm_tn(A, B, Cm, 64, 100, 128, rows_per_group=32, deterministic=True)


def test_step_gradients_with_every_map_level_product_on_the_slab_kernel():
    """The per-bucket products of stair_plan_backward (FilterFrame's dense layer is used by up to three buckets of a step) take the
    slab kernel from 4096 rows on; lowered to 64 rows, the six-question full-size gradient check runs every one of them through it:
    several products into ONE weight are queued before the single reduction launch, which must add them one after the other."""
    from stair_amd._lib import lib, check
    import test_gpu_train
    check(lib.stair_set_tn_slab_min_rows(64))
    try:
        test_gpu_train.test_full_size_backward_sample('bf16x3')
    finally:
        check(lib.stair_set_tn_slab_min_rows(4096))
