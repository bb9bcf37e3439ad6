"""Thin tensor-level wrappers over the C ABI (include/stair_hip.h).

torch is used only for device memory and the current stream; all arithmetic happens in
libstair_hip.so.  Every function requires CUDA(ROCm) float32 contiguous tensors and raises otherwise
-- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import GemmArgs, GemmPlanesArgs, GemmTnArgs, LstmArgs, LstmBwdArgs, TileMlpArgs, check, lib

ACT = {None: 0, 'none': 0, 'relu': 1, 'sigmoid': 2}
MATMUL_MODES = {'f32': 0, 'bf16x3': 1, 'bf16': 2}


def set_matmul_mode(mode, split_min_rows=1):
    """'f32' (exact fp32 MFMA), 'bf16x3' (split-precision bf16 MFMA, the default, inside the 1e-4 logit budget) or
    'bf16' (one bf16 product per operand pair: BASELINE configs[1]'s precision, top-1 identity only)."""
    check(lib.stair_set_matmul_mode(MATMUL_MODES[mode]))
    check(lib.stair_set_split_min_rows(split_min_rows))


def get_matmul_mode():
    return {v: k for k, v in MATMUL_MODES.items()}[lib.stair_get_matmul_mode()]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a tensor on the GPU (the HIP path has no CPU fallback)' % name)
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError('%s must be contiguous %s' % (name, dtype))
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _splitk_scratch(args, M, N, K, device):
    """Scratch that lets a small launch split its K loop (stair_gemm_args.splitk_ws); kept alive by the caller."""
    ws = torch.empty(max(1, -(-K // 128)) * M * N, device=device, dtype=torch.float32)
    args.splitk_ws, args.splitk_ws_floats = ws.data_ptr(), ws.numel()
    return ws


def linear(x, weight, bias=None, act=None, splitk=False):
    """act(x @ weight.T + bias) through stair_gemm_f32.  x [..., K] -> [..., N].  splitk: hand the launch the scratch a
    plan gives its vector-level layers (small launches then split K; deterministic)."""
    _req(x, 'x'); _req(weight, 'weight')
    K = x.shape[-1]
    N = weight.shape[0]
    assert weight.shape[1] == K
    x2 = x.reshape(-1, K)
    out = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
    a = GemmArgs()
    a.A, a.lda, a.a_gstride, a.a_gidx = x2.data_ptr(), K, K, None
    a.W, a.ldw, a.bias = weight.data_ptr(), K, (bias.data_ptr() if bias is not None else None)
    a.C, a.ldc, a.c_gstride, a.c_gidx = out.data_ptr(), N, N, None
    a.row_scale, a.rs_gstride, a.rs_gidx = None, 0, None
    a.groups, a.rows_per_group, a.N, a.K, a.act = x2.shape[0], 1, N, K, ACT[act]
    ws = _splitk_scratch(a, x2.shape[0], N, K, x.device) if splitk else None
    check(lib.stair_gemm_f32(C.byref(a), _stream()))
    return out.reshape(*x.shape[:-1], N)


def gemm_grouped(A, a_gstride, a_gidx, W, bias, Cmat, c_gstride, c_gidx, groups, rows_per_group, N, K, act=None,
                 lda=None, ldc=None, row_scale=None, rs_gstride=0, rs_gidx=None, accumulate=False, splitk=False):
    """Raw grouped form (see stair_gemm_args)."""
    a = GemmArgs()
    a.A, a.lda, a.a_gstride, a.a_gidx = A.data_ptr(), lda or K, a_gstride, (a_gidx.data_ptr() if a_gidx is not None else None)
    a.W, a.ldw, a.bias = W.data_ptr(), K, (bias.data_ptr() if bias is not None else None)
    a.C, a.ldc, a.c_gstride, a.c_gidx = Cmat.data_ptr(), ldc or N, c_gstride, (c_gidx.data_ptr() if c_gidx is not None else None)
    a.row_scale = row_scale.data_ptr() if row_scale is not None else None
    a.rs_gstride, a.rs_gidx = rs_gstride, (rs_gidx.data_ptr() if rs_gidx is not None else None)
    a.groups, a.rows_per_group, a.N, a.K, a.act = groups, rows_per_group, N, K, ACT[act]
    a.accumulate = 1 if accumulate else 0
    ws = _splitk_scratch(a, groups * rows_per_group, N, K, A.device) if splitk else None
    check(lib.stair_gemm_f32(C.byref(a), _stream()))


def split_planes(x, lo=True):
    """fp32 tensor -> (hi, lo) bf16 planes with x = hi + lo + O(2^-17 |x|) (stair_split_planes); lo=False: hi only,
    i.e. plain round-to-nearest-even bf16 (the stored clip-feature format of BASELINE.json configs[1])."""
    _req(x, 'x')
    if x.numel() % 8:
        raise ValueError('element count must be a multiple of 8')
    hi = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    lo_t = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if lo else None
    check(lib.stair_split_planes(_ptr(x), _ptr(hi), _ptr(lo_t), x.numel(), _stream()))
    return (hi, lo_t) if lo else hi


def split_planes_tiled(w):
    """[N, K] fp32 weight -> (hi, lo) bf16 planes in the tiled layout [K/32, N, 32] (stair_split_planes_tiled)."""
    _req(w, 'w')
    N, K = w.shape
    if K % 32:
        raise ValueError('K must be a multiple of 32')
    hi = torch.empty(K // 32, N, 32, device=w.device, dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    check(lib.stair_split_planes_tiled(_ptr(w), _ptr(hi), _ptr(lo), N, K, _stream()))
    return hi, lo


def gemm_planes(a_hi, a_lo, w_hi, w_lo, bias=None, act=None, out=None, w_frag_rows=None):
    """act((a_hi + a_lo) @ (w_hi + w_lo).T + bias) on bf16 planes staged by LDS-DMA (stair_gemm_planes).
    a_hi [M,K] bf16, a_lo None (A exact in bf16: two MFMA products per pair) or [M,K]; w_hi, w_lo [N,K].
    w_frag_rows = N: w_hi is ONE fragment-order image of the [N, K] weight (ops.pack_wfrag), w_lo is ignored (pass w_hi): the
    waves load their own W fragments global -> VGPR (stair_gemm_planes_args.w_tiled == 2; a_lo None, no activation)."""
    _req(a_hi, 'a_hi', torch.bfloat16); _req(w_hi, 'w_hi', torch.bfloat16); _req(w_lo, 'w_lo', torch.bfloat16)
    if a_lo is not None:
        _req(a_lo, 'a_lo', torch.bfloat16)
    M, K = a_hi.shape
    if w_frag_rows is not None:
        N = int(w_frag_rows)
        assert w_hi.numel() == 2 * N * K
        if out is None:
            out = torch.empty(M, N, device=a_hi.device, dtype=torch.float32)
        a = GemmPlanesArgs()
        a.A_hi, a.A_lo, a.lda = a_hi.data_ptr(), None, K
        a.W_hi, a.W_lo, a.ldw = w_hi.data_ptr(), w_hi.data_ptr(), K
        a.bias = bias.data_ptr() if bias is not None else None
        a.C, a.ldc = out.data_ptr(), out.stride(0)
        a.M, a.N, a.K, a.act, a.w_tiled = M, N, K, ACT[act], 2
        check(lib.stair_gemm_planes(C.byref(a), _stream()))
        return out
    tiled = w_hi.dim() == 3
    N = w_hi.shape[1] if tiled else w_hi.shape[0]
    assert w_hi.shape == w_lo.shape == ((K // 32, N, 32) if tiled else (N, K))
    if out is None:
        out = torch.empty(M, N, device=a_hi.device, dtype=torch.float32)
    a = GemmPlanesArgs()
    a.A_hi, a.A_lo, a.lda = a_hi.data_ptr(), (a_lo.data_ptr() if a_lo is not None else None), K
    a.W_hi, a.W_lo, a.ldw = w_hi.data_ptr(), w_lo.data_ptr(), K
    a.bias = bias.data_ptr() if bias is not None else None
    a.C, a.ldc = out.data_ptr(), out.stride(0)
    a.M, a.N, a.K, a.act, a.w_tiled = M, N, K, ACT[act], 1 if tiled else 0
    check(lib.stair_gemm_planes(C.byref(a), _stream()))
    return out


def gemm_tn(A, B, Cmat, M, N, K, rows_per_group=1, b_gstride=None, b_gidx=None, row_scale=None, rs_gstride=0,
            rs_gidx=None, colsum=None, colsum2=None, lda=None, deterministic=False):
    """Cmat[N,K] += A[M,N]^T @ (rs * B[M,K]) -- weight gradients (see stair_gemm_tn_args); colsum[N] (and
    colsum2) += column sums of A, the bias gradient of the same layer."""
    a = GemmTnArgs()
    a.A, a.lda = A.data_ptr(), (lda if lda is not None else N)      # lda > N: A is a column block of a wider matrix
    a.b_is_bf16 = 1 if B.dtype == torch.bfloat16 else 0        # stored clip features: exact bf16 rows, two products per pair
    a.B, a.ldb, a.b_gstride = B.data_ptr(), K, (b_gstride if b_gstride is not None else K * rows_per_group)
    a.b_gidx = b_gidx.data_ptr() if b_gidx is not None else None
    a.row_scale = row_scale.data_ptr() if row_scale is not None else None
    a.rs_gstride, a.rs_gidx = rs_gstride, (rs_gidx.data_ptr() if rs_gidx is not None else None)
    a.C, a.ldc = Cmat.data_ptr(), K
    a.M, a.rows_per_group, a.N, a.K = M, rows_per_group, N, K
    a.colsum = colsum.data_ptr() if colsum is not None else None
    a.colsum2 = colsum2.data_ptr() if colsum2 is not None else None
    if deterministic:           # slab partials + fixed-order reduction (stair_gemm_tn_slabs): bit-identical from run to run
        n = lib.stair_gemm_tn_slabs_scratch(M, N, K)
        scratch = torch.empty(n, device=A.device, dtype=torch.float32)
        check(lib.stair_gemm_tn_slabs(C.byref(a), scratch.data_ptr(), n, _stream()))
        return
    check(lib.stair_gemm_tn_f32(C.byref(a), _stream()))


def lstm_bidir(x, seq_off, max_len, weights, save=False, coop=True, seq_len=None, x_planes=False):
    """Bidirectional LSTM over packed ragged sequences.

    x [rows, I]; seq_off int32 [n+1] (device); weights = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r).
    Returns (out [rows, 2*Hh], h_n [n, 2*Hh]); with save=True also (gates, cbuf) for lstm_bidir_bwd.
    seq_len: int32 [n] (device) when the storage is padded -- sequence s holds seq_len[s] data rows of its span.
    x_planes: fp32 rows only -- the input projection as ONE plane GEMM on zero-padded hi / lo planes of x and W_ih
    (stair_lstm_args.x_planes_ws: how the plan runs the text encoder, E = 300).
    """
    bf = x.dtype == torch.bfloat16           # stored bf16 input rows (clip features): plane GEMM input projection
    _req(x, 'x', torch.bfloat16 if bf else torch.float32); _req(seq_off, 'seq_off', torch.int32)
    for w in weights:
        _req(w, 'lstm weight')
    rows, I = x.shape
    n = seq_off.numel() - 1
    Hh = weights[1].shape[1]
    out = torch.empty(rows, 2 * Hh, device=x.device, dtype=torch.float32)
    h_n = torch.empty(n, 2 * Hh, device=x.device, dtype=torch.float32)
    xproj = torch.empty(rows, 8 * Hh, device=x.device, dtype=torch.float32)
    bias_ws = torch.empty(8 * Hh, device=x.device, dtype=torch.float32)
    pack_ws = torch.empty(8 * Hh * Hh, device=x.device, dtype=torch.float32)
    a = LstmArgs()
    a.x, a.ldx, a.rows, a.n, a.max_len, a.I, a.Hh = (None if bf else x.data_ptr()), I, rows, n, max_len, I, Hh
    if bf:
        planes = torch.empty(2 * 8 * Hh * I, device=x.device, dtype=torch.bfloat16)
        a.x_bf16, a.wih_planes_ws = x.data_ptr(), planes.data_ptr()
    elif x_planes:
        Ip = (I + 31) // 32 * 32
        planes = torch.empty(2 * 8 * Hh * Ip, device=x.device, dtype=torch.bfloat16)
        xpl = torch.empty(2 * rows * Ip, device=x.device, dtype=torch.bfloat16)
        a.wih_planes_ws, a.x_planes_ws = planes.data_ptr(), xpl.data_ptr()
    a.seq_off = seq_off.data_ptr()
    if seq_len is not None:
        a.seq_len = seq_len.data_ptr()
    if coop and Hh == 256:       # scratch for the cooperative recurrence (hidden units split over co-resident workgroups)
        coop_ws = torch.empty(int(lib.stair_lstm_coop_ws_bytes(n)), device=x.device, dtype=torch.uint8)
        a.coop_ws, a.coop_ws_bytes = coop_ws.data_ptr(), coop_ws.numel()
    for d in range(2):
        a.w_ih[d], a.w_hh[d] = weights[4 * d].data_ptr(), weights[4 * d + 1].data_ptr()
        a.b_ih[d], a.b_hh[d] = weights[4 * d + 2].data_ptr(), weights[4 * d + 3].data_ptr()
    a.xproj_ws, a.bias_ws, a.whh_pack_ws = xproj.data_ptr(), bias_ws.data_ptr(), pack_ws.data_ptr()
    a.out, a.ldo, a.h_n = out.data_ptr(), 2 * Hh, h_n.data_ptr()
    cbuf = torch.empty(rows, 2 * Hh, device=x.device, dtype=torch.float32) if save else None
    a.cbuf = cbuf.data_ptr() if save else None
    check(lib.stair_lstm_bidir_fwd(C.byref(a), _stream()))
    return (out, h_n, xproj, cbuf) if save else (out, h_n)


def lstm_bidir_bwd(x, seq_off, max_len, weights, out, gates, cbuf, d_out, d_hn=None, coop=True, seq_len=None):
    """Gradients of (w_ih, w_hh, b_ih, b_hh) x 2 directions given d_out [rows, 2Hh] and d_hn [n, 2Hh].
    `gates` (from lstm_bidir(save=True)) is overwritten.  coop: use the cooperative BPTT kernel where it applies
    (Hh = 256, split matmul modes); seq_len: per-sequence lengths of padded storage (int32 [n])."""
    rows, I = x.shape
    n = seq_off.numel() - 1
    Hh = weights[1].shape[1]
    grads = [torch.zeros_like(w) for w in weights]
    a = LstmBwdArgs()
    bf = x.dtype == torch.bfloat16
    a.x, a.ldx, a.rows, a.n, a.max_len, a.I, a.Hh = (None if bf else x.data_ptr()), I, rows, n, max_len, I, Hh
    if bf:
        a.x_bf16 = x.data_ptr()
    a.seq_off = seq_off.data_ptr()
    pack_ws = torch.empty(8 * Hh * Hh, device=x.device, dtype=torch.float32)
    hprev = torch.empty(rows, 2 * Hh, device=x.device, dtype=torch.float32)
    for d in range(2):
        a.w_hh[d] = weights[4 * d + 1].data_ptr()
        a.dw_ih[d], a.dw_hh[d] = grads[4 * d].data_ptr(), grads[4 * d + 1].data_ptr()
        a.db_ih[d], a.db_hh[d] = grads[4 * d + 2].data_ptr(), grads[4 * d + 3].data_ptr()
    a.gates, a.cbuf, a.out, a.ldo = gates.data_ptr(), cbuf.data_ptr(), out.data_ptr(), 2 * Hh
    a.d_out, a.ldd = d_out.data_ptr(), 2 * Hh
    a.d_hn = d_hn.data_ptr() if d_hn is not None else None
    a.whh_pack_ws, a.hprev_ws = pack_ws.data_ptr(), hprev.data_ptr()
    if seq_len is not None:
        a.seq_len = seq_len.data_ptr()
    if coop and Hh == 256:
        coop_ws = torch.empty(int(lib.stair_lstm_coop_bwd_ws_bytes(n)), device=x.device, dtype=torch.uint8)
        a.coop_ws, a.coop_ws_bytes = coop_ws.data_ptr(), coop_ws.numel()
    check(lib.stair_lstm_bidir_bwd(C.byref(a), _stream()))
    return grads


def l2normalize(x):
    """x / max(||x||, 1e-12) over the last dim (module_net.py:211-216 applied row-wise)."""
    _req(x, 'x')
    H = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.stair_l2normalize_fwd(_ptr(x), _ptr(out), x.numel() // H, H, _stream()))
    return out


def cosine_attn(F, f_idx, Kmat, k_idx, npairs, T, H, out=None, out_idx=None):
    """(cos + 1) * 0.49 rows; F [G,T,H], Kmat [*,H]; returns att [npairs, T] unless `out` is given."""
    _req(F, 'F'); _req(Kmat, 'Kmat')
    if out is None:
        out = torch.empty(npairs, T, device=F.device, dtype=torch.float32)
    check(lib.stair_cosine_attn_fwd(_ptr(F), T * H, _ptr(f_idx), _ptr(Kmat), _ptr(k_idx), _ptr(out), _ptr(out_idx),
                                    npairs, T, H, _stream()))
    return out


def cosine_attn_bwd(F, f_idx, Kmat, k_idx, d_att, npairs, T, H, out_idx=None):
    """Adjoint of cosine_attn: returns (dF like F, dK like Kmat) for d_att [npairs (or rows of out_idx), T]."""
    _req(F, 'F'); _req(Kmat, 'Kmat'); _req(d_att, 'd_att')
    dF, dK = torch.zeros_like(F), torch.zeros_like(Kmat)
    check(lib.stair_cosine_attn_bwd(_ptr(F), T * H, _ptr(f_idx), _ptr(Kmat), _ptr(k_idx), _ptr(d_att), _ptr(out_idx), _ptr(dF), _ptr(dK),
                                    npairs, T, H, _stream()))
    return dF, dK


def cosine_topk(queries, keys, k, q_idx=None, n=None):
    """k best rows of `keys` [C,H] by cosine similarity for each query row (evaluate.py:95-98).  queries [rows,H];
    q_idx (int32, optional) picks the rows to rank.  Returns (idx [n,k] int32, sim [n,k]) best first."""
    _req(queries, 'queries'); _req(keys, 'keys')
    if queries.dim() != 2 or keys.dim() != 2 or queries.shape[1] != keys.shape[1]:
        raise ValueError('queries [rows,H] and keys [C,H] must share H')
    if q_idx is not None:
        _req(q_idx, 'q_idx', torch.int32)
    n = (q_idx.shape[0] if q_idx is not None else queries.shape[0]) if n is None else n
    Cn, H = keys.shape
    idx = torch.empty(n, k, device=queries.device, dtype=torch.int32)
    sim = torch.empty(n, k, device=queries.device, dtype=torch.float32)
    ws = torch.empty(Cn, device=queries.device, dtype=torch.float32)
    check(lib.stair_cosine_topk(_ptr(queries), queries.stride(0), _ptr(q_idx), _ptr(keys), _ptr(ws), n, Cn, H, k,
                                _ptr(idx), _ptr(sim), _stream()))
    return idx, sim


def temporal_relate(att, att_idx, att_k, n, T, mode, conv, ksize, w6):
    """Temporal relate nets; att [rows,T]; returns r [n,T]."""
    _req(att, 'att')
    out = torch.empty(n, T, device=att.device, dtype=torch.float32)
    arr = (C.c_void_p * 6)(*[(w.data_ptr() if w is not None else None) for w in w6])
    check(lib.stair_temporal_relate_fwd(_ptr(att), _ptr(att_idx), _ptr(att_k), _ptr(out), None, n, T, mode,
                                        1 if conv else 0, ksize, arr, _stream()))
    return out


def temporal_relate_bwd(att, att_idx, att_k, d_out, n, T, mode, conv, ksize, w6):
    """Adjoint of temporal_relate: returns (d_att like att, [dw0, db0, dw2, db2, dw4, db4] like w6) for d_out [n, T]."""
    _req(att, 'att'); _req(d_out, 'd_out')
    d_att = torch.zeros_like(att)
    dws = [torch.zeros_like(w) if w is not None else None for w in w6]
    arr = (C.c_void_p * 6)(*[(w.data_ptr() if w is not None else None) for w in w6])
    darr = (C.c_void_p * 6)(*[(w.data_ptr() if w is not None else None) for w in dws])
    out_idx = torch.arange(n, dtype=torch.int32, device=att.device)
    check(lib.stair_temporal_relate_bwd(_ptr(att), _ptr(att_idx), _ptr(att_k), _ptr(d_out), _ptr(out_idx), _ptr(d_att), n, T, mode,
                                        1 if conv else 0, ksize, arr, darr, _stream()))
    return d_att, dws


class kernel_accounting:
    """Context manager over stair_acct_enable / stair_acct_dump: which kernel variants the launchers selected inside the
    block, with launch counts, algorithmic bytes and algorithmic flops.  `.table` = {kernel: (launches, bytes, flops)}."""

    def __enter__(self):
        lib.stair_acct_enable(1)
        self.table = {}
        return self

    def __exit__(self, *exc):
        n = lib.stair_acct_dump(None, 0)
        buf = C.create_string_buffer(n)
        lib.stair_acct_dump(buf, n)
        lib.stair_acct_enable(0)
        for line in buf.value.decode().splitlines():
            k, c, b, f = line.split()
            self.table[k] = (int(c), int(b), int(f))
        return False


TILE_TAILS = {None: 0, 'store': 1, 'sum_rows': 2, 'cosine': 3, 'rowdot_sigmoid': 4, 'layernorm': 5}


def pack_wfrag(weight, transpose=False):
    """fp32 [N, K] weight -> bf16 hi / lo planes in MFMA fragment order for tile_mlp (stair_pack_wfrag); transpose: the planes
    of weight.T (weight is then stored [K, N])."""
    _req(weight, 'weight')
    N, K = (weight.shape[1], weight.shape[0]) if transpose else weight.shape
    planes = torch.empty(2 * N * K, dtype=torch.bfloat16, device=weight.device)
    check(lib.stair_pack_wfrag(_ptr(weight), _ptr(planes), N, K, 1 if transpose else 0, _stream()))
    return planes


def tile_mlp(x, layers, tail, x_idx=None, row_scale=None, rs_idx=None, save=False, mid_rowdot=None, **kw):
    """Fused per-clip tile operator (stair_tile_mlp_fwd): x [n_tiles, T, 512] fp32, layers = [(weight, bias, act)] with act in
    (None, 'relu'); tail in TILE_TAILS with its operands in kw (out, out_idx, out_gstride, kb, pair_first, pair_cnt, att_idx,
    att, vw, vb, extra, gamma, beta, eps, len).  mid_rowdot = (vw, vb, extra): FilterFrame's attention between layers 2 and 3.
    Returns (saves: one [cnt, T, H] tensor per layer when save=True, rs_out or None)."""
    _req(x, 'x')
    T, H = x.shape[-2], x.shape[-1]
    cnt = int(x_idx.numel()) if x_idx is not None else x.shape[0]
    a = TileMlpArgs()
    a.X, a.x_gstride, a.x_idx = x.data_ptr(), T * H, (x_idx.data_ptr() if x_idx is not None else None)
    if row_scale is not None:
        a.row_scale, a.rs_idx = row_scale.data_ptr(), (rs_idx.data_ptr() if rs_idx is not None else None)
    keep = []
    for l, (w, b, act) in enumerate(layers):
        planes = pack_wfrag(w)
        keep.append(planes)
        a.W[l], a.bias[l], a.act[l] = planes.data_ptr(), (b.data_ptr() if b is not None else None), ACT[act]
    a.n_layers = len(layers)
    saves = []
    if save:
        for l in range(len(layers)):
            sv = torch.empty(cnt, T, H, device=x.device)
            saves.append(sv)
            a.save[l] = sv.data_ptr()
    rs_out = None
    if mid_rowdot is not None:
        vw, vb, extra = mid_rowdot
        rs_out = torch.empty(cnt, T, device=x.device)
        a.mid_rowdot, a.vw, a.vb, a.extra, a.rs_out = 1, vw.data_ptr(), vb.data_ptr(), (extra.data_ptr() if extra is not None else None), rs_out.data_ptr()
    a.tail = TILE_TAILS[tail]
    for name in ('out', 'out_idx', 'kb', 'pair_first', 'pair_cnt', 'att_idx', 'att', 'gamma', 'beta', 'len'):
        if kw.get(name) is not None:
            setattr(a, name, kw[name].data_ptr())
    if mid_rowdot is None:
        for name in ('vw', 'vb', 'extra'):
            if kw.get(name) is not None:
                setattr(a, name, kw[name].data_ptr())
    a.out_gstride = int(kw.get('out_gstride', T * H))
    a.ln_eps = float(kw.get('eps', 1e-5))
    a.cnt, a.T, a.H = cnt, T, H
    check(lib.stair_tile_mlp_fwd(C.byref(a), _stream()))
    return saves, rs_out


def tile_temporal_bwd(dy, a_saved, gamma, weight, feat, rs, dfeat, drs, dgamma, dbeta, eps=1e-5, dy_idx=None, feat_idx=None,
                      rs_idx=None, dfeat_idx=None, exclusive=False):
    """Temporal's backward as one chain per tile (stair_tile_mlp_args.ln_bwd + tail ROWSCALE_ADJ; autograd of
    /root/reference/video_nmn/modules.py:310-327, y = LayerNorm(a), a = ReLU(Lin(r_t feat_t))): dy [*, T, 512] gradient tiles of y
    (tile i = dy[dy_idx[i]]), a_saved [n, T, 512] the saved post-ReLU rows, weight [512, 512] of the dense layer, feat tiles, rs
    [*, T] the per-frame scales.  ADDS r_t (dZ W)_t into dfeat[dfeat_idx[i]], (dZ W)_t . feat_t into drs[rs_idx[i]], the LayerNorm
    parameter gradients into dgamma / dbeta; returns dZ [n, T, 512] (the operand of the dense layer's weight-gradient product).
    exclusive: no two instances share a dfeat tile (plain read - add - write instead of float atomics)."""
    _req(dy, 'dy')
    n, T, H = a_saved.shape
    a = TileMlpArgs()
    a.X, a.x_gstride, a.x_idx = dy.data_ptr(), T * H, (dy_idx.data_ptr() if dy_idx is not None else None)
    a.ln_bwd, a.in_mask, a.in_mask_gstride, a.in_scale = 1, a_saved.data_ptr(), T * H, 1.0
    dz = torch.empty(n, T, H, device=dy.device)
    a.save_in = dz.data_ptr()
    a.gamma, a.dgamma, a.dbeta, a.ln_eps = gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), float(eps)
    planes = pack_wfrag(weight, transpose=True)
    a.W[0], a.act[0], a.n_layers = planes.data_ptr(), 0, 1
    a.tail = 8
    a.out, a.out_gstride, a.out_idx = dfeat.data_ptr(), T * H, (dfeat_idx.data_ptr() if dfeat_idx is not None else None)
    a.adj_feat, a.adj_feat_gstride, a.adj_feat_idx = feat.data_ptr(), T * H, (feat_idx.data_ptr() if feat_idx is not None else None)
    a.adj_rs, a.adj_rs_idx, a.adj_drs = rs.data_ptr(), (rs_idx.data_ptr() if rs_idx is not None else None), drs.data_ptr()
    a.cnt, a.T, a.H = n, T, H
    a.acc_exclusive = 1 if exclusive else 0
    check(lib.stair_tile_mlp_fwd(C.byref(a), _stream()))
    return dz


VEC_IN = {'a': 0, 'cat2': 1, 'xor': 2, 'exists': 3, 'mask': 4}


def vec_group(problems):
    """The row-wise Linear layers of a program level as ONE launch (stair_vec_group, csrc/vec_group.hip).  problems: dicts with
      kind 'fwd' | 'adj', rows, a, b, ia, ib (operand rows [*, lda] / int32 index tensors or None), pack in VEC_IN, in_scale,
      kred (default 512), W [N, ldw], bias, N, act (None | 'relu' | ('mask', emask, escale)), out, io, accumulate, in_save;
      adjoint problems: adj (the forward layer's input form), fa, fb, fia, fib, ga, gb.
    Every tensor is a contiguous float32 / int32 GPU tensor; row strides are taken from the tensors' last-but-one stride."""
    from ._lib import VecProblem
    arr = (VecProblem * len(problems))()

    def ptr(t):
        return t.data_ptr() if t is not None else None

    def ld(t, default=0):
        return int(t.stride(-2)) if t is not None and t.dim() >= 2 else default
    for q, d in zip(arr, problems):
        q.kind = 0 if d['kind'] == 'fwd' else 1
        q.rows = int(d['rows'])
        q.a, q.b, q.ia, q.ib = ptr(d['a']), ptr(d.get('b')), ptr(d.get('ia')), ptr(d.get('ib'))
        q.lda, q.ldb = int(d.get('lda', ld(d['a']))), int(d.get('ldb', ld(d.get('b'))))
        q.pack, q.in_scale, q.kred = VEC_IN[d.get('pack', 'a')], float(d.get('in_scale', 1.0)), int(d.get('kred', 512))
        q.W, q.ldw, q.bias, q.N = ptr(d['W']), int(d.get('ldw', ld(d['W']))), ptr(d.get('bias')), int(d['N'])
        q.wplanes = ptr(d.get('planes'))
        act = d.get('act')
        if isinstance(act, tuple):
            q.act, q.emask, q.ldm, q.escale = 2, ptr(act[1]), ld(act[1]), float(act[2])
        else:
            q.act = {None: 0, 'relu': 1}[act]
        q.out, q.io, q.ldo, q.accumulate = ptr(d.get('out')), ptr(d.get('io')), int(d.get('ldo', ld(d.get('out')))), int(bool(d.get('accumulate')))
        q.in_save, q.ld_save = ptr(d.get('in_save')), ld(d.get('in_save'))
        q.adj = VEC_IN[d['adj']] if d.get('adj') else 0
        q.fa, q.fb, q.fia, q.fib = ptr(d.get('fa')), ptr(d.get('fb')), ptr(d.get('fia')), ptr(d.get('fib'))
        q.ldfa, q.ldfb = int(d.get('ldfa', ld(d.get('ga'), 512))), int(d.get('ldfb', ld(d.get('gb'), 512)))
        q.ga, q.gb = ptr(d.get('ga')), ptr(d.get('gb'))
        q.gia, q.gib = ptr(d.get('gia')), ptr(d.get('gib'))
    check(lib.stair_vec_group(arr, len(problems), _stream()))


VEC_PACKS = {'cat2': 1, 'xor': 2, 'exists': 3}


def vec_mlp(kind, a_rows, a_idx, b_rows, b_idx, layers, out, out_row_idx, save=False):
    """A vector-level module as ONE launch of the fused tile operator (stair_tile_mlp_args.vec_pack): 64 instances per tile, the
    first layer over the concatenation [a, b] ('cat2': Compare / Equals / ToAction), [|a - b|, a, b] ('xor') or [a, b, a * b]
    ('exists') of the rows a = a_rows[a_idx[i]], b = b_rows[b_idx[i]] -- never materialised for the GEMM --, an optional second
    layer, and row i of the result stored at out[out_row_idx[i]].  layers = [(weight [512, 2H | 3H], bias, act), (weight [512, 512],
    bias, act)?].  Returns (cat [n, 2H | 3H], [saved activation per layer]) when save=True, else (None, [])."""
    _req(a_rows, 'a_rows'); _req(b_rows, 'b_rows'); _req(out, 'out')
    _req(a_idx, 'a_idx', torch.int32); _req(b_idx, 'b_idx', torch.int32); _req(out_row_idx, 'out_row_idx', torch.int32)
    H = a_rows.shape[-1]
    n = int(a_idx.numel())
    nseg = 2 if kind == 'cat2' else 3
    a = TileMlpArgs()
    a.vec_pack, a.vec_cnt = VEC_PACKS[kind], n
    a.pk_a, a.pk_b, a.pk_a_idx, a.pk_b_idx = a_rows.data_ptr(), b_rows.data_ptr(), a_idx.data_ptr(), b_idx.data_ptr()
    keep = []
    for l, (w, b, act) in enumerate(layers):
        _req(w, 'weight')
        if l == 0:
            assert w.shape == (H, nseg * H), w.shape
            planes = torch.empty(nseg * 2 * H * H, dtype=torch.bfloat16, device=w.device)
            for j in range(nseg):
                check(lib.stair_pack_wfrag_ld(w.data_ptr() + 4 * j * H, nseg * H, planes.data_ptr() + 2 * j * 2 * H * H, H, H, _stream()))
        else:
            planes = pack_wfrag(w)
        keep.append(planes)
        a.W[l], a.bias[l], a.act[l] = planes.data_ptr(), (b.data_ptr() if b is not None else None), ACT[act]
    a.n_layers = len(layers)
    cat, saves = None, []
    if save:
        cat = torch.empty(n, nseg * H, device=out.device)
        a.cat_save = cat.data_ptr()
        for l in range(len(layers)):
            sv = torch.empty(n, H, device=out.device)
            saves.append(sv)
            a.save[l] = sv.data_ptr()
    a.tail = 7
    a.out, a.out_gstride, a.out_row_idx = out.data_ptr(), H, out_row_idx.data_ptr()
    a.cnt, a.T, a.H = (n + 63) // 64, 64, H
    check(lib.stair_tile_mlp_fwd(C.byref(a), _stream()))
    return cat, saves

