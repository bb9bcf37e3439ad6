"""Cooperative LSTM kernels (csrc/lstm_coop.hip: hidden units split over 4 co-resident workgroups, W_hh in registers,
per-step exchange through global slabs) against the one-workgroup kernels of csrc/lstm.hip, which the other test files
pin to the oracle (nn.LSTM of /root/reference/video_nmn/module_net.py:39-47,151-163 and its autograd).  Both compute
the same split-bf16 products with fp32 accumulation, in a different order: agreement to rounding, for every group
geometry the launchers choose (one and two sequence tiles per group, several chunks per group, ragged lengths, padded
storage, batches of one)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
Hh = 256


def _weights(I, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(Hh)
    u = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(DEV)
    return [t for _ in range(2) for t in (u(4 * Hh, I), u(4 * Hh, Hh), u(4 * Hh), u(4 * Hh))]


def _case(n, lo, hi, I, seed, padded):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
    if padded:
        off = np.arange(n + 1) * hi
        seq_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    else:
        off = np.concatenate([[0], np.cumsum(lens)])
        seq_len = None
    rows = int(off[-1])
    x = torch.randn(rows, I, generator=g).to(DEV)
    d_out = torch.randn(rows, 2 * Hh, generator=g).to(DEV)
    d_hn = torch.randn(n, 2 * Hh, generator=g).to(DEV)
    return lens, torch.tensor(off, dtype=torch.int32, device=DEV), seq_len, x, d_out, d_hn


def _data_rows(lens, off, seq_len):
    o = off.cpu().numpy()
    return torch.tensor(np.concatenate([np.arange(o[s], o[s] + lens[s]) for s in range(len(lens))]), device=DEV, dtype=torch.long)


@pytest.mark.parametrize('n,lo,hi,padded', [(1, 7, 7, False), (5, 1, 9, False), (31, 2, 12, False), (33, 12, 12, False), (100, 1, 20, True),
                                            (1030, 3, 10, False), (2048, 6, 6, False), (2050, 1, 8, True), (4500, 2, 5, False)])
def test_cooperative_forward_and_bptt_match_the_one_workgroup_kernels(n, lo, hi, padded, monkeypatch):
    from stair_amd import ops
    monkeypatch.setenv('STAIR_LSTM_COOP_BWD_MAX_N', '1000000')      # the launcher's default hands n > 1024 to the one-workgroup kernel
    assert ops.get_matmul_mode() == 'bf16x3'
    I = 64
    ws = _weights(I, n)
    lens, off, seq_len, x, d_out, d_hn = _case(n, lo, hi, I, 100 + n, padded)
    rows_ = _data_rows(lens, off, seq_len)
    got, ref = {}, {}
    for coop, res in ((True, got), (False, ref)):
        out, h_n, gates, cbuf = ops.lstm_bidir(x, off, hi, ws, save=True, coop=coop, seq_len=seq_len)
        res['out'], res['h_n'], res['gates'], res['cbuf'] = out.clone(), h_n.clone(), gates.clone(), cbuf.clone()
    for k in ('out', 'gates', 'cbuf'):
        assert float((got[k][rows_] - ref[k][rows_]).abs().max()) < 2e-5, k
    assert float((got['h_n'] - ref['h_n']).abs().max()) < 2e-5
    if padded:       # rows past a sequence's length read as zero downstream
        mask = torch.ones(x.shape[0], dtype=torch.bool, device=DEV); mask[rows_] = False
        assert float(got['out'][mask].abs().max()) == 0.0
    # BPTT from the SAME saved state (the one-workgroup kernel's), so that only the reverse recurrence differs
    grads = {}
    for coop in (True, False):
        gates = ref['gates'].clone()
        grads[coop] = ops.lstm_bidir_bwd(x, off, hi, ws, ref['out'], gates, ref['cbuf'], d_out, d_hn, coop=coop, seq_len=seq_len)
        grads[coop].append(gates[rows_])             # the gate gradients themselves, written in place
    for a, b_ in zip(grads[True], grads[False]):
        scale = max(1.0, float(b_.abs().max()))
        assert float((a - b_).abs().max()) < 1e-4 * scale, (a.shape, float((a - b_).abs().max()), scale)


def test_cooperative_kernels_are_deterministic():
    from stair_amd import ops
    I, n = 64, 700
    ws = _weights(I, 3)
    lens, off, seq_len, x, d_out, d_hn = _case(n, 1, 16, I, 9, False)
    runs = []
    for _ in range(2):
        out, h_n, gates, cbuf = ops.lstm_bidir(x, off, 16, ws, save=True)
        g = ops.lstm_bidir_bwd(x, off, 16, ws, out, gates, cbuf, d_out, d_hn)
        runs.append([out, h_n, gates])      # gates now holds the gate gradients: fixed-order sums of the exchanged partials
    for a, b_ in zip(*runs):
        assert torch.equal(a, b_)


@pytest.mark.parametrize('cap', [0, 31, 64, 100])
def test_capped_device_falls_back_and_matches(cap):
    """stair_lstm_coop_limit: with fewer co-resident workgroups than a geometry needs, the launchers take smaller
    geometries (cap 64, 100: one or two groups per direction, several chunks per group) or the one-workgroup kernels
    (cap < 32) -- the results must not change beyond rounding, and nothing may be left waiting."""
    from stair_amd import ops
    from stair_amd._lib import lib
    I, n = 64, 300
    ws = _weights(I, 5)
    lens, off, seq_len, x, d_out, d_hn = _case(n, 1, 16, I, 21, False)
    rows_ = _data_rows(lens, off, seq_len)
    ref = ops.lstm_bidir(x, off, 16, ws, save=True, coop=False)
    ref = [t.clone() for t in ref]
    gref = ops.lstm_bidir_bwd(x, off, 16, ws, ref[0], ref[2].clone(), ref[3], d_out, d_hn, coop=False)
    try:
        lib.stair_lstm_coop_limit(cap)
        with ops.kernel_accounting() as acct:
            got = ops.lstm_bidir(x, off, 16, ws, save=True, coop=True)
            g = ops.lstm_bidir_bwd(x, off, 16, ws, ref[0], ref[2].clone(), ref[3], d_out, d_hn, coop=True)
        torch.cuda.synchronize()
    finally:
        lib.stair_lstm_coop_limit(-1)
    if cap < 32:
        assert 'lstm_rec_x3' in acct.table and 'lstm_bwd_x3' in acct.table and 'lstm_rec_coop' not in acct.table, sorted(acct.table)
    else:
        assert 'lstm_rec_coop' in acct.table and 'lstm_bwd_coop' in acct.table, sorted(acct.table)
    for a, b_ in zip(got, ref):
        sel = rows_ if a.shape[0] == x.shape[0] else slice(None)
        assert float((a[sel] - b_[sel]).abs().max()) < 2e-5
    for a, b_ in zip(g, gref):
        assert float((a - b_).abs().max()) < 1e-4 * max(1.0, float(b_.abs().max()))


def test_timeout_word_blocks_the_optimizer_and_raises():
    """The error path of a cooperative hand-off that timed out, with the flag INJECTED (no real timeout is provoked): the
    plan's status word is set between forward and backward, as the kernel would set it -> stair_adam_step's guard leaves
    parameters and moments untouched, Trainer.check() / the next step raise StairError, BatchResult.check() raises too."""
    from stair_amd import spec, synth
    from stair_amd._lib import StairError
    from stair_amd.module_net import BatchResult, VideoNMN
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    m = VideoNMN(config)
    w = synth.make_weights(config, 0)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    m = m.to(DEV)
    qs = [synth.make_question(config, 2, i, form=synth.ALL_FORMS[i % 12]) for i in range(12)]
    video = torch.stack([torch.as_tensor(q['video_features']) for q in qs]).to(DEV)
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    args = ([q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs], video, question,
            [q['question'].shape[0] for q in qs], torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV))
    tr = Trainer(m, dropout=0.0)
    tr.step(*args)                                   # a good step: moments are non-zero afterwards
    tr.check()
    before = (tr.flat_p.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone())
    orig = BatchResult.backward

    def failing_backward(self, *a, **k):
        self.status_word().fill_(1)                  # what lstm_rec_coop_kernel does when a spin runs out
        return orig(self, *a, **k)
    BatchResult.backward = failing_backward
    try:
        _, res = tr.step(*args)
    finally:
        BatchResult.backward = orig
    torch.cuda.synchronize()
    for a, b_ in zip(before, (tr.flat_p, tr.exp_avg, tr.exp_avg_sq)):
        assert torch.equal(a, b_)                    # the guard refused the update on the device
    with pytest.raises(StairError, match='timed out'):
        res.check()
    with pytest.raises(StairError, match='skipped'):
        tr.check()
    tr.step(*args)                                   # the next plan starts with a clean word
    tr.check()
    assert not torch.equal(before[0], tr.flat_p)
