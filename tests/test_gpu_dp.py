"""Data-parallel training on ONE card: two processes, each a rank of a gloo group, each running Trainer.step(world=2)
on its round-robin shard of the window -- the path bench.py --gpus N takes with nccl (= RCCL), rehearsed here with
gloo so that a single-GPU box can run it.  The post-step state must equal the single-process step over the whole
window (/root/reference/train_module.py:386-412: ONE accumulation window, one Adam step), including
  * a ragged split (13 questions -> 7 + 6): the loss normalisation is the GLOBAL window, found by an all-reduce;
  * the touched mask riding in the gradient bucket (one collective), union over ranks;
  * intermediate supervision with contrastive gold: the class pools are those of the global window on every rank.
Only sums of fp32 partial gradients are re-ordered between the two runs, hence the tight tolerances."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

from stair_amd import spec, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40,
              object_types=10, dropout=0.25)          # dropout in the config: the Trainer gets dropout=0.0 explicitly
N_Q, WINDOW = 13, 5


def _questions(supervised):
    qs = [synth.make_question(CONFIG, 5, i, form=synth.ALL_FORMS[i % len(synth.ALL_FORMS)]) for i in range(N_Q)]
    for q in qs:
        if supervised:
            sg = synth.make_gold(CONFIG, 3, q, keep=1.0)
            q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v)
                                   for k, v in sg.items()}
    return qs


def _model(dev):
    from stair_amd import losses as L
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(CONFIG, pretrain_modules=set(L.CRITERION_MODULES))
    w = synth.make_weights(CONFIG, 2)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(CONFIG)})
    return m.to(dev)


def _pack(qs, dev, bf16):
    video = torch.stack([torch.as_tensor(q['video_features']) for q in qs]).to(dev)
    video = video.to(torch.bfloat16).contiguous() if bf16 else video.float().contiguous()
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(dev, torch.float32).contiguous()
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
    return ([q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs], video, question,
            [q['question'].shape[0] for q in qs], answers)


def _run(rank, world, supervised, bf16, out, table=False, overlap=None, fail_rank=None):
    from stair_amd import losses as L
    from stair_amd.train import Trainer
    dev = torch.device('cuda', 0)
    qs = _questions(supervised)
    mine = qs[rank::world]                                    # local i <-> global position rank + i * world
    model = _model(dev)
    # table: the contrastive pools travel as a presence matrix summed on the device (losses.ClassTable), built from each
    # rank's OWN shard and merged once -- instead of the per-step all_gather_object of the class lists
    class_table = L.ClassTable.from_questions(mine, world) if table else None
    tr = Trainer(model, world=world, rank=rank, dropout=0.0, contrastive_window=WINDOW, lr=1e-3, class_table=class_table,
                 overlap_allreduce=overlap)
    state = {}
    for it in range(2):                                        # the second step exercises 'ever'-touched bookkeeping
        progs, spans, video, question, q_lens, answers = _pack(mine, dev, bf16)
        if fail_rank is not None and it == 1:                  # the second step's pass "fails" on ONE rank
            before = (tr.flat_p.clone(), tr.exp_avg.clone(), tr.steps.clone(), tr.touched.clone())
            tr.inject_failure = rank == fail_rank
        loss, _ = tr.step(progs, spans, video, question, q_lens, answers, questions=mine if supervised else None)
        torch.cuda.synchronize()
        if fail_rank is not None and it == 1:
            from stair_amd._lib import StairError
            state['unchanged'] = all(torch.equal(a, b) for a, b in zip(before, (tr.flat_p, tr.exp_avg, tr.steps, tr.touched)))
            state['guard'] = int(tr.guard[0])
            try:
                tr.check()
                state['raised'] = False
            except StairError:
                state['raised'] = True
        state['grad%d' % it] = tr.flat_g.cpu().clone()
        state['loss%d' % it] = loss.cpu().clone()
    state.update(params=tr.flat_p.cpu().clone(), touched=tr.touched.cpu().clone(), steps=tr.steps.cpu().clone(),
                 seen=tr.questions_seen)
    torch.save(state, out)


def _worker(rank, world, port, supervised, bf16, out_dir, table=False, overlap=None, fail_rank=None, tag='rank', full_size=False):
    import torch.distributed as dist
    if full_size:                    # the reference's own sizes: the fused / grouped kernels, whose step is bit-reproducible
        globals()['CONFIG'] = dict(spec.DEFAULT_CONFIG, max_video_length=40)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _run(rank, world, supervised, bf16, os.path.join(out_dir, '%s%d.pt' % (tag, rank)), table, overlap, fail_rank)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('supervised,bf16,table', [(False, False, False), (True, False, False), (True, True, False), (True, False, True)])
def test_two_rank_trainer_step_equals_single_process_step(supervised, bf16, table):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        _run(0, 1, supervised, bf16, os.path.join(d, 'solo.pt'))          # the solo run pools by class lists (no table)
        mp.spawn(_worker, args=(2, _free_port(), supervised, bf16, d, table), nprocs=2, join=True)
        solo = torch.load(os.path.join(d, 'solo.pt'))
        ranks = [torch.load(os.path.join(d, 'rank%d.pt' % r)) for r in range(2)]
    assert solo['seen'] == ranks[0]['seen'] == ranks[1]['seen'] == 2 * N_Q
    for key in ('params', 'touched', 'steps', 'grad0', 'grad1'):
        assert torch.equal(ranks[0][key], ranks[1][key]), key                   # both ranks hold the same state
    assert torch.equal(solo['touched'], ranks[0]['touched']) and torch.equal(solo['steps'], ranks[0]['steps'])
    for it in range(2):
        g, r = solo['grad%d' % it], ranks[0]['grad%d' % it]
        if it == 0:
            assert float((g - r).abs().max()) <= 2e-5 * float(g.abs().max()), it     # same terms, summed in another order
        else:
            # the second step starts from weights that differ in the last bit (Adam on gradients summed in another order): a ReLU unit
            # whose pre-activation lies within that rounding of zero takes the other branch in one of the two runs, which moves one
            # row of a weight gradient -- and what lies upstream of it -- by a finite amount ("ReLU kink flips").
            # Seen once: Filter's second layer and, upstream of it, the video encoder, 0.5 % of the largest gradient; every other
            # tensor to 1e-6.  So: the whole bucket to 1 % in L2, every entry to 1 % of the largest gradient
            d = (g - r).abs()
            assert float(d.norm()) <= 1e-2 * float(g.norm()), it
            assert float(d.max()) <= 1e-2 * float(g.abs().max()), it
        # the decoder losses of the shards are those of the solo run, question by question
        both = torch.empty(N_Q)
        both[0::2], both[1::2] = ranks[0]['loss%d' % it], ranks[1]['loss%d' % it]
        assert torch.allclose(both, solo['loss%d' % it], rtol=1e-5, atol=1e-6), it
    dp = (solo['params'] - ranks[0]['params']).abs()
    # Adam divides by sqrt(v): where a gradient is ~0 the step direction is ill-conditioned; lr = 1e-3, two steps
    # (and where a ReLU unit flipped in the second step, see above, a row of entries moves by up to the step size)
    assert float((dp > 2e-5).float().mean()) < 5e-3 and float(dp.max()) <= 2.1e-3


def test_two_piece_exchange_equals_one_piece_bit_for_bit():
    """Trainer(overlap_allreduce=True): [module + decoder gradients] reduced on a side stream from the backward pass's
    "module gradients are final" event on, [encoder gradients | mask | status] after the pass -- against the whole bucket in one
    collective after the pass, two supervised steps each.  The exchange itself is bit-identical either way (a + b; pinned on CPU
    tensors in tests/test_sharding_gloo.py); two RUNS of the backward pass are not (fp32 atomics in the dX fan-in), so the two
    jobs are compared as two runs of one job are: masks, step counts and losses exactly, gradients to 2e-5 of their largest
    entry -- a piece reduced before it was final, or a zero_grad overtaking the side stream, is wrong by whole terms."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), True, True, d, True, False, None, 'one'), nprocs=2, join=True)
        mp.spawn(_worker, args=(2, _free_port(), True, True, d, True, True, None, 'two'), nprocs=2, join=True)
        one = [torch.load(os.path.join(d, 'one%d.pt' % r)) for r in range(2)]
        two = [torch.load(os.path.join(d, 'two%d.pt' % r)) for r in range(2)]
    for r in range(2):
        for key in ('touched', 'steps'):
            assert torch.equal(one[r][key], two[r][key]), (r, key)
        for key in ('loss0', 'loss1'):
            assert torch.allclose(one[r][key], two[r][key], rtol=1e-5, atol=1e-6), (r, key)
        for key in ('grad0', 'grad1'):
            a, b = one[r][key], two[r][key]
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (r, key)
        dp = (one[r]['params'] - two[r]['params']).abs()
        assert float((dp > 2e-5).float().mean()) < 2e-3 and float(dp.max()) <= 2.1e-3
    for key in ('params', 'grad0', 'grad1', 'touched', 'steps'):
        assert torch.equal(two[0][key], two[1][key]), key       # both ranks of the overlapped job hold the same state, bit for bit


def test_two_piece_exchange_is_bit_identical_at_full_size():
    """The same comparison where the step itself is reproducible (hidden size 512: fused tile operators, grouped vector-level
    launches, deterministic fan-in): overlapped two-piece exchange against one collective after the pass -- gradients, weights,
    step counts bit for bit on both ranks.  The module-level shadows must have reached the fp32 gradients before the early piece
    is reduced (stair_plan_backward flushes them ahead of the 'module gradients final' event)."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), False, True, d, False, False, None, 'one', True), nprocs=2, join=True)
        mp.spawn(_worker, args=(2, _free_port(), False, True, d, False, True, None, 'two', True), nprocs=2, join=True)
        one = [torch.load(os.path.join(d, 'one%d.pt' % r)) for r in range(2)]
        two = [torch.load(os.path.join(d, 'two%d.pt' % r)) for r in range(2)]
    for r in range(2):
        for key in ('params', 'touched', 'steps', 'grad0', 'grad1', 'loss0', 'loss1'):
            assert torch.equal(one[r][key], two[r][key]), (r, key)
    assert float(one[0]['grad0'].abs().max()) > 0


@pytest.mark.parametrize('overlap', [False, True])
def test_a_pass_that_fails_on_one_rank_is_refused_by_every_rank(overlap):
    """The status word of a step travels in the gradient bucket: when ONE rank's pass reports a failure (here injected after
    the backward pass of rank 1's second step) the reduced word is set on BOTH ranks, both Adam kernels refuse the update --
    weights, moments, per-tensor step counts and the 'ever touched' mask stay those of the first step -- and both ranks'
    Trainer.check() raise for that step (round 3 read the local word only: the healthy rank applied the NaN sum and hung in
    the next collective)."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), False, True, d, False, overlap, 1, 'f'), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, 'f%d.pt' % r)) for r in range(2)]
    for r in ranks:
        assert r['guard'] == 1 and r['unchanged'] and r['raised']
    assert torch.equal(ranks[0]['params'], ranks[1]['params'])


def test_native_allreduce_entry_point_single_rank():
    """stair_comm_* / stair_allreduce_grads (RCCL resolved at run time from libstair_hip.so) on a one-rank communicator:
    the sum over one rank is the identity, on the caller's stream; a two-rank RCCL communicator needs two GPUs and is
    left to the driver's multi-GPU run (Trainer(native_allreduce=True))."""
    import torch
    from stair_amd.comm import NativeComm
    from stair_amd._lib import StairError
    comm = NativeComm(0, 1)
    x = torch.randn(1 << 20, device='cuda:0')
    ref = x.clone()
    comm.allreduce_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    with pytest.raises(TypeError):
        comm.allreduce_(x.double())
    comm.close()
    with pytest.raises(StairError):
        NativeComm(1, 1)            # rank outside the world
