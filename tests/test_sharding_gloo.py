"""world_size-2 gloo test of the multi-GPU (here: multi-process CPU) inference path: questions shard
round-robin across ranks with no data-path collective, predictions are gathered once, and every rank
ends with the same full prediction list and accuracy.  The per-rank predictor is the oracle (CPU) so
the test exercises the sharding / gather logic of stair_amd/evaluate.py without a GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import nmn_oracle as O
    from stair_amd import evaluate as E, spec, synth
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    w = O.to_torch(synth.make_weights(config, 0))
    qs = synth.make_questions(config, 3, 13, forms=synth.ALL_FORMS)       # odd count: ragged shards

    def predict_fn(sub):
        return [int(torch.argmax(O.forward(w, config, q, return_res_by_step=False)['logits'])) for q in sub]

    acc, preds = E.evaluate(None, qs, unk_token_id=15, rank=rank, world=world, predict_fn=predict_fn)
    torch.save({'acc': acc, 'preds': preds, 'mine': E.shard_indices(len(qs), rank, world)}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_evaluation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    assert r0['preds'] == r1['preds'] and r0['acc'] == r1['acc']
    assert sorted(r0['mine'] + r1['mine']) == list(range(13)) and not set(r0['mine']) & set(r1['mine'])
    # single-process reference
    sys.path.insert(0, ROOT)
    from oracle import nmn_oracle as O
    from stair_amd import evaluate as E, spec, synth
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    w = O.to_torch(synth.make_weights(config, 0))
    qs = synth.make_questions(config, 3, 13, forms=synth.ALL_FORMS)
    solo = [int(torch.argmax(O.forward(w, config, q, return_res_by_step=False)['logits'])) for q in qs]
    assert r0['preds'] == solo
    assert r0['acc'] == E.accuracy(solo, [q['answer'] for q in qs], 15)


def test_accuracy_rule_ignores_unk_gold():
    from stair_amd import evaluate as E
    assert E.accuracy([1, 2, 5], [1, 3, 5], unk_token_id=5) == pytest.approx(1 / 3)    # train_module.py:252-253
    assert E.shard_indices(7, 1, 3) == [1, 4]


def test_clip_aligned_sharding_keeps_clips_whole_and_ranks_balanced():
    """shard_by_clip: disjoint cover, no clip on two ranks, loads within the largest clip."""
    from stair_amd import evaluate as E
    import numpy as np
    rs = np.random.RandomState(0)
    clips = [np.zeros((4, 8), dtype=np.float32) for _ in range(11)]
    owner = [int(c) for c in rs.choice(11, size=97, p=np.arange(1, 12) / 66.0)]
    qs = [{'video_features': clips[c], 'qid': i} for i, c in enumerate(owner)]
    for world in (1, 2, 3, 8):
        shards = [E.shard_by_clip(qs, r, world) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(97))
        clip_rank = {}
        for r, s in enumerate(shards):
            for i in s:
                assert clip_rank.setdefault(owner[i], r) == r
        biggest = max(owner.count(c) for c in set(owner))
        assert max(map(len, shards)) - min(map(len, shards)) <= biggest
    # explicit ids take precedence over buffer identity
    qs2 = [{'video_features': np.zeros((4, 8), dtype=np.float32), 'video_id': 'v%d' % (i % 3)} for i in range(9)]
    s0, s1 = E.shard_by_clip(qs2, 0, 2), E.shard_by_clip(qs2, 1, 2)
    assert {i % 3 for i in s0}.isdisjoint({i % 3 for i in s1}) and len(s0) + len(s1) == 9


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import nmn_oracle as O
    from stair_amd import spec, synth
    from stair_amd.train import reduce_gradients
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    names = [n for n, _ in spec.weight_table(config)]
    wnp = synth.make_weights(config, 0)
    w = {k: torch.from_numpy(wnp[k].copy()).requires_grad_(True) for k in names}
    forms = ['P1', 'P4', 'P3', 'P6']                   # rank 0 gets P1, P3; rank 1 gets P4, P6
    qs = [synth.make_question(config, 9, i, form=f) for i, f in enumerate(forms)]
    G = len(qs)
    loss = 0.0
    for i in range(rank, G, world):
        logits = O.forward(w, config, qs[i], return_res_by_step=False)['logits']
        loss = loss + torch.nn.functional.cross_entropy(logits.unsqueeze(0), torch.tensor([qs[i]['answer']])) / G
    loss.backward()
    flat = torch.cat([(w[n].grad if w[n].grad is not None else torch.zeros_like(w[n])).reshape(-1) for n in names])
    touched = torch.tensor([int(w[n].grad is not None) for n in names], dtype=torch.int32)
    reduce_gradients(flat, touched, world)
    torch.save({'flat': flat, 'touched': touched}, os.path.join(out_dir, 'g%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_reduction_equals_single_process(tmp_path):
    """Sharding a window over 2 ranks + one flat all-reduce == the single-process gradient of the whole
    window (loss normalised by the GLOBAL window size), and the touched mask is the union over ranks."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, 'g0.pt'))
    g1 = torch.load(os.path.join(tmp_path, 'g1.pt'))
    assert torch.equal(g0['flat'], g1['flat']) and torch.equal(g0['touched'], g1['touched'])
    sys.path.insert(0, ROOT)
    from oracle import nmn_oracle as O
    from stair_amd import spec, synth
    config = dict(spec.DEFAULT_CONFIG, hidden_size=64, video_size=128, answer_vocab_length=16, max_video_length=40, object_types=10)
    names = [n for n, _ in spec.weight_table(config)]
    wnp = synth.make_weights(config, 0)
    w = {k: torch.from_numpy(wnp[k].copy()).requires_grad_(True) for k in names}
    qs = [synth.make_question(config, 9, i, form=f) for i, f in enumerate(['P1', 'P4', 'P3', 'P6'])]
    loss = 0.0
    for q in qs:
        logits = O.forward(w, config, q, return_res_by_step=False)['logits']
        loss = loss + torch.nn.functional.cross_entropy(logits.unsqueeze(0), torch.tensor([q['answer']])) / len(qs)
    loss.backward()
    flat = torch.cat([(w[n].grad if w[n].grad is not None else torch.zeros_like(w[n])).reshape(-1) for n in names])
    touched = torch.tensor([int(w[n].grad is not None) for n in names], dtype=torch.int32)
    assert torch.allclose(g0['flat'], flat, rtol=1e-5, atol=1e-7)
    assert torch.equal(g0['touched'], touched)
    # Equals is only used by P4 (rank 1's shard): rank 0 alone would have left it untouched
    assert int(touched[names.index('submodules.Equals.param.0.weight')]) == 1


def _worker_exchange(rank, world, port, out_dir):
    """The host side of the one exchange step of a DP training step, on CPU tensors: the bucket all-reduce with the touched
    mask in its tail, and the global contrastive class pools."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # losses.contrastive_windows / train.reduce_gradients are pure host logic; import them without the HIP library's GPU calls
    from stair_amd import losses as L
    from stair_amd.train import reduce_gradients
    total, ntensor = 512, 5
    bucket = torch.zeros(total + 256)
    flat_g = bucket[:total]
    flat_g.copy_(torch.arange(total, dtype=torch.float32) * (rank + 1))
    touched = torch.tensor([1, 0, 0, 1, 0] if rank == 0 else [0, 0, 1, 1, 0], dtype=torch.int32)
    reduce_gradients(flat_g, touched, world, bucket)
    # the same exchange in two pieces (module + decoder gradients first, [encoder gradients | mask | status] second) with the
    # status word of ONE rank set: bit-identical sums, the union mask, and the failure visible on BOTH ranks
    g = torch.Generator().manual_seed(100 + rank)
    vals = torch.randn(total, generator=g) * 1e3
    outs = {}
    for mode, split in (('one', None), ('two', 256)):
        b2 = torch.zeros(total + 256)
        b2[:total].copy_(vals)
        t2 = torch.tensor([1, 0, 0, 1, 0] if rank == 0 else [0, 0, 1, 1, 0], dtype=torch.int32)
        st = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
        reduce_gradients(b2[:total], t2, world, b2, status=st, split=split)
        outs[mode] = (b2[:total].clone(), t2.clone(), int(st[0]))
    ok = torch.tensor([0], dtype=torch.int32)
    b3 = torch.zeros(total + 256)
    reduce_gradients(b3[:total], torch.zeros(ntensor, dtype=torch.int32), world, b3, status=ok, split=256)
    # rank r holds the contrastive golds of global positions r, r + 2, ...: classes chosen so that windows overlap
    entries = [(g, 'class_%d' % (g % 5), np.full((1 + g % 3, 4), float(g % 5), dtype=np.float32)) for g in range(rank, 21, world)]
    names, embs, rows, win_range, slot_of = L.contrastive_windows(entries, 8, world)
    torch.save({'one': outs['one'], 'two': outs['two'], 'ok_status': int(ok[0]), 'g': flat_g.clone(), 'touched': touched, 'names': names, 'rows': rows.tolist(), 'win_range': win_range,
                'slot_of': {'%d/%s' % k: v for k, v in slot_of.items()}, 'emb0': [e.tolist() for e in embs]}, os.path.join(out_dir, 'x%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_bucket_and_global_contrastive_windows(tmp_path):
    import numpy as np
    world = 2
    mp.spawn(_worker_exchange, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'x0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'x1.pt'))
    # ONE all-reduce: gradients summed, touched mask = union over ranks
    assert torch.equal(r0['g'], torch.arange(512, dtype=torch.float32) * 3) and torch.equal(r0['g'], r1['g'])
    assert r0['touched'].tolist() == r1['touched'].tolist() == [1, 0, 1, 1, 0]
    # two collectives == one collective, bit for bit; one rank's failure reaches every rank; no failure -> no flag
    for r in (r0, r1):
        assert torch.equal(r['one'][0], r['two'][0]) and torch.equal(r['one'][0], r0['one'][0])
        assert r['one'][1].tolist() == r['two'][1].tolist() == [1, 0, 1, 1, 0]
        assert r['one'][2] == r['two'][2] == 1 and r['ok_status'] == 0
    # every rank built the tables of the whole window, i.e. the single-process tables
    sys.path.insert(0, ROOT)
    from stair_amd import losses as L
    entries = [(g, 'class_%d' % (g % 5), np.full((1 + g % 3, 4), float(g % 5), dtype=np.float32)) for g in range(21)]
    names, embs, rows, win_range, slot_of = L.contrastive_windows(entries, 8, 1)
    for r in (r0, r1):
        assert r['names'] == names and r['rows'] == rows.tolist() and r['win_range'] == win_range
        assert r['slot_of'] == {'%d/%s' % k: v for k, v in slot_of.items()}
    assert sorted(win_range) == [0, 1, 2] and win_range[0][1] == 5 and win_range[2][1] == 5
