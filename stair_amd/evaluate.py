"""Accuracy evaluation driver: the batched, multi-GPU counterpart of the reference's
``evaluate_by_module`` accuracy loop (/root/reference/train_module.py:219-270, the working one) and
of ``evaluate.py:28-62`` (whose argmax over dim=1 of a 1-D tensor crashes as shipped; the intended
rule is the one implemented here).

Questions are independent, so rank r of a world of W processes takes questions r, r+W, r+2W, ...
and runs them in batches on its own GPU with NO data-path collective; predictions are gathered on the
host once at the end (torch.distributed.all_gather_object -- RCCL is not needed for it, and the CPU
test uses gloo).
"""
from __future__ import annotations

import ctypes as C
import json

import torch

from ._lib import check, lib


def shard_indices(n, rank, world):
    """Indices of the questions rank `rank` processes (round-robin)."""
    return list(range(rank, n, world))


def clip_key(d):
    """What tells clips apart in a question list: `video_id` when present, else the feature buffer itself
    (AGQADataset gives every question of a video the same tensor, dataset.py:183)."""
    if 'video_id' in d:
        return ('id', d['video_id'])
    from .module_net import _clip_key
    return _clip_key(d['video_features'])


def shard_by_clip(questions, rank, world):
    """Clip-aligned sharding for the per-video encoder cache: all questions of a clip go
    to ONE rank, so no clip is encoded on two GPUs.  Clips are dealt largest-first to the least-loaded rank
    (deterministic: ties by first appearance), which keeps the ranks within one clip's questions of each other.
    Returns this rank's question indices, clip by clip."""
    groups, order = {}, []
    for i, d in enumerate(questions):
        k = clip_key(d)
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append(i)
    load = [0] * world
    mine = []
    for k in sorted(order, key=lambda k: (-len(groups[k]), groups[k][0])):
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += len(groups[k])
        if r == rank:
            mine.append(groups[k])
    return [i for g in sorted(mine, key=lambda g: g[0]) for i in g]


def group_by_frames(batch, ragged=True):
    """Launch groups of a question list.  Clips of different frame counts run in ONE batch (stair_plan_build_ragged), so
    with ragged=True (Conv1d-Temporal configurations) there is a single group; ragged=False buckets by frame count, which
    the Linear(T,T) Temporal configurations still need (they fix T = max_video_length like the reference)."""
    if ragged:
        return {0: list(range(len(batch)))} if len(batch) else {}
    groups = {}
    for i, d in enumerate(batch):
        groups.setdefault(int(d['video_features'].shape[0]), []).append(i)
    return groups


def predict(model, questions, batch_size=1024):
    """Top-1 answer ids (python ints) for a list of question dicts, in order."""
    preds = [None] * len(questions)
    for T, idxs in sorted(group_by_frames(questions, ragged=model.config['max_video_length'] > 32).items()):
        # questions of one clip next to each other, so a launch batch encodes each of its clips once
        first = {}
        for i in idxs:
            first.setdefault(clip_key(questions[i]), i)
        idxs = sorted(idxs, key=lambda i: (first[clip_key(questions[i])], i))
        for s in range(0, len(idxs), batch_size):
            chunk = idxs[s:s + batch_size]
            res = model.forward_batch([questions[i] for i in chunk])
            for i, p in zip(chunk, res.pred.cpu().tolist()):
                preds[i] = int(p)
    return preds


def accuracy(preds, golds, unk_token_id):
    """train_module.py:252-253: a prediction counts only if it equals the gold answer AND the gold
    answer is not <UNK>."""
    acc = [int(p == g and g != unk_token_id) for p, g in zip(preds, golds)]
    return sum(acc) / max(1, len(acc))


def evaluate(model, questions, unk_token_id, rank=0, world=1, batch_size=1024, predict_fn=None, preds_file=None,
             id2word=None, shard='round_robin'):
    """Sharded accuracy evaluation.  Every rank passes the SAME full question list; returns
    (accuracy, preds) on every rank.  `predict_fn(questions) -> list[int]` defaults to the HIP path
    (tests substitute a CPU function to exercise the sharding logic without a GPU).
    shard: 'round_robin'  or 'clip' (whole clips per rank, see shard_by_clip)."""
    if shard not in ('round_robin', 'clip'):
        raise ValueError("shard must be 'round_robin' or 'clip'")
    mine = shard_indices(len(questions), rank, world) if shard == 'round_robin' else shard_by_clip(questions, rank, world)
    fn = predict_fn or (lambda qs: predict(model, qs, batch_size))
    local = fn([questions[i] for i in mine])
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, local))
    else:
        gathered = [(mine, local)]
    preds = [None] * len(questions)
    for idxs, vals in gathered:
        for i, p in zip(idxs, vals):
            preds[i] = int(p)
    golds = [int(q['answer']) for q in questions]
    acc = accuracy(preds, golds, unk_token_id)
    if preds_file is not None and rank == 0:       # train_module.py:267-268 layout
        w = (lambda i: id2word[str(i)] if id2word and str(i) in id2word else i)
        json.dump({'preds': [w(p) for p in preds], 'golds': [w(g) for g in golds],
                   'qa_ids': [q.get('qa_id') for q in questions]}, open(preds_file, 'w'))
    return acc, preds


def filter_text_results(model, questions, filter_vocab, phrase_embeddings, batch_size=1024, top=10):
    """Batched counterpart of /root/reference/evaluate.py:65-117 (get_filter_text_results): for every `Filter`
    node of every question, the `top` phrases of `filter_vocab` whose text-encoder representation is most similar
    (cosine) to the node's output.  phrase_embeddings[i] = word embeddings [L_i, E] of filter_vocab[i] (the reference
    gets them from the dataset's GloVe table, dataset.py:248-255).  Returns
        {qa_id: {program_idx: (level, keyword_text, [top phrases])}}
    with level as stat_module_levels and keyword_text the Filter's second operand, as the reference records them.
    One stair_cosine_topk launch ranks all Filter nodes of a batch; questions whose program holds no Filter map to {}."""
    from . import frontend, ops
    import numpy as np
    if len(filter_vocab) != len(phrase_embeddings):
        raise ValueError('one embedding per vocabulary phrase')
    reps = model.encode_phrases(phrase_embeddings)                      # [C, H]
    H = model.config['hidden_size']
    out = {}
    for T, idxs in sorted(group_by_frames(questions, ragged=model.config['max_video_length'] > 32).items()):
        for s in range(0, len(idxs), batch_size):
            chunk = [questions[i] for i in idxs[s:s + batch_size]]
            res = model.forward_batch(chunk)
            where, slots = [], []
            for qi, d in enumerate(chunk):
                prog = d['nmn_program_list']
                out[d.get('qa_id', idxs[s + qi])] = {}
                if 'Filter' not in prog:
                    continue
                levels = frontend.module_levels(prog)
                children, _ = frontend.children_and_parents(prog)
                pidx = d.get('nmn_program_idx') or list(range(len(prog)))
                for i, tok in enumerate(prog):
                    if tok == 'Filter':
                        where.append((d.get('qa_id', idxs[s + qi]), pidx[i], levels[i], prog[children[i][1]].replace('_', ' ')))
                        slots.append(res.node_info(qi, i)[1])
            if not slots:
                continue
            vec = res._arena(res.info.vec_off, res.info.n_vec, H)
            q_idx = torch.tensor(np.asarray(slots, dtype=np.int32), device=vec.device)
            best, _ = ops.cosine_topk(vec, reps, min(top, len(filter_vocab)), q_idx=q_idx)
            for (qa, pi, lvl, kw), row in zip(where, best.cpu().tolist()):
                out[qa][pi] = (lvl, kw, [filter_vocab[c] for c in row])
    return out


def evaluate_by_module(model, questions, unk_token_id, batch_size=1024, module_loss_weight=1.0, preds_file=None, id2word=None):
    """Batched counterpart of /root/reference/train_module.py:219-270 (`evaluate_by_module`, the validation pass of the
    training loop): returns (accuracy, {module: mean validation loss}).  Supervised nodes are those of
    `model.pretrain_modules` whose program_idx has a gold value in the question's `sg_res_by_step`; contrastive modules
    are scored with the 'cont-valid' cosine metric, the decoder with cross entropy against `answer`; a module without
    any scored node reports +inf, as the reference does.  module_loss_weight == 0 skips the module scores (:232)."""
    from . import losses as L
    modules = set(model.pretrain_modules) | {'decoder'}
    losses = {m: [] for m in L.CRITERION_MODULES}
    preds = [None] * len(questions)
    for T, idxs in sorted(group_by_frames(questions, ragged=model.config['max_video_length'] > 32).items()):
        for s in range(0, len(idxs), batch_size):
            chunk_idx = idxs[s:s + batch_size]
            chunk = [questions[i] for i in chunk_idx]
            res = model.forward_batch(chunk)
            if module_loss_weight != 0:
                for m, vals in L.evaluate_module_losses(model, res, chunk, modules & L.CRITERION_MODULES).items():
                    losses[m].extend(vals)
            answers = torch.tensor([int(q['answer']) for q in chunk], dtype=torch.int32).to(res.logits.device)
            ce = torch.empty(len(chunk), dtype=torch.float32, device=res.logits.device)
            check(lib.stair_loss_decoder_ce(C.c_void_p(res.logits.data_ptr()), C.c_void_p(answers.data_ptr()), C.c_void_p(ce.data_ptr()),
                                            len(chunk), res.logits.shape[1], C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            losses['decoder'].extend(ce.cpu().tolist())
            for i, p in zip(chunk_idx, res.pred.cpu().tolist()):
                preds[i] = int(p)
    golds = [int(q['answer']) for q in questions]
    valid = {m: (sum(v) / len(v) if v else float('inf')) for m, v in losses.items()}
    if preds_file is not None:
        w = (lambda i: id2word[i] if id2word and i in id2word else i)
        json.dump({'preds': [w(p) for p in preds], 'golds': [w(g) for g in golds],
                   'qa_ids': [q.get('qa_id') for q in questions]}, open(preds_file, 'w'))
    return accuracy(preds, golds, unk_token_id), valid
