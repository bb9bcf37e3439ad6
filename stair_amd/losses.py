"""Host side of the per-module intermediate-supervision losses (BASELINE.json configs[4]): which program
nodes are supervised, with what gold value, in which contrastive window -- the bookkeeping of
/root/reference/train_module.py:351-406 -- and the launches of the stair_loss_* kernels that inject the
gradients into a training plan's gradient arenas.

Gold values per question live in ``q['sg_res_by_step'] = {program_idx: gold}`` exactly as
``AGQADataset.__getitem__`` produces them (/root/reference/video_nmn/dataset.py:200-221):
    Localize                -> tuple of (start, end) frame intervals, one per keyword
    Temporal / ExistsFrame  -> one (start, end)
    Exists / Xor / Equals   -> bool
    Filter / ToAction / Superlative -> list of (class_name, GloVe embedding [L,300])
    FilterFrame             -> {entity: (start, end)} (left out of training by default, args.py:62
                               modules_no_intermediate_train, but scored in validation); needs model.object_index
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import ops, spec
from ._lib import check, lib

CONTRASTIVE = ('Filter', 'Superlative', 'ToAction')
CRITERION_MODULES = frozenset({'Exists', 'Xor', 'Equals', 'Filter', 'ToAction', 'FilterFrame', 'ExistsFrame',
                               'Superlative', 'Localize', 'Temporal', 'decoder'})     # train_module.py:36-48


def object_index(word2id):
    """{word: column of the FilterFrame head} from the reference's IDX-style {word: id} table: several words may share
    an id and the column is the rank of the id (train_module.py:49-54)."""
    rank = {i: n for n, i in enumerate(sorted(set(word2id.values())))}
    return {w: rank[i] for w, i in word2id.items()}


def _span_mask(start, end, T):
    """train_module.py:67-81 span_to_attention, float32 like the reference's tensor."""
    import math
    g = np.zeros(T, dtype=np.float32)
    start = min(T - 0.002, max(0.001, start))
    end = min(T - 0.001, end)
    si, ei = math.ceil(start), math.floor(end)
    if si < ei:
        g[si:ei] += 1
    if si <= ei:
        g[si - 1] += np.float32(si - start)
        g[ei] += np.float32(end - ei)
    else:
        g[ei] += np.float32(end - start)
    return g


def filterframe_target(gold, T, O, word2index):
    """train_module.py:147-154: one interval mask per gold entity in its object column, rows normalised to sum 1."""
    g = np.zeros((T, O), dtype=np.float32)
    for name, interval in gold.items():
        g[:, word2index[name]] = _span_mask(float(interval[0]), float(interval[1]), T)
    with np.errstate(divide='ignore', invalid='ignore'):
        g = g / g.sum(axis=1, keepdims=True)
    g[~np.isfinite(g)] = 0.0
    return g


def _filterframe_launch(model, res, items, scale, grads):
    """items: [(map slot, gold dict)].  Returns the per-item losses (device tensor)."""
    dev = res.logits.device
    H, T, O = model.config['hidden_size'], res.info.T, model.config['object_types']
    index = getattr(model, 'object_index', None)
    if index is None:
        raise RuntimeError('FilterFrame loss needs model.object_index = losses.object_index(json.load(open(word2id_filename)))')
    head = model.submodules['FilterFrame'].pretrain_head
    if grads and head.weight.grad is None:
        raise RuntimeError('pretrain head of FilterFrame has no .grad buffer (use stair_amd.train.Trainer)')
    gold = torch.from_numpy(np.stack([filterframe_target(g, T, O, index) for _, g in items])).to(dev)
    slot = torch.tensor([a[0] for a in items], dtype=torch.int32, device=dev)
    out = torch.empty(len(items), device=dev)
    inf = res.info
    mp = res._ws[inf.map_off: inf.map_off + inf.n_map * T * H]
    gmap = res.grad_arena('map') if grads else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(lib.stair_loss_filterframe(p(mp), p(gmap), p(slot), p(gold), p(head.weight), p(head.bias),
                                     p(head.weight.grad) if grads else None, p(head.bias.grad) if grads else None,
                                     len(items), T, H, O, C.c_float(scale), p(out),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def supervised_nodes(question, pretrain_modules):
    """{program_idx: token position} as VideoNMN.forward records res_by_step (module_net.py:107-113): module
    tokens with a program_idx, in pretrain_modules, never the root (i == 0); the scan runs from the last token
    to the first, so for duplicated indices (Compare programs) the EARLIEST position wins."""
    prog, idx = question['nmn_program_list'], question['nmn_program_idx']
    out = {}
    for i in range(len(prog) - 1, 0, -1):
        if prog[i] in spec.ARITY and idx[i] is not None and prog[i] in pretrain_modules:
            out[idx[i]] = i
    return out


class GoldPack:
    """The supervised nodes of ONE question in array form -- what a data-loader worker prepares once per question (the
    reference does the equivalent in AGQADataset.__getitem__, dataset.py:200-221, inside its DataLoader workers), so
    that the training loop only concatenates arrays.  Token positions are relative to the question's program.
      att_pos / att_kind (0 Localize: K rows at the node's slot, 1 Temporal: the related_attn row, 2 ExistsFrame) /
      att_iv [n, 2, 2] float64 gold (start, end) intervals (row 1 only for Localize with two keywords) / att_mod
      head[module] = (pos, label)                      Exists, Xor (2-way CE), Equals (MSE)
      cont = [(pos, module, [(class_name, emb)])]     Filter, ToAction, Superlative
      ff = [(pos, {entity: (start, end)})]             FilterFrame"""
    __slots__ = ('att_pos', 'att_kind', 'att_iv', 'att_mod', 'head', 'cont', 'ff')

    def __init__(self, question, pretrain_modules, no_intermediate):
        sg = question.get('sg_res_by_step') or {}
        prog = question['nmn_program_list']
        ap, ak, aiv, am = [], [], [], []
        self.head, self.cont, self.ff = {}, [], []
        if sg:
            for step, i in supervised_nodes(question, pretrain_modules).items():
                module = prog[i]
                if step not in sg or module in no_intermediate or module == 'decoder' or sg[step] is None:
                    continue
                gold = sg[step]
                if module == 'Localize':
                    iv = [tuple(map(float, g)) for g in gold][:2]
                    ap.append(i); ak.append(0); aiv.append(iv + [(0.0, 0.0)] * (2 - len(iv))); am.append(module)
                elif module in ('Temporal', 'ExistsFrame'):
                    ap.append(i); ak.append(1 if module == 'Temporal' else 2)
                    aiv.append([tuple(map(float, gold)), (0.0, 0.0)]); am.append(module)
                elif module in ('Exists', 'Xor', 'Equals'):
                    h = self.head.setdefault(module, ([], []))
                    h[0].append(i); h[1].append(int(bool(gold)))
                elif module in CONTRASTIVE:
                    self.cont.append((i, module, list(gold)))
                elif module == 'FilterFrame':
                    self.ff.append((i, gold))
                else:
                    raise NotImplementedError('intermediate loss for %s' % module)
        self.att_pos = np.asarray(ap, dtype=np.int64)
        self.att_kind = np.asarray(ak, dtype=np.int64)
        self.att_iv = np.asarray(aiv, dtype=np.float64).reshape(-1, 2, 2)
        self.att_mod = am
        self.head = {m: (np.asarray(p, dtype=np.int64), np.asarray(l, dtype=np.int32)) for m, (p, l) in self.head.items()}


def compile_gold(question, pretrain_modules=CRITERION_MODULES, no_intermediate=('FilterFrame',)):
    """GoldPack of a question, cached on the dict (key '_gold_pack') as long as its sg_res_by_step object is the same."""
    key = (id(question.get('sg_res_by_step')), frozenset(pretrain_modules), tuple(no_intermediate))
    hit = question.get('_gold_pack')
    if hit is not None and hit[0] == key:
        return hit[1]
    pack = GoldPack(question, pretrain_modules, no_intermediate)
    question['_gold_pack'] = (key, pack)
    return pack


def contrastive_windows(entries, window, world=1):
    """The class pools of train_module.py:386-402 over the GLOBAL accumulation window.
    entries: [(global question position, class_name, embedding)] of this rank's contrastive golds; with world > 1 every
    rank contributes its entries (all_gather_object: a few KB of host data) and builds the same table, so that a
    data-parallel step pools exactly the classes the single-process window of the reference pools.
    Returns ({window id: {class_name: row}}, [embedding per row], {window id: (first row, count)})."""
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, [(g, n, np.asarray(e, dtype=np.float32)) for g, n, e in entries])
        entries = [t for part in gathered for t in part]
    pools = {}
    for gpos, name, emb in sorted(entries, key=lambda t: (t[0], t[1])):
        pools.setdefault(gpos // window if window else 0, {}).setdefault(name, emb)
    table, embs, rng = {}, [], {}
    for wid in sorted(pools):
        start = len(embs)
        table[wid] = {}
        for name, emb in pools[wid].items():
            table[wid][name] = len(embs)
            embs.append(torch.as_tensor(np.asarray(emb), dtype=torch.float32))
        rng[wid] = (start, len(embs) - start)
    return table, embs, rng


def apply_module_losses(model, res, questions, scale, pretrain_modules=CRITERION_MODULES,
                        no_intermediate=('FilterFrame',), window=32, world=1, rank=0, window_base=0):
    """Evaluate every intermediate loss of the batch and add scale * gradient into res's gradient arenas
    (call res.zero_grad_arenas() first and res.backward(..., keep_arenas=True) afterwards).
    Local question i sits at global position window_base + rank + i * world of the accumulation window (the round-robin
    sharding of Trainer.step); contrastive classes are pooled per `window` GLOBAL questions (contrastive_windows).
    Returns ({loss_kind: per-item losses tensor}, set of extra parameter names that received a gradient)."""
    dev = res.logits.device
    H, T = model.config['hidden_size'], res.info.T
    _, slot_t, aux_t, _, rel_t = res.node_table()
    base = np.asarray(res._prog_off, dtype=np.int64)
    packs = [compile_gold(q, pretrain_modules, no_intermediate) for q in questions]

    losses, touched = {}, set()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vec, gvec = res._arena(res.info.vec_off, res.info.n_vec, H), res.grad_arena('vec')
    att, gatt = res._arena(res.info.att_off, res.info.n_att, T), res.grad_arena('att')
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev, non_blocking=True)

    # ---- attention criteria (Localize / Temporal / ExistsFrame) ----
    sel = [qi for qi, p in enumerate(packs) if p.att_pos.size]
    if sel:
        tok = np.concatenate([packs[qi].att_pos + base[qi] for qi in sel])
        kind = np.concatenate([packs[qi].att_kind for qi in sel])
        iv = np.concatenate([packs[qi].att_iv for qi in sel])                    # [n, 2, 2]
        slot = np.where(kind == 1, rel_t[tok], slot_t[tok])
        K = np.where(kind == 0, aux_t[tok], 1).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(K)])
        rows = np.concatenate([iv[:, 0][:, None, :], iv[:, 1][:, None, :]], axis=1)   # [n, 2, 2]
        keep = np.arange(2)[None, :] < K[:, None]
        ivf = torch.as_tensor(np.ascontiguousarray(rows[keep])).to(dev)             # [sum K, 2] float64
        slot_d, K_d, off_d = i32(slot), i32(K), i32(off)
        out = torch.empty(len(tok), device=dev)
        check(lib.stair_loss_attention(C.c_void_p(att.data_ptr()), C.c_void_p(gatt.data_ptr()), C.c_void_p(slot_d.data_ptr()),
                                       C.c_void_p(K_d.data_ptr()), C.c_void_p(off_d.data_ptr()), C.c_void_p(ivf.data_ptr()),
                                       len(tok), T, C.c_float(scale), C.c_void_p(out.data_ptr()), stream))
        losses['attention'] = out
    # ---- linear heads (Exists / Xor / Equals) ----
    for module in ('Exists', 'Xor', 'Equals'):
        sel = [qi for qi, p in enumerate(packs) if module in p.head]
        if not sel:
            continue
        if not model.config['have_pretrain_head']:
            raise RuntimeError('%s loss needs have_pretrain_head (modules.py)' % module)
        head = model.submodules[module].pretrain_head
        if head.weight.grad is None:
            raise RuntimeError('pretrain head of %s has no .grad buffer (use stair_amd.train.Trainer)' % module)
        tok = np.concatenate([packs[qi].head[module][0] + base[qi] for qi in sel])
        slot_d, lab_d = i32(slot_t[tok]), i32(np.concatenate([packs[qi].head[module][1] for qi in sel]))
        out = torch.empty(len(tok), device=dev)
        check(lib.stair_loss_head(head.weight.shape[0], C.c_void_p(vec.data_ptr()), C.c_void_p(gvec.data_ptr()),
                                  C.c_void_p(slot_d.data_ptr()), C.c_void_p(lab_d.data_ptr()), C.c_void_p(head.weight.data_ptr()),
                                  C.c_void_p(head.bias.data_ptr()), C.c_void_p(head.weight.grad.data_ptr()),
                                  C.c_void_p(head.bias.grad.data_ptr()), len(tok), H, C.c_float(scale),
                                  C.c_void_p(out.data_ptr()), stream))
        losses[module] = out
        touched.update({'submodules.%s.pretrain_head.weight' % module, 'submodules.%s.pretrain_head.bias' % module})
    # ---- FilterFrame ----
    ff_items = [(int(slot_t[base[qi] + pos]), gold) for qi, p in enumerate(packs) for pos, gold in p.ff]
    if ff_items:
        losses['FilterFrame'] = _filterframe_launch(model, res, ff_items, scale, True)
        touched.update({'submodules.FilterFrame.pretrain_head.weight', 'submodules.FilterFrame.pretrain_head.bias'})
    # ---- contrastive (Filter / ToAction / Superlative): class representations of every global window ----
    cont_items, entries = [], []
    for qi, p in enumerate(packs):
        if not p.cont:
            continue
        gpos = window_base + rank + qi * world
        for pos, module, gold in p.cont:
            for class_name, emb in gold:
                cont_items.append((int(slot_t[base[qi] + pos]), gpos // window if window else 0, class_name))
                entries.append((gpos, class_name, emb))
    if cont_items or world > 1:
        table, embs, win_range = contrastive_windows(entries, window, world)
    if cont_items:
        lens = [e.shape[0] for e in embs]
        x = torch.cat(embs).to(dev).contiguous()
        seq_off = i32(np.concatenate([[0], np.cumsum(lens)]))
        _, h_n = ops.lstm_bidir(x, seq_off, max(lens), [w.detach() for w in model._lstm_weights('text_encoder')])
        G = ops.l2normalize(h_n)
        slot_d = i32([c[0] for c in cont_items]); pos_d = i32([table[c[1]][c[2]] for c in cont_items])
        ws_ = i32([win_range[c[1]][0] for c in cont_items]); wc = i32([win_range[c[1]][1] for c in cont_items])
        out = torch.empty(len(cont_items), device=dev)
        check(lib.stair_loss_contrastive(C.c_void_p(vec.data_ptr()), C.c_void_p(gvec.data_ptr()), C.c_void_p(slot_d.data_ptr()),
                                         C.c_void_p(pos_d.data_ptr()), C.c_void_p(ws_.data_ptr()), C.c_void_p(wc.data_ptr()),
                                         C.c_void_p(G.data_ptr()), len(cont_items), H, max(r[1] for r in win_range.values()),
                                         C.c_float(scale), C.c_void_p(out.data_ptr()), stream))
        losses['contrastive'] = out
    return losses, touched


def evaluate_module_losses(model, res, questions, pretrain_modules=CRITERION_MODULES):
    """Validation-time scores of every supervised node of a batch, {module: [loss, ...]} in question order -- the
    inner loop of train_module.py:231-243 (`evaluate_by_module`).  Works on an inference plan: the stair_loss_*
    kernels run with NULL gradient pointers.  Contrastive modules are scored with the reference's 'cont-valid'
    metric (:127-132), the cosine between the node's output and the mean of the question's own gold class
    representations; that one reduction (a few hundred [H] rows) is plain torch on the device."""
    dev = res.logits.device
    H, T = model.config['hidden_size'], res.info.T
    att_items, head_items, cont_items, embs, ff_items = [], {'Exists': [], 'Xor': [], 'Equals': []}, [], [], []
    out = {m: [] for m in sorted(pretrain_modules)}
    _, slot_t, aux_t, _, rel_t = res.node_table()
    for qi, q in enumerate(questions):
        sg = q.get('sg_res_by_step') or {}
        prog = q['nmn_program_list']
        for step, i in supervised_nodes(q, pretrain_modules).items():
            module = prog[i]
            if step not in sg or sg[step] is None or module == 'decoder':
                continue
            gold = sg[step]
            tok_ = int(res._prog_off[qi]) + i
            slot, aux, rel = int(slot_t[tok_]), int(aux_t[tok_]), int(rel_t[tok_])
            if module == 'Localize':
                att_items.append((slot, aux, [tuple(map(float, gold[r])) for r in range(aux)], module))
            elif module == 'Temporal':
                att_items.append((rel, 1, [tuple(map(float, gold))], module))
            elif module == 'ExistsFrame':
                att_items.append((slot, 1, [tuple(map(float, gold))], module))
            elif module in head_items:
                head_items[module].append((slot, int(bool(gold))))
            elif module in CONTRASTIVE:
                if len(gold) == 0:
                    out[module].append(0.0)                               # "no results found" (:129-130)
                    continue
                cont_items.append((module, slot, len(embs), len(gold), len(out[module])))
                out[module].append(None)
                embs.extend(torch.as_tensor(e, dtype=torch.float32) for _, e in gold)
            elif module == 'FilterFrame':
                ff_items.append((slot, gold))
            else:
                raise NotImplementedError('validation loss for %s' % module)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vec = res._arena(res.info.vec_off, res.info.n_vec, H)
    att = res._arena(res.info.att_off, res.info.n_att, T)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
    if att_items:
        slot = i32([a[0] for a in att_items]); K = i32([a[1] for a in att_items])
        off = i32(np.concatenate([[0], np.cumsum([a[1] for a in att_items])]).tolist())
        iv = torch.tensor([p for a in att_items for p in a[2]], dtype=torch.float64, device=dev)
        val = torch.empty(len(att_items), device=dev)
        check(lib.stair_loss_attention(C.c_void_p(att.data_ptr()), None, C.c_void_p(slot.data_ptr()), C.c_void_p(K.data_ptr()),
                                       C.c_void_p(off.data_ptr()), C.c_void_p(iv.data_ptr()), len(att_items), T,
                                       C.c_float(0.0), C.c_void_p(val.data_ptr()), stream))
        for a, v in zip(att_items, val.cpu().tolist()):
            out[a[3]].append(v)
    for module, items in head_items.items():
        if not items:
            continue
        if not model.config['have_pretrain_head']:
            raise RuntimeError('%s loss needs have_pretrain_head (modules.py)' % module)
        head = model.submodules[module].pretrain_head
        slot = i32([a[0] for a in items]); lab = i32([a[1] for a in items])
        val = torch.empty(len(items), device=dev)
        check(lib.stair_loss_head(head.weight.shape[0], C.c_void_p(vec.data_ptr()), None, C.c_void_p(slot.data_ptr()),
                                  C.c_void_p(lab.data_ptr()), C.c_void_p(head.weight.data_ptr()), C.c_void_p(head.bias.data_ptr()),
                                  None, None, len(items), H, C.c_float(0.0), C.c_void_p(val.data_ptr()), stream))
        out[module].extend(val.cpu().tolist())
    if ff_items:
        out['FilterFrame'].extend(_filterframe_launch(model, res, ff_items, 0.0, False).cpu().tolist())
    if cont_items:
        reps = model.encode_phrases(embs)                                  # [sum of gold sizes, H], L2-normalised
        seg = torch.repeat_interleave(torch.arange(len(cont_items), device=dev), i32([c[3] for c in cont_items]).long())
        mean = torch.zeros(len(cont_items), H, device=dev).index_add_(0, seg, reps)
        mean /= torch.tensor([c[3] for c in cont_items], dtype=torch.float32, device=dev).unsqueeze(1)
        pred = vec[i32([c[1] for c in cont_items]).long()]
        cos = torch.nn.functional.cosine_similarity(pred, mean, dim=1).cpu().tolist()
        for c, v in zip(cont_items, cos):
            out[c[0]][c[4]] = v
    return out
