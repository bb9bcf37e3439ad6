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


def slot_groups(slot):
    """Items that add into the same gradient slot, for stair_loss_groups: (order int32 [n], grp_off int32 [groups + 1]); order
    is a stable sort by slot, so inside a group the items keep their batch order (the order the sums are formed in)."""
    slot = np.asarray(slot, dtype=np.int64)
    order = np.argsort(slot, kind='stable')
    ss = slot[order]
    starts = np.flatnonzero(np.concatenate([[True], ss[1:] != ss[:-1]])) if len(ss) else np.zeros(0, dtype=np.int64)
    return np.ascontiguousarray(order, dtype=np.int32), np.ascontiguousarray(np.concatenate([starts, [len(ss)]]), dtype=np.int32)


def _filterframe_launch(model, res, items, scale, grads):
    """items: [(map slot, gold dict)] or [(map slot, gold dict, frames of the item's clip)] when the batch mixes clip
    lengths.  Returns the per-item losses (device tensor)."""
    dev = res.logits.device
    H, T, O = model.config['hidden_size'], res.info.T, model.config['object_types']
    index = getattr(model, 'object_index', None)
    if index is None:
        raise RuntimeError('FilterFrame loss needs model.object_index = losses.object_index(json.load(open(word2id_filename)))')
    head = model.submodules['FilterFrame'].pretrain_head
    if grads and head.weight.grad is None:
        raise RuntimeError('pretrain head of FilterFrame has no .grad buffer (use stair_amd.train.Trainer)')
    frames = [int(a[2]) if len(a) > 2 else T for a in items]

    def target(gold_dict, L):            # the criterion sees pred.size(0) = the clip's own frames (train_module.py:146)
        g = np.zeros((T, O), dtype=np.float32)
        g[:L] = filterframe_target(gold_dict, L, O, index)
        return g
    gold = torch.from_numpy(np.stack([target(a[1], L) for a, L in zip(items, frames)])).to(dev)
    slot = torch.tensor([a[0] for a in items], dtype=torch.int32, device=dev)
    len_d = torch.tensor(frames, dtype=torch.int32, device=dev) if any(L != T for L in frames) else None
    out = torch.empty(len(items), device=dev)
    inf = res.info
    mp = res._ws[inf.map_off: inf.map_off + inf.n_map * T * H]
    gmap = res.grad_arena('map') if grads else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if grads:                           # reproducible sums: items of one tile evaluated by one workgroup, in order
        order, goff = slot_groups([a[0] for a in items])
        order_d, goff_d = torch.from_numpy(order).to(dev), torch.from_numpy(goff).to(dev)
        check(lib.stair_loss_groups(p(order_d), p(goff_d), len(goff) - 1))
    check(lib.stair_loss_filterframe_len(p(mp), p(gmap), p(slot), p(gold), p(head.weight), p(head.bias),
                                         p(head.weight.grad) if grads else None, p(head.bias.grad) if grads else None, p(len_d),
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


class GoldBatch:
    """The gold intermediates of a BATCH as flat arrays -- what a data loader's collate step hands to the training loop (the
    reference's loader hands batch-1 dicts, dataset.py:463-464; its per-question bookkeeping, train_module.py:351-406, then runs in
    the stepping process).  With a GoldBatch, prepare_module_losses is a handful of numpy operations whatever the batch size;
    from a list of question dicts it walks the questions in Python (6.7 ms per 2048 questions on the bench host).
      att_q / att_pos / att_kind / att_iv      attention criteria: question index, token position, kind, gold intervals [n, 2, 2]
      head[module] = (q, pos, label)            Exists / Xor / Equals
      cg_q / cg_pos / cg_name (/ cg_cls)       one entry per (contrastive node, gold class): names, and their ClassTable rows
      cg_emb                                   the embeddings of those entries (list mode only)
      ff = [(q, pos, gold dict)]               FilterFrame (off by default)"""

    def __init__(self, questions, pretrain_modules=CRITERION_MODULES, no_intermediate=('FilterFrame',), class_table=None):
        packs = [compile_gold(q, pretrain_modules, no_intermediate) for q in questions]
        self.n = len(packs)
        self.key = (frozenset(pretrain_modules), tuple(no_intermediate))
        cat = lambda parts, dt, shape=None: (np.concatenate(parts).astype(dt, copy=False) if parts else np.zeros((0,) + (shape or ()), dtype=dt))
        sel = [qi for qi, p in enumerate(packs) if p.att_pos.size]
        self.att_q = cat([np.full(packs[qi].att_pos.size, qi, dtype=np.int64) for qi in sel], np.int64)
        self.att_pos = cat([packs[qi].att_pos for qi in sel], np.int64)
        self.att_kind = cat([packs[qi].att_kind for qi in sel], np.int64)
        self.att_iv = cat([packs[qi].att_iv for qi in sel], np.float64, (2, 2))
        self.head = {}
        for module in ('Exists', 'Xor', 'Equals'):
            sel = [qi for qi, p in enumerate(packs) if module in p.head]
            if sel:
                self.head[module] = (cat([np.full(packs[qi].head[module][0].size, qi, dtype=np.int64) for qi in sel], np.int64),
                                     cat([packs[qi].head[module][0] for qi in sel], np.int64),
                                     cat([packs[qi].head[module][1] for qi in sel], np.int32))
        cq, cp, cn, ce = [], [], [], []
        for qi, p in enumerate(packs):
            for pos, module, gold in p.cont:
                for class_name, emb in gold:
                    cq.append(qi); cp.append(pos); cn.append(class_name); ce.append(emb)
        self.cg_q, self.cg_pos = np.asarray(cq, dtype=np.int64), np.asarray(cp, dtype=np.int64)
        self.cg_name, self.cg_emb = cn, ce
        self.class_table = class_table
        self.cg_cls = None
        if class_table is not None:
            try:
                self.cg_cls = np.asarray([class_table.index[n] for n in cn], dtype=np.int64)
            except KeyError as e:
                raise KeyError('class %s is not in the ClassTable (build it from the whole dataset: ClassTable.from_questions)' % e)
        self.ff = [(qi, pos, gold) for qi, p in enumerate(packs) for pos, gold in p.ff]

    def select(self, keep):
        """The batch with the gold of the questions where keep[q] is False removed (train_module.py:350: no intermediate losses
        once global_steps >= train_module_before_iters)."""
        keep = np.asarray(keep, dtype=bool)
        out = object.__new__(GoldBatch)
        out.n, out.key, out.class_table = self.n, self.key, self.class_table
        m = keep[self.att_q]
        out.att_q, out.att_pos, out.att_kind, out.att_iv = self.att_q[m], self.att_pos[m], self.att_kind[m], self.att_iv[m]
        out.head = {}
        for module, (q, pos, lab) in self.head.items():
            m = keep[q]
            if m.any():
                out.head[module] = (q[m], pos[m], lab[m])
        m = keep[self.cg_q] if self.cg_q.size else np.zeros(0, dtype=bool)
        idx = np.nonzero(m)[0]
        out.cg_q, out.cg_pos = self.cg_q[m], self.cg_pos[m]
        out.cg_name, out.cg_emb = [self.cg_name[i] for i in idx], [self.cg_emb[i] for i in idx]
        out.cg_cls = self.cg_cls[m] if self.cg_cls is not None else None
        out.ff = [t for t in self.ff if keep[t[0]]]
        return out


def collate_gold(questions, pretrain_modules=CRITERION_MODULES, no_intermediate=('FilterFrame',), class_table=None):
    """GoldBatch of a list of question dicts (the data loader's collate step)."""
    return GoldBatch(questions, pretrain_modules, no_intermediate, class_table)


def contrastive_windows(entries, window, world=1):
    """The class pools of train_module.py:386-402 over the GLOBAL accumulation window.
    entries: [(global question position, class_name, embedding)] of this rank's contrastive golds; with world > 1 every
    rank contributes its entries (all_gather_object: a few KB of host data) and builds the same tables, so that a
    data-parallel step pools exactly the classes the single-process window of the reference pools.
    A class representation depends on the class name only (text encoder without gradient + L2Normalize,
    module_net.py:78-89), so each DISTINCT class is encoded once per step and the windows index into that table.
    Returns (names: distinct class names, sorted; embs: their embeddings; rows: int32 row of `names` for every window
    entry, window after window; win_range: {window id: (first entry, count)}; slot_of: {(window id, class_name): entry})."""
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, [(g, n, np.asarray(e, dtype=np.float32)) for g, n, e in entries])
        entries = [t for part in gathered for t in part]
    first, pools = {}, {}
    for gpos, name, emb in entries:
        first.setdefault(name, emb)
        pools.setdefault(gpos // window if window else 0, set()).add(name)
    names = sorted(first)
    row_of = {n: i for i, n in enumerate(names)}
    rows, win_range, slot_of = [], {}, {}
    for wid in sorted(pools):
        start = len(rows)
        for name in sorted(pools[wid]):
            slot_of[(wid, name)] = len(rows)
            rows.append(row_of[name])
        win_range[wid] = (start, len(rows) - start)
    return names, [first[n] for n in names], np.asarray(rows, dtype=np.int32), win_range, slot_of


class ClassTable:
    """Every class name the contrastive criteria can meet (AGQA: the phrases of data/AGQA/filter_answers.json) in ONE order
    shared by all ranks, with its word embeddings [L, text_size] (dataset.py:200-221 hands them over per question).  With
    a table the data-parallel step needs no per-step exchange of class lists between hosts: a rank marks the classes of its
    own questions in a [windows, classes] presence matrix on the device and that matrix is summed over the ranks by one
    small all-reduce on the stream (prepare_module_losses); stair_loss_contrastive_table reads the pools from it."""

    def __init__(self, names, embeddings):
        order = sorted(range(len(names)), key=lambda i: names[i])
        self.names = [names[i] for i in order]
        if len(set(self.names)) != len(self.names):
            raise ValueError('duplicate class names')
        self.embeddings = [np.asarray(embeddings[i], dtype=np.float32) for i in order]
        self.index = {n: i for i, n in enumerate(self.names)}
        self._dev = None

    def __len__(self):
        return len(self.names)

    @classmethod
    def from_questions(cls, questions, world=1):
        """Collect the table from the gold intermediates of a dataset (every rank: its own shard; the shards' tables are
        merged ONCE here, at start-up -- this is the only host-side exchange of class names)."""
        found = {}
        for q in questions:
            for gold in (q.get('sg_res_by_step') or {}).values():
                if isinstance(gold, list):
                    for name, emb in gold:
                        found.setdefault(name, np.asarray(emb, dtype=np.float32))
        if world > 1:
            import torch.distributed as dist
            parts = [None] * world
            dist.all_gather_object(parts, found)
            found = {}
            for part in parts:
                for name, emb in part.items():
                    found.setdefault(name, emb)
        names = sorted(found)
        return cls(names, [found[n] for n in names])

    def device_rows(self, device, text_size):
        """(x [sum L, E], seq_off [n_cls + 1], max L) on the device, uploaded once."""
        if self._dev is None or self._dev[0].device != device:
            embs = [e.reshape(-1, text_size) for e in self.embeddings]
            lens = [int(e.shape[0]) for e in embs]
            x = torch.from_numpy(np.ascontiguousarray(np.concatenate(embs))).to(device)
            off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=device)
            self._dev = (x, off, max(lens))
        return self._dev


class _Staging:
    """A ring of pinned host buffers + device buffers per model: every index / label / interval array of a step's loss launches
    goes to the GPU in ONE asynchronous copy (a pageable `.to(device)` per array blocks the host on the stream each time).
    A single buffer makes the host wait until the GPU has executed the PREVIOUS step's copy -- i.e. it cannot run ahead of the
    GPU by more than one step; with `depth` slots a slot is waited for only when it comes round again."""

    def __init__(self, depth=4):
        self.slots = [dict(host=None, dev=None, event=None) for _ in range(depth)]
        self.at = 0

    def upload(self, arrays, device):
        """arrays: list of contiguous numpy arrays (int32 / float32 / float64).  Returns device views in the same order, valid until
        the slot is reused `depth` uploads later."""
        offs, total = [], 0
        for a in arrays:
            total = (total + 15) // 16 * 16               # kernels want 16-byte aligned operands
            offs.append(total)
            total += a.nbytes
        total = max(16, (total + 15) // 16 * 16)
        sl = self.slots[self.at]
        self.at = (self.at + 1) % len(self.slots)
        if sl['event'] is not None:
            sl['event'].synchronize()                     # the copy issued `depth` uploads ago has left this slot's pinned buffer
        if sl['host'] is None or sl['host'].numel() < total or sl['dev'].device != device:
            cap = max(total * 2, 1 << 16)
            sl['host'] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            sl['dev'] = torch.empty(cap, dtype=torch.uint8, device=device)
        hv = sl['host'].numpy()
        for a, o in zip(arrays, offs):
            hv[o: o + a.nbytes] = a.view(np.uint8).reshape(-1)
        sl['dev'][:total].copy_(sl['host'][:total], non_blocking=True)
        sl['event'] = torch.cuda.Event()
        sl['event'].record()
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}
        return [sl['dev'][o: o + a.nbytes].view(tdt[a.dtype]) for a, o in zip(arrays, offs)]


def prepare_module_losses(model, res, questions, pretrain_modules=CRITERION_MODULES, no_intermediate=('FilterFrame',), window=32,
                          world=1, rank=0, window_base=0, class_table=None, global_batch=None):
    """Everything of apply_module_losses that needs the PLAN but not the forward results: the index / target arrays of every
    criterion (one upload), the contrastive class tables and the encoding of the distinct classes.  Call it from
    VideoNMN.run_programs(before_run=...) so that this host work (a few ms per 2048 questions) is done before the forward
    pass is enqueued and the stream goes from the forward straight into the loss kernels and the backward pass.
    class_table (a ClassTable shared by all ranks) + global_batch: the contrastive pools are exchanged as a presence matrix on
    the device instead of class lists between hosts (see ClassTable).
    Returns an opaque dict for launch_module_losses."""
    dev = res.pred.device
    H, T = model.config['hidden_size'], res.info.T
    _, slot_t, aux_t, _, rel_t = res.node_table()
    base = np.asarray(res._prog_off, dtype=np.int64)
    # a GoldBatch (the loader's collate output) or a list of question dicts, collated here
    gb = questions if isinstance(questions, GoldBatch) else GoldBatch(questions, pretrain_modules, no_intermediate, class_table)
    if gb.n != res.info.n_questions:
        raise ValueError('gold batch of %d questions for a plan of %d' % (gb.n, res.info.n_questions))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    up, plan = [], {}                                    # arrays to upload, and where each launch finds its own

    def stage(key, *arrays):
        plan[key] = (len(up), len(arrays))
        up.extend(arrays)

    # ---- attention criteria (Localize / Temporal / ExistsFrame) ----
    n_att = int(gb.att_q.size)
    if n_att:
        tok = gb.att_pos + base[gb.att_q]
        kind, iv = gb.att_kind, gb.att_iv                                        # iv [n, 2, 2]
        slot = np.where(kind == 1, rel_t[tok], slot_t[tok])
        K = np.where(kind == 0, aux_t[tok], 1).astype(np.int64)
        keep = np.arange(2)[None, :] < K[:, None]
        qf = res.question_frames
        att_len = i32(np.asarray(qf)[gb.att_q]) if qf is not None else np.zeros(0, np.int32)
        stage('att', i32(slot), i32(K), i32(np.concatenate([[0], np.cumsum(K)])), np.ascontiguousarray(iv[keep], dtype=np.float64), att_len,
              *slot_groups(slot))
    # ---- linear heads (Exists / Xor / Equals) ----
    n_head = {}
    for module in ('Exists', 'Xor', 'Equals'):
        if module not in gb.head:
            continue
        if not model.config['have_pretrain_head']:
            raise RuntimeError('%s loss needs have_pretrain_head (modules.py)' % module)
        if model.submodules[module].pretrain_head.weight.grad is None:
            raise RuntimeError('pretrain head of %s has no .grad buffer (use stair_amd.train.Trainer)' % module)
        hq, hpos, hlab = gb.head[module]
        tok = hpos + base[hq]
        n_head[module] = len(tok)
        stage(module, i32(slot_t[tok]), i32(hlab), *slot_groups(slot_t[tok]))
    # ---- contrastive (Filter / ToAction / Superlative) ----
    n_cg = int(gb.cg_q.size)
    c_slot = slot_t[base[gb.cg_q] + gb.cg_pos] if n_cg else np.zeros(0, dtype=np.int64)
    c_gpos = window_base + rank + gb.cg_q * world
    c_wid = (c_gpos // window if window else np.zeros_like(c_gpos))
    c_name = gb.cg_name
    table_mode = class_table is not None
    if table_mode:
        G_ = int(global_batch if global_batch is not None else gb.n * world)
        wid0 = window_base // window if window else 0
        n_win = ((window_base + G_ - 1) // window if window else 0) - wid0 + 1
        presence = np.zeros((n_win, len(class_table)), dtype=np.float32)
        if gb.cg_cls is not None and gb.class_table is class_table:
            c_cls = gb.cg_cls
        else:
            try:
                c_cls = np.asarray([class_table.index[n] for n in c_name], dtype=np.int64)
            except KeyError as e:
                raise KeyError('class %s is not in the ClassTable (build it from the whole dataset: ClassTable.from_questions)' % e)
        if n_cg:
            presence[c_wid - wid0, c_cls] = 1.0
        stage('cont', i32(c_slot), i32(c_cls), i32(c_wid - wid0), presence.reshape(-1), *slot_groups(c_slot))
    entries = [] if table_mode else [(int(g), n, e) for g, n, e in zip(c_gpos, c_name, gb.cg_emb)]
    c_slot, c_wid = (c_slot, c_wid) if table_mode else (c_slot.tolist(), c_wid.tolist())
    cw = contrastive_windows(entries, window, world) if ((n_cg or world > 1) and not table_mode) else None
    lens, max_classes = [], 0
    if n_cg and not table_mode:
        names, embs, rows, win_range, slot_of = cw
        embs = [np.asarray(e, dtype=np.float32).reshape(-1, model.config['text_size']) for e in embs]
        lens = [int(e.shape[0]) for e in embs]
        max_classes = max(r[1] for r in win_range.values())
        stage('cont', i32(c_slot), i32([slot_of[(w, n)] for w, n in zip(c_wid, c_name)]),
              i32([win_range[w][0] for w in c_wid]), i32([win_range[w][1] for w in c_wid]), rows,
              i32(np.concatenate([[0], np.cumsum(lens)])), np.ascontiguousarray(np.concatenate(embs)), *slot_groups(c_slot))
    qf_ = res.question_frames
    ff_items = [(int(slot_t[base[qi] + pos]), gold, int(qf_[qi]) if qf_ is not None else T) for qi, pos, gold in gb.ff]
    # ---- ONE upload ----
    staging = model.__dict__.setdefault('_loss_staging', _Staging())
    d = staging.upload(up, dev) if up else []
    got = lambda key: d[plan[key][0]: plan[key][0] + plan[key][1]]
    prep = {'n_att': n_att, 'n_head': n_head, 'n_cont': n_cg, 'ff_items': ff_items, 'max_classes': max_classes,
            'att': got('att') if n_att else None, 'head': {m: got(m) for m in n_head}}
    if table_mode:
        slot_d, cls_d, win_d, pres_d, c_order_d, c_goff_d = got('cont')
        prep['cont_groups'] = (c_order_d, c_goff_d)
        if world > 1:                       # the pools of the GLOBAL windows: one small device all-reduce, no host round trip
            import torch.distributed as dist
            dist.all_reduce(pres_d, op=dist.ReduceOp.SUM)
        if n_cg:
            x, seq_off, max_len = class_table.device_rows(dev, model.config['text_size'])
            _, h_n = ops.lstm_bidir(x, seq_off, max_len, [w.detach() for w in model._lstm_weights('text_encoder')])
            prep['cont_table'] = (slot_d, cls_d, win_d, pres_d, ops.l2normalize(h_n), len(class_table))
    elif n_cg:
        # class representations: text encoder without gradient + L2Normalize (module_net.py:78-89); they depend on the weights
        # only, so they are encoded here, ahead of the forward pass, on the same stream
        slot_d, pos_d, ws_d, wc_d, rows_d, seq_off, x, c_order_d, c_goff_d = got('cont')
        prep['cont_groups'] = (c_order_d, c_goff_d)
        x = x.view(-1, model.config['text_size'])
        _, h_n = ops.lstm_bidir(x, seq_off, max(lens), [w.detach() for w in model._lstm_weights('text_encoder')])
        prep['cont'] = (slot_d, pos_d, ws_d, wc_d, ops.l2normalize(h_n).index_select(0, rows_d.long()))    # every window's classes, window after window
    return prep


def launch_module_losses(model, res, prep, scale):
    """The loss kernels of a prepared batch (prepare_module_losses): values + scale * gradient into res's gradient arenas.
    Returns ({loss_kind: per-item losses tensor}, set of extra parameter names that received a gradient)."""
    dev = res.pred.device
    H, T = model.config['hidden_size'], res.info.T
    losses, touched = {}, set()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vec, gvec = res._arena(res.info.vec_off, res.info.n_vec, H), res.grad_arena('vec')
    att, gatt = res._arena(res.info.att_off, res.info.n_att, T), res.grad_arena('att')
    P = lambda t: C.c_void_p(t.data_ptr())
    # reproducible sums (like the decoder-only step): head-weight gradients through the context's fixed-point shadows, which the
    # backward pass that follows adds to the gradients; arena gradients group by group (stair_loss_groups)
    model._bind_grads()                 # the shadows are found through the bound gradient buffers
    check(lib.stair_grad_shadows_begin(model._ctx, stream))
    groups = lambda order_d, goff_d: check(lib.stair_loss_groups(P(order_d), P(goff_d), goff_d.numel() - 1))
    if prep['n_att']:
        slot_d, K_d, off_d, iv_d, len_d, order_d, goff_d = prep['att']
        out = torch.empty(prep['n_att'], device=dev)
        groups(order_d, goff_d)
        check(lib.stair_loss_attention_len(P(att), P(gatt), P(slot_d), P(K_d), P(off_d), P(iv_d), P(len_d) if len_d.numel() else None,
                                           prep['n_att'], T, C.c_float(scale), P(out), stream))
        losses['attention'] = out
    for module, n_items in prep['n_head'].items():
        head = model.submodules[module].pretrain_head
        slot_d, lab_d, order_d, goff_d = prep['head'][module]
        out = torch.empty(n_items, device=dev)
        groups(order_d, goff_d)
        check(lib.stair_loss_head(head.weight.shape[0], P(vec), P(gvec), P(slot_d), P(lab_d), P(head.weight), P(head.bias),
                                  P(head.weight.grad), P(head.bias.grad), n_items, H, C.c_float(scale), P(out), stream))
        losses[module] = out
        touched.update({'submodules.%s.pretrain_head.weight' % module, 'submodules.%s.pretrain_head.bias' % module})
    # ---- FilterFrame (off by default, args.py:62) ----
    if prep['ff_items']:
        losses['FilterFrame'] = _filterframe_launch(model, res, prep['ff_items'], scale, True)
        touched.update({'submodules.FilterFrame.pretrain_head.weight', 'submodules.FilterFrame.pretrain_head.bias'})
    if prep['n_cont'] and 'cont_table' in prep:
        slot_d, cls_d, win_d, pres_d, reps, n_cls = prep['cont_table']
        out = torch.empty(prep['n_cont'], device=dev)
        groups(*prep['cont_groups'])
        check(lib.stair_loss_contrastive_table(P(vec), P(gvec), P(slot_d), P(cls_d), P(win_d), P(pres_d), P(reps), prep['n_cont'],
                                               n_cls, H, C.c_float(scale), P(out), stream))
        losses['contrastive'] = out
    elif prep['n_cont']:
        slot_d, pos_d, ws_d, wc_d, G = prep['cont']
        out = torch.empty(prep['n_cont'], device=dev)
        groups(*prep['cont_groups'])
        check(lib.stair_loss_contrastive(P(vec), P(gvec), P(slot_d), P(pos_d), P(ws_d), P(wc_d), P(G), prep['n_cont'], H,
                                         prep['max_classes'], C.c_float(scale), P(out), stream))
        losses['contrastive'] = out
    return losses, touched


def apply_module_losses(model, res, questions, scale, pretrain_modules=CRITERION_MODULES,
                        no_intermediate=('FilterFrame',), window=32, world=1, rank=0, window_base=0, class_table=None,
                        global_batch=None):
    """Evaluate every intermediate loss of the batch and add scale * gradient into res's gradient arenas
    (call res.zero_grad_arenas() first and res.backward(..., keep_arenas=True) afterwards).
    Local question i sits at global position window_base + rank + i * world of the accumulation window (the round-robin
    sharding of Trainer.step); contrastive classes are pooled per `window` GLOBAL questions (contrastive_windows).
    Host work per step: concatenating the questions' gold packs (losses.compile_gold, prepared once per question),
    one gather through the plan's node table and ONE upload; no per-node library call.  = prepare_module_losses +
    launch_module_losses; Trainer.step calls the two halves either side of the forward pass.
    Returns ({loss_kind: per-item losses tensor}, set of extra parameter names that received a gradient)."""
    prep = prepare_module_losses(model, res, questions, pretrain_modules, no_intermediate, window, world, rank, window_base,
                                 class_table, global_batch)
    return launch_module_losses(model, res, prep, scale)


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
                att_items.append((slot, aux, [tuple(map(float, gold[r])) for r in range(aux)], module, qi))
            elif module == 'Temporal':
                att_items.append((rel, 1, [tuple(map(float, gold))], module, qi))
            elif module == 'ExistsFrame':
                att_items.append((slot, 1, [tuple(map(float, gold))], module, qi))
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
                ff_items.append((slot, gold) if res.question_frames is None else (slot, gold, int(res.question_frames[qi])))
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
        qf = res.question_frames
        lens = i32([int(qf[a[4]]) for a in att_items]) if qf is not None else None
        check(lib.stair_loss_attention_len(C.c_void_p(att.data_ptr()), None, C.c_void_p(slot.data_ptr()), C.c_void_p(K.data_ptr()),
                                           C.c_void_p(off.data_ptr()), C.c_void_p(iv.data_ptr()),
                                           C.c_void_p(lens.data_ptr()) if lens is not None else None, len(att_items), T,
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
        cnt = np.asarray([c[3] for c in cont_items], dtype=np.int64)
        seg_off = i32(np.concatenate([[0], np.cumsum(cnt)]).tolist())
        slot_d = i32([c[1] for c in cont_items])
        score = torch.empty(len(cont_items), device=dev)
        check(lib.stair_score_cosine_to_mean(C.c_void_p(vec.data_ptr()), C.c_void_p(slot_d.data_ptr()), C.c_void_p(reps.data_ptr()),
                                             C.c_void_p(seg_off.data_ptr()), C.c_void_p(score.data_ptr()), len(cont_items), H, stream))
        cos = score.cpu().tolist()
        for c, v in zip(cont_items, cos):
            out[c[0]][c[4]] = v
    return out
