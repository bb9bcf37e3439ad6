"""Training driver: the batched, data-parallel counterpart of the reference's loop
(/root/reference/train_module.py:341-412) for the decoder-loss configuration (module_loss_weight = 0):

    for each window of `global_batch` questions:        # the reference: gradient_accumulation = 32, batch 1
        loss = sum_i CE(logits_i, answer_i) * decoder_loss_weight / global_batch      (:376-380, :372)
        loss.backward(); Adam.step(); zero_grad(); LambdaLR.step()                     (:408-412)

One window = one batched forward + one HIP backward pass per rank (questions sharded across ranks);
gradients live in ONE flat fp32 buffer that is sum-reduced over RCCL once per step.  The buffer is laid out
`[module + decoder gradients | encoder gradients | touched mask | status word]`:

* the per-parameter "touched" mask rides behind the gradients (a parameter is touched if ANY rank's shard used
  its module), so that parameters of modules no rank used are skipped by Adam exactly as torch skips grad == None;
* the plan's status word (a cooperative recurrence that timed out: NaN gradients) rides there too, so the reduced
  word is the SAME on every rank: Adam's guard and Trainer.check() read it, and either every rank applies the
  step or none does (a failure local to one rank would otherwise poison the replicas through the sum and then
  leave its peers waiting in the next collective);
* with `overlap_allreduce` (default at world > 1) the bucket goes in two pieces: the module + decoder part -- final
  when the last program level's backward returns (stair_plan_set_backward_event) -- is reduced on a side stream
  while BPTT and the encoders' dW_ih / dW_hh products still run; `[encoder gradients | mask | status]` follows on
  the main stream.  Sums of two ranks are bit-identical to the one-piece reduction.

Parameters, gradients and Adam moments are flat buffers whose per-tensor segments start on multiples of 256
floats; the model's Parameters are views into them.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib

SEG = 256


def reduce_gradients(flat_g, touched, world, bucket=None, comm=None, status=None, split=None, side_stream=None, ready_event=None,
                     late_events=None):
    """The exchange step of data-parallel training: sum the flat fp32 gradient bucket over all ranks and OR the
    per-parameter touched mask (a parameter is "touched" if ANY rank's shard used its module).
    With `bucket` (a buffer whose head is flat_g and whose tail has room for the mask and one status word, as Trainer lays
    it out) the mask travels as floats behind the gradients and ONE all-reduce does both; without it, two collectives.
    status (optional, needs bucket): a one-element integer tensor, this rank's "the pass failed" word; it is summed in the
    word behind the mask and comes back as 1 on EVERY rank when any rank's was set.
    split (optional, needs bucket): the bucket goes as two collectives, bucket[:split] and bucket[split:] (the second holds
    mask and status).  On CUDA with side_stream + ready_event the first one is issued on side_stream once ready_event has
    fired -- beside whatever the current stream is still doing -- and the current stream waits for it at the end.
    late_events: an (e0, e1) pair of torch.cuda.Event(enable_timing=True) recorded on the current stream around the part of
    the exchange the current stream has to sit through (all of it in one-piece mode).
    Backend-agnostic (RCCL on GPUs, gloo in the CPU test); in place.  comm: a stair_amd.comm.NativeComm -- the bucket then
    goes through the C ABI's stair_allreduce_grads (RCCL from libstair_hip.so, same stream, no torch collective)."""
    if world <= 1:
        return flat_g, touched
    import torch.distributed as dist

    def allreduce(x):
        if comm is not None:
            comm.allreduce_(x)
        else:
            dist.all_reduce(x, op=dist.ReduceOp.SUM)

    if bucket is None:
        assert status is None and split is None
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX)
        return flat_g, touched
    n, k = flat_g.numel(), touched.numel()
    tail = bucket[n: n + k]
    tail.copy_(touched.to(bucket.dtype))
    if status is not None:
        bucket[n + k: n + k + 1].copy_((status != 0).to(bucket.dtype))
    if split is None:
        if late_events:
            late_events[0].record()
        allreduce(bucket)
    else:
        early, late = bucket[:split], bucket[split:]
        if side_stream is not None:
            with torch.cuda.stream(side_stream):
                if ready_event is not None:
                    side_stream.wait_event(ready_event)
                else:
                    side_stream.wait_stream(torch.cuda.current_stream())
                allreduce(early)
        else:
            allreduce(early)
        if late_events:
            late_events[0].record()
        allreduce(late)
        if side_stream is not None:
            torch.cuda.current_stream().wait_stream(side_stream)
    if late_events:
        late_events[1].record()
    touched.copy_((tail > 0).to(touched.dtype))
    if status is not None:
        status.copy_((bucket[n + k: n + k + 1] > 0).to(status.dtype))
    return flat_g, touched


class _PinnedRing:
    """Small host -> device uploads (the per-step touched mask) without tying the host to the stream: a pageable
    `.to(device)` makes the host wait until the stream has reached the copy -- the end of the backward pass -- and a single
    page-locked buffer has to wait for its previous copy, which is the same thing one step later.  A ring of `depth`
    page-locked buffers waits only for the copy issued `depth` steps ago, so the host may run that far ahead."""

    def __init__(self, numel, depth=8):
        self.bufs = [torch.empty(numel, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def upload(self, values, device):
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.bufs[k].copy_(torch.tensor(values, dtype=torch.int32))
        t = self.bufs[k].to(device, non_blocking=True)
        self.events[k] = torch.cuda.Event()
        self.events[k].record()
        return t


class Trainer:
    def __init__(self, model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoder_loss_weight=1.0,
                 module_loss_weight=1.0, contrastive_window=32, no_intermediate=('FilterFrame',),
                 scheduler_start_factor=1.0, scheduler_end_factor=0.1, scheduler_total_iters=200000, world=1,
                 skip_untouched='ever', train_module_before_iters=1e10, train_decoder_after_iters=0, rank=0,
                 dropout=None, dropout_seed=0, native_allreduce=False, class_table=None, global_batch=None, overlap_allreduce=None):
        """skip_untouched: 'ever'   -- a parameter is skipped by Adam until the first window that sends it a gradient
                                      (torch 1.13, which the reference pins: zero_grad() keeps zero tensors afterwards);
                           'window' -- skipped in every window that does not touch it (torch >= 2.0, set_to_none=True)."""
        # train_module.py:350,376 gate the two loss families by the reference's global_steps (one per QUESTION there):
        # intermediate losses while global_steps < train_module_before_iters, decoder loss once global_steps >
        # train_decoder_after_iters.  Question i of a rank's shard has global step seen + 1 + rank + i * world.
        # class_table (losses.ClassTable, the same on every rank): the contrastive pools of a data-parallel step are then
        # exchanged as a [windows, classes] presence matrix summed on the device -- without it every supervised step at
        # world > 1 gathers the ranks' class lists through the host (all_gather_object), a blocking second collective.
        # global_batch: the window size summed over all ranks when it is the same every step (a data loader with drop_last).
        # Without it -- here or in step() -- every step at world > 1 agrees on the size with an extra all-reduce and a host
        # read-back (.item(): the host then waits for the stream, which ends its run-ahead); ragged last windows need that.
        # overlap_allreduce: None = on at world > 1; False = the whole bucket in ONE collective after the backward pass.
        self.global_batch = global_batch
        self.overlap_allreduce = (world > 1) if overlap_allreduce is None else bool(overlap_allreduce)
        self.inject_failure = False                          # test hook: set this rank's status word after the backward pass
        self.class_table = class_table
        self.allreduce_events = None                        # set to [] to collect (start, end) events around the step's collective
        self.before_iters, self.after_iters, self.rank = train_module_before_iters, train_decoder_after_iters, rank
        self.questions_seen = 0
        # nn.Dropout(config['dropout']) of the reference's model.train() (modules.py `D` positions, args.py:31).  Default:
        # the model's own config['dropout'], i.e. the reference's training recipe; the parity pins are defined at dropout = 0
        # (torch's Philox masks cannot be reproduced) and the tests / bench pass dropout=0.0 explicitly.
        # Every step and rank draws fresh masks: seed = dropout_seed + step * world + rank.
        self.dropout = float(model.config.get('dropout', 0.0) if dropout is None else dropout)
        self.dropout_seed = int(dropout_seed)
        assert skip_untouched in ('ever', 'window')
        self.skip_untouched = skip_untouched
        self.model, self.world = model, world
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.decoder_loss_weight = decoder_loss_weight
        self.module_loss_weight, self.contrastive_window, self.no_intermediate = module_loss_weight, contrastive_window, no_intermediate
        self.sched = (scheduler_start_factor, scheduler_end_factor, scheduler_total_iters)
        self.iters = 0                                      # optimizer steps taken (LambdaLR counter)
        params = dict(model.named_parameters())
        names = model._weight_names
        dev = next(model.parameters()).device
        assert dev.type == 'cuda', 'Trainer needs the model on the GPU'
        # flat order: everything but the encoders first (their gradients are final before BPTT starts), the encoders last
        late = [i for i, nme in enumerate(names) if '_encoder.' in nme]
        order = [i for i in range(len(names)) if i not in set(late)] + late
        offs, total = [0] * len(names), 0
        self.split = None
        for i in order:
            if late and i == late[0]:
                self.split = total                           # bucket[:split] = module + decoder gradients, bucket[split:] = the rest
            offs[i] = total
            total += (params[names[i]].numel() + SEG - 1) // SEG * SEG
        self.n = total
        self.flat_p = torch.zeros(total, device=dev)
        mask_room = (len(names) + 1 + SEG - 1) // SEG * SEG   # touched mask as floats + the status word
        self.bucket = torch.zeros(total + mask_room, device=dev)       # [gradients | touched mask | status]
        self.flat_g = self.bucket[:total]
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        seg_of_block = torch.empty(total // SEG, dtype=torch.int32)
        for i, nme in enumerate(names):
            p = params[nme]
            view = self.flat_p[offs[i]: offs[i] + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view                                    # Parameters become views of the flat buffer
            p.grad = self.flat_g[offs[i]: offs[i] + p.numel()].view_as(p)
            nb = (p.numel() + SEG - 1) // SEG
            seg_of_block[offs[i] // SEG: offs[i] // SEG + nb] = i
        self.seg_of_block = seg_of_block.to(dev)
        self.steps = torch.zeros(len(names), device=dev)     # per-tensor Adam step counts
        self.touched = torch.zeros(len(names), dtype=torch.int32, device=dev)
        self._mask_ring = _PinnedRing(len(names))
        self._status = []                                    # (step, pinned int32 copy of the step's REDUCED status word, event) per step in flight
        self.guard = torch.zeros(1, dtype=torch.int32, device=dev)      # status of the current step, summed over ranks: Adam's guard
        self._side = self._ev_ready = None
        if self.overlap_allreduce and world > 1 and self.split:
            self._side = torch.cuda.Stream(device=dev)
            self._ev_ready = torch.cuda.Event()
            self._ev_ready.record()                          # creates the hipEvent_t the backward pass records (stair_plan_set_backward_event)
        self.allreduce_late_events = None                    # set to [] to collect (e0, e1) around the exposed part of the exchange
        self.comm = None
        if native_allreduce and world > 1:          # the step's one collective through the C ABI (stair_allreduce_grads)
            from .comm import NativeComm
            self.comm = NativeComm(rank, world)
        self.offsets = offs

    def lr_factor(self):
        """train_module.py:328-331."""
        start, end, total = self.sched
        return end if self.iters > total else start + (end - start) / total * self.iters

    def step(self, programs, spans, video, question, q_lens, answers, global_batch=None, questions=None, video_index=None,
             video_len=None):
        """One optimizer step over this rank's shard of a window.  `questions` -- the dicts, with 'sg_res_by_step', or their
        losses.GoldBatch (losses.collate_gold: the data loader's collate step; the per-question bookkeeping then costs the stepping
        process nothing) -- switches the per-module intermediate losses on (train_module.py:351-373, 388-406).
        video_index: questions that share a clip (see VideoNMN.run_programs).
        Returns (per-question decoder CE of the local shard, BatchResult)."""
        from . import losses as L
        n = len(programs)
        if global_batch is None:
            global_batch = self.global_batch
        if global_batch is None and self.world > 1:           # ragged shards: the window is the SUM of the shard sizes
            import torch.distributed as dist
            cnt = torch.tensor([n], dtype=torch.int64, device=self.flat_g.device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            global_batch = int(cnt.item())
        G = global_batch or n
        self.check(wait=False)                                # a failed earlier step surfaces here at the latest
        self.flat_g.zero_()                                   # optimizer.zero_grad()
        gstep = [self.questions_seen + 1 + self.rank + i * self.world for i in range(n)]
        self.questions_seen += G
        if gstep[0] <= self.after_iters:                      # some questions still without decoder loss
            answers = torch.where(torch.tensor([g > self.after_iters for g in gstep], device=answers.device), answers,
                                  torch.full_like(answers, -1))
        if questions is not None and gstep[-1] >= self.before_iters:
            if isinstance(questions, L.GoldBatch):
                questions = questions.select([g < self.before_iters for g in gstep])
            else:
                questions = [q if g < self.before_iters else dict(q, sg_res_by_step={}) for q, g in zip(questions, gstep)]
        drop = (self.dropout, self.dropout_seed + self.iters * self.world + self.rank) if self.dropout > 0 else None
        supervised = questions is not None and self.module_loss_weight != 0
        prep = {}

        def prepare(res_):        # host side of the intermediate losses: needs the plan only, runs before the pass is enqueued
            prep['p'] = L.prepare_module_losses(self.model, res_, questions, no_intermediate=self.no_intermediate,
                                                window=self.contrastive_window, world=self.world, rank=self.rank,
                                                class_table=self.class_table, global_batch=G)
        res = self.model.run_programs(programs, spans, video, question, q_lens, train=True, video_index=video_index, dropout=drop,
                                      video_len=video_len, before_run=prepare if supervised else None)
        extra = set()
        tl = res.touched()                                    # known from the plan: uploaded BEFORE the backward pass is enqueued
        if supervised:
            res.zero_grad_arenas()
            self.module_losses, extra = L.launch_module_losses(self.model, res, prep['p'], self.module_loss_weight / G)
            if extra:
                tl = [t_ or (nme in extra) for t_, nme in zip(tl, self.model._weight_names)]
        t = self._mask_ring.upload(tl, self.touched.device)
        overlap = self._side is not None
        loss = res.backward(answers, self.decoder_loss_weight / G, keep_arenas=supervised,
                            ready_event=self._ev_ready if overlap else None)
        if self.inject_failure:
            res.status_word().fill_(1)
        self.guard.copy_(res.status_word())
        late_ev = None
        if self.allreduce_late_events is not None and self.world > 1:
            late_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.allreduce_late_events.append(late_ev)
        e0 = e1 = None
        if self.allreduce_events is not None and self.world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # the exchange: gradients summed, mask OR-ed, status OR-ed -- over RCCL / xGMI; in two pieces when overlapped
        reduce_gradients(self.flat_g, t, self.world, self.bucket, self.comm, status=self.guard,
                         split=self.split if overlap else None, side_stream=self._side, ready_event=self._ev_ready, late_events=late_ev)
        if e0 is not None:
            e1.record()
            self.allreduce_events.append((e0, e1))
        # bookkeeping of a step the guard refuses stays that of the last good step (on the device: the host does not wait)
        ok = (self.guard == 0).to(torch.int32)
        if self.skip_untouched == 'ever':
            self.touched = torch.maximum(self.touched, t * ok)
        else:
            self.touched.copy_(t)
        self.steps += (self.touched * ok).to(torch.float32)
        lr = self.lr * self.lr_factor()
        check(lib.stair_adam_step(C.c_void_p(self.flat_p.data_ptr()), C.c_void_p(self.flat_g.data_ptr()),
                                  C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                  C.c_void_p(self.seg_of_block.data_ptr()), C.c_void_p(self.touched.data_ptr()),
                                  C.c_void_p(self.steps.data_ptr()), C.c_float(lr), C.c_float(self.betas[0]),
                                  C.c_float(self.betas[1]), C.c_float(self.eps), C.c_float(self.wd), self.n,
                                  C.c_void_p(self.guard.data_ptr()),          # guard: no update when ANY rank's pass failed
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if len(self._status) < 16:
            host = torch.empty(1, dtype=torch.int32).pin_memory()
        else:                                                 # the oldest entry's buffer is reused: look at it first
            self._raise_if_failed(*self._status[0])
            host = self._status.pop(0)[1]
        host.copy_(self.guard, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._status.append((self.iters, host, ev))
        self.iters += 1                                       # scheduler.step()
        return loss, res

    def _raise_if_failed(self, step, host, ev):
        from ._lib import StairError
        ev.synchronize()
        if int(host[0]) != 0:
            self._status = []
            raise StairError('optimizer step %d was skipped on every rank: a cooperative LSTM recurrence timed out on at least one of '
                             'them (its workgroups were not co-resident -- another queue on that GPU?); weights, Adam moments and '
                             'per-tensor step counts are those of the last good step.  STAIR_LSTM_COOP=0 selects the '
                             'one-workgroup kernels' % step)

    def check(self, wait=True):
        """Raises StairError if a step reported a failed pass (a cooperative LSTM hand-off that timed out: the gradients of
        that step held NaN).  The word it reads is the step's status summed over all ranks -- the one Adam's guard read -- so at
        world > 1 every rank raises for the same step, whichever rank failed, and none of them has applied the update:
        parameters, moments and per-tensor step counts are those of the last good step.  The host-side counters (`iters`, the
        LambdaLR position, and `questions_seen`) have advanced past the refused window, as the data loader has.
        wait=False looks only at steps the GPU has finished."""
        keep = []
        for entry in self._status:
            if not wait and not entry[2].query():
                keep.append(entry)
                continue
            self._raise_if_failed(*entry)
        self._status = keep[-16:]
