"""Host-side mirror of the reference's ``video_nmn/module_net.py``: a ``VideoNMN`` with the same
constructor, ``forward(data, ...)`` contract, ``state_dict`` keys and helper entry points
(/root/reference/video_nmn/module_net.py:11-216), executing on libstair_hip.so.

Differences that matter to a caller:

* ``forward_batch`` runs MANY questions in one pass (the reference is batch-1 by construction,
  train_module.py:282); ``forward(data)`` is the drop-in single-question form built on it.
* no autograd graph: training goes through ``run_programs(train=True)`` + ``BatchResult.backward`` (HIP
  backward kernels) driven by ``stair_amd.train.Trainer``; only the decoder cross-entropy loss is built.
* programs whose operand kinds do not fit a module raise ``StairError`` at plan-build time instead
  of a torch shape error in the middle of execution.

torch is plumbing here (device memory, streams, Parameter containers); the arithmetic is HIP.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np
import torch
from torch import nn

from . import frontend, ops, spec
from ._lib import PlanInfo, StairConfig, StairError, check, lib

VAL_STR, VAL_VEC, VAL_MAP, VAL_ATT, VAL_FRAME, VAL_PAIR = range(6)
RUN_INDEX_RESIDENT, RUN_VIDEO_BF16, RUN_PROJECTED = 1, 2, 4          # stair_plan_run_flags / stair_plan_backward flags (include/stair_hip.h)
PLAN_EXT_PROJECTION = 4                                              # stair_plan_build* flag


class L2Normalize(nn.Module):
    """module_net.py:211-216."""

    def forward(self, feat):
        return ops.l2normalize(feat)


class _Node(nn.Module):
    """Parameter container reproducing the reference's attribute paths (``param.representation.0`` ...).
    Calling a node that holds ``weight``/``bias`` applies it as a Linear (used for pretrain heads)."""

    def __getitem__(self, key):
        return self._modules[str(key)]

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class _TemporalNode(_Node):
    """TemporalModule.pretrain_head returns the relate output of the LAST call (modules.py:285-288)."""
    related_attn = None

    def pretrain_head(self, *args):
        return self.related_attn


def _init_param(name, p, config):
    H = config['hidden_size']
    with torch.no_grad():
        if name.endswith('layer_norm.weight'):
            p.fill_(1.0)
        elif name.endswith('layer_norm.bias'):
            p.zero_()
        elif name.endswith('Relate.beta'):
            p.uniform_(0.0, 1.0)                                   # torch.rand, modules.py:420
        elif '_encoder.' in name:
            b = 1.0 / math.sqrt(H // 2)
            p.uniform_(-b, b)                                      # nn.LSTM default
        else:
            table = dict(spec.weight_table(config))
            wshape = table[name[:-5] + '.weight'] if name.endswith('.bias') else tuple(p.shape)
            b = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            p.uniform_(-b, b)                                      # nn.Linear / nn.Conv1d default


def _clip_key(x):
    """Identity of a clip's feature buffer: same storage, offset and shape = same clip."""
    if isinstance(x, torch.Tensor):
        return ('t', x.untyped_storage().data_ptr(), x.storage_offset(), tuple(x.shape), tuple(x.stride()))
    if isinstance(x, np.ndarray):
        return ('n', x.__array_interface__['data'][0], x.shape, x.strides)
    return ('o', id(x))


class BatchResult:
    """Outputs of one batched pass plus read access to every intermediate program value.

    `logits` and `pred` are tensors of their own.  Everything else -- node(), related_attn(), token_feature,
    question_feature, grad_arena() -- is a VIEW into the model's shared workspace and is overwritten by the next
    run_programs / forward_batch / Trainer.step on the same model: clone what must outlive it (VideoNMN.forward does)."""

    def __init__(self, model, plan, info, ws, logits, pred, prog_off, programs, video=None, question=None):
        self._model, self._plan, self.info, self._ws = model, plan, info, ws
        self.logits, self.pred = logits, pred
        self._prog_off, self._programs = prog_off, programs
        self._video, self._question = video, question
        self.question_frames = None          # [n] frames per question when the batch mixes clip lengths, else None (all info.T)

    def status_word(self):
        """The plan's status word (int32 view of one workspace element): 0 = every pass on this workspace since run_programs
        completed normally; 1 = a cooperative recurrence timed out (stair_lstm_args.status) and the results hold NaN."""
        off = self.info.status_off
        return self._ws[off: off + 1].view(torch.int32)

    def check(self):
        """Synchronises and raises StairError when a pass of this batch reported a failure (stair_plan_status)."""
        check(lib.stair_plan_status(self._plan, C.c_void_p(self._ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def zero_grad_arenas(self):
        check(lib.stair_plan_zero_grads(self._plan, C.c_void_p(self._ws.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def grad_arena(self, kind):
        """View of a gradient arena ('vec' [n_vec,H], 'map' [n_map,T,H], 'att' [n_att,T]) of a training plan."""
        H, T, inf = self._model.config['hidden_size'], self.info.T, self.info
        if kind == 'vec':
            return self._ws[inf.gvec_off: inf.gvec_off + inf.n_vec * H].view(inf.n_vec, H)
        if kind == 'map':
            return self._ws[inf.gmap_off: inf.gmap_off + inf.n_map * T * H].view(inf.n_map, T, H)
        return self._ws[inf.gatt_off: inf.gatt_off + inf.n_att * T].view(inf.n_att, T)

    def backward(self, answers, loss_scale=1.0, keep_arenas=False, ready_event=None):
        """Reverse pass of a train=True run: decoder cross entropy against `answers` (int32 [n] on the GPU),
        gradients of loss_scale * sum_i CE_i accumulated into the model's gradient buffers.
        keep_arenas: the gradient arenas were zeroed by zero_grad_arenas() and hold injected loss gradients.
        ready_event: a torch.cuda.Event that has been recorded once (so that its hipEvent_t exists); the pass records it on the
        current stream when every gradient except the two encoders' is final (stair_plan_set_backward_event).
        Returns the unscaled per-question losses [n]."""
        ops._req(answers, 'answers', torch.int32)
        loss = torch.empty(self.info.n_questions, dtype=torch.float32, device=answers.device)
        self._model._bind_grads()
        flags = (1 if keep_arenas else 0) | (RUN_VIDEO_BF16 if self._video.dtype == torch.bfloat16 else 0)
        check(lib.stair_plan_set_backward_event(self._plan, C.c_void_p(ready_event.cuda_event if ready_event is not None else None)))
        check(lib.stair_plan_backward(self._model._ctx, self._plan, C.c_void_p(self._video.data_ptr()),
                                      C.c_void_p(self._question.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                      self._ws.numel() * 4, C.c_void_p(answers.data_ptr()), C.c_float(loss_scale),
                                      C.c_void_p(loss.data_ptr()), flags,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return loss

    def capture_graph(self):
        """Record this batch's forward pass (encoders, every program level, decoder, argmax) into a hipGraph and
        return a CapturedPlan whose replay() re-runs it with ONE launch -- BASELINE.json configs[3].  The graph
        reads self's video / question tensors and writes self.logits / self.pred in place: copy new inputs into
        those tensors (same programs, spans and lengths -- i.e. the same plan) and replay.  The plan gets a
        private workspace, because the model's shared one is rewritten by every other batch."""
        if self.info.gvec_off >= 0:
            raise StairError('capture_graph is for inference plans')
        return CapturedPlan(self)

    def touched(self):
        """bool per canonical weight: does this batch's program mix send a gradient into it?"""
        n = len(self._model._weight_names)
        arr = (C.c_int32 * n)()
        check(lib.stair_plan_touched(self._model._ctx, self._plan, arr, n))
        return [bool(v) for v in arr]

    def __del__(self):
        if getattr(self, '_plan', None):
            lib.stair_plan_destroy(self._plan)
            self._plan = None

    def _arena(self, off, rows, cols):
        return self._ws[off: off + rows * cols].view(rows, cols)

    def node_table(self):
        """(kind, slot, aux, level, rel_slot) of EVERY program token of the batch as int32 numpy arrays indexed by
        prog_off[q] + i -- one library call (stair_plan_nodes) instead of one per node."""
        tab = getattr(self, '_node_table', None)
        if tab is None:
            n = self.info.n_nodes
            tab = tuple(np.empty(n, dtype=np.int32) for _ in range(5))
            ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
            check(lib.stair_plan_nodes(self._plan, ip(tab[0]), ip(tab[1]), ip(tab[2]), ip(tab[3]), ip(tab[4]), n))
            self._node_table = tab
        return tab

    def node_info(self, q, i):
        k, s, a, l, r = (C.c_int32() for _ in range(5))
        check(lib.stair_plan_node(self._plan, int(self._prog_off[q] + i), C.byref(k), C.byref(s), C.byref(a),
                                  C.byref(l), C.byref(r)))
        return k.value, s.value, a.value, l.value, r.value

    def node(self, q, i):
        """Value of token i of question q: a str for keywords, else a tensor VIEW into the workspace."""
        H, T, inf = self._model.config['hidden_size'], self.info.T, self.info
        kind, slot, aux, _, _ = self.node_info(q, i)
        if kind == VAL_STR:
            return self._programs[q][i]
        if kind == VAL_VEC:
            return self._arena(inf.vec_off, inf.n_vec, H)[slot]
        if kind == VAL_MAP:
            return self._ws[inf.map_off: inf.map_off + inf.n_map * T * H].view(inf.n_map, T, H)[slot]
        if kind == VAL_ATT:
            return self._arena(inf.att_off, inf.n_att, T)[slot: slot + aux]
        if kind == VAL_FRAME:
            return self._arena(inf.att_off, inf.n_att, T)[slot]
        if kind == VAL_PAIR:
            v = self._arena(inf.vec_off, inf.n_vec, H)
            return torch.stack([v[slot], v[aux]])
        raise StairError('node has no value')

    def saved(self, q, i, which=0):
        """Training plans: view of the activation the backward pass keeps for token i of question q (stair_plan_saved_offset):
        [T,H] tiles of the tile MLPs, the [H] hidden row of Exists / ToAction; i = None: the decoder's hidden row [2H].
        None when the node keeps no such buffer."""
        H, T = self._model.config['hidden_size'], self.info.T
        tok = -1 - q if i is None else int(self._prog_off[q] + i)
        off = C.c_int64(-1)
        check(lib.stair_plan_saved_offset(self._plan, tok, which, C.byref(off)))
        if off.value < 0:
            return None
        if i is None:
            return self._ws[off.value: off.value + 2 * H]
        if self._programs[q][i] in ('Exists', 'ToAction'):
            return self._ws[off.value: off.value + H]
        return self._ws[off.value: off.value + T * H].view(T, H)

    def related_attn(self, q, i):
        _, _, _, _, rel = self.node_info(q, i)
        return self._arena(self.info.att_off, self.info.n_att, self.info.T)[rel] if rel >= 0 else None

    def levels(self, q):
        return [self.node_info(q, i)[3] for i in range(len(self._programs[q]))]

    @property
    def token_feature(self):
        return self._arena(self.info.tok_off, self.info.n_tok_rows, self._model.config['hidden_size'])

    @property
    def question_feature(self):
        return self._arena(self.info.qfeat_off, self.info.n_questions, self._model.config['hidden_size'])


class CapturedPlan:
    """A plan's forward pass as a replayable hipGraph (torch.cuda.CUDAGraph on ROCm).  Holds the BatchResult (plan,
    static input/output tensors) and its own workspace alive."""

    def __init__(self, res):
        self.result = res
        model, info = res._model, res.info
        dev = res._video.device
        self._ws = torch.empty((info.workspace_bytes + 3) // 4, dtype=torch.float32, device=dev)
        self._proj = None
        if getattr(res, '_proj', None) is not None:       # the plan keeps its projections outside the workspace: a private buffer too
            self._proj = torch.empty(res._proj.numel(), dtype=torch.float32, device=dev)     # (the replay recomputes them: no RUN_PROJECTED)
            check(lib.stair_plan_set_projection(model._ctx, res._plan, C.c_void_p(self._proj.data_ptr()), self._proj.numel()))
        self.logits, self.pred = res.logits, res.pred
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            check(lib.stair_plan_upload(res._plan, C.c_void_p(self._ws.data_ptr()), self._ws.numel() * 4,
                                        C.c_void_p(side.cuda_stream)))
            self._enqueue(side)                       # warm-up outside the capture (lazy module loading, attributes)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._enqueue(torch.cuda.current_stream(dev))

    def _enqueue(self, stream):
        res = self.result
        check(lib.stair_plan_run_flags(res._model._ctx, res._plan, C.c_void_p(res._video.data_ptr()),
                                       C.c_void_p(res._question.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                       self._ws.numel() * 4, C.c_void_p(self.logits.data_ptr()),
                                       C.c_void_p(self.pred.data_ptr()),
                                       RUN_INDEX_RESIDENT | (RUN_VIDEO_BF16 if res._video.dtype == torch.bfloat16 else 0),
                                       C.c_void_p(stream.cuda_stream)))

    def replay(self):
        self.graph.replay()
        return self.logits, self.pred


class VideoNMN(nn.Module):
    def __init__(self, config, debug=False, pretrain_modules=set()):
        super().__init__()
        self.debug = debug
        self.config = dict(config)
        self.pretrain_modules = pretrain_modules
        self.contrastive_head = L2Normalize()
        self.words_to_keep = set(spec.KEYWORDS)

        self.submodules = nn.ModuleDict()
        for name in spec.MODULE_NAMES:
            self.submodules[name] = _TemporalNode() if name == 'Temporal' else _Node()
        for name in ('video_encoder', 'text_encoder', 'decoder'):
            self.submodules[name] = _Node()
        for key in spec.state_dict_keys(self.config):
            parts = key.split('.')[1:]                  # drop 'submodules'
            node = self.submodules[parts[0]]
            for part in parts[1:-1]:
                if part not in node._modules:
                    if parts[0] == 'Superlative' and part == 'localize_module':
                        node.add_module(part, self.submodules['Localize'])      # module_net.py:31-32
                    else:
                        node.add_module(part, _Node())
                node = node._modules[part]
            if parts[-1] not in node._parameters:
                shape = dict(spec.weight_table(self.config))[key]
                p = nn.Parameter(torch.empty(*shape))
                _init_param(key, p, self.config)
                node.register_parameter(parts[-1], p)
        if self.config['have_pretrain_head']:
            for name in ('Filter', 'Superlative', 'ToAction'):
                self.submodules[name].pretrain_head = self.contrastive_head     # modules.py:110,234,363
            for name in ('Localize', 'HasItem', 'ExistsFrame'):
                self.submodules[name].pretrain_head = nn.Identity()

        cfg = StairConfig(self.config['hidden_size'], self.config['video_size'], self.config['text_size'],
                          self.config['answer_vocab_length'], self.config['max_video_length'],
                          self.config['object_types'], 1 if self.config['have_pretrain_head'] else 0)
        handle = C.c_void_p()
        check(lib.stair_ctx_create(C.byref(cfg), C.byref(handle)))
        self._ctx = handle
        self._weight_names = [lib.stair_weight_name(self._ctx, i).decode() for i in range(lib.stair_weight_count(self._ctx))]
        self._bound = {}
        self._gbound = {}
        self._ws = None
        self._proj = None
        # run_programs: enqueue the encoders' input projections before the plan is built (STAIR_EARLY_PROJECTION=0: inside the plan's pass)
        self.early_projection = os.environ.get('STAIR_EARLY_PROJECTION', '1') != '0'
        self._programs = frontend.ProgramCache()

    def __del__(self):
        ctx = self.__dict__.get('_ctx')
        if ctx:
            lib.stair_ctx_destroy(ctx)
            self.__dict__['_ctx'] = None        # plain dict write: nn.Module.__setattr__ may be gone at shutdown

    OPTIONS = {'matmul_mode': 0, 'tile_mlp': 1, 'tile_queue': 2, 'vec_group': 3, 'tn_slab_min_rows': 4}
    MATMUL_MODES = {'f32': 0, 'bf16x3': 1, 'bf16': 2}

    def set_option(self, name, value):
        """Per-context override of a process-wide library setting (stair_ctx_set_option): 'matmul_mode' ('f32' / 'bf16x3' / 'bf16'),
        'tile_mlp', 'tile_queue', 'vec_group' (0 / 1), 'tn_slab_min_rows'; None restores the process default.  In force for this
        model's forward and backward passes only: other models of the process keep their own settings."""
        if name == 'matmul_mode' and isinstance(value, str):
            value = self.MATMUL_MODES[value]
        check(lib.stair_ctx_set_option(self._ctx, self.OPTIONS[name], -1 if value is None else int(value)))

    # ---------------------------------------------------------------------------------------
    def _bind_weights(self):
        """(Re)hand the current parameter storage to the ctx; pointers are borrowed by the library."""
        sd = dict(self.named_parameters())
        for i, name in enumerate(self._weight_names):
            p = sd[name]
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('parameter %s must be a contiguous float32 tensor on the GPU; call model.cuda()' % name)
            if self._bound.get(name) != p.data_ptr():
                check(lib.stair_ctx_set_weight(self._ctx, i, C.c_void_p(p.data_ptr()), p.numel()))
                self._bound[name] = p.data_ptr()

    def _bind_grads(self):
        sd = dict(self.named_parameters())
        for i, name in enumerate(self._weight_names):
            g = sd[name].grad
            if g is None or not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
                raise RuntimeError('parameter %s has no contiguous float32 GPU .grad buffer; use stair_amd.train.Trainer '
                                   'or allocate p.grad = torch.zeros_like(p)' % name)
            if self._gbound.get(name) != g.data_ptr():
                check(lib.stair_ctx_set_grad(self._ctx, i, C.c_void_p(g.data_ptr()), g.numel()))
                self._gbound[name] = g.data_ptr()

    def _projection(self, nfloats, device):
        """The shared buffer of the encoders' input projections (same lifetime rules as the shared workspace: the next batch rewrites it)."""
        if self._proj is None or self._proj.numel() < nfloats or self._proj.device != device:
            self._proj = torch.empty(int(nfloats * 1.25), dtype=torch.float32, device=device)
        return self._proj

    def _workspace(self, nbytes, device):
        n = (nbytes + 3) // 4
        if self._ws is None or self._ws.numel() < n or self._ws.device != device:
            self._ws = torch.empty(int(n * 1.25), dtype=torch.float32, device=device)
        return self._ws

    # ---------------------------------------------------------------------------------------
    def run_programs(self, programs, spans, video, question, q_lens, train=False, video_index=None, dropout=None, video_len=None,
                     before_run=None, cse=True):
        """Batched pass.  programs: list of token lists; spans: list of {pos: (lo, hi)}; video
        [n,T,V] and question [sum(q_lens), E] float32 on the GPU.  Returns a BatchResult.

        dropout (optional, training plans only): (p, seed) applies nn.Dropout(p) at the reference's positions with
        masks derived from `seed` (stair_plan_set_dropout); None or p = 0 is the eval / dropout=0 arithmetic that every
        parity test pins.
        video_index (optional, [n] ints): question q asks about clip video[video_index[q]]; video is then
        [n_videos,T,V] and each clip is encoded once instead of once per question (module_net.py:74 encodes per
        question; the results are identical).
        video_len (optional, [n_videos] ints): clips of different frame counts in one batch -- clip v holds video_len[v]
        <= T frames at the front of video[v], ZERO (at least finite) padding behind -- data.pack_questions and forward_batch
        pad with zeros; the padded rows take part in the weight-gradient products with zero coefficients, so NaN / Inf there
        would poison dW_ih (dataset.py:137-143 keeps every clip's own length); each
        question is computed as the reference computes a clip of its own length (stair_plan_build_ragged).
        cse: share common subexpressions across the batch (include/stair_hip.h STAIR_PLAN_NO_CSE): a node that depends only on
        the clip, keyword strings and identical question spans is computed once and aliased by every other occurrence; the
        values are those of the expanded computation (module_net.py:100-106), gradients of all users accumulate.  Ignored when
        dropout is active: the reference draws one mask per question and occurrence, a shared node would be dropped once for all.
        before_run (optional): called with the BatchResult after the plan is built (node table, slots and offsets are
        known) and BEFORE the pass is enqueued -- host work that only needs the plan (the loss driver's index arrays)
        then overlaps the previous step's GPU work instead of sitting between this step's forward and backward."""
        n = len(programs)
        if video_index is not None:
            video_index = np.ascontiguousarray(np.asarray(video_index, dtype=np.int32))
            if video_index.shape != (n,):
                raise ValueError('video_index must have one entry per question')
            if video.dim() != 3 or n == 0 or video_index.min() < 0 or video_index.max() >= video.shape[0]:
                raise ValueError('video must be [n_videos, T, V] and video_index must point into it')
        elif video.dim() != 3 or video.shape[0] != n:
            raise ValueError('video must be [n, T, V]')
        # clip features: fp32, or the stored bf16 format of BASELINE.json configs[1] (video.dtype == torch.bfloat16): the
        # input projection of the video encoder then runs on the bf16 rows directly (two MFMA products per operand pair)
        ops._req(video, 'video', torch.bfloat16 if video.dtype == torch.bfloat16 else torch.float32)
        ops._req(question, 'question')
        if video.dtype == torch.bfloat16 and self.config['video_size'] % 32:
            raise ValueError('bf16 clip features need video_size % 32 == 0')
        T = video.shape[1]
        self._bind_weights()
        # The encoders' input projections need the inputs only: enqueue them BEFORE packing the programs and building the plan (3-4 ms
        # of host work at 2048 questions), so the device is busy meanwhile (stair_encoders_project; results bit-identical)
        proj = None
        if self.early_projection and n > 0 and video.shape[0] > 0 and question.shape[0] > 0:
            nf = int(lib.stair_projection_floats(self._ctx, video.shape[0], T, question.shape[0]))
            proj = self._projection(nf, video.device)
            check(lib.stair_encoders_project(self._ctx, C.c_void_p(video.data_ptr()), 1 if video.dtype == torch.bfloat16 else 0,
                                             video.shape[0], T, C.c_void_p(question.data_ptr()), question.shape[0],
                                             C.c_void_p(proj.data_ptr()), proj.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        cache = self._programs
        compiled = [cache.get(p, sp) for p, sp in zip(programs, spans)]       # packed once per distinct (program, spans)
        prog_off, tokens, lo, hi, q_off = frontend.pack_batch(compiled, q_lens)
        if question.shape[0] != int(q_off[-1]):
            raise ValueError('question rows (%d) != sum(q_lens) (%d)' % (question.shape[0], int(q_off[-1])))

        def ip(a):
            return a.ctypes.data_as(C.POINTER(C.c_int32))
        plan = C.c_void_p()
        # under dropout every occurrence of a node draws its own mask (module_net.py:100-106 in model.train()): nothing is shared
        drop_on = dropout is not None and dropout[0] > 0
        if video_len is not None:
            video_len = np.ascontiguousarray(np.asarray(video_len, dtype=np.int32))
            if video_len.shape != (video.shape[0],):
                raise ValueError('video_len must have one entry per clip')
            if bool((video_len == T).all()):
                video_len = None
        check(lib.stair_plan_build_ragged(self._ctx, n, ip(prog_off), ip(tokens), ip(lo), ip(hi), ip(q_off), video.shape[0],
                                          ip(video_index) if video_index is not None else None,
                                          ip(video_len) if video_len is not None else None, T,
                                          (1 if train else 0) | (0 if cse and not drop_on else 2) | (PLAN_EXT_PROJECTION if proj is not None else 0),
                                          C.byref(plan)))
        try:
            info = PlanInfo()
            check(lib.stair_plan_get_info(plan, C.byref(info)))
            if dropout is not None and dropout[0] > 0:
                if not train:
                    raise ValueError('dropout is a training-mode operation (train=True)')
                check(lib.stair_plan_set_dropout(plan, C.c_float(float(dropout[0])), C.c_uint64(int(dropout[1]) & (2 ** 64 - 1))))
            ws = self._workspace(info.workspace_bytes, video.device)
            if proj is not None:
                check(lib.stair_plan_set_projection(self._ctx, plan, C.c_void_p(proj.data_ptr()), proj.numel()))
            A = self.config['answer_vocab_length']
            logits = torch.empty(n, A, dtype=torch.float32, device=video.device)
            pred = torch.empty(n, dtype=torch.int32, device=video.device)
        except Exception:
            lib.stair_plan_destroy(plan)
            raise
        res = BatchResult(self, plan, info, ws, logits, pred, prog_off, programs, video, question)     # owns the plan from here on
        res._proj = proj
        if video_len is not None:       # frames of every question's clip (the loss driver masks its criteria with them)
            res.question_frames = video_len[video_index] if video_index is not None else video_len
        if before_run is not None:
            before_run(res)
        check(lib.stair_plan_run_flags(self._ctx, plan, C.c_void_p(video.data_ptr()), C.c_void_p(question.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), ws.numel() * 4, C.c_void_p(logits.data_ptr()),
                                       C.c_void_p(pred.data_ptr()),
                                       (RUN_VIDEO_BF16 if video.dtype == torch.bfloat16 else 0) | (RUN_PROJECTED if proj is not None else 0),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if train:       # a training plan keeps its logits inside the workspace for the backward pass; hand the caller a copy
            res.logits = ws[info.logits_off: info.logits_off + n * A].view(n, A).clone()    # (n x A floats) that survives the next step
        return res

    def forward_batch(self, batch, train=False, share_videos=True, dropout=None, cse=True):
        """batch: list of question dicts in the reference layout (dataset.py:191-233).  Clips may differ in their
        number of frames (they are padded to the longest of the batch and run with their own lengths, see run_programs);
        Linear-Temporal configurations (max_video_length <= 32) need every clip at max_video_length, like the reference.
        Tensors may live on the host; they are moved once, packed.

        Questions about the same clip share one encoder pass.  AGQADataset hands every question of a video the
        same `self.video_feats[video_id]` tensor (dataset.py:183), so clips are told apart by the memory their
        `video_features` occupy (or by a `video_id` entry, when every dict carries one)."""
        dev = next(self.parameters()).device
        clips, index = [], None
        if share_videos and len(batch) > 1:
            by_id = all('video_id' in d for d in batch)
            seen, index = {}, []
            for d in batch:
                key = d['video_id'] if by_id else _clip_key(d['video_features'])
                if key not in seen:
                    seen[key] = len(clips)
                    clips.append(d['video_features'])
                index.append(seen[key])
            if len(clips) == len(batch):
                index = None
        else:
            clips = [d['video_features'] for d in batch]
        clips = [torch.as_tensor(c) for c in clips]
        # clips stored as bf16 (data.load_clip_features(..., dtype='bf16')) stay bf16 on the device
        vdtype = torch.bfloat16 if all(c.dtype == torch.bfloat16 for c in clips) else torch.float32
        frames = [int(c.shape[0]) for c in clips]
        video_len = None
        if len(set(frames)) > 1:                   # mixed clip lengths: pad to the longest, keep every clip's own length
            video_len, Tm = frames, max(frames)
            clips = [torch.nn.functional.pad(c.to(vdtype), (0, 0, 0, Tm - c.shape[0])) for c in clips]
        video = torch.stack(clips).to(dev, vdtype).contiguous()
        qs = [torch.as_tensor(d['question']) for d in batch]
        question = torch.cat(qs).to(dev, torch.float32).contiguous()
        return self.run_programs([d['nmn_program_list'] for d in batch],
                                 [d['prog_str_to_question_tokens'] for d in batch], video, question,
                                 [q.shape[0] for q in qs], train=train, video_index=index, dropout=dropout, video_len=video_len, cse=cse)

    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, data, return_res_by_step=True, return_result_of_each_step=False, test_mode=False):
        """Single-question form with the reference's return dict (module_net.py:65-145)."""
        program_list, program_idx = data['nmn_program_list'], data['nmn_program_idx']
        res = self.forward_batch([data])
        heads = self.config['have_pretrain_head']

        new_sg_res_by_step = dict()
        if not test_mode:
            for key, value in data.get('sg_res_by_step', {}).items():      # module_net.py:78-89
                if isinstance(value, list) and len(value) and isinstance(value[0][1], torch.Tensor):
                    ans = []
                    for v_class_name, v_emb in value:
                        _, v_emb = self.encode_question_no_grad(v_emb)
                        ans.append((v_class_name, self.contrastive_head(v_emb)))
                    new_sg_res_by_step[key] = ans
                else:
                    new_sg_res_by_step[key] = value

        res_by_step, each = dict(), []
        arity = spec.ARITY
        if return_res_by_step or return_result_of_each_step:
            values = [res.node(0, i) for i in range(len(program_list))]
            values = [v.clone() if isinstance(v, torch.Tensor) else v for v in values]
            # children of each module token, in pop order (module_net.py:100-106)
            stack, children = [], [None] * len(program_list)
            for i in range(len(program_list) - 1, -1, -1):
                if program_list[i] in arity:
                    children[i] = [stack.pop() for _ in range(arity[program_list[i]])]
                stack.append(i)
            for i in range(len(program_list) - 1, -1, -1):
                prog = program_list[i]
                if prog in arity:
                    result = values[i]
                    if prog == 'Temporal':
                        self.submodules['Temporal'].related_attn = res.related_attn(0, i).clone()
                    if return_res_by_step and program_idx[i] is not None and prog in self.pretrain_modules and i != 0:
                        out = self.submodules[prog].pretrain_head(result) if heads else result
                        res_by_step[program_idx[i]] = (prog, out)
                    if return_result_of_each_step:
                        params = [values[c] for c in children[i]]
                        if heads and prog in self.pretrain_modules:
                            each.append((params, self.submodules[prog].pretrain_head(result)))
                        else:
                            each.append((params, result))
                elif return_result_of_each_step:
                    each.append(([], values[i]))

        ret = {'logits': res.logits[0], 'res_by_step': res_by_step}
        if return_result_of_each_step:
            ret['result_of_each_step'] = list(reversed(each))
        if not test_mode:
            ret['sg_res_by_step'] = new_sg_res_by_step
        return ret

    @torch.no_grad()
    def encode_question_no_grad(self, question):
        return self.encode_question(question)

    def _lstm_weights(self, enc):
        m = self.submodules[enc]
        return [getattr(m, n + sfx) for sfx in ('', '_reverse')
                for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]

    def encode_question(self, question):
        """module_net.py:151-158 -> (token_feature [Q,H], sentence feature [H])."""
        dev = next(self.parameters()).device
        x = torch.as_tensor(question).to(dev, torch.float32).contiguous()
        off = torch.tensor([0, x.shape[0]], dtype=torch.int32, device=dev)
        out, h_n = ops.lstm_bidir(x, off, x.shape[0], self._lstm_weights('text_encoder'))
        return out, h_n[0]

    @torch.no_grad()
    def encode_phrases(self, phrase_embeddings):
        """Batched form of evaluate.py:66-76 (get_kw_representations): every phrase [L_i, E] through the text
        encoder in one ragged pass, sentence feature -> contrastive_head.  Returns [C, H] L2-normalised reps."""
        dev = next(self.parameters()).device
        xs = [torch.as_tensor(p).to(dev, torch.float32).reshape(-1, self.config['text_size']) for p in phrase_embeddings]
        lens = [int(x.shape[0]) for x in xs]
        if not xs or min(lens) <= 0:
            raise ValueError('every phrase needs at least one word embedding')
        off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
        _, h_n = ops.lstm_bidir(torch.cat(xs).contiguous(), off, max(lens), self._lstm_weights('text_encoder'))
        return ops.l2normalize(h_n)

    def encode_video(self, video_feat):
        """module_net.py:160-163 -> [T,H]."""
        dev = next(self.parameters()).device
        x = torch.as_tensor(video_feat).to(dev, torch.float32).contiguous()
        off = torch.tensor([0, x.shape[0]], dtype=torch.int32, device=dev)
        out, _ = ops.lstm_bidir(x, off, x.shape[0], self._lstm_weights('video_encoder'))
        return out
