#!/usr/bin/env python3
"""Benchmark of the NMN hot path (BASELINE.json metric: questions/sec on AGQA2-shaped inputs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode train|infer] [--features bf16|f32]
                    [--supervision] [--dropout P]

Default = BASELINE.json configs[1] ("AGQA2 full train, I3D rgb+flow [T=64,2048] feats, bf16, 1 MI355X"): one step = one
optimizer step over a window of B synthetic questions per GPU -- program packing, plan build, encode_video,
encode_question, every program level, decoder, cross-entropy loss, the full backward pass (BPTT through both bi-LSTMs
included), one flat-bucket RCCL all-reduce when N > 1, Adam + LambdaLR.  Clip features are STORED in bf16 (the input
format configs[1] names; the oracle that checks the answers is fed the same rounded values, which fp32 holds exactly);
everything downstream is fp32 storage with split-bf16 (hi/lo) MFMA products.  `--features f32` keeps fp32 clips,
`--supervision` adds the per-module intermediate losses of configs[4] to the timed step, `--mode infer` times the
forward path + argmax only.  Shapes: T=64 frames x V=2048, H=512, A=172, programs from the 8-form corpus of SURVEY.md
Appendix B, inputs resident in HBM before the timed region.  For N > 1 the driver launches this file under
torch.distributed.run; questions shard across ranks, timing is barrier-bracketed, the max over ranks is reported.
Rank 0 prints ONE JSON line.

Objects on the line besides the contract's fields:
  roofline      the dominant kernel -- the video bi-LSTM input projection, a plane GEMM (csrc/gemm_planes.hip) on the
                stored bf16 features with M = B*T, N = 8*Hh (both directions in one launch), K = V -- timed live with
                events on its launch stream: algorithmic 2MNK / launch time against the 2.5 PFLOP/s dense bf16 MFMA
                peak; `traffic` = HBM bytes per launch from the PMC passes under profiles/ (read from that file).
  roofline_hbm  whole-path algorithmic bytes x q/s vs 8 TB/s -- reported separately, never blended.
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the
                same questions, at one thread and at all cores of the GPU's CPU share; also the top-1 / logit checker.
  batch_sweep   train and inference rates at 32 / 128 / 512 / 2048 questions per GPU per step (configs[2] is 128).
  supervised_step  the configs[4] step (gold intermediates, all module losses) next to the decoder-only step.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stair_amd import spec, synth  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md, "HBM3E peak BW" (spec)
ALGO_BYTES_PER_QUESTION = 1.93e6   # SURVEY.md section 8(d), mean over the 8 forms, fp32, weights/128
ALGO_FLOP_PER_QUESTION = 0.86e9
def _newest(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, 'profiles', n)):
            return os.path.join(ROOT, 'profiles', n)
    return os.path.join(ROOT, 'profiles', names[-1])


PMC_FILE = _newest('r04_pmc_dominant.json', 'r03_pmc_dominant.json')     # written by tools/summarize_prof.py from the PMC passes
PMC_STEP_FILE = _newest('r04_pmc_whole_step.json')                        # tools/collect_pmc_step.sh: both counters over whole training steps


def make_batch(config, B, T, seed, device, features):
    """B synthetic questions; programs/spans from the deterministic generator, tensors drawn on the GPU."""
    qs = [synth.make_question(config, seed, i, T=T, with_video=False) for i in range(B)]
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    video = torch.randn(B, T, config['video_size'], device=device, generator=g)
    if features == 'bf16':
        video = video.to(torch.bfloat16)                     # the STORED format; every consumer sees exactly these values
    q_lens = [q['question'].shape[0] for q in qs]
    question = torch.randn(sum(q_lens), config['text_size'], device=device, generator=g)
    return qs, video, question, q_lens


def pmc_traffic(M, N, K):
    """HBM bytes per launch of the dominant kernel, read from the committed PMC summary of tools/pmc_planes.py."""
    try:
        rows = json.load(open(PMC_FILE))
    except Exception:
        return None, 'profiles/r03_pmc_dominant.json missing'
    vals = {}
    for r in rows:
        if 'gemm_planes' in r['kernel'] and r.get('shape') == [M, N, K]:
            vals[r['counter']] = r['mean_per_dispatch']
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        return None, 'shape not profiled'
    # rocprofv3 reports KiB.  FETCH_SIZE is DOUBLED: on gfx950 a 16-byte-per-lane streaming read (this kernel's LDS-DMA loads) is
    # tallied at half its bytes (MI355X_MICROARCH.md, HBM section) -- calibrated on this kernel with an N = 256 launch, where every A
    # panel has ONE column tile and is read exactly once: FETCH_SIZE 285 MB against 537 MB of A + 2 MB of W
    # (profiles/r04_pmc_planes_calibration.json).  WRITE_SIZE is exact.  The counter includes Infinity-Cache hits: at N = 2048 the
    # corrected 2.69 GB are A once (0.54 GB) + the 16.8 MB of W planes re-fetched by each XCD once per group of 4 row panels
    # (8 column tiles x 2 MB of planes exceed the 4 MB L2: 512 / 4 x 16.8 MB = 2.15 GB, served on-die, not from HBM).
    return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), \
        ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (%s, tools/collect_pmc_planes.sh); FETCH_SIZE x 2 per the gfx950 '
         'correction, calibrated in profiles/r04_pmc_planes_calibration.json; memory-side requests incl. Infinity-Cache hits: A once + W planes '
         're-fetched per 4 row panels and XCD' % os.path.relpath(PMC_FILE, ROOT))


def whole_step_traffic():
    """memory-side bytes per question of a whole training step (FETCH_SIZE + WRITE_SIZE summed over every kernel of 4 steps)"""
    try:
        d = json.load(open(PMC_STEP_FILE))
        return {'fetch_bytes_per_step_as_reported': d['fetch_bytes_per_step'], 'write_bytes_per_step': d['write_bytes_per_step'],
                'bytes_per_question_as_reported': d['bytes_per_question'],
                'bytes_per_question_reads_doubled': int((2 * d['fetch_bytes_per_step'] + d['write_bytes_per_step']) / d['questions_per_step']),
                'source': os.path.relpath(PMC_STEP_FILE, ROOT) + ' (tools/collect_pmc_step.sh; training step, 2048 questions)',
                'note': 'against 1.93 MB algorithmic for the FORWARD pass alone: saved activations, gate matrices, dZ regions and gradient '
                        'arenas of the training step make up the difference; wide reads are tallied at half their bytes on gfx950, so the '
                        'truth lies between the two figures'}
    except Exception:
        return None


def time_dominant_kernel(model, B, T, device, features, iters=10):
    """HIP-event timing of the input-projection launch on its own stream.  bf16 features: ONE plane GEMM for both
    directions (M = B*T, N = 8*Hh, K = V).  fp32 features: the register-staged split kernel, one direction (N = 4*Hh)."""
    from stair_amd import ops
    H, V = model.config['hidden_size'], model.config['video_size']
    enc = model.submodules['video_encoder']
    M, K = B * T, V
    if features == 'bf16':
        N = 4 * H
        x = torch.randn(M, K, device=device).to(torch.bfloat16)
        w = torch.cat([enc.weight_ih_l0, enc.weight_ih_l0_reverse]).contiguous()
        wh, wl = ops.split_planes_tiled(w)
        b = torch.cat([enc.bias_ih_l0, enc.bias_ih_l0_reverse]).contiguous()
        out = torch.empty(M, N, device=device)
        run = lambda: ops.gemm_planes(x, None, wh, wl, b, out=out)
        name = 'gemm_planes_kernel<1,0,tiled> (video bi-LSTM input projection on stored bf16 features, both directions)'
    else:
        N = 2 * H
        x = torch.randn(M, K, device=device)
        out = torch.empty(M, 4 * H, device=device)
        run = lambda: ops.gemm_grouped(x, K, None, enc.weight_ih_l0, enc.bias_ih_l0, out, 4 * H, None, M, 1, N, K, lda=K, ldc=4 * H)
        name = 'gemm_bf16x3_t256_kernel<0> (video bi-LSTM input projection on fp32 features, one direction)'
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (M, N, K), name


def _oracle_inputs(qs, video, question, q_lens, n):
    off = np.concatenate([[0], np.cumsum(q_lens)])
    return video[:n].float().cpu(), question[:off[n]].cpu(), off        # bf16 clips widen exactly


def cpu_baseline_train(config, weights, qs, video, question, q_lens, budget_s=12.0, max_q=256):
    """The reference's training loop shape on the CPU: batch-1 forward + CE + backward per question through the
    oracle (torch autograd), one Adam step per 32 questions (train_module.py:341-412)."""
    from oracle import nmn_oracle as O
    names = [n for n, _ in spec.weight_table(config)]
    w = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    opt = torch.optim.Adam([w[n] for n in names], lr=2e-4)
    n = min(max_q, len(qs))
    vid, qst, off = _oracle_inputs(qs, video, question, q_lens, n)
    t0 = time.perf_counter()
    done = 0
    for i in range(n):
        d = dict(qs[i], video_features=vid[i], question=qst[off[i]:off[i + 1]])
        logits = O.forward(w, config, d, return_res_by_step=False)['logits']
        loss = torch.nn.functional.cross_entropy(logits.unsqueeze(0), torch.tensor([qs[i]['answer']])) / 32
        loss.backward()
        done += 1
        if done % 32 == 0:
            opt.step(); opt.zero_grad(set_to_none=False)
        if time.perf_counter() - t0 > budget_s:
            break
    return done / (time.perf_counter() - t0), done


def cpu_baseline(config, weights, qs, video, question, q_lens, budget_s=8.0, max_q=512):
    from oracle import nmn_oracle as O
    w = O.to_torch(weights)
    n = min(max_q, len(qs))
    vid, qst, off = _oracle_inputs(qs, video, question, q_lens, n)
    preds, logits = [], []
    t0 = time.perf_counter()
    done = 0
    with torch.no_grad():
        for i in range(n):
            d = dict(qs[i], video_features=vid[i], question=qst[off[i]:off[i + 1]])
            r = O.forward(w, config, d, return_res_by_step=False)
            preds.append(int(torch.argmax(r['logits'])))
            logits.append(r['logits'])
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    return done / dt, done, preds, torch.stack(logits)


GEMM_FAMILIES = {       # accounting key (csrc STAIR_ACCT_MFMA) -> substring of the kernel's name
    'tile_mlp': 'tile_mlp_kernel', 'gemm_bf16x3_t256': 'gemm_bf16x3_t256_kernel', 'gemm_bf16x3_w8': 'gemm_bf16x3_w8_kernel',
    'gemm_bf16x3': 'gemm_bf16x3_kernel', 'gemm_bf16x3_splitk': None, 'gemm_tn_bf16x3_t256': 'gemm_tn_bf16x3_t256_kernel',
    'gemm_tn_bf16x3': 'gemm_tn_bf16x3_kernel', 'gemm_planes': 'gemm_planes_kernel', 'gemm_tn_tr': 'gemm_tn_tr_kernel'}


def gemm_family_rooflines(step, device):
    """Per MFMA-kernel family of ONE training step: algorithmic flops (2MNK, from the launchers' accounting) over the kernel
    time of the same step (torch.profiler's device trace): time-weighted algorithmic TFLOP/s against the dense bf16 peak.
    `module_level` = every product except the video encoder's input projection and its weight gradient -- the family the
    step spends most of its time in.  Returns None when the device trace is unavailable."""
    from stair_amd import ops
    try:
        from torch.profiler import ProfilerActivity, profile
        step()
        torch.cuda.synchronize()
        with ops.kernel_accounting() as acct:
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                step()
                torch.cuda.synchronize()
        times = {}
        for ev in prof.key_averages():
            us = getattr(ev, 'device_time_total', None)
            if us is None:
                us = getattr(ev, 'cuda_time_total', 0.0)
            times[ev.key] = times.get(ev.key, 0.0) + float(us)
    except Exception as e:          # no device tracing on this box: say so instead of guessing
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:120])}
    fam = {}
    for key, sub in GEMM_FAMILIES.items():
        if key not in acct.table:
            continue
        launches, nbytes, flops = acct.table[key]
        name = sub or GEMM_FAMILIES['gemm_bf16x3']
        us = sum(t for k, t in times.items() if name in k and (name != 'gemm_bf16x3_kernel' or 'tn' not in k))
        fam.setdefault(name, [0, 0.0, 0])
        fam[name][0] += launches; fam[name][2] += flops
        fam[name][1] = us
    out, mod_us, mod_fl, mod_n = {}, 0.0, 0, 0
    for name, (launches, us, flops) in fam.items():
        if us <= 0:
            continue
        out[name] = {'launches_per_step': launches, 'ms_per_step': round(us / 1e3, 3), 'algorithmic_TFLOPs': round(flops / us / 1e6, 1),
                     'frac_of_bf16_peak': round(flops / us / 1e6 / BF16_MFMA_PEAK_TFLOPS, 4)}
        if name not in ('gemm_planes_kernel', 'gemm_tn_tr_kernel'):
            mod_us += us; mod_fl += flops; mod_n += launches
    if mod_us > 0:
        out['module_level'] = {'bound': 'mfma', 'launches_per_step': mod_n, 'ms_per_step': round(mod_us / 1e3, 3),
                               'achieved': round(mod_fl / mod_us / 1e6, 1), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(mod_fl / mod_us / 1e6 / BF16_MFMA_PEAK_TFLOPS, 4),
                               'note': 'all MFMA products of the step except the video input projection and its weight gradient: fused tile operators, '
                                       'vector-level layers, text-encoder projection, weight-gradient reductions; algorithmic 2MNK / device time of one step '
                                       '(torch.profiler); three executed MFMAs per algorithmic product'}
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=2048, help='questions per GPU per step (window size per rank)')
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--mode', choices=['train', 'infer'], default='train')
    ap.add_argument('--features', choices=['bf16', 'f32'], default='bf16', help='storage format of the clip features (configs[1]: bf16)')
    ap.add_argument('--supervision', action='store_true', help='configs[4]: gold intermediates + every per-module loss inside the timed step')
    ap.add_argument('--dropout', type=float, default=0.0, help='nn.Dropout p of the training step (the reference trains with 0.25; parity is defined at 0)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--native-allreduce', action='store_true', help='N > 1: the gradient all-reduce through the C ABI (stair_allreduce_grads, RCCL from libstair_hip.so) instead of torch.distributed')
    ap.add_argument('--no-extras', action='store_true', help='skip the supplementary figures (clean kernel profiles)')
    args = ap.parse_args()

    from stair_amd import launch
    if args.gpus > 1 and not launch.launched_as_rank():
        # `python bench.py --gpus N` on its own: start the N ranks (fresh processes; this one has not touched the GPU -- the
        # count comes from the kfd topology and the *_VISIBLE_DEVICES lists, no HIP call) and relay rank 0's line.  Never run
        # fewer ranks than asked for.
        ndev = launch.visible_gpu_count()
        if ndev < args.gpus and os.environ.get('STAIR_DIST_BACKEND', 'nccl') == 'nccl':      # (a gloo rehearsal wraps the ranks around the visible cards)
            print('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, ndev), file=sys.stderr)
            sys.exit(2)
        code, line, out = launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        if line is not None:
            print(json.dumps(line), flush=True)
        else:
            sys.stderr.write(out[-4000:])
        sys.exit(code)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # STAIR_DIST_BACKEND=gloo lets several ranks rehearse the N>1 path on one card (ranks wrap around the visible devices);
    # the driver's runs use the default: nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('STAIR_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)                 # before the process group: RCCL binds the communicator to this device
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    device = torch.device('cuda', dev_index)

    from stair_amd import losses as L, ops
    from stair_amd.module_net import VideoNMN      # raises if libstair_hip.so is missing
    config = dict(spec.DEFAULT_CONFIG)
    weights = synth.make_weights(config, 0)
    model = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
    model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
    model = model.to(device)

    B, T = args.batch, args.frames
    qs, video, question, q_lens = make_batch(config, B, T, seed=rank, device=device, features=args.features)
    programs = [q['nmn_program_list'] for q in qs]
    spans = [q['prog_str_to_question_tokens'] for q in qs]
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=device)
    off = np.concatenate([[0], np.cumsum(q_lens)])

    def gold_questions(sel):
        """the same questions with synthetic gold intermediates (sg_res_by_step in the layout of dataset.py:200-221) and
        their gold packs, i.e. what a data-loader worker prepares per question (losses.compile_gold)"""
        out = []
        for q in sel:
            q = dict(q)
            sg = synth.make_gold(config, 0, q, T=T)
            q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v) for k, v in sg.items()}
            L.compile_gold(q)
            out.append(q)
        return out

    trainer = None
    gold_qs = gold_questions(qs) if (args.supervision or not args.no_extras) and args.mode == 'train' else None
    # the contrastive classes of the whole job, one order on every rank: pools of a data-parallel step are then summed on the device
    class_table = L.ClassTable.from_questions(gold_qs, world) if gold_qs is not None else None
    # the loader's collate step: the batch's gold intermediates as flat arrays (losses.GoldBatch), built once per batch outside the
    # stepping process' critical path -- train_module.py's loop does this bookkeeping per question inside the step (:351-406)
    gold_batches = {}

    def gold_batch(nq):
        if nq not in gold_batches:
            gold_batches[nq] = L.collate_gold(gold_qs[:nq], class_table=class_table)
        return gold_batches[nq]
    if args.mode == 'train':
        from stair_amd.train import Trainer
        trainer = Trainer(model, world=world, rank=rank, dropout=args.dropout, native_allreduce=args.native_allreduce,
                          class_table=class_table)
    # The batch is a few hundred thousand long-lived Python objects (question dicts, gold packs).  A generation-2 pass of the
    # cyclic collector walks all of them -- 30-50 ms, i.e. two whole steps -- every few supervised steps; a training job's
    # data loader hands batches over from worker processes and never holds this many objects in the stepping process.
    gc.collect()
    gc.freeze()

    def run_step(nq=B, supervised=False):
        """one pass over the first nq questions of the rank's batch"""
        if trainer is not None:
            # every rank holds nq questions: the window size is known, so the step needs no size exchange (and no host sync)
            return trainer.step(programs[:nq], spans[:nq], video[:nq], question[:off[nq]], q_lens[:nq], answers[:nq],
                                global_batch=nq * world, questions=gold_batch(nq) if supervised else None)[1]
        return model.run_programs(programs[:nq], spans[:nq], video[:nq], question[:off[nq]], q_lens[:nq])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        barrier()
        return time.perf_counter() - t0, r

    if trainer is not None and world > 1:
        trainer.allreduce_events = []
        trainer.allreduce_late_events = []
    elapsed, res = timed(lambda: run_step(B, args.supervision), args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = B * args.steps * world / elapsed

    extras = {}
    if world > 1:
        # how many ranks the collective really spans: a sum of ones through the communicator (and ncclCommCount on the native one)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        extras['collective_ranks'] = int(probe.item())
        if backend == 'nccl':
            extras['rccl_ranks'] = extras['collective_ranks']
        extras['collective_backend'] = ('stair_allreduce_grads (RCCL from libstair_hip.so)' if args.native_allreduce else
                                        'torch.distributed %s' % backend)
        if trainer is not None:
            ev = trainer.allreduce_events[-args.steps:]
            trainer.allreduce_events = None
            ar = torch.tensor([sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))], device=device, dtype=torch.float64)
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)
            extras['allreduce_ms_per_step'] = round(float(ar.item()), 4)
            extras['allreduce_bytes_per_step'] = int(trainer.bucket.numel() * 4)

            def exposed_ms(events):
                """what the main stream sits through after its backward pass: from the start of the late collective
                ([encoder gradients | mask | status]) to the point where both pieces have arrived; max over ranks"""
                x = torch.tensor([sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))], device=device, dtype=torch.float64)
                dist.all_reduce(x, op=dist.ReduceOp.MAX)
                return round(float(x.item()), 4)
            torch.cuda.synchronize()
            extras['allreduce_exposed_ms_per_step'] = exposed_ms(trainer.allreduce_late_events[-args.steps:])
            extras['allreduce_overlap'] = ('two pieces: [module + decoder gradients] (%d bytes) on a side stream from the backward pass\'s '
                                           '"module gradients final" event on, beside BPTT and the encoders\' dW products; [encoder gradients | '
                                           'touched mask | status] (%d bytes) after the pass' % (trainer.split * 4, (trainer.bucket.numel() - trainer.split) * 4)
                                           if trainer._side is not None else 'off: one collective after the backward pass')
            trainer.allreduce_late_events = None
            if trainer.comm is not None:
                extras['rccl_ranks_native_comm'] = trainer.comm.ranks()[1]
        if trainer is not None:
            # BASELINE configs[2]: "AGQA2 train DP=8 over xGMI, RCCL grad all-reduce, global batch 1024" = 1024 / N questions per GPU
            # per step (128 at N = 8), the latency-bound regime; same step, same collectives, smaller window
            nq2 = max(1, min(B, 1024 // world))
            k2 = max(10, args.steps)
            trainer.allreduce_events, trainer.allreduce_late_events = [], []
            dt2, _ = timed(lambda: run_step(nq2, args.supervision), k2, 3)
            t2 = torch.tensor([dt2], device=device, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            ar2 = torch.tensor([sum(a.elapsed_time(b) for a, b in trainer.allreduce_events[-k2:]) / k2], device=device, dtype=torch.float64)
            dist.all_reduce(ar2, op=dist.ReduceOp.MAX)
            extras['configs2_global_1024'] = {
                'questions_per_gpu_per_step': nq2, 'global_batch': nq2 * world, 'collective_ranks': extras['collective_ranks'],
                'train_ms_per_step': round(float(t2.item()) / k2 * 1e3, 3), 'train_questions_per_s': round(nq2 * world * k2 / float(t2.item()), 1),
                'allreduce_ms_per_step': round(float(ar2.item()), 4),
                'allreduce_exposed_ms_per_step': exposed_ms(trainer.allreduce_late_events[-k2:]),
                'note': 'BASELINE configs[2] (global batch 1024 over N GPUs); the headline `value` is configs[1]\'s 2048 questions per GPU per step'}
            trainer.allreduce_events = trainer.allreduce_late_events = None
        if trainer is not None and not args.no_extras:
            # Supplementary data-parallel legs.  Every rank issues the SAME sequence of torch collectives whatever happens inside a
            # leg (a rank that fails locally records the error and still takes part), so a local failure cannot hang the job.
            if not args.supervision:
                # configs[4] under data parallelism: per-module losses, contrastive pools of the GLOBAL windows via the class table
                k_s, err = max(3, args.steps // 2), None
                try:
                    dt_s, _ = timed(lambda: run_step(B, True), k_s, 2)
                except Exception as e:
                    dt_s, err = float('inf'), '%s: %s' % (type(e).__name__, str(e)[:160])
                    barrier(); barrier()
                ts = torch.tensor([dt_s if err is None else 1e30], device=device, dtype=torch.float64)
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                extras['supervised_step_dp'] = ({'error': err} if err else {
                    'train_questions_per_s': round(B * k_s * world / float(ts.item()), 1), 'ms_per_step': round(float(ts.item()) / k_s * 1e3, 3),
                    'classes_in_table': len(class_table),
                    'note': 'BASELINE configs[4]: gradient all-reduce + one [windows, classes] presence all-reduce on the device per step; '
                            'no host-side collective'})
            if not args.native_allreduce and backend == 'nccl':
                # the same step with the collective through the C ABI (stair_allreduce_grads), and the two all-reduces compared bit for
                # bit on one bucket
                from stair_amd.comm import NativeComm
                comm, err, same_local = None, None, 0.0
                g = torch.Generator(device=device).manual_seed(77 + rank)
                a = torch.randn(trainer.bucket.numel(), device=device, generator=g)
                b = a.clone()
                dist.all_reduce(a)
                try:
                    comm = NativeComm(rank, world)
                    comm.allreduce_(b)
                    same_local = 1.0 if torch.equal(a, b) else 0.0
                except Exception as e:
                    err = '%s: %s' % (type(e).__name__, str(e)[:160])
                same = torch.tensor([same_local if err is None else -1.0], device=device)
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                del a, b
                k_n = max(3, args.steps // 2)
                if float(same.item()) >= 0.0:            # the native communicator works on every rank: time the step through it
                    trainer.comm = comm
                    dt_n, _ = timed(lambda: run_step(B, False), k_n, 2)
                    trainer.comm = None
                    tn = torch.tensor([dt_n], device=device, dtype=torch.float64)
                    dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                    extras['native_allreduce'] = {'train_questions_per_s': round(B * k_n * world / float(tn.item()), 1),
                                                  'ms_per_step': round(float(tn.item()) / k_n * 1e3, 3), 'rccl_ranks': comm.ranks()[1],
                                                  'bitwise_equal_to_torch_allreduce': bool(same.item() == 1.0)}
                else:
                    extras['native_allreduce'] = {'error': err or 'the native communicator failed on another rank'}
    if not args.no_extras and world == 1:
        # ---- forward-only rate, shared clips, host-fed pipeline (SURVEY 8d "a second figure including H2D") ----
        dt, r_inf = timed(lambda: model.run_programs(programs, spans, video, question, q_lens), 3, 1)
        extras['inference_questions_per_s_per_gpu'] = round(3 * B / dt, 1)
        if B % 8 == 0:
            vidx = [i // 8 for i in range(B)]
            vshared = video[:B // 8].contiguous()
            dt, _ = timed(lambda: model.run_programs(programs, spans, vshared, question, q_lens, video_index=vidx), 3, 1)
            extras['inference_8_questions_per_clip_questions_per_s_per_gpu'] = round(3 * B / dt, 1)
        host_v, host_q = video.cpu().pin_memory(), question.cpu().pin_memory()
        dv, dq = [torch.empty_like(video), torch.empty_like(video)], [torch.empty_like(question), torch.empty_like(question)]
        copy_stream = torch.cuda.Stream(device=device)
        ready, done = [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
        main_stream = torch.cuda.current_stream(device)

        def stage(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[i % 2])               # the compute that last read this buffer
                dv[i % 2].copy_(host_v, non_blocking=True)
                dq[i % 2].copy_(host_q, non_blocking=True)
                ready[i % 2].record(copy_stream)
        for e in done:
            e.record(main_stream)
        torch.cuda.synchronize()
        n_it = 4
        ti = time.perf_counter()
        stage(0)
        for i in range(n_it):
            if i + 1 < n_it:
                stage(i + 1)
            main_stream.wait_event(ready[i % 2])
            model.run_programs(programs, spans, dv[i % 2], dq[i % 2], q_lens)
            done[i % 2].record(main_stream)
        torch.cuda.synchronize()
        extras['inference_h2d_inclusive_questions_per_s_per_gpu'] = round(n_it * B / (time.perf_counter() - ti), 1)
        extras['inference_h2d_bytes_per_question'] = int(video[0].numel() * video.element_size())
        del host_v, host_q, dv, dq

        if args.mode == 'train':
            # ---- batch sweep: the regime BASELINE.json's other configs name (configs[2]: 128 questions per GPU per step) ----
            sweep = []
            for nq in (32, 128, 512, 2048):
                if nq > B:
                    continue
                k = 12 if nq <= 512 else 4
                dt_t, _ = timed(lambda: run_step(nq), k, 2)
                dt_i, _ = timed(lambda: model.run_programs(programs[:nq], spans[:nq], video[:nq], question[:off[nq]], q_lens[:nq]), k, 2)
                sweep.append({'questions_per_step': nq, 'train_ms_per_step': round(dt_t / k * 1e3, 3), 'train_questions_per_s': round(nq * k / dt_t, 1),
                              'infer_ms_per_batch': round(dt_i / k * 1e3, 3), 'infer_questions_per_s': round(nq * k / dt_i, 1)})
            extras['batch_sweep'] = sweep
            # ---- configs[4]: the step with per-module intermediate supervision, next to the decoder-only step ----
            if not args.supervision:
                def host_and_wall(fn, k, warm):
                    """(host time to ENQUEUE a step, wall time of a step): the first without waiting for the GPU"""
                    for _ in range(warm):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(k):
                        fn()
                    t1 = time.perf_counter()
                    torch.cuda.synchronize()
                    return (t1 - t0) / k, (time.perf_counter() - t0) / k
                h_s, dt_s = host_and_wall(lambda: run_step(B, True), 6, 3)
                h_p, dt_p = host_and_wall(lambda: run_step(B, False), 6, 3)
                n_gold = sum(len(q['sg_res_by_step']) for q in gold_qs)
                extras['supervised_step'] = {'train_questions_per_s': round(B / dt_s, 1), 'ms_per_step': round(dt_s * 1e3, 3),
                                             'host_enqueue_ms_per_step': round(h_s * 1e3, 3),
                                             'decoder_only_ms_per_step': round(dt_p * 1e3, 3), 'decoder_only_host_enqueue_ms_per_step': round(h_p * 1e3, 3),
                                             'supervised_nodes_per_step': n_gold,
                                             'note': 'BASELINE configs[4] on one GPU: gold intermediates on ~85 % of the supervisable nodes, attention / head / '
                                                     'contrastive (32-question windows, class table) criteria + decoder CE in one step; the batch\'s gold '
                                                     'intermediates arrive collated (losses.collate_gold), as a data loader hands them over'}
            # ---- the reference's training recipe runs nn.Dropout(0.25) (args.py:31; modules.py D positions); parity is defined at 0 ----
            if args.dropout == 0.0:
                trainer.dropout = 0.25
                dt_d, _ = timed(lambda: run_step(B, False), 6, 3)
                trainer.dropout = 0.0
                extras['dropout_0.25_step'] = {'train_questions_per_s': round(6 * B / dt_d, 1), 'ms_per_step': round(dt_d / 6 * 1e3, 3),
                                               'note': 'the recipe the reference trains with (args.py:31): counter-based masks (stair_plan_set_dropout) drawn inside the fused operators, backward without stored masks'}
            # ---- clips of their own lengths in one launch batch (dataset.py:137-143 keeps every clip's frame count): T uniform
            # in 16..64 (mean 40), padded to 64; the same questions, the frames past a clip's length are padding ----
            rng = np.random.RandomState(0)
            vlen = rng.randint(16, T + 1, size=B).astype(np.int32)
            vmask = (torch.arange(T, device=device)[None, :] < torch.as_tensor(vlen, device=device)[:, None]).unsqueeze(-1)
            vr = (video * vmask).contiguous()
            dt_rt, _ = timed(lambda: trainer.step(programs, spans, vr, question, q_lens, answers, video_len=vlen), 4, 2)
            dt_ri, _ = timed(lambda: model.run_programs(programs, spans, vr, question, q_lens, video_len=vlen), 4, 2)
            extras['ragged_clip_lengths'] = {'frames': 'uniform 16..%d (mean %.1f), one launch batch, padded to %d' % (T, float(vlen.mean()), T),
                                             'train_questions_per_s': round(4 * B / dt_rt, 1), 'train_ms_per_step': round(dt_rt / 4 * 1e3, 3),
                                             'infer_questions_per_s': round(4 * B / dt_ri, 1)}
            del vr
            # ---- the other storage / arithmetic modes, a few steps each ----
            other = 'f32' if args.features == 'bf16' else 'bf16'
            v2 = video.float() if other == 'f32' else video.to(torch.bfloat16)
            step2 = lambda: trainer.step(programs, spans, v2, question, q_lens, answers)
            dt2, _ = timed(step2, 3, 1)
            extras['%s_feature_storage' % other] = {'train_questions_per_s': round(3 * B / dt2, 1), 'ms_per_step': round(dt2 / 3 * 1e3, 3)}
            vf = video.float() if args.features == 'bf16' else video
            ops.set_matmul_mode('f32')                                   # exact fp32 MFMA everywhere (fp32 features: the plane GEMMs are split kernels)
            dtf, _ = timed(lambda: trainer.step(programs, spans, vf, question, q_lens, answers), 2, 1)
            ops.set_matmul_mode('bf16')                                  # one bf16 product per operand pair: top-1 identity only, never the headline
            dtb, _ = timed(lambda: trainer.step(programs, spans, vf, question, q_lens, answers), 3, 1)
            ops.set_matmul_mode('bf16x3')
            extras['exact_f32_mfma_mode'] = {'train_questions_per_s': round(2 * B / dtf, 1), 'ms_per_step': round(dtf / 2 * 1e3, 3),
                                             'note': 'STAIR_MATMUL=f32: v_mfma_f32_32x32x2_f32 everywhere, fp32 features'}
            extras['bf16_single_product_mode'] = {'train_questions_per_s': round(3 * B / dtb, 1), 'ms_per_step': round(dtb / 3 * 1e3, 3),
                                                  'note': 'STAIR_MATMUL=bf16: one bf16 MFMA product per operand pair, fp32 accumulate, fp32 features; outside the '
                                                          '1e-4 logit budget by design, so it is reported beside `value`, never as it'}
            # ---- per-kernel-family rooflines: LAST, because torch's profiler stays hooked into every launch of the process afterwards
            #      (the legs above would read ~1.3 ms more per step: r03 reported the supervised step that way) ----
            fam = gemm_family_rooflines(lambda: run_step(B, False), device)
            if fam:
                extras['roofline_gemm_families'] = fam
                t = fam.get('tile_mlp_kernel')
                if t:
                    # live time of the kernel: HIP events around each of its launches (stair_tile_timing), three steps, no profiler
                    try:
                        import ctypes as C
                        from stair_amd._lib import lib as _lib
                        from stair_amd import ops as _ops
                        run_step(B, False); torch.cuda.synchronize()
                        _lib.stair_tile_timing(1)
                        with _ops.kernel_accounting() as acct3:
                            for _ in range(3):
                                run_step(B, False)
                            torch.cuda.synchronize()
                        ms, nl = C.c_double(0.0), C.c_int32(0)
                        _lib.stair_tile_timing_read(C.byref(ms), C.byref(nl))
                        _lib.stair_tile_timing(0)
                        if nl.value > 0 and ms.value > 0 and 'tile_mlp' in acct3.table:
                            fl = acct3.table['tile_mlp'][2] / 3.0
                            t = dict(t, ms_per_step=round(ms.value / 3.0, 3), launches_per_step=nl.value // 3,
                                     algorithmic_TFLOPs=round(fl / (ms.value / 3.0) / 1e9, 1),
                                     frac_of_bf16_peak=round(fl / (ms.value / 3.0) / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4))
                            fam['tile_mlp_kernel'] = t
                    except Exception as e:      # keep the profiler's figure
                        t = dict(t, timing_error='%s: %s' % (type(e).__name__, str(e)[:80]))
                if t:       # by SUMMED time per step the fused tile operator (about ten launches) is the largest kernel: its own roofline entry
                    tile = {'bound': 'mfma', 'kernel': 'tile_mlp_kernel (fused map-level MLP chains, forward and backward, all launches of one step)',
                            'achieved': t['algorithmic_TFLOPs'], 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': t['frac_of_bf16_peak'],
                            'ms_per_step': t['ms_per_step'], 'traffic': None,
                            'note': 'algorithmic 2MNK of every tile layer of a step / device time of the kernel in the same steps (HIP events around each '
                                    'launch on its stream, three steps); three executed bf16 MFMAs per algorithmic product'}
                    try:
                        pm = json.load(open(os.path.join(ROOT, 'profiles', 'r03_pmc_tile_summary.json')))
                        tile['traffic'] = pm['per_tile']
                        tile['traffic_note'] = 'HBM-side bytes PER TILE of a two-layer launch from separate FETCH_SIZE / WRITE_SIZE passes (profiles/r03_pmc_tile_summary.json; reads doubled per the gfx950 correction): algorithmic 131 072 B in, 256 B out in inference, two 131 072 B saves in training'
                    except (OSError, ValueError, KeyError):
                        pass
                    extras['roofline_tile_operator'] = tile
            del v2, vf

    if rank == 0:
        gemm_ms, (Mg, Ng, Kg), kname = time_dominant_kernel(model, B, T, device, args.features)
        gemm_flop = 2.0 * Mg * Ng * Kg
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12            # ALGORITHMIC flops (2MNK) per second
        nprod = 2 if args.features == 'bf16' else 3
        traffic, tnote = pmc_traffic(Mg, Ng, Kg) if args.features == 'bf16' else (None, 'fp32-feature kernel: see profiles/r01_g_pmc_dominant_t256.json')
        mode_txt = ('AGQA2 full train (BASELINE.json configs[%s]): I3D-like [T=%d,V=%d] features stored in %s, H=512, A=172, 8 program forms, %s, '
                    'one Adam step per window' % ('4' if args.supervision else '1', T, config['video_size'], args.features,
                                                  'decoder CE + per-module intermediate losses' if args.supervision else 'decoder CE loss')
                    if args.mode == 'train' else
                    'AGQA2-shaped inference, I3D-like [T=%d,V=%d] features stored in %s, H=512, A=172, 8 program forms' % (T, config['video_size'], args.features))
        line = {
            'metric': ('questions/sec on AGQA2-shaped synthetic features, training step (forward + losses + backward + Adam)'
                       if args.mode == 'train' else
                       'questions/sec on AGQA2-shaped synthetic features (NMN forward: encode -> program -> decoder -> argmax)'),
            'value': round(qps, 1), 'unit': 'questions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'data': 'synthetic',
            'dtype': ('%s clip features; f32 storage/accumulate elsewhere; products on bf16 MFMA as split hi/lo pairs (~4e-6 rel. error): '
                      '2 per operand pair where an operand is the stored bf16 clip, else 3' % args.features),
            'config': {'workload': mode_txt, 'questions_per_gpu_per_step': B, 'mode': args.mode, 'dropout': args.dropout,
                       'encoder_projections_ahead_of_plan': bool(model.early_projection),
                       'parallelism': ('dp%d (questions sharded round-robin, one flat fp32 gradient all-reduce per step, touched mask in the same bucket)'
                                       if args.mode == 'train' else 'dp%d (questions sharded, no collective)') % world},
            'roofline': {'bound': 'mfma', 'kernel': '%s, M=%d N=%d K=%d' % (kname, Mg, Ng, Kg),
                         'achieved': round(achieved, 2), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_note': tnote,
                         'launch_ms': round(gemm_ms, 4),
                         'note': 'achieved = algorithmic 2MNK / launch time against the dense bf16 MFMA peak; the kernel executes %d bf16 MFMAs per '
                                 'algorithmic product by design (executed %.0f TFLOP/s = %.3f of peak)' % (nprod, nprod * achieved, nprod * achieved / BF16_MFMA_PEAK_TFLOPS)},
            'roofline_hbm': {'bound': 'hbm', 'scope': 'whole path, algorithmic bytes x q/s (per GPU)',
                             'achieved': round(ALGO_BYTES_PER_QUESTION * qps / world / 1e9, 2), 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': round(ALGO_BYTES_PER_QUESTION * qps / world / 1e9 / HBM_PEAK_GBS, 5),
                             'traffic': whole_step_traffic() if args.mode == 'train' else None},
            'path_tflops': round(ALGO_FLOP_PER_QUESTION * (3.0 if args.mode == 'train' else 1.0) * qps / world / 1e12, 2),
        }
        line.update(extras)
        # what the matrix pipes of THIS device sustain on random operands (back-to-back MFMAs from registers, no memory): the chip
        # holds its clock well under 2.4 GHz under such a load, so this -- not the datasheet's 2.5 PFLOP/s -- is the ceiling of a
        # kernel's EXECUTED flop rate here; `roofline.peak` stays the datasheet figure
        try:
            import ctypes as C
            from stair_amd._lib import lib as _lib, check as _check
            tf = C.c_double(0.0)
            _check(_lib.stair_mfma_probe(20000, 3, C.byref(tf), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ex = nprod * achieved
            line['roofline']['sustained_mfma_probe'] = {
                'TFLOP/s': round(tf.value, 1), 'frac_of_datasheet_peak': round(tf.value / BF16_MFMA_PEAK_TFLOPS, 4),
                'kernel_executed_over_sustained': round(ex / tf.value, 4),
                'note': 'stair_mfma_probe: v_mfma_f32_32x32x16_bf16 back to back from registers on every CU (2 waves per SIMD), random '
                        'operands, 3 launches of ~5 ms after a warm-up, same process and device as the line; the dominant kernel executes '
                        '%.0f TFLOP/s of MFMAs = %.2f of what the pipes sustain' % (ex, ex / tf.value)}
        except Exception as e:
            line['roofline']['sustained_mfma_probe'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:120])}
        if not args.no_cpu_baseline and world == 1:        # the CPU leg runs on rank 0 at N=1 only
            # the box gives one GPU a 16-core CPU share; more ATen threads than that only thrash
            ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
            model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
            res = model.run_programs(programs, spans, video, question, q_lens)     # parity check below on the initial weights
            per_threads = {}
            for nt in (1, ncores):
                torch.set_num_threads(nt)
                if args.mode == 'train':
                    tq, tn = cpu_baseline_train(config, weights, qs, video, question, q_lens, budget_s=8.0 if nt == 1 else 12.0)
                    per_threads[nt] = (tq, tn)
                iq, n_done, preds, cpu_logits = cpu_baseline(config, weights, qs, video, question, q_lens, budget_s=5.0 if nt == 1 else 8.0)
                per_threads[('inf', nt)] = (iq, n_done)
            sample_t = ('first %d questions: batch-1 oracle forward + CE + autograd backward, Adam every 32 (the reference loop shape, '
                        'train_module.py:341-412), ATen CPU, clips = the same bf16-rounded values')
            sample_i = 'first %d questions of the rank-0 batch through oracle/nmn_oracle.py (batch-1 forward, ATen CPU)'
            if args.mode == 'train':
                line['cpu_baseline'] = {'value': round(per_threads[ncores][0], 1), 'unit': 'questions/s', 'cores': ncores, 'kind': 'port',
                                        'sample': sample_t % per_threads[ncores][1], 'cpu_model': cpu_model(),
                                        'one_thread': {'value': round(per_threads[1][0], 1), 'cores': 1, 'sample': sample_t % per_threads[1][1]}}
                line['cpu_baseline_inference'] = {'value': round(per_threads[('inf', ncores)][0], 1), 'unit': 'questions/s', 'cores': ncores, 'kind': 'port',
                                                  'sample': sample_i % per_threads[('inf', ncores)][1],
                                                  'one_thread': {'value': round(per_threads[('inf', 1)][0], 1), 'cores': 1}}
            else:
                line['cpu_baseline'] = {'value': round(per_threads[('inf', ncores)][0], 1), 'unit': 'questions/s', 'cores': ncores, 'kind': 'port',
                                        'sample': sample_i % per_threads[('inf', ncores)][1], 'cpu_model': cpu_model(),
                                        'one_thread': {'value': round(per_threads[('inf', 1)][0], 1), 'cores': 1}}
            gpu_pred = res.pred[:n_done].cpu().tolist()
            line['top1_agreement_vs_oracle'] = round(sum(int(a == b) for a, b in zip(gpu_pred, preds)) / max(1, n_done), 4)
            line['max_abs_logit_diff_vs_oracle'] = float((res.logits[:n_done].cpu() - cpu_logits).abs().max())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
