#!/usr/bin/env python3
"""Benchmark of the NMN hot path (BASELINE.json metric: questions/sec on AGQA2-shaped inputs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode train|infer]

Default mode `train` = BASELINE.json configs[1] ("AGQA2 full train ... 1 MI355X"): one step = one optimizer
step over a window of B synthetic questions per GPU -- program encoding, plan build, encode_video,
encode_question, every program level, decoder, cross-entropy loss, the full backward pass (BPTT through both
bi-LSTMs included), one flat-bucket RCCL gradient all-reduce when N>1, Adam + LambdaLR.  `--mode infer` times
the forward path + argmax only (configs[3] without hipGraph).  Shapes: T=64 frames x V=2048 features, H=512,
A=172, programs drawn from the 8-form corpus of SURVEY.md Appendix B, inputs resident in HBM before the timed
region.  For N>1 the driver launches this file under torch.distributed.run; questions shard across ranks,
timing is barrier-bracketed and the max over ranks is reported.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (fp32 MFMA GEMM of the LSTM input projection), timed live with
                events on the launch stream: achieved TFLOP/s vs the 157.3 TFLOP/s fp32 matrix peak.
  roofline_hbm  whole-path algorithmic bytes (1.93 MB/question incl. weights/128) x q/s
                vs 8 TB/s -- reported separately, never blended.
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded
                sample of the same questions; also the checker for top-1 agreement.
"""
import argparse
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


def make_batch(config, B, T, seed, device):
    """B synthetic questions; programs/spans from the deterministic generator, tensors drawn on the GPU."""
    qs = [synth.make_question(config, seed, i, T=T, with_video=False) for i in range(B)]
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    video = torch.randn(B, T, config['video_size'], device=device, generator=g)
    q_lens = [q['question'].shape[0] for q in qs]
    question = torch.randn(sum(q_lens), config['text_size'], device=device, generator=g)
    return qs, video, question, q_lens


# Memory-side bytes per launch of the dominant kernel from the PMC passes of tools/pmc_dominant.py
# (profiles/r01_g_pmc_dominant_t256.json): 2 x FETCH_SIZE (the gfx950 correction for 16-B-per-lane reads) + WRITE_SIZE, KiB.
PMC_TRAFFIC_BYTES = {(131072, 1024, 2048): int((2 * 787070.0 + 524288.0) * 1024)}
PMC_TRAFFIC_NOTE = ('bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH doubled per the gfx950 '
                    'correction; Infinity-Cache hits are counted, so the excess over the 1.62 GB algorithmic A+W+C is W tiles '
                    're-read through L2 (8 MB of W per XCD > 4 MB L2; 3.73 GB with the 128 x 128 tile); null for shapes that were not profiled')


def time_dominant_kernel(model, B, T, device, iters=10):
    """HIP-event timing of the input-projection GEMM launch (M=B*T, N=4*Hh, K=V), same stream."""
    from stair_amd import ops
    H, V = model.config['hidden_size'], model.config['video_size']
    M, N, K = B * T, 2 * H, V
    x = torch.randn(M, K, device=device)
    w = model.submodules['video_encoder'].weight_ih_l0
    b = model.submodules['video_encoder'].bias_ih_l0
    out = torch.empty(M, 4 * H, device=device)
    run = lambda: ops.gemm_grouped(x, K, None, w, b, out, 4 * H, None, M, 1, N, K, lda=K, ldc=4 * H)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K


def cpu_baseline_train(config, weights, qs, video, question, q_lens, budget_s=15.0, max_q=256):
    """The reference's training loop shape on the CPU: batch-1 forward + CE + backward per question through the
    oracle (torch autograd), one Adam step per 32 questions (train_module.py:341-412)."""
    from oracle import nmn_oracle as O
    names = [n for n, _ in spec.weight_table(config)]
    w = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    opt = torch.optim.Adam([w[n] for n in names], lr=2e-4)
    off = np.concatenate([[0], np.cumsum(q_lens)])
    n = min(max_q, len(qs))
    vid, qst = video[:n].cpu(), question[:off[n]].cpu()
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


def cpu_baseline(config, weights, qs, video, question, q_lens, budget_s=15.0, max_q=512):
    from oracle import nmn_oracle as O
    w = O.to_torch(weights)
    off = np.concatenate([[0], np.cumsum(q_lens)])
    n = min(max_q, len(qs))
    vid = video[:n].cpu()
    qst = question[:off[n]].cpu()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=2048, help='questions per GPU per step (window size per rank)')
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mode', choices=['train', 'infer'], default='train')
    ap.add_argument('--no-extras', action='store_true', help='skip the supplementary inference figures (clean kernel profiles)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
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

    from stair_amd.module_net import VideoNMN      # raises if libstair_hip.so is missing
    config = dict(spec.DEFAULT_CONFIG)
    weights = synth.make_weights(config, 0)
    model = VideoNMN(config)
    model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
    model = model.to(device)

    B, T = args.batch, args.frames
    qs, video, question, q_lens = make_batch(config, B, T, seed=rank, device=device)
    programs = [q['nmn_program_list'] for q in qs]
    spans = [q['prog_str_to_question_tokens'] for q in qs]

    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=device)
    trainer = None
    if args.mode == 'train':
        from stair_amd.train import Trainer
        trainer = Trainer(model, world=world, rank=rank, dropout=args.dropout)

    def step():
        if trainer is not None:
            return trainer.step(programs, spans, video, question, q_lens, answers)[1]
        return model.run_programs(programs, spans, video, question, q_lens)

    res = None
    for _ in range(args.warmup):
        res = step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_q = B * args.steps * world
    qps = total_q / elapsed

    infer_qps = None
    if args.mode == 'train' and not args.no_extras:          # the forward-only rate in the same process, for reference
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for _ in range(3):
            r_inf = model.run_programs(programs, spans, video, question, q_lens)
        torch.cuda.synchronize()
        infer_qps = 3 * B / (time.perf_counter() - ti)
        res = r_inf
    # questions that share a clip encode it once.  Same B questions, 8 per clip (AGQA asks tens per video).
    shared_qps = None
    if B % 8 == 0 and not args.no_extras:
        vidx = [i // 8 for i in range(B)]
        vshared = video[:B // 8].contiguous()
        model.run_programs(programs, spans, vshared, question, q_lens, video_index=vidx)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for _ in range(3):
            model.run_programs(programs, spans, vshared, question, q_lens, video_index=vidx)
        torch.cuda.synchronize()
        shared_qps = 3 * B / (time.perf_counter() - ti)

    # BASELINE configs[1] names bf16: the single-product mode (top-1 identity only, never the headline) measured in the same
    # process -- inference first (same weights as r_inf, so the answers can be compared), then optimizer steps.
    bf16_mode = None
    if args.mode == 'train' and not args.no_extras:
        from stair_amd import ops as _ops2
        default_mode = _ops2.get_matmul_mode()
        if default_mode != 'bf16':
            _ops2.set_matmul_mode('bf16')
            rb = model.run_programs(programs, spans, video, question, q_lens)
            torch.cuda.synchronize()
            ti = time.perf_counter()
            for _ in range(3):
                rb = model.run_programs(programs, spans, video, question, q_lens)
            torch.cuda.synchronize()
            b_inf = 3 * B / (time.perf_counter() - ti)
            agree_b = float((rb.pred == r_inf.pred).float().mean())
            dlogit_b = float((rb.logits - r_inf.logits).abs().max())
            step()
            barrier()
            ti = time.perf_counter()
            for _ in range(3):
                step()
            barrier()
            b_train = 3 * B * world / (time.perf_counter() - ti)
            _ops2.set_matmul_mode(default_mode)
            bf16_mode = {'train_questions_per_s': round(b_train, 1), 'inference_questions_per_s_per_gpu': round(b_inf, 1),
                         'top1_agreement_vs_default_mode': round(agree_b, 4), 'max_abs_logit_diff_vs_default_mode': dlogit_b,
                         'note': 'STAIR_MATMUL=bf16: one bf16 MFMA product per operand pair, fp32 accumulate; outside the 1e-4 '
                                 'logit budget by design, so it is reported beside `value`, never as it'}

    # Host-fed pipeline (SURVEY 8d "a second figure including H2D"): the batch's features start in pinned host memory;
    # a copy stream stages batch i+1 into the other of two device buffers while batch i is computed.  Never `value`.
    h2d_qps = None
    if world == 1 and not args.no_extras:
        host_v = video.cpu().pin_memory()
        host_q = question.cpu().pin_memory()
        dv = [torch.empty_like(video), torch.empty_like(video)]
        dq = [torch.empty_like(question), torch.empty_like(question)]
        copy_stream = torch.cuda.Stream(device=device)
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        main_stream = torch.cuda.current_stream(device)

        def stage(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[i % 2])               # the compute that last read this buffer
                dv[i % 2].copy_(host_v, non_blocking=True)
                dq[i % 2].copy_(host_q, non_blocking=True)
                ready[i % 2].record(copy_stream)
        n_it = 4
        for e in done:
            e.record(main_stream)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        stage(0)
        for i in range(n_it):
            if i + 1 < n_it:
                stage(i + 1)
            main_stream.wait_event(ready[i % 2])
            model.run_programs(programs, spans, dv[i % 2], dq[i % 2], q_lens)
            done[i % 2].record(main_stream)
        torch.cuda.synchronize()
        h2d_qps = n_it * B / (time.perf_counter() - ti)
        del host_v, host_q, dv, dq

    if rank == 0:
        from stair_amd import ops as _ops
        mm = _ops.get_matmul_mode()
        split = mm != 'f32'
        nprod = 3 if mm == 'bf16x3' else 1
        gemm_ms, gemm_flop = time_dominant_kernel(model, B, T, device)
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12            # ALGORITHMIC flops (2MNK) per second
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        line = {
            'metric': ('questions/sec on AGQA2-shaped synthetic features, training step (forward + CE + backward + Adam)'
                       if args.mode == 'train' else
                       'questions/sec on AGQA2-shaped synthetic features (NMN forward: encode -> program -> decoder -> argmax)'),
            'value': round(qps, 1), 'unit': 'questions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'data': 'synthetic',
            'dtype': ('f32 storage/accumulate; products as bf16x3 split (hi*hi+hi*lo+lo*hi on bf16 MFMA, ~4e-6 rel. error)' if mm == 'bf16x3'
                      else 'f32 storage/accumulate; single bf16 MFMA product per operand pair (~2e-3 rel. error, top-1 identity only)' if mm == 'bf16' else 'f32'),
            'config': {'workload': ('AGQA2 full train (BASELINE.json configs[1]): I3D-like [T=%d,V=%d] features, H=512, A=172, 8 program '
                                    'forms, decoder CE loss, fp32, one Adam step per window' if args.mode == 'train' else
                                    'AGQA2-shaped inference, I3D-like [T=%d,V=%d] features, H=512, A=172, 8 program forms, fp32')
                                   % (T, config['video_size']),
                       'questions_per_gpu_per_step': B, 'mode': args.mode,
                       'parallelism': ('dp%d (questions sharded, one flat fp32 gradient all-reduce per step)' if args.mode == 'train'
                                       else 'dp%d (questions sharded, no collective)') % world},
            'roofline': {'bound': 'mfma', 'kernel': '%s (LSTM input projection, M=%d N=%d K=%d)' % (
                             'gemm_bf16x3_t256_kernel<0>' if split else 'gemm_f32_kernel', B * T, 2 * config['hidden_size'], config['video_size']),
                         'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4), 'traffic': PMC_TRAFFIC_BYTES.get((B * T, 2 * config['hidden_size'], config['video_size'])),
                         'traffic_note': PMC_TRAFFIC_NOTE, 'launch_ms': round(gemm_ms, 4),
                         'note': ('achieved = algorithmic 2MNK / launch time against the dense bf16 MFMA peak; the kernel executes 3 bf16 '
                                  'MFMAs per algorithmic product by design: executed %.0f TFLOP/s = %.3f of peak; the exact fp32-MFMA kernel '
                                  'peaks at 157.3' % (3 * achieved, 3 * achieved / peak)) if mm == 'bf16x3' else
                                 ('single bf16 product per pair' if mm == 'bf16' else 'exact fp32 MFMA')},
            'roofline_hbm': {'bound': 'hbm', 'scope': 'whole path, algorithmic bytes x q/s (per GPU)',
                             'achieved': round(ALGO_BYTES_PER_QUESTION * qps / world / 1e9, 2), 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': round(ALGO_BYTES_PER_QUESTION * qps / world / 1e9 / HBM_PEAK_GBS, 5)},
            'path_tflops': round(ALGO_FLOP_PER_QUESTION * (3.0 if args.mode == 'train' else 1.0) * qps / world / 1e12, 2),
        }
        if infer_qps is not None:
            line['inference_questions_per_s_per_gpu'] = round(infer_qps, 1)
        if h2d_qps is not None:
            line['inference_h2d_inclusive_questions_per_s_per_gpu'] = round(h2d_qps, 1)
        if shared_qps is not None:
            line['inference_8_questions_per_clip_questions_per_s_per_gpu'] = round(shared_qps, 1)
        if bf16_mode is not None:
            line['bf16_single_product_mode'] = bf16_mode
        if not args.no_cpu_baseline and world == 1:        # the CPU leg runs on rank 0 at N=1 only
            # the box gives one GPU a 16-core CPU share; more ATen threads than that only thrash
            ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
            torch.set_num_threads(ncores)
            if args.mode == 'train':
                tq, tn = cpu_baseline_train(config, weights, qs, video, question, q_lens)
                line['cpu_baseline'] = {'value': round(tq, 1), 'unit': 'questions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                        'sample': 'first %d questions: batch-1 oracle forward + CE + autograd backward, Adam every 32 '
                                                  '(the reference loop shape, train_module.py:341-412), ATen CPU' % tn}
                model.load_state_dict({k: torch.from_numpy(weights[k].copy()) for k in spec.state_dict_keys(config)})
                res = model.run_programs(programs, spans, video, question, q_lens)     # parity check below on the initial weights
            cpu_qps, n_done, preds, cpu_logits = cpu_baseline(config, weights, qs, video, question, q_lens,
                                                              budget_s=8.0 if args.mode == 'train' else 15.0)
            gpu_pred = res.pred[:n_done].cpu().tolist()
            agree = sum(int(a == b) for a, b in zip(gpu_pred, preds)) / max(1, n_done)
            maxdiff = float((res.logits[:n_done].cpu() - cpu_logits).abs().max())
            inf = {'value': round(cpu_qps, 1), 'unit': 'questions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                   'sample': 'first %d questions of the rank-0 batch through oracle/nmn_oracle.py (batch-1 forward, ATen CPU)' % n_done}
            if args.mode == 'train':
                line['cpu_baseline_inference'] = inf
            else:
                line['cpu_baseline'] = inf
            line['top1_agreement_vs_oracle'] = round(agree, 4)
            line['max_abs_logit_diff_vs_oracle'] = maxdiff
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
