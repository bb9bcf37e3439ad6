#!/usr/bin/env python3
"""Forward pass per batch: eager launches vs the captured hipGraph (BatchResult.capture_graph), by batch size."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import spec, synth
from stair_amd.module_net import VideoNMN

config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
model = VideoNMN(config)
model.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
model = model.to('cuda:0')
for B in (8, 32, 128, 512, 2048):
    qs = synth.make_questions(config, 0, B)
    progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
    video = torch.randn(B, 64, config['video_size'], device='cuda:0').to(torch.bfloat16)        # stored bf16 clips (BASELINE configs[1])
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to('cuda:0')
    q_lens = [q['question'].shape[0] for q in qs]
    run = lambda: model.run_programs(progs, spans, video, question, q_lens)
    for _ in range(8): res = run()          # warm-up: the page-locked index buffers of this batch size are allocated once (ms each)
    torch.cuda.synchronize()
    iters = 20
    t = time.perf_counter()
    for _ in range(iters): res = run()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / iters
    cap = res.capture_graph()
    cap.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters): cap.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / iters
    print('B=%5d launches=%4d  eager %8.3f ms (%9.0f q/s)   graph %8.3f ms (%9.0f q/s)   x%.2f' % (
        B, res.info.n_launches, eager * 1e3, B / eager, graph * 1e3, B / graph, eager / graph), flush=True)
