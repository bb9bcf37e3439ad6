#!/usr/bin/env python3
"""Wall time of each of the first N training steps of a fresh Trainer (2048 questions, one clip each): which steps pay for what."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stair_amd import spec, synth, losses as L
from stair_amd.module_net import VideoNMN
from stair_amd.train import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
qs = [synth.make_question(config, 0, i, T=64, forms=synth.PAPER_FORMS, with_video=False) for i in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(B, 64, 2048, device=dev, generator=g).to(torch.bfloat16)
q_lens = [q['question'].shape[0] for q in qs]
question = torch.randn(sum(q_lens), 300, device=dev, generator=g)
answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
progs = [q['nmn_program_list'] for q in qs]
spans = [q['prog_str_to_question_tokens'] for q in qs]
m = VideoNMN(config)
m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
m = m.to(dev)
tr = Trainer(m, dropout=0.0)
times, hosts = [], []
for i in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(progs, spans, video, question, q_lens, answers)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3); hosts.append((t1 - t0) * 1e3)
print('per-step wall ms (synchronised after every step):', ' '.join('%.2f' % t for t in times))
print('host enqueue ms:', ' '.join('%.2f' % t for t in hosts))
# back-to-back (no sync between steps), groups of 5
torch.cuda.synchronize()
for grp in range(6):
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(progs, spans, video, question, q_lens, answers)
    torch.cuda.synchronize()
    print('back-to-back group %d: %.3f ms per step' % (grp, (time.perf_counter() - t0) / 5 * 1e3))
