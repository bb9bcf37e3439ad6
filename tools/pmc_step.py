#!/usr/bin/env python3
"""K training steps of the bench workload (bench.make_batch, BASELINE configs[1]: 2048 questions, T = 64, bf16 clips) and nothing
else, for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over the WHOLE step (tools/collect_pmc_step.sh sums the
counters of every stair:: kernel and divides by K).  argv: [steps] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from stair_amd import spec, synth                                  # noqa: E402
from stair_amd.module_net import VideoNMN                          # noqa: E402
from stair_amd.train import Trainer                                # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
model = VideoNMN(config)
model.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
model = model.to(dev)
qs, video, question, q_lens = bench.make_batch(config, B, 64, seed=0, device=dev, features='bf16')
programs = [q['nmn_program_list'] for q in qs]
spans = [q['prog_str_to_question_tokens'] for q in qs]
answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
tr = Trainer(model, dropout=0.0)
for _ in range(K):
    tr.step(programs, spans, video, question, q_lens, answers, global_batch=B)
torch.cuda.synchronize()
print('steps %d questions_per_step %d' % (K, B))
