#!/usr/bin/env python3
"""Run the same training step several times on fresh Trainers and list which parameter gradients differ between runs
(run-to-run determinism of stair_plan_backward).  argv: [questions] [repeats] [forms: paper|all] [supervised 0|1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import spec, synth, losses as L                      # noqa: E402
from stair_amd.module_net import VideoNMN                           # noqa: E402
from stair_amd.train import Trainer                                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
forms = synth.ALL_FORMS if (len(sys.argv) > 3 and sys.argv[3] == 'all') else synth.PAPER_FORMS
sup = len(sys.argv) > 4 and sys.argv[4] == '1'
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
qs = [synth.make_question(config, 0, i, T=64, forms=forms, with_video=False) for i in range(B)]
if sup:
    for q in qs:
        sg = synth.make_gold(config, 0, q, T=64)
        q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v) for k, v in sg.items()}
g = torch.Generator(device=dev).manual_seed(1)
nclips = max(1, B // 2)                                    # two questions per clip: shared clips and common subexpressions
video = torch.randn(nclips, 64, 2048, device=dev, generator=g).to(torch.bfloat16)
vidx = [i % nclips for i in range(B)]
q_lens = [q['question'].shape[0] for q in qs]
question = torch.randn(sum(q_lens), 300, device=dev, generator=g)
answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
progs = [q['nmn_program_list'] for q in qs]
spans = [q['prog_str_to_question_tokens'] for q in qs]
runs = []
for r in range(R):
    m = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    m = m.to(dev)
    tr = Trainer(m, dropout=0.0)
    _, res = tr.step(progs, spans, video, question, q_lens, answers, video_index=vidx, questions=qs if sup else None)
    torch.cuda.synchronize()
    runs.append(({n: p.grad.detach().clone() for n, p in m.named_parameters()}, tr.flat_p.clone()))
    if r == 0:
        print('plan: staging vec %d map %d att %d, aliased %d' % (res.info.n_vec_stage, res.info.n_map_stage, res.info.n_att_stage, res.info.n_aliased))
bad = []
for n in runs[0][0]:
    d = max(float((runs[0][0][n] - runs[r][0][n]).abs().max()) for r in range(1, R))
    if d != 0.0:
        bad.append((n, d, float(runs[0][0][n].abs().max())))
print('%d of %d gradient tensors differ between runs' % (len(bad), len(runs[0][0])))
for n, d, mx in bad:
    print('  %-60s max |diff| %.3g (max |g| %.3g)' % (n, d, mx))
print('weights after Adam equal:', all(torch.equal(runs[0][1], runs[r][1]) for r in range(1, R)))
