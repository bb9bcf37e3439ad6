import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stair_amd import spec, synth, losses as L
from stair_amd.module_net import VideoNMN
from stair_amd.train import Trainer
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
B = 64
qs = [synth.make_question(config, 0, i, T=64, forms=synth.ALL_FORMS, with_video=False) for i in range(B)]
for q in qs:
    q['sg_res_by_step'] = synth.make_gold(config, 0, q, T=64)
g = torch.Generator(device=dev).manual_seed(1)
nclips = B // 2
video = torch.randn(nclips, 64, 2048, device=dev, generator=g).to(torch.bfloat16)
vidx = [i % nclips for i in range(B)]
q_lens = [q['question'].shape[0] for q in qs]
question = torch.randn(sum(q_lens), 300, device=dev, generator=g)
progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
runs = []
for r in range(3):
    m = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    m = m.to(dev)
    tr = Trainer(m, dropout=0.0)
    prep = {}
    def prepare(res_):
        prep['p'] = L.prepare_module_losses(m, res_, qs, window=32)
    res = m.run_programs(progs, spans, video, question, q_lens, train=True, video_index=vidx, before_run=prepare)
    res.zero_grad_arenas()
    tr.flat_g.zero_()
    losses, extra = L.launch_module_losses(m, res, prep['p'], 1.0 / B)
    torch.cuda.synchronize()
    runs.append({'vec': res.grad_arena('vec').clone(), 'att': res.grad_arena('att').clone(), 'map': res.grad_arena('map').clone(),
                 **{'loss_' + k: v.clone() for k, v in losses.items()},
                 **{'g_' + n: p.grad.clone() for n, p in m.named_parameters() if 'pretrain_head' in n}})
for k in runs[0]:
    d = max(float((runs[0][k] - runs[r][k]).abs().max()) for r in (1, 2))
    print('%-50s %s max diff %.3g' % (k, tuple(runs[0][k].shape), d))
