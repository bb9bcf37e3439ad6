#!/usr/bin/env python3
"""cProfile of the host side of one training step (decoder-only and supervised): where the Python time goes."""
import cProfile, pstats, sys, os, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import spec, synth, losses as L
from stair_amd.module_net import VideoNMN
from stair_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
m = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
m = m.to(dev)
qs = [synth.make_question(config, 0, i, T=64, with_video=False) for i in range(B)]
for q in qs:
    sg = synth.make_gold(config, 0, q, T=64)
    q['sg_res_by_step'] = {k: ([(n, torch.from_numpy(np.asarray(e))) for n, e in v] if isinstance(v, list) else v) for k, v in sg.items()}
    L.compile_gold(q)
video = torch.randn(B, 64, 2048, device=dev).to(torch.bfloat16)
q_lens = [q['question'].shape[0] for q in qs]
question = torch.randn(sum(q_lens), 300, device=dev)
answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
tr = Trainer(m, dropout=0.0, class_table=L.ClassTable.from_questions(qs))
gold = L.collate_gold(qs, class_table=tr.class_table)          # the loader's collate step
import gc; gc.freeze()
for sup in (False, True):
    for _ in range(2):
        tr.step(progs, spans, video, question, q_lens, answers, questions=gold if sup else None)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        tr.step(progs, spans, video, question, q_lens, answers, questions=gold if sup else None)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
    print('==== supervised =', sup, ' (3 steps)'); print('\n'.join(s.getvalue().splitlines()[:64]))
