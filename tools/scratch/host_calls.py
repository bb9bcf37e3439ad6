#!/usr/bin/env python3
"""Host time of every C-ABI call of the first steps of a fresh Trainer (which call is slow while things warm up)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stair_amd import spec, synth, _lib
from stair_amd.module_net import VideoNMN
from stair_amd.train import Trainer

acc = collections.OrderedDict()
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); acc[self.name] = acc.get(self.name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
class Proxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k):
        v = getattr(self._lib, k)
        return Timed(k, v) if callable(v) else v
import stair_amd.module_net as MN, stair_amd.train as TR
MN.lib = Proxy(_lib.lib); TR.lib = MN.lib

B = 2048
dev = torch.device('cuda', 0)
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 0)
qs = [synth.make_question(config, 0, i, T=64, forms=synth.PAPER_FORMS, with_video=False) for i in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(B, 64, 2048, device=dev, generator=g).to(torch.bfloat16)
q_lens = [q['question'].shape[0] for q in qs]
question = torch.randn(sum(q_lens), 300, device=dev, generator=g)
answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
m = VideoNMN(config); m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)}); m = m.to(dev)
tr = Trainer(m, dropout=0.0)
for i in range(14):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(progs, spans, video, question, q_lens, answers)
    host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    top = sorted(acc.items(), key=lambda kv: -kv[1])[:5]
    print('step %2d host %.2f ms: %s' % (i, host, ', '.join('%s %.2f' % kv for kv in top)), flush=True)
