import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import torch, bench
from stair_amd import spec, synth
from stair_amd.module_net import VideoNMN
dev = torch.device('cuda:0')
config = dict(spec.DEFAULT_CONFIG)
model = VideoNMN(config).to(dev)
qs, video, question, q_lens = bench.make_batch(config, 2048, 64, seed=0, device=dev, features='bf16')
progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
for _ in range(3):
    r = model.run_programs(progs, spans, video, question, q_lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = model.run_programs(progs, spans, video, question, q_lens)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per batch %.2f ms; with final sync %.2f ms per batch' % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    r = model.run_programs(progs, spans, video, question, q_lens)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(s.getvalue()[:2600])
