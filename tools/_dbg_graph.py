import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo')
from stair_amd import spec, synth
from stair_amd.module_net import VideoNMN
config = dict(spec.DEFAULT_CONFIG)
w = synth.make_weights(config, 4)
m = VideoNMN(config); m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)}); m = m.to('cuda:0')
qs = synth.make_questions(config, 31, 24, forms=synth.ALL_FORMS)
res = m.forward_batch(qs)
want = res.logits.clone()
cap = res.capture_graph()
torch.cuda.synchronize()
junk = [torch.randn(1 << 20, device='cuda:0') for _ in range(8)]
for i in range(4):
    cap.logits.zero_()
    lg, _ = cap.replay(); torch.cuda.synchronize()
    print('replay', i, 'max diff', float((lg - want).abs().max()), flush=True)
