import os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_dp as T
if __name__ == '__main__':
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        T._run(0, 1, True, False, os.path.join(d, 'solo.pt'))
        mp.spawn(T._worker, args=(2, T._free_port(), True, False, d, False), nprocs=2, join=True)
        solo = torch.load(os.path.join(d, 'solo.pt'))
        r0 = torch.load(os.path.join(d, 'rank0.pt'))
    from stair_amd.train import Trainer
    m = T._model(torch.device('cuda', 0))
    tr = Trainer(m, dropout=0.0)
    names = m._weight_names
    params = dict(m.named_parameters())
    for it in range(2):
        g, r = solo['grad%d' % it], r0['grad%d' % it]
        print('step', it, 'max|g|', float(g.abs().max()))
        for i, n in enumerate(names):
            o, k = tr.offsets[i], params[n].numel()
            dd = float((g[o:o + k] - r[o:o + k]).abs().max())
            if dd > 1e-6:
                print('   %-55s diff %.3g  |solo| %.3g |rank| %.3g' % (n, dd, float(g[o:o + k].abs().max()), float(r[o:o + k].abs().max())))
