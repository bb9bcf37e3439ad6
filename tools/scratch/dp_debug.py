import os, sys, torch, tempfile, subprocess
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
mode = sys.argv[1]
if mode == 'child':
    import test_gpu_dp as T
    table = sys.argv[3] == '1'
    T._run(0, 1, True, False, sys.argv[2], table)
    sys.exit(0)
d = tempfile.mkdtemp()
outs = {}
for det in ('1', '0'):
    for table in ('0', '1'):
        p = os.path.join(d, 'o_%s_%s.pt' % (det, table))
        env = dict(os.environ, STAIR_DETERMINISTIC=det)
        subprocess.run([sys.executable, __file__, 'child', p, table], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs[(det, table)] = torch.load(p)
ref = outs[('0', '0')]
for k, v in outs.items():
    for it in range(2):
        g = v['grad%d' % it]; r = ref['grad%d' % it]
        print('det', k[0], 'table', k[1], 'step', it, 'max|g-ref|', float((g - r).abs().max()), 'max|g|', float(r.abs().max()), 'nonzero diff idx', (g - r).abs().gt(1e-6 * float(r.abs().max())).nonzero().flatten()[:5].tolist())
