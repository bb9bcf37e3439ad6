import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stair_amd import ops
dev = 'cuda:0'
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, N, K = 131072, 2048, 2048
x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev) / K ** 0.5
out = torch.empty(M, N, device=dev)
th, tl = ops.split_planes_tiled(w)
if os.environ.get('WR') == '1':
    wf = ops.pack_wfrag(w)
    t = timeit(lambda: ops.gemm_planes(x, None, wf, wf, out=out, w_frag_rows=N))
else:
    t = timeit(lambda: ops.gemm_planes(x, None, th, tl, out=out))
ref = (x[:512].float() @ w.t())
err = float((out[:512] - ref).abs().max())
ref2 = (x[-512:].float() @ w.t())
err = max(err, float((out[-512:] - ref2).abs().max()))
print('WR=%s STAGGER=%s  M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s algorithmic, %.0f executed; max err vs fp32 (512 rows) %.2e' % (
    os.environ.get('WR', '0'), os.environ.get('STAIR_PLANES_STAGGER', '1'), M, N, K, t, 2.0 * M * N * K / t / 1e9, 4.0 * M * N * K / t / 1e9, err))
