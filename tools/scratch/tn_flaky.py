#!/usr/bin/env python3
"""Diagnostic: run-to-run agreement of the 128 x 128 TN kernel (gemm_tn_bf16x3_kernel) when two working workgroups share a CU.
Child mode (argv[1] == 'child'): repeats one product and reports how many repeats differ from the fp64 product by more than rounding."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child():
    import torch
    from stair_amd import ops
    dev = 'cuda:0'
    reps = int(os.environ.get('REPS', '150'))
    for (M, N, K, rs) in ((6700, 1024, 256, False), (6656, 1024, 256, False), (6704, 1024, 256, False), (6700, 1024, 256, True), (3001, 512, 512, False)):
        g = torch.Generator(device=dev).manual_seed(M)
        A = torch.randn(M, N, device=dev, generator=g)
        B = torch.randn(M, K, device=dev, generator=g)
        scale = torch.rand(M, device=dev, generator=g) if rs else None
        ref = (A.double().T @ (B.double() * (scale.double()[:, None] if rs else 1.0))).float()
        tol = 2e-5 * float(ref.abs().max())
        bad, worst, first = 0, 0.0, None
        outs = []
        for i in range(reps):
            Cm = torch.zeros(N, K, device=dev)
            ops.gemm_tn(A, B, Cm, M, N, K, row_scale=scale, rs_gstride=1 if rs else 0)
            outs.append(Cm)
        torch.cuda.synchronize()
        for i, Cm in enumerate(outs):
            d = (Cm - ref).abs()
            e = float(d.max())
            worst = max(worst, e)
            if e > tol:
                bad += 1
                if first is None:
                    idx = (d > tol).nonzero()
                    first = 'rep %d: %d elements off, n in [%d,%d], k in [%d,%d], max %.4g, n%%128 set %s' % (
                        i, idx.shape[0], int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, 1].min()), int(idx[:, 1].max()), e,
                        sorted(set((idx[:, 0] // 128).tolist()))[:8])
        print('  M=%d N=%d K=%d rs=%d: %d / %d repeats off (tol %.3g, worst %.4g)%s' % (M, N, K, rs, bad, reps, tol, worst, ('  ' + first) if first else ''), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
        sys.exit(0)
    for name, env in (('default build (no scale registers without a row scale)', {}),
                      ('old kernels (scale registers, 12 B scratch)', {'STAIR_TN_RS_ALWAYS': '1'}),
                      ('old kernels, one workgroup per CU', {'STAIR_TN_RS_ALWAYS': '1', 'STAIR_TN_LDS': '98304'}),
                      ('old kernels, 256 workgroups', {'STAIR_TN_RS_ALWAYS': '1', 'STAIR_TN_BLOCKS': '256'})):
        print(name, flush=True)
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=e, check=False, timeout=280)
