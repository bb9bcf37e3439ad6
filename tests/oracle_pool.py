"""Test infrastructure: the oracle's window gradient (autograd of oracle/nmn_oracle.py over many questions) computed by a pool
of CPU-only worker processes -- one batch-1 forward + backward per question, as /root/reference/train_module.py:341-380 runs
them, a question costs ~0.2 s on one core at full size, so 1 000+ questions need the box's cores side by side."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(args):
    path, lo, hi, n_total, seed, threads = args
    os.environ['HIP_VISIBLE_DEVICES'] = ''              # CPU only: the parent owns the GPU
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(threads)
    from oracle import nmn_oracle as O
    from stair_amd import spec, synth
    blob = torch.load(path, weights_only=False)          # written by this test run (questions with tensors)
    config, qs = blob['config'], blob['questions'][lo:hi]
    weights = synth.make_weights(config, seed)
    names = [n for n, _ in spec.weight_table(config)]
    w = {k: torch.from_numpy(weights[k].copy()).requires_grad_(True) for k in names}
    per_q = []
    for q in qs:
        lg = O.forward(w, config, dict(q, video_features=q['video_features'].float()), return_res_by_step=False)['logits']
        ce = torch.nn.functional.cross_entropy(lg.unsqueeze(0), torch.tensor([q['answer']]))
        per_q.append(float(ce.detach()))
        (ce / n_total).backward()
    return lo, per_q, {n: (w[n].grad if w[n].grad is not None else None) for n in names}


def window_gradients(config, seed, questions, workers=4, threads=4, tmp_dir=None):
    """(per-question CE, {name: gradient of mean CE or None}) over `questions` (dicts with CPU tensors).
    At most 4 workers: a GPU box lets 6 processes touch the card, and the children of a process that has initialised HIP are
    counted whatever they do (they are started with HIP_VISIBLE_DEVICES empty and never call the GPU)."""
    import tempfile

    import torch
    import torch.multiprocessing as mp
    n = len(questions)
    with tempfile.TemporaryDirectory(dir=tmp_dir) as d:
        path = os.path.join(d, 'window.pt')
        torch.save({'config': config, 'questions': questions}, path)
        step = (n + workers - 1) // workers
        jobs = [(path, lo, min(n, lo + step), n, seed, threads) for lo in range(0, n, step)]
        ctx = mp.get_context('spawn')
        saved = {k: os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES')}
        os.environ['HIP_VISIBLE_DEVICES'] = ''              # inherited by the workers at start-up (the parent's runtime is already up)
        try:
            pool = ctx.Pool(min(len(jobs), 4))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        with pool:
            parts = []
            for done, part in enumerate(pool.imap_unordered(_worker, jobs)):
                parts.append(part)
                print('  oracle pool: %d / %d chunks' % (done + 1, len(jobs)), flush=True)
    parts.sort(key=lambda p: p[0])
    per_q = [v for p in parts for v in p[1]]
    grads = {}
    for name in parts[0][2]:
        gs = [p[2][name] for p in parts if p[2][name] is not None]
        grads[name] = torch.stack(gs).sum(0) if gs else None
    return per_q, grads
