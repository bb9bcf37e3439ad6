"""One process per GPU: start N ranks of a script under torch.distributed.run and relay rank 0's result line.

The data-parallel path (the reference itself is single-process, /root/reference/train_module.py:282) runs
one process per GPU with RCCL between them.  A caller that asks for N GPUs without having been launched as a rank
(`python bench.py --gpus 8`) must not silently run one rank: spawn_ranks starts the N ranks as FRESH child processes --
before the parent has touched the GPU; a process that has initialised HIP must not replace or fork itself -- waits for
them, relays rank 0's JSON line and fails loudly when a child fails or the line does not report N ranks."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys


def launched_as_rank(env=None):
    """True inside a rank started by torch.distributed.run / torchrun (WORLD_SIZE and RANK in the environment)."""
    env = os.environ if env is None else env
    return 'WORLD_SIZE' in env and 'RANK' in env


def visible_gpu_count(env=None, topology='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs this process could use, WITHOUT a HIP / HSA call (a parent that is about to spawn ranks must stay clean):
    the kfd topology lists every node; GPU nodes have simd_count > 0.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES restrict the set (a comma list; an empty string hides every GPU)."""
    env = os.environ if env is None else env
    n = 0
    try:
        for node in sorted(os.listdir(topology)):
            try:
                with open(os.path.join(topology, node, 'properties')) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except OSError:
        n = 0
    for key in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if key in env:
            listed = [x for x in env[key].split(',') if x.strip() != '']
            n = min(n, len(listed))
    return n


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_command(script, argv, n, port=None, python=None):
    """The command line the driver itself uses for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
            '--master-addr', '127.0.0.1', '--master-port', str(port or free_port()), script] + list(argv)


def last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith('{') and line.endswith('}'):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def spawn_ranks(script, argv, n, env=None, timeout=None, expect_key='n_gpus'):
    """Run `script argv` as n ranks, return (exit code, parsed JSON line of rank 0 or None, captured stdout).
    exit code != 0 when torch.distributed.run failed, when no JSON line came back, or when line[expect_key] != n."""
    env = dict(os.environ if env is None else env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK', 'ROLE_RANK', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL between processes needs it on this driver
    proc = subprocess.run(rank_command(script, argv, n), env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout)
    line = last_json_line(proc.stdout or '')
    code = proc.returncode
    if code == 0 and line is None:
        print('spawn_ranks: the ranks printed no JSON line', file=sys.stderr)
        code = 3
    elif code == 0 and expect_key is not None and line.get(expect_key) != n:
        print('spawn_ranks: asked for %d ranks, the line reports %s = %r' % (n, expect_key, line.get(expect_key)), file=sys.stderr)
        code = 4
    return code, line, proc.stdout or ''
