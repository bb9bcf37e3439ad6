"""stair_amd.launch: `bench.py --gpus N` without a launcher around it must start N ranks itself, relay rank 0's line and fail
loudly on anything less (CPU, gloo, world size 2)."""
import os
import subprocess
import sys

from stair_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PROBE = os.path.join(HERE, '_launch_probe.py')


def test_rank_detection_and_command_line():
    assert not launch.launched_as_rank({})
    assert not launch.launched_as_rank({'WORLD_SIZE': '2'})
    assert launch.launched_as_rank({'WORLD_SIZE': '2', 'RANK': '0'})
    cmd = launch.rank_command('bench.py', ['--gpus', '4', '--steps', '3'], 4, port=29555, python='python')
    # the form the driver itself uses for N > 1
    assert cmd == ['python', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
                   '--master-port', '29555', 'bench.py', '--gpus', '4', '--steps', '3']
    assert launch.last_json_line('x\n{"a": 1}\nnot json {\n') == {'a': 1}
    assert launch.last_json_line('nothing here') is None


def test_gpu_count_without_touching_the_gpu(tmp_path):
    """The pre-spawn probe reads the kfd topology (GPU nodes have simd_count > 0) and the *_VISIBLE_DEVICES lists; it makes no
    HIP call, so the parent stays clean for the ranks it starts."""
    for i, simd in enumerate([0, 0, 256, 256, 256]):          # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % (0 if simd else 64, simd))
    top = str(tmp_path)
    assert launch.visible_gpu_count({}, top) == 3
    assert launch.visible_gpu_count({'HIP_VISIBLE_DEVICES': '0,2'}, top) == 2
    assert launch.visible_gpu_count({'ROCR_VISIBLE_DEVICES': ''}, top) == 0
    assert launch.visible_gpu_count({}, str(tmp_path / 'missing')) == 0
    import inspect
    assert 'torch' not in inspect.getsource(launch.visible_gpu_count)


def test_two_ranks_are_started_and_the_line_is_relayed():
    env = dict(os.environ, RANK='5', WORLD_SIZE='9')             # stale rank variables of the caller must not leak into the children
    code, line, out = launch.spawn_ranks(PROBE, [], 2, env=env, timeout=300)
    assert code == 0, out
    assert line == {'n_gpus': 2, 'ranks_in_collective': 2}


def test_a_failing_rank_fails_the_launch():
    code, line, _ = launch.spawn_ranks(PROBE, ['--fail-rank', '1'], 2, timeout=300)
    assert code != 0


def test_fewer_ranks_than_asked_for_is_an_error():
    code, line, _ = launch.spawn_ranks(PROBE, ['--lie'], 2, timeout=300)
    assert code == 4 and line['n_gpus'] == 1
    code, line, _ = launch.spawn_ranks(PROBE, ['--silent'], 2, timeout=300)
    assert code == 3 and line is None


def test_bench_refuses_to_run_fewer_ranks_than_gpus():
    """No GPU in the CPU container: `bench.py --gpus 2` must exit non-zero instead of running one rank; a WORLD_SIZE that
    contradicts --gpus is refused as well (both before any GPU call)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env, capture_output=True, text=True, timeout=300)
    if 'GPU(s) visible' in r.stderr:            # the CPU container; on a multi-GPU box the launch itself would start
        assert r.returncode == 2 and r.stdout.strip() == ''
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                       env=dict(env, RANK='0', WORLD_SIZE='4', LOCAL_RANK='0'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE=4' in r.stderr
