"""bench.py's launch contract without a GPU: `python3 bench.py --gpus N` starts its own N ranks as a child
torch.distributed.run (the parent never touches the GPU), relays their exit code, and a rank without a GPU of its own
fails with a message that says so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_launches_its_own_ranks_and_fails_from_the_children():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this node has the GPUs: the launch would run the benchmark')
    env = dict(os.environ, OMP_NUM_THREADS='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert 'needs GPU index' in r.stderr and 'rank 1 of 2' in r.stderr          # the children's own message
    assert 'launch with torch.distributed.run' not in r.stderr                      # not the old refusal of the parent
    assert r.stdout.strip() == ''                                                    # no JSON line from a failed run


def test_bench_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'must agree' in r.stderr
