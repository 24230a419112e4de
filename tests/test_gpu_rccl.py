"""The RCCL code path on the 1-GPU box: every other multi-process test uses gloo, and `bench.py`'s `init_process_group('nccl')`
only runs under a launcher.  Both are exercised here with ONE rank, each in a fresh child process that initialises the process
group before anything else touches the GPU (reference DP semantics AE3D.py:46-48, 86-104)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    e.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0',
              'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    return e


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize('dtype', ['bf16'])
def test_training_step_through_rccl_all_reduce_is_bit_identical_to_the_plain_step(dtype):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_child.py'), dtype], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r['backend'] == 'nccl' and r['world'] == 1
    assert r['buckets'] >= 2 and r['launch_order'] == sorted(r['launch_order'])      # launched in backward order as they fill
    assert all(w != 'bool' for w in r['work_types']), r['work_types']                # real collective handles ...
    assert all(w == 'bool' for w in r['plain_work_types']), r['plain_work_types']    # ... and none in the plain step
    assert r['grads_bit_identical'] and r['weights_bit_identical'] and r['all_gather_ok']


def test_bench_under_the_launcher_with_one_rank_takes_the_rccl_path():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the driver's launch form for N > 1, with one rank.
    The line must come from the launcher path (process group on 'nccl', metric all-reduce executed)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', _env()['MASTER_PORT'], os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2',
           '--cpu-samples', '0', '--no-breakdown']
    e = dict(os.environ)
    e['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r['n_gpus'] == 1 and r['rccl_world_size'] == 1 and r['process_group'] == 'nccl'
    assert r['value'] > 0 and r['global_metrics']['samples'] == 256
    # training mode: gradient buckets through RCCL, with the overlap report the 8-GPU run will fill in
    cmd = cmd[:cmd.index('--steps')] + ['--steps', '3', '--warmup', '2', '--mode', 'train']
    cmd[cmd.index('--master-port') + 1] = _env()['MASTER_PORT']
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    g = r['gradient_all_reduce']
    assert r['rccl_world_size'] == 1 and g is not None and g['buckets'] >= 2 and g['all_reduce_ms_back_to_back'] > 0
    assert g['launch_order'] == sorted(g['launch_order'])
