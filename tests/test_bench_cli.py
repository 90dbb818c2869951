"""bench.py plumbing: the multi-GPU launch command (`--gpus N` on its own starts the ranks as a CHILD
`python -m torch.distributed.run`, before anything touches the GPU) and the argument contract of the driver."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_ranks_dry_run_prints_the_launcher_command():
    env = dict(os.environ, RANK='3', LOCAL_RANK='3')          # stale launcher variables must not leak into the child's environment
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '5', '--warmup', '2', '--dry-run'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    cmd = d['cmd']
    assert d['dry_run'] is True and d['cwd'] == ROOT
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '5', '--warmup', '2']            # same arguments, --dry-run dropped
    assert d['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'                              # dmabuf IPC: RCCL needs it on this pool


def test_world_size_mismatch_is_refused_before_any_gpu_call():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and b'WORLD_SIZE=2' in out.stderr


@pytest.mark.gpu
def test_one_rank_rehearsal_of_the_multi_gpu_bench_on_rccl():
    """The driver's N > 1 runs happen on hardware this repository never sees; this is their rehearsal on the 1-GPU box: bench.py under the
    driver's own launcher with ONE rank and MMSEG_BENCH_FORCE_DIST=1 goes through every line of the data-parallel branch on the real
    backend -- RCCL process group, weight broadcast, per-arena gradient all-reduces overlapped with the backward pass, barrier + max over
    ranks, the replica check -- and must print the contract's JSON line with the dp block."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MMSEG_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--size', '64', '--steps', '2', '--warmup', '1',
           '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['value'] > 0 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp1'
    assert len(d['per_rank_ms_per_step']) == 1
    dp = d['dp']
    assert dp['replicas_identical_after_timed_region'] is True
    assert dp['collectives_per_iteration'] > 0 and 0 < dp['overlapped_with_backward_per_iteration'] <= dp['collectives_per_iteration']
    assert d['roofline']['frac'] > 0
