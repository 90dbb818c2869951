"""bench.py plumbing that needs no GPU: the multi-GPU launch command (`--gpus N` on its own starts the ranks as a CHILD
`python -m torch.distributed.run`, before anything touches the GPU) and the argument contract of the driver."""
import json
import os
import subprocess
import sys

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
