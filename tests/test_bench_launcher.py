"""bench.py --gpus N starts N ranks itself (the reference's analogue is
tools/dist_train.sh:12-20).  --dry-run keeps the launcher, the rendezvous (gloo) and the
block timing protocol, and drops all device work, so this runs on a CPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks():
    line = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1', '--blocks', '2')
    assert line['n_gpus'] == 2 and line['config']['ranks_launched'] == 2
    assert line['steps'] == 3 and line['warmup'] == 1 and line['dry_run'] is True


def test_gpus_1_runs_in_process():
    line = _run('--dry-run', '--steps', '2', '--warmup', '0', '--blocks', '1')
    assert line['n_gpus'] == 1 and line['config']['ranks_launched'] == 1
