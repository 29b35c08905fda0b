"""Profile driver: a few steps of one bench workload, nothing else (for rocprofv3).
Takes bench.py's flags (--workload, --batch, --proposals, --timesteps, --calibrations,
--share, --group-max, --steps)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
if '--steps' not in sys.argv:
    sys.argv += ['--steps', '5']
a = bench.parse()
dev = torch.device('cuda:0')
wl = bench.make_workload(a, dev, seed=0)
for _ in range(a.steps):
    wl.step()
torch.cuda.synchronize()
print('done', wl.name)
