"""Profile driver: a few steps of one bench workload, nothing else (for rocprofv3)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='root')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--proposals', type=int, default=10)
a = ap.parse_args()
dev = torch.device('cuda:0')
wl = (bench.RootWorkload if a.workload == 'root' else bench.WindowWorkload)(a, dev, seed=0)
for _ in range(a.steps):
    wl.step()
torch.cuda.synchronize()
print('done', wl.name)
