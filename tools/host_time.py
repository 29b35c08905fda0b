"""Host (enqueue) time per bench step vs GPU time: is the step loop host-bound?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.argv = ['x', '--no-cpu-baseline']
a = bench.parse()
wl = bench.RootWorkload(a, torch.device('cuda:0'), seed=0)
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    wl.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3 * (t1 - t0) / 200:.4f} ms per step, total {1e3 * (t2 - t0) / 200:.4f} ms per step')
