"""Does the heatmap row pitch matter to the root kernel (L2 channel spread of the texel rows)?
Same workload with the map width varied; kernel time per launch."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import synthetic as S

base = copy.deepcopy(S.PANOPTIC)
for w in (240, 241, 242, 243, 244, 248, 252, 256, 260):
    S.PANOPTIC['heatmap_size'] = [w, 128]
    sys.argv = ['x', '--no-cpu-baseline']
    a = bench.parse()
    wl = bench.RootWorkload(a, torch.device('cuda:0'), seed=0)
    for _ in range(5):
        wl.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        wl.repack()
        e0.record(); wl.kernel(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'W={w:4d} row pitch {w*64:6d} B  kernel {ts[len(ts)//2]:.4f} ms', flush=True)
    del wl
