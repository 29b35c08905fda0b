"""cProfile of the drop-in ProjectLayer.forward at B = 32 (host side of the module-level
figure in the bench line): where the python time goes, asynchronous launches."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, tempo_amd
args = bench.parse(["--workload", "root", "--no-cpu-baseline", "--batch", os.environ.get("B", "32")])
dev = torch.device('cuda:0')
wl = bench.RootWorkload(args, dev, 0)
cfg = wl.cfg
centers, scales, cameras = wl.meta
layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
layer.materialize_grids = False
call = lambda: layer(wl.fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                     centers, scales, cameras)
with torch.no_grad():
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host time per call {(t1 - t0) * 10:.3f} ms (async), with final sync {(t2 - t0) * 10:.3f} ms')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        call()
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
