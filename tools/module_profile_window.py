"""cProfile of the drop-in ProjectLayerWithMask.forward the way tempo_pose.py:152-156 calls it
(one call per candidate and timestep, B frames each)."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases, tempo_amd
dev = torch.device('cuda:0')
for B in (1, 4):
    spec = dict(cfg='panoptic', B=B, sub_cube=[64, 64, 64], hm='noise', seed=2, invalid_every=0)
    case = cases.to_device(cases.build_window(None, full=spec), dev)
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    call = lambda: layer(*case['args'])
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
        print(f'B={B}: host {(t1 - t0) * 10:.3f} ms per call, with final sync {(t2 - t0) * 10:.3f} ms per call')
        if B == 1:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(100):
                call()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
