import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import project_layer_ref as O
from tempo_amd import ops, synthetic as S
from tempo_amd.camera import pack_cameras
for cfg, cams in ((S.PANOPTIC, S.ring_cameras(5, seed=3)), (S.H36M, S.h36m_cameras())):
    centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
    table = pack_cameras(centers, scales, cameras, cfg['image_size'])[0]
    grid = O.compute_grid(cfg['space_size'], cfg['space_center'], cfg['cube_size'])
    pix, g, inside = ops.project_points(grid.cuda(), table.cuda(), cfg['heatmap_size'], cfg['image_size'])
    w, h = cfg['heatmap_size']
    for v, cam in enumerate(cams):
        xy = O.SimpleCameraTorch(cam, 'cpu').world_to_pixel(grid)
        width, height = centers[0][v] * 2
        trans = torch.as_tensor(O.get_affine_transform(centers[0][v], scales[0][v] / 200.0, 0, cfg['image_size']), dtype=torch.float)
        sg = torch.clamp(O._sample_coords(xy, trans, w, h, cfg['image_size'], width, height, 'cpu'), -1.1, 1.1)
        gg = g[v].cpu()
        bad = (gg.view(torch.int32) != sg.view(torch.int32)).any(1).nonzero().flatten()
        print('view', v, 'mismatch', len(bad))
        for i in bad[:5].tolist():
            print('   ', i, 'pix', xy[i].tolist(), 'gpu', gg[i].tolist(), 'ref', sg[i].tolist(), 'trans', trans.tolist() if i == bad[0] else '')
