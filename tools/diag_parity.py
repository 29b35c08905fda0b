"""Diagnostic (not a test): where do GPU results, the oracle on THIS host's CPU
and the committed golden files (oracle on the build container's CPU) differ?"""
import os, sys, subprocess
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from oracle import project_layer_ref as O

print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-400", shell=True, capture_output=True, text=True).stdout)
print('torch cpu capability', torch.backends.cpu.get_cpu_capability(), 'threads', torch.get_num_threads())
GOLD = os.path.join(ROOT, 'tests', 'golden')
from tempo_amd import numerics
print('host matmul flavour:', numerics.probe(), 'n=4096:', numerics.probe(n=4096), 'n=600:', numerics.probe(n=600), 'n=500000:', numerics.probe(n=500000))

def bits(b):
    V = b.shape[-1]
    return (b.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)

for name in cases.ROOT_CASES:
    case = cases.build_root(name)
    L = O.ProjectLayer(case['image_size'], case['heatmap_size'])
    ref, _ = L(*case['args'])
    gold = np.load(os.path.join(GOLD, name + '.npz'))
    d = np.abs(ref.numpy() - gold['cubes'])
    print(name, 'oracle(here) vs golden: max', d.max(), 'ndiff', int((d != 0).sum()),
          'mask ndiff', int((bits(L.last_bounding).numpy().astype(np.uint32) != gold['mask']).sum()))
    if torch.cuda.is_available():
        import tempo_amd
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size']); layer.keep_mask = True
        with torch.no_grad():
            got, _ = layer(*cases.to_device(case, 'cuda:0')['args'])
        layer.mm_mode = 'fma'
        with torch.no_grad():
            gotf, _ = layer(*cases.to_device(case, 'cuda:0')['args'])
        gf = gotf.cpu().numpy()
        print('   gpu(fma) vs golden: max', np.abs(gf - gold['cubes']).max(), 'ndiff', int((gf != gold['cubes']).sum()))
        g = got.cpu().numpy()
        print('   gpu vs golden: max', np.abs(g - gold['cubes']).max(), 'ndiff', int((g != gold['cubes']).sum()),
              '| gpu vs oracle(here): max', np.abs(g - ref.numpy()).max(), 'ndiff', int((g != ref.numpy()).sum()),
              '| mask vs golden ndiff', int((layer.last_mask.cpu().numpy().astype(np.uint32) != gold['mask']).sum()))

# stage-by-stage on one camera set
from tempo_amd import synthetic as S
from tempo_amd.camera import pack_cameras
cfg = S.PANOPTIC
cams = S.ring_cameras(5, seed=3)
centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
table = pack_cameras(centers, scales, cameras, cfg['image_size'], 'auto')[0]
grid = O.compute_grid(cfg['space_size'], cfg['space_center'], cfg['cube_size'])
import ctypes
emul = ctypes.CDLL(os.path.join(ROOT, 'tests', 'emul', '_emul.so'))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
if torch.cuda.is_available():
    from tempo_amd import ops
    pix, g, inside = ops.project_points(grid.cuda(), table.cuda(), cfg['heatmap_size'], cfg['image_size'])
w, h = cfg['heatmap_size']
for v, cam in enumerate(cams):
    co = O.SimpleCameraTorch(cam, 'cpu')
    xc = co.world_to_camera(grid)
    xy = co.world_to_pixel(grid)
    n = grid.shape[0]
    epix = np.zeros((n, 2), np.float32); eins = np.zeros(n, np.int32); eg = np.zeros((n, 2), np.float32)
    rec = table[v].numpy().copy(); gn = grid.numpy().copy()
    f = ctypes.c_float
    emul.emul_project(P(rec), P(gn), ctypes.c_long(n), f(w), f(h), f(cfg['image_size'][0]), f(cfg['image_size'][1]), int(numerics.abi_flag('auto') != 0), P(epix), P(eins), P(eg))
    msg = f'view {v}: emul(cpu) vs oracle(here) pix ndiff {int((epix.view(np.int32) != xy.numpy().view(np.int32)).sum())}'
    if torch.cuda.is_available():
        msg += f' | gpu vs emul pix ndiff {int((pix[v].cpu().numpy().view(np.int32) != epix.view(np.int32)).sum())} grid ndiff {int((g[v].cpu().numpy().view(np.int32) != eg.view(np.int32)).sum())}'
    print(msg)
# matmul order probe on this host
X = torch.rand(100000, 3) * 8000 - 4000; R = torch.randn(3, 3)
out = (X @ R).numpy()
x, y, z = (X[:, i].numpy().astype(np.float64) for i in range(3))
for j in range(1):
    a, b, c = (float(R[k, j]) for k in range(3))
    f32 = np.float32
    chain = (f32(f32(f32(x * a) .astype(np.float64) + y * b).astype(np.float64) + z * c)).astype(np.float32)  # fma chain emulated in f64
    unf = (f32(x * a) + f32(y * b)).astype(np.float32); unf = (unf + f32(z * c)).astype(np.float32)
    print('mm probe col0: fma-chain mismatches', int((chain != out[:, 0]).sum()), 'unfused mismatches', int((unf != out[:, 0]).sum()))
import torch.nn.functional as F
im = torch.rand(1, 1, 128, 240); gg = (torch.rand(1, 1, 50000, 2) * 2.3 - 1.15).clamp(-1.1, 1.1)
r = F.grid_sample(im, gg, align_corners=False).view(-1)
print('grid_sample probe checksum', float(r.double().sum()))
