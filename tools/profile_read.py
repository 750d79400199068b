"""Dev tool: where the host time of pipeline.read_frame / _pack goes (threads vs GIL)."""
import os, sys, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from dfu3d_amd import kitti_io, synth
from dfu3d_amd.params import NUSC_CLASSES
from dfu3d_amd.pipeline import read_frame
from dfu3d_amd.calibration import Calibration
root = "/dev/shm/dfu3d_prof"
shutil.rmtree(root, ignore_errors=True)
H, W, M = 900, 1600, 8
sc = synth.make_scene(0, H=H, W=W, M=M, cams=6, dense=True, device="cuda:0")
pts = sc.points.cpu().numpy()
img = np.zeros((H, W, 3), np.uint8)
for i in range(24):
    c = i % 6; n = int(sc.n_inst[c])
    kitti_io.write_frame(root, i, pts, sc.calibs[c], img, sc.masks[c][:n].cpu().numpy(), sc.inst_class[c][:n].cpu().numpy(),
                         np.full(n, 0.9, np.float32), sc.inst_box[c][:n].cpu().numpy(), NUSC_CLASSES, sc.depth[c].cpu().numpy())
stems = ["%06d" % i for i in range(24)]
def T(f, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
s = stems[0]
print("fromfile bin   %.2f ms" % T(lambda: np.fromfile(os.path.join(root, 'velodyne', s + '.bin'), dtype=np.float32)))
print("calib parse    %.2f ms" % T(lambda: Calibration(os.path.join(root, 'calib', s + '.txt'))))
print("calib record   %.2f ms" % T(lambda: Calibration(os.path.join(root, 'calib', s + '.txt')).record()))
print("np.load depth  %.2f ms" % T(lambda: np.load(os.path.join(root, 'depth_2', s + '.npy'))))
def npz():
    z = np.load(os.path.join(root, 'seg_2', s + '.npz')); return z['masks']
print("npz masks      %.2f ms" % T(npz))
print("read_frame     %.2f ms" % T(lambda: read_frame(root, s)))
for w in (1, 4, 8, 12, 16, 24):
    pool = ThreadPoolExecutor(w)
    print("pool %2d: 24 frames %.1f ms" % (w, T(lambda: list(pool.map(lambda q: read_frame(root, q), stems)))))
    pool.shutdown()
import ctypes
libc = ctypes.CDLL("libc.so.6")
print("mallopt", libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30))   # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD
for w in (1, 4, 8, 12, 16):
    pool = ThreadPoolExecutor(w)
    print("mallopt pool %2d: 24 frames %.1f ms" % (w, T(lambda: list(pool.map(lambda q: read_frame(root, q), stems)))))
    pool.shutdown()
frames = [read_frame(root, q) for q in stems]
depth = torch.zeros((24, H, W)).pin_memory(); masks = torch.zeros((24, M, H, W), dtype=torch.uint8).pin_memory()
def fill(i):
    f = frames[i]; depth[i] = torch.from_numpy(f.depth); m = f.masks.shape[0]
    masks[i, :m] = torch.from_numpy(f.masks)
def fill_np(i):
    f = frames[i]; np.copyto(depth[i].numpy(), f.depth); m = f.masks.shape[0]
    np.copyto(masks[i, :m].numpy(), f.masks)
print("fill serial    %.1f ms" % T(lambda: [fill(i) for i in range(24)]))
pool = ThreadPoolExecutor(12)
print("fill pool12    %.1f ms" % T(lambda: list(pool.map(fill, range(24)))))
print("fill_np pool12 %.1f ms" % T(lambda: list(pool.map(fill_np, range(24)))))
print("pin small x9   %.2f ms" % T(lambda: [torch.zeros((24, 8)).pin_memory() for _ in range(9)]))
t = torch.zeros((24, 8), dtype=torch.int32)
def assigns():
    for i in range(24):
        for j in range(8):
            t[i, j] = 1
print("192 scalar assigns %.2f ms" % T(assigns))
print("H2D depth+masks %.1f ms" % T(lambda: (depth.to("cuda:0", non_blocking=True), masks.to("cuda:0", non_blocking=True), torch.cuda.synchronize())))
shutil.rmtree(root, ignore_errors=True)
