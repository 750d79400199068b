"""Measurement of SURVEY.md §8 row f-3 on the GPU box: rotated BEV IoU matrix and rotated NMS at the sizes OpenPCDet's
post-processing uses (NMS_PRE_MAXSIZE 4096, tools/cfgs/.../centerpoint_nuscenes2kitti.yaml:82-93).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import stages as st
from dfu3d_amd.pcdet_kitti import iou3d_nms_utils as U
from oracle import iou3d_oracle as I

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
b = np.zeros((n, 7), np.float32)
b[:, :2] = rng.uniform(-50, 50, (n, 2)); b[:, 2] = -1
b[:, 3:6] = rng.uniform(0.6, 6.0, (n, 3)); b[:, 6] = rng.uniform(-3.2, 3.2, n)
s = rng.permutation(n).astype(np.float32)
tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
def timed(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
ms_iou = timed(lambda: st.boxes_bev(tb, tb, iou=True))
ms_nms = timed(lambda: U.nms_gpu(tb, ts, 0.2))
kept = int(U.nms_gpu(tb, ts, 0.2)[0].numel())
m = min(n, 512)
t0 = time.perf_counter(); I.boxes_bev(b[:m], b[:m]); t_iou = time.perf_counter() - t0
t0 = time.perf_counter(); I.nms(b[:m], s[:m], 0.2); t_nms = time.perf_counter() - t0
# pairs whose circumscribed circles are apart leave at once; the others clip a quadrilateral in LDS: VALU / LDS bound
print(json.dumps({"boxes": n, "iou_matrix_ms": round(ms_iou, 3), "iou_pairs_per_s": round(n * n / (ms_iou * 1e-3), 0),
                  "iou_matrix_write_GBs": round(4.0 * n * n / (ms_iou * 1e-3) / 1e9, 1),
                  "nms_ms": round(ms_nms, 3), "nms_kept": kept,
                  "cpu_baseline": {"iou_pairs_per_s": round(m * m / t_iou, 0), "nms_ms": round(t_nms * 1e3, 1), "cores": 1,
                                   "kind": "port", "sample": "%dx%d IoU matrix and NMS of %d boxes, oracle/iou3d_oracle (NumPy float64)" % (m, m, m)}}))
