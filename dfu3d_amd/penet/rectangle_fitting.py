"""Mirror of tools/PENet/rectangle_fitting/rectangle_fitting.py: LShapeFitting
with the same `fitting(ox, oy) -> (rects, idsets)` contract
(rectangle_fitting.py:39-53), computed by dfu3d_range_cluster +
dfu3d_lshape_fit on the GPU.  The 15 s wall-clock abort of the reference
(rectangle_fitting.py:186-188, hazard H2) does not exist here."""
import numpy as np
import torch

from .. import stages as st
from ..labels import C_CLUSTER, C_RECT_C, C_THETA
from ..params import Params


class RectangleData:
    """rectangle_fitting.py:194-199: a, b, c of the four edge lines a*x+b*y=c."""

    def __init__(self):
        self.a = [None] * 4
        self.b = [None] * 4
        self.c = [None] * 4


class LShapeFitting:
    def __init__(self, params: Params = None, device="cuda:0"):
        p = params or Params()
        self.R0 = p.R0                      # rectangle_fitting.py:36
        self.Rd = p.Rd                      # rectangle_fitting.py:37
        self.dtheta_deg_for_serarch = p.dtheta_deg   # rectangle_fitting.py:35 (sic)
        self._p = p
        self._dev = torch.device(device)

    def fitting(self, ox, oy):
        ox = np.asarray(ox, np.float64).reshape(-1)
        oy = np.asarray(oy, np.float64).reshape(-1)
        n = ox.shape[0]
        if n == 0:
            return [], []
        d = self._dev
        cap = n + 8
        f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).to(d)
        px = torch.zeros(cap, dtype=torch.float64, device=d)
        py = torch.zeros(cap, dtype=torch.float64, device=d)
        px[:n], py[:n] = f64(ox), f64(oy)
        pz = torch.zeros(cap, dtype=torch.float64, device=d)
        base = torch.zeros(1, dtype=torch.int64, device=d)
        cnt = torch.full((1,), n, dtype=torch.int32, device=d)
        label = torch.zeros(cap, dtype=torch.int32, device=d)
        st.range_cluster(px, py, base, cnt, 1, self.R0, self.Rd, label, cap)
        n_theta, dtheta = Params(dtheta_deg=self.dtheta_deg_for_serarch).thetas()
        calib = torch.zeros(st.CALIB_FLOATS, dtype=torch.float32, device=d)
        rows = torch.zeros(n * st.ROW_DOUBLES, dtype=torch.float64, device=d)
        n_rows = torch.zeros(1, dtype=torch.int32, device=d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        z32 = lambda k: torch.zeros(k, dtype=torch.int32, device=d)
        st.lshape_fit(px, py, pz, label, base, cnt, 1, 1, calib, z32(1), z32(1),
                      torch.zeros(4, dtype=torch.float32, device=d),
                      torch.zeros(1, dtype=torch.float32, device=d), n_theta, dtheta,
                      self._p.car_aspect_max, torch.zeros(cap, dtype=torch.float64, device=d),
                      torch.zeros(cap, dtype=torch.float64, device=d), z32(cap), n, rows, n_rows,
                      status, cap)
        k = int(n_rows.item())
        R = rows.view(n, st.ROW_DOUBLES)[:k].cpu().numpy()
        R = R[np.argsort(R[:, C_CLUSTER])]
        lab = label[:n].cpu().numpy()
        rects, idsets = [], []
        for r in R:
            th = r[C_THETA]
            c, s = np.cos(th), np.sin(th)
            rect = RectangleData()
            rect.a = [c, -s, c, -s]
            rect.b = [s, c, s, c]
            rect.c = [r[C_RECT_C], r[C_RECT_C + 1], r[C_RECT_C + 2], r[C_RECT_C + 3]]
            rects.append(rect)
        for root in np.unique(lab):
            idsets.append(set(np.nonzero(lab == root)[0].tolist()))
        return rects, idsets
