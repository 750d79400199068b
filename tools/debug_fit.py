import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, math
from dfu3d_amd import stages as st
from dfu3d_amd.params import Params
from oracle import penet_oracle as O
rng = np.random.default_rng(14)
def lshape(cx, cy, L, Wd, yaw, n):
    k = n // 2
    e1 = np.stack([rng.uniform(-L / 2, L / 2, k), np.full(k, -Wd / 2)], 1)
    e2 = np.stack([np.full(n - k, -L / 2), rng.uniform(-Wd / 2, Wd / 2, n - k)], 1)
    q = np.vstack([e1, e2]) + rng.normal(0, 0.02, (n, 2))
    R = np.array([[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]])
    return (q @ R.T + np.array([cx, cy]))[rng.permutation(n)]
xy = lshape(15, -9, 4.4, 1.8, -0.9, 4000)
n = len(xy); cap = n + 8; dev = "cuda:0"
p = Params(); n_theta, dtheta = p.thetas()
px = torch.zeros(cap, dtype=torch.float64, device=dev); py = px.clone(); pz = px.clone()
px[:n] = torch.from_numpy(xy[:, 0]); py[:n] = torch.from_numpy(xy[:, 1])
base = torch.zeros(1, dtype=torch.int64, device=dev); cnt = torch.full((1,), n, dtype=torch.int32, device=dev)
label = torch.zeros(cap, dtype=torch.int32, device=dev)
st.range_cluster(px, py, base, cnt, 1, p.R0, p.Rd, label, cap)
nws = int(st._lib.lib().dfu3d_lshape_fit_ws_doubles(cap, 4096))
ws = torch.full((nws,), -777.0, dtype=torch.float64, device=dev)
rows = torch.zeros(8 * st.ROW_DOUBLES, dtype=torch.float64, device=dev); nr = torch.zeros(1, dtype=torch.int32, device=dev); stt = torch.zeros(1, dtype=torch.int32, device=dev)
z32 = lambda k: torch.zeros(k, dtype=torch.int32, device=dev)
sx = torch.zeros(cap, dtype=torch.float64, device=dev); sy = sx.clone()
st.lshape_fit(px, py, pz, label, base, cnt, 1, 1, torch.zeros(48, dtype=torch.float32, device=dev), z32(1), z32(1),
              torch.zeros(4, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.float32, device=dev), n_theta, dtheta,
              5.0, sx, sy, z32(cap), 8, rows, nr, stt, cap, ws)
torch.cuda.synchronize()
w = ws.cpu().numpy(); cap_big = cap // 2048 + 1
print("counter", w[:1].view(np.int32)[:2], "desc", w[2:10], "cap_big", cap_big, "nrows", int(nr))
cost = w[2 + cap_big * 8: 2 + cap_big * 8 + 96]
x, y = xy[:, 0], xy[:, 1]
oc = []
for k in range(89):
    th = k * dtheta; c, s = np.cos(th), np.sin(th)
    oc.append(O.variance_criterion(x * c + y * s, x * (-s) + y * c))
oc = np.array(oc)
np.set_printoptions(linewidth=250, precision=4); print("gpu", cost[:96]); print("orc", oc)
print("max abs diff", np.abs(cost[:89] - oc).max(), "argmax gpu", cost[:89].argmax(), "orc", oc.argmax())
print("members equal?", np.array_equal(np.sort(sx[:n].cpu().numpy()), np.sort(x)))
print("row", rows[:24].cpu().numpy())
