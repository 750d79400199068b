"""Dev tool (GPU box): undecided fraction of the float32 tier per camera / depth range (dfu3d_selftest_classify)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dfu3d_amd import stages as st, synth
rng = np.random.default_rng(0)
geom, _ = st.make_geom()
for H, W in ((900, 1600), (180, 320)):
    for yaw in synth.CAM_YAWS_DEG:
        cal = synth.make_calibration(yaw, H, W, rng)
        for key_axis, (d_lo, d_hi) in ((1, (0.01, 3.0)), (2, (0.5, 95.0)), (2, (2.0, 60.0))):
            r = st.selftest_classify(cal.record(), H, W, geom, key_axis, 4_000_000, seed=int(yaw), d_lo=d_lo, d_hi=d_hi)
            print(H, W, yaw, (d_lo, d_hi), r, "undecided %.4f" % (r["undecided"] / r["tried"]), flush=True)
