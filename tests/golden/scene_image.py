"""Deterministic RGB test pattern shared by capture_goldens.py and the tests
(kept as a formula so the 900x1600x3 image is not stored in the fixtures)."""
import numpy as np


def pattern_image(H=900, W=1600):
    yy, xx = np.mgrid[0:H, 0:W]
    return np.stack([(xx * 3 + yy * 5) % 256, (xx + 2 * yy) % 256,
                     (xx // 4 * 7 + yy // 4 * 3) % 256], -1).astype(np.int32)
