"""TEST INFRASTRUCTURE (not product code): CPU restatement of `la_sampling`
(/root/reference/pcdet/datasets/augmentor/database_sampler_virtual.py:307-351), row f-4 of SURVEY.md §8.

Same operations on the same NumPy float32 values as the reference's loop (to_sphere_coords :307-317; per point
`point[1] // vert_res`, `point[2] // hor_res` and the key `str(vert) + '_' + str(hor)` :330-333; np.argmin of theta per
bin :345; dict order = first-seen order; fewer than 5 rows -> the input :348-349).
PINNED: golden G10 (tests/golden/g10_la_sampling.npz, written by tests/golden/capture_la_sampling_golden.py) holds the
outputs of the reference's own DADataBaseSampler.la_sampling -- the module imported unmodified, its two unused relative
imports (a compiled CUDA extension, box_utils) provided as empty stand-ins -- on 53 seeded objects at two
resolutions; tests/test_oracle_la_sampling.py asks for the same rows in the same order, bit for bit, for every one.
NumPy semantics: `float32 // python float` stays float32 under NumPy >= 2 (this image; what G10 pins) and promotes to
float64 under NumPy 1.x; the capture script records how many keys would differ (`keys64_differ`)."""
import numpy as np


def to_sphere_coords(points):
    r = np.linalg.norm(points[:, 0:3], ord=2, axis=-1)
    with np.errstate(all="ignore"):
        theta = np.arccos(points[:, 2] / r)
        fan = np.arctan(points[:, 1] / points[:, 0])
    new_points = points.copy()
    new_points[:, 0] = r
    new_points[:, 1] = theta
    new_points[:, 2] = fan
    return new_points


def la_sampling(points, vert_res=0.006, hor_res=0.003, return_index=False):
    points = np.asarray(points)
    sp = to_sphere_coords(points)
    voxels = {}
    with np.errstate(all="ignore"):
        for i in range(len(sp)):
            key = str(sp[i, 1] // vert_res) + '_' + str(sp[i, 2] // hor_res)
            voxels.setdefault(key, []).append(i)
    idx = []
    for key, members in voxels.items():
        th = sp[members, 1]
        idx.append(members[int(np.argmin(th))])
    if len(idx) < 5:
        return (points, np.arange(len(points))) if return_index else points
    idx = np.asarray(idx, np.int64)
    return (points[idx], idx) if return_index else points[idx]
