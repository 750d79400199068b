"""TEST INFRASTRUCTURE (not product code): CPU restatement of `la_sampling`
(/root/reference/pcdet/datasets/augmentor/database_sampler_virtual.py:307-351), row f-4 of SURVEY.md §8.

Same operations on the same NumPy float32 values as the reference's loop (to_sphere_coords :307-317; per point
`point[1] // vert_res`, `point[2] // hor_res` and the key `str(vert) + '_' + str(hor)` :330-333; np.argmin of theta per
bin :345; dict order = first-seen order; fewer than 5 rows -> the input :348-349).  The reference module itself imports
the whole pcdet package (CUDA extensions) and cannot be imported here: parity unpinned against a run of the reference,
pinned by construction (the statements are the reference's own NumPy calls, element by element)."""
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
