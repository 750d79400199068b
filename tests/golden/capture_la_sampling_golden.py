"""Golden G10: the reference's own `la_sampling` (row f-4 of SURVEY.md section 8) on seeded objects.

Run in the build container (needs /root/reference; nothing at test time does):
    python tests/golden/capture_la_sampling_golden.py        ->  tests/golden/g10_la_sampling.npz

/root/reference/pcdet/datasets/augmentor/database_sampler_virtual.py is imported UNMODIFIED.  Its two relative imports
(`...ops.iou3d_nms.iou3d_nms_utils`: a compiled CUDA extension, `...utils.box_utils`) are not used by `la_sampling` /
`to_sphere_coords`; like numba in capture_eval_golden.py they are provided as empty import-time stand-ins, and the
module is loaded as a member of a package skeleton so that pcdet/__init__ (which imports the compiled ops) is not
touched.  `DADataBaseSampler.la_sampling` (:307-351) is then called unbound on an object made with object.__new__ (the method uses
no state).

Per object the file holds the input (n, C) float32, the reference's output rows, the indices of those rows in the
input (first match; -1 x n when the object came back unchanged because fewer than five bins survived) and a flag
`fragile`: one of the object's float32 angles lies within 2 ulp of a bin edge, or two theta of one bin lie within
2 ulp of each other.  Only for those objects may an implementation whose arccos / arctan differ from this host's libm
in the last bit select other rows; every other object must match bit for bit.

NumPy semantics recorded with the data (hazard): `point[1] // vert_res` is float32 // Python float.  Under
NumPy >= 2 (NEP 50; this image: see `numpy_version` in the file) the result stays float32 and the divisor is rounded
to float32 first; under NumPy 1.x, which the reference was written against (its requirements are unpinned), the
same expression promotes to float64 and divides by the double 0.006 -- bin edges then differ by ~1e-8 rad and
about 2e-6 of all points change bins.  G10 pins the NumPy 2 reading (the one a present-day install of the
reference computes); the float64 reading is kept as `keys64_differ`: the number of points per object whose key
differs under NumPy 1.x promotion, for the record.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/pcdet/datasets/augmentor/database_sampler_virtual.py'


def load_reference():
    for name in ('pcdet', 'pcdet.ops', 'pcdet.ops.iou3d_nms', 'pcdet.ops.iou3d_nms.iou3d_nms_utils', 'pcdet.utils',
                 'pcdet.utils.box_utils', 'pcdet.datasets', 'pcdet.datasets.augmentor'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules['pcdet.ops.iou3d_nms'].iou3d_nms_utils = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils']
    sys.modules['pcdet.utils'].box_utils = sys.modules['pcdet.utils.box_utils']
    spec = importlib.util.spec_from_file_location('pcdet.datasets.augmentor.database_sampler_virtual', REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_objects():
    rng = np.random.default_rng(1010)
    objs, tags = [], []

    def obj(n, centre, C=8, spread=(0.9, 0.45, 0.4)):
        p = np.zeros((n, C), np.float32)
        p[:, :3] = (np.asarray(centre) + rng.normal(0, spread, (n, 3))).astype(np.float32)
        p[:, 3:] = rng.random((n, C - 3)).astype(np.float32)
        return p
    sizes = [1, 2, 3, 4, 5, 6, 9, 17, 33, 64, 65, 128, 200, 333, 500, 777, 1000, 1500, 2500, 4000]
    for k in range(44):
        n = sizes[k % len(sizes)]
        d, a = rng.uniform(4, 60), rng.uniform(-np.pi, np.pi)
        objs.append(obj(n, [d * np.cos(a), d * np.sin(a), rng.uniform(-1.5, 0.5)], C=8 if k % 3 else 4))
        tags.append("random")
    objs.append(np.zeros((7, 8), np.float32)); tags.append("all_nan")                     # r = 0: every angle NaN -> one bin
    z = obj(40, [20.0, 3.0, -1.0]); z[::5, :3] = 0.0
    objs.append(z); tags.append("some_nan")
    objs.append(obj(300, [0.0, 12.0, -1.0])); tags.append("x_around_zero")                # huge |y/x|, both signs
    q = obj(200, [15.0, 0.0, -1.0]); q[::7, 0] = 0.0
    objs.append(q); tags.append("x_exactly_zero")                                         # y/x = +-inf -> fan = +-pi/2
    objs.append(obj(30, [30.0, 0.0, 0.0], spread=(0.01, 0.01, 0.01))); tags.append("few_bins")   # < 5 survivors -> unchanged
    objs.append(obj(3, [10.0, 1.0, 0.0])); tags.append("three_points")
    t = obj(64, [25.0, -4.0, -0.5]); t[1::2] = t[0::2]
    objs.append(t); tags.append("duplicates")                                             # equal theta inside a bin: first wins
    objs.append(obj(500, [-18.0, 7.0, -1.2])); tags.append("behind")                      # x < 0: arctan folds
    objs.append(obj(400, [6.0, -2.0, 2.5])); tags.append("above")                         # theta < pi/2
    return objs, tags


def fragile(ref, sampler, points, vr, hr):
    sp = ref.DADataBaseSampler.to_sphere_coords(sampler, points.copy())
    th, fan = sp[:, 1], sp[:, 2]
    vrf, hrf = np.float32(vr), np.float32(hr)
    with np.errstate(all='ignore'):
        for arr, res in ((th, vrf), (fan, hrf)):
            for k in (-2, -1, 1, 2):
                moved = arr.copy()
                for _ in range(abs(k)):
                    moved = np.nextafter(moved, np.float32(np.inf if k > 0 else -np.inf))
                a, b = arr // res, moved // res
                if np.any((a != b) & ~(np.isnan(a) & np.isnan(b))):
                    return True
        keys = np.stack([th // vrf, fan // hrf], 1)
    keys = np.where(np.isnan(keys), np.float32(1e30), keys)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    for b in np.unique(inv):
        t = np.sort(th[inv == b])
        t = t[~np.isnan(t)]
        if len(t) > 1 and (np.diff(t) <= 2 * np.spacing(t[:-1])).any():
            return True
    return False


def main():
    ref = load_reference()
    sampler = object.__new__(ref.DADataBaseSampler)
    objs, tags = make_objects()
    out = {"numpy_version": np.array(np.__version__), "n_objects": np.array(len(objs)), "tags": np.array(tags),
           "res": np.array([[0.006, 0.003], [0.02, 0.01]])}
    for ri, (vr, hr) in enumerate(out["res"].tolist()):
        for k, o in enumerate(objs):
            with np.errstate(all='ignore'):
                got = ref.DADataBaseSampler.la_sampling(sampler, o.copy(), vr, hr)
            got = np.asarray(got, np.float32).reshape(-1, o.shape[1])
            unchanged = got.shape == o.shape and np.array_equal(got.view(np.uint32), o.view(np.uint32)) and (
                len(o) < 5 or len(np.unique(got.view(np.uint32), axis=0)) == len(got) or True)
            # indices of the selected rows (first match in input order, each input row used once)
            idx = np.full(len(got), -1, np.int64)
            used = np.zeros(len(o), bool)
            ov, gv = o.view(np.uint32), got.view(np.uint32)
            for r in range(len(got)):
                m = np.nonzero((ov == gv[r]).all(1) & ~used)[0]
                assert len(m), "an output row is not an input row"
                idx[r] = m[0]
                used[m[0]] = True
            sp = ref.DADataBaseSampler.to_sphere_coords(sampler, o.copy())
            with np.errstate(all='ignore'):
                k32 = np.stack([sp[:, 1] // vr, sp[:, 2] // hr], 1)
                k64 = np.stack([sp[:, 1].astype(np.float64) // vr, sp[:, 2].astype(np.float64) // hr], 1)
            diff64 = int(np.sum(~((k32.astype(np.float64) == k64) | (np.isnan(k32) & np.isnan(k64))).all(1)))
            if ri == 0:
                out["in/%d" % k] = o
            out["out/%d/%d" % (ri, k)] = got
            out["idx/%d/%d" % (ri, k)] = idx
            out["fragile/%d/%d" % (ri, k)] = np.array(fragile(ref, sampler, o, vr, hr))
            out["keys64_differ/%d/%d" % (ri, k)] = np.array(diff64)
    path = os.path.join(HERE, "g10_la_sampling.npz")
    np.savez_compressed(path, **out)
    nf = sum(bool(out["fragile/0/%d" % k]) for k in range(len(objs)))
    print("wrote %s: %d objects x %d resolutions, %d fragile at the product resolution, numpy %s, %.1f KB"
          % (path, len(objs), len(out["res"]), nf, np.__version__, os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
