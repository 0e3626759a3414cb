"""Writes tests/golden/heightfield.npz: outputs of the reference's ``convert_height_field_to_mesh``
(isaaclab/terrains/height_field/utils.py:79-173) for seeded height fields, with and without the slope correction.

Runs only where the reference checkout is mounted (the build container); the fixture it writes is data.
"""

import os

import numpy as np

import _refload as R


def main():
    R.placeholder("trimesh")
    R.placeholder("isaaclab")
    R.placeholder("isaaclab.terrains")
    R.placeholder("isaaclab.terrains.height_field")
    m = R.load("isaaclab.terrains.height_field.utils", f"{R.L}/terrains/height_field/utils.py")
    rng = np.random.default_rng(11)
    out = {}
    cases = {
        "noise": (rng.integers(-8, 9, size=(13, 17)).astype(np.int16), 0.1, 0.005, None),
        "noise_corrected": (rng.integers(-20, 21, size=(15, 12)).astype(np.int16), 0.1, 0.005, 0.75),
        "stairs_corrected": ((np.minimum.outer(np.arange(19) // 3, np.arange(23) // 3) * 30).astype(np.int16), 0.1, 0.005, 0.75),
        "float_heights": (rng.uniform(-0.3, 0.3, size=(9, 9)), 0.25, 1.0, None),
    }
    for tag, (hf, hs, vs, thr) in cases.items():
        v, t = m.convert_height_field_to_mesh(hf, hs, vs, thr)
        out[f"{tag}_heights"] = hf
        out[f"{tag}_params"] = np.array([hs, vs, -1.0 if thr is None else thr])
        out[f"{tag}_vertices"] = v
        out[f"{tag}_triangles"] = t.astype(np.int64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "heightfield.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
