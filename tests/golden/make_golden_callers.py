"""Golden outputs of tests/callers/reference_callers.cpp linked with the REAL reference (oracle/_ref/callers_ref, built by
`make -C oracle callers` from /root/reference): what the reference computes when its own hosts' call sequences are run on the
CPU. tests/test_ref_callers.py holds the device build of the same source to these numbers on the GPU box, where the reference
does not exist.  Usage: python tests/golden/make_golden_callers.py   (in the build container)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import callers_util as cu  # noqa: E402


def slim(name, rec):
    """Keeps what the test compares: particle arrays of the first and the last frame only, counts as small integers."""
    frames = sorted({int(k[5:k.index(".")]) for k in rec if k.startswith("frame")})
    out = {}
    for k, v in rec.items():
        if k.startswith("frame"):
            f, field = int(k[5:k.index(".")]), k[k.index(".") + 1:]
            if field in ("pos", "vel", "grid_vel") and f not in (frames[0], frames[-1]):
                if field == "pos":
                    out[f"frame{f}.count"] = np.int64(len(v) // 3)
                continue
            if field == "occupation":
                v = v.astype(np.uint16)
        out[k] = v
    return out


def main():
    exe = cu.build_reference()
    d = tempfile.mkdtemp()
    blob = {}
    for name in cu.SCENARIOS:
        rec, stdout, texts = cu.run(exe, name, d)
        for k, v in slim(name, rec).items():
            blob[f"{name}/{k}"] = v
        for fn, text in texts.items():
            blob[f"{name}/{fn}"] = np.frombuffer(text.encode(), dtype=np.uint8)
        blob[f"{name}/stdout"] = np.frombuffer(stdout.encode(), dtype=np.uint8)
    path = os.path.join(cu.util.GOLDEN_DIR, "ref_callers.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes,", len(blob), "arrays")


if __name__ == "__main__":
    main()
