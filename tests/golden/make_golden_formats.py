"""Writes tests/golden/formats/* with the REAL reference's writers (oracle/_ref: point_cloud::save_to_naive,
mesh::save_obj). Build container only. The files are outputs (data), inputs come from tests/test_formats.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import loader as orc  # noqa: E402
from tests import test_formats as tf  # noqa: E402

if __name__ == "__main__":
    orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so is not built")
    open(os.path.join(tf.GOLD, "points.txt"), "wb").write(orc.ref_points_text(tf.sample_points()))
    pos, idx, uv = tf.sample_mesh()
    for mode in range(8):
        text, _ = orc.ref_mesh_obj(pos, idx, uv if mode & 2 else None, bool(mode & 1), bool(mode & 4))
        open(os.path.join(tf.GOLD, f"mesh_mode{mode}.obj"), "wb").write(text)
    print(sorted(os.listdir(tf.GOLD)))
