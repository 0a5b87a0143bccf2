"""Meshes for the voxelizer tests: name -> (positions, indices, cell_size, ref_offset, ref_size)."""
import numpy as np

from libfluid_amd import scenes


def make(name):
    if name == "sphere":  # closed surface inside the reference grid, unit cells
        pos, idx = scenes.icosphere((10.3, 9.7, 11.1), 6.4, 2)
        return pos, idx, 1.0, (0.0, 0.0, 0.0), (24, 24, 24)
    if name == "box_rot":  # rotated box, fractional cell size and offsets: the running sums of the cell centres round
        pos, idx = scenes.box_mesh((2.0, 3.0, 4.0), (7.5, 6.25, 9.0))
        pos = scenes.rotate_mesh(pos, (1.0, 2.0, 0.5), 0.6, (4.0, 4.0, 6.0))
        return pos, idx, 0.3, (-1.25, 0.37, 2.01), (40, 40, 40)
    if name == "sphere_clip":  # sticks out of the reference grid on the low and the high side
        pos, idx = scenes.icosphere((2.2, 5.1, 13.4), 4.7, 2)
        return pos, idx, 1.0, (0.0, 0.0, 0.0), (16, 16, 16)
    if name == "open_sheet":  # not closed: nothing is interior after the flood fill
        pos = np.array([[1.1, 1.2, 3.3], [9.7, 1.4, 3.9], [5.2, 8.8, 4.4], [9.9, 9.1, 3.1]], dtype=np.float64)
        idx = np.array([0, 1, 2, 1, 3, 2], dtype=np.uint64)
        return pos, idx, 0.5, (0.1, 0.2, 0.3), (24, 24, 16)
    if name == "long_sliver":  # bounding box longer than one LDS chunk of cell centres (128) along x and y
        pos = np.array([[0.31, 0.52, 0.77], [19.93, 14.1, 1.21], [0.95, 13.6, 0.9]], dtype=np.float64)
        idx = np.array([0, 1, 2], dtype=np.uint64)
        return pos, idx, 0.1, (0.013, -0.027, 0.05), (210, 150, 20)
    if name == "two_shells":  # a sphere inside a box: the cavity between them is interior (not reachable from the corner)
        p1, i1 = scenes.box_mesh((2.2, 2.2, 2.2), (13.8, 13.8, 13.8))
        p2, i2 = scenes.icosphere((8.0, 8.0, 8.0), 3.3, 1)
        return np.concatenate([p1, p2]), np.concatenate([i1, i2 + np.uint64(len(p1))]), 1.0, (0.0, 0.0, 0.0), (16, 16, 16)
    raise KeyError(name)


NAMES = ["sphere", "box_rot", "sphere_clip", "open_sheet", "long_sliver", "two_shells"]
