"""oracle/builder.py (the CPU restatement of the reference's per-frame octree
build) against tables the reference's own N3Tree.refine produced
(tests/golden/topology_points_*.npz), plus properties of the result.  No GPU."""
import os

import numpy as np
import pytest

from oracle import builder as ob

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["a", "b"])
def test_refine_loop_matches_reference_tables(name):
    g = np.load(os.path.join(G, f"topology_points_{name}.npz"))
    topo = ob.Topology(N=2)
    for _ in range(int(g["depth"]) - 1):
        ob.refine(topo, ob.unique_leaves(topo, ob.descend(topo, g["points"], g["offset"], g["scaling"])))
    child, data, pd = topo.arrays()
    assert topo.n == int(g["n_internal"])
    np.testing.assert_array_equal(child, g["child"])
    np.testing.assert_array_equal(pd, g["parent_depth"])
    np.testing.assert_array_equal(data, g["data"].reshape(data.shape))      # refine() only copies the sentinel


def test_construct_gives_every_point_a_leaf_and_smallest_index_wins():
    g = np.load(os.path.join(G, "topology_points_b.npz"))
    pts, off, scl = g["points"], g["offset"], g["scaling"]
    child, data, pd = ob.build_from_points(pts, off, scl, int(g["depth"]))
    np.testing.assert_array_equal(child, g["child"])
    topo = ob.Topology(N=2)
    topo.child, topo.data, topo.parent_depth, topo.n = child, data, pd, child.shape[0]
    packed = ob.descend(topo, pts, off, scl)
    ids = data.reshape(-1)[packed]
    assert ids.min() >= 0 and ids.max() < pts.shape[0]
    assert (ids <= np.arange(pts.shape[0])).all()               # the keeper is never a later point
    np.testing.assert_array_equal(packed[ids], packed)         # ... and it lies in the same leaf
    # every occupied leaf is a finest-level leaf; everything else holds the sentinel
    occupied = data.reshape(-1) != ob.EMPTY_INDEX
    assert occupied.sum() == np.unique(packed).shape[0]
    assert (pd[np.nonzero(occupied)[0] // 8, 1] == int(g["depth"]) - 1).all()


def test_depth_one_is_the_root_alone():
    pts = np.float32([[0.1, 0.2, 0.3], [0.9, 0.9, 0.9], [0.12, 0.22, 0.32]])
    child, data, pd = ob.build_from_points(pts, np.zeros(3, np.float32), np.ones(3, np.float32), 1)
    assert child.shape[0] == 1 and not child.any()
    assert data.reshape(-1)[0] == 0 and data.reshape(-1)[7] == 1
    assert (data.reshape(-1)[1:7] == ob.EMPTY_INDEX).all()
