"""GPU tests of the octree builder (csrc/svoxt_build.hip) through
N3Tree.build_from_points / construct_tree -> ctypes -> C ABI: bit-exact against
the CPU oracle (oracle/builder.py), against the tables the reference's
N3Tree.refine produced (tests/golden/topology_points_*.npz), and against this
package's own step-by-step route `tree[points].refine()` x (depth-1) +
`construct_tree(points)`."""
import os

import numpy as np
import pytest
import torch

import svox_t_amd as svox
from oracle import builder as ob

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def cloud(n, seed, radius=(0.5, 0.5, 0.5), center=(0.5, 0.5, 0.5), spread=0.05):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.6 + spread * rng.normal(size=(n, 1))
    pts = (np.asarray(center) + np.asarray(radius) * r * d).astype(np.float32)
    pts[: max(1, n // 50)] *= 3.0
    if n >= 40:
        pts[n // 2: n // 2 + n // 20] = pts[: n // 20]
    return pts


def hip_build(gpu, pts, depth, radius=0.5, center=(0.5, 0.5, 0.5), reserve=0):
    tree = svox.N3Tree(N=2, data_dim=4, radius=radius, center=list(center), map_location=gpu)
    n = tree.build_from_points(torch.from_numpy(pts).to(gpu), depth, reserve=reserve)
    return tree, n


def tables(tree):
    n = tree.n_internal
    return (tree.child[:n].cpu().numpy(), tree.data[:n].cpu().numpy().reshape(n, 2, 2, 2),
            tree.parent_depth[:n].cpu().numpy())


@pytest.mark.parametrize("name", ["a", "b"])
def test_builder_matches_reference_tables(gpu, name):
    g = np.load(os.path.join(G, f"topology_points_{name}.npz"))
    tree, n = hip_build(gpu, g["points"], int(g["depth"]), radius=g["radius"].tolist(), center=g["center"].tolist())
    np.testing.assert_array_equal(tree.offset.cpu().numpy(), g["offset"])
    np.testing.assert_array_equal(tree.invradius.cpu().numpy(), g["scaling"])
    child, data, pd = tables(tree)
    assert n == int(g["n_internal"])
    np.testing.assert_array_equal(child, g["child"])
    np.testing.assert_array_equal(pd, g["parent_depth"])


@pytest.mark.parametrize("n,depth,radius,center", [
    (1, 1, 0.5, (0.5, 0.5, 0.5)),
    (1, 6, 0.5, (0.5, 0.5, 0.5)),
    (7, 2, 0.5, (0.5, 0.5, 0.5)),
    (500, 3, 0.5, (0.5, 0.5, 0.5)),
    (5000, 5, [1.0, 1.2, 0.8], (0.1, -0.2, 0.3)),
    (40000, 7, 0.5, (0.5, 0.5, 0.5)),
    (40000, 8, [2.0, 1.0, 1.5], (1.0, 0.0, -1.0)),
])
def test_builder_matches_oracle(gpu, n, depth, radius, center):
    pts = cloud(n, seed=depth * 1000 + n, radius=[radius] * 3 if isinstance(radius, float) else radius, center=center)
    tree, cnt = hip_build(gpu, pts, depth, radius=radius, center=center)
    want_child, want_data, want_pd = ob.build_from_points(pts, tree.offset.cpu().numpy(),
                                                          tree.invradius.cpu().numpy(), depth)
    child, data, pd = tables(tree)
    assert cnt == want_child.shape[0]
    np.testing.assert_array_equal(child, want_child)
    np.testing.assert_array_equal(pd, want_pd)
    np.testing.assert_array_equal(data, want_data)


def test_builder_equals_the_step_by_step_route(gpu):
    """The reference's call sequence, through this package's N3TreeView.refine /
    N3Tree.refine / construct_tree, gives the same three tables."""
    pts = cloud(20000, seed=3, radius=[1.0, 1.2, 0.8], center=(0.1, -0.2, 0.3))
    depth = 6
    fused, _ = hip_build(gpu, pts, depth, radius=[1.0, 1.2, 0.8], center=(0.1, -0.2, 0.3))
    step = svox.N3Tree(N=2, data_dim=4, radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3], map_location=gpu)
    p = torch.from_numpy(pts).to(gpu)
    for _ in range(depth - 1):
        step[p].refine()
    step.construct_tree(p)
    assert step.n_internal == fused.n_internal
    for a, b in zip(tables(step), tables(fused)):
        np.testing.assert_array_equal(a, b)


def test_built_tree_serves_queries_and_renders(gpu):
    """Row i of `features` is point i: a query at the points reads the keeper's row;
    the renderer accepts the tree (acceleration grid rebuilt for the new tables)."""
    pts = cloud(30000, seed=5)
    tree, n = hip_build(gpu, pts, 7, reserve=100)
    assert tree.capacity == n + 100
    assert int(tree.child[n:].abs().sum()) == 0 and int((tree.data[n:] != svox.svox.EMPTY_INDEX).sum()) == 0
    p = torch.from_numpy(pts).to(gpu)
    feats = torch.randn(pts.shape[0], 4, device=gpu)
    feats[:, 3] = feats[:, 3].abs() * 20
    vals, node_ids, data_ids = tree(feats, p, want_node_ids=True, want_data_ids=True)
    ids = data_ids.cpu().numpy()
    assert ids.min() >= 0 and (ids <= np.arange(len(ids))).all()
    np.testing.assert_array_equal(vals.cpu().numpy(), feats.cpu().numpy()[ids])
    # same leaf <=> same keeper
    nid = node_ids.cpu().numpy()
    assert len(np.unique(nid)) == len(np.unique(ids))
    from svox_t_amd import synth
    o, d, v = synth.pinhole_rays(64, 64)
    r = svox.VolumeRenderer(tree)
    out = r(feats, svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu)))
    assert float(out[:, 3].max()) > 0.5                                   # the sphere is seen
    # rebuilding in place for the next frame: the renderer follows (no stale grid)
    tree.build_from_points(p * 0.5 + 0.25, 7)
    out2 = r(feats, svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu)))
    assert not torch.equal(out, out2)
    from oracle import oracle as O
    n2 = tree.n_internal
    ot = O.Tree(feats.cpu().numpy(), tree.data[:n2].cpu().numpy(), tree.child[:n2].cpu().numpy())
    want = O.volume_render(ot, o.numpy(), d.numpy(), v.numpy(), O.make_options())
    np.testing.assert_array_equal(out2.cpu().numpy(), want)


def test_construct_tree_overwrites_and_is_deterministic(gpu):
    pts = cloud(5000, seed=9)
    tree, _ = hip_build(gpu, pts, 5)
    p = torch.from_numpy(pts).to(gpu)
    before = tree.data.clone()
    tree.construct_tree(p.flip(0))                       # other order: index i now means point P-1-i
    again = tree.data.clone()
    tree.construct_tree(p)
    assert torch.equal(tree.data, before)
    assert not torch.equal(again, before)
    # N = 3 goes through the generic descent
    t3 = svox.N3Tree(N=3, data_dim=4, init_reserve=64, map_location=gpu)
    t3.refine(1)
    t3.construct_tree(p)
    _, packed = t3(torch.zeros(5000, 4, device=gpu), p, want_node_ids=True)
    got = t3.data.view(-1)[packed].cpu().numpy()
    pk = packed.cpu().numpy()
    want = np.full(pk.max() + 1, 1 << 40, dtype=np.int64)
    np.minimum.at(want, pk, np.arange(len(pk)))
    np.testing.assert_array_equal(got, want[pk])


def test_full_depth_properties(gpu):
    """depth 10 (the largest): node count = number of distinct cell prefixes; every
    point finds a finest leaf that keeps the smallest index among its points."""
    rng = np.random.default_rng(1)
    pts = rng.random((200000, 3), dtype=np.float32)
    tree, n = hip_build(gpu, pts, 10)
    p = torch.from_numpy(pts).to(gpu)
    q = torch.clamp(p, 0.0, float(np.float32(1.0 - 1e-6)))
    fix = (q * 4194304.0).to(torch.int64)
    want = 1
    for lvl in range(1, 10):
        c = fix >> (22 - lvl)
        key = (c[:, 0] << 40) | (c[:, 1] << 20) | c[:, 2]
        want += int(torch.unique(key).numel())
    assert n == want
    assert int(tree.parent_depth[:n, 1].max()) == 9
    _, packed, data_ids = tree(torch.zeros(len(pts), 4, device=gpu), p, want_node_ids=True, want_data_ids=True)
    ids = data_ids.cpu().numpy()
    pk = packed.cpu().numpy()
    assert ids.min() >= 0
    first = np.full(pk.max() + 1, 1 << 40, dtype=np.int64)
    np.minimum.at(first, pk, np.arange(len(pk)))
    np.testing.assert_array_equal(ids, first[pk])
    assert (tree.parent_depth[packed // 8, 1] == 9).all()


def test_builder_argument_errors(gpu):
    p = torch.rand(10, 3, device=gpu)
    tree = svox.N3Tree(N=2, data_dim=4, map_location=gpu)
    with pytest.raises(RuntimeError):
        tree.build_from_points(p, 0)
    with pytest.raises(RuntimeError):
        tree.build_from_points(p, 11)
    with pytest.raises(RuntimeError):
        svox.N3Tree(N=3, data_dim=4, map_location=gpu).build_from_points(p, 3)
    with pytest.raises(RuntimeError):
        svox.N3Tree(N=2, data_dim=4).build_from_points(p.cpu(), 3)
    assert tree.build_from_points(torch.empty(0, 3, device=gpu), 4) == 1         # no points: the root alone


@pytest.mark.parametrize("N", [2, 3])
def test_refine_on_gpu_equals_refine_on_cpu(gpu, N):
    """N3Tree.refine with tensors on the GPU runs svoxt_refine; on the CPU the tensor
    ops of svox.py:535-546.  Same tables, selective and full, with and without node_id."""
    g = torch.Generator().manual_seed(N)
    a = svox.N3Tree(N=N, data_dim=4, init_reserve=4)
    b = svox.N3Tree(N=N, data_dim=4, init_reserve=4, map_location=gpu)
    for rnd in range(4 if N == 2 else 3):
        leaves = a._all_leaves()
        pick = torch.rand(len(leaves), generator=g) < (1.0 if rnd == 0 else 0.4)
        sel = leaves[pick]
        nid = torch.arange(len(sel), dtype=torch.int32) * 7 if rnd == 2 else None
        # give the leaves distinct data words so that inheritance is visible
        a.data.view(-1)[: a.n_internal * N ** 3] = torch.arange(a.n_internal * N ** 3, dtype=torch.int32)
        b.data.view(-1)[: b.n_internal * N ** 3] = torch.arange(b.n_internal * N ** 3, dtype=torch.int32, device=gpu)
        ra = a.refine(1, sel=tuple(sel.T), leaf_node=sel, node_id=nid)
        rb = b.refine(1, sel=tuple(sel.to(gpu).T), leaf_node=sel.to(gpu), node_id=None if nid is None else nid.to(gpu))
        assert ra == rb and a.n_internal == b.n_internal
        n = a.n_internal
        assert torch.equal(a.child[:n], b.child[:n].cpu())
        assert torch.equal(a.data[:n], b.data[:n].cpu())
        assert torch.equal(a.parent_depth[:n], b.parent_depth[:n].cpu())
    assert torch.equal(a._all_leaves(), b._all_leaves())
