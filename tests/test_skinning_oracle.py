"""oracle/skinning.py (warp_vertices, svox_kernel.cu:123-211) against direct float64
linear algebra and finite differences.  No GPU."""
import numpy as np

from oracle import skinning as S


def case(Q=60, J=6, B=3, seed=0):
    rng = np.random.default_rng(seed)
    mats = rng.normal(size=(J, 4, 4)).astype(np.float32)
    mats[:, 3] = [0, 0, 0, 1]
    p = rng.normal(size=(Q, 3)).astype(np.float32)
    sw = rng.random((Q, B)).astype(np.float32)
    sw[rng.random((Q, B)) < 0.3] = 0
    sw[rng.random((Q, B)) < 0.05] = -0.25                    # non-positive weights are skipped (:140)
    ji = rng.integers(0, J, size=(Q, B))
    return mats, p, sw, ji


def blended64(mats, sw, ji):
    Q, B = sw.shape
    M = np.zeros((Q, 4, 4))
    for j in range(B):
        M[:, :3, :] += np.where(sw[:, j] > 0, sw[:, j], 0.0)[:, None, None] * mats[ji[:, j], :3, :].astype(np.float64)
    M[:, 3, 3] = 1.0
    return M


def test_forward_is_the_blended_affine_map():
    mats, p, sw, ji = case()
    v, m = S.warp_vertices(mats, p, sw, ji)
    M = blended64(mats, sw, ji)
    np.testing.assert_allclose(m, M, atol=5e-7)
    assert (m[:, 3] == [0, 0, 0, 1]).all()
    np.testing.assert_allclose(v, np.einsum("qij,qj->qi", M[:, :3, :3], p) + M[:, :3, 3], atol=2e-6)
    # one joint, weight one: the joint's own transform
    v1, m1 = S.warp_vertices(mats, p, np.ones((len(p), 1), np.float32), np.full((len(p), 1), 2))
    np.testing.assert_array_equal(m1[:, :3], np.broadcast_to(mats[2, :3], (len(p), 3, 4)))


def test_backward_matches_finite_differences():
    mats, p, sw, ji = case(Q=24, J=4, B=3, seed=3)
    rng = np.random.default_rng(9)
    gv = rng.normal(size=p.shape).astype(np.float32)
    gm = rng.normal(size=(len(p), 4, 4)).astype(np.float32)
    gp, gmat, gabs, gsw = S.warp_vertices_backward(mats, p, sw, ji, gv, gm)

    def loss(mats_, p_, sw_):
        M = blended64(mats_, sw_, ji)
        v = np.einsum("qij,qj->qi", M[:, :3, :3], p_) + M[:, :3, 3]
        return (v * gv).sum() + (M[:, :3] * gm[:, :3]).sum()

    eps = 1e-4
    m64, p64, s64 = mats.astype(np.float64), p.astype(np.float64), sw.astype(np.float64)
    for a in range(mats.shape[0]):
        for r in range(3):
            for c in range(4):
                e = np.zeros_like(m64)
                e[a, r, c] = eps
                fd = (loss(m64 + e, p64, s64) - loss(m64 - e, p64, s64)) / (2 * eps)
                assert abs(fd - gmat[a, r, c]) <= 1e-5 * gabs[a, r, c] + 1e-6
    assert not gmat[:, 3].any()                                 # row 3 of the joint matrices is never read
    for q in range(0, len(p), 5):
        for i in range(3):
            e = np.zeros_like(p64)
            e[q, i] = eps
            fd = (loss(m64, p64 + e, s64) - loss(m64, p64 - e, s64)) / (2 * eps)
            assert abs(fd - gp[q, i]) < 1e-4 * max(1.0, abs(fd))
    for q, j in np.argwhere(sw > 0)[::7]:
        e = np.zeros_like(s64)
        e[q, j] = eps
        fd = (loss(m64, p64, s64 + e) - loss(m64, p64, s64 - e)) / (2 * eps)
        assert abs(fd - gsw[q, j]) < 1e-4 * max(1.0, abs(fd))
    assert (gsw[sw <= 0] == 0).all()
