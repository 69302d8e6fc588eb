"""CPU restatement of the reference's linear blend skinning of points
(warp_vertices, svox_t/csrc/svox_kernel.cu:123-154, and its backward, :156-211).

TEST INFRASTRUCTURE ONLY.  numpy, float32, the reference's order of operations
(one rounded product and one rounded add per `+=`); the joint-matrix gradient,
which the reference accumulates with float atomics in undefined order, is
summed in float64 and returned with the summed magnitudes for tolerances.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _blend(matrices, sw, ji):
    """matrix_out rows 0..2 ([Q, 3, 4]): sum over bound joints with positive weight, j ascending (:139-146)."""
    Q, B = sw.shape
    mo = np.zeros((Q, 3, 4), f32)
    for j in range(B):
        w = sw[:, j].astype(f32)
        use = w > 0
        term = (w[:, None, None] * matrices[ji[:, j], :3, :]).astype(f32)
        mo = np.where(use[:, None, None], (mo + term).astype(f32), mo)
    return mo


def warp_vertices(matrices, points, skinning_weights, joint_index):
    """-> (vertices_out [Q, 3], matrix_out [Q, 4, 4]) float32."""
    matrices = np.asarray(matrices, f32)
    p = np.asarray(points, f32)
    sw = np.asarray(skinning_weights, f32)
    ji = np.asarray(joint_index, np.int64)
    mo = _blend(matrices, sw, ji)
    out = np.zeros((p.shape[0], 4, 4), f32)
    out[:, :3, :] = mo
    out[:, 3, 3] = 1.0                                               # :148
    v = np.empty_like(p)
    for i in range(3):                                               # :151-153, left to right
        acc = (p[:, 0] * mo[:, i, 0]).astype(f32)
        acc = (acc + (p[:, 1] * mo[:, i, 1]).astype(f32)).astype(f32)
        acc = (acc + (p[:, 2] * mo[:, i, 2]).astype(f32)).astype(f32)
        v[:, i] = (acc + mo[:, i, 3]).astype(f32)
    return v, out


def warp_vertices_backward(matrices, points, skinning_weights, joint_index, grad_vertices, grad_matrix_out):
    """-> (grad_points [Q, 3] f32, grad_matrices [J, 4, 4] f64, |.| sums [J, 4, 4] f64,
    grad_skinning_weights [Q, B] f32)."""
    matrices = np.asarray(matrices, f32)
    p = np.asarray(points, f32)
    sw = np.asarray(skinning_weights, f32)
    ji = np.asarray(joint_index, np.int64)
    gv = np.asarray(grad_vertices, f32)
    gm = np.asarray(grad_matrix_out, f32)[:, :3, :]
    Q, B = sw.shape
    mo = _blend(matrices, sw, ji)
    gp = np.empty_like(p)
    for i in range(3):                                               # :193
        acc = (gv[:, 0] * mo[:, 0, i]).astype(f32)
        acc = (acc + (gv[:, 1] * mo[:, 1, i]).astype(f32)).astype(f32)
        gp[:, i] = (acc + (gv[:, 2] * mo[:, 2, i]).astype(f32)).astype(f32)
    tg = np.empty((Q, 3, 4), f32)                                    # tmp_grad_matrix, :194-197
    tg[:, :, :3] = (gv[:, :, None] * p[:, None, :]).astype(f32)
    tg[:, :, 3] = gv
    gsw = np.zeros((Q, B), f32)
    gmat = np.zeros((matrices.shape[0], 4, 4), np.float64)
    gabs = np.zeros_like(gmat)
    for j in range(B):
        w = sw[:, j]
        use = w > 0
        m = matrices[ji[:, j], :3, :]
        acc = np.zeros(Q, f32)
        for src in (gm, tg):                                         # loop at :176-185, then the one at :200-208
            for a in range(3):
                for b in range(4):
                    acc = (acc + (m[:, a, b] * src[:, a, b]).astype(f32)).astype(f32)
        gsw[:, j] = np.where(use, acc, f32(0))
        for src in (gm, tg):
            contrib = (w[:, None, None] * src).astype(f32).astype(np.float64)
            contrib[~use] = 0.0
            np.add.at(gmat[:, :3, :], ji[:, j], contrib)
            np.add.at(gabs[:, :3, :], ji[:, j], np.abs(contrib))
    return gp, gmat, gabs, gsw
