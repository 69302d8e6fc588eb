"""Second oracle: a vectorised PyTorch (CPU) volume renderer, written
independently of oracle/svoxt_oracle.cpp and differentiable by autograd.

TEST INFRASTRUCTURE ONLY (same rules as oracle.py).  It is the working version
of the `cuda=False` branch the reference asserts away
(svox_t/renderer.py:224-301): same algorithm as the CUDA `trace_ray`
(svox_t/csrc/rt_kernel.cu:222-328), expressed as batched tensor operations over
the set of rays that are still marching.

Two uses:
  * forward cross-check of the C++ oracle (agreement to ~1e-6, not bit for bit:
    exp / sigmoid come from torch here);
  * an *independent derivation of the gradient*: torch.autograd differentiates
    the compositing formula, whereas the reference (and the oracle, and the HIP
    kernel) use the hand-derived two-pass formula of rt_kernel.cu:331-496.

The stepping arithmetic (which leaf a sample falls in, the step length) is done
in float32 with the reference's operation order so that this renderer visits
exactly the same leaves; compositing is done in `dtype` (float64 for gradient
checks).
"""
from __future__ import annotations

import numpy as np
import torch

from . import oracle as O


def _dda_unit(cen, inv):
    tmin = torch.zeros(cen.shape[0], dtype=torch.float32)
    tmax = torch.full((cen.shape[0],), 1e9, dtype=torch.float32)
    for i in range(3):
        t1 = -cen[:, i] * inv[:, i]
        t2 = t1 + inv[:, i]
        tmin = torch.maximum(tmin, torch.minimum(t1, t2))
        tmax = torch.minimum(tmax, torch.maximum(t1, t2))
    return tmin, tmax


def _locate(child, N, pos):
    """Vectorised root->leaf descent (svox_t/csrc/include/common.cuh:63-100).
    Returns flat slot index, leaf-local coords, cube_sz."""
    hi = np.float32(1.0 - 1e-6)
    p = pos.clamp(0.0, float(hi)).clone()
    B = p.shape[0]
    node = torch.zeros(B, dtype=torch.long)
    cube = torch.full((B,), float(N), dtype=torch.float32)
    slot = torch.zeros(B, dtype=torch.long)
    local = torch.zeros_like(p)
    todo = torch.arange(B)
    Nf = np.float32(N)
    while todo.numel():
        q = p[todo] * Nf
        f = torch.floor(q)
        q = q - f
        u = f.long().clamp_(0, N - 1)
        s = ((node[todo] * N + u[:, 0]) * N + u[:, 1]) * N + u[:, 2]
        skip = child[s].long()
        leaf = skip == 0
        done = todo[leaf]
        slot[done] = s[leaf]
        local[done] = q[leaf]
        p[todo] = q
        go = todo[~leaf]
        node[go] = node[go] + skip[~leaf]
        cube[go] = cube[go] * Nf
        todo = go
    return slot, local, cube


def volume_render(tree: O.Tree, origins, dirs, vdirs, opt: O.RenderOptions,
                  features: torch.Tensor = None, dtype=torch.float64):
    """Returns out [Q, C+1] (torch, `dtype`).  If `features` (a torch tensor,
    possibly requiring grad) is given it replaces tree.features in the
    compositing, so `out.backward()` yields d out / d features."""
    f32 = torch.float32
    child = torch.from_numpy(tree.child.reshape(-1))
    data = torch.from_numpy(tree.data.reshape(-1))
    feats32 = torch.from_numpy(tree.features)
    feats = feats32.to(dtype) if features is None else features
    M, K, N = tree.M, tree.K, tree.N
    offset = torch.from_numpy(tree.offset.astype(np.float32))
    scaling = torch.from_numpy(tree.scaling.astype(np.float32))
    o = torch.as_tensor(np.asarray(origins), dtype=f32)
    d = torch.as_tensor(np.asarray(dirs), dtype=f32)
    Q = o.shape[0]

    # ---- per-ray set-up (float32, reference operation order) ----
    o = offset + scaling * o
    d = d * scaling
    nrm = torch.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    delta_scale = 1.0 / nrm
    d = d * delta_scale[:, None]
    inv = (1.0 / (d.double() + 1e-9)).float()
    tmin, tmax = _dda_unit(o, inv)
    hit = ~((tmax < 0) | (tmin > tmax))

    C = O.out_data_dim(opt, K) - 1
    rgba = opt.format == O.FORMAT_RGBA
    if not rgba:
        basis = torch.from_numpy(O.basis(opt.format, opt.basis_dim, vdirs, tree.extra)).to(dtype)
        comp = slice(opt.min_comp, opt.max_comp + 1)

    color = torch.zeros(Q, C, dtype=dtype)
    light = torch.ones(Q, dtype=dtype)
    stopped = torch.zeros(Q, dtype=torch.bool)
    ids = torch.nonzero(hit).squeeze(1)
    t = tmin[ids].clone()
    while ids.numel():
        pos = o[ids] + t[:, None] * d[ids]
        slot, local, cube = _locate(child, N, pos)
        idx = data[slot].long()
        valid = (idx >= 0) & (idx < M)
        s_tmin, s_tmax = _dda_unit(local, inv[ids])
        delta_t = (s_tmax - s_tmin) / cube + np.float32(opt.step_size)
        sigma32 = torch.where(valid, feats32[idx.clamp(0, M - 1), K - 1], torch.zeros((), dtype=f32))
        act = sigma32 > opt.sigma_thresh
        if act.any():
            a_ids = ids[act]
            row = feats[idx[act]]
            sig = row[:, K - 1]
            att = torch.exp(-(delta_t[act] * delta_scale[a_ids]).to(dtype) * sig)
            w = light[a_ids] * (1.0 - att)
            if rgba:
                rgb = torch.sigmoid(row[:, :C])
            else:
                coeff = row[:, :C * opt.basis_dim].reshape(-1, C, opt.basis_dim)
                rgb = torch.sigmoid((coeff[:, :, comp] * basis[a_ids][:, None, comp]).sum(-1))
            color = color.index_add(0, a_ids, w[:, None] * rgb)
            light = light.index_put((a_ids,), light[a_ids] * att)
            if opt.stop_thresh > 0 or True:
                st = light[a_ids].detach() <= opt.stop_thresh
                if st.any():
                    stopped[a_ids[st]] = True
        t = t + delta_t
        keep = (t < tmax[ids]) & ~stopped[ids]
        ids, t = ids[keep], t[keep]

    bg = float(opt.background_brightness)
    scale = torch.where(stopped, 1.0 / (1.0 - light), torch.ones((), dtype=dtype))
    add = torch.where(stopped, torch.zeros((), dtype=dtype), light * bg)
    color = color * scale[:, None] + add[:, None]
    miss = ~hit
    color = torch.where(miss[:, None], torch.full((), bg, dtype=dtype), color)
    alpha = torch.where(miss, torch.zeros((), dtype=dtype), 1.0 - light)
    return torch.cat([color, alpha[:, None]], dim=1)
