"""Boundary (b), route B of INTEGRATION.md: the reference's own argument-packing code -- run in the
build container on this repo's spec classes by tests/golden/make_shim_fields.py -- and this repo's
counterparts (svox_t_amd.N3Tree._spec, VolumeRenderer._get_options, _rays_spec_from_rays,
_make_camera_spec) fill the same fields with the same values.  CPU only: packing calls no kernel.

Reference: svox_t/svox.py:899-925, svox_t/renderer.py:44-58, 408-439."""
import json
import os

import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import renderer as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shim_fields.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def dec(x):
    if isinstance(x, dict) and x.get("tensor"):
        return torch.tensor(x["values"], dtype=getattr(torch, x["dtype"])).reshape(x["shape"])
    return x


def same(got, want, name):
    want = dec(want)
    if isinstance(want, torch.Tensor):
        assert isinstance(got, torch.Tensor), name
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), (name, got.dtype, got.shape, want.dtype, want.shape)
        assert torch.equal(got.cpu(), want), name
    elif isinstance(want, float):
        assert float(got) == pytest.approx(want, rel=1e-7), name          # float fields travel as C floats
    else:
        assert got == want, (name, got, want)


@pytest.fixture(scope="module")
def tree(gold):
    s, tw = gold["scenario"], gold["tree"]["world"]
    n = tw["n_internal"]
    child, data, pd = dec(tw["child"]), dec(tw["data"]), dec(tw["parent_depth"])
    t = svox.N3Tree.from_arrays(child, data, pd, dec(gold["features"]), data_format=s["data_format"],
                                radius=s["radius"], center=s["center"])
    t._n_internal.fill_(n)
    t.filled = n                      # the reference's tables carry reserve rows behind the n used ones
    return t


def test_tree_spec_fields(gold, tree):
    feats = dec(gold["features"])
    mi = {k: dec(v) for k, v in gold["motion_inputs"].items()}
    variants = {
        "world": lambda: tree._spec(feats),
        "local": lambda: tree._spec(feats, world=False),
        "motion": lambda: tree._spec(feats, joint_features=mi["joint_features"], skinning_weights=mi["skinning_weights"],
                                     joint_index=mi["joint_index"], transformation_matrices=mi["transformation_matrices"]),
    }
    for name, make in variants.items():
        spec = make()
        assert isinstance(spec, _C.TreeSpec)
        for field, want in gold["tree"][name].items():
            same(getattr(spec, field), want, f"tree.{name}.{field}")
    tree._weight_accum = torch.zeros(tree.child.shape)
    try:
        for field, want in gold["tree"]["weights"].items():
            same(getattr(tree._spec(feats), field), want, f"tree.weights.{field}")
    finally:
        tree._weight_accum = None


def test_render_options_fields(gold, tree):
    s = gold["scenario"]
    r = svox.VolumeRenderer(tree, **s["renderer"])
    for name, opt in (("default", r._get_options()), ("fast", r._get_options(True))):
        assert isinstance(opt, _C.RenderOptions)
        for field, want in gold["options"][name].items():
            same(getattr(opt, field), want, f"options.{name}.{field}")
    r.sigma_thresh = 0.25
    for field, want in gold["options"]["override"].items():
        same(getattr(r._get_options(True), field), want, f"options.override.{field}")
    rn = svox.VolumeRenderer(tree, ndc=svox.NDCConfig(**s["ndc"]))
    for field, want in gold["options"]["ndc"].items():
        same(getattr(rn._get_options(), field), want, f"options.ndc.{field}")
    # ... and the C struct the options are marshalled into keeps them (11 fields, declaration order)
    co = _C._pack_opts(rn._get_options())
    assert [f for f, _ in co._fields_] == list(gold["options"]["ndc"].keys())


def test_rays_and_camera_spec_fields(gold):
    ri = {k: dec(v) for k, v in gold["rays_inputs"].items()}
    spec = R._rays_spec_from_rays(svox.Rays(ri["origins"], ri["dirs"], ri["viewdirs"]))
    assert isinstance(spec, _C.RaysSpec)
    for field, want in gold["rays"].items():
        same(getattr(spec, field), want, f"rays.{field}")
    cam, c2w = gold["scenario"]["camera"], dec(gold["camera_inputs"]["c2w"])
    cs = R._make_camera_spec(c2w, cam["width"], cam["height"], cam["fx"], cam["fy"])
    assert isinstance(cs, _C.CameraSpec)
    for field, want in gold["camera"].items():
        same(getattr(cs, field), want, f"camera.{field}")


def test_shim_presents_what_the_reference_looks_for():
    """helpers._get_c_extension (svox_t/helpers.py:363-376) accepts any module with `query_vertical`;
    the autograd functions call these names (renderer.py:60-138, svox.py:38-76)."""
    for name in ("query_vertical", "query_vertical_backward", "volume_render", "volume_render_backward",
                 "volume_render_image", "volume_render_image_backward", "render_depth", "opacity_render",
                 "opacity_render_backward", "motion_render", "motion_feature_render",
                 "motion_feature_render_backward", "warp_vertices", "warp_vertices_backward", "construct_tree",
                 "TreeSpec", "RaysSpec", "CameraSpec", "RenderOptions"):
        assert hasattr(_C, name), name
