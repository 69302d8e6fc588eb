#!/usr/bin/env python3
"""Drop-in evidence for boundary (b), INTEGRATION.md route B: install this repo's operator shim
(`svox_t_amd.csrc`) under the name the reference looks for (`svox_t.csrc`,
svox_t/helpers.py:363-376), import the REFERENCE package on top of it, and let the reference's own
argument-packing code -- N3Tree._spec (svox.py:899-925), VolumeRenderer._get_options
(renderer.py:408-439), _rays_spec_from_rays / _make_camera_spec (renderer.py:44-58) -- fill this
repo's spec classes.  The resulting field values go to tests/golden/shim_fields.json; the CPU test
tests/test_shim_fields.py asserts that this repo's own counterparts produce the same fields.

    python tests/golden/make_shim_fields.py        # this container only: needs /root/reference

Only data is stored (field names, dtypes, shapes, values): nothing of the reference's source.
No GPU is needed: packing specs calls no kernel.
"""
import json
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")

import svox_t_amd.csrc as shim                      # noqa: E402
sys.modules["svox_t.csrc"] = shim                   # what `import svox_t.csrc` will find
import svox_t as ref                                # noqa: E402  (the reference)
import svox_t.renderer as ref_renderer              # noqa: E402

assert ref_renderer._C is shim, "the reference did not pick up the shim"

# the scenario, shared with tests/test_shim_fields.py through the json
SCEN = dict(N=2, data_dim=13, data_format="SH4", radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3],
            init_reserve=64, n_rays=5, seed=3,
            renderer=dict(step_size=2e-3, background_brightness=0.5, min_comp=1, max_comp=-1),
            ndc=dict(width=640, height=480, focal=500.0),
            camera=dict(width=80, height=60, fx=111.5, fy=112.5))


def enc(x):
    if isinstance(x, torch.Tensor):
        return {"tensor": True, "dtype": str(x.dtype).replace("torch.", ""), "shape": list(x.shape),
                "values": x.detach().reshape(-1).tolist()}
    if isinstance(x, (int, float, bool)) or x is None:
        return x
    raise TypeError(type(x))


def fields(obj, names):
    return {n: enc(getattr(obj, n)) for n in names}


TREE_FIELDS = ["features", "data", "child", "parent_depth", "extra_data", "offset", "scaling", "_weight_accum",
               "joint_features", "skinning_weights", "joint_index", "n_internal", "transformation_matrices"]
OPT_FIELDS = ["step_size", "background_brightness", "format", "basis_dim", "ndc_width", "ndc_height", "ndc_focal",
              "min_comp", "max_comp", "sigma_thresh", "stop_thresh"]
RAY_FIELDS = ["origins", "dirs", "vdirs"]
CAM_FIELDS = ["c2w", "fx", "fy", "width", "height"]


def main():
    g = torch.Generator().manual_seed(SCEN["seed"])
    t = ref.N3Tree(N=SCEN["N"], data_dim=SCEN["data_dim"], init_reserve=SCEN["init_reserve"],
                   radius=SCEN["radius"], center=SCEN["center"], data_format=SCEN["data_format"])
    t.refine(1)                                       # 9 nodes: root + 8 children (CPU; no resize with this reserve)
    feats = torch.randn(7, SCEN["data_dim"], generator=g)
    out = {"scenario": SCEN, "features": enc(feats), "tree": {}, "options": {}, "rays": {}, "camera": {}}
    out["tree"]["world"] = fields(t._spec(feats), TREE_FIELDS)
    out["tree"]["local"] = fields(t._spec(feats, world=False), TREE_FIELDS)
    jf, sw = torch.randn(4, 6, generator=g), torch.rand(7, 2, generator=g)
    ji = torch.randint(0, 4, (7, 2), generator=g).int()
    xf = torch.randn(7, 3, 3, generator=g)
    out["motion_inputs"] = dict(joint_features=enc(jf), skinning_weights=enc(sw), joint_index=enc(ji),
                                transformation_matrices=enc(xf))
    out["tree"]["motion"] = fields(t._spec(feats, joint_features=jf, skinning_weights=sw, joint_index=ji,
                                           transformation_matrices=xf), TREE_FIELDS)
    t._weight_accum = torch.zeros(t.child.shape)
    out["tree"]["weights"] = fields(t._spec(feats), TREE_FIELDS)
    t._weight_accum = None

    r = ref.VolumeRenderer(t, **SCEN["renderer"])
    out["options"]["default"] = fields(r._get_options(), OPT_FIELDS)
    out["options"]["fast"] = fields(r._get_options(True), OPT_FIELDS)
    r.sigma_thresh = 0.25                              # the override attribute (renderer.py:434-438)
    out["options"]["override"] = fields(r._get_options(True), OPT_FIELDS)
    rn = ref.VolumeRenderer(t, ndc=ref.NDCConfig(**SCEN["ndc"]))
    out["options"]["ndc"] = fields(rn._get_options(), OPT_FIELDS)

    o, d = torch.randn(SCEN["n_rays"], 3, generator=g), torch.randn(SCEN["n_rays"], 3, generator=g)
    rays = ref.Rays(o, d, torch.nn.functional.normalize(d, dim=-1))
    out["rays_inputs"] = dict(origins=enc(rays.origins), dirs=enc(rays.dirs), viewdirs=enc(rays.viewdirs))
    out["rays"] = fields(ref_renderer._rays_spec_from_rays(rays), RAY_FIELDS)
    c2w = torch.eye(4)[:3] + 0.1 * torch.randn(3, 4, generator=g)
    cam = SCEN["camera"]
    out["camera_inputs"] = dict(c2w=enc(c2w))
    out["camera"] = fields(ref_renderer._make_camera_spec(c2w, cam["width"], cam["height"], cam["fx"], cam["fy"]), CAM_FIELDS)

    path = os.path.join(HERE, "shim_fields.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
