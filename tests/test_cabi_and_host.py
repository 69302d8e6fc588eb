"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/svoxt.h declares; argument validation (which runs before any HIP call)
returns the documented codes; the Python operator layer rejects what the
reference's TORCH_CHECKs reject.  No compute, no GPU."""
import ctypes
import os
import re

import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "svoxt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svoxt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_C.EXPORTS)          # the ctypes table and the header agree
    assert lib.svoxt_abi_version() == _C.ABI_VERSION == 22


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors have the sizes and field offsets gcc gives the structs of include/svoxt.h."""
    import subprocess
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "svoxt.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(svoxt_options), sizeof(svoxt_rays),'
                   ' sizeof(svoxt_tree), sizeof(svoxt_sample_lists), offsetof(svoxt_rays, c2w), offsetof(svoxt_rays, fy),'
                   ' offsetof(svoxt_tree, accel_log2)); printf("%zu %zu\\n", offsetof(svoxt_tree, xform_dim),'
                   ' sizeof(svoxt_motion)); printf(" %zu %zu %zu\\n", offsetof(svoxt_sample_lists, coef_bytes),'
                   ' offsetof(svoxt_tree, sigma_mask_thresh), offsetof(svoxt_sample_lists, flags)); return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(_C._COptions), ctypes.sizeof(_C._CRays), ctypes.sizeof(_C._CTree),
                   ctypes.sizeof(_C._CLists), _C._CRays.c2w.offset, _C._CRays.fy.offset, _C._CTree.accel_log2.offset,
                   _C._CTree.xform_dim.offset, ctypes.sizeof(_C._CMotion), _C._CLists.coef_bytes.offset,
                   _C._CTree.sigma_mask_thresh.offset, _C._CLists.flags.offset]
    assert got[:3] == [44, 64, 136]                     # (svoxt_tree grew by exp_table in ABI v17)


def test_no_kernel_keeps_private_arrays_in_scratch_memory():
    """What the compiler reported for every kernel of the last build (svox_t_amd/build.py writes the
    -Rpass-analysis=kernel-resource-usage remarks next to the library).  No kernel may have a dynamic
    stack, and none more than a few spilled registers' worth of scratch: a per-lane array (feature
    row, basis, accumulators) that lands in scratch memory costs a memory round trip per access
    (NOTEBOOK.md step 13: forward 0.33 -> 0.28 ms when 12 bytes left it) -- and scratch is the only place
    where an out-of-range private index could fault instead of reading another register.  The one
    unexplained abort in this project's records (r02, gpurun_out/r2q/t5.log) was an EXPERIMENT build
    that capped shade_tile_kernel at 64 registers: its SH16 instance, the first kernel of that test
    run with a large spill (152 bytes per lane), is what was running (NOTEBOOK.md 4.1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_svoxt_build", os.path.join(ROOT, "svox_t_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    if not os.path.exists(b.RESOURCES_PATH):
        pytest.skip("library built without resource remarks (an older build.py)")
    res = b.kernel_resources()
    assert len(res) > 200                         # every instance of every kernel template
    worst = {}
    for name, r in res.items():
        assert r["Dynamic Stack"] == "False", name
        sc = int(r["ScratchSize [bytes/lane]"])
        if sc:
            worst[name] = sc
    # today: grad_wide_kernel (12 bytes in the exact instances at their 80-register budget, 32-40 in the
    # native-math ones at the 64 registers that let four workgroups share a CU: loop-invariant LDS addresses and
    # pointers, written once in front of a sweep) and the INSTRUMENTATION instances of grad_fused_kernel<SH9>
    # (counting: 16-20 bytes; checked, r04: 8).  Nothing above 48 bytes, as a matter of speed (a spill that appears
    # is a regression to remove) -- NOT because larger frames are known to be unsafe: the r03 fault came with the
    # first dispatch of an 88-byte frame (grad_fused_kernel<SH9> when its sort / reduce phase was a function), and
    # the ISA of that build, recreated in r04 (exp/fault_r03_recreate.sh, profiles/r04_fault_isa.txt), holds only
    # compile-time-offset register spills, stored under a full EXEC mask: no private array in memory, no index that
    # could leave it; the checked instances (tests/test_gpu_checked_backward.py) find no LDS / pool / table index
    # out of range at any full-size geometry either (NOTEBOOK.md 4.1).
    # (r04: the table instances of grad_wide_kernel are compiled for 64 registers -- four workgroups per CU: backward
    # 1.70 -> 1.59 ms at depth 9 -- and keep 36-52 bytes of loop invariants in scratch for it)
    assert all(v <= 64 for v in worst.values()), worst
    import re

    def instrumentation(k):
        m = re.search(r"grad_fused_kernelILi1ELi9ELb[01]ELb([01])ELi\dELb[01]ELb([01])E", k)
        return m is not None and "1" in m.groups()

    def rotations_over_handover(k):       # (r04) grad_fused_kernel<SH, 9, ..., TERMS 3, ..., XF>: 12 bytes (a row AND the hand-over in registers)
        return re.search(r"grad_fused_kernelILi1ELi9ELb1ELb0ELi3ELb0ELb0ELb1E", k) is not None and worst[k] <= 16
    assert all(("grad_wide_kernel" in k) or instrumentation(k) or rotations_over_handover(k) for k in worst), worst


def test_out_data_dim():
    o = _C.RenderOptions()
    o.format, o.basis_dim = 0, -1
    assert _C.get_out_data_dim(o, 4) == 4 and _C.get_out_data_dim(o, 32) == 32
    o.format, o.basis_dim = 1, 9
    assert _C.get_out_data_dim(o, 28) == 4
    o.basis_dim = 0
    with pytest.raises(RuntimeError):
        _C.get_out_data_dim(o, 28)


def test_validation_codes_without_touching_the_gpu():
    lib = _C._lib
    assert lib.svoxt_volume_render_fwd(None, None, None, None, None) == 1      # SVOXT_ERR_INVALID
    assert b"tree is NULL" in lib.svoxt_last_error()
    t = _C._CTree()
    assert lib.svoxt_render_depth(ctypes.byref(t), None, None, None, None) == 1
    assert b"NULL" in lib.svoxt_last_error()
    buf = (ctypes.c_float * 96)()
    p = ctypes.c_void_p((ctypes.addressof(buf) + 63) & ~63)        # list records must sit on 64-byte lines
    t = _C._CTree(features=p, M=1, K=13, N=3, data=p, child=p, n_internal=1, offset=p, scaling=p, xform=p, xform_dim=3)
    # sample lists combine with per-leaf view rotations only for SH payloads on N = 2 trees
    o = _C._COptions(format=1, basis_dim=4, min_comp=0, max_comp=3)
    r = _C._CRays(Q=0)
    l = _C._CLists(rec=p, aux=p, max_samples=8)
    assert lib.svoxt_volume_render_fwd_record(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None,
                                              ctypes.byref(l), None) == 2       # SVOXT_ERR_UNSUPPORTED
    assert lib.svoxt_can_record(ctypes.byref(t), ctypes.byref(o)) == 0
    t.N = 2
    assert lib.svoxt_can_record(ctypes.byref(t), ctypes.byref(o)) == 1
    assert lib.svoxt_volume_render_fwd_record(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None,
                                              ctypes.byref(l), None) == 0       # empty batch: accepted
    t.xform_dim = 5
    assert lib.svoxt_volume_render_fwd(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None, None) == 1
    assert b"xform_dim" in lib.svoxt_last_error()
    t.xform, t.xform_dim = None, 0
    assert lib.svoxt_can_record(ctypes.byref(t), ctypes.byref(o)) == 1
    # SG / ASG payloads with an SH-sized lobe count: lists only together with the per-tile backward's hand-over (2)
    o2 = _C._COptions(format=2, basis_dim=4, min_comp=0, max_comp=3)
    t2 = _C._CTree(features=p, M=1, K=13, N=2, data=p, child=p, n_internal=1, offset=p, scaling=p, extra=p, extra_rows=4, extra_cols=4)
    assert lib.svoxt_can_record(ctypes.byref(t2), ctypes.byref(o2)) == 2
    assert lib.svoxt_fwd_fills_terms(ctypes.byref(t2), ctypes.byref(o2), 0) == 3
    t2.N = 3
    assert lib.svoxt_can_record(ctypes.byref(t2), ctypes.byref(o2)) == 0
    t2.N, o2.basis_dim, o2.max_comp, t2.K, t2.extra_rows = 2, 6, 5, 19, 6     # six lobes: the generic kernels, no lists
    assert lib.svoxt_can_record(ctypes.byref(t2), ctypes.byref(o2)) == 0
    # the table-editing entry points validate before launching
    assert lib.svoxt_refine(None, 3, 2, 1, 2, None, None, None, None, None) == 1
    assert b"capacity" in lib.svoxt_last_error()
    assert lib.svoxt_build_workspace_bytes(0) == -1 and lib.svoxt_build_workspace_bytes(8) > 0
    assert lib.svoxt_motion_workspace_bytes(10, 33) == -1
    t.K = 4
    r = _C._CRays(Q=0)
    o = _C._COptions(format=1, basis_dim=7)
    assert lib.svoxt_volume_render_fwd(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None, None) == 1
    assert b"basis_dim" in lib.svoxt_last_error()
    # Q == 0 is a valid no-op (empty ray batch)
    o = _C._COptions(format=0, basis_dim=-1)
    assert lib.svoxt_volume_render_fwd(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None, None) == 0
    # a ray order (svoxt_rays.order) and the image hint are two answers to the same question; camera mode has neither
    r = _C._CRays(Q=64, origins=p, dirs=p, vdirs=p, image_width=8, image_height=8, order=p)
    assert lib.svoxt_volume_render_fwd(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None, None) == 1
    assert b"rays.order" in lib.svoxt_last_error()
    r = _C._CRays(Q=64, image_width=8, image_height=8, c2w=p, fx=1.0, fy=1.0, order=p)
    assert lib.svoxt_render_depth(ctypes.byref(t), ctypes.byref(r), ctypes.byref(o), None, None) == 1
    assert b"camera mode" in lib.svoxt_last_error()
    # the helpers that go with svoxt_ray_order
    assert lib.svoxt_permute_rows(None, None, None, 4, 3, 0, None) == 1
    assert lib.svoxt_permute_rows(None, None, None, 0, 3, 0, None) == 0          # nothing to move
    assert lib.svoxt_gather_rays(ctypes.byref(r), None, None, None, None, None) == 1
    assert lib.svoxt_sigma_mask_bytes(65) == 16 and lib.svoxt_sigma_mask_bytes(-1) == -1


def test_operator_layer_rejects_cpu_and_noncontiguous_tensors():
    tree = svox.N3Tree(N=2, data_dim=4, init_reserve=4)
    r = svox.VolumeRenderer(tree)
    rays = svox.Rays(torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        r(tree.features, rays)                         # tree not on a GPU
    with pytest.raises(RuntimeError, match="GPU"):
        r(tree.features, rays, cuda=False)             # the reference asserts here
    spec = tree._spec(tree.features)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _C.volume_render(spec, svox.renderer._rays_spec_from_rays(rays), r._get_options())
    with pytest.raises(RuntimeError, match="GPU"):
        tree(tree.features, torch.zeros(2, 3))
    with pytest.raises(NotImplementedError):
        _C.grid_weight_render(None, None, None)


def test_spec_and_options_field_mapping():
    """N3Tree._spec (svox.py:899-925) and VolumeRenderer._get_options
    (renderer.py:408-439): defaults, `fast`, overrides, max_comp wrap, world=False."""
    tree = svox.N3Tree(N=2, data_dim=28, init_reserve=4, data_format="SH9",
                       radius=[1.0, 2.0, 4.0], center=[1.0, 0.0, 0.0])
    r = svox.VolumeRenderer(tree)
    o = r._get_options()
    assert (o.step_size, o.background_brightness, o.format, o.basis_dim) == (1e-3, 1.0, 1, 9)
    assert (o.min_comp, o.max_comp, o.ndc_width, o.sigma_thresh, o.stop_thresh) == (0, 8, -1, 0.0, 0.0)
    o = r._get_options(fast=True)
    assert (o.sigma_thresh, o.stop_thresh) == (1e-2, 1e-2)
    r.sigma_thresh = 0.5
    assert r._get_options(fast=True).sigma_thresh == 0.5 and r._get_options().stop_thresh == 0.0
    r2 = svox.VolumeRenderer(tree, ndc=svox.NDCConfig(640, 480, 500.0), min_comp=1, max_comp=3)
    o = r2._get_options()
    assert (o.ndc_width, o.ndc_height, o.ndc_focal, o.min_comp, o.max_comp) == (640, 480, 500.0, 1, 3)
    s = tree._spec(tree.features)
    assert s.features is tree.features and s.n_internal == 1
    assert s.offset is tree.offset and s.scaling is tree.invradius
    assert s.extra_data.shape == (0, 0) and s.transformation_matrices.shape == (0, 0, 0)
    assert s._weight_accum.numel() == 0
    s = tree._spec(tree.features, world=False)
    assert s.offset.tolist() == [0, 0, 0] and s.scaling.tolist() == [1, 1, 1]
    with tree.accumulate_weights() as acc:
        assert tree._spec(tree.features)._weight_accum is acc.value
        with pytest.raises(RuntimeError, match="locked"):
            tree.refine(1)
    assert tree._weight_accum is None
