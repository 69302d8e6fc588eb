"""The CHECKED instances of the per-tile backwards (svoxt_set_bwd_check, ABI v17).

Round 3 saw one GPU memory fault in grad_fused_kernel<SH, 9, single march> after a refactoring that had turned its
sort / reduce phase into a function.  The ISA of that build (exp/fault_r03_recreate.sh, profiles/r04_fault_isa.txt)
shows 42 scratch instructions, every one a compile-time-offset register spill: no private array went to memory, so no
private index could have left its array.  What remains possible on the kernel's side is an LDS / pool / table index
out of range that only some geometry produces (the 768 / 896 record-slot overrun of r03 was of that kind and showed
at full size only).  These instances compare EVERY such index with its extent before it is used and count violations
per site; this test runs them ONCE over the full-size geometries and asserts zero."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from tests.util import Case

pytestmark = pytest.mark.gpu


def _run(c, gpu, shape):
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    gout = synth.grad_output(c.Q, 4 if c.K != 32 else 32).to(gpu)
    out = r(tree.features, c.rays_gpu(gpu), image_shape=shape)
    out.backward(gout)
    return tree.features.grad.clone(), _C.LAST_ROUTE["backward"]


CASES = [
    # (name, case arguments, image shape, module attributes to set, expected kernel)
    ("cfg3 SH9, forward's hand-over (the default route)", dict(depth=8, K=28, data_format="SH9", width=800, height=800),
     (800, 800), {}, "grad_fused_kernel<EXACT>"),
    ("cfg3 SH9, single march (the instance that faulted in r03)", dict(depth=8, K=28, data_format="SH9", width=800, height=800),
     (800, 800), {"BWD_EXACT": False}, "grad_fused_kernel (one sweep"),
    ("cfg3 SH9, one-kernel recording forward (lane-major hand-over)", dict(depth=8, K=28, data_format="SH9", width=800, height=800),
     (800, 800), {"FWD_SPLIT": "0"}, "grad_fused_kernel<EXACT>"),
    ("cfg3 SH9, no hand-over (both sweeps gather the rows)", dict(depth=8, K=28, data_format="SH9", width=800, height=800),
     (800, 800), {"BWD_TERMS": False}, "grad_fused_kernel<EXACT>"),
    ("SH16 full size", dict(depth=8, K=49, data_format="SH16", width=800, height=800), (800, 800), {}, "grad_fused_kernel<EXACT>"),
    ("SH25 full size (passes of 896 records)", dict(depth=8, K=76, data_format="SH25", width=800, height=800),
     (800, 800), {}, "grad_fused_kernel<EXACT>"),
    ("RGBA rows of 4 floats", dict(depth=8, K=4, data_format="RGBA", width=800, height=800), (800, 800), {}, "grad_fused_kernel<EXACT>"),
    ("cfg4: depth 9, rows of 32 floats, 1024 x 1024", dict(depth=9, K=32, data_format="RGBA", width=1024, height=1024),
     (1024, 1024), {}, "grad_wide_kernel"),
    ("a ragged image (tiles cut by the border), depth 6 SH4", dict(depth=6, K=13, data_format="SH4", width=200, height=136),
     (136, 200), {}, "grad_fused_kernel<EXACT>"),
]


def test_checked_instances_find_no_index_out_of_range_at_full_size(gpu, monkeypatch, capsys):
    report = []
    for name, kw, shape, attrs, expect in CASES:
        assert all(hasattr(_C, k) for k in attrs)
        c = Case(**kw)
        with monkeypatch.context() as m:
            for k, v in attrs.items():
                m.setattr(_C, k, v)
            want, route = _run(c, gpu, shape)                      # the production instance
            assert route.startswith(expect), (name, route)
            with _C.bwd_check(gpu) as chk:
                got, route2 = _run(c, gpu, shape)
                torch.cuda.synchronize()
            bad, tiles = chk.read()
        assert route2 == route
        report.append((name, tiles, bad))
        assert tiles > 0, f"{name}: no checked instance ran"
        assert bad == {}, f"{name}: index violations per site {bad}"
        # the checked instance does the production instance's work (float-atomic order aside)
        scale = want.abs().max().item()
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() <= 1e-4 * scale, name
    with capsys.disabled():
        for name, tiles, bad in report:
            print(f"\n[checked backward] {name}: {tiles} tiles, violations {bad or 'none'}")
