"""Report how far the HIP gradient sits from the oracle's, as a fraction of the
1e-5 * S tolerance (NOTEBOOK.md 4).  Test tooling: uses oracle/.

    python tests/tools/grad_margin.py        # on the GPU box, from the repo root
"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import svox_t_amd as svox
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case
dev = torch.device("cuda:0")
for name, kw in (("d8_sh9_800", dict(depth=8, K=28, data_format="SH9", width=800, height=800)),
                 ("d6_rgba32", dict(depth=6, K=32, data_format="RGBA", width=96, height=96)),
                 ("d5_rgba4", dict(depth=5, K=4, data_format="RGBA", width=64, height=64))):
    c = Case(**kw)
    tree = c.tree(dev); r = svox.VolumeRenderer(tree)
    for trial in range(3):
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(dev), image_shape=(kw["height"], kw["width"]))
        g = synth.grad_output(c.Q, out.shape[1], seed=trial)
        out.backward(g.to(dev))
        got = tree.features.grad.cpu().numpy().astype(np.float64)
        want, S = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
        err = np.abs(got - want); ratio = err / (1e-5 * S + 1e-30)
        sig = ratio[:, -1].max(); col = ratio[:, :-1].max()
        print(name, trial, "worst ratio to the 1e-5*S bound: sigma col %.3f, colour cols %.3f" % (sig, col))
