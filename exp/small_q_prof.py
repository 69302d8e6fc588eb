import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C
_C.GRAD_SCRATCH = os.environ.get('PROBE_SCRATCH', '1') == '1'
_C.SORT_RAYS_MIN = int(os.environ.get('PROBE_SORT_MIN', '16384'))
dev = torch.device("cuda:0")
Q = int(os.environ.get("PROBE_Q", "4096"))
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = [t.to(dev) for t in synth.pinhole_rays(800, 800)]
idx = torch.randint(0, o.shape[0], (Q,), device=dev)
rays = svox.Rays(o[idx], d[idx], v[idx])
p = feats.clone().requires_grad_(True)
go = torch.ones((Q, 4), device=dev)
for _ in range(50):
    out = r(p, rays); out.backward(go); p.grad = None
torch.cuda.synchronize()
