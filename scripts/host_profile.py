#!/usr/bin/env python3
"""GPU box: where the HOST time of one forward+backward step goes (cProfile over 200 steps)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
from svox_t_amd import synth
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, synth.shell_features(st.n_features, 28), data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(800, 800)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
g = synth.grad_output(640000, 4).to(dev)
f = tree.features
def step():
    f.grad = None
    r(f, rays, image_shape=(800, 800)).backward(g)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()                 # host done enqueueing
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/200:.3f} ms/step, with GPU drain {1e3*(t2-t0)/200:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
