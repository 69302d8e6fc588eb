"""Run by tests/test_gpu_nccl_world1.py as a child process on the GPU box: the "nccl" (= RCCL) branch of
svox_t_amd/parallel.py and bench.py's process-group set-up, in a group of ONE rank on cuda:0, with
parallel.FORCE_COLLECTIVES so that every collective is really issued to RCCL (each is the identity at one rank, so the
expected results are the single-process ones).  What this catches before the first 8-GPU run: API misuse
(init_process_group(device_id=...), all_gather_into_tensor / all_reduce(async_op=True) argument forms, work handles),
the side-stream hand-offs of OverlappedGradReducer, stream ordering between the kernels and the collectives.
What it cannot: point-to-point traffic (a rank may not send to itself; direct_all_reduce / sparse_all_reduce run with
an empty peer list here -- their exchange logic is covered by the world-2 / world-3 gloo tests on CPU) and anything
about xGMI.  Prints one JSON line."""
import datetime
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29617")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)   # bench.py's call
    import svox_t_amd as svox
    from svox_t_amd import parallel, synth
    parallel.FORCE_COLLECTIVES = True
    res = {}

    st = synth.shell_tree(6)
    K = 28
    feats = synth.shell_features(st.n_features, K)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
    parallel.broadcast_tree(tree)                          # dist.broadcast x 6 on device tensors
    renderer = svox.VolumeRenderer(tree)
    W, H = 128, 96
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    gout = synth.grad_output(W * H, 4).to(dev)

    # single-process reference
    f0 = tree.features.detach().clone().requires_grad_(True)
    want = renderer(f0, rays, image_shape=(H, W))
    want.backward(gout)
    torch.cuda.synchronize()
    gwant = f0.grad.clone()

    # 1. render_sharded: all_gather_into_tensor forward, all_reduce backward
    f1 = tree.features.detach().clone().requires_grad_(True)
    full = parallel.render_sharded(renderer, f1, rays, image_shape=(H, W))
    (full * gout).sum().backward()
    torch.cuda.synchronize()
    res["render_sharded_out_equal"] = bool(torch.equal(full.detach(), want.detach()))
    res["render_sharded_grad_maxdiff"] = float((f1.grad - gwant).abs().max() / gwant.abs().max())

    # 2. OverlappedGradReducer, every mode: the side stream waits for the backward, the compute stream for the reducer
    for mode in ("all_reduce", "direct", "touched"):
        red = parallel.OverlappedGradReducer(dist, backend="nccl", mode=mode, chunk_bytes=1 << 20)
        f2 = tree.features.detach().clone().requires_grad_(True)
        acc = torch.zeros_like(gwant)
        for _ in range(3):                                   # gradient accumulation: one gradient in flight under the next step
            f2.grad = None
            out = renderer(f2, rays, image_shape=(H, W))
            out.backward(gout)
            red.start(f2.grad)
            g = red.wait()
            acc += g
        torch.cuda.synchronize()
        res[f"reducer_{mode}_maxdiff"] = float((acc / 3 - gwant).abs().max() / gwant.abs().max())
        res[f"reducer_{mode}_chunks"] = len(red.chunks(gwant))

    # 3. direct_all_reduce / sparse_all_reduce called directly (empty peer lists; the host-side mask gather runs)
    g3 = gwant.clone()
    parallel.direct_all_reduce(dist, g3)
    g4 = gwant.clone()
    _, stats = parallel.sparse_all_reduce(dist, g4)
    torch.cuda.synchronize()
    res["direct_equal"] = bool(torch.equal(g3, gwant))
    res["sparse_equal"] = bool(torch.equal(g4, gwant))
    res["sparse_stats"] = stats

    # 4. gather_pixels_async under a backward: all_gather_into_tensor(async_op=True) and gather(dst=0, async_op=True)
    f5 = tree.features.detach().clone().requires_grad_(True)
    out = renderer(f5, rays, image_shape=(H, W))
    gathered = torch.empty_like(out.detach())
    h = parallel.gather_pixels_async(dist, gathered, out.detach(), backend="nccl")
    out.backward(gout)
    h.wait()
    torch.cuda.synchronize()
    res["gather_all_equal"] = bool(torch.equal(gathered, want.detach()))
    gathered.zero_()
    h = parallel.gather_pixels_async(dist, gathered, out.detach(), backend="nccl", dst=0)
    h.wait()
    torch.cuda.synchronize()
    res["gather_dst_equal"] = bool(torch.equal(gathered, want.detach()))

    # 5. render_cameras: image mode, broadcast of the shape, all_gather_into_tensor, all_reduce
    poses = torch.stack([torch.from_numpy(synth.camera_pose(azimuth_deg=30.0 + 45.0 * k)).float() for k in range(2)]).to(dev)
    f6 = tree.features.detach().clone().requires_grad_(True)
    ims = parallel.render_cameras(renderer, f6, poses, width=W, height=H, fx=1111.111 * W / 800.0)
    ims.sum().backward()
    f7 = tree.features.detach().clone().requires_grad_(True)
    ref = torch.stack([renderer.render_persp(f7, poses[k], width=W, height=H, fx=1111.111 * W / 800.0) for k in range(2)])
    ref.sum().backward()
    torch.cuda.synchronize()
    res["cameras_out_equal"] = bool(torch.equal(ims.detach(), ref.detach()))
    res["cameras_grad_maxdiff"] = float((f6.grad - f7.grad).abs().max() / f7.grad.abs().max())

    # 6. the timing collectives of bench.py: barrier + all_reduce(MAX) of a float64 on the device
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["max_reduce"] = float(t.item())
    res["backend"] = dist.get_backend()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
