"""world_size-2 gloo test of the ray-sharding / gather / gradient all-reduce
logic (svox_t_amd/parallel.py) on CPU, with the PyTorch oracle renderer as the
injected render function."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from svox_t_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_image_shard_bounds_are_bands_of_eight_rows():
    for H, W, world in ((800, 800, 8), (64, 40, 3), (24, 16, 2), (20, 16, 2), (16, 16, 4)):
        b = [parallel.image_shard_bounds(H, W, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == H * W and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        if H % 8 == 0 and W % 8 == 0 and H // 8 >= world:
            assert all(lo % (8 * W) == 0 and hi % (8 * W) == 0 for lo, hi in b)


def test_shard_bounds_cover_exactly_once():
    for Q in (0, 1, 7, 64, 640000, 640001):
        for W in (1, 2, 3, 8):
            b = [parallel.shard_bounds(Q, W, r) for r in range(W)]
            assert b[0][0] == 0 and b[-1][1] == Q
            assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, Q, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import svox_t_amd as svox
        from oracle import oracle as O
        from oracle import torch_renderer as TR
        from svox_t_amd import synth
        from tests.util import Case
        c = Case(depth=4, K=4, data_format="RGBA", width=Q, height=1)   # Q rays
        ot, opt = c.oracle_tree(), c.oracle_opts()
        o, d, v = (torch.cat([t] * 1) for t in (c.origins, c.dirs, c.vdirs))
        # a less degenerate batch: one image row is all similar rays; jitter them
        g = torch.Generator().manual_seed(0)
        d = d + 0.2 * torch.randn(d.shape, generator=g)
        rays = svox.Rays(o, d, d.clone())

        def render_fn(features, r):
            return TR.volume_render(ot, r.origins.numpy(), r.dirs.numpy(), r.viewdirs.numpy(), opt,
                                    features=features)

        feats = torch.from_numpy(ot.features).double().requires_grad_(True)
        full = parallel.render_sharded(render_fn, feats, rays, image_shape=(16, Q // 16) if Q % 128 == 0 else None)
        gout = synth.grad_output(Q, 4).double()
        (full * gout).sum().backward()
        # single-process reference
        f2 = torch.from_numpy(ot.features).double().requires_grad_(True)
        ref = render_fn(f2, rays)
        (ref * gout).sum().backward()
        np.save(os.path.join(result_dir, f"ok{rank}.npy"), np.array([
            float((full - ref).abs().max()), float((feats.grad - f2.grad).abs().max()),
            float(f2.grad.abs().max()), full.shape[0]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Q", [64, 37, 128])     # even split, ragged split, image bands of 8 rows
def test_render_sharded_two_ranks(tmp_path, Q):
    port = 29500 + (os.getpid() + Q) % 2000
    mp.spawn(_worker, args=(2, port, Q, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        dout, dgrad, gmax, n = np.load(tmp_path / f"ok{r}.npy")
        assert n == Q
        assert dout == 0.0                                  # gathered image == single-process image
        assert gmax > 0 and dgrad <= 1e-12 * max(gmax, 1.0) # all-reduced gradient == single-process gradient


def _camera_worker(rank, world, port, n_cam, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from oracle import torch_renderer as TR
        from svox_t_amd import synth
        from tests.util import Case
        c = Case(depth=4, K=4, data_format="RGBA", width=8, height=8)
        ot, opt = c.oracle_tree(), c.oracle_opts()
        W, H, fx = 12, 10, 16.0
        # n_cam == 8: the cameras of BASELINE configs[4] / SURVEY.md 8(d) config 5 (azimuth 30 + 45 k degrees),
        # four per rank here (camera i on rank i % 2), one per rank on eight GPUs: the same gather layout
        az = (lambda k: 30.0 + 45.0 * k) if n_cam == 8 else (lambda k: 25.0 + 50.0 * k)
        poses = torch.stack([torch.from_numpy(synth.camera_pose(azimuth_deg=az(k)).astype(np.float32))
                             for k in range(n_cam)])

        def render_fn(features, c2w):        # the oracle's camera rays + the PyTorch oracle renderer
            o, d, v = O.camera_rays(c2w.numpy(), fx, fx, W, H)
            return TR.volume_render(ot, o, d, v, opt, features=features).reshape(H, W, -1)

        feats = torch.from_numpy(ot.features).double().requires_grad_(True)
        full = parallel.render_cameras(render_fn, feats, poses)
        gout = synth.grad_output(n_cam * H * W, 4).double().reshape(n_cam, H, W, 4)
        (full * gout).sum().backward()
        f2 = torch.from_numpy(ot.features).double().requires_grad_(True)
        ref = torch.stack([render_fn(f2, poses[k]) for k in range(n_cam)])
        (ref * gout).sum().backward()
        # every camera in ITS slot of the gathered [n_cam, H, W, C+1] (a permuted layout has equal max-norm overall)
        slot_ok = all(torch.equal(full[k], ref[k]) for k in range(n_cam))
        distinct = n_cam < 2 or not torch.equal(ref[0], ref[1])
        np.save(os.path.join(result_dir, f"cam{rank}.npy"), np.array([
            float((full - ref).abs().max()), float((feats.grad - f2.grad).abs().max()),
            float(f2.grad.abs().max()), full.shape[0], float(ref[..., 3].max()), float(slot_ok and distinct)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cam", [2, 3, 1, 8])   # one per rank, a ragged split, fewer cameras than ranks, config 5's eight
def test_render_cameras_two_ranks(tmp_path, n_cam):
    port = 31500 + (os.getpid() + n_cam) % 2000
    mp.spawn(_camera_worker, args=(2, port, n_cam, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        dout, dgrad, gmax, n, amax, slots = np.load(tmp_path / f"cam{r}.npy")
        assert n == n_cam and amax > 0.1 and slots == 1.0
        assert dout == 0.0
        assert gmax > 0 and dgrad <= 1e-12 * max(gmax, 1.0)


def _reducer_worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        red = parallel.OverlappedGradReducer(dist, backend="gloo", chunk_bytes=4096)
        outs = []
        for step in range(3):                                   # three batches of an accumulation step
            grad = torch.randn(1000 + step, 28, generator=g)
            assert len(red.chunks(grad)) > 1 and red.chunks(grad)[0] == (0, 4096 // (28 * 4))
            mine = grad.clone()
            red.start(grad)                                     # waits for the previous one, sends this one off
            work_between = mine.sum()                           # (the next batch's forward would run here)
            outs.append((mine, grad, work_between))
        last = red.wait()
        assert last is outs[-1][1] and red.wait() is None
        torch.save([(m, r) for m, r, _ in outs], os.path.join(result_dir, f"red{rank}.pt"))
        # pixel gather: all ranks, and to rank 0 only
        local = torch.full((5, 4), float(rank))
        full = torch.empty((world * 5, 4))
        parallel.gather_pixels_async(dist, full, local, backend="gloo").wait()
        assert all((full[r * 5:(r + 1) * 5] == r).all() for r in range(world))
        only0 = torch.full((world * 5, 4), -1.0)
        parallel.gather_pixels_async(dist, only0, local, backend="gloo", dst=0).wait()
        if rank == 0:
            assert torch.equal(only0, full)
        else:
            assert (only0 == -1).all()
    finally:
        dist.destroy_process_group()


def test_overlapped_chunked_grad_reduce_equals_plain_sum(tmp_path):
    """OverlappedGradReducer (row chunks, one gradient in flight, wait() before the read) on two
    gloo ranks: every reduced gradient equals the sum of the two ranks' gradients, as ONE plain
    all-reduce would give; gather_pixels_async to all ranks and to one."""
    world = 2
    port = 29500 + (os.getpid() % 400) + 401
    mp.spawn(_reducer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), f"red{r}.pt")) for r in range(world))
    for (mine_a, red_a), (mine_b, red_b) in zip(a, b):
        assert torch.equal(red_a, red_b)
        torch.testing.assert_close(red_a, mine_a + mine_b, rtol=0, atol=0)


def _exchange_worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        # integer-valued floats: every order of additions gives the same bits, so "equal to the plain sum" is exact
        for M, K in ((1001, 28), (7, 3), (2, 5), (512, 32)):      # ragged over 3 ranks, fewer rows than ranks' share, ...
            g = torch.Generator().manual_seed(1000 * rank + M)
            mine = torch.randint(-50, 50, (M, K), generator=g).float()
            want = mine.clone()
            dist.all_reduce(want)
            got = parallel.direct_all_reduce(dist, mine.clone())
            res[f"direct_{M}"] = bool(torch.equal(got, want))
            # rows touched by some ranks only: rank r touches blocks where (block + r) % 3 != 0, nobody touches every 5th
            blocks = torch.arange((M + 15) // 16)
            keep = (((blocks + rank) % 3 != 0) & (blocks % 5 != 4)).repeat_interleave(16)[:M]
            sp = mine * keep[:, None]
            want = sp.clone()
            dist.all_reduce(want)
            got, stats = parallel.sparse_all_reduce(dist, sp.clone(), block_rows=16)
            res[f"sparse_{M}"] = bool(torch.equal(got, want))
            res[f"stats_{M}"] = stats
        # through the reducer, as bench.py uses it
        for mode in ("direct", "touched"):
            red = parallel.OverlappedGradReducer(dist, backend="gloo", mode=mode)
            grad = torch.randint(-9, 9, (333, 28), generator=torch.Generator().manual_seed(rank)).float()
            want = grad.clone()
            dist.all_reduce(want)
            red.start(grad)
            res[f"reducer_{mode}"] = bool(torch.equal(red.wait(), want))
        torch.save(res, os.path.join(result_dir, f"ex{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_and_touched_block_exchanges_equal_the_plain_sum(tmp_path, world):
    """parallel.direct_all_reduce (reduce-scatter + all-gather as two rounds of simultaneous point-to-point
    transfers: the form NOTEBOOK.md 7 prices for the xGMI mesh) and parallel.sparse_all_reduce (only the row blocks
    somebody touched) against dist.all_reduce on 2 and 3 gloo ranks: ragged row splits, fewer rows than ranks,
    block masks that differ per rank, blocks nobody touched, a short last block."""
    port = 30500 + (os.getpid() + 7 * world) % 900
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"ex{r}.pt"))
        bad = [k for k, v in res.items() if isinstance(v, bool) and not v]
        assert not bad, (r, bad)
        st = res["stats_1001"]
        # fewer blocks travel than a dense exchange would move, both rounds
        assert 0 < st["round1_blocks"] < st["dense_blocks"] / 2 and 0 < st["round2_blocks"] < st["dense_blocks"] / 2 * 1.0 + 1, st
