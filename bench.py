#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of volume_render forward + backward on synthetic
800x800 renders of the depth-8 SH9 shell tree (BASELINE.json configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ...] [--forward-only] [--route plain]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one ray batch: VolumeRenderer.forward on Q = W*H rays
followed by backward() of a fixed upstream gradient, i.e. volume_render + volume_render_backward
through the autograd.Function surface.  Inputs (tree topology, feature table, rays, upstream
gradient) are resident in HBM before the timed region.  With N > 1 every rank holds a replica of
the tree and renders its own camera (weak scaling: per-GPU work fixed); the gradient all-reduce of
a step runs on a side stream, in row chunks, under the next step's forward, and the pixels are
gathered while the backward runs (svox_t_amd/parallel.py).

Prints ONE JSON line on rank 0 (contract in the task statement).  `roofline` prices the bytes
THIS implementation cannot avoid moving to and from memory for the dominant kernel group (what
the algorithm as run reads and writes once: rays, pixels, sample records, the distinct feature
rows and tree words touched, the memset, the atomic requests, the row compaction -- counted on
the device before the timed region) against the 8 TB/s HBM peak; `reference_equivalent_gbps`
keeps SURVEY.md 8(d)'s figure for the reference's algorithm (three marches, 8 B per gradient
float) next to it.  The working set of configs 1-3 lives in the 256 MiB Infinity Cache, so
`limits` names what actually bounds each kernel.  `cpu_baseline`: the CPU oracle on the same
workload, host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# exp/atomic_bench.hip: the memory side takes 64-byte float-atomic requests at
# ~22 G/s whatever their size or scope (3.0 M wave-atomics of two segments each: 0.614 ms)
ATOMIC_REQUESTS_PER_S = 22e9
# us per leaf crossing of a wavefront alone on its SIMD: r02 timeline of march_rec_kernel
# (exp/trace_march.py, profiles/r02_march_timeline.txt) for the stepping alone; r01 timeline of
# render_fwd_kernel ("Timelines") for stepping + shading in one chain
US_PER_CROSSING_UNLOADED = {"march": 0.72, "march+shade": 1.14}

WORKLOADS = {
    # name: (depth, K, data_format, width, height)
    "d8_sh9_800": (8, 28, "SH9", 800, 800),          # BASELINE configs[1]/[2] -- the metric's config
    "d5_rgba_64": (5, 4, "RGBA", 64, 64),            # configs[0]
    "d9_rgba32_1024": (9, 32, "RGBA", 1024, 1024),   # configs[3]
}


def pmc_traffic(workload, forward_only, group):
    """(bytes per launch group, note) from this round's committed rocprofv3 --pmc passes of the same
    command (scripts/pmc_passes.sh -> scripts/pmc_summary.py), or (None, reason)."""
    name = f"r02_{workload}{'_fwd' if forward_only else ''}_pmc.json"
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            counters = json.load(f)["counters"]
        ks = {"forward": ("fwd", "march", "shade"), "backward": ("bwd", "fused", "merge", "wide", "compact")}[group]
        fetch = sum(counters[k]["FETCH_SIZE"] for k in ks if k in counters) * 1024.0
        write = sum(counters[k]["WRITE_SIZE"] for k in ks if k in counters) * 1024.0
        if fetch + write == 0:
            return None, f"no kernel of the {group} group in profiles/{name}"
        return int(2 * fetch + write), (
            f"(2 x FETCH_SIZE + WRITE_SIZE) x 1024 from profiles/{name} (separate --pmc passes of this command): "
            "MI355X_MICROARCH.md 'HBM': gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -- doubled; "
            "WRITE_SIZE exact for streaming stores and float atomics.  Memory-side requests of the L2s: "
            "Infinity-Cache hits are included, so this bounds HBM bytes from above")
    except Exception as exc:   # no profile committed for this build / workload
        return None, f"no PMC summary profiles/{name} ({exc.__class__.__name__})"


def reference_equivalent_bytes(cnt, Q, M, K, C):
    """the bytes of the REFERENCE's algorithm.  cnt = (rays_hit, steps S, levels sum L, valid, active)."""
    _, S, L, V, A = cnt
    march = 4 * L + 4 * S + 4 * V + 4 * (K - 1) * A
    fwd = Q * (36 + 4 * (C + 1)) + march
    bwd = 4 * M * K + Q * (36 + 4 * (C + 1)) + 2 * march + A * (8 * (K - 1) + 8)
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="d8_sh9_800", choices=list(WORKLOADS))
    ap.add_argument("--route", default="hinted", choices=["hinted", "plain"],
                    help="hinted: VolumeRenderer.forward(..., image_shape=(H, W)); plain: exactly the two calls the "
                         "reference's own autograd function makes on the operator module (no hint, no extra argument)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain", action="store_true", help="skip the extra plain-route measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for dry runs")
    ap.add_argument("--share-device", action="store_true",
                    help="dry run: put every rank on cuda:0 (to rehearse the N>1 code path on a 1-GPU box)")
    ap.add_argument("--forward-only", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist_mod
        dist = dist_mod
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=300), **kw)

    import svox_t_amd as svox
    import svox_t_amd.csrc as _C
    from svox_t_amd import parallel, synth
    from svox_t_amd.renderer import _rays_spec_from_rays

    depth, K, fmt, W, H = WORKLOADS[args.workload]
    Q = W * H
    st = synth.shell_tree(depth)
    feats = synth.shell_features(st.n_features, K)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
    renderer = svox.VolumeRenderer(tree)
    # rank r renders camera r (azimuth 30 + 45 r degrees), SURVEY.md 8(d)
    pose = synth.camera_pose(azimuth_deg=30.0 + 45.0 * rank)
    o, d, v = synth.pinhole_rays(W, H, c2w=pose)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    features = tree.features
    M = features.shape[0]
    opt = renderer._get_options()
    C = _C.get_out_data_dim(opt, K) - 1
    gout = synth.grad_output(Q, C + 1).to(dev)

    class _ReferenceShaped(torch.autograd.Function):
        """svox_t/renderer.py:60-77: the two calls the reference's own function makes."""

        @staticmethod
        def forward(ctx, data, tspec, rspec, ropt):
            out = _C.volume_render(tspec, rspec, ropt)
            ctx.tree, ctx.rays, ctx.opt = tspec, rspec, ropt
            return out

        @staticmethod
        def backward(ctx, grad_out):
            return _C.volume_render_backward(ctx.tree, ctx.rays, ctx.opt, grad_out.contiguous()), None, None, None

    def render(route):
        if route == "plain":
            rs = _C.RaysSpec()
            rs.origins, rs.dirs, rs.vdirs = rays.origins, rays.dirs, rays.viewdirs
            return _ReferenceShaped.apply(features, tree._spec(features), rs, opt)
        return renderer(features, rays, image_shape=(H, W))

    # ---- one-off device-side counts (before the timed region) -------------------------------
    rs_hint = _rays_spec_from_rays(rays, (H, W))
    rs_hint.need_grad = False
    spec = tree._spec(features)
    cnt = _C.count_forward(spec, rs_hint, opt).cpu().tolist()
    touched = _C.count_touched(spec, rs_hint, opt)
    atomic_requests = merged_rows = rays_with_samples = None
    if not args.forward_only:
        with _C.bwd_counters(dev) as ctr:
            features.grad = None
            render(args.route).backward(gout)
            torch.cuda.synchronize()
        atomic_requests, merged_rows = ctr.read()
        features.grad = None
    reducer = parallel.OverlappedGradReducer(dist, backend=args.backend) if dist is not None else None
    gathered = torch.empty((world * Q, C + 1), dtype=torch.float32, device=dev) if dist is not None else None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(i=None, route=args.route):
        e = ev[i] if i is not None else None
        features.grad = None              # (a gradient still travelling is held by the reducer)
        if e: e[0].record()
        if args.forward_only:
            with torch.no_grad():                              # inference: nothing is recorded for a backward
                out = render(route)
        else:
            out = render(route)
        if e: e[1].record()
        gather = None
        if dist is not None:
            # the pixels are final after the forward: gather them while the backward runs
            gather = parallel.gather_pixels_async(dist, gathered, out.detach(), backend=args.backend)
        if not args.forward_only:
            out.backward(gout)
        if e: e[2].record()
        if dist is not None:
            if not args.forward_only:
                # gradient accumulation over batches: the previous batch's reduced gradient is complete
                # here; this batch's travels (side stream, row chunks) under the next batch's work
                reducer.start(features.grad)
            if gather is not None:
                gather.wait()
        return out

    # Setup, not measurement: the first process on a fresh box has been seen to run its first
    # dozens of steps far below steady state (allocator growth, code-object loads, clocks).
    # Bring the device there before the W warm-up steps the contract asks for.
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    if reducer is not None:
        reducer.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if reducer is not None:
        reducer.wait()                    # the last gradient is reduced inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    route_fwd, route_bwd = _C.LAST_ROUTE["forward"], (None if args.forward_only else _C.LAST_ROUTE["backward"])
    forward_terms = bool(_C.LAST_ROUTE.get("forward_terms"))       # (of the timed route: the runs below take others)
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    bwd_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps

    # the other route, for the record (same process, after the timed region)
    other = None
    if world == 1 and not args.no_plain:
        oroute = "plain" if args.route == "hinted" else "hinted"
        for _ in range(3):
            step(route=oroute)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(route=oroute)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / args.steps
        other = {"route": oroute, "value": round(Q / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dt * 1e3, 4),
                 "what": ("the two calls the reference's own autograd function makes (svox_t/renderer.py:60-77) on "
                          "svox_t_amd.csrc: no image hint; the operator layer orders the rays, records and replays "
                          "the sample lists by itself" if oroute == "plain" else
                          "VolumeRenderer.forward(..., image_shape=(H, W))")}

    # round 1's headline arithmetic, for the record (same process, after the timed region): the backward
    # that takes accum from the forward's output instead of adding it up like the reference's first pass
    single_march = None
    if world == 1 and not args.no_plain and not args.forward_only and _C.BWD_EXACT:
        _C.BWD_EXACT = False
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
        finally:
            _C.BWD_EXACT = True
        single_march = {"setting": "SVOXT_BWD_EXACT=0", "value": round(Q / dt / 1e6, 3), "unit": "Mrays/s",
                        "ms_per_step": round(dt * 1e3, 4),
                        "what": "the single-march backward round 1's headline (BENCH_r01: 1038.5) was measured with: within "
                                "1e-5 of the summed magnitudes, but 36 % of the sigma-column entries differ from the "
                                "reference's by more than 1e-5 of their own value (tests/test_gpu_query_and_misc.py); "
                                "`value` above is the exact backward"}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * Q / (elapsed / args.steps) / 1e6
        ref_fwd, ref_bwd = reference_equivalent_bytes(cnt, Q, M, K, C)
        A = cnt[4]
        stride = K if (K <= 8 or K % 16 == 0) else (K + 15) // 16 * 16
        tree_bytes = 4 * touched.get("grid_cells", 0) + 8 * touched.get("node_pairs", 0) + \
            4 * (touched.get("child_words", 0) + touched.get("data_words", 0))
        recording = not args.forward_only
        fwd_parts = {
            "rays": 36 * Q, "pixels_written": 4 * (C + 1) * Q,
            "feature_rows_read": 4 * K * touched["rows_valid"], "tree_words_read": tree_bytes,
            "records_written": 8 * A if recording else 0, "aux_written": 16 * Q if recording else 0,
            # (att, e0, e1, e2) per sample, left by the recording forward for the exact backward
            "backward_terms_written": 16 * A if (recording and forward_terms) else 0,
        }
        bwd_parts = None
        if not args.forward_only:
            bwd_parts = {
                "grad_memset": 4 * M * stride, "upstream_gradient_read": 4 * (C + 1) * Q, "aux_read": 16 * Q,
                "rays": 36 * cnt[0], "records_read": 8 * A,
                # with the forward's hand-over the backward reads 16 B per sample instead of the feature rows
                "feature_rows_read": 0 if forward_terms else 4 * K * touched["rows_composited"],
                "terms_read": 16 * A if forward_terms else 0,
                "atomic_requests_64B": 64 * atomic_requests if atomic_requests else 4 * K * A,
                "row_compaction": (4 * M * stride + 4 * M * K) if stride != K else 0,
            }
            if "grad_wide_kernel" in (route_bwd or "") or "ONEPASS" in (route_bwd or ""):
                # sweep 1 -> sweep 2: (attenuation,) second-pass total_color per sample, written and read
                bwd_parts["sweep_handover"] = (16 if "grad_wide_kernel" in route_bwd else 8) * A
            if atomic_requests is None or not atomic_requests:
                bwd_parts["atomic_requests_note"] = "one row per sample (not counted on the device for this route; " \
                    "grad_wide_kernel merges rows per tile and window of 16 list positions: exp/reuse_probe.py)"
        fwd_bytes = sum(fwd_parts.values())
        bwd_bytes = sum(v for v in bwd_parts.values() if not isinstance(v, str)) if bwd_parts else 0
        if args.forward_only or fwd_ms >= bwd_ms:
            dom, dom_kernel, dom_ms, dom_bytes, dom_ref = "forward", route_fwd, fwd_ms, fwd_bytes, ref_fwd
        else:
            dom, dom_kernel, dom_ms, dom_bytes, dom_ref = "backward", route_bwd, bwd_ms, bwd_bytes, ref_bwd
            dom_kernel += " + grad memset + row compaction"
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_note = pmc_traffic(args.workload, args.forward_only, dom) if world == 1 else \
            (None, "PMC profiles exist for N=1 only")
        chain = "march" if "march_rec" in (route_fwd or "") else "march+shade"
        limits = {"forward": {
            "bound": "dependent chain of the longest ray (tree words -> step -> next tree words), not bytes",
            "longest_ray_crossings": touched["longest_ray_crossings"],
            "us_per_crossing_unloaded": US_PER_CROSSING_UNLOADED[chain],
            "floor_ms": round(touched["longest_ray_crossings"] * US_PER_CROSSING_UNLOADED[chain] * 1e-3, 4),
            "measured_ms": round(fwd_ms, 4)}}
        if "shade_chan" in (route_fwd or ""):
            limits["forward"]["note"] = ("two kernels: the march is bound by this chain; shade_chan_kernel by the vector ALUs "
                                         "(one exponential and one double-precision divide per channel and sample) and, when the "
                                         "feature table exceeds the 256 MiB Infinity Cache, by HBM (see roofline.traffic)")
        if atomic_requests:
            limits["backward"] = {
                "bound": "rate at which the memory side takes 64-byte float-atomic requests (exp/atomic_bench.hip: 22 G/s)",
                "atomic_requests": atomic_requests, "merged_rows": merged_rows,
                "floor_ms": round(atomic_requests / ATOMIC_REQUESTS_PER_S * 1e3, 4), "measured_ms": round(bwd_ms, 4)}
        res = {
            "metric": "Mrays/s fwd+bwd, 800×800 render, depth-8 SH9 N3Tree, 1→8 MI355X"
                      if args.workload == "d8_sh9_800" and not args.forward_only
                      else f"Mrays/s {'fwd' if args.forward_only else 'fwd+bwd'}, {args.workload}",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"depth-{depth} shell N3Tree (n_internal {st.n_internal}, M {M}), "
                            f"{fmt} data_dim {K}, {W}x{H} pinhole rays per GPU, "
                            f"{'forward' if args.forward_only else 'forward+backward'}, "
                            f"step_size 1e-3, thresholds 0",
                "rays_per_gpu": Q,
                "route": args.route,
                "backward_arithmetic": None if args.forward_only else
                ("exact (every contribution the reference's formula)" if _C.BWD_EXACT else "single march (SVOXT_BWD_EXACT=0)"),
                "partitioning": "replicated tree, one camera (ray batch) per GPU"
                                + ("; all-gather of pixels under the backward, all-reduce of grad in row chunks on a "
                                   "side stream under the next step's forward" if world > 1 else ""),
            },
            "kernel_ms": {"forward": round(fwd_ms, 4), "backward": round(bwd_ms, 4)},
            "kernels": {"forward": route_fwd, "backward": route_bwd},
            "counters": dict(zip(("rays_hit", "steps", "levels", "valid", "active"), cnt)),
            "touched": touched,
            "compulsory_bytes": {"forward": fwd_parts, "backward": bwd_parts,
                                 "step_total": fwd_bytes + bwd_bytes,
                                 "step_gbps": round((fwd_bytes + bwd_bytes) / (ms_per_step * 1e-3) / 1e9, 2)},
            "roofline": {
                "bound": "hbm",
                "kernel": dom_kernel,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "achieved_note": f"compulsory bytes of the {dom} as run (compulsory_bytes.{dom}: each item counted once, "
                                 "distinct rows / tree words / atomic requests counted on the device) / its mean duration "
                                 "from HIP events on the launch stream; the working set of this config sits in the 256 MiB "
                                 "Infinity Cache, so the fraction of the HBM roofline is low by construction: see limits",
                "reference_equivalent_gbps": round(dom_ref / (dom_ms * 1e-3) / 1e9, 2),
                "reference_equivalent_note": "SURVEY.md 8(d): the bytes the REFERENCE's algorithm moves for the same result "
                                             "(a tree march per pass, 8 B per gradient float) / this kernel group's time",
            },
            "limits": limits,
        }
        if other is not None:
            res["other_route"] = other
        if single_march is not None:
            res["tolerance_mode"] = single_march
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(st, feats, o, d, v, fmt, K, gout.cpu(), args.forward_only)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(st, feats, o, d, v, fmt, K, gout, forward_only):
    """The CPU oracle (oracle/, a restatement -- the reference's own CPU renderer
    asserts) on the same workload, all host cores (OpenMP)."""
    from oracle import oracle as O
    from svox_t_amd.helpers import DataFormat
    df = DataFormat(fmt)
    ot = O.Tree(feats.numpy(), st.data, st.child)
    opt = O.make_options(format=df.format, basis_dim=df.basis_dim)
    Q = o.shape[0]
    rays = (o.numpy(), d.numpy(), v.numpy())
    g = gout.numpy()
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < 10.0 and reps < 64:          # ~10 s of wall time on the host cores
        O.volume_render(ot, *rays, opt)
        if not forward_only:
            O.volume_render_backward(ot, *rays, opt, g)
        reps += 1
        dt = time.perf_counter() - t0
    return {
        "value": round(reps * Q / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"the full workload ({Q} rays, {'forward' if forward_only else 'forward+backward'}) "
                  f"x {reps} repetitions, {dt:.1f} s wall, OpenMP over rays on {O.num_threads()} threads",
    }


if __name__ == "__main__":
    main()
