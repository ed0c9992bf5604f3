#!/usr/bin/env python3
"""bench.py -- mapper train-iteration throughput of the rasterizer hot path on MI355X.

One "step" = one pass of the hot path over one keyframe of synthetic input, inputs resident in HBM:
  forward raster (preprocess, scan, duplicate, radix sort, ranges, render)  ->  backward raster
  (tile backward, fused per-Gaussian backward) with a fixed dL/dimage  [-> RCCL all-reduce of the
  Gaussian-parameter gradients when N > 1: keyframe-parallel training, one keyframe per GPU].

Contract: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.  `value` = keyframe-iterations per second over all N GPUs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def algorithmic_bytes(P, P_vis, R, W, H, passes_depth, passes_tile):
    """algorithmic bytes per kernel of one fwd+bwd raster.  The sort line is this design's
    two-level sort (P depth keys, then R tile keys; per pass: count reads the 8-B key, scatter reads and writes the
    12-B key+value pair), which moves fewer bytes than the reference's (8+24*passes)*R single 64-bit sort."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {
        "preprocess_fwd_kernel": 104 * P_vis + 8 * (P - P_vis),
        "scan_block_sums_kernel": 8 * P,
        "duplicate_with_keys_kernel": 20 * P + 12 * R,
        "radix_sort(all passes)": 32 * (passes_depth * P + passes_tile * R),
        "identify_tile_ranges_kernel": 8 * R + 8 * T,
        "render_fwd_kernel": 40 * R + 20 * W * H,
        "render_bwd_kernel": 40 * R + 20 * W * H + 36 * R,
        "preprocess_bwd_kernel": (92 + 132 + 108) * P,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2_1080p", help="scene config name (segs_slam_amd.scenes.CONFIGS)")
    ap.add_argument("--mode", default="raster", choices=["raster", "trainer", "scaffold"],
                    help="raster: fwd+bwd raster (+all-reduce); trainer: + fused L1/SSIM loss and fused Adam over the "
                         "Gaussians; scaffold: the anchor-level mapper step (prefilter, neural-Gaussian MLPs, raster, loss, "
                         "their backward, Adam) over --anchors anchors x 10 offsets (SURVEY 8d config 3)")
    ap.add_argument("--anchors", type=int, default=50000)
    ap.add_argument("--appearance-dim", type=int, default=32, help="scaffold mode: Model.appearance_dim (ScanNet configurations: 16)")
    ap.add_argument("--no-feat-bank", action="store_true", help="scaffold mode: Model.use_feat_bank = 0 (ScanNet configurations)")
    ap.add_argument("--sync-forward", action="store_true",
                    help="use the reference-shaped forward that blocks on a D2H copy of num_rendered every step "
                         "(default: resident no-sync entry points after one calibrating step)")
    ap.add_argument("--graph", action="store_true", help="replay the resident fwd+bwd from a captured hipGraph (N=1 only); the "
                    "dominant kernel's live HIP-event timing is then taken from the eager warm-up pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks use cuda:0 (with --backend gloo): exercises the N > 1 code path on a one-GPU box; not a measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (exercises the RCCL calls on a one-GPU box)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel table to stderr")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print there (RCCL writes its version banner to stdout when the
    # first communicator is created) are sent to stderr for the whole run, the line goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from segs_slam_amd import scenes
    from segs_slam_amd.raster_engine import KernelProfile, RasterEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- workload: same Gaussians on every rank, one keyframe (camera pose) per rank
    sc = scenes.make_config_scene(args.workload, keyframe=rank)
    cam = sc.camera
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    bg, m3, col, op, sca, rot = t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    dL = t(sc.dL_dout_color)
    eng = RasterEngine(sc.P, cam.width, cam.height, dev, resident=not args.sync_forward)

    if args.mode == "scaffold":
        from segs_slam_amd import neural_gaussians as ng
        dims = ng.ModelDims(appearance_dim=args.appearance_dim, use_feat_bank=not args.no_feat_bank)
        model = ng.synthetic_model(args.anchors, dims, cam, dev, seed=0)   # replicas must be identical; the keyframe differs per rank
        tstep = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
        eng = tstep.engine
        kfs = [ng.Keyframe(view, proj, campos, torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)]
        gts = [torch.rand(3, cam.height, cam.width, device=dev)]
        tstep.keyframe_for = lambda step, n: 0
    if args.mode == "trainer":
        from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
        tstep = TrainerStep.on_gpu(sc, dev)
        eng = tstep.engine
        kfs = [keyframe_tensors(cam, dev)]
        gts = [torch.rand(3, cam.height, cam.width, device=dev)]
        tstep.keyframe_for = lambda step, n: 0

    def step():
        if args.mode in ("trainer", "scaffold"):
            tstep.training_once(kfs, gts)
            return
        eng.forward(bg, m3, col, op, sca, rot, view, proj, campos, cam.tanfovx, cam.tanfovy)
        eng.backward(dL)
        if use_dist:
            dist.all_reduce(eng.grads_flat)  # sum of per-keyframe parameter gradients over xGMI (RCCL)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup; the per-kernel breakdown (all kernels, HIP events) is taken on the warmup steps
    with KernelProfile() as prof_all:
        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
    breakdown = prof_all.result
    sort_names = ("radix_count_kernel", "radix_scan_kernel", "radix_scatter_kernel")
    per_step = {k: v["total_ms"] / max(args.warmup, 1) for k, v in breakdown.items()}
    dominant = max((k for k in per_step if k != "memset"), key=lambda k: per_step[k])

    # ---- timed region: exactly K steps, barrier + sync on both sides; the dominant kernel is timed live
    # with HIP events on its launch stream inside this region
    graph = None
    if args.graph and world == 1 and args.mode == "raster" and not args.sync_forward:
        eng.check()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        torch.cuda.synchronize()
    fence()
    with KernelProfile([dominant] if graph is None else []) as prof_dom:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                step()
        fence()
        elapsed = time.perf_counter() - t0
    if graph is not None:
        prof_dom.result = {dominant: breakdown[dominant]}
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        eng.check()
        # R of the formulas below is the REFERENCE's num_rendered (bounding-square duplication, taken from the calibrating
        # reference-shaped forward); the resident forwards bin fewer instances (tight rectangles, dead instances dropped)
        R = eng.R_reference or eng.R
        P_vis = int((eng.radii > 0).sum().item())
        gx, gy = (cam.width + 15) // 16, (cam.height + 15) // 16
        passes_tile = (max(1, int(gx * gy - 1).bit_length()) + 7) // 8
        passes = int(round(breakdown.get("radix_scatter_kernel", {"launches": 0})["launches"] / max(args.warmup, 1)))
        passes_depth = max(passes - passes_tile, 0)
        ab = algorithmic_bytes(eng.P, P_vis, R, cam.width, cam.height, passes_depth, passes_tile)
        dom = prof_dom.result[dominant]
        dom_bytes = ab[dominant] if dominant in ab else ab["radix_sort(all passes)"] / (3 * max(passes, 1))
        traffic, traffic_src = pmc_traffic(args.workload, dominant) if args.mode == "raster" else (None, None)
        achieved = dom_bytes / (dom["avg_ms"] * 1e-3) / 1e9
        total_bytes = sum(ab.values())
        ms_per_step = elapsed / args.steps * 1e3
        raster_ms = sum(v for k, v in per_step.items())
        out = {
            "metric": "mapper train-iters/sec (fwd+bwd raster)",
            "value": world * args.steps / elapsed,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {eng.P} Gaussians, {cam.width}x{cam.height}, 1 keyframe per GPU, "
                                   "fwd+bwd raster" + (", RCCL all-reduce of parameter grads" if world > 1 else "")
                                   + (" + L1/SSIM loss + fused Adam" if args.mode == "trainer" else "")
                                   + (f"; anchor-level mapper step: {args.anchors} anchors x 10 offsets -> neural Gaussians "
                                      f"(appearance_dim {args.appearance_dim}, feature bank {'off' if args.no_feat_bank else 'on'}; MLPs fwd+bwd), "
                                      "L1/SSIM, fused Adam" if args.mode == "scaffold" else ""),
                       "P": eng.P, "P_visible": P_vis, "num_rendered": R, "instances_binned": eng.R, "width": cam.width, "height": cam.height,
                       "sort_passes": {"depth_keys_P": passes_depth, "tile_keys_R": passes_tile}, "parallelism": f"keyframe-dp{world}",
                       "forward": "sync (reference API)" if args.sync_forward else "resident (no host sync)"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                         "note": "tile kernels are VALU-bound, not HBM-bound (SURVEY 8d); see DESIGN.md"},
            "raster": {"algorithmic_bytes": total_bytes, "kernel_ms_sum": raster_ms,
                       "achieved_GBps": total_bytes / (raster_ms * 1e-3) / 1e9,
                       "kernel_ms": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
                       "radix_sort_ms": round(sum(per_step.get(k, 0.0) for k in sort_names), 4)},
        }
        try:   # the tile kernels are reported in (pixel, Gaussian) pairs per second next to the byte figure
            out["raster"].update(pair_rates(eng, cam, per_step))
        except Exception as e:  # noqa: BLE001  (measurement garnish only; never fail the bench line over it)
            out["raster"]["pairs_error"] = str(e)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc)
        if args.breakdown:
            for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]):
                print(f"  {k:32s} {v:8.4f} ms/step", file=sys.stderr)
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.destroy_process_group()


def pair_rates(eng, cam, per_step):
    """Upper bound sum_tiles len(tile) * 256 and exact count sum(n_contrib) of (pixel, Gaussian) pairs of the last forward,
    and the two tile kernels' rates on the upper bound (what they iterate over before masks and early termination)."""
    import ctypes as C
    import torch
    from segs_slam_amd import _capi
    img = eng._img_r if getattr(eng, "_last_resident", False) else eng.img.tensor
    tiles = ((cam.width + 15) // 16) * ((cam.height + 15) // 16)
    ranges = torch.zeros((tiles, 2), dtype=torch.int32, device=eng.device)
    ncontrib = torch.zeros((cam.height, cam.width), dtype=torch.int32, device=eng.device)
    st = _capi.lib().segs_debug_unpack_image(C.c_void_p(img.data_ptr()), cam.width, cam.height, C.c_void_p(ranges.data_ptr()), None,
                                             C.c_void_p(ncontrib.data_ptr()), C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream))
    _capi.check(st, "segs_debug_unpack_image")
    torch.cuda.synchronize()
    upper = int((ranges[:, 1] - ranges[:, 0]).to(torch.int64).sum().item()) * 256
    exact = int(ncontrib.to(torch.int64).sum().item())
    out = {"pairs_upper_bound": upper, "pairs_exact_sum_n_contrib": exact}
    for k in ("render_fwd_kernel", "render_bwd_kernel"):
        if k in per_step and per_step[k] > 0:
            out[k.replace("_kernel", "") + "_Gpairs_per_s"] = round(upper / (per_step[k] * 1e-3) / 1e9, 1)
    return out


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and
    WRITE_SIZE need separate passes, so they cannot be collected inside the timed run): profiles/rNN_pmc_<workload>.json,
    written by tools/collect_profiles.sh + tools/pmc_summary.py (FETCH_SIZE doubled per the gfx950 correction)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{workload}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        tab = json.load(f)
    e = tab.get("segs::" + kernel)
    if not e or "hbm_bytes" not in e:
        return None, None
    return e["hbm_bytes"], os.path.relpath(files[-1], ROOT)


def cpu_baseline(sc, budget_s: float = 12.0, max_iters: int = 10):
    """The CPU oracle (a port of the reference algorithm; the reference has no CPU raster path)
    timed on this host: whole fwd+bwd iterations of the same workload until about `budget_s` seconds of wall time are
    spent (at least 2, at most `max_iters`); the first iteration is reported but not counted (page faults, thread start)."""
    from oracle import gs_oracle
    cam = sc.camera
    o = gs_oracle.Oracle()
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start < budget_s and len(times) < max_iters + 1):
        t0 = time.perf_counter()
        o.forward(sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.scale_modifier, sc.rotations,
                  cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.height, cam.width)
        t1 = time.perf_counter()
        o.backward(sc.dL_dout_color)
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
    timed = times[1:]
    fwd = sum(t[0] for t in timed) / len(timed)
    bwd = sum(t[1] for t in timed) / len(timed)
    return {"value": 1.0 / (fwd + bwd), "unit": "iters/s", "cores": int(gs_oracle.lib().gso_threads()), "kind": "port",
            "sample": f"{len(timed)} fwd+bwd iterations of the same scene after one untimed (mean fwd {fwd:.2f} s, bwd {bwd:.2f} s; "
                      f"first iteration {sum(times[0]):.2f} s), OpenMP oracle",
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
