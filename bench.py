#!/usr/bin/env python3
"""bench.py -- mapper train-iteration throughput of the rasterizer hot path on MI355X.

One "step" = one pass of the hot path over one keyframe of synthetic input, inputs resident in HBM:
  forward raster (preprocess, depth sort, instance emission, tile sort, ranges, render)  ->  backward raster
  (tile backward, fused per-Gaussian backward) with a fixed dL/dimage  [-> RCCL all-reduce of the
  Gaussian-parameter gradients when N > 1: keyframe-parallel training, one keyframe per GPU].

Contract: python bench.py --gpus N --steps K --warmup W prints ONE JSON line (rank 0).  With N > 1 the ranks are either
the ones torch.distributed.run started (RANK / WORLD_SIZE in the environment) or, when the script is started plainly,
N child processes this script starts itself BEFORE anything touches the GPU.  `value` = keyframe-iterations per second over
all N GPUs.  Default workload: `1080p_3m` (3 M Gaussians at 1920x1080: the size north_star names, outside the 256 MiB
Infinity Cache).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak; 256 CUs x 4 SIMDs at 2.4 GHz, a wave64 VALU instruction occupies its
# SIMD for 2 cycles; chip-wide global float-atomic rate ("Global float atomics": 1.26-1.36 TB/s of added bytes)
HBM_PEAK_GBPS = 8000.0
VALU_PEAK_GINST = 1024 * 2.4 / 2.0        # G wave-instructions per second
ATOMIC_PEAK_GBPS = 1300.0
LIVE_EVENT_KERNELS = ("render_bwd_kernel", "render_fwd_kernel", "preprocess_bwd_kernel")


def algorithmic_bytes(P, P_vis, R, W, H, passes_depth, passes_tile):
    """algorithmic bytes per kernel of one fwd+bwd raster over R instances.  The sort line is this
    design's two-level sort (P depth keys, then R tile keys, both 32-bit: per pass the count reads the 4-byte key, the
    scatter reads and writes the 8-byte key+value pair = 20 B per key and pass), which moves far fewer bytes than the
    reference's (8+24*passes)*R single 64-bit sort."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {
        "preprocess_fwd_kernel": 104 * P_vis + 8 * (P - P_vis),
        "scan_block_sums_kernel": 8 * P,
        "duplicate_with_keys_kernel": 20 * P + 12 * R,
        "radix_sort(all passes)": 20 * (passes_depth * P + passes_tile * R),
        "identify_tile_ranges_kernel": 8 * R + 8 * T,
        "render_fwd_kernel": 40 * R + 20 * W * H,
        "render_bwd_kernel": 40 * R + 20 * W * H + 36 * R,
        "preprocess_bwd_kernel": (92 + 132 + 108) * P,
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed iterations, median and p10 / p90
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="1080p_3m", help="scene config name (segs_slam_amd.scenes.CONFIGS)")
    ap.add_argument("--mode", default="raster", choices=["raster", "trainer", "scaffold"],
                    help="raster: fwd+bwd raster (+all-reduce); trainer: + fused L1/SSIM loss and fused Adam over the "
                         "Gaussians; scaffold: the anchor-level mapper step (prefilter, neural-Gaussian MLPs, raster, loss, "
                         "their backward, Adam) over --anchors anchors x 10 offsets (SURVEY 8d config 3)")
    ap.add_argument("--anchors", type=int, default=50000)
    ap.add_argument("--appearance-dim", type=int, default=32, help="scaffold mode: Model.appearance_dim (ScanNet configurations: 16)")
    ap.add_argument("--no-feat-bank", action="store_true", help="scaffold mode: Model.use_feat_bank = 0 (ScanNet configurations)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "sharded", "dense"],
                    help="N > 1: gradient exchange. sharded = reduce-scatter -> Adam on the rank's shard -> all-gather; dense = one "
                         "all-reduce (the overflow word rides in it) and a full Adam on every rank; auto = by bucket size "
                         "(keyframe_parallel.BucketExchange.AUTO_SHARD_BYTES)")
    ap.add_argument("--dense-allreduce", action="store_true",
                    help="trainer / scaffold mode with N > 1: one all-reduce and a full Adam on every rank instead of "
                         "reduce-scatter -> sharded Adam -> all-gather")
    ap.add_argument("--sync-forward", action="store_true",
                    help="use the reference-shaped forward that blocks on a D2H copy of num_rendered every step "
                         "(default: resident no-sync entry points after one calibrating step)")
    ap.add_argument("--graph", action="store_true", help="N=1 only: replay the step from a captured hipGraph -- raster mode: the resident "
                    "fwd+bwd; trainer / scaffold mode: the whole iteration (staging buffers for keyframe, target and learning rates). The "
                    "dominant kernel's HIP-event timing then comes from the profiled steps after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks use cuda:0 (with --backend gloo): exercises the N > 1 code path on a one-GPU box; not a measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (exercises the RCCL calls on a one-GPU box)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--fuse-projection", type=int, default=1, help="scaffold mode: 1 = the neural forward runs the rasterizer's per-Gaussian stage itself (SURVEY 8f n3)")
    ap.add_argument("--neural-flags", type=int, default=0, help="segs_neural_set_flags bits for A/B profiles (1 = SEGS_NEURAL_ONE_KERNEL_BACKWARD)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra blocks of the line (mapper_step, config3, render_only_ms): A/B runs and profiles")
    return ap.parse_args()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, rendezvous on
    127.0.0.1) before this process has touched the GPU or imported torch, and hand rank 0's stdout -- the one JSON line --
    through.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, pending = 0, list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or code
                for q in pending:       # a rank died: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    sharded_opt = False if (args.dense_allreduce or args.exchange == "dense") else (True if args.exchange == "sharded" else "auto")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    # stdout carries exactly one JSON line: libraries that print there (RCCL writes its version banner to stdout when the
    # first communicator is created) are sent to stderr for the whole run, the line goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np

    from segs_slam_amd import scenes

    # The C++ drop-in path (libgaussian_rasterizer.so -> libcuda_rasterizer.so -> C ABI) is timed by its own driver in a child
    # process that has finished before this process touches the GPU.
    want_extras = args.gpus == 1 and args.mode == "raster" and not args.no_extras and "WORLD_SIZE" not in os.environ
    sc0 = scenes.make_config_scene(args.workload, keyframe=0) if want_extras else None
    cpp_dropin = cpp_dropin_block(sc0) if want_extras else None

    import torch
    import torch.distributed as dist

    from segs_slam_amd.raster_engine import KernelProfile, RasterEngine
    if args.neural_flags:
        from segs_slam_amd import _capi
        _capi.lib().segs_neural_set_flags(args.neural_flags)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    sc = sc0 if (sc0 is not None and rank == 0) else scenes.make_config_scene(args.workload, keyframe=rank)
    cam = sc.camera
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    bg, m3, col, op, sca, rot = t(sc.bg), t(sc.means3D), t(sc.colors), t(sc.opacity), t(sc.scales), t(sc.rotations)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    dL = t(sc.dL_dout_color)
    tstep = None
    if args.mode == "scaffold":
        from segs_slam_amd import neural_gaussians as ng
        dims = ng.ModelDims(appearance_dim=args.appearance_dim, use_feat_bank=not args.no_feat_bank)
        model = ng.synthetic_model(args.anchors, dims, cam, dev, seed=0)   # replicas must be identical; the keyframe differs per rank
        tstep = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
        tstep.fuse_projection = bool(args.fuse_projection)
        tstep.sharded_optimizer = sharded_opt
        tstep.single_rank_collectives = args.force_dist
        eng = tstep.engine
        kfs = [ng.Keyframe(view, proj, campos, torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)]
        gts = [torch.rand(3, cam.height, cam.width, device=dev)]
        tstep.keyframe_for = lambda step, n: 0
    elif args.mode == "trainer":
        from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
        tstep = TrainerStep.on_gpu(sc, dev, sharded_optimizer=sharded_opt, single_rank_collectives=args.force_dist)
        eng = tstep.engine
        kfs = [keyframe_tensors(cam, dev)]
        gts = [torch.rand(3, cam.height, cam.width, device=dev)]
        tstep.keyframe_for = lambda step, n: 0
    else:
        eng = RasterEngine(sc.P, cam.width, cam.height, dev, resident=not args.sync_forward)
    raster_ex = raster_params = None
    if tstep is None and use_dist:
        # the exchange of a keyframe-parallel step over the flat Gaussian bucket, through the same BucketExchange the trainer
        # steps use: reduce-scatter of the gradients + all-gather of a parameter-sized bucket (what follows the sharded Adam),
        # or one dense all-reduce -- the same bytes per link either way
        from segs_slam_amd.keyframe_parallel import BucketExchange
        # (no optimizer in this mode, so nothing for a sharded exchange to save: "auto" means the dense one here)
        raster_ex = BucketExchange(eng.grads_flat.numel(), dev, None, sharded=False if sharded_opt == "auto" else sharded_opt,
                                   single_rank_collectives=args.force_dist, grads=eng.grads_flat)
        raster_params = torch.zeros_like(eng.grads_flat) if raster_ex.sharded else None

    def step():
        if tstep is not None:
            tstep.training_once(kfs, gts)
            return
        eng.forward(bg, m3, col, op, sca, rot, view, proj, campos, cam.tanfovx, cam.tanfovy)
        eng.backward(dL)
        if raster_ex is not None:     # sum of per-keyframe parameter gradients over xGMI (RCCL)
            status = getattr(eng, "_status", None)
            raster_ex.reduce_flag_async(status[3:4] if (status is not None and eng._last_resident) else None)
            raster_ex.wait_flag()
            raster_ex.reduce_gradients(eng.grads_flat)
            if raster_params is not None:
                raster_ex.gather(raster_params)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- W untimed warm-up steps (the first one is the calibrating, synchronising forward of the resident engine)
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + sync on both sides; the tile kernels are timed live with HIP events on
    # their launch stream inside this region
    graph = None
    step_graph = bool(args.graph and world == 1 and tstep is not None)
    if step_graph:
        # trainer / scaffold modes: the step object replays its own whole-iteration graph (enable_graph); a few more steps so
        # that the capture itself happens before the timed region
        tstep.enable_graph(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    if args.graph and world == 1 and args.mode == "raster" and not args.sync_forward:
        eng.check()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        torch.cuda.synchronize()
    fence()
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one record per step: p10 / p50 / p90
    with KernelProfile(LIVE_EVENT_KERNELS if (graph is None and not step_graph) else []) as prof_live:
        t0 = time.perf_counter()
        step_events[0].record()
        for i in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                step()
            step_events[i + 1].record()
        fence()
        elapsed = time.perf_counter() - t0
    per_step_ms = sorted(step_events[i].elapsed_time(step_events[i + 1]) for i in range(args.steps))
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel table: the same steady-state steps once more, every kernel bracketed by HIP events (not in the timed
    # region: two event records per kernel would perturb it)
    n_prof = max(1, min(args.steps, 20))
    replays = getattr(tstep, "graph_replays", 0) if step_graph else 0
    if step_graph:
        tstep.enable_graph(False)      # the per-kernel table needs the eager launches (events cannot go inside a replay)
    with KernelProfile() as prof_all:
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
    breakdown = prof_all.result
    per_step = {k: v["total_ms"] / n_prof for k, v in breakdown.items()}
    dominant = max((k for k in per_step if k != "memset"), key=lambda k: per_step[k])
    sort_names = ("radix_count_kernel", "radix_scan_kernel", "radix_scatter_kernel")

    dist_extra = None
    if use_dist and args.mode == "raster" and not args.no_extras:
        try:
            dist_extra = dist_blocks(world, rank, dev, sharded_opt, args.force_dist)
        except Exception as e:  # noqa: BLE001  (every rank takes the same path: a failure is one of construction, not of one rank)
            dist_extra = {"error": f"{type(e).__name__}: {e}"}
    if args.mode != "raster" and tstep is not None:
        # a training step whose forward outgrew the resident capacity is dropped on the device and run again at the top of the NEXT
        # call (by every rank together): resolve the last one here, on every rank, before anything is reported
        tstep.finish()
    if rank == 0:
        eng.check(raise_on_overflow=args.mode == "raster")     # (the static raster workload can never outgrow its calibration)
        # The reference's num_rendered R (bounding-square duplication, from the calibrating reference-shaped forward) prices
        # the reference's work; the resident forward bins fewer instances (tight rectangles) and drops the dead ones in the
        # first tile-id pass, so its tile kernels walk `instances_live`: the dominant kernel's roofline is priced on THOSE.
        R_ref = eng.R_reference or eng.R
        R_live = getattr(eng, "R_live", 0) or eng.R
        P_vis = int((eng.radii > 0).sum().item())
        gx, gy = (cam.width + 15) // 16, (cam.height + 15) // 16
        passes_tile = (max(1, int(gx * gy - 1).bit_length()) + 7) // 8
        passes = int(round(breakdown.get("radix_scatter_kernel", {"launches": 0})["launches"] / n_prof))
        passes_depth = max(passes - passes_tile, 0)
        ab_ref = algorithmic_bytes(eng.P_active, P_vis, R_ref, cam.width, cam.height, passes_depth, passes_tile)
        ab_live = algorithmic_bytes(eng.P_active, P_vis, R_live, cam.width, cam.height, passes_depth, passes_tile)
        live = prof_live.result.get(dominant)
        dom = live if live else breakdown[dominant]
        dom_src = "HIP events inside the timed region" if live else f"HIP events over {n_prof} steady-state steps after the timed region"

        def dom_bytes(ab):
            return ab[dominant] if dominant in ab else ab["radix_sort(all passes)"] / (3 * max(passes, 1))
        t_dom = dom["avg_ms"] * 1e-3
        achieved = dom_bytes(ab_live) / t_dom / 1e9
        pmc, pmc_src = pmc_entry(args.workload, dominant) if args.mode == "raster" else (None, None)
        total_bytes = sum(ab_ref.values())
        ms_per_step = elapsed / args.steps * 1e3
        raster_ms = sum(per_step.values())
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
            "config": {"workload": f"{args.workload}: {eng.P_active} Gaussians, {cam.width}x{cam.height}, 1 keyframe per GPU, "
                                   "fwd+bwd raster" + ((", RCCL all-reduce of parameter grads" if not raster_ex.sharded else
                                                        ", RCCL reduce-scatter of parameter grads + all-gather of the parameter bucket")
                                                       if world > 1 and tstep is None else "")
                                   + (" + L1/SSIM loss + fused Adam" if args.mode == "trainer" else "")
                                   + (f"; anchor-level mapper step: {args.anchors} anchors x 10 offsets -> neural Gaussians "
                                      f"(appearance_dim {args.appearance_dim}, feature bank {'off' if args.no_feat_bank else 'on'}; MLPs fwd+bwd), "
                                      "L1/SSIM, fused Adam" if args.mode == "scaffold" else "")
                                   + ((", dense all-reduce" if not tstep._exchange_sharded() else ", reduce-scatter -> sharded Adam -> all-gather")
                                      if (world > 1 and tstep is not None) else ""),
                       "P": eng.P_active, "P_visible": P_vis, "num_rendered": R_ref, "instances_binned": eng.R, "instances_live": R_live,
                       "width": cam.width, "height": cam.height,
                       "sort_passes": {"depth_keys_P": passes_depth, "tile_keys_R": passes_tile}, "parallelism": f"keyframe-dp{world}",
                       "forward": "sync (reference API)" if args.sync_forward else "resident (no host sync)",
                       "hipgraph": (f"whole iteration replayed from a captured hipGraph ({replays} replays)" if step_graph else
                                    ("resident fwd+bwd replayed from a captured hipGraph" if graph is not None else "off"))},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc.get("hbm_bytes") if pmc else None, "traffic_source": pmc_src,
                         "traffic_stale": pmc.get("stale") if pmc else None,
                         "algorithmic_bytes_per_launch": dom_bytes(ab_live), "priced_on": "instances_live (what the launch walks)",
                         "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"], "timing": dom_src,
                         "on_reference_num_rendered": {"algorithmic_bytes_per_launch": dom_bytes(ab_ref),
                                                       "achieved": dom_bytes(ab_ref) / t_dom / 1e9,
                                                       "frac": dom_bytes(ab_ref) / t_dom / 1e9 / HBM_PEAK_GBPS},
                         "note": "the tile kernels are VALU-/atomic-bound, not HBM-bound (SURVEY 8d): see roofline_valu, roofline_atomic"},
            "roofline_valu": None, "roofline_atomic": None,
            "raster": {"algorithmic_bytes": total_bytes, "algorithmic_bytes_live": sum(ab_live.values()), "kernel_ms_sum": raster_ms,
                       "kernel_ms_source": f"HIP events over {n_prof} steady-state steps after the timed region",
                       "achieved_GBps": total_bytes / (raster_ms * 1e-3) / 1e9,
                       "whole_step_GBps": total_bytes / (ms_per_step * 1e-3) / 1e9,
                       "whole_step_frac_of_hbm_peak": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "kernel_ms": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
                       "radix_sort_ms": round(sum(per_step.get(k, 0.0) for k in sort_names), 4)},
        }
        if pmc and "SQ_INSTS_VALU" in pmc:
            a = pmc["SQ_INSTS_VALU"] / t_dom / 1e9
            out["roofline_valu"] = {"bound": "valu", "kernel": dominant, "achieved": a, "peak": VALU_PEAK_GINST,
                                    "unit": "G wave-instructions/s", "frac": a / VALU_PEAK_GINST,
                                    "wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "source": pmc_src, "stale": pmc.get("stale"),
                                    "note": "SQ_INSTS_VALU per launch / launch time against 1024 SIMDs x 2.4 GHz / 2 cycles per "
                                            "wave64 instruction; v_exp/v_rcp/permlane take 8 and v_cndmask/v_cmp/DPP 4 cycles"}
        if pmc and "write_bytes" in pmc:
            a = pmc["write_bytes"] / t_dom / 1e9
            out["roofline_atomic"] = {"bound": "global float atomics", "kernel": dominant, "achieved": a, "peak": ATOMIC_PEAK_GBPS,
                                      "unit": "GB/s", "frac": a / ATOMIC_PEAK_GBPS, "write_bytes_per_launch": pmc["write_bytes"],
                                      "algorithmic_atomic_bytes": 36 * R_live, "source": pmc_src, "stale": pmc.get("stale")}
        try:   # the tile kernels are reported in (pixel, Gaussian) pairs per second next to the byte figure
            out["raster"].update(pair_rates(eng, cam, per_step))
        except Exception as e:  # noqa: BLE001  (measurement garnish only; never fail the bench line over it)
            out["raster"]["pairs_error"] = str(e)
        pct = lambda q: per_step_ms[min(len(per_step_ms) - 1, int(q * len(per_step_ms)))]  # noqa: E731
        out["step_ms"] = {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90), "min": per_step_ms[0], "max": per_step_ms[-1],
                          "source": "one HIP event per step on the launch stream, inside the timed region (device time between "
                                    "step ends; the line's ms_per_step is the host clock over all K steps)"}
        if world == 1 and args.mode == "raster" and not args.no_extras:
            try:
                out["dropin"] = dropin_block(sc, cpp_dropin, world * args.steps / elapsed)
            except Exception as e:  # noqa: BLE001  (extra blocks never fail the bench line)
                out["dropin"] = {"error": f"{type(e).__name__}: {e}"}
            extras = [("render_only_ms", lambda: render_only(eng, (bg, m3, col, op, sca, rot, view, proj, campos, cam.tanfovx, cam.tanfovy))),
                      ("mapper_step", lambda: mapper_step_block(dev)), ("replica_step", lambda: replica_step_block(dev)),
                      ("config3", lambda: config3_block(dev)),
                      ("config3_reference_schedule", lambda: config3_reference_schedule_block(dev))]
            for key, fn in extras:
                try:
                    out[key] = fn()
                except Exception as e:  # noqa: BLE001  (extra blocks never fail the bench line)
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
        if dist_extra is not None:
            out["keyframe_parallel"] = dist_extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc)
        if args.breakdown:
            for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]):
                print(f"  {k:32s} {v:8.4f} ms/step", file=sys.stderr)
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpp_dropin_block(sc, steps: int = 50, warmup: int = 10):
    """fwd+bwd iterations per second of the headline workload through the C++ drop-in: GaussianRasterizer::forward and the
    autograd backward of csrc/torch_boundary/libgaussian_rasterizer.so, driven by `boundary_test --bench` (a child process;
    the scene travels in a temporary file).  Twice: default flags (the reference's R / point_list / ranges bit for bit) and
    SEGS_RASTER_TIGHT_BINNING (segs_raster.h)."""
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "segs-slam_amd", "csrc", "torch_boundary", "boundary_test")
    if not os.path.exists(exe):
        return {"error": "csrc/torch_boundary/boundary_test is not built"}
    cam = sc.camera
    out = {"what": "GaussianRasterizer::forward + autograd backward (C++/LibTorch-ROCm drop-in, reference signatures), fresh outputs and "
                   "allocator-grown scratch per call, one host synchronisation on num_rendered per forward; timed by boundary_test --bench "
                   f"in a child process, {steps} steps after {warmup}"}
    with tempfile.TemporaryDirectory(prefix="segs_bench_") as d:
        path = os.path.join(d, "scene.bin")
        with open(path, "wb") as f:
            np.array([sc.P, cam.width, cam.height], np.int32).tofile(f)
            np.array([cam.tanfovx, cam.tanfovy], np.float32).tofile(f)
            for a in (sc.bg, sc.means3D, sc.colors, sc.opacity, sc.scales, sc.rotations, cam.world_view_transform,
                      cam.full_proj_transform, cam.camera_center, sc.dL_dout_color):
                np.ascontiguousarray(a, np.float32).tofile(f)
        py = {"what": "segs_slam_amd.rasterize_points.RasterizeGaussiansCUDA + RasterizeGaussiansBackwardCUDA (mirror of "
                      "src/rasterize_points.cu:36-193 over the C ABI), timed by tools/dropin_python.py in a child process of its own per variant"}
        for label, flags in (("reference_lists", 0), ("tight_binning", 32)):
            for dst, cmd in ((out, [exe, "--bench", path, str(steps), str(warmup), str(flags)]),
                             (py, [sys.executable, os.path.join(ROOT, "tools", "dropin_python.py"), path, str(steps), str(warmup), str(flags)])):
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    dst[label] = json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
                except Exception as e:  # noqa: BLE001
                    dst[label] = {"error": f"{type(e).__name__}: {e}"}
    return {"cpp": out, "python": py}


def dropin_block(sc, children, resident_its):
    """What an unchanged SEGS-SLAM gets by swapping the libraries: fwd+bwd iterations per second of the headline workload through
    the reference-shaped entry points (RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA: fresh output tensors and scratch
    per call, one host synchronisation on num_rendered per forward) -- (i) the Python mirror of src/rasterize_points.cu over the
    C ABI, (ii) the C++/LibTorch-ROCm library -- each with the reference's lists and with SEGS_RASTER_TIGHT_BINNING, each variant
    in a child process of its own (cpp_dropin_block: a process that calls nothing else, like the mapper); next to the resident
    path the headline `value` is measured on."""
    cam = sc.camera
    return {"workload": f"{sc.P} Gaussians, {cam.width}x{cam.height} (the headline workload), fwd+bwd raster",
            "resident_path_iters_per_s": resident_its,
            "python_reference_shaped": (children or {}).get("python"), "cpp_gaussian_rasterizer": (children or {}).get("cpp")}


def _timed_dist_steps(step_fn, ex_of, world, dev, steps, warmup):
    """`steps` keyframe-parallel iterations between two barriers (max over ranks), every collective of the exchange bracketed by
    events on the launch stream.  -> (wall seconds, {collective: {ms per step, bytes per call, bus GB/s}}, exchange ms per step)."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    ex = ex_of()
    ex.timing = []
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tt = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    per = {}
    for label, a, b, nbytes in ex.timing:
        e = per.setdefault(label, {"ms": 0.0, "calls": 0, "bytes_per_call": nbytes})
        e["ms"] += a.elapsed_time(b)
        e["calls"] += 1
    ex.timing = None
    total_ms = 0.0
    for label, e in per.items():
        ms = e["ms"] / e["calls"]
        total_ms += e["ms"] / steps
        # bus bandwidth as rccl-tests defines it: all-reduce moves 2 (N-1)/N of the buffer per rank, the two halves (N-1)/N
        factor = (2.0 if label == "all_reduce" else 1.0) * (world - 1) / max(world, 1)
        per[label] = {"ms_per_call": ms, "calls_per_step": e["calls"] / steps, "bytes_per_call": e["bytes_per_call"],
                      "bus_GBps": (e["bytes_per_call"] * factor / (ms * 1e-3) / 1e9) if (ms > 0 and world > 1) else None}
    return float(tt.item()), per, total_ms


def dist_blocks(world, rank, dev, sharded_opt, force, steps: int = 40, warmup: int = 8):
    """N > 1 (or --force-dist): BASELINE's two 8-GPU configurations as blocks of the same line, so that the first scaling run
    records them and not only the raster headline's 168 MB worst case -- config 5 (anchor-level mapper step, 300 k anchors x 10
    offsets, 1200x680, ScanNet model dimensions) and config 4 (trainer step over explicit Gaussians, 640x480, TUM intrinsics),
    one keyframe per rank, the gradient exchange timed by events around its collectives."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from segs_slam_amd import neural_gaussians as ng, scenes
    from segs_slam_amd.gaussian_trainer import TrainerStep, keyframe_tensors
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    out = {"world_size_reported_by_backend": dist.get_world_size(), "backend": dist.get_backend()}
    # ---- config 5
    cam = scenes.make_config_camera("c5", keyframe=rank)
    dims = ng.ModelDims(appearance_dim=16, use_feat_bank=False)
    model = ng.synthetic_model(300_000, dims, scenes.make_config_camera("c5"), dev, seed=0)     # identical replicas
    tstep = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    tstep.sharded_optimizer, tstep.single_rank_collectives = sharded_opt, force
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    gt = torch.rand(3, cam.height, cam.width, device=dev)
    tstep.keyframe_for = lambda step, n: 0
    wall, per, ex_ms = _timed_dist_steps(lambda: tstep.training_once([kf], [gt]), tstep._exchange, world, dev, steps, warmup)
    tstep.finish()                     # (a drop of the last iteration is run again here, by every rank)
    ex = tstep._exchange()
    out["mapper_step_c5"] = {"workload": f"anchor-level mapper step, {model.A} anchors x 10 offsets, {cam.width}x{cam.height}, appearance_dim 16, "
                                         "no feature bank, one keyframe per rank",
                             "iters_per_s": world * steps / wall, "ms_per_step": wall / steps * 1e3, "steps": steps, "warmup": warmup,
                             "exchange": "reduce-scatter -> sharded Adam -> all-gather" if ex.sharded else "dense all-reduce",
                             "exchanged_MB": ex.n * 4 / 1e6, "bucket_MB": (ex.offset + ex.n) * 4 / 1e6,
                             "frozen_anchor_segment_left_out": ex.offset > 0, "exchange_ms_per_step": ex_ms, "collectives": per,
                             "dropped_steps": tstep.dropped_steps(), "redone_steps": tstep.redone_steps, "lost_steps": tstep.lost_steps()}
    del tstep, model
    torch.cuda.empty_cache()
    # ---- config 4
    sc = scenes.make_config_scene("c4", keyframe=rank)
    ts4 = TrainerStep.on_gpu(sc, dev, sharded_optimizer=sharded_opt, single_rank_collectives=force)
    kfs, gts = [keyframe_tensors(sc.camera, dev)], [torch.rand(3, sc.camera.height, sc.camera.width, device=dev)]
    ts4.keyframe_for = lambda step, n: 0
    wall, per, ex_ms = _timed_dist_steps(lambda: ts4.training_once(kfs, gts), lambda: ts4.exchange, world, dev, steps, warmup)
    ts4.finish()
    out["trainer_step_c4"] = {"workload": f"trainer step (raster + L1/SSIM + fused Adam), {sc.P} Gaussians, {sc.camera.width}x{sc.camera.height}, one keyframe per rank",
                              "iters_per_s": world * steps / wall, "ms_per_step": wall / steps * 1e3, "steps": steps, "warmup": warmup,
                              "exchange": "reduce-scatter -> sharded Adam -> all-gather" if ts4.exchange.sharded else "dense all-reduce",
                              "exchanged_MB": ts4.exchange.n * 4 / 1e6, "exchange_ms_per_step": ex_ms, "collectives": per}
    return out


def _percentiles(ms):
    ms = sorted(ms)
    q = lambda f: ms[min(len(ms) - 1, int(f * len(ms)))]  # noqa: E731
    return {"mean": sum(ms) / len(ms), "p10": q(0.1), "p50": q(0.5), "p90": q(0.9)}


def render_only(eng, fwd_args, iters: int = 50):
    """Forward-only render of the headline workload through the resident path: the reference's own per-frame figure
    (render_time.txt of renderAndRecordKeyframe, src/gaussian_mapper.cpp:1791-1808), HIP events per call."""
    import torch
    for _ in range(3):
        eng.forward(*fwd_args)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        eng.forward(*fwd_args)
        ev[i + 1].record()
    torch.cuda.synchronize()
    eng.check()
    out = _percentiles([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)])
    out["what"] = f"forward-only raster (preprocess, binning, render), resident path, {iters} frames, HIP events per frame"
    return out


def _timed_steps(tstep, kfs, gts, steps):
    """`steps` training_once calls: (iters/s by wall clock, ms per step, percentiles of one HIP event per step)."""
    import torch
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        tstep.training_once(kfs, gts)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return steps / wall, wall / steps * 1e3, _percentiles([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])


def _warm_to_plateau(tstep, kfs, gts, chunk: int = 20, min_iters: int = 60, max_iters: int = 400, tol: float = 0.005):
    """Train until the number of candidates with a positive neural opacity -- the reference's compacted P, which the random
    target of these blocks drives up over the first iterations, and with it every per-Gaussian kernel's work -- has stopped
    moving: less than `tol` between two chunks, at least `min_iters` iterations.  Returns (iterations run, live count trail)."""
    import torch
    trail, done = [int(tstep.neural.mask().sum().item())], 0
    while done < max_iters:
        for _ in range(chunk):
            tstep.training_once(kfs, gts)
        done += chunk
        torch.cuda.synchronize()
        trail.append(int(tstep.neural.mask().sum().item()))
        if done >= min_iters and abs(trail[-1] - trail[-2]) <= tol * max(trail[-1], 1):
            break
    return done, trail


def mapper_step_block(dev, steps: int = 50, warmup: int = 10):
    """The real mapper iteration at BASELINE config 5's size, untimed-region block of the bench line: 300 k anchors x 10
    offsets (3 M candidate Gaussians) at 1200x680, ScanNet model dimensions (appearance_dim 16, no feature bank):
    prefilter + neural-Gaussian MLPs fwd/bwd + raster fwd/bwd + L1/SSIM + fused Adam."""
    import numpy as np
    import torch
    from segs_slam_amd import neural_gaussians as ng, scenes
    cam = scenes.make_config_camera("c5")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    dims = ng.ModelDims(appearance_dim=16, use_feat_bank=False)
    model = ng.synthetic_model(300_000, dims, cam, dev, seed=0)
    tstep = ng.ScaffoldTrainerStep(model, cam.width, cam.height)
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    gt = torch.rand(3, cam.height, cam.width, device=dev)
    tstep.keyframe_for = lambda step, n: 0
    # ONE number, ONE state: the bench trains on a random target, under which the live candidates climb over the first ~40
    # iterations and the step with them.  `early` = the window of rounds 2-4 (10 warm-up + 50 timed steps, a moving state);
    # the block's own figures are taken at the PLATEAU (live count steady to 0.5 % over 20 iterations).
    for _ in range(warmup):
        tstep.training_once([kf], [gt])
    e_ips, e_ms, e_pc = _timed_steps(tstep, [kf], [gt], steps)
    early = {"iters_per_s": e_ips, "ms_per_step": e_ms, "step_ms": e_pc, "warmup": warmup, "steps": steps,
             "neural_opacity_positive_after": int(tstep.neural.mask().sum().item())}
    plateau_iters, trail = _warm_to_plateau(tstep, [kf], [gt])
    ips, ms, pc = _timed_steps(tstep, [kf], [gt], steps)
    wall = steps / ips
    phases = tstep.profile_phases(kf, gt, 20)
    for _ in range(3):
        tstep.render(kf)
    rev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    rev[0].record()
    for i in range(20):
        tstep.render(kf)
        rev[i + 1].record()
    torch.cuda.synchronize()
    eng = tstep.engine
    eng.check()
    # how much of the candidate domain is alive: rows the rasterizer bins (radius > 0 after the opacity skip and the frustum
    # cull) and rows whose neural opacity is positive (the reference's compacted P) out of anchors x offsets candidate rows
    n_binned = int((eng.radii[:eng.P_active] > 0).sum().item())
    n_kept = int(tstep.neural.mask().sum().item())
    n_visible_anchors = int((tstep.visible_radii[:model.A] > 0).sum().item())
    live = {"candidate_rows": eng.P_active, "visible_anchors": n_visible_anchors, "neural_opacity_positive": n_kept,
            "binned_by_the_rasterizer": n_binned, "binned_fraction_of_candidates": n_binned / max(eng.P_active, 1)}
    return {"workload": f"anchor-level mapper step, config-5 size: {model.A} anchors x {dims.n_offsets} offsets -> {eng.P_active} candidate "
                        f"Gaussians, {cam.width}x{cam.height}, appearance_dim 16, no feature bank; instances binned {eng.R}, live {eng.R_live}",
            "state": f"plateau: live candidates steady to 0.5 % ({trail[-2]} -> {trail[-1]}) after {warmup + steps + plateau_iters} iterations on the block's random target",
            "iters_per_s": steps / wall, "ms_per_step": wall / steps * 1e3, "steps": steps, "warmup": warmup + steps + plateau_iters, "step_ms": pc,
            "phase_ms": {k: round(v, 4) for k, v in phases.items()},
            "render_only_ms": _percentiles([rev[i].elapsed_time(rev[i + 1]) for i in range(20)]),
            "dropped_steps": tstep.dropped_steps(), "redone_steps": tstep.redone_steps, "candidates": live,
            "neural_opacity_positive_trail": trail, "early": early}


def replica_step_block(dev, anchors: int = 50_000, steps: int = 50, warmup: int = 10, iteration: int = 10_000,
                       variants=(("fused", True), ("autograd_mirror", False))):
    """The Replica mapper iteration with its frequency regulariser ON, untimed-region block: the step
    cfg/gaussian_mapper/RGB-D/Replica/office0.yaml describes (mapper_config.make_mapper_step: feature bank, appearance_dim 32,
    scaling regulariser, row mask, multi_scale_loss over 3 scales with lambda 0.01 between iterations 5 000 and 25 500,
    densification statistics every iteration) at an iteration inside that window, 1200x680, `anchors` x 10 offsets (BASELINE
    config 2's ~500 k Gaussians).  Timed twice: with the fused frequency path (csrc/freq_loss.hip + cached |FFT(gt)| + one
    rfft2 / irfft2 per scale) and with the torch.fft + autograd mirror of the reference's op chain it replaces."""
    import numpy as np
    import torch
    from segs_slam_amd import mapper_config as mc, neural_gaussians as ng, scenes
    cfg = mc.load_committed_config("cfg/gaussian_mapper/RGB-D/Replica/office0.yaml")
    cam = scenes.make_config_camera("c2")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kf = ng.Keyframe(t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center),
                     torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev), cam.tanfovx, cam.tanfovy)
    gt = torch.rand(3, cam.height, cam.width, device=dev)
    out = {"workload": f"Replica office0 mapper step (cfg values of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml), iteration {iteration}+: "
                       f"{anchors} anchors x 10 offsets, {cam.width}x{cam.height}, feature bank on, appearance_dim {cfg.model.appearance_dim}, "
                       f"multi_scale_loss scales {list(cfg.scales)} lambda {cfg.lambda_frequency_high}, training_statis every iteration",
           "steps": steps, "warmup": warmup}
    for label, fused in variants:
        model = ng.synthetic_model(anchors, cfg.model, cam, dev, seed=0)
        tstep = mc.make_mapper_step(cfg, model, cam.width, cam.height)
        assert tstep.freq_reg is not None
        tstep.freq_reg["fused"] = fused
        tstep.keyframe_for = lambda step, n: 0
        # between two adjust_anchor iterations of the cfg's schedule (every 100 from 1 500): the timed steps hold none
        tstep.iteration = iteration
        # ONE state: the statistics run every iteration as the cfg says, but adjust_anchor (every 100 from 1 500) is held off for the
        # whole block -- on the block's random target it would grow the map threefold during the plateau warm-up, and the figure
        # would no longer be "the Replica step at `anchors` anchors"
        if tstep.densifier is not None:
            tstep.densifier.p.update_from = 10 ** 9
        for _ in range(warmup):
            tstep.training_once([kf], [gt])
        e_ips, e_ms, e_pc = _timed_steps(tstep, [kf], [gt], steps)          # rounds 3-4's window: a moving state
        plateau_iters, trail = _warm_to_plateau(tstep, [kf], [gt], max_iters=200)
        ips, ms, pc = _timed_steps(tstep, [kf], [gt], steps)
        wall = steps / ips
        low_on, high_on = tstep._freq_active()
        phases = tstep.profile_phases(kf, gt, 20)
        tstep.engine.check()
        out[label] = {"state": f"plateau: live candidates steady to 0.5 % ({trail[-2]} -> {trail[-1]}) after {warmup + steps + plateau_iters} iterations",
                      "iters_per_s": steps / wall, "ms_per_step": wall / steps * 1e3, "step_ms": pc,
                      "early": {"iters_per_s": e_ips, "ms_per_step": e_ms, "step_ms": e_pc, "warmup": warmup, "steps": steps},
                      "anchors": tstep.model.A, "neural_opacity_positive_trail": trail,
                      "phase_ms": {k: round(v, 4) for k, v in phases.items()}, "frequency_terms_on": {"low": low_on, "high": high_on},
                      "dropped_steps": tstep.dropped_steps(), "redone_steps": tstep.redone_steps, "instances_binned": tstep.engine.R}
    return out


def config3_block(dev, max_iters: int = 3000, target_anchors: int = 200_000, steady_iters: int = 200):
    """SURVEY 8d config 3, untimed-region block: the mapper loop with statistics + anchor growing + pruning on the synthetic
    64-keyframe orbit (segs_slam_amd.config3) from ~50 k anchors until anchors x 10 ~ 2 M Gaussians, then `steady_iters` more
    iterations at that size.  The Replica gradient threshold (2e-4) grows this synthetic map to ~64 k anchors and stops
    (profiles/r03_config3_growth.txt); the block runs the SAME loop with the threshold at 2e-5 so that the sizes SURVEY names
    are reached inside a bounded run -- stated in `densify_grad_threshold`."""
    import torch
    from segs_slam_amd import config3, densify
    params = densify.DensifyParams(start_stat=100, update_from=300, update_interval=100, update_until=10 ** 9,
                                   densify_grad_threshold=2e-5)
    run = config3.Config3Run(dev, params=params)
    res = run.run(max_iters, target_anchors)
    params.update_until = 0            # steady state at the final size: statistics and adjust_anchor off
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steady_iters):
        run.step.training_once(run.keyframes, run.targets)
    torch.cuda.synchronize()
    steady = steady_iters / (time.perf_counter() - t0)
    finite = bool(torch.isfinite(run.model.params).all())
    res.update({"densify_grad_threshold": params.densify_grad_threshold, "schedule": "statistics from iteration 100, adjust_anchor from 300 every 100",
                "keyframes": len(run.keyframes), "image": f"{run.cam.width}x{run.cam.height}", "iters_per_s_at_final_size": steady,
                "parameters_finite": finite,
                "what": "wall clock over all iterations of the growth run, adjust_anchor iterations and their host synchronisations included"})
    return res


def config3_reference_schedule_block(dev, iters: int = 2200):
    """Config 3's loop at the REFERENCE's own densification values (cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:130-137:
    statistics from iteration 500, adjust_anchor from 1 500 every 100, gradient threshold 2e-4 -- the value at :137; OpenCV's
    FileNode::operator[] returns the first of the file's two definitions, 1e-3 at :91, under which even fewer anchors grow).
    On this synthetic teacher the map stalls near 64 k anchors under that threshold (profiles/r03_config3_growth.txt), which is why
    the `config3` block lowers it to reach the sizes SURVEY names; this block prints the stall next to it: same loop, same
    teacher, the reference's schedule, `iters` iterations (seven adjust_anchor calls)."""
    import torch
    from segs_slam_amd import config3, densify
    params = densify.DensifyParams(start_stat=500, update_from=1500, update_interval=100, update_until=10 ** 9,
                                   densify_grad_threshold=2e-4)
    run = config3.Config3Run(dev, params=params)
    res = run.run(iters, 10 ** 9)
    res.update({"densify_grad_threshold": params.densify_grad_threshold,
                "schedule": "statistics from iteration 500, adjust_anchor from 1500 every 100 (office0.yaml:130-137)",
                "parameters_finite": bool(torch.isfinite(run.model.params).all()),
                "what": "the reference's densification hyper-parameters on the synthetic teacher: the map barely grows (the stall the config3 block's "
                        "lower threshold avoids); wall clock over all iterations, adjust_anchor iterations included"})
    return res


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


def pmc_entry(workload, kernel):
    """Per-launch counters of `kernel` from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE, WRITE_SIZE
    and the SQ counters need separate passes, so they cannot be collected inside the timed run): the newest
    profiles/rNN_pmc_<workload>.json, written by tools/collect_profiles.sh + tools/pmc_summary.py (FETCH_SIZE doubled per the
    gfx950 correction; the calibrating first step is left out)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{workload}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        tab = json.load(f)
    e = tab.get("segs::" + kernel)
    if not e:
        return None, None
    # counters cannot be collected inside the timed run; say so when they were taken from another state of the kernel sources
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import kernel_source_sha
    e = dict(e, stale=tab.get("_kernel_source_sha") != kernel_source_sha(ROOT))
    return e, os.path.relpath(files[-1], ROOT)


def cpu_baseline(sc, budget_s: float = 15.0, max_iters: int = 10):
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
