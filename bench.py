#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the wavefront path tracer on the reference's sample scene at 1920x1080.

One "step" = `--spp-per-step` samples per pixel over the whole 1080p image on every GPU (the reference's frame is
1 spp: src/window.rs:166).  The timed region runs the library as it ships (its own choice of passes in flight and pass
size); on one GPU a second leg repeats the steps with ONE pass in flight and per-kernel HIP events, for the `roofline`
block (kernels of two passes that overlap each report the time they shared the GPU), and a third runs the reference's
window loop (`frame_loop`).  `--scaling strong` fixes the total work instead (3840x2160, --total-spp split over the GPUs).
With N > 1 there are two ways to drive the GPUs, both sample-index shards with ONE reduction of the u32 accumulators
inside the timed region:
  * launched by torch.distributed.run, one rank per GPU (the default): every rank owns a PtGpu handle, the
    accumulators are summed with an RCCL all-reduce issued through torch.distributed;
  * `--inproc`: ONE process, the library's own PtGroup (one handle + host thread per device, ncclReduce inside
    libptgpu) - what a single-process host like the reference's window loop uses.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
VALU_PEAK_GINST = 923.0  # wave64 v_fma_f32 issued per second chip-wide, measured (profiles/micro/valu_exec_mask.hip)


def source_hash() -> str:
    """Identifies the kernels a profile was taken from: a hash over the device sources.  Counter files under profiles/
    carry the hash of the tree they were collected with; figures taken from them are marked stale when it differs."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")) and f != "pt_group.h":   # (pt_group.h is host code only: no kernel depends on it)
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def host_info() -> dict:
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {"cpu_model": model, "host_nproc": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0))}
    try:   # cgroup v2 CPU quota of this job, in CPUs
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_quota"] = None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        info["cgroup_cpu_quota"] = None
    return info


def cpu_baseline(sc, budget_s: float, threads: int, note: str = "the same", warm: bool = True, hoist_plane_axes: int = 0):
    """The reference's CPU path as restated in oracle/ (8-thread 32x32 tile pool, xorshift RNG, platform libm),
    timed on a bounded number of 1-spp passes of the workload (`note` says how the sample relates to it).
    `hoist_plane_axes` times the variant that computes each plane's normalised axes once per pass instead of once per
    ray and plane as src/scene.rs:244-253 does (same image; a fairer CPU figure, reported beside the faithful one)."""
    from oracle import ptref as R
    osc = R.OracleScene(sc)
    opt = R.options(R.RNG_XORSHIFT, R.MATH_LIBM, hoist_plane_axes=hoist_plane_axes)
    import ctypes as C
    import numpy as np
    W, H = osc.width, osc.height
    p32 = np.zeros((H, W, 3), dtype=np.uint32)
    p8 = np.zeros((H, W, 3), dtype=np.uint8)
    bb = R.Backbuffer(W, H, 1, p32.ctypes.data, p8.ctypes.data)
    stats = R.Stats()
    L = R.lib()
    if warm:
        L.ptref_render_pass_pool(C.byref(osc.desc), C.byref(opt), C.byref(bb), threads, 0, C.byref(R.Stats()))  # warm-up pass
    t0 = time.perf_counter()
    passes = 0
    while True:
        L.ptref_render_pass_pool(C.byref(osc.desc), C.byref(opt), C.byref(bb), threads, passes + 1, C.byref(stats))
        bb.num_samples += 1
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 64:
            break
    return {
        "value": round(stats.rays / dt / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"{passes} passes of 1 spp over {note} scene at {W}x{H} ({stats.rays} rays, {dt:.1f} s), "
                  f"{threads}-thread 32x32 tile pool as src/window.rs:110,167",
        "ms_per_pass": round(1e3 * dt / passes, 2),
    }


def frame_loop_figures(make, sc, frames: int, spp_per_frame: int = 1) -> dict:
    """The reference's own usage (src/window.rs:127-229): `spp_per_frame` samples per pixel per frame (the reference: 1) and
    the display bytes handed to the window every frame.  `make()` returns a fresh PtGpu or PtGroup.  Two loops over the
    same frames: `synchronous` = render + present, one after the other, as a host that waits for its frame runs them;
    `pipelined` = ptgpu_frame / ptgpu_group_frame (the frame's last pass writes the display bytes, the call returns the
    frame before).  Msamples/s = pixels x samples / wall time, host-side, present and readback included."""
    W, H = sc.image_settings.width, sc.image_settings.height
    out = {"frames": frames, "spp_per_frame": spp_per_frame, "resolution": f"{W}x{H}"}
    for mode in ("synchronous", "pipelined"):
        gpu = make()
        gpu.upload_scene(sc)
        step = (lambda: (gpu.render(spp_per_frame), gpu.present_pixels8())) if mode == "synchronous" else (lambda: gpu.frame(spp_per_frame))
        for _ in range(16):
            step()
        if mode == "pipelined":
            gpu.frame(0)
        gpu.sync()
        gpu.clear()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        if mode == "pipelined":
            _, shown = gpu.frame(0)      # the flush: the last frame has arrived on the host
            assert shown == frames * spp_per_frame, (shown, frames)
        dt = time.perf_counter() - t0
        out[mode] = {"ms_per_frame": round(1e3 * dt / frames, 4), "msamples_per_s": round(W * H * spp_per_frame * frames / dt / 1e6, 1)}
        gpu.close()
    return out


def profile_figures(workload: str, kernel: str, avg_launch_s: float, here: str):
    """HBM traffic and VALU figures of `kernel` from the committed counter run of the same command (profiles/), stamped
    with the source hash they were collected under; `stale` when the kernels have changed since."""
    tag = {"sample": "r03", "room": "r03_room", "quads": "r03_quads"}[workload]
    traffic, valu, stamp = None, None, None
    cpath = os.path.join(ROOT, "profiles", f"{tag}_counters.json")
    if not os.path.exists(cpath):
        return None, None, None
    try:
        doc = json.load(open(cpath))
        stamp = {"file": f"profiles/{tag}_counters.json", "source_hash": doc.get("_source_hash"), "commit": doc.get("_commit"),
                 "stale": doc.get("_source_hash") != here}
        ctr = doc.get(kernel, {})
        if "hbm_bytes_per_launch" in ctr:
            traffic = ctr["hbm_bytes_per_launch"]["total"]
        insts = ctr["SQ_INSTS_VALU"]["per_launch"]
        lanes = ctr["SQ_THREAD_CYCLES_VALU"]["per_launch"] / insts / 64.0
        rate = insts / avg_launch_s / 1e9
        valu = {"wave_insts_per_launch": round(insts), "achieved": round(rate, 1), "peak": VALU_PEAK_GINST,
                "unit": "G wave-instructions/s", "frac": round(rate / VALU_PEAK_GINST, 3), "lane_utilisation": round(lanes, 3),
                "profile": stamp}
    except Exception:
        pass
    return traffic, valu, stamp


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="8 steps of 512 spp: four times the 1024 spp the 1080p configuration is quoted on")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="0 = 512 (64 for the BVH scene): a step is >= 0.1 s, so that 8 steps are about a second of GPU time and the one "
                         "reduction of a multi-GPU run is small beside them; strong scaling: total-spp / steps over ALL GPUs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU renders spp-per-step per step (the work grows with N); strong: the sample scene at 3840x2160 "
                         "(unless --width/--height say otherwise), --total-spp samples per pixel IN TOTAL split over the GPUs by sample index")
    ap.add_argument("--total-spp", type=int, default=4096, help="strong scaling: samples per pixel of the whole job")
    ap.add_argument("--width", type=int, default=0, help="0 = 1920 (3840 for --scaling strong)")
    ap.add_argument("--height", type=int, default=0, help="0 = 1080 (2160 for --scaling strong)")
    ap.add_argument("--max-depth", type=int, default=0, help="0 = the workload's own (4; 16 for room)")
    ap.add_argument("--workload", choices=["sample", "quads", "room"], default="sample",
                    help="sample: the reference's scene (headline); quads: ~1M-rectangle height field (BVH); "
                         "room: closed diffuse room at max_depth 16 (queue compaction)")
    ap.add_argument("--grid", type=int, default=1000, help="quads: rectangles per side")
    ap.add_argument("--fused", type=int, default=-1)
    ap.add_argument("--accel", type=int, default=0, help="0 = library choice (BVH above 100 primitives), 1 = linear scan, 2 = BVH")
    ap.add_argument("--batch", type=int, default=0, help="samples per pixel per pipeline pass (0 = library default)")
    ap.add_argument("--passes", type=int, default=0,
                    help="pipeline passes in flight in the timed region; 0 = library default (2; BVH scenes: one large pass).  With 1 the timed "
                         "region itself carries the per-kernel events and supplies the roofline block (what profiles/collect.sh traces)")
    ap.add_argument("--inproc", action="store_true",
                    help="drive the --gpus devices from this one process through the library's PtGroup (in-library RCCL) "
                         "instead of one torch.distributed rank per GPU")
    ap.add_argument("--no-events", action="store_true", help="no second leg with per-kernel HIP events (no roofline block)")
    ap.add_argument("--frame-loop", type=int, default=300, metavar="FRAMES",
                    help="after the timed region: the window loop of the reference (1 spp per frame + the display bytes on the host every "
                         "frame, src/window.rs:127-229) for this many frames at 1080p and at 1024x1024, reported as `frame_loop`; 0 = skip")
    ap.add_argument("--spp-per-frame", type=int, default=1, help="samples per pixel per frame of the frame loop (the reference: 1)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from pathtracer_amd import binding, distributed as D, scene as S

    rank, world, local_rank = D.env_rank()
    dist = None
    torch = None
    if args.inproc:
        assert world == 1, "--inproc is a single-process mode: do not launch it under torch.distributed.run"
        n_gpus = max(1, args.gpus)
    else:
        import torch
        import torch.distributed as dist
        if os.environ.get("PTGPU_BENCH_ONE_DEVICE") == "1":
            # test configuration (one GPU at hand): every rank renders on device 0; RCCL refuses two ranks on one device, so
            # the exchange then goes over gloo - everything else is the code an N-GPU run executes
            local_rank = 0
        if world > 1 or os.environ.get("PTGPU_FORCE_COLLECTIVE"):
            torch.cuda.set_device(local_rank)
            backend = os.environ.get("PTGPU_BENCH_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        assert world == max(1, args.gpus) or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
        n_gpus = world
    use_dist = dist is not None and dist.is_initialized()

    strong = args.scaling == "strong"
    args.width = args.width or (3840 if strong else 1920)
    args.height = args.height or (2160 if strong else 1080)
    if args.workload == "sample":
        sc = S.sample_scene(args.width, args.height, camera_width=5.0 * args.width / args.height)
        wl = f"scenes/sample at {args.width}x{args.height} (camera.width {5.0 * args.width / args.height:.7f})"
        depth = args.max_depth or 4
    elif args.workload == "quads":
        sc = S.quad_field_scene(args.width, args.height, grid=args.grid, n_spheres=1000)
        wl = f"procedural {args.grid}x{args.grid} rectangle height field + 1000 spheres (seed 12345) at {args.width}x{args.height}, BVH"
        depth = args.max_depth or 4
    else:
        sc = S.diffuse_room_scene(args.width, args.height)
        wl = f"closed diffuse room, 27 primitives (seed 777) at {args.width}x{args.height}"
        depth = args.max_depth or 16
    args.max_depth = depth
    if strong:
        args.spp_per_step = args.spp_per_step or max(1, args.total_spp // args.steps)   # of the whole job, over all GPUs
        args.total_spp = args.spp_per_step * args.steps
    else:
        args.spp_per_step = args.spp_per_step or (64 if args.workload == "quads" else 512)
    events_in_timed_region = args.passes == 1 and not args.no_events
    cfg = dict(max_depth=args.max_depth, fuse_trace_shade=args.fused, samples_per_batch=args.batch, concurrent_passes=args.passes,
               accel=args.accel, profile_events=1 if events_in_timed_region else 0)
    lib = binding.load_library()

    if args.inproc:
        gpu = binding.PtGroup(n_devices=n_gpus, **cfg)
        # weak: every device renders spp_per_step per step; strong: the step's samples are split over the devices
        step_spp = args.spp_per_step if strong else args.spp_per_step * n_gpus

        def render_step(step):
            gpu.render(step_spp)

        def barrier():
            gpu.sync()

        def exchange():
            gpu.present_pixels8()                  # the reduction onto device 0 + the display bytes, as a window host ends a frame
            return args.steps * step_spp
    else:
        dev = torch.device("cuda", local_rank)
        gpu = binding.PtGpu(device=local_rank, shard_index=rank, shard_count=world, **cfg)

        def render_step(step):
            if strong:   # this rank's share of the step's global sample indices (the split rule of the library's own group)
                mine = lib.ptgpu_group_shard_samples(step * args.spp_per_step, args.spp_per_step, rank, world)
                if mine:
                    gpu.render(mine)
            else:
                gpu.render(args.spp_per_step)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        def exchange():
            mine = lib.ptgpu_group_shard_samples(0, args.total_spp, rank, world) if strong else args.steps * args.spp_per_step
            total = D.combine_accumulators(gpu, mine)
            torch.cuda.synchronize(dev)
            return total

    gpu.upload_scene(sc)
    for w in range(args.warmup):
        render_step(w)
    gpu.sync()
    if args.inproc:
        gpu.present_pixels8()                      # creates the RCCL communicators
    elif use_dist:
        D.combine_accumulators(gpu, args.warmup * args.spp_per_step)  # warms RCCL up too
    gpu.clear()

    barrier()
    t0 = time.perf_counter()
    for step in range(args.steps):
        render_step(step)
    gpu.sync()
    t_exchange = time.perf_counter()
    total_spp = exchange()
    t_exchange = time.perf_counter() - t_exchange   # the one reduction of the job, inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0

    st = gpu.stats()
    rays, samples = float(st["rays"]), float(st["samples"])
    if use_dist and world > 1:
        red = dev if dist.get_backend() == "nccl" else torch.device("cpu")   # (gloo: the one-device test configuration)
        agg = torch.tensor([rays, samples], dtype=torch.float64, device=red)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays, samples, elapsed = float(agg[0].item()), float(agg[1].item()), float(tmax[0].item())

    # second leg (one GPU): the same steps with ONE pass in flight, the pass size the library chose, and every kernel
    # bracketed by HIP events on the stream it is launched on
    one_pass = None
    exchange_device_ms = gpu.exchange_stats()["ms"] if args.inproc else None
    if events_in_timed_region:
        one_pass = {"stats": st, "elapsed": elapsed}
    elif not args.no_events and not strong and rank == 0 and (not args.inproc or n_gpus == 1):
        # (N > 1: rank 0 alone repeats its own shard's steps - the same launches per pass as at N = 1 - while the other
        # ranks wait at the closing barrier)
        gpu.close()
        gpu = binding.PtGpu(device=local_rank, shard_index=rank, shard_count=world,
                            **dict(cfg, concurrent_passes=1, samples_per_batch=st["samples_per_pass"], profile_events=1))
        try:
            gpu.upload_scene(sc)
            for _ in range(args.warmup):
                gpu.render(args.spp_per_step)
            gpu.sync()
            gpu.clear()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                gpu.render(args.spp_per_step)
            gpu.sync()
            one_pass = {"stats": gpu.stats(), "elapsed": time.perf_counter() - t1}
        except binding.PtGpuError as e:
            if world == 1:
                raise
            # (the headline of an N-GPU run is not to be lost to the extra leg: the line then goes out without the block)
            print(f"bench.py: one-pass leg failed on rank 0: {e}", file=sys.stderr, flush=True)

    if rank == 0:
        here = source_hash()
        pixels = args.width * args.height
        out = {
            "metric": f"Mrays/s (and sec-to-N-spp, field sec_to_1024_spp) at {args.width}x{args.height} on {'scenes/sample' if args.workload == 'sample' else 'the ' + args.workload + ' stress scene'}; ray = one nearest-hit query: primary + bounce + shadow",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl}, max_depth {args.max_depth}, " + (f"{args.total_spp} spp in total over all GPUs" if strong else f"{args.spp_per_step} spp per step per GPU"),
                       "spp_per_step": args.spp_per_step,
                       "parallelism": f"sample-index shards x{n_gpus}" + (", one process (PtGroup, in-library RCCL)" if args.inproc else ""),
                       "accumulated_spp": total_spp,
                       "passes_in_flight": st["passes_in_flight"], "samples_per_pass": st["samples_per_pass"],
                       "library_defaults": args.passes == 0 and args.batch == 0},
            "msamples_per_s": round(samples / elapsed / 1e6, 2),
            "sec_to_1024_spp": round(1024.0 * pixels / (samples / elapsed), 3),
            "rays_per_sample": round(rays / max(samples, 1.0), 3),
            "timed_region_s": round(elapsed, 3),
            "source_hash": here,
            "queue_gib": round(st["queue_bytes"] / 2**30, 2),
        }
        if strong:
            out["sec_to_total_spp"] = {"total_spp": args.total_spp, "seconds": round(elapsed, 4), "of_which_exchange_s": round(t_exchange, 4)}
        if one_pass is not None:
            ost = one_pass["stats"]
            # dominant kernel: the fused trace+shade launches (or trace, in the two-kernel pipeline)
            kms, kl = ost["kernel_ms"], ost["kernel_launches"]
            name = max(("trace_shade", "trace", "shade", "terminal", "gather", "resolve", "raygen"), key=lambda k: kms[k])
            traced = ost["rays"] - ost["shadow_rays"]          # records of the traced levels
            children = ost["records"] - ost["samples"]         # records appended to levels >= 1
            if name == "trace_shade":
                # per traced record: read ray 32 (not at level 0: primary rays are generated in the kernel),
                # write coef-or-leaf-value 16 (the node kind travels in its fourth word); per appended child: write
                # ray 32 + parent 4 (the 4-byte child link written by the records of even levels is not priced)
                alg = 48.0 * traced - 32.0 * ost["samples"] + 36.0 * children
            elif name == "trace":
                # read ray 32, write prim id 4; BVH: one 64-byte fetch per traversal step (inner node or leaf
                # primitive), counted by the kernel itself (deterministic per ray, independent of scheduling)
                alg = 36.0 * traced + 64.0 * ost["bvh_steps"]
            elif name == "shade":
                alg = 52.0 * traced + 36.0 * children   # read ray 32 + hit 4, write node 16
            else:
                alg = 0.0
            ms = kms[name]
            n_launch = max(1, kl[name])
            avg_s = ms * 1e-3 / n_launch
            traffic, valu, stamp = profile_figures(args.workload, name, avg_s, here)
            alg_rate = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            roof = {
                "bound": "hbm", "kernel": name,
                "achieved": round(alg_rate, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg_rate / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_profile": stamp,
                "launches": int(kl[name]), "avg_launch_ms": round(ms / n_launch, 5),
                "algorithmic_bytes_per_launch": round(alg / n_launch, 1),
                "kernel_ms": {k: round(v, 3) for k, v in kms.items() if v > 0},
                "measured_with": {"passes_in_flight": 1, "samples_per_pass": ost["samples_per_pass"],
                                  "mrays_per_s": round(ost["rays"] / one_pass["elapsed"] / 1e6, 2),
                                  "why": "kernels of two passes in flight overlap and each reports the time it shared the GPU; `value` is the "
                                         "library default, this leg repeats its steps with one pass in flight and per-kernel HIP events"},
            }
            if name == "trace":
                # The traversal's algorithmic bytes (64 per step) are mostly served by L1/L2: priced against HBM they can
                # exceed its peak, which is no roofline.  The fraction reported is what HBM saw (counter traffic of the
                # committed profile of this command over this run's launch time); the algorithmic rate stays as a field.
                roof["algorithmic_rate_gbs"] = roof["achieved"]
                hbm_rate = (traffic / avg_s / 1e9) if (traffic and avg_s > 0) else None
                roof["achieved"] = round(hbm_rate, 2) if hbm_rate is not None else None
                roof["frac"] = round(hbm_rate / HBM_PEAK_GBS, 4) if hbm_rate is not None else None
                roof["note"] = ("BVH traversal: `achieved`/`frac` are HBM-side (counter bytes per launch / launch time); "
                                "`algorithmic_rate_gbs` prices 64 B per traversal step wherever it is served from")
            out["roofline"] = roof
            if valu is not None:
                # what actually binds the fused kernel: vector instruction issue (see `limiter`)
                out["valu"] = valu
                out["roofline"]["limiter"] = "valu_issue" if valu["frac"] > (out["roofline"]["frac"] or 0.0) else "hbm"
        if use_dist or args.inproc:
            # the one exchange step of the job (this rank's wall time for it: waiting for the slowest rank included)
            out["exchange"] = {"op": ("ncclReduce(SUM, u32) onto device 0 + pixels8 readback (PtGroup)" if args.inproc
                                      else "all_reduce(SUM) of the u32 accumulator") + ", once per timed region",
                               "bytes": pixels * 12, "ms": round(1e3 * t_exchange, 3)}
            if args.inproc:
                out["exchange"]["device_ms"] = round(exchange_device_ms, 3)
        if st["bvh_nodes"]:
            out["bvh"] = {"nodes": st["bvh_nodes"], "leaf_prims": st["bvh_leaf_prims"], "depth": st["bvh_depth"],
                          "steps_per_traced_ray": round(st["bvh_steps"] / max(1, st["rays"] - st["shadow_rays"]), 2)}
        if args.frame_loop > 0 and (n_gpus == 1 or args.inproc) and not strong:
            # the reference's own usage, beside the bulk figure: the library defaults (passes in flight, pass size), no events
            fl_cfg = dict(max_depth=args.max_depth, accel=args.accel)
            if args.inproc:
                make = lambda: binding.PtGroup(n_devices=n_gpus, **fl_cfg)
            else:
                make = lambda: binding.PtGpu(device=local_rank, **fl_cfg)
            gpu.close()
            loops = [frame_loop_figures(make, sc, args.frame_loop, args.spp_per_frame)]
            if args.workload == "sample" and (args.width, args.height) != (1024, 1024):
                # the scene as the reference ships it: 1024 x 1024 (scenes/sample/sample.scene:2-3)
                loops.append(frame_loop_figures(make, S.sample_scene(1024, 1024), args.frame_loop, args.spp_per_frame))
            out["frame_loop"] = {"what": "render of spp_per_frame samples + display bytes on the host, every frame (src/window.rs:127-229); "
                                         "pipelined = ptgpu_frame, synchronous = ptgpu_render + ptgpu_present_pixels8",
                                 "loops": loops, "bulk_msamples_per_s": out["msamples_per_s"],
                                 "pipelined_over_bulk": round(loops[0]["pipelined"]["msamples_per_s"] / out["msamples_per_s"], 3)}
        if n_gpus == 1 and not args.no_cpu_baseline:
            host = host_info()
            if args.workload == "quads":
                # the reference scans every primitive per ray (src/scene.rs:263-297): ~1M tests per ray, so the
                # bounded sample is the same scene and camera at 128x64 pixels (8 full tiles, one per thread)
                small = S.quad_field_scene(128, 64, grid=args.grid, n_spheres=1000)
                out["cpu_baseline"] = cpu_baseline(small, 1.0, 8, "the same (linear scan, reduced image)", warm=False)
            else:
                out["cpu_baseline"] = cpu_baseline(sc, args.cpu_baseline_seconds, 8)
                # beside the faithful 8-thread figure (the reference's constant, src/window.rs:110): the same pool with
                # per-plane axes hoisted out of the ray loop, and the faithful code on every CPU this process may run on
                nproc = max(1, host["affinity_cpus"])
                hoisted = cpu_baseline(sc, args.cpu_baseline_seconds / 4, 8, hoist_plane_axes=1)
                wide = cpu_baseline(sc, args.cpu_baseline_seconds / 4, nproc)
                out["cpu_baseline"]["variants"] = {
                    "plane_axes_hoisted_8_threads": {"value": hoisted["value"], "unit": "Mrays/s", "cores": 8},
                    "faithful_nproc_threads": {"value": wide["value"], "unit": "Mrays/s", "cores": nproc}}
            out["cpu_baseline"].update(host)
        print(json.dumps(out), flush=True)

    gpu.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
