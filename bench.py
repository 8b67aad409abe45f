#!/usr/bin/env python3
"""Headline benchmark: rays/sec and ms/frame on scenes/teapot.json at 1920x1080.

One step = one frame, timed from ray generation to the framebuffer on the host: every GPU renders
its bands of 8 pixel rows of the image (scene replicated per GPU, built once and cloned) and copies
exactly those bands over its own PCIe link into one page-locked host frame; there is no GPU-to-GPU
exchange.  `python bench.py --gpus N` is one process that drives N GPUs with one native worker thread
per GPU (the host library's FrameLoop, the many-frame form of parallel_render, src/parallel.rs:183-204).
With fewer GPUs than N the parts are dealt round-robin over the GPUs there are (a rehearsal; the line
says so).  Launched by torch.distributed.run with one rank per GPU (WORLD_SIZE > 1), rank 0 runs that same loop over all N
GPUs, so that the whole frame lies in one page-locked host frame when the clock stops, and the other ranks take part
in the barriers and the MAX over the ranks' times (bench_launched; --per-rank-loops lets every rank render and copy
its own part into host memory of its own instead, --rccl-gather gathers the parts on rank 0 with RCCL).

A ray is one World::intersect query of the reference algorithm (primary, shadow, reflection,
refraction).  The scene stays resident in HBM; scene load and the BVH build are outside the timed
region and reported in `setup`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

FP64_VALU_PEAK = 78.6e12  # MI355X vector FP64, FLOP/s (datasheet; counts an FMA as two)
HBM_PEAK = 8.0e12         # bytes/s (datasheet)

# Algorithmic work of the generation kernel per traversal event.
# "survey": the coefficients the measurement plan of this build fixes (flops per 4-wide node visit,
# triangle test, quadric test, shaded hit); roofline.frac uses these.
# "f64": what this implementation's f64 arithmetic amounts to per event (a quadric test includes the 30
# operations of the ray transform, a gate the transform and three slab axes); the box tests of a node
# visit are single precision here and are counted apart, as f32 operations.
SURVEY = {"node": 22, "tri": 46, "quad": 60, "hit": 150}
F64 = {"tri": 46, "quad": 60 + 30, "gate": 36 + 30, "hit": 150}
F32_PER_NODE = 4 * 11
BYTES_PIXEL = 3
RECORD_BYTES = 13 * 8 + 4 * 4  # one ray record of a generation buffer

DEPTH = int(os.environ.get("RT_BENCH_DEPTH", "4"))  # frames in flight per GPU = host frames in the ring


def ring_depth(parts, steps=200):
    """Frames in flight per GPU.  A part of the frame is a short launch that lasts as long as its slowest wave whatever
    else the GPU could do, and a copy costs about 8 us whatever it moves: the loop therefore renders several frames of
    a part with one launch and moves them with one copy, a third of the ring at most and four at most
    (FrameLoop::run), and ramps the launch size up and down at the ends of a stretch.  A ring of 12 or 16 (four frames
    per launch, three or four launches in flight) is the fastest in the long run, a ring of 8 (two per launch, four in
    flight) over a short timed region, where filling and draining the pipeline counts (tools/part_period.py and the
    sweeps in profiles/: one GPU rendering its part of N, ms per frame on the host, 200 / 20 timed frames: N = 1
    0.117 / 0.149 with 12 and 0.120 / 0.136 with 8, against 0.129 / 0.138 one frame per launch in a ring of 4; N = 8
    0.020 / 0.029 and 0.024 / 0.031 against 0.034 / 0.040; 16 against 12 in the long run: N = 8 0.019 against 0.022)."""
    if "RT_BENCH_DEPTH" in os.environ:
        return DEPTH
    return 16 if steps >= 64 else 8
# Untimed frames before the warmup, as a stretch of their own that ends with every GPU idle.  A fresh process pays
# about 20 ms once: code objects and runtime pools with the first launches, and about 5.6 ms for each of three DMA
# engines on that engine's first copy -- the runtime turns to a second and third engine only when a copy finds the
# others busy, which used to happen somewhere in the first hundred frames (one run in six had it inside its 20 timed
# frames: 0.42 instead of 0.136 ms per frame).  The library now starts the engines before a device's first download
# (prime_copy_engines in csrc/rt_api.cpp; tools/probes/stall_probe.py: no slow stretch after the first in 8 fresh
# processes), so a short stretch is enough: clocks, caches, the first launches.
PRIMING = 32


def host_cores():
    """Cores this process may use: its affinity mask, cut down to a cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
        source = "sched_getaffinity"
    except AttributeError:
        n, source = os.cpu_count() or 1, "os.cpu_count"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < n:
            n, source = int(quota) // int(period), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    env = os.environ.get("OMP_NUM_THREADS")  # the GPU box states its CPU share per GPU here
    if env and env.isdigit() and 0 < int(env) < n:
        n, source = int(env), "OMP_NUM_THREADS (the box's CPU share)"
    return n, source


def sum_stats(stats):
    keys = ("primary", "shadow", "reflect", "refract", "kernel_ms", "kernel_launches", "node_visits", "tri_tests",
            "quadric_tests", "gate_tests", "shadow_node_visits", "shadow_tri_tests", "shadow_quadric_tests", "shadow_gate_tests")
    return {k: sum(getattr(s, k) for s in stats) for k in keys}


def roofline_of(counted, lone, piped, scene_bytes, frame_period_s, n_parts):
    """counted: summed statistics of one instrumented frame (all parts); lone / piped: summed statistics
    of frames rendered one at a time / in the timed loop."""
    nv = counted["node_visits"] + counted["shadow_node_visits"]
    tt = counted["tri_tests"] + counted["shadow_tri_tests"]
    qt = counted["quadric_tests"] + counted["shadow_quadric_tests"]
    gt = counted["gate_tests"] + counted["shadow_gate_tests"]
    hits = counted["shadow"]
    launches = max(1, counted["kernel_launches"])  # per frame: one per part and generation
    flops_survey = nv * SURVEY["node"] + tt * SURVEY["tri"] + qt * SURVEY["quad"] + hits * SURVEY["hit"]
    flops_f64 = tt * F64["tri"] + qt * F64["quad"] + gt * F64["gate"] + hits * F64["hit"]
    f32_ops = nv * F32_PER_NODE
    secondary = counted["reflect"] + counted["refract"]
    records = (counted["primary"] if secondary else 0) + secondary
    hbm = launches * scene_bytes + counted["primary"] * BYTES_PIXEL + 2 * records * RECORD_BYTES
    # the measurement plan's byte model: every ray, visit, test and hit priced as if its records came from memory
    rays = counted["primary"] + counted["shadow"] + secondary
    plan_bytes = rays * (2 * 80 + 2 * 32) + nv * 64 + tt * 72 + qt * 136 + hits * (24 + 200) + launches * scene_bytes
    avg_ms = piped["kernel_ms"] / max(1, piped["kernel_launches"])
    lone_ms = lone["kernel_ms"] / max(1, lone["kernel_launches"])
    # A launch of the timed loop renders several frames of its part (rt_render_frames): per launch means per that many
    # frames.  The instrumented frame and the lone launches are one frame per launch.
    fpl = (piped["primary"] / max(1, counted["primary"])) * launches / max(1, piped["kernel_launches"])
    per_frame_launch = flops_survey / launches
    per_launch = per_frame_launch * fpl
    # Counter figures collected by separate rocprofv3 --pmc runs of this bench (tools/gpu_round.sh) and kept under
    # profiles/: quoted only while the kernel sources they were collected on are the ones in this tree.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_stamp import kernel_sources_sha256
    tree = kernel_sources_sha256(ROOT)
    counters, stale = {}, []
    for name, keys in (("pmc_f64.json", ("f64_ops_per_launch", "valu_issue_busy", "td_busy", "active_lane_share")),
                       ("traffic.json", ("hbm_bytes_per_launch",))):
        path = os.path.join(ROOT, "profiles", name)
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        if doc.get("kernel_sources_sha256") != tree:
            stale.append(name)
            continue
        for k in keys:
            counters[k] = doc.get("k_generation", {}).get(k)
    pmc, traffic = counters.get("f64_ops_per_launch"), counters.get("hbm_bytes_per_launch")
    if pmc:
        pmc *= fpl      # the counter runs launch one frame at a time (tools/gpu_round.sh)
    if traffic:
        traffic *= fpl  # an upper bound: frames of one launch read the scene once between them
    frac = per_launch / (avg_ms * 1e-3) / FP64_VALU_PEAK
    return {
        "bound": "fp64_valu", "achieved": per_launch / (avg_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK / 1e12,
        "unit": "TFLOP/s", "frac": frac, "traffic": traffic,
        "kernel": "k_generation", "launches_per_frame": launches, "frames_per_launch": fpl, "avg_launch_ms": avg_ms,
        "lone_launch_ms": lone_ms, "flops_per_launch": per_launch, "flops_per_launch_of_one_frame": per_frame_launch,
        "frac_lone_launch": per_frame_launch / (lone_ms * 1e-3) / FP64_VALU_PEAK,
        # all GPUs together over the frame period: what the node delivers while launches overlap
        "frac_per_frame_period": flops_survey / frame_period_s / (FP64_VALU_PEAK * n_parts),
        "f64_flops_per_launch_this_implementation": flops_f64 / launches * fpl,
        "frac_f64_this_implementation": flops_f64 / launches * fpl / (avg_ms * 1e-3) / FP64_VALU_PEAK,
        "f32_box_test_ops_per_launch": f32_ops / launches * fpl,
        "pmc_f64_ops_per_launch": pmc,
        "frac_pmc_f64": None if not pmc else pmc / (avg_ms * 1e-3) / FP64_VALU_PEAK,
        # the ceilings that bind this kernel, from the same counter runs (lone launches under the profiler)
        "valu_issue_busy": counters.get("valu_issue_busy"), "td_busy": counters.get("td_busy"),
        "active_lane_share": counters.get("active_lane_share"),
        "counters_note": ("profiles/%s were collected on other kernel sources than this tree's: not quoted" % " and ".join(stale)) if stale
                         else "traffic and counter figures: profiles/traffic.json, profiles/pmc_f64.json, collected on this tree's kernel sources",
        "hbm_bytes_per_launch": hbm / launches * fpl, "frac_hbm": hbm / launches * fpl / (avg_ms * 1e-3) / HBM_PEAK,
        "plan_bytes_per_launch": plan_bytes / launches * fpl,
        "plan_bytes_over_hbm_peak": plan_bytes / launches * fpl / (avg_ms * 1e-3) / HBM_PEAK,
        "events_per_frame": {"node_visits": nv, "tri_tests": tt, "quadric_tests": qt, "gate_tests": gt, "shaded_hits": hits},
        "coefficients": {"frac": SURVEY, "f64_this_implementation": F64, "f32_per_node_visit": F32_PER_NODE},
        "note": "no dense contraction on this path, so no MFMA: the ceiling is vector FP64 (78.6 TFLOP/s datasheet, an FMA "
                "counted as two; the path is compiled without contraction to match the reference bit for bit, so half of "
                "that is reachable). frac prices the events of one instrumented frame with the plan's coefficients "
                "(node 22, triangle 46, quadric 60, shaded hit 150), times the frames a launch of the timed loop renders "
                "(frames_per_launch), over avg_launch_ms, the mean k_generation duration of the timed loop from HIP events "
                "on the library's streams; consecutive launches overlap on two streams, so avg_launch_ms exceeds the "
                "frame period times frames_per_launch (profiles/ holds the kernel timeline); lone_launch_ms is one "
                "frame by itself. hbm_bytes_per_launch is the traffic the algorithm cannot avoid (scene once, pixels, ray "
                "records); the per-visit record fetches are cache hits.",
    }


def metric_name(args):
    """BASELINE.json's metric; another scene or size (a development run) is named as what it is."""
    if os.path.basename(args.scene) == "teapot.json" and (args.width, args.height) == (1920, 1080):
        return "rays/sec (node) on scenes/teapot.json at 1080p"
    return "rays/sec (node) on %s at %dx%d (not the headline configuration)" % (os.path.relpath(args.scene, ROOT), args.width, args.height)


def bench_single_process(args, timed=None):
    """One process drives all N GPUs (one native worker thread per GPU); every frame is complete in one page-locked
    host frame when the clock stops.  timed: under a launcher, the bracket every rank takes part in (barrier and
    device synchronisation on both sides, MAX over the ranks' times); None: the native loop's own clock."""
    pkg = load_package()
    have = pkg.bindings.device_count()
    if have < 1:
        raise SystemExit("no GPU visible")
    n = args.gpus
    devices = [k % have for k in range(n)]
    W, H = args.width, args.height
    t_load = time.perf_counter()
    scene = pkg.Scene.from_file(args.scene, base_dir=ROOT, device=None)  # host side only; the loop owns the GPU copies
    scene.set_canvas(W, H)
    depth = ring_depth(n, args.steps)
    loop = scene.frame_loop(devices, depth=depth)
    load_s = time.perf_counter() - t_load
    info, linfo = scene.info(), loop.info()

    # The timed region of the metric: frames on the host.  Then the same frames left in HBM, frames
    # rendered one at a time (lone launches), and one instrumented frame.
    def leg(to_host):
        if timed is None:  # the loop's own clock: K frames between two rendezvous of its workers with idle GPUs
            return loop.run(args.warmup, args.steps, to_host)
        if args.warmup > 0:
            loop.run(0, args.warmup, to_host)  # untimed, ends with the GPUs idle
        dt, (_, stats) = timed(lambda: loop.run(0, args.steps, to_host))
        return dt, stats

    loop.run(0, PRIMING, not args.hbm_only)  # untimed
    if args.hbm_only:  # profiling aid: rocprofv3 turns the DMA copies into copy kernels that share the GPU with the render
        host_s, host_stats = leg(False)
        sha, hbm_s = None, None
    else:
        host_s, host_stats = leg(True)
        last_slot = (args.steps - 1) % depth if timed is not None else (args.warmup + args.steps - 1) % depth
        frame = loop.host_frame(last_slot).copy()
        sha = hashlib.sha256(frame.tobytes()).hexdigest()
        hbm_s = None if args.no_hbm_leg else leg(False)[0]
    lone = sum_stats(loop.blocking_frames(10, False))
    counted = sum_stats(loop.blocking_frames(1, True))
    piped = sum_stats(host_stats)
    rays_per_frame = (piped["primary"] + piped["shadow"] + piped["reflect"] + piped["refract"]) / args.steps
    # Beside the metric, for N > 1: whole frames dealt to the GPUs (one single-GPU loop per GPU, each on a thread of
    # its own; the native loops run without the interpreter lock) instead of one frame cut into N parts.
    dealt = None
    if n > 1 and not args.no_dealt_leg and not args.hbm_only:
        import threading
        loops = [scene.frame_loop([d], depth=depth) for d in devices]
        gate = threading.Barrier(n + 1)

        def deal(lp):
            lp.run(0, PRIMING + args.warmup, True)
            gate.wait()
            gate.wait()
            lp.run(0, args.steps, True)

        threads = [threading.Thread(target=deal, args=(lp,)) for lp in loops]
        for t in threads:
            t.start()
        gate.wait()  # every loop warm, every GPU idle

        def run_all():
            gate.wait()
            for t in threads:
                t.join()

        if timed is None:
            t0 = time.perf_counter()
            run_all()
            dealt = time.perf_counter() - t0
        else:
            dealt = timed(run_all)[0]
        for lp in loops:
            lp.close()
    roofline = roofline_of(counted, lone, piped, linfo["scene_bytes"], host_s / args.steps, n)

    # Beside the metric: the same loop for a few seconds, in the ring of a long run.  The timed region above is a few
    # milliseconds; this is what the loop sustains (and long enough for an outside observer of the GPU to see it busy).
    sustained = None
    if args.sustained_seconds > 0 and not (args.no_cpu_baseline or args.hbm_only):
        long_depth = ring_depth(n, 1 << 20)
        lp = loop if depth == long_depth else scene.frame_loop(devices, depth=long_depth)
        lp.run(0, 64, True)
        frames = max(1000, int(args.sustained_seconds / (host_s / args.steps)))
        sec, st = lp.run(0, frames, True)
        done = sum_stats(st)
        sustained = {"value": (done["primary"] + done["shadow"] + done["reflect"] + done["refract"]) / sec, "unit": "rays/s",
                     "frames": frames, "seconds": sec, "ms_per_step": sec / frames * 1e3,
                     "pipeline": {"frames_in_flight_per_gpu": long_depth, "frames_of_a_part_per_launch_and_copy": lp.frames_per_launch},
                     "note": "not the metric's timed region: the same frames to the host, back to back for a few seconds"}
        if lp is not loop:
            lp.close()

    cpu_baseline = None
    if n == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle as orc  # test infrastructure, used here only as the timed CPU baseline

        cores, source = host_cores()
        ref = orc.OracleScene.from_file(args.scene, base_dir=ROOT)
        ref.set_canvas(W, H)
        s = args.cpu_stride
        c0 = time.perf_counter()
        res = ref.render(jobs=cores, xstep=s, ystep=s)
        cpu_s = time.perf_counter() - c0
        cpu_rays = sum(res["counts"])
        c1 = time.perf_counter()
        res1 = ref.render(jobs=1, xstep=4 * s, ystep=4 * s)
        cpu1_s = time.perf_counter() - c1
        # the oracle's frame is there anyway: the last timed frame on the host is checked against it
        checked = None
        if sha is not None:
            import numpy as np
            got = np.asarray(frame).reshape(H, W, 3)[::s, ::s].astype(np.int16)
            want = orc.quantise(res["rgb64"][::s, ::s]).astype(np.int16)
            levels = np.abs(got - want).max(axis=2)
            checked = {"pixels_compared": int(levels.size), "pixels_differing": int((levels > 0).sum()),
                       "pixels_off_by_more_than_one_level": int((levels > 1).sum()),
                       "note": "8-bit frame of the last timed step in host memory against ceil(clamp(255 c)) of the CPU baseline's frame"}
        cpu_baseline = {"value": cpu_rays / cpu_s, "unit": "rays/s", "cores": cores, "kind": "port", "frame_check": checked,
                        "cores_from": source, "cpus_of_the_machine": os.cpu_count(),
                        "single_core_value": sum(res1["counts"]) / cpu1_s,
                        "sample": "C++ restatement of the reference algorithm (group box gate + linear scan of all %d "
                                  "triangles per ray), -j %d = every core this process may use, every %dth pixel of every "
                                  "%dth row of the same %dx%d frame: %d rays in %.2f s; single core: every %dth pixel and row, "
                                  "%d rays in %.2f s" % (
                                      info["triangles"], cores, s, s, W, H, cpu_rays, cpu_s, 4 * s, sum(res1["counts"]), cpu1_s)}

    out = {
        "metric": metric_name(args),
        "value": rays_per_frame * args.steps / host_s,
        "unit": "rays/s",
        "n_gpus": n,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": host_s / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s at %dx%d: generated models/teapot.obj (%d smooth triangles in %d groups) + floor plane, "
                        "one point light, recursion depth 5" % (os.path.relpath(args.scene, ROOT), W, H, info["triangles"], info["gates"]),
            "rays_per_frame": rays_per_frame,
            "timed_region": "frames left in HBM (--hbm-only: a profiling run, not the metric's region)" if args.hbm_only
                            else "ray generation to the finished frame in page-locked host memory",
            "parallelism": "one process, %d part(s) = bands of %d pixel rows dealt round-robin, one worker thread, one scene copy "
                           "and one PCIe link per GPU, all parts copied into one shared host frame; no collective" % (n, loop.band_rows) + (
                               "" if timed is None else "; started by a launcher with one rank per GPU: rank 0 drives the GPUs, the "
                               "other ranks hold their GPU and take part in the barriers and the MAX over the ranks' times"),
            "frame_assembled_in_timed_region": True,
            "devices": devices,
            "rehearsal": have < n,
        },
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "frames_left_in_hbm": None if hbm_s is None else {
            "value": rays_per_frame * args.steps / hbm_s, "unit": "rays/s", "ms_per_step": hbm_s / args.steps * 1e3,
            "note": "the same loop without the copies to the host (not the metric's timed region)"},
        "sustained": sustained,
        "whole_frames_dealt_to_the_gpus": None if dealt is None else {
            "value": n * rays_per_frame * args.steps / dealt, "unit": "rays/s", "frames_per_step": n,
            "ms_per_step": dealt / args.steps * 1e3, "scaling": "weak",
            "note": "not the metric: every GPU renders whole frames of its own and copies them to a page-locked ring of its "
                    "own (frames of an animation dealt round-robin); the metric above cuts each frame into parts"},
        "PCIe_GBps": W * H * 3 * args.steps / host_s / 1e9,
        "breakdown_ms": {"frame": host_s / args.steps * 1e3, "generation_kernels_per_frame": piped["kernel_ms"] / args.steps,
                         "lone_frame_kernel": lone["kernel_ms"] / 10},
        "pipeline": {"frames_in_flight_per_gpu": depth, "frames_of_a_part_per_launch_and_copy": loop.frames_per_launch,
                     "untimed_priming_frames": PRIMING},
        "frame_sha256": sha,
        "setup": {"scene_load_and_bvh_s": load_s, "bvh_build_s": linfo["build_seconds"], "scene_bytes": linfo["scene_bytes"],
                  "bvh_builds": 1},
    }
    if have < n:
        out["config"]["note"] = "only %d GPU(s) visible: the %d parts share them, so this line rehearses the path and is no scaling figure" % (have, n)
    print(json.dumps(out))
    loop.close()


def bench_launched(args, world):
    """Started by torch.distributed.run with one rank per GPU.  The frame has to lie whole in one host buffer when
    the clock stops (the reference fills one canvas, src/parallel.rs:186, 194-201).  Page-locked memory cannot be
    shared between processes on the machines this was built on (hipHostRegister refuses shared mappings), so rank 0
    runs the one-process loop over all N GPUs -- one worker thread, one scene copy and one PCIe link per GPU, every
    GPU copying its bands by DMA into the same page-locked frame -- while every other rank holds its GPU and takes
    part in what the launcher form of the bench asks of all ranks: a barrier and a device synchronisation on both
    sides of the K timed frames and the MAX over the ranks' times.  No collective on the data path.
    --per-rank-loops selects the arrangement in which every rank renders and copies its own part (bench_ranks)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")  # gloo: a rehearsal on a machine with fewer GPUs than ranks
    have = max(1, torch.cuda.device_count())
    dev_index = local_rank % have if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
        red_device = device
    else:
        dist.init_process_group(backend)
        red_device = torch.device("cpu")
    torch.zeros(1, device=device)  # every rank initialises its GPU

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed(work):
        sync()
        t0 = time.perf_counter()
        result = work() if work is not None else None
        sync()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), result

    if rank == 0:
        bench_single_process(args, timed)
    else:  # the same brackets, in the same order, with nothing between them
        legs = 1 + (0 if (args.no_hbm_leg or args.hbm_only) else 1) + (0 if (args.no_dealt_leg or args.hbm_only) else 1)
        for _ in range(legs):
            timed(None)
    dist.barrier()
    dist.destroy_process_group()


def bench_ranks(args, world):
    """One rank per GPU under torch.distributed, no data-path collective: every rank runs the host library's
    FrameLoop for its own part of the frame (bands of 8 pixel rows, part = rank) and copies exactly those
    bands over its own GPU's PCIe link into one ring of frames in shared memory that every rank has mapped
    and page-locked.  torch.distributed carries only the barriers around the timed region, the MAX over the
    ranks' times and the sums of the ray counts."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank path on a machine with fewer GPUs than ranks
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    have = max(1, torch.cuda.device_count())
    dev_index = local_rank % have if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
        red_device = device
    else:
        dist.init_process_group(backend)
        red_device = torch.device("cpu")

    pkg = load_package()
    W, H = args.width, args.height
    # Every rank keeps its bands in a page-locked ring of its own; the shared mapping serves to put the last frame
    # together for the checksum, after the timed region.  With RT_BENCH_SHARED_RING=1 the ranks try to page-lock the
    # shared mapping itself (rt_host_register) and, when every rank succeeds, copy their bands straight into it.
    name = ["rt_amd_bench_%d_%d" % (os.getpid(), int(time.time()))] if rank == 0 else [None]
    dist.broadcast_object_list(name, src=0)
    shared = pkg.bindings.SharedFrameRing(name[0], W, H, DEPTH, create=True, register=False) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        shared = pkg.bindings.SharedFrameRing(name[0], W, H, DEPTH, create=False, register=False)
    pinned = [0]
    if os.environ.get("RT_BENCH_SHARED_RING", "0") == "1":  # opt-in: no machine of this build's pool accepts the registration
        pinned[0] = 1 if shared.register() else 0
    flags = [None] * world
    dist.all_gather_object(flags, pinned[0])
    assembled_by_dma = all(flags)
    if assembled_by_dma:
        ring = shared
    else:
        shared.unregister()
        ring = pkg.bindings.HostFrameRing(W, H, DEPTH)
    t_load = time.perf_counter()
    scene = pkg.Scene.from_file(args.scene, base_dir=ROOT, device=None)
    scene.set_canvas(W, H)
    loop = scene.frame_loop([dev_index], depth=DEPTH, first_part=rank, total_parts=world, ring=ring)
    load_s = time.perf_counter() - t_load
    info, linfo = scene.info(), loop.info()

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def all_sum(d):
        keys = sorted(d)
        t = torch.tensor([float(d[k]) for k in keys], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return {k: float(v) for k, v in zip(keys, t.tolist())}

    def timed_loop(to_host):
        loop.run(0, PRIMING + args.warmup, to_host)  # untimed
        sync()
        t0 = time.perf_counter()
        _, stats = loop.run(0, args.steps, to_host)  # returns when this rank's frames are rendered and copied
        sync()
        dt = time.perf_counter() - t0
        t_max = torch.tensor([dt], dtype=torch.float64, device=red_device)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item()), stats

    host_s, host_stats = timed_loop(True)
    last = (args.steps - 1) % DEPTH
    if not assembled_by_dma:  # outside the timed region: the ranks' bands of the last frame put together for the checksum
        for y0, y1 in pkg.tiles.part_bands(H, rank, world, loop.band_rows):
            shared.frame(last)[y0:y1] = ring.frame(last)[y0:y1]
        sync()
    sha = hashlib.sha256(shared.frame(last).tobytes()).hexdigest() if rank == 0 else None
    sync()  # nobody renders over the frame while rank 0 reads it
    hbm_s = None if args.no_hbm_leg else timed_loop(False)[0]
    lone_mine = sum_stats(loop.blocking_frames(10, False))
    counted_mine = sum_stats(loop.blocking_frames(1, True))
    piped_mine = sum_stats(host_stats)
    piped = all_sum(piped_mine)
    # Beside the metric: whole frames dealt to the GPUs instead of one frame cut into parts (a step = one frame per
    # GPU, each rank renders and copies whole frames on its own): what the node delivers when frames are independent.
    dealt = None
    if not args.no_dealt_leg:
        own = pkg.bindings.HostFrameRing(W, H, DEPTH)
        whole = scene.frame_loop([dev_index], depth=DEPTH, first_part=0, total_parts=1, ring=own)
        whole.run(0, PRIMING + args.warmup, True)
        sync()
        t0 = time.perf_counter()
        whole.run(0, args.steps, True)
        sync()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dealt = float(dt.item())
        whole.close()
        own.close()
    rays_per_frame = (piped["primary"] + piped["shadow"] + piped["reflect"] + piped["refract"]) / args.steps
    roofline = roofline_of(counted_mine, lone_mine, piped_mine, linfo["scene_bytes"], host_s / args.steps, 1)
    roofline["scope"] = "rank 0's part of the frame"
    counted = all_sum(counted_mine)
    flops_frame = ((counted["node_visits"] + counted["shadow_node_visits"]) * SURVEY["node"]
                   + (counted["tri_tests"] + counted["shadow_tri_tests"]) * SURVEY["tri"]
                   + (counted["quadric_tests"] + counted["shadow_quadric_tests"]) * SURVEY["quad"] + counted["shadow"] * SURVEY["hit"])
    roofline["frac_per_frame_period"] = flops_frame / (host_s / args.steps) / (FP64_VALU_PEAK * world)
    if rank == 0:
        print(json.dumps({
            "metric": metric_name(args) + (
                "" if assembled_by_dma else ", every rank's part in host memory of its own (frame not assembled)"),
            "value": rays_per_frame * args.steps / host_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": host_s / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%s at %dx%d: generated models/teapot.obj (%d smooth triangles in %d groups) + floor plane, "
                            "one point light, recursion depth 5" % (os.path.relpath(args.scene, ROOT), W, H, info["triangles"], info["gates"]),
                "rays_per_frame": rays_per_frame,
                "timed_region": "ray generation to the finished frame in page-locked host memory: " + (
                    "one shared-memory frame that every rank fills with its own bands by DMA" if assembled_by_dma else
                    "every rank's bands in a page-locked ring of its own (the ranks do not assemble them in one mapping inside "
                    "the timed region: that needs hipHostRegister on a shared mapping, RT_BENCH_SHARED_RING=1, which the "
                    "machines this was built on refuse)"),
                "frame_assembled_by_dma_in_shared_memory": assembled_by_dma,
                "frame_assembled_in_timed_region": bool(assembled_by_dma),
                "parallelism": "one rank per GPU, %d parts = bands of %d pixel rows dealt round-robin, scene replicated, each "
                               "rank copies its bands over its own PCIe link into the shared frame; no collective on the "
                               "data path" % (world, loop.band_rows),
                "rehearsal": backend != "nccl"},
            "roofline": roofline, "cpu_baseline": None,
            "frames_left_in_hbm": None if hbm_s is None else {
                "value": rays_per_frame * args.steps / hbm_s, "unit": "rays/s", "ms_per_step": hbm_s / args.steps * 1e3,
                "note": "the same loop without the copies to the host (not the metric's timed region)"},
            "whole_frames_dealt_to_the_gpus": None if dealt is None else {
                "value": world * rays_per_frame * args.steps / dealt, "unit": "rays/s", "frames_per_step": world,
                "ms_per_step": dealt / args.steps * 1e3, "scaling": "weak",
                "note": "not the metric: every GPU renders whole frames of its own and copies them to its own page-locked "
                        "ring (frames of an animation dealt round-robin); the metric above cuts each frame into parts"},
            "PCIe_GBps": W * H * 3 * args.steps / host_s / 1e9,
            "pipeline": {"frames_in_flight_per_gpu": DEPTH, "untimed_priming_frames": PRIMING},
            "frame_sha256": sha,
            "setup": {"scene_load_and_bvh_s": load_s, "bvh_build_s": linfo["build_seconds"], "scene_bytes": linfo["scene_bytes"]},
        }))
    loop.close()
    sync()
    if ring is not shared:
        ring.close()
    shared.close()
    dist.destroy_process_group()


def bench_ranks_rccl_gather(args, world):
    """The variant with a collective (--rccl-gather): every rank renders its interleaved 32x32 tiles in the
    compact layout, rank 0 gathers them (one RCCL gather per BUCKET frames) into whole frames in its HBM and
    copies every frame to pinned host memory, so every frame crosses rank 0's PCIe link."""
    import torch
    import torch.distributed as dist

    BUCKET = 4
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank path on a machine with fewer GPUs than ranks
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)

    pkg = load_package()
    t_load = time.perf_counter()
    scene = pkg.Scene.from_file(args.scene, base_dir=ROOT, device=dev_index)
    scene.set_canvas(args.width, args.height)
    load_s = time.perf_counter() - t_load
    info, foot = scene.info(), scene.footprint()
    W, H = args.width, args.height
    slots = 2 * BUCKET  # one bucket renders while the other is gathered
    gather = pkg.tiles.FrameGather(W, H, device, rank, world, depth=slots, bucket=BUCKET)
    host_frames = [torch.empty((H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)] if rank == 0 else None
    copy_stream = torch.cuda.ExternalStream(scene.copy_stream_ptr(), device=device)
    buffer_free = [torch.cuda.Event() for _ in range(slots)]
    rendered = [torch.cuda.Event() for _ in range(slots)]
    stream_objects = {}

    def render_stream():  # the library stream the last frame went to
        p = scene.stream_ptr()
        if p not in stream_objects:
            stream_objects[p] = torch.cuda.ExternalStream(p, device=device)
        return stream_objects[p]

    step_no = [0]

    def finish_bucket(b, download):
        first = b * BUCKET
        cur = torch.cuda.current_stream()
        for k in range(first, first + BUCKET):
            cur.wait_event(rendered[k])  # frames alternate between the library's streams
        gather.gather_bucket(b)  # rank 0: the bucket's whole frames, in HBM
        if rank == 0 and download:
            for k in range(first, first + BUCKET):
                frame = gather.frame(k)
                scene.download(frame.data_ptr(), host_frames[k].data_ptr(), frame.numel(), after_stream=cur.cuda_stream)
            buffer_free[first].record(copy_stream)
        else:
            buffer_free[first].record()

    def step(download):
        step_no[0] += 1
        k = (step_no[0] - 1) % slots
        if k % BUCKET == 0:
            buffer_free[k].synchronize()  # the bucket's last gather (and host copies) are done
        scene.render_device(rgb8_ptr=gather.mine[k].data_ptr(), part=rank, n_parts=world, wait=False, compact=True)
        rendered[k].record(render_stream())
        if k % BUCKET == BUCKET - 1:
            finish_bucket(k // BUCKET, download)

    def flush(download):
        if step_no[0] % BUCKET:
            finish_bucket(((step_no[0] - 1) % slots) // BUCKET, download)
            step_no[0] += BUCKET - step_no[0] % BUCKET

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(download, priming):
        for _ in range(priming):
            step(download)
        flush(download)
        scene.wait()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(download)
        flush(download)
        stats = scene.wait()
        sync()
        dt = time.perf_counter() - t0
        t_max = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item()), stats

    host_elapsed, st = timed_loop(True, PRIMING + args.warmup)
    hbm_elapsed, _ = timed_loop(False, PRIMING + args.warmup)
    rays_step = torch.tensor([float(st.rays) / args.steps], dtype=torch.float64, device=device)
    dist.all_reduce(rays_step, op=dist.ReduceOp.SUM)
    rays_per_frame = float(rays_step.item())
    scratch = torch.zeros((H, W, 3), dtype=torch.uint8, device=device)
    counted = sum_stats([scene.render_device(rgb8_ptr=scratch.data_ptr(), part=rank, n_parts=world, counters=True)])
    lone = sum_stats([scene.render_device(rgb8_ptr=scratch.data_ptr(), part=rank, n_parts=world) for _ in range(10)])
    roofline = roofline_of(counted, lone, sum_stats([st]), foot["scene_bytes"], host_elapsed / args.steps, 1)
    roofline["scope"] = "rank 0's part of the frame"
    if rank == 0:
        print(json.dumps({
            "metric": metric_name(args),
            "value": rays_per_frame * args.steps / host_elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": host_elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%s at %dx%d: generated models/teapot.obj (%d smooth triangles in %d groups) + floor plane, "
                            "one point light, recursion depth 5" % (os.path.relpath(args.scene, ROOT), W, H, info["triangles"], info["gates"]),
                "rays_per_frame": rays_per_frame,
                "timed_region": "ray generation to the finished frame in page-locked host memory of rank 0",
                "parallelism": "one rank per GPU, interleaved 32x32 tiles, scene replicated, one RCCL gather per %d frames on "
                               "rank 0, then rank 0's DMA to the host" % BUCKET},
            "roofline": roofline, "cpu_baseline": None,
            "frames_left_in_hbm": {"value": rays_per_frame * args.steps / hbm_elapsed, "unit": "rays/s",
                                   "ms_per_step": hbm_elapsed / args.steps * 1e3},
            "setup": {"scene_load_and_bvh_s": load_s, "bvh_build_s": foot["build_seconds"], "scene_bytes": foot["scene_bytes"]},
        }))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "teapot.json"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the extra leg that leaves the frames in HBM")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="length of the extra leg that runs the loop back to back (0: none; none without the CPU baseline)")
    ap.add_argument("--hbm-only", action="store_true",
                    help="time only the loop that leaves the frames in HBM (for profiler runs: under rocprofv3 the runtime "
                         "turns the DMA copies to the host into copy kernels that share the GPU with the render kernel)")
    ap.add_argument("--no-dealt-leg", action="store_true",
                    help="N > 1: skip the extra leg that deals whole frames to the GPUs")
    ap.add_argument("--per-rank-loops", action="store_true",
                    help="under a launcher with one rank per GPU: every rank renders and copies its own part into a page-locked "
                         "ring of its own; the frame is then not put together inside the timed region (the line says so)")
    ap.add_argument("--rccl-gather", action="store_true",
                    help="under a launcher with one rank per GPU: gather the parts on rank 0 with RCCL instead of "
                         "letting every rank copy its bands into the shared host frame")
    ap.add_argument("--cpu-stride", type=int, default=1,
                    help="CPU baseline renders every n-th pixel of every n-th row (1: the whole frame, about 10 s on 16 cores; "
                         "the single-core figure uses every 4n-th)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        if world != args.gpus:
            raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
        if args.rccl_gather:
            bench_ranks_rccl_gather(args, world)
        elif args.per_rank_loops:
            bench_ranks(args, world)
        else:
            bench_launched(args, world)
    else:
        bench_single_process(args)


if __name__ == "__main__":
    main()
