"""The many-frame form of parallel_render (src/parallel.rs:183-204): one worker and one scene copy per
part, every part copying its own bands of pixel rows (8 rows high in the loop; 16 or 32 on request) into one shared
host frame.  On the one GPU of
the test box all parts run on device 0; the frame must equal the frame one part renders, byte for byte."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,size", [("teapot", (200, 150)), ("glass", (161, 97)), ("example", (64, 32))])
def test_parts_fill_one_host_frame(pkg, root, name, size):
    path = os.path.join(root, "scenes", name + ".json")
    whole = pkg.Scene.from_file(path, base_dir=root, device=0)
    whole.set_canvas(*size)
    want = whole.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    host = pkg.Scene.from_file(path, base_dir=root, device=None)  # the loop owns its GPU copies
    host.set_canvas(*size)
    for parts in (1, 2, 3, 7):  # 150 rows = 4 whole bands and one of 22 rows; 7 parts: some own nothing
        loop = host.frame_loop([0] * parts, depth=3)
        seconds, stats = loop.run(2, 5, True)
        assert seconds > 0 and len(stats) == parts
        assert sum(s.primary for s in stats) == 5 * size[0] * size[1]
        for slot in range(3):
            assert np.array_equal(loop.host_frame(slot), want), "parts=%d slot=%d" % (parts, slot)
        counted = loop.blocking_frames(1, counters=True)
        assert sum(s.primary for s in counted) == size[0] * size[1]
        assert sum(s.node_visits + s.quadric_tests for s in counted) > 0
        loop.close()


def test_loops_that_render_some_parts_share_a_ring(pkg, root):
    """One loop per part, as one process per GPU runs them: part k of 3 into a ring the caller owns; after all
    three have run, every ring slot holds the whole frame."""
    path = os.path.join(root, "scenes", "teapot.json")
    whole = pkg.Scene.from_file(path, base_dir=root, device=0)
    whole.set_canvas(200, 150)
    want = whole.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    host = pkg.Scene.from_file(path, base_dir=root, device=None)
    host.set_canvas(200, 150)
    ring = pkg.bindings.HostFrameRing(200, 150, 3)
    rays = 0
    for part in range(3):
        loop = host.frame_loop([0], depth=3, first_part=part, total_parts=3, ring=ring)
        seconds, stats = loop.run(1, 3, True)
        rays += sum(s.primary for s in stats)
        if part < 2:
            assert not np.array_equal(ring.frame(0), want)  # bands of the parts still to come are missing
        loop.close()
    assert rays == 3 * 200 * 150
    for slot in range(3):
        assert np.array_equal(ring.frame(slot), want)
    with pytest.raises(Exception):
        host.frame_loop([0, 0], depth=3, first_part=2, total_parts=3, ring=ring)  # parts 2 and 3 of 3
    ring.close()


@pytest.mark.parametrize("rows", [16, 8])
def test_finer_bands_assemble_the_same_frame(pkg, root, rows):
    """Bands of 16 and 8 rows instead of 32 (RT_FLAG_BANDS_16 / _8; RT_AMD_BAND_ROWS for the frame loop): the parts of
    a frame rendered with finer bands put together the frame one part renders, on a canvas whose height is no
    multiple of the band height."""
    import torch
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "glass.json"), base_dir=root, device=0)
    w, h = 200, 141
    scene.set_canvas(w, h)
    want, _ = scene.render()
    for n in (2, 3, 5):
        frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        for part in range(n):
            scene.render_device(rgb8_ptr=frame.data_ptr(), part=part, n_parts=n, bands=True, band_rows=rows)
            mine = torch.zeros(h, dtype=torch.bool)
            for y0, y1 in pkg.tiles.part_bands(h, part, n, rows):
                mine[y0:y1] = True
            got = frame.cpu().numpy()
            # rows of other parts rendered so far or still empty; this part's rows are final
            assert np.array_equal(got[mine.numpy()], want["rgb8"][mine.numpy()])
        assert np.array_equal(frame.cpu().numpy(), want["rgb8"])
    os.environ["RT_AMD_BAND_ROWS"] = str(rows)
    try:
        host_scene = pkg.Scene.from_file(os.path.join(root, "scenes", "teapot.json"), base_dir=root, device=None)
        host_scene.set_canvas(320, 180)
        loop = host_scene.frame_loop([0, 0, 0], depth=2)
        assert loop.band_rows == rows
        loop.run(1, 3, True)
        got = loop.host_frame(1).copy()  # frames 0..3 went to slots 0, 1, 0, 1
        loop.close()
    finally:
        os.environ.pop("RT_AMD_BAND_ROWS", None)
    one = pkg.Scene.from_file(os.path.join(root, "scenes", "teapot.json"), base_dir=root, device=0)
    one.set_canvas(320, 180)
    assert np.array_equal(got, one.render()[0]["rgb8"])


@pytest.mark.parametrize("name", ["teapot", "glass"])
def test_eight_frames_in_flight(pkg, root, name):
    """The ring depth bench.py uses when a frame is cut into four parts or more: eight device frames, eight host frames
    and eight copy fences per part.  The glass scene's frames are several generations of rays each: they are enqueued
    unverified on two sets of ray records in turn and copied one frame late."""
    path = os.path.join(root, "scenes", name + ".json")
    one = pkg.Scene.from_file(path, base_dir=root, device=0)
    one.set_canvas(240, 136)
    want = one.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    host = pkg.Scene.from_file(path, base_dir=root, device=None)
    host.set_canvas(240, 136)
    loop = host.frame_loop([0, 0, 0, 0], depth=8)
    seconds, stats = loop.run(3, 21, True)  # 24 frames: every slot written three times
    assert sum(s.primary for s in stats) == 21 * 240 * 136
    for slot in range(8):
        assert np.array_equal(loop.host_frame(slot), want), slot
    seconds, stats = loop.run(0, 5, False)  # left in HBM: collected by the wait at the end of the stretch
    assert sum(s.primary for s in stats) == 5 * 240 * 136
    loop.close()


def test_clone_renders_the_same_frame(pkg, root):
    import ctypes as C
    amd, _ = pkg.bindings.libraries()
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "glass.json"), base_dir=root, device=0)
    scene.set_canvas(120, 80)
    want, _ = scene.render()
    twin = C.c_void_p()
    assert amd.rt_scene_clone(scene._dev(), 0, C.byref(twin)) == 0
    cam = scene.camera()
    a8 = np.zeros((80, 120, 3), np.uint8)
    a64 = np.zeros((80, 120, 3), np.float64)
    leaf = np.full((80, 120), -1, np.int32)
    st = pkg.bindings.RtStats()
    rc = amd.rt_render_host(twin, C.byref(cam), 0, 1, 0, a8.ctypes.data_as(C.c_void_p), a64.ctypes.data_as(C.c_void_p),
                            leaf.ctypes.data_as(C.c_void_p), C.byref(st))
    amd.rt_scene_destroy.argtypes = [C.c_void_p]
    amd.rt_scene_destroy(twin)
    assert rc == 0
    assert np.array_equal(a8, want["rgb8"]) and np.array_equal(leaf, want["hit_leaf"])
    assert np.array_equal(a64, want["rgb64"])


def run_bench(root, *flags):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-hbm-leg", *flags],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_runs_bare_with_two_gpus_requested(root):
    """`python bench.py --gpus 2` as a plain child process (no launcher): rc 0, n_gpus 2, and the gathered
    frame equals the frame of the one-GPU run."""
    one = run_bench(root, "--gpus", "1", "--steps", "2", "--warmup", "1")
    two = run_bench(root, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["devices"] in ([0, 1], [0, 0])
    assert one["frame_sha256"] == two["frame_sha256"]
    assert one["config"]["rays_per_frame"] == two["config"]["rays_per_frame"]
    for line in (one, two):
        assert line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
        assert line["roofline"]["frac"] > 0 and line["roofline"]["launches_per_frame"] == line["n_gpus"]


def test_bench_under_the_launcher_assembles_the_same_frame(root):
    """bench.py as the round-end driver starts it for N > 1 (torch.distributed.run, one rank per GPU), here with
    both ranks on the one GPU and gloo for the barriers: the frame the two ranks assemble in the shared ring
    equals the frame of the one-GPU run, and the ray count is the same."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one = run_bench(root, "--gpus", "1", "--steps", "3", "--warmup", "1")
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-hbm-leg"],
                         capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, "\n".join(l for l in out.stderr.splitlines() if l.startswith("[rank"))[-3000:]
    two = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["value"] > 0
    assert two["frame_sha256"] == one["frame_sha256"]
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    assert "no collective" in two["config"]["parallelism"]
    # the whole frame lies in one host buffer when the clock stops (the reference fills one canvas, src/parallel.rs:186)
    assert two["config"]["frame_assembled_in_timed_region"] is True
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt_amd_bench_")]


def test_bench_with_one_loop_per_rank_says_that_the_frame_is_not_assembled(root):
    """--per-rank-loops: every rank renders and copies its own part into host memory of its own; the ranks' bands are
    put together after the timed region only, for the checksum, and the line must say so."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one = run_bench(root, "--gpus", "1", "--steps", "3", "--warmup", "1")
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-hbm-leg", "--no-dealt-leg", "--per-rank-loops"],
                         capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, "\n".join(l for l in out.stderr.splitlines() if l.startswith("[rank"))[-3000:]
    two = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert two["frame_sha256"] == one["frame_sha256"]
    assert two["config"]["frame_assembled_in_timed_region"] is False and "not assembled" in two["metric"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt_amd_bench_")]


def test_a_scene_chooses_streams_that_share_no_hardware_queue(pkg, root, capfd, monkeypatch):
    """A scene takes its copy stream and its render streams from a pool, keeping those on which two lingering kernels
    run side by side (streams on one hardware queue take turns, and a copy stream that shares one with a render
    stream halves the frame loop's rate).  The choice is reported under RT_AMD_DEBUG; the frames are the same with
    the streams as made (RT_AMD_NO_QUEUE_TEST)."""
    import re
    path = os.path.join(root, "scenes", "teapot.json")
    monkeypatch.setenv("RT_AMD_DEBUG", "1")
    chosen = pkg.Scene.from_file(path, base_dir=root, device=0)
    err = capfd.readouterr().err
    found = re.findall(r"streams: (\d+) of (\d+) on hardware queues of their own", err)
    assert found, err
    distinct, wanted = map(int, found[-1])
    assert wanted == 4 and 2 <= distinct <= 4  # four hardware queues per process unless something else holds them
    chosen.set_canvas(200, 150)
    want = chosen.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    monkeypatch.setenv("RT_AMD_NO_QUEUE_TEST", "1")
    as_made = pkg.Scene.from_file(path, base_dir=root, device=None)
    as_made.set_canvas(200, 150)
    assert "streams:" not in capfd.readouterr().err
    loop = as_made.frame_loop([0, 0], depth=8)
    loop.run(1, 6, True)
    for slot in range(6):
        assert np.array_equal(loop.host_frame(slot), want)
    loop.close()
