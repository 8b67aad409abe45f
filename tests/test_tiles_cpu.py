"""The multi-GPU partition and framebuffer gather, exercised on CPU with two gloo ranks."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_parts_partition_the_image(pkg):
    for (w, h, n) in [(1920, 1080, 8), (100, 70, 3), (31, 9, 2), (64, 64, 1)]:
        seen = torch.zeros(w * h, dtype=torch.int32)
        for p in range(n):
            idx = pkg.tiles.part_pixel_index(w, h, p, n)
            seen[idx] += 1
            ys, xs = idx // w, idx % w
            assert bool((((xs // 32) + (ys // 32)) % n == p).all())
        assert bool((seen == 1).all())
        # the compact layout lists the same pixels, each once, plus the out-of-image slots of edge tiles
        for p in range(n):
            c = pkg.tiles.compact_pixel_index(w, h, p, n)
            assert c.numel() % 1024 == 0
            assert torch.equal(torch.sort(c[c >= 0]).values, pkg.tiles.part_pixel_index(w, h, p, n))
    sizes = [pkg.tiles.part_pixel_index(1920, 1080, p, 8).numel() for p in range(8)]
    assert max(sizes) - min(sizes) <= 0.05 * max(sizes)  # balanced to a few percent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        truth = torch.stack([xs % 251, ys % 241, (xs * 7 + ys * 13) % 239], dim=-1).to(torch.uint8)
        gather = pkg.tiles.FrameGather(w, h, "cpu", rank, world)
        ok = True
        for k in (0, 1, 0):  # both buffer sets, and reuse
            # what this rank's GPU would have rendered in the compact layout
            idx = pkg.tiles.compact_pixel_index(w, h, rank, world, outside=-1)
            inside = idx >= 0
            gather.mine[k].zero_()
            gather.mine[k][: idx.numel()][inside] = truth.view(-1, 3)[idx[inside]]
            frame = gather(k)
            if rank == 0:
                ok = ok and bool(torch.equal(frame, truth))
        # buckets: two buckets of three frames, each frame different, one gather per bucket
        bucketed = pkg.tiles.FrameGather(w, h, "cpu", rank, world, depth=6, bucket=3)
        idx = pkg.tiles.compact_pixel_index(w, h, rank, world, outside=-1)
        inside = idx >= 0
        for b in (0, 1, 0):
            for f in range(3):
                k = 3 * b + f
                bucketed.mine[k].fill_(255)
                bucketed.mine[k][: idx.numel()][inside] = (truth + (k + b)).view(-1, 3)[idx[inside]]
            bucketed.gather_bucket(b)
            if rank == 0:
                for f in range(3):
                    ok = ok and bool(torch.equal(bucketed.frame(3 * b + f), truth + (3 * b + f + b)))
        if rank == 0:
            out.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_rebuilds_the_frame():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 150, 100, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def _ring_worker(rank, world, port, w, h, out):
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        depth = 3
        name = ["rt_amd_test_%d" % os.getpid()] if rank == 0 else [None]
        dist.broadcast_object_list(name, src=0)
        ring = pkg.bindings.SharedFrameRing(name[0], w, h, depth, create=True, register=False) if rank == 0 else None
        dist.barrier()
        ok = True
        if rank != 0:
            try:
                pkg.bindings.SharedFrameRing(name[0], w, h + 1, depth, create=False, register=False)
                ok = False  # a ring of another size must be refused
            except ValueError:
                pass
            ring = pkg.bindings.SharedFrameRing(name[0], w, h, depth, create=False, register=False)
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        truth = np.stack([xs % 251, ys % 241, (xs * 7 + ys * 13) % 239], axis=-1).astype(np.uint8)
        for slot in range(depth):  # what this rank's GPU copies: its own bands of 32 rows, nothing else
            for y0, y1 in pkg.tiles.part_bands(h, rank, world):
                ring.frame(slot)[y0:y1] = truth[y0:y1] + slot
        dist.barrier()
        if rank == 0:
            for slot in range(depth):
                ok = ok and bool(np.array_equal(ring.frame(slot), truth + slot))
        flag = [ok]
        gathered = [None] * world
        dist.all_gather_object(gathered, flag)
        dist.barrier()
        path = ring.path
        ring.close()
        dist.barrier()
        if rank == 0:
            out.put(all(g[0] for g in gathered) and not os.path.exists(path))
    finally:
        dist.destroy_process_group()


def test_two_ranks_assemble_frames_in_a_shared_ring():
    """The launcher path of bench.py without its GPU work: two processes map one shared-memory ring, each
    writes only its own bands, and rank 0 reads whole frames; the creator removes the file."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, 70, 100, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_bands_partition_the_rows(pkg):
    for h, n in [(1080, 8), (1080, 1), (31, 2), (64, 3), (100, 5)]:
        seen = np.zeros(h, int)
        for p in range(n):
            for y0, y1 in pkg.tiles.part_bands(h, p, n):
                assert y0 % 32 == 0 and (y1 - y0 == 32 or y1 == h)
                assert (y0 // 32) % n == p
                seen[y0:y1] += 1
        assert (seen == 1).all()
