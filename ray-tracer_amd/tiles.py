"""Image-tile partition over GPUs and the framebuffer gather.

The reference spreads pixels over `jobs` threads through one work queue
(src/parallel.rs:183-204).  Here the image is cut into 32x32 tiles, tile
(tx, ty) belongs to part (tx + ty) % n_parts, every part is rendered by one GPU
(one process per GPU under torch.distributed) and rank 0 gathers the tiles:
the only exchange of the path, a gather with no reduction.
"""
import torch
import torch.distributed as dist

TILE = 32


def part_pixel_index(width, height, part, n_parts, device="cpu"):
    """Linear pixel indices (y * width + x) of the pixels that belong to `part`, in row-major order."""
    ys, xs = torch.meshgrid(torch.arange(height, device=device), torch.arange(width, device=device), indexing="ij")
    owner = ((xs // TILE) + (ys // TILE)) % n_parts
    return (ys * width + xs)[owner == part].reshape(-1)


class FrameGather:
    """Gathers every rank's tiles of an (H, W, C) frame into rank 0's frame, in place."""

    def __init__(self, width, height, channels, dtype, device, rank, world):
        self.rank, self.world = rank, world
        self.idx = [part_pixel_index(width, height, p, world, device) for p in range(world)]
        n_max = max(int(i.numel()) for i in self.idx)
        self.mine = torch.zeros((n_max, channels), dtype=dtype, device=device)
        self.parts = [torch.zeros((n_max, channels), dtype=dtype, device=device) for _ in range(world)] if rank == 0 else None

    def __call__(self, frame):
        if self.world == 1:
            return frame
        flat = frame.view(-1, frame.shape[-1])
        n = self.idx[self.rank].numel()
        self.mine[:n] = flat[self.idx[self.rank]]
        dist.gather(self.mine, self.parts, dst=0)
        if self.rank == 0:
            for p in range(1, self.world):
                flat[self.idx[p]] = self.parts[p][: self.idx[p].numel()]
        return frame
