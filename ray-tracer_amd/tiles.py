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
TILE_PIXELS = TILE * TILE


def part_pixel_index(width, height, part, n_parts, device="cpu"):
    """Linear pixel indices (y * width + x) of the pixels that belong to `part`, in row-major order."""
    ys, xs = torch.meshgrid(torch.arange(height, device=device), torch.arange(width, device=device), indexing="ij")
    owner = ((xs // TILE) + (ys // TILE)) % n_parts
    return (ys * width + xs)[owner == part].reshape(-1)


def part_tiles(width, height, part, n_parts):
    """Tile numbers (ty * tiles_x + tx) owned by `part`, in the order rt_render walks them."""
    tiles_x, tiles_y = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    ty, tx = torch.meshgrid(torch.arange(tiles_y), torch.arange(tiles_x), indexing="ij")
    mine = ((tx + ty) % n_parts) == part
    return (ty * tiles_x + tx)[mine].reshape(-1), tiles_x


def compact_pixel_index(width, height, part, n_parts, device="cpu", outside=-1):
    """For every slot of the compact rgb8 layout of rt_render (RT_FLAG_COMPACT): the linear pixel index
    y * width + x it holds, or `outside` for slots of a tile that fall outside the image.  Slot order:
    tile by tile, inside a tile sixteen 8x8 blocks (block b at columns 8 (b % 4), rows 8 (b // 4)),
    each block row-major."""
    tiles, tiles_x = part_tiles(width, height, part, n_parts)
    slot = torch.arange(TILE_PIXELS)
    block, lane = slot // 64, slot % 64
    dx = (block % 4) * 8 + lane % 8
    dy = (block // 4) * 8 + lane // 8
    x = (tiles % tiles_x)[:, None] * TILE + dx[None, :]
    y = (tiles // tiles_x)[:, None] * TILE + dy[None, :]
    idx = y * width + x
    idx[(x >= width) | (y >= height)] = outside
    return idx.reshape(-1).to(device)


def part_bands(height, part, n_parts, rows=TILE):
    """Row ranges [(y0, y1), ...] of the bands of `rows` pixel rows (32, 16 or 8) that belong to `part` under
    RT_FLAG_BANDS (band b belongs to part b % n_parts)."""
    n_bands = (height + rows - 1) // rows
    return [(b * rows, min(height, (b + 1) * rows)) for b in range(part, n_bands, n_parts)]


def download_bands(scene, frame_ptr, host_ptr, width, height, part, n_parts, after_stream=0):
    """Enqueues the copies of this part's bands from a full-size device frame to a full-size host frame."""
    band_bytes = TILE * width * 3
    bands = part_bands(height, part, n_parts)
    full = [b for b in bands if b[1] - b[0] == TILE]
    if full:
        off = full[0][0] * width * 3
        scene.download_2d(frame_ptr + off, host_ptr + off, band_bytes, len(full), n_parts * band_bytes, after_stream)
    for y0, y1 in bands:
        if y1 - y0 != TILE:  # the last band of an image whose height is not a multiple of 32
            off = y0 * width * 3
            scene.download(frame_ptr + off, host_ptr + off, (y1 - y0) * width * 3, after_stream)


class FrameGather:
    """Collects the parts of a frame on rank 0.

    Every rank renders its part in the compact layout into `mine[k]` (k < depth: several buffers so that
    frame n + 1 can be rendered while frame n is still being gathered).  The buffers form depth / bucket
    buckets of `bucket` consecutive frames; `gather_bucket(b)` sends a whole bucket to rank 0 in one
    gather and, there, scatters it into the (H, W, 3) uint8 frames of the bucket with one indexed copy:
    when a frame renders faster than a collective starts, a rank moves several frames per collective.
    `__call__(k)` is the per-frame form (bucket = 1).  With one rank nothing is exchanged: the frame is
    rendered in place.
    """

    def __init__(self, width, height, device, rank, world, backend=None, depth=2, bucket=1):
        self.rank, self.world, self.device = rank, world, device
        self.width, self.height = width, height
        self.backend = backend or (dist.get_backend() if world > 1 else None)
        self.depth, self.bucket = depth, bucket
        if depth % bucket:
            raise ValueError("depth %d is not a multiple of bucket %d" % (depth, bucket))
        if world == 1:
            return
        n_pix = width * height
        idx = [compact_pixel_index(width, height, p, world, outside=n_pix) for p in range(world)]
        self.slots = [int(i.numel()) for i in idx]
        n_max = max(self.slots)
        n_buckets = depth // bucket
        self.mine_all = torch.zeros((n_buckets, bucket, n_max, 3), dtype=torch.uint8, device=device)
        self.mine = [self.mine_all[k // bucket, k % bucket] for k in range(depth)]
        if rank == 0:
            # one index for all parts and all frames of a bucket; padded slots and out-of-image slots
            # land on the spare row that follows every frame
            pad = torch.full((world, bucket, n_max), n_pix, dtype=torch.int64)
            for p in range(world):
                pad[p, :, : self.slots[p]] = idx[p][None, :]
            pad += (torch.arange(bucket, dtype=torch.int64) * (n_pix + 1))[None, :, None]
            self.index = pad.reshape(-1).to(device)
            self.recv = [torch.zeros((world, bucket, n_max, 3), dtype=torch.uint8, device=device)
                         for _ in range(n_buckets)]
            self.canvas = [torch.zeros((bucket, n_pix + 1, 3), dtype=torch.uint8, device=device)
                           for _ in range(n_buckets)]

    def frame(self, k):
        """(H, W, 3) view of buffer k on rank 0 after its bucket was gathered."""
        return self.canvas[k // self.bucket][k % self.bucket, :-1].view(self.height, self.width, 3)

    def gather_bucket(self, b):
        """Gathers frames b * bucket ... (b + 1) * bucket - 1 on rank 0 (a collective: every rank calls it)."""
        if self.world == 1:
            return
        mine = self.mine_all[b]
        if self.backend == "gloo" and mine.is_cuda:
            # rehearsal on a machine without RCCL peers: stage through host memory
            mine = mine.cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(mine, parts, dst=0)
            if self.rank == 0:
                self.recv[b].copy_(torch.stack(parts))
        else:
            dist.gather(mine, list(self.recv[b].unbind(0)) if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            self.canvas[b].view(-1, 3).index_copy_(0, self.index, self.recv[b].view(-1, 3))

    def __call__(self, k):
        if self.world == 1:
            return None
        if self.bucket != 1:
            raise ValueError("per-frame gather needs bucket = 1")
        self.gather_bucket(k)
        return self.frame(k) if self.rank == 0 else None
