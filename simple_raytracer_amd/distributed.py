"""One frame across several GPUs: one process per GPU, screen tiles interleaved over the ranks, tile buffers gathered
on rank 0 (torch.distributed: RCCL over xGMI for device tensors) and scattered into the frame.

This is what the reference does with threads and a shared Color[] (Game1.cs:157, 161-175, 376-380): its row bands
become tiles (tile t belongs to rank t % world_size, which balances the cheap sky rows and the expensive sphere
rows), and the shared buffer becomes a gather.  Ranks never exchange anything during tracing: every rank holds the
whole scene, and the RNG is keyed by the global pixel index, so the frame does not depend on the number of ranks.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _native as N


def tile_grid(width, height, tile_w, tile_h):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def tile_position(t, tiles_x):
    """(row, column) of tile number t: rows are numbered from a start that moves one column per row, so that
    "tile t belongs to rank t % world_size" puts the ranks on diagonals of the grid (srt_render_params in srt.h)."""
    row = t // tiles_x
    return row, (t % tiles_x + row) % tiles_x


def tile_number(col, row, tiles_x):
    """Number of the tile at (column, row): the inverse of tile_position."""
    return row * tiles_x + (col - row) % tiles_x


def local_tile_ids(width, height, tile_w, tile_h, rank, world_size):
    """Global ids of the tiles a rank renders, in the order they sit in its buffer."""
    tx, ty = tile_grid(width, height, tile_w, tile_h)
    return list(range(rank, tx * ty, world_size))


def assemble_tiles_host(gathered, width, height, tile_w, tile_h, world_size):
    """Scatter gathered tile buffers (world_size rows of equal length, rank r's buffer in row r) into the row-major
    frame.  Host-side counterpart of srt_assemble_tiles for buffers that live in host memory."""
    gathered = np.asarray(gathered).reshape(world_size, -1)
    tx, _ = tile_grid(width, height, tile_w, tile_h)
    frame = np.zeros((height, width), dtype=gathered.dtype)
    area = tile_w * tile_h
    for r in range(world_size):
        for j, t in enumerate(local_tile_ids(width, height, tile_w, tile_h, r, world_size)):
            row, col = tile_position(t, tx)
            y0, x0 = row * tile_h, col * tile_w
            hh, ww = min(tile_h, height - y0), min(tile_w, width - x0)
            frame[y0:y0 + hh, x0:x0 + ww] = gathered[r, j * area:(j + 1) * area].reshape(tile_h, tile_w)[:hh, :ww]
    return frame


class TiledFrame:
    """Buffers and steps of a tile-partitioned render on this rank.  `device` is a torch device: "cuda" tensors are
    rendered into directly and gathered over RCCL; "cpu" tensors (used by the CPU test-suite with gloo) go through the
    blocking host entry point."""

    def __init__(self, renderer, width, height, rank=None, world_size=None, tile=32, device="cuda", group=None, **render_kw):
        self.r = renderer
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.width, self.height, self.tile = width, height, tile
        self.device = torch.device(device)
        self.render_kw = render_kw
        mk = lambda rk: renderer.params(width, height, layout=N.LAYOUT_TILES, tile_w=tile, tile_h=tile, rank=rk,
                                        world_size=self.world, **render_kw)
        self.params = mk(self.rank)
        self.params0 = mk(0)
        self.local_pixels = renderer.local_pixels(self.params)
        self.stride = renderer.local_pixels(self.params0)   # rank 0 owns the most tiles; every rank pads to it
        self.out = torch.zeros(self.stride, dtype=torch.int32, device=self.device)
        self.gathered = self.frame = None
        if self.rank == 0:
            self.gathered = torch.zeros(self.world * self.stride, dtype=torch.int32, device=self.device)
            self.gather_list = list(self.gathered.view(self.world, self.stride).unbind(0))
            self.frame = torch.zeros(width * height, dtype=torch.int32, device=self.device)

    def set_seed(self, seed):
        self.params.seed = seed
        self.params0.seed = seed

    def render_local(self):
        """Traces this rank's tiles into its buffer (asynchronously on torch's current stream for device buffers)."""
        if self.device.type == "cuda":
            self.r.render_device(self.params, self.out.data_ptr(), self.out.numel(), torch.cuda.current_stream().cuda_stream)
        else:
            buf = self.r.render(self.width, self.height, layout=N.LAYOUT_TILES, tile_w=self.tile, tile_h=self.tile,
                                rank=self.rank, world_size=self.world, **dict(self.render_kw, seed=self.params.seed))
            self.out[:buf.size] = torch.from_numpy(buf.view(np.int32))

    def gather(self):
        """The one exchange step: every rank's tile buffer to rank 0."""
        if self.world == 1:
            if self.rank == 0:
                self.gathered.copy_(self.out)
            return
        if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
            # rehearsal without RCCL (several ranks sharing one GPU): gloo gathers host tensors only
            host = self.out.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0, group=self.group)
            if self.rank == 0:
                self.gathered.view(self.world, self.stride).copy_(torch.stack(parts))
            return
        dist.gather(self.out, self.gather_list if self.rank == 0 else None, dst=0, group=self.group)

    def assemble(self):
        """Rank 0: tiles -> row-major frame (H*W packed pixels).  Returns the frame tensor, None on other ranks."""
        if self.rank != 0:
            return None
        if self.device.type == "cuda":
            self.r.assemble_tiles(self.params0, self.gathered.data_ptr(), self.stride, self.frame.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        else:
            f = assemble_tiles_host(self.gathered.numpy().view(np.uint32), self.width, self.height, self.tile, self.tile, self.world)
            self.frame.copy_(torch.from_numpy(f.reshape(-1).view(np.int32)))
        return self.frame

    def step(self):
        self.render_local()
        self.gather()
        return self.assemble()
