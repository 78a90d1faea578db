"""The N > 1 path on the CPU: two processes, gloo backend.  Each rank traces its tiles (through the CPU harness of
the kernel source, since this container has no GPU), the tile buffers are gathered on rank 0 with
torch.distributed and scattered into the frame, which must equal the oracle's single-process frame."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import helpers
from simple_raytracer_amd import distributed as D

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import helpers, simple_raytracer_amd as srt
    from simple_raytracer_amd.distributed import TiledFrame
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h = 150, 70
    spec = helpers.default_spec()
    world_s, lights, cam = spec.host(w, h)
    r = srt.Renderer(library={emu!r}, strict=False)
    r.set_scene(world_s, lights, cam, world_kind=srt.WORLD_BVHNODE)
    tf = TiledFrame(r, w, h, tile=16, device="cpu", samples=2, max_depth=4, seed=11)
    frame = tf.step()
    rays = torch.tensor([float(r.stats()["rays_path"])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        from oracle import orc
        ref, st = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=4, seed=11)
        img = frame.numpy().view(np.uint32).reshape(h, w)
        assert (img == ref).all(), "assembled frame differs from the oracle"
        assert int(rays.item()) == st["rays_path"], (rays.item(), st["rays_path"])
        print("DIST-OK", world)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_tile_gather_matches_oracle(emu_lib, oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=helpers.ROOT, emu=emu_lib))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DIST-OK 2" in out.stdout


def test_tile_ownership_covers_the_frame_once():
    w, h, tw, th = 100, 52, 16, 8
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            seen += D.local_tile_ids(w, h, tw, th, rank, world)
        tx, ty = D.tile_grid(w, h, tw, th)
        assert sorted(seen) == list(range(tx * ty))
        # rank 0 owns the most tiles: every rank's buffer fits rank 0's stride
        assert all(len(D.local_tile_ids(w, h, tw, th, r, world)) <= len(D.local_tile_ids(w, h, tw, th, 0, world)) for r in range(world))


def test_assemble_tiles_host_inverts_the_layout():
    w, h, tw, th, world = 70, 45, 16, 8, 3
    tx, ty = D.tile_grid(w, h, tw, th)
    frame = np.arange(w * h, dtype=np.uint32).reshape(h, w) + 1
    stride = len(D.local_tile_ids(w, h, tw, th, 0, world)) * tw * th
    gathered = np.zeros((world, stride), dtype=np.uint32)
    for r in range(world):
        for j, t in enumerate(D.local_tile_ids(w, h, tw, th, r, world)):
            row, col = D.tile_position(t, tx)
            y0, x0 = row * th, col * tw
            tile = np.zeros((th, tw), dtype=np.uint32)
            hh, ww = min(th, h - y0), min(tw, w - x0)
            tile[:hh, :ww] = frame[y0:y0 + hh, x0:x0 + ww]
            gathered[r, j * tw * th:(j + 1) * tw * th] = tile.reshape(-1)
    assert (D.assemble_tiles_host(gathered, w, h, tw, th, world) == frame).all()


def test_tile_numbering_puts_ranks_on_diagonals():
    """tiles_x a multiple of the rank count (3840 pixels = 120 tiles of 32 over 2, 4 or 8 ranks) must not hand every rank
    the same columns in every row: with the numbering of srt_render_params a rank's tiles move one column per row."""
    tiles_x, tiles_y = 120, 68
    for world in (2, 4, 8):
        owner = np.empty((tiles_y, tiles_x), dtype=np.int64)
        for t in range(tiles_x * tiles_y):
            row, col = D.tile_position(t, tiles_x)
            assert D.tile_number(col, row, tiles_x) == t
            owner[row, col] = t % world
        for r in range(world):
            cols_per_row = [set(np.nonzero(owner[row] == r)[0] % world) for row in range(world)]
            assert len(set(map(frozenset, cols_per_row))) == world      # a different set of columns in each of `world` rows
            share = (owner == r).mean()
            assert abs(share - 1.0 / world) < 1e-3
        # every block of world x world tiles holds every rank in every row and every column
        block = owner[:world, :world]
        assert all(len(set(block[i])) == world and len(set(block[:, i])) == world for i in range(world))
