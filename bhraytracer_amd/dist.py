"""Tile partition + framebuffer gather for the multi-GPU path.

The image is cut into square tiles; tile t (row-major) belongs to rank t mod N — interleaved,
because cost is spatially uneven (glass / mesh pixels cost far more than background).  Every rank
renders only its tiles (bhrt_opts.rank/world_size/tile_size), then ONE collective moves each
rank's packed tiles to everybody: an all_gather over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in
the CPU tests).  No other collective exists on this path: samples are keyed by global (pixel,
sample), so the image does not depend on N.

torch is plumbing here (device buffers + torch.distributed); the render itself is libbhrt.so.
"""
import torch


def tile_grid(width: int, height: int, tile: int):
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    return tx, ty, tx * ty


def owned_mask(width: int, height: int, tile: int, rank: int, world: int) -> torch.Tensor:
    """(H, W) bool: pixels whose tile belongs to `rank` (same rule as pixel_of() in kernels.hip)."""
    tx, ty, _ = tile_grid(width, height, tile)
    jj = torch.arange(height).unsqueeze(1) // tile
    ii = torch.arange(width).unsqueeze(0) // tile
    return ((jj * tx + ii) % world) == rank


def pack_tiles(img: torch.Tensor, tile: int, rank: int, world: int) -> torch.Tensor:
    """(H, W, C) image -> (tiles_per_rank, tile, tile, C) buffer holding this rank's tiles (zero padded)."""
    H, W, Cc = img.shape
    tx, ty, n_tiles = tile_grid(W, H, tile)
    per_rank = (n_tiles + world - 1) // world
    padded = torch.zeros((ty * tile, tx * tile, Cc), dtype=img.dtype, device=img.device)
    padded[:H, :W] = img
    t = padded.view(ty, tile, tx, tile, Cc).permute(0, 2, 1, 3, 4).reshape(n_tiles, tile, tile, Cc)
    mine = torch.zeros((per_rank, tile, tile, Cc), dtype=img.dtype, device=img.device)
    own = t[rank::world]
    mine[: own.shape[0]] = own
    return mine


def unpack_tiles(allb: torch.Tensor, width: int, height: int, tile: int) -> torch.Tensor:
    """(world, tiles_per_rank, tile, tile, C) gathered buffer -> (H, W, C) image (tile t = k*world + r)."""
    world, per_rank, _, _, Cc = allb.shape
    tx, ty, n_tiles = tile_grid(width, height, tile)
    full = allb.permute(1, 0, 2, 3, 4).reshape(per_rank * world, tile, tile, Cc)[:n_tiles]
    img = full.reshape(ty, tx, tile, tile, Cc).permute(0, 2, 1, 3, 4).reshape(ty * tile, tx * tile, Cc)
    return img[:height, :width]


def gather_frame_dev(rgb8: torch.Tensor, radiance: torch.Tensor, tile: int, rank: int, world: int, group=None, scratch=None, via_host=False, force=False):
    """GPU path of the exchange: native pack (bhrt_tiles_pack_dev) -> ONE all_gather of byte blocks -> native unpack
    into the same (H, W, 3) uint8 / float32 device tensors, in place.  `scratch`: dict reused across frames.
    via_host: move the blocks through host memory and a CPU backend (rehearsal of N ranks on one GPU, where RCCL
    cannot run); pack and unpack still run on the device.  force: go through pack / all_gather / unpack even in a world of one (tests: the RCCL
    calls of this path on a one-GPU box)."""
    import torch.distributed as dist
    import bhraytracer_amd as B
    if world == 1 and not force:
        return
    H, W, _ = rgb8.shape
    bb = B.tiles_block_bytes(W, H, tile, world)
    scratch = scratch if scratch is not None else {}
    if scratch.get("bytes") != (bb, world):
        scratch["mine"] = torch.empty(bb, dtype=torch.uint8, device=rgb8.device)
        scratch["all"] = torch.empty(bb * world, dtype=torch.uint8, device=rgb8.device)
        scratch["bytes"] = (bb, world)
    stream = torch.cuda.current_stream().cuda_stream
    B.tiles_pack_dev(rgb8.data_ptr(), radiance.data_ptr(), W, H, tile, rank, world, scratch["mine"].data_ptr(), stream)
    if via_host:
        torch.cuda.current_stream().synchronize()
        all_cpu = torch.empty(bb * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(all_cpu, scratch["mine"].cpu(), group=group)
        scratch["all"].copy_(all_cpu)
    else:
        dist.all_gather_into_tensor(scratch["all"], scratch["mine"], group=group)  # RCCL over xGMI
    B.tiles_unpack_dev(scratch["all"].data_ptr(), W, H, tile, world, rgb8.data_ptr(), radiance.data_ptr(), stream)


def gather_framebuffer(img: torch.Tensor, tile: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All ranks end up with the complete image. One all_gather, (W*H*C*itemsize)/world bytes per rank."""
    import torch.distributed as dist
    if world == 1:
        return img
    mine = pack_tiles(img, tile, rank, world).contiguous()
    allb = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=img.dtype, device=img.device)
    dist.all_gather_into_tensor(allb, mine, group=group)  # rank r's block lands at rows [r*per_rank, (r+1)*per_rank)
    return unpack_tiles(allb.view((world,) + tuple(mine.shape)), img.shape[1], img.shape[0], tile)


def photon_build_sharded(scene, opts, max_photons: int, rank: int, world: int, group=None, batch: int = 1 << 20, device=None) -> int:
    """BuildCausticPhotonMap over `world` GPUs.  Emission is keyed by the emission index: per batch of `batch`
    emissions (2^20, the batch of the single-GPU build, so both stop after the same number of emissions when a scene runs out of
    its emission budget) rank r runs the r-th slice on its GPU (bhrt_photon_emit_range), ONE all_gather moves the slices' records
    to everybody, and every rank keeps the first max_photons records in emission order and installs the same map
    (bhrt_photon_install) — byte for byte the map bhrt_photon_build makes on one GPU.
    `device`: where the exchange buffers live.  A GPU (backend "nccl" = RCCL): the records go from the emission kernel's output
    through the all_gather into the install without touching the host.  None: host memory (gloo).
    A failure on one rank (HIP error, ...) is carried in the size exchange, so every rank raises instead of waiting forever."""
    import torch.distributed as dist
    blocks = batch // 256
    assert blocks >= world and batch % 256 == 0
    lo, hi = 256 * (blocks * rank // world), 256 * (blocks * (rank + 1) // world)  # this rank's slice of every batch
    cap = max(4 * (hi - lo), 4096)
    mine = torch.empty((cap, 24), dtype=torch.uint8, device=device)
    kept, total, e0 = [], 0, 0
    budget = max_photons * 4096 + (1 << 24)  # like BuildPhotons: a scene without a caustic path ends here
    while total < max_photons and e0 < budget:
        n, err = 0, None
        try:
            n, ok = scene.photon_emit_range_into(opts, e0 + lo, hi - lo, mine.data_ptr(), cap)
            if not ok:  # more photons than there was room for: once more with room for all
                cap = n
                mine = torch.empty((cap, 24), dtype=torch.uint8, device=device)
                n, ok = scene.photon_emit_range_into(opts, e0 + lo, hi - lo, mine.data_ptr(), cap)
                assert ok
        except Exception as e:  # noqa: BLE001 — reported to every rank below
            n, err = -1, e
        if world == 1:
            if err is not None:
                raise err
            blocks_now = [mine[:n].clone()]
        else:
            sizes = torch.zeros(world, dtype=torch.int64, device=device)
            sizes[rank] = n
            dist.all_reduce(sizes, group=group)
            sizes = sizes.cpu()
            if int(sizes.min()) < 0:
                bad = [r for r in range(world) if int(sizes[r]) < 0]
                raise RuntimeError(f"photon emission failed on rank(s) {bad}" + (f": {err}" if err is not None else ""))
            width = max(int(sizes.max()), 1)
            buf = torch.zeros((width, 24), dtype=torch.uint8, device=device)
            buf[:n] = mine[:n]
            allb = torch.empty((world * width, 24), dtype=torch.uint8, device=device)
            dist.all_gather_into_tensor(allb, buf, group=group)
            blocks_now = [allb[r * width: r * width + int(sizes[r])] for r in range(world)]
        for blk in blocks_now:  # rank order = emission order
            kept.append(blk)
            total += len(blk)
        e0 += batch
    if total == 0:
        raise RuntimeError("photon map: no photon reached a photon surface")
    rec = torch.cat(kept)[:max_photons].contiguous()
    if device is not None:
        torch.cuda.synchronize()
    return scene.photon_install_ptr(rec.data_ptr(), len(rec))
