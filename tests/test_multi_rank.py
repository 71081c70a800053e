"""The N > 1 path on CPU: two gloo ranks pack their interleaved tiles, all_gather them and de-interleave —
the same code bench.py runs over RCCL/xGMI (bhraytracer_amd/dist.py).  The per-rank 'render' here is the CPU
oracle restricted to the rank's tiles, so the gathered image must equal the single-rank oracle image."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENES


def _worker(rank, world, port, blob, W, H, tile, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bhraytracer_amd.dist as BD
    import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = O.render(blob, W, H, 1, gi=1, threads=2, want_samples=False)
    mask = BD.owned_mask(W, H, tile, rank, world)
    rad = torch.from_numpy(full["radiance"].copy())
    rad[~mask] = 0                      # this rank only "rendered" its own tiles
    rgb = torch.from_numpy(full["rgb8"].copy())
    rgb[~mask] = 0
    g_rad = BD.gather_framebuffer(rad, tile, rank, world)
    g_rgb = BD.gather_framebuffer(rgb, tile, rank, world)
    ok = torch.equal(g_rad, torch.from_numpy(full["radiance"])) and torch.equal(g_rgb, torch.from_numpy(full["rgb8"]))
    share = int(mask.sum())
    t = torch.tensor([share], dtype=torch.int64)
    dist.all_reduce(t)
    ok = ok and int(t.item()) == W * H
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile", [32, 20])
def test_two_rank_tile_gather_gloo(tile, load_scene, tmp_path):
    sc = load_scene("c2_glass_small")
    port = 29500 + (os.getpid() % 2000) + tile
    mp.spawn(_worker, args=(2, port, sc.flat_bytes(), sc.width, sc.height, tile, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "1" and open(tmp_path / "ok1").read() == "1"


def test_pack_unpack_roundtrip_any_world():
    import bhraytracer_amd.dist as BD
    W, H = 75, 50
    img = torch.arange(W * H * 3, dtype=torch.float32).reshape(H, W, 3)
    for tile in (8, 32, 17):
        for world in (1, 2, 3, 8):
            packs = [BD.pack_tiles(torch.where(BD.owned_mask(W, H, tile, r, world).unsqueeze(-1), img, torch.zeros_like(img)), tile, r, world)
                     for r in range(world)]
            assert torch.equal(BD.unpack_tiles(torch.stack(packs), W, H, tile), img)
            masks = torch.stack([BD.owned_mask(W, H, tile, r, world) for r in range(world)])
            assert torch.equal(masks.sum(0), torch.ones(H, W, dtype=torch.int64))


def test_bench_never_falls_through_to_one_gpu():
    """`bench.py --gpus N` must run N ranks or fail: with a torchrun environment of another size it refuses (exit code != 0) instead of
    benchmarking a different world; without one it starts the ranks itself (covered on the GPU box by the test below)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun: two ranks are started (here both on the one GPU of the box, exchange through gloo:
    --rehearse-on-one-gpu), the JSON line reports n_gpus = 2 and the spp of two ranks."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload", "c1", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "2 spp (1 per GPU, weak scaling)" in d["config"]["workload"] and d["value"] > 0


@pytest.mark.gpu
def test_bench_strong_scaling_of_the_headline_over_four_ranks():
    """`python bench.py --gpus 4 --scaling strong` on the headline workload (BASELINE config 3: 64 spp for the WHOLE job, the frame split over the
    ranks by tiles): four ranks rehearsed on the one GPU of the box (a box allows at most six GPU processes; the exchange goes through gloo), the
    line says which scaling it measured and counts the rays of all ranks — the same rays as one rank rendering the whole frame."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--workload", "c3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-configs", "--samples-per-pass", str(1 << 22)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-on-one-gpu", "--scaling", "strong"] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and "64 spp (16 per GPU, strong scaling)" in d["config"]["workload"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert d1["n_gpus"] == 1 and d["config"]["rays_per_frame"] == d1["config"]["rays_per_frame"]
    assert d["config"]["camera_samples_per_frame"] == d1["config"]["camera_samples_per_frame"] == 1920 * 1080 * 64


_RCCL_WORLD_OF_ONE = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import bhraytracer_amd as B, bhraytracer_amd.dist as BD
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str({port})
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
sc = B.Scene({scene!r}); sc.upload(0)
H, W = sc.height, sc.width
rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev); rad = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
opts = B.default_opts(spp=2, gi_bounces=2, seed=4)
sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())
torch.cuda.synchronize()
rgb0, rad0 = rgb.clone(), rad.clone()
BD.gather_frame_dev(rgb, rad, 32, 0, 1, force=True)            # pack -> RCCL all_gather_into_tensor -> unpack, in place
torch.cuda.synchronize()
assert torch.equal(rgb, rgb0) and torch.equal(rad.view(torch.int32), rad0.view(torch.int32)), "frame changed by the exchange"
# the collectives photon_build_sharded issues, on device tensors
sizes = torch.zeros(1, dtype=torch.int64, device=dev); sizes[0] = 12345
dist.all_reduce(sizes); assert int(sizes.cpu()[0]) == 12345
buf = torch.arange(24 * 1000, dtype=torch.int32, device=dev).to(torch.uint8).reshape(1000, 24)
allb = torch.empty_like(buf); dist.all_gather_into_tensor(allb, buf); assert torch.equal(allb, buf)
dist.barrier(); dist.destroy_process_group()
print("rccl world of one: ok")
"""


@pytest.mark.gpu
def test_the_rccl_calls_of_the_exchange_in_a_world_of_one(tmp_path):
    """Multi-GPU runs are not ours to launch; what a one-GPU box can show is that the RCCL calls the N > 1 path makes — process group on a
    device, all_gather_into_tensor of the packed byte blocks between the native pack and unpack kernels on torch's stream, the int64
    all_reduce and the (n, 24) uint8 all_gather of the sharded photon build — run on device tensors and leave the frame as it was."""
    import subprocess
    code = _RCCL_WORLD_OF_ONE.format(root=ROOT, port=29700 + os.getpid() % 1000, scene=os.path.join(SCENES, "c2_glass_small.xml"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world of one: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_bench_line_of_the_default_run_carries_what_the_contract_asks_for():
    """`python bench.py` as the driver runs it (fewer steps here): ONE JSON line whose headline is BASELINE config 3 with `roofline` (incl. the
    lane-slot figures the PMC passes gave) and `cpu_baseline` (the reference where oracle/_ref travelled, else the port), the other configurations
    under `configs`, the D2H / PNG lines, exit code 0."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--config-steps", "1", "--config-warmup", "0", "--cpu-sample-spp", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None and d["unit"] == "Mrays/s"
    assert "c3_mesh.xml 1920x1080, 64 spp" in d["config"]["workload"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert rf["traffic"] and 0 < rf["valu_lane_frac"] < 1 and rf["valu_lanes_per_inst"] <= 64
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "2 of the workload's 64 samples per pixel" in cb["sample"]
    assert set(d["configs"]) == {"c2", "c3room", "c4", "c5"} and all("error" not in v and v["ms_per_step"] > 0 for v in d["configs"].values())
    assert d["configs"]["c5"]["photon"]["build_s"] < 0.5 and d["configs"]["c5"]["photon"]["lane_pass_found_per_query"] > 1
    assert d["wall_clock_extra"]["d2h_ms"] > 0 and d["wall_clock_extra"]["png_encode_ms"] > 0
