"""The C++ host program above the C ABI (bhraytracer_amd/bhrt, csrc/bhrt_main.cpp = the reference's main(), Main.cpp:418-431):
XML scene in, PNG out (RenderImage::SaveImage, Scenes/scene.h:628-644), on one GPU and through the multi-GPU path (one process, one
host thread per GPU, RCCL all-gather of the packed tiles) over a one-device communicator — what a one-GPU box can run."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENES, same_bits

CLI = os.path.join(ROOT, "bhraytracer_amd", "bhrt")


def _run(args, cwd):
    r = subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _png(path):
    from PIL import Image
    im = Image.open(path)
    assert im.mode == "RGB"
    return np.asarray(im)


def test_cli_info_without_a_gpu(tmp_path):
    out = _run(["info", os.path.join(SCENES, "c3_mesh_small.xml")], SCENES)
    assert "Render image width: 320" in out and "Render image height: 240" in out and "288 triangles" in out
    r = subprocess.run([CLI, "render"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["one_device", "gpus_1"])
def test_cli_renders_the_oracle_s_image(mode, load_scene, O, tmp_path):
    sc = load_scene("c3_room_small")
    png, f32 = str(tmp_path / "x.png"), str(tmp_path / "x.f32")
    extra = ["--gpus", "1"] if mode == "gpus_1" else ["--device", "0"]
    out = _run(["render", os.path.join(SCENES, "c3_room_small.xml"), "-o", png, "--radiance", f32, "--spp", "3", "--gi", "3", "--seed", "9"] + extra, SCENES)
    assert "wrote " + png in out
    if mode == "gpus_1":
        assert "RCCL tile gather" in out
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 3, gi=3, seed=9, want_samples=False, threads=16)
    img = _png(png)
    assert img.shape == (sc.height, sc.width, 3) and np.array_equal(img, ro["rgb8"])          # gamma + Color24 + PNG encode / decode
    rad = np.fromfile(f32, np.float32).reshape(sc.height, sc.width, 3)
    assert np.nanmax(np.abs(rad - ro["radiance"])) <= 1e-4 and same_bits(rad, ro["radiance"])


@pytest.mark.gpu
def test_cli_photon_map_over_the_multi_gpu_path(load_scene, O, B, tmp_path):
    """--gpus 1 --photons N: emission by ranges + install (the multi-GPU build) gives the single-GPU map, and the frame with the
    caustic term equals the frame of the library's own build."""
    xml = os.path.join(SCENES, "c5_caustics.xml")
    a, b = str(tmp_path / "a.dat"), str(tmp_path / "b.dat")
    _run(["render", xml, "-o", str(tmp_path / "a.png"), "--radiance", str(tmp_path / "a.f32"), "--spp", "2", "--gi", "2", "--photons", "20000", "--photon-out", a, "--gpus", "1"], SCENES)
    _run(["render", xml, "-o", str(tmp_path / "b.png"), "--radiance", str(tmp_path / "b.f32"), "--spp", "2", "--gi", "2", "--photons", "20000", "--photon-out", b], SCENES)
    assert open(a, "rb").read() == open(b, "rb").read() and os.path.getsize(a) == 20000 * 24
    assert np.array_equal(_png(str(tmp_path / "a.png")), _png(str(tmp_path / "b.png")))
    assert open(tmp_path / "a.f32", "rb").read() == open(tmp_path / "b.f32", "rb").read()
    # the cached photon pass: --photon-file
    _run(["render", xml, "-o", str(tmp_path / "c.png"), "--spp", "2", "--gi", "2", "--photon-file", a, "--gpus", "1"], SCENES)
    assert np.array_equal(_png(str(tmp_path / "c.png")), _png(str(tmp_path / "a.png")))


@pytest.mark.gpu
def test_cli_eight_ranks_rehearsed_on_one_gpu(tmp_path):
    """`bhrt render --gpus 8 --rehearse`: the N > 1 control flow of the C++ host — eight host threads with a scene handle each, the three
    rendezvous points, per-rank bhrt_render_dev of the interleaved tiles, bhrt_tiles_pack_dev, the exchange (N device-to-device copies in the
    place of ncclAllGather: a one-GPU box has no peers), bhrt_tiles_unpack_dev — on BASELINE config 4's 3840x2160 frame at 1 spp: the PNG and
    the radiance are the single-rank render's, byte for byte."""
    from conftest import ensure_mesh
    ensure_mesh(224)
    xml = os.path.join(SCENES, "c4_mesh_4k.xml")
    a, b = tmp_path / "a", tmp_path / "b"
    out = _run(["render", xml, "-o", f"{a}.png", "--radiance", f"{a}.f32", "--spp", "1", "--gi", "3", "--seed", "5", "--gpus", "8", "--rehearse"], SCENES)
    assert "8 GPU(s)" in out and "rehearsed" in out and out.count("GPU ") >= 8
    _run(["render", xml, "-o", f"{b}.png", "--radiance", f"{b}.f32", "--spp", "1", "--gi", "3", "--seed", "5", "--device", "0"], SCENES)
    assert open(f"{a}.png", "rb").read() == open(f"{b}.png", "rb").read()
    assert open(f"{a}.f32", "rb").read() == open(f"{b}.f32", "rb").read() and os.path.getsize(f"{a}.f32") == 3840 * 2160 * 12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 3])
def test_cli_photon_map_of_n_rehearsed_ranks(n, tmp_path):
    """The caustic map emitted by N ranks (every batch of 2^20 emissions cut into N index ranges like dist.py::photon_build_sharded cuts it —
    N = 3 does not divide the batch's 4096 blocks), strung together on the host and installed on every rank, is the single-GPU map file for
    file, and the frame gathered from the N ranks is the single-rank frame."""
    xml = os.path.join(SCENES, "c5_caustics.xml")
    a, b = tmp_path / "a", tmp_path / "b"
    args = ["--spp", "2", "--gi", "2", "--photons", "30000"]
    out = _run(["render", xml, "-o", f"{a}.png", "--radiance", f"{a}.f32", "--photon-out", f"{a}.dat", "--gpus", str(n), "--rehearse"] + args, SCENES)
    assert f"on {n} GPU(s)" in out
    _run(["render", xml, "-o", f"{b}.png", "--radiance", f"{b}.f32", "--photon-out", f"{b}.dat"] + args, SCENES)
    assert open(f"{a}.dat", "rb").read() == open(f"{b}.dat", "rb").read() and os.path.getsize(f"{a}.dat") == 30000 * 24
    assert open(f"{a}.png", "rb").read() == open(f"{b}.png", "rb").read()
    assert open(f"{a}.f32", "rb").read() == open(f"{b}.f32", "rb").read()


@pytest.mark.gpu
def test_cli_a_rank_that_fails_before_the_collective_does_not_hang_the_others(tmp_path):
    """A rank whose pack fails (test knob BHRT_TEST_FAIL_PACK = rank) keeps every rank out of the exchange: the program ends with an error
    instead of leaving N - 1 threads in an all-gather that never completes."""
    xml = os.path.join(SCENES, "c2_glass_small.xml")
    r = subprocess.run([CLI, "render", xml, "-o", str(tmp_path / "x.png"), "--spp", "1", "--gpus", "4", "--rehearse"], cwd=SCENES, capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, BHRT_TEST_FAIL_PACK="2"))
    assert r.returncode == 1 and "pack failed" in r.stderr and not (tmp_path / "x.png").exists()
