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
