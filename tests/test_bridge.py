"""The drop-in binding of INTEGRATION.md as a build product (oracle/ref_harness/bhrt_bridge.cpp -> oracle/_ref/bhrt_bridge): the
reference program — its own Main.cpp as text, its own translation units, the GLUT stub of ref_harness — with the bodies of
BeginRender / StopRender (Main.cpp:178-245) replaced by calls into libbhrt.so through include/bhrt.h.  `make -C oracle bridge`
needs /root/reference (dev container); the binary travels to the GPU box with oracle/_ref/.

CPU: it compiles against the reference's headers, links, loads the scene with the reference's own LoadScene, and BeginRender() ends
with BHRT_ERR_NO_DEVICE cleanly (no crash, no image written).  GPU: the PNG the reference program writes through the bridge
(RenderImage::SaveImage = lodepng, Scenes/scene.h:628-644) holds the pixels of the oracle's frame."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REFERENCE, ROOT, SCENES, have_gpu

BRIDGE = os.path.join(ROOT, "oracle", "_ref", "bhrt_bridge")


def _build():
    if os.path.exists(os.path.join(REFERENCE, "Main.cpp")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "bridge"], check=True)
    if not os.path.exists(BRIDGE):
        pytest.skip("oracle/_ref/bhrt_bridge is built where /root/reference exists (dev container) and travels from there")


def test_bridge_links_and_reports_no_device_cleanly(tmp_path):
    _build()
    if have_gpu():
        pytest.skip("a GPU is present: covered by the gpu test below")
    out = tmp_path / "x.png"
    r = subprocess.run([BRIDGE, "c1_sphere_plane.xml", str(out), "2"], cwd=SCENES, capture_output=True, text=True, timeout=120)
    assert r.returncode == 4, r.stdout[-1500:] + r.stderr[-1500:]          # BHRT_ERR_NO_DEVICE (include/bhrt.h)
    assert "object [ball] <ball> - Sphere" in r.stdout                        # the reference's own LoadScene ran (xmlload.cpp:65)
    assert "bhrt: no HIP device" in r.stdout and not out.exists()
    # the symbols of the seam are the bridge's, the reference's bodies are still in the binary under their other names
    syms = subprocess.run(["nm", "-C", BRIDGE], capture_output=True, text=True).stdout
    assert " T BeginRender()" in syms and " T BeginRender_ref()" in syms and " T StopRender()" in syms and "bhrt_render" in syms


@pytest.mark.gpu
def test_bridge_writes_the_oracle_s_image(load_scene, O, tmp_path):
    if not os.path.exists(BRIDGE):
        pytest.skip("oracle/_ref/bhrt_bridge not built (needs /root/reference in the dev container)")
    from PIL import Image
    sc = load_scene("c2_glass_small")
    out = tmp_path / "x.png"
    r = subprocess.run([BRIDGE, "c2_glass_small.xml", str(out), "4"], cwd=SCENES, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Mrays/s" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 4, gi=3, seed=0, want_samples=False, threads=16)   # bhrt_default_opts: GI 3, 16 bounces, seed 0
    img = np.asarray(Image.open(out))
    assert img.shape == (sc.height, sc.width, 3) and np.array_equal(img, ro["rgb8"])
