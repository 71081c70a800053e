"""The C-ABI boundary: the library loads, exports every symbol include/bhrt.h declares, the host-side
entry points work without a GPU, and compute entry points refuse (no CPU fallback) without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, SCENES, have_gpu


def test_exports_every_declared_symbol(B):
    hdr = open(os.path.join(ROOT, "include", "bhrt.h")).read()
    declared = set(re.findall(r"\b(bhrt_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = B.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libbhrt.so lacks {missing}"
    assert declared == set(B.EXPORTS), "bhraytracer_amd.EXPORTS out of sync with include/bhrt.h"


def test_struct_layouts_match_header(B):
    # sizes the C side reports through behaviour: default opts round-trip and info fields
    o = B.default_opts()
    assert (o.spp, o.gi_bounces, o.internal_bounces, o.jitter, o.gamma, o.world_size, o.tile_size) == (32, 3, 16, 1, 1, 1, 32)
    assert C.sizeof(B.Opts) == 16 * 4
    assert C.sizeof(B.Hits) == 4 * C.sizeof(C.c_void_p)


def test_load_scene_and_info(load_scene):
    sc = load_scene("c3_mesh_small")
    i = sc.info
    assert (i.width, i.height) == (320, 240)
    assert (i.n_nodes, i.n_meshes, i.n_triangles, i.n_lights, i.n_materials) == (3, 1, 288, 1, 3)
    fv = sc.flat_view()
    assert fv.header.total_bytes == len(sc.flat_bytes()) == i.flat_bytes
    assert [n.obj_type for n in fv.nodes] == [2, 3, 1]


def test_error_paths(B, tmp_path):
    with pytest.raises(B.BhrtError):
        B.Scene(str(tmp_path / "does_not_exist.xml"))
    bad = tmp_path / "bad.xml"
    bad.write_text("<xml><scene></scene></xml>")  # no <camera> tag: LoadScene returns 0 (xmlload.cpp:85-89)
    with pytest.raises(B.BhrtError, match="camera"):
        B.Scene(str(bad))
    bad.write_text("<xml><scene><object type='sphere'></scene></xml>")
    with pytest.raises(B.BhrtError):
        B.Scene(str(bad))


def test_missing_mesh_is_a_warning_not_an_error(B, tmp_path):
    # xmlload.cpp:209-214,253: the node keeps a null object and the rest of the scene loads
    xml = tmp_path / "missing_mesh.xml"
    xml.write_text("""<xml><scene>
      <object type="obj" name="Resource\\Data\\teapot.obj" material="m"><scale value="2"/></object>
      <object type="sphere" name="s" material="m"/>
      <material type="blinn" name="m"><diffuse value="0.5"/></material>
      <light type="point" name="l"><intensity value="10"/><position z="10"/></light>
    </scene><camera><position z="10"/><target z="0"/><up y="1"/></camera></xml>""")
    sc = B.Scene(str(xml))
    assert sc.info.n_nodes == 2 and sc.info.n_meshes == 0
    assert any("teapot.obj" in w for w in sc.warnings())
    assert [n.obj_type for n in sc.flat_view().nodes] == [0, 1]
    assert (sc.width, sc.height) == (200, 150)  # Camera::Init defaults (scene.h:521-522)


def test_png_round_trip(B, tmp_path):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    p = str(tmp_path / "o.png")
    B.save_png(p, img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), img)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device behaviour")
def test_compute_refuses_without_a_device(load_scene, B):
    sc = load_scene("c1_sphere_plane")
    with pytest.raises(B.BhrtError, match="(?i)device"):
        sc.upload(0)
    with pytest.raises(B.BhrtError):
        sc.trace_closest(np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32))
    with pytest.raises(B.BhrtError):
        sc.render(B.default_opts(spp=1))
    # the entry points added later: global photon map, images beside the colour image, device BVH build
    with pytest.raises(B.BhrtError):
        sc.photon_build_global(B.default_opts(), 100)
    with pytest.raises(B.BhrtError):
        sc.first_hit()
    with pytest.raises(B.BhrtError, match="(?i)device"):
        B.bvh_build(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), np.uint32([[0, 1, 2]]))
    from conftest import SCENES
    import os
    with pytest.raises(B.BhrtError):                    # a scene with a mesh asked to build its BVH on a device that is not there
        B.Scene(os.path.join(SCENES, "c3_mesh_small.xml"), bvh_device=0)
    assert B.Scene(os.path.join(SCENES, "c3_mesh_small.xml")).info.n_triangles == 288   # the host front-end needs no device
