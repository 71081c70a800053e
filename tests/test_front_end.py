"""Host front-end: BVH build (two independent builders agree node for node), BVH invariants, OBJ rules,
XML schema coverage of the reference's own scene files."""
import glob
import os

import numpy as np
import pytest

from conftest import REFERENCE, SCENES


def test_bvh_front_end_equals_oracle_restatement(load_scene, O):
    # bhraytracer_amd/csrc/scene_host.cpp::BuildBvh vs oracle/bhrt_oracle.cpp::oracle_bvh_build (cyBVH.h:122-142)
    for name in ("c3_mesh_small", "c4_textured"):
        m = load_scene(name).flat_view().mesh_arrays(0)
        nodes, elems = O.bvh_build(m["v"], m["f"], 4)
        assert nodes.shape[0] == m["bvh_raw"].shape[0]
        assert np.array_equal(nodes[1:], m["bvh_raw"][1:])
        assert np.array_equal(elems, m["elems"])


def test_bvh_invariants(load_scene):
    fv = load_scene("c3_mesh_small").flat_view()
    m = fv.mesh_arrays(0)
    data, bounds, parent, v, f = m["bvh_data"], m["bvh_bounds"], m["bvh_parent"], m["v"], m["f"]
    seen = np.zeros(len(f), int)
    stack = [(1, 0)]
    maxdepth = 0
    while stack:
        n, d = stack.pop()
        maxdepth = max(maxdepth, d)
        if data[n] & 0x80000000:
            cnt, off = ((data[n] >> 28) & 7) + 1, data[n] & 0x0FFFFFFF
            assert cnt <= 4                                   # objects.h:59: SetMesh(this, 4)
            for e in m["elems"][off:off + cnt]:
                seen[e] += 1
                tri = v[f[e]]
                assert np.all(tri.min(0) >= bounds[n, :3]) and np.all(tri.max(0) <= bounds[n, 3:])
        else:
            c1 = data[n] & 0x7FFFFFFF
            assert c1 % 2 == 0                                # first child even -> sibling = id ^ 1 (device_trace.h)
            for c in (c1, c1 + 1):
                assert parent[c] == n
                assert np.all(bounds[c, :3] >= bounds[n, :3]) and np.all(bounds[c, 3:] <= bounds[n, 3:])
                stack.append((c, d + 1))
    assert np.all(seen == 1)                                  # every triangle in exactly one leaf
    assert maxdepth == fv.meshes[0].bvh_depth <= 64


def test_obj_rules(B, tmp_path):
    # fan triangulation, negative (relative) indices, v/vt/vn forms, comments, blank runs (cyTriMesh.h:379-438)
    obj = tmp_path / "t.obj"
    obj.write_text("# comment\n\nv 0 0 0\nv 1 0 0\nv   1 1 0\nv 0 1 0\nv 0.5 0.5 1\n"
                   "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                   "f 1/1 2/2 3/3 4/4\n"          # quad -> (0,1,2) (0,2,3)
                   "f -5/-5 -4/-4 -1/-1\n"        # relative indices
                   "f 2/2   3/3\t5/5\n")
    xml = tmp_path / "s.xml"
    xml.write_text(f"""<xml><scene><object type="obj" name="{obj}" material="m"/>
      <material type="blinn" name="m"/><light type="point" name="l"><intensity value="1"/></light></scene>
      <camera><position z="5"/><target z="0"/><up y="1"/></camera></xml>""")
    m = B.Scene(str(xml)).flat_view().mesh_arrays(0)
    assert m["f"].tolist()[:3] == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    assert m["ft"].tolist()[:3] == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    assert m["f"].shape[0] == 4
    # no vn in the file -> area-weighted vertex normals, unit length (cyTriMesh.h:248-261)
    assert np.allclose(np.linalg.norm(m["vn"], axis=1), 1.0, atol=1e-6)
    assert np.array_equal(m["fn"], m["f"])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference scenes only exist in the development container")
def test_all_shipped_scenes_load(B):
    files = sorted(glob.glob(os.path.join(REFERENCE, "Resource", "**", "*.xml"), recursive=True))
    assert len(files) == 19
    cwd = os.getcwd()
    os.chdir(REFERENCE)  # the reference resolves asset paths against its working directory
    try:
        for f in files:
            sc = B.Scene(f)
            assert sc.info.n_nodes > 0 and sc.width > 0
            # every mesh is absent from the repository (.gitignore:75): warning + null object, never an error
            for w in sc.warnings():
                assert "Cannot load file" in w or "texture" in w
    finally:
        os.chdir(cwd)


def test_png_textures_of_every_bit_depth(B, tmp_path):
    """Texture.cpp:76 decodes with lodepng::decode(..., LCT_RGB, 8), which accepts every colour type and bit depth; the front-end's own
    reader has to deliver the same RGB8 texels (16-bit samples: the high byte; grey below 8 bits: scaled to 0..255; palettes of any
    index width).  An interlaced file and a file with absurd dimensions are 'texture failed to load' warnings, not crashes."""
    import struct, zlib
    from PIL import Image
    rng = np.random.RandomState(4)
    W, H = 19, 7                                   # odd width: sub-byte rows end inside a byte
    cases = {}
    g8 = rng.randint(0, 256, (H, W)).astype(np.uint8)
    for bits in (1, 2, 4):
        v = (g8 >> (8 - bits)).astype(np.uint8)
        rows = []
        for y in range(H):
            bitstr = "".join(format(int(x), f"0{bits}b") for x in v[y])
            bitstr += "0" * (-len(bitstr) % 8)
            rows.append(b"\x00" + bytes(int(bitstr[k:k + 8], 2) for k in range(0, len(bitstr), 8)))
        cases[f"grey{bits}"] = ((0, bits, b"".join(rows)), np.repeat((v.astype(np.uint32) * 255 // ((1 << bits) - 1)).astype(np.uint8)[..., None], 3, 2))
    g16 = rng.randint(0, 65536, (H, W)).astype(">u2")
    cases["grey16"] = ((0, 16, b"".join(b"\x00" + g16[y].tobytes() for y in range(H))), np.repeat((g16 >> 8).astype(np.uint8)[..., None], 3, 2))
    rgb16 = rng.randint(0, 65536, (H, W, 3)).astype(">u2")
    cases["rgb16"] = ((2, 16, b"".join(b"\x00" + rgb16[y].tobytes() for y in range(H))), (rgb16 >> 8).astype(np.uint8))

    def write_png(path, ctype, bits, raw, plte=None, interlace=0, w=W, h=H):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, ctype, 0, 0, interlace))
        if plte is not None:
            data += chunk(b"PLTE", plte)
        open(path, "wb").write(data + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))

    pal = rng.randint(0, 256, (16, 3)).astype(np.uint8)
    idx = rng.randint(0, 16, (H, W)).astype(np.uint8)
    rows = []
    for y in range(H):
        nib = list(idx[y]) + [0] * (W % 2)
        rows.append(b"\x00" + bytes((nib[k] << 4) | nib[k + 1] for k in range(0, len(nib), 2)))
    expected = {k: v[1] for k, v in cases.items()}
    for name, ((ctype, bits, raw), _) in cases.items():
        write_png(tmp_path / f"{name}.png", ctype, bits, raw)
    write_png(tmp_path / "pal4.png", 3, 4, b"".join(rows), plte=pal.tobytes())
    expected["pal4"] = pal[idx]
    Image.fromarray(rng.randint(0, 256, (H, W, 4)).astype(np.uint8), "RGBA").save(tmp_path / "rgba8.png")   # PIL's encoder: filters, several chunks
    expected["rgba8"] = np.asarray(Image.open(tmp_path / "rgba8.png").convert("RGB"))
    write_png(tmp_path / "interlaced.png", 0, 8, b"\x00" * 50, interlace=1)
    write_png(tmp_path / "huge.png", 2, 8, b"\x00" * 64, w=1 << 30, h=1 << 30)
    names = list(expected) + ["interlaced", "huge"]
    mats = "".join(f'<material type="blinn" name="m{k}"><diffuse texture="{n}.png"/></material>' for k, n in enumerate(names))
    objs = "".join(f'<object type="sphere" name="s{k}" material="m{k}"><translate x="{3 * k}"/></object>' for k in range(len(names)))
    xml = tmp_path / "tex.xml"
    xml.write_text(f"<xml><scene>{objs}{mats}<light type=\"point\" name=\"l\"><intensity value=\"1\"/><position z=\"9\"/></light></scene>"
                   "<camera><position y=\"-9\"/><target z=\"0\"/><up z=\"1\"/><width value=\"8\"/><height value=\"8\"/></camera></xml>")
    sc = B.Scene(str(xml))
    fv = sc.flat_view()
    assert len(fv.textures) == len(expected)
    for t, name in zip(fv.textures, expected):
        assert (t.width, t.height) == (W, H), name
        texels = fv.np(t.off_data, W * H * 3, np.uint8).reshape(H, W, 3)
        assert np.array_equal(texels, expected[name]), name
    w = [x for x in sc.warnings() if "texture" in x]
    assert len(w) == 2 and "interlaced.png" in w[0] and "huge.png" in w[1]
