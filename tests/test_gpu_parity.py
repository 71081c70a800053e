"""Parity of the HIP path (through the C ABI, libbhrt.so) with the CPU oracle and with the golden vectors the
compiled reference produced.  Bars (BASELINE.json north_star): hit indices bit-exact; float radiance within
1e-4 per channel — in practice every float below is bit-identical, which is what is asserted unless noted.
Run on the GPU box: python -m pytest tests -m gpu"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, SCENES, same_bits

pytestmark = pytest.mark.gpu
TOL = 1e-4  # per-channel radiance tolerance stated by north_star


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gpu(B):
    if B.device_count() < 1:
        pytest.fail("no HIP device: the render path has no CPU fallback, GPU tests cannot run here")
    return B


# ---------------------------------------------------------------------------------------------------- hits
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_primary_hits_vs_reference_golden(case, gpu, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    o, d = O.primary_rays(sc.flat_view())
    h = sc.trace_closest(o, d, gpu.SIDE_FRONT)
    H, W = int(g["height"]), int(g["width"])
    assert sha(h["node"].reshape(H, W).astype(np.int32)) == str(g["primary_node_sha"])   # every pixel, reference's own table
    assert sha(h["t"].reshape(H, W)) == str(g["primary_z_sha"])
    st = int(g["primary_step"])
    hit = g["primary_node"] >= 0
    assert np.array_equal(h["front"].reshape(H, W)[::st, ::st][hit], g["primary_front"][hit])


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("side", [1, 2, 3])
def test_secondary_rays_vs_golden_and_oracle(case, side, gpu, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    h = sc.trace_closest(g["rays_o"], g["rays_d"], side)
    assert np.array_equal(h["node"], g[f"rays_node_{side}"])
    hit = h["node"] >= 0
    assert same_bits(h["t"][hit], g[f"rays_z_{side}"][hit])
    assert np.array_equal(h["front"][hit], g[f"rays_front_{side}"][hit])
    r = O.trace_closest(sc.flat_bytes(), g["rays_o"], g["rays_d"], side)
    assert np.array_equal(h["prim"], r["prim"])          # triangle ids (the reference does not keep them)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_shadow_rays_vs_golden(case, gpu, load_scene, golden):
    g = golden(case)
    sc = load_scene(case)
    vis = sc.trace_shadow(g["shadow_o"], g["shadow_d"], 1.0)
    assert np.array_equal(vis.astype(np.int8), g["shadow_vis"])


def test_edge_cases(gpu, load_scene, O):
    sc = load_scene("c3_mesh_small")
    blob = sc.flat_bytes()
    # empty batch
    h = sc.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert h["node"].shape == (0,)
    # ragged batch sizes around the wave / block size, degenerate rays (zero direction, axis-parallel, huge, NaN)
    rng = np.random.RandomState(3)
    for n in (1, 63, 64, 65, 255, 257, 1000):
        o = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[::7] = 0
        d[1::7, :2] = 0
        d[2::7, 0] = 0
        o[3::11] *= 1e18
        if n > 20:
            d[5, 1] = np.nan
            o[9, 2] = np.inf
        for side in (1, 3):
            h = sc.trace_closest(o, d, side)
            r = O.trace_closest(blob, o, d, side)
            assert np.array_equal(h["node"], r["node"])
            assert np.array_equal(h["prim"], r["prim"])
            assert same_bits(h["t"], r["t"])
        tmx = rng.uniform(0.5, 50, n).astype(np.float32)
        assert np.array_equal(sc.trace_shadow(o, d, tmx), O.trace_shadow(blob, o, d, tmx))
    tm = rng.uniform(0.5, 50, 500).astype(np.float32)
    o = rng.uniform(-10, 10, (500, 3)).astype(np.float32)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    assert np.array_equal(sc.trace_shadow(o, d, tm), O.trace_shadow(blob, o, d, tm))


def test_rays_through_bvh_box_corners(gpu, B, O, tmp_path):
    """The closest-hit traversal takes the slab comparisons from approximate quotients when they are decisive and evaluates a box
    exactly otherwise (device_trace.h::box_fast).  Rays from one BVH box corner through another graze box faces, edges and
    corners, which puts tMin ~ tMax and tmin(child 1) ~ tmin(child 2) within a few ulp: the hit records still have to equal
    the oracle's bit for bit."""
    import shutil
    from conftest import SCENES
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    xml = tmp_path / "m.xml"
    xml.write_text("""<xml><scene><object type="obj" name="mesh_small.obj" material="m"/>
      <material type="blinn" name="m"/><light type="point" name="l"><intensity value="1"/><position z="9"/></light></scene>
      <camera><position y="-9" z="1"/><target z="0"/><up z="1"/><width value="16"/><height value="16"/></camera></xml>""")
    sc = B.Scene(str(xml))
    v, f = [], []
    for line in open(tmp_path / "mesh_small.obj"):
        t = line.split()
        if t and t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t and t[0] == "f":
            f.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    nodes, _, _ = B.bvh_build(np.float32(v), np.uint32(f))
    boxes = nodes[1:, :6].copy().view(np.float32)
    rng = np.random.default_rng(8)
    corners = np.stack([np.where(rng.random((4000, 3)) < 0.5, boxes[rng.integers(0, len(boxes), 4000)][:, :3],
                                 boxes[rng.integers(0, len(boxes), 4000)][:, 3:]) for _ in range(2)])
    a, b = corners[0].astype(np.float32), corners[1].astype(np.float32)
    keep = np.abs(b - a).min(axis=1) > 0
    a, b = a[keep], b[keep]
    d = (b - a).astype(np.float32)
    o = (a - np.float32(3.0) * d).astype(np.float32)          # start outside, pass through both corners
    blob = sc.flat_bytes()
    for side in (1, 3):
        h, r = sc.trace_closest(o, d, side), O.trace_closest(blob, o, d, side)
        assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
    assert (r["node"] >= 0).sum() > len(o) // 10
    tm = np.full(len(o), 10.0, np.float32)
    assert np.array_equal(sc.trace_shadow(o, d, tm), O.trace_shadow(blob, o, d, tm))


def test_deep_bvh_takes_the_parent_link_traversal(gpu, B, O, tmp_path):
    """k_trace_mesh keeps the traversal path in LDS when every mesh has at most 32 BVH levels (and ids below 2^17) and walks
    parent links otherwise.  Triangles at exponentially spaced positions make MeanSplit peel off two at a time: 48 levels."""
    n = 100
    with open(tmp_path / "chain.obj", "w") as fp:
        for k in range(n):
            x = 1.6 ** k * 1e-6
            fp.write(f"v {x!r} -0.3 {0.2 + 0.001 * k!r}\nv {x * 1.05!r} 0.3 {0.2!r}\nv {x!r} 0.0 {0.6 + 0.002 * k!r}\n")
        fp.write("vt 0 0 0\nvn 0 -1 0\n")
        for k in range(n):
            fp.write(f"f {3 * k + 1}/1/1 {3 * k + 2}/1/1 {3 * k + 3}/1/1\n")
    xml = tmp_path / "deep_bvh.xml"
    xml.write_text("""<xml><scene><background r="0.1" g="0.1" b="0.2"/><environment r="0.5" g="0.5" b="0.5"/>
      <object type="plane" name="g" material="g"><scale value="50"/></object>
      <object type="obj" name="chain.obj" material="m"><scale x="0.00002" y="8" z="8"/><translate x="-6"/></object>
      <object type="sphere" name="s" material="m"><translate x="3" z="1"/></object>
      <material type="blinn" name="g"><diffuse r="0.7" g="0.7" b="0.7"/><specular value="0"/></material>
      <material type="blinn" name="m"><diffuse r="0.8" g="0.3" b="0.2"/><specular value="0.4"/><glossiness value="30"/></material>
      <light type="point" name="l"><intensity value="200"/><position x="2" y="-8" z="12"/><size value="1"/></light></scene>
      <camera><position x="0" y="-22" z="6"/><target x="0" y="0" z="2"/><up z="1"/><fov value="50"/><width value="128"/><height value="72"/></camera></xml>""")
    sc = B.Scene(str(xml))
    assert sc.info.max_bvh_depth > 32
    blob = sc.flat_bytes()
    gs, st = sc.render_samples(B.default_opts(spp=2, gi_bounces=2, seed=4), 0, 0, sc.width, sc.height)
    ro = O.render(blob, sc.width, sc.height, 2, gi=2, seed=4)["samples"]
    assert same_bits(gs, ro)
    z, _, _ = sc.first_hit()
    mesh_px = (O.first_hit(blob, sc.width, sc.height)[0].reshape(z.shape) == z).all()
    assert mesh_px and (z < 1e30).any()
    dev = B.Scene(str(xml), bvh_device=0)                       # the device build on a 48-level tree
    assert dev.flat_bytes() == blob


def test_mesh_with_more_than_2_17_bvh_nodes(gpu, B, O, tmp_path):
    """The LDS path of the BVH traversal holds 16-bit pair indices below 2^17 nodes and 32-bit ones above: a 320 k-triangle
    mesh (213 k nodes) takes the 32-bit kernels.  Primary hits of every pixel and GI radiance of a region against the oracle;
    the device BVH build against the host's on the same mesh."""
    import shutil, sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mesh
    mesh = os.path.join(SCENES, "gen", "mesh_400.obj")
    if not os.path.exists(mesh):
        gen_mesh.generate(mesh, 400)
    os.makedirs(tmp_path / "gen", exist_ok=True)
    shutil.copy(mesh, tmp_path / "gen" / "mesh_400.obj")
    xml = tmp_path / "big.xml"
    xml.write_text(open(os.path.join(SCENES, "c3_mesh.xml")).read().replace("gen/mesh_224.obj", "gen/mesh_400.obj")
                   .replace('<width value="1920"/>', '<width value="480"/>').replace('<height value="1080"/>', '<height value="270"/>'))
    sc = B.Scene(str(xml))
    assert sc.info.n_triangles == 320000 and sc.info.n_bvh_nodes > (1 << 17) and sc.info.max_bvh_depth <= 32
    blob = sc.flat_bytes()
    o, d = O.primary_rays(sc.flat_view())
    h, r = sc.trace_closest(o, d, 1), O.trace_closest(blob, o, d, 1)
    assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
    region = (180, 90, 300, 170)
    gs, _ = sc.render_samples(B.default_opts(spp=2, seed=3), *region)
    ro = O.render(blob, sc.width, sc.height, 2, seed=3, region=region)
    assert same_bits(gs, ro["samples"])
    assert B.Scene(str(xml), bvh_device=0).flat_bytes() == blob


def test_rays_that_go_through_several_different_meshes(gpu, B, O, tmp_path):
    """Three mesh nodes of two different meshes (one of them glass, one inside a rotated group) in a closed box: a GI or refraction ray
    enters one mesh's box after the other, so in the streamed mesh kernel (k_trace_mesh_stream: a wave walks ONE mesh at a time) lanes
    wait at a mesh node while their wave walks another mesh, and a ray's hit so far comes from an earlier mesh.  Every pixel's primary hit
    and the per-sample radiance of the whole (small) frame against the oracle."""
    import shutil, sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mesh
    gen_mesh.generate(str(tmp_path / "mesh_b.obj"), 20)  # 800 triangles; mesh_small.obj has 288
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    xml = tmp_path / "meshes.xml"
    xml.write_text("""<xml><scene><background r="0.1" g="0.1" b="0.2"/><environment value="0.4"/>
      <object type="plane" name="floor" material="wall"><scale value="14"/></object>
      <object type="plane" name="back" material="wall"><scale value="14"/><rotate angle="90" x="1"/><translate y="9" z="6"/></object>
      <object type="obj" name="mesh_small.obj" material="glass"><scale value="2.2"/><translate x="-3" y="1" z="2.4"/></object>
      <object type="obj" name="mesh_b.obj" material="red"><scale x="2.5" y="2" z="2.8"/><rotate angle="35" z="1"/><translate x="2.5" y="3" z="2.9"/></object>
      <object name="grp"><rotate angle="-20" z="1"/><translate x="0.5" y="-2.5" z="0"/>
        <object type="obj" name="mesh_small.obj" material="red"><scale value="1.3"/><translate z="1.4"/></object>
        <object type="sphere" name="s" material="mirror"><scale value="0.9"/><translate x="2.6" z="0.9"/></object>
      </object>
      <material type="blinn" name="wall"><diffuse r="0.7" g="0.7" b="0.65"/><specular value="0.1"/><glossiness value="20"/></material>
      <material type="blinn" name="red"><diffuse r="0.8" g="0.25" b="0.2"/><specular value="0.5"/><glossiness value="60"/></material>
      <material type="blinn" name="mirror"><diffuse value="0.05"/><specular value="0.9"/><glossiness value="2000"/></material>
      <material type="blinn" name="glass"><diffuse value="0.05"/><specular value="0.8"/><glossiness value="80"/><refraction value="0.85" index="1.5"/><absorption r="0.05" g="0.02" b="0.1"/></material>
      <light type="point" name="p"><intensity value="260"/><position x="-2" y="-9" z="14"/><size value="1.5"/></light>
      <light type="ambient" name="a"><intensity value="0.1"/></light>
      </scene><camera><position x="0.5" y="-17" z="6.5"/><target x="0" y="1" z="2.2"/><up z="1"/><fov value="38"/><width value="128"/><height value="96"/></camera></xml>""")
    sc = B.Scene(str(xml))
    assert sc.info.n_triangles == 288 + 800
    blob = sc.flat_bytes()
    fv = sc.flat_view()
    assert sum(1 for n in fv.nodes if n.obj_type == 3) == 3 and len(fv.meshes) >= 2
    o, d = O.primary_rays(fv)
    for side in (1, 3):
        h, r = sc.trace_closest(o, d, side), O.trace_closest(blob, o, d, side)
        assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
    hit_mesh_nodes = {int(n) for n in np.unique(r["node"]) if n >= 0 and fv.nodes[int(n)].obj_type == 3}
    assert len(hit_mesh_nodes) == 3  # all three mesh nodes are in view
    gs, st = sc.render_samples(B.default_opts(spp=4, gi_bounces=3, seed=21), 0, 0, sc.width, sc.height)
    ro = O.render(blob, sc.width, sc.height, 4, gi=3, seed=21, region=(0, 0, sc.width, sc.height))
    assert same_bits(gs, ro["samples"])
    assert st.closest_rays > 3 * st.camera_samples


@pytest.mark.parametrize("tex_xf", ['<scale x="0.1" y="0.1"/>', '<scale x="0.013" y="0.31"/><rotate angle="33" z="1"/><translate x="0.21" y="-0.37"/>',
                                    '<scale x="3" y="2"/>'])
def test_checker_footprints_from_inside_one_cell_to_the_horizon(gpu, B, O, tmp_path, tex_xf):
    """Texture::Sample's 31 footprint taps on a checker (scene.h:318-337; device_shade.h::checker_taps_in_one_cell skips the lookups when the
    footprint provably stays inside one half-tile): a large textured ground seen from low above it — footprints from a fraction of a cell
    under the camera to hundreds of cells at the horizon, cell edges at every distance, also through a rotated, anisotropic texture transform
    and with a texture coarser than the plane.  Every sample of the frame against the oracle, which looks all 32 taps up."""
    xml = tmp_path / "checker.xml"
    xml.write_text(f"""<xml><scene><background r="0.2" g="0.3" b="0.5"/><environment value="0.6"/>
      <object type="plane" name="ground" material="g"><scale value="60"/></object>
      <object type="plane" name="wall" material="w"><scale x="30" y="8" z="1"/><rotate angle="90" x="1"/><translate y="40" z="8"/></object>
      <object type="sphere" name="s" material="m"><scale value="2"/><translate x="3" y="6" z="2"/></object>
      <material type="blinn" name="g"><diffuse r="1" g="0.9" b="0.8" texture="checkerboard"><color1 r="0.1" g="0.15" b="0.2"/><color2 r="0.9" g="0.8" b="0.7"/>{tex_xf}</diffuse>
        <specular r="0.6" g="0.6" b="0.6" texture="checkerboard"><color1 value="0.1"/><color2 value="0.9"/><scale x="0.07" y="0.05"/></specular><glossiness value="40"/></material>
      <material type="blinn" name="w"><diffuse value="0.8" texture="checkerboard"><color1 r="0.7" g="0.2" b="0.2"/><color2 value="0.9"/><scale x="0.05" y="0.2"/></diffuse><specular value="0.2"/><glossiness value="10"/></material>
      <material type="blinn" name="m"><diffuse value="0.1"/><specular value="0.9"/><glossiness value="500"/></material>
      <light type="point" name="p"><intensity value="700"/><position x="-10" y="-5" z="25"/></light>
      </scene><camera><position x="0" y="-30" z="1.2"/><target x="0" y="10" z="1"/><up z="1"/><fov value="55"/><width value="320"/><height value="200"/></camera></xml>""")
    sc = B.Scene(str(xml))
    gs, st = sc.render_samples(B.default_opts(spp=3, gi_bounces=2, seed=4), 0, 0, sc.width, sc.height)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 3, gi=2, seed=4, region=(0, 0, sc.width, sc.height), threads=16)
    assert same_bits(gs, ro["samples"])
    assert len(np.unique(gs[:, :, 0])) > 1000  # the checker is there, filtered: far more than two shades


def test_nested_scene_graph_depth3(gpu, B, O, tmp_path):
    # recursive() back-transforms only through the hit node and its direct parent (Main.cpp:407-412):
    # a depth-3 node leaves p/N in its grandparent's space; t and node index are what the tracer returns
    xml = tmp_path / "deep.xml"
    xml.write_text("""<xml><scene>
      <object name="a"><translate x="1" z="2"/><rotate angle="30" z="1"/>
        <object name="b" type="sphere" material="m"><scale value="2"/><translate y="1"/>
          <object name="c" type="sphere" material="m"><scale x="0.5" y="0.7" z="0.4"/><translate x="2.5" z="1"/>
            <object name="d" type="plane" material="m"><scale value="3"/><rotate angle="70" x="1"/><translate z="-1"/></object>
          </object></object></object>
      <material type="blinn" name="m"/><light type="point" name="l"><intensity value="50"/><position z="15"/></light>
      </scene><camera><position y="-20" z="6"/><target z="2"/><up z="1"/><width value="160"/><height value="120"/></camera></xml>""")
    sc = B.Scene(str(xml))
    assert sc.info.max_node_depth == 4
    o, d = O.primary_rays(sc.flat_view())
    h = sc.trace_closest(o, d, 3)
    r = O.trace_closest(sc.flat_bytes(), o, d, 3)
    assert np.array_equal(h["node"], r["node"]) and same_bits(h["t"], r["t"])
    assert set(np.unique(r["node"])) >= {-1, 1, 2, 3}
    opts = B.default_opts(spp=2, gi_bounces=2)
    gs, _ = sc.render_samples(opts, 40, 30, 120, 90)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 2, gi=2, region=(40, 30, 120, 90))
    assert same_bits(gs, ro["samples"])


def test_several_mesh_nodes_park_and_resume(gpu, B, O, tmp_path):
    """Two mesh nodes (one nested under a transformed group, both refractive so rays also start inside them) between analytic
    objects: the render path parks a ray at the FIRST mesh whose root box it hits and k_trace_mesh resumes it there, runs
    the rest of the scene graph and may enter the second mesh inline; shadow rays are parked the same way.  Per-sample
    radiance, hits from all three sides and shadows must equal the oracle's bit for bit."""
    import shutil
    from conftest import SCENES
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    xml = tmp_path / "two_meshes.xml"
    xml.write_text("""<xml><scene>
      <background r="0.1" g="0.1" b="0.2"/><environment r="0.5" g="0.5" b="0.6"/>
      <object type="sphere" name="s0" material="red"><scale value="1.2"/><translate x="-5" y="1" z="1.2"/></object>
      <object type="obj" name="mesh_small.obj" material="glass"><scale value="2.5"/><translate x="-1.5" y="0" z="3"/></object>
      <object type="plane" name="ground" material="white"><scale value="25"/></object>
      <object name="grp"><rotate angle="35" z="1"/><translate x="3" y="2" z="0"/>
        <object type="obj" name="mesh_small.obj" material="blue"><scale x="2" y="1.5" z="2.2"/><rotate angle="20" x="1"/><translate z="2.6"/></object>
        <object type="sphere" name="s1" material="red"><scale value="0.8"/><translate x="2.5" z="0.8"/></object>
      </object>
      <material type="blinn" name="white"><diffuse value="0.8"/><specular value="0.1"/></material>
      <material type="blinn" name="red"><diffuse r="0.8" g="0.2" b="0.2"/><specular value="0.5"/><glossiness value="20"/></material>
      <material type="blinn" name="blue"><diffuse r="0.2" g="0.3" b="0.8"/><specular value="0.6"/><glossiness value="40"/></material>
      <material type="blinn" name="glass"><diffuse value="0.05"/><specular value="0.6"/><glossiness value="60"/>
        <refraction value="0.9" index="1.5"/><absorption r="0.02" g="0.05" b="0.02"/></material>
      <light type="ambient" name="a"><intensity value="0.1"/></light>
      <light type="point" name="p"><intensity value="250"/><position x="2" y="-8" z="16"/><size value="1.5"/></light>
      </scene><camera><position x="1" y="-22" z="9"/><target x="0" y="0" z="2.5"/><up z="1"/><fov value="35"/>
      <width value="200"/><height value="150"/></camera></xml>""")
    sc = B.Scene(str(xml))
    assert sc.info.n_meshes >= 1 and sum(1 for n in sc.flat_view().nodes if n.obj_type == 3) == 2
    blob = sc.flat_bytes()
    o, d = O.primary_rays(sc.flat_view())
    for side in (1, 2, 3):
        h = sc.trace_closest(o, d, side)
        r = O.trace_closest(blob, o, d, side)
        assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
    assert len(set(np.unique(r["node"])) & {1, 4}) == 2                    # both mesh nodes are hit by primary rays
    region = (20, 15, 180, 135)
    for spp, gi in ((3, 3), (2, 0)):
        opts = B.default_opts(spp=spp, gi_bounces=gi, seed=11)
        gs, st = sc.render_samples(opts, *region)
        ro = O.render(blob, sc.width, sc.height, spp, gi=gi, seed=11, region=region)
        assert np.nanmax(np.abs(gs - ro["samples"])) <= 1e-4
        assert same_bits(gs, ro["samples"])
    rgb, rad, st = sc.render(B.default_opts(spp=2, gi_bounces=2, seed=4))
    ro = O.render(blob, sc.width, sc.height, 2, gi=2, seed=4, want_samples=False)
    assert np.array_equal(rgb, ro["rgb8"]) and same_bits(rad, ro["radiance"])
    # the wavefront emits a frame's refraction and GI rays together; the recursion skips the GI ray when the refraction term
    # alone already reaches white (MtlBlinn.cpp:118-123): a handful of extra rays, same radiance
    assert 0 <= st.closest_rays - ro["stats"].closest_rays <= ro["stats"].closest_rays // 1000


def test_axis_parallel_rays_take_wave_steps_of_their_own(gpu, B, O, tmp_path):
    """Box::IntersectRay leaves out an axis whose direction component is zero (Box.cpp:13-28): a ray parallel to a coordinate
    axis of a mesh's space hits every box it passes on the other axes and walks most of the BVH, in the reference and here.  The render path
    sets such rays aside (one of them would hold up its whole wave step) and traces and shades them in wave steps of their own at the end of
    the pass; nothing may depend on that.  A camera straight above an unrotated mesh without jitter: every ray of the image's middle column
    and middle row has a zero component — camera rays, with their GI and shadow rays behind them.  Bit for bit against the oracle."""
    import shutil
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    xml = tmp_path / "above.xml"
    xml.write_text("""<xml><scene><background r="0.1" g="0.1" b="0.2"/><environment r="0.4" g="0.4" b="0.5"/>
      <object type="plane" name="floor" material="w"><scale value="20"/></object>
      <object type="obj" name="mesh_small.obj" material="g"><scale value="3"/><translate z="4"/></object>
      <object type="sphere" name="s" material="r"><scale value="1.5"/><translate x="6" y="2" z="1.5"/></object>
      <material type="blinn" name="w"><diffuse value="0.8"/><specular value="0.1"/></material>
      <material type="blinn" name="r"><diffuse r="0.8" g="0.2" b="0.2"/><specular value="0.4"/><glossiness value="20"/></material>
      <material type="blinn" name="g"><diffuse value="0.05"/><specular value="0.5"/><glossiness value="60"/><refraction value="0.9" index="1.5"/></material>
      <light type="point" name="p"><intensity value="300"/><position x="3" y="-4" z="18"/><size value="1"/></light></scene>
      <camera><position x="0" y="0" z="30"/><target x="0" y="0" z="0"/><up x="0" y="1" z="0"/><fov value="40"/><width value="64"/><height value="48"/></camera></xml>""")
    sc = B.Scene(str(xml))
    blob = sc.flat_bytes()
    o, d = O.primary_rays(sc.flat_view())
    assert ((d == 0).sum(axis=1) >= 1).sum() >= 64 + 48 - 1              # the premise: the middle column and row are axis-parallel rays
    for spp, gi in ((2, 2), (1, 0)):
        opts = B.default_opts(spp=spp, gi_bounces=gi, seed=6, jitter=0)
        gs, st = sc.render_samples(opts, 0, 0, sc.width, sc.height)
        ro = O.render(blob, sc.width, sc.height, spp, gi=gi, seed=6, jitter=0)
        assert same_bits(gs, ro["samples"])
        assert st.deferred_rays >= 30 * spp                              # those that enter the mesh's root box were set aside
        assert 0 <= st.closest_rays - ro["stats"].closest_rays <= ro["stats"].closest_rays // 1000   # see test_several_mesh_nodes_park_and_resume
    rgb, rad, st = sc.render(B.default_opts(spp=2, gi_bounces=2, seed=6, jitter=0))
    ro = O.render(blob, sc.width, sc.height, 2, gi=2, seed=6, jitter=0, want_samples=False)
    assert np.array_equal(rgb, ro["rgb8"]) and same_bits(rad, ro["radiance"]) and st.deferred_rays > 0


# ---------------------------------------------------------------------------------------------------- radiance
@pytest.mark.parametrize("case,spp,gi", [("c1_sphere_plane", 3, 3), ("c2_glass_small", 4, 3), ("c3_mesh_small", 3, 3),
                                         ("c4_textured", 3, 2), ("c2_glass_small", 2, 0), ("c2_glass_small", 2, -1),
                                         ("c3_room_small", 3, 3)])
def test_per_sample_radiance_vs_oracle(case, spp, gi, gpu, load_scene, O):
    sc = load_scene(case)
    W, H = sc.width, sc.height
    region = (W // 5, H // 5, W - W // 5, H - H // 5)
    opts = gpu.default_opts(spp=spp, gi_bounces=gi, seed=11)
    gs, st = sc.render_samples(opts, *region)
    ro = O.render(sc.flat_bytes(), W, H, spp, gi=gi, seed=11, region=region)
    assert np.nanmax(np.abs(gs - ro["samples"])) <= TOL
    assert same_bits(gs, ro["samples"])                      # stronger than the bar: identical bits


@pytest.mark.parametrize("case", ["c2_glass_small", "c4_textured"])
def test_full_frame_vs_oracle(case, gpu, load_scene, O):
    sc = load_scene(case)
    opts = gpu.default_opts(spp=3, gi_bounces=3)
    rgb, rad, st = sc.render(opts)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 3, gi=3, want_samples=False)
    assert np.abs(rad - ro["radiance"]).max() <= TOL
    assert same_bits(rad, ro["radiance"])
    assert np.array_equal(rgb, ro["rgb8"])                   # gamma + Color24 (Main.cpp:220-230)
    assert st.camera_samples == sc.width * sc.height * 3
    # the same rays and Shade() calls as the recursion (shadow rays may exceed the oracle's: the wavefront traces
    # the direct-light ray of a frame whose refraction/GI term later turns out >= 1, MtlBlinn.cpp:118,125)
    assert st.closest_rays == ro["stats"].closest_rays and st.shade_calls == ro["stats"].shade_calls
    assert ro["stats"].shadow_rays <= st.shadow_rays <= ro["stats"].shadow_rays * 1.01


def test_no_jitter_no_gamma_options(gpu, load_scene, O):
    sc = load_scene("c1_sphere_plane")
    opts = gpu.default_opts(spp=1, gi_bounces=-1, jitter=0, gamma=0)
    rgb, rad, _ = sc.render(opts)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 1, gi=-1, jitter=0, want_samples=False)
    assert same_bits(rad, ro["radiance"])
    q = np.clip((rad * 255 + np.float32(0.5)).astype(np.int32), 0, 255).astype(np.uint8)
    assert np.array_equal(rgb, q)


@pytest.mark.parametrize("name", ["c2_glass_small", "c3_mesh_small"])
def test_a_pass_that_overflows_its_frame_pool_is_cut_in_half_and_redone(gpu, load_scene, monkeypatch, name):
    """RenderRange's retry: k_shade flags a Shade() frame beyond the pool (cap_frames), the flag reaches the host with the step's counters
    (published by k_shade's last workgroup), the pass is redone with half the pixels — and the frame is the same, byte for byte, with
    the statistics of the passes that completed only.  (The mesh scene: the abandoned pass has any-hit kernels of its earlier steps on the second
    stream, and a step's any-hit rays still waiting for the next step's mesh walk.)"""
    sc = load_scene(name)
    opts = gpu.default_opts(spp=4, gi_bounces=3, seed=9)
    base_rgb, base_rad, st = sc.render(opts)
    assert st.passes == 1
    frames_needed = st.shade_calls if hasattr(st, "shade_calls") else st.camera_samples * 2
    monkeypatch.setenv("BHRT_TEST_FRAME_CAP", "1")  # no environment variable reaches the knob (a stray one must not send a user's render through retries)
    sc.knob("frame_cap", max(1, int(frames_needed) // 3))  # a third of what the one-pass frame needs
    rgb, rad, st2 = sc.render(opts)
    sc.knob("frame_cap", 0)
    monkeypatch.delenv("BHRT_TEST_FRAME_CAP")
    assert st2.passes >= 3
    assert np.array_equal(rgb, base_rgb) and same_bits(rad, base_rad)
    assert (st2.camera_samples, st2.closest_rays, st2.shadow_rays) == (st.camera_samples, st.closest_rays, st.shadow_rays)
    rgb3, rad3, st3 = sc.render(opts)  # the knob is gone, the remembered pass size of the scene/options is not a smaller frame pool
    assert np.array_equal(rgb3, base_rgb) and same_bits(rad3, base_rad)


# ---------------------------------------------------------------------------------------------------- full-size properties
def test_tile_partition_and_pass_size_invariance_full_size(gpu, load_scene):
    """BASELINE config 2 at full size (1920x1080): the image must not depend on how it is cut into passes or
    ranks — N logical ranks on one device reproduce the single-rank image byte for byte."""
    sc = load_scene("c2_glass")
    assert (sc.width, sc.height) == (1920, 1080)
    spp = 2
    base_rgb, base_rad, st = sc.render(gpu.default_opts(spp=spp))
    assert st.camera_samples == 1920 * 1080 * spp
    # determinism
    rgb2, rad2, _ = sc.render(gpu.default_opts(spp=spp))
    assert np.array_equal(base_rgb, rgb2) and same_bits(base_rad, rad2)
    # small passes
    rgb3, rad3, st3 = sc.render(gpu.default_opts(spp=spp, samples_per_pass=300000))
    assert st3.passes > 5 and same_bits(base_rad, rad3) and np.array_equal(base_rgb, rgb3)
    # 3 logical ranks, odd tile size
    import bhraytracer_amd.dist as BD
    import torch
    acc_rgb = np.zeros_like(base_rgb)
    acc_rad = np.zeros_like(base_rad)
    total = 0
    for r in range(3):
        rgb, rad, s = sc.render(gpu.default_opts(spp=spp, rank=r, world_size=3, tile_size=24))
        m = BD.owned_mask(1920, 1080, 24, r, 3).numpy()
        assert not rgb[~m].any() and not rad[~m].any()       # other ranks' pixels untouched
        acc_rgb[m] = rgb[m]
        acc_rad[m] = rad[m]
        total += s.camera_samples
    assert total == 1920 * 1080 * spp
    assert np.array_equal(acc_rgb, base_rgb) and same_bits(acc_rad, base_rad)


def test_photon_frame_full_size_invariances(gpu, load_scene):
    """BASELINE config 5 at full size (1920x1080, 1 M photons): build determinism, and a photon-mapped frame that does not
    depend on passes or ranks (the gather is tile-local; all three gather passes run: the focus under the glass sphere holds
    > 10^5 photons inside one gather radius)."""
    sc = load_scene("c5_caustics_hd")
    assert (sc.width, sc.height) == (1920, 1080)
    o = gpu.default_opts(spp=1, gi_bounces=1, seed=2)
    n = sc.photon_build(o, 1000000)
    assert n == 1000000
    first = sc.photon_get().copy()
    assert sc.photon_build(o, 1000000) == n and np.array_equal(sc.photon_get(), first)      # same map again, byte for byte
    # the gather at full size: around the focus under the glass sphere one radius holds up to 10^5 photons -> the selection pass with its
    # shrinking bound and repeated compactions; further out the plain walk.  Same photons as the oracle's LocatePhotons.
    from test_photon import check_selected_photons
    import oracle_lib as O
    O.photon_attach(first)
    rng = np.random.RandomState(21)
    p = np.concatenate([rng.uniform([-13, -11, 0.0], [-5, -2, 0.0], (260, 3)), rng.uniform([-15, -20, 0.0], [15, 10, 0.0], (140, 3))]).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, 1]), (len(p), 1))
    n_heavy, n_sel = check_selected_photons(sc, O, p, nrm, 0.5)
    assert n_heavy > 50 and n_sel > 50
    o.photon_map = 1
    base_rgb, base_rad, st = sc.render(o)
    off_rgb, off_rad, _ = sc.render(gpu.default_opts(spp=1, gi_bounces=1, seed=2))
    assert not same_bits(base_rad, off_rad)                                                # the caustic term is there
    o2 = gpu.default_opts(spp=1, gi_bounces=1, seed=2, samples_per_pass=500000, photon_map=1)
    rgb2, rad2, st2 = sc.render(o2)
    assert st2.passes > 3 and same_bits(base_rad, rad2) and np.array_equal(base_rgb, rgb2)
    import bhraytracer_amd.dist as BD
    acc_rad = np.zeros_like(base_rad)
    for r in range(2):
        rgb, rad, s = sc.render(gpu.default_opts(spp=1, gi_bounces=1, seed=2, rank=r, world_size=2, tile_size=32, photon_map=1))
        m = BD.owned_mask(1920, 1080, 32, r, 2).numpy()
        acc_rad[m] = rad[m]
    assert same_bits(acc_rad, base_rad)


def test_large_mesh_full_size_primary_and_radiance(gpu, B, O):
    """BASELINE config 3 geometry (100,352 triangles) at 1920x1080: every primary hit index / t bit-exact
    against the oracle, radiance on a region."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mesh
    mesh = os.path.join(SCENES, "gen", "mesh_224.obj")
    if not os.path.exists(mesh):
        gen_mesh.generate(mesh, 224)
    sc = B.Scene(os.path.join(SCENES, "c3_mesh.xml"))
    assert sc.info.n_triangles == 100352
    o, d = O.primary_rays(sc.flat_view())
    h = sc.trace_closest(o, d, 1)
    r = O.trace_closest(sc.flat_bytes(), o, d, 1)
    assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
    region = (860, 420, 1060, 560)
    gs, _ = sc.render_samples(B.default_opts(spp=2), *region)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 2, region=region)
    assert same_bits(gs, ro["samples"])
    # shadow rays from the hit points
    hit = r["node"] >= 0
    P = r["attrs"][hit][::7, 1:4]
    L = np.array(list(sc.flat_view().lights[-1].vec), np.float32)
    sd = (L[None] - P).astype(np.float32)
    assert np.array_equal(sc.trace_shadow(P, sd, 1.0), O.trace_shadow(sc.flat_bytes(), P, sd, 1.0))


@pytest.mark.gpu
def test_tile_exchange_pack_unpack(B):
    """bhrt_tiles_pack_dev / _unpack_dev (the multi-GPU framebuffer exchange, logical ranks on one GPU): the blocks of
    all ranks, concatenated the way all_gather_into_tensor delivers them, unpack to the full image; layout = dist.py's."""
    import torch
    from bhraytracer_amd import dist as BD
    for (W, H, tile, world) in ((100, 70, 32, 3), (1920, 1080, 32, 8), (64, 64, 16, 1), (50, 33, 7, 2)):
        g = torch.Generator().manual_seed(W + world)
        rad = torch.rand((H, W, 3), generator=g).cuda()
        rgb = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).cuda()
        bb = B.tiles_block_bytes(W, H, tile, world)
        allb = torch.zeros(bb * world, dtype=torch.uint8, device="cuda")
        for r in range(world):
            own = BD.owned_mask(W, H, tile, r, world).cuda().unsqueeze(-1)
            # a rank only has its own pixels; the others hold garbage that must never be read
            rad_r = torch.where(own, rad, torch.full_like(rad, float("nan")))
            rgb_r = torch.where(own, rgb, torch.full_like(rgb, 77))
            B.tiles_pack_dev(rgb_r.data_ptr(), rad_r.data_ptr(), W, H, tile, r, world, allb[r * bb:].data_ptr())
            px = BD.pack_tiles(rad_r, tile, r, world)
            torch.cuda.synchronize()
            mine = allb[r * bb:(r + 1) * bb]
            got = mine[: px.numel() * 4].view(torch.float32).view(px.shape)
            m = ~torch.isnan(px)
            assert torch.equal(got[m], px[m])           # same block layout as the torch reference implementation
        out_rad = torch.zeros_like(rad); out_rgb = torch.zeros_like(rgb)
        B.tiles_unpack_dev(allb.data_ptr(), W, H, tile, world, out_rgb.data_ptr(), out_rad.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out_rad, rad) and torch.equal(out_rgb, rgb)


@pytest.mark.gpu
def test_logical_ranks_exchange_matches_single_rank_render():
    """Two processes on this GPU (gloo, blocks through the host): render own tiles, native pack -> all_gather -> native
    unpack, compare with the single-rank frame byte for byte (tools/verify_multi_rank.py).  RCCL itself needs one GPU per rank."""
    import socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "verify_multi_rank.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("identical") == 4 + 2 and "MISMATCH" not in r.stdout   # 4 frame exchanges + the photon map built over 2 ranks (one line per rank)


@pytest.mark.gpu
def test_random_scenes_vs_oracle(B, O, tmp_path):
    """tools/fuzz_parity.py on fixed seeds: random scene graphs (nested groups, mesh instances, refraction, textures, all
    light types), hits and per-sample radiance bit for bit.  (400 further seeds were run by hand when this was added.)"""
    import shutil, sys
    from conftest import ROOT, SCENES
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    for seed in range(101, 117):
        ok, nbad, n_nodes, gi = fuzz_parity.check(seed, B, O, str(tmp_path))
        assert ok, (seed, nbad, n_nodes, gi)
