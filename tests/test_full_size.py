"""BASELINE.json's full sizes under -m gpu: config 4 (the mesh scene at 3840x2160, 256 spp over 8 ranks) and the closed-room
mesh workload SURVEY.md 8(d) asks for beside C3 (the Cornell room of Resource/Data/proj13.xml with the 100,352-triangle mesh
in the teapot's place).  Pins: tests/golden/c4_mesh_4k.npz and c3_room.npz — produced by the compiled reference
(tests/golden/make_golden.py) — for the hit tables, the oracle for triangle ids and keyed-RNG radiance, and the
size-independent properties of the path (partition / pass invariance).  Everything is bit-exact."""
import hashlib

import numpy as np
import pytest

from conftest import SCENES, same_bits

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star's per-channel radiance tolerance; the assertions below are stronger (identical bits)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gpu(B):
    if B.device_count() < 1:
        pytest.fail("no HIP device: the render path has no CPU fallback, GPU tests cannot run here")
    return B


def _primary_vs_golden_and_oracle(sc, g, O):
    o, d = O.primary_rays(sc.flat_view())
    h = sc.trace_closest(o, d, 1)
    H, W = int(g["height"]), int(g["width"])
    assert (sc.width, sc.height) == (W, H)
    # every pixel against the reference's own table (recursive(), Main.cpp:389): node index and the bits of z
    assert sha(h["node"].reshape(H, W).astype(np.int32)) == str(g["primary_node_sha"])
    assert sha(h["t"].reshape(H, W)) == str(g["primary_z_sha"])
    st = int(g["primary_step"])
    hit = g["primary_node"] >= 0
    assert np.array_equal(h["front"].reshape(H, W)[::st, ::st][hit], g["primary_front"][hit])
    r = O.trace_closest(sc.flat_bytes(), o, d, 1)  # triangle ids: the reference does not keep them, the oracle does
    assert np.array_equal(h["prim"], r["prim"]) and np.array_equal(h["node"], r["node"]) and same_bits(h["t"], r["t"])
    return r


# ---------------------------------------------------------------------------------------------------- BASELINE config 4
def test_c4_primary_hits_of_every_4k_pixel(gpu, load_scene, golden, O):
    sc = load_scene("c4_mesh_4k")
    assert (sc.width, sc.height) == (3840, 2160) and sc.info.n_triangles == 100352
    r = _primary_vs_golden_and_oracle(sc, golden("c4_mesh_4k"), O)
    assert (r["prim"] >= 0).sum() > 300000  # the mesh covers a good part of the frame


def test_c4_secondary_and_shadow_rays_vs_golden(gpu, load_scene, golden):
    g = golden("c4_mesh_4k")
    sc = load_scene("c4_mesh_4k")
    for side in (1, 2, 3):
        h = sc.trace_closest(g["rays_o"], g["rays_d"], side)
        assert np.array_equal(h["node"], g[f"rays_node_{side}"])
        hit = h["node"] >= 0
        assert same_bits(h["t"][hit], g[f"rays_z_{side}"][hit]) and np.array_equal(h["front"][hit], g[f"rays_front_{side}"][hit])
    assert np.array_equal(sc.trace_shadow(g["shadow_o"], g["shadow_d"], 1.0).astype(np.int8), g["shadow_vis"])


def test_c4_eight_logical_ranks_equal_the_single_rank_frame(gpu, load_scene):
    """3840x2160, 8 logical ranks with 32x32 interleaved tiles (tile t -> rank t mod 8): the union of the
    ranks' tiles is the single-rank frame byte for byte, each rank touches only its own pixels, and the frame does not
    depend on the number of passes (the single-rank frame is cut into >= 2 passes here as it is at 256 spp)."""
    import bhraytracer_amd.dist as BD
    sc = load_scene("c4_mesh_4k")
    W, H, spp = 3840, 2160, 2
    base_rgb, base_rad, st = sc.render(gpu.default_opts(spp=spp, gi_bounces=3, seed=7, samples_per_pass=6000000))
    assert st.passes >= 2 and st.camera_samples == W * H * spp
    one_rgb, one_rad, st1 = sc.render(gpu.default_opts(spp=spp, gi_bounces=3, seed=7))
    assert st1.passes == 1 and np.array_equal(one_rgb, base_rgb) and same_bits(one_rad, base_rad)
    del one_rgb, one_rad
    acc_rgb, acc_rad = np.zeros_like(base_rgb), np.zeros_like(base_rad)
    total = closest = 0
    for r in range(8):
        rgb, rad, s = sc.render(gpu.default_opts(spp=spp, gi_bounces=3, seed=7, rank=r, world_size=8, tile_size=32))
        m = BD.owned_mask(W, H, 32, r, 8).numpy()
        assert not rgb[~m].any() and not rad[~m].any()  # other ranks' pixels untouched
        acc_rgb[m], acc_rad[m] = rgb[m], rad[m]
        total += s.camera_samples
        closest += s.closest_rays
    assert total == W * H * spp and closest == st.closest_rays
    assert np.array_equal(acc_rgb, base_rgb) and same_bits(acc_rad, base_rad)


def test_c4_one_rank_at_the_real_256_spp(gpu, load_scene, O):
    """One of BASELINE config 4's eight ranks at the real sample count: 3840x2160, 256 spp, tile 32 -> 1,036,800 owned
    pixels x 256 = 2.65e8 camera samples in four passes of 2^26, so the slot -> (pixel, sample) -> tile arithmetic
    (kernels.hip::pixel_of, sample_addr, camera_ray) runs at its full magnitude.  Per-sample radiance and the resolved pixels
    of an owned tile in the LAST pass against the oracle at 256 spp."""
    sc = load_scene("c4_mesh_4k")
    rank, world, tile, spp = 5, 8, 32, 256
    tiles_x = 3840 // tile
    # a tile of rank 5 in tile row 60 of 68 (owned tile 900 of 1020: rendered in the last of the four passes), on the ground plane
    t = next(t for t in range(tiles_x * (2160 // tile + 1) - 1, 0, -1) if t % world == rank and t // tiles_x == 60 and 40 <= t % tiles_x <= 60)
    x0, y0 = (t % tiles_x) * tile, (t // tiles_x) * tile
    region = (x0, y0, x0 + 32, y0 + 16)
    opts = gpu.default_opts(spp=spp, gi_bounces=3, seed=0, rank=rank, world_size=world, tile_size=tile, samples_per_pass=1 << 26)
    gs, st = sc.render_samples(opts, *region)
    assert st.passes >= 4 and st.camera_samples == 1036800 * spp
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, spp, gi=3, seed=0, region=region, threads=16)
    assert np.nanmax(np.abs(gs - ro["samples"])) <= TOL
    assert same_bits(gs, ro["samples"])
    # the in-order sample average of Main.cpp:150-170 over 256 samples
    acc = np.zeros((gs.shape[0], 3), np.float32)
    for s in range(spp):
        acc += gs[:, s, :]
    assert same_bits(acc / np.float32(spp), ro["radiance"].reshape(-1, 3))
    # a region that belongs to another rank stays empty
    gs2, _ = sc.render_samples(gpu.default_opts(spp=1, gi_bounces=0, rank=(rank + 1) % world, world_size=world, tile_size=tile), *region)
    assert not gs2.any()


def test_c4_frame_larger_than_one_pass_sizes_its_frame_pool_from_the_first_frame(gpu):
    """BASELINE config 4 per GPU: 3840x2160 x 32 spp = 2.65e8 sample slots do not fit into one pass with six Shade() frames per slot; the first
    frame runs in two passes and notes what they needed (1.1 frames per slot), the second provides that + 30 % and — memory permitting: other
    scenes of this test session hold workspaces too — runs in one pass.  Same frame, byte for byte."""
    import os
    sc = gpu.Scene(os.path.join(SCENES, "c4_mesh_4k.xml"))
    try:
        opts = gpu.default_opts(spp=32, gi_bounces=3, seed=2)
        rgb1, rad1, st1 = sc.render(opts)
        assert st1.camera_samples == 3840 * 2160 * 32 and st1.passes >= 2
        rgb2, rad2, st2 = sc.render(opts)
        assert st2.passes <= st1.passes
        assert np.array_equal(rgb1, rgb2) and same_bits(rad1, rad2)
        assert (st2.closest_rays, st2.shadow_rays, st2.shade_calls) == (st1.closest_rays, st1.shadow_rays, st1.shade_calls)
    finally:
        sc.close()


# ---------------------------------------------------------------------------------------------------- closed-room mesh workload
def test_c3_room_primary_hits_vs_reference(gpu, load_scene, golden, O):
    sc = load_scene("c3_room")
    assert (sc.width, sc.height) == (1920, 1080) and sc.info.n_triangles == 100352
    r = _primary_vs_golden_and_oracle(sc, golden("c3_room"), O)
    assert (r["node"] >= 0).all()  # a closed room: every camera ray hits something
    assert (r["prim"] >= 0).sum() > 50000


def test_c3_room_secondary_and_shadow_rays_vs_golden(gpu, load_scene, golden, O):
    g = golden("c3_room")
    sc = load_scene("c3_room")
    for side in (1, 2, 3):
        h = sc.trace_closest(g["rays_o"], g["rays_d"], side)
        assert np.array_equal(h["node"], g[f"rays_node_{side}"])
        hit = h["node"] >= 0
        assert same_bits(h["t"][hit], g[f"rays_z_{side}"][hit]) and np.array_equal(h["front"][hit], g[f"rays_front_{side}"][hit])
    assert np.array_equal(sc.trace_shadow(g["shadow_o"], g["shadow_d"], 1.0).astype(np.int8), g["shadow_vis"])
    # rays that start INSIDE the glass mesh (the refraction chain's HIT_FRONT_AND_BACK walks, MtlBlinn.cpp:476-519)
    rng = np.random.RandomState(12)
    o = (np.array([2, 5, 5.9], np.float32) + rng.normal(size=(20000, 3)) * 1.5).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    for side in (2, 3):
        h, r = sc.trace_closest(o, d, side), O.trace_closest(sc.flat_bytes(), o, d, side)
        assert np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and same_bits(h["t"], r["t"])
        assert np.array_equal(h["front"][r["node"] >= 0], r["front"][r["node"] >= 0])


def test_c3_room_radiance_regions_vs_oracle(gpu, load_scene, O):
    """Per-sample radiance through GI depth 3 and 16 internal bounces inside the glass mesh: a region on the mesh, one on
    the floor under it (GI rays into the mesh from every direction), one on the mirror sphere."""
    sc = load_scene("c3_room")
    blob = sc.flat_bytes()
    for region, spp in (((980, 500, 1028, 532), 3), ((900, 800, 948, 816), 2), ((1180, 700, 1212, 716), 2)):
        gs, st = sc.render_samples(gpu.default_opts(spp=spp, gi_bounces=3, seed=5), *region)
        ro = O.render(blob, sc.width, sc.height, spp, gi=3, seed=5, region=region, threads=16)
        assert np.nanmax(np.abs(gs - ro["samples"])) <= TOL
        assert same_bits(gs, ro["samples"])
    assert st.closest_rays / st.camera_samples > 4.0  # the closed room's ray count per camera sample (~5.5 + ~4.9 shadow)


def test_c3_room_partition_and_pass_invariance(gpu, load_scene):
    import bhraytracer_amd.dist as BD
    sc = load_scene("c3_room")
    base_rgb, base_rad, st = sc.render(gpu.default_opts(spp=1, gi_bounces=3, seed=3))
    rgb2, rad2, st2 = sc.render(gpu.default_opts(spp=1, gi_bounces=3, seed=3, samples_per_pass=400000))
    assert st2.passes > 3 and np.array_equal(rgb2, base_rgb) and same_bits(rad2, base_rad)
    acc = np.zeros_like(base_rad)
    for r in range(4):
        _, rad, _ = sc.render(gpu.default_opts(spp=1, gi_bounces=3, seed=3, rank=r, world_size=4, tile_size=32))
        m = BD.owned_mask(1920, 1080, 32, r, 4).numpy()
        acc[m] = rad[m]
    assert same_bits(acc, base_rad)


# ---------------------------------------------------------------------------------------------------- whole frames
@pytest.mark.parametrize("name,spp", [("c2_glass", 2), ("c3_mesh", 2), ("c3_room", 2), ("c4_mesh_4k", 1)])
def test_every_sample_of_a_whole_frame_vs_oracle(gpu, load_scene, O, name, spp):
    """Not a region: EVERY sample of the full-size frame (2.07 M / 8.29 M pixels, GI depth 3, 16 internal bounces) against the oracle's
    keyed-RNG run of the same frame — 0.6-3 s of oracle time on the box's 16 cores.  Identical bits."""
    sc = load_scene(name)
    gs, st = sc.render_samples(gpu.default_opts(spp=spp, gi_bounces=3, seed=77), 0, 0, sc.width, sc.height)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, spp, gi=3, seed=77, region=(0, 0, sc.width, sc.height), threads=16)
    assert gs.shape[0] == sc.width * sc.height and st.camera_samples == sc.width * sc.height * spp
    assert same_bits(gs, ro["samples"])
    assert same_bits(sc.render(gpu.default_opts(spp=spp, gi_bounces=3, seed=77))[1].reshape(-1, 3), ro["radiance"].reshape(-1, 3))


def test_every_sample_of_a_whole_caustic_frame_vs_oracle(gpu, load_scene, O):
    """BASELINE config 5's scene at 1920x1080 with a 200 k-photon caustic map (keyed emission: the same map on both sides), every sample of
    the frame: the exact replay gives the oracle's bits, the one-wave-per-query selection stays within north_star's 1e-4 (measured 5e-6)."""
    import os
    sc = gpu.Scene(os.path.join(SCENES, "c5_caustics_hd.xml"))  # a scene of its own: the photon map stays with it
    n_photons = 200000
    sc.photon_build(gpu.default_opts(seed=3), n_photons)
    bal, _, _ = O.photon_build(sc.flat_bytes(), n_photons, seed=3)
    assert np.array_equal(sc.photon_get(), bal)
    region = (0, 0, sc.width, sc.height)
    opts = gpu.default_opts(spp=1, gi_bounces=2, seed=3, photon_map=1)
    opts.photon_exact = 1
    gx, stx = sc.render_samples(opts, *region)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 1, gi=2, seed=3, region=region, photon=1, threads=16)
    assert same_bits(gx, ro["samples"])
    assert stx.photon_heavy_queries > 5000 and stx.photon_exact_queries >= stx.photon_heavy_queries
    opts.photon_exact = 0
    gs, st = sc.render_samples(opts, *region)
    assert np.nanmax(np.abs(gs - ro["samples"])) <= TOL
    assert st.photon_heavy_queries == stx.photon_heavy_queries and st.photon_exact_queries <= st.photon_heavy_queries // 20


def test_config5_as_it_stands_one_million_photons_whole_frame_vs_oracle(gpu, O):
    """BASELINE config 5 at its own size (what tools/soak_parity.py caustics runs by hand): 10^6 stored caustic photons — the same map byte
    for byte as the oracle's BuildCausticPhotonMap (Main.cpp:342-386; ~6.4e7 emissions) — and every sample of the 1920x1080 frame at 2 spp,
    GI depth 3, against the oracle's render with its own k-NN gather (cyPhotonMap.h:332-382,421-498): exact replay = identical bits, the
    default selection pass within north_star's 1e-4 (measured 1.1e-5).  About a minute, most of it the oracle on the box's 16 cores."""
    import os
    sc = gpu.Scene(os.path.join(SCENES, "c5_caustics_hd.xml"))
    n_photons = 1000000
    sc.photon_build(gpu.default_opts(seed=0), n_photons)
    bal, _, n_emit = O.photon_build(sc.flat_bytes(), n_photons, seed=0)
    assert bal.shape[0] == n_photons and n_emit > 10 * n_photons
    assert np.array_equal(sc.photon_get(), bal)
    region = (0, 0, sc.width, sc.height)
    assert (sc.width, sc.height) == (1920, 1080)
    opts = gpu.default_opts(spp=2, gi_bounces=3, seed=1, photon_map=1)
    opts.photon_exact = 1
    gx, stx = sc.render_samples(opts, *region)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 2, gi=3, seed=1, region=region, photon=1, threads=16)["samples"]
    assert same_bits(gx, ro)
    assert stx.photon_heavy_queries > 50000 and stx.photon_exact_queries >= stx.photon_heavy_queries
    opts.photon_exact = 0
    gs, st = sc.render_samples(opts, *region)
    assert np.nanmax(np.abs(gs - ro)) <= TOL
    assert st.photon_heavy_queries == stx.photon_heavy_queries


@pytest.mark.parametrize("name,spp", [("c3_mesh", 2), ("c3_room", 2)])
def test_leaf_skip_gives_the_same_frames(gpu, load_scene, O, name, spp):
    """bhrt_opts.leaf_skip = 1: the mesh walks leave out a box-missed LEAF sibling (TriObj.cpp:245-248,263-266,286-300) when the host's per-leaf
    bounds prove that IntersectTriangle (TriObj.cpp:68-189) cannot accept any of its triangles for this ray (scene_host.cpp::ComputeLeafSkip).
    ~70 % of those visits go (250 M per closed-room frame at 64 spp); every sample of the whole frame keeps the oracle's bits."""
    sc = load_scene(name)
    m = sc.flat_view().meshes[0]
    assert m.skip_omax > 0 and 0 < m.skip_k0 < 1e-3 and m.skip_big > 10 * m.skip_k0
    opts = gpu.default_opts(spp=spp, gi_bounces=3, seed=78)
    opts.leaf_skip = 1
    gs, st = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, spp, gi=3, seed=78, region=(0, 0, sc.width, sc.height), threads=16)
    assert same_bits(gs, ro["samples"])
    opts.leaf_skip = 0
    g0, st0 = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    assert same_bits(g0, gs) and st0.closest_rays == st.closest_rays and st0.shadow_rays == st.shadow_rays


@pytest.mark.parametrize("name,spp", [("c3_mesh", 2), ("c3_room", 2), ("c4_textured", 4)])
def test_any_hit_work_beside_the_pass_gives_the_same_frames(gpu, load_scene, name, spp):
    """The any-hit kernels of a wave step (GenLight::Shadow, GenLight.cpp:10-69) run on a second stream beside the next step's closest-hit kernels, with a
    shadow queue per step parity; their visibilities are only read when the frames are folded at the end of the pass.  Knob "shadow_overlap" = 0 puts them
    back in front of the next step on the pass's own stream: every sample of the whole frame has the same bits either way, and the same ray counts.  (The
    default, overlapped path is the one every other whole-frame test compares with the oracle.)"""
    sc = load_scene(name)
    opts = gpu.default_opts(spp=spp, gi_bounces=3, seed=79)
    g1, st1 = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    sc.knob("shadow_overlap", 0)
    g0, st0 = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    sc.knob("shadow_overlap", 1)
    g2, st2 = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    assert same_bits(g0, g1) and same_bits(g2, g1)
    assert st0.closest_rays == st1.closest_rays == st2.closest_rays and st0.shadow_rays == st1.shadow_rays == st2.shadow_rays and st1.shadow_rays > 0
