"""The CPU oracle against the golden vectors the compiled reference produced (tests/golden/make_golden.py).
Everything here is bit-exact: integer hit indices, float bits of z / p / N / uv, per-sample radiance
(sequential RNG + libm = the mode in which the reference itself was run), and the RGB8 bytes."""
import hashlib

import numpy as np
import pytest

from conftest import FULL_SIZE_CASES, GOLDEN_CASES, same_bits


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", GOLDEN_CASES + FULL_SIZE_CASES)
def test_primary_hit_table(case, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    o, d = O.primary_rays(sc.flat_view())
    r = O.trace_closest(sc.flat_bytes(), o, d, 1)
    H, W = int(g["height"]), int(g["width"])
    node = r["node"].reshape(H, W)
    assert sha(node.astype(np.int32)) == str(g["primary_node_sha"])
    assert sha(r["attrs"][:, 0].reshape(H, W)) == str(g["primary_z_sha"])
    assert sha(r["attrs"].reshape(H, W, 16)[..., 1:7][node >= 0]) == str(g["primary_pN_sha"])
    st = int(g["primary_step"])
    sub = (slice(None, None, st), slice(None, None, st))
    assert np.array_equal(node[sub], g["primary_node"])
    hit = g["primary_node"] >= 0
    assert np.array_equal(r["front"].reshape(H, W)[sub][hit], g["primary_front"][hit])
    a = r["attrs"].reshape(H, W, 16)[sub]
    assert same_bits(a[..., :9][hit], g["primary_attrs"][hit])      # z, p, N, u, v
    assert same_bits(a[..., 10:16][hit], g["primary_duvw"][hit])    # ray differentials (planes)


@pytest.mark.parametrize("case", GOLDEN_CASES + FULL_SIZE_CASES)
@pytest.mark.parametrize("side", [1, 2, 3])
def test_secondary_rays_all_hit_sides(case, side, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    r = O.trace_closest(sc.flat_bytes(), g["rays_o"], g["rays_d"], side)
    assert np.array_equal(r["node"], g[f"rays_node_{side}"])
    hit = r["node"] >= 0
    assert hit.sum() > 50
    assert same_bits(r["t"][hit], g[f"rays_z_{side}"][hit])
    assert np.array_equal(r["front"][hit], g[f"rays_front_{side}"][hit])


@pytest.mark.parametrize("case", GOLDEN_CASES + FULL_SIZE_CASES)
def test_shadow_rays(case, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    vis = O.trace_shadow(sc.flat_bytes(), g["shadow_o"], g["shadow_d"], 1.0)
    assert np.array_equal(vis.astype(np.int8), g["shadow_vis"])
    assert 0 < (vis == 0).sum() < len(vis)


@pytest.mark.parametrize("case", GOLDEN_CASES + FULL_SIZE_CASES)
def test_integrator_per_sample_radiance(case, load_scene, golden, O):
    g = golden(case)
    sc = load_scene(case)
    region = tuple(int(x) for x in g["render_region"])
    r = O.render(sc.flat_bytes(), sc.width, sc.height, int(g["render_spp"]), gi=int(g["render_gi"]),
                 rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM, region=region, threads=4)
    assert same_bits(r["samples"], g["render_samples"])
    assert same_bits(r["radiance"].reshape(-1, 3), g["render_radiance"])
    assert np.array_equal(r["rgb8"].reshape(-1, 3), g["render_rgb8"])


# ---------------------------------------------------------------------------------------------------- the whole program
def test_whole_program_begin_render_vs_the_reference_s_own(load_scene, golden, O):
    """oracle_begin_render against RenderImage::GetPixels() after the reference's OWN BeginRender() (Main.cpp:178-242; harness
    command `beginrender`, which restates nothing): one rand() stream for the process, pixel loop in the reference's order, 32 spp,
    GI depth 3.  The same number of rand() draws and the same bytes.  This is what ties the per-(pixel, sample) restatement used
    everywhere else (camera frame, RandomPositionInPixel, sample average, gamma, Color24) to the real function."""
    g = golden("begin_render")
    sc = load_scene(str(g["room_scene"]))
    rgb, _, _, draws = O.begin_render(sc.flat_bytes(), sc.width, sc.height)
    assert draws == int(g["room_draws"])
    assert np.array_equal(rgb, g["room_rgb8"])
    fv = sc.flat_view().header.camera
    cam = np.array(list(fv.top_left) + list(fv.dd_x) + list(fv.dd_y), np.float32)
    assert same_bits(cam, g["room_camera"][:9])                               # Main.cpp:179-192
    assert same_bits(np.float32(sc.flat_view().header.all_light_intensity), g["room_camera"][9])  # Main.cpp:116-123


def test_whole_program_photon_map_build_and_frame_vs_the_reference_s_own(load_scene, golden, O):
    """The -DUSE_PhotonMap build: the reference's own BuildCausticPhotonMap() (Main.cpp:342-386, 1,000,000 photons, its emission
    loop, ScalePhotonPowers, PrepareForIrradianceEstimation, the .dat it writes) followed by its own BeginRender() pixel loop with
    the k = 1000 gather in every Shade().  The oracle must arrive at the same 24 MB of photons, draw count and pixels."""
    import hashlib
    g = golden("begin_render")
    sc = load_scene(str(g["caustic_scene"]))
    n = int(g["caustic_photons"])
    rgb, ph, emitted, draws = O.begin_render(sc.flat_bytes(), sc.width, sc.height, photon_budget=n)
    assert len(ph) == n and draws == int(g["caustic_draws"]) and emitted > n
    half = n // 2 - 1
    ph[:, 19] &= np.where(np.arange(1, n + 1) < half, 0x0B, 0x08).astype(np.uint8)   # the bits the reference initialises
    assert np.array_equal(ph[::997], g["caustic_photons_every_997"])
    assert hashlib.sha256(np.ascontiguousarray(ph).tobytes()).hexdigest() == str(g["caustic_photons_sha"])
    assert np.array_equal(rgb, g["caustic_rgb8"])
