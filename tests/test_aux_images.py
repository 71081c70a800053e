"""Images beside the colour image: RenderImage's z-buffer and ComputeZBufferImage (scene.h:532,578-600;
the z store is the commented-out Main.cpp:231), the first-hit normal / albedo images DenoiseImage could be given
(Main.cpp:70-71) and its "color" input, colorArray (Main.cpp:202,219-229).

CPU: the oracle against tests/golden/aux_images.npz, produced by the reference's own recursive(), ComputeZBufferImage,
TextureMap sampling and pow().  GPU: bhrt_first_hit / bhrt_zbuffer_image_dev / bhrt_color_image_dev against the oracle
(device-math mode), every bit."""
import hashlib

import numpy as np
import pytest

from conftest import same_bits

SCENES4 = ("c1_sphere_plane", "c2_glass_small", "c3_mesh_small", "c4_textured")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", SCENES4)
def test_oracle_aux_images_vs_reference_golden(name, load_scene, golden, O):
    g = golden("aux_images")
    sc = load_scene(name)
    z, nrm, alb = O.first_hit(sc.flat_bytes(), sc.width, sc.height, math=O.MATH_LIBM)
    assert same_bits(z[::7], g[name + "_z_every7"]) and same_bits(nrm[::7], g[name + "_normal_every7"]) and same_bits(alb[::7], g[name + "_albedo_every7"])
    assert sha(z) == str(g[name + "_z_sha"]) and sha(nrm) == str(g[name + "_normal_sha"]) and sha(alb) == str(g[name + "_albedo_sha"])
    assert np.array_equal(O.zbuffer_image(z), g[name + "_zimg"])           # the reference's ComputeZBufferImage
    assert same_bits(O.color_image(g[name + "_radiance"], 1, O.MATH_LIBM), g[name + "_color"])  # pow(c, 1/2.2f) kept as floats
    if name == "c4_textured":
        assert len(np.unique(alb[z < 1e30], axis=0)) > 50                    # texture lookups, not constants


def test_zbuffer_image_degenerate_ranges(O):
    big = np.float32(1e30)
    assert np.array_equal(O.zbuffer_image(np.full(7, big, np.float32)), np.zeros(7, np.uint8))            # nothing hit
    one = np.array([big, 3.5, big], np.float32)                                                          # zmax == zmin: 0 / 0
    assert np.array_equal(O.zbuffer_image(one), np.zeros(3, np.uint8))
    two = np.array([2.0, 4.0, 3.0, big], np.float32)
    assert list(O.zbuffer_image(two)) == [255, 0, 127, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES4 + ("c2_glass",))
def test_gpu_aux_images_vs_oracle(name, B, load_scene, O):
    import torch
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    sc = load_scene(name)
    W, H = sc.width, sc.height
    oz, on, oa = O.first_hit(sc.flat_bytes(), W, H)
    z, nrm, alb = sc.first_hit()
    assert same_bits(z.reshape(-1), oz) and same_bits(nrm.reshape(-1, 3), on) and same_bits(alb.reshape(-1, 3), oa)
    assert (oz < 1e30).sum() > W * H // 10
    dev = torch.device("cuda", 0)
    dz, dn = torch.zeros(H * W, dtype=torch.float32, device=dev), torch.zeros((H * W, 3), dtype=torch.float32, device=dev)
    sc.first_hit_dev(dz.data_ptr(), dn.data_ptr(), 0)                       # device outputs, albedo not wanted
    torch.cuda.synchronize()
    assert same_bits(dz.cpu().numpy(), oz) and same_bits(dn.cpu().numpy(), on)
    img = torch.zeros(H * W, dtype=torch.uint8, device=dev)
    sc.zbuffer_image_dev(dz.data_ptr(), H * W, img.data_ptr())
    assert np.array_equal(img.cpu().numpy(), O.zbuffer_image(oz))
    for zs in (np.full(5, 1e30, np.float32), np.array([1e30, 3.5, 1e30], np.float32), np.array([2.0, 4.0, 3.0, 1e30], np.float32)):
        t, out = torch.from_numpy(zs).to(dev), torch.full((len(zs),), 9, dtype=torch.uint8, device=dev)
        sc.zbuffer_image_dev(t.data_ptr(), len(zs), out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), O.zbuffer_image(zs))
    # colorArray from a rendered radiance image
    rgb, rad = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev), torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    sc.render_dev(B.default_opts(spp=1, gi_bounces=1, seed=2), rgb.data_ptr(), rad.data_ptr())
    col = torch.zeros_like(rad)
    for gamma in (1, 0):
        sc.color_image_dev(rad.data_ptr(), H * W, gamma, col.data_ptr())
        torch.cuda.synchronize()
        assert same_bits(col.cpu().numpy().reshape(-1), O.color_image(rad.cpu().numpy().reshape(-1), gamma))
    # consistent with the bytes the render stored: Color24(colorArray) == rgb8
    sc.color_image_dev(rad.data_ptr(), H * W, 1, col.data_ptr())
    c = col.cpu().numpy()
    q = np.clip((c * np.float32(255) + np.float32(0.5)).astype(np.int64), 0, 255).astype(np.uint8)
    assert np.array_equal(q, rgb.cpu().numpy())
