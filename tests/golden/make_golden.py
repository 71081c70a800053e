#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/ref_harness = the unmodified
reference compiled in place from /root/reference by oracle/Makefile).  Runs only in the development
container; the resulting fixtures are data (inputs + expected outputs) and are committed.

The reference has no tests or golden vectors of its own, so these are the pins:
  <scene>_primary   hit table of the un-jittered camera rays (Main.cpp:145,153 -> recursive(), Main.cpp:389):
                    node index, front flag, and the bits of z / p / N / uv for every 4th (8th for wide images) pixel,
                    plus SHA-256 digests of the complete tables
  <scene>_rays      seeded pseudo-random secondary rays with hitSide FRONT / BACK / FRONT_AND_BACK
  <scene>_shadow    GenLight::Shadow (GenLight.cpp:10) for rays from primary hit points towards the light
  <scene>_render    per-sample radiance of MtlBlinn::Shade (MtlBlinn.cpp:89) over a pixel region, rand()
                    interposed by the sequential stream of include/bhrt_rng.h, plus the gamma/Color24 bytes
  c5_caustics_photon  caustic photon map: emission, balance, gathers, radiance with the caustic term
  global_photon       the global photon map (BuildPhotonMap, Main.cpp:251-317) of several scenes

  aux_images          RenderImage's z-buffer + ComputeZBufferImage, first-hit normal / albedo, colorArray (post-gamma floats)

  begin_render        the reference's OWN BeginRender() as a whole program (ref_harness `beginrender`: nothing restated), 32 spp, one
                      rand() stream for the process: RenderImage::GetPixels() of a 64x48 frame, and — the -DUSE_PhotonMap build — of a
                      32x24 frame rendered after its own BuildCausticPhotonMap() (1,000,000 photons; digest of causticPhotonMap.dat)

`make_golden.py global_photon` / `make_golden.py aux_images` / `make_golden.py begin_render` regenerate only that file; `make_golden.py case <name>...` only
the named scene cases.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SCENES = os.path.join(ROOT, "tests", "scenes")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
HARNESS_PM = os.path.join(ROOT, "oracle", "_ref", "ref_harness_pm")  # reference built with -DUSE_PhotonMap

# scene file, render region, spp, gi
CASES = {
    "c1_sphere_plane": ("c1_sphere_plane.xml", (270, 180, 334, 228), 4, 3),
    "c2_glass_small": ("c2_glass_small.xml", (120, 110, 184, 158), 4, 3),
    "c3_mesh_small": ("c3_mesh_small.xml", (110, 70, 174, 118), 4, 3),
    "c4_textured": ("c4_textured.xml", (100, 90, 164, 138), 4, 2),
    # the Cornell room of proj13.xml with a glass mesh in the teapot's place (SURVEY.md 8d, C3 closed room): small, committed mesh
    "c3_room_small": ("c3_room_small.xml", (136, 96, 200, 144), 4, 3),
    # ... and BASELINE's full sizes (the 100,352-triangle mesh of tools/gen_mesh.py, generated on demand under tests/scenes/gen/):
    "c3_room": ("c3_room.xml", (980, 500, 1028, 532), 2, 3),          # 1920x1080 closed room
    "c4_mesh_4k": ("c4_mesh_4k.xml", (1700, 900, 1748, 932), 2, 3),   # BASELINE config 4: 3840x2160
}
FULL_SIZE = ("c3_room", "c4_mesh_4k")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(scene, prefix, *args):
    subprocess.run([HARNESS, scene, prefix, *map(str, args)], cwd=SCENES, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)


def secondary_rays(seed, n, lo, hi):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return o, d


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the development container")
    tmp = tempfile.mkdtemp(prefix="bhrt_golden_")
    if sys.argv[1:] in (["global_photon"], ["aux_images"], ["begin_render"]):
        {"global_photon": global_photon_case, "aux_images": aux_case, "begin_render": begin_render_case}[sys.argv[1]](tmp)
        subprocess.run(["rm", "-rf", tmp])
        return
    only = sys.argv[2:] if sys.argv[1:2] == ["case"] else None
    for name, (xml, region, spp, gi) in CASES.items():
        if only is not None and name not in only:
            continue
        if name in FULL_SIZE:
            mesh = os.path.join(SCENES, "gen", "mesh_224.obj")
            if not os.path.exists(mesh):
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import gen_mesh
                gen_mesh.generate(mesh, 224)
        pre = os.path.join(tmp, name)
        run(xml, pre, "dump", "primary")
        cam = np.fromfile(pre + ".camera_f32", np.float32)
        W, H = int(cam[11]), int(cam[12])
        pi = np.fromfile(pre + ".primary_i32", np.int32).reshape(H, W, 2)
        pf = np.fromfile(pre + ".primary_f32", np.float32).reshape(H, W, 16)
        step = 16 if W > 1000 else 8 if W > 400 else 4
        sub = (slice(None, None, step), slice(None, None, step))
        out = {
            "width": W, "height": H, "primary_step": step,
            "primary_node_sha": sha(pi[..., 0]), "primary_z_sha": sha(pf[..., 0]),
            "primary_pN_sha": sha(pf[..., 1:7][pi[..., 0] >= 0]),
            "primary_node": pi[..., 0][sub].astype(np.int16), "primary_front": pi[..., 1][sub].astype(np.int8),
            "primary_attrs": pf[sub][..., :9].copy(),  # z, p, N, u, v (uvw.z of a sphere hit is uninitialised in the reference)
            "primary_duvw": pf[sub][..., 10:16].copy(),
        }
        # secondary rays, all three hit sides
        o, d = secondary_rays(1234, 3000, -12.0, 12.0)
        o[:, 2] = np.abs(o[:, 2]) * 0.8 + 0.05
        for side in (1, 2, 3):
            rays = np.concatenate([o, d, np.full((len(o), 1), side, np.float32)], axis=1).astype(np.float32)
            rays.tofile(pre + ".rays_in")
            run(xml, pre, "--rays", pre + ".rays_in", "rays")
            ri = np.fromfile(pre + ".rays_i32", np.int32).reshape(-1, 2)
            rf = np.fromfile(pre + ".rays_f32", np.float32).reshape(-1, 16)
            out[f"rays_node_{side}"] = ri[:, 0].astype(np.int16)
            out[f"rays_front_{side}"] = ri[:, 1].astype(np.int8)
            out[f"rays_z_{side}"] = rf[:, 0].copy()
        out["rays_o"], out["rays_d"] = o, d
        # shadow rays from every 4th primary hit towards the brightest point light
        lights = np.fromfile(pre + ".lights_f32", np.float32)[:-1].reshape(-1, 8)
        pl = lights[lights[:, 0] == 2]
        hitmask = out["primary_node"] >= 0
        P = out["primary_attrs"][..., 1:4][hitmask]
        if len(pl) and len(P):
            L = pl[-1, 4:7]
            sd = (L[None, :] - P).astype(np.float32)
            sr = np.concatenate([P, sd, np.ones((len(P), 1), np.float32)], axis=1).astype(np.float32)
            sr.tofile(pre + ".shadow_in")
            run(xml, pre, "--shadow", pre + ".shadow_in", "shadow")
            out["shadow_o"], out["shadow_d"] = P.astype(np.float32), sd
            out["shadow_vis"] = np.fromfile(pre + ".shadow_f32", np.float32).astype(np.int8)
        # integrator
        x0, y0, x1, y1 = region
        run(xml, pre, "--spp", spp, "--gi", gi, "--region", x0, y0, x1, y1, "render")
        npx = (x1 - x0) * (y1 - y0)
        out["render_region"] = np.array(region, np.int32)
        out["render_spp"], out["render_gi"] = spp, gi
        out["render_samples"] = np.fromfile(pre + ".samples_f32", np.float32).reshape(npx, spp, 3)
        out["render_radiance"] = np.fromfile(pre + ".radiance_f32", np.float32).reshape(npx, 3)
        out["render_rgb8"] = np.fromfile(pre + ".rgb8", np.uint8).reshape(npx, 3)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {int((pi[..., 0] >= 0).sum())}/{W * H} primary hits")
    if only is None:
        photon_case(tmp)
        global_photon_case(tmp)
        aux_case(tmp)
        begin_render_case(tmp)
    subprocess.run(["rm", "-rf", tmp])


def photon_case(tmp):
    """Caustic photon map (Main.cpp:342-386, cyPhotonMap.h): the -DUSE_PhotonMap build of the reference emits, balances,
    gathers and renders; rand() interposed by ONE sequential stream for the whole emission loop."""
    name, xml, n_photons, region, spp, gi = "c5_caustics_photon", "c5_caustics.xml", 6000, (110, 160, 158, 196), 2, 2
    pre = os.path.join(tmp, name)
    rng = np.random.RandomState(5)
    q = np.concatenate([rng.uniform([-14, -20, 0.0], [14, 10, 0.0], (150, 3)), np.tile([0, 0, 1.0], (150, 1))], 1)
    q2 = np.concatenate([rng.uniform([-11, -9, 0.0], [-5, -3, 0.0], (250, 3)), np.tile([0, 0, 1.0], (250, 1))], 1)  # under the glass sphere
    q3 = np.concatenate([rng.uniform([-9, -7, 0.0], [-7, -5, 0.0], (100, 3)), np.tile([0, 0, -1.0], (100, 1))], 1)  # wrong-side normals
    q = np.concatenate([q, q2, q3]).astype(np.float32)
    q.tofile(pre + ".q")
    subprocess.run([HARNESS_PM, os.path.join(SCENES, xml), pre, "--photons", str(n_photons), "--gather", pre + ".q", "--spp", str(spp), "--gi", str(gi),
                    "--region", *map(str, region), "photons", "gather", "render"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    meta = np.fromfile(pre + ".photons_meta", np.uint64)
    npx = (region[2] - region[0]) * (region[3] - region[1])
    out = {
        "n_photons": n_photons, "stored": int(meta[0]), "emitted": int(meta[1]), "draws": int(meta[2]), "half": int(np.int64(meta[3])),
        # byte 19 (planeAndDirZ) has uninitialised low bits in the reference: bit 3 (sign of dirZ) always valid,
        # bits 0-1 (split plane) valid for internal nodes (index < half)
        "photons_emitted": np.fromfile(pre + ".photons_emitted", np.uint8).reshape(-1, 24),
        "photons_balanced": np.fromfile(pre + ".photons_balanced", np.uint8).reshape(-1, 24),
        "gather_q": q, "gather_out": np.fromfile(pre + ".gather_f32", np.float32).reshape(-1, 6),
        "render_region": np.array(region, np.int32), "render_spp": spp, "render_gi": gi,
        "render_samples": np.fromfile(pre + ".samples_f32", np.float32).reshape(npx, spp, 3),
    }
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {out['stored']} photons from {out['emitted']} emissions, "
          f"{int((out['gather_out'][:, :3].sum(1) > 0).sum())}/{len(q)} queries lit")


def aux_case(tmp):
    """Images beside the colour image, from the reference's own code where it has any: RenderImage::ComputeZBufferImage run
    on the z-buffer the commented-out store of Main.cpp:231 would fill; HitInfo::N and MtlBlinn's diffuse.Sample(uvw, duvw) of
    the first hit (the optional DenoiseImage inputs, Main.cpp:70-71); colorArray (Main.cpp:202,229) of the render region."""
    out = {}
    for name in ("c1_sphere_plane", "c2_glass_small", "c3_mesh_small", "c4_textured"):
        xml, region, spp, gi = CASES[name]
        pre = os.path.join(tmp, "aux_" + name)
        x0, y0, x1, y1 = region
        run(xml, pre, "--spp", spp, "--gi", gi, "--region", x0, y0, x1, y1, "aux", "render")
        z = np.fromfile(pre + ".aux_z_f32", np.float32)
        nrm = np.fromfile(pre + ".aux_normal_f32", np.float32).reshape(-1, 3)
        alb = np.fromfile(pre + ".aux_albedo_f32", np.float32).reshape(-1, 3)
        out[name + "_z_sha"], out[name + "_normal_sha"], out[name + "_albedo_sha"] = sha(z), sha(nrm), sha(alb)
        out[name + "_z_every7"], out[name + "_normal_every7"], out[name + "_albedo_every7"] = z[::7], nrm[::7], alb[::7]
        out[name + "_zimg"] = np.fromfile(pre + ".aux_zimg_u8", np.uint8)
        npx = (x1 - x0) * (y1 - y0)
        out[name + "_color"] = np.fromfile(pre + ".color_f32", np.float32).reshape(npx, 3)
        out[name + "_radiance"] = np.fromfile(pre + ".radiance_f32", np.float32).reshape(npx, 3)
        print(f"aux_images/{name}: {int((z < 1e30).sum())}/{len(z)} pixels hit, z image range {out[name + '_zimg'].min()}..{out[name + '_zimg'].max()}")
    path = os.path.join(HERE, "aux_images.npz")
    np.savez_compressed(path, **out)
    print(f"aux_images: {os.path.getsize(path) / 1024:.0f} KiB")


def begin_render_case(tmp):
    """The reference's own BeginRender() (Main.cpp:178-242) — camera frame, CalculateLightsIntensity, the OpenMP pixel loop with one
    thread (columns outside, rows inside), PathTracing at the compiled-in 32 spp / GI depth 3, gamma, Color24 — and, in the
    -DUSE_PhotonMap build, its own BuildCausticPhotonMap() (Main.cpp:342-386: always 1,000,000 photons, written to
    Resource/causticPhotonMap.dat under the working directory) in front of it.  rand() is interposed by ONE stream that is never
    reset, like libc's.  The oracle's whole-program mode (oracle_begin_render) has to reproduce the pixels byte for byte."""
    import shutil
    work = os.path.join(tmp, "begin")
    os.makedirs(os.path.join(work, "Resource", "Result"))
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), work)
    out = {}
    for name, harness, xml in (("room", HARNESS, "begin_room.xml"), ("caustic", HARNESS_PM, "begin_caustic.xml")):
        pre = os.path.join(work, name)
        subprocess.run([harness, os.path.join(SCENES, xml), pre, "beginrender"], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        meta = np.fromfile(pre + ".begin_meta", np.uint64)
        W, H = int(meta[1]), int(meta[2])
        out[name + "_scene"] = xml
        out[name + "_rgb8"] = np.fromfile(pre + ".begin_rgb8", np.uint8).reshape(H, W, 3)
        out[name + "_draws"] = int(meta[0])
        out[name + "_camera"] = np.fromfile(pre + ".begin_camera_f32", np.float32)  # topLeft, dd_x, dd_y, allLightIntensity
        print(f"begin_render/{name}: {W}x{H}, {int(meta[0])} rand() draws, mean byte {out[name + '_rgb8'].mean():.2f}")
    dat = np.fromfile(os.path.join(work, "Resource", "causticPhotonMap.dat"), np.uint8).reshape(-1, 24)
    # byte 19 (planeAndDirZ): bit 3 always valid, bits 0-1 (split plane) valid for internal nodes (index < half), the rest uninitialised
    half = len(dat) // 2 - 1
    mask = np.where(np.arange(1, len(dat) + 1) < half, 0x0B, 0x08).astype(np.uint8)
    dat[:, 19] &= mask
    out["caustic_photons"] = len(dat)
    out["caustic_photons_sha"] = sha(dat)
    out["caustic_photons_every_997"] = dat[::997].copy()
    print(f"begin_render/caustic: {len(dat)} photons in causticPhotonMap.dat")
    path = os.path.join(HERE, "begin_render.npz")
    np.savez_compressed(path, **out)
    print(f"begin_render: {os.path.getsize(path) / 1024:.0f} KiB")


GLOBAL_CASES = {"c5_caustics": 3000, "c2_glass_small": 2000, "c4_textured": 1500}


def global_photon_case(tmp):
    """Global photon map (BuildPhotonMap Main.cpp:251-295, TracePhotonRay Main.cpp:296-317, RandomPhotonBounce
    MtlBlinn.cpp:140-202): the reference's emission loop with the photon budget as a parameter, then its own
    ScalePhotonPowers + PrepareForIrradianceEstimation; one sequential rand() stream for the whole loop."""
    out = {}
    for name, n in GLOBAL_CASES.items():
        pre = os.path.join(tmp, "g_" + name)
        subprocess.run([HARNESS_PM, os.path.join(SCENES, name + ".xml"), pre, "--photons", str(n), "gphotons"], cwd=SCENES, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        meta = np.fromfile(pre + ".gphotons_meta", np.uint64)
        out[name + "_n"] = n
        out[name + "_meta"] = meta  # stored, emissions, rand() draws, halfStoredPhotons
        out[name + "_emitted"] = np.fromfile(pre + ".gphotons_emitted", np.uint8).reshape(-1, 24)
        out[name + "_balanced"] = np.fromfile(pre + ".gphotons_balanced", np.uint8).reshape(-1, 24)
        print(f"global_photon/{name}: {int(meta[0])} photons from {int(meta[1])} emissions, {int(meta[2])} draws")
    path = os.path.join(HERE, "global_photon.npz")
    np.savez_compressed(path, **out)
    print(f"global_photon: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
