"""Caustic photon map: Main.cpp:319-386, DataStructure/cyPhotonMap.h,
MtlBlinn.cpp:203-303,329-342.

CPU: the oracle against the golden vectors of the reference built with -DUSE_PhotonMap (tests/golden/
c5_caustics_photon.npz): emitted photons, kd-balanced array, k-NN irradiance estimates, radiance with the
caustic term — all bit-exact.  GPU: the HIP emission / gather kernels against the oracle in keyed + device-math mode."""
import os

import numpy as np
import pytest

from conftest import SCENES, same_bits


def check_photon_bytes(mine, ref, half=None):
    """Reference photons carry uninitialised bits in byte 19 (planeAndDirZ): bit 3 (dirZ sign) is always set
    deliberately, bits 0-1 (split plane) only for internal kd nodes."""
    m = np.ones(24, bool)
    m[19] = False
    assert np.array_equal(mine[:, m], ref[:, m])
    assert np.array_equal(mine[:, 19] & 8, ref[:, 19] & 8)
    if half is not None:
        internal = np.arange(1, len(ref) + 1) < half
        assert np.array_equal(mine[internal, 19] & 3, ref[internal, 19] & 3)


def test_oracle_photon_map_vs_reference_golden(load_scene, golden, O):
    g = golden("c5_caustics_photon")
    sc = load_scene("c5_caustics")
    bal, emitted, n_emit = O.photon_build(sc.flat_bytes(), int(g["n_photons"]), rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM)
    assert len(bal) == int(g["stored"]) and n_emit == int(g["emitted"])
    check_photon_bytes(emitted, g["photons_emitted"])                      # emission, bounce rules, 24-byte packing
    check_photon_bytes(bal, g["photons_balanced"], int(g["half"]))         # left-balanced kd-tree in heap order
    q = g["gather_q"]
    irr, d = O.photon_gather(q[:, :3], q[:, 3:], 0.5)                      # EstimateIrradiance<1000>(..., 0.5, p, &N)
    assert same_bits(np.concatenate([irr, d], 1), g["gather_out"])
    lit = g["gather_out"][:, :3].sum(1) > 0
    assert 100 < lit.sum() < len(q)
    region = tuple(int(x) for x in g["render_region"])
    r = O.render(sc.flat_bytes(), sc.width, sc.height, int(g["render_spp"]), gi=int(g["render_gi"]), rng=O.RNG_SEQUENTIAL,
                 math=O.MATH_LIBM, region=region, threads=4, photon=1)
    assert same_bits(r["samples"], g["render_samples"])                    # Shade() with the caustic term (MtlBlinn.cpp:329-342)


def test_oracle_global_photon_map_vs_reference_golden(load_scene, golden, O):
    """BuildPhotonMap / TracePhotonRay / RandomPhotonBounce (Main.cpp:251-317, MtlBlinn.cpp:140-202) of the reference
    (tests/golden/global_photon.npz) on a glass-over-plane scene, a box of diffuse and glossy objects far from the light
    (~100 emissions per stored photon) and a textured scene: emission count, emitted bytes, balanced bytes."""
    g = golden("global_photon")
    for name in ("c5_caustics", "c2_glass_small", "c4_textured"):
        sc = load_scene(name)
        n = int(g[name + "_n"])
        meta = g[name + "_meta"]
        bal, emitted, n_emit = O.photon_build_global(sc.flat_bytes(), n, rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM)
        assert len(bal) == int(meta[0]) == n and n_emit == int(meta[1]), name
        check_photon_bytes(emitted, g[name + "_emitted"])
        check_photon_bytes(bal, g[name + "_balanced"], int(np.int64(meta[3])))
    # not the caustic map: photons are stored at the first diffuse hit too
    cb, _, ce = O.photon_build(load_scene("c5_caustics").flat_bytes(), 3000, rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM)
    assert ce > 10 * int(g["c5_caustics_meta"][1])


def test_photon_record_layout_and_direction_quirk(load_scene, O):
    sc = load_scene("c5_caustics")
    bal, emitted, _ = O.photon_build(sc.flat_bytes(), 500)
    assert bal.dtype == np.uint8 and bal.shape == (500, 24)
    rec = np.dtype([("pos", "<f4", 3), ("power", "<f4"), ("color", "u1", 3), ("plane", "u1"), ("dx", "<i2"), ("dy", "<i2")])
    p = emitted.view(rec).reshape(-1)
    assert np.all(np.isfinite(p["pos"])) and np.all(p["power"] > 0)
    assert np.allclose(p["power"].astype(np.float64).sum() * 0 + p["color"].max(axis=1), 255)   # SetPower: max channel -> 255
    # powers were scaled by 1/N (ScalePhotonPowers, Main.cpp:380): total flux is O(light intensity), not O(N * intensity)
    assert p["power"].sum() < 1000
    # balancing is a permutation
    assert sorted(map(bytes, emitted[:, :19])) == sorted(map(bytes, bal[:, :19]))
    assert np.array_equal(O.photon_balance(emitted), bal)


def test_heap_mode_more_than_1000_candidates(load_scene, O):
    """> MAX_PhotonCountInArea photons inside the radius: the estimate switches to the max-heap with a shrinking
    radius (cyPhotonMap.h:458-495); the result must not depend on threads/calls."""
    sc = load_scene("c5_caustics")
    O.photon_build(sc.flat_bytes(), 30000)
    rng = np.random.RandomState(1)
    p = rng.uniform([-9.0, -7.0, 0.0], [-7.0, -5.0, 0.0], (64, 3)).astype(np.float32)
    n = np.tile(np.float32([0, 0, 1]), (64, 1))
    big, _ = O.photon_gather(p, n, 3.0)      # radius 3: thousands of candidates under the glass sphere
    small, _ = O.photon_gather(p, n, 0.5)
    assert (big.sum(1) > 0).all()
    again, _ = O.photon_gather(p, n, 3.0)
    assert same_bits(big, again)
    assert not same_bits(big, small)


# ------------------------------------------------------------------------------------------------ GPU
def check_selected_photons(sc, O, p, nrm, radius):
    """The default gather (bhrt_opts.photon_exact = 0) against the oracle's LocatePhotons: queries with fewer than 1000 photons in the
    radius keep the walk's own summation order (identical bits); the others must use exactly the reference's photons — the SET the heap
    history ends with, including its quirk of throwing out the farthest of the first 1000 unconditionally — and its np.dist2[0]; their sums
    run in another order, so the estimate is compared to a few ulp of the float sums (north_star: 1e-4 on radiance)."""
    gi, gd, knn, kc, dm = sc.photon_gather_ex(p, nrm, radius)
    oi, od = O.photon_gather(p, nrm, radius)
    oidx, ocnt, od2 = O.photon_knn(p, nrm, radius)
    heavy = ocnt >= 1000
    assert same_bits(gi[~heavy], oi[~heavy]) and same_bits(gd[~heavy], od[~heavy])
    sel = heavy & (kc > 0)                                   # answered by the selection pass (the rest went to the exact replay)
    assert same_bits(gi[heavy & ~sel], oi[heavy & ~sel]) and same_bits(gd[heavy & ~sel], od[heavy & ~sel])
    if heavy.any():
        assert sel.sum() >= 0.9 * heavy.sum()
        assert np.array_equal(kc[sel], ocnt[sel]) and same_bits(dm[sel], od2[sel])
        assert np.array_equal(np.sort(knn[sel], axis=1), np.sort(oidx[sel], axis=1))
        assert np.allclose(gi[sel], oi[sel], rtol=2e-5, atol=0) and np.allclose(gd[sel], od[sel], rtol=0, atol=2e-5)
    return int(heavy.sum()), int(sel.sum())


@pytest.mark.gpu
def test_gpu_photon_map_vs_oracle(B, load_scene, O):
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    sc = load_scene("c5_caustics")
    N = 20000
    opts = B.default_opts(seed=3)
    n = sc.photon_build(opts, N)
    bal, emitted, n_emit = O.photon_build(sc.flat_bytes(), N, seed=3)        # keyed + device math: same streams as the kernels
    assert n == len(bal) == N
    got = sc.photon_get()
    assert np.array_equal(got, bal)                                         # emission + compaction + balance: every byte
    # k-NN irradiance estimate, normal and heap mode, wrong-side normals
    rng = np.random.RandomState(9)
    p = np.concatenate([rng.uniform([-14, -20, 0.0], [14, 10, 0.0], (300, 3)), rng.uniform([-11, -9, 0.0], [-5, -3, 0.0], (500, 3))]).astype(np.float32)
    nr = np.tile(np.float32([0, 0, 1]), (len(p), 1))
    nr[::9] = [0, 0, -1]
    for radius in (0.5, 2.5):
        gi, gd = sc.photon_gather(p, nr, radius, exact=True)                # bhrt_opts.photon_exact: the heap history replayed
        oi, od = O.photon_gather(p, nr, radius)
        assert same_bits(gi, oi) and same_bits(gd, od)
        check_selected_photons(sc, O, p, nr, radius)                        # default: the same photons, selected by a wave
    assert (oi.sum(1) > 0).sum() > 300
    # the three gather passes agree: lane walk only / every query by a whole wave (walk order rebuilt by rank sort) / default mix
    import os
    rn = rng.normal(size=(len(p), 3)).astype(np.float32)
    rn /= np.linalg.norm(rn, axis=1, keepdims=True)
    for radius in (0.5, 1.1):
        oi, od = O.photon_gather(p, rn, radius)
        for budget in (1000000000, 1, 0):  # lane walk only / every query by a whole wave / the default mix (knob off)
            sc.knob("gather_lane_budget", budget)
            gi, gd = sc.photon_gather(p, rn, radius, exact=True)
            assert same_bits(gi, oi) and same_bits(gd, od), (radius, budget)
            check_selected_photons(sc, O, p, rn, radius)
    found = (oi.sum(1) > 0)
    assert found.sum() > 100 and (~found).sum() > 100
    # radiance with the caustic term
    region = (90, 150, 200, 215)
    ropts = B.default_opts(spp=2, gi_bounces=2, seed=3, photon_map=1)
    gs, st = sc.render_samples(ropts, *region)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 2, gi=2, seed=3, region=region, photon=1)
    assert np.nanmax(np.abs(gs - ro["samples"])) <= 1e-4                    # north_star's bar, selection pass for the heavy queries
    ropts.photon_exact = 1
    gx, stx = sc.render_samples(ropts, *region)
    assert same_bits(gx, ro["samples"])                                     # the exact replay: identical bits
    assert st.photon_heavy_queries == stx.photon_heavy_queries == stx.photon_exact_queries and st.photon_exact_queries <= st.photon_heavy_queries // 20
    off, _ = sc.render_samples(B.default_opts(spp=2, gi_bounces=2, seed=3), *region)
    assert not same_bits(gs, off)                                           # the caustic actually contributes here
    # export = the reference's .dat (24-byte records, Main.cpp:383-385)
    import os, tempfile
    path = os.path.join(tempfile.mkdtemp(), "causticPhotonMap.dat")
    sc.photon_export(path)
    assert np.array_equal(np.fromfile(path, np.uint8).reshape(-1, 24), bal)
    # import: as is (the cached photon pass) and through InitializePhotonMapByFile, which balances the records again
    # (cyPhotonMap.h:409-417)
    ref_i, ref_d = sc.photon_gather(p, nr, 0.5, exact=True)
    sc2 = load_scene("c5_caustics")
    sc2.photon_import(path)
    assert np.array_equal(sc2.photon_get(), bal)
    gi2, gd2 = sc2.photon_gather(p, nr, 0.5, exact=True)
    assert same_bits(gi2, ref_i) and same_bits(gd2, ref_d)
    sc2.photon_import(path, rebalance=True)
    rebal = O.photon_balance(bal)
    assert np.array_equal(sc2.photon_get(), rebal)
    O.photon_attach(rebal)
    oi3, od3 = O.photon_gather(p, nr, 0.5)
    gi3, gd3 = sc2.photon_gather(p, nr, 0.5, exact=True)
    assert same_bits(gi3, oi3) and same_bits(gd3, od3)
    with pytest.raises(B.BhrtError):
        sc2.photon_import(path + ".missing")


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("c5_caustics", 50000), ("c2_glass_small", 3000), ("c4_textured", 3000), ("c3_mesh_small", 3000)])
def test_gpu_global_photon_map_vs_oracle(name, n, B, load_scene, O, tmp_path):
    """bhrt_photon_build_global: every byte of the balanced global map equals the oracle's (keyed streams, device math);
    the .dat it writes holds the same records; the caustic map installed for gathers is left alone."""
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    sc = load_scene(name)
    opts = B.default_opts(seed=5)
    dat = str(tmp_path / "photonmap.dat")
    got = sc.photon_build_global(opts, n, dat_path=dat)
    bal, _, n_emit = O.photon_build_global(sc.flat_bytes(), n, seed=5)
    assert len(got) == len(bal) == n and n_emit > 0
    assert np.array_equal(got, bal)
    assert np.array_equal(np.fromfile(dat, np.uint8).reshape(-1, 24), bal)
    if name == "c5_caustics":
        k = sc.photon_build(opts, 2000)
        before = sc.photon_get()
        sc.photon_build_global(opts, 500)
        assert k == 2000 and np.array_equal(sc.photon_get(), before)


@pytest.mark.gpu
def test_gpu_photon_map_on_random_scenes(B, O, tmp_path):
    """Seeded random scenes (tools/fuzz_parity.py; seeds whose scenes have a caustic path, so that the emission loop
    ends): emission through nested groups and mesh instances, balance, gathers at two radii with random normals and
    photon-mapped radiance, all bit for bit against the oracle.  (27 such scenes were run by hand.)"""
    import shutil, sys
    from conftest import ROOT, SCENES
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    shutil.copy(os.path.join(SCENES, "mesh_small.obj"), tmp_path / "mesh_small.obj")
    for seed in (10, 13, 17, 31, 44):
        rng = np.random.default_rng(seed)
        xml = str(tmp_path / f"f{seed}.xml")
        fuzz_parity.random_scene(rng, xml)
        sc = B.Scene(xml)
        blob = sc.flat_bytes()
        n = sc.photon_build(B.default_opts(seed=seed), 3000)
        bal, _, _ = O.photon_build(blob, 3000, seed=seed)
        assert n == len(bal) and np.array_equal(sc.photon_get(), bal), seed
        q = np.random.default_rng(seed + 1)
        pts = sc.photon_get()[:, :12].copy().view(np.float32).reshape(-1, 3)
        p = (pts[q.integers(0, len(pts), 400)] + q.normal(scale=0.2, size=(400, 3))).astype(np.float32)
        nr = q.normal(size=(400, 3)).astype(np.float32)
        nr /= np.linalg.norm(nr, axis=1, keepdims=True)
        for radius in (0.5, 3.0):
            gi_, gd = sc.photon_gather(p, nr, radius, exact=True)
            oi, od = O.photon_gather(p, nr, radius)
            assert same_bits(gi_, oi) and same_bits(gd, od), (seed, radius)
            O.photon_attach(sc.photon_get())
            check_selected_photons(sc, O, p, nr, radius)
        gs, _ = sc.render_samples(B.default_opts(spp=1, gi_bounces=1, seed=seed, photon_map=1, photon_exact=1), 0, 0, sc.width, sc.height)
        rs = O.render(blob, sc.width, sc.height, 1, gi=1, seed=seed, region=(0, 0, sc.width, sc.height), photon=1)["samples"]
        assert same_bits(gs, rs), seed
        gs, _ = sc.render_samples(B.default_opts(spp=1, gi_bounces=1, seed=seed, photon_map=1), 0, 0, sc.width, sc.height)
        assert np.nanmax(np.abs(gs - rs)) <= 1e-4, seed


@pytest.mark.gpu
def test_gpu_photon_render_pass_size_and_partition_invariance(B, load_scene):
    """The caustic term is gathered once per pass over all of the pass's frames: the image must not depend on how the frame
    is cut into passes (1 / 82 / 554 passes) or into ranks (tiles of 2 and 3 logical ranks add up to the full image)."""
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    sc = load_scene("c5_caustics")
    sc.upload(0)
    sc.photon_build(B.default_opts(seed=1), 50000)
    ref = sc.render(B.default_opts(spp=4, gi_bounces=2, seed=1, photon_map=1))
    assert ref[2].passes == 1
    for per_pass, n_pass in ((4000, 82), (4 * 148, 554)):
        r = sc.render(B.default_opts(spp=4, gi_bounces=2, seed=1, photon_map=1, samples_per_pass=per_pass))
        assert r[2].passes == n_pass and np.array_equal(r[0], ref[0]) and r[1].tobytes() == ref[1].tobytes()
    for world in (2, 3):
        acc = np.zeros_like(ref[1])
        for rank in range(world):
            acc += sc.render(B.default_opts(spp=4, gi_bounces=2, seed=1, photon_map=1, rank=rank, world_size=world, tile_size=16))[1]
        assert acc.tobytes() == ref[1].tobytes()


@pytest.mark.gpu
def test_gpu_sharded_photon_build_equals_single_build(B, load_scene):
    """Multi-GPU build of the caustic map (bhraytracer_amd/dist.py::photon_build_sharded): ranks emit disjoint
    emission-index ranges, the records are concatenated in emission order, the first max_photons are installed.  Logical
    ranks on one GPU here (the exchange itself is covered by tools/verify_multi_rank.py): the installed map equals
    bhrt_photon_build's byte for byte, for the caustic and (records only) the global emission."""
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    sc = load_scene("c5_caustics")
    opts = B.default_opts(seed=7)
    N = 30000
    sc.photon_build(opts, N)
    single = sc.photon_get()
    for world, per_rank in ((2, 1 << 16), (3, 87 * 256)):
        kept, total, e0 = [], 0, 0
        while total < N:
            for rank in range(world):
                blk = sc.photon_emit_range(opts, e0 + rank * per_rank, per_rank)
                kept.append(blk)
                total += len(blk)
            e0 += per_rank * world
        assert sc.photon_install(np.concatenate(kept)[:N]) == N
        assert np.array_equal(sc.photon_get(), single), world
    g = sc.photon_build_global(opts, 5000)
    rec = np.concatenate([sc.photon_emit_range(opts, r * 4096, 4096, global_map=True) for r in range(4)])
    assert len(rec) >= 5000 and len(g) == 5000
    with pytest.raises(B.BhrtError):
        sc.photon_emit_range(opts, 0, 100)          # not a multiple of 256
    with pytest.raises(B.BhrtError):
        sc.photon_emit_range(opts, 0, 4096, global_map=True, capacity=16)   # buffer too small


@pytest.mark.gpu
def test_gpu_photon_records_stay_on_the_device(B, load_scene):
    """The multi-GPU build with device-resident exchange buffers (what dist.photon_build_sharded does under RCCL): emission writes
    its records to a device pointer, they are strung together on the device and installed from a device pointer — and a buffer
    that is too small reports how much room is needed.  World size 1 through photon_build_sharded(device=cuda) itself."""
    import torch
    import bhraytracer_amd.dist as BD
    sc = load_scene("c5_caustics")
    opts = B.default_opts(seed=7)
    N = 30000
    sc.photon_build(opts, N)
    single = sc.photon_get().copy()
    dev = torch.device("cuda", 0)
    small = torch.zeros((16, 24), dtype=torch.uint8, device=dev)
    n, ok = sc.photon_emit_range_into(opts, 0, 1 << 16, small.data_ptr(), 16)
    assert not ok and n > 16 and not small.any()                  # nothing written, the needed size reported
    parts, total, e0 = [], 0, 0
    while total < N:
        buf = torch.zeros((n + 4096, 24), dtype=torch.uint8, device=dev)
        m, ok = sc.photon_emit_range_into(opts, e0, 1 << 16, buf.data_ptr(), len(buf))
        assert ok
        parts.append(buf[:m])
        total += m
        e0 += 1 << 16
    rec = torch.cat(parts)[:N].contiguous()
    torch.cuda.synchronize()
    assert sc.photon_install_ptr(rec.data_ptr(), N) == N and np.array_equal(sc.photon_get(), single)
    assert BD.photon_build_sharded(sc, opts, N, 0, 1, device=dev) == N and np.array_equal(sc.photon_get(), single)
    assert BD.photon_build_sharded(sc, opts, N, 0, 1, device=None, batch=1 << 18) == N and np.array_equal(sc.photon_get(), single)


def test_the_list_locate_photons_ends_with_is_a_set_rule(O):
    """What the HIP path's selection pass relies on (device_photon.h): with A = the acceptable photons inside the radius in walk order, F its
    first 1000 and m the farthest of F, LocatePhotons ends with the 1000 nearest of A - {m} (the 1001st photon replaces the heap's root
    unconditionally, np.dist2[0] is still r^2 then; from there on it is a streaming selection), and np.dist2[0] = their largest distance.
    Checked here on the CPU against the oracle's faithful replay of the heap (itself pinned to the reference): a Python walk of the
    balanced map gives A in walk order, numpy applies the rule."""
    rng = np.random.RandomState(11)
    n = 9000
    rec = np.zeros((n, 24), np.uint8)
    pos = rng.uniform([-1.2, -1.2, 0], [1.2, 1.2, 0.3], (n, 3)).astype(np.float32)
    rec[:, 0:12] = pos.view(np.uint8).reshape(n, 12)
    rec[:, 12:16] = np.frombuffer(np.float32(1.0 / n).tobytes(), np.uint8)
    rec[:, 16:19] = 255
    rec[:, 19] = 0x8          # dirX = dirY = 0, dirZ < 0: direction (0, 0, -1), accepted by every query with normal (0, 0, 1)
    bal = O.photon_balance(rec)
    O.photon_attach(bal)
    P = bal[:, 0:12].copy().view(np.float32).reshape(n, 3)
    axis = bal[:, 19] & 3
    half = n // 2 - 1
    r2 = np.float32(0.5) * np.float32(0.5)

    def d2_of(i, q):  # photon i (1-based), float32 like Vec3f::LengthSquared
        d = (P[i - 1] - q).astype(np.float32)
        return np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    def walk(i, q, out):  # LocatePhotons' order with the radius fixed at r (what holds until the 1001st photon)
        if i < half:
            dist = np.float32(q[axis[i - 1]] - P[i - 1][axis[i - 1]])
            near, far = (2 * i + 1, 2 * i) if dist > 0 else (2 * i, 2 * i + 1)
            walk(near, q, out)
            if np.float32(dist * dist) < r2:
                walk(far, q, out)
        if d2_of(i, q) < r2:
            out.append(i)

    q = np.concatenate([rng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 0.0], (40, 3)), rng.uniform([-1.6, -1.6, 0.0], [1.6, 1.6, 0.0], (20, 3))]).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, 1]), (len(q), 1))
    idx, cnt, d2max = O.photon_knn(q, nrm, 0.5)
    heavy = 0
    for k in range(len(q)):
        A = []
        walk(1, q[k], A)
        dist = {i: d2_of(i, q[k]) for i in A}
        if len(A) <= 1000:
            expect, bound = sorted(A), r2
        else:
            heavy += 1
            F = A[:1000]
            m = max(F, key=lambda i: dist[i])
            rest = sorted((i for i in A if i != m), key=lambda i: dist[i])[:1000]
            expect, bound = sorted(rest), max(dist[i] for i in rest)
        assert cnt[k] == len(expect) and np.array_equal(np.sort(idx[k][:cnt[k]]), np.uint32(expect)), k
        assert np.float32(d2max[k]) == np.float32(bound), k
    assert heavy >= 20


def _records(pos, rng):
    """(n, 24) uint8 photon records (cyPhotonMap.h:72-90) at the given positions, everything else random, power 0 (ScalePhotonPowers keeps it)."""
    n = len(pos)
    rec = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    rec[:, :12] = np.ascontiguousarray(pos, np.float32).view(np.uint8).reshape(n, 12)
    rec[:, 12:16] = 0
    return rec


@pytest.mark.gpu
def test_device_balance_equals_the_oracle_on_tie_heavy_maps(B, load_scene, O):
    """k_pb_level (device_photon_build.h): PrepareForIrradianceEstimation / BalanceSegment (cyPhotonMap.h:236-328) on the device.  The balanced map
    depends on the exact swap sequence of every Hoare partition wherever keys are equal (and deep segments split along the axis of an INHERITED
    box, where a caustic on a floor has nothing but equal keys), so the maps here are made of ties: a floor (one z), a coarse grid (few distinct
    values per axis), duplicates of a handful of points, all-equal points, -0 beside +0, and the small sizes around the median rule's cases.
    Every byte of the installed map equals the oracle's restatement of the reference's routine."""
    sc = B.Scene(os.path.join(SCENES, "c5_caustics.xml"))
    sc.upload(0)
    rng = np.random.default_rng(11)
    cases = []
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 31, 33, 64, 65, 100, 1000, 4097, 50000, 300000):
        kind = n % 5
        if kind == 0:    # a floor: z equal everywhere, x / y continuous
            p = rng.uniform(-3, 3, (n, 3)); p[:, 2] = 0.25
        elif kind == 1:  # coarse grid: 4 values per axis
            p = rng.integers(0, 4, (n, 3)) * 0.5 - 1.0
        elif kind == 2:  # duplicates of a few points, zeros of both signs
            base = rng.normal(size=(5, 3)); base[0] = [0.0, -0.0, 0.0]; base[1] = [-0.0, 0.0, -0.0]
            p = base[rng.integers(0, 5, n)]
        elif kind == 3:  # everything equal
            p = np.tile(rng.normal(size=(1, 3)), (n, 1))
        else:            # continuous cloud with a dense cluster
            p = rng.normal(size=(n, 3)); p[: n // 2] = p[: n // 2] * 1e-3 + 2.0
        cases.append(_records(p, rng))
    cases.append(_records(np.repeat(rng.uniform(-1, 1, (1000, 3)), 100, axis=0), rng))  # 100 copies of each of 1000 points, grouped
    for rec in cases:
        n = sc.photon_install(rec)
        got = sc.photon_get()
        want = O.photon_balance(rec)
        assert got.shape == want.shape == (n, 24) and np.array_equal(got, want), n


@pytest.mark.gpu
def test_device_balance_and_host_routine_agree_on_a_built_map(tmp_path):
    """Second opinion: the same 200 k-photon build with the balance done by the host routine (photon_host.cpp, BHRT_PHOTON_BALANCE_HOST=1: the
    records make the round trip through the host) and on the device (default: they never leave HBM) — the same file, byte for byte."""
    import subprocess
    from conftest import ROOT
    cli = os.path.join(ROOT, "bhraytracer_amd", "bhrt")
    xml = os.path.join(SCENES, "c5_caustics.xml")
    outs = []
    for host in ("0", "1"):
        dat = tmp_path / f"m{host}.dat"
        r = subprocess.run([cli, "render", xml, "-o", str(tmp_path / f"m{host}.png"), "--spp", "1", "--gi", "1", "--photons", "200000", "--photon-out", str(dat)],
                           cwd=SCENES, capture_output=True, text=True, timeout=600, env=dict(os.environ, BHRT_PHOTON_BALANCE_HOST=host))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(open(dat, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) == 200000 * 24
