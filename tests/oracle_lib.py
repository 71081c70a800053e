"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so) — the CHECKER.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

RNG_SEQUENTIAL, RNG_KEYED = 0, 1
MATH_LIBM, MATH_DEVICE = 0, 1
HIT_FLOATS = 16


class OracleOpts(C.Structure):
    _fields_ = [("spp", C.c_int32), ("gi_bounces", C.c_int32), ("internal_bounces", C.c_int32), ("seed", C.c_uint32),
                ("rng_mode", C.c_int32), ("math_mode", C.c_int32), ("jitter", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("threads", C.c_int32), ("photon_gather", C.c_int32)]


class OracleStats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shade_calls", C.c_uint64),
                ("samples", C.c_uint64), ("seconds", C.c_double)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {lib().oracle_last_error().decode()}")


def trace_closest(blob: bytes, origins, dirs, hit_side=1):
    rays = np.ascontiguousarray(np.concatenate([origins, dirs], axis=1), np.float32)
    n = rays.shape[0]
    node = np.empty(n, np.int32)
    face = np.empty(n, np.int32)
    front = np.empty(n, np.int32)
    attrs = np.empty((n, HIT_FLOATS), np.float32)
    _check(lib().oracle_trace_closest(C.c_char_p(blob), _p(rays), int(hit_side), C.c_size_t(n), _p(node), _p(face),
                                      _p(front), _p(attrs)))
    return {"node": node, "prim": face, "front": front, "t": attrs[:, 0].copy(), "attrs": attrs}


def trace_shadow(blob: bytes, origins, dirs, tmax):
    rays = np.ascontiguousarray(np.concatenate([origins, dirs], axis=1), np.float32)
    n = rays.shape[0]
    tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)), np.float32)
    vis = np.empty(n, np.float32)
    _check(lib().oracle_trace_shadow(C.c_char_p(blob), _p(rays), _p(tm), C.c_size_t(n), _p(vis)))
    return vis


def render(blob: bytes, width, height, spp, gi=3, bounces=16, seed=0, rng=RNG_KEYED, math=MATH_DEVICE, jitter=1,
           region=None, threads=8, want_samples=True, photon=0):
    x0, y0, x1, y1 = region if region else (0, 0, width, height)
    o = OracleOpts(spp, gi, bounces, seed, rng, math, jitter, x0, y0, x1, y1, threads, photon)
    npx = (x1 - x0) * (y1 - y0)
    samples = np.zeros((npx, spp, 3), np.float32) if want_samples else None
    rad = np.zeros((npx, 3), np.float32)
    rgb = np.zeros((npx, 3), np.uint8)
    st = OracleStats()
    _check(lib().oracle_render(C.c_char_p(blob), C.byref(o), _p(samples) if want_samples else None, _p(rad), _p(rgb),
                               C.byref(st)))
    return {"samples": samples, "radiance": rad.reshape(y1 - y0, x1 - x0, 3), "rgb8": rgb.reshape(y1 - y0, x1 - x0, 3),
            "stats": st}


def photon_build(blob: bytes, max_photons, seed=0, rng=RNG_KEYED, math=MATH_DEVICE):
    """BuildCausticPhotonMap (Main.cpp:342-386); the map stays attached for photon_gather / render(photon=1).
    Returns (balanced (n,24) uint8, emission-order (n,24) uint8, n_emitted)."""
    o = OracleOpts(1, 3, 16, seed, rng, math, 1, 0, 0, 0, 0, 1, 0)
    out = np.zeros((max_photons, 24), np.uint8)
    ns, ne = C.c_uint32(), C.c_uint64()
    _check(lib().oracle_photon_build(C.c_char_p(blob), C.byref(o), int(max_photons), _p(out), C.byref(ns), C.byref(ne)))
    unb = np.zeros((ns.value, 24), np.uint8)
    _check(lib().oracle_photon_unbalanced(_p(unb)))
    return out[:ns.value].copy(), unb, ne.value


def begin_render(blob: bytes, width, height, spp=32, gi=3, bounces=16, seed=0, photon_budget=0, math=MATH_LIBM):
    """BeginRender() as one program with one rand() stream (Main.cpp:178-242).  Returns (rgb8 (H, W, 3), balanced photons or None,
    emissions, draws)."""
    o = OracleOpts(spp, gi, bounces, seed, RNG_SEQUENTIAL, math, 1, 0, 0, 0, 0, 1, 1 if photon_budget else 0)
    rgb = np.zeros((height, width, 3), np.uint8)
    ph = np.zeros((max(photon_budget, 1), 24), np.uint8)
    ns, ne, nd = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    _check(lib().oracle_begin_render(C.c_char_p(blob), C.byref(o), int(photon_budget), _p(rgb), _p(ph), C.byref(ns), C.byref(ne), C.byref(nd)))
    return rgb, (ph[: ns.value].copy() if photon_budget else None), ne.value, nd.value


def photon_build_global(blob: bytes, max_photons, seed=0, rng=RNG_KEYED, math=MATH_DEVICE):
    """BuildPhotonMap (Main.cpp:251-317): the global photon map.  Returns (balanced, emission order, emissions)."""
    o = OracleOpts(1, 0, 0, seed, rng, math, 1, 0, 0, 0, 0, 1, 0)
    out = np.zeros((max_photons, 24), np.uint8)
    em = np.zeros((max_photons, 24), np.uint8)
    ns, ne = C.c_uint32(0), C.c_uint64(0)
    _check(lib().oracle_photon_build_global(C.c_char_p(blob), C.byref(o), int(max_photons), _p(out), _p(em), C.byref(ns), C.byref(ne)))
    return out[: ns.value].copy(), em[: ns.value].copy(), int(ne.value)


def first_hit(blob: bytes, width, height, math=MATH_DEVICE):
    """First hit of every pixel's un-jittered camera ray: (z (H*W), normal (H*W,3), albedo (H*W,3))."""
    n = width * height
    z, nrm, alb = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    _check(lib().oracle_first_hit(C.c_char_p(blob), int(math), _p(z), _p(nrm), _p(alb)))
    return z, nrm, alb


def zbuffer_image(z):
    """RenderImage::ComputeZBufferImage (scene.h:578-600)."""
    z = np.ascontiguousarray(z, np.float32)
    img = np.zeros(z.size, np.uint8)
    _check(lib().oracle_zbuffer_image(_p(z), C.c_size_t(z.size), _p(img)))
    return img


def color_image(radiance, gamma=1, math=MATH_DEVICE):
    """colorArray of BeginRender (Main.cpp:219-229): pow(c, 1/2.2f) as floats."""
    r = np.ascontiguousarray(radiance, np.float32)
    out = np.zeros_like(r)
    _check(lib().oracle_color_image(_p(r), C.c_size_t(r.size), int(gamma), int(math), _p(out)))
    return out


def photon_attach(balanced):
    a = np.ascontiguousarray(balanced, np.uint8)
    _check(lib().oracle_photon_attach(_p(a), a.shape[0]))


def photon_balance(emitted):
    a = np.ascontiguousarray(emitted, np.uint8)
    out = np.zeros_like(a)
    _check(lib().oracle_photon_balance(_p(a), a.shape[0], _p(out)))
    return out


def photon_gather(p, nrm, radius=0.5):
    p = np.ascontiguousarray(p, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    irr = np.zeros_like(p)
    d = np.zeros_like(p)
    _check(lib().oracle_photon_gather(_p(p), _p(nrm), C.c_size_t(p.shape[0]), C.c_float(radius), _p(irr), _p(d)))
    return irr, d


def photon_knn(p, nrm, radius=0.5):
    """The list LocatePhotons ends with: (indices (n, 1000) ascending and 0-padded, count (n,), np.dist2[0] (n,))."""
    p = np.ascontiguousarray(p, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    idx = np.zeros((p.shape[0], 1000), np.uint32)
    cnt = np.zeros(p.shape[0], np.uint32)
    d2 = np.zeros(p.shape[0], np.float32)
    _check(lib().oracle_photon_knn(_p(p), _p(nrm), C.c_size_t(p.shape[0]), C.c_float(radius), _p(idx), _p(cnt), _p(d2)))
    return idx, cnt, d2


def bvh_build(v, f, max_per_leaf=4):
    v = np.ascontiguousarray(v, np.float32)
    f = np.ascontiguousarray(f, np.uint32)
    nf = f.shape[0]
    cap = 2 * nf + 2
    nodes = np.zeros((cap, 8), np.uint32)
    elems = np.zeros(nf, np.uint32)
    n = lib().oracle_bvh_build(_p(v), _p(f), nf, max_per_leaf, _p(nodes), C.c_size_t(cap), _p(elems))
    if n < 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return nodes[:n].copy(), elems


def math_eval(fn: int, mode: int, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    _check(lib().oracle_math_eval(fn, mode, _p(a), _p(bb) if bb is not None else None, C.c_size_t(a.size), _p(out)))
    return out


def primary_rays(flat_view):
    """The reference's un-jittered camera rays (Main.cpp:145,153) for every pixel, row-major j*W+i."""
    c = flat_view.header.camera
    W, H = c.width, c.height
    tl = np.array(list(c.top_left), np.float32)
    dx = np.array(list(c.dd_x), np.float32)
    dy = np.array(list(c.dd_y), np.float32)
    pos = np.array(list(c.pos), np.float32)
    ii, jj = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    pc = (tl[None, None, :] + ii[..., None] * dx[None, None, :]) - jj[..., None] * dy[None, None, :]
    d = (pc - pos[None, None, :]).reshape(-1, 3).astype(np.float32)
    o = np.broadcast_to(pos, d.shape).astype(np.float32).copy()
    return o, d
