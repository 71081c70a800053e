"""bhraytracer_amd — ctypes binding of libbhrt.so (include/bhrt.h).

Plumbing only: the product is the C-ABI library (host front-end in C++ + hand-written gfx950
HIP kernels).  This module loads it for the tests, bench.py and tools, and fails loudly when
the library has not been built — there is no Python or CPU fallback for the render path.
"""
import ctypes as C
import os

import numpy as np

from .flat import FlatView

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BHRT_LIB") or os.path.join(_HERE, "libbhrt.so")  # BHRT_LIB: load another build of the same ABI

SIDE_FRONT, SIDE_BACK, SIDE_BOTH = 1, 2, 3


class BhrtError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32)] + [
        (n, C.c_uint32) for n in ("n_nodes", "n_meshes", "n_triangles", "n_bvh_nodes", "n_materials", "n_lights",
                                  "n_textures", "max_node_depth", "max_bvh_depth")] + [
        ("flat_bytes", C.c_uint64), ("n_warnings", C.c_uint32)]


class Opts(C.Structure):
    _fields_ = [("spp", C.c_int32), ("gi_bounces", C.c_int32), ("internal_bounces", C.c_int32), ("seed", C.c_uint32),
                ("jitter", C.c_int32), ("gamma", C.c_int32), ("photon_map", C.c_int32),
                ("rank", C.c_int32), ("world_size", C.c_int32), ("tile_size", C.c_int32),
                ("samples_per_pass", C.c_int32), ("timers", C.c_int32), ("photon_exact", C.c_int32), ("leaf_skip", C.c_int32), ("photon_radius", C.c_float), ("reserved", C.c_int32 * 1)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shade_calls", C.c_uint64),
                ("camera_samples", C.c_uint64), ("passes", C.c_uint32), ("wave_iterations", C.c_uint32),
                ("seconds_total", C.c_double), ("seconds_trace_closest", C.c_double),
                ("seconds_trace_shadow", C.c_double), ("seconds_shade", C.c_double), ("seconds_other", C.c_double),
                ("launches_trace_closest", C.c_uint64), ("launches_trace_shadow", C.c_uint64),
                ("seconds_photon_gather", C.c_double), ("seconds_photon_heavy", C.c_double),
                ("photon_queries", C.c_uint64), ("photon_heavy_queries", C.c_uint64), ("photon_wave_queries", C.c_uint64),
                ("photon_exact_queries", C.c_uint64), ("photon_nodes_visited", C.c_uint64), ("deferred_rays", C.c_uint64),
                ("photon_lane_queries", C.c_uint64), ("photon_lane_nodes", C.c_uint64), ("photon_found", C.c_uint64), ("reserved", C.c_double * 1)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class Hits(C.Structure):
    _fields_ = [("t", C.c_void_p), ("node", C.c_void_p), ("prim", C.c_void_p), ("front", C.c_void_p)]


_lib = None

# every symbol include/bhrt.h declares
EXPORTS = [
    "bhrt_last_error", "bhrt_default_opts", "bhrt_scene_load_xml", "bhrt_scene_free", "bhrt_scene_info",
    "bhrt_scene_warning", "bhrt_scene_flat", "bhrt_scene_upload", "bhrt_device_count",
    "bhrt_trace_closest_host", "bhrt_trace_closest_dev", "bhrt_trace_shadow_host", "bhrt_trace_shadow_dev",
    "bhrt_render", "bhrt_render_dev", "bhrt_render_samples", "bhrt_photon_build", "bhrt_photon_gather_host",
    "bhrt_photon_get", "bhrt_photon_export", "bhrt_photon_import", "bhrt_photon_build_global", "bhrt_save_png", "bhrt_math_eval_dev",
    "bhrt_tiles_block_bytes", "bhrt_tiles_pack_dev", "bhrt_tiles_unpack_dev",
    "bhrt_first_hit", "bhrt_first_hit_dev", "bhrt_zbuffer_image_dev", "bhrt_color_image_dev",
    "bhrt_scene_load_xml_ex", "bhrt_bvh_build", "bhrt_photon_emit_range", "bhrt_photon_install", "bhrt_scene_clone", "bhrt_host_alloc", "bhrt_host_free", "bhrt_photon_gather_host_ex", "bhrt_scene_knob",
]


def lib():
    """Loads libbhrt.so; raises if it is missing (run `python -m bhraytracer_amd.build` first)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BhrtError(f"{LIB_PATH} not found: build it with `python -m bhraytracer_amd.build` "
                            "(the render path has no fallback)")
        # torch ships its own libamdhip64 under the SONAME of the system's: whichever copy a process loads first serves
        # both.  Two HIP runtimes in one process do not both see the GPU, so when torch is present (tests, bench.py:
        # device tensors, torch.distributed) it goes first.  The library itself does not depend on torch.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        L.bhrt_last_error.restype = C.c_char_p
        L.bhrt_scene_free.restype = None
        L.bhrt_default_opts.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise BhrtError(f"bhrt error {rc}: {lib().bhrt_last_error().decode(errors='replace')}")


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().bhrt_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().bhrt_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """A loaded scene = the reference's LoadScene() globals behind one handle (Main.cpp:17-37,43)."""

    def __init__(self, xml_path: str, bvh_device: int = -1):
        """bvh_device >= 0: mesh BVHs are built on that HIP device (bhrt_scene_load_xml_ex) instead of by the host front-end."""
        self._h = C.c_void_p()
        _check(lib().bhrt_scene_load_xml_ex(os.fsencode(xml_path), int(bvh_device), C.byref(self._h)))
        self.info = Info()
        _check(lib().bhrt_scene_info(self._h, C.byref(self.info)))
        self._flat = None

    def close(self):
        if self._h:
            lib().bhrt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def width(self):
        return self.info.width

    @property
    def height(self):
        return self.info.height

    def warnings(self):
        out = []
        for i in range(self.info.n_warnings):
            s = C.c_char_p()
            _check(lib().bhrt_scene_warning(self._h, i, C.byref(s)))
            out.append(s.value.decode(errors="replace"))
        return out

    def flat_bytes(self) -> bytes:
        if self._flat is None:
            p, n = C.c_void_p(), C.c_uint64()
            _check(lib().bhrt_scene_flat(self._h, C.byref(p), C.byref(n)))
            self._flat = C.string_at(p, n.value)
        return self._flat

    def flat_view(self) -> FlatView:
        return FlatView(self.flat_bytes())

    def upload(self, device: int = 0):
        _check(lib().bhrt_scene_upload(self._h, device))

    # ---- hot path, host buffers -------------------------------------------------------------
    def trace_closest(self, origins, dirs, hit_side=SIDE_FRONT):
        """recursive() for n rays; origins/dirs: (n,3) float32. Returns dict of numpy arrays."""
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        soa = np.ascontiguousarray(np.concatenate([o.T, d.T], axis=0), np.float32)  # ox[n] oy[n] oz[n] dx..
        t = np.empty(n, np.float32)
        node = np.empty(n, np.int32)
        prim = np.empty(n, np.int32)
        front = np.empty(n, np.int32)
        h = Hits(_ptr(t), _ptr(node), _ptr(prim), _ptr(front))
        _check(lib().bhrt_trace_closest_host(self._h, _ptr(soa), int(hit_side), C.c_size_t(n), h))
        return {"t": t, "node": node, "prim": prim, "front": front}

    def trace_shadow(self, origins, dirs, tmax):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        soa = np.ascontiguousarray(np.concatenate([o.T, d.T], axis=0), np.float32)
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)), np.float32)
        vis = np.empty(n, np.float32)
        _check(lib().bhrt_trace_shadow_host(self._h, _ptr(soa), _ptr(tm), C.c_size_t(n), _ptr(vis)))
        return vis

    def render(self, opts: Opts, want_radiance=True):
        """BeginRender(): returns (rgb8 HxWx3 uint8, radiance HxWx3 float32 or None, Stats)."""
        W, H = self.width, self.height
        rgb = np.zeros((H, W, 3), np.uint8)
        rad = np.zeros((H, W, 3), np.float32) if want_radiance else None
        st = Stats()
        _check(lib().bhrt_render(self._h, C.byref(opts), _ptr(rgb), _ptr(rad) if want_radiance else None, C.byref(st)))
        return rgb, rad, st

    def render_dev(self, opts: Opts, d_rgb8_ptr: int, d_radiance_ptr: int):
        """Same with outputs left in HBM (raw device pointers, e.g. torch tensor .data_ptr())."""
        st = Stats()
        _check(lib().bhrt_render_dev(self._h, C.byref(opts), C.c_void_p(d_rgb8_ptr or None),
                                     C.c_void_p(d_radiance_ptr or None), C.byref(st), None))
        return st

    # ---- images beside the colour image (RenderImage z-buffer, DenoiseImage inputs) ------------
    def first_hit(self):
        """First hit of every pixel's un-jittered camera ray: z (H, W), normal (H, W, 3), albedo (H, W, 3), host arrays."""
        H, W = self.height, self.width
        z, nrm, alb = np.zeros((H, W), np.float32), np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
        _check(lib().bhrt_first_hit(self._h, _ptr(z), _ptr(nrm), _ptr(alb)))
        return z, nrm, alb

    def first_hit_dev(self, d_z: int = 0, d_normal: int = 0, d_albedo: int = 0, stream: int = 0):
        _check(lib().bhrt_first_hit_dev(self._h, C.c_void_p(d_z or None), C.c_void_p(d_normal or None), C.c_void_p(d_albedo or None),
                                        C.c_void_p(stream or None)))

    def zbuffer_image_dev(self, d_z: int, n: int, d_img: int, stream: int = 0):
        """RenderImage::ComputeZBufferImage (scene.h:578-600) on device buffers."""
        _check(lib().bhrt_zbuffer_image_dev(self._h, C.c_void_p(d_z), C.c_size_t(n), C.c_void_p(d_img), C.c_void_p(stream or None)))

    def color_image_dev(self, d_radiance: int, n_pixels: int, gamma: int, d_color: int, stream: int = 0):
        """colorArray of BeginRender (Main.cpp:219-229): the gamma-corrected float image DenoiseImage is given."""
        _check(lib().bhrt_color_image_dev(self._h, C.c_void_p(d_radiance), C.c_size_t(n_pixels), int(gamma), C.c_void_p(d_color),
                                          C.c_void_p(stream or None)))

    # ---- caustic photon map ------------------------------------------------------------------
    def photon_build(self, opts: Opts, max_photons: int) -> int:
        n = C.c_uint32(0)
        _check(lib().bhrt_photon_build(self._h, C.byref(opts), int(max_photons), C.byref(n)))
        return n.value

    def knob(self, name: str, value: int):
        """Knobs ("frame_cap", "gather_lane_budget", "gather_stats", "shadow_overlap"): which internal path a render takes, never its result (bhrt_scene_knob)."""
        _check(lib().bhrt_scene_knob(self._h, name.encode(), int(value)))

    def photon_get(self) -> np.ndarray:
        """Balanced (heap-order) photon array as (n, 24) uint8 records (cyPhotonMap.h:72-90)."""
        n = C.c_uint32(0)
        _check(lib().bhrt_photon_get(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 24), np.uint8)
        _check(lib().bhrt_photon_get(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def photon_gather(self, p, nrm, radius=0.5, exact=False):
        """EstimateIrradiance<1000> for n points; exact: bhrt_opts.photon_exact (the reference's heap history replayed for heavy queries)."""
        irr, d, _, _, _ = self.photon_gather_ex(p, nrm, radius, exact=exact, want_knn=False)
        return irr, d

    def photon_gather_ex(self, p, nrm, radius=0.5, exact=False, want_knn=True):
        """photon_gather with the choice of bhrt_opts.photon_exact; returns (irr, dir, knn (cnt, 1000) uint32, knn_count (cnt,), d2max (cnt,))."""
        p = np.ascontiguousarray(p, np.float32)
        nrm = np.ascontiguousarray(nrm, np.float32)
        irr, d = np.zeros_like(p), np.zeros_like(p)
        cnt = p.shape[0]
        knn = np.zeros((cnt, 1000), np.uint32) if want_knn else None
        kc, dm = np.zeros(cnt, np.uint32), np.zeros(cnt, np.float32)
        _check(lib().bhrt_photon_gather_host_ex(self._h, _ptr(p), _ptr(nrm), C.c_size_t(cnt), C.c_float(radius), 1 if exact else 0, _ptr(irr), _ptr(d),
                                                _ptr(knn) if want_knn else None, _ptr(kc), _ptr(dm)))
        return irr, d, knn, kc, dm

    def photon_emit_range(self, opts: Opts, e0: int, count: int, global_map: bool = False, capacity: int = 0) -> np.ndarray:
        """Emissions [e0, e0 + count): the photons they store, emission order, unscaled power, (n, 24) uint8 (multi-GPU build)."""
        capacity = capacity or 16 * count
        out = np.zeros((capacity, 24), np.uint8)
        n = C.c_uint32(0)
        _check(lib().bhrt_photon_emit_range(self._h, C.byref(opts), int(bool(global_map)), C.c_uint64(e0), int(count), _ptr(out), int(capacity), C.byref(n)))
        return out[: n.value].copy()

    def photon_emit_range_into(self, opts: Opts, e0: int, count: int, out_ptr: int, capacity: int, global_map: bool = False):
        """Same, records written to `out_ptr` (host or device memory, room for `capacity` records).  Returns (n, ok): ok is False when
        n > capacity (nothing usable was written; call again with room for n)."""
        n = C.c_uint32(0)
        rc = lib().bhrt_photon_emit_range(self._h, C.byref(opts), int(bool(global_map)), C.c_uint64(e0), int(count), C.c_void_p(out_ptr), int(capacity), C.byref(n))
        if rc != 0 and n.value <= capacity:
            _check(rc)
        return n.value, rc == 0

    def photon_install_ptr(self, ptr: int, n: int) -> int:
        """Emission-order records at `ptr` (host or device memory) -> scaled, balanced and installed caustic map."""
        _check(lib().bhrt_photon_install(self._h, C.c_void_p(ptr), int(n)))
        return n

    def photon_install(self, records: np.ndarray) -> int:
        """Emission-order records (n, 24) -> scaled, balanced and installed caustic map."""
        r = np.ascontiguousarray(records, np.uint8).reshape(-1, 24)
        _check(lib().bhrt_photon_install(self._h, _ptr(r), len(r)))
        return len(r)

    def photon_build_global(self, opts: Opts, max_photons: int, dat_path=None) -> np.ndarray:
        """BuildPhotonMap (Main.cpp:251-295): the global photon map, balanced (n, 24) uint8 records."""
        out = np.zeros((max_photons, 24), np.uint8)
        n = C.c_uint32(0)
        _check(lib().bhrt_photon_build_global(self._h, C.byref(opts), int(max_photons), _ptr(out), int(max_photons), C.byref(n),
                                              os.fsencode(dat_path) if dat_path else None))
        return out[: n.value].copy()

    def photon_export(self, path: str):
        _check(lib().bhrt_photon_export(self._h, os.fsencode(path)))

    def photon_import(self, path: str, rebalance: bool = False):
        """Loads a .dat of 24-byte records; rebalance=True mirrors PhotonMap::InitializePhotonMapByFile."""
        _check(lib().bhrt_photon_import(self._h, os.fsencode(path), 1 if rebalance else 0))

    def render_samples(self, opts: Opts, x0, y0, x1, y1):
        out = np.zeros(((y1 - y0) * (x1 - x0), opts.spp, 3), np.float32)
        st = Stats()
        _check(lib().bhrt_render_samples(self._h, C.byref(opts), x0, y0, x1, y1, _ptr(out), C.byref(st)))
        return out, st


def math_eval_dev(fn: int, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    out = np.empty_like(a)
    _check(lib().bhrt_math_eval_dev(int(fn), _ptr(a), _ptr(bb) if bb is not None else None, C.c_size_t(a.size), _ptr(out)))
    return out


def tiles_block_bytes(width: int, height: int, tile: int, world: int) -> int:
    f = lib().bhrt_tiles_block_bytes
    f.restype = C.c_size_t
    return int(f(width, height, tile, world))


def bvh_build(vertices, faces, max_per_leaf=4, device=0):
    """cyBVH::Build on the device (bhrt_bvh_build): returns (nodes (n+1, 8) uint32 words of bhrt_bvh_node incl. the unused
    slot 0, element order (n_faces,) uint32, depth)."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
    nodes = np.zeros((2 * len(f) + 1, 8), np.uint32)
    elems = np.zeros(len(f), np.uint32)
    n, depth = C.c_uint32(0), C.c_uint32(0)
    _check(lib().bhrt_bvh_build(_ptr(v), len(v), _ptr(f), len(f), int(max_per_leaf), int(device), _ptr(nodes), len(nodes), C.byref(n), _ptr(elems),
                                C.byref(depth)))
    return nodes[: n.value + 1].copy(), elems, depth.value


def tiles_pack_dev(rgb8_ptr: int, radiance_ptr: int, width, height, tile, rank, world, block_ptr: int, stream: int = 0):
    _check(lib().bhrt_tiles_pack_dev(C.c_void_p(rgb8_ptr), C.c_void_p(radiance_ptr), width, height, tile, rank, world, C.c_void_p(block_ptr),
                                     C.c_void_p(stream)))


def tiles_unpack_dev(blocks_ptr: int, width, height, tile, world, rgb8_ptr: int, radiance_ptr: int, stream: int = 0):
    _check(lib().bhrt_tiles_unpack_dev(C.c_void_p(blocks_ptr), width, height, tile, world, C.c_void_p(rgb8_ptr), C.c_void_p(radiance_ptr),
                                       C.c_void_p(stream)))


def save_png(path: str, rgb8: np.ndarray):
    a = np.ascontiguousarray(rgb8, np.uint8)
    _check(lib().bhrt_save_png(os.fsencode(path), _ptr(a), a.shape[1], a.shape[0]))
