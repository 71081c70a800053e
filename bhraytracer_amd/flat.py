"""ctypes mirror of include/bhrt_flat.h: read-only views into the flattened scene blob.

Host-side plumbing for tests, bench and tools; the kernels read the same bytes in HBM.
"""
import ctypes as C

import numpy as np

MAGIC = 0x54524842
VERSION = 10
BIGFLOAT = np.float32(1.0e30)

OBJ_NONE, OBJ_SPHERE, OBJ_PLANE, OBJ_MESH = 0, 1, 2, 3
LIGHT_AMBIENT, LIGHT_DIRECT, LIGHT_POINT = 0, 1, 2


class Xform(C.Structure):
    _fields_ = [("tm", C.c_float * 9), ("pos", C.c_float * 3), ("itm", C.c_float * 9)]


class Node(C.Structure):
    _fields_ = [("xf", Xform), ("parent", C.c_int32), ("depth", C.c_int32), ("obj_type", C.c_int32),
                ("mesh", C.c_int32), ("material", C.c_int32), ("subtree_end", C.c_int32), ("pad", C.c_int32 * 5)]


class BvhNode(C.Structure):
    _fields_ = [("b", C.c_float * 6), ("data", C.c_uint32), ("parent", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("nv", C.c_uint32), ("nf", C.c_uint32), ("nvn", C.c_uint32), ("nvt", C.c_uint32),
                ("n_bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("off_v", C.c_uint64), ("off_vn", C.c_uint64), ("off_vt", C.c_uint64),
                ("off_f", C.c_uint64), ("off_fn", C.c_uint64), ("off_ft", C.c_uint64),
                ("off_bvh", C.c_uint64), ("off_elems", C.c_uint64), ("off_tris", C.c_uint64), ("off_dbvh", C.c_uint64), ("off_leaf_tris", C.c_uint64),
                ("bound_min", C.c_float * 3), ("bound_max", C.c_float * 3), ("bvh_nested", C.c_uint32), ("pad0", C.c_uint32),
                ("off_dparent", C.c_uint64), ("skip_k0", C.c_float), ("skip_k1", C.c_float), ("skip_big", C.c_float), ("skip_omax", C.c_float)]


class TexMap(C.Structure):
    _fields_ = [("xf", Xform), ("texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("color1", C.c_float * 3), ("color2", C.c_float * 3), ("off_data", C.c_uint64)]


class TexColor(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("map", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("diffuse", TexColor), ("specular", TexColor), ("refraction", TexColor),
                ("glossiness", C.c_float), ("absorption", C.c_float * 3), ("ior", C.c_float),
                ("refraction_glossiness", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float * 3), ("vec", C.c_float * 3), ("size", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float), ("focaldist", C.c_float), ("dof", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32),
                ("top_left", C.c_float * 3), ("dd_x", C.c_float * 3), ("dd_y", C.c_float * 3)]


class Header(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("total_bytes", C.c_uint64),
                ("n_nodes", C.c_uint32), ("n_meshes", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_texmaps", C.c_uint32), ("n_textures", C.c_uint32),
                ("off_nodes", C.c_uint64), ("off_meshes", C.c_uint64), ("off_materials", C.c_uint64),
                ("off_lights", C.c_uint64), ("off_texmaps", C.c_uint64), ("off_textures", C.c_uint64),
                ("camera", Camera), ("background", TexColor), ("environment", TexColor),
                ("all_light_intensity", C.c_float), ("max_node_depth", C.c_uint32), ("reserved", C.c_uint32 * 6)]


class FlatView:
    """Parsed view of a flat scene blob (bytes-like). Keeps the buffer alive."""

    def __init__(self, blob: bytes):
        self.buf = bytes(blob)
        self.header = Header.from_buffer_copy(self.buf[:C.sizeof(Header)])
        if self.header.magic != MAGIC or self.header.version != VERSION:
            raise ValueError("not a bhrt flat scene blob")
        h = self.header
        self.nodes = self._array(Node, h.off_nodes, h.n_nodes)
        self.meshes = self._array(Mesh, h.off_meshes, h.n_meshes)
        self.materials = self._array(Material, h.off_materials, h.n_materials)
        self.lights = self._array(Light, h.off_lights, h.n_lights)
        self.texmaps = self._array(TexMap, h.off_texmaps, h.n_texmaps)
        self.textures = self._array(Texture, h.off_textures, h.n_textures)

    def _array(self, typ, off, n):
        size = C.sizeof(typ)
        return [typ.from_buffer_copy(self.buf[off + i * size: off + (i + 1) * size]) for i in range(n)]

    def np(self, off, count, dtype):
        return np.frombuffer(self.buf, dtype=dtype, count=count, offset=off)

    def mesh_arrays(self, i):
        m = self.meshes[i]
        bvh = self.np(m.off_bvh, m.n_bvh_nodes * 8, np.uint32).reshape(-1, 8)
        return {
            "v": self.np(m.off_v, m.nv * 3, np.float32).reshape(-1, 3),
            "vn": self.np(m.off_vn, m.nvn * 3, np.float32).reshape(-1, 3),
            "vt": self.np(m.off_vt, m.nvt * 3, np.float32).reshape(-1, 3),
            "f": self.np(m.off_f, m.nf * 3, np.uint32).reshape(-1, 3),
            "fn": self.np(m.off_fn, m.nf * 3, np.uint32).reshape(-1, 3),
            "ft": self.np(m.off_ft, m.nf * 3, np.uint32).reshape(-1, 3),
            "bvh_bounds": bvh[:, :6].copy().view(np.float32),
            "bvh_data": bvh[:, 6].copy(),
            "bvh_parent": bvh[:, 7].copy(),
            "bvh_raw": bvh,
            "elems": self.np(m.off_elems, m.nf, np.uint32),
            "dbvh_raw": self.np(m.off_dbvh, m.n_bvh_nodes * 8, np.uint32).reshape(-1, 8),
        }
