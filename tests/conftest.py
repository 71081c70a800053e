import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "tests", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/BHRayTracer"
REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLDEN_CASES = ["c1_sphere_plane", "c2_glass_small", "c3_mesh_small", "c4_textured", "c3_room_small"]
# BASELINE's full sizes, pinned by the compiled reference as well: the closed-room mesh scene at 1920x1080 and
# BASELINE config 4's 3840x2160 frame; both use the 100,352-triangle mesh that tools/gen_mesh.py writes on demand
FULL_SIZE_CASES = ["c3_room", "c4_mesh_4k"]


def ensure_mesh(n=224):
    """tests/scenes/gen/mesh_<n>.obj (deterministic generator, no RNG; git-ignored)."""
    path = os.path.join(SCENES, "gen", f"mesh_{n}.obj")
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_mesh
        gen_mesh.generate(path, n)
    return path


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """libbhrt.so (C-ABI library) and liboracle.so (checker) are built once per session if missing/stale."""
    from bhraytracer_amd.build import build
    build()
    import oracle_lib
    oracle_lib.build()


@pytest.fixture(scope="session")
def B():
    import bhraytracer_amd
    return bhraytracer_amd


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    return oracle_lib


_scene_cache = {}


@pytest.fixture(scope="session")
def load_scene(B):
    def _load(name):
        path = name if os.path.isabs(name) else os.path.join(SCENES, name if name.endswith(".xml") else name + ".xml")
        if path not in _scene_cache:
            if "gen/mesh_224.obj" in open(path).read():
                ensure_mesh(224)
            _scene_cache[path] = B.Scene(path)
        return _scene_cache[path]
    return _load


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return _load


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def have_gpu():
    try:
        import bhraytracer_amd
        return bhraytracer_amd.device_count() > 0
    except Exception:
        return False
