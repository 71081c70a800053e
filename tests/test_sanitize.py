"""CPU sanitizer job (SURVEY.md 5 "Race detection / sanitizers": "CPU restatement built with -fsanitize=address,undefined in tests").

tests/sanitize/Makefile compiles the product's host front-end — scene_host.cpp (xmlload schema, OBJ / MTL rules of cyTriMesh.h:263-547, PPM
reader, BVH build, flattening, leaf-skip bounds), mini_xml.h, png_io.cpp, photon_host.cpp, capi_host.cpp — and the oracle (oracle/bhrt_oracle.cpp)
with AddressSanitizer + UBSan (-fno-sanitize-recover) into tests/sanitize/_build/san_driver; the two device entry points the host objects
reference are stubs (no GPU code in that build).  The driver loads scenes through the C ABI and runs the oracle on every blob that loads.
Inputs: every scene under tests/scenes/, the reference's own shipped XMLs where /root/reference exists, and a few hundred mutated XML / OBJ /
MTL / PNG / PPM files (truncated, bytes flipped, counts and indices huge / negative / zero / nan).  A malformed input must end in an error
code or a warning — never in a sanitizer report, which aborts the driver."""
import glob
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REFERENCE, ROOT, SCENES, ensure_mesh

SAN_DIR = os.path.join(ROOT, "tests", "sanitize")
DRIVER = os.path.join(SAN_DIR, "_build", "san_driver")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="1")


@pytest.fixture(scope="module")
def driver():
    subprocess.run(["make", "-s", "-j4", "-C", SAN_DIR], check=True, timeout=900)
    return DRIVER


def _run(driver, files, cwd, extra=()):
    r = subprocess.run([driver] + list(extra) + list(files), cwd=cwd, capture_output=True, text=True, errors="replace", timeout=600, env=ENV)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-1500:] + "\n" + r.stderr[-6000:])
    return r.stdout


def test_every_scene_of_the_repo_loads_clean_under_asan_and_ubsan(driver):
    ensure_mesh(224)
    files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(SCENES, "*.xml")))
    out = _run(driver, files, SCENES, extra=["--photons"])
    assert out.count(": ok ") == len(files) and "100352 triangles" in out


def test_the_reference_s_shipped_scenes_load_clean(driver):
    data = os.path.join(REFERENCE, "Resource", "Data")
    if not os.path.isdir(data):
        pytest.skip("the reference's own scene files are only in the development container")
    files = sorted(glob.glob(os.path.join(data, "*.xml")))
    out = _run(driver, files, REFERENCE)  # the reference resolves asset paths against its working directory
    assert len(files) >= 15 and out.count(": ok ") + out.count(": error ") == len(files) and out.count(": ok ") >= 15


def _mutations(rng, data: bytes, n, text):
    """n mutated copies of one input file."""
    out = []
    numbers = [b"-1", b"0", b"4294967296", b"2147483648", b"-2147483649", b"1e39", b"-1e39", b"nan", b"inf", b"99999999999999999999", b"0x7fffffff", b""]
    for k in range(n):
        b = bytearray(data)
        kind = k % 5
        if kind == 0 and len(b) > 2:                      # truncated
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:                                   # bytes flipped
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2 and text:                          # a numeric token replaced
            toks = [(m.start(), m.end()) for m in __import__("re").finditer(rb"-?\d+\.?\d*(e-?\d+)?", bytes(b))]
            if toks:
                for _ in range(int(rng.integers(1, 4))):
                    s, e = toks[int(rng.integers(0, len(toks)))]
                    b = bytearray(bytes(b[:s]) + numbers[int(rng.integers(0, len(numbers)))] + bytes(b[e:]))
                    toks = [(m.start(), m.end()) for m in __import__("re").finditer(rb"-?\d+\.?\d*(e-?\d+)?", bytes(b))]
                    if not toks:
                        break
        elif kind == 3:                                   # a slice duplicated or removed
            i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
            b = b[:i] + b[j:] if rng.random() < 0.5 else b[:j] + b[i:j] + b[j:]
        else:                                             # a run of one byte
            i = int(rng.integers(0, len(b)))
            b[i:i + int(rng.integers(1, 64))] = bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 64))
        out.append(bytes(b))
    return out


OBJ_CASES = [
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0 0\nvn 0 0 1\nf -1/-1/-1 -2/-2/-2 -3/-3/-3\n",            # negative = relative indices (cyTriMesh.h:379-438)
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 4 5 6 7 8\n",                                           # polygon fan over missing vertices
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0/0/0 0/0/0 0/0/0\n",                                         # index 0
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 4294967297/1/1 2/2/2 3/3/3\n",                                # wraps in 32 bits
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -4 -5 -2147483648\n",                                         # relative index before the first vertex
    "v nan nan nan\nv inf 0 0\nv 0 -inf 0\nvt 0 0\nvn 0 0 0\nf 1/1/1 2/1/1 3/1/1\n",            # non-finite coordinates: bounds, normals, leaf-skip bounds
    "v 1e38 1e38 1e38\nv -1e38 -1e38 -1e38\nv 1e38 -1e38 1e38\nvt 0 0\nf 1/1 2/1 3/1\n",       # products overflow
    "f 1 2 3\n" * 50,                                                                           # faces, no vertices
    "v 0 0 0\n" * 3 + "f 1 2 3\n" * 2000,                                                       # 2000 identical degenerate triangles: forced BVH halving
    "mtllib missing.mtl\nusemtl nothing\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1// 2// 3//\n",           # empty index fields, missing material library
    "v 0 0 0 1 2 3 4\nv 1\nv\nvt\nvn 1\nf\nf 1\nf 1 2\n",                                       # short and long records
    "",
]


def test_mutated_inputs_end_in_errors_not_in_sanitizer_reports(driver, tmp_path):
    rng = np.random.default_rng(20260105)
    for f in ("mesh_small.obj", "tex_small.png", "tex_small.ppm"):
        shutil.copy(os.path.join(SCENES, f), tmp_path / f)
    base_xml = open(os.path.join(SCENES, "c4_textured.xml"), "rb").read()
    files, pngs = [], []
    for k, m in enumerate(_mutations(rng, base_xml, 120, True)):            # the scene file itself
        (tmp_path / f"x{k}.xml").write_bytes(m)
        files.append(f"x{k}.xml")
    assets = [("mesh_small.obj", True, 70), ("tex_small.png", False, 70), ("tex_small.ppm", True, 50)]
    for name, text, n in assets:                                            # its assets, each mutated copy behind a scene of its own
        data = open(os.path.join(SCENES, name), "rb").read()
        stem, ext = os.path.splitext(name)
        for k, m in enumerate(_mutations(rng, data, n, text)):
            mut = f"{stem}_m{k}{ext}"
            (tmp_path / mut).write_bytes(m)
            (tmp_path / f"{stem}{ext[1:]}_m{k}.xml").write_bytes(base_xml.replace(name.encode(), mut.encode()))
            files.append(f"{stem}{ext[1:]}_m{k}.xml")
            if ext == ".png":
                pngs += ["--png", mut]
    for k, text in enumerate(OBJ_CASES):                                    # the OBJ index rules, by hand
        (tmp_path / f"case{k}.obj").write_text(text)
        (tmp_path / f"case{k}.xml").write_bytes(base_xml.replace(b"mesh_small.obj", f"case{k}.obj".encode()))
        files.append(f"case{k}.xml")
    hdr = lambda w, h, mx: f"P6\n{w} {h}\n{mx}\n".encode()                  # PPM headers that promise more than the file holds
    for k, (w, h, mx) in enumerate([(65536, 65536, 255), (-1, 4, 255), (4, 0, 255), (4, 4, 65535), (4, 4, 0), (2147483647, 2, 255), (1, 1, 255)]):
        (tmp_path / f"hdr{k}.ppm").write_bytes(hdr(w, h, mx) + b"\x10" * 48)
        (tmp_path / f"hdr{k}.xml").write_bytes(base_xml.replace(b"tex_small.ppm", f"hdr{k}.ppm".encode()))
        files.append(f"hdr{k}.xml")
    out = _run(driver, files, str(tmp_path), extra=pngs)
    n_ok, n_err = out.count(": ok "), out.count(": error ")
    assert n_ok + n_err == len(files) and len(files) > 300
    assert n_err > 20 and n_ok > 100   # both outcomes occur: rejected files, and damaged ones the loader accepts with warnings like the reference does
