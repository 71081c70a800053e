"""Builds libbhrt.so (the C-ABI library: host front-end + gfx950 HIP kernels) in-tree.

    python -m bhraytracer_amd.build            # or: from bhraytracer_amd.build import build; build()

Host TUs are compiled with g++, kernels.hip with hipcc (--offload-arch=gfx950); both with
-ffp-contract=off — bit-exact parity with the reference needs unfused IEEE single operations.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libbhrt.so")
CLI = os.path.join(HERE, "bhrt")

HOST_SRCS = ["scene_host.cpp", "png_io.cpp", "capi_host.cpp", "photon_host.cpp"]
HIP_SRCS = ["kernels.hip", "bvh_build.hip", "gather_sort.hip"]
COMMON = ["-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OUT, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    objs = []
    for s in HOST_SRCS:
        src, obj = os.path.join(CSRC, s), os.path.join(OUT, s + ".o")
        if force or _newer([src] + headers, obj):
            _run(["g++"] + COMMON + ["-Wall", "-c", src, "-o", obj])
        objs.append(obj)
    for s in HIP_SRCS:
        src, obj = os.path.join(CSRC, s), os.path.join(OUT, s + ".o")
        if force or _newer([src] + headers, obj):
            _run([hipcc, "--offload-arch=gfx950"] + COMMON + ["-Wno-unused-result", "-c", src, "-o", obj])
        objs.append(obj)
    if force or _newer(objs, LIB):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", LIB])
    main_src = os.path.join(CSRC, "bhrt_main.cpp")
    if force or _newer([main_src, LIB] + headers, CLI):  # the C++ host program above the C ABI (Main.cpp:418-431)
        # host-only program (no device code): hipcc for the HIP runtime headers, linked against the C-ABI library and RCCL
        _run([hipcc, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-result", main_src, "-L" + HERE, "-lbhrt", "-L/opt/rocm/lib", "-lrccl", "-lpthread",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + "/opt/rocm/lib", "-o", CLI])
    return LIB


def build_variant(name: str, defines) -> str:
    """Experiment aid: the same library with extra -D flags on the HIP TUs, as bhraytracer_amd/_variants/libbhrt_<name>.so
    (load it with BHRT_LIB=<path>; A/B runs of one kernel change on the GPU box).  Not part of build()."""
    vdir = os.path.join(HERE, "_variants")
    os.makedirs(vdir, exist_ok=True)
    build(verbose=False)  # host objects
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = [os.path.join(OUT, s + ".o") for s in HOST_SRCS]
    for s in HIP_SRCS:
        obj = os.path.join(vdir, f"{name}_{s}.o")
        _run([hipcc, "--offload-arch=gfx950"] + COMMON + ["-Wno-unused-result", "-Wno-pass-failed"] + list(defines) + ["-c", os.path.join(CSRC, s), "-o", obj])
        objs.append(obj)
    lib = os.path.join(vdir, f"libbhrt_{name}.so")
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", lib])
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        k = sys.argv.index("--variant")
        build_variant(sys.argv[k + 1], sys.argv[k + 2:])
    else:
        build(force="--force" in sys.argv)
