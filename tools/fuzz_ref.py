#!/usr/bin/env python3
"""Randomised oracle-vs-REFERENCE parity (development container only: needs /root/reference compiled into oracle/_ref):
the random scenes of tools/fuzz_parity.py rendered by the unmodified reference (rand() interposed by the sequential
stream) and by the oracle in sequential / libm mode; per-sample radiance, RGB8 and the scene dump compared bit for bit.
Usage: python tools/fuzz_ref.py [n_scenes] [first_seed]"""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    import bhraytracer_amd as B
    import oracle_lib as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = tempfile.mkdtemp()
    shutil.copy(os.path.join(ROOT, "tests", "scenes", "mesh_small.obj"), tmp)
    bad = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        xml = f"fuzz_{seed}.xml"
        fuzz_parity.random_scene(rng, os.path.join(tmp, xml))
        gi = int(rng.integers(0, 4))
        pre = os.path.join(tmp, f"r{seed}")
        r = subprocess.run([HARNESS, xml, pre, "--spp", "2", "--gi", str(gi), "--seed", str(seed), "--region", "0", "0", "96", "72", "render"], cwd=tmp,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode != 0:
            print(f"seed {seed}: reference exited with {r.returncode} (crash in the reference: skipped)", flush=True)
            continue
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            sc = B.Scene(xml)
        finally:
            os.chdir(cwd)
        o = O.render(sc.flat_bytes(), sc.width, sc.height, 2, gi=gi, seed=seed, rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM, region=(0, 0, 96, 72), threads=4)
        ref = np.fromfile(pre + ".samples_f32", np.float32).reshape(96 * 72, 2, 3)
        a, b = o["samples"].view(np.uint32), ref.view(np.uint32)
        same = (a == b) | (np.isnan(o["samples"]) & np.isnan(ref))
        rgb_ok = np.array_equal(o["rgb8"].reshape(-1, 3), np.fromfile(pre + ".rgb8", np.uint8).reshape(-1, 3))
        ok = bool(same.all()) and rgb_ok
        print(f"seed {seed}: gi {gi}: {'identical' if ok else f'MISMATCH ({int((~same).sum())} values, rgb8 {rgb_ok})'}", flush=True)
        bad += not ok
    print("all identical" if not bad else f"{bad} scene(s) differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
