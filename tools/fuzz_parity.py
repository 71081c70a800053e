#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity: seeded random scenes (spheres, planes, instances of tests/scenes/mesh_small.obj, nested
transformed groups, Blinn materials with and without refraction / absorption / checker textures, ambient + direct + point
lights with and without size), small images, per-sample radiance and hits compared bit for bit.
Usage: python tools/fuzz_parity.py [n_scenes] [first_seed] [leaf_skip]        (needs a GPU; the oracle is the checker; leaf_skip: bhrt_opts.leaf_skip = 1)"""
import os, shutil, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_scene(rng, path):
    def f(a, b): return f"{rng.uniform(a, b):.3f}"
    mats = []
    for k in range(4):
        refr = rng.random() < 0.35
        tex = rng.random() < 0.3
        d = f'<diffuse r="{f(0.1, 0.9)}" g="{f(0.1, 0.9)}" b="{f(0.1, 0.9)}"' + (' texture="checkerboard"><color1 value="0.2"/><color2 value="0.9"/><scale x="0.25" y="0.25"/></diffuse>' if tex else '/>')
        m = f'<material type="blinn" name="m{k}">{d}<specular value="{f(0.0, 0.9)}"/><glossiness value="{f(2, 120)}"/>'
        if refr:
            m += f'<refraction value="{f(0.5, 0.95)}" index="{f(1.1, 2.0)}"/><absorption r="{f(0, 0.2)}" g="{f(0, 0.2)}" b="{f(0, 0.2)}"/>'
        mats.append(m + '</material>')

    def obj(depth):
        kind = rng.choice(["sphere", "plane", "obj", "group"], p=[0.4, 0.2, 0.2, 0.2])
        xf = ""
        if rng.random() < 0.8: xf += f'<scale x="{f(0.5, 3)}" y="{f(0.5, 3)}" z="{f(0.5, 3)}"/>'
        if rng.random() < 0.6: xf += f'<rotate angle="{f(-90, 90)}" x="{f(-1, 1)}" y="{f(-1, 1)}" z="{f(0.1, 1)}"/>'
        xf += f'<translate x="{f(-6, 6)}" y="{f(-4, 8)}" z="{f(0, 5)}"/>'
        mat = f'material="m{rng.integers(0, 4)}"'
        if kind == "group" and depth < 2:
            inner = "".join(obj(depth + 1) for _ in range(rng.integers(1, 3)))
            return f'<object name="g">{xf}{inner}</object>'
        if kind == "obj": return f'<object type="obj" name="mesh_small.obj" {mat}>{xf}</object>'
        if kind == "plane": return f'<object type="plane" name="p" {mat}><scale value="{f(3, 8)}"/><rotate angle="{f(-30, 30)}" x="1"/><translate x="{f(-3, 3)}" y="{f(0, 6)}" z="{f(0, 3)}"/></object>'
        return f'<object type="sphere" name="s" {mat}>{xf}</object>'

    objs = '<object type="plane" name="ground" material="m0"><scale value="30"/></object>' + "".join(obj(0) for _ in range(rng.integers(3, 7)))
    lights = f'<light type="point" name="p"><intensity value="{f(100, 400)}"/><position x="{f(-8, 8)}" y="{f(-12, -4)}" z="{f(10, 20)}"/>' + (f'<size value="{f(0.5, 2.5)}"/>' if rng.random() < 0.6 else '') + '</light>'
    if rng.random() < 0.5: lights += f'<light type="ambient" name="a"><intensity value="{f(0.05, 0.3)}"/></light>'
    if rng.random() < 0.4: lights += f'<light type="direct" name="d"><intensity value="{f(0.2, 0.8)}"/><direction x="{f(-1, 1)}" y="{f(0.2, 1)}" z="{f(-1, -0.2)}"/></light>'
    xml = (f'<xml><scene><background r="{f(0, 0.3)}" g="{f(0, 0.3)}" b="{f(0, 0.3)}"/><environment value="{f(0.2, 0.8)}"/>{objs}{"".join(mats)}{lights}</scene>'
           f'<camera><position x="{f(-3, 3)}" y="-25" z="{f(4, 12)}"/><target x="0" y="0" z="2"/><up z="1"/><fov value="{f(25, 50)}"/><width value="96"/><height value="72"/></camera></xml>')
    open(path, "w").write(xml)


def check(seed, B, O, tmp):
    rng = np.random.default_rng(seed)
    path = os.path.join(tmp, f"fuzz_{seed}.xml")
    random_scene(rng, path)
    sc = B.Scene(path)
    blob = sc.flat_bytes()
    o, d = O.primary_rays(sc.flat_view())
    ok = True
    for side in (1, 3):
        h, r = sc.trace_closest(o, d, side), O.trace_closest(blob, o, d, side)
        ok &= np.array_equal(h["node"], r["node"]) and np.array_equal(h["prim"], r["prim"]) and np.array_equal(h["t"].view(np.uint32), r["t"].view(np.uint32))
    spp, gi = 2, int(rng.integers(0, 4))
    opts = B.default_opts(spp=spp, gi_bounces=gi, seed=seed)
    opts.leaf_skip = 1 if "leaf_skip" in sys.argv[1:] else 0
    gs, _ = sc.render_samples(opts, 0, 0, sc.width, sc.height)
    ro = O.render(blob, sc.width, sc.height, spp, gi=gi, seed=seed, region=(0, 0, sc.width, sc.height))["samples"]
    same = (gs.view(np.uint32) == ro.view(np.uint32)) | (np.isnan(gs) & np.isnan(ro))
    ok &= bool(same.all())
    return ok, int((~same).sum()), sc.info.n_nodes, gi


def main():
    import bhraytracer_amd as B
    import oracle_lib as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = tempfile.mkdtemp()
    shutil.copy(os.path.join(ROOT, "tests", "scenes", "mesh_small.obj"), tmp)
    bad = 0
    for seed in range(s0, s0 + n):
        ok, nbad, nn, gi = check(seed, B, O, tmp)
        print(f"seed {seed}: nodes {nn} gi {gi}: {'identical' if ok else f'MISMATCH ({nbad} values)'}", flush=True)
        bad += not ok
    print("all identical" if not bad else f"{bad} scene(s) differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
