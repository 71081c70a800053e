#!/usr/bin/env python3
"""Deterministic procedural OBJ: a displaced UV sphere of n x n quads (2*n*n triangles).

No mesh ships with the reference (its .gitignore drops *.obj), so the mesh
configs of BASELINE.json use this generator: n=224 gives the 100,352-triangle mesh of config 3.
The file carries `vt` lines (the reference dereferences null without them) and
quad faces, so the loader's fan triangulation (cyTriMesh.h:379-438) is exercised.  No RNG.
"""
import math
import os
import sys


def generate(path: str, n: int) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write("# bhrt procedural mesh: displaced UV sphere, %d x %d quads\n" % (n, n))
        for j in range(n + 1):
            theta = math.pi * j / n
            for i in range(n + 1):
                phi = 2.0 * math.pi * i / n
                r = 1.0 + 0.12 * math.sin(5.0 * theta) * math.cos(4.0 * phi) + 0.05 * math.cos(9.0 * phi + 3.0 * theta)
                x = r * math.sin(theta) * math.cos(phi)
                y = r * math.sin(theta) * math.sin(phi)
                z = r * math.cos(theta)
                f.write("v %.6f %.6f %.6f\n" % (x, y, z))
        for j in range(n + 1):
            for i in range(n + 1):
                f.write("vt %.6f %.6f 0.000000\n" % (i / n, 1.0 - j / n))
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i + 1
                b = a + 1
                c = a + (n + 1) + 1
                d = a + (n + 1)
                f.write("f %d/%d %d/%d %d/%d %d/%d\n" % (a, a, d, d, c, c, b, b))
    os.replace(tmp, path)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: gen_mesh.py out.obj n")
    generate(sys.argv[1], int(sys.argv[2]))
