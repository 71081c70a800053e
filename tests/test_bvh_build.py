"""cyBVH::Build on the device (DataStructure/cyBVH.h:122-142,242-328).

The device build (bvh_build.hip, level-parallel) has to give the reference's tree node for node.  The checker is the ORACLE's
restatement of cyBVH::Build (oracle/bhrt_oracle.cpp::oracle_bvh_build, pinned against the compiled reference's own BVH dumps in
tests/test_ref_parity.py): node ids, boxes incl. the sign of zeros, leaf ranges, parent links and element order of the device build
are compared with it word for word — for the mesh scenes and for 24 adversarial triangle soups.  The product's own host build
(scene_host.cpp::BuildBvh, the recursive MeanSplit) is only the second opinion: the whole flattened scene of a host-built and a
device-built load must be the same bytes (breadth-first copy, leaf-ordered triangle records and depth included)."""
import os

import numpy as np
import pytest

XML = """<xml><scene>
  <object type="obj" name="{obj}" material="m"><scale value="2"/></object>
  <material type="blinn" name="m"><diffuse r="0.5" g="0.5" b="0.5"/></material>
  <light type="point" name="l"><intensity value="1"/><position x="0" y="0" z="10"/></light>
</scene>
<camera><position x="0" y="-10" z="3"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="40"/><width value="32"/><height value="24"/></camera>
</xml>"""


def write_obj(path, v, f):
    with open(path, "w") as fp:
        for p in v:
            fp.write("v %s %s %s\n" % tuple(repr(float(x)) for x in p))
        fp.write("vt 0 0 0\nvn 0 0 1\n")
        for t in f:
            fp.write("f %d/1/1 %d/1/1 %d/1/1\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def soups():
    """Triangle soups that reach every branch of the build: ties, both zeros, axes on which nothing separates (the rotation
    by one), forced halving, single leaves."""
    out = []
    for seed in range(24):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 4, 5, 9, 17, 64, 300, 1500, 5000]))
        kind = seed % 6
        if kind == 0:    # uniform cloud of small triangles
            c = rng.uniform(-5, 5, (n, 1, 3)); tri = c + rng.normal(scale=0.2, size=(n, 3, 3))
        elif kind == 1:  # quantised coordinates: many equal centres and box faces, zeros of both signs
            tri = rng.integers(-2, 3, (n, 3, 3)).astype(np.float64) * 0.5
            tri[rng.random(tri.shape) < 0.1] = -0.0
        elif kind == 2:  # every triangle the same: no axis separates anything
            tri = np.repeat(rng.normal(size=(1, 3, 3)), n, axis=0)
        elif kind == 3:  # one huge triangle, everything else on its far side: all centres right of the midpoint
            tri = np.array([90.0, 0, 0]) + rng.normal(scale=0.5, size=(n, 3, 3))
            tri[0] = [[0, 0, 0], [100, 0, 0], [100, 1, 0]]
        elif kind == 4:  # flat in two axes
            tri = np.zeros((n, 3, 3)); tri[..., 0] = np.sort(rng.uniform(0, 10, (n, 3)), axis=1)
        else:            # two far clusters + shared vertices
            c = np.where(rng.random((n, 1, 1)) < 0.5, -50.0, 50.0) * np.array([1, 0, 0]); tri = c + rng.normal(size=(n, 3, 3))
        v = tri.reshape(-1, 3).astype(np.float32)
        f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        if kind == 5:
            f[1:, 0] = f[:-1, 2]
        out.append((f"soup{seed}_{n}", v, f))
    return out


@pytest.mark.gpu
def test_device_bvh_equals_the_oracle_on_scenes(B, O):
    from conftest import SCENES, ensure_mesh
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    ensure_mesh(224)
    for name in ("c3_mesh_small", "c3_mesh", "c3_room"):
        path = os.path.join(SCENES, name + ".xml")
        dev = B.Scene(path, bvh_device=0)
        m = dev.flat_view().mesh_arrays(0)
        nodes, elems = O.bvh_build(m["v"], m["f"], 4)                        # cyBVH.h:122-142,242-328 restated in oracle/
        assert nodes.shape[0] == m["bvh_raw"].shape[0] == dev.info.n_bvh_nodes, name
        assert np.array_equal(nodes[1:], m["bvh_raw"][1:]), name             # bounds, data word and parent link of every node
        assert np.array_equal(elems, m["elems"]), name
        dn, de, depth = B.bvh_build(m["v"], m["f"])                          # the bare entry point on the same arrays
        assert np.array_equal(dn[1:], nodes[1:]) and np.array_equal(de, elems) and depth == dev.info.max_bvh_depth
        host = B.Scene(path)                                                 # second opinion: the product's host build, whole blob
        assert host.flat_bytes() == dev.flat_bytes(), name
    assert dev.info.n_triangles > 100000


@pytest.mark.gpu
def test_device_bvh_equals_the_oracle_on_triangle_soups(B, O, tmp_path):
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    branches = set()
    for name, v, f in soups():
        write_obj(tmp_path / (name + ".obj"), v, f)
        xml = tmp_path / (name + ".xml")
        xml.write_text(XML.format(obj=name + ".obj"))
        host, dev = B.Scene(str(xml)), B.Scene(str(xml), bvh_device=0)
        assert host.info.n_triangles == len(f)
        assert host.flat_bytes() == dev.flat_bytes(), name
        nodes, elems, depth = B.bvh_build(v, f)                              # the bare entry point
        assert len(nodes) == host.info.n_bvh_nodes and depth == host.info.max_bvh_depth and sorted(elems) == list(range(len(f)))
        on, oe = O.bvh_build(v, f, 4)                                        # the checker: the oracle's cyBVH::Build
        assert np.array_equal(nodes[1:], on[1:]) and np.array_equal(elems, oe), name
        dm = dev.flat_view().mesh_arrays(0)
        assert np.array_equal(dm["bvh_raw"][1:], on[1:]) and np.array_equal(dm["elems"], oe), name
        branches.add((len(f) <= 4, depth > 0))
    assert len(branches) >= 2
    with pytest.raises(B.BhrtError):
        B.bvh_build(np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32))   # index out of range
