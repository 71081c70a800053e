import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bhraytracer_amd as B, oracle_lib as O
xml = "/tmp/deep.xml"
open(xml, "w").write("""<xml><scene>
  <object name="a"><translate x="1" z="2"/><rotate angle="30" z="1"/>
    <object name="b" type="sphere" material="m"><scale value="2"/><translate y="1"/>
      <object name="c" type="sphere" material="m"><scale x="0.5" y="0.7" z="0.4"/><translate x="2.5" z="1"/>
        <object name="d" type="plane" material="m"><scale value="3"/><rotate angle="70" x="1"/><translate z="-1"/></object>
      </object></object></object>
  <material type="blinn" name="m"/><light type="point" name="l"><intensity value="50"/><position z="15"/></light>
  </scene><camera><position y="-20" z="6"/><target z="2"/><up z="1"/><width value="160"/><height value="120"/></camera></xml>""")
sc = B.Scene(xml)
for gi in (-1, 0, 2):
    opts = B.default_opts(spp=2, gi_bounces=gi)
    gs, st = sc.render_samples(opts, 40, 30, 120, 90)
    ro = O.render(sc.flat_bytes(), sc.width, sc.height, 2, gi=gi, region=(40, 30, 120, 90))
    rs = ro["samples"]
    eq = (gs.view(np.uint32) == rs.view(np.uint32)) | (np.isnan(gs) & np.isnan(rs))
    bad = ~eq.all(axis=2)
    print("gi", gi, "bad", bad.sum(), "of", bad.size, "nan gpu", np.isnan(gs).sum(), "nan orc", np.isnan(rs).sum())
    for p, s in np.argwhere(bad)[:6]:
        print("  pix", p, (40 + p % 80, 30 + p // 80), "s", s, "gpu", gs[p, s], "orc", rs[p, s])
