#!/usr/bin/env python3
"""Workload of the PMC passes (tools/profile_round.sh): a calibration launch with a KNOWN byte count in this path's own
access pattern (one dword per lane, SoA streams: k_math_eval reads two float arrays and writes one), then ONE frame of a
bench workload.  FETCH_SIZE / WRITE_SIZE of the calibration launch give the counter-to-bytes factor for that pattern
(MI355X_MICROARCH.md: only 16-B-per-lane streaming is calibrated; anything else must be calibrated on a known count)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (same HIP runtime as bench.py)
import bhraytracer_amd as B
import bench

CAL_N = 1 << 27  # 128 Mi floats per array: 512 MiB read x2 + 512 MiB written, past the 256 MiB Infinity Cache


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    a = np.ones(CAL_N, np.float32); b = np.full(CAL_N, 3.0, np.float32)
    B.math_eval_dev(8, a, b)  # k_math_eval: out[i] = a[i] / b[i]
    scene_rel, W, H, spp, gi = bench.WORKLOADS[wl]
    sc = B.Scene(os.path.join(ROOT, scene_rel)); sc.upload(0)
    opts = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0)
    if wl == "c5":  # BASELINE config 5: the caustic map first (its kernels appear in the counter files as well)
        sc.photon_build(opts, 1000000)
        opts.photon_map = 1
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    st = sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())
    torch.cuda.synchronize()
    print("frame:", st.as_dict())


if __name__ == "__main__":
    main()
