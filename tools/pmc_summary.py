#!/usr/bin/env python3
"""Post-process gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>/ and profiles/pmc_traffic.json.

Traffic per launch = (FETCH_SIZE x f_read + WRITE_SIZE x f_write) x 1024 / launches, where the factors come from the
calibration launch inside the same pass (k_math_eval on 3 x 512 MiB of dword-per-lane streams, known byte counts)."""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAL_BYTES_READ = 2 * 4 * (1 << 27)
CAL_BYTES_WRITTEN = 4 * (1 << 27)


def short(name):
    if "rocprim" in name:  # the library radix sort of the gather's (cell, query) pairs (gather_sort.hip): counted with the cell sort
        return "k_gather_cell_radix_sort"
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("bhrt::", "")
    return n.split("<")[0]


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    wls = sys.argv[2:] or ["c2"]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    for wl in wls:
        for f in glob.glob(os.path.join(src, f"trace_{wl}", "**", f"{wl}_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"{wl}_kernel_stats.csv"))
        bj = os.path.join(src, f"bench_under_rocprof_{wl}.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(dst, f"{wl}_bench_under_rocprof.json"))
        # the same run with BHRT_SHADOW_OVERLAP=0 (mesh workloads): every kernel alone on the GPU
        for f in glob.glob(os.path.join(src, f"trace_alone_{wl}", "**", f"{wl}_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"{wl}_kernel_stats_alone.csv"))
        bj = os.path.join(src, f"bench_under_rocprof_alone_{wl}.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(dst, f"{wl}_bench_under_rocprof_alone.json"))
        # last frame timeline from the kernel trace
        for f in glob.glob(os.path.join(src, f"trace_{wl}", "**", f"{wl}_kernel_trace.csv"), recursive=True):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
            # a frame ends with its last k_resolve: the last frame = everything after the previous frame's last k_resolve
            res = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
            ends = [i for k, i in enumerate(res) if k + 1 == len(res) or res[k + 1] != i + 1]  # last k_resolve of each pass/frame
            if len(ends) >= 2:
                lo = ends[-2] + 1
                while lo < len(rows) and "k_resolve" in rows[lo]["Kernel_Name"]:
                    lo += 1
                t0 = int(rows[lo]["Start_Timestamp"])
                with open(os.path.join(dst, f"{wl}_last_frame_timeline.csv"), "w") as o:
                    o.write("kernel,start_us,duration_us,grid_size,workgroup_size,vgpr,lds_bytes\n")
                    for r in rows[lo:ends[-1] + 1]:
                        o.write(f"{short(r['Kernel_Name'])},{(int(r['Start_Timestamp']) - t0) / 1e3:.1f},"
                                f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f},{r.get('Grid_Size_X', r.get('Grid_Size'))},"
                                f"{r.get('Workgroup_Size_X', r.get('Workgroup_Size'))},{r.get('VGPR_Count', '')},{r.get('LDS_Block_Size', '')}\n")
        fetch, nf = counters(os.path.join(src, f"pmc_fetch_{wl}"))
        write, _ = counters(os.path.join(src, f"pmc_write_{wl}"))
        sq, _ = counters(os.path.join(src, f"pmc_sq_{wl}"))
        if not fetch or not write:
            continue
        cal_f = fetch["k_math_eval"]["FETCH_SIZE"] * 1024.0
        cal_w = write["k_math_eval"]["WRITE_SIZE"] * 1024.0
        f_read = CAL_BYTES_READ / cal_f
        f_write = CAL_BYTES_WRITTEN / cal_w
        out = {"_calibration": {"kernel": "k_math_eval (dword per lane, 2 x 512 MiB read, 512 MiB written)",
                                "FETCH_SIZE_bytes_reported": cal_f, "bytes_read": CAL_BYTES_READ, "read_factor": f_read,
                                "WRITE_SIZE_bytes_reported": cal_w, "bytes_written": CAL_BYTES_WRITTEN, "write_factor": f_write}}
        per_launch = {}
        for k in sorted(fetch):
            if k == "k_math_eval" or not k.startswith("k_"):
                continue
            n = max(1, nf.get(k, 1))
            rd = fetch[k]["FETCH_SIZE"] * 1024.0 * f_read
            wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 * f_write
            ent = {"launches": n, "read_bytes": rd, "written_bytes": wr, "bytes_per_launch": (rd + wr) / n}
            if k in sq:
                s = sq[k]
                ent.update({c: s[c] for c in s})
                if s.get("SQ_WAVES"):
                    ent["valu_insts_per_wave"] = s["SQ_INSTS_VALU"] / s["SQ_WAVES"]
                if s.get("SQ_THREAD_CYCLES_VALU") and s.get("SQ_INSTS_VALU"):
                    ent["valu_lanes_per_inst"] = s["SQ_THREAD_CYCLES_VALU"] / s["SQ_INSTS_VALU"]  # of 64: lanes enabled in an average VALU instruction
                if s.get("SQ_WAVE_CYCLES"):
                    # quad-cycle units (MI355X_MICROARCH.md): share of the waves' resident time in which a VALU instruction issues
                    ent["valu_active_share_of_wave_time"] = s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"]
            out[k] = ent
            per_launch[k + "_bytes_per_launch"] = (rd + wr) / n
        # the closest-hit trace of a wave step is k_trace_closest + (scenes with meshes) the parked-ray finish and its sort;
        # bench.py times them as one unit, so their traffic is summed per trace launch as well
        group = [k for k in out if k in ("k_trace_closest", "k_trace_mesh", "k_trace_mesh_stream", "k_park_count", "k_park_scatter", "k_scan_tiles", "k_scan_sums", "k_scan_add")]
        if "k_trace_closest" in out:
            n = out["k_trace_closest"]["launches"]
            per_launch["k_trace_closest_bytes_per_launch"] = sum(out[k]["read_bytes"] + out[k]["written_bytes"] for k in group) / n
        group = [k for k in out if k in ("k_trace_shadow", "k_trace_shadow_park", "k_shadow_mesh")]
        if group:
            n = max(out[k]["launches"] for k in group)
            per_launch["k_trace_shadow_bytes_per_launch"] = sum(out[k]["read_bytes"] + out[k]["written_bytes"] for k in group) / n
        # the caustic gather of one pass = cell sort + lane pass + one-wave-per-query pass (+ exact replay); bench.py times it as one unit
        group = [k for k in out if k.startswith("k_photon_gather") or k.startswith("k_gather_cell")]
        if group:
            n = max(1, out.get("k_photon_gather_fast", {}).get("launches", 1))
            per_launch["k_photon_gather_bytes_per_launch"] = sum(out[k]["read_bytes"] + out[k]["written_bytes"] for k in group) / n
        # VALU issue fraction of the workload's heavy kernels: wave-level VALU instructions of ONE frame (SQ_INSTS_VALU, PMC pass) x 4 cycles
        # (a wave64 instruction occupies its SIMD's 16 lanes for 4 cycles; measured 3.1-4.6 for this path's mix) over the
        # SIMD-cycles the kernel had: 1024 SIMDs x its time per frame (kernel-trace run of the same workload) x 2.4 GHz
        frames = None
        try:
            bj2 = json.load(open(os.path.join(dst, f"{wl}_bench_under_rocprof.json")))
            frames = bj2["steps"] + bj2["warmup"] + 1  # + the untimed first frame
            if wl == "c5":
                frames += 1  # + bench.py's untimed statistics frame (knob "gather_stats"): its lane pass is the kernel's second instantiation, same short name
        except Exception:
            pass
        ktime = collections.defaultdict(float)
        sp = os.path.join(dst, f"{wl}_kernel_stats_alone.csv")  # durations with the GPU to the kernel, like the counter passes
        if not os.path.exists(sp):
            sp = os.path.join(dst, f"{wl}_kernel_stats.csv")
        else:
            try:
                bj3 = json.load(open(os.path.join(dst, f"{wl}_bench_under_rocprof_alone.json")))
                frames = bj3["steps"] + bj3["warmup"] + 1
            except Exception:
                pass
        if os.path.exists(sp):
            for r in csv.DictReader(open(sp)):
                ktime[short(r["Name"])] += float(r["TotalDurationNs"]) * 1e-9
        for k in ("k_shade", "k_trace_mesh_stream", "k_trace_mesh", "k_shadow_mesh", "k_photon_gather_fast", "k_photon_gather_select"):
            e = out.get(k)
            if e and frames and ktime.get(k) and e.get("SQ_INSTS_VALU"):
                sec = ktime[k] / frames
                e["seconds_per_frame"] = sec
                e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * 4.0 / (1024 * sec * 2.4e9)
                name = {"k_trace_mesh_stream": "k_trace_closest", "k_trace_mesh": "k_trace_closest", "k_shadow_mesh": "k_trace_shadow", "k_photon_gather_fast": "k_photon_gather"}.get(k, k)
                per_launch.setdefault(name + "_valu_issue_frac", e["valu_issue_frac"])
                if e.get("valu_lanes_per_inst"):
                    per_launch.setdefault(name + "_valu_lanes_per_inst", e["valu_lanes_per_inst"])  # bench.py: roofline.valu_lane_frac = issue x lanes / 64
        json.dump(out, open(os.path.join(dst, f"{wl}_pmc_one_frame.json"), "w"), indent=1)
        traffic[wl] = per_launch
    traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py (one frame), "
                       "scaled by the factors of the in-pass calibration launch (profiles/<tag>/<wl>_pmc_one_frame.json), per launch")
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
