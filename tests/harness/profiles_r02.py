"""Turns the rocprofv3 outputs of tests/harness/make_profiles.sh (gpurun_out/<tag>/) into the
summaries kept under profiles/: kernel stats of the bench command, per-kernel counter totals with
derived figures for the headline configuration and for the large scenes, the counter calibration on
the FMA microbenchmark, and the HBM traffic per frame that bench.py reports.

usage: python tests/harness/profiles_r02.py gpurun_out/profiles_r02 profiles r02
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
N_SIMD, N_CU, N_XCD = 1024, 256, 8
PEAK_HBM_GBS = 8000.0
F_BOX, F_TRI, F_SPH, F_SHADE = 24, 52, 31, 160


def counters_of(root, kernel_substr, which="sum"):
    """Counter totals over the dispatches of the kernels whose name contains kernel_substr."""
    agg = collections.defaultdict(float)
    meta = {}
    for f in glob.glob(root + "/*/*/*_counter_collection.csv") + glob.glob(root + "/*/*_counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if kernel_substr in r["Kernel_Name"]]
        if not rows:
            continue
        if which == "last":
            last = max(int(r["Dispatch_Id"]) for r in rows)
            rows = [r for r in rows if int(r["Dispatch_Id"]) == last]
        for r in rows:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            meta = {x: r[x] for x in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                      "Workgroup_Size", "Grid_Size")}
    return dict(agg), meta


def kernel_ms(root, kernel_substr):
    for f in glob.glob(root + "/kt/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Name"]:
                return float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    return None, 0


def workload_line(root):
    for line in open(root + "/kt.log"):
        m = re.match(r"ms ([\d.]+) ray_all calls (\d+) rays traversed (\d+) frames (\d+) scene (\S+) (\d+) (\d+)", line)
        if m:
            return dict(ms_by_hip_events=float(m.group(1)), ray_all_calls=int(m.group(2)), rays_traversed=int(m.group(3)),
                        frames=int(m.group(4)), scene=m.group(5), width=int(m.group(6)), height=int(m.group(7)))
    return {}


def derived(c, ms, work):
    """Figures derived from the raw counters of one launch of the persistent kernel."""
    cycles = c["GRBM_GUI_ACTIVE"] / max(1, round(c["GRBM_GUI_ACTIVE"] / (ms * 1e-3 * 2.4e9 * N_XCD))) / N_XCD if "GRBM_GUI_ACTIVE" in c else None
    # GRBM_GUI_ACTIVE is summed over the XCDs and over the passes that collected it
    passes = max(1, round(c["GRBM_GUI_ACTIVE"] / (ms * 1e-3 * 2.4e9 * N_XCD)))
    cycles = c["GRBM_GUI_ACTIVE"] / passes / N_XCD
    d = {"kernel_ms": ms, "gpu_cycles": cycles, "clock_GHz": cycles / (ms * 1e-3) / 1e9}
    d["lanes_active_per_valu_inst"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
    d["valu_insts_per_simd_cycle"] = c["SQ_INSTS_VALU"] / (cycles * N_SIMD)
    d["salu_insts_per_simd_cycle"] = c["SQ_INSTS_SALU"] / (cycles * N_SIMD)
    d["waves_resident_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (cycles * N_SIMD)
    d["wave_cycles_waiting_for_an_instruction"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    d["l1_requests_reaching_l2"] = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    # the L1 (TCP) takes one access (a 64-byte piece of one line) per clock: its rate against the ceiling that
    # tests/harness/valu_calib.hip reaches with loads whose lanes all touch different lines
    d["l1_accesses_per_cu_cycle"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cycles * N_CU)
    d["l1_accesses_per_load_instruction"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["TA_TOTAL_WAVEFRONTS_sum"]
    d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (cycles * N_CU)
    d["td_busy_frac"] = c["TD_TD_BUSY_sum"] / (cycles * N_CU)
    d["tcp_latency_cycles_per_wave_instruction"] = c["TCP_TCP_LATENCY_sum"] / c["TA_TOTAL_WAVEFRONTS_sum"]
    d["lds_bank_conflict_per_lds_inst"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]
    # HBM: FETCH_SIZE counts 64-byte units as if they were 32 (gfx950 requests are 128 B tallied at 64): x2; KiB units
    rd = c["FETCH_SIZE"] * 1024 * 2
    wr = c["WRITE_SIZE"] * 1024
    d["hbm_read_bytes"] = rd
    d["hbm_write_bytes"] = wr
    d["hbm_GBps"] = (rd + wr) / (ms * 1e-3) / 1e9
    d["hbm_frac_of_peak"] = d["hbm_GBps"] / PEAK_HBM_GBS
    if work:
        d.update(work)
        d["Grays_per_s_traversed"] = work["rays_traversed"] / (ms * 1e-3) / 1e9
        d["valu_lane_insts_per_ray"] = c["SQ_THREAD_CYCLES_VALU"] / work["rays_traversed"]  # the counter adds the active lanes of every VALU instruction
        d["hbm_bytes_per_ray"] = (rd + wr) / work["rays_traversed"]
        d["l2_read_requests_per_ray"] = c["TCP_TCC_READ_REQ_sum"] / work["rays_traversed"]
        d["l1_accesses_per_ray"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / work["rays_traversed"]
    return d


os.makedirs(dst, exist_ok=True)
# 1. bench under rocprof
shutil.copy(src + "/bench_kernel_stats.csv", "%s/%s_bench_kernel_stats.csv" % (dst, tag))
shutil.copy(src + "/bench_under_rocprof.json", "%s/%s_bench_under_rocprof.json" % (dst, tag))

# 2. calibration
calib = {"source": "tests/harness/valu_calib.hip under rocprofv3 --pmc (tests/harness/make_profiles.sh); 20000 x 32 FMAs per wave",
         "runs": {}}
names = {"fma_0_8": "v_fma_f32, all 64 lanes, 8 waves/SIMD", "fma_1_8": "v_fma_f32, lanes 0-31, 8 waves/SIMD",
         "fma_3_8": "v_fma_f32, even lanes, 8 waves/SIMD", "fma_2_8": "v_fma_f32, lanes 0-15, 8 waves/SIMD",
         "pk_0_8": "v_pk_fma_f32, all 64 lanes, 8 waves/SIMD", "fma_0_2": "v_fma_f32, all 64 lanes, 2 waves/SIMD"}
for key, label in names.items():
    root = "%s/calib_%s" % (src, key)
    c, meta = counters_of(root, "fma_kernel", which="last")
    if not c:
        continue
    line = [l for l in open(root + ".log") if l.startswith(("fma ", "pk_fma "))]
    ms = float(line[-1].split()[-4]) if line else None
    wps = int(key.split("_")[-1])
    n_fma = 20000 * 32 * wps * N_SIMD  # wave-instructions of the measured kind in the dispatch
    cycles = c["GRBM_GUI_ACTIVE"] / N_XCD
    calib["runs"][key] = {
        "what": label, "ms": ms, "fma_wave_instructions": n_fma, "gpu_cycles": cycles,
        "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"],
        "SQ_ACTIVE_INST_VALU2": c.get("SQ_ACTIVE_INST_VALU2"), "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"],
        "SQ_BUSY_CYCLES": c["SQ_BUSY_CYCLES"], "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"],
        "valu_insts_per_simd_cycle": c["SQ_INSTS_VALU"] / (cycles * N_SIMD),
        "fma_per_simd_cycle_from_known_count": n_fma / (cycles * N_SIMD),
        "active_inst_valu_per_inst": c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
        "lanes_active_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
    }
if "fma_0_8" in calib["runs"]:
    calib["ceiling_plain_fma"] = calib["runs"]["fma_0_8"]["valu_insts_per_simd_cycle"]
if "pk_0_8" in calib["runs"]:
    calib["ceiling_packed_fma"] = calib["runs"]["pk_0_8"]["valu_insts_per_simd_cycle"]

# 2b. vector-memory calibration (valu_calib l): clocks per 16-byte load instruction under masks, and the
# L1 counters of the same runs
if os.path.exists(src + "/vmem_calibration.txt"):
    shutil.copy(src + "/vmem_calibration.txt", "%s/%s_vmem_calibration.txt" % (dst, tag))
if os.path.exists(src + "/fma_mix_rate.txt"):
    shutil.copy(src + "/fma_mix_rate.txt", "%s/%s_fma_mix_rate.txt" % (dst, tag))
vmc = glob.glob(src + "/vmem_pmc/*/*_counter_collection.csv")
if vmc:
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(vmc[0])):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["grid"] = int(r["Grid_Size"])
    ds = [d for d in sorted(per) if per[d]["grid"] > 4096]
    mask_names = ["all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15", "every 4th lane", "one lane"]
    runs, i = [], 0
    for spread in ("contiguous", "divergent"):
        for mname in mask_names:
            for wps in (2, 6):
                c = per[ds[i + 1]]  # the second repetition of each shape
                i += 2
                cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
                runs.append({"addresses": spread, "mask": mname, "waves_per_simd": wps,
                             "l1_accesses_per_load_instruction": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["TA_TOTAL_WAVEFRONTS_sum"],
                             "l1_accesses_per_cu_cycle": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cyc * N_CU),
                             "ta_busy": c["TA_TA_BUSY_sum"] / (cyc * N_CU), "td_busy": c["TD_TD_BUSY_sum"] / (cyc * N_CU)})
    json.dump({"source": "tests/_build/valu_calib l 4000 under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TOTAL_WAVEFRONTS_sum "
                         "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE: 8 independent 16-byte loads per lane and iteration from a 64 KB table",
               "runs": runs}, open("%s/%s_vmem_counters.json" % (dst, tag), "w"), indent=1)
vm = dst + "/%s_vmem_calibration.txt" % tag
if os.path.exists(vm):
    rows = {}
    for line in open(vm):
        m = re.match(r"(contiguous|divergent)\s+(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[(m.group(1), m.group(2), int(m.group(3)))] = float(m.group(5))
    if ("divergent", "all 64 lanes", 6) in rows:
        calib["l1_access_ceiling_per_cu_cycle"] = 64.0 / rows[("divergent", "all 64 lanes", 6)]
        calib["l1_access_ceiling_source"] = ("profiles/%s_vmem_calibration.txt: 64-lane 16-byte loads whose lanes touch 64 different lines, "
                                             "six waves per SIMD, %.1f CU clocks per instruction" % (tag, rows[("divergent", "all 64 lanes", 6)]))
        calib["clocks_per_contiguous_load_instruction"] = rows.get(("contiguous", "all 64 lanes", 6))

# 3. PMC summaries
traffic = {}
calib["kernel"] = {}
for key, cfg, out_name in (("pmc_c", "c", "%s_pmc_summary.json" % tag), ("pmc_hf256", None, "%s_large_hf256.json" % tag),
                           ("pmc_x5", "e", "%s_large_x5.json" % tag), ("pmc_x5s", "e_visible", "%s_large_x5s.json" % tag)):
    root = "%s/%s" % (src, key)
    if not os.path.isdir(root):
        continue
    c, meta = counters_of(root, "render_persistent_kernel")
    ms, calls = kernel_ms(root, "render_persistent_kernel")
    work = workload_line(root)
    out = {"source": "rocprofv3 --kernel-trace --stats and separate --pmc passes of tests/harness/prof_run.py %s %s "
                     "(tests/harness/pmc.sh); one launch of the persistent kernel" % (work.get("frames"), work.get("scene")),
           "kernel": meta, "counters": c, "derived": derived(c, ms, work)}
    json.dump(out, open("%s/%s" % (dst, out_name), "w"), indent=1)
    d = out["derived"]
    if cfg:
        traffic[cfg] = {"hbm_bytes_per_frame": (d["hbm_read_bytes"] + d["hbm_write_bytes"]) / work["frames"],
                        "hbm_read_bytes_per_frame": d["hbm_read_bytes"] / work["frames"],
                        "hbm_write_bytes_per_frame": d["hbm_write_bytes"] / work["frames"],
                        "source": "profiles/%s: FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte requests at 64) + WRITE_SIZE x 1024, "
                                  "one %d-frame launch, separate --pmc passes" % (out_name, work["frames"])}
        calib["kernel"][cfg] = {"valu_insts_per_simd_cycle": d["valu_insts_per_simd_cycle"],
                                "lanes_active_per_valu_inst": d["lanes_active_per_valu_inst"],
                                "l1_accesses_per_cu_cycle": d["l1_accesses_per_cu_cycle"],
                                "l1_accesses_per_ray": d.get("l1_accesses_per_ray")}
    print(key, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}))
json.dump(traffic, open(dst + "/pmc_traffic.json", "w"), indent=1)
json.dump(calib, open(dst + "/valu_calibration.json", "w"), indent=1)
print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in calib["runs"].items()}, indent=1))
