#!/usr/bin/env python3
"""Condense a gpurun_out/<run>/ directory of rocprofv3 CSVs into the small summaries committed under profiles/.
usage: tools/collect_profiles.py gpurun_out/r01b r01b"""
import csv
import glob
import json
import os
import shutil
import sys

base, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def stats(sub, name):
    f = sorted(glob.glob(os.path.join(base, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if not f:
        return
    rows = list(csv.DictReader(open(f[-1])))          # the newest run (gpurun merges results of earlier calls into the same directory)
    with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if float(r["Percentage"]) < 0.001 and "tsta" not in r["Name"]:
                continue
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(sub, counter, kernel):
    f = sorted(glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    res = []
    for r in csv.DictReader(open(f[-1])) if f else []:
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            res.append(dict(counter=counter, value_kb=float(r["Counter_Value"]), grid=int(r["Grid_Size"]),
                            wg=int(r["Workgroup_Size"]), vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                            dur_ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return res


stats("prof_stats", "bench")
stats("prof_cfg", "configs")
fe = pmc("prof_fetch", "FETCH_SIZE", "psa_score_packed")
wr = pmc("prof_write", "WRITE_SIZE", "psa_score_packed")
if fe and wr:
    fb, wb = fe[0]["value_kb"] * 1024, wr[0]["value_kb"] * 1024
    t = {"kernel": "psa_score_packed_kernel<32,2>", "pairs_per_gpu": 8192, "len": 50000,
         "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb,
         "fetch_correction": "x2: gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md, HBM); the expected "
                             "boundary-row reads (4096 couples x 24 band hand-offs x 50000 columns x 8 B = 39.3 GB) + sequence re-reads "
                             "(24 x 0.41 GB) agree with 2x raw",
         "hbm_bytes_per_launch": 2 * fb + wb, "pmc_rows": fe + wr,
         "what_it_is": "band hand-off rows (per column and couple one dword H' + one dword Ebar', both pairs' 16-bit halves side by side, "
                       "per 2048-row band; round 1 stored two int2 = 16 B), i.e. workspace traffic, <1% of HBM bandwidth; algorithmic "
                       "bytes are 0.82 GB per launch",
         "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                    "--no-cpu-baseline --no-generic (separate passes)"}
    t["_build"] = open(os.path.join(base, "build_info.txt")).read().split()[1] if os.path.exists(os.path.join(base, "build_info.txt")) else None
    json.dump(t, open(os.path.join(out, "traffic_psa_score.json"), "w"), indent=1)
# HBM traffic of the configs' dominant kernels (bench_configs.py 2 3 4 5t under separate FETCH_SIZE / WRITE_SIZE passes):
# profiles/traffic_configs.json, read by bench_configs.py for roofline.traffic
def pmc_by_kernel(sub, counter):
    f = sorted(glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f[-1])) if f else []:
        if r["Counter_Name"] == counter:
            acc.setdefault(r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").replace("tsta::", ""), []).append(float(r["Counter_Value"]) * 1024)
    return acc


def build_of_run():
    """16 hex digits of the library the profiled run used: <run>/build_info.txt (written on the GPU box by the profile script) or,
    failing that, the tree's tsta_amd/lib/BUILD_INFO (valid only while the sources have not changed since the run)."""
    for f in (os.path.join(base, "build_info.txt"), os.path.join(os.path.dirname(out), "tsta_amd", "lib", "BUILD_INFO")):
        if os.path.exists(f):
            w = open(f).read().split()
            if len(w) >= 2:
                return w[1]
    return None


cf, cw = pmc_by_kernel("prof_cfg_fetch", "FETCH_SIZE"), pmc_by_kernel("prof_cfg_write", "WRITE_SIZE")
if cf and cw:
    tc = {"_build": build_of_run(), "_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench_configs.py 2 3 4 5t; "
                  "hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, "
                  "median over the launches of that kernel in the run; poa_dataflow_kernel: mean x 4 = one set of the README example's four "
                  "fills (the unit configs['4'].roofline.achieved is quoted on)"}
    for k in sorted(set(cf) & set(cw)):
        if not any(x in k for x in ("psa_", "poa_")):
            continue
        mult = 4 if k == "poa_dataflow_kernel" else 1
        if mult == 1:                                             # one shape per kernel: the median rejects the small warm-up launch
            fb, wb = sorted(cf[k])[len(cf[k]) // 2], sorted(cw[k])[len(cw[k]) // 2]
        else:                                                     # the four fills of the example, equally often: mean x 4
            fb, wb = sum(cf[k]) / len(cf[k]), sum(cw[k]) / len(cw[k])
        tc[k] = {"launches_seen": len(cf[k]), "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb,
                 "hbm_bytes_per_launch": mult * (2 * fb + wb)}
    json.dump(tc, open(os.path.join(out, "traffic_configs.json"), "w"), indent=1)
# SQ counters of the headline kernel (VALU issue evidence): profiles/valu_psa_score.json, read by bench.py
def sq(sub, kernel):
    f = sorted(glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc, dur = {}, 0
    for r in csv.DictReader(open(f[-1])) if f else []:
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc, dur


c1, dur1 = sq("prof_sq", "psa_score_packed")
c2, _ = sq("prof_sq2", "psa_score_packed")
if c1.get("SQ_INSTS_VALU"):
    P, L = 8192, 50000
    cells = float(P) * L * L
    insts = c1["SQ_INSTS_VALU"]
    n_simd = 256 * 4
    clk = (c2.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) / (dur1 * 1e-9) if c2.get("GRBM_GUI_ACTIVE") else None     # GRBM sums over the 8 XCDs
    v = {"kernel": "psa_score_packed_kernel<32,2>", "pairs_per_gpu": P, "len": L, "duration_ns": dur1, "counters": {**c2, **c1},
         "valu_wave_instructions": insts, "ops_per_cell": insts * 64.0 / cells,
         "valu_instr_per_simd": insts / n_simd,
         "shader_clock_GHz_from_GRBM_GUI_ACTIVE": clk / 1e9 if clk else None,
         "cycles_per_valu_instruction_per_simd": (clk * dur1 * 1e-9) / (insts / n_simd) if clk else None,
         "SQ_ACTIVE_INST_VALU_over_SQ_INSTS_VALU_quadcycles": c1.get("SQ_ACTIVE_INST_VALU", 0.0) / insts,
         "reading": "SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md, cycle constants): a ratio of 1.0 means every VALU "
                    "instruction of this kernel holds its SIMD's vector pipe for 4 cycles; kernel cycles / VALU instructions per SIMD "
                    "= cycles_per_valu_instruction_per_simd: the two waves of a SIMD together issue one VALU instruction every that "
                    "many cycles, i.e. the SIMD's VALU issue is saturated when it is ~3.9-4.0 (tools/cell_microbench: the same mix "
                    "alone sustains 3.88-4.0)",
         "command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU "
                    "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic "
                    "--no-related --no-verify   (second pass: GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR "
                    "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS)"}
    v["_build"] = open(os.path.join(base, "build_info.txt")).read().split()[1] if os.path.exists(os.path.join(base, "build_info.txt")) else None
    json.dump(v, open(os.path.join(out, "valu_psa_score.json"), "w"), indent=1)
for name in ("cell_microbench", "valu_microbench", "step_microbench"):
    if os.path.exists(os.path.join(base, name + ".txt")):
        shutil.copy(os.path.join(base, name + ".txt"), os.path.join(out, "%s_%s.txt" % (tag, name)))
for f, name in (("bench_default.json", "bench_default"), ("bench_prof.json", "bench_under_rocprof"), ("configs.jsonl", "configs"),
                ("side_timings.txt", "side_timings")):
    if os.path.exists(os.path.join(base, f)):
        shutil.copy(os.path.join(base, f), os.path.join(out, "%s_%s.%s" % (tag, name, f.split(".")[-1])))
print(sorted(os.listdir(out)))
