"""Turns the rocprofv3 output of tools/collect_profiles.sh into the committed summaries:
  profiles/r02_<task>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --task <task>` (hipGraph, 2000 steps)
  profiles/r02_pmc_traffic.json          HBM bytes per fw_step launch (2 x FETCH_SIZE + WRITE_SIZE; gfx950 tallies 128-B read
                                         requests at 64 B: MI355X_MICROARCH.md, HBM section; calibrated in round 1 with
                                         tools/calib_pmc.hip, profiles/r01_d_pmc_traffic.json)
  profiles/r02_valu_count.json           SQ counters per launch and the lane-instructions per env-step bench.py's
                                         roofline.valu uses (SQ_INSTS_VALU x 64 lanes / envs)
usage: python tools/summarize_profiles.py gpurun_out/r02_prof"""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_prof")
prof = os.path.join(ROOT, "profiles")
N_ENVS = 4096
traffic, valu = {}, {}
for task in sorted(os.listdir(src)):
    d = os.path.join(src, task)
    if not os.path.isdir(d):
        continue
    st = os.path.join(d, "stats", "p_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(prof, f"r02_{task}_kernel_stats.csv"))
    bj = os.path.join(src, f"{task}.bench.json")
    if os.path.exists(bj):
        with open(bj) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(prof, f"r02_{task}_bench.json"), "w") as f:
                f.write(lines[-1] + "\n")
    ctr = defaultdict(lambda: defaultdict(list))            # kernel -> counter -> values per dispatch
    for sub in ("fetch", "write", "sq"):
        p = os.path.join(d, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        with open(p) as f:
            for row in csv.DictReader(f):
                if "fw_step_kernel" in row["Kernel_Name"]:
                    ctr[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kern, cs in ctr.items():
        mean = {c: sum(v[64:]) / max(len(v[64:]), 1) for c, v in cs.items()}          # skip the warm-up launches
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rd, wr = 2.0 * mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0      # counters are in KB
            traffic[task] = {"kernel": kern, "launches": len(cs["FETCH_SIZE"]) - 64, "FETCH_SIZE_KB": mean["FETCH_SIZE"],
                             "WRITE_SIZE_KB": mean["WRITE_SIZE"], "read": rd, "write": wr, "total": rd + wr,
                             "fetch_factor": 2.0}
        if "SQ_INSTS_VALU" in mean:
            valu[task] = {"kernel": kern, "envs": N_ENVS, **{c: mean[c] for c in sorted(mean) if c.startswith("SQ_")},
                          "lane_instructions_per_env_step": mean["SQ_INSTS_VALU"] * 64.0 / N_ENVS,
                          "valu_instructions_per_wave": mean["SQ_INSTS_VALU"] / max(mean.get("SQ_WAVES", 1.0), 1.0),
                          "note": "SQ_INSTS_VALU counts wave-level VALU instructions of every wave of the launch (step waves and worker "
                                  "waves; all VALU classes: fp64, integer, conversions, DPP moves); x 64 lanes / 4096 envs"}
with open(os.path.join(prof, "r02_pmc_traffic.json"), "w") as f:
    json.dump({"what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
                       "`python3 bench.py --task T --steps 256 --warmup 64 --no-cpu-baseline --no-graph`, 4096 envs, fp64; bytes per "
                       "fw_step launch = 2 x FETCH_SIZE + WRITE_SIZE (worker half of the grid included)", "tasks": traffic}, f, indent=1)
with open(os.path.join(prof, "r02_valu_count.json"), "w") as f:
    json.dump(valu, f, indent=1)
print(json.dumps({"traffic": {k: round(v["total"]) for k, v in traffic.items()},
                  "valu_lane_instr_per_env_step": {k: round(v["lane_instructions_per_env_step"]) for k, v in valu.items()}}, indent=1))
