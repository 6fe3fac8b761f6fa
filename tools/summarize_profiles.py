"""Turns the rocprofv3 output of tools/collect_profiles.sh into the committed summaries (ROUND = r03 unless given):
  profiles/ROUND_<task>[_n<envs>]_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --task <task>` (hipGraph)
  profiles/ROUND_<task>[_n<envs>]_bench.json          the JSON line of that same run
  profiles/ROUND_pmc_traffic.json          HBM bytes per fw_step launch (2 x FETCH_SIZE + WRITE_SIZE; gfx950 tallies 128-B read
                                           requests at 64 B: MI355X_MICROARCH.md, HBM section; calibrated in round 1 with
                                           tools/calib_pmc.hip, profiles/r01_d_pmc_traffic.json)
  profiles/ROUND_valu_count.json           SQ counters per launch and the lane-instructions per env-step bench.py's
                                           roofline.valu uses (SQ_INSTS_VALU x 64 lanes / envs)
Keys of the two json files are `<task>` for the 4096-env runs and `<task>_n<envs>` for the others (e.g. the large-N roofline
evidence at 2^20 envs on the one-lane-per-env mapping).
usage: python tools/summarize_profiles.py gpurun_out/r03_prof [r03]"""
import csv, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", f"{rnd}_prof")
prof = os.path.join(ROOT, "profiles")
WORDS = {"waypoints": 94, "waypoints_wind": 94, "objlock": 203, "combined": 193}
traffic, valu = {}, {}
for p in (os.path.join(prof, f"{rnd}_pmc_traffic.json"), os.path.join(prof, f"{rnd}_valu_count.json")):
    if os.path.exists(p):                      # keep the keys of earlier collections of the same round
        with open(p) as f:
            old = json.load(f)
        (traffic if "pmc" in p else valu).update(old.get("tasks", old) if "pmc" in p else old)
for key in sorted(os.listdir(src)):
    d = os.path.join(src, key)
    if not os.path.isdir(d):
        continue
    m = re.match(r"(.+?)(?:_n(\d+))?$", key)
    task, envs = m.group(1), int(m.group(2) or 4096)
    st = os.path.join(d, "stats", "p_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(prof, f"{rnd}_{key}_kernel_stats.csv"))
    bj = os.path.join(src, f"{key}.bench.json")
    if os.path.exists(bj):
        with open(bj) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(prof, f"{rnd}_{key}_bench.json"), "w") as f:
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
        skip = {c: min(64, len(v) // 4) for c, v in cs.items()}                          # the warm-up launches
        mean = {c: sum(v[skip[c]:]) / max(len(v[skip[c]:]), 1) for c, v in cs.items()}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rd, wr = 2.0 * mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0      # counters are in KB
            alg = WORDS.get(task, 0) * 8 * envs
            traffic[key] = {"kernel": kern, "envs": envs, "launches": len(cs["FETCH_SIZE"]) - skip["FETCH_SIZE"], "FETCH_SIZE_KB": mean["FETCH_SIZE"],
                            "WRITE_SIZE_KB": mean["WRITE_SIZE"], "read": rd, "write": wr, "total": rd + wr,
                            "fetch_factor": 2.0, "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg if alg else None}
        if "SQ_INSTS_VALU" in mean:
            valu[key] = {"kernel": kern, "envs": envs, **{c: mean[c] for c in sorted(mean) if c.startswith("SQ_")},
                         "lane_instructions_per_env_step": mean["SQ_INSTS_VALU"] * 64.0 / envs,
                         "valu_instructions_per_wave": mean["SQ_INSTS_VALU"] / max(mean.get("SQ_WAVES", 1.0), 1.0),
                         "note": "SQ_INSTS_VALU counts wave-level VALU instructions of every wave of the launch (step waves and worker "
                                 "waves; all VALU classes: fp64, integer, conversions, DPP moves); x 64 lanes / envs"}
with open(os.path.join(prof, f"{rnd}_pmc_traffic.json"), "w") as f:
    json.dump({"what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
                       "`python3 bench.py --task T --steps 256 --warmup 64 --repeats 1 --no-cpu-baseline --no-graph [--envs-per-gpu N]`, fp64; bytes per "
                       "fw_step launch = 2 x FETCH_SIZE + WRITE_SIZE (worker half of the grid included)", "tasks": traffic}, f, indent=1)
with open(os.path.join(prof, f"{rnd}_valu_count.json"), "w") as f:
    json.dump(valu, f, indent=1)
print(json.dumps({"traffic": {k: round(v["total"]) for k, v in traffic.items()},
                  "valu_lane_instr_per_env_step": {k: round(v["lane_instructions_per_env_step"]) for k, v in valu.items()}}, indent=1))
