"""Turns the rocprofv3 output of tools/collect_learner_pmc.sh into profiles/ROUND_learner_pmc.json (+ the kernel statistics of the
graph-replayed bench run as profiles/ROUND_rollout_<key>_kernel_stats.csv and its JSON line).

Per learner kernel and (task, envs) key: launches and average duration from the `--kernel-trace --stats` pass of the DEFAULT
(hipGraph) run; per-launch means of every counter from the eager counter passes; HBM-side traffic = 2 x FETCH_SIZE + WRITE_SIZE
(KB -> bytes; the gfx950 read correction of MI355X_MICROARCH.md, calibrated with tools/calib_pmc.hip); the algorithmic bytes /
MFMA flops of tools/learner_accounting.py and the roofline fractions that follow.  Every `frac` can be recomputed from the
numbers in the file.
usage: python tools/summarize_learner_pmc.py gpurun_out/r04_learner [r04]"""
import csv, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import learner_accounting as A

rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", f"{rnd}_learner")
prof = os.path.join(ROOT, "profiles")
KERNELS = ("fw_collect_kernel", "fw_collect_close_kernel", "fw_ppo_update_kernel", "fw_ppo_pack_kernel", "fw_render_kernel")
STEP_WORDS = {"waypoints": 94, "waypoints_wind": 94, "objlock": 203, "combined": 193}
OBS_DIM = {"waypoints": 28, "waypoints_wind": 28, "objlock": 56, "combined": 28}
HP = {"waypoints": (16, 128, 20), "objlock": (8, 64, 10), "combined": (8, 128, 20)}      # n_steps, batch, epochs of tools/bench_rollout.py

out_path = os.path.join(prof, f"{rnd}_learner_pmc.json")
result = {}
if os.path.exists(out_path):
    with open(out_path) as f:
        result = json.load(f).get("runs", {})


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"(?:fwsim::)?(\w+)", name)
    return m.group(1) if m else name


for key in sorted(os.listdir(src)):
    d = os.path.join(src, key)
    if not os.path.isdir(d):
        continue
    m = re.match(r"(.+?)(?:_n(\d+))?$", key)
    task, envs = m.group(1), int(m.group(2) or 4096)
    run = {"task": task, "envs": envs, "kernels": {}}
    bj = os.path.join(src, f"{key}.bench.json")
    if os.path.exists(bj):
        with open(bj) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        if lines:
            run["bench_line"] = json.loads(lines[-1])
            with open(os.path.join(prof, f"{rnd}_rollout_{key}_bench_under_profiler.json"), "w") as f:
                f.write(lines[-1] + "\n")
    st = os.path.join(d, "stats", "p_kernel_stats.csv")
    dur = {}
    if os.path.exists(st):
        shutil.copy(st, os.path.join(prof, f"{rnd}_rollout_{key}_kernel_stats.csv"))
        with open(st) as f:
            for row in csv.DictReader(f):
                k = short(row["Name"])
                if k.startswith(KERNELS):
                    e = {"full_name": row["Name"][:120], "launches": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                         "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3}
                    # fw_ppo_update_kernel<32> / <64>: the bench run's second configuration (batch 4096) runs the other instantiation;
                    # the reference-hyper-parameter launches are the long ones
                    if k not in dur or e["max_us"] > dur[k]["max_us"]:
                        dur[k] = e
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in ("fetch", "write", "sq", "mfma", "lds"):
        p = os.path.join(d, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k.startswith(KERNELS):
                    ctr[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(set(dur) | set(ctr)):
        e = dict(dur.get(k, {}))
        cs = ctr.get(k, {})
        # the first launch of a kernel pays code upload / cold caches: leave it out where there are several
        mean = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()}
        e["counters_per_launch"] = {c: mean[c] for c in sorted(mean)}
        e["counter_launches"] = {c: len(v) for c, v in cs.items()}
        traffic = None
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            traffic = 2.0 * mean["FETCH_SIZE"] * 1024.0 + mean["WRITE_SIZE"] * 1024.0
            e["traffic_bytes_per_launch"] = {"read": 2.0 * mean["FETCH_SIZE"] * 1024.0, "write": mean["WRITE_SIZE"] * 1024.0, "total": traffic, "fetch_factor": 2.0}
        D = OBS_DIM.get(task, 28)
        acct = None
        if k.startswith("fw_collect_kernel"):
            acct = A.collect_step(envs, D, STEP_WORDS.get(task, 94))
        elif k.startswith("fw_collect_close"):
            acct = A.collect_close(envs, D, HP.get(task, (16, 128, 20))[0])
        elif k.startswith("fw_ppo_update_kernel"):
            T, B, ep = HP.get(task, (16, 128, 20))
            acct = A.ppo_update(ep * (T * envs // B), B, D)
        elif k.startswith("fw_ppo_pack"):
            T, B, ep = HP.get(task, (16, 128, 20))
            acct = A.ppo_pack(ep * (T * envs // B), B, D)
        elif k.startswith("fw_render"):
            acct = A.render(envs, 32)
        if k.startswith(("fw_ppo_update_kernel", "fw_ppo_pack_kernel")) and "max_us" in e:
            # the bench run launches this kernel for two configurations (the reference's minibatch size, then batch 4096): the
            # reference-hyper-parameter launches are the long ones
            e["avg_us_all_launches"] = e["avg_us"]; e["avg_us"] = e["max_us"]
            e["avg_us_note"] = "longest launch = an update with the reference's hyper-parameters (the run also holds batch-4096 updates)"
        if acct is not None:
            e["algorithmic"] = acct
            if "avg_us" in e:
                if k.startswith("fw_ppo_update_kernel"):
                    e["roofline"] = A.roofline_mfma(acct["mfma_flops"], e["avg_us"], acct["mfma_peak_tflops"], traffic)
                    e["roofline"]["note"] = "fp32 MFMA peak of the workgroups the sequential dependency lets the update use (one CU each)"
                else:
                    e["roofline"] = A.roofline_hbm(acct["bytes"], e["avg_us"], traffic, acct.get("l2_served_bytes", 0.0))
            if traffic is not None and acct["bytes"]:
                e["traffic_over_algorithmic"] = traffic / acct["bytes"]      # (>= 1 by construction: `bytes` counts every buffer once)
        if "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"] > 0:
            wc = mean["SQ_WAVE_CYCLES"]
            e["fractions_of_wave_cycles"] = {c: mean[c] / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                                       "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if c in mean}
        if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            e["lds_bank_conflict_rate"] = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"]
        run["kernels"][k] = e
    result[key] = run

with open(out_path, "w") as f:
    json.dump({"what": "tools/collect_learner_pmc.sh: rocprofv3 --kernel-trace --stats of `python3 tools/bench_rollout.py <task> <envs>` (hipGraph replays: durations) "
                       "and --pmc passes of `... pmc` (eager launches; FETCH_SIZE / WRITE_SIZE / SQ / MFMA / LDS counters in separate runs), fp64 envs. "
                       "traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024); algorithmic bytes / flops: tools/learner_accounting.py; "
                       "peaks: HBM 8000 GB/s, fp32 MFMA 157.3 TFLOP/s per chip / 256 CUs per workgroup used", "runs": result}, f, indent=1)
brief = {}
for key, run in result.items():
    for k, e in run["kernels"].items():
        r = e.get("roofline")
        brief[f"{key}:{k}"] = {"avg_us": e.get("avg_us"), "frac": r and round(r["frac"], 4), "bound": r and r["bound"],
                               "traffic_over_alg": e.get("traffic_over_algorithmic") and round(e["traffic_over_algorithmic"], 2)}
print(json.dumps(brief, indent=1))
