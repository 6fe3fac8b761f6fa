"""Do the figures README.md / DESIGN.md quote still follow from the committed profiles?

Round 4's documents drifted (a test count two rounds old, an update time from before its last optimisation, the rocprofv3 kernel
average quoted as the driver-timed headline).  Every figure below is (document, a regex with ONE group around the number as it is
written there, where the number comes from in profiles/rNN_*, relative tolerance); `python tools/check_docs.py` prints a table and
exits non-zero on a mismatch or on a sentence that can no longer be found, tests/test_docs.py runs it on the CPU.

Sources (ROUND = the newest rNN with a headline bench):
  rNN_headline_bench.json            `python bench.py` on one MI355X, as the driver runs it (not under a profiler)
  rNN_<task>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of bench.py --task <task>  (tools/collect_profiles.sh)
  rNN_rollout_bench.jsonl            tools/bench_rollout.py per task (tools/collect_rollout_benches.sh)
  rNN_render_<res>px_kernel_stats.csv rocprofv3 --kernel-trace --stats of tools/bench_render.py 4096 <res> render_only (tools/collect_render_pmc.sh)
  the test suite itself              `pytest --collect-only -m gpu` / `-m "not gpu"`
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_round():
    r = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_headline_bench.json")))
    if not r:
        raise SystemExit("no profiles/rNN_headline_bench.json")
    return os.path.basename(r[-1])[:3]


RND = latest_round()
P = lambda name: os.path.join(ROOT, "profiles", f"{RND}_{name}")


def headline():
    with open(P("headline_bench.json")) as f:
        return json.loads([l for l in f if l.startswith("{")][-1])


def kernel_avg_us(task):
    with open(P(f"{task}_kernel_stats.csv")) as f:
        rows = [r for r in csv.DictReader(f) if "fw_step" in r["Name"]]
    return float(rows[0]["AverageNs"]) / 1e3


def rollout(task, envs=4096):
    with open(P("rollout_bench.jsonl")) as f:
        for l in f:
            d = json.loads(l)
            if d["task"] == task and d["envs"] == envs:
                return d
    raise KeyError(task)


def render_avg_us(res):
    with open(P(f"render_{res}px_kernel_stats.csv")) as f:
        rows = [r for r in csv.DictReader(f) if "fw_render_kernel" in r["Name"]]
    return float(rows[0]["AverageNs"]) / 1e3


def collected(marker):
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", marker],
                         capture_output=True, text=True, cwd=ROOT).stdout
    m = re.search(r"(\d+)/\d+ tests collected|(\d+) tests collected", out)
    return int(m.group(1) or m.group(2))


FIGURES = [
    # README
    ("README.md", r"headline \(fp64[^|]*\| \*\*([\d.]+) M env-steps/s\*\*", lambda: headline()["value"] / 1e6, 0.01, "driver-form headline"),
    ("README.md", r"\(([\d.]+) µs per vec-step as `bench.py` times it", lambda: headline()["ms_per_step"] * 1e3, 0.01, "driver-form µs per vec-step"),
    ("README.md", r"the kernel alone ([\d.]+) µs by rocprofv3", lambda: kernel_avg_us("waypoints"), 0.01, "rocprofv3 kernel average"),
    ("README.md", r"([\d.]+) M \(the \d+ cores of the job's quota\)", lambda: headline()["cpu_baseline"]["value"] / 1e6, 0.05, "CPU restatement"),
    ("README.md", r"ObjLock ([\d.]+) µs, combined", lambda: kernel_avg_us("objlock"), 0.01, "ObjLock kernel"),
    ("README.md", r"combined \(20 obstacles[^)]*\) ([\d.]+) µs per vec-step", lambda: kernel_avg_us("combined"), 0.01, "combined kernel"),
    ("README.md", r"\*\*([\d.]+) k \(waypoints\) /", lambda: rollout("waypoints")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, waypoints"),
    ("README.md", r"\(waypoints\) / ([\d.]+) k \(ObjLock\)", lambda: rollout("objlock")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, ObjLock"),
    ("README.md", r"\(ObjLock\) / ([\d.]+) k \(combined\) env-steps/s end to end", lambda: rollout("combined")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, combined"),
    ("README.md", r"10 240-minibatch update in ([\d.]+) s", lambda: rollout("waypoints")["update_s"], 0.02, "update seconds"),
    # DESIGN
    ("DESIGN.md", r"through the C ABI, (\d+) tests\)", lambda: collected("gpu"), 0.0, "-m gpu tests collected"),
    ("DESIGN.md", r"Headline \(`python bench.py`[^*]*\*\*([\d.]+) M env-steps/s\*\*", lambda: headline()["value"] / 1e6, 0.01, "driver-form headline"),
    ("DESIGN.md", r"the kernel alone \*\*([\d.]+) µs\*\* per 4096-env step", lambda: kernel_avg_us("waypoints"), 0.01, "rocprofv3 kernel average"),
    ("DESIGN.md", r"CPU restatement:\s+([\d.]+) M env-steps/s", lambda: headline()["cpu_baseline"]["value"] / 1e6, 0.05, "CPU restatement"),
    ("DESIGN.md", r"update ([\d.]+)\s+s per 65 536 samples", lambda: rollout("waypoints")["update_s"], 0.02, "update seconds"),
    ("DESIGN.md", r"\| `fw_render_kernel` \| 33.5 MB written \| ([\d.]+) µs", lambda: render_avg_us(32), 0.01, "fw_render at 32 x 32"),
    ("DESIGN.md", r"\| `fw_render_kernel` \|[^|]*\|[^|]*64²: ([\d.]+) µs", lambda: render_avg_us(64), 0.01, "fw_render at 64 x 64"),
    ("DESIGN.md", r"\| `fw_render_kernel` \|[^|]*\|[^|]*128²: ([\d.]+) µs", lambda: render_avg_us(128), 0.01, "fw_render at 128 x 128"),
    ("README.md", r"`fw_render` ([\d.]+) µs per 4096 × 32² images", lambda: render_avg_us(32), 0.01, "fw_render at 32 x 32"),
    ("DESIGN.md", r"end to end ([\d.]+) k \(waypoints\)", lambda: rollout("waypoints")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, waypoints"),
    ("DESIGN.md", r"\(waypoints\) / ([\d.]+) k \(ObjLock\)", lambda: rollout("objlock")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, ObjLock"),
    ("DESIGN.md", r"\(ObjLock\) / ([\d.]+) k \(combined\)", lambda: rollout("combined")["end_to_end_env_steps_per_s_reference_hparams"] / 1e3, 0.02, "end to end, combined"),
    ("DESIGN.md", r"Collector ([\d.]+) µs per vec-step", lambda: rollout("waypoints")["rollout_us_per_vec_step"], 0.02, "collector µs per vec-step"),
]


def check(verbose=True):
    bad = []
    texts = {}
    for doc, rx, src, tol, what in FIGURES:
        if doc not in texts:
            with open(os.path.join(ROOT, doc)) as f:
                texts[doc] = re.sub(r"[ \t]*\n[ \t]*", " ", f.read())      # (a sentence may wrap)
        m = re.search(rx, texts[doc], re.S)
        if not m:
            bad.append(f"{doc}: the sentence carrying '{what}' was not found ({rx})")
            continue
        quoted, want = float(m.group(1)), float(src())
        ok = abs(quoted - want) <= tol * abs(want)          # (tolerance 0: exact -- counts)
        if verbose:
            print(f"{'ok ' if ok else 'BAD'} {doc:10s} {what:32s} quoted {quoted:10.3f}   profiles/{RND}: {want:10.3f}")
        if not ok:
            bad.append(f"{doc}: {what}: quoted {quoted}, profiles/{RND} say {want:.4g} (tolerance {tol:.0%})")
    return bad


if __name__ == "__main__":
    problems = check()
    for b in problems:
        print("MISMATCH:", b)
    sys.exit(1 if problems else 0)
