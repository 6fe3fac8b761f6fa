#!/usr/bin/env python
"""bench.py -- env-steps/sec of the fused fixed-wing step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): env-steps/sec at N parallel envs, FixedwingWaypoints.
Workload (configs[1]): the headline TRAIN_CONFIG of
train/train_Fixedwing_Waypoints_v3.py:27-55 (8 targets, reach 4 m, sparse reward,
euler, dome 100 m, 120 s, ctx 2, 30 Hz agent => 8 physics ticks + 4 reward/obs
evaluations per env-step, motor noise on, auto-reset on), 4096 envs per GPU,
physics-only step(): actions are U(-1,1) device tensors cycled from a pool of 64
(torch.Generator seed 0); scenarios from Philox(seed 42).  fp64 arithmetic (the
reference's).  A "step" is one fw_step launch over the rank's 4096 envs; envs
shard across ranks with NO data-path collective (weak scaling, each rank keys
its RNG on the global env id).

`--gpus N` is honoured whichever way the script is started: under torch.distributed.run (WORLD_SIZE set) it must equal the
world size (anything else exits non-zero -- a `--gpus 8` request never prints `"n_gpus": 1`); started bare with N > 1 it
launches its own N rank processes BEFORE this process makes any GPU call and forwards rank 0's line.

Timing: W warm-up steps, then `--repeats` (default 30) timed regions of EXACTLY K steps each, every region bracketed by a
barrier + torch.cuda.synchronize() on both sides and max-reduced over ranks; `ms_per_step` / `value` are the MEDIAN region
(a single sub-millisecond sample is fragile), `steps` stays K, `repeats` and the spread are reported beside it.

One JSON line on rank 0.  `roofline` prices the step kernel against HBM:
algorithmic bytes per env-step = 94 words (40 read + 54
written) x 8 B = 752 B (376 B in fp32); `achieved` = bytes per launch / mean launch
time measured with HIP events on the launch stream over the timed region.
`cpu_baseline` = the plain-C oracle ("port": the PyBullet reference cannot run,
its deps are absent) on the host cores, same workload, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import pyflyt_drone_amd as P  # noqa: E402
from pyflyt_drone_amd import config as K  # noqa: E402

ENVS_PER_GPU = 4096
WORDS_PER_ENV_STEP = 94
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
POOL = 64
# Vector-issue ceiling of the chip for 64-bit lane operations: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-instr/s
# (one wave64 fp64 VALU instruction occupies its SIMD for 4 cycles).  DESIGN.md section 6 gives the instruction counts.
VALU_PEAK_TLANE = 256 * 4 * 16 * 2.4e9 / 1e12

# --task: the four step kernels, their config (reference TRAIN_CONFIGs) and algorithmic words per env-step.
#   waypoints       94 = 40 read + 54 written
#   waypoints_wind  same state traffic as waypoints (the 7 wind words are already among the 40 read)
#   objlock         rigid 13 + actuators 6 + action 4 + wind 7 + counters 2 + duck 3 + vision scalars 8 + frame 8 + history 27
#                   = 78 read; state 23 + vision scalars 8 + frame 8 + history 27 + obs 56 + reward / flags / info 3 = 125
#                   written => 203 words ("about 190")
#   combined        waypoints' 94 + duck 3 + vision scalars 8 + frame 8 + phase 2 read (21) + vision scalars 8 + frame 8 +
#                   phase 2 written (18) + 3 x 20 obstacle words read = 193 words
TASKS = {
    "waypoints": ("FixedwingWaypoints-v3 TRAIN_CONFIG (8 targets, sparse, euler, 30 Hz)", lambda K, dt: K.train_waypoints_v3_config(dtype=dt), 94),
    "waypoints_wind": ("FixedwingWaypoints-v3 TRAIN_CONFIG + gust wind of train_objlock.py:74-85", lambda K, dt: K.train_waypoints_v3_config(dtype=dt, wind_config=K.TRAIN_OBJLOCK_WIND), 94),
    "objlock": ("FixedwingObjLock TRAIN_CONFIG (train_objlock.py:27-86: dome 200 m, gust wind, camera 480 x 480 every 12 control steps)", lambda K, dt: K.train_objlock_config(dtype=dt), 203),
    "combined": ("Waypoint+ObjLock TRAIN_CONFIG (train_Fixedwing_Waypoints_ObjLock.py:35-92: 8 targets, 20 obstacles, camera every 6 control steps)", lambda K, dt: K.train_waypoint_objlock_config(dtype=dt), 193),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=30, help="timed regions of --steps launches each; the median is reported")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-collector", action="store_true", help="skip the informational `collector` object (rollout collector + PPO update rates)")
    ap.add_argument("--sweep", action="store_true", help="also print an N-sweep (stderr) 2^10..2^22")
    ap.add_argument("--task", default="waypoints", choices=sorted(TASKS),
                    help="step kernel to time (default: the headline FixedwingWaypoints workload of BASELINE.json)")
    return ap.parse_args()


def action_pool(n, dtype, device):
    g = torch.Generator(device="cpu").manual_seed(0)
    return [(torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).to(dtype).to(device) for _ in range(POOL)]


class Stepper:
    """Issues fw_step launches over the action pool, either eagerly or as replays of ONE captured hipGraph of
    `graph_len` launches (captured from the torch stream the C ABI launches on); a remainder that does not fill a
    replay is issued eagerly.  `counts` says what a run() really did, so the bench line can print it."""

    def __init__(self, env, pool, use_graph, graph_len=POOL):
        self.env, self.pool, self.graph = env, pool, None
        self.i = 0
        self.graph_len = int(graph_len) if use_graph else 0
        self.counts = {"replays": 0, "eager": 0}
        if self.graph_len > 0:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for a in pool[:2]:
                    env.step_tensor(a)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for j in range(self.graph_len):
                    env.step_tensor(pool[j % len(pool)])
            self.graph = g

    def actions_of(self, k):
        """The action tensors run(k) will use, in order (for checkers that step an oracle alongside)."""
        if self.graph is None:
            return [self.pool[(self.i + j) % len(self.pool)] for j in range(k)]
        full, rest = divmod(k, self.graph_len)
        one = [self.pool[j % len(self.pool)] for j in range(self.graph_len)]
        return one * full + one[:rest]

    def run(self, k):
        """Run exactly k steps."""
        if self.graph is None:
            for _ in range(k):
                self.env.step_tensor(self.pool[self.i % len(self.pool)]); self.i += 1
            self.counts["eager"] += k
            return
        full, rest = divmod(k, self.graph_len)
        for _ in range(full):
            self.graph.replay()
        for j in range(rest):
            self.env.step_tensor(self.pool[j % len(self.pool)])
        self.counts["replays"] += full; self.counts["eager"] += rest

    def describe(self, counts):
        if self.graph is None:
            return f"eager x{counts['eager']}"
        return f"hipGraph({self.graph_len} launches) x{counts['replays']} replays + {counts['eager']} eager"


def host_cores():
    """(threads to use, nproc, affinity, cgroup CPU quota or None): ALL host cores this job may run on -- the CPU affinity
    mask capped by the cgroup's CPU quota.  (The GPU boxes report 256 cores in the mask but give a one-GPU job a quota of
    16 CPUs; 256 OpenMP threads under that quota were measured at 33 k env-steps/s -- throttled spin-waits -- against
    4.9 M with 16, so the quota is what "all host cores" means there.)"""
    nproc = os.cpu_count() or 1
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    quota = None
    try:                                                             # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q = f.read().split()
        quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        pass
    cores = avail if quota is None else max(1, min(avail, int(quota + 0.5)))
    return cores, nproc, avail, quota


def cpu_baseline(cfg, n, seconds_target=20.0):
    """the C restatement of the env step (the PyBullet reference cannot run here) on ALL host cores
    the job may use on the GPU box (host_cores(): affinity mask capped by the cgroup CPU quota), OpenMP over envs, same
    N / scenarios / action pool as the GPU run; 5 repeats, median; plus one thread (`single_thread_value`, the shape of one
    reference worker).  Bounded to about `seconds_target` seconds of CPU work (the protocol's 2 000 timed vec-steps per
    repeat is the upper limit)."""
    from oracle import fw_oracle as O          # checker code used as the reported CPU baseline only
    # (-O3 -march=native build of the same C, made by main() before the GPU was touched; the strict build stays the checker)
    cores_all, nproc, avail, quota = host_cores()
    want = int(os.environ.get("FW_BENCH_THREADS", "0")) or cores_all
    env = O.OracleEnv(cfg, n, seed=42, fast=True)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = [(torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).numpy().astype(env.dtype) for _ in range(POOL)]
    obs = np.empty((n, env.obs_dim), env.dtype); rew = np.empty(n, env.dtype)
    te = np.empty(n, np.uint8); tr = np.empty(n, np.uint8); info = np.empty((n, 8), np.int32)

    def timed(budget_s, max_steps, k0):
        t0 = time.perf_counter(); k = 0
        while k < max_steps and (k < 8 or time.perf_counter() - t0 < budget_s):
            env.step_timed_only(acts[(k0 + k) % POOL], obs, rew, te, tr, info); k += 1
        return n * k / (time.perf_counter() - t0), k

    def at_threads(threads, share, max_steps):
        got = O.set_threads(max(1, threads), fast=True)
        timed(0.5, 50, 0)                                          # warm-up (thread team start, caches)
        reps = [timed(seconds_target * share / 5, max_steps, 100 + 2000 * r) for r in range(5)]
        return got, float(np.median([v for v, _ in reps])), reps

    cores, multi, reps = at_threads(want, 0.75, 2000)
    _, single, _ = at_threads(1, 0.25, 400)
    O.set_threads(cores, fast=True)
    return {"value": multi, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"median of 5 repeats x {reps[0][1]} vec-steps x {n} envs, same config / scenarios / action pool, OpenMP over envs on "
                      f"all {cores} host cores the job may use (affinity {avail} of nproc {nproc}, cgroup CPU quota {quota}; BASELINE.md section 3); "
                      f"C restatement built -O3 -march=native (not PyBullet: PyFlyt/pybullet are not installable here)",
            "single_thread_value": single, "nproc": nproc, "affinity": avail, "cgroup_cpu_quota": quota,
            "cpu_model": O._cpu_model(), "repeats": [v for v, _ in reps]}


def collector_rates(task, n):
    """Informational, NOT the metric: the other half of north_star on the same GPU -- the rollout collector (policy / value forward,
    env step and VecNormalize statistics of a vec-step in one launch: fw_collect_step; fw_collect_close per rollout) and the fused
    PPO update (fw_ppo_update), with the reference's hyper-parameters for this task (tools/bench_rollout.py is the full tool)."""
    import torch
    from pyflyt_drone_amd import rollout as R
    try:
        from pyflyt_drone_amd import config as K
        cfg = TASKS[task][1](K, "float64")
        hp = dict(n_steps=16, batch_size=128, n_epochs=20) if task in ("waypoints", "waypoints_wind") else \
             dict(n_steps=8, batch_size=64, n_epochs=10) if task == "objlock" else dict(n_steps=8, batch_size=128, n_epochs=20)
        import pyflyt_drone_amd as P
        ppo = R.PPO(R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, n, seed=42)), R.PPOConfig(**hp))
        for _ in range(3):
            ppo.collect_rollouts()
        torch.cuda.synchronize()
        reps, t0 = 20, time.perf_counter()
        for _ in range(reps):
            ppo.collect_rollouts()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        T = hp["n_steps"]
        ppo.check_collect_status()                 # raises if a wait inside a fw_collect_step launch ran out (the rate would be void)
        ppo.train(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ppo.train(); torch.cuda.synchronize()
        upd = time.perf_counter() - t0
        roll = dt / reps
        n_mb = hp["n_epochs"] * (T * n // hp["batch_size"])
        # rooflines of the two learner kernels (tools/learner_accounting.py itemises the algorithmic bytes / flops).
        # Launch durations here are wall-clock shares of graph replays (a vec-step = one fw_collect_step launch + 1/T of the closing
        # launch; an update = one fw_ppo_update launch + its host side); profiles/rNN_learner_pmc.json (latest round) holds the rocprofv3 durations
        # and the counter traffic of the same kernels.
        from tools import learner_accounting as LA
        words = TASKS[task][2]
        traffic = {}
        try:
            import glob
            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_learner_pmc.json")))[-1]) as f:      # the latest round's
                kk = json.load(f)["runs"].get(task if n == 4096 else f"{task}_n{n}", {}).get("kernels", {})
            for name, e in kk.items():
                if "traffic_bytes_per_launch" in e:
                    traffic[name.split("_kernel")[0]] = e["traffic_bytes_per_launch"]["total"]
        except (OSError, KeyError, ValueError):
            pass
        acc_c, acc_u = LA.collect_step(n, ppo.env.obs_dim, words), LA.ppo_update(n_mb, hp["batch_size"], ppo.env.obs_dim)
        roof = {"fw_collect_step": LA.roofline_hbm(acc_c["bytes"], dt * 1e6 / (reps * T), traffic.get("fw_collect"), acc_c["l2_served_bytes"]),
                "fw_ppo_update": LA.roofline_mfma(acc_u["mfma_flops"], upd * 1e6, acc_u["mfma_peak_tflops"], traffic.get("fw_ppo_update"))}
        roof["fw_collect_step"]["note"] = ("one launch per vec-step; `frac` is priced on the unique bytes (every buffer once); the parameter image and the "
                                           "statistics fetched again by every act wave never leave the L2 and are listed as l2_served_bytes "
                                           "(items: tools/learner_accounting.py); the launch is latency-bound (act chain in front of the env step)")
        roof["fw_ppo_update"]["note"] = (f"{n_mb} sequential minibatches in one launch on {acc_u['workgroups']} workgroups; peak = fp32 MFMA of those CUs "
                                         f"({acc_u['mfma_peak_tflops']:.2f} TFLOP/s); {acc_u['flops_per_minibatch'] / 1e6:.2f} MFLOP per minibatch")
        return {"note": "informational: rollout collector + PPO update on the same GPU (north_star's other half); `value` above is the physics-only metric",
                "envs": n, "one_launch_collect": bool(ppo._one_launch), "us_per_vec_step": dt * 1e6 / (reps * T),
                "collected_env_steps_per_s": reps * T * n / dt, "update_s": upd, "update_minibatches": n_mb, "us_per_minibatch": upd * 1e6 / n_mb,
                "update_l2_paths": ppo._fused.last_paths if ppo._fused is not None else None,
                "collect_fallbacks": int(ppo.collect_fallbacks),
                "end_to_end_env_steps_per_s": T * n / (roll + upd), "ppo": hp, "roofline": roof}
    except Exception as e:          # never let the informational part take the metric down
        return {"error": f"{type(e).__name__}: {e}"}


def collector_rates_sharded(task, n, world, rank, local_rank, barrier):
    """world > 1: the same informational object as `collector`, for the SHARDED job -- every rank owns n envs, n_steps holds the
    reference's 65 536 samples per update (rollout.n_steps_for: 16 / world steps of 4096 envs), statistics all-reduce once per
    rollout, ONE all-gather of the rollout shards per update, the update replicated on every rank.  What
    the number is for: DESIGN section 7 predicts "collection scales with the ranks, the replicated update does not" -- end to end
    stays near the one-GPU rate at the reference's 128-sample minibatches -- and this measures it on the job's own backend.
    Every rank calls it (collectives inside); max over ranks of the wall times."""
    import torch.distributed as td
    from pyflyt_drone_amd import rollout as R
    try:
        cfg = TASKS[task][1](K, "float64")
        T = R.n_steps_for(65536, n, world)
        hp = dict(n_steps=T, batch_size=128, n_epochs=20)
        ppo = R.PPO(R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, n, device=local_rank, seed=42, global_env_offset=rank * n)), R.PPOConfig(**hp))
        for _ in range(2):
            ppo.collect_rollouts(); ppo.train()
        def timed(fn, reps):
            barrier(); t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            barrier()
            t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64)
            if td.get_backend() == "nccl":
                t = t.to(ppo.device)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            return float(t.item())
        roll = timed(ppo.collect_rollouts, 10)
        def it():
            ppo.collect_rollouts(); ppo.train()
        ppo.train()                                    # (the rollouts above were collected without an update in between: start from a fresh pair)
        e2e = timed(it, 3)
        samples = T * n * world
        return {"note": "informational: sharded rollout collector + replicated PPO update (one all-gather of the rollout shards per update); "
                        "`value` above is the physics-only metric",
                "envs_per_rank": n, "world": world, "n_steps": T, "samples_per_update": samples, "one_launch_collect": bool(ppo._one_launch),
                "collect_fallbacks": int(ppo.collect_fallbacks), "us_per_vec_step": roll * 1e6 / T,
                "collected_env_steps_per_s": samples / roll, "iteration_s": e2e, "update_s_estimate": e2e - roll,
                "update_allgather_ms": float(ppo.allgather_ms), "update_allgather_bytes_per_rank": float(ppo.allgather_bytes),
                "replica_checksum": float(ppo.replica_checksum()), "end_to_end_env_steps_per_s": samples / e2e, "ppo": hp}
    except Exception as e:          # never let the informational part take the metric down
        return {"error": f"{type(e).__name__}: {e}"}


class Watchdog:
    """The sharded `collector` object is the one part of the line that runs collectives nobody could rehearse on the real fabric
    before the driver's scaling run.  If it has not come back after `seconds`, every rank leaves on its own clock -- rank 0
    prints the line it already has (without the object) first -- instead of sitting in a collective until RCCL's own timeout
    takes the measured physics line down with it."""

    def __init__(self, seconds, rank, line_fn):
        import threading
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self.rank, self.line_fn = rank, line_fn

    def _fire(self):
        if self.rank == 0:
            try:
                print(self.line_fn(), flush=True)
            finally:
                os._exit(0)
        time.sleep(2.0)
        os._exit(0)

    def __enter__(self):
        self._t.start(); return self

    def __exit__(self, *a):
        self._t.cancel()


def self_launch(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE): start the N rank processes ourselves -- one per GPU, the same
    env contract torch.distributed.run provides, rendezvous on 127.0.0.1 -- BEFORE this process has made any GPU call (it never
    makes one: a process that has initialised the GPU must not start other programs on the GPU boxes), forward their output
    (only rank 0 prints the JSON line) and exit with the worst return code."""
    import socket
    import subprocess
    n = args.gpus
    if not os.environ.get("FW_BENCH_SINGLE_DEVICE") and not os.environ.get("FW_BENCH_DRY"):
        have = torch.cuda.device_count()             # counting devices does not initialise the GPU on this image
        if have < n:
            print(json.dumps({"error": f"--gpus {n} requested but only {have} HIP device(s) are visible", "n_gpus_requested": n}), flush=True)
            return 3
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FW_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p_ in list(pending):
                code = p_.poll()
                if code is None:
                    continue
                pending.remove(p_)
                if code != 0:
                    rc = rc or code
                    for q in pending:                    # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    return rc


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # never report a different job than the one asked for (e.g. `--gpus 8` under a 1-rank launcher)
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} does not match WORLD_SIZE={world}: start one rank per GPU "
                                       f"(torch.distributed.run --nproc-per-node {args.gpus}) or run `python bench.py --gpus {args.gpus}` bare",
                              "n_gpus_requested": args.gpus, "world_size": world}), flush=True)
        sys.exit(3)
    # FW_DIST_FORCE=1 (tests, one-GPU box): a ONE-rank job still goes through the process group and every collective below --
    # the only way to run the RCCL branches (device tensors, all_gather_into_tensor) without a second GPU
    dist = world > 1 or (bool(os.environ.get("FW_DIST_FORCE")) and env_world is not None)
    dry = bool(os.environ.get("FW_BENCH_DRY"))         # launcher rehearsal on a box without a GPU (tests only): no env, no timing
    if not args.no_cpu_baseline and world == 1 and not dry:
        # the CPU-baseline library is compiled for THIS machine's cores (-march=native); do it before anything touches the
        # GPU: a process that has initialised the GPU must not start other programs (make / gcc) on the GPU boxes
        from oracle import fw_oracle as _O
        _O.build_fast()
    if not dry and not torch.cuda.is_available():
        print(json.dumps({"error": "no HIP device: pyflyt_drone_amd has no CPU fallback"}))
        sys.exit(2)
    # Rehearsal knobs for a one-GPU box (never set by the driver): FW_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # FW_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device) for the barrier / max-reduce.
    if os.environ.get("FW_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("FW_BENCH_BACKEND", "nccl")
    if not dry:
        torch.cuda.set_device(local_rank)
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)
    def count_ranks(device):
        """How many ranks really take part: an all-reduce(SUM) of ones over the job's backend -- on the GPUs when it is RCCL
        (the driver's scaling run is the only place N > 1 ranks ever meet over xGMI; this is the collective it can check)."""
        ones = torch.ones(1, dtype=torch.float64, device=device)
        if dist:
            td.all_reduce(ones, op=td.ReduceOp.SUM)
        return int(round(float(ones.item())))

    def gather_per_rank(x, device):
        """[x of rank 0, x of rank 1, ...] on every rank (one small all-gather over the job's backend)."""
        if not dist:
            return [float(x)]
        mine = torch.tensor([float(x)], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(parts, mine)
        return [float(q.item()) for q in parts]

    backend_name = ("RCCL (torch.distributed 'nccl')" if backend == "nccl" else backend) if dist else "none (single process)"
    if dry:
        if os.environ.get("FW_BENCH_DRY_DIE_RANK") == str(rank):      # tests: a rank that dies must take the job down, not hang it
            os._exit(9)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        per_rank = [float(rank + 1)]
        if dist:
            td.barrier(); td.all_reduce(t, op=td.ReduceOp.MAX)
            per_rank = gather_per_rank(float(rank + 1), "cpu")
        seen = count_ranks("cpu")
        if rank == 0:
            print(json.dumps({"dry_run": True, "metric": "launcher rehearsal only (FW_BENCH_DRY): nothing was measured", "value": None,
                              "n_gpus": world, "ranks_seen": seen, "backend": backend_name, "per_rank": per_rank,
                              "max_rank_plus_one": float(t.item()), "steps": args.steps, "warmup": args.warmup}), flush=True)
        if dist:
            td.barrier(); td.destroy_process_group()
        if seen != args.gpus:
            sys.exit(4)
        return
    # every rank of the job, counted over the collective backend itself, before anything is measured
    ranks_seen = count_ranks(torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
    if ranks_seen != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"all-reduce over {backend_name} counted {ranks_seen} rank(s), --gpus asked for {args.gpus}",
                              "n_gpus_requested": args.gpus, "ranks_seen": ranks_seen, "backend": backend_name}), flush=True)
        if dist:
            td.destroy_process_group()
        sys.exit(4)
    n = args.envs_per_gpu
    task_name, task_cfg, task_words = TASKS[args.task]
    cfg = task_cfg(K, args.dtype)
    env = P.FixedwingVecEnv(cfg, n, device=local_rank, seed=42, global_env_offset=rank * n)
    env.reset_tensor()
    pool = action_pool(n, env.torch_dtype, env.device)
    # the graph holds min(64, steps) launches, so that the timed region is made of replays whatever --steps is
    stepper = Stepper(env, pool, use_graph=not args.no_graph, graph_len=max(1, min(POOL, args.steps)))

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    stepper.run(args.warmup)
    # `repeats` timed regions of EXACTLY `steps` launches, each between two barriers + synchronisations, max over ranks;
    # the median region is reported.  HIP events on the launch stream bracket the same launches for the kernel's own time.
    reps = max(1, args.repeats)
    walls, devs = [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timed = None
    for _ in range(reps):
        barrier()
        before = dict(stepper.counts)
        t0 = time.perf_counter()
        ev0.record()
        stepper.run(args.steps)
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        timed = {k: stepper.counts[k] - before[k] for k in before}
        if dist:
            t = torch.tensor([wall], dtype=torch.float64, device=env.device if backend == "nccl" else "cpu")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            wall = float(t.item())
        walls.append(wall); devs.append(ev0.elapsed_time(ev1))
    wall = float(np.median(walls))
    dev_ms = float(np.median(devs))
    # every rank's OWN clock beside the max-reduced one: the launch time its HIP events saw (median region), so that a slow GPU or a
    # rank that waits in the barriers shows in the scaling record instead of disappearing in the max
    per_rank_ms = gather_per_rank(dev_ms / args.steps, env.device if backend == "nccl" else "cpu")

    # Update-time exchange of a sharded training job: ONE all-gather of the rank's rollout shard
    # (obs 28 + action 4 + log-prob + advantage + return = 35 float32 per sample, 16 steps x 4096 envs = 65 536 samples per
    # rank, the reference's samples per update) -- timed here so that the scaling record carries it; not part of `value`.
    allgather = None
    if dist:
        try:
            T_ROLL = 16
            shard = torch.zeros((T_ROLL, n, env.obs_dim + 7), dtype=torch.float32, device=env.device)
            if backend == "nccl":
                out_g = torch.empty((world * T_ROLL, n, env.obs_dim + 7), dtype=torch.float32, device=env.device)
                run_g = lambda: td.all_gather_into_tensor(out_g, shard)
            else:
                hs = shard.cpu(); parts_g = [torch.empty_like(hs) for _ in range(world)]
                run_g = lambda: td.all_gather(parts_g, hs)
            for _ in range(3):
                run_g()
            barrier()
            tg = time.perf_counter()
            REPS = 10
            for _ in range(REPS):
                run_g()
            barrier()
            dtg = (time.perf_counter() - tg) / REPS
            tt = torch.tensor([dtg], dtype=torch.float64, device=env.device if backend == "nccl" else "cpu")
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            allgather = {"ms": float(tt.item()) * 1e3, "bytes_per_rank": shard.numel() * 4, "samples_per_rank": T_ROLL * n,
                         "collective": f"all_gather_into_tensor ({'RCCL' if backend == 'nccl' else backend}), once per PPO update",
                         "algbw_GBps": shard.numel() * 4 * (world - 1) / float(tt.item()) / 1e9}
        except Exception as e:                    # never lose the bench line to the side measurement
            allgather = {"error": repr(e)}

    word = 8 if args.dtype == "float64" else 4
    bytes_per_launch = task_words * word * n
    launch_s = dev_ms * 1e-3 / args.steps
    achieved = bytes_per_launch / launch_s / 1e9

    if args.sweep and rank == 0:
        for p in range(10, 23):
            m = 1 << p
            e2 = P.FixedwingVecEnv(cfg, m, device=local_rank, seed=42)
            e2.reset_tensor()
            a2 = (torch.rand((m, 4), device=e2.device, dtype=torch.float64) * 2 - 1).to(e2.torch_dtype)
            for _ in range(10):
                e2.step_tensor(a2)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sreps = 50
            s0.record()
            for _ in range(sreps):
                e2.step_tensor(a2)
            s1.record(); torch.cuda.synchronize()
            us = s0.elapsed_time(s1) * 1e3 / sreps
            print(f"[sweep] N=2^{p}={m}: lanes/env {e2.lanes_per_env} (waves/SIMD {e2.g8_waves if e2.lanes_per_env == 8 else 2}), {us:.1f} us/launch, {m / us:.1f} M env-steps/s, "
                  f"{task_words * word * m / us / 1e3:.1f} GB/s algorithmic", file=sys.stderr, flush=True)
            e2.close()

    # HBM bytes per launch from the PMC counters are collected offline (rocprofv3 --pmc cannot run inside this process:
    # tools/collect_profiles.sh); the newest committed summary is quoted when it was taken on this very workload, else null.
    traffic, traffic_src = None, None
    for rnd in ("r05", "r04", "r03", "r02"):
        pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if traffic is None and n == 4096 and args.dtype == "float64" and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                tt = json.load(f).get("tasks", {}).get(args.task)
            if tt:
                traffic = tt["total"]
                traffic_src = f"profiles/{rnd}_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE per launch, separate --pmc passes; gfx950 FETCH_SIZE correction x 2)"

    # The ceiling that really binds this kernel: 64-bit vector issue.  Lane-instructions per env-step come from the
    # SQ_INSTS_VALU counter of the shipped kernel (profiles/rNN_valu_count.json, made by tools/summarize_profiles.py:
    # wave-level VALU instructions per launch x 64 lanes / envs); achieved = that x env-steps per second.
    valu = None
    for rnd in ("r05", "r04", "r03", "r02"):
        vpath = os.path.join(ROOT, "profiles", f"{rnd}_valu_count.json")
        if valu is None and os.path.exists(vpath):
            with open(vpath) as f:
                vc = json.load(f).get(args.task if (args.dtype == "float64" and n == 4096) else "", None)
            if vc:
                lane_instr = vc["lane_instructions_per_env_step"]
                ach = lane_instr * n / launch_s / 1e12
                valu = {"bound": "valu64", "achieved": ach, "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s", "frac": ach / VALU_PEAK_TLANE,
                        "lane_instructions_per_env_step": lane_instr, "source": f"profiles/{rnd}_valu_count.json"}
    out = None
    sharded = None

    def line():
        o = dict(out)
        if dist and not args.no_collector:
            o["collector"] = sharded if sharded is not None else {"error": "the sharded collector did not come back within its time limit; the line was printed without it"}
        return json.dumps(o)

    if rank == 0:
        out = {
            "metric": "env-steps/sec at N parallel envs (FixedwingWaypoints)" if args.task == "waypoints"
                      else f"env-steps/sec at N parallel envs ({args.task}: not the BASELINE.json headline)",
            "value": world * n * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "backend": backend_name,
            "steps": args.steps,
            "warmup": args.warmup,
            "repeats": reps,
            "ms_per_step": wall * 1e3 / args.steps,
            "ms_per_step_min": min(walls) * 1e3 / args.steps,
            "ms_per_step_max": max(walls) * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "float64" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{task_name}, "
                                   f"{n} envs/GPU x {world} GPU, physics-only step(), motor noise + auto-reset on",
                       "envs_per_gpu": n, "obs_dim": env.obs_dim, "ticks_per_env_step": 8, "lanes_per_env": env.lanes_per_env, "waves_per_simd": (env.g8_waves if env.lanes_per_env == 8 else 2),
                       "launch": stepper.describe(timed), "timing": f"median of {reps} regions of {args.steps} launches, each between barrier + synchronize",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "fw_step_kernel", "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "words_per_env_step": task_words,
                         "valu": valu,
                         "note": "HBM is the nominal roofline of this element-wise path (SURVEY.md section 8d) and stays the primary object; "
                                 "the ceiling that BINDS is 64-bit vector issue / latency -- see `valu` (DESIGN.md section 6)"},
        }
        if allgather is not None:
            out["update_allgather"] = allgather
        if dist:
            out["per_rank_kernel_ms_per_step"] = per_rank_ms
            out["per_rank_kernel_ms_per_step_min"], out["per_rank_kernel_ms_per_step_max"] = min(per_rank_ms), max(per_rank_ms)
        if not args.no_cpu_baseline and world == 1:          # timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, n)
        if world == 1 and not args.no_collector and not args.no_cpu_baseline and n <= 8192:
            out["collector"] = collector_rates(args.task, n)
    if dist and not args.no_collector and n <= 8192:
        env.close()
        with Watchdog(float(os.environ.get("FW_BENCH_COLLECTOR_TIMEOUT", "150")), rank, line):
            sharded = collector_rates_sharded(args.task, n, world, rank, local_rank, barrier)
    if rank == 0:
        print(line(), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
