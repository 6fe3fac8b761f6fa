"""Dev tool: where does the slowest wave of each fw_step launch spend its cycles?

Builds nothing itself: expects tools/_build/libfwsim_prof.so (bash tools/build_prof.sh: fwsim.hip with -DFW_PROFILE, ISA-checked);
--phases expects libfwsim_prof_ph.so (bash tools/build_prof.sh phases) and splits the capture steps with 5+ envs due.
usage: python tools/wave_profile.py <waypoints|waypoints_wind|objlock|combined> [steps]
"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libfwsim_prof.so")
CFG = {"waypoints": K.train_waypoints_v3_config, "objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config,
       "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND)}
PHASES = "--phases" in sys.argv
if PHASES:
    sys.argv.remove("--phases"); _lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libfwsim_prof_ph.so")
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
N = int(os.environ.get("N", 4096))
e = P.FixedwingVecEnv(CFG[which](), N, seed=42); e.reset_tensor()
L = _lib.lib()
L.fw_debug_profile.restype = C.c_int32; L.fw_debug_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
nblk = L.fw_debug_profile(e._h, None, 1)
g = torch.Generator().manual_seed(0)
acts = [(torch.rand((N, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(64)]
for i in range(steps): e.step_tensor(acts[i % 64])
torch.cuda.synchronize()
buf = np.zeros((256, 2 * nblk, 12), dtype=np.int64)
L.fw_debug_profile(e._h, buf.ctypes.data, 1)
st, wk = buf[:, :nblk, :], buf[:, nblk:, 0]
tot = st[:, :, 0]
slow = tot.argmax(axis=1)
S = st[np.arange(256), slow]                      # slowest step wave of each launch
if PHASES:     # built with -DFW_PROFILE -DFW_PROFILE_PHASES: phases of capture_body in capture steps with 5+ envs due
    ne = st[:, :, 6].sum()
    print(f"{which}: capture steps with 5+ envs due: {ne / st[:, :, 6].size:.3f} per wave-step; cycles each: " + "  ".join(
        f"{nm} {st[:, :, k].sum() / max(1, ne):.0f}" for nm, k in (("set-up+screening", 1), ("duck", 3), ("ground", 4), ("clear+draw", 5), ("sums+means", 9))))
    sys.exit(0)
names = ["total", "prologue", "reset", "aviary", "task", "epilogue"]
cap_words = which in ("objlock", "combined")
print(f"{which} N={N}: {nblk} step waves/launch, last 256 of {steps} launches; cycles (s_memtime)")
print("  mean over all waves : " + "  ".join(f"{n} {st[:, :, k].mean():8.0f}" for k, n in enumerate(names)))
print("  slowest wave/launch : " + "  ".join(f"{n} {S[:, k].mean():8.0f}" for k, n in enumerate(names)))
it = st[:, :, 6] & 0xFF; nr = (st[:, :, 6] >> 8) & 0xFF; nh = (st[:, :, 6] >> 16) & 0xFF
if not cap_words:      # (the camera kernels use these words for their capture steps, below)
    print(f"  slowest wave reset split: terminal obs {S[:, 8].mean():.0f}  swap-in / begin_reset {S[:, 9].mean():.0f}  first compute_state {S[:, 10].mean():.0f}")
print(f"  loop iterations: mean {it.mean():.2f}  slowest-wave mean {(S[:, 6] & 0xFF).mean():.2f}  max {it.max()}")
print(f"  env resets/launch {nr.sum(axis=1).mean():.1f}  of which swapped-in shadows {nh.sum(axis=1).mean():.1f}")
print(f"  waves with a reset: {100.0 * (nr > 0).mean():.1f}%   with an in-kernel (fallback) reset: {100.0 * ((nr - nh) > 0).mean():.1f}%")
cap = st[:, :, 11] & ((1 << 48) - 1); ncap = st[:, :, 11] >> 48
if cap.any():
    print(f"  camera: envs capturing per wave-step {ncap.mean():.2f} of {N // nblk}; cycles in captures per wave-step: mean {cap.mean():.0f}  "
          f"slowest wave {(S[:, 11] & ((1 << 48) - 1)).mean():.0f}; waves with a capture {100.0 * (ncap > 0).mean():.1f}%")
    names_m = ["1 env due", "2", "3-4", "5+"]
    for idx, nm in zip((8, 9, 10, 7), names_m):
        v = st[:, :, idx]; cyc = v & ((1 << 40) - 1); ne = v >> 40
        vs = S[:, idx]; cs = vs & ((1 << 40) - 1); ns = vs >> 40
        if ne.sum():
            print(f"    capture steps with {nm}: {ne.mean():.2f} per wave-step, {cyc.sum() / max(1, ne.sum()):.0f} cycles each (all waves); "
                  f"slowest wave: {ns.mean():.2f} per step, {cs.sum() / max(1, ns.sum()):.0f} cycles each")
if wk.any():
    print(f"  shadow worker waves: busy {100.0 * (wk > 2000).mean():.1f}%  mean busy cycles {wk[wk > 2000].mean() if (wk > 2000).any() else 0:.0f}  max {wk.max()}")
