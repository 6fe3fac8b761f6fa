"""Dev tool: the two-wave camera kernels (step wave + capture wave) under the per-wave cycle accounting of tools/wave_profile.py.
usage: python tools/wave_profile_cw.py <objlock|combined> [steps]      (needs tools/_build/libfwsim_prof.so: bash tools/build_prof.sh)"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, _lib
_lib.LIB_PATH = os.environ.get("FW_PROF_LIB") or os.path.join(ROOT, "tools", "_build", "libfwsim_prof.so")
CFG = {"objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config}
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
st, hp = buf[:, :nblk, :], buf[:, nblk:, :]
slow = st[:, :, 0].argmax(axis=1)
S = st[np.arange(256), slow]; H = hp[np.arange(256), slow]
f = lambda a: f"{a.mean():9.0f}"
print(f"{which} N={N}: {nblk} workgroups/launch, last 256 of {steps} launches; cycles")
print(f"  step wave, mean        : total {f(st[:,:,0])} prologue {f(st[:,:,1])} reset {f(st[:,:,2])} physics {f(st[:,:,3])} waiting for frames {f(st[:,:,4])} epilogue {f(st[:,:,5])}  immediate sub-steps {st[:,:,8].mean():.2f}  requests {st[:,:,9].mean():.2f}")
print(f"  step wave, slowest     : total {f(S[:,0])} prologue {f(S[:,1])} reset {f(S[:,2])} physics {f(S[:,3])} waiting for frames {f(S[:,4])} epilogue {f(S[:,5])}  immediate sub-steps {S[:,8].mean():.2f}  requests {S[:,9].mean():.2f}")
print(f"  step wave, mean        : posting {f(st[:,:,7])} frame halves of the task logic {f(st[:,:,10])} frame-independent halves {f(st[:,:,11])}   slowest: {f(S[:,7])} {f(S[:,10])} {f(S[:,11])}")
print(f"  capture wave, mean     : busy {f(hp[:,:,0])} shadow work {f(hp[:,:,1])} captures {f(hp[:,:,2])} requests {hp[:,:,3].mean():.2f} alive {f(hp[:,:,4])}   with shadow work: {100.0 * (hp[:,:,1] > 2000).mean():.1f}% of waves")
print(f"  capture wave of slowest: busy {f(H[:,0])} shadow work {f(H[:,1])} captures {f(H[:,2])} requests {H[:,3].mean():.2f} alive {f(H[:,4])}")
print(f"  counters: {e.get_counters()}")
