"""Where a wave of fw_render spends its life: python tools/render_wave_profile.py  (on the GPU box, with a -DFW_RENDER_PROF build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFW_RENDER_PROF -o tools/_build/libfwsim_rprof.so pyflyt-drone_amd/csrc/fwsim.hip
    FWSIM_LIB=$PWD/tools/_build/libfwsim_rprof.so python tools/render_wave_profile.py > profiles/rNN_render_wave_profile.txt 2>&1)
Lane 0 of every wave leaves the cycle counter at six points (csrc/fwsim_render.hpp, FW_RP); fw_render of that build prints the mean
phase lengths of the set-up waves and of the others when FWSIM_RENDER_PROF_DUMP is set.  Timing only: the stamps cost a few stores."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K

for n in (1024, 4096, 16384):
    venv = P.FixedwingVecEnv(K.train_waypoint_objlock_config(), n, seed=42)
    venv.reset_tensor()
    for res in (32, 64):
        out = torch.empty((n, 2, res, res), dtype=torch.float32, device=venv.device)
        os.environ.pop("FWSIM_RENDER_PROF_DUMP", None)
        for _ in range(3):
            venv.render_tensor(res, out=out)
        torch.cuda.synchronize()
        os.environ["FWSIM_RENDER_PROF_DUMP"] = "1"
        venv.render_tensor(res, out=out)
        torch.cuda.synchronize()
    os.environ.pop("FWSIM_RENDER_PROF_DUMP", None)
    venv.close()
