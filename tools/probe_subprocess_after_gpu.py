"""Probe (GPU box): may a process that has initialised the GPU start a child program?  Decides how multi-rank GPU tests
are launched (tests/conftest.py starts its helper ranks before the GPU is touched either way)."""
import subprocess, sys, torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
try:
    r = subprocess.run([sys.executable, "-c", "print('child ran')"], capture_output=True, text=True, timeout=60)
    print("rc", r.returncode, "out", r.stdout.strip(), "err", r.stderr.strip()[-300:])
except BaseException as e:
    print("refused:", repr(e))
