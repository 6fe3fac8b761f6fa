"""Dev tool: cycle split of fw_ppo_update (needs tools/_build/libfwsim_ppoprof.so built with -DFW_PPO_PROF)."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyflyt_drone_amd import _lib
if not os.environ.get("PPO_PRODUCT"):       # PPO_PRODUCT=1: time the product library (no cycle stamps: the split prints as zeros)
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libfwsim_ppoprof.so")
from pyflyt_drone_amd import rollout as R
L = _lib.lib()
CASES = [(28, 128, None), (28, 128, "32x4"), (28, 64, None), (56, 64, None), (28, 256, None), (28, 256, "64x4")]
if len(sys.argv) > 1:          # "D,B[,CHxN[,e0]]" ...   (e0: FWSIM_PPO_EARLY=0)
    CASES = [(int(a.split(",")[0]), int(a.split(",")[1]), ((a.split(",")[2] or None) if a.count(",") > 1 else None) if a.count(",") < 3 else ((a.split(",")[2] or None), a.split(",")[3]))
             for a in sys.argv[1:]]
for D, B, split in CASES:
    os.environ.pop("FWSIM_PPO_EARLY", None)
    if isinstance(split, tuple):
        split, extra = split
        if extra == "e0":
            os.environ["FWSIM_PPO_EARLY"] = "0"
        split = (split or "") and split
        tag = " early=0"
    else:
        tag = ""
    if split:
        os.environ["FWSIM_PPO_SPLIT"] = split
    else:
        os.environ.pop("FWSIM_PPO_SPLIT", None)
    S, n_mb = 65536, 2000
    P = L.fw_ppo_param_count(D)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    flat = torch.randn(P, device="cuda", generator=g) * 0.1
    m = torch.zeros(L.fw_ppo_moment_count(), device="cuda"); v = torch.zeros_like(m)
    obs = torch.randn((S, D), device="cuda", generator=g); act = torch.randn((S, 4), device="cuda", generator=g)
    lp = torch.randn(S, device="cuda", generator=g) - 4; adv = torch.randn(S, device="cuda", generator=g); ret = torch.randn(S, device="cuda", generator=g)
    perm = torch.randint(0, S, (n_mb * B,), device="cuda", generator=g, dtype=torch.int32)
    loss = torch.zeros(40, device="cuda")
    H = R._PpoHyper(lr=3e-4, clip_range=0.2, ent_coef=0.001, vf_coef=0.5, max_grad_norm=0.5, beta1=0.9, beta2=0.999, eps=1e-5, norm_adv=1, step0=0)
    ws = torch.zeros(int(L.fw_ppo_update_workspace_bytes(n_mb, B, D)), dtype=torch.uint8, device="cuda")
    def run():
        rc = L.fw_ppo_update(R._p(flat), R._p(m), R._p(v), R._p(obs), R._p(act), R._p(lp), R._p(adv), R._p(ret), R._p(perm), n_mb, B, D,
                             C.byref(H), R._p(loss), R._p(ws), ws.numel(), None)
        assert rc == 0
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l = loss.tolist()
    print(f"D={D} B={B} cut={split or 'default'}{tag}: {dt / n_mb * 1e6:.2f} us/minibatch; cycles/minibatch pi: exchange {l[3]:.0f} gather {l[4]:.0f} net {l[5]:.0f} norm+adam {l[6]:.0f} | V: exchange {l[7]:.0f} gather {l[8]:.0f} net {l[9]:.0f} norm+adam {l[10]:.0f} || pi finish: reductions {l[11]:.0f} hand-off {l[12]:.0f} norm {l[13]:.0f} tile Adam {l[14]:.0f} thread Adam (+ weight all-gather) + barrier {l[15]:.0f} || pi hand-off: stores + wait + barrier {l[16]:.0f} flag + next gather {l[17]:.0f} poll (thread 0) {l[18]:.0f} barrier {l[19]:.0f} loads {l[28]:.0f} sums {l[29]:.0f} || pi chunk phases (per minibatch): L1 {l[20]:.0f} L2 {l[21]:.0f} head {l[22]:.0f} dWo {l[23]:.0f} G2 {l[24]:.0f} dW2 {l[25]:.0f} G1 {l[26]:.0f} dW1 {l[27]:.0f} || pi closing: thread Adam {l[30]:.0f} wait for the partners' weights {l[31]:.0f} wait + fetch + LDS writes {l[32]:.0f} closing barrier {l[33]:.0f}", flush=True)
