"""Algorithmic work of the learner kernels, per launch -- the numerators of the `roofline` objects in bench.py's `collector`
and of profiles/rNN_learner_pmc.json (tools/summarize_learner_pmc.py).  Peaks: MI355X_MICROARCH.md (HBM 8 TB/s; fp32 MFMA
157.3 TFLOP/s per chip = 0.6145 TFLOP/s per CU, 256 CUs).

"Algorithmic" = what the arithmetic needs to be moved once, whoever holds it in cache: every figure below is a count of
array elements x element size, itemised so that a reader can recompute it.  Two totals per kernel since round 5:
  * `unique_bytes`    -- every buffer counted ONCE per launch: the figure an HBM roofline may be priced on (`bytes` is this);
  * `l2_served_bytes` -- re-reads of a buffer that is already counted (the parameter image and the statistics, fetched again by
                         every act wave): they never leave the L2, so they are reported under their own name and are NOT part of
                         the HBM fraction (round 4 added them in: 7.4 % "of HBM" for a launch whose counter traffic was 0.54 x
                         that sum).
"""
HBM_PEAK_GBPS = 8000.0
MFMA_F32_TFLOPS_CHIP = 157.3
CUS = 256
H = 64                      # hidden width of the reference's MlpPolicy


def net_param_bytes(D: int, KO: int) -> int:
    Dp = (D + 1) & ~1
    return 4 * (Dp * H + H + H * H + H + H * KO + KO)


def net_macs_forward(D: int, KO: int) -> int:
    return D * H + H * H + H * KO


def net_macs_backward(D: int, KO: int) -> int:
    """dWo, G2 = g Wo^T, dW2, G1 = G2 W2^T, dW1 (no input gradient)."""
    return 2 * H * KO + 2 * H * H + D * H


def collect_step(N: int, D: int, step_words: int, env_word: int = 8, rows_per_act_wave: int = 16) -> dict:
    """One fw_collect_step launch = one vec-step of the collector: act waves (policy + value forward of every env, buffer rows,
    finalisation of the previous step), the env step, the VecNormalize partial sums and their fold."""
    n_chunks = -(-N // rows_per_act_wave)
    nblk = -(-N // 8)
    image = net_param_bytes(D, 4) + 16 + net_param_bytes(D, 1)      # both networks + log_std
    stats = (4 * D + 8) * 8                                         # mean, var, totals: 4 D + 8 doubles
    it = {
        "env_step (fw_step's words per env-step x N)": step_words * env_word * N,
        "raw observations read (the policy and the value wave of a chunk read the same rows: counted once)": N * D * env_word,
        "previous step read by the value waves (reward, two flags)": N * (env_word + 2),
        "rollout-buffer rows written (obs D, action 4, log-prob, value, reward, start: float32)": N * (D + 8) * 4,
        "clipped actions written for the step waves": N * 4 * env_word,
        "parameter image (policy net, value net, log_std), once": image,
        "statistics (mean, var, totals), once": stats,
        "partial sums written by the step waves and read by the fold waves": 2 * nblk * (2 * D + 2) * 8,
    }
    l2 = {
        "parameter image again by every further act wave (a policy and a value wave per 16-row chunk)": n_chunks * image - image,
        "statistics again by every further act wave": 2 * n_chunks * stats - stats,
        "raw observations again by the chunk's second act wave": N * D * env_word,
    }
    flops = 2 * N * (net_macs_forward(D, 4) + net_macs_forward(D, 1))
    return {"bytes": sum(it.values()), "unique_bytes": sum(it.values()), "l2_served_bytes": sum(l2.values()), "items": it,
            "l2_served_items": l2, "mfma_flops": flops}


def collect_close(N: int, D: int, T: int, env_word: int = 8, rows_per_act_wave: int = 16) -> dict:
    n_chunks = -(-N // rows_per_act_wave)
    it = {
        "GAE: values, rewards, episode starts read; advantages, returns written ([T, N] float32)": 5 * T * N * 4,
        "last observation read (raw) and written normalised": N * D * (env_word + 4),
        "last step read (reward, flags), last values / starts / rewards written": N * (env_word + 2 + 12),
        "parameter image of the value net, once": net_param_bytes(D, 1),
    }
    l2 = {"parameter image of the value net again by every further wave": (n_chunks - 1) * net_param_bytes(D, 1)}
    return {"bytes": sum(it.values()), "unique_bytes": sum(it.values()), "l2_served_bytes": sum(l2.values()), "items": it,
            "l2_served_items": l2, "mfma_flops": 2 * N * net_macs_forward(D, 1)}


def ppo_split(B: int, max_blocks: int = 8):
    """csrc/fwsim_ppo.hpp ppo_split: (samples per pass, blocks per network) of a B-sample minibatch."""
    def cut(ch):
        c = B // ch
        return ch, (8 if (c >= 8 and max_blocks >= 8) else 4 if c >= 4 else 2 if c >= 2 else 1)
    def cost(s):
        return ((B // s[0] + s[1] - 1) // s[1]) * {64: 10, 32: 6, 16: 4}[s[0]] + (1 if s[1] == 2 else 0)
    best = cut(16)
    if B % 32 == 0 and cost(cut(32)) <= cost(best):
        best = cut(32)
    if B % 64 == 0 and cost(cut(64)) <= cost(best):
        best = cut(64)
    return best


def ppo_update(n_mb: int, B: int, D: int) -> dict:
    """One fw_ppo_update launch = n_mb sequential minibatches of B samples through both networks, forward and backward."""
    macs = B * sum(net_macs_forward(D, ko) + net_macs_backward(D, ko) for ko in (4, 1))
    blocks = 2 * ppo_split(B)[1]
    it = {"gathered rows per minibatch (obs D, action 4, old log-prob, advantage, return: float32) + index": B * ((D + 7) * 4 + 4)}
    return {"bytes": n_mb * sum(it.values()), "items_per_minibatch": it, "mfma_flops": 2 * n_mb * macs, "flops_per_minibatch": 2 * macs,
            "workgroups": blocks, "mfma_peak_tflops": blocks * MFMA_F32_TFLOPS_CHIP / CUS}


def ppo_pack(n_mb: int, B: int, D: int) -> dict:
    """The parallel pre-pass of fw_ppo_update: every minibatch's rows gathered and written in walking order."""
    Dq = ((D + 3) & ~3) + 8
    it = {"rows gathered (obs D, action 4, old log-prob, advantage x 2 (statistics + row), return: float32) + index": n_mb * B * ((D + 8) * 4 + 4),
          "packed rows written ((D rounded up to 4) + 8 floats)": n_mb * B * Dq * 4}
    return {"bytes": sum(it.values()), "items": it, "mfma_flops": 0}


def render(N: int, res: int) -> dict:
    return {"bytes": N * 2 * res * res * 4, "items": {"mask + depth, float32 [N, 2, res, res] written": N * 2 * res * res * 4}, "mfma_flops": 0}


def roofline_hbm(bytes_per_launch: float, launch_us: float, traffic=None, l2_served_bytes: float = 0.0) -> dict:
    """`frac` is priced on the unique bytes only; `l2_inclusive_gbps` (unique + re-reads out of the L2, over the same time) is what
    the launch moves through its L2 and is not an HBM fraction."""
    ach = bytes_per_launch / (launch_us * 1e-6) / 1e9
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
           "algorithmic_bytes_per_launch": bytes_per_launch, "unique_bytes": bytes_per_launch, "launch_us": launch_us}
    if l2_served_bytes:
        out["l2_served_bytes"] = l2_served_bytes
        out["l2_inclusive_gbps"] = (bytes_per_launch + l2_served_bytes) / (launch_us * 1e-6) / 1e9
    if traffic:
        out["traffic_over_algorithmic"] = traffic / bytes_per_launch
    return out


def roofline_mfma(flops_per_launch: float, launch_us: float, peak_tflops: float, traffic=None) -> dict:
    ach = flops_per_launch / (launch_us * 1e-6) / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": peak_tflops, "unit": "TFLOP/s", "frac": ach / peak_tflops, "traffic": traffic,
            "algorithmic_flops_per_launch": flops_per_launch, "launch_us": launch_us}
