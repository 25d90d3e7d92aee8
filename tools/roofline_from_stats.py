"""Per-kernel roofline fractions of ONE learner step from a contention-free rocprofv3 summary.

    python tools/roofline_from_stats.py profiles/r04_a_v_only_kernel_stats.csv [--batch 8192 --obs 88 --act 16 --hidden 512,512,256]
    python tools/roofline_from_stats.py profiles/r04_a_p_only_kernel_stats.csv --p-only

Input: the `*_kernel_stats.csv` of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-streams --v-only | --p-only ...`
(`tools/profile_bench.sh`): one stream, so a kernel's average duration carries no cross-stream contention.  Output: a
markdown table -- algorithmic work per step (SURVEY 8(d) formulas: logical widths, no padding), launches per step, average
duration, achieved rate and the fraction of the roof that bounds the kernel (fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s;
MI355X_MICROARCH.md).  A template instantiation that serves several layer shapes in a step (the two dX GEMMs, the two big dW
GEMMs) is priced with the work of all its launches in a step over the sum of their durations; a launch that serves K steps (the
replay gather, the draws, round 4: the target policy's forward) with the work of those K steps over its duration.
"""
import argparse
import csv

PEAK_TF, PEAK_TB = 157.3, 8.0


def macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--obs", type=int, default=88)
    ap.add_argument("--act", type=int, default=16)
    ap.add_argument("--hidden", default="512,512,256")
    ap.add_argument("--p-only", action="store_true", help="the CSV is a P-learner trace (bench.py --p-only)")
    a = ap.parse_args()
    B, O, A = a.batch, a.obs, a.act
    h = [int(x) for x in a.hidden.split(",")]
    rows = {r["Name"]: r for r in csv.DictReader(open(a.csv))}

    def find(prefix):   # every instantiation that starts with the prefix, pooled (e.g. the two k_gemm<2, 64, 64, ...> of a P step)
        hits = [r for n, r in rows.items() if n.startswith(prefix)]
        if not hits:
            return None
        calls = sum(int(r["Calls"]) for r in hits)
        total = sum(float(r["TotalDurationNs"]) for r in hits)
        return {"Calls": calls, "AverageNs": total / calls}

    crit = [O + A] + h + [1]
    actor = [O] + h + [A]
    params_c = 2 * (macs(crit) + sum(crit[1:]))
    params_a = macs(actor) + sum(actor[1:])
    # steps in the trace = launches of the optimiser kernel (one per step); every other kernel's launches per step follow from
    # the CSV's own call counts (tools/profile_bench.sh traces with --burn-in-ms 0 --no-roofline: nothing but steps launches them)
    opt = find("k_adamw")
    steps = int(opt["Calls"]) if opt else 0
    per_batch = B * ((2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4)
    # (name prefix, label, work per LEARNER STEP, kind)
    if a.p_only:
        who = "P"
        table = [
            ("void k_mlp_fwd_fused<1, 2, false>", "actor fused forward (+tanh, action dropped into the critic's input; stashing)", 2.0 * B * macs(actor), "flop"),
            ("void k_mlp_fwd_fused<2, 2, false>", "frozen twin-critic fused forward on [obs, pi(obs)] (stashing, + compact Q)", 2.0 * B * 2 * macs(crit), "flop"),
            ("k_dpg_minnet_head", "DPG loss + partition by owning net + compact head dX (round 4: one launch)", B * (2 * 4 + 2 * h[2] * 4), "byte"),
            ("void k_gemm<1, 64, 64", "compact dX GEMMs (+ELU') of the critic's hidden layers 3 and 2: ONE net per sample (64 x 64 tiles)", 2.0 * B * (h[2] * h[1] + h[1] * h[0]), "flop"),
            ("void k_gemm<1, 128, 64", "compact dX GEMMs (+ELU') of the critic's hidden layers 3 and 2: ONE net per sample (128 x 64 tiles)", 2.0 * B * (h[2] * h[1] + h[1] * h[0]), "flop"),
            ("void k_dx_slice<", "action slice of the critic's layer-1 dX through tanh' + the ACTOR's head backward (round 4: one launch)",
             B * (h[0] + 2 * h[2]) * 4.0, "byte"),   # reads dZ1 (h0) + the actor's last hidden layer (h2), writes its dZ (h2): 0.5 GFLOP ride along
            ("void k_gemm<1, 128, 128", "dX GEMMs (+ELU') of the actor's hidden layers 3 and 2", 2.0 * B * (h[2] * h[1] + h[1] * h[0]), "flop"),
            ("void k_gemm<2, 128, 128", "dW GEMM of the actor's hidden layer 2 (16 batch splits)", 2.0 * B * h[1] * h[0], "flop"),
            ("void k_gemm<2, 64, 64", "dW GEMMs of the actor's layers 3 and 1", 2.0 * B * (h[2] * h[1] + O * h[0]), "flop"),
            ("k_reduce_slabs", "sum of the 16 dW slabs (+ head fold + sum g^2)", params_a * 4.0 * 17, "byte"),
            ("k_adamw", "clip + AdamW + re-pack (+ loss fold)", params_a * 28.0, "byte"),
            ("void k_replay_gather_obs", "obs gather + normalise into the actor's input tile (one launch per 4 steps)", B * (2 * O * 4 + 8), "byte"),
            ("void k_replay_gather_fused", "obs gather (generic kernel)", B * (2 * O * 4 + 8), "byte"),
            ("k_philox_draws", "randint draws of 4 steps (torch's numbers)", B * 8, "byte"),
        ]
    else:
        who = "V"
        table = [
            ("void k_mlp_fwd_fused<2, 2, false>", "twin-critic fused forward (target + current; the current one also runs the Q head's backward)", 2 * (2.0 * B * 2 * macs(crit)), "flop"),
            ("void k_mlp_fwd_fused<2, 2, true>", "target policy's fused forward (+tanh, target noise) for the next 8 steps: one launch per 8 steps (round 4)", 2.0 * B * macs(actor), "flop"),
            ("void k_mlp_fwd_fused<1, 2, false>", "actor fused forward (+tanh, target noise), per step", 2.0 * B * macs(actor), "flop"),
            ("void k_gemm<1, 128, 128", "dX GEMMs (+ELU') of hidden layers 3 and 2, both nets", 2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
            ("void k_gemm<2, 128, 128", "dW GEMMs of hidden layers 3 and 2, both nets (16 batch splits)", 2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
            ("void k_gemm<2, 64, 64", "dW GEMM of layer 1, both nets", 2.0 * B * 2 * (O + A) * h[0], "flop"),
            ("void k_gemm<2, 128, 64", "dW GEMM of layer 1, both nets", 2.0 * B * 2 * (O + A) * h[0], "flop"),
            ("void k_skinny_bwd<1, 1", "Q-head backward: TD target + MSE + dX + dW + db in one pass, both nets", 2.0 * B * h[2] * 4 * 2, "byte"),
            ("k_reduce_slabs", "sum of the 16 dW slabs (+ head fold + sum g^2)", params_c * 4.0 * 17, "byte"),
            ("k_adamw", "clip + AdamW + Polyak + re-pack (+ loss fold)", params_c * 36.0, "byte"),
            ("void k_replay_gather_fast", "fused replay gather + normalise + cat (one launch per 8 steps)", per_batch, "byte"),
            ("k_philox_draws", "randint + normal draws of 8 steps (torch's numbers)", B * 8 + B * A * 4, "byte"),
            ("k_td_mse", "TD target + MSE loss + dL/dQ", B * (4 * 4 + 2 * 4 + 2 * 4), "byte"),
        ]
    print(f"| kernel | role | launches / {who} step | avg us / launch | us / {who} step | work / step | achieved | roof | frac |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_us = 0.0
    for prefix, label, work, kind in table:
        r = find(prefix)
        if r is None or not steps:
            continue
        us = float(r["AverageNs"]) / 1e3
        n = int(r["Calls"]) / steps
        if prefix.startswith("void k_mlp_fwd_fused<1, 2, false>"):
            n = round(n)   # (the rollout's policy forward uses the same kernel during set-up)
            if n == 0:
                continue
        tot_us += n * us
        ach = work / (n * us * 1e-6) / 1e12
        unit, peak, scale, wu = ("TFLOP/s", PEAK_TF, 1e9, "GFLOP") if kind == "flop" else ("TB/s", PEAK_TB, 1e6, "MB")
        short = prefix.replace("void ", "").rstrip("<")
        print(f"| `{short}…` | {label} | {n:.3g} | {us:.1f} | {n * us:.1f} | {work / scale:.2f} {wu} | {ach:.2f} {unit} | "
              f"{'MFMA' if kind == 'flop' else 'HBM'} {peak} | {ach / peak:.2f} |")
    print(f"\n{steps} {who} steps in the trace; sum of the listed launches: {tot_us:.0f} us per {who} step.")


if __name__ == "__main__":
    main()
