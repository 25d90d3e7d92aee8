"""Per-kernel roofline fractions of ONE V-learner step from a contention-free rocprofv3 summary.

    python tools/roofline_from_stats.py profiles/r02_d_v_only_kernel_stats.csv [--batch 8192 --obs 88 --act 16 --hidden 512,512,256]

Input: the `*_kernel_stats.csv` of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-streams --v-only ...`
(`tools/profile_bench.sh`): one stream, so a kernel's average duration carries no cross-stream contention.  Output: a
markdown table -- algorithmic work per launch (SURVEY 8(d) formulas: logical widths, no padding), average duration, achieved
rate and the fraction of the roof that bounds the kernel (fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s; MI355X_MICROARCH.md).
A template instantiation that serves several layer shapes in a step (the two dX GEMMs, the two big dW GEMMs) is priced with
the work of all its launches in a step over the sum of their durations.
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
    a = ap.parse_args()
    B, O, A = a.batch, a.obs, a.act
    h = [int(x) for x in a.hidden.split(",")]
    rows = {r["Name"]: r for r in csv.DictReader(open(a.csv))}

    def find(prefix):
        hits = [r for n, r in rows.items() if n.startswith(prefix)]
        return hits[0] if hits else None

    crit = [O + A] + h + [1]
    actor = [O] + h + [A]
    params_c = 2 * (macs(crit) + sum(crit[1:]))
    # V steps in the trace = launches of the optimiser kernel (one per step); every other kernel's launches per step follow from
    # the CSV's own call counts (tools/profile_bench.sh traces with --burn-in-ms 0 --no-roofline: nothing but steps launches them)
    opt = find("k_adamw")
    steps = int(opt["Calls"]) if opt else 0
    per_batch = B * ((2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4)
    # (name prefix, label, unit work per V step, kind)
    table = [
        ("void k_mlp_fwd_fused<2, 2>", "twin-critic fused forward (target + current; round 3: the current one also runs the Q head's backward)", 2 * (2.0 * B * 2 * macs(crit)), "flop"),
        ("void k_mlp_fwd_fused<1, 2>", "actor fused forward (+tanh, target noise)", 2.0 * B * macs(actor), "flop"),
        ("void k_gemm<1, 128, 128", "dX GEMMs (+ELU') of hidden layers 3 and 2, both nets", 2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
        ("void k_gemm<2, 128, 128", "dW GEMMs of hidden layers 3 and 2, both nets (16 batch splits)",
         2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
        ("void k_gemm<2, 64, 64", "dW GEMM of layer 1, both nets", 2.0 * B * 2 * (O + A) * h[0], "flop"),
        ("void k_gemm<2, 128, 64", "dW GEMM of layer 1, both nets", 2.0 * B * 2 * (O + A) * h[0], "flop"),
        ("void k_skinny_bwd<1, 1", "Q-head backward: TD target + MSE + dX + dW + db in one pass, both nets", 2.0 * B * h[2] * 4 * 2, "byte"),
        ("k_reduce_slabs", "sum of the 16 dW slabs (+ head fold + sum g^2)", params_c * 4.0 * 17, "byte"),
        ("k_adamw", "clip + AdamW + Polyak + re-pack (+ loss fold)", params_c * 36.0, "byte"),
        ("void k_replay_gather_fast", "fused replay gather + normalise + cat (one launch per 8 steps)", per_batch, "byte"),
        ("k_philox_draws", "randint + normal draws of 8 steps (torch's numbers)", B * 8 + B * A * 4, "byte"),
        ("k_td_mse", "TD target + MSE loss + dL/dQ", B * (4 * 4 + 2 * 4 + 2 * 4), "byte"),
    ]
    print("| kernel | role | launches / V step | avg us / launch | us / V step | work / step | achieved | roof | frac |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_us = 0.0
    for prefix, label, work, kind in table:
        r = find(prefix)
        if r is None or not steps:
            continue
        us = float(r["AverageNs"]) / 1e3
        n = int(r["Calls"]) / steps
        if prefix == "void k_mlp_fwd_fused<1, 2>":
            n = round(n)   # (the rollout's policy forward uses the same kernel during set-up)
        tot_us += n * us
        ach = work / (n * us * 1e-6) / 1e12
        unit, peak, scale, wu = ("TFLOP/s", PEAK_TF, 1e9, "GFLOP") if kind == "flop" else ("TB/s", PEAK_TB, 1e6, "MB")
        print(f"| `{prefix.replace('void ', '')}…` | {label} | {n:.3g} | {us:.1f} | {n * us:.1f} | {work / scale:.2f} {wu} | {ach:.2f} {unit} | "
              f"{'MFMA' if kind == 'flop' else 'HBM'} {peak} | {ach / peak:.2f} |")
    print(f"\n{steps} V steps in the trace; sum of the listed launches: {tot_us:.0f} us per V step.")


if __name__ == "__main__":
    main()
