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
    # (name prefix, label, launches per V step, unit work per step, kind)
    table = [
        ("void k_mlp_fwd_fused<2, 2>", "twin-critic fused forward (target + current)", 2, 2 * (2.0 * B * 2 * macs(crit)), "flop"),
        ("void k_mlp_fwd_fused<1, 2>", "actor fused forward (+tanh, target noise)", 1, 2.0 * B * macs(actor), "flop"),
        ("void k_gemm<1, 128, 128", "dX GEMMs (+ELU') of hidden layers 3 and 2, both nets", 2,
         2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
        ("void k_gemm<2, 128, 128", "dW GEMMs of hidden layers 3 and 2, both nets (16 batch splits)", 2,
         2.0 * B * 2 * (h[2] * h[1] + h[1] * h[0]), "flop"),
        ("void k_gemm<2, 64, 64", "dW GEMM of layer 1, both nets", 1, 2.0 * B * 2 * (O + A) * h[0], "flop"),
        ("void k_skinny_bwd<1, 1>", "Q-head backward: dX + dW + db in one pass, both nets", 1, 2.0 * B * h[2] * 4 * 2, "byte"),
        ("k_reduce_slabs", "sum of the 16 dW slabs (+ head fold + sum g^2)", 1, params_c * 4.0 * 17, "byte"),
        ("k_adamw", "clip + AdamW + Polyak + re-pack", 1, params_c * 36.0, "byte"),
        ("void k_replay_gather_fast", "fused replay gather + normalise + cat", 1,
         B * ((2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4), "byte"),
        ("k_td_mse", "TD target + MSE loss + dL/dQ", 1, B * (4 * 4 + 2 * 4 + 2 * 4), "byte"),
    ]
    print(f"| kernel | role | launches / V step | avg us / launch | work / step | achieved | roof | frac |")
    print("|---|---|---|---|---|---|---|---|")
    tot_us = 0.0
    for prefix, label, n, work, kind in table:
        r = find(prefix)
        if r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        tot_us += n * us
        if kind == "flop":
            ach = work / (n * us * 1e-6) / 1e12
            print(f"| `{prefix.replace('void ', '')}…` | {label} | {n} | {us:.1f} | {work / 1e9:.2f} GFLOP | {ach:.1f} TFLOP/s | MFMA {PEAK_TF} | {ach / PEAK_TF:.2f} |")
        else:
            ach = work / (n * us * 1e-6) / 1e12
            print(f"| `{prefix.replace('void ', '')}…` | {label} | {n} | {us:.1f} | {work / 1e6:.2f} MB | {ach:.2f} TB/s | HBM {PEAK_TB} | {ach / PEAK_TB:.2f} |")
    print(f"\nSum of the listed launches: {tot_us:.0f} us per V step (the torch RNG launches and the graph's generator fills are not listed).")


if __name__ == "__main__":
    main()
