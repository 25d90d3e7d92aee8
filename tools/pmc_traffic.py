"""Builds profiles/<tag>_pmc_traffic.json from two `rocprofv3 --kernel-trace --pmc ...` passes over

    python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-streams --v-only --burn-in-ms 0 --no-roofline

(FETCH_SIZE in one pass, WRITE_SIZE in the other: together they do not fit the TCC counter slots).  Units and the gfx950
correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB at the L2 <-> fabric boundary,
FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled here, WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [batches per gather launch = 8]

The replay gather in that trace is the launch that serves the next 8 V-learner steps at once (PQLVLearner._prefetch, algo.rng=auto).
"""
import collections
import csv
import json
import sys

MFMA = ("k_gemm", "k_mlp_fwd_fused", "k_fwd_narrow", "k_dx_slice")
GATHER = ("k_replay_gather_fast", "k_replay_gather_fused")


def per_kernel(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        tot[name] += float(r["Counter_Value"])
        calls[name] += 1
    return tot, calls


def family(tot, calls, keys):
    names = [n for n in tot if any(k in n for k in keys)]
    return sum(tot[n] for n in names), sum(calls[n] for n in names)


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    batches = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    # MFMA launch groups in the trace = V steps (warm-up, timed and the graph captures' dry runs) = launches of the optimiser kernel:
    # a --v-only run prepares and steps the V-learner only.  Each step runs the three fused forwards (target actor, target critic,
    # critic -- the last one carries the Q head's backward since round 3) + the critic backward's five GEMMs (the rollout's policy
    # forward of set-up is a fused launch too; its ~40 launches of 4096 rows stay in the family total: < 1 % of it)
    def groups(calls):
        n = sum(c for k, c in calls.items() if k.startswith("k_adamw"))
        assert n, "no optimiser launch in the trace"
        return n
    v_steps_f, v_steps_w = groups(fc), groups(wc)
    mf_f, _ = family(ft, fc, MFMA)
    mf_w, _ = family(wt, wc, MFMA)
    g_f, g_fc = family(ft, fc, GATHER)
    g_w, g_wc = family(wt, wc, GATHER)
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 48 --warmup 16 "
                  "--no-cpu-baseline --no-streams --v-only --burn-in-ms 0 --no-roofline",
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); counters sit at "
                "the L2 <-> fabric boundary, Infinity-Cache hits included",
        "mfma_launch_groups_in_trace": [v_steps_f, v_steps_w],
        "mfma_family_fetch_raw_kb": mf_f / v_steps_f,
        "mfma_family_write_kb": mf_w / v_steps_w,
        "mfma_family_per_v_step_bytes": (2.0 * mf_f / v_steps_f + mf_w / v_steps_w) * 1024.0,
        "gather_fetch_raw_kb": g_f / g_fc,
        "gather_write_kb": g_w / g_wc,
        "gather_per_launch_bytes": (2.0 * g_f / g_fc + g_w / g_wc) * 1024.0,
        "gather_batches_per_launch": batches,
        "gather_launches_in_trace": [g_fc, g_wc],
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
