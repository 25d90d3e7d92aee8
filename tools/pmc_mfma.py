"""Matrix-pipe utilisation per kernel from one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
GRBM_GUI_ACTIVE` pass over `python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-streams --v-only --repeat 1`:

    python tools/pmc_mfma.py <counter_collection.csv> <out.json>

Per kernel (averaged over its dispatches): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE of one XCD x SIMDs) as
rocprofv3's derived metric of that name defines it (1024 SIMDs on an MI355X; a v_mfma_f32_32x32x2_f32 keeps its SIMD's matrix
pipe busy for 64 cycles), and the fp32 FLOPs the matrix pipe actually executed,
SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 -- padded tiles included, so it sits a little above the algorithmic count.
"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
XCDS = 8   # the CSV's GRBM_GUI_ACTIVE is the SUM over the 8 XCDs' counters (2.56 M for a 132-us kernel = 8 x 132 us x 2.43 GHz)


def main():
    path, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            calls[name] += 1
    res = {}
    for name, c in acc.items():
        busy, act, mops = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        if busy <= 0 or act <= 0:
            continue
        res[name] = {"dispatches": calls[name], "mfma_util": busy / (act / XCDS * SIMDS), "gui_active_cycles_per_dispatch": act / XCDS / calls[name],
                     "mfma_busy_cycles_per_dispatch": busy / calls[name], "gflop_per_dispatch_from_counters": mops * 512 / calls[name] / 1e9}
    doc = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- python3 bench.py "
                     "--steps 48 --warmup 16 --no-cpu-baseline --no-streams --v-only --repeat 1",
           "note": "mfma_util = MFMA-busy cycles / (GPU-active cycles x 1024 SIMDs): the fraction of matrix-pipe issue time in use, at "
                   "whatever clock the kernel ran; counters are sums over the dispatches of a kernel name", "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_dispatch"] * kv[1]["dispatches"]):
        print("%-40s n=%4d  MfmaUtil %.3f  %.2f GFLOP/dispatch" % (k[:40], v["dispatches"], v["mfma_util"], v["gflop_per_dispatch_from_counters"]))


if __name__ == "__main__":
    main()
