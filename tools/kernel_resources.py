#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of libpqlk.so, read from the code objects' own metadata.

    python tools/kernel_resources.py [path/to/libpqlk.so]          # table, spilling kernels marked
    python tools/kernel_resources.py --check                        # exit 1 if any kernel spills or uses scratch

A kernel that spills keeps working and keeps passing every parity test; only its speed tells.  tests/test_host_cpu.py runs
`resources()` over the built library so a spill cannot come back unnoticed (round 3: k_mlp_fwd_fused<1,4> grew to 59 spilled
VGPRs when the TD head was added to its template).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pql_amd", "csrc", "libpqlk.so")
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def resources(lib=LIB):
    """-> {demangled kernel name: {key: int}} for every gfx950 kernel in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)   # llvm-objdump writes the bundles NEXT to its input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, f)], check=True, capture_output=True,
                                   text=True).stdout
            cur = None
            # one YAML map per kernel under amdhsa.kernels; '.name' is not the first key, so collect per "- " item
            for block in re.split(r"\n\s*- \.", notes):
                m = re.search(r"(?:^|\n)\s*\.?name:\s+(\S+)", block)
                if not m or ".vgpr_count" not in block and "vgpr_count:" not in block:
                    continue
                cur = {}
                for k in KEYS:
                    mk = re.search(rf"\.?{k}:\s+(\d+)", block)
                    if mk:
                        cur[k] = int(mk.group(1))
                out[m.group(1)] = cur
    names = list(out)
    filt = shutil.which("c++filt")
    dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True) if filt else None
    if dem is not None and dem.returncode == 0:
        pretty = dem.stdout.strip().split("\n")
        out = {re.sub(r"\(.*", "", p): out[n] for p, n in zip(pretty, names)}
    return out


def full_drains(lib=LIB, want=()):
    """-> {demangled kernel name: number of `s_waitcnt vmcnt(0)` instructions} for the kernels whose name contains one of `want`.
    The gather kernels are latency-bound chains of loads and stores on gfx9's ONE in-order vector-memory counter: their speed hangs
    on the compiler keeping counted waits.  Round 4: a wave-uniform branch added inside k_replay_gather_fast's per-record loop (and
    later a template flag + selects in place of a branch) made the compiler drain the queue before every store -- 2 -> 14 full
    drains, the same bits, 23 -> 28-30 us.  Parity tests cannot see that; this count can."""
    out = {}
    filt = shutil.which("c++filt")
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.M | re.S):
                name = m.group(1)
                if filt:
                    name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
                name = re.sub(r"\(.*", "", name)
                if any(w in name for w in want):
                    out[name] = len(re.findall(r"s_waitcnt vmcnt\(0\)", m.group(2)))
    return out


def spilling(res):
    """Kernels that go through scratch memory: spilled VGPRs or a private segment.  (SGPRs spilled into VGPR lanes --
    `sgpr_spill_count` with no scratch -- cost a v_readlane each and are listed, not failed.)"""
    return {k: v for k, v in res.items() if v.get("vgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0)}


def main(argv):
    check = "--check" in argv
    args = [a for a in argv if not a.startswith("--")]
    res = resources(args[0] if args else LIB)
    bad = spilling(res)
    print(f"{'kernel':70s} vgpr agpr sgpr  lds(B) scratch(B) vspill sspill")
    for k in sorted(res):
        v = res[k]
        print(f"{k[:70]:70s} {v.get('vgpr_count', 0):4d} {v.get('agpr_count', 0):4d} {v.get('sgpr_count', 0):4d} "
              f"{v.get('group_segment_fixed_size', 0):7d} {v.get('private_segment_fixed_size', 0):10d} "
              f"{v.get('vgpr_spill_count', 0):6d} {v.get('sgpr_spill_count', 0):6d}{'  <-- SCRATCH' if k in bad else ''}")
    print(f"{len(res)} kernels, {len(bad)} spilling / using scratch")
    return 1 if (check and bad) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
