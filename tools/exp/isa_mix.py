#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -S x.hip -o x.s
    python tools/exp/isa_mix.py x.s <substring of the mangled kernel name> [min MFMAs per block]

Per block: MFMAs, VALU instructions (and how many separate runs they form between MFMAs -- an isolated VALU
instruction costs an fp32-MFMA stream ~7 cycles, one inside a run ~2-3, profiles/r03_notes.md), LDS, VMEM, SALU,
waitcnt/barrier counts.
"""
import re, sys

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop")): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[start].split(":")[0])
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) and not s.startswith(".LBB"): continue
        if s.startswith(".LBB"):
            blocks.append((name, cur)); name, cur = s.split(":")[0], []
            continue
        op = s.split()[0]
        if re.match(r"^[a-z_0-9]+$", op): cur.append((op, s))
    blocks.append((name, cur))
    tot = {}
    for name, ins in blocks:
        c = {}
        runs = 0; prev = None; scratch = 0
        for op, s in ins:
            k = classify(op); c[k] = c.get(k, 0) + 1
            if op.startswith("scratch_"): scratch += 1
            if k in ("valu", "acc") and prev not in ("valu", "acc"): runs += 1
            if k in ("mfma", "valu", "acc"): prev = k
        for k, v in c.items(): tot[k] = tot.get(k, 0) + v
        if c.get("mfma", 0) >= min_mfma:
            last = ins[-1][1] if ins else ""
            print(f"{name:12s} n={len(ins):5d} mfma={c.get('mfma',0):4d} valu={c.get('valu',0):4d} acc={c.get('acc',0):3d} valu_runs={runs:4d} "
                  f"lds={c.get('lds',0):4d} vmem={c.get('vmem',0):3d} scratch={scratch:3d} salu={c.get('salu',0):4d} wait={c.get('wait',0):3d}  | {last[:50]}")
    print("whole kernel:", tot)

if __name__ == "__main__":
    main()
