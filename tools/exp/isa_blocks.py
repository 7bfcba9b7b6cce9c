#!/usr/bin/env python
"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a `hipcc -S` listing (spill traffic included).
   python tools/exp/isa_blocks.py listing.s <substring of the mangled kernel name>"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().splitlines()
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if re.match(r"[A-Za-z_]\w*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print(lines[start], len(body), "lines")
    blocks, cur, name = [], [], "entry"
    for l in body:
        if re.match(r"\.LBB\d+_\d+:", l):
            blocks.append((name, cur))
            cur, name = [], l.rstrip(":")
        else:
            cur.append(l.strip())
    blocks.append((name, cur))
    tot = {}
    for name, b in blocks:
        ins = [l for l in b if l and not l.startswith((";", "."))]
        cnt = {}
        for l in ins:
            op = l.split()[0]
            k = ("mfma" if op.startswith("v_mfma") else "readlane" if "readlane" in op else "writelane" if "writelane" in op
                 else "scratch" if op.startswith("scratch") else "ds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer", "global"))
                 else "waitcnt" if op == "s_waitcnt" else "barrier" if op == "s_barrier" else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
            cnt[k] = cnt.get(k, 0) + 1
            tot[k] = tot.get(k, 0) + 1
        if cnt.get("mfma") or cnt.get("scratch", 0) > 4 or cnt.get("readlane", 0) + cnt.get("writelane", 0) > 8:
            print(f"{name:12s} {len(ins):5d}", dict(sorted(cnt.items())))
    print("total", dict(sorted(tot.items())))


if __name__ == "__main__":
    main()
