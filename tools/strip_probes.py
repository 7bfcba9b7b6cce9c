#!/usr/bin/env python
"""Remove the probe-build branches from a kernel source: every #if / #ifdef / #ifndef / #elif block whose condition is made
of probe macros only is resolved as the PRODUCT build resolves it (the macros undefined, value 0) and the dead branch is
deleted.  Used once, in round 5, to take the `-DECO_*_PROBE=bits` / `-DECO_*_TS` instrumentation of rounds 3-4 out of the
product translation units; the instrumentation lives on as tools/exp/probes.patch, which tools/exp/build_variant.sh applies to
a scratch copy of csrc/ before it compiles a probe variant.  `python tools/strip_probes.py in.hip out.hip`."""
import re
import sys

PROBES = {"ECO_EPI_PROBE_NOSTORE", "ECO_SPANP_PROBE", "ECO_SPANP_TS", "ECO_SPAN_PROBE", "ECO_STEMB_PROBE", "ECO_STEMB_TS",
          "ECO_STEM_PROBE", "ECO_WFUSED_PROBE", "ECO_WGEMM_PROBE"}
DIRECTIVE = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$")


def probe_only(cond: str) -> bool:
    ids = set(re.findall(r"[A-Za-z_]\w*", cond)) - {"defined"}
    return bool(ids) and ids <= PROBES


def evaluate(kind: str, cond: str) -> bool:
    cond = cond.split("//")[0].strip()
    if kind == "ifdef":
        return False
    if kind == "ifndef":
        return True
    e = re.sub(r"defined\s*\(\s*\w+\s*\)", "0", cond)
    e = re.sub(r"defined\s+\w+", "0", e)
    for p in PROBES:
        e = re.sub(r"\b%s\b" % p, "0", e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    return bool(eval(e, {"__builtins__": {}}))


def strip(lines):
    out = []
    # stack entries: dict(probe=bool, taken=bool (a branch of this chain was already kept), keep=bool (current branch kept))
    stack = []
    for ln in lines:
        m = DIRECTIVE.match(ln)
        emitting = all(s["keep"] for s in stack if s["probe"])
        if not m:
            if emitting:
                out.append(ln)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            cond = rest.split("//")[0].strip()
            if probe_only(cond):
                v = evaluate(kind, cond) if emitting else False
                stack.append(dict(probe=True, taken=v, keep=v))
            else:
                stack.append(dict(probe=False, taken=True, keep=True))
                if emitting:
                    out.append(ln)
        elif kind == "elif":
            top = stack[-1]
            if top["probe"]:
                outer = all(s["keep"] for s in stack[:-1] if s["probe"])
                cond = rest.split("//")[0].strip()
                assert probe_only(cond), ln
                v = (not top["taken"]) and outer and evaluate("if", cond)
                top["keep"] = v
                top["taken"] = top["taken"] or v
            elif emitting:
                out.append(ln)
        elif kind == "else":
            top = stack[-1]
            if top["probe"]:
                outer = all(s["keep"] for s in stack[:-1] if s["probe"])
                top["keep"] = (not top["taken"]) and outer
                top["taken"] = True
            elif emitting:
                out.append(ln)
        else:  # endif
            top = stack.pop()
            if not top["probe"] and all(s["keep"] for s in stack if s["probe"]):
                out.append(ln)
    assert not stack
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    with open(src) as f:
        lines = f.read().split("\n")
    with open(dst, "w") as f:
        f.write("\n".join(strip(lines)))
