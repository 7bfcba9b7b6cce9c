"""Condense an eco_time.py report: time per kernel family and the total."""
import collections, re, sys
fam = collections.OrderedDict()
tot = 0.0
for line in open(sys.argv[1]):
    m = re.search(r"forward:\s+([\d.]+) ms\..*\[eco::(\w+)", line)
    if not m:
        if line.startswith("Average"):
            print(line.strip()[:60])
        continue
    ms, k = float(m.group(1)), m.group(2)
    if "input transform" in line: k += "(in)"
    f = fam.setdefault(k, [0.0, 0])
    f[0] += ms; f[1] += 1; tot += ms
for k, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:32s} {n:3d} launches {ms:8.3f} ms")
print(f"{'total':32s} {tot:8.3f} ms")
