mkdir -p gpurun_out/dma
timeout 600 python -m pytest tests/test_blocked.py tests/test_siblings.py tests/test_eco_full_size.py tests/test_reference_logits.py tests/test_stemb.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do for d in 0 1; do
  ECO_CONVB_DMA_BUF=$d python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/dma/d${d}_$r.txt
  echo "== dma_buf=$d $r $(grep Average gpurun_out/dma/d${d}_$r.txt | cut -c1-40)"
done; done
python - <<'PY'
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\s*(.*?)\s+forward:\s+([\d.]+) ms\..*\[(.*)\]", l)
        if m and "dma" in m.group(3): d[m.group(1).split('+')[0].split(' ')[0]] = float(m.group(2))
    return d
t = {v: [load(f"gpurun_out/dma/d{v}_{r}.txt") for r in (1, 2)] for v in (0, 1)}
for k in t[0][0]:
    a = min(q[k] for q in t[0]); b = min(q[k] for q in t[1])
    print(f"{k:34s} flat {a:.4f}  descriptor {b:.4f}  {100 * (b - a) / a:+.1f}%")
print("sum", sum(min(q[k] for q in t[0]) for k in t[0][0]), sum(min(q[k] for q in t[1]) for k in t[0][0]))
PY
