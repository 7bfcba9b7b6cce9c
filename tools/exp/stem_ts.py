"""Timeline of the stem kernel's phases for the two workgroups of a CU (probe build, ECO_STEM_PROBE=4)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import eco_amd as caffe
from eco_amd import hip
lib = hip.load()
n, H, W, cout = 512, 224, 224, 64
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-120, 130, size=(n, 3, H, W)).astype(np.float32)).cuda()
w = (rng.normal(size=(cout, 3, 7, 7)) / 12).astype(np.float32)
wp = np.empty(74 * cout * 2, np.float32)
lib.stem_pack_weights(w.ctypes.data, cout, wp.ctypes.data)
wpd = torch.from_numpy(wp).cuda()
b = torch.zeros(cout).cuda(); sc = torch.ones(cout).cuda(); sh = torch.zeros(cout).cuda()
y = torch.empty(n, cout, 56, 56).cuda()
for _ in range(3):
    lib.stem_forward(x.data_ptr(), wpd.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, y.data_ptr(), n, H, W, cout, max_workgroups=0)
torch.cuda.synchronize()
raw = ctypes.CDLL(hip.LIB_PATH)
ts = np.zeros(1024 * 64, np.uint64)
raw.eco_stem_probe_read(ts.ctypes.data_as(ctypes.c_void_p))
ts = ts.reshape(1024, 64).astype(np.int64)
for blk in (0, 5, 100):
    a, c = ts[blk], ts[blk + 256]
    t0 = min(a[0], c[0])
    print(f"block {blk} / {blk + 256} (same CU): stamps relative to the first, in units of 1000 ticks; per patch: start, reduction issued, barrier passed, epilogue done")
    for name, v in (("A", a), ("B", c)):
        print(" ", name, " ".join(f"{(int(t) - int(t0)) / 1000:.1f}" for t in v[:24]))
