"""Cycle stamps of stemb_kernel's phases per patch (probe build -DECO_STEMB_TS): 0 loop top, 1 reduction issued, 2 barrier,
3 next patch stored + loads issued, 4 window offsets done, then per m-tile i: 5+4i stage written, 6+4i barrier, 7+4i pooled +
stored, 8+4i barrier."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import eco_amd  # noqa
from eco_amd import hip
lib = hip.load()
raw = ctypes.CDLL(hip.LIB_PATH)
n, H, W, cout = 1024, 224, 224, 64
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-120, 130, size=(n, 3, H, W)).astype(np.float32)).cuda()
w = (rng.normal(size=(cout, 3, 7, 7)) / 12).astype(np.float32)
wp = np.zeros(lib.stemb_weight_elems(cout), np.uint16)
lib.stemb_pack_weights(w.ctypes.data, cout, wp.ctypes.data)
wpd = torch.from_numpy(wp.view(np.int16)).cuda()
b = torch.zeros(cout).cuda(); sc = torch.ones(cout).cuda(); sh = torch.zeros(cout).cuda()
y = torch.empty(n * cout * 56 * 56, dtype=torch.bfloat16).cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record()
    lib.stemb_forward(x.data_ptr(), wpd.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, y.data_ptr(), n, H, W, cout)
    e1.record()
torch.cuda.synchronize()
print(f"stemb {e0.elapsed_time(e1):.4f} ms")
ts = np.zeros(64 * 2 * 8 * 16, np.uint64)
assert raw.eco_stemb_ts_read(ts.ctypes.data_as(ctypes.c_void_p)) == 0
ts = ts.reshape(64, 2, 8, 16).astype(np.int64)
names = ["top", "reduced", "bar", "patch st+ld", "woff", "stage0", "bar", "pool0", "bar", "stage1", "bar", "pool1", "bar"]
for blk in (0, 9):
    for wv in (0,):
        print(f" block {blk} wave {3 * wv}: per patch, phase durations in cycles")
        for k in range(1, 5):
            r = ts[blk, wv, k]
            d = [int(r[i + 1] - r[i]) for i in range(12)]
            nxt = int(ts[blk, wv, k + 1][0] - r[12])
            print("  patch", k, f"[st+ld = wait {int(r[13] - r[2])} + cvt/ds_write {int(r[14] - r[13])} + load issue {int(r[3] - r[14])}]", " ".join(f"{nm}:{v}" for nm, v in zip(names[1:], d)), f"-> next top:{nxt}", f"| total {int(ts[blk, wv, k + 1][0] - r[0])}")
