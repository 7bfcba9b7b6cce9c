"""Cycle stamps of the persistent span kernel around an item boundary (probe build -DECO_SPANP_TS).
Per item of waves 0 and 3: slot 1+3t / 2+3t / 3+3t = before the wait / after the wait / after the barrier of tap t < 6
of the item's first group; 19 = before the epilogue, 20 = its stores issued."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import eco_amd  # noqa
from eco_amd import hip
lib = hip.load()
raw = ctypes.CDLL(hip.LIB_PATH)
CASES = {"conv2_3x3": (1024, 64, 192, (56, 56), (3, 3), (1, 1), (1, 1)),
         "res3b_1": (32, 128, 128, (32, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
         "inc3a_3x3": (1024, 64, 64, (28, 28), (3, 3), (1, 1), (1, 1))}
NUM_CU = int(os.environ.get("TS_NUM_CU", "0")) or None
for name in sys.argv[1:] or list(CASES):
    n, cin, cout, in_sp, kernel, stride, pad = CASES[name]
    out_sp = in_sp
    g = hip.conv_geom(n, cin, cout, in_sp, kernel, stride, pad, out_sp)
    plan = lib.convb_plan(g, hip.DT_BF16, NUM_CU)
    S = int(np.prod(out_sp))
    rng = np.random.default_rng(0)
    w = (rng.normal(size=(cout, cin) + kernel) / np.sqrt(cin * np.prod(kernel))).astype(np.float32)
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    wpd = torch.from_numpy(wp.view(np.int16)).cuda()
    x = torch.randn(n * cin * S, device="cuda")
    if os.environ.get("TS_RELU"):
        x = torch.relu(x)        # half zeros, as behind a ReLU: the clock the power limit holds is higher
    if os.environ.get("TS_ZERO"):
        x = torch.zeros_like(x)  # no toggling in the operands: what the clock does without the matrix pipes' switching power
    x = x.to(torch.bfloat16)
    y = torch.empty(n * cout * S, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(cout, device="cuda"); sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    ep = hip.ConvEpilogue()
    ep.bias = b.data_ptr(); ep.residual = hip.null_view(); ep.raw = hip.null_view()
    ep.bn_scale = sc.data_ptr(); ep.bn_shift = sh.data_ptr(); ep.relu = 1
    ep.act = hip.View(y.data_ptr(), (cout // 8) * S, 0, S, 1)
    ws = torch.empty(max(plan.ws_bytes, 4) // 4, device="cuda") if plan.ws_bytes else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        e0.record()
        lib.convb_forward(g, plan, x.data_ptr(), wpd.data_ptr(), ep, ws.data_ptr() if ws is not None else None)
        e1.record()
    torch.cuda.synchronize()
    print(f"== {name}: {hip.convb_kernel_name(plan)} pgrid {plan.pgrid} {e0.elapsed_time(e1):.4f} ms")
    ts = np.zeros(64 * 2 * 8 * 32, np.uint64)
    rc = raw.eco_spanp_ts_read(ts.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    ts = ts.reshape(64, 2, 8, 32).astype(np.int64)
    b0 = ts[:, 0]                       # wave 0 of every recorded block
    first = b0[:, 0, 1]
    last = np.array([max(int(b0[q, k, 20]) for k in range(8) if b0[q, k, 20] >= b0[q, 0, 1]) for q in range(64)])
    span = int(last.max() - first[first > 0].min())
    us = e0.elapsed_time(e1) * 1e3
    print(f"   block 0: first tap wait -> last epilogue end {int(last[0] - first[0])} ticks over a {us:.1f} us launch = {(last[0] - first[0]) / us / 1e3:.2f} ticks / ns")
    print(f"   first tap wait -> last epilogue end over the 64 recorded blocks: {span / 1e3:.1f} k ticks; launch {us:.1f} us -> >= {span / us:.0f} ticks/us")
    for blk in (0, 9):
        for wv in (0,):
            t = ts[blk, wv]
            print(f" block {blk} wave {3 * wv}: per item, ticks relative to the item's first stamp (x100 cycles)")
            for k in range(1, 4):
                r = t[k]
                if r[1] == 0:
                    continue
                base = t[k - 1][19]     # previous item's epilogue start
                def d(v): return f"{(int(v) - int(base)) / 100:7.1f}"
                line = f"  item {k}: epi_begin 0, stores_issued {d(t[k-1][20])} |"
                for tap in range(6):
                    line += f" t{tap}: wait {d(r[1 + 3 * tap])}->{d(r[2 + 3 * tap])} bar {d(r[3 + 3 * tap])} |"
                line += f" next epi {d(r[19])}"
                line += f" || tap4: bar {d(r[15])} dma {d(r[28])} frags {d(r[29])} mfma_issued {d(r[30])} next wait {d(r[16])}"
                e = t[k - 1]
                line += " || epilogue: " + " ".join(f"{(int(e[21 + q]) - int(e[19])) / 100:.1f}" for q in range(7) if e[21 + q])
                print(line)
