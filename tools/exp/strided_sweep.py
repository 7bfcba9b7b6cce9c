"""Strided 3x3x3 convs (res4a_1 / res5a_1 geometry): the gather kernel under other (bn, ksplit) than the plan's choice (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
for name, n, cin, cout, insp in (("res4a", 32, 128, 256, (16, 28, 28)), ("res5a", 32, 256, 512, (8, 14, 14))):
    outsp = tuple(s // 2 for s in insp)
    g = hip.conv_geom(n, cin, cout, insp, (3, 3, 3), (2, 2, 2), (1, 1, 1), outsp)
    base = lib.conv_plan(g)
    print(name, "plan:", base.bm, base.bn, base.ksplit, base.split_tiles, base.ws_bytes, flush=True)
    x = torch.relu(torch.randn(n, cin, *insp, device=dev))
    S = int(np.prod(outsp))
    w = (np.random.default_rng(0).standard_normal((cout, cin, 3, 3, 3)) / 60).astype(np.float32)
    y = torch.empty(n, cout, *outsp, device=dev)
    bias = torch.randn(cout, device=dev)
    ep = hip.ConvEpilogue()
    ep.bias = bias.data_ptr()
    ep.residual, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
    ep.raw = hip.plain_view(y.data_ptr(), cout, S)
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    ref = None
    s0 = torch.cuda.current_stream().cuda_stream
    for bn in (256, 128):
        for ks in (1, 2, 3, 4, 5, 6, 8):
            plan = lib.conv_plan(g)
            plan.bn, plan.ksplit = bn, ks
            tiles = -(-cout // plan.bm) * -(-(n * S) // bn)
            plan.split_tiles = tiles if ks > 1 else 0
            plan.ws_bytes = ks * n * cout * S * 4 + (1 << 20) if ks > 1 else 0
            wp = np.empty(plan.wp_elems, np.float32); kt = np.empty(plan.ktab_elems, np.int32)
            try:
                lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
                dwp, dkt = torch.from_numpy(wp).to(dev), torch.from_numpy(kt).to(dev)
                ws = torch.empty(max(plan.ws_bytes, 4) // 4, device=dev)
                f = lambda: lib.conv_forward(g, plan, x.data_ptr(), dwp.data_ptr(), dkt.data_ptr(), ep, ws.data_ptr(), s0)
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                got = y.clone()
                if ref is None:
                    ref = got
                err = float((got - ref).abs().max() / ref.abs().max())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record(); torch.cuda.synchronize()
                mark = " <- plan" if (bn, ks) == (base.bn, base.ksplit) else ""
                print(f"{name} bn={bn} ksplit={ks}: {e0.elapsed_time(e1) / 20:.4f} ms  (vs first: {err:.1e}){mark}", flush=True)
            except hip.EcoError as e:
                print(f"{name} bn={bn} ksplit={ks}: {e}", flush=True)
