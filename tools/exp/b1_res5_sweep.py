"""One clip per step: res5's stride-1 3x3x3 convs (512 -> 512, 4x7x7; conv_span_kernel, split-K 62 by the plan) and the strided pairs
under other (bn, ksplit): what does the online step's 0.23 + 0.21 ms of direct launches cost at other split factors?  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
cases = [("res5b", 1, 512, 512, (4, 7, 7), 1), ("res5a pair", 1, 256, 1024, (8, 14, 14), 2), ("res4a pair", 1, 128, 512, (16, 28, 28), 2)]
for name, n, cin, cout, insp, stride in cases:
    outsp = tuple(s // stride for s in insp)
    g = hip.conv_geom(n, cin, cout, insp, (3, 3, 3), (stride,) * 3, (1, 1, 1), outsp)
    base = lib.conv_plan(g)
    print(name, "plan: bm", base.bm, "bn", base.bn, "ksplit", base.ksplit, "mode", base.mode, "split_tiles", base.split_tiles, flush=True)
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
        for ks in (base.ksplit, 8, 16, 24, 32, 48, 64, 96, 128):
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
                for _ in range(50):
                    f()
                e1.record(); torch.cuda.synchronize()
                mark = " <- plan" if (bn, ks) == (base.bn, base.ksplit) else ""
                print(f"{name} bn={bn} ksplit={ks}: {1e3 * e0.elapsed_time(e1) / 50:.1f} us  (vs first: {err:.1e}){mark}", flush=True)
            except hip.EcoError as e:
                print(f"{name} bn={bn} ksplit={ks}: {str(e)[:100]}", flush=True)
