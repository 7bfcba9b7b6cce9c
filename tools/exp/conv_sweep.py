"""(bn, split-K) sweep of the fp32 gather kernel on ECO-Full's small-N convolutions (GPU box)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
CASES = [("4e_3x3", 512, 128, 192, (14, 14), (3, 3), (2, 2), (1, 1), (7, 7)),
         ("4e_double_3x3_2", 512, 192, 256, (14, 14), (3, 3), (2, 2), (1, 1), (7, 7)),
         ("3c_double_3x3_2", 512, 96, 96, (28, 28), (3, 3), (2, 2), (1, 1), (14, 14)),
         ("3c_3x3", 512, 128, 160, (28, 28), (3, 3), (2, 2), (1, 1), (14, 14)),
         ("5b_pool_proj", 512, 1024, 128, (7, 7), (1, 1), (1, 1), (0, 0), (7, 7))]
s0 = torch.cuda.current_stream().cuda_stream
for name, n, cin, cout, insp, k, st, pd, outsp in CASES:
    g = hip.conv_geom(n, cin, cout, insp, k, st, pd, outsp)
    base = lib.conv_plan(g)
    S = int(np.prod(outsp))
    x = torch.relu(torch.randn(n, cin, *insp, device=dev))
    w = (np.random.default_rng(0).standard_normal((cout, cin) + k) / 30).astype(np.float32)
    y = torch.empty(n, cout, *outsp, device=dev)
    bias = torch.randn(cout, device=dev)
    ep = hip.ConvEpilogue()
    ep.bias = bias.data_ptr()
    ep.residual, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
    ep.raw = hip.plain_view(y.data_ptr(), cout, S)
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    print(f"{name}: plan bm={base.bm} bn={base.bn} ksplit={base.ksplit} split_tiles={base.split_tiles}", flush=True)
    for bn in (256, 128):
        if bn == 128 and base.bm != 128:
            continue
        for ks in (1, 2, 3, 4, 6):
            plan = lib.conv_plan(g)
            plan.bn, plan.ksplit = bn, ks
            tiles = math.ceil(cout / plan.bm) * math.ceil(n * S / bn)
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
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record(); torch.cuda.synchronize()
                print(f"   bn={bn} ksplit={ks} ({tiles * ks} workgroups): {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
            except hip.EcoError as e:
                print(f"   bn={bn} ksplit={ks}: {e}", flush=True)
