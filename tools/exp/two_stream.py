"""Do res4a_1 and res4a_down (same input, same geometry, 1568 workgroups = 3.06 rounds each) finish sooner on two streams than
back to back on one?  (GPU box.)"""
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
    plan = lib.conv_plan(g)
    x = torch.relu(torch.randn(n, cin, *insp, device=dev))
    S = int(np.prod(outsp))
    legs = []
    for k in range(2):
        w = (np.random.default_rng(k).standard_normal((cout, cin, 3, 3, 3)) / 60).astype(np.float32)
        wp = np.empty(plan.wp_elems, np.float32); kt = np.empty(plan.ktab_elems, np.int32)
        lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
        dwp, dkt = torch.from_numpy(wp).to(dev), torch.from_numpy(kt).to(dev)
        y = torch.empty(n, cout, *outsp, device=dev)
        ws = torch.empty(max(plan.ws_bytes, 4) // 4, device=dev)
        bias = torch.randn(cout, device=dev)
        ep = hip.ConvEpilogue()
        ep.bias = bias.data_ptr()
        ep.residual, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
        ep.raw = hip.plain_view(y.data_ptr(), cout, S)
        ep.bn_scale = ep.bn_shift = None
        ep.relu = 0
        legs.append((dwp, dkt, y, ws, bias, ep))
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream()

    def run(two):
        a, b = legs
        if not two:
            lib.conv_forward(g, plan, x.data_ptr(), a[0].data_ptr(), a[1].data_ptr(), a[5], a[3].data_ptr(), s0.cuda_stream)
            lib.conv_forward(g, plan, x.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), b[5], b[3].data_ptr(), s0.cuda_stream)
            return
        ev = torch.cuda.Event(); ev.record(s0); s1.wait_event(ev)
        lib.conv_forward(g, plan, x.data_ptr(), a[0].data_ptr(), a[1].data_ptr(), a[5], a[3].data_ptr(), s0.cuda_stream)
        lib.conv_forward(g, plan, x.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), b[5], b[3].data_ptr(), s1.cuda_stream)
        ev2 = torch.cuda.Event(); ev2.record(s1); s0.wait_event(ev2)

    for two in (False, True, False, True):
        for _ in range(3):
            run(two)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for _ in range(20):
            run(two)
        e1.record(s0); torch.cuda.synchronize()
        print(f"{name}: {'two streams' if two else 'one stream '}  {e0.elapsed_time(e1) / 20:.4f} ms for the pair (ksplit {plan.ksplit}, bm {plan.bm} bn {plan.bn})", flush=True)
