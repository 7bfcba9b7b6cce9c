"""Per-launch times of a trunk layer on the F(4x4,3x3)+depth-taps route and on the F(4x4x4,3x3x3) route (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.EcoLib(os.environ.get("ECO_LIB", hip.LIB_PATH))
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("res3", 128, 128, 16, 28, 28), ("res4", 256, 256, 8, 14, 14), ("res5", 512, 512, 4, 7, 7)]
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, cin, cout, D, H, W in shapes:
    x = torch.relu(torch.randn(B, cin, D, H, W, device=dev))
    res = torch.randn(B, cout, D, H, W, device=dev)
    y = torch.empty(B, cout, D, H, W, device=dev)
    yr = torch.empty(B, cout, D, H, W, device=dev)
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    bias = torch.randn(cout, device=dev)
    S = D * H * W
    ep = hip.ConvEpilogue()
    ep.bias = bias.data_ptr()
    ep.residual = hip.plain_view(res.data_ptr(), cout, S)
    ep.raw = hip.plain_view(yr.data_ptr(), cout, S)
    ep.act = hip.plain_view(y.data_ptr(), cout, S)
    ep.act2 = hip.null_view()
    ep.bn_scale, ep.bn_shift, ep.relu = sc.data_ptr(), sh.data_ptr(), 1
    th, tw, td = (H + 3) // 4, (W + 3) // 4, (D + 3) // 4
    p2 = lib.wgemm_plan(B, cin, cout, D, th, tw, 3)
    p3 = lib.wgemm_plan(B, cin, cout, td, th, tw, 1, None, points=216)
    for tag, p in (("2-D", p2), ("3-D", p3)):
        v = torch.empty(p.v_elems, device=dev)
        u = torch.randn(p.u_elems, device=dev) * 0.05
        m = torch.empty(p.m_elems, device=dev)
        if tag == "2-D":
            fi = lambda: lib.wino_input_pk_forward(p, x.data_ptr(), v.data_ptr(), H, W, s)
            fo = lambda: lib.wino_output_dm_forward(p, m.data_ptr(), H, W, ep, s)
        else:
            fi = lambda: lib.wino3_input_forward(p, x.data_ptr(), v.data_ptr(), D, H, W, s)
            fo = lambda: lib.wino3_output_forward(p, m.data_ptr(), D, H, W, ep, s)
        fg = lambda: lib.wgemm_forward(p, v.data_ptr(), u.data_ptr(), m.data_ptr(), s)
        ti = timeit(fi); tg = timeit(fg); to = timeit(fo)
        vb, mb = p.v_elems * 4 / 1e6, p.m_elems * 4 / 1e6
        xb = x.numel() * 4 / 1e6
        print(f"{name} {tag}: input {ti:.4f} ms ({(xb + vb) / ti / 1e3:.2f} TB/s)  gemm {tg:.4f} ms (bn={p.bn} ks={p.ksplit})  "
              f"output {to:.4f} ms ({(mb + 4 * xb * cout / cin) / to / 1e3:.2f} TB/s)  sum {ti + tg + to:.4f}", flush=True)
