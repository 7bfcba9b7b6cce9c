"""Does a trunk layer run faster in chunks of images whose V / M fit the 256 MB Infinity Cache (same scratch reused per chunk)?
(GPU box.)  Input transform + GEMM + output transform (residual + raw + BN/ReLU epilogue) per chunk."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
B = 32
s = torch.cuda.current_stream().cuda_stream
for name, cin, cout, D, H, W in (("res3", 128, 128, 16, 28, 28), ("res4", 256, 256, 8, 14, 14), ("res5", 512, 512, 4, 7, 7)):
    x = torch.relu(torch.randn(B, cin, D, H, W, device=dev))
    res = torch.randn(B, cout, D, H, W, device=dev)
    y = torch.empty(B, cout, D, H, W, device=dev)
    yr = torch.empty(B, cout, D, H, W, device=dev)
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    bias = torch.randn(cout, device=dev)
    S = D * H * W
    th, tw, td = (H + 3) // 4, (W + 3) // 4, (D + 3) // 4
    for chunk in (32, 16, 8, 4, 2):
        p = lib.wgemm_plan(chunk, cin, cout, td, th, tw, 1, None, points=216)
        v = torch.empty(p.v_elems, device=dev)
        u = torch.randn(p.u_elems, device=dev) * 0.05
        m = torch.empty(p.m_elems, device=dev)
        eps = []
        for c0 in range(0, B, chunk):
            ep = hip.ConvEpilogue()
            ep.bias = bias.data_ptr()
            off = c0 * cout * S * 4
            ep.residual = hip.plain_view(res.data_ptr() + off, cout, S)
            ep.raw = hip.plain_view(yr.data_ptr() + off, cout, S)
            ep.act = hip.plain_view(y.data_ptr() + off, cout, S)
            ep.act2 = hip.null_view()
            ep.bn_scale, ep.bn_shift, ep.relu = sc.data_ptr(), sh.data_ptr(), 1
            eps.append((c0, ep))

        def layer():
            for c0, ep in eps:
                lib.wino3_input_forward(p, x.data_ptr() + c0 * cin * S * 4, v.data_ptr(), D, H, W, s)
                lib.wgemm_forward(p, v.data_ptr(), u.data_ptr(), m.data_ptr(), s)
                lib.wino3_output_forward(p, m.data_ptr(), D, H, W, ep, s)

        for _ in range(3):
            layer()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {chunk:2d} images per chunk ({B // chunk} x 3 launches, V+M {8 * (p.v_elems + p.m_elems) / 2 / 1e6:.0f} MB per chunk, "
              f"bn={p.bn} ks={p.ksplit}): {e0.elapsed_time(e1) / 10:.4f} ms per layer", flush=True)
