"""Microbenchmark: the transformed-domain GEMM at the shapes a 3-D Winograd F(4x4x4,3x3x3) route would give it
(216 points, K = cin) beside today's (36 points, K = 3 cin).  Random operands (power!).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [  # name, cin, cout, D, H, W
    ("res3", 128, 128, 16, 28, 28), ("res4", 256, 256, 8, 14, 14), ("res5", 512, 512, 4, 7, 7)]


def plan(n, cin, cout, d, th, tw, kd, points):
    p = hip.WGemmPlan()
    lib._check(lib._dll.eco_wgemm_plan_create(n, cin, cout, d, th, tw, kd, points, 0, C.byref(p)))
    return p


def run(p, label, flops):
    v = torch.randn(p.v_elems, device=dev)
    u = torch.randn(p.u_elems, device=dev)
    m = torch.empty(p.m_elems, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.wgemm_forward(p, v.data_ptr(), u.data_ptr(), m.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.wgemm_forward(p, v.data_ptr(), u.data_ptr(), m.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{label:40s} bm={p.bm} bn={p.bn} ks={p.ksplit} stages={p.nstages}  {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s "
          f"(V {p.v_elems * 4 / 1e6:.0f} MB, M {p.m_elems * 4 / 1e6:.0f} MB, U {p.u_elems * 4 / 1e6:.0f} MB)", flush=True)
    return ms


for name, cin, cout, D, H, W in shapes:
    th, tw = (H + 3) // 4, (W + 3) // 4
    p2 = plan(B, cin, cout, D, th, tw, 3, 36)
    f2 = 2.0 * 36 * B * th * tw * D * cout * cin * 3
    t2 = run(p2, f"{name} 2-D F(4x4) + 3 depth taps", f2)
    td = (D + 3) // 4
    p3 = plan(B, cin, cout, td, th, tw, 1, 216)
    f3 = 2.0 * 216 * B * th * tw * td * cout * cin
    t3 = run(p3, f"{name} 3-D F(4x4x4)", f3)
    for bn in (128, 256):
        for ks in (1, 2):
            if ks > p3.nstages // 4:
                continue
            q = plan(B, cin, cout, td, th, tw, 1, 216)
            q.bn = bn; q.ksplit = ks
            q.m_elems = 216 * ks * cout * B * th * tw * td
            run(q, f"   forced bn={bn} ks={ks}", f3)
