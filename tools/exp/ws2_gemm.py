"""Microbenchmark: the transformed-domain GEMM at the shapes the stride-2 polyphase Winograd route gives it
(320 points, K = 8 cin, one position per 4x7x7 output tile), beside the direct strided kernel's time in the step
(res4a: 0.82 ms + reduce per conv, res5a: 0.43).  Random operands.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from eco_amd import hip

lib = hip.EcoLib(hip.LIB_PATH)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


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
    print(f"{label:44s} bm={p.bm} bn={p.bn} ks={p.ksplit} stages={p.nstages}  {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s "
          f"(V {p.v_elems * 4 / 1e6:.0f} MB, M {p.m_elems * 4 / 1e6:.0f} MB, U {p.u_elems * 4 / 1e6:.0f} MB)", flush=True)
    return ms


# name, cin, cout, output dims (Do, Ho, Wo)
for name, cin, cout, Do, Ho, Wo in [("res4a", 128, 256, 8, 14, 14), ("res4a pair", 128, 512, 8, 14, 14),
                                    ("res5a", 256, 512, 4, 7, 7), ("res5a pair", 256, 1024, 4, 7, 7)]:
    td, th, tw = Do // 4, Ho // 7, Wo // 7
    pts = 320
    p = plan(B, 8 * cin, cout, td, th, tw, 1, pts)
    f = 2.0 * pts * B * td * th * tw * cout * 8 * cin
    direct = 2.0 * B * Do * Ho * Wo * cout * cin * 27
    t = run(p, f"{name} 5x8x8 points, K = {8 * cin}", f)
    print(f"    direct-equivalent {direct / t / 1e9:.1f} TFLOP/s ({direct / 1e9:.1f} GFLOP direct, {f / 1e9:.1f} executed)")
    for bn in (128, 256):
        for ks in (1, 2, 4):
            q = plan(B, 8 * cin, cout, td, th, tw, 1, pts)
            q.bn = bn; q.ksplit = ks
            q.m_elems = pts * ks * cout * B * td * th * tw
            run(q, f"   forced bn={bn} ks={ks}", f)
    # F(4,2)^3 (125 points, 14 -> 16 / 7 -> 8 overhang) for comparison: same kernel, points is only grid.y
