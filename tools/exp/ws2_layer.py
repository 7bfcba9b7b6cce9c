"""Microbenchmark: res4a_1 | res4a_down on the stride-2 polyphase route (csrc/eco_wino_s2.hip), launch by launch, beside the
direct kernel (two conv_mfma launches + their split-K reduces).  ReLU'd random input.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from eco_amd import hip

lib = hip.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cin, couts, (Do, Ho, Wo) = 128, (256, 256), (8, 14, 14)
D, H, W = 2 * Do, 2 * Ho, 2 * Wo
ctot = sum(couts)
S = Do * Ho * Wo
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, label, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:50s} {ms:.4f} ms", flush=True)
    return ms


x = torch.relu(torch.randn(B, cin, D, H, W, device=dev))
plan = lib.wgemm_plan(B, 8 * cin, ctot, Do // 4, Ho // 7, Wo // 7, 1, None, points=320)
print(f"plan bm={plan.bm} bn={plan.bn} ks={plan.ksplit} V {plan.v_elems * 4 / 1e6:.0f} MB M {plan.m_elems * 4 / 1e6:.0f} MB U {plan.u_elems * 4 / 1e6:.0f} MB")
v = torch.empty(plan.v_elems, device=dev)
m = torch.empty(plan.m_elems, device=dev)
up = torch.randn(plan.u_elems, device=dev) * 0.01
bias = torch.randn(ctot, device=dev)
sc, sh = torch.rand(ctot, device=dev) + 0.5, torch.randn(ctot, device=dev)
res = torch.randn(B, couts[0], Do, Ho, Wo, device=dev)
y_raw = [torch.empty(B, c, Do, Ho, Wo, device=dev) for c in couts]
y_act = [torch.empty(B, c, Do, Ho, Wo, device=dev) for c in couts]
eps = []
c0 = 0
for k, c in enumerate(couts):
    ep = hip.ConvEpilogue()
    ep.bias = bias.data_ptr() + 4 * c0
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    if k == 0:     # res4a_1: BN + ReLU
        ep.bn_scale, ep.bn_shift, ep.relu = sc.data_ptr() + 4 * c0, sh.data_ptr() + 4 * c0, 1
        ep.act = hip.plain_view(y_act[k].data_ptr(), c, S)
    else:          # res4a_down: raw
        ep.raw = hip.plain_view(y_raw[k].data_ptr(), c, S)
    eps.append((c0, c, ep))
    c0 += c
t_in = timeit(lambda: lib.wino_s2_input_forward(plan, x.data_ptr(), v.data_ptr(), D, H, W, s), "input transform")
t_g = timeit(lambda: lib.wgemm_forward(plan, v.data_ptr(), up.data_ptr(), m.data_ptr(), s), "GEMM (320 points, K = 1024, 512 rows)")
t_o = [timeit(lambda c0=c0, c=c, ep=ep: lib.wino_s2_output_forward(plan, m.data_ptr(), c0, c, Do, Ho, Wo, ep, s), f"output transform [{c0}, {c0 + c})")
       for c0, c, ep in eps]
print(f"pair: {t_in + t_g + sum(t_o):.4f} ms")
vb, mb = plan.v_elems * 4, plan.m_elems * 4
xb = x.numel() * 4
print(f"input transform {(xb + vb) / t_in / 1e9:.2f} TB/s; output {(mb / 2 + B * 256 * S * 4) / t_o[0] / 1e9:.2f} TB/s")

# the direct kernel on the same problem
for c in couts[:1]:
    g = hip.conv_geom(B, cin, c, (D, H, W), (3, 3, 3), (2, 2, 2), (1, 1, 1), (Do, Ho, Wo))
    p = lib.conv_plan(g, None)
    wp = torch.randn(p.wp_elems, device=dev) * 0.01
    kt_h = np.empty(p.ktab_elems, np.int32)
    w_h = (np.random.default_rng(0).standard_normal((c, cin, 3, 3, 3)) * 0.02).astype(np.float32)
    wp_h = np.empty(p.wp_elems, np.float32)
    lib.conv_pack_weights(g, p, w_h.ctypes.data, wp_h.ctypes.data, kt_h.ctypes.data)
    wp = torch.from_numpy(wp_h).to(dev); kt = torch.from_numpy(kt_h).to(dev)
    ws = torch.empty(max(p.ws_bytes // 4, 1), device=dev)
    ep = eps[0][2]
    timeit(lambda: lib.conv_forward(g, p, x.data_ptr(), wp.data_ptr(), kt.data_ptr(), ep, ws.data_ptr() if p.ws_bytes else None, s),
           f"direct conv_mfma (one conv, split-K {p.ksplit}, incl. reduce)")

# the same GEMM on N(0,1) operands (is the time data-dependent?)
v2 = torch.randn(plan.v_elems, device=dev)
up2 = torch.randn(plan.u_elems, device=dev)
timeit(lambda: lib.wgemm_forward(plan, v2.data_ptr(), up2.data_ptr(), m.data_ptr(), s), "GEMM, N(0,1) V and U")
timeit(lambda: lib.wgemm_forward(plan, v.data_ptr(), up2.data_ptr(), m.data_ptr(), s), "GEMM, transformed V, N(0,1) U")
timeit(lambda: lib.wgemm_forward(plan, v2.data_ptr(), up.data_ptr(), m.data_ptr(), s), "GEMM, N(0,1) V, 0.01 N(0,1) U")
print("V stats: zeros %.3f, |V| mean %.3f max %.1f" % ((v == 0).float().mean().item(), v.abs().mean().item(), v.abs().max().item()))
for bn, ks in ((128, 1), (128, 2), (256, 2)):
    q = lib.wgemm_plan(B, 8 * cin, ctot, Do // 4, Ho // 7, Wo // 7, 1, None, points=320)
    q.bn = bn; q.ksplit = ks
    m2 = torch.empty(320 * ks * ctot * plan.q, device=dev)
    timeit(lambda: lib.wgemm_forward(q, v.data_ptr(), up.data_ptr(), m2.data_ptr(), s), f"GEMM, real operands, bn={bn} ks={ks}")
