"""Sanity sweep on the GPU: ECO-Lite / ECO-Full at several (num_segments, clips) -- which polyphase forms the default plan takes and
that its logits agree with the all-direct plan (winograd=False) to fp32 rounding."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eco_amd as caffe
from eco_amd import models, fillers
from eco_amd.netspec import NetSpec

for variant, N, B in [("lite", 4, 32), ("lite", 8, 16), ("lite", 8, 5), ("lite", 16, 3), ("lite", 32, 4), ("lite", 12, 11), ("full", 8, 8), ("full", 16, 5)]:
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(B * N, seed=7)
    net = caffe.Net(proto, caffe.TEST, params=params)
    forms = [l.split("[")[1][:48] for l in net.op_labels() if "stride-2 winograd" in l and "input transform" in l]
    out = net.forward(data=x)[spec.outputs[0]].copy()
    del net
    ref = caffe.Net(proto, caffe.TEST, params=params, winograd=False).forward(data=x)[spec.outputs[0]]
    err = float(np.abs(out - ref).max() / np.abs(ref).max())
    print(f"{variant} N={N} B={B}: {len(forms)} polyphase groups {forms}; default vs direct {err:.2e}; top-1 equal {bool((out.argmax(1) == ref.argmax(1)).all())}", flush=True)
    assert np.isfinite(out).all() and err < 1e-4
