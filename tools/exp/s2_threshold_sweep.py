"""Small batches: which position threshold should the polyphase forms have?  (Engine.wino_s2_min_positions; fp32, N = 16)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import eco_amd as caffe
from eco_amd import models, fillers
from eco_amd.netspec import NetSpec

N = 16
for B in (4, 8, 16, 24, 32):
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = torch.from_numpy(fillers.synthetic_frames(B * N, seed=1234)).cuda()
    for thr in (128, 256, 512, 128, 256):
        net = caffe.Net(proto, caffe.TEST, params=params)
        net._engine.wino_s2_min_positions = thr
        net._engine.build()
        net.set_input_device("data", x)
        reps = 200 if B <= 2 else 60
        for _ in range(10):
            net.forward_device()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net.forward_device()
        e1.record(); torch.cuda.synchronize()
        forms = sum("stride-2 winograd" in l and "input transform" in l for l in net.op_labels())
        print(f"B={B} min_positions={thr}: {e0.elapsed_time(e1) / reps:.4f} ms per step; polyphase groups {forms}; launches {len(net.op_labels())}", flush=True)
        del net
