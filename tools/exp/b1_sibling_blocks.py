"""Online recognition (fp32, one clip per step): the strided pairs as one direct launch (Engine.sibling_blocks) against two."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import eco_amd as caffe
from eco_amd import models, fillers
from eco_amd.netspec import NetSpec

N = 16
for B in (1, 2, 4):
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = torch.from_numpy(fillers.synthetic_frames(B * N, seed=1234)).cuda()
    outs = {}
    for sb in (False, True, False, True):
        net = caffe.Net(proto, caffe.TEST, params=params)
        net._engine.sibling_blocks = sb
        net._engine.build()
        net.set_input_device("data", x)
        for _ in range(20):
            net.forward_device()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            net.forward_device()
        e1.record(); torch.cuda.synchronize()
        outs[sb] = net.blobs[spec.outputs[0]].tensor.float().cpu().numpy().copy()
        print(f"B={B} sibling_blocks={sb}: {e0.elapsed_time(e1) / 200:.4f} ms per step; launches {len(net.op_labels())}", flush=True)
        del net
    print("  max |diff| / max|logit|:", float(np.abs(outs[True] - outs[False]).max() / np.abs(outs[False]).max()))
