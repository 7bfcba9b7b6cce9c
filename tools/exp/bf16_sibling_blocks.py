"""configs[4] (bf16, N=32, 32 clips): the residual blocks' strided first conv and projection shortcut as ONE LDS-DMA launch
(Engine.sibling_blocks) against two, in one process on one box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import eco_amd as caffe
from eco_amd import models, fillers
from eco_amd.netspec import NetSpec

N, B = 32, 32
proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
spec = NetSpec.from_prototxt(proto)
params = fillers.synthetic_params(spec)
x = torch.from_numpy(fillers.synthetic_frames(B * N, seed=1234)).cuda()
outs = {}
for sb in (False, True, False, True):
    net = caffe.Net(proto, caffe.TEST, params=params, dtype="bf16")
    net._engine.sibling_blocks = sb
    net._engine.build()
    net.set_input_device("data", x)
    for _ in range(5):
        net.forward_device()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net.forward_device()
    e1.record(); torch.cuda.synchronize()
    prof = net._engine.profile(3)
    strided = [(p["label"][:60], round(p["ms"], 4)) for p in prof if "res4a_1" in p["label"] or "res4a_down" in p["label"] or "res5a_1" in p["label"] or "res5a_down" in p["label"] or p["label"].startswith("res4a_2") or p["label"].startswith("res5a_2")]
    out = net.blobs[spec.outputs[0]].tensor.float().cpu().numpy().copy()
    outs[sb] = out
    print(f"sibling_blocks={sb}: {e0.elapsed_time(e1) / 20:.4f} ms per step; launches {len(prof)}; {strided}", flush=True)
    del net
print("max |diff| / max|logit|:", float(np.abs(outs[True] - outs[False]).max() / np.abs(outs[False]).max()))
