"""Online recognition (fp32, N = 16, 1 / 2 / 4 clips per step): the size rules of the stride-1 Winograd routes
(Engine.wino_min_tiles: F(4x4,3x3) route at all; Engine.wino3_min_positions: F(4x4x4,3x3x3) instead of it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import eco_amd as caffe
from eco_amd import models, fillers
from eco_amd.netspec import NetSpec

N = 16
for B in (1, 2, 4):
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = torch.from_numpy(fillers.synthetic_frames(B * N, seed=1234)).cuda()
    for mt, m3 in ((64, 128), (16, 128), (256, 128), (1024, 128), (64, 32), (64, 16), (64, 512), (16, 16), (64, 128)):
        net = caffe.Net(proto, caffe.TEST, params=params)
        net._engine.wino_min_tiles = mt
        net._engine.wino3_min_positions = m3
        net._engine.build()
        net.set_input_device("data", x)
        for _ in range(10):
            net.forward_device()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            net.forward_device()
        e1.record(); torch.cuda.synchronize()
        labels = net.op_labels()
        print(f"B={B} wino_min_tiles={mt} wino3_min_positions={m3}: {e0.elapsed_time(e1) / 200:.4f} ms; launches {len(labels)}; "
              f"3-D routes {sum('F(4x4x4' in l and 'input' in l for l in labels)}, 2-D-tile routes {sum('F(4x4,3x3) input' in l for l in labels)}", flush=True)
        del net
