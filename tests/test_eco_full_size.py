"""Full-size ECO parity on a real MI355X (`-m gpu`), through the pycaffe-style Net -> C ABI.

* ECO-Lite at BASELINE.json configs[0] (num_segments=4, 1 clip, 224x224): logits and every
  materialised blob vs the CPU oracle, fused and unfused plans.
* ECO-Full (configs[3] topology) at num_segments=4, 1 clip.
* configs[1] size (num_segments=16, 32 clips): the oracle is too slow for 32 clips, so the check is
  (a) 1 clip vs the oracle and (b) size-independent properties: clips are independent units
  (batch row i == the same clip run alone; permuting clips permutes logits), and the fused and
  unfused plans agree.
Tolerance: 1e-3 relative to max|logit| (north_star), in practice ~1e-6.
"""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.net import Net
from eco_amd.netspec import NetSpec

pytestmark = pytest.mark.gpu
TOL = 1e-3


def relerr(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("variant", ["lite", "full"])
def test_eco_n4_b1_vs_oracle(variant):
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=4, num_clips=1)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(4)
    ref = orc.forward(spec, params, {"data": x}, keep="all")
    # default plan (a single N=4 clip runs every conv directly), the layer-by-layer plan, and the Winograd
    # route forced onto the 3-D trunk in both tile sizes
    for fuse, wino in ((True, True), (False, True), (True, 4), (True, 2)):
        net = Net(proto, params=params, fuse=fuse, winograd=wino)
        net.blobs["data"].data[...] = x
        out = net.forward()["fc8"]
        assert np.isfinite(out).all()
        assert relerr(out, ref["fc8"]) < TOL
        assert out.argmax() == ref["fc8"].argmax()
        worst = 0.0
        for name in net.blobs:
            if name in net._engine.tensors:
                got = net.blobs[name].data
                worst = max(worst, relerr(got, ref[name].reshape(got.shape)))
        assert worst < TOL, worst


def test_eco_lite_c2_properties():
    N, B = 16, 32
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    assert abs(spec.conv_fc_flops() / 1e9 - 2975.13) < 0.01
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(B * N)
    net = Net(proto, params=params)
    out = net.forward(data=x)["fc8"].copy()
    assert out.shape == (B, 400) and np.isfinite(out).all()
    scale = np.abs(out).max()
    # (a) one clip against the CPU oracle
    spec1 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=N, num_clips=1))
    ref0 = orc.forward(spec1, params, {"data": x[:N]})["fc8"]
    assert relerr(out[:1], ref0) < TOL and out[0].argmax() == ref0.argmax()
    # (b1) clips are independent: row 17 of the batch == that clip alone
    net1 = Net(models.eco_lite_deploy(num_segments=N, num_clips=1), params=params)
    alone = net1.forward(data=x[17 * N:18 * N])["fc8"]
    assert np.abs(alone[0] - out[17]).max() < 1e-5 * scale
    # (b2) permuting clips permutes the logits
    perm = np.random.default_rng(0).permutation(B)
    xp = x.reshape(B, N, 3, 224, 224)[perm].reshape(B * N, 3, 224, 224)
    outp = net.forward(data=xp)["fc8"]
    assert np.abs(outp - out[perm]).max() < 1e-5 * scale
    # (b3) the layer-by-layer plan agrees with the fused plan
    del net1
    netu = Net(proto, params=params, fuse=False)
    outu = netu.forward(data=x)["fc8"]
    assert np.abs(outu - out).max() < 1e-5 * scale


def test_eco_lite_n32_shapes():
    """configs[4] geometry (num_segments=32: r2Dto3D dim 32, global_pool 8x7x7; README.md:85-95), 2 clips, fp32."""
    proto = models.eco_lite_deploy(num_segments=32, num_clips=2)
    spec = NetSpec.from_prototxt(proto)
    assert spec.blob_shapes["res2b_bn"] == (2, 96, 32, 28, 28) and spec.blob_shapes["res5b_bn"] == (2, 512, 8, 7, 7)
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(64)
    out = Net(proto, params=params).forward(data=x)["fc8"]
    spec1 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=32, num_clips=1))
    ref = orc.forward(spec1, params, {"data": x[:32]})["fc8"]
    assert relerr(out[:1], ref) < TOL


def test_hipgraph_replay_matches_eager():
    """Net.forward_device(graph=True): the launch list replayed as one hipGraph gives bit-identical logits,
    also after a parameter edit (re-capture) -- the graph path is plumbing, not a different computation."""
    import torch
    proto = models.eco_lite_deploy(num_segments=4, num_clips=1)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    net = Net(proto, params=params)
    net.blobs["data"].tensor.copy_(torch.from_numpy(fillers.synthetic_frames(4)).cuda())
    net.forward_device()
    torch.cuda.synchronize()
    eager = net.blobs["fc8"].tensor.clone()
    for _ in range(3):
        net.forward_device(graph=True)
    torch.cuda.synchronize()
    assert torch.equal(net.blobs["fc8"].tensor, eager)
    net.params["fc8"][1].data[...] += 2.0
    net.forward_device(graph=True)
    torch.cuda.synchronize()
    assert torch.allclose(net.blobs["fc8"].tensor, eager + 2.0, rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_hipgraph_replay_bf16_with_dynamic_items():
    """The bf16 path under hipGraph replay: the persistent kernels draw their items from counter slots chosen at capture
    time and cleared by each launch's last workgroup, so every replay starts from zeroed counters -- logits bit-identical to
    the eager run over five replays (which workgroup computes a tile changes, what it computes does not)."""
    import torch
    proto = models.eco_lite_deploy(num_segments=16, num_clips=4)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    net = Net(proto, params=params, dtype="bf16")
    net.blobs["data"].tensor.copy_(torch.from_numpy(fillers.synthetic_frames(64)).cuda())
    net.forward_device()
    torch.cuda.synchronize()
    eager = net.blobs["fc8"].tensor.clone()
    for _ in range(5):
        net.forward_device(graph=True)
        torch.cuda.synchronize()
        assert torch.equal(net.blobs["fc8"].tensor, eager)


@pytest.mark.gpu
def test_caffe_time_style_report():
    """tools/eco_time.py (`caffe time` for the HIP path) runs and reports every launch of the fused plan."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "eco_time.py"), "--segments", "4", "--clips", "1",
                          "--iterations", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # 35 operators (conv1 + pool1 are one stem launch; since round 5 pool2 rides on conv2_3x3's fused Winograd launch); at a
    # single N=4 clip conv2_3x3 (4 x 14 x 14 tile positions per
    # point), the seven inception 3x3 convs and the three res3 convs (196 each) are above the Winograd size rule of 64
    # and take the Winograd route -- the eight 2-D ones as two launches (input transform, fused GEMM + output
    # transform), the res3 ones as three; res4 (32) and res5 (4) run direct; sibling 1x1 convs that share a launch
    # are joined by " | " in its label
    assert "Average Forward pass" in out.stdout
    assert out.stdout.count("forward:") + out.stdout.count(" | ") == 35 + 8 + 2 * 3
    assert out.stdout.count("partial window maxima") == 1
    assert out.stdout.count("winograd F(4x4,3x3)") == 2 * 11 and out.stdout.count("eco::wfused_kernel") == 8


def _fast_oracle(spec, params, x, **kw):
    """The CPU oracle with its convolutions on the compiled reference im2col + OpenBLAS sgemm (oracle/_ref) when
    that library travelled with the snapshot: same arithmetic, several times faster than the NumPy im2col."""
    try:
        import eco_ref
        if eco_ref.available():
            return orc.forward(spec, params, {"data": x}, conv_impl=lambda *a: eco_ref.convolution(*a), **kw)
    except Exception:
        pass
    return orc.forward(spec, params, {"data": x}, **kw)


def test_eco_full_c4_n16_b32():
    """BASELINE.json configs[3]: ECO-Full, num_segments=16, 32 clips.  Three clips of the batch against the CPU
    oracle (each run alone: 129 GFLOP per clip on the CPU), and the size-independent properties for the rest:
    row i of the batch == that clip alone on the GPU, permuting clips permutes logits."""
    N, B = 16, 32
    proto = models.eco_full_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    assert abs(spec.conv_fc_flops() / 1e9 - 4122.43) < 0.01
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(B * N)
    net = Net(proto, params=params)
    out = net.forward(data=x)["fc8"].copy()
    assert out.shape == (B, 400) and np.isfinite(out).all()
    scale = np.abs(out).max()
    proto1 = models.eco_full_deploy(num_segments=N, num_clips=1)
    spec1 = NetSpec.from_prototxt(proto1)
    for clip in (0, 13, 31):
        ref = _fast_oracle(spec1, params, x[clip * N:(clip + 1) * N])["fc8"]
        assert relerr(out[clip:clip + 1], ref) < TOL and out[clip].argmax() == ref.argmax(), clip
    net1 = Net(proto1, params=params)
    for clip in (5, 22):
        alone = net1.forward(data=x[clip * N:(clip + 1) * N])["fc8"]
        assert np.abs(alone[0] - out[clip]).max() < 1e-5 * scale, clip
    perm = np.random.default_rng(1).permutation(B)
    xp = x.reshape(B, N, 3, 224, 224)[perm].reshape(B * N, 3, 224, 224)
    outp = net.forward(data=xp)["fc8"]
    assert np.abs(outp - out[perm]).max() < 1e-5 * scale


# bf16 storage: 2^-9 relative rounding per stored activation / weight, ~40 stored tensors deep: a random walk of that depth
# gives ~6 * 2^-9 = 1.2e-2 of a logit's OWN magnitude; against the LARGEST logit the measured worst case is 3.9e-3.  Round 4
# stated 3e-2 (7.7x the measurement: a tile-edge bug corrupting a fraction of a percent of outputs would have passed).
BF16_TOL = 1e-2     # vs the plain fp32 reference, of the largest logit
BF16_TOL_Q = 3e-3   # vs the reference evaluated with the SAME storage rounding (accumulation order / double rounding only:
                    # measured 9.2e-4)


def test_eco_lite_c5_bf16_n32():
    """BASELINE.json configs[4]: ECO-Lite num_segments=32 (r2Dto3D 32x96x28x28, global_pool 8x7x7), bf16, 32 clips
    per GPU.  The oracle stays fp32; two comparisons for clip 0: (a) against the oracle run with the same storage
    rounding (weights and every stored activation rounded to bf16 where the blocked path rounds) -- what remains
    is accumulation order and double rounding; (b) FOUR clips (0, 11, 20, 31) against the plain fp32 oracle within the
    stated bf16 tolerance (1e-2 of the largest logit, 3e-3 for (a); measured values, top-1 and top-5 agreement printed).  Then the clip-independence / permutation properties at
    the full batch."""
    from eco_amd import blocked
    N, B = 32, 32
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    assert abs(spec.conv_fc_flops() / 1e9 - 5950.25) < 0.01
    assert spec.blob_shapes["res2b_bn"] == (B, 96, 32, 28, 28) and spec.blob_shapes["res5b_bn"] == (B, 512, 8, 7, 7)
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(B * N)
    net = Net(proto, params=params, dtype="bf16")
    out = net.forward(data=x)["fc8"].copy()
    assert out.shape == (B, 400) and np.isfinite(out).all()
    scale = np.abs(out).max()
    stored = {n for n, t in net._engine.tensors.items() if t.dt}
    spec1 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=N, num_clips=1))
    qp = {k: [blocked.bf16_round(b) if (i == 0 and spec1.layer(k).type == "Convolution") else b for i, b in enumerate(v)]
          for k, v in params.items()}
    ref_q = _fast_oracle(spec1, qp, x[:N], store_hook=lambda name, v: blocked.bf16_round(v) if name in stored else v,
                         input_hook=lambda name, v: blocked.bf16_round(v))["fc8"]
    e_q = relerr(out[:1], ref_q)
    # (b) four clips of the batch (first, two in the middle, last) against the plain fp32 oracle, each run alone on the
    # CPU: max rel err, top-1 and top-5 agreement per clip (round-3 verdict: one clip was thin for a 3e-2 tolerance)
    worst, top5, top1 = 0.0, [], []
    for clip in (0, 11, 20, 31):
        ref = _fast_oracle(spec1, params, x[clip * N:(clip + 1) * N])["fc8"]
        e = relerr(out[clip:clip + 1], ref)
        worst = max(worst, e)
        assert e < BF16_TOL, (clip, e)
        # top-1: equal, or the fp32 reference itself is a near-tie -- the class bf16 picks is within twice the measured
        # absolute error of the reference's maximum (random-init logits DO tie that closely: clip 11's top two are 5.8
        # apart at max|logit| 8578, and bf16 storage moves logits by up to ~30; measured on MI355X, round 4)
        abs_err = float(np.abs(out[clip] - ref[0]).max())
        got1, ref1 = int(out[clip].argmax()), int(ref.argmax())
        top1.append(got1 == ref1)
        assert ref[0, ref1] - ref[0, got1] <= 2.0 * abs_err, (clip, got1, ref1, float(ref[0, ref1] - ref[0, got1]), abs_err)
        top5.append(len(set(np.argsort(-out[clip])[:5]) & set(np.argsort(-ref[0])[:5])))
    print(f"bf16 N=32: rel err vs rounding-aware oracle (clip 0) {e_q:.3e}; vs fp32 oracle over clips 0/11/20/31 worst "
          f"{worst:.3e}, top-1 equal {top1} (else a reference near-tie within 2x the error), top-5 overlap {top5}; "
          f"max|logit| {np.abs(ref).max():.1f}")
    assert e_q < BF16_TOL_Q and min(top5) >= 4
    net1 = Net(models.eco_lite_deploy(num_segments=N, num_clips=1), params=params, dtype="bf16")
    alone = net1.forward(data=x[17 * N:18 * N])["fc8"]
    # a single clip gets other split-K factors: another fp32 summation order, so a stored bf16 value may round
    # the other way now and then -- far below the storage tolerance, but not bit-identical
    assert np.abs(alone[0] - out[17]).max() < 5e-3 * scale
    perm = np.random.default_rng(0).permutation(B)
    xp = x.reshape(B, N, 3, 224, 224)[perm].reshape(B * N, 3, 224, 224)
    outp = net.forward(data=xp)["fc8"]
    assert np.abs(outp - out[perm]).max() < 5e-3 * scale   # tile boundaries move with the clip order


def test_eco_lite_c2_direct_and_minimal_filtering_plans_agree():
    """configs[1] with every convolution evaluated directly (winograd=False: the reference's arithmetic up to summation order)
    against the default plan -- F(4x4,3x3), F(4x4x4,3x3x3) and, since round 6, the stride-2 polyphase routes of res4a / res5a --
    over all 32 clips: fp32 rounding only (1e-4 of the largest logit; measured value printed); both against the oracle on clip 0."""
    N, B = 16, 32
    proto = models.eco_lite_deploy(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    x = fillers.synthetic_frames(B * N)
    net = Net(proto, params=params)
    labels = net.op_labels()
    assert sum("stride-2 winograd" in l for l in labels) == 6 and sum("F(4x4x4,3x3x3) input" in l for l in labels) == 9, labels
    assert not any("conv_mfma" in (op[3].get("kernel") or "") for op in net._engine.ops)     # no direct strided launch is left
    out = net.forward(data=x)["fc8"].copy()
    scale = np.abs(out).max()
    spec1 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=N, num_clips=1))
    ref = _fast_oracle(spec1, params, x[:N])["fc8"]
    native = Net(proto, params=params, winograd=False).forward(data=x)["fc8"]
    print(f"default plan vs CPU oracle {relerr(out[:1], ref):.3e}; direct plan vs oracle {relerr(native[:1], ref):.3e}; "
          f"default vs direct over 32 clips {np.abs(out - native).max() / scale:.3e}")
    assert relerr(out[:1], ref) < TOL and relerr(native[:1], ref) < TOL and out[0].argmax() == ref.argmax()
    assert np.abs(out - native).max() < 1e-4 * scale and (out.argmax(1) == native.argmax(1)).all()


@pytest.mark.gpu
def test_bench_line_says_what_is_useful_and_what_moves():
    """bench.py's JSON line (round-4 verdict item 7): `useful_frac` beside `frac` for the dominant kernel, `step_useful_frac`,
    one `per_kernel` entry per family with its algorithmic bytes / PMC traffic / ratio fields (the PMC figures themselves are null
    unless profiles/hbm_traffic_latest.json belongs to the running sources), and the strict top-1 at the top level."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extra-configs",
                          "--clips-per-gpu", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    r = line["roofline"]
    assert 0 < r["useful_frac"] <= r["frac"] + 1e-9 and 0 < r["step_useful_frac"] <= r["step_frac"] + 1e-9
    fams = r["per_kernel"]
    assert "eco::wgemm_kernel" in fams and "eco::wino3_input_kernel" in fams and "eco::wino3_output_kernel" in fams
    for k, v in fams.items():
        assert {"useful_frac_of_own_floor", "algorithmic_gb_per_launch", "traffic_gb_per_launch", "traffic_ratio"} <= set(v), k
    assert line["top1_equal"] is True and line["max_rel_err_vs_cpu_ref"] < 1e-3
    assert line["parity"]["top1_equal_per_clip"] == [True, True] and line["cpu_baseline"]["kind"] == "reference"
