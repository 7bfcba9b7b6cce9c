"""Regression tests for the round-5 advisor findings.

1. (medium) work-counter slots of the dynamic-share bf16 launches (csrc/eco_api.hip, counter_slot_index): hipStreamPerThread is
   one handle for a different stream per host thread -> static shares; the capture ring no longer wraps (two live graphs could
   hold one slot) -> static shares once the 192 slots are gone, eco_counters_release_capture_slots() returns them; a full stream
   table recycles entries of streams without work in flight.
2. (low) lds_ld16 of csrc/eco_wino3.hip is a compiler-tracked LDS load now: no inline-asm ds_read in the product sources.
3. (low) hip.pool_kernel_name mirrors eco_pool_forward_strided's dispatch for channel-slice outputs.
4. (low) include/eco_hip.h states the real plane bound of the F(4x4x4,3x3x3) transforms.
5. (low) eco_wfused_pool_forward validates every argument before its first launch; the engine does not fuse pool2 into a conv
   that was planned without the partial-maxima scratch.
"""
import os
import re

import numpy as np
import pytest

from eco_amd import hip
from tests.conftest import ROOT


def test_stream_table_hands_out_one_slot_per_stream(backend):
    lib = backend.lib
    if backend.kind == "hip":
        pytest.skip("the emulator build has no stream objects: fake handles are only safe there")
    base = 0x7000_0000_0000
    first = lib.counter_slot_probe(base)
    assert 0 <= first < 64 and lib.counter_slot_probe(base) == first            # a stream keeps its slot
    seen = {lib.counter_slot_probe(base + 64 * k) for k in range(1, 200)}
    assert -1 in seen                                                            # the 64-entry table fills up: static shares
    assert all(s == -1 or 0 <= s < 64 for s in seen)
    assert lib.counter_slot_probe(base) == first                                 # ... and an old entry is still its own


@pytest.mark.gpu
def test_per_thread_sentinel_and_capture_slots(hip_backend):
    import torch
    lib = hip_backend.lib
    assert lib.counter_slot_probe(2) == -1                      # hipStreamPerThread: static shares
    null = lib.counter_slot_probe(None)
    assert 0 <= null < 64 and lib.counter_slot_probe(1) == null  # hipStreamLegacy is the null stream
    lib.counters_release_capture_slots()
    got = []
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(8, device="cuda")
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(200):
            got.append(lib.counter_slot_probe(s))
        x += 1
    assert got[:192] == list(range(64, 256)) and got[192:] == [-1] * 8       # handed out once, never wrapped
    lib.counters_release_capture_slots()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        again = lib.counter_slot_probe(torch.cuda.current_stream().cuda_stream)
        x += 1
    assert again == 64
    lib.counters_release_capture_slots()
    # many short-lived streams: the table recycles idle / destroyed entries instead of ending on static shares for good
    slots = []
    for _ in range(100):
        st = torch.cuda.Stream()
        slots.append(lib.counter_slot_probe(st.cuda_stream))
    assert -1 not in slots


def test_no_inline_asm_lds_reads_in_the_3d_transforms():
    src = open(os.path.join(ROOT, "eco-efficient-video-understanding_amd", "csrc", "eco_wino3.hip")).read()
    assert "asm volatile(\"ds_read" not in src and "address_space(3)" in src and "__builtin_assume_aligned" in src


def test_pool_kernel_name_follows_the_strided_dispatch():
    g = hip.pool_geom(4, 32, (28, 28), (3, 3), (2, 2), (0, 0), (14, 14), "MAX")
    assert hip.pool_kernel_name(g) == "eco::maxpool2d_k3s2_kernel<2>"
    S = 14 * 14
    assert hip.pool_kernel_name(g, 96 * S, 64 * S) == "eco::maxpool2d_k3s2_kernel<2>"      # aligned slice: still the fast path
    g7 = hip.pool_geom(4, 32, (14, 14), (3, 3), (2, 2), (0, 0), (7, 7), "MAX")
    assert hip.pool_kernel_name(g7, 96 * 49, 33 * 49).startswith("eco::pool2d_k3_kernel")   # a slice that starts off 16 bytes
    ga = hip.pool_geom(4, 32, (28, 28), (3, 3), (1, 1), (1, 1), (28, 28), "AVE")
    assert hip.pool_kernel_name(ga) == "eco::avgpool2d_k3s1p1_kernel<4>"
    assert hip.pool_kernel_name(ga, 64 * 784, 0) == "eco::pool2d_k3_kernel"                 # the AVE fast path writes dense blobs only
    gg = hip.pool_geom(4, 32, (7, 7), (7, 7), (1, 1), (0, 0), (1, 1), "AVE")
    assert hip.pool_kernel_name(gg) == "eco::global_avg_kernel" and hip.pool_kernel_name(gg, 64, 0) == "eco::pool_kernel"


def test_header_states_the_real_wino3_plane_bound(backend):
    text = open(os.path.join(ROOT, "include", "eco_hip.h")).read()
    assert "52x52" in text and "~56x56" not in text.split("eco_wino3_weight_transform")[0].split("F(4x4x4,3x3x3) for the 3-D trunk")[1]
    assert backend.lib.wino3_lds_bytes(1, 13, 13) <= 152 * 1024 < backend.lib.wino3_lds_bytes(1, 14, 14)


def test_wfused_pool_checks_alignment_before_launching(backend):
    lib = backend.lib
    plan = lib.wgemm_plan(1, 64, 32, 1, 2, 2, 1, None)
    ep = hip.ConvEpilogue()
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    scratch = backend.dev(np.full(lib.wfused_pool_scratch_elems(plan), 7.0, np.float32))
    y = backend.empty((1, 32, 4, 4 + 1))
    v = backend.dev(np.zeros(plan.v_elems, np.float32))
    up = backend.dev(np.zeros(lib.wfused_weight_elems(plan), np.float32))
    with pytest.raises(hip.EcoError, match="8-byte aligned"):
        lib.wfused_pool_forward(plan, backend.ptr(v), backend.ptr(up), 8, 8, ep, backend.ptr(scratch), backend.ptr(y) + 4)
    assert (backend.host(scratch, (lib.wfused_pool_scratch_elems(plan),)) == 7.0).all()     # nothing ran
