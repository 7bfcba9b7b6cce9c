#!/usr/bin/env python
"""Extract the known-answer vectors of the reference's own unit tests into
tests/golden/reference_vectors.json.

Run in the authoring container only (needs /root/reference); the JSON is committed so that the
CPU and GPU suites can pin the oracle and the HIP kernels without the reference tree.  Nothing is
computed here: every number is copied out of the reference test sources cited per case
(caffe_3d/src/caffe/test/*.cpp).
"""
import json
import os
import re
import sys

REF = "/root/reference/caffe_3d/src/caffe/test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def read(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def lineno(text, pos):
    return text.count("\n", 0, pos) + 1


def block(text, start, end):
    i = text.index(start)
    j = text.index(end, i)
    return text[i:j], lineno(text, i), lineno(text, j)


def brace_array(chunk, decl):
    m = re.search(re.escape(decl) + r"\s*=\s*\{([^}]*)\}", chunk, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [float(x) for x in re.findall(r"-?\d+\.?\d*", body)]


def indexed_assignments(chunk):
    """`...mutable_cpu_data()[i + k] = v;` / `[k] = v;` -> list ordered by k"""
    vals = {}
    for k, v in re.findall(r"mutable_cpu_data\(\)\[(?:i \+\s*)?(\d+)\]\s*=\s*(-?[\d.]+);", chunk):
        vals[int(k)] = float(v)
    return [vals[k] for k in sorted(vals)]


def indexed_expectations(chunk):
    vals = {}
    # only the data blob's expectations (blob_top_), not the argmax mask (blob_top_mask_)
    for k, v in re.findall(r"blob_top_->cpu_data\(\)\[(?:i \+\s*)?(\d+)\],\s*([^,\)]+?)\s*[,\)]", chunk):
        vals[int(k)] = float(eval(v))  # "8.0 / 9" style literals
    return [vals[k] for k in sorted(vals)]


def main():
    cases = []
    pool = read("test_pooling_layer.cpp")
    blas = read("test_util_blas.cpp")

    # ---- GEMM {1..6} x {1..12}
    c, a, b = block(blas, "TYPED_TEST(GemmTest, TestGemmCPUGPU)", "// Test when we have a transposed A")
    cases.append(dict(name="gemm_2x3_3x4", source=f"test_util_blas.cpp:{a}-{b}", op="gemm",
                      a=brace_array(c, "TypeParam data[12]")[:6], a_shape=[2, 3],
                      b=brace_array(c, "TypeParam data[12]"), b_shape=[3, 4],
                      expected=brace_array(c, "TypeParam result[8]"), tol=0.0))

    # ---- 2-D MAX 2x2 on [1 2 5 2 3; 9 4 1 4 8; 1 2 5 2 3]
    c, a, b = block(pool, "void TestForwardSquare()", "// Test for 3x 2 rectangular pooling layer with kernel_h > kernel_w")
    cases.append(dict(name="maxpool2d_square_k2", source=f"test_pooling_layer.cpp:{a}-{b}", op="pool", method="MAX",
                      x=indexed_assignments(c), x_shape=[1, 1, 3, 5], kernel=[2, 2], stride=[1, 1], pad=[0, 0],
                      expected=indexed_expectations(c), y_shape=[1, 1, 2, 4], tol=0.0))

    # ---- 2-D MAX 3x3 stride 2 pad 2
    c, a, b = block(pool, "TYPED_TEST(PoolingLayerTest, TestForwardMaxPadded)", "TYPED_TEST(PoolingLayerTest, TestGradientMaxTopMask)")
    cases.append(dict(name="maxpool2d_padded_k3s2p2", source=f"test_pooling_layer.cpp:{a}-{b}", op="pool", method="MAX",
                      x=indexed_assignments(c), x_shape=[1, 1, 3, 3], kernel=[3, 3], stride=[2, 2], pad=[2, 2],
                      expected=indexed_expectations(c), y_shape=[1, 1, 3, 3], tol=1e-8))

    # ---- 2-D AVE 3x3 stride 1 pad 1 on constant 2
    c, a, b = block(pool, "TYPED_TEST(PoolingLayerTest, TestForwardAve)", "TYPED_TEST(PoolingLayerTest, TestGradientAve)")
    cases.append(dict(name="avepool2d_k3s1p1_const2", source=f"test_pooling_layer.cpp:{a}-{b}", op="pool", method="AVE",
                      x=[2.0] * 9, x_shape=[1, 1, 3, 3], kernel=[3, 3], stride=[1, 1], pad=[1, 1],
                      expected=indexed_expectations(c), y_shape=[1, 1, 3, 3], tol=1e-5))

    # ---- pooled-shape rules (TestSetup / TestSetupPadded on a [2,3,6,5] bottom)
    c, a, b = block(pool, "TYPED_TEST(PoolingLayerTest, TestSetup)", "TYPED_TEST(PoolingLayerTest, TestSetupGlobalPooling)")
    cases.append(dict(name="pooled_shape_rules", source=f"test_pooling_layer.cpp:{a}-{b}", op="pooled_shape",
                      in_hw=[6, 5], rules=[dict(kernel=3, stride=2, pad=0, out_hw=[3, 2]),
                                           dict(kernel=3, stride=2, pad=1, out_hw=[4, 3])]))
    assert "EXPECT_EQ(this->blob_top_->height(), 3);" in c and "EXPECT_EQ(this->blob_top_->width(), 2);" in c
    assert "EXPECT_EQ(this->blob_top_->height(), 4);" in c and "EXPECT_EQ(this->blob_top_->width(), 3);" in c

    # ---- 3-D MAX 2x2x2 on the 4x3x6 "randperm" volume (cuDNN test)
    cin, a0, _ = block(pool, "void SetUp3DTestBottomBlob(const int num, const int channels)", "// test for 2x2x2 pooling")
    c, a, b = block(pool, "void TestForwardCube()", "// test for 2x2x3 pooling")
    cases.append(dict(name="maxpool3d_cube_k2", source=f"test_pooling_layer.cpp:{a0}-{b}", op="pool", method="MAX",
                      x=brace_array(cin, "const int input[]"), x_shape=[1, 1, 4, 3, 6], kernel=[2, 2, 2], stride=[1, 1, 1],
                      pad=[0, 0, 0], expected=brace_array(c, "const int output[]"), y_shape=[1, 1, 3, 2, 5], tol=0.0))

    # ---- 3-D AVE 3x3x3 stride 1 pad 1 on the 27-cube (cuDNN test; divisor includes padding)
    c, a, b = block(pool, "TYPED_TEST(CuDNNPoolingLayerTest3D, TestForwardAve3DCuDNN)", "TYPED_TEST(CuDNNPoolingLayerTest3D, TestGradientMax3DCuDNN)")
    cases.append(dict(name="avepool3d_k3s1p1_cube27", source=f"test_pooling_layer.cpp:{a}-{b}", op="pool", method="AVE",
                      x=brace_array(c, "const int input[]"), x_shape=[1, 1, 3, 3, 3], kernel=[3, 3, 3], stride=[1, 1, 1],
                      pad=[1, 1, 1], expected=brace_array(c, "const Dtype output[]"), y_shape=[1, 1, 3, 3, 3], tol=1e-4))

    for cse in cases:
        if cse["op"] == "pool":
            n_in, n_out = 1, 1
            for d in cse["x_shape"]:
                n_in *= d
            for d in cse["y_shape"]:
                n_out *= d
            assert len(cse["x"]) == n_in and len(cse["expected"]) == n_out, cse["name"]
    with open(OUT, "w") as f:
        json.dump(dict(generated_by="tests/golden/make_golden.py", reference="/root/reference/caffe_3d/src/caffe/test",
                       cases=cases), f, indent=1)
    print(f"wrote {len(cases)} cases to {OUT}")


if __name__ == "__main__":
    sys.exit(main())
