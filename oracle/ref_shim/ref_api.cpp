// oracle/ref_shim/ref_api.cpp -- TEST INFRASTRUCTURE.  C entry points over the reference sources that are
// compiled, unmodified, from /root/reference into oracle/_ref/libeco_ref.so (see oracle/Makefile):
//   caffe_3d/src/caffe/util/im2col.cpp                  im2col_cpu / im2col_nd_cpu
//   caffe_3d/src/caffe/layers/pooling_layer.cpp         PoolingLayer::LayerSetUp / Reshape / Forward_cpu
//   caffe_3d/src/caffe/layers/bn_layer.cpp              BNLayer (4-D blobs: the only ones its CPU code takes)
//   caffe_3d/src/caffe/layers/permute_layer.cpp         PermuteLayer + Permute()
//   caffe_3d/src/caffe/layers/eltwise_layer.cpp         EltwiseLayer
//   caffe_3d/src/caffe/layers/concat_layer.cpp          ConcatLayer
//   caffe_3d/src/caffe/layers/inner_product_layer.cpp   InnerProductLayer
//   caffe_3d/src/caffe/layers/reshape_layer.cpp         ReshapeLayer (shape rules)
//   caffe_3d/src/caffe/layers/relu_layer.cpp            ReLULayer
//   caffe_3d/src/caffe/layers/base_conv_layer.cpp       BaseConvolutionLayer (LayerSetUp / Reshape / forward_cpu_gemm / _bias)
//   caffe_3d/src/caffe/layers/conv_layer.cpp            ConvolutionLayer (compute_output_shape / Forward_cpu)
// i.e. every layer type of the ECO deploy graphs that computes (Split / Dropout(TEST) are identities), as the
// reference's own object code: ref_convolution_layer_forward below runs the compiled class.
// plus ref_conv_forward, a second convolution forward that performs exactly the call sequence of ConvolutionLayer::Forward_cpu
// (layers/conv_layer.cpp:28-43 -> BaseConvolutionLayer::forward_cpu_gemm / forward_cpu_bias,
// layers/base_conv_layer.cpp:264-287 -> caffe_cpu_gemm, util/math_functions.cpp:12-21): per image, the
// REFERENCE im2col into a col buffer, one cblas_sgemm W[cout x K] * col[K x S], and the bias as a rank-1 sgemm
// against an all-ones multiplier -- kept because it can spread the images of a batch over host threads (bench.py's
// image-parallel CPU baseline); tests/test_oracle_ref.py pins it bit-identically to the compiled class.  cblas_sgemm is the one third-party routine (the reference links
// ATLAS/OpenBLAS/MKL, unpinned); here it is SciPy's bundled OpenBLAS, resolved at run time by the Python side
// and handed in as a function pointer, so this library links nothing.
// Used to pin oracle/eco_oracle.py (tests/test_oracle_ref.py) and as bench.py's CPU baseline.
#include <thread>

#include "caffe/util/im2col.hpp"
#include "caffe/util/math_functions.hpp"
#include "caffe/vision_layers.hpp"

using namespace caffe;

namespace ref_shim { sgemm_fn g_sgemm = nullptr; }

namespace {
typedef std::vector<Blob<float>*> BlobVec;
void fill(Blob<float>& b, const std::vector<int>& shape, const float* src) {
  b.Reshape(shape);
  if (src) caffe_copy(b.count(), src, b.mutable_cpu_data());
}
std::vector<int> shape_of(const int* shape, int naxes) { return std::vector<int>(shape, shape + naxes); }
}  // namespace

extern "C" {

// The cblas_sgemm every caffe_cpu_gemm<float> of the compiled layer files goes through (SciPy's OpenBLAS).
void ref_set_sgemm(void* fn) { ref_shim::g_sgemm = (ref_shim::sgemm_fn)fn; }

// BNLayer::LayerSetUp / Reshape / Forward_cpu, TEST phase (bn_layer.cpp:10-207).  x: [n,c,h,w] (<= 4 axes: LegacyShape
// CHECK-fails beyond, bn_layer.cpp:70-73 via blob.hpp); blobs = slope, bias, running mean, running variance [1,c].
int ref_bn_forward(const float* x, const int* shape, int naxes, const float* slope, const float* bias, const float* mean,
                   const float* var, float eps, int frozen, float* y) {
  if (naxes < 2 || naxes > 4) return -1;
  LayerParameter lp;
  lp.bn_param_.eps_ = eps;
  lp.bn_param_.frozen_ = frozen != 0;
  BNLayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, shape_of(shape, naxes), x);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);                       // creates the four [1,c] parameter blobs through the fillers
  const float* src[4] = {slope, bias, mean, var};
  for (int i = 0; i < 4; ++i) caffe_copy(layer.blobs()[i]->count(), src[i], layer.blobs()[i]->mutable_cpu_data());
  layer.Reshape(bv, tv);
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// PermuteLayer (permute_layer.cpp:9-114).  out_shape receives the permuted shape; y may be NULL to query it.
int ref_permute_forward(const float* x, const int* shape, int naxes, const int* order, int norder, float* y, int* out_shape) {
  LayerParameter lp;
  for (int i = 0; i < norder; ++i) lp.permute_param_.order_.push_back((unsigned)order[i]);
  PermuteLayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, shape_of(shape, naxes), y ? x : nullptr);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);
  layer.Reshape(bv, tv);
  for (int i = 0; i < naxes; ++i) out_shape[i] = top.shape(i);
  if (!y) return 0;
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// EltwiseLayer (eltwise_layer.cpp:10-119).  op: 0 PROD, 1 SUM, 2 MAX; coeffs may be NULL (all ones).
int ref_eltwise_forward(const float* const* xs, int nbottom, const int* shape, int naxes, int op, const float* coeffs, float* y) {
  LayerParameter lp;
  lp.eltwise_param_.operation_ = (EltwiseParameter_EltwiseOp)op;
  if (coeffs) lp.eltwise_param_.coeff_.assign(coeffs, coeffs + nbottom);
  EltwiseLayer<float> layer(lp);
  std::vector<Blob<float> > bottoms(nbottom);
  Blob<float> top;
  BlobVec bv, tv(1, &top);
  for (int i = 0; i < nbottom; ++i) { fill(bottoms[i], shape_of(shape, naxes), xs[i]); bv.push_back(&bottoms[i]); }
  layer.LayerSetUp(bv, tv);
  layer.Reshape(bv, tv);
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// ConcatLayer (concat_layer.cpp:9-70).  shapes: nbottom rows of naxes ints; out_shape receives the top shape; y may be NULL.
int ref_concat_forward(const float* const* xs, int nbottom, const int* shapes, int naxes, int axis, float* y, int* out_shape) {
  LayerParameter lp;
  lp.concat_param_.axis_ = axis;
  lp.concat_param_.has_axis_ = true;
  ConcatLayer<float> layer(lp);
  std::vector<Blob<float> > bottoms(nbottom);
  Blob<float> top;
  BlobVec bv, tv(1, &top);
  for (int i = 0; i < nbottom; ++i) { fill(bottoms[i], shape_of(shapes + i * naxes, naxes), y ? xs[i] : nullptr); bv.push_back(&bottoms[i]); }
  layer.LayerSetUp(bv, tv);
  layer.Reshape(bv, tv);
  for (int i = 0; i < naxes; ++i) out_shape[i] = top.shape(i);
  if (!y) return 0;
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// InnerProductLayer (inner_product_layer.cpp:13-93): w [num_output, K], b [num_output] or NULL; y [M, num_output].
int ref_inner_product_forward(const float* x, const int* shape, int naxes, const float* w, const float* b, int num_output,
                              int axis, float* y) {
  LayerParameter lp;
  lp.inner_product_param_.num_output_ = (unsigned)num_output;
  lp.inner_product_param_.bias_term_ = b != nullptr;
  lp.inner_product_param_.axis_ = axis;
  InnerProductLayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, shape_of(shape, naxes), x);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);
  caffe_copy(layer.blobs()[0]->count(), w, layer.blobs()[0]->mutable_cpu_data());
  if (b) caffe_copy(layer.blobs()[1]->count(), b, layer.blobs()[1]->mutable_cpu_data());
  layer.Reshape(bv, tv);
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// ReshapeLayer::LayerSetUp / Reshape (reshape_layer.cpp:9-90): the 0 / -1 rules.  Returns the number of top axes.
int ref_reshape_shape(const int* shape, int naxes, const long long* dims, int ndims, int axis, int num_axes, int* out_shape) {
  LayerParameter lp;
  lp.reshape_param_.shape_.dim_.assign(dims, dims + ndims);
  lp.reshape_param_.axis_ = axis;
  lp.reshape_param_.num_axes_ = num_axes;
  ReshapeLayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, shape_of(shape, naxes), nullptr);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);
  layer.Reshape(bv, tv);
  for (int i = 0; i < top.num_axes(); ++i) out_shape[i] = top.shape(i);
  return top.num_axes();
}

// ReLULayer::Forward_cpu (relu_layer.cpp:10-20)
int ref_relu_forward(const float* x, long count, float negative_slope, float* y) {
  LayerParameter lp;
  lp.relu_param_.negative_slope_ = negative_slope;
  ReLULayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, std::vector<int>(1, (int)count), x);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.Reshape(bv, tv);
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

// ConvolutionLayer<float>: LayerSetUp / Reshape (base_conv_layer.cpp:13-262) / Forward_cpu (conv_layer.cpp:28-43), the
// compiled class itself.  x: [n, cin, spatial...] (naxes = 2 + nsp); kernel / stride / pad: nk / ns / np repeated-field
// entries exactly as a prototxt gives them (one value, or one per spatial axis; ns / np may be 0 = schema defaults);
// w: [num_output, cin, kernel...]; b: [num_output] or NULL (bias_term false).  out_shape receives the top shape; y may
// be NULL to query it.
int ref_convolution_layer_forward(const float* x, const int* shape, int naxes, const float* w, const float* b, int num_output,
                                  const int* kernel, int nk, const int* stride, int ns, const int* pad, int np,
                                  int force_nd_im2col, float* y, int* out_shape) {
  LayerParameter lp;
  ConvolutionParameter& cp = lp.convolution_param_;
  cp.num_output_ = (unsigned)num_output;
  cp.bias_term_ = b != nullptr;
  cp.force_nd_im2col_ = force_nd_im2col != 0;
  for (int i = 0; i < nk; ++i) cp.kernel_size_.push_back((unsigned)kernel[i]);
  for (int i = 0; i < ns; ++i) cp.stride_.push_back((unsigned)stride[i]);
  for (int i = 0; i < np; ++i) cp.pad_.push_back((unsigned)pad[i]);
  ConvolutionLayer<float> layer(lp);
  Blob<float> bottom, top;
  fill(bottom, shape_of(shape, naxes), y ? x : nullptr);
  BlobVec bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);                        // creates blobs_[0] (and [1]) through the constant filler
  layer.Reshape(bv, tv);
  for (int i = 0; i < naxes; ++i) out_shape[i] = top.shape(i);
  if (!y) return 0;
  caffe_copy(layer.blobs()[0]->count(), w, layer.blobs()[0]->mutable_cpu_data());
  if (b) caffe_copy(layer.blobs()[1]->count(), b, layer.blobs()[1]->mutable_cpu_data());
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

typedef void (*sgemm_fn)(int order, int transa, int transb, int m, int n, int k, float alpha, const float* a, int lda,
                         const float* b, int ldb, float beta, float* c, int ldc);
enum { kRowMajor = 101, kNoTrans = 111 };

// conv_im2col_cpu (include/caffe/vision_layers.hpp:102-114): 2-D layers use im2col_cpu, N-D layers im2col_nd_cpu.
// in_shape = {C, spatial...}; col_shape = {C*prod(kernel), out...}.
void ref_im2col(const float* x, int nsp, const int* in_shape, const int* col_shape, const int* kernel, const int* pad,
                const int* stride, float* col) {
  if (nsp == 2)
    im2col_cpu(x, in_shape[0], in_shape[1], in_shape[2], kernel[0], kernel[1], pad[0], pad[1], stride[0], stride[1], col);
  else
    im2col_nd_cpu(x, nsp, in_shape, col_shape, kernel, pad, stride, col);
}

// ConvolutionLayer::Forward_cpu for one bottom.  image_threads == 1 is the reference's structure (images in
// sequence, BLAS free to use its threads); image_threads > 1 spreads the images over that many host threads
// (each with its own col buffer; the caller sets BLAS to one thread) -- same arithmetic per image.
int ref_conv_forward(const float* x, const float* w, const float* bias, float* y, int n, int cin, int cout, int nsp,
                     const int* in_sp, const int* kernel, const int* stride, const int* pad, sgemm_fn sgemm,
                     int image_threads) {
  if (nsp < 1 || nsp > 3 || !sgemm) return -1;
  int in_shape[4], col_shape[4], out_sp[3];
  in_shape[0] = cin;
  long s_in = 1, s_out = 1, kdim = cin;
  bool is_1x1 = true;
  for (int i = 0; i < nsp; ++i) {
    in_shape[1 + i] = in_sp[i];
    out_sp[i] = (in_sp[i] + 2 * pad[i] - kernel[i]) / stride[i] + 1;   // conv_layer.cpp:19-22
    col_shape[1 + i] = out_sp[i];
    s_in *= in_sp[i]; s_out *= out_sp[i]; kdim *= kernel[i];
    is_1x1 = is_1x1 && kernel[i] == 1 && stride[i] == 1 && pad[i] == 0;  // base_conv_layer.cpp:110-117
  }
  col_shape[0] = (int)kdim;
  const long bottom_dim = (long)cin * s_in, top_dim = (long)cout * s_out;
  std::vector<float> ones((size_t)s_out, 1.0f);                          // bias_multiplier_ (:255-260)
  auto run = [&](int n0, int n1) {
    std::vector<float> col(is_1x1 ? 0 : (size_t)(kdim * s_out));
    for (int i = n0; i < n1; ++i) {
      const float* xi = x + i * bottom_dim;
      float* yi = y + i * top_dim;
      const float* col_buff = xi;
      if (!is_1x1) {
        ref_im2col(xi, nsp, in_shape, col_shape, kernel, pad, stride, col.data());
        col_buff = col.data();
      }
      sgemm(kRowMajor, kNoTrans, kNoTrans, cout, (int)s_out, (int)kdim, 1.0f, w, (int)kdim, col_buff, (int)s_out, 0.0f,
            yi, (int)s_out);
      if (bias)
        sgemm(kRowMajor, kNoTrans, kNoTrans, cout, (int)s_out, 1, 1.0f, bias, 1, ones.data(), (int)s_out, 1.0f, yi,
              (int)s_out);
    }
  };
  if (image_threads <= 1 || n == 1) {
    run(0, n);
  } else {
    const int t = image_threads < n ? image_threads : n;
    std::vector<std::thread> th;
    for (int k = 0; k < t; ++k) th.emplace_back(run, (int)((long)n * k / t), (int)((long)n * (k + 1) / t));
    for (auto& q : th) q.join();
  }
  return 0;
}

// PoolingLayer (2-D blobs: the only case the reference's CPU Forward implements, pooling_layer.cpp:177-201).
// method: 0 MAX, 1 AVE.  shape = {n, c, h, w}; out_shape receives the pooled shape; y may be NULL to query it.
// kernel == NULL means global_pooling.
int ref_pool_forward(const float* x, const int* shape, int naxes, int method, const int* kernel, const int* stride,
                     const int* pad, float* y, int* out_shape) {
  LayerParameter lp;
  PoolingParameter* pp = lp.mutable_pooling_param();
  pp->pool_ = method == 0 ? PoolingParameter_PoolMethod_MAX : PoolingParameter_PoolMethod_AVE;
  const int nsp = naxes - 2;
  if (!kernel) {
    pp->global_pooling_ = true;
  } else {
    for (int i = 0; i < nsp; ++i) pp->kernel_size_.push_back(kernel[i]);
  }
  for (int i = 0; i < nsp; ++i) {
    if (stride) pp->stride_.push_back(stride[i]);
    if (pad) pp->pad_.push_back(pad[i]);
  }
  PoolingLayer<float> layer(lp);
  Blob<float> bottom, top;
  bottom.Reshape(std::vector<int>(shape, shape + naxes));
  std::vector<Blob<float>*> bv(1, &bottom), tv(1, &top);
  layer.LayerSetUp(bv, tv);
  layer.Reshape(bv, tv);
  for (int i = 0; i < naxes; ++i) out_shape[i] = top.shape(i);
  if (!y) return 0;
  if (naxes != 4) return -1;
  caffe_copy(bottom.count(), x, bottom.mutable_cpu_data());
  layer.Forward_cpu(bv, tv);
  caffe_copy(top.count(), top.cpu_data(), y);
  return 0;
}

}  // extern "C"
