// oracle/ref_shim: the handful of caffe math helpers the compiled reference files call (util/math_functions.cpp:
// caffe_set :59-70, caffe_copy :86-100), plus cblas_sgemm bound to SciPy's bundled OpenBLAS for ref_api.cpp.
#pragma once
#include <cstring>

#include "caffe/common.hpp"   // as the reference header does (vector, CHECK macros)
namespace caffe {
template <typename Dtype>
inline void caffe_set(const int N, const Dtype alpha, Dtype* Y) {
  if (alpha == 0) { memset(Y, 0, sizeof(Dtype) * N); return; }
  for (int i = 0; i < N; ++i) Y[i] = alpha;
}
template <typename Dtype>
inline void caffe_copy(const int N, const Dtype* X, Dtype* Y) { if (X != Y) memcpy(Y, X, sizeof(Dtype) * N); }
}  // namespace caffe
