// oracle/ref_shim: the caffe math helpers the compiled reference files call, restated from
// caffe_3d/src/caffe/util/math_functions.cpp and include/caffe/util/mkl_alternate.hpp (lines cited per function).
// The three BLAS entry points the TEST-phase forwards reach -- cblas_sgemm (caffe_cpu_gemm :12-21), and through
// it nothing else -- are SciPy's bundled OpenBLAS, resolved by the Python side and handed in once
// (ref_set_sgemm); gemv / axpy / scal only run in TRAIN branches and backward passes and are plain loops here.
#pragma once
#include <math.h>   // as include/caffe/util/mkl_alternate.hpp:12 does: pow(float, float) resolves as it does there
#include <cstring>

#include "caffe/common.hpp"   // as the reference header does (vector, CHECK macros)

enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };

namespace ref_shim {
typedef void (*sgemm_fn)(int order, int transa, int transb, int m, int n, int k, float alpha, const float* a, int lda,
                         const float* b, int ldb, float beta, float* c, int ldc);
extern sgemm_fn g_sgemm;   // set by ref_set_sgemm (ref_api.cpp)
// reference GEMM for the double instantiations and for runs without a BLAS pointer: C = alpha op(A) op(B) + beta C
template <typename T>
inline void naive_gemm(int ta, int tb, int M, int N, int K, T alpha, const T* A, const T* B, T beta, T* C) {
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      T acc = 0;
      for (int k = 0; k < K; ++k)
        acc += (ta == CblasNoTrans ? A[(long)i * K + k] : A[(long)k * M + i]) * (tb == CblasNoTrans ? B[(long)k * N + j] : B[(long)j * K + k]);
      C[(long)i * N + j] = alpha * acc + (beta == T(0) ? T(0) : beta * C[(long)i * N + j]);
    }
}
}  // namespace ref_shim

namespace caffe {
// math_functions.cpp:56-70
template <typename Dtype>
inline void caffe_set(const int N, const Dtype alpha, Dtype* Y) {
  if (alpha == 0) { memset(Y, 0, sizeof(Dtype) * N); return; }
  for (int i = 0; i < N; ++i) Y[i] = alpha;
}
// math_functions.cpp:86-100 (CPU branch)
template <typename Dtype>
inline void caffe_copy(const int N, const Dtype* X, Dtype* Y) { if (X != Y) memcpy(Y, X, sizeof(Dtype) * N); }

// math_functions.cpp:12-32: lda / ldb as the reference computes them, then cblas_?gemm(CblasRowMajor, ...)
template <typename Dtype>
inline void caffe_cpu_gemm(const CBLAS_TRANSPOSE TransA, const CBLAS_TRANSPOSE TransB, const int M, const int N, const int K,
                           const Dtype alpha, const Dtype* A, const Dtype* B, const Dtype beta, Dtype* C) {
  ref_shim::naive_gemm<Dtype>(TransA, TransB, M, N, K, alpha, A, B, beta, C);
}
template <>
inline void caffe_cpu_gemm<float>(const CBLAS_TRANSPOSE TransA, const CBLAS_TRANSPOSE TransB, const int M, const int N,
                                  const int K, const float alpha, const float* A, const float* B, const float beta, float* C) {
  const int lda = (TransA == CblasNoTrans) ? K : M;
  const int ldb = (TransB == CblasNoTrans) ? N : K;
  if (ref_shim::g_sgemm) ref_shim::g_sgemm(CblasRowMajor, TransA, TransB, M, N, K, alpha, A, lda, B, ldb, beta, C, N);
  else ref_shim::naive_gemm<float>(TransA, TransB, M, N, K, alpha, A, B, beta, C);
}
// math_functions.cpp:34-46 (cblas_?gemv, row major): TRAIN branches / backward only
template <typename Dtype>
inline void caffe_cpu_gemv(const CBLAS_TRANSPOSE TransA, const int M, const int N, const Dtype alpha, const Dtype* A,
                           const Dtype* x, const Dtype beta, Dtype* y) {
  const int rows = TransA == CblasNoTrans ? M : N, cols = TransA == CblasNoTrans ? N : M;
  for (int i = 0; i < rows; ++i) {
    Dtype acc = 0;
    for (int j = 0; j < cols; ++j) acc += (TransA == CblasNoTrans ? A[(long)i * N + j] : A[(long)j * N + i]) * x[j];
    y[i] = alpha * acc + (beta == Dtype(0) ? Dtype(0) : beta * y[i]);
  }
}
// math_functions.cpp:48-54 (cblas_?axpy): Y += alpha X.  EltwiseLayer SUM reaches it with alpha = coeff (1 in every
// ECO graph, where alpha * x is exact whatever the BLAS does)
template <typename Dtype>
inline void caffe_axpy(const int N, const Dtype alpha, const Dtype* X, Dtype* Y) { for (int i = 0; i < N; ++i) Y[i] += alpha * X[i]; }
// math_functions.cpp:102-120 / mkl_alternate.hpp:83-93 (scal then axpy)
template <typename Dtype>
inline void caffe_cpu_axpby(const int N, const Dtype alpha, const Dtype* X, const Dtype beta, Dtype* Y) {
  for (int i = 0; i < N; ++i) Y[i] = beta * Y[i];
  for (int i = 0; i < N; ++i) Y[i] += alpha * X[i];
}
template <typename Dtype>
inline void caffe_cpu_scale(const int n, const Dtype alpha, const Dtype* x, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = alpha * x[i]; }
// math_functions.cpp:72-84
template <typename Dtype>
inline void caffe_add_scalar(const int N, const Dtype alpha, Dtype* Y) { for (int i = 0; i < N; ++i) Y[i] += alpha; }
// mkl_alternate.hpp:75-78 (DEFINE_VSL_BINARY_FUNC) through math_functions.cpp:132-178
template <typename Dtype>
inline void caffe_add(const int n, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = a[i] + b[i]; }
template <typename Dtype>
inline void caffe_sub(const int n, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = a[i] - b[i]; }
template <typename Dtype>
inline void caffe_mul(const int n, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = a[i] * b[i]; }
template <typename Dtype>
inline void caffe_div(const int n, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = a[i] / b[i]; }
// mkl_alternate.hpp:56 (y[i] = pow(a[i], b)) through math_functions.cpp:180-190
template <typename Dtype>
inline void caffe_powx(const int n, const Dtype* a, const Dtype b, Dtype* y) { for (int i = 0; i < n; ++i) y[i] = pow(a[i], b); }
// math_functions.cpp:236-250 (boost uniform_real): only the TRAIN-phase STOCHASTIC_SUM branch of EltwiseLayer calls it
template <typename Dtype>
inline void caffe_rng_uniform(const int n, const Dtype a, const Dtype b, Dtype* r) {
  for (int i = 0; i < n; ++i) r[i] = a + (b - a) * Dtype(rand()) / Dtype(RAND_MAX);
}
}  // namespace caffe
