// TEST INFRASTRUCTURE (oracle/ref_shim): minimal stand-in for caffe/common.hpp so that a few reference source
// files (util/im2col.cpp and layers/{pooling,bn,permute,eltwise,concat,inner_product,reshape,relu}_layer.cpp)
// compile *unmodified, from where they lie under /root/reference* without glog / gflags / boost / protobuf /
// CUDA.  Only what those files use is provided.
// Never part of the product; see oracle/Makefile.
#pragma once
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace caffe {
using std::shared_ptr;
using std::string;
using std::vector;
}  // namespace caffe

// glog-like CHECK macros: evaluate, and on failure print the streamed message and abort (LOG(FATAL)).
namespace ref_shim {
struct Fatal {
  std::ostringstream os;
  Fatal(const char* file, int line, const char* what) { os << file << ":" << line << " Check failed: " << what << " "; }
  [[noreturn]] ~Fatal() { std::cerr << os.str() << std::endl; std::abort(); }
  template <typename T> Fatal& operator<<(const T& v) { os << v; return *this; }
};
struct Voidify { void operator&(const Fatal&) {} };
// LOG(INFO) / LOG(WARNING) / LOG(ERROR): swallowed; LOG(FATAL) aborts like a failed CHECK.
struct Quiet { template <typename T> Quiet& operator<<(const T&) { return *this; } };
struct Log_INFO : Quiet { Log_INFO(const char*, int) {} };
struct Log_WARNING : Quiet { Log_WARNING(const char*, int) {} };
struct Log_ERROR : Quiet { Log_ERROR(const char*, int) {} };
struct Log_FATAL : Fatal { Log_FATAL(const char* f, int l) : Fatal(f, l, "LOG(FATAL)") {} };
}  // namespace ref_shim
#define CHECK(c) (c) ? (void)0 : ref_shim::Voidify() & ref_shim::Fatal(__FILE__, __LINE__, #c)
#define CHECK_OP(a, b, op) CHECK((a) op (b))
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define DCHECK(c) CHECK(c)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define LOG(sev) ref_shim::Log_##sev(__FILE__, __LINE__)
#define NOT_IMPLEMENTED LOG(FATAL) << "Not Implemented Yet"
#define NO_GPU LOG(FATAL) << "Cannot use GPU in CPU-only Caffe: check mode."

// CPU_ONLY build of the reference: GPU entry points are stubs (util/device_alternate.hpp:9-31).
#define STUB_GPU(classname)                                                                       \
  template <typename Dtype>                                                                       \
  void classname<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { NO_GPU; } \
  template <typename Dtype>                                                                       \
  void classname<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, \
                                      const vector<Blob<Dtype>*>& bottom) { NO_GPU; }
#define INSTANTIATE_CLASS(classname) \
  template class classname<float>;   \
  template class classname<double>
