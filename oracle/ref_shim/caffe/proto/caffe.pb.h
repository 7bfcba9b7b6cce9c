// oracle/ref_shim: hand-written stand-in for the protoc-generated caffe.pb.h -- only PoolingParameter /
// LayerParameter with the accessors layers/pooling_layer.cpp calls (schema: caffe.proto PoolingParameter).
#pragma once
#include <vector>
namespace caffe {
enum Phase { TRAIN = 0, TEST = 1 };
enum PoolingParameter_PoolMethod {
  PoolingParameter_PoolMethod_MAX = 0, PoolingParameter_PoolMethod_AVE = 1, PoolingParameter_PoolMethod_STOCHASTIC = 2
};
class PoolingParameter {
 public:
  PoolingParameter_PoolMethod pool_ = PoolingParameter_PoolMethod_MAX;
  std::vector<unsigned> kernel_size_, stride_, pad_;
  bool has_kh_ = false, has_kw_ = false, has_sh_ = false, has_sw_ = false, has_ph_ = false, has_pw_ = false;
  unsigned kh_ = 0, kw_ = 0, sh_ = 1, sw_ = 1, ph_ = 0, pw_ = 0;
  bool global_pooling_ = false;
  PoolingParameter_PoolMethod pool() const { return pool_; }
  bool global_pooling() const { return global_pooling_; }
  int kernel_size_size() const { return (int)kernel_size_.size(); }
  unsigned kernel_size(int i) const { return kernel_size_[i]; }
  int stride_size() const { return (int)stride_.size(); }
  unsigned stride(int i) const { return stride_[i]; }
  int pad_size() const { return (int)pad_.size(); }
  unsigned pad(int i) const { return pad_[i]; }
  bool has_kernel_h() const { return has_kh_; }
  bool has_kernel_w() const { return has_kw_; }
  bool has_stride_h() const { return has_sh_; }
  bool has_stride_w() const { return has_sw_; }
  bool has_pad_h() const { return has_ph_; }
  bool has_pad_w() const { return has_pw_; }
  unsigned kernel_h() const { return kh_; }
  unsigned kernel_w() const { return kw_; }
  unsigned stride_h() const { return sh_; }
  unsigned stride_w() const { return sw_; }
  unsigned pad_h() const { return ph_; }
  unsigned pad_w() const { return pw_; }
};
class LayerParameter {
 public:
  PoolingParameter pooling_param_;
  const PoolingParameter& pooling_param() const { return pooling_param_; }
  PoolingParameter* mutable_pooling_param() { return &pooling_param_; }
};
}  // namespace caffe
