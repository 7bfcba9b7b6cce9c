// oracle/ref_shim: hand-written stand-in for the protoc-generated caffe.pb.h -- only the messages and accessors the
// compiled reference layer files call (schema: caffe_3d/src/caffe/proto/caffe.proto; ConvolutionParameter,
// PoolingParameter, BNParameter,
// PermuteParameter, EltwiseParameter, ConcatParameter, InnerProductParameter, ReshapeParameter / BlobShape,
// ReLUParameter, FillerParameter, ParamSpec, LayerParameter).  Defaults are the schema's.
#pragma once
#include <string>
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
// caffe.proto FillerParameter: only "constant" is ever instantiated here (parameters are supplied by the caller)
class FillerParameter {
 public:
  std::string type_ = "constant";
  float value_ = 0.0f;
  const std::string& type() const { return type_; }
  float value() const { return value_; }
};
class ConvolutionParameter {   // caffe.proto:506-555 (group 1, axis 1, bias_term true, force_nd_im2col false by default)
 public:
  unsigned num_output_ = 0, group_ = 1;
  bool bias_term_ = true, force_nd_im2col_ = false;
  int axis_ = 1;
  std::vector<unsigned> kernel_size_, stride_, pad_;
  bool has_kh_ = false, has_kw_ = false, has_sh_ = false, has_sw_ = false, has_ph_ = false, has_pw_ = false;
  unsigned kh_ = 0, kw_ = 0, sh_ = 0, sw_ = 0, ph_ = 0, pw_ = 0;
  FillerParameter weight_filler_, bias_filler_;
  unsigned num_output() const { return num_output_; }
  unsigned group() const { return group_; }
  bool bias_term() const { return bias_term_; }
  bool force_nd_im2col() const { return force_nd_im2col_; }
  int axis() const { return axis_; }
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
  const FillerParameter& weight_filler() const { return weight_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
};
class ParamSpec {
 public:
  float lr_mult_ = 1.0f, decay_mult_ = 1.0f;
  void set_lr_mult(float v) { lr_mult_ = v; }
  void set_decay_mult(float v) { decay_mult_ = v; }
};
// the slice of RepeatedPtrField<ParamSpec> bn_layer.cpp uses
class ParamSpecList {
 public:
  std::vector<ParamSpec> v_;
  ParamSpec* Add() { v_.emplace_back(); return &v_.back(); }
};
class BNParameter {   // caffe.proto: slope_filler (1), bias_filler (0), momentum 0.9, eps 1e-5, frozen false
 public:
  FillerParameter slope_filler_, bias_filler_;
  float momentum_ = 0.9f, eps_ = 1e-5f;
  bool frozen_ = false;
  BNParameter() { slope_filler_.value_ = 1.0f; }
  const FillerParameter& slope_filler() const { return slope_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
  float momentum() const { return momentum_; }
  float eps() const { return eps_; }
  bool frozen() const { return frozen_; }
};
class PermuteParameter {
 public:
  std::vector<unsigned> order_;
  int order_size() const { return (int)order_.size(); }
  unsigned order(int i) const { return order_[i]; }
};
enum EltwiseParameter_EltwiseOp {
  EltwiseParameter_EltwiseOp_PROD = 0, EltwiseParameter_EltwiseOp_SUM = 1, EltwiseParameter_EltwiseOp_MAX = 2,
  EltwiseParameter_EltwiseOp_STOCHASTIC_SUM = 3
};
class EltwiseParameter {
 public:
  EltwiseParameter_EltwiseOp operation_ = EltwiseParameter_EltwiseOp_SUM;
  std::vector<float> coeff_;
  bool stable_prod_grad_ = true;
  EltwiseParameter_EltwiseOp operation() const { return operation_; }
  int coeff_size() const { return (int)coeff_.size(); }
  float coeff(int i) const { return coeff_[i]; }
  bool stable_prod_grad() const { return stable_prod_grad_; }
};
class ConcatParameter {
 public:
  int axis_ = 1;
  unsigned concat_dim_ = 1;
  bool has_axis_ = false, has_concat_dim_ = false;
  bool has_axis() const { return has_axis_; }
  bool has_concat_dim() const { return has_concat_dim_; }
  int axis() const { return axis_; }
  unsigned concat_dim() const { return concat_dim_; }
};
class InnerProductParameter {
 public:
  unsigned num_output_ = 0;
  bool bias_term_ = true;
  int axis_ = 1;
  FillerParameter weight_filler_, bias_filler_;
  unsigned num_output() const { return num_output_; }
  bool bias_term() const { return bias_term_; }
  int axis() const { return axis_; }
  const FillerParameter& weight_filler() const { return weight_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
};
class BlobShape {
 public:
  std::vector<long long> dim_;
  int dim_size() const { return (int)dim_.size(); }
  long long dim(int i) const { return dim_[i]; }
};
class ReshapeParameter {
 public:
  BlobShape shape_;
  int axis_ = 0, num_axes_ = -1;
  const BlobShape& shape() const { return shape_; }
  int axis() const { return axis_; }
  int num_axes() const { return num_axes_; }
};
class ReLUParameter {
 public:
  float negative_slope_ = 0.0f;
  float negative_slope() const { return negative_slope_; }
};
class LayerParameter {
 public:
  ConvolutionParameter convolution_param_;
  PoolingParameter pooling_param_;
  BNParameter bn_param_;
  PermuteParameter permute_param_;
  EltwiseParameter eltwise_param_;
  ConcatParameter concat_param_;
  InnerProductParameter inner_product_param_;
  ReshapeParameter reshape_param_;
  ReLUParameter relu_param_;
  ParamSpecList param_;
  const ConvolutionParameter& convolution_param() const { return convolution_param_; }
  const PoolingParameter& pooling_param() const { return pooling_param_; }
  PoolingParameter* mutable_pooling_param() { return &pooling_param_; }
  const BNParameter& bn_param() const { return bn_param_; }
  const PermuteParameter& permute_param() const { return permute_param_; }
  const EltwiseParameter& eltwise_param() const { return eltwise_param_; }
  const ConcatParameter& concat_param() const { return concat_param_; }
  const InnerProductParameter& inner_product_param() const { return inner_product_param_; }
  const ReshapeParameter& reshape_param() const { return reshape_param_; }
  const ReLUParameter& relu_param() const { return relu_param_; }
  int param_size() const { return (int)param_.v_.size(); }
  ParamSpecList* mutable_param() { return &param_; }
  ParamSpec* mutable_param(int i) { return &param_.v_[i]; }
};
}  // namespace caffe
