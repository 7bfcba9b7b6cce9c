// oracle/ref_shim: declarations of the layer classes whose method BODIES come from the reference's own .cpp files
// (data members as in include/caffe/common_layers.hpp / vision_layers.hpp / neuron_layers.hpp of caffe_3d; every
// Forward_cpu is made public here so that ref_api.cpp can call it without Layer::Forward's GPU plumbing).
#pragma once
#include "caffe/filler.hpp"
#include "caffe/layer.hpp"
namespace caffe {
#define REF_SHIM_LAYER_METHODS(name)                                                                          \
  explicit name(const LayerParameter& param) : Layer<Dtype>(param) {}                                          \
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);                \
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);                   \
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);               \
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);               \
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,               \
                            const vector<Blob<Dtype>*>& bottom);                                               \
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,               \
                            const vector<Blob<Dtype>*>& bottom);

template <typename Dtype>
class BNLayer : public Layer<Dtype> {   // common_layers.hpp:780-825
 public:
  REF_SHIM_LAYER_METHODS(BNLayer)
 protected:
  bool frozen_;
  Dtype bn_momentum_, bn_eps_;
  int num_, channels_, height_, width_;
  Blob<Dtype> broadcast_buffer_, spatial_statistic_, batch_statistic_, x_norm_, x_inv_std_;
  Blob<Dtype> spatial_sum_multiplier_, batch_sum_multiplier_;
};

template <typename Dtype>
class PermuteLayer : public Layer<Dtype> {   // common_layers.hpp:600-634
 public:
  REF_SHIM_LAYER_METHODS(PermuteLayer)
 protected:
  int num_axes_;
  bool need_permute_;
  Blob<int> permute_order_, old_steps_, new_steps_;
};

template <typename Dtype>
class EltwiseLayer : public Layer<Dtype> {   // common_layers.hpp:157-188
 public:
  REF_SHIM_LAYER_METHODS(EltwiseLayer)
 protected:
  EltwiseParameter_EltwiseOp op_;
  vector<Dtype> coeffs_;
  Blob<int> max_idx_;
  Blob<Dtype> rng_buffer_;
  bool stable_prod_grad_;
};

template <typename Dtype>
class ConcatLayer : public Layer<Dtype> {   // common_layers.hpp:82-150
 public:
  REF_SHIM_LAYER_METHODS(ConcatLayer)
 protected:
  int count_, num_concats_, concat_input_size_, concat_axis_;
};

template <typename Dtype>
class InnerProductLayer : public Layer<Dtype> {   // common_layers.hpp:313-345
 public:
  REF_SHIM_LAYER_METHODS(InnerProductLayer)
 protected:
  int M_, K_, N_;
  bool bias_term_;
  Blob<Dtype> bias_multiplier_;
};

template <typename Dtype>
class ReshapeLayer : public Layer<Dtype> {   // common_layers.hpp:380-420: Forward / Backward are empty in the header
 public:
  explicit ReshapeLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom) {}
 protected:
  vector<int> copy_axes_;
  int inferred_axis_;
  int constant_count_;
};

template <typename Dtype>
class ReLULayer : public Layer<Dtype> {   // neuron_layers.hpp (NeuronLayer::Reshape = top like bottom, neuron_layer.cpp:9-12)
 public:
  explicit ReLULayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { top[0]->ReshapeLike(*bottom[0]); }
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
};
}  // namespace caffe
