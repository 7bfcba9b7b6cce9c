// oracle/ref_shim: declaration of PoolingLayer with the data members layers/pooling_layer.cpp defines its
// methods over (include/caffe/vision_layers.hpp:468-519).  Every method BODY comes from the reference .cpp.
// (The reference's vision_layers.hpp pulls in common_layers.hpp / neuron_layers.hpp, which eltwise_layer.cpp,
// concat_layer.cpp, inner_product_layer.cpp and relu_layer.cpp rely on: same here.)
#pragma once
#include "caffe/common_layers.hpp"
#include "caffe/layer.hpp"
namespace caffe {
template <typename Dtype>
class PoolingLayer : public Layer<Dtype> {
 public:
  explicit PoolingLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
  const std::vector<int>& pooled_shape() const { return pooled_shape_; }
 protected:
  std::vector<int> kernel_shape_, stride_, pad_;
  int num_spatial_axes_;
  int channels_;
  std::vector<int> input_shape_, pooled_shape_;
  bool global_pooling_;
  Blob<Dtype> rand_idx_;
  Blob<int> max_idx_;
};
}  // namespace caffe
