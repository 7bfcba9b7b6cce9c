// oracle/ref_shim: the slice of Layer<Dtype> (include/caffe/layer.hpp:28-345) a CPU forward needs.
#pragma once
#include "caffe/blob.hpp"
#include "caffe/layer_factory.hpp"
namespace caffe {
template <typename Dtype>
class Layer {
 public:
  explicit Layer(const LayerParameter& param) : layer_param_(param), phase_(TEST) {}
  virtual ~Layer() {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom) = 0;
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom) {}
  const LayerParameter& layer_param() const { return layer_param_; }
  vector<shared_ptr<Blob<Dtype> > >& blobs() { return blobs_; }
  void set_phase(Phase p) { phase_ = p; }
 protected:
  LayerParameter layer_param_;
  Phase phase_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;     // the layer's learnable parameters (layer.hpp:296)
  vector<bool> param_propagate_down_;
};
}  // namespace caffe
