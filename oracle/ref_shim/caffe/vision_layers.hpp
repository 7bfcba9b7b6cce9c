// oracle/ref_shim: declarations of PoolingLayer (include/caffe/vision_layers.hpp:468-519) and of BaseConvolutionLayer /
// ConvolutionLayer (:26-170, :172-247) with the data members layers/pooling_layer.cpp, base_conv_layer.cpp and
// conv_layer.cpp define their methods over.  Every out-of-line method BODY comes from the reference .cpp files; the
// only code here is the pair of private one-call dispatchers the reference keeps inside the class declaration
// (conv_im2col_cpu / conv_col2im_cpu, :102-127: the 2-D routine unless force_nd_im2col or another rank).
// (The reference's vision_layers.hpp pulls in common_layers.hpp / neuron_layers.hpp, which eltwise_layer.cpp,
// concat_layer.cpp, inner_product_layer.cpp and relu_layer.cpp rely on: same here.)
#pragma once
#include "caffe/common_layers.hpp"
#include "caffe/layer.hpp"
#include "caffe/util/im2col.hpp"
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
template <typename Dtype>
class BaseConvolutionLayer : public Layer<Dtype> {
 public:
  explicit BaseConvolutionLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
 protected:
  void forward_cpu_gemm(const Dtype* input, const Dtype* weights, Dtype* output, bool skip_im2col = false);
  void forward_cpu_bias(Dtype* output, const Dtype* bias);
  void backward_cpu_gemm(const Dtype* input, const Dtype* weights, Dtype* output);
  void weight_cpu_gemm(const Dtype* input, const Dtype* output, Dtype* weights);
  void backward_cpu_bias(Dtype* bias, const Dtype* input);
  virtual bool reverse_dimensions() = 0;
  virtual void compute_output_shape() = 0;

  Blob<int> kernel_shape_, stride_, pad_, conv_input_shape_, input_shape_;
  vector<int> col_buffer_shape_, output_shape_;
  int num_spatial_axes_, bottom_dim_, top_dim_, channel_axis_, num_, channels_, out_spatial_dim_, weight_offset_;
  int group_, num_output_;
  bool force_nd_im2col_, bias_term_, is_1x1_;

 private:
  bool two_d() const { return !force_nd_im2col_ && num_spatial_axes_ == 2; }
  void conv_im2col_cpu(const Dtype* data, Dtype* col_buff) {
    const int *in = conv_input_shape_.cpu_data(), *k = kernel_shape_.cpu_data(), *p = pad_.cpu_data(), *s = stride_.cpu_data();
    if (two_d()) im2col_cpu(data, conv_in_channels_, in[1], in[2], k[0], k[1], p[0], p[1], s[0], s[1], col_buff);
    else im2col_nd_cpu(data, num_spatial_axes_, in, col_buffer_shape_.data(), k, p, s, col_buff);
  }
  void conv_col2im_cpu(const Dtype* col_buff, Dtype* data) {
    const int *in = conv_input_shape_.cpu_data(), *k = kernel_shape_.cpu_data(), *p = pad_.cpu_data(), *s = stride_.cpu_data();
    if (two_d()) col2im_cpu(col_buff, conv_in_channels_, in[1], in[2], k[0], k[1], p[0], p[1], s[0], s[1], data);
    else col2im_nd_cpu(col_buff, num_spatial_axes_, in, col_buffer_shape_.data(), k, p, s, data);
  }
  int num_kernels_im2col_, num_kernels_col2im_, conv_out_channels_, conv_in_channels_, conv_out_spatial_dim_;
  int kernel_dim_, col_offset_, output_offset_;
  Blob<Dtype> col_buffer_, bias_multiplier_;
};

template <typename Dtype>
class ConvolutionLayer : public BaseConvolutionLayer<Dtype> {
 public:
  explicit ConvolutionLayer(const LayerParameter& param) : BaseConvolutionLayer<Dtype>(param) {}
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom);
 protected:
  virtual bool reverse_dimensions() { return false; }
  virtual void compute_output_shape();
};
}  // namespace caffe
