// oracle/ref_shim: host-only Blob with the members the compiled reference files touch (shape bookkeeping as in
// include/caffe/blob.hpp:24-282: row-major, offset(n,c,h,w), legacy accessors).
#pragma once
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"
namespace caffe {
template <typename Dtype>
class Blob {
 public:
  Blob() {}
  void Reshape(const vector<int>& shape) {
    shape_ = shape;
    count_ = 1;
    for (size_t i = 0; i < shape.size(); ++i) count_ *= shape[i];
    if ((int)data_.size() < count_) { data_.resize(count_); diff_.resize(count_); }
  }
  void ReshapeLike(const Blob& o) { Reshape(o.shape()); }
  const vector<int>& shape() const { return shape_; }
  int shape(int i) const { return shape_[i < 0 ? i + (int)shape_.size() : i]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  int LegacyShape(int i) const { CHECK_LE(num_axes(), 4); return i < num_axes() ? shape_[i] : 1; }
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int offset(const int n, const int c = 0, const int h = 0, const int w = 0) const {
    return ((n * channels() + c) * height() + h) * width() + w;
  }
  const Dtype* cpu_data() const { return data_.data(); }
  Dtype* mutable_cpu_data() { return data_.data(); }
  const Dtype* cpu_diff() const { return diff_.data(); }
  Dtype* mutable_cpu_diff() { return diff_.data(); }
 private:
  vector<int> shape_;
  int count_ = 0;
  vector<Dtype> data_, diff_;
};
}  // namespace caffe
