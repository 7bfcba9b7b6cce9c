// oracle/ref_shim: host-only Blob with the members the compiled reference files touch (shape bookkeeping as in
// include/caffe/blob.hpp:24-282: row-major, offset(n,c,h,w), legacy accessors, count(start[, end]),
// CanonicalAxisIndex, ShareData / ShareDiff).
#pragma once
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"
namespace caffe {
const int kMaxBlobAxes = INT_MAX;   // blob.hpp:13 of this fork (N-D blobs)
template <typename Dtype>
class Blob {
 public:
  Blob() : data_(new vector<Dtype>()), diff_(new vector<Dtype>()) {}
  explicit Blob(const vector<int>& shape) : data_(new vector<Dtype>()), diff_(new vector<Dtype>()) { Reshape(shape); }
  void Reshape(const vector<int>& shape) {
    shape_ = shape;
    count_ = 1;
    for (size_t i = 0; i < shape.size(); ++i) count_ *= shape[i];
    if ((int)data_->size() < count_) { data_->resize(count_); diff_->resize(count_); }
  }
  void Reshape(const int num, const int channels, const int height, const int width) {
    vector<int> s(4);
    s[0] = num; s[1] = channels; s[2] = height; s[3] = width;
    Reshape(s);
  }
  void ReshapeLike(const Blob& o) { Reshape(o.shape()); }
  const vector<int>& shape() const { return shape_; }
  int shape(int i) const { return shape_[CanonicalAxisIndex(i)]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  string shape_string() const {   // blob.hpp:56-63
    std::ostringstream stream;
    for (size_t i = 0; i < shape_.size(); ++i) stream << shape_[i] << " ";
    stream << "(" << count_ << ")";
    return stream.str();
  }
  int count(int start_axis, int end_axis) const {
    CHECK_LE(start_axis, end_axis);
    CHECK_GE(start_axis, 0);
    CHECK_LE(end_axis, num_axes());
    int c = 1;
    for (int i = start_axis; i < end_axis; ++i) c *= shape_[i];
    return c;
  }
  int count(int start_axis) const { return count(start_axis, num_axes()); }
  int CanonicalAxisIndex(int axis_index) const {
    CHECK_GE(axis_index, -num_axes());
    CHECK_LT(axis_index, num_axes());
    return axis_index < 0 ? axis_index + num_axes() : axis_index;
  }
  int LegacyShape(int i) const { CHECK_LE(num_axes(), 4); return i < num_axes() ? shape_[i] : 1; }
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int offset(const int n, const int c = 0, const int h = 0, const int w = 0) const {
    return ((n * channels() + c) * height() + h) * width() + w;
  }
  const Dtype* cpu_data() const { return data_->data(); }
  Dtype* mutable_cpu_data() { return data_->data(); }
  const Dtype* cpu_diff() const { return diff_->data(); }
  Dtype* mutable_cpu_diff() { return diff_->data(); }
  void ShareData(const Blob& other) { CHECK_EQ(count_, other.count()); data_ = other.data_; }
  void ShareDiff(const Blob& other) { CHECK_EQ(count_, other.count()); diff_ = other.diff_; }
 private:
  vector<int> shape_;
  int count_ = 0;
  shared_ptr<vector<Dtype> > data_, diff_;
};
}  // namespace caffe
