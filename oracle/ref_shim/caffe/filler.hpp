// oracle/ref_shim: Filler / GetFiller (include/caffe/filler.hpp) reduced to the constant filler -- the compiled
// LayerSetUp bodies create parameter blobs through it; ref_api.cpp then overwrites them with the caller's values.
#pragma once
#include "caffe/blob.hpp"
namespace caffe {
template <typename Dtype>
class Filler {
 public:
  explicit Filler(const FillerParameter& param) : filler_param_(param) {}
  virtual ~Filler() {}
  virtual void Fill(Blob<Dtype>* blob) {
    Dtype* d = blob->mutable_cpu_data();
    for (int i = 0; i < blob->count(); ++i) d[i] = Dtype(filler_param_.value());
  }
 protected:
  FillerParameter filler_param_;
};
template <typename Dtype>
Filler<Dtype>* GetFiller(const FillerParameter& param) {
  CHECK(param.type() == "constant") << "ref_shim: only the constant filler exists here";
  return new Filler<Dtype>(param);
}
}  // namespace caffe
