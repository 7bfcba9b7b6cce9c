// eco_ops.hip -- the memory-bound operators of the ECO path as stand-alone gfx950 kernels:
// pooling (2-D / 3-D, MAX / AVE), folded BN (+ReLU), ReLU, Eltwise SUM, Concat copy,
// Permute, InnerProduct, the fused global-avg-pool + fc tail, Softmax.
// Reference operators are cited at each entry point in include/eco_hip.h.
// All of these are HBM/L2-bandwidth bound: consecutive lanes touch consecutive addresses,
// reductions use 64-lane butterfly shuffles (no LDS round trip), nothing is reshaped into
// a GEMM.
#include <float.h>

#include "eco_common.h"

namespace eco {

constexpr int kThreads = 256;

static inline int grid_for(long count, long per_block = kThreads) {
  long g = ceil_div(count, per_block);
  if (g < 1) g = 1;
  if (g > 1048576) g = 1048576;  // grid-stride beyond this
  return (int)g;
}

// ---------------------------------------------------------------------------- pooling
struct PoolArgs {
  const float* x;
  float* y;
  long total;  // n*c*Do*Ho*Wo
  int Di, Hi, Wi, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int method;
  int c;           // channels per image
  long y_extra;    // floats between the images of y beyond c * Do*Ho*Wo (0: dense; > 0: y is a channel slice of a Concat top)
};

__global__ __launch_bounds__(256) void pool_kernel(const PoolArgs a) {
  const long s_in = (long)a.Di * a.Hi * a.Wi;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < a.total; i += (long)gridDim.x * kThreads) {
    const int ow = (int)(i % a.Wo);
    long t = i / a.Wo;
    const int oh = (int)(t % a.Ho);
    t /= a.Ho;
    const int od = (int)(t % a.Do);
    const long nc = t / a.Do;
    const float* xp = a.x + nc * s_in;
    int ds = od * a.sd - a.pd, hs = oh * a.sh - a.ph, ws = ow * a.sw - a.pw;
    float r;
    if (a.method == ECO_POOL_MAX) {
      const int de = min(ds + a.kd, a.Di), he = min(hs + a.kh, a.Hi), we = min(ws + a.kw, a.Wi);
      ds = max(ds, 0); hs = max(hs, 0); ws = max(ws, 0);
      r = -FLT_MAX;
      for (int d = ds; d < de; ++d)
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) r = fmaxf(r, ld(xp + ((long)d * a.Hi + h) * a.Wi + w));
    } else {
      int de = min(ds + a.kd, a.Di + a.pd), he = min(hs + a.kh, a.Hi + a.ph), we = min(ws + a.kw, a.Wi + a.pw);
      const float size = (float)((de - ds) * (he - hs) * (we - ws));
      ds = max(ds, 0); hs = max(hs, 0); ws = max(ws, 0);
      de = min(de, a.Di); he = min(he, a.Hi); we = min(we, a.Wi);
      r = 0.0f;
      for (int d = ds; d < de; ++d)
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) r += ld(xp + ((long)d * a.Hi + h) * a.Wi + w);
      r /= size;
    }
    st(a.y + i + (a.y_extra ? (nc / a.c) * a.y_extra : 0l), r);
  }
}

// 2-D 3x3 windows of any stride / pad (the planes the vector fast paths below do not take: 7x7 and 14x14 -> 7x7 in
// ECO-Full's inception_4e / 5a / 5b pools): the nine loads of an output are independent and all in flight before the
// first is used -- the generic kernel's runtime-bounded loops fetch them one round trip at a time.  Same values,
// same order of the sum / max.
template <int METHOD>
__global__ __launch_bounds__(256) void pool2d_k3_kernel(const PoolArgs a) {
  const long s_in = (long)a.Hi * a.Wi;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < a.total; i += (long)gridDim.x * kThreads) {
    const int ow = (int)(i % a.Wo);
    const long t = i / a.Wo;
    const int oh = (int)(t % a.Ho);
    const long nc = t / a.Ho;
    const float* xp = a.x + nc * s_in;
    const int hs = oh * a.sh - a.ph, ws = ow * a.sw - a.pw;
    float v[3][3];
    bool ok[3][3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int h = hs + dh, w = ws + dw;
        ok[dh][dw] = (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
        v[dh][dw] = ld(xp + (ok[dh][dw] ? (long)h * a.Wi + w : 0l));
      }
    float r = METHOD == ECO_POOL_MAX ? -FLT_MAX : 0.0f;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw)
        if (ok[dh][dw]) r = METHOD == ECO_POOL_MAX ? fmaxf(r, v[dh][dw]) : r + v[dh][dw];
    if (METHOD != ECO_POOL_MAX) {   // divisor: the window clipped to the padded image (pooling_layer.cpp:240-262)
      const int he = min(hs + 3, a.Hi + a.ph), we = min(ws + 3, a.Wi + a.pw);
      r /= (float)((he - hs) * (we - ws));
    }
    st(a.y + i + (a.y_extra ? (nc / a.c) * a.y_extra : 0l), r);
  }
}

// VEC consecutive floats as one 16-, 8- or 4-byte access.
template <int VEC>
__device__ __forceinline__ void ld_vec(const float* p, float (&v)[VEC]) {
  if (VEC == 4) {
    const float4 q = ld((const float4*)p);
    v[0] = q.x; v[1 % VEC] = q.y; v[2 % VEC] = q.z; v[3 % VEC] = q.w;
  } else if (VEC == 2) {
    const float2 q = ld((const float2*)p);
    v[0] = q.x; v[1 % VEC] = q.y;
  } else {
    v[0] = ld(p);
  }
}
template <int VEC>
__device__ __forceinline__ void st_vec(float* p, const float (&v)[VEC]) {
  if (VEC == 4) st((float4*)p, make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]));
  else if (VEC == 2) st((float2*)p, make_float2(v[0], v[1 % VEC]));
  else st(p, v[0]);
}

// 2-D MAX 3x3 stride 2, no padding (pool1 / pool2 / the ECO-Full stride-2 pools): one thread per VEC
// (4 or 2) consecutive outputs of a row.  It reads 2*VEC (+1) consecutive input floats of each of the
// three rows as two vector loads (+1 scalar) and writes one vector -> global accesses are 16 (8) bytes
// per lane and contiguous across lanes.  Ceil-mode clipping (pooling_layer.cpp:131-147,199-225): the
// last column and the 3rd row simply do not exist at the right / bottom edge.
template <int VEC>
__global__ __launch_bounds__(256) void maxpool2d_k3s2_kernel(const float* x, float* y, long planes, int Hi, int Wi,
                                                             int Ho, int Wo, int c, long y_extra) {
  const int wq = Wo / VEC;
  const long total = planes * Ho * wq;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int q = (int)(i % wq);
    const long t = i / wq;
    const int oh = (int)(t % Ho);
    const long pl = t / Ho;
    const float* xp = x + pl * Hi * Wi + (long)(2 * oh) * Wi + 2 * VEC * q;
    const bool has_last = 2 * VEC * q + 2 * VEC < Wi;
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (2 * oh + r >= Hi) break;
      float in[2 * VEC + 1], lo[VEC], hi[VEC];
      ld_vec<VEC>(xp + (long)r * Wi, lo);
      ld_vec<VEC>(xp + (long)r * Wi + VEC, hi);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { in[e] = lo[e]; in[VEC + e] = hi[e]; }
      in[2 * VEC] = has_last ? ld(xp + (long)r * Wi + 2 * VEC) : -FLT_MAX;
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], fmaxf(fmaxf(in[2 * e], in[2 * e + 1]), in[2 * e + 2]));
    }
    st_vec<VEC>(y + (pl * Ho + oh) * Wo + VEC * q + (y_extra ? (pl / c) * y_extra : 0l), m);   // (y_extra: a Concat slice)
  }
}

// 2-D AVE 3x3 stride 1 pad 1 (inception *_pool): VEC consecutive outputs of kAvgRows consecutive rows per thread.
// Every input row is fetched once per thread (one vector load plus the two neighbours) and its horizontal
// 3-sums feed up to three output rows, so the rows are read 1.5x instead of 3x.  Zero padding, divisor 9
// everywhere (the reference's window size including padding, pooling_layer.cpp:247-262, equals 9 for every
// position of this geometry); rows are accumulated top to bottom as before.
constexpr int kAvgRows = 4;
template <int VEC>
__global__ __launch_bounds__(256) void avgpool2d_k3s1p1_kernel(const float* x, float* y, long planes, int H, int W) {
  const int wq = W / VEC;
  const int hq = (H + kAvgRows - 1) / kAvgRows;
  const long total = planes * hq * wq;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int q = (int)(i % wq);
    const long t = i / wq;
    const int oh0 = (int)(t % hq) * kAvgRows;
    const long pl = t / hq;
    const float* xp = x + pl * H * W + VEC * q;
    float rs[kAvgRows + 2][VEC];  // horizontal 3-sums of input rows oh0-1 .. oh0+kAvgRows
#pragma unroll
    for (int r = 0; r < kAvgRows + 2; ++r) {
      const int h = oh0 - 1 + r;
      const bool ok = h >= 0 && h < H;
      const float* row = xp + (ok ? (long)h * W : 0l);
      float in[VEC + 2], mid[VEC];
      ld_vec<VEC>(row, mid);
      in[0] = (q > 0) ? ld(row - 1) : 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) in[1 + e] = mid[e];
      in[VEC + 1] = (VEC * q + VEC < W) ? ld(row + VEC) : 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) rs[r][e] = ok ? in[e] + in[e + 1] + in[e + 2] : 0.0f;
    }
    const float inv = 1.0f / 9.0f;
#pragma unroll
    for (int o = 0; o < kAvgRows; ++o) {
      const int oh = oh0 + o;
      if (oh >= H) break;
      float sum[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) sum[e] = ((0.0f + rs[o][e]) + rs[o + 1][e] + rs[o + 2][e]) * inv;
      st_vec<VEC>(y + (pl * H + oh) * W + VEC * q, sum);
    }
  }
}

// The same pooling followed by a per-channel affine map and ReLU, written through a strided view:
//   y(img, ch, :) = relu(((avgpool3x3(x) + bias[ch]) * scale[ch]) + shift[ch]).
// This is what remains of  AVE pool 3x3/1/1 -> 1x1 conv -> BN -> ReLU  (inception_3a/3b_pool + pool_proj,
// models_ECO_Lite/kinetics/deploy.prototxt:330-400) once the two linear maps are exchanged: the 1x1 convolution is
// applied to the block's input (as one more member of the block's sibling 1x1 launch, without its bias) and the
// window average -- zero padding, constant divisor 9: a linear map with constant coefficients -- to its cout output
// channels instead of to the cin input channels (192 / 256 -> 32 / 64 in ECO).  conv(avg(x)) = avg(conv(x)) exactly
// in exact arithmetic; in fp32 the two orders differ by rounding.
template <int VEC>
__global__ __launch_bounds__(256) void avgpool2d_k3s1p1_affine_kernel(const float* x, const float* bias, const float* scale,
                                                                      const float* shift, float floor_v, eco_view dst,
                                                                      long planes, int C, int H, int W) {
  const int wq = W / VEC;
  const int hq = (H + kAvgRows - 1) / kAvgRows;
  const long total = planes * hq * wq;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int q = (int)(i % wq);
    const long t = i / wq;
    const int oh0 = (int)(t % hq) * kAvgRows;
    const long pl = t / hq;
    const float* xp = x + pl * H * W + VEC * q;
    float rs[kAvgRows + 2][VEC];  // horizontal 3-sums of input rows oh0-1 .. oh0+kAvgRows
#pragma unroll
    for (int r = 0; r < kAvgRows + 2; ++r) {
      const int h = oh0 - 1 + r;
      const bool ok = h >= 0 && h < H;
      const float* row = xp + (ok ? (long)h * W : 0l);
      float in[VEC + 2], mid[VEC];
      ld_vec<VEC>(row, mid);
      in[0] = (q > 0) ? ld(row - 1) : 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) in[1 + e] = mid[e];
      in[VEC + 1] = (VEC * q + VEC < W) ? ld(row + VEC) : 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) rs[r][e] = ok ? in[e] + in[e + 1] + in[e + 2] : 0.0f;
    }
    const int img = (int)(pl / C), ch = (int)(pl - (long)img * C);
    const float b = bias ? ld(bias + ch) : 0.0f;
    const float sc = scale ? ld(scale + ch) : 1.0f, sh = scale ? ld(shift + ch) : 0.0f;
    float* yp = dst.ptr + view_base(dst, img, 0) + (long)ch * dst.stride_c + VEC * q;
    const float inv = 1.0f / 9.0f;
#pragma unroll
    for (int o = 0; o < kAvgRows; ++o) {
      const int oh = oh0 + o;
      if (oh >= H) break;
      float out[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        out[e] = fmaxf((((0.0f + rs[o][e]) + rs[o + 1][e] + rs[o + 2][e]) * inv + b) * sc + sh, floor_v);
      st_vec<VEC>(yp + (long)oh * W, out);
    }
  }
}

// Whole-volume average (global_pool): one wave per (n,c) row, butterfly reduction.
__global__ __launch_bounds__(256) void global_avg_kernel(const float* x, float* y, long rows, int s) {
  const int lane = lane_id();
  const int wave = uniform((int)(threadIdx.x >> 6));
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xp = x + row * s;
    float acc = 0.0f;
    for (int i = lane; i < s; i += kWave) acc += ld(xp + i);
    acc = wave_sum(acc);
    if (lane == 0) st(y + row, acc / (float)s);
  }
}

// ---------------------------------------------------------------------------- elementwise
__global__ __launch_bounds__(256) void bn_kernel(const float* x, float* y, const float* scale, const float* shift,
                                                 long total, long c, long inner, int relu) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long ch = (i / inner) % c;
    float v = ld(x + i) * ld(scale + ch) + ld(shift + ch);
    if (relu) v = fmaxf(v, 0.0f);
    st(y + i, v);
  }
}

__global__ __launch_bounds__(256) void relu_kernel(const float* x, float* y, long total, float slope) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const float v = ld(x + i);
    st(y + i, fmaxf(v, 0.0f) + slope * fminf(v, 0.0f));
  }
}

__global__ __launch_bounds__(256) void eltwise_sum_kernel(const float* a, const float* b, float* y, long total,
                                                          float ca, float cb) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads)
    st(y + i, ca * ld(a + i) + cb * ld(b + i));
}

__global__ __launch_bounds__(256) void concat_copy_kernel(const float* x, float* y, long total, long cx, long cy,
                                                          long c0, long inner) {
  const long per_outer = cx * inner;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long o = i / per_outer, r = i - o * per_outer;
    st(y + (o * cy + c0) * inner + r, ld(x + i));
  }
}

struct PermuteArgs {
  const float* x;
  float* y;
  long total;
  int naxes;
  int out_shape[6];
  long in_stride_of_out_axis[6];
};

__global__ __launch_bounds__(256) void permute_kernel(const PermuteArgs a) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < a.total; i += (long)gridDim.x * kThreads) {
    long rem = i, src = 0;
    for (int ax = a.naxes - 1; ax >= 0; --ax) {
      const long idx = rem % a.out_shape[ax];
      rem /= a.out_shape[ax];
      src += idx * a.in_stride_of_out_axis[ax];
    }
    st(a.y + i, ld(a.x + src));
  }
}

// ---------------------------------------------------------------------------- fc / tail
// One wave per output element: lanes stride over K, butterfly-reduce.
__global__ __launch_bounds__(256) void inner_product_kernel(const float* x, const float* w, const float* bias,
                                                            float* y, long m, long n, long k) {
  const int lane = lane_id();
  const int wave = uniform((int)(threadIdx.x >> 6));
  const long total = m * n;
  for (long o = (long)blockIdx.x * 4 + wave; o < total; o += (long)gridDim.x * 4) {
    const long mi = o / n, ni = o - mi * n;
    const float* xp = x + mi * k;
    const float* wp = w + ni * k;
    float acc = 0.0f;
    for (long i = lane; i < k; i += kWave) acc += ld(xp + i) * ld(wp + i);
    acc = wave_sum(acc);
    if (lane == 0) st(y + o, acc + (bias ? ld(bias + ni) : 0.0f));
  }
}

constexpr int kTailMaxC = 2048;
constexpr int kTailThreads = 1024;        // 16 waves per workgroup
// Workgroups to aim for: every one of them pools its clip's whole volume again, so a big volume (ECO-Full's 2-D
// stream: 16 x 1024 x 49 floats per clip) is cut into few logit blocks and a small one (the 3-D stream: 512 x 196)
// into many -- there the fc rows, one wave per logit, are the longer phase.
constexpr int kTailWorkgroupsBigVolume = 64, kTailWorkgroupsSmallVolume = 256;

// grid = (logit blocks, b), 1024 threads.  Each workgroup pools its clip's C channels into LDS (one wave per
// channel, four loads in flight per lane, 64-lane butterfly reduce), then its 16 waves produce its block of logits
// (one wave per logit: lanes stride over C, butterfly reduce).  The logits of a clip are cut into blocks so that
// 64-256 workgroups run whatever the batch (a single clip's 400 x 512 fc on one CU took 0.155 ms of a 1.75 ms
// online step); every block pools the clip again, from L2.
// `t` > 1: the clip's volume is spread over t consecutive images of c x s each (a 2-D stream's frames,
// x[b*t + f][c][s]) and the mean runs over all t*s values of a channel: 2-D global pool + segment consensus.
__global__ __launch_bounds__(1024) void global_avgpool_fc_kernel(const float* x, const float* w, const float* bias,
                                                                 float* y, int c, int s, int t, int n_out, int wk,
                                                                 int c0, int accumulate, int out_per_block) {
  __shared__ float pooled[kTailMaxC];
  constexpr int kWaves = kTailThreads / kWave;
  const int lane = lane_id();
  const int wave = uniform((int)(threadIdx.x >> 6));
  const int b = (int)blockIdx.y;
  const float* xb = x + (long)b * t * c * s;
  const float inv = 1.0f / ((float)s * (float)t);
  const long fstride = (long)c * s;
  // one channel per wave at a time; a lane owns positions lane, lane+64, ... of each of the t frames (no division in
  // the loop, four independent partial sums so that four loads are in flight)
  // single-frame volumes of 65 .. 256 positions (the 3-D tail at num_segments 16: 4 x 7 x 7 = 196; the 8 x 7 x 7 = 392 of
  // num_segments 32 takes the one-channel loop below, as do the c % 16 remainder channels -- whose four partial sums are
  // combined in another order: same value to fp32 rounding, not bit-identical across c % 16): four channels of the wave at a time,
  // all sixteen loads issued before the first butterfly -- one channel at a time the loop was a chain of L2 round trips
  // (57 us of the 1.2 ms online step for a single clip's 512 x 196 volume, round 3)
  int ch = wave;
  if (t == 1 && s > kWave && s <= 4 * kWave) {
    for (; ch + 3 * kWaves < c; ch += 4 * kWaves) {
      float q[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* xp = xb + (long)(ch + u * kWaves) * s;
#pragma unroll
        for (int v = 0; v < 4; ++v) q[u][v] = lane + v * kWave < s ? ld(xp + lane + v * kWave) : 0.0f;
      }
      float a4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] = (q[u][0] + q[u][1]) + (q[u][2] + q[u][3]);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += shfl_xor(a4[u], m);
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pooled[ch + u * kWaves] = a4[u] * inv;
      }
    }
  }
  for (; ch < c; ch += kWaves) {
    const float* xp = xb + (long)ch * s;
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
    if (s <= kWave) {          // 2-D stream: one load per frame (7x7 planes), eight frames in flight
      int f = 0;
      for (; f + 7 < t; f += 8) {
        const float* row = xp + f * fstride + lane;
        if (lane < s) {
          float q[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) q[u] = ld(row + u * fstride);
          p0 += q[0] + q[4]; p1 += q[1] + q[5]; p2 += q[2] + q[6]; p3 += q[3] + q[7];
        }
      }
      for (; f < t; ++f)
        if (lane < s) p0 += ld(xp + f * fstride + lane);
    } else {
      for (int f = 0; f < t; ++f) {
        const float* row = xp + f * fstride;
        int r = lane;
        for (; r + 3 * kWave < s; r += 4 * kWave) {
          p0 += ld(row + r); p1 += ld(row + r + kWave); p2 += ld(row + r + 2 * kWave); p3 += ld(row + r + 3 * kWave);
        }
        for (; r < s; r += kWave) p0 += ld(row + r);
      }
    }
    const float a0 = wave_sum((p0 + p1) + (p2 + p3));
    if (lane == 0) pooled[ch] = a0 * inv;
  }
  __syncthreads();
  const int o_begin = (int)blockIdx.x * out_per_block;
  const int o_end = min(o_begin + out_per_block, n_out);
  for (int o = o_begin + wave; o < o_end; o += kWaves) {
    const float* wr = w + (long)o * wk + c0;
    float acc = 0.0f, acc1 = 0.0f;
    int i = lane;
    for (; i + 7 * kWave < c; i += 8 * kWave) {   // eight weight loads in flight
      float wq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wq[u] = ld(wr + i + u * kWave);
#pragma unroll
      for (int u = 0; u < 8; u += 2) { acc += pooled[i + u * kWave] * wq[u]; acc1 += pooled[i + (u + 1) * kWave] * wq[u + 1]; }
    }
    for (; i < c; i += kWave) acc += pooled[i] * ld(wr + i);
    acc = wave_sum(acc + acc1);
    if (lane == 0) {
      float* yp = y + (long)b * n_out + o;
      float v = acc + (bias ? ld(bias + o) : 0.0f);
      if (accumulate) v += ld((const float*)yp);
      st(yp, v);
    }
  }
}

// VideoData output contract on the GPU: uint8 HWC (interleaved BGR) -> cropped, mean-subtracted, scaled
// planar fp32.  One thread per 4 consecutive output pixels of a row (12 contiguous input bytes, 4 floats
// into each of the three colour planes).  HBM-bound: 3 B read + 12 B written per pixel.
__global__ __launch_bounds__(256) void video_input_kernel(const uint8_t* frames, float* y, long num_frames, int H, int W,
                                                          int ch, int cw, int h_off, int w_off, float m0, float m1,
                                                          float m2, float scale, int mirror) {
  const int wq = (cw + 3) / 4;
  const long total = num_frames * ch * wq;
  const float mean[3] = {m0, m1, m2};
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int q = (int)(i % wq);
    const long t = i / wq;
    const int h = (int)(t % ch);
    const long f = t / ch;
    const uint8_t* src = frames + ((f * H + (h_off + h)) * (long)W + w_off) * 3;
    float* dst = y + f * 3 * ch * cw + (long)h * cw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int w = 4 * q + e;  // output column
      if (w >= cw) break;
      const int ws = mirror ? (cw - 1 - w) : w;  // source column inside the crop (data_transformer.cpp:272-276)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        st(dst + (long)c * ch * cw + w, ((float)ld(src + (long)ws * 3 + c) - mean[c]) * scale);
    }
  }
}

// Softmax over axis 1 of [outer, c, inner]: one thread per (outer, inner) column.
__global__ __launch_bounds__(256) void softmax_kernel(const float* x, float* y, long outer, long c, long inner) {
  const long total = outer * inner;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long o = i / inner, in = i - o * inner;
    const float* xp = x + o * c * inner + in;
    float* yp = y + o * c * inner + in;
    float m = -FLT_MAX;
    for (long j = 0; j < c; ++j) m = fmaxf(m, ld(xp + j * inner));
    float sum = 0.0f;
    for (long j = 0; j < c; ++j) sum += expf(ld(xp + j * inner) - m);
    for (long j = 0; j < c; ++j) st(yp + j * inner, expf(ld(xp + j * inner) - m) / sum);
  }
}

// AccuracyLayer::Forward (layers/accuracy_layer.cpp:46-92) over [outer, c, inner] scores and outer*inner
// labels: a sample is a hit iff its label is among the first top_k entries of the scores sorted by
// std::greater<pair<score, class>> -- i.e. fewer than top_k classes have a larger score, or the same score
// and a larger class index.  One wave per sample; out[0] = hits / counted (labels equal to ignore_label are
// skipped when has_ignore).  Single workgroup: the evaluator tail handles a clip batch.
__global__ __launch_bounds__(256) void accuracy_kernel(const float* x, const float* label, float* out, long outer,
                                                       long c, long inner, int top_k, int has_ignore, int ignore_label) {
  __shared__ int s_hits[4], s_cnt[4];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  int hits = 0, cnt = 0;
  const long total = outer * inner;
  for (long i = wave; i < total; i += 4) {
    const long o = i / inner, in = i - o * inner;
    const int lv = (int)ld(label + i);
    if (has_ignore && lv == ignore_label) continue;
    const float* xp = x + o * c * inner + in;
    const float vl = ld(xp + (long)lv * inner);
    float before = 0.0f;  // classes ranked ahead of the label (exact in fp32: c < 2^24)
    for (long j = lane; j < c; j += 64) {
      const float v = ld(xp + j * inner);
      before += (v > vl || (v == vl && j > lv)) ? 1.0f : 0.0f;
    }
    before = wave_sum(before);
    hits += before < (float)top_k ? 1 : 0;
    ++cnt;
  }
  if (lane == 0) { s_hits[wave] = hits; s_cnt[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int h = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3], n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    st(out, (float)h / (float)n);
  }
}

// SoftmaxWithLossLayer::Forward (layers/softmax_loss_layer.cpp:52-84): loss = -sum log(max(prob[label],
// FLT_MIN)) over the counted samples / (normalize ? counted : outer).  One wave per sample (max, sum of
// exponentials and the label's exponential by butterfly reductions), partial sums combined in wave order.
__global__ __launch_bounds__(256) void softmax_loss_kernel(const float* x, const float* label, float* out, long outer,
                                                           long c, long inner, int normalize, int has_ignore,
                                                           int ignore_label) {
  __shared__ float s_loss[4];
  __shared__ int s_cnt[4];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  float loss = 0.0f;
  int cnt = 0;
  const long total = outer * inner;
  for (long i = wave; i < total; i += 4) {
    const long o = i / inner, in = i - o * inner;
    const int lv = (int)ld(label + i);
    if (has_ignore && lv == ignore_label) continue;
    const float* xp = x + o * c * inner + in;
    float m = -FLT_MAX;
    for (long j = lane; j < c; j += 64) m = fmaxf(m, ld(xp + j * inner));
    m = wave_max(m);
    float sum = 0.0f;
    for (long j = lane; j < c; j += 64) sum += expf(ld(xp + j * inner) - m);
    sum = wave_sum(sum);
    const float prob = expf(ld(xp + (long)lv * inner) - m) / sum;
    loss -= logf(fmaxf(prob, FLT_MIN));
    ++cnt;
  }
  if (lane == 0) { s_loss[wave] = loss; s_cnt[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = ((s_loss[0] + s_loss[1]) + s_loss[2]) + s_loss[3];
    const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    st(out, l / (normalize ? (float)n : (float)outer));
  }
}

}  // namespace eco

using namespace eco;

extern "C" int eco_pool_forward_strided(const eco_pool_geom* g, const float* x, float* y, int64_t y_image_stride, void* stream) {
  clear_error();
  ECO_REQUIRE(g && x && y, "pool: null argument");
  ECO_REQUIRE(g->n > 0 && g->c > 0, "pool: n and c must be positive");
  ECO_REQUIRE(g->method == ECO_POOL_MAX || g->method == ECO_POOL_AVE, "pool: unknown pooling method %d", g->method);
  bool global = true;
  for (int i = 0; i < 3; ++i) {
    ECO_REQUIRE(g->in[i] > 0 && g->kernel[i] > 0 && g->stride[i] > 0 && g->pad[i] >= 0 && g->out[i] > 0,
                "pool: bad geometry on axis %d", i);
    ECO_REQUIRE(g->pad[i] < g->kernel[i], "pool: pad must be smaller than kernel (axis %d)", i);
    // ceil rule + last-window clip (pooling_layer.cpp:131-147); float division as the reference does
    int o = (int)ceilf((float)(g->in[i] + 2 * g->pad[i] - g->kernel[i]) / (float)g->stride[i]) + 1;
    if (g->pad[i] && (o - 1) * g->stride[i] >= g->in[i] + g->pad[i]) --o;
    ECO_REQUIRE(o == g->out[i], "pool: output dim %d is %d, expected %d", i, g->out[i], o);
    global = global && g->kernel[i] == g->in[i] && g->pad[i] == 0 && g->out[i] == 1;
  }
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)g->n * g->c;
  const long s_in = (long)g->in[0] * g->in[1] * g->in[2];
  // y as a channel slice of a wider tensor (a Concat top: concat_layer.cpp:60-81 copies the bottoms there; here the pooling
  // writes its slice itself): images y_image_stride floats apart, the c channels of an image contiguous
  const long s_out = (long)g->out[0] * g->out[1] * g->out[2];
  ECO_REQUIRE(y_image_stride == 0 || y_image_stride >= (int64_t)g->c * s_out, "pool: y_image_stride %ld is smaller than an image (%ld)",
              (long)y_image_stride, (long)g->c * s_out);
  const long y_extra = y_image_stride ? (long)y_image_stride - (long)g->c * s_out : 0l;
  if (y_extra == 0 && global && g->method == ECO_POOL_AVE && s_in >= 32 && s_in < 2147483647l) {
    hipLaunchKernelGGL((global_avg_kernel), dim3(grid_for(rows, 4)), dim3(kThreads), 0, s, x, y, rows, (int)s_in);
    return check_launch("eco_pool_forward(global)");
  }
  const bool two_d = g->in[0] == 1 && g->kernel[0] == 1 && g->stride[0] == 1 && g->pad[0] == 0;
  const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && y_extra % 4 == 0;
  if (two_d && aligned && g->method == ECO_POOL_MAX && g->kernel[1] == 3 && g->kernel[2] == 3 && g->stride[1] == 2 &&
      g->stride[2] == 2 && g->pad[1] == 0 && g->pad[2] == 0 && g->in[2] % 4 == 0 && g->out[2] % 2 == 0 &&
      2 * (g->out[2] - 1) + 2 <= g->in[2]) {
    // rows are 16-byte aligned (Wi % 4 == 0); 4 outputs per thread when Wo % 4 == 0, else 2 (e.g. 28 -> 14)
    if (g->out[2] % 4 == 0) {
      const long total = rows * g->out[1] * (g->out[2] / 4);
      hipLaunchKernelGGL((maxpool2d_k3s2_kernel<4>), dim3(grid_for(total)), dim3(kThreads), 0, s, x, y, rows, g->in[1],
                         g->in[2], g->out[1], g->out[2], g->c, y_extra);
    } else {
      const long total = rows * g->out[1] * (g->out[2] / 2);
      hipLaunchKernelGGL((maxpool2d_k3s2_kernel<2>), dim3(grid_for(total)), dim3(kThreads), 0, s, x, y, rows, g->in[1],
                         g->in[2], g->out[1], g->out[2], g->c, y_extra);
    }
    return check_launch("eco_pool_forward(max 3x3 s2)");
  }
  if (y_extra == 0 && two_d && aligned && g->method == ECO_POOL_AVE && g->kernel[1] == 3 && g->kernel[2] == 3 && g->stride[1] == 1 &&
      g->stride[2] == 1 && g->pad[1] == 1 && g->pad[2] == 1 && g->in[2] % 2 == 0 && g->in[1] >= 2 && g->in[2] >= 4) {
    const long hq = (g->in[1] + kAvgRows - 1) / kAvgRows;
    if (g->in[2] % 4 == 0) {
      const long total = rows * hq * (g->in[2] / 4);
      hipLaunchKernelGGL((avgpool2d_k3s1p1_kernel<4>), dim3(grid_for(total)), dim3(kThreads), 0, s, x, y, rows,
                         g->in[1], g->in[2]);
    } else {  // e.g. 14x14: 8-byte accesses
      const long total = rows * hq * (g->in[2] / 2);
      hipLaunchKernelGGL((avgpool2d_k3s1p1_kernel<2>), dim3(grid_for(total)), dim3(kThreads), 0, s, x, y, rows,
                         g->in[1], g->in[2]);
    }
    return check_launch("eco_pool_forward(ave 3x3 s1 p1)");
  }
  PoolArgs a;
  a.x = x; a.y = y;
  a.Di = g->in[0]; a.Hi = g->in[1]; a.Wi = g->in[2];
  a.Do = g->out[0]; a.Ho = g->out[1]; a.Wo = g->out[2];
  a.kd = g->kernel[0]; a.kh = g->kernel[1]; a.kw = g->kernel[2];
  a.sd = g->stride[0]; a.sh = g->stride[1]; a.sw = g->stride[2];
  a.pd = g->pad[0]; a.ph = g->pad[1]; a.pw = g->pad[2];
  a.method = g->method;
  a.c = g->c; a.y_extra = y_extra;
  a.total = rows * a.Do * a.Ho * a.Wo;
  if (two_d && a.kh == 3 && a.kw == 3) {
    if (a.method == ECO_POOL_MAX) hipLaunchKernelGGL((pool2d_k3_kernel<ECO_POOL_MAX>), dim3(grid_for(a.total)), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((pool2d_k3_kernel<ECO_POOL_AVE>), dim3(grid_for(a.total)), dim3(kThreads), 0, s, a);
    return check_launch("eco_pool_forward(3x3)");
  }
  hipLaunchKernelGGL((pool_kernel), dim3(grid_for(a.total)), dim3(kThreads), 0, s, a);
  return check_launch("eco_pool_forward");
}

extern "C" int eco_pool_forward(const eco_pool_geom* g, const float* x, float* y, void* stream) {
  return eco_pool_forward_strided(g, x, y, 0, stream);
}

extern "C" int eco_avgpool_affine_forward(const float* x, const float* bias, const float* bn_scale, const float* bn_shift,
                                          int32_t relu, const eco_view* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                                          void* stream) {
  clear_error();
  ECO_REQUIRE(x && dst && dst->ptr && n > 0 && c > 0 && h > 0 && w > 0, "avgpool_affine: bad argument");
  ECO_REQUIRE(!bn_scale == !bn_shift, "avgpool_affine: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(dst->t >= 1 && dst->stride_c >= 1, "avgpool_affine: view needs t >= 1 and stride_c >= 1");
  int vec = 4;
  while (vec > 1 && (w % vec || ((uintptr_t)x % (4 * vec)) || ((uintptr_t)dst->ptr % (4 * vec)) || dst->stride_b % vec ||
                     dst->stride_t % vec || dst->stride_c % vec))
    vec /= 2;
  const long planes = (long)n * c;
  const long hq = (h + kAvgRows - 1) / kAvgRows;
  const long total = planes * hq * (w / vec);
  const float floor_v = relu ? 0.0f : -FLT_MAX;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(grid_for(total)), block(kThreads);
  if (vec == 4) hipLaunchKernelGGL((avgpool2d_k3s1p1_affine_kernel<4>), grid, block, 0, s, x, bias, bn_scale, bn_shift, floor_v, *dst, planes, c, h, w);
  else if (vec == 2) hipLaunchKernelGGL((avgpool2d_k3s1p1_affine_kernel<2>), grid, block, 0, s, x, bias, bn_scale, bn_shift, floor_v, *dst, planes, c, h, w);
  else hipLaunchKernelGGL((avgpool2d_k3s1p1_affine_kernel<1>), grid, block, 0, s, x, bias, bn_scale, bn_shift, floor_v, *dst, planes, c, h, w);
  return check_launch("eco_avgpool_affine_forward");
}

extern "C" int eco_bn_forward(const float* x, float* y, const float* scale, const float* shift, int64_t n, int64_t c,
                              int64_t inner, int relu, void* stream) {
  clear_error();
  ECO_REQUIRE(x && y && scale && shift, "bn: null argument");
  ECO_REQUIRE(n > 0 && c > 0 && inner > 0, "bn: bad shape");
  const long total = n * c * inner;
  hipLaunchKernelGGL((bn_kernel), dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, scale, shift,
                     total, (long)c, (long)inner, relu);
  return check_launch("eco_bn_forward");
}

extern "C" int eco_relu_forward(const float* x, float* y, int64_t count, float negative_slope, void* stream) {
  clear_error();
  ECO_REQUIRE(x && y && count > 0, "relu: bad argument");
  hipLaunchKernelGGL((relu_kernel), dim3(grid_for(count)), dim3(kThreads), 0, (hipStream_t)stream, x, y, (long)count,
                     negative_slope);
  return check_launch("eco_relu_forward");
}

extern "C" int eco_eltwise_sum_forward(const float* a, const float* b, float* y, int64_t count, float ca, float cb,
                                       void* stream) {
  clear_error();
  ECO_REQUIRE(a && b && y && count > 0, "eltwise: bad argument");
  hipLaunchKernelGGL((eltwise_sum_kernel), dim3(grid_for(count)), dim3(kThreads), 0, (hipStream_t)stream, a, b, y,
                     (long)count, ca, cb);
  return check_launch("eco_eltwise_sum_forward");
}

extern "C" int eco_concat_copy(const float* x, float* y, int64_t outer, int64_t cx, int64_t cy, int64_t c0,
                               int64_t inner, void* stream) {
  clear_error();
  ECO_REQUIRE(x && y, "concat: null argument");
  ECO_REQUIRE(outer > 0 && cx > 0 && inner > 0 && c0 >= 0 && c0 + cx <= cy, "concat: slice [%ld,%ld) outside %ld channels",
              (long)c0, (long)(c0 + cx), (long)cy);
  const long total = outer * cx * inner;
  hipLaunchKernelGGL((concat_copy_kernel), dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, total,
                     (long)cx, (long)cy, (long)c0, (long)inner);
  return check_launch("eco_concat_copy");
}

extern "C" int eco_permute_forward(const float* x, float* y, int32_t naxes, const int32_t* in_shape,
                                   const int32_t* order, void* stream) {
  clear_error();
  ECO_REQUIRE(x && y && in_shape && order, "permute: null argument");
  ECO_REQUIRE(naxes >= 1 && naxes <= 6, "permute: %d axes unsupported (1..6)", naxes);
  long in_stride[6];
  long s = 1;
  for (int i = naxes - 1; i >= 0; --i) {
    ECO_REQUIRE(in_shape[i] > 0, "permute: bad shape");
    in_stride[i] = s;
    s *= in_shape[i];
  }
  PermuteArgs a;
  a.x = x; a.y = y; a.total = s; a.naxes = naxes;
  unsigned seen = 0;
  for (int i = 0; i < naxes; ++i) {
    ECO_REQUIRE(order[i] >= 0 && order[i] < naxes && !(seen & (1u << order[i])), "permute: there are duplicate orders");
    seen |= 1u << order[i];
    a.out_shape[i] = in_shape[order[i]];
    a.in_stride_of_out_axis[i] = in_stride[order[i]];
  }
  hipLaunchKernelGGL((permute_kernel), dim3(grid_for(a.total)), dim3(kThreads), 0, (hipStream_t)stream, a);
  return check_launch("eco_permute_forward");
}

extern "C" int eco_inner_product_forward(const float* x, const float* w, const float* bias, float* y, int64_t m,
                                         int64_t n, int64_t k, void* stream) {
  clear_error();
  ECO_REQUIRE(x && w && y, "inner_product: null argument");
  ECO_REQUIRE(m > 0 && n > 0 && k > 0, "inner_product: bad shape");
  hipLaunchKernelGGL((inner_product_kernel), dim3(grid_for(m * n, 4)), dim3(kThreads), 0, (hipStream_t)stream, x, w,
                     bias, y, (long)m, (long)n, (long)k);
  return check_launch("eco_inner_product_forward");
}

extern "C" int eco_global_avgpool_fc_forward(const float* x, const float* w, const float* bias, float* y, int64_t b,
                                             int64_t c, int64_t s, int64_t n_out, int64_t wk, int64_t c0,
                                             int accumulate, void* stream) {
  return eco_global_avgpool_fc_seg_forward(x, w, bias, y, b, 1, c, s, n_out, wk, c0, accumulate, stream);
}

extern "C" int eco_global_avgpool_fc_seg_forward(const float* x, const float* w, const float* bias, float* y, int64_t b,
                                                 int64_t t, int64_t c, int64_t s, int64_t n_out, int64_t wk, int64_t c0,
                                                 int accumulate, void* stream) {
  clear_error();
  ECO_REQUIRE(x && w && y, "global_avgpool_fc: null argument");
  ECO_REQUIRE(b > 0 && c > 0 && s > 0 && n_out > 0 && t > 0 && s * t < 2147483647l, "global_avgpool_fc: bad shape");
  ECO_REQUIRE(c <= kTailMaxC, "global_avgpool_fc: %ld channels exceed the %d-channel LDS buffer", (long)c, kTailMaxC);
  ECO_REQUIRE(c0 >= 0 && c0 + c <= wk, "global_avgpool_fc: weight columns [%ld,%ld) outside row length %ld", (long)c0,
              (long)(c0 + c), (long)wk);
  ECO_REQUIRE(b <= 65535 && s < 2147483647l, "global_avgpool_fc: batch too large for one launch");
  const bool big = c * s * t > 262144;                              // > 1 MB of fp32 per clip
  long blocks = ceil_div(big ? kTailWorkgroupsBigVolume : kTailWorkgroupsSmallVolume, b);   // logit blocks per clip
  if (blocks > ceil_div(n_out, 16)) blocks = ceil_div(n_out, 16);   // at least one logit per wave
  const int out_per_block = (int)(ceil_div(ceil_div(n_out, blocks), 16) * 16);
  dim3 grid((unsigned)ceil_div(n_out, out_per_block), (unsigned)b);
  hipLaunchKernelGGL((global_avgpool_fc_kernel), grid, dim3(kTailThreads), 0, (hipStream_t)stream, x, w, bias, y, (int)c,
                     (int)s, (int)t, (int)n_out, (int)wk, (int)c0, accumulate, out_per_block);
  return check_launch("eco_global_avgpool_fc_forward");
}

extern "C" int eco_video_input_forward(const uint8_t* frames, float* y, int64_t num_frames, int32_t height,
                                       int32_t width, int32_t crop_h, int32_t crop_w, int32_t h_off, int32_t w_off,
                                       const float mean[3], float scale, int32_t mirror, void* stream) {
  clear_error();
  ECO_REQUIRE(frames && y && mean, "video_input: null argument");
  ECO_REQUIRE(num_frames > 0 && height > 0 && width > 0 && crop_h > 0 && crop_w > 0, "video_input: bad shape");
  // DataTransformer CHECK_GE(datum_height, crop_size) / CHECK_GE(datum_width, crop_size)
  ECO_REQUIRE(h_off >= 0 && w_off >= 0 && h_off + crop_h <= height && w_off + crop_w <= width,
              "video_input: crop %dx%d at (%d,%d) does not fit the %dx%d frame", crop_h, crop_w, h_off, w_off, height, width);
  const long total = (long)num_frames * crop_h * ((crop_w + 3) / 4);
  hipLaunchKernelGGL((video_input_kernel), dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, frames, y,
                     (long)num_frames, height, width, crop_h, crop_w, h_off, w_off, mean[0], mean[1], mean[2], scale,
                     mirror);
  return check_launch("eco_video_input_forward");
}

extern "C" int eco_softmax_forward(const float* x, float* y, int64_t outer, int64_t c, int64_t inner, void* stream) {
  clear_error();
  ECO_REQUIRE(x && y && outer > 0 && c > 0 && inner > 0, "softmax: bad argument");
  hipLaunchKernelGGL((softmax_kernel), dim3(grid_for(outer * inner)), dim3(kThreads), 0, (hipStream_t)stream, x, y,
                     (long)outer, (long)c, (long)inner);
  return check_launch("eco_softmax_forward");
}

extern "C" int eco_accuracy_forward(const float* x, const float* label, float* out, int64_t outer, int64_t c,
                                    int64_t inner, int32_t top_k, int32_t has_ignore_label, int32_t ignore_label,
                                    void* stream) {
  clear_error();
  ECO_REQUIRE(x && label && out && outer > 0 && c > 0 && inner > 0, "accuracy: bad argument");
  ECO_REQUIRE(top_k >= 1 && top_k <= c, "accuracy: top_k must be in [1, %ld] (got %d)", (long)c, top_k);
  hipLaunchKernelGGL((accuracy_kernel), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, x, label, out, (long)outer,
                     (long)c, (long)inner, top_k, has_ignore_label, ignore_label);
  return check_launch("eco_accuracy_forward");
}

extern "C" int eco_softmax_loss_forward(const float* x, const float* label, float* out, int64_t outer, int64_t c,
                                        int64_t inner, int32_t normalize, int32_t has_ignore_label,
                                        int32_t ignore_label, void* stream) {
  clear_error();
  ECO_REQUIRE(x && label && out && outer > 0 && c > 0 && inner > 0, "softmax loss: bad argument");
  hipLaunchKernelGGL((softmax_loss_kernel), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, x, label, out,
                     (long)outer, (long)c, (long)inner, normalize, has_ignore_label, ignore_label);
  return check_launch("eco_softmax_loss_forward");
}
