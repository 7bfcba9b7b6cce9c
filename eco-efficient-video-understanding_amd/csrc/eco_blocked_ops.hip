// eco_blocked_ops.hip -- the memory-bound operators of the channel-blocked bf16 path (BASELINE.json configs[4]): pooling, the
// exchanged AVE pool + affine of the inception blocks, and the global_pool -> reshape -> dropout -> fc tail, on
// X[n][c/8][d][h][w][c%8] tensors (one thread = the eight channels of one position = one 16-byte vector; csrc/eco_blocked.hip
// has the layout's rationale and the convolutions).  Replaces, for this storage layout, PoolingLayer::Forward
// (pooling_layer.cpp:131-147,199-262; cudnn_pooling_layer.cu:13-22) and the pool + InnerProduct tail
// (inner_product_layer.cu:14-25).  HBM-bound, nothing reshaped into GEMMs.  (Split out of eco_blocked.hip in round 6.)
#include <float.h>
#include <string.h>

#include "eco_blocked.h"

namespace eco {

// ------------------------------------------------------------------------------------------------------------------
// Pooling on blocked tensors: one thread per (image, block, od, oh, ow) = eight channels of one output position,
// every window element one 16-byte (bf16) / two 16-byte (fp32) loads; consecutive lanes take consecutive output
// positions.  Window rules as pool_kernel in eco_ops.hip (Caffe ceil rule; MAX clips to the image, AVE divides by
// the window size including padding clipped to in+pad).  Overlapping windows re-read through L1/L2.
struct PoolBArgs {
  const void* x;
  void* y;
  long total;  // n * cblocks * Do*Ho*Wo
  int Di, Hi, Wi, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int method;
};

__global__ __launch_bounds__(256) void poolb_kernel(const PoolBArgs a) {
  const long s_in = (long)a.Di * a.Hi * a.Wi;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long)gridDim.x * 256) {
    const int ow = (int)(i % a.Wo);
    long t = i / a.Wo;
    const int oh = (int)(t % a.Ho);
    t /= a.Ho;
    const int od = (int)(t % a.Do);
    const long ncb = t / a.Do;
    const long xb = ncb * s_in;
    int ds = od * a.sd - a.pd, hs = oh * a.sh - a.ph, ws = ow * a.sw - a.pw;
    float r[8];
    if (a.method == ECO_POOL_MAX) {
      const int de = min(ds + a.kd, a.Di), he = min(hs + a.kh, a.Hi), we = min(ws + a.kw, a.Wi);
      ds = max(ds, 0); hs = max(hs, 0); ws = max(ws, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = -FLT_MAX;
      for (int d = ds; d < de; ++d)
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) {
            float f[8];
            block_to_f32(load_block(a.x, xb + ((long)d * a.Hi + h) * a.Wi + w), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], f[e]);
          }
    } else {
      int de = min(ds + a.kd, a.Di + a.pd), he = min(hs + a.kh, a.Hi + a.ph), we = min(ws + a.kw, a.Wi + a.pw);
      const float size = (float)((de - ds) * (he - hs) * (we - ws));
      ds = max(ds, 0); hs = max(hs, 0); ws = max(ws, 0);
      de = min(de, a.Di); he = min(he, a.Hi); we = min(we, a.Wi);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = 0.0f;
      for (int d = ds; d < de; ++d)
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) {
            float f[8];
            block_to_f32(load_block(a.x, xb + ((long)d * a.Hi + h) * a.Wi + w), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] += f[e];
          }
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] /= size;
    }
    const float lo[4] = {r[0], r[1], r[2], r[3]}, hi[4] = {r[4], r[5], r[6], r[7]};
    store_quad(a.y, i, 0, lo);
    store_quad(a.y, i, 1, hi);
  }
}

// 2-D 3x3 windows (every pooling layer of the BN-Inception head: pool1 / pool2 MAX 3x3 s2, inception_3x_pool AVE 3x3 s1
// p1): the nine block loads of an output are independent and all in flight before the first is used (the generic
// kernel's runtime-bounded loops fetch them one round trip at a time: 3.5 TB/s).  Same arithmetic, same order.
template <int METHOD>
__global__ __launch_bounds__(256) void poolb_k3_kernel(const PoolBArgs a) {
  const long s_in = (long)a.Hi * a.Wi;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long)gridDim.x * 256) {
    const int ow = (int)(i % a.Wo);
    const long t = i / a.Wo;
    const int oh = (int)(t % a.Ho);
    const long ncb = t / a.Ho;
    const long xb = ncb * s_in;
    const int hs = oh * a.sh - a.ph, ws = ow * a.sw - a.pw;
    uint4 v[3][3];
    bool ok[3][3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int h = hs + dh, w = ws + dw;
        ok[dh][dw] = (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
        v[dh][dw] = load_block(a.x, xb + (ok[dh][dw] ? (long)h * a.Wi + w : 0l));
      }
    if constexpr (METHOD == ECO_POOL_MAX) {
      // bf16 MAX without leaving bf16: x -> x ^ ((x >> 15) & 0x7fff) (arithmetic shift per 16-bit half) maps the
      // sign-magnitude patterns onto two's-complement order and is its own inverse, so the window is eight packed signed
      // 16-bit maxima per dword (v_pk_max_i16) -- five VALU instructions per loaded dword where unpacking two values to
      // fp32, two v_max and the in-image selects took eight, and no conversion back.  The maximum of bf16 values is one of
      // them: bit-identical to the fp32 route.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      auto key = [](unsigned x) {
        const s16x2 q = __builtin_bit_cast(s16x2, x);
        return __builtin_bit_cast(s16x2, x ^ (__builtin_bit_cast(unsigned, q >> 15) & 0x7fff7fffu));
      };
      const s16x2 lowest = {(short)-32768, (short)-32768};
      s16x2 m[4] = {lowest, lowest, lowest, lowest};
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const unsigned q[4] = {v[dh][dw].x, v[dh][dw].y, v[dh][dw].z, v[dh][dw].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = __builtin_elementwise_max(m[e], ok[dh][dw] ? key(q[e]) : lowest);
        }
      unsigned o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(unsigned, key(__builtin_bit_cast(unsigned, m[e])));
      st((uint4*)a.y + i, make_uint4(o[0], o[1], o[2], o[3]));
      continue;
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = METHOD == ECO_POOL_MAX ? -FLT_MAX : 0.0f;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        float f[8];
        block_to_f32(v[dh][dw], f);
        if (ok[dh][dw]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = METHOD == ECO_POOL_MAX ? fmaxf(r[e], f[e]) : r[e] + f[e];
        }
      }
    if (METHOD != ECO_POOL_MAX) {   // divisor: the window clipped to the padded image (pooling_layer.cpp:240-262)
      const int he = min(hs + 3, a.Hi + a.ph), we = min(ws + 3, a.Wi + a.pw);
      const float size = (float)((he - hs) * (we - ws));
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] /= size;
    }
    const float lo[4] = {r[0], r[1], r[2], r[3]}, hi[4] = {r[4], r[5], r[6], r[7]};
    store_quad(a.y, i, 0, lo);
    store_quad(a.y, i, 1, hi);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// global_pool (AVE over the whole volume) -> reshape -> dropout(TEST) -> fc on a blocked volume x[b][c/8][s][8]:
// grid = (ceil(n_out/128), b), 1024 threads.  Pooling: one wave per channel block, lanes stride over the s
// positions accumulating 8 channels each, 64-lane butterfly per channel; then one wave per logit as in
// global_avgpool_fc_kernel.  fp32 weights / bias / logits whatever the storage type.
constexpr int kTailBThreads = 1024;
constexpr int kTailBMaxC = 2048;
constexpr int kTailBOut = 128;

// The AVE 3x3 / stride 1 / pad 1 pool that runs BEHIND its 1x1 projection (the engine's pool_commute pre-pass, as
// avgpool2d_k3s1p1_affine_kernel in eco_ops.hip does for the fp32 path): z = conv1x1(x) without bias -> window sum / 9
// (the divisor counts the padding: pooling_layer.cpp:247-262) + bias, folded BN, ReLU, into a blocked view (a Concat
// slice).  One thread per (image, 8-channel block, position); the nine block loads are issued before the first is used.
struct PoolBAffArgs {
  const void* x;
  const float* bias;
  const float* scale;
  const float* shift;
  eco_view dst;
  long total;   // n * cblocks * H * W
  int CB, H, W;
  float floor_v;
};
__global__ __launch_bounds__(256) void poolb_avg_affine_kernel(const PoolBAffArgs a) {
  const long plane = (long)a.H * a.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % a.W);
    const long t = i / a.W;
    const int h = (int)(t % a.H);
    const long ncb = t / a.H;
    const int cb = (int)(ncb % a.CB), img = (int)(ncb / a.CB);
    const long xb = ncb * plane;
    uint4 v[3][3];
    bool ok[3][3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int hh = h - 1 + dh, ww = w - 1 + dw;
        ok[dh][dw] = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        v[dh][dw] = load_block(a.x, xb + (ok[dh][dw] ? (long)hh * a.W + ww : 0l));
      }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = 0.0f;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        float f[8];
        block_to_f32(v[dh][dw], f);
        if (ok[dh][dw]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] += f[e];
        }
      }
    const float inv = 1.0f / 9.0f;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cb * 8 + e;
      const float b = a.bias ? ld(a.bias + ch) : 0.0f;
      const float sc = a.scale ? ld(a.scale + ch) : 1.0f, sh = a.scale ? ld(a.shift + ch) : 0.0f;
      y[e] = fmaxf((r[e] * inv + b) * sc + sh, a.floor_v);
    }
    const long o = view_base(a.dst, img, h * a.W + w) + (long)cb * a.dst.stride_c;
    const float lo[4] = {y[0], y[1], y[2], y[3]}, hi[4] = {y[4], y[5], y[6], y[7]};
    store_quad(a.dst.ptr, o, 0, lo);
    store_quad(a.dst.ptr, o, 1, hi);
  }
}

__global__ __launch_bounds__(1024) void global_avgpool_fc_b_kernel(const void* x, const float* w, const float* bias,
                                                                   float* y, int c, int s, int n_out, int wk, int c0,
                                                                   int accumulate) {
  __shared__ float pooled[kTailBMaxC];
  constexpr int kWaves = kTailBThreads / kWave;
  const int lane = lane_id();
  const int wave = uniform((int)(threadIdx.x >> 6));
  const int b = (int)blockIdx.y;
  const int cblocks = c / 8;
  const float inv = 1.0f / (float)s;
  for (int cb = wave; cb < cblocks; cb += kWaves) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    const long base = ((long)b * cblocks + cb) * s;
    for (int i = lane; i < s; i += kWave) {
      float f[8];
      block_to_f32(load_block(x, base + i), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = wave_sum(acc[e]);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) pooled[cb * 8 + e] = acc[e] * inv;
    }
  }
  __syncthreads();
  const int o_begin = (int)blockIdx.x * kTailBOut;
  const int o_end = min(o_begin + kTailBOut, n_out);
  for (int o = o_begin + wave; o < o_end; o += kWaves) {
    const float* wr = w + (long)o * wk + c0;
    float acc = 0.0f;
    for (int i = lane; i < c; i += kWave) acc += pooled[i] * ld(wr + i);
    acc = wave_sum(acc);
    if (lane == 0) {
      float* yp = y + (long)b * n_out + o;
      float v = acc + (bias ? ld(bias + o) : 0.0f);
      if (accumulate) v += ld((const float*)yp);
      st(yp, v);
    }
  }
}

}  // namespace eco

using namespace eco;

extern "C" int eco_poolb_forward(const eco_pool_geom* g, int32_t dt, const void* x, void* y, void* stream) {
  clear_error();
  ECO_REQUIRE(g && x && y, "poolb: null argument");
  const int ns = dt == ECO_DT_BF16 ? 1 : 0;
  ECO_REQUIRE(ns != 0, "poolb: storage type must be ECO_DT_BF16");
  ECO_REQUIRE(g->n > 0 && g->c > 0 && g->c % 8 == 0, "poolb: channels (%d) must be a positive multiple of 8", g->c);
  ECO_REQUIRE(g->method == ECO_POOL_MAX || g->method == ECO_POOL_AVE, "poolb: unknown pooling method %d", g->method);
  for (int i = 0; i < 3; ++i) {
    ECO_REQUIRE(g->in[i] > 0 && g->kernel[i] > 0 && g->stride[i] > 0 && g->pad[i] >= 0 && g->out[i] > 0,
                "poolb: bad geometry (axis %d)", i);
    ECO_REQUIRE(g->pad[i] < g->kernel[i], "poolb: pad must be smaller than kernel (axis %d)", i);
    ECO_REQUIRE((g->out[i] - 1) * g->stride[i] < g->in[i] + g->pad[i], "poolb: last window starts outside the padded input");
  }
  PoolBArgs a;
  a.x = x; a.y = y;
  a.Di = g->in[0]; a.Hi = g->in[1]; a.Wi = g->in[2];
  a.Do = g->out[0]; a.Ho = g->out[1]; a.Wo = g->out[2];
  a.kd = g->kernel[0]; a.kh = g->kernel[1]; a.kw = g->kernel[2];
  a.sd = g->stride[0]; a.sh = g->stride[1]; a.sw = g->stride[2];
  a.pd = g->pad[0]; a.ph = g->pad[1]; a.pw = g->pad[2];
  a.method = g->method;
  a.total = (long)g->n * (g->c / 8) * a.Do * a.Ho * a.Wo;
  const dim3 grid(grid_for_b(a.total)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (a.Di == 1 && a.kd == 1 && a.kh == 3 && a.kw == 3) {
    if (a.method == ECO_POOL_MAX) hipLaunchKernelGGL((poolb_k3_kernel<ECO_POOL_MAX>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((poolb_k3_kernel<ECO_POOL_AVE>), grid, block, 0, s, a);
    return check_launch("eco_poolb_forward");
  }
  hipLaunchKernelGGL((poolb_kernel), grid, block, 0, s, a);
  return check_launch("eco_poolb_forward");
}

extern "C" int eco_poolb_avg_affine_forward(int32_t dt, const void* x, const float* bias, const float* bn_scale,
                                            const float* bn_shift, int32_t relu, const eco_view* dst, int64_t n, int32_t c,
                                            int32_t h, int32_t w, void* stream) {
  clear_error();
  const int ns = dt == ECO_DT_BF16 ? 1 : 0;
  ECO_REQUIRE(ns != 0, "poolb affine: storage type must be ECO_DT_BF16");
  ECO_REQUIRE(x && dst && dst->ptr && n > 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0,
              "poolb affine: bad argument (channels must be a positive multiple of 8, got %d)", c);
  ECO_REQUIRE(!bn_scale == !bn_shift, "poolb affine: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(dst->t >= 1 && dst->stride_c >= 1, "poolb affine: view needs t >= 1 and stride_c >= 1");
  PoolBAffArgs a;
  a.x = x; a.bias = bias; a.scale = bn_scale; a.shift = bn_shift; a.dst = *dst;
  a.CB = c / 8; a.H = h; a.W = w;
  a.total = (long)n * a.CB * h * w;
  a.floor_v = relu ? 0.0f : -FLT_MAX;
  const dim3 grid(grid_for_b(a.total)), block(256);
  hipLaunchKernelGGL((poolb_avg_affine_kernel), grid, block, 0, (hipStream_t)stream, a);
  return check_launch("eco_poolb_avg_affine_forward");
}

extern "C" int eco_global_avgpool_fc_b_forward(const void* x, int32_t dt, const float* w, const float* bias, float* y,
                                               int64_t b, int64_t c, int64_t s, int64_t n_out, int64_t wk, int64_t c0,
                                               int accumulate, void* stream) {
  clear_error();
  ECO_REQUIRE(x && w && y && b > 0 && c > 0 && s > 0 && n_out > 0, "global_avgpool_fc_b: bad argument");
  const int ns = dt == ECO_DT_BF16 ? 1 : 0;
  ECO_REQUIRE(ns != 0, "global_avgpool_fc_b: storage type must be ECO_DT_BF16");
  ECO_REQUIRE(c % 8 == 0 && c <= kTailBMaxC, "global_avgpool_fc_b: %ld channels (multiple of 8, at most %d)", (long)c, kTailBMaxC);
  ECO_REQUIRE(c0 >= 0 && c0 + c <= wk, "global_avgpool_fc_b: weight columns [%ld,%ld) outside row length %ld", (long)c0,
              (long)(c0 + c), (long)wk);
  ECO_REQUIRE(b <= 65535 && s < 2147483647l, "global_avgpool_fc_b: batch too large for one launch");
  dim3 grid((unsigned)ceil_div(n_out, kTailBOut), (unsigned)b);
  hipLaunchKernelGGL((global_avgpool_fc_b_kernel), grid, dim3(kTailBThreads), 0, (hipStream_t)stream, x, w, bias, y,
                     (int)c, (int)s, (int)n_out, (int)wk, (int)c0, accumulate);
  return check_launch("eco_global_avgpool_fc_b_forward");
}
