// eco_wino.hip -- Winograd F(2x2, 3x3) front and back ends for stride-1, pad-1 (kd)x3x3 convolutions.
//
// The spatial 3x3 part of the kernel is evaluated with the minimal-filtering algorithm of Lavin & Gray
// ("Fast Algorithms for Convolutional Neural Networks", 2015): a 2x2 output tile needs 16 multiplies per
// (input channel, depth tap) instead of 36, so the MFMA work of a 3x3x3 convolution drops 2.25x; the depth
// taps stay direct.  Three launches replace one conv:
//   1. wino_input_kernel :  V[p][b][c][d][th][tw] = (B^T x_tile B)[p],  p = 4*i + j, 4x4 input tiles at stride 2
//   2. 16 independent (3,1,1) convolutions M_p = U_p (*) V_p over (c, depth tap) -- the ordinary conv kernel,
//      one grid slice per transform point (eco_conv_forward_batched), weights U_p = (G g G^T)[p]
//   3. wino_output_kernel:  y_tile = A^T m A, then the usual fused epilogue (bias, residual, raw store, folded
//      BN, ReLU, strided views)
// 1 and 3 are HBM-bound streaming kernels (V and M are 4x the activation size each).  Results differ from the
// direct evaluation by fp32 rounding only (~1e-6 relative; the path's tolerance is 1e-3).
#include <stdint.h>
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kWinoThreads = 256;

// One thread per (b, c, d, th, tw): 4x4 input tile at rows 2*th-1.., cols 2*tw-1.. (zero outside the image).
__global__ __launch_bounds__(kWinoThreads) void wino_input_kernel(const float* x, float* v, long planes /* b*c*d */,
                                                                  int H, int W, int TH, int TW) {
  const long tiles = planes * TH * TW;
  for (long idx = (long)blockIdx.x * kWinoThreads + threadIdx.x; idx < tiles; idx += (long)gridDim.x * kWinoThreads) {
    const int tw = (int)(idx % TW);
    const long r = idx / TW;
    const int th = (int)(r % TH);
    const long plane = r / TH;
    const float* xp = x + plane * H * W;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = 2 * th - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = 2 * tw - 1 + j;
        const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        d[i][j] = ok ? ld(xp + (ok ? (long)h * W + w : 0l)) : 0.0f;
      }
    }
    // t = B^T d  (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]), then v = t B
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j];
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = d[2][j] - d[1][j];
      t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st(v + (long)(4 * i + 0) * tiles + idx, t[i][0] - t[i][2]);
      st(v + (long)(4 * i + 1) * tiles + idx, t[i][1] + t[i][2]);
      st(v + (long)(4 * i + 2) * tiles + idx, t[i][2] - t[i][1]);
      st(v + (long)(4 * i + 3) * tiles + idx, t[i][1] - t[i][3]);
    }
  }
}

struct WinoOutArgs {
  const float* m;
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act;
  int relu;
  int n, cout, D, H, W, TH, TW;
};

// One thread per (b, k, d, th, tw): y = A^T m A (A^T = [1 1 1 0; 0 1 -1 -1]) and the conv epilogue on the
// 2x2 outputs that fall inside the image.
__global__ __launch_bounds__(kWinoThreads) void wino_output_kernel(const WinoOutArgs a) {
  const long tiles = (long)a.n * a.cout * a.D * a.TH * a.TW;
  const int s_out = a.D * a.H * a.W;
  for (long idx = (long)blockIdx.x * kWinoThreads + threadIdx.x; idx < tiles; idx += (long)gridDim.x * kWinoThreads) {
    const int tw = (int)(idx % a.TW);
    long r = idx / a.TW;
    const int th = (int)(r % a.TH);
    r /= a.TH;
    const int d = (int)(r % a.D);
    r /= a.D;
    const int ch = (int)(r % a.cout);
    const int img = (int)(r / a.cout);
    float m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = ld(a.m + (long)(4 * i + j) * tiles + idx);
    float s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[0][j] = m[0][j] + m[1][j] + m[2][j];
      s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    float y[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      y[p][0] = s[p][0] + s[p][1] + s[p][2];
      y[p][1] = s[p][1] - s[p][2] - s[p][3];
    }
    const float b = a.bias ? ld(a.bias + ch) : 0.0f;
    const float sc = a.bn_scale ? ld(a.bn_scale + ch) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + ch) : 0.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int h = 2 * th + p;
      if (h >= a.H) continue;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int w = 2 * tw + q;
        if (w >= a.W) continue;
        const int sp = (d * a.H + h) * a.W + w;
        (void)s_out;
        float val = y[p][q] + b;
        if (a.residual.ptr)
          val += ld((const float*)a.residual.ptr + view_base(a.residual, img, sp) + (long)ch * a.residual.stride_c);
        if (a.raw.ptr) st(a.raw.ptr + view_base(a.raw, img, sp) + (long)ch * a.raw.stride_c, val);
        if (a.act.ptr) {
          float o = val * sc + sh;
          if (a.relu) o = fmaxf(o, 0.0f);
          st(a.act.ptr + view_base(a.act, img, sp) + (long)ch * a.act.stride_c, o);
        }
      }
    }
  }
}

static unsigned wino_grid(long total) {
  long b = ceil_div(total, kWinoThreads);
  if (b > 1048576) b = 1048576;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace eco

using namespace eco;

extern "C" int eco_wino_weight_transform(const float* w, int32_t cout, int32_t cin, int32_t kd, float* u) {
  clear_error();
  ECO_REQUIRE(w && u && cout > 0 && cin > 0 && kd > 0, "winograd weights: bad argument");
  // u[p][co][ci][z] = (G g G^T)[i][j], p = 4*i + j, g = w[co][ci][z][0..2][0..2], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
  const long plane = (long)cout * cin * kd;
  for (long e = 0; e < plane; ++e) {
    const float* g = w + e * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      t[0][j] = g[0 * 3 + j];
      t[1][j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
      t[2][j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
      t[3][j] = g[2 * 3 + j];
    }
    for (int i = 0; i < 4; ++i) {
      u[(long)(4 * i + 0) * plane + e] = t[i][0];
      u[(long)(4 * i + 1) * plane + e] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[(long)(4 * i + 2) * plane + e] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[(long)(4 * i + 3) * plane + e] = t[i][2];
    }
  }
  return ECO_OK;
}

extern "C" int eco_wino_input_forward(const float* x, float* v, int64_t planes, int32_t h, int32_t w, void* stream) {
  clear_error();
  ECO_REQUIRE(x && v && planes > 0 && h > 0 && w > 0, "winograd input transform: bad argument");
  const int TH = (h + 1) / 2, TW = (w + 1) / 2;
  hipLaunchKernelGGL((wino_input_kernel), dim3(wino_grid(planes * TH * TW)), dim3(kWinoThreads), 0, (hipStream_t)stream,
                     x, v, (long)planes, h, w, TH, TW);
  return check_launch("eco_wino_input_forward");
}

extern "C" int eco_wino_output_forward(const float* m, int32_t n, int32_t cout, int32_t d, int32_t h, int32_t w,
                                       const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  ECO_REQUIRE(m && ep && n > 0 && cout > 0 && d > 0 && h > 0 && w > 0, "winograd output transform: bad argument");
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "winograd output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "winograd output transform: bn_scale and bn_shift must be given together");
  const eco_view* views[3] = {&ep->residual, &ep->raw, &ep->act};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "winograd output transform: view needs t >= 1 and stride_c >= 1");
  WinoOutArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.relu = ep->relu;
  a.n = n; a.cout = cout; a.D = d; a.H = h; a.W = w; a.TH = (h + 1) / 2; a.TW = (w + 1) / 2;
  const long tiles = (long)n * cout * d * a.TH * a.TW;
  hipLaunchKernelGGL((wino_output_kernel), dim3(wino_grid(tiles)), dim3(kWinoThreads), 0, (hipStream_t)stream, a);
  return check_launch("eco_wino_output_forward");
}
