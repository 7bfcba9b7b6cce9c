// eco_wino.hip -- Winograd F(MxM, 3x3) front and back ends (M = 2 or 4) for stride-1, pad-1 (kd)x3x3 convolutions.
//
// The spatial 3x3 part of the kernel is evaluated with the minimal-filtering algorithm of Lavin & Gray
// ("Fast Algorithms for Convolutional Neural Networks", 2015): an MxM output tile needs (M+2)^2 multiplies per
// (input channel, depth tap) instead of 9*M^2, so the MFMA work of a 3x3x3 convolution drops 2.25x (M = 2) or
// 4x (M = 4); the depth taps stay direct.  With T = M + 2, three launches replace one conv:
//   1. wino_input_kernel :  V[p][b][c][d][th][tw] = (B^T x_tile B)[p],  p = T*i + j, TxT input tiles at stride M
//   2. T*T independent (3,1,1) convolutions M_p = U_p (*) V_p over (c, depth tap) -- the ordinary conv kernel,
//      one grid slice per transform point (eco_conv_forward_batched), weights U_p = (G g G^T)[p]
//   3. wino_output_kernel:  y_tile = A^T m A, then the usual fused epilogue (bias, residual, raw store, folded
//      BN, ReLU, strided views)
// 1 and 3 are HBM-bound streaming kernels (V and M are T^2/M^2 = 4x / 2.25x the activation size each).  Results
// differ from the direct evaluation by fp32 rounding only: ~1e-6 relative for M = 2, ~1e-5 for M = 4 (larger
// transform constants); the path's tolerance is 1e-3.
#include <stdint.h>
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kWinoThreads = 256;

// Transform matrices (Lavin & Gray 2015, section 4).  M = outputs per tile and dimension, T = M + 2 inputs.
template <int M>
struct WinoMat;
template <>
struct WinoMat<2> {
  static constexpr int T = 4;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float G[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};
template <>
struct WinoMat<4> {
  static constexpr int T = 6;
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                     {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][3] = {{0.25f, 0, 0},
                                    {-1.0f / 6, -1.0f / 6, -1.0f / 6},
                                    {-1.0f / 6, 1.0f / 6, -1.0f / 6},
                                    {1.0f / 24, 1.0f / 12, 1.0f / 6},
                                    {1.0f / 24, -1.0f / 12, 1.0f / 6},
                                    {0, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
};

template <int VEC>
struct WinoVec;
template <>
struct WinoVec<1> { typedef float type; };
template <>
struct WinoVec<2> { typedef float2 type; };
template <>
struct WinoVec<4> { typedef float4 type; };

// One thread per (b, c, d, th, tw): T x T input tile at rows M*th-1.., cols M*tw-1.. (zero outside the image),
// v = B^T d B scattered to the T*T transform planes.  The matrices are compile-time constants: zero terms
// vanish and +-1 terms become adds when the loops are unrolled.  The M middle columns of a tile row start at a
// multiple of M, so they are fetched with 4*VEC-byte loads when W % VEC == 0 (launcher's choice).
template <int M, int VEC>
__global__ __launch_bounds__(kWinoThreads) void wino_input_kernel(const float* x, float* v, long planes /* b*c*d */,
                                                                  int H, int W, int TH, int TW) {
  constexpr int T = WinoMat<M>::T;
  typedef typename WinoVec<VEC>::type vec_t;
  const long tiles = planes * TH * TW;
  for (long idx = (long)blockIdx.x * kWinoThreads + threadIdx.x; idx < tiles; idx += (long)gridDim.x * kWinoThreads) {
    const int tw = (int)(idx % TW);
    const long r = idx / TW;
    const int th = (int)(r % TH);
    const long plane = r / TH;
    const float* xp = x + plane * H * W;
    float d[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const int h = M * th - 1 + i;
      const bool hok = (unsigned)h < (unsigned)H;
      const float* rp = xp + (hok ? (long)h * W : 0l);
      const int wl = M * tw - 1, wr = M * tw + M;  // the two edge columns
      const bool lok = hok && wl >= 0, rok = hok && wr < W;
      d[i][0] = lok ? ld(rp + (lok ? wl : 0)) : 0.0f;
      d[i][T - 1] = rok ? ld(rp + (rok ? wr : 0)) : 0.0f;
#pragma unroll
      for (int j0 = 0; j0 < M; j0 += VEC) {
        const int w0 = M * tw + j0;
        const bool ok = hok && w0 < W;  // W % VEC == 0: the VEC columns are inside or outside together
        vec_t q = ld((const vec_t*)(rp + (ok ? w0 : 0)));
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[i][1 + j0 + e] = ok ? ((const float*)&q)[e] : 0.0f;
      }
    }
    float t[T][T];  // t = B^T d
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = 0; j < T; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k)
          if (WinoMat<M>::BT[i][k] != 0.0f) acc += WinoMat<M>::BT[i][k] * d[k][j];
        t[i][j] = acc;
      }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = 0; j < T; ++j) {  // v = t B
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k)
          if (WinoMat<M>::BT[j][k] != 0.0f) acc += t[i][k] * WinoMat<M>::BT[j][k];
        st(v + (long)(T * i + j) * tiles + idx, acc);
      }
  }
}

struct WinoOutArgs {
  const float* m;
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;
  int relu;
  int n, cout, D, H, W, TH, TW;
};

// One thread per (b, k, d, th, tw): y = A^T m A and the conv epilogue on the M x M outputs that fall inside
// the image.  VEC consecutive outputs of a row are moved with one 4*VEC-byte access (the launcher checks that
// W, the view strides and the pointers allow it), so a wave writes whole 256/512-byte row segments.
template <int M, int VEC>
__global__ __launch_bounds__(kWinoThreads) void wino_output_kernel(const WinoOutArgs a) {
  constexpr int T = WinoMat<M>::T;
  typedef typename WinoVec<VEC>::type vec_t;
  const long tiles = (long)a.n * a.cout * a.D * a.TH * a.TW;
  for (long idx = (long)blockIdx.x * kWinoThreads + threadIdx.x; idx < tiles; idx += (long)gridDim.x * kWinoThreads) {
    const int tw = (int)(idx % a.TW);
    long r = idx / a.TW;
    const int th = (int)(r % a.TH);
    r /= a.TH;
    const int d = (int)(r % a.D);
    r /= a.D;
    const int ch = (int)(r % a.cout);
    const int img = (int)(r / a.cout);
    float m[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = 0; j < T; ++j) m[i][j] = ld(a.m + (long)(T * i + j) * tiles + idx);
    float s[M][T];  // s = A^T m
#pragma unroll
    for (int p = 0; p < M; ++p)
#pragma unroll
      for (int j = 0; j < T; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k)
          if (WinoMat<M>::AT[p][k] != 0.0f) acc += WinoMat<M>::AT[p][k] * m[k][j];
        s[p][j] = acc;
      }
    const float b = a.bias ? ld(a.bias + ch) : 0.0f;
    const float sc = a.bn_scale ? ld(a.bn_scale + ch) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + ch) : 0.0f;
    const long o_res = a.residual.ptr ? view_base(a.residual, img, 0) + (long)ch * a.residual.stride_c : 0;
    const long o_raw = a.raw.ptr ? view_base(a.raw, img, 0) + (long)ch * a.raw.stride_c : 0;
    const long o_act = a.act.ptr ? view_base(a.act, img, 0) + (long)ch * a.act.stride_c : 0;
    const long o_act2 = a.act2.ptr ? view_base(a.act2, img, 0) + (long)ch * a.act2.stride_c : 0;
#pragma unroll
    for (int p = 0; p < M; ++p) {
      const int h = M * th + p;
      if (h >= a.H) continue;
#pragma unroll
      for (int q0 = 0; q0 < M; q0 += VEC) {
        const int w0 = M * tw + q0;
        if (w0 >= a.W) continue;  // W % VEC == 0: the VEC outputs are inside or outside together
        float val[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float y = 0.0f;  // y = s A
#pragma unroll
          for (int k = 0; k < T; ++k)
            if (WinoMat<M>::AT[q0 + e][k] != 0.0f) y += s[p][k] * WinoMat<M>::AT[q0 + e][k];
          val[e] = y + b;
        }
        const int sp = (d * a.H + h) * a.W + w0;
        if (a.residual.ptr) {
          const vec_t rv = ld((const vec_t*)((const float*)a.residual.ptr + o_res + sp));
#pragma unroll
          for (int e = 0; e < VEC; ++e) val[e] += ((const float*)&rv)[e];
        }
        if (a.raw.ptr) {
          vec_t ov;
#pragma unroll
          for (int e = 0; e < VEC; ++e) ((float*)&ov)[e] = val[e];
          st((vec_t*)(a.raw.ptr + o_raw + sp), ov);
        }
        if (a.act.ptr) {
          vec_t ov;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float o = val[e] * sc + sh;
            if (a.relu) o = fmaxf(o, 0.0f);
            ((float*)&ov)[e] = o;
          }
          st((vec_t*)(a.act.ptr + o_act + sp), ov);
          if (a.act2.ptr) st((vec_t*)(a.act2.ptr + o_act2 + sp), ov);
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

template <int M>
static void weight_transform(const float* w, long plane, float* u) {
  constexpr int T = WinoMat<M>::T;
  // u[p][co][ci][z] = (G g G^T)[i][j], p = T*i + j, g = w[co][ci][z][0..2][0..2]
  for (long e = 0; e < plane; ++e) {
    const float* g = w + e * 9;
    float t[T][3];
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < 3; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) acc += WinoMat<M>::G[i][k] * g[k * 3 + j];
        t[i][j] = acc;
      }
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) acc += t[i][k] * WinoMat<M>::G[j][k];
        u[(long)(T * i + j) * plane + e] = acc;
      }
  }
}

}  // namespace eco

using namespace eco;

extern "C" int eco_wino_weight_transform(const float* w, int32_t cout, int32_t cin, int32_t kd, int32_t tile_m,
                                         float* u) {
  clear_error();
  ECO_REQUIRE(w && u && cout > 0 && cin > 0 && kd > 0, "winograd weights: bad argument");
  ECO_REQUIRE(tile_m == 2 || tile_m == 4, "winograd: output tile must be 2 (F(2x2,3x3)) or 4 (F(4x4,3x3)), got %d", tile_m);
  const long plane = (long)cout * cin * kd;
  if (tile_m == 2) weight_transform<2>(w, plane, u);
  else weight_transform<4>(w, plane, u);
  return ECO_OK;
}

extern "C" int eco_wino_input_forward(const float* x, float* v, int64_t planes, int32_t h, int32_t w, int32_t tile_m,
                                      void* stream) {
  clear_error();
  ECO_REQUIRE(x && v && planes > 0 && h > 0 && w > 0, "winograd input transform: bad argument");
  ECO_REQUIRE(tile_m == 2 || tile_m == 4, "winograd: output tile must be 2 or 4, got %d", tile_m);
  const int TH = (h + tile_m - 1) / tile_m, TW = (w + tile_m - 1) / tile_m;
  const dim3 grid(wino_grid(planes * TH * TW)), block(kWinoThreads);
  hipStream_t st_ = (hipStream_t)stream;
  int vec = tile_m;
  while (vec > 1 && (w % vec || ((uintptr_t)x % (4 * vec)))) vec /= 2;
  const long pl = (long)planes;
  if (tile_m == 2) {
    if (vec == 2) hipLaunchKernelGGL((wino_input_kernel<2, 2>), grid, block, 0, st_, x, v, pl, h, w, TH, TW);
    else hipLaunchKernelGGL((wino_input_kernel<2, 1>), grid, block, 0, st_, x, v, pl, h, w, TH, TW);
  } else {
    if (vec == 4) hipLaunchKernelGGL((wino_input_kernel<4, 4>), grid, block, 0, st_, x, v, pl, h, w, TH, TW);
    else if (vec == 2) hipLaunchKernelGGL((wino_input_kernel<4, 2>), grid, block, 0, st_, x, v, pl, h, w, TH, TW);
    else hipLaunchKernelGGL((wino_input_kernel<4, 1>), grid, block, 0, st_, x, v, pl, h, w, TH, TW);
  }
  return check_launch("eco_wino_input_forward");
}

extern "C" int eco_wino_output_forward(const float* m, int32_t n, int32_t cout, int32_t d, int32_t h, int32_t w,
                                       int32_t tile_m, const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  ECO_REQUIRE(m && ep && n > 0 && cout > 0 && d > 0 && h > 0 && w > 0, "winograd output transform: bad argument");
  ECO_REQUIRE(tile_m == 2 || tile_m == 4, "winograd: output tile must be 2 or 4, got %d", tile_m);
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "winograd output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "winograd output transform: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "winograd output transform: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "winograd output transform: segmented (sibling) launches exist for the direct kernels only");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "winograd output transform: view needs t >= 1 and stride_c >= 1");
  WinoOutArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.n = n; a.cout = cout; a.D = d; a.H = h; a.W = w;
  a.TH = (h + tile_m - 1) / tile_m; a.TW = (w + tile_m - 1) / tile_m;
  const long tiles = (long)n * cout * d * a.TH * a.TW;
  // widest access every view allows: W, the strides and the base pointers must be multiples of it
  int vec = tile_m;
  auto limit = [&](const eco_view& v) {
    if (!v.ptr) return;
    while (vec > 1 && (((uintptr_t)v.ptr % (4 * vec)) || v.stride_b % vec || v.stride_t % vec || v.stride_c % vec)) vec /= 2;
  };
  while (vec > 1 && w % vec) vec /= 2;
  limit(a.residual); limit(a.raw); limit(a.act); limit(a.act2);
  const dim3 grid(wino_grid(tiles)), block(kWinoThreads);
  hipStream_t st_ = (hipStream_t)stream;
  if (tile_m == 2) {
    if (vec == 2) hipLaunchKernelGGL((wino_output_kernel<2, 2>), grid, block, 0, st_, a);
    else hipLaunchKernelGGL((wino_output_kernel<2, 1>), grid, block, 0, st_, a);
  } else {
    if (vec == 4) hipLaunchKernelGGL((wino_output_kernel<4, 4>), grid, block, 0, st_, a);
    else if (vec == 2) hipLaunchKernelGGL((wino_output_kernel<4, 2>), grid, block, 0, st_, a);
    else hipLaunchKernelGGL((wino_output_kernel<4, 1>), grid, block, 0, st_, a);
  }
  return check_launch("eco_wino_output_forward");
}
