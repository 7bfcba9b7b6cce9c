// eco_wino3.hip -- Winograd F(4x4x4, 3x3x3): front and back end of the 3-D trunk's stride-1 pad-1 3x3x3 convolutions
// (res3a_2 ... res5b_2: models_ECO_Lite/kinetics/deploy.prototxt:1162-1660).
//
// Rounds 2-4 evaluated these layers with the minimal-filtering algorithm over (h, w) only and took the three depth taps
// directly: 36 points x 3 taps = 108 multiplies per (input channel, 4x4 output tile and plane).  The depth axis of every
// trunk stage is a multiple of four at the benchmark geometry (16 / 8 / 4 planes), so the same algorithm nests a third
// time without tile padding: a 4x4x4 output tile costs 216 multiplies per input channel instead of 432 -- the
// transformed-domain GEMMs, the largest kernel family of the step, do HALF the work (measured, random operands:
// res3 0.81 -> 0.42 ms, res4 0.46 -> 0.25, res5 0.26 -> 0.13 per layer) -- while V and M grow from 2.53x / 2.25x to
// 3.375x the activation (6^3 points per 4^3 outputs).  The reference leaves the algorithm to cuDNN
// (cudnn_conv_layer.cu:15-65); results differ from the direct evaluation by fp32 rounding (tests: 1e-4 of the largest
// output per layer, 1e-3 on logits).
//
//   V3[p][c/2][r][c%2]   p = (pz*6 + py)*6 + px, r = ((td*n + b)*TH + th)*TW + tw   (TD = ceil(D/4) depth tiles)
//   M3[p][slice][cout][r]
//
// are what eco_wgemm.hip's kernel already consumes / produces for kd = 1 (a plan with d = TD, points = 216): the GEMM
// only sees six times as many, three times shorter problems.  This file owns the two HBM-bound ends:
//
//   wino3_input_kernel   one workgroup per (channel pair, depth tile, group of GB images).  Phase 1 walks the zero-
//                        padded planes in 16-byte column groups: the six input planes of the depth tile are loaded
//                        (coalesced along w, each element once per depth tile), transformed along DEPTH in registers
//                        and parked in LDS as six transformed planes.  Phase 2: one thread per (depth point, tile)
//                        reads its 6x6 window from LDS (halo overlap costs LDS reads, not global ones), applies the
//                        2-D transform and writes its 36 points for both channels of the pair as 8-byte stores,
//                        consecutive lanes on consecutive positions.
//   wino3_output_kernel  one workgroup per (output channel, depth tile, group of GB images).  Phase A: one thread per
//                        (depth point, tile) loads its 36 products (each a contiguous run across the wave), sums the
//                        split-K slices, applies the 2-D output transform and parks 4 rows x 4 columns in LDS.  Phase B:
//                        one thread per (output plane, tile row) folds the six depth points, applies the fused epilogue
//                        (bias, Eltwise residual, raw store, folded BN, ReLU, both activated destinations; strided
//                        views) and stores one 16-byte row piece; lanes walk a plane's rows in memory order.
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kW3P = 216;

// 1-D transforms of F(4,3) (Lavin & Gray 2015), factored.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w3_bt(const float (&x)[6], float (&y)[6]) {
  const float a = x[4] - 4.0f * x[2], b = x[3] - 4.0f * x[1];
  const float c = x[4] - x[2], d = 2.0f * (x[3] - x[1]);
  y[0] = 4.0f * x[0] - 5.0f * x[2] + x[4];
  y[1] = a + b;
  y[2] = a - b;
  y[3] = c + d;
  y[4] = c - d;
  y[5] = 4.0f * x[1] - 5.0f * x[3] + x[5];
}
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void w3_at(const float (&m)[6], float (&y)[4]) {
  const float t1 = m[1] + m[2], t2 = m[1] - m[2], t3 = m[3] + m[4], t4 = m[3] - m[4];
  y[0] = m[0] + t1 + t3;
  y[1] = t2 + 2.0f * t4;
  y[2] = t1 + 4.0f * t3;
  y[3] = t2 + 8.0f * t4 + m[5];
}

// A 16-byte LDS read that IS one ds_read_b128: from dynamic LDS with run-time row pitches the compiler splits a float4
// load into two ds_read2_b32 (it cannot see the 16-byte alignment the layout guarantees).  Told that the pointer is an
// LDS address (address space 3) with 16-byte alignment it emits the instruction itself AND tracks its lgkmcnt -- round 5
// issued the read from inline asm with a plain "=v" output, whose register the compiler took for valid at once (round-5
// advisor finding: a copy or spill between the 18 reads and the wait would have moved stale data).
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 lds_ld16(const float* p) {
#ifdef ECO_EMU
  return *(const f4*)p;
#else
  typedef __attribute__((address_space(3))) const f4 lds_f4_t;
  return *(lds_f4_t*)__builtin_assume_aligned((__attribute__((address_space(3))) const float*)p, 16);
#endif
}
__device__ __forceinline__ void lds_wait(f4&, f4&, f4&) {}   // (the compiler's own wait counts cover the reads now)

struct Wino3InArgs {
  const float* x;   // [n][cin][D][H][W]
  float* v;         // V3
  int n, cin, D, H, W, TD, TH, TW;
  int GB, nbg;      // images per workgroup, image groups
  int PH, PW;       // parked plane: rows h + 1 (one zero row above), columns w + 4 (16-byte aligned interior), zero borders
  int Q;            // positions per channel-pair row: TD * n * TH * TW
  long v_pstride;   // floats between points
};

// VEC: widest load the input allows along w (W % VEC == 0, x aligned to 4*VEC bytes).
template <int VEC>
__global__ __launch_bounds__(512) void wino3_input_kernel(const Wino3InArgs a) {
  ECO_DYNAMIC_LDS(zd);                    // [6 pz][2 e][GB][PH][PW]
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  // each XCD takes a contiguous range of (channel pair, depth tile, image group): the runs neighbouring workgroups write
  // (392 / 256 bytes per point: not whole cache lines) meet in ONE L2 and leave it as full lines
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bg = wg % a.nbg, t0 = wg / a.nbg;
  const int td = t0 % a.TD, cp = t0 / a.TD;
  const int b0 = bg * a.GB;
  const int pwv = a.PW / 4;
  const int plane = a.PH * a.PW;          // floats per parked plane
  const int pstride = 2 * a.GB * plane;   // floats between depth points in LDS
  const long hw = (long)a.H * a.W;

  // ---- phase 1: depth transform of the six planes, parked with zero borders ----
  const int slots = 2 * a.GB * a.PH * pwv;
  for (int s = tid; s < slots; s += nthr) {
    const int pv = s % pwv;
    int t = s / pwv;
    const int ph = t % a.PH;
    t /= a.PH;
    const int bl = t % a.GB, e = t / a.GB;
    const int h = ph - 1, w0 = 4 * pv - 4, b = b0 + bl;
    float xin[6][4];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) xin[k][j] = 0.0f;
    if ((unsigned)h < (unsigned)a.H && w0 >= 0 && w0 < a.W && b < a.n) {
      const float* xp = a.x + (((long)b * a.cin + 2 * cp + e) * a.D) * hw + (long)h * a.W + w0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int d = 4 * td - 1 + k;                      // workgroup-uniform
        if ((unsigned)d >= (unsigned)a.D) continue;
        const float* q = xp + (long)d * hw;
        if (VEC == 4) {
          const float4 v4 = ld((const float4*)q);
          xin[k][0] = v4.x; xin[k][1] = v4.y; xin[k][2] = v4.z; xin[k][3] = v4.w;
        } else if (VEC == 2) {
          const float2 lo = ld((const float2*)q);
          xin[k][0] = lo.x; xin[k][1] = lo.y;
          if (w0 + 2 < a.W) {
            const float2 hi = ld((const float2*)(q + 2));
            xin[k][2] = hi.x; xin[k][3] = hi.y;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (w0 + j < a.W) xin[k][j] = ld(q + j);
        }
      }
    }
    float4 out[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float col[6] = {xin[0][j], xin[1][j], xin[2][j], xin[3][j], xin[4][j], xin[5][j]};
      float y[6];
      w3_bt(col, y);
#pragma unroll
      for (int pz = 0; pz < 6; ++pz) ((float*)&out[pz])[j] = y[pz];
    }
    float* dst = zd + ((e * a.GB + bl) * a.PH + ph) * a.PW + 4 * pv;
#pragma unroll
    for (int pz = 0; pz < 6; ++pz) *(float4*)(dst + pz * pstride) = out[pz];
  }
  __syncthreads();

  // ---- phase 2: 2-D transform of every (depth point, tile) window, both channels of the pair ----
  const int tpi = a.TH * a.TW, tpw = a.GB * tpi;
  for (int it = tid; it < 6 * tpw; it += nthr) {
    const int pz = it / tpw, tile = it - pz * tpw;
    const int bl = tile / tpi, tt = tile - bl * tpi;
    const int th = tt / a.TW, tw = tt - th * a.TW;
    const int b = b0 + bl;
    if (b >= a.n) continue;
    float v[2][6][6];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* src = zd + pz * pstride + ((e * a.GB + bl) * a.PH + 4 * th) * a.PW + 4 * tw + 3;
      float t1[6][6];
      // a window row = columns 4 tw + 3 ... 4 tw + 8 of the parked plane: three aligned 16-byte reads (the two single
      // columns as 4-byte reads at a lane stride of 16 bytes are 4-way bank conflicts)
      f4 lo[6], mid[6], hi[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        lo[i] = lds_ld16(src + i * a.PW - 3);
        mid[i] = lds_ld16(src + i * a.PW + 1);
        hi[i] = lds_ld16(src + i * a.PW + 5);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) lds_wait(lo[i], mid[i], hi[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) {        // rows of the window: transform along w
        const float row[6] = {lo[i].w, mid[i].x, mid[i].y, mid[i].z, mid[i].w, hi[i].x};
        w3_bt(row, t1[i]);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {        // ... then along h
        const float col[6] = {t1[0][j], t1[1][j], t1[2][j], t1[3][j], t1[4][j], t1[5][j]};
        float y[6];
        w3_bt(col, y);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[e][i][j] = y[i];
      }
    }
    const long r = ((long)(td * a.n + b) * a.TH + th) * a.TW + tw;
    float* vo = a.v + (long)pz * 36 * a.v_pstride + ((long)cp * a.Q + r) * 2;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) st((float2*)(vo + (long)(6 * i + j) * a.v_pstride), make_float2(v[0][i][j], v[1][i][j]));
  }
}

struct Wino3OutArgs {
  const float* m;   // M3[216][ksplit][cout][Q]
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;
  int relu;
  int n, cout, D, H, W, TD, TH, TW;
  int GB, nbg, Q, ksplit;
  long m_pstride;   // floats between points = ksplit * cout * Q
};

template <int VEC>
struct W3Vec;
template <>
struct W3Vec<1> { typedef float type; };
template <>
struct W3Vec<2> { typedef float2 type; };
template <>
struct W3Vec<4> { typedef float4 type; };

// VEC: widest access W and every view allow (W % VEC == 0, bases and strides multiples of VEC floats).
template <int VEC>
__global__ __launch_bounds__(512) void wino3_output_kernel(const Wino3OutArgs a) {
  typedef typename W3Vec<VEC>::type vec_t;
  ECO_DYNAMIC_LDS(lds);
  float4* const sp4 = (float4*)lds;       // [6 pz][4 rows][tpw tiles]: a tile row of the 2-D transformed products
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);   // as in the input transform: neighbouring runs of M through one L2
  const int bg = wg % a.nbg, t0 = wg / a.nbg;
  const int td = t0 % a.TD, ch = t0 / a.TD;
  const int b0 = bg * a.GB;
  const int tpi = a.TH * a.TW, tpw = a.GB * tpi;

  // ---- phase A: 2-D output transform per (depth point, tile) ----
  for (int it = tid; it < 6 * tpw; it += nthr) {
    const int pz = it / tpw, tile = it - pz * tpw;
    const int bl = tile / tpi, tt = tile - bl * tpi;
    const int b = b0 + bl;
    if (b >= a.n) continue;
    const long r = (long)(td * a.n + b) * tpi + tt;
    const float* mp = a.m + (long)pz * 36 * a.m_pstride + (long)ch * a.Q + r;
    float mm[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) mm[i][j] = ld(mp + (long)(6 * i + j) * a.m_pstride);
    for (int sl = 1; sl < a.ksplit; ++sl) {
      const float* ms = mp + (long)sl * a.cout * a.Q;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) mm[i][j] += ld(ms + (long)(6 * i + j) * a.m_pstride);
    }
    float s4[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float col[6] = {mm[0][j], mm[1][j], mm[2][j], mm[3][j], mm[4][j], mm[5][j]};
      float y[4];
      w3_at(col, y);
#pragma unroll
      for (int i = 0; i < 4; ++i) s4[i][j] = y[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y[4];
      w3_at(s4[i], y);
      sp4[(pz * 4 + i) * tpw + tile] = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
  __syncthreads();

  // ---- phase B: fold the six depth points, epilogue, store: lanes walk (tw, row, th, image, output plane) ----
  const float bias = a.bias ? ld(a.bias + ch) : 0.0f;
  const float sc = a.bn_scale ? ld(a.bn_scale + ch) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + ch) : 0.0f;
  const int rows = 4 * a.TH;
  for (int it = tid; it < 4 * a.GB * rows * a.TW; it += nthr) {
    const int tw = it % a.TW;
    int t = it / a.TW;
    const int hr = t % rows;              // = 4 * th + row
    t /= rows;
    const int bl = t % a.GB, od = t / a.GB;
    const int th = hr >> 2, row = hr & 3;
    const int b = b0 + bl, d = 4 * td + od, w0 = 4 * tw;
    if (b >= a.n || d >= a.D || hr >= a.H) continue;
    const int tile = (bl * a.TH + th) * a.TW + tw;
    float4 s[6];
#pragma unroll
    for (int pz = 0; pz < 6; ++pz) s[pz] = sp4[(pz * 4 + row) * tpw + tile];
    // y = A^T[od] . s: od 0: s0 + t1 + t3, 1: t2 + 2 t4, 2: t1 + 4 t3, 3: t2 + 8 t4 + s5
    const float c0 = od == 0 ? 1.0f : 0.0f, c5 = od == 3 ? 1.0f : 0.0f, kk = (float)(1 << od);
    const bool odd = od & 1;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s1 = ((const float*)&s[1])[j], s2 = ((const float*)&s[2])[j];
      const float s3 = ((const float*)&s[3])[j], s4v = ((const float*)&s[4])[j];
      const float ta = odd ? s1 - s2 : s1 + s2, tb = odd ? s3 - s4v : s3 + s4v;
      y[j] = c0 * ((const float*)&s[0])[j] + ta + kk * tb + c5 * ((const float*)&s[5])[j] + bias;
    }
    const long spo = ((long)d * a.H + hr) * a.W + w0;
    float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.residual.ptr) {
      const float* rp = a.residual.ptr + view_base(a.residual, b, 0) + (long)ch * a.residual.stride_c + spo;
#pragma unroll
      for (int j0 = 0; j0 < 4; j0 += VEC)
        if (w0 + j0 < a.W) {
          const vec_t q = ld((const vec_t*)(rp + j0));
#pragma unroll
          for (int e = 0; e < VEC; ++e) res[j0 + e] = ((const float*)&q)[e];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] += res[j];
    if (a.raw.ptr) {
      float* op = a.raw.ptr + view_base(a.raw, b, 0) + (long)ch * a.raw.stride_c + spo;
#pragma unroll
      for (int j0 = 0; j0 < 4; j0 += VEC)
        if (w0 + j0 < a.W) {
          vec_t q;
#pragma unroll
          for (int e = 0; e < VEC; ++e) ((float*)&q)[e] = y[j0 + e];
          st((vec_t*)(op + j0), q);
        }
    }
    if (a.act.ptr) {
      float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = y[j] * sc + sh;
        z[j] = a.relu ? fmaxf(v, 0.0f) : v;
      }
      float* op = a.act.ptr + view_base(a.act, b, 0) + (long)ch * a.act.stride_c + spo;
      float* op2 = a.act2.ptr ? a.act2.ptr + view_base(a.act2, b, 0) + (long)ch * a.act2.stride_c + spo : nullptr;
#pragma unroll
      for (int j0 = 0; j0 < 4; j0 += VEC)
        if (w0 + j0 < a.W) {
          vec_t q;
#pragma unroll
          for (int e = 0; e < VEC; ++e) ((float*)&q)[e] = z[j0 + e];
          st((vec_t*)(op + j0), q);
          if (op2) st((vec_t*)(op2 + j0), q);
        }
    }
  }
}

// G of F(4,3): u = G g
static void w3_g(const float (&g)[3], float (&u)[6]) {
  u[0] = 0.25f * g[0];
  u[1] = (-1.0f / 6) * (g[0] + g[1] + g[2]);
  u[2] = (-1.0f / 6) * (g[0] - g[1] + g[2]);
  u[3] = (1.0f / 24) * g[0] + (1.0f / 12) * g[1] + (1.0f / 6) * g[2];
  u[4] = (1.0f / 24) * g[0] - (1.0f / 12) * g[1] + (1.0f / 6) * g[2];
  u[5] = g[2];
}

// Images per workgroup and the LDS the two kernels need for them.
struct Wino3Shape { int GB, nbg, PH, PW, tpw; size_t lds_in, lds_out; int thr_in, thr_out; };
static Wino3Shape wino3_shape(int n, int th, int tw) {
  Wino3Shape s;
  const int tpi = th * tw;
  s.GB = 32 / tpi;
  if (s.GB < 1) s.GB = 1;
  if (s.GB > n) s.GB = n;
  s.nbg = (int)ceil_div(n, s.GB);
  s.PH = 4 * th + 2;
  s.PW = 4 * tw + 8;
  s.tpw = s.GB * tpi;
  s.lds_in = (size_t)6 * 2 * s.GB * s.PH * s.PW * 4;
  s.lds_out = (size_t)6 * 4 * s.tpw * 16;
  auto thr = [](long items) { long t = (items + 63) / 64 * 64; return (int)(t < 64 ? 64 : t > 512 ? 512 : t); };
  s.thr_in = thr(6L * s.tpw);
  s.thr_out = thr(6L * s.tpw);
  return s;
}

}  // namespace eco

using namespace eco;

static int wino3_check_plan(const eco_wgemm_plan* p, int32_t d, int32_t h, int32_t w, const char* who) {
  ECO_REQUIRE(p != nullptr, "%s: null plan", who);
  ECO_REQUIRE(p->points == kW3P && p->kd == 1, "%s: needs an F(4x4x4,3x3x3) plan (points = 216, kd = 1), got points=%d kd=%d", who,
              p->points, p->kd);
  ECO_REQUIRE(d > 0 && h > 0 && w > 0 && p->d == (d + 3) / 4 && p->th == (h + 3) / 4 && p->tw == (w + 3) / 4,
              "%s: plan is for %dx%dx%d tiles, volume %dx%dx%d needs %dx%dx%d", who, p->d, p->th, p->tw, d, h, w, (d + 3) / 4,
              (h + 3) / 4, (w + 3) / 4);
  ECO_REQUIRE(p->n > 0 && p->cin > 0 && p->cin % 16 == 0 && p->cout > 0, "%s: bad plan", who);
  return ECO_OK;
}

extern "C" int64_t eco_wino3_lds_bytes(int32_t n, int32_t th, int32_t tw) {
  if (n <= 0 || th <= 0 || tw <= 0) return -1;
  const Wino3Shape s = wino3_shape(n, th, tw);
  return (int64_t)(s.lds_in > s.lds_out ? s.lds_in : s.lds_out);
}

extern "C" int eco_wino3_weight_transform(const float* w, int32_t cout, int32_t cin, float* u) {
  clear_error();
  ECO_REQUIRE(w && u && cout > 0 && cin > 0, "winograd 3-D weights: bad argument");
  // u[p][co][ci] = (G (x) G (x) G) g, p = (pz*6 + py)*6 + px, g = w[co][ci][0..2][0..2][0..2]
  const long plane = (long)cout * cin;
  for (long e = 0; e < plane; ++e) {
    const float* g = w + e * 27;
    float t1[3][3][6];   // along x
    for (int z = 0; z < 3; ++z)
      for (int y = 0; y < 3; ++y) {
        const float gi[3] = {g[(z * 3 + y) * 3 + 0], g[(z * 3 + y) * 3 + 1], g[(z * 3 + y) * 3 + 2]};
        w3_g(gi, t1[z][y]);
      }
    float t2[3][6][6];   // along y
    for (int z = 0; z < 3; ++z)
      for (int px = 0; px < 6; ++px) {
        const float gi[3] = {t1[z][0][px], t1[z][1][px], t1[z][2][px]};
        float o[6];
        w3_g(gi, o);
        for (int py = 0; py < 6; ++py) t2[z][py][px] = o[py];
      }
    for (int py = 0; py < 6; ++py)
      for (int px = 0; px < 6; ++px) {
        const float gi[3] = {t2[0][py][px], t2[1][py][px], t2[2][py][px]};
        float o[6];
        w3_g(gi, o);
        for (int pz = 0; pz < 6; ++pz) u[(long)((pz * 6 + py) * 6 + px) * plane + e] = o[pz];
      }
  }
  return ECO_OK;
}

extern "C" int eco_wino3_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t d, int32_t h, int32_t w,
                                       void* stream) {
  clear_error();
  if (int rc = wino3_check_plan(plan, d, h, w, "winograd 3-D input transform")) return rc;
  ECO_REQUIRE(x && v, "winograd 3-D input transform: null argument");
  ECO_REQUIRE(((uintptr_t)v & 7) == 0, "winograd 3-D input transform: v must be 8-byte aligned");
  const Wino3Shape s = wino3_shape(plan->n, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_in <= (size_t)kEcoMaxDynamicLds, "winograd 3-D input transform: %dx%d planes need %zu bytes of LDS (max %d)", h, w,
              s.lds_in, kEcoMaxDynamicLds);
  Wino3InArgs a;
  a.x = x; a.v = v; a.n = plan->n; a.cin = plan->cin; a.D = d; a.H = h; a.W = w;
  a.TD = plan->d; a.TH = plan->th; a.TW = plan->tw;
  a.GB = s.GB; a.nbg = s.nbg; a.PH = s.PH; a.PW = s.PW;
  a.Q = (int)plan->q;
  a.v_pstride = (long)(plan->cin / 2) * plan->q * 2;
  const long grid = (long)(plan->cin / 2) * a.TD * a.nbg;
  ECO_REQUIRE(grid < 2147483647l, "winograd 3-D input transform: too many workgroups");
  int vec = 4;
  while (vec > 1 && (w % vec || ((uintptr_t)x % (4 * vec)))) vec /= 2;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 g((unsigned)grid), b((unsigned)s.thr_in);
#define ECO_W3IN(V)                                                                                   \
  do {                                                                                                \
    if (s.lds_in > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wino3_input_kernel<V>), "winograd 3-D input transform"); \
    hipLaunchKernelGGL((wino3_input_kernel<V>), g, b, s.lds_in, st_, a);                              \
  } while (0)
  if (vec == 4) ECO_W3IN(4);
  else if (vec == 2) ECO_W3IN(2);
  else ECO_W3IN(1);
#undef ECO_W3IN
  return check_launch("eco_wino3_input_forward");
}

extern "C" int eco_wino3_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t d, int32_t h, int32_t w,
                                        const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  if (int rc = wino3_check_plan(plan, d, h, w, "winograd 3-D output transform")) return rc;
  ECO_REQUIRE(m && ep, "winograd 3-D output transform: null argument");
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "winograd 3-D output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "winograd 3-D output transform: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "winograd 3-D output transform: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "winograd 3-D output transform: segmented (sibling) launches exist for the direct kernels only");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "winograd 3-D output transform: view needs t >= 1 and stride_c >= 1");
  const Wino3Shape s = wino3_shape(plan->n, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_out <= (size_t)kEcoMaxDynamicLds, "winograd 3-D output transform: %dx%d planes need %zu bytes of LDS (max %d)", h,
              w, s.lds_out, kEcoMaxDynamicLds);
  Wino3OutArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.n = plan->n; a.cout = plan->cout; a.D = d; a.H = h; a.W = w; a.TD = plan->d; a.TH = plan->th; a.TW = plan->tw;
  a.GB = s.GB; a.nbg = s.nbg; a.Q = (int)plan->q; a.ksplit = plan->ksplit;
  a.m_pstride = (long)plan->ksplit * plan->cout * plan->q;
  int vec = 4;
  auto limit = [&](const eco_view& v) {
    if (!v.ptr) return;
    while (vec > 1 && (((uintptr_t)v.ptr % (4 * vec)) || v.stride_b % vec || v.stride_t % vec || v.stride_c % vec)) vec /= 2;
  };
  while (vec > 1 && w % vec) vec /= 2;
  limit(a.residual); limit(a.raw); limit(a.act); limit(a.act2);
  const long grid = (long)plan->cout * a.TD * a.nbg;
  ECO_REQUIRE(grid < 2147483647l, "winograd 3-D output transform: too many workgroups");
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 g((unsigned)grid), b((unsigned)s.thr_out);
#define ECO_W3OUT(V)                                                                                     \
  do {                                                                                                   \
    if (s.lds_out > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wino3_output_kernel<V>), "winograd 3-D output transform"); \
    hipLaunchKernelGGL((wino3_output_kernel<V>), g, b, s.lds_out, st_, a);                               \
  } while (0)
  if (vec == 4) ECO_W3OUT(4);
  else if (vec == 2) ECO_W3OUT(2);
  else ECO_W3OUT(1);
#undef ECO_W3OUT
  return check_launch("eco_wino3_output_forward");
}
