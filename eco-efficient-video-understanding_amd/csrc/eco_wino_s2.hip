// eco_wino_s2.hip -- the STRIDED 3x3x3 convolutions of the 3-D trunk (res4a_1 / res4a_down: stride 2, pad 1;
// models_ECO_Lite/kinetics/deploy.prototxt:1262-1330) as minimal-filtering problems on the transformed-domain GEMM.
//
// Rounds 1-5 ran these layers as a direct implicit GEMM (eco_conv.hip, conv_mfma_kernel: K = 27 cin, split-K with a
// reduce launch; 0.82 ms + 0.06 per layer at 32 clips, the largest kernel family of the round-5 step).  A stride-2
// correlation is the sum of EIGHT stride-1 correlations over the even / odd sub-lattices of its input (polyphase form):
//
//   out[o] = w0 x[2o-1] + w1 x[2o] + w2 x[2o+1] = (w0, w2) * x_odd[o-1 .. o]  +  (0, w1) * x_even[o-1 .. o]      per axis,
//
// i.e. per axis a 2-tap filter on each phase.  F(m, 2) evaluates m outputs of a 2-tap filter with m + 1 multiplies, and
// because every phase uses the SAME transform points, the eight phases add up in the transformed domain exactly like
// input channels do: one GEMM per point with K = 8 cin.  With F(4,2) over depth and F(7,2) over rows and columns the
// output volumes of res4a (8 x 14 x 14) tile without any overhang -- 5 x 8 x 8 = 320 points per 4 x 7 x 7 = 196 outputs,
// 320 * 8 / 196 = 13.1 multiplies per output and input channel instead of 27 -- and V / M are only 1.63x the input /
// output volume.  Measured (GEMM alone, 32 clips, K = 1024): res4a_1 and res4a_down as ONE problem of 512 output
// channels 0.66 ms at 131 TFLOP/s, against 2 x (0.82 + 0.06) ms on the direct kernel.  res5a (4 x 7 x 7 outputs: one
// tile per clip, 32 positions per point) would stream 1.3 GB of transformed weights per layer on this form (measured 0.68 ms
// per conv against 0.43 direct) and takes the 2-D form below instead.
//
// THE 2-D FORM (points = 64): where the output volume has too few 4 x 7 x 7 tiles for that (res5a: 4 x 7 x 7 outputs, one tile
// per clip -- 32 positions per point would stream 1.3 GB of transformed weights per layer) or there is no depth axis at all
// (ECO-Full's strided 2-D 3x3 convs: inception_3c / 4e, models_ECO_Full/kinetics/deploy.prototxt:1854-1990, 3420-3560), only
// rows and columns are transformed, F(7,2) x F(7,2) on the four (row, column) phases, and the depth taps stay direct -- they
// join the reduction: K = kz * 4 cin with kz = 3 (stride-2 depth, pad 1) or 1 (2-D), and every OUTPUT PLANE is a position:
//   V2[p][k/2][r][k%2]   p = ay*8 + ax,  k = ((c*kz + tz)*2 + fy)*2 + fx,  r = ((b*Do + od)*TH + th)*TW + tw,
//                        the (tz, od) entry transformed from input plane 2 od + tz - 1 (zeros outside the volume)
// 64 * 4 * kz / 49 = 15.7 (kz = 3) / 5.2 (kz = 1) multiplies per output and input channel instead of 27 / 9.
//
//   V[p][k/2][r][k%2]    p = (az*8 + ay)*8 + ax,  k = ((c*2 + fz)*2 + fy)*2 + fx  (f = 1: odd phase),
//                        r = ((b*TD + td)*TH + th)*TW + tw                       (the GEMM's kd = 1 layout, eco_wgemm.hip)
//   M[p][slice][cout][r]
//
// Transform constants: Cook-Toom on the points (0, 1, -1, 1/2, inf) for F(4,2) and (0, +-1, +-2, +-1/2, inf) for F(7,2)
// (the point set of F(6,3): B^T is that algorithm's); rows of B^T scaled to small rationals, the inverse scales folded
// into G on the host (float64).  The reference leaves the algorithm to cuDNN (cudnn_conv_layer.cu:15-65); results
// differ from the direct evaluation by fp32 rounding: ~1e-4 of the largest output per layer (eight-point transforms
// nested twice; tests: 5e-4 per layer, 1e-3 on logits).
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kS2P = 320;    // F(4,2) x F(7,2) x F(7,2): 5 x 8 x 8 points
constexpr int kS2P2 = 64;    // the 2-D form: F(7,2) x F(7,2)

// ---- 1-D transforms ------------------------------------------------------------------------------------------------
// F(4,2), points (0, 1, -1, 1/2, inf).  B^T rows scaled by (1, 2, 6, 3/8, 2):
//   [1 -2 -1 2 0; 0 -1 1 2 0; 0 -1 3 -2 0; 0 1 0 -1 0; 0 1 -2 -1 2]
__device__ __forceinline__ void s2_bt5(const float (&d)[5], float (&y)[5]) {
  const float a = d[2] - d[1], b = 2.0f * d[3];
  y[0] = d[0] - 2.0f * d[1] - d[2] + b;
  y[1] = a + b;
  y[2] = a + 2.0f * (d[2] - d[3]);
  y[3] = d[1] - d[3];
  y[4] = d[1] - 2.0f * d[2] - d[3] + 2.0f * d[4];
}
//   A^T = [1 1 1 1 0; 0 1 -1 1/2 0; 0 1 1 1/4 0; 0 1 -1 1/8 1]
__device__ __forceinline__ void s2_at4(const float (&m)[5], float (&o)[4]) {
  const float s = m[1] + m[2], t = m[1] - m[2];
  o[0] = m[0] + s + m[3];
  o[1] = t + 0.5f * m[3];
  o[2] = s + 0.25f * m[3];
  o[3] = t + 0.125f * m[3] + m[4];
}
// F(7,2), points (0, 1, -1, 2, -2, 1/2, -1/2, inf).  B^T = the B^T of F(6,3) (Lavin & Gray 2015):
//   [1 0 -21/4 0 21/4 0 -1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0;
//    0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
__device__ __forceinline__ void s2_bt8(const float (&d)[8], float (&y)[8]) {
  const float p1 = d[2] + d[6] - 4.25f * d[4], q1 = d[1] + d[5] - 4.25f * d[3];
  const float p2 = 0.25f * d[2] - 1.25f * d[4] + d[6], q2 = 0.5f * d[1] - 2.5f * d[3] + 2.0f * d[5];
  const float p3 = 4.0f * d[2] - 5.0f * d[4] + d[6], q3 = 2.0f * d[1] - 2.5f * d[3] + 0.5f * d[5];
  y[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
  y[1] = p1 + q1;
  y[2] = p1 - q1;
  y[3] = p2 + q2;
  y[4] = p2 - q2;
  y[5] = p3 + q3;
  y[6] = p3 - q3;
  y[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
}
//   A^T[o][a] = point_a ^ o (a < 7), column inf = e_6
__device__ __forceinline__ void s2_at7(const float (&m)[8], float (&o)[7]) {
  const float s1 = m[1] + m[2], t1 = m[1] - m[2];
  const float s2 = m[3] + m[4], t2 = m[3] - m[4];
  const float s3 = m[5] + m[6], t3 = m[5] - m[6];
  o[0] = m[0] + s1 + s2 + s3;
  o[1] = t1 + 2.0f * t2 + 0.5f * t3;
  o[2] = s1 + 4.0f * s2 + 0.25f * s3;
  o[3] = t1 + 8.0f * t2 + 0.125f * t3;
  o[4] = s1 + 16.0f * s2 + 0.0625f * s3;
  o[5] = t1 + 32.0f * t2 + 0.03125f * t3;
  o[6] = s1 + 64.0f * s2 + 0.015625f * s3 + m[7];
}

// ---- input transform -----------------------------------------------------------------------------------------------
struct S2InArgs {
  const float* x;   // [n][cin][D][H][W], D = 8 TD, H = 14 TH, W = 14 TW
  float* v;
  int n, cin, D, H, W, TD, TH, TW;
  int ntg;          // depth-tile groups per image (TDG tiles each)
  int PH, PW;       // parked plane: rows H + 2 (two zero rows above), columns W + 4 rounded up to 4 (four zero columns left: 16-byte stores)
  int Q;            // positions per k-pair row: n * TD * TH * TW
  long v_pstride;   // floats between points
};

// One workgroup per (input channel, image, group of TDG depth tiles).  Phase 1 walks the plane in 16-byte column groups:
// the 8 TDG + 2 input planes of the group are loaded once (coalesced along w), split into their even / odd depth phases
// and transformed along DEPTH in registers (5 -> 5 per phase and tile), parked in LDS as 10 TDG zero-bordered planes.
// Phase 2: one thread per (position, depth phase, depth point, row phase) reads the eight rows of its phase, 16
// consecutive columns each (both column phases), transforms along w and then along h, and writes its 8 x 8 points for
// the column-phase PAIR as 8-byte stores -- k is ordered so that the two column phases are the GEMM's k-pair.
template <int TDG, int VEC>
__global__ __launch_bounds__(256) void wino_s2_input_kernel(const S2InArgs a) {
  ECO_DYNAMIC_LDS(zd);   // [TDG][2 fz][5 az][PH][PW]
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  // each XCD takes a contiguous range of (channel, image, group): the 64-byte runs neighbouring workgroups write per
  // (point, k-pair) meet in one L2
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tg = wg % a.ntg, t0 = wg / a.ntg;
  const int b = t0 % a.n, c = t0 / a.n;
  const int td0 = tg * TDG;
  const int plane = a.PH * a.PW;
  const long hw = (long)a.H * a.W;
  constexpr int NPL = 8 * TDG + 2;

  // ---- phase 1 ----
  const int pwv = a.PW / 4;
  const float* const xc = a.x + ((long)b * a.cin + c) * a.D * hw;
  for (int s = tid; s < a.PH * pwv; s += nthr) {
    const int pv = s % pwv, ph = s / pwv;
    const int h = ph - 2, w0 = 4 * pv - 4;
    float xin[NPL][4];
#pragma unroll
    for (int k = 0; k < NPL; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) xin[k][j] = 0.0f;
    if (h >= 0 && w0 >= 0 && w0 < a.W) {
      const float* xp = xc + (long)h * a.W + w0;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        const int d = 8 * td0 - 2 + k;                     // workgroup-uniform
        if ((unsigned)d >= (unsigned)a.D) continue;
        const float* q = xp + (long)d * hw;
        if (VEC == 4) {
          const float4 v4 = ld((const float4*)q);
          xin[k][0] = v4.x; xin[k][1] = v4.y; xin[k][2] = v4.z; xin[k][3] = v4.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (w0 + j < a.W) xin[k][j] = ld(q + j);
        }
      }
    }
    float* dst = zd + ph * a.PW + 4 * pv;
#pragma unroll
    for (int tl = 0; tl < TDG; ++tl)
#pragma unroll
      for (int fz = 0; fz < 2; ++fz) {
        float4 out[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float col[5] = {xin[8 * tl + fz][j], xin[8 * tl + 2 + fz][j], xin[8 * tl + 4 + fz][j], xin[8 * tl + 6 + fz][j],
                                xin[8 * tl + 8 + fz][j]};
          float y[5];
          s2_bt5(col, y);
#pragma unroll
          for (int az = 0; az < 5; ++az) ((float*)&out[az])[j] = y[az];
        }
#pragma unroll
        for (int az = 0; az < 5; ++az) *(float4*)(dst + ((tl * 2 + fz) * 5 + az) * plane) = out[az];
      }
  }
  __syncthreads();

  // ---- phase 2 ----
  const int tpp = a.TH * a.TW, npos = TDG * tpp;
  for (int it = tid; it < 20 * npos; it += nthr) {
    const int pos = it % npos, combo = it / npos;
    const int tl = pos / tpp, tt = pos - tl * tpp;
    const int th = tt / a.TW, tw = tt - th * a.TW;
    const int fy = combo & 1, za = combo >> 1;            // za = fz * 5 + az
    const int fz = za / 5, az = za - fz * 5;
    const int td = td0 + tl;
    if (td >= a.TD) continue;
    const float* src = zd + ((tl * 2 + fz) * 5 + az) * plane + (14 * th + fy) * a.PW + 14 * tw + 2;
    float t[8][2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float row[16];
      const float2* rp = (const float2*)(src + 2 * i * a.PW);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float2 q = rp[j];
        row[2 * j] = q.x;
        row[2 * j + 1] = q.y;
      }
#pragma unroll
      for (int fx = 0; fx < 2; ++fx) {
        const float dd[8] = {row[fx], row[2 + fx], row[4 + fx], row[6 + fx], row[8 + fx], row[10 + fx], row[12 + fx], row[14 + fx]};
        s2_bt8(dd, t[i][fx]);
      }
    }
    const long r = (long)(b * a.TD + td) * tpp + tt;
    const long kp = ((long)c * 2 + fz) * 2 + fy;
    float* vo = a.v + (long)az * 64 * a.v_pstride + (kp * a.Q + r) * 2;
#pragma unroll
    for (int ax = 0; ax < 8; ++ax) {
      float y0[8], y1[8];
      {
        const float c0[8] = {t[0][0][ax], t[1][0][ax], t[2][0][ax], t[3][0][ax], t[4][0][ax], t[5][0][ax], t[6][0][ax], t[7][0][ax]};
        s2_bt8(c0, y0);
        const float c1[8] = {t[0][1][ax], t[1][1][ax], t[2][1][ax], t[3][1][ax], t[4][1][ax], t[5][1][ax], t[6][1][ax], t[7][1][ax]};
        s2_bt8(c1, y1);
      }
#pragma unroll
      for (int ay = 0; ay < 8; ++ay) st((float2*)(vo + (long)(ay * 8 + ax) * a.v_pstride), make_float2(y0[ay], y1[ay]));
    }
  }
}

// ---- output transform ----------------------------------------------------------------------------------------------
struct S2OutArgs {
  const float* m;   // M[320][ksplit][ctot][Q]
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;
  int relu;
  int n, c0, cout, Do, Ho, Wo, TD, TH, TW;   // this member's channels [c0, c0 + cout) of the GEMM's ctot rows
  int GB, nbg, Q, ksplit;
  long m_pstride;   // floats between points = ksplit * ctot * Q
  long m_sstride;   // floats between split-K slices = ctot * Q
};

// One workgroup per (output channel, group of GB images).  Phase A: one thread per (depth point, position) loads its
// 8 x 8 products (each a contiguous run across the wave), sums the split-K slices, applies the 2-D output transform
// (8 x 8 -> 7 x 7) and parks the tile in LDS as part of a depth-point plane.  Phase B: one thread per (image, depth
// tile, plane position) folds the five depth points into four output planes and applies the fused epilogue (bias,
// Eltwise residual, raw store, folded BN, ReLU, both activated destinations; strided views); lanes walk a plane in
// memory order.
// ZT = false: the 2-D form -- no depth points (a.TD counts OUTPUT PLANES, each its own position), phase B is the epilogue alone.
template <bool ZT>
__global__ __launch_bounds__(256) void wino_s2_output_kernel(const S2OutArgs a) {
  constexpr int NZ = ZT ? 5 : 1;
  ECO_DYNAMIC_LDS(sp);   // [NZ az][GB * TD][Ho * Wo]
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bg = wg % a.nbg, ch = wg / a.nbg;
  const int b0 = bg * a.GB;
  const int tpp = a.TH * a.TW, tpi = a.TD * tpp;
  const int gb = a.n - b0 < a.GB ? a.n - b0 : a.GB;       // images of this group
  const int npos = gb * tpi;
  const int S = a.Ho * a.Wo;
  const int npl = a.GB * a.TD;

  // ---- phase A ----
  for (int it = tid; it < NZ * npos; it += nthr) {
    const int pos = it % npos, az = it / npos;
    const int bl = pos / tpi, rem = pos - bl * tpi;
    const int td = rem / tpp, tt = rem - td * tpp;
    const int th = tt / a.TW, tw = tt - th * a.TW;
    const float* mp = a.m + (long)az * 64 * a.m_pstride + (long)(a.c0 + ch) * a.Q + (long)b0 * tpi + pos;
    float mm[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) mm[i][j] = ld(mp + (long)(8 * i + j) * a.m_pstride);
    for (int sl = 1; sl < a.ksplit; ++sl) {
      const float* ms = mp + (long)sl * a.m_sstride;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) mm[i][j] += ld(ms + (long)(8 * i + j) * a.m_pstride);
    }
    float t[8][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s2_at7(mm[i], t[i]);     // along w
    float* dst = sp + ((long)az * npl + bl * a.TD + td) * S + (7 * th) * a.Wo + 7 * tw;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float col[8] = {t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j], t[6][j], t[7][j]};
      float y[7];
      s2_at7(col, y);                                   // along h
#pragma unroll
      for (int i = 0; i < 7; ++i) dst[i * a.Wo + j] = y[i];
    }
  }
  __syncthreads();

  // ---- phase B ----
  const float bias = a.bias ? ld(a.bias + ch) : 0.0f;
  const float sc = a.bn_scale ? ld(a.bn_scale + ch) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + ch) : 0.0f;
  const long az_stride = (long)npl * S;
  for (int it = tid; it < gb * a.TD * S; it += nthr) {
    const int s = it % S, pl = it / S;
    const int bl = pl / a.TD, td = pl - bl * a.TD;
    const int b = b0 + bl;
    const float* src = sp + (long)pl * S + s;
    constexpr int NO = ZT ? 4 : 1;                      // output planes per item
    float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ZT) {
      const float mz[5] = {src[0], src[az_stride], src[2 * az_stride], src[3 * az_stride], src[4 * az_stride]};
      s2_at4(mz, y);
    } else {
      y[0] = src[0];
    }
    const long spo = (long)(NO * td) * S + s;
    const long rb = a.residual.ptr ? view_base(a.residual, b, 0) + (long)ch * a.residual.stride_c + spo : 0;
    float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.residual.ptr) {
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (NO * td + o < a.Do) res[o] = ld(a.residual.ptr + rb + (long)o * S);
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) y[o] += bias + res[o];
    if (a.raw.ptr) {
      float* op = a.raw.ptr + view_base(a.raw, b, 0) + (long)ch * a.raw.stride_c + spo;
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (NO * td + o < a.Do) st(op + (long)o * S, y[o]);
    }
    if (a.act.ptr) {
      float* op = a.act.ptr + view_base(a.act, b, 0) + (long)ch * a.act.stride_c + spo;
      float* op2 = a.act2.ptr ? a.act2.ptr + view_base(a.act2, b, 0) + (long)ch * a.act2.stride_c + spo : nullptr;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        if (NO * td + o >= a.Do) continue;
        const float v = y[o] * sc + sh;
        const float z = a.relu ? fmaxf(v, 0.0f) : v;
        st(op + (long)o * S, z);
        if (op2) st(op2 + (long)o * S, z);
      }
    }
  }
}

// ---- the 2-D form's input transform ---------------------------------------------------------------------------------
struct S2dInArgs {
  const float* x;   // [n][cin][D][H][W]; D = 2 Do (kz = 3) or Do (kz = 1; D = 1 for a 2-D blob), H = 14 TH, W = 14 TW
  float* v;
  int n, cin, D, H, W, Do, TH, TW, KZ;
  int GB, nbg;      // images per workgroup, image groups
  int PH, PW;       // parked plane: rows H + 2, columns W + 4 rounded up to 4, zero borders above / left
  int Q;            // positions per k-pair row: n * Do * TH * TW
  long v_pstride;
};

// One workgroup per (input channel, group of GB images): the channel's planes of the group are parked in LDS with zero
// borders (phase 1, 16-byte column groups), then one thread per (position = (image, output plane, tile), depth tap, row
// phase) transforms its 8 rows x 16 columns along w and h and writes 8 x 8 points for the column-phase pair (phase 2), as
// in the 3-D form.  A depth tap that falls outside the volume (plane -1) writes zeros: V is a dense GEMM operand.
template <int VEC>
__global__ __launch_bounds__(256) void wino_s2d_input_kernel(const S2dInArgs a) {
  ECO_DYNAMIC_LDS(zd);   // [GB][D][PH][PW]
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bg = wg % a.nbg, c = wg / a.nbg;
  const int b0 = bg * a.GB;
  const int gb = a.n - b0 < a.GB ? a.n - b0 : a.GB;
  const int plane = a.PH * a.PW;
  const long hw = (long)a.H * a.W;

  // ---- phase 1 ----  (eight slots per thread and pass: the loads of a pass are all in flight before the first LDS store -- one
  // load per loop iteration left every iteration a memory round trip of its own: res5a 0.064 ms for 152 MB)
  const int pwv = a.PW / 4;
  const int slots = gb * a.D * a.PH * pwv;
  constexpr int U = 8;
  for (int s0 = tid; s0 < slots; s0 += nthr * U) {
    float4 q[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u * nthr;
      q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      off[u] = -1;
      if (s >= slots) continue;
      const int pv = s % pwv;
      int t = s / pwv;
      const int ph = t % a.PH;
      t /= a.PH;                                            // = bl * D + d
      const int bl = t / a.D, d = t - bl * a.D;
      const int h = ph - 2, w0 = 4 * pv - 4;
      off[u] = t * plane + ph * a.PW + 4 * pv;
      if (h >= 0 && w0 >= 0 && w0 < a.W) {
        const float* xp = a.x + (((long)(b0 + bl) * a.cin + c) * a.D + d) * hw + (long)h * a.W + w0;
        if (VEC == 4) {
          q[u] = ld((const float4*)xp);
        } else {
          q[u].x = ld(xp);
          if (w0 + 1 < a.W) q[u].y = ld(xp + 1);
          if (w0 + 2 < a.W) q[u].z = ld(xp + 2);
          if (w0 + 3 < a.W) q[u].w = ld(xp + 3);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) *(float4*)(zd + off[u]) = q[u];
  }
  __syncthreads();

  // ---- phase 2 ----
  const int tpp = a.TH * a.TW, tpi = a.Do * tpp, npos = gb * tpi;
  for (int it = tid; it < 2 * a.KZ * npos; it += nthr) {
    const int pos = it % npos, combo = it / npos;
    const int bl = pos / tpi, rem = pos - bl * tpi;
    const int od = rem / tpp, tt = rem - od * tpp;
    const int th = tt / a.TW, tw = tt - th * a.TW;
    const int fy = combo & 1, tz = combo >> 1;
    const int d = a.KZ == 3 ? 2 * od + tz - 1 : od;
    const long r = (long)(b0 + bl) * tpi + rem;
    const long kp = ((long)c * a.KZ + tz) * 2 + fy;
    float* vo = a.v + (kp * a.Q + r) * 2;
    if (d < 0) {                                          // (d <= D - 1 always: D = 2 Do)
#pragma unroll 8
      for (int p = 0; p < 64; ++p) st((float2*)(vo + (long)p * a.v_pstride), make_float2(0.0f, 0.0f));
      continue;
    }
    const float* src = zd + ((long)bl * a.D + d) * plane + (14 * th + fy) * a.PW + 14 * tw + 2;
    float t[8][2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float row[16];
      const float2* rp = (const float2*)(src + 2 * i * a.PW);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float2 q = rp[j];
        row[2 * j] = q.x;
        row[2 * j + 1] = q.y;
      }
#pragma unroll
      for (int fx = 0; fx < 2; ++fx) {
        const float dd[8] = {row[fx], row[2 + fx], row[4 + fx], row[6 + fx], row[8 + fx], row[10 + fx], row[12 + fx], row[14 + fx]};
        s2_bt8(dd, t[i][fx]);
      }
    }
#pragma unroll
    for (int ax = 0; ax < 8; ++ax) {
      float y0[8], y1[8];
      {
        const float c0[8] = {t[0][0][ax], t[1][0][ax], t[2][0][ax], t[3][0][ax], t[4][0][ax], t[5][0][ax], t[6][0][ax], t[7][0][ax]};
        s2_bt8(c0, y0);
        const float c1[8] = {t[0][1][ax], t[1][1][ax], t[2][1][ax], t[3][1][ax], t[4][1][ax], t[5][1][ax], t[6][1][ax], t[7][1][ax]};
        s2_bt8(c1, y1);
      }
#pragma unroll
      for (int ay = 0; ay < 8; ++ay) st((float2*)(vo + (long)(ay * 8 + ax) * a.v_pstride), make_float2(y0[ay], y1[ay]));
    }
  }
}

// G of the two algorithms with the row scales of the B^T forms above folded in (u = G g for a 2-tap filter g).
static const double kS2G5[5][2] = {{1.0, 0.0}, {0.5, 0.5}, {1.0 / 6, -1.0 / 6}, {8.0 / 3, 4.0 / 3}, {0.0, 0.5}};
static const double kS2G8[8][2] = {{1.0, 0.0},          {-2.0 / 9, -2.0 / 9}, {-2.0 / 9, 2.0 / 9},   {1.0 / 90, 2.0 / 90},
                                   {1.0 / 90, -2.0 / 90}, {32.0 / 45, 16.0 / 45}, {32.0 / 45, -16.0 / 45}, {0.0, 1.0}};

struct S2Shape { int TDG, ntg, PH, PW; size_t lds_in; int GB, nbg; size_t lds_out; };
static S2Shape s2_shape(int n, int td, int th, int tw) {
  S2Shape s;
  s.TDG = td >= 2 ? 2 : 1;
  s.ntg = (int)ceil_div(td, s.TDG);
  s.PH = 14 * th + 2;
  s.PW = (14 * tw + 4 + 3) / 4 * 4;
  s.lds_in = (size_t)s.TDG * 10 * s.PH * s.PW * 4;
  // images per workgroup of the output transform: at least 32 positions per (point, channel) run, LDS permitting
  const int tpi = td * th * tw;
  const size_t per_image = (size_t)5 * td * 49 * th * tw * 4;
  int gb = (int)ceil_div(32, tpi);
  while (gb > 1 && gb * per_image > (size_t)64 * 1024) --gb;
  if (gb > n) gb = n;
  if (gb < 1) gb = 1;
  s.GB = gb;
  s.nbg = (int)ceil_div(n, gb);
  s.lds_out = gb * per_image;
  return s;
}

struct S2dShape { int GB, nbg, PH, PW; size_t lds_in; int GBo, nbgo; size_t lds_out; };
static S2dShape s2d_shape(int n, int kz, int od, int th, int tw) {
  S2dShape s;
  const int D = kz == 3 ? 2 * od : od;
  s.PH = 14 * th + 2;
  s.PW = (14 * tw + 4 + 3) / 4 * 4;
  const int tpi = od * th * tw;
  const size_t in_image = (size_t)D * s.PH * s.PW * 4, out_image = (size_t)od * 49 * th * tw * 4;
  auto group = [&](size_t per_image, size_t budget) {
    int gb = (int)ceil_div(32, tpi);                      // at least 32 positions per (point, k-pair / channel) run
    while (gb > 1 && gb * per_image > budget) --gb;
    if (gb > n) gb = n;
    return gb < 1 ? 1 : gb;
  };
  s.GB = group(in_image, 80 * 1024);
  s.nbg = (int)ceil_div(n, s.GB);
  s.lds_in = s.GB * in_image;
  s.GBo = group(out_image, 64 * 1024);
  s.nbgo = (int)ceil_div(n, s.GBo);
  s.lds_out = s.GBo * out_image;
  return s;
}

}  // namespace eco

using namespace eco;

// plan: eco_wgemm_plan_create(n, 8 * cin, ctot, TD, TH, TW, kd = 1, points = 320); (d, h, w) = the OUTPUT volume
static int s2_check_plan(const eco_wgemm_plan* p, int32_t od, int32_t oh, int32_t ow, const char* who) {
  ECO_REQUIRE(p != nullptr, "%s: null plan", who);
  ECO_REQUIRE(p->points == kS2P && p->kd == 1, "%s: needs a stride-2 polyphase plan (points = 320, kd = 1), got points=%d kd=%d", who,
              p->points, p->kd);
  ECO_REQUIRE(od > 0 && oh > 0 && ow > 0 && od % 4 == 0 && oh % 7 == 0 && ow % 7 == 0,
              "%s: the output volume %dx%dx%d must tile by 4x7x7", who, od, oh, ow);
  ECO_REQUIRE(p->d == od / 4 && p->th == oh / 7 && p->tw == ow / 7, "%s: plan is for %dx%dx%d tiles, output volume %dx%dx%d needs %dx%dx%d",
              who, p->d, p->th, p->tw, od, oh, ow, od / 4, oh / 7, ow / 7);
  ECO_REQUIRE(p->n > 0 && p->cin > 0 && p->cin % 16 == 0 && p->cout > 0, "%s: bad plan", who);
  return ECO_OK;
}

extern "C" int64_t eco_wino_s2_lds_bytes(int32_t n, int32_t td, int32_t th, int32_t tw) {
  if (n <= 0 || td <= 0 || th <= 0 || tw <= 0) return -1;
  const S2Shape s = s2_shape(n, td, th, tw);
  return (int64_t)(s.lds_in > s.lds_out ? s.lds_in : s.lds_out);
}

extern "C" int eco_wino_s2_weight_transform(const float* w, int32_t cout, int32_t cin, float* u) {
  clear_error();
  ECO_REQUIRE(w && u && cout > 0 && cin > 0, "stride-2 winograd weights: bad argument");
  // u[p][co][k] = (G5 (x) G8 (x) G8) g_f,  k = ((ci*2 + fz)*2 + fy)*2 + fx,  per axis g_1 = (w0, w2), g_0 = (0, w1)
  const long K = 8L * cin, plane = (long)cout * K;
  for (long co = 0; co < cout; ++co)
    for (long ci = 0; ci < cin; ++ci) {
      const float* g = w + (co * cin + ci) * 27;
      for (int f = 0; f < 8; ++f) {
        const int fz = f >> 2, fy = (f >> 1) & 1, fx = f & 1;
        double g2[2][2][2];
        for (int tz = 0; tz < 2; ++tz)
          for (int ty = 0; ty < 2; ++ty)
            for (int tx = 0; tx < 2; ++tx) {
              // tap index on the 3-tap axis: odd phase (w0, w2), even phase (absent, w1)
              const int kz = fz ? 2 * tz : (tz ? 1 : -1), ky = fy ? 2 * ty : (ty ? 1 : -1), kx = fx ? 2 * tx : (tx ? 1 : -1);
              g2[tz][ty][tx] = (kz < 0 || ky < 0 || kx < 0) ? 0.0 : (double)g[(kz * 3 + ky) * 3 + kx];
            }
        double t1[2][2][8];   // along x
        for (int tz = 0; tz < 2; ++tz)
          for (int ty = 0; ty < 2; ++ty)
            for (int ax = 0; ax < 8; ++ax) t1[tz][ty][ax] = kS2G8[ax][0] * g2[tz][ty][0] + kS2G8[ax][1] * g2[tz][ty][1];
        double t2[2][8][8];   // along y
        for (int tz = 0; tz < 2; ++tz)
          for (int ay = 0; ay < 8; ++ay)
            for (int ax = 0; ax < 8; ++ax) t2[tz][ay][ax] = kS2G8[ay][0] * t1[tz][0][ax] + kS2G8[ay][1] * t1[tz][1][ax];
        const long k = ci * 8 + f;
        for (int az = 0; az < 5; ++az)
          for (int ay = 0; ay < 8; ++ay)
            for (int ax = 0; ax < 8; ++ax)
              u[(long)((az * 8 + ay) * 8 + ax) * plane + co * K + k] =
                  (float)(kS2G5[az][0] * t2[0][ay][ax] + kS2G5[az][1] * t2[1][ay][ax]);
      }
    }
  return ECO_OK;
}

extern "C" int eco_wino_s2_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t d, int32_t h, int32_t w,
                                         void* stream) {
  clear_error();
  ECO_REQUIRE(d > 0 && h > 0 && w > 0 && d % 2 == 0 && h % 2 == 0 && w % 2 == 0,
              "stride-2 winograd input transform: the input volume %dx%dx%d must have even extents", d, h, w);
  if (int rc = s2_check_plan(plan, d / 2, h / 2, w / 2, "stride-2 winograd input transform")) return rc;
  ECO_REQUIRE(x && v, "stride-2 winograd input transform: null argument");
  ECO_REQUIRE(plan->cin % 8 == 0, "stride-2 winograd input transform: the plan's cin must be 8 x the layer's");
  ECO_REQUIRE(((uintptr_t)v & 7) == 0, "stride-2 winograd input transform: v must be 8-byte aligned");
  const S2Shape s = s2_shape(plan->n, plan->d, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_in <= (size_t)kEcoMaxDynamicLds, "stride-2 winograd input transform: %dx%d planes need %zu bytes of LDS (max %d)", h,
              w, s.lds_in, kEcoMaxDynamicLds);
  S2InArgs a;
  a.x = x; a.v = v; a.n = plan->n; a.cin = plan->cin / 8; a.D = d; a.H = h; a.W = w;
  a.TD = plan->d; a.TH = plan->th; a.TW = plan->tw;
  a.ntg = s.ntg; a.PH = s.PH; a.PW = s.PW;
  a.Q = (int)plan->q;
  a.v_pstride = (long)(plan->cin / 2) * plan->q * 2;
  const long grid = (long)a.cin * a.n * a.ntg;
  ECO_REQUIRE(grid < 2147483647l, "stride-2 winograd input transform: too many workgroups");
  const bool vec4 = w % 4 == 0 && ((uintptr_t)x & 15) == 0;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 g((unsigned)grid), b(256);
#define ECO_S2IN(T, V)                                                                                         \
  do {                                                                                                         \
    if (s.lds_in > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wino_s2_input_kernel<T, V>), "stride-2 winograd input transform"); \
    hipLaunchKernelGGL((wino_s2_input_kernel<T, V>), g, b, s.lds_in, st_, a);                                  \
  } while (0)
  if (s.TDG == 2) {
    if (vec4) ECO_S2IN(2, 4);
    else ECO_S2IN(2, 1);
  } else {
    if (vec4) ECO_S2IN(1, 4);
    else ECO_S2IN(1, 1);
  }
#undef ECO_S2IN
  return check_launch("eco_wino_s2_input_forward");
}

extern "C" int eco_wino_s2_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t c0, int32_t cout, int32_t od,
                                          int32_t oh, int32_t ow, const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  if (int rc = s2_check_plan(plan, od, oh, ow, "stride-2 winograd output transform")) return rc;
  ECO_REQUIRE(m && ep, "stride-2 winograd output transform: null argument");
  ECO_REQUIRE(c0 >= 0 && cout > 0 && c0 + cout <= plan->cout, "stride-2 winograd output transform: channels [%d, %d) of the plan's %d",
              c0, c0 + cout, plan->cout);
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "stride-2 winograd output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "stride-2 winograd output transform: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "stride-2 winograd output transform: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "stride-2 winograd output transform: one launch per member (c0, cout) instead of a segmented epilogue");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "stride-2 winograd output transform: view needs t >= 1 and stride_c >= 1");
  const S2Shape s = s2_shape(plan->n, plan->d, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_out <= (size_t)kEcoMaxDynamicLds, "stride-2 winograd output transform: %dx%d planes need %zu bytes of LDS (max %d)",
              oh, ow, s.lds_out, kEcoMaxDynamicLds);
  S2OutArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.n = plan->n; a.c0 = c0; a.cout = cout; a.Do = od; a.Ho = oh; a.Wo = ow; a.TD = plan->d; a.TH = plan->th; a.TW = plan->tw;
  a.GB = s.GB; a.nbg = s.nbg; a.Q = (int)plan->q; a.ksplit = plan->ksplit;
  a.m_sstride = (long)plan->cout * plan->q;
  a.m_pstride = (long)plan->ksplit * a.m_sstride;
  const long grid = (long)cout * a.nbg;
  ECO_REQUIRE(grid < 2147483647l, "stride-2 winograd output transform: too many workgroups");
  if (s.lds_out > 64 * 1024) ECO_RAISE_DYNAMIC_LDS(wino_s2_output_kernel<true>, "stride-2 winograd output transform");
  hipLaunchKernelGGL((wino_s2_output_kernel<true>), dim3((unsigned)grid), dim3(256), s.lds_out, (hipStream_t)stream, a);
  return check_launch("eco_wino_s2_output_forward");
}

// ---- the 2-D form: plan = eco_wgemm_plan_create(n, 4 * kz * cin, ctot, Do, Ho / 7, Wo / 7, kd = 1, points = 64) ------------
static int s2d_check_plan(const eco_wgemm_plan* p, int32_t kz, int32_t od, int32_t oh, int32_t ow, const char* who) {
  ECO_REQUIRE(p != nullptr, "%s: null plan", who);
  ECO_REQUIRE(p->points == kS2P2 && p->kd == 1, "%s: needs a 2-D stride-2 polyphase plan (points = 64, kd = 1), got points=%d kd=%d", who,
              p->points, p->kd);
  ECO_REQUIRE(kz == 1 || kz == 3, "%s: kz must be 1 (no depth taps) or 3 (stride-2 depth taps), got %d", who, kz);
  ECO_REQUIRE(od > 0 && oh > 0 && ow > 0 && oh % 7 == 0 && ow % 7 == 0, "%s: the output planes %dx%d must tile by 7x7", who, oh, ow);
  ECO_REQUIRE(p->d == od && p->th == oh / 7 && p->tw == ow / 7, "%s: plan is for %d planes of %dx%d tiles, output volume %dx%dx%d needs %d of %dx%d",
              who, p->d, p->th, p->tw, od, oh, ow, od, oh / 7, ow / 7);
  ECO_REQUIRE(p->n > 0 && p->cin > 0 && p->cin % 16 == 0 && p->cin % (4 * kz) == 0 && p->cout > 0, "%s: bad plan", who);
  return ECO_OK;
}

extern "C" int64_t eco_wino_s2d_lds_bytes(int32_t n, int32_t kz, int32_t od, int32_t th, int32_t tw) {
  if (n <= 0 || (kz != 1 && kz != 3) || od <= 0 || th <= 0 || tw <= 0) return -1;
  const S2dShape s = s2d_shape(n, kz, od, th, tw);
  return (int64_t)(s.lds_in > s.lds_out ? s.lds_in : s.lds_out);
}

extern "C" int eco_wino_s2d_weight_transform(const float* w, int32_t cout, int32_t cin, int32_t kz, float* u) {
  clear_error();
  ECO_REQUIRE(w && u && cout > 0 && cin > 0 && (kz == 1 || kz == 3), "2-D stride-2 winograd weights: bad argument");
  // u[p][co][k] = (G8 (x) G8) g_f,  k = ((ci*kz + tz)*2 + fy)*2 + fx,  g = w[co][ci][tz], per axis g_1 = (w0, w2), g_0 = (0, w1)
  const long K = 4L * kz * cin, plane = (long)cout * K;
  for (long co = 0; co < cout; ++co)
    for (long ci = 0; ci < cin; ++ci)
      for (int tz = 0; tz < kz; ++tz) {
        const float* g = w + ((co * cin + ci) * kz + tz) * 9;
        for (int f = 0; f < 4; ++f) {
          const int fy = f >> 1, fx = f & 1;
          double g2[2][2];
          for (int ty = 0; ty < 2; ++ty)
            for (int tx = 0; tx < 2; ++tx) {
              const int ky = fy ? 2 * ty : (ty ? 1 : -1), kx = fx ? 2 * tx : (tx ? 1 : -1);
              g2[ty][tx] = (ky < 0 || kx < 0) ? 0.0 : (double)g[ky * 3 + kx];
            }
          const long k = ((ci * kz + tz) * 2 + fy) * 2 + fx;
          for (int ay = 0; ay < 8; ++ay)
            for (int ax = 0; ax < 8; ++ax) {
              double acc = 0.0;
              for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) acc += kS2G8[ay][ty] * kS2G8[ax][tx] * g2[ty][tx];
              u[(long)(ay * 8 + ax) * plane + co * K + k] = (float)acc;
            }
        }
      }
  return ECO_OK;
}

extern "C" int eco_wino_s2d_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t kz, int32_t d, int32_t h,
                                          int32_t w, void* stream) {
  clear_error();
  ECO_REQUIRE(d > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && (kz != 3 || d % 2 == 0),
              "2-D stride-2 winograd input transform: the input volume %dx%dx%d must have even extents", d, h, w);
  const int od = kz == 3 ? d / 2 : d;
  if (int rc = s2d_check_plan(plan, kz, od, h / 2, w / 2, "2-D stride-2 winograd input transform")) return rc;
  ECO_REQUIRE(x && v, "2-D stride-2 winograd input transform: null argument");
  ECO_REQUIRE(((uintptr_t)v & 7) == 0, "2-D stride-2 winograd input transform: v must be 8-byte aligned");
  const S2dShape s = s2d_shape(plan->n, kz, od, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_in <= (size_t)kEcoMaxDynamicLds, "2-D stride-2 winograd input transform: %dx%dx%d volumes need %zu bytes of LDS (max %d)",
              d, h, w, s.lds_in, kEcoMaxDynamicLds);
  S2dInArgs a;
  a.x = x; a.v = v; a.n = plan->n; a.cin = plan->cin / (4 * kz); a.D = d; a.H = h; a.W = w; a.Do = od;
  a.TH = plan->th; a.TW = plan->tw; a.KZ = kz;
  a.GB = s.GB; a.nbg = s.nbg; a.PH = s.PH; a.PW = s.PW;
  a.Q = (int)plan->q;
  a.v_pstride = (long)(plan->cin / 2) * plan->q * 2;
  const long grid = (long)a.cin * a.nbg;
  ECO_REQUIRE(grid < 2147483647l, "2-D stride-2 winograd input transform: too many workgroups");
  const bool vec4 = w % 4 == 0 && ((uintptr_t)x & 15) == 0;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 g((unsigned)grid), b(256);
  if (vec4) {
    if (s.lds_in > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wino_s2d_input_kernel<4>), "2-D stride-2 winograd input transform");
    hipLaunchKernelGGL((wino_s2d_input_kernel<4>), g, b, s.lds_in, st_, a);
  } else {
    if (s.lds_in > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wino_s2d_input_kernel<1>), "2-D stride-2 winograd input transform");
    hipLaunchKernelGGL((wino_s2d_input_kernel<1>), g, b, s.lds_in, st_, a);
  }
  return check_launch("eco_wino_s2d_input_forward");
}

extern "C" int eco_wino_s2d_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t c0, int32_t cout, int32_t od,
                                           int32_t oh, int32_t ow, const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  ECO_REQUIRE(plan != nullptr, "2-D stride-2 winograd output transform: null plan");
  // (kz does not matter on this side: points and tiles are what is checked)
  if (int rc = s2d_check_plan(plan, 1, od, oh, ow, "2-D stride-2 winograd output transform")) return rc;
  ECO_REQUIRE(m && ep, "2-D stride-2 winograd output transform: null argument");
  ECO_REQUIRE(c0 >= 0 && cout > 0 && c0 + cout <= plan->cout, "2-D stride-2 winograd output transform: channels [%d, %d) of the plan's %d",
              c0, c0 + cout, plan->cout);
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "2-D stride-2 winograd output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "2-D stride-2 winograd output transform: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "2-D stride-2 winograd output transform: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "2-D stride-2 winograd output transform: one launch per member (c0, cout) instead of a segmented epilogue");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "2-D stride-2 winograd output transform: view needs t >= 1 and stride_c >= 1");
  const S2dShape s = s2d_shape(plan->n, 1, od, plan->th, plan->tw);
  ECO_REQUIRE(s.lds_out <= (size_t)kEcoMaxDynamicLds, "2-D stride-2 winograd output transform: %dx%dx%d volumes need %zu bytes of LDS (max %d)",
              od, oh, ow, s.lds_out, kEcoMaxDynamicLds);
  S2OutArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.n = plan->n; a.c0 = c0; a.cout = cout; a.Do = od; a.Ho = oh; a.Wo = ow; a.TD = od; a.TH = plan->th; a.TW = plan->tw;
  a.GB = s.GBo; a.nbg = s.nbgo; a.Q = (int)plan->q; a.ksplit = plan->ksplit;
  a.m_sstride = (long)plan->cout * plan->q;
  a.m_pstride = (long)plan->ksplit * a.m_sstride;
  const long grid = (long)cout * a.nbg;
  ECO_REQUIRE(grid < 2147483647l, "2-D stride-2 winograd output transform: too many workgroups");
  if (s.lds_out > 64 * 1024) ECO_RAISE_DYNAMIC_LDS(wino_s2_output_kernel<false>, "2-D stride-2 winograd output transform");
  hipLaunchKernelGGL((wino_s2_output_kernel<false>), dim3((unsigned)grid), dim3(256), s.lds_out, (hipStream_t)stream, a);
  return check_launch("eco_wino_s2d_output_forward");
}
