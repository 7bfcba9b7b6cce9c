// eco_wgemm.hip -- the transformed-domain GEMMs of the Winograd F(4x4,3x3) route as a dedicated dense kernel.
//
// Round 1 ran the 36 point-convolutions M_p = U_p (*) V_p through the general gather kernel of eco_conv.hip
// (per-position address decode, tap masks, 4-byte gathers, VGPR staging): 0.58-0.68 of the fp32 MFMA peak on the
// 3-D trunk, 0.40-0.48 of their floor on the short-K 2-D layers -- 7.6 ms of an 18.1 ms step.  Between an input
// transform and an output transform that this file owns as well, nothing forces that generality: the layouts of
// V and M are free.  So
//
//   V[p][c/2][d'][r][c%2]     fp32, r = (b*TH + th)*TW + tw, d' = d + pd with an all-zero plane at either end
//                             (kd = 3): channel PAIRS interleaved (one 8-byte LDS read feeds two MFMA k-steps),
//                             positions depth-major, so that depth tap dz of channel pair cp is the SAME row
//                             shifted by dz*NB positions -- a dense, contiguous operand with no padding logic
//   U[p][mblock][stage][8][BMP][2]   stage = (16 input channels, depth tap); a stage's weights are one contiguous
//                             block in HBM and in LDS
//   M[p][slice][cout][d][r]   raw products, one row per output channel (lane = position: 128-byte stores)
//
// and the kernel is a plain batched GEMM: both operands go global -> LDS by LDS-DMA (16 bytes per lane, no VGPR
// staging: the registers go to a 4x2 wave tile of 32x32 accumulators), three stage buffers with counted vmcnt
// and a non-draining barrier exactly as eco_blocked.hip's bf16 kernels, k-pair fragments by ds_read_b64,
// v_mfma_f32_32x32x2_f32.  Split-K slices are separate rows of M that the output transform sums while it reads
// (no reduce launch).  Arithmetic is that of the round-1 route -- same products, fp32 accumulation; only the
// summation order over (channel, depth tap) inside a point changes.
//
// Replaces cudnn_conv_layer.cu:15-65 / base_conv_layer.cpp:264-287 for the stride-1 3x3(x3) convolutions together
// with eco_wino.hip's weight transform.
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kWgP = 36;   // F(4x4,3x3): 6x6 transform points
constexpr int kWgP3 = 216; // F(4x4x4,3x3x3): 6x6x6 transform points (the GEMM only sees more, shorter problems)
constexpr int kWgPS2 = 320; // stride-2 3x3x3 as eight polyphase F(4,2) x F(7,2) x F(7,2) problems: 5x8x8 points (eco_wino_s2.hip)
constexpr int kWgPS2D = 64; // ... and its 2-D form, F(7,2) x F(7,2) with direct depth taps in the reduction
constexpr int kWgKp = 8;             // k-pairs (16 reduction elements) per stage

struct WGemmArgs {
  const float* v;     // [P][cp][Q][2]
  const float* u;     // [P][mblocks][nstages][8][bmp][2]
  float* m;           // [P][ksplit][cout][ntot]
  int cout, cp, kd, nstages, ksplit;
  int Q, NB, ntot;    // row length / plane size / output positions (all in positions)
  int mblocks, bmp, nblk_n;
  long v_pstride, u_pstride, m_pstride;   // floats between points
};

// The three buffers of a stage: A [8][BMP][2], B [8][BN][2] floats.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgemm_kernel(const WGemmArgs a) {
  ECO_CLOCK("wgemm");
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int BMP = (BM + 63) / 64 * 64;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BN % 128 == 0, "");
  constexpr int A_PER_WAVE = BMP / 64;            // 1 KB pieces of the weight block per wave (BMP/16 pieces over 4 waves)
  constexpr int B_PER_WAVE = 2 * (BN / 128);      // 8 pair-rows x BN/128 pieces of 1 KB, two rows per wave
  constexpr int P = A_PER_WAVE + B_PER_WAVE;
  constexpr int A_VEC = kWgKp * BMP / 2;          // 16-byte vectors per stage buffer
  constexpr int B_VEC = kWgKp * BN / 2;

  ECO_DYNAMIC_LDS(lds_f);
  uint4* const Aw = (uint4*)lds_f;                // [3][A_VEC]
  uint4* const Bw = Aw + 3 * A_VEC;               // [3][B_VEC]

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  // XCD-aware numbering over the WHOLE launch (points x slices x tiles): the hardware deals consecutive blocks of the linearised
  // grid to the eight XCDs in turn; xcd_remap gives every XCD a contiguous range of logical work items, so the M-blocks of a
  // position tile -- consecutive items -- read their V rows through ONE L2.  (Until round 6 the remap ran inside a point only: with
  // the four / eight tiles per point of the stride-2 problems every M-block of a position tile sat on a different XCD and V was
  // fetched four / eight times -- PMC: 2.18 GB per res4a launch against 1.17 algorithmic.)
  const int ntiles = a.mblocks * a.nblk_n;
  const int per_pt = (int)gridDim.x;                       // = ntiles * ksplit
  const int work = xcd_remap((int)blockIdx.y * per_pt + (int)blockIdx.x, per_pt * (int)gridDim.y);
  const int pt = work / per_pt, rest = work - pt * per_pt;
  const int slice = rest / ntiles;
  const int tile = rest - slice * ntiles;
  const int mblk = tile % a.mblocks, nblk = tile / a.mblocks;
  const int n0 = nblk * BN;
  const int s_begin = (int)((long)slice * a.nstages / a.ksplit);
  const int s_end = (int)((long)(slice + 1) * a.nstages / a.ksplit);

  const float* const vp = a.v + (long)pt * a.v_pstride;
  const float* const up = a.u + (long)pt * a.u_pstride + (long)mblk * a.nstages * (kWgKp * BMP * 2);

  int l_stage = s_begin;
  auto issue_stage = [&](int buf) {
    const int cg = l_stage / a.kd, dz = l_stage - cg * a.kd;
    // weights: the stage's block is contiguous -- piece i of the wave = 64 lanes x 16 B
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q) {
      const int piece = wave * A_PER_WAVE + q;
      glds16((const uint4*)(up + (long)l_stage * (kWgKp * BMP * 2)) + piece * 64 + lane, Aw + buf * A_VEC + piece * 64);
    }
    // positions: wave w stages channel-pair rows 2w, 2w+1; a row is BN positions x 8 B = BN/128 pieces
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr;
      const float* src = vp + 2 * ((long)(cg * kWgKp + row) * a.Q + n0 + (long)dz * a.NB);
#pragma unroll
      for (int q = 0; q < BN / 128; ++q)
        glds16((const uint4*)src + q * 64 + lane, Bw + buf * B_VEC + row * (BN / 2) + q * 64);
    }
    ++l_stage;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  int ia[TM], ib[TN];   // float2 index of this lane's fragments in pair-row `half` of a stage buffer
#pragma unroll
  for (int i = 0; i < TM; ++i) { ia[i] = half * BMP + (wm * TM + i) * 32 + l31; ECO_OPAQUE(ia[i]); }
#pragma unroll
  for (int j = 0; j < TN; ++j) { ib[j] = half * BN + (wn * TN + j) * 32 + l31; ECO_OPAQUE(ib[j]); }

  // One stage = four steps t of two pair-rows (2t for lanes 0-31, 2t+1 for lanes 32-63: four k per step).  The
  // fragments of step t+1 are read while the 2*TM*TN MFMAs of step t issue (double-buffered registers, order pinned
  // with sched_fence): the compiler's own schedule fetched two steps at once with ds_read2st64_b64 -- which the LDS
  // serves at half the rate of two ds_read_b64 -- one MFMA ahead of their first use, and waited for them.
  auto compute = [&](int buf) {
    const float2* Ab = (const float2*)(Aw + buf * A_VEC);
    const float2* Bb = (const float2*)(Bw + buf * B_VEC);
    float2 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = Ab[ia[i]];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = Bb[ib[j]];
#pragma unroll
    for (int t = 0; t < kWgKp / 2; ++t) {
      const int cur = t & 1;
      if (t + 1 < kWgKp / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[cur ^ 1][i] = Ab[2 * (t + 1) * BMP + ia[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[cur ^ 1][j] = Bb[2 * (t + 1) * BN + ib[j]];
      }
      sched_fence();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(af[cur][i].x, bf[cur][j].x, acc[i][j]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(af[cur][i].y, bf[cur][j].y, acc[i][j]);
      sched_fence();
    }
  };

  if (s_begin < s_end) {
    issue_stage(0);
    if (s_begin + 1 < s_end) issue_stage(1);
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      if (s + 1 < s_end) wait_dma_all_but<P>(); else wait_dma_all_but<0>();
      wg_barrier_nodrain();
      if (s + 2 < s_end) issue_stage(buf == 0 ? 2 : buf - 1);   // (buf + 2) % 3: last read before this barrier
      sched_fence();
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
  // raw store: M[p][slice][channel][position], lane = position (128 contiguous bytes per half-wave)
  float* const mo = a.m + (long)pt * a.m_pstride + (long)slice * a.cout * a.ntot;
  const int mw = mblk * BM + wm * TM * 32, nw = n0 + wn * TN * 32;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + l31;
    if (n >= a.ntot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ch < a.cout) st(mo + (long)ch * a.ntot + n, acc[i][j][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Input transform into the pair-interleaved, depth-major, depth-padded layout.  One thread per (channel pair, padded
// depth plane d', position r = (b, th, tw), channel parity): a 6x6 tile -> V = B^T d B, one float per transform
// point; consecutive threads write consecutive floats (the parity is the fastest index of V).  The two padding
// planes get zeros.  Transform constants: Lavin & Gray 2015, F(4x4,3x3), as in eco_wino.hip.
struct WinoInPkArgs {
  const float* x;
  float* v;
  int n, cin, D, H, W, TH, TW, pd;
  int Q, NB;
  long v_pstride;
};

__device__ __forceinline__ void wino_bt_d_b(const float (&d)[6][6], float (&v)[6][6]) {
  constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                              {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  float t[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (BT[i][k] != 0.0f) acc += BT[i][k] * d[k][j];
      t[i][j] = acc;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (BT[j][k] != 0.0f) acc += t[i][k] * BT[j][k];
      v[i][j] = acc;
    }
}

// The 6x6 input tile of (channel plane xp, tile th, tw), zero outside the image.
template <int VEC>
__device__ __forceinline__ void wino_load_tile(const float* xp, int H, int W, int th, int tw, float (&dd)[6][6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int h = 4 * th - 1 + i;
    const bool hok = (unsigned)h < (unsigned)H;
    const float* rp = xp + (hok ? (long)h * W : 0l);
    const int wl = 4 * tw - 1, wr = 4 * tw + 4;
    const bool lok = hok && wl >= 0, rok = hok && wr < W;
    dd[i][0] = lok ? ld(rp + (lok ? wl : 0)) : 0.0f;
    dd[i][5] = rok ? ld(rp + (rok ? wr : 0)) : 0.0f;
    if (VEC == 4) {
      const bool ok = hok && 4 * tw < W;   // W % 4 == 0: the four columns are inside or outside together
      const float4 q = ld((const float4*)(rp + (ok ? 4 * tw : 0)));
      dd[i][1] = ok ? q.x : 0.0f; dd[i][2] = ok ? q.y : 0.0f; dd[i][3] = ok ? q.z : 0.0f; dd[i][4] = ok ? q.w : 0.0f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = 4 * tw + j;
        const bool ok = hok && w < W;
        dd[i][1 + j] = ok ? ld(rp + (ok ? w : 0)) : 0.0f;
      }
    }
  }
}

// Both layouts below are written as runs of consecutive floats per transform point: a workgroup owns the 256 floats
// [base, base + 256) of each of the 36 planes.  STAGED: every thread parks its 36 values in LDS ([36][256] floats) and
// the run of a point leaves as ONE 16-byte store per lane of one wave (1 KB per instruction, nine instructions per
// wave) instead of 36 four-byte stores per lane: the round-2 kernels were store-ISSUE bound (rocprof: wait_inst 0.50-0.75
// of the wave cycles at 3.7-5.0 TB/s).  Needs total % 4 == 0 and 16-byte aligned planes; otherwise the scalar form.
template <bool STAGED>
__device__ __forceinline__ void wino_store_points(float* vbase, long v_pstride, long base, long total, const float (&v)[6][6],
                                                  bool active, float* stage) {
  const int tid = (int)threadIdx.x;
  if (!STAGED) {
    if (active) {
#pragma unroll
      for (int p = 0; p < kWgP; ++p) st(vbase + (long)p * v_pstride + base + tid, v[p / 6][p % 6]);
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < kWgP; ++p) stage[p * 256 + tid] = v[p / 6][p % 6];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  if (base + 4 * lane < total) {
#pragma unroll
    for (int k = 0; k < kWgP / 4; ++k) {
      const int p = wave * (kWgP / 4) + k;
      const float4 q = *(const float4*)(stage + p * 256 + 4 * lane);
      st((float4*)(vbase + (long)p * v_pstride + base + 4 * lane), q);
    }
  }
  __syncthreads();
}

// Pair layout V[p][c/2][d'][r][c%2]: float index idx = ((cp*Dp + dp)*NB + r)*2 + e.  The two padding planes get zeros.
template <int VEC, bool STAGED>
__global__ __launch_bounds__(256) void wino_input_pk_kernel(const WinoInPkArgs a) {
  __shared__ __attribute__((aligned(16))) float stage[STAGED ? kWgP * 256 : 4];
  const int Dp = a.D + 2 * a.pd;
  const long total = (long)(a.cin / 2) * Dp * a.NB * 2;     // one thread per (channel pair, plane, position, parity)
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
    const long idx = base + threadIdx.x;
    const bool active = idx < total;
    float v[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) v[i][j] = 0.0f;
    if (active) {
      const int e = (int)(idx & 1);
      const long pos = idx >> 1;
      const int r = (int)(pos % a.NB);
      const long t1 = pos / a.NB;
      const int dp = (int)(t1 % Dp);
      const int cp = (int)(t1 / Dp);
      const int d = dp - a.pd;
      if ((unsigned)d < (unsigned)a.D) {   // else a padding plane: zeros at every transform point
        const int tw = r % a.TW, t2 = r / a.TW;
        const int th = t2 % a.TH, b = t2 / a.TH;
        float dd[6][6];
        wino_load_tile<VEC>(a.x + (((long)b * a.cin + 2 * cp + e) * a.D + d) * a.H * a.W, a.H, a.W, th, tw, dd);
        wino_bt_d_b(dd, v);
      }
    }
    wino_store_points<STAGED>(a.v, a.v_pstride, base, total, v, active, stage);
  }
}

// The same transform into the layout of the fused 2-D kernel below, V4[p][c/8][r][c%2][(c/2)%4]: the four k-pair
// elements a lane of wfused_kernel consumes in a row are one 16-byte vector.  One thread per float of a point's
// plane, consecutive threads on consecutive floats; D = 1, no padding planes.
template <int VEC, bool STAGED>
__global__ __launch_bounds__(256) void wino_input_q4_kernel(const WinoInPkArgs a) {
  __shared__ __attribute__((aligned(16))) float stage[STAGED ? kWgP * 256 : 4];
  const long total = (long)a.cin * a.NB;
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
    const long idx = base + threadIdx.x;
    const bool active = idx < total;
    float v[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) v[i][j] = 0.0f;
    if (active) {
      const int c4 = (int)(idx & 3), e = (int)((idx >> 2) & 1);
      const long pos = idx >> 3;
      const int r = (int)(pos % a.NB);
      const int q = (int)(pos / a.NB);
      const int ch = 8 * q + 2 * c4 + e;
      const int tw = r % a.TW, t2 = r / a.TW;
      const int th = t2 % a.TH, b = t2 / a.TH;
      float dd[6][6];
      wino_load_tile<VEC>(a.x + ((long)b * a.cin + ch) * a.H * a.W, a.H, a.W, th, tw, dd);
      wino_bt_d_b(dd, v);
    }
    wino_store_points<STAGED>(a.v, a.v_pstride, base, total, v, active, stage);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Output transform from M[p][slice][cout][d][r]: y tile = A^T (sum over slices m) A, then the fused epilogue
// (bias, Eltwise residual, raw store, folded BN, ReLU, both activated destinations; strided views).
struct WinoOutDmArgs {
  const float* m;
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;
  int relu;
  int n, cout, D, H, W, TH, TW;
  int NB, ntot, ksplit;
  long m_pstride;
  float* pool9;     // wfused_kernel<..., POOL = true>: P9[n][cout][3 TH][3 TW], see wino_pool9_store
};

template <int VEC>
struct WgVec;
template <>
struct WgVec<1> { typedef float type; };
template <>
struct WgVec<2> { typedef float2 type; };
template <>
struct WgVec<4> { typedef float4 type; };

// Element offsets of (image img, channel ch, spatial 0) in the four views of the epilogue (0 where a view is absent).
struct WinoViewOffsets { long res, raw, act, act2; };
__device__ __forceinline__ WinoViewOffsets wino_view_offsets(const WinoOutDmArgs& a, int img, int ch) {
  WinoViewOffsets o;
  o.res = a.residual.ptr ? view_base(a.residual, img, 0) + (long)ch * a.residual.stride_c : 0;
  o.raw = a.raw.ptr ? view_base(a.raw, img, 0) + (long)ch * a.raw.stride_c : 0;
  o.act = a.act.ptr ? view_base(a.act, img, 0) + (long)ch * a.act.stride_c : 0;
  o.act2 = a.act2.ptr ? view_base(a.act2, img, 0) + (long)ch * a.act2.stride_c : 0;
  return o;
}

// Second half of an output tile: y = s4 A with s4 = A^T m (4 x 6), then the fused epilogue, for channel `ch` (view
// offsets `o`), depth `d`, tile (th, tw).
// PRE (the stand-alone output transform): the tile's residual values are fetched before the first store -- as written
// below a row's residual load follows the previous row's stores, which may alias it for all the compiler knows: four
// dependent round trips per tile, visible on the latency-bound res4 / res5 launches (res5b_2 took twice res5b_1's time;
// output transforms 0.78 -> 0.72 ms per step).  Issuing them, and the parameters, even earlier -- ahead of the 36 loads of
// M -- was slower again (more registers live across those loads).  wfused_kernel (no residuals in its layers, 246 VGPRs)
// keeps the in-loop form.
template <int VEC, bool PRE = false>
__device__ __forceinline__ void wino_output_store(const WinoOutDmArgs& a, const float (&s4)[4][6], int ch, const WinoViewOffsets& o,
                                                  int d, int th, int tw) {
  typedef typename WgVec<VEC>::type vec_t;
  const unsigned cho = 4u * (unsigned)ch;
  const float b = a.bias ? ld_su<float>(a.bias, cho) : 0.0f;
  const float sc = a.bn_scale ? ld_su<float>(a.bn_scale, cho) : 1.0f, sh = a.bn_scale ? ld_su<float>(a.bn_shift, cho) : 0.0f;
  const float floor_v = a.relu ? 0.0f : -3.402823466e38f;   // one v_max instead of v_max + v_cndmask per output
  const bool plain = a.act.ptr && !a.residual.ptr && !a.raw.ptr;
  const float sh2 = b * sc + sh;
  vec_t resv[4][4 / VEC];
  if (PRE && !plain && a.residual.ptr) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q0 = 0; q0 < 4; q0 += VEC)
        if (4 * th + p < a.H && 4 * tw + q0 < a.W)
          resv[p][q0 / VEC] = ld((const vec_t*)((const float*)a.residual.ptr + o.res + (d * a.H + 4 * th + p) * a.W + 4 * tw + q0));
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int h = 4 * th + p;
    if (h >= a.H) continue;
    // y = s A, factored (10 VALU instructions instead of 18: beside f32 MFMAs each one is matrix-pipe time)
    const float t1 = s4[p][1] + s4[p][2], t2 = s4[p][1] - s4[p][2], t3 = s4[p][3] + s4[p][4], t4 = s4[p][3] - s4[p][4];
    const float yrow[4] = {s4[p][0] + t1 + t3, t2 + 2.0f * t4, t1 + 4.0f * t3, t2 + 8.0f * t4 + s4[p][5]};
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += VEC) {
      const int w0 = 4 * tw + q0;
      if (w0 >= a.W) continue;  // W % VEC == 0: the VEC outputs are inside or outside together
      const int sp = (d * a.H + h) * a.W + w0;
      if (plain) {   // only the activated value is wanted: (y + b)*sc + sh = y*sc + (b*sc + sh), one FMA and one max per output
        vec_t ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ((float*)&ov)[e] = fmaxf(yrow[q0 + e] * sc + sh2, floor_v);
        st((vec_t*)(a.act.ptr + o.act + sp), ov);
        if (a.act2.ptr) st((vec_t*)(a.act2.ptr + o.act2 + sp), ov);
        continue;
      }
      float val[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) val[e] = yrow[q0 + e] + b;
      if (a.residual.ptr) {
        const vec_t rv = PRE ? resv[p][q0 / VEC] : ld((const vec_t*)((const float*)a.residual.ptr + o.res + sp));
#pragma unroll
        for (int e = 0; e < VEC; ++e) val[e] += ((const float*)&rv)[e];
      }
      if (a.raw.ptr) {
        vec_t ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ((float*)&ov)[e] = val[e];
        st((vec_t*)(a.raw.ptr + o.raw + sp), ov);
      }
      if (a.act.ptr) {
        vec_t ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ((float*)&ov)[e] = fmaxf(val[e] * sc + sh, floor_v);
        st((vec_t*)(a.act.ptr + o.act + sp), ov);
        if (a.act2.ptr) st((vec_t*)(a.act2.ptr + o.act2 + sp), ov);
      }
    }
  }
}

// The activated tile folded towards a MAX 3x3 stride-2 unpadded pooling that follows the convolution (conv2_3x3 -> pool2,
// models_ECO_Lite/kinetics/deploy.prototxt:103-128): pooled row 2 th needs tile rows {0, 1, 2}, pooled row 2 th + 1 rows {2, 3}
// and row 0 of the tile below, pooled row 2 th - 1 row 0 of this tile -- so a 4 x 4 tile contributes to 3 x 3 pooled cells, one of
// them alone.  The nine partial maxima are stored as a 3 x 3 block of P9[img][ch][3 TH][3 TW] (9 floats per tile instead of the
// 16 of the tile itself: the conv's own output is never written), and pool9_finish_kernel folds the blocks of neighbouring
// tiles into the pooled blob.  (Measured and dropped: the eight shared cells as device-scope atomic max straight into the pooled
// blob -- tools/ubench/atomic_pool.hip: 1.22 ms for conv2_3x3's tiles against 0.43 for storing them whole.)
// Planes must tile by 4 exactly (H % 4 == 0, W % 4 == 0); only the activated value exists (no raw / residual / second view).
__device__ __forceinline__ void wino_pool9_store(const WinoOutDmArgs& a, const float (&s4)[4][6], int ch, int img, int th, int tw) {
  const unsigned cho = 4u * (unsigned)ch;
  const float b = a.bias ? ld_su<float>(a.bias, cho) : 0.0f;
  const float sc = a.bn_scale ? ld_su<float>(a.bn_scale, cho) : 1.0f, sh = a.bn_scale ? ld_su<float>(a.bn_shift, cho) : 0.0f;
  const float floor_v = a.relu ? 0.0f : -3.402823466e38f;
  const float sh2 = b * sc + sh;
  float cm[4][3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float t1 = s4[p][1] + s4[p][2], t2 = s4[p][1] - s4[p][2], t3 = s4[p][3] + s4[p][4], t4 = s4[p][3] - s4[p][4];
    const float y0 = fmaxf((s4[p][0] + t1 + t3) * sc + sh2, floor_v), y1 = fmaxf((t2 + 2.0f * t4) * sc + sh2, floor_v);
    const float y2 = fmaxf((t1 + 4.0f * t3) * sc + sh2, floor_v), y3 = fmaxf((t2 + 8.0f * t4 + s4[p][5]) * sc + sh2, floor_v);
    cm[p][0] = y0;
    cm[p][1] = fmaxf(fmaxf(y0, y1), y2);
    cm[p][2] = fmaxf(y2, y3);
  }
  float* o = a.pool9 + (((long)img * a.cout + ch) * (3 * a.TH) + 3 * th) * (3 * a.TW) + 3 * tw;
  const int pitch = 3 * a.TW;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    st(o + j, cm[0][j]);
    st(o + pitch + j, fmaxf(fmaxf(cm[0][j], cm[1][j]), cm[2][j]));
    st(o + 2 * pitch + j, fmaxf(cm[2][j], cm[3][j]));
  }
}

// P9 -> the pooled blob y[n*c][2 TH][2 TW]: pooled row 2 t = block row 3 t + 1; pooled row 2 t + 1 = max(block row 3 t + 2, block
// row 3 (t + 1)) (the tile below, if there is one: pooling_layer.cpp:131-147 clips the window at the bottom edge); columns alike.
// One thread per tile column of a pooled row: six floats of P9 in, two out.
__global__ __launch_bounds__(256) void pool9_finish_kernel(const float* p9, float* y, long planes, int TH, int TW) {
  const long total = planes * 2 * TH * TW;
  const int pitch = 3 * TW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int s = (int)(i % TW);
    const long t1 = i / TW;
    const int oh = (int)(t1 % (2 * TH));
    const long pl = t1 / (2 * TH);
    const int t = oh >> 1;
    const float* r0 = p9 + (pl * 3 * TH + 3 * t + 1 + (oh & 1)) * pitch + 3 * s + 1;
    const bool two_rows = (oh & 1) && t + 1 < TH, right = s + 1 < TW;
    float o0 = ld(r0), o1 = fmaxf(ld(r0 + 1), right ? ld(r0 + 2) : -3.402823466e38f);
    if (two_rows) {
      const float* r1 = r0 + pitch;
      o0 = fmaxf(o0, ld(r1));
      o1 = fmaxf(o1, fmaxf(ld(r1 + 1), right ? ld(r1 + 2) : -3.402823466e38f));
    }
    st((float2*)(y + (pl * 2 * TH + oh) * (2 * TW) + 2 * s), make_float2(o0, o1));
  }
}

template <int VEC>
__device__ __forceinline__ void wino_output_from_s4(const WinoOutDmArgs& a, const float (&s4)[4][6], int ch, int img, int d,
                                                    int th, int tw) {
  wino_output_store<VEC, true>(a, s4, ch, wino_view_offsets(a, img, ch), d, th, tw);
}

// One output tile: y = A^T m A, then the fused epilogue.
template <int VEC>
__device__ __forceinline__ void wino_output_tile(const WinoOutDmArgs& a, const float (&m)[6][6], int ch, int img, int d,
                                                 int th, int tw) {
  float s4[4][6];   // A^T m, factored as in wino_output_store
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float t1 = m[1][j] + m[2][j], t2 = m[1][j] - m[2][j], t3 = m[3][j] + m[4][j], t4 = m[3][j] - m[4][j];
    s4[0][j] = m[0][j] + t1 + t3;
    s4[1][j] = t2 + 2.0f * t4;
    s4[2][j] = t1 + 4.0f * t3;
    s4[3][j] = t2 + 8.0f * t4 + m[5][j];
  }
  wino_output_from_s4<VEC>(a, s4, ch, img, d, th, tw);
}

// A thread takes one tile column r of one (channel, depth) row of M, in M's own order (ch, d, r): every point's load
// is contiguous across the wave (256 B per point) and the stores are runs of one H x W plane per (img, ch, d).
// (The (img, ch, d, t) order -- stores contiguous over the whole tensor, M read in runs of TH*TW floats: 196 / 64 /
// 16 bytes in res3 / 4 / 5 -- was 9 % slower over the step's 17 transforms; 2 or 4 columns per thread with vector
// loads of M were slower again: 72 / 144 live tile values.)
// STAGED (round 3): the 256 consecutive floats a workgroup needs of each point's plane are one 1 KB run; wave w moves
// the runs of points 9w .. 9w+8 into LDS by LDS-DMA (16 bytes per lane, nine instructions per wave and slice, all in
// flight at once, no VGPRs) and every thread then picks its 36 values with conflict-free ds_read_b32 -- instead of 36
// four-byte global loads per thread and slice, which kept the small res4 / res5 launches latency bound (0.22-0.45 of
// the HBM floor).  Needs 16-byte aligned planes (cout * ntot and the plane stride multiples of 4 floats).
template <int VEC, bool STAGED>
__global__ __launch_bounds__(256) void wino_output_dm_kernel(const WinoOutDmArgs a) {
  __shared__ __attribute__((aligned(16))) float stage[STAGED ? kWgP * 256 : 4];
  const long total = (long)a.cout * a.ntot;
  const int tpp = a.TH * a.TW;
  const int tid = (int)threadIdx.x;
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
    const long q = base + tid;
    const bool active = q < total;
    float m[6][6];
    if (STAGED) {
      const int lane = tid & 63, wave = uniform(tid >> 6);
      const bool dma_ok = base + 4 * lane < total;       // total % 4 == 0: a lane's four floats are inside or outside together
      for (int sl = 0; sl < a.ksplit; ++sl) {
        const float* src = a.m + (long)sl * a.cout * a.ntot + base + 4 * lane;
        if (dma_ok) {
#pragma unroll
          for (int k = 0; k < kWgP / 4; ++k) {
            const int p = wave * (kWgP / 4) + k;
            glds16((const uint4*)(src + (long)p * a.m_pstride), (uint4*)(stage + p * 256));
          }
        }
        wait_dma_all_but<0>();
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kWgP; ++p) {
          const float v = stage[p * 256 + tid];
          m[p / 6][p % 6] = sl == 0 ? v : m[p / 6][p % 6] + v;
        }
        __syncthreads();
      }
    } else if (active) {
      const float* mp = a.m + q;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float* qq = mp + (long)(6 * i + j) * a.m_pstride;
          float s = ld(qq);
          for (int sl = 1; sl < a.ksplit; ++sl) s += ld(qq + (long)sl * a.cout * a.ntot);
          m[i][j] = s;
        }
    }
    if (!active) continue;
    const int ch = (int)(q / a.ntot);
    const int rem = (int)(q - (long)ch * a.ntot);
    const int d = rem / a.NB, r = rem - d * a.NB;
    const int img = r / tpp, t = r - img * tpp;
    const int th = t / a.TW, tw = t - th * a.TW;
    wino_output_tile<VEC>(a, m, ch, img, d, th, tw);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused transformed-domain GEMM + output transform for the short-reduction 2-D layers (kd = 1, cin = 64 / 96:
// conv2_3x3 and the inception 3x3 convs).  As separate launches these layers write M (2.25x the output, fp32) and
// read it back: conv2_3x3 alone moves 5.5 GB that way.  Here a workgroup owns 32 output channels x 32 tile columns
// and walks the 36 points in three groups of two transform rows (12 points): wave w computes the 32x32 products of
// three points of the group with both operands loaded straight from global memory into registers (four consecutive
// k-pair elements of a lane are one 16-byte vector in V4's and the packed U's layouts, consecutive lanes on
// consecutive vectors; the stream runs three 16-k-pair chunks ahead of the MFMAs, across group boundaries), parks them in LDS as
// M[12][32][32] (48 KB: two workgroups per CU, one transforming while the other multiplies), and after a barrier
// every thread folds the group's two rows into the A^T m partial sums of its four (channel, column) pairs.  After
// the third group the second half of the transform and the epilogue run from registers.  M never exists in HBM.
// The tile cannot be larger (36 points x 32 x 32 x 4 B = 144 KB is all of the LDS), so each product reads its
// operands from L2 once: 8 flop per byte, L2-bandwidth bound (measured 11.5 TB/s with the MFMAs removed).
struct WFusedArgs {
  const float* v;     // V4[36][KP/4][NB][2][4]  (wino_input_q4_kernel)
  const float* u;     // [36][mblocks][KP/4][64 lanes][4]: lane (m = l&31, half = l>>5) holds A[m][2*(4*q+i)+half], i < 4
  WinoOutDmArgs o;    // epilogue, views, image / tile geometry (D = 1)
  int mblocks, nblk;
  int Q;
  long v_pstride, u_pstride;
};

#ifndef ECO_WFUSED_R
#define ECO_WFUSED_R 3
#define ECO_WFUSED_OCC 2
#endif
template <int KP, int VEC, bool POOL = false>
__global__ __launch_bounds__(256, ECO_WFUSED_OCC) void wfused_kernel(const WFusedArgs a) {
  ECO_CLOCK("wfused");
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int CH = 16;                 // k-pairs per chunk
  constexpr int NCH = KP / CH;           // chunks per point
  constexpr int NPT = 9;                 // points per wave: three per group
  constexpr int T = NPT * NCH;           // chunks per wave
  constexpr int R = ECO_WFUSED_R;        // chunks in flight: 3 x 8 sixteen-byte loads per lane (4 would spill)
  static_assert(KP % CH == 0, "");
  ECO_DYNAMIC_LDS(lds);                  // M[12][32][32]: the group's points, local index lp = 6*(0 | 1: which row of the pair) + column
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tile = xcd_remap((int)blockIdx.x, a.mblocks * a.nblk);
  const int mb = tile % a.mblocks, nb = tile / a.mblocks;
  const int n0 = nb * 32;
  const unsigned a_voff = 16u * (unsigned)lane;             // this lane's 16-byte vector of a packed-U piece
  const unsigned b_voff = 16u * (unsigned)(2 * l31 + half);  // ... and within a (q, 32 columns) row of V4 ([col][2][4])

  float4 ra[R][CH / 4], rb[R][CH / 4];
  auto issue = [&](int slot, int t) {    // chunk t: point index t / NCH = 3*group + k, local point lp = wave + 4*k
    const int pi = t / NCH, c = t % NCH;
    const int lp = wave + 4 * (pi % 3), rr = lp >= 6 ? 1 : 0;
    const int g = pi / 3;                // group g holds transform rows (1, 2), (3, 4), (0, 5): see the fold below
    const int p = 6 * (g == 0 ? 1 + rr : g == 1 ? 3 + rr : 5 * rr) + lp - 6 * rr;
    // wave-uniform bases (SALU) + this lane's constant byte offset: no per-lane address arithmetic per load
    const float* us = a.u + (long)p * a.u_pstride + ((long)mb * KP + c * CH) * 64;
    const float* vs = a.v + (long)p * a.v_pstride + ((long)(c * (CH / 4)) * a.Q + n0) * 8;
#pragma unroll
    for (int k4 = 0; k4 < CH / 4; ++k4) ra[slot][k4] = ld_su<float4>(us + k4 * 256, a_voff);
#pragma unroll
    for (int k4 = 0; k4 < CH / 4; ++k4) rb[slot][k4] = ld_su<float4>(vs + (long)k4 * a.Q * 8, b_voff);
  };
#pragma unroll
  for (int t = 0; t < R; ++t) issue(t, t);
  sched_fence();
  float s4[4][4][6];                     // A^T m partial sums of this thread's pairs (channel (tid>>5) + 8u, column tid&31)
  const int col = tid & 31, mrow0 = tid >> 5;
  f32x16 acc;
  static_for<T>([&](auto TT) __attribute__((always_inline)) {   // every chunk index a compile-time constant
    constexpr int t = decltype(TT)::value;
#pragma unroll
    for (int k4 = 0; k4 < CH / 4; ++k4) {
      const float4 av = ra[t % R][k4], bv = rb[t % R][k4];
      // a point's first product takes the constant 0 as its C operand: no 16 v_mov per point to clear the accumulator
      acc = mfma_32x32x2(av.x, bv.x, (t % NCH == 0 && k4 == 0) ? kZero16 : acc);
      acc = mfma_32x32x2(av.y, bv.y, acc);
      acc = mfma_32x32x2(av.z, bv.z, acc);
      acc = mfma_32x32x2(av.w, bv.w, acc);
    }
    sched_fence();
    if constexpr (t + R < T) issue(t % R, t + R);
    sched_fence();
    if constexpr (t % NCH == NCH - 1) {
      constexpr int pi = t / NCH;
      const int lp = wave + 4 * (pi % 3);
      float* mo = lds + (lp * 32 + 4 * half) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) mo[((r & 3) + 8 * (r >> 2)) * 32] = acc[r];
      if constexpr (pi % 3 == 2) {       // the group's 12 products are complete: fold its two rows into s4
        constexpr int g = pi / 3;
        __syncthreads();
        // A^T m with the rows taken in the pairs the transform factors into -- t = m1 +- m2 feeds (1, 1, 1, 1) /
        // (1, -1, 1, -1), t = m3 +- m4 feeds (1, 2, 4, 8) / (1, -2, 4, -8) -- 10 VALU instructions per column instead
        // of 18, and nothing to clear first
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float ma = lds[((j * 32) + mrow0 + 8 * u) * 32 + col];
            const float mb2 = lds[(((6 + j) * 32) + mrow0 + 8 * u) * 32 + col];
            if (g == 0) {          // rows 1, 2
              const float t1 = ma + mb2, t2 = ma - mb2;
              s4[u][0][j] = t1; s4[u][1][j] = t2; s4[u][2][j] = t1; s4[u][3][j] = t2;
            } else if (g == 1) {   // rows 3, 4
              const float t3 = ma + mb2, t4 = ma - mb2;
              s4[u][0][j] += t3; s4[u][1][j] += 2.0f * t4; s4[u][2][j] += 4.0f * t3; s4[u][3][j] += 8.0f * t4;
            } else {               // rows 0, 5
              s4[u][0][j] += ma; s4[u][3][j] += mb2;
            }
          }
        if (g < 2) __syncthreads();
      }
    }
  });
  const int tpp = a.o.TH * a.o.TW;
  const int r = n0 + col;
  if (r < a.o.NB) {
    const int img = r / tpp, tt = r - img * tpp;
    const int th = tt / a.o.TW, tw = tt - th * a.o.TW;
    // the image / tile decode and the view bases once per thread; the four channels are 8 * stride_c apart
    WinoViewOffsets o = wino_view_offsets(a.o, img, mb * 32 + mrow0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = mb * 32 + mrow0 + 8 * u;
      if (POOL) {
        if (ch < a.o.cout) wino_pool9_store(a.o, s4[u], ch, img, th, tw);
        continue;
      }
      if (ch < a.o.cout) wino_output_store<VEC>(a.o, s4[u], ch, o, 0, th, tw);
      o.res += 8 * a.o.residual.stride_c; o.raw += 8 * a.o.raw.stride_c;
      o.act += 8 * a.o.act.stride_c; o.act2 += 8 * a.o.act2.stride_c;
    }
  }
}

static unsigned wg_grid(long total) {
  long b = ceil_div(total, 256);
  if (b > 1048576) b = 1048576;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace eco

using namespace eco;

static int wgemm_check_plan(const eco_wgemm_plan* p) {
  ECO_REQUIRE(p != nullptr, "wgemm: null plan");
  ECO_REQUIRE(p->points == kWgP || ((p->points == kWgP3 || p->points == kWgPS2 || p->points == kWgPS2D) && p->kd == 1),
              "wgemm: F(4x4,3x3) (36 transform points) or, with kd = 1, F(4x4x4,3x3x3) (216) / the stride-2 polyphase forms (320, 64) "
              "are supported, got %d", p->points);
  ECO_REQUIRE(p->n > 0 && p->cin > 0 && p->cin % 16 == 0 && p->cout > 0 && p->d > 0 && p->th > 0 && p->tw > 0 &&
                  (p->kd == 1 || p->kd == 3),
              "wgemm: bad problem (n=%d cin=%d cout=%d d=%d tiles %dx%d kd=%d; cin must be a multiple of 16)", p->n, p->cin,
              p->cout, p->d, p->th, p->tw, p->kd);
  ECO_REQUIRE((p->bm == 128 || p->bm == 96 || p->bm == 64 || p->bm == 32) && (p->bn == 128 || p->bn == 256),
              "wgemm: unsupported tile %dx%d", p->bm, p->bn);
  ECO_REQUIRE(p->nstages == (p->cin / 16) * p->kd && p->ksplit >= 1 && p->ksplit <= p->nstages, "wgemm: bad plan");
  return ECO_OK;
}

extern "C" int eco_wgemm_plan_create(int32_t n, int32_t cin, int32_t cout, int32_t d, int32_t th, int32_t tw, int32_t kd,
                                     int32_t points, int32_t num_cu, eco_wgemm_plan* plan) {
  clear_error();
  ECO_REQUIRE(plan != nullptr && num_cu >= 0, "wgemm: bad argument");
  if (num_cu == 0) num_cu = current_device_num_cu();   // as eco_conv_plan_create does
  memset(plan, 0, sizeof(*plan));
  plan->n = n; plan->cin = cin; plan->cout = cout; plan->d = d; plan->th = th; plan->tw = tw; plan->kd = kd;
  plan->points = points;
  int bm;
  if (cout <= 32) bm = 32;
  else if (cout <= 64) bm = 64;
  else if (cout <= 96) bm = 96;
  else {
    bm = 128;
    long best = ceil_div(cout, 128) * 128;
    const int cands[2] = {96, 64};
    for (int c : cands) {
      const long padded = ceil_div(cout, c) * c;
      if (padded < best) { best = padded; bm = c; }
    }
  }
  plan->bm = bm;
  plan->nstages = (cin / 16) * kd;
  plan->ksplit = 1;
  plan->bn = 256;
  if (cin % 16 == 0 && cin > 0 && n > 0 && d > 0 && th > 0 && tw > 0) {
    // Tile width and split-K by a small cost model.  The CUs are MFMA-bound once two workgroups share one, so a
    // launch takes (workgroups on the fullest CU) x (one workgroup's MFMA time on a whole CU) -- 576 workgroups on
    // 256 CUs cost three of those, not 576/768 of a round -- plus ~3 us of pipeline fill and store per round of
    // resident workgroups; pick the (bn, ksplit) that minimises it.
    const long nb = (long)n * th * tw, ntot = nb * d;
    const long mblocks = ceil_div(cout, bm);
    double best = 1e30;
    int max_sp = plan->nstages / 4;
    if (max_sp > 4) max_sp = 4;
    if (max_sp < 1) max_sp = 1;
    for (int bn = 256; bn >= 128; bn -= 128)
      for (int sp = 1; sp <= max_sp; ++sp) {
        const long wgs = mblocks * ceil_div(ntot, bn) * sp * points;
        const long lds = 3L * kWgKp * ((bm + 63) / 64 * 64 + bn) * 8;          // three stage buffers
        long occ = 160 * 1024 / lds;                                         // workgroups resident per CU (LDS-bound)
        if (occ > 4) occ = 4;
        const long slots = occ * num_cu;
        const double t_cu = 2.0 * bm * bn * 16.0 * (plan->nstages / (double)sp) / (157.3e12 / num_cu);
        // every extra slice is one more write of M here and one more read in the output transform (~4 TB/s each,
        // measured on the transforms): without this term res4 (1152 tiles on 512 slots) took three slices and the
        // output transform gave back what the GEMM had gained
        const double t_slices = (sp - 1) * 2.0 * (double)points * cout * ntot * 4.0 / 4e12;
        const double t = (double)ceil_div(wgs, (long)num_cu) * t_cu + (double)ceil_div(wgs, slots) * 3e-6 + t_slices;
        if (t < best * 0.97) { best = t; plan->bn = bn; plan->ksplit = sp; }
      }
  }
  // Short reductions (the 2-D layers: 4-6 stages) sit on the HBM ridge and spend a third of a tile's life storing:
  // 128-wide tiles keep three workgroups per CU instead of two in flight (measured over the seven 2-D GEMMs of the
  // configs[1] step: 2.07 -> 1.98 ms).
  if (plan->nstages <= 6 && plan->ksplit == 1) plan->bn = 128;
  // F(4x4x4,3x3x3): six times as many problems with a third of the reduction each -- the same trade (measured on the three
  // trunk stages at 32 clips: 0.446 / 0.267 ms with 256-wide tiles, 0.421 / 0.254 with 128-wide ones)
  if (points == kWgP3 && plan->ksplit == 1) plan->bn = 128;
  const long nb = (long)n * th * tw;
  const int pd = kd / 2;
  plan->mblocks = (int)ceil_div(cout, bm);
  plan->bmp = (bm + 63) / 64 * 64;
  plan->q = (int64_t)(d + 2 * pd) * nb;
  plan->u_elems = (int64_t)points * plan->mblocks * plan->nstages * kWgKp * plan->bmp * 2;
  // one tile of slack behind the last point: the last tile of a row reads up to bn positions past its end
  plan->v_elems = ((int64_t)points * (cin / 2) * plan->q + 256 + 2 * nb) * 2;
  plan->m_elems = (int64_t)points * plan->ksplit * cout * nb * d;
  return wgemm_check_plan(plan);
}

extern "C" int eco_wgemm_pack_weights(const eco_wgemm_plan* plan, const float* u, float* up) {
  clear_error();
  if (int rc = wgemm_check_plan(plan)) return rc;
  ECO_REQUIRE(u && up, "wgemm pack: null argument");
  memset(up, 0, sizeof(float) * (size_t)plan->u_elems);
  const int cin = plan->cin, cout = plan->cout, kd = plan->kd, bm = plan->bm, bmp = plan->bmp;
  // u[p][co][ci][z] -> up[p][mblock][stage = (ci/16)*kd + z][(ci%16)/2][co - mblock*bm][ci%2]
  for (int p = 0; p < plan->points; ++p)
    for (int co = 0; co < cout; ++co) {
      const int mb = co / bm, ml = co - mb * bm;
      for (int ci = 0; ci < cin; ++ci)
        for (int z = 0; z < kd; ++z) {
          const long stage = (long)(ci / 16) * kd + z;
          const long o = ((((long)p * plan->mblocks + mb) * plan->nstages + stage) * kWgKp + (ci % 16) / 2) * bmp * 2 +
                         (long)ml * 2 + (ci & 1);
          up[o] = u[(((long)p * cout + co) * cin + ci) * kd + z];
        }
    }
  return ECO_OK;
}

extern "C" int eco_wino_input_pk_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t h, int32_t w,
                                         void* stream) {
  clear_error();
  if (int rc = wgemm_check_plan(plan)) return rc;
  ECO_REQUIRE(x && v && h > 0 && w > 0, "winograd input transform: bad argument");
  ECO_REQUIRE(plan->th == (h + 3) / 4 && plan->tw == (w + 3) / 4, "winograd input transform: plan is for %dx%d tiles", plan->th,
              plan->tw);
  WinoInPkArgs a;
  a.x = x; a.v = v; a.n = plan->n; a.cin = plan->cin; a.D = plan->d; a.H = h; a.W = w; a.TH = plan->th; a.TW = plan->tw;
  a.pd = plan->kd / 2;
  a.NB = plan->n * plan->th * plan->tw;
  a.Q = (int)plan->q;
  a.v_pstride = (long)(plan->cin / 2) * plan->q * 2;
  const long total = (long)(plan->cin / 2) * (plan->d + 2 * a.pd) * a.NB * 2;
  const bool vec4 = w % 4 == 0 && ((uintptr_t)x & 15) == 0;
  const bool staged = total % 4 == 0 && a.v_pstride % 4 == 0 && ((uintptr_t)v & 15) == 0;   // 16-byte stores of V
  const dim3 grid(wg_grid(total)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (staged) {
    if (vec4) hipLaunchKernelGGL((wino_input_pk_kernel<4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_input_pk_kernel<1, true>), grid, block, 0, s, a);
  } else {
    if (vec4) hipLaunchKernelGGL((wino_input_pk_kernel<4, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_input_pk_kernel<1, false>), grid, block, 0, s, a);
  }
  return check_launch("eco_wino_input_pk_forward");
}

extern "C" int eco_wino_input_q4_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t h, int32_t w,
                                         void* stream) {
  clear_error();
  if (int rc = wgemm_check_plan(plan)) return rc;
  ECO_REQUIRE(x && v && h > 0 && w > 0, "winograd input transform: bad argument");
  ECO_REQUIRE(plan->kd == 1 && plan->d == 1 && plan->cin % 8 == 0, "winograd input transform (fused layout): 2-D layers, cin %% 8 == 0");
  ECO_REQUIRE(plan->th == (h + 3) / 4 && plan->tw == (w + 3) / 4, "winograd input transform: plan is for %dx%d tiles", plan->th,
              plan->tw);
  WinoInPkArgs a;
  a.x = x; a.v = v; a.n = plan->n; a.cin = plan->cin; a.D = 1; a.H = h; a.W = w; a.TH = plan->th; a.TW = plan->tw;
  a.pd = 0;
  a.NB = plan->n * plan->th * plan->tw;
  a.Q = a.NB;
  a.v_pstride = (long)plan->cin * a.NB;
  const long total = (long)plan->cin * a.NB;
  const bool vec4 = w % 4 == 0 && ((uintptr_t)x & 15) == 0;
  const bool staged = total % 4 == 0 && a.v_pstride % 4 == 0 && ((uintptr_t)v & 15) == 0;   // 16-byte stores of V
  const dim3 grid(wg_grid(total)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (staged) {
    if (vec4) hipLaunchKernelGGL((wino_input_q4_kernel<4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_input_q4_kernel<1, true>), grid, block, 0, s, a);
  } else {
    if (vec4) hipLaunchKernelGGL((wino_input_q4_kernel<4, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_input_q4_kernel<1, false>), grid, block, 0, s, a);
  }
  return check_launch("eco_wino_input_q4_forward");
}

template <int TM, int TN, int WM, int WN>
static int launch_wgemm(const WGemmArgs& a, int points, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BMP = (BM + 63) / 64 * 64;
  const size_t lds = (size_t)3 * kWgKp * (BMP + BN) * 8;
  if (lds > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((wgemm_kernel<TM, TN, WM, WN>), "wgemm");
  const int grid = a.mblocks * a.nblk_n * a.ksplit;
  hipLaunchKernelGGL((wgemm_kernel<TM, TN, WM, WN>), dim3(grid, points), dim3(256), lds, stream, a);
  return check_launch("eco_wgemm_forward");
}

extern "C" int eco_wgemm_forward(const eco_wgemm_plan* plan, const float* v, const float* up, float* m, void* stream) {
  clear_error();
  if (int rc = wgemm_check_plan(plan)) return rc;
  ECO_REQUIRE(v && up && m, "wgemm: null argument");
  ECO_REQUIRE((((uintptr_t)v | (uintptr_t)up) & 15) == 0, "wgemm: operands must be 16-byte aligned");
  WGemmArgs a;
  a.v = v; a.u = up; a.m = m;
  a.cout = plan->cout; a.cp = plan->cin / 2; a.kd = plan->kd; a.nstages = plan->nstages; a.ksplit = plan->ksplit;
  a.NB = plan->n * plan->th * plan->tw;
  a.Q = (int)plan->q;
  a.ntot = a.NB * plan->d;
  a.mblocks = plan->mblocks; a.bmp = plan->bmp;
  a.nblk_n = (int)ceil_div(a.ntot, plan->bn);
  a.v_pstride = (long)a.cp * plan->q * 2;
  a.u_pstride = (long)plan->mblocks * plan->nstages * kWgKp * plan->bmp * 2;
  a.m_pstride = (long)plan->ksplit * plan->cout * a.ntot;
  hipStream_t s = (hipStream_t)stream;
  if (plan->bn == 256) {
    switch (plan->bm) {
      case 128: return launch_wgemm<4, 2, 1, 4>(a, plan->points, s);
      case 96: return launch_wgemm<3, 2, 1, 4>(a, plan->points, s);
      case 64: return launch_wgemm<2, 2, 1, 4>(a, plan->points, s);
      case 32: return launch_wgemm<1, 2, 1, 4>(a, plan->points, s);
    }
  } else {
    switch (plan->bm) {
      case 128: return launch_wgemm<2, 2, 2, 2>(a, plan->points, s);
      case 96: return launch_wgemm<3, 1, 1, 4>(a, plan->points, s);
      case 64: return launch_wgemm<2, 1, 1, 4>(a, plan->points, s);
      case 32: return launch_wgemm<1, 1, 1, 4>(a, plan->points, s);
    }
  }
  return fail(ECO_ERR_INVALID, "wgemm: unsupported tile %dx%d", plan->bm, plan->bn);
}

extern "C" int eco_wino_output_dm_forward(const eco_wgemm_plan* plan, const float* m, int32_t h, int32_t w,
                                          const eco_conv_epilogue* ep, void* stream) {
  clear_error();
  if (int rc = wgemm_check_plan(plan)) return rc;
  ECO_REQUIRE(m && ep && h > 0 && w > 0, "winograd output transform: bad argument");
  ECO_REQUIRE(plan->th == (h + 3) / 4 && plan->tw == (w + 3) / 4, "winograd output transform: plan is for %dx%d tiles",
              plan->th, plan->tw);
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "winograd output transform: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "winograd output transform: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "winograd output transform: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "winograd output transform: segmented (sibling) launches exist for the direct kernels only");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "winograd output transform: view needs t >= 1 and stride_c >= 1");
  WinoOutDmArgs a;
  a.m = m; a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.n = plan->n; a.cout = plan->cout; a.D = plan->d; a.H = h; a.W = w; a.TH = plan->th; a.TW = plan->tw;
  a.NB = plan->n * plan->th * plan->tw;
  a.ntot = a.NB * plan->d;
  a.ksplit = plan->ksplit;
  a.m_pstride = (long)plan->ksplit * plan->cout * a.ntot;
  int vec = 4;
  auto limit = [&](const eco_view& v) {
    if (!v.ptr) return;
    while (vec > 1 && (((uintptr_t)v.ptr % (4 * vec)) || v.stride_b % vec || v.stride_t % vec || v.stride_c % vec)) vec /= 2;
  };
  while (vec > 1 && w % vec) vec /= 2;
  limit(a.residual); limit(a.raw); limit(a.act); limit(a.act2);
  const long tiles = (long)a.n * a.cout * a.D * a.TH * a.TW;
  const dim3 grid(wg_grid(tiles)), block(256);
  hipStream_t s = (hipStream_t)stream;
  // LDS-DMA of M for the small launches (res4 / res5: a round or two of workgroups, latency bound: 0.045 -> 0.035 ms per
  // res5 transform); the large ones stream at 4.9-5.3 TB/s either way and lose 5-10 % to the extra barriers
  const bool staged = ((long)a.cout * a.ntot) % 4 == 0 && a.m_pstride % 4 == 0 && ((uintptr_t)m & 15) == 0 &&
                      (long)a.cout * a.ntot <= (2l << 20);
  if (staged) {
    if (vec == 4) hipLaunchKernelGGL((wino_output_dm_kernel<4, true>), grid, block, 0, s, a);
    else if (vec == 2) hipLaunchKernelGGL((wino_output_dm_kernel<2, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_output_dm_kernel<1, true>), grid, block, 0, s, a);
  } else {
    if (vec == 4) hipLaunchKernelGGL((wino_output_dm_kernel<4, false>), grid, block, 0, s, a);
    else if (vec == 2) hipLaunchKernelGGL((wino_output_dm_kernel<2, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wino_output_dm_kernel<1, false>), grid, block, 0, s, a);
  }
  return check_launch("eco_wino_output_dm_forward");
}

// ---- fused GEMM + output transform (wfused_kernel) ------------------------------------------------------------------
static int wfused_check(const eco_wgemm_plan* p) {
  if (int rc = wgemm_check_plan(p)) return rc;
  ECO_REQUIRE(p->kd == 1 && p->d == 1 && p->cin % 32 == 0 && p->cin >= 64 && p->cin <= 224 && p->cout % 32 == 0,
              "wfused: 2-D layers with 64..224 input channels and output channels, both multiples of 32 (got cin=%d cout=%d d=%d kd=%d)",
              p->cin, p->cout, p->d, p->kd);
  return ECO_OK;
}

extern "C" int64_t eco_wfused_weight_elems(const eco_wgemm_plan* plan) {
  if (!plan || plan->cout <= 0 || plan->cin <= 0) return 0;
  return (int64_t)plan->points * ((plan->cout + 31) / 32) * (plan->cin / 2) * 64;
}

extern "C" int eco_wfused_pack_weights(const eco_wgemm_plan* plan, const float* u, float* up) {
  clear_error();
  if (int rc = wfused_check(plan)) return rc;
  ECO_REQUIRE(u && up, "wfused pack: null argument");
  const int cin = plan->cin, cout = plan->cout, kp = cin / 2, mblocks = cout / 32;
  // u[p][co][ci] -> up[p][co/32][(ci/2)/4][lane = (ci%2)*32 + co%32][(ci/2)%4]: the A fragments of four consecutive
  // k-pairs are one 16-byte vector per lane, consecutive lanes on consecutive vectors
  for (int p = 0; p < plan->points; ++p)
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci) {
        const int kq = ci / 2, lane = (ci & 1) * 32 + co % 32;
        up[((((long)p * mblocks + co / 32) * (kp / 4) + kq / 4) * 64 + lane) * 4 + kq % 4] = u[((long)p * cout + co) * cin + ci];
      }
  return ECO_OK;
}

static int wfused_launch(const eco_wgemm_plan* plan, const float* v, const float* up, int32_t h, int32_t w,
                         const eco_conv_epilogue* ep, float* pool9, void* stream) {
  clear_error();
  if (int rc = wfused_check(plan)) return rc;
  ECO_REQUIRE(v && up && ep && h > 0 && w > 0, "wfused: bad argument");
  ECO_REQUIRE(plan->th == (h + 3) / 4 && plan->tw == (w + 3) / 4, "wfused: plan is for %dx%d tiles", plan->th, plan->tw);
  if (pool9) {
    ECO_REQUIRE(h % 4 == 0 && w % 4 == 0, "wfused + pooling: %dx%d planes do not tile by 4", h, w);
    ECO_REQUIRE(!ep->raw.ptr && !ep->act.ptr && !ep->act2.ptr && !ep->residual.ptr,
                "wfused + pooling: only the pooled activation exists (no raw / act / act2 / residual views)");
    ECO_REQUIRE(((uintptr_t)pool9 & 3) == 0, "wfused + pooling: misaligned scratch");
  } else {
    ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "wfused: at least one of raw/act outputs is required");
  }
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "wfused: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "wfused: act2 needs act");
  ECO_REQUIRE(ep->nseg == 0, "wfused: segmented (sibling) launches exist for the direct kernels only");
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* vw : views)
    ECO_REQUIRE(!vw->ptr || (vw->t >= 1 && vw->stride_c >= 1), "wfused: view needs t >= 1 and stride_c >= 1");
  WFusedArgs a;
  a.v = v; a.u = up;
  a.o.m = nullptr; a.o.bias = ep->bias; a.o.bn_scale = ep->bn_scale; a.o.bn_shift = ep->bn_shift;
  a.o.residual = ep->residual; a.o.raw = ep->raw; a.o.act = ep->act; a.o.act2 = ep->act2; a.o.relu = ep->relu;
  a.o.n = plan->n; a.o.cout = plan->cout; a.o.D = 1; a.o.H = h; a.o.W = w; a.o.TH = plan->th; a.o.TW = plan->tw;
  a.o.NB = plan->n * plan->th * plan->tw;
  a.o.ntot = a.o.NB; a.o.ksplit = 1; a.o.m_pstride = 0;
  a.o.pool9 = pool9;
  a.mblocks = plan->cout / 32;
  a.nblk = (int)ceil_div(a.o.NB, 32);
  a.Q = a.o.NB;
  a.v_pstride = (long)plan->cin * a.o.NB;
  a.u_pstride = (long)a.mblocks * (plan->cin / 2) * 64;
  ECO_REQUIRE(((uintptr_t)v & 15) == 0 && ((uintptr_t)up & 15) == 0, "wfused: operands must be 16-byte aligned");
  int vec = 4;
  auto limit = [&](const eco_view& vw) {
    if (!vw.ptr) return;
    while (vec > 1 && (((uintptr_t)vw.ptr % (4 * vec)) || vw.stride_b % vec || vw.stride_t % vec || vw.stride_c % vec)) vec /= 2;
  };
  while (vec > 1 && w % vec) vec /= 2;
  limit(a.o.residual); limit(a.o.raw); limit(a.o.act); limit(a.o.act2);
  const long grid = (long)a.mblocks * a.nblk;
  ECO_REQUIRE(grid < 2147483647l, "wfused: too many tiles for one launch");
  const size_t lds = sizeof(float) * 12 * 32 * 32;
  hipStream_t s = (hipStream_t)stream;
#define ECO_WFUSED_LAUNCH(KP, VEC)                                                                                         \
  do {                                                                                                                     \
    ECO_WFUSED_RAISE(KP, VEC);                                                                                             \
    hipLaunchKernelGGL((wfused_kernel<KP, VEC>), dim3((unsigned)grid), dim3(256), lds, s, a);                              \
  } while (0)
#define ECO_WFUSED_RAISE(KP, VEC) ECO_RAISE_DYNAMIC_LDS((wfused_kernel<KP, VEC>), "wfused")
#define ECO_WFUSED_POOL(KP)                                                                                                \
  do {                                                                                                                     \
    ECO_RAISE_DYNAMIC_LDS((wfused_kernel<KP, 1, true>), "wfused");                                                         \
    hipLaunchKernelGGL((wfused_kernel<KP, 1, true>), dim3((unsigned)grid), dim3(256), lds, s, a);                          \
  } while (0)
#define ECO_WFUSED_KP(KP)                                                                                                  \
  case 2 * KP:                                                                                                             \
    if (pool9) ECO_WFUSED_POOL(KP);                                                                                        \
    else if (vec == 4) ECO_WFUSED_LAUNCH(KP, 4); else if (vec == 2) ECO_WFUSED_LAUNCH(KP, 2); else ECO_WFUSED_LAUNCH(KP, 1); \
    break
  switch (plan->cin) {
    ECO_WFUSED_KP(32); ECO_WFUSED_KP(48); ECO_WFUSED_KP(64); ECO_WFUSED_KP(80); ECO_WFUSED_KP(96); ECO_WFUSED_KP(112);
    default: return fail(ECO_ERR_INVALID, "wfused: unsupported cin %d", plan->cin);
  }
#undef ECO_WFUSED_KP
#undef ECO_WFUSED_LAUNCH
#undef ECO_WFUSED_POOL
#undef ECO_WFUSED_RAISE
  return check_launch("eco_wfused_forward");
}

extern "C" int eco_wfused_forward(const eco_wgemm_plan* plan, const float* v, const float* up, int32_t h, int32_t w,
                                  const eco_conv_epilogue* ep, void* stream) {
  return wfused_launch(plan, v, up, h, w, ep, nullptr, stream);
}

extern "C" int64_t eco_wfused_pool_scratch_elems(const eco_wgemm_plan* plan) {
  return plan ? (int64_t)plan->n * plan->cout * 9 * plan->th * plan->tw : 0;
}

extern "C" int eco_wfused_pool_forward(const eco_wgemm_plan* plan, const float* v, const float* up, int32_t h, int32_t w,
                                       const eco_conv_epilogue* ep, float* scratch, float* y, void* stream) {
  clear_error();
  // every argument is checked before the first launch: an error return leaves nothing enqueued (round-5 advisor finding)
  ECO_REQUIRE(scratch && y, "wfused + pooling: null scratch / output");
  ECO_REQUIRE(((uintptr_t)y & 7) == 0, "wfused + pooling: the pooled blob must be 8-byte aligned");
  if (int rc = wfused_launch(plan, v, up, h, w, ep, scratch, stream)) return rc;
  const long planes = (long)plan->n * plan->cout;
  const long total = planes * 2 * plan->th * plan->tw;
  hipLaunchKernelGGL((pool9_finish_kernel), dim3(wg_grid(total)), dim3(256), 0, (hipStream_t)stream, scratch, y, planes, plan->th,
                     plan->tw);
  return check_launch("eco_wfused_pool_forward");
}
