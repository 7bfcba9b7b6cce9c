// eco_conv.hip -- N-D convolution forward as an implicit-GEMM fp32-MFMA kernel for gfx950,
// with the bias / Eltwise-SUM residual / folded-BN / ReLU epilogue and strided
// (Concat-slice, r2Dto3D+Permute) stores fused in.
//
// Replaces, for the forward path, the reference's per-image im2col + SGEMM + bias-GEMM
//   ConvolutionLayer::Forward_{cpu,gpu}      caffe_3d/src/caffe/layers/conv_layer.cpp:28-43
//   BaseConvolutionLayer::forward_*_gemm/bias layers/base_conv_layer.cpp:264-287
//   im2col_cpu / im2col_nd_core_cpu           util/im2col.cpp:27-64, 91-158
//   (GPU: im2col_gpu_kernel / im2col_nd_gpu_kernel util/im2col.cu:12-161, cuDNN conv
//    layers/cudnn_conv_layer.cu:15-65)
// and the layers the executor fuses behind it (BN bn_layer.cpp:93-207, ReLU
// relu_layer.cpp:10-20, Eltwise eltwise_layer.cpp:66-72, Concat concat_layer.cpp:54-70,
// Reshape+Permute reshape_layer.cpp:88 / permute_layer.cpp:9-26).
//
// GEMM view (no col buffer is ever materialised):
//   Y[m, n] = sum_k Wp[k, m] * X[base(n) + koff(k)] * valid(n, tap(k))
//   m = output channel, n = flattened (image, od, oh, ow) output position,
//   k = c*taps + tap  (the reference's own weight order [cout][cin][kd][kh][kw]).
// Per workgroup (256 threads = 4 waves): a BM x BN output tile; the reduction runs in
// stages of KC rows.  Each stage: Wp rows are fetched as float4 (packed K-major, so rows
// are contiguous), the im2col rows are *gathered* straight from the NC[D]HW input with a
// per-position validity bit mask (zero padding) and a per-k offset table, both are staged
// through double-buffered LDS, and every wave accumulates TM x TN 32x32 tiles with
// v_mfma_f32_32x32x2_f32.  LDS layout is [k][m] / [k][n], so a wave's A/B fragment reads
// are 32 consecutive floats per half-wave: bank-conflict free ds_read_b32.
// Output tile: lane = output position (32 consecutive positions per half-wave), register =
// output channel -> each store instruction writes 128 contiguous bytes per half-wave into
// the N,C,[D,]H,W destination.
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "eco_common.h"

namespace eco {

struct ConvKernelArgs {
  const float* x;
  const float* wp;
  const int32_t* ktab;
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;
  // sibling convs run as one (eco_conv_epilogue::nseg): 32-row tiles at or above seg_begin[s] write through
  // seg_act[s], whose ptr the host moved back by seg_begin[s] channels so that the global channel indexes it
  int nseg, seg_begin[ECO_MAX_SEG], seg_relu[ECO_MAX_SEG];
  eco_view seg_act[ECO_MAX_SEG];
  int relu;
  int cin, cout, mpad, kpad;
  int Di, Hi, Wi, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int s_in, s_out;      // Di*Hi*Wi, Do*Ho*Wo
  long img_stride_in;   // cin * s_in
  int ntot;             // n * s_out output positions
  int nblk_m, nblk_n;
  // Split-K region: the last n_split tiles (all of them for a full split) have their reduction cut into
  // ksplit slices; partial sums go to ws[slice][cout][n - n_split0] and a second launch reduces them.
  int ksplit, n_main, n_split;   // n_main + n_split = nblk_m * nblk_n
  int n_split0;                  // first output position covered by the split region
  // Order of the flattened output positions n: 0 = (img, d, h, w) as in memory; 1 = depth-major (d, img, h, w)
  // (span kernel, 3-D convs): a tile then holds one depth plane of several clips, and the tiles of the first /
  // last plane skip the depth taps that only see zero padding.
  int dmajor, n_img, bn_tile;
  // Split-K slices per tile (span kernel): tile columns [col_long0, col_long1) are cut into ns_long slices,
  // the others (depth-major: tiles of the first / last depth plane, 1/3 of the depth taps dead) into
  // ns_short <= ns_long.  Gather kernels: every split tile has ksplit slices (ns_short = ns_long = ksplit).
  int col_long0, col_long1, ns_short, ns_long;
  float* ws;
  // Stream-K launches (conv_streamk_kernel): sk_wgs persistent workgroups share sk_units = stages of all tiles;
  // sk_cum[col] = stages of one M-block's tiles in columns < col (int32 behind the packed gather table);
  // sk_flags[w] = 1 once workgroup w's partial block ws[w][BM*BN] is in memory.
  int sk_wgs;
  long sk_units;
  const int32_t* sk_cum;
  int* sk_flags;
  // Batched launches (gridDim.y = batch; the (M+2)^2 transform points of the Winograd path): element strides of
  // the input, the packed weights, the output views and the split-K workspace between batch entries.
  long bstride_x, bstride_w, bstride_out, bstride_ws;
  int batch;
};

// The launch's arguments for batch entry blockIdx.y.
__device__ __forceinline__ ConvKernelArgs batch_args(const ConvKernelArgs& a0) {
  ConvKernelArgs a = a0;
  const long bz = (long)blockIdx.y;
  a.x += bz * a0.bstride_x;
  a.wp += bz * a0.bstride_w;
  if (a.residual.ptr) a.residual.ptr += bz * a0.bstride_out;
  if (a.raw.ptr) a.raw.ptr += bz * a0.bstride_out;
  if (a.act.ptr) a.act.ptr += bz * a0.bstride_out;
  if (a.act2.ptr) a.act2.ptr += bz * a0.bstride_out;
  if (a.ws) a.ws += bz * a0.bstride_ws;
  return a;
}

constexpr int kKoffBits = 26;
constexpr int kKoffMask = (1 << kKoffBits) - 1;
constexpr int kNeverTap = 63;  // validity-mask bit that is never set (used by K padding)

// Compute units of the calling thread's current device (hipDeviceProp_t::multiProcessorCount, what
// eco_device_info reports); plans are sized against this unless the caller names a count.


// Output position n -> (image, spatial index) under the launch's position order.
__device__ __forceinline__ void decode_pos(const ConvKernelArgs& a, int n, int& img, int& sp) {
  if (a.dmajor) {
    const int phw = a.Ho * a.Wo, per_d = a.n_img * phw;
    const int d = n / per_d, r = n - d * per_d;
    img = r / phw;
    sp = d * phw + (r - img * phw);
  } else {
    img = n / a.s_out;
    sp = n - img * a.s_out;
  }
}

// Depth-major launches: the depth taps [zlo, zhi] that touch at least one real input plane for some position
// of the tile [n0, n0 + bn) (the tile lies in output depth planes dlo..dhi; tap z reads input plane
// d*sd + z - pd).  Other position orders: every tap.
__device__ __forceinline__ void live_depth_taps(const ConvKernelArgs& a, int n0, int bn, int& zlo, int& zhi) {
  zlo = 0;
  zhi = a.kd - 1;
  if (a.dmajor) {
    const int per_d = a.n_img * a.Ho * a.Wo;
    const int nlast = (n0 + bn < a.ntot ? n0 + bn : a.ntot) - 1;
    const int dlo = n0 / per_d, dhi = nlast / per_d;
    if (a.pd - dhi * a.sd > 0) zlo = a.pd - dhi * a.sd;
    if (a.Di - 1 + a.pd - dlo * a.sd < zhi) zhi = a.Di - 1 + a.pd - dlo * a.sd;
  }
}
// Split-K slices of the tiles of column `col` (see ConvKernelArgs).
__device__ __forceinline__ int col_slices(const ConvKernelArgs& a, int col) {
  return (col >= a.col_long0 && col < a.col_long1) ? a.ns_long : a.ns_short;
}

// Epilogue shared by both kernels: bias, Eltwise-SUM residual, raw store, folded BN, ReLU,
// activated store.  `acc[i][j]` is the wave's (i,j) 32x32 tile: register r of lane l holds
// channel mw + i*32 + (r&3) + 8*(r>>2) + 4*(l>>5) at position nw + j*32 + (l&31).
// Work is ordered tile-row (i) -> 4-channel register group (g) -> tile-column (j) so that only
// 12 per-channel parameters and 8 values are live at a time (keeps the kernel at the main
// loop's register budget), and every batch of loads is issued before the stores that follow.
// Ep (optional): the workgroup's per-channel parameters staged in LDS by conv_stage_params -- Ep[0..] bias,
// Ep[EPS..] BN scale, Ep[2 EPS..] BN shift of channels m0 .. (defaults past cout); nullptr = load them from global memory
// here.  The loads sit at the head of a dependent chain (parameters -> FMA -> store) of twelve round trips per tile: on
// the 12-stage 1x1 reductions staging them at kernel start is worth 8 % (profiles/r03_notes.md).
template <int TM, int TN, bool PLAIN>
__device__ __forceinline__ void conv_epilogue_impl(const ConvKernelArgs& a, f32x16 (&acc)[TM][TN], int mw, int nw,
                                                   int half, int l31, const float* Ep = nullptr, int EPS = 0, int m0 = 0) {
  long e_res[TN], e_raw[TN], e_act[TN], e_act2[TN];
  int e_img[TN], e_sp[TN];
  bool e_ok[TN];
  const bool has_bias = a.bias != nullptr, has_bn = a.bn_scale != nullptr, has_res = a.residual.ptr != nullptr;
  const bool has_raw = a.raw.ptr != nullptr, has_act = a.act.ptr != nullptr, has_act2 = has_act && a.act2.ptr != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + l31;
    e_ok[j] = n < a.ntot;
    const int nn = e_ok[j] ? n : 0;
    int img, sp;
    decode_pos(a, nn, img, sp);
    e_res[j] = has_res ? view_base(a.residual, img, sp) : 0;
    e_raw[j] = has_raw ? view_base(a.raw, img, sp) : 0;
    e_act[j] = has_act ? view_base(a.act, img, sp) : 0;
    e_act2[j] = has_act2 ? view_base(a.act2, img, sp) : 0;
    e_img[j] = img; e_sp[j] = sp;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // the destination of this 32-row tile (wave-uniform): `act`, or a sibling's own tensor
    float* aptr = a.act.ptr;
    long astride_c = a.act.stride_c;
    int relu = a.relu;
    const int mt = mw + i * 32;
    if (a.nseg > 0 && mt >= a.seg_begin[0]) {
      // (constant indices only: a run-time index into the kernel argument struct makes the compiler copy it to scratch)
      // every candidate is loaded (constant indices, scalar loads) and the VALUES are selected: a conditional load, or
      // a run-time index, into the kernel argument struct makes the compiler copy the struct to scratch memory
      long sstride_b = a.seg_act[0].stride_b;
      aptr = a.seg_act[0].ptr; astride_c = a.seg_act[0].stride_c; relu = a.seg_relu[0];
#pragma unroll
      for (int q = 1; q < ECO_MAX_SEG; ++q) {
        float* const qp = a.seg_act[q].ptr;
        const long qc = a.seg_act[q].stride_c, qb = a.seg_act[q].stride_b;
        const int qr = a.seg_relu[q];
        const bool take = q < a.nseg && mt >= a.seg_begin[q];
        aptr = take ? qp : aptr; astride_c = take ? qc : astride_c; sstride_b = take ? qb : sstride_b; relu = take ? qr : relu;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) e_act[j] = (long)e_img[j] * sstride_b + e_sp[j];
    }   // (tiles ascend: once past seg_begin[0] a wave never returns to `act`, whose offsets e_act held so far)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int chg = mw + i * 32 + 8 * g + 4 * half;  // first of this group's 4 consecutive channels
      float pb[4], ps[4], ph[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chs = (chg + q) < a.cout ? (chg + q) : 0;
        if (Ep) {
          pb[q] = Ep[chg + q - m0]; ps[q] = Ep[EPS + chg + q - m0]; ph[q] = Ep[2 * EPS + chg + q - m0];
        } else {
          pb[q] = has_bias ? ld(a.bias + chs) : 0.0f;
          ps[q] = has_bn ? ld(a.bn_scale + chs) : 1.0f;
          ph[q] = has_bn ? ld(a.bn_shift + chs) : 0.0f;
        }
      }
      // Beside f32 MFMAs every VALU instruction is matrix-pipe time (profiles/r03_notes.md), and the 1x1 convolutions
      // have one output per 100-160 MFMA k-steps: when the value v = acc + bias is not itself needed (no raw store, no
      // residual) the bias goes into the shift once per channel -- (acc + b)*s + h = acc*s + (b*s + h) -- and ReLU is one
      // v_max against 0 or -FLT_MAX instead of a compare-and-select.
      constexpr bool plain_act = PLAIN;          // (!has_res && !has_raw, decided once per kernel: conv_epilogue below)
      const float floor_v = relu ? 0.0f : -FLT_MAX;
      float ph2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ph2[q] = pb[q] * ps[q] + ph[q];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!e_ok[j]) continue;
        if constexpr (plain_act) {
          if (has_act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float y = fmaxf(acc[i][j][4 * g + q] * ps[q] + ph2[q], floor_v);
              if (chg + q < a.cout) {
                st(aptr + e_act[j] + (long)(chg + q) * astride_c, y);
                if (has_act2) st(a.act2.ptr + e_act2[j] + (long)(chg + q) * a.act2.stride_c, y);
              }
            }
          }
          continue;
        }
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + pb[q];
        if (has_res) {
          float rv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rv[q] = ld((const float*)a.residual.ptr + e_res[j] +
                       (long)((chg + q) < a.cout ? (chg + q) : 0) * a.residual.stride_c);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[q];
        }
        if (has_raw) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (chg + q < a.cout) st(a.raw.ptr + e_raw[j] + (long)(chg + q) * a.raw.stride_c, v[q]);
        }
        if (has_act) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float y = fmaxf(v[q] * ps[q] + ph[q], floor_v);
            if (chg + q < a.cout) {
              st(aptr + e_act[j] + (long)(chg + q) * astride_c, y);
              if (has_act2) st(a.act2.ptr + e_act2[j] + (long)(chg + q) * a.act2.stride_c, y);
            }
          }
        }
      }
    }
  }
}

// The two forms as separate code (one run-time branch per kernel; as one loop nest the unroller gave up on the doubled
// body and the accumulators went to scratch memory).
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvKernelArgs& a, f32x16 (&acc)[TM][TN], int mw, int nw, int half,
                                              int l31, const float* Ep = nullptr, int EPS = 0, int m0 = 0) {
  if (a.residual.ptr == nullptr && a.raw.ptr == nullptr) conv_epilogue_impl<TM, TN, true>(a, acc, mw, nw, half, l31, Ep, EPS, m0);
  else conv_epilogue_impl<TM, TN, false>(a, acc, mw, nw, half, l31, Ep, EPS, m0);
}

// bias / BN scale / BN shift of channels m0 .. m0 + BMP - 1 into LDS (0 / 1 / 0 where absent or past cout) when the
// kernel starts; visible after the caller's next workgroup barrier.  (Plain loads + ds_write: an LDS-DMA form of the
// same -- no registers, nothing waited for -- measured slower on both paths, profiles/r03_notes.md.)
template <int BMP>
__device__ __forceinline__ void conv_stage_params(const ConvKernelArgs& a, int m0, float* Ep) {
  const int t = (int)threadIdx.x;
  if (t < BMP) {
    const int ch = m0 + t;
    const bool in = ch < a.cout;
    Ep[t] = (in && a.bias) ? ld(a.bias + ch) : 0.0f;
    Ep[BMP + t] = (in && a.bn_scale) ? ld(a.bn_scale + ch) : 1.0f;
    Ep[2 * BMP + t] = (in && a.bn_scale) ? ld(a.bn_shift + ch) : 0.0f;
  }
}

// Split-K: a workgroup that only covered a slice of the reduction stores its raw accumulators to
// ws[slice][channel][position] (positions contiguous: 128 B per half-wave, like the real epilogue).
template <int TM, int TN>
__device__ __forceinline__ void conv_store_partial(const ConvKernelArgs& a, f32x16 (&acc)[TM][TN], int slice, int mw,
                                                   int nw, int half, int l31) {
  const int nsplit_pos = a.ntot - a.n_split0;
  float* base = a.ws + (long)slice * a.cout * nsplit_pos - a.n_split0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + l31;
    if (n >= a.ntot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ch < a.cout) st(base + (long)ch * nsplit_pos + n, acc[i][j][r]);
      }
  }
}

// Second pass of split-K: sum the slices in a fixed order (deterministic) and apply the epilogue.
// VEC = 4 handles four consecutive positions per thread with 16-byte accesses (the launcher checks that
// positions n..n+3 are contiguous in every view: s_out, the region bounds and all strides are multiples
// of 4 and the pointers are 16-byte aligned); VEC = 1 is the general form.  HBM-bound: S slices read +
// outputs written once.
template <int VEC>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvKernelArgs a0) {
  const ConvKernelArgs a = batch_args(a0);
  const int nsplit_pos = a.ntot - a.n_split0;
  const long total = (long)a.cout * nsplit_pos;
  const long slice_stride = total;
  for (long idx = ((long)blockIdx.x * 256 + threadIdx.x) * VEC; idx < total; idx += (long)gridDim.x * 256 * VEC) {
    const int ch = (int)(idx / nsplit_pos), n = a.n_split0 + (int)(idx - (long)ch * nsplit_pos);
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.0f;
    const int nsl = col_slices(a, n / a.bn_tile);  // tiles with dead depth taps were cut into fewer slices
    for (int sidx = 0; sidx < nsl; ++sidx) {
      const float* p = (const float*)a.ws + sidx * slice_stride + idx;
      if (VEC == 4) {
        const float4 q = ld((const float4*)p);
        v[0] += q.x; v[1 % VEC] += q.y; v[2 % VEC] += q.z; v[3 % VEC] += q.w;
      } else {
        v[0] += ld(p);
      }
    }
    int img, sp;
    decode_pos(a, n, img, sp);
    const float b = a.bias ? ld(a.bias + ch) : 0.0f;
    // (the BN parameters too, ahead of the raw store: behind it their loads would be a round trip of their own -- the
    // store may alias them for all the compiler knows)
    const float sc = (a.act.ptr && a.bn_scale) ? ld(a.bn_scale + ch) : 1.0f, sh = (a.act.ptr && a.bn_scale) ? ld(a.bn_shift + ch) : 0.0f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] += b;
    if (a.residual.ptr) {
      const float* r = (const float*)a.residual.ptr + view_base(a.residual, img, sp) + (long)ch * a.residual.stride_c;
      if (VEC == 4) {
        const float4 q = ld((const float4*)r);
        v[0] += q.x; v[1 % VEC] += q.y; v[2 % VEC] += q.z; v[3 % VEC] += q.w;
      } else {
        v[0] += ld(r);
      }
    }
    if (a.raw.ptr) {
      float* o = a.raw.ptr + view_base(a.raw, img, sp) + (long)ch * a.raw.stride_c;
      if (VEC == 4) st((float4*)o, make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]));
      else st(o, v[0]);
    }
    if (a.act.ptr) {
      eco_view av = a.act;       // sibling launches: the channel's own destination (ptr already moved back)
      int relu = a.relu;
      if (a.nseg > 0 && ch >= a.seg_begin[0]) {
        av = a.seg_act[0]; relu = a.seg_relu[0];
#pragma unroll
        for (int q = 1; q < ECO_MAX_SEG; ++q) {
          const eco_view qv = a.seg_act[q];
          const int qr = a.seg_relu[q];
          const bool take = q < a.nseg && ch >= a.seg_begin[q];
          av.ptr = take ? qv.ptr : av.ptr; av.stride_b = take ? qv.stride_b : av.stride_b;
          av.stride_c = take ? qv.stride_c : av.stride_c; av.stride_t = take ? qv.stride_t : av.stride_t;
          av.t = take ? qv.t : av.t; relu = take ? qr : relu;
        }
      }
      float y[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        y[e] = v[e] * sc + sh;
        if (relu) y[e] = fmaxf(y[e], 0.0f);
      }
      float* o = av.ptr + view_base(av, img, sp) + (long)ch * av.stride_c;
      if (VEC == 4) st((float4*)o, make_float4(y[0], y[1 % VEC], y[2 % VEC], y[3 % VEC]));
      else st(o, y[0]);
      if (a.act2.ptr) {
        float* o2 = a.act2.ptr + view_base(a.act2, img, sp) + (long)ch * a.act2.stride_c;
        if (VEC == 4) st((float4*)o2, make_float4(y[0], y[1 % VEC], y[2 % VEC], y[3 % VEC]));
        else st(o2, y[0]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The convolution kernel.  MODE selects how the im2col rows of a stage are addressed:
//
// MODE = ECO_CONV_MODE_CTAP ("channel-tile / constant-tap", cin % KC == 0; every ECO conv except
// conv1_7x7_s2).  Reduction order k' = (cc*taps + tap)*KC + ci with channel c = cc*KC + ci: all KC
// rows of a stage share one kernel tap, so
//   * the zero-padding predicate is one bit test per thread per stage (not per element),
//   * the gather address is  x + [uniform: (cc*KC + ci)*s_in + tap offset] + [per-thread: base(n)]
//     -> a scalar base + one 32-bit vector offset per load, no per-element VALU and no table,
//   * consecutive stages walk the 27 (9, 1) taps of the same KC channels, i.e. the same cache
//     lines shifted by a tap -> the im2col re-reads are L1/L2 hits.
// MODE = ECO_CONV_MODE_TABLE (any cin; conv1: cin = 3, K = 147).  Reference order k = c*taps + tap;
// each row's input offset and tap come from the int32 table ktab[k] = tap << 26 | offset, fetched
// with scalar loads two stages ahead, and the padding predicate is per element.
//
// In both modes the next stage's global loads are issued *between* the MFMAs of the current stage
// (one or two per k-pair step), the fragment registers are double-buffered so the ds_reads of step
// kk+1 are in flight under the MFMAs of step kk, and the only non-overlapped work per stage is the
// LDS write of the prefetched registers plus one barrier.
//
// Register budget: 64 accumulators per 2x2 wave tile leave ~100 VGPRs for four workgroups per CU;
// the second launch-bound argument (waves per SIMD) holds the allocator to that.
// The reduction of stages [c_begin, c_end) of the output tile at (m0, n0) into `acc` (cleared here unless keep_acc: the
// stream-K kernel starts a shared tile's last segment from the other workgroups' partial sums): everything of
// conv_mfma_kernel between "which tile, which stages" and "what to do with the sums", so that the one-tile-per-workgroup
// kernel and the persistent stream-K kernel below share it.  zlo / zhi / tap_lo / taps_live: the tile's live depth taps.
template <int TM, int TN, int WM, int WN, int KC, int MODE>
__device__ __forceinline__ void conv_reduce_segment(const ConvKernelArgs& a, int m0, int n0, int c_begin, int c_end, int zlo,
                                                    int zhi, int nstages_all, float (&As)[2][KC][32 * TM * WM],
                                                    float (&Bs)[2][KC][32 * TN * WN], f32x16 (&acc)[TM][TN],
                                                    bool keep_acc = false) {
  constexpr bool CTAP = MODE == ECO_CONV_MODE_CTAP;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int KG = 256 / BN;         // threads sharing one output position in the gather
  constexpr int EPT = KC / KG;         // gathered elements per thread per stage
  constexpr int KSTEPS = KC / 2;       // MFMA k-pair steps per stage
  constexpr int BPS = EPT / KSTEPS;    // gather loads issued per k-pair step (1 or 2)
  static_assert(EPT % KSTEPS == 0 && BPS >= 1, "");
  constexpr int A_F4 = KC * BM / 4;
  constexpr int A_ITERS = (A_F4 + 255) / 256;
  static_assert(A_ITERS <= KSTEPS, "");
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int taps = a.kd * a.kh * a.kw, khw = a.kh * a.kw;
  const int tap_lo = zlo * khw, taps_live = (zhi - zlo + 1) * khw;
  (void)nstages_all;

  // ---- gather role of this thread: one output position, EPT of the stage's KC rows ----
  const int pos_l = tid % BN;
  const int kg = uniform(tid / BN);
  int in_base = 0;
  unsigned long long mask = 0ull;
  {
    const int n = n0 + pos_l;
    if (n < a.ntot) {
      int img, sp;
      decode_pos(a, n, img, sp);
      const int ow = sp % a.Wo, t = sp / a.Wo;
      const int oh = t % a.Ho, od = t / a.Ho;
      const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
      in_base = (int)((long)img * a.img_stride_in + ((long)id0 * a.Hi + ih0) * a.Wi + iw0);
      // tap-validity mask, built separably (kw + kh + kd steps instead of kd*kh*kw): bit
      // ((z*kh + y)*kw + x) is set iff input (id0+z, ih0+y, iw0+x) lies inside the image.
      unsigned long long mw = 0ull, mhw = 0ull;
      for (int xx = 0; xx < a.kw; ++xx) mw |= (unsigned long long)((unsigned)(iw0 + xx) < (unsigned)a.Wi) << xx;
      for (int y = 0; y < a.kh; ++y)
        if ((unsigned)(ih0 + y) < (unsigned)a.Hi) mhw |= mw << (y * a.kw);
      for (int z = 0; z < a.kd; ++z)
        if ((unsigned)(id0 + z) < (unsigned)a.Di) mask |= mhw << (z * khw);
    }
  }

  // ---- the stage being loaded ----
  // CTAP: uniform (cc, tap) walk over the live taps + this thread's predicate / offset
  int l_cc = CTAP ? c_begin / taps_live : 0, l_tap = CTAP ? tap_lo + c_begin % taps_live : 0;
  int l_kx = l_tap % a.kw, l_ky = (l_tap / a.kw) % a.kh, l_kz = l_tap / khw;
  // uniform: packed-weight rows of the stage being loaded
  const float* l_wp = a.wp + m0 + (CTAP ? (long)l_cc * taps + l_tap : (long)c_begin) * KC * a.mpad;
  const float* l_xb = a.x;           // uniform: x + cc*KC*s_in + tap offset
  int l_voff = 0;                    // per thread: base(n) if the tap is inside the image, else the
  unsigned l_ok = 0;                 // offset back to the start of the channel plane (always in bounds)
  // TABLE: this thread's table entries (uniform across the wave) for the stage being loaded and the next
  int kt_cur[EPT], kt_nxt[EPT];
  auto fetch_table = [&](int stage, int (&dst)[EPT]) {
    const int sidx = stage < nstages_all ? stage : nstages_all - 1;
#pragma unroll
    for (int j = 0; j < EPT; ++j) dst[j] = ld(a.ktab + sidx * KC + kg + j * KG);
  };
  auto begin_stage = [&]() {
    if (CTAP) {
      const int toff = (l_kz * a.Hi + l_ky) * a.Wi + l_kx;
      l_xb = a.x + ((long)l_cc * KC * a.s_in + toff);
      l_ok = (mask >> l_tap) & 1ull ? ~0u : 0u;
      l_voff = l_ok ? in_base : -toff;
    } else {
      l_ok = 0;
    }
  };
  auto next_stage = [&](int stage) {  // `stage` = index of the stage that becomes "being loaded"
    l_wp += (long)KC * a.mpad;
    if (CTAP) {
      ++l_tap;
      if (++l_kx == a.kw) {
        l_kx = 0;
        if (++l_ky == a.kh) {
          l_ky = 0;
          if (++l_kz > zhi) {  // next channel tile: back to the first live tap, skipping the dead ones' rows
            l_kz = zlo;
            l_tap = tap_lo;
            ++l_cc;
            l_wp += (long)(taps - taps_live) * KC * a.mpad;
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < EPT; ++j) kt_cur[j] = kt_nxt[j];
      fetch_table(stage + 1, kt_nxt);
    }
    begin_stage();
  };

  float4 areg[A_ITERS];
  float breg[EPT];
  auto load_a = [&](int i) {
    const int idx = tid + i * 256;
    if (A_F4 % 256 == 0 || idx < A_F4) {
      const int row = idx / (BM / 4), c4 = idx % (BM / 4);
      areg[i] = ld((const float4*)(l_wp + (long)row * a.mpad + c4 * 4));
    }
  };
  auto load_b = [&](int j) {
    if (CTAP) {
      breg[j] = ld(l_xb + (long)(kg + j * KG) * a.s_in + l_voff);
    } else {
      const int off = kt_cur[j] & kKoffMask;
      const unsigned tap = (unsigned)kt_cur[j] >> kKoffBits;
      const unsigned ok = (unsigned)(mask >> tap) & 1u;
      l_ok |= ok << j;
      breg[j] = ld(a.x + (ok ? in_base + off : 0));
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int idx = tid + i * 256;
      if (A_F4 % 256 == 0 || idx < A_F4) {
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        *(float4*)&As[buf][row][c4 * 4] = areg[i];
      }
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) Bs[buf][kg + j * KG][pos_l] = ((l_ok >> j) & 1u) ? breg[j] : 0.0f;
  };

  if (!keep_acc) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  // Fragment registers are double-buffered across k-pair steps: the ds_reads of step kk+1 are issued
  // before the MFMAs of step kk, so a wave does not sit on LDS latency between MFMA groups.
  float af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int kk, int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i) af[slot][i] = As[buf][2 * kk + half][(wm * TM + i) * 32 + l31];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[slot][j] = Bs[buf][2 * kk + half][(wn * TN + j) * 32 + l31];
  };
  auto mfma_step = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(af[slot][i], bf[slot][j], acc[i][j]);
  };

  if (!CTAP) {
    fetch_table(c_begin, kt_cur);
    fetch_table(c_begin + 1, kt_nxt);
  }
  begin_stage();
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) load_a(i);
#pragma unroll
  for (int j = 0; j < EPT; ++j) load_b(j);
  store_stage(0);
  __syncthreads();
  for (int c = c_begin; c + 1 < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    next_stage(c + 1);
    read_frags(buf, 0, 0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      if (kk + 1 < KSTEPS) read_frags(buf, kk + 1, (kk + 1) & 1);
      // all of the next stage's loads go out in the first half of this stage (2*BPS gathers per step),
      // so even the last one has half a stage of MFMAs to land before store_stage needs it
      if (kk < A_ITERS) load_a(kk);
      if (kk < KSTEPS / 2) {
#pragma unroll
        for (int q = 0; q < 2 * BPS; ++q) load_b(kk * 2 * BPS + q);
      }
      mfma_step(kk & 1);
      sched_fence();
    }
    store_stage(buf ^ 1);
    __syncthreads();
  }
  {
    const int buf = (c_end - 1 - c_begin) & 1;
    read_frags(buf, 0, 0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      if (kk + 1 < KSTEPS) read_frags(buf, kk + 1, (kk + 1) & 1);
      mfma_step(kk & 1);
      sched_fence();
    }
  }
}

template <int TM, int TN, int WM, int WN, int KC, int MODE>
__global__ __launch_bounds__(256, (TM * TN <= 4 ? 3 : 2)) void conv_mfma_kernel(const ConvKernelArgs a0) {
  ECO_CLOCK("conv_mfma");
  const ConvKernelArgs a = batch_args(a0);
  constexpr bool CTAP = MODE == ECO_CONV_MODE_CTAP;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BN == 128 || BN == 256, "");
  constexpr int KG = 256 / BN;         // threads sharing one output position in the gather
  constexpr int EPT = KC / KG;         // gathered elements per thread per stage
  constexpr int KSTEPS = KC / 2;       // MFMA k-pair steps per stage
  constexpr int BPS = EPT / KSTEPS;    // gather loads issued per k-pair step (1 or 2)
  static_assert(EPT % KSTEPS == 0 && BPS >= 1, "");
  constexpr int A_F4 = KC * BM / 4;
  constexpr int A_ITERS = (A_F4 + 255) / 256;
  static_assert(A_ITERS <= KSTEPS, "");

  __shared__ __attribute__((aligned(16))) float As[2][KC][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][KC][BN];

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  // Hardware blocks [0, n_main) are whole tiles; the rest are (slice, tile) pairs of the split-K region,
  // slice-major so neighbours share the K range.  Each range gets its own XCD-contiguous remap, and the
  // split region comes last in dispatch order so its short blocks fill the tail of the launch.
  int tile, slice, nslices;
  if ((int)blockIdx.x < a.n_main) {
    tile = xcd_remap((int)blockIdx.x, a.n_main);
    slice = 0;
    nslices = 1;
  } else {
    // (slice, tile) pairs that exist, slice-major: slices [0, ns_short) of every split tile, then slices
    // [ns_short, ns_long) of the long columns only (depth-major launches: tiles of the first / last depth
    // plane have fewer live taps and fewer slices) -- numbered densely so that the XCD remap deals every
    // XCD the same number of live workgroups.  Other launches: ns_short = ns_long = ksplit.
    const int c0 = a.n_main / a.nblk_m;
    const int cl0 = a.col_long0 > c0 ? a.col_long0 : c0;
    const int n_long = (a.col_long1 > cl0 ? a.col_long1 - cl0 : 0) * a.nblk_m;
    const int n_all = a.n_split * a.ns_short;
    const int lid = xcd_remap((int)blockIdx.x - a.n_main, n_all + n_long * (a.ns_long - a.ns_short));
    if (lid < n_all) {
      slice = lid / a.n_split;
      tile = a.n_main + (lid - slice * a.n_split);
    } else {
      const int r = lid - n_all;
      slice = a.ns_short + r / n_long;
      tile = cl0 * a.nblk_m + r % n_long;
    }
    nslices = col_slices(a, tile / a.nblk_m);
  }
  const bool sliced = (int)blockIdx.x >= a.n_main;
  const int mblk = tile % a.nblk_m, nblk = tile / a.nblk_m;
  const int m0 = mblk * BM, n0 = nblk * BN;
  // Reduction work list: stages (channel tile cc, tap) with the tap's depth index in the live range of this
  // tile (depth-major launches skip the depth taps that see only padding; otherwise every tap), cc-major.
  const int khw = a.kh * a.kw;
  int zlo, zhi;
  live_depth_taps(a, n0, BN, zlo, zhi);
  const int taps_live = (zhi - zlo + 1) * khw;
  const int nstages_all = CTAP ? (a.cin / KC) * taps_live : a.kpad / KC;
  const int c_begin = (int)((long)slice * nstages_all / nslices);
  const int c_end = (int)((long)(slice + 1) * nstages_all / nslices);

  f32x16 acc[TM][TN];
  conv_reduce_segment<TM, TN, WM, WN, KC, MODE>(a, m0, n0, c_begin, c_end, zlo, zhi, nstages_all, As, Bs, acc);
  if (sliced)
    conv_store_partial<TM, TN>(a, acc, slice, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);
  else
    conv_epilogue<TM, TN>(a, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);
}

// ------------------------------------------------------------------------------------------
// Stream-K form of the gather kernel (round 3; the strided 3x3x3 convs res4a_1 / res4a_down / res5a_1 / res5a_down).
// One tile per workgroup leaves these launches with 392 (100) tiles for 256 CUs; cutting every tile's reduction into
// 4 (5) slices, as rounds 1-2 did, balances them to 6.1 -> 7 workgroups on the fullest CU (87 %), writes and re-reads
// every partial sum (205 MB per res4a launch) and needs a second launch to reduce them.  Here sk_wgs persistent
// workgroups (two per CU) each take an equal share of the launch's STAGES -- unit u = (tile, stage), tiles in
// (column, M-block) order with the column's own live stage count (depth-major launches skip dead depth taps) --
// wherever the tile boundaries fall.  A workgroup walks its range from the END.  Whole tiles go straight to the
// epilogue.  A tile shared with other workgroups leaves as a partial block in the workspace (16-byte write-through
// stores): block 2w+1 with a published flag when the tile's last stage lies in a later workgroup (always the range's
// first segment, so it is out early), block 2w when it lies here (always the last segment).  After its loop the
// workgroup that owns a shared tile's last stage rebuilds the sum from memory -- its own block, then the hand-off blocks
// of the earlier workgroups, nearest first down to the one with the tile's first stage: an order fixed by the geometry,
// not by timing -- and runs the epilogue.  (Adding the others' partials to the live accumulators instead, before or
// after the main loop, either serialised the finisher behind its neighbours or spilled 330 registers.)  Waits only ever
// point at lower-numbered workgroups, which the hardware dispatches first.  No second launch.
//
// MEASURED (profiles/r03_notes.md): correct, balanced -- and no faster than the split launches (2.39 against 2.42 ms for
// the four convs).  Workgroups at different reduction stages of different tiles read different rows of the packed weights
// and different input planes at the same time: L2 hit rate 0.65 against 0.93, 6.7x the HBM reads, 16 % more time per
// stage -- what the even shares win, the lost locality spends.  The planner therefore keeps the split launches
// (slice-major order: every resident workgroup is in the same part of the reduction); ECO_STREAMK=1 selects this kernel.
template <int TM, int TN, int WM, int WN, int KC>
__global__ __launch_bounds__(256, 2) void conv_streamk_kernel(const ConvKernelArgs a) {
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float As[2][KC][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][KC][BN];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
#ifdef ECO_EMU
  const int w = (int)blockIdx.x;        // the emulator runs blocks in index order, possibly on one host thread
#else
  const int w = xcd_remap((int)blockIdx.x, a.sk_wgs);   // neighbouring tiles on one XCD
#endif
  const long U = a.sk_units;
  const long u0 = U * w / a.sk_wgs, u1 = U * (w + 1) / a.sk_wgs;

  int fin_m0 = -1, fin_n0 = 0;       // the shared tile this workgroup finishes (if any) and the unit of its first stage
  long fin_first = 0;
  long u_hi = u1;
  while (u_hi > u0) {
    // the tile that holds unit u_hi - 1: column by bisection over the prefix table (scaled by the M-blocks)
    int lo = 0, hi = a.nblk_n;           // invariant: cum[lo] * nblk_m <= u_hi - 1 < cum[hi] * nblk_m
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((long)ld(a.sk_cum + mid) * a.nblk_m <= u_hi - 1) lo = mid; else hi = mid;
    }
    const int col = lo;
    const long col_u0 = (long)ld(a.sk_cum + col) * a.nblk_m;
    const int ns = ld(a.sk_cum + col + 1) - ld(a.sk_cum + col);        // stages of a tile of this column
    const int mblk = (int)((u_hi - 1 - col_u0) / ns);
    const long tile_u0 = col_u0 + (long)mblk * ns;
    const long seg_lo = u0 > tile_u0 ? u0 : tile_u0;
    const int c_begin = (int)(seg_lo - tile_u0), c_end = (int)(u_hi - tile_u0);
    const int m0 = mblk * BM, n0 = col * BN;
    int zlo, zhi;
    live_depth_taps(a, n0, BN, zlo, zhi);

    f32x16 acc[TM][TN];
    conv_reduce_segment<TM, TN, WM, WN, KC, ECO_CONV_MODE_CTAP>(a, m0, n0, c_begin, c_end, zlo, zhi, ns, As, Bs, acc);

    if (c_begin == 0 && c_end == ns) {
      conv_epilogue<TM, TN>(a, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);      // a whole tile
    } else {
      // A shared tile: this range's stages go to the workspace as a partial block -- block 2w + 1 if the tile's last
      // stage lies in a later workgroup (always this range's first segment: handed off, flag published), block 2w if
      // it lies here (always the last segment: this workgroup finishes the tile below, once the others are in).
      const bool handoff = c_end < ns;
      float* pb = a.ws + (((long)(2 * w + (handoff ? 1 : 0))) * 4 + wave) * (TM * TN * 16 * 64) + 4 * lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {   // one address register pair per 32x32 tile, the four register groups by immediate
          st_writethrough16<0>(pb, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
          st_writethrough16<1024>(pb, make_float4(acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]));
          st_writethrough16<2048>(pb, make_float4(acc[i][j][8], acc[i][j][9], acc[i][j][10], acc[i][j][11]));
          st_writethrough16<3072>(pb, make_float4(acc[i][j][12], acc[i][j][13], acc[i][j][14], acc[i][j][15]));
          pb += 1024;
          ECO_OPAQUE64(pb);
        }
      wait_own_stores();
      __syncthreads();
      if (handoff) {
        if (tid == 0) flag_publish(a.sk_flags + w, 1);
      } else {
        fin_m0 = m0; fin_n0 = n0; fin_first = tile_u0;      // finish it after the loop
      }
    }
    __syncthreads();   // As / Bs are reused by the next segment
    u_hi = seg_lo;
  }
  if (fin_m0 < 0) return;
  // Finish the shared tile: own block + the hand-off blocks of the earlier workgroups that hold its other stages (nearest
  // first, down to the one with the tile's first stage: an order fixed by the geometry), then the epilogue.  The
  // accumulators are rebuilt from memory here -- with nothing of the reduction live any more -- instead of being carried
  // out of the main loop and merged with a second definition (that merge cost the kernel 330 spilled registers).
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  for (int w2 = w; w2 >= 0; --w2) {
    if (w2 < w) {
      if (tid == 0) flag_wait(a.sk_flags + w2, 1);
      __syncthreads();
    }
    const float* blk = a.ws + (((long)(2 * w2 + (w2 < w ? 1 : 0))) * 4 + wave) * (TM * TN * 16 * 64) + 4 * lane;
#pragma unroll
    for (int t0 = 0; t0 + 4 <= TM * TN; t0 += 4) {   // four 32x32 tiles = sixteen 16-byte loads per round trip
      float4 q[4][4];
      ld_partial_tiles4(blk + t0 * 1024, q);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x16& c = acc[(t0 + t) / TN][(t0 + t) % TN];
          c[4 * g] += q[t][g].x; c[4 * g + 1] += q[t][g].y; c[4 * g + 2] += q[t][g].z; c[4 * g + 3] += q[t][g].w;
        }
      sched_fence();
    }
#pragma unroll
    for (int t = (TM * TN) / 4 * 4; t < TM * TN; ++t) {   // 2- and 6-tile waves: the rest one tile at a time
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = ld_device_scope(blk + t * 1024 + (r >> 2) * 256 + (r & 3));
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t / TN][t % TN][r] += v[r];
      sched_fence();
    }
    if (w2 < w && U * w2 / a.sk_wgs <= fin_first) break;       // w2's range holds the tile's first stage
  }
  conv_epilogue<TM, TN>(a, acc, fin_m0 + wm * TM * 32, fin_n0 + wn * TN * 32, half, l31);
}

// ------------------------------------------------------------------------------------------
// "Span" kernel (ECO_CONV_MODE_SPAN): stride-1, same-size kd x 3 x 3 convolutions with cin % 16 == 0 --
// every res3/res4/res5 stride-1 conv, conv2_3x3 and all inception 3x3 convs (87 % of ECO-Lite's flops).
//
// For such a conv the im2col row of tap (z, y, x) of channel c is the channel's *own flattened plane
// sequence shifted by a constant*:  X_c[v + (z-kd/2)*H*W + (y-1)*W + (x-1)]  (v = flattened (img, d, h, w)
// position), masked where the shifted tap leaves the image.  So instead of gathering 16 x BN elements for
// each of the 9 (y, x) taps, a workgroup stages once per (16-channel tile, z) group the *span*
//   Bsp[ci][e] = X_{cc*16+ci}[n0 - (W+1) + (z-kd/2)*H*W + e],   e in [0, BN + 2*(W+1))
// in LDS and the MFMA B-fragments of the group's 9 stages are read from it at offset y*W + x, with one
// select per fragment for the zero padding.  Global gather instructions and LDS stores for the B operand
// drop ~6-8x; the weight operand, the stage / k-pair structure, split-K and the epilogue are those of
// conv_mfma_kernel (packed-weight order is CTAP's: k' = ((cc*kd + z)*9 + y*3 + x)*16 + ci).
//
// The span of the next group is fetched two channels per stage (one coalesced dword load per 256
// consecutive span elements) under the MFMAs of the current group and stored to the other span buffer at
// the end of each stage; weights are double-buffered per stage as before.  LDS is dynamic
// (2*16*BM + 2*16*span_len floats: 40 KB for res3, 52 KB for 28x28 inception convs, 59 KB for conv2_3x3).
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256, (TM * TN <= 4 ? 4 : TM * TN <= 6 ? 3 : 2)) void conv_span_kernel(const ConvKernelArgs a) {
  constexpr int KC = 16, KSTEPS = 8, T2 = 9;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int NCOL = 2;  // span columns per thread: span_len <= 512
  // Weight operand: through a double-buffered LDS tile shared by the four waves (one barrier per stage), or
  // -- for the 128x256 tile, where each fragment feeds four MFMAs and two waves per SIMD leave registers to
  // spare -- straight from global memory into a register ring (one barrier per group of 9 stages).  Measured:
  // direct is +1 % on the 128x256 tile and -1.5..-3 % on the others (2-4x duplicated L1 traffic).
  constexpr bool ALDS = !(TM == 2 && TN == 4);
  constexpr int A_F4 = KC * BM / 4;
  constexpr int A_ITERS = (A_F4 + 255) / 256;

  ECO_DYNAMIC_LDS(lds);
  const int halo = a.Wi + 1;
  const int span_len = BN + 2 * halo;
  float* Bsp = lds;                      // [2][KC/2][span_len][2]  (k-pair interleaved: row p = channels 2p, 2p+1)
  float* As = lds + 2 * KC * span_len + 4;  // [2][KC/2][BM][2] behind the span buffers and the dummy slot (ALDS)

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  int tile, slice, nslices;
  if ((int)blockIdx.x < a.n_main) {
    tile = xcd_remap((int)blockIdx.x, a.n_main);
    slice = 0;
    nslices = 1;
  } else {
    // (slice, tile) pairs that exist, slice-major: slices [0, ns_short) of every split tile, then slices
    // [ns_short, ns_long) of the long columns only -- numbered densely so that the XCD remap deals every
    // XCD the same number of live workgroups.
    const int c0 = a.n_main / a.nblk_m;                               // first split column
    const int cl0 = a.col_long0 > c0 ? a.col_long0 : c0;             // long columns of the split region
    const int n_long = (a.col_long1 > cl0 ? a.col_long1 - cl0 : 0) * a.nblk_m;
    const int n_all = a.n_split * a.ns_short;
    const int lid = xcd_remap((int)blockIdx.x - a.n_main, n_all + n_long * (a.ns_long - a.ns_short));
    if (lid < n_all) {
      slice = lid / a.n_split;
      tile = a.n_main + (lid - slice * a.n_split);
    } else {
      const int r = lid - n_all;
      slice = a.ns_short + r / n_long;
      tile = cl0 * a.nblk_m + r % n_long;
    }
    nslices = col_slices(a, tile / a.nblk_m);
  }
  const int mblk = tile % a.nblk_m, nblk = tile / a.nblk_m;
  const int m0 = mblk * BM, n0 = nblk * BN;
  // Reduction work list of this tile: "live" groups l = (cc, z) with z in [zlo, zhi], cc-major; a split-K
  // slice is an equal share of it.
  int zlo, zhi;
  live_depth_taps(a, n0, BN, zlo, zhi);
  const int nz = zhi - zlo + 1;
  const int nlive = (a.cin / KC) * nz;
  const bool sliced = (int)blockIdx.x >= a.n_main;
  const int l_begin = (int)((long)slice * nlive / nslices);
  const int l_end = (int)((long)(slice + 1) * nlive / nslices);
  auto live_group = [&](int l) -> int { const int cc = l / nz; return cc * a.kd + zlo + (l - cc * nz); };
  const int hw = a.Hi * a.Wi;
  const int S = a.s_in;  // == s_out

  // ---- zero-padding masks of this lane's TN fragment positions: bit (z*9 + y*3 + x) ----
  unsigned fmask[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    fmask[j] = 0u;
    const int n = n0 + (wn * TN + j) * 32 + l31;
    if (n < a.ntot) {
      int img_, sp;
      decode_pos(a, n, img_, sp);
      const int w = sp % a.Wi, t = sp / a.Wi;
      const int h = t % a.Hi, d = t / a.Hi;
      unsigned mw = 0u, mhw = 0u;
      for (int xx = 0; xx < 3; ++xx) mw |= (unsigned)((unsigned)(w - 1 + xx) < (unsigned)a.Wi) << xx;
      for (int y = 0; y < 3; ++y)
        if ((unsigned)(h - 1 + y) < (unsigned)a.Hi) mhw |= mw << (3 * y);
      for (int z = 0; z < a.kd; ++z)
        if ((unsigned)(d - a.pd + z) < (unsigned)a.Di) fmask[j] |= mhw << (9 * z);
    }
  }

  // ---- span columns of this thread: element e = tid + q*256 of every channel row ----
  int col_base[NCOL], col_sp[NCOL];  // offset of (img, channel 0, sp) / sp, for the centre depth tap
  bool col_in[NCOL];
#pragma unroll
  for (int q = 0; q < NCOL; ++q) {
    const int e = tid + q * 256;
    col_in[q] = e < span_len;
    const int v = n0 - halo + e;
    col_base[q] = 0;
    col_sp[q] = -(1 << 29);  // never inside [0, S) whatever depth shift is added
    if (col_in[q] && v >= 0 && v < a.ntot) {
      int img, sp;
      decode_pos(a, v, img, sp);
      col_base[q] = (int)((long)img * a.img_stride_in + sp);
      col_sp[q] = sp;
    }
  }

  // uniform description of the group being loaded
  auto group_xoff = [&](int g, int& shift) -> int {  // the whole input fits int32 offsets (checked by the host)
    const int cc = g / a.kd, z = g - cc * a.kd;
    shift = (z - a.pd) * hw;
    return cc * KC * S + shift;
  };
  // The zero fill of out-of-image columns is applied when the value is *stored* to LDS, so that the wait for
  // the global load lands there and not right behind the load.
  auto span_ok = [&](int shift, int q) -> bool { return col_in[q] && (unsigned)(col_sp[q] + shift) < (unsigned)S; };
  auto span_load = [&](int xoff, bool ok, int ci, int q) -> float {
    return ld(a.x + (ok ? xoff + ci * S + col_base[q] : 0));
  };
  // columns past the end of the span go to one dummy float behind the buffers: no divergent branch in the loop
  const int dummy = 2 * KC * span_len;
  auto span_store = [&](int sbuf, int p, int q, bool ok, float v0, float v1) {  // channels 2p, 2p+1
    float2 v;
    v.x = ok ? v0 : 0.0f;
    v.y = ok ? v1 : 0.0f;
    *(float2*)&Bsp[col_in[q] ? 2 * ((sbuf * (KC / 2) + p) * span_len + tid + q * 256) : dummy] = v;
  };

  // ---- weights.  Direct form: lane (m = l31, half h) of a wave needs, for the step pair t of a stage, exactly
  // the two floats wp[pair-row 2t+h][m][0..1] of the packed image, so every fragment is one 8-byte global
  // load per lane (256 contiguous bytes per half-wave, L1/L2 hits: all N-tiles of an M-block read the same
  // rows).  The four step pairs of a stage form a register ring: the load for pair t of the *next* stage is
  // issued right after the MFMAs of pair t, ~6 MFMA steps before it is needed, and the only LDS hazard left
  // is the span double buffer.  LDS form: the stage's 8 pair-rows x BM x 2 floats are copied by all threads
  // (16-byte loads, one stage ahead) and read back as fragments like the span.
  const float* const wp_lane = a.wp + 2 * ((long)half * a.mpad + m0 + wm * TM * 32 + l31);  // pair-row `half`
  const float* const wp_tile = a.wp + 2 * m0;
  long l_woff = 0;  // float offset of the stage being loaded in the packed image
  float2 af[KSTEPS / 2][TM];
  float4 areg[A_ITERS];
  auto load_a = [&](int t) {  // direct: fragment pair t
#pragma unroll
    for (int i = 0; i < TM; ++i) af[t][i] = ld((const float2*)(wp_lane + l_woff + (long)t * 4 * a.mpad + i * 64));
  };
  auto load_a_tile = [&](int i) {  // LDS form: 16-byte piece i of the stage image
    const int idx = tid + i * 256;
    if (A_F4 % 256 == 0 || idx < A_F4) {
      const int row = idx / (BM / 2), c4 = idx % (BM / 2);
      areg[i] = ld((const float4*)(wp_tile + l_woff + (long)row * 2 * a.mpad + c4 * 4));
    }
  };
  auto store_a_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int idx = tid + i * 256;
      if (A_F4 % 256 == 0 || idx < A_F4) *(float4*)&As[(long)buf * KC * BM + (long)idx * 4] = areg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float2 bf[2][TN];
  bool okj[TN];        // this stage's tap is inside the image at fragment position j
  int frag_off = 0;    // y*W + x of this stage's tap
  int abuf = 0;        // LDS form: weight buffer of the current stage
  int ia[TM], ib[TN];  // float index of this lane's fragment i / j in pair-row `half` of buffer 0 (tap offset 0)
#pragma unroll
  for (int i = 0; i < TM; ++i) { ia[i] = 2 * (half * BM + (wm * TM + i) * 32 + l31); ECO_OPAQUE(ia[i]); }
#pragma unroll
  for (int j = 0; j < TN; ++j) { ib[j] = 2 * (half * span_len + (wn * TN + j) * 32 + l31); ECO_OPAQUE(ib[j]); }
  auto read_frags = [&](int sbuf, int t, int slot) {
    const float* bp = Bsp + 2 * ((sbuf * (KC / 2) + 2 * t) * span_len + frag_off);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[slot][j] = *(const float2*)(bp + ib[j]);
    if (ALDS) {
      const float* ap = As + 2 * (abuf * (KC / 2) + 2 * t) * BM;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[slot][i] = *(const float2*)(ap + ia[i]);
    }
  };
  auto mfma_step = [&](int kk) {  // the zero-padding select sits here, a scheduling region after the LDS read
    const int t = kk >> 1, slot = t & 1, aslot = ALDS ? slot : t;
    float b[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = okj[j] ? ((kk & 1) ? bf[slot][j].y : bf[slot][j].x) : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = mfma_32x32x2((kk & 1) ? af[aslot][i].y : af[aslot][i].x, b[j], acc[i][j]);
  };

  // ---- prologue: whole span of the first live group, weights of its first stage ----
  int l = l_begin;
  if (l < l_end) {
    const int g = live_group(l);
    int shift;
    const int xoff = group_xoff(g, shift);
    bool ok[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) ok[q] = span_ok(shift, q);
#pragma unroll 2
    for (int p = 0; p < KC / 2; ++p)
#pragma unroll
      for (int q = 0; q < NCOL; ++q)
        span_store(0, p, q, ok[q], span_load(xoff, ok[q], 2 * p, q), span_load(xoff, ok[q], 2 * p + 1, q));
    l_woff = (long)g * T2 * KC * a.mpad;
    if (ALDS) {
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) load_a_tile(i);
      store_a_tile(0);
    } else {
#pragma unroll
      for (int t = 0; t < KSTEPS / 2; ++t) load_a(t);
    }
  }
  __syncthreads();

  int sbuf = 0;
#pragma unroll 1
  for (; l < l_end; ++l) {
    const int g = live_group(l);
    const int z = g % a.kd;
    const bool next_group = l + 1 < l_end;
    const int gn = next_group ? live_group(l + 1) : g;
    int nshift = 0;
    const int nxoff = next_group ? group_xoff(gn, nshift) : 0;
    bool nok[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) nok[q] = next_group && span_ok(nshift, q);
#pragma unroll 1
    for (int t2 = 0; t2 < T2; ++t2) {
      // The stage body is branch-free and barrier-free: the last stage of the slice re-loads its own weights,
      // stage 8 of a group re-stages channels 14/15, and without a next group the span loads hit address 0
      // and land in the idle span buffer.  Conditional loads would split the body into basic blocks and make
      // the compiler drain vmcnt at every join.
      const int tap = z * T2 + t2;
      const int y = t2 / 3, xx = t2 - 3 * y;
      frag_off = y * a.Wi + xx;
#pragma unroll
      for (int j = 0; j < TN; ++j) okj[j] = (fmask[j] >> tap) & 1u;
      // weights of the next stage: the next tap of this group, or tap 0 of the next live group
      const int nstage = t2 + 1 < T2 ? g * T2 + t2 + 1 : (next_group ? gn * T2 : g * T2 + t2);
      l_woff = (long)nstage * KC * a.mpad;
      const int p0 = t2 < KC / 2 ? t2 : KC / 2 - 1;  // channel pair (2*p0, 2*p0+1) of the next group's span
      float sreg[2][NCOL];
      read_frags(sbuf, 0, 0);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        if ((kk & 1) == 0 && kk + 2 < KSTEPS) read_frags(sbuf, kk / 2 + 1, (kk / 2 + 1) & 1);
        if (kk < 2) {
#pragma unroll
          for (int q = 0; q < NCOL; ++q) sreg[kk][q] = span_load(nxoff, nok[q], 2 * p0 + kk, q);
        }
        if (ALDS && kk < A_ITERS) load_a_tile(kk);
        mfma_step(kk);
        if (!ALDS && (kk & 1)) load_a(kk >> 1);  // ring slot t is free again: fetch pair t of the next stage
        sched_fence();
      }
#pragma unroll
      for (int q = 0; q < NCOL; ++q) span_store(sbuf ^ 1, p0, q, nok[q], sreg[0][q], sreg[1][q]);
      if (ALDS) {
        store_a_tile(abuf ^ 1);
        abuf ^= 1;
        if (t2 + 1 < T2) __syncthreads();  // weights of the next stage are in place
      }
    }
    __syncthreads();  // the next group's span is complete; this group's buffer may be overwritten
    sbuf ^= 1;
  }
  if (sliced)
    conv_store_partial<TM, TN>(a, acc, slice, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);
  else
    conv_epilogue<TM, TN>(a, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 stride-1 convolutions (ECO_CONV_MODE_POINT: the *_1x1, *_reduce and *_pool_proj layers of the inception
// blocks: K = 192-576, cout 32-192 over 401 408 positions) as a plain GEMM staged by LDS-DMA.
//
// Y[m, n] = sum_c Wp[c, m] * X[img(n), c, sp(n)]: row c of the position operand is the channel's own plane,
// contiguous inside an image, so a stage of 16 channels is 16 rows of BN consecutive positions that go
// global -> LDS as 16-byte pieces (four positions per lane; a plane size that is a multiple of 4 keeps a lane's
// four positions inside one image) -- no address decode per element, no tap mask, no VGPR staging, no ds_write.
// These layers sit near the HBM ridge (89-140 flop/B): what they need is bytes in flight, which three stage
// buffers of DMA give (two stages of 16-24 KB per workgroup outstanding while the third is multiplied).  The
// gather kernel ran them at 0.45-0.50 of their floor.  Same packed weights (CTAP order = channel order for a
// single tap), same epilogue and views as conv_mfma_kernel.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_point_kernel(const ConvKernelArgs a) {
  ECO_CLOCK("conv_point");
  constexpr int KC = 16;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int BMP = (BM + 63) / 64 * 64;
  static_assert(WM * WN == 4 && BN == 256, "");
  constexpr int A_PER_WAVE = BMP / 64;   // 16 rows x BMP floats = BMP/16 pieces of 1 KB
  constexpr int B_PER_WAVE = 4;          // 16 rows x 256 positions: one piece per row
  constexpr int P = A_PER_WAVE + B_PER_WAVE;

  ECO_DYNAMIC_LDS(lds);
  float* const As = lds;                        // [3][KC][BMP]
  float* const Bs = lds + 3 * KC * BMP;         // [3][KC][BN]

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int tile = xcd_remap((int)blockIdx.x, a.nblk_m * a.nblk_n);
  const int mblk = tile % a.nblk_m, nblk = tile / a.nblk_m;
  const int m0 = mblk * BM, n0 = nblk * BN;
  const int nstages = a.cin / KC;
  __shared__ __attribute__((aligned(16))) float Ep[3 * BMP];   // bias / BN scale / BN shift of this workgroup's rows
  conv_stage_params<BMP>(a, m0, Ep);

  // this lane's four positions n0 + 4*lane .. +3 (one image: s_out % 4 == 0)
  long lane_base;
  {
    const int n = n0 + 4 * lane;
    const int nn = n < a.ntot ? n : 0;          // past the end: any valid address (those columns are never stored)
    const int img = nn / a.s_out, sp = nn - img * a.s_out;
    lane_base = (long)img * a.img_stride_in + sp;
  }
  int l_stage = 0;
  auto issue_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q) {
      const int i = (wave * A_PER_WAVE + q) * 64 + lane;           // float4 index in the [KC][BMP] tile
      const int row = i / (BMP / 4), c4 = i - row * (BMP / 4);
      glds16((const uint4*)(a.wp + (long)(l_stage * KC + row) * a.mpad + m0 + c4 * 4),
             (uint4*)(As + buf * KC * BMP) + (wave * A_PER_WAVE + q) * 64);
    }
#pragma unroll
    for (int q = 0; q < B_PER_WAVE; ++q) {
      const int row = wave * B_PER_WAVE + q;
      glds16((const uint4*)(a.x + lane_base + (long)(l_stage * KC + row) * a.s_in), (uint4*)(Bs + (buf * KC + row) * BN));
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

  auto compute = [&](int buf) {
    const float* Ab = As + buf * KC * BMP + half * BMP + wm * TM * 32 + l31;
    const float* Bb = Bs + buf * KC * BN + half * BN + wn * TN * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = Ab[2 * kk * BMP + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bb[2 * kk * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(af[i], bf[j], acc[i][j]);
    }
  };

  issue_stage(0);
  if (nstages > 1) issue_stage(1);
  int buf = 0;
  for (int s = 0; s < nstages; ++s) {
    if (s + 1 < nstages) wait_dma_all_but<P>(); else wait_dma_all_but<0>();
    wg_barrier_nodrain();
    if (s + 2 < nstages) issue_stage(buf == 0 ? 2 : buf - 1);
    sched_fence();
    compute(buf);
    buf = buf == 2 ? 0 : buf + 1;
  }
  conv_epilogue<TM, TN>(a, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31, Ep, BMP, m0);
}

static int validate_geom(const eco_conv_geom* g) {
  ECO_REQUIRE(g != nullptr, "conv: null geometry");
  ECO_REQUIRE(g->n > 0 && g->cin > 0 && g->cout > 0, "conv: n/cin/cout must be positive (n=%d cin=%d cout=%d)",
              g->n, g->cin, g->cout);
  long taps = 1;
  for (int i = 0; i < 3; ++i) {
    ECO_REQUIRE(g->in[i] > 0 && g->kernel[i] > 0 && g->stride[i] > 0 && g->pad[i] >= 0,
                "conv: Filter/stride dimensions must be nonzero (axis %d)", i);
    const int o = (g->in[i] + 2 * g->pad[i] - g->kernel[i]) / g->stride[i] + 1;
    ECO_REQUIRE(g->in[i] + 2 * g->pad[i] >= g->kernel[i] && o == g->out[i],
                "conv: output dim %d is %d, expected (in+2*pad-kernel)/stride+1 = %d", i, g->out[i], o);
    taps *= g->kernel[i];
  }
  ECO_REQUIRE(taps < kNeverTap, "conv: %ld kernel taps exceed the 63-tap validity mask", taps);
  const long s_in = (long)g->in[0] * g->in[1] * g->in[2];
  const long s_out = (long)g->out[0] * g->out[1] * g->out[2];
  ECO_REQUIRE((long)g->cin * s_in <= kKoffMask, "conv: per-image input (%ld elems) exceeds the 2^26 gather-offset range",
              (long)g->cin * s_in);
  ECO_REQUIRE((long)g->n * s_out < 2147483647l, "conv: too many output positions for int32 indexing");
  return ECO_OK;
}

}  // namespace eco

using namespace eco;

// Live depth taps of tile column `col` (bn positions wide) of a depth-major span launch; kd otherwise.
// Depth-major position order (and dead depth-tap skipping): 3-deep kernels of the span and constant-tap kernels.
static bool host_dmajor(const eco_conv_geom* g, int mode) {
  return (mode == ECO_CONV_MODE_SPAN || mode == ECO_CONV_MODE_CTAP) && g->kernel[0] == 3;
}

static int host_col_nz(const eco_conv_geom* g, int mode, int bn, long col) {
  const int kd = g->kernel[0];
  if (!host_dmajor(g, mode)) return kd;
  const long ntot = (long)g->n * g->out[0] * g->out[1] * g->out[2];
  const long per_d = (long)g->n * g->out[1] * g->out[2];
  const long n0 = col * bn, nlast = (n0 + bn < ntot ? n0 + bn : ntot) - 1;
  const long dlo = n0 / per_d, dhi = nlast / per_d;
  const long sd = g->stride[0];
  const long zlo = g->pad[0] - dhi * sd > 0 ? g->pad[0] - dhi * sd : 0;
  const long zhi = g->in[0] - 1 + g->pad[0] - dlo * sd < kd - 1 ? g->in[0] - 1 + g->pad[0] - dlo * sd : kd - 1;
  return (int)(zhi - zlo + 1);
}

// Slices per tile column for split factor sp: columns [c_long0, c_long1) (all live depth taps) get ns_long,
// the others (tiles inside the first / last depth plane) ns_short = ceil(sp * nz / kd).  Falls back to one
// class when the columns do not form that pattern.
static void host_split_layout(const eco_conv_geom* g, int mode, int bn, int sp, int* c_long0, int* c_long1,
                              int* ns_short, int* ns_long) {
  const int kd = g->kernel[0];
  const long ntot = (long)g->n * g->out[0] * g->out[1] * g->out[2];
  const int ncols = (int)ceil_div(ntot, bn);
  *c_long0 = 0; *c_long1 = ncols; *ns_short = *ns_long = sp;
  int nzmin = kd, nzmax = 0;
  for (int c = 0; c < ncols; ++c) {
    const int nz = host_col_nz(g, mode, bn, c);
    if (nz < nzmin) nzmin = nz;
    if (nz > nzmax) nzmax = nz;
  }
  auto slices = [&](int nz) { const int ns = (sp * nz + kd - 1) / kd; return ns < 1 ? 1 : ns; };
  if (nzmin == nzmax) { *ns_short = *ns_long = slices(nzmax); return; }
  int a = 0, b = ncols;
  while (a < ncols && host_col_nz(g, mode, bn, a) != nzmax) ++a;
  while (b > a && host_col_nz(g, mode, bn, b - 1) != nzmax) --b;
  for (int c = 0; c < ncols; ++c) {
    const int nz = host_col_nz(g, mode, bn, c);
    if ((c >= a && c < b) ? nz != nzmax : nz != nzmin) { *ns_short = *ns_long = slices(nzmax); return; }  // irregular
  }
  *c_long0 = a; *c_long1 = b; *ns_short = slices(nzmin); *ns_long = slices(nzmax);
}

extern "C" int eco_conv_plan_create(const eco_conv_geom* g, eco_conv_plan* plan) {
  return eco_conv_plan_create_batched(g, 0, 1, plan);
}

extern "C" int eco_conv_plan_create_ex(const eco_conv_geom* g, int32_t num_cu, eco_conv_plan* plan) {
  return eco_conv_plan_create_batched(g, num_cu, 1, plan);
}

extern "C" int eco_conv_plan_create_batched(const eco_conv_geom* g, int32_t num_cu, int32_t batch, eco_conv_plan* plan) {
  clear_error();
  ECO_REQUIRE(num_cu >= 0, "conv: num_cu must be positive (or 0 for the current device)");
  if (num_cu == 0) num_cu = current_device_num_cu();
  ECO_REQUIRE(batch >= 1, "conv: batch must be positive");
  if (int rc = validate_geom(g)) return rc;
  ECO_REQUIRE(plan != nullptr, "conv: null plan");
  int bm;
  if (g->cout <= 32) bm = 32;
  else if (g->cout <= 64) bm = 64;
  else if (g->cout <= 96) bm = 96;
  else {
    bm = 128;
    long best = ceil_div(g->cout, 128) * 128;
    const int cands[2] = {96, 64};
    for (int c : cands) {
      const long padded = ceil_div(g->cout, c) * c;
      if (padded < best) { best = padded; bm = c; }
    }
  }
  plan->bm = bm;
  plan->bn = (bm == 128) ? 128 : 256;
  if (bm == 128) {
    // With few output tiles (the res4/res5 stages) the wider 128x256 tile wins (~6 % measured): half the
    // weight loads, fragment reads and barriers per MFMA at 2 workgroups/CU; split-K keeps the CUs busy.
    // With thousands of tiles (res3) 128x128 at 4 workgroups/CU is as fast and quantises better.
    const long ntot_ = (long)g->n * g->out[0] * g->out[1] * g->out[2];
    if (ceil_div(g->cout, 128) * ceil_div(ntot_, 128) * batch <= 4L * num_cu) plan->bn = 256;
    // batched GEMM-like launches with several M-blocks (res4's transformed convs): the wide tile is 5 % faster
    if (batch > 1 && g->cout >= 256) plan->bn = 256;
  }
  plan->kc = 16;
  plan->mode = (g->cin % plan->kc == 0) ? ECO_CONV_MODE_CTAP : ECO_CONV_MODE_TABLE;
  {
    // stride-1, same-size (kd) x 3 x 3 convolutions stage input *spans* instead of per-tap gathers
    bool span = plan->mode == ECO_CONV_MODE_CTAP && g->kernel[1] == 3 && g->kernel[2] == 3 && g->pad[1] == 1 &&
                g->pad[2] == 1 && (g->kernel[0] == 1 || g->kernel[0] == 3) && g->pad[0] == g->kernel[0] / 2;
    for (int i = 0; i < 3; ++i) span = span && g->stride[i] == 1 && g->out[i] == g->in[i];
    span = span && plan->bn + 2 * (g->in[2] + 1) <= 512;  // two span columns per thread
    if (span) plan->mode = ECO_CONV_MODE_SPAN;
    // 1x1 stride-1 unpadded convolutions over planes whose size is a multiple of 4, with enough positions to
    // fill 256-wide tiles: the LDS-DMA GEMM kernel (single launches only; the gather kernel keeps the batched form)
    bool point = plan->mode == ECO_CONV_MODE_CTAP && batch == 1;
    for (int i = 0; i < 3; ++i) point = point && g->kernel[i] == 1 && g->stride[i] == 1 && g->pad[i] == 0;
    const long s_out_ = (long)g->out[0] * g->out[1] * g->out[2];
    // (one 256-wide tile per CU is enough: round 5 measured ECO-Full's 14 x 14 sibling groups -- 100 352 positions, until then
    // on the gather kernel behind a 4-tiles-per-CU rule -- 2.5 % faster here, inception_4e_3x3_reduce 0.395 -> 0.345 ms)
    point = point && s_out_ % 4 == 0 && (long)g->n * s_out_ >= 1L * num_cu * 256;
    if (point) { plan->mode = ECO_CONV_MODE_POINT; plan->bn = 256; }
  }
  plan->k = g->cin * g->kernel[0] * g->kernel[1] * g->kernel[2];
  plan->kpad = (int)(ceil_div(plan->k, plan->kc) * plan->kc);
  plan->mpad = (int)(ceil_div(g->cout, 128) * 128);
  if (plan->mpad < ceil_div(g->cout, bm) * bm) plan->mpad = (int)(ceil_div(g->cout, bm) * bm);
  plan->mpad = (int)(ceil_div(plan->mpad, 4) * 4);
  plan->wp_elems = (int64_t)plan->kpad * plan->mpad;
  plan->ktab_elems = plan->kpad;
  // Split-K.  (a) With fewer tiles than resident workgroup slots the CUs are unevenly loaded (e.g. 392 tiles
  // = 1.53 per CU -> the busiest CU does 2) and under-occupied: cut the whole reduction into S slices so
  // that tiles*S quantises well.  (b) With many tiles the last, partial round of workgroups leaves most
  // CUs idle while a few finish (3136 tiles over 1024 slots: 3 rounds + 64 stragglers = 4.5 % measured):
  // split only those trailing tiles, S ways, so that they form one more full round of short blocks.
  // Either way every slice keeps >= 8 stages, the partial sums cost S write+read passes over the split
  // region's outputs (priced at ~4 TB/s against ~100 TFLOP/s of MFMA time), and a second, deterministic
  // launch reduces them and applies the epilogue.
  plan->ksplit = 1;
  plan->split_tiles = 0;
  plan->ws_bytes = 0;
  plan->streamk_wgs = 0;
  if (plan->mode != ECO_CONV_MODE_POINT) {
    const long s_out = (long)g->out[0] * g->out[1] * g->out[2];
    const long ntot = (long)g->n * s_out;
    const long mblocks = ceil_div(g->cout, bm);
    const long tiles = mblocks * ceil_div(ntot, plan->bn);
    const int nstages = plan->kpad / plan->kc;
    const int ngroups = plan->mode == ECO_CONV_MODE_SPAN ? nstages / 9 : nstages;  // split-K granularity
    const int occ = (bm == 128 && plan->bn == 256) ? 2 : (bm == 96 ? 3 : 4);  // resident workgroups per CU
    const long slots = (long)num_cu * occ;
    // Depth-major span launches (3-D kernels) skip the depth taps that only see padding: a tile whose
    // positions lie in the first/last depth plane has nz = 2 (or 1) live taps of kd = 3 and gets
    // ceil(sp*nz/kd) slices, so all workgroups stay ~1/sp of a full reduction long.
    const long ncols = ceil_div(ntot, plan->bn);
    auto split_wgs = [&](long col0, int sp, double* work) -> long {  // workgroups / live share of columns >= col0
      int cl0, cl1, nss, nsl;
      host_split_layout(g, plan->mode, plan->bn, sp, &cl0, &cl1, &nss, &nsl);
      long wgs = 0, nz_sum = 0;
      for (long c = col0; c < ncols; ++c) {
        nz_sum += host_col_nz(g, plan->mode, plan->bn, c);
        wgs += (long)((c >= cl0 && c < cl1) ? nsl : nss) * mblocks;
      }
      if (work) *work = (double)nz_sum / ((double)g->kernel[0] * (ncols - col0 > 0 ? ncols - col0 : 1));
      return wgs;
    };
    int max_sp = 64;
    if (max_sp > nstages / 8) max_sp = nstages / 8;
    if (max_sp > ngroups) max_sp = ngroups;
    // A batched launch (gridDim.y = batch entries of this plan, e.g. the Winograd transform points) fills the
    // device with tiles*batch workgroups; every entry gets the same split.
    if (tiles * batch < slots) {
      double work = 1.0;
      split_wgs(0, 1, &work);
      const double t_flops = 2.0 * ntot * g->cout * plan->k * work * batch / 100e12;
      double best = 1e30;
      for (int sp = 1; sp <= max_sp; ++sp) {
        const long wgs = split_wgs(0, sp, nullptr) * batch;
        const double eff = ((double)wgs / num_cu) / (double)ceil_div(wgs, num_cu);
        const double t = t_flops / eff + (sp > 1 ? 2.0 * sp * work * ntot * g->cout * 4.0 * batch / 4e12 + 5e-6 : 0.0);
        if (t < best * 0.97) { best = t; plan->ksplit = sp; }  // prefer fewer slices unless >3 % better
      }
      if (plan->ksplit > 1) plan->split_tiles = (int)tiles;
    } else {
      const long rem_all = (tiles * batch) % slots;           // stragglers of the last round, all entries
      if (rem_all > 0 && 2 * rem_all < slots) {
        long rem = ceil_div(ceil_div(rem_all, batch), mblocks) * mblocks;  // per entry, whole columns of M-blocks
        long sp = slots / (rem * batch);
        if (sp > max_sp) sp = max_sp;
        if (sp >= 2 && rem < tiles) { plan->ksplit = (int)sp; plan->split_tiles = (int)rem; }
      }
    }
    // Stream-K (conv_streamk_kernel), opt-in (ECO_STREAMK=1 in the environment when the plan is made), where the few-tile
    // rule above would cut EVERY tile of a single CTAP launch: two persistent workgroups per CU share the launch's
    // stages evenly across tile boundaries -- no quantisation, at most two partial blocks per workgroup instead of one
    // per (tile, slice), no reduce launch; measured no faster than the split it replaces (lost L2 locality, see the
    // kernel).  The column prefix table rides behind the gather table.
    plan->streamk_wgs = 0;
    static const bool want_streamk = getenv("ECO_STREAMK") != nullptr;   // opt-in: measured no faster (see the kernel)
    if (want_streamk && plan->ksplit > 1 && plan->split_tiles == tiles && plan->mode == ECO_CONV_MODE_CTAP && batch == 1 &&
        (long)nstages * tiles >= 16L * 2 * num_cu) {
      plan->streamk_wgs = 2 * num_cu;
      plan->ksplit = 1;
      plan->split_tiles = 0;
      plan->ws_bytes = ((int64_t)2 * plan->streamk_wgs * bm * plan->bn * 4 + 255) / 256 * 256 + (int64_t)plan->streamk_wgs * 4;
      plan->ktab_elems = plan->kpad + ncols + 1;
    }
    if (plan->ksplit > 1) {
      const long split_cols = plan->split_tiles / mblocks;   // N-blocks in the split region
      long split_pos = split_cols * plan->bn;
      if (split_pos > ntot) split_pos = ntot;
      const long n_split0 = (ceil_div(ntot, plan->bn) - split_cols) * plan->bn;
      plan->ws_bytes = (int64_t)plan->ksplit * g->cout * (ntot - n_split0) * 4;
      (void)split_pos;
    }
  }
  return ECO_OK;
}

extern "C" int eco_conv_pack_weights(const eco_conv_geom* g, const eco_conv_plan* plan, const float* w,
                                     float* wp, int32_t* ktab) {
  clear_error();
  if (int rc = validate_geom(g)) return rc;
  ECO_REQUIRE(plan && w && wp && ktab, "conv pack: null argument");
  const int taps = g->kernel[0] * g->kernel[1] * g->kernel[2];
  const int K = g->cin * taps;
  ECO_REQUIRE(plan->k == K && plan->kpad >= K && plan->mpad >= g->cout, "conv pack: plan does not match geometry");
  ECO_REQUIRE(plan->mode == ECO_CONV_MODE_TABLE ||
                  ((plan->mode == ECO_CONV_MODE_CTAP || plan->mode == ECO_CONV_MODE_SPAN || plan->mode == ECO_CONV_MODE_POINT) &&
                   g->cin % plan->kc == 0),
              "conv pack: bad plan mode");
  const long s_in = (long)g->in[0] * g->in[1] * g->in[2];
  memset(wp, 0, sizeof(float) * (size_t)plan->wp_elems);
  for (int k = 0; k < plan->kpad; ++k) {
    if (k >= K) {
      ktab[k] = (int32_t)((unsigned)kNeverTap << kKoffBits);
      continue;
    }
    int c, tap;
    if (plan->mode != ECO_CONV_MODE_TABLE) {  // CTAP / SPAN: k' = (cc*taps + tap)*kc + ci, channel c = cc*kc + ci
      const int ci = k % plan->kc, rest = k / plan->kc;
      tap = rest % taps;
      c = (rest / taps) * plan->kc + ci;
    } else {                                  // k = c*taps + tap (the reference's own weight order)
      c = k / taps;
      tap = k % taps;
    }
    const int kx = tap % g->kernel[2], ky = (tap / g->kernel[2]) % g->kernel[1], kz = tap / (g->kernel[2] * g->kernel[1]);
    const long off = (long)c * s_in + ((long)kz * g->in[1] + ky) * g->in[2] + kx;
    ktab[k] = (int32_t)(((unsigned)tap << kKoffBits) | (unsigned)off);
    // gather kernels: K-major rows wp[k][m]; span kernel: k-pair interleaved wp[k/2][m][k%2]
    const bool paired = plan->mode == ECO_CONV_MODE_SPAN;
    float* row = paired ? wp + (long)(k >> 1) * 2 * plan->mpad + (k & 1) : wp + (long)k * plan->mpad;
    const int mstep = paired ? 2 : 1;
    const long wk = (long)c * taps + tap;  // column of w[cout][cin*taps]
    for (int m = 0; m < g->cout; ++m) row[(long)m * mstep] = w[(long)m * K + wk];
  }
  if (plan->streamk_wgs > 0) {
    // stream-K: stages of one M-block's tiles in the columns ahead of each column (dead depth taps skipped as the kernel
    // skips them), ncols + 1 entries behind the gather table
    const long ntot = (long)g->n * g->out[0] * g->out[1] * g->out[2];
    const long ncols = ceil_div(ntot, plan->bn);
    ECO_REQUIRE(plan->ktab_elems >= plan->kpad + ncols + 1, "conv pack: plan has no room for the stream-K table");
    int32_t cum = 0;
    for (long c = 0; c <= ncols; ++c) {
      ktab[plan->kpad + c] = cum;
      if (c < ncols) cum += (g->cin / plan->kc) * host_col_nz(g, plan->mode, plan->bn, c) * g->kernel[1] * g->kernel[2];
    }
  }
  return ECO_OK;
}

template <int TM, int TN, int WM, int WN>
static int launch_conv_point(const ConvKernelArgs& a, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BMP = (BM + 63) / 64 * 64;
  const size_t lds_bytes = sizeof(float) * 3 * 16 * (size_t)(BMP + BN);
  if (lds_bytes > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((conv_point_kernel<TM, TN, WM, WN>), "conv");
  hipLaunchKernelGGL((conv_point_kernel<TM, TN, WM, WN>), dim3(a.nblk_m * a.nblk_n), dim3(256), lds_bytes, stream, a);
  return check_launch("eco_conv_forward");
}

template <int TM, int TN, int WM, int WN, int KC>
static int launch_conv(const ConvKernelArgs& a, int mode, hipStream_t stream) {
  const int c0 = a.n_main / a.nblk_m, cl0 = a.col_long0 > c0 ? a.col_long0 : c0;
  const int n_long = (a.col_long1 > cl0 ? a.col_long1 - cl0 : 0) * a.nblk_m;
  const int grid = a.n_main + a.n_split * a.ns_short + n_long * (a.ns_long - a.ns_short);
  if (mode == ECO_CONV_MODE_SPAN) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const bool alds = !(TM == 2 && TN == 4);  // see conv_span_kernel
    const size_t lds_bytes = sizeof(float) * (2 * KC * (size_t)(BN + 2 * (a.Wi + 1)) + 4 + (alds ? 2 * KC * BM : 0));
    hipLaunchKernelGGL((conv_span_kernel<TM, TN, WM, WN>), dim3(grid), dim3(256), lds_bytes, stream, a);
    return check_launch("eco_conv_forward");
  }
  if (mode == ECO_CONV_MODE_CTAP)
    hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, KC, ECO_CONV_MODE_CTAP>), dim3(grid, a.batch), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, KC, ECO_CONV_MODE_TABLE>), dim3(grid, a.batch), dim3(256), 0, stream, a);
  return check_launch("eco_conv_forward");
}

static int check_view(const eco_view& v, const char* what) {
  if (!v.ptr) return ECO_OK;
  ECO_REQUIRE(v.t >= 1 && v.stride_c >= 1, "conv: %s view needs t >= 1 and stride_c >= 1", what);
  return ECO_OK;
}

static int conv_forward_impl(const eco_conv_geom* g, const eco_conv_plan* plan, const float* x, const float* wp,
                             const int32_t* ktab, const eco_conv_epilogue* ep, void* workspace, int batch,
                             long stride_x, long stride_wp, long stride_out, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  ECO_REQUIRE(plan && x && wp && ktab && ep, "conv: null argument");
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "conv: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "conv: bn_scale and bn_shift must be given together");
  if (int rc = check_view(ep->residual, "residual")) return rc;
  if (int rc = check_view(ep->raw, "raw")) return rc;
  if (int rc = check_view(ep->act, "act")) return rc;
  if (int rc = check_view(ep->act2, "act2")) return rc;
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "conv: act2 (second destination of the activated output) needs act");
  ECO_REQUIRE(plan->k == g->cin * g->kernel[0] * g->kernel[1] * g->kernel[2] && plan->kc == 16 &&
                  plan->kpad % plan->kc == 0 && plan->mpad % 4 == 0,
              "conv: plan does not match geometry");
  ConvKernelArgs a;
  a.x = x; a.wp = wp; a.ktab = ktab;
  a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.nseg = ep->nseg;
  for (int q = 0; q < ECO_MAX_SEG; ++q) { a.seg_begin[q] = 0; a.seg_relu[q] = 0; a.seg_act[q] = eco_view{nullptr, 0, 0, 0, 1}; }
  if (ep->nseg) {
    ECO_REQUIRE(ep->nseg >= 1 && ep->nseg <= ECO_MAX_SEG, "conv: 1 to %d extra output segments (got %d)", ECO_MAX_SEG, ep->nseg);
    ECO_REQUIRE(ep->act.ptr && ep->act.t == 1 && !ep->raw.ptr && !ep->residual.ptr && !ep->act2.ptr && batch == 1,
                "conv: a segmented launch takes plain act destinations only (no raw / residual / act2 / batch)");
    int prev = 0;
    for (int s = 0; s < ep->nseg; ++s) {
      const eco_view& v = ep->seg_act[s];
      ECO_REQUIRE(v.ptr && v.t == 1 && v.stride_c >= 1, "conv: segment %d needs a plain destination view", s + 1);
      ECO_REQUIRE(ep->seg_begin[s] > prev && ep->seg_begin[s] < g->cout && ep->seg_begin[s] % 32 == 0,
                  "conv: segment boundary %d must be a multiple of 32 inside (%d, %d)", ep->seg_begin[s], prev, g->cout);
      prev = a.seg_begin[s] = ep->seg_begin[s];
      a.seg_relu[s] = ep->seg_relu[s];
      a.seg_act[s] = v;
      a.seg_act[s].ptr = v.ptr - (long)ep->seg_begin[s] * v.stride_c;
    }
  }
  a.cin = g->cin; a.cout = g->cout; a.mpad = plan->mpad; a.kpad = plan->kpad;
  a.Di = g->in[0]; a.Hi = g->in[1]; a.Wi = g->in[2];
  a.Do = g->out[0]; a.Ho = g->out[1]; a.Wo = g->out[2];
  a.kd = g->kernel[0]; a.kh = g->kernel[1]; a.kw = g->kernel[2];
  a.sd = g->stride[0]; a.sh = g->stride[1]; a.sw = g->stride[2];
  a.pd = g->pad[0]; a.ph = g->pad[1]; a.pw = g->pad[2];
  a.s_in = a.Di * a.Hi * a.Wi; a.s_out = a.Do * a.Ho * a.Wo;
  a.img_stride_in = (long)a.cin * a.s_in;
  a.ntot = g->n * a.s_out;
  a.n_img = g->n;
  a.bn_tile = plan->bn;
  a.dmajor = host_dmajor(g, plan->mode) ? 1 : 0;
  a.nblk_m = (int)ceil_div(g->cout, plan->bm);
  a.nblk_n = (int)ceil_div(a.ntot, plan->bn);
  ECO_REQUIRE((long)a.nblk_m * plan->bm <= plan->mpad, "conv: plan mpad too small for bm");
  ECO_REQUIRE(plan->mode == ECO_CONV_MODE_TABLE ||
                  ((plan->mode == ECO_CONV_MODE_CTAP || plan->mode == ECO_CONV_MODE_SPAN || plan->mode == ECO_CONV_MODE_POINT) &&
                   g->cin % plan->kc == 0),
              "conv: bad plan mode");
  if (plan->mode == ECO_CONV_MODE_POINT) {
    bool ok = plan->bn == 256 && plan->ksplit == 1 && batch == 1 && a.s_out % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
              ((uintptr_t)wp & 15) == 0 && plan->mpad % 4 == 0;
    for (int i = 0; i < 3; ++i) ok = ok && g->kernel[i] == 1 && g->stride[i] == 1 && g->pad[i] == 0;
    ECO_REQUIRE(ok, "conv: the point kernel needs a 1x1 stride-1 unpadded geometry with a plane size that is a multiple of 4 "
                    "and 16-byte aligned operands");
    ECO_REQUIRE((long)(a.nblk_m - 1) * plan->bm + (plan->bm + 63) / 64 * 64 <= plan->mpad, "conv: plan mpad too small for the point kernel");
  }
  if (plan->mode == ECO_CONV_MODE_SPAN) {
    bool ok = g->kernel[1] == 3 && g->kernel[2] == 3 && g->pad[1] == 1 && g->pad[2] == 1 &&
              (g->kernel[0] == 1 || g->kernel[0] == 3) && g->pad[0] == g->kernel[0] / 2 &&
              plan->bn + 2 * (g->in[2] + 1) <= 512 && plan->kpad == plan->k;
    for (int i = 0; i < 3; ++i) ok = ok && g->stride[i] == 1 && g->out[i] == g->in[i];
    ECO_REQUIRE(ok, "conv: the span kernel needs a stride-1 same-size (kd)x3x3 geometry");
    ECO_REQUIRE(plan->ksplit <= plan->kpad / (plan->kc * 9), "conv: more split-K slices than channel-tile groups");
  }
  ECO_REQUIRE((long)g->n * a.img_stride_in < 2147483647l, "conv: input tensor too large for int32 gather offsets");
  const int mode = plan->mode;
  const int ntiles = a.nblk_m * a.nblk_n;
  ECO_REQUIRE(plan->ksplit >= 1 && (plan->ksplit == 1 || plan->kpad / plan->kc >= plan->ksplit),
              "conv: bad split-K factor %d", plan->ksplit);
  ECO_REQUIRE((plan->ksplit == 1) == (plan->split_tiles == 0) && plan->split_tiles >= 0 && plan->split_tiles <= ntiles &&
                  plan->split_tiles % a.nblk_m == 0,
              "conv: bad split-K region (%d of %d tiles, %d slices)", plan->split_tiles, ntiles, plan->ksplit);
  ECO_REQUIRE(plan->ksplit == 1 || workspace != nullptr, "conv: plan needs a %ld-byte workspace", (long)plan->ws_bytes);
  a.ksplit = plan->ksplit;
  a.n_split = plan->split_tiles;
  a.n_main = ntiles - a.n_split;
  a.n_split0 = (a.n_main / a.nblk_m) * plan->bn;
  ECO_REQUIRE(plan->ksplit == 1 || (int64_t)plan->ksplit * a.cout * (a.ntot - a.n_split0) * 4 <= plan->ws_bytes,
              "conv: plan workspace too small");
  a.ws = (float*)workspace;
  a.sk_wgs = 0; a.sk_units = 0; a.sk_cum = nullptr; a.sk_flags = nullptr;
  a.batch = batch;
  a.bstride_x = stride_x; a.bstride_w = stride_wp; a.bstride_out = stride_out;
  a.bstride_ws = plan->ws_bytes / 4;
  ECO_REQUIRE(batch >= 1 && batch <= 65535, "conv: bad batch %d", batch);
  ECO_REQUIRE(batch == 1 || mode != ECO_CONV_MODE_SPAN, "conv: the span kernel has no batched form");
  host_split_layout(g, mode, plan->bn, plan->ksplit, &a.col_long0, &a.col_long1, &a.ns_short, &a.ns_long);
  int rc = ECO_OK;
  hipStream_t s = (hipStream_t)stream;
  if (mode == ECO_CONV_MODE_POINT) {
    switch (plan->bm) {
      case 128: return launch_conv_point<4, 2, 1, 4>(a, s);
      case 96: return launch_conv_point<3, 2, 1, 4>(a, s);
      case 64: return launch_conv_point<2, 2, 1, 4>(a, s);
      case 32: return launch_conv_point<1, 2, 1, 4>(a, s);
      default: return fail(ECO_ERR_INVALID, "conv: unsupported point-kernel tile bm=%d", plan->bm);
    }
  }
  if (plan->streamk_wgs > 0) {
    ECO_REQUIRE(mode == ECO_CONV_MODE_CTAP && batch == 1 && plan->ksplit == 1 && workspace != nullptr && ktab != nullptr,
                "conv: bad stream-K plan (CTAP single launches with a workspace)");
    const int64_t blocks = ((int64_t)2 * plan->streamk_wgs * plan->bm * plan->bn * 4 + 255) / 256 * 256;   // two per workgroup
    ECO_REQUIRE(blocks + (int64_t)plan->streamk_wgs * 4 <= plan->ws_bytes && plan->ktab_elems >= plan->kpad + a.nblk_n + 1,
                "conv: stream-K plan workspace / table too small");
    // The finishing workgroup of a shared tile spins on flags of other workgroups: every one of the plan's workgroups
    // must be resident.  A plan sized for more compute units than the current device has (num_cu given by the caller)
    // would deadlock, so it is refused here rather than launched (round-3 advisor finding).
    ECO_REQUIRE(plan->streamk_wgs <= 2 * current_device_num_cu(),
                "conv: stream-K plan has %d persistent workgroups, the current device runs at most %d side by side",
                plan->streamk_wgs, 2 * current_device_num_cu());
    a.sk_wgs = plan->streamk_wgs;
    a.sk_cum = ktab + plan->kpad;
    a.sk_flags = (int*)((char*)workspace + blocks);
    // total stages: the prefix table's last entry x M-blocks, recomputed here (host side) from the geometry
    long cum = 0;
    for (long c = 0; c < a.nblk_n; ++c) cum += (long)(g->cin / plan->kc) * host_col_nz(g, mode, plan->bn, c) * g->kernel[1] * g->kernel[2];
    a.sk_units = cum * a.nblk_m;
    if (hipMemsetAsync(a.sk_flags, 0, (size_t)a.sk_wgs * 4, s) != hipSuccess)
      return fail(ECO_ERR_RUNTIME, "conv: cannot clear the stream-K flags");
    const dim3 grid((unsigned)a.sk_wgs), block(256);
    if (plan->bm == 128 && plan->bn == 256) hipLaunchKernelGGL((conv_streamk_kernel<2, 4, 2, 2, 16>), grid, block, 0, s, a);
    else if (plan->bm == 128 && plan->bn == 128) hipLaunchKernelGGL((conv_streamk_kernel<2, 2, 2, 2, 16>), grid, block, 0, s, a);
    else if (plan->bm == 96) hipLaunchKernelGGL((conv_streamk_kernel<3, 2, 1, 4, 16>), grid, block, 0, s, a);
    else if (plan->bm == 64) hipLaunchKernelGGL((conv_streamk_kernel<2, 2, 1, 4, 16>), grid, block, 0, s, a);
    else if (plan->bm == 32) hipLaunchKernelGGL((conv_streamk_kernel<1, 2, 1, 4, 16>), grid, block, 0, s, a);
    else return fail(ECO_ERR_INVALID, "conv: unsupported stream-K tile %dx%d", plan->bm, plan->bn);
    return check_launch("eco_conv_forward(stream-K)");
  }
  switch (plan->bm) {
    case 128:
      ECO_REQUIRE(plan->bn == 128 || plan->bn == 256, "conv: bad plan");
      rc = plan->bn == 128 ? launch_conv<2, 2, 2, 2, 16>(a, mode, s) : launch_conv<2, 4, 2, 2, 16>(a, mode, s);
      break;
    case 96: ECO_REQUIRE(plan->bn == 256, "conv: bad plan"); rc = launch_conv<3, 2, 1, 4, 16>(a, mode, s); break;
    case 64: ECO_REQUIRE(plan->bn == 256, "conv: bad plan"); rc = launch_conv<2, 2, 1, 4, 16>(a, mode, s); break;
    case 32: ECO_REQUIRE(plan->bn == 256, "conv: bad plan"); rc = launch_conv<1, 2, 1, 4, 16>(a, mode, s); break;
    default: return fail(ECO_ERR_INVALID, "conv: unsupported block tile bm=%d", plan->bm);
  }
  if (rc != ECO_OK || a.ksplit == 1) return rc;
  auto vec_ok = [](const eco_view& v) {
    return !v.ptr || (((uintptr_t)v.ptr & 15) == 0 && v.stride_b % 4 == 0 && v.stride_t % 4 == 0 && v.stride_c % 4 == 0);
  };
  const bool vec4 = a.s_out % 4 == 0 && (!a.dmajor || (a.Ho * a.Wo) % 4 == 0) && a.n_split0 % 4 == 0 && (a.ntot - a.n_split0) % 4 == 0 &&
                    ((uintptr_t)a.ws & 15) == 0 && vec_ok(a.residual) && vec_ok(a.raw) && vec_ok(a.act) && vec_ok(a.act2) &&
                    vec_ok(a.seg_act[0]) && vec_ok(a.seg_act[1]) && vec_ok(a.seg_act[2]);
  long rblocks = ceil_div((long)a.cout * (a.ntot - a.n_split0), 256L * (vec4 ? 4 : 1));
  if (rblocks > 262144) rblocks = 262144;
  if (vec4)
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<4>), dim3((unsigned)rblocks, a.batch), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<1>), dim3((unsigned)rblocks, a.batch), dim3(256), 0, s, a);
  return check_launch("eco_conv_forward(split-K reduce)");
}

extern "C" int eco_conv_forward(const eco_conv_geom* g, const eco_conv_plan* plan, const float* x, const float* wp,
                                const int32_t* ktab, const eco_conv_epilogue* ep, void* workspace, void* stream) {
  clear_error();
  return conv_forward_impl(g, plan, x, wp, ktab, ep, workspace, 1, 0, 0, 0, stream);
}

extern "C" int eco_conv_forward_batched(const eco_conv_geom* g, const eco_conv_plan* plan, const float* x,
                                        const float* wp, const int32_t* ktab, const eco_conv_epilogue* ep,
                                        void* workspace, int32_t batch, int64_t stride_x, int64_t stride_wp,
                                        int64_t stride_out, void* stream) {
  clear_error();
  ECO_REQUIRE(plan && plan->mode != ECO_CONV_MODE_SPAN, "conv (batched): span-mode plans have no batched form");
  return conv_forward_impl(g, plan, x, wp, ktab, ep, workspace, batch, stride_x, stride_wp, stride_out, stream);
}
