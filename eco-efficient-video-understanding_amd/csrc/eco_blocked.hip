// eco_blocked.hip -- the channel-blocked ("NC8") path on the bf16 matrix cores of gfx950.
//
// BASELINE.json configs[4] asks for ECO-Lite in bf16.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32
// MFMA the NCHW kernels of eco_conv.hip use, and every lane must hand it EIGHT consecutive reduction elements of
// one output channel (A) / one output position (B).  With the reference's N,C,[D,]H,W layout those eight would be
// eight channel planes apart; so this path keeps activations channel-blocked,
//
//     X[n][c/8][d][h][w][c%8]          ("NC8": 8 channels of one position = one 16-byte vector of bf16)
//
// which makes a lane's operand ONE 16-byte load (global or LDS), keeps every streaming kernel at 16 bytes per
// lane with consecutive lanes on consecutive positions, and still lets Concat (channel offsets are multiples of 8
// in every ECO graph) and r2Dto3D+Permute be pure stride arithmetic in the producer's epilogue.  The layout is
// internal: the `data` input stays fp32 N,3,H,W (eco_stem_pack_forward re-lays it), the logits leave as fp32
// [B, classes], and the host mirrors blobs back to N,C,... fp32 when a caller looks at them.
//
// Storage type `dt`: ECO_DT_BF16 -- bf16 activations and weights, fp32 accumulation, fp32 bias / BN parameters: the
// configs[4] arithmetic.  (Rounds 2-5 also carried ECO_DT_F32X3 -- fp32 storage, every operand split exactly into three
// bf16 terms, six products per multiply: fp32-class results, but measured 27.6 ms per configs[1] step against 17.3 ms on the
// fp32 MFMA, never a reported configuration -- and with it the register-staged kernel of round 2; round 6 removed both,
// and the per-tile span kernel of round 3, whose role the persistent kernel took over in round 4.)
//
// Replaces, for this storage layout, the same reference operators as eco_conv.hip / eco_ops.hip:
//   ConvolutionLayer::Forward (conv_layer.cpp:28-43, base_conv_layer.cpp:264-287, cudnn_conv_layer.cu:15-65) with
//   the BN / ReLU / Eltwise / Concat / Reshape+Permute layers fused behind it (bn_layer.cpp:93-207,
//   relu_layer.cpp:10-20, eltwise_layer.cpp:66-72, concat_layer.cpp:54-70, permute_layer.cpp:9-26),
//   PoolingLayer::Forward (pooling_layer.cpp:131-147,199-262; cudnn_pooling_layer.cu:13-22) and the
//   global_pool -> reshape -> dropout -> fc tail (inner_product_layer.cu:14-25).
#include <float.h>
#include <string.h>

#define ECO_GLDS_READFIRSTLANE 1   // see eco_device.h, glds16
#include <atomic>

#include "eco_blocked.h"

namespace eco {

#ifndef ECO_EMU
__device__ __attribute__((aligned(256))) const uint4 g_zero_page[16] = {};
#endif
// The all-zero DMA source (see device_zero_page below): the device symbol, or the page the emulator build passes in.
__device__ __forceinline__ const uint4* zero_page_ptr(const uint4* host_given) {
#ifdef ECO_EMU
  return host_given;
#else
  (void)host_given;
  return g_zero_page;
#endif
}


constexpr int kCbs = 4;   // channel blocks (of 8) per reduction stage: 32 reduction elements, two MFMA k-steps

struct ConvBArgs {
  const void* x;
  const uint4* wp;
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  eco_view residual, raw, act, act2;   // blocked views: strides in 8-channel blocks
  // sibling convs as one launch (eco_conv_epilogue::nseg): 32-row tiles at or above seg_begin[s] write through
  // seg_act[s] at channel block (c - seg_begin[s]) / 8, with seg_relu[s]
  int nseg, seg_begin[ECO_MAX_SEG], seg_relu[ECO_MAX_SEG];
  eco_view seg_act[ECO_MAX_SEG];
  int relu;
  int cblocks, cout, mpad, nstages, taps;
  int Di, Hi, Wi, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  long img_stride_in, cb_stride_in;   // in blocks
  int s_out, ntot, nblk_m, nblk_n, ksplit;
  float* ws;
  // split-K workspace ws[slice][channel][position - ws_n0], ws_pitch positions per row, ws_slices slices summed by the
  // reduce launch (a whole-tensor split: 0 / ntot / ksplit; the persistent kernel's K-split tail: its last position tiles)
  int ws_n0, ws_pitch, ws_slices;
  // ws_frag = 1 (bf16 launches whose destinations take the 16-byte-store epilogue): the partial sums leave in the MFMA
  // fragment layout instead -- ws[slice][tile - ws_tile0][wave][i][j][g][lane] float4 = registers 4g..4g+3 of the wave's
  // 32x32 tile (i, j): 8*TM one-KB stores per wave and slice where the [channel][position] form takes 32*TM 256-byte
  // ones, and the reduce launch (convb_splitk_reduce_frag_kernel) rebuilds a tile's accumulators with 16-byte loads and runs
  // the same epilogue as an unsplit tile.  ws_ntl = tiles the workspace covers.
  int ws_frag, ws_tile0, ws_ntl;
  int wide;            // 1: every destination view ends below 2 GB -> the 16-byte-store epilogue (convb_epilogue_wide)
  int lean;            // 1 (needs wide): one destination per 32-row tile, bias + BN + ReLU only -> convb_epilogue_lean
  // LDS-DMA kernel, descriptor form (dma_buf = 1: input + dma_bias and the packed weights end below 2 GB): a position
  // piece is fetched at descriptor offset lane register + SGPR, the lane register = (its position's first tap + dma_bias)
  // in bytes -- dma_bias blocks, the padding's reach, keep it non-negative where the first tap lies before the tensor
  int dma_buf;
  unsigned dma_x_bytes, dma_wp_bytes, dma_bias;
  FastDiv d_sout;      // position -> image by multiply-high
};

__device__ __forceinline__ void decode_out(const ConvBArgs& a, int n, int& img, int& sp) {
  img = n / a.s_out;
  sp = n - img * a.s_out;
}

// Fused epilogue (same algebra as conv_epilogue in eco_conv.hip) on blocked views.  Register r of lane l holds
// channel mw + i*32 + (r&3) + 8*(r>>2) + 4*(l>>5) at position nw + j*32 + (l&31): the four registers of a group
// g = r>>2 are four consecutive channels = half of one 8-channel block, so lanes l and l+32 together write whole
// 16-byte (bf16) / 32-byte (fp32) vectors, and consecutive lanes write consecutive vectors.
// The per-channel epilogue parameters of a workgroup's BM rows, fetched into LDS when the kernel starts: Ep[0] = bias,
// Ep[1] = BN scale, Ep[2] = BN shift (0 / 1 / 0 where absent or past cout).  The epilogue used to load them from global
// memory group by group -- twelve dependent round trips of ~400 cycles per tile, behind which the stores waited: on the
// short reductions (conv2_3x3, the inception 3x3s: 18-27 stages per tile) that chain was 15-20 % of the launch
// (profiles/r03_notes.md).  Visible to every wave after the main loop's first barrier.
template <int BMP>
__device__ __forceinline__ void convb_stage_params(const ConvBArgs& a, int m0, float* Ep, bool fold = false) {
  const int t = (int)threadIdx.x;
  if (t < BMP) {
    const int ch = m0 + t;
    const bool in = ch < a.cout;
    const float b = (in && a.bias) ? ld(a.bias + ch) : 0.0f;
    const float sc = (in && a.bn_scale) ? ld(a.bn_scale + ch) : 1.0f;
    const float sh = (in && a.bn_scale) ? ld(a.bn_shift + ch) : 0.0f;
    Ep[t] = b;
    Ep[BMP + t] = sc;
    Ep[2 * BMP + t] = fold ? fmaf(b, sc, sh) : sh;   // fold: (v + b) * sc + sh as v * sc + (b * sc + sh) (convb_epilogue_lean)
  }
}

// (Ep: convb_stage_params' array for the rows starting at m0; EPS = its row pitch)
// (e_img / e_sp / e_ok: image, spatial index and validity of this lane's TN fragment positions)
template <int TM, int TN>
__device__ __forceinline__ void convb_epilogue_at(const ConvBArgs& a, f32x16 (&acc)[TM][TN], int mw, int half,
                                                  const float* Ep, int EPS, int m0, const int (&e_img)[TN],
                                                  const int (&e_sp)[TN], const bool (&e_ok)[TN]) {
  long e_res[TN], e_raw[TN], e_act[TN], e_act2[TN];
  const bool has_res = a.residual.ptr != nullptr;
  const bool has_raw = a.raw.ptr != nullptr, has_act = a.act.ptr != nullptr, has_act2 = has_act && a.act2.ptr != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int img = e_img[j], sp = e_sp[j];
    e_res[j] = has_res ? view_base(a.residual, img, sp) : 0;
    e_raw[j] = has_raw ? view_base(a.raw, img, sp) : 0;
    e_act[j] = has_act ? view_base(a.act, img, sp) : 0;
    e_act2[j] = has_act2 ? view_base(a.act2, img, sp) : 0;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // the destination of this 32-row tile (wave-uniform): `act`, or a sibling's own tensor
    void* aptr = a.act.ptr;
    long astride_c = a.act.stride_c;
    int relu = a.relu, cb0 = 0;
    const int mt = mw + i * 32;
    if (a.nseg > 0 && mt >= a.seg_begin[0]) {
      // (constant indices only: a run-time index into the kernel argument struct makes the compiler copy it to scratch)
      // every candidate is loaded (constant indices, scalar loads) and the VALUES are selected: a conditional load, or
      // a run-time index, into the kernel argument struct makes the compiler copy the struct to scratch memory
      long sstride_b = a.seg_act[0].stride_b;
      aptr = a.seg_act[0].ptr; astride_c = a.seg_act[0].stride_c; relu = a.seg_relu[0]; cb0 = a.seg_begin[0] / 8;
#pragma unroll
      for (int q = 1; q < ECO_MAX_SEG; ++q) {
        void* const qp = a.seg_act[q].ptr;
        const long qc = a.seg_act[q].stride_c, qb = a.seg_act[q].stride_b;
        const int qr = a.seg_relu[q], qbeg = a.seg_begin[q];
        const bool take = q < a.nseg && mt >= qbeg;
        aptr = take ? qp : aptr; astride_c = take ? qc : astride_c; sstride_b = take ? qb : sstride_b;
        relu = take ? qr : relu; cb0 = take ? qbeg / 8 : cb0;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) e_act[j] = (long)e_img[j] * sstride_b + e_sp[j];
    }   // (tiles ascend: once past seg_begin[0] a wave never returns to `act`)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch0 = mw + i * 32 + 8 * g + 4 * half;
      if (ch0 >= a.cout) continue;   // cout is a multiple of 8: the quad is inside or outside as a whole
      const int cbk = (mw + i * 32) / 8 + g;
      float pb[4], ps[4], ph[4];
      {
        const float4 b4 = *(const float4*)(Ep + (ch0 - m0)), s4 = *(const float4*)(Ep + EPS + (ch0 - m0)),
                     h4 = *(const float4*)(Ep + 2 * EPS + (ch0 - m0));
        pb[0] = b4.x; pb[1] = b4.y; pb[2] = b4.z; pb[3] = b4.w;
        ps[0] = s4.x; ps[1] = s4.y; ps[2] = s4.z; ps[3] = s4.w;
        ph[0] = h4.x; ph[1] = h4.y; ph[2] = h4.z; ph[3] = h4.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!e_ok[j]) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + pb[q];
        if (has_res) {
          float rv[4];
          load_quad(a.residual.ptr, e_res[j] + (long)cbk * a.residual.stride_c, half, rv);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[q];
        }
        if (has_raw) store_quad(a.raw.ptr, e_raw[j] + (long)cbk * a.raw.stride_c, half, v);
        if (has_act) {
          float y[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            y[q] = v[q] * ps[q] + ph[q];
            if (relu) y[q] = fmaxf(y[q], 0.0f);
          }
          store_quad(aptr, e_act[j] + (long)(cbk - cb0) * astride_c, half, y);
          if (has_act2) store_quad(a.act2.ptr, e_act2[j] + (long)cbk * a.act2.stride_c, half, y);
        }
      }
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void convb_epilogue(const ConvBArgs& a, f32x16 (&acc)[TM][TN], int mw, int nw, int half,
                                               int l31, const float* Ep, int EPS, int m0) {
  int e_img[TN], e_sp[TN];
  bool e_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + l31;
    e_ok[j] = n < a.ntot;
    decode_out(a, e_ok[j] ? n : 0, e_img[j], e_sp[j]);
  }
  convb_epilogue_at<TM, TN>(a, acc, mw, half, Ep, EPS, m0, e_img, e_sp, e_ok);
}

// Epilogue of the persistent kernel: the same algebra as convb_epilogue_at, re-laid so that (1) every store is a whole
// 16-byte channel block -- the two half-blocks a 32x32 MFMA tile leaves in lanes l and l + 32 are brought together with
// v_permlane32_swap, after which lane l of the wave owns position nw + l: sixteen 1 KB store instructions per destination
// and tile instead of thirty-two 512-byte ones -- and (2) the stores go through buffer descriptors with the range check as
// their predicate (gst16_buf), so their NUMBER is a compile-time function of which destinations exist: the kernel's counted
// waits can step over them (profiles/r04_notes.md: waiting for a tile's stores to be acknowledged before the next tile's
// first tap cost conv2_3x3 a third of its time).  Residual blocks are fetched for the whole tile before the first store.
// Returns the number of store instructions issued per wave.  (bf16 storage only.)
__device__ __forceinline__ unsigned view_lane_offset(const eco_view& v, int img, int sp, bool ok) {
  return ok ? (unsigned)(view_base(v, img, sp) * 16) : kBufOob;
}
// (mw: first channel of the wave's rows; m0: first channel of the workgroup's rows = row 0 of Ep)
template <int TM>
__device__ __forceinline__ int convb_epilogue_wide(const ConvBArgs& a, f32x16 (&acc)[TM][2], int mw, int m0, int n_lane,
                                                   int half, const float* Ep, int EPS, const FastDiv& d_sout
                                                   ) {
  const bool has_res = a.residual.ptr != nullptr;
  const bool has_raw = a.raw.ptr != nullptr, has_act = a.act.ptr != nullptr, has_act2 = has_act && a.act2.ptr != nullptr;
  const bool ok = n_lane < a.ntot;
  const unsigned nn = ok ? (unsigned)n_lane : 0u;
  const int img = (int)fastdiv(nn, d_sout), sp = (int)(nn - (unsigned)img * (unsigned)a.s_out);
  constexpr unsigned kAll = 0x7fffffffu;   // (range check = the lane predicate only: valid offsets are below 2 GB by plan)
  const unsigned v_raw = has_raw ? view_lane_offset(a.raw, img, sp, ok) : kBufOob;
  const unsigned v_act2 = has_act2 ? view_lane_offset(a.act2, img, sp, ok) : kBufOob;
  const BufRsrc r_raw = make_buf_rsrc(a.raw.ptr, kAll), r_act2 = make_buf_rsrc(a.act2.ptr, kAll);
  // residual blocks of this lane's position: all of the tile's, ahead of the first store (a load behind a store waits for
  // the store's acknowledgement: the memory counter retires in order)
  uint4 res[TM][4];
  if (has_res) {
    const long rb = view_base(a.residual, img, sp);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cbk = (mw + i * 32) / 8 + g;
        res[i][g] = (ok && cbk * 8 < a.cout) ? ld((const uint4*)a.residual.ptr + rb + (long)cbk * a.residual.stride_c)
                                             : make_uint4(0u, 0u, 0u, 0u);
      }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // the destination of this 32-row tile (wave-uniform): `act`, or a sibling's own tensor (values selected, never a
    // run-time index into the kernel argument struct: that would copy it to scratch memory)
    const int mt = mw + i * 32;
    void* aptr = a.act.ptr;
    long astride_c = a.act.stride_c, astride_b = a.act.stride_b;
    int relu = a.relu, cb0 = 0;
    bool seg = false;
    if (a.nseg > 0 && mt >= a.seg_begin[0]) {
      seg = true;
      aptr = a.seg_act[0].ptr; astride_c = a.seg_act[0].stride_c; astride_b = a.seg_act[0].stride_b;
      relu = a.seg_relu[0]; cb0 = a.seg_begin[0] / 8;
#pragma unroll
      for (int q = 1; q < ECO_MAX_SEG; ++q) {
        void* const qp = a.seg_act[q].ptr;
        const long qc = a.seg_act[q].stride_c, qb = a.seg_act[q].stride_b;
        const int qr = a.seg_relu[q], qbeg = a.seg_begin[q];
        const bool take = q < a.nseg && mt >= qbeg;
        aptr = take ? qp : aptr; astride_c = take ? qc : astride_c; astride_b = take ? qb : astride_b;
        relu = take ? qr : relu; cb0 = take ? qbeg / 8 : cb0;
      }
    }
    const float floor_ = relu ? 0.0f : -__builtin_inff();   // ReLU / no ReLU as one v_max either way
    const BufRsrc r_act = make_buf_rsrc(aptr, kAll);
    const unsigned v_act = !has_act ? kBufOob : !ok ? kBufOob
                           : seg ? (unsigned)(((long)img * astride_b + sp) * 16) : (unsigned)(view_base(a.act, img, sp) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cbk = mt / 8 + g;
      const bool rows = cbk * 8 < a.cout;     // (cout is a multiple of 8: a block is inside or outside as a whole; uniform)
      const int ch0 = mt + 8 * g + 4 * half;
      const float4 b4 = *(const float4*)(Ep + (ch0 - m0)), s4 = *(const float4*)(Ep + EPS + (ch0 - m0)),
                   h4 = *(const float4*)(Ep + 2 * EPS + (ch0 - m0));
      const float pb[4] = {b4.x, b4.y, b4.z, b4.w}, ps[4] = {s4.x, s4.y, s4.z, s4.w}, ph[4] = {h4.x, h4.y, h4.z, h4.w};
      // the residual block holds this lane's POSITION (all 8 channels): hand the halves the partner lane needs across, so
      // that rq[j] = channels 4*half .. 4*half+3 of fragment position j
      unsigned rq[2][2] = {{0u, 0u}, {0u, 0u}};
      if (has_res) {
        unsigned r01x = res[i][g].x, r01y = res[i][g].y, r23x = res[i][g].z, r23y = res[i][g].w;
        permlane32_swap(r01x, r23x);
        permlane32_swap(r01y, r23y);
        rq[0][0] = r01x; rq[0][1] = r01y; rq[1][0] = r23x; rq[1][1] = r23y;
      }
      unsigned praw[2][2], pact[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + pb[q];
        if (has_res) {
          v[0] += bf16_bits_to_f32(rq[j][0] & 0xffffu); v[1] += bf16_bits_to_f32(rq[j][0] >> 16);
          v[2] += bf16_bits_to_f32(rq[j][1] & 0xffffu); v[3] += bf16_bits_to_f32(rq[j][1] >> 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[q] = fmaxf(v[q] * ps[q] + ph[q], floor_);
        }
        praw[j][0] = pack_bf16x2(v[0], v[1]); praw[j][1] = pack_bf16x2(v[2], v[3]);
        pact[j][0] = pack_bf16x2(y[0], y[1]); pact[j][1] = pack_bf16x2(y[2], y[3]);
      }
      // lanes 0..31: [own channels 0-3 | partner's 4-7] of position j = 0; lanes 32..63: [partner's 0-3 | own 4-7] of j = 1
      if (has_raw) {
        permlane32_swap(praw[0][0], praw[1][0]);
        permlane32_swap(praw[0][1], praw[1][1]);
        gst16_buf(r_raw, rows ? v_raw : kBufOob, (unsigned)((long)cbk * a.raw.stride_c * 16),
                  make_uint4(praw[0][0], praw[0][1], praw[1][0], praw[1][1]));
      }
      if (has_act) {
        permlane32_swap(pact[0][0], pact[1][0]);
        permlane32_swap(pact[0][1], pact[1][1]);
        const uint4 q = make_uint4(pact[0][0], pact[0][1], pact[1][0], pact[1][1]);
        gst16_buf(r_act, rows ? v_act : kBufOob, (unsigned)((long)(cbk - cb0) * astride_c * 16), q);
        if (has_act2) gst16_buf(r_act2, rows ? v_act2 : kBufOob, (unsigned)((long)cbk * a.act2.stride_c * 16), q);
      }
    }
  }
  return TM * 4 * ((has_raw ? 1 : 0) + (has_act ? 1 : 0) + (has_act2 ? 1 : 0));
}

// Split-K partial sums: ws[slice][channel][position] fp32 (positions contiguous per lane group).
template <int TM, int TN>
__device__ __forceinline__ void convb_store_partial(const ConvBArgs& a, f32x16 (&acc)[TM][TN], int slice, int mw, int nw,
                                                    int half, int l31) {
  float* base = a.ws + (long)slice * a.cout * a.ws_pitch - a.ws_n0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + l31;
    if (n >= a.ntot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ch < a.cout) st(base + (long)ch * a.ws_pitch + n, acc[i][j][r]);
      }
  }
}

// ... in the fragment layout (ConvBArgs::ws_frag).  `stores` of the caller: 8*TM buffer stores per wave when counted.
template <int TM>
__device__ __forceinline__ int convb_store_partial_frag(const ConvBArgs& a, f32x16 (&acc)[TM][2], int slice, int tile,
                                                        int wave, int lane, const BufRsrc& rws) {
  const unsigned so = (unsigned)((((long)slice * a.ws_ntl + (tile - a.ws_tile0)) * 4 + wave) * (TM * 8) * 64 * 16);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        gst16_buf(rws, (unsigned)((((i * 2 + j) * 4 + g) * 64 + lane) * 16), so,
                  make_uint4(__builtin_bit_cast(unsigned, q.x), __builtin_bit_cast(unsigned, q.y), __builtin_bit_cast(unsigned, q.z), __builtin_bit_cast(unsigned, q.w)));
      }
  return TM * 8;
}

// The epilogue of the commonest launch -- one destination per 32-row tile (`act` or a sibling segment's tensor), bias +
// folded BN (+ ReLU), nothing else (no raw copy, no residual, no second destination) -- as its own lean body:
// y = max(acc * scale + shift', 0 or -inf) with
// shift' = bias * scale + shift folded once per workgroup (convb_stage_params fold).  Round 4's cycle stamps
// (tools/exp/spanp_ts.py) had the general body at ~690 cycles per 8-channel block group (three parameter reads from LDS
// waited for on the spot, ~70 instructions with six uniform branches and spilled-SGPR reloads) -- 11 k cycles per
// 128 x 256 tile, 5.8 k of a 25 k-cycle inception 3x3 item.  Here the parameters of a 32-row tile are read in one go,
// the next tile's while this one is converted, and a group is 8 FMA + 8 max + 4 conversions + 2 lane swaps + the store.
template <int TM, bool SEG>
__device__ __forceinline__ int convb_epilogue_lean(const ConvBArgs& a, f32x16 (&acc)[TM][2], int mw, int m0, int n_lane,
                                                   int half, const float* Ep, int EPS, const FastDiv& d_sout) {
  const bool ok = n_lane < a.ntot;
  const unsigned nn = ok ? (unsigned)n_lane : 0u;
  const int img = (int)fastdiv(nn, d_sout), sp = (int)(nn - (unsigned)img * (unsigned)a.s_out);
  constexpr unsigned kAll = 0x7fffffffu;
  const unsigned v_plain = ok ? (unsigned)(view_base(a.act, img, sp) * 16) : kBufOob;
  float4 ps[2][4], ph[2][4];
  auto load_params = [&](int i, int slot) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = mw + i * 32 + 8 * g + 4 * half - m0;
      ps[slot][g] = *(const float4*)(Ep + EPS + c);
      ph[slot][g] = *(const float4*)(Ep + 2 * EPS + c);
    }
  };
  load_params(0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i + 1 < TM) load_params(i + 1, (i + 1) & 1);
    const int mt = mw + i * 32;
    // destination of this 32-row tile (wave-uniform): `act`, or a sibling's own tensor (as in convb_epilogue_wide)
    void* aptr = a.act.ptr;
    long astride_c = a.act.stride_c, astride_b = a.act.stride_b;
    int cb0 = 0, relu = a.relu;
    bool seg = false;
    if (SEG && a.nseg > 0 && mt >= a.seg_begin[0]) {   // (SEG = false: the caller's launches never carry segments)
      seg = true;
      aptr = a.seg_act[0].ptr; astride_c = a.seg_act[0].stride_c; astride_b = a.seg_act[0].stride_b;
      cb0 = a.seg_begin[0] / 8; relu = a.seg_relu[0];
#pragma unroll
      for (int q = 1; q < ECO_MAX_SEG; ++q) {
        void* const qp = a.seg_act[q].ptr;
        const long qc = a.seg_act[q].stride_c, qb = a.seg_act[q].stride_b;
        const int qbeg = a.seg_begin[q], qr = a.seg_relu[q];
        const bool take = q < a.nseg && mt >= qbeg;
        aptr = take ? qp : aptr; astride_c = take ? qc : astride_c; astride_b = take ? qb : astride_b;
        cb0 = take ? qbeg / 8 : cb0; relu = take ? qr : relu;
      }
    }
    const float floor_ = relu ? 0.0f : -__builtin_inff();   // ReLU as max(y, 0); no ReLU as max(y, -inf): one instruction either way
    const BufRsrc r_act = make_buf_rsrc(aptr, kAll);
    const unsigned v_act = !seg ? v_plain : ok ? (unsigned)(((long)img * astride_b + sp) * 16) : kBufOob;
    const unsigned cstep = (unsigned)(astride_c * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cbk = mt / 8 + g;
      const bool rows = cbk * 8 < a.cout;
      const float4 s4 = ps[i & 1][g], h4 = ph[i & 1][g];
      const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {h4.x, h4.y, h4.z, h4.w};
      unsigned pact[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = fmaxf(fmaf(acc[i][j][4 * g + q], sc[q], sh[q]), floor_);
        pact[j][0] = pack_bf16x2(y[0], y[1]);
        pact[j][1] = pack_bf16x2(y[2], y[3]);
      }
      permlane32_swap(pact[0][0], pact[1][0]);
      permlane32_swap(pact[0][1], pact[1][1]);
      gst16_buf(r_act, rows ? v_act : kBufOob, (unsigned)(cbk - cb0) * cstep, make_uint4(pact[0][0], pact[0][1], pact[1][0], pact[1][1]));
    }
  }
  return TM * 4;
}

// Second pass of a fragment-layout split: one workgroup per (tile, 32-row m-tile of its waves) -- the producing kernel's
// four waves, one of their TM m-tiles each, so that a tail of 16 tiles still spreads over 64 workgroups -- each lane sums
// its eight float4 over the slices in a fixed order (all loads of up to eight slices in flight: 1 KB per wave instruction)
// and the m-tile gets the epilogue an unsplit tile would have had.  WN = waves side by side in N (4: 128/96/64/32 x 256
// tiles; 2: 256 / 128 x 128 tiles).
template <int TM, int WN>
__global__ __launch_bounds__(256) void convb_splitk_reduce_frag_kernel(const ConvBArgs a) {
  constexpr int WM = 4 / WN, BM = 32 * TM * WM, BN = 64 * WN, BMP = (BM + 63) / 64 * 64;
  constexpr int kMaxSlices = 8;                 // (plans split at most eight ways)
  __shared__ __attribute__((aligned(16))) float Ep[3 * BMP];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), half = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tl = (int)blockIdx.x / TM, i = (int)blockIdx.x - tl * TM;   // local tile, m-tile of every wave
  const int tile = a.ws_tile0 + tl;
  const int mblk = tile % a.nblk_m, nblk = tile / a.nblk_m;
  const int m0 = mblk * BM, n0 = nblk * BN;
  convb_stage_params<BMP>(a, m0, Ep);
  f32x16 acc[1][2];
  const float4* base = (const float4*)a.ws + (((long)tl * 4 + wave) * (TM * 8) + i * 8) * 64 + lane;
  const long sstride = (long)a.ws_ntl * 4 * (TM * 8) * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 q[kMaxSlices];
#pragma unroll
      for (int sl = 0; sl < kMaxSlices; ++sl)
        q[sl] = sl < a.ws_slices ? ld(base + sl * sstride + (j * 4 + g) * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 sum = q[0];
#pragma unroll
      for (int sl = 1; sl < kMaxSlices; ++sl) { sum.x += q[sl].x; sum.y += q[sl].y; sum.z += q[sl].z; sum.w += q[sl].w; }
      acc[0][j][4 * g] = sum.x; acc[0][j][4 * g + 1] = sum.y; acc[0][j][4 * g + 2] = sum.z; acc[0][j][4 * g + 3] = sum.w;
    }
  __syncthreads();   // Ep
  convb_epilogue_wide<1>(a, acc, m0 + (wm * TM + i) * 32, m0, n0 + wn * 64 + lane, half, Ep, BMP, a.d_sout);
}

// Second pass of split-K: one thread per (position, 8-channel block) sums the slices in a fixed order and applies
// the epilogue; writes whole blocks.
__global__ __launch_bounds__(256) void convb_splitk_reduce_kernel(const ConvBArgs a) {
  const long total = (long)(a.cout / 8) * a.ws_pitch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cbk = (int)(idx / a.ws_pitch), nl = (int)(idx - (long)cbk * a.ws_pitch), n = a.ws_n0 + nl;
    int img, sp;
    decode_out(a, n, img, sp);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int ch0 = cbk * 8 + 4 * half;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = 0.0f;
        for (int sl = 0; sl < a.ws_slices; ++sl) s += ld(a.ws + ((long)sl * a.cout + ch0 + q) * a.ws_pitch + nl);
        v[q] = s + (a.bias ? ld(a.bias + ch0 + q) : 0.0f);
      }
      if (a.residual.ptr) {
        float rv[4];
        load_quad(a.residual.ptr, view_base(a.residual, img, sp) + (long)cbk * a.residual.stride_c, half, rv);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += rv[q];
      }
      if (a.raw.ptr) store_quad(a.raw.ptr, view_base(a.raw, img, sp) + (long)cbk * a.raw.stride_c, half, v);
      if (a.act.ptr) {
        eco_view av = a.act;       // sibling launches: the channel block's own destination
        int relu = a.relu, cb0 = 0;
        if (a.nseg > 0 && cbk * 8 >= a.seg_begin[0]) {
          av = a.seg_act[0]; relu = a.seg_relu[0]; cb0 = a.seg_begin[0] / 8;
#pragma unroll
          for (int sq = 1; sq < ECO_MAX_SEG; ++sq) {
            const eco_view qv = a.seg_act[sq];
            const int qr = a.seg_relu[sq], qbeg = a.seg_begin[sq];
            const bool take = sq < a.nseg && cbk * 8 >= qbeg;
            av.ptr = take ? qv.ptr : av.ptr; av.stride_b = take ? qv.stride_b : av.stride_b;
            av.stride_c = take ? qv.stride_c : av.stride_c; av.stride_t = take ? qv.stride_t : av.stride_t;
            av.t = take ? qv.t : av.t; relu = take ? qr : relu; cb0 = take ? qbeg / 8 : cb0;
          }
        }
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[q] = v[q] * (a.bn_scale ? ld(a.bn_scale + ch0 + q) : 1.0f) + (a.bn_scale ? ld(a.bn_shift + ch0 + q) : 0.0f);
          if (relu) y[q] = fmaxf(y[q], 0.0f);
        }
        store_quad(av.ptr, view_base(av, img, sp) + (long)(cbk - cb0) * av.stride_c, half, y);
        if (a.act2.ptr) store_quad(a.act2.ptr, view_base(a.act2, img, sp) + (long)cbk * a.act2.stride_c, half, y);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 storage: the same implicit GEMM with both operands staged by LDS-DMA (global_load_lds_dwordx4) and larger
// wave tiles.  Why: with 16x the MFMA rate the register-staged kernel above is bound by the LDS array, not by the
// matrix cores -- a 2x2 wave tile reads 1 KB of fragments per MFMA and every staged KB costs a 13-cycle
// ds_write_b128 issue plus 8 LDS-array cycles: reads + writes fill the 256 B/clk array at ~36 % MFMA utilisation
// (measured: 0.9 PFLOP/s on res3).  Here
//   * a wave owns TM x TN = 4x2 tiles (128 channels x 64 positions, 128 accumulator registers): 6 fragment reads
//     per 8 MFMAs instead of 4 per 4,
//   * the stage's weight rows and position blocks go global -> LDS without passing through VGPRs: one
//     instruction moves 64 lanes x 16 B into 1 KB of LDS (half the LDS-array cycles of ds_write_b128, no VGPR
//     staging, no store issue), consecutive lanes = consecutive channels / positions, which is exactly the
//     [block][m or n] fragment layout (conflict-free ds_read_b128),
//   * zero padding costs nothing: a lane whose tap falls outside the image points its DMA at a 16-byte zero page,
//   * every lane serves ONE position for the whole reduction (wave w stages positions [64*(w % (BN/64)), +64) of
//     the blocks kb = w / (BN/64) (+ 4/(BN/64)...)), so the per-stage address is one add.
// Pipeline (three stage buffers): the DMA of stage s+2 is issued at the top of stage s, so every piece has two
// stages of MFMAs to land.  Top of stage s: wait until all but this wave's newest stage of pieces has landed
// (s_waitcnt vmcnt(P)), then one workgroup barrier -- it publishes stage s to every wave and, because every wave
// has finished the MFMAs of stage s-1 by then, frees that stage's buffer for the DMA of stage s+2.  The buffers
// are separate __shared__ arrays and the stage loop is unrolled by three so that every access names its array
// statically: hipcc orders a ds_read behind outstanding LDS-DMA by alias analysis, and with one array it drains
// the whole DMA queue in front of every fragment read (measured in the ISA: s_waitcnt vmcnt(0) before the first
// ds_read of each stage).
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void convb_dma_kernel(const ConvBArgs a, const uint4* zero_page) {
  ECO_CLOCK("dma");
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int BMP = (BM + 63) / 64 * 64;   // weight rows staged per block (whole 64-lane pieces)
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BN == 128 || BN == 256, "");
  constexpr int NCH = BN / 64;               // 64-position chunks of the tile
  constexpr int B_PER_WAVE = kCbs * NCH / 4; // DMA pieces of the position operand per wave and stage (2 or 4)
  constexpr int KB_STEP = 4 / NCH;           // wave w stages blocks kb = w / NCH + q * KB_STEP
  constexpr int A_PER_WAVE = kCbs * BMP / 64 / 4;   // DMA pieces of the weight operand per wave and stage
  constexpr int P = B_PER_WAVE + A_PER_WAVE;        // pieces in flight per wave and stage

  __shared__ __attribute__((aligned(16))) uint4 A0[kCbs * BMP], A1[kCbs * BMP], A2[kCbs * BMP];
  __shared__ __attribute__((aligned(16))) uint4 B0[kCbs * BN], B1[kCbs * BN], B2[kCbs * BN];

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int ntiles = a.nblk_m * a.nblk_n;
  const int slice = (int)blockIdx.x / ntiles;
  const int tile = xcd_remap((int)blockIdx.x - slice * ntiles, ntiles);
  const int mblk = tile % a.nblk_m, nblk = tile / a.nblk_m;
  const int m0 = mblk * BM, n0 = nblk * BN;
  constexpr int BMP_E = (BM + 63) / 64 * 64;
  __shared__ __attribute__((aligned(16))) float Ep[3 * BMP_E];   // bias / BN scale / BN shift of this workgroup's rows
  convb_stage_params<BMP_E>(a, m0, Ep, TN == 2 && a.lean != 0 && a.ksplit == 1);
#ifndef ECO_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // Ep's ds_writes retired before the first (non-draining) barrier
#endif
  const int s_begin = (int)((long)slice * a.nstages / a.ksplit);
  const int s_end = (int)((long)(slice + 1) * a.nstages / a.ksplit);

  // ---- the one position this lane stages: chunk (wave % NCH), lane ----
  const int chunk = wave % NCH, kb0 = wave / NCH;
  const int khw = a.kh * a.kw;
  long in_base = 0;
  unsigned long long mask = 0ull;
  {
    const int n = n0 + chunk * 64 + lane;
    if (n < a.ntot) {
      int img, sp;
      decode_out(a, n, img, sp);
      const int ow = sp % a.Wo, t = sp / a.Wo;
      const int oh = t % a.Ho, od = t / a.Ho;
      const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
      in_base = (long)img * a.img_stride_in + ((long)id0 * a.Hi + ih0) * a.Wi + iw0;
      unsigned long long mw_ = 0ull, mhw = 0ull;
      for (int xx = 0; xx < a.kw; ++xx) mw_ |= (unsigned long long)((unsigned)(iw0 + xx) < (unsigned)a.Wi) << xx;
      for (int y = 0; y < a.kh; ++y)
        if ((unsigned)(ih0 + y) < (unsigned)a.Hi) mhw |= mw_ << (y * a.kw);
      for (int z = 0; z < a.kd; ++z)
        if ((unsigned)(id0 + z) < (unsigned)a.Di) mask |= mhw << (z * khw);
    }
  }
  const uint4* const xv = (const uint4*)a.x;
  // out-of-image taps read the zero page: one integer select per piece (no divergent second instruction)
  const long zoff = (long)(((intptr_t)zero_page_ptr(zero_page) - (intptr_t)xv) / 16);
  // descriptor form (a.dma_buf; as the persistent span kernel's glds16_buf): no 64-bit lane arithmetic and no zero page --
  // a piece is SGPR base + SGPR offset + one lane register that stays put for the workgroup's lifetime, and a tap outside
  // the image is an out-of-range offset that writes zeros
  const BufRsrc rxb = make_buf_rsrc((const char*)a.x - a.dma_bias, a.dma_x_bytes + a.dma_bias);
  const BufRsrc rwb = make_buf_rsrc(a.wp, a.dma_wp_bytes);
  const unsigned xvo = (unsigned)(in_base * 16 + (long)a.dma_bias);   // (dma_buf: 0 <= in_base * 16 + bias < 2^31)
  unsigned wvo[A_PER_WAVE];
#pragma unroll
  for (int q = 0; q < A_PER_WAVE; ++q) {
    const int piece = wave + 4 * q, row = piece / (BMP / 64), mc = piece % (BMP / 64);
    wvo[q] = (unsigned)(row * a.mpad + m0 + mc * 64 + lane) * 16u;
  }
  const unsigned wstage16 = (unsigned)(kCbs * a.mpad) * 16u;

  // the stage whose DMA is issued next: uniform (cg, tap) walk
  int l_cg = s_begin / a.taps, l_tap = s_begin - l_cg * a.taps;
  int l_kx = l_tap % a.kw, l_ky = (l_tap / a.kw) % a.kh, l_kz = l_tap / khw;
  int l_stage = s_begin;
  auto issue_stage = [&](uint4* Ab, uint4* Bb) {   // DMA of stage l_stage into (Ab, Bb), then step the walk
    const long toff = ((long)l_kz * a.Hi + l_ky) * a.Wi + l_kx;
    // all-ones where the tap is inside the image.  Opaque to the optimiser: a recognisable select gets turned
    // into a divergent branch with one copy of the DMA instruction in each arm, and then a wave with both kinds
    // of lanes issues more pieces than wait_dma_all_but<P> accounts for.
    if (a.dma_buf) {
      const unsigned vo = ((mask >> l_tap) & 1ull) ? xvo : kBufOob;
#pragma unroll
      for (int q = 0; q < B_PER_WAVE; ++q) {
        const int kb = kb0 + q * KB_STEP;
        const int cb = min(l_cg * kCbs + kb, a.cblocks - 1);   // zero-weight padding group: any finite data
        glds16_buf(rxb, vo, (unsigned)((toff + (long)cb * a.cb_stride_in) * 16), Bb + kb * BN + chunk * 64);
      }
#pragma unroll
      for (int q = 0; q < A_PER_WAVE; ++q) {
        const int piece = wave + 4 * q;       // piece = row * (BMP/64) + m-chunk
        const int row = piece / (BMP / 64), mc = piece % (BMP / 64);
        glds16_buf(rwb, wvo[q], (unsigned)l_stage * wstage16, Ab + row * BMP + mc * 64);
      }
    } else {
    long sel = -(long)((mask >> l_tap) & 1ull);
    ECO_OPAQUE64(sel);
#pragma unroll
    for (int q = 0; q < B_PER_WAVE; ++q) {
      const int kb = kb0 + q * KB_STEP;
      const int cb = min(l_cg * kCbs + kb, a.cblocks - 1);   // zero-weight padding group: any finite data
      const long real = in_base + toff + (long)cb * a.cb_stride_in;
      glds16(xv + (zoff ^ ((zoff ^ real) & sel)), Bb + kb * BN + chunk * 64);
    }
#pragma unroll
    for (int q = 0; q < A_PER_WAVE; ++q) {
      const int piece = wave + 4 * q;       // piece = row * (BMP/64) + m-chunk
      const int row = piece / (BMP / 64), mc = piece % (BMP / 64);
      glds16(a.wp + ((long)l_stage * kCbs + row) * a.mpad + m0 + mc * 64 + lane, Ab + row * BMP + mc * 64);
    }
    }
    ++l_stage;
    ++l_tap;
    if (++l_kx == a.kw) {
      l_kx = 0;
      if (++l_ky == a.kh) {
        l_ky = 0;
        if (++l_kz == a.kd) { l_kz = 0; l_tap = 0; ++l_cg; }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto compute = [&](const uint4* Ab, const uint4* Bb) {
    // both k-steps' fragments are read before the first one's products issue (the second read's latency passes under
    // eight MFMAs: the strided 3x3x3 launches -4 % on top of the descriptor form)
    uint4 af[kCbs / 2][TM], bf[kCbs / 2][TN];
#pragma unroll
    for (int ks = 0; ks < kCbs / 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = Ab[(2 * ks + half) * BMP + (wm * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[ks][j] = Bb[(2 * ks + half) * BN + (wn * TN + j) * 32 + l31];
    }
    sched_fence();
#pragma unroll
    for (int ks = 0; ks < kCbs / 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j]);
      sched_fence();
    }
  };
  // One stage: its own pieces (issued two stages ago) have landed once at most the newest stage's P are pending.
  auto stage = [&](int s, const uint4* Ab, const uint4* Bb, uint4* An, uint4* Bn) {
    if (s + 1 < s_end) wait_dma_all_but<P>(); else wait_dma_all_but<0>();
    wg_barrier_nodrain();
    if (s + 2 < s_end) issue_stage(An, Bn);
    sched_fence();
    compute(Ab, Bb);
  };

  if (s_begin < s_end) {
    issue_stage(A0, B0);
    if (s_begin + 1 < s_end) issue_stage(A1, B1);
    for (int s = s_begin; s < s_end; s += 3) {
      stage(s, A0, B0, A2, B2);
      if (s + 1 < s_end) stage(s + 1, A1, B1, A0, B0);
      if (s + 2 < s_end) stage(s + 2, A2, B2, A1, B1);
    }
  }
  if (!(s_begin < s_end)) __syncthreads();   // (no stage ran: Ep has not been published by a barrier yet)
  if (a.ksplit > 1) {
    if constexpr (TN == 2) {
      if (a.ws_frag) {
        const BufRsrc rws = make_buf_rsrc(a.ws, 0x7fffffffu);
        convb_store_partial_frag<TM>(a, acc, slice, mblk + nblk * a.nblk_m, wave, lane, rws);
        return;
      }
    }
    convb_store_partial<TM, TN>(a, acc, slice, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31);
  } else {
    if constexpr (TN == 2) {   // 64-position wave tiles: whole 16-byte blocks per lane
      if (a.lean) {   // (ksplit == 1 here: Ep carries the folded shift)
        convb_epilogue_lean<TM, true>(a, acc, m0 + wm * TM * 32, m0, n0 + wn * 64 + lane, half, Ep, BMP_E, a.d_sout);
        return;
      }
      if (a.wide) {
        convb_epilogue_wide<TM>(a, acc, m0 + wm * TM * 32, m0, n0 + wn * 64 + lane, half, Ep, BMP_E, a.d_sout);
        return;
      }
    }
    convb_epilogue<TM, TN>(a, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, half, l31, Ep, BMP_E, m0);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent span kernel (round 4): the same arithmetic as convb_span_kernel, restructured around what the probe builds
// of round 4 measured (profiles/r04_notes.md): with the MFMAs REMOVED the span kernel still took 0.87 of conv2_3x3's
// 1.07 ms and 0.50 of res3b_1's 0.73 ms -- per tap ~400 cycles of address arithmetic, wait selection and loop control in
// every wave next to 512 cycles of MFMA issue, and per tile ~12 k cycles of exposed set-up (integer divisions, epilogue
// parameters, the first span's trip to HBM) that an 18-tap tile (conv2_3x3, the inception 3x3s) pays every 9 k cycles of
// matrix work.  Here
//   * one workgroup per slot (2 per CU) walks its tiles: the next tile's span and first two weight taps are issued while
//     the current tile's last group runs, its index arithmetic (fastdiv, no software divides) sits under that DMA, the
//     epilogue parameters are staged once per workgroup;
//   * both operands go through buffer descriptors (glds16_buf): a piece's address is SGPR base + SGPR offset + a lane
//     register that stays put for the whole tile (the flat form needed a 64-bit VALU add and a zero-page select per
//     piece), and out-of-volume planes / tile overhang are the descriptor's range check writing zeros;
//   * the three taps of a kernel row are unrolled with compile-time ring slots and LDS immediates (a row starts at ring
//     slot 0 because 3 % 3 == 0; the span pitch in LDS is a constant 384 positions); wait counts follow what the previous
//     issue slot actually issued.
// Tiles: 32*TM channels x 256 positions, four waves side by side in N, as before.  Items = (slice, tile); with split-K an
// item ends in partial sums instead of the epilogue.
struct SpanPArgs {
  unsigned x_bytes, wp_bytes;
  int ntiles;                      // nblk_m * nblk_n
  // items: tiles [0, t_main) whole (epilogue), then the last t_tail tiles cut into kb slices of their groups each (partial
  // sums + the reduce launch): a whole-tensor split-K is t_main = 0, a plain launch t_tail = 0, and a launch whose tile
  // count leaves a partial last round per CU splits just that remainder -- e.g. res4's 784 tiles on 256 CUs: 768 + 16 x 8
  int t_main, t_tail, kb, nitems;
  // Dynamic items (null: item k of a workgroup = L + k * grid): ctr[m] = tickets handed out for M-block m, ctr[4] =
  // workgroups that have left; zero before the launch, zeroed again by the last workgroup to leave.
  unsigned* ctr;
  FastDiv d_sout, d_hw, d_w, d_ks, d_kd, d_tail;
};

template <int TM>
__global__ __launch_bounds__(256, 2) void convb_spanp_kernel(const ConvBArgs a, const SpanPArgs pa, int span_pieces) {
  ECO_CLOCK("spanp");
  constexpr int TN = 2;
  constexpr int BM = 32 * TM, BN = 256;
  constexpr int BMP = (BM + 63) / 64 * 64;
  constexpr int APW = kCbs * BMP / 64 / 4;   // weight pieces per wave and tap: 1 or 2
  constexpr int SPITCH = 384;                // positions per span row in LDS (span_pieces <= 6)
  constexpr int NB = 3;                      // weight ring slots
  constexpr int T2 = 9;
  static_assert(T2 % NB == 0, "a group must start at ring slot 0");

  __shared__ __attribute__((aligned(16))) float Ep[3 * BMP];   // bias / BN scale / BN shift of this workgroup's rows
  ECO_DYNAMIC_LDS(lds_f);
  uint4* const Aw = (uint4*)lds_f;                 // [NB][kCbs][BMP]
  uint4* const Bsp = Aw + NB * kCbs * BMP;         // [2][kCbs][SPITCH]

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // ---- this workgroup's items: item k = L + k * grid.  The logical id L (a bijection of the block index; grid is a
  //      multiple of 8 * nblk_m) orders blocks by (chunk of 8 * nblk_m consecutive blocks, XCD = b % 8, M-block): the
  //      M-blocks of one position tile land on one XCD (hardware places block b on XCD b % 8), AND the workgroups with
  //      L < R are a run of consecutive blocks for every R -- so the items of a last, partial round spread over all eight
  //      XCDs and one per CU (blocks b and b + grid/2 share a CU).  Round 4 first used L = (b % 8) * (grid / 8) + b / 8:
  //      res4's 272 second-round items then all sat on XCDs 0-4, four items on each of their CUs and two on the others'
  //      (1.2 waves per SIMD over the launch, profiles/r04_bf16_pmc_sq.csv). ----
  const int grid = (int)gridDim.x;
  const int bx = (int)blockIdx.x;
  const int L = ((bx / (8 * a.nblk_m)) * 8 + bx % 8) * a.nblk_m + (bx / 8) % a.nblk_m;
  const int mblk = L % a.nblk_m;                   // (grid and ntiles are multiples of nblk_m: every item of L has this M-block)
  const int m0 = mblk * BM;
  // The first item of a workgroup is L.  After it: item k = L + k * grid (pa.ctr null), or the next item of this
  // M-block nobody has taken yet -- id (grid / nblk_m + ticket) * nblk_m + mblk, tickets from pa.ctr[mblk].  The two
  // workgroups of a CU (blocks b and b + grid/2) share its SIMDs wave for wave and the instruction arbiter prefers the
  // older wave: with equal shares the first-dispatched workgroup ran its items at the speed of a lone wave and its partner
  // on what was left (res3b_1: 6.25 items in 424 us against 6.0 in 538 us, the last 114 us alone; res4: two items against
  // one -- tools/exp/clock_probe2.sh).  Taking items as they come, both finish together, and a K-split tail's short items
  // go last by themselves.
  int item = L;
  const bool dynamic = pa.ctr != nullptr;
  auto leave = [&]() {     // (one lane) count this workgroup out; the last one out clears the launch's counters
    if (dynamic && tid == 0) {
      const unsigned gone = counter_fetch_add(pa.ctr + 4, 1u);
      if (gone == (unsigned)grid - 1u) {
#pragma unroll
        for (int q = 0; q < 5; ++q) counter_store(pa.ctr + q, 0u);
      }
    }
  };
  if (item >= pa.nitems) { leave(); return; }      // (fewer items than workgroups: uniform exit, before any barrier)
  __shared__ int next_id[1];                       // id of the item after this one (written by thread 0)
  const int id0 = (grid / a.nblk_m) * a.nblk_m + mblk;   // ticket 0's id
  convb_stage_params<BMP>(a, m0, Ep, a.lean != 0);
#ifndef ECO_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // Ep's ds_writes retired before the first (non-draining) barrier
#endif

  const BufRsrc rx = make_buf_rsrc(a.x, pa.x_bytes), rw = make_buf_rsrc(a.wp, pa.wp_bytes);
  const BufRsrc rws = make_buf_rsrc(a.ws, 0x7fffffffu);   // partial sums (fragment layout: the plan keeps them below 2 GB)
  const int hw = a.Hi * a.Wi, halo = a.Wi + 1;
  const int ngroups = (a.nstages / a.taps) * a.kd;
  const unsigned cbs16 = (unsigned)a.cb_stride_in * 16u;
  const unsigned wstage16 = (unsigned)(kCbs * a.mpad) * 16u;      // bytes of one tap stage of packed weights
  const int nchunks = wave + 4 < span_pieces ? 2 : 1;             // span pieces `wave` and `wave + 4`
  const int SPW = kCbs * nchunks;                                  // span DMA pieces of this wave per group

  // weight pieces of this wave: piece = wave + 4q -> (row, 64-channel chunk); lane offset fixed for the workgroup
  unsigned wv[APW];
  int wl[APW];                                                     // LDS index inside a ring slot
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int piece = wave + 4 * q, row = piece / (BMP / 64), mc = piece % (BMP / 64);
    wv[q] = (unsigned)(row * a.mpad + m0 + mc * 64 + lane) * 16u;
    wl[q] = row * BMP + mc * 64;
  }
  const int a_lane = half * BMP + l31;
  const int b_lane = half * SPITCH + wave * 64 + l31;

  // ---- per-item lane state ----
  struct Item {
    int n0, tile, slice, g_begin, g_end;   // slice < 0: a whole tile (epilogue); else slice `slice` of pa.kb (partial sums)
    unsigned spv[2];     // byte offset of span element (chunk c, this lane) at depth shift 0
    int spd[2];          // its depth index (hugely negative: never valid)
    unsigned fmask[TN];  // in-plane tap masks of the lane's fragment positions: bit y*3 + x
  };
  auto make_item = [&](int id) {
    Item it;
    int tile = id;
    it.slice = -1;
    it.g_begin = 0;
    it.g_end = ngroups;
    if (id >= pa.t_main) {
      const int j = id - pa.t_main;
      const int sl = (int)fastdiv((unsigned)j, pa.d_tail);
      tile = pa.t_main + j - sl * pa.t_tail;
      it.slice = sl;
      it.g_begin = (int)fastdiv((unsigned)(sl * ngroups), pa.d_ks);
      it.g_end = (int)fastdiv((unsigned)((sl + 1) * ngroups), pa.d_ks);
    }
    it.tile = tile;
    it.n0 = (tile / a.nblk_m) * BN;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int v = it.n0 - halo + (wave + 4 * c) * 64 + lane;
      it.spv[c] = 0u;
      it.spd[c] = -(1 << 20);
      if (v >= 0 && v < a.ntot) {
        const unsigned img = fastdiv((unsigned)v, pa.d_sout), sp = (unsigned)v - img * (unsigned)a.s_out;
        it.spv[c] = (unsigned)((long)img * a.img_stride_in + sp) * 16u;
        it.spd[c] = (int)fastdiv(sp, pa.d_hw);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = it.n0 + wave * 64 + j * 32 + l31;
      const bool inside = n < a.ntot;
      const unsigned nn = inside ? (unsigned)n : 0u;
      const unsigned img = fastdiv(nn, pa.d_sout), sp = nn - img * (unsigned)a.s_out;
      const unsigned r = sp - fastdiv(sp, pa.d_hw) * (unsigned)hw;
      const int h = (int)fastdiv(r, pa.d_w), w = (int)r - h * a.Wi;
      unsigned mw_ = 0u, fm = 0u;
#pragma unroll
      for (int xx = 0; xx < 3; ++xx) mw_ |= (unsigned)((unsigned)(w - 1 + xx) < (unsigned)a.Wi) << xx;
#pragma unroll
      for (int y = 0; y < 3; ++y)
        if ((unsigned)(h - 1 + y) < (unsigned)a.Hi) fm |= mw_ << (3 * y);
      it.fmask[j] = inside ? fm : 0u;
    }
    return it;
  };

  // ---- DMA issue ----
  auto issue_span = [&](unsigned sv0, unsigned sv1, int sd0, int sd1, int cg, int z, int sbuf) {
    const int dz = z - a.pd;
    const unsigned zshift = (unsigned)(dz * hw * 16);          // (two's complement: added to spv only where the plane exists)
    const unsigned sv[2] = {sv0, sv1};
    const int sd[2] = {sd0, sd1};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nchunks) {
        const unsigned vo = (unsigned)(sd[c] + dz) < (unsigned)a.Di ? sv[c] + zshift : kBufOob;
#pragma unroll
        for (int kb = 0; kb < kCbs; ++kb) {
          const int cb = min(cg * kCbs + kb, a.cblocks - 1);   // zero-weight padding group: any finite data
          glds16_buf(rx, vo, (unsigned)cb * cbs16, Bsp + (sbuf * kCbs + kb) * SPITCH + (wave + 4 * c) * 64);
        }
      }
    }
  };
  auto issue_weights = [&](int stage, int abuf) {
    const unsigned so = (unsigned)stage * wstage16;
#pragma unroll
    for (int q = 0; q < APW; ++q) glds16_buf(rw, wv[q], so, Aw + abuf * kCbs * BMP + wl[q]);
  };
  // all but the newest `n` memory operations of this wave have completed (n: one of the four counts an issue slot can
  // have; the counter retires in order, LDS-DMA pieces and stores alike)
  auto wait_newest = [&](int n) {
    if (n == 0) wait_dma_all_but<0>();
    else if (n == APW) wait_dma_all_but<APW>();
    else if (n == APW + kCbs) wait_dma_all_but<APW + kCbs>();
    else wait_dma_all_but<APW + 2 * kCbs>();
  };
  // the same at the first tap behind an epilogue, whose `st` store instructions (a multiple of S = 4*TM, counted by
  // convb_epilogue_wide) are newer than every piece that tap needs: any immediate <= n + st is a correct wait
  constexpr int S = 4 * TM;
  auto wait_newest_behind_stores = [&](int n, int st) {
    const int m = (n >= APW ? APW : 0) + st;
    if (m >= APW + 3 * S) wait_dma_all_but<APW + 3 * S>();
    else if (m >= APW + 2 * S) wait_dma_all_but<APW + 2 * S>();
    else if (m >= APW + S) wait_dma_all_but<APW + S>();
    else if (m >= APW) wait_dma_all_but<APW>();
    else wait_dma_all_but<0>();
  };

  f32x16 acc[TM][TN];
  Item cur = make_item(item);
  int cg = (int)fastdiv((unsigned)cur.g_begin, pa.d_kd), z = cur.g_begin - cg * a.kd;
  issue_span(cur.spv[0], cur.spv[1], cur.spd[0], cur.spd[1], cg, z, 0);
  issue_weights(cg * a.taps + z * T2, 0);
  issue_weights(cg * a.taps + z * T2 + 1, 1);
  int newest = APW;          // pieces issued after the ones the next tap needs
  int stores = 0;            // store instructions of the previous item's epilogue, still in front of the next wait
  int sbuf = 0;
#if defined(ECO_SPANP_PRIO) && !defined(ECO_EMU)
  // The two workgroups of a CU (blocks b and b + grid/2) share its SIMDs wave for wave, and the instruction arbiter
  // prefers the older wave: the first-dispatched workgroup ran its items at the speed of a lone wave (res3b_1: 6.25 items in
  // 424 us) and the other one on what was left (6.0 items in 538 us, the last 114 us alone -- tools/exp/clock_probe2.sh).
  // They take the higher wave priority in turns, tap by tap (item by item left the workgroup with more items on the low
  // turn while its partner was still there: res4 +10 %).
  const int prio_half = bx >= grid / 2 ? 1 : 0;
#endif

  for (;;) {
    // the item after this one (uniform): by arithmetic, or -- dynamic -- read from next_id behind the first barrier of this
    // item's last group (wave 0 draws and publishes it ahead of that barrier)
    int nitem = item + grid;
    bool have_next_item = nitem < pa.nitems;
    Item nxt = cur;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    for (int g = cur.g_begin; g < cur.g_end; ++g) {
      // ---- the group after this one: (channel group, depth tap) and whose lane state stages its span ----
      const bool last_group = g + 1 == cur.g_end;
      bool have_next = true;
      int ncg = cg, nz = z + 1;
      if (nz == a.kd) { nz = 0; ++ncg; }
      auto resolve_next = [&]() {                      // the next item's lane state and first (channel group, depth tap)
        have_next = have_next_item;
        if (have_next_item) {
          nxt = make_item(nitem);                      // VALU under the DMA / the other workgroup's MFMAs
          ncg = (int)fastdiv((unsigned)nxt.g_begin, pa.d_kd);
          nz = nxt.g_begin - ncg * a.kd;
        }
      };
      if (last_group && !dynamic) resolve_next();
      unsigned nsv0 = last_group ? nxt.spv[0] : cur.spv[0], nsv1 = last_group ? nxt.spv[1] : cur.spv[1];
      int nsd0 = last_group ? nxt.spd[0] : cur.spd[0], nsd1 = last_group ? nxt.spd[1] : cur.spd[1];
      const int stage0 = cg * a.taps + z * T2;
      int nstage0 = ncg * a.taps + nz * T2;
      // three kernel rows, the three taps of a row unrolled (ring slot = column because 3 % NB == 0): compile-time LDS
      // immediates without nine copies of the body competing for registers
#pragma unroll 1
      for (int y = 0; y < 3; ++y) {
        const uint4* brow = Bsp + sbuf * kCbs * SPITCH + y * a.Wi + b_lane;
        unsigned rmask[TN];   // this row's three mask bits of each fragment position
#pragma unroll
        for (int j = 0; j < TN; ++j) rmask[j] = cur.fmask[j] >> (3 * y);
        static_for<3>([&](auto xc) {
          constexpr int xx = decltype(xc)::value;
          constexpr int abuf = xx;                     // t2 % NB with t2 = 3y + xx
          const int t2 = 3 * y + xx;
#if defined(ECO_SPANP_PRIO) && !defined(ECO_EMU)
          if ((xx ^ y ^ prio_half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
          if (stores) { wait_newest_behind_stores(newest, stores); stores = 0; }
          else wait_newest(newest);
          // Dynamic items: the next item's ticket is drawn here, at the first tap of this item's last group -- the last moment
          // that keeps the next item's span and first weights in flight under this group -- by wave 0 on the scalar unit
          // (s_atomic_add: ~600 cycles, nothing in the vector memory counter; the other waves meet it at the barrier below).
          // Drawn an item ahead, every workgroup had reserved its second item before any work was done and res4 (1.5 items
          // per workgroup) ran 28 % slower than with static shares; the compiler's own returning atomic is waited for with
          // vmcnt(0) where it is issued.
          if (xx == 0 && y == 0 && last_group && dynamic && wave == 0) {
#ifdef ECO_EMU
            if (tid == 0)
#endif
            {
              const unsigned ticket = counter_draw_wave(pa.ctr + mblk);
              if (lane == 0) next_id[0] = id0 + (int)ticket * a.nblk_m;
            }
#ifndef ECO_EMU
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before the (non-draining) barrier below
#endif
          }
          wg_barrier_nodrain();
          if (xx == 0 && y == 0 && last_group && dynamic) {   // behind this barrier thread 0's next_id[0] is visible
            nitem = uniform(next_id[0]);
            have_next_item = nitem < pa.nitems;
            resolve_next();
            nsv0 = nxt.spv[0]; nsv1 = nxt.spv[1]; nsd0 = nxt.spd[0]; nsd1 = nxt.spd[1];
            nstage0 = ncg * a.taps + nz * T2;
          }
          int cnt = 0;
          if (xx == 0 && y == 0 && have_next) { issue_span(nsv0, nsv1, nsd0, nsd1, ncg, nz, sbuf ^ 1); cnt += SPW; }
          if (t2 + 2 < T2) { issue_weights(stage0 + t2 + 2, (xx + 2) % NB); cnt += APW; }
          else if (have_next) { issue_weights(nstage0 + t2 + 2 - T2, (xx + 2) % NB); cnt += APW; }
          newest = cnt;
          sched_fence();
          const uint4* Ab = Aw + abuf * kCbs * BMP + a_lane;
          const uint4* Bb = brow + xx;
          uint4 af[2][TM], bf[2][TN];
          auto read_frags = [&](int slot, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = Ab[2 * ks * BMP + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = Bb[2 * ks * SPITCH + j * 32];
          };
          read_frags(0, 0);
#pragma unroll
          for (int ks = 0; ks < kCbs / 2; ++ks) {
            if (ks + 1 < kCbs / 2) read_frags((ks + 1) & 1, ks + 1);
            sched_fence();
#pragma unroll
            for (int j = 0; j < TN; ++j) {   // all-ones where this tap is inside the plane (an AND per dword: no exec-masked reads)
              const unsigned okm = (unsigned)((int)(rmask[j] << (31 - xx)) >> 31);
              uint4& q = bf[ks & 1][j];
              q = make_uint4(q.x & okm, q.y & okm, q.z & okm, q.w & okm);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j]);
            sched_fence();
          }
        });
      }
      sbuf ^= 1;
      cg = ncg; z = nz;
    }
    if (cur.slice >= 0) {
      if (a.ws_frag) stores = convb_store_partial_frag<TM>(a, acc, cur.slice, cur.tile, wave, lane, rws);   // (counted: stepped over)
      else {   // (compiler-counted stores: the next wait drains them.  The row offset is made opaque here: as a loop invariant its
               // 128 row addresses were computed when the kernel starts and spilled -- 60 scratch stores per workgroup for a
               // fallback path)
        int m0p = m0;
        ECO_OPAQUE(m0p);
        convb_store_partial<TM, TN>(a, acc, cur.slice, m0p, cur.n0 + wave * 64, half, l31);
      }
    } else
    if (a.lean)
      stores = convb_epilogue_lean<TM, false>(a, acc, m0, m0, cur.n0 + wave * 64 + lane, half, Ep, BMP, pa.d_sout);
    else
      stores = convb_epilogue_wide<TM>(a, acc, m0, m0, cur.n0 + wave * 64 + lane, half, Ep, BMP, pa.d_sout);
    if (!have_next_item) break;
    cur = nxt;
    item = nitem;
  }
  leave();
}

// ------------------------------------------------------------------------------------------------------------------
// Stem input: fp32 N,3,H,W (the `data` blob, VideoData contract) -> zero-padded pixel-interleaved image
//   P[f][h + 3][w + 3][4] (channel 3 = 0), rows of W + 8 pixels, H + 6 rows
// in the path's storage type.  For conv1_7x7_s2 (stride 2, pad 3) the seven taps of kernel row ky of output
// (oh, ow) are then the 28 consecutive elements that start at block (2*oh + ky)*(W+8)/2 + ow: the stem runs as an
// ordinary blocked convolution with 4 "channel blocks" (kx pairs), a (1,7,1) kernel over rows, stride (1,2,1), no
// padding and no predicates.  One thread per output block (2 pixels).  HBM-bound: 24 B read, 16 / 32 B written.
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* x, void* y, long frames, int H, int W) {
  const int wb = (W + 8) / 2, hp = H + 6;
  const long total = frames * hp * wb;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i % wb);
    const long t = i / wb;
    const int r = (int)(t % hp);
    const long f = t / hp;
    const int h = r - 3;
    float v[8];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const int w = 2 * b + px - 3;
      const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[4 * px + c] = ok ? ld(x + ((f * 3 + c) * H + (ok ? h : 0)) * (long)W + (ok ? w : 0)) : 0.0f;
      v[4 * px + 3] = 0.0f;
    }
    const float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    store_quad(y, i, 0, lo);
    store_quad(y, i, 1, hi);
  }
}

static bool is_stem(const eco_conv_geom* g) {
  return g->cin == 3 && g->in[0] == 1 && g->kernel[0] == 1 && g->kernel[1] == 7 && g->kernel[2] == 7 &&
         g->stride[1] == 2 && g->stride[2] == 2 && g->pad[1] == 3 && g->pad[2] == 3 && g->in[2] % 2 == 0;
}

static int ns_of(int dt) { return dt == ECO_DT_BF16 ? 1 : 0; }   // (1 = bf16 storage: the only storage type of this path)

static uint16_t host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
}  // namespace eco

using namespace eco;

static int validate_convb_geom(const eco_conv_geom* g, int dt) {
  ECO_REQUIRE(g != nullptr, "convb: null geometry");
  ECO_REQUIRE(ns_of(dt) != 0, "convb: storage type must be ECO_DT_BF16 (got %d)", dt);
  ECO_REQUIRE(g->n > 0 && g->cin > 0 && g->cout > 0, "convb: n/cin/cout must be positive");
  long taps = 1;
  for (int i = 0; i < 3; ++i) {
    ECO_REQUIRE(g->in[i] > 0 && g->kernel[i] > 0 && g->stride[i] > 0 && g->pad[i] >= 0,
                "convb: Filter/stride dimensions must be nonzero (axis %d)", i);
    const int o = (g->in[i] + 2 * g->pad[i] - g->kernel[i]) / g->stride[i] + 1;
    ECO_REQUIRE(g->in[i] + 2 * g->pad[i] >= g->kernel[i] && o == g->out[i],
                "convb: output dim %d is %d, expected (in+2*pad-kernel)/stride+1 = %d", i, g->out[i], o);
    taps *= g->kernel[i];
  }
  ECO_REQUIRE(taps < 63, "convb: %ld kernel taps exceed the 63-tap validity mask", taps);
  ECO_REQUIRE(g->cout % 8 == 0, "convb: cout = %d is not a multiple of the 8-channel block", g->cout);
  ECO_REQUIRE(is_stem(g) || g->cin % 8 == 0,
              "convb: cin = %d is not a multiple of the 8-channel block (the only other form is the 3-channel 7x7 "
              "stride-2 stem)", g->cin);
  ECO_REQUIRE((long)g->n * g->out[0] * g->out[1] * g->out[2] < 2147483647l, "convb: too many output positions");
  return ECO_OK;
}

extern "C" int eco_convb_plan_create(const eco_conv_geom* g, int32_t dt, int32_t num_cu, eco_convb_plan* plan) {
  clear_error();
  if (int rc = validate_convb_geom(g, dt)) return rc;
  ECO_REQUIRE(plan != nullptr && num_cu >= 0, "convb: bad argument");
  if (num_cu == 0) num_cu = current_device_num_cu();   // as eco_conv_plan_create does
  const int ns = ns_of(dt);
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
  const long ntot = (long)g->n * g->out[0] * g->out[1] * g->out[2];
  plan->bm = bm;
  plan->bn = bm == 128 ? 128 : 256;
  if (ns == 1) {
    // LDS-DMA kernel (bf16): 4x2 wave tiles -- 256 x 128 blocks for the wide layers (res4 / res5), 128 x 256 for
    // 128-channel layers (res3) when the problem has enough positions to fill such tiles
    if (bm == 128 && g->cout % 256 == 0 && ntot >= 4096) { plan->bm = bm = 256; plan->bn = 128; }
    else if (bm == 128 && ntot >= 8192) plan->bn = 256;
  }
  plan->dt = dt;
  plan->span_pieces = 0;
  if (ns == 1 && !is_stem(g)) {
    // stride-1 same-size (kd)x3x3: the span kernel (256-position tiles; the span of BN + 2*(W+1) positions in at
    // most eight 64-position DMA pieces)
    bool span = g->kernel[1] == 3 && g->kernel[2] == 3 && g->pad[1] == 1 && g->pad[2] == 1 &&
                (g->kernel[0] == 1 || g->kernel[0] == 3) && g->pad[0] == g->kernel[0] / 2;
    for (int i = 0; i < 3; ++i) span = span && g->stride[i] == 1 && g->out[i] == g->in[i];
    const int pieces = (int)ceil_div(256 + 2 * (g->in[2] + 1), 64);
    if (span && pieces <= 8 && ntot >= 2048) {
      plan->span_pieces = pieces;
      if (bm == 256) plan->bm = bm = 128;   // the weight stream per MFMA depends on BN only: keep 256 positions
      plan->bn = 256;
    }
  }
  // persistent span kernel: two workgroups per CU, the grid a multiple of 8 XCDs x the M-blocks of a position tile
  plan->pgrid = 0;
  if (plan->span_pieces) {
    const int unit = 8 * (int)ceil_div(g->cout, bm);
    plan->pgrid = (int)(2L * num_cu / unit) * unit;
    if (plan->pgrid < unit) plan->pgrid = unit;
  }
  plan->tail_tiles = 0;
  plan->tail_ksplit = 1;
  plan->stem = is_stem(g) ? 1 : 0;
  plan->cblocks = plan->stem ? 4 : g->cin / 8;
  const int taps = plan->stem ? 7 : g->kernel[0] * g->kernel[1] * g->kernel[2];
  plan->nstages = (int)ceil_div(plan->cblocks, kCbs) * taps;   // a partial last channel group is zero-padded
  // whole 64-row pieces are staged per M-block (LDS-DMA moves 64 lanes x 16 bytes): pad the rows accordingly
  plan->mpad = (int)((ceil_div(g->cout, bm) - 1) * bm + ceil_div(bm, 64) * 64);
  plan->wp_vecs = (int64_t)ns * plan->nstages * kCbs * plan->mpad;
  // split-K: with fewer tiles than resident workgroup slots cut the reduction so that tiles * slices fills them
  // (slices of >= 4 stages; partial sums cost one fp32 write + read of the outputs per slice)
  plan->ksplit = 1;
  plan->ws_bytes = 0;
  const long tiles = ceil_div(g->cout, bm) * ceil_div(ntot, plan->bn);
  const long slots = 2L * num_cu;
  if (tiles < slots) {
    long sp = slots / tiles;
    if (sp > 8) sp = 8;
    if (sp > plan->nstages / 4) sp = plan->nstages / 4;
    if (plan->span_pieces && sp > plan->nstages / 9) sp = plan->nstages / 9;   // the span kernel splits whole groups
    if (plan->pgrid > 0 && sp >= 2) {
      // persistent form: the slots walk ceil(tiles * s / slots) rounds of items, an item = ceil(groups / s) groups of nine
      // taps + ~14 taps' worth of set-up and partial-sum stores, the reduce launch ~3 taps per slice.  res5 (196 tiles, 48
      // groups, 512 slots): s = 2 -> 1 round x 230, s = 5 -> 2 x 104 (measured 0.199 -> 0.188 ms; 0.21 at 4, 0.19 at 6)
      const long groups = plan->nstages / 9, smax = groups < 8 ? groups : 8;
      long best = sp, best_cost = -1;
      for (long c = 2; c <= smax; ++c) {
        const long cost = ceil_div(tiles * c, (long)plan->pgrid) * (ceil_div(groups, c) * 9 + 14) + 3 * c;
        if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
      }
      sp = best;
    }
    if (sp >= 2) {
      plan->ksplit = (int)sp;
      plan->ws_bytes = (int64_t)sp * tiles * bm * plan->bn * 4;   // whole tiles: the fragment layout of the partial sums
    }
  }
  // K-split tail of the persistent span kernel: with more tiles than CUs and a remainder r = tiles mod CUs, r CUs would
  // run one whole item more than the others -- res4: 784 tiles on 256 CUs, 16 of them four items instead of three, a
  // quarter of the launch with 240 CUs idle.  The last r tiles are cut into kb slices of their channel-group range so
  // that r * kb <= CUs: one short extra round for everybody, partial sums for r tiles only.
  if (plan->pgrid > 0 && plan->ksplit == 1) {
    const long cus = plan->pgrid / 2, r = tiles % cus;
    const long groups = plan->nstages / 9;
    long kb = r > 0 ? cus / r : 0;
    if (kb > 8) kb = 8;
    if (kb > groups) kb = groups;
    // worth it on long reductions only: the tail costs a reduce launch and a round trip of its partial sums -- measured
    // +12-15 % on the 2-3-group inception 3x3s (64 tail tiles of an 18-27-tap item), -15 % on res4, -2 % on res3
    const long rounds = ceil_div(tiles, cus);
    const bool pays = groups >= 12 && (kb - 1) * 20 >= kb * rounds;      // predicted gain (1 - 1/kb) / rounds >= 5 %
    if (tiles > cus && r > 0 && kb >= 2 && pays && r % ceil_div(g->cout, bm) == 0) {
      plan->tail_tiles = (int)r;
      plan->tail_ksplit = (int)kb;
      plan->ws_bytes = (int64_t)kb * r * bm * plan->bn * 4;
    }
  }
#ifdef ECO_CONVB_KSPLIT_ENV   // experiment builds only (tools/exp): the K-split tail as "tiles,slices" from the environment
  if (plan->span_pieces && plan->pgrid > 0 && plan->ksplit == 1 && getenv("ECO_CONVB_TAIL")) {
    long r = 0, kb = 1;
    if (sscanf(getenv("ECO_CONVB_TAIL"), "%ld,%ld", &r, &kb) == 2 && r > 0 && r < tiles && kb >= 2 && kb <= 8 &&
        kb <= plan->nstages / 9 && r % ceil_div(g->cout, bm) == 0 && tiles > plan->pgrid) {
      plan->tail_tiles = (int)r;
      plan->tail_ksplit = (int)kb;
      plan->ws_bytes = (int64_t)kb * r * bm * plan->bn * 4;
    }
  }
#endif
#ifdef ECO_CONVB_KSPLIT_ENV   // experiment builds only (tools/exp): split-K factor of span plans from the environment
  if (plan->span_pieces && getenv("ECO_CONVB_KSPLIT")) {
    long sp = atol(getenv("ECO_CONVB_KSPLIT"));
    const long ng = plan->nstages / 9;
    if (tiles < 2 * slots && sp >= 1 && sp <= ng) {
      plan->ksplit = (int)sp;
      plan->ws_bytes = sp > 1 ? (int64_t)sp * tiles * bm * plan->bn * 4 : 0;
    }
  }
#endif
  return ECO_OK;
}

extern "C" int eco_convb_pack_weights(const eco_conv_geom* g, const eco_convb_plan* plan, const float* w, void* wp) {
  clear_error();
  ECO_REQUIRE(plan && w && wp, "convb pack: null argument");
  if (int rc = validate_convb_geom(g, plan->dt)) return rc;
  uint16_t* out = (uint16_t*)wp;   // [nstages][kCbs][mpad][8]
  memset(out, 0, (size_t)plan->wp_vecs * 16);
  const int taps_full = g->kernel[0] * g->kernel[1] * g->kernel[2];
  const long K = (long)g->cin * taps_full;
  auto put = [&](int stage, int row, int m, int e, float v) {
    out[((((long)stage) * kCbs + row) * plan->mpad + m) * 8 + e] = host_bf16(v);
  };
  if (plan->stem) {
    // stage = kernel row ky; block j, element e <-> (kx, c) = (2*j + e/4, e%4); kx = 7 and c = 3 do not exist
    for (int m = 0; m < g->cout; ++m)
      for (int c = 0; c < 3; ++c)
        for (int ky = 0; ky < 7; ++ky)
          for (int kx = 0; kx < 7; ++kx)
            put(ky, kx / 2, m, 4 * (kx % 2) + c, w[(long)m * K + ((long)c * 7 + ky) * 7 + kx]);
    return ECO_OK;
  }
  // stage = cg*taps + tap; row r, element e <-> channel (cg*4 + r)*8 + e
  for (int m = 0; m < g->cout; ++m)
    for (int c = 0; c < g->cin; ++c) {
      const int cb = c / 8, e = c % 8, cg = cb / kCbs, row = cb % kCbs;
      for (int tap = 0; tap < taps_full; ++tap)
        put(cg * taps_full + tap, row, m, e, w[(long)m * K + (long)c * taps_full + tap]);
    }
  return ECO_OK;
}

// 16 bytes of zeros in device memory, the DMA source of out-of-image taps: a zero-initialised __device__ variable of
// this code object (eco::g_zero_page) -- present on every device the library is loaded on, no allocation, no memset, no
// synchronisation at launch time (round 2 allocated a page per thread and device inside eco_convb_forward, which broke
// the header's "entry points never allocate or synchronise" contract and could not run under stream capture).  The
// emulator build hands the kernels a registered host page instead.
static const uint4* device_zero_page() {
#ifdef ECO_EMU
  alignas(16) static const uint4 z = {0u, 0u, 0u, 0u};
  emu_register_buffer(&z, sizeof(z));   // the test harness clears its registry between tests
  return &z;
#else
  return nullptr;                       // the kernels use the device symbol
#endif
}

template <int TM, int TN, int WM, int WN>
static int launch_convb_dma(const ConvBArgs& a0, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BMP = (BM + 63) / 64 * 64;
  const uint4* zp = device_zero_page();
  ConvBArgs a = a0;
  {   // descriptor-addressed DMA where the 32-bit offsets reach: the input (plus the padding's reach in front of it) and the
      // packed weights below 2 GB; ECO_CONVB_DMA_BUF=0 keeps the flat form (A/B runs)
    static const int on = [] { const char* e = getenv("ECO_CONVB_DMA_BUF"); return (e && e[0] == '0') ? 0 : 1; }();
    const long x_bytes = (long)a.ntot / a.s_out * a.img_stride_in * 16;
    const long bias = (((long)a.pd * a.Hi + a.ph) * a.Wi + a.pw) * 16;
    const long wp_bytes = (long)a.nstages * kCbs * a.mpad * 16;
    const long lim = (1l << 31) - (1l << 20);
    a.dma_buf = (on && x_bytes + bias < lim && wp_bytes < lim) ? 1 : 0;
    a.dma_x_bytes = a.dma_buf ? (unsigned)x_bytes : 0u;
    a.dma_wp_bytes = a.dma_buf ? (unsigned)wp_bytes : 0u;
    a.dma_bias = a.dma_buf ? (unsigned)bias : 0u;
  }
  ECO_REQUIRE((((uintptr_t)a.x | (uintptr_t)a.wp) & 15) == 0, "convb: input and packed weights must be 16-byte aligned");
  const int grid = a.nblk_m * a.nblk_n * a.ksplit;
  static_assert(3 * kCbs * (BMP + BN) * 16 <= 160 * 1024, "three stage buffers must fit the CU's LDS");
  hipLaunchKernelGGL((convb_dma_kernel<TM, TN, WM, WN>), dim3(grid), dim3(256), 0, stream, a, zp);
  return check_launch("eco_convb_forward");
}

// Work counters of the persistent kernel's dynamic item distribution: 256 slots of 8 counters, zero at module load; a launch's
// last workgroup clears its slot again.  Eager launches take ONE slot per stream (launches of a stream run in order; up to 64
// streams, then static shares), launches recorded by a stream capture take slots 64 .. 255 in turn and the graph keeps them
// (round-4 advisor: a sequence number modulo 256 let a replaying graph and eager launches on another stream meet in one slot).
// The bound that remains is stated in include/eco_hip.h ("work counters"); eco_counters_reset() clears both arrays.
#ifdef ECO_EMU
static unsigned eco_spanp_counters[256 * 8];
static unsigned* spanp_counter_base() { return eco_spanp_counters; }
#else
__device__ unsigned eco_spanp_counters[256 * 8];
static unsigned* spanp_counter_base() {
  static unsigned* base[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(eco_spanp_counters)) != hipSuccess) return nullptr;
    base[dev] = (unsigned*)p;
  }
  return base[dev];
}
#endif
static unsigned* spanp_counter_slot(void* stream) {
  unsigned* base = spanp_counter_base();
  const int slot = counter_slot_index(stream);       // one per stream, or the capture ring; -1: static shares
  return (base && slot >= 0) ? base + 8 * slot : nullptr;
}
namespace eco {
int spanp_counters_reset(void* stream) {
  unsigned* base = spanp_counter_base();
  if (!base) return fail(ECO_ERR_RUNTIME, "counters_reset: no device");
#ifdef ECO_EMU
  (void)stream;
  memset(base, 0, sizeof(unsigned) * 256 * 8);
#else
  if (hipMemsetAsync(base, 0, sizeof(unsigned) * 256 * 8, (hipStream_t)stream) != hipSuccess)
    return fail(ECO_ERR_RUNTIME, "counters_reset: hipMemsetAsync failed");
#endif
  return ECO_OK;
}
}  // namespace eco
static bool spanp_dynamic() {
  static const int on = [] { const char* e = getenv("ECO_SPANP_DYNAMIC"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}

// The persistent form (convb_spanp_kernel): 2 workgroups per CU walk the (slice, tile) items.
static bool spanp_enabled() {
  static const int on = [] { const char* e = getenv("ECO_SPANP"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}
template <int TM>
static int launch_convb_spanp(const ConvBArgs& a, const eco_convb_plan* plan, hipStream_t stream) {
  constexpr int BM = 32 * TM, BMP = (BM + 63) / 64 * 64;
  ECO_REQUIRE((((uintptr_t)a.x | (uintptr_t)a.wp) & 15) == 0, "convb: input and packed weights must be 16-byte aligned");
  SpanPArgs pa;
  pa.x_bytes = (unsigned)((long)a.ntot / a.s_out * a.img_stride_in * 16);
  pa.wp_bytes = (unsigned)(plan->wp_vecs * 16);
  pa.ntiles = a.nblk_m * a.nblk_n;
  // whole-tensor split: every tile in ksplit slices; K-split tail: the plan's last tail_tiles in tail_ksplit slices
  pa.t_tail = a.ksplit > 1 ? pa.ntiles : a.ws_slices > 1 ? plan->tail_tiles : 0;
  pa.kb = a.ksplit > 1 ? a.ksplit : a.ws_slices > 1 ? plan->tail_ksplit : 1;
  pa.t_main = pa.ntiles - pa.t_tail;
  pa.nitems = pa.t_main + pa.t_tail * pa.kb;
  // dynamic items where a draw (~600 cycles of one wave, once per item) is small against what it balances: long items (the
  // trunk: >= 54 taps) or many per workgroup (conv2_3x3: 49); the 18-27-tap inception launches (six items per workgroup, no
  // measurable difference between the two workgroups of a CU) keep their static shares
  {
    const long groups_per_item = (long)(a.nstages / a.taps) * a.kd / (pa.kb > 1 && pa.t_main == 0 ? pa.kb : 1);
    const bool worth = groups_per_item * 9 >= 54 || (long)pa.nitems >= 16l * plan->pgrid;
    pa.ctr = (spanp_dynamic() && worth && a.nblk_m <= 4 && plan->pgrid % a.nblk_m == 0) ? spanp_counter_slot((void*)stream) : nullptr;
  }
  pa.d_sout = fastdiv_make((unsigned)a.s_out);
  pa.d_hw = fastdiv_make((unsigned)(a.Hi * a.Wi));
  pa.d_w = fastdiv_make((unsigned)a.Wi);
  pa.d_ks = fastdiv_make((unsigned)pa.kb);
  pa.d_kd = fastdiv_make((unsigned)a.kd);
  pa.d_tail = fastdiv_make((unsigned)(pa.t_tail > 0 ? pa.t_tail : 1));
  const size_t lds = (size_t)(3 * kCbs * BMP + 2 * kCbs * 384) * 16;
  if (lds > 64 * 1024) ECO_RAISE_DYNAMIC_LDS((convb_spanp_kernel<TM>), "convb");
  hipLaunchKernelGGL((convb_spanp_kernel<TM>), dim3(plan->pgrid), dim3(256), lds, stream, a, pa, plan->span_pieces);
  return check_launch("eco_convb_forward");
}

extern "C" int eco_convb_forward(const eco_conv_geom* g, const eco_convb_plan* plan, const void* x, const void* wp,
                                 const eco_conv_epilogue* ep, void* workspace, void* stream) {
  clear_error();
  ECO_REQUIRE(plan && x && wp && ep, "convb: null argument");
  if (int rc = validate_convb_geom(g, plan->dt)) return rc;
  ECO_REQUIRE(ep->raw.ptr || ep->act.ptr, "convb: at least one of raw/act outputs is required");
  ECO_REQUIRE(!ep->bn_scale == !ep->bn_shift, "convb: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(!ep->act2.ptr || ep->act.ptr, "convb: act2 needs act");
  if (ep->nseg) {
    ECO_REQUIRE(ep->nseg >= 1 && ep->nseg <= ECO_MAX_SEG, "convb: 1 to %d extra output segments (got %d)", ECO_MAX_SEG, ep->nseg);
    ECO_REQUIRE(ep->act.ptr && ep->act.t == 1 && !ep->raw.ptr && !ep->residual.ptr && !ep->act2.ptr,
                "convb: a segmented launch takes plain act destinations only (no raw / residual / act2)");
    int prev = 0;
    for (int s = 0; s < ep->nseg; ++s) {
      ECO_REQUIRE(ep->seg_act[s].ptr && ep->seg_act[s].t == 1 && ep->seg_act[s].stride_c >= 1,
                  "convb: segment %d needs a plain destination view", s + 1);
      ECO_REQUIRE(ep->seg_begin[s] > prev && ep->seg_begin[s] < g->cout && ep->seg_begin[s] % 32 == 0,
                  "convb: segment boundary %d must be a multiple of 32 inside (%d, %d)", ep->seg_begin[s], prev, g->cout);
      prev = ep->seg_begin[s];
    }
  }
  const eco_view* views[4] = {&ep->residual, &ep->raw, &ep->act, &ep->act2};
  for (const eco_view* v : views)
    ECO_REQUIRE(!v->ptr || (v->t >= 1 && v->stride_c >= 1), "convb: view needs t >= 1 and stride_c >= 1");
  const int ns = ns_of(plan->dt);
  ConvBArgs a;
  a.x = x; a.wp = (const uint4*)wp;
  a.bias = ep->bias; a.bn_scale = ep->bn_scale; a.bn_shift = ep->bn_shift;
  a.residual = ep->residual; a.raw = ep->raw; a.act = ep->act; a.act2 = ep->act2; a.relu = ep->relu;
  a.nseg = ep->nseg;
  for (int s = 0; s < ECO_MAX_SEG; ++s) {
    a.seg_begin[s] = s < ep->nseg ? ep->seg_begin[s] : 0;
    a.seg_relu[s] = s < ep->nseg ? ep->seg_relu[s] : 0;
    a.seg_act[s] = s < ep->nseg ? ep->seg_act[s] : eco_view{nullptr, 0, 0, 0, 1};
  }
  a.cout = g->cout; a.mpad = plan->mpad; a.nstages = plan->nstages; a.cblocks = plan->cblocks;
  a.Do = g->out[0]; a.Ho = g->out[1]; a.Wo = g->out[2];
  if (plan->stem) {
    // the packed image of eco_stem_pack_forward: rows of (W+8)/2 blocks, H+6 rows; kernel rows are the taps
    const int H = g->in[1], W = g->in[2];
    ECO_REQUIRE(plan->cblocks == 4 && plan->nstages == 7, "convb: stem plan does not match geometry");
    a.Di = 1; a.Hi = H + 6; a.Wi = (W + 8) / 2;
    a.kd = 1; a.kh = 7; a.kw = 1; a.sd = 1; a.sh = 2; a.sw = 1; a.pd = 0; a.ph = 0; a.pw = 0;
    a.taps = 7;
    a.img_stride_in = (long)a.Hi * a.Wi;
    a.cb_stride_in = 1;
    ECO_REQUIRE(g->out[2] + 3 <= a.Wi && 2 * (g->out[1] - 1) + 7 <= a.Hi, "convb: stem geometry out of the packed image");
  } else {
    ECO_REQUIRE(plan->cblocks == g->cin / 8 &&
                    plan->nstages == (int)ceil_div(plan->cblocks, kCbs) * g->kernel[0] * g->kernel[1] * g->kernel[2],
                "convb: plan does not match geometry");
    a.Di = g->in[0]; a.Hi = g->in[1]; a.Wi = g->in[2];
    a.kd = g->kernel[0]; a.kh = g->kernel[1]; a.kw = g->kernel[2];
    a.sd = g->stride[0]; a.sh = g->stride[1]; a.sw = g->stride[2];
    a.pd = g->pad[0]; a.ph = g->pad[1]; a.pw = g->pad[2];
    a.taps = a.kd * a.kh * a.kw;
    a.cb_stride_in = (long)a.Di * a.Hi * a.Wi;
    a.img_stride_in = (long)plan->cblocks * a.cb_stride_in;
  }
  a.s_out = a.Do * a.Ho * a.Wo;
  a.ntot = g->n * a.s_out;
  a.nblk_m = (int)ceil_div(g->cout, plan->bm);
  a.nblk_n = (int)ceil_div(a.ntot, plan->bn);
  ECO_REQUIRE((long)(a.nblk_m - 1) * plan->bm + (plan->bm + 63) / 64 * 64 <= plan->mpad, "convb: plan mpad too small");
  ECO_REQUIRE(plan->ksplit >= 1 && plan->ksplit <= plan->nstages, "convb: bad split-K factor %d", plan->ksplit);
  ECO_REQUIRE(plan->ksplit == 1 || (workspace && (int64_t)plan->ksplit * g->cout * a.ntot * 4 <= plan->ws_bytes),
              "convb: plan needs a %ld-byte workspace", (long)plan->ws_bytes);
  a.ksplit = plan->ksplit;
  a.ws = (float*)workspace;
  a.ws_n0 = 0; a.ws_pitch = a.ntot; a.ws_slices = plan->ksplit;
  a.ws_frag = 0; a.ws_tile0 = 0; a.ws_ntl = a.nblk_m * a.nblk_n;
  // (the 16-byte-store epilogue addresses every destination as descriptor base + 32-bit offset: views must end below 2 GB)
  auto view_fits = [&](const eco_view& v) {
    if (!v.ptr) return true;
    const long nb = (g->n - 1) / v.t;
    return (nb * v.stride_b + (long)(v.t - 1) * v.stride_t + (long)(g->cout / 8) * v.stride_c + a.s_out) * 16 < (1l << 31) - (1l << 20);
  };
  bool views_fit = view_fits(ep->residual) && view_fits(ep->raw) && view_fits(ep->act) && view_fits(ep->act2);
  for (int sgi = 0; sgi < ep->nseg; ++sgi) views_fit = views_fit && view_fits(ep->seg_act[sgi]);
  a.wide = (ns == 1 && views_fit) ? 1 : 0;
  {   // one destination per 32-row tile, bias + BN (+ ReLU per segment), nothing else: the lean epilogue
    a.lean = (a.wide && a.act.ptr && !a.raw.ptr && !a.residual.ptr && !a.act2.ptr) ? 1 : 0;
  }
  a.d_sout = fastdiv_make((unsigned)a.s_out);
  // whole-tensor split: the fragment layout when the epilogue it ends in is the 16-byte one, every instance of the
  // launched kernel has 64-position wave tiles (all bf16 ones) and the padded workspace fits plan and descriptor
  const int64_t frag_bytes = (int64_t)plan->ksplit * a.nblk_m * a.nblk_n * plan->bm * plan->bn * 4;
  if (plan->ksplit > 1 && plan->ksplit <= 8 && a.wide && !plan->stem && frag_bytes <= plan->ws_bytes && frag_bytes < (1l << 31) - (1l << 20))
    a.ws_frag = 1;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (plan->span_pieces) {
    bool ok = ns == 1 && !plan->stem && plan->bn == 256 && g->kernel[1] == 3 && g->kernel[2] == 3 && g->pad[1] == 1 &&
              g->pad[2] == 1 && (g->kernel[0] == 1 || g->kernel[0] == 3) && g->pad[0] == g->kernel[0] / 2 &&
              plan->span_pieces * 64 >= 256 + 2 * (g->in[2] + 1) && plan->span_pieces <= 8 &&
              plan->ksplit <= (plan->nstages / a.taps) * a.kd;
    for (int i = 0; i < 3; ++i) ok = ok && g->stride[i] == 1 && g->out[i] == g->in[i];
    ECO_REQUIRE(ok, "convb: the span kernel needs a bf16 stride-1 same-size (kd)x3x3 geometry");
    // persistent form: descriptor-addressed DMA needs both operands below 2 GB, a span of at most 6 pieces (its LDS pitch)
    // and a grid the plan sized for the device (a plan of an older header has pgrid 0)
    const long x_bytes = (long)g->n * a.img_stride_in * 16, wp_bytes = plan->wp_vecs * 16;
    const bool persistent = spanp_enabled() && views_fit && plan->pgrid >= 8 * a.nblk_m && plan->pgrid % (8 * a.nblk_m) == 0 &&
                            plan->span_pieces <= 6 && x_bytes < (1l << 31) - (1l << 20) && wp_bytes < (1l << 31) - (1l << 20) &&
                            plan->ksplit * ((plan->nstages / a.taps) * a.kd) < (1 << 20);
    // K-split tail (persistent form only; the per-tile kernel runs such a plan unsplit): the reduce launch covers the
    // tail's positions
    const bool tail = persistent && plan->ksplit == 1 && plan->tail_tiles > 0 && plan->tail_ksplit > 1;
    if (tail) {
      ECO_REQUIRE(plan->tail_tiles % a.nblk_m == 0 && plan->tail_tiles < a.nblk_m * a.nblk_n &&
                      plan->tail_ksplit <= (plan->nstages / a.taps) * a.kd,
                  "convb: bad K-split tail (%d tiles x %d)", plan->tail_tiles, plan->tail_ksplit);
      a.ws_n0 = (a.nblk_n - plan->tail_tiles / a.nblk_m) * plan->bn;
      a.ws_pitch = a.ntot - a.ws_n0;
      a.ws_slices = plan->tail_ksplit;
      a.ws_tile0 = a.nblk_m * a.nblk_n - plan->tail_tiles;
      a.ws_ntl = plan->tail_tiles;
      const int64_t tfrag = (int64_t)plan->tail_ksplit * plan->tail_tiles * plan->bm * plan->bn * 4;
      a.ws_frag = (a.wide && plan->tail_ksplit <= 8 && tfrag <= plan->ws_bytes && tfrag < (1l << 31) - (1l << 20)) ? 1 : 0;
      ECO_REQUIRE(workspace && (a.ws_frag || (int64_t)a.ws_slices * g->cout * a.ws_pitch * 4 <= plan->ws_bytes),
                  "convb: plan needs a %ld-byte workspace", (long)plan->ws_bytes);
    }
    if (persistent) {
      if (a.nseg > 0) a.lean = 0;   // (its lean epilogue is the segment-free instance)
      switch (plan->bm) {
        case 128: rc = launch_convb_spanp<4>(a, plan, s); break;
        case 96: rc = launch_convb_spanp<3>(a, plan, s); break;
        case 64: rc = launch_convb_spanp<2>(a, plan, s); break;
        case 32: rc = launch_convb_spanp<1>(a, plan, s); break;
        default: return fail(ECO_ERR_INVALID, "convb: unsupported span tile bm=%d", plan->bm);
      }
    } else {
      // fallback (operands or views that do not end below 2 GB, spans of more than six pieces, ECO_SPANP=0): the per-tap LDS-DMA
      // kernel on the same packed weights and the same 256-position tiles (round 6: the per-tile span kernel of round 3, the
      // third implementation of these layers, is gone; tools/exp keeps nothing of it either)
      switch (plan->bm) {
        case 128: rc = launch_convb_dma<4, 2, 1, 4>(a, s); break;
        case 96: rc = launch_convb_dma<3, 2, 1, 4>(a, s); break;
        case 64: rc = launch_convb_dma<2, 2, 1, 4>(a, s); break;
        case 32: rc = launch_convb_dma<1, 2, 1, 4>(a, s); break;
        default: return fail(ECO_ERR_INVALID, "convb: unsupported span tile bm=%d", plan->bm);
      }
    }
  } else
  switch (plan->bm) {
    case 256:
      ECO_REQUIRE(plan->bn == 128 && ns == 1, "convb: bad plan");
      rc = launch_convb_dma<4, 2, 2, 2>(a, s);
      break;
    case 128:
      ECO_REQUIRE(plan->bn == 128 || plan->bn == 256, "convb: bad plan");
      rc = plan->bn == 256 ? launch_convb_dma<4, 2, 1, 4>(a, s) : launch_convb_dma<2, 2, 2, 2>(a, s);
      break;
    case 96: ECO_REQUIRE(plan->bn == 256, "convb: bad plan"); rc = launch_convb_dma<3, 2, 1, 4>(a, s); break;
    case 64: ECO_REQUIRE(plan->bn == 256, "convb: bad plan"); rc = launch_convb_dma<2, 2, 1, 4>(a, s); break;
    case 32: ECO_REQUIRE(plan->bn == 256, "convb: bad plan"); rc = launch_convb_dma<1, 2, 1, 4>(a, s); break;
    default: return fail(ECO_ERR_INVALID, "convb: unsupported block tile bm=%d", plan->bm);
  }
  if (rc != ECO_OK || a.ws_slices == 1) return rc;
  if (a.ws_frag) {
    const int wn = plan->bn / 64;   // waves side by side in N of the producing kernel (4 or 2)
    const int tm = plan->bm / 32 / (4 / wn);
#define ECO_FRAG_REDUCE(TMv, WNv) hipLaunchKernelGGL((convb_splitk_reduce_frag_kernel<TMv, WNv>), dim3(a.ws_ntl * TMv), dim3(256), 0, s, a)
    if (wn == 4 && tm == 4) ECO_FRAG_REDUCE(4, 4);
    else if (wn == 4 && tm == 3) ECO_FRAG_REDUCE(3, 4);
    else if (wn == 4 && tm == 2) ECO_FRAG_REDUCE(2, 4);
    else if (wn == 4 && tm == 1) ECO_FRAG_REDUCE(1, 4);
    else if (wn == 2 && tm == 4) ECO_FRAG_REDUCE(4, 2);
    else if (wn == 2 && tm == 2) ECO_FRAG_REDUCE(2, 2);
    else return fail(ECO_ERR_INVALID, "convb: no fragment-layout reduce for a %d x %d tile", plan->bm, plan->bn);
#undef ECO_FRAG_REDUCE
    return check_launch("eco_convb_forward(split-K reduce)");
  }
  const int rgrid = grid_for_b((long)(a.cout / 8) * a.ws_pitch);
  hipLaunchKernelGGL((convb_splitk_reduce_kernel), dim3(rgrid), dim3(256), 0, s, a);
  return check_launch("eco_convb_forward(split-K reduce)");
}

extern "C" int eco_stem_pack_forward(const float* x, void* y, int64_t frames, int32_t h, int32_t w, int32_t dt,
                                     void* stream) {
  clear_error();
  ECO_REQUIRE(x && y && frames > 0 && h > 0 && w > 0 && w % 2 == 0, "stem pack: bad argument (W must be even)");
  const int ns = ns_of(dt);
  ECO_REQUIRE(ns != 0, "stem pack: storage type must be ECO_DT_BF16");
  const long total = (long)frames * (h + 6) * ((w + 8) / 2);
  hipLaunchKernelGGL((stem_pack_kernel), dim3(grid_for_b(total)), dim3(256), 0, (hipStream_t)stream, x, y, (long)frames, h, w);
  return check_launch("eco_stem_pack_forward");
}
