// eco_stemb.hip -- the BN-Inception stem of the channel-blocked bf16 path as ONE kernel: conv1_7x7_s2 (3 -> cout, 7x7,
// stride 2, pad 3) + bias + folded BN + ReLU + pool1_3x3_s2 (MAX 3x3, stride 2, Caffe ceil rule), fp32 frames in, pooled
// activations out in the blocked layout Y[n][c/8][h][w][c%8] (bf16).
//
//   conv1_7x7_s2 / conv1_7x7_s2_bn / conv1_relu_7x7 / pool1_3x3_s2   models_ECO_Lite/kinetics/deploy.prototxt:8-77
//   ConvolutionLayer::Forward (conv_layer.cpp:28-43, base_conv_layer.cpp:264-287), BN TEST branch
//   (bn_layer.cpp:93-207), ReLU (relu_layer.cpp:10-20), PoolingLayer MAX (pooling_layer.cpp:131-147,199-237)
//
// Until round 3 the blocked path ran the stem as three launches -- eco_stem_pack_forward (frames -> zero-padded
// pixel-interleaved bf16 image, 0.23 ms for 1024 frames), conv1 as a blocked convolution with 4 "channel blocks" (1.25 ms,
// 0.21 of its floor: K = 7 x 32 for 147 real taps and 64 output channels per 16 KB position stage) and pool1 (0.40 ms) --
// with conv1's 1.64 GB of bf16 output written and read back in between.  This kernel has the structure of the fp32
// stem (eco_stem.hip): persistent workgroups, two per CU, walk 8 x 14 patches of POOLED outputs;
//   * the packed weights (11 k-steps x cout x 16 bf16) stay in LDS for the workgroup's lifetime; the 39 x 64 x 3 input
//     patch a pooled patch depends on is loaded once as fp32 (the next patch's words are in flight in registers under
//     the current patch's work), rounded to bf16 -- the rounding eco_stem_pack_forward applies -- and stored row-major;
//   * k is ordered (c, ky | kx): one MFMA k-step (16) = two kernel rows of 8 taps (kx = 7 has zero weights, as has the
//     22nd row), so the B fragment of v_mfma_f32_32x32x16_bf16 -- eight consecutive k of one position -- is the eight
//     consecutive patch elements starting at column 2q of row (c, 2r + ky): 16 bytes at a 4-byte aligned LDS address
//     (two ds_read2_b32), a per-lane base plus a compile-time offset per k-step, no im2col, no masks;
//   * 17 x 29 conv outputs (one halo row / column) = 512 position columns x cout in 11 x 8 MFMAs per wave -- 2.8 k
//     cycles where the fp32 reduction takes 38 k -- then bias / BN in fp32 on the accumulators, one m-tile (32 channels)
//     at a time through an LDS stage of bf16 PAIRS for the 3 x 3 stride-2 max, ReLU, and one 16-byte store per (pooled
//     position, 8-channel block).  conv1's output never exists in HBM.
// Measured (1024 frames of 224 x 224): 0.69 ms against 1.88 ms for the three launches; probe builds (-DECO_STEMB_PROBE)
// put 0.22 ms on the reduction, 0.19 on the pooling reads and stores, 0.28 on everything else (bias / BN + stage
// writes, patch stores, five barriers per patch).  With an fp32 stage of 16 channels per pass it was 0.80 ms.  Double
// buffering the fragments and issuing the frame loads a whole epilogue earlier changed nothing: the phases are short
// and strictly ordered by the barriers, and two workgroups per CU (250 VGPRs: 128 accumulators) is all that overlaps them.
// Arithmetic: bf16 operands (frames and weights rounded to nearest even), fp32 products and sums, fp32 bias / BN, one
// rounding to bf16 at the store -- the blocked path's convention (eco_blocked.hip).  The MAX window commutes with that
// rounding (monotonic), so the result equals pooling the rounded conv output.
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "eco_common.h"

namespace eco {

constexpr int kSbPH = 8, kSbPW = 14;                       // pooled patch per workgroup
constexpr int kSbCR = 2 * kSbPH + 1, kSbCQ = 2 * kSbPW + 1;   // conv patch: 17 x 29
constexpr int kSbNPos = kSbCR * kSbCQ;                     // 493 conv positions, padded to 512 columns
// Pitch of a conv row in the pooling stage (dwords).  29 would do; 39 makes 2 * pitch = 14 (mod 32): the pooling threads of a
// 32-lane group -- 14 per pooled row, rows 2 * pitch apart -- then read 32 different banks (with 29, rows ph and ph + 1
// shared eight of them: 482 M bank-conflict cycles per six launches, round-4 PMC).  The stage grows from 31 to 42 KB; two
// workgroups still fit a CU (2 x 80.6 KB).
#ifndef ECO_STEMB_PITCH
#define ECO_STEMB_PITCH 39
#endif
constexpr int kSbSP = ECO_STEMB_PITCH;
constexpr int kSbIR = 2 * (kSbCR - 1) + 7;                 // 39 input rows per channel
constexpr int kSbIQ = 64;                                  // stored columns per row: 63 + the zero-weight tap's column
constexpr int kSbRows = 3 * kSbIR;                         // 117 patch rows
constexpr int kSbSteps = 11;                               // 22 (c, ky) rows of 8 taps = 176 k, 16 per MFMA
constexpr int kSbRowBytes = kSbIQ * 2;

// byte offset of kernel row rho = c*7 + ky within the patch, relative to a position's base (row 2r, column 2q)
constexpr int sb_tap(int rho) { return ((rho / 7) * kSbIR + rho % 7) * kSbRowBytes; }
// k-step s: lanes 0-31 take row 2s, lanes 32-63 row 2s + 1 (the 22nd row has zero weights: it re-reads row 20)
constexpr int sb_row1(int s) { return 2 * s + 1 < 21 ? 2 * s + 1 : 20; }
constexpr int sb_delta(int s) { return sb_tap(sb_row1(s)) - sb_tap(2 * s); }
constexpr int kSbNT = 3;                                   // the three half-wave differences: next row, next channel, none
constexpr int sb_type(int s) { return sb_delta(s) == kSbRowBytes ? 0 : sb_delta(s) == (kSbIR - 6) * kSbRowBytes ? 1 : sb_delta(s) == 0 ? 2 : -1; }
constexpr bool sb_types_ok() {
  for (int s = 0; s < kSbSteps; ++s)
    if (sb_type(s) < 0) return false;
  return true;
}
static_assert(sb_types_ok(), "every k-step's half-wave offset difference is one of the three known constants");

struct StemBArgs {
  const float* x;        // [n][3][H][W] fp32 frames
  const uint4* wp;       // [11][2][cout] x 8 bf16: k-step s, half g, channel m -> w[m][rho = 2s + g][kx = 0..7]
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  uint4* y;              // [n][cout/8][PHo][PWo] x 8 bf16
  int n, H, W, cout, Ho, Wo, PHo, PWo;
  int relu, tiles_h, tiles_w;
  int total;             // patches (n * tiles_h * tiles_w); a workgroup takes patch blockIdx.x, + gridDim.x, ...
  unsigned* ctr;         // ... or, given a work counter (ctr[0] tickets, ctr[1] workgroups gone; zero before the launch, zeroed
                         // again by the last workgroup to leave), blockIdx.x and then whatever patch is next when it asks
};

// sixteen bytes from a 4-byte aligned LDS address (a position's columns start at an even bf16 index)
__device__ __forceinline__ uint4 lds_ld16_a4(const unsigned char* p) {
#ifdef ECO_EMU
  uint4 v;
  memcpy(&v, p, 16);
  return v;
#else
  const unsigned* q = (const unsigned*)p;
  return make_uint4(q[0], q[1], q[2], q[3]);
#endif
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { u16x2 r; memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ unsigned from_u16x2(u16x2 v) { unsigned r; memcpy(&r, &v, 4); return r; }

#define ECO_SBTS(slot) do { } while (0)

// TMC = cout / 32 (1 or 2 m-tiles; every wave holds all channels of its 128 columns)
template <int TMC>
__global__ __launch_bounds__(256, 2) void stemb_kernel(const StemBArgs a) {
  ECO_CLOCK("stemb");
  constexpr int COUT = 32 * TMC;
  constexpr int XS_BYTES = kSbRows * kSbRowBytes;          // 14976
  constexpr int W_VECS = kSbSteps * 2 * COUT;              // 16-byte vectors of packed weights
  constexpr int STAGE_LD = kSbCR * kSbSP + 3;              // staging row of one channel pair
  constexpr int XU4 = (kSbRows + 15) / 16, WU = (W_VECS + 255) / 256;
  ECO_DYNAMIC_LDS(lds);
  unsigned char* const Xs = (unsigned char*)lds;           // bf16 [3][39][64]
  uint4* const Ws = (uint4*)(Xs + XS_BYTES);               // [11][2][COUT], resident for the workgroup's lifetime
  float* const Es = (float*)(Ws + W_VECS);                 // [2][COUT]: BN scale, bias * scale + shift
  float* const Ss = Es + 2 * COUT;                         // staging [16][STAGE_LD] fp32

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tpf = a.tiles_h * a.tiles_w;

#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int i = tid + 256 * u;
    if (i < W_VECS) Ws[i] = ld(a.wp + i);
  }
  if (tid < COUT) {
    const float b = a.bias ? ld(a.bias + tid) : 0.0f;
    const float sc = a.bn_scale ? ld(a.bn_scale + tid) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + tid) : 0.0f;
    Es[tid] = sc;
    Es[COUT + tid] = b * sc + sh;
  }

  // A wave takes whole patch rows (lane = column): the row arithmetic is scalar, a lane's address is base + lane,
  // and every load is in flight before the first one is waited for.
  // A patch row is 64 columns: sixteen lanes take four columns each (one 16-byte load, 4-byte aligned: the patch starts
  // three columns left of a multiple of four), a wave four rows per instruction, the workgroup sixteen -- eight loads and
  // eight 8-byte LDS stores per thread and patch.  One column per lane (thirty 4-byte loads and thirty ds_write_b16 per
  // thread) cost ~230 cycles per load instruction wherever it was placed in the patch's schedule -- 6-9 k of a patch's
  // 26 k cycles (round-4 cycle stamps, tools/exp/stemb_ts.py): the address path takes a wave-instruction at a time, 256
  // bytes or 1 KB alike.  The loads go through a buffer descriptor over the frame (no branches: a load inside a divergent
  // branch is waited for on the spot, eight memory latencies in a row): rows outside the image take an out-of-range
  // offset and read zeros, columns outside it are zeroed by a per-patch lane mask (the four columns of a lane straddle
  // the border at column 0 and at column W only); the one load that would start before the frame (channel 0, row 0,
  // columns -3..0) is moved three columns right and its first element handed to column 0.
  uint4 xv[XU4];          // as loaded: the masks below are applied when the patch is stored, a patch period later -- a select
  unsigned xmask = 0u;    // next to the load would wait for the load (bits 0-3: columns inside the image; 4 + u: `before`)
  const int lrow = 4 * wave + (lane >> 4), lcol = 4 * (lane & 15);
  auto load_patch = [&](int patch) {
    const int f = patch / tpf, t = patch - f * tpf;
    const int by = t / a.tiles_w, bx = t - by * a.tiles_w;
    const int ih0 = 4 * kSbPH * by - 3;                     // first input row / column
    const BufRd rx = make_buf_rd(a.x + (long)f * 3 * a.H * a.W, (unsigned)(3 * a.H * a.W) * 4u);
    const int w = 4 * kSbPW * bx - 3 + lcol;
    const bool m0 = (unsigned)w < (unsigned)a.W, m1 = (unsigned)(w + 1) < (unsigned)a.W,
               m2 = (unsigned)(w + 2) < (unsigned)a.W, m3 = (unsigned)(w + 3) < (unsigned)a.W;
    unsigned mk = (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u);
    int lr = lrow;
    ECO_OPAQUE(lr);   // (per patch: hoisted out of the patch loop, the eight rows' channel / row terms were spilled, and a spill
                      // reload between two loads waits for every load before it -- the memory counter retires in order)
#pragma unroll
    for (int u = 0; u < XU4; ++u) {
      const int row = 16 * u + lr;                         // (c, rr)
      const int c = (row >= kSbIR) + (row >= 2 * kSbIR), h = ih0 + row - c * kSbIR;
      const bool rok = row < kSbRows && (unsigned)h < (unsigned)a.H;
      const int e = (c * a.H + h) * a.W + w;               // element of the frame (one frame is below 2^29 elements)
      const bool before = rok && e < 0;                    // (c = 0, h = 0, w = -3: only column 0 exists)
      const unsigned voff = !rok ? kBufOob : (unsigned)(before ? e + 3 : e) * 4u;
      mk |= before ? 16u << u : 0u;
      xv[u] = gld16_buf(rx, voff);
    }
    xmask = mk;
  };
  auto store_patch = [&]() {
    uint2* const xs = (uint2*)Xs;                          // a row = sixteen 8-byte units
    const unsigned mk = xmask;
#pragma unroll
    for (int u = 0; u < XU4; ++u)
      if (16 * u + lrow < kSbRows) {
        const uint4 q = xv[u];
        const float v0 = (mk & 1u) ? __builtin_bit_cast(float, q.x) : 0.0f, v1 = (mk & 2u) ? __builtin_bit_cast(float, q.y) : 0.0f,
                    v2 = (mk & 4u) ? __builtin_bit_cast(float, q.z) : 0.0f,
                    v3 = (mk & 8u) ? __builtin_bit_cast(float, (mk & (16u << u)) ? q.x : q.w) : 0.0f;
        xs[(16 * u + lrow) * (kSbIQ / 4) + (lane & 15)] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
  };

  // ---- this lane's four position columns: conv position p = wave*128 + j*32 + l31 -> patch byte 2r*128 + 4q ----
  int pbase[4], soff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int p = wave * 128 + j * 32 + l31;
    if (p >= kSbNPos) p = kSbNPos - 1;                     // padding columns: any valid position (never stored)
    const int r = p / kSbCQ, q = p - r * kSbCQ;
    pbase[j] = 2 * r * kSbRowBytes + 4 * q;
    // the pooling stage keeps a conv row's 29 columns parity split (15 even, then 14 odd): the 3x3 stride-2 windows of
    // consecutive pooled columns read consecutive words
    soff[j] = r * kSbSP + (q & 1) * ((kSbSP + 1) / 2) + (q >> 1);
  }
  const bool last_col_ok = wave * 128 + 96 + l31 < kSbNPos;   // only wave 3's j = 3 has padding columns
  const uint4* const wl = Ws + half * COUT + l31;          // A[m = l31 (+32 i)][k = 16 s + 8 half + e]
  // The stage holds one m-tile (32 channels) at a time as bf16 PAIRS -- row = channel pair, one dword per position --:
  // half the LDS traffic of an fp32 stage on both sides, and one pass per m-tile instead of two.  The values are
  // already rounded as the store would round them; MAX and ReLU commute with that rounding.
  // pooling threads: 14 x 8 pooled positions x 2 = 224 of the 256; thread (pos, clo) takes the 8-channel blocks clo, clo + 2
  const int pw = tid % kSbPW, ph = (tid / kSbPW) % kSbPH, clo = tid / (kSbPW * kSbPH);
  unsigned* const Su = (unsigned*)Ss;
  const unsigned* const sp0 = Su + 4 * clo * STAGE_LD + 2 * ph * kSbSP + pw;
  unsigned* const sw0 = Su + 2 * half * STAGE_LD;
  const float relu_floor = a.relu ? 0.0f : -FLT_MAX;
  const float stage_floor = a.relu ? 0.0f : -__builtin_inff();      // ReLU ahead of the stage (see the pooling loop)
  const unsigned stage_mask = a.relu ? 0x7fff7fffu : 0xffffffffu;   // (max(-0, +0) may be -0: its sign bit would win an unsigned maximum)
  const int cblocks = a.cout / 8;

  // Patches: the first is blockIdx.x; after it either every gridDim.x-th, or -- dynamic -- the next one nobody has taken
  // (patch gridDim.x + ticket), drawn by wave 0 on the scalar unit when the patch's loads are about to be issued (two patches
  // ahead of its reduction) and passed through LDS behind a barrier the loop has anyway.  The two workgroups of a CU do not
  // run at the same speed (the instruction arbiter prefers the older wave): with equal shares block b finished its 56
  // patches in 421 us and block b + 256 in 510 us (tools/exp/clock_probe2.sh).
  const bool dynamic = a.ctr != nullptr;
  __shared__ int next_patch;
  auto draw = [&]() {      // wave 0; the value is read behind the next barrier
    if (dynamic && wave == 0) {
#ifdef ECO_EMU
      if (tid == 0)
#endif
      {
        const unsigned t = counter_draw_wave(a.ctr);
        if (lane == 0) next_patch = (int)gridDim.x + (int)t;
      }
    }
  };
  int patch = (int)blockIdx.x;
  int next = patch + (int)gridDim.x;
  if (dynamic) {
    draw();
    __syncthreads();
    next = uniform(next_patch);
    __syncthreads();       // (read before the next draw overwrites it)
  }
  load_patch(patch);
  store_patch();
  if (next < a.total) load_patch(next);
#if !defined(ECO_EMU) && !defined(ECO_STEMB_NOSTAGGER)
  // Workgroups that start together stay in step: every CU fetched its next patch in the same ~7 k cycles of a 24 k-cycle
  // patch period -- 16 MB asked of HBM at once, every wave waiting at its load instructions, the memory idle for the rest
  // (round-4 cycle stamps).  The eight XCDs start their patch loops an eighth of a period apart instead.
  for (int q = (int)(blockIdx.x & 7u); q > 0; --q) __builtin_amdgcn_s_sleep(47);   // ~3 k cycles a step
#endif
  __syncthreads();

  while (true) {
    // (the words of `next` are in flight, or landed, in xv)
    ECO_SBTS(0);

    f32x16 acc[TMC][4];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses: base register of the k-step's type + compile-time immediate; opaque per patch so that the
    // compiler does not materialise all 11 x 4 addresses outside the patch loop
    int pb[kSbNT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pb[0][j] = pbase[j] + (half ? kSbRowBytes : 0);
      pb[1][j] = pbase[j] + (half ? (kSbIR - 6) * kSbRowBytes : 0);
      pb[2][j] = pbase[j];
#pragma unroll
      for (int t = 0; t < kSbNT; ++t) ECO_OPAQUE(pb[t][j]);
    }
    uint4 af[2][TMC], bf[2][4];
#pragma unroll
    for (int i = 0; i < TMC; ++i) af[0][i] = wl[32 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = lds_ld16_a4(Xs + pb[sb_type(0)][j] + sb_tap(0));
    static_for<kSbSteps>([&](auto S) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value;
      constexpr int cur = s & 1;
      if constexpr (s + 1 < kSbSteps) {   // step s + 1's fragments are read before step s's products issue
        constexpr int ty = sb_type(s + 1), imm = sb_tap(2 * (s + 1));
#pragma unroll
        for (int i = 0; i < TMC; ++i) af[cur ^ 1][i] = wl[(s + 1) * 2 * COUT + 32 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[cur ^ 1][j] = lds_ld16_a4(Xs + pb[ty][j] + imm);
      }
      sched_fence();
#pragma unroll
      for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x16_bf16(af[cur][i], bf[cur][j], acc[i][j]);
      sched_fence();
    });
    ECO_SBTS(1);
    if (next < a.total) draw();   // the patch after `next`
    __syncthreads();   // every wave is done with Xs: the next patch may land
    ECO_SBTS(2);
    const int next2 = !dynamic ? next + (int)gridDim.x : next < a.total ? uniform(next_patch) : a.total;
    if (next < a.total) {
      store_patch();
      ECO_SBTS(14);
    }
    // the patch after it: its loads have the whole epilogue below and the next reduction to land (xv is live across the
    // reduction either way)
    if (next < a.total && next2 < a.total) load_patch(next2);

    ECO_SBTS(3);
    // ---- per 16 channels: bias / BN on the accumulators into the stage, 3x3 stride-2 max over it, ReLU, pooled store.
    // A thread's nine window offsets are fixed per patch (taps outside the conv image -- the MAX window is clipped to
    // it, pooling_layer.cpp:207-212 -- re-read tap (0,0), which a stored output always has). ----
    const int f = patch / tpf, t = patch - f * tpf;
    const int by = t / a.tiles_w, bx = t - by * a.tiles_w;
    const int r0 = 2 * kSbPH * by, q0 = 2 * kSbPW * bx;    // first conv row / column of the patch
    const int gph = kSbPH * by + ph, gpw = kSbPW * bx + pw;
    const bool pool_thread = clo < 2 && gph < a.PHo && gpw < a.PWo;
    int woff[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        woff[dy * 3 + dx] = (r0 + 2 * ph + dy < a.Ho && q0 + 2 * pw + dx < a.Wo)
                                ? dy * kSbSP + (dx & 1) * ((kSbSP + 1) / 2) + (dx >> 1) : 0;
    uint4* const yp0 = a.y + (((long)f * cblocks + clo) * a.PHo + gph) * a.PWo + gpw;   // block clo of m-tile 0
    const long yblk = (long)a.PHo * a.PWo;
    ECO_SBTS(4);
    static_for<TMC>([&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {                     // accumulator registers 2rp, 2rp + 1: two consecutive channels
        const int row = (rp & 1) + 4 * (rp >> 1);          // + 2*half: this lane's pair row within the 16
        const int ch = 32 * i + 2 * row + 4 * half;
        const float2 sc = *(const float2*)(Es + ch), sh2 = *(const float2*)(Es + COUT + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < 3 || last_col_ok)
            sw0[row * STAGE_LD + soff[j]] = pack_bf16x2(fmaxf(acc[i][j][2 * rp] * sc.x + sh2.x, stage_floor),
                                                        fmaxf(acc[i][j][2 * rp + 1] * sc.y + sh2.y, stage_floor)) & stage_mask;
      }
      ECO_SBTS(5 + 4 * i);
      __syncthreads();
      ECO_SBTS(6 + 4 * i);
      if (pool_thread) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const unsigned* const spb = sp0 + 8 * it * STAGE_LD;   // block clo + 2 it = pair rows 4 (clo + 2 it) ...
          unsigned o[4];
          if (a.relu) {
            // ReLU went in front of the stage (it commutes with the MAX window): every staged value is a non-negative bf16,
            // and those order like their bit patterns -- eight packed unsigned 16-bit maxima (v_pk_max_u16) per channel pair
            // where unpacking to fp32 took four VALU instructions per read (stem 0.59 -> 0.555 ms)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const unsigned* sp = spb + u * STAGE_LD;
              u16x2 m = as_u16x2(sp[woff[0]]);
#pragma unroll
              for (int k = 1; k < 9; ++k) m = __builtin_elementwise_max(m, as_u16x2(sp[woff[k]]));
              o[u] = from_u16x2(m);
            }
          } else
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned* sp = spb + u * STAGE_LD;
            unsigned v = sp[woff[0]];
            float lo = bf16_bits_to_f32(v & 0xffffu), hi = bf16_bits_to_f32(v >> 16);
#pragma unroll
            for (int k = 1; k < 9; ++k) {
              v = sp[woff[k]];
              lo = fmaxf(lo, bf16_bits_to_f32(v & 0xffffu));
              hi = fmaxf(hi, bf16_bits_to_f32(v >> 16));
            }
            lo = fmaxf(lo, relu_floor);                    // ReLU commutes with the MAX window
            hi = fmaxf(hi, relu_floor);
            o[u] = pack_bf16x2(lo, hi);                    // exact: both are bf16 values already
          }
          st(yp0 + (4 * i + 2 * it) * yblk, make_uint4(o[0], o[1], o[2], o[3]));
        }
      }
      ECO_SBTS(7 + 4 * i);
      __syncthreads();
      ECO_SBTS(8 + 4 * i);
    });
    if (next >= a.total) break;
    patch = next;
    next = next2;
  }
  if (dynamic && tid == 0) {   // count this workgroup out; the last one out clears the launch's counters
    const unsigned gone = counter_fetch_add(a.ctr + 1, 1u);
    if (gone == gridDim.x - 1u) { counter_store(a.ctr, 0u); counter_store(a.ctr + 1, 0u); }
  }
}

}  // namespace eco

using namespace eco;

static int stemb_dims(int h, int w, int* ho, int* wo, int* pho, int* pwo) {
  *ho = (h + 6 - 7) / 2 + 1;
  *wo = (w + 6 - 7) / 2 + 1;
  // pooling_layer.cpp:131-147 with kernel 3, stride 2, pad 0: ceil((in - 3) / 2) + 1
  *pho = (*ho - 3 + 1) / 2 + 1;
  *pwo = (*wo - 3 + 1) / 2 + 1;
  return *ho >= 3 && *wo >= 3;
}

static unsigned short stemb_bf16(float f) {   // round to nearest even, as the device conversion
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Work counters of the dynamic patch distribution (as eco_spanp_counters in eco_blocked.hip): one slot per stream (or the
// capture ring), cleared again by the launch's last workgroup.
#ifdef ECO_EMU
static unsigned eco_stemb_counters[256 * 2];
static unsigned* stemb_counter_base() { return eco_stemb_counters; }
#else
__device__ unsigned eco_stemb_counters[256 * 2];
static unsigned* stemb_counter_base() {
  static unsigned* base[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(eco_stemb_counters)) != hipSuccess) return nullptr;
    base[dev] = (unsigned*)p;
  }
  return base[dev];
}
#endif
static unsigned* stemb_counter_slot(void* stream) {
  unsigned* base = stemb_counter_base();
  const int slot = counter_slot_index(stream);       // one per stream, or the capture ring; -1: static shares
  return (base && slot >= 0) ? base + 2 * slot : nullptr;
}
namespace eco {
int stemb_counters_reset(void* stream) {
  unsigned* base = stemb_counter_base();
  if (!base) return fail(ECO_ERR_RUNTIME, "counters_reset: no device");
#ifdef ECO_EMU
  (void)stream;
  memset(base, 0, sizeof(unsigned) * 256 * 2);
#else
  if (hipMemsetAsync(base, 0, sizeof(unsigned) * 256 * 2, (hipStream_t)stream) != hipSuccess)
    return fail(ECO_ERR_RUNTIME, "counters_reset: hipMemsetAsync failed");
#endif
  return ECO_OK;
}
}  // namespace eco

extern "C" int64_t eco_stemb_weight_elems(int32_t cout) { return (int64_t)kSbSteps * 2 * cout * 8; }

extern "C" int eco_stemb_pack_weights(const float* w, int32_t cout, void* wp) {
  clear_error();
  ECO_REQUIRE(w && wp && (cout == 32 || cout == 64), "stemb: weights for 32 or 64 output channels (got %d)", cout);
  unsigned short* o = (unsigned short*)wp;
  memset(o, 0, sizeof(unsigned short) * (size_t)eco_stemb_weight_elems(cout));
  // w[m][c][ky][kx] -> wp[s][g][m][e]: kernel row rho = c*7 + ky = 2s + g, tap kx = e (e = 7 and rho = 21 stay zero)
  for (int m = 0; m < cout; ++m)
    for (int rho = 0; rho < 21; ++rho)
      for (int kx = 0; kx < 7; ++kx)
        o[(((long)(rho / 2) * 2 + rho % 2) * cout + m) * 8 + kx] = stemb_bf16(w[((long)m * 21 + rho) * 7 + kx]);
  return ECO_OK;
}

extern "C" int eco_stemb_forward(const float* x, const void* wp, const float* bias, const float* bn_scale,
                                 const float* bn_shift, int32_t relu, void* y, int32_t n, int32_t h, int32_t w,
                                 int32_t cout, int32_t max_workgroups, void* stream) {
  clear_error();
  ECO_REQUIRE(x && wp && y && n > 0 && h > 0 && w > 0 && max_workgroups >= 0, "stemb: bad argument");
  ECO_REQUIRE(cout == 32 || cout == 64, "stemb: 32 or 64 output channels (got %d)", cout);
  ECO_REQUIRE(!bn_scale == !bn_shift, "stemb: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(((uintptr_t)wp & 15) == 0 && ((uintptr_t)y & 15) == 0, "stemb: packed weights and output must be 16-byte aligned");
  StemBArgs a;
  a.x = x; a.wp = (const uint4*)wp; a.bias = bias; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.y = (uint4*)y;
  a.n = n; a.H = h; a.W = w; a.cout = cout; a.relu = relu;
  ECO_REQUIRE(stemb_dims(h, w, &a.Ho, &a.Wo, &a.PHo, &a.PWo), "stemb: image %dx%d too small for conv 7x7/2 + pool 3x3/2", h, w);
  a.tiles_h = (int)ceil_div(a.PHo, kSbPH);
  a.tiles_w = (int)ceil_div(a.PWo, kSbPW);
  const long total = (long)n * a.tiles_h * a.tiles_w;
  ECO_REQUIRE(total < 2147483647l, "stemb: too many patches for one launch");
  a.total = (int)total;
  // persistent workgroups, two per CU: the weights are loaded once per workgroup, not once per patch
  const long cap = max_workgroups ? max_workgroups : 2l * current_device_num_cu();
  const long grid = total < cap ? total : cap;
  static const int dyn = [] { const char* e = getenv("ECO_STEMB_DYNAMIC"); return (e && e[0] == '0') ? 0 : 1; }();
  a.ctr = (dyn && total > 2 * grid) ? stemb_counter_slot(stream) : nullptr;   // (worth a draw per patch only with several patches per workgroup)
  const size_t lds = (size_t)kSbRows * kSbRowBytes + (size_t)kSbSteps * 2 * cout * 16 + sizeof(float) * (size_t)(2 * cout + 16 * (kSbCR * kSbSP + 3));
  hipStream_t s = (hipStream_t)stream;
  if (cout == 64) ECO_RAISE_DYNAMIC_LDS(stemb_kernel<2>, "stemb");
  else ECO_RAISE_DYNAMIC_LDS(stemb_kernel<1>, "stemb");
  if (cout == 64) hipLaunchKernelGGL((stemb_kernel<2>), dim3((unsigned)grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((stemb_kernel<1>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return check_launch("eco_stemb_forward");
}
