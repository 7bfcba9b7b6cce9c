// eco_stem.hip -- the BN-Inception stem as ONE kernel: conv1_7x7_s2 (3 -> cout, 7x7, stride 2, pad 3) + bias +
// folded BN + ReLU + pool1_3x3_s2 (MAX 3x3, stride 2, Caffe ceil rule).
//
//   conv1_7x7_s2 / conv1_7x7_s2_bn / conv1_relu_7x7 / pool1_3x3_s2   models_ECO_Lite/kinetics/deploy.prototxt:8-77
//   ConvolutionLayer::Forward (conv_layer.cpp:28-43, base_conv_layer.cpp:264-287), BN TEST branch
//   (bn_layer.cpp:93-207), ReLU (relu_layer.cpp:10-20), PoolingLayer MAX (pooling_layer.cpp:131-147,199-237)
//
// As separate launches the stem cost 2.05 ms of the 17.3 ms step: conv1 is a K = 147 reduction per output (ten
// 16-row stages of per-element gathers in the table-mode kernel: 0.46 of its floor) whose 1.64 GB output is written,
// then read again by the pool.  Here a workgroup (persistent, two per CU) walks 8 x 14 patches of POOLED outputs:
//   * the whole packed weight block (74 k-pairs x cout) sits in LDS for the workgroup's lifetime, and the 39 x 63 x 3
//     input patch a pooled patch depends on (zero padded) is loaded into LDS once -- the next patch's words are in
//     flight in registers under the current reduction; no stage loop, no gathers from global memory inside it;
//   * the 17 x 29 conv outputs under the patch (one halo row / column, 10 % extra work) are computed as a
//     cout x 512 x 148 GEMM on v_mfma_f32_32x32x2_f32: wave w owns 128 of the 512 position columns, all channels;
//     the B fragment of k = (c, ky, kx) for conv position (r, q) is the LDS word patch[c][2r + ky][2q + kx], i.e.
//     a per-lane base plus a per-k offset (round 3: an instruction immediate, see below);
//   * bias, folded BN and ReLU are applied to the accumulators, which go through LDS (reusing the patch's space,
//     16 channels at a time) so that the 3x3 stride-2 windows can be taken across lanes; only the pooled
//     cout x 8 x 14 values are stored.  conv1's output never exists in HBM.
// Measured (512 frames of 224 x 224, warm): 1.35 ms = 65 % of the fp32-MFMA peak on the executed 139 GFLOP; the
// first cut took 2.1 ms with its operand loads in rolled loops (one memory round trip each) and a table lookup in
// front of every fragment read.
// Round 3: the round-2 version read both operands with a two-word lane stride (patch word 2q + kx, weight word
// 2m + half): every fragment ds_read_b32 was 2-way bank conflicted (SQ_LDS_BANK_CONFLICT 134 M cycles per launch, the
// only kernel of the step with any) and each k-step advanced the per-lane patch offset with five VALU instructions.
// Now the patch rows are stored column-parity split (even columns, then odd columns: column 2q + kx is word
// (kx & 1)*32 + q + (kx >> 1), so consecutive positions read consecutive words), the weights as [k][cout], and
// the offset of tap k is a compile-time immediate of the ds_read: the lanes of the upper half-wave (k odd) differ from
// the lower half by one of five constants, kept as five per-lane base registers -- no VALU in the reduction at all.
// What does NOT help (measured, profiles/r03_notes.md): starting the two workgroups of a CU half a patch apart, or
// raising the epilogue's wave priority.  A wave issuing back-to-back f32 MFMAs keeps its SIMD's VALU to itself: the
// other workgroup's epilogue wave on that SIMD gets no VALU slot and one LDS instruction per 32 cycles until the
// reduction ends (tools/ubench/mfma_partner.hip), so reduction and epilogue of the two workgroups take turns whatever
// their phase.  The levers are the epilogue's own instruction count and latency: the BN scale is folded into the
// LDS-resident weights and the shift into the accumulators' start value, ReLU is applied to the pooled values.
// fp32 arithmetic: the same products as the reference's sgemm, summed per output in k order by the MFMA chain.
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kStemPH = 8, kStemPW = 14;                 // pooled patch per workgroup
constexpr int kStemCR = 2 * kStemPH + 1, kStemCQ = 2 * kStemPW + 1;   // conv patch: 17 x 29
constexpr int kStemNPos = kStemCR * kStemCQ;             // 493 conv positions, padded to 512 columns
constexpr int kStemIR = 2 * (kStemCR - 1) + 7, kStemIQ = 2 * (kStemCQ - 1) + 7;   // input patch: 39 x 63
constexpr int kStemK = 147, kStemKP = 74;                // k = (c, ky, kx); k-pairs (148: one zero row)

// LDS word of tap k = (c, ky, kx) relative to a position's base word (2r*IQ + q): patch row c*IR + ky, column kx in the
// parity-split row layout.
constexpr int stem_ko(int k) {
  const int c = k / 49, ky = (k % 49) / 7, kx = k % 7;
  return (c * kStemIR + ky) * kStemIQ + (kx & 1) * 32 + (kx >> 1);
}
// k-step kp multiplies taps 2kp (lanes 0-31) and 2kp+1 (lanes 32-63); the zero-weight padding row k = 147 re-reads tap
// 145, a word of the position's own receptive field (a word outside it could turn 0 * inf into a NaN the reference's
// output would not have).
constexpr int stem_k1(int kp) { return 2 * kp + 1 < kStemK ? 2 * kp + 1 : kStemK - 2; }
constexpr int stem_delta(int kp) { return stem_ko(stem_k1(kp)) - stem_ko(2 * kp); }
constexpr int stem_imm(int kp) { return stem_delta(kp) < 0 ? stem_ko(stem_k1(kp)) : stem_ko(2 * kp); }
constexpr int kStemDeltaChan = stem_ko(49) - stem_ko(48);   // the one pair that straddles two input channels
// the five (lower half, upper half) base adjustments: next column even->odd, odd->even, next kernel row, next channel,
// and the padding step
constexpr int kStemNT = 5;
constexpr int stem_type(int kp) {
  const int d = stem_delta(kp);
  return d == 32 ? 0 : d == -31 ? 1 : d == kStemIQ - 3 ? 2 : d == kStemDeltaChan ? 3 : d == 31 ? 4 : -1;
}
constexpr bool stem_types_ok() {
  for (int kp = 0; kp < kStemKP; ++kp)
    if (stem_type(kp) < 0 || stem_imm(kp) < 0) return false;
  return true;
}
static_assert(stem_types_ok(), "every k-step's half-wave offset difference is one of the five known constants");

#define STEM_STAMP(slot) do { } while (0)

struct StemArgs {
  const float* x;        // [n][3][H][W]
  const float* wp;       // [148][cout] packed weights (k-major), zero for k = 147
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  float* y;              // [n][cout][PHo][PWo]
  int n, H, W, cout, Ho, Wo, PHo, PWo;   // conv / pooled output sizes
  int relu, tiles_h, tiles_w;
  int total;             // patches (n * tiles_h * tiles_w); a workgroup takes patch blockIdx.x, + gridDim.x, ...
};

// TMC = cout / 32 (1 or 2 m-tiles; every wave holds all channels of its 128 columns)
template <int TMC>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
  ECO_CLOCK("stem");
  constexpr int COUT = 32 * TMC;
  constexpr int IN_ROWS = 3 * kStemIR;                   // 117 patch rows of 63 words
  constexpr int IN_WORDS = IN_ROWS * kStemIQ;            // 7371
  constexpr int W_WORDS = kStemKP * COUT * 2;
  constexpr int STAGE_CH = 16;                           // channels pooled per pass
  constexpr int STAGE_LD = kStemNPos + 3;                // 496: staging row of one channel
  constexpr int XS_WORDS = STAGE_CH * STAGE_LD;          // 7936 >= IN_WORDS: the patch and the stage share it
  static_assert(XS_WORDS >= IN_WORDS, "stage must cover the patch");
  constexpr int XU = (IN_ROWS + 3) / 4, WU = (W_WORDS / 4 + 255) / 256;
  ECO_DYNAMIC_LDS(lds);
  float* const Xs = lds;                                 // [3][39][63], each row: even columns, then odd columns
  float* const Ss = lds;                                 // staging [16][STAGE_LD] (after the reduction)
  float* const Ws = lds + XS_WORDS;                      // [148][COUT], resident for the workgroup's lifetime
  float* const Es = Ws + W_WORDS;                        // [2][COUT]: BN scale (used once), accumulator start value

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tpf = a.tiles_h * a.tiles_w;

  // ---- once per workgroup: the packed weights, with the folded-BN scale multiplied in, and the per-channel constant
  // the accumulators start from: y = (conv + bias) * scale + shift = sum_k (w * scale) x + (bias * scale + shift).
  // f32 MFMAs and VALU instructions of the same SIMD do not overlap on gfx950 (tools/ubench/mfma_valu_overlap.hip:
  // every VALU instruction costs ~6 cycles of the matrix pipe, whichever wave issues it), so the three VALU
  // instructions per conv output the epilogue used to spend here came straight out of the reduction's time. ----
  {
    float4 wv[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = tid + 256 * u;
      wv[u] = i < W_WORDS / 4 ? ld((const float4*)a.wp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < COUT) {
      const float b = a.bias ? ld(a.bias + tid) : 0.0f;
      const float sc = a.bn_scale ? ld(a.bn_scale + tid) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + tid) : 0.0f;
      Es[tid] = sc;
      Es[COUT + tid] = b * sc + sh;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WU; ++u)
      if (tid + 256 * u < W_WORDS / 4) {
        const float4 sc = ((const float4*)Es)[(tid + 256 * u) % (COUT / 4)];   // row k of [148][COUT]: channel = word % COUT
        ((float4*)Ws)[tid + 256 * u] = make_float4(wv[u].x * sc.x, wv[u].y * sc.y, wv[u].z * sc.z, wv[u].w * sc.w);
      }
  }

  // A wave takes whole patch rows (lane = column): the row arithmetic is scalar, a lane's address is base + lane,
  // and every load is in flight before the first one is waited for.
  float xv[XU];
  auto load_patch = [&](int patch) {
    const int f = patch / tpf, t = patch - f * tpf;
    const int by = t / a.tiles_w, bx = t - by * a.tiles_w;
    const int ih0 = 4 * kStemPH * by - 3, iw0 = 4 * kStemPW * bx - 3;     // first input row / column
    const float* xf = a.x + (long)f * 3 * a.H * a.W;
    const int w = iw0 + lane;
    const bool wok = lane < kStemIQ && (unsigned)w < (unsigned)a.W;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int row = wave + 4 * u;                      // (c, rr), wave-uniform
      const int c = row / kStemIR, h = ih0 + row - c * kStemIR;
      const bool ok = wok && row < IN_ROWS && (unsigned)h < (unsigned)a.H;
      xv[u] = ok ? ld(xf + ((long)c * a.H + h) * a.W + w) : 0.0f;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int u = 0; u < XU; ++u)
      if (lane < kStemIQ && wave + 4 * u < IN_ROWS) Xs[(wave + 4 * u) * kStemIQ + (lane & 1) * 32 + (lane >> 1)] = xv[u];
  };

  // ---- this lane's four position columns: conv position p = wave*128 + j*32 + l31 -> patch word 2r*IQ + q ----
  int pbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int p = wave * 128 + j * 32 + l31;
    if (p >= kStemNPos) p = kStemNPos - 1;                // padding columns: any valid word (never stored)
    const int r = p / kStemCQ, q = p - r * kStemCQ;
    pbase[j] = 2 * r * kStemIQ + q;
  }
  const bool last_col_ok = wave * 128 + 96 + l31 < kStemNPos;   // only wave 3's j = 3 has padding columns
  const float* wl = Ws + half * COUT + l31;               // A[m = l31 (+32 i)][k = 2 kp + half]
  // pooling threads: 14 x 8 pooled positions x 2 channel phases = 224 of the 256
  const int pw = tid % kStemPW, ph = (tid / kStemPW) % kStemPH, clo = tid / (kStemPW * kStemPH);
  // The pooling stage keeps a conv row's 29 columns parity split as well (15 even, then 14 odd): the 3x3 stride-2 windows
  // of consecutive pooled columns then read consecutive words (conflict-free ds_read_b32; plain order read them at a
  // two-word lane stride), at the price of 2-way conflicted ds_write_b32 in the epilogue -- which the LDS serves at
  // full rate (MI355X_MICROARCH.md, LDS).
  const float* const sp0 = Ss + clo * STAGE_LD + 2 * ph * kStemCQ + pw;
  int soff[4];                                            // stage word of this lane's four conv positions
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int p = wave * 128 + j * 32 + l31;
    if (p >= kStemNPos) p = kStemNPos - 1;
    const int r = p / kStemCQ, q = p - r * kStemCQ;
    soff[j] = r * kStemCQ + (q & 1) * ((kStemCQ + 1) / 2) + (q >> 1);
  }
  float* const sw0 = Ss + 4 * half * STAGE_LD;
  const long ych = (long)a.PHo * a.PWo;
  const float relu_floor = a.relu ? 0.0f : -FLT_MAX;

  int stamp_n = 0;
  (void)stamp_n;
  int patch = (int)blockIdx.x;
  load_patch(patch);
  store_patch();
  __syncthreads();

  while (true) {
    const int next = patch + (int)gridDim.x;
    STEM_STAMP(0);
    if (next < a.total) load_patch(next);                 // in flight under the reduction below

    f32x16 acc[TMC][4];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float c0 = Es[COUT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j][r] = c0;
      }

    // Fragment addresses: base register of the k-step's type + compile-time immediate (stem_imm).  The bases are made
    // opaque per patch: as loop invariants the compiler would otherwise materialise all 74 x 4 fragment addresses
    // outside the patch loop and spill them.
    int pb[kStemNT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pb[0][j] = pbase[j] + (half ? 32 : 0);
      pb[1][j] = pbase[j] + (half ? 0 : 31);
      pb[2][j] = pbase[j] + (half ? kStemIQ - 3 : 0);
      pb[3][j] = pbase[j] + (half ? kStemDeltaChan : 0);
      pb[4][j] = pbase[j] + (half ? 31 : 0);
#pragma unroll
      for (int t = 0; t < kStemNT; ++t) ECO_OPAQUE(pb[t][j]);
    }
    float af[2][TMC], bf[2][4];
#pragma unroll
    for (int i = 0; i < TMC; ++i) af[0][i] = wl[32 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = Xs[pb[stem_type(0)][j] + stem_imm(0)];
    // unrolled over the 74 k-steps with the step as a compile-time constant (a `#pragma unroll` loop leaves the tap
    // tables as run-time arithmetic until after the unroller has priced -- and refused -- the body)
    static_for<kStemKP>([&](auto KP) __attribute__((always_inline)) {
      constexpr int kp = decltype(KP)::value;
      constexpr int cur = kp & 1;
      if constexpr (kp + 1 < kStemKP) {   // step kp + 1's fragments are read before step kp's products issue
        constexpr int ty = stem_type(kp + 1), imm = stem_imm(kp + 1);
#pragma unroll
        for (int i = 0; i < TMC; ++i) af[cur ^ 1][i] = wl[(kp + 1) * 2 * COUT + 32 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[cur ^ 1][j] = Xs[pb[ty][j] + imm];
      }
      sched_fence();   // reads of step kp + 1, then the products of step kp: the reads' latency hides under 8 MFMAs
#pragma unroll
      for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x2(af[cur][i], bf[cur][j], acc[i][j]);
      sched_fence();
    });
    STEM_STAMP(1);
    __syncthreads();   // every wave is done with Xs: the space becomes the pooling stage
    STEM_STAMP(2);

    // ---- per 16 channels: epilogue into the stage, 3x3 stride-2 max over it, pooled store.  A thread's nine window
    // offsets are fixed per patch (taps outside the conv image -- the MAX window is clipped to it,
    // pooling_layer.cpp:207-212 -- re-read tap (0,0), which a stored output always has); its channel advances by
    // two per step: immediate offsets only. ----
    const int f = patch / tpf, t = patch - f * tpf;
    const int by = t / a.tiles_w, bx = t - by * a.tiles_w;
    const int r0 = 2 * kStemPH * by, q0 = 2 * kStemPW * bx;          // first conv row / column of the patch
    const int gph = kStemPH * by + ph, gpw = kStemPW * bx + pw;
    const bool pool_thread = clo < 2 && gph < a.PHo && gpw < a.PWo;
    int woff[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        woff[dy * 3 + dx] = (r0 + 2 * ph + dy < a.Ho && q0 + 2 * pw + dx < a.Wo)
                                ? dy * kStemCQ + (dx & 1) * ((kStemCQ + 1) / 2) + (dx >> 1) : 0;
    float* const yp0 = a.y + (((long)f * a.cout + clo) * a.PHo + gph) * a.PWo + gpw;
#pragma unroll
    for (int i = 0; i < TMC; ++i) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = 8 * rh + rr;
          const int cl0 = (rr & 3) + 8 * (rr >> 2);        // + 4*half: this lane's channel within the 16
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < 3 || last_col_ok) sw0[cl0 * STAGE_LD + soff[j]] = acc[i][j][r];
        }
        __syncthreads();
        if (pool_thread) {
#pragma unroll
          for (int u = 0; u < STAGE_CH / 2; ++u) {
            const float* sp = sp0 + 2 * u * STAGE_LD;
            float m = sp[woff[0]];
#pragma unroll
            for (int k = 1; k < 9; ++k) m = fmaxf(m, sp[woff[k]]);
            st(yp0 + (32 * i + 16 * rh + 2 * u) * ych, fmaxf(m, relu_floor));   // ReLU commutes with the MAX window
          }
        }
        __syncthreads();
      }
    }
    STEM_STAMP(3);
    if (next >= a.total) break;
    store_patch();
    __syncthreads();
    patch = next;
  }
}

}  // namespace eco

using namespace eco;


static int stem_dims(int h, int w, int* ho, int* wo, int* pho, int* pwo) {
  *ho = (h + 6 - 7) / 2 + 1;
  *wo = (w + 6 - 7) / 2 + 1;
  // pooling_layer.cpp:131-147 with kernel 3, stride 2, pad 0: ceil((in - 3) / 2) + 1
  *pho = (*ho - 3 + 1) / 2 + 1;
  *pwo = (*wo - 3 + 1) / 2 + 1;
  return *ho >= 3 && *wo >= 3;
}

extern "C" int eco_stem_pack_weights(const float* w, int32_t cout, float* wp) {
  clear_error();
  ECO_REQUIRE(w && wp && (cout == 32 || cout == 64), "stem: weights for 32 or 64 output channels (got %d)", cout);
  memset(wp, 0, sizeof(float) * (size_t)kStemKP * cout * 2);
  for (int k = 0; k < kStemK; ++k)
    for (int m = 0; m < cout; ++m) wp[(long)k * cout + m] = w[(long)m * kStemK + k];
  return ECO_OK;
}

extern "C" int eco_stem_forward(const float* x, const float* wp, const float* bias, const float* bn_scale,
                                const float* bn_shift, int32_t relu, float* y, int32_t n, int32_t h, int32_t w,
                                int32_t cout, int32_t max_workgroups, void* stream) {
  clear_error();
  ECO_REQUIRE(x && wp && y && n > 0 && h > 0 && w > 0 && max_workgroups >= 0, "stem: bad argument");
  ECO_REQUIRE(cout == 32 || cout == 64, "stem: 32 or 64 output channels (got %d)", cout);
  ECO_REQUIRE(!bn_scale == !bn_shift, "stem: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(((uintptr_t)wp & 15) == 0, "stem: packed weights must be 16-byte aligned");
  StemArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.y = y;
  a.n = n; a.H = h; a.W = w; a.cout = cout; a.relu = relu;
  ECO_REQUIRE(stem_dims(h, w, &a.Ho, &a.Wo, &a.PHo, &a.PWo), "stem: image %dx%d too small for conv 7x7/2 + pool 3x3/2", h, w);
  a.tiles_h = (int)ceil_div(a.PHo, kStemPH);
  a.tiles_w = (int)ceil_div(a.PWo, kStemPW);
  const long total = (long)n * a.tiles_h * a.tiles_w;
  ECO_REQUIRE(total < 2147483647l, "stem: too many patches for one launch");
  a.total = (int)total;
  // persistent workgroups, two per CU: the weights are loaded once per workgroup, not once per patch
  const long cap = max_workgroups ? max_workgroups : 2l * current_device_num_cu();
  const long grid = total < cap ? total : cap;
  const size_t lds = sizeof(float) * (size_t)(16 * (kStemNPos + 3) + kStemKP * cout * 2 + 3 * cout);
  hipStream_t s = (hipStream_t)stream;
  if (cout == 64) ECO_RAISE_DYNAMIC_LDS(stem_kernel<2>, "stem");
  else ECO_RAISE_DYNAMIC_LDS(stem_kernel<1>, "stem");
  if (cout == 64) hipLaunchKernelGGL((stem_kernel<2>), dim3((unsigned)grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((stem_kernel<1>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return check_launch("eco_stem_forward");
}
