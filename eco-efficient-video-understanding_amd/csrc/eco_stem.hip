// eco_stem.hip -- the BN-Inception stem as ONE kernel: conv1_7x7_s2 (3 -> cout, 7x7, stride 2, pad 3) + bias +
// folded BN + ReLU + pool1_3x3_s2 (MAX 3x3, stride 2, Caffe ceil rule).
//
//   conv1_7x7_s2 / conv1_7x7_s2_bn / conv1_relu_7x7 / pool1_3x3_s2   models_ECO_Lite/kinetics/deploy.prototxt:8-77
//   ConvolutionLayer::Forward (conv_layer.cpp:28-43, base_conv_layer.cpp:264-287), BN TEST branch
//   (bn_layer.cpp:93-207), ReLU (relu_layer.cpp:10-20), PoolingLayer MAX (pooling_layer.cpp:131-147,199-237)
//
// As separate launches the stem cost 2.05 ms of the 17.3 ms step: conv1 is a K = 147 reduction per output (ten
// 16-row stages of per-element gathers in the table-mode kernel: 0.46 of its floor) whose 1.64 GB output is written,
// then read again by the pool.  Here a workgroup owns an 8 x 14 patch of POOLED outputs of one frame:
//   * the 39 x 63 x 3 input patch it depends on (zero padded) and the whole packed weight block (74 k-pairs x cout)
//     are loaded into LDS once -- no stage loop, no gathers from global memory inside the reduction;
//   * the 17 x 29 conv outputs under the patch (one halo row / column, 10 % extra work) are computed as a
//     cout x 512 x 148 GEMM on v_mfma_f32_32x32x2_f32: wave w owns 128 of the 512 position columns, all channels;
//     the B fragment of k = (c, ky, kx) for conv position (r, q) is the LDS word patch[c][2r + ky][2q + kx], i.e.
//     a per-lane base plus a per-k offset that is wave-uniform per half (two scalars per k-pair from a table);
//   * bias, folded BN and ReLU are applied to the accumulators, which go through LDS (reusing the operand space,
//     32 channels at a time) so that the 3x3 stride-2 windows can be taken across lanes; only the pooled
//     cout x 8 x 14 values are stored.  conv1's output never exists in HBM.
// fp32 arithmetic: the same products as the reference's sgemm, summed per output in k order by the MFMA chain.
#include <float.h>
#include <string.h>

#include "eco_common.h"

namespace eco {

constexpr int kStemPH = 8, kStemPW = 14;                 // pooled patch per workgroup
constexpr int kStemCR = 2 * kStemPH + 1, kStemCQ = 2 * kStemPW + 1;   // conv patch: 17 x 29
constexpr int kStemNPos = kStemCR * kStemCQ;             // 493 conv positions, padded to 512 columns
constexpr int kStemIR = 2 * (kStemCR - 1) + 7, kStemIQ = 2 * (kStemCQ - 1) + 7;   // input patch: 39 x 63
constexpr int kStemK = 147, kStemKP = 74;                // k = (c, ky, kx); k-pairs (148: one zero row)

struct StemArgs {
  const float* x;        // [n][3][H][W]
  const float* wp;       // [74][cout][2] packed weights (k-pair interleaved), zero for k = 147
  const int* koff;       // [148] LDS word offset of k in the input patch: c*IR*IQ + ky*IQ + kx
  const float* bias;
  const float* bn_scale;
  const float* bn_shift;
  float* y;              // [n][cout][PHo][PWo]
  int n, H, W, cout, Ho, Wo, PHo, PWo;   // conv / pooled output sizes
  int relu, tiles_h, tiles_w;
};

// TMC = cout / 32 (1 or 2 m-tiles; every wave holds all channels of its 128 columns)
template <int TMC>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
  constexpr int COUT = 32 * TMC;
  constexpr int IN_WORDS = 3 * kStemIR * kStemIQ;        // 7371
  constexpr int W_WORDS = kStemKP * COUT * 2;
  constexpr int STAGE_LD = kStemNPos + 3;                // 496: staging row of one channel
  ECO_DYNAMIC_LDS(lds);
  float* const Xs = lds;                                 // [3][39][63]
  float* const Ws = lds + ((IN_WORDS + 3) & ~3);         // [74][COUT][2]
  float* const Ss = lds;                                 // staging [32][STAGE_LD] (after the reduction)

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tpf = a.tiles_h * a.tiles_w;
  const int f = (int)blockIdx.x / tpf, t = (int)blockIdx.x - f * tpf;
  const int by = t / a.tiles_w, bx = t - by * a.tiles_w;
  const int r0 = 2 * kStemPH * by, q0 = 2 * kStemPW * bx;        // first conv row / column of the patch
  const int ih0 = 2 * r0 - 3, iw0 = 2 * q0 - 3;                  // first input row / column

  // ---- operands into LDS ----
  for (int i = tid; i < IN_WORDS; i += 256) {
    const int q = i % kStemIQ, rr = (i / kStemIQ) % kStemIR, c = i / (kStemIQ * kStemIR);
    const int h = ih0 + rr, w = iw0 + q;
    const bool ok = (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
    Xs[i] = ok ? ld(a.x + (((long)f * 3 + c) * a.H + (ok ? h : 0)) * a.W + (ok ? w : 0)) : 0.0f;
  }
  for (int i = tid; i < W_WORDS / 4; i += 256) ((float4*)Ws)[i] = ld((const float4*)a.wp + i);
  __syncthreads();

  // ---- this lane's four position columns: conv position p = wave*128 + j*32 + l31 -> patch word 2r*IQ + 2q ----
  int pbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int p = wave * 128 + j * 32 + l31;
    if (p >= kStemNPos) p = kStemNPos - 1;                // padding columns: any valid word (never stored)
    const int r = p / kStemCQ, q = p - r * kStemCQ;
    pbase[j] = 2 * r * kStemIQ + 2 * q;
  }
  f32x16 acc[TMC][4];
#pragma unroll
  for (int i = 0; i < TMC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float* wl = Ws + 2 * l31 + half;                  // A[m = l31 (+32 i)][k = 2 kp + half]
#pragma unroll 2
  for (int kp = 0; kp < kStemKP; ++kp) {
    const int k0 = ld(a.koff + 2 * kp), k1 = ld(a.koff + 2 * kp + 1);   // wave-uniform: scalar loads
    const int ko = half ? k1 : k0;
    float af[TMC], bf[4];
#pragma unroll
    for (int i = 0; i < TMC; ++i) af[i] = wl[(kp * COUT + 32 * i) * 2];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = Xs[pbase[j] + ko];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x2(af[i], bf[j], acc[i][j]);
  }
  __syncthreads();   // every wave is done with Xs / Ws: the space becomes the pooling stage

  // ---- per 32 channels: epilogue into the stage, 3x3 stride-2 max over it, pooled store ----
#pragma unroll
  for (int i = 0; i < TMC; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = (r & 3) + 8 * (r >> 2) + 4 * half, ch = 32 * i + cl;
      const float b = a.bias ? ld(a.bias + ch) : 0.0f;
      const float sc = a.bn_scale ? ld(a.bn_scale + ch) : 1.0f, sh = a.bn_scale ? ld(a.bn_shift + ch) : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = wave * 128 + j * 32 + l31;
        float v = (acc[i][j][r] + b) * sc + sh;
        if (a.relu) v = fmaxf(v, 0.0f);
        if (p < kStemNPos) Ss[cl * STAGE_LD + p] = v;
      }
    }
    __syncthreads();
    for (int o = tid; o < 32 * kStemPH * kStemPW; o += 256) {
      const int pw = o % kStemPW, ph = (o / kStemPW) % kStemPH, cl = o / (kStemPW * kStemPH);
      const int gph = kStemPH * by + ph, gpw = kStemPW * bx + pw;
      if (gph >= a.PHo || gpw >= a.PWo) continue;
      const float* sp = Ss + cl * STAGE_LD + 2 * ph * kStemCQ + 2 * pw;
      float m = -FLT_MAX;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)   // MAX clips its window to the conv image (pooling_layer.cpp:207-212)
          if (r0 + 2 * ph + dy < a.Ho && q0 + 2 * pw + dx < a.Wo) m = fmaxf(m, sp[dy * kStemCQ + dx]);
      st(a.y + (((long)f * a.cout + 32 * i + cl) * a.PHo + gph) * a.PWo + gpw, m);
    }
    if (i + 1 < TMC) __syncthreads();
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

extern "C" int eco_stem_pack_weights(const float* w, int32_t cout, float* wp, int32_t* koff) {
  clear_error();
  ECO_REQUIRE(w && wp && koff && (cout == 32 || cout == 64), "stem: weights for 32 or 64 output channels (got %d)", cout);
  memset(wp, 0, sizeof(float) * (size_t)kStemKP * cout * 2);
  for (int k = 0; k < 2 * kStemKP; ++k) {
    const int kk = k < kStemK ? k : 0;                 // the padding row multiplies zero weights: any in-range word
    const int c = kk / 49, ky = (kk % 49) / 7, kx = kk % 7;
    koff[k] = c * kStemIR * kStemIQ + ky * kStemIQ + kx;
    if (k < kStemK)
      for (int m = 0; m < cout; ++m) wp[((long)(k / 2) * cout + m) * 2 + (k & 1)] = w[(long)m * kStemK + k];
  }
  return ECO_OK;
}

extern "C" int eco_stem_forward(const float* x, const float* wp, const int32_t* koff, const float* bias,
                                const float* bn_scale, const float* bn_shift, int32_t relu, float* y, int32_t n,
                                int32_t h, int32_t w, int32_t cout, void* stream) {
  clear_error();
  ECO_REQUIRE(x && wp && koff && y && n > 0 && h > 0 && w > 0, "stem: bad argument");
  ECO_REQUIRE(cout == 32 || cout == 64, "stem: 32 or 64 output channels (got %d)", cout);
  ECO_REQUIRE(!bn_scale == !bn_shift, "stem: bn_scale and bn_shift must be given together");
  ECO_REQUIRE(((uintptr_t)wp & 15) == 0, "stem: packed weights must be 16-byte aligned");
  StemArgs a;
  a.x = x; a.wp = wp; a.koff = koff; a.bias = bias; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.y = y;
  a.n = n; a.H = h; a.W = w; a.cout = cout; a.relu = relu;
  ECO_REQUIRE(stem_dims(h, w, &a.Ho, &a.Wo, &a.PHo, &a.PWo), "stem: image %dx%d too small for conv 7x7/2 + pool 3x3/2", h, w);
  a.tiles_h = (int)ceil_div(a.PHo, kStemPH);
  a.tiles_w = (int)ceil_div(a.PWo, kStemPW);
  const long grid = (long)n * a.tiles_h * a.tiles_w;
  ECO_REQUIRE(grid < 2147483647l, "stem: too many patches for one launch");
  const int in_words = (3 * kStemIR * kStemIQ + 3) & ~3;
  size_t lds = sizeof(float) * (size_t)(in_words + kStemKP * cout * 2);
  const size_t stage = sizeof(float) * 32 * (kStemNPos + 3);
  if (stage > lds) lds = stage;
  hipStream_t s = (hipStream_t)stream;
#ifndef ECO_EMU
  {
    static thread_local bool raised[2] = {false, false};
    const int idx = cout == 64;
    if (!raised[idx]) {
      hipError_t e = cout == 64 ? hipFuncSetAttribute((const void*)stem_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                                : hipFuncSetAttribute((const void*)stem_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return fail(ECO_ERR_RUNTIME, "stem: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      raised[idx] = true;
    }
  }
#endif
  if (cout == 64) hipLaunchKernelGGL((stem_kernel<2>), dim3((unsigned)grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((stem_kernel<1>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return check_launch("eco_stem_forward");
}
