/*
 * eco_hip.h -- C ABI of libeco_hip.so: MI355X (gfx950) HIP kernels for the ECO
 * inference forward path (BN-Inception 2-D head -> r2Dto3D -> 3D-ResNet-18 trunk
 * -> global_pool -> fc).
 *
 * Drop-in boundary.  Each entry point replaces the GPU forward of one reference
 * operator (caffe_3d `Layer<Dtype>::Forward_gpu`, include/caffe/layer.hpp:444-477)
 * or a fused group of them; the reference file:line each one stands in for is
 * given at its declaration.  The contract mirrors the Layer plug-point:
 *
 *   - plain pointers and PODs only (no torch / HIP types in signatures; the
 *     stream is passed as an opaque `void*` that must be a hipStream_t or NULL),
 *   - tensors are fp32, row-major N,C,[D,]H,W exactly like caffe `Blob`
 *     (include/caffe/blob.hpp:24-282); weights keep the reference layouts
 *     ([Cout,Cin,(kd,)kh,kw], BN 4 x [1,C], fc [out,in]); the bf16 path at the end of this
 *     header keeps its activations channel-blocked internally,
 *   - every function returns ECO_OK or a negative error code and never aborts,
 *     allocates, or synchronises: the caller owns memory and the stream
 *     (the reference LOG(FATAL)s instead; `eco_last_error()` carries the text a
 *     CHECK would have printed),
 *   - thread-compatible: no global state except the thread-local error string -- and the two items declared under
 *     "Process-level state" below (environment switches read once, the work counters of the dynamic-share launches).
 *
 * A second build of the same sources against a CPU fiber emulator
 * (tests/emu/, libeco_emu.so) exports the identical symbols; it exists only so
 * that the CPU test-suite can exercise kernel index math without a GPU and is
 * never loaded by the product package.
 */
#ifndef ECO_HIP_H_
#define ECO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECO_ABI_VERSION 19

#define ECO_OK 0
#define ECO_ERR_INVALID (-1) /* bad argument / geometry not supported on this path */
#define ECO_ERR_RUNTIME (-2) /* HIP runtime error (launch failure, bad device)     */

/* reduction (K) order of a packed convolution */
#define ECO_CONV_MODE_TABLE 0 /* k = c*taps + tap, per-k gather table (any cin)            */
#define ECO_CONV_MODE_CTAP 1  /* k = (c/kc*taps + tap)*kc + c%kc, one tap per stage (cin%kc==0) */
#define ECO_CONV_MODE_SPAN 2  /* CTAP order; stride-1 same-size (kd)x3x3 convs stage input spans in LDS */
#define ECO_CONV_MODE_POINT 3 /* CTAP order; 1x1 stride-1 convs as a GEMM staged by LDS-DMA              */

#define ECO_POOL_MAX 0
#define ECO_POOL_AVE 1

/* ---- library / device ------------------------------------------------------------ */

/* ABI version of the loaded library (== ECO_ABI_VERSION of the header it was built from). */
int eco_abi_version(void);
/* 1 when this is the gfx950 HIP build, 0 for the CPU emulator build (tests only). */
int eco_is_device_build(void);
/* Text of the last error on this thread ("" if none). Replaces glog CHECK/LOG(FATAL) output. */
const char* eco_last_error(void);
/* Caffe::SetDevice / device_query (caffe_3d/src/caffe/common.cpp:140-190). */
int eco_device_count(int* count);
int eco_set_device(int device);
/* Fills name (NUL-terminated, <= name_len), compute-unit count and HBM bytes of `device`. */
int eco_device_info(int device, char* name, size_t name_len, int* num_cu, uint64_t* hbm_bytes);
/* PCI address of `device` as sysfs spells it ("0000:c1:00.0", NUL-terminated, <= len): what a multi-process launcher
 * needs to put each rank on the cores of ITS GPU's NUMA node (the reference is one unpinned MPI process per GPU,
 * caffe_3d/tools/caffe.cpp:150-200).  (v17) */
int eco_device_pci_bus_id(int device, char* pci, size_t len);
/* SHA-256 (64 hex digits) of the sources this library was built from -- the .hip and .h files of csrc, this header and the
 * Makefile, concatenated in sorted path order -- so that measurements (profiles/) can be tied to a build by what it was
 * built FROM: two builds of identical sources differ in their bytes, not in this digest.  "unknown" when the library
 * was compiled outside csrc/Makefile.  (v17) */
const char* eco_source_digest(void);

/* ---- Process-level state (everything the library keeps besides the thread-local error string) --------------------
 *
 * ENVIRONMENT SWITCHES.  Read with getenv() the first time the path they guard is taken, then cached for the life of the
 * process; none changes results beyond fp32 summation order, all select between forms that the tests run against the same
 * oracle (tests/test_fallback_paths.py, on the emulator and on the GPU).  Unset = the default in brackets.
 *   ECO_SPANP=0           bf16 stride-1 3x3(x3) convolutions on the per-tap LDS-DMA kernel (the fallback of views >= 2 GB) instead of
 *                         the persistent span kernel [persistent]
 *   ECO_SPANP_DYNAMIC=0   the persistent kernel's workgroups take equal static item shares instead of drawing from a counter [dynamic]
 *   ECO_STEMB_DYNAMIC=0   the same for the fused bf16 stem's patches [dynamic]
 *   ECO_CONVB_DMA_BUF=0   the LDS-DMA kernel addresses its operands with 64-bit flat addresses instead of buffer descriptors
 *                         (the form views >= 2 GB fall back to by themselves) [descriptors]
 *   (ECO_CONVB_TAIL / ECO_CONVB_KSPLIT exist in experiment builds compiled with -DECO_CONVB_KSPLIT_ENV only: tools/exp)
 *   ECO_STREAMK=1         eco_conv_plan_create: stream-K persistent form of the fp32 gather kernel [split-K + reduce launch]
 * No other entry point reads the environment.
 *
 * WORK COUNTERS.  The persistent bf16 kernels (convb_spanp_kernel, stemb_kernel) draw work items from a counter in
 * device memory that the launch's last workgroup resets to zero.  The counters live in two module-scope arrays of 256
 * slots per device.  Slot choice (eco_counter_slot_probe reports it):
 *   - an eager launch uses the one slot of ITS STREAM (launches of one stream execute in order), for up to 64 distinct
 *     streams per process; the null stream and hipStreamLegacy are one stream; hipStreamPerThread -- one handle value for a
 *     different stream in every host thread -- always takes static shares.  When the table is full, the entry of a stream
 *     without work in flight (destroyed or idle) is recycled; if every one of the 64 is busy the launch takes static shares;
 *   - a launch recorded during stream capture takes the next of 192 capture slots and the captured graph keeps it.  The
 *     slots are handed out ONCE (v19; v18 wrapped around, so two live graphs -- re-captures after a reshape, graphs of
 *     several Net objects -- could end up on one slot and race when replayed on different streams): once the 192 are
 *     gone, captured launches take static shares.  eco_counters_release_capture_slots() returns them all to the pool; call
 *     it only when every graph captured so far has been destroyed.
 * What this guarantees: any number of host threads / streams may launch eagerly at the same time, and captured graphs may
 * replay next to each other and next to eager launches; no combination shares a counter between launches that can run
 * concurrently.  Static shares (slot -1) are always correct, only less balanced.  What it does not: a kernel that faults
 * mid-flight leaves its slot non-zero -- call eco_counters_reset(stream) after recovering the device.
 * (Round-4 / round-5 advisor findings; until v17 a sequence number modulo 256 chose the slot.) */
int eco_counters_reset(void* stream);
int eco_counters_release_capture_slots(void);   /* (v19) */
int eco_counter_slot_probe(void* stream);       /* (v19) slot (0..255) a dynamic-share launch on `stream` would take now, -1 = static */

/* ---- convolution (+ fused bias / residual / BN / ReLU epilogue) -------------------- */

/* Geometry of one N-D cross-correlation, as resolved by
 * BaseConvolutionLayer::LayerSetUp/Reshape (layers/base_conv_layer.cpp:13-261) and
 * ConvolutionLayer::compute_output_shape (layers/conv_layer.cpp:12-25).
 * 2-D layers use in[0]=out[0]=kernel[0]=stride[0]=1, pad[0]=0.  group=1, dilation=1. */
typedef struct eco_conv_geom {
  int32_t n;         /* images (2-D) or clips (3-D)           */
  int32_t cin, cout;
  int32_t in[3];     /* input  D,H,W                           */
  int32_t kernel[3]; /* kd,kh,kw                               */
  int32_t stride[3];
  int32_t pad[3];
  int32_t out[3];    /* output D,H,W = (in+2*pad-kernel)/stride+1 (validated) */
} eco_conv_geom;

/* Tiling chosen for a geometry; fixes the packed-weight layout. */
typedef struct eco_conv_plan {
  int32_t bm, bn, kc; /* block tile: bm output channels x bn output positions, kc reduction rows per stage */
  int32_t k;          /* cin*kd*kh*kw                                   */
  int32_t kpad;       /* k rounded up to a multiple of kc               */
  int32_t mpad;       /* cout rounded up to a multiple of 128           */
  int64_t wp_elems;   /* floats in the packed weight buffer  (kpad*mpad) */
  int64_t ktab_elems; /* int32 entries in the gather table (kpad; stream-K plans: + columns + 1 for the stage prefix table) */
  int32_t mode;       /* ECO_CONV_MODE_*: reduction order of the packed weights / kernel family */
  int32_t ksplit;     /* >1: the last split_tiles output tiles have their reduction cut into ksplit slices */
  int64_t ws_bytes;   /* bytes of device scratch eco_conv_forward needs for this plan (0 if none) */
  int32_t split_tiles; /* 0 (no split-K), all tiles (few-tile layers) or the tail of a many-tile launch */
  int32_t streamk_wgs; /* >0: stream-K -- that many persistent workgroups share the launch's reduction stages evenly
                        * across tile boundaries (ksplit = 1, split_tiles = 0; ws_bytes = two bm x bn blocks and one
                        * flag per workgroup); chosen where every tile of a single launch would otherwise be split */
} eco_conv_plan;

/* Strided view of an N,C,[D,]H,W output (or residual) tensor.  Element
 * (img, c, sp) with sp the row-major index over the spatial dims lives at
 *   ptr[(img / t)*stride_b + (img % t)*stride_t + c*stride_c + sp].
 * Plain tensor: t=1, stride_b=C*S, stride_c=S.  Writing into a channel slice of a
 * Concat top (layers/concat_layer.cpp:54-70) = larger stride_b + offset ptr.
 * Writing through r2Dto3D+Permute [B*T,C,H,W] -> [B,C,T,H,W]
 * (layers/reshape_layer.cpp:88, layers/permute_layer.cpp:9-26) = t=T,
 * stride_b=C*T*S, stride_t=S, stride_c=T*S. */
typedef struct eco_view {
  float* ptr;
  int64_t stride_b, stride_t, stride_c;
  int32_t t;
} eco_view;

/* Fused epilogue, applied per output element v = conv(x,w)[img,c,sp]:
 *   v += bias[c]                       (forward_cpu_bias, base_conv_layer.cpp:282-287)
 *   v += residual(img,c,sp)            (Eltwise SUM, layers/eltwise_layer.cpp:66-72)
 *   raw(img,c,sp) = v                  (the conv / eltwise top itself)
 *   a = v*bn_scale[c] + bn_shift[c]    (BN TEST branch folded: scale=gamma/sqrt(var+eps),
 *                                       shift=beta-mean*scale; bn_layer.cpp:93-207,
 *                                       same algebra as python/gen_bn_inference.py:121-134)
 *   act(img,c,sp) = relu ? max(a,0) : a   (layers/relu_layer.cpp:10-20)
 *                                      NON-FINITE VALUES: +-Inf propagate everywhere.  A NaN reaches `raw` as NaN; in
 *                                      `act` the activation is evaluated as ONE v_max_f32 against a floor (0 with
 *                                      ReLU, -FLT_MAX / -Inf without: IEEE maxNum returns the non-NaN operand), so a
 *                                      NaN pre-activation is stored as that floor, where the reference's std::max
 *                                      (relu_layer.cpp:17) would keep the NaN.  A caller that uses NaN as a diagnostic
 *                                      signal should look at `raw` (or run fuse=False: eco_bn_forward /
 *                                      eco_relu_forward propagate it).
 *   act2(img,c,sp) = the same value       (a second destination of the activated output: the blob feeds both
 *                                          a 2-D consumer and, through r2Dto3D + Permute, the 3-D trunk --
 *                                          inception_3c_double_3x3_1_bn of ECO-Full,
 *                                          models_ECO_Full/kinetics/deploy.prototxt:1835-1870)
 * Any of bias / residual.ptr / raw.ptr / act.ptr / act2.ptr may be NULL (that step is skipped);
 * bn_scale==NULL means a = v.  At least one of raw.ptr / act.ptr must be non-NULL; act2 needs act.
 *
 * Sibling convolutions: convs of one geometry that read the same bottom -- the 1x1 / 3x3_reduce /
 * double_3x3_reduce convs of an Inception block (models_ECO_Lite/kinetics/deploy.prototxt:130-330), a residual
 * block's first conv and its projection shortcut (res4a_1 / res4a_down, res5a_1 / res5a_down) -- can run as ONE
 * conv whose weights, bias and folded BN vectors are the members' concatenated along the output channel:
 * nseg = members - 1 (0 = plain, at most ECO_MAX_SEG).  Channels [0, seg_begin[0]) go to `act` with `relu` as usual; channels
 * [seg_begin[s], seg_begin[s+1] or cout) go to seg_act[s] at channel (c - seg_begin[s]) with seg_relu[s].  A member
 * that wants its raw value gets scale 1 / shift 0 / no ReLU.  Boundaries are multiples of 32; segmented launches
 * take act-style destinations only (no residual / raw / act2), plain views (t = 1), the direct kernels
 * (eco_conv_forward, eco_convb_forward; the Winograd output transforms refuse them). */
#define ECO_MAX_SEG 3   /* extra output segments of a sibling launch: up to four member convolutions */
typedef struct eco_conv_epilogue {
  const float* bias;
  eco_view residual; /* read-only */
  eco_view raw;
  const float* bn_scale;
  const float* bn_shift;
  int32_t relu;
  eco_view act;
  eco_view act2;
  int32_t nseg;
  int32_t seg_begin[ECO_MAX_SEG];
  int32_t seg_relu[ECO_MAX_SEG];
  eco_view seg_act[ECO_MAX_SEG];
} eco_conv_epilogue;

/* Validates `g` (fills nothing) and chooses the tiling for the calling thread's current device (its
 * compute-unit count as eco_device_info reports it: 256 on MI355X). */
int eco_conv_plan_create(const eco_conv_geom* g, eco_conv_plan* plan);
/* Same, for a device with `num_cu` compute units (0 = the current device; the CPU test-suite uses tiny
 * values to reach the many-tile code paths with emulator-sized problems). */
int eco_conv_plan_create_ex(const eco_conv_geom* g, int32_t num_cu, eco_conv_plan* plan);
/* Plan for eco_conv_forward_batched: `batch` entries of this geometry share one launch, so the tile
 * count that is weighed against the device is tiles * batch. */
int eco_conv_plan_create_batched(const eco_conv_geom* g, int32_t num_cu, int32_t batch, eco_conv_plan* plan);
/* HOST function.  Re-lays caffe weights w[cout][cin][kd][kh][kw] (host pointer) into the
 * kernel's K-major image (zero padded; wp[kpad][mpad] for the gather modes, k-pair
 * interleaved wp[kpad/2][mpad][2] for ECO_CONV_MODE_SPAN -- the layout belongs to the plan)
 * and builds the gather table ktab[kpad] (host pointers, sizes from the plan).  The caller
 * uploads both. */
int eco_conv_pack_weights(const eco_conv_geom* g, const eco_conv_plan* plan,
                          const float* w, float* wp, int32_t* ktab);
/* ConvolutionLayer::Forward_gpu (layers/conv_layer.cu, cudnn_conv_layer.cu:15-65) as one
 * implicit-GEMM MFMA kernel, optionally fused with the BN / ReLU / Eltwise / Concat /
 * Permute layers that follow it.  x, wp, ktab and all epilogue pointers are DEVICE pointers.
 * `workspace` is caller-owned device scratch of at least plan->ws_bytes (may be NULL when that is 0;
 * one buffer can serve every layer launched on the same stream).  When plan->ksplit > 1 a second,
 * deterministic reduce launch follows on the same stream. */
int eco_conv_forward(const eco_conv_geom* g, const eco_conv_plan* plan, const float* x,
                     const float* wp, const int32_t* ktab, const eco_conv_epilogue* ep,
                     void* workspace, void* stream);
/* `batch` independent convolutions of one geometry / plan in a single launch (gridDim.y = batch):
 * entry b reads x + b*stride_x, weights wp + b*stride_wp (same gather table), and writes through
 * the epilogue's views moved by b*stride_out elements; `workspace` holds batch * plan.ws_bytes.
 * Used for the (M+2)^2 transform points of the Winograd path (36 for F(4x4,3x3), 16 for F(2x2,3x3)).  Gather-kernel plans only. */
int eco_conv_forward_batched(const eco_conv_geom* g, const eco_conv_plan* plan, const float* x,
                             const float* wp, const int32_t* ktab, const eco_conv_epilogue* ep,
                             void* workspace, int32_t batch, int64_t stride_x, int64_t stride_wp,
                             int64_t stride_out, void* stream);

/* ---- Winograd F(MxM,3x3) path, M = tile_m in {2, 4}, for stride-1, pad-1 (kd)x3x3 convolutions
 * (csrc/eco_wino.hip).  Same result as eco_conv_forward up to fp32 rounding (cudnn_conv_layer.cu:15-65
 * leaves the algorithm to cuDNN, which picks Winograd for such shapes too).  T = M + 2 transform points per
 * dimension; tiles TH = ceil(H/M), TW = ceil(W/M).
 *   u = eco_wino_weight_transform(w)             HOST: u[T*T][cout][cin][kd] = G g G^T per (co, ci, z)
 *   eco_wino_input_forward(x -> v)               v[T*T][planes][TH][TW], planes = n*cin*d
 *   eco_conv_forward_batched(v -> m, batch=T*T)  geometry n, cin->cout, in (d,TH,TW), kernel (kd,1,1),
 *                                                pad (kd/2,0,0); weights = pack(u[p]) per point; raw-only epilogue
 *   eco_wino_output_forward(m -> y)              y tile = A^T m A, then the fused epilogue `ep` */
int eco_wino_weight_transform(const float* w, int32_t cout, int32_t cin, int32_t kd, int32_t tile_m, float* u);
int eco_wino_input_forward(const float* x, float* v, int64_t planes, int32_t h, int32_t w, int32_t tile_m,
                           void* stream);
int eco_wino_output_forward(const float* m, int32_t n, int32_t cout, int32_t d, int32_t h, int32_t w,
                            int32_t tile_m, const eco_conv_epilogue* ep, void* stream);

/* ---- stand-alone operators (one per reference layer type) -------------------------- */

/* PoolingLayer::Forward_gpu (layers/pooling_layer.cu:12-81; N-D via cuDNN,
 * layers/cudnn_pooling_layer.cu:13-22).  x: [n,c,in...] -> y: [n,c,out...]; nsp spatial
 * dims (1..3, leading entries of the arrays unused when nsp<3 are given as size 1).
 * MAX clips windows to the image; AVE divides by the window size including padding
 * clipped to in+pad (pooling_layer.cpp:199-262).  out[] must equal the ceil-rule
 * pooled shape (pooling_layer.cpp:131-147). */
typedef struct eco_pool_geom {
  int32_t n, c;
  int32_t in[3], kernel[3], stride[3], pad[3], out[3];
  int32_t method; /* ECO_POOL_MAX | ECO_POOL_AVE */
} eco_pool_geom;
int eco_pool_forward(const eco_pool_geom* g, const float* x, float* y, void* stream);
/* The same with y as a channel slice of a wider tensor -- a Concat top (concat_layer.cpp:60-81 would copy the pooled blob
 * there; inception_3c_pool / inception_4e_pool of ECO-Full, models_ECO_Full/kinetics/deploy.prototxt:1960-1990,3200-3230):
 * y points at the slice's first channel of image 0, images are y_image_stride floats apart (0 = dense, as
 * eco_pool_forward), the c channels of an image contiguous.  (v18) */
int eco_pool_forward_strided(const eco_pool_geom* g, const float* x, float* y, int64_t y_image_stride, void* stream);

/* AVE pooling 3x3 / stride 1 / pad 1 (divisor 9 everywhere: pooling_layer.cpp:247-262) of x[n,c,h,w], followed by
 * y = relu ? max(a, 0) : a with a = (avg + bias[c]) * bn_scale[c] + bn_shift[c], written through the strided view
 * `dst` (e.g. a channel slice of a Concat top).  bias / bn_scale+bn_shift may be NULL.  With the 1x1 convolution that
 * follows such a pool in an Inception block applied to the pool's INPUT instead (both maps are linear and the pool's
 * coefficients constant, so they commute), this kernel finishes  pool -> conv -> BN -> ReLU
 * (models_ECO_Lite/kinetics/deploy.prototxt:330-400; pooling_layer.cpp, conv_layer.cpp:28-43, bn_layer.cpp:93-207)
 * on cout instead of cin channels.  x must be a dense N,C,H,W tensor (planes contiguous, no strides); 16-byte accesses are
 * used when x, the view's base and w allow it (w % 4 == 0 and aligned pointers), 4-byte ones otherwise. */
int eco_avgpool_affine_forward(const float* x, const float* bias, const float* bn_scale, const float* bn_shift,
                               int32_t relu, const eco_view* dst, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/* BNLayer TEST/frozen forward with folded statistics (+ optional in-place ReLU):
 * y = x*scale[c] + shift[c]; relu -> max(y,0).  x,y: [n,c,inner]; y may alias x.
 * (layers/bn_layer.cu:12-125, cudnn_bn_layer.cu:17-37, relu_layer.cu:10-15) */
int eco_bn_forward(const float* x, float* y, const float* scale, const float* shift,
                   int64_t n, int64_t c, int64_t inner, int relu, void* stream);
/* ReLULayer::Forward_gpu (layers/relu_layer.cu:10-15): y = max(x,0) + slope*min(x,0). */
int eco_relu_forward(const float* x, float* y, int64_t count, float negative_slope, void* stream);
/* EltwiseLayer SUM of two bottoms: y = ca*a + cb*b (layers/eltwise_layer.cu:49-53). */
int eco_eltwise_sum_forward(const float* a, const float* b, float* y, int64_t count,
                            float ca, float cb, void* stream);
/* ConcatLayer::Forward_gpu for one bottom (layers/concat_layer.cu:10-46): copies
 * x[outer][cx][inner] into y[outer][cy][inner] at channel offset c0. */
int eco_concat_copy(const float* x, float* y, int64_t outer, int64_t cx, int64_t cy, int64_t c0,
                    int64_t inner, void* stream);
/* PermuteLayer::Forward_gpu (layers/permute_layer.cu:11-50): y[i0..] = x permuted,
 * top axis k = bottom axis order[k]; naxes <= 6. */
int eco_permute_forward(const float* x, float* y, int32_t naxes, const int32_t* in_shape,
                        const int32_t* order, void* stream);
/* InnerProductLayer::Forward_gpu (layers/inner_product_layer.cu:14-25):
 * y[m][n] = sum_k x[m][k]*w[n][k] + bias[n] (bias may be NULL). */
int eco_inner_product_forward(const float* x, const float* w, const float* bias, float* y,
                              int64_t m, int64_t n, int64_t k, void* stream);
/* Fused tail global_pool (AVE over the whole D*H*W volume) -> reshape -> dropout(TEST) -> fc:
 * y[b][o] = bias[o] + sum_c w[o][c0 + c] * mean_s x[b][c][s]   (c in [0,c), w row length wk).
 * accumulate != 0 adds into y instead of overwriting (used for the ECO-Full concat+fc8N split).
 * (cudnn_pooling_layer.cu:13-22 + reshape_layer.cpp:88 + dropout_layer.cpp:46-48 +
 *  inner_product_layer.cu:14-25) */
int eco_global_avgpool_fc_forward(const float* x, const float* w, const float* bias, float* y,
                                  int64_t b, int64_t c, int64_t s, int64_t n_out, int64_t wk,
                                  int64_t c0, int accumulate, void* stream);
/* The same with the clip's features spread over t consecutive images, x[b*t + f][c][s]: the mean runs over the
 * t*s values of a channel.  That is ECO-Full's 2-D stream tail in one launch -- global_pool2D (AVE over the
 * 7x7 plane) -> dropout -> reshape [-1,1,T,C] -> segment_consensus (AVE over T) -> reshape -> its columns of
 * fc8N (models_ECO_Full/kinetics/deploy.prototxt:4607-4690); with accumulate it adds into the logits the 3-D
 * stream's call left in y. */
int eco_global_avgpool_fc_seg_forward(const float* x, const float* w, const float* bias, float* y,
                                      int64_t b, int64_t t, int64_t c, int64_t s, int64_t n_out, int64_t wk,
                                      int64_t c0, int accumulate, void* stream);
/* GPU-side input stage = the VideoData TEST-phase output contract
 * (layers/video_data_layer.cpp:107-119, util/io.cpp:368-421 ReadSegmentRGBToDatum, DataTransformer::Transform
 * data_transformer.cpp:147-330): decoded frames arrive as uint8 H x W x 3 interleaved (OpenCV BGR order, channel
 * index c = 0,1,2 kept as is), one after another; each is cropped to crop_h x crop_w at (h_off, w_off)
 * (TEST: centre crop, h_off=(H-crop)/2), optionally mirrored in w, converted to planar fp32
 *   y[f][c][h][w] = (float(frames[f][h_off+h][w_off+w][c]) - mean[c]) * scale
 * which is exactly the [B*N, 3, crop, crop] blob the deploy net takes as `data`.  Shipping uint8 frames
 * and converting on the GPU moves 4x fewer bytes over PCIe than fp32 frames. */
int eco_video_input_forward(const uint8_t* frames, float* y, int64_t num_frames, int32_t height, int32_t width,
                            int32_t crop_h, int32_t crop_w, int32_t h_off, int32_t w_off, const float mean[3],
                            float scale, int32_t mirror, void* stream);
/* SoftmaxLayer::Forward_gpu over axis 1 of [outer, c, inner] (layers/softmax_layer.cu:14-71). */
int eco_softmax_forward(const float* x, float* y, int64_t outer, int64_t c, int64_t inner,
                        void* stream);
/* AccuracyLayer::Forward (layers/accuracy_layer.cpp:46-92; TEST-phase `top1` / `top5` of
 * the train/val prototxts): scores [outer, c, inner], labels [outer*inner] (float-coded class
 * ids, as the reference's label blob).  *out (device, the layer's 0-axis top) = fraction of the
 * counted samples whose label is in the top_k of std::greater<pair<score, class>>. */
int eco_accuracy_forward(const float* x, const float* label, float* out, int64_t outer, int64_t c,
                         int64_t inner, int32_t top_k, int32_t has_ignore_label, int32_t ignore_label,
                         void* stream);
/* SoftmaxWithLossLayer::Forward (layers/softmax_loss_layer.cpp:52-84; TEST-phase `loss`):
 * *out = -sum log(max(softmax(x)[label], FLT_MIN)) / (normalize ? counted : outer). */
int eco_softmax_loss_forward(const float* x, const float* label, float* out, int64_t outer, int64_t c,
                             int64_t inner, int32_t normalize, int32_t has_ignore_label,
                             int32_t ignore_label, void* stream);

/* ---- the BN-Inception stem as one launch (csrc/eco_stem.hip) ------------------------------------------------
 * conv1_7x7_s2 (3 -> cout in {32, 64}, 7x7, stride 2, pad 3) + bias + folded BN + ReLU + pool1_3x3_s2 (MAX 3x3,
 * stride 2, ceil rule): models_ECO_Lite/kinetics/deploy.prototxt:8-77; conv_layer.cpp:28-43, bn_layer.cpp:93-207,
 * relu_layer.cpp:10-20, pooling_layer.cpp:131-147,199-237.  x: [n,3,h,w] fp32 -> y: [n,cout,PH,PW] with
 * HO = (h-1)/2+1, PH = ceil((HO-3)/2)+1 (same for w); conv1's own output is never written. */
/* HOST: w[cout][3][7][7] -> wp[148][cout] (k-major, k = c*49 + ky*7 + kx; row 147 zero).  The kernel multiplies the
 * folded-BN scale into its LDS copy of the weights and starts the accumulators from bias*scale + shift. */
int eco_stem_pack_weights(const float* w, int32_t cout, float* wp);
/* max_workgroups: 0 = two persistent workgroups per compute unit of the current device. */
int eco_stem_forward(const float* x, const float* wp, const float* bias, const float* bn_scale, const float* bn_shift,
                     int32_t relu, float* y, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t max_workgroups,
                     void* stream);

/* ---- the same stem for the channel-blocked bf16 path (csrc/eco_stemb.hip, ABI v16) ----------------------------
 * x: [n,3,h,w] fp32 frames -> y: [n][cout/8][PH][PW][8] bf16 (the blocked layout of eco_convb_forward); operands
 * rounded to bf16 (nearest even), fp32 products / sums / bias / BN, one rounding at the store.  Replaces the three
 * launches eco_stem_pack_forward + eco_convb_forward (stem plan) + eco_poolb_forward. */
int64_t eco_stemb_weight_elems(int32_t cout);   /* bf16 elements of the packed weights: 11 * 2 * cout * 8 */
/* HOST: w[cout][3][7][7] -> wp[s][g][cout][8] bf16: kernel row rho = c*7 + ky = 2s + g, tap kx = e (kx = 7 and row 21
 * zero): one v_mfma_f32_32x32x16_bf16 k-step = two kernel rows. */
int eco_stemb_pack_weights(const float* w, int32_t cout, void* wp);
int eco_stemb_forward(const float* x, const void* wp, const float* bias, const float* bn_scale, const float* bn_shift,
                      int32_t relu, void* y, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t max_workgroups,
                      void* stream);

/* ---- Winograd F(4x4,3x3) route on a dedicated transformed-domain GEMM (csrc/eco_wgemm.hip) -------------------
 *
 * The same three steps as the eco_wino_* / eco_conv_forward_batched route above for tile_m = 4, with layouts chosen
 * for a dense GEMM kernel (operands staged by LDS-DMA, no address decode, no tap masks):
 *   v[p][cin/2][d + 2*(kd/2)][r][2]   channel pairs interleaved, positions r = (b*TH + th)*TW + tw depth-major,
 *                                     one all-zero plane at either end of the depth axis when kd = 3
 *   up = eco_wgemm_pack_weights(u)    u = eco_wino_weight_transform(w, tile_m = 4) -> [p][mblock][stage][8][bmp][2]
 *   m[p][slice][cout][d][r]           raw products; split-K slices are summed by the output transform
 *   eco_wino_input_pk_forward(x -> v); eco_wgemm_forward(v, up -> m); eco_wino_output_dm_forward(m -> y, epilogue)
 * Same results as eco_conv_forward up to fp32 rounding (cudnn_conv_layer.cu:15-65 leaves the algorithm to cuDNN). */
typedef struct eco_wgemm_plan {
  int32_t n, cin, cout, d, th, tw; /* clips/images, channels, depth, tiles per plane (ceil(H/4), ceil(W/4))   */
  int32_t kd;                      /* 1 (2-D 3x3) or 3 (3x3x3, depth taps direct)                             */
  int32_t points;                  /* 36; 216 = F(4x4x4,3x3x3), 320 / 64 = the stride-2 forms below (kd = 1)        */
  int32_t bm, bn;                  /* block tile: output channels x positions                                 */
  int32_t nstages;                 /* (cin/16) * kd stages of 16 reduction elements                           */
  int32_t ksplit;                  /* split-K slices (rows of m)                                              */
  int32_t mblocks, bmp;            /* ceil(cout/bm); bm rounded up to 64 (rows of a packed weight block)      */
  int64_t q;                       /* positions per channel-pair row of v: (d + 2*(kd/2)) * n*th*tw           */
  int64_t u_elems, v_elems, m_elems; /* floats in up / v (with read slack) / m                                */
} eco_wgemm_plan;
/* cin % 16 == 0.  num_cu = 0 plans for 256 compute units. */
int eco_wgemm_plan_create(int32_t n, int32_t cin, int32_t cout, int32_t d, int32_t th, int32_t tw, int32_t kd,
                          int32_t points, int32_t num_cu, eco_wgemm_plan* plan);
/* HOST function: u[p][cout][cin][kd] (eco_wino_weight_transform) -> up (plan->u_elems floats). */
int eco_wgemm_pack_weights(const eco_wgemm_plan* plan, const float* u, float* up);
int eco_wino_input_pk_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t h, int32_t w, void* stream);
int eco_wgemm_forward(const eco_wgemm_plan* plan, const float* v, const float* up, float* m, void* stream);
int eco_wino_output_dm_forward(const eco_wgemm_plan* plan, const float* m, int32_t h, int32_t w,
                               const eco_conv_epilogue* ep, void* stream);

/* Fused form of the last two steps for the 2-D layers (kd = 1, d = 1, cin = 64..224 and cout multiples of 32:
 * conv2_3x3 and the inception 3x3 convs, models_ECO_Lite/kinetics/deploy.prototxt:78-330): the 36 transformed-domain
 * products of a 32-channel x 32-tile block stay in LDS and are output-transformed there; M never goes to HBM.
 * V comes from eco_wino_input_q4_forward (the same transform, four k-pairs of a position as one 16-byte vector:
 * V4[36][cin/8][tiles][2][4], plan->v_elems floats suffice), the epilogue is eco_wino_output_dm_forward's; weights
 * are packed by eco_wfused_pack_weights from u[36][cout][cin] (eco_wino_weight_transform) into
 * eco_wfused_weight_elems floats. */
int eco_wino_input_q4_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t h, int32_t w, void* stream);
int64_t eco_wfused_weight_elems(const eco_wgemm_plan* plan);
int eco_wfused_pack_weights(const eco_wgemm_plan* plan, const float* u, float* up); /* HOST */
int eco_wfused_forward(const eco_wgemm_plan* plan, const float* v, const float* up, int32_t h, int32_t w,
                       const eco_conv_epilogue* ep, void* stream);
/* The same with the MAX 3x3 stride-2 unpadded Pooling that consumes the activated output folded in (conv2_3x3 + BN + ReLU ->
 * pool2, models_ECO_Lite/kinetics/deploy.prototxt:103-128; pooling_layer.cpp:131-147,199-225): every 4x4 output tile leaves
 * as the 3x3 partial maxima it contributes (scratch: eco_wfused_pool_scratch_elems floats, [n][cout][3 th][3 tw]) and a
 * second, small launch folds neighbouring tiles into y[n][cout][2 th][2 tw]; the convolution's own output is never written.
 * h and w must be multiples of 4; `ep` supplies bias / folded BN / relu only (every view must be null).  (v18) */
int64_t eco_wfused_pool_scratch_elems(const eco_wgemm_plan* plan);
int eco_wfused_pool_forward(const eco_wgemm_plan* plan, const float* v, const float* up, int32_t h, int32_t w,
                            const eco_conv_epilogue* ep, float* scratch, float* y, void* stream);

/* ---- Winograd F(4x4x4,3x3x3) for the 3-D trunk (csrc/eco_wino3.hip, ABI v18) ---------------------------------
 *
 * The stride-1 pad-1 3x3x3 convolutions (res3a_2 ... res5b_2, models_ECO_Lite/kinetics/deploy.prototxt:1162-1660)
 * with the minimal-filtering algorithm nested over depth as well: 216 multiplies per input channel and 4x4x4 output
 * tile instead of the 432 of the F(4x4,3x3) + direct-depth-taps route above; the depth axis of every trunk stage is
 * a multiple of 4 at num_segments 16 / 32, other depths are padded per tile.  Same three steps, same GEMM kernel:
 *   plan = eco_wgemm_plan_create(n, cin, cout, ceil(D/4), ceil(H/4), ceil(W/4), kd = 1, points = 216, ...)
 *   v[p][cin/2][r][2], m[p][slice][cout][r]   p = (pz*6 + py)*6 + px, r = ((td*n + b)*th + ty)*tw + tx
 *   up = eco_wgemm_pack_weights(u), u = eco_wino3_weight_transform(w)       u[216][cout][cin] = (G x G x G) w
 *   eco_wino3_input_forward(x -> v); eco_wgemm_forward(v, up -> m); eco_wino3_output_forward(m -> y, epilogue)
 * The two transforms keep a group of images' planes in LDS: eco_wino3_lds_bytes(n, th, tw) must not exceed 152 KB
 * (planes up to 52x52: 48 (4 th + 2)(4 tw + 8) bytes for one image; 56x56 needs 174 KB) -- ask eco_wino3_lds_bytes, the
 * entry points fail with ECO_ERR_INVALID otherwise.  Results equal eco_conv_forward's up to fp32
 * rounding (cudnn_conv_layer.cu:15-65 leaves the algorithm to cuDNN). */
int eco_wino3_weight_transform(const float* w, int32_t cout, int32_t cin, float* u); /* HOST */
int64_t eco_wino3_lds_bytes(int32_t n, int32_t th, int32_t tw);
int eco_wino3_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t d, int32_t h, int32_t w,
                            void* stream);
int eco_wino3_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t d, int32_t h, int32_t w,
                             const eco_conv_epilogue* ep, void* stream);

/* ---- STRIDE-2 3x3x3 convolutions as polyphase minimal-filtering problems (csrc/eco_wino_s2.hip, ABI v19) ------
 *
 * res4a_1 / res4a_down (stride 2, pad 1, 3x3x3; models_ECO_Lite/kinetics/deploy.prototxt:1262-1330).  Per axis
 * out[o] = (w0, w2) * x_odd[o-1 .. o] + (0, w1) * x_even[o-1 .. o]: eight stride-1 two-tap problems on the even / odd
 * sub-lattices of the input, F(4,2) over depth and F(7,2) over rows and columns, all on the same 5 x 8 x 8 = 320
 * transform points, so that the phases accumulate in the transformed domain like input channels (K = 8 cin):
 * 13.1 multiplies per output and input channel instead of 27, no tile overhang where the OUTPUT volume tiles by
 * 4 x 7 x 7 (8 x 14 x 14 at num_segments 16) and the input volume is exactly twice the output volume.
 *   plan = eco_wgemm_plan_create(n, 8 * cin, ctot, Do/4, Ho/7, Wo/7, kd = 1, points = 320, ...)
 *   v[p][k/2][r][2], m[p][slice][ctot][r]    p = (az*8 + ay)*8 + ax, k = ((c*2 + fz)*2 + fy)*2 + fx (f = 1: odd phase),
 *                                            r = ((b*td + tz)*th + ty)*tw + tx
 *   up = eco_wgemm_pack_weights(u), u = eco_wino_s2_weight_transform(w)      u[320][ctot][8 cin]
 *   eco_wino_s2_input_forward(x -> v); eco_wgemm_forward(v, up -> m); eco_wino_s2_output_forward(m -> y, epilogue)
 * Convolutions of one geometry that read the same blob (a residual block's first conv and its projection shortcut)
 * concatenate their weights along cout (ctot rows), share v and the GEMM, and take one output-transform launch each:
 * channels [c0, c0 + cout) of m with the member's own epilogue.  eco_wino_s2_lds_bytes(n, td, th, tw) must not exceed
 * 152 KB (input planes up to ~56 x 56).  Results equal eco_conv_forward's up to fp32 rounding, ~1e-4 of the largest
 * output per layer (cudnn_conv_layer.cu:15-65 leaves the algorithm to cuDNN). */
int eco_wino_s2_weight_transform(const float* w, int32_t cout, int32_t cin, float* u); /* HOST */
int64_t eco_wino_s2_lds_bytes(int32_t n, int32_t td, int32_t th, int32_t tw);
/* (d, h, w): the INPUT volume (= twice the output volume) */
int eco_wino_s2_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t d, int32_t h, int32_t w,
                              void* stream);
/* (od, oh, ow): the OUTPUT volume; ep->nseg must be 0 */
int eco_wino_s2_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t c0, int32_t cout, int32_t od, int32_t oh,
                               int32_t ow, const eco_conv_epilogue* ep, void* stream);
/* The 2-D form of the same idea, for output volumes with too few 4 x 7 x 7 tiles (res5a_1 / res5a_down: 4 x 7 x 7 outputs) and
 * for the strided 2-D 3x3 convolutions of ECO-Full (inception_3c / 4e, models_ECO_Full/kinetics/deploy.prototxt:1854-1990,
 * 3420-3560): F(7,2) x F(7,2) over rows and columns on 8 x 8 = 64 points, the depth taps direct and part of the reduction.
 * kz = 3: a 3x3x3 kernel, stride 2 / pad 1 on all three axes (input depth = 2 od); kz = 1: a (1x)3x3 kernel, stride (1,)2,2,
 * pad (0,)1,1 (input depth = od; od = 1 for 2-D blobs).  Every output plane is a position of its own:
 *   plan = eco_wgemm_plan_create(n, 4 * kz * cin, ctot, od, oh/7, ow/7, kd = 1, points = 64, ...)
 *   v[p][k/2][r][2], m[p][slice][ctot][r]    p = ay*8 + ax, k = ((c*kz + tz)*2 + fy)*2 + fx, r = ((b*od + z)*th + ty)*tw + tx
 *   u = eco_wino_s2d_weight_transform(w[cout][cin][kz][3][3], ...)             u[64][ctot][4 kz cin]
 * and the same three steps.  eco_wino_s2d_lds_bytes(n, kz, od, th, tw) <= 152 KB. */
int eco_wino_s2d_weight_transform(const float* w, int32_t cout, int32_t cin, int32_t kz, float* u); /* HOST */
int64_t eco_wino_s2d_lds_bytes(int32_t n, int32_t kz, int32_t od, int32_t th, int32_t tw);
int eco_wino_s2d_input_forward(const eco_wgemm_plan* plan, const float* x, float* v, int32_t kz, int32_t d, int32_t h,
                               int32_t w, void* stream);
int eco_wino_s2d_output_forward(const eco_wgemm_plan* plan, const float* m, int32_t c0, int32_t cout, int32_t od, int32_t oh,
                                int32_t ow, const eco_conv_epilogue* ep, void* stream);

/* ---- channel-blocked ("NC8") path on the bf16 matrix cores (csrc/eco_blocked.hip) -------------------------
 *
 * BASELINE.json configs[4] (ECO-Lite, bf16).  Activations are kept as X[n][c/8][d][h][w][c%8] so that the eight
 * reduction elements a lane feeds v_mfma_f32_32x32x16_bf16 are one 16-byte vector; the layout is internal to
 * this path (the `data` input is the reference's fp32 N,3,H,W blob, the logits leave as fp32 [B, classes]).
 * Same operators, epilogue algebra and error conventions as the fp32 entry points above; in every
 * eco_view / eco_conv_epilogue handed to these functions `ptr` addresses elements of the storage type and the
 * strides count 8-channel blocks:  block(img, c, sp) = (img / t)*stride_b + (img % t)*stride_t + (c/8)*stride_c + sp.
 * bias / bn_scale / bn_shift / fc weights / logits are fp32. */
#define ECO_DT_BF16 1  /* bf16 storage; products of bf16 operands accumulated in fp32                         */
/* (ECO_DT_F32X3 = 3 -- fp32 storage, every operand split exactly into three bf16 terms -- existed until v18: 27.6 ms per
 * configs[1] step against 17.3 ms on the fp32 MFMA, never a reported configuration; removed in v19 with its kernel) */

typedef struct eco_convb_plan {
  int32_t bm, bn;    /* block tile: output channels x output positions                                  */
  int32_t dt;        /* ECO_DT_*                                                                         */
  int32_t stem;      /* 1: the 3-channel 7x7 stride-2 pad-3 stem; x is the image of eco_stem_pack_forward */
  int32_t cblocks;   /* input channel blocks reduced over (cin/8; 4 for the stem)                        */
  int32_t nstages;   /* reduction stages of 32 elements: ceil(cblocks/4) * taps (zero-padded weights)       */
  int32_t mpad;      /* cout rounded up to a multiple of bm                                              */
  int32_t ksplit;    /* > 1: reduction cut into ksplit slices, summed by a second deterministic launch   */
  int32_t span_pieces; /* > 0: stride-1 same-size (kd)x3x3 span kernel; 64-position DMA pieces per staged span */
  int32_t pgrid;     /* > 0 (span plans, v17): workgroups of the persistent span kernel, 2 per compute unit,   */
                     /* a multiple of 8 XCDs x the M-blocks of a position tile; 0 = one workgroup per tile   */
  int64_t wp_vecs;   /* 16-byte vectors in the packed weights: nstages * 4 * mpad                          */
  int64_t ws_bytes;  /* device scratch eco_convb_forward needs (0: none)                                 */
  int32_t tail_tiles;  /* > 0 (persistent span plans with ksplit == 1, v17): the last tail_tiles tiles are   */
  int32_t tail_ksplit; /* cut into tail_ksplit slices of their reduction (partial sums + a reduce launch     */
                       /* over their positions only), so that a partial last round per CU becomes a short one */
} eco_convb_plan;

/* Requirements: cout % 8 == 0 and cin % 8 == 0, or the stem geometry (cin 3, 2-D 7x7, stride 2, pad 3, even W).
 * num_cu = 0 sizes the plan for 256 compute units. */
int eco_convb_plan_create(const eco_conv_geom* g, int32_t dt, int32_t num_cu, eco_convb_plan* plan);
/* HOST function: caffe weights w[cout][cin][kd][kh][kw] (fp32) -> wp[stage][4][mpad][8] bf16
 * (stage = channel-group*taps + tap). */
int eco_convb_pack_weights(const eco_conv_geom* g, const eco_convb_plan* plan, const float* w, void* wp);
/* ConvolutionLayer::Forward_gpu (+ fused BN / ReLU / Eltwise / Concat / Reshape+Permute) on blocked tensors. */
int eco_convb_forward(const eco_conv_geom* g, const eco_convb_plan* plan, const void* x, const void* wp,
                      const eco_conv_epilogue* ep, void* workspace, void* stream);
/* fp32 N,3,H,W frames (the VideoData contract, video_data_layer.cpp:107-119) -> the zero-padded pixel-interleaved
 * image the stem plan reads: y[f][h+3][w+3][4] in the storage type, (H+6) x (W+8) pixels per frame. */
int eco_stem_pack_forward(const float* x, void* y, int64_t frames, int32_t h, int32_t w, int32_t dt, void* stream);
/* PoolingLayer::Forward_gpu on a blocked tensor x[n][c/8][in...][8] -> y[n][c/8][out...][8] (rules of eco_pool_forward). */
int eco_poolb_forward(const eco_pool_geom* g, int32_t dt, const void* x, void* y, void* stream);
/* The AVE 3x3 / stride 1 / pad 1 pool behind its 1x1 projection on the blocked path (ABI v16; the fp32 form is
 * eco_avgpool_affine_forward): x = the projection's raw products [n][c/8][h][w][8] without bias -> window sum / 9 + bias,
 * folded BN, ReLU -> dst (a blocked view: strides in 8-channel vectors, as every view of eco_convb_forward). */
int eco_poolb_avg_affine_forward(int32_t dt, const void* x, const float* bias, const float* bn_scale,
                                 const float* bn_shift, int32_t relu, const eco_view* dst, int64_t n, int32_t c, int32_t h,
                                 int32_t w, void* stream);
/* eco_global_avgpool_fc_forward on a blocked volume x[b][c/8][s][8]. */
int eco_global_avgpool_fc_b_forward(const void* x, int32_t dt, const float* w, const float* bias, float* y,
                                    int64_t b, int64_t c, int64_t s, int64_t n_out, int64_t wk, int64_t c0,
                                    int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECO_HIP_H_ */
