/* include/omni3d_hip.h -- C ABI of libomni3d_hip.so (gfx950 / MI355X).
 *
 * Every entry point takes raw DEVICE pointers, explicit sizes and a hipStream_t (passed as
 * void*), returns an int status (0 = OMNI_OK, 1 = bad argument, 2 = launch failure), never
 * allocates and never throws.  No torch types appear in any signature.  Each declaration cites
 * the reference interface it replaces (paths relative to the facebookresearch/omni3d checkout).
 * The reference itself has no FFI boundary (it is pure Python over detectron2 / torchvision /
 * pytorch3d); INTEGRATION.md shows the ctypes stub a maintainer would add per entry point.
 */
#ifndef OMNI3D_HIP_H
#define OMNI3D_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- IoU3D (evaluation) */

/* pytorch3d._C.iou_box3d(boxes1, boxes2) as called at
 * cubercnn/evaluation/omni3d_evaluation.py:155.  boxes1 (N,8,3), boxes2 (M,8,3) fp32 corner
 * lists in the order documented at omni3d_evaluation.py:117-142.  Writes vol (N,M) [nullable]
 * and iou (N,M).  valid1 [nullable] is an int32 (N) mask: rows with valid1[i]==0 are written as
 * zeros without being computed (box3d_overlap, omni3d_evaluation.py:151-164).  overflow
 * [nullable] is an int32 counter incremented when a pair exceeded the LDS triangle capacity. */
int omni_iou_box3d(const float* boxes1, int N, const float* boxes2, int M, const int* valid1, float* vol,
                   float* iou, int* overflow, void* stream);

/* Paired / ragged form of the same computation, for the evaluator's per-(image, category)
 * groups (Omni3Deval.computeIoU, omni3d_evaluation.py:1359-1431): pair p compares
 * boxes1[idx1[p]] with boxes2[idx2[p]]; vol [nullable] and iou have npairs entries. */
int omni_iou_box3d_pairs(const float* boxes1, const float* boxes2, const int* idx1, const int* idx2,
                         long long npairs, const int* valid1, float* vol, float* iou, int* overflow,
                         void* stream);

/* The same with the launch variant chosen by the caller: lanes_per_pair 64 | 32 | 16 = 1 | 2 | 4 pairs per wavefront over
 * full-capacity (160-triangle) LDS lists; 1000 + lanes = first pass over 96-triangle lists (more waves resident per CU), pairs
 * that outgrow them are marked and recomputed at full capacity by a second kernel of the same call; 0 = the production choice
 * (1032).  Every variant computes the same per-pair algorithm (same triangle order, same epsilon decisions); used by the parity
 * tests and by tools/bench_iou3d.py.  Replaces the same call site as above. */
int omni_iou_box3d_pairs_algo(const float* boxes1, const float* boxes2, const int* idx1, const int* idx2,
                              long long npairs, const int* valid1, float* vol, float* iou, int* overflow,
                              int lanes_per_pair, void* stream);

/* _check_coplanar (omni3d_evaluation.py:65-86) and _check_nonzero (:89-104) fused:
 * valid[i] = coplanar(i) && nonzero(i); counts [nullable, int32[2]] += {#non-coplanar, #zero}. */
int omni_box3d_validity(const float* boxes, int N, float eps_coplanar, float eps_nonzero, int* valid,
                        int* counts, void* stream);

/* ------------------------------------------------- convolution / linear (fp32 MFMA, NHWC) */

/* torch.nn.Conv2d forward as used by the DLA-34 bottom-up (cubercnn/modeling/backbone/dla.py:
 * 43-51,159-161,241-245,291-295), detectron2 FPN (dla.py:500-506) and StandardRPNHead
 * (configs/Base.yaml:49); and torch.nn.Linear (H=W=R=S=1: FastRCNNConvFCHead,
 * cubercnn/modeling/roi_heads/cube_head.py:70,108-144).
 * x is NHWC fp32 (N,H,W,C) with pixel pitch ldx floats, w is KRSC fp32, bias [nullable] (K),
 * out NHWC (N,OH,OW,K) with pitch ldo.  C and ldx must be multiples of 4.  relu != 0 fuses ReLU. */
int omni_conv2d_fwd(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C,
                    int K, int R, int S, int stride, int pad, int ldx, int ldo, int relu, void* stream);

/* grad wrt the input of the same convolution (autograd of the call sites above):
 * dx (N,H,W,C) pitch lddx (=|+= when accumulate) from dy (N,OH,OW,K) pitch lddy. K % 4 == 0. */
int omni_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int lddy, int lddx, int accumulate, void* stream);

/* grad wrt the weights: dw (K,R,S,C) overwritten, or atomically accumulated into when accumulate != 0 (weight
 * gradients land directly in the flat gradient bucket).  Split-K over output pixels, fp32 atomics. */
int omni_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int ldx, int lddy, int accumulate, void* stream);

/* The same three operations with the ALGORITHM chosen by the caller instead of by the launcher's shape heuristics
 * (cuDNN's "algo" argument; the reference reaches it through torch.backends.cudnn.benchmark, tools/train_net.py sets
 * nothing and takes the default).  tile: 0 = automatic | 1 = 128x128 | 2 = 64x64 | 3 = 128x64 | 4 = 256x32 (fwd/dgrad) or
 * 32x128 (wgrad).  splits: 0 = automatic | >= 1 = reduction splits (> 1 ends with fp32 atomics into a zeroed, dense output;
 * dgrad with accumulate != 0: on top of dx's content, any pitch -- the gradient fan-in target of functional.fanout).
 * omni_conv2d_fwd/dgrad/wgrad == the _algo form with tile = splits = 0.  Used by tools/bench_kernels.py and by tests that
 * exercise a given tile shape on a small problem; there is no process-global tuning state. */
int omni_conv2d_fwd_algo(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C,
                         int K, int R, int S, int stride, int pad, int ldx, int ldo, int relu, int tile, int splits,
                         void* stream);
int omni_conv2d_dgrad_algo(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R,
                           int S, int stride, int pad, int lddy, int lddx, int accumulate, int tile, int splits,
                           void* stream);
int omni_conv2d_wgrad_algo(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R,
                           int S, int stride, int pad, int ldx, int lddy, int accumulate, int tile, void* stream);

/* Deterministic forms (round 4; csrc/split_reduce.h).  The reference's PyTorch-CPU path gives bit-identical results from run to
 * run; a split reduction that meets in the output through fp32 atomics does not (the order of the adds follows the order in which
 * workgroups finish).  These entry points are the three operations above with the reduction splits meeting in a caller-provided
 * workspace instead: every split stores its partial tile to `ws` (ws_floats floats), arrives at one of `ctr`'s n_ctr unsigned
 * counters (ZERO on entry, zero again on exit), and the tile's last-arriving workgroup adds the partial tiles in split order and
 * runs the normal epilogue -- bias / ReLU (fwd), overwrite or add-to-carry (dgrad, accumulate != 0), overwrite or add to the
 * gradient bucket (wgrad, accumulate != 0: plain read-modify-write, each element has one owner per launch).  No zero-fill, no
 * follow-up ReLU pass, and omni_conv2d_fwd_det emits BatchNorm statistics (stats, see omni_conv2d_fwd_stats) from split launches too.
 * plan != NULL: nothing is launched; plan[0..3] = {tile, reduction splits, counters needed (0 = not split), workspace floats
 * needed} for exactly the launch the other arguments describe, so the caller sizes `ws` / `ctr` with the launcher's own heuristics
 * (pass plan[0] / plan[1] back as tile / splits).  Replaces the same torch.nn.Conv2d / nn.Linear forward and backward calls as
 * omni_conv2d_fwd / dgrad / wgrad (dla.py:43-51, cube_head.py:70,108-163). */
int omni_conv2d_fwd_det(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K,
                        int R, int S, int stride, int pad, int ldx, int ldo, int relu, int tile, int splits, float* stats,
                        int stats_rows, int* nblk_out, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan,
                        void* stream);

/* omni_conv2d_fwd_det for an input that is the channel concatenation of nsrc <= 6 dense NHWC tensors xs[s] (N, H, W, cs[s]),
 * cs[s] % 32 == 0, WITHOUT forming the concatenation: conv1x1(torch.cat(children, 1)) of the DLA Root
 * (cubercnn/modeling/backbone/dla.py:166-172).  1 x 1, stride 1, no padding; out (N, H, W, K) with pixel pitch ldo.  Same tiles,
 * splits, epilogues and bit-identical results as the single-tensor entry on the concatenated input.  xs / cs are HOST arrays.
 * ctr == NULL && plan == NULL: non-deterministic form (split reductions meet through atomics). */
int omni_conv2d_fwd_multi_det(const void* const* xs, const int* cs, int nsrc, const float* w, const float* bias, float* out, int N, int H,
                              int W, int K, int ldo, int relu, int tile, int splits_req, float* stats, int stats_rows, int* nblk_out,
                              float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream);
int omni_conv2d_dgrad_det(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int lddy, int lddx, int accumulate, int tile, int splits, float* ws,
                          long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream);
int omni_conv2d_wgrad_det(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int ldx, int lddy, int accumulate, int tile, float* ws, long long ws_floats,
                          int* ctr, int n_ctr, long long* plan, void* stream);

/* OPT-IN EXPERIMENT, round 6 (csrc/gemm_split.hip; SURVEY.md section 7 "3 x bf16 split"): the batched NT GEMM of omni_gemm_batched
 * -- out[b] (M x N) = A[b] (M x K) B[b] (N x K)^T, fp32 in / out, K % 32 == 0 -- on the bf16 matrix cores from an error-free split of
 * the operands into 2 (terms = 3 products) or 3 (terms = 6) bf16 planes, fp32 accumulation.  Never selected by default: the measured
 * path multiplies fp32 operands (v_mfma_f32_32x32x2_f32). */
int omni_gemm_batched_split(const float* A, const float* B, float* out, int batch, int M, int N, int K, int terms, void* stream);
/* Round 6 (csrc/dgrad_s2.hip): data gradient of the 3x3 / stride-2 / pad-1 convolutions that open every DLA level and ResNet stage
 * (cubercnn/modeling/backbone/dla.py:43-51, 186-215; autograd of nn.Conv2d).  dx (N, H, W, C) [pixel pitch lddx] (=, or += when
 * accumulate != 0: a gradient fan-in target) from dy (N, (H-1)/2+1, (W-1)/2+1, K) [pitch lddy] and w (K, 3, 3, C); C, K multiples
 * of 32.  One workgroup computes all four parity classes of a 16 x 16 dx tile from one staged dy tile: dy is read once, every dx
 * element written once by its owner -- deterministic, no atomics, no zero-fill. */
int omni_conv2d_s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int lddy, int lddx,
                         int accumulate, void* stream);
/* Round 6: n <= 64 direct weight gradients (nn.Conv2d backward of the 1 x 1 roots / projections / laterals and the stride-2 3 x 3
 * layers, cubercnn/modeling/backbone/dla.py:43-51, 159-172, 205-214; detectron2 FPN laterals) in as few launches as their tile shapes
 * allow -- normally one per backward stage.  Dense tensors (ldx = C, lddy = K), square filters; arrays are HOST arrays of n entries.
 * nsrc[i] > 0: x of problem i is the channel concatenation of xs[i * 6 + s] (widths cs[i * 6 + s]), as omni_conv2d_wgrad_multi_det.
 * Deterministic form (ctr != NULL): every problem keeps the split structure of its own omni_conv2d_wgrad_det launch -- bit-identical
 * results; plan != NULL: plan[2] = counters, plan[3] = workspace floats needed, nothing is launched. */
int omni_conv2d_wgrad_batch_det(const void* const* x, const void* const* dy, const void* const* dw, const int* N, const int* H, const int* W,
                                const int* C, const int* K, const int* R, const int* stride, const int* pad, const int* accumulate,
                                const void* const* xs, const int* cs, const int* nsrc, int n, float* ws, long long ws_floats, int* ctr,
                                int n_ctr, long long* plan, void* stream);
/* omni_conv2d_wgrad_det for an input that is the channel concatenation of nsrc <= 6 dense NHWC tensors (omni_conv2d_fwd_multi_det):
 * dw (K, 1, 1, sum cs) = dy^T x of the DLA Root's 1 x 1 convolution (dla.py:166-172) without the concatenated copy; cs[s] % 4 == 0,
 * xs / cs HOST arrays, dy (N, H, W, K) with pixel pitch lddy.  Bit-identical to the single-tensor entry on the concatenated input.
 * ctr == NULL && plan == NULL: non-deterministic form. */
int omni_conv2d_wgrad_multi_det(const void* const* xs, const int* cs, int nsrc, const float* dy, float* dw, int N, int H, int W, int K,
                                int lddy, int accumulate, int tile, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan,
                                void* stream);

/* ------------------------------------------------------- BatchNorm / pooling / FPN (NHWC) */

/* nn.BatchNorm2d in training mode (+ fused ReLU and residual add): cubercnn/modeling/backbone/
 * dla.py:46-66 (BasicBlock), :162-172 (Root), :214 (project), :244,294 (conv levels).
 * x, y, residual [nullable]: NHWC fp32 with P = N*H*W pixels, C % 4 == 0, C <= 1024.
 * running_mean/var [nullable] updated with `momentum` (unbiased var) like F.batch_norm.
 * Outputs kept for backward: mean_rstd (2C), scale_shift (2C).  ws: >= 2C*258 doubles scratch. */
int omni_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                float* running_mean, float* running_var, float* mean_rstd, float* scale_shift, double* ws,
                int P, int C, float eps, float momentum, int relu, void* stream);

/* Round 5: the same forward / backward with every variant behind explicit arguments (the entry points above and the *_partials /
 * *_carry forms below are these with defaults).  partial [nullable] = [nblk][2][C] statistics rows already written by the producer of x
 * (forward) / of dy (backward); NULL = a reduction pass runs first (ws as above).  ldy / lddy / ldc: pixel pitches in floats of y, dy
 * and res_carry (0 or C = dense; a DLA Root child, dla.py:171, is written into / read from its channel slice of the concatenated
 * tensor).  fuse_rows: with <= fuse_rows partial rows and C % 16 == 0 the finalize is folded into the apply launch -- every workgroup
 * owns 16 channels of a pixel chunk and sums the rows of those channels itself in the finalize kernel's order: same bits, one
 * dependent launch less (0 = always the separate finalize launch). */
int omni_bn_fwd_algo(const float* x, const float* partial, int nblk, const float* gamma, const float* beta, const float* residual,
                     float* y, long long ldy, float* running_mean, float* running_var, float* mean_rstd, float* scale_shift,
                     double* ws, int P, int C, float eps, float momentum, int relu, int fuse_rows, void* stream);
int omni_bn_bwd_algo(const float* x, const float* dy, long long lddy, const float* y, const float* gamma, const float* mean_rstd,
                     const float* partial, int nblk, float* dx, float* dres, const float* res_carry, long long ldc, float* dgamma,
                     float* dbeta, double* ws, float* coef, int P, int C, int relu, int accumulate_param_grads, int fuse_rows,
                     void* stream);

/* eval-mode / frozen BatchNorm (solver/build.py:71-76 freeze_bn): y = relu?(x*scale+shift(+res)). */
int omni_bn_apply(const float* x, const float* scale_shift, const float* residual, float* y, int P, int C,
                  int relu, void* stream);

/* backward of omni_bn_fwd.  dy = grad wrt y; dres [nullable] = grad wrt residual;
 * ws >= 2C*258 doubles, coef 3C floats scratch.  relu: 0 = none; 1 = mask dy by y > 0, y = the forward output;
 * 2 = layers without residual: `y` points at the forward pass's scale_shift (2C floats) and the mask is recomputed as
 * x * scale + shift > 0, the output tensor is not read. */
int omni_bn_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* mean_rstd,
                float* dx, float* dres, float* dgamma, float* dbeta, double* ws, float* coef, int P, int C,
                int relu, int accumulate_param_grads, void* stream);

/* nn.MaxPool2d(2, stride=2) (dla.py:209) forward / backward, NHWC. */
int omni_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* Grouped convolution `nn.Conv2d(C, K, R, stride, pad, groups=G, bias=False)` of DLA's BottleneckX (cubercnn/modeling/backbone/
 * dla.py:112-153): x (N,H,W,C), w (K,R,S,C/G), out / dy (N,OH,OW,K); C/G and K/G multiples of 4.  One launch of the implicit-GEMM
 * kernels per group on channel slices (pixel pitch C / K). */
int omni_grouped_conv2d_fwd(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int S, int stride,
                            int pad, int groups, void* stream);
int omni_grouped_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S, int stride,
                              int pad, int groups, void* stream);
int omni_grouped_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int stride,
                              int pad, int groups, void* stream);

/* Depthwise convolution `nn.Conv2d(C, C, R, padding=pad, stride=stride, groups=C, bias=False)` of torchvision's mnasnet1_0
 * (lifted by cubercnn/modeling/backbone/mnasnet.py:14-17): R in {3, 5}, stride in {1, 2}; x (N,H,W,C), w (R,R,C) tap-major,
 * y / dy (N,OH,OW,C), dw (R,R,C) overwritten. */
int omni_dwconv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int R, int stride, int pad, void* stream);
int omni_dwconv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int R, int stride, int pad, void* stream);
int omni_dwconv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int R, int stride, int pad, void* stream);

/* nn.AvgPool2d(2, stride=2) of torchvision densenet121's transitions (cubercnn/modeling/backbone/densenet.py:14-15, 27-30)
 * forward / backward, NHWC; dy (N, H/2, W/2, C) -> dx (N, H, W, C). */
int omni_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* F.max_pool2d(x, kernel_size=1, stride=2) (dla.py:474, resnet.py:55): y[n,oh,ow]=x[n,2oh,2ow]. */
int omni_subsample2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_subsample2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* detectron2 FPN top-down step: out = lateral + F.interpolate(top, 2.0, "nearest"); and the
 * gradient wrt `top` (2x2 block sums). */
int omni_upsample2_add(const float* lat, const float* top, float* out, int N, int H, int W, int C,
                       void* stream);
int omni_upsample2_bwd(const float* dout, float* dtop, int N, int H, int W, int C, void* stream);

/* Gradient fan-in ("carry") forms of four backward kernels.  An activation with several consumers -- the input of a DLA Tree
 * (max-pool + first block, cubercnn/modeling/backbone/dla.py:204-222), a block output that is the next block's input, its residual
 * and a Root child (:60-66, :171-173), an FPN top-down map (detectron2 FPN.forward, built at dla.py:500-506), an FPN output read
 * by the RPN head and by ROIAlign (cubercnn/modeling/roi_heads/roi_heads.py:267,362) -- has a gradient that is the SUM of its
 * consumers' gradients; autograd forms that sum with one elementwise add kernel per extra consumer.  Here the consumer that runs
 * later reads what the earlier ones produced (`carry`: NHWC, same extent as the output, pixel pitch ldc floats -- a whole tensor or a
 * channel slice of the Root's concatenated gradient; ldc >= channels, ldc % 4 == 0, 16-byte aligned) and writes the sum:
 *   omni_wino_out_carry      y = A^T M A + carry                 (Winograd data gradient, tile as omni_wino_out)
 *   omni_bn_bwd_carry        dres = masked dy + res_carry        (res_carry nullable: then == omni_bn_bwd)
 *   omni_maxpool2_bwd_carry  dx = routed dy + carry              (carry nullable; H, W even when given)
 *   omni_upsample2_bwd_carry dtop = 2x2 block sums + carry       (carry nullable)
 *   omni_subsample2_bwd_carry dx = carry + dy at the even pixels (the p6 = p5[::2, ::2] level; one pass instead of fill + scatter + add)
 * The implicit-GEMM data gradient takes the same role through omni_conv2d_dgrad(accumulate = 1, lddx = the carry's pitch).
 * omni_bn_bwd_carry / omni_maxpool2_bwd_carry also read their OUTPUT gradient dy with a pixel pitch lddy (same constraints): a
 * Root child with no other consumer gets its slice of the concatenated gradient without a copy. */
int omni_wino_out_carry(const float* M, const float* carry, long long ldc, float* y, int N, int H, int W, int K, int tile, void* stream);
int omni_bn_bwd_carry(const float* x, const float* dy, long long lddy, const float* y, const float* gamma, const float* mean_rstd, float* dx,
                      float* dres, const float* res_carry, long long ldc, float* dgamma, float* dbeta, double* ws, float* coef,
                      int P, int C, int relu, int accumulate_param_grads, void* stream);
int omni_maxpool2_bwd_carry(const float* x, const float* dy, long long lddy, const float* carry, long long ldc, float* dx, int N, int H,
                            int W, int C, void* stream);
int omni_subsample2_bwd_carry(const float* dy, const float* carry, long long ldc, float* dx, int N, int H, int W, int C, void* stream);
int omni_upsample2_bwd_carry(const float* dout, const float* carry, long long ldc, float* dtop, int N, int H, int W, int C,
                             void* stream);

/* GeneralizedRCNN.preprocess_image (called at cubercnn/modeling/meta_arch/rcnn3d.py:46,87):
 * uint8 planar (N,3,H,W) -> fp32 NHWC (N,PH,PW,4), (v-mean)/std, channel 3 and padding = 0. */
int omni_preprocess(const unsigned char* img, float* out, int N, int H, int W, int PH, int PW, float m0,
                    float m1, float m2, float s0, float s1, float s2, void* stream);
/* The same for a batch staged into fixed-size slots: image_hw (N,2) device ints = the valid height / width of image n inside its
 * H x W slot; everything outside is zero padding like the reference's ImageList.from_tensors of the real sizes
 * (rcnn3d.py:46 -> detectron2 ImageList).  Lets one captured training step serve every batch of a size bucket
 * (configs/Base.yaml:10-13 draws a new short edge per image; cubercnn/solver/autoreplay.py). */
int omni_preprocess_masked(const unsigned char* img, const int* image_hw, float* out, int N, int H, int W, int PH, int PW,
                           float m0, float m1, float m2, float s0, float s1, float s2, void* stream);
/* Round 6: the same from N <= 64 SEPARATE images (imgs: HOST array of N device pointers to planar uint8 (3, H, W) images) -- the
 * `torch.stack` of ImageList.from_tensors folded into the read; image_hw nullable. */
int omni_preprocess_multi(const void* const* imgs, const int* image_hw, float* out, int N, int H, int W, int PH, int PW, float m0,
                          float m1, float m2, float s0, float s1, float s2, void* stream);

/* ----------------------------------------------------------- index-exact selection kernels */

/* Row-wise sorted top-k (descending; equal keys -> lower index first).  Stands in for
 * `logits.sort(descending=True)[:k]` of detectron2 find_top_rpn_proposals (RPN configured at
 * configs/Base.yaml:49-54) and for torch.multinomial (== top-k of w / Exp(1)) at
 * cubercnn/modeling/proposal_generator/rpn.py:318,322.  Element (r,i) = keys[r*pitch + i*estride].
 * out_val / out_idx: (rows, k); slots >= min(k,n) hold -inf / -1.  k <= 8192. */
int omni_topk_rows(const float* keys, int rows, int n, long long pitch, int estride, int k, float* out_val,
                   int* out_idx, void* stream);
/* The same for `nseg` (<= 8) column segments of every row in one launch: the per-FPN-level pre-NMS top-k of
 * detectron2 find_top_rpn_proposals (one sort per level and image upstream).  seg_off / seg_n are HOST int arrays;
 * out_val / out_idx: (rows, nseg, k), indices relative to the segment start. */
int omni_topk_segments(const float* keys, int rows, long long pitch, int estride, int nseg, const int* seg_off,
                       const int* seg_n, int k, float* out_val, int* out_idx, void* stream);

/* Greedy NMS (torchvision.ops.nms semantics: suppress iff IoU > thr, areas (x2-x1)*(y2-y1)) for Q
 * independent score-sorted problems; detectron2 batched_nms = one problem per (image, level) /
 * (image, class): RPN [upstream], cubercnn/modeling/roi_heads/fast_rcnn.py:105.
 * boxes (Q,nmax,4); counts [nullable] (Q); valid [nullable] (Q,nmax); keep (Q,nmax) int32 0/1;
 * mask_ws: Q*nmax*ceil(nmax/64) 64-bit words scratch. nmax <= 8192. */
int omni_nms_sorted(const float* boxes, const int* counts, const int* valid, int Q, int nmax, float iou_thr,
                    unsigned long long* mask_ws, int* keep, void* stream);

/* ---------------------------------------------- RPN / ROI-head box logic (index-exact parts) */

/* detectron2 pairwise_iou (mode 0) / pairwise_ioa (mode 1): cubercnn/modeling/proposal_generator/
 * rpn.py:62,100; roi_heads/roi_heads.py:881,892.  out (N, M). */
int omni_pairwise_iou(const float* boxes1, int N, const float* boxes2, int M, int mode, float* out, void* stream);

/* RPNWithIgnore.label_and_sample_anchors, matching half (rpn.py:62-75) = pairwise_iou + detectron2
 * Matcher(thr, labels, allow_low_quality) + per-GT best anchor, for a batch of B images.
 * anchors (A,4); gt: concatenated valid GT boxes (G,4) with gt_off (B+1); expo (B,A) Exp(1) variates.
 * Outputs (B,A): matched_val, matched_idx (GT index inside the image), match_label int8,
 * key_pos / key_neg = (iou+eps)/E for the positive / negative candidates (-inf elsewhere) whose
 * top-k is the IoU-weighted multinomial of rpn.py:318,322; gt_best_idx (G); gt_best_bits (G) scratch. */
int omni_rpn_match(const float* anchors, int A, const float* gt, const int* gt_off, int B, int G, float thr_lo,
                   float thr_hi, int l0, int l1, int l2, int allow_low_quality, const float* expo, float eps,
                   float* matched_val, int* matched_idx, signed char* match_label, int* gt_best_bits,
                   int* gt_best_idx, float* key_pos, float* key_neg, void* stream);

/* Second half (rpn.py:79-105): labels (B,A) int8 in {-1,0,1} from the sampled top-k lists, forced
 * best-anchor positives and ignore regions (ign (Gi,4), ign_off (B+1)).  counts (B,2) [nullable]. */
int omni_rpn_finalize_labels(const float* anchors, int A, int B, const int* gt_off, const float* ign,
                             const int* ign_off, const signed char* match_label, const int* gt_best_idx,
                             const float* pos_val, const int* pos_idx, const float* neg_val, const int* neg_idx,
                             int kpos, int kneg, int batch_per_image, float ignore_thresh, signed char* labels,
                             int* counts, void* stream);

/* RPN head tensors: level_ptrs is a HOST array of nlev device pointers to (B, hw_l, 16) fp32
 * [3 logits | 12 deltas | pad], level_hw a HOST int array.  Gathers the logits into (B, A). */
int omni_rpn_gather_logits(const void* const* level_ptrs, const int* level_hw, int nlev, int B, float* logits,
                           void* stream);

/* The two 1x1 convolutions of detectron2's StandardRPNHead (objectness_logits: nn.Conv2d(256, 3, 1), anchor_deltas:
 * nn.Conv2d(256, 12, 1); configs/Base.yaml:49, reached from cubercnn/modeling/proposal_generator/rpn.py:129-135) over ALL FPN levels
 * in one launch per direction, as one 16-wide product per pixel: y[p] = [3 logits | 12 deltas | 0] (the layout above).
 * t / y / dy / dt: HOST arrays of nlev (<= 8) device pointers -- t_l (P_l, 256) fp32 NHWC = relu(conv(x_l)), y_l / dy_l (P_l, 16),
 * dt_l (P_l, 256); pix: HOST array of P_l = B * H_l * W_l; w_obj (3, 256), w_del (12, 256), biases (3), (12).
 *   fwd    y_l = t_l . [w_obj | w_del | 0]^T + [b_obj | b_del | 0]                 (fp32 MFMA 16x16x4, operands straight from HBM)
 *   dgrad  dt_l = dy_l . [w_obj | w_del | 0], zeroed where t_l <= 0 when relu_mask != 0 (the ReLU backward of the shared conv)
 *   wgrad  dw_obj, db_obj, dw_del, db_del [each nullable] = (accumulate == 0) or += the sums over every pixel of every level;
 *          partial = scratch of partial_rows (>= 1; 512 fills the chip) * (15 * 256 + 16) floats; fixed summation order, no atomics. */
int omni_rpn_head16_fwd(const void* const* t, const long long* pix, int nlev, const float* w_obj, const float* b_obj,
                        const float* w_del, const float* b_del, const void* const* y, void* stream);
int omni_rpn_head16_dgrad(const void* const* dy, const void* const* t, const long long* pix, int nlev, const float* w_obj,
                          const float* w_del, int relu_mask, const void* const* dt, void* stream);
int omni_rpn_head16_wgrad(const void* const* dy, const void* const* t, const long long* pix, int nlev, float* partial,
                          int partial_rows, float* dw_obj, float* db_obj, float* dw_del, float* db_del, int accumulate, void* stream);

/* RPNWithIgnore.losses with OBJECTNESS_UNCERTAINTY "IoUness" (rpn.py:129-204, 206-273): sums (6 doubles)
 * = [sum BCE*t, sum L1*t, #pos, #neg, sum sigmoid(pos), sum sigmoid(non-pos)]. */
int omni_rpn_loss_fwd(const void* const* level_ptrs, const int* level_hw, int nlev, int B, const float* anchors,
                      const signed char* labels, const int* matched_idx, const float* gt, const int* gt_off,
                      double* sums, void* stream);
int omni_rpn_loss_bwd(const void* const* level_ptrs, const void* const* dlevel_ptrs, const int* level_hw, int nlev,
                      int B, const float* anchors, const signed char* labels, const int* matched_idx,
                      const float* gt, const int* gt_off, const float* g_cls, const float* g_loc, float inv_norm,
                      void* stream);

/* The same with OBJECTNESS_UNCERTAINTY "none" (rpn.py:181-195, detectron2's own RPN losses): objectness BCE against the 0 / 1
 * labels over every sampled anchor, unweighted L1 on the foreground deltas; same sums layout. */
int omni_rpn_loss_plain_fwd(const void* const* level_ptrs, const int* level_hw, int nlev, int B, const float* anchors,
                            const signed char* labels, const int* matched_idx, const float* gt, const int* gt_off,
                            double* sums, void* stream);
int omni_rpn_loss_plain_bwd(const void* const* level_ptrs, const void* const* dlevel_ptrs, const int* level_hw, int nlev,
                            int B, const float* anchors, const signed char* labels, const int* matched_idx,
                            const float* gt, const int* gt_off, const float* g_cls, const float* g_loc, float inv_norm,
                            void* stream);

/* detectron2 RPN._decode_proposals + Boxes.clip + nonempty filter of find_top_rpn_proposals, applied
 * to the selected per-level top-k anchors only.  slot_level (Ktot), idx (B,Ktot), image_hw (B,2) are
 * device arrays; boxes (B,Ktot,4), valid (B,Ktot) out. */
int omni_rpn_decode(const void* const* level_ptrs, const int* level_hw, int nlev, int B, int Ktot,
                    const int* slot_level, const int* idx, const float* anchors, const int* image_hw,
                    float scale_clamp, float min_size, float* boxes, int* valid, void* stream);

/* Post-NMS glue of find_top_rpn_proposals (detectron2; the `keep[:post_nms_topk]` step of the RPN configured in
 * /root/reference/configs/Base.yaml:49-54).  omni_rpn_mask_scores: masked (n) = keep ? scores : -inf, the input of the
 * post-NMS ranking.  omni_rpn_collect: boxes (B,N,4), top_v / top_i (B,P) = that ranking (sorted, -inf / -1 padded)
 * -> prop (B,P,4) (zeros behind the real ones), count (B). */
int omni_rpn_mask_scores(const float* scores, const int* keep, long long n, float* masked, void* stream);
int omni_rpn_collect(const float* boxes, const float* top_v, const int* top_i, int B, int N, int P, float* prop,
                     int* count, void* stream);

/* ROIHeads3D.label_and_sample_proposals (roi_heads.py:862-929): append GT, Matcher(iou_thr), ignore
 * regions, IoU-weighted sampling of <= nfg_max foreground + background up to batch_per_image.
 * expo (B, 2048).  Outputs (B, batch_per_image): boxes, class (num_classes = bg, -2 = padding),
 * global GT row, matched IoU; counts (B,2). */
int omni_roi_sample(const float* prop_boxes, const int* prop_count, int B, int pmax, const float* gt,
                    const int* gt_cls, const int* gt_off, const float* ign, const int* ign_off, const float* expo,
                    float iou_thr, float ignore_thresh, float eps, int num_classes, int batch_per_image,
                    int nfg_max, int append_gt, float* out_boxes, int* out_cls, int* out_gt, float* out_iou,
                    int* out_counts, void* stream);
/* Round 6 forms of the two subsampling steps: expo == NULL draws the Exp(1) variates INSIDE the kernel (csrc/philox.h: Philox4x32-10
 * keyed by draw_state[0] = seed, counter (element, row, draw_state[1]); the last workgroup to take its ticket advances
 * draw_state[1]; ticket: one zeroed int32) -- replaces `Tensor.exponential_()` + torch's graph-safe generator bookkeeping, six
 * launches per step for the randomness of detectron2 `subsample_labels` (rpn.py:41-127 / roi_heads.py:862-929 call sites).
 * omni_roi_sample_draw also emits out_row (out_gt with -1 -> 0, the `clamp(min=0)` of the loss call sites) and contiguous copies
 * of the first `first` slots per image (the cube head's ROIs: roi_heads.py:341-362). */
int omni_rpn_match_draw(const float* anchors, int A, const float* gt, const int* gt_off, int B, int G, float thr_lo,
                        float thr_hi, int l0, int l1, int l2, int allow_low_quality, const float* expo, long long* draw_state,
                        int* ticket, float eps, float* matched_val, int* matched_idx, signed char* match_label, int* gt_best_bits,
                        int* gt_best_idx, float* key_pos, float* key_neg, void* stream);
int omni_roi_sample_draw(const float* prop_boxes, const int* prop_count, int B, int pmax, const float* gt, const int* gt_cls,
                         const int* gt_off, const float* ign, const int* ign_off, const float* expo, long long* draw_state, int* ticket,
                         float iou_thr, float ignore_thresh, float eps, int num_classes, int batch_per_image, int nfg_max,
                         int append_gt, float* out_boxes, int* out_cls, int* out_gt, float* out_iou, int* out_counts, int* out_row,
                         int first, float* first_boxes, int* first_cls, int* first_row, void* stream);

/* ----------------------------------------------------------------------------- ROIAlign */

/* detectron2 assign_boxes_to_levels (ROIPooler, canonical 224 / level 4). */
int omni_roi_levels(const float* rois, int R, int min_level, int max_level, float canonical_size,
                    int canonical_level, int* levels, void* stream);

/* detectron2 ROIPooler + torchvision roi_align(aligned=True, sampling_ratio=0): box pooler
 * (roi_heads.py:267) and cube pooler (roi_heads.py:166-171,362).  level_ptrs/level_hw/level_scale are
 * HOST arrays; features (B,H_l,W_l,C) NHWC; out (R,P,P,C). */
int omni_roi_align_fwd(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                       const float* rois, const int* batch_idx, const int* levels, int R, int P, int C,
                       float* out, void* stream);
/* Round 6: torchvision roi_align with its `aligned` switch exposed: aligned = 0 is detectron2's POOLER_TYPE "ROIAlign" (no half-pixel
 * shift, ROI sides of at least one pixel), 1 = "ROIAlignV2" (cubercnn/modeling/roi_heads/roi_heads.py:166-171 passes
 * MODEL.ROI_BOX_HEAD.POOLER_TYPE / MODEL.ROI_CUBE_HEAD.POOLER_TYPE through to detectron2's ROIPooler).  Any P; the backward adds into
 * dlevel_ptrs with fp32 atomics (the caller zeroes them). */
int omni_roi_align_fwd_mode(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                            const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, int aligned, float* out,
                            void* stream);
int omni_roi_align_bwd_mode(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                            const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, int aligned,
                            const float* dout, void* stream);
/* Round 6: POOLER_TYPE "ROIPool" = torchvision.ops.roi_pool as detectron2's ROIPooler applies it level by level (the reference passes
 * MODEL.ROI_BOX_HEAD.POOLER_TYPE / MODEL.ROI_CUBE_HEAD.POOLER_TYPE through: cubercnn/modeling/roi_heads/roi_heads.py:166-171).
 * out and argmax (R, P, P, C); argmax = h * W + w of the first maximum inside the ROI's image and level, -1 for an empty bin (out 0).
 * The backward adds dout at the argmax pixels with fp32 atomics (the caller zeroes dlevel_ptrs). */
int omni_roi_pool_fwd(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev, const float* rois,
                      const int* batch_idx, const int* levels, int R, int P, int C, float* out, int* argmax, void* stream);
int omni_roi_pool_bwd(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev, const int* batch_idx,
                      const int* levels, int R, int P, int C, const float* dout, const int* argmax, void* stream);
/* Round 6: forward that also writes the first `first` ROIs of every block of `per_image` to out2 ((R / per_image) * first, P, P, C) --
 * the box head's and the cube head's pooled features in one pass (roi_heads.py:166-171, 267, 362), no slice copy. */
int omni_roi_align_fwd2(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                        const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, float* out,
                        float* out2, int per_image, int first, void* stream);
int omni_roi_align_bwd(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                       const float* rois, const int* batch_idx, const int* levels, int R, int P, int C,
                       const float* dout, void* stream);
/* The same with two gradient tensors (P == 7): dout (R,7,7,C) [nullable] and dout2 ((R / per_image) * first, 7, 7, C) [nullable] for
 * the first `first` ROIs of every block of `per_image` (the cube head's ROIs are a prefix of the box head's in one shared pass). */
int omni_roi_align_bwd2(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                        const float* rois, const int* batch_idx, const int* levels, int R, int P, int C,
                        const float* dout, const float* dout2, int per_image, int first, void* stream);
/* Deterministic form (round 4; P == 7, R <= 4096, B = images): the OUTPUT owns the sum -- one workgroup per 8 x 4 pixel tile of a
 * (level, image), all channels: its four waves deal out the ROIs that touch the tile (wave s: entries s, s + 4, ... of the list in
 * ascending ROI index) and meet in wave order, so an element's value depends on the inputs alone; every element of dlevel_ptrs[l]
 * is written exactly once (OVERWRITES: no zero-fill by the caller), no atomics: two runs are bit-identical.
 * C <= 256, B <= 255.  ws: scratch for per-ROI footprint records (plan != NULL: plan[3] = floats needed, nothing is launched);
 * ctr / n_ctr are not used by this kernel.  Same call site and gradient arguments as omni_roi_align_bwd2. */
int omni_roi_align_bwd_det(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev, int B,
                           const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, const float* dout,
                           const float* dout2, int per_image, int first, float* ws, long long ws_floats, int* ctr, int n_ctr,
                           long long* plan, void* stream);

/* ------------------------------------------------------------------- box-head / cube losses */

/* FastRCNNOutputs.losses (cubercnn/modeling/roi_heads/fast_rcnn.py:145-260) on the fused prediction
 * tensor pred (R, ldp) = [K+1 logits | 4K deltas].  sums (7 doubles). */
int omni_box_loss_fwd(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, const float* gt,
                      const int* gt_row, float wx, float wy, float ww, float wh, double* sums, void* stream);
int omni_box_loss_bwd(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, const float* gt,
                      const int* gt_row, float wx, float wy, float ww, float wh, const double* sums,
                      const float* g_cls, const float* g_reg, float* dpred, void* stream);

/* FastRCNNOutputLayers.predict_boxes_for_gt_classes (detectron2), called at cubercnn/modeling/roi_heads/roi_heads.py:276-289
 * (MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES): out (R,4) = Box2BoxTransform.apply_deltas of each row's GT-class deltas
 * (class clamped to [0, K-1]; cls < 0 copies the proposal) on its proposal box, unclipped. */
int omni_box_decode_gt_class(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, float wx, float wy,
                             float ww, float wh, float scale_clamp, float* out, void* stream);

/* ROIHeads3D._forward_cube decode + losses (cubercnn/modeling/roi_heads/roi_heads.py:374-768) on the fused cube-head
 * outputs head (F, ldh) = [xy 2K | z K*bins (bin*K + class) | dims 3K | pose Pn*K | uncert K (if confidence)],
 * Pn = 6 / 4 / 3 for POSE_TYPE 6d / quaternion / euler (cube_head.py:118-136,175-192), bins = max(CLUSTER_BINS, 1).
 * mode = MODEL.ROI_CUBE_HEAD configuration: bits 0-1 Z_TYPE (0 direct, 1 sigmoid, 2 log, 3 clusters), bits 2-3 dimensions
 * (0 priors 'exp', 1 priors 'sigmoid', 2 DIMS_PRIORS_ENABLED False), bits 4-5 POSE_TYPE (0 6d, 1 quaternion, 2 euler),
 * bit 6 ALLOCENTRIC_POSE, bit 7 VIRTUAL_DEPTH, bit 8 CHAMFER_POSE, bit 9 INVERSE_Z_WEIGHT, bit 10 USE_CONFIDENCE > 0,
 * bit 11 LOSS_W_JOINT > 0, bit 12 DISENTANGLED_LOSS False (configs/Base.yaml = 0xDC0).
 * zscales (K, bins): 2D-scale prior of every depth cluster (roi_heads.py:123-130), zstats (K, bins, 2): its depth mean / std
 * (:133-143); null when bins == 1 / Z_TYPE is not 'clusters'.  Bit 12 needs dimensions = 2: the reference's own entangled
 * dimension loss cannot be evaluated with priors (roi_heads.py:620-622 divides an (n,3) by an (n,2,3) tensor).
 * w_*: MODEL.ROI_CUBE_HEAD.LOSS_W_{DIMS,POSE,XY,Z,JOINT}, used for the logged `Cube/total_3D_loss` (:651-695). */
int omni_cube_loss_fwd(const float* head, int ldh, int F, int K, int mode, int bins, const float* zscales, const float* zstats,
                       const float* boxes, const int* cls, const int* img,
                       const float* Ks, const float* v2r, const float* priors, const float* gt3d,
                       const float* gtpose, const int* gt_row, float w_dims, float w_pose, float w_xy, float w_z, float w_joint,
                       float* vals, float* jac, float* red, void* stream);
int omni_cube_loss_bwd(const float* vals, const float* jac, const float* red, const float* gk, const int* cls,
                       const float* boxes, int F, int K, int mode, int bins, const float* zscales, int ldh, float* dhead,
                       void* stream);
/* inference outputs (roi_heads.py:771-819): cube3d (F,9), pose (F,9), verts (F,24). */
int omni_cube_decode(const float* head, int ldh, int F, int K, int mode, int bins, const float* zscales, const float* zstats,
                     const float* boxes, const int* cls, const int* img,
                     const float* Ks, const float* v2r, const float* ratio, const float* priors, float* cube3d,
                     float* pose, float* verts, void* stream);
/* util.get_cuboid_verts_faces (cubercnn/util/math_util.py:116-219): box3d (n,6), R (n,9) -> (n,24). */
int omni_cuboid_corners(const float* box3d, const float* R, int n, float* verts, void* stream);

/* ------------------------------------------- BatchNorm statistics from the producing kernel's epilogue
 * The conv -> BatchNorm pairs of the bottom-up (cubercnn/modeling/backbone/dla.py:46-66,162-172,214,244; torchvision
 * BasicBlock via resnet.py:17-27): the convolution / Winograd output transform / stem kernel also writes per-workgroup
 * partial sums and sums of squares of its output, stats [rows][2][K] floats, so training-mode BatchNorm skips its own
 * statistics pass (omni_bn_fwd_partials = finalize + apply).  *nblk_out = partial rows written; 0 means "not produced"
 * (split reduction chosen, buffer too small, unsupported channel count) and the caller uses omni_bn_fwd. */
int omni_conv2d_fwd_stats(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int ldx, int ldo, float* stats, int stats_rows, int* nblk_out, void* stream);
int omni_wino_out_stats(const float* M, float* y, int N, int H, int W, int K, int tile, float* stats, int stats_rows,
                        int* nblk_out, void* stream);
/* The same idea in the backward pass: the Winograd data-gradient transform of the convolution ABOVE a BatchNorm(+ReLU) writes that
 * BatchNorm's output gradient dy, and emits the per-workgroup partial sums (sum dz, sum dz * xhat) its backward pass starts with
 * (dz = dy masked by x * scale + shift > 0 when scale_shift != NULL); omni_bn_bwd_partials = finalize + apply on those rows
 * (arguments as omni_bn_bwd).  *nblk_out == 0: not produced, run omni_bn_bwd. */
int omni_wino_out_bn_bwd_stats(const float* M, float* y, int N, int H, int W, int K, int tile, const float* bn_x,
                               const float* mean_rstd, const float* scale_shift, float* stats, int stats_rows, int* nblk_out,
                               void* stream);
int omni_bn_bwd_partials(const float* x, const float* dy, const float* y, const float* gamma, const float* mean_rstd,
                         const float* partial, int nblk, float* dx, float* dres, float* dgamma, float* dbeta, float* coef,
                         int P, int C, int relu, int accumulate_param_grads, void* stream);
int omni_stem_conv_fwd_stats(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx,
                             int ldo, float* stats, int stats_rows, int* nblk_out, void* stream);
int omni_bn_fwd_partials(const float* x, const float* partial, int nblk, const float* gamma, const float* beta,
                         const float* residual, float* y, float* running_mean, float* running_var, float* mean_rstd,
                         float* scale_shift, int P, int C, float eps, float momentum, int relu, void* stream);

/* ---------------------------------------------------------------- GEMM engine (csrc/gemm_engine.hip)
 * The cuBLAS calls behind nn.Linear (detectron2 FastRCNNConvFCHead; cubercnn/modeling/roi_heads/cube_head.py:70,108-163),
 * behind 1x1 nn.Conv2d (FPN laterals, DLA roots / projections, cubercnn/modeling/backbone/dla.py:159-161,214) and the point
 * GEMMs of the Winograd path, forward / data gradient / weight gradient:
 *   form 0 "NT": C = A (M x K) * B (N x K)^T     form 1 "NN": C = A (M x K) * B (K x N)     form 2 "TN": C = A (K x M)^T * B (K x N)
 * batch problems at element strides stride_a/b/c; pitches lda/ldb/ldc; splits > 1 or accumulate != 0 => fp32 atomics into C
 * (zeroed by the caller unless accumulating); bias (N) / ReLU on the direct-store path.  tile: 1 = 256x128, 2 = 128x128.
 * workgroups: persistent workgroups walking the (problem, tile, split) list (0 = one per CU). */
int omni_gemm_engine(const float* A, const float* B, float* C, const float* bias, int form, int batch, int M, int N, int K,
                     int lda, int ldb, int ldc, long long stride_a, long long stride_b, long long stride_c, int splits,
                     int relu, int accumulate, int tile, int workgroups, void* stream);
/* Deterministic form (round 4, csrc/split_reduce.h; same call sites): the parts of a cut tile -- splits > 1, or the left-over tiles
 * of the balanced split (splits == -1) -- meet in `ws` (ws_floats floats) and are summed in part order by the tile's last-arriving
 * workgroup (ctr: n_ctr zeroed unsigned counters, left zeroed), which applies bias / ReLU and overwrites C or, accumulate != 0,
 * adds to it with a plain read-modify-write; the caller zeroes nothing.  plan != NULL: plan[0..3] = {0, parts per cut tile,
 * counters needed, workspace floats needed}, nothing is launched. */
int omni_gemm_engine_det(const float* A, const float* B, float* C, const float* bias, int form, int batch, int M, int N, int K,
                         int lda, int ldb, int ldc, long long stride_a, long long stride_b, long long stride_c, int splits,
                         int relu, int accumulate, int tile, int workgroups, float* ws, long long ws_floats, int* ctr, int n_ctr,
                         long long* plan, void* stream);

/* dst (cols, rows) = src (rows, cols)^T, contiguous row-major fp32.  Brings the fc1-class weights (K_out x C_in) into the
 * (C_in x K_out) layout the data gradient of torch.nn.Linear (FastRCNNConvFCHead fc1, cube_head.py:70) multiplies in the
 * engine's fastest ("NT") form. */
int omni_transpose2d(const float* src, float* dst, int rows, int cols, void* stream);

/* ---------------------------------------------------------------- batched inference
 * fast_rcnn_inference / fast_rcnn_inference_single_image (cubercnn/modeling/roi_heads/fast_rcnn.py:33-116) for all images
 * of a batch with fixed shapes.  pred (B*P, ld) = [K+1 logits | 4K deltas]; rois (B*P,4); count (B); image_hw (B,2).
 *  omni_det_scores: probs (B*P,K) softmax, boxes (B*P,K,4) decoded (Box2BoxTransform weights wx..wh) and clipped,
 *    scores (B, P*K) = prob where the row is valid (finite, p < count) and prob > score_thresh, else -inf;
 *  [omni_topk_rows over scores -> vals / idx (B, cap), stable descending]
 *  omni_det_nms_boxes: candidate boxes + class * (max coordinate + 1) (detectron2 batched_nms) -> nms_boxes (B,cap,4), valid;
 *  [omni_nms_sorted -> keep (B, cap)]
 *  omni_det_compact: the first topk kept candidates per image -> out_box (B,topk,4), out_score, out_cls, out_roi, out_count. */
int omni_det_scores(const float* pred, int ld, const float* rois, const int* count, const int* image_hw, int B, int P, int K,
                    float wx, float wy, float ww, float wh, float score_thresh, float* scores, float* probs, float* boxes,
                    void* stream);
int omni_det_nms_boxes(const float* boxes, const float* vals, const int* idx, int B, int PK, int K, int cap, float* nms_boxes,
                       int* valid, void* stream);
int omni_det_compact(const int* keep, const int* valid, const float* vals, const int* idx, const float* boxes, int B, int PK,
                     int K, int cap, int topk, float* out_box, float* out_score, int* out_cls, int* out_roi, int* out_count,
                     void* stream);

/* ------------------------------------------------------------------ input pipeline
 * detectron2 T.ResizeShortestEdge -> ResizeTransform.apply_image = PIL Image.resize(BILINEAR) on the uint8 image, and
 * RandomFlip (horizontal), as configured by cubercnn/data/dataset_mapper.py:25-27 + configs/Base.yaml:10-13.  Bit-exact
 * with Pillow's 8-bit resampling: the fixed-point coefficient rows are built by the caller (omni3d_amd/kernels/resize.py,
 * the arithmetic of Pillow's precompute_coeffs) -- bounds (n_out, 2) = [first source index, count], kk (n_out, ksize) int32.
 * src (planes, H, W) -> dst (planes, HO, WO), tmp = planes*H*WO bytes scratch; flip != 0 mirrors the output columns. */
int omni_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, unsigned char* tmp, int planes, int H, int W,
                            int HO, int WO, const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v,
                            const int* kk_v, int ksize_v, int flip, void* stream);

/* ---------------------------------------------------------------------------- optimizer */

/* torch.optim.SGD step (cubercnn/solver/build.py:49-56, tools/train_net.py:250) over a flat bucket; the gradient is
 * read as grad * grad_scale (1/world after a summing all-reduce: DDP's averaging, tools/train_net.py:449-454). */
int omni_sgd_step(float* param, const float* grad, float* momentum_buf, long long n, float lr, float momentum,
                  float dampening, float weight_decay, int nesterov, int first_step, float grad_scale,
                  const float* skip_flag, void* stream);
/* torch.optim.Adam / AdamW (+ amsgrad) as built by cubercnn/solver/build.py:58-65 (eps 1e-2, betas (0.9, 0.999), the parameter
 * groups' lr / weight decay), single-tensor form, over one group's range of the flat buckets.  decoupled: 0 Adam (L2 into the
 * gradient), 1 AdamW (p *= 1 - lr wd).  max_exp_avg_sq: null or the amsgrad running maximum.  step: device float = number of this
 * update (bias corrections); omni_adam_tick advances it once per optimizer step unless skip_flag is set. */
int omni_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int decoupled, const float* step, float grad_scale,
                   const float* skip_flag, void* stream);
int omni_adam_tick(float* step, const float* skip_flag, void* stream);
/* The loop's divergence guard (tools/train_net.py:157-285: allreduce_dict :186, rolling-loss test :194-215, skip / step
 * :245-253, retry decision :258-270) around ONE small all-reduce.  vec (n + 2): [n loss scalars | sum | non-finite-gradient flag];
 * omni_guard_pre writes the sum, the caller all-reduces vec over the ranks (sum), omni_guard_post averages by `world`, updates
 * state (3) = [rolling loss (NaN = unset), iterations_success, iterations_explode], writes skip (1) for omni_sgd_step and
 * out (n + 3) = [skipped, retry, total, n reduced losses], and clears the flag. */
int omni_guard_pre(float* vec, int n, void* stream);
int omni_guard_post(float* vec, int n, int world, float stabilize, float half_period, float tolerance, float gamma, float* state,
                    float* skip, float* out, void* stream);
/* ---- round 6 (csrc/glue.hip): the scalar-sized ATen launches around the losses and the loop, one launch each.
 * omni_scale_vec: out[i] = (float)(src[i * stride] * coef[i] / max(*denom, denom_min)), i < n <= 16, in double, rounded once --
 *   `loss * weight`, `loss_sum / max(count, 1)` (detectron2 fast_rcnn.py losses(); cubercnn/modeling/roi_heads/roi_heads.py:745-768);
 *   src double when src_f64 else float, stride 0 = one broadcast scalar; coef: n HOST doubles or NULL; denom: device scalar or NULL.
 * omni_sum_vectors: out[0] / out[1] / out[2] = sum of the first nfirst vectors / of the rest / of all -- `sum(loss_dict.values())`
 *   (tools/train_net.py:180) split at the backward-stage boundary; vecs / lens: nvec <= 16 HOST entries (device pointers, lengths).
 * omni_guard_gather: vec[i] = *scalars[i], vec[n] = their sum -- the stacking of tools/train_net.py:186 + omni_guard_pre.
 * omni_bump_counters: *counters[i] += delta (nn.BatchNorm2d's num_batches_tracked of every layer; HOST array of device pointers).
 * omni_zero: zero-fill as a kernel node (optimizer.zero_grad() of the flat gradient bucket, tools/train_net.py:217). */
int omni_scale_vec(const void* src, int src_f64, int stride, const double* coef, const void* denom, double denom_min, int n, float* out,
                   void* stream);
int omni_sum_vectors(const void* const* vecs, const int* lens, int nvec, int nfirst, float* out, void* stream);
int omni_guard_gather(const void* const* scalars, int n, float* vec, void* stream);
int omni_bump_counters(const void* const* counters, int n, long long delta, void* stream);
int omni_zero(void* p, long long nbytes, void* stream);
/* the isnan/isinf gradient scan of tools/train_net.py:222-233 as one pass; flag[0] = 1 if any. */
int omni_nonfinite_any(const float* grad, long long n, float* flag, void* stream);

/* backward of the ReLU fused into conv / linear epilogues, and the bias gradient (per-channel sum
 * of dy over P pixels; ws 2C*258 doubles scratch).  autograd of the nn.Conv2d / nn.Linear call sites.  accumulate: 0 = overwrite | 1 = add (small tensors: fp32 atomics) | 3 = add
 * deterministically (partial rows + fixed-order finalize, round 4). */
int omni_relu_bwd(const float* dy, const float* y, float* dz, long long n, void* stream);
int omni_bias_grad(const float* dy, int P, int C, float* db, double* ws, int accumulate, void* stream);

/* nn.MaxPool2d(3, stride=2, padding=1) of the torchvision ResNet stem (cubercnn/modeling/backbone/resnet.py:34,52),
 * NHWC forward / backward (gather form, deterministic). */
int omni_maxpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------- Winograd F(2x2,3x3) path for wide 3x3 / s1 / p1 convolutions
 * (detectron2 FPN output convs built at cubercnn/modeling/backbone/dla.py:500-506 and StandardRPNHead.conv,
 * configs/Base.yaml:49 -- nn.Conv2d(256, 256, 3, padding=1) upstream).  H and W even, channels % 4 == 0.
 * T = N*H/2*W/2 tiles.  V / M / dM: [16][T][channels]; U: [16][K][C]; U' (rotated, channel-transposed filter): [16][C][K]. */
/* tile = 2: F(2x2,3x3), P = 16 points;  tile = 4: F(4x4,3x3), P = 36 points, H and W multiples of 4.  T = N*(H/tile)*(W/tile);
 * the [16] above reads [P]. */
int omni_wino_in(const float* x, float* V, int N, int H, int W, int C, int tile, void* stream);
/* conv -> BatchNorm(+ReLU) -> 3x3 conv (the inside of every DLA / torchvision BasicBlock, dla.py:60-66): omni_bn_finalize_fwd turns the
 * first convolution's epilogue statistics into (scale, shift) and omni_wino_in_affine applies them (+ ReLU) while it loads the input
 * tiles of the second -- affine = [scale (C) | shift (C)], zero padding outside the image as for the normalised tensor, which is
 * never stored.  affine NULL == omni_wino_in. */
int omni_wino_in_affine(const float* x, const float* affine, int relu, float* V, int N, int H, int W, int C, int tile, void* stream);
int omni_bn_finalize_fwd(const float* partial, int nblk, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* mean_rstd, float* scale_shift, int P, int C, float eps, float momentum,
                         void* stream);
int omni_wino_out(const float* M, const float* bias, float* y, int N, int H, int W, int K, int relu, int tile, void* stream);
int omni_wino_dy(const float* dy, float* dM, int N, int H, int W, int K, int tile, void* stream);
/* omni_wino_weights for n <= 48 filters in ONE launch (the weights are fixed during a step: every filter transform of the forward pass
 * at once).  g / U / U_flip: HOST arrays of device pointers (U[i] / U_flip[i] nullable, not both); K / C / tile: HOST int arrays. */
int omni_wino_weights_multi(const void* const* g, const void* const* U, const void* const* U_flip, const int* K, const int* C,
                            const int* tile, int n, void* stream);
/* backward: dM (as omni_wino_dy) and V_dy (as omni_wino_in of dy) from one read of dy */
int omni_wino_dy_in(const float* dy, float* dM, float* Vd, int N, int H, int W, int K, int tile, void* stream);
/* Row-range forms (round 4): the Winograd-domain array is a row range of a wider (points, rows_total, channels) array holding the tiles
 * of several tensors side by side -- the FPN levels under the RPN's shared 3x3 convolution (detectron2 StandardRPNHead.conv,
 * configs/Base.yaml:49) -- so that ONE batched GEMM over rows_total rows serves all of them.  V / M / dM / Vd point at the tensor's
 * first row; plane = rows_total * channels floats between point planes.  omni_wino_out_rows: carry != NULL selects the fan-in form of
 * omni_wino_out_carry (no bias / ReLU then). */
int omni_wino_in_rows(const float* x, float* V, int N, int H, int W, int C, int tile, long long plane, void* stream);
int omni_wino_out_rows(const float* M, const float* bias /*nullable*/, const float* carry /*nullable*/, long long ldc, float* y, int N,
                       int H, int W, int K, int relu, int tile, long long plane, void* stream);
int omni_wino_dy_in_rows(const float* dy, float* dM, float* Vd, int N, int H, int W, int K, int tile, long long plane, void* stream);
int omni_wino_weights(const float* g, float* U /*nullable*/, float* U_flip /*nullable: U'*/, int K, int C, int tile, void* stream);
int omni_wino_dweights(const float* dU, float* dg, int K, int C, int accumulate, int tile, void* stream);
/* n <= 16 of them in one launch, each ADDED into its gradient view; sources naming the same dg (the RPN's shared convolution: one
 * weight gradient per FPN level) are added in the order given, as the separate launches would */
int omni_wino_dweights_multi(const void* const* dU, const void* const* dg, const int* K, const int* C, const int* tile, int n,
                             void* stream);
/* `batch` independent dense GEMMs in one launch (the 16 Winograd points):
 * fwd: out[b](M,K) = x[b](M,C) * w[b](K,C)^T;  wgrad: dw[b](K,C) = dy[b](M,K)^T * x[b](M,C) (overwrites dw). */
int omni_gemm_batched_fwd(const float* x, const float* w, float* out, int batch, int M, int C, int K, void* stream);
/* algo: 0 = automatic | 1 = persistent workgroups walking the (problem, tile) list (`workgroups` of them, multiple of 8,
 * 0 = default; C % 32 == 0) | 2 = one 128x128 tile per workgroup | 3 = one 64x64 tile per workgroup | 4 = 64x64 tiles with
 * 2-4 slabs of buffer-load prefetch in flight (short reductions: the small-map point GEMMs; C % 64 == 0) */
int omni_gemm_batched_fwd_algo(const float* x, const float* w, float* out, int batch, int M, int C, int K, int algo,
                               int workgroups, void* stream);
int omni_gemm_batched_wgrad(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, void* stream);
/* algo: 0 = automatic | 1 = the implicit-GEMM weight-gradient tiles (128x128 / 64x64) | 2 = 64x64 tiles with 4 slabs of
 * buffer-load prefetch in flight (short reductions).  Same call site as omni_gemm_batched_wgrad (the Winograd-domain weight
 * gradient of torch.nn.Conv2d's backward, dla.py:43-51). */
int omni_gemm_batched_wgrad_algo(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, int algo,
                                 void* stream);
/* deterministic form (see omni_conv2d_fwd_det): the row splits of the Winograd-domain weight gradient meet in `ws` in split order */
int omni_gemm_batched_wgrad_det(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, int algo, float* ws,
                                long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream);
/* n <= 16 such weight-gradient GEMMs of different shapes in ONE launch (round 4: all the Winograd layers whose data gradients a
 * backward stage has produced; each alone is a latency-bound launch on 256 CUs).  Problem i: dw[i] (batch[i], K[i], C[i]) =
 * dy[i] (batch[i], M[i], K[i])^T x[i] (batch[i], M[i], C[i]); tiles, row splits and (ctr != NULL) the ordered split reduction are
 * those of omni_gemm_batched_wgrad_det(algo 2) on that problem alone -- bit-identical to n separate calls.  ws / ctr hold the
 * problems' regions back to back; plan: [0] = 2, [1] = 0, [2] = counters, [3] = workspace floats of the whole call. */
int omni_gemm_batched_wgrad_multi(const void* const* x, const void* const* dy, const void* const* dw, const int* batch, const int* M,
                                  const int* C, const int* K, int n, float* ws, long long ws_floats, int* ctr, int n_ctr,
                                  long long* plan, void* stream);

/* Direct convolution for the full-resolution, few-channel DLA-34 stem layers (cubercnn/modeling/backbone/dla.py:241-247):
 * out (N,H,W,16) = conv(x (N,H,W,C), w (16,R,R,C)), stride 1, padding R/2; (C, R) = (4, 7) [base_layer, image padded
 * 3 -> 4 channels] or (16, 3) [level0; also its data gradient with the rotated, channel-transposed filter]. */
int omni_stem_conv_fwd(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                       void* stream);
/* Round 5: the data gradients of the full-resolution stem layers on the same design (input halo + whole filter in LDS, MFMA 16x16x4).
 * omni_stem_conv_dgrad: 3x3 16 -> 16 stride 1 (level0, dla.py:246-247) from the layer's FORWARD filter -- the 180-degree rotation and
 * the channel transposition happen while the filter is staged.  omni_stem_conv_s2_dgrad: 3x3 stride 2 pad 1, 16 -> 32 (level1,
 * dla.py:248-249): dx (N,H,W,16) from dy (N,(H-1)/2+1,(W-1)/2+1,32); parity classes of the strided gradient inside one launch. */
int omni_stem_conv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int lddy, int lddx,
                         void* stream);
int omni_stem_conv_s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int lddy, int lddx,
                            void* stream);
/* Its weight gradient: dw (16,R,R,C) = (accumulate == 0) or += sum over pixels of dy (N,H,W,16) (x) x (N,H,W,C). */
int omni_stem_conv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                         int accumulate, void* stream);
/* deterministic form (round 4): per-workgroup partial filter gradients in `ws` (plan != NULL: plan[3] = floats needed, nothing
 * launched), added in a fixed order by a second launch; same call site as omni_stem_conv_wgrad */
int omni_stem_conv_wgrad_det(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                             int accumulate, float* ws, long long ws_floats, long long* plan, void* stream);
/* Round 6: the first layer (cubercnn/modeling/backbone/dla.py:241-245, 7x7 3 -> 16) on the 4-channel padded image with the filter
 * as the model holds it -- w / dw (16, R, R, cw), cw = 3 < C = 4: no padded copy of the filter, no slice + add of its gradient.
 * stats [nullable] as omni_stem_conv_fwd_stats; the weight gradient is the deterministic form (ws / plan as omni_stem_conv_wgrad_det). */
int omni_stem_conv_fwd_cw(const float* x, const float* w, int cw, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                          float* stats, int stats_rows, int* nblk_out, void* stream);
int omni_stem_conv_wgrad_det_cw(const float* x, const float* dy, float* dw, int cw, int N, int H, int W, int C, int K, int R, int ldx,
                                int lddy, int accumulate, float* ws, long long ws_floats, long long* plan, void* stream);
/* The stride-2 member of the family (round 4): dw (32,3,3,16) from x (N,H,W,16) and dy (N,H/2,W/2,32), H and W even -- the
 * weight gradient of DLA-34's level1 convolution (cubercnn/modeling/backbone/dla.py:291-295; torch.nn.Conv2d backward).
 * deterministic != 0: ws / plan as omni_stem_conv_wgrad_det; 0: fp32 atomics into dw. */
int omni_stem_conv_s2_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                            int accumulate, int deterministic, float* ws, long long ws_floats, long long* plan, void* stream);

/* Greedy detection <-> ground-truth matching of Omni3Deval.evaluateImg (cubercnn/evaluation/omni3d_evaluation.py:1433-1551,
 * 3D mode) for all (image, category) groups x A depth ranges x T IoU thresholds.  ious: ragged (D_g, G_g) matrices at
 * iou_off[g] (rows = detections in descending score order); dt_off / gt_off: (ngroups + 1) prefix offsets; max_gt <= 1024.
 * Out: dt_match (A,T,sumD) matched gt index within the group (original order) or -1; gt_match (A,T,sumG) matched dt index or
 * -1; dt_ignore (A,T,sumD); gt_order (A,sumG) the stable ignore-last order; gt_ig (A,sumG) `_ignore` per original gt. */
int omni_eval_match(const float* ious, const long long* iou_off, const int* dt_off, const int* gt_off, const int* gt_ignore,
                    const float* gt_range, const float* dt_range, const float* areas, const double* thrs, int ngroups, int A,
                    int T, int sumD, int sumG, int max_gt, int* dt_match, int* gt_match, unsigned char* dt_ignore, int* gt_order,
                    unsigned char* gt_ig, void* stream);

/* Omni3Deval.accumulate (omni3d_evaluation.py:1172-1313).  order (N): all detections sorted by (category, descending score)
 * (stable); cat_off (K+1); rank (sumD): position of a detection in its image's score-sorted list; dt_match / dt_ignore:
 * outputs of omni_eval_match; npig (K,A) non-ignored ground truths; has_e (K); rec_thrs (R); max_dets (M).
 * precision / scores (T,R,K,A,M) and recall (T,K,A,M): doubles, pre-filled with -1 by the caller. */
int omni_eval_accumulate(const int* order, const int* cat_off, const int* rank, const double* score, const int* dt_match,
                         const unsigned char* dt_ignore, const int* npig, const int* has_e, const double* rec_thrs,
                         const int* max_dets, int K, int A, int M, int T, int R, int sumD, double* precision, double* recall,
                         double* scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNI3D_HIP_H */
