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

/* grad wrt the weights: dw (K,R,S,C) overwritten.  Split-K over output pixels, fp32 atomics. */
int omni_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int ldx, int lddy, void* stream);

/* ------------------------------------------------------- BatchNorm / pooling / FPN (NHWC) */

/* nn.BatchNorm2d in training mode (+ fused ReLU and residual add): cubercnn/modeling/backbone/
 * dla.py:46-66 (BasicBlock), :162-172 (Root), :214 (project), :244,294 (conv levels).
 * x, y, residual [nullable]: NHWC fp32 with P = N*H*W pixels, C % 4 == 0, C <= 1024.
 * running_mean/var [nullable] updated with `momentum` (unbiased var) like F.batch_norm.
 * Outputs kept for backward: mean_rstd (2C), scale_shift (2C).  ws: >= 2C doubles scratch. */
int omni_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                float* running_mean, float* running_var, float* mean_rstd, float* scale_shift, double* ws,
                int P, int C, float eps, float momentum, int relu, void* stream);

/* eval-mode / frozen BatchNorm (solver/build.py:71-76 freeze_bn): y = relu?(x*scale+shift(+res)). */
int omni_bn_apply(const float* x, const float* scale_shift, const float* residual, float* y, int P, int C,
                  int relu, void* stream);

/* backward of omni_bn_fwd.  dy = grad wrt y; dres [nullable] = grad wrt residual;
 * ws >= 2C doubles, coef 3C floats scratch. */
int omni_bn_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* mean_rstd,
                float* dx, float* dres, float* dgamma, float* dbeta, double* ws, float* coef, int P, int C,
                int relu, void* stream);

/* nn.MaxPool2d(2, stride=2) (dla.py:209) forward / backward, NHWC. */
int omni_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* F.max_pool2d(x, kernel_size=1, stride=2) (dla.py:474, resnet.py:55): y[n,oh,ow]=x[n,2oh,2ow]. */
int omni_subsample2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int omni_subsample2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* detectron2 FPN top-down step: out = lateral + F.interpolate(top, 2.0, "nearest"); and the
 * gradient wrt `top` (2x2 block sums). */
int omni_upsample2_add(const float* lat, const float* top, float* out, int N, int H, int W, int C,
                       void* stream);
int omni_upsample2_bwd(const float* dout, float* dtop, int N, int H, int W, int C, void* stream);

/* GeneralizedRCNN.preprocess_image (called at cubercnn/modeling/meta_arch/rcnn3d.py:46,87):
 * uint8 planar (N,3,H,W) -> fp32 NHWC (N,PH,PW,4), (v-mean)/std, channel 3 and padding = 0. */
int omni_preprocess(const unsigned char* img, float* out, int N, int H, int W, int PH, int PW, float m0,
                    float m1, float m2, float s0, float s1, float s2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNI3D_HIP_H */
