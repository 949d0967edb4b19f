// Non-GEMM kernels of the GenConViT path: HBM-bound NHWC kernels with
// fp32 math, 64-lane wave reductions and LDS-staged cross-channel statistics.
//   K3  stem conv4x4s4 + LayerNorm2d            stem_ln_kernel
//   K4  depthwise 7x7 + LayerNorm               dwconv7_ln_kernel
//   K6  LayerNorm2d + 2x2 space-to-depth        ln_patchify_kernel   (conv2x2s2 itself is a GEMM)
//   K7  global-avg-pool + LayerNorm2d           pool_ln_kernel       (fc is a GEMM)
//   K1/K9 first 3->16 conv (Cin=3 is not MFMA-shaped)   conv3_first_kernel
//   K2/K12 last 16->3 ConvTranspose2d           convt2_small_kernel
//   K10/K11 split-K reduce + bias + reparameterise      reparam_kernel, kl_rows_kernel
//   K8  500->2 head tail                        head_tail_kernel
//   K13/K14 bilinear 112->224 (+ per-frame MSE) resize_mse_kernel
//   K15 sigmoid -> mean over rows               vote_kernel
#pragma once
#include "common.h"

namespace gcv {

// ------------------------------------------------------------------ launch wrappers (defined in kernels_impl.h)
template <typename T> int launch_stem_ln(const T* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* wp,
                                         const float* bias, const float* lnw, const float* lnb, T* out, int nimg,
                                         int Ho, int Wo, float eps, hipStream_t s);
template <typename T> int launch_dwconv7_ln(const T* x, const float* wdw, const float* bdw, const float* lnw,
                                            const float* lnb, T* y, int nimg, int H, int W, int C, float eps,
                                            hipStream_t s);
// rolling-strip variant (dwconv_roll.h): W % 7 == 0 shapes of ConvNeXt-T; `applicable` says whether it covers a shape
template <typename T> bool dwconv_roll_applicable(int H, int W, int C);
template <typename T> int launch_dwconv7_ln_roll(const T* x, const float* wdw, const float* bdw, const float* lnw,
                                                 const float* lnb, T* y, int nimg, int H, int W, int C, float eps,
                                                 hipStream_t s);
template <typename T> int launch_ln_patchify(const T* x, const float* w, const float* b, T* out, int nimg, int H,
                                             int W, int C, float eps, hipStream_t s);
template <typename T> int launch_layernorm_rows(const T* x, const float* w, const float* b, T* out, int64_t rows,
                                                int C, float eps, hipStream_t s);
// seg_n > 0: image b of the launch -> output row (b % seg_n) * row_stride + row0 + b / seg_n (see pool_ln_kernel)
template <typename T> int launch_pool_ln(const T* x, const float* w, const float* b, T* out, int nimg, int HW, int C,
                                         float eps, hipStream_t s, int seg_n = 0, int row_stride = 1, int row0 = 0);
template <typename T> int launch_conv3_first(const T* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                             const float* wp, const float* bias, T* out, int nimg, int H, int W,
                                             bool pool, int act, hipStream_t s);
template <typename T> int launch_convt2_small(const T* x, const float* wp, const float* bias, T* out, int nimg, int H,
                                              int W, int act, hipStream_t s);
template <typename T> int launch_reparam(const float* partial, int splitk, const float* bias, const float* eps,
                                         float* mu_out, T* z_nhwc, int B, int N, hipStream_t s);
template <typename T> int launch_head_tail(const T* h, const float* w, const float* bias, float* logits, int B, int K,
                                           hipStream_t s);
// the same with the hidden layer's split-K partials (S, B, K) fp32 reduced on the way in: h = act(sum + b1)
template <typename T> int launch_head_tail_splitk(const float* partial, int S, const float* b1, int act, const float* w,
                                                  const float* bias, float* logits, int B, int K, hipStream_t s);
template <typename T> int launch_resize_mse(const T* xhat, const T* img, T* recon, float* msepart, float* mse, int B,
                                            hipStream_t s);
template <typename T> int launch_swin_window_attn(const T* qkv, const float* rpb, T* out, int nimg, int H, int W, int C,
                                                  int nH, int shift, hipStream_t s);
template <typename T> int launch_patch_merge_ln(const T* x, const float* w, const float* b, T* out, int nimg, int H,
                                                int W, int C, float eps, hipStream_t s);
template <typename T> int launch_mean_tokens(const T* x, T* out, int nimg, int L, int C, hipStream_t s);
template <typename T> int launch_preprocess(const unsigned char* u8, T* out, int n, int H, int W, hipStream_t s);
// N4 (face.hip): crop + cv2.INTER_AREA resize of n face boxes, uint8 RGB
int launch_face_crop_resize(const unsigned char* frames, int nframes, int H, int W, const int* boxes5, int n,
                            unsigned char* out, int S, hipStream_t s);
int launch_kl(const float* partial, int splitk, const float* bias, const float* mu, float* rowsum, float* kl, int B,
              int N, hipStream_t s);
int launch_vote(const float* logits, int rows, float* mean2, hipStream_t s);
int launch_vote_segments(const float* logits, int B, int nets, const int* off, int nvid, float* mean2, hipStream_t s);

}  // namespace gcv
