/* genconvit_hip.h — C ABI of libgenconvit_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the GenConViT `ed` + `vae` inference forward.  The reference
 * (ctxnn/GenConViT, pure PyTorch) has no FFI layer; each entry point below replaces the
 * Python/ATen call it cites (paths relative to the reference root).  A maintainer binds
 * these with ctypes/cffi exactly as genconvit_amd/_lib.py does.
 *
 * Conventions
 *   - all pointers are device pointers (e.g. torch.Tensor.data_ptr()) unless stated otherwise
 *   - `gcv_stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); work is enqueued,
 *     never synchronised, by the forward calls
 *   - return value 0 = ok; non-zero = error, text via gcv_last_error() (thread-local)
 *   - a handle owns its packed weights + workspace; not thread-safe; one handle per stream
 *   - the handle-based calls (gcv_*_forward, gcv_load_*) make the handle's device current for their launches and
 *     restore the caller's device before returning; the handle-less calls (gcv_vote*, gcv_preprocess, gcv_k_*)
 *     launch on `stream` as given: the device that owns that stream must be the current one
 *   - frames: (B,3,224,224) NCHW contiguous in the handle's storage dtype, already normalised like
 *     model/pred_func.py:95-108 (preprocess_frame); logits are always fp32 (B,2)
 */
#ifndef GENCONVIT_HIP_H
#define GENCONVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(GCV_BUILD)
#pragma GCC visibility push(default)
#endif

typedef struct gcv_handle gcv_handle;
typedef void* gcv_stream;

enum { GCV_F32 = 0, GCV_BF16 = 1, GCV_F16 = 2 };               /* storage dtype of a handle      */
enum { GCV_ACT_NONE = 0, GCV_ACT_RELU = 1, GCV_ACT_GELU = 2, GCV_ACT_LEAKY = 3 };

/* One named fp32 tensor of a reference state_dict (key names as in weight/{ed,vae}.pth,
 * SURVEY.md Appendix A.3).  `data` may be host or device memory. */
typedef struct {
  const char* name;
  const void* data;     /* fp32, contiguous */
  int64_t numel;
  int on_device;
} gcv_tensor_desc;

const char* gcv_last_error(void);

/* Handle lifetime.  Replaces GenConViT.__init__'s module construction + .to(device)/.half()
 * (model/genconvit.py:9-64, model/pred_func.py:50-62).  `max_batch` sizes the workspace arena. */
int  gcv_create(gcv_handle** h, int device, int dtype, int max_batch);
void gcv_destroy(gcv_handle* h);
size_t gcv_workspace_bytes(const gcv_handle* h);

/* Weight loading: replaces load_state_dict of GenConViTED / GenConViTVAE
 * (model/genconvit.py:16-21,30-35,47-56).  Tensors are re-packed once (conv weights to GEMM
 * order, BatchNorm folded, mu/var columns permuted to NHWC, cast to the handle dtype).
 * Keys that never run in forward (embedder.*, *.patch_embed.*, encoder.fc1/fc2, fc3,
 * num_batches_tracked) are ignored; a missing on-path key is an error. */
int gcv_load_ed (gcv_handle* h, const gcv_tensor_desc* w, int n);
int gcv_load_vae(gcv_handle* h, const gcv_tensor_desc* w, int n);
/* Swin-T embedder weights (timm swin_tiny_patch4_window7_224 keys under `prefix`, e.g. "embedder.");
 * constructed-but-never-executed in the reference forward (model/genconvit_ed.py:69-70). */
int gcv_load_swin(gcv_handle* h, const gcv_tensor_desc* w, int n, const char* prefix);

/* GenConViTED.forward (model/genconvit_ed.py:77-88): logits[b] = fc2(gelu(fc(gelu(cat(
 * backbone(decoder(encoder(x))), backbone(x)))))). */
int gcv_ed_forward(gcv_handle* h, const void* x_nchw, int batch, float* logits, gcv_stream stream);

/* GenConViTVAE.forward (model/genconvit_vae.py:107-116) with the encoder's single
 * torch.randn_like draw (model/genconvit_vae.py:46) made an explicit fp32 input `eps` (B,12544).
 *   recon224 : nullable, (B,3,224,224) in the handle dtype = transforms.Resize((224,224))(x_hat) (:116)
 *   mse      : nullable, (B) fp32 per-frame mean((recon224 - x)^2); its mean is the reference's
 *              nn.MSELoss()(recons, images) (train/train_vae.py:24,76)
 *   kl       : nullable, (1) fp32 = Encoder.kl (model/genconvit_vae.py:58)
 * backbone(x) (:111) depends on nothing but the input: called on its own, gcv_vae_forward enqueues it on a side stream
 * owned by the handle, forked from `stream` at the call and joined back into it by an event before the head, while the
 * encoder / decoder chain and backbone(x_hat) run on `stream` itself; inside gcv_genconvit_forward everything of the VAE
 * stays on one stream.  GCV_VAE_SPLIT=0 / 1 (read when the handle is created) forces the merged / the split schedule.
 * Either way nothing changes for the caller: all work is ordered after what `stream` held at the call and before what
 * it is given next; no synchronisation. */
int gcv_vae_forward(gcv_handle* h, const void* x_nchw, const float* eps, int batch, float* logits,
                    void* recon224, float* mse, float* kl, gcv_stream stream);

/* GenConViT.forward for net = 'genconvit' (model/genconvit.py:66-75): logits_2Bx2 = cat((ed(x), vae(x)[0]), dim=0),
 * rows 0..B-1 = ED logits of frames 0..B-1, rows B..2B-1 = VAE logits.  The two networks are independent until the
 * concat: they run on two internal streams forked from `stream` and joined back into it with events, so a C caller
 * gets the overlapped step (one network's small-grid kernels fill the other's tails) without managing streams.
 * `h_ed` / `h_vae`: two handles of one device and dtype with gcv_load_ed / gcv_load_vae done (each owns its own
 * workspace).  eps as in gcv_vae_forward. */
int gcv_genconvit_forward(gcv_handle* h_ed, gcv_handle* h_vae, const void* x_nchw, const float* eps, int batch,
                          float* logits_2Bx2, gcv_stream stream);

/* timm convnext_tiny forward alone (call sites model/genconvit_ed.py:82-83,
 * model/genconvit_vae.py:111-112): which = 0 the ED backbone, 1 the VAE backbone;
 * x (B,3,res,res) -> logits1000 (B,1000) in the handle dtype. */
int gcv_convnext_forward(gcv_handle* h, int which, const void* x_nchw, int batch, int res,
                         void* logits1000, gcv_stream stream);

/* timm swin_tiny_patch4_window7_224 forward (only executed by HybridEmbed.__init__,
 * model/model_embedder.py:22): x (B,3,224,224) -> (B,1000) in the handle dtype. */
int gcv_swin_forward(gcv_handle* h, const void* x_nchw, int batch, void* logits1000, gcv_stream stream);

/* preprocess_frame (model/pred_func.py:95-108 + the "vid" Normalize of dataset/loader.py:63-65,77) on the device:
 * uint8 NHWC face crops (n,H,W,3) -> ((x/255) - mean) / std as NCHW in `dtype` (row N1 of SURVEY.md §8f). */
int gcv_preprocess(int dtype, const void* frames_u8_nhwc, void* out_nchw, int n, int H, int W, gcv_stream s);

/* Row N4: the crop + resize of face_rec (model/pred_func.py:79-85),
 *   cv2.resize(frame[top:bottom, left:right], (224, 224), interpolation=cv2.INTER_AREA),
 * for n faces in one launch.  frames: (nframes,H,W,3) uint8 RGB on the device; boxes5: int32 device array of n rows
 * (frame index, top, right, bottom, left) — face_recognition's (top, right, bottom, left) order behind the index of the
 * frame the face was found in; out: (n,size,size,3) uint8.  The reference's RGB<->BGR swaps around the resize cancel.
 * A row outside its frame produces zeros (and reads nothing).  Face DETECTION (dlib) and video decode (decord) stay
 * third-party CPU code on the caller's side. */
int gcv_face_crop_resize(const void* frames_u8_nhwc, int nframes, int H, int W, const int* boxes5, int n,
                         void* out_u8_nhwc, int size, gcv_stream s);

/* pred_vid's reduction (model/pred_func.py:120,125): mean2[c] = mean_r sigmoid(logits[r][c]). */
int gcv_vote(const float* logits, int rows, float* mean2, gcv_stream stream);

/* Row N3: the same vote for several videos batched into ONE forward.  logits rows are [net 0 frames 0..B-1; net 1 ...]
 * (model/genconvit.py:74); video v owns frames [offsets[v], offsets[v+1]) (int32 device array of n_videos+1 entries);
 * mean2[v][c] = mean over its frames and nets of sigmoid(logit[.][c]) — what max_prediction_value reduces per video. */
int gcv_vote_segments(const float* logits, int batch, int nets, const int* offsets, int n_videos, float* mean2,
                      gcv_stream stream);

/* ---- multi-GPU: frame shards, one process per GPU, RCCL over xGMI (new capability: the reference is single
 * device, model/pred_func.py:15).  The only exchange of the path is one all-gather of per-frame logits before the
 * vote.  RCCL is bound at run time (dlopen: $GCV_RCCL_PATH, an already loaded librccl, then
 * /opt/rocm/lib) so that the library shares the process's RCCL the way it shares its HIP runtime.
 *   gcv_comm_available : 1 when RCCL and its entry points could be bound in this process (dlopen + dlsym only), else 0
 *                        with the reason in gcv_last_error(): the cheap probe every rank runs before the collective
 *   gcv_comm_count     : ranks in the communicator as RCCL reports them (ncclCommCount)
 *   gcv_comm_unique_id : rank 0 fills 128 bytes (ncclUniqueId); the caller ships them to every rank (any channel)
 *   gcv_comm_create    : collective over all `world` ranks (ncclCommInitRank) on `device`
 *   gcv_allgather_logits: all[r * n_local .. (r+1) * n_local) = rank r's `local` (n_local floats, equal on all
 *                        ranks: pad ragged shards), enqueued on `stream`; world = 1 is a device copy */
typedef struct gcv_comm gcv_comm;
int  gcv_comm_available(void);
int  gcv_comm_count(gcv_comm* c);
int  gcv_comm_unique_id(void* id128);
int  gcv_comm_create(gcv_comm** c, int world, int rank, const void* id128, int device);
void gcv_comm_destroy(gcv_comm* c);
int  gcv_allgather_logits(gcv_comm* c, const float* local, int n_local, float* all, gcv_stream stream);

/* Per-launch timing with HIP events on the launch stream.  After gcv_profile_enable(h,1) every
 * kernel launch of the following forwards is bracketed by events; gcv_profile_report() waits for
 * them and returns a JSON array aggregated by op tag (launches, ms, algorithmic flops / bytes)
 * and clears the records.  The returned string lives until the next call on the handle.  While profiling is on, every
 * tagged launch is also a roctx range of the same name (roctxRangePushA / roctxRangePop, bound at run time from
 * librocprofiler-sdk-roctx / libroctx64 when present), so `rocprofv3 --marker-trace --kernel-trace` groups kernels by op. */
int gcv_profile_enable(gcv_handle* h, int on);
const char* gcv_profile_report(gcv_handle* h);

/* ---- per-kernel entry points (unit parity tests; same kernels the forwards launch) ---- */
enum { GCV_A_PLAIN = 0, GCV_A_IM2COL3_POOL = 1, GCV_A_IM2COL3_S2 = 2 };
enum { GCV_EPI_BIAS_ACT = 0, GCV_EPI_RESID = 1, GCV_EPI_POOL4 = 2, GCV_EPI_CONVT = 3, GCV_EPI_SPLITK = 4 };

typedef struct {
  const void* A; const void* Wt; void* C;
  const float* bias; const float* gamma; const void* resid; float* partial;
  int M, N, K, lda, ldc, act, splitk, k_per_split, H, W, cin_log2, cout_log2;
} gcv_gemm_args;

/* C = epilogue(A * Wt^T): nn.Linear / Conv2d-as-GEMM / ConvTranspose2d(k=s=2) / split-K slab */
int gcv_k_gemm(int dtype, int a_mode, int epi, const gcv_gemm_args* a, gcv_stream s);
int gcv_k_stem_ln(int dtype, const void* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* wp,
                  const float* bias, const float* lnw, const float* lnb, void* out, int nimg, int Ho, int Wo,
                  float eps, gcv_stream s);
int gcv_k_dwconv7_ln(int dtype, const void* x, const float* wdw, const float* bdw, const float* lnw,
                     const float* lnb, void* y, int nimg, int H, int W, int C, float eps, gcv_stream s);
int gcv_k_ln_patchify(int dtype, const void* x, const float* w, const float* b, void* out, int nimg, int H, int W,
                      int C, float eps, gcv_stream s);
int gcv_k_layernorm_rows(int dtype, const void* x, const float* w, const float* b, void* out, int64_t rows, int C,
                         float eps, gcv_stream s);
int gcv_k_pool_ln(int dtype, const void* x, const float* w, const float* b, void* out, int nimg, int HW, int C,
                  float eps, gcv_stream s);
int gcv_k_conv3_first(int dtype, const void* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* wp,
                      const float* bias, void* out, int nimg, int H, int W, int pool, int act, gcv_stream s);
int gcv_k_convt2_small(int dtype, const void* x, const float* wp, const float* bias, void* out, int nimg, int H,
                       int W, int act, gcv_stream s);
int gcv_k_reparam(int dtype, const float* partial, int splitk, const float* bias, const float* eps, float* mu_out,
                  void* z_nhwc, int B, int N, gcv_stream s);
int gcv_k_head_tail(int dtype, const void* h, const float* w, const float* bias, float* logits, int B, int K,
                    gcv_stream s);
/* The head as the networks run it (model/genconvit_ed.py:87, model/genconvit_vae.py:114: fc2(act(fc(.)))): the hidden layer's
 * split-K partials (splitk, B, K) fp32 are reduced on the way in, h = act(sum + b1) rounded to the storage dtype, then
 * logits = h . w^T + bias with w (2, K), in one kernel.  act: GCV_ACT_* code. */
int gcv_k_head_tail_splitk(int dtype, const float* partial, int splitk, const float* b1, int act, const float* w,
                           const float* bias, float* logits, int B, int K, gcv_stream s);
int gcv_k_resize_mse(int dtype, const void* xhat, const void* img, void* recon, float* msepart, float* mse, int B,
                     gcv_stream s);

/* Swin-T pieces (timm swin_tiny_patch4_window7_224): W-MSA / SW-MSA over 7x7
 * windows with relative-position bias + shift mask on a (B,H,W,3C) qkv tensor; PatchMerging's
 * 2x2 gather + LayerNorm(4C); mean over tokens. */
int gcv_k_swin_window_attn(int dtype, const void* qkv, const float* rpb, void* out, int nimg, int H, int W, int C,
                           int nH, int shift, gcv_stream s);
int gcv_k_patch_merge_ln(int dtype, const void* x, const float* w, const float* b, void* out, int nimg, int H, int W,
                         int C, float eps, gcv_stream s);
int gcv_k_mean_tokens(int dtype, const void* x, void* out, int nimg, int L, int C, gcv_stream s);

/* Fused ConvNeXt MLP for C = 96 / 192, 16-bit storage:
 * out = resid + gamma * (W2 . GELU(W1 . x + b1) + b2) with the 4C hidden activation kept on chip
 * (timm ConvNeXtBlock: mlp.fc1 -> GELU -> mlp.fc2 -> * gamma -> + shortcut).  w2_f32: (C,4C) fp32 device. */
int gcv_k_fused_mlp(int dtype, int C, const void* x, const void* w1, const float* b1, const float* w2_f32,
                    const float* b2, const float* gamma, const void* resid, void* out, int M, gcv_stream s);
/* The last block of ConvNeXt stage 0 / 1 as the network runs it: the MLP above with the stage boundary's
 * `downsample` LayerNorm2d + the 2x2 space-to-depth of its Conv2d(k=2, s=2) in the epilogue (timm ConvNeXtStage.downsample,
 * SURVEY A.1; call sites model/genconvit_ed.py:82-83, model/genconvit_vae.py:111-112).  Tokens [tok0[i], tok0[i+1]) (the
 * last segment ends at M) are whole images of hw[i] = H*W pixels, W = wd[i], H and W even; `out` receives the patch rows
 * (M/4, 4C), (dy, dx, c) innermost, segment i starting at row out0[i] = tok0[i] / 4; the (M, C) residual stream is not
 * written.  C = 96 (M >= 65536 tokens: the LDS-resident kernel) or C = 192; nseg <= 4; host int arrays. */
int gcv_k_fused_mlp_lnp(int dtype, int C, const void* x, const void* w1, const float* b1, const float* w2_f32,
                        const float* b2, const float* gamma, const void* resid, const float* ln_w, const float* ln_b,
                        float eps, int nseg, const int* tok0, const int* hw, const int* wd, const int* out0, void* out, int M,
                        gcv_stream s);
/* The same launch, timed: the weights are packed once, then `iters` launches of the MLP kernel(s) alone are bracketed by
 * HIP events on `s` (the call synchronises).  ms3[0] = average ms per MLP; for the C = 384 kernel pair ms3[1] / ms3[2] are
 * pw1+GELU / pw2+scale+residual timed separately, else 0.  Used by profiles/microbench.py; `out` may alias `resid`. */
int gcv_k_fused_mlp_timed(int dtype, int C, const void* x, const void* w1, const float* b1, const float* w2_f32,
                          const float* b2, const float* gamma, const void* resid, void* out, int M, int iters,
                          float* ms3, gcv_stream s);

#if defined(GCV_BUILD)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* GENCONVIT_HIP_H */
