// Swin-T (row A6 of SURVEY.md §8a): timm 0.6.5 swin_tiny_patch4_window7_224 — the HybridEmbed
// "embedder" the reference constructs (model/genconvit_ed.py:69-70, model/genconvit_vae.py:96,98) and
// runs exactly once, at construction, to probe output dims (model/model_embedder.py:22).  It never
// feeds the logits, so this path is measured and parity-checked on its own.
//   patch embed  : conv4x4 s4 + LN(96, eps 1e-5)                      -> stem_ln_kernel
//   block        : LN -> qkv GEMM -> window attention (rel-pos bias, shift mask) -> proj GEMM + residual
//                  LN -> fc1 GEMM + GELU -> fc2 GEMM + residual
//   patch merging: 2x2 gather + LN(4C) -> Linear(4C, 2C, no bias)
//   head         : LN(768) -> mean over 49 tokens -> Linear(768, 1000)
#pragma once
#include "gemm.h"
#include "kernels.h"
#include "net.h"

namespace gcv {

template <typename T> struct SwinBlockW {
  float *n1_w, *n1_b, *rpb, *qkv_b, *proj_b, *n2_w, *n2_b, *fc1_b, *fc2_b;
  T *qkv_w, *proj_w, *fc1_w, *fc2_w;
};
template <typename T> struct SwinW {
  float *pe_w, *pe_b, *pe_lnw, *pe_lnb;      // patch embed [48][96] + LN
  SwinBlockW<T> blk[12];
  struct { float *ln_w, *ln_b; T* red_w; } merge[3];
  float *norm_w, *norm_b, *head_b, *ones;    // ones: unit "layer scale" for the residual epilogue
  T* head_w;
};

static const int kSwinDims[4] = {96, 192, 384, 768};
static const int kSwinDepths[4] = {2, 2, 6, 2};
static const int kSwinHeads[4] = {3, 6, 12, 24};

template <typename T, typename Net> int pack_swin(Net& net, const TensorMap& w, const std::string& p, SwinW<T>& o) {
  auto& st = net.ws_swin;
  {
    std::vector<float> v, t(48 * 96);
    GCV_TRY(Net::fetch(w, p + "patch_embed.proj.weight", 96 * 48, v));
    for (int co = 0; co < 96; ++co)
      for (int k = 0; k < 48; ++k) t[k * 96 + co] = v[co * 48 + k];
    GCV_UP(o.pe_w, st, t);
  }
  GCV_TRY(net.up_f32(w, p + "patch_embed.proj.bias", 96, st, o.pe_b));
  GCV_TRY(net.up_f32(w, p + "patch_embed.norm.weight", 96, st, o.pe_lnw));
  GCV_TRY(net.up_f32(w, p + "patch_embed.norm.bias", 96, st, o.pe_lnb));
  int bi = 0;
  for (int i = 0; i < 4; ++i) {
    const int C = kSwinDims[i], nH = kSwinHeads[i];
    for (int j = 0; j < kSwinDepths[i]; ++j, ++bi) {
      const std::string b = p + "layers." + std::to_string(i) + ".blocks." + std::to_string(j) + ".";
      SwinBlockW<T>& k = o.blk[bi];
      GCV_TRY(net.up_f32(w, b + "norm1.weight", C, st, k.n1_w));
      GCV_TRY(net.up_f32(w, b + "norm1.bias", C, st, k.n1_b));
      GCV_TRY(net.up_f32(w, b + "attn.relative_position_bias_table", 169 * nH, st, k.rpb));
      GCV_TRY(net.up_cast(w, b + "attn.qkv.weight", (int64_t)3 * C * C, st, k.qkv_w));
      GCV_TRY(net.up_f32(w, b + "attn.qkv.bias", 3 * C, st, k.qkv_b));
      GCV_TRY(net.up_cast(w, b + "attn.proj.weight", (int64_t)C * C, st, k.proj_w));
      GCV_TRY(net.up_f32(w, b + "attn.proj.bias", C, st, k.proj_b));
      GCV_TRY(net.up_f32(w, b + "norm2.weight", C, st, k.n2_w));
      GCV_TRY(net.up_f32(w, b + "norm2.bias", C, st, k.n2_b));
      GCV_TRY(net.up_cast(w, b + "mlp.fc1.weight", (int64_t)4 * C * C, st, k.fc1_w));
      GCV_TRY(net.up_f32(w, b + "mlp.fc1.bias", 4 * C, st, k.fc1_b));
      GCV_TRY(net.up_cast(w, b + "mlp.fc2.weight", (int64_t)4 * C * C, st, k.fc2_w));
      GCV_TRY(net.up_f32(w, b + "mlp.fc2.bias", C, st, k.fc2_b));
    }
    if (i < 3) {
      const std::string d = p + "layers." + std::to_string(i) + ".downsample.";
      GCV_TRY(net.up_f32(w, d + "norm.weight", 4 * C, st, o.merge[i].ln_w));
      GCV_TRY(net.up_f32(w, d + "norm.bias", 4 * C, st, o.merge[i].ln_b));
      GCV_TRY(net.up_cast(w, d + "reduction.weight", (int64_t)2 * C * 4 * C, st, o.merge[i].red_w));
    }
  }
  GCV_TRY(net.up_f32(w, p + "norm.weight", 768, st, o.norm_w));
  GCV_TRY(net.up_f32(w, p + "norm.bias", 768, st, o.norm_b));
  GCV_TRY(net.up_cast(w, p + "head.weight", 1000 * 768, st, o.head_w));
  GCV_TRY(net.up_f32(w, p + "head.bias", 1000, st, o.head_b));
  std::vector<float> ones(768, 1.0f);
  GCV_UP(o.ones, st, ones);
  return 0;
}

template <typename T, typename Net> int run_swin(Net& net, const SwinW<T>& w, const T* x, int B, T* logits1000) {
  auto& ar = net.arena;
  hipStream_t cur = net.cur;
  int H = 56;
  int64_t M = (int64_t)B * H * H;
  const size_t mk = ar.mark();
  T* X = ar.template get<T>(M * 96);
  T* Y = ar.template get<T>(M * 96);
  T* QKV = ar.template get<T>(M * 288);
  T* ATT = ar.template get<T>(M * 96);
  T* Hd = ar.template get<T>(M * 384);
  T* Pool = ar.template get<T>((int64_t)B * 768);
  if (!ar.dry && ar.overflow) { set_error("workspace arena too small"); return -6; }

  GCV_TRY(net.run("swin.patch_embed_ln", 2.0 * M * 96 * 48, sizeof(T) * (double)M * (48 + 96), [&] {
    return launch_stem_ln<T>(x, (int64_t)3 * 224 * 224, 224 * 224, 224, 1, w.pe_w, w.pe_b, w.pe_lnw, w.pe_lnb, X, B, H, H,
                             1e-5f, cur);
  }));
  int bi = 0;
  for (int i = 0; i < 4; ++i) {
    const int C = kSwinDims[i], nH = kSwinHeads[i];
    for (int j = 0; j < kSwinDepths[i]; ++j, ++bi) {
      const SwinBlockW<T>& k = w.blk[bi];
      const int shift = (j % 2 == 1 && H > 7) ? 3 : 0;
      GCV_TRY(net.run("swin.ln", 8.0 * M * C, 2.0 * sizeof(T) * (double)M * C,
                      [&] { return launch_layernorm_rows<T>(X, k.n1_w, k.n1_b, Y, M, C, 1e-5f, cur); }));
      GemmArgs q{};
      q.A = Y; q.lda = C; q.Wt = k.qkv_w; q.C = QKV; q.ldc = 3 * C; q.bias = k.qkv_b;
      q.M = (int)M; q.N = 3 * C; q.K = C; q.act = ACT_NONE; q.splitk = 1;
      GCV_TRY(net.gemm("swin.qkv_gemm", q, A_PLAIN, EPI_BIAS_ACT));
      GCV_TRY(net.run("swin.window_attn", 4.0 * M * 49 * C, sizeof(T) * (double)M * 4 * C, [&] {
        return launch_swin_window_attn<T>(QKV, k.rpb, ATT, B, H, H, C, nH, shift, cur);
      }));
      GemmArgs pj{};
      pj.A = ATT; pj.lda = C; pj.Wt = k.proj_w; pj.C = X; pj.ldc = C; pj.bias = k.proj_b; pj.gamma = w.ones;
      pj.resid = X; pj.M = (int)M; pj.N = C; pj.K = C; pj.act = ACT_NONE; pj.splitk = 1;
      GCV_TRY(net.gemm("swin.proj_res_gemm", pj, A_PLAIN, EPI_RESID));
      GCV_TRY(net.run("swin.ln", 8.0 * M * C, 2.0 * sizeof(T) * (double)M * C,
                      [&] { return launch_layernorm_rows<T>(X, k.n2_w, k.n2_b, Y, M, C, 1e-5f, cur); }));
      GemmArgs f1{};
      f1.A = Y; f1.lda = C; f1.Wt = k.fc1_w; f1.C = Hd; f1.ldc = 4 * C; f1.bias = k.fc1_b;
      f1.M = (int)M; f1.N = 4 * C; f1.K = C; f1.act = ACT_GELU; f1.splitk = 1;
      GCV_TRY(net.gemm("swin.fc1_gelu_gemm", f1, A_PLAIN, EPI_BIAS_ACT));
      GemmArgs f2{};
      f2.A = Hd; f2.lda = 4 * C; f2.Wt = k.fc2_w; f2.C = X; f2.ldc = C; f2.bias = k.fc2_b; f2.gamma = w.ones;
      f2.resid = X; f2.M = (int)M; f2.N = C; f2.K = 4 * C; f2.act = ACT_NONE; f2.splitk = 1;
      GCV_TRY(net.gemm("swin.fc2_res_gemm", f2, A_PLAIN, EPI_RESID));
    }
    if (i < 3) {
      GCV_TRY(net.run("swin.patch_merge_ln", 8.0 * M * C, 2.0 * sizeof(T) * (double)M * C, [&] {
        return launch_patch_merge_ln<T>(X, w.merge[i].ln_w, w.merge[i].ln_b, Y, B, H, H, C, 1e-5f, cur);
      }));
      H /= 2;
      M = (int64_t)B * H * H;
      GemmArgs r{};
      r.A = Y; r.lda = 4 * C; r.Wt = w.merge[i].red_w; r.C = X; r.ldc = 2 * C; r.bias = nullptr;
      r.M = (int)M; r.N = 2 * C; r.K = 4 * C; r.act = ACT_NONE; r.splitk = 1;
      GCV_TRY(net.gemm("swin.merge_gemm", r, A_PLAIN, EPI_BIAS_ACT));
    }
  }
  GCV_TRY(net.run("swin.ln", 8.0 * M * 768, 2.0 * sizeof(T) * (double)M * 768,
                  [&] { return launch_layernorm_rows<T>(X, w.norm_w, w.norm_b, Y, M, 768, 1e-5f, cur); }));
  GCV_TRY(net.run("swin.mean_tokens", 1.0 * M * 768, sizeof(T) * (double)M * 768,
                  [&] { return launch_mean_tokens<T>(Y, Pool, B, 49, 768, cur); }));
  GemmArgs hd{};
  hd.A = Pool; hd.lda = 768; hd.Wt = w.head_w; hd.C = logits1000; hd.ldc = 1000; hd.bias = w.head_b;
  hd.M = B; hd.N = 1000; hd.K = 768; hd.act = ACT_NONE; hd.splitk = 1;
  GCV_TRY(net.gemm("swin.head_gemm", hd, A_PLAIN, EPI_BIAS_ACT));
  ar.release(mk);
  return 0;
}

}  // namespace gcv
