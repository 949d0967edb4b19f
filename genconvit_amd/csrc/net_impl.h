// NetImpl<T>: weight packing + forward schedules (included by net_{f32,f16,bf16}.hip).
//
// Data layout in HBM
//   activations : NHWC "token major" (tokens, C) in T; several images / passes that share a weight
//                 set are concatenated along the token axis so pointwise GEMMs run once over all of them
//   GEMM weights: (N, K) row-major in T (nn.Linear layout); conv weights re-ordered to K = (ky,kx,ci)
//   small params: fp32 (biases, LayerNorm affine, layer-scale gamma, depthwise taps [49][C])
//   workspace   : one arena sized by a dry run of both forwards at max_batch (no allocation per call)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fused_mlp.h"
#include "mlp_pair.h"
#include "xs_mlp.h"
#include "gemm.h"
#include "kernels.h"
#include "net.h"
#include "swin.h"

namespace gcv {

// ---------------------------------------------------------------- host conversions
template <typename T> inline T host_cvt(float v);
template <> inline float host_cvt<float>(float v) { return v; }
template <> inline half_t host_cvt<half_t>(float v) { return (half_t)v; }
template <> inline bf16_t host_cvt<bf16_t>(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  uint16_t h;
  if ((u & 0x7fffffffu) > 0x7f800000u) h = (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  else h = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);              // round to nearest even
  bf16_t r;
  std::memcpy(&r, &h, 2);
  return r;
}

// ---------------------------------------------------------------- device memory owners
struct WeightStore {
  std::vector<void*> ptrs;
  size_t bytes = 0;
  ~WeightStore() { clear(); }
  void clear() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
    bytes = 0;
  }
  void* raw(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n ? n : 16) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    bytes += n;
    return p;
  }
  template <typename U> U* upload(const std::vector<U>& h) {
    U* p = (U*)raw(h.size() * sizeof(U));
    if (!p) return nullptr;
    if (hipMemcpy(p, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
  }
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;
  bool overflow = false;
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  void* alloc(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    peak = std::max(peak, off);
    if (!dry && off > cap) { overflow = true; return base; }
    return base + a;
  }
  template <typename U> U* get(int64_t n) { return (U*)alloc((size_t)n * sizeof(U)); }
};

// device-side permute+cast for the 25088x12544 mu/var matrices: columns c*196+hw -> hw*128+c so the
// NHWC encoder output (B,14,14,128) can be used as the GEMM A operand without a transpose.
template <typename T>
__global__ void __launch_bounds__(256) pack_mu_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t total) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int64_t n = o / 25088;
  const int k = (int)(o - n * 25088);
  const int hw = k >> 7, c = k & 127;
  dst[o] = from_f<T>(src[n * 25088 + c * 196 + hw]);
}

// ---------------------------------------------------------------- packed weights
template <typename T> struct CnxBlockW {
  float *dw_w, *dw_b, *ln_w, *ln_b, *fc1_b, *fc2_b, *gamma;
  T *fc1_w, *fc2_w;
  T* fc2_wc;       // 16-bit, C <= 192: W2 packed for the fused MLP kernels (else null)
  T* fc_wp;        // 16-bit, C <= 192: W1 | W2 records of the x-stationary fused MLP (xs_mlp.h) (else null)
  T *fc1_wf, *fc2_wf;   // 16-bit, C = 384: W1 / W2 in MFMA-fragment order for the pw1 / pw2 kernel pair (else null)
};
template <typename T> struct CnxW {
  float *stem_w, *stem_b, *stem_lnw, *stem_lnb;
  CnxBlockW<T> blk[18];
  struct { float *ln_w, *ln_b, *b; T* w; } down[3];
  float *head_lnw, *head_lnb, *head_fc_b;
  T* head_fc_w;
};
template <typename T> struct HeadW {
  T* fc_w;        // (500, 2000)
  float* fc_b;
  float* fc2_w;   // [2][500]
  float* fc2_b;
};
template <typename T> struct EdW {
  float *enc1_w, *enc1_b;            // [27][16]
  T* enc_w[4];                       // layers 2..5 (Cout, 9*Cin)
  float* enc_b[4];
  T* dec_w[4];                       // layers 1..4 ((dy,dx,co), ci)
  float* dec_b[4];
  float *dec5_w, *dec5_b;            // [16][2][2][3]
  HeadW<T> head;
};
template <typename T> struct VaeW {
  float *enc1_w, *enc1_b;            // BN folded
  T* enc_w[3];
  float* enc_b[3];
  T *mu_w, *var_w;                   // (12544, 25088) K-permuted
  float *mu_b, *var_b;
  T* dec_w[3];
  float* dec_b[3];
  float *dec4_w, *dec4_b;
  HeadW<T> head;
};

static const int kDims[4] = {96, 192, 384, 768};
static const int kDepths[4] = {3, 3, 9, 3};

template <typename T> struct Seg {
  const T* x;
  int64_t sb, sc, sy, sx;   // element strides of the (n,3,H,W) input view
  int n, H, W;
  T* out;                   // backbone logits (n,1000) written at out + i*out_ld
  int out_ld;
  int act;
};

template <typename T> struct NetImpl : NetBase {
  WeightStore ws_ed, ws_vae, ws_swin;
  CnxW<T> bb_ed{}, bb_vae{};
  EdW<T> ed{};
  VaeW<T> vae{};
  SwinW<T> swin{};
  bool has_ed = false, has_vae = false, has_swin = false;
  Arena arena;
  hipStream_t cur = nullptr;
  bool use_fused_mlp = exp_env("GCV_NO_FUSED_MLP") == nullptr;   // A/B switches for profiling
  // (GCV_EXPERIMENTS builds only, see common.h exp_env: the round-2 kernels behind their old switches)
  bool use_fused_mlp384 = exp_env("GCV_FUSED_MLP384") != nullptr;
#ifdef GCV_NO_LNP_EPILOGUE
  bool use_lnp_epilogue = false;                         // A/B builds: LayerNorm-patchify as its own launch everywhere
#else
  bool use_lnp_epilogue = true;
#endif
  bool use_mlp_pair = exp_env("GCV_NO_MLP_PAIR") == nullptr;     // C = 384: pw1 / pw2 kernel pair (mlp_pair.h)
  bool use_xs_mlp = exp_env("GCV_MLP_LEGACY") == nullptr;        // C = 192: x-stationary fused MLP (xs_mlp.h)
  // Schedule of vae_forward.  SPLIT: backbone(x) — which depends on nothing but the input — runs on a side stream while
  // the encoder / mu GEMM / decoder chain (small, latency-bound launches) and then backbone(x_hat) run on the caller's
  // stream.  MERGED: one stream, one two-segment backbone pass.  Alone on the GPU the split hides the codec chain
  // (vae B=32 bf16: 18.3k vs 15.2k frames/s); inside gcv_genconvit_forward the ED network already fills those gaps and
  // the merged pass wins (7.49 vs 7.58-7.62 ms per step in paired runs: the round-3 MLP kernels pay per 256-token pass
  // and prefer one launch over two smaller ones).  Default: split when called on its own, merged inside the ensemble;
  // GCV_VAE_SPLIT=0 / 1 forces one of them.  Both are under test (tests/test_parity_gpu.py, fresh_vae).
  int vae_split_env = [] { const char* e = std::getenv("GCV_VAE_SPLIT"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
  hipStream_t vae_side = nullptr;                      // (one per handle; the ED and the VAE network are separate handles)
  hipEvent_t vae_fork = nullptr, vae_join = nullptr;
  // the caller's stream waits for the side stream before the head reads its half of `feat` — and on every early return,
  // so that whatever was enqueued is ordered before the caller's next work
  struct Join {
    hipStream_t s = nullptr; hipEvent_t ev = nullptr;
    void arm(hipStream_t s_, hipEvent_t ev_) { s = s_; ev = ev_; }
    void now() { if (ev) { (void)hipStreamWaitEvent(s, ev, 0); ev = nullptr; } }
    ~Join() { now(); }
  };
  // run one backbone pass on the side stream, forked from `s` here
  int side_pass(const CnxW<T>& w, const Seg<T>* seg, hipStream_t s, Join& join) {
    if (!arena.dry) {
      if (!vae_side) {
        GCV_CHECK_HIP(hipStreamCreateWithFlags(&vae_side, hipStreamNonBlocking));
        GCV_CHECK_HIP(hipEventCreateWithFlags(&vae_fork, hipEventDisableTiming));
        GCV_CHECK_HIP(hipEventCreateWithFlags(&vae_join, hipEventDisableTiming));
      }
      GCV_CHECK_HIP(hipEventRecord(vae_fork, s));
      GCV_CHECK_HIP(hipStreamWaitEvent(vae_side, vae_fork, 0));
      cur = vae_side;
    }
    const int rc = run_convnext(w, seg, 1, true);
    cur = s;
    if (!arena.dry && hipEventRecord(vae_join, vae_side) == hipSuccess) join.arm(s, vae_join);
    return rc;
  }

  ~NetImpl() override {
    if (arena.base) (void)hipFree(arena.base);
    if (vae_side) (void)hipStreamDestroy(vae_side);
    if (vae_fork) (void)hipEventDestroy(vae_fork);
    if (vae_join) (void)hipEventDestroy(vae_join);
  }

  size_t workspace_bytes() const override { return arena.cap; }

  // ------------------------------------------------------------ launch plumbing
  template <class F> int run(const char* tag, double flops, double bytes, F&& f) {
    if (arena.dry) return 0;
    if (!prof.enabled) return f();
    ProfRecord r;
    r.tag = tag;
    r.flops = flops;
    r.bytes = bytes;
    r.e0 = prof.get_event();
    r.e1 = prof.get_event();
    roctx_push(tag);
    GCV_CHECK_HIP(hipEventRecord(r.e0, cur));
    const int rc = f();
    GCV_CHECK_HIP(hipEventRecord(r.e1, cur));
    roctx_pop();
    prof.recs.push_back(r);
    return rc;
  }

  int gemm(const char* tag, const GemmArgs& g, int a_mode, int epi) {
    const double flops = 2.0 * g.M * (double)g.N * g.K;
    double a_bytes = (a_mode == A_PLAIN) ? (double)g.M * g.K : (double)g.M * (1 << g.cin_log2);
    if (a_mode == A_IM2COL3_S2) a_bytes *= 4.0;     // input has 4x the pixels of the output
    double c_bytes = (double)g.M * g.N;
    if (epi == EPI_POOL4) c_bytes *= 0.25;
    if (epi == EPI_RESID) c_bytes *= 2.0;
    double bytes = sizeof(T) * (a_bytes + (double)g.N * g.K + c_bytes);
    if (epi == EPI_SPLITK) bytes = sizeof(T) * (a_bytes + (double)g.N * g.K) + 4.0 * g.splitk * (double)g.M * g.N;
    return run(tag, flops, bytes, [&] { return launch_gemm<T>(g, a_mode, epi, cur); });
  }

  // ------------------------------------------------------------ weight fetch helpers
  static int fetch(const TensorMap& w, const std::string& name, int64_t numel, std::vector<float>& out) {
    auto it = w.find(name);
    if (it == w.end()) { set_error("missing weight tensor '" + name + "'"); return -4; }
    if (it->second.numel != numel) {
      set_error("weight tensor '" + name + "' has " + std::to_string(it->second.numel) + " elements, expected " +
                std::to_string(numel));
      return -4;
    }
    out.resize((size_t)numel);
    if (it->second.on_device) GCV_CHECK_HIP(hipMemcpy(out.data(), it->second.data, numel * 4, hipMemcpyDeviceToHost));
    else std::memcpy(out.data(), it->second.data, (size_t)numel * 4);
    return 0;
  }
  static std::vector<T> cast_vec(const std::vector<float>& v) {
    std::vector<T> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = host_cvt<T>(v[i]);
    return o;
  }

  int up_f32(const TensorMap& w, const std::string& name, int64_t n, WeightStore& st, float*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, n, v));
    GCV_UP(dst, st, v);
    return 0;
  }
  int up_cast(const TensorMap& w, const std::string& name, int64_t n, WeightStore& st, T*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, n, v));
    std::vector<T> c = cast_vec(v);
    GCV_UP(dst, st, c);
    return 0;
  }
  // Conv2d weight (Cout,Cin,kh,kw) [* per-Cout scale] -> (Cout, (ky,kx,ci)) in T
  int up_conv_gemm(const TensorMap& w, const std::string& name, int cout, int cin, int kh, int kw,
                   const std::vector<float>* scale, WeightStore& st, T*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, (int64_t)cout * cin * kh * kw, v));
    std::vector<T> o((size_t)cout * cin * kh * kw);
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int ky = 0; ky < kh; ++ky)
          for (int kx = 0; kx < kw; ++kx) {
            float x = v[(((size_t)co * cin + ci) * kh + ky) * kw + kx];
            if (scale) x *= (*scale)[co];
            o[(size_t)co * cin * kh * kw + ((size_t)(ky * kw + kx)) * cin + ci] = host_cvt<T>(x);
          }
    GCV_UP(dst, st, o);
    return 0;
  }
  // ConvTranspose2d weight (Cin,Cout,2,2) -> ((dy,dx,co), ci) in T
  int up_convt_gemm(const TensorMap& w, const std::string& name, int cin, int cout, WeightStore& st, T*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, (int64_t)cin * cout * 4, v));
    std::vector<T> o((size_t)cin * cout * 4);
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co)
        for (int dy = 0; dy < 2; ++dy)
          for (int dx = 0; dx < 2; ++dx)
            o[((size_t)(dy * 2 + dx) * cout + co) * cin + ci] = host_cvt<T>(v[(((size_t)ci * cout + co) * 2 + dy) * 2 + dx]);
    GCV_UP(dst, st, o);
    return 0;
  }
  // ConvTranspose2d weight (16,3,2,2) -> [ci][dy][dx][co] fp32
  int up_convt_small(const TensorMap& w, const std::string& name, WeightStore& st, float*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, 16 * 3 * 4, v));
    std::vector<float> o(16 * 12);
    for (int ci = 0; ci < 16; ++ci)
      for (int co = 0; co < 3; ++co)
        for (int dy = 0; dy < 2; ++dy)
          for (int dx = 0; dx < 2; ++dx) o[ci * 12 + (dy * 2 + dx) * 3 + co] = v[((ci * 3 + co) * 2 + dy) * 2 + dx];
    GCV_UP(dst, st, o);
    return 0;
  }
  // first conv (16,3,3,3) [* scale] -> [27][16] fp32, k = (ky*3+kx)*3 + ci
  int up_conv_first(const TensorMap& w, const std::string& name, const std::vector<float>* scale, WeightStore& st,
                    float*& dst) {
    std::vector<float> v;
    GCV_TRY(fetch(w, name, 16 * 27, v));
    std::vector<float> o(27 * 16);
    for (int co = 0; co < 16; ++co)
      for (int ci = 0; ci < 3; ++ci)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx)
            o[((ky * 3 + kx) * 3 + ci) * 16 + co] = v[((co * 3 + ci) * 3 + ky) * 3 + kx] * (scale ? (*scale)[co] : 1.0f);
    GCV_UP(dst, st, o);
    return 0;
  }

  int pack_convnext(const TensorMap& w, const std::string& p, WeightStore& st, CnxW<T>& o) {
    {
      std::vector<float> v, t(48 * 96);
      GCV_TRY(fetch(w, p + "stem.0.weight", 96 * 48, v));
      for (int co = 0; co < 96; ++co)
        for (int k = 0; k < 48; ++k) t[k * 96 + co] = v[co * 48 + k];
      GCV_UP(o.stem_w, st, t);
    }
    GCV_TRY(up_f32(w, p + "stem.0.bias", 96, st, o.stem_b));
    GCV_TRY(up_f32(w, p + "stem.1.weight", 96, st, o.stem_lnw));
    GCV_TRY(up_f32(w, p + "stem.1.bias", 96, st, o.stem_lnb));
    int bi = 0;
    for (int i = 0; i < 4; ++i) {
      const int C = kDims[i];
      if (i > 0) {
        const int Cp = kDims[i - 1];
        const std::string d = p + "stages." + std::to_string(i) + ".downsample.";
        GCV_TRY(up_f32(w, d + "0.weight", Cp, st, o.down[i - 1].ln_w));
        GCV_TRY(up_f32(w, d + "0.bias", Cp, st, o.down[i - 1].ln_b));
        GCV_TRY(up_conv_gemm(w, d + "1.weight", C, Cp, 2, 2, nullptr, st, o.down[i - 1].w));
        GCV_TRY(up_f32(w, d + "1.bias", C, st, o.down[i - 1].b));
      }
      for (int j = 0; j < kDepths[i]; ++j, ++bi) {
        const std::string b = p + "stages." + std::to_string(i) + ".blocks." + std::to_string(j) + ".";
        CnxBlockW<T>& k = o.blk[bi];
        {
          std::vector<float> v, t((size_t)49 * C);
          GCV_TRY(fetch(w, b + "conv_dw.weight", (int64_t)C * 49, v));
          for (int c = 0; c < C; ++c)
            for (int q = 0; q < 49; ++q) t[(size_t)q * C + c] = v[(size_t)c * 49 + q];
          GCV_UP(k.dw_w, st, t);
        }
        GCV_TRY(up_f32(w, b + "conv_dw.bias", C, st, k.dw_b));
        GCV_TRY(up_f32(w, b + "norm.weight", C, st, k.ln_w));
        GCV_TRY(up_f32(w, b + "norm.bias", C, st, k.ln_b));
        GCV_TRY(up_cast(w, b + "mlp.fc1.weight", (int64_t)4 * C * C, st, k.fc1_w));
        GCV_TRY(up_f32(w, b + "mlp.fc1.bias", 4 * C, st, k.fc1_b));
        GCV_TRY(up_cast(w, b + "mlp.fc2.weight", (int64_t)4 * C * C, st, k.fc2_w));
        k.fc2_wc = nullptr;
        k.fc1_wf = k.fc2_wf = nullptr;
        k.fc_wp = nullptr;
        if constexpr (sizeof(T) == 2) {
          if (xs_mlp_default(C) && use_xs_mlp) {
            k.fc_wp = (T*)st.raw(xs_mlp_packed_elems(C) * sizeof(T));
            if (!k.fc_wp) { set_error("hipMalloc failed for the packed fc1 | fc2 records"); return -5; }
            GCV_TRY((launch_pack_xs_mlp<T, T>(k.fc1_w, k.fc2_w, k.fc_wp, C, nullptr)));
            GCV_CHECK_HIP(hipDeviceSynchronize());
          }
          if (mlp_pair_supported(C) && use_mlp_pair) {
            k.fc1_wf = (T*)st.raw((size_t)4 * C * C * sizeof(T));
            k.fc2_wf = (T*)st.raw((size_t)4 * C * C * sizeof(T));
            if (!k.fc1_wf || !k.fc2_wf) { set_error("hipMalloc failed for the fragment-major fc1 / fc2"); return -5; }
            GCV_TRY((launch_pack_w1_frag<T, T>(k.fc1_w, k.fc1_wf, C, nullptr)));
            {
              // the block's layer scale is folded into the packed fc2 (mlp_pair.h): packed from the fp32 source so that
              // gamma * W2 is rounded to T once
              std::vector<float> v, gv;
              GCV_TRY(fetch(w, b + "mlp.fc2.weight", (int64_t)4 * C * C, v));
              GCV_TRY(fetch(w, b + "gamma", C, gv));
              struct DevBuf {                      // freed on every exit path
                float* p = nullptr;
                ~DevBuf() { if (p) (void)hipFree(p); }
              } tmp, tg;
              GCV_CHECK_HIP(hipMalloc((void**)&tmp.p, v.size() * 4));
              GCV_CHECK_HIP(hipMalloc((void**)&tg.p, gv.size() * 4));
              GCV_CHECK_HIP(hipMemcpy(tmp.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
              GCV_CHECK_HIP(hipMemcpy(tg.p, gv.data(), gv.size() * 4, hipMemcpyHostToDevice));
              GCV_TRY((launch_pack_w2_frag<T, float>(tmp.p, tg.p, k.fc2_wf, C, nullptr)));
              GCV_CHECK_HIP(hipDeviceSynchronize());
            }
          }
          if ((C <= 192 && !k.fc_wp) || (C == 384 && use_fused_mlp384)) {   // stages with a round-2 fused MLP kernel in use
            std::vector<float> v;
            GCV_TRY(fetch(w, b + "mlp.fc2.weight", (int64_t)4 * C * C, v));
            struct DevBuf {                        // freed on every exit path
              float* p = nullptr;
              ~DevBuf() { if (p) (void)hipFree(p); }
            } tmp;
            GCV_CHECK_HIP(hipMalloc((void**)&tmp.p, v.size() * 4));
            GCV_CHECK_HIP(hipMemcpy(tmp.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
            k.fc2_wc = (T*)st.raw(v.size() * sizeof(T));
            if (!k.fc2_wc) { set_error("hipMalloc failed for packed fc2"); return -5; }
            GCV_TRY(launch_pack_w2_chunks<T>(tmp.p, k.fc2_wc, C, nullptr));
            GCV_CHECK_HIP(hipDeviceSynchronize());
          }
        }
        GCV_TRY(up_f32(w, b + "mlp.fc2.bias", C, st, k.fc2_b));
        GCV_TRY(up_f32(w, b + "gamma", C, st, k.gamma));
      }
    }
    GCV_TRY(up_f32(w, p + "head.norm.weight", 768, st, o.head_lnw));
    GCV_TRY(up_f32(w, p + "head.norm.bias", 768, st, o.head_lnb));
    GCV_TRY(up_cast(w, p + "head.fc.weight", 1000 * 768, st, o.head_fc_w));
    GCV_TRY(up_f32(w, p + "head.fc.bias", 1000, st, o.head_fc_b));
    return 0;
  }

  int pack_head(const TensorMap& w, WeightStore& st, HeadW<T>& h) {
    GCV_TRY(up_cast(w, "fc.weight", 500 * 2000, st, h.fc_w));
    GCV_TRY(up_f32(w, "fc.bias", 500, st, h.fc_b));
    GCV_TRY(up_f32(w, "fc2.weight", 2 * 500, st, h.fc2_w));
    GCV_TRY(up_f32(w, "fc2.bias", 2, st, h.fc2_b));
    return 0;
  }

  int load_ed(const TensorMap& w) override {
    GCV_CHECK_HIP(hipSetDevice(device));
    has_ed = false;
    ws_ed.clear();
    GCV_TRY(up_conv_first(w, "encoder.features.0.weight", nullptr, ws_ed, ed.enc1_w));
    GCV_TRY(up_f32(w, "encoder.features.0.bias", 16, ws_ed, ed.enc1_b));
    const int ech[5] = {16, 32, 64, 128, 256};
    const int eidx[4] = {3, 6, 9, 12};
    for (int l = 0; l < 4; ++l) {
      const std::string n = "encoder.features." + std::to_string(eidx[l]);
      GCV_TRY(up_conv_gemm(w, n + ".weight", ech[l + 1], ech[l], 3, 3, nullptr, ws_ed, ed.enc_w[l]));
      GCV_TRY(up_f32(w, n + ".bias", ech[l + 1], ws_ed, ed.enc_b[l]));
    }
    const int dch[5] = {256, 128, 64, 32, 16};
    const int didx[4] = {0, 2, 4, 6};
    for (int l = 0; l < 4; ++l) {
      const std::string n = "decoder.features." + std::to_string(didx[l]);
      GCV_TRY(up_convt_gemm(w, n + ".weight", dch[l], dch[l + 1], ws_ed, ed.dec_w[l]));
      GCV_TRY(up_f32(w, n + ".bias", dch[l + 1], ws_ed, ed.dec_b[l]));
    }
    GCV_TRY(up_convt_small(w, "decoder.features.8.weight", ws_ed, ed.dec5_w));
    GCV_TRY(up_f32(w, "decoder.features.8.bias", 3, ws_ed, ed.dec5_b));
    GCV_TRY(pack_convnext(w, "backbone.", ws_ed, bb_ed));
    GCV_TRY(pack_head(w, ws_ed, ed.head));
    has_ed = true;
    return 0;
  }

  int pack_mu(const TensorMap& w, const std::string& name, T*& dst) {
    auto it = w.find(name);
    const int64_t total = (int64_t)12544 * 25088;
    if (it == w.end() || it->second.numel != total) { set_error("missing or mis-sized '" + name + "'"); return -4; }
    dst = (T*)ws_vae.raw((size_t)total * sizeof(T));
    if (!dst) { set_error("hipMalloc failed for " + name); return -5; }
    const float* src = it->second.data;
    if (it->second.on_device) {
      hipLaunchKernelGGL((pack_mu_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, 0, src, dst, total);
      GCV_CHECK_HIP(hipGetLastError());
      GCV_CHECK_HIP(hipDeviceSynchronize());
      return 0;
    }
    // host source (the published 2.6 GB VAE checkpoint, model/genconvit.py:16-21): streamed through a 49 MB staging
    // buffer, 512 rows at a time (the permutation stays inside a row), so the device never holds an fp32 copy
    constexpr int64_t kRows = 512;
    struct DevBuf {
      float* p = nullptr;
      ~DevBuf() { if (p) (void)hipFree(p); }
    } tmp;
    GCV_CHECK_HIP(hipMalloc((void**)&tmp.p, (size_t)(kRows * 25088 * 4)));
    for (int64_t r0 = 0; r0 < 12544; r0 += kRows) {
      const int64_t rows = std::min<int64_t>(kRows, 12544 - r0), n = rows * 25088;
      GCV_CHECK_HIP(hipMemcpy(tmp.p, src + r0 * 25088, (size_t)n * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((pack_mu_kernel<T>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, 0, tmp.p, dst + r0 * 25088, n);
      GCV_CHECK_HIP(hipGetLastError());
      GCV_CHECK_HIP(hipDeviceSynchronize());
    }
    return 0;
  }

  int load_vae(const TensorMap& w) override {
    GCV_CHECK_HIP(hipSetDevice(device));
    has_vae = false;
    ws_vae.clear();
    const int ech[5] = {3, 16, 32, 64, 128};
    const int eidx[4] = {0, 3, 6, 9};
    for (int l = 0; l < 4; ++l) {
      const int c = ech[l + 1];
      const std::string cn = "encoder.features." + std::to_string(eidx[l]);
      const std::string bn = "encoder.features." + std::to_string(eidx[l] + 1);
      std::vector<float> cb, g, be, rm, rv;
      GCV_TRY(fetch(w, cn + ".bias", c, cb));
      GCV_TRY(fetch(w, bn + ".weight", c, g));
      GCV_TRY(fetch(w, bn + ".bias", c, be));
      GCV_TRY(fetch(w, bn + ".running_mean", c, rm));
      GCV_TRY(fetch(w, bn + ".running_var", c, rv));
      // eval-mode BatchNorm2d (eps 1e-5) folded into the conv: y = conv(x)*s + (b - mean)*s + beta
      std::vector<float> scale(c), bias(c);
      for (int i = 0; i < c; ++i) {
        scale[i] = g[i] / std::sqrt(rv[i] + 1e-5f);
        bias[i] = (cb[i] - rm[i]) * scale[i] + be[i];
      }
      if (l == 0) {
        GCV_TRY(up_conv_first(w, cn + ".weight", &scale, ws_vae, vae.enc1_w));
        GCV_UP(vae.enc1_b, ws_vae, bias);
      } else {
        GCV_TRY(up_conv_gemm(w, cn + ".weight", c, ech[l], 3, 3, &scale, ws_vae, vae.enc_w[l - 1]));
        GCV_UP(vae.enc_b[l - 1], ws_vae, bias);
      }
    }
    GCV_TRY(pack_mu(w, "encoder.mu.weight", vae.mu_w));
    GCV_TRY(up_f32(w, "encoder.mu.bias", 12544, ws_vae, vae.mu_b));
    vae.var_w = nullptr;
    vae.var_b = nullptr;
    if (w.count("encoder.var.weight")) {     // only needed for the optional KL output
      GCV_TRY(pack_mu(w, "encoder.var.weight", vae.var_w));
      GCV_TRY(up_f32(w, "encoder.var.bias", 12544, ws_vae, vae.var_b));
    }
    const int dch[4] = {256, 64, 32, 16};
    const int didx[3] = {0, 2, 4};
    for (int l = 0; l < 3; ++l) {
      const std::string n = "decoder.features." + std::to_string(didx[l]);
      GCV_TRY(up_convt_gemm(w, n + ".weight", dch[l], dch[l + 1], ws_vae, vae.dec_w[l]));
      GCV_TRY(up_f32(w, n + ".bias", dch[l + 1], ws_vae, vae.dec_b[l]));
    }
    GCV_TRY(up_convt_small(w, "decoder.features.6.weight", ws_vae, vae.dec4_w));
    GCV_TRY(up_f32(w, "decoder.features.6.bias", 3, ws_vae, vae.dec4_b));
    GCV_TRY(pack_convnext(w, "convnext_backbone.", ws_vae, bb_vae));
    GCV_TRY(pack_head(w, ws_vae, vae.head));
    has_vae = true;
    return 0;
  }

  int load_swin(const TensorMap& w, const std::string& prefix) override {
    GCV_CHECK_HIP(hipSetDevice(device));
    has_swin = false;
    ws_swin.clear();
    GCV_TRY((pack_swin<T>(*this, w, prefix, swin)));
    has_swin = true;
    return 0;
  }

  // ------------------------------------------------------------ ConvNeXt-T over token segments
  // keep = true leaves the token buffers allocated (the caller releases its own mark): two passes of one forward that run
  // on different streams must not share them
  int run_convnext(const CnxW<T>& w, const Seg<T>* segs, int nseg, bool keep = false) {
    int h[4], wd[4];
    int64_t m[4], moff[4], M = 0;
    int ntot = 0;
    GCV_REQUIRE(nseg >= 1 && nseg <= 4, "1..4 segments");
    for (int s = 0; s < nseg; ++s) {
      GCV_REQUIRE(segs[s].H % 4 == 0 && segs[s].W % 4 == 0 && segs[s].n > 0, "segment geometry");
      h[s] = segs[s].H / 4;
      wd[s] = segs[s].W / 4;
      m[s] = (int64_t)segs[s].n * h[s] * wd[s];
      moff[s] = M;
      M += m[s];
      ntot += segs[s].n;
    }
    GCV_REQUIRE(M * 384 < (int64_t)1 << 31, "token count too large for 32-bit GEMM indexing");
    const size_t mk = arena.mark();
    T* X = arena.get<T>(M * 96);
    T* Y = arena.get<T>(M * 96);
    T* Hd = arena.get<T>(M * 384);
    T* Pool = arena.get<T>((int64_t)ntot * 768);
    if (!arena.dry && arena.overflow) { set_error("workspace arena too small: batch exceeds max_batch"); return -6; }

    for (int s = 0; s < nseg; ++s) {
      const Seg<T>& g = segs[s];
      GCV_TRY(run("cnx.stem_ln", 2.0 * m[s] * 96 * 48, sizeof(T) * (double)m[s] * (48 + 96), [&] {
        return launch_stem_ln<T>(g.x, g.sb, g.sc, g.sy, g.sx, w.stem_w, w.stem_b, w.stem_lnw, w.stem_lnb,
                                 X + moff[s] * 96, g.n, h[s], wd[s], 1e-6f, cur);
      }));
    }
    int bi = 0;
    bool lnp_fused = false;       // the previous stage's last MLP has already written the LayerNorm'ed patches (into Hd)
    for (int i = 0; i < 4; ++i) {
      const int C = kDims[i];
      if (i > 0) {
        const int Cp = kDims[i - 1];
        int64_t newM = 0, noff[4];
        for (int s = 0; s < nseg;) {
          // neighbouring segments of one geometry (ED: reconstruction + original pass) are contiguous on both sides:
          // one launch over all their images, as for the depthwise kernel below
          int e = s, nimg = 0;
          int64_t mm = 0;
          const int64_t noff0 = newM;
          while (e < nseg && h[e] == h[s] && wd[e] == wd[s]) {
            noff[e] = newM;
            GCV_REQUIRE(h[e] / 2 > 0 && wd[e] / 2 > 0, "image too small for ConvNeXt downsampling");
            newM += (int64_t)segs[e].n * (h[e] / 2) * (wd[e] / 2);
            nimg += segs[e].n; mm += m[e]; ++e;
          }
          if (!lnp_fused)
            GCV_TRY(run("cnx.ln_patchify", 8.0 * mm * Cp, 2.0 * sizeof(T) * (double)mm * Cp, [&] {
              return launch_ln_patchify<T>(X + moff[s] * Cp, w.down[i - 1].ln_w, w.down[i - 1].ln_b,
                                           Y + noff0 * 4 * Cp, nimg, h[s], wd[s], Cp, 1e-6f, cur);
            }));
          s = e;
        }
        for (int s = 0; s < nseg; ++s) {
          h[s] /= 2;
          wd[s] /= 2;
          m[s] = (int64_t)segs[s].n * h[s] * wd[s];
          moff[s] = noff[s];
        }
        M = newM;
        GemmArgs g{};
        g.A = lnp_fused ? Hd : Y; g.lda = 4 * Cp; g.Wt = w.down[i - 1].w; g.C = X; g.ldc = C; g.bias = w.down[i - 1].b;
        lnp_fused = false;
        g.M = (int)M; g.N = C; g.K = 4 * Cp; g.act = ACT_NONE; g.splitk = 1;
        GCV_TRY(gemm("cnx.down_gemm", g, A_PLAIN, EPI_BIAS_ACT));
      }
      for (int j = 0; j < kDepths[i]; ++j, ++bi) {
        const CnxBlockW<T>& k = w.blk[bi];
        for (int s = 0; s < nseg;) {
          // neighbouring segments of one geometry (ED: reconstruction + original pass) are contiguous in the token
          // buffer: one launch over all their images (256 images fill the 256 CUs with whole-image row bands)
          int e = s + 1, nimg = segs[s].n;
          int64_t mm = m[s];
          while (e < nseg && h[e] == h[s] && wd[e] == wd[s]) { nimg += segs[e].n; mm += m[e]; ++e; }
          GCV_TRY(run("cnx.dwconv7_ln", 2.0 * 49 * mm * C, 2.0 * sizeof(T) * (double)mm * C + 49.0 * C * 4, [&] {
            return launch_dwconv7_ln<T>(X + moff[s] * C, k.dw_w, k.dw_b, k.ln_w, k.ln_b, Y + moff[s] * C, nimg, h[s],
                                        wd[s], C, 1e-6f, cur);
          }));
          s = e;
        }
        if constexpr (sizeof(T) == 2) {
          if (k.fc_wp) {
            XsMlpArgs xa{Y, k.fc_wp, k.fc1_b, k.fc2_b, k.gamma, X, X, (int)M};
            if (j == kDepths[i] - 1 && i < 3 && use_lnp_epilogue) {   // stage boundary in the epilogue (see the C = 96 case below)
              bool even = true;
              int64_t o0 = 0;
              for (int s = 0; s < nseg; ++s) {
                xa.lnp_tok0[s] = (int)moff[s]; xa.lnp_hw[s] = h[s] * wd[s]; xa.lnp_wd[s] = wd[s]; xa.lnp_out0[s] = (int)o0;
                o0 += (int64_t)segs[s].n * (h[s] / 2) * (wd[s] / 2);
                even = even && h[s] % 2 == 0 && wd[s] % 2 == 0;
              }
              if (even) {
                xa.out = Hd; xa.lnp_w = w.down[i].ln_w; xa.lnp_b = w.down[i].ln_b; xa.lnp_eps = 1e-6f; xa.lnp_nseg = nseg;
                lnp_fused = true;
              }
            }
            GCV_TRY(run("cnx.fused_mlp", 16.0 * M * C * (double)C, 3.0 * sizeof(T) * (double)M * C + 16.0 * C * C,
                        [&] { return launch_xs_mlp<T>(xa, C, cur); }));
            continue;
          }
          if (k.fc2_wc && use_fused_mlp && (C < 384 || use_fused_mlp384)) {
            MlpArgs ma{Y, k.fc1_w, k.fc1_b, k.fc2_wc, k.fc2_b, k.gamma, X, X, (int)M};
            // last block of the stage on the LDS-resident kernel: its epilogue applies the stage boundary's LayerNorm2d +
            // space-to-depth and writes the down-sampling GEMM's operand (into Hd: Y is still being read as x_ln); the
            // residual stream ends here and is not written
            if (j == kDepths[i] - 1 && i < 3 && use_lnp_epilogue && fused_mlp_res_applies(C, M)) {
              ma.out = Hd;
              ma.lnp_w = w.down[i].ln_w; ma.lnp_b = w.down[i].ln_b; ma.lnp_eps = 1e-6f; ma.lnp_nseg = nseg;
              int64_t o0 = 0;
              bool even = true;
              for (int s = 0; s < nseg; ++s) {
                ma.lnp_tok0[s] = (int)moff[s]; ma.lnp_hw[s] = h[s] * wd[s]; ma.lnp_wd[s] = wd[s]; ma.lnp_out0[s] = (int)o0;
                o0 += (int64_t)segs[s].n * (h[s] / 2) * (wd[s] / 2);
                even = even && h[s] % 2 == 0 && wd[s] % 2 == 0;
              }
              if (even) lnp_fused = true;
              else { ma.out = X; ma.lnp_nseg = 0; }     // odd maps drop their last row / column: the separate kernel does that
            }
            GCV_TRY(run("cnx.fused_mlp", 16.0 * M * C * (double)C, 3.0 * sizeof(T) * (double)M * C + 16.0 * C * C,
                        [&] { return launch_fused_mlp<T>(ma, C, cur); }));
            continue;
          }
        }
        if constexpr (sizeof(T) == 2) {
          if (k.fc1_wf && k.fc2_wf) {
            MlpPairArgs pa{Y, k.fc1_wf, k.fc1_b, k.fc2_wf, k.fc2_b, k.gamma, X, X, Hd, (int)M};
            GCV_TRY(run("cnx.pw1_gelu", 8.0 * M * C * (double)C, sizeof(T) * (5.0 * M * C + 4.0 * C * C),
                        [&] { return launch_xs_pw1<T>(pa, C, cur); }));
            GCV_TRY(run("cnx.pw2_scale_res", 8.0 * M * C * (double)C, sizeof(T) * (6.0 * M * C + 4.0 * C * C),
                        [&] { return launch_pw2f<T>(pa, C, cur); }));
            continue;
          }
        }
        GemmArgs g1{};
        g1.A = Y; g1.lda = C; g1.Wt = k.fc1_w; g1.C = Hd; g1.ldc = 4 * C; g1.bias = k.fc1_b;
        g1.M = (int)M; g1.N = 4 * C; g1.K = C; g1.act = ACT_GELU; g1.splitk = 1;
        GCV_TRY(gemm("cnx.pw1_gelu", g1, A_PLAIN, EPI_BIAS_ACT));
        GemmArgs g2{};
        g2.A = Hd; g2.lda = 4 * C; g2.Wt = k.fc2_w; g2.C = X; g2.ldc = C; g2.bias = k.fc2_b; g2.gamma = k.gamma;
        g2.resid = X; g2.M = (int)M; g2.N = C; g2.K = 4 * C; g2.act = ACT_NONE; g2.splitk = 1;
        GCV_TRY(gemm("cnx.pw2_scale_res", g2, A_PLAIN, EPI_RESID));
      }
    }
    // pooling + LayerNorm: one launch over neighbouring segments of one map size (their tokens are contiguous).  When the
    // passes are the column blocks of one feature matrix (ED, VAE: equal frame counts, out = base + 1000 s, one leading
    // dimension, one activation) the pooled rows are written frame-major — row nseg * frame + pass — so that ONE classifier
    // GEMM over all nseg * n rows with ldc = 1000 writes the (n, nseg * 1000) matrix (round 4; one launch per pass before)
    bool one_fc = nseg > 1;
    for (int s = 1; s < nseg && one_fc; ++s)
      one_fc = segs[s].n == segs[0].n && segs[s].out == segs[0].out + 1000 * s && segs[s].out_ld == 1000 * nseg &&
               segs[0].out_ld == 1000 * nseg && segs[s].act == segs[0].act;
    int no = 0;
    for (int s = 0; s < nseg; ++s) {
      if (s == 0 || h[s] * wd[s] != h[s - 1] * wd[s - 1]) {
        int e = s, nimg = 0;
        int64_t mm = 0;
        while (e < nseg && h[e] * wd[e] == h[s] * wd[s]) { nimg += segs[e].n; mm += m[e]; ++e; }
        GCV_TRY(run("cnx.pool_ln", 2.0 * mm * 768, sizeof(T) * (double)mm * 768, [&] {
          if (one_fc)
            return launch_pool_ln<T>(X + moff[s] * 768, w.head_lnw, w.head_lnb, Pool, nimg, h[s] * wd[s], 768, 1e-6f, cur,
                                     segs[0].n, nseg, s);
          return launch_pool_ln<T>(X + moff[s] * 768, w.head_lnw, w.head_lnb, Pool + (int64_t)no * 768, nimg, h[s] * wd[s],
                                   768, 1e-6f, cur);
        }));
      }
      if (!one_fc) {
        GemmArgs g{};
        g.A = Pool + (int64_t)no * 768; g.lda = 768; g.Wt = w.head_fc_w; g.C = segs[s].out; g.ldc = segs[s].out_ld;
        g.bias = w.head_fc_b; g.M = segs[s].n; g.N = 1000; g.K = 768; g.act = segs[s].act; g.splitk = 1;
        GCV_TRY(gemm("cnx.head_fc", g, A_PLAIN, EPI_BIAS_ACT));
      }
      no += segs[s].n;
    }
    if (one_fc) {
      GemmArgs g{};
      g.A = Pool; g.lda = 768; g.Wt = w.head_fc_w; g.C = segs[0].out; g.ldc = 1000;
      g.bias = w.head_fc_b; g.M = ntot; g.N = 1000; g.K = 768; g.act = segs[0].act; g.splitk = 1;
      GCV_TRY(gemm("cnx.head_fc", g, A_PLAIN, EPI_BIAS_ACT));
    }
    if (!keep) arena.release(mk);
    return 0;
  }

  int run_head(const HeadW<T>& hw, const T* feat, int B, int act, float* logits) {
    // fc (2000 -> 500) eight ways split-K into fp32 partials; their sum, the bias, the activation and fc2 (500 -> 2) are one
    // small kernel (round 4: as one GEMM the layer was a chain of 32 K tiles on eight workgroups, 37 us at 128 rows)
    const int SK = 8, KPS = 256;
    const size_t mk = arena.mark();
    float* part = arena.get<float>((int64_t)SK * B * 500 + 8);
    GemmArgs g{};
    g.A = feat; g.lda = 2000; g.Wt = hw.fc_w; g.partial = part;
    g.M = B; g.N = 500; g.K = 2000; g.act = ACT_NONE; g.splitk = SK; g.k_per_split = KPS;
    GCV_TRY(gemm("head.fc", g, A_PLAIN, EPI_SPLITK));
    GCV_TRY(run("head.fc2", 2.0 * B * 2 * 500, 4.0 * (double)SK * B * 500, [&] {
      return launch_head_tail_splitk<T>(part, SK, hw.fc_b, act, hw.fc2_w, hw.fc2_b, logits, B, 500, cur);
    }));
    arena.release(mk);
    return 0;
  }

  int check_batch(int B) {
    GCV_REQUIRE(B >= 1, "batch must be >= 1");
    GCV_REQUIRE(B <= max_batch, "batch exceeds the max_batch this handle was created with");
    return 0;
  }

  // ------------------------------------------------------------ ED (model/genconvit_ed.py:77-88)
  int ed_forward(const void* xv, int B, float* logits, hipStream_t s) override {
    if (!arena.dry) {
      GCV_REQUIRE(has_ed, "ED weights not loaded (gcv_load_ed)");
      GCV_TRY(check_batch(B));
      GCV_REQUIRE(xv && logits, "null input/output");
    }
    cur = s;
    const T* x = (const T*)xv;
    const size_t mk = arena.mark();
    T* e1 = arena.get<T>((int64_t)B * 112 * 112 * 16);
    T* e2 = arena.get<T>((int64_t)B * 56 * 56 * 32);
    T* e3 = arena.get<T>((int64_t)B * 28 * 28 * 64);
    T* e4 = arena.get<T>((int64_t)B * 14 * 14 * 128);
    T* e5 = arena.get<T>((int64_t)B * 7 * 7 * 256);
    T* d1 = arena.get<T>((int64_t)B * 14 * 14 * 128);
    T* d2 = arena.get<T>((int64_t)B * 28 * 28 * 64);
    T* d3 = arena.get<T>((int64_t)B * 56 * 56 * 32);
    T* d4 = arena.get<T>((int64_t)B * 112 * 112 * 16);
    T* rec = arena.get<T>((int64_t)B * 224 * 224 * 3);
    T* feat = arena.get<T>((int64_t)B * 2000);
    if (!arena.dry && arena.overflow) { set_error("workspace arena too small"); return -6; }

    GCV_TRY(run("ed.enc1_conv3_relu_pool", 2.0 * B * 224 * 224 * 16 * 27,
                sizeof(T) * (double)B * (3 * 224 * 224 + 16 * 112 * 112), [&] {
      return launch_conv3_first<T>(x, 3 * 224 * 224, 224 * 224, 224, 1, ed.enc1_w, ed.enc1_b, e1, B, 224, 224, true,
                                   ACT_RELU, cur);
    }));
    {
      const T* in[4] = {e1, e2, e3, e4};
      T* out[4] = {e2, e3, e4, e5};
      const int Hs[4] = {112, 56, 28, 14};
      const int cl[4] = {4, 5, 6, 7};
      for (int l = 0; l < 4; ++l) {
        GemmArgs g{};
        g.A = in[l]; g.Wt = ed.enc_w[l]; g.C = out[l]; g.bias = ed.enc_b[l];
        g.M = B * Hs[l] * Hs[l]; g.N = 2 << cl[l]; g.K = 9 << cl[l]; g.ldc = g.N; g.act = ACT_RELU; g.splitk = 1;
        g.H = Hs[l]; g.W = Hs[l]; g.cin_log2 = cl[l];
        GCV_TRY(gemm("ed.enc_conv3_relu_pool", g, A_IM2COL3_POOL, EPI_POOL4));
      }
    }
    {
      const T* in[4] = {e5, d1, d2, d3};
      T* out[4] = {d1, d2, d3, d4};
      const int Hs[4] = {7, 14, 28, 56};
      const int col[4] = {7, 6, 5, 4};   // log2(Cout)
      for (int l = 0; l < 4; ++l) {
        GemmArgs g{};
        g.A = in[l]; g.lda = 2 << col[l]; g.Wt = ed.dec_w[l]; g.C = out[l]; g.bias = ed.dec_b[l];
        g.M = B * Hs[l] * Hs[l]; g.N = 4 << col[l]; g.K = 2 << col[l]; g.act = ACT_RELU; g.splitk = 1;
        g.H = Hs[l]; g.W = Hs[l]; g.cout_log2 = col[l];
        GCV_TRY(gemm("ed.dec_convT_relu", g, A_PLAIN, EPI_CONVT));
      }
    }
    GCV_TRY(run("ed.dec5_convT_relu", 2.0 * B * 112 * 112 * 16 * 12,
                sizeof(T) * (double)B * (16 * 112 * 112 + 3 * 224 * 224),
                [&] { return launch_convt2_small<T>(d4, ed.dec5_w, ed.dec5_b, rec, B, 112, 112, ACT_RELU, cur); }));
    // both backbone passes share weights and shape -> one 2B-image token stream (running backbone(orig) on a side stream
    // under the encoder / decoder chain instead, as the VAE does, measured 8.69 vs 8.25 ms per step: the merged launches
    // are worth more than the overlap).  cat order (genconvit_ed.py:85): [backbone(recon), backbone(orig)], GELU (:75)
    Seg<T> segs[2];
    segs[0] = Seg<T>{rec, (int64_t)224 * 224 * 3, 1, 224 * 3, 3, B, 224, 224, feat, 2000, ACT_GELU};
    segs[1] = Seg<T>{x, (int64_t)3 * 224 * 224, 224 * 224, 224, 1, B, 224, 224, feat + 1000, 2000, ACT_GELU};
    GCV_TRY(run_convnext(bb_ed, segs, 2));
    GCV_TRY(run_head(ed.head, feat, B, ACT_GELU, logits));
    arena.release(mk);
    return 0;
  }

  // ------------------------------------------------------------ VAE (model/genconvit_vae.py:107-116)
  int vae_forward(const void* xv, const float* eps, int B, float* logits, void* recon224, float* mse, float* kl,
                  hipStream_t s) override {
    if (!arena.dry) {
      GCV_REQUIRE(has_vae, "VAE weights not loaded (gcv_load_vae)");
      GCV_TRY(check_batch(B));
      GCV_REQUIRE(xv && eps && logits, "null input/eps/output");
      GCV_REQUIRE(!kl || vae.var_w, "KL requested but encoder.var weights were not loaded");
    }
    cur = s;
    const T* x = (const T*)xv;
    // split-K plan of the 25088-deep mu / var GEMMs (392 K tiles of 64 = 8 * 49): 8 ways at 128-row tiles (98 x 8 = 784
    // workgroups), 7 ways for batches of 64 frames and fewer, whose 32- / 64-row tiles leave room for every one of the 686
    // workgroups at once (vae B = 32 bf16, same box: 0.143 -> 0.118 ms; at B = 128 seven ways measure 0.193 against 0.175)
    const int SPLITK = B <= 64 ? 7 : 8, KPS = 25088 / SPLITK;
    const size_t mk = arena.mark();
    T* v1 = arena.get<T>((int64_t)B * 112 * 112 * 16);
    T* v2 = arena.get<T>((int64_t)B * 56 * 56 * 32);
    T* v3 = arena.get<T>((int64_t)B * 28 * 28 * 64);
    T* v4 = arena.get<T>((int64_t)B * 14 * 14 * 128);
    float* part = arena.get<float>((int64_t)8 * B * 12544);
    float* mu = arena.get<float>((int64_t)B * 12544);
    float* rowsum = arena.get<float>(B + 8);
    T* z = arena.get<T>((int64_t)B * 12544);
    T* d1 = arena.get<T>((int64_t)B * 14 * 14 * 64);
    T* d2 = arena.get<T>((int64_t)B * 28 * 28 * 32);
    T* d3 = arena.get<T>((int64_t)B * 56 * 56 * 16);
    T* xhat = arena.get<T>((int64_t)B * 112 * 112 * 3);
    T* feat = arena.get<T>((int64_t)B * 2000);
    float* msepart = arena.get<float>((int64_t)B * 196);
    if (!arena.dry && arena.overflow) { set_error("workspace arena too small"); return -6; }

    // cat order (genconvit_vae.py:113): [backbone(x @224), backbone(x_hat @112)], activation ReLU (:104)
    Seg<T> segs[2];
    segs[0] = Seg<T>{x, (int64_t)3 * 224 * 224, 224 * 224, 224, 1, B, 224, 224, feat, 2000, ACT_RELU};
    segs[1] = Seg<T>{xhat, (int64_t)112 * 112 * 3, 1, 112 * 3, 3, B, 112, 112, feat + 1000, 2000, ACT_RELU};
    const bool split = (vae_split_env >= 0 ? vae_split_env != 0 : !in_ensemble) && !prof.enabled;   // (profiled steps stay on one stream: serial per-kernel times)
    Join join;
    if (split) GCV_TRY(side_pass(bb_vae, &segs[0], s, join));

    GCV_TRY(run("vae.enc1_conv3s2_bn_leaky", 2.0 * B * 112 * 112 * 16 * 27,
                sizeof(T) * (double)B * (3 * 224 * 224 + 16 * 112 * 112), [&] {
      return launch_conv3_first<T>(x, 3 * 224 * 224, 224 * 224, 224, 1, vae.enc1_w, vae.enc1_b, v1, B, 224, 224,
                                   false, ACT_LEAKY, cur);
    }));
    {
      const T* in[3] = {v1, v2, v3};
      T* out[3] = {v2, v3, v4};
      const int Hs[3] = {112, 56, 28};
      const int cl[3] = {4, 5, 6};
      for (int l = 0; l < 3; ++l) {
        GemmArgs g{};
        g.A = in[l]; g.Wt = vae.enc_w[l]; g.C = out[l]; g.bias = vae.enc_b[l];
        g.M = B * (Hs[l] / 2) * (Hs[l] / 2); g.N = 2 << cl[l]; g.K = 9 << cl[l]; g.ldc = g.N; g.act = ACT_LEAKY;
        g.splitk = 1; g.H = Hs[l]; g.W = Hs[l]; g.cin_log2 = cl[l];
        GCV_TRY(gemm("vae.enc_conv3s2_bn_leaky", g, A_IM2COL3_S2, EPI_BIAS_ACT));
      }
    }
    {
      GemmArgs g{};
      g.A = v4; g.lda = 25088; g.Wt = vae.mu_w; g.partial = part; g.M = B; g.N = 12544; g.K = 25088;
      g.splitk = SPLITK; g.k_per_split = KPS; g.act = ACT_NONE;
      GCV_TRY(gemm("vae.mu_gemm_splitk", g, A_PLAIN, EPI_SPLITK));
      GCV_TRY(run("vae.reparam", 4.0 * B * 12544, 4.0 * (SPLITK + 2) * (double)B * 12544, [&] {
        return launch_reparam<T>(part, SPLITK, vae.mu_b, eps, mu, z, B, 12544, cur);
      }));
      if (kl) {
        g.Wt = vae.var_w;
        GCV_TRY(gemm("vae.var_gemm_splitk", g, A_PLAIN, EPI_SPLITK));
        GCV_TRY(run("vae.kl", 6.0 * B * 12544, 4.0 * (SPLITK + 1) * (double)B * 12544,
                    [&] { return launch_kl(part, SPLITK, vae.var_b, mu, rowsum, kl, B, 12544, cur); }));
      }
    }
    {
      const T* in[3] = {z, d1, d2};
      T* out[3] = {d1, d2, d3};
      const int Hs[3] = {7, 14, 28};
      const int cin[3] = {256, 64, 32};
      const int col[3] = {6, 5, 4};
      for (int l = 0; l < 3; ++l) {
        GemmArgs g{};
        g.A = in[l]; g.lda = cin[l]; g.Wt = vae.dec_w[l]; g.C = out[l]; g.bias = vae.dec_b[l];
        g.M = B * Hs[l] * Hs[l]; g.N = 4 << col[l]; g.K = cin[l]; g.act = ACT_LEAKY; g.splitk = 1;
        g.H = Hs[l]; g.W = Hs[l]; g.cout_log2 = col[l];
        GCV_TRY(gemm("vae.dec_convT_leaky", g, A_PLAIN, EPI_CONVT));
      }
    }
    GCV_TRY(run("vae.dec4_convT_leaky", 2.0 * B * 56 * 56 * 16 * 12,
                sizeof(T) * (double)B * (16 * 56 * 56 + 3 * 112 * 112),
                [&] { return launch_convt2_small<T>(d3, vae.dec4_w, vae.dec4_b, xhat, B, 56, 56, ACT_LEAKY, cur); }));
    if (after_chain && !arena.dry) GCV_TRY(after_chain());
    if (split) { GCV_TRY(run_convnext(bb_vae, &segs[1], 1, true)); }
    else { GCV_TRY(run_convnext(bb_vae, segs, 2)); }
    join.now();
    GCV_TRY(run_head(vae.head, feat, B, ACT_RELU, logits));
    if (recon224 || mse) {
      GCV_TRY(run("vae.resize_mse", 30.0 * B * 224 * 224, sizeof(T) * (double)B * (3 * 112 * 112 + 6 * 224 * 224), [&] {
        return launch_resize_mse<T>(xhat, x, (T*)recon224, msepart, mse, B, cur);
      }));
    }
    arena.release(mk);
    return 0;
  }

  // standalone backbone pass (unit parity / A5): x NCHW (B,3,res,res) -> (B,1000) in T
  int convnext_forward(int which, const void* xv, int B, int res, void* logits1000, hipStream_t s) override {
    GCV_REQUIRE(which == 0 ? has_ed : has_vae, "backbone weights not loaded");
    GCV_TRY(check_batch(B));
    GCV_REQUIRE(res % 4 == 0 && res >= 32 && res <= 224, "resolution must be a multiple of 4 in [32,224]");
    cur = s;
    Seg<T> seg{(const T*)xv, (int64_t)3 * res * res, (int64_t)res * res, res, 1, B, res, res, (T*)logits1000, 1000, ACT_NONE};
    return run_convnext(which == 0 ? bb_ed : bb_vae, &seg, 1);
  }

  int swin_forward(const void* xv, int B, void* logits1000, hipStream_t s) override {
    if (!arena.dry) {
      GCV_REQUIRE(has_swin, "Swin weights not loaded (gcv_load_swin)");
      GCV_TRY(check_batch(B));
    }
    cur = s;
    if (!arena.dry) GCV_REQUIRE(xv && logits1000, "null input/output");
    return run_swin<T>(*this, swin, (const T*)xv, B, (T*)logits1000);
  }

  int init() override {
    GCV_CHECK_HIP(hipSetDevice(device));
    arena.dry = true;
    arena.off = arena.peak = 0;
    int rc = ed_forward(nullptr, max_batch, nullptr, nullptr);
    if (!rc) rc = vae_forward(nullptr, nullptr, max_batch, nullptr, nullptr, nullptr, nullptr, nullptr);
    in_ensemble = true;                    // both VAE schedules (see vae_split_env): the arena holds the larger footprint
    if (!rc) rc = vae_forward(nullptr, nullptr, max_batch, nullptr, nullptr, nullptr, nullptr, nullptr);
    in_ensemble = false;
    if (!rc) rc = swin_forward(nullptr, max_batch, nullptr, nullptr);
    arena.dry = false;
    if (rc) return rc;
    arena.cap = arena.peak + 4096;
    arena.off = 0;
    GCV_CHECK_HIP(hipMalloc((void**)&arena.base, arena.cap));
    return 0;
  }
};

}  // namespace gcv
