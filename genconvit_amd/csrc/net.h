// Host-side driver of the GenConViT path: weight packing, workspace arena, per-launch
// profiler and the ED / VAE / ConvNeXt-T forward schedules (one instantiation per storage dtype).
#pragma once
#include <map>
#include <memory>
#include <functional>
#include <string>
#include <vector>

#include "common.h"

namespace gcv {

struct TensorRef {
  const float* data;
  int64_t numel;
  bool on_device;
};
typedef std::map<std::string, TensorRef> TensorMap;

// ---- per-launch profiler (HIP events on the launch stream) ------------------
struct ProfRecord {
  std::string tag;      // semantic op, e.g. "cnx.pw1", "cnx.dwconv_ln"
  double flops;         // algorithmic FLOPs of the launch (2*M*N*K for GEMMs)
  double bytes;         // algorithmic HBM bytes of the launch (unique reads + writes)
  hipEvent_t e0, e1;
};

// roctx ranges: while a handle's profiling is on, every tagged launch is also bracketed by a
// roctxRangePushA(tag) / roctxRangePop pair, so `rocprofv3 --marker-trace --kernel-trace` groups the kernels by the same
// semantic tags gcv_profile_report() uses.  The marker library is bound at run time (rocprofiler-sdk's roctx, then the legacy
// libroctx64) and only when profiling is switched on; without it the calls are no-ops.
void roctx_push(const char* tag);
void roctx_pop();

struct Profiler {
  bool enabled = false;
  std::vector<ProfRecord> recs;
  std::vector<hipEvent_t> pool;
  size_t next_event = 0;
  hipEvent_t get_event();
  void reset() { recs.clear(); next_event = 0; }
  ~Profiler();
};

// ---- abstract network (dtype erased) ----------------------------------------
struct NetBase {
  int device = 0;
  int dtype = 0;
  int max_batch = 0;
  Profiler prof;
  bool in_ensemble = false;   // set by gcv_genconvit_forward around vae_forward: the VAE shares the GPU with the ED network
  // host-side enqueue order of the ensemble (gcv_genconvit_forward): called by vae_forward once its encoder -> mu ->
  // decoder chain is enqueued and before its backbone pass, to enqueue the ED network on its own stream in between
  std::function<int()> after_chain;
  virtual ~NetBase() {}
  virtual int init() = 0;
  virtual int load_ed(const TensorMap& w) = 0;
  virtual int load_vae(const TensorMap& w) = 0;
  virtual int load_swin(const TensorMap& w, const std::string& prefix) = 0;
  virtual int ed_forward(const void* x, int B, float* logits, hipStream_t s) = 0;
  virtual int vae_forward(const void* x, const float* eps, int B, float* logits, void* recon224, float* mse,
                          float* kl, hipStream_t s) = 0;
  virtual int convnext_forward(int which /*0 ed backbone, 1 vae backbone*/, const void* x, int B, int res,
                               void* logits1000, hipStream_t s) = 0;
  virtual int swin_forward(const void* x, int B, void* logits1000, hipStream_t s) = 0;
  virtual size_t workspace_bytes() const = 0;
};

NetBase* make_net_f32();
NetBase* make_net_f16();
NetBase* make_net_bf16();

}  // namespace gcv
