// Swin-T (A6: the HybridEmbed "embedder", constructed but never executed by the reference's
// forward — SURVEY.md §0.4): packed weights + forward schedule.  Measured separately from the logits path.
#pragma once
#include "net.h"

namespace gcv {

struct WeightStore;

template <typename T> struct SwinW {
  int placeholder = 0;
};

template <typename T> int pack_swin(const TensorMap&, const std::string&, WeightStore&, SwinW<T>&) {
  set_error("Swin-T path not built yet");
  return -7;
}

template <typename T, typename Net> int run_swin(Net& net, const SwinW<T>&, const T*, int, T*) {
  if (net.arena.dry) return 0;
  set_error("Swin-T path not built yet");
  return -7;
}

}  // namespace gcv
