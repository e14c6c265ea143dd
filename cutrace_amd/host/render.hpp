// render.hpp — C++ host mirror of the reference's render entry point, on top of the C-ABI.
//
//   reference:  template <S, size_t bounces = 10, size_t tpb = 256>
//               void cutrace::gpu::render(const S &scene, float fudge, float &max,
//                    grid<float> &depth_map, grid<vector> &color_map, grid<vector> &normal_map,
//                    size_t &render_ms, size_t &total_ms);          (inc/kernel.hpp:86-130)
//
// Same argument meaning and the same error behaviour (errors are printed to stderr by the
// C-ABI and execution continues; nothing throws).  `bounces` is a run-time argument here
// (the kernel keeps an explicit stack), `tpb` has no equivalent (the launch geometry is the
// kernel's business).
#ifndef CUTRACE_AMD_RENDER_HPP
#define CUTRACE_AMD_RENDER_HPP
#include <chrono>
#include <cstddef>

#include "cutrace_amd.h"
#include "grid.hpp"

namespace cutrace::gpu {

inline void render(ctr_scene *scene, size_t bounces, float fudge, float &max, grid<float> &depth_map,
                   grid<vector> &color_map, grid<vector> &normal_map, size_t &render_ms, size_t &total_ms) {
  auto start = std::chrono::high_resolution_clock::now();
  uint64_t w = 0, h = 0;
  ctr_scene_size(scene, &w, &h);
  depth_map.resize(w, h);   // kernel.hpp:93-95
  color_map.resize(w, h);
  normal_map.resize(w, h);
  ctr_render_stats st{};
  static_assert(sizeof(vector) == 3 * sizeof(float), "grid<vector> must be packed AoS");
  ctr_render(scene, fudge, (int)bounces, nullptr, depth_map.data(), &color_map.data()->x, &normal_map.data()->x, &st);
  max = st.max_depth;       // kernel.hpp:120-125 (reduced on the GPU instead of a host scan)
  auto end = std::chrono::high_resolution_clock::now();
  render_ms = (size_t)st.kernel_ms;
  total_ms = (size_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

}  // namespace cutrace::gpu
#endif
