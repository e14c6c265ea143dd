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

// kernel.hpp:93-95 resizes the three grids; here they become the three parts of ONE page-locked block
// (kept for the life of the process, like the reference's scene memory), so that the frame arrives with a
// single direct DMA instead of kernel.hpp:110-114's 3·h row copies through pageable memory
inline void adopt_frame(uint64_t w, uint64_t h, grid<float> &depth_map, grid<vector> &color_map, grid<vector> &normal_map) {
  static_assert(sizeof(vector) == 3 * sizeof(float), "grid<vector> must be packed AoS");
  static float *frame = nullptr;
  static uint64_t frame_px = 0;
  float *fd = nullptr, *fc = nullptr, *fn = nullptr;
  if (frame_px != w * h) {
    ctr_frame_free(frame);
    frame = nullptr;
    frame_px = 0;
    if (w * h && ctr_frame_alloc(w * h, &fd, &fc, &fn) == CTR_OK) { frame = fd; frame_px = w * h; }
  }
  if (frame) {
    depth_map.adopt(frame, w, h);
    color_map.adopt(reinterpret_cast<vector *>(frame + w * h), w, h);
    normal_map.adopt(reinterpret_cast<vector *>(frame + 4 * w * h), w, h);
  } else {  // no page-locked memory to be had: ordinary grids, staged copies
    depth_map.resize(w, h);
    color_map.resize(w, h);
    normal_map.resize(w, h);
  }
}

inline void render(ctr_scene *scene, size_t bounces, float fudge, float &max, grid<float> &depth_map,
                   grid<vector> &color_map, grid<vector> &normal_map, size_t &render_ms, size_t &total_ms) {
  auto start = std::chrono::high_resolution_clock::now();
  uint64_t w = 0, h = 0;
  ctr_scene_size(scene, &w, &h);
  adopt_frame(w, h, depth_map, color_map, normal_map);
  ctr_render_stats st{};
  ctr_render(scene, fudge, (int)bounces, nullptr, depth_map.data(), &color_map.data()->x, &normal_map.data()->x, &st);
  max = st.max_depth;       // kernel.hpp:120-125 (reduced on the GPU instead of a host scan)
  auto end = std::chrono::high_resolution_clock::now();
  render_ms = (size_t)st.kernel_ms;
  total_ms = (size_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

// the same frame row-tiled over the devices of a ctr_multi group (ctr_render_multi)
inline void render_multi(ctr_multi *group, size_t bounces, float fudge, float &max, grid<float> &depth_map,
                         grid<vector> &color_map, grid<vector> &normal_map, size_t &render_ms, size_t &total_ms) {
  auto start = std::chrono::high_resolution_clock::now();
  ctr_render_stats st{};
  uint64_t w = 0, h = 0;
  ctr_multi_size(group, &w, &h);
  adopt_frame(w, h, depth_map, color_map, normal_map);
  ctr_render_multi(group, fudge, (int)bounces, 8, depth_map.data(), &color_map.data()->x, &normal_map.data()->x, &st);
  max = st.max_depth;
  auto end = std::chrono::high_resolution_clock::now();
  render_ms = (size_t)st.kernel_ms;
  total_ms = (size_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

}  // namespace cutrace::gpu
#endif
