// main.cpp — `cutrace <scene file>`: drop-in for the reference CLI (main.cu:8-47).
// Same argv, same exit codes (-1 usage, -2 load failure), same stdout lines, same three
// output files in the CWD.  bounces = 5 and fudge = 1e-3 are the values main.cu:30 uses.
//
// Extras (ignored by anything that drives the reference): environment overrides
//   CUTRACE_BOUNCES, CUTRACE_WIDTH, CUTRACE_HEIGHT, CUTRACE_DEVICE
// because the reference has no flags (main.cu:8-12) and benchmarking needs them, and
//   CUTRACE_DEVICES=N   row-tile the frame over the first N GPUs of the node (ctr_render_multi: one RCCL
//                       gather to GPU 0, same three files); CUTRACE_DEVICE_LIST=0,2,... names them instead
#include <cstdlib>
#include <iostream>
#include <thread>
#include <vector>

#include "cutrace_amd.h"
#include "cutrace_host.h"
#include "grid.hpp"
#include "render.hpp"

static long env_long(const char *name, long def) {
  const char *v = getenv(name);
  return (v && *v) ? atol(v) : def;
}

// the three output files of main.cu:34-41 (inc/images.hpp:26-88), written at the same time: they are independent, and a 1080p JPEG is
// 10-45 ms of host work each (138 ms one after the other with a single-threaded encoder, 15 ms now: scripts/cpu_jpeg_time.py) in a
// process whose render call takes 13
static void write_images(cutrace::grid<float> &depth_map, cutrace::grid<cutrace::vector> &color_map,
                         cutrace::grid<cutrace::vector> &normal_map, float max_d) {
  std::thread t_depth([&] { ctr_write_depth_map("./depth_map.jpg", depth_map.data(), depth_map.cols(), depth_map.rows(), max_d); });
  std::thread t_normal([&] { ctr_write_normal_map("./normal_map.jpg", &normal_map.data()->x, normal_map.cols(), normal_map.rows()); });
  ctr_write_colorized("./frame.jpg", &color_map.data()->x, color_map.cols(), color_map.rows());
  t_depth.join();
  t_normal.join();
}

int main(int argc, const char **argv) {
  if (argc < 2) {
    std::cerr << "Usage: " << argv[0] << " <scene file>\n";
    return -1;
  }

  ctr_host_scene *hs = nullptr;
  int st = ctr_host_scene_load(argv[1], &hs);
  if (st != CTR_OK) {
    ctr_dump_schema();
    ctr_host_scene_free(hs);
    return -2;
  }
  long ow = env_long("CUTRACE_WIDTH", 0), oh = env_long("CUTRACE_HEIGHT", 0);
  const ctr_scene_desc *desc = ctr_host_scene_desc(hs);
  if (ow > 0 || oh > 0) {
    ctr_host_scene_set_size(hs, ow > 0 ? (uint64_t)ow : desc->cam.w, oh > 0 ? (uint64_t)oh : desc->cam.h);
    desc = ctr_host_scene_desc(hs);
  }

  long n_dev = env_long("CUTRACE_DEVICES", 1);
  std::vector<int> devs;
  if (const char *list = getenv("CUTRACE_DEVICE_LIST")) {  // explicit list, e.g. "0,2,4,6" (or "0,0": rehearsal on one GPU)
    for (const char *p = list; *p;) {
      devs.push_back(atoi(p));
      while (*p && *p != ',') p++;
      if (*p == ',') p++;
    }
    n_dev = (long)devs.size();
  } else {
    for (long i = 0; i < n_dev; i++) devs.push_back((int)i);
  }
  if (n_dev > 1) {
    ctr_multi *group = nullptr;
    if (ctr_multi_create(desc, devs.data(), (int)n_dev, &group) != CTR_OK) {
      ctr_host_scene_free(hs);
      return -3;
    }
    ctr_dump_scene(desc);
    float max_d;
    cutrace::grid<float> depth_map;
    cutrace::grid<cutrace::vector> color_map;
    cutrace::grid<cutrace::vector> normal_map;
    size_t render, total;
    cutrace::gpu::render_multi(group, (size_t)env_long("CUTRACE_BOUNCES", 5), 1e-3, max_d, depth_map, color_map, normal_map,
                               render, total);
    std::cout << "Render time was " << render << " ms; kernel time with setup/teardown was " << total << " ms.\n";
    write_images(depth_map, color_map, normal_map, max_d);
    ctr_multi_destroy(group);
    ctr_host_scene_free(hs);
    return 0;
  }

  ctr_scene *scene = nullptr;
  if (ctr_scene_create(desc, (int)env_long("CUTRACE_DEVICE", 0), &scene) != CTR_OK) {
    // message already printed (print-and-continue is all the reference does, inc/cuda.hpp:12-22);
    // without a scene there is nothing to render
    ctr_host_scene_free(hs);
    return -3;
  }

  ctr_dump_scene(desc);

  float max_d;
  cutrace::grid<float> depth_map;
  cutrace::grid<cutrace::vector> color_map;
  cutrace::grid<cutrace::vector> normal_map;
  size_t render, total;
  cutrace::gpu::render(scene, (size_t)env_long("CUTRACE_BOUNCES", 5), 1e-3, max_d, depth_map, color_map, normal_map,
                       render, total);

  std::cout << "Render time was " << render << " ms; kernel time with setup/teardown was " << total << " ms.\n";

  write_images(depth_map, color_map, normal_map, max_d);

  ctr_scene_destroy(scene);
  ctr_host_scene_free(hs);
  return 0;
}
