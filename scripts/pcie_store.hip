// Microbenchmark: how fast can kernels STORE a 1920x1080 frame (28 B per pixel: depth f32, colour 3xf32, normal 3xf32)
// into page-locked host memory, as a function of the store pattern?  Compared with one hipMemcpyAsync (DMA) of the
// same 58 MB.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/bin/pcie_store scripts/pcie_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// pattern A: what the render kernel does — one wave per TW x TH pixel tile, lane = pixel; depth one dword, colour and
// normal one dwordx3 each
template <int TW, int TH>
__global__ __launch_bounds__(64) void tile_store(float *depth, float *color, float *normal, int w, int h, const float *src) {
  const int tiles_x = (w + TW - 1) / TW;
  const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x, x = tx * TW + lane % TW, y = ty * TH + lane / TW;
  if (x >= w || y >= h) return;
  const size_t px = (size_t)y * w + x;
  const float v = src[px & 1023];
  depth[px] = v;
  float3 c = make_float3(v, v + 1.f, v + 2.f);
  *reinterpret_cast<float3 *>(color + 3 * px) = c;   // (12-byte aligned: the compiler emits one dwordx3)
  *reinterpret_cast<float3 *>(normal + 3 * px) = c;
}

// pattern B: a wave copies the three buffers' parts of a GW x 8 pixel group from device memory, 16 bytes per lane over
// the flattened (row, 16-byte chunk) index — runs of GW*4 / GW*12 bytes
template <int GW>
__global__ __launch_bounds__(64) void group_copy(const float *sd, const float *sc, const float *sn, float *depth, float *color, float *normal, int w, int h) {
  const int groups_x = w / GW;
  const int g = blockIdx.x, gx = g % groups_x, gy = g / groups_x;
  const int lane = threadIdx.x;
  auto part = [&](const float *s, float *d, int fpp) {   // fpp floats per pixel
    const int chunks_row = GW * fpp / 4, total = chunks_row * 8;
    for (int c = lane; c < total; c += 64) {
      const int r = c / chunks_row, k = c % chunks_row;
      const size_t off = ((size_t)(gy * 8 + r) * w + (size_t)gx * GW) * fpp + 4 * (size_t)k;
      *reinterpret_cast<float4 *>(d + off) = *reinterpret_cast<const float4 *>(s + off);
    }
  };
  part(sd, depth, 1);
  part(sc, color, 3);
  part(sn, normal, 3);
}

// pattern C: plain streaming copy, 16 bytes per lane, grid-stride
__global__ __launch_bounds__(256) void stream_copy(const float4 *s, float4 *d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int main() {
  const int w = 1920, h = 1080;
  const size_t px = (size_t)w * h, n = 7 * px;
  float *host = nullptr, *dev = nullptr, *src = nullptr;
  CK(hipHostMalloc((void **)&host, n * sizeof(float), hipHostMallocDefault));
  CK(hipMalloc((void **)&dev, n * sizeof(float)));
  CK(hipMalloc((void **)&src, 1024 * sizeof(float)));
  CK(hipMemset(dev, 0, n * sizeof(float)));
  CK(hipMemset(src, 0, 1024 * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto &&launch) {
    std::vector<float> ms;
    for (int rep = 0; rep < 12; rep++) {
      CK(hipEventRecord(e0, nullptr));
      launch();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (rep >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const float med = ms[ms.size() / 2];
    printf("%-58s %.3f ms  %.1f GB/s\n", name, med, n * 4.0 / med / 1e6);
  };
  timeit("hipMemcpyAsync D2H (DMA), 58 MB", [&] { CK(hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, nullptr)); });
  timeit("tile stores 8x8   (runs 32 / 96 B) -> host", [&] { tile_store<8, 8><<<(w / 8) * (h / 8), 64>>>(host, host + px, host + 4 * px, w, h, src); });
  timeit("tile stores 16x4  (runs 64 / 192 B) -> host", [&] { tile_store<16, 4><<<(w / 16) * (h / 4), 64>>>(host, host + px, host + 4 * px, w, h, src); });
  timeit("tile stores 32x2  (runs 128 / 384 B) -> host", [&] { tile_store<32, 2><<<(w / 32) * (h / 2), 64>>>(host, host + px, host + 4 * px, w, h, src); });
  timeit("tile stores 64x1  (runs 256 / 768 B) -> host", [&] { tile_store<64, 1><<<(w / 64) * h, 64>>>(host, host + px, host + 4 * px, w, h, src); });
  timeit("tile stores 8x8 -> DEVICE memory", [&] { tile_store<8, 8><<<(w / 8) * (h / 8), 64>>>(dev, dev + px, dev + 4 * px, w, h, src); });
  timeit("group copy 64x8 px (runs 256 / 768 B), 16 B per lane -> host", [&] { group_copy<64><<<(w / 64) * (h / 8), 64>>>(dev, dev + px, dev + 4 * px, host, host + px, host + 4 * px, w, h); });
  timeit("group copy 128x8 px (runs 512 / 1536 B) -> host", [&] { group_copy<128><<<(w / 128) * (h / 8), 64>>>(dev, dev + px, dev + 4 * px, host, host + px, host + 4 * px, w, h); });
  timeit("group copy 640x8 px (runs 2560 / 7680 B) -> host", [&] { group_copy<640><<<(w / 640) * (h / 8), 64>>>(dev, dev + px, dev + 4 * px, host, host + px, host + 4 * px, w, h); });
  for (int blocks : {64, 256, 1024, 4096})
    timeit((std::string("stream copy 16 B per lane, ") + std::to_string(blocks) + " blocks -> host").c_str(),
           [&] { stream_copy<<<blocks, 256>>>((const float4 *)dev, (float4 *)host, n / 4); });
  return 0;
}
