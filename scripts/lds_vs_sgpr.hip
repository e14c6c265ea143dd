// lds_vs_sgpr.hip — gfx950 microbenchmark for the north-star wording "triangle batches staged through LDS".
//
// In the render kernel all 64 lanes of a wave test the SAME 64-byte record (triangle / node) at the same time.
// Two ways to get such a record to the VALU:
//   sgpr: one s_load_dwordx16 into SGPRs, the VALU instructions read them as scalar operands (what the kernel does;
//         such instructions issue at half rate, scripts/valu_issue.hip)
//   lds : the workgroup stages a batch of records in LDS (coalesced global loads, one barrier), then every wave reads
//         each record as a broadcast (4 x ds_read_b128, all lanes the same address) into 16 VGPRs and the VALU
//         instructions run at full rate on VGPR operands
// Both do the same arithmetic per record and lane — 16 v_fma_f32, or 8 v_pk_fma_f32 on the record's aligned pairs,
// which is the form the render kernel's tests have — over the same records, with 1..8 waves per SIMD on every CU.  Printed: ns per (wave x record) per SIMD, i.e. SIMD time one record costs.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/lds_vs_sgpr scripts/lds_vs_sgpr.hip && scripts/bin/lds_vs_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REC = 16;       // floats per record (64 bytes)
constexpr int BATCH = 256;    // records staged per batch (16 KB of LDS per workgroup)

typedef float f2 __attribute__((ext_vector_type(2)));

// PACKED = false: 16 v_fma_f32 per record; true: 8 v_pk_fma_f32 (the record's floats taken as aligned pairs, which is
// how the render kernel's node and triangle tests consume them)
template <bool PACKED>
__device__ __forceinline__ float work(const float *r, float x, float acc) {
  if (PACKED) {
    f2 a[4] = {{acc, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const f2 xx = {x, x};
#pragma unroll
    for (int j = 0; j < REC / 2; j++) {
      const f2 rr = {r[2 * j], r[2 * j + 1]};
      a[j & 3] = __builtin_elementwise_fma(rr, xx, a[j & 3]);
    }
    const f2 t = (a[0] + a[1]) + (a[2] + a[3]);
    return t.x + t.y;
  }
  float a[4] = {acc, 0.f, 0.f, 0.f};  // four independent chains
#pragma unroll
  for (int j = 0; j < REC; j++) a[j & 3] = __builtin_fmaf(r[j], x, a[j & 3]);
  return (a[0] + a[1]) + (a[2] + a[3]);
}

template <bool PACKED>
__global__ __launch_bounds__(256) void via_sgpr(const float *__restrict__ recs, int n_rec, float *out, float x) {
  float acc = (float)threadIdx.x;
  for (int i = 0; i < n_rec; i++) {
    const float *r = recs + (size_t)__builtin_amdgcn_readfirstlane(i) * REC;  // wave-uniform address: scalar loads
    float v[REC];
#pragma unroll
    for (int j = 0; j < REC; j++) v[j] = r[j];
    acc = work<PACKED>(v, x, acc);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the scalar path again, every wave walking its own pseudo-random sequence of records (a BVH walk is not a stream):
// no two waves ask for the same line at the same time, so nothing merges on the way to L2
__global__ __launch_bounds__(256) void via_sgpr_scattered(const float *__restrict__ recs, int n_rec, int steps, float *out, float x) {
  float acc = (float)threadIdx.x;
  uint32_t i = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u);
  for (int s = 0; s < steps; s++) {
    i = i * 1664525u + 1013904223u;  // (scalar ALU)
    const float *r = recs + (size_t)((i >> 8) & (uint32_t)(n_rec - 1)) * REC;  // (n_rec: a power of two)
    float v[REC];
#pragma unroll
    for (int j = 0; j < REC; j++) v[j] = r[j];
    acc = work<true>(v, x, acc);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void via_lds(const float *__restrict__ recs, int n_rec, float *out, float x) {
  __shared__ float4 stage[BATCH * REC / 4];
  float acc = (float)threadIdx.x;
  for (int b0 = 0; b0 < n_rec; b0 += BATCH) {
    __syncthreads();
    // the batch, coalesced: 256 threads x 16 bytes = 4 KB per pass, 4 passes
    const float4 *src = reinterpret_cast<const float4 *>(recs + (size_t)b0 * REC);
#pragma unroll
    for (int p = 0; p < BATCH * REC / 4 / 256; p++) stage[p * 256 + threadIdx.x] = src[p * 256 + threadIdx.x];
    __syncthreads();
    for (int i = 0; i < BATCH; i++) {
      float v[REC];
#pragma unroll
      for (int q = 0; q < REC / 4; q++) {
        const float4 t = stage[i * (REC / 4) + q];  // every lane the same address: broadcast read
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      acc = work<PACKED>(v, x, acc);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const int n_rec = 16384;  // 1 MB of records: L2-resident, like a mesh
  float *recs, *out;
  CK(hipMalloc((void **)&recs, (size_t)n_rec * REC * sizeof(float)));
  std::vector<float> h((size_t)n_rec * REC);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f / (float)(1 + i % 97);
  CK(hipMemcpy(recs, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%d CUs; %d records of 64 B; ns of SIMD time per (wave x record)\n", n_cu, n_rec);
  for (int packed = 0; packed < 2; packed++) {
    printf("%s\n%-16s %10s %10s %8s\n", packed ? "8 v_pk_fma_f32 per record and lane" : "16 v_fma_f32 per record and lane",
           "waves per SIMD", "sgpr", "lds", "lds/sgpr");
    for (int k : {1, 2, 4, 6, 8}) {
      const int blocks = n_cu * k;  // 256-thread blocks: 4 waves = one per SIMD; k blocks per CU
      float ms[2] = {0, 0};
      for (int which = 0; which < 2; which++) {
        std::vector<float> t;
        for (int rep = 0; rep < 7; rep++) {
          CK(hipEventRecord(e0, nullptr));
          if (which == 0) {
            if (packed) via_sgpr<true><<<blocks, 256>>>(recs, n_rec, out, 1.0001f);
            else via_sgpr<false><<<blocks, 256>>>(recs, n_rec, out, 1.0001f);
          } else {
            if (packed) via_lds<true><<<blocks, 256>>>(recs, n_rec, out, 1.0001f);
            else via_lds<false><<<blocks, 256>>>(recs, n_rec, out, 1.0001f);
          }
          CK(hipEventRecord(e1, nullptr));
          CK(hipEventSynchronize(e1));
          float x;
          CK(hipEventElapsedTime(&x, e0, e1));
          if (rep >= 2) t.push_back(x);
        }
        std::sort(t.begin(), t.end());
        ms[which] = t[t.size() / 2];
      }
      // every SIMD ran k waves, each n_rec records: SIMD time per wave-record = elapsed / (k * n_rec)
      const double ns_s = ms[0] * 1e6 / ((double)k * n_rec), ns_l = ms[1] * 1e6 / ((double)k * n_rec);
      printf("%-16d %10.2f %10.2f %8.2f\n", k, ns_s, ns_l, ns_l / ns_s);
    }
  }
  printf("scalar path, every wave its own random sequence of records, 8 v_pk_fma_f32 per record; by working set\n");
  printf("%-16s %12s %12s %12s %12s\n", "waves per SIMD", "8 KB", "64 KB", "1 MB", "4 MB");
  float *big;
  CK(hipMalloc((void **)&big, (size_t)65536 * REC * sizeof(float)));
  CK(hipMemset(big, 0, (size_t)65536 * REC * sizeof(float)));
  for (int k : {1, 2, 4, 6, 8}) {
    printf("%-16d", k);
    for (int set : {128, 1024, 16384, 65536}) {
      const int steps = 16384, blocks = n_cu * k;
      std::vector<float> t;
      for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, nullptr));
        via_sgpr_scattered<<<blocks, 256>>>(big, set, steps, out, 1.0001f);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float x;
        CK(hipEventElapsedTime(&x, e0, e1));
        if (rep >= 2) t.push_back(x);
      }
      std::sort(t.begin(), t.end());
      printf(" %12.2f", t[t.size() / 2] * 1e6 / ((double)k * steps));
    }
    printf("\n");
  }
  return 0;
}
