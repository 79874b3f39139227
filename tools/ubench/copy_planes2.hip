// copy_planes2.hip — second sweep of the 7-planes-in / 7-planes-out copy: block size, XCD-aware tile
// order, read-only and write-only ceilings (all accesses non-temporal, 8 B per lane unless noted).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0 copy, 1 read only, 2 write only.  XCD: remap block -> tile so that each of the 8 XCDs (blocks are
// dealt round-robin to XCDs) walks one contiguous eighth of the planes.
template <class V, int BS, int MODE, int XCD>
__global__ void __launch_bounds__(BS) k(const float* __restrict__ in, float* __restrict__ out, long n, float* sink) {
  constexpr int W = sizeof(V) / 4;
  long tile = blockIdx.x;
  if (XCD) { const long per = gridDim.x / 8; tile = (blockIdx.x % 8) * per + blockIdx.x / 8; }
  const long base = tile * BS * W;
  V r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (MODE != 2) r[k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(in + k * n + base) + threadIdx.x);
    else r[k] = V(1.0f);
  }
  if (MODE == 1) {
    float s = 0; 
#pragma unroll
    for (int k = 0; k < 7; ++k) s += *reinterpret_cast<float*>(&r[k]);
    if (s == 12345.0f) *sink = s;
    return;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(r[k], reinterpret_cast<V*>(out + k * n + base) + threadIdx.x);
}

template <class V, int BS, int MODE, int XCD>
void run(const char* name, const float* in, float* out, long n, float* sink) {
  constexpr int W = sizeof(V) / 4;
  const int blocks = (int)(n / (BS * W));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k<V, BS, MODE, XCD><<<blocks, BS>>>(in, out, n, sink);
  CHK(hipDeviceSynchronize());
  const int reps = 20;
  CHK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k<V, BS, MODE, XCD><<<blocks, BS>>>(in, out, n, sink);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (MODE == 0 ? 56.0 : 28.0) * n;
  printf("%-52s %7.4f ms  %7.1f GB/s\n", name, ms, bytes / (ms * 1e6));
}

int main() {
  const long n = 64L << 20;
  float *in, *out, *sink;
  CHK(hipMalloc(&in, 7 * n * sizeof(float))); CHK(hipMalloc(&out, 7 * n * sizeof(float))); CHK(hipMalloc(&sink, 4));
  CHK(hipMemset(in, 0, 7 * n * sizeof(float)));
  run<f2, 64, 0, 0>("copy 8 B/lane block 64", in, out, n, sink);
  run<f2, 128, 0, 0>("copy 8 B/lane block 128", in, out, n, sink);
  run<f2, 256, 0, 0>("copy 8 B/lane block 256", in, out, n, sink);
  run<f2, 512, 0, 0>("copy 8 B/lane block 512", in, out, n, sink);
  run<f2, 1024, 0, 0>("copy 8 B/lane block 1024", in, out, n, sink);
  run<float, 256, 0, 0>("copy 4 B/lane block 256", in, out, n, sink);
  run<float, 512, 0, 0>("copy 4 B/lane block 512", in, out, n, sink);
  run<f2, 256, 0, 1>("copy 8 B/lane block 256, XCD-contiguous tiles", in, out, n, sink);
  run<f2, 512, 0, 1>("copy 8 B/lane block 512, XCD-contiguous tiles", in, out, n, sink);
  run<f4, 256, 0, 1>("copy 16 B/lane block 256, XCD-contiguous tiles", in, out, n, sink);
  run<f2, 256, 1, 0>("read only 8 B/lane block 256", in, out, n, sink);
  run<f4, 256, 1, 0>("read only 16 B/lane block 256", in, out, n, sink);
  run<f2, 256, 2, 0>("write only 8 B/lane block 256", in, out, n, sink);
  run<f4, 256, 2, 0>("write only 16 B/lane block 256", in, out, n, sink);
  return 0;
}
