// copy_planes.hip — HBM ceiling for the ray-state access pattern of the sequential trace:
// 7 float planes read and 7 written per ray (56 B/ray), no arithmetic.  Variants differ in bytes per lane
// per access, tiles per block, and cache policy of the accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <class V, int NT> __device__ inline V ldv(const V* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <class V, int NT> __device__ inline void stv(V* p, V v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// one tile (256 lanes x sizeof(V)) per block per plane; TPB tiles per block processed back to back
template <class V, int NT, int TPB>
__global__ void __launch_bounds__(256) k_copy(const float* __restrict__ in, float* __restrict__ out, long n) {
  constexpr int W = sizeof(V) / 4;
  for (int t = 0; t < TPB; ++t) {
    const long base = ((long)blockIdx.x * TPB + t) * 256 * W;
    if (base >= n) return;
    V r[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) r[k] = ldv<V, NT>(reinterpret_cast<const V*>(in + k * n + base) + threadIdx.x);
#pragma unroll
    for (int k = 0; k < 7; ++k) stv<V, NT>(reinterpret_cast<V*>(out + k * n + base) + threadIdx.x, r[k]);
  }
}

template <class V, int NT, int TPB>
void run(const char* name, const float* in, float* out, long n) {
  constexpr int W = sizeof(V) / 4;
  const long tiles = n / (256 * W);
  const int blocks = (int)((tiles + TPB - 1) / TPB);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k_copy<V, NT, TPB><<<blocks, 256>>>(in, out, n);
  CHK(hipDeviceSynchronize());
  const int reps = 20;
  CHK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k_copy<V, NT, TPB><<<blocks, 256>>>(in, out, n);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-44s %7.4f ms  %7.1f GB/s\n", name, ms, 56.0 * n / (ms * 1e6));
}

int main() {
  const long n = 64L << 20;
  float *in, *out;
  CHK(hipMalloc(&in, 7 * n * sizeof(float))); CHK(hipMalloc(&out, 7 * n * sizeof(float)));
  CHK(hipMemset(in, 0, 7 * n * sizeof(float)));
  run<float, 0, 1>("4 B/lane, 1 tile/block", in, out, n);
  run<f2, 0, 1>("8 B/lane, 1 tile/block", in, out, n);
  run<f4, 0, 1>("16 B/lane, 1 tile/block", in, out, n);
  run<f2, 1, 1>("8 B/lane, 1 tile/block, nontemporal", in, out, n);
  run<f4, 1, 1>("16 B/lane, 1 tile/block, nontemporal", in, out, n);
  run<f2, 0, 4>("8 B/lane, 4 tiles/block", in, out, n);
  run<f2, 1, 4>("8 B/lane, 4 tiles/block, nontemporal", in, out, n);
  run<f4, 0, 4>("16 B/lane, 4 tiles/block", in, out, n);
  run<f2, 0, 16>("8 B/lane, 16 tiles/block", in, out, n);
  CHK(hipMemcpy(out, in, 7 * n * sizeof(float), hipMemcpyDeviceToDevice));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) CHK(hipMemcpyAsync(out, in, 7 * n * sizeof(float), hipMemcpyDeviceToDevice, 0));
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("%-44s %7.4f ms  %7.1f GB/s\n", "hipMemcpy D2D", ms, 56.0 * n / (ms * 1e6));
  return 0;
}
