// valu_rates.hip — issue-rate microbenchmark for the gfx950 vector ALU.
// Each kernel runs 8 independent chains of one instruction per loop iteration on every lane; the host
// reports wave-instructions per nanosecond per SIMD, so ratios between rows give relative issue costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void __launch_bounds__(256) ub(float* out, int iters, float seed) {
  float a[8]; f2 p[8];
  for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x * 1e-3f + k; p[k] = f2{a[k], a[k] + 0.5f}; }
  const float c = 1.0000001f; const f2 pc = f2{c, c};
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {
#define S(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      REP8(S)
#undef S
    } else if (OP == 1) {
#define S(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
      REP8(S)
#undef S
    } else if (OP == 2) {
#define S(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
      REP8(S)
#undef S
    } else if (OP == 3) {
#define S(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
      REP8(S)
#undef S
    } else if (OP == 4) {
#define S(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
      REP8(S)
#undef S
    } else if (OP == 5) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
      REP8(S)
#undef S
    } else if (OP == 6) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(pc));
      REP8(S)
#undef S
    } else if (OP == 7) {  // compare to an SGPR pair + select through it
#define S(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : "vcc");
      REP8(S)
#undef S
    } else if (OP == 8) {
#define S(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      REP8(S)
#undef S
    } else if (OP == 9) {  // 4 plain + 1 transcendental interleaved: does the transcendental overlap?
#define S(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      REP8(S)
#undef S
      asm volatile("v_rcp_f32 %0, %0" : "+v"(a[0]));
      asm volatile("v_rcp_f32 %0, %0" : "+v"(a[4]));
    } else if (OP == 10) {
#define S(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
      REP8(S)
#undef S
    } else if (OP == 11) {
#define S(k) asm volatile("v_mov_b64 %0, %1" : "+v"(p[k]) : "v"(pc));
      REP8(S)
#undef S
    }
  }
  float s = 0; for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char* name, int per_iter, float* out, int waves_per_simd) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs, 4 waves per block
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  ub<OP><<<blocks, 256>>>(out, 100, 1.0f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  ub<OP><<<blocks, 256>>>(out, iters, 1.0f);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double inst_per_simd = (double)waves_per_simd * iters * per_iter;
  printf("%-28s waves/SIMD %d  %8.3f ms  %7.4f wave-instr/ns/SIMD  (%.2f ns per wave-instr)\n", name, waves_per_simd, ms,
         inst_per_simd / (ms * 1e6), ms * 1e6 / inst_per_simd);
}

int main() {
  float* out; CHK(hipMalloc(&out, 256 * 16 * 256 * sizeof(float)));
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_mul_f32", 8, out, w);
    run<1>("v_pk_mul_f32", 8, out, w);
    run<5>("v_fma_f32", 8, out, w);
    run<6>("v_pk_fma_f32", 8, out, w);
    run<10>("v_pk_add_f32", 8, out, w);
    run<8>("v_max_f32", 8, out, w);
    run<11>("v_mov_b64", 8, out, w);
    run<2>("v_rcp_f32", 8, out, w);
    run<3>("v_sqrt_f32", 8, out, w);
    run<4>("v_rsq_f32", 8, out, w);
    run<7>("v_cmp+v_cndmask (2 instr)", 16, out, w);
    run<9>("8 v_mul + 2 v_rcp", 10, out, w);
  }
  return 0;
}
