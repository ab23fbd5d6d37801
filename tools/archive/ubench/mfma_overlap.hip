// Micro-benchmark: which instruction classes overlap with fp32 MFMA on gfx950, from the same wave and from a second
// wave on the SIMD.  Build: hipcc -O3 --offload-arch=gfx950 mfma_overlap.hip -o mfma_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int K, int BIG>
__global__ void bench(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x4 acc[16];
  f32x16 accb[4];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b - i};
  float s[8];
  for (int i = 0; i < 8; ++i) s[i] = a * i;
  lds[threadIdx.x] = a;
  __syncthreads();
  const float* gp = out + (threadIdx.x & 63) * 4;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (BIG) accb[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accb[i & 3], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int r = (i * K + k) & 7;
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
        if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[r]) : "v"(s[(r + 1) & 7]));
        if (MODE == 3) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 8)); v[r] = t; }
        if (MODE == 4) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gp)); v[r] = f32x2{t[0], t[1]}; }
        if (MODE == 5) asm volatile("s_nop 0");
        if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(s[r]) : "v"(s[(r + 1) & 7]));
      }
    }
    if (MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
  for (int i = 0; i < 4; ++i) r += accb[i][0];
  for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1] + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x + 1024] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE, int K, int BIG>
void run(const char* name, int waves_per_simd) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (256 * 1024 + 2048) * sizeof(float)); hipMalloc(&cyc, 8 * 32);
  hipMemset(out, 0, (256 * 1024 + 2048) * sizeof(float));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((bench<MODE, K, BIG>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  unsigned long long cc[32]; hipMemcpy(cc, cyc, 8 * 32, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < 4 * waves_per_simd; ++w) { if (cc[2 * w] < lo) lo = cc[2 * w]; if (cc[2 * w + 1] > hi) hi = cc[2 * w + 1]; }
  const double per_iter = (double)(hi - lo) / iters;
  const double mfma_cycles = 16.0 * (BIG ? 64 : 32) * waves_per_simd;
  printf("%-28s K=%d %s waves/SIMD=%d : %8.1f cycles/iter (all waves)  MFMA-only ideal %6.0f  -> extra %7.1f  (per extra instr %.2f)\n", name, K,
         BIG ? "32x32x2" : "16x16x4", waves_per_simd, per_iter, mfma_cycles, per_iter - mfma_cycles, K ? (per_iter - mfma_cycles) / (16.0 * K * waves_per_simd) : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 0>("mfma only", w);
    run<0, 0, 1>("mfma only", w);
    run<1, 2, 0>("v_pk_fma_f32", w);
    run<1, 4, 0>("v_pk_fma_f32", w);
    run<1, 4, 1>("v_pk_fma_f32", w);
    run<2, 4, 0>("v_add_f32", w);
    run<2, 8, 1>("v_add_f32", w);
    run<6, 4, 0>("v_mov_b32", w);
    run<3, 1, 0>("ds_read_b64", w);
    run<3, 2, 1>("ds_read_b64", w);
    run<4, 1, 0>("global_load_dwordx4 (L1)", w);
    run<5, 4, 0>("s_nop", w);
    run<1, 1, 0>("v_pk_fma_f32", w);
    run<1, 1, 1>("v_pk_fma_f32", w);
    run<1, 2, 1>("v_pk_fma_f32", w);
  }
  return 0;
}
