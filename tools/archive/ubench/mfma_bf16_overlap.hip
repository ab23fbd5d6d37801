// Micro-benchmark: bf16 MFMA issue rate on gfx950 (the K = 16 legacy shape vs the K = 32 gfx950 shape) and which VALU / LDS
// instructions hide beside them, from the same wave and from a second wave on the SIMD (the fp32 twin is mfma_overlap.hip).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_bf16_overlap.hip -o mfma_bf16_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: v_mfma_f32_16x16x16_bf16 (4 bf16 per lane), 1: v_mfma_f32_16x16x32_bf16 (8 per lane)
template <int MODE, int K, int SHAPE>
__global__ void bench(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  bf16x8 A8, B8;
  for (int e = 0; e < 8; ++e) { A8[e] = (__bf16)(a + e); B8[e] = (__bf16)(b - e); }
  s16x4 A4 = {1, 2, 3, 4}, B4 = {5, 6, 7, 8};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b - i};
  float s[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { s[i] = a * i; u[i] = threadIdx.x * 77u + i; }
  lds[threadIdx.x] = a;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A4, B4, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A8, B8, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int r = (i * K + k) & 7;
        if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
        if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[r]) : "v"(s[(r + 1) & 7]));
        if (MODE == 3) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 8)); v[r] = t; }
        if (MODE == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(s[r]), "v"(s[(r + 1) & 7]));
        if (MODE == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]), "v"(u[(r + 3) & 7]));
        if (MODE == 6) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
      }
    }
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1] + s[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x + 1024] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE, int K, int SHAPE>
void run(const char* name, int waves_per_simd) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (256 * 1024 + 2048) * sizeof(float)); hipMalloc(&cyc, 8 * 32);
  hipMemset(out, 0, (256 * 1024 + 2048) * sizeof(float));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((bench<MODE, K, SHAPE>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  unsigned long long cc[32]; hipMemcpy(cc, cyc, 8 * 32, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < 4 * waves_per_simd; ++w) { if (cc[2 * w] < lo) lo = cc[2 * w]; if (cc[2 * w + 1] > hi) hi = cc[2 * w + 1]; }
  const double per_iter = (double)(hi - lo) / iters;
  printf("%-22s K=%d %s waves/SIMD=%d : %8.1f cycles per 16 MFMAs x %d waves = %6.2f cycles/MFMA/SIMD   (%.2f per filler)\n", name, K,
         SHAPE ? "16x16x32" : "16x16x16", waves_per_simd, per_iter, waves_per_simd, per_iter / (16.0 * waves_per_simd),
         K ? per_iter / (16.0 * K * waves_per_simd) : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 0>("mfma only", w);
    run<0, 0, 1>("mfma only", w);
    run<2, 1, 0>("v_add_f32", w);
    run<2, 2, 0>("v_add_f32", w);
    run<2, 4, 0>("v_add_f32", w);
    run<2, 4, 1>("v_add_f32", w);
    run<2, 8, 1>("v_add_f32", w);
    run<1, 2, 0>("v_pk_add_f32", w);
    run<1, 2, 1>("v_pk_add_f32", w);
    run<4, 2, 0>("v_cvt_pk_bf16_f32", w);
    run<4, 4, 1>("v_cvt_pk_bf16_f32", w);
    run<5, 2, 0>("v_perm_b32", w);
    run<6, 2, 0>("v_and_b32", w);
    run<6, 4, 1>("v_and_b32", w);
    run<3, 1, 0>("ds_read_b64", w);
    run<3, 2, 1>("ds_read_b64", w);
  }
  return 0;
}
