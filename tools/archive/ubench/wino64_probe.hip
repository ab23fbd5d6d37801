// Feasibility probe (timing only, NO epilogue: results are not a convolution): the main loop of a Winograd F(2x2x2, 3x3x3) kernel
// shaped "one wave per SIMD, 64 output channels per wave" -- 4 waves = the 4 xi_z of ONE tile z-row (16 tiles = 2 x 8 x 8 voxels),
// 64 MFMA 16x16x4 per k-step and wave (256 accumulator registers), against the production shape of conv_wino.hip (8 waves, two per
// SIMD, 32 output channels per wave, 32 MFMAs per k-step, 128 accumulators).  The operand side is the production kernel's: halo block
// of a 16-channel chunk staged channel-major in LDS (double buffered, one barrier per chunk), raw reads one k-step ahead, the
// separable B^T transform in packed fp32, weights straight from L2 into MFMA B registers one k-step ahead.
// Build: hipcc -O3 --offload-arch=gfx950 wino64_probe.hip -o wino64_probe ; run on the GPU box: ./wino64_probe [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 256;
constexpr int CKW = 16;
constexpr int PY = 12, PZ = 144;
constexpr int HZ = 4, HY = 10, HX = 10, HV = HZ * HY * HX;      // halo block 4 x 10 x 10
constexpr int CP = HZ * PZ + 2;
constexpr int NLOAD = (HV * 4 + kT - 1) / kT;                    // 7
constexpr int BUF = CKW * CP;
constexpr int NB = 4;                                            // 16-cout blocks per wave

struct Args {
  const float* x;
  const f32x4* wp;
  float* y;
  int B, D, H, W, Cin, Cout;
  int nbz, nby, nbx, ntb, ncs;
};

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_bt01(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
__device__ __forceinline__ f32x2 pk_bt23(f32x2 p, f32x2 q) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
  return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__global__ __launch_bounds__(kT, 1) void wino64_main_loop(const Args a) {
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int xz = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = xi_z
  const int tl = lane & 15, kq = lane >> 4;
  const int tx = tl & 3, ty = tl >> 2;

  const int cs = blockIdx.x % a.ncs;
  const int tb0 = blockIdx.x / a.ncs, tstride = gridDim.x / a.ncs;
  if (tb0 >= a.ntb) return;
  const int niter = (a.ntb - tb0 + tstride - 1) / tstride;

  struct Blk { const float* xb; int hoff, z0, y0, x0; };
  auto decode = [&](int t) -> Blk {
    Blk bi;
    const int bx = t % a.nbx;
    int t2 = t / a.nbx;
    const int by = t2 % a.nby; t2 /= a.nby;
    const int bz = t2 % a.nbz;
    const int b = t2 / a.nbz;
    bi.z0 = bz * 2; bi.y0 = by * 8; bi.x0 = bx * 8;
    bi.xb = a.x + static_cast<long long>(b) * a.D * a.H * a.W * a.Cin;
    bi.hoff = (((bi.z0 - 1) * a.H + (bi.y0 - 1)) * a.W + (bi.x0 - 1)) * a.Cin;
    return bi;
  };
  int ldst[NLOAD];
  unsigned so[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    int p = it * kT + tid;
    if (p > HV * 4 - 1) p = HV * 4 - 1;
    const int hv = p >> 2, q4 = p & 3;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    ldst[it] = ((q4 * 4) * CP + hz * PZ + hy * PY + hx) * 4;
  }
  const unsigned vol_bytes = static_cast<unsigned>(a.D * a.H * a.W) * a.Cin * 4u;
  auto set_offs = [&](const Blk& bi) {
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      int p = it * kT + tid;
      if (p > HV * 4 - 1) p = HV * 4 - 1;
      const int hv = p >> 2, q4 = p & 3;
      const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      const int roff = ((hz * a.H + hy) * a.W + hx) * a.Cin + q4 * 4;
      const int gz = bi.z0 - 1 + hz, gy = bi.y0 - 1 + hy, gx = bi.x0 - 1 + hx;
      const bool ok = static_cast<unsigned>(gz) < static_cast<unsigned>(a.D) && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) &&
                      static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
      so[it] = ok ? static_cast<unsigned>(bi.hoff + roff) * 4u : 0x80000000u;
    }
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  auto stage_store = [&](int it, int bufbytes, const f32x4& v) {
    float* d = reinterpret_cast<float*>(sInB + (ldst[it] + bufbytes));
    d[0] = v[0]; d[CP] = v[1]; d[2 * CP] = v[2]; d[3 * CP] = v[3];
  };

  const int za = xz == 0 ? 0 : xz == 2 ? 2 : 1;
  const int zb = xz == 0 ? 2 : xz == 1 ? 2 : xz == 2 ? 1 : 3;
  const float qs = xz == 1 ? 1.f : -1.f;
  const f32x2 qs2 = {qs, qs};
  const int abase = kq * CP + (2 * ty) * PY + 2 * tx;
  const int offAb = (abase + za * PZ) * 4, offBb = (abase + zb * PZ) * 4;
  f32x2 ra[8], rb[8], T[8], U[8], A2[8];
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb, ib = idxbytes + offBb;
    asm volatile("" : "+v"(ia), "+v"(ib));
    __builtin_assume((ia & 7) == 0);
    __builtin_assume((ib & 7) == 0);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      ra[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY) * 4);
      ra[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ia + (y * PY + 2) * 4);
      rb[y * 2 + 0] = *reinterpret_cast<const f32x2*>(sInB + ib + (y * PY) * 4);
      rb[y * 2 + 1] = *reinterpret_cast<const f32x2*>(sInB + ib + (y * PY + 2) * 4);
    }
  };
  auto transform = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) T[j] = pk_fma(rb[j], qs2, ra[j]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      U[0 + h] = pk_sub(T[0 + h], T[4 + h]);
      U[2 + h] = pk_add(T[2 + h], T[4 + h]);
      U[4 + h] = pk_sub(T[4 + h], T[2 + h]);
      U[6 + h] = pk_sub(T[2 + h], T[6 + h]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A2[k * 2 + 0] = pk_bt01(U[k * 2], U[k * 2 + 1]);
      A2[k * 2 + 1] = pk_bt23(U[k * 2], U[k * 2 + 1]);
    }
  };

  const int nk4 = a.Cin >> 2;
  f32x4 bq[NB][4];
  const unsigned laneb = static_cast<unsigned>(lane) * 16u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * 256u);
  // packed weights of conv_wino.hip: [cout/32][xi_z][cin/4][cout/16 % 2][xi_y][cin % 4][cout % 16][xi_x]; this wave's 64 couts = slices 2cs, 2cs+1
  auto issue_b = [&](int nbb, int k4) {
    const int kl = k4 < nk4 ? k4 : 0;
    const unsigned sb = static_cast<unsigned>(((2 * cs + (nbb >> 1)) * 4 + xz) * nk4 + kl) * 8192u + (nbb & 1) * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[nbb][q] = buf_load16(wsrd, laneb + q * 1024u, sb);
  };

  f32x4 acc[NB][16];
  const int nchunk = a.Cin / CKW;
  Blk cur = decode(tb0);
  set_offs(cur);
  {
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, vol_bytes);
    f32x4 stg[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stg[it] = buf_load16(srd0, so[it], 0u);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) stage_store(it, 0, stg[it]);
  }
  __syncthreads();

  int pb = 0;
  float sink = 0.f;
  for (int it = 0; it < niter; ++it) {
    const int tn = tb0 + (it + 1 < niter ? it + 1 : it) * tstride;
    const Blk nxt = decode(tn);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUF * 4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) issue_b(nb, 0);

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = ((chunk + pb) & 1) * BUF * 4, bn = BUF * 4 - bo;
      const bool lastc = chunk + 1 == nchunk;
      if (lastc) set_offs(nxt);
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, vol_bytes);
      const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
      f32x4 stg[NLOAD];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        transform();
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 2) {
#pragma unroll
          for (int s = 0; s < NLOAD; ++s) stage_store(s, bn, stg[s]);
        }
        if (ks == 3) __syncthreads();
        raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[i >> 1][i & 1], bq[nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          issue_b(nb, chunk * 4 + ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks == 0) {
#pragma unroll
          for (int s = 0; s < NLOAD; ++s) stg[s] = buf_load16(ssrd, so[s], schunk);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // stand-in for the epilogue: fold the accumulators (keeps them live) -- NOT the inverse transform
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) sink += acc[nb][i][0] + acc[nb][i][1] + acc[nb][i][2] + acc[nb][i][3];
    __syncthreads();
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
  a.y[static_cast<long long>(blockIdx.x) * kT + tid] = sink;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4;
  const int D = 64, H = 96, W = 64, C = 128;
  const size_t nx = static_cast<size_t>(B) * D * H * W * C;
  float *x, *wp, *y;
  hipMalloc(&x, nx * 4); hipMalloc(&wp, (static_cast<size_t>(64) * C * C + 1024) * 4); hipMalloc(&y, 256 * kT * 4);
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<float>((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  for (size_t off = 0; off < nx; off += h.size()) hipMemcpy(x + off, h.data(), (nx - off < h.size() ? nx - off : h.size()) * 4, hipMemcpyHostToDevice);
  for (size_t off = 0; off < static_cast<size_t>(64) * C * C; off += h.size()) hipMemcpy(wp + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  Args a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.y = y;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = C; a.Cout = C;
  a.nbz = D / 2; a.nby = H / 8; a.nbx = W / 8; a.ntb = B * a.nbz * a.nby * a.nbx; a.ncs = C / 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(wino64_main_loop, dim3(256), dim3(kT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double fl = 2.0 * B * D * H * W * C * C * 27 / 3.375;
    printf("wino64 main loop B=%d: %.3f ms  executed %.1f TFLOP/s (%.3f of 157.3)  [%s]\n", B, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
