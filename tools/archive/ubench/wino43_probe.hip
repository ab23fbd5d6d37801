// Feasibility probe (timing only, NO epilogue: the accumulators are folded, not inverse-transformed): the main loop of a MIXED Winograd
// convolution F(4,3) along x, F(2,3) along y and z for the 3x3x3 SAME layers of GeneratorBE3 (reference: slim.conv3d behind ops.py:15-16,
// model.py:68) -- 4 x 4 x 6 = 96 transform points per 2 x 2 x 4 output tile: 6 instead of 8 matrix-core MACs per output voxel and
// (cin, cout) pair (-25 % vs conv_wino.hip's F(2,3)^3, 4.5x fewer than the direct form), for an input transform that costs the same
// per output voxel (the x stage has 12 instead of 4 operations per row, over twice the outputs).
// Shape: ONE wave per SIMD (4 waves = the 4 xi_z of one tile z-row), 16 tiles = 2 x 8 x 16 output voxels per workgroup, 32 output
// channels per wave: 48 MFMA 16x16x4 per k-step (192 accumulator registers), 48 packed-fp32 transform ops, 24 ds_read_b64, 12 weight
// loads; the 512-register budget of a lone wave pays for a weight prefetch NPRE k-steps deep, which also moves the first weight wait
// behind a chunk's HBM-latency staging loads NPRE + 1 k-steps away (vmcnt retires in order: conv_wino.hip has 1.5 k-steps there).
// Build: hipcc -O3 --offload-arch=gfx950 wino43_probe.hip -o wino43_probe ; run on the GPU box: ./wino43_probe [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 256;
constexpr int CKW = 16;
constexpr int HZ = 4, HY = 10, HX = 18, HV = HZ * HY * HX;      // halo block 4 x 10 x 18 (outputs 2 x 8 x 16)
constexpr int PY = 24, PZ = HY * PY;                             // row pitch 24: the 16 tiles' ds_read_b64 hit 32 distinct bank pairs
constexpr int CP = HZ * PZ + 2;                                  // +2: the two cin%4 halves of a 32-lane group use the other bank pairs
constexpr int NLOAD = (HV * 4 + kT - 1) / kT;                    // 12
constexpr int BUF = CKW * CP;
constexpr int NPT = 24;                                          // (xi_y, xi_x) points per xi_z

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
// (q.lo * c.lo + r.lo, q.lo * c.hi + r.lo): both halves from the LOW words of q and r
__device__ __forceinline__ f32x2 pk_fma_ll(f32x2 q, f32x2 c, f32x2 r) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(q), "v"(c), "v"(r));
  return d;
}
// (p.hi * c.lo + q.hi, p.hi * c.hi + q.hi): both halves from the HIGH words of p and q
__device__ __forceinline__ f32x2 pk_fma_hh(f32x2 p, f32x2 c, f32x2 q) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(p), "v"(c), "v"(q));
  return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int NPRE, int NOSTAGE, int EARLY = 0>
__global__ __launch_bounds__(kT, 1) void wino43_main_loop(const Args a) {
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
    bi.z0 = bz * 2; bi.y0 = by * 8; bi.x0 = bx * 16;
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
  const int abase = kq * CP + (2 * ty) * PY + 4 * tx;
  const int offAb = (abase + za * PZ) * 4, offBb = (abase + zb * PZ) * 4;
  f32x2 ra[12], rb[12];      // raw inputs [row][x pair] of planes za / zb
  f32x2 A2[12];              // A operands of a k-step: per xi_y the pairs (o0,o5), (o1,o3), (o2,o4) of the x transform
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb, ib = idxbytes + offBb;
    asm volatile("" : "+v"(ia), "+v"(ib));
    __builtin_assume((ia & 7) == 0);
    __builtin_assume((ib & 7) == 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        ra[r * 3 + p] = *reinterpret_cast<const f32x2*>(sInB + ia + (r * PY + 2 * p) * 4);
        rb[r * 3 + p] = *reinterpret_cast<const f32x2*>(sInB + ib + (r * PY + 2 * p) * 4);
      }
  };
  const f32x2 c41 = {-4.f, -1.f}, c12 = {1.f, 2.f}, cm12 = {-1.f, -2.f}, c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f};
  auto transform = [&]() {
    f32x2 T[12], U[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) T[j] = pk_fma(rb[j], qs2, ra[j]);                    // z
#pragma unroll
    for (int p = 0; p < 3; ++p) {                                                     // y: B^T of F(2,3)
      U[0 * 3 + p] = pk_sub(T[0 * 3 + p], T[2 * 3 + p]);
      U[1 * 3 + p] = pk_add(T[1 * 3 + p], T[2 * 3 + p]);
      U[2 * 3 + p] = pk_sub(T[2 * 3 + p], T[1 * 3 + p]);
      U[3 * 3 + p] = pk_sub(T[1 * 3 + p], T[3 * 3 + p]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                                     // x: B^T of F(4,3) on (d0,d1) (d2,d3) (d4,d5)
      const f32x2 P = U[r * 3], Q = U[r * 3 + 1], R = U[r * 3 + 2];
      A2[r * 3 + 0] = pk_fma(P, c4, pk_fma(Q, cm5, R));                               // (4d0 - 5d2 + d4, 4d1 - 5d3 + d5)
      const f32x2 ac = pk_fma_ll(Q, c41, R);                                          // (d4 - 4d2, d4 - d2)
      const f32x2 be = pk_fma_hh(P, c41, Q);                                          // (d3 - 4d1, d3 - d1)
      A2[r * 3 + 1] = pk_fma(be, c12, ac);                                            // (o1, o3)
      A2[r * 3 + 2] = pk_fma(be, cm12, ac);                                           // (o2, o4)
    }
  };

  const int nk4 = a.Cin >> 2;
  f32x4 bq[NPRE][2][6];
  const unsigned laneb = static_cast<unsigned>(lane) * 16u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * 96u * 4u);
  // packed weights: [cout/32][xi_z][cin/4][cout/16 % 2][q = point/4][lane = (cin%4, cout%16)][point % 4]
  const unsigned wbase = static_cast<unsigned>((cs * 4 + xz) * nk4) * 12288u;
  auto issue_b = [&](int slot, int nb, int k4) {
    const int kl = k4 < nk4 ? k4 : k4 - nk4;
    const unsigned sb = wbase + static_cast<unsigned>(kl) * 12288u + nb * 6144u;
#pragma unroll
    for (int q = 0; q < 6; ++q) bq[slot][nb][q] = buf_load16(wsrd, laneb + q * 1024u, sb);
  };

  f32x4 acc[2][NPT];
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
  f32x4 stgE[NLOAD];      // EARLY: the staging registers are live across chunks (loads issued FOUR k-steps before their LDS writes)
  if (EARLY) {
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, vol_bytes);
#pragma unroll
    for (int s = 0; s < NLOAD; ++s) stgE[s] = buf_load16(srd0, so[s], CKW * 4u);      // chunk 1 of the first block
  }
  for (int it = 0; it < niter; ++it) {
    const int tn = tb0 + (it + 1 < niter ? it + 1 : it) * tstride;
    const Blk nxt = decode(tn);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < NPT; ++i) acc[nb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUF * 4);
#pragma unroll
    for (int s = 0; s < NPRE; ++s) { issue_b(s, 0, s); issue_b(s, 1, s); }

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = ((chunk + pb) & 1) * BUF * 4, bn = BUF * 4 - bo;
      const bool lastc = EARLY ? chunk + 2 >= nchunk : chunk + 1 == nchunk;
      if (EARLY ? chunk + 2 == nchunk : lastc) set_offs(nxt);
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, vol_bytes);
      const unsigned schunk = static_cast<unsigned>(EARLY ? (chunk + 2) % nchunk : (lastc ? 0 : chunk + 1)) * (CKW * 4u);
      f32x4 stg[NLOAD];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        constexpr int dummy = 0; (void)dummy;
        const int slot = ks % NPRE;
        transform();
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 2 && !NOSTAGE) {
#pragma unroll
          for (int s = 0; s < NLOAD; ++s) stage_store(s, bn, EARLY ? stgE[s] : stg[s]);
        }
        if (ks == 3) __syncthreads();
        raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int i = 0; i < NPT; ++i) {
            // point i = xi_y * 6 + xi_x; the x transform leaves xi_x in the pairs (0,5) (1,3) (2,4)
            const int xy = i / 6, xx = i % 6;
            const int pr = xx == 0 || xx == 5 ? 0 : (xx == 1 || xx == 3 ? 1 : 2), hf = (xx == 5 || xx == 3 || xx == 4) ? 1 : 0;
            acc[nb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[xy * 3 + pr][hf], bq[slot][nb][i >> 2][i & 3], acc[nb][i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          issue_b(slot, nb, chunk * 4 + ks + NPRE);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks == 0 && !NOSTAGE && !EARLY) {
#pragma unroll
          for (int s = 0; s < NLOAD; ++s) stg[s] = buf_load16(ssrd, so[s], schunk);
        }
        if (ks == 2 && !NOSTAGE && EARLY) {      // the registers were just written to LDS: reload them for the chunk after next
#pragma unroll
          for (int s = 0; s < NLOAD; ++s) stgE[s] = buf_load16(ssrd, so[s], schunk);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // stand-in for the epilogue: fold the accumulators (keeps them live) -- NOT the inverse transform
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < NPT; ++i) sink += acc[nb][i][0] + acc[nb][i][1] + acc[nb][i][2] + acc[nb][i][3];
    __syncthreads();
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
  a.y[static_cast<long long>(blockIdx.x) * kT + tid] = sink;
}

template <int NPRE, int NOSTAGE, int EARLY = 0>
static void run(const Args& a, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wino43_main_loop<NPRE, NOSTAGE, EARLY>), dim3(256), dim3(kT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double alg = 2.0 * a.B * a.D * a.H * a.W * a.Cin * a.Cout * 27;
    const double fl = alg * 6.0 / 27.0;
    printf("wino43 main loop %-34s B=%d: %.3f ms  executed %.1f TFLOP/s (%.3f of 157.3)  direct-equivalent %.0f TFLOP/s  [%s]\n", name, a.B, ms,
           fl / ms / 1e9, fl / ms / 1e9 / 157.3, alg / ms / 1e9, hipGetErrorString(hipGetLastError()));
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4;
  const int D = 64, H = 96, W = 64, C = 128;
  const size_t nx = static_cast<size_t>(B) * D * H * W * C;
  const size_t nw = static_cast<size_t>(96) * C * C + 4096;
  float *x, *wp, *y;
  hipMalloc(&x, nx * 4); hipMalloc(&wp, nw * 4); hipMalloc(&y, 256 * kT * 4);
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<float>((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  for (size_t off = 0; off < nx; off += h.size()) hipMemcpy(x + off, h.data(), (nx - off < h.size() ? nx - off : h.size()) * 4, hipMemcpyHostToDevice);
  for (size_t off = 0; off < nw; off += h.size()) hipMemcpy(wp + off, h.data(), (nw - off < h.size() ? nw - off : h.size()) * 4, hipMemcpyHostToDevice);
  Args a;
  a.x = x; a.wp = reinterpret_cast<const f32x4*>(wp); a.y = y;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = C; a.Cout = C;
  a.nbz = D / 2; a.nby = H / 8; a.nbx = W / 16; a.ntb = B * a.nbz * a.nby * a.nbx; a.ncs = C / 32;
  run<2, 0, 1>(a, "2 ahead, staging 4 k-steps ahead");
  run<1, 0, 1>(a, "1 ahead, staging 4 k-steps ahead");
  run<1, 0>(a, "weights 1 k-step ahead");
  run<2, 0>(a, "weights 2 k-steps ahead");
  run<2, 1>(a, "2 ahead, no staging (bound)");
  run<1, 1>(a, "1 ahead, no staging (bound)");
  return 0;
}
