// Feasibility probe (timing only, NO epilogue: the accumulators are folded, not inverse-transformed): the main loop of a MIXED Winograd
// convolution F(4,3) along x, F(2,3) along y and z for the 3x3x3 SAME layers of GeneratorBE3 (reference: slim.conv3d behind ops.py:15-16,
// model.py:68) -- 4 x 4 x 6 = 96 transform points per 2 x 2 x 4 output tile: 6 instead of 8 matrix-core MACs per output voxel and
// (cin, cout) pair (-25 % vs conv_wino.hip's F(2,3)^3, 4.5x fewer than the direct form), for an input transform that costs the same
// per output voxel (the x stage has 12 instead of 4 operations per row, over twice the outputs).
// Shape B (this file): TWO waves per SIMD like conv_wino.hip -- 8 waves = 4 xi_z x 2 halves of the 32-channel cout slice, 16 tiles =
// 2 x 8 x 16 output voxels per workgroup, 16 output channels per wave: 24 MFMA 16x16x4 per k-step (96 accumulator registers).  The two
// cout halves would both run the whole input transform, so its z stage is moved into the STAGING threads (the LDS holds the four
// xi_z planes B^T_z d instead of the four raw planes: same size), leaving 36 packed-fp32 ops (y, x stages), 12 ds_read_b64 and 6 weight
// loads per wave and k-step.  Staging item = (halo row position, channel quad | pair) with all four planes: 8 loads, 12 packed ops,
// 24 ds_write_b32 per thread and chunk.
// Build: hipcc -O3 --offload-arch=gfx950 wino43b_probe.hip -o wino43b_probe ; run on the GPU box: ./wino43b_probe [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int CKW = 16;
constexpr int HZ = 4, HY = 10, HX = 18, HV = HZ * HY * HX;      // halo block 4 x 10 x 18 (outputs 2 x 8 x 16)
constexpr int PY = 24, PZ = HY * PY;                             // row pitch 24: the 16 tiles' ds_read_b64 hit 32 distinct bank pairs
constexpr int CP = HZ * PZ + 2;                                  // +2: the two cin%4 halves of a 32-lane group use the other bank pairs
constexpr int NPOS = HY * HX;                                     // 180 (y, x) halo positions, 720 (position, channel quad) items
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

template <int NOSTAGE>
__global__ __launch_bounds__(kT, 1) void wino43b_main_loop(const Args a) {
  __shared__ __attribute__((aligned(16))) float sIn[2 * BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xz = wave & 3, ch = wave >> 2;      // xi_z; 16-channel half of the cout slice
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
  // ---- staging plan: item A = (position tid >> 2 | 128 + ..., channel quad tid & 3) for tid < 512 covers positions 0..127;
  // the remaining 52 positions x 4 quads = 208 quad items are cut into 416 channel-PAIR items (threads 0..415): every thread loads
  // 4 planes x 16 B (item A) and, below 416, 4 planes x 8 B (item B)
  const int posA = tid >> 2, qA = tid & 3;                       // positions 0..127
  const bool hasB = tid < 416;
  const int posB = 128 + (tid >> 3), pB = tid & 7;               // positions 128..179, channel pair 0..7
  const int hyA = posA / HX, hxA = posA % HX, hyB = (hasB ? posB : 128) / HX, hxB = (hasB ? posB : 128) % HX;
  const int ldA = ((qA * 4) * CP + hyA * PY + hxA) * 4, ldB = ((pB * 2) * CP + hyB * PY + hxB) * 4;
  unsigned soA[4], soB[4];
  const unsigned vol_bytes = static_cast<unsigned>(a.D * a.H * a.W) * a.Cin * 4u;
  auto set_offs = [&](const Blk& bi) {
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) {
      const int gz = bi.z0 - 1 + hz;
      {
        const int gy = bi.y0 - 1 + hyA, gx = bi.x0 - 1 + hxA;
        const bool ok = static_cast<unsigned>(gz) < static_cast<unsigned>(a.D) && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) &&
                        static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
        soA[hz] = ok ? static_cast<unsigned>(bi.hoff + ((hz * a.H + hyA) * a.W + hxA) * a.Cin + qA * 4) * 4u : 0x80000000u;
      }
      {
        const int gy = bi.y0 - 1 + hyB, gx = bi.x0 - 1 + hxB;
        const bool ok = hasB && static_cast<unsigned>(gz) < static_cast<unsigned>(a.D) && static_cast<unsigned>(gy) < static_cast<unsigned>(a.H) &&
                        static_cast<unsigned>(gx) < static_cast<unsigned>(a.W);
        soB[hz] = ok ? static_cast<unsigned>(bi.hoff + ((hz * a.H + hyB) * a.W + hxB) * a.Cin + pB * 2) * 4u : 0x80000000u;
      }
    }
  };
  char* sInB = reinterpret_cast<char*>(sIn);
  f32x4 sA[4];
  f32x2 sB[4];
  auto stage_load = [&](__amdgpu_buffer_rsrc_t srd, unsigned chunkbytes) {
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) sA[hz] = buf_load16(srd, soA[hz], chunkbytes);
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) sB[hz] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srd, soB[hz], chunkbytes, 0));
  };
  // z stage of the input transform (B^T of F(2,3) over the four raw planes) + the writes of the four xi_z planes
  auto stage_store = [&](int bufbytes) {
    f32x2 lo[4], hi[4];
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) { lo[hz] = f32x2{sA[hz][0], sA[hz][1]}; hi[hz] = f32x2{sA[hz][2], sA[hz][3]}; }
    const f32x2 tl0 = pk_sub(lo[0], lo[2]), tl1 = pk_add(lo[1], lo[2]), tl2 = pk_sub(lo[2], lo[1]), tl3 = pk_sub(lo[1], lo[3]);
    const f32x2 th0 = pk_sub(hi[0], hi[2]), th1 = pk_add(hi[1], hi[2]), th2 = pk_sub(hi[2], hi[1]), th3 = pk_sub(hi[1], hi[3]);
    const f32x2 tb0_ = pk_sub(sB[0], sB[2]), tb1 = pk_add(sB[1], sB[2]), tb2 = pk_sub(sB[2], sB[1]), tb3 = pk_sub(sB[1], sB[3]);
    float* d = reinterpret_cast<float*>(sInB + (ldA + bufbytes));
    const f32x2 tl[4] = {tl0, tl1, tl2, tl3}, th[4] = {th0, th1, th2, th3}, tb[4] = {tb0_, tb1, tb2, tb3};
#pragma unroll
    for (int z = 0; z < 4; ++z) {
      d[z * PZ] = tl[z][0]; d[CP + z * PZ] = tl[z][1]; d[2 * CP + z * PZ] = th[z][0]; d[3 * CP + z * PZ] = th[z][1];
    }
    if (hasB) {
      float* e = reinterpret_cast<float*>(sInB + (ldB + bufbytes));
#pragma unroll
      for (int z = 0; z < 4; ++z) { e[z * PZ] = tb[z][0]; e[CP + z * PZ] = tb[z][1]; }
    }
  };

  const int offAb = (kq * CP + xz * PZ + (2 * ty) * PY + 4 * tx) * 4;
  f32x2 ra[12];              // this wave's xi_z plane: [row][x pair]
  f32x2 A2[12];              // A operands: per xi_y the pairs (o0,o5), (o1,o3), (o2,o4) of the x transform
  auto raw_read = [&](int idxbytes) {
    int ia = idxbytes + offAb;
    asm volatile("" : "+v"(ia));
    __builtin_assume((ia & 7) == 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra[r * 3 + p] = *reinterpret_cast<const f32x2*>(sInB + ia + (r * PY + 2 * p) * 4);
  };
  const f32x2 c41 = {-4.f, -1.f}, c12 = {1.f, 2.f}, cm12 = {-1.f, -2.f}, c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f};
  auto transform = [&]() {
    f32x2 U[12];
#pragma unroll
    for (int p = 0; p < 3; ++p) {                                                     // y: B^T of F(2,3)
      U[0 * 3 + p] = pk_sub(ra[0 * 3 + p], ra[2 * 3 + p]);
      U[1 * 3 + p] = pk_add(ra[1 * 3 + p], ra[2 * 3 + p]);
      U[2 * 3 + p] = pk_sub(ra[2 * 3 + p], ra[1 * 3 + p]);
      U[3 * 3 + p] = pk_sub(ra[1 * 3 + p], ra[3 * 3 + p]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                                     // x: B^T of F(4,3) on (d0,d1) (d2,d3) (d4,d5)
      const f32x2 P = U[r * 3], Q = U[r * 3 + 1], R = U[r * 3 + 2];
      A2[r * 3 + 0] = pk_fma(P, c4, pk_fma(Q, cm5, R));
      const f32x2 ac = pk_fma_ll(Q, c41, R);
      const f32x2 be = pk_fma_hh(P, c41, Q);
      A2[r * 3 + 1] = pk_fma(be, c12, ac);
      A2[r * 3 + 2] = pk_fma(be, cm12, ac);
    }
  };

  const int nk4 = a.Cin >> 2;
  f32x4 bq[6];
  const unsigned laneb = static_cast<unsigned>(lane) * 16u;
  const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.wp, static_cast<unsigned>(a.Cin) * a.Cout * 96u * 4u);
  // packed weights: [cout/32][xi_z][cin/4][cout/16 % 2][q = point/4][lane = (cin%4, cout%16)][point % 4]
  const unsigned wbase = static_cast<unsigned>((cs * 4 + xz) * nk4) * 12288u + static_cast<unsigned>(ch) * 6144u;
  auto issue_b = [&](int q0, int q1, int k4) {
    const int kl = k4 < nk4 ? k4 : k4 - nk4;
    const unsigned sb = wbase + static_cast<unsigned>(kl) * 12288u;
#pragma unroll
    for (int q = 0; q < 6; ++q) if (q >= q0 && q < q1) bq[q] = buf_load16(wsrd, laneb + q * 1024u, sb);
  };

  f32x4 acc[NPT];
  const int nchunk = a.Cin / CKW;
  Blk cur = decode(tb0);
  set_offs(cur);
  {
    const __amdgpu_buffer_rsrc_t srd0 = make_srd(cur.xb, vol_bytes);
    stage_load(srd0, 0u);
    stage_store(0);
  }
  __syncthreads();

  int pb = 0;
  float sink = 0.f;
  for (int it = 0; it < niter; ++it) {
    const int tn = tb0 + (it + 1 < niter ? it + 1 : it) * tstride;
    const Blk nxt = decode(tn);
#pragma unroll
    for (int i = 0; i < NPT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    raw_read(pb * BUF * 4);
    issue_b(0, 6, 0);

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int bo = ((chunk + pb) & 1) * BUF * 4, bn = BUF * 4 - bo;
      const bool lastc = chunk + 1 == nchunk;
      if (lastc) set_offs(nxt);
      const __amdgpu_buffer_rsrc_t ssrd = make_srd(lastc ? nxt.xb : cur.xb, vol_bytes);
      const unsigned schunk = static_cast<unsigned>(lastc ? 0 : chunk + 1) * (CKW * 4u);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        transform();
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 2 && !NOSTAGE) stage_store(bn);
        if (ks == 3) __syncthreads();
        raw_read(ks < 3 ? bo + (ks + 1) * 16 * CP : bn);
        __builtin_amdgcn_sched_barrier(0);
        // two halves of 12 MFMAs, the weights of the next k-step requested as each half's registers free up
#pragma unroll
        for (int hf2 = 0; hf2 < 2; ++hf2) {
#pragma unroll
          for (int i = hf2 * 12; i < hf2 * 12 + 12; ++i) {
            const int xy = i / 6, xx = i % 6;
            const int pr = xx == 0 || xx == 5 ? 0 : (xx == 1 || xx == 3 ? 1 : 2), hf = (xx == 5 || xx == 3 || xx == 4) ? 1 : 0;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[xy * 3 + pr][hf], bq[i >> 2][i & 3], acc[i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          issue_b(hf2 * 3, hf2 * 3 + 3, chunk * 4 + ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks == 0 && !NOSTAGE) stage_load(ssrd, schunk);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    __syncthreads();
    pb = (pb + nchunk) & 1;
    cur = nxt;
  }
  a.y[static_cast<long long>(blockIdx.x) * kT + tid] = sink;
}

template <int NOSTAGE>
static void run(const Args& a, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wino43b_main_loop<NOSTAGE>), dim3(256), dim3(kT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double alg = 2.0 * a.B * a.D * a.H * a.W * a.Cin * a.Cout * 27;
    const double fl = alg * 6.0 / 27.0;
    printf("wino43b main loop %-34s B=%d: %.3f ms  executed %.1f TFLOP/s (%.3f of 157.3)  direct-equivalent %.0f TFLOP/s  [%s]\n", name, a.B, ms,
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
  run<0>(a, "z-staged, two waves per SIMD");
  run<1>(a, "no staging (bound)");
  return 0;
}
