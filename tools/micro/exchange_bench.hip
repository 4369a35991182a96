// Micro-benchmark of the per-timestep all-gather between the 8 workgroups of a cluster (the exchange of
// the persistent LSTM forward kernel), in isolation: which store / load flavour, poll strategy and
// layout gives the shortest step.   build: hipcc -O3 --offload-arch=gfx950 exchange_bench.hip -o exchange_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kS = 0xFFFFFFFFu;
constexpr int NCL = 8, US = 32, H = 256, ROWS = 16;

struct Args {
  float* y;        // [ncl][ROWS][T][H]  (row stride T*H like the real layout) or [ncl][T][ROWS][H] (dense)
  float* junk;     // background traffic target
  unsigned long long* out;   // [4] cycles total, sweeps, checksum
  int T, dense, delay, bg, ring;
};

template <int STA, int LDA, int MODE, int PLA>
__global__ __launch_bounds__(256, 1) void k(Args a) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, b = blockIdx.x;
  const int x = b & 7, slot = b >> 3, per = (G >> 3) / NCL;
  const int cl = x * per + slot / NCL, j = slot % NCL;
  const long rowS = a.dense ? H : (long)a.T * H;       // floats between rows
  const long stepS = a.dense ? (long)ROWS * H : H;     // floats between steps
  float* base = a.y + (long)cl * ROWS * a.T * H;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, ROWS * a.T * H * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc((void*)(a.junk + (long)b * a.T * 4096), 0, a.T * 4096 * 4, 0x00020000);
  unsigned long long t0, t1, sweeps = 0, ta, tb, tc, twait = 0, tpull = 0;
  double chk = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int tt = 0; tt < a.T; ++tt) {
    const int t = a.ring ? tt % 3 : tt;
    // "compute"
    for (int d = 0; d < a.delay; ++d) __builtin_amdgcn_s_sleep(8);
    // publish own slice: waves 0,1 (128 lanes): 16 rows x 8 pieces
    if (wave < 2) {
      const int row = tid & 15, c4 = (tid >> 4);   // like the MFMA layout: 16 rows x 64B per instruction
      f32x4 v = {(float)(tt + 1), (float)j, (float)row, (float)c4};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)((row * rowS + t * stepS + j * US + c4 * 4) * 4), 0, STA);
      if (a.bg) {   // background: stash-like stores
        for (int i = 0; i < 5; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rj, (int)(((long)t * 4096 + i * 512 + tid * 4) * 4), 0, 0);
      }
    }
    u32x4 buf[4];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
    if (MODE == 0) {
      if (wave == 3 && lane < 2 * NCL && lane / 2 != j) {
        const int voff = (int)((15 * rowS + t * stepS + (lane / 2) * US + (lane & 1) * 16 + 12) * 4);
        while (true) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, LDA);
          if (v[0] != kS && v[1] != kS && v[2] != kS && v[3] != kS) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    } else if (MODE == 2) {
      __builtin_amdgcn_s_sleep(20);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
    unsigned pend = 0;
    for (int n = 0; n < 4; ++n) {
      const int e = tid + 256 * n, c4 = e % 64;
      if ((c4 * 4) / US != j) pend |= 1u << n;
    }
    bool first = true;
    while (pend) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n, c4 = e % 64, row = e / 64;
        if (first || ((pend >> n) & 1u))
          buf[n] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((row * rowS + t * stepS + c4 * 4) * 4), 0, PLA);
      }
      first = false;
      ++sweeps;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (buf[n][0] != kS && buf[n][1] != kS && buf[n][2] != kS && buf[n][3] != kS) pend &= ~(1u << n);
      if (pend && MODE != 0) __builtin_amdgcn_s_sleep(MODE == 1 ? 2 : 6);
    }
    if (a.ring && wave < 2 && tt >= 1) {
      const int row = tid & 15, c4 = (tid >> 4), ts = (tt - 1) % 3;
      u32x4 arm = {kS, kS, kS, kS};
      __builtin_amdgcn_raw_buffer_store_b128(arm, r, (int)((row * rowS + ts * stepS + j * US + c4 * 4) * 4), 0, STA);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc)::"memory");
    twait += tb - ta; tpull += tc - tb;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int e = tid + 256 * n, c4 = e % 64;
      if ((c4 * 4) / US != j) {
        f32x4 v = __builtin_bit_cast(f32x4, buf[n]);
        chk += v[0] + v[1];
        smem[(e * 4) % 4096] = v[2];
      }
    }
    __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (cl == 0 && j == 1 && tid == 5) {
    a.out[0] = t1 - t0;
    a.out[1] = sweeps;
    a.out[3] = twait; a.out[4] = tpull;
    a.out[2] = (unsigned long long)chk;
  }
}

template <int STA, int LDA, int MODE, int PLA>
void run(const char* name, Args a, int ncl) {
  const size_t ybytes = (size_t)ncl * ROWS * a.T * H * 4;
  float ms = 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<STA, LDA, MODE, PLA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(a.y, 0xFF, ybytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<STA, LDA, MODE, PLA>), dim3(ncl * NCL), dim3(256), 100 * 1024, 0, a);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); exit(1); }
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long o[8];
  hipMemcpy(o, a.out, sizeof(o), hipMemcpyDeviceToHost);
  // expected checksum for cluster 0 member 0 thread 5: sum over t, over its 4 pieces not from member 0 of (t+1) + member
  printf("%-32s ring=%d dense=%d delay=%d bg=%d: %7.1f cyc/step %5.2f us/step  wait %6.1f pull %6.1f sweeps %.2f\n", name, a.ring, a.dense, a.delay,
         a.bg, (double)o[0] / a.T, ms * 1e3 / a.T, (double)o[3] / a.T, (double)o[4] / a.T, (double)o[1] / a.T);
}

int main(int argc, char** argv) {
  const int T = 128, ncl = 32;
  Args a;
  a.T = T;
  hipMalloc(&a.y, (size_t)ncl * ROWS * T * H * 4);
  hipMalloc(&a.junk, (size_t)256 * T * 4096 * 4);
  hipMalloc(&a.out, 64);
  hipMemset(a.out, 0, 64);
  for (int ring = 0; ring < 2; ++ring)
  for (int delay = 0; delay <= 8; delay += 8) {
    a.dense = ring; a.bg = 1; a.delay = delay; a.ring = ring;
    run<16, 16, 0, 16>("st sc1   poll sc1 pull sc1", a, ncl);
    run<0, 16, 0, 16>("st plain poll sc1 pull sc1", a, ncl);
    run<0, 16, 0, 0>("st plain poll sc1 pull plain", a, ncl);
    run<0, 16, 0, 1>("st plain poll sc1 pull sc0", a, ncl);
    run<0, 16, 0, 2>("st plain poll sc1 pull nt", a, ncl);
    run<0, 16, 0, 17>("st plain poll sc1 pull sc0sc1", a, ncl);
    run<0, 16, 1, 16>("st plain direct sc1 sleep2", a, ncl);
    run<1, 16, 0, 16>("st sc0   poll sc1 pull sc1", a, ncl);
    run<2, 16, 0, 16>("st nt    poll sc1 pull sc1", a, ncl);
  }
  return 0;
}
