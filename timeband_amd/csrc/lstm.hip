// K2 / K3: persistent LSTM recurrent forward and fused backward-through-time for gfx950.
//
// Replaces aten::lstm / its autograd (cuDNN or oneDNN under torch) as called by the reference
// at source/utils/lstm_layer.py:31-33,44-46 (generator lstm0/1/2) and :75,83 (discriminator):
// torch.nn.LSTM semantics, gate order i,f,g,o, two biases, batch_first, zero initial state.
//
// Decomposition (one layer):
//   forward : P = x.W_ih^T + b_ih + b_hh        (K1 GEMM, all timesteps at once)
//             persistent recurrent kernel over t (this file): a = P_t + h_{t-1}.W_hh^T,
//             fused sigmoid/tanh, c/h update; activated gates overwrite P in place (BPTT stash)
//   backward: persistent BPTT kernel over t (this file): dgates_t overwrite the stash in place,
//             dh_{t-1} = dgates_t.W_hh; bias grads accumulated in registers
//             then K1 GEMMs: dx = dG.W_ih, dW_ih = dG^T.x, dW_hh = dG^T.h_prev
//
// Persistent kernel layout.  The batch is cut into tiles of 16*MT rows; a CLUSTER of NCL = H/US
// workgroups (one per CU, all co-resident: grid <= #CUs) owns one tile.  Member j owns hidden
// units [j*US, (j+1)*US): in forward the 4*US rows of W_hh that produce its gates, held in
// VGPRs for the whole sequence as MFMA A-operand fragments, split into bf16 hi + lo (common.h:
// three v_mfma_f32_16x16x32_bf16 per product, fp32-grade); c stays in registers too.  The products
// are computed TRANSPOSED (gates^T = W_hh . h^T), so a lane's four accumulator registers are four
// consecutive hidden units of one batch row and every global access of the step is 16 bytes wide.
//
// Exchange.  Each timestep every member needs the whole h_{t-1} tile (forward: all-gather) or the
// sum of all members' partials of dh_{t-1} (backward: reduce-scatter).  Both go through a small ring
// of three slots per cluster in the workspace (hot in L2, one TLB page) that is armed with an
// all-ones sentinel: every word of a slot is written once per turn, so the data is its own ready
// flag -- no counters, no store drain.  Forward: the member's finishing waves store their h slice to
// the ring (and to y for the next layer), the other waves -- which have no memory traffic of their
// own queued -- poll one piece per storing wave and then pull the other members' slices straight
// into the LDS h tile; the producer re-arms its slice two steps later (when every consumer has
// provably read it) and waits for that store before it publishes again.  Backward: the consumer
// re-arms what it has summed.  At start-up the members exchange the id of the XCD they run on
// (cluster_hello): when they share one -- the round-robin placement the block->cluster map aims
// for -- they share an L2 and the exchange uses PLAIN stores that stay there (measured: hand-off
// 0.6 us instead of 1.8 us, tools/micro/exchange_bench.hip); otherwise it uses write-through
// (sc1) stores, valid under any placement.  Loads of exchanged data are always L1-bypassing (sc1).
// All spins are bounded (2 s) and report through the caller's err word.  NCL == 1 (H <= 64) needs
// no global exchange at all.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <random>

int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream, int accumulate);
int tb_gemm_launch2(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                    long ldc, const float* bias, const float* bias2, int M, int N, int K, int act, int b_shift,
                    int b_period, float* ws, size_t ws_bytes, hipStream_t stream, int accumulate);

#ifdef TB_STAMPS
// Diagnostic build only (tools/lstm_stamps.py): per-segment cycle sums of one workgroup's step loop.
__device__ unsigned long long tb_stamps[2][4][12];
#define STAMP_DECL unsigned long long stamp_prev, stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory")
#define STAMP(i)                                                                          \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long _t;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");            \
    stamp_acc[i] += _t - stamp_prev;                                                      \
    stamp_prev = _t;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define STAMP_FLUSH(k, cl, j)                                                             \
  do {                                                                                    \
    if ((cl) == 0 && (j) == 0 && (threadIdx.x & 63) == 0)                                 \
      for (int _i = 0; _i < 12; ++_i) tb_stamps[k][threadIdx.x >> 6][_i] = stamp_acc[_i]; \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(k, cl, j)
#endif

// ---- scratch of the entry point that allocates nothing (tb_lstm_fwd_static) ------------------------
// The exchange area of the forward kernel (handshake tags + h ring) for callers that cannot bring a
// workspace: a fixed array in the code object.  Each GPU the library runs on holds its own copy at its own
// address until the process ends; the name below is only a host-side handle -- the device address comes from
// hipGetSymbolAddress (static_scratch(), once per device) and is what every launch is given.
constexpr size_t kStaticScratchBytes = 8u << 20;   // largest forward exchange area is 6.3 MB (H=128, 4 tiles x 64 clusters)
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// device address of `scratch` on the CURRENT device (looked up once per device)
int static_scratch(void** out) {
  int dev = -1;
  TB_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return tb_set_error(TB_ERR_ARG, "static scratch: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    TB_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return tb_set_error(TB_ERR_HIP, "static scratch: hipGetSymbolAddress returned null");
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return TB_OK;
}

#ifndef TB_POLL_ABOVE
#define TB_POLL_ABOVE 8      // clusters of more members poll one piece per producer before they pull (A/B knob)
#endif
constexpr unsigned kSentinel = 0xFFFFFFFFu;   // all-ones: a NaN pattern no computed h equals
constexpr int kBufFlags = 0x00020000;         // raw buffer descriptor word 3 (32-bit raw, bounds-checked)
constexpr int kSC1 = 16;                      // aux: sc1 = write-through store / L1-bypassing load

struct FwdArgs {
  float* P;            // [B][T][4H] in: pre-activations from the projection (XK == 0); out: activated gates
  const float* x;      // [B][T][in_dim]   } fused input projection (XK > 0): the kernel computes
  const float* Wih;    // [4H][in_dim]     } x_t . W_ih^T + b_ih + b_hh itself and P is output only
  const float* b_ih;   // [4H] or nullptr  }
  const float* b_hh;   // [4H] or nullptr  }
  int in_dim;
  const float* Whh;    // [4H][H]
  float* y;            // [B][T][H]
  float* cst;          // [B][T][H] cell stash (train) or nullptr
  float* ring;         // [nclusters][3][MT*16][H] exchange ring (NCL > 1) of {bf16 hi | bf16 lo} words; every member arms its own slices at start-up
  unsigned long long* hello;   // [nclusters][NCL] start-up handshake tags
  unsigned long long epoch;    // this launch's handshake epoch
  int* err;            // [4] sticky error words
  int B, T;
  long p_row;          // floats between the pre-activation rows of consecutive batch rows (0: T * 4H, dense); inference over
                       // overlapping windows (tb_lstm_fwd_windows) sets stride * 4H: P is then one row per SERIES row
  int b_begin;         // first batch row of this launch
  int rows;            // batch rows per cluster: 16 * MT, or fewer valid rows of a 16-row tile (see make_plan)
  int nclusters;
  int train;
  unsigned long long timeout_ticks;
};

// cluster/member decode: keep the members of a cluster on blocks with equal (blockIdx % 8), i.e.
// on one XCD under the observed round-robin placement (speed only; correctness never depends on it:
// cluster_hello() checks the placement at run time)
__device__ __forceinline__ void cluster_decode(int NCL, int& cl, int& j) {
  const int G = gridDim.x, b = blockIdx.x;
  if ((G & 7) == 0 && ((G >> 3) % NCL) == 0) {
    const int x = b & 7, slot = b >> 3, per = (G >> 3) / NCL;
    cl = x * per + slot / NCL;
    j = slot % NCL;
  } else {
    cl = b / NCL;
    j = b % NCL;
  }
}

// Start-up handshake of a cluster.  Every member publishes a tag {launch epoch | id of the XCD it runs on}
// with a write-through (sc1) store and reads the others' (sc1 loads work across XCDs) until they carry THIS
// launch's epoch.  The epoch is unique per launch (a random 60-bit base per process plus a counter), so
// whatever the handshake words held before -- an earlier launch's tags, another layout's data, fresh
// memory -- cannot be mistaken for it, and nothing has to be cleared before a launch.
// A member sends its tag only after the stores that ARM ITS OWN PART of the exchange area have been
// acknowledged (callers: sc1 stores, then vmcnt(0), then this function, which starts with a barrier), so a
// member that has seen every tag knows the whole area is armed.
// Returns 1 if ALL members share one XCD -- then they share one L2, and the per-step exchange may use
// plain stores (which stay in that L2: a third of the hand-off latency of write-through stores) --, 0 if
// not (the exchange then uses write-through stores, valid under any placement), -1 on timeout.
// "all equal" is symmetric, so every member takes the same decision.
__device__ __forceinline__ int cluster_hello(unsigned long long* hello, unsigned long long epoch, int NCL, int j,
                                             unsigned long long timeout, int* err, int* s_word) {
  __syncthreads();   // every thread's arming stores are acknowledged
  unsigned myx;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myx));
  myx &= 0xFu;
  if (threadIdx.x == 0) tb_store_agent_u64(hello + j, (epoch << 4) | myx);
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned v = myx;
    bool ok = true;
    for (int m = lane; m < NCL; m += 64) {
      const unsigned long long t0 = wall_clock64();
      unsigned it = 0;
      unsigned long long w;
      while (((w = tb_load_agent_u64(hello + m)) >> 4) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if ((++it & 1023u) == 0 && wall_clock64() - t0 > timeout) {
          ok = false;
          break;
        }
      }
      if ((unsigned)(w & 0xFu) != myx) v = 0xFFFFFFF0u;
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
    const bool all_eq = __builtin_amdgcn_ballot_w64(v != myx) == 0ull;
    if (lane == 0) {
      if (!all_ok) atomicExch(err, 1);
      *s_word = all_ok ? (all_eq ? 1 : 0) : -1;
    }
  }
  __syncthreads();
  return *s_word;
}

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

template <int AUX>
__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, long float_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void stg4(__amdgpu_buffer_rsrc_t r, long float_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(float_off * 4), 0, AUX);
}
// N = 4, 2 or 1 consecutive floats per lane (the first N elements of an f32x4)
template <int N, int AUX>
__device__ __forceinline__ f32x4 ldgn(__amdgpu_buffer_rsrc_t r, long float_off) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (N == 4) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, AUX));
  else if (N == 2) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(float_off * 4), 0, AUX);
    v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]);
  } else v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(float_off * 4), 0, AUX));
  return v;
}
template <int N, int AUX>
__device__ __forceinline__ void stgn(__amdgpu_buffer_rsrc_t r, long float_off, f32x4 v) {
  if (N == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(float_off * 4), 0, AUX);
  else if (N == 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(v[0]), __float_as_uint(v[1])}, r, (int)(float_off * 4), 0, AUX);
  else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), r, (int)(float_off * 4), 0, AUX);
}
template <int N>
__device__ __forceinline__ void stxn(bool same_l2, __amdgpu_buffer_rsrc_t r, long float_off, f32x4 v) {
  if (same_l2) stgn<N, 0>(r, float_off, v);
  else stgn<N, kSC1>(r, float_off, v);
}
// exchange store: plain when the cluster shares one L2, write-through otherwise
__device__ __forceinline__ void stx4(bool same_l2, __amdgpu_buffer_rsrc_t r, long float_off, f32x4 v) {
  if (same_l2) stg4<0>(r, float_off, v);
  else stg4<kSC1>(r, float_off, v);
}
template <int N>
__device__ __forceinline__ bool armedn(f32x4 v) {   // any of the first N words still the sentinel
  bool r = __float_as_uint(v[0]) == kSentinel;
  if (N >= 2) r = r || __float_as_uint(v[1]) == kSentinel;
  if (N >= 4) r = r || __float_as_uint(v[2]) == kSentinel || __float_as_uint(v[3]) == kSentinel;
  return r;
}
__device__ __forceinline__ bool armed(u32x4 v) {
  return v[0] == kSentinel || v[1] == kSentinel || v[2] == kSentinel || v[3] == kSentinel;
}

// XK > 0: the input projection is fused (in_dim <= XK, XK = 32 or 64): the member's W_ih rows sit in LDS as split
// bf16 A-fragments, x_t tiles are fetched two steps ahead into a double-buffered LDS tile, and each step's MFMA block
// gets XK / 32 more k-steps -- the projection GEMM, its [B][T][4H] output and the kernel's read of it are gone.  Whether
// a layer is fused depends on (H, in_dim) only, never on the batch: every variant that can serve a layer does the same
// arithmetic, so a window's result does not depend on the launch it is computed in.
template <int H, int US, int KS, int MT, int XK, int CMP = 1>
__global__ __launch_bounds__(256, 1) void lstm_fwd_kernel(FwdArgs a) {
  constexpr int NUG = US / 16;          // 16-unit groups per member
  constexpr int NCL = H / US;           // members per cluster
  constexpr int HK = H < 32 ? 32 : H;   // k extent, padded to one MFMA k-step
  constexpr int KPER = HK / KS;         // k range per wave
  constexpr int NKS = KPER / 32;        // 32-deep MFMA k-steps per wave
  constexpr int LDH = HK + 8;           // LDS row stride of the h tile (bf16 elements; 16-byte pad)
  constexpr int NACT = NUG * KS;        // waves doing MFMA work
  // Every active wave finishes part of its unit group: the KS waves that split the k range of a unit
  // group each take UPW = 4 / KS of a lane's 4 units (they swap the matching parts of their partial
  // sums through LDS), so gate math, publish and pull are spread over all the waves of the workgroup.
  constexpr int UPW = 4 / KS;
  // CMP > 1 (single-workgroup fused layers whose 16-row tiles hold only 16 / CMP valid rows, see make_plan): the
  // elementwise part of a step is COMPACTED -- the MFMA leaves a lane (col, q) 4 units of batch row col, of which only
  // col < 16 / CMP exist; lane col then finishes units sub*UE .. of row col % (16 / CMP), sub = col / (16 / CMP), taking
  // the sums from lane col % (16 / CMP) of its 16-lane row by DPP.  Every lane works on a unit that exists: a quarter
  // (CMP = 4) of the exp / rcp work per step, which is what bounds these layers.  Same operations per (row, unit),
  // same bits.
  constexpr int UE = UPW / CMP;
  static_assert(CMP == 1 || (XK > 0 && H / US == 1 && MT == 1 && KS == 1), "compaction: fused single-workgroup one-tile layers only");
  // floats of k-split scratch: [ug][dst][src != dst][lane][4 gates][UPW] (a wave keeps its own part in registers, so
  // there is no [dst][dst] slot: KS - 1 sources per destination, indexed (src - dst - 1) mod KS)
  constexpr int RED = KS > 1 ? NUG * KS * (KS - 1) * 256 * UPW : 0;
  // exchange (NCL > 1): wave w pulls the slices of producers m = w, w + 4, ...
  constexpr int NPW = 4;
  constexpr int NPP = (NCL + NPW - 1) / NPW;            // producers per wave
  constexpr int LPP = 16 * (US / 4) / 64;               // 16-byte loads per lane per producer (one 16-row tile)
  constexpr int NLD = NPP * LPP;                        // loads per lane per step
  constexpr int RS = MT * 16 * H;                       // floats per ring slot
  // STAGE (one tile per cluster, H = 128 / 256): the pre-activations come in and y / the stash go out
  // through LDS.  A lane finishes 2 units of one row, so its own loads and stores are 8 bytes at 16 rows x 4
  // places per instruction (16 cache lines each; issuing one costs the wave ~90 cycles, 11 of them per step).
  // Staged, the workgroup moves whole 128-byte runs: 2 loads + 3 stores of 16 bytes per lane and step, each
  // instruction 8 full lines.  pbuf is double-buffered (step t+1 lands while step t is read); sbuf is
  // written at the end of a step and flushed at the top of the next one, behind the barrier between them.
  constexpr bool STAGE = MT == 1 && NCL > 1 && KS == 2 && US == 32;
  constexpr int PROW = 4 * US + 8, SROW = 6 * US + 8;   // padded LDS rows: [4 gates][US], [4 gates, c, y][US]
  // EMU2: the 4-member layout of hidden 256 (64 units per member, the whole k range in one wave) adds up in the order
  // of the 8-member layout's two k-part waves -- two accumulator sets, one per k half, the x part dealt between them,
  // joined as (own half + pre-activation) + other half with "own" by the unit's position -- so a window's result
  // stays bit-identical whichever layout its batch gets.
  constexpr bool EMU2 = H == 256 && US == 64 && KS == 1;
  constexpr bool FUSE = XK > 0;
  constexpr int NXS = XK / 32;                          // x k-steps
  constexpr int LDX = XK + 8;                           // LDS row stride of the W_ih / x tiles (bf16 elements)
  constexpr int NXT = 16 * (XK / 4);                    // threads that move one 16-row x tile (16 bytes each)
  constexpr int PBUF = (STAGE && !FUSE) ? 2 * 16 * PROW : 0, SBUF = STAGE ? 16 * SROW : 0;
  static_assert(!FUSE || ((XK == 32 || XK == 64) && NXT <= 256 && (KS == 1 || KS == 2)), "bad fused-projection config");
  static_assert(NACT <= 4 && KPER % 32 == 0 && H % US == 0 && (KS == 1 || KS == 2 || KS == 4), "bad config");
  static_assert(NCL == 1 || (NACT == 4 && (16 * (US / 4)) % 64 == 0 && NLD <= 32 && NPP * NUG <= 64), "bad exchange config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  // h_{t-1} tile as split bf16 (hi, lo): [MT*16 rows][LDH], one ds_read_b128 per MFMA B operand
  unsigned short* hiL = reinterpret_cast<unsigned short*>(smem);
  unsigned short* loL = hiL + MT * 16 * LDH;
  float* red = smem + MT * 16 * LDH;                   // [NUG][KS dst][KS-1 src][64 lanes][4 gates][UPW]
  int* s_dead = reinterpret_cast<int*>(red + RED);
  int* s_word = s_dead + 1;
  float* pbuf = red + RED + 4;                         // [2][16][PROW]   (STAGE, not fused)
  float* sbuf = pbuf + PBUF;                           // [16][SROW]      (STAGE)
  unsigned short* wiH = reinterpret_cast<unsigned short*>(sbuf + SBUF);   // [4 gates][US][LDX] W_ih rows, bf16 hi  (FUSE)
  unsigned short* wiL = wiH + 4 * US * LDX;                               //                               bf16 lo
  unsigned short* xH = wiL + 4 * US * LDX;                                // [2][MT*16][LDX] x_t tiles (t & 1), bf16 hi
  unsigned short* xL = xH + 2 * MT * 16 * LDX;
  float* bsum = reinterpret_cast<float*>(xL + 2 * MT * 16 * LDX);        // [4 gates][US] b_ih + b_hh of the member's units

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform branches and addresses
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(NCL, cl, j);
  const bool active = wave < NACT;
  const int ug = active ? wave % NUG : 0, kp = active ? wave / NUG : 0;
  const int uw = j * US + ug * 16 + col;               // W_hh row (per gate) this lane feeds to the MFMA
  const int u0m = j * US + ug * 16 + q * 4 + kp * UPW;  // first of the UPW hidden units whose sums the MFMA leaves in this lane
  const int sub = CMP > 1 ? col / (16 / CMP) : 0;
  const int rowl = CMP > 1 ? col % (16 / CMP) : col;   // batch row (of the tile) this lane finishes
  const int u0 = u0m + sub * UE;                       // first of the UE hidden units this lane finishes
  const int b0 = a.b_begin + cl * a.rows;
  const int T = a.T;
  const long rowP = a.p_row ? a.p_row : (long)T * 4 * H, rowY = (long)T * H;
  if (tid == 0) *s_dead = 0;
  for (int i = tid; i < MT * 16 * LDH; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;   // hi and lo, pad included
  if (NCL > 1) {
    // arm this member's slices of the three ring slots (write-through: the members may sit on different
    // XCDs); acknowledged before the handshake tag goes out, see cluster_hello()
    float* rg = a.ring + (size_t)cl * 3 * RS;
    const f32x4 arm = __builtin_bit_cast(f32x4, (u32x4){kSentinel, kSentinel, kSentinel, kSentinel});
    constexpr int PER = MT * 16 * (US / 4);
    const __amdgpu_buffer_rsrc_t rA_ = __builtin_amdgcn_make_buffer_rsrc((void*)rg, 0, 3 * RS * 4, kBufFlags);
    for (int i = tid; i < 3 * PER; i += 256) {
      const int slot = i / PER, r = i % PER, row = r / (US / 4), c4 = r % (US / 4);
      stg4<kSC1>(rA_, (long)slot * RS + (long)row * H + j * US + c4 * 4, arm);
    }
  }
  // descriptors over this cluster's rows: loads of rows >= B return 0, stores to them are dropped
  int vrows = a.B - b0;
  if (vrows > a.rows) vrows = a.rows;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(a.P + (long)b0 * rowP), 0, vrows > 0 ? (int)(((vrows - 1) * rowP + (long)T * 4 * H) * 4) : 0, kBufFlags);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)((a.train ? a.cst : a.y) + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(NCL > 1 ? a.ring + (size_t)cl * 3 * RS : a.y), 0, NCL > 1 ? 3 * RS * 4 : 0, kBufFlags);
  // fused projection: this cluster's rows of x; every thread of the first NXT moves 16 bytes of a 16-row tile, the
  // rest (and columns >= in_dim, steps >= T) ask for an out-of-range offset, which returns 0 without touching memory --
  // so EVERY wave issues exactly one load per tile and step (the vmcnt guard before the publish counts on it)
  const int in = FUSE ? a.in_dim : 0;
  const long rowX = (long)T * in;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? a.x + (long)b0 * rowX : a.y), 0, FUSE ? (int)(vrows * rowX * 4) : 0, kBufFlags);
  const int xr = tid / (FUSE ? XK / 4 : 1), xc = (tid % (FUSE ? XK / 4 : 1)) * 4;
  const bool xmine = FUSE && tid < NXT && xc < in;
  auto xload = [&](int tt, int mt) -> f32x4 {
    const long off = (xmine && tt < T) ? ((long)(mt * 16 + xr) * rowX + (long)tt * in + xc) : 0x1FFFFFFFl;
    return ldg4<0>(rX, off);
  };
  auto xstore = [&](int tt, int mt, f32x4 v) {
    if (tid < NXT) {
      u32x2 sh, sl;
      tb_split4(v, sh, sl);
      const int o = ((tt & 1) * MT * 16 + mt * 16 + xr) * LDX + xc;
      *reinterpret_cast<u32x2*>(xH + o) = sh;
      *reinterpret_cast<u32x2*>(xL + o) = sl;
    }
  };

  // W_hh fragments (A operand of v_mfma_f32_16x16x32_bf16), split into bf16 hi/lo, resident for the
  // whole sequence: A[m][k] = W_hh[g*H + uw][k], k-step s of this lane covers k = kp*KPER + s*32 + q*8 .. +7
  bf16x8 whi[4][NKS], wlo[4][NKS];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int k0 = kp * KPER + s * 32 + q * 8;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (active && k0 < H) {
        const float* wp = a.Whh + (long)(g * H + uw) * H + k0;
        v0 = *reinterpret_cast<const f32x4*>(wp);
        v1 = *reinterpret_cast<const f32x4*>(wp + 4);
      }
      tb_split8(v0, v1, whi[g][s], wlo[g][s]);
      // 4-member layout: 256 registers of fragments do not fit the 256 architectural VGPRs beside the working set.  Left
      // to the allocator, half of them are spilled to accumulation registers and copied back in front of every MFMA
      // (4 v_accvgpr_read per use, all through one temporary); pinned there as values of the AGPR class, the MFMAs
      // read them in place.
      if (EMU2 || (H == 512 && US == 32)) asm volatile("" : "+a"(wlo[g][s]));   // the 16-member layout of hidden 512 holds the same 256 registers
    }
  // W_ih fragments: in LDS for the clustered variants (no registers to spare: two of their launches must fit a CU),
  // in registers like W_hh for single-workgroup layers (re-reading them every step would put 40-80 KB per step
  // through the CU's LDS port: +0.3 us on a 1.1 us step)
  constexpr bool WREG = FUSE && NCL == 1;
  bf16x8 wxh[WREG ? 4 : 1][WREG ? NXS : 1], wxl[WREG ? 4 : 1][WREG ? NXS : 1];
  if (WREG) {
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wih, 0, 4 * H * in * 4, kBufFlags);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int sx = 0; sx < NXS; ++sx) {
        const int k0 = sx * 32 + q * 8;
        const long o = (long)(g * H + uw) * in + k0;
        const f32x4 v0 = ldg4<0>(rW, (active && k0 < in) ? o : 0x1FFFFFFFl);
        const f32x4 v1 = ldg4<0>(rW, (active && k0 + 4 < in) ? o + 4 : 0x1FFFFFFFl);
        tb_split8(v0, v1, wxh[WREG ? g : 0][WREG ? sx : 0], wxl[WREG ? g : 0][WREG ? sx : 0]);
      }
  }
  if (FUSE) {
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wih, 0, 4 * H * in * 4, kBufFlags);
    for (int i = tid; i < (WREG ? 0 : 4 * US * (XK / 4)); i += 256) {
      const int row = i / (XK / 4), c4 = (i % (XK / 4)) * 4, g = row / US, r = row % US;
      const f32x4 v = ldg4<0>(rW, c4 < in ? (long)(g * H + j * US + r) * in + c4 : 0x1FFFFFFFl);
      u32x2 sh, sl;
      tb_split4(v, sh, sl);
      *reinterpret_cast<u32x2*>(wiH + row * LDX + c4) = sh;
      *reinterpret_cast<u32x2*>(wiL + row * LDX + c4) = sl;
    }
    for (int i = tid; i < 4 * US; i += 256) {
      const int o = (i / US) * H + j * US + i % US;
      bsum[i] = (a.b_ih ? a.b_ih[o] : 0.f) + (a.b_hh ? a.b_hh[o] : 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 v0 = xload(0, mt), v1 = xload(1, mt);
      xstore(0, mt, v0);
      xstore(1, mt, v1);
    }
  }
  float c[MT][UE], pre[MT][4][UPW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = (active && !STAGE && !FUSE) ? ldgn<UPW, 0>(rP, (long)(mt * 16 + col) * rowP + g * H + u0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < UPW; ++u) pre[mt][g][u] = v[u];
    }
#pragma unroll
    for (int u = 0; u < UE; ++u) c[mt][u] = 0.f;
  }
  bool same_l2 = false;
  if (STAGE && !FUSE) {   // step 0's pre-activations into pbuf[0] (the handshake's barriers publish them)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = tid + 256 * i, seg = pc >> 3, c4 = pc & 7, r = seg >> 2, g = seg & 3;
      *reinterpret_cast<f32x4*>(pbuf + r * PROW + g * US + c4 * 4) = ldg4<0>(rP, (long)r * rowP + g * H + j * US + c4 * 4);
    }
  }
  if (NCL > 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the arming stores (and the loads above) have landed
    const int hs = cluster_hello(a.hello + (size_t)cl * NCL, a.epoch, NCL, j, a.timeout_ticks, a.err, s_word);   // has barriers
    if (hs < 0) return;
    same_l2 = hs == 1;
  } else {
    __syncthreads();
  }

  // the first pre-activations are complete before the loop (see the note at the prefetch wait below):
  // otherwise the loop-carried `pre` counts as pending and the gate math starts with a vmcnt(0)
  if (!STAGE && !FUSE) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < UPW; ++u) asm volatile("" : "+v"(pre[mt][g][u]));
  }
  // (W_ih . x_tt^T) of tile mt into r (from zero): r[g] = gate g, or -- DEAL -- r[0], r[1] = gates kp, kp + 2
  constexpr bool DEAL = FUSE && (NXS % KS != 0);
  constexpr int NR = DEAL ? 2 : 4;
  auto xraw = [&](int tt, int mt, f32x4 (&acc)[NR]) {
#pragma unroll
    for (int g = 0; g < NR; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (FUSE && active) {
          // (W_ih . x_t^T): x k-step s goes to the k-part wave s % KS; with fewer x k-steps than k-part waves the
          // gates are dealt out instead (the partial sums meet in the k-split swap below either way)
          int opq = 0;   // opaque zero: keeps the loop-invariant W_ih reads in the loop (hoisted, they cost 32-64 registers)
          asm volatile("" : "+v"(opq));
          if constexpr (!DEAL) {
#pragma unroll
            for (int sx = 0; sx < NXS; ++sx) {
              if (sx % KS != kp) continue;
              const int xo = ((tt & 1) * MT * 16 + mt * 16 + col) * LDX + sx * 32 + q * 8;
              const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xH + xo));
              const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xL + xo));
              bf16x8 ah[4], al[4];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                if (WREG) {
                  ah[g] = wxh[WREG ? g : 0][WREG ? sx : 0];
                  al[g] = wxl[WREG ? g : 0][WREG ? sx : 0];
                } else {
                  const int wo = (g * US + ug * 16 + col) * LDX + sx * 32 + q * 8 + opq;
                  ah[g] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiH + wo));
                  al[g] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiL + wo));
                }
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[g], bl, acc[g], 0, 0, 0);
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[g], bh, acc[g], 0, 0, 0);
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[g], bh, acc[g], 0, 0, 0);
            }
          } else {
            // one x k-step, two k-part waves: wave kp takes gates kp and kp + 2 (operands picked by address, the two
            // results dealt into the accumulators by a wave-uniform select)
            static_assert(NXS == 1 && KS == 2, "gate dealing is written for one k-step and two k-part waves");
            const int xo = ((tt & 1) * MT * 16 + mt * 16 + col) * LDX + q * 8;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xH + xo));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xL + xo));
            const int w0 = (kp * US + ug * 16 + col) * LDX + q * 8 + opq, w1 = w0 + 2 * US * LDX;
            const bf16x8 ah0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiH + w0));
            const bf16x8 al0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiL + w0));
            const bf16x8 ah1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiH + w1));
            const bf16x8 al1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wiL + w1));
            f32x4 t0 = acc[0], t1 = acc[1];
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bl, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bl, t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, bh, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, bh, t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bh, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bh, t1, 0, 0, 0);
            acc[0] = t0;
            acc[1] = t1;
          }
    }
  };
  auto xdeal = [&](const f32x4 (&r)[NR], f32x4 (&acc)[4]) {
    if constexpr (DEAL) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      acc[0] = kp == 0 ? r[0] : zero;
      acc[1] = kp == 0 ? zero : r[0];
      acc[2] = kp == 0 ? r[1] : zero;
      acc[3] = kp == 0 ? zero : r[1];
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = r[g];
    }
  };
  // One tile per cluster: the x part of step t+1 is computed in the shadow of step t's hand-off (after the publish,
  // before the poll) into the accumulators that the next MFMA block continues from -- the same operations in the same
  // order as in the variants that do it at the top of the MFMA block, so the same bits.
  // Single-workgroup layers (NCL == 1) have no hand-off, but the matrix pipe idles during the gate math: there the x
  // part of a tile's next step is issued in front of the gate math of its current one, for every tile.
  constexpr bool XAHEAD = FUSE && (MT == 1 || NCL == 1);
  f32x4 accx[XAHEAD ? MT : 1][NR];
  if (XAHEAD) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xraw(0, mt, accx[mt]);
  }
  // (the handshake's barriers above made the W_ih rows and x_0 visible)
  // STAGE: y and the stash of step tt from sbuf to HBM as whole 128-byte runs.  Piece (i, thread): item = 2 i + (wave >= 2)
  // -- the four gates, c, y -- of row (tid >> 3) & 15, so every wave's store instruction has ONE destination (no per-lane
  // branches), and the three LDS reads are all requested before the first store.  (Written as read -> store three times
  // over with per-lane destinations, hipcc waited for each read in turn behind divergent branches: ~750 cycles at the
  // top of every step.)
  auto flush = [&](int tt) {
    const int r = (tid >> 3) & 15, c4 = (tid & 7) * 4;
    const int hi = __builtin_amdgcn_readfirstlane(wave >> 1);   // scalar: the destination select below is a branch, not a waterfall loop
    f32x4 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const f32x4*>(sbuf + r * SROW + (2 * i + hi) * US + c4);
    const long oy = (long)r * rowY + (long)tt * H + j * US + c4;
    if (a.train) {
#pragma unroll
      for (int i = 0; i < 2; ++i) stg4<0>(rP, (long)r * rowP + (long)tt * 4 * H + (2 * i + hi) * H + j * US + c4, v[i]);
    }
    if (hi) stg4<0>(rY, oy, v[2]);
    else if (a.train) stg4<0>(rC, oy, v[2]);
  };
  STAMP_DECL;
#ifdef TB_STAMPS
  stamp_acc[10] = same_l2 ? 1000 * T : 0;   // shows as 1000 per step when the cluster shares an L2
  { unsigned xx; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xx)); stamp_acc[11] = (unsigned long long)xx * T; }
#endif
  // Schedule.  The MT batch tiles of a cluster are independent recurrences; they are stepped in turn
  // ("phases"), each phase = MFMA of one tile, then -- concurrently -- the finishing waves' gate math
  // and publish for that tile and the puller waves' pull for the tile whose MFMA comes NEXT:
  //   phase (t, x):  MFMA_x(t) | barrier | gates_x(t), publish h_x(t)  ||  pull for tile (x+1) % MT | barrier
  // so with MT >= 2 a tile's hand-off (publish -> visible -> pulled) hides under another tile's MFMA and
  // gate math, and a step of the cluster costs MT x (MFMA + max(gates, pull)) instead of
  // MT x (MFMA + gates) + hand-off + pull.  The pull of phase x < MT-1 fetches h_{x+1}(t-1), the pull
  // of phase MT-1 fetches h_0(t); MT == 1 is the plain schedule.
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int x = 0; x < MT; ++x) {
      const int mt = x;
      STAMP(0);
      if (STAGE && t > 0) flush(t - 1);   // y and the stash of step t-1, staged in sbuf before the barrier that ended that step
      // (A) every consumer has read h_mt(t-2) (it published h_mt(t-1) after that, and our pull of
      // h_mt(t-1) is complete): re-arm this wave's part of that slot, which is written next at step t+1
      if (NCL > 1 && t >= 2) {
        const f32x4 arm = __builtin_bit_cast(f32x4, (u32x4){kSentinel, kSentinel, kSentinel, kSentinel});
        if (STAGE) {
          // the member's whole slice as row-contiguous 16-byte pieces from the first 16*US/4 threads (8 full
          // lines per instruction instead of 16 partial ones from every wave).  Any wave may re-arm any word
          // of the slice: the store is acknowledged before that wave publishes (the guard below), hence
          // before the barrier that ends this step, and the slot is written again only in the next step.
          if (tid < 16 * (US / 4)) stx4(same_l2, rR, (long)((t + 1) % 3) * RS + (long)(tid / (US / 4)) * H + j * US + (tid % (US / 4)) * 4, arm);
        } else {
          stxn<UPW>(same_l2, rR, (long)((t + 1) % 3) * RS + (long)(mt * 16 + col) * H + u0, arm);
        }
      }
      // next step's pre-activations of this tile: independent of the recurrence.  Issued HERE, a whole MFMA
      // + gate phase before the next hand-off load of this wave: loads return in order, so an HBM miss
      // issued just before a poll holds the poll and the pull behind it (measured: +2000 cycles per step)
      float nxt[4][UPW];
      f32x4 pl[2];
      f32x4 xl = {0.f, 0.f, 0.f, 0.f};
      if (FUSE) xl = xload(t + 2, mt);   // x_{t+2} of this tile, into the LDS half that x_t leaves in this phase
      if (STAGE && !FUSE && t + 1 < T) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pc = tid + 256 * i, seg = pc >> 3, c4 = pc & 7, r = seg >> 2, g = seg & 3;
          pl[i] = ldg4<0>(rP, (long)r * rowP + (long)(t + 1) * 4 * H + g * H + j * US + c4 * 4);
        }
      }
      if (!STAGE && !FUSE && t + 1 < T && active) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ldgn<UPW, 0>(rP, (long)(mt * 16 + col) * rowP + (long)(t + 1) * 4 * H + g * H + u0);
#pragma unroll
          for (int u = 0; u < UPW; ++u) nxt[g][u] = v[u];
        }
      }
      // (B) (W_hh . h_{t-1}^T) on the matrix cores (h_{-1} = 0: nothing to do at t = 0)
      f32x4 acc[4], acc2[EMU2 ? 4 : 1];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < (EMU2 ? 4 : 1); ++g) acc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (FUSE) {
        const int am = XAHEAD ? mt : 0;
        if (!XAHEAD) xraw(t, mt, accx[0]);
        xdeal(accx[am], acc);
        if (EMU2 && NXS == 1) {   // the 8-member layout deals one x k-step by gate: 0, 2 to the low k half, 1, 3 to the high one
          acc2[EMU2 ? 1 : 0] = acc[1]; acc2[EMU2 ? 3 : 0] = acc[3];
          acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      if ((FUSE || t > 0) && active) {
        if (!FUSE || t > 0) {
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const int ho = (mt * 16 + col) * LDH + kp * KPER + s * 32 + q * 8;
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(hiL + ho));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(loL + ho));
          // three passes over the four independent accumulators: no MFMA waits on the one before it
          if (EMU2 && s >= NKS / 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc2[EMU2 ? g : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][s], bl, acc2[EMU2 ? g : 0], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc2[EMU2 ? g : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[g][s], bh, acc2[EMU2 ? g : 0], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc2[EMU2 ? g : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][s], bh, acc2[EMU2 ? g : 0], 0, 0, 0);
            continue;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][s], bl, acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[g][s], bh, acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][s], bh, acc[g], 0, 0, 0);
        }
        }
        if (KS > 1) {   // hand every other k-part wave of this unit group the partial sums of ITS units
#pragma unroll
          for (int d = 0; d < KS; ++d) {
            if (d == kp) continue;
            float* dst = red + (size_t)(((ug * KS + d) * (KS - 1) + (kp - d - 1 + KS) % KS) * 64 + lane) * 4 * UPW;
            if (UPW == 2) {
              *reinterpret_cast<f32x4*>(dst) = (f32x4){acc[0][d * 2], acc[0][d * 2 + 1], acc[1][d * 2], acc[1][d * 2 + 1]};
              *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){acc[2][d * 2], acc[2][d * 2 + 1], acc[3][d * 2], acc[3][d * 2 + 1]};
            } else {
              *reinterpret_cast<f32x4*>(dst) = (f32x4){acc[0][d], acc[1][d], acc[2][d], acc[3][d]};
            }
          }
        }
      }
      STAMP(1);
      __syncthreads();   // all reads of this tile's h done; k-split partials visible
      STAMP(2);
      // which tile / step is pulled in this phase (for the MFMA of the next phase)
      const int pt = (x + 1 < MT) ? x + 1 : 0;
      const int ps = (x + 1 < MT) ? t - 1 : t;
      const bool pull = NCL > 1 && ps >= 0 && ps + 1 < T;
      const long sbase = (long)(ps % 3) * RS + (long)pt * 16 * H;
      // With several tiles per cluster the slices pulled in this phase were published a whole phase ago:
      // their loads are issued BEFORE the gate math and checked after it (what is still armed is re-pulled)
      u32x4 buf[NLD];
      if (MT > 1 && pull) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int m = wave + (n / LPP) * NPW;
          const int pc = (n % LPP) * 64 + lane, row = pc / (US / 4), c4 = pc % (US / 4);
          if (m < NCL && m != j) buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)((sbase + (long)row * H + m * US + c4 * 4) * 4), 0, kSC1);
        }
      }
      float st[4][UE], hn[UE];
      if (active) {
        // (C) gate math, state update, h into LDS and the ring -- UE consecutive units of one batch row per lane
        float z[4][UE];
        if (FUSE) {
          const float* pb = bsum + (u0 - j * US);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 w = UE == 4 ? *reinterpret_cast<const f32x4*>(pb + g * US)
                                    : (UE == 2 ? (f32x4){pb[g * US], pb[g * US + 1], 0.f, 0.f} : (f32x4){pb[g * US], 0.f, 0.f, 0.f});
#pragma unroll
            for (int u = 0; u < UE; ++u) pre[mt][g][u] = w[u];
          }
        } else if (STAGE) {
          const float* pb = pbuf + (t & 1) * 16 * PROW + col * PROW + (u0 - j * US);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(pb + g * US);
            pre[mt][g][0] = __uint_as_float(w[0]);
            pre[mt][g][UPW - 1] = __uint_as_float(w[1]);
          }
        }
        if constexpr (CMP > 1) {
          // the sums of this lane's units sit in lane rowl of its 16-lane row, elements sub*UE ..: row_shr by sub * (16 / CMP)
          auto shr = [](float v, auto n) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + decltype(n)::value, 0xF, 0xF, false)); };
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if constexpr (CMP == 2) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const float hi = shr(acc[g][2 + u], std::integral_constant<int, 8>{});
                z[g][u] = (sub ? hi : acc[g][u]) + pre[mt][g][u];
              }
            } else {
              static_assert(CMP == 2 || CMP == 4, "compaction by 2 or 4");
              const float s1 = shr(acc[g][1], std::integral_constant<int, 4>{}), s2 = shr(acc[g][2], std::integral_constant<int, 8>{}),
                          s3 = shr(acc[g][3], std::integral_constant<int, 12>{});
              z[g][0] = (sub == 0 ? acc[g][0] : (sub == 1 ? s1 : (sub == 2 ? s2 : s3))) + pre[mt][g][0];
            }
          }
        } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int u = 0; u < UPW; ++u) {
            if (EMU2) {   // units 0, 1 of a lane's four: finished by the low-k wave of the 8-member layout; 2, 3: by the high-k one
              const float lo = acc[g][u], hi = acc2[EMU2 ? g : 0][u];
              z[g][u] = u < 2 ? (lo + pre[mt][g][u]) + hi : (hi + pre[mt][g][u]) + lo;
            } else {
              // kp is wave-uniform: a scalar branch per k part with constant accumulator indices (indexed by kp directly,
              // hipcc reads all 16 accumulators and picks with three selects per value)
#pragma unroll
              for (int kk = 0; kk < KS; ++kk)
                if (kp == kk) z[g][u] = acc[g][kk * UPW + u] + pre[mt][g][u];
            }
          }
        }
        if constexpr (KS > 1) if (FUSE || t > 0) {
#pragma unroll
          for (int sr = 0; sr < KS; ++sr) {
            if (sr == kp) continue;
            const float* src = red + (size_t)(((ug * KS + kp) * (KS - 1) + (sr - kp - 1 + KS) % KS) * 64 + lane) * 4 * UPW;
            if (UPW == 2) {
              const f32x4 p0 = *reinterpret_cast<const f32x4*>(src), p1 = *reinterpret_cast<const f32x4*>(src + 4);
              z[0][0] += p0[0]; z[0][1] += p0[1]; z[1][0] += p0[2]; z[1][1] += p0[3];
              z[2][0] += p1[0]; z[2][1] += p1[1]; z[3][0] += p1[2]; z[3][1] += p1[3];
            } else {
              const f32x4 p0 = *reinterpret_cast<const f32x4*>(src);
              z[0][0] += p0[0]; z[1][0] += p0[1]; z[2][0] += p0[2]; z[3][0] += p0[3];
            }
          }
        }
        // single-workgroup layers: the matrix pipe works on this tile's next x part while the VALU does the gate math
        if (XAHEAD && NCL == 1 && t + 1 < T) xraw(t + 1, mt, accx[XAHEAD ? mt : 0]);
#pragma unroll
        for (int u = 0; u < UE; ++u) {
          const float ig = fsig(z[0][u]), fg = fsig(z[1][u]), gg = ftanh(z[2][u]), og = fsig(z[3][u]);
          // explicit fma: left to -ffp-contract the compiler picks either product, differently per instantiation
          const float cn = __builtin_fmaf(fg, c[mt][u], ig * gg);
          c[mt][u] = cn;
          hn[u] = og * ftanh(cn);
          st[0][u] = ig; st[1][u] = fg; st[2][u] = gg; st[3][u] = og;
        }
        // h_t of these units: split into bf16 hi / lo for the LDS tile; the ring carries them as one
        // {hi:16 | lo:16} word per element, so the consumers only regroup the halves
        unsigned short* ph = hiL + (mt * 16 + rowl) * LDH + u0;
        unsigned short* pl = loL + (mt * 16 + rowl) * LDH + u0;
        f32x4 hx = {0.f, 0.f, 0.f, 0.f};
        if constexpr (UE == 4) {
          u32x2 sh, sl;
          tb_split4((f32x4){hn[0], hn[1], hn[2], hn[3]}, sh, sl);
          *reinterpret_cast<u32x2*>(ph) = sh;
          *reinterpret_cast<u32x2*>(pl) = sl;
          hx = __builtin_bit_cast(f32x4, (u32x4){__builtin_amdgcn_perm(sh[0], sl[0], 0x05040100u), __builtin_amdgcn_perm(sh[0], sl[0], 0x07060302u),
                                                __builtin_amdgcn_perm(sh[1], sl[1], 0x05040100u), __builtin_amdgcn_perm(sh[1], sl[1], 0x07060302u)});
        } else if constexpr (UE == 2) {
          unsigned sh, sl;
          tb_split2(hn[0], hn[1], sh, sl);
          *reinterpret_cast<unsigned*>(ph) = sh;
          *reinterpret_cast<unsigned*>(pl) = sl;
          hx[0] = __uint_as_float(__builtin_amdgcn_perm(sh, sl, 0x05040100u));
          hx[1] = __uint_as_float(__builtin_amdgcn_perm(sh, sl, 0x07060302u));
        } else {
          unsigned sh, sl;
          tb_split2(hn[0], 0.f, sh, sl);
          *ph = (unsigned short)(sh & 0xFFFFu);
          *pl = (unsigned short)(sl & 0xFFFFu);
          hx[0] = __uint_as_float((sh << 16) | (sl & 0xFFFFu));
        }
        if (NCL > 1 && t + 1 < T) {
          // the re-arm of (A) (and every older store of this wave) is acknowledged before new data enters
          // the ring: a consumer that sees h_t can no longer see an un-armed slot t+1.  Memory operations
          // of a wave complete in order: all but the 4 prefetch loads (and, with MT > 1, the early pull loads)
          // issued after the re-arm.
          STAMP(5);
          // The count is the MINIMUM any wave issues there: the wave that has this member among its producers
          // skips that producer's LPP early pull loads (a larger count would not force the re-arm to complete).
          constexpr int NPF = FUSE ? 1 : (STAGE ? 2 : 4);   // prefetch loads every wave issues after the re-arm
          constexpr int AFTER_REARM = MT == 1 ? NPF : NPF + NLD - LPP;
          static_assert(AFTER_REARM >= 0 && AFTER_REARM < 64, "vmcnt immediate");
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER_REARM) : "memory");
          STAMP(6);
          stxn<UPW>(same_l2, rR, (long)(t % 3) * RS + (long)(mt * 16 + col) * H + u0, hx);
        }
      }
      if constexpr (STAGE) {
        // y and the stash of this step into sbuf (flushed at the top of the next step; the flush of step t-1 read
        // it before this step's first barrier).  Here rather than after the pull: it ends the life of st / hn
        // before the pull buffers begin theirs.
        float* sb = sbuf + col * SROW + (u0 - j * US);
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(sb + g * US) = (u32x2){__float_as_uint(st[g][0]), __float_as_uint(st[g][UPW - 1])};
        *reinterpret_cast<u32x2*>(sb + 4 * US) = (u32x2){__float_as_uint(c[mt][0]), __float_as_uint(c[mt][UPW - 1])};
        *reinterpret_cast<u32x2*>(sb + 5 * US) = (u32x2){__float_as_uint(hn[0]), __float_as_uint(hn[UPW - 1])};
      }
      if (XAHEAD && NCL > 1 && t + 1 < T) xraw(t + 1, 0, accx[0]);
      STAMP(3);
      if (pull) {
        // (D) the other members' slices of h_pt(ps) from the ring into the LDS tile; wave w takes producers
        // w, w+4, ...  The ring is sentinel-armed and every slot word is written once per turn, so the data
        // is its own ready flag: pull, then re-pull whatever is still armed (a piece is complete when none of
        // its 4 words is armed: its parts come from different waves).  No separate poll of one piece in front of
        // the pull: the x part above (or the other tiles' phases) covers most of the hand-off, so the first pull
        // usually finds the data and a poll only added an L2 round trip (2.62 -> 2.48 us/step at hidden 256).
        // Only the ring store above precedes these loads in this wave's memory queue: y and the stash are
        // stored after the pull.
        // Clusters of more than 8 members (hidden 512: 32 members) keep the poll: with that many producers per wave the
        // first pull mostly comes back armed and the re-pull loop hammers the ring (5.06 against 4.52 us/step without it).
        constexpr bool POLL = NCL > (TB_POLL_ABOVE);
        if (POLL && MT == 1 && lane < NPP * NUG) {
          const int m = wave + (lane / NUG) * NPW;
          if (m < NCL && m != j) {
            const int voff = (int)((sbase + (long)15 * H + m * US + (lane % NUG) * 16 + 12) * 4);
            const unsigned long long t0 = wall_clock64();
            unsigned it = 0;
            while (armed(__builtin_amdgcn_raw_buffer_load_b128(rR, voff, 0, kSC1))) {
              __builtin_amdgcn_s_sleep(1);
              if ((++it & 1023u) == 0 && wall_clock64() - t0 > a.timeout_ticks) {
                atomicExch(a.err, 1);
                *s_dead = 1;
                break;
              }
            }
          }
        }
        STAMP(4);
        // first round without per-lane predicates (which pieces a wave pulls is wave-uniform: straight-line loads, no
        // exec juggling); the loop below only re-pulls what came back armed
        if (MT == 1) {
#pragma unroll
          for (int n = 0; n < NLD; ++n) {
            const int m = wave + (n / LPP) * NPW;
            const int pc = (n % LPP) * 64 + lane, row = pc / (US / 4), c4 = pc % (US / 4);
            if (m < NCL && m != j) buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)((sbase + (long)row * H + m * US + c4 * 4) * 4), 0, kSC1);
          }
        }
        unsigned pend = 0;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int m = wave + (n / LPP) * NPW;
          if (m < NCL && m != j && armed(buf[n])) pend |= 1u << n;
        }
        unsigned it = 0;
        unsigned long long t0 = 0;
        while (pend) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int n = 0; n < NLD; ++n) {
            const int m = wave + (n / LPP) * NPW;
            const int pc = (n % LPP) * 64 + lane, row = pc / (US / 4), c4 = pc % (US / 4);
            if ((pend >> n) & 1u) buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)((sbase + (long)row * H + m * US + c4 * 4) * 4), 0, kSC1);
          }
#pragma unroll
          for (int n = 0; n < NLD; ++n)
            if (((pend >> n) & 1u) && !armed(buf[n])) pend &= ~(1u << n);
          if (pend && (++it & 255u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > a.timeout_ticks) {
              atomicExch(a.err, 1);
              *s_dead = 1;
              break;
            }
          }
        }
        STAMP(7);
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int m = wave + (n / LPP) * NPW;
          const int pc = (n % LPP) * 64 + lane, row = pt * 16 + pc / (US / 4), c4 = pc % (US / 4);
          if (m < NCL && m != j) {
            const u32x2 sh = {__builtin_amdgcn_perm(buf[n][1], buf[n][0], 0x07060302u), __builtin_amdgcn_perm(buf[n][3], buf[n][2], 0x07060302u)};
            const u32x2 sl = {__builtin_amdgcn_perm(buf[n][1], buf[n][0], 0x05040100u), __builtin_amdgcn_perm(buf[n][3], buf[n][2], 0x05040100u)};
            *reinterpret_cast<u32x2*>(hiL + row * LDH + m * US + c4 * 4) = sh;
            *reinterpret_cast<u32x2*>(loL + row * LDH + m * US + c4 * 4) = sl;
          }
        }
      }
      // Make the wait for the prefetched pre-activations fall HERE, where everything older than the pull
      // has long completed: left alone, the compiler waits for them with vmcnt(0) at their first use in
      // the next phase's gate math -- behind the prefetch loads issued just before it (HBM latency, ~1200
      // cycles on the critical path of every step).
      if (FUSE) xstore(t + 2, mt, xl);   // (t + 2) & 1 == t & 1: x_t was read before this phase's first barrier
      if (STAGE) {
        if (!FUSE && t + 1 < T) {   // the next step's pre-activations into the other pbuf half
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int pc = tid + 256 * i, seg = pc >> 3, c4 = pc & 7, r = seg >> 2, g = seg & 3;
            *reinterpret_cast<f32x4*>(pbuf + ((t + 1) & 1) * 16 * PROW + r * PROW + g * US + c4 * 4) = pl[i];
          }
        }
      }
      if (!STAGE && !FUSE) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int u = 0; u < UPW; ++u) asm volatile("" : "+v"(nxt[g][u]));
      }
      if (active && !STAGE) {
        // (E) y for the next layer and the BPTT stash (activated gates over P in place, cell state): after
        // the pull, so that no HBM store sits in front of the hand-off loads
        // Compacted: the UPW units of a row go back to the lane the MFMA left them in (row_shl within the 16-lane row),
        // so that y and the stash leave as the same 16-byte stores as from whole tiles (as 4-byte stores from every
        // lane the kernel alone was as fast, but the iteration around it 1.4 % slower: four times the store
        // instructions' address work in the memory pipe the co-running kernels share).  Lanes with sub != 0 sit on
        // rows >= 16 / CMP, which the descriptors drop.
        auto whole = [&](const float (&src)[UE]) -> f32x4 {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if constexpr (CMP == 1) {
#pragma unroll
            for (int u = 0; u < UE; ++u) o[u] = src[u];
          } else {
            auto shl = [](float v, auto n) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + decltype(n)::value, 0xF, 0xF, false)); };
            if constexpr (CMP == 2) {
              o[0] = src[0]; o[1] = src[1];
              o[2] = shl(src[0], std::integral_constant<int, 8>{}); o[3] = shl(src[1], std::integral_constant<int, 8>{});
            } else {
              o[0] = src[0]; o[1] = shl(src[0], std::integral_constant<int, 4>{});
              o[2] = shl(src[0], std::integral_constant<int, 8>{}); o[3] = shl(src[0], std::integral_constant<int, 12>{});
            }
          }
          return o;
        };
        stgn<UPW, 0>(rY, (long)(mt * 16 + col) * rowY + (long)t * H + u0m, whole(hn));
        if (a.train) {
#pragma unroll
          for (int g = 0; g < 4; ++g) stgn<UPW, 0>(rP, (long)(mt * 16 + col) * rowP + (long)t * 4 * H + g * H + u0m, whole(st[g]));
          stgn<UPW, 0>(rC, (long)(mt * 16 + col) * rowY + (long)t * H + u0m, whole(c[mt]));
        }
        if (!FUSE) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < UPW; ++u) pre[mt][g][u] = nxt[g][u];
        }
      }
      STAMP(8);
      __syncthreads();   // the pulled tile is complete for its MFMA in the next phase
      STAMP(9);
    }
    if (NCL > 1 && *s_dead) break;
  }
  if (STAGE && !*s_dead) flush(T - 1);   // y and the stash of the last step (the loop ends with a barrier)
  STAMP_FLUSH(0, cl, j);
}

// ----------------------------------------------------------------------------------------
struct BwdArgs {
  float* G;            // [B][T][4H] in: activated gates; out: dgates (wrt pre-activations)
  const float* cst;    // [B][T][H]
  const float* dy;     // [B][T][H]
  const float* Whh;    // [4H][H]
  float* X;            // [nclusters][3][NCL consumers][NCL producers][US/16][MT*16][16] exchange slabs (NCL > 1)
  float* dbpart;       // [nclusters_total][4H] per-cluster bias-grad partials
  unsigned long long* hello;   // [nclusters][NCL] start-up handshake tags
  unsigned long long epoch;    // this launch's handshake epoch
  int* err;
  int B, T;
  int b_begin;
  int rows;            // batch rows per cluster (<= 16 * MT)
  int cl_begin;        // index of this launch's first cluster in dbpart
  int nclusters;
  unsigned long long timeout_ticks;
  // input gradient inside the kernel (DXK > 0): every member writes its partial dx = dgates[:, member's gate rows] . W_ih[those rows]
  const float* Wih;    // [4H][in_dim]
  float* dxpart;       // [NCL][clusters][T][16][in_dim] partial input gradients (a cluster's step is one contiguous block: one page
                       // per store instead of 16), summed by dx_reduce_kernel; nullptr: not wanted
  int ncl_total;       // clusters of the whole batch (all launches)
  int in_dim;
  int dx_b_begin;      // batch rows before it need no input gradient
  int store_dg;        // 0: nobody reads dgates afterwards (no weight gradients wanted, dx formed in the kernel): G is left as it is
};

template <int H, int US, int MT, int DXK = 0>
__global__ __launch_bounds__(256, 1) void lstm_bwd_kernel(BwdArgs a) {
  constexpr int NCL = H / US;
  constexpr int KD = 4 * US;                       // k dimension = this member's gate rows
  constexpr int NKD = KD / 32;                     // 32-deep MFMA k-steps
  constexpr int LDG = KD + 8;                      // LDS row stride of the dgates tile (bf16 elements)
  constexpr int NTW = H >= 64 ? H / 64 : 1;        // 16-col output tiles per wave
  constexpr int NAW = H >= 64 ? 4 : H / 16;        // waves doing MFMA work
  constexpr int UI = US / 16;                      // units per elementwise item: 16 items per row, so that
  constexpr int QPR = US / UI;                     // one 16-row tile is exactly 256 items -- every thread works
  constexpr int NE = MT * 16 * QPR;                // (row, unit group) items of the tile
  constexpr int NQ = (NE + 255) / 256;             // items per thread
  constexpr int LDH = H + 4;                       // dhL row stride (floats)
  static_assert(UI == 1 || UI == 2 || UI == 4, "US must be 16, 32 or 64");
  static_assert((256 % QPR) == 0 && (NE % 256 == 0 || NE < 256), "bad config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  // dgates_t of this member as split bf16 (hi, lo): [MT*16 rows][LDG]
  // (DXK > 0: two such tiles, by step parity -- the partial input gradient reads a step's tile in the shadow of the hand-off,
  // with no barrier between it and the next step's gate math, which therefore writes the other one)
  constexpr int NB = DXK > 0 ? 2 : 1;
  constexpr int DGT = MT * 16 * LDG;               // elements of one image (hi or lo)
  unsigned short* dg0 = reinterpret_cast<unsigned short*>(smem);
  float* dhL = smem + NB * DGT;                    // [MT][16][LDH] this member's partial dh_{t-1}
  float* dbL = dhL + MT * 16 * LDH;                // [256][16] bias-grad reduction
  int* s_dead = reinterpret_cast<int*>(dbL + 4096);
  int* s_word = s_dead + 1;
  // DXK > 0: W_ih^T of this member's gate rows as split bf16, [DXK input features][LDG] (k = member's gate row, contiguous):
  // the A operand of the partial input gradient dx^T = W_ih^T . dgates^T, whose B operand is the dgates tile above
  unsigned short* wxH = reinterpret_cast<unsigned short*>(dbL + 4096 + 4);
  unsigned short* wxL = wxH + (DXK > 0 ? DXK : 1) * LDG;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform branches and addresses
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(NCL, cl, j);
  const int b0 = a.b_begin + cl * a.rows;
  const int T = a.T;
  const long rowG = (long)T * 4 * H, rowY = (long)T * H;
  constexpr int XSLAB = NCL * NCL * MT * 16 * US;  // floats per slab
  // Inside a (consumer, producer) block the 16-unit groups come first, then the rows: [US/16][MT*16][16].  A wave's
  // partial-sum store -- 16 rows x 16 units -- is then one contiguous KB (8 full lines) instead of 64 bytes in each of
  // 16 lines, and a reduce-scatter load instruction reads runs of 4 rows x 64 bytes.
  auto xoff = [](int row, int u) { return (u >> 4) * (MT * 16 * 16) + row * 16 + (u & 15); };
  float* Xc = a.X + (size_t)cl * 3 * XSLAB;
  if (tid == 0) *s_dead = 0;
  int vrows = a.B - b0;
  if (vrows > a.rows) vrows = a.rows;
  const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.G + (long)b0 * rowG), 0, (int)(vrows * rowG * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cst + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)Xc, 0, NCL > 1 ? 3 * XSLAB * 4 : 0, kBufFlags);
  // partial input gradient of this member: rows of this cluster in dxpart[j]; clusters whose rows all lie before
  // dx_b_begin (the real half of the discriminator's real + fake pass) skip it
  const int xin = DXK > 0 ? a.in_dim : 0;
  const bool dxon = DXK > 0 && a.dxpart != nullptr && b0 + vrows > a.dx_b_begin;
  const __amdgpu_buffer_rsrc_t rDX = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dxon ? a.dxpart + ((size_t)j * a.ncl_total + a.cl_begin + cl) * T * (MT * 16) * xin : a.X), 0,
      dxon ? (int)((long)T * (MT * 16) * xin * 4) : 0, kBufFlags);
  if (DXK > 0) {
    for (int i = tid; i < DXK * KD; i += 256) {
      const int m = i / KD, k = i % KD;
      const int grow = (k / US) * H + j * US + (k % US);
      const float wv = (dxon && m < xin) ? a.Wih[(long)grow * xin + m] : 0.f;
      unsigned sh, sl;
      tb_split2(wv, 0.f, sh, sl);
      wxH[m * LDG + k] = (unsigned short)(sh & 0xFFFFu);
      wxL[m * LDG + k] = (unsigned short)(sl & 0xFFFFu);
    }
  }
  if (NCL > 1) {
    // arm the part of the three slabs this member CONSUMES ([slab][j][every producer][rows][units], one
    // contiguous run per slab; its own never-written slot included); write-through, acknowledged before
    // the handshake tag goes out, see cluster_hello()
    const f32x4 arm = __builtin_bit_cast(f32x4, (u32x4){kSentinel, kSentinel, kSentinel, kSentinel});
    constexpr int PER = NCL * MT * 16 * US / 4;   // 16-byte pieces per slab
    for (int i = tid; i < 3 * PER; i += 256) {
      const int slab = i / PER, r = i % PER;
      stg4<kSC1>(rX, (long)slab * XSLAB + (long)j * NCL * MT * 16 * US + (long)r * 4, arm);
    }
  }

  // W_hh fragments (A operand of the transposed product dh^T = W_hh^T . dgates^T), split bf16 hi/lo:
  // A[m][k] = W_hh[grow(k)][n0 + m], k = s*32 + q*8 + i, grow(k) = (k/US)*H + j*US + k%US
  bf16x8 wbh[NTW][NKD], wbl[NTW][NKD];
  const bool mactive = wave < NAW;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int s = 0; s < NKD; ++s) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (mactive) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = s * 32 + q * 8 + i;
          const int grow = (k / US) * H + j * US + (k % US);
          const float wv = a.Whh[(long)grow * H + (wave * NTW + nt) * 16 + col];
          if (i < 4) v0[i] = wv; else v1[i - 4] = wv;
        }
      }
      tb_split8(v0, v1, wbh[nt][s], wbl[nt][s]);
      // 4-member layout (256 registers of fragments): the lo halves live in accumulation registers, where the MFMAs read
      // them in place (left to the allocator they are spilled there and copied back in front of every use)
      if ((H == 256 && US == 64) || (H == 512 && US == 32)) asm volatile("" : "+a"(wbl[nt][s]));
    }
  // elementwise items: e = tid + 256*n -> unit group qd (constant per thread) of UI units, row, mt
  // (f32x4 containers, the first UI elements in use)
  const int qd = tid % QPR;
  const int ul0 = qd * UI, uabs0 = j * US + ul0;
  f32x4 dc[NQ], dhrec[NQ], dbacc[4];
  // Per item, the factors of a step's gate gradients that do not depend on dh_t -- computed a step AHEAD, in the shadow of
  // the previous step's hand-off, so that only seven multiply-adds per unit sit between the reduce-scatter and the LDS tile:
  //   [0] o (1 - tanh^2 c_t)   [1] tanh(c_t) o (1 - o)   [2] g i (1 - i)   [3] i (1 - g^2)   [4] c_{t-1} f (1 - f)   [5] f
  // then dy_t, and c_{t-1} (which is the next step's c_t).  (f32x4 containers, the first UI elements in use.)
  f32x4 fac[NQ][6], dyv[NQ], cprev[NQ];
  auto mkfac = [&](f32x4 (&o)[6], const f32x4& gi, const f32x4& gf, const f32x4& gg, const f32x4& go, const f32x4& ct, const f32x4& cp) {
#pragma unroll
    for (int r = 0; r < UI; ++r) {
      const float tc = ftanh(ct[r]);
      // explicit fma: left to -ffp-contract the compiler may pick differently per instantiation, and the layouts must agree bit for bit
      o[0][r] = go[r] * __builtin_fmaf(-tc, tc, 1.f);
      o[1][r] = tc * go[r] * (1.f - go[r]);
      o[2][r] = gg[r] * gi[r] * (1.f - gi[r]);
      o[3][r] = gi[r] * __builtin_fmaf(-gg[r], gg[r], 1.f);
      o[4][r] = cp[r] * gf[r] * (1.f - gf[r]);
      o[5][r] = gf[r];
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) dbacc[g] = zero4;
  f32x4 raw0[NQ][6];   // the last step's stash: i, f, g, o, c_{T-1}, c_{T-2}
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    dc[n] = zero4; dhrec[n] = zero4;
    const int e = tid + 256 * n;
    const int row = (e / QPR) & 15, mt = e / (16 * QPR);
    const long rr = mt * 16 + row;
    const bool ok = e < NE;
#pragma unroll
    for (int g = 0; g < 4; ++g) raw0[n][g] = ok ? ldgn<UI, 0>(rG, rr * rowG + (long)(T - 1) * 4 * H + g * H + uabs0) : zero4;
    raw0[n][4] = ok ? ldgn<UI, 0>(rC, rr * rowY + (long)(T - 1) * H + uabs0) : zero4;
    raw0[n][5] = (ok && T > 1) ? ldgn<UI, 0>(rC, rr * rowY + (long)(T - 2) * H + uabs0) : zero4;
    dyv[n] = ok ? ldgn<UI, 0>(rD, rr * rowY + (long)(T - 1) * H + uabs0) : zero4;
  }
  bool same_l2 = false;
  if (NCL > 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the arming stores (and the first step's stash) have landed
    const int hs = cluster_hello(a.hello + (size_t)cl * NCL, a.epoch, NCL, j, a.timeout_ticks, a.err, s_word);   // has barriers
    if (hs < 0) return;
    same_l2 = hs == 1;
  } else {
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the first step's stash is in registers before the loop
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    mkfac(fac[n], raw0[n][0], raw0[n][1], raw0[n][2], raw0[n][3], raw0[n][4], raw0[n][5]);
    cprev[n] = raw0[n][5];
  }

  STAMP_DECL;
  for (int t = T - 1; t >= 0; --t) {
    STAMP(0);
    unsigned short* dgHi = dg0 + (NB > 1 ? (t & 1) * 2 * DGT : 0);
    unsigned short* dgLo = dgHi + DGT;
    f32x4 nx[NQ][6];
    // prefetch step t-1 (its c_t is this step's c_{t-1}, already held), FIRST thing in the step: loads of a wave return
    // in order, so the reduce-scatter pull at the end of the step comes back no sooner than these HBM reads -- issued behind
    // the gate math they left it 0.3 us less (BPTT without them at all: 1.92 against 2.44 us/step at hidden 256).  The gate
    // math below works on values computed in the previous step, so no wait of its own can fall on these loads.
    if (t > 0) {
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        const int e = tid + 256 * n;
        const int row = (e / QPR) & 15, mt = e / (16 * QPR);
        const long rr = mt * 16 + row;
        const bool ok = e < NE;
        // branch-free: an item that does not exist (or step -1) asks for an offset past the descriptor's end, which returns 0
        constexpr long kOut = 0x1FFFFFFFl;
#pragma unroll
        for (int g = 0; g < 4; ++g) nx[n][g] = ldgn<UI, 0>(rG, ok ? rr * rowG + (long)(t - 1) * 4 * H + g * H + uabs0 : kOut);
        nx[n][4] = ldgn<UI, 0>(rC, (ok && t > 1) ? rr * rowY + (long)(t - 2) * H + uabs0 : kOut);
        nx[n][5] = ldgn<UI, 0>(rD, ok ? rr * rowY + (long)(t - 1) * H + uabs0 : kOut);
      }
    }
    // (A) gate gradients for this member's units
    f32x4 dgv[NQ][4];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
#pragma unroll
      for (int g = 0; g < 4; ++g) dgv[n][g] = zero4;
#pragma unroll
      for (int r = 0; r < UI; ++r) {
        const float dh = dyv[n][r] + dhrec[n][r];
        const float dcn = __builtin_fmaf(dh, fac[n][0][r], dc[n][r]);
        dgv[n][0][r] = dcn * fac[n][2][r];
        dgv[n][1][r] = dcn * fac[n][4][r];
        dgv[n][2][r] = dcn * fac[n][3][r];
        dgv[n][3][r] = dh * fac[n][1][r];
        dc[n][r] = dcn * fac[n][5][r];
      }
      if (e < NE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          dbacc[g] += dgv[n][g];
          unsigned short* ph = dgHi + (mt * 16 + row) * LDG + g * US + ul0;
          unsigned short* pl = dgLo + (mt * 16 + row) * LDG + g * US + ul0;
          if (UI == 4) {
            u32x2 sh, sl;
            tb_split4(dgv[n][g], sh, sl);
            *reinterpret_cast<u32x2*>(ph) = sh;
            *reinterpret_cast<u32x2*>(pl) = sl;
          } else {
            unsigned sh, sl;
            tb_split2(dgv[n][g][0], dgv[n][g][1], sh, sl);
            if (UI == 2) {
              *reinterpret_cast<unsigned*>(ph) = sh;
              *reinterpret_cast<unsigned*>(pl) = sl;
            } else {
              *ph = (unsigned short)(sh & 0xFFFFu);
              *pl = (unsigned short)(sl & 0xFFFFu);
            }
          }
        }
      }
    }
    STAMP(1);
    const bool last = t == 0;           // dh_{-1} is never used
    const int par = t % 3;              // exchange slab of this step
    if (!last) {
      __syncthreads();
      STAMP(2);
      if (NCL > 1 && *s_dead) break;   // a re-poll timed out (set before the barrier: uniform)
      // (B) partial dh_{t-1}^T = W_hh[rows of this member]^T . dgates_t^T
      if (mactive) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 acc[NTW];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[nt] = zero4;
#pragma unroll
          for (int s = 0; s < NKD; ++s) {
            const int go = (mt * 16 + col) * LDG + s * 32 + q * 8;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgHi + go));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgLo + go));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbh[nt][s], bl, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbl[nt][s], bh, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbh[nt][s], bh, acc[nt], 0, 0, 0);
          }
          STAMP(9);
          // lane: batch row `col`, output units n0 .. n0+3.  ONE branch on the store kind around the whole burst (a
          // branch per store: ~20 scalar branches per step in this block alone)
          auto put = [&](auto sc) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              const int n0 = (wave * NTW + nt) * 16 + q * 4;
              const int m = ((wave * NTW + nt) * 16) / US;   // wave-uniform (q * 4 < 16 never changes it): a scalar branch
              if (NCL == 1 || m == j) *reinterpret_cast<f32x4*>(dhL + (mt * 16 + col) * LDH + n0) = acc[nt];
              else stg4<decltype(sc)::value>(rX, (long)par * XSLAB + (long)(m * NCL + j) * MT * 16 * US + xoff(mt * 16 + col, n0 % US), acc[nt]);
            }
          };
          if (same_l2) put(std::integral_constant<int, 0>{});
          else put(std::integral_constant<int, kSC1>{});
        }
      }
      STAMP(10);
      // (B0) [issued behind the MFMA block and the partial-sum stores: nothing waits for it before the end of the step]
      // Re-arm what the previous step consumed: this member's whole region of that step's slab -- its own
      // never-written slot included -- as linear 16-byte pieces (whole lines; per-item 8-byte stores to 7 places
      // took 7 store instructions of 16 partial lines each).  Every thread's loads of that region completed
      // before the barrier above (their sums feed the gate math that precedes it), so any thread may re-arm any
      // word.  The slab of step t+1 is written again at step t-2; its writers get there only after consuming this
      // member's partial of step t-1, which is stored after the NEXT step's copy of the reduce-scatter wait --
      // and that wait covers these stores.  So the re-arm always lands before the next write.
      if (NCL > 1 && t < T - 1) {
        const f32x4 arm = __builtin_bit_cast(f32x4, (u32x4){kSentinel, kSentinel, kSentinel, kSentinel});
        constexpr int PER = NCL * MT * 16 * US / 4;   // 16-byte pieces of this member's region
        const long rb = (long)((t + 1) % 3) * XSLAB + (long)j * NCL * MT * 16 * US;
        auto rearm = [&](auto sc) {
#pragma unroll
          for (int i = 0; i < (PER + 255) / 256; ++i) {
            const int pc = tid + 256 * i;
            if (PER % 256 == 0 || pc < PER) stg4<decltype(sc)::value>(rX, rb + (long)pc * 4, arm);
          }
        };
        if (same_l2) rearm(std::integral_constant<int, 0>{});
        else rearm(std::integral_constant<int, kSC1>{});
      }
      STAMP(3);
      __syncthreads();   // own columns of dhL visible
      STAMP(4);
    }
    // dgates of this step over the gate stash, in place (after the publish: not on its critical path) -- unless no
    // kernel after this one reads them (the generator phase's pass through D: input gradient only, formed below)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
      if (e < NE && a.store_dg) {
#pragma unroll
        for (int g = 0; g < 4; ++g) stgn<UI, 0>(rG, (long)(mt * 16 + row) * rowG + (long)t * 4 * H + g * H + uabs0, dgv[n][g]);
      }
    }
    if (DXK > 0 && last) __syncthreads();   // the last step has no MFMA block and none of its barriers: the dgates tile is complete behind this one
    if (DXK > 0 && dxon && wave < DXK / 16) {
      // partial dx_t^T = W_ih^T[this member's gate rows] . dgates_t^T, in the shadow of the hand-off: the dgates tile stays
      // in LDS until the next step's gate math.  Wave w: input features 16 w .. 16 w + 15; lane: batch row `col`, 4 features.
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // three accumulators (one per term of the split product): no MFMA waits for the one before it
        f32x4 ax0 = zero4, ax1 = zero4, ax2 = zero4;
#pragma unroll
        for (int s = 0; s < NKD; ++s) {
          const int go = (mt * 16 + col) * LDG + s * 32 + q * 8, wo = (wave * 16 + col) * LDG + s * 32 + q * 8;
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgHi + go));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgLo + go));
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wxH + wo));
          const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wxL + wo));
          ax0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, ax0, 0, 0, 0);
          ax1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, ax1, 0, 0, 0);
          ax2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, ax2, 0, 0, 0);
        }
        const int f0 = wave * 16 + q * 4;
        stg4<0>(rDX, (f0 < xin && mt * 16 + col < vrows) ? ((long)t * (MT * 16) + mt * 16 + col) * xin + f0 : 0x1FFFFFFFl, (ax0 + ax1) + ax2);
      }
    }
    STAMP(5);
    if (last) break;
    // the next step's dh-independent factors, while the partial sums travel (c_{t-1} is held; its stash and dy were
    // requested a whole MFMA block ago)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      mkfac(fac[n], nx[n][0], nx[n][1], nx[n][2], nx[n][3], cprev[n], nx[n][4]);
      cprev[n] = nx[n][4];
      dyv[n] = nx[n][5];
    }
    // Clusters of more than 8 members (hidden 512: 32) keep the poll of one piece per producer by one wave, and the
    // barrier that hands its result to the others: every thread pulling 32 mostly armed pieces over and over costs far
    // more than the round trip saved (6.19 against 3.83 us/step).
    constexpr bool POLL = NCL > (TB_POLL_ABOVE);
    if constexpr (POLL) {
      if (wave == 0 && lane < NCL && lane != j) {
        const int voff = (int)(((long)par * XSLAB + ((long)(j * NCL + lane) * MT * 16 + MT * 16 - 1) * US + US - 4) * 4);
        const unsigned long long t0 = wall_clock64();
        unsigned it = 0;
        while (true) {
          if (!armed(__builtin_amdgcn_raw_buffer_load_b128(rX, voff, 0, kSC1))) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 1023u) == 0 && wall_clock64() - t0 > a.timeout_ticks) {
            atomicExch(a.err, 1);
            *s_dead = 1;
            break;
          }
        }
      }
      STAMP(6);
      __syncthreads();
      STAMP(7);
      if (*s_dead) break;
    }
    // (C) reduce-scatter: dh_{t-1} for this member's units = own partial + the other members' partials
    // (every slot is loaded, the never-written own one included: one predicate-free burst).
    // Data-as-flag: the slabs are armed with the sentinel (all-ones) before the launch, a slot that holds a value is
    // ready, and the consumer re-arms what it has consumed.  Every thread pulls its pieces and pulls again what came
    // back armed -- no poll of one piece per producer and no barrier in front: with the factor math above in the
    // shadow of the hand-off the first pull usually finds the data (2.58 -> 2.47 us/step at hidden 256), and the own
    // partial in dhL was made visible by the barrier behind the MFMA block.  A timeout is seen by the whole
    // workgroup behind the next step's first barrier.
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
      if (e < NE) {
        f32x4 s = zero4;
#pragma unroll
        for (int r = 0; r < UI; ++r) s[r] = dhL[(mt * 16 + row) * LDH + uabs0 + r];
        if (NCL > 1) {
          const long base = (long)par * XSLAB + (long)(j * NCL) * MT * 16 * US + xoff(mt * 16 + row, ul0);
          f32x4 v[NCL];
#pragma unroll
          for (int m = 0; m < NCL; ++m) v[m] = ldgn<UI, kSC1>(rX, base + (long)m * MT * 16 * US);
          {
            unsigned pend = 0;
#pragma unroll
            for (int m = 0; m < NCL; ++m)
              if (armedn<UI>(v[m])) pend |= 1u << m;
            pend &= ~(1u << j);   // the own slot is never written (branch-free: `m != j` inside the loop is a scalar branch per producer)
            unsigned it = 0;
            unsigned long long t0 = 0;
            while (pend) {
              asm volatile("" ::: "memory");
#pragma unroll
              for (int m = 0; m < NCL; ++m)
                if ((pend >> m) & 1u) {
                  v[m] = ldgn<UI, kSC1>(rX, base + (long)m * MT * 16 * US);
                  if (!armedn<UI>(v[m])) pend &= ~(1u << m);
                }
              if (pend && (++it & 255u) == 0) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > a.timeout_ticks) {
                  atomicExch(a.err, 1);
                  *s_dead = 1;
                  break;
                }
              }
            }
          }
#pragma unroll
          for (int m = 0; m < NCL; ++m) {
            const bool other = m != j;
#pragma unroll
            for (int r = 0; r < UI; ++r) s[r] += other ? v[m][r] : 0.f;
          }
          // (what was consumed is re-armed in the next step, as whole lines: see (B0) there)
        }
        dhrec[n] = s;
      }
    }
    STAMP(8);
    if (NCL == 1) __syncthreads();   // dhL free for the next step (NCL > 1: the waits' barriers order it)
  }
  STAMP_FLUSH(1, cl, j);
  // bias grads of this cluster: reduce the threads that share a unit group, one partial row per cluster
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dbL + tid * 16 + g * 4) = dbacc[g];
  __syncthreads();
  if (tid < 4 * US) {
    const int g = tid / US, ul = tid % US;
    constexpr int NTH = NE < 256 ? NE : 256;
    float s = 0.f;
    for (int th = ul / UI; th < NTH; th += QPR) s += dbL[th * 16 + g * 4 + (ul % UI)];
    a.dbpart[(size_t)(a.cl_begin + cl) * 4 * H + g * H + j * US + ul] = s;
  }
}

// The tail of a layer's backward in one launch: both weight-gradient products' split-K sums (fixed order: bitwise
// reproducible) with their `+=` into the gradient buffers, and the bias gradients from the per-cluster partials --
// instead of two reduction kernels per product and one for the bias (7 small launches per layer backward, each with
// its gap, in front of the next layer's BPTT kernel).
struct WFinishArgs {
  TbGemmDeferred g[2];
  const float* dbpart;
  int nparts, nb;
  float* db_ih;
  float* db_hh;
  int db_acc;
};

__global__ void wgrad_finish_kernel(WFinishArgs f) {
  const long n0 = f.g[0].splits ? (long)f.g[0].M * f.g[0].N : 0, n1 = f.g[1].splits ? (long)f.g[1].M * f.g[1].N : 0;
  const long total = n0 + n1 + f.nb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < n0 + n1) {
      const TbGemmDeferred& g = i < n0 ? f.g[0] : f.g[1];
      const long e = i < n0 ? i : i - n0, n = i < n0 ? n0 : n1;
      float s = 0.f;
      for (int z0 = 0; z0 < g.splits; z0 += 16) {   // sixteen slabs in flight, added in slab order (one load after the other: a
        float v[16];                                // 512-slab sum took 65-110 us at cfg1's unaligned input width)
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = z0 + u < g.splits ? g.slab[(size_t)(z0 + u) * n + e] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
      float* c = g.C + (e / g.N) * g.ldc + e % g.N;
      *c = g.accumulate ? *c + s : s;
    } else {
      const int k = (int)(i - n0 - n1);
      float s = 0.f;
      for (int p0 = 0; p0 < f.nparts; p0 += 8) {   // eight partials in flight, added in cluster order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p0 + u < f.nparts ? f.dbpart[(size_t)(p0 + u) * f.nb + k] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      if (f.db_ih) f.db_ih[k] = f.db_acc ? f.db_ih[k] + s : s;
      if (f.db_hh) f.db_hh[k] = f.db_acc ? f.db_hh[k] + s : s;
    }
  }
}

// the same, 4 elements per thread (16-byte loads, 8 slabs in flight): every N, ldc and pointer a multiple of 4 / 16
// Four neighbouring lanes share one group of 4 elements: lane p sums the slabs z = p, p + 4, ..., the four partial sums
// meet in two butterfly steps (a fixed order: bitwise reproducible).
__global__ void wgrad_finish4_kernel(WFinishArgs f) {
  const long q0 = f.g[0].splits ? (long)f.g[0].M * f.g[0].N / 4 : 0, q1 = f.g[1].splits ? (long)f.g[1].M * f.g[1].N / 4 : 0;
  const long total = (q0 + q1) * 4 + f.nb;    // a multiple of 4 lanes per element group: no group straddles a wave
  const long iters = (total + (long)gridDim.x * blockDim.x - 1) / ((long)gridDim.x * blockDim.x);
  for (long it = 0; it < iters; ++it) {       // uniform trip count: the butterfly needs all four lanes of a group
    const long i = it * (long)gridDim.x * blockDim.x + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < (q0 + q1) * 4) {
      const long eg = i >> 2;
      const int part = (int)(i & 3);
      const TbGemmDeferred& g = eg < q0 ? f.g[0] : f.g[1];
      const long e = eg < q0 ? eg : eg - q0, n4 = eg < q0 ? q0 : q1;
      const f32x4* sl = reinterpret_cast<const f32x4*>(g.slab) + e;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int z = part;
      for (; z + 12 < g.splits; z += 16) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = sl[(size_t)(z + 4 * k) * n4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[k];
      }
      for (; z < g.splits; z += 4) s += sl[(size_t)z * n4];
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], 2);
      if (part == 0) {
        const long el = e * 4;
        f32x4* c = reinterpret_cast<f32x4*>(g.C + (el / g.N) * g.ldc + el % g.N);
        *c = g.accumulate ? *c + s : s;
      }
    } else if (i < total) {
      const int k = (int)(i - (q0 + q1) * 4);
      float s = 0.f;
      for (int p0 = 0; p0 < f.nparts; p0 += 8) {   // eight partials in flight, added in cluster order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p0 + u < f.nparts ? f.dbpart[(size_t)(p0 + u) * f.nb + k] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      if (f.db_ih) f.db_ih[k] = f.db_acc ? f.db_ih[k] + s : s;
      if (f.db_hh) f.db_hh[k] = f.db_acc ? f.db_hh[k] + s : s;
    }
  }
}

int launch_wgrad_finish(const WFinishArgs& f, hipStream_t s) {
  const long n0 = f.g[0].splits ? (long)f.g[0].M * f.g[0].N : 0, n1 = f.g[1].splits ? (long)f.g[1].M * f.g[1].N : 0;
  if (n0 + n1 + f.nb <= 0) return TB_OK;
  bool vec = true;
  for (int k = 0; k < 2; ++k)
    if (f.g[k].splits && ((f.g[k].N & 3) || (f.g[k].ldc & 3) || (reinterpret_cast<uintptr_t>(f.g[k].slab) & 15) ||
                          (reinterpret_cast<uintptr_t>(f.g[k].C) & 15)))
      vec = false;
  const long total = n0 + n1 + f.nb;   // vectorised: 4 lanes per group of 4 elements -- the same count
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (vec) hipLaunchKernelGGL(wgrad_finish4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f);
  else hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f);
  TB_LAUNCH_CHECK("wgrad_finish_kernel");
  return TB_OK;
}

// dx[b][t][:] = sum over the members' partials part[m][b / rows][t][b % rows][:], in member order; part[m] is `stride` floats
// behind part[m-1].  Rows of dx from r0 = (first batch row) * T on, n4 groups of 4 floats.
__global__ void dx_reduce_kernel(const float* __restrict__ part, int nparts, long stride, long r0, long n4, int T, int in_dim,
                                 int rows, float* __restrict__ dx) {
  const int g4 = in_dim / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = r0 + i / g4;
    const int f = (int)(i % g4) * 4;
    const long b = r / T, t = r % T;
    const size_t src = (((size_t)(b / rows) * T + t) * rows + (b % rows)) * in_dim + f;
    f32x4 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = m < nparts ? *reinterpret_cast<const f32x4*>(part + (size_t)m * stride + src) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 s = v[0];
#pragma unroll
    for (int m = 1; m < 8; ++m) s += v[m];
    *reinterpret_cast<f32x4*>(dx + r * in_dim + f) = s;
  }
}

__global__ void db_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* db_ih, float* db_hh,
                                 int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
  if (db_ih) db_ih[i] = accumulate ? db_ih[i] + s : s;
  if (db_hh) db_hh[i] = accumulate ? db_hh[i] + s : s;
}


// ---- launch planning ---------------------------------------------------------------------
struct Plan {
  int H, US, KS, MT;
  int XK;              // forward: padded input width of the fused projection (32 / 64), 0 = projection by GEMM
  int NCL;
  int rows_per_cluster;
  int clusters_per_launch;
  size_t lds_fwd, lds_bwd;
};

// More than one batch tile per cluster only when the batch has more tiles than the chip has clusters.  (The
// forward kernel interleaves the tiles of a cluster, which hides a tile's hand-off under another tile's
// compute -- 9 % faster than stepping them in lockstep at B = 1024 -- but two interleaved tiles on half the
// CUs are still slower than one tile per cluster on all of them: 4.5 vs 3.4 us per step at B = 512.)
// Forward launches of layers with a narrow input (in_dim <= 64, a multiple of 4; hidden <= 256) compute the input
// projection inside the recurrent kernel.  The choice depends on (H, in_dim) only -- see lstm_fwd_kernel.
int fused_xk(int H, int in_dim) {
  // TB_FUSE_PROJ: bit mask of the hidden sizes that fuse (1: <= 64, 2: 128, 4: 256; 0 = never)
  static const int mask = []() { const char* e = getenv("TB_FUSE_PROJ"); return e ? atoi(e) : 7; }();
  if (!(mask & (H <= 64 ? 1 : (H == 128 ? 2 : 4)))) return 0;
  if (H > 256 || in_dim <= 0 || in_dim > 64 || (in_dim & 3)) return 0;
  if (H == 256 && in_dim > 32) return 0;   // 268 registers: would break the two-launches-per-CU rule of make_plan
  return in_dim <= 32 ? 32 : 64;
}

bool make_plan(int H, int B, Plan& p, bool fwd = false, int in_dim = 0) {
  int US, KS;
  switch (H) {
    case 16: US = 16; KS = 1; break;
    case 32: US = 32; KS = 1; break;
    case 64: US = 64; KS = 1; break;
    case 128: US = 32; KS = 2; break;
    case 256: US = 32; KS = 2; break;
    case 512: US = 16; KS = 4; break;
    default: return false;
  }
  // BPTT at hidden 256 with more batch tiles than the 8-member layout has clusters (B > 512 on 256 CUs: the
  // discriminator's real + fake pass): 4-member clusters, 64 units per member, W_hh^T on 256 registers.  A step costs
  // more (4.3 against 2.8 us: twice the MFMA work per CU) but one launch covers 64 tiles instead of two launches of 32
  // (0.55 against 0.75 ms at B = 1024).  Such a workgroup holds ~390 registers per lane, so NO second clustered
  // workgroup fits beside it on a CU: the launch is EXCLUSIVE (tb_lstm_exclusive) -- the caller must not have another
  // clustered launch in flight (the Python host orders it: ops.Runtime.clustered_begin).  TB_WIDE256=0 turns it off.
  static const bool wide256 = []() { const char* e = getenv("TB_WIDE256"); return !e || atoi(e) != 0; }();
  if (H == 256 && wide256 && B > 16 * (tb_num_cus() / 8)) { US = 64; KS = 1; }
  // Hidden 512: 16-member clusters (32 units per member, two k-part waves per unit group, W_hh on 256 registers like the
  // 4-member layout of hidden 256) instead of 32 members of 16 units: half the exchange fan-in, twice the clusters per
  // launch.  For EVERY batch size -- the k split decides the summation order, and a window's result must not depend
  // on the batch it sits in.  ~390 registers per lane: such a launch is EXCLUSIVE as well.  TB_WIDE512=0 turns it off.
  static const bool wide512 = []() { const char* e = getenv("TB_WIDE512"); return !e || atoi(e) != 0; }();
  if (H == 512 && wide512) { US = 32; KS = 2; }
  p.H = H; p.US = US; p.KS = KS; p.NCL = H / US;
  const int XK = fwd ? fused_xk(H, in_dim) : 0;
  p.XK = XK;
  const int cus = tb_num_cus();
  const int max_clusters = cus / p.NCL > 0 ? cus / p.NCL : 1;
  int MT = 1;
  // Tiles per cluster.  Two clustered kernels may run at once (the trainer overlaps streams), and a cluster
  // only makes progress with all its workgroups resident: every launch is at most one workgroup per CU, and
  // the variants that can meet on a full-size batch stay within half a SIMD's registers (256), so two such
  // launches always fit side by side.  BPTT at H=256 with two tiles needs 288 and is no faster than two
  // one-tile launches (6.0 vs 2 x 2.9 us/step), hence the cap of 1 there.
  // two interleaved tiles per cluster in the forward pass: 127.6 against 130.5 ms per cfg4 step (BPTT with two tiles: 112
  // against 102 ms, not built in)
  static const int mt512 = []() { const char* e = getenv("TB_MT512"); return e ? atoi(e) : 2; }();
  const int mt_cap = (US == 64 && H == 256) ? 1 : (US == 32 && H == 512) ? (fwd ? mt512 : 1) : (fwd ? ((H >= 256 || H == 64) ? 2 : 4) : ((H >= 256) ? 1 : (H == 64 ? 2 : 4)));
  const int NUG = US / 16;
  const int HK = H < 32 ? 32 : H;
  const int NCL = p.NCL;
  auto lds_f = [=](int mt) {
    size_t b = ((size_t)mt * 16 * (HK + 8) + (size_t)(KS > 1 ? NUG * KS * (KS - 1) * 256 * (4 / KS) : 0)) * 4 + 16;
    if (mt == 1 && NCL > 1 && KS == 2 && US == 32)   // STAGE buffers of lstm_fwd_kernel: pbuf [2][16][4*US+8] (not fused), sbuf [16][6*US+8]
      b += ((size_t)(XK ? 0 : 2 * 16 * (4 * US + 8)) + (size_t)16 * (6 * US + 8)) * 4;
    if (XK)   // W_ih rows [4*US][XK+8] and x tiles [2][mt*16][XK+8], bf16 hi and lo each
      b += ((size_t)4 * US + (size_t)2 * mt * 16) * (XK + 8) * 2 * 2 + (size_t)4 * US * 4;
    return b;
  };
  auto lds_b = [=](int mt) { return ((size_t)mt * 16 * (4 * US + 8) + (size_t)mt * 16 * (H + 4) + 4096) * 4 + 16; };
  // ... and within half a CU's LDS (80 KiB): a clustered variant leaves room for a second clustered workgroup
  constexpr size_t kLdsHalf = 80 * 1024;
  const size_t lds_cap = (US == 32 && H == 512) ? 2 * kLdsHalf - 1024 : kLdsHalf;   // an exclusive launch has the CU to itself
  auto fits = [=](int mt) { return NCL == 1 || (fwd ? lds_f(mt) : lds_b(mt)) <= lds_cap; };
  while (MT < mt_cap && (B + 16 * MT - 1) / (16 * MT) > max_clusters && fits(2 * MT)) MT *= 2;
  p.MT = MT;
  // Fewer VALID rows per 16-row MFMA tile when the batch has far fewer tiles than the chip has clusters (hidden 64
  // at batch 512: 32 workgroups on 256 CUs).  A step's cost there is the workgroup's own memory pipe -- 24 KB of
  // stash per step and CU: 1.39 us per step in training against 0.90 without the stash -- so the same rows are
  // spread over 2 or 4 times as many workgroups, each computing a 16-row tile of which only 8 or 4 rows exist
  // (the descriptors drop the rest).  Results per row do not depend on the tile they sit in.
  int rows = 16 * MT;
  if (MT == 1 && p.NCL == 1 && fwd) {   // (measured: no gain for BPTT or for the clustered variants)
    const char* e = getenv("TB_TILE_ROWS");     // experiment knob: lower bound 16 / 8 / 4 (16 = always whole tiles)
    const int floor_rows = (e && (atoi(e) == 16 || atoi(e) == 8)) ? atoi(e) : 4;
    while (rows > floor_rows && (B + rows / 2 - 1) / (rows / 2) <= max_clusters) rows /= 2;
  }
  p.rows_per_cluster = rows;
  p.clusters_per_launch = max_clusters;
  p.lds_fwd = lds_f(MT);
  p.lds_bwd = lds_b(MT);
  if (!fits(MT)) return false;
  return true;
}

template <int H, int US, int KS, int MT, int XK, int CMP = 1>
int launch_fwd_t(const FwdArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_kernel<H, US, KS, MT, XK, CMP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_fwd_kernel<H, US, KS, MT, XK, CMP>), dim3(grid), dim3(256), lds, s, a);
  TB_LAUNCH_CHECK("lstm_fwd_kernel");
  return TB_OK;
}

template <int H, int US, int MT, int DXK = 0>
int launch_bwd_t(const BwdArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<H, US, MT, DXK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<H, US, MT, DXK>), dim3(grid), dim3(256), lds, s, a);
  TB_LAUNCH_CHECK("lstm_bwd_kernel");
  return TB_OK;
}

#define FWD_CASE(h, us, ks, mt)                                                       \
  if (p.H == h && p.MT == mt) {                                                       \
    if (p.XK == 32) return launch_fwd_t<h, us, ks, mt, 32>(a, grid, p.lds_fwd, s);    \
    if (p.XK == 64) return launch_fwd_t<h, us, ks, mt, 64>(a, grid, p.lds_fwd, s);    \
    return launch_fwd_t<h, us, ks, mt, 0>(a, grid, p.lds_fwd, s);                     \
  }
#define FWD_CASE32(h, us, ks, mt)                                                     \
  if (p.H == h && p.MT == mt) {                                                       \
    if (p.XK == 32) return launch_fwd_t<h, us, ks, mt, 32>(a, grid, p.lds_fwd, s);    \
    if (p.XK == 0) return launch_fwd_t<h, us, ks, mt, 0>(a, grid, p.lds_fwd, s);      \
  }
#define FWD_CASE0(h, us, ks, mt) \
  if (p.H == h && p.MT == mt) return launch_fwd_t<h, us, ks, mt, 0>(a, grid, p.lds_fwd, s)
#define BWD_CASE(h, us, mt) \
  if (p.H == h && p.MT == mt) return launch_bwd_t<h, us, mt>(a, grid, p.lds_bwd, s)

int launch_fwd(const Plan& p, const FwdArgs& a, int grid, hipStream_t s) {
  FWD_CASE(16, 16, 1, 1); FWD_CASE(16, 16, 1, 2); FWD_CASE(16, 16, 1, 4);
  FWD_CASE(32, 32, 1, 1); FWD_CASE(32, 32, 1, 2); FWD_CASE(32, 32, 1, 4);
  // fused hidden-64 layers on partly filled tiles (make_plan: 8 or 4 valid rows): the compacted elementwise part
  static const int compact = []() { const char* e = getenv("TB_COMPACT"); return e ? atoi(e) : 1; }();   // A/B knob: 0 off, 2 inference only, 3 training only
  if ((compact == 1 || (compact == 2 && !a.train) || (compact == 3 && a.train)) && p.H == 64 && p.MT == 1 && p.XK && a.rows < 16) {
    if (a.rows == 8 && p.XK == 32) return launch_fwd_t<64, 64, 1, 1, 32, 2>(a, grid, p.lds_fwd, s);
    if (a.rows == 8 && p.XK == 64) return launch_fwd_t<64, 64, 1, 1, 64, 2>(a, grid, p.lds_fwd, s);
    if (a.rows == 4 && p.XK == 32) return launch_fwd_t<64, 64, 1, 1, 32, 4>(a, grid, p.lds_fwd, s);
    if (a.rows == 4 && p.XK == 64) return launch_fwd_t<64, 64, 1, 1, 64, 4>(a, grid, p.lds_fwd, s);
  }
  FWD_CASE(64, 64, 1, 1); FWD_CASE(64, 64, 1, 2);
  FWD_CASE(128, 32, 2, 1); FWD_CASE(128, 32, 2, 2); FWD_CASE(128, 32, 2, 4);
  if (p.US == 64) { FWD_CASE32(256, 64, 1, 1); }
  FWD_CASE32(256, 32, 2, 1); FWD_CASE32(256, 32, 2, 2);
  if (p.US == 32) { FWD_CASE0(512, 32, 2, 1); FWD_CASE0(512, 32, 2, 2); }
  FWD_CASE0(512, 16, 4, 1); FWD_CASE0(512, 16, 4, 2);
  return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: no kernel for H=%d MT=%d", p.H, p.MT);
}

// The input gradient inside the BPTT kernel (DXK = 32): for clustered one-tile layers with a narrow input (in <= 32, a
// multiple of 4: the discriminator) every member adds W_ih^T . dgates^T for its gate rows in the shadow of the hand-off and
// writes a partial dx; dx_reduce_kernel sums the NCL partials.  It replaces the dx GEMM, i.e. one of the three passes over the
// dgates matrix (268 - 537 MB at cfg2).  TB_DX_FUSE=0 turns it off.
constexpr int kDXK = 32;
bool dx_in_kernel(const Plan& p, int in_dim) {
  static const bool on = []() { const char* e = getenv("TB_DX_FUSE"); return !e || atoi(e) != 0; }();
  return on && p.NCL > 1 && p.NCL <= 8 && p.MT == 1 && in_dim <= kDXK && (in_dim & 3) == 0 && (p.H == 256 || p.H == 128);
}
size_t dx_extra_lds(const Plan& p) {   // second dgates tile + the W_ih^T tile (hi and lo)
  const size_t ldg = 4 * p.US + 8;
  return (size_t)p.MT * 16 * ldg * 4 + (size_t)2 * kDXK * ldg * 2 + 16;
}
size_t dx_part_bytes(const Plan& p, int B, int T, int in_dim) {
  const size_t ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  return dx_in_kernel(p, in_dim) ? tb_align_up((size_t)p.NCL * ncl_total * p.rows_per_cluster * T * in_dim * sizeof(float), 256) : 0;
}

int launch_bwd(const Plan& p, const BwdArgs& a, int grid, hipStream_t s) {
  if (a.dxpart) {
    const size_t lds = p.lds_bwd + dx_extra_lds(p);
    if (p.H == 256 && p.US == 64 && p.MT == 1) return launch_bwd_t<256, 64, 1, kDXK>(a, grid, lds, s);
    if (p.H == 256 && p.US == 32 && p.MT == 1) return launch_bwd_t<256, 32, 1, kDXK>(a, grid, lds, s);
    if (p.H == 128 && p.MT == 1) return launch_bwd_t<128, 32, 1, kDXK>(a, grid, lds, s);
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: no in-kernel input gradient for H=%d MT=%d", p.H, p.MT);
  }
  BWD_CASE(16, 16, 1); BWD_CASE(16, 16, 2); BWD_CASE(16, 16, 4);
  BWD_CASE(32, 32, 1); BWD_CASE(32, 32, 2); BWD_CASE(32, 32, 4);
  BWD_CASE(64, 64, 1); BWD_CASE(64, 64, 2);
  BWD_CASE(128, 32, 1); BWD_CASE(128, 32, 2); BWD_CASE(128, 32, 4);
  if (p.US == 64) { BWD_CASE(256, 64, 1); }
  BWD_CASE(256, 32, 1); BWD_CASE(256, 32, 2);
  if (p.US == 32) { BWD_CASE(512, 32, 1); }
  BWD_CASE(512, 16, 1);
  return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: no kernel for H=%d MT=%d", p.H, p.MT);
}

constexpr unsigned long long kTimeoutTicks = 200000000ull;   // 2 s of the 100 MHz realtime counter

// the kernels address one cluster's rows through 32-bit buffer descriptors
bool tile_fits_descriptor(const Plan& p, int T, int H) {
  return (long long)p.rows_per_cluster * T * 4 * H * 4 < 0x7FFFFF00ll;
}

// workspace carve: [hello | exchange (fwd: h ring, bwd: dh slabs) | dbpart | gemm split-k slabs]
// Nothing in it has to be initialised: the kernels arm the exchange area themselves and the handshake
// tags carry a per-launch epoch.
unsigned long long next_epoch() {
  static const unsigned long long base = []() {
    std::random_device rd;
    return ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
  }();
  static std::atomic<unsigned long long> counter{0};
  return (base + counter.fetch_add(1)) & 0x0FFFFFFFFFFFFFFFull;   // 60 bits: the tag keeps 4 for the XCD id
}

struct Carve {
  unsigned long long* hello; size_t hello_bytes;
  float* X; size_t x_bytes;
  float* dbpart;
  float* gemm_ws; size_t gemm_ws_bytes;     // two regions of gemm_ws_bytes each (backward): the two weight-gradient
  float* gemm_ws2;                          // products keep their slabs until the common finish kernel
  float* dxpart;                            // backward, in-kernel input gradient: [NCL][B][T][in] partials (or nullptr)
};

size_t carve(const Plan& p, int B, int H, int in_dim, bool bwd, char* base, size_t total, Carve* c, int T = 0) {
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  size_t off = 0;
  const int ncl_launch = ncl_total < p.clusters_per_launch ? ncl_total : p.clusters_per_launch;
  const size_t hello_bytes = p.NCL > 1 ? tb_align_up((size_t)ncl_launch * p.NCL * sizeof(unsigned long long), 256) : 0;
  // rings of 3 slots per cluster: forward rows x H floats (h_t all-gather), backward NCL x rows x H
  // (one partial of dh per producer, reduce-scatter)
  const size_t x_bytes = p.NCL > 1 ? tb_align_up((size_t)ncl_launch * 3 * (bwd ? p.NCL : 1) * (16 * p.MT) * H * sizeof(float), 256) : 0;
  const size_t dbpart_bytes = bwd ? tb_align_up((size_t)ncl_total * 4 * H * sizeof(float), 256) : 0;
  const size_t gemm_bytes = bwd ? 2 * tb_align_up(tb_gemm_ws_bytes(), 256) : 0;
  if (c) { c->hello = reinterpret_cast<unsigned long long*>(base + off); c->hello_bytes = hello_bytes; }
  off += hello_bytes;
  if (c) { c->X = reinterpret_cast<float*>(base + off); c->x_bytes = x_bytes; }
  off += x_bytes;
  if (c) c->dbpart = reinterpret_cast<float*>(base + off);
  off += dbpart_bytes;
  const size_t dx_bytes = bwd ? dx_part_bytes(p, B, T, in_dim) : 0;
  if (c) c->dxpart = dx_bytes ? reinterpret_cast<float*>(base + off) : nullptr;
  off += dx_bytes;
  if (c) {
    c->gemm_ws = reinterpret_cast<float*>(base + off);
    c->gemm_ws_bytes = total > off ? (total - off) / 2 / 256 * 256 : 0;
    c->gemm_ws2 = reinterpret_cast<float*>(base + off + c->gemm_ws_bytes);
  }
  off += gemm_bytes;
  return off;
}

}  // namespace

static int lstm_fwd_impl(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                         float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                         void* ws, size_t ws_bytes, int* err, void* stream, bool prof, long p_row = 0) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0 || T <= 0) return TB_OK;
  if (!x || !w_ih || !w_hh || !y || !ws || !err) return tb_set_error(TB_ERR_ARG, "lstm_fwd: null pointer");
  if (reinterpret_cast<uintptr_t>(ws) & 255) return tb_set_error(TB_ERR_ARG, "lstm_fwd: the workspace must be 256-byte aligned");
  if (train && !cstash) return tb_set_error(TB_ERR_ARG, "lstm_fwd: train needs a cell stash");
  Plan p;
  // p_row != 0: the pre-activations are given (one row per series row, windows overlap): no projection here, no fusion
  if (!make_plan(H, B, p, true, p_row ? 0 : in_dim))
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: hidden size %d not supported (16,32,64,128,256,512)", H);
  // the gate buffer: projection output + BPTT stash; not touched by a fused inference launch
  if (!gates && (train || !p.XK)) return tb_set_error(TB_ERR_ARG, "lstm_fwd: null gate buffer");
  Carve c;
  const size_t need = carve(p, B, H, in_dim, false, static_cast<char*>(ws), ws_bytes, &c);
  if (need > ws_bytes) return tb_set_error(TB_ERR_WORKSPACE, "lstm_fwd: workspace %zu < %zu", ws_bytes, need);
  if (!tile_fits_descriptor(p, T, H)) return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: T*H too large (%d x %d)", T, H);
  // P = x.W_ih^T + (b_ih + b_hh), written straight into the gate stash (the GEMM epilogue sums the two
  // bias vectors first, then adds)
  // -- unless the layer's input is narrow enough for the recurrent kernel to do it itself (Plan::XK)
  int rc = TB_OK;
  if (!p.XK && !p_row)
    rc = tb_gemm_launch2(x, in_dim, 0, w_ih, in_dim, 0, gates, 4L * H, b_ih, b_hh, B * T, 4 * H, in_dim, 0, 0, 1,
                         nullptr, 0, s, 0);
  if (rc) return rc;
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  // algorithmic bytes of one layer forward: read x, write h, write the 5h stash
  for (int c0 = 0; c0 < ncl_total; c0 += p.clusters_per_launch) {
    const int ncl = (ncl_total - c0) < p.clusters_per_launch ? (ncl_total - c0) : p.clusters_per_launch;
    FwdArgs a;
    a.P = gates; a.Whh = w_hh; a.y = y; a.cst = cstash;
    a.x = x; a.Wih = w_ih; a.b_ih = b_ih; a.b_hh = b_hh; a.in_dim = in_dim;
    a.ring = c.X;
    a.hello = c.hello;
    a.epoch = next_epoch();
    a.err = err;
    a.B = B; a.T = T;
    a.p_row = p_row;
    a.b_begin = c0 * p.rows_per_cluster;
    a.rows = p.rows_per_cluster;
    a.nclusters = ncl;
    a.train = train;
    a.timeout_ticks = kTimeoutTicks;
    // timed region = the recurrent kernel(s): the event goes in after the fill that arms the exchange area
    if (prof && c0 == 0) tb_prof_begin(TB_K_LSTM_FWD, s, 4.0 * B * T * (train ? (in_dim + 6.0 * H) : (in_dim + 1.0 * H)));
    rc = launch_fwd(p, a, ncl * p.NCL, s);
    if (rc) return rc;
  }
  if (prof) tb_prof_end(TB_K_LSTM_FWD, s);
  return TB_OK;
}


extern "C" {

#ifdef TB_STAMPS
int tb_debug_stamps(unsigned long long* out) {   // diagnostic build only
  return tb_check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(tb_stamps), sizeof(unsigned long long) * 2 * 4 * 12), "stamps");
}
#endif

size_t tb_lstm_workspace_bytes(int B, int T, int in_dim, int H, int backward) {
  Plan p;
  if (!make_plan(H, B, p, backward == 0, in_dim)) return 0;
  return carve(p, B, H, in_dim, backward != 0, nullptr, 0, nullptr, T);
}

int tb_lstm_exclusive(int B, int in_dim, int H, int backward) {
  Plan p;
  if (!make_plan(H, B, p, backward == 0, in_dim)) return 0;
  return ((p.H == 256 && p.US == 64) || (p.H == 512 && p.US == 32)) ? 1 : 0;
}

int tb_lstm_supported(int H) {
  Plan p;
  return make_plan(H, 16, p) ? 1 : 0;
}

int tb_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                void* ws, size_t ws_bytes, int* err, void* stream) {
  return lstm_fwd_impl(x, w_ih, w_hh, b_ih, b_hh, y, gates, cstash, B, T, in_dim, H, train, ws, ws_bytes, err, stream, true);
}

// Inference over OVERLAPPING windows of a series (the cleaning / prediction pass, cleaner.py:67-68 over the stride-`stride`
// windows of dataset.py:207-212): window b's pre-activations of step t are row first_row + b * stride + t of `pseries`
// -- x.W_ih^T + b_ih + b_hh computed ONCE per series row by the caller (tb_linear_act_fwd) instead of once per (window, step):
// forecast_len times fewer projection flops, and the windows' encoder rows are never gathered.  Read-only on pseries.
int tb_lstm_fwd_windows(const float* pseries, long series_rows, long first_row, long stride, const float* w_hh, float* y,
                        int B, int T, int H, void* ws, size_t ws_bytes, int* err, void* stream) {
  if (B <= 0 || T <= 0) return TB_OK;
  if (!pseries || stride < 1 || first_row < 0 || first_row + (long)(B - 1) * stride + T > series_rows)
    return tb_set_error(TB_ERR_ARG, "lstm_fwd_windows: windows leave the series");
  if ((long)stride * 4 * H * (long)16 * 4 >= 0x7FFFFF00ll) return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd_windows: stride too large");
  // x / w_ih are not read (no projection); the gate buffer is the series projection, read-only (train = 0)
  return lstm_fwd_impl(pseries, pseries, w_hh, nullptr, nullptr, y, const_cast<float*>(pseries) + first_row * 4 * H, nullptr, B, T,
                       0, H, 0, ws, ws_bytes, err, stream, true, stride * 4L * H);
}

// Looks up the static scratch of the current device (tb_init calls it, so the lookup is not on a caller's
// first launch; tb_lstm_fwd_static repeats it if tb_init was skipped).
int tb_lstm_static_init(void) {
  void* p = nullptr;
  return static_scratch(&p);
}

size_t tb_lstm_static_scratch_bytes(void) { return kStaticScratchBytes; }

// tb_lstm_fwd without a workspace argument and without any allocation: the exchange area lives in the
// library's static scratch.  Same kernels, same results.  Calls that overlap on one GPU (two streams, two
// threads) share that GPU's array and race; the array keeps the last call's contents, which is harmless: the
// kernels arm their exchange area themselves and the handshake tags carry a per-launch epoch.
int tb_lstm_fwd_static(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                       int* err, void* stream) {
  if (B <= 0 || T <= 0) return TB_OK;
  Plan p;
  if (!make_plan(H, B, p, true, in_dim))
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd_static: hidden size %d not supported (16,32,64,128,256,512)", H);
  const size_t need = carve(p, B, H, in_dim, false, nullptr, 0, nullptr);
  if (need > kStaticScratchBytes)
    return tb_set_error(TB_ERR_WORKSPACE, "lstm_fwd_static: this shape needs %zu bytes of exchange area, the static scratch "
                        "holds %zu; use tb_lstm_fwd with a caller workspace", need, kStaticScratchBytes);
  void* ptr = nullptr;
  int rc = static_scratch(&ptr);
  if (rc) return rc;
  return lstm_fwd_impl(x, w_ih, w_hh, b_ih, b_hh, y, gates, cstash, B, T, in_dim, H, train, ptr, kStaticScratchBytes, err,
                       stream, false);
}

static int lstm_bwd_impl(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                         float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                         float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, int* err,
                         void* stream, int dx_row_begin);

int tb_lstm_bwd(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, int* err,
                void* stream) {
  return lstm_bwd_impl(dy, x, w_ih, w_hh, y, gates, cstash, dx, dw_ih, dw_hh, db_ih, db_hh, B, T, in_dim, H, accumulate, ws,
                       ws_bytes, err, stream, 0);
}

// tb_lstm_bwd with dx computed for the batch rows [dx_row_begin, B) only; the rows before are left untouched.
int tb_lstm_bwd_rows(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                     float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                     float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, int* err,
                     void* stream, int dx_row_begin) {
  if (dx_row_begin < 0 || dx_row_begin > B) return tb_set_error(TB_ERR_ARG, "lstm_bwd_rows: dx_row_begin out of range");
  return lstm_bwd_impl(dy, x, w_ih, w_hh, y, gates, cstash, dx, dw_ih, dw_hh, db_ih, db_hh, B, T, in_dim, H, accumulate, ws,
                       ws_bytes, err, stream, dx_row_begin);
}

}  // extern "C"

static int lstm_bwd_impl(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                         float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                         float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, int* err,
                         void* stream, int dx_row_begin) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0 || T <= 0) return TB_OK;
  if (!dy || !x || !w_ih || !w_hh || !y || !gates || !cstash || !ws || !err)
    return tb_set_error(TB_ERR_ARG, "lstm_bwd: null pointer");
  if (reinterpret_cast<uintptr_t>(ws) & 255) return tb_set_error(TB_ERR_ARG, "lstm_bwd: the workspace must be 256-byte aligned");
  Plan p;
  if (!make_plan(H, B, p))
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: hidden size %d not supported (16,32,64,128,256,512)", H);
  Carve c;
  const size_t need = carve(p, B, H, in_dim, true, static_cast<char*>(ws), ws_bytes, &c, T);
  if (need > ws_bytes) return tb_set_error(TB_ERR_WORKSPACE, "lstm_bwd: workspace %zu < %zu", ws_bytes, need);
  // input gradient inside the kernel (see dx_in_kernel): only when somebody wants it, and within the co-residency rule
  const bool want_dx = dx && dx_row_begin < B;
  const bool excl = (p.H == 256 && p.US == 64) || (p.H == 512 && p.US == 32);
  const bool dxk = want_dx && c.dxpart && (excl || p.lds_bwd + dx_extra_lds(p) <= 80 * 1024);
  if (!tile_fits_descriptor(p, T, H)) return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: T*H too large (%d x %d)", T, H);
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  int rc;
  // algorithmic bytes of one layer backward: read 5h stash + dy, read x, write dx
  for (int c0 = 0; c0 < ncl_total; c0 += p.clusters_per_launch) {
    const int ncl = (ncl_total - c0) < p.clusters_per_launch ? (ncl_total - c0) : p.clusters_per_launch;
    BwdArgs a;
    a.G = gates; a.cst = cstash; a.dy = dy; a.Whh = w_hh;
    a.X = c.X;
    a.dbpart = c.dbpart;
    a.hello = c.hello;
    a.err = err;
    a.B = B; a.T = T;
    a.b_begin = c0 * p.rows_per_cluster;
    a.rows = p.rows_per_cluster;
    a.cl_begin = c0;
    a.epoch = next_epoch();
    a.nclusters = ncl;
    a.timeout_ticks = kTimeoutTicks;
    a.Wih = w_ih; a.in_dim = in_dim; a.dxpart = dxk ? c.dxpart : nullptr; a.dx_b_begin = dx_row_begin; a.ncl_total = ncl_total;
    // dgates go back to memory for the weight-gradient products (here, or tb_lstm_wgrad later: then the caller asks for the
    // bias gradients here) and for the dx GEMM; a pass that wants the input gradient only and forms it in the kernel -- the
    // generator phase's way through D -- leaves `gates` alone (268 MB of stores per launch at cfg2)
    a.store_dg = (dw_ih || dw_hh || db_ih || db_hh || (want_dx && !dxk)) ? 1 : 0;
    if (c0 == 0) tb_prof_begin(TB_K_LSTM_BWD, s, 4.0 * B * T * (2.0 * in_dim + 6.0 * H));
    rc = launch_bwd(p, a, ncl * p.NCL, s);
    if (rc) return rc;
  }
  tb_prof_end(TB_K_LSTM_BWD, s);
  const int M = B * T;
  if (dxk) {   // dx = the sum of the members' partials
    const long r0 = (long)dx_row_begin * T, n4 = ((long)M - r0) * in_dim / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dx_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, c.dxpart, p.NCL,
                       (long)ncl_total * p.rows_per_cluster * T * in_dim, r0, n4, T, in_dim, p.rows_per_cluster, dx);
    TB_LAUNCH_CHECK("dx_reduce_kernel");
  } else if (want_dx) {   // dx = dG . W_ih            [M,4H].[4H,in]   (first: the layer below waits for it)
    const long r0 = (long)dx_row_begin * T;   // only the rows whose input gradient somebody reads
    rc = tb_gemm_launch(gates + r0 * 4 * H, 4L * H, 0, w_ih, in_dim, 1, dx + r0 * in_dim, in_dim, nullptr, (int)(M - r0), in_dim,
                        4 * H, 0, 0, 1, c.gemm_ws, c.gemm_ws_bytes, s, 0);
    if (rc) return rc;
  }
  WFinishArgs f;
  f.g[0].splits = f.g[1].splits = 0;
  if (dw_ih) {  // dW_ih = dG^T . x        [4H,M].[M,in]
    rc = tb_gemm_launch_deferred(gates, 4L * H, 1, x, in_dim, 1, dw_ih, in_dim, 4 * H, in_dim, M, 0, 1, c.gemm_ws,
                                 c.gemm_ws_bytes, s, accumulate, &f.g[0]);
    if (rc) return rc;
  }
  if (dw_hh) {  // dW_hh = dG^T . h_prev,  h_prev[b,t] = y[b,t-1], 0 at t = 0
    rc = tb_gemm_launch_deferred(gates, 4L * H, 1, y, H, 1, dw_hh, H, 4 * H, H, M, 1, T, c.gemm_ws2, c.gemm_ws_bytes, s,
                                 accumulate, &f.g[1]);
    if (rc) return rc;
  }
  const bool db = db_ih || db_hh;
  f.dbpart = c.dbpart; f.nparts = ncl_total; f.nb = db ? 4 * H : 0;
  f.db_ih = db_ih; f.db_hh = db_hh; f.db_acc = accumulate;
  return launch_wgrad_finish(f, s);
}

extern "C" {

size_t tb_lstm_wgrad_workspace_bytes(int in_dim, int H) {
  (void)in_dim; (void)H;
  return 2 * tb_align_up(tb_gemm_ws_bytes(), 256);
}

// The two weight-gradient products of tb_lstm_bwd on their own: dW_ih = dG^T.x, dW_hh = dG^T.h_prev.  A caller
// that passes dw_ih = dw_hh = NULL to tb_lstm_bwd can run them here on ANOTHER stream (after an event behind
// tb_lstm_bwd): nothing of the backward chain depends on them -- only dx feeds the layer below -- and beside
// the latency-bound BPTT kernel of that layer they use HBM bandwidth that would otherwise idle.
int tb_lstm_wgrad(const float* x, const float* y, const float* dgates, float* dw_ih, float* dw_hh, int B, int T,
                  int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0 || T <= 0) return TB_OK;
  if (!x || !y || !dgates || !ws) return tb_set_error(TB_ERR_ARG, "lstm_wgrad: null pointer");
  if (ws_bytes < tb_lstm_wgrad_workspace_bytes(in_dim, H)) return tb_set_error(TB_ERR_WORKSPACE, "lstm_wgrad: workspace too small");
  const int M = B * T;
  int rc;
  const size_t half = ws_bytes / 2 / 256 * 256;
  WFinishArgs f;
  f.g[0].splits = f.g[1].splits = 0;
  if (dw_ih) {
    rc = tb_gemm_launch_deferred(dgates, 4L * H, 1, x, in_dim, 1, dw_ih, in_dim, 4 * H, in_dim, M, 0, 1,
                                 static_cast<float*>(ws), half, s, accumulate, &f.g[0]);
    if (rc) return rc;
  }
  if (dw_hh) {
    rc = tb_gemm_launch_deferred(dgates, 4L * H, 1, y, H, 1, dw_hh, H, 4 * H, H, M, 1, T,
                                 reinterpret_cast<float*>(static_cast<char*>(ws) + half), half, s, accumulate, &f.g[1]);
    if (rc) return rc;
  }
  f.dbpart = nullptr; f.nparts = 0; f.nb = 0; f.db_ih = f.db_hh = nullptr; f.db_acc = 0;
  return launch_wgrad_finish(f, s);
}

}  // extern "C"
