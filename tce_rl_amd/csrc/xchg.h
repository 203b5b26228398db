// One-shot gradient exchange between the env shards of one node, inside the
// kernels that finish an epoch (no separate collective launch, no host round
// trip): SURVEY 5 / 8e -- the reference has no counterpart (no collectives under
// mprl/); the sum over ranks stands for the global-batch `.mean()` of its losses
// (mprl/rl/agent/temporal_correlated_agent.py:716,736).
//
// Every rank owns ONE peer-visible buffer (uncached device memory, exported
// with hipIpcGetMemHandle and mapped by every peer): two slots of
// {block flags | data}.  A collective with sequence number `seq` uses slot
// seq & 1.  A workgroup that owns elements [e0, e1) of the flat gradient
//   1. stores its locally reduced values into its OWN data slot (system-scope
//      stores), fences, and sets its own flag[block] = seq (release);
//   2. waits until flag[block] of every peer shows seq (acquire, bounded by a
//      wall-clock limit: on expiry the status word is set and the kernel goes
//      on, so a dead peer can never hang the grid);
//   3. reads the peers' values over xGMI and adds them IN RANK ORDER 0 .. W-1,
//      so every rank forms bit-identical sums (= a rank-ordered all-reduce).
// Only the same-numbered workgroup of the same launch on the peers is waited
// for: no grid-wide barrier.  Slot reuse is safe with two slots: a rank writes
// slot s for seq + 2 only after its seq + 1 kernel has seen the peers' seq + 1
// flags, i.e. after every peer's seq kernel (the last reader of slot s) has
// completed (launches of one exchange are in stream order on every rank).
// Messages are 20 - 640 KB: latency-bound, which is why they ride inside the
// finish kernels instead of a ring collective of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int XCHG_MAX_WORLD = 8;          // one node
constexpr int XCHG_MAX_BLOCKS = 4096;      // flags per slot
constexpr int64_t XCHG_FLAG_BYTES = (int64_t)XCHG_MAX_BLOCKS * 4;

// What a kernel gets (by value).  world == 1 (or base[rank] == nullptr): no
// exchange, xchg_* return their input.
struct XchgView {
  char* base[XCHG_MAX_WORLD];      // every rank's buffer as mapped in THIS process
  int rank, world;
  unsigned seq;                    // sequence number of this collective (>= 1)
  int64_t flag_off, data_off;      // byte offsets of this collective's slot
  unsigned long long limit;        // wall_clock64 ticks (100 MHz) a wait may take
  int* status;                     // host-mapped word: 0 ok, 1 + peer that timed out
  double* partial;                 // [XCHG_MAX_BLOCKS] per-workgroup scratch (|g|^2 partial sums)
  // wait telemetry (device memory, 4 words): [0] += ticks workgroup 0 spent
  // waiting for its slowest peer, [1] = max over collectives of that, [2] += 1
  // per collective, [3] scratch.  100 MHz ticks; read by tce_xchg_wait_stats.
  unsigned long long* wait;
};

inline XchgView xchg_none() {
  XchgView v;
  for (int i = 0; i < XCHG_MAX_WORLD; ++i) v.base[i] = nullptr;
  v.rank = 0; v.world = 1; v.seq = 0; v.flag_off = v.data_off = 0; v.limit = 0;
  v.status = nullptr;
  v.partial = nullptr;
  v.wait = nullptr;
  return v;
}

// host side (csrc/xchg.hip): the view of the NEXT collective of exchange `x`
// moving `bytes` of data with `blocks` flagging workgroups; nullptr x -> none.
// Returns non-zero (tce_last_error) when the message does not fit.
int xchg_next(void* x, int64_t bytes, int blocks, XchgView* out);

#ifdef __HIPCC__
__device__ inline bool xchg_on(const XchgView& X) { return X.world > 1; }

template <typename real> struct XchgBits;
template <> struct XchgBits<float> {
  typedef unsigned type;
  static __device__ inline unsigned enc(float v) { return __float_as_uint(v); }
  static __device__ inline float dec(unsigned v) { return __uint_as_float(v); }
};
template <> struct XchgBits<double> {
  typedef unsigned long long type;
  static __device__ inline unsigned long long enc(double v) {
    return (unsigned long long)__double_as_longlong(v);
  }
  static __device__ inline double dec(unsigned long long v) {
    return __longlong_as_double((long long)v);
  }
};

// 1. store one locally reduced value of element e into the own slot
template <typename real>
__device__ inline void xchg_put(const XchgView& X, int64_t e, real v) {
  typedef typename XchgBits<real>::type bits;
  bits* mine = reinterpret_cast<bits*>(X.base[X.rank] + X.data_off);
  __hip_atomic_store(mine + e, XchgBits<real>::enc(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// 2. all of this workgroup's xchg_put are done: publish flag[block], wait for
// the peers' flag[block].  Every thread of the workgroup must call it.
__device__ inline void xchg_sync(const XchgView& X, int block) {
  __threadfence_system();
  __syncthreads();
  const int t = threadIdx.x;
  if (t == 0) {
    unsigned* f = reinterpret_cast<unsigned*>(X.base[X.rank] + X.flag_off) + block;
    __hip_atomic_store(f, X.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t < X.world && t != X.rank) {
    const unsigned* f = reinterpret_cast<const unsigned*>(X.base[t] + X.flag_off) + block;
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - X.seq) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && wall_clock64() - t0 > X.limit) {
        if (X.status)
          __hip_atomic_store(X.status, 1 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    // how long the slowest peer kept workgroup 0 waiting (one straggler's
    // jitter is everyone's: the N > 1 line reports it per exchange)
    if (block == 0 && X.wait) {
      const unsigned long long dt = wall_clock64() - t0;
      atomicMax(X.wait + 3, dt);           // scratch: max over the peers of THIS collective
    }
  }
  if (block == 0 && X.wait) {
    __syncthreads();
    if (t == 0) {
      const unsigned long long w = atomicExch(X.wait + 3, 0ull);
      X.wait[0] += w;
      if (w > X.wait[1]) X.wait[1] = w;
      X.wait[2] += 1;
    }
  }
  __syncthreads();
  __threadfence_system();
}

// 3. the sum over ranks of element e in rank order; `mine` = what this rank put.
// All peers' loads are issued before the first add (a run-time loop over the
// ranks would make them W - 1 dependent round trips over xGMI).
template <typename real>
__device__ inline real xchg_get(const XchgView& X, int64_t e, real mine) {
  typedef typename XchgBits<real>::type bits;
  real v[XCHG_MAX_WORLD];
#pragma unroll
  for (int r = 0; r < XCHG_MAX_WORLD; ++r) {
    v[r] = mine;
    if (r < X.world && r != X.rank) {
      const bits* p = reinterpret_cast<const bits*>(X.base[r] + X.data_off) + e;
      v[r] = XchgBits<real>::dec(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
  }
  real s = v[0];
#pragma unroll
  for (int r = 1; r < XCHG_MAX_WORLD; ++r)
    if (r < X.world) s += v[r];
  return s;
}

// The three steps for a workgroup whose threads each own at most ONE element
// (own: this thread has one; e: its index).  Every thread must call it.
template <typename real>
__device__ inline real xchg_sum1(const XchgView& X, int block, bool own, int64_t e, real v) {
  if (!xchg_on(X)) return v;
  if (own) xchg_put<real>(X, e, v);
  xchg_sync(X, block);
  return own ? xchg_get<real>(X, e, v) : v;
}
#endif  // __HIPCC__
