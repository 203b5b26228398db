// Host side of the one-shot gradient exchange (csrc/xchg.h): the peer-visible
// buffer of one rank, its export / mapping through HIP IPC, the sequence
// numbers, and the two stand-alone collectives built on the device helpers --
// an in-place rank-ordered sum all-reduce and "all-reduce + flat Adam" as ONE
// launch (what the sharded update issued as torch.distributed.all_reduce +
// tce_adam_once_*).  The epoch kernels of mlp / mlpw / smlp / pmlp / objective
// use the same helpers inside their own finish kernels.
#include "common.h"
#include "xchg.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct Xchg {
  int rank = 0, world = 1;
  char* local = nullptr;
  int64_t cap = 0;                       // data bytes per slot
  char* peer[XCHG_MAX_WORLD] = {};
  bool opened[XCHG_MAX_WORLD] = {};      // mapped with hipIpcOpenMemHandle
  unsigned seq = 0;
  int* status_host = nullptr;
  int* status_dev = nullptr;
  unsigned long long limit = 0;
  int64_t n_coll = 0, n_bytes = 0;
  double* partial = nullptr;             // scratch of xchg_adam_kernel
  unsigned* ticket = nullptr;
  unsigned long long* wait = nullptr;    // telemetry words of xchg_sync (csrc/xchg.h)
};

#define X_HIP(call)                                                     \
  do {                                                                  \
    hipError_t e__ = (call);                                            \
    if (e__ != hipSuccess) {                                            \
      tce_set_error(hipGetErrorString(e__));                            \
      return 2;                                                         \
    }                                                                   \
  } while (0)

constexpr int XA_BT = 1024, XA_EPT = 4, XA_PER = XA_BT * XA_EPT;

// dst [n] = sum over ranks of src [n], rank order (dst may be src)
template <typename real>
__global__ __launch_bounds__(XA_BT) void xchg_allreduce_kernel(XchgView X, const real* src,
                                                               real* dst, int64_t n) {
  const int64_t i0 = (int64_t)blockIdx.x * XA_PER;
  real mine[XA_EPT];
#pragma unroll
  for (int q = 0; q < XA_EPT; ++q) {
    const int64_t i = i0 + threadIdx.x + q * XA_BT;
    mine[q] = i < n ? src[i] : real(0);
    if (i < n && xchg_on(X)) xchg_put<real>(X, i, mine[q]);
  }
  if (!xchg_on(X)) {
    if (dst != src) {
#pragma unroll
      for (int q = 0; q < XA_EPT; ++q) {
        const int64_t i = i0 + threadIdx.x + q * XA_BT;
        if (i < n) dst[i] = mine[q];
      }
    }
    return;
  }
  xchg_sync(X, blockIdx.x);
#pragma unroll
  for (int q = 0; q < XA_EPT; ++q) {
    const int64_t i = i0 + threadIdx.x + q * XA_BT;
    if (i < n) dst[i] = xchg_get<real>(X, i, mine[q]);
  }
}

// all-reduce (sum, rank order; the sum stays in grad) + the Adam step of
// adam_once_kernel without clipping (the factor on the gradient is gscale) in
// ONE launch: every workgroup exchanges and applies its own slice; the norm of
// the record comes from per-workgroup partial sums added in workgroup order by
// the last one to finish.
template <typename real>
__global__ __launch_bounds__(XA_BT) void xchg_adam_kernel(
    XchgView X, real* __restrict__ p, real* __restrict__ grad, real* __restrict__ m,
    real* __restrict__ v, int64_t n, real* __restrict__ state, real* __restrict__ norms_out,
    real step, real lr, real b1, real b2, real eps, real wd, real gscale,
    double* __restrict__ partial, unsigned* __restrict__ ticket) {
  __shared__ real red[16];
  __shared__ int last_s;
  const int64_t i0 = (int64_t)blockIdx.x * XA_PER;
  real g[XA_EPT];
#pragma unroll
  for (int q = 0; q < XA_EPT; ++q) {
    const int64_t i = i0 + threadIdx.x + q * XA_BT;
    g[q] = i < n ? grad[i] : real(0);
    if (i < n && xchg_on(X)) xchg_put<real>(X, i, g[q]);
  }
  if (xchg_on(X)) {
    xchg_sync(X, blockIdx.x);
#pragma unroll
    for (int q = 0; q < XA_EPT; ++q) {
      const int64_t i = i0 + threadIdx.x + q * XA_BT;
      if (i < n) {
        g[q] = xchg_get<real>(X, i, g[q]);
        grad[i] = g[q];
      }
    }
  }
  real step_size, bc2s;
  adam_coef(lr, b1, b2, step, step_size, bc2s);
  real sq = 0;
#pragma unroll
  for (int q = 0; q < XA_EPT; ++q) {
    const int64_t i = i0 + threadIdx.x + q * XA_BT;
    if (i >= n) continue;
    sq += g[q] * g[q];
    real w = p[i], mi = m[i], vi = v[i];
    adam_elem(g[q] * gscale, w, mi, vi, b1, b2, eps, wd, step_size, bc2s);
    m[i] = mi;
    v[i] = vi;
    p[i] = w;
  }
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (double)sq;
    __threadfence();
    last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last_s || threadIdx.x != 0) return;
  __threadfence();
  double tot = 0;
  for (unsigned b = 0; b < gridDim.x; ++b) tot += __hip_atomic_load(partial + b, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT);
  const real before = (real)sqrt(tot) * gscale;
  state[0] = step;
  state[1] = before;
  state[2] = before;
  state[3] = gscale;
  if (norms_out) { norms_out[0] = before; norms_out[1] = before; }
  *ticket = 0u;
}

// out [world][n] = every rank's mine [n] (one workgroup: the small per-step
// statistics -- (count, mean, M2) triples, column sums, the critic split)
template <typename real>
__global__ __launch_bounds__(XA_BT) void xchg_allgather_kernel(XchgView X,
                                                               const real* __restrict__ mine,
                                                               real* __restrict__ out, int n) {
  const int world = X.world < 1 ? 1 : X.world;
  if (xchg_on(X)) {
    for (int i = threadIdx.x; i < n; i += XA_BT) xchg_put<real>(X, i, mine[i]);
    xchg_sync(X, 0);
  }
  typedef typename XchgBits<real>::type bits;
  for (int e = threadIdx.x; e < world * n; e += XA_BT) {
    const int r = e / n, i = e - r * n;
    real v;
    if (r == X.rank || !xchg_on(X)) {
      v = mine[i];
    } else {
      const bits* p = reinterpret_cast<const bits*>(X.base[r] + X.data_off) + i;
      v = XchgBits<real>::dec(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    out[e] = v;
  }
}

template <typename real>
int xchg_allreduce(void* x, const real* src, real* dst, int64_t n, hipStream_t st) {
  TCE_CHECK_ARG(x && src && dst && n > 0, "xchg_allreduce: null exchange / buffer, n <= 0");
  const int blocks = (int)ceil_div(n, XA_PER);
  XchgView X;
  if (xchg_next(x, n * (int64_t)sizeof(real), blocks, &X)) return 1;
  hipLaunchKernelGGL(xchg_allreduce_kernel<real>, dim3(blocks), dim3(XA_BT), 0, st, X, src, dst,
                     n);
  TCE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int xchg_next(void* xv, int64_t bytes, int blocks, XchgView* out) {
  *out = xchg_none();
  if (!xv) return 0;
  Xchg* x = static_cast<Xchg*>(xv);
  TCE_CHECK_ARG(bytes <= x->cap && blocks <= XCHG_MAX_BLOCKS,
                "xchg: the message does not fit the exchange buffer (tce_xchg_create max_bytes)");
  for (int r = 0; r < x->world; ++r)
    TCE_CHECK_ARG(x->peer[r] != nullptr, "xchg: a peer is not connected (tce_xchg_connect)");
  x->seq += 1;
  x->n_coll += 1;
  x->n_bytes += bytes;
  const int slot = (int)(x->seq & 1u);
  for (int r = 0; r < x->world; ++r) out->base[r] = x->peer[r];
  out->rank = x->rank;
  out->world = x->world;
  out->seq = x->seq;
  out->flag_off = slot * XCHG_FLAG_BYTES;
  out->data_off = 2 * XCHG_FLAG_BYTES + slot * x->cap;
  out->limit = x->limit;
  out->status = x->status_dev;
  out->partial = x->partial;
  out->wait = x->wait;
  return 0;
}

extern "C" {

int tce_xchg_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int tce_xchg_create(int rank, int world, int64_t max_bytes, void** out) {
  TCE_CHECK_ARG(out && world >= 1 && world <= XCHG_MAX_WORLD && rank >= 0 && rank < world &&
                    max_bytes > 0,
                "xchg_create: 1 <= world <= 8, 0 <= rank < world, max_bytes > 0");
  Xchg* x = new Xchg();
  x->rank = rank;
  x->world = world;
  x->cap = (max_bytes + 255) / 256 * 256;
  const size_t total = (size_t)(2 * XCHG_FLAG_BYTES + 2 * x->cap);
  void* p = nullptr;
  // uncached device memory: stores of one device become visible to the others
  // inside a running kernel (what RCCL allocates its own buffers as)
  hipError_t e = hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    delete x;
    tce_set_error(hipGetErrorString(e));
    return 2;
  }
  x->local = static_cast<char*>(p);
  x->peer[rank] = x->local;
  X_HIP(hipMemset(p, 0, total));
  X_HIP(hipHostMalloc((void**)&x->status_host, 64, hipHostMallocMapped));
  x->status_host[0] = 0;
  X_HIP(hipHostGetDevicePointer((void**)&x->status_dev, x->status_host, 0));
  X_HIP(hipMalloc((void**)&x->partial, sizeof(double) * XCHG_MAX_BLOCKS));
  X_HIP(hipMalloc((void**)&x->ticket, 64));
  X_HIP(hipMemset(x->ticket, 0, 64));
  X_HIP(hipMalloc((void**)&x->wait, 64));
  X_HIP(hipMemset(x->wait, 0, 64));
  X_HIP(hipDeviceSynchronize());
  const char* ms = getenv("TCE_XCHG_TIMEOUT_MS");
  const double lim_ms = ms ? atof(ms) : 20000.0;
  x->limit = (unsigned long long)(lim_ms * 1e5);          // 100 MHz ticks
  *out = x;
  return 0;
}

int tce_xchg_export(void* xv, void* handle_out) {
  TCE_CHECK_ARG(xv && handle_out, "xchg_export: null argument");
  Xchg* x = static_cast<Xchg*>(xv);
  hipIpcMemHandle_t h;
  X_HIP(hipIpcGetMemHandle(&h, x->local));
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

// handles: world x tce_xchg_handle_bytes() in rank order (the own entry is skipped)
int tce_xchg_connect(void* xv, const void* handles) {
  TCE_CHECK_ARG(xv && handles, "xchg_connect: null argument");
  Xchg* x = static_cast<Xchg*>(xv);
  const char* hb = static_cast<const char*>(handles);
  for (int r = 0; r < x->world; ++r) {
    if (r == x->rank || x->peer[r]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, hb + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    X_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    x->peer[r] = static_cast<char*>(p);
    x->opened[r] = true;
  }
  return 0;
}

// ranks that live in ONE process (tests): the peer's buffer by its pointer
int tce_xchg_connect_local(void* xv, int peer_rank, void* peer_x) {
  TCE_CHECK_ARG(xv && peer_x && peer_rank >= 0 && peer_rank < XCHG_MAX_WORLD,
                "xchg_connect_local: bad argument");
  Xchg* x = static_cast<Xchg*>(xv);
  Xchg* y = static_cast<Xchg*>(peer_x);
  TCE_CHECK_ARG(peer_rank < x->world && y->rank == peer_rank && y->cap == x->cap,
                "xchg_connect_local: the peer is not rank `peer_rank` of an equal exchange");
  x->peer[peer_rank] = y->local;
  return 0;
}

int tce_xchg_destroy(void* xv) {
  if (!xv) return 0;
  Xchg* x = static_cast<Xchg*>(xv);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < x->world; ++r)
    if (x->opened[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
  if (x->local) (void)hipFree(x->local);
  if (x->status_host) (void)hipHostFree(x->status_host);
  if (x->partial) (void)hipFree(x->partial);
  if (x->ticket) (void)hipFree(x->ticket);
  if (x->wait) (void)hipFree(x->wait);
  delete x;
  return 0;
}

// 0: every wait so far was answered; 1 + r: a wait for rank r ran into the limit
int tce_xchg_status(void* xv) {
  if (!xv) return 0;
  return __atomic_load_n(static_cast<Xchg*>(xv)->status_host, __ATOMIC_RELAXED);
}

int tce_xchg_set_timeout_ms(void* xv, double ms) {
  TCE_CHECK_ARG(xv && ms > 0, "xchg_set_timeout_ms: bad argument");
  static_cast<Xchg*>(xv)->limit = (unsigned long long)(ms * 1e5);
  return 0;
}

// collectives issued on this exchange and their payload bytes (per rank)
int tce_xchg_counters(void* xv, int64_t* collectives, int64_t* bytes) {
  TCE_CHECK_ARG(xv && collectives && bytes, "xchg_counters: null argument");
  Xchg* x = static_cast<Xchg*>(xv);
  *collectives = x->n_coll;
  *bytes = x->n_bytes;
  return 0;
}

// How long workgroup 0 of this rank's collectives waited for its slowest peer
// since the last reset: sum and maximum in microseconds, number of collectives.
// Waits for the device (a blocking copy of the four telemetry words).
int tce_xchg_wait_stats(void* xv, double* total_us, double* max_us, int64_t* collectives,
                        int reset) {
  TCE_CHECK_ARG(xv && total_us && max_us && collectives, "xchg_wait_stats: null argument");
  Xchg* x = static_cast<Xchg*>(xv);
  unsigned long long w[4] = {0, 0, 0, 0};
  X_HIP(hipMemcpy(w, x->wait, sizeof(w), hipMemcpyDeviceToHost));
  *total_us = (double)w[0] * 1e-2;                          // 100 MHz ticks
  *max_us = (double)w[1] * 1e-2;
  *collectives = (int64_t)w[2];
  if (reset) X_HIP(hipMemset(x->wait, 0, 64));
  return 0;
}

int tce_xchg_allgather_f64(void* x, const double* mine, double* out, int64_t n, void* stream) {
  TCE_CHECK_ARG(x && mine && out && n > 0 && n <= (1 << 20),
                "xchg_allgather: null exchange / buffer, n outside [1, 2^20]");
  XchgView X;
  if (xchg_next(x, n * (int64_t)sizeof(double), 1, &X)) return 1;
  hipLaunchKernelGGL(xchg_allgather_kernel<double>, dim3(1), dim3(XA_BT), 0, (hipStream_t)stream,
                     X, mine, out, (int)n);
  TCE_LAUNCH_CHECK();
  return 0;
}

int tce_xchg_allreduce_f32(void* x, float* buf, int64_t n, void* stream) {
  return xchg_allreduce<float>(x, buf, buf, n, (hipStream_t)stream);
}
int tce_xchg_allreduce_f64(void* x, double* buf, int64_t n, void* stream) {
  return xchg_allreduce<double>(x, buf, buf, n, (hipStream_t)stream);
}
int tce_xchg_allreduce_to_f32(void* x, const float* src, float* dst, int64_t n, void* stream) {
  return xchg_allreduce<float>(x, src, dst, n, (hipStream_t)stream);
}
int tce_xchg_allreduce_to_f64(void* x, const double* src, double* dst, int64_t n,
                              void* stream) {
  return xchg_allreduce<double>(x, src, dst, n, (hipStream_t)stream);
}

}  // extern "C"

namespace {
extern "C" int tce_adam_once_f32(float*, const float*, float*, float*, int64_t, float*, float*,
                                 float, float, float, float, float, float, float, float, void*);
extern "C" int tce_adam_once_f64(double*, const double*, double*, double*, int64_t, double*,
                                 double*, double, double, double, double, double, double, double,
                                 double, void*);
inline int adam_once_any(float* p, const float* g, float* m, float* v, int64_t n, float* s,
                         float* no, float step, float lr, float b1, float b2, float eps, float wd,
                         float clip, float gs, void* st) {
  return tce_adam_once_f32(p, g, m, v, n, s, no, step, lr, b1, b2, eps, wd, clip, gs, st);
}
inline int adam_once_any(double* p, const double* g, double* m, double* v, int64_t n, double* s,
                         double* no, double step, double lr, double b1, double b2, double eps,
                         double wd, double clip, double gs, void* st) {
  return tce_adam_once_f64(p, g, m, v, n, s, no, step, lr, b1, b2, eps, wd, clip, gs, st);
}

template <typename real>
int xchg_adam(void* xv, real* param, real* grad, real* m, real* v, int64_t n, real* state,
              real* norms_out, real step, real lr, real b1, real b2, real eps, real wd, real clip,
              real gscale, void* stream) {
  TCE_CHECK_ARG(xv && param && grad && m && v && state && n > 0 && step >= real(1),
                "xchg_adam: null exchange / buffer, n <= 0, step < 1");
  hipStream_t st = (hipStream_t)stream;
  if (clip > real(0)) {
    // the clip factor needs the norm of the WHOLE summed gradient before any
    // element is applied: all-reduce, then the one-launch step
    // (checked BEFORE the collective is issued: a caller that falls back on the
    // error must find grad untouched and the sequence numbers in step)
    TCE_CHECK_ARG(n <= (1 << 17), "xchg_adam: clipping needs n <= 2^17 (tce_adam_once)");
    if (xchg_allreduce<real>(xv, grad, grad, n, st)) return 1;
    return adam_once_any(param, grad, m, v, n, state, norms_out, step, lr, b1, b2, eps, wd, clip,
                         gscale, stream);
  }
  Xchg* x = static_cast<Xchg*>(xv);
  const int blocks = (int)ceil_div(n, XA_PER);
  XchgView X;
  if (xchg_next(xv, n * (int64_t)sizeof(real), blocks, &X)) return 1;
  hipLaunchKernelGGL(xchg_adam_kernel<real>, dim3(blocks), dim3(XA_BT), 0, st, X, param, grad, m,
                     v, n, state, norms_out, step, lr, b1, b2, eps, wd, gscale, x->partial,
                     x->ticket);
  TCE_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" {

int tce_xchg_adam_f32(void* x, float* param, float* grad, float* m, float* v, int64_t n,
                      float* state, float* norms_out, float step, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float clip, float grad_scale,
                      void* stream) {
  return xchg_adam<float>(x, param, grad, m, v, n, state, norms_out, step, lr, beta1, beta2, eps,
                          weight_decay, clip, grad_scale, stream);
}
int tce_xchg_adam_f64(void* x, double* param, double* grad, double* m, double* v, int64_t n,
                      double* state, double* norms_out, double step, double lr, double beta1,
                      double beta2, double eps, double weight_decay, double clip,
                      double grad_scale, void* stream) {
  return xchg_adam<double>(x, param, grad, m, v, n, state, norms_out, step, lr, beta1, beta2, eps,
                           weight_decay, clip, grad_scale, stream);
}

}  // extern "C"
