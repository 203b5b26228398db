// Fused critic epoch for WIDE and fp64 value networks  D_in -> H -> H -> 1
// (H = 256: the box-pushing / table-tennis critics, fp32 or fp64; H = 128 in
// fp64) on the exact matrix instructions v_mfma_f32_16x16x4_f32 /
// v_mfma_f64_16x16x4_f64, templated on the arithmetic type.
//
// Replaces, per critic epoch (mprl/rl/agent/temporal_correlated_agent.py:343-366):
//   values_new = critic(states[..., :-2 dof])       mprl/util/util_nn.py:225-246
//   loss = value_loss(values_new, returns, old_vs)  :688-716
//   loss.backward()
// Configs: mprl/config/box_push_random_init/tcp/entire/shared.yaml:7,95-96
// (float64, 256 x 2), mprl/config/table_tennis_4d/tcp/entire/shared.yaml:98-103.
//
// W2 (H*H*s = 256 KiB .. 512 KiB) does not fit the LDS next to anything else,
// and the dW2 accumulators (same size) do not fit the registers beside the
// chains.  Two launches per epoch:
//
//  chain kernel (mlpw_chain_kernel): as in mlp.hip everything is computed
//   TRANSPOSED ([hidden x batch]); a wave owns 16 batch rows and keeps H1 and
//   H2 / dY2 of those rows in registers as MFMA result tiles, which ARE the B
//   operands of the next layer (forward X -> H1 -> H2 -> v, backward dY2 ->
//   dH1 -> dY1 never leave the registers).  The A operands (W2 rows forward,
//   W2^T rows backward) stream through LDS in panels of PU output units,
//   double buffered: while the 8 (fp32) / 4 (fp64) waves of the workgroup
//   contract panel p, panel p + 1 is fetched from L2 into registers and
//   written to the other buffer; ONE barrier per panel.  W2 stays L2 resident
//   (every workgroup streams the same 2 x H*H*s bytes per tile).  The kernel
//   also writes H1, dY2 and dY1 of its rows to HBM ([R][H] each) and sums
//   dw3, db3 and the loss.
//  gradient kernel (mlpw_grad_kernel): dW2 = dY2^T H1 and dW1 = dY1^T X as a
//   split-K product over those arrays: a workgroup takes a contiguous range
//   of rows, stages KC rows at a time in LDS ([row][unit], double buffered)
//   and keeps its dW2 / dW1 block in the registers of its 8 waves (fp64,
//   H = 256: two workgroups per row range, half of dW2 each); db2 / db1 are
//   the column sums of its dY2 / dY1 fragments (one add per fragment load
//   instead of 128 DPP row sums per tile in the chain kernel).
//  mlpw_finish_kernel reduces the per-workgroup slabs in fixed order (+ Adam).
//
// MFMA-bound: per row 2 (D H + H H + H) forward + 2 H H (dH1) + 2 (H H + D H)
// (weight gradients) flop; the HBM round trip of H1 / dY2 / dY1 (6 R H s bytes
// per epoch) overlaps with the matrix work of both kernels.
//
// Result-tile layouts differ: register i of lane group g holds tile row 4 g + i
// (f32) but 4 i + g (f64, measured: scripts/probe_mfma_layout.hip).  All
// activation images (registers, LDS, HBM) are therefore kept in POSITION order:
// position 16 J + 4 g + i holds unit 16 J + drow(g, i); weights are staged with
// their input columns permuted to match (mlpw_prep_kernel), the slabs are
// written back in unit order.
#pragma once
#include "common.h"

namespace {

enum { W_TANH = 0, W_RELU = 1, W_LEAKY = 2, W_SOFTPLUS = 3 };

template <typename real> struct WV;
template <> struct WV<float> {
  typedef float v4 __attribute__((ext_vector_type(4), aligned(16)));
  typedef float v2 __attribute__((ext_vector_type(2), aligned(8)));
  typedef float acc __attribute__((ext_vector_type(4)));
  typedef v4 chunk;                                          // 16 bytes
};
template <> struct WV<double> {
  typedef double v4 __attribute__((ext_vector_type(4), aligned(16)));
  typedef double v2 __attribute__((ext_vector_type(2), aligned(16)));
  typedef double acc __attribute__((ext_vector_type(4)));
  typedef v2 chunk;                                          // 16 bytes
};

__device__ inline WV<float>::acc wmfma(float a, float b, WV<float>::acc c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ inline WV<double>::acc wmfma(double a, double b, WV<double>::acc c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// tile row held by register i of lane group g
template <typename real>
__host__ __device__ inline int drow(int g, int i) {
  return sizeof(real) == 4 ? 4 * g + i : 4 * i + g;
}
// unit held at storage position p (an involution)
template <typename real>
__host__ __device__ inline int unit_of_pos(int p) {
  if (sizeof(real) == 4) return p;
  return (p & ~15) | ((p & 3) << 2) | ((p >> 2) & 3);
}

template <typename real, int ACT>
__device__ inline real wact(real y) {
  if (ACT == W_TANH) return tanh(y);
  if (ACT == W_RELU) return y > real(0) ? y : real(0);
  if (ACT == W_LEAKY) return y > real(0) ? y : real(0.01) * y;
  return y > real(20) ? y : log1p(exp(y));
}
// derivative expressed with the OUTPUT h = act(y)
template <typename real, int ACT>
__device__ inline real wact_d(real h) {
  if (ACT == W_TANH) return real(1) - h * h;
  if (ACT == W_RELU) return h > real(0) ? real(1) : real(0);
  if (ACT == W_LEAKY) return h > real(0) ? real(1) : real(0.01);
  return real(1) - exp(-h);
}

// sum over the 16 lanes of a DPP row (all 16 get it)
__device__ inline double row16_sum(double v) {
  v = dpp_sum8(v);
  v += dpp_perm_f64<0x140>(v);
  return v;
}
template <int CTRL>
__device__ inline float dpp_perm_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ inline float row16_sum(float v) {
  v += dpp_perm_f32<0x141>(v);          // row_half_mirror
  v += dpp_perm_f32<0x4E>(v);           // quad_perm [2,3,0,1]
  v += dpp_perm_f32<0xB1>(v);           // quad_perm [1,0,3,2]
  v += dpp_perm_f32<0x140>(v);          // row_mirror
  return v;
}

template <typename real>
struct WArgs {
  const real* x;         // states, row r = (n, t): x + n * env_stride + t * row_stride
  int64_t env_stride, row_stride;
  int T;
  int64_t R;
  int din, act;
  const real *w1, *b1, *b2, *w3, *b3;     // torch Linear layout
  const real *w2p, *w2tp;                 // prepared by mlpw_prep_kernel
  const real *ret, *old_v;
  real clip;
  real* values;                           // [R] (nullable)
  real *h1s, *dy2s, *dy1s;                // [R][H], position order (backward)
  real* partials;                         // [grid][P + 2]
  int P;
  // minibatch: logical row r is row row_index[r] of x / ret / old_v (the
  // workspace rows h1s / dy2s / dy1s stay logical); nullptr: rows in place
  const int64_t* row_index;
};

// fp32 chain-kernel shape (scripts/mlpw_variant.py builds others to compare)
// fp64: 8 waves = two per SIMD at 256 registers each, which holds ONE [hidden x 16
// rows] image, not two (PARK in the chain kernel: H2 waits in the workspace, H1 is
// re-read per backward panel).  6.41 -> 6.16 ms per C3 epoch against 4 waves with
// both images resident: the second wave hides the panel start-up / barrier / epilogue
// stalls that nothing covered at one wave per SIMD (VERDICT r2's suggestion).
#ifndef MLPW_F64_WAVES
#define MLPW_F64_WAVES 8
#endif
#ifndef MLPW_F32_WAVES
#define MLPW_F32_WAVES 8          // waves per workgroup
#define MLPW_F32_PU 32            // output units per W2 panel
#define MLPW_F32_WGS 1            // workgroups per compute unit
#endif

template <typename real>
struct WCfg {
  // chain kernel: two waves per SIMD (fp64: see MLPW_F64_WAVES)
  static constexpr int WAVES = sizeof(real) == 4 ? MLPW_F32_WAVES : MLPW_F64_WAVES;
  static constexpr int WGS = sizeof(real) == 4 ? MLPW_F32_WGS : 1;   // chain workgroups per CU
  // copies of the db1 / db2 / dw3 accumulators in LDS: one per wave (plain
  // read-modify-write, fixed summation order); fewer copies than waves would
  // go through LDS atomics
  static constexpr int NACC = WAVES;
  static constexpr int NT = WAVES * 64;
  static constexpr int TILE = WAVES * 16;                   // batch rows per tile
  static constexpr int PU = sizeof(real) == 4 ? MLPW_F32_PU : 16;    // output units per panel
  static constexpr int NTILE = PU / 16;
  static constexpr int WPAD = sizeof(real) == 4 ? 8 : 2;    // panel pitch = H + WPAD
  static constexpr int KC = sizeof(real) == 4 ? 16 : 8;     // gradient kernel: rows per stage
};

// rows of the [rows][H] workspaces: R rounded up to whole tiles of either type
__host__ __device__ inline int64_t mlpw_ws_rows(int64_t R) { return (R + 127) / 128 * 128; }

__host__ __device__ inline int64_t mlpw_num_params(int din, int H) {
  return (int64_t)H * din + H + (int64_t)H * H + H + H + 1;
}

// w2p[u][p] = W2[u][unit(p)];  w2tp[u1][p2] = W2[unit(p2)][u1]
template <typename real>
__global__ __launch_bounds__(256) void mlpw_prep_kernel(const real* __restrict__ w2, int H,
                                                        real* __restrict__ w2p,
                                                        real* __restrict__ w2tp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * H) return;
  const int r = e / H, p = e - r * H;
  const int u = unit_of_pos<real>(p);
  w2p[e] = w2[r * H + u];
  w2tp[e] = w2[u * H + r];
}

// ---------------------------------------------------------------------------
// chain kernel
// ---------------------------------------------------------------------------
// dW1 / db1 inside the chain kernel (round 3): the dY1 tile of a backward panel
// goes through a small LDS transpose instead of the [R][H] workspace, the
// gradient kernel reads dY2 and H1 only.  For the shapes whose LDS has room for
// the two transpose buffers (D_in <= 24, H = 256: both shipped wide critics).
#ifdef MLPW_NO_FUSE
template <typename real, int H, int KPG> struct WFuse { static constexpr bool on = false; };
#else
template <typename real, int H, int KPG> struct WFuse {
  typedef WCfg<real> C;
  static constexpr int NOT1 = C::NTILE * 2;                  // dW1 output tiles per panel (32 features)
  // fp32 only: with two waves per SIMD (one of either turn group) the extra
  // MFMAs spread evenly; the one-wave-per-SIMD fp64 kernel would wait at every
  // panel barrier for the two waves whose turn it is
  static constexpr bool on = sizeof(real) == 4 && H == 256 && KPG == 6 && C::WGS == 1 &&
                             C::WAVES % NOT1 == 0 && (H / C::PU) % (C::WAVES / NOT1) == 0;
};
#endif

// panels by LDS-DMA (fp32) or through registers (fp64: measured slower by DMA --
// one wave per SIMD sits in the barrier's vmcnt(0) instead of in its MFMAs)
#ifndef MLPW_GLDS_F32
#define MLPW_GLDS_F32 1
#define MLPW_GLDS_F64 0
#endif
template <typename real> struct WGlds {
  static constexpr bool on = sizeof(real) == 4 ? MLPW_GLDS_F32 : MLPW_GLDS_F64;
};

template <typename real, int H, int KPG>
struct ChainLds {
  static constexpr int W1P = 4 * KPG + 4;
  static constexpr int WP = H + WCfg<real>::WPAD;
  static constexpr int PANEL = WCfg<real>::PU * WP;
  // dY1 transpose: [2][TILE rows][PU positions], pitch PU + 4 (fp32: 36 = 4 x
  // odd, the 16-byte row writes of 16 consecutive rows are conflict free) /
  // PU + 2 (fp64)
  static constexpr int TP1 = WCfg<real>::PU + (sizeof(real) == 4 ? 4 : 2);
  static constexpr int T1 = WFuse<real, H, KPG>::on ? 2 * WCfg<real>::TILE * TP1 : 0;
  // the tile's X rows [TILE][4 KPG features], pitch odd
  static constexpr int XP = 4 * KPG + 1;
  static constexpr int XS = WFuse<real, H, KPG>::on ? WCfg<real>::TILE * XP : 0;
  // PARK (fp64): the signs of the tile's H1, one bit per element -- a byte per
  // thread and pair of unit blocks
  static constexpr int SG = sizeof(real) == 8 ? WCfg<real>::NT * (H / 32) / 8 : 0;
  static constexpr size_t bytes(bool bwd) {
    return sizeof(real) * ((size_t)H * W1P + 3 * H + 2 * PANEL +
                           (bwd ? (size_t)WCfg<real>::NACC * H + T1 + XS + SG : 0)) + 64;
  }
};

#ifdef MLPW_STAMP
// diagnostic build (scripts/mlpw_stamps.py): cycles per phase of the chain
// kernel (wave 0 of workgroup 0), left in the first elements of the dY1 rows
#define WSTAMP(k) { const long long tn_ = __builtin_readcyclecounter(); stt_[k] += tn_ - tp_; tp_ = tn_; }
#else
#define WSTAMP(k)
#endif

// k-block after which panel_mma runs its mid() callback (even: the one-tile
// path steps by two)
#ifndef MLPW_MID_AT
#define MLPW_MID_AT(NJ) ((NJ) / 2)
#endif

// nothing moves across: keeps the compiler from hoisting every LDS read of an
// unrolled phase to its top (which spills) and the next step's reads ahead of
// this step's MFMAs
__device__ inline void wfence() { __builtin_amdgcn_sched_barrier(0); }

// acc[jj] += panel rows (16 jj + m) . B operand tiles (H / 16 of them).  The
// A fragments of k-block Jk + 1 are read while block Jk is multiplied.
// `mid()` is called once in the middle of the MFMA sequence: work that does
// not depend on this panel's result (the LDS writes of the NEXT panel, fetched
// into registers at the top of the step) issues in the shadow of the matrix
// pipe instead of after the last MFMA.
// `hook(h)`, h = 0 .. WHooks<real>::N - 1, is called once per k-block pair
// (fp64) / k-block (fp32) between two groups of MFMAs: a slice of work left
// over from the PREVIOUS panel (its activation epilogue, the dW1 products of
// an earlier one) issues while the matrix pipe works on this panel, instead of
// after the panel's last MFMA with the pipe idle.
struct NoMid { __device__ void operator()() const {} };
struct NoHook { __device__ void operator()(int) const {} };
template <typename real, int H> struct WHooks {
  static constexpr int N = WCfg<real>::NTILE == 1 ? H / 32 : H / 16;   // hook calls per panel
  static constexpr int NE = 4 * WCfg<real>::NTILE;                     // result elements per lane and panel
  // elements [lo(h), lo(h + 1)) of the previous panel are finished in hook h >= 1
  // (hook 0 requests what they need from the LDS)
  static constexpr int lo(int h) { return h < 1 ? 0 : ((h - 1) * NE + N - 2) / (N - 1); }
};
// (timing experiment only -- wrong results: no barrier between panel steps)
#ifdef MLPW_NOSYNC
#define WSTEP_SYNC() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define WSTEP_SYNC() __syncthreads()
#endif
#ifndef MLPW_DEFER
#define MLPW_DEFER 1              // 0: every panel's epilogue right behind its MFMAs (round 2)
#endif
template <typename real, int H, typename Mid = NoMid, typename Hook = NoHook>
__device__ inline void panel_mma(const real* pan, int m, int g,
                                 const typename WV<real>::acc* bop,
                                 typename WV<real>::acc* acc, Mid mid = Mid(), Hook hook = Hook()) {
  typedef typename WV<real>::v4 v4;
  typedef typename WV<real>::acc vacc;
  constexpr int WP = H + WCfg<real>::WPAD;
  constexpr int NTILE = WCfg<real>::NTILE;
  constexpr int NJ = H / 16;
  // (fp64: two lanes of every ds_read_b128 group meet on a bank whatever the
  // pitch; a pair-swap of the 32-byte pieces of rows 4..11 removes that and
  // changed nothing measurable -- the LDS is not the limit here)
  const real* p = pan + m * WP + 4 * g;
  if (NTILE == 1) {
    // one output tile: two interleaved accumulation chains over the even / odd
    // k-blocks (a dependent MFMA needs more than its issue interval)
    vacc alt = {0, 0, 0, 0};
    v4 A[2][2];
    A[0][0] = *reinterpret_cast<const v4*>(p);
    A[0][1] = *reinterpret_cast<const v4*>(p + 16);
#pragma unroll
    for (int Jk = 0; Jk < NJ; Jk += 2) {
      const int b = (Jk >> 1) & 1;
      // the next block's fragments are requested behind this block's first
      // MFMAs: the wait at the top of the next block finds them there
#pragma unroll
      for (int i = 0; i < 1; ++i) {
        acc[0] = wmfma(A[b][0][i], bop[Jk][i], acc[0]);
        alt = wmfma(A[b][1][i], bop[Jk + 1][i], alt);
      }
      wfence();
      if (Jk + 2 < NJ) {
        A[b ^ 1][0] = *reinterpret_cast<const v4*>(p + 16 * (Jk + 2));
        A[b ^ 1][1] = *reinterpret_cast<const v4*>(p + 16 * (Jk + 3));
      }
      hook(Jk >> 1);
      wfence();
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        acc[0] = wmfma(A[b][0][i], bop[Jk][i], acc[0]);
        alt = wmfma(A[b][1][i], bop[Jk + 1][i], alt);
      }
      wfence();
      if (Jk == MLPW_MID_AT(NJ)) { mid(); wfence(); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[0][i] += alt[i];
  } else {
    v4 A[2][NTILE];
#pragma unroll
    for (int jj = 0; jj < NTILE; ++jj)
      A[0][jj] = *reinterpret_cast<const v4*>(p + 16 * jj * WP);
#pragma unroll
    for (int Jk = 0; Jk < NJ; ++Jk) {
      const int b = Jk & 1;
#pragma unroll
      for (int i = 0; i < 1; ++i)
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) acc[jj] = wmfma(A[b][jj][i], bop[Jk][i], acc[jj]);
      wfence();
      if (Jk + 1 < NJ) {
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj)
          A[b ^ 1][jj] = *reinterpret_cast<const v4*>(p + 16 * jj * WP + 16 * (Jk + 1));
      }
      hook(Jk);
      wfence();
#pragma unroll
      for (int i = 1; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) acc[jj] = wmfma(A[b][jj][i], bop[Jk][i], acc[jj]);
      wfence();
      if (Jk == MLPW_MID_AT(NJ)) { mid(); wfence(); }
    }
  }
  wfence();
}

template <typename real, int H, int KPG, int ACT, bool BWD>
__global__ __launch_bounds__(WCfg<real>::NT, 1) void mlpw_chain_kernel(WArgs<real> a) {
  typedef typename WV<real>::v4 v4;
  typedef typename WV<real>::v2 v2;
  typedef typename WV<real>::acc vacc;
  typedef WCfg<real> C;
  typedef ChainLds<real, H, KPG> LD;
  constexpr int NJ = H / 16, NP = H / C::PU, NTILE = C::NTILE, NT = C::NT;
  constexpr int W1P = LD::W1P, WP = LD::WP;
  constexpr int NSTEP = BWD ? 2 * NP : NP;                   // panels per tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* W1s = reinterpret_cast<real*>(smem_raw);             // [H][W1P] natural rows
  real* Bs = W1s + H * W1P;                                  // b1 | b2 | w3, position order
  real* pan = Bs + 3 * H;                                    // [2][PU][WP]
  real* gacc = pan + 2 * LD::PANEL;                          // [NACC][H] dw3 (positions)
  real* t1buf = gacc + C::NACC * H;                          // [2][TILE][TP1] dY1 of a panel (FUSE)
  real* xs = t1buf + LD::T1;                                 // [TILE][XP] X rows of the tile (FUSE)
  unsigned char* h1m = reinterpret_cast<unsigned char*>(xs + LD::XS);   // [H / 32][NT] sign bits of H1 (PARK)
  constexpr int XP = LD::XP;
  constexpr bool FUSE = BWD && WFuse<real, H, KPG>::on;
  constexpr int TP1 = LD::TP1;
  // FUSE: this wave's share of dW1 (+ db1 through a column of ones): panel sp's
  // NTILE x 2 output tiles [16 unit positions x 16 features] go to the waves
  // ((sp * NTILE * 2) + 2 at + xt) mod WAVES, each contracting all TILE rows
  constexpr int NOT1 = NTILE * 2;                            // dW1 output tiles per panel
  constexpr int NG1 = C::WAVES / NOT1 > 0 ? C::WAVES / NOT1 : 1;   // panel groups taking turns
  constexpr int NPW = FUSE ? NP / NG1 : 1;                   // panels (accumulators) per wave
  vacc gw1[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) gw1[q] = (vacc){0, 0, 0, 0};
  __shared__ real sred[2 * C::WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int my_grp = wave_u / NOT1, my_at = (wave_u % NOT1) >> 1, my_xt = wave_u & 1;
  const int m = lane & 15, g = lane >> 4;                    // m: A row / batch column
  const int din = a.din;
  const int64_t ntiles = (a.R + C::TILE - 1) / C::TILE;

  // ---- resident images
  for (int e = tid; e < H * 4 * KPG; e += NT) {
    const int u = e / (4 * KPG), f = e - u * 4 * KPG;
    W1s[u * W1P + f] = f < din ? a.w1[u * din + f] : real(0);
  }
  for (int e = tid; e < H; e += NT) {
    const int u = unit_of_pos<real>(e);
    Bs[e] = a.b1[u];
    Bs[H + e] = a.b2[u];
    Bs[2 * H + e] = a.w3[u];
  }
  if (BWD)
    for (int e = tid; e < C::NACC * H; e += NT) gacc[e] = real(0);

  // ---- panel stream: step s of a tile reads W2p rows (s < NP) or W2Tp rows
  auto panel_src = [&](int s) -> const real* {
    return s < NP ? a.w2p + (int64_t)s * C::PU * H : a.w2tp + (int64_t)(s - NP) * C::PU * H;
  };
  // a panel moves in 16-byte chunks, chunk q NT + t by thread t: consecutive
  // lanes load consecutive 16 bytes and write them to consecutive LDS banks
  // (a thread writing 64 or 128 contiguous bytes put 4 / 8 lanes of every
  // ds_write_b128 group on one bank).  ONE 32-bit lane offset per panel: the
  // chunks differ in the uniform base (global) / an immediate offset (LDS).
  typedef typename WV<real>::chunk chunk;
  constexpr int EPC = 16 / (int)sizeof(real);                // elements per chunk
  constexpr int CPR = H / EPC;                               // chunks per panel row
  constexpr int NCH = C::PU * CPR / NT;                      // chunks per thread and panel
  static_assert(C::PU * CPR % NT == 0 && NT % CPR == 0, "panel copy");
  chunk stg[NCH];
  const unsigned goff = (unsigned)tid * EPC;
  const int soff = (tid / CPR) * WP + (tid % CPR) * EPC;
  auto fetch = [&](int s) {
    // uniform (SGPR) base + 32-bit lane offset; the empty asm keeps the
    // compiler from hoisting one 64-bit lane address per panel out of the tile
    // loop (32 registers, which then spill).  It also hides where the pointer
    // came from: without the explicit global address space these would be
    // FLAT loads, which count in lgkmcnt as well.
    const real* base = panel_src(s);
    asm volatile("" : "+s"(base));
    typedef const __attribute__((address_space(1))) chunk* gchunkp;
#pragma unroll
    for (int q = 0; q < NCH; ++q) stg[q] = *(gchunkp)(base + q * NT * EPC + goff);
  };
  auto stash = [&](int buf) {
    real* dst = pan + buf * LD::PANEL + soff;
#pragma unroll
    for (int q = 0; q < NCH; ++q) *reinterpret_cast<chunk*>(dst + q * (NT / CPR) * WP) = stg[q];
  };
  // GLDS: panels go from L2 straight into the LDS (global_load_lds_dwordx4: one wave
  // instruction = 1 KiB of one panel row, wave-uniform LDS address + 16 bytes
  // per lane): no staging registers, no ds_write pass.  The barrier that ends a
  // step waits for them (hipcc drains vmcnt before a barrier while an LDS-DMA
  // is in flight).
  constexpr int IPR = H * (int)sizeof(real) >= 1024 ? H * (int)sizeof(real) / 1024 : 1;   // instructions per panel row
  constexpr int NDMA = C::PU * IPR / C::WAVES;               // per wave and panel
  static_assert(C::PU * IPR % C::WAVES == 0, "panel DMA");
  constexpr bool GLDS = WGlds<real>::on && H * sizeof(real) >= 1024;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  auto dma = [&](int s, int buf) {
    const real* base = panel_src(s);
    asm volatile("" : "+s"(base));
    typedef const __attribute__((address_space(1))) void* gvp;
    typedef __attribute__((address_space(3))) void* lvp;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
      const int e = wave_s * NDMA + q, row = e / IPR, part = e % IPR;
      __builtin_amdgcn_global_load_lds(
          (gvp)(base + row * H + part * (1024 / (int)sizeof(real)) + lane * EPC),
          (lvp)(pan + buf * LD::PANEL + row * WP + part * (1024 / (int)sizeof(real))), 16, 0, 0);
    }
  };
  if (GLDS) {
    dma(0, 0);
  } else {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  // the panel steps of a tile alternate between the two buffers; their number
  // per tile is even, so step s always reads buffer s & 1 (compile-time
  // addresses)
  static_assert(NSTEP % 2 == 0, "panel buffers");

  const real b3 = a.b3[0];
  const real inv_n = real(1) / (real)a.R;
  real loss_sum = 0, gb3 = 0;
  real* my_acc = gacc + (wave % C::NACC) * H;
  constexpr bool ACC_ATOMIC = C::NACC < C::WAVES;

  // x fragment: lane group g holds features KPG g + s of the lane's row (zeros
  // past D_in); one lane address, the features are immediate offsets of it
  auto load_x = [&](int64_t tile, real* dst) {
    int64_t r = tile * C::TILE + wave * 16 + m;
    if (r >= a.R) r = a.R - 1;
    if (a.row_index) r = a.row_index[r];
    const int64_t ne = r / a.T;
    const int t = (int)(r - ne * a.T);
    const real* xp = a.x + ne * a.env_stride + t * a.row_stride + KPG * g;
    const int nk = din - KPG * g;                            // features of this lane group
#pragma unroll
    for (int s = 0; s < KPG; ++s) dst[s] = s < nk ? xp[s] : real(0);
  };
  real xn[KPG];
  load_x(blockIdx.x, xn);

#ifdef MLPW_STAMP
  long long stt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tp_ = __builtin_readcyclecounter();
#endif
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    WSTAMP(7)
    const int64_t r = tile * C::TILE + wave * 16 + m;
    const bool rok = r < a.R;
    const bool more = tile + gridDim.x < ntiles;
    real xb[KPG];
#pragma unroll
    for (int s = 0; s < KPG; ++s) xb[s] = rok ? xn[s] : real(0);
    real rt = 0, ov = 0;
    if (BWD) {
      int64_t rp = rok ? r : a.R - 1;
      if (a.row_index) rp = a.row_index[rp];
      rt = a.ret[rp];
      if (a.clip > real(0)) ov = a.old_v[rp];
    }
    // (FUSE / every backward build: the next tile's rows are requested in the
    // last backward step -- KPG registers (fp64: twice that) less through the tile)
#ifndef MLPW_XLATE
#define MLPW_XLATE 1
#endif
    constexpr bool XLATE = MLPW_XLATE && BWD;
    if (more && !(FUSE || XLATE)) load_x(tile + gridDim.x, xn);
    // one turn: dW1 tile (panel sp, my_at, my_xt) += dY1[rows][positions]^T X[rows][features]
    // over the TILE rows (A from the transpose buffer sp & 1, B = X from L2; the
    // feature D_in is a column of ones: db1)
    auto dw1_turn = [&](int sp, vacc& accum) {
      const real* tb = t1buf + (sp & 1) * C::TILE * TP1 + 16 * my_at + m;
      const int f = 16 * my_xt + m;
      const int fc = f < din ? f : din - 1;
      // (opaque per turn: the X rows are the same in every turn of a tile, and
      // the compiler would otherwise keep all of them in registers)
      int jo = g;
      asm volatile("" : "+v"(jo));
      const real* xf = xs + fc;
      constexpr int CH = 4;                                  // k-steps per batch of loads
#pragma unroll
      for (int k0 = 0; k0 < C::TILE / 4; k0 += CH) {
        real av[CH], bv[CH];
#pragma unroll
        for (int kk = 0; kk < CH; ++kk) {
          const int j = 4 * (k0 + kk) + jo;                  // row of the tile
          av[kk] = tb[j * TP1];
          const real xv = xf[j * XP];
          bv[kk] = f < din ? xv : (f == din ? real(1) : real(0));
        }
        wfence();
#pragma unroll
        for (int kk = 0; kk < CH; ++kk) accum = wmfma(av[kk], bv[kk], accum);
        wfence();
      }
    };

    // PARK with a piecewise-linear activation: act'(H1) is one bit per element
    // (h > 0).  The 64 bits of a lane wait in the LDS (held in two registers
    // they cost 90 spilled ones) instead of a second pass over the H1 rows in
    // the backward panels: - 1.7 GB of reads per C3 epoch, 5.98 -> 5.95 ms
    // (with nothing in their place: 5.82)
#ifndef MLPW_MASK1
#define MLPW_MASK1 1
#endif
    constexpr bool SIGN1 = MLPW_MASK1 && BWD && sizeof(real) == 8 && C::WAVES == 8 &&
                           (ACT == W_RELU || ACT == W_LEAKY);
    // ---- layer 1: H1^T = act(W1 X^T + b1), two row blocks at a time
    // (go: the lane group, opaque per tile -- the forward-only kernel has no LDS
    // stores in its loop, and the compiler would otherwise keep every bias, w3
    // and W1 value it reads in registers across the tiles: 121 spilled
    // registers in the fp64 build)
    int go = g;
    asm volatile("" : "+v"(go));
    vacc h1[NJ];
#pragma unroll
    for (int J = 0; J < NJ; J += 2) {
      vacc c0 = *reinterpret_cast<const v4*>(Bs + 16 * J + 4 * go);
      vacc c1 = *reinterpret_cast<const v4*>(Bs + 16 * J + 16 + 4 * go);
      const real* p0 = W1s + (16 * J + m) * W1P + KPG * go;
      const real* p1 = p0 + 16 * W1P;
#pragma unroll
      for (int s = 0; s < KPG; s += 2) {
        const v2 a0 = *reinterpret_cast<const v2*>(p0 + s);
        const v2 a1 = *reinterpret_cast<const v2*>(p1 + s);
        c0 = wmfma(a0[0], xb[s], c0);
        c1 = wmfma(a1[0], xb[s], c1);
        c0 = wmfma(a0[1], xb[s + 1], c0);
        c1 = wmfma(a1[1], xb[s + 1], c1);
      }
      if (SIGN1) {
        // (y > 0 and act(y) > 0 are the same predicate for both activations:
        // one compare serves the activation and the bit)
        const real slope = ACT == W_RELU ? real(0) : real(0.01);
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool p0 = c0[i] > real(0), p1 = c1[i] > real(0);
          c0[i] = p0 ? c0[i] : slope * c0[i];
          c1[i] = p1 ? c1[i] : slope * c1[i];
          bits |= p0 ? 1u << i : 0u;
          bits |= p1 ? 16u << i : 0u;
        }
        h1m[(J >> 1) * NT + tid] = (unsigned char)bits;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c0[i] = wact<real, ACT>(c0[i]);
          c1[i] = wact<real, ACT>(c1[i]);
        }
      }
      h1[J] = c0;
      h1[J + 1] = c1;
      wfence();
    }
    // rows of this lane in the [R][H] workspaces (position order)
    real* ph = a.h1s + r * H + 4 * g;
    real* pd = a.dy2s + r * H + 4 * g;
    real* p1s = a.dy1s + r * H + 4 * g;
    // (the workspaces hold whole tiles: rows past R are written and read like
    // the others -- no data-dependent branch around a store, so the waits on
    // the shared load / store counter stay countable -- and never used)
    // (the H1 rows are stored behind the first panel fetch below)

    WSTAMP(0)
    // ---- layer 2 through the W2 panels: H2^T = act(W2 H1^T + b2).  Their
    // share of v = w3 . H2 is taken at once; (backward) the tiles stay in
    // registers -- they become dY2, the B operand of the backward panels -- next
    // to H1, which the backward panels need for act'(H1).  (Round 2 parked H2
    // in the dY2 rows of the workspace and read H1 back: 3 x R x H elements of
    // HBM traffic per epoch for 64 registers.)
    real v = 0;
    // PARK (fp64 with two waves per SIMD: 256 registers per lane): ONE register
    // image of [hidden x 16 rows] -- H1 through the forward panels, then dY2.
    // The H2 tiles wait in the dY2 rows of the workspace (each lane reads back
    // what it wrote), H1 for act'(H1) is re-read per backward panel.
    constexpr bool PARK = BWD && sizeof(real) == 8 && C::WAVES == 8;
    vacc dy2_own[BWD && !PARK ? NJ : 1];
    vacc* dy2 = PARK ? h1 : dy2_own;
    vacc park_t[NTILE];
    // The epilogue of panel s - 1 (bias, activation, its share of v) runs inside
    // the MFMAs of panel s, one element per hook; only the last panel's is
    // exposed.  (The bias is added there too: the accumulators start at zero
    // instead of waiting for an LDS read at the top of every panel.)
    // (fp32 only: + 0.5 %; the fp64 kernel measured the same either way and needs
    // the registers)
    constexpr bool DEFER = MLPW_DEFER && sizeof(real) == 4;
    vacc facc[2][NTILE];
    auto fwd_elem = [&](int sp, int n, const vacc* b2r, const vacc* w3r) {
      const int jj = n >> 2, i = n & 3;
      const real hv = wact<real, ACT>(facc[sp & 1][jj][i] + b2r[jj][i]);
      v += w3r[jj][i] * hv;
      if (PARK) park_t[jj][i] = hv;
      else if (BWD) dy2[sp * NTILE + jj][i] = hv;
    };
    vacc b2r[NTILE], w3r[NTILE];
    const int bo = 4 * go;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const bool last = !BWD && s == NP - 1;
      // another panel follows (GLDS: no branch around the DMA -- after the last
      // tile panel 0 is fetched once more into the free buffer and never read)
      const bool pre = GLDS || !last || more;
      // (GLDS: the other buffer is free since the last barrier)
      if (pre) {
        if (GLDS) dma(last ? 0 : s + 1, (s & 1) ^ 1);
        else fetch(last ? 0 : s + 1);
      }
      // H1 for the gradient kernel: behind the panel fetch, so that nothing
      // the fetch needs (the fp64 build reloads a spilled lane offset for it)
      // waits for these 2 KiB per row to be acknowledged
      if (BWD && s == 0) {
#pragma unroll
        for (int J = 0; J < NJ; ++J) *reinterpret_cast<v4*>(ph + 16 * J) = h1[J];
      }
#pragma unroll
      for (int jj = 0; jj < NTILE; ++jj) facc[s & 1][jj] = (vacc){0, 0, 0, 0};
      // the other buffer is free since the last barrier: the next panel goes
      // there in the middle of this panel's MFMAs (its fetch was issued at the
      // top of the step; the LDS writes issue beside the matrix pipe)
      panel_mma<real, H>(
          pan + (s & 1) * LD::PANEL, m, g, h1, facc[s & 1],
          [&]() { if (!GLDS && pre) stash((s & 1) ^ 1); },
          [&](int h) {
            if (s == 0 || !DEFER) return;
            typedef WHooks<real, H> HK;
            if (h == 0) {
#pragma unroll
              for (int jj = 0; jj < NTILE; ++jj) {
                b2r[jj] = *reinterpret_cast<const v4*>(Bs + H + 16 * ((s - 1) * NTILE + jj) + bo);
                w3r[jj] = *reinterpret_cast<const v4*>(Bs + 2 * H + 16 * ((s - 1) * NTILE + jj) + bo);
              }
            } else {
#pragma unroll
              for (int n = HK::lo(h); n < HK::lo(h + 1); ++n) fwd_elem(s - 1, n, b2r, w3r);
            }
          });
#ifndef MLPW_PARK_EARLY
#define MLPW_PARK_EARLY 1
#endif
      // PARK, last forward panel: H1 was the B operand for the last time -- the
      // parked H2 tiles come back into its registers behind the panel's
      // MFMAs instead of behind the loss (the last panel's own tiles never
      // leave the registers)
      if (MLPW_PARK_EARLY && PARK && s == NP - 1) {
#pragma unroll
        for (int J = 0; J < NJ - NTILE; ++J) dy2[J] = *reinterpret_cast<const v4*>(pd + 16 * J);
      }
      WSTAMP(4)
      // (v pinned here: where a step ends in a branch -- the register-staged
      // build -- the compiler otherwise sinks every panel's epilogue to the end
      // of the tile, with all the raw accumulators alive until then: 214
      // spilled registers in the fp64 forward kernel)
      asm volatile("" : "+v"(v));
      if (s == NP - 1 || !DEFER) {
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) {
          b2r[jj] = *reinterpret_cast<const v4*>(Bs + H + 16 * (s * NTILE + jj) + bo);
          w3r[jj] = *reinterpret_cast<const v4*>(Bs + 2 * H + 16 * (s * NTILE + jj) + bo);
        }
#pragma unroll
        for (int n = 0; n < 4 * NTILE; ++n) fwd_elem(s, n, b2r, w3r);
        if (PARK && !(MLPW_PARK_EARLY && s == NP - 1)) {
#pragma unroll
          for (int jj = 0; jj < NTILE; ++jj)
            *reinterpret_cast<v4*>(a.dy2s + r * H + 4 * g + 16 * (s * NTILE + jj)) = park_t[jj];
        }
      }
      WSTAMP(6)
      WSTEP_SYNC();
      // past the tile's first barrier no wave is still in the previous tile's
      // last dW1 turn: the X rows of this tile replace the old ones
      if (FUSE && s == 0) {
        real* xw = xs + (wave * 16 + m) * XP + KPG * g;
#pragma unroll
        for (int k = 0; k < KPG; ++k) xw[k] = xb[k];
      }
    }

    WSTAMP(1)
    // ---- value head, loss, dL/dv
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    v += b3;
    if (a.values && rok && g == 0) a.values[r] = v;

    if (BWD) {
      real dv;
      {
        const real e = v - rt;
        real l = e * e, d = real(2) * e;
        if (a.clip > real(0)) {
          const real dlt = v - ov;
          const real cl = dlt < -a.clip ? -a.clip : (dlt > a.clip ? a.clip : dlt);
          const real e2 = ov + cl - rt;
          if (e2 * e2 > l) {
            l = e2 * e2;
            d = (dlt > -a.clip && dlt < a.clip) ? real(2) * e2 : real(0);
          }
        }
        if (!rok) { l = 0; d = 0; }
        dv = d * inv_n;
        if (g == 0) { loss_sum += l; gb3 += dv; }
      }
      // dY2 = dv w3 act'(H2) in place of the H2 tiles: written to the workspace
      // for the gradient kernel and kept as the B operand of the backward
      // panels; dw3 = sums over the 16 batch lanes of a row (db2 and db1 are
      // column sums of the dY2 / dY1 rows: the gradient kernel takes them from
      // its A fragments for free)
      if (PARK && MLPW_PARK_EARLY) {
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) dy2[NJ - NTILE + jj] = park_t[jj];
      } else if (PARK) {
        // H2 back from the workspace into the image H1 occupied (its last use as
        // the B operand was the last forward panel)
#pragma unroll
        for (int J = 0; J < NJ; ++J) dy2[J] = *reinterpret_cast<const v4*>(pd + 16 * J);
      }
#ifndef MLPW_DY2_GROUP
#define MLPW_DY2_GROUP 4
#endif
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const v4 w3v = *reinterpret_cast<const v4*>(Bs + 2 * H + 16 * J + 4 * g);
        vacc t3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const real hv = dy2[J][i];
          t3[i] = dv * hv;
          dy2[J][i] = dv * w3v[i] * wact_d<real, ACT>(hv);
        }
        *reinterpret_cast<v4*>(pd + 16 * J) = dy2[J];
        // (tried: a four-value butterfly -- 5 cross-lane adds instead of 16 -- for
        // these row sums: no measurable difference, fp32 or fp64)
#pragma unroll
        for (int i = 0; i < 4; ++i) t3[i] = row16_sum(t3[i]);
        if (m == 0) {
          real* q3 = my_acc + 16 * J + 4 * g;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (ACC_ATOMIC) unsafeAtomicAdd(q3 + i, t3[i]);
            else q3[i] += t3[i];
          }
        }
        // (the tiles of a group overlap: LDS reads, cross-lane sums and the
        // read-modify-write of the dw3 slab are latency chains)
        if (J % MLPW_DY2_GROUP == MLPW_DY2_GROUP - 1) wfence();
      }
      WSTAMP(2)
      // ---- dH1^T = W2^T dY2^T through the W2^T panels; dY1 = dH1 act'(H1)
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const bool last = s == NP - 1;
        const bool pre = GLDS || !last || more;
        if (pre) {
          if (GLDS) dma(last ? 0 : NP + s + 1, ((NP + s) & 1) ^ 1);
          else fetch(last ? 0 : NP + s + 1);
        }
        if ((FUSE || XLATE) && last && more) load_x(tile + gridDim.x, xn);
        vacc h1p[NTILE];                                       // PARK: this panel's H1 tile, from the workspace
        unsigned h1b[NTILE];                                   // ... or its sign bits
        if (PARK && !SIGN1) {
#pragma unroll
          for (int jj = 0; jj < NTILE; ++jj)
            h1p[jj] = *reinterpret_cast<const v4*>(ph + 16 * (s * NTILE + jj));
        } else if (SIGN1) {
#pragma unroll
          for (int jj = 0; jj < NTILE; ++jj) h1b[jj] = h1m[((s * NTILE + jj) >> 1) * NT + tid];
        }
        vacc acc[NTILE];
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) acc[jj] = (vacc){0, 0, 0, 0};
        panel_mma<real, H>(pan + ((NP + s) & 1) * LD::PANEL, m, g, dy2, acc, [&]() {
          if (!GLDS && pre) stash(((NP + s) & 1) ^ 1);
          // the dY1 tile of panel s - 1 became visible at the last barrier: the
          // waves whose turn it is take its dW1 tiles now, beside this panel's MFMAs
          if (FUSE && s > 0 && ((s - 1) % NG1) == my_grp) dw1_turn(s - 1, gw1[(s - 1) / NG1]);
        });
#pragma unroll
        for (int jj = 0; jj < NTILE; ++jj) {
          const int J = s * NTILE + jj;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (SIGN1)
              acc[jj][i] *= ((h1b[jj] >> (4 * (J & 1) + i)) & 1u)
                                ? real(1) : (ACT == W_RELU ? real(0) : real(0.01));
            else
              acc[jj][i] *= wact_d<real, ACT>(PARK ? h1p[jj][i] : h1[J][i]);
          }
          if (FUSE)
            *reinterpret_cast<v4*>(t1buf + (s & 1) * C::TILE * TP1 + (wave * 16 + m) * TP1 +
                                   16 * jj + 4 * g) = acc[jj];
          else
            *reinterpret_cast<v4*>(p1s + 16 * J) = acc[jj];
        }
        WSTEP_SYNC();
      }
      // the last panel's dW1 tiles (its dY1 tile is visible since the barrier above)
      if (FUSE && ((NP - 1) % NG1) == my_grp) dw1_turn(NP - 1, gw1[(NP - 1) / NG1]);
      WSTAMP(3)
    }
  }
#ifdef MLPW_STAMP
  if (BWD && blockIdx.x == 0 && tid == 0)
    for (int k = 0; k < 8; ++k) a.dy1s[k] = (real)stt_[k];
#endif

  if (FUSE) {
    // ---- slab sections W1, b1 (unit order): this wave's tiles
    real* outp = a.partials + (int64_t)blockIdx.x * (a.P + 2);
    real* oW1 = outp;
    real* ob1 = outp + (int64_t)H * din;
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int sp = q * NG1 + my_grp;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pos = C::PU * sp + 16 * my_at + drow<real>(g, i);
        const int u = unit_of_pos<real>(pos);
        const int f = 16 * my_xt + m;
        if (f < din) oW1[(int64_t)u * din + f] = gw1[q][i];
        else if (f == din) ob1[u] = gw1[q][i];
      }
    }
  }
  if (BWD) {
    // ---- slab: w3 (unit order), b3, loss
    loss_sum = wave_sum(loss_sum);
    gb3 = wave_sum(gb3);
    if (lane == 0) { sred[wave] = gb3; sred[C::WAVES + wave] = loss_sum; }
    __syncthreads();
    real* out = a.partials + (int64_t)blockIdx.x * (a.P + 2);
    real* ob1 = out + (int64_t)H * din;
    real* ob2 = ob1 + H + (int64_t)H * H;
    real* ow3 = ob2 + H;
    for (int p = tid; p < H; p += NT) {
      real s = 0;
#pragma unroll
      for (int w = 0; w < C::NACC; ++w) s += gacc[w * H + p];
      ow3[unit_of_pos<real>(p)] = s;
    }
    if (tid == 0) {
      real s3 = 0, sl = 0;
#pragma unroll
      for (int w = 0; w < C::WAVES; ++w) { s3 += sred[w]; sl += sred[C::WAVES + w]; }
      ow3[H] = s3;                                            // b3
      ow3[H + 1] = sl;                                        // sum of the losses of this workgroup's rows
      ow3[H + 2] = 0;
    }
  }
}

// ---------------------------------------------------------------------------
// gradient kernel: dW2 = dY2^T H1, dW1 = dY1^T X over a range of rows
// ---------------------------------------------------------------------------
template <typename real, int H>
struct GradCfg {
  static constexpr int NSPLIT = (sizeof(real) == 8 && H > 128) ? 2 : 1;
  static constexpr int UR = H / NSPLIT;          // dY2 / dY1 units per workgroup
  static constexpr int UW = UR / 8;              // per wave
  static constexpr int NAT = UW / 16;
  static_assert(UW % 16 == 0, "wave slice");
};

template <typename real, int H, int KPG>
struct GradLds {
  static constexpr int XW = KPG == 6 ? 32 : 48;  // X columns (>= D_in), multiple of 16
  static constexpr int PX = XW == 32 ? 48 : 80;  // pitches = 16 mod 32 (conflict-free column reads)
  static constexpr int PA = GradCfg<real, H>::UR + 16;
  static constexpr int PB = H + 16;
  static constexpr bool FUSE = WFuse<real, H, KPG>::on;     // dW1 / db1 come from the chain kernel
  static constexpr int ROW = FUSE ? PA + PB : 2 * PA + PB + PX;
  // rows per stage (fused: the LDS the dY1 / X images took holds twice the rows)
#ifndef MLPW_KC_FUSED
#define MLPW_KC_FUSED 1           // (2: measured 1 % slower)
#endif
#ifndef MLPW_GRAD_GLDS
#define MLPW_GRAD_GLDS 1
#endif
#ifndef MLPW_GRAD_GLDS_F64
#define MLPW_GRAD_GLDS_F64 1
#endif
  // (tried: three stage buffers with the rows requested two stages ahead,
  // counted vmcnt + bare s_barrier -- 0.4 % slower: HBM latency is not what the
  // gradient kernel waits for)
  static constexpr int NBUF = 2;
  static constexpr int KC = WCfg<real>::KC * (FUSE ? MLPW_KC_FUSED : 1);
  static constexpr int BUF = KC * ROW;
  static constexpr size_t bytes() { return sizeof(real) * NBUF * (size_t)BUF + 64; }
};

template <typename real, int H, int KPG>
__global__ __launch_bounds__(512, 1) void mlpw_grad_kernel(WArgs<real> a) {
  typedef typename WV<real>::v4 v4;
  typedef typename WV<real>::acc vacc;
  typedef GradCfg<real, H> G;
  typedef GradLds<real, H, KPG> LD;
  constexpr int KC = LD::KC, NJ = H / 16, NAT = G::NAT, NXT = LD::XW / 16;
  constexpr int NT = 512;
  constexpr int NVA = (KC * G::UR / 4 + NT - 1) / NT;        // staged 4-chunks per thread
  constexpr int NVB = (KC * H / 4 + NT - 1) / NT;
  constexpr int NVX = (KC * LD::XW + NT - 1) / NT;
  constexpr bool FUSE = LD::FUSE;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real* lds = reinterpret_cast<real*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int din = a.din;
  // (row range, unit half) of this workgroup.  NSPLIT = 2 (fp64, H = 256): the two
  // halves of a row range read the same H1 rows; they are placed on the SAME XCD
  // (workgroups go to the XCDs round robin: b and b + 8 meet) and are resident
  // together, so the second read of a stage hits that XCD's L2 instead of HBM.
  int wg_range = blockIdx.x, wg_half = 0;
  const int wg_ranges = gridDim.x / G::NSPLIT;
  if (G::NSPLIT == 2) {
    const int b = blockIdx.x;
#ifndef MLPW_GRAD_PLAIN_PAIRS
    if (wg_ranges % 8 == 0) {
      const int xcd = b & 7, slot = b >> 3;
      wg_range = (slot >> 1) * 8 + xcd;
      wg_half = slot & 1;
    } else
#endif
    {
      wg_range = b % wg_ranges;
      wg_half = b / wg_ranges;
    }
  }
  const int ubase = wg_half * G::UR;
  // rows of this workgroup: a multiple of KC per workgroup
  int64_t per = (a.R + wg_ranges - 1) / wg_ranges;
  per = (per + KC - 1) / KC * KC;
  const int64_t r_lo = wg_range * per;
  const int64_t r_hi = tmin<int64_t>(a.R, r_lo + per);

  vacc acc2[NAT][NJ], acc1[NAT][NXT];
#pragma unroll
  for (int at = 0; at < NAT; ++at) {
#pragma unroll
    for (int bt = 0; bt < NJ; ++bt) acc2[at][bt] = (vacc){0, 0, 0, 0};
#pragma unroll
    for (int xt = 0; xt < NXT; ++xt) acc1[at][xt] = (vacc){0, 0, 0, 0};
  }

  real sb2[NAT], sb1[NAT];
#pragma unroll
  for (int at = 0; at < NAT; ++at) { sb2[at] = 0; sb1[at] = 0; }
  v4 sa2[NVA], sa1[NVA], sb[NVB];
  real sx[NVX];
  // GL: the dY2 / dY1 / H1 rows of a stage go from HBM straight into the LDS
  // (one global_load_lds_dwordx4 per 1 KiB row; X, if staged at all, still
  // through registers).  Rows past R of the last stage come from the workspace
  // like the others: the chain kernel wrote them (whole tiles), dY2 / dY1 as
  // exact zeros.
  constexpr int EPI = 1024 / (int)sizeof(real);              // elements per DMA instruction
  constexpr bool GL = G::UR % EPI == 0 && H % EPI == 0 && KC % 8 == 0 &&
                      (sizeof(real) == 4 ? MLPW_GRAD_GLDS : MLPW_GRAD_GLDS_F64);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  auto dma = [&](int64_t r0, int buf) {
    typedef const __attribute__((address_space(1))) void* gvp;
    typedef __attribute__((address_space(3))) void* lvp;
    real* A2 = lds + buf * LD::BUF;
    real* A1 = A2 + KC * LD::PA;
    real* B2 = FUSE ? A1 : A1 + KC * LD::PA;
#pragma unroll
    for (int q = 0; q < KC / 8; ++q) {
      const int row = wave_s * (KC / 8) + q;
      const int64_t off = (r0 + row) * H + lane * (EPI / 64);
#pragma unroll
      for (int e = 0; e < G::UR / EPI; ++e) {
        __builtin_amdgcn_global_load_lds((gvp)(a.dy2s + off + ubase + e * EPI),
                                         (lvp)(A2 + row * LD::PA + e * EPI), 16, 0, 0);
        if (!FUSE)
          __builtin_amdgcn_global_load_lds((gvp)(a.dy1s + off + ubase + e * EPI),
                                           (lvp)(A1 + row * LD::PA + e * EPI), 16, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < H / EPI; ++e)
        __builtin_amdgcn_global_load_lds((gvp)(a.h1s + off + e * EPI),
                                         (lvp)(B2 + row * LD::PB + e * EPI), 16, 0, 0);
    }
  };
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < (GL ? 0 : NVA); ++q) {
      const int idx = q * NT + tid;
      const int row = idx / (G::UR / 4), c4 = idx - row * (G::UR / 4);
      const int64_t r = r0 + row;
      const bool ok = idx < KC * G::UR / 4 && r < r_hi;
      const int64_t off = (ok ? r : r_lo) * H + ubase + 4 * c4;
      const v4 z = {0, 0, 0, 0};
      sa2[q] = ok ? *reinterpret_cast<const v4*>(a.dy2s + off) : z;
      if (!FUSE) sa1[q] = ok ? *reinterpret_cast<const v4*>(a.dy1s + off) : z;
    }
#pragma unroll
    for (int q = 0; q < (GL ? 0 : NVB); ++q) {
      const int idx = q * NT + tid;
      const int row = idx / (H / 4), c4 = idx - row * (H / 4);
      const int64_t r = r0 + row;
      const bool ok = idx < KC * H / 4 && r < r_hi;
      const v4 z = {0, 0, 0, 0};
      sb[q] = ok ? *reinterpret_cast<const v4*>(a.h1s + (ok ? r : r_lo) * H + 4 * c4) : z;
    }
#pragma unroll
    for (int q = 0; q < (FUSE ? 0 : NVX); ++q) {
      const int idx = q * NT + tid;
      const int row = idx / LD::XW, f = idx - row * LD::XW;
      const int64_t r = r0 + row;
      real val = 0;
      if (idx < KC * LD::XW && r < r_hi && f < din) {
        const int64_t rp = a.row_index ? a.row_index[r] : r;
        const int64_t ne = rp / a.T;
        const int t = (int)(rp - ne * a.T);
        val = a.x[ne * a.env_stride + t * a.row_stride + f];
      }
      sx[q] = val;
    }
  };
  auto stash = [&](int buf) {
    real* A2 = lds + buf * LD::BUF;
    real* A1 = A2 + KC * LD::PA;
    real* B2 = FUSE ? A1 : A1 + KC * LD::PA;
    real* BX = B2 + KC * LD::PB;
#pragma unroll
    for (int q = 0; q < (GL ? 0 : NVA); ++q) {
      const int idx = q * NT + tid;
      if (idx < KC * G::UR / 4) {
        const int row = idx / (G::UR / 4), c4 = idx - row * (G::UR / 4);
        *reinterpret_cast<v4*>(A2 + row * LD::PA + 4 * c4) = sa2[q];
        if (!FUSE) *reinterpret_cast<v4*>(A1 + row * LD::PA + 4 * c4) = sa1[q];
      }
    }
#pragma unroll
    for (int q = 0; q < (GL ? 0 : NVB); ++q) {
      const int idx = q * NT + tid;
      if (idx < KC * H / 4) {
        const int row = idx / (H / 4), c4 = idx - row * (H / 4);
        *reinterpret_cast<v4*>(B2 + row * LD::PB + 4 * c4) = sb[q];
      }
    }
#pragma unroll
    for (int q = 0; q < (FUSE ? 0 : NVX); ++q) {
      const int idx = q * NT + tid;
      if (idx < KC * LD::XW) {
        const int row = idx / LD::XW, f = idx - row * LD::XW;
        BX[row * LD::PX + f] = sx[q];
      }
    }
  };

  int cur = 0;
  if (r_lo < r_hi) {
    if (GL) dma(r_lo, 0);
    fetch(r_lo);
    stash(0);
  }
  __syncthreads();
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += KC) {
    const bool more = r0 + KC < r_hi;
    if (more) {
      if (GL) dma(r0 + KC, cur ^ 1);         // free since the last barrier
      fetch(r0 + KC);
    }
    const real* A2 = lds + cur * LD::BUF;
    const real* A1 = A2 + KC * LD::PA;
    const real* B2 = FUSE ? A1 : A1 + KC * LD::PA;
    const real* BX = B2 + KC * LD::PB;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int row = 4 * ks + g;
      real a2[NAT], a1[NAT];
#pragma unroll
      for (int at = 0; at < NAT; ++at) {
        a2[at] = A2[row * LD::PA + wave * G::UW + 16 * at + m];
        sb2[at] += a2[at];                                    // db2, db1: column sums
        if (!FUSE) {
          a1[at] = A1[row * LD::PA + wave * G::UW + 16 * at + m];
          sb1[at] += a1[at];
        }
      }
#pragma unroll
      for (int bt = 0; bt < NJ; ++bt) {
        const real b = B2[row * LD::PB + 16 * bt + m];
#pragma unroll
        for (int at = 0; at < NAT; ++at) acc2[at][bt] = wmfma(a2[at], b, acc2[at][bt]);
      }
#pragma unroll
      for (int xt = 0; xt < (FUSE ? 0 : NXT); ++xt) {
        const real b = BX[row * LD::PX + 16 * xt + m];
#pragma unroll
        for (int at = 0; at < NAT; ++at) acc1[at][xt] = wmfma(a1[at], b, acc1[at][xt]);
      }
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- slab sections W1, b1, W2, b2 (unit order) of this workgroup's units
  real* out = a.partials + (int64_t)wg_range * (a.P + 2);
  real* oW1 = out;
  real* ob1 = out + (int64_t)H * din;
  real* oW2 = ob1 + H;
  real* ob2 = oW2 + (int64_t)H * H;
#pragma unroll
  for (int at = 0; at < NAT; ++at) {
    real v2 = sb2[at], v1 = sb1[at];                          // rows 4 ks + g: sum over g
    v2 += __shfl_xor(v2, 16, 64);
    v2 += __shfl_xor(v2, 32, 64);
    v1 += __shfl_xor(v1, 16, 64);
    v1 += __shfl_xor(v1, 32, 64);
    if (g == 0) {
      const int u = unit_of_pos<real>(ubase + wave * G::UW + 16 * at + m);
      ob2[u] = v2;
      if (!FUSE) ob1[u] = v1;
    }
  }
#pragma unroll
  for (int at = 0; at < NAT; ++at)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pa = ubase + wave * G::UW + 16 * at + drow<real>(g, i);
      const int ua = unit_of_pos<real>(pa);
#pragma unroll
      for (int bt = 0; bt < NJ; ++bt)
        oW2[(int64_t)ua * H + unit_of_pos<real>(16 * bt + m)] = acc2[at][bt][i];
#pragma unroll
      for (int xt = 0; xt < (FUSE ? 0 : NXT); ++xt) {
        const int f = 16 * xt + m;
        if (f < din) oW1[(int64_t)ua * din + f] = acc1[at][xt][i];
      }
    }
}

// ---------------------------------------------------------------------------
// slab reduction (+ Adam), the arithmetic of mlp_finish_kernel for either type
// ---------------------------------------------------------------------------
template <typename real>
struct WAdam {
  real *param, *m, *v, *state;
  real lr, b1, b2, eps, wd, step;
};

constexpr int WFIN_GROUPS = 16;

// (env shards: gradient only here, the exchange + Adam follow as one small launch --
// see mlp_finish_kernel, csrc/mlp_shared.h)
template <typename real>
__global__ __launch_bounds__(64 * WFIN_GROUPS) void mlpw_finish_kernel(
    const real* __restrict__ partials, int nparts, int P, int64_t R, real* __restrict__ grad,
    real* __restrict__ stats, WAdam<real> ad) {
  __shared__ real part[WFIN_GROUPS][64];
  __shared__ real red[WFIN_GROUPS];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + col;
  real s = 0;
  if (p < P + 1) {
    const real* src = partials + p;
#pragma unroll 4
    for (int i = grp; i < nparts; i += WFIN_GROUPS) s += src[(int64_t)i * (P + 2)];
  }
  part[grp][col] = s;
  __syncthreads();
  real sq = 0;
  if (grp == 0 && p < P + 1) {
    real g0 = 0;
#pragma unroll
    for (int k = 0; k < WFIN_GROUPS; ++k) g0 += part[k][col];
    if (p < P) {
      grad[p] = g0;
      sq = g0 * g0;
      if (ad.param) {
        real w = ad.param[p], mi = ad.m[p], vi = ad.v[p], step_size, bc2s;
        adam_coef(ad.lr, ad.b1, ad.b2, ad.step, step_size, bc2s);
        adam_elem(g0, w, mi, vi, ad.b1, ad.b2, ad.eps, ad.wd, step_size, bc2s);
        ad.m[p] = mi;
        ad.v[p] = vi;
        ad.param[p] = w;
      }
    } else {
      stats[0] = g0 / (real)R;
    }
  }
  const real tot = block_sum(sq, red);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[1], tot);
    if (ad.param && blockIdx.x == 0) ad.state[0] = ad.step;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
inline int mlpw_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <typename real, int H, int KPG, int ACT>
int mlpw_launch_act(WArgs<real> a, real* workspace, real* grad, real* stats, int max_wg,
                    const real* w2, WAdam<real> ad, hipStream_t st) {
  typedef WCfg<real> C;
  const bool bwd = a.partials != nullptr;
  const int64_t ntiles = (a.R + C::TILE - 1) / C::TILE;
  int grid = mlpw_cu_count();
  if (max_wg > 0 && max_wg < grid) grid = max_wg;
  int cgrid = grid * C::WGS;                                // chain kernel: WGS workgroups per CU
  if (ntiles < cgrid) cgrid = (int)ntiles;
  if (ntiles < grid) grid = (int)ntiles;
  // workspace: w2p | w2tp | h1s | dy2s | dy1s
  real* w2p = workspace;
  real* w2tp = w2p + (int64_t)H * H;
  a.w2p = w2p;
  a.w2tp = w2tp;
  a.h1s = w2tp + (int64_t)H * H;
  const int64_t rows = mlpw_ws_rows(a.R);                   // whole tiles
  a.dy2s = a.h1s + rows * H;
  a.dy1s = a.dy2s + rows * H;
  hipLaunchKernelGGL(mlpw_prep_kernel<real>, dim3((H * H + 255) / 256), dim3(256), 0, st, w2,
                     H, w2p, w2tp);
  TCE_LAUNCH_CHECK();
  const size_t lds = ChainLds<real, H, KPG>::bytes(bwd);
  if (bwd) {
    static bool set = false;
    if (!set) {
      tce_lds_limit(reinterpret_cast<const void*>(&mlpw_chain_kernel<real, H, KPG, ACT, true>), (size_t)(lds));
      const int glds0 = (int)GradLds<real, H, KPG>::bytes();
      tce_lds_limit(reinterpret_cast<const void*>(&mlpw_grad_kernel<real, H, KPG>), (size_t)(glds0));
      set = true;
    }
    hipLaunchKernelGGL((mlpw_chain_kernel<real, H, KPG, ACT, true>), dim3(cgrid), dim3(C::NT), lds,
                       st, a);
    TCE_LAUNCH_CHECK();
    const dim3 ggrid(grid * GradCfg<real, H>::NSPLIT);
    const size_t glds = GradLds<real, H, KPG>::bytes();
    hipLaunchKernelGGL((mlpw_grad_kernel<real, H, KPG>), ggrid, dim3(512), glds, st, a);
    TCE_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlpw_finish_kernel<real>, dim3((a.P + 1 + 63) / 64),
                       dim3(64 * WFIN_GROUPS), 0, st, a.partials, grid, a.P, a.R, grad, stats,
                       ad);
    TCE_LAUNCH_CHECK();
  } else {
    static bool set = false;
    if (!set) {
      tce_lds_limit(reinterpret_cast<const void*>(&mlpw_chain_kernel<real, H, KPG, ACT, false>), (size_t)(lds));
      set = true;
    }
    hipLaunchKernelGGL((mlpw_chain_kernel<real, H, KPG, ACT, false>), dim3(cgrid), dim3(C::NT), lds,
                       st, a);
    TCE_LAUNCH_CHECK();
  }
  return 0;
}

template <typename real, int H, int KPG>
int mlpw_launch(WArgs<real> a, real* workspace, real* grad, real* stats, int max_wg,
                const real* w2, WAdam<real> ad, hipStream_t st) {
  switch (a.act) {
#ifdef MLPW_ONLY_LEAKY
    case W_LEAKY:
      return mlpw_launch_act<real, H, KPG, W_LEAKY>(a, workspace, grad, stats, max_wg, w2, ad, st);
  }
  tce_set_error("mlpw_critic: activation not built");
  return 1;
}
template <typename real, int H, int KPG>
int mlpw_launch_unused(WArgs<real> a, real* workspace, real* grad, real* stats, int max_wg,
                       const real* w2, WAdam<real> ad, hipStream_t st) {
  switch (a.act) {
#endif
#ifndef MLPW_ONLY_RELU
    case W_TANH:
      return mlpw_launch_act<real, H, KPG, W_TANH>(a, workspace, grad, stats, max_wg, w2, ad, st);
    case W_LEAKY:
      return mlpw_launch_act<real, H, KPG, W_LEAKY>(a, workspace, grad, stats, max_wg, w2, ad, st);
    case W_SOFTPLUS:
      return mlpw_launch_act<real, H, KPG, W_SOFTPLUS>(a, workspace, grad, stats, max_wg, w2, ad,
                                                       st);
#endif
    case W_RELU:
      return mlpw_launch_act<real, H, KPG, W_RELU>(a, workspace, grad, stats, max_wg, w2, ad, st);
  }
  tce_set_error("mlpw_critic: activation not built");
  return 1;
}

}  // namespace
