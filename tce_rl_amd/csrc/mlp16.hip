// Fused critic MLP epoch on the f16 matrix cores with SPLIT operands
// (v_mfma_f32_16x16x32_f16, fp32 accumulate): the same launch as mlp.hip --
// forward + value loss + backward + per-workgroup gradient slabs for the value
// network D_in -> 128 -> 128 -> 1 (mprl/rl/agent/temporal_correlated_agent.py:
// 343-366, mprl/util/util_nn.py:225-246) -- at fp32-grade accuracy and 1/5 of
// the matrix-core cycles of the exact-fp32 form.
//
// Arithmetic.  Every fp32 operand x is carried as two f16 numbers
//     hi = f16(x),   lo = f16((x - hi) * 2^11)        x = hi + 2^-11 lo (22+ bits)
// and a product of two operands as three MFMAs into two fp32 accumulators
//     M += a_hi b_hi        X += a_hi b_lo + a_lo b_hi       a b ~ M + 2^-11 X
// (the dropped a_lo b_lo term is <= 2^-22 relative).  f16 x f16 products are
// exact in the matrix core and accumulate in fp32, so the result differs from an
// fp32 FMA chain by a few 1e-7 relative -- the size of the fp32 summation-order
// noise itself (tests/test_mlp16_gpu.py measures both against fp64).  The
// backward signal is scaled by a power of two G ~ R / 2 before it is split (so
// dL/dv ~ 1e-6 does not sink below the f16 range) and the weight-gradient
// accumulators are unscaled by 1 / G on the way out, both exact.
//
// Orientation, as in mlp.hip: activations transposed ([hidden x batch], batch
// column on the lane).  A 16x16 result tile converted pairwise to f16 is the B
// operand of the next layer's MFMA with the k order (lane group g, element j)
// <-> hidden unit 32 kb + 16 (j >> 2) + 4 g + (j & 3); the weights are read in
// that order, so the chains X -> H1 -> H2 -> v and dY2 -> dH1 stay in
// registers.  Weight gradients contract over the batch, which sits on the lanes:
// the tiles go once to LDS as [batch][unit] f16 images and come back through
// the hardware transpose read ds_read_b64_tr_b16 (4 batch rows x 16 units per
// 16-lane group), as A and B fragments with k = batch.
//
// LDS (157,184 B): W1 image [unit][48 features] x {hi, lo}; W2 image [h2][h1]
// x {hi, lo} -- ONE image serves the forward (8-byte row reads) and, through
// the transpose read, dH1 = W2^T dY2; biases / w3 in fp32; two [64 batch][128]
// x {hi, lo} images (H1 then X; dY2 then dY1).  All images sit at power-of-two
// pitches with XOR swizzles found by scripts/lds_banks.py: every read and
// write pattern of the kernel is bank-conflict free.
//
// Roles and phases are those of mlp_critic_bwd_kernel (mlp.hip): waves 0-3 run
// the chains of 16 rows each, waves 4-7 own the weight-gradient accumulators.
#include "mlp_shared.h"

extern "C" int tce_xchg_adam_f32(void* xchg, float* param, float* grad, float* m, float* v,
                                 int64_t n, float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 fp4 __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) Frag { u32 r[4]; };   // 8 f16: one A or B operand (read from LDS as 16 bytes)

#ifndef M16_RING
#define M16_RING 3          // operand ring of the chain loops (reads run RING - 1 steps ahead)
#endif
#ifndef M16_GRING
#define M16_GRING 2         // B-operand ring of the gradient loops
#endif
constexpr int RING = M16_RING, GRING = M16_GRING;
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;
constexpr int P1B = 96, P2B = 256, PTB = 256, PXB = 128;      // image pitches, bytes
constexpr int W1_IMG = HID * P1B, W2_IMG = HID * P2B;        // bytes per part
constexpr int T_IMG = ROWS_PER_TILE * PTB, X_IMG = ROWS_PER_TILE * PXB;
constexpr int OFF_W1 = 0;
constexpr int OFF_W2 = OFF_W1 + 2 * W1_IMG;
constexpr int OFF_BS = OFF_W2 + 2 * W2_IMG;                  // b1 | b2 | w3 (fp32)
constexpr int OFF_TH = OFF_BS + 3 * HID * 4;                 // H1 image, later X image
constexpr int OFF_TD = OFF_TH + 2 * T_IMG;                   // dY2 image, later dY1
constexpr int LDS16_BYTES = OFF_TD + 2 * T_IMG;
static_assert(LDS16_BYTES <= 160 * 1024, "LDS budget");
static_assert(OFF_W2 % 256 == 0 && OFF_TH % 256 == 0 && OFF_TD % 256 == 0, "image alignment");

// byte-offset swizzles inside an image row (row = image row index)
__device__ inline int swz2(int row) { return ((row & 7) << 5) ^ (((row >> 3) & 1) << 4); }
__device__ inline int swzT(int row) {
  return ((row & 3) << 5) ^ (((row >> 2) & 1) * 0x88) ^ (((row >> 3) & 1) << 4);
}
__device__ inline int swzX(int row) { return (row & 7) << 4; }

__device__ inline u32 pk_f16(float a, float b) {
  const h2v v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(u32, v);
}
// (a, b) -> packed hi parts and packed scaled lo parts (v_cvt_pk_f16_f32, two
// v_cvt_f32_f16, the residuals, v_cvt_pk_f16_f32)
__device__ inline void split2(float a, float b, u32& hi, u32& lo) {
  const f32x2 v = {a, b};
  const h2v h = __builtin_convertvector(v, h2v);
  const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * LO_SCALE;
  hi = __builtin_bit_cast(u32, h);
  lo = __builtin_bit_cast(u32, __builtin_convertvector(r, h2v));
}
// element `half` (0 / 1) of a packed {hi, lo} pair back to fp32
__device__ inline float join_parts(u32 hi, u32 lo, int half) {
  const h2v h = __builtin_bit_cast(h2v, hi), l = __builtin_bit_cast(h2v, lo);
  return (float)h[half] + (float)l[half] * LO_INV;
}
__device__ inline f32x4 mfma16(const Frag& a, const Frag& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b),
                                                c, 0, 0, 0);
}
__device__ inline void mma3(const Frag& ah, const Frag& al, const Frag& bh, const Frag& bl,
                            f32x4& m, f32x4& x) {
  m = mfma16(ah, bh, m);
  x = mfma16(ah, bl, x);
  x = mfma16(al, bh, x);
}
__device__ inline u32x2 lds_rd64(const char* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ inline u32x2 lds_tr64(const char* p) {
  const fp4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
      (__attribute__((address_space(3))) fp4*)(p));
  return __builtin_bit_cast(u32x2, v);
}
// a value the optimizer must treat as new here: keeps per-phase address
// arithmetic (one v_xad_u32 per read) from being hoisted out of the tile loop
// as dozens of loop-invariant registers
__device__ inline int fresh(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ inline float freshf(float v) { asm volatile("" : "+v"(v)); return v; }
// nothing moves across: keeps the reads of a later step ahead of this step's MFMAs
__device__ inline void fence_sched() { __builtin_amdgcn_sched_barrier(0); }
__device__ inline void put2(Frag& f, int s, u32x2 v) { f.r[2 * s] = v.x; f.r[2 * s + 1] = v.y; }

// 4 result tiles' worth of one lane -> B operand parts of one 32-deep k-step
__device__ inline void pack_parts(const f32x4& t0, const f32x4& t1, Frag& hi, Frag& lo) {
  split2(t0[0], t0[1], hi.r[0], lo.r[0]);
  split2(t0[2], t0[3], hi.r[1], lo.r[1]);
  split2(t1[0], t1[1], hi.r[2], lo.r[2]);
  split2(t1[2], t1[3], hi.r[3], lo.r[3]);
}

struct Mlp16Args {
  MlpArgs a;
  float gscale;          // power of two applied to dL/dv before the split
};

// weights -> f16 {hi, lo} images (all threads of the workgroup)
__device__ inline void stage_weights16(const MlpArgs& a, char* sm, int tid, int nthreads) {
  const int din = a.din;
  for (int e = tid; e < HID * HID; e += nthreads) {
    const int h2 = e >> 7, p = e & 127;
    const float w = a.w2[e];
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * LO_SCALE);
    const int off = h2 * P2B + ((2 * p) ^ swz2(h2));
    *reinterpret_cast<_Float16*>(sm + OFF_W2 + off) = hi;
    *reinterpret_cast<_Float16*>(sm + OFF_W2 + W2_IMG + off) = lo;
  }
  for (int e = tid; e < HID * (P1B / 2); e += nthreads) {
    const int u = e / (P1B / 2), f = e - u * (P1B / 2);
    const float w = f < din ? a.w1[u * din + f] : 0.f;
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * LO_SCALE);
    *reinterpret_cast<_Float16*>(sm + OFF_W1 + u * P1B + 2 * f) = hi;
    *reinterpret_cast<_Float16*>(sm + OFF_W1 + W1_IMG + u * P1B + 2 * f) = lo;
  }
  float* Bs = reinterpret_cast<float*>(sm + OFF_BS);
  for (int e = tid; e < HID; e += nthreads) {
    Bs[e] = a.b1[e];
    Bs[HID + e] = a.b2[e];
    Bs[2 * HID + e] = a.w3[e];
  }
}

// This lane's features of the cursor's row: dst[8 kb + j] = X[r][32 kb + 8 g + j]
// (clamped addresses); returns the clamped row.
template <int NKB1>
__device__ inline int64_t load_x16(const MlpArgs& a, const RowCursor& cur, int g, float* dst) {
  const bool in = cur.r < a.R;
  const int64_t ne = in ? cur.ne : cur.last_ne;
  const int t = in ? cur.t : cur.last_t;
  const float* xr = a.x + ne * a.env_stride + t * a.row_stride;
  if (NKB1 == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      dst[j] = xr[k < a.din ? k : a.din - 1];
    }
  } else {
    // D_in >= 32: the first 32 features exist in every row (one base address,
    // immediate offsets); features 32 .. D_in - 1 sit in lane group 0 only
    const float* xg = xr + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = xg[j];
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[8 + j] = xr[32 + j < a.din ? 32 + j : a.din - 1];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[8 + j] = 0.f;
    }
  }
  return in ? cur.r : a.R - 1;
}

// NKB1: 32-feature k-steps of layer 1: 1 for D_in <= 32, else 2.
template <int ACT, int NKB1>
__global__ __launch_bounds__(2 * MLP_BT, 1) void mlp_critic_bwd16_kernel(Mlp16Args aa) {
  constexpr int NCB = NKB1 == 1 ? 2 : 3;                       // 16-feature blocks of dW1
  constexpr int NKT = HID / 32;                                // 32-deep k-steps over a hidden layer
  extern __shared__ __attribute__((aligned(256))) char sm[];
  const MlpArgs& a = aa.a;
  const int din = a.din;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int role = tid >> 8;                                   // 0 chain, 1 gradient
  const int wave = (tid >> 6) & 3;
  const int c = lane & 15, g = lane >> 4;
  const int q = (lane >> 2) & 3, pp = lane & 3;                // transpose-read roles
  stage_weights16(a, sm, tid, 2 * MLP_BT);
  __syncthreads();
  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const int P = mlp_num_params(din);
  float* out = a.partials + (int64_t)blockIdx.x * (P + 2);
  float* oW1 = out;
  float* ob1 = oW1 + HID * din;
  float* oW2 = ob1 + HID;
  float* ob2 = oW2 + HID * HID;
  float* ow3 = ob2 + HID;
  float* ob3 = ow3 + HID;
  __shared__ float sc[8];
  float* red = reinterpret_cast<float*>(sm + OFF_TH);          // [HID][4 waves] (after the tile loop)
  // k rows of a transpose read: batch row (or h2 unit) 32 kb + 16 s + 4 g + q
  const int krow = 4 * g + q;

  if (role == 0) {
    // ======================= chain waves =======================
    const float b3 = a.b3[0];
    const float inv_n = 1.f / (float)a.R;
    const float gs = aa.gscale;
    float gw3[NB][4];
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) gw3[m][i] = 0.f;
    float gb3 = 0.f, loss_sum = 0.f;
    const int trow = wave * 16 + c;                            // this lane's row in the tile images
    const int tsw0 = swzT(trow);
    const char* w1rd = sm + OFF_W1 + c * P1B + 16 * g;         // + 16 mb P1B + 64 kb
    const int w2sw0 = swz2(c);                                 // forward rows 16 mb + c
    const int w2tsw0 = swz2(krow);                             // transposed rows 32 kb + 16 s + 4 g + q
    RowCursor cur(a, blockIdx.x, wave, c);
    float xn[8 * NKB1], retn, oldn = 0.f;
    {
      const int64_t rcn = load_x16<NKB1>(a, cur, g, xn);
      retn = a.ret[rcn];
      if (a.clip > 0.f) oldn = a.old_v[rcn];
    }
#ifdef M16_STAMP
    long long stt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
#define STAMP(k) { const long long tn = __builtin_readcyclecounter(); stt[k] += tn - tprev; tprev = tn; }
#else
#define STAMP(k)
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t r = cur.r;
      const bool rok = r < a.R;
      // the bias / w3 reads stay inside the tile (hoisted out of the loop they
      // would pin 96 registers): the offset is opaque to the optimizer
      int bs_off = OFF_BS;
      asm volatile("" : "+v"(bs_off));
      const float* Bs = reinterpret_cast<const float*>(sm + bs_off);
      // ---- P1: forward chain
      // (features past D_in hold finite duplicates: their W1 columns are zero and
      // their dW1 columns are dropped; rows past R get dL/dv = 0 below)
      Frag Xh[NKB1], Xl[NKB1];
#pragma unroll
      for (int kb = 0; kb < NKB1; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          split2(xn[8 * kb + 2 * i], xn[8 * kb + 2 * i + 1], Xh[kb].r[i], Xl[kb].r[i]);
      const float rt = retn, ov = oldn;
      STAMP(0)
      // F2: Y1^T = W1 X^T + b1, packed at once to the B operands of layer 2.
      // Step st = (unit block mb, k-step kb); operand reads run 2 steps ahead.
      Frag H1h[NKT], H1l[NKT];
      {
        constexpr int NST = NB * NKB1;
        Frag Ah[RING], Al[RING];
        auto ld = [&](int st, int b) {
          const char* p = w1rd + 16 * (st / NKB1) * P1B + 64 * (st % NKB1);
          Ah[b] = *reinterpret_cast<const Frag*>(p);
          Al[b] = *reinterpret_cast<const Frag*>(p + W1_IMG);
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) ld(i, i);
        f32x4 accM, accX, tprev;
        f32x4 biasn = *reinterpret_cast<const f32x4*>(Bs + 4 * g);      // read one block ahead
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          const int mb = st / NKB1, kb = st % NKB1;
          if (st + RING - 1 < NST) ld(st + RING - 1, (st + RING - 1) % RING);
          f32x4 biasc = biasn;
          if (kb == 0 && mb + 1 < NB) biasn = *reinterpret_cast<const f32x4*>(Bs + 16 * (mb + 1) + 4 * g);
          fence_sched();
          if (kb == 0) {
            accM = biasc;
            accX = (f32x4){0, 0, 0, 0};
          }
          mma3(Ah[st % RING], Al[st % RING], Xh[kb], Xl[kb], accM, accX);
          if (kb == NKB1 - 1) {
            f32x4 t;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = act_f<ACT>(accM[i] + accX[i] * LO_INV);
            if (mb & 1) pack_parts(tprev, t, H1h[mb >> 1], H1l[mb >> 1]);
            else tprev = t;
          }
        }
      }
      STAMP(1)
      // F4: Y2^T = W2 H1^T + b2.  Step st = (h2 block mb, k-step kb).
      f32x4 h2[NB];
      float vdot = 0.f;                    // w3 . H2 of this lane's units
      {
        const int w2sw = fresh(w2sw0);
        Frag Ah[RING], Al[RING];
        auto ld = [&](int st, int b) {
          const char* rowp = sm + OFF_W2 + (16 * (st >> 2) + c) * P2B;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int off = (64 * (st & 3) + 32 * s + 8 * g) ^ w2sw;
            put2(Ah[b], s, lds_rd64(rowp + off));
            put2(Al[b], s, lds_rd64(rowp + W2_IMG + off));
          }
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) ld(i, i);
        f32x4 accM, accX, w3v;
        f32x4 biasn = *reinterpret_cast<const f32x4*>(Bs + HID + 4 * g);  // read one block ahead
#pragma unroll
        for (int st = 0; st < NB * NKT; ++st) {
          const int mb = st >> 2, kb = st & 3;
#ifndef M16_F4_NOLD
          if (st + RING - 1 < NB * NKT) ld(st + RING - 1, (st + RING - 1) % RING);
#endif
          f32x4 biasc = biasn;
          if (kb == 0) {
            if (mb + 1 < NB) biasn = *reinterpret_cast<const f32x4*>(Bs + HID + 16 * (mb + 1) + 4 * g);
            w3v = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + 16 * mb + 4 * g);
          }
          fence_sched();
          if (kb == 0) {
            accM = biasc;
            accX = (f32x4){0, 0, 0, 0};
          }
#ifdef M16_F4_NOMFMA
          asm volatile("" :: "v"(Ah[st % RING].r[0]), "v"(Ah[st % RING].r[1]), "v"(Ah[st % RING].r[2]), "v"(Ah[st % RING].r[3]),
                       "v"(Al[st % RING].r[0]), "v"(Al[st % RING].r[1]), "v"(Al[st % RING].r[2]), "v"(Al[st % RING].r[3]));
#else
          mma3(Ah[st % RING], Al[st % RING], H1h[kb], H1l[kb], accM, accX);
#endif
          if (kb == NKT - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h2[mb][i] = act_f<ACT>(accM[i] + accX[i] * LO_INV);
              vdot += w3v[i] * h2[mb][i];
            }
          }
        }
      }
      STAMP(2)
      // value, loss and dL/dv (mean over ALL rows R of the epoch)
      float dv;
      {
        const float v = sum_lane_groups(vdot) + b3;
        if (a.values && rok && g == 0) a.values[r] = v;
        const float e = v - rt;
        float l = e * e, d = 2.f * e;
        if (a.clip > 0.f) {
          const float dlt = v - ov;
          const float cl = fminf(fmaxf(dlt, -a.clip), a.clip);
          const float e2 = ov + cl - rt;
          if (e2 * e2 > l) { l = e2 * e2; d = (dlt > -a.clip && dlt < a.clip) ? 2.f * e2 : 0.f; }
        }
        if (!rok) { l = 0.f; d = 0.f; }
        dv = d * inv_n;
        if (g == 0) loss_sum += l;
        if (g == 0) gb3 += dv;
      }
      // G dY2 = G dv w3 act'(H2), packed at once to the B operands of dH1; dw3 partials
      Frag Dh[NKT], Dl[NKT];
      {
        const float dvs = dv * gs;
#pragma unroll
        for (int kb = 0; kb < NKT; ++kb) {
          f32x4 t[2];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int m = 2 * kb + ps;
            const f32x4 w3v = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + 16 * m + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float hv = freshf(h2[m][i]);
              gw3[m][i] += dv * hv;
              t[ps][i] = dvs * w3v[i] * act_d<ACT>(hv);
            }
          }
          pack_parts(t[0], t[1], Dh[kb], Dl[kb]);
          fence_sched();
        }
      }
      STAMP(3)
      __syncthreads();                     // end P1: gradient waves finished dW1(i-1)
      STAMP(4)
      // ---- P2: [batch][unit] images of H1 and dY2; the next tile's rows are fetched meanwhile
      {
        cur.advance(a.T);
        const int64_t rcn = load_x16<NKB1>(a, cur, g, xn);
        retn = a.ret[rcn];
        if (a.clip > 0.f) oldn = a.old_v[rcn];
      }
      {
        const int tsw = fresh(tsw0);
        char* th = sm + OFF_TH + trow * PTB;
        char* td = sm + OFF_TD + trow * PTB;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
          const int off = (32 * m + 8 * g) ^ tsw;
          const int kb = m >> 1, s = m & 1;
          *reinterpret_cast<u32x2*>(th + off) = (u32x2){H1h[kb].r[2 * s], H1h[kb].r[2 * s + 1]};
          *reinterpret_cast<u32x2*>(th + T_IMG + off) = (u32x2){H1l[kb].r[2 * s], H1l[kb].r[2 * s + 1]};
          *reinterpret_cast<u32x2*>(td + off) = (u32x2){Dh[kb].r[2 * s], Dh[kb].r[2 * s + 1]};
          *reinterpret_cast<u32x2*>(td + T_IMG + off) = (u32x2){Dl[kb].r[2 * s], Dl[kb].r[2 * s + 1]};
        }
      }
      STAMP(5)
      __syncthreads();                     // end P2
      STAMP(6)
      // ---- P3: dH1^T = W2^T dY2^T (A = W2 through the transpose read), dY1.
      // Step st = (hidden-1 block pb, k-step kb over h2).
      Frag E1h[NKT], E1l[NKT];
      {
        const int w2tsw = fresh(w2tsw0);
        Frag Ah[RING], Al[RING];
        auto ld = [&](int st, int b) {
          const int pb = st >> 2, kb = st & 3;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const char* p = sm + OFF_W2 + (32 * kb + 16 * s + krow) * P2B + ((32 * pb + 8 * pp) ^ w2tsw);
            put2(Ah[b], s, lds_tr64(p));
            put2(Al[b], s, lds_tr64(p + W2_IMG));
          }
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) ld(i, i);
        f32x4 accM, accX, dprev;
        // H1 of block pb (this lane's own stores of P2), for act'
        const int tswr = fresh(tsw0);
        const char* hrow = sm + OFF_TH + trow * PTB;
        u32x2 hh[2], hl[2];
#pragma unroll
        for (int st = 0; st < NB * NKT; ++st) {
          const int pb = st >> 2, kb = st & 3;
          if (st + RING - 1 < NB * NKT) ld(st + RING - 1, (st + RING - 1) % RING);
          if (kb == 0) {
            const int off = (32 * pb + 8 * g) ^ tswr;
            hh[pb & 1] = lds_rd64(hrow + off);
            hl[pb & 1] = lds_rd64(hrow + T_IMG + off);
          }
          fence_sched();
          if (kb == 0) { accM = (f32x4){0, 0, 0, 0}; accX = (f32x4){0, 0, 0, 0}; }
          mma3(Ah[st % RING], Al[st % RING], Dh[kb], Dl[kb], accM, accX);
          if (kb == NKT - 1) {
            const int pb2 = pb >> 1, ps = pb & 1;
            f32x4 d;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              d[i] = (accM[i] + accX[i] * LO_INV) *
                     act_d<ACT>(join_parts(hh[pb & 1][i >> 1], hl[pb & 1][i >> 1], i & 1));
            if (ps) pack_parts(dprev, d, E1h[pb2], E1l[pb2]);
            else dprev = d;
          }
        }
      }
      STAMP(7)
      __syncthreads();                     // end P3: gradient waves finished dW2(i)
      STAMP(8)
      // ---- P4: dY1 image over dY2, X image over H1
      {
        const int tsw = fresh(tsw0);
        char* td = sm + OFF_TD + trow * PTB;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
          const int off = (32 * m + 8 * g) ^ tsw;
          const int kb = m >> 1, s = m & 1;
          *reinterpret_cast<u32x2*>(td + off) = (u32x2){E1h[kb].r[2 * s], E1h[kb].r[2 * s + 1]};
          *reinterpret_cast<u32x2*>(td + T_IMG + off) = (u32x2){E1l[kb].r[2 * s], E1l[kb].r[2 * s + 1]};
        }
        char* xs = sm + OFF_TH + trow * PXB;
        const int xsw = swzX(trow);
#pragma unroll
        for (int kb = 0; kb < NKB1; ++kb) {
          if (kb == 0 || g < 2) {
            const int off = (64 * kb + 16 * g) ^ xsw;
            *reinterpret_cast<u32x4*>(xs + off) = (u32x4){Xh[kb].r[0], Xh[kb].r[1], Xh[kb].r[2], Xh[kb].r[3]};
            *reinterpret_cast<u32x4*>(xs + X_IMG + off) = (u32x4){Xl[kb].r[0], Xl[kb].r[1], Xl[kb].r[2], Xl[kb].r[3]};
          }
        }
      }
      STAMP(9)
      __syncthreads();                     // end P4
      STAMP(10)
    }
    __syncthreads();                       // gradient waves: dW1 of the last tile
    // ---- dw3: reduce over the 16 batch lanes, then over the 4 chain waves
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v3 = gw3[m][i];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v3 += __shfl_xor(v3, off, 16);
        if (c == 0) red[(16 * m + 4 * g + i) * 4 + wave] = v3;
      }
    float s3 = (g == 0) ? gb3 : 0.f, sl = (g == 0) ? loss_sum : 0.f;
    s3 = wave_sum(s3);
    sl = wave_sum(sl);
    if (lane == 0) { sc[wave] = s3; sc[4 + wave] = sl; }
    __syncthreads();
    if (tid < HID) ow3[tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    if (tid == 0) {
      ob3[0] = sc[0] + sc[1] + sc[2] + sc[3];
      ob3[1] = sc[4] + sc[5] + sc[6] + sc[7];   // sum of squared errors of this WG
      ob3[2] = 0.f;
    }
#ifdef M16_STAMP
    __syncthreads();
    if (tid == 0 && blockIdx.x == 0)
      for (int k = 0; k < 12; ++k) out[k] = (float)stt[k];
#endif
  } else {
    // ======================= gradient waves =======================
    // this wave's output rows: unit blocks mb = 2 wave + mi (h2 for dW2, hidden-1 for dW1)
    f32x4 gW2M[2][NB], gW2X[2][NB];      // [mi][p block]: rows 4 g + i, column c
    f32x4 gW1M[2][NCB], gW1X[2][NCB];    // [mi][feature block]
    f32x4 gb2[2], gb1[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int n = 0; n < NB; ++n) { gW2M[mi][n] = (f32x4){0, 0, 0, 0}; gW2X[mi][n] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
      for (int n = 0; n < NCB; ++n) { gW1M[mi][n] = (f32x4){0, 0, 0, 0}; gW1X[mi][n] = (f32x4){0, 0, 0, 0}; }
      gb2[mi] = (f32x4){0, 0, 0, 0};
      gb1[mi] = (f32x4){0, 0, 0, 0};
    }
    const int tsw0 = swzT(krow);
    const int xsw0 = swzX(krow);
    int tsw = tsw0, xsw = xsw0;
    const Frag ones = {{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}};     // 1.0
    const Frag ones_lo = {{0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u}};  // 2^-11
    // A fragments of this wave's two unit blocks from the Td image, k-step kb
    auto load_a = [&](int kb, Frag* ah, Frag* al) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char* p = sm + OFF_TD + (32 * kb + 16 * s + krow) * PTB +
                          ((32 * (2 * wave + mi) + 8 * pp) ^ tsw);
          put2(ah[mi], s, lds_tr64(p));
          put2(al[mi], s, lds_tr64(p + T_IMG));
        }
    };
    // dW1[unit][f] += sum_b dY1[b][unit] X[b][f].  Step st = (batch k-step kb, feature block n).
    auto dw1 = [&]() {
      tsw = fresh(tsw0);
      xsw = fresh(xsw0);
      Frag ah[2], al[2], Bh[GRING], Bl[GRING];
      auto ldb = [&](int st, int b) {
        const int kb = st / NCB, n = st % NCB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char* p = sm + OFF_TH + (32 * kb + 16 * s + krow) * PXB + ((32 * n + 8 * pp) ^ xsw);
          put2(Bh[b], s, lds_tr64(p));
          put2(Bl[b], s, lds_tr64(p + X_IMG));
        }
      };
      load_a(0, ah, al);
#pragma unroll
      for (int i = 0; i < GRING - 1; ++i) ldb(i, i);
#pragma unroll
      for (int st = 0; st < 2 * NCB; ++st) {
        const int n = st % NCB;
        if (st + GRING - 1 < 2 * NCB) ldb(st + GRING - 1, (st + GRING - 1) % GRING);
        fence_sched();
        if (n == 0) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            gb1[mi] = mfma16(ah[mi], ones, gb1[mi]);
            gb1[mi] = mfma16(al[mi], ones_lo, gb1[mi]);
          }
        }
        mma3(ah[0], al[0], Bh[st % GRING], Bl[st % GRING], gW1M[0][n], gW1X[0][n]);
        mma3(ah[1], al[1], Bh[st % GRING], Bl[st % GRING], gW1M[1][n], gW1X[1][n]);
        if (st == NCB - 1) { fence_sched(); load_a(1, ah, al); }
      }
    };
#ifdef M16_STAMP
    long long gst[4] = {0, 0, 0, 0};
    long long gprev = __builtin_readcyclecounter();
#define GSTAMP(k) { const long long tn = __builtin_readcyclecounter(); gst[k] += tn - gprev; gprev = tn; }
#else
#define GSTAMP(k)
#endif
    bool first = true;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      if (!first) dw1();                   // P1: previous tile
      first = false;
      GSTAMP(0)
      __syncthreads();                     // end P1
      __syncthreads();                     // end P2
      GSTAMP(1)
      // ---- P3: dW2[h2][p] += sum_b dY2[b][h2] H1[b][p], db2 through a ones operand.
      // Step st = (batch k-step kb, hidden-1 block n).
      {
        tsw = fresh(tsw0);
        Frag ah[2], al[2], Bh[GRING], Bl[GRING];
        auto ldb = [&](int st, int b) {
          const int kb = st >> 3, n = st & 7;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const char* p = sm + OFF_TH + (32 * kb + 16 * s + krow) * PTB + ((32 * n + 8 * pp) ^ tsw);
            put2(Bh[b], s, lds_tr64(p));
            put2(Bl[b], s, lds_tr64(p + T_IMG));
          }
        };
        load_a(0, ah, al);
#pragma unroll
        for (int i = 0; i < GRING - 1; ++i) ldb(i, i);
#pragma unroll
        for (int st = 0; st < 2 * NB; ++st) {
          const int n = st & 7;
          if (st + GRING - 1 < 2 * NB) ldb(st + GRING - 1, (st + GRING - 1) % GRING);
          fence_sched();
          if (n == 0) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              gb2[mi] = mfma16(ah[mi], ones, gb2[mi]);
              gb2[mi] = mfma16(al[mi], ones_lo, gb2[mi]);
            }
          }
          mma3(ah[0], al[0], Bh[st % GRING], Bl[st % GRING], gW2M[0][n], gW2X[0][n]);
          mma3(ah[1], al[1], Bh[st % GRING], Bl[st % GRING], gW2M[1][n], gW2X[1][n]);
          if (st == NB - 1) { fence_sched(); load_a(1, ah, al); }
        }
      }
      GSTAMP(2)
      __syncthreads();                     // end P3
      __syncthreads();                     // end P4
      GSTAMP(3)
    }
    if (!first) dw1();                     // last tile
    __syncthreads();
    // ---- this workgroup's partial slab: [W1 | b1 | W2 | b2 | w3 | b3 | loss | pad]
    const float ig = 1.f / aa.gscale;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = 16 * (2 * wave + mi) + 4 * g + i;
#pragma unroll
        for (int n = 0; n < NB; ++n)
          oW2[u * HID + 16 * n + c] = (gW2M[mi][n][i] + gW2X[mi][n][i] * LO_INV) * ig;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
          const int f = 16 * n + c;
          const float v = (gW1M[mi][n][i] + gW1X[mi][n][i] * LO_INV) * ig;
          if (f < din) oW1[u * din + f] = v;
        }
        if (c == 0) {
          ob1[u] = gb1[mi][i] * ig;
          ob2[u] = gb2[mi][i] * ig;
        }
      }
    }
    __syncthreads();                       // matches the chain waves' final barrier
#ifdef M16_STAMP
    __syncthreads();
    if (tid == 256 && blockIdx.x == 0)
      for (int k = 0; k < 4; ++k) out[12 + k] = (float)gst[k];
#endif
  }
}

template <int ACT>
void launch16(const Mlp16Args& aa, int grid, hipStream_t st) {
  if (aa.a.din <= 32) {
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_bwd16_kernel<ACT, 1>), (size_t)(LDS16_BYTES));
    hipLaunchKernelGGL((mlp_critic_bwd16_kernel<ACT, 1>), dim3(grid), dim3(2 * MLP_BT), LDS16_BYTES,
                       st, aa);
  } else {
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_bwd16_kernel<ACT, 2>), (size_t)(LDS16_BYTES));
    hipLaunchKernelGGL((mlp_critic_bwd16_kernel<ACT, 2>), dim3(grid), dim3(2 * MLP_BT), LDS16_BYTES,
                       st, aa);
  }
}

}  // namespace

extern "C" {

// tce_mlp_critic_f32's backward launch (same buffers, same slab / gradient /
// stats / fused-Adam contract, partials != NULL required) on the f16 matrix
// cores with split operands: see the file header for the arithmetic.
int tce_mlp_critic_f16x2(const float* x, int64_t env_stride, int64_t row_stride, int T,
                         int64_t R, int din, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3,
                         int act, const float* returns, const float* old_values, float clip,
                         float* values, float* partials, float* grad, float* stats,
                         int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                         float* adam_state, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float adam_step, float grad_scale, void* xchg,
                         void* stream) {
  TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && R > 0 && T > 0,
                "mlp_critic_f16x2: null buffer / bad sizes");
  TCE_CHECK_ARG(din >= 1 && din <= MAX_DIN, "mlp_critic_f16x2: 1 <= D_in <= 40");
  TCE_CHECK_ARG(act >= 0 && act <= 3, "mlp_critic_f16x2: unknown activation");
  TCE_CHECK_ARG(partials && returns && grad && stats, "mlp_critic_f16x2: backward buffers missing");
  TCE_CHECK_ARG(!(clip > 0.f && !old_values), "mlp_critic_f16x2: old values missing");
  TCE_CHECK_ARG(!adam_param || (adam_m && adam_v && adam_state && adam_step >= 1.f),
                "mlp_critic_f16x2: fused Adam needs its state buffers");
  Mlp16Args aa;
  aa.a = MlpArgs{x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, w3, b3,
                 returns, old_values, clip, values, partials, nullptr, nullptr};
  // dL/dv = 2 (v - ret) / R: G = 2^floor(log2 R) / 2 brings it to the order of the error
  int ex = 0;
  (void)frexpf((float)R, &ex);
  aa.gscale = ldexpf(1.f, tmax(ex - 2, 0));
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = ceil_div(R, ROWS_PER_TILE);
  int cap = 256;
  if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
  const int grid = (int)tmin<int64_t>(cap, ntiles);
  switch (act) {
    case 0: launch16<ACT_TANH>(aa, grid, st); break;
    case 1: launch16<ACT_RELU>(aa, grid, st); break;
    case 2: launch16<ACT_LEAKY>(aa, grid, st); break;
    default: launch16<ACT_SOFTPLUS>(aa, grid, st); break;
  }
  TCE_LAUNCH_CHECK();
  const int P = mlp_num_params(din);
  TCE_CHECK_ARG(!xchg || adam_param, "mlp_critic_f16x2: an exchange needs the fused Adam step");
  // env shards: the slab reduction leaves the local gradient, the exchange + Adam
  // follow as ONE small launch (few waiting workgroups; csrc/mlp_shared.h)
  AdamArgs ad{xchg ? nullptr : adam_param, adam_m, adam_v, adam_state, lr, beta1, beta2, eps,
              weight_decay, adam_step};
  hipLaunchKernelGGL(mlp_finish_kernel, dim3((unsigned)ceil_div(P + 1, 64)),
                     dim3(64 * FIN_GROUPS), 0, st, partials, grid, P, R, grad, stats, ad);
  TCE_LAUNCH_CHECK();
  if (xchg)
    return tce_xchg_adam_f32(xchg, adam_param, grad, adam_m, adam_v, P, adam_state, stats + 2,
                             adam_step, lr, beta1, beta2, eps, weight_decay, 0.f, grad_scale,
                             stream);
  return 0;
}

}  // extern "C"
