// Fused critic MLP epoch on the bf16 matrix cores with THREE-PART operands
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the same launch as mlp.hip --
// forward + value loss + backward + per-workgroup gradient slabs for the value
// network D_in -> 128 -> 128 -> 1 (mprl/rl/agent/temporal_correlated_agent.py:
// 343-366, mprl/util/util_nn.py:225-246) -- with operands that are NOT narrower
// than fp32.
//
// Arithmetic.  Every fp32 operand x is carried as three bf16 numbers
//     b0 = bf16(x), b1 = bf16(x - b0), b2 = bf16(x - b0 - b1)    x = b0 + b1 + b2
// EXACTLY (3 x 8 significand bits, the exponent range of fp32: no scaling, no
// range restriction), and a product of two operands as the six partial
// products of order <= 2
//     a b ~ a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)
// each exact in the matrix core, accumulated in fp32.  The three dropped terms
// (a1 b2, a2 b1, a2 b2) are <= 2^-25 |a b| together with round-to-nearest parts:
// measured 5e-10 of the operand scale on 128-term dot products, 35 x below the
// rounding noise of an fp32 FMA chain (1.8e-8) -- the result is as close to
// the fp64 truth as the exact-fp32 kernel's (tests/test_mlpb_gpu.py holds it to
// the same bounds).  Six bf16 MFMAs cost 6/16 of one fp32 MFMA of the same
// shape: the matrix-core floor of an epoch drops from 1.69 ms to 0.60 ms.
//
// Layout (csrc/mlp16.hip keeps the weights in LDS and the activations in
// registers; three-part images of both do not fit 160 KB, so this kernel turns
// it around):
//   * a workgroup is 4 waves, ONE per SIMD, 512 registers each; wave w owns the
//     hidden units [32 w, 32 w + 32) of BOTH layers for the whole launch: its
//     slices of W1 (three parts) and of W2 / W2^T (parts 0 and 1, pinned to the
//     accumulation half of the register file, where the matrix core reads A
//     operands from) never move; its rows of dW1 / dW2 sit in accumulators;
//   * the THIRD parts of W2 are one [h2][h1] image in LDS (32 KB): row reads give
//     the forward A fragments, transpose reads those of W2^T -- whose register
//     parts, and dY2 as their B operand, follow the k order of the transpose
//     read.  (All three parts in registers: 240 + 88 accumulators + the working
//     set is 94 % of the register file; the allocator spilled 100 - 190.)
//   * activations live in LDS as [64 batch rows][units] images x 3 parts: X, H1,
//     and dY2 later overwritten by dY1.  Every wave reads ALL of an image as B
//     fragments, computes its 32 units for the 64 rows of the tile and writes its
//     slice of the next image.  The weight gradients contract over the batch: A
//     (own dY slice) and B (H1, X) come back through ds_read_b64_tr_b16;
//   * the unit a result row m of block mb stands for is 32 w + 8 (m >> 2) +
//     4 mb + (m & 3): a lane then holds 8 CONSECUTIVE units of one batch row
//     (both blocks) and stores them with one 16-byte write per part;
//   * images sit at power-of-two pitches with XOR swizzles (scripts/lds_banks.py,
//     scripts/lds_swizzle_search.py): the 16-byte writes, the row reads and the
//     transpose reads of the activation images are bank-conflict free (the
//     transpose reads of the W2 image: 2-way);
//   * every lane role (c, g, q, pp, swizzles) is recomputed from the lane id where
//     a phase starts (LANE_ROLES): as loop invariants they were two dozen
//     registers the allocator spilled -- and a reload in front of an MFMA waits
//     for EVERY outstanding load (one counter), the prefetched rows included.
//   Six workgroup barriers per tile (LDS-only: fence "local" + s_barrier, so that
//   the prefetched rows stay in flight): H1 image | value partials | dY2 image |
//   H1 and dY2 free | X image free | next X image.  Inside a phase the LDS reads
//   of the next step, the epilogue of the previous row block and the three-way
//   splits ride between the MFMAs of the current step (sched_group_barrier).
//   One wave per SIMD is issue-bound: an MFMA holds the port ~10 cycles, a VALU
//   instruction beside it costs ~5 (scripts/probe_mfma_fill.hip) -- KERNELS.md
//   has the cycle budget of a tile and what was tried.
#include "mlp_shared.h"

extern "C" int tce_xchg_adam_f32(void* xchg, float* param, float* grad, float* m, float* v,
                                 int64_t n, float* state, float* norms_out, float step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay,
                                 float clip, float grad_scale, void* stream);

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp4 __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct alignas(16) Frag { u32 r[4]; };     // 8 bf16: one part of an A or B operand
struct Frag3 { Frag p[3]; };               // the three parts

constexpr int PT = 256, PX = 128;                              // image pitches, bytes
constexpr int T_PART = ROWS_PER_TILE * PT, X_PART = ROWS_PER_TILE * PX;
constexpr int OFF_X = 0;                                       // X image
constexpr int OFF_H = OFF_X + 3 * X_PART;                      // H1 image
constexpr int OFF_D = OFF_H + 3 * T_PART;                      // dY2, later dY1
constexpr int OFF_W2P = OFF_D + 3 * T_PART;                    // third parts of W2: [h2][h1] bf16
constexpr int OFF_BS = OFF_W2P + HID * PT;                     // b1 | b2 | w3 (fp32)
constexpr int OFF_VP = OFF_BS + 3 * HID * 4;                   // [64 rows][4 waves] value partials
constexpr int LDSB_BYTES = OFF_VP + ROWS_PER_TILE * 4 * 4;
static_assert(LDSB_BYTES <= 160 * 1024, "LDS budget");

// byte-offset swizzles inside an image row (scripts/lds_banks.py rules)
__device__ inline int swzT(int row) {      // H1 image
  return ((row & 1) << 5) ^ (((row >> 1) & 1) << 6) ^ (((row >> 2) & 1) * 144);
}
__device__ inline int swzD(int row) {      // dY image: + the 8-byte reads of dH1
  return swzT(row) ^ (((row >> 3) & 1) << 7);
}
__device__ inline int swzX(int row) { return (row & 7) << 4; }
__device__ inline int swzW(int row) {      // W2 third-part image (rows = permuted units)
  return ((row & 1) << 6) ^ (((row >> 1) & 1) << 7) ^ (((row >> 3) & 1) << 5) ^ (((row >> 4) & 1) * 48);
}

__device__ inline u32 pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(u32, __builtin_convertvector(v, bf2));
}
// (a, b) -> the packed parts (v_cvt_pk_bf16_f32, shift / mask, v_pk_add_f32: 9 instructions)
__device__ inline void split3(float a, float b, u32& p0, u32& p1, u32& p2) {
  p0 = pk_bf16(a, b);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk_bf16(ra, rb);
  p2 = pk_bf16(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}
// 8 values (this lane's 8 consecutive k) -> one three-part fragment
__device__ inline void split_frag(const float* v, Frag3& f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) split3(v[2 * i], v[2 * i + 1], f.p[0].r[i], f.p[1].r[i], f.p[2].r[i]);
}
__device__ inline f32x4 mfma(const Frag& a, const Frag& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b),
                                                 c, 0, 0, 0);
}
// 2 x 2 result tiles, six partial products each, smallest first; consecutive
// MFMAs go to different accumulators
__device__ inline void mma6x4(const Frag3& a0, const Frag3& a1, const Frag3& b0, const Frag3& b1,
                              f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11) {
#define MLPB_T(i, j)                     \
  c00 = mfma(a0.p[i], b0.p[j], c00);     \
  c01 = mfma(a0.p[i], b1.p[j], c01);     \
  c10 = mfma(a1.p[i], b0.p[j], c10);     \
  c11 = mfma(a1.p[i], b1.p[j], c11);
  MLPB_T(0, 2) MLPB_T(2, 0) MLPB_T(1, 1) MLPB_T(0, 1) MLPB_T(1, 0) MLPB_T(0, 0)
#undef MLPB_T
}
__device__ inline void mma6x2(const Frag3& a0, const Frag3& a1, const Frag3& b, f32x4& c0, f32x4& c1) {
#define MLPB_T(i, j)                  \
  c0 = mfma(a0.p[i], b.p[j], c0);     \
  c1 = mfma(a1.p[i], b.p[j], c1);
  MLPB_T(0, 2) MLPB_T(2, 0) MLPB_T(1, 1) MLPB_T(0, 1) MLPB_T(1, 0) MLPB_T(0, 0)
#undef MLPB_T
}
__device__ inline u32x2 lds_tr64(const char* p) {
  const fp4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)(p));
  return __builtin_bit_cast(u32x2, v);
}
__device__ inline int fresh(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ inline void pin_acc(Frag& f) {
  u32x4 v = {f.r[0], f.r[1], f.r[2], f.r[3]};
  asm volatile("" : "+a"(v));
  f.r[0] = v.x; f.r[1] = v.y; f.r[2] = v.z; f.r[3] = v.w;
}

// B fragment (three parts) of batch-row block nb, k-step kb: 16-byte row reads
template <int PITCH, int PART>
__device__ inline void ld_b(const char* img, int row, int colb, int swz, Frag3& f) {
  const char* p = img + row * PITCH + (colb ^ swz);
#pragma unroll
  for (int q = 0; q < 3; ++q) f.p[q] = *reinterpret_cast<const Frag*>(p + q * PART);
}
// transposed fragment (k = batch rows 32 kb ..): unit block at byte column colb
template <int PITCH, int PART>
__device__ inline void ld_t(const char* img, int kb, int krow, int colb, int swz, Frag3& f) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const char* p = img + (32 * kb + 16 * s + krow) * PITCH + (colb ^ swz);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u32x2 v = lds_tr64(p + q * PART);
      f.p[q].r[2 * s] = v.x;
      f.p[q].r[2 * s + 1] = v.y;
    }
  }
}
// this lane's 8 consecutive units of one batch row -> the three images
__device__ inline void st_row(char* img, int row, int colb, int swz, const f32x4& lo, const f32x4& hi) {
  Frag q0, q1, q2;
  split3(lo[0], lo[1], q0.r[0], q1.r[0], q2.r[0]);
  split3(lo[2], lo[3], q0.r[1], q1.r[1], q2.r[1]);
  split3(hi[0], hi[1], q0.r[2], q1.r[2], q2.r[2]);
  split3(hi[2], hi[3], q0.r[3], q1.r[3], q2.r[3]);
  char* p = img + row * PT + (colb ^ swz);
  *reinterpret_cast<Frag*>(p) = q0;
  *reinterpret_cast<Frag*>(p + T_PART) = q1;
  *reinterpret_cast<Frag*>(p + 2 * T_PART) = q2;
}

// one B fragment against the two unit blocks: the six partial products of each
// go to two accumulators (p: small terms + a0 b0 ... alternating), so that an
// accumulator is written by every 4th MFMA only
__device__ inline void mma6s(const Frag3& a0, const Frag3& a1, const Frag3& b, f32x4& p0, f32x4& q0,
                             f32x4& p1, f32x4& q1) {
#define MLPB_T(i, j, k, l)          \
  p0 = mfma(a0.p[i], b.p[j], p0);   \
  q0 = mfma(a0.p[k], b.p[l], q0);   \
  p1 = mfma(a1.p[i], b.p[j], p1);   \
  q1 = mfma(a1.p[k], b.p[l], q1);
  MLPB_T(0, 2, 2, 0) MLPB_T(1, 1, 0, 1) MLPB_T(1, 0, 0, 0)
#undef MLPB_T
}
// one step's LDS operands of L2 / dH1: the B fragment and the third parts of the
// two weight fragments
struct PB { Frag3 b; Frag w0, w1; };
// mma6s with the weight parts 0, 1 in registers (a[0], a[1]) and part 2 from LDS
__device__ inline void mma6w(const Frag* a0, const Frag& a02, const Frag* a1, const Frag& a12,
                             const Frag3& b, f32x4& p0, f32x4& q0, f32x4& p1, f32x4& q1) {
  p0 = mfma(a0[0], b.p[2], p0);
  q0 = mfma(a02, b.p[0], q0);
  p1 = mfma(a1[0], b.p[2], p1);
  q1 = mfma(a12, b.p[0], q1);
  p0 = mfma(a0[1], b.p[1], p0);
  q0 = mfma(a0[0], b.p[1], q0);
  p1 = mfma(a1[1], b.p[1], p1);
  q1 = mfma(a1[0], b.p[1], q1);
  p0 = mfma(a0[1], b.p[0], p0);
  q0 = mfma(a0[0], b.p[0], q0);
  p1 = mfma(a1[1], b.p[0], p1);
  q1 = mfma(a1[0], b.p[0], q1);
}
// the lane's roles, recomputed from the lane id where a phase starts: nothing but
// the lane id itself stays live across the tile loop (dozens of loop-invariant
// address registers otherwise: the register file has no room for them)
#define LANE_ROLES()                                                        \
  const int lf_ = fresh(lane);                                              \
  const int c = lf_ & 15, g = lf_ >> 4, q = (lf_ >> 2) & 3, pp = lf_ & 3;   \
  const int krow = 4 * g + q, ocol = 2 * (ub + 8 * g);                      \
  (void)c; (void)g; (void)q; (void)pp; (void)krow; (void)ocol
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// workgroup barrier that orders LDS traffic only: global loads issued before it (the
// next tile's rows) stay in flight across it (__syncthreads waits for them too)
__device__ inline void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// ask the scheduler for N x (1 MFMA, V VALU instructions): the epilogue of the
// previous row block rides in the issue slots the MFMAs of this one leave free
// (an MFMA holds the issue port for 8 of its 16 cycles)
template <int N, int ND, int V, int DPER = 1>
__device__ inline void interleave() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
    if (i * DPER < ND) __builtin_amdgcn_sched_group_barrier(0x100, DPER, 0);    // DS reads
    __builtin_amdgcn_sched_group_barrier(0x002, V, 0);                          // VALU
  }
}

// S steps of (load the B fragments of step s + 1, multiply step s): one
// scheduling region per step, the reads interleaved with the MFMAs by the caller
template <int S, class LD, class MM>
__device__ inline void pipe2(LD ld, MM mm) {
  Frag3 B[2][2];
  ld(0, B[0][0], B[0][1]);
  FENCE();
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) ld(s + 1, B[(s + 1) & 1][0], B[(s + 1) & 1][1]);
    mm(s, B[s & 1][0], B[s & 1][1]);
    FENCE();
  }
}

// NKB1: 32-feature k-steps of layer 1: 1 for D_in <= 32, else 2.
// XV: base and strides of the rows are multiples of 16 bytes: a thread's 16 features come as
// four float4 loads (116 cycles per scalar load instruction were measured at issue: 64 lanes
// of one instruction touch 48 cache lines).
template <int ACT, int NKB1, bool XV>
__global__ __launch_bounds__(MLP_BT, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_critic_bwdb_kernel(MlpArgs a) {
  constexpr int NCB = NKB1 == 1 ? 2 : 3;                       // 16-feature blocks of dW1
  constexpr int NKT = HID / 32;                                // 32-deep k-steps over a hidden layer
  constexpr int NRB = ROWS_PER_TILE / 16;                      // batch-row blocks of a tile
  extern __shared__ __attribute__((aligned(256))) char sm[];
  const int din = a.din;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;                      // (prologue / epilogue; the phases recompute them)
  float* Bs = reinterpret_cast<float*>(sm + OFF_BS);
  float* vpart = reinterpret_cast<float*>(sm + OFF_VP);
  for (int e = tid; e < HID; e += MLP_BT) {
    Bs[e] = a.b1[e];
    Bs[HID + e] = a.b2[e];
    Bs[2 * HID + e] = a.w3[e];
  }
  // ---- third parts of W2 -> ONE [h2][h1] image: row reads give the forward A
  // fragments, transpose reads those of W2^T
  for (int e = tid; e < HID * HID / 2; e += MLP_BT) {
    const int h2 = e >> 6, pc = e & 63;
    u32 q0, q1, q2;
    split3(a.w2[h2 * HID + 2 * pc], a.w2[h2 * HID + 2 * pc + 1], q0, q1, q2);
    *reinterpret_cast<u32*>(sm + OFF_W2P + h2 * PT + ((4 * pc) ^ swzW(h2))) = q2;
  }
  // ---- this wave's weight slices as A fragments (parts 0, 1 of W2 / W2^T, all of
  // W1): row m = c of block mb is unit ub + 8 (c >> 2) + 4 mb + (c & 3).  k order:
  // forward 32 kb + 8 g + j; W2^T follows the transpose read its third part comes
  // through: element 4 s + i <-> h2 = 32 kb + 16 s + 4 g + i
  const int ub = 32 * wave;
  Frag3 W1f[2][NKB1];
  Frag W2f[2][NKT][2], W2t[2][NKT][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int u = ub + 8 * (c >> 2) + 4 * mb + (c & 3);
    float v[8];
    Frag3 t;
#pragma unroll
    for (int kb = 0; kb < NKB1; ++kb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = 32 * kb + 8 * g + j;
        const float w = a.w1[u * din + (f < din ? f : din - 1)];
        v[j] = f < din ? w : 0.f;
      }
      split_frag(v, W1f[mb][kb]);
    }
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a.w2[u * HID + 32 * kb + 8 * g + j];
      split_frag(v, t);
      W2f[mb][kb][0] = t.p[0];
      W2f[mb][kb][1] = t.p[1];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a.w2[(32 * kb + 16 * (j >> 2) + 4 * g + (j & 3)) * HID + u];
      split_frag(v, t);
      W2t[mb][kb][0] = t.p[0];
      W2t[mb][kb][1] = t.p[1];
    }
  }
  // W2 / W2^T fragments are only ever matrix-core operands: pinned to the
  // accumulation half of the register file (the MFMA reads them from there), so
  // that the 256 architectural registers are left to the values the VALU touches
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb)
#pragma unroll
      for (int q3 = 0; q3 < 2; ++q3) {
        pin_acc(W2f[mb][kb][q3]);
        pin_acc(W2t[mb][kb][q3]);
      }
  const int64_t ntiles = (a.R + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const int P = mlp_num_params(din);
  float* out = a.partials + (int64_t)blockIdx.x * (P + 2);
  float* oW1 = out;
  float* ob1 = oW1 + HID * din;
  float* oW2 = ob1 + HID;
  float* ob2 = oW2 + HID * HID;
  float* ow3 = ob2 + HID;
  float* ob3 = ow3 + HID;

  // gradient accumulators: rows = natural units 16 (2 wave + mr) + 4 g + i, column c of block n
  f32x4 gW2[2][NB], gW1[2][NCB];
#pragma unroll
  for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
    for (int n = 0; n < NB; ++n) gW2[mr][n] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int n = 0; n < NCB; ++n) gW1[mr][n] = (f32x4){0, 0, 0, 0};
  }
  // per-lane partial sums over the batch rows this lane sees (unit = ub + 8 g + 4 mb + i)
  f32x4 gb1[2], gb2[2], gw3[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    gb1[mb] = (f32x4){0, 0, 0, 0};
    gb2[mb] = (f32x4){0, 0, 0, 0};
    gw3[mb] = (f32x4){0, 0, 0, 0};
  }
  float gb3 = 0.f, loss_sum = 0.f;
  const float b3 = a.b3[0];
  const float inv_n = 1.f / (float)a.R;

  // ---- X staging: thread -> (row tid >> 2, 16-feature chunk tid & 3): 16 scalar loads with
  // clamped feature indices (float4 loads of the same chunks were tried: no faster, and the
  // register allocation of the dH1 phase got worse); features >= D_in are masked when the
  // image is written
  RowCursor cur(a, blockIdx.x, wave, lane >> 2);
  float xn[16];
  auto load_x = [&]() {
    const int xch = fresh(lane) & 3;
    const bool xact = xch < 2 * NKB1;
    if (xact) {
      const bool in = cur.r < a.R;
      const int64_t ne = in ? cur.ne : cur.last_ne;
      const int t = in ? cur.t : cur.last_t;
      const float* xr = a.x + ne * a.env_stride + t * a.row_stride;
      if (XV) {
        // float4s clamped into the row (what lies past D_in is masked in store_x)
        const int rs4 = fresh((int)a.row_stride) - 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int o = 16 * xch + 4 * m;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + (o < rs4 ? o : rs4));
          xn[4 * m] = v[0]; xn[4 * m + 1] = v[1]; xn[4 * m + 2] = v[2]; xn[4 * m + 3] = v[3];
        }
      } else {
        const int dl = fresh(din);              // (not a loop invariant: 16 hoisted 64-bit offsets otherwise)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int f = 16 * xch + j;
          xn[j] = xr[f < dl ? f : dl - 1];        // (masked in store_x: no use of the value here)
        }
      }
    }
  };
  // the three-way split of the fetched rows (VALU: rides behind the MFMAs of dW1) and, after
  // everyone is done with the X image, their six 16-byte stores
  Frag xq[3][2];
  auto split_x = [&]() {
    const int xch = fresh(lane) & 3;
    if (xch < 2 * NKB1) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 16 * xch + 8 * h + 2 * i;
          split3(f < din ? xn[8 * h + 2 * i] : 0.f, f + 1 < din ? xn[8 * h + 2 * i + 1] : 0.f,
                 xq[0][h].r[i], xq[1][h].r[i], xq[2][h].r[i]);
        }
    }
  };
  auto store_x = [&]() {
    const int lf = fresh(lane);
    const int xch = lf & 3, xrow = 16 * wave + (lf >> 2);
    if (xch < 2 * NKB1) {
      char* xb = sm + OFF_X + xrow * PX;
      const int sw = swzX(xrow);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int off = (32 * xch + 16 * h) ^ sw;
        *reinterpret_cast<Frag*>(xb + off) = xq[0][h];
        *reinterpret_cast<Frag*>(xb + X_PART + off) = xq[1][h];
        *reinterpret_cast<Frag*>(xb + 2 * X_PART + off) = xq[2][h];
      }
    }
  };
  auto advance_x = [&]() { cur.advance(a.T); };
  load_x();
  split_x();
  store_x();
  advance_x();
  __syncthreads();

#ifdef MLPB_STAMP
  long long stt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_readcyclecounter();
#define STAMP(k) { const long long tn = __builtin_readcyclecounter(); stt[k] += tn - tprev; tprev = tn; }
#else
#define STAMP(k)
#endif
  const char* ximg = sm + OFF_X;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    STAMP(15)
    const int64_t r0 = tile * ROWS_PER_TILE;
    float retv[NRB], oldv[NRB];
#pragma unroll
    for (int nb = 0; nb < NRB; ++nb) {
      const int64_t r = r0 + 16 * nb + (fresh(lane) & 15);
      const int64_t rc = r < a.R ? r : a.R - 1;
      retv[nb] = a.ret[rc];
      oldv[nb] = a.clip > 0.f ? a.old_v[rc] : 0.f;
    }

    // ---- L1: H1^T[own units][64 rows] = W1 X^T + b1, one 16-row block at a time
    {
      LANE_ROLES();
      const int swX = swzX(c), swT = swzT(c);
      const f32x4 bia0 = *reinterpret_cast<const f32x4*>(Bs + ub + 8 * g);
      const f32x4 bia1 = *reinterpret_cast<const f32x4*>(Bs + ub + 8 * g + 4);
      f32x4 acc[2][2];                           // [row block parity][unit block]
      Frag hq[3];
      // the epilogue of row block nb in NKB1 pieces: one unit block each (activation and
      // the three-way split of its 4 units), the 16-byte stores with the last
      auto finish = [&](int nb, int piece) {
        const f32x4* r = acc[nb & 1];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          if (NKB1 == 1 || mb == piece) {
            f32x4 h;
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = act_f<ACT>(r[mb][i]);
            split3(h[0], h[1], hq[0].r[2 * mb], hq[1].r[2 * mb], hq[2].r[2 * mb]);
            split3(h[2], h[3], hq[0].r[2 * mb + 1], hq[1].r[2 * mb + 1], hq[2].r[2 * mb + 1]);
          }
        if (piece == NKB1 - 1) {
          char* p = sm + OFF_H + (16 * nb + c) * PT + (ocol ^ swT);
          *reinterpret_cast<Frag*>(p) = hq[0];
          *reinterpret_cast<Frag*>(p + T_PART) = hq[1];
          *reinterpret_cast<Frag*>(p + 2 * T_PART) = hq[2];
        }
      };
      Frag3 B[2];
      int bX[NKB1];
#pragma unroll
      for (int kb = 0; kb < NKB1; ++kb) bX[kb] = fresh(OFF_X + c * PX + ((64 * kb + 16 * g) ^ swX));
      auto ldx = [&](int s, Frag3& b) {
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3)
          b.p[q3] = *reinterpret_cast<const Frag*>(sm + bX[s % NKB1] + 16 * (s / NKB1) * PX + q3 * X_PART);
      };
      ldx(0, B[0]);
      FENCE();
#pragma unroll
      for (int s = 0; s < NRB * NKB1; ++s) {
        const int nb = s / NKB1, kb = s % NKB1;
        if (s + 1 < NRB * NKB1) ldx(s + 1, B[(s + 1) & 1]);
        f32x4* r = acc[nb & 1];
        if (kb == 0) { r[0] = bia0; r[1] = bia1; }
        mma6s(W1f[0][kb], W1f[1][kb], B[s & 1], r[0], r[0], r[1], r[1]);
        if (nb > 0) finish(nb - 1, kb);
        interleave<12, 3, 4>();
        FENCE();
      }
#pragma unroll
      for (int k = 0; k < NKB1; ++k) finish(NRB - 1, k);
    }
    STAMP(0)
    lds_barrier();                             // A: H1 image complete
    STAMP(1)

    // ---- L2: H2^T = W2 H1^T + b2, value partials
    f32x4 h2v[2][NRB];
    {
      LANE_ROLES();
      const int swT = swzT(c);
      const f32x4 bia0 = *reinterpret_cast<const f32x4*>(Bs + HID + ub + 8 * g);
      const f32x4 bia1 = *reinterpret_cast<const f32x4*>(Bs + HID + ub + 8 * g + 4);
      // third parts of this lane's two W2 rows
      const int wrow0 = ub + 8 * (c >> 2) + (c & 3), wrow1 = wrow0 + 4;
      const int sw0 = swzW(wrow0), sw1 = swzW(wrow1);
      f32x4 acc[2][2];
      const f32x4 w30 = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + ub + 8 * g);
      const f32x4 w31 = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + ub + 8 * g + 4);
      float vd = 0.f;
      // piece k of the epilogue of row block nb: result rows i = k of both unit blocks
      auto finish = [&](int nb, int k) {
        const f32x4* r = acc[nb & 1];
        if (k == 0) vd = 0.f;
        h2v[0][nb][k] = act_f<ACT>(r[0][k]);
        h2v[1][nb][k] = act_f<ACT>(r[1][k]);
        vd += w30[k] * h2v[0][nb][k];
        vd += w31[k] * h2v[1][nb][k];
        if (k == 3) {
          const float vs = sum_lane_groups(vd);
          if (g == 0) vpart[(16 * nb + c) * 4 + wave] = vs;
        }
      };
      PB ring[2];
      int bH[NKT], bW0[NKT], bW1[NKT];           // address registers: one per k-step
#pragma unroll
      for (int kb = 0; kb < NKT; ++kb) {
        bH[kb] = fresh(OFF_H + c * PT + ((64 * kb + 16 * g) ^ swT));
        bW0[kb] = fresh(OFF_W2P + wrow0 * PT + ((64 * kb + 16 * g) ^ sw0));
        bW1[kb] = fresh(OFF_W2P + wrow1 * PT + ((64 * kb + 16 * g) ^ sw1));
      }
      auto ld = [&](int s, PB& r) {
        const int nb = s / NKT, kb = s % NKT;
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3)
          r.b.p[q3] = *reinterpret_cast<const Frag*>(sm + bH[kb] + 16 * nb * PT + q3 * T_PART);
        r.w0 = *reinterpret_cast<const Frag*>(sm + bW0[kb]);
        r.w1 = *reinterpret_cast<const Frag*>(sm + bW1[kb]);
      };
      ld(0, ring[0]);
      FENCE();
#pragma unroll
      for (int s = 0; s < NRB * NKT; ++s) {
        const int nb = s / NKT, kb = s % NKT;
        if (s + 1 < NRB * NKT) ld(s + 1, ring[(s + 1) & 1]);
        f32x4* r = acc[nb & 1];
        if (kb == 0) { r[0] = bia0; r[1] = bia1; }
        const PB& rr = ring[s & 1];
        mma6w(W2f[0][kb], rr.w0, W2f[1][kb], rr.w1, rr.b, r[0], r[0], r[1], r[1]);
        if (nb > 0) finish(nb - 1, kb);
        interleave<12, 5, 2>();
        FENCE();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) finish(NRB - 1, k);
    }
    STAMP(2)
    lds_barrier();                             // B: value partials complete
    STAMP(3)

    // ---- value, loss, dL/dv (mean over ALL rows R of the epoch); dY2 = dv w3 act'(H2)
    {
      LANE_ROLES();
      const int swD = swzD(c);
      const f32x4 w30 = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + ub + 8 * g);
      const f32x4 w31 = *reinterpret_cast<const f32x4*>(Bs + 2 * HID + ub + 8 * g + 4);
#pragma unroll
      for (int nb = 0; nb < NRB; ++nb) {
        const f32x4 vp = *reinterpret_cast<const f32x4*>(vpart + (16 * nb + c) * 4);
        const float v = ((vp[0] + vp[1]) + (vp[2] + vp[3])) + b3;
        const int64_t r = r0 + 16 * nb + c;
        const bool rok = r < a.R;
        const float rt = retv[nb], ov = oldv[nb];
        const float e = v - rt;
        float l = e * e, d = 2.f * e;
        if (a.clip > 0.f) {
          const float dlt = v - ov;
          const float cl = fminf(fmaxf(dlt, -a.clip), a.clip);
          const float e2 = ov + cl - rt;
          if (e2 * e2 > l) { l = e2 * e2; d = (dlt > -a.clip && dlt < a.clip) ? 2.f * e2 : 0.f; }
        }
        if (!rok) { l = 0.f; d = 0.f; }
        const float dv = d * inv_n;
        if (wave == 0 && g == 0) {
          if (a.values && rok) a.values[r] = v;
          loss_sum += l;
          gb3 += dv;
        }
        f32x4 d0, d1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ha = h2v[0][nb][i], hb = h2v[1][nb][i];
          gw3[0][i] += dv * ha;
          gw3[1][i] += dv * hb;
          d0[i] = dv * w30[i] * act_d<ACT>(ha);
          d1[i] = dv * w31[i] * act_d<ACT>(hb);
        }
        gb2[0] += d0;
        gb2[1] += d1;
        st_row(sm + OFF_D, 16 * nb + c, ocol, swD, d0, d1);
      }
    }
    STAMP(4)
    lds_barrier();                             // C: dY2 image complete
    STAMP(5)

    // ---- dW2[h2][p] += sum_b dY2[b][h2] H1[b][p]: A = own units of the dY2 image,
    // B = every unit block of the H1 image, both through the transpose read
    {
      LANE_ROLES();
      const int swTk = swzT(krow), swDk = swzD(krow);
      Frag3 A[2][2];
      ld_t<PT, T_PART>(sm + OFF_D, 0, krow, 32 * (2 * wave) + 8 * pp, swDk, A[0][0]);
      ld_t<PT, T_PART>(sm + OFF_D, 0, krow, 32 * (2 * wave + 1) + 8 * pp, swDk, A[0][1]);
      pipe2<2 * (NB / 2)>(
          [&](int s, Frag3& b0, Frag3& b1) {
            const int kb = s / (NB / 2), np = s % (NB / 2);
            if (s == 1) {
              ld_t<PT, T_PART>(sm + OFF_D, 1, krow, 32 * (2 * wave) + 8 * pp, swDk, A[1][0]);
              ld_t<PT, T_PART>(sm + OFF_D, 1, krow, 32 * (2 * wave + 1) + 8 * pp, swDk, A[1][1]);
            }
            ld_t<PT, T_PART>(sm + OFF_H, kb, krow, 32 * (2 * np) + 8 * pp, swTk, b0);
            ld_t<PT, T_PART>(sm + OFF_H, kb, krow, 32 * (2 * np + 1) + 8 * pp, swTk, b1);
          },
          [&](int s, const Frag3& b0, const Frag3& b1) {
            const int kb = s / (NB / 2), np = s % (NB / 2);
            mma6x4(A[kb][0], A[kb][1], b0, b1, gW2[0][2 * np], gW2[0][2 * np + 1], gW2[1][2 * np],
                   gW2[1][2 * np + 1]);
            interleave<24, 24, 1>();
          });
    }
    STAMP(6)
    // ---- dH1^T[own units][rows] = W2^T dY2^T; dY1 = dH1 act'(H1) stays in registers
    Frag dq[NRB][3];                             // dY1 of this lane's 8 units, split, per row block
    {
      LANE_ROLES();
      const int swT = swzT(c), swD = swzD(c);
      const int swWk = swzW(krow);      // rows 32 kb + 16 s + krow: bits 0..3 = krow, bit 4 = s
      f32x4 acc[2][2];
      Frag3 hp[2];                               // H1 of this lane's units (its own stores of L1)
      // piece j of the epilogue of row block nb: the units 2 j, 2 j + 1 of this lane's 8
      // (unit block j >> 1, result rows 2 (j & 1) and + 1)
      auto finish = [&](int nb, int j) {
        const f32x4* r = acc[nb & 1];
        const Frag3& h = hp[nb & 1];
        float lo = __uint_as_float(h.p[0].r[j] << 16), hi = __uint_as_float(h.p[0].r[j] & 0xffff0000u);
        if (ACT != ACT_RELU && ACT != ACT_LEAKY) {   // (those two need the sign only: part 0 has it)
          lo = (lo + __uint_as_float(h.p[1].r[j] << 16)) + __uint_as_float(h.p[2].r[j] << 16);
          hi = (hi + __uint_as_float(h.p[1].r[j] & 0xffff0000u)) + __uint_as_float(h.p[2].r[j] & 0xffff0000u);
        }
        const int mb = j >> 1, i0 = 2 * (j & 1);
        const float ea = r[mb][i0] * act_d<ACT>(lo);
        const float eb = r[mb][i0 + 1] * act_d<ACT>(hi);
        // (split at once, behind the MFMAs of the next block: after barrier D only stores are left)
        split3(ea, eb, dq[nb][0].r[j], dq[nb][1].r[j], dq[nb][2].r[j]);
        gb1[mb][i0] += ea;
        gb1[mb][i0 + 1] += eb;
      };
      PB ring[2];
      // one address register per (k-step, half) of the dY2 reads and per half of the
      // W2^T third-part reads: row blocks / parts / k-steps are immediate offsets
      int bD[NKT][2], bW[2];
#pragma unroll
      for (int kb = 0; kb < NKT; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) bD[kb][h] = fresh(OFF_D + c * PT + ((64 * kb + 32 * h + 8 * g) ^ swD));
#pragma unroll
      for (int h = 0; h < 2; ++h)
        bW[h] = fresh(OFF_W2P + (16 * h + krow) * PT + ((2 * (ub + 8 * pp)) ^ swWk ^ (h * 48)));
      const int bHo = fresh(OFF_H + c * PT + (ocol ^ swT));
      auto ld = [&](int s, PB& r) {
        const int nb = s / NKT, kb = s % NKT;
        // dY2 in the k order of the transpose read: two 8-byte pieces per part
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int q3 = 0; q3 < 3; ++q3) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(sm + bD[kb][h] + 16 * nb * PT + q3 * T_PART);
            r.b.p[q3].r[2 * h] = v.x;
            r.b.p[q3].r[2 * h + 1] = v.y;
          }
        }
        // third parts of W2^T: rows h2 = 32 kb + 16 h + krow, the 4 units of chunk pp (+ 4: block 1)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x2 v0 = lds_tr64(sm + bW[h] + 32 * kb * PT);
          const u32x2 v1 = lds_tr64(sm + bW[h] + 32 * kb * PT + 8);
          r.w0.r[2 * h] = v0.x; r.w0.r[2 * h + 1] = v0.y;
          r.w1.r[2 * h] = v1.x; r.w1.r[2 * h + 1] = v1.y;
        }
        if (kb == 1) {
#pragma unroll
          for (int q3 = 0; q3 < ((ACT == ACT_RELU || ACT == ACT_LEAKY) ? 1 : 3); ++q3)
            hp[nb & 1].p[q3] = *reinterpret_cast<const Frag*>(sm + bHo + 16 * nb * PT + q3 * T_PART);
        }
      };
      ld(0, ring[0]);
      FENCE();
#pragma unroll
      for (int s = 0; s < NRB * NKT; ++s) {
        const int nb = s / NKT, kb = s % NKT;
        if (s + 1 < NRB * NKT) ld(s + 1, ring[(s + 1) & 1]);
        f32x4* r = acc[nb & 1];
        if (kb == 0) { r[0] = (f32x4){0, 0, 0, 0}; r[1] = (f32x4){0, 0, 0, 0}; }
        const PB& rr = ring[s & 1];
        mma6w(W2t[0][kb], rr.w0, W2t[1][kb], rr.w1, rr.b, r[0], r[0], r[1], r[1]);
        if (nb > 0) finish(nb - 1, kb);
        if (kb == 1) ld_b<PT, T_PART>(sm + OFF_H, 16 * nb + c, ocol, swT, hp[nb & 1]);
        interleave<12, 14, 2, 2>();   // (the reads early in the step: they feed the next one)
        FENCE();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) finish(NRB - 1, k);
    }
    STAMP(7)
    lds_barrier();                             // D: everyone is done with the H1 and dY2 images
    STAMP(8)
    // the next tile's rows are fetched behind dW1 (nothing in this kernel is reloaded
    // from scratch any more, so no wait in front of an MFMA shares their counter)
    load_x();
    advance_x();
    FENCE();
    STAMP(13)
    // ---- dY1 over this wave's slice of the dY2 image (read back by this wave only), then
    // dW1[unit][f] += sum_b dY1[b][unit] X[b][f]: the rows of k-step 1 are written while
    // the MFMAs of k-step 0 run
    {
      LANE_ROLES();
      const int swD = swzD(c), swDk = swzD(krow), swXk = swzX(krow);
      auto st_dq = [&](int nb) {
        char* p = sm + OFF_D + (16 * nb + c) * PT + (ocol ^ swD);
        *reinterpret_cast<Frag*>(p) = dq[nb][0];
        *reinterpret_cast<Frag*>(p + T_PART) = dq[nb][1];
        *reinterpret_cast<Frag*>(p + 2 * T_PART) = dq[nb][2];
      };
      st_dq(0);
      st_dq(1);
      st_dq(2);
      st_dq(3);
      FENCE();
      STAMP(14)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        Frag3 a0, a1, b[NCB];
        ld_t<PT, T_PART>(sm + OFF_D, kb, krow, 32 * (2 * wave) + 8 * pp, swDk, a0);
        ld_t<PT, T_PART>(sm + OFF_D, kb, krow, 32 * (2 * wave + 1) + 8 * pp, swDk, a1);
#pragma unroll
        for (int n = 0; n < NCB; ++n) ld_t<PX, X_PART>(ximg, kb, krow, 32 * n + 8 * pp, swXk, b[n]);
        FENCE();
#pragma unroll
        for (int n = 0; n < NCB; ++n) mma6x2(a0, a1, b[n], gW1[0][n], gW1[1][n]);
        if (kb == 1) split_x();                    // (the rows were requested a phase ago)
        interleave<12 * NCB, 0, 3>();
        FENCE();
      }
    }
    STAMP(9)
    lds_barrier();                             // E: everyone is done with the X image
    STAMP(10)
    store_x();
    STAMP(11)
    lds_barrier();                             // F: the next tile's X image complete
    STAMP(12)
  }

  // ---- this workgroup's partial slab: [W1 | b1 | W2 | b2 | w3 | b3 | loss | pad]
#pragma unroll
  for (int mr = 0; mr < 2; ++mr)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = 16 * (2 * wave + mr) + 4 * g + i;
#pragma unroll
      for (int n = 0; n < NB; ++n) oW2[u * HID + 16 * n + c] = gW2[mr][n][i];
#pragma unroll
      for (int n = 0; n < NCB; ++n) {
        const int f = 16 * n + c;
        if (f < din) oW1[u * din + f] = gW1[mr][n][i];
      }
    }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s1 = gb1[mb][i], s2 = gb2[mb][i], s3 = gw3[mb][i];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 16);
        s2 += __shfl_xor(s2, off, 16);
        s3 += __shfl_xor(s3, off, 16);
      }
      if (c == 0) {
        const int u = ub + 8 * g + 4 * mb + i;
        ob1[u] = s1;
        ob2[u] = s2;
        ow3[u] = s3;
      }
    }
#ifdef MLPB_STAMP
  __syncthreads();
  if (tid == 0 && blockIdx.x == 0)
    for (int k = 0; k < 16; ++k) out[k] = (float)stt[k];
  if (tid == 0 && blockIdx.x == 0) return;
#endif
  if (wave == 0) {
    const float s3 = wave_sum(g == 0 ? gb3 : 0.f), sl = wave_sum(g == 0 ? loss_sum : 0.f);
    if (lane == 0) {
      ob3[0] = s3;
      ob3[1] = sl;                             // sum of squared errors of this workgroup
      ob3[2] = 0.f;
    }
  }
}

template <int ACT>
void launchb(const MlpArgs& a, int grid, hipStream_t st) {
  const bool xv = ((uintptr_t)a.x % 16 == 0) && a.env_stride % 4 == 0 && a.row_stride % 4 == 0 &&
                  a.row_stride >= 4;
#define MLPB_LAUNCH(NK, V)                                                                                  \
  do {                                                                                                      \
    tce_lds_limit(reinterpret_cast<const void*>(mlp_critic_bwdb_kernel<ACT, NK, V>), (size_t)(LDSB_BYTES)); \
    hipLaunchKernelGGL((mlp_critic_bwdb_kernel<ACT, NK, V>), dim3(grid), dim3(MLP_BT), LDSB_BYTES, st, a);  \
  } while (0)
  if (a.din <= 32) {
    if (xv) MLPB_LAUNCH(1, true); else MLPB_LAUNCH(1, false);
  } else {
    if (xv) MLPB_LAUNCH(2, true); else MLPB_LAUNCH(2, false);
  }
#undef MLPB_LAUNCH
}

}  // namespace

extern "C" {

// tce_mlp_critic_f32's backward launch (same buffers, same slab / gradient /
// stats / fused-Adam contract, partials != NULL required) on the bf16 matrix
// cores with three-part operands: see the file header for the arithmetic.
int tce_mlp_critic_bf16x3(const float* x, int64_t env_stride, int64_t row_stride, int T,
                          int64_t R, int din, const float* w1, const float* b1,
                          const float* w2, const float* b2, const float* w3, const float* b3,
                          int act, const float* returns, const float* old_values, float clip,
                          float* values, float* partials, float* grad, float* stats,
                          int max_workgroups, float* adam_param, float* adam_m, float* adam_v,
                          float* adam_state, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float adam_step, float grad_scale, void* xchg,
                          void* stream) {
  TCE_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && R > 0 && T > 0,
                "mlp_critic_bf16x3: null buffer / bad sizes");
  TCE_CHECK_ARG(din >= 1 && din <= MAX_DIN, "mlp_critic_bf16x3: 1 <= D_in <= 40");
  TCE_CHECK_ARG(act >= 0 && act <= 3, "mlp_critic_bf16x3: unknown activation");
  TCE_CHECK_ARG(partials && returns && grad && stats, "mlp_critic_bf16x3: backward buffers missing");
  TCE_CHECK_ARG(!(clip > 0.f && !old_values), "mlp_critic_bf16x3: old values missing");
  TCE_CHECK_ARG(!adam_param || (adam_m && adam_v && adam_state && adam_step >= 1.f),
                "mlp_critic_bf16x3: fused Adam needs its state buffers");
  TCE_CHECK_ARG(!xchg || adam_param, "mlp_critic_bf16x3: an exchange needs the fused Adam step");
  const MlpArgs a{x, env_stride, row_stride, T, R, din, w1, b1, w2, b2, w3, b3,
                  returns, old_values, clip, values, partials, nullptr, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = ceil_div(R, ROWS_PER_TILE);
  int cap = 256;
  if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
  const int grid = (int)tmin<int64_t>(cap, ntiles);
  switch (act) {
    case 0: launchb<ACT_TANH>(a, grid, st); break;
    case 1: launchb<ACT_RELU>(a, grid, st); break;
    case 2: launchb<ACT_LEAKY>(a, grid, st); break;
    default: launchb<ACT_SOFTPLUS>(a, grid, st); break;
  }
  TCE_LAUNCH_CHECK();
  const int P = mlp_num_params(din);
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
