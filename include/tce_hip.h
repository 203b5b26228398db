/* libtce_hip.so -- C ABI of the MI355X (gfx950) TCE/BBRL hot path.
 *
 * The reference (BruceGeLi/TCE_RL) has no FFI: its boundary is a set of Python
 * classes.  This header is the C-ABI layer the build puts underneath those
 * classes; every entry point names the reference code it replaces (paths are
 * relative to /root/reference/).  Conventions:
 *   - all pointers are DEVICE pointers (HIP), row-major, contiguous unless a
 *     stride argument says otherwise; bool tensors are 1 byte per element;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue;
 *   - return 0 on success, non-zero on error (message: tce_last_error());
 *     nothing aborts the process;
 *   - *_f32 / *_f64 are the two dtypes the reference accepts
 *     (mprl/util/util_data_structure.py:59-78);
 *   - the library keeps no pointers after a call returns.
 */
#ifndef TCE_HIP_H
#define TCE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------- */
const char* tce_last_error(void);
int tce_version(void);
int tce_device_count(void);

/* ---- GAE + segment advantage ------------------------------------------
 * TemporalCorrelatedAgent.get_advantage_return
 *   (mprl/rl/agent/temporal_correlated_agent.py:118-181) and, when P > 0, the
 * un-normalised `value_subtraction` segment advantage (:236-279) fused in.
 * rewards/dones/tl_dones/adv/ret [N,T]; values [N,T+1]; pairs int64 [P,2];
 * seg_out [N,P]; partials double [tce_gae_num_partials(N), 3] =
 * per-workgroup (count, mean, M2) of seg_out for the global normalisation.
 * Results are bit-identical to the reference's op order (no FMA contraction).
 */
int64_t tce_gae_num_partials(int64_t N);
int tce_gae_f32(const float* rewards, const float* values, const uint8_t* dones,
                const uint8_t* tl_dones, float* adv, float* ret,
                const int64_t* pairs, int P, float* seg_out, double* partials,
                int64_t N, int T, float gamma, float lam, int use_gae,
                void* stream);
int tce_gae_f64(const double* rewards, const double* values, const uint8_t* dones,
                const uint8_t* tl_dones, double* adv, double* ret,
                const int64_t* pairs, int P, double* seg_out, double* partials,
                int64_t N, int T, double gamma, double lam, int use_gae,
                void* stream);

/* ---- global mean / unbiased std (advantage normalisation) ---------------
 * `(x - x.mean()) / (x.std() + 1e-8)` of temporal_correlated_agent.py:213-215,
 * 230-233,281-284 and black_box_agent.py:93-97, split so that multi-GPU runs
 * can merge the (count, mean, M2) triples of all ranks before normalising.
 * partials: double [tce_moments_num_partials(), 3]; stats: double[3] =
 * {count, mean, M2}.  tce_normalize: y = clamp((x-mean)/(std+eps), +-clip);
 * stats == NULL -> clamp only; clip <= 0 -> no clamp; single_std_one: std := 1
 * when count == 1 (black_box_agent.py:95).
 */
int64_t tce_moments_num_partials(void);
int tce_moments_partial_f32(const float* x, int64_t n, double* partials, void* stream);
int tce_moments_partial_f64(const double* x, int64_t n, double* partials, void* stream);
int tce_moments_finalize(const double* partials, int nparts, double* stats, void* stream);
int tce_normalize_f32(const float* x, float* y, int64_t n, const double* stats,
                      float eps, float clip, int single_std_one, void* stream);
int tce_normalize_f64(const double* x, double* y, int64_t n, const double* stats,
                      double eps, double clip, int single_std_one, void* stream);

/* ---- other segment-advantage modes -------------------------------------
 * `accumulate` (temporal_correlated_agent.py:211-228): out[n,p] =
 * sum_{t=a..b inclusive} f(adv[n,t]), f = optional normalise (stats) + clamp.
 * `accumulated_rewards` (:288-319): (sum_{[a,b)} gamma^t r - column mean)/gamma^a.
 */
int tce_segment_accumulate_f32(const float* adv, const int64_t* pairs, int P,
                               float* out, int64_t N, int T, const double* stats,
                               float eps, float clip, void* stream);
int tce_segment_accumulate_f64(const double* adv, const int64_t* pairs, int P,
                               double* out, int64_t N, int T, const double* stats,
                               double eps, double clip, void* stream);
int tce_segment_accrew_f32(const float* rewards, const int64_t* pairs, int P,
                           float* out, int64_t N, int T, float gamma, void* stream);
int tce_segment_accrew_f64(const double* rewards, const int64_t* pairs, int P,
                           double* out, int64_t N, int T, double gamma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TCE_HIP_H */
