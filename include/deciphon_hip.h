/* deciphon_hip.h -- C ABI of the MI355X-native Deciphon Viterbi scan path.
 *
 * This is the batched operator interface a libdeciphon build binds in place of
 * its per-thread DP engine: where c-core/thread.c:98-128 (process_window) calls
 * viterbi_null / viterbi_cost / viterbi_path on one (profile, window) at a time
 * through c-core/viterbi.h:38-52, a caller hands this library the same
 * parameters for many windows at once and gets the same numbers back.
 * Plain pointers and sizes only; every function returns 0 or a DCP_E* code of
 * c-core/deciphon.h:34-116 (the enum is part of the Python-visible ABI and is
 * reused, not extended).  The per-problem drop-in for viterbi.h itself is in
 * dcp_viterbi.h; the reference's outer dcp_scan_ / dcp_batch_ API is in
 * deciphon.h of this directory.
 *
 * There is no CPU fallback: every entry point that computes fails with
 * DCP_EFUNCUSE when no gfx950 device is usable.
 */
#ifndef DECIPHON_HIP_H
#define DECIPHON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCP_HIP_TABLE_SIZE 1364 /* viterbi_table_size(), c-core/viterbi.c:13,739 */
#define DCP_HIP_NUM_TRANS 8     /* enum core_trans_id, c-core/viterbi.h:22-32 */
#define DCP_HIP_NUM_XTRANS 13   /* enum extr_trans_id, c-core/viterbi.h:4-19 */

struct dcp_hip;

/* Number of usable HIP devices (0 when there is none; never fails). */
int dcp_hip_device_count(void);

/* One engine per GPU and per host thread (replaces the per-thread `struct work`
 * array of c-core/scan.c:28-33 / c-core/work.c:24-51).  NULL when the device
 * cannot be initialised. */
struct dcp_hip *dcp_hip_new(int device);
void dcp_hip_del(struct dcp_hip *);
char const *dcp_hip_strerror(struct dcp_hip const *); /* detail of the last failure */

/* ---- profiles (replaces work_setup -> protein_setup_viterbi, c-core/work.c:24-45) ----
 * dcp_hip_add_profile takes DP costs exactly as viterbi_set_core_trans /
 * _set_match / _set_null / _set_background receive them (c-core/viterbi.c:407-444):
 *   trans[8][K] in enum core_trans_id order, match[1364][K], null[1364], bg[1364].
 * dcp_hip_add_protein takes what protein_unpack yields (c-core/protein.c:283-351),
 * natural-log probabilities: node_trans[K+1][7] (MM,MI,MD,IM,II,DM,DD),
 * node_emission[K+1][1364], BMk[K], null/bg emission[1364], and applies the
 * mapping of protein_setup_viterbi (c-core/protein.c:353-394).
 * Both return the profile's index through *index.  Core sizes up to 16383 (state ids keep 14
 * bits for k + 1, c-core/state.h:27-39); DCP_ELARGECORESIZE beyond.  The delete costs MD, DD must
 * be non-negative (-log-probabilities are): the kernels take E = min M and bound delete runs
 * with that; DCP_EFUNCUSE / DCP_EFDATA otherwise. */
int dcp_hip_add_profile(struct dcp_hip *, int K, float const *trans, float const *match,
                        float const *null_cost, float const *bg_cost, int *index);
int dcp_hip_add_protein(struct dcp_hip *, int K, float const *node_trans, float const *node_emission,
                        float const *BMk, float const *null_lprob, float const *bg_lprob, int *index);
/* Reads proteins [first, first+count) of a pressed database (count < 0: to the
 * end); replaces database_reader_open + protein_reader + protein_iter_next
 * (c-core/database_reader.c:26-80, c-core/protein_reader.c:29-128). */
int dcp_hip_load_dcp(struct dcp_hip *, char const *path, int first, int count);
/* how many staging chunks the last dcp_hip_load_dcp went through (256 MiB each; DECIPHON_HIP_STAGE_MB
 * shrinks them, never below one profile: a hook for tests of the double-buffered path) */
int dcp_hip_load_chunks(struct dcp_hip const *);
/* Bytes of HBM the resident profile tables occupy (rows + transitions, padded). */
int64_t dcp_hip_pool_bytes(struct dcp_hip const *);
int dcp_hip_num_profiles(struct dcp_hip const *);
int dcp_hip_profile_core_size(struct dcp_hip const *, int index);
char const *dcp_hip_profile_accession(struct dcp_hip const *, int index);
/* Uploads everything added so far to HBM (idempotent). */
int dcp_hip_commit_profiles(struct dcp_hip *);
void dcp_hip_clear_profiles(struct dcp_hip *);

/* ---- sequences (replaces batch_encode -> sequence_encode, c-core/batch.c:60-70) ----
 * nt holds nucleotide indices A,C,G,T/U = 0..3 of nseq sequences back to back;
 * offsets[nseq+1].  dcp_hip_encode does what dcp_batch_add does to one string
 * (uppercase + disambiguate, c-core/sequence.c:15-45, c-core/disambiguate.c:37-86)
 * and writes n indices. */
int dcp_hip_encode(char const *data, int64_t n, uint8_t *out);
int dcp_hip_set_sequences(struct dcp_hip *, int nseq, uint8_t const *nt, int64_t const *offsets);

/* ---- special transitions (replaces work_reset -> xtrans_setup, c-core/work.c:47-51) ---- */
int dcp_hip_set_mode(struct dcp_hip *, int multi_hits, int hmmer3_compat);
/* Overrides the special-transition costs for amino lengths 0..rows-1 with the
 * caller's own (what a caller of viterbi_set_extr_trans, c-core/viterbi.c:383-405,
 * may pass): xt[rows][13] in enum extr_trans_id order.  Lengths beyond the table
 * keep following dcp_hip_set_mode. */
int dcp_hip_set_xtrans_table(struct dcp_hip *, int rows, float const *xt);
/* The 13 costs xtrans_setup_viterbi would set for a window whose amino length
 * is seq_size (c-core/xtrans.c:21-68), in enum extr_trans_id order. */
void dcp_hip_xtrans(int seq_size, int multi_hits, int hmmer3_compat, float xt[DCP_HIP_NUM_XTRANS]);

/* ---- the DP ---------------------------------------------------------------------------
 * A window is the half-open range [start, stop) of sequence `seq` scored against
 * `profile`, i.e. one process_window call (c-core/thread.c:98-128). */
struct dcp_hip_window
{
  int32_t profile;
  int32_t seq;
  int32_t start;
  int32_t stop;
};

/* viterbi_null + viterbi_cost for n windows (c-core/thread.c:114-117):
 * null_cost[i] / alt_cost[i] are the raw return values (the caller negates them
 * into log-likelihoods and forms lrt = -2*(null - alt), c-core/lrt.h:6-9). */
int dcp_hip_cost(struct dcp_hip *, int n, struct dcp_hip_window const *, float *null_cost, float *alt_cost);

/* The same followed by process_window's filter (c-core/thread.c:118-121) on the device: only the windows
 * whose lrt = -2 * (null - alt) (c-core/lrt.h:6-9) is finite and >= 0 come back -- *nhits of them, their indices
 * into the window array in increasing order and their lrt; hit_window and hit_lrt must hold n entries. */
int dcp_hip_cost_hits(struct dcp_hip *, int n, struct dcp_hip_window const *, int *nhits, int32_t *hit_window,
                      float *hit_lrt);
/* The same in two halves: _begin stages the windows and enqueues the kernels and returns while the GPU works;
 * _end waits for the OLDEST batch begun and delivers what dcp_hip_cost_hits would have.  Two batches may be
 * outstanding per engine (the second queues behind the first, so the GPU does not drain between them); a third
 * _begin is a DCP_EFUNCUSE.  While batches are outstanding the engine accepts only _begin, _end, the read-only
 * queries and dcp_hip_path (which has buffers and streams of its own); everything else is a DCP_EFUNCUSE. */
int dcp_hip_cost_hits_begin(struct dcp_hip *, int n, struct dcp_hip_window const *);
int dcp_hip_cost_hits_end(struct dcp_hip *, int *nhits, int32_t *hit_window, float *hit_lrt);

/* viterbi_path + trellis_unzip for n windows (c-core/thread.c:124-126).  Results
 * stay valid until the next dcp_hip_path / dcp_hip_del.
 * The steps come from a fast pass (cost pass with the DP values kept in HBM + a traceback
 * that picks, at every visited state, the first candidate equal to the stored minimum --
 * the reference's strict-< rule); windows in which that meets an exact fp32 tie only the
 * reference's pass order resolves are redone with the literal pass-by-pass kernel.  The
 * trellis itself is produced (by the literal kernel, for the whole batch) only when
 * dcp_hip_path_trellis is called.  DECIPHON_HIP_PATH=literal forces the literal pass.
 * A window with no finite path at all (viterbi_cost = +inf, which the reference never sends
 * here: c-core/thread.c:118-121) yields 0 steps and score +inf. */
int dcp_hip_path(struct dcp_hip *, int n, struct dcp_hip_window const *);
/* Sets aside `bytes` of HBM for the DP tables of dcp_hip_path now (never shrinks).  VRAM is
 * cleared when allocated, in the background: called early (before the profiles are loaded) the
 * clearing overlaps the load and the cost pass.  Clamped to a quarter of the free device memory.
 * Optional: dcp_hip_path allocates on demand. */
int dcp_hip_path_reserve(struct dcp_hip *, int64_t bytes);
/* how many windows of the last dcp_hip_path needed the literal pass */
int dcp_hip_path_redone(struct dcp_hip const *);
/* number of steps of window i's path (S ... T) */
int dcp_hip_path_nsteps(struct dcp_hip const *, int i);
/* state ids (c-core/state.h:9-25, state.c:92-96) and emission lengths of every step */
int dcp_hip_path_steps(struct dcp_hip const *, int i, int32_t *state_ids, int32_t *seqsizes);
/* The same without a copy: *steps points at window i's *nsteps steps as the engine holds them, one word each --
 * state id in the low 16 bits, emission length in the high -- valid until the next dcp_hip_path / dcp_hip_del
 * (what struct imm_step carries apart from the score, c-core/trellis.c:147-167). */
int dcp_hip_path_steps_packed(struct dcp_hip const *, int i, uint32_t const **steps, int32_t *nsteps);
/* the packed back-pointers themselves: xnodes[L+1], nodes[(L+1)*K] (c-core/trellis.h:12-21) */
int dcp_hip_path_trellis(struct dcp_hip const *, int i, uint32_t const **xnodes, uint16_t const **nodes);
/* score of the path pass's own DP (equals alt_cost of dcp_hip_cost) */
float dcp_hip_path_score(struct dcp_hip const *, int i);

/* ---- measurement ------------------------------------------------------------------------
 * Stages n windows in HBM once, then times `reps` launches of the cost pass over
 * them with HIP events on the engine's own stream (inputs resident, nothing
 * copied inside the timed region).  *ms receives the mean milliseconds per
 * launch, *cells the DP cells (sum of K*L) one launch computes. */
int dcp_hip_cost_bench(struct dcp_hip *, int n, struct dcp_hip_window const *, int warmup, int reps,
                       float *ms, double *cells, float *null_cost, float *alt_cost);
/* The same in two steps: dcp_hip_stage copies the window list to HBM (untimed);
 * dcp_hip_run_staged launches the cost pass `reps` times over it and returns when the
 * last launch has finished; *ms = HIP-event time of all `reps` launches together,
 * *cells = DP cells of ONE launch.  dcp_hip_fetch_staged copies the scores back. */
int dcp_hip_stage(struct dcp_hip *, int n, struct dcp_hip_window const *);
int dcp_hip_run_staged(struct dcp_hip *, int reps, float *ms, double *cells);
int dcp_hip_fetch_staged(struct dcp_hip *, float *null_cost, float *alt_cost);

#ifdef __cplusplus
}
#endif

#endif
