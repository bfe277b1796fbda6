/* dcp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of Deciphon's quasi-codon Viterbi scan path, used as
 * the parity oracle for the HIP kernels.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (deciphon_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  Checked bit-for-bit (scores as fp32 bit patterns,
 * every trellis word, every unzipped path) against the reference's own
 * c-core/viterbi.c compiled unmodified from /root/reference (oracle/_ref, see
 * oracle/Makefile) and against the reference's committed products.tsv golden
 * (control/tests/files/snap.dcs) -- tests/test_oracle_*.py.
 *
 * Profiles are handed over in DP-parameter space ("costs" = negated
 * log-probabilities, +inf = impossible), the layout protein_setup_viterbi
 * (c-core/protein.c:353-394) feeds into viterbi_set_*():
 *   trans[8][K]   order BM,MM,MI,MD,IM,II,DM,DD  (enum core_trans_id, c-core/viterbi.h:22-32)
 *   match[1364][K] code-major; null[1364]; bg[1364]
 *   xt[13]        order RR,SN,NN,SB,NB,EB,JB,EJ,JJ,EC,CC,ET,CT (enum extr_trans_id, c-core/viterbi.h:4-19)
 * Sequences are uint8 nucleotide indices A,C,G,T(U) = 0..3.
 */
#ifndef DCP_ORACLE_H
#define DCP_ORACLE_H

#include <stdint.h>

#define ORC_TABLE_SIZE 1364
#define ORC_NUM_TRANS 8
#define ORC_NUM_XTRANS 13

enum { ORC_BM, ORC_MM, ORC_MI, ORC_MD, ORC_IM, ORC_II, ORC_DM, ORC_DD };
enum { ORC_RR, ORC_SN, ORC_NN, ORC_SB, ORC_NB, ORC_EB, ORC_JB, ORC_EJ, ORC_JJ,
       ORC_EC, ORC_CC, ORC_ET, ORC_CT };

/* code of the len-mer starting at pos: imm_eseq_get() as used by
 * c-core/thread.c:92-96 (third-party imm, restated). */
int orc_code(uint8_t const *seq, int pos, int len);

/* c-core/xtrans.c:21-68: length-dependent special transitions, already negated
 * into DP costs in the order of enum extr_trans_id. */
void orc_xtrans(int seq_size, int multi_hits, int hmmer3_compat, float xt[ORC_NUM_XTRANS]);

/* c-core/protein.c:353-394: node-major log-prob profile -> DP cost arrays. */
void orc_setup_profile(int K, float const *node_trans /*[(K+1)*7] MM,MI,MD,IM,II,DM,DD*/,
                       float const *node_emission /*[(K+1)*1364]*/, float const *BMk /*[K]*/,
                       float const *null_lprob /*[1364]*/, float const *bg_lprob /*[1364]*/,
                       float *trans /*[8*K]*/, float *match /*[1364*K]*/, float *null_cost,
                       float *bg_cost);

/* c-core/viterbi.c:696-719 */
float orc_null(float const *null_cost, float RR, uint8_t const *seq, int L);

/* c-core/viterbi.c:451-600 (+602-694 when xnodes/nodes are non-NULL).
 * xnodes: uint32[L+1], nodes: uint16[(L+1)*K].  ref_lanes is the SIMD width
 * whose cross-lane E tie rule is reproduced (8 = the reference's -mavx2 build). */
float orc_cost(int K, float const *trans, float const *match, float const *null_cost,
               float const *bg_cost, float const xt[ORC_NUM_XTRANS], uint8_t const *seq,
               int L, int ref_lanes, uint32_t *xnodes, uint16_t *nodes);

/* c-core/trellis.c:147-167.  Returns number of steps, or -1 if cap is too small. */
int orc_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, int *state_ids,
              int *seqsizes, int cap);

/* c-core/lrt.h:6-9 */
float orc_lrt(float null_loglik, float alt_loglik);

/* c-core/window.c:7-37; returns 1 and updates the window, or 0 at the end. */
struct orc_window { int core_size, seq_size, start, stop, idx, last_hit_pos; };
struct orc_window orc_window_setup(int seq_size, int core_size);
int orc_window_next(struct orc_window *w);

/* c-core/thread.c:130-166: B..E hit spans of an unzipped path.  Writes up to
 * cap (hit_start, hit_stop, first_step, end_step) quadruples in the order the
 * reference would emit product lines; returns the count.  *last_hit_pos receives
 * the value window_set_last_hit_position() is called with (or is left alone). */
int orc_hits(int const *state_ids, int const *seqsizes, int nsteps, int *hits, int cap,
             int *last_hit_pos);

/* c-core/disambiguate.c:37-86 + uppercase.c + the ACGT(U)->0..3 encoding.
 * Returns 0, or a DCP_E* code. out must hold n bytes. */
int orc_encode(char const *data, int n, uint8_t *out);

/* c-core/state.c:46-90 */
void orc_state_name(int state_id, char *name /*>=8 bytes*/);

/* c-core/partition_size.c:13-16 */
long orc_partition_size(long nelems, long nparts, long idx);

#endif
