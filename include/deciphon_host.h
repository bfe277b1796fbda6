/* deciphon_host.h -- C ABI of the host-side pieces that sit either side of the
 * GPU path: the pressed-database reader and the scalar bookkeeping of
 * process_window.  None of these touch the GPU; they exist so that a caller of
 * deciphon_hip.h (and the tests) can prepare inputs and interpret outputs
 * exactly as the reference does.  All return 0 or a DCP_E* code
 * (c-core/deciphon.h:34-116) unless stated otherwise.
 */
#ifndef DECIPHON_HOST_H
#define DECIPHON_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- .dcp reader: database_reader_open (c-core/database_reader.c:26-80),
 * protein_reader offsets (c-core/protein_reader.c:112-128) and protein_unpack
 * (c-core/protein.c:283-351) ---- */
struct dcp_db;
int dcp_db_open(char const *path, struct dcp_db **out);
void dcp_db_close(struct dcp_db *);
int dcp_db_num_proteins(struct dcp_db const *);
float dcp_db_epsilon(struct dcp_db const *);
int dcp_db_entry_dist(struct dcp_db const *);
int dcp_db_has_ga(struct dcp_db const *);
int64_t dcp_db_protein_offset(struct dcp_db const *, int i); /* byte offset in the file */
int dcp_db_protein_core_size(struct dcp_db const *, int i, int *core_size);
/* node_trans[(K+1)*7], node_emission[(K+1)*1364], BMk[K], null[1364], bg[1364],
 * accession[32], consensus[K+1]; any pointer may be NULL to skip it */
int dcp_db_read_protein(struct dcp_db const *, int i, float *node_trans, float *node_emission, float *BMk,
                        float *null_lprob, float *bg_lprob, char *accession, char *consensus);

/* The distributions quasi-codon decoding uses (decoder_setup, c-core/decoder.c:21-36): for entry 0 = null model,
 * 1 = background, 2 + n = node n (n = 0..K): nucltp[(K+3)*4] nucleotide log-probabilities and codonm[(K+3)*125]
 * codon marginals (5 x 5 x 5, index 4 = any nucleotide).  Any pointer may be NULL. */
int dcp_db_read_nuclt_dist(struct dcp_db const *, int i, float *nucltp, float *codonm, int *gencode);
/* decoder_decode (c-core/decoder.c:38-58; arithmetic of third-party imm's imm_frame_cond_decode restated, see
 * csrc/host_logic.h): the codon (nucleotide indices) that most likely produced the n = 1..5 nucleotides nt under
 * error rate epsilon.  Returns 0, or DCP_EDECODON.  dcp_gencode_amino_of: imm_gencode_decode, 0 = unknown table. */
int dcp_decode_quasi_codon(float epsilon, float const *nucltp4, float const *codonm125, uint8_t const *nt, int n,
                           uint8_t codon[3]);
char dcp_gencode_amino_of(int gencode_id, uint8_t const codon[3]);

/* partition_size (c-core/partition_size.c:13-16): proteins of partition idx out of nparts */
long dcp_partition_size(long nelems, long nparts, long idx);
/* core sizes of all proteins (read from the protein heads; nothing else of a protein is touched) */
int dcp_db_core_sizes(struct dcp_db const *, int32_t *core_sizes);
/* first[nparts + 1]: partition p holds proteins first[p] .. first[p+1]-1.  balanced = 0: the rule above, as
 * c-core/protein_reader.c:112-128 applies it.  balanced = 1: contiguous and in order all the same, boundaries
 * where the running sum of core sizes is closest to p/nparts of the total (DP cells go with K). */
int dcp_db_partition_bounds(struct dcp_db const *, int nparts, int balanced, int32_t *first);
/* the same rule on a plain array of n core sizes (core_sizes may be NULL when balanced = 0) */
int dcp_partition_bounds_of(int n, int32_t const *core_sizes, int nparts, int balanced, int32_t *first);

/* ---- window iteration: window_setup / window_next / window_set_last_hit_position
 * (c-core/window.c:7-50) ---- */
struct dcp_window
{
  int32_t core_size, seq_size;
  int32_t start, stop, idx, last_hit_pos;
};
void dcp_window_setup(struct dcp_window *, int seq_size, int core_size);
int dcp_window_next(struct dcp_window *); /* 1 = a window was produced, 0 = end */

/* ---- trellis_unzip (c-core/trellis.c:147-167): steps of the traceback in path
 * order.  *nsteps receives the count; fails with DCP_ENOMEM when cap is too small. */
int dcp_trellis_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, int cap, int32_t *state_ids,
                      int32_t *seqsizes, int *nsteps);

/* ---- the hit of a window (c-core/thread.c:130-166): from the first B to the last
 * E of the path.  Returns 1 and fills hit[5] = {hit_start, hit_stop, begin_step,
 * end_step, last_hit_pos}, or 0 when the path has no B. */
int dcp_path_hit(int nsteps, int32_t const *state_ids, int32_t const *seqsizes, int32_t hit[5]);

/* state_name (c-core/state.c:46-90); name must hold 8 bytes */
void dcp_state_name_of(int state_id, char *name);
int dcp_state_is_mute_id(int state_id);

/* lrt (c-core/lrt.h:6-9) */
float dcp_lrt_of(float null_loglik, float alt_loglik);

#ifdef __cplusplus
}
#endif

#endif
