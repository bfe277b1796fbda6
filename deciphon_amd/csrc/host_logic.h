// host_logic.h -- the scalar host-side pieces around the DP (parameters in,
// paths out).  Each function names the reference code it stands in for.
#pragma once
#include "dcp_types.h"
#include <stdint.h>
#include <string>
#include <vector>

// c-core/xtrans.c:21-68: special transitions of a window whose amino length is
// seq_size, negated into DP costs, in enum extr_trans_id order.
void dcp_xtrans(int seq_size, bool multi_hits, bool hmmer3_compat, float xt[DCP_NUM_XTRANS]);

// c-core/protein.c:353-394 (protein_setup_viterbi): node-major log-probs ->
// DP costs in the device layout, padded to Kp columns with +inf: trans[8][Kp] and
// rows[1364][DCP_ROW_HDR + Kp] = { null, bg, 0, 0, match[0..Kp) }.
void dcp_setup_profile(int K, int Kp, float const *node_trans, float const *node_emission, float const *BMk,
                       float const *null_lprob, float const *bg_lprob, float *trans, float *rows);

// c-core/sequence.c:15-45 (uppercase + disambiguate, c-core/disambiguate.c:37-86)
// followed by the A,C,G,T/U -> 0..3 indexing imm_eseq applies.  0 or DCP_E*.
int dcp_encode_sequence(char const *data, int64_t n, uint8_t *out);

// c-core/trellis.c:147-167 (trellis_unzip) with previous_state/emission_size
// (:51-113).  Appends (state_id, seqsize) steps in path order.  0 or DCP_E*.
int dcp_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, std::vector<int32_t> &state_ids,
              std::vector<int32_t> &seqsizes);

// c-core/state.c:46-90
void dcp_state_name(int state_id, char name[8]);
bool dcp_state_is_mute(int state_id); // c-core/state.c:36-43

// c-core/thread.c:130-166: the one hit of a window spans from the first B to
// the last E of its path.  Returns false when the path holds no B.
struct DcpHit
{
  int hit_start, hit_stop; // window-relative [start, stop)
  int begin_step, end_step; // path steps [B, one past the last E)
  int last_hit_pos;         // what window_set_last_hit_position receives
};
bool dcp_find_hit(std::vector<int32_t> const &state_ids, std::vector<int32_t> const &seqsizes, DcpHit &hit);
// the same on packed steps (state id | emission length << 16, dcp_hip_path_steps_packed)
bool dcp_find_hit_packed(uint32_t const *steps, int n, DcpHit &hit);

// c-core/window.c:7-37
struct DcpWindow
{
  int core_size, seq_size;
  int start = -1, stop = 0, idx = -1, last_hit_pos = -1;
  DcpWindow(int seq_size_, int core_size_) : core_size(core_size_), seq_size(seq_size_) {}
  bool next();
};

// c-core/lrt.h:6-9
inline float dcp_lrt(float null_loglik, float alt_loglik) { return -2 * (null_loglik - alt_loglik); }

// Contiguous partitions of n profiles for nparts workers: first[p] .. first[p + 1] is partition p.
// balanced = false: the reference's rule, ceil((n - p) / nparts) profiles each (c-core/partition_size.c:13-16
// as c-core/protein_reader.c:112-128 applies it).  balanced = true: still contiguous and in order (the row
// order of products.tsv depends on that, c-core/product.c:63-81), but the boundaries sit where the running
// sum of core sizes comes closest to p/nparts of the total -- DP cells are proportional to K, not to the
// number of profiles.
void dcp_partition_bounds(int n, int32_t const *core_sizes, int nparts, bool balanced, int32_t *first);

// ---- quasi-codon decoding: decoder_decode (c-core/decoder.c:38-58) + imm_gencode_decode ----
// The arithmetic is third-party imm's imm_frame_cond_decode (absent here; unpinned HEAD in the reference's CI).
// Restated from the published quasi-codon model, whose marginal form the pressed tables themselves confirm
// (tests/test_decoder.py: sum over codons of P(codon) P(z | codon) reproduces the emission table of every
// node of the reference's minifam.dcp to fp32 rounding):  z = 1..5 nucleotides emitted for codon x under
// per-base error rate e --
//   |z| = 1: e^2 (1-e)^2 / 3       * #{ i : x_i = z_1 }
//   |z| = 2: 2 e (1-e)^3 / 3       * #{ deletions of one base of x that leave z }
//          + e^3 (1-e) / 3         * ( p(z_1) #{ i : x_i = z_2 } + p(z_2) #{ i : x_i = z_1 } )
//   |z| = 3: (1-e)^4 [x = z] + 4 e^2 (1-e)^2 / 9 * sum_j p(z_j) #{ deletions of one base of x that leave z \ j }
//          + e^4 p(z_1) p(z_2) p(z_3)
//   |z| = 4: e (1-e)^3 / 2 * sum_j p(z_j) [x = z \ j] + e^3 (1-e) / 9 * sum_{i<j} p(z_i) p(z_j) #{ deletions of one base of x that leave z \ i,j }
//   |z| = 5: e^2 (1-e)^2 / 10 * sum_{i<j} p(z_i) p(z_j) [x = z \ i,j]
// with p = the state's nucleotide distribution; the decoded codon maximises P(x) P(z | x) over the 64 codons,
// P(x) from the state's codon marginals (first maximum in A,C,G,T order).  Parity with imm is pinned only by
// the reference's committed products.tsv (three hits, every step an exact codon): unpinned beyond that.
// z: nucleotide indices 0..3, n = 1..5.  Returns false when no codon has positive probability.
bool dcp_decode_codon(float epsilon, float const nucltp[4], float const codonm[125], uint8_t const *z, int n,
                      uint8_t codon[3]);
// the same from probabilities made once: base[4] = exp(nucltp), prior[64] = exp(codonm) of the 64 codons
bool dcp_decode_codon_prob(double epsilon, double const base[4], double const prior[64], uint8_t const *z, int n,
                           uint8_t codon[3]);
// imm_gencode_decode: amino acid of a codon under NCBI translation table `gencode_id`; 0 when the table is unknown
char dcp_gencode_amino(int gencode_id, uint8_t const codon[3]);
