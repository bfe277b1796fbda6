// host_capi.cpp -- C ABI of include/deciphon_host.h over dcp_db / host_logic.
#include "../../include/deciphon_host.h"
#include "dcp_db.h"
#include "dcp_errors.h"
#include "host_logic.h"

#include <string.h>
#include <string>
#include <vector>

struct dcp_db
{
  DcpDbReader reader;
};

extern "C" {

int dcp_db_open(char const *path, struct dcp_db **out)
{
  if (!path || !out) return DCP_EFUNCUSE;
  dcp_db *x = new dcp_db;
  int rc = x->reader.open(path);
  if (rc)
  {
    delete x;
    *out = nullptr;
    return rc;
  }
  *out = x;
  return 0;
}

void dcp_db_close(struct dcp_db *x) { delete x; }
int dcp_db_num_proteins(struct dcp_db const *x) { return x ? x->reader.num_proteins() : 0; }
float dcp_db_epsilon(struct dcp_db const *x) { return x ? x->reader.header().epsilon : 0.0f; }
int dcp_db_entry_dist(struct dcp_db const *x) { return x ? x->reader.header().entry_dist : 0; }
int dcp_db_has_ga(struct dcp_db const *x) { return x ? (int)x->reader.header().has_ga : 0; }

int64_t dcp_db_protein_offset(struct dcp_db const *x, int i)
{
  if (!x || i < 0 || i > x->reader.num_proteins()) return -1;
  return x->reader.protein_offset(i);
}

int dcp_db_protein_core_size(struct dcp_db const *x, int i, int *core_size)
{
  if (!x || !core_size) return DCP_EFUNCUSE;
  std::string acc;
  return x->reader.read_protein_head(i, *core_size, acc);
}

int dcp_db_core_sizes(struct dcp_db const *x, int32_t *core_sizes)
{
  if (!x || !core_sizes) return DCP_EFUNCUSE;
  std::string acc;
  for (int i = 0; i < x->reader.num_proteins(); ++i)
  {
    int K = 0;
    int rc = x->reader.read_protein_head(i, K, acc);
    if (rc) return rc;
    core_sizes[i] = K;
  }
  return 0;
}

int dcp_db_partition_bounds(struct dcp_db const *x, int nparts, int balanced, int32_t *first)
{
  if (!x || !first) return DCP_EFUNCUSE;
  if (nparts < 1) return DCP_EZEROPART;
  int const n = x->reader.num_proteins();
  std::vector<int32_t> K;
  if (balanced)
  {
    K.resize((size_t)n);
    int rc = dcp_db_core_sizes(x, K.data());
    if (rc) return rc;
  }
  dcp_partition_bounds(n, balanced ? K.data() : nullptr, nparts, balanced != 0, first);
  return 0;
}

int dcp_db_read_nuclt_dist(struct dcp_db const *x, int i, float *nucltp, float *codonm, int *gencode)
{
  if (!x) return DCP_EFUNCUSE;
  DcpDecoder d;
  int rc = x->reader.read_decoder(i, d);
  if (rc) return rc;
  if (nucltp) memcpy(nucltp, d.nucltp.data(), d.nucltp.size() * sizeof(float));
  if (codonm) memcpy(codonm, d.codonm.data(), d.codonm.size() * sizeof(float));
  if (gencode) *gencode = d.gencode;
  return 0;
}

int dcp_decode_quasi_codon(float epsilon, float const *nucltp4, float const *codonm125, uint8_t const *nt, int n,
                           uint8_t codon[3])
{
  if (!nucltp4 || !codonm125 || !nt || !codon) return DCP_EFUNCUSE;
  for (int i = 0; i < n && i < 5; ++i)
    if (nt[i] > 3) return DCP_ESEQABC;
  return dcp_decode_codon(epsilon, nucltp4, codonm125, nt, n, codon) ? 0 : DCP_EDECODON;
}

char dcp_gencode_amino_of(int gencode_id, uint8_t const codon[3]) { return dcp_gencode_amino(gencode_id, codon); }

int dcp_partition_bounds_of(int n, int32_t const *core_sizes, int nparts, int balanced, int32_t *first)
{
  if (n < 0 || !first || (balanced && n > 0 && !core_sizes)) return DCP_EFUNCUSE;
  if (nparts < 1) return DCP_EZEROPART;
  dcp_partition_bounds(n, core_sizes, nparts, balanced != 0, first);
  return 0;
}

int dcp_db_read_protein(struct dcp_db const *x, int i, float *node_trans, float *node_emission, float *BMk,
                        float *null_lprob, float *bg_lprob, char *accession, char *consensus)
{
  if (!x) return DCP_EFUNCUSE;
  DcpProtein p;
  int rc = x->reader.read_protein(i, p);
  if (rc) return rc;
  if (node_trans) memcpy(node_trans, p.trans.data(), p.trans.size() * sizeof(float));
  if (node_emission) memcpy(node_emission, p.emission.data(), p.emission.size() * sizeof(float));
  if (BMk) memcpy(BMk, p.BMk.data(), p.BMk.size() * sizeof(float));
  if (null_lprob) memcpy(null_lprob, p.null_emission.data(), p.null_emission.size() * sizeof(float));
  if (bg_lprob) memcpy(bg_lprob, p.bg_emission.data(), p.bg_emission.size() * sizeof(float));
  if (accession)
  {
    strncpy(accession, p.accession.c_str(), 31);
    accession[31] = 0;
  }
  if (consensus) memcpy(consensus, p.consensus.c_str(), p.consensus.size() + 1);
  return 0;
}

void dcp_window_setup(struct dcp_window *w, int seq_size, int core_size)
{
  w->core_size = core_size;
  w->seq_size = seq_size;
  w->start = -1;
  w->stop = 0;
  w->idx = -1;
  w->last_hit_pos = -1;
}

int dcp_window_next(struct dcp_window *w)
{
  DcpWindow x(w->seq_size, w->core_size);
  x.start = w->start;
  x.stop = w->stop;
  x.idx = w->idx;
  x.last_hit_pos = w->last_hit_pos;
  bool ok = x.next();
  w->start = x.start;
  w->stop = x.stop;
  w->idx = x.idx;
  return ok ? 1 : 0;
}

int dcp_trellis_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, int cap, int32_t *state_ids,
                      int32_t *seqsizes, int *nsteps)
{
  if (!xnodes || !nodes || !state_ids || !seqsizes || !nsteps || K < 1 || L < 0) return DCP_EFUNCUSE;
  std::vector<int32_t> ids, sizes;
  int rc = dcp_unzip(K, L, xnodes, nodes, ids, sizes);
  if (rc) return rc;
  *nsteps = (int)ids.size();
  if ((int)ids.size() > cap) return DCP_ENOMEM;
  memcpy(state_ids, ids.data(), ids.size() * sizeof(int32_t));
  memcpy(seqsizes, sizes.data(), sizes.size() * sizeof(int32_t));
  return 0;
}

int dcp_path_hit(int nsteps, int32_t const *state_ids, int32_t const *seqsizes, int32_t hit[5])
{
  if (nsteps < 0 || !state_ids || !seqsizes || !hit) return 0;
  std::vector<int32_t> ids(state_ids, state_ids + nsteps), sizes(seqsizes, seqsizes + nsteps);
  DcpHit h;
  if (!dcp_find_hit(ids, sizes, h)) return 0;
  hit[0] = h.hit_start;
  hit[1] = h.hit_stop;
  hit[2] = h.begin_step;
  hit[3] = h.end_step;
  hit[4] = h.last_hit_pos;
  return 1;
}

void dcp_state_name_of(int state_id, char *name) { dcp_state_name(state_id, name); }
int dcp_state_is_mute_id(int state_id) { return dcp_state_is_mute(state_id) ? 1 : 0; }
float dcp_lrt_of(float null_loglik, float alt_loglik) { return dcp_lrt(null_loglik, alt_loglik); }

} // extern "C"
