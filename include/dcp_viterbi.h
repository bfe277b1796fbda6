/* dcp_viterbi.h -- per-problem drop-in for the reference's DP engine interface,
 * c-core/viterbi.h:4-52: same symbol names, argument meaning and return values, so
 * that protein_setup_viterbi (c-core/protein.c:353-394), xtrans_setup_viterbi
 * (c-core/xtrans.c:53-68), process_window (c-core/thread.c:114-128) and
 * c-core/test_protein.c:69-76 link against it unchanged.
 *
 * One call = one DP on the GPU, synchronously: this is the parity seam, not the
 * throughput path (that is deciphon_hip.h).  The code callback cannot run on the
 * device, so it is evaluated on the host for every (pos, len) first; it has to
 * describe a sequence -- code_fn(pos, len) must equal the code of the len-mer
 * spelled by code_fn(pos..pos+len-1, 1), as imm_eseq_get does -- otherwise
 * viterbi_null / viterbi_cost return NaN and viterbi_path returns DCP_EFUNCUSE.
 * Every run starts from the state viterbi_setup leaves (the reference keeps the
 * previous run's last DP row as row 0, c-core/viterbi.c:471-473).
 * The HIP device is DECIPHON_HIP_DEVICE (default 0); viterbi_new returns NULL
 * when it cannot be used.
 */
#ifndef DCP_VITERBI_H
#define DCP_VITERBI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum extr_trans_id
{
  EXTR_TRANS_RR, EXTR_TRANS_SN, EXTR_TRANS_NN, EXTR_TRANS_SB, EXTR_TRANS_NB, EXTR_TRANS_EB, EXTR_TRANS_JB,
  EXTR_TRANS_EJ, EXTR_TRANS_JJ, EXTR_TRANS_EC, EXTR_TRANS_CC, EXTR_TRANS_ET, EXTR_TRANS_CT,
};

enum core_trans_id
{
  CORE_TRANS_BM, CORE_TRANS_MM, CORE_TRANS_MI, CORE_TRANS_MD, CORE_TRANS_IM, CORE_TRANS_II, CORE_TRANS_DM,
  CORE_TRANS_DD,
};

typedef int (*viterbi_code_fn)(int pos, int len, void *arg);

/* layout of c-core/trellis.h:12-21 */
struct trellis
{
  int core_size;
  uint32_t *xnodes; /* [L+1] */
  uint16_t *nodes;  /* [(L+1) * core_size] */
  uint32_t *xnode;
  uint16_t *node;
};

struct viterbi;

struct viterbi *viterbi_new(void);
void viterbi_del(struct viterbi const *);

int viterbi_setup(struct viterbi *, int K);
void viterbi_set_extr_trans(struct viterbi *, enum extr_trans_id, float scalar);
void viterbi_set_core_trans(struct viterbi *, enum core_trans_id, float scalar, int k);
void viterbi_set_null(struct viterbi *, float scalar, int code);
void viterbi_set_background(struct viterbi *, float scalar, int code);
void viterbi_set_match(struct viterbi *, float scalar, int k, int code);
float viterbi_null(struct viterbi *, int L, viterbi_code_fn, void *);
float viterbi_cost(struct viterbi *, int L, viterbi_code_fn, void *);
int viterbi_path(struct viterbi *, int L, viterbi_code_fn, void *);

struct trellis *viterbi_trellis(struct viterbi *);
int viterbi_table_size(void);

#ifdef __cplusplus
}
#endif

#endif
