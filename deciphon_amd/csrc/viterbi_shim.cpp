// viterbi_shim.cpp -- include/dcp_viterbi.h: the reference's per-problem DP
// interface (c-core/viterbi.h) served by the GPU engine, one window per call.
#include "../../include/dcp_viterbi.h"
#include "../../include/deciphon_hip.h"
#include "dcp_errors.h"
#include "dcp_types.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct viterbi
{
  dcp_hip *eng = nullptr;
  int K = 0;
  std::vector<float> trans, match, nul, bg;
  float xt[DCP_NUM_XTRANS];
  bool dirty = true;
  struct trellis tr;
  std::vector<uint32_t> xnodes;
  std::vector<uint16_t> nodes;
};

namespace
{

// evaluates the callback into nucleotide indices and checks that it describes a sequence
bool sequence_of(int L, viterbi_code_fn fn, void *arg, std::vector<uint8_t> &nt)
{
  static int const off[6] = {0, 0, 4, 20, 84, 340};
  nt.resize((size_t)L);
  for (int pos = 0; pos < L; ++pos)
  {
    int const c = fn(pos, 1, arg);
    if (c < 0 || c > 3) return false;
    nt[(size_t)pos] = (uint8_t)c;
  }
  for (int len = 2; len <= 5; ++len)
    for (int pos = 0; pos + len <= L; ++pos)
    {
      int idx = 0;
      for (int i = 0; i < len; ++i) idx = idx * 4 + nt[(size_t)(pos + i)];
      if (fn(pos, len, arg) != off[len] + idx) return false;
    }
  return true;
}

int stage(viterbi *x, int L, viterbi_code_fn fn, void *arg, dcp_hip_window *w)
{
  if (!x || !x->eng || x->K < 1 || L < 1 || !fn) return DCP_EFUNCUSE;
  std::vector<uint8_t> nt;
  if (!sequence_of(L, fn, arg, nt)) return DCP_EFUNCUSE;
  int rc = 0;
  if (x->dirty)
  {
    dcp_hip_clear_profiles(x->eng);
    int idx = 0;
    if ((rc = dcp_hip_add_profile(x->eng, x->K, x->trans.data(), x->match.data(), x->nul.data(), x->bg.data(), &idx)))
      return rc;
    if ((rc = dcp_hip_commit_profiles(x->eng))) return rc;
    x->dirty = false;
  }
  int64_t const offs[2] = {0, L};
  if ((rc = dcp_hip_set_sequences(x->eng, 1, nt.data(), offs))) return rc;
  // the caller's own special transitions, whatever the window length
  int const rows = (L / 3 > 1 ? L / 3 : 1) + 1;
  std::vector<float> table((size_t)rows * DCP_NUM_XTRANS);
  for (int r = 0; r < rows; ++r) memcpy(table.data() + (size_t)r * DCP_NUM_XTRANS, x->xt, sizeof(x->xt));
  if ((rc = dcp_hip_set_mode(x->eng, 1, 0))) return rc;
  if ((rc = dcp_hip_set_xtrans_table(x->eng, rows, table.data()))) return rc;
  *w = dcp_hip_window{0, 0, 0, L};
  return 0;
}

} // namespace

extern "C" {

struct viterbi *viterbi_new(void)
{
  int device = 0;
  if (char const *d = getenv("DECIPHON_HIP_DEVICE")) device = atoi(d);
  dcp_hip *eng = dcp_hip_new(device);
  if (!eng) return nullptr;
  viterbi *x = new viterbi;
  x->eng = eng;
  memset(&x->tr, 0, sizeof(x->tr));
  for (float &v : x->xt) v = INFINITY;
  return x;
}

void viterbi_del(struct viterbi const *cx)
{
  viterbi *x = const_cast<viterbi *>(cx);
  if (!x) return;
  dcp_hip_del(x->eng);
  delete x;
}

int viterbi_setup(struct viterbi *x, int K) // c-core/viterbi.c:336-381: everything +inf
{
  if (!x || K < 1) return DCP_EFUNCUSE;
  x->K = K;
  x->trans.assign((size_t)DCP_NUM_TRANS * K, INFINITY);
  x->match.assign((size_t)DCP_TABLE_SIZE * K, INFINITY);
  x->nul.assign(DCP_TABLE_SIZE, INFINITY);
  x->bg.assign(DCP_TABLE_SIZE, INFINITY);
  // extr_trans_init (c-core/viterbi.c:272-286) leaves RR untouched; it is set before use
  for (int i = 0; i < DCP_NUM_XTRANS; ++i)
    if (i != EXTR_TRANS_RR) x->xt[i] = INFINITY;
  x->dirty = true;
  return 0;
}

void viterbi_set_extr_trans(struct viterbi *x, enum extr_trans_id id, float scalar) { x->xt[(int)id] = scalar; }

void viterbi_set_core_trans(struct viterbi *x, enum core_trans_id id, float scalar, int k)
{
  x->trans[(size_t)id * (size_t)x->K + (size_t)k] = scalar;
  x->dirty = true;
}

void viterbi_set_null(struct viterbi *x, float scalar, int code)
{
  x->nul[(size_t)code] = scalar;
  x->dirty = true;
}

void viterbi_set_background(struct viterbi *x, float scalar, int code)
{
  x->bg[(size_t)code] = scalar;
  x->dirty = true;
}

void viterbi_set_match(struct viterbi *x, float scalar, int k, int code)
{
  x->match[(size_t)code * (size_t)x->K + (size_t)k] = scalar;
  x->dirty = true;
}

float viterbi_null(struct viterbi *x, int L, viterbi_code_fn fn, void *arg)
{
  if (x && L == 0) return -x->xt[EXTR_TRANS_RR]; // R[0] = -RR, c-core/viterbi.c:703,718
  dcp_hip_window w;
  float nul = NAN, alt = NAN;
  if (stage(x, L, fn, arg, &w) || dcp_hip_cost(x->eng, 1, &w, &nul, &alt)) return NAN;
  return nul;
}

float viterbi_cost(struct viterbi *x, int L, viterbi_code_fn fn, void *arg)
{
  if (x && L == 0) return INFINITY; // xs[0].T of an untouched row, c-core/viterbi.c:599
  dcp_hip_window w;
  float nul = NAN, alt = NAN;
  if (stage(x, L, fn, arg, &w) || dcp_hip_cost(x->eng, 1, &w, &nul, &alt)) return NAN;
  return alt;
}

int viterbi_path(struct viterbi *x, int L, viterbi_code_fn fn, void *arg)
{
  dcp_hip_window w;
  int rc = stage(x, L, fn, arg, &w);
  if (rc) return rc;
  if ((rc = dcp_hip_path(x->eng, 1, &w))) return rc;
  uint32_t const *xn = nullptr;
  uint16_t const *nd = nullptr;
  if ((rc = dcp_hip_path_trellis(x->eng, 0, &xn, &nd))) return rc;
  x->xnodes.assign(xn, xn + (L + 1));
  x->nodes.assign(nd, nd + (size_t)(L + 1) * (size_t)x->K);
  x->tr.core_size = x->K;
  x->tr.xnodes = x->tr.xnode = x->xnodes.data();
  x->tr.nodes = x->tr.node = x->nodes.data();
  return 0;
}

struct trellis *viterbi_trellis(struct viterbi *x) { return &x->tr; }

int viterbi_table_size(void) { return DCP_TABLE_SIZE; }

} // extern "C"
