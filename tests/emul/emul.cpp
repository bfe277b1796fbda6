// emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"
#include "../../deciphon_amd/csrc/viterbi_pack.h"

template <int Q, int W>
static void cost_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt, float *out)
{
  static thread_local CostWave<Q, W> w; // 64*W-lane vectors are large: keep them off the stack
  w.init(pool, pf, codes, xt);
  w.run(L, out);
}

template <int Q, int W>
static float path_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt,
                    uint32_t *xnodes, uint16_t *nodes)
{
  static thread_local PathWave<Q, W> w;
  w.init(pool, pf, codes, xt, xnodes, nodes);
  return w.run(L);
}

extern "C" int emul_cost(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L, float const *xt,
                         float *out)
{
  DcpCodeRow const *c = codes;
  switch (pf->Q * 100 + pf->W)
  {
  case 101: cost_q<1, 1>(pool, *pf, c, L, xt, out); return 0;
  case 201: cost_q<2, 1>(pool, *pf, c, L, xt, out); return 0;
  case 301: cost_q<3, 1>(pool, *pf, c, L, xt, out); return 0;
  case 401: cost_q<4, 1>(pool, *pf, c, L, xt, out); return 0;
  case 801: cost_q<8, 1>(pool, *pf, c, L, xt, out); return 0;
  case 802: cost_q<8, 2>(pool, *pf, c, L, xt, out); return 0;
  case 804: cost_q<8, 4>(pool, *pf, c, L, xt, out); return 0;
  case 808: cost_q<8, 8>(pool, *pf, c, L, xt, out); return 0;
  case 601: cost_q<6, 1>(pool, *pf, c, L, xt, out); return 0;
  case 602: cost_q<6, 2>(pool, *pf, c, L, xt, out); return 0;
  case 604: cost_q<6, 4>(pool, *pf, c, L, xt, out); return 0;
  case 302: cost_q<3, 2>(pool, *pf, c, L, xt, out); return 0;
  case 304: cost_q<3, 4>(pool, *pf, c, L, xt, out); return 0;
  case 308: cost_q<3, 8>(pool, *pf, c, L, xt, out); return 0;
  case 402: cost_q<4, 2>(pool, *pf, c, L, xt, out); return 0;
  case 404: cost_q<4, 4>(pool, *pf, c, L, xt, out); return 0;
  case 408: cost_q<4, 8>(pool, *pf, c, L, xt, out); return 0;
  case 416: cost_q<4, 16>(pool, *pf, c, L, xt, out); return 0;
  default: return -1;
  }
}

extern "C" int emul_path(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L, float const *xt,
                         uint32_t *xnodes, uint16_t *nodes, float *score)
{
  DcpCodeRow const *c = codes;
  switch (pf->Q * 100 + pf->W)
  {
  case 101: *score = path_q<1, 1>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 201: *score = path_q<2, 1>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 301: *score = path_q<3, 1>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 401: *score = path_q<4, 1>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 302: *score = path_q<3, 2>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 304: *score = path_q<3, 4>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 308: *score = path_q<3, 8>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 402: *score = path_q<4, 2>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 404: *score = path_q<4, 4>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 408: *score = path_q<4, 8>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 416: *score = path_q<4, 16>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  default: return -1;
  }
}

// ---- fast path pass: cost pass with the DP table stored, then the scalar traceback ----
template <int Q, int W>
static void store_q_(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt,
                     float *out, float *cells, float *sp)
{
  static thread_local CostWave<Q, W, true> w;
  w.tab_cells = cells;
  w.tab_sp = sp;
  w.init(pool, pf, codes, xt);
  w.run(L, out);
}

extern "C" int emul_cost_store(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L,
                               float const *xt, float *out, float *cells, float *sp)
{
  switch (pf->Q * 100 + pf->W)
  {
  case 101: store_q_<1, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 201: store_q_<2, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 301: store_q_<3, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 401: store_q_<4, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 801: store_q_<8, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 802: store_q_<8, 2>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 804: store_q_<8, 4>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 601: store_q_<6, 1>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 602: store_q_<6, 2>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 302: store_q_<3, 2>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 304: store_q_<3, 4>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 402: store_q_<4, 2>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  case 404: store_q_<4, 4>(pool, *pf, codes, L, xt, out, cells, sp); return 0;
  default: return -1;
  }
}

extern "C" int emul_traceback(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L,
                              float const *xt, float const *cells, float const *sp, uint32_t *buf, long cap)
{
  DcpTraceIn in;
  in.K = pf->K;
  in.Kp = pf->Kp;
  in.L = L;
  in.sp = sp;
  in.cells = cells;
  in.rows = pool + pf->rows_off;
  in.trans = pool + pf->trans_off;
  in.codes = codes;
  in.xt = xt;
  return dcp_traceback(in, buf, cap);
}

// rows of multi-wave cost passes that left the one-barrier protocol since the last call
extern "C" long emul_fallback_rows(void)
{
  long const n = em_fallback_rows;
  em_fallback_rows = 0;
  return n;
}

// ---- StripWave: profiles longer than one workgroup's registers, strip by strip ----
template <int Q, int W>
static void strip_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt,
                    float *out, float *ring, float *cells, float *sp)
{
  if (cells)
  {
    static thread_local StripWave<Q, W, true> w;
    w.tab_cells = cells;
    w.tab_sp = sp;
    w.ring = ring;
    w.tick = 0;
    w.init(pool, pf, codes, xt);
    w.run(L, out);
  }
  else
  {
    static thread_local StripWave<Q, W, false> w;
    w.ring = ring;
    w.tick = 0;
    w.init(pool, pf, codes, xt);
    w.run(L, out);
  }
}

// ring: float[10 * Kp] scratch; cells/sp: the DP table or NULL
extern "C" int emul_strip_cost(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L,
                               float const *xt, float *out, float *ring, float *cells, float *sp)
{
  switch (pf->Q * 100 + pf->W)
  {
  case 101: strip_q<1, 1>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  case 201: strip_q<2, 1>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  case 102: strip_q<1, 2>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  case 202: strip_q<2, 2>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  case 402: strip_q<4, 2>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  case 104: strip_q<1, 4>(pool, *pf, codes, L, xt, out, ring, cells, sp); return 0;
  default: return -1;
  }
}

// ---- the pass-by-pass trellis replayed row by row from the DP table (row_replay.h) ----
#include "../../deciphon_amd/csrc/row_replay.h"
#include <vector>
extern "C" int emul_replay(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L,
                           float const *xt, float const *cells, float const *sp, uint32_t *xnodes, uint16_t *nodes)
{
  DcpTraceIn in;
  in.K = pf->K;
  in.Kp = pf->Kp;
  in.L = L;
  in.sp = sp;
  in.cells = cells;
  in.rows = pool + pf->rows_off;
  in.trans = pool + pf->trans_off;
  in.codes = codes;
  in.xt = xt;
  std::vector<float> acc((size_t)3 * pf->K);
  xnodes[0] = 0;
  for (int k = 0; k < pf->K; ++k) nodes[k] = 0;
  for (int l = 1; l <= L; ++l) dcp_replay_row(in, l, acc.data(), xnodes + l, nodes + (size_t)l * pf->K);
  return 0;
}

// ---- several windows per wavefront (viterbi_pack.h) ----
template <int Q, int S>
static void pack_qs(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, uint32_t ncodes, float const *xt_table,
                    DcpPack const &pk, float *out)
{
  static thread_local PackWave<Q, S> w;
  em_lanes = 64;
  w.init(pool, pf, codes, ncodes, xt_table, pk);
  w.run(pk.Lmax, out, pk, xt_table);
}

// the same with the rows of the first NLDS emission lengths read from an "LDS" copy of the table
template <int Q, int S, int NLDS>
static void pack_lds_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, uint32_t ncodes, float const *xt_table,
                       DcpPack const &pk, float *out)
{
  int const RL = DCP_PACK_LDS_ROW(Q, S), NR = DCP_PACK_LDS_ROWS(NLDS);
  std::vector<float> table((size_t)NR * RL, INFINITY);
  for (int c = 0; c < NR; ++c)
    for (int j = 0; j < RL && j < pf.Kp + DCP_ROW_HDR; ++j)
      table[(size_t)c * RL + j] = pool[pf.rows_off + (size_t)c * (pf.Kp + DCP_ROW_HDR) + j];
  static thread_local PackWave<Q, S, dcp_lazy_turns(Q), NLDS> w;
  em_lanes = 64;
  w.init(pool, pf, codes, ncodes, xt_table, pk, table.data());
  w.run(pk.Lmax, out, pk, xt_table);
}

extern "C" int emul_cost_pack_lds(int Q, int S, float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes,
                                  uint32_t ncodes, float const *xt_table, DcpPack const *pk, float *out)
{
  switch (Q * 100 + S)
  {
  case 104: pack_lds_q<1, 4, 5>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 204: pack_lds_q<2, 4, 5>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 404: pack_lds_q<4, 4, 5>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 208: pack_lds_q<2, 8, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 408: pack_lds_q<4, 8, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 216: pack_lds_q<2, 16, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 316: pack_lds_q<3, 16, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 416: pack_lds_q<4, 16, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 232: pack_lds_q<2, 32, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 332: pack_lds_q<3, 32, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 432: pack_lds_q<4, 32, 3>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  default: return -1;
  }
}

extern "C" int emul_cost_pack(int Q, int S, float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes,
                              uint32_t ncodes, float const *xt_table, DcpPack const *pk, float *out)
{
  switch (Q * 100 + S)
  {
  case 104: pack_qs<1, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 204: pack_qs<2, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 404: pack_qs<4, 4>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 208: pack_qs<2, 8>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 408: pack_qs<4, 8>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 216: pack_qs<2, 16>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 232: pack_qs<2, 32>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 316: pack_qs<3, 16>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 416: pack_qs<4, 16>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 332: pack_qs<3, 32>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 432: pack_qs<4, 32>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 632: pack_qs<6, 32>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  case 832: pack_qs<8, 32>(pool, *pf, codes, ncodes, xt_table, *pk, out); return 0;
  default: return -1;
  }
}

// ---- the fast path pass in blocks (dcp_types.h): checkpoints, then block by block from the last to the first ----
template <int Q, int W>
static int path_blocks_qw(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt, int B,
                          uint32_t *buf, long cap, float *score)
{
  int const nb = dcp_num_blocks(L, B);
  size_t const cf = (size_t)dcp_ckpt_floats(pf.Kp, W);
  std::vector<float> ckpt((size_t)(nb > 1 ? nb - 1 : 0) * cf, NAN);
  float out[2] = {NAN, NAN};
  if (nb > 1)
  {
    static thread_local CostWave<Q, W> w;
    w.ckpt_in = nullptr;
    w.row_base = 0;
    w.ckpt_out = ckpt.data();
    w.ckpt_every = B;
    w.init(pool, pf, codes, xt);
    w.run(L, out);
  }
  int const slots = dcp_block_slots(L, B);
  std::vector<float> sp((size_t)slots * DCP_SP_STRIDE), cells((size_t)slots * 3 * pf.Kp);
  DcpTraceState st;
  memset(&st, 0, sizeof st);
  int r = 0;
  for (int block = nb - 1; block >= 0 && r == 0; --block)
  {
    std::fill(sp.begin(), sp.end(), NAN); // nothing may be read that this block did not write
    std::fill(cells.begin(), cells.end(), NAN);
    static thread_local CostWave<Q, W, true> w;
    w.ckpt_out = nullptr;
    w.ckpt_every = 0;
    w.tab_sp = sp.data();
    w.tab_cells = cells.data();
    w.row_base = block * B;
    w.ckpt_in = block > 0 ? ckpt.data() + (size_t)(block - 1) * cf : nullptr;
    w.init(pool, pf, codes, xt);
    int const last = B > 0 ? (block + 1) * B + 5 : L;
    w.run(L, out, last < L ? last : L);
    DcpTraceIn in;
    in.K = pf.K;
    in.Kp = pf.Kp;
    in.L = L;
    in.sp = sp.data();
    in.cells = cells.data();
    in.rows = pool + pf.rows_off;
    in.trans = pool + pf.trans_off;
    in.codes = codes;
    in.xt = xt;
    in.row_base = block * B;
    in.lo = block > 0 ? block * B + 5 : -1;
    r = dcp_traceback(in, buf, cap, &st);
  }
  *score = out[1];
  return r;
}

extern "C" int emul_path_blocks(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L, float const *xt,
                                int B, uint32_t *buf, long cap, float *score)
{
  switch (pf->Q * 100 + pf->W)
  {
  case 101: return path_blocks_qw<1, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 201: return path_blocks_qw<2, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 301: return path_blocks_qw<3, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 401: return path_blocks_qw<4, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 601: return path_blocks_qw<6, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 801: return path_blocks_qw<8, 1>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 602: return path_blocks_qw<6, 2>(pool, *pf, codes, L, xt, B, buf, cap, score);
  case 404: return path_blocks_qw<4, 4>(pool, *pf, codes, L, xt, B, buf, cap, score);
  default: return -100;
  }
}
