// TEST INFRASTRUCTURE ONLY (see emul.cpp): the fast path pass in blocks
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"
#include "../../deciphon_amd/csrc/viterbi_pack.h"
#include <vector>

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
