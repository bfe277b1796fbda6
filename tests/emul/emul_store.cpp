// emul_store.cpp (see emul.cpp) -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"

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

