// emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"

thread_local long em_fallback_rows = 0;
thread_local long em_votes = 0, em_votes_true = 0;

template <int Q, int W>
static void cost_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt, float *out)
{
  static thread_local CostWave<Q, W, false, DCP_COST_POLICY(Q, W)> w; // 64*W-lane vectors are large: keep them off the stack
  w.init(pool, pf, codes, xt);
  w.run(L, out);
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
  case 501: cost_q<5, 1>(pool, *pf, c, L, xt, out); return 0; // on the 384-column layout of (6,1)
  case 701: cost_q<7, 1>(pool, *pf, c, L, xt, out); return 0; // on the 512-column layout of (8,1)
  case 1001: cost_q<10, 1>(pool, *pf, c, L, xt, out); return 0; // one wave on the 768-column layout of (6,2)
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

// rows of multi-wave cost passes that left the one-barrier protocol since the last call
extern "C" long emul_fallback_rows(void)
{
  long const n = em_fallback_rows;
  em_fallback_rows = 0;
  return n;
}


// votes of the lazy D->D loops since the last call: out[0] = taken, out[1] = carried (one more turn each)
extern "C" void emul_votes(long *out)
{
  out[0] = em_votes;
  out[1] = em_votes_true;
  em_votes = em_votes_true = 0;
}
