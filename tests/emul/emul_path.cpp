// emul_path.cpp (see emul.cpp) -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"

template <int Q, int W>
static float path_q(float const *pool, DcpProfileDev const &pf, DcpCodeRow const *codes, int L, float const *xt,
                    uint32_t *xnodes, uint16_t *nodes)
{
  static thread_local PathWave<Q, W> w;
  w.init(pool, pf, codes, xt, xnodes, nodes);
  return w.run(L);
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

