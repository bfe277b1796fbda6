// emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"

template <int Q>
static void cost_q(float const *pool, DcpProfileDev const &pf, uint4 const *codes, int L, float const *xt, float *out)
{
  CostWave<Q> w;
  w.init(pool, pf, codes, xt);
  w.run(L, out);
}

template <int Q>
static float path_q(float const *pool, DcpProfileDev const &pf, uint4 const *codes, int L, float const *xt,
                    uint32_t *xnodes, uint16_t *nodes)
{
  PathWave<Q> w;
  w.init(pool, pf, codes, xt, xnodes, nodes);
  return w.run(L);
}

extern "C" int emul_cost(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L, float const *xt,
                         float *out)
{
  uint4 const *c = reinterpret_cast<uint4 const *>(codes);
  switch (pf->Q)
  {
  case 1: cost_q<1>(pool, *pf, c, L, xt, out); return 0;
  case 2: cost_q<2>(pool, *pf, c, L, xt, out); return 0;
  case 3: cost_q<3>(pool, *pf, c, L, xt, out); return 0;
  case 4: cost_q<4>(pool, *pf, c, L, xt, out); return 0;
  default: return -1;
  }
}

extern "C" int emul_path(float const *pool, DcpProfileDev const *pf, DcpCodeRow const *codes, int L, float const *xt,
                         uint32_t *xnodes, uint16_t *nodes, float *score)
{
  uint4 const *c = reinterpret_cast<uint4 const *>(codes);
  switch (pf->Q)
  {
  case 1: *score = path_q<1>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 2: *score = path_q<2>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 3: *score = path_q<3>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  case 4: *score = path_q<4>(pool, *pf, c, L, xt, xnodes, nodes); return 0;
  default: return -1;
  }
}
