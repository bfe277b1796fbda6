// emul_strip.cpp (see emul.cpp) -- TEST INFRASTRUCTURE ONLY.
// Instantiates the kernel logic (deciphon_amd/csrc/viterbi_body.h) on the
// lock-step wave emulator and exports it with a C ABI for tests/test_emul_*.py.
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"

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

