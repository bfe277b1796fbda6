// TEST INFRASTRUCTURE ONLY (see emul.cpp): PackWave, several windows per wavefront
#include "lane_ops_emul.h"
#include "../../deciphon_amd/csrc/viterbi_body.h"
#include "../../deciphon_amd/csrc/traceback.h"
#include "../../deciphon_amd/csrc/viterbi_pack.h"
#include <vector>

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

