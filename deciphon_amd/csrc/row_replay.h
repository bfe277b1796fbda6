// row_replay.h -- the trellis words of ONE row from the DP table, pass by pass.
//
// With the final value of every state of every row in HBM (the table the fast path pass
// keeps, traceback.h), the back-pointers of row l depend on rows l-5 .. l-1 only through
// values that are already final: every row can be replayed on its own, in the reference's
// order -- passes t = min(5,l) .. 1; per pass N, B, J, C, then M (BM,MM,IM,DM) and I (II,MI)
// of every position, D <- M+MD, the cross-lane E of an 8-lane build, the D->D chain, B and
// T (c-core/viterbi.c:485-586) -- with strict-< updates, which is where exact fp32 ties get
// their winner.  One row is one thread's work here: it is the pass-by-pass path kernel for
// profiles beyond 4096 positions, where the register-resident kernel (PathWave) does not
// reach; rows of a window run side by side, so the time of a window is one row's time.
// Scalar on purpose: the same source is compiled for the host and checked against the
// oracle there (tests/emul).
#pragma once
#include "traceback.h"

#define DCP_RR_UPD(cur, ptr, val, newptr)                                      \
  do                                                                           \
  {                                                                            \
    float const v_ = (val);                                                    \
    float const x_ = __builtin_fminf((cur), v_);                               \
    if (!(x_ == (cur))) (ptr) = (newptr);                                      \
    (cur) = x_;                                                                \
  } while (0)

// acc: scratch of 3*K floats (the running M, I, D of the row).  Writes xnode (c-core/trellis.h:
// 42-56) and nodes[0..K) (c-core/trellis.h:12-21, c-core/viterbi.c:631-694).  l >= 1.
DCP_HD void dcp_replay_row(DcpTraceIn const &in, int l, float *acc, uint32_t *xnode, uint16_t *nodes)
{
  float const INF = __builtin_inff();
  int const K = in.K, Kp = in.Kp;
  size_t const stride = (size_t)Kp + DCP_ROW_HDR;
  auto SP = [&](int z, int i) { return in.sp[(size_t)z * DCP_SP_STRIDE + i]; }; // 0 N, 1 B, 2 J, 3 E, 4 C
  auto CELL = [&](int z, int s, int k) { return k < 0 ? INF : in.cells[((size_t)z * 3 + s) * (size_t)Kp + k]; };
  auto TR = [&](int id, int k) { return in.trans[(size_t)id * Kp + k]; };
  float const *xt = in.xt;
  float *Ma = acc, *Ia = acc + K, *Da = acc + 2 * (size_t)K;

  // the 8-lane layout of the reference build the goldens come from (c-core/viterbi.c:195-199,220-221)
  int Qr = (K - 1) / DCP_REF_LANES + 1;
  if (Qr < 2) Qr = 2;

  float aN = INF, aB = INF, aJ = INF, aE = INF, aC = INF, aT = INF;
  float const aS = INF; // S exists at row 0 only (c-core/viterbi.c:471-473)
  unsigned pN = 0, pB = 0, pJ = 0, pE = 0, pC = 0, pT = 0;
  for (int k = 0; k < K; ++k)
  {
    Ma[k] = Ia[k] = Da[k] = INF;
    nodes[k] = 0; // pointer fields: M bits 0-4, D bit 5, I bits 6-9 (c-core/state.h:27-39)
  }

  for (int t = l < 5 ? l : 5; t > 0; --t)
  {
    int const z = l - t;
    unsigned const u = (unsigned)(t - 1);
    float const *row = in.rows + (size_t)in.codes[l].c[t - 1] * stride;
    float const nil = row[0], bg = row[1];
    float const *ma = row + DCP_ROW_HDR;
    float const zS = z == 0 ? 0.0f : INF, zN = SP(z, 0), zB = SP(z, 1), zJ = SP(z, 2), zE = SP(z, 3), zC = SP(z, 4);

    DCP_RR_UPD(aN, pN, zS + xt[DCP_SN] + nil, 0 + u); // c-core/viterbi.c:492-493
    DCP_RR_UPD(aN, pN, zN + xt[DCP_NN] + nil, 5 + u);
    DCP_RR_UPD(aB, pB, aS + xt[DCP_SB], 0); // :495-496
    DCP_RR_UPD(aB, pB, aN + xt[DCP_NB], 1);
    DCP_RR_UPD(aJ, pJ, zE + xt[DCP_EJ] + nil, 0 + u); // :498-499
    DCP_RR_UPD(aJ, pJ, zJ + xt[DCP_JJ] + nil, 5 + u);
    DCP_RR_UPD(aC, pC, zE + xt[DCP_EC] + nil, 0 + u); // :501-502
    DCP_RR_UPD(aC, pC, zC + xt[DCP_CC] + nil, 5 + u);

    // what the stale D candidate of each reference lane's first position sees: M of the
    // position before it and its own D, both BEFORE this pass (c-core/viterbi.c:507,538)
    float lastM_before[DCP_REF_LANES], D_before[DCP_REF_LANES];
    for (int e = 0; e < DCP_REF_LANES; ++e)
    {
      int const k = e * Qr;
      lastM_before[e] = k > 0 && k - 1 < K ? Ma[k - 1] : INF;
      D_before[e] = k < K ? Da[k] : INF;
    }

    for (int k = 0; k < K; ++k) // :512-536, one position at a time
    {
      unsigned pm = nodes[k] & 31u, pi = (nodes[k] >> 6) & 15u;
      DCP_RR_UPD(Ma[k], pm, (zB + TR(DCP_BM, k)) + ma[k], 0 + u);
      DCP_RR_UPD(Ma[k], pm, (CELL(z, 0, k - 1) + TR(DCP_MM, k)) + ma[k], 5 + u);
      DCP_RR_UPD(Ma[k], pm, (CELL(z, 1, k - 1) + TR(DCP_IM, k)) + ma[k], 10 + u);
      DCP_RR_UPD(Ma[k], pm, (CELL(z, 2, k - 1) + TR(DCP_DM, k)) + ma[k], 15 + u);
      DCP_RR_UPD(Ia[k], pi, (CELL(z, 1, k) + TR(DCP_II, k)) + bg, 5 + u);
      DCP_RR_UPD(Ia[k], pi, (CELL(z, 0, k) + TR(DCP_MI, k)) + bg, 0 + u);
      nodes[k] = (uint16_t)((nodes[k] & (1u << 5)) | pm | (pi << 6));
    }
    for (int k = 0; k < K; ++k) // :538 and the stripe-0 repair :553-555
    {
      unsigned pd = (nodes[k] >> 5) & 1u;
      DCP_RR_UPD(Da[k], pd, (k > 0 ? Ma[k - 1] : INF) + TR(DCP_MD, k), 0);
      nodes[k] = (uint16_t)((nodes[k] & ~(1u << 5)) | (pd << 5));
    }

    // E of this pass (:540-541,556-558; intrinsics.h:151-160): every reference lane keeps the
    // first candidate that attains its own minimum in the order ME(0), DE(0)*, ME(1), DE(1), ...,
    // DE(0); the lanes are merged by the maximum packed (name << 28 | lane << 24 | q) among those
    // equal to the minimum
    {
      float best = INF;
      float lane_val[DCP_REF_LANES];
      uint32_t lane_ptr[DCP_REF_LANES];
      for (int e = 0; e < DCP_REF_LANES; ++e)
      {
        float v = INF;
        uint32_t p = 0;
        for (int q = 0; q < Qr; ++q)
        {
          int const k = e * Qr + q;
          float const m = k < K ? Ma[k] : INF;
          float d = INF;
          if (k < K) d = q == 0 ? __builtin_fminf(D_before[e], lastM_before[e] + TR(DCP_MD, k)) : Da[k];
          DCP_RR_UPD(v, p, m, (0x1u << 28) | (uint32_t)q);
          DCP_RR_UPD(v, p, d, (0x2u << 28) | (uint32_t)q);
        }
        {
          int const k = e * Qr;
          DCP_RR_UPD(v, p, k < K ? Da[k] : INF, (0x2u << 28) | 0u);
        }
        lane_val[e] = v;
        lane_ptr[e] = p | ((uint32_t)e << 24);
        best = __builtin_fminf(best, v);
      }
      uint32_t bestptr = 0;
      bool any = false;
      for (int e = 0; e < DCP_REF_LANES; ++e)
        if (lane_val[e] == best)
        {
          if (!any || lane_ptr[e] > bestptr) bestptr = lane_ptr[e];
          any = true;
        }
      aE = best;
      int const k = (int)((bestptr >> 24) & 0xF) * Qr + (int)(bestptr & 0x00FFFFFFu);
      pE = bestptr & (0x1u << 28) ? (unsigned)(2 * k) : bestptr & (0x2u << 28) ? (unsigned)(2 * k + 1) : 0u; // :676-680
    }

    for (int k = 1; k < K; ++k) // :561-580: the lazy D->D loop is one serial chain
    {
      unsigned pd = (nodes[k] >> 5) & 1u;
      DCP_RR_UPD(Da[k], pd, Da[k - 1] + TR(DCP_DD, k), 1);
      nodes[k] = (uint16_t)((nodes[k] & ~(1u << 5)) | (pd << 5));
    }

    DCP_RR_UPD(aB, pB, aE + xt[DCP_EB], 2); // :582-583
    DCP_RR_UPD(aB, pB, aJ + xt[DCP_JB], 3);
    DCP_RR_UPD(aT, pT, aE + xt[DCP_ET], 0); // :585-586
    DCP_RR_UPD(aT, pT, aC + xt[DCP_CT], 1);
  }

  // after(), c-core/viterbi.c:631-694; shifts c-core/trellis.h:42-56
  *xnode = (pN << 0) | (pB << 4) | (pE << 6) | (pC << 21) | (pT << 25) | (pJ << 26);
  nodes[0] = (uint16_t)(nodes[0] & ~(1u << 5));          // position 0 has no D pointer
  nodes[K - 1] = (uint16_t)(nodes[K - 1] & ~(15u << 6)); // the last position has no I pointer
}
