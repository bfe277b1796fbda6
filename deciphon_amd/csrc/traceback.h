// traceback.h -- the path of a window from its DP table, without a trellis.
//
// The fast path pass: the cost kernel (CostWave<.., STORE>) leaves the final value of
// every state of every row in HBM; this walk then goes from T at row L back to S at row 0
// and, at each state it visits, re-evaluates that state's candidates exactly as the
// reference forms them -- (x + transition) + emission, c-core/viterbi.c:224,526-536 -- in
// the reference's order (t = 5..1; BM,MM,IM,DM / II,MI / EC,CC ...).  The reference keeps
// the FIRST candidate that attains the minimum (strict-< updates), so the back-pointer is
// the first candidate equal to the stored value: the same step trellis_unzip would read
// from the trellis (c-core/trellis.c:51-113,147-167), at O(path) instead of O(K*L*5) work.
//
// Where the reference's choice depends on more than the final values -- an exact fp32 tie
// between MD and DD, between the candidates of B or T, or a D state tying the row minimum E
// (c-core/viterbi.c:538-586 resolves those by pass history) -- the walk gives up
// (DCP_TB_TIE) and the caller runs the literal path kernel for that window.
#pragma once
#include "dcp_types.h"

#ifndef DCP_HD
#ifdef __HIPCC__
#define DCP_HD __host__ __device__ inline
#else
#define DCP_HD inline
#endif
#endif

enum { DCP_TB_OVERFLOW = -1, DCP_TB_TIE = -2, DCP_TB_BAD = -3 };

struct DcpTraceIn
{
  int K, Kp, L;
  float const *sp;    // [(L+1)][DCP_SP_STRIDE]: N,B,J,E,C of every row
  float const *cells; // [(L+1)][3][Kp]: M,I,D
  float const *rows;  // profile emission rows [1364][DCP_ROW_HDR + Kp]
  float const *trans; // [8][Kp]
  DcpCodeRow const *codes; // row l = codes of the t-mers ending at window position l
  float const *xt;    // 13 special transitions
  // blocks (dcp_types.h): the table holds rows row_base .. and the walk stops, to be resumed on the block
  // before, at the first state whose stage is <= lo (lo < 0: never)
  int row_base = 0, lo = -1;
};

// Writes the steps (state_id | seqsize << 16) from the END of buf backwards; returns their
// number, or a DCP_TB_* code -- or 0 when the walk stopped at in.lo with where it stands in *st
// (st != NULL: resume from *st unless it is fresh, i.e. zeroed -- no state id is 0).
DCP_HD int dcp_traceback(DcpTraceIn const &in, uint32_t *buf, int64_t cap, DcpTraceState *st = nullptr)
{
  enum
  {
    ST_M = 0 << 14, ST_I = 1 << 14, ST_D = 2 << 14, ST_X = 3 << 14, // c-core/state.h:9-25
    ST_S = ST_X | 3, ST_N = ST_X | 4, ST_B = ST_X | 5, ST_E = ST_X | 6, ST_J = ST_X | 7, ST_C = ST_X | 8, ST_T = ST_X | 9,
  };
  float const INF = __builtin_inff();
  int const K = in.K, Kp = in.Kp;
  size_t const stride = (size_t)Kp + DCP_ROW_HDR;
  int const base = in.row_base;
  auto SP = [&](int l, int i) { return in.sp[(size_t)(l - base) * DCP_SP_STRIDE + i]; }; // 0 N, 1 B, 2 J, 3 E, 4 C
  auto CELL = [&](int l, int s, int k) { return k < 0 ? INF : in.cells[((size_t)(l - base) * 3 + s) * (size_t)Kp + k]; };
  auto ROW = [&](int l, int t) { return in.rows + (size_t)in.codes[l].c[t - 1] * stride; };
  auto TR = [&](int id, int k) { return in.trans[(size_t)id * Kp + k]; };
  float const *xt = in.xt;

  int state = ST_T, stage = in.L;
  int64_t n = 0;
  if (st && st->state != 0)
  {
    state = st->state;
    stage = st->stage;
    n = st->n;
  }
  while (state != ST_S || stage)
  {
    if (stage <= in.lo)
    {
      st->state = state;
      st->stage = stage;
      st->n = n;
      return 0;
    }
    int size = 0, prev = -1;
    if ((state & ST_X) == ST_X)
    {
      if (state == ST_T)
      {
        float const a = SP(stage, 3) + xt[DCP_ET], b = SP(stage, 4) + xt[DCP_CT];
        if (a == b) return a < INF ? DCP_TB_TIE : DCP_TB_BAD;
        prev = a < b ? ST_E : ST_C;
      }
      else if (state == ST_N || state == ST_J || state == ST_C)
      {
        int const self = state == ST_N ? 0 : state == ST_J ? 2 : 4;
        float const target = SP(stage, self);
        float const t_in = state == ST_N ? xt[DCP_SN] : state == ST_J ? xt[DCP_EJ] : xt[DCP_EC];
        float const t_self = state == ST_N ? xt[DCP_NN] : state == ST_J ? xt[DCP_JJ] : xt[DCP_CC];
        for (int t = stage < 5 ? stage : 5; t >= 1 && prev < 0; --t)
        {
          int const z = stage - t;
          float const nil = ROW(stage, t)[0];
          float const from = state == ST_N ? (z == 0 ? 0.0f : INF) : SP(z, 3); // S of row z, or E of row z
          if ((from + t_in) + nil == target) { prev = state == ST_N ? ST_S : ST_E; size = t; }
          else if ((SP(z, self) + t_self) + nil == target) { prev = state; size = t; }
        }
        if (prev < 0 || !(target < INF)) return DCP_TB_BAD;
      }
      else if (state == ST_B)
      {
        if (stage == 0) prev = ST_S; // row 0: B = S + SB (c-core/viterbi.c:473)
        else
        {
          float const target = SP(stage, 1);
          int const eN = SP(stage, 0) + xt[DCP_NB] == target, eE = SP(stage, 3) + xt[DCP_EB] == target,
                    eJ = SP(stage, 2) + xt[DCP_JB] == target;
          if (eN + eE + eJ != 1 || !(target < INF)) return eN + eE + eJ > 1 ? DCP_TB_TIE : DCP_TB_BAD;
          prev = eN ? ST_N : eE ? ST_E : ST_J;
        }
      }
      else if (state == ST_E)
      {
        // E = min over k of M (c-core/viterbi.c:540-558).  Several M equal to it: the
        // reference's lanes (k = e*Qr + q) keep the first q per lane and the highest lane.
        float const target = SP(stage, 3);
        if (!(target < INF)) return DCP_TB_BAD;
        int Qr = (K - 1) / DCP_REF_LANES + 1;
        if (Qr < 2) Qr = 2;
        int best = -1;
        for (int k = 0; k < K; ++k)
        {
          if (CELL(stage, 2, k) == target) return DCP_TB_TIE; // a D candidate at the minimum: pass history decides
          if (CELL(stage, 0, k) == target && (best < 0 || k / Qr > best / Qr)) best = k;
        }
        if (best < 0) return DCP_TB_BAD;
        prev = ST_M | (best + 1);
      }
      else
        return DCP_TB_BAD;
    }
    else
    {
      int const k = (state & 0x3FFF) - 1;
      int const kind = state & ST_X;
      if (k < 0 || k >= K) return DCP_TB_BAD;
      if (kind == ST_M)
      {
        float const target = CELL(stage, 0, k);
        if (!(target < INF)) return DCP_TB_BAD;
        float const BM = TR(DCP_BM, k), MM = TR(DCP_MM, k), IM = TR(DCP_IM, k), DM = TR(DCP_DM, k);
        for (int t = stage < 5 ? stage : 5; t >= 1 && prev < 0; --t)
        {
          int const z = stage - t;
          float const m = ROW(stage, t)[DCP_ROW_HDR + k];
          if ((SP(z, 1) + BM) + m == target) prev = ST_B;
          else if ((CELL(z, 0, k - 1) + MM) + m == target) prev = ST_M | k;
          else if ((CELL(z, 1, k - 1) + IM) + m == target) prev = ST_I | k;
          else if ((CELL(z, 2, k - 1) + DM) + m == target) prev = ST_D | k;
          if (prev >= 0) size = t;
        }
        if (prev < 0) return DCP_TB_BAD;
      }
      else if (kind == ST_I)
      {
        float const target = CELL(stage, 1, k);
        if (!(target < INF)) return DCP_TB_BAD;
        float const II = TR(DCP_II, k), MI = TR(DCP_MI, k);
        for (int t = stage < 5 ? stage : 5; t >= 1 && prev < 0; --t)
        {
          int const z = stage - t;
          float const bg = ROW(stage, t)[1];
          if ((CELL(z, 1, k) + II) + bg == target) prev = ST_I | (k + 1);
          else if ((CELL(z, 0, k) + MI) + bg == target) prev = ST_M | (k + 1);
          if (prev >= 0) size = t;
        }
        if (prev < 0) return DCP_TB_BAD;
      }
      else
      {
        float const a = CELL(stage, 0, k - 1) + TR(DCP_MD, k), b = CELL(stage, 2, k - 1) + TR(DCP_DD, k);
        if (a == b) return a < INF ? DCP_TB_TIE : DCP_TB_BAD;
        prev = (a < b ? ST_M : ST_D) | k;
      }
    }
    if (n + 1 >= cap) return DCP_TB_OVERFLOW;
    buf[cap - 1 - n] = (uint32_t)state | ((uint32_t)size << 16);
    ++n;
    state = prev;
    stage -= size;
    if (stage < 0) return DCP_TB_BAD;
  }
  if (n >= cap) return DCP_TB_OVERFLOW;
  buf[cap - 1 - n] = (uint32_t)state;
  return (int)(n + 1);
}
