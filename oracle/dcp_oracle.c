/* dcp_oracle.c -- TEST INFRASTRUCTURE ONLY (see dcp_oracle.h).
 *
 * Plain scalar C restatement of the reference algorithm, written position by
 * position (k = 0..K-1) instead of in the reference's striped SIMD packs.
 * Every function cites the reference lines it follows.  The only place the
 * reference's SIMD width leaks into results is the cross-lane tie rule for the
 * E back-pointer; it is reproduced in e_state() for `ref_lanes` lanes.
 */
#include "dcp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INF INFINITY

/* ---- codes: imm_eseq_get(seq,pos,len,min_seq=1), SURVEY 8a row S ---------- */
int orc_code(uint8_t const *seq, int pos, int len)
{
  static int const off[6] = {0, 0, 4, 20, 84, 340};
  int idx = 0;
  for (int i = 0; i < len; ++i) idx = idx * 4 + seq[pos + i];
  return off[len] + idx;
}

/* ---- c-core/xtrans.c:21-51 then :53-68 ------------------------------------ */
void orc_xtrans(int seq_size, int multi_hits, int hmmer3_compat, float xt[ORC_NUM_XTRANS])
{
  float L = (float)seq_size;
  float q = 0.0f;
  float log_q = -INFINITY; /* IMM_LPROB_ZERO */
  if (multi_hits)
  {
    q = 0.5f;
    log_q = (float)log(0.5);
  }
  /* double log() on float operands, result narrowed to float (xtrans.c:38-40) */
  float lp = (float)(log(L) - log(L + 2 + q / (1 - q)));
  float l1p = (float)(log(2 + q / (1 - q)) - log(L + 2 + q / (1 - q)));
  float lr = (float)(log(L) - log(L + 1));

  float NN = lp, CC = lp, JJ = lp;
  float NB = l1p, CT = l1p, JB = l1p;
  float RR = lr;
  float EJ = log_q;
  float EC = (float)log(1 - q);
  if (hmmer3_compat) NN = CC = JJ = logf(1);

  xt[ORC_RR] = -RR;
  xt[ORC_SN] = -0 - NN;
  xt[ORC_NN] = -NN;
  xt[ORC_SB] = -0 - NB;
  xt[ORC_NB] = -NB;
  xt[ORC_EB] = -EJ - JB;
  xt[ORC_JB] = -JB;
  xt[ORC_EJ] = -EJ - JJ;
  xt[ORC_JJ] = -JJ;
  xt[ORC_EC] = -EC - CC;
  xt[ORC_CC] = -CC;
  xt[ORC_ET] = -EC - CT;
  xt[ORC_CT] = -CT;
}

/* ---- c-core/protein.c:353-394 --------------------------------------------- */
void orc_setup_profile(int K, float const *node_trans, float const *node_emission,
                       float const *BMk, float const *null_lprob, float const *bg_lprob,
                       float *trans, float *match, float *null_cost, float *bg_cost)
{
  for (int i = 0; i < ORC_NUM_TRANS * K; ++i) trans[i] = INF; /* viterbi.c:235-245 */
  for (int k = 0; k < K; ++k) trans[ORC_BM * K + k] = -BMk[k];
  for (int k = 0; k + 1 < K; ++k)
  {
    float const *t = node_trans + 7 * k; /* trans.h:8-27: MM,MI,MD,IM,II,DM,DD */
    trans[ORC_MM * K + k + 1] = -t[0];
    trans[ORC_MI * K + k + 0] = -t[1];
    trans[ORC_MD * K + k + 1] = -t[2];
    trans[ORC_IM * K + k + 1] = -t[3];
    trans[ORC_II * K + k + 0] = -t[4];
    trans[ORC_DM * K + k + 1] = -t[5];
    trans[ORC_DD * K + k + 1] = -t[6];
  }
  trans[ORC_MI * K + K - 1] = INF;
  trans[ORC_II * K + K - 1] = INF;
  for (int c = 0; c < ORC_TABLE_SIZE; ++c)
  {
    null_cost[c] = -null_lprob[c];
    bg_cost[c] = -bg_lprob[c];
    for (int k = 0; k < K; ++k)
      match[(size_t)c * K + k] = -node_emission[(size_t)k * ORC_TABLE_SIZE + c];
  }
}

static inline int imin(int a, int b) { return a < b ? a : b; }

/* ---- c-core/viterbi.c:696-719 --------------------------------------------- */
float orc_null(float const *null_cost, float RR, uint8_t const *seq, int L)
{
  float R[6];
  for (int i = 0; i < 6; ++i) R[i] = INF;
  R[0] = -RR;
  for (int l = 1; l <= L; ++l)
  {
    R[imin(5, l)] = INF;
    for (int t = imin(5, l); t > 0; --t)
    {
      float nil = null_cost[orc_code(seq, l - t, t)];
      float tmp = fminf(R[t], R[t - 1] + RR + nil);
      R[t] = R[t - 1];
      R[t - 1] = tmp;
    }
  }
  return R[0];
}

/* strict-< update that remembers who improved: acc()/facc() with save=1
 * (viterbi.c:201-212, intrinsics.h:144-149).  Ties keep the earlier pointer. */
#define UPD(cur, ptr, val, newptr)                                             \
  do                                                                           \
  {                                                                            \
    float v_ = (val);                                                          \
    float x_ = fminf((cur), v_);                                               \
    if (!(x_ == (cur))) (ptr) = (newptr);                                      \
    (cur) = x_;                                                                \
  } while (0)

struct specials { float S, N, B, J, E, C, T; };

static void specials_init(struct specials *x)
{
  x->S = x->N = x->B = x->J = x->E = x->C = x->T = INF; /* viterbi.c:259-270 */
}

/* The reference reduces E across SIMD lanes: every lane e of `lanes` owns
 * positions k = e*Q + q (viterbi.c:220-221), keeps the first candidate that
 * attains its own minimum in the order ME(0),DE(0)*,ME(1),DE(1),...,DE(0)
 * (viterbi.c:540-541,555-556; DE(0)* uses the stale M of the previous lane,
 * viterbi.c:507,538), and the lanes are merged by taking the maximum packed
 * (name<<28 | lane<<24 | q) among lanes equal to the minimum
 * (viterbi.c:557-558, intrinsics.h:151-160).  Returns the trellis E field. */
static unsigned e_state(int K, int lanes, float const *Ma, float const *Da,
                        float const *Mbefore, float const *Dbefore, float const *MD,
                        float *Eout)
{
  int Q = (K - 1) / lanes + 1;
  if (Q < 2) Q = 2; /* viterbi.c:195-199 */
  float best = INF;
  uint32_t bestptr = 0;
  int any = 0;
  float lane_val[64];
  uint32_t lane_ptr[64];
  for (int e = 0; e < lanes; ++e)
  {
    float v = INF;
    uint32_t p = 0;
    for (int q = 0; q < Q; ++q)
    {
      int k = e * Q + q;
      float m = k < K ? Ma[k] : INF;
      float d = INF;
      if (k < K)
      {
        if (q == 0)
        {
          float lastMa = k > 0 ? Mbefore[k - 1] : INF;
          d = fminf(Dbefore[k], lastMa + MD[k]);
        }
        else
          d = Da[k];
      }
      UPD(v, p, m, (0x1u << 28) | (uint32_t)q);
      UPD(v, p, d, (0x2u << 28) | (uint32_t)q);
    }
    {
      int k = e * Q;
      float d = k < K ? Da[k] : INF;
      UPD(v, p, d, (0x2u << 28) | 0u);
    }
    lane_val[e] = v;
    lane_ptr[e] = p | ((uint32_t)e << 24);
    best = fminf(best, v);
  }
  for (int e = 0; e < lanes; ++e)
  {
    if (lane_val[e] == best)
    {
      if (!any || lane_ptr[e] > bestptr) bestptr = lane_ptr[e];
      any = 1;
    }
  }
  *Eout = best;
  int q = (int)(bestptr & 0x00FFFFFFu);
  int e = (int)((bestptr >> 24) & 0xF);
  int k = e * Q + q;
  if (bestptr & (0x1u << 28)) return (unsigned)(2 * k + 0); /* viterbi.c:676-680 */
  if (bestptr & (0x2u << 28)) return (unsigned)(2 * k + 1);
  return 0;
}

/* ---- c-core/viterbi.c:451-600, :602-694 ----------------------------------- */
float orc_cost(int K, float const *trans, float const *match, float const *null_cost,
               float const *bg_cost, float const xt[ORC_NUM_XTRANS], uint8_t const *seq,
               int L, int ref_lanes, uint32_t *xnodes, uint16_t *nodes)
{
  float const *BM = trans + ORC_BM * K, *MM = trans + ORC_MM * K;
  float const *MI = trans + ORC_MI * K, *MD = trans + ORC_MD * K;
  float const *IM = trans + ORC_IM * K, *II = trans + ORC_II * K;
  float const *DM = trans + ORC_DM * K, *DD = trans + ORC_DD * K;
  int const path = xnodes && nodes;

  /* ring of 6 rows (TIME_FRAME, viterbi.c:12); row l lives in slot l % 6 */
  struct specials xs[6];
  float *M = malloc(sizeof(float) * 6 * K * 3);
  float *I = M + 6 * K, *D = I + 6 * K;
  float *Mbefore = malloc(sizeof(float) * 2 * K);
  float *Dbefore = Mbefore + K;
  uint8_t *pM = malloc((size_t)3 * K);
  uint8_t *pI = pM + K, *pD = pI + K;
  for (int s = 0; s < 6; ++s) specials_init(&xs[s]);
  for (int i = 0; i < 6 * K * 3; ++i) M[i] = INF;

  xs[0].S = 0; /* viterbi.c:471-473 */
  xs[0].B = xt[ORC_SB];
  if (path)
  {
    xnodes[0] = 0; /* before(): every field 0, viterbi.c:602-629 */
    memset(nodes, 0, sizeof(uint16_t) * K);
  }

  for (int l = 1; l <= L; ++l)
  {
    struct specials *a = &xs[l % 6];
    float *Ma = M + (l % 6) * K, *Ia = I + (l % 6) * K, *Da = D + (l % 6) * K;
    specials_init(a);
    for (int k = 0; k < K; ++k) Ma[k] = Ia[k] = Da[k] = INF;
    /* prev_*_state_init, viterbi.c:288-306, held directly as trellis fields */
    unsigned pN = 0, pB = 0, pJ = 0, pE = 0, pC = 0, pT = 0;
    for (int k = 0; k < K; ++k) pM[k] = pI[k] = pD[k] = 0;

    for (int t = imin(5, l); t > 0; --t)
    {
      struct specials const *z = &xs[(l - t) % 6];
      float const *Mz = M + ((l - t) % 6) * K;
      float const *Iz = I + ((l - t) % 6) * K;
      float const *Dz = D + ((l - t) % 6) * K;
      int code = orc_code(seq, l - t, t);
      float nil = null_cost[code];
      float bg = bg_cost[code];
      float const *ma = match + (size_t)code * K;
      unsigned u = (unsigned)(t - 1);

      UPD(a->N, pN, z->S + xt[ORC_SN] + nil, 0 + u); /* viterbi.c:492-493 */
      UPD(a->N, pN, z->N + xt[ORC_NN] + nil, 5 + u);
      UPD(a->B, pB, a->S + xt[ORC_SB], 0); /* :495-496 */
      UPD(a->B, pB, a->N + xt[ORC_NB], 1);
      UPD(a->J, pJ, z->E + xt[ORC_EJ] + nil, 0 + u); /* :498-499 */
      UPD(a->J, pJ, z->J + xt[ORC_JJ] + nil, 5 + u);
      UPD(a->C, pC, z->E + xt[ORC_EC] + nil, 0 + u); /* :501-502 */
      UPD(a->C, pC, z->C + xt[ORC_CC] + nil, 5 + u);

      memcpy(Mbefore, Ma, sizeof(float) * K);
      memcpy(Dbefore, Da, sizeof(float) * K);

      for (int k = 0; k < K; ++k) /* :512-551, one position at a time */
      {
        float lastMz = k > 0 ? Mz[k - 1] : INF; /* shift(): +inf enters at k=0 */
        float lastIz = k > 0 ? Iz[k - 1] : INF;
        float lastDz = k > 0 ? Dz[k - 1] : INF;
        UPD(Ma[k], pM[k], (z->B + BM[k]) + ma[k], 0 + u);
        UPD(Ma[k], pM[k], (lastMz + MM[k]) + ma[k], 5 + u);
        UPD(Ma[k], pM[k], (lastIz + IM[k]) + ma[k], 10 + u);
        UPD(Ma[k], pM[k], (lastDz + DM[k]) + ma[k], 15 + u);
        UPD(Ia[k], pI[k], (Iz[k] + II[k]) + bg, 5 + u);
        UPD(Ia[k], pI[k], (Mz[k] + MI[k]) + bg, 0 + u);
      }
      for (int k = 0; k < K; ++k) /* :538 and the stripe-0 repair :553-555 */
      {
        float lastMa = k > 0 ? Ma[k - 1] : INF;
        UPD(Da[k], pD[k], lastMa + MD[k], 0);
      }

      pE = e_state(K, ref_lanes, Ma, Da, Mbefore, Dbefore, MD, &a->E); /* :540-541,556-558 */

      for (int k = 1; k < K; ++k) /* :561-580: lazy D->D == one serial chain */
        UPD(Da[k], pD[k], Da[k - 1] + DD[k], 1);

      UPD(a->B, pB, a->E + xt[ORC_EB], 2); /* :582-583 */
      UPD(a->B, pB, a->J + xt[ORC_JB], 3);
      UPD(a->T, pT, a->E + xt[ORC_ET], 0); /* :585-586 */
      UPD(a->T, pT, a->C + xt[ORC_CT], 1);
    }

    if (path) /* after(), viterbi.c:631-694; shifts trellis.h:42-56 */
    {
      xnodes[l] = (pN << 0) | (pB << 4) | (pE << 6) | (pC << 21) | (pT << 25) | (pJ << 26);
      uint16_t *row = nodes + (size_t)l * K;
      for (int k = 0; k < K; ++k)
      {
        unsigned w = pM[k];
        if (k > 0) w |= (unsigned)pD[k] << 5;
        if (k + 1 < K) w |= (unsigned)pI[k] << 6;
        row[k] = (uint16_t)w;
      }
    }
  }

  float r = xs[L % 6].T; /* viterbi.c:599 (for L = 0: the +inf initial value) */
  free(M);
  free(Mbefore);
  free(pM);
  return r;
}

/* ---- c-core/state.h:9-25, state.c:25,92-96 -------------------------------- */
enum
{
  ST_M = 0 << 14, ST_I = 1 << 14, ST_D = 2 << 14, ST_X = 3 << 14,
  ST_S = ST_X | 3, ST_N = ST_X | 4, ST_B = ST_X | 5, ST_E = ST_X | 6,
  ST_J = ST_X | 7, ST_C = ST_X | 8, ST_T = ST_X | 9,
};
static int st_msb(int id) { return id & (3 << 14); }
static int st_is_core(int id) { return st_msb(id) != ST_X; }
static int st_core_idx(int id) { return (id & 0x3FFF) - 1; }

static unsigned xfield(uint32_t x, int state) /* trellis.c:125-135 */
{
  switch (state)
  {
  case ST_N: return (x >> 0) & 0xF;
  case ST_B: return (x >> 4) & 0x3;
  case ST_E: return (x >> 6) & 0x7FFF;
  case ST_C: return (x >> 21) & 0xF;
  case ST_T: return (x >> 25) & 0x1;
  case ST_J: return (x >> 26) & 0xF;
  default: return 0;
  }
}

static unsigned nfield(uint16_t x, int state) /* trellis.c:137-145 */
{
  if (st_msb(state) == ST_M) return x & 0x1F;
  if (st_msb(state) == ST_D) return (x >> 5) & 0x1;
  return (x >> 6) & 0xF;
}

/* ---- c-core/trellis.c:147-167 with :51-113 -------------------------------- */
int orc_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, int *state_ids,
              int *seqsizes, int cap)
{
  int n = 0;
  int state = ST_T;
  int stage = L;
  while (state != ST_S || stage)
  {
    uint32_t xw = xnodes[stage];
    uint16_t nw = st_is_core(state) ? nodes[(size_t)stage * K + st_core_idx(state)] : 0;
    int idx = st_is_core(state) ? st_core_idx(state) : 0;
    int size = 0, prev = 0;

    if (!st_is_core(state))
    {
      unsigned v = xfield(xw, state);
      if (state == ST_N || state == ST_C || state == ST_J) size = (int)(v % 5) + 1;
      if (state == ST_S) prev = ST_S;
      else if (state == ST_N) prev = v / 5 ? ST_N : ST_S;
      else if (state == ST_B) prev = (int[]){ST_S, ST_N, ST_E, ST_J}[v];
      else if (state == ST_E) prev = (v % 2 ? ST_D : ST_M) | (int)(v / 2 + 1);
      else if (state == ST_C) prev = v / 5 ? ST_C : ST_E;
      else if (state == ST_T) prev = v ? ST_C : ST_E;
      else if (state == ST_J) prev = v / 5 ? ST_J : ST_E;
    }
    else if (st_msb(state) == ST_M)
    {
      unsigned v = nfield(nw, state);
      size = (int)(v % 5) + 1;
      int s = (int)(v / 5);
      if (s == 0) prev = ST_B;
      else if (idx <= 0) return -2; /* BUG_ON(idx <= 0), trellis.c:72 */
      else prev = (int[]){0, ST_M, ST_I, ST_D}[s] | idx;
    }
    else if (st_msb(state) == ST_D)
    {
      unsigned v = nfield(nw, state);
      if (idx <= 0) return -2;
      prev = (v ? ST_D : ST_M) | idx;
    }
    else
    {
      unsigned v = nfield(nw, state);
      size = (int)(v % 5) + 1;
      prev = (v / 5 ? ST_I : ST_M) | (idx + 1);
    }

    if (n >= cap) return -1;
    state_ids[n] = state;
    seqsizes[n] = size;
    ++n;
    state = prev;
    stage -= size;
    if (stage < 0) return -2;
  }
  if (n >= cap) return -1;
  state_ids[n] = state;
  seqsizes[n] = 0;
  ++n;
  for (int i = 0, j = n - 1; i < j; ++i, --j) /* imm_path_reverse */
  {
    int s = state_ids[i]; state_ids[i] = state_ids[j]; state_ids[j] = s;
    int z = seqsizes[i]; seqsizes[i] = seqsizes[j]; seqsizes[j] = z;
  }
  return n;
}

float orc_lrt(float null_loglik, float alt_loglik) { return -2 * (null_loglik - alt_loglik); }

/* ---- c-core/window.c:7-37 -------------------------------------------------- */
struct orc_window orc_window_setup(int seq_size, int core_size)
{
  return (struct orc_window){core_size, seq_size, -1, 0, -1, -1};
}

int orc_window_next(struct orc_window *x)
{
  if (x->stop == x->seq_size) return 0;
  int stop_miss = x->stop + 1;
  int start_miss = x->start + 1 > x->start + x->last_hit_pos + 1 ? x->start + 1
                                                                 : x->start + x->last_hit_pos + 1;
  if (stop_miss - x->core_size * 4 > start_miss) start_miss = stop_miss - x->core_size * 4;
  x->start = start_miss;
  x->stop = start_miss + imin(x->core_size * 50, 100000);
  x->stop = imin(x->stop, x->seq_size);
  x->idx += 1;
  return 1;
}

/* ---- c-core/thread.c:130-166 ----------------------------------------------- */
int orc_hits(int const *state_ids, int const *seqsizes, int nsteps, int *hits, int cap,
             int *last_hit_pos)
{
  int it = 0, hit_start = 0;
  while (it < nsteps && state_ids[it] != ST_B)
  {
    hit_start += seqsizes[it];
    ++it;
  }
  if (it >= nsteps) return 0;
  int hit_stop_run = hit_start;
  int begin = it;
  int end = it + 1;
  int line_hit_stop = 0;
  for (;;)
  {
    it = end;
    line_hit_stop = hit_stop_run;
    while (it < nsteps && state_ids[it] != ST_E)
    {
      hit_stop_run += seqsizes[it];
      ++it;
    }
    if (it >= nsteps)
    {
      *last_hit_pos = line_hit_stop - 1;
      break;
    }
    end = it + 1;
  }
  if (cap < 1) return 0;
  hits[0] = hit_start;
  hits[1] = line_hit_stop;
  hits[2] = begin;
  hits[3] = end;
  return 1;
}

/* ---- c-core/sequence.c:15-45, uppercase.c, disambiguate.c:37-86 ------------ */
int orc_encode(char const *data, int n, uint8_t *out)
{
  enum { A, C, G, T, U };
  size_t count[5] = {0};
  char *s = malloc((size_t)n + 1);
  for (int i = 0; i < n; ++i)
  {
    char c = data[i];
    if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
    s[i] = c;
    if (c == 'A') count[A]++;
    if (c == 'C') count[C]++;
    if (c == 'G') count[G]++;
    if (c == 'T') count[T]++;
    if (c == 'U') count[U]++;
  }
  if (count[T] > 0 && count[U] > 0)
  {
    free(s);
    return 74; /* DCP_ENUCLTSEQTU */
  }
  static char const *const amb = "RYMKSWHBVDNX";
  static int const sets[12][4] = {
      {A, G, -1, -1}, {C, T, -1, -1}, {A, C, -1, -1}, {G, T, -1, -1},
      {C, G, -1, -1}, {A, T, -1, -1}, {A, C, T, -1}, {C, G, T, -1},
      {A, C, G, -1},  {A, G, T, -1},  {A, C, G, T},  {A, C, G, T}};
  for (int i = 0; i < n; ++i)
  {
    char const *p = s[i] ? strchr(amb, s[i]) : NULL;
    if (!p) continue;
    int const *set = sets[p - amb];
    int best = set[0];
    for (int j = 1; j < 4 && set[j] >= 0; ++j)
      if (count[set[j]] > count[best]) best = set[j];
    s[i] = "ACGTU"[best];
  }
  int rc = 0;
  for (int i = 0; i < n; ++i)
  {
    switch (s[i])
    {
    case 'A': out[i] = 0; break;
    case 'C': out[i] = 1; break;
    case 'G': out[i] = 2; break;
    case 'T': out[i] = 3; break;
    case 'U': out[i] = 3; break;
    default: rc = 57; break; /* DCP_ESEQABC */
    }
  }
  free(s);
  return rc;
}

void orc_state_name(int id, char *name)
{
  if (st_msb(id) == ST_X)
  {
    static char const names[] = "FRGSNBEJCT";
    int n = id & 0x3FFF;
    name[0] = n < 10 ? names[n] : '?';
    name[1] = 0;
    return;
  }
  name[0] = st_msb(id) == ST_M ? 'M' : st_msb(id) == ST_I ? 'I' : 'D';
  int v = st_core_idx(id) + 1, len = 0;
  char tmp[8];
  do { tmp[len++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int i = 0; i < len; ++i) name[1 + i] = tmp[len - 1 - i];
  name[1 + len] = 0;
}

long orc_partition_size(long nelems, long nparts, long idx)
{
  long x = nelems - idx;
  if (x < 0) x = 0;
  return (x + nparts - 1) / nparts;
}
