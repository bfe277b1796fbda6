// lane_ops_emul.h -- TEST INFRASTRUCTURE ONLY.
// A lock-step 64-lane emulation of the vocabulary in
// deciphon_amd/csrc/lane_ops_gpu.h, so that deciphon_amd/csrc/viterbi_body.h
// (the kernel logic itself, unchanged) can be unit-tested on a machine without a
// GPU.  It is never part of the product library.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../deciphon_amd/csrc/dcp_types.h"

#define DCP_FN inline
#define DCP_WAVE 64


// A "group" of W wavefronts is emulated as ONE vector of 64*W lanes: values that the GPU
// moves across wave boundaries through LDS simply shift / reduce across the whole vector.
#define EM_MAX_LANES 1024
static thread_local int em_lanes = 64;

struct lf { float v[EM_MAX_LANES]; };
struct lu { uint32_t v[EM_MAX_LANES]; };
struct lm { bool v[EM_MAX_LANES]; };

#define EM_FOR for (int i_ = 0; i_ < em_lanes; ++i_)

inline lf lf_splat(float x) { lf r; EM_FOR r.v[i_] = x; return r; }
inline lu lu_splat(uint32_t x) { lu r; EM_FOR r.v[i_] = x; return r; }
inline lf operator+(lf a, lf b) { lf r; EM_FOR r.v[i_] = a.v[i_] + b.v[i_]; return r; }
inline lf operator+(lf a, float b) { lf r; EM_FOR r.v[i_] = a.v[i_] + b; return r; }
inline lf operator+(float a, lf b) { lf r; EM_FOR r.v[i_] = a + b.v[i_]; return r; }
inline lf lmin(lf a, lf b) { lf r; EM_FOR r.v[i_] = fminf(a.v[i_], b.v[i_]); return r; }
inline lf lmin3(lf a, lf b, lf c) { return lmin(lmin(a, b), c); }
inline lm llt(lf a, lf b) { lm r; EM_FOR r.v[i_] = a.v[i_] < b.v[i_]; return r; }
inline lm llt_u(lu a, lu b) { lm r; EM_FOR r.v[i_] = a.v[i_] < b.v[i_]; return r; }
inline lm leq(lf a, lf b) { lm r; EM_FOR r.v[i_] = a.v[i_] == b.v[i_]; return r; }
inline lm lequ(lu a, lu b) { lm r; EM_FOR r.v[i_] = a.v[i_] == b.v[i_]; return r; }
inline lm land(lm a, lm b) { lm r; EM_FOR r.v[i_] = a.v[i_] && b.v[i_]; return r; }
inline lm lor(lm a, lm b) { lm r; EM_FOR r.v[i_] = a.v[i_] || b.v[i_]; return r; }
inline lm lnot(lm a) { lm r; EM_FOR r.v[i_] = !a.v[i_]; return r; }
inline lf lsel(lm m, lf a, lf b) { lf r; EM_FOR r.v[i_] = m.v[i_] ? a.v[i_] : b.v[i_]; return r; }
inline lu lselu(lm m, lu a, lu b) { lu r; EM_FOR r.v[i_] = m.v[i_] ? a.v[i_] : b.v[i_]; return r; }
inline lu lminu(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] < b.v[i_] ? a.v[i_] : b.v[i_]; return r; }
inline lu lmaxu(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] > b.v[i_] ? a.v[i_] : b.v[i_]; return r; }
inline lu operator+(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] + b.v[i_]; return r; }
inline lu operator+(lu a, uint32_t b) { lu r; EM_FOR r.v[i_] = a.v[i_] + b; return r; }
inline lu operator-(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] - b.v[i_]; return r; }
inline lu operator*(lu a, uint32_t b) { lu r; EM_FOR r.v[i_] = a.v[i_] * b; return r; }
inline lu operator<<(lu a, int b) { lu r; EM_FOR r.v[i_] = a.v[i_] << b; return r; }
inline lu operator|(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] | b.v[i_]; return r; }

inline void sched_fence() {}
inline lu lane_ids() { lu r; EM_FOR r.v[i_] = (uint32_t)i_; return r; }

inline lf lane_shift_up(lf x, float fill)
{
  lf r;
  r.v[0] = fill;
  for (int i = 1; i < em_lanes; ++i) r.v[i] = x.v[i - 1];
  return r;
}

inline lf lane_shift_up_keep(lf x, lf &keep)
{
  keep = lane_shift_up(x, keep.v[0]);
  return keep;
}
inline lf lf_pin(float x) { return lf_splat(x); }

inline float wave_min(lf v) { float m = v.v[0]; EM_FOR m = fminf(m, v.v[i_]); return m; }
inline uint32_t wave_minu(lu v) { uint32_t m = v.v[0]; EM_FOR m = v.v[i_] < m ? v.v[i_] : m; return m; }
// votes taken / votes that carried since the last emul_votes() call (diagnostic: extra lazy D->D turns per row)
extern thread_local long em_votes, em_votes_true;
inline bool wave_any(lm m)
{
  ++em_votes;
  EM_FOR if (m.v[i_]) { ++em_votes_true; return true; }
  return false;
}
template <int P> inline void wave_priority() {}
inline uint64_t wave_ballot(lm m) { uint64_t b = 0; EM_FOR if (m.v[i_]) b |= 1ull << i_; return b; }
inline float read_lane(lf x, int lane) { return x.v[lane]; }
inline uint32_t read_laneu(lu x, int lane) { return x.v[lane]; }

enum { GS_M, GS_I, GS_D, GS_E, GS_F, GS_X, GS_T0, GS_T1, GS_X0, GS_X1, GS_SLOTS };

// rows of the cost pass that took the exchange-until-stable fallback (tests read and reset it)
template <int Q> struct DcpStashChunk { static constexpr int N = Q % 4 == 0 ? 4 : Q % 2 == 0 ? 2 : 1; }; // as lane_ops_gpu.h
extern thread_local long em_fallback_rows; // defined in emul.cpp: one counter for every translation unit

template <int W> struct Group
{
  lu lane;
  void init()
  {
    em_lanes = 64 * W;
    lane = lane_ids();
    put_carry_inf(0);
    put_carry_inf(1);
  }
  // ---- one-barrier rows: here the W waves are real 64-lane segments of the vector ----
  struct Rec { float m, i, d, e; };
  Rec rec[2][16];
  float tdd[16];
  lf seg_shift_up(lf x, lf fill)
  {
    lf r;
    for (int i = 0; i < em_lanes; ++i) r.v[i] = (i % 64 == 0) ? fill.v[i] : x.v[i - 1];
    return r;
  }
  bool seg_any(lm m) { return wave_any(m); } // extra turns in a converged wave change nothing
  lm seg_first() { lm r; EM_FOR r.v[i_] = (i_ % 64) == 0; return r; }
  template <int Q> void put_tdd(lf const (&DD)[Q])
  {
    for (int w = 0; w < W; ++w)
    {
      float t = 0.0f;
      for (int i = 64 * w; i < 64 * w + 64; ++i)
        for (int q = 0; q < Q; ++q)
          if (!(i == 64 * w && q == 0)) t += DD[q].v[i];
      tdd[w] = t;
    }
  }
  void put_rec(int par, lf m_last, lf i_last, lf d_last, lf m_all)
  {
    for (int w = 0; w < W; ++w)
    {
      float e = m_all.v[64 * w];
      for (int i = 64 * w; i < 64 * w + 64; ++i) e = fminf(e, m_all.v[i]);
      rec[par][w] = Rec{m_last.v[64 * w + 63], i_last.v[64 * w + 63], d_last.v[64 * w + 63], e};
    }
  }
  void get_prev(int par, lf &Mp, lf &Ip, lf &Dp)
  {
    for (int i = 0; i < em_lanes; ++i)
    {
      int const w = i / 64;
      Rec const &p = w > 0 ? rec[par][w - 1] : carry[par];
      Mp.v[i] = p.m;
      Ip.v[i] = p.i;
      Dp.v[i] = p.d;
    }
  }
  void get_e_could(int par, float &E, bool &could)
  {
    float m = rec[par][0].e;
    for (int w = 1; w < W; ++w) m = fminf(m, rec[par][w].e);
    bool any = false;
    for (int w = 0; w < W; ++w)
    {
      float const s = m + tdd[w];
      float const bound = fminf(s * 0.9999f, s * 1.0001f);
      any = any || bound < rec[par][w].d;
    }
    E = m;
    could = any;
  }
  void get_nj(int, lf X, float &N, float &J)
  {
    N = X.v[0];
    J = X.v[1];
  }
  // values parked between uses (LDS on the GPU)
  lf stashv[8][16];
  template <int Q, int SLOTS> void stash_q(int slot, lf const (&v)[Q]) { for (int q = 0; q < Q; ++q) stashv[slot][q] = v[q]; }
  template <int Q, int SLOTS> void unstash_q(int slot, lf (&v)[Q]) { for (int q = 0; q < Q; ++q) v[q] = stashv[slot][q]; }
  template <int Q, int SLOTS> void unstash_chunk(int slot, int j, lf (&v)[DcpStashChunk<Q>::N])
  {
    for (int i = 0; i < DcpStashChunk<Q>::N; ++i) v[i] = stashv[slot][DcpStashChunk<Q>::N * j + i];
  }
  // ---- StripWave ----
  Rec carry[2];
  float tdds[64][16];
  template <int Q> void put_tdd_strip(int s, lf const (&DD)[Q])
  {
    for (int w = 0; w < W; ++w)
    {
      float t = 0.0f;
      for (int i = 64 * w; i < 64 * w + 64; ++i)
        for (int q = 0; q < Q; ++q)
          if (!(i == 64 * w && q == 0)) t += DD[q].v[i];
      tdds[s][w] = t;
    }
  }
  void put_carry(int par, lf m, lf i, lf d) { carry[par] = Rec{m.v[em_lanes - 1], i.v[em_lanes - 1], d.v[em_lanes - 1], INFINITY}; }
  void put_carry_inf(int par) { carry[par] = Rec{INFINITY, INFINITY, INFINITY, INFINITY}; }
  void get_e_could_row(int par, int s, float floor_e, float &E, bool &could)
  {
    float m = rec[par][0].e;
    for (int w = 1; w < W; ++w) m = fminf(m, rec[par][w].e);
    float const lo = fminf(m, floor_e);
    bool any = false;
    for (int w = 0; w < W; ++w)
    {
      float const v = lo + tdds[s][w];
      any = any || fminf(v * 0.9999f, v * 1.0001f) < rec[par][w].d;
    }
    E = m;
    could = any;
  }
  lf get_shift_carry(int, lf x, lf fill) { return lane_shift_up(x, fill.v[0]); }
  void note_fallback() { ++em_fallback_rows; }

  void put_last(int, lf) {}
  void put_min(int, lf) {}
  void put_minu(int, lu) {}
  void put_lanes4(int, lf) {}
  void put_any(int, lm) {}
  void put_count(int, lm) {}
  void sync() {}
  lf get_shift(int, lf x, float fill) { return lane_shift_up(x, fill); }
  lf get_shift_keep(int, lf x, lf &keep) { return lane_shift_up_keep(x, keep); }
  float get_min(int, lf x) { return wave_min(x); }
  uint32_t get_minu(int, lu x) { return wave_minu(x); }
  float get_lane(int, lf x, int l) { return read_lane(x, l); }
  bool get_any(int, lm m) { return wave_any(m); }
  int get_count(int, lm m) { int c = 0; EM_FOR c += m.v[i_] ? 1 : 0; return c; }
};

template <int Q> inline void load_q(float const *row, lu lane, lf (&out)[Q])
{
  for (int q = 0; q < Q; ++q) EM_FOR out[q].v[i_] = row[lane.v[i_] * Q + q];
}

// emission rows { null, bg, 0, 0, match[0..Kp) } behind one byte offset (plain pointers here)
struct RowSrc { char const *base; uint32_t bytes; };
inline RowSrc rowsrc_make(float const *base, uint32_t bytes) { return RowSrc{reinterpret_cast<char const *>(base), bytes}; }
template <int Q> inline lu row_lane_offset(lu lane) { return lane * (uint32_t)(Q * 4) + (uint32_t)16; }
inline void load_row_hdr(RowSrc const &r, uint32_t soff, float &nil, float &bg)
{
  float const *h = reinterpret_cast<float const *>(r.base + soff);
  nil = h[0];
  bg = h[1];
}
template <int Q> inline void load_row_q(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[Q])
{
  for (int q = 0; q < Q; ++q)
    EM_FOR
    {
      uint32_t const at = soff + voff.v[i_] + 4u * (uint32_t)q;
      out[q].v[i_] = at + 4u <= r.bytes ? *reinterpret_cast<float const *>(r.base + at) : 0.0f; // buffer range check
    }
}

template <int Q> inline void store_q(float *row, lu lane, lf const (&v)[Q])
{
  for (int q = 0; q < Q; ++q) EM_FOR row[lane.v[i_] * Q + q] = v[q].v[i_];
}
inline void store_lane(float *p, lu lane, lf v) { EM_FOR p[lane.v[i_]] = v.v[i_]; }
inline lf load_lane(float const *p, lu lane) { lf r; EM_FOR r.v[i_] = p[lane.v[i_]]; return r; }
inline void store_sp_lane0(float *p, lu, lf N, lf B, lf J, lf E, lf C)
{
  p[0] = N.v[0]; p[1] = B.v[0]; p[2] = J.v[0]; p[3] = E.v[0]; p[4] = C.v[0];
}

template <int Q> inline void store_nodes_q(uint16_t *row, int K, lu lane, lu const (&w)[Q])
{
  for (int q = 0; q < Q; ++q)
    EM_FOR
    {
      int k = (int)lane.v[i_] * Q + q;
      if (k < K) row[k] = (uint16_t)w[q].v[i_];
    }
}

inline void store_u32_lane0(uint32_t *p, lu, uint32_t v) { *p = v; }
inline void store_f32_lane0(float *p, lu, float v) { *p = v; }

// ---- several windows per wavefront (viterbi_pack.h): always one 64-lane wave ----
inline lu lane_shr(lu x, int s) { lu r; EM_FOR r.v[i_] = x.v[i_] >> s; return r; }
inline lf lneg(lf x) { lf r; EM_FOR r.v[i_] = -x.v[i_]; return r; }
inline lu operator&(lu a, lu b) { lu r; EM_FOR r.v[i_] = a.v[i_] & b.v[i_]; return r; }

struct PackSrc
{
  float const *rows;
  int Kp;
  DcpCodeRow const *codes;
  uint32_t ncodes;
  lu col_off; // byte offset inside a record
};
inline PackSrc packsrc_make(float const *rows, int Kp, DcpCodeRow const *code_rows, uint32_t ncode_rows, lu col_off)
{
  return PackSrc{rows, Kp, code_rows, ncode_rows, col_off};
}
inline void load_code_row(PackSrc const &s, lu row, lu (&code)[5])
{
  for (int t = 0; t < 5; ++t) EM_FOR code[t].v[i_] = row.v[i_] < s.ncodes ? s.codes[row.v[i_]].c[t] : 0u; // range check
}
inline lf quad_bcast0(lf x) { lf r; EM_FOR r.v[i_] = x.v[i_ & ~3]; return r; }
template <int S> inline lf group_bcast0(lf x) { lf r; EM_FOR r.v[i_] = x.v[i_ & ~(S - 1)]; return r; }
template <int Q> inline void load_pack_q(PackSrc const &s, lu code, lf (&out)[Q])
{
  for (int q = 0; q < Q; ++q)
    EM_FOR out[q].v[i_] = s.rows[(size_t)code.v[i_] * (size_t)(s.Kp + DCP_ROW_HDR) + s.col_off.v[i_] / 4 + q];
}
template <int Q> inline void load_cols(float const *row, lu col, lf (&out)[Q])
{
  for (int q = 0; q < Q; ++q) EM_FOR out[q].v[i_] = row[col.v[i_] + q];
}
inline lu load_u32_at(uint32_t const *p, lu i) { lu r; EM_FOR r.v[i_] = p[i.v[i_]]; return r; }
inline lf load_f32_at(float const *p, lu i) { lf r; EM_FOR r.v[i_] = p[i.v[i_]]; return r; }
inline void store_f32_where(float *p, lu i, lm m, lf v) { EM_FOR if (m.v[i_]) p[i.v[i_]] = v.v[i_]; }
template <int S> inline lf group_min(lf v)
{
  lf r;
  for (int g = 0; g < 64 / S; ++g)
  {
    float m = v.v[g * S];
    for (int i = g * S; i < g * S + S; ++i) m = fminf(m, v.v[i]);
    for (int i = g * S; i < g * S + S; ++i) r.v[i] = m;
  }
  return r;
}

inline void add_quad0_x5(lf (&r)[5], lf const (&s)[5], lf const (&e)[5])
{
  for (int t = 0; t < 5; ++t) r[t] = s[t] + quad_bcast0(e[t]);
}
template <int S> inline lf group_min01(lf v)
{
  lf r;
  EM_FOR r.v[i_] = fminf(v.v[i_ & ~(S - 1)], v.v[(i_ & ~(S - 1)) + 1]);
  return r;
}

static thread_local lf em_pack_stash[6][8];
template <int Q> inline void pack_stash(int slot, lf const (&v)[Q]) { for (int q = 0; q < Q; ++q) em_pack_stash[slot][q] = v[q]; }
struct PackFold { int unused; };
inline void pack_unstash_issue(PackFold &) {}
template <int Q> inline void pack_unstash_wait(PackFold &, lf (&BM)[Q], lf (&MM)[Q], lf (&IM)[Q], lf (&DM)[Q], lf (&II)[Q], lf (&MI)[Q])
{
  for (int q = 0; q < Q; ++q)
  {
    BM[q] = em_pack_stash[0][q];
    MM[q] = em_pack_stash[1][q];
    IM[q] = em_pack_stash[2][q];
    DM[q] = em_pack_stash[3][q];
    II[q] = em_pack_stash[4][q];
    MI[q] = em_pack_stash[5][q];
  }
}

typedef float lds_float;
template <int N> inline void load_lds_q(lds_float const *t, lu idx, lf (&out)[N])
{
  for (int q = 0; q < N; ++q) EM_FOR out[q].v[i_] = t[idx.v[i_] + q];
}
