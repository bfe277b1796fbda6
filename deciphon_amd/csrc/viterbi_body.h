// viterbi_body.h -- the quasi-codon Viterbi recurrences, one wavefront per
// (profile x window) problem, written against the lane vocabulary of
// lane_ops_gpu.h (tests/emul/ re-instantiates it with 64-wide arrays).
//
// Layout: lane e of the wave owns the Q consecutive profile positions
// k = e*Q + q (the reference's striping k = e*Q + q, c-core/viterbi.c:220-221,
// with 64 lanes instead of 4/8/16).  Rows l = 1..L are sequential (B of row l
// needs E of row l, c-core/viterbi.c:582); the five emission lengths t = 1..5 of
// a row and the K positions are the parallel work.
//
//   CostWave<Q>  viterbi_null + viterbi_cost (c-core/viterbi.c:696-724): scores only.
//   PathWave<Q>  viterbi_path (c-core/viterbi.c:726-732): back-pointers, pass by
//                pass exactly as the reference orders its strict-< updates.
#pragma once
#include "dcp_types.h"

#ifndef DCP_FN
#error "include a lane_ops_*.h before viterbi_body.h"
#endif

#define DCP_INF (__builtin_inff())

// ring slot of row l-t when row l has phase P = l % 5
#define DCP_SL(P, t) (((P) + 5 - (t)) % 5)

// =============================================================================
// Scores only.  Only minima matter here, and fp32 min is exact and rounding is
// monotone, so min_i((x_i + t_i) + m) == (min_i (x_i + t_i)) + m bit for bit.
// That lets every finished row z be folded ONCE into
//     Mpre_z[k] = min(B_z+BM[k], M_z[k-1]+MM[k], I_z[k-1]+IM[k], D_z[k-1]+DM[k])
//     Ipre_z[k] = min(I_z[k]+II[k], M_z[k]+MI[k])
// and row l then needs only  M_l[k] = min_t (Mpre_{l-t}[k] + match[c_t][k]),
// I_l[k] = min_t (Ipre_{l-t}[k] + bg[c_t])  -- the same fp32 values the reference
// gets from c-core/viterbi.c:526-536, in 35 instead of 120 VALU ops per cell.
// The four scalar chains N, J, C (c-core/viterbi.c:492-502) and the null model R
// (:713) all have the shape X_l = min_t(Xpre_{l-t} + null[c_t]); they ride in
// lanes 0..3 of one register.
// E_l = min_k M_l[k]: the D_l[k] candidates of c-core/viterbi.c:541 are each
// some M_l[j] plus non-negative delete costs (costs are -log-probabilities), so
// they never lower the minimum.
// =============================================================================
// Delete runs that survive about six to eight positions are common on real profiles (PMC on the minifam bench:
// 2.5 turns per row with 3 positions per lane): that many positions are covered without asking.
#ifndef DCP_LAZY_POSITIONS
#define DCP_LAZY_POSITIONS 6
#endif
constexpr int dcp_lazy_turns(int Q) { return Q >= DCP_LAZY_POSITIONS ? 1 : (DCP_LAZY_POSITIONS + Q - 1) / Q; }

// STORE = true additionally writes every row's final values to a DP table in HBM
// (cells[l][{M,I,D}][Kp] and specials[l][8] = N,B,J,E,C) for the traceback of
// traceback.h -- the fast path pass.
// How a shape trades registers for LDS and prefetch distance (POLICY, bits):
//   1  the next row's emissions are asked for behind the D chain instead of before it: MD, DD and D are then not alive
//      beside them (Q registers fewer at the peak for each of the five emission lengths);
//   2  MD, DD, II, MI wait in LDS between their uses (a row needs each for a few instructions);
//   4  all eight transition arrays do, and a single wave takes the six of the fold back a few positions at a time.
// Defaults: (8,1) parks four arrays (260 -> under 256 VGPRs: two waves per SIMD instead of one), the multi-wave shapes of
// 8 positions all eight (the exchange needs room), single waves beyond 8 positions all of it (10 positions: 226 VGPRs).
#ifndef DCP_SHAPE_POLICY
#define DCP_SHAPE_POLICY(Q, W) ((W) == 1 ? ((Q) > 8 ? 7 : (Q) == 8 ? 2 : 0) : ((Q) >= 8 ? 6 : 0))
#endif
// The bulk cost kernels (dcp_cost_kernel) run (4,1) and (5,1) with bit 1 as well: 147 -> 128 VGPRs puts a fourth
// wavefront on every SIMD, and (5,1) fits three without spilling inside the row loop (viterbi_kernels.hip,
// DCP_COST_WAVES).  The path pass's kernels -- a few lone wavefronts that wait out every load -- keep the early request.
#ifndef DCP_COST_POLICY
#define DCP_COST_POLICY(Q, W) (DCP_SHAPE_POLICY(Q, W) | ((W) == 1 && ((Q) == 4 || (Q) == 5) ? 1 : 0))
#endif
template <int Q, int W, bool STORE = false, int POLICY = DCP_SHAPE_POLICY(Q, W)> struct CostWave
{
  // lazy D->D turns taken before the first vote: a turn is 2Q + 2 instructions straight-line, a vote costs a
  // ballot, a scalar branch and the register copies of a loop.  Extra turns change nothing (min is idempotent).
  static constexpr int TURNS = dcp_lazy_turns(Q);
  static constexpr bool LATE_FETCH = (POLICY & 1) != 0;
  static constexpr bool STASH = (POLICY & 6) != 0;
  static constexpr bool STASH6 = (POLICY & 4) != 0;
  static constexpr bool CHUNKED = STASH6 && W == 1;
  static constexpr int SLOTS = STASH6 ? 8 : 4; // arrays parked per wave
  Group<W> g;
  float *__restrict__ tab_cells = nullptr; // [(L+1)][3][Kp]
  float *__restrict__ tab_sp = nullptr;    // [(L+1)][DCP_SP_STRIDE]
  int tabKp = 0;
  // blocks (dcp_types.h, "fast path pass in blocks"): ckpt_every > 0 saves the folded ring after every
  // ckpt_every-th row into ckpt_out; ckpt_in resumes from a saved ring at row row_base, and rows are stored at
  // table slot l - row_base
  float *__restrict__ ckpt_out = nullptr;
  float const *__restrict__ ckpt_in = nullptr;
  int ckpt_every = 0;
  int row_base = 0;

  DCP_FN void save_ring(float *__restrict__ to)
  {
#pragma unroll
    for (int s = 0; s < 5; ++s)
    {
      store_q<Q>(to + (size_t)s * tabKp, g.lane, Mpre[s]);
      store_q<Q>(to + (size_t)(5 + s) * tabKp, g.lane, Ipre[s]);
      store_lane(to + (size_t)10 * tabKp + s * 64 * W, g.lane, Spre[s]);
    }
    store_lane(to + (size_t)10 * tabKp + 5 * 64 * W, g.lane, X);
  }
  DCP_FN void load_ring(float const *__restrict__ from)
  {
#pragma unroll
    for (int s = 0; s < 5; ++s)
    {
      load_q<Q>(from + (size_t)s * tabKp, g.lane, Mpre[s]);
      load_q<Q>(from + (size_t)(5 + s) * tabKp, g.lane, Ipre[s]);
      Spre[s] = load_lane(from + (size_t)10 * tabKp + s * 64 * W, g.lane);
    }
    X = load_lane(from + (size_t)10 * tabKp + 5 * 64 * W, g.lane);
  }
  lf BM[Q], MM[Q], MI[Q], MD[Q], IM[Q], II[Q], DM[Q], DD[Q];
  lf Mpre[5][Q], Ipre[5][Q], Spre[5];
  lf em[5][Q];
  lf sa, sb;
  lf X;
  lf NBv, EBv, JBv;    // uniform, pinned to VGPRs
  lf shM, shI, shD;    // destinations of the k-1 shifts; lane 0 stays +inf
  float nil[5], bgv[5];
  float ET, CT, RR;
  float E;
  RowSrc rows;
  lu voff;
  uint32_t stride_bytes;
  DcpCodeRow const *__restrict__ codes;

  DcpCodeRow cr; // codes of the next row to fetch, already resident in SGPRs

  // Issues everything the row coded in `cr` needs for its five emission lengths (one
  // scalar offset each), then pulls the codes of the row after it.  Codes run two rows
  // ahead of the DP and emissions one, so neither the scalar nor the vector load
  // latency sits on the row-to-row critical path.
  DCP_FN void fetch(int l_after, int L)
  {
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      uint32_t const off = cr.c[t] * stride_bytes;
      load_row_hdr(rows, off, nil[t], bgv[t]);
      load_row_q<Q>(rows, voff, off, em[t]);
    }
    cr = codes[l_after <= L ? l_after : L];
  }

  DCP_FN void init(float const *__restrict__ pool, DcpProfileDev const &pf, DcpCodeRow const *__restrict__ code_rows,
                   float const *__restrict__ xt)
  {
    // Wavefronts with more work per row take longer over the same window: they get the SIMD's issue slots first
    // (a wavefront alone issues every ~7.5 cycles, so what the others lose they make up once it is done).  On a
    // launch of one generation of mixed classes -- minifam x 1000 reads -- the classes then end together.
#ifndef DCP_WAVE_PRIO
#define DCP_WAVE_PRIO(Q, W) ((W) > 1 || (Q) >= 6 ? 3 : (Q) >= 4 ? 2 : (Q) == 3 ? 1 : 0)
#endif
    wave_priority<DCP_WAVE_PRIO(Q, W)>();
    g.init();
    lu const lane = g.lane;
    int const Kp = pf.Kp;
    stride_bytes = (uint32_t)(Kp + DCP_ROW_HDR) * 4u;
    rows = rowsrc_make(pool + pf.rows_off, (uint32_t)DCP_TABLE_SIZE * stride_bytes);
    voff = row_lane_offset<Q>(lane);
    codes = code_rows;
    float const *__restrict__ trans = pool + pf.trans_off;
    load_q<Q>(trans + DCP_BM * Kp, lane, BM);
    load_q<Q>(trans + DCP_MM * Kp, lane, MM);
    load_q<Q>(trans + DCP_MI * Kp, lane, MI);
    load_q<Q>(trans + DCP_MD * Kp, lane, MD);
    load_q<Q>(trans + DCP_IM * Kp, lane, IM);
    load_q<Q>(trans + DCP_II * Kp, lane, II);
    load_q<Q>(trans + DCP_DM * Kp, lane, DM);
    load_q<Q>(trans + DCP_DD * Kp, lane, DD);
    NBv = lf_pin(xt[DCP_NB]);
    EBv = lf_pin(xt[DCP_EB]);
    JBv = lf_pin(xt[DCP_JB]);
    shM = shI = shD = lf_splat(DCP_INF);
    g.put_tdd(DD); // W > 1: what running through a whole wave of delete states costs (row())
    if constexpr (STASH)
    {
      g.template stash_q<Q, SLOTS>(0, MD);
      g.template stash_q<Q, SLOTS>(1, DD);
      g.template stash_q<Q, SLOTS>(2, II);
      g.template stash_q<Q, SLOTS>(3, MI);
      if constexpr (STASH6)
      {
        g.template stash_q<Q, SLOTS>(4, IM);
        g.template stash_q<Q, SLOTS>(5, DM);
        g.template stash_q<Q, SLOTS>(6, BM);
        g.template stash_q<Q, SLOTS>(7, MM);
      }
    }
    ET = xt[DCP_ET];
    CT = xt[DCP_CT];
    RR = xt[DCP_RR];
    float const SN = xt[DCP_SN], SB = xt[DCP_SB];
    lm const l0 = lequ(lane, lu_splat(0)), l1 = lequ(lane, lu_splat(1));
    lm const l2 = lequ(lane, lu_splat(2)), l3 = lequ(lane, lu_splat(3));
    lf const inf = lf_splat(DCP_INF);
    // lane: 0 = N, 1 = J, 2 = C, 3 = R.  Xpre = min(E + sa, X + sb)
    sa = lsel(l1, lf_splat(xt[DCP_EJ]), lsel(l2, lf_splat(xt[DCP_EC]), inf));
    sb = lsel(l0, lf_splat(xt[DCP_NN]),
              lsel(l1, lf_splat(xt[DCP_JJ]), lsel(l2, lf_splat(xt[DCP_CC]), lsel(l3, lf_splat(RR), inf))));
    // row 0 (c-core/viterbi.c:471-473, :703): S = 0, B = SB, R = -RR, rest +inf
#pragma unroll
    for (int s = 0; s < 5; ++s)
    {
      Spre[s] = inf;
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        Mpre[s][q] = inf;
        Ipre[s][q] = inf;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) Mpre[0][q] = lf_splat(SB) + BM[q];
    Spre[0] = lsel(l0, lf_splat(0.0f + SN), lsel(l3, lf_splat(-RR + RR), inf));
    X = lsel(l3, lf_splat(-RR), inf);
    E = DCP_INF;
    tabKp = Kp;
    if (ckpt_in)
      load_ring(ckpt_in); // a block that starts at row row_base > 0
    else if (STORE)
      store_row0(SB);
  }

  template <int P> DCP_FN void row(int l, int L)
  {
    lf M[Q], I[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      M[q] = lmin3(lmin3(Mpre[DCP_SL(P, 5)][q] + em[4][q], Mpre[DCP_SL(P, 4)][q] + em[3][q],
                         Mpre[DCP_SL(P, 3)][q] + em[2][q]),
                   Mpre[DCP_SL(P, 2)][q] + em[1][q], Mpre[DCP_SL(P, 1)][q] + em[0][q]);
      I[q] = lmin3(lmin3(Ipre[DCP_SL(P, 5)][q] + bgv[4], Ipre[DCP_SL(P, 4)][q] + bgv[3],
                         Ipre[DCP_SL(P, 3)][q] + bgv[2]),
                   Ipre[DCP_SL(P, 2)][q] + bgv[1], Ipre[DCP_SL(P, 1)][q] + bgv[0]);
    }
    X = lmin3(lmin3(Spre[DCP_SL(P, 5)] + nil[4], Spre[DCP_SL(P, 4)] + nil[3], Spre[DCP_SL(P, 3)] + nil[2]),
              Spre[DCP_SL(P, 2)] + nil[1], Spre[DCP_SL(P, 1)] + nil[0]);

    // emissions of this row are consumed: fetch the next row's behind the rest
    if constexpr (!LATE_FETCH)
      if (l < L) fetch(l + 2, L);

    lf m = M[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) m = lmin(m, M[q]);
    lf D[Q];
    lf Msh0, Ish0, Dsh0, B;
    float N, J;
    if constexpr (STASH)
    {
      g.template unstash_q<Q, SLOTS>(0, MD);
      g.template unstash_q<Q, SLOTS>(1, DD);
    }
    if constexpr (W == 1)
    {
      Msh0 = lane_shift_up_keep(M[Q - 1], shM);
      Ish0 = lane_shift_up_keep(I[Q - 1], shI);
      E = wave_min(m);
      N = read_lane(X, 0);
      J = read_lane(X, 1);
      B = lmin3(N + NBv, E + EBv, J + JBv); // c-core/viterbi.c:495-496,582-583 (uniform)

      // D_l[k] = min(M_l[k-1] + MD[k], D_l[k-1] + DD[k])  (c-core/viterbi.c:538,553-580):
      // serial inside a lane, then carried across lanes until no lane improves (the
      // reference's lazy D->D loop, c-core/viterbi.c:569-580)
      D[0] = Msh0 + MD[0];
#pragma unroll
      for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
      Dsh0 = lane_shift_up_keep(D[Q - 1], shD);
      lf x = Dsh0 + DD[0];
#pragma unroll
      for (int turn = 0; turn < TURNS; ++turn)
      {
        D[0] = lmin(D[0], x);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh0 = lane_shift_up_keep(D[Q - 1], shD);
        x = Dsh0 + DD[0];
      }
      while (wave_any(llt(x, D[0]))) // one more lane boundary per turn
      {
        D[0] = lmin(D[0], x);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh0 = lane_shift_up_keep(D[Q - 1], shD);
        x = Dsh0 + DD[0];
      }
    }
    else
    {
      // W wavefronts, ONE barrier per row.  Before it every wave finishes what it can alone:
      // its D chain with nothing entering at its first lane.  It publishes one record
      // {M, I, D of its last position, min of its M}; after the barrier the first lane of
      // each wave takes M, I, D of the previous wave's last position as its k-1 neighbour.
      // That is exact provided the D published by every wave is already final, i.e. nothing
      // entering a wave at its first lane can run through ALL its positions and still arrive
      // below the published value.  Whatever enters costs at least E (it is some M of this row
      // plus non-negative costs) and running through wave w adds tdd(w) = sum of DD over its
      // positions but the first, so  E + tdd(w) >= D_last(w)  for every w (with a margin for the
      // order of the fp32 additions) proves it -- every wave evaluates the same test on the
      // same records.  If it fails (profiles whose delete runs are nearly free) the row falls
      // back to the exchange-until-stable protocol below, barriers and all.
      int const par = l & 1;
      lf const inf = lf_splat(DCP_INF);
      lf x, Dsh;
      lm better;
      D[0] = g.seg_shift_up(M[Q - 1], inf) + MD[0];
#pragma unroll
      for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
      Dsh = g.seg_shift_up(D[Q - 1], inf);
      x = Dsh + DD[0];
      better = llt(x, D[0]);
      while (g.seg_any(better))
      {
        D[0] = lmin(D[0], x);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh = g.seg_shift_up(D[Q - 1], inf);
        x = Dsh + DD[0];
        better = llt(x, D[0]);
      }
      g.put_rec(par, M[Q - 1], I[Q - 1], D[Q - 1], m);
      g.put_lanes4(GS_X0 + par, X);
      g.sync();
      lf Mp, Ip, Dp;
      bool could;
      g.get_prev(par, Mp, Ip, Dp);
      g.get_e_could(par, E, could);
      g.get_nj(GS_X0 + par, X, N, J);
      Msh0 = g.seg_shift_up(M[Q - 1], Mp);
      Ish0 = g.seg_shift_up(I[Q - 1], Ip);
      B = lmin3(N + NBv, E + EBv, J + JBv);
      if (!could)
      {
        // the first lane of each wave: D[0] = min(M[k-1] + MD, D[k-1] + DD) with the neighbour's
        // values; then on through the wave while it improves anything
        D[0] = lsel(g.seg_first(), lmin(Mp + MD[0], Dp + DD[0]), D[0]);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh0 = g.seg_shift_up(D[Q - 1], Dp);
        x = Dsh0 + DD[0];
        better = llt(x, D[0]);
        while (g.seg_any(better))
        {
          D[0] = lmin(D[0], x);
#pragma unroll
          for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
          Dsh0 = g.seg_shift_up(D[Q - 1], Dp);
          x = Dsh0 + DD[0];
          better = llt(x, D[0]);
        }
      }
      else
      {
        g.note_fallback();
        D[0] = Msh0 + MD[0];
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
        g.put_last(GS_D, D[Q - 1]);
        g.sync();
        Dsh0 = g.get_shift(GS_D, D[Q - 1], DCP_INF);
        x = Dsh0 + DD[0];
        better = llt(x, D[0]);
        g.put_any(GS_F, better);
        g.sync();
        while (g.get_any(GS_F, better))
        {
          D[0] = lmin(D[0], x);
#pragma unroll
          for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
          g.put_last(GS_D, D[Q - 1]);
          g.sync();
          Dsh0 = g.get_shift(GS_D, D[Q - 1], DCP_INF);
          x = Dsh0 + DD[0];
          better = llt(x, D[0]);
          g.put_any(GS_F, better);
          g.sync();
        }
      }
    }

    // fold row l into the ring (slot P held row l-5, no longer needed)
    if constexpr (LATE_FETCH)
    {
      // the next row's emissions are asked for only now (ten positions per lane with MD, DD and D alive beside them
      // would not fit 256 registers: 23 scratch accesses per row)
      sched_fence();
      if (l < L) fetch(l + 2, L);
      sched_fence();
    }
    if constexpr (CHUNKED)
    {
      // more than eight positions in one wave: the six arrays of the fold come back a few positions at a time
      constexpr int N = DcpStashChunk<Q>::N;
#pragma unroll
      for (int j = 0; j < Q / N; ++j)
      {
        lf bm[N], mm[N], im[N], dm[N], ii[N], mi[N];
        g.template unstash_chunk<Q, SLOTS>(2, j, ii);
        g.template unstash_chunk<Q, SLOTS>(3, j, mi);
        g.template unstash_chunk<Q, SLOTS>(4, j, im);
        g.template unstash_chunk<Q, SLOTS>(5, j, dm);
        g.template unstash_chunk<Q, SLOTS>(6, j, bm);
        g.template unstash_chunk<Q, SLOTS>(7, j, mm);
#pragma unroll
        for (int i = 0; i < N; ++i)
        {
          int const q = N * j + i;
          lf const Ml = q ? M[q ? q - 1 : 0] : Msh0;
          lf const Il = q ? I[q ? q - 1 : 0] : Ish0;
          lf const Dl = q ? D[q ? q - 1 : 0] : Dsh0;
          Mpre[P][q] = lmin3(B + bm[i], Ml + mm[i], lmin(Il + im[i], Dl + dm[i]));
          Ipre[P][q] = lmin(I[q] + ii[i], M[q] + mi[i]);
        }
        sched_fence(); // (the scheduler would bring every chunk's LDS reads up front: 36 registers more at the peak)
      }
    }
    else if constexpr (STASH)
    {
      g.template unstash_q<Q, SLOTS>(2, II);
      g.template unstash_q<Q, SLOTS>(3, MI);
      if constexpr (STASH6)
      {
        g.template unstash_q<Q, SLOTS>(4, IM);
        g.template unstash_q<Q, SLOTS>(5, DM);
        g.template unstash_q<Q, SLOTS>(6, BM);
        g.template unstash_q<Q, SLOTS>(7, MM);
      }
    }
    if constexpr (!CHUNKED)
    {
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        lf const Ml = q ? M[q ? q - 1 : 0] : Msh0;
        lf const Il = q ? I[q ? q - 1 : 0] : Ish0;
        lf const Dl = q ? D[q ? q - 1 : 0] : Dsh0;
        Mpre[P][q] = lmin3(B + BM[q], Ml + MM[q], lmin(Il + IM[q], Dl + DM[q]));
        Ipre[P][q] = lmin(I[q] + II[q], M[q] + MI[q]);
      }
    }
    Spre[P] = lmin(lf_splat(E) + sa, X + sb);
    if (STORE)
    {
      float *row = tab_cells + (size_t)(l - row_base) * 3 * (size_t)tabKp;
      store_q<Q>(row, g.lane, M);
      store_q<Q>(row + tabKp, g.lane, I);
      store_q<Q>(row + 2 * tabKp, g.lane, D);
      float const C = g.get_lane(GS_X0 + (l & 1), X, 2);
      store_sp_lane0(tab_sp + (size_t)(l - row_base) * DCP_SP_STRIDE, g.lane, lf_splat(N), B, lf_splat(J), lf_splat(E),
                     lf_splat(C));
    }
  }

  // row 0 of the DP table: S = 0 (implicit), B = SB, everything else +inf
  DCP_FN void store_row0(float SB)
  {
    lf inf[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) inf[q] = lf_splat(DCP_INF);
    store_q<Q>(tab_cells, g.lane, inf);
    store_q<Q>(tab_cells + tabKp, g.lane, inf);
    store_q<Q>(tab_cells + 2 * tabKp, g.lane, inf);
    lf const i = lf_splat(DCP_INF);
    store_sp_lane0(tab_sp, g.lane, i, lf_splat(SB), i, i, i);
  }

  // out[0] = viterbi_null(), out[1] = viterbi_cost().  Lend < L: stop after row Lend (a block; out untouched).
  DCP_FN void run(int L, float *out, int Lend = -1)
  {
    if (Lend < 0 || Lend > L) Lend = L;
    int l = row_base + 1; // row_base is a multiple of 5: the block starts in phase 1 like row 1
    if (l <= Lend)
    {
      cr = codes[l];
      fetch(l + 1, Lend);
    }
    for (; l + 4 <= Lend; l += 5)
    {
      row<1>(l, Lend);
      row<2>(l + 1, Lend);
      row<3>(l + 2, Lend);
      row<4>(l + 3, Lend);
      row<0>(l + 4, Lend);
      if (ckpt_every > 0 && (l + 4) % ckpt_every == 0 && l + 4 + 5 < L) // checkpoint j exists iff j * B < L - 5
        save_ring(ckpt_out + (size_t)((l + 4) / ckpt_every - 1) * (size_t)dcp_ckpt_floats(tabKp, W));
    }
    if (l <= Lend) row<1>(l++, Lend);
    if (l <= Lend) row<2>(l++, Lend);
    if (l <= Lend) row<3>(l++, Lend);
    if (l <= Lend) row<4>(l++, Lend);
    if (Lend < L) return;
    g.sync();
    g.put_lanes4(GS_X, X);
    g.sync();
    float const C = g.get_lane(GS_X, X, 2);
    float const R = g.get_lane(GS_X, X, 3);
    float const T = L > 0 ? __builtin_fminf(E + ET, C + CT) : DCP_INF; // c-core/viterbi.c:585-586,599
    store_f32_lane0(out + 0, g.lane, R);
    store_f32_lane0(out + 1, g.lane, T);
  }
};

// =============================================================================
// Scores of profiles too long for the registers of one workgroup (K > 4096).
// The same recurrences as CostWave, strip by strip: a strip is KS = 64*Q*W positions, the
// workgroup walks the S = Kp/KS strips of a row left to right, and what CostWave keeps in
// registers lives in HBM/L2 between visits:
//   ring[0][z][k] = rest_z[k] = min(M_z[k-1]+MM[k], I_z[k-1]+IM[k], D_z[k-1]+DM[k])
//   ring[1][z][k] = Ipre_z[k]                       (z = row % 5, five rows back)
// The B term of Mpre_z[k] = min(B_z + BM[k], rest_z[k]) cannot be folded in when row z is
// finished strip by strip -- B_z needs E_z = min over ALL strips -- so it is applied when the
// row is used: M_l[k] = min_t(min(B_{l-t} + BM[k], rest_{l-t}[k]) + match[c_t][k]), the same
// fp32 values.  k-1 across a strip boundary and the D chain entering a strip come from the
// previous strip of the same row, which is complete: its last position is "the record of the
// wave before wave 0" of the one-barrier exchange (CostWave::row).
// =============================================================================
template <int Q, int W, bool STORE = false> struct StripWave
{
  Group<W> g;
  float *__restrict__ tab_cells = nullptr; // [(L+1)][3][Kp]
  float *__restrict__ tab_sp = nullptr;    // [(L+1)][DCP_SP_STRIDE]
  float *__restrict__ ring = nullptr;      // [2][5][Kp]
  float const *__restrict__ trans = nullptr;
  int Kp = 0, S = 0;
  lf Spre[5];
  lf sa, sb, X;
  float Bz[5]; // B of the five previous rows, by ring slot
  float NB, EB, JB, ET, CT, RR, E;
  RowSrc rows;
  lu voff;
  uint32_t stride_bytes;
  DcpCodeRow const *__restrict__ codes;
  int tick = 0; // parity of the next exchange

  enum { KS = 64 * Q * W };

  DCP_FN void init(float const *__restrict__ pool, DcpProfileDev const &pf, DcpCodeRow const *__restrict__ code_rows,
                   float const *__restrict__ xt)
  {
    g.init();
    lu const lane = g.lane;
    Kp = pf.Kp;
    S = Kp / KS;
    stride_bytes = (uint32_t)(Kp + DCP_ROW_HDR) * 4u;
    rows = rowsrc_make(pool + pf.rows_off, (uint32_t)DCP_TABLE_SIZE * stride_bytes);
    voff = row_lane_offset<Q>(lane);
    codes = code_rows;
    trans = pool + pf.trans_off;
    NB = xt[DCP_NB];
    EB = xt[DCP_EB];
    JB = xt[DCP_JB];
    ET = xt[DCP_ET];
    CT = xt[DCP_CT];
    RR = xt[DCP_RR];
    float const SN = xt[DCP_SN], SB = xt[DCP_SB];
    lm const l0 = lequ(lane, lu_splat(0)), l1 = lequ(lane, lu_splat(1));
    lm const l2 = lequ(lane, lu_splat(2)), l3 = lequ(lane, lu_splat(3));
    lf const inf = lf_splat(DCP_INF);
    sa = lsel(l1, lf_splat(xt[DCP_EJ]), lsel(l2, lf_splat(xt[DCP_EC]), inf));
    sb = lsel(l0, lf_splat(xt[DCP_NN]),
              lsel(l1, lf_splat(xt[DCP_JJ]), lsel(l2, lf_splat(xt[DCP_CC]), lsel(l3, lf_splat(RR), inf))));
#pragma unroll
    for (int z = 0; z < 5; ++z)
    {
      Spre[z] = inf;
      Bz[z] = DCP_INF;
    }
    Bz[0] = SB; // row 0: B = S + SB, everything else +inf (c-core/viterbi.c:471-473)
    Spre[0] = lsel(l0, lf_splat(0.0f + SN), lsel(l3, lf_splat(-RR + RR), inf));
    X = lsel(l3, lf_splat(-RR), inf);
    E = DCP_INF;
    lf infq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) infq[q] = inf;
    for (int s = 0; s < S; ++s)
    {
      lf DD[Q];
      load_q<Q>(trans + (size_t)DCP_DD * Kp + (size_t)s * KS, lane, DD);
      g.put_tdd_strip(s, DD); // what running through a whole wave of this strip's delete states costs
      for (int z = 0; z < 10; ++z) store_q<Q>(ring + (size_t)z * Kp + (size_t)s * KS, lane, infq);
      if (STORE)
        for (int z = 0; z < 3; ++z) store_q<Q>(tab_cells + (size_t)z * Kp + (size_t)s * KS, lane, infq);
    }
    if (STORE) store_sp_lane0(tab_sp, lane, inf, lf_splat(SB), inf, inf, inf);
    g.put_carry_inf(0);
    g.put_carry_inf(1);
    g.sync();
  }

  template <int P> DCP_FN void row(int l)
  {
    lu const lane = g.lane;
    lf const inf = lf_splat(DCP_INF);
    DcpCodeRow const cr = codes[l];
    uint32_t off[5];
    float nil[5], bgv[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      off[t] = cr.c[t] * stride_bytes;
      load_row_hdr(rows, off[t], nil[t], bgv[t]);
    }
    X = lmin3(lmin3(Spre[DCP_SL(P, 5)] + nil[4], Spre[DCP_SL(P, 4)] + nil[3], Spre[DCP_SL(P, 3)] + nil[2]),
              Spre[DCP_SL(P, 2)] + nil[1], Spre[DCP_SL(P, 1)] + nil[0]);
    g.put_lanes4(GS_X0 + (l & 1), X); // visible after this row's first exchange
    float Erun = DCP_INF;
    for (int s = 0; s < S; ++s)
    {
      size_t const col = (size_t)s * KS;
      lf BM[Q], MM[Q], MI[Q], MD[Q], IM[Q], II[Q], DM[Q], DD[Q];
      load_q<Q>(trans + (size_t)DCP_BM * Kp + col, lane, BM);
      load_q<Q>(trans + (size_t)DCP_MM * Kp + col, lane, MM);
      load_q<Q>(trans + (size_t)DCP_MI * Kp + col, lane, MI);
      load_q<Q>(trans + (size_t)DCP_MD * Kp + col, lane, MD);
      load_q<Q>(trans + (size_t)DCP_IM * Kp + col, lane, IM);
      load_q<Q>(trans + (size_t)DCP_II * Kp + col, lane, II);
      load_q<Q>(trans + (size_t)DCP_DM * Kp + col, lane, DM);
      load_q<Q>(trans + (size_t)DCP_DD * Kp + col, lane, DD);
      lf M[Q], I[Q], D[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) M[q] = I[q] = inf;
#pragma unroll
      for (int t = 5; t >= 1; --t)
      {
        lf r[Q], ip[Q], em[Q];
        load_q<Q>(ring + (size_t)DCP_SL(P, t) * Kp + col, lane, r);
        load_q<Q>(ring + (size_t)(5 + DCP_SL(P, t)) * Kp + col, lane, ip);
        load_row_q<Q>(rows, voff + (uint32_t)(col * 4), off[t - 1], em);
#pragma unroll
        for (int q = 0; q < Q; ++q)
        {
          M[q] = lmin(M[q], lmin(Bz[DCP_SL(P, t)] + BM[q], r[q]) + em[q]);
          I[q] = lmin(I[q], ip[q] + bgv[t - 1]);
        }
      }
      lf m = M[0];
#pragma unroll
      for (int q = 1; q < Q; ++q) m = lmin(m, M[q]);

      // the one-barrier exchange of CostWave::row, with the previous strip's last position
      // standing in front of wave 0
      int const par = tick;
      tick ^= 1;
      lf x, Dsh;
      lm better;
      D[0] = g.seg_shift_up(M[Q - 1], inf) + MD[0];
#pragma unroll
      for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
      Dsh = g.seg_shift_up(D[Q - 1], inf);
      x = Dsh + DD[0];
      better = llt(x, D[0]);
      while (g.seg_any(better))
      {
        D[0] = lmin(D[0], x);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh = g.seg_shift_up(D[Q - 1], inf);
        x = Dsh + DD[0];
        better = llt(x, D[0]);
      }
      g.put_rec(par, M[Q - 1], I[Q - 1], D[Q - 1], m);
      g.sync();
      lf Mp, Ip, Dp;
      float Es;
      bool could;
      g.get_prev(par, Mp, Ip, Dp);
      g.get_e_could_row(par, s, Erun, Es, could);
      lf const Msh0 = g.seg_shift_up(M[Q - 1], Mp);
      lf const Ish0 = g.seg_shift_up(I[Q - 1], Ip);
      lf Dsh0;
      if (!could)
      {
        D[0] = lsel(g.seg_first(), lmin(Mp + MD[0], Dp + DD[0]), D[0]);
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
        Dsh0 = g.seg_shift_up(D[Q - 1], Dp);
        x = Dsh0 + DD[0];
        better = llt(x, D[0]);
        while (g.seg_any(better))
        {
          D[0] = lmin(D[0], x);
#pragma unroll
          for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
          Dsh0 = g.seg_shift_up(D[Q - 1], Dp);
          x = Dsh0 + DD[0];
          better = llt(x, D[0]);
        }
      }
      else
      {
        g.note_fallback();
        D[0] = Msh0 + MD[0];
#pragma unroll
        for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
        g.put_last(GS_D, D[Q - 1]);
        g.sync();
        Dsh0 = g.get_shift_carry(GS_D, D[Q - 1], Dp);
        x = Dsh0 + DD[0];
        better = llt(x, D[0]);
        g.put_any(GS_F, better);
        g.sync();
        while (g.get_any(GS_F, better))
        {
          D[0] = lmin(D[0], x);
#pragma unroll
          for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
          g.put_last(GS_D, D[Q - 1]);
          g.sync();
          Dsh0 = g.get_shift_carry(GS_D, D[Q - 1], Dp);
          x = Dsh0 + DD[0];
          better = llt(x, D[0]);
          g.put_any(GS_F, better);
          g.sync();
        }
      }
      Erun = __builtin_fminf(Erun, Es);

      // fold this strip of row l into ring slot P (it held row l-5, already consumed above)
      lf rest[Q], ipre[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        lf const Ml = q ? M[q ? q - 1 : 0] : Msh0;
        lf const Il = q ? I[q ? q - 1 : 0] : Ish0;
        lf const Dl = q ? D[q ? q - 1 : 0] : Dsh0;
        rest[q] = lmin3(Ml + MM[q], Il + IM[q], Dl + DM[q]);
        ipre[q] = lmin(I[q] + II[q], M[q] + MI[q]);
      }
      store_q<Q>(ring + (size_t)P * Kp + col, lane, rest);
      store_q<Q>(ring + (size_t)(5 + P) * Kp + col, lane, ipre);
      if (STORE)
      {
        float *trow = tab_cells + (size_t)l * 3 * (size_t)Kp + col;
        store_q<Q>(trow, lane, M);
        store_q<Q>(trow + Kp, lane, I);
        store_q<Q>(trow + 2 * (size_t)Kp, lane, D);
      }
      // what stands in front of wave 0 in the next exchange: this strip's last position, or
      // nothing when the next exchange opens a new row
      if (s + 1 < S)
        g.put_carry(tick, M[Q - 1], I[Q - 1], D[Q - 1]);
      else
        g.put_carry_inf(tick);
    }
    E = Erun;
    float N, J;
    g.get_nj(GS_X0 + (l & 1), X, N, J);
    float const B = __builtin_fminf(__builtin_fminf(N + NB, E + EB), J + JB); // c-core/viterbi.c:495-496,582-583
    Bz[P] = B;
    Spre[P] = lmin(lf_splat(E) + sa, X + sb);
    if (STORE)
    {
      float const C = g.get_lane(GS_X0 + (l & 1), X, 2);
      store_sp_lane0(tab_sp + (size_t)l * DCP_SP_STRIDE, lane, lf_splat(N), lf_splat(B), lf_splat(J), lf_splat(E),
                     lf_splat(C));
    }
  }

  // out[0] = viterbi_null(), out[1] = viterbi_cost()
  DCP_FN void run(int L, float *out)
  {
    int l = 1;
    for (; l + 4 <= L; l += 5)
    {
      row<1>(l);
      row<2>(l + 1);
      row<3>(l + 2);
      row<4>(l + 3);
      row<0>(l + 4);
    }
    if (l <= L) row<1>(l++);
    if (l <= L) row<2>(l++);
    if (l <= L) row<3>(l++);
    if (l <= L) row<4>(l++);
    g.sync();
    g.put_lanes4(GS_X, X);
    g.sync();
    float const C = g.get_lane(GS_X, X, 2);
    float const R = g.get_lane(GS_X, X, 3);
    float const T = L > 0 ? __builtin_fminf(E + ET, C + CT) : DCP_INF;
    store_f32_lane0(out + 0, g.lane, R);
    store_f32_lane0(out + 1, g.lane, T);
  }
};

// =============================================================================
// Back-pointers.  Pointers depend on the ORDER of the reference's strict-<
// updates whenever two candidates tie exactly in fp32, so this pass keeps the
// reference's structure: for t = min(5,l)..1, each candidate as (x + trans) + emis,
// BM,MM,IM,DM / II,MI / MD then DD; E recomputed per pass.  Pointers are held
// directly as trellis fields (c-core/viterbi.c:631-694, c-core/trellis.h:42-56).
// =============================================================================
#define DCP_UPD(cur, ptr, val, newptr)                                         \
  do                                                                           \
  {                                                                            \
    lf const v_ = (val);                                                       \
    (ptr) = lselu(llt(v_, (cur)), lu_splat(newptr), (ptr));                    \
    (cur) = lmin((cur), v_);                                                   \
  } while (0)

#define DCP_UPDS(cur, ptr, val, newptr)                                        \
  do                                                                           \
  {                                                                            \
    float const v_ = (val);                                                    \
    if (v_ < (cur)) (ptr) = (newptr);                                          \
    (cur) = __builtin_fminf((cur), v_);                                        \
  } while (0)

template <int Q, int W> struct PathWave
{
  Group<W> g;
  lf BM[Q], MM[Q], MI[Q], MD[Q], IM[Q], II[Q], DM[Q], DD[Q];
  // ring of the five previous rows; *sh = value of position k-1 for q = 0
  lf M[5][Q], I[5][Q], D[5][Q], Msh[5], Ish[5], Dsh[5];
  float S[5], N[5], B[5], J[5], E[5], C[5];
  float xt[DCP_NUM_XTRANS];
  int K;
  RowSrc rows;
  lu voff;
  uint32_t stride_bytes;
  DcpCodeRow const *__restrict__ codes;
  DcpCodeRow cr;          // codes of the next row to fetch
  lf em[5][Q];            // emissions of the row being computed, one set per emission length
  float nil[5], bgv[5];
  uint32_t *__restrict__ xnodes;
  uint16_t *__restrict__ nodes;

  DCP_FN void init(float const *__restrict__ pool, DcpProfileDev const &pf, DcpCodeRow const *__restrict__ code_rows,
                   float const *__restrict__ xtp, uint32_t *__restrict__ xn, uint16_t *__restrict__ nd)
  {
    g.init();
    lu const lane = g.lane;
    K = pf.K;
    int const Kp = pf.Kp;
    stride_bytes = (uint32_t)(Kp + DCP_ROW_HDR) * 4u;
    rows = rowsrc_make(pool + pf.rows_off, (uint32_t)DCP_TABLE_SIZE * stride_bytes);
    voff = row_lane_offset<Q>(lane);
    codes = code_rows;
    xnodes = xn;
    nodes = nd;
    float const *__restrict__ trans = pool + pf.trans_off;
    load_q<Q>(trans + DCP_BM * Kp, lane, BM);
    load_q<Q>(trans + DCP_MM * Kp, lane, MM);
    load_q<Q>(trans + DCP_MI * Kp, lane, MI);
    load_q<Q>(trans + DCP_MD * Kp, lane, MD);
    load_q<Q>(trans + DCP_IM * Kp, lane, IM);
    load_q<Q>(trans + DCP_II * Kp, lane, II);
    load_q<Q>(trans + DCP_DM * Kp, lane, DM);
    load_q<Q>(trans + DCP_DD * Kp, lane, DD);
#pragma unroll
    for (int i = 0; i < DCP_NUM_XTRANS; ++i) xt[i] = xtp[i];
    lf const inf = lf_splat(DCP_INF);
#pragma unroll
    for (int s = 0; s < 5; ++s)
    {
      S[s] = N[s] = B[s] = J[s] = E[s] = C[s] = DCP_INF;
      Msh[s] = Ish[s] = Dsh[s] = inf;
#pragma unroll
      for (int q = 0; q < Q; ++q) M[s][q] = I[s][q] = D[s][q] = inf;
    }
    S[0] = 0.0f; // c-core/viterbi.c:471-473
    B[0] = xt[DCP_SB];
    // row 0 of the trellis: every field 0 (before(), c-core/viterbi.c:602-629)
    store_u32_lane0(xnodes, lane, 0u);
    lu zero[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) zero[q] = lu_splat(0);
    store_nodes_q<Q>(nodes, K, lane, zero);
  }

  // E back-pointer of the last pass: the reference reduces E over its own SIMD
  // lanes (k = e*Qr + q), first-wins inside a lane in the order
  // ME(0),DE(0)*,ME(1),DE(1),..,DE(0) and max(name,lane,q) across lanes
  // (c-core/viterbi.c:540-541,555-558, c-core/intrinsics.h:151-160).  A single
  // candidate equal to the minimum needs none of that.
  DCP_FN uint32_t e_field(float v, lf const (&Ma)[Q], lf const (&Da)[Q], lf const (&Mbefore)[Q],
                          lf const (&Dbefore)[Q])
  {
    if (!(v < DCP_INF)) return 0u;
    lu const lane = g.lane;
    lf const vv = lf_splat(v);
    lu field = lu_splat(0xffffffffu);
    lm has = llt_u(lu_splat(1), lu_splat(0)); // all false
    lm multi = has;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      lu const k2 = (lane * (uint32_t)Q + (uint32_t)q) * 2u;
      lm const mM = leq(Ma[q], vv), mD = leq(Da[q], vv);
      multi = lor(multi, lor(land(has, lor(mM, mD)), land(mM, mD)));
      has = lor(has, lor(mM, mD));
      field = lselu(mD, lminu(field, k2 + 1u), field);
      field = lselu(mM, lminu(field, k2), field);
    }
    g.put_count(GS_F, has);
    g.put_any(GS_X, multi);
    g.put_minu(GS_T0, field);
    g.sync();
    bool const single = g.get_count(GS_F, has) == 1 && !g.get_any(GS_X, multi);
    uint32_t const first = g.get_minu(GS_T0, field);
    if (single) return first;

    // exact tie between distinct candidates: apply the reference's rule
    int Qr = (K - 1) / DCP_REF_LANES + 1;
    if (Qr < 2) Qr = 2; // c-core/viterbi.c:195-199
    lf stale[Q];
    g.put_last(GS_M, Mbefore[Q - 1]);
    g.sync();
    lf const Mb0 = g.get_shift(GS_M, Mbefore[Q - 1], DCP_INF);
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      lf const lastMb = q ? Mbefore[q ? q - 1 : 0] : Mb0;
      stale[q] = lmin(Dbefore[q], lastMb + MD[q]);
    }
    uint32_t best = 0;
    for (int er = 0; er < DCP_REF_LANES; ++er)
    {
      lu ord = lu_splat(0xffffffffu);
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        lu const k = lane * (uint32_t)Q + (uint32_t)q;
        lu const lo = lu_splat((uint32_t)(er * Qr));
        lm const in = land(lnot(llt_u(k, lo)), llt_u(k, lo + (uint32_t)Qr));
        lu const qr = k - lo;
        lu const ordM = qr * 2u;
        lu const ordD = lselu(lequ(qr, lu_splat(0)),
                              lselu(leq(stale[q], vv), lu_splat(1u), lu_splat((uint32_t)(2 * Qr))), qr * 2u + 1u);
        ord = lselu(land(in, leq(Ma[q], vv)), lminu(ord, ordM), ord);
        ord = lselu(land(in, leq(Da[q], vv)), lminu(ord, ordD), ord);
      }
      int const slot = (er & 1) ? GS_T1 : GS_T0; // alternate: a slot is re-published only after a later sync
      g.put_minu(slot, ord);
      g.sync();
      uint32_t const w = g.get_minu(slot, ord);
      if (w != 0xffffffffu)
      {
        uint32_t name, qr;
        if (w == (uint32_t)(2 * Qr)) { name = 2u; qr = 0u; }
        else { name = (w & 1u) ? 2u : 1u; qr = w >> 1; }
        uint32_t const packed = (name << 28) | ((uint32_t)er << 24) | qr;
        if (packed > best) best = packed;
      }
    }
    uint32_t const q = best & 0x00ffffffu, e = (best >> 24) & 0xfu;
    uint32_t const k = e * (uint32_t)Qr + q;
    return 2u * k + ((best >> 28) == 2u ? 1u : 0u);
  }

  // one emission length t of row l; z = ring slot of row l-t
  template <int T, int Z>
  DCP_FN void pass(int l, lf (&Ma)[Q], lf (&Ia)[Q], lf (&Da)[Q], lu (&pM)[Q], lu (&pI)[Q], lu (&pD)[Q], float &Na,
                   float &Ba, float &Ja, float &Ea, float &Ca, float &Ta, uint32_t &pN, uint32_t &pB, uint32_t &pJ,
                   uint32_t &pE, uint32_t &pC, uint32_t &pT)
  {
    (void)l;
    constexpr uint32_t u = (uint32_t)(T - 1);
    float const nil = this->nil[T - 1];
    float const bg = this->bgv[T - 1];
    lf const(&em)[Q] = this->em[T - 1];

    DCP_UPDS(Na, pN, (S[Z] + xt[DCP_SN]) + nil, 0u + u); // c-core/viterbi.c:492-493
    DCP_UPDS(Na, pN, (N[Z] + xt[DCP_NN]) + nil, 5u + u);
    DCP_UPDS(Ba, pB, Na + xt[DCP_NB], 1u);               // :495-496 (S of a row > 0 is +inf)
    DCP_UPDS(Ja, pJ, (E[Z] + xt[DCP_EJ]) + nil, 0u + u); // :498-499
    DCP_UPDS(Ja, pJ, (J[Z] + xt[DCP_JJ]) + nil, 5u + u);
    DCP_UPDS(Ca, pC, (E[Z] + xt[DCP_EC]) + nil, 0u + u); // :501-502
    DCP_UPDS(Ca, pC, (C[Z] + xt[DCP_CC]) + nil, 5u + u);

    lf Mbefore[Q], Dbefore[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      Mbefore[q] = Ma[q];
      Dbefore[q] = Da[q];
    }

    lf const Bz = lf_splat(B[Z]);
#pragma unroll
    for (int q = 0; q < Q; ++q) // c-core/viterbi.c:526-536
    {
      lf const Ml = q ? M[Z][q ? q - 1 : 0] : Msh[Z];
      lf const Il = q ? I[Z][q ? q - 1 : 0] : Ish[Z];
      lf const Dl = q ? D[Z][q ? q - 1 : 0] : Dsh[Z];
      DCP_UPD(Ma[q], pM[q], (Bz + BM[q]) + em[q], 0u + u);
      DCP_UPD(Ma[q], pM[q], (Ml + MM[q]) + em[q], 5u + u);
      DCP_UPD(Ma[q], pM[q], (Il + IM[q]) + em[q], 10u + u);
      DCP_UPD(Ma[q], pM[q], (Dl + DM[q]) + em[q], 15u + u);
      DCP_UPD(Ia[q], pI[q], (I[Z][q] + II[q]) + lf_splat(bg), 5u + u);
      DCP_UPD(Ia[q], pI[q], (M[Z][q] + MI[q]) + lf_splat(bg), 0u + u);
    }
    g.put_last(GS_M, Ma[Q - 1]);
    g.sync();
    lf const Mash0 = g.get_shift(GS_M, Ma[Q - 1], DCP_INF);
#pragma unroll
    for (int q = 0; q < Q; ++q) // :538 and the stripe-0 repair :553-555
    {
      lf const lastMa = q ? Ma[q ? q - 1 : 0] : Mash0;
      DCP_UPD(Da[q], pD[q], lastMa + MD[q], 0u);
    }

    lf m = lmin(Ma[0], Da[0]); // :540-541,556-558
#pragma unroll
    for (int q = 1; q < Q; ++q) m = lmin(m, lmin(Ma[q], Da[q]));
    g.put_min(GS_E, m);
    g.sync();
    Ea = g.get_min(GS_E, m);
    if (T == 1) pE = e_field(Ea, Ma, Da, Mbefore, Dbefore);

    // D -> D: serial in k; done per lane, then carried across lanes until nothing
    // improves (:561-580).  Strict-< updates make the pointers order-free.
#pragma unroll
    for (int q = 1; q < Q; ++q) DCP_UPD(Da[q], pD[q], Da[q - 1] + DD[q], 1u);
    for (;;)
    {
      g.put_last(GS_D, Da[Q - 1]);
      g.sync();
      lf const x = g.get_shift(GS_D, Da[Q - 1], DCP_INF) + DD[0];
      lm const better = llt(x, Da[0]);
      g.put_any(GS_F, better);
      g.sync();
      if (!g.get_any(GS_F, better)) break;
      DCP_UPD(Da[0], pD[0], x, 1u);
#pragma unroll
      for (int q = 1; q < Q; ++q) DCP_UPD(Da[q], pD[q], Da[q - 1] + DD[q], 1u);
    }

    DCP_UPDS(Ba, pB, Ea + xt[DCP_EB], 2u); // :582-583
    DCP_UPDS(Ba, pB, Ja + xt[DCP_JB], 3u);
    DCP_UPDS(Ta, pT, Ea + xt[DCP_ET], 0u); // :585-586
    DCP_UPDS(Ta, pT, Ca + xt[DCP_CT], 1u);
  }

  // as CostWave::fetch: emissions run one row ahead of the DP, codes two
  DCP_FN void fetch(int l_after, int L)
  {
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      uint32_t const off = cr.c[t] * stride_bytes;
      load_row_hdr(rows, off, nil[t], bgv[t]);
      load_row_q<Q>(rows, voff, off, em[t]);
    }
    cr = codes[l_after <= L ? l_after : L];
  }

  template <int P> DCP_FN float row(int l, int L)
  {
    lf const inf = lf_splat(DCP_INF);
    lf Ma[Q], Ia[Q], Da[Q];
    lu pM[Q], pI[Q], pD[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      Ma[q] = Ia[q] = Da[q] = inf;
      pM[q] = pI[q] = pD[q] = lu_splat(0); // prev_core_state_init, c-core/viterbi.c:288-293
    }
    float Na = DCP_INF, Ba = DCP_INF, Ja = DCP_INF, Ea = DCP_INF, Ca = DCP_INF, Ta = DCP_INF;
    uint32_t pN = 0, pB = 0, pJ = 0, pE = 0, pC = 0, pT = 0; // prev_extr_state_init, :295-306

    if (l >= 5) pass<5, DCP_SL(P, 5)>(l, Ma, Ia, Da, pM, pI, pD, Na, Ba, Ja, Ea, Ca, Ta, pN, pB, pJ, pE, pC, pT);
    if (l >= 4) pass<4, DCP_SL(P, 4)>(l, Ma, Ia, Da, pM, pI, pD, Na, Ba, Ja, Ea, Ca, Ta, pN, pB, pJ, pE, pC, pT);
    if (l >= 3) pass<3, DCP_SL(P, 3)>(l, Ma, Ia, Da, pM, pI, pD, Na, Ba, Ja, Ea, Ca, Ta, pN, pB, pJ, pE, pC, pT);
    if (l >= 2) pass<2, DCP_SL(P, 2)>(l, Ma, Ia, Da, pM, pI, pD, Na, Ba, Ja, Ea, Ca, Ta, pN, pB, pJ, pE, pC, pT);
    pass<1, DCP_SL(P, 1)>(l, Ma, Ia, Da, pM, pI, pD, Na, Ba, Ja, Ea, Ca, Ta, pN, pB, pJ, pE, pC, pT);
    // this row's emissions are consumed: fetch the next row's while the pointers are packed
    if (l < L) fetch(l + 2, L);

    // after(): pack the row's pointers (c-core/viterbi.c:631-694)
    lu const lane = g.lane;
    store_u32_lane0(xnodes + l, lane, (pN << 0) | (pB << 4) | (pE << 6) | (pC << 21) | (pT << 25) | (pJ << 26));
    lu w[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      lu const k = lane * (uint32_t)Q + (uint32_t)q;
      lu word = pM[q];
      word = word | lselu(lequ(k, lu_splat(0)), lu_splat(0), pD[q] << 5);              // node 0 has no D
      word = word | lselu(llt_u(k + 1u, lu_splat((uint32_t)K)), pI[q] << 6, lu_splat(0)); // node K-1 has no I
      w[q] = word;
    }
    store_nodes_q<Q>(nodes + (size_t)l * (size_t)K, K, lane, w);

    // row l replaces row l-5 in the ring
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      M[P][q] = Ma[q];
      I[P][q] = Ia[q];
      D[P][q] = Da[q];
    }
    g.put_last(GS_M, Ma[Q - 1]);
    g.put_last(GS_I, Ia[Q - 1]);
    g.put_last(GS_D, Da[Q - 1]);
    g.sync();
    Msh[P] = g.get_shift(GS_M, Ma[Q - 1], DCP_INF);
    Ish[P] = g.get_shift(GS_I, Ia[Q - 1], DCP_INF);
    Dsh[P] = g.get_shift(GS_D, Da[Q - 1], DCP_INF);
    S[P] = DCP_INF;
    N[P] = Na;
    B[P] = Ba;
    J[P] = Ja;
    E[P] = Ea;
    C[P] = Ca;
    return Ta;
  }

  // returns T of the last row (the path pass's own score; equals viterbi_cost)
  DCP_FN float run(int L)
  {
    float T = DCP_INF;
    if (L > 0)
    {
      cr = codes[1];
      fetch(2, L);
    }
    int l = 1;
    for (; l + 4 <= L; l += 5)
    {
      T = row<1>(l, L);
      T = row<2>(l + 1, L);
      T = row<3>(l + 2, L);
      T = row<4>(l + 3, L);
      T = row<0>(l + 4, L);
    }
    if (l <= L) T = row<1>(l++, L);
    if (l <= L) T = row<2>(l++, L);
    if (l <= L) T = row<3>(l++, L);
    if (l <= L) T = row<4>(l++, L);
    return T;
  }
};
