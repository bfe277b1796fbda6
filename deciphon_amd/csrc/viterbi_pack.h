// viterbi_pack.h -- viterbi_null + viterbi_cost of SEVERAL windows per wavefront.
//
// CostWave (viterbi_body.h) gives one window a whole wavefront: a profile of K positions uses
// ceil(K / Q) of 64 lanes, and everything a DP row pays once per wavefront -- the chains of the
// special states, the E reduction, B, the k-1 shifts, the control of the lazy D->D loop -- is paid
// per window.  Here a wavefront is cut into G = 64 / S groups of S lanes and each group runs its
// own window of the SAME profile: the per-row work is paid once per G windows and short profiles
// fill the lanes (K = 3: 16 windows per wavefront instead of 3 busy lanes of 64).
//
//   lane = g * S + e.  e = 0 is the group's SEPARATOR: it owns no position and its transitions are
//   the +inf padding of the tables, so its M, I, D stay +inf whatever it adds to them -- which is
//   exactly what the k-1 neighbour of position 0 must be (shift() injects +inf there,
//   c-core/intrinsics.h:95-106).  Every k-1 shift is then one plain DPP wave_shr:1 for any S.
//   What the separator does load, where the other lanes load their match emissions, is the HEADER of
//   the emission row, { null[c], bg[c] }: the group takes its null and background emissions from it
//   (DPP quad broadcast to the special-state lanes, ds_swizzle broadcast to the whole group) and a DP
//   row costs five row loads and two code loads, nothing else.
//   e = 1 .. S-1 own positions k = (e - 1) * Q + q: a group holds K <= (S - 1) * Q.
//   The special states ride in lanes e = 0..3 of a separate register (N, J, C and the null
//   model R), as in CostWave.
//
// The recurrences, their fp32 association and the folded rows (Mpre, Ipre) are CostWave's: see the
// header of viterbi_body.h for why they give the reference's bits (c-core/viterbi.c:451-600,696-719).
// What changes is only where uniform values live: codes, null/bg emissions, special transitions and
// E, B are per GROUP, so they sit in VGPRs (loaded / reduced per lane) instead of SGPRs.  Windows of
// one wavefront may differ in length: the row loop runs to the longest, and a group captures its
// results at its own last row.
#pragma once
#include "dcp_types.h"

#ifndef DCP_FN
#error "include a lane_ops_*.h before viterbi_pack.h"
#endif

#ifndef DCP_INF
#error "include viterbi_body.h before viterbi_pack.h"
#endif
#ifndef DCP_SL
#define DCP_SL(P, t) (((P) + 5 - (t)) % 5)
#endif

// A load costs the vector cache a cycle or two per 128-B line it touches, whatever it uses of the line, and a
// pack touches G different table rows per load: the packed kernels are bound by lines, not by instructions
// (profiles/r02_exp_*).  So the rows of the SHORT emission lengths are read from an LDS copy the workgroup
// shares (dcp_cost_pack_lds_kernel) -- a gather there costs bank conflicts only.  NLDS = emission lengths
// served from LDS: the codes of lengths 1..NLDS are the first 4, 20, 84, 340, 1364 rows of the table, so
// NLDS = 5 is the whole table (groups of four lanes: 44-87 KB), NLDS = 4 leaves the 1024 five-mer rows in L2
// (340 rows: 27-136 KB for groups of 8..32 lanes), NLDS = 3 keeps 84 rows.  LDS rows hold the header and the
// (S - 1) Q position columns, DCP_PACK_LDS_ROW(Q, S) floats.
#define DCP_PACK_LDS_ROW(Q, S) ((DCP_ROW_HDR + ((S)-1) * (Q) + 3) / 4 * 4)
#define DCP_PACK_LDS_ROWS(NLDS) ((NLDS) == 5 ? 1364 : (NLDS) == 4 ? 340 : (NLDS) == 3 ? 84 : (NLDS) == 2 ? 20 : 4)

// TURNS = lazy D->D turns taken before the first vote (dcp_lazy_turns, viterbi_body.h)
// LATE: the next row's operands are asked for behind the D chain instead of before it (CostWave's POLICY bit 1).  With
// four positions per lane that is 180 -> 155 VGPRs, three wavefronts per SIMD instead of two: K = 124 913 -> 987 GCUPS,
// the LDS variants in workgroups of twelve wavefronts +4..7 % (K = 10..60); three positions per lane stay at three
// wavefronts (148 -> 141 VGPRs) and gain nothing.
#ifndef DCP_PACK_LATE
#define DCP_PACK_LATE(Q) ((Q) == 4)
#endif
template <int Q, int S, int TURNS = dcp_lazy_turns(Q), int NLDS = 0, bool LATE = DCP_PACK_LATE(Q)> struct PackWave
{
  static constexpr bool LDSTAB = NLDS > 0;
  static_assert(S == 4 || S == 8 || S == 16 || S == 32, "groups of 4, 8, 16 or 32 lanes");
  enum { G = 64 / S, CAP = (S - 1) * Q };
  // Q = 3, 4: the six transition arrays the fold uses once per row wait in LDS (6 KB per wavefront) instead of
  // 6Q registers: 148 VGPRs with three positions per lane (three wavefronts per SIMD), 180 with four -- 155 and a
  // third wavefront per SIMD once the next row's operands are requested late (LATE).  MD and DD (the D chain and its
  // turns) stay in registers.
#ifndef DCP_PACK_STASH
#define DCP_PACK_STASH 1
#endif
  static constexpr bool STASH = DCP_PACK_STASH && Q >= 3 && Q <= 4;
  lf MD[Q], DD[Q];
  lf Mpre[5][Q], Ipre[5][Q], Spre[5];
  enum { EQ = Q < 2 ? 2 : Q }; // floats a lane loads per emission row: the separator needs two, { null, bg }
  lf em[5][EQ];
  lu code[5];           // codes of the next row to fetch (per group)
  lf sa, sb, nbjb;      // special transitions by special lane: Xpre = min(E + sa, X + sb); B candidates X + nbjb
  lf X, E;
  lf EBv;
  lf shM, shI, shD;     // destinations of the k-1 shifts (lane 0 of the wave stays +inf)
  lf Xs, Es;            // X and E of the group's own last row
  lu Lg;                // window length of the lane's group (0: idle group)
  lu crow;              // index of the group's code row 0
  PackSrc src;
  lds_float const *lds_rows = nullptr; // LDSTAB: the workgroup's copy of the emission table
  lu lds_off;                          // LDSTAB: float offset of the lane's operands inside a row

  lf kBM[STASH ? 1 : Q], kMM[STASH ? 1 : Q], kIM[STASH ? 1 : Q], kDM[STASH ? 1 : Q], kII[STASH ? 1 : Q], kMI[STASH ? 1 : Q];
  DCP_FN void set_fold_trans(lf const (&BM)[Q], lf const (&MM)[Q], lf const (&IM)[Q], lf const (&DM)[Q],
                             lf const (&II)[Q], lf const (&MI)[Q])
  {
    if constexpr (STASH)
    {
      pack_stash<Q>(0, BM);
      pack_stash<Q>(1, MM);
      pack_stash<Q>(2, IM);
      pack_stash<Q>(3, DM);
      pack_stash<Q>(4, II);
      pack_stash<Q>(5, MI);
    }
    else
    {
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        kBM[q] = BM[q];
        kMM[q] = MM[q];
        kIM[q] = IM[q];
        kDM[q] = DM[q];
        kII[q] = II[q];
        kMI[q] = MI[q];
      }
    }
  }
  DCP_FN void get_fold_trans(PackFold &f, lf (&BM)[Q], lf (&MM)[Q], lf (&IM)[Q], lf (&DM)[Q], lf (&II)[Q], lf (&MI)[Q])
  {
    if constexpr (STASH)
      pack_unstash_wait<Q>(f, BM, MM, IM, DM, II, MI);
    else
    {
#pragma unroll
      for (int q = 0; q < Q; ++q)
      {
        BM[q] = kBM[q];
        MM[q] = kMM[q];
        IM[q] = kIM[q];
        DM[q] = kDM[q];
        II[q] = kII[q];
        MI[q] = kMI[q];
      }
    }
  }

  DCP_FN void fetch_codes(int l) { load_code_row(src, crow + (uint32_t)l, code); }

  // emissions (separator: null/bg) of the row whose codes sit in `code`
  DCP_FN void fetch_rows()
  {
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      if (t < NLDS) // the codes of emission length t + 1 are rows < DCP_PACK_LDS_ROWS(t + 1)
        load_lds_q<EQ>(lds_rows, code[t] * (uint32_t)DCP_PACK_LDS_ROW(Q, S) + lds_off, em[t]);
      else
        load_pack_q<EQ>(src, code[t], em[t]);
    }
  }

  DCP_FN void init(float const *__restrict__ pool, DcpProfileDev const &pf, DcpCodeRow const *__restrict__ code_rows,
                   uint32_t ncode_rows, float const *__restrict__ xt_table, DcpPack const &pk,
                   lds_float const *lds_table = nullptr)
  {
    lds_rows = lds_table;
    lu const lane = lane_ids();
    lu const e = lane & lu_splat((uint32_t)(S - 1)); // lane in its group
    lu const g = lane_shr(lane, S == 4 ? 2 : S == 8 ? 3 : S == 16 ? 4 : 5);
    int const Kp = pf.Kp;
    // column of the lane's first position; the separator takes its transitions from the last Q columns of the
    // padded arrays (+inf) and reads the header of every emission row (byte offset 0)
    lm const sep = lequ(e, lu_splat(0));
    lu const col = lselu(sep, lu_splat((uint32_t)(Kp - Q)), (e - lu_splat(1)) * (uint32_t)Q);
    src = packsrc_make(pool + pf.rows_off, Kp, code_rows, ncode_rows,
                       lselu(sep, lu_splat(0), (col + (uint32_t)DCP_ROW_HDR) * 4u));
    lds_off = lselu(sep, lu_splat(0), col + (uint32_t)DCP_ROW_HDR);
    float const *__restrict__ trans = pool + pf.trans_off;
    lf BM[Q], MM[Q], MI[Q], IM[Q], II[Q], DM[Q];
    load_cols<Q>(trans + DCP_BM * Kp, col, BM);
    load_cols<Q>(trans + DCP_MM * Kp, col, MM);
    load_cols<Q>(trans + DCP_MI * Kp, col, MI);
    load_cols<Q>(trans + DCP_MD * Kp, col, MD);
    load_cols<Q>(trans + DCP_IM * Kp, col, IM);
    load_cols<Q>(trans + DCP_II * Kp, col, II);
    load_cols<Q>(trans + DCP_DM * Kp, col, DM);
    load_cols<Q>(trans + DCP_DD * Kp, col, DD);
    set_fold_trans(BM, MM, IM, DM, II, MI);
    Lg = load_u32_at(reinterpret_cast<uint32_t const *>(pk.L), g);
    crow = load_u32_at(pk.code_row, g);
    lu const xrow = load_u32_at(reinterpret_cast<uint32_t const *>(pk.xt_row), g) * (uint32_t)DCP_XT_STRIDE;
    lf const inf = lf_splat(DCP_INF);
    lm const l0 = lequ(e, lu_splat(0)), l1 = lequ(e, lu_splat(1)), l2 = lequ(e, lu_splat(2)), l3 = lequ(e, lu_splat(3));
    lf const NB = load_f32_at(xt_table, xrow + (uint32_t)DCP_NB), JB = load_f32_at(xt_table, xrow + (uint32_t)DCP_JB);
    lf const RR = load_f32_at(xt_table, xrow + (uint32_t)DCP_RR), SN = load_f32_at(xt_table, xrow + (uint32_t)DCP_SN);
    lf const SB = load_f32_at(xt_table, xrow + (uint32_t)DCP_SB);
    EBv = load_f32_at(xt_table, xrow + (uint32_t)DCP_EB);
    // e: 0 = N, 1 = J, 2 = C, 3 = R.  Xpre = min(E + sa, X + sb); B = min(E + EB, N + NB, J + JB)
    sa = lsel(l1, load_f32_at(xt_table, xrow + (uint32_t)DCP_EJ), lsel(l2, load_f32_at(xt_table, xrow + (uint32_t)DCP_EC), inf));
    sb = lsel(l0, load_f32_at(xt_table, xrow + (uint32_t)DCP_NN),
              lsel(l1, load_f32_at(xt_table, xrow + (uint32_t)DCP_JJ),
                   lsel(l2, load_f32_at(xt_table, xrow + (uint32_t)DCP_CC), lsel(l3, RR, inf))));
    nbjb = lsel(l0, NB, lsel(l1, JB, inf));
    shM = shI = shD = inf;
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
    for (int q = 0; q < Q; ++q) Mpre[0][q] = SB + BM[q];  // (BM still in registers here)
    Spre[0] = lsel(l0, lf_splat(0.0f) + SN, lsel(l3, lneg(RR) + RR, inf));
    X = lsel(l3, lneg(RR), inf);
    E = inf;
    Xs = X;
    Es = inf;
  }

  template <int P> DCP_FN void row(int l, int Lmax)
  {
    lf M[Q], I[Q], D[Q];
    lf bgv[5], hdr[5], sp[5], xs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      bgv[t] = group_bcast0<S>(em[t][1]); // every lane: bg[c_t] of its group
      hdr[t] = em[t][0];                  // the separator's: null[c_t] of the group
      sp[t] = Spre[DCP_SL(P, t + 1)];
    }
    add_quad0_x5(xs, sp, hdr); // lanes e = 0..3 (the special states) add null[c_t] of their group
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
    X = lmin3(lmin3(xs[4], xs[3], xs[2]), xs[1], xs[0]);

    // this row's operands are consumed: the next row's (its codes arrived a row ago) go out now,
    // together with the codes of the row after it
    if constexpr (!LATE)
    {
      if (l < Lmax)
      {
        fetch_rows();
        fetch_codes(l + 2);
      }
    }

    lf m = M[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) m = lmin(m, M[q]);
    lf const Msh0 = lane_shift_up_keep(M[Q - 1], shM);
    lf const Ish0 = lane_shift_up_keep(I[Q - 1], shI);
    E = group_min<S>(m);                                   // E_l = min_k M_l[k] (see viterbi_body.h)
    lf const B = lmin(E + EBv, group_min01<S>(X + nbjb)); // c-core/viterbi.c:495-496,582-583 (N, J: lanes 0, 1)

    PackFold fold;
    if constexpr (STASH) pack_unstash_issue(fold); // back from LDS while the D chain runs

    // D_l[k] = min(M_l[k-1] + MD[k], D_l[k-1] + DD[k]) (c-core/viterbi.c:538,553-580): serial inside a
    // lane, then carried across lanes until no lane improves (the reference's lazy loop, :569-580)
    D[0] = Msh0 + MD[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) D[q] = lmin(M[q - 1] + MD[q], D[q - 1] + DD[q]);
    lf Dsh0 = lane_shift_up_keep(D[Q - 1], shD);
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
    while (wave_any(llt(x, D[0])))
    {
      D[0] = lmin(D[0], x);
#pragma unroll
      for (int q = 1; q < Q; ++q) D[q] = lmin(D[q], D[q - 1] + DD[q]);
      Dsh0 = lane_shift_up_keep(D[Q - 1], shD);
      x = Dsh0 + DD[0];
    }

    if constexpr (LATE)
    {
      sched_fence();
      if (l < Lmax)
      {
        fetch_rows();
        fetch_codes(l + 2);
      }
      sched_fence();
    }
    // fold row l into the ring (slot P held row l-5, no longer needed)
    lf BM[Q], MM[Q], IM[Q], DM[Q], II[Q], MI[Q];
    get_fold_trans(fold, BM, MM, IM, DM, II, MI);
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      lf const Ml = q ? M[q ? q - 1 : 0] : Msh0;
      lf const Il = q ? I[q ? q - 1 : 0] : Ish0;
      lf const Dl = q ? D[q ? q - 1 : 0] : Dsh0;
      Mpre[P][q] = lmin3(B + BM[q], Ml + MM[q], lmin(Il + IM[q], Dl + DM[q]));
      Ipre[P][q] = lmin(I[q] + II[q], M[q] + MI[q]);
    }
    Spre[P] = lmin(E + sa, X + sb);
    // the group's own last row: keep what its results are made of
    lm const last = lequ(Lg, lu_splat((uint32_t)l));
    Xs = lsel(last, X, Xs);
    Es = lsel(last, E, Es);
  }

  // out[2 * slot] = viterbi_null(), out[2 * slot + 1] = viterbi_cost() of every group's window
  DCP_FN void run(int Lmax, float *__restrict__ out, DcpPack const &pk, float const *__restrict__ xt_table)
  {
    if (Lmax > 0)
    {
      fetch_codes(1);
      fetch_rows();
      fetch_codes(2);
    }
    int l = 1;
    for (; l + 4 <= Lmax; l += 5)
    {
      row<1>(l, Lmax);
      row<2>(l + 1, Lmax);
      row<3>(l + 2, Lmax);
      row<4>(l + 3, Lmax);
      row<0>(l + 4, Lmax);
    }
    if (l <= Lmax) row<1>(l++, Lmax);
    if (l <= Lmax) row<2>(l++, Lmax);
    if (l <= Lmax) row<3>(l++, Lmax);
    if (l <= Lmax) row<4>(l++, Lmax);
    // lane e = 2 holds C, e = 3 holds R of the group's last row (c-core/viterbi.c:585-586,599,718)
    lu const lane = lane_ids();
    lu const e = lane & lu_splat((uint32_t)(S - 1));
    lu const g = lane_shr(lane, S == 4 ? 2 : S == 8 ? 3 : S == 16 ? 4 : 5);
    lu const slot = load_u32_at(reinterpret_cast<uint32_t const *>(pk.out), g);
    lu const xrow = load_u32_at(reinterpret_cast<uint32_t const *>(pk.xt_row), g) * (uint32_t)DCP_XT_STRIDE;
    lm const active = llt_u(lu_splat(0), Lg);
    lf const T = lmin(Es + load_f32_at(xt_table, xrow + (uint32_t)DCP_ET), Xs + load_f32_at(xt_table, xrow + (uint32_t)DCP_CT));
    store_f32_where(out, slot * 2u + 1u, land(active, lequ(e, lu_splat(2))), T);
    store_f32_where(out, slot * 2u, land(active, lequ(e, lu_splat(3))), Xs);
  }
};
