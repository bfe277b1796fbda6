// dcp_types.h -- plain structs shared by host code, HIP kernels and the wave emulator.
#pragma once
#include <stdint.h>

#define DCP_TABLE_SIZE 1364  // 4+16+64+256+1024 quasi-codon codes, c-core/viterbi.c:13
#define DCP_NUM_TRANS 8      // BM MM MI MD IM II DM DD, c-core/viterbi.h:22-32
#define DCP_NUM_XTRANS 13    // RR SN NN SB NB EB JB EJ JJ EC CC ET CT, c-core/viterbi.h:4-19
#define DCP_XT_STRIDE 16     // floats per row of the special-transition table
#define DCP_REF_LANES 8      // SIMD width of the reference build whose E tie rule is reproduced (-mavx2)
#define DCP_MODEL_MAX 16384  // c-core/model.h:12

enum { DCP_BM, DCP_MM, DCP_MI, DCP_MD, DCP_IM, DCP_II, DCP_DM, DCP_DD };
enum { DCP_RR, DCP_SN, DCP_NN, DCP_SB, DCP_NB, DCP_EB, DCP_JB, DCP_EJ, DCP_JJ, DCP_EC, DCP_CC, DCP_ET, DCP_CT };

// One sequence position: codes of the 1..5-mers that END at this position,
// i.e. code_fn(pos - t, t) for t = 1..5 in c[t-1] (0 where the t-mer would start
// before the sequence).  32 bytes, one code per dword, so that a DP row costs one
// scalar dwordx8 load and no unpacking.
struct DcpCodeRow
{
  uint32_t c[8];
};

#define DCP_SP_STRIDE 8 // floats per row of special-state values in a DP table: N,B,J,E,C,0,0,0
#define DCP_ROW_HDR 4    // floats in front of every emission row: null[c], bg[c], 0, 0
#define DCP_MAX_STRIPS 8 // K > 4096: strips of 2048 positions (StripWave), up to 16384 padded positions

// A profile resident in HBM, in DP-parameter space (costs = -log-prob, +inf =
// impossible), padded to Kp = 64*Q*W positions with +inf.  All arrays live in
// one device pool of floats; the fields are offsets into it (in floats), so that
// kernels address them as kernel-argument base + scalar offset.
//   rows : 1364 emission rows, code-major, each DCP_ROW_HDR + Kp floats:
//          { null[c], bg[c], 0, 0, match[c][0..Kp) } -- everything a DP row needs
//          for one emission length sits behind ONE scalar offset c * stride
//   trans: [8][Kp]
struct DcpProfileDev
{
  int32_t K;         // core size
  int32_t Kp;        // padded positions
  int32_t Q;         // positions per lane
  int32_t W;         // waves per problem (1 for K <= 256)
  int64_t rows_off;  // [1364][DCP_ROW_HDR + Kp]
  int64_t trans_off; // [8][Kp]
  int64_t pad0, pad1;
};

// One (profile x sequence-window) DP problem.
struct DcpProblem
{
  int32_t profile;  // index into the DcpProfileDev array
  int32_t L;        // window length
  int64_t code_row; // index of the window's position 0 in the DcpCodeRow array (row l of the DP reads code_row + l)
  int32_t xt_row;   // row of the special-transition table = max(L/3, 1), c-core/thread.c:112
  int32_t out;      // slot in the output arrays
  int64_t trellis;  // path pass: offset (in bytes) of this problem's trellis in the arena
};

// Up to 16 windows of ONE profile that share a wavefront (viterbi_pack.h): group g of the wave runs window g.
// L[g] = 0 marks an idle group.  code_row is an index into the DcpCodeRow array (the array stays below 2^32 rows).
struct DcpPack
{
  int32_t profile;
  int32_t Lmax;          // the longest window of the pack: the row loop runs to it
  int32_t L[16];
  int32_t xt_row[16];    // max(L / 3, 1), c-core/thread.c:112
  int32_t out[16];       // slot in the output arrays
  uint32_t code_row[16]; // index of the window's position 0 in the DcpCodeRow array
};

// ---- fast path pass in blocks (checkpoint + recompute) -------------------------------------------------
// A window's DP table (12 B per cell) is never held whole.  A first pass (the cost kernel) leaves the FOLDED ring
// of five rows -- Mpre[5][Kp], Ipre[5][Kp], Spre[5] and X per lane -- every B rows (B a multiple of 5): checkpoint j is the
// state after row j * B.  Blocks are then taken from the last to the first: block j is recomputed from checkpoint
// j (block 0 from row 0) over rows j*B + 1 .. min(L, (j+1)*B + 5) into a table of B + 6 rows (row l at slot
// l - j*B), and the traceback walks the part of the path whose stage lies in (j*B + 5, (j+1)*B + 5] -- every
// row it reads there, stage - 5 .. stage, is in the block's table -- before handing over to block j - 1.
#ifdef __HIPCC__
#define DCP_HDI __host__ __device__ inline
#else
#define DCP_HDI inline
#endif
#define DCP_CKPT_ROWS_DEFAULT 500
#define DCP_CKPT_SP 6 // lane rows of specials per checkpoint: Spre[5], X
DCP_HDI long long dcp_ckpt_floats(int Kp, int W) { return 10LL * Kp + (long long)DCP_CKPT_SP * 64 * W; } // W waves per window

// blocks of a window of L rows with checkpoints every B rows (B = 0: one block, the whole window)
DCP_HDI int dcp_num_blocks(int L, int B) { return B <= 0 || L <= B + 5 ? 1 : (L - 5 + B - 1) / B; }

// rows a block's table holds, the row the block starts from included (row 0 for block 0)
DCP_HDI int dcp_block_slots(int L, int B) { return (B <= 0 || L <= B + 5 ? L : B + 5) + 1; }

// floats of one block's table: specials[slots][DCP_SP_STRIDE] + cells[slots][3][Kp]
DCP_HDI long long dcp_block_table_floats(int L, int Kp, int B)
{
  return (long long)dcp_block_slots(L, B) * (DCP_SP_STRIDE + 3LL * Kp);
}

// where the traceback of one window stands between blocks (all zero = not started)
struct DcpTraceState
{
  int32_t state;  // state id to visit next (c-core/state.h:9-25)
  int32_t stage;  // its DP row
  int32_t status; // 0 = under way, 1 = finished, < 0 = DCP_TB_*
  int32_t pad;
  int64_t n;      // steps written so far (from the end of the step buffer backwards)
};
