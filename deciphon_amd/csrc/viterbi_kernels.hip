// viterbi_kernels.hip -- gfx950 kernels of the Viterbi scan path.
//
// One workgroup per (profile x window) problem: a single wavefront for K <= 256
// (Q = 1..4 positions per lane), W = 2..16 wavefronts of Q = 4 beyond that.
// Problems of one launch share the kernel class (Q, W), K <= 64*Q*W.
#include <stdlib.h>

#include "lane_ops_gpu.h"
#include "viterbi_body.h"
#include "viterbi_pack.h"
#include "traceback.h"
#include "viterbi_kernels.h"
#include "row_replay.h"

// A device address that reaches a kernel as an INTEGER (the table and checkpoint addresses in DcpProblem::trellis and
// ckpt_addr[]) would make every access through it a flat_* instruction: the compiler cannot know the address space,
// flat accesses may complete out of order, and each use of a loaded value then waits for ALL outstanding memory
// operations -- the emission prefetch of the next row and the stores of this one included (vmcnt(0) twice per row).
// Going through an address_space(1) pointer tells it the memory is global: global_load / global_store, counted waits.
template <class T> __device__ __forceinline__ T *dcp_global(uintptr_t address)
{
  return (T *)(__attribute__((address_space(1))) T *)address;
}

// Workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2 (MI355X_MICROARCH.md, Workgroup
// dispatch): with the plain blockIdx -> problem mapping the windows of one profile (neighbours in the sorted
// problem list) land on all eight L2s and every L2 holds the tables of every profile in flight.  This gives
// XCD x the x-th contiguous eighth of the list instead (the bijective form for any grid size), so an L2 sees
// an eighth of the profiles.  A speed choice only: nothing depends on where a workgroup runs.
__device__ __forceinline__ int dcp_xcd_remap_any(int b, int n)
{
  int const q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ int dcp_xcd_remap(int b, int n)
{
  // a launch of about one generation of wavefronts keeps the plain order: there the eighths would be uneven in
  // time (windows of different profiles take different time) with nothing left to even them out
  return n < 16384 ? b : dcp_xcd_remap_any(b, n);
}
// The path pass's kernels take the eighths whatever the size of the launch: all their wavefronts are resident at once
// (nothing to even out), and what they wait for is memory -- the 2301 hit windows of the headline scan fetch 140 GB of
// emission rows and write 66 GB of tables in 50 ms (scripts/pmc_path.sh).  With the windows of a profile on one XCD
// instead of all eight, the pass takes 46.5 ms instead of 51 (profiles/r03_path_pass_pmc.txt).

// Wavefronts per SIMD the register allocator must leave room for.  A wavefront issues a VALU every ~7.5 cycles
// whatever its instruction-level parallelism (profiles/r02_valu_rates.txt), so two wavefronts per SIMD cap the
// issue rate at 640 G/s and three at 729.  Holding a shape to fewer registers by the bound alone spills into the row
// loop (Q = 3 / 4 to 96 / 128 VGPRs: 6-7 scratch reloads per row, 881 -> 620 GCUPS at K = 173, 1000 -> 878 at K = 256;
// (6,1), (7,1) to 168: 26-55 per row -- profiles/r02_exp_trans_stash_occupancy.txt).  Asking for the next row's
// emissions later in the row (DCP_COST_POLICY bit 1, viterbi_body.h) frees Q registers per emission length at the
// peak instead: (4,1) 147 -> 128 VGPRs = four wavefronts per SIMD instead of three (K = 256: 1102 -> 1199 GCUPS),
// (5,1) fits 168 with nothing spilled inside the loop (K = 300: 961 -> 1006), and ten positions per lane fit one
// wavefront at all ((10,1), 226 VGPRs: K = 513..640 run 32-38 % faster than as two wavefronts of five).  Where the
// wavefronts per SIMD stay what they were it brings nothing ((3,1), (6,1), (7,1): profiles/r03_exp_register_policy.txt).
#ifndef DCP_COST_WAVES
#define DCP_COST_WAVES(Q, W) ((W) == 1 && (Q) == 4 ? 4 : (W) == 1 && (Q) == 5 ? 3 : (Q) >= 8 ? 2 : 1)
#endif
template <int Q, int W, int POLICY = DCP_COST_POLICY(Q, W), int WAVES = DCP_COST_WAVES(Q, W)>
__global__ __launch_bounds__(64 * W, WAVES) void dcp_cost_kernel(float const *__restrict__ pool,
                                                      DcpProfileDev const *__restrict__ profiles,
                                                      DcpProblem const *__restrict__ problems,
                                                      DcpCodeRow const *__restrict__ code_rows,
                                                      float const *__restrict__ xt_table,
                                                      float *__restrict__ out, int nprob)
{
  if ((int)blockIdx.x >= nprob) return;
#ifdef DCP_EXP_LDSPAD // timing experiment (profiles/r02_exp_*): LDS nobody uses holds the wavefronts per SIMD down
  __shared__ float pad[DCP_EXP_LDSPAD / 4];
  if (nprob < 0) out[0] = pad[threadIdx.x];
#endif
  int const p = dcp_xcd_remap((int)blockIdx.x, nprob);
  DcpProblem const pb = problems[p];
  DcpProfileDev const pf = profiles[pb.profile];
  CostWave<Q, W, false, POLICY> w;
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE);
  w.run(pb.L, out + 2 * (size_t)pb.out);
}

// Fast path pass in blocks (dcp_types.h).  First the checkpoints: the cost pass once more over the hit windows,
// leaving the folded ring of five rows every B rows (windows of a single block need none and leave at once).
template <int Q, int W>
__global__ __launch_bounds__(64 * W, (Q >= 8 ? 2 : 1)) void dcp_cost_ckpt_kernel(float const *__restrict__ pool,
                                                           DcpProfileDev const *__restrict__ profiles,
                                                           DcpProblem const *__restrict__ problems,
                                                           DcpCodeRow const *__restrict__ code_rows,
                                                           float const *__restrict__ xt_table,
                                                           int64_t const *__restrict__ ckpt_addr, int B,
                                                           float *__restrict__ out, int nprob)
{
  if ((int)blockIdx.x >= nprob) return;
  DcpProblem const pb = problems[dcp_xcd_remap_any((int)blockIdx.x, nprob)];
  if (dcp_num_blocks(pb.L, B) <= 1) return;
  DcpProfileDev const pf = profiles[pb.profile];
  CostWave<Q, W> w;
  w.ckpt_out = dcp_global<float>((uintptr_t)ckpt_addr[pb.out]);
  w.ckpt_every = B;
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE);
  w.run(pb.L, out + 2 * (size_t)pb.out);
}

// Then, block by block from the last to the first: the rows of block `block` of every window that has one,
// recomputed from its checkpoint into the window's table -- float specials[slots][8] followed by float
// cells[slots][3][Kp], slots = rows of a block + 1, row l at slot l - block * B.  B = 0: the whole window is one
// block and the table holds all its rows (what the trellis replay of the strip class reads).
// G > 0: G blocks of every window side by side, one workgroup each -- the blocks of a window are independent once its
// checkpoints exist, and a lone wavefront per window leaves the GPU empty and waits out every row's latency by itself
// (profiles/r03_scan_pipeline.txt).  Workgroup b takes window b / G and, in launch `it`, its block
// nb - 1 - (it * G + b % G) (the last blocks first), into table b % G of the window's G tables.
template <int Q, int W>
__global__ __launch_bounds__(64 * W, (Q >= 8 ? 2 : 1)) void dcp_cost_store_kernel(float const *__restrict__ pool,
                                                            DcpProfileDev const *__restrict__ profiles,
                                                            DcpProblem const *__restrict__ problems,
                                                            DcpCodeRow const *__restrict__ code_rows,
                                                            float const *__restrict__ xt_table,
                                                            unsigned char *__restrict__ arena,
                                                            int64_t const *__restrict__ ckpt_addr, int B, int block,
                                                            int G, int it, float *__restrict__ out, int nprob)
{
  int p, sub = 0;
  if (G > 0)
  {
    if ((int)blockIdx.x >= nprob * G) return;
    int const b = dcp_xcd_remap_any((int)blockIdx.x, nprob * G); // (window, block) pairs in eighths: whole windows, mostly
    p = b / G;
    sub = b % G;
  }
  else
  {
    if ((int)blockIdx.x >= nprob) return;
    p = dcp_xcd_remap_any((int)blockIdx.x, nprob);
  }
  DcpProblem const pb = problems[p];
  if (G > 0) block = dcp_num_blocks(pb.L, B) - 1 - (it * G + sub);
  if (block < 0 || block >= dcp_num_blocks(pb.L, B)) return;
  DcpProfileDev const pf = profiles[pb.profile];
  CostWave<Q, W, true> w;
  int const slots = dcp_block_slots(pb.L, B);
  // integer arithmetic: the engine passes arena = 0 and absolute table addresses in pb.trellis
  w.tab_sp = dcp_global<float>((uintptr_t)arena + (uintptr_t)pb.trellis) + (size_t)sub * dcp_block_table_floats(pb.L, pf.Kp, B);
  w.tab_cells = w.tab_sp + (size_t)slots * DCP_SP_STRIDE;
  w.row_base = block * B;
  if (block > 0)
    w.ckpt_in = dcp_global<float const>((uintptr_t)ckpt_addr[pb.out]) + (size_t)(block - 1) * (size_t)dcp_ckpt_floats(pf.Kp, W);
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE);
  int const last = B > 0 ? (block + 1) * B + 5 : pb.L;
  w.run(pb.L, out + 2 * (size_t)pb.out, last < pb.L ? last : pb.L);
}

// Profiles beyond 4096 positions: one workgroup walks each row strip by strip (StripWave).
template <int Q, int W, bool STORE>
__global__ __launch_bounds__(64 * W) void dcp_strip_kernel(float const *__restrict__ pool,
                                                       DcpProfileDev const *__restrict__ profiles,
                                                       DcpProblem const *__restrict__ problems,
                                                       DcpCodeRow const *__restrict__ code_rows,
                                                       float const *__restrict__ xt_table,
                                                       unsigned char *__restrict__ arena, float *__restrict__ ring,
                                                       float *__restrict__ out, int nprob)
{
  // the grid is at most DCP_RING_SLOTS workgroups, each owning one ring and taking problems in turn
  for (int p = (int)blockIdx.x; p < nprob; p += (int)gridDim.x)
  {
    DcpProblem const pb = problems[p];
    DcpProfileDev const pf = profiles[pb.profile];
    StripWave<Q, W, STORE> w;
    w.ring = ring + (size_t)blockIdx.x * DCP_RING_FLOATS;
    if (STORE)
    {
      w.tab_sp = dcp_global<float>((uintptr_t)arena + (uintptr_t)pb.trellis);
      w.tab_cells = w.tab_sp + (size_t)(pb.L + 1) * DCP_SP_STRIDE;
    }
    w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE);
    w.run(pb.L, out + 2 * (size_t)pb.out);
    __syncthreads(); // the next problem re-initialises the LDS records
  }
}

// one row of one window's trellis
__device__ void dcp_replay_one(float const *__restrict__ pool, DcpProfileDev const *__restrict__ profiles,
                               DcpProblem const pb, DcpCodeRow const *__restrict__ code_rows,
                               float const *__restrict__ xt_table, unsigned char *__restrict__ arena,
                               int64_t const *__restrict__ table_addr, int64_t const *__restrict__ scratch_addr,
                               float *__restrict__ out)
{
  int const l = (int)(blockIdx.x * 64u + threadIdx.x);
  if (l > pb.L) return;
  DcpProfileDev const pf = profiles[pb.profile];
  uint32_t *xnodes = reinterpret_cast<uint32_t *>(arena + pb.trellis);
  uint16_t *nodes = reinterpret_cast<uint16_t *>(xnodes + (pb.L + 1)) + (size_t)l * pf.K;
  if (l == 0) // before(): every field 0 (c-core/viterbi.c:602-629)
  {
    xnodes[0] = 0;
    for (int k = 0; k < pf.K; ++k) nodes[k] = 0;
    return;
  }
  DcpTraceIn in;
  in.K = pf.K;
  in.Kp = pf.Kp;
  in.L = pb.L;
  in.sp = dcp_global<float const>((uintptr_t)table_addr[pb.out]);
  in.cells = in.sp + (size_t)(pb.L + 1) * DCP_SP_STRIDE;
  in.rows = pool + pf.rows_off;
  in.trans = pool + pf.trans_off;
  in.codes = code_rows + pb.code_row;
  in.xt = xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE;
  float *acc = dcp_global<float>((uintptr_t)scratch_addr[pb.out]) + (size_t)l * 3 * pf.K;
  dcp_replay_row(in, l, acc, xnodes + l, nodes);
  if (l == pb.L) // T of the last row: the score viterbi_path returns (c-core/viterbi.c:585-586,599)
    out[pb.out] = __builtin_fminf(in.sp[(size_t)l * DCP_SP_STRIDE + 3] + in.xt[DCP_ET],
                                  in.sp[(size_t)l * DCP_SP_STRIDE + 4] + in.xt[DCP_CT]);
}

// The pass-by-pass trellis of profiles beyond 4096 positions: every row replayed from the DP table
// by one thread (row_replay.h).  blockIdx.y (strided: the y extent of a grid stops at 65535) = problem,
// blockIdx.x * 64 + threadIdx.x = row.
__global__ __launch_bounds__(64) void dcp_replay_kernel(
    float const *__restrict__ pool, DcpProfileDev const *__restrict__ profiles, DcpProblem const *__restrict__ problems,
    DcpCodeRow const *__restrict__ code_rows, float const *__restrict__ xt_table, unsigned char *__restrict__ arena,
    int64_t const *__restrict__ table_addr, int64_t const *__restrict__ scratch_addr, float *__restrict__ out, int nprob)
{
  for (int p = (int)blockIdx.y; p < nprob; p += (int)gridDim.y)
    dcp_replay_one(pool, profiles, problems[p], code_rows, xt_table, arena, table_addr, scratch_addr, out);
}

hipError_t dcp_launch_replay(DcpLaunch const &a, int64_t const *table_addr, int64_t const *scratch_addr, int max_rows)
{
  if (a.nprob <= 0) return hipSuccess;
  hipLaunchKernelGGL(dcp_replay_kernel, dim3((unsigned)((max_rows + 63) / 64), (unsigned)(a.nprob < 65535 ? a.nprob : 65535)), dim3(64), 0, a.stream,
                     a.pool, a.profiles, a.problems, a.code_rows, a.xt_table, a.arena, table_addr, scratch_addr, a.out,
                     a.nprob);
  return hipGetLastError();
}

// Fast path pass, step 2: one WAVEFRONT walks one window's DP table back from T to S.
// Same decisions as the scalar dcp_traceback() of traceback.h (which documents them and
// is what the CPU tests exercise); here the candidates of the visited state are spread
// over the lanes in the reference's order -- lane j = (5 - t) * names + name -- so that
// one step costs two load round trips instead of a chain of them, and the first
// candidate equal to the stored value is the lowest set bit of a ballot.
__device__ int dcp_traceback_wave(DcpTraceIn const &in, uint32_t *buf, int64_t cap, DcpTraceState *st)
{
  enum
  {
    ST_M = 0 << 14, ST_I = 1 << 14, ST_D = 2 << 14, ST_X = 3 << 14,
    ST_S = ST_X | 3, ST_N = ST_X | 4, ST_B = ST_X | 5, ST_E = ST_X | 6, ST_J = ST_X | 7, ST_C = ST_X | 8, ST_T = ST_X | 9,
  };
  float const INF = __builtin_inff();
  int const lane = (int)(threadIdx.x & 63);
  int const K = in.K, Kp = in.Kp;
  size_t const stride = (size_t)Kp + DCP_ROW_HDR;
  int const base = in.row_base;
  auto SP = [&](int l, int i) { return in.sp[(size_t)(l - base) * DCP_SP_STRIDE + i]; };
  auto CELL = [&](int l, int s, int k) { return k < 0 ? INF : in.cells[((size_t)(l - base) * 3 + s) * (size_t)Kp + k]; };
  auto TR = [&](int id, int k) { return in.trans[(size_t)id * Kp + k]; };
  float const *xt = in.xt;

  int state = ST_T, stage = in.L;
  int64_t n = 0;
  if (st && st->state != 0) // resume where the block after this one stopped (uniform: every lane reads the same)
  {
    state = st->state;
    stage = st->stage;
    n = st->n;
  }
  while (state != ST_S || stage)
  {
    if (stage <= in.lo) // the rest of the path lies in the block before this one
    {
      if (lane == 0)
      {
        st->state = state;
        st->stage = stage;
        st->n = n;
      }
      return 0;
    }
    int size = 0, prev = -1;
    DcpCodeRow const cr = in.codes[stage];
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
        if (!(target < INF)) return DCP_TB_BAD;
        float const t_in = state == ST_N ? xt[DCP_SN] : state == ST_J ? xt[DCP_EJ] : xt[DCP_EC];
        float const t_self = state == ST_N ? xt[DCP_NN] : state == ST_J ? xt[DCP_JJ] : xt[DCP_CC];
        int const t = 5 - (lane >> 1), which = lane & 1;
        bool hit = false;
        if (lane < 10 && t <= stage)
        {
          int const z = stage - t;
          float const nil = in.rows[(size_t)cr.c[t - 1] * stride];
          float const from = state == ST_N ? (z == 0 ? 0.0f : INF) : SP(z, 3);
          float const cand = which == 0 ? (from + t_in) + nil : (SP(z, self) + t_self) + nil;
          hit = cand == target;
        }
        unsigned long long const mask = __ballot(hit);
        if (!mask) return DCP_TB_BAD;
        int const j = __ffsll((long long)mask) - 1;
        size = 5 - (j >> 1);
        prev = (j & 1) ? state : (state == ST_N ? ST_S : ST_E);
      }
      else if (state == ST_B)
      {
        if (stage == 0) prev = ST_S;
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
        float const target = SP(stage, 3);
        if (!(target < INF)) return DCP_TB_BAD;
        int Qr = (K - 1) / DCP_REF_LANES + 1;
        if (Qr < 2) Qr = 2;
        int key = -1; // (reference lane << 16) | (65535 - k): highest lane, then lowest k
        bool dtie = false;
        for (int k = lane; k < K; k += 64)
        {
          dtie = dtie || CELL(stage, 2, k) == target;
          if (CELL(stage, 0, k) == target)
          {
            int const cand = ((k / Qr) << 16) | (65535 - k);
            key = cand > key ? cand : key;
          }
        }
        if (__ballot(dtie)) return DCP_TB_TIE;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
        {
          int const o = __shfl_xor(key, d);
          key = o > key ? o : key;
        }
        if (key < 0) return DCP_TB_BAD;
        prev = ST_M | ((65535 - (key & 0xffff)) + 1);
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
        int const t = 5 - (lane >> 2), name = lane & 3; // BM, MM, IM, DM
        bool hit = false;
        if (lane < 20 && t <= stage)
        {
          int const z = stage - t;
          float const m = in.rows[(size_t)cr.c[t - 1] * stride + DCP_ROW_HDR + k];
          float const x = name == 0 ? SP(z, 1) : CELL(z, name - 1, k - 1);
          float const tr = TR(name == 0 ? DCP_BM : name == 1 ? DCP_MM : name == 2 ? DCP_IM : DCP_DM, k);
          hit = (x + tr) + m == target;
        }
        unsigned long long const mask = __ballot(hit);
        if (!mask) return DCP_TB_BAD;
        int const j = __ffsll((long long)mask) - 1;
        size = 5 - (j >> 2);
        int const nm = j & 3;
        prev = nm == 0 ? ST_B : (nm == 1 ? ST_M : nm == 2 ? ST_I : ST_D) | k;
      }
      else if (kind == ST_I)
      {
        float const target = CELL(stage, 1, k);
        if (!(target < INF)) return DCP_TB_BAD;
        int const t = 5 - (lane >> 1), name = lane & 1; // II, MI
        bool hit = false;
        if (lane < 10 && t <= stage)
        {
          int const z = stage - t;
          float const bg = in.rows[(size_t)cr.c[t - 1] * stride + 1];
          float const x = name == 0 ? CELL(z, 1, k) : CELL(z, 0, k);
          float const tr = TR(name == 0 ? DCP_II : DCP_MI, k);
          hit = (x + tr) + bg == target;
        }
        unsigned long long const mask = __ballot(hit);
        if (!mask) return DCP_TB_BAD;
        int const j = __ffsll((long long)mask) - 1;
        size = 5 - (j >> 1);
        prev = ((j & 1) ? ST_M : ST_I) | (k + 1);
      }
      else
      {
        float const a = CELL(stage, 0, k - 1) + TR(DCP_MD, k), b = CELL(stage, 2, k - 1) + TR(DCP_DD, k);
        if (a == b) return a < INF ? DCP_TB_TIE : DCP_TB_BAD;
        prev = (a < b ? ST_M : ST_D) | k;
      }
    }
    if (n + 1 >= cap) return DCP_TB_OVERFLOW;
    if (lane == 0) buf[cap - 1 - n] = (uint32_t)state | ((uint32_t)size << 16);
    ++n;
    state = prev;
    stage -= size;
    if (stage < 0) return DCP_TB_BAD;
  }
  if (n >= cap) return DCP_TB_OVERFLOW;
  if (lane == 0) buf[cap - 1 - n] = (uint32_t)state;
  return (int)(n + 1);
}

__global__ __launch_bounds__(64) void dcp_traceback_kernel(
    float const *__restrict__ pool, DcpProfileDev const *__restrict__ profiles, DcpProblem const *__restrict__ problems,
    DcpCodeRow const *__restrict__ code_rows, float const *__restrict__ xt_table,
    unsigned char const *__restrict__ arena, uint32_t *__restrict__ steps, int64_t const *__restrict__ step_off,
    int32_t *__restrict__ nsteps, DcpTraceState *__restrict__ states, int B, int block, int G, int it, int nprob)
{
  int const p = (int)blockIdx.x;
  if (p >= nprob) return;
  DcpProblem const pb = problems[p];
  int const nb = dcp_num_blocks(pb.L, B);
  DcpTraceState *st = states + pb.out;
  if (st->status != 0) return; // finished, or given up, in a later block
  DcpProfileDev const pf = profiles[pb.profile];
  DcpTraceIn in;
  in.K = pf.K;
  in.Kp = pf.Kp;
  in.L = pb.L;
  in.rows = pool + pf.rows_off;
  in.trans = pool + pf.trans_off;
  in.codes = code_rows + pb.code_row;
  in.xt = xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE;
  float const *tables = dcp_global<float const>((uintptr_t)arena + (uintptr_t)pb.trellis);
  // G > 0: through the (up to) G blocks launch `it` of dcp_cost_store_kernel has just written, the highest first
  for (int sub = 0; sub < (G > 0 ? G : 1); ++sub)
  {
    int const b = G > 0 ? nb - 1 - (it * G + sub) : block;
    if (b < 0 || b >= nb) return;
    in.sp = tables + (size_t)sub * dcp_block_table_floats(pb.L, pf.Kp, B);
    in.cells = in.sp + (size_t)dcp_block_slots(pb.L, B) * DCP_SP_STRIDE;
    in.row_base = b * B;
    in.lo = b > 0 ? b * B + 5 : -1;
    int const r = dcp_traceback_wave(in, steps + step_off[pb.out], step_off[pb.out + 1] - step_off[pb.out], st);
    if (r != 0)
    {
      if ((threadIdx.x & 63) == 0)
      {
        st->status = r > 0 ? 1 : r;
        nsteps[pb.out] = r;
      }
      return;
    }
  }
}

// The fast path pass of ONE window from start to end in one workgroup (what dcp_cost_ckpt_kernel, then per block
// dcp_cost_store_kernel + dcp_traceback_kernel do in 1 + 2 x blocks launches): the checkpoints, then block by block
// from the last to the first the rows of the block into the window's table and the traceback through them by the
// workgroup's first wavefront.  Windows of one launch no longer wait for each other between blocks -- a window of
// two blocks is done after two -- and nothing returns to the host in between.  The table rows a traceback reads
// were written by wavefronts of its own workgroup: the workgroup barrier orders them (one CU, one vector L1).
template <int Q, int W>
__global__ __launch_bounds__(64 * W, (Q >= 8 ? 2 : 1)) void dcp_path_blocks_kernel(
    float const *__restrict__ pool, DcpProfileDev const *__restrict__ profiles, DcpProblem const *__restrict__ problems,
    DcpCodeRow const *__restrict__ code_rows, float const *__restrict__ xt_table, int64_t const *__restrict__ ckpt_addr,
    int B, float *__restrict__ out, uint32_t *__restrict__ steps, int64_t const *__restrict__ step_off,
    int32_t *__restrict__ nsteps, DcpTraceState *__restrict__ states, int nprob)
{
  if ((int)blockIdx.x >= nprob) return;
  __shared__ int walk_over;
  DcpProblem const pb = problems[dcp_xcd_remap_any((int)blockIdx.x, nprob)];
  DcpProfileDev const pf = profiles[pb.profile];
  DcpCodeRow const *codes = code_rows + pb.code_row;
  float const *xt = xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE;
  int const nb = dcp_num_blocks(pb.L, B);
  float *ckpt = nb > 1 ? dcp_global<float>((uintptr_t)ckpt_addr[pb.out]) : nullptr;
  if (nb > 1)
  {
    CostWave<Q, W> w;
    w.ckpt_out = ckpt;
    w.ckpt_every = B;
    w.init(pool, pf, codes, xt);
    // the path pass runs beside the cost kernels of the batches in flight (dcp_scan_run): its few wavefronts are bound
    // by latency, so they go first wherever they share a SIMD -- what they take from the others is a few per cent
    wave_priority<3>();
    w.run(pb.L, out + 2 * (size_t)pb.out);
  }
  int const slots = dcp_block_slots(pb.L, B);
  float *tab_sp = dcp_global<float>((uintptr_t)pb.trellis);
  for (int block = nb - 1; block >= 0; --block)
  {
    __syncthreads(); // the checkpoints are written; the walk through the block above has left the table
    {
      CostWave<Q, W, true> w;
      w.tab_sp = tab_sp;
      w.tab_cells = tab_sp + (size_t)slots * DCP_SP_STRIDE;
      w.row_base = block * B;
      if (block > 0) w.ckpt_in = ckpt + (size_t)(block - 1) * (size_t)dcp_ckpt_floats(pf.Kp, W);
      w.init(pool, pf, codes, xt);
      wave_priority<3>();
      int const last = B > 0 ? (block + 1) * B + 5 : pb.L;
      w.run(pb.L, out + 2 * (size_t)pb.out, last < pb.L ? last : pb.L);
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64)
    {
      DcpTraceState *st = states + pb.out;
      DcpTraceIn in;
      in.K = pf.K;
      in.Kp = pf.Kp;
      in.L = pb.L;
      in.sp = tab_sp;
      in.cells = tab_sp + (size_t)slots * DCP_SP_STRIDE;
      in.rows = pool + pf.rows_off;
      in.trans = pool + pf.trans_off;
      in.codes = codes;
      in.xt = xt;
      in.row_base = block * B;
      in.lo = block > 0 ? block * B + 5 : -1;
      int const r = dcp_traceback_wave(in, steps + step_off[pb.out], step_off[pb.out + 1] - step_off[pb.out], st);
      if (threadIdx.x == 0)
      {
        walk_over = r != 0;
        if (r != 0)
        {
          st->status = r > 0 ? 1 : r;
          nsteps[pb.out] = r;
        }
      }
    }
    __syncthreads();
    if (walk_over) break; // finished, or given up (a tie the values cannot resolve: the literal pass takes it)
  }
}

// All single-wave classes in one launch: small scans (a few thousand windows spread
// over several classes) would otherwise run their per-class kernels one after the
// other, each too small to fill 1024 SIMDs.  Costs the register budget of the
// largest class, so it is used only when the launch is small (engine.cpp).
__global__ __launch_bounds__(64) void dcp_cost_kernel_fused(float const *__restrict__ pool,
                                                            DcpProfileDev const *__restrict__ profiles,
                                                            DcpProblem const *__restrict__ problems,
                                                            DcpCodeRow const *__restrict__ code_rows,
                                                            float const *__restrict__ xt_table,
                                                            float *__restrict__ out, int nprob)
{
  if ((int)blockIdx.x >= nprob) return;
  int const p = dcp_xcd_remap((int)blockIdx.x, nprob);
  DcpProblem const pb = problems[p];
  DcpProfileDev const pf = profiles[pb.profile];
  DcpCodeRow const *codes = code_rows + pb.code_row;
  float const *xt = xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE;
  float *o = out + 2 * (size_t)pb.out;
  switch (pf.Q)
  {
  case 1: { CostWave<1, 1> w; w.init(pool, pf, codes, xt); w.run(pb.L, o); break; }
  case 2: { CostWave<2, 1> w; w.init(pool, pf, codes, xt); w.run(pb.L, o); break; }
  case 3: { CostWave<3, 1> w; w.init(pool, pf, codes, xt); w.run(pb.L, o); break; }
  default: { CostWave<4, 1> w; w.init(pool, pf, codes, xt); w.run(pb.L, o); break; }
  }
}

// Several windows of one profile per wavefront (viterbi_pack.h): one workgroup = one wavefront = one DcpPack.
// (four positions per lane with the next row's operands asked for late: 180 -> 155 VGPRs, three wavefronts per SIMD)
#ifndef DCP_PACK_WAVES
#define DCP_PACK_WAVES(Q) ((Q) == 4 ? 3 : 1)
#endif
template <int Q, int S, bool LATE = DCP_PACK_LATE(Q), int WAVES = DCP_PACK_WAVES(Q)>
__global__ __launch_bounds__(64, WAVES) void dcp_cost_pack_kernel(float const *__restrict__ pool,
                                                           DcpProfileDev const *__restrict__ profiles,
                                                           DcpPack const *__restrict__ packs,
                                                           DcpCodeRow const *__restrict__ code_rows, uint32_t ncode_rows,
                                                           float const *__restrict__ xt_table, float *__restrict__ out,
                                                           int npack)
{
  if ((int)blockIdx.x >= npack) return;
  DcpPack const &pk = packs[dcp_xcd_remap((int)blockIdx.x, npack)];
  DcpProfileDev const pf = profiles[pk.profile];
  PackWave<Q, S, dcp_lazy_turns(Q), 0, LATE> w;
  w.init(pool, pf, code_rows, ncode_rows, xt_table, pk);
  w.run(pk.Lmax, out, pk, xt_table);
}

// The same with the rows of the short emission lengths in LDS (viterbi_pack.h, NLDS): a workgroup of WG
// wavefronts = WG packs of ONE profile (groups[blockIdx] = first pack, number of packs) copies the first
// DCP_PACK_LDS_ROWS(NLDS) rows of the profile's table -- header and position columns -- once, and every
// wavefront gathers those operands from there.
template <int Q, int S, int WG, int NLDS, bool LATE = DCP_PACK_LATE(Q)>
__global__ __launch_bounds__(64 * WG) void dcp_cost_pack_lds_kernel(float const *__restrict__ pool,
                                                                   DcpProfileDev const *__restrict__ profiles,
                                                                   DcpPack const *__restrict__ packs,
                                                                   int2 const *__restrict__ groups,
                                                                   DcpCodeRow const *__restrict__ code_rows,
                                                                   uint32_t ncode_rows, float const *__restrict__ xt_table,
                                                                   float *__restrict__ out, int ngroups)
{
  constexpr int RL = DCP_PACK_LDS_ROW(Q, S), NR = DCP_PACK_LDS_ROWS(NLDS);
  __shared__ __attribute__((aligned(16))) float table[NR * RL];
  if ((int)blockIdx.x >= ngroups) return;
  int2 const grp = groups[blockIdx.x];
  DcpProfileDev const pf = profiles[packs[grp.x].profile];
  float const *__restrict__ rows = pool + pf.rows_off;
  int const stride = pf.Kp + DCP_ROW_HDR;
  for (int i = (int)threadIdx.x; i < NR * RL; i += 64 * WG)
  {
    int const c = i / RL, j = i - c * RL;
    table[i] = j < stride ? rows[(size_t)c * stride + j] : __builtin_inff();
  }
  __syncthreads();
  int const wave = (int)(threadIdx.x >> 6);
  if (wave >= grp.y) return; // no barrier follows
  DcpPack const &pk = packs[grp.x + wave];
  PackWave<Q, S, dcp_lazy_turns(Q), NLDS, LATE> w;
  w.init(pool, pf, code_rows, ncode_rows, xt_table, pk, (lds_float const *)table);
  w.run(pk.Lmax, out, pk, xt_table);
}

template <int Q, int W>
__global__ __launch_bounds__(64 * W) void dcp_path_kernel(float const *__restrict__ pool,
                                                      DcpProfileDev const *__restrict__ profiles,
                                                      DcpProblem const *__restrict__ problems,
                                                      DcpCodeRow const *__restrict__ code_rows,
                                                      float const *__restrict__ xt_table,
                                                      unsigned char *__restrict__ arena,
                                                      float *__restrict__ out, int nprob)
{
  int const p = (int)blockIdx.x;
  if (p >= nprob) return;
  DcpProblem const pb = problems[p];
  DcpProfileDev const pf = profiles[pb.profile];
  // trellis of a problem: uint32 xnodes[L+1] then uint16 nodes[(L+1)*K]
  uint32_t *xnodes = reinterpret_cast<uint32_t *>(arena + pb.trellis);
  uint16_t *nodes = reinterpret_cast<uint16_t *>(xnodes + (pb.L + 1));
  PathWave<Q, W> w;
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE, xnodes, nodes);
  float const T = w.run(pb.L);
  store_f32_lane0(out + pb.out, w.g.lane, T);
}

// Code rows of one encoded sequence: row r (1..n) holds the codes of the
// 1..5-mers covering positions r-t..r-1 (imm_eseq_get, third-party imm;
// off[t] + sum idx*4^(t-1-i), A,C,G,T = 0..3).
__global__ void dcp_encode_kernel(unsigned char const *__restrict__ nt, int64_t const *__restrict__ seq_off,
                                  int64_t const *__restrict__ row_off, int nseq, DcpCodeRow *__restrict__ rows)
{
  // gridDim.y is capped at 65535 (a batch of metagenomic reads is larger): stride over the sequences
  for (int s = (int)blockIdx.y; s < nseq; s += (int)gridDim.y)
  {
    int64_t const n = seq_off[s + 1] - seq_off[s];
    unsigned char const *x = nt + seq_off[s];
    DcpCodeRow *out = rows + row_off[s];
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x)
    {
      DcpCodeRow cr;
      unsigned const off[5] = {0u, 4u, 20u, 84u, 340u};
      unsigned idx = 0;
#pragma unroll
      for (int t = 1; t <= 5; ++t)
      {
        // extend the t-mer to the left: new symbol is the most significant digit
        bool const ok = r - t >= 0;
        unsigned const sym = ok ? x[r - t] : 0u;
        idx += sym << (2 * (t - 1));
        cr.c[t - 1] = ok ? off[t - 1] + idx : 0u;
      }
      cr.c[5] = cr.c[6] = cr.c[7] = 0;
      out[r] = cr;
    }
  }
}

// trellis_unzip on the device (c-core/trellis.c:147-167 with previous_state and
// emission_size, :51-113): one thread walks one problem's trellis from T at stage L
// back to S at stage 0 and writes the steps, packed as state_id | seqsize << 16, from
// the END of its buffer backwards, so that they read forwards in path order.
// nsteps[p] = number of steps, or -1 when the buffer was too small / the trellis is
// inconsistent (the host then unzips that one itself).
__global__ void dcp_unzip_kernel(DcpProfileDev const *__restrict__ profiles, DcpProblem const *__restrict__ problems,
                                 unsigned char const *__restrict__ arena, uint32_t *__restrict__ steps,
                                 int64_t const *__restrict__ step_off, int32_t *__restrict__ nsteps, int nprob)
{
  int const p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (p >= nprob) return;
  DcpProblem const pb = problems[p];
  int const K = profiles[pb.profile].K;
  int const L = pb.L;
  uint32_t const *xnodes = reinterpret_cast<uint32_t const *>(arena + pb.trellis);
  uint16_t const *nodes = reinterpret_cast<uint16_t const *>(xnodes + (L + 1));
  uint32_t *buf = steps + step_off[pb.out];
  int64_t const cap = step_off[pb.out + 1] - step_off[pb.out];
  enum
  {
    ST_M = 0 << 14, ST_I = 1 << 14, ST_D = 2 << 14, ST_X = 3 << 14, // c-core/state.h:9-25
    ST_S = ST_X | 3, ST_N = ST_X | 4, ST_B = ST_X | 5, ST_E = ST_X | 6, ST_J = ST_X | 7, ST_C = ST_X | 8, ST_T = ST_X | 9,
  };
  int state = ST_T, stage = L;
  int64_t n = 0;
  bool bad = false;
  while ((state != ST_S || stage) && !bad)
  {
    int size = 0, prev = 0;
    if ((state & ST_X) == ST_X)
    {
      uint32_t const x = xnodes[stage];
      if (state == ST_N) { unsigned v = x & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_N : ST_S; }
      else if (state == ST_B) { unsigned v = (x >> 4) & 0x3; prev = v == 0 ? ST_S : v == 1 ? ST_N : v == 2 ? ST_E : ST_J; }
      else if (state == ST_E) { unsigned v = (x >> 6) & 0x7FFF; prev = (v & 1 ? ST_D : ST_M) | (int)(v / 2 + 1); }
      else if (state == ST_C) { unsigned v = (x >> 21) & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_C : ST_E; }
      else if (state == ST_T) { unsigned v = (x >> 25) & 0x1; prev = v ? ST_C : ST_E; }
      else if (state == ST_J) { unsigned v = (x >> 26) & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_J : ST_E; }
      else bad = true;
    }
    else
    {
      int const idx = (state & 0x3FFF) - 1;
      if (idx < 0 || idx >= K) { bad = true; break; }
      uint16_t const w = nodes[(size_t)stage * (size_t)K + (size_t)idx];
      int const kind = state & ST_X;
      if (kind == ST_M)
      {
        unsigned v = w & 0x1F;
        size = (int)(v % 5) + 1;
        unsigned s = v / 5;
        if (s == 0) prev = ST_B;
        else if (idx <= 0) bad = true;
        else prev = (s == 1 ? ST_M : s == 2 ? ST_I : ST_D) | idx;
      }
      else if (kind == ST_D)
      {
        unsigned v = (w >> 5) & 0x1;
        if (idx <= 0) bad = true;
        else prev = (v ? ST_D : ST_M) | idx;
      }
      else
      {
        unsigned v = (w >> 6) & 0xF;
        size = (int)(v % 5) + 1;
        prev = (v / 5 ? ST_I : ST_M) | (idx + 1);
      }
    }
    if (bad || n + 1 >= cap) { bad = true; break; }
    buf[cap - 1 - n] = (uint32_t)state | ((uint32_t)size << 16);
    ++n;
    state = prev;
    stage -= size;
    if (stage < 0) bad = true;
  }
  if (!bad && n < cap)
  {
    buf[cap - 1 - n] = (uint32_t)state; // the start state, no emission
    ++n;
  }
  else
    bad = true;
  nsteps[pb.out] = bad ? -1 : (int32_t)n;
}

// The filter of process_window (c-core/thread.c:118-121) on the device: lrt = -2 * (null - alt) of every window
// (c-core/lrt.h:6-9, the same fp32 operations), and the windows that go on to the path pass -- lrt finite and
// >= 0 -- appended to a list: hits[0] counts them, then (window, lrt bits) pairs, in no particular order (the
// host sorts the few that there are).
__global__ void dcp_lrt_filter_kernel(float const *__restrict__ out, int n, uint32_t *__restrict__ hits)
{
  int const i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  bool keep = false;
  float lrt = 0.0f;
  if (i < n)
  {
    float const null_loglik = -out[2 * (size_t)i], alt_loglik = -out[2 * (size_t)i + 1];
    lrt = -2 * (null_loglik - alt_loglik);
    keep = lrt >= 0.0f && lrt < __builtin_inff(); // finite (not NaN, not +inf) and not negative
  }
  unsigned long long const mask = __ballot(keep);
  if (!mask) return;
  int const lane = (int)(threadIdx.x & 63);
  uint32_t base = 0;
  if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(hits, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, __ffsll((long long)mask) - 1);
  if (keep)
  {
    uint32_t const at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    hits[1 + 2 * (size_t)at] = (uint32_t)i;
    hits[2 + 2 * (size_t)at] = __float_as_uint(lrt);
  }
}

hipError_t dcp_launch_lrt_filter(float const *out, int n, uint32_t *hits, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(dcp_lrt_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, hits);
  return hipGetLastError();
}

// Packs the steps of all windows back to back (window i: compact_off[i] .. compact_off[i+1])
// so that one small D2H copy carries every path: the per-window buffers are sized for the
// worst case and mostly empty.
__global__ void dcp_compact_steps_kernel(uint32_t const *__restrict__ steps, int64_t const *__restrict__ step_off,
                                         int64_t const *__restrict__ compact_off, uint32_t *__restrict__ out, int n)
{
  int const i = (int)blockIdx.x;
  if (i >= n) return;
  int64_t const count = compact_off[i + 1] - compact_off[i];
  uint32_t const *src = steps + step_off[i + 1] - count; // the steps end at the buffer's end
  uint32_t *dst = out + compact_off[i];
  for (int64_t j = threadIdx.x; j < count; j += blockDim.x) dst[j] = src[j];
}

hipError_t dcp_launch_compact_steps(uint32_t const *steps, int64_t const *step_off, int64_t const *compact_off,
                                    uint32_t *out, int n, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(dcp_compact_steps_kernel, dim3((unsigned)n), dim3(256), 0, stream, steps, step_off, compact_off,
                     out, n);
  return hipGetLastError();
}

hipError_t dcp_launch_unzip(DcpLaunch const &a, uint32_t *steps, int64_t const *step_off, int32_t *nsteps)
{
  if (a.nprob <= 0) return hipSuccess;
  unsigned const blocks = (unsigned)((a.nprob + 63) / 64);
  hipLaunchKernelGGL(dcp_unzip_kernel, dim3(blocks), dim3(64), 0, a.stream, a.profiles, a.problems, a.arena, steps,
                     step_off, nsteps, a.nprob);
  return hipGetLastError();
}

template <int Q, int W, int POLICY = DCP_COST_POLICY(Q, W), int WAVES = DCP_COST_WAVES(Q, W)>
static hipError_t launch_cost_qw(DcpLaunch const &a)
{
  hipLaunchKernelGGL((dcp_cost_kernel<Q, W, POLICY, WAVES>), dim3((unsigned)a.nprob), dim3(64 * W), 0, a.stream, a.pool,
                     a.profiles, a.problems, a.code_rows, a.xt_table, a.out, a.nprob);
  return hipGetLastError();
}

template <int Q, int W> static hipError_t launch_path_qw(DcpLaunch const &a)
{
  hipLaunchKernelGGL((dcp_path_kernel<Q, W>), dim3((unsigned)a.nprob), dim3(64 * W), 0, a.stream, a.pool, a.profiles,
                     a.problems, a.code_rows, a.xt_table, a.arena, a.out, a.nprob);
  return hipGetLastError();
}

template <bool STORE> static hipError_t launch_strip(DcpLaunch const &a)
{
  if (!a.ring) return hipErrorInvalidValue;
  unsigned const grid = (unsigned)(a.nprob < DCP_RING_SLOTS ? a.nprob : DCP_RING_SLOTS);
  hipLaunchKernelGGL((dcp_strip_kernel<4, 8, STORE>), dim3(grid), dim3(512), 0, a.stream, a.pool,
                     a.profiles, a.problems, a.code_rows, a.xt_table, a.arena, a.ring, a.out, a.nprob);
  return hipGetLastError();
}

int dcp_class_of(int K)
{
  if (K < 1) return -1;
  // classes 0..3: one wave, Q = 1..4.  K = 61..64 take the 128-column layout: the packed cost kernel that runs
  // them (32 lanes x 3 positions, viterbi_pack.h) reads 96 columns of a row
  if (K <= 256) return K > 60 && K <= 64 ? 1 : (K + 63) / 64 - 1;
  // Padded sizes 384, 512, 768, 1024, 1536, 2048, 4096.  The cost kernels run them with 6 or 8 positions per
  // lane -- (6,1) (8,1) (6,2) (4,4) (6,4) (8,4) (8,8) -- which halves or quarters the wavefronts that meet at
  // the row barrier; 8 per lane fits 256 VGPRs (two waves per SIMD) only with the transition arrays parked in
  // LDS between their uses (CostWave::STASH).  Measured against the 3/4-per-lane shapes on Pfam-structured
  // tables: K=384 639 -> 857 GCUPS, 512 674 -> 884, 768 517 -> 691, 1536 328 -> 593, 2048 427 -> 548,
  // 4096 190 -> 337; (8,2) for 1024 brought nothing over (4,4).  The pass-by-pass path kernel keeps 3/4.
  if (K <= 384) return 4;
  if (K <= 512) return 5;
  if (K <= 768) return 6;
  if (K <= 1024) return 7;
  if (K <= 1536) return 8;
  if (K <= 2048) return 9;
  if (K <= 4096) return 10;
  if (K <= DCP_MAX_CORE_SIZE) return DCP_STRIP_CLASS;
  return -1;
}

void dcp_class_shape(int cls, int *Q, int *W)
{
  // the shape of the cost / cost+store kernels; the pass-by-pass path kernel keeps at most 4 positions
  // per lane and runs the same padded layout as (3, 2W) / (4, 2W) where this says (6, W) / (8, W)
  static int const q[DCP_NUM_CLASSES] = {1, 2, 3, 4, 6, 8, 6, 4, 6, 8, 8, 4};
  static int const w[DCP_NUM_CLASSES] = {1, 1, 1, 1, 1, 1, 2, 4, 4, 4, 8, 8}; // the strip class: per strip
  *Q = q[cls];
  *W = w[cls];
}

hipError_t dcp_launch_cost(int cls, DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 0: return launch_cost_qw<1, 1>(a);
  case 1: return launch_cost_qw<2, 1>(a);
  case 2: return launch_cost_qw<3, 1>(a);
  case 3: return launch_cost_qw<4, 1>(a);
  case 4: return launch_cost_qw<6, 1>(a);
  case 5: return launch_cost_qw<8, 1>(a);
  case 6: return launch_cost_qw<6, 2>(a);
  case 7: return launch_cost_qw<4, 4>(a);
  case 8: return launch_cost_qw<6, 4>(a);
  case 9: return launch_cost_qw<8, 4>(a);
  case 10: return launch_cost_qw<8, 8>(a);
  case DCP_STRIP_CLASS: return launch_strip<false>(a);
  default: return hipErrorInvalidValue;
  }
}

// The classes whose rows are padded to 384, 512 and 768 columns: a profile that fits with one position per lane
// less -- K <= 320 in (5,1) instead of (6,1), K <= 448 in (7,1) instead of (8,1) -- runs that way on the same
// tables (it reads the first 64 Q W columns of a row): a sixth or an eighth fewer instructions per row.  K <= 640 on
// the 768-column layout runs as ONE wavefront of ten positions per lane, (10,1), instead of two of five: no barrier
// and no exchange per row, 325 instead of 2 x 242 VALU instructions (K = 520 / 576 / 640: 530 / 585 / 651 -> 715 /
// 810 / 857 GCUPS).  The engine sorts those windows to the front of their class.
int dcp_class_narrow_limit(int cls) { return cls == 4 ? 320 : cls == 5 ? 448 : cls == 6 ? 640 : 0; }

hipError_t dcp_launch_cost_narrow(int cls, DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 4: return launch_cost_qw<5, 1>(a);
  case 5: return launch_cost_qw<7, 1>(a);
  case 6: return launch_cost_qw<10, 1>(a); // one wavefront on the two-wave layout
  default: return hipErrorInvalidValue;
  }
}

template <int Q, int W>
static hipError_t launch_store_qw(DcpLaunch const &a, int64_t const *ckpt_addr, int B, int block, int G, int it)
{
  hipLaunchKernelGGL((dcp_cost_store_kernel<Q, W>), dim3((unsigned)a.nprob * (unsigned)(G > 0 ? G : 1)), dim3(64 * W), 0,
                     a.stream, a.pool, a.profiles, a.problems, a.code_rows, a.xt_table, a.arena, ckpt_addr, B, block, G, it,
                     a.out, a.nprob);
  return hipGetLastError();
}

hipError_t dcp_launch_cost_store(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B, int block, int G, int it)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 0: return launch_store_qw<1, 1>(a, ckpt_addr, B, block, G, it);
  case 1: return launch_store_qw<2, 1>(a, ckpt_addr, B, block, G, it);
  case 2: return launch_store_qw<3, 1>(a, ckpt_addr, B, block, G, it);
  case 3: return launch_store_qw<4, 1>(a, ckpt_addr, B, block, G, it);
  case 4: return launch_store_qw<6, 1>(a, ckpt_addr, B, block, G, it);
  case 5: return launch_store_qw<8, 1>(a, ckpt_addr, B, block, G, it);
  case 6: return launch_store_qw<6, 2>(a, ckpt_addr, B, block, G, it);
  case 7: return launch_store_qw<4, 4>(a, ckpt_addr, B, block, G, it);
  case 8: return launch_store_qw<6, 4>(a, ckpt_addr, B, block, G, it);
  case 9: return launch_store_qw<8, 4>(a, ckpt_addr, B, block, G, it);
  case 10: return launch_store_qw<8, 8>(a, ckpt_addr, B, block, G, it);
  case DCP_STRIP_CLASS: return B == 0 && block == 0 && G == 0 ? launch_strip<true>(a) : hipErrorInvalidValue; // whole tables only
  default: return hipErrorInvalidValue;
  }
}

template <int Q, int W> static hipError_t launch_ckpt_qw(DcpLaunch const &a, int64_t const *ckpt_addr, int B)
{
  hipLaunchKernelGGL((dcp_cost_ckpt_kernel<Q, W>), dim3((unsigned)a.nprob), dim3(64 * W), 0, a.stream, a.pool, a.profiles,
                     a.problems, a.code_rows, a.xt_table, ckpt_addr, B, a.out, a.nprob);
  return hipGetLastError();
}

hipError_t dcp_launch_cost_ckpt(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 0: return launch_ckpt_qw<1, 1>(a, ckpt_addr, B);
  case 1: return launch_ckpt_qw<2, 1>(a, ckpt_addr, B);
  case 2: return launch_ckpt_qw<3, 1>(a, ckpt_addr, B);
  case 3: return launch_ckpt_qw<4, 1>(a, ckpt_addr, B);
  case 4: return launch_ckpt_qw<6, 1>(a, ckpt_addr, B);
  case 5: return launch_ckpt_qw<8, 1>(a, ckpt_addr, B);
  case 6: return launch_ckpt_qw<6, 2>(a, ckpt_addr, B);
  case 7: return launch_ckpt_qw<4, 4>(a, ckpt_addr, B);
  case 8: return launch_ckpt_qw<6, 4>(a, ckpt_addr, B);
  case 9: return launch_ckpt_qw<8, 4>(a, ckpt_addr, B);
  case 10: return launch_ckpt_qw<8, 8>(a, ckpt_addr, B);
  default: return hipErrorInvalidValue;
  }
}

template <int Q, int W>
static hipError_t launch_path_blocks_qw(DcpLaunch const &a, int64_t const *ckpt_addr, int B, uint32_t *steps,
                                        int64_t const *step_off, int32_t *nsteps, DcpTraceState *states)
{
  hipLaunchKernelGGL((dcp_path_blocks_kernel<Q, W>), dim3((unsigned)a.nprob), dim3(64 * W), 0, a.stream, a.pool, a.profiles,
                     a.problems, a.code_rows, a.xt_table, ckpt_addr, B, a.out, steps, step_off, nsteps, states, a.nprob);
  return hipGetLastError();
}

hipError_t dcp_launch_path_blocks(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B, uint32_t *steps,
                                  int64_t const *step_off, int32_t *nsteps, DcpTraceState *states)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 0: return launch_path_blocks_qw<1, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 1: return launch_path_blocks_qw<2, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 2: return launch_path_blocks_qw<3, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 3: return launch_path_blocks_qw<4, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 4: return launch_path_blocks_qw<6, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 5: return launch_path_blocks_qw<8, 1>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 6: return launch_path_blocks_qw<6, 2>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 7: return launch_path_blocks_qw<4, 4>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 8: return launch_path_blocks_qw<6, 4>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 9: return launch_path_blocks_qw<8, 4>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  case 10: return launch_path_blocks_qw<8, 8>(a, ckpt_addr, B, steps, step_off, nsteps, states);
  default: return hipErrorInvalidValue;
  }
}

hipError_t dcp_launch_traceback(DcpLaunch const &a, uint32_t *steps, int64_t const *step_off, int32_t *nsteps,
                                DcpTraceState *states, int B, int block, int G, int it)
{
  if (a.nprob <= 0) return hipSuccess;
  hipLaunchKernelGGL(dcp_traceback_kernel, dim3((unsigned)a.nprob), dim3(64), 0, a.stream, a.pool, a.profiles, a.problems,
                     a.code_rows, a.xt_table, a.arena, steps, step_off, nsteps, states, B, block, G, it, a.nprob);
  return hipGetLastError();
}

// ---- several windows per wavefront: the shapes (lanes per group, positions per lane) by core size ----
static int const pack_S[DCP_NUM_PACK_SHAPES] = {4, 4, 4, 8, 8, 16, 16, 16, 32, 32, 32};
static int const pack_Q[DCP_NUM_PACK_SHAPES] = {1, 2, 4, 2, 4, 2, 3, 4, 2, 3, 4};

int dcp_pack_shape_of(int K)
{
  // DECIPHON_HIP_PACK_PREFER=<shape>: that shape for every profile it holds (throughput experiments)
  static int const prefer = getenv("DECIPHON_HIP_PACK_PREFER") ? atoi(getenv("DECIPHON_HIP_PACK_PREFER")) : -1;
  if (prefer >= 0 && prefer < DCP_NUM_PACK_SHAPES && K <= (pack_S[prefer] - 1) * pack_Q[prefer]) return prefer;
  for (int i = 0; i < DCP_NUM_PACK_SHAPES; ++i)
    if (K <= (pack_S[i] - 1) * pack_Q[i]) return i; // the first that holds it costs the fewest instructions per cell
  return -1;
}

void dcp_pack_shape(int shape, int *Q, int *S)
{
  *Q = pack_Q[shape];
  *S = pack_S[shape];
}

template <int Q, int S, bool LATE = DCP_PACK_LATE(Q), int WAVES = DCP_PACK_WAVES(Q)>
static hipError_t launch_pack_qs(DcpLaunch const &a, DcpPack const *packs, int npack, uint32_t ncode_rows)
{
  hipLaunchKernelGGL((dcp_cost_pack_kernel<Q, S, LATE, WAVES>), dim3((unsigned)npack), dim3(64), 0, a.stream, a.pool,
                     a.profiles, packs, a.code_rows, ncode_rows, a.xt_table, a.out, npack);
  return hipGetLastError();
}

hipError_t dcp_launch_cost_pack(int shape, DcpLaunch const &a, DcpPack const *packs, int npack, uint32_t ncode_rows)
{
  if (npack <= 0) return hipSuccess;
  switch (shape)
  {
  case 0: return launch_pack_qs<1, 4>(a, packs, npack, ncode_rows);
  case 1: return launch_pack_qs<2, 4>(a, packs, npack, ncode_rows);
  case 2: return launch_pack_qs<4, 4>(a, packs, npack, ncode_rows);
  case 3: return launch_pack_qs<2, 8>(a, packs, npack, ncode_rows);
  case 4: return launch_pack_qs<4, 8>(a, packs, npack, ncode_rows);
  case 5: return launch_pack_qs<2, 16>(a, packs, npack, ncode_rows);
  case 6: return launch_pack_qs<3, 16>(a, packs, npack, ncode_rows);
  case 7: return launch_pack_qs<4, 16>(a, packs, npack, ncode_rows);
  case 8: return launch_pack_qs<2, 32>(a, packs, npack, ncode_rows);
  case 9: return launch_pack_qs<3, 32>(a, packs, npack, ncode_rows);
  case 10: return launch_pack_qs<4, 32>(a, packs, npack, ncode_rows);
  default: return hipErrorInvalidValue;
  }
}

// wavefronts per workgroup of the LDS variants, by shape: as many as the registers let a CU hold.  Groups of 32
// lanes keep every row in L2: two rows per load are not what binds them, and the LDS variants measured 3-5 %
// slower there (K = 93: 711 against 745 GCUPS) while groups of 8 and 16 gained up to 27 % (K = 28: 580 -> 737).
static int const pack_lds_wg[DCP_NUM_PACK_SHAPES] = {16, 16, 12, 16, 12, 16, 12, 12, 0, 0, 0};
int dcp_pack_lds_waves(int shape) { return shape >= 0 && shape < DCP_NUM_PACK_SHAPES ? pack_lds_wg[shape] : 0; }

template <int Q, int S, int WG, int NLDS, bool LATE = DCP_PACK_LATE(Q)>
static hipError_t launch_pack_lds(DcpLaunch const &a, DcpPack const *packs, int2 const *groups, int ngroups, uint32_t ncode_rows)
{
  hipLaunchKernelGGL((dcp_cost_pack_lds_kernel<Q, S, WG, NLDS, LATE>), dim3((unsigned)ngroups), dim3(64 * WG), 0, a.stream,
                     a.pool, a.profiles, packs, groups, a.code_rows, ncode_rows, a.xt_table, a.out, ngroups);
  return hipGetLastError();
}

hipError_t dcp_launch_cost_pack_lds(int shape, DcpLaunch const &a, DcpPack const *packs, int2 const *groups, int ngroups,
                                    uint32_t ncode_rows)
{
  if (ngroups <= 0) return hipSuccess;
  switch (shape) // (Q, S, wavefronts, emission lengths in LDS): LDS bytes
  {
  case 0: return launch_pack_lds<1, 4, 16, 5>(a, packs, groups, ngroups, ncode_rows);  // 44 KB
  case 1: return launch_pack_lds<2, 4, 16, 5>(a, packs, groups, ngroups, ncode_rows);  // 65 KB
  case 2: return launch_pack_lds<4, 4, 12, 5>(a, packs, groups, ngroups, ncode_rows);  // 87 KB
  case 3: return launch_pack_lds<2, 8, 16, 4>(a, packs, groups, ngroups, ncode_rows);  // 27 KB
  case 4: return launch_pack_lds<4, 8, 12, 4>(a, packs, groups, ngroups, ncode_rows);  // 44 KB
  case 5: return launch_pack_lds<2, 16, 16, 4>(a, packs, groups, ngroups, ncode_rows); // 49 KB
  case 6: return launch_pack_lds<3, 16, 12, 4>(a, packs, groups, ngroups, ncode_rows); // 71 KB
  case 7: return launch_pack_lds<4, 16, 12, 4>(a, packs, groups, ngroups, ncode_rows);  // 87 KB
  case 8: return launch_pack_lds<2, 32, 16, 4>(a, packs, groups, ngroups, ncode_rows); // 92 KB
  case 9: return launch_pack_lds<3, 32, 12, 4>(a, packs, groups, ngroups, ncode_rows); // 136 KB
  case 10: return launch_pack_lds<4, 32, 8, 3>(a, packs, groups, ngroups, ncode_rows); // 43 KB: lengths 1..3 only
  default: return hipErrorInvalidValue;
  }
}

hipError_t dcp_launch_cost_fused(DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  hipLaunchKernelGGL(dcp_cost_kernel_fused, dim3((unsigned)a.nprob), dim3(64), 0, a.stream, a.pool, a.profiles,
                     a.problems, a.code_rows, a.xt_table, a.out, a.nprob);
  return hipGetLastError();
}

hipError_t dcp_launch_path(int cls, DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (cls)
  {
  case 0: return launch_path_qw<1, 1>(a);
  case 1: return launch_path_qw<2, 1>(a);
  case 2: return launch_path_qw<3, 1>(a);
  case 3: return launch_path_qw<4, 1>(a);
  case 4: return launch_path_qw<3, 2>(a);
  case 5: return launch_path_qw<4, 2>(a);
  case 6: return launch_path_qw<3, 4>(a);
  case 7: return launch_path_qw<4, 4>(a);
  case 8: return launch_path_qw<3, 8>(a);
  case 9: return launch_path_qw<4, 8>(a);
  case 10: return launch_path_qw<4, 16>(a);
  default: return hipErrorInvalidValue;
  }
}

hipError_t dcp_launch_encode(unsigned char const *nt, int64_t const *seq_off, int64_t const *row_off, int nseq,
                             int64_t max_len, DcpCodeRow *rows, hipStream_t stream)
{
  if (nseq <= 0) return hipSuccess;
  unsigned bx = (unsigned)((max_len + 1 + 255) / 256);
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(dcp_encode_kernel, dim3(bx, (unsigned)(nseq < 65535 ? nseq : 65535)), dim3(256), 0, stream, nt,
                     seq_off, row_off, nseq, rows);
  return hipGetLastError();
}
