// viterbi_kernels.h -- launch interface of the HIP kernels (host side).
#pragma once
#include "dcp_types.h"
#include <hip/hip_runtime.h>

// kernel classes by padded size Kp (dcp_class_of / dcp_class_shape in viterbi_kernels.hip):
//   0..3: Kp = 64..256, one wavefront, 1..4 positions per lane
//   4..10: Kp = 384, 512, 768, 1024, 1536, 2048, 4096; cost kernels (6,1) (8,1) (6,2) (4,4) (6,4) (8,4) (8,8),
//          pass-by-pass path kernel (3,2) (4,2) (3,4) (4,4) (3,8) (4,8) (4,16)
//   11:    4096 < K <= 16383, strip by strip
#define DCP_NUM_CLASSES 12
#define DCP_STRIP_CLASS 11      // K > 4096: strips of DCP_STRIP_POSITIONS, state ring in HBM (StripWave)
#define DCP_STRIP_POSITIONS 2048 // 64 lanes x 4 positions x 8 wavefronts
#define DCP_MAX_CORE_SIZE 16383 // state ids keep 14 bits for k + 1 (c-core/state.h:27-39)
// per-problem ring scratch of the strip class: rest[5][Kp] + Ipre[5][Kp] floats at the largest Kp
#define DCP_RING_FLOATS ((size_t)10 * DCP_MAX_STRIPS * DCP_STRIP_POSITIONS)
#define DCP_RING_SLOTS 1024 // workgroups of the strip kernel in flight (each takes problems in turn)
int dcp_class_of(int K);                      // -1 when K is not covered
void dcp_class_shape(int cls, int *Q, int *W);

// Several windows per wavefront (viterbi_pack.h), cost pass of short profiles: shape i runs groups of S lanes
// with Q positions per lane, (S - 1) * Q positions at most -- (4,1) (4,2) (4,4) (8,2) (8,4) (16,2) (16,3) (16,4)
// (32,2) (32,3) (32,4): K <= 3, 6, 12, 14, 28, 30, 45, 60, 62, 93, 124.  The tables keep the layout of the class
// the profile belongs to.
#define DCP_NUM_PACK_SHAPES 11
int dcp_pack_shape_of(int K); // -1: no shape holds K
void dcp_pack_shape(int shape, int *Q, int *S);

struct DcpLaunch
{
  float const *pool;             // device: all profile arrays
  DcpProfileDev const *profiles; // device
  DcpProblem const *problems;    // device, all of one kernel class
  DcpCodeRow const *code_rows;   // device
  float const *xt_table;         // device, [rows][DCP_XT_STRIDE]
  float *out;                    // device: cost pass [2*slots] (null, alt); path pass [slots]
  unsigned char *arena;          // device: trellises (path pass only)
  float *ring = nullptr;         // device: strip class only, DCP_RING_SLOTS x DCP_RING_FLOATS
  int nprob;
  hipStream_t stream;
};

hipError_t dcp_launch_cost(int cls, DcpLaunch const &a);
// core sizes up to it have a cost kernel of their own on the class's layout (0: none): (5,1) below (6,1), (7,1) below
// (8,1), and K <= 640 as ONE wavefront of ten positions per lane instead of (6,2)'s two
int dcp_class_narrow_limit(int cls);
hipError_t dcp_launch_cost_narrow(int cls, DcpLaunch const &a); // cost pass of those windows
hipError_t dcp_launch_path(int cls, DcpLaunch const &a);
// trellis_unzip of every problem of a.problems (all classes): steps[step_off[out] .. step_off[out+1]) is the
// buffer of problem `out`; its steps end at the buffer's end
hipError_t dcp_launch_unzip(DcpLaunch const &a, uint32_t *steps, int64_t const *step_off, int32_t *nsteps);
// fast path pass: cost pass that stores each window's DP table at arena + problem.trellis, then the
// traceback of every problem of a.problems (all classes) into steps / nsteps (as dcp_launch_unzip)
// (in blocks, dcp_types.h: B rows between checkpoints, 0 = whole windows; ckpt_addr[out] = the window's checkpoints)
hipError_t dcp_launch_cost_ckpt(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B);
// G = 0: block `block` of every window.  G > 0: launch `it` of the groups of G blocks -- block nb - 1 - (it * G + g) of
// every window, g = 0 .. G - 1, into table g of the window's G block tables (dcp_block_table_floats apart), one
// workgroup per (window, g); the traceback then walks those blocks, the highest first
hipError_t dcp_launch_cost_store(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B, int block, int G = 0, int it = 0);
hipError_t dcp_launch_traceback(DcpLaunch const &a, uint32_t *steps, int64_t const *step_off, int32_t *nsteps,
                                DcpTraceState *states, int B, int block, int G = 0, int it = 0);
// the same for every window of a.problems (one class, not the strip class) in ONE launch: a workgroup takes its window
// through the checkpoints, then block by block through rows + traceback (DcpProblem::trellis = the table's address)
hipError_t dcp_launch_path_blocks(int cls, DcpLaunch const &a, int64_t const *ckpt_addr, int B, uint32_t *steps,
                                  int64_t const *step_off, int32_t *nsteps, DcpTraceState *states);
// strip class: the trellis replayed row by row from the DP tables at table_addr[out] (scratch: 3*K floats per row)
hipError_t dcp_launch_replay(DcpLaunch const &a, int64_t const *table_addr, int64_t const *scratch_addr, int max_rows);
hipError_t dcp_launch_compact_steps(uint32_t const *steps, int64_t const *step_off, int64_t const *compact_off,
                                    uint32_t *out, int n, hipStream_t stream);
// packs of one shape: one wavefront each; a.problems is not used
hipError_t dcp_launch_cost_pack(int shape, DcpLaunch const &a, DcpPack const *packs, int npack, uint32_t ncode_rows);
// lrt of every window from out[2n] = (null, alt); hits[0] = number of windows with a finite lrt >= 0 (zeroed by
// the caller), then (window, lrt bits) pairs, unordered
hipError_t dcp_launch_lrt_filter(float const *out, int n, uint32_t *hits, hipStream_t stream);
// the same with the table rows of the short emission lengths in LDS (all five lengths for groups of four lanes):
// workgroups of dcp_pack_lds_waves(shape) wavefronts, groups[i] = {first pack, number of packs (<= that many,
// all of one profile)}
int dcp_pack_lds_waves(int shape); // 0: the shape reads every row from global memory
hipError_t dcp_launch_cost_pack_lds(int shape, DcpLaunch const &a, DcpPack const *packs, int2 const *groups, int ngroups,
                                    uint32_t ncode_rows);
// every problem of classes 0..3 (single-wave) in one launch
hipError_t dcp_launch_cost_fused(DcpLaunch const &a);
hipError_t dcp_launch_encode(unsigned char const *nt, int64_t const *seq_off, int64_t const *row_off, int nseq,
                             int64_t max_len, DcpCodeRow *rows, hipStream_t stream);
