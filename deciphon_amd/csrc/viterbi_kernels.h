// viterbi_kernels.h -- launch interface of the HIP kernels (host side).
#pragma once
#include "dcp_types.h"
#include <hip/hip_runtime.h>

#define DCP_MAX_Q 4 // single-wave kernels cover K <= 64 * DCP_MAX_Q

struct DcpLaunch
{
  float const *pool;             // device: all profile arrays
  DcpProfileDev const *profiles; // device
  DcpProblem const *problems;    // device, all with the same Q
  DcpCodeRow const *code_rows;   // device
  float const *xt_table;         // device, [rows][DCP_XT_STRIDE]
  float *out;                    // device: cost pass [2*slots] (null, alt); path pass [slots]
  unsigned char *arena;          // device: trellises (path pass only)
  int nprob;
  hipStream_t stream;
};

hipError_t dcp_launch_cost(int Q, DcpLaunch const &a);
hipError_t dcp_launch_path(int Q, DcpLaunch const &a);
hipError_t dcp_launch_encode(unsigned char const *nt, int64_t const *seq_off, int64_t const *row_off, int nseq,
                             int64_t max_len, DcpCodeRow *rows, hipStream_t stream);
