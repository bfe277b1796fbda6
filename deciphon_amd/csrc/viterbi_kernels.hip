// viterbi_kernels.hip -- gfx950 kernels of the Viterbi scan path.
//
// One wavefront (= one 64-thread workgroup) per (profile x window) problem.
// Problems of one launch share the positions-per-lane count Q (K <= 64*Q).
#include "lane_ops_gpu.h"
#include "viterbi_body.h"
#include "viterbi_kernels.h"

template <int Q>
__global__ __launch_bounds__(64) void dcp_cost_kernel(float const *__restrict__ pool,
                                                      DcpProfileDev const *__restrict__ profiles,
                                                      DcpProblem const *__restrict__ problems,
                                                      uint4 const *__restrict__ code_rows,
                                                      float const *__restrict__ xt_table,
                                                      float *__restrict__ out, int nprob)
{
  int const p = (int)blockIdx.x;
  if (p >= nprob) return;
  DcpProblem const pb = problems[p];
  DcpProfileDev const pf = profiles[pb.profile];
  CostWave<Q> w;
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE);
  w.run(pb.L, out + 2 * (size_t)pb.out);
}

template <int Q>
__global__ __launch_bounds__(64) void dcp_path_kernel(float const *__restrict__ pool,
                                                      DcpProfileDev const *__restrict__ profiles,
                                                      DcpProblem const *__restrict__ problems,
                                                      uint4 const *__restrict__ code_rows,
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
  PathWave<Q> w;
  w.init(pool, pf, code_rows + pb.code_row, xt_table + (size_t)pb.xt_row * DCP_XT_STRIDE, xnodes, nodes);
  float const T = w.run(pb.L);
  store_f32_lane0(out + pb.out, w.lane, T);
}

// Code rows of one encoded sequence: row r (1..n) holds the codes of the
// 1..5-mers covering positions r-t..r-1 (imm_eseq_get, third-party imm;
// off[t] + sum idx*4^(t-1-i), A,C,G,T = 0..3).
__global__ void dcp_encode_kernel(unsigned char const *__restrict__ nt, int64_t const *__restrict__ seq_off,
                                  int64_t const *__restrict__ row_off, int nseq, DcpCodeRow *__restrict__ rows)
{
  int const s = (int)blockIdx.y;
  if (s >= nseq) return;
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
      cr.c[t - 1] = ok ? (uint16_t)(off[t - 1] + idx) : (uint16_t)0;
    }
    cr.c[5] = cr.c[6] = cr.c[7] = 0;
    out[r] = cr;
  }
}

template <int Q> static hipError_t launch_cost_q(DcpLaunch const &a)
{
  hipLaunchKernelGGL(dcp_cost_kernel<Q>, dim3((unsigned)a.nprob), dim3(64), 0, a.stream, a.pool, a.profiles, a.problems,
                     reinterpret_cast<uint4 const *>(a.code_rows), a.xt_table, a.out, a.nprob);
  return hipGetLastError();
}

template <int Q> static hipError_t launch_path_q(DcpLaunch const &a)
{
  hipLaunchKernelGGL(dcp_path_kernel<Q>, dim3((unsigned)a.nprob), dim3(64), 0, a.stream, a.pool, a.profiles, a.problems,
                     reinterpret_cast<uint4 const *>(a.code_rows), a.xt_table, a.arena, a.out, a.nprob);
  return hipGetLastError();
}

hipError_t dcp_launch_cost(int Q, DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (Q)
  {
  case 1: return launch_cost_q<1>(a);
  case 2: return launch_cost_q<2>(a);
  case 3: return launch_cost_q<3>(a);
  case 4: return launch_cost_q<4>(a);
  default: return hipErrorInvalidValue;
  }
}

hipError_t dcp_launch_path(int Q, DcpLaunch const &a)
{
  if (a.nprob <= 0) return hipSuccess;
  switch (Q)
  {
  case 1: return launch_path_q<1>(a);
  case 2: return launch_path_q<2>(a);
  case 3: return launch_path_q<3>(a);
  case 4: return launch_path_q<4>(a);
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
  hipLaunchKernelGGL(dcp_encode_kernel, dim3(bx, (unsigned)nseq), dim3(256), 0, stream, nt, seq_off, row_off, nseq,
                     rows);
  return hipGetLastError();
}
