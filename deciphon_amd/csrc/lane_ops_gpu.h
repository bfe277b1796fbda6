// lane_ops_gpu.h -- per-lane vocabulary of the Viterbi kernels on gfx950 (wave64).
//
// viterbi_body.h is written against this small set of names so that the very
// same source can be instantiated a second time by tests/emul/ with 64-wide
// array types (a lock-step wave emulator used only to unit-test the kernel
// logic on a machine without a GPU).  Here every "lane" type is the plain
// scalar a HIP thread holds, and the cross-lane operations are DPP / readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "dcp_types.h"

#define DCP_FN __device__ __forceinline__
#define DCP_WAVE 64

typedef float lf;    // one fp32 per lane
typedef uint32_t lu; // one u32 per lane
typedef bool lm;     // one predicate per lane

DCP_FN lf lf_splat(float x) { return x; }
DCP_FN lu lu_splat(uint32_t x) { return x; }
DCP_FN lf lmin(lf a, lf b) { return __builtin_fminf(a, b); }
DCP_FN lf lmin3(lf a, lf b, lf c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
DCP_FN lm llt(lf a, lf b) { return a < b; }
DCP_FN lm llt_u(lu a, lu b) { return a < b; }
DCP_FN lm leq(lf a, lf b) { return a == b; }
DCP_FN lm lequ(lu a, lu b) { return a == b; }
DCP_FN lm land(lm a, lm b) { return a && b; }
DCP_FN lm lor(lm a, lm b) { return a || b; }
DCP_FN lm lnot(lm a) { return !a; }
DCP_FN lf lsel(lm m, lf a, lf b) { return m ? a : b; }
DCP_FN lu lselu(lm m, lu a, lu b) { return m ? a : b; }
DCP_FN lu lminu(lu a, lu b) { return a < b ? a : b; }
DCP_FN lu lmaxu(lu a, lu b) { return a > b ? a : b; }

// lane index inside the wave, 0..63
DCP_FN lu lane_ids() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// lane e receives x of lane e-1; lane 0 receives `fill` (DPP wave_shr:1, the
// 64-lane analogue of shift() in c-core/intrinsics.h:95-106)
DCP_FN lf lane_shift_up(lf x, float fill)
{
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}

// The same into a register that is kept across rows: wave_shr:1 never writes lane 0,
// so once lane 0 of `keep` holds the fill value it stays there and no constant has to
// be re-materialised per shift.
DCP_FN lf lane_shift_up_keep(lf x, lf &keep)
{
  keep = __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(x), 0x138, 0xf, 0xf, false));
  return keep;
}

// a uniform value the compiler must hold in a VGPR (so that VALU ops can pair it with
// an SGPR operand instead of copying the SGPR first)
// nothing is scheduled across this point
DCP_FN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
DCP_FN lf lf_pin(float x)
{
  lf v = x;
  asm volatile("" : "+v"(v));
  return v;
}

// min over the 64 lanes, returned to every lane (uniform).  Inside a row of 16 a butterfly -- quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror as DPP operands of v_min_f32, written with the builtin so that the
// compiler folds each step into one v_min_f32_dpp, places the wait states a DPP read after a VALU write needs
// and may fill them with other instructions of the row.  Across rows row_bcast:15 / :31, which must leave the
// other rows alone: the compiler does not fold that form, so those two stay assembly (it adds no wait states
// inside asm: the two a DPP needs after the VALU write of its source are written out).
#define DCP_DPP_MIN_ALL(v, ctrl) \
  __builtin_fminf((v), __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true)))
DCP_FN float wave_min(lf v)
{
  v = DCP_DPP_MIN_ALL(v, 0xB1);  // quad_perm:[1,0,3,2]
  v = DCP_DPP_MIN_ALL(v, 0x4E);  // quad_perm:[2,3,0,1]
  v = DCP_DPP_MIN_ALL(v, 0x141); // row_half_mirror
  v = DCP_DPP_MIN_ALL(v, 0x140); // row_mirror: every row holds its minimum in all lanes
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

DCP_FN uint32_t wave_minu(lu v)
{
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// issue priority of this wavefront among those of its SIMD (0..3; s_setprio)
template <int P> DCP_FN void wave_priority() { __builtin_amdgcn_s_setprio(P); }

DCP_FN bool wave_any(lm m) { return __builtin_amdgcn_ballot_w64(m) != 0ull; }
DCP_FN uint64_t wave_ballot(lm m) { return __builtin_amdgcn_ballot_w64(m); }
DCP_FN float read_lane(lf x, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
DCP_FN uint32_t read_laneu(lu x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, lane); }

// ---- the group of lanes that shares one DP problem ------------------------------
// W = 1: one wavefront; every exchange is DPP / readlane and the put_*/sync
// calls vanish.  W > 1: a workgroup of W wavefronts (K > 256); values that cross
// a wave boundary go through a few LDS words, published by put_*(), made
// visible by sync() (s_barrier) and consumed by the matching get_*().  A slot is
// never re-published before every wave has passed a later sync(), so one buffer
// per slot is enough.
enum { GS_M, GS_I, GS_D, GS_E, GS_F, GS_X, GS_T0, GS_T1, GS_X0, GS_X1, GS_SLOTS };

// Values parked in LDS between their uses (CostWave, POLICY bits 2 / 4): chunk j of a lane sits at
// ((wave*SLOTS + slot)*CHUNKS + j)*64 + lane chunks -- float4s, float2s or floats, the widest that divides Q -- so a
// wave reads 64 consecutive chunks.
template <int Q> struct DcpStashChunk
{
  static constexpr int N = Q % 4 == 0 ? 4 : Q % 2 == 0 ? 2 : 1; // floats per chunk
  typedef typename std::conditional<N == 4, float4, typename std::conditional<N == 2, float2, float>::type>::type type;
};
template <int Q, int W, int SLOTS> DCP_FN typename DcpStashChunk<Q>::type *dcp_stash_mem()
{
  __shared__ typename DcpStashChunk<Q>::type mem[W * SLOTS * (Q / DcpStashChunk<Q>::N) * 64];
  return mem;
}
template <int N> DCP_FN void dcp_chunk_get(typename std::conditional<N == 4, float4, typename std::conditional<N == 2, float2, float>::type>::type const &t, lf *v)
{
  if constexpr (N == 1)
    v[0] = t;
  else
  {
    v[0] = t.x;
    v[1] = t.y;
    if constexpr (N == 4)
    {
      v[2] = t.z;
      v[3] = t.w;
    }
  }
}
template <int Q, int W, int SLOTS> DCP_FN void dcp_stash(int wave, lu lane, int slot, lf const (&v)[Q])
{
  constexpr int N = DcpStashChunk<Q>::N;
  auto *mem = dcp_stash_mem<Q, W, SLOTS>() + (size_t)(wave * SLOTS + slot) * (Q / N) * 64 + (lane & 63u);
#pragma unroll
  for (int j = 0; j < Q / N; ++j)
  {
    if constexpr (N == 4)
      mem[j * 64] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    else if constexpr (N == 2)
      mem[j * 64] = make_float2(v[2 * j], v[2 * j + 1]);
    else
      mem[j * 64] = v[j];
  }
}
template <int Q, int W, int SLOTS> DCP_FN void dcp_unstash(int wave, lu lane, int slot, lf (&v)[Q])
{
  constexpr int N = DcpStashChunk<Q>::N;
  auto const *mem = dcp_stash_mem<Q, W, SLOTS>() + (size_t)(wave * SLOTS + slot) * (Q / N) * 64 + (lane & 63u);
#pragma unroll
  for (int j = 0; j < Q / N; ++j) dcp_chunk_get<N>(mem[j * 64], v + N * j);
}
// chunk j alone (DcpStashChunk<Q>::N positions): for users that walk an array once, a few positions at a time
template <int Q, int W, int SLOTS>
DCP_FN void dcp_unstash_chunk(int wave, lu lane, int slot, int j, lf (&v)[DcpStashChunk<Q>::N])
{
  constexpr int N = DcpStashChunk<Q>::N;
  dcp_chunk_get<N>(dcp_stash_mem<Q, W, SLOTS>()[((size_t)(wave * SLOTS + slot) * (Q / N) + j) * 64 + (lane & 63u)], v);
}

template <int W> struct Group;

template <> struct Group<1>
{
  lu lane; // 0..63
  DCP_FN void init() { lane = lane_ids(); }
  DCP_FN void put_last(int, lf) {}
  DCP_FN void put_min(int, lf) {}
  DCP_FN void put_minu(int, lu) {}
  DCP_FN void put_lanes4(int, lf) {}
  DCP_FN void put_any(int, lm) {}
  template <int Q> DCP_FN void put_tdd(lf const (&)[Q]) {}
  template <int Q, int SLOTS> DCP_FN void stash_q(int slot, lf const (&v)[Q]) { dcp_stash<Q, 1, SLOTS>(0, lane, slot, v); }
  template <int Q, int SLOTS> DCP_FN void unstash_q(int slot, lf (&v)[Q]) { dcp_unstash<Q, 1, SLOTS>(0, lane, slot, v); }
  template <int Q, int SLOTS> DCP_FN void unstash_chunk(int slot, int j, lf (&v)[DcpStashChunk<Q>::N])
  {
    dcp_unstash_chunk<Q, 1, SLOTS>(0, lane, slot, j, v);
  }
  DCP_FN void put_count(int, lm) {}
  DCP_FN void sync() {}
  DCP_FN lf get_shift(int, lf x, float fill) { return lane_shift_up(x, fill); }
  DCP_FN lf get_shift_keep(int, lf x, lf &keep) { return lane_shift_up_keep(x, keep); }
  DCP_FN float get_min(int, lf x) { return wave_min(x); }
  DCP_FN uint32_t get_minu(int, lu x) { return wave_minu(x); }
  DCP_FN float get_lane(int, lf x, int l) { return read_lane(x, l); }
  DCP_FN bool get_any(int, lm m) { return wave_any(m); }
  DCP_FN int get_count(int, lm m) { return __builtin_popcountll(wave_ballot(m)); }
};

template <int W> struct Group
{
  lu lane;        // 0..64*W-1, position of this lane in the group
  int wave;       // wave index inside the workgroup (uniform)
  float *lds;     // [GS_SLOTS][16] words
  float4 *rec;    // [2][17] per-row records {M, I, D of the wave's last position, min M} by row parity;
                  // entry 0 = all +inf (the neighbour of wave 0), wave w at entry w + 1
  float *tdd;     // [16] sum of DD over each wave's positions but its first
  float tddv[W <= 8 ? W : 1]; // the same in registers (W <= 8)
  DCP_FN void init()
  {
    static_assert(W <= 16, "a workgroup holds at most 16 wavefronts");
    __shared__ float scratch[GS_SLOTS * 16];
    __shared__ float4 records[2 * 17];
    __shared__ float through[16 + DCP_MAX_STRIPS * 16]; // [16] of CostWave, then [strip][16] of StripWave
    lds = scratch;
    rec = records;
    tdd = through;
    if (threadIdx.x < 2) records[threadIdx.x * 17] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
    wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    lane = lane_ids() + (uint32_t)wave * 64u;
  }
  DCP_FN bool last_lane() const { return (lane & 63u) == 63u; }
  DCP_FN void put_last(int slot, lf x)
  {
    if (last_lane()) lds[slot * 16 + wave] = x;
  }
  DCP_FN void put_min(int slot, lf x)
  {
    float const m = wave_min(x);
    if (last_lane()) lds[slot * 16 + wave] = m;
  }
  DCP_FN void put_minu(int slot, lu x)
  {
    uint32_t const m = wave_minu(x);
    if (last_lane()) lds[slot * 16 + wave] = __uint_as_float(m);
  }
  DCP_FN void put_lanes4(int slot, lf x) // lanes 0..3 of wave 0 hold the special states
  {
    if (lane < 4u) lds[slot * 16 + lane] = x;
  }
  DCP_FN void put_any(int slot, lm m)
  {
    bool const a = wave_any(m);
    if (last_lane()) lds[slot * 16 + wave] = a ? 1.0f : 0.0f;
  }
  DCP_FN void put_count(int slot, lm m)
  {
    int const c = __builtin_popcountll(wave_ballot(m));
    if (last_lane()) lds[slot * 16 + wave] = (float)c;
  }
  DCP_FN void sync() { __syncthreads(); }

  // ---- one-barrier rows of the cost pass (viterbi_body.h, CostWave::row) ----
  DCP_FN lf seg_shift_up(lf x, lf fill) // wave_shr:1, the wave's first lane takes `fill`
  {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
  }
  template <int Q, int SLOTS> DCP_FN void stash_q(int slot, lf const (&v)[Q]) { dcp_stash<Q, W, SLOTS>(wave, lane, slot, v); }
  template <int Q, int SLOTS> DCP_FN void unstash_q(int slot, lf (&v)[Q]) { dcp_unstash<Q, W, SLOTS>(wave, lane, slot, v); }
  DCP_FN bool seg_any(lm m) const { return wave_any(m); }
  DCP_FN lm seg_first() const { return (lane & 63u) == 0u; }
  template <int Q> DCP_FN void put_tdd(lf const (&DD)[Q])
  {
    float t = seg_first() ? 0.0f : DD[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) t += DD[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    if (last_lane()) tdd[wave] = t;
    __syncthreads(); // also publishes the +inf record of init()
    if constexpr (W <= 8)
    {
#pragma unroll
      for (int w = 0; w < W; ++w) tddv[w] = tdd[w];
    }
  }
  DCP_FN void put_rec(int par, lf m_last, lf i_last, lf d_last, lf m_all)
  {
    float const e = wave_min(m_all);
    if (last_lane()) rec[par * 17 + wave + 1] = make_float4(m_last, i_last, d_last, e);
  }
  // the previous wave's record (one b128 read; +inf before the first wave)
  DCP_FN void get_prev(int par, lf &Mp, lf &Ip, lf &Dp) const
  {
    float4 const r = rec[par * 17 + wave];
    Mp = r.x;
    Ip = r.y;
    Dp = r.z;
  }
  // E = min over the waves' minima; could anything entering a wave at its first lane still
  // lower the D it published?  (one b64 read per wave)
  DCP_FN void get_e_could(int par, float &E, bool &could) const
  {
    float const *r = reinterpret_cast<float const *>(rec + par * 17 + 1);
    if constexpr (W <= 8)
    {
      float d[W], e[W];
#pragma unroll
      for (int w = 0; w < W; ++w)
      {
        float2 const de = *reinterpret_cast<float2 const *>(r + 4 * w + 2);
        d[w] = de.x;
        e[w] = de.y;
      }
      float m = e[0];
#pragma unroll
      for (int w = 1; w < W; ++w) m = __builtin_fminf(m, e[w]);
      bool any = false;
#pragma unroll
      for (int w = 0; w < W; ++w)
      {
        float const s = m + tddv[w];
        any = any || __builtin_fminf(s * 0.9999f, s * 1.0001f) < d[w];
      }
      E = m;
      could = any;
    }
    else // 1024-thread workgroups are capped at 128 VGPRs: two passes over LDS instead of 2W registers
    {
      float m = r[3];
#pragma unroll
      for (int w = 1; w < W; ++w) m = __builtin_fminf(m, r[4 * w + 3]);
      bool any = false;
#pragma unroll
      for (int w = 0; w < W; ++w)
      {
        float const s = m + tdd[w];
        any = any || __builtin_fminf(s * 0.9999f, s * 1.0001f) < r[4 * w + 2];
      }
      E = m;
      could = any;
    }
  }
  // ---- StripWave (K > 4096): the same exchange with the previous strip in front of wave 0 ----
  template <int Q> DCP_FN void put_tdd_strip(int s, lf const (&DD)[Q])
  {
    float t = seg_first() ? 0.0f : DD[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) t += DD[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    if (last_lane()) tdd[16 + s * 16 + wave] = t;
  }
  DCP_FN void put_carry(int par, lf m, lf i, lf d)
  {
    if (wave == W - 1 && last_lane()) rec[par * 17] = make_float4(m, i, d, __builtin_inff());
  }
  DCP_FN void put_carry_inf(int par)
  {
    float const inf = __builtin_inff();
    if (wave == W - 1 && last_lane()) rec[par * 17] = make_float4(inf, inf, inf, inf);
  }
  // E of this strip; could anything entering a wave at its first lane (it costs at least
  // min(floor_e, E)) still lower the D that wave published?
  DCP_FN void get_e_could_row(int par, int s, float floor_e, float &E, bool &could) const
  {
    float const *r = reinterpret_cast<float const *>(rec + par * 17 + 1);
    float m = r[3];
#pragma unroll
    for (int w = 1; w < W; ++w) m = __builtin_fminf(m, r[4 * w + 3]);
    float const lo = __builtin_fminf(m, floor_e);
    bool any = false;
#pragma unroll
    for (int w = 0; w < W; ++w)
    {
      float const v = lo + tdd[16 + s * 16 + w];
      any = any || __builtin_fminf(v * 0.9999f, v * 1.0001f) < r[4 * w + 2];
    }
    E = m;
    could = any;
  }
  DCP_FN lf get_shift_carry(int slot, lf x, lf fill) { return get_shift(slot, x, fill); }

  // N and J of the special states (lanes 0, 1 of wave 0's X), one b64 read
  DCP_FN void get_nj(int slot, lf, float &N, float &J) const
  {
    float2 const v = *reinterpret_cast<float2 const *>(lds + slot * 16);
    N = v.x;
    J = v.y;
  }
  DCP_FN void note_fallback() const {}

  DCP_FN lf get_shift(int slot, lf x, float fill)
  {
    float const prev = wave > 0 ? lds[slot * 16 + wave - 1] : fill;
    return lane_shift_up(x, prev);
  }
  DCP_FN lf get_shift_keep(int slot, lf x, lf &) { return get_shift(slot, x, __builtin_inff()); }
  DCP_FN float get_min(int slot, lf)
  {
    float m = lds[slot * 16];
#pragma unroll
    for (int w = 1; w < W; ++w) m = __builtin_fminf(m, lds[slot * 16 + w]);
    return m;
  }
  DCP_FN uint32_t get_minu(int slot, lu)
  {
    uint32_t m = __float_as_uint(lds[slot * 16]);
#pragma unroll
    for (int w = 1; w < W; ++w)
    {
      uint32_t const v = __float_as_uint(lds[slot * 16 + w]);
      m = v < m ? v : m;
    }
    return m;
  }
  DCP_FN float get_lane(int slot, lf, int l) { return lds[slot * 16 + l]; }
  DCP_FN bool get_any(int slot, lm)
  {
    float a = 0.0f;
#pragma unroll
    for (int w = 0; w < W; ++w) a += lds[slot * 16 + w];
    return a != 0.0f;
  }
  DCP_FN int get_count(int slot, lm)
  {
    float a = 0.0f;
#pragma unroll
    for (int w = 0; w < W; ++w) a += lds[slot * 16 + w];
    return (int)a;
  }
};

// ---- memory -------------------------------------------------------------------
// Each lane owns Q consecutive profile positions k = lane*Q + q; a padded row of
// Kp = 64*Q floats is therefore read as one coalesced dword x Q load per lane.
template <int Q> DCP_FN void load_q(float const *__restrict__ row, lu lane, lf (&out)[Q]);
template <> DCP_FN void load_q<1>(float const *__restrict__ row, lu lane, lf (&out)[1]) { out[0] = row[lane]; }
template <> DCP_FN void load_q<2>(float const *__restrict__ row, lu lane, lf (&out)[2])
{
  float2 v = reinterpret_cast<float2 const *>(row)[lane];
  out[0] = v.x;
  out[1] = v.y;
}
template <> DCP_FN void load_q<3>(float const *__restrict__ row, lu lane, lf (&out)[3])
{
  // 12-byte elements: dwordx3 (the struct keeps 4-byte alignment)
  struct f3 { float x, y, z; };
  f3 v = reinterpret_cast<f3 const *>(row)[lane];
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
}
template <> DCP_FN void load_q<4>(float const *__restrict__ row, lu lane, lf (&out)[4])
{
  float4 v = reinterpret_cast<float4 const *>(row)[lane];
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
  out[3] = v.w;
}

template <> DCP_FN void load_q<5>(float const *__restrict__ row, lu lane, lf (&out)[5])
{
  float const *p = row + (size_t)lane * 5; // 20-byte elements, 4-byte aligned
#pragma unroll
  for (int q = 0; q < 5; ++q) out[q] = p[q];
}
template <> DCP_FN void load_q<7>(float const *__restrict__ row, lu lane, lf (&out)[7])
{
  float const *p = row + (size_t)lane * 7;
#pragma unroll
  for (int q = 0; q < 7; ++q) out[q] = p[q];
}
template <> DCP_FN void load_q<6>(float const *__restrict__ row, lu lane, lf (&out)[6])
{
  float2 const *p = reinterpret_cast<float2 const *>(row + (size_t)lane * 6); // 24-byte elements, 8-byte aligned
  float2 const a = p[0], b = p[1], c = p[2];
  out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y; out[4] = c.x; out[5] = c.y;
}
template <> DCP_FN void load_q<8>(float const *__restrict__ row, lu lane, lf (&out)[8])
{
  float4 const *p = reinterpret_cast<float4 const *>(row + (size_t)lane * 8);
  float4 const a = p[0], b = p[1];
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
}

template <> DCP_FN void load_q<10>(float const *__restrict__ row, lu lane, lf (&out)[10])
{
  float2 const *p = reinterpret_cast<float2 const *>(row + (size_t)lane * 10); // 40-byte elements, 8-byte aligned
#pragma unroll
  for (int j = 0; j < 5; ++j)
  {
    float2 const a = p[j];
    out[2 * j] = a.x; out[2 * j + 1] = a.y;
  }
}
// ---- emission rows: { null, bg, 0, 0, match[0..Kp) } behind one scalar byte offset ----
// Addressed through a buffer resource: the row offset travels in an SGPR (soffset) and
// the lane's own offset in one VGPR computed once, so a row read costs no VALU at all.
typedef unsigned int dcp_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int dcp_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int dcp_u32x4 __attribute__((ext_vector_type(4)));

struct RowSrc
{
  __amdgpu_buffer_rsrc_t rsrc;
  char const *base;
};

DCP_FN RowSrc rowsrc_make(float const *__restrict__ base, uint32_t bytes)
{
  RowSrc r;
  r.base = reinterpret_cast<char const *>(base);
  r.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
  return r;
}

// byte offset of a lane's Q positions inside a row
template <int Q> DCP_FN lu row_lane_offset(lu lane) { return lane * (uint32_t)(Q * 4) + (uint32_t)(DCP_ROW_HDR * 4); }

DCP_FN void load_row_hdr(RowSrc const &r, uint32_t soff, float &nil, float &bg)
{
  float2 const h = *reinterpret_cast<float2 const *>(r.base + soff); // uniform address: s_load_dwordx2
  nil = h.x;
  bg = h.y;
}

template <int Q> DCP_FN void load_row_q(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[Q]);
template <> DCP_FN void load_row_q<1>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[1])
{
  out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r.rsrc, voff, soff, 0));
}
template <> DCP_FN void load_row_q<2>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[2])
{
  dcp_u32x2 const v = __builtin_amdgcn_raw_buffer_load_b64(r.rsrc, voff, soff, 0);
  out[0] = __uint_as_float(v.x);
  out[1] = __uint_as_float(v.y);
}
template <> DCP_FN void load_row_q<3>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[3])
{
  dcp_u32x3 const v = __builtin_amdgcn_raw_buffer_load_b96(r.rsrc, voff, soff, 0);
  out[0] = __uint_as_float(v.x);
  out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z);
}
template <> DCP_FN void load_row_q<4>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[4])
{
  dcp_u32x4 const v = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  out[0] = __uint_as_float(v.x);
  out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z);
  out[3] = __uint_as_float(v.w);
}

template <> DCP_FN void load_row_q<5>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[5])
{
  dcp_u32x4 const a = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(a.y); out[2] = __uint_as_float(a.z);
  out[3] = __uint_as_float(a.w);
  out[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r.rsrc, voff + 16u, soff, 0));
}
template <> DCP_FN void load_row_q<7>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[7])
{
  dcp_u32x4 const a = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  dcp_u32x3 const b = __builtin_amdgcn_raw_buffer_load_b96(r.rsrc, voff + 16u, soff, 0);
  out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(a.y); out[2] = __uint_as_float(a.z);
  out[3] = __uint_as_float(a.w); out[4] = __uint_as_float(b.x); out[5] = __uint_as_float(b.y);
  out[6] = __uint_as_float(b.z);
}
template <> DCP_FN void load_row_q<6>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[6])
{
  dcp_u32x4 const a = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  dcp_u32x2 const b = __builtin_amdgcn_raw_buffer_load_b64(r.rsrc, voff + 16u, soff, 0);
  out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(a.y); out[2] = __uint_as_float(a.z);
  out[3] = __uint_as_float(a.w); out[4] = __uint_as_float(b.x); out[5] = __uint_as_float(b.y);
}
template <> DCP_FN void load_row_q<8>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[8])
{
  dcp_u32x4 const a = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  dcp_u32x4 const b = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff + 16u, soff, 0);
  out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(a.y); out[2] = __uint_as_float(a.z);
  out[3] = __uint_as_float(a.w); out[4] = __uint_as_float(b.x); out[5] = __uint_as_float(b.y);
  out[6] = __uint_as_float(b.z); out[7] = __uint_as_float(b.w);
}

template <> DCP_FN void load_row_q<10>(RowSrc const &r, lu voff, uint32_t soff, lf (&out)[10])
{
  dcp_u32x4 const a = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff, soff, 0);
  dcp_u32x4 const b = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, voff + 16u, soff, 0);
  dcp_u32x2 const c = __builtin_amdgcn_raw_buffer_load_b64(r.rsrc, voff + 32u, soff, 0);
  out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(a.y); out[2] = __uint_as_float(a.z);
  out[3] = __uint_as_float(a.w); out[4] = __uint_as_float(b.x); out[5] = __uint_as_float(b.y);
  out[6] = __uint_as_float(b.z); out[7] = __uint_as_float(b.w); out[8] = __uint_as_float(c.x);
  out[9] = __uint_as_float(c.y);
}
// one DP-table row plane: the lane's Q values at row[lane*Q ..], rows padded to Kp.
// Non-temporal stores: tables and checkpoints are written once and read once, by another kernel, 66 GB of them for the
// headline scan's 2301 hits -- through the L2 they evicted the emission rows every wavefront of the path pass keeps
// re-reading (the pass: 43.3 -> 35.7 ms on one box, profiles/r03_path_pass_pmc.txt).
template <int Q> DCP_FN void store_q(float *__restrict__ row, lu lane, lf const (&v)[Q])
{
  typedef float dcp_f32x2 __attribute__((ext_vector_type(2)));
  typedef float dcp_f32x4 __attribute__((ext_vector_type(4)));
  float *p = row + (size_t)lane * Q;
  if constexpr (Q == 1)
    __builtin_nontemporal_store(v[0], p);
  else if constexpr (Q == 2)
  {
    dcp_f32x2 const t = {v[0], v[1]};
    __builtin_nontemporal_store(t, reinterpret_cast<dcp_f32x2 *>(p));
  }
  else if constexpr (Q == 4)
  {
    dcp_f32x4 const t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<dcp_f32x4 *>(p));
  }
  else
  {
#pragma unroll
    for (int q = 0; q < Q; ++q) __builtin_nontemporal_store(v[q], p + q); // (the compiler merges them: x4 + x2 for six)
  }
}

// one float per lane of the group, lane-indexed
DCP_FN void store_lane(float *__restrict__ p, lu lane, lf v) { p[lane] = v; }
DCP_FN lf load_lane(float const *__restrict__ p, lu lane) { return p[lane]; }

DCP_FN void store_sp_lane0(float *__restrict__ p, lu lane, lf N, lf B, lf J, lf E, lf C)
{
  if (lane == 0)
  {
    *reinterpret_cast<float4 *>(p) = make_float4(N, B, J, E);
    p[4] = C;
  }
}

// trellis node words of one row: positions k = lane*Q + q < K
template <int Q> DCP_FN void store_nodes_q(uint16_t *__restrict__ row, int K, lu lane, lu const (&w)[Q])
{
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    int k = (int)lane * Q + q;
    if (k < K) row[k] = (uint16_t)w[q];
  }
}

// a store done by exactly one lane of the wave
DCP_FN void store_u32_lane0(uint32_t *p, lu lane, uint32_t v)
{
  if (lane == 0) *p = v;
}
DCP_FN void store_f32_lane0(float *p, lu lane, float v)
{
  if (lane == 0) *p = v;
}

// ---- several windows per wavefront (viterbi_pack.h) -----------------------------------------
// Uniform-per-window values are per GROUP of lanes there, so they are fetched and reduced per lane.
DCP_FN lu lane_shr(lu x, int s) { return x >> s; }
DCP_FN lf lneg(lf x) { return -x; }

typedef int dcp_i32x4 __attribute__((ext_vector_type(4)));
typedef float dcp_f32x2 __attribute__((ext_vector_type(2)));
typedef float dcp_f32x3 __attribute__((ext_vector_type(3)));
typedef float dcp_f32x4 __attribute__((ext_vector_type(4)));
// Structured buffer loads (MUBUF idxen [+ offen]): address = base + vindex * stride + voffset, the multiply
// done by the address unit.  hipcc has no builtin for them; the LLVM intrinsics are reached by name, which
// keeps the compiler's own s_waitcnt bookkeeping (an inline-asm load would not have it).
__device__ float dcp_sbl1(dcp_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.f32");
__device__ dcp_f32x2 dcp_sbl2(dcp_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2f32");
__device__ dcp_f32x3 dcp_sbl3(dcp_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v3f32");
__device__ dcp_f32x4 dcp_sbl4(dcp_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");

DCP_FN dcp_i32x4 dcp_make_rsrc(void const *base, uint32_t stride_bytes, uint32_t num_records)
{
  uint64_t const a = (uint64_t)base;
  dcp_i32x4 r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(((uint32_t)(a >> 32) & 0xffffu) | (stride_bytes << 16)); // stride: 14 bits from bit 48
  r.z = (int)num_records;                                             // records of `stride` bytes
  r.w = 0x00020000;                                                   // as the raw resources of RowSrc
  return r;
}

struct PackSrc
{
  dcp_i32x4 rows;  // the profile's emission rows: record c = { null[c], bg[c], 0, 0, match[c][0..Kp) }
  dcp_i32x4 codes; // DcpCodeRow records
  lu col_off;      // byte offset of the lane's first position inside a record
};

// col_off: byte offset inside a record -- of the lane's first position, or 0 for a lane that reads the header
DCP_FN PackSrc packsrc_make(float const *__restrict__ rows, int Kp, DcpCodeRow const *__restrict__ code_rows,
                            uint32_t ncode_rows, lu col_off)
{
  PackSrc s;
  s.rows = dcp_make_rsrc(rows, (uint32_t)(Kp + DCP_ROW_HDR) * 4u, (uint32_t)DCP_TABLE_SIZE);
  s.codes = dcp_make_rsrc(code_rows, (uint32_t)sizeof(DcpCodeRow), ncode_rows);
  s.col_off = col_off;
  return s;
}

// codes of DP row `row` (an index into the code-row array) of the lane's window; rows past the array read 0
DCP_FN void load_code_row(PackSrc const &s, lu row, lu (&code)[5])
{
  dcp_f32x4 const a = dcp_sbl4(s.codes, (int)row, 0, 0, 0);
  float const b = dcp_sbl1(s.codes, (int)row, 16, 0, 0);
  code[0] = __float_as_uint(a.x);
  code[1] = __float_as_uint(a.y);
  code[2] = __float_as_uint(a.z);
  code[3] = __float_as_uint(a.w);
  code[4] = __float_as_uint(b);
}

// lane 0 of every quad to its four lanes (DPP quad_perm:[0,0,0,0]); lane 0 of every group of S to its S lanes
// (ds_swizzle in bit mode: the source lane is the own lane with the low log2(S) bits cleared -- an LDS-pipe
// instruction that touches no LDS memory and needs no address)
DCP_FN lf quad_bcast0(lf x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x00, 0xf, 0xf, true)); }
template <int S> DCP_FN lf group_bcast0(lf x)
{
  if (S == 4) return quad_bcast0(x);
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (~(S - 1)) & 0x1f));
}

// r[t] = s[t] + e[t] of lane 0 of the quad, t = 0..4: the quad broadcast rides as the DPP operand of the add (the
// compiler keeps a v_mov_b32_dpp per operand here).  A DPP source wants two wait states after a VALU write.
DCP_FN void add_quad0_x5(lf (&r)[5], lf const (&s)[5], lf const (&e)[5])
{
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %5, %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %6, %11 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %7, %12 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %8, %13 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %9, %14 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4])
      : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]));
}

template <int Q> DCP_FN void load_pack_q(PackSrc const &s, lu code, lf (&out)[Q]);
template <> DCP_FN void load_pack_q<1>(PackSrc const &s, lu code, lf (&out)[1])
{
  out[0] = dcp_sbl1(s.rows, (int)code, (int)s.col_off, 0, 0);
}
template <> DCP_FN void load_pack_q<2>(PackSrc const &s, lu code, lf (&out)[2])
{
  dcp_f32x2 const v = dcp_sbl2(s.rows, (int)code, (int)s.col_off, 0, 0);
  out[0] = v.x;
  out[1] = v.y;
}
template <> DCP_FN void load_pack_q<3>(PackSrc const &s, lu code, lf (&out)[3])
{
  dcp_f32x3 const v = dcp_sbl3(s.rows, (int)code, (int)s.col_off, 0, 0);
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
}
template <> DCP_FN void load_pack_q<4>(PackSrc const &s, lu code, lf (&out)[4])
{
  dcp_f32x4 const v = dcp_sbl4(s.rows, (int)code, (int)s.col_off, 0, 0);
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
  out[3] = v.w;
}
template <> DCP_FN void load_pack_q<6>(PackSrc const &s, lu code, lf (&out)[6])
{
  dcp_f32x4 const a = dcp_sbl4(s.rows, (int)code, (int)s.col_off, 0, 0);
  dcp_f32x2 const b = dcp_sbl2(s.rows, (int)code, (int)s.col_off + 16, 0, 0);
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y;
}
template <> DCP_FN void load_pack_q<8>(PackSrc const &s, lu code, lf (&out)[8])
{
  dcp_f32x4 const a = dcp_sbl4(s.rows, (int)code, (int)s.col_off, 0, 0);
  dcp_f32x4 const b = dcp_sbl4(s.rows, (int)code, (int)s.col_off + 16, 0, 0);
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
}

// Q consecutive floats of a table row starting at the lane's column
template <int Q> DCP_FN void load_cols(float const *__restrict__ row, lu col, lf (&out)[Q])
{
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = row[col + (uint32_t)q];
}
DCP_FN lu load_u32_at(uint32_t const *__restrict__ p, lu i) { return p[i]; }
DCP_FN lf load_f32_at(float const *__restrict__ p, lu i) { return p[i]; }
DCP_FN void store_f32_where(float *__restrict__ p, lu i, lm m, lf v)
{
  if (m) p[i] = v;
}

// min over the S lanes of each group, returned to every lane of the group.  Butterfly inside a row of 16
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: DPP operands of v_min_f32); groups of 32 then take
// the other row of their pair with a ds_swizzle.
#define DCP_DPP_MIN(v, ctrl) \
  __builtin_fminf((v), __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true)))
template <int S> DCP_FN lf group_min(lf v)
{
  v = DCP_DPP_MIN(v, 0xB1);               // quad_perm:[1,0,3,2]
  v = DCP_DPP_MIN(v, 0x4E);               // quad_perm:[2,3,0,1]
  if (S >= 8) v = DCP_DPP_MIN(v, 0x141);  // row_half_mirror
  if (S >= 16) v = DCP_DPP_MIN(v, 0x140); // row_mirror
  if (S == 32)
  {
    // every row is uniform now: take the other row of the pair (ds_swizzle, lane ^ 16: the LDS pipe, no memory)
    lf const t = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401f));
    v = __builtin_fminf(v, t);
  }
  return v;
}

// min(v[lane 0], v[lane 1]) of every group, in all its lanes: one quad DPP step and the group broadcast
template <int S> DCP_FN lf group_min01(lf v) { return group_bcast0<S>(DCP_DPP_MIN(v, 0xB1)); }

// N consecutive floats of an LDS table at a per-lane float index (a multiple of N where N is 2 or 4 and the
// callers' row length and column offsets are: one ds_read_b64 / b128)
typedef __attribute__((address_space(3))) float lds_float;
template <int N> DCP_FN void load_lds_q(lds_float const *t, lu idx, lf (&out)[N])
{
  if constexpr (N == 4)
  {
    typedef __attribute__((address_space(3))) dcp_f32x4 const *p4;
    dcp_f32x4 const v = *(p4)(t + idx);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
  else
  {
#pragma unroll
    for (int q = 0; q < N; ++q) out[q] = t[idx + (uint32_t)q];
  }
}

// Transition arrays of PackWave that a row needs once (in the fold): parked in LDS, one float4 per lane and
// array (a wave reads 64 consecutive float4s: no bank conflicts), so that they hold no registers meanwhile.
DCP_FN float4 *pack_stash_mem()
{
  __shared__ float4 mem[6 * 64];
  return mem;
}
template <int Q> DCP_FN void pack_stash(int slot, lf const (&v)[Q])
{
  static_assert(Q <= 4, "one float4 per lane");
  pack_stash_mem()[slot * 64 + (int)(threadIdx.x & 63u)] =
      make_float4(v[0], v[Q > 1 ? 1 : 0], v[Q > 2 ? 2 : 0], v[Q > 3 ? 3 : 0]);
}
// All six arrays back at once.  Written as assembly in two halves -- the six ds_read_b128 now, the wait where
// the values are first needed -- because the compiler would otherwise keep the (loop-invariant) LDS contents
// in registers across the whole row loop, which is exactly what parking them is meant to avoid.
struct PackFold
{
  dcp_f32x4 a[6];
};
DCP_FN void pack_unstash_issue(PackFold &f)
{
  uint32_t const addr = (uint32_t)(uintptr_t)(pack_stash_mem() + (threadIdx.x & 63u)); // LDS byte address
  asm volatile("ds_read_b128 %0, %6\n\t"
               "ds_read_b128 %1, %6 offset:1024\n\t"
               "ds_read_b128 %2, %6 offset:2048\n\t"
               "ds_read_b128 %3, %6 offset:3072\n\t"
               "ds_read_b128 %4, %6 offset:4096\n\t"
               "ds_read_b128 %5, %6 offset:5120"
               : "=v"(f.a[0]), "=v"(f.a[1]), "=v"(f.a[2]), "=v"(f.a[3]), "=v"(f.a[4]), "=v"(f.a[5])
               : "v"(addr)
               : "memory");
}
template <int Q> DCP_FN void pack_unstash_wait(PackFold &f, lf (&BM)[Q], lf (&MM)[Q], lf (&IM)[Q], lf (&DM)[Q], lf (&II)[Q],
                                               lf (&MI)[Q])
{
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.a[4]), "+v"(f.a[5]));
  lf(*const dst[6])[Q] = {&BM, &MM, &IM, &DM, &II, &MI};
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    (*dst[i])[0] = f.a[i].x;
    if (Q > 1) (*dst[i])[Q > 1 ? 1 : 0] = f.a[i].y;
    if (Q > 2) (*dst[i])[Q > 2 ? 2 : 0] = f.a[i].z;
    if (Q > 3) (*dst[i])[Q > 3 ? 3 : 0] = f.a[i].w;
  }
}
