// engine.cpp -- device-resident state and the C ABI of include/deciphon_hip.h.
//
// HBM layout (one engine = one GPU):
//   pool      float[]          all profiles back to back; per profile
//                              rows[1364][4+Kp] = {null, bg, 0, 0, match[0..Kp)} | trans[8][Kp],
//                              Kp = 64*Q*W, padding = +inf (DcpProfileDev holds the offsets)
//   profiles  DcpProfileDev[]
//   code_rows DcpCodeRow[]     per sequence len+1 rows of 32 B (built on the GPU
//                              from 1 B/nt by dcp_encode_kernel)
//   xt_table  float[S+1][16]   special transitions per amino length S (host-computed:
//                              they need double-precision log, c-core/xtrans.c:26-45)
//   problems  DcpProblem[]     sorted by (Q, profile) so that neighbouring
//                              workgroups hit the same emission table in L2
//   out       float[]          (null, alt) per window
//   arena     bytes            trellises of the path pass
#include "../../include/deciphon_hip.h"
#include "dcp_db.h"
#include "dcp_errors.h"
#include "dcp_types.h"
#include "host_logic.h"
#include "viterbi_kernels.h"

#include <algorithm>
#include <atomic>
#include <deque>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

namespace
{

template <class T> struct DevBuf
{
  T *p = nullptr;
  size_t cap = 0; // elements
  ~DevBuf() { release(); }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    release();
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e != hipSuccess)
    {
      p = nullptr;
      return e;
    }
    cap = want;
    return hipSuccess;
  }
};

// Pinned host memory for results that come back while other batches are in flight: a device-to-host copy into
// PAGEABLE memory waits for everything the device has been given (measured: the 19 KB hit list of one batch took
// 370 ms, the rest of the next batch's cost pass), a copy into pinned memory only for its own stream.
template <class T> struct PinBuf
{
  T *p = nullptr;
  size_t cap = 0;
  ~PinBuf()
  {
    if (p) (void)hipHostFree(p);
  }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t const want = n + n / 4 + 1024;
    hipError_t const e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess)
    {
      p = nullptr;
      return e;
    }
    cap = want;
    return hipSuccess;
  }
};

// DP tables of the fast path pass: chunks that are allocated as slices need them and kept until
// the engine goes (the driver wipes VRAM that is freed, and an allocation that lands on memory
// still being wiped waits for it at ~30 GB/s -- scripts/alloc_timing.py; growing without ever
// freeing never meets that).  place() hands out device addresses, reset() starts a new slice.
struct TableArena
{
  static constexpr size_t CHUNK = (size_t)2 << 30;
  struct Chunk { unsigned char *p; size_t size, used; };
  std::vector<Chunk> chunks;
  size_t held = 0;      // bytes in all chunks
  size_t placed = 0;    // bytes handed out since reset()
  double alloc_ms = 0;  // time spent in hipMalloc since reset()
  size_t cur = 0;
  ~TableArena()
  {
    for (Chunk &c : chunks) (void)hipFree(c.p);
  }
  void reset()
  {
    for (Chunk &c : chunks) c.used = 0;
    cur = 0;
    placed = 0;
    alloc_ms = 0;
  }
  // nullptr when `bytes` more would take the arena past `budget` (or the device is full)
  unsigned char *place(size_t bytes, size_t budget)
  {
    bytes = (bytes + 255) & ~(size_t)255;
    for (; cur < chunks.size(); ++cur)
    {
      Chunk &c = chunks[cur];
      if (c.size - c.used >= bytes)
      {
        unsigned char *at = c.p + c.used;
        c.used += bytes;
        placed += bytes;
        return at;
      }
    }
    size_t want = std::max(bytes, std::min(CHUNK, budget > held ? budget - held : 0));
    if (held + want > budget)
    {
      if (placed != 0) return nullptr; // the slice ends here
      // a lone table is tried whatever the budget says -- unless the budget is a hard limit
      // (DECIPHON_HIP_PATH_STRICT=1: the caller then fails with DCP_ENOMEM, as trellis_setup does when realloc fails)
      char const *strict = getenv("DECIPHON_HIP_PATH_STRICT");
      if (strict && strict[0] == '1') return nullptr;
      want = bytes;
    }
    auto const t0 = std::chrono::steady_clock::now();
    unsigned char *p = nullptr;
    if (hipMalloc((void **)&p, want) != hipSuccess)
    {
      (void)hipGetLastError();
      return nullptr;
    }
    double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    alloc_ms += ms;
    if (getenv("DECIPHON_HIP_TIMING")) fprintf(stderr, "TableArena: chunk %zu, %.2f GB in %.1f ms\n", chunks.size(), (double)want / 1e9, ms);
    chunks.push_back(Chunk{p, want, bytes});
    held += want;
    placed += bytes;
    return p;
  }
};

struct HostProfile
{
  int K, Kp, Q, W, cls;
  int pack = -1; // shape of the packed cost kernel (several windows per wavefront), -1: none
  bool narrow = false; // fits its class with one position per lane less (dcp_class_narrow_limit)
  int64_t pool_off; // floats
  std::string accession;
};

struct PathResult
{
  int K = 0, L = 0;
  float score = 0;
  size_t trellis_off = 0;      // bytes into d_trellis, valid when has_trellis
  bool has_trellis = false;    // the literal path kernel has run for this window
  bool trellis_on_host = false;
  // the unzipped path, one word per step: state id (c-core/state.h:9-25) | emission length << 16 -- as the device
  // wrote it, in the pinned buffer it came back in (dcp_hip::h_steps), or in `owned` when the host unzipped the trellis
  uint32_t const *steps = nullptr;
  int32_t nsteps = 0;
  std::vector<uint32_t> owned;
};

} // namespace

struct dcp_hip
{
  int device = 0;
  hipStream_t stream = nullptr;
  // one side stream per kernel class, so that the kernels of different
  // classes (few problems each in small scans) share the GPU instead of queueing
  hipStream_t qstream[DCP_NUM_CLASSES] = {nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[DCP_NUM_CLASSES] = {nullptr};
  hipStream_t pstream[DCP_NUM_PACK_SHAPES] = {nullptr}; // the packed cost kernels, one stream per shape
  hipEvent_t pjoin_ev[DCP_NUM_PACK_SHAPES] = {nullptr};
  hipStream_t nstream[DCP_NUM_CLASSES] = {nullptr};     // the narrow cost kernels of classes 4..6
  hipEvent_t njoin_ev[DCP_NUM_CLASSES] = {nullptr};
  std::string err;

  // profiles
  std::vector<HostProfile> profiles;
  size_t committed = 0; // profiles whose descriptors are published
  DevBuf<float> d_pool;
  size_t pool_used = 0; // floats of d_pool holding profiles
  int load_chunks = 0;  // staging chunks the last dcp_hip_load_dcp went through
  DevBuf<DcpProfileDev> d_profiles;

  // sequences
  std::vector<int64_t> seq_off, row_off;
  DevBuf<unsigned char> d_nt;
  DevBuf<int64_t> d_seq_off, d_row_off;
  DevBuf<DcpCodeRow> d_rows;

  // mode
  bool mode_set = false;
  bool multi_hits = true, hmmer3_compat = false;
  DevBuf<float> d_xt;
  int xt_rows = 0;
  std::vector<float> xt_override; // [rows][DCP_XT_STRIDE], dcp_hip_set_xtrans_table

  // problems / results.  Three sets of window lists and result buffers ("banks"): 0 and 1 for cost passes -- two
  // batches may be outstanding at once (dcp_hip_cost_hits_begin / _end), the second queued behind the first on the same
  // kernel streams so that the GPU never drains between them -- and 2 for the path pass, which has its own streams too
  // (path_set) and may run while cost batches are in flight.  `cur` is the bank the code below works on.
  struct Bank
  {
    DevBuf<DcpProblem> d_problems;
    DevBuf<DcpPack> d_packs;         // cost pass: windows of short profiles, several per wavefront
    DevBuf<int2> d_pack_groups;      // ... and, for four-lane groups, the packs of one profile that share a workgroup
    DevBuf<float> d_out;             // (null, alt) per window
    DevBuf<uint32_t> d_hits;         // dcp_hip_cost_hits: count, then (window, lrt bits) pairs
    DevBuf<float> d_ring;            // strip class (K > 4096): the rings of folded rows, one per workgroup in flight
    PinBuf<uint32_t> h_hits;         // ... on the host: the whole list comes back behind the filter
    hipEvent_t done_ev = nullptr;    // an outstanding batch: recorded behind its last device operation
    int n = -1;                      // windows of the outstanding batch, -1: none
    // the lists go up from pinned memory: a copy from PAGEABLE memory waits for everything the device has been given
    // (the upload of a batch begun while another was in flight took as long as the rest of that batch's cost pass)
    PinBuf<DcpProblem> h_problems;
    PinBuf<DcpPack> h_packs;
    PinBuf<int2> h_groups;
    hipEvent_t up_ev = nullptr; // recorded behind the uploads: the pinned lists are not rewritten before
    bool up_pending = false;
  };
  Bank bank[3];
  int cur = 0;
  int outstanding[2] = {-1, -1}; // banks of the batches begun and not yet ended, oldest first
  hipStream_t upload_stream = nullptr;
  // the path pass's own streams and events, swapped with stream / qstream / fork_ev / join_ev for its duration
  struct StreamSet
  {
    hipStream_t stream = nullptr, qstream[DCP_NUM_CLASSES] = {nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[DCP_NUM_CLASSES] = {nullptr};
  } path_set;
  DevBuf<int64_t> d_aux;           // strip class, literal path pass: table and scratch addresses per window
  DevBuf<int64_t> d_ckpt_addr;     // fast path pass: checkpoint address per window
  DevBuf<DcpTraceState> d_trace;   // fast path pass: where each window's traceback stands between blocks
  TableArena tables;               // DP tables of the fast path pass
  std::vector<int64_t> table_addr; // per window of the slice being staged (device addresses)
  std::vector<int> path_order;     // fast path pass: request windows, slowest first
  std::vector<dcp_hip_window> path_sorted;
  DevBuf<unsigned char> d_trellis; // trellises of the literal path pass
  std::vector<dcp_hip_window> path_wins; // the windows of the last dcp_hip_path
  int path_redone = 0;                   // how many of them needed the literal pass
  int path_group = 1;                    // blocks of a window computed side by side in the fast path pass
  PinBuf<int32_t> h_nsteps;        // path pass results on the host (pinned: see PinBuf)
  // the steps of a dcp_hip_path call stay where the copies from the device put them (PathResult::steps points there):
  // one pinned buffer per slice of the fast pass and one for the literal pass, reused by the next call
  std::deque<PinBuf<uint32_t>> h_steps;
  size_t h_steps_used = 0;
  PinBuf<float> h_out;
  DevBuf<uint32_t> d_steps, d_compact;
  DevBuf<int64_t> d_step_off, d_compact_off;
  DevBuf<int32_t> d_nsteps;
  std::vector<std::vector<unsigned char>> host_trellis; // fetched on demand, one per window
  std::vector<PathResult> paths;
  std::vector<DcpProblem> staged_problems; // dcp_hip_stage
  int staged_c_begin[DCP_NUM_CLASSES + 1] = {0};
  int staged_c_wide[DCP_NUM_CLASSES] = {0};
  int staged_pk_begin[DCP_NUM_PACK_SHAPES + 1] = {0};
  int staged_pg_begin[DCP_NUM_PACK_SHAPES + 1] = {0};
  double staged_cells = 0;
  int staged_n = -1;
};

#define BK(x) ((x)->bank[(x)->cur])

namespace
{

int fail(dcp_hip *x, int rc, char const *what, hipError_t e = hipSuccess)
{
  char buf[256];
  if (e != hipSuccess)
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(buf, sizeof buf, "%s", what);
  x->err = buf;
  return rc;
}

#define HIP_TRY(x, call, rc)                                                   \
  do                                                                           \
  {                                                                            \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return fail((x), (rc), #call, e_);                   \
  } while (0)

int ensure_xt(dcp_hip *x, int rows_needed)
{
  if (!x->mode_set) return fail(x, DCP_EFUNCUSE, "dcp_hip_set_mode has not been called");
  if (rows_needed <= x->xt_rows) return 0;
  // a window of the scan has at most 100 000 nucleotides (c-core/window.c:13), i.e. 33 333 amino acids: one table
  // covers them all, so that the table is never replaced while kernels that read it are in flight
  int rows = std::max(rows_needed, 33336);
  std::vector<float> tab((size_t)rows * DCP_XT_STRIDE, 0.0f);
  for (int s = 1; s < rows; ++s) dcp_xtrans(s, x->multi_hits, x->hmmer3_compat, tab.data() + (size_t)s * DCP_XT_STRIDE);
  if (!x->xt_override.empty())
    memcpy(tab.data(), x->xt_override.data(), std::min(tab.size(), x->xt_override.size()) * sizeof(float));
  HIP_TRY(x, hipDeviceSynchronize(), DCP_EFUNCUSE); // nothing may still be reading the old table
  HIP_TRY(x, x->d_xt.reserve(tab.size()), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpy(x->d_xt.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice), DCP_EFUNCUSE);
  x->xt_rows = rows;
  return 0;
}

struct Staged
{
  std::vector<DcpProblem> problems;       // sorted by (class, profile); cost pass: without the packed ones
  int c_begin[DCP_NUM_CLASSES + 1] = {0}; // problems of class c are [c_begin[c], c_begin[c+1])
  int c_wide[DCP_NUM_CLASSES] = {0};      // ... the narrow profiles' first: [c_begin[c], c_wide[c])
  std::vector<DcpPack> packs;             // cost pass: sorted by (shape, profile)
  int pk_begin[DCP_NUM_PACK_SHAPES + 1] = {0};
  std::vector<int2> pack_groups;          // shapes with an LDS variant: {first pack, count} per workgroup
  int pg_begin[DCP_NUM_PACK_SHAPES + 1] = {0};
  double cells = 0;
  size_t arena_bytes = 0;
  Staged() = default;
  Staged(Staged const &) = delete;
  Staged &operator=(Staged const &) = delete;
};

// for functions that enqueue copies from vectors of their own: whichever way they leave, the stream has passed the
// copies before the vectors go (a no-op when the function has synchronised already)
struct StreamDrain
{
  hipStream_t s;
  ~StreamDrain() { (void)hipStreamSynchronize(s); }
};

enum ArenaKind { ARENA_NONE, ARENA_TRELLIS, ARENA_TABLE };

// DP table of one window: float specials[(L+1)][8], float cells[(L+1)][3][Kp] (traceback.h)
size_t table_bytes(int L, int Kp) { return ((size_t)L + 1) * (DCP_SP_STRIDE + 3 * (size_t)Kp) * 4; }

// The fast path pass in blocks (dcp_types.h): rows between checkpoints.  DECIPHON_HIP_CKPT_ROWS overrides (tests
// use small blocks; 0 = whole windows, the tables of round 1); always a multiple of 5.
int ckpt_rows()
{
  int B = DCP_CKPT_ROWS_DEFAULT;
  if (char const *e = getenv("DECIPHON_HIP_CKPT_ROWS")) B = atoi(e);
  if (B < 0) B = 0;
  return B - B % 5;
}
// one block's table, then the window's checkpoints (16-byte aligned)
size_t block_table_bytes(int L, int Kp, int B) { return (size_t)dcp_block_slots(L, B) * (DCP_SP_STRIDE + 3 * (size_t)Kp) * 4; }
size_t ckpt_bytes(int L, int Kp, int W, int B) { return (size_t)(dcp_num_blocks(L, B) - 1) * (size_t)dcp_ckpt_floats(Kp, W) * 4; }
// (G tables side by side when G blocks of a window are computed at once: dcp_cost_store_kernel)
size_t fast_bytes(int L, int Kp, int W, int B, int G = 1)
{
  return (((size_t)std::min(G, dcp_num_blocks(L, B)) * block_table_bytes(L, Kp, B) + 15) & ~(size_t)15) + ckpt_bytes(L, Kp, W, B);
}

// v reordered by key(v[i]) in [0, nkeys), equal keys keeping their order: count, prefix, scatter
template <class Key> void bucket_stable(std::vector<DcpProblem> &v, int nkeys, Key key)
{
  std::vector<size_t> at((size_t)nkeys + 1, 0);
  std::vector<unsigned char> k(v.size());
  for (size_t i = 0; i < v.size(); ++i) ++at[(size_t)(k[i] = (unsigned char)key(v[i])) + 1];
  bool one = false;
  for (int j = 0; j < nkeys; ++j)
  {
    one = one || at[(size_t)j + 1] == v.size();
    at[(size_t)j + 1] += at[(size_t)j];
  }
  if (one) return; // a single key: already in order
  std::vector<DcpProblem> out(v.size());
  for (size_t i = 0; i < v.size(); ++i) out[at[k[i]]++] = v[i];
  v.swap(out);
}

// the n problems at p ordered by window length, longest first, equal lengths keeping their order
void longest_first(DcpProblem *p, size_t n)
{
  if (n < 2) return;
  int lens[32];
  int nl = 0;
  bool sorted = true;
  for (size_t i = 0; i < n && nl <= 32; ++i)
  {
    sorted = sorted && (i == 0 || p[i].L <= p[i - 1].L);
    int j = 0;
    while (j < nl && lens[j] != p[i].L) ++j;
    if (j == nl)
    {
      if (nl == 32)
      {
        nl = 33;
        break;
      }
      lens[nl++] = p[i].L;
    }
  }
  if (sorted && nl <= 32) return;
  if (nl > 32)
  {
    std::stable_sort(p, p + n, [](DcpProblem const &a, DcpProblem const &b) { return a.L > b.L; });
    return;
  }
  std::sort(lens, lens + nl, [](int a, int b) { return a > b; });
  size_t at[33] = {0};
  for (size_t i = 0; i < n; ++i)
  {
    int j = 0;
    while (lens[j] != p[i].L) ++j;
    ++at[j + 1];
  }
  for (int j = 0; j < nl; ++j) at[j + 1] += at[j];
  std::vector<DcpProblem> out(n);
  for (size_t i = 0; i < n; ++i)
  {
    int j = 0;
    while (lens[j] != p[i].L) ++j;
    out[at[j]++] = p[i];
  }
  std::copy(out.begin(), out.end(), p);
}

// validates windows and builds the device problem list
// (origin: the stream the lists are uploaded on -- x->stream unless a batch is begun asynchronously)
int stage(dcp_hip *x, int n, dcp_hip_window const *w, ArenaKind arena_kind, Staged &st, hipStream_t origin = nullptr)
{
  if (!origin) origin = x->stream;
  if (n < 0 || (n > 0 && !w)) return fail(x, DCP_EFUNCUSE, "bad window array");
  if (BK(x).n >= 0) return fail(x, DCP_EFUNCUSE, "a dcp_hip_cost_hits_begin is outstanding on these buffers: call dcp_hip_cost_hits_end first");
  if (x->committed != x->profiles.size()) return fail(x, DCP_EFUNCUSE, "profiles not committed");
  int const nseq = (int)x->seq_off.size() - 1;
  int max_s = 1;
  st.problems.resize((size_t)n);
  size_t arena = 0;
  bool by_profile = true; // the windows come in ascending profile order (dcp_scan_run's do): no sort needed below
  for (int i = 0; i < n; ++i)
  {
    by_profile = by_profile && (i == 0 || w[i].profile >= w[i - 1].profile);
    if (w[i].profile < 0 || w[i].profile >= (int)x->profiles.size()) return fail(x, DCP_EFUNCUSE, "bad profile index");
    if (w[i].seq < 0 || w[i].seq >= nseq) return fail(x, DCP_EFUNCUSE, "bad sequence index");
    int64_t const len = x->seq_off[(size_t)w[i].seq + 1] - x->seq_off[(size_t)w[i].seq];
    if (w[i].start < 0 || w[i].stop < w[i].start || w[i].stop > len) return fail(x, DCP_EFUNCUSE, "bad window range");
    int const L = w[i].stop - w[i].start;
    if (L < 1) return fail(x, DCP_EZEROSEQ, "empty window");
    HostProfile const &hp = x->profiles[(size_t)w[i].profile];
    DcpProblem &p = st.problems[(size_t)i];
    p.profile = w[i].profile;
    p.L = L;
    p.code_row = x->row_off[(size_t)w[i].seq] + w[i].start;
    p.xt_row = std::max(L / 3, 1); // c-core/thread.c:112
    p.out = i;
    p.trellis = 0;
    max_s = std::max(max_s, p.xt_row);
    st.cells += (double)hp.K * (double)L;
    if (arena_kind != ARENA_NONE)
    {
      size_t const bytes = arena_kind == ARENA_TRELLIS
                               ? ((size_t)L + 1) * 4 + ((size_t)L + 1) * (size_t)hp.K * 2 // c-core/trellis.h:12-21
                               : table_bytes(L, hp.Kp);
      // trellises: offsets into d_trellis; DP tables: addresses the caller placed in x->tables
      p.trellis = arena_kind == ARENA_TRELLIS ? (int64_t)arena : x->table_addr[(size_t)i];
      arena += (bytes + 15) & ~(size_t)15;
    }
  }
  st.arena_bytes = arena;
  // Cost pass: the windows of short profiles go several to a wavefront (viterbi_pack.h).  They leave the
  // problem list and come back as packs: up to G windows of ONE profile each, longest first, so that the
  // windows of a pack are of similar length (a pack runs as many rows as its longest window).
  // DECIPHON_HIP_PACK=0 keeps every window on the one-window-per-wavefront kernels (tests compare the two).
  st.packs.clear();
  for (int s = 0; s <= DCP_NUM_PACK_SHAPES; ++s) st.pk_begin[s] = 0;
  char const *pack_env = getenv("DECIPHON_HIP_PACK");
  bool const packing = arena_kind == ARENA_NONE && !(pack_env && pack_env[0] == '0') &&
                       (uint64_t)x->row_off.back() < ((uint64_t)1 << 32); // code rows are addressed by u32 index
  if (packing)
  {
    std::vector<DcpProblem> packed, rest;
    for (DcpProblem const &p : st.problems)
      (x->profiles[(size_t)p.profile].pack >= 0 ? packed : rest).push_back(p);
    if (by_profile)
    {
      // order (shape, profile, longest first) without a comparison sort over everything: a stable scatter by shape
      // keeps the profiles ascending, then each profile's run is ordered by length -- a counting pass when the run
      // holds few distinct lengths (the windows of a chain: whole windows and one tail)
      bucket_stable(packed, DCP_NUM_PACK_SHAPES, [&](DcpProblem const &a) { return x->profiles[(size_t)a.profile].pack; });
      for (size_t b = 0; b < packed.size();)
      {
        size_t e = b + 1;
        while (e < packed.size() && packed[e].profile == packed[b].profile) ++e;
        longest_first(packed.data() + b, e - b);
        b = e;
      }
    }
    else
      std::stable_sort(packed.begin(), packed.end(), [&](DcpProblem const &a, DcpProblem const &b) {
        int sa = x->profiles[(size_t)a.profile].pack, sb = x->profiles[(size_t)b.profile].pack;
        if (sa != sb) return sa < sb;
        if (a.profile != b.profile) return a.profile < b.profile;
        return a.L > b.L;
      });
    int shape = 0;
    for (size_t i = 0; i < packed.size();)
    {
      int const sh = x->profiles[(size_t)packed[i].profile].pack;
      while (shape < sh) st.pk_begin[++shape] = (int)st.packs.size();
      int pq = 0, ps = 0;
      dcp_pack_shape(sh, &pq, &ps);
      int const G = 64 / ps;
      DcpPack pk;
      memset(&pk, 0, sizeof pk);
      pk.profile = packed[i].profile;
      pk.Lmax = packed[i].L;
      int g = 0;
      for (; g < G && i < packed.size() && packed[i].profile == pk.profile; ++g, ++i)
      {
        pk.L[g] = packed[i].L;
        pk.xt_row[g] = packed[i].xt_row;
        pk.out[g] = packed[i].out;
        pk.code_row[g] = (uint32_t)packed[i].code_row;
      }
      st.packs.push_back(pk);
    }
    while (shape < DCP_NUM_PACK_SHAPES) st.pk_begin[++shape] = (int)st.packs.size();
    st.problems.swap(rest);
    // four-lane groups: the packs of one profile go WG to a workgroup, which shares the profile's table in LDS
    st.pack_groups.clear();
    char const *lds_env = getenv("DECIPHON_HIP_PACK_LDS");
    bool const lds_tables = !(lds_env && lds_env[0] == '0');
    for (int s = 0; s < DCP_NUM_PACK_SHAPES; ++s)
    {
      st.pg_begin[s] = (int)st.pack_groups.size();
      int const wg = lds_tables ? dcp_pack_lds_waves(s) : 0;
      if (!wg) continue;
      for (int i = st.pk_begin[s]; i < st.pk_begin[s + 1];)
      {
        int j = i;
        while (j < st.pk_begin[s + 1] && j - i < wg && st.packs[(size_t)j].profile == st.packs[(size_t)i].profile) ++j;
        st.pack_groups.push_back(make_int2(i - st.pk_begin[s], j - i));
        i = j;
      }
    }
    st.pg_begin[DCP_NUM_PACK_SHAPES] = (int)st.pack_groups.size();
  }
  int const nu = (int)st.problems.size(); // windows that keep a wavefront (or a workgroup) to themselves
  // by (class, the narrow profiles of a class first -- one position per lane less: their own launch of the cost pass --,
  // profile)
  if (by_profile)
    bucket_stable(st.problems, 2 * DCP_NUM_CLASSES, [&](DcpProblem const &a) {
      HostProfile const &hp = x->profiles[(size_t)a.profile];
      return 2 * hp.cls + (hp.narrow ? 0 : 1);
    });
  else
    std::stable_sort(st.problems.begin(), st.problems.end(), [&](DcpProblem const &a, DcpProblem const &b) {
      HostProfile const &pa = x->profiles[(size_t)a.profile], &pb = x->profiles[(size_t)b.profile];
      if (pa.cls != pb.cls) return pa.cls < pb.cls;
      if (pa.narrow != pb.narrow) return pa.narrow;
      return a.profile < b.profile;
    });
  int i = 0;
  for (int c = 0; c < DCP_NUM_CLASSES; ++c)
  {
    st.c_begin[c] = i;
    while (i < nu && x->profiles[(size_t)st.problems[(size_t)i].profile].cls == c &&
           x->profiles[(size_t)st.problems[(size_t)i].profile].narrow)
      ++i;
    st.c_wide[c] = i;
    while (i < nu && x->profiles[(size_t)st.problems[(size_t)i].profile].cls == c) ++i;
  }
  st.c_begin[DCP_NUM_CLASSES] = i;
  if (i != nu) return fail(x, DCP_ELARGECORESIZE, "profile outside every kernel class");
  if (st.c_begin[DCP_STRIP_CLASS + 1] > st.c_begin[DCP_STRIP_CLASS])
    HIP_TRY(x, BK(x).d_ring.reserve((size_t)DCP_RING_SLOTS * DCP_RING_FLOATS), DCP_ENOMEM);
  int rc = ensure_xt(x, max_s + 1);
  if (rc) return rc;
  x->staged_n = -1; // the device problem list is about to be replaced
  // every allocation first: after the first copy is enqueued nothing below can fail but a copy itself
  HIP_TRY(x, BK(x).d_problems.reserve((size_t)std::max(nu, 1)), DCP_ENOMEM);
  if (!st.packs.empty()) HIP_TRY(x, BK(x).d_packs.reserve(st.packs.size()), DCP_ENOMEM);
  if (!st.pack_groups.empty()) HIP_TRY(x, BK(x).d_pack_groups.reserve(st.pack_groups.size()), DCP_ENOMEM);
  dcp_hip::Bank &B = BK(x);
  if (B.up_pending) HIP_TRY(x, hipEventSynchronize(B.up_ev), DCP_EFUNCUSE); // the previous lists have gone up
  B.up_pending = false;
  if (nu)
  {
    HIP_TRY(x, B.h_problems.reserve((size_t)nu), DCP_ENOMEM);
    memcpy(B.h_problems.p, st.problems.data(), (size_t)nu * sizeof(DcpProblem));
    HIP_TRY(x, hipMemcpyAsync(B.d_problems.p, B.h_problems.p, (size_t)nu * sizeof(DcpProblem), hipMemcpyHostToDevice, origin),
            DCP_EFUNCUSE);
  }
  if (!st.packs.empty())
  {
    HIP_TRY(x, B.h_packs.reserve(st.packs.size()), DCP_ENOMEM);
    memcpy(B.h_packs.p, st.packs.data(), st.packs.size() * sizeof(DcpPack));
    HIP_TRY(x, hipMemcpyAsync(B.d_packs.p, B.h_packs.p, st.packs.size() * sizeof(DcpPack), hipMemcpyHostToDevice, origin),
            DCP_EFUNCUSE);
  }
  if (!st.pack_groups.empty())
  {
    HIP_TRY(x, B.h_groups.reserve(st.pack_groups.size()), DCP_ENOMEM);
    memcpy(B.h_groups.p, st.pack_groups.data(), st.pack_groups.size() * sizeof(int2));
    HIP_TRY(x, hipMemcpyAsync(B.d_pack_groups.p, B.h_groups.p, st.pack_groups.size() * sizeof(int2), hipMemcpyHostToDevice,
                              origin),
            DCP_EFUNCUSE);
  }
  HIP_TRY(x, hipEventRecord(B.up_ev, origin), DCP_EFUNCUSE);
  B.up_pending = true;
  return 0;
}

DcpLaunch launch_args(dcp_hip *x, Staged const &st, int c)
{
  DcpLaunch a;
  a.pool = x->d_pool.p;
  a.profiles = x->d_profiles.p;
  a.problems = BK(x).d_problems.p + st.c_begin[c];
  a.code_rows = x->d_rows.p;
  a.xt_table = x->d_xt.p;
  a.out = BK(x).d_out.p;
  a.arena = x->d_trellis.p;
  a.nprob = st.c_begin[c + 1] - st.c_begin[c];
  a.stream = x->stream;
  a.ring = BK(x).d_ring.p;
  return a;
}

// Launches the path (or cost) kernels of every class present: the classes run
// concurrently on their own streams, forked from and joined back into x->stream.
int launch_all(dcp_hip *x, Staged const &st, bool path)
{
  int classes = 0;
  for (int c = 0; c < DCP_NUM_CLASSES; ++c) classes += st.c_begin[c + 1] > st.c_begin[c];
  bool const fork = classes > 1;
  if (fork) HIP_TRY(x, hipEventRecord(x->fork_ev, x->stream), DCP_EFUNCUSE);
  std::vector<hipEvent_t> joins; // joined after the last launch (see launch_cost_all)
  for (int c = 0; c < DCP_NUM_CLASSES; ++c)
  {
    DcpLaunch a = launch_args(x, st, c);
    if (a.nprob <= 0) continue;
    if (path && c == DCP_STRIP_CLASS) continue; // replayed from the DP table instead (path_literal)
    if (fork)
    {
      a.stream = x->qstream[c];
      HIP_TRY(x, hipStreamWaitEvent(a.stream, x->fork_ev, 0), DCP_EFUNCUSE);
    }
    HIP_TRY(x, path ? dcp_launch_path(c, a) : dcp_launch_cost(c, a), DCP_EFUNCUSE);
    if (fork)
    {
      HIP_TRY(x, hipEventRecord(x->join_ev[c], a.stream), DCP_EFUNCUSE);
      joins.push_back(x->join_ev[c]);
    }
  }
  for (hipEvent_t ev : joins) HIP_TRY(x, hipStreamWaitEvent(x->stream, ev, 0), DCP_EFUNCUSE);
  return 0;
}

// Cost pass.  The packed kernels (short profiles, several windows per wavefront) have one stream per shape.
// Of the rest, a small launch that mixes single-wave classes goes out as ONE fused kernel (classes 0..3
// are contiguous in the sorted problem list); large launches keep one kernel per class, which fills the GPU
// by itself and has its own register budget.  Everything is forked from `origin` (the stream the window lists were
// uploaded on; x->stream by default) and joined into x->stream.
// reps > 1 (measurement): every kernel `reps` times in its own stream before the join -- the passes of a kernel class
// follow each other without waiting for the other classes, exactly as the batches of a scan do when a second batch is
// begun while the first is in flight (dcp_hip_cost_hits_begin): no kernel's tail leaves the GPU idle but the last's.
int launch_cost_all(dcp_hip *x, Staged const &st, hipStream_t origin = nullptr, int reps = 1)
{
  if (!origin) origin = x->stream;
  int const single_wave = st.c_begin[4] - st.c_begin[0];
  int mixed = 0;
  for (int c = 0; c < 4; ++c) mixed += st.c_begin[c + 1] > st.c_begin[c];
  bool const fused = mixed >= 2 && single_wave <= 16384;
  int kernels = fused ? 1 : 0;
  for (int c = fused ? 4 : 0; c < DCP_NUM_CLASSES; ++c) kernels += st.c_begin[c + 1] > st.c_begin[c];
  for (int s = 0; s < DCP_NUM_PACK_SHAPES; ++s) kernels += st.pk_begin[s + 1] > st.pk_begin[s];
  char const *narrow_env = getenv("DECIPHON_HIP_NARROW");
  bool const narrow = !(narrow_env && narrow_env[0] == '0');
  for (int c = 4; narrow && c < DCP_NUM_CLASSES; ++c) kernels += st.c_wide[c] > st.c_begin[c];
  bool const fork = kernels > 1 || origin != x->stream;
  if (fork) HIP_TRY(x, hipEventRecord(x->fork_ev, origin), DCP_EFUNCUSE);
  // x->stream joins the kernels only after the last launch: a wait is a barrier in x->stream's hardware queue,
  // and a stream that shares that queue would start its kernel behind every barrier issued before
  // (profiles/r02_step_timeline.txt)
  std::vector<hipEvent_t> joins;
  // Launch order = start order (the hardware runs a few queues side by side and takes kernels as they come):
  // the classes with the fewest, longest-running workgroups go first -- multi-wave groups, then 8, 6, 4, 3
  // positions per lane -- and the packed kernels with their many short wavefronts last, where they fill what
  // the tails of the others leave idle.  (The other way round the K > 384 classes ran alone at the end of a
  // Pfam-shaped step: 33 + 50 ms of 456, profiles/r02_b.)
  for (int c = DCP_NUM_CLASSES - 1; c >= (fused ? 4 : 0); --c)
  {
    DcpLaunch b = launch_args(x, st, c);
    if (b.nprob <= 0) continue;
    if (fork)
    {
      b.stream = x->qstream[c];
      HIP_TRY(x, hipStreamWaitEvent(b.stream, x->fork_ev, 0), DCP_EFUNCUSE);
    }
    // the windows that fit with one position per lane less (dcp_class_narrow_limit) lead the class's list and have
    // their own kernel, on its own stream.  DECIPHON_HIP_NARROW=0: the class's kernel for all (tests compare).
    int const nn = narrow ? st.c_wide[c] - st.c_begin[c] : 0;
    if (nn > 0)
    {
      DcpLaunch n = b;
      n.nprob = nn;
      if (fork)
      {
        n.stream = x->nstream[c];
        HIP_TRY(x, hipStreamWaitEvent(n.stream, x->fork_ev, 0), DCP_EFUNCUSE);
      }
      for (int r = 0; r < reps; ++r) HIP_TRY(x, dcp_launch_cost_narrow(c, n), DCP_EFUNCUSE);
      if (fork)
      {
        HIP_TRY(x, hipEventRecord(x->njoin_ev[c], n.stream), DCP_EFUNCUSE);
        joins.push_back(x->njoin_ev[c]);
      }
      b.problems += nn;
      b.nprob -= nn;
    }
    if (b.nprob > 0)
      for (int r = 0; r < reps; ++r) HIP_TRY(x, dcp_launch_cost(c, b), DCP_EFUNCUSE);
    if (fork)
    {
      HIP_TRY(x, hipEventRecord(x->join_ev[c], b.stream), DCP_EFUNCUSE);
      joins.push_back(x->join_ev[c]);
    }
  }
  if (fused)
  {
    DcpLaunch a = launch_args(x, st, 0);
    a.nprob = single_wave;
    if (fork)
    {
      a.stream = x->qstream[0];
      HIP_TRY(x, hipStreamWaitEvent(a.stream, x->fork_ev, 0), DCP_EFUNCUSE);
    }
    for (int r = 0; r < reps; ++r) HIP_TRY(x, dcp_launch_cost_fused(a), DCP_EFUNCUSE);
    if (fork)
    {
      HIP_TRY(x, hipEventRecord(x->join_ev[0], a.stream), DCP_EFUNCUSE);
      joins.push_back(x->join_ev[0]);
    }
  }
  for (int s = DCP_NUM_PACK_SHAPES - 1; s >= 0; --s)
  {
    int const np = st.pk_begin[s + 1] - st.pk_begin[s];
    if (np <= 0) continue;
    DcpLaunch a = launch_args(x, st, 0);
    if (fork)
    {
      a.stream = x->pstream[s];
      HIP_TRY(x, hipStreamWaitEvent(a.stream, x->fork_ev, 0), DCP_EFUNCUSE);
    }
    int const ng = st.pg_begin[s + 1] - st.pg_begin[s];
    for (int r = 0; r < reps; ++r)
    {
      if (ng > 0)
        HIP_TRY(x, dcp_launch_cost_pack_lds(s, a, BK(x).d_packs.p + st.pk_begin[s], BK(x).d_pack_groups.p + st.pg_begin[s], ng,
                                            (uint32_t)x->row_off.back()),
                DCP_EFUNCUSE);
      else
        HIP_TRY(x, dcp_launch_cost_pack(s, a, BK(x).d_packs.p + st.pk_begin[s], np, (uint32_t)x->row_off.back()), DCP_EFUNCUSE);
    }
    if (fork)
    {
      HIP_TRY(x, hipEventRecord(x->pjoin_ev[s], a.stream), DCP_EFUNCUSE);
      joins.push_back(x->pjoin_ev[s]);
    }
  }
  for (hipEvent_t ev : joins) HIP_TRY(x, hipStreamWaitEvent(x->stream, ev, 0), DCP_EFUNCUSE);
  return 0;
}

// the path pass works on its own bank and streams (see dcp_hip::bank): swapped in for the duration of a call
struct PathContext
{
  dcp_hip *x;
  int saved_cur;
  static void swap_streams(dcp_hip *x)
  {
    std::swap(x->stream, x->path_set.stream);
    std::swap(x->fork_ev, x->path_set.fork_ev);
    for (int c = 0; c < DCP_NUM_CLASSES; ++c)
    {
      std::swap(x->qstream[c], x->path_set.qstream[c]);
      std::swap(x->join_ev[c], x->path_set.join_ev[c]);
    }
  }
  explicit PathContext(dcp_hip *x_) : x(x_), saved_cur(x_->cur)
  {
    swap_streams(x);
    x->cur = 2;
  }
  ~PathContext()
  {
    swap_streams(x);
    x->cur = saved_cur;
  }
};

// batches begun and not ended
int outstanding_batches(dcp_hip const *x) { return (x->outstanding[0] >= 0) + (x->outstanding[1] >= 0); }

} // namespace

extern "C" {

void dcp_hip_del(struct dcp_hip *x);
int dcp_hip_path_trellis(struct dcp_hip const *x, int i, uint32_t const **xnodes, uint16_t const **nodes);

int dcp_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

struct dcp_hip *dcp_hip_new(int device)
{
  int n = dcp_hip_device_count();
  if (device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  dcp_hip *x = new dcp_hip;
  x->device = device;
  bool ok = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&x->fork_ev, hipEventDisableTiming) == hipSuccess;
  int prio_low = 0, prio_high = 0;
  ok = ok && hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) == hipSuccess;
  // The multi-wave classes (K > 640) run on high-priority streams.  A workgroup of several wavefronts of 160-256
  // registers each is placed only where that much is free at once, and beside kernels of small wavefronts (the packed
  // kernels, 3 or 4 positions per lane) every slot that frees is taken by one of those first: on the headline workload
  // (6,2) -- 1 % of the cells -- then lasted 317 of the pass's 364 ms and held up the kernel behind it in its hardware
  // queue (the 32-lane packs, 11 % of the cells, which ran alone at the end).  A high-priority queue is served first.
  // DECIPHON_HIP_PRIO_FROM: first class (viterbi_kernels.h) that gets one; 99 = none (experiments).
  int prio_from = 6;
  if (char const *e = getenv("DECIPHON_HIP_PRIO_FROM")) prio_from = atoi(e);
  for (int c = 0; ok && c < DCP_NUM_CLASSES; ++c)
  {
    ok = ok && hipStreamCreateWithPriority(&x->qstream[c], hipStreamNonBlocking, c >= prio_from ? prio_high : 0) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&x->join_ev[c], hipEventDisableTiming) == hipSuccess;
  }
  for (int c = 0; ok && c < DCP_NUM_PACK_SHAPES; ++c)
  {
    ok = ok && hipStreamCreateWithFlags(&x->pstream[c], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&x->pjoin_ev[c], hipEventDisableTiming) == hipSuccess;
  }
  for (int c = 4; ok && c <= 6; ++c)
  {
    ok = ok && hipStreamCreateWithFlags(&x->nstream[c], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&x->njoin_ev[c], hipEventDisableTiming) == hipSuccess;
  }
  // The path pass: streams of its own, alternately of high and of normal priority.  The runtime feeds four hardware
  // queues per priority level and a queue runs its kernels one after the other; a path pass is one chain of kernels per
  // class (checkpoints, then blocks and traceback in turns), few wavefronts each, bound by latency: on one level the
  // fifth to seventh chain started only when one of the first four had ended (the pass of 2301 hits of the headline scan:
  // 55-57 ms, on two levels 48-50).  dcp_scan_run calls it with no cost batch in flight; beside one, the chains on the
  // normal level queue behind the cost kernels.
  if (char const *e = getenv("DECIPHON_HIP_PATH_STREAM_PRIORITY")) // experiment: 0 = the same priority as the cost streams
    if (e[0] == '0') prio_high = 0;
  ok = ok && hipStreamCreateWithPriority(&x->path_set.stream, hipStreamNonBlocking, prio_high) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&x->path_set.fork_ev, hipEventDisableTiming) == hipSuccess;
  for (int c = 0; ok && c < DCP_NUM_CLASSES; ++c)
  {
    ok = ok && hipStreamCreateWithPriority(&x->path_set.qstream[c], hipStreamNonBlocking, c % 2 ? prio_high : 0) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&x->path_set.join_ev[c], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipStreamCreateWithFlags(&x->upload_stream, hipStreamNonBlocking) == hipSuccess;
  for (int b = 0; ok && b < 3; ++b)
  {
    if (b < 2) ok = ok && hipEventCreateWithFlags(&x->bank[b].done_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&x->bank[b].up_ev, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok)
  {
    dcp_hip_del(x);
    return nullptr;
  }
  x->seq_off.assign(1, 0);
  x->row_off.assign(1, 0);
  return x;
}

void dcp_hip_del(struct dcp_hip *x)
{
  if (!x) return;
  (void)hipSetDevice(x->device);
  (void)hipDeviceSynchronize(); // batches begun and never ended included
  for (int c = 0; c < DCP_NUM_CLASSES; ++c)
  {
    if (x->qstream[c]) (void)hipStreamDestroy(x->qstream[c]);
    if (x->join_ev[c]) (void)hipEventDestroy(x->join_ev[c]);
    if (x->path_set.qstream[c]) (void)hipStreamDestroy(x->path_set.qstream[c]);
    if (x->path_set.join_ev[c]) (void)hipEventDestroy(x->path_set.join_ev[c]);
  }
  if (x->path_set.stream) (void)hipStreamDestroy(x->path_set.stream);
  if (x->path_set.fork_ev) (void)hipEventDestroy(x->path_set.fork_ev);
  if (x->upload_stream) (void)hipStreamDestroy(x->upload_stream);
  for (int b = 0; b < 3; ++b)
  {
    if (x->bank[b].done_ev) (void)hipEventDestroy(x->bank[b].done_ev);
    if (x->bank[b].up_ev) (void)hipEventDestroy(x->bank[b].up_ev);
  }
  for (int c = 0; c < DCP_NUM_PACK_SHAPES; ++c)
  {
    if (x->pstream[c]) (void)hipStreamDestroy(x->pstream[c]);
    if (x->pjoin_ev[c]) (void)hipEventDestroy(x->pjoin_ev[c]);
  }
  for (int c = 0; c < DCP_NUM_CLASSES; ++c)
  {
    if (x->nstream[c]) (void)hipStreamDestroy(x->nstream[c]);
    if (x->njoin_ev[c]) (void)hipEventDestroy(x->njoin_ev[c]);
  }
  if (x->fork_ev) (void)hipEventDestroy(x->fork_ev);
  if (x->stream) (void)hipStreamDestroy(x->stream);
  delete x;
}

char const *dcp_hip_strerror(struct dcp_hip const *x) { return x ? x->err.c_str() : "no engine"; }

// ---- profiles: HBM is the only resident copy ---------------------------------------
// A profile is laid out on the host in a staging buffer (rows | trans, +inf padded) and
// copied behind the profiles already resident; nothing Pfam-sized is ever held twice.

static size_t profile_floats(int Kp)
{
  size_t const floats = (size_t)DCP_TABLE_SIZE * ((size_t)Kp + DCP_ROW_HDR) + (size_t)DCP_NUM_TRANS * (size_t)Kp;
  return (floats + 3) & ~(size_t)3; // keep every profile 16-byte aligned for the dwordx4 row loads
}

static int describe(dcp_hip *x, int K, char const *accession, HostProfile &hp)
{
  if (K < 1 || K > DCP_MODEL_MAX) return fail(x, DCP_ELARGECORESIZE, "core size out of range");
  int const cls = dcp_class_of(K);
  if (cls < 0) return fail(x, DCP_ELARGECORESIZE, "core size beyond DCP_MAX_CORE_SIZE (16383: state ids keep 14 bits for k + 1)");
  hp.K = K;
  hp.cls = cls;
  dcp_class_shape(cls, &hp.Q, &hp.W);
  hp.Kp = 64 * hp.Q * hp.W;
  if (cls == DCP_STRIP_CLASS) hp.Kp *= (K + hp.Kp - 1) / hp.Kp; // whole strips
  hp.pool_off = 0;
  hp.accession = accession ? accession : "";
  hp.narrow = K <= dcp_class_narrow_limit(cls);
  hp.pack = dcp_pack_shape_of(K);
  if (hp.pack >= 0)
  {
    int pq = 0, ps = 0;
    dcp_pack_shape(hp.pack, &pq, &ps);
    if (hp.W != 1 || ps * pq > hp.Kp) hp.pack = -1; // the shape reads S * Q columns of a row
  }
  return 0;
}

// grows the device pool to `floats`, keeping what is resident
static int ensure_pool(dcp_hip *x, size_t floats)
{
  if (floats <= x->d_pool.cap) return 0;
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  DevBuf<float> bigger;
  HIP_TRY(x, bigger.reserve(std::max(floats, x->d_pool.cap + x->d_pool.cap / 2)), DCP_ENOMEM);
  if (x->pool_used)
    HIP_TRY(x, hipMemcpy(bigger.p, x->d_pool.p, x->pool_used * sizeof(float), hipMemcpyDeviceToDevice), DCP_EFUNCUSE);
  x->d_pool.release();
  x->d_pool.p = bigger.p;
  x->d_pool.cap = bigger.cap;
  bigger.p = nullptr;
  bigger.cap = 0;
  return 0;
}

// one profile from a host staging buffer to the end of the pool
static int push_profile(dcp_hip *x, HostProfile hp, std::vector<float> const &staged, int *index)
{
  int rc = ensure_pool(x, x->pool_used + staged.size());
  if (rc) return rc;
  hp.pool_off = (int64_t)x->pool_used;
  HIP_TRY(x, hipMemcpy(x->d_pool.p + x->pool_used, staged.data(), staged.size() * sizeof(float), hipMemcpyHostToDevice),
          DCP_EFUNCUSE);
  x->pool_used += staged.size();
  if (index) *index = (int)x->profiles.size();
  x->profiles.push_back(hp);
  return 0;
}

// The kernels take E_l = min_k M_l[k] (viterbi_body.h) and bound what a delete run can carry across a
// wavefront (CostWave::row): both need the delete costs MD, DD to be non-negative, which -log-probabilities
// are.  Anything else (a positive log-probability in a corrupt file, a hand-made table) is refused.
static bool delete_costs_ok(float const *trans, int K, int Kp)
{
  for (int k = 0; k < K; ++k)
    if (!(trans[(size_t)DCP_MD * Kp + k] >= 0.0f) || !(trans[(size_t)DCP_DD * Kp + k] >= 0.0f)) return false;
  return true;
}

int dcp_hip_add_profile(struct dcp_hip *x, int K, float const *trans, float const *match, float const *null_cost,
                        float const *bg_cost, int *index)
{
  if (!x || !trans || !match || !null_cost || !bg_cost) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  HostProfile hp;
  int rc = describe(x, K, nullptr, hp);
  if (rc) return rc;
  int const Kp = hp.Kp;
  size_t const stride = (size_t)Kp + DCP_ROW_HDR;
  std::vector<float> buf(profile_floats(Kp), INFINITY);
  float *r = buf.data();
  float *t = r + (size_t)DCP_TABLE_SIZE * stride;
  for (int id = 0; id < DCP_NUM_TRANS; ++id) memcpy(t + (size_t)id * Kp, trans + (size_t)id * K, sizeof(float) * K);
  if (!delete_costs_ok(t, K, Kp)) return fail(x, DCP_EFUNCUSE, "negative (or NaN) delete cost: costs are -log-probabilities");
  for (int c = 0; c < DCP_TABLE_SIZE; ++c)
  {
    float *hdr = r + (size_t)c * stride;
    hdr[0] = null_cost[c];
    hdr[1] = bg_cost[c];
    hdr[2] = hdr[3] = 0.0f;
    memcpy(hdr + DCP_ROW_HDR, match + (size_t)c * K, sizeof(float) * K);
  }
  return push_profile(x, hp, buf, index);
}

int dcp_hip_add_protein(struct dcp_hip *x, int K, float const *node_trans, float const *node_emission,
                        float const *BMk, float const *null_lprob, float const *bg_lprob, int *index)
{
  if (!x || !node_trans || !node_emission || !BMk || !null_lprob || !bg_lprob) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  HostProfile hp;
  int rc = describe(x, K, nullptr, hp);
  if (rc) return rc;
  std::vector<float> buf(profile_floats(hp.Kp), INFINITY);
  float *r = buf.data();
  float *t = r + (size_t)DCP_TABLE_SIZE * ((size_t)hp.Kp + DCP_ROW_HDR);
  dcp_setup_profile(K, hp.Kp, node_trans, node_emission, BMk, null_lprob, bg_lprob, t, r);
  if (!delete_costs_ok(t, K, hp.Kp)) return fail(x, DCP_EFDATA, "positive (or NaN) delete log-probability in the protein");
  return push_profile(x, hp, buf, index);
}

// Streams proteins [first, first+count) of a pressed database into HBM: core sizes are read
// first (so the pool is sized once), then the proteins are unpacked and transposed into
// code-major rows by up to 16 host threads, chunk by chunk, into two pinned staging buffers whose
// H2D copies overlap the unpacking of the next chunk.
int dcp_hip_load_dcp(struct dcp_hip *x, char const *path, int first, int count)
{
  if (!x || !path) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  DcpDbReader db;
  int rc = db.open(path);
  if (rc) return fail(x, rc, "cannot open database");
  int const N = db.num_proteins();
  if (first < 0 || first > N) return fail(x, DCP_EINVALPART, "first protein out of range");
  int const last = count < 0 ? N : std::min(N, first + count);
  int const n = last - first;
  if (n <= 0) return 0;

  std::vector<HostProfile> hps((size_t)n);
  std::vector<size_t> off((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i)
  {
    int K = 0;
    std::string acc;
    if ((rc = db.read_protein_head(first + i, K, acc))) return fail(x, rc, "cannot read protein");
    if ((rc = describe(x, K, acc.c_str(), hps[(size_t)i]))) return rc;
    off[(size_t)i + 1] = off[(size_t)i] + profile_floats(hps[(size_t)i].Kp);
  }
  if ((rc = ensure_pool(x, x->pool_used + off[(size_t)n]))) return rc;

  // staging chunks of 256 MiB; DECIPHON_HIP_STAGE_MB shrinks them (never below one profile of the largest
  // size present), which is how the tests drive a small database through many chunks
  size_t chunk_floats = std::max<size_t>((size_t)64 << 20, profile_floats(DCP_MAX_CORE_SIZE));
  if (char const *e = getenv("DECIPHON_HIP_STAGE_MB"))
  {
    size_t largest = 0;
    for (int i = 0; i < n; ++i) largest = std::max(largest, off[(size_t)i + 1] - off[(size_t)i]);
    chunk_floats = std::max<size_t>(((size_t)std::max(atol(e), 1L) << 20) / sizeof(float), largest);
  }
  int chunks = 0;
  float *stage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool busy[2] = {false, false};
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(x->stream);
    for (int b = 0; b < 2; ++b)
    {
      if (stage[b]) (void)hipHostFree(stage[b]);
      if (done[b]) (void)hipEventDestroy(done[b]);
    }
  };
  for (int b = 0; b < 2; ++b)
  {
    if (hipHostMalloc((void **)&stage[b], std::min(chunk_floats, off[(size_t)n]) * sizeof(float), hipHostMallocDefault) !=
            hipSuccess ||
        hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess)
    {
      cleanup();
      return fail(x, DCP_ENOMEM, "cannot allocate pinned staging buffers");
    }
  }
  int b = 0;
  for (int i0 = 0; i0 < n;)
  {
    int i1 = i0 + 1;
    while (i1 < n && off[(size_t)i1 + 1] - off[(size_t)i0] <= chunk_floats) ++i1;
    if (busy[b] && hipEventSynchronize(done[b]) != hipSuccess)
    {
      cleanup();
      return fail(x, DCP_EFUNCUSE, "staging copy failed");
    }
    float *buf = stage[b];
    // plain threads, joined per chunk: no runtime is left spinning next to the HIP callbacks
    std::atomic<int> next_protein{i0}, bad{0};
    auto work = [&]() {
      DcpProtein p;
      for (int i = next_protein.fetch_add(1); i < i1; i = next_protein.fetch_add(1))
      {
        int r = db.read_protein(first + i, p);
        if (r || p.core_size != hps[(size_t)i].K)
        {
          int expected = 0;
          bad.compare_exchange_strong(expected, r ? r : DCP_EFDATA);
          continue;
        }
        float *rows = buf + (off[(size_t)i] - off[(size_t)i0]);
        float *trans = rows + (size_t)DCP_TABLE_SIZE * ((size_t)hps[(size_t)i].Kp + DCP_ROW_HDR);
        dcp_setup_profile(p.core_size, hps[(size_t)i].Kp, p.trans.data(), p.emission.data(), p.BMk.data(),
                          p.null_emission.data(), p.bg_emission.data(), trans, rows);
        if (!delete_costs_ok(trans, p.core_size, hps[(size_t)i].Kp))
        {
          int expected = 0;
          bad.compare_exchange_strong(expected, DCP_EFDATA); // a positive delete log-probability
        }
      }
    };
    {
      unsigned nthreads = std::min<unsigned>({std::max(1u, std::thread::hardware_concurrency()), 16u, (unsigned)(i1 - i0)});
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
      work();
      for (std::thread &t : pool) t.join();
    }
    if (bad)
    {
      cleanup();
      return fail(x, bad, "cannot read protein");
    }
    size_t const floats = off[(size_t)i1] - off[(size_t)i0];
    if (hipMemcpyAsync(x->d_pool.p + x->pool_used + off[(size_t)i0], buf, floats * sizeof(float), hipMemcpyHostToDevice,
                       x->stream) != hipSuccess ||
        hipEventRecord(done[b], x->stream) != hipSuccess)
    {
      cleanup();
      return fail(x, DCP_EFUNCUSE, "staging copy failed");
    }
    busy[b] = true;
    b ^= 1;
    i0 = i1;
    ++chunks;
  }
  cleanup();
  x->load_chunks = chunks;
  for (int i = 0; i < n; ++i)
  {
    hps[(size_t)i].pool_off = (int64_t)(x->pool_used + off[(size_t)i]);
    x->profiles.push_back(hps[(size_t)i]);
  }
  x->pool_used += off[(size_t)n];
  return 0;
}

int dcp_hip_num_profiles(struct dcp_hip const *x) { return x ? (int)x->profiles.size() : 0; }

int dcp_hip_load_chunks(struct dcp_hip const *x) { return x ? x->load_chunks : 0; }

int64_t dcp_hip_pool_bytes(struct dcp_hip const *x) { return x ? (int64_t)(x->pool_used * sizeof(float)) : 0; }

int dcp_hip_profile_core_size(struct dcp_hip const *x, int i)
{
  if (!x || i < 0 || i >= (int)x->profiles.size()) return -1;
  return x->profiles[(size_t)i].K;
}

char const *dcp_hip_profile_accession(struct dcp_hip const *x, int i)
{
  if (!x || i < 0 || i >= (int)x->profiles.size()) return nullptr;
  return x->profiles[(size_t)i].accession.c_str();
}

int dcp_hip_commit_profiles(struct dcp_hip *x)
{
  if (!x) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  if (x->committed == x->profiles.size()) return 0;
  // the tables are already in HBM; what is published here are the profile descriptors
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  std::vector<DcpProfileDev> dev(x->profiles.size());
  for (size_t i = 0; i < dev.size(); ++i)
  {
    HostProfile const &hp = x->profiles[i];
    dev[i].K = hp.K;
    dev[i].Kp = hp.Kp;
    dev[i].Q = hp.Q;
    dev[i].W = hp.W;
    dev[i].rows_off = hp.pool_off;
    dev[i].trans_off = dev[i].rows_off + (int64_t)DCP_TABLE_SIZE * (hp.Kp + DCP_ROW_HDR);
    dev[i].pad0 = dev[i].pad1 = 0;
  }
  HIP_TRY(x, x->d_profiles.reserve(dev.size()), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->d_profiles.p, dev.data(), dev.size() * sizeof(DcpProfileDev), hipMemcpyHostToDevice,
                            x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  x->committed = x->profiles.size();
  return 0;
}

void dcp_hip_clear_profiles(struct dcp_hip *x)
{
  if (!x) return;
  x->pool_used = 0;
  x->profiles.clear();
  x->committed = 0;
}

int dcp_hip_encode(char const *data, int64_t n, uint8_t *out)
{
  if (n < 0 || (n > 0 && (!data || !out))) return DCP_EFUNCUSE;
  return dcp_encode_sequence(data, n, out);
}

int dcp_hip_set_sequences(struct dcp_hip *x, int nseq, uint8_t const *nt, int64_t const *offsets)
{
  if (!x || nseq < 0 || !offsets || (nseq > 0 && !nt)) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  x->seq_off.assign(offsets, offsets + nseq + 1);
  x->row_off.resize((size_t)nseq + 1);
  int64_t max_len = 0, rows = 0;
  if (x->seq_off[0] != 0) return fail(x, DCP_EFUNCUSE, "offsets[0] must be 0");
  for (int i = 0; i < nseq; ++i)
  {
    int64_t len = x->seq_off[(size_t)i + 1] - x->seq_off[(size_t)i];
    if (len < 0) return fail(x, DCP_EFUNCUSE, "offsets must not decrease");
    max_len = std::max(max_len, len);
    x->row_off[(size_t)i] = rows;
    rows += len + 1;
  }
  x->row_off[(size_t)nseq] = rows;
  int64_t const total = x->seq_off[(size_t)nseq];
  for (int64_t i = 0; i < total; ++i)
    if (nt[i] > 3) return fail(x, DCP_ESEQABC, "nucleotide index above 3");
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  HIP_TRY(x, x->d_nt.reserve((size_t)std::max<int64_t>(total, 1)), DCP_ENOMEM);
  HIP_TRY(x, x->d_seq_off.reserve((size_t)nseq + 1), DCP_ENOMEM);
  HIP_TRY(x, x->d_row_off.reserve((size_t)nseq + 1), DCP_ENOMEM);
  HIP_TRY(x, x->d_rows.reserve((size_t)std::max<int64_t>(rows, 1)), DCP_ENOMEM);
  if (total)
    HIP_TRY(x, hipMemcpyAsync(x->d_nt.p, nt, (size_t)total, hipMemcpyHostToDevice, x->stream), DCP_EFUNCUSE);
  HIP_TRY(x, hipMemcpyAsync(x->d_seq_off.p, x->seq_off.data(), ((size_t)nseq + 1) * sizeof(int64_t),
                            hipMemcpyHostToDevice, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipMemcpyAsync(x->d_row_off.p, x->row_off.data(), ((size_t)nseq + 1) * sizeof(int64_t),
                            hipMemcpyHostToDevice, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, dcp_launch_encode(x->d_nt.p, x->d_seq_off.p, x->d_row_off.p, nseq, max_len, x->d_rows.p, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  return 0;
}

int dcp_hip_set_mode(struct dcp_hip *x, int multi_hits, int hmmer3_compat)
{
  if (!x) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  bool mh = multi_hits != 0, h3 = hmmer3_compat != 0;
  if (x->mode_set && (mh != x->multi_hits || h3 != x->hmmer3_compat)) x->xt_rows = 0;
  x->multi_hits = mh;
  x->hmmer3_compat = h3;
  x->mode_set = true;
  return 0;
}

int dcp_hip_set_xtrans_table(struct dcp_hip *x, int rows, float const *xt)
{
  if (!x || rows < 0 || (rows > 0 && !xt)) return DCP_EFUNCUSE;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "cost batches are outstanding (dcp_hip_cost_hits_begin): end them first");
  x->xt_override.assign((size_t)rows * DCP_XT_STRIDE, 0.0f);
  for (int r = 0; r < rows; ++r)
    memcpy(x->xt_override.data() + (size_t)r * DCP_XT_STRIDE, xt + (size_t)r * DCP_NUM_XTRANS,
           sizeof(float) * DCP_NUM_XTRANS);
  x->xt_rows = 0; // rebuild the device table at the next launch
  return 0;
}

void dcp_hip_xtrans(int seq_size, int multi_hits, int hmmer3_compat, float xt[DCP_HIP_NUM_XTRANS])
{
  dcp_xtrans(seq_size, multi_hits != 0, hmmer3_compat != 0, xt);
}

int dcp_hip_cost(struct dcp_hip *x, int n, struct dcp_hip_window const *w, float *null_cost, float *alt_cost)
{
  if (!x || (n > 0 && (!null_cost || !alt_cost))) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  bool const timing = getenv("DECIPHON_HIP_TIMING") != nullptr && n > 1000;
  auto const t0 = std::chrono::steady_clock::now();
  Staged st;
  int rc = stage(x, n, w, ARENA_NONE, st);
  if (rc) return rc;
  if (n == 0) return 0;
  HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  if (timing) (void)hipStreamSynchronize(x->stream);
  auto const t1 = std::chrono::steady_clock::now();
  if ((rc = launch_cost_all(x, st))) return rc;
  if (timing) (void)hipStreamSynchronize(x->stream);
  auto const t2 = std::chrono::steady_clock::now();
  std::vector<float> out(2 * (size_t)n);
  HIP_TRY(x, hipMemcpyAsync(out.data(), BK(x).d_out.p, out.size() * sizeof(float), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  for (int i = 0; i < n; ++i)
  {
    null_cost[i] = out[2 * (size_t)i];
    alt_cost[i] = out[2 * (size_t)i + 1];
  }
  if (timing)
  {
    auto const t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "dcp_hip_cost: %d windows; stage %.1f ms, kernels %.1f ms, fetch %.1f ms\n", n, ms(t0, t1), ms(t1, t2),
            ms(t2, t3));
  }
  return 0;
}

int dcp_hip_cost_hits_begin(struct dcp_hip *x, int n, struct dcp_hip_window const *w)
{
  if (!x) return DCP_EFUNCUSE;
  if (x->outstanding[1] >= 0) return fail(x, DCP_EFUNCUSE, "two batches are outstanding already: call dcp_hip_cost_hits_end first");
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  int const bank = x->outstanding[0] == 0 ? 1 : 0; // the one the batch in flight (if any) does not use
  x->cur = bank;
  struct Restore
  {
    dcp_hip *x;
    ~Restore() { x->cur = 0; }
  } restore{x};
  dcp_hip::Bank &B = x->bank[bank];
  Staged st;
  // the lists go up on a stream of their own and the kernels fork from there: a batch begun while another is in
  // flight is ordered behind it only kernel class by kernel class (the class streams), not as a whole
  bool const timing = getenv("DECIPHON_HIP_TIMING") != nullptr && n > 1000;
  auto const t0 = std::chrono::steady_clock::now();
  int rc = stage(x, n, w, ARENA_NONE, st, x->upload_stream);
  if (rc) return rc;
  auto const t1 = std::chrono::steady_clock::now();
  if (n > 0)
  {
    HIP_TRY(x, B.d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
    HIP_TRY(x, B.d_hits.reserve(1 + 2 * (size_t)n), DCP_ENOMEM);
    HIP_TRY(x, B.h_hits.reserve(1 + 2 * (size_t)n), DCP_ENOMEM);
    HIP_TRY(x, hipMemsetAsync(B.d_hits.p, 0, sizeof(uint32_t), x->upload_stream), DCP_EFUNCUSE);
    if ((rc = launch_cost_all(x, st, x->upload_stream))) return rc;
    HIP_TRY(x, dcp_launch_lrt_filter(B.d_out.p, n, B.d_hits.p, x->stream), DCP_EFUNCUSE);
    // count and list together (8 B per window at most: nothing beside the kernels), into pinned memory
    HIP_TRY(x, hipMemcpyAsync(B.h_hits.p, B.d_hits.p, (1 + 2 * (size_t)n) * sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream),
            DCP_EFUNCUSE);
    HIP_TRY(x, hipEventRecord(B.done_ev, x->stream), DCP_EFUNCUSE);
  }
  B.n = n;
  x->outstanding[x->outstanding[0] >= 0 ? 1 : 0] = bank;
  if (timing)
  {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "dcp_hip_cost_hits_begin: %d windows; stage %.1f ms, enqueue %.1f ms\n", n, ms(t0, t1),
            ms(t1, std::chrono::steady_clock::now()));
  }
  return 0;
}

int dcp_hip_cost_hits_end(struct dcp_hip *x, int *nhits, int32_t *hit_window, float *hit_lrt)
{
  if (!x || !nhits) return DCP_EFUNCUSE;
  int const bank = x->outstanding[0];
  if (bank < 0) return fail(x, DCP_EFUNCUSE, "dcp_hip_cost_hits_end without dcp_hip_cost_hits_begin");
  dcp_hip::Bank &B = x->bank[bank];
  int const n = B.n;
  *nhits = 0;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  // whatever happens below, the batch is over once its device work is
  auto const t0 = std::chrono::steady_clock::now();
  hipError_t const waited = n > 0 ? hipEventSynchronize(B.done_ev) : hipSuccess;
  auto const t1 = std::chrono::steady_clock::now();
  B.n = -1;
  x->outstanding[0] = x->outstanding[1];
  x->outstanding[1] = -1;
  if (waited != hipSuccess) return fail(x, DCP_EFUNCUSE, "hipEventSynchronize", waited);
  if (n == 0) return 0;
  if (!hit_window || !hit_lrt) return DCP_EFUNCUSE;
  uint32_t const count = B.h_hits.p[0];
  if (getenv("DECIPHON_HIP_TIMING") && n > 1000)
    fprintf(stderr, "dcp_hip_cost_hits_end: %d windows, %u hits; waited %.1f ms\n", n, count,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
  if (count == 0) return 0;
  uint32_t const *pairs = B.h_hits.p + 1;
  std::vector<std::pair<uint32_t, uint32_t>> hits(count);
  for (uint32_t i = 0; i < count; ++i) hits[i] = {pairs[2 * (size_t)i], pairs[2 * (size_t)i + 1]};
  std::sort(hits.begin(), hits.end()); // the device appends in no particular order
  for (uint32_t i = 0; i < count; ++i)
  {
    hit_window[i] = (int32_t)hits[i].first;
    memcpy(hit_lrt + i, &hits[i].second, sizeof(float));
  }
  *nhits = (int)count;
  return 0;
}

int dcp_hip_cost_hits(struct dcp_hip *x, int n, struct dcp_hip_window const *w, int *nhits, int32_t *hit_window,
                      float *hit_lrt)
{
  if (!x || !nhits || (n > 0 && (!hit_window || !hit_lrt))) return DCP_EFUNCUSE;
  *nhits = 0;
  if (outstanding_batches(x)) return fail(x, DCP_EFUNCUSE, "batches are outstanding: dcp_hip_cost_hits_end first");
  int rc = dcp_hip_cost_hits_begin(x, n, w);
  if (rc) return rc;
  return dcp_hip_cost_hits_end(x, nhits, hit_window, hit_lrt);
}

int dcp_hip_cost_bench(struct dcp_hip *x, int n, struct dcp_hip_window const *w, int warmup, int reps, float *ms,
                       double *cells, float *null_cost, float *alt_cost)
{
  if (!x || n <= 0 || reps <= 0 || !ms || !cells) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  Staged st;
  int rc = stage(x, n, w, ARENA_NONE, st);
  if (rc) return rc;
  HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  for (int i = 0; i < warmup; ++i)
    if ((rc = launch_cost_all(x, st))) return rc;
  hipEvent_t e0, e1;
  HIP_TRY(x, hipEventCreate(&e0), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventCreate(&e1), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventRecord(e0, x->stream), DCP_EFUNCUSE);
  for (int i = 0; i < reps; ++i)
    if ((rc = launch_cost_all(x, st))) return rc;
  HIP_TRY(x, hipEventRecord(e1, x->stream), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventSynchronize(e1), DCP_EFUNCUSE);
  float total = 0;
  HIP_TRY(x, hipEventElapsedTime(&total, e0, e1), DCP_EFUNCUSE);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms = total / (float)reps;
  *cells = st.cells;
  if (null_cost && alt_cost)
  {
    std::vector<float> out(2 * (size_t)n);
    HIP_TRY(x, hipMemcpyAsync(out.data(), BK(x).d_out.p, out.size() * sizeof(float), hipMemcpyDeviceToHost, x->stream),
            DCP_EFUNCUSE);
    HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
    for (int i = 0; i < n; ++i)
    {
      null_cost[i] = out[2 * (size_t)i];
      alt_cost[i] = out[2 * (size_t)i + 1];
    }
  }
  return 0;
}

int dcp_hip_stage(struct dcp_hip *x, int n, struct dcp_hip_window const *w)
{
  if (!x || n <= 0) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  Staged st;
  int rc = stage(x, n, w, ARENA_NONE, st);
  if (rc) return rc;
  HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  memcpy(x->staged_c_begin, st.c_begin, sizeof(st.c_begin));
  memcpy(x->staged_c_wide, st.c_wide, sizeof(st.c_wide));
  memcpy(x->staged_pk_begin, st.pk_begin, sizeof(st.pk_begin));
  memcpy(x->staged_pg_begin, st.pg_begin, sizeof(st.pg_begin));
  x->staged_cells = st.cells;
  x->staged_n = n;
  return 0;
}

int dcp_hip_run_staged(struct dcp_hip *x, int reps, float *ms, double *cells)
{
  if (!x || x->staged_n <= 0 || reps < 0) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  Staged st;
  memcpy(st.c_begin, x->staged_c_begin, sizeof(st.c_begin));
  memcpy(st.c_wide, x->staged_c_wide, sizeof(st.c_wide));
  memcpy(st.pk_begin, x->staged_pk_begin, sizeof(st.pk_begin));
  memcpy(st.pg_begin, x->staged_pg_begin, sizeof(st.pg_begin));
  hipEvent_t e0, e1;
  HIP_TRY(x, hipEventCreate(&e0), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventCreate(&e1), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventRecord(e0, x->stream), DCP_EFUNCUSE);
  // every pass joined before the next starts.  DECIPHON_HIP_STEP_JOIN=0 (experiment): the passes follow each other as
  // the batches of a scan do (dcp_hip_cost_hits_begin while another batch is in flight), kernel class by kernel class --
  // measured no faster on the bench's 0.4 s steps (profiles/r03_scan_pipeline.txt)
  char const *join_env = getenv("DECIPHON_HIP_STEP_JOIN");
  int rc = 0;
  if (join_env && join_env[0] == '0' && reps > 0)
    rc = launch_cost_all(x, st, nullptr, reps);
  else
    for (int i = 0; i < reps && !rc; ++i) rc = launch_cost_all(x, st);
  if (rc) return rc;
  HIP_TRY(x, hipEventRecord(e1, x->stream), DCP_EFUNCUSE);
  HIP_TRY(x, hipEventSynchronize(e1), DCP_EFUNCUSE);
  float total = 0;
  HIP_TRY(x, hipEventElapsedTime(&total, e0, e1), DCP_EFUNCUSE);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (ms) *ms = total;
  if (cells) *cells = x->staged_cells;
  return 0;
}

int dcp_hip_fetch_staged(struct dcp_hip *x, float *null_cost, float *alt_cost)
{
  if (!x || x->staged_n <= 0 || !null_cost || !alt_cost) return DCP_EFUNCUSE;
  size_t const n = (size_t)x->staged_n;
  std::vector<float> out(2 * n);
  HIP_TRY(x, hipMemcpyAsync(out.data(), BK(x).d_out.p, out.size() * sizeof(float), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  for (size_t i = 0; i < n; ++i)
  {
    null_cost[i] = out[2 * i];
    alt_cost[i] = out[2 * i + 1];
  }
  return 0;
}

} // extern "C"

namespace
{

// step buffers: a path has at most L emitting steps; mute steps (S, B, E, T, D runs) are few
// in practice.  DECIPHON_HIP_UNZIP_CAP (steps) overrides the capacity: a test hook for the
// overflow fallbacks.
std::vector<int64_t> step_offsets(dcp_hip *x, Staged const &st, int n)
{
  int64_t cap_override = 0;
  if (char const *e = getenv("DECIPHON_HIP_UNZIP_CAP")) cap_override = atoll(e);
  std::vector<int64_t> off((size_t)n + 1, 0);
  for (DcpProblem const &p : st.problems)
    off[(size_t)p.out + 1] =
        cap_override > 0 ? cap_override : 2 * (int64_t)p.L + 2 * (int64_t)x->profiles[(size_t)p.profile].K + 64;
  for (int i = 0; i < n; ++i) off[(size_t)i + 1] += off[(size_t)i];
  return off;
}

// Brings the step counts and then only the steps actually written to the host:
// steps[compact[i] .. compact[i+1]) are window i's (empty where nsteps[i] < 0).
int fetch_steps(dcp_hip *x, int n, int32_t const *&nsteps, std::vector<int64_t> &compact, uint32_t const *&steps,
                size_t &total_steps)
{
  HIP_TRY(x, x->h_nsteps.reserve((size_t)std::max(n, 1)), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->h_nsteps.p, x->d_nsteps.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  nsteps = x->h_nsteps.p;
  compact.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) compact[(size_t)i + 1] = compact[(size_t)i] + (nsteps[i] > 0 ? nsteps[i] : 0);
  size_t const total = (size_t)compact[(size_t)n];
  total_steps = total;
  steps = nullptr;
  if (!total) return 0;
  HIP_TRY(x, x->d_compact_off.reserve((size_t)n + 1), DCP_ENOMEM);
  HIP_TRY(x, x->d_compact.reserve(total), DCP_ENOMEM);
  if (x->h_steps_used == x->h_steps.size()) x->h_steps.emplace_back();
  PinBuf<uint32_t> &h_steps = x->h_steps[x->h_steps_used++];
  HIP_TRY(x, h_steps.reserve(total), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->d_compact_off.p, compact.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                            x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, dcp_launch_compact_steps(x->d_steps.p, x->d_step_off.p, x->d_compact_off.p, x->d_compact.p, n, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipMemcpyAsync(h_steps.p, x->d_compact.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  steps = h_steps.p;
  return 0;
}

int fetch_trellis(dcp_hip *x, int i)
{
  PathResult &r = x->paths[(size_t)i];
  if (r.trellis_on_host) return 0;
  std::vector<unsigned char> &buf = x->host_trellis[(size_t)i];
  size_t const bytes = ((size_t)r.L + 1) * 4 + ((size_t)r.L + 1) * (size_t)r.K * 2;
  buf.resize(bytes);
  HIP_TRY(x, hipMemcpyAsync(buf.data(), x->d_trellis.p + r.trellis_off, bytes, hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE);
  r.trellis_on_host = true;
  return 0;
}

size_t path_budget(dcp_hip *x);

// The literal path pass (viterbi_path as the reference runs it, pass by pass, with the
// trellis in HBM) + trellis_unzip on the device, for the windows path_wins[idx[..]].
int path_literal(dcp_hip *x, std::vector<int> const &idx)
{
  int const n = (int)idx.size();
  if (n == 0) return 0;
  std::vector<dcp_hip_window> w((size_t)n);
  for (int j = 0; j < n; ++j) w[(size_t)j] = x->path_wins[(size_t)idx[(size_t)j]];
  // Profiles beyond 4096 positions (strip class): the register-resident path kernel does not reach
  // them; their trellis is replayed row by row from the DP table (row_replay.h).  Step 1, before
  // the problem list below replaces this one on the device: the tables.
  std::vector<int64_t> tab((size_t)n, 0), scr((size_t)n, 0);
  int max_rows = 0;
  {
    std::vector<int> sl; // local indices of the strip-class windows
    for (int j = 0; j < n; ++j)
      if (x->profiles[(size_t)w[(size_t)j].profile].cls == DCP_STRIP_CLASS) sl.push_back(j);
    if (!sl.empty())
    {
      std::vector<dcp_hip_window> ws(sl.size());
      x->tables.reset();
      x->table_addr.clear();
      // as much as the device gives (these windows are rare) unless the budget is a hard limit
      char const *strict = getenv("DECIPHON_HIP_PATH_STRICT");
      size_t const room = strict && strict[0] == '1' ? path_budget(x) : (size_t)1 << 40;
      for (size_t i = 0; i < sl.size(); ++i)
      {
        dcp_hip_window const &v = w[(size_t)sl[i]];
        ws[i] = v;
        HostProfile const &hp = x->profiles[(size_t)v.profile];
        int const L = v.stop - v.start;
        unsigned char *t = L >= 0 ? x->tables.place(table_bytes(L, hp.Kp), room) : nullptr;
        unsigned char *a = t ? x->tables.place(((size_t)L + 1) * 3 * (size_t)hp.K * sizeof(float), room) : nullptr;
        if (!t || !a) return fail(x, DCP_ENOMEM, "no device memory for the DP table of a long profile's path pass");
        x->table_addr.push_back((int64_t)(uintptr_t)t);
        tab[(size_t)sl[i]] = (int64_t)(uintptr_t)t;
        scr[(size_t)sl[i]] = (int64_t)(uintptr_t)a;
        max_rows = std::max(max_rows, L + 1);
      }
      Staged ss;
      int rc0 = stage(x, (int)ws.size(), ws.data(), ARENA_TABLE, ss);
      if (rc0) return rc0;
      HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
      DcpLaunch a = launch_args(x, ss, DCP_STRIP_CLASS);
      a.arena = nullptr;
      HIP_TRY(x, dcp_launch_cost_store(DCP_STRIP_CLASS, a, nullptr, 0, 0), DCP_EFUNCUSE);
    }
  }
  Staged st;
  int rc = stage(x, n, w.data(), ARENA_TRELLIS, st);
  if (rc) return rc;
  HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  HIP_TRY(x, x->d_trellis.reserve(st.arena_bytes), DCP_ENOMEM);
  if ((rc = launch_all(x, st, true))) return rc;
  if (max_rows > 0) // step 2 for the strip class: the rows of every such window side by side
  {
    HIP_TRY(x, x->d_aux.reserve(2 * (size_t)n), DCP_ENOMEM);
    HIP_TRY(x, hipMemcpyAsync(x->d_aux.p, tab.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, x->stream),
            DCP_EFUNCUSE);
    HIP_TRY(x, hipMemcpyAsync(x->d_aux.p + n, scr.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, x->stream),
            DCP_EFUNCUSE);
    DcpLaunch a = launch_args(x, st, DCP_STRIP_CLASS);
    HIP_TRY(x, dcp_launch_replay(a, x->d_aux.p, x->d_aux.p + n, max_rows), DCP_EFUNCUSE);
    HIP_TRY(x, hipStreamSynchronize(x->stream), DCP_EFUNCUSE); // tab/scr are read by the copies above
  }

  std::vector<int64_t> step_off = step_offsets(x, st, n);
  size_t const total_steps = (size_t)step_off[(size_t)n];
  HIP_TRY(x, x->d_steps.reserve(total_steps), DCP_ENOMEM);
  HIP_TRY(x, x->d_step_off.reserve((size_t)n + 1), DCP_ENOMEM);
  HIP_TRY(x, x->d_nsteps.reserve((size_t)n), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->d_step_off.p, step_off.data(), ((size_t)n + 1) * sizeof(int64_t),
                            hipMemcpyHostToDevice, x->stream),
          DCP_EFUNCUSE);
  {
    DcpLaunch a = launch_args(x, st, 0);
    a.problems = BK(x).d_problems.p;
    a.nprob = n;
    HIP_TRY(x, dcp_launch_unzip(a, x->d_steps.p, x->d_step_off.p, x->d_nsteps.p), DCP_EFUNCUSE);
  }
  HIP_TRY(x, x->h_out.reserve((size_t)n), DCP_ENOMEM);
  float const *out = x->h_out.p;
  int32_t const *nsteps = nullptr;
  std::vector<int64_t> compact;
  uint32_t const *steps = nullptr;
  size_t total_fetched = 0;
  HIP_TRY(x, hipMemcpyAsync(x->h_out.p, BK(x).d_out.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  if ((rc = fetch_steps(x, n, nsteps, compact, steps, total_fetched))) return rc;
  // every earlier trellis offset pointed into the arena that was just rewritten
  for (PathResult &r : x->paths) r.has_trellis = r.trellis_on_host = false;
  for (DcpProblem const &p : st.problems)
  {
    int const i = idx[(size_t)p.out];
    PathResult &r = x->paths[(size_t)i];
    r.K = x->profiles[(size_t)p.profile].K;
    r.L = p.L;
    r.score = out[(size_t)p.out];
    r.trellis_off = (size_t)p.trellis;
    r.has_trellis = true;
    r.trellis_on_host = false;
    r.steps = nullptr;
    r.nsteps = 0;
    r.owned.clear();
    int32_t const ns = nsteps[(size_t)p.out];
    // a window with no finite path at all (score +inf) has no steps: the reference never walks such
    // a trellis (process_window stops at a non-finite lrt, c-core/thread.c:118-121)
    if (!(r.score < INFINITY)) continue;
    if (ns >= 0)
    {
      r.steps = steps + compact[(size_t)p.out];
      r.nsteps = ns;
    }
    else
    {
      // the device buffer was too small for this path: fetch the trellis and unzip here
      if ((rc = fetch_trellis(x, i))) return rc;
      uint32_t const *xn = reinterpret_cast<uint32_t const *>(x->host_trellis[(size_t)i].data());
      uint16_t const *nd = reinterpret_cast<uint16_t const *>(xn + (r.L + 1));
      std::vector<int32_t> ids, sizes;
      if ((rc = dcp_unzip(r.K, r.L, xn, nd, ids, sizes))) return fail(x, rc, "trellis_unzip failed");
      r.owned.resize(ids.size());
      for (size_t k = 0; k < ids.size(); ++k) r.owned[k] = (uint32_t)ids[k] | ((uint32_t)sizes[k] << 16);
      r.steps = r.owned.data();
      r.nsteps = (int32_t)r.owned.size();
    }
  }
  return 0;
}

// The fast path pass: the cost pass once more with every row's values kept in HBM, then a
// traceback that reads the back-pointers off those values (traceback.h).  Windows whose
// traceback meets an exact tie the values alone cannot resolve come back in `redo`.
// DECIPHON_HIP_TIMING=1: phase times of the path pass on stderr (synchronises between phases)
struct PathTimer
{
  bool on = getenv("DECIPHON_HIP_TIMING") != nullptr;
  hipStream_t stream;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  std::string line;
  explicit PathTimer(hipStream_t s) : stream(s) {}
  void lap(char const *what)
  {
    if (!on) return;
    (void)hipStreamSynchronize(stream);
    auto const now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof buf, " %s %.1f ms", what, std::chrono::duration<double, std::milli>(now - t).count());
    line += buf;
    t = now;
  }
};

// windows [b, e) of x->path_sorted (window i there is window x->path_order[i] of the request)
int path_fast(dcp_hip *x, int b, int e, std::vector<int> &redo)
{
  int const n = e - b;
  PathTimer tm(x->stream);
  Staged st;
  int rc = stage(x, n, x->path_sorted.data() + b, ARENA_TABLE, st);
  if (rc) return rc;
  tm.lap("stage");
  HIP_TRY(x, BK(x).d_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  int const B = ckpt_rows();
  // checkpoints sit behind each window's block table (dcp_hip_path placed fast_bytes per window)
  std::vector<int64_t> ckpt_addr((size_t)n, 0), step_off;
  StreamDrain drain{x->stream}; // destroyed before the two: an early return does not pull them from under a copy
  int max_blocks = 1;
  for (DcpProblem const &p : st.problems)
  {
    HostProfile const &hp = x->profiles[(size_t)p.profile];
    if (hp.cls != DCP_STRIP_CLASS) // (the strip class keeps whole tables: dcp_hip_path placed table_bytes for it)
    {
      int const nb = dcp_num_blocks(p.L, B);
      ckpt_addr[(size_t)p.out] =
          p.trellis + (int64_t)(((size_t)std::min(x->path_group, nb) * block_table_bytes(p.L, hp.Kp, B) + 15) & ~(size_t)15);
      max_blocks = std::max(max_blocks, nb);
    }
  }
  HIP_TRY(x, x->d_ckpt_addr.reserve((size_t)n), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->d_ckpt_addr.p, ckpt_addr.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, x->stream),
          DCP_EFUNCUSE);
  HIP_TRY(x, x->d_trace.reserve((size_t)n), DCP_ENOMEM);
  HIP_TRY(x, hipMemsetAsync(x->d_trace.p, 0, (size_t)n * sizeof(DcpTraceState), x->stream), DCP_EFUNCUSE);
  step_off = step_offsets(x, st, n);
  size_t const total_steps = (size_t)step_off[(size_t)n];
  HIP_TRY(x, x->d_steps.reserve(total_steps), DCP_ENOMEM);
  HIP_TRY(x, x->d_step_off.reserve((size_t)n + 1), DCP_ENOMEM);
  HIP_TRY(x, x->d_nsteps.reserve((size_t)n), DCP_ENOMEM);
  HIP_TRY(x, hipMemcpyAsync(x->d_step_off.p, step_off.data(), ((size_t)n + 1) * sizeof(int64_t),
                            hipMemcpyHostToDevice, x->stream),
          DCP_EFUNCUSE);
  // the classes of a slice are each too small to fill the GPU and each lasts as long as its longest window:
  // they go out on their own streams, forked from and joined back into x->stream
  int classes = 0;
  for (int c = 0; c < DCP_NUM_CLASSES; ++c) classes += st.c_begin[c + 1] > st.c_begin[c];
  bool const fork = classes > 1;
  // The checkpoints of the windows that have more than one block, then the blocks from the last to the first.
  // Every class does that on its own stream -- checkpoints, then per block its rows and the traceback through it --
  // without waiting for the others: a class's windows are done when ITS slowest is, and the store kernel of one
  // class runs beside the traceback of another (one join at the end).
  {
    char const *fused_env = getenv("DECIPHON_HIP_PATH_FUSED");
    bool const fused = !(fused_env && fused_env[0] == '0');
    if (fork) HIP_TRY(x, hipEventRecord(x->fork_ev, x->stream), DCP_EFUNCUSE);
    std::vector<hipEvent_t> joins;
    for (int c = 0; c < DCP_NUM_CLASSES; ++c)
    {
      DcpLaunch a = launch_args(x, st, c);
      if (a.nprob <= 0) continue;
      a.arena = nullptr; // DcpProblem::trellis holds the table's address
      if (fork)
      {
        a.stream = x->qstream[c];
        HIP_TRY(x, hipStreamWaitEvent(a.stream, x->fork_ev, 0), DCP_EFUNCUSE);
      }
      if (c == DCP_STRIP_CLASS) // their tables hold the whole window: one block
      {
        HIP_TRY(x, dcp_launch_cost_store(c, a, nullptr, 0, 0), DCP_EFUNCUSE);
        HIP_TRY(x, dcp_launch_traceback(a, x->d_steps.p, x->d_step_off.p, x->d_nsteps.p, x->d_trace.p, 0, 0), DCP_EFUNCUSE);
      }
      else if (x->path_group <= 1 && fused) // one launch: every window walks its own blocks (dcp_path_blocks_kernel)
        HIP_TRY(x, dcp_launch_path_blocks(c, a, x->d_ckpt_addr.p, B, x->d_steps.p, x->d_step_off.p, x->d_nsteps.p, x->d_trace.p),
                DCP_EFUNCUSE);
      else
      {
        // The checkpoints; then, G blocks at a time from the last to the first, the rows of those blocks of every
        // window -- a workgroup per (window, block): G times the wavefronts, each walking 1 / blocks of the rows --
        // and the traceback through them.  (G = 1 with DECIPHON_HIP_PATH_FUSED=0: a launch per block and phase.)
        int const G = std::max(x->path_group, 1);
        if (max_blocks > 1) HIP_TRY(x, dcp_launch_cost_ckpt(c, a, x->d_ckpt_addr.p, B), DCP_EFUNCUSE);
        for (int it = 0; it * G < max_blocks; ++it)
        {
          HIP_TRY(x, dcp_launch_cost_store(c, a, x->d_ckpt_addr.p, B, 0, G, it), DCP_EFUNCUSE);
          HIP_TRY(x, dcp_launch_traceback(a, x->d_steps.p, x->d_step_off.p, x->d_nsteps.p, x->d_trace.p, B, 0, G, it), DCP_EFUNCUSE);
        }
      }
      if (fork)
      {
        HIP_TRY(x, hipEventRecord(x->join_ev[c], a.stream), DCP_EFUNCUSE);
        joins.push_back(x->join_ev[c]);
      }
    }
    for (hipEvent_t ev : joins) HIP_TRY(x, hipStreamWaitEvent(x->stream, ev, 0), DCP_EFUNCUSE);
  }
  tm.lap("traceback");
  HIP_TRY(x, x->h_out.reserve(2 * (size_t)n), DCP_ENOMEM);
  float const *out = x->h_out.p;
  int32_t const *nsteps = nullptr;
  std::vector<int64_t> compact;
  uint32_t const *steps = nullptr;
  size_t total_fetched = 0;
  HIP_TRY(x, hipMemcpyAsync(x->h_out.p, BK(x).d_out.p, 2 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, x->stream),
          DCP_EFUNCUSE);
  if ((rc = fetch_steps(x, n, nsteps, compact, steps, total_fetched))) return rc;
  tm.lap("fetch");
  for (DcpProblem const &p : st.problems)
  {
    PathResult &r = x->paths[(size_t)x->path_order[(size_t)(b + p.out)]];
    r.K = x->profiles[(size_t)p.profile].K;
    r.L = p.L;
    r.score = out[2 * (size_t)p.out + 1]; // the alt score of the same DP
    r.has_trellis = r.trellis_on_host = false;
    r.owned.clear();
    int32_t const ns = nsteps[(size_t)p.out];
    r.steps = ns >= 0 ? steps + compact[(size_t)p.out] : nullptr;
    r.nsteps = ns >= 0 ? ns : 0;
    if (ns < 0) redo.push_back(x->path_order[(size_t)(b + p.out)]);
  }
  tm.lap("results");
  if (tm.on)
    fprintf(stderr, "dcp_hip_path: %d windows, tables %.2f GB of %.2f GB held (hipMalloc %.1f ms), %zu steps, %zu to redo;%s\n",
            n, (double)x->tables.placed / 1e9, (double)x->tables.held / 1e9, x->tables.alloc_ms, total_fetched,
            redo.size(), tm.line.c_str());
  return 0;
}

// HBM the fast pass fills with DP tables.  A slice takes as long as its longest window however
// few windows it holds, so more memory means fewer, fuller slices -- but VRAM is cleared when it
// is allocated (35 GB/s, scripts/alloc_timing.py), so an arena sized for the whole request costs
// more than the slices it saves unless the engine lives long.  Default: what dcp_hip_path_reserve
// set aside, at least 4 GB -- enough for thousands of windows since the tables are held a block at a time
// (12 B per cell of 505 rows plus 40 B per position and 500 rows of checkpoints, about a fifteenth of a 10 kb
// window's whole table).  DECIPHON_HIP_PATH_BUDGET_MB overrides.
size_t path_budget(dcp_hip *x)
{
  if (char const *e = getenv("DECIPHON_HIP_PATH_BUDGET_MB")) return (size_t)std::max(atol(e), 1L) << 20;
  size_t const want = std::max(x->tables.held, 2 * TableArena::CHUNK);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return want;
  size_t const margin = (size_t)6 << 30; // steps, trellis redo's, the caller's own buffers
  size_t const avail = free_b + x->tables.held;
  return std::max(std::min(want, avail > 2 * margin ? avail - margin : avail / 2), (size_t)256 << 20);
}

} // namespace

extern "C" {

int dcp_hip_path(struct dcp_hip *x, int n, struct dcp_hip_window const *w)
{
  if (!x || n < 0 || (n > 0 && !w)) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  PathContext ctx(x); // its own window lists, result buffers and streams: cost batches may be in flight
  x->h_steps_used = 0;
  x->paths.clear();
  x->path_wins.assign(w, w + n);
  x->paths.resize((size_t)n);
  x->host_trellis.assign((size_t)n, std::vector<unsigned char>());
  if (n == 0) return 0;
  std::vector<int> redo;
  // every window is checked here, whichever pass takes it (path_literal indexes x->profiles before stage() looks)
  int const nseq = (int)x->seq_off.size() - 1;
  for (int i = 0; i < n; ++i)
  {
    if (w[i].profile < 0 || w[i].profile >= (int)x->profiles.size()) return fail(x, DCP_EFUNCUSE, "bad profile index");
    if (w[i].seq < 0 || w[i].seq >= nseq) return fail(x, DCP_EFUNCUSE, "bad sequence index");
    int64_t const len = x->seq_off[(size_t)w[i].seq + 1] - x->seq_off[(size_t)w[i].seq];
    if (w[i].start < 0 || w[i].stop < w[i].start || w[i].stop > len) return fail(x, DCP_EFUNCUSE, "bad window range");
  }
  char const *mode = getenv("DECIPHON_HIP_PATH"); // "literal": skip the fast pass (tests, debugging)
  if (mode && strcmp(mode, "literal") == 0)
    for (int i = 0; i < n; ++i) redo.push_back(i);
  else
  {
    // slowest windows first, so that the slices of quick windows do not each wait for a slow one
    // (a window's time is its rows times its class's time per row)
    std::vector<double> cost((size_t)n);
    for (int i = 0; i < n; ++i)
    {
      int const W = x->profiles[(size_t)w[i].profile].W;
      cost[(size_t)i] = (double)(w[i].stop - w[i].start) * (W == 1 ? 1.0 : W == 2 ? 2.0 : W == 4 ? 2.5 : W == 8 ? 3.0 : 4.0);
    }
    x->path_order.resize((size_t)n);
    for (int i = 0; i < n; ++i) x->path_order[(size_t)i] = i;
    std::stable_sort(x->path_order.begin(), x->path_order.end(),
                     [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    x->path_sorted.resize((size_t)n);
    for (int i = 0; i < n; ++i) x->path_sorted[(size_t)i] = w[x->path_order[(size_t)i]];
    // slices bounded by the HBM their DP tables take
    size_t const budget = path_budget(x);
    int const B = ckpt_rows();
    dcp_hip_window const *ws = x->path_sorted.data();
    // How many blocks of a window are computed side by side: as many as the budget holds tables for, for ALL the
    // windows of the request at once -- with few hits every block of every window (the rows of a window are then
    // walked once by one wavefront, for the checkpoints, and once by many); with many hits one (a slice of windows
    // fills the GPU by itself).  DECIPHON_HIP_PATH_GROUP overrides.
    {
      double one = 0, fixed = 0;
      int most = 1;
      for (int i = 0; i < n; ++i)
      {
        HostProfile const &hp = x->profiles[(size_t)ws[i].profile];
        int const L = ws[i].stop - ws[i].start;
        if (hp.cls == DCP_STRIP_CLASS)
          fixed += (double)table_bytes(L, hp.Kp);
        else
        {
          one += (double)block_table_bytes(L, hp.Kp, B);
          fixed += (double)ckpt_bytes(L, hp.Kp, hp.W, B);
          most = std::max(most, dcp_num_blocks(L, B));
        }
      }
      double const room = 0.9 * (double)budget - fixed;
      int G = one > 0 && room > one ? (int)std::min<double>(room / one, (double)most) : 1;
      if (char const *e = getenv("DECIPHON_HIP_PATH_GROUP")) G = std::max(atoi(e), 1);
      x->path_group = std::max(1, std::min(G, most));
    }
    for (int b = 0; b < n;)
    {
      int e = b;
      x->tables.reset();
      x->table_addr.clear();
      while (e < n)
      {
        // one block's table and the checkpoints (dcp_types.h); the whole table beyond 4096 positions
        HostProfile const &hp = x->profiles[(size_t)ws[e].profile];
        int const L = ws[e].stop - ws[e].start;
        unsigned char *at = x->tables.place(
            hp.cls == DCP_STRIP_CLASS ? table_bytes(L, hp.Kp) : fast_bytes(L, hp.Kp, hp.W, B, x->path_group), budget);
        if (!at) break;
        x->table_addr.push_back((int64_t)(uintptr_t)at);
        ++e;
      }
      if (e == b) return fail(x, DCP_ENOMEM, "a window's DP table does not fit the device memory left");
      int rc = path_fast(x, b, e, redo);
      if (rc) return rc;
      b = e;
    }
    std::sort(redo.begin(), redo.end());
  }
  x->path_redone = (int)redo.size();
  return path_literal(x, redo);
}

int dcp_hip_path_reserve(struct dcp_hip *x, int64_t bytes)
{
  if (!x || bytes < 0) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  {
    // never more than a quarter of what is free right now: several scans may share the device
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)bytes > x->tables.held + free_b / 4)
      bytes = (int64_t)(x->tables.held + free_b / 4);
  }
  x->tables.reset();
  // place() allocates chunk after chunk until the arena holds `bytes`
  while (x->tables.held < (size_t)bytes)
  {
    size_t const step = std::min(TableArena::CHUNK, (size_t)bytes - x->tables.held);
    size_t const before = x->tables.held;
    x->tables.cur = x->tables.chunks.size(); // past every chunk: force a new one
    if (!x->tables.place(step, x->tables.held + step) || x->tables.held == before)
      return fail(x, DCP_ENOMEM, "dcp_hip_path_reserve: hipMalloc failed");
  }
  x->tables.reset();
  return 0;
}

int dcp_hip_path_nsteps(struct dcp_hip const *x, int i)
{
  if (!x || i < 0 || i >= (int)x->paths.size()) return -1;
  return (int)x->paths[(size_t)i].nsteps;
}

int dcp_hip_path_steps(struct dcp_hip const *x, int i, int32_t *state_ids, int32_t *seqsizes)
{
  if (!x || i < 0 || i >= (int)x->paths.size() || !state_ids || !seqsizes) return DCP_EFUNCUSE;
  PathResult const &r = x->paths[(size_t)i];
  for (int32_t k = 0; k < r.nsteps; ++k)
  {
    state_ids[k] = (int32_t)(r.steps[k] & 0xffffu);
    seqsizes[k] = (int32_t)(r.steps[k] >> 16);
  }
  return 0;
}

int dcp_hip_path_steps_packed(struct dcp_hip const *x, int i, uint32_t const **steps, int32_t *nsteps)
{
  if (!x || i < 0 || i >= (int)x->paths.size() || !steps || !nsteps) return DCP_EFUNCUSE;
  *steps = x->paths[(size_t)i].steps;
  *nsteps = x->paths[(size_t)i].nsteps;
  return 0;
}

int dcp_hip_path_trellis(struct dcp_hip const *cx, int i, uint32_t const **xnodes, uint16_t const **nodes)
{
  dcp_hip *x = const_cast<dcp_hip *>(cx);
  if (!x || i < 0 || i >= (int)x->paths.size() || !xnodes || !nodes) return DCP_EFUNCUSE;
  HIP_TRY(x, hipSetDevice(x->device), DCP_EFUNCUSE);
  PathContext ctx(x);
  if (!x->paths[(size_t)i].has_trellis)
  {
    // The fast path pass keeps no trellis.  Somebody wants one: run the literal pass for the
    // whole batch once (its paths replace the fast ones; they are the same steps).
    std::vector<int> all((size_t)x->paths.size());
    for (size_t j = 0; j < all.size(); ++j) all[j] = (int)j;
    int rc = path_literal(x, all);
    if (rc) return rc;
  }
  int rc = fetch_trellis(x, i);
  if (rc) return rc;
  uint32_t const *xn = reinterpret_cast<uint32_t const *>(x->host_trellis[(size_t)i].data());
  *xnodes = xn;
  *nodes = reinterpret_cast<uint16_t const *>(xn + (x->paths[(size_t)i].L + 1));
  return 0;
}

int dcp_hip_path_redone(struct dcp_hip const *x) { return x ? x->path_redone : 0; }

float dcp_hip_path_score(struct dcp_hip const *x, int i)
{
  if (!x || i < 0 || i >= (int)x->paths.size()) return NAN;
  return x->paths[(size_t)i].score;
}

} // extern "C"
