// scan.cpp -- the reference's outer API (include/deciphon.h) on top of the engine.
//
// Replaces c-core/scan.c (orchestration), thread.c (thread_run / process_window),
// workload.c / work.c (profile iteration), batch.c (reads), the products.tsv
// writer (product.c, product_thread.c) and the quasi-codon decoding of the match
// column (decoder.c, match.c: host_logic.cpp dcp_decode_codon_prob), minus HMMER
// (the evalue column is "nan" and HMMER's row filter does not run).
//
// The reference walks profile-major: for each profile, for each read, for each
// window -- one DP at a time per thread.  Windows of ONE (profile, read) pair form
// a chain (the next window starts after the previous window's hit,
// c-core/window.c:21-31), but different pairs are independent and hits are rare:
// dcp_scan_run scores the no-hit chain of every pair in one launch and then lets
// only the pairs that did hit walk their real chains (see there).  Rows are emitted
// in the reference's order (profile, then read, then window) whatever the order of work.
#include "../../include/deciphon.h"
#include "../../include/deciphon_hip.h"
#include "dcp_db.h"
#include "dcp_errors.h"
#include "host_logic.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <map>
#include <future>
#include <mutex>
#include <memory>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

struct dcp_batch
{
  struct Seq
  {
    long id;
    std::string name;
    std::string text;        // uppercased + disambiguated (what dcp_batch_add stores, c-core/sequence.c:15-45)
    std::vector<uint8_t> nt; // indices 0..3
    bool has_t = false, has_u = false;
  };
  std::vector<Seq> seqs;
};

struct dcp_scan
{
  dcp_hip *eng = nullptr;
  int device = 0;
  bool multi_hits = true, hmmer3_compat = false;
  void (*callback)(void *) = nullptr;
  void *userdata = nullptr;
  std::atomic<bool> interrupted{false};
  std::atomic<int> done_proteins{0};
  int num_proteins = 0;   // of this partition
  int index_offset = 0;   // global index of local profile 0 (workload_index, c-core/workload.c:95)
  std::string abc_name = "dna";
  std::vector<std::string> products;
  double timing[DCP_SCAN_TIMING_VALUES] = {0}; // dcp_scan_last_timing
  // quasi-codon decoding (c-core/decoder.c): the database stays mapped, and the distributions of a profile are
  // read from it the first time one of its windows yields a hit
  std::unique_ptr<DcpDbReader> db;
  // a decoder is handed out empty and filled by whichever row-formatting thread needs it first (reading a profile's
  // distributions and exponentiating them is a fraction of a millisecond -- times hundreds of profiles with hits)
  struct LazyDecoder
  {
    std::once_flag once;
    int rc = 0;
    DcpDecoder dec;
  };
  std::vector<std::shared_ptr<LazyDecoder>> decoders; // by local profile
};

struct dcp_press
{
  int unused;
};

namespace
{

int loglevel() // c-core/loglevel.c:9-16
{
  static thread_local int level = 2;
  static thread_local bool cached = false;
  if (!cached)
  {
    char const *x = getenv("DECIPHON_LOGLEVEL");
    if (x) level = atoi(x);
    cached = true;
  }
  return level;
}

int raise(int rc, char const *func, char const *detail = nullptr) // c-core/error.c:103-121
{
  if (rc && loglevel() <= 2)
    fprintf(stderr, "%s %s%s%s.\n", func, dcp_error_string(rc), detail ? ". Detail: " : "", detail ? detail : "");
  return rc;
}

int mkdir_p(std::string const &dir)
{
  if (mkdir(dir.c_str(), 0755) == 0 || errno == EEXIST) return 0;
  return DCP_EMKDIR;
}

// DECIPHON_HIP_TIMING=1: phase times of dcp_scan_run on stderr
struct Phase
{
  double reads = 0, windows = 0, cost = 0, path = 0, rows = 0, write = 0, callbacks = 0;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t = t0;
  double total() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  double lap()
  {
    auto n = std::chrono::steady_clock::now();
    double d = std::chrono::duration<double>(n - t).count();
    t = n;
    return d;
  }
};

struct Row
{
  int profile, seq, window;
  std::string text;
};

// product_thread_add_match + write_match (c-core/product_thread.c:40-79,112-148) without HMMER: every step of the
// hit as "<nucleotides>,<state>,<codon>,<amino>", the last two from decoder_decode / imm_gencode_decode
// (c-core/match.c:66-89, c-core/decoder.c:38-58) for the emitting states.  *rc receives DCP_EDECODON when a step
// cannot be decoded (the reference fails the scan there).
std::string format_row(dcp_batch::Seq const &seq, int window, int wstart, int wstop, DcpHit const &hit,
                       char const *accession, char const *abc, float lrt, uint32_t const *steps,
                       DcpDecoder const &dec, std::atomic<int> *rc)
{
  // a step: state id in the low 16 bits, emission length above (dcp_hip_path_steps_packed)
  auto step_id = [&](int i) { return (int)(steps[i] & 0xffffu); };
  auto step_size = [&](int i) { return (int)(steps[i] >> 16); };
  char const *sym = seq.has_u ? "ACGU" : "ACGT";
  char head[256];
  snprintf(head, sizeof head, "%ld\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\t%.1f\tnan\t", seq.id, window, wstart, wstop, 0,
           hit.hit_start, hit.hit_stop, accession, abc, (double)lrt);
  std::string out = head;
  out.reserve(out.size() + (size_t)(hit.end_step - hit.begin_step) * 12);
  int pos = 0;
  for (int i = 0; i < hit.begin_step; ++i) pos += step_size(i);
  for (int i = hit.begin_step; i < hit.end_step; ++i)
  {
    if (i > hit.begin_step) out += ';';
    char name[8];
    int const n = step_size(i), id = step_id(i);
    dcp_state_name(id, name);
    out.append(seq.text, (size_t)(wstart + pos), (size_t)n);
    out += ',';
    out += name;
    out += ',';
    if (!dcp_state_is_mute(id))
    {
      // insert states decode against the background, match states against their node, N / J / C against the
      // null model (c-core/decoder.c:43-49)
      int const kind = id >> 14, k = (id & 0x3FFF) - 1;
      size_t const entry = kind == 1 ? 1 : kind == 0 ? 2 + (size_t)k : 0;
      uint8_t codon[3] = {0, 0, 0};
      bool ok = !(kind <= 1 && (k < 0 || k > dec.core_size)) && n >= 1 && n <= 5;
      if (ok)
      {
        // the code of the n-mer (imm_eseq's indexing: SURVEY 8a row S) keys the decoder's memo
        static unsigned const code_off[6] = {0, 0, 4, 20, 84, 340};
        unsigned code = 0;
        for (int t = 0; t < n; ++t) code = code * 4 + seq.nt[(size_t)(wstart + pos + t)];
        std::atomic<uint8_t> &slot = dec.memo[entry * DCP_TABLE_SIZE + code_off[n] + code];
        uint8_t m = slot.load(std::memory_order_relaxed);
        if (m == 0xFF)
        {
          bool const found = dcp_decode_codon_prob((double)dec.epsilon, dec.base.data() + 4 * entry, dec.prior.data() + 64 * entry,
                                                   seq.nt.data() + wstart + pos, n, codon);
          m = found ? (uint8_t)(codon[0] * 16 + codon[1] * 4 + codon[2]) : (uint8_t)0xFE;
          slot.store(m, std::memory_order_relaxed);
        }
        ok = m != 0xFE;
        codon[0] = (uint8_t)(m >> 4);
        codon[1] = (uint8_t)((m >> 2) & 3);
        codon[2] = (uint8_t)(m & 3);
      }
      char const amino = ok ? dcp_gencode_amino(dec.gencode, codon) : 0;
      if (!ok || !amino)
      {
        int expected = 0;
        rc->compare_exchange_strong(expected, !ok ? DCP_EDECODON : DCP_EGENCODEID);
      }
      else
      {
        out += sym[codon[0]];
        out += sym[codon[1]];
        out += sym[codon[2]];
        out += ',';
        out += amino;
      }
      if (!ok || !amino) out += ',';
    }
    else
      out += ',';
    pos += n;
  }
  return out;
}

int setup_common(dcp_scan *x, char const *dbfile, int device, int index, int nparts, bool balanced, bool multi_hits,
                 bool hmmer3_compat, void (*callback)(void *), void *userdata)
{
  if (!x || !dbfile) return raise(DCP_EFUNCUSE, __func__);
  if (nparts < 1) return raise(DCP_EZEROPART, __func__);
  if (index < 0 || index >= nparts) return raise(DCP_EINVALPART, __func__);
  DcpDbReader db;
  int rc = db.open(dbfile);
  if (rc) return raise(rc, __func__, dbfile);
  int const abc = db.header().abc_typeid;
  if (!(abc == 4 || abc == 5)) return raise(DCP_ENUCLTNOSUPPORT, __func__); // IMM_DNA / IMM_RNA
  x->abc_name = abc == 4 ? "dna" : "rna";
  int const N = db.num_proteins();
  nparts = std::min(nparts, std::max(N, 1)); // c-core/scan.c:102
  if (index >= nparts)
  {
    x->num_proteins = 0;
    x->index_offset = N;
  }
  else
  {
    // contiguous partitions, in database order: by count as the reference does (c-core/partition_size.c:13-16,
    // c-core/protein_reader.c:112-128), or -- balanced -- by the running sum of core sizes
    std::vector<int32_t> K, first((size_t)nparts + 1);
    if (balanced)
    {
      K.resize((size_t)N);
      std::string acc;
      for (int i = 0; i < N; ++i)
      {
        int k = 0;
        if ((rc = db.read_protein_head(i, k, acc))) return raise(rc, __func__, dbfile);
        K[(size_t)i] = k;
      }
    }
    dcp_partition_bounds(N, balanced ? K.data() : nullptr, nparts, balanced, first.data());
    x->index_offset = first[(size_t)index];
    x->num_proteins = first[(size_t)index + 1] - first[(size_t)index];
  }
  x->db.reset(new DcpDbReader);
  if ((rc = x->db->open(dbfile))) return raise(rc, __func__, dbfile);
  x->decoders.clear();
  if (x->eng) dcp_hip_del(x->eng);
  x->eng = dcp_hip_new(device);
  if (!x->eng) return raise(DCP_EFUNCUSE, __func__, "no usable HIP device (there is no CPU fallback)");
  x->device = device;
  {
    // HBM for the path pass's DP tables, first: VRAM is cleared on allocation, in the background,
    // and the clearing then overlaps the database load and the first cost pass.  Best effort:
    // dcp_hip_path allocates what it needs anyway.  16 GB (never more than a quarter of what is free) hold the
    // tables of ~8000 hit windows of a median profile at once: the path pass of a scan's first hits is then one slice.
    char const *mb = getenv("DECIPHON_HIP_PATH_BUDGET_MB");
    (void)dcp_hip_path_reserve(x->eng, mb ? (int64_t)std::max(atol(mb), 1L) << 20 : (int64_t)16 << 30);
  }
  if (x->num_proteins > 0)
  {
    if ((rc = dcp_hip_load_dcp(x->eng, dbfile, x->index_offset, x->num_proteins)))
      return raise(rc, __func__, dcp_hip_strerror(x->eng));
    if ((rc = dcp_hip_commit_profiles(x->eng))) return raise(rc, __func__, dcp_hip_strerror(x->eng));
  }
  if ((rc = dcp_hip_set_mode(x->eng, multi_hits, hmmer3_compat))) return raise(rc, __func__);
  x->multi_hits = multi_hits;
  x->hmmer3_compat = hmmer3_compat;
  x->callback = callback;
  x->userdata = userdata;
  x->interrupted = false;
  x->done_proteins = 0;
  return 0;
}

} // namespace

extern "C" {

struct dcp_scan *dcp_scan_new(void) { return new (std::nothrow) dcp_scan; }

void dcp_scan_del(struct dcp_scan const *cx)
{
  dcp_scan *x = const_cast<dcp_scan *>(cx);
  if (!x) return;
  if (x->eng) dcp_hip_del(x->eng);
  delete x;
}

int dcp_scan_setup(struct dcp_scan *x, char const *dbfile, int port, int num_threads, bool multi_hits,
                   bool hmmer3_compat, bool cache, void (*callback)(void *), void *userdata)
{
  (void)port;
  (void)cache;
  if (num_threads > 128) return raise(DCP_EMANYTHREADS, __func__); // THREAD_MAX, c-core/thread.h:7
  int device = 0;
  if (char const *d = getenv("DECIPHON_HIP_DEVICE")) device = atoi(d);
  return setup_common(x, dbfile, device, 0, 1, false, multi_hits, hmmer3_compat, callback, userdata);
}

int dcp_scan_setup_partition(struct dcp_scan *x, char const *dbfile, int device, int index, int nparts,
                             bool multi_hits, bool hmmer3_compat, void (*callback)(void *), void *userdata)
{
  return setup_common(x, dbfile, device, index, nparts, false, multi_hits, hmmer3_compat, callback, userdata);
}

int dcp_scan_setup_partition_balanced(struct dcp_scan *x, char const *dbfile, int device, int index, int nparts,
                                      bool multi_hits, bool hmmer3_compat, void (*callback)(void *), void *userdata)
{
  return setup_common(x, dbfile, device, index, nparts, true, multi_hits, hmmer3_compat, callback, userdata);
}

int dcp_scan_partition_range(struct dcp_scan const *x, int *first, int *count)
{
  if (!x || !first || !count) return DCP_EFUNCUSE;
  *first = x->index_offset;
  *count = x->num_proteins;
  return 0;
}

int dcp_scan_run(struct dcp_scan *x, struct dcp_batch *batch, char const *product_dir)
{
  if (!x || !batch || !product_dir) return raise(DCP_EFUNCUSE, __func__);
  if (!x->eng) return raise(DCP_EFUNCUSE, __func__, "dcp_scan_setup has not succeeded");
  x->done_proteins = 0;
  x->interrupted = false;
  x->products.clear();
  int rc = 0;
  Phase ph;

  // batch_encode (c-core/batch.c:60-70): every read goes to HBM once
  int const nseq = (int)batch->seqs.size();
  for (dcp_batch::Seq const &s : batch->seqs) // c-core/sequence.c:61-72
  {
    if (x->abc_name == "dna" && s.has_u) return raise(DCP_EDBDNASEQRNA, __func__);
    if (x->abc_name == "rna" && s.has_t) return raise(DCP_EDBRNASEQDNA, __func__);
  }
  std::vector<int64_t> off((size_t)nseq + 1, 0);
  for (int i = 0; i < nseq; ++i) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)batch->seqs[(size_t)i].nt.size();
  std::vector<uint8_t> nt((size_t)off[(size_t)nseq]);
  for (int i = 0; i < nseq; ++i)
    memcpy(nt.data() + off[(size_t)i], batch->seqs[(size_t)i].nt.data(), batch->seqs[(size_t)i].nt.size());
  if ((rc = dcp_hip_set_sequences(x->eng, nseq, nt.data(), off.data()))) return raise(rc, __func__, dcp_hip_strerror(x->eng));

  // product_open (c-core/product.c:14-32)
  std::string const dir = product_dir;
  if ((rc = mkdir_p(dir))) return raise(rc, __func__, product_dir);

  std::vector<Row> rows;
  // rows are formatted off the main thread, one task per path pass; joined before the sort below
  struct Job
  {
    int profile, seq, widx, wstart, wstop;
    float lrt;
    DcpHit hit;
    // the path: first where the engine holds it (dcp_hip_path_steps_packed), then a copy of the job's own -- the
    // formatter threads make it first thing, and the next path pass waits for them to have made it (steps_copied)
    uint32_t const *steps = nullptr;
    int32_t nsteps = 0;
    std::vector<uint32_t> owned;
    std::shared_ptr<dcp_scan::LazyDecoder> dec;
  };
  std::deque<std::vector<Row>> formatted;
  std::shared_future<void> steps_copied; // set once the formatters of the last path batch hold their own copies of the steps
  std::atomic<int> decode_rc{0};
  x->decoders.resize((size_t)std::max(dcp_hip_num_profiles(x->eng), 0));
  struct Joiner
  {
    std::vector<std::thread> threads;
    void add(std::thread t) { threads.push_back(std::move(t)); }
    void join()
    {
      for (std::thread &t : threads)
        if (t.joinable()) t.join();
      threads.clear();
    }
    ~Joiner() { join(); }
  } formatters;
  ph.reads += ph.lap();
  int rounds = 0;
  size_t nwindows = 0, nhits = 0;
  int const nprof = dcp_hip_num_profiles(x->eng);
  // Windows of ONE (profile, read) pair form a chain -- where window w + 1 starts depends on the hit of window w
  // (c-core/window.c:21-31, c-core/thread.c:162) -- but hits are rare, and while a pair has had none its chain is the
  // same for every pair with that read length and core size.  So the profiles are scored SPECULATIVELY, chunk by
  // chunk: every window of every pair's no-hit chain in one batch (cost pass + LRT filter on the device,
  // c-core/thread.c:114-121).  Two batches are kept in flight (dcp_hip_cost_hits_begin): the host builds, sorts and
  // uploads the window list of the next chunk while the GPU scores this one, and the kernels of a chunk follow those
  // of the chunk before class by class, so the GPU never drains in between.  Pairs without a hit are done.  The
  // hits go through the path pass (c-core/thread.c:123-166: viterbi_path, trellis_unzip, hit span, last_hit_pos) when
  // the cost batches are through; a pair with hits then walks its real chain: while the windows that follow a hit are
  // still the speculated ones their scores stand, otherwise they are scored again -- in a few small rounds at the end.
  // Rows are emitted in the reference's order (profile, read, window) whatever the order of work.
  // DECIPHON_HIP_SPECULATE=0: nothing is assumed, every pair goes round by round (the tests compare the two).
  char const *spec_env = getenv("DECIPHON_HIP_SPECULATE");
  bool const speculate = !(spec_env && spec_env[0] == '0');
  char const *beside_env = getenv("DECIPHON_HIP_PATH_BESIDE"); // experiment: path passes beside the cost batches in flight
  bool const path_beside = beside_env && beside_env[0] == '1';
  typedef std::vector<std::pair<int, int>> Chain; // [start, stop) of the windows of a pair that never hits
  auto make_chain = [](int seq_size, int core_size, Chain &c) {
    c.clear();
    DcpWindow w(seq_size, core_size);
    while (w.next()) c.emplace_back(w.start, w.stop);
  };
  // The chains of ONE profile by read length, made as the reads ask for them and dropped with the profile: reads of a
  // batch often share a length (then this is one chain per profile), but real reads need not -- a cache over all
  // (length, core size) pairs of a Pfam-sized scan of 1e4 reads of 1e4 lengths would hold 2e8 chains.
  std::map<int, Chain> chains_of_profile;
  std::deque<Chain> kept_chains; // the chains of the pairs that hit: they walk them again (PairState::spec)
  // chunks of profiles: small enough for the window table of a chunk (2^21 pairs); the first one is kept short
  // (~1e10 DP cells, a dozen milliseconds of cost pass) so that the GPU starts early and the host builds and sorts
  // the window list of the second chunk meanwhile (14 ms for the headline's 416 k windows; 4e10 cells while the upload
  // of that list still waited for the device, profiles/r03_exp_register_policy.txt: 0.497-0.499 -> 0.493-0.495 s).
  // DECIPHON_HIP_CHUNK_CELLS: cells per chunk (experiments).
  std::vector<std::pair<int, int>> chunks;
  {
    double read_nt = 0;
    for (dcp_batch::Seq const &sq : batch->seqs) read_nt += (double)sq.nt.size();
    size_t const max_pairs = 1u << 21;
    int const by_pairs = nseq > 0 ? (int)std::max<size_t>(1, max_pairs / (size_t)nseq) : std::max(nprof, 1);
    char const *cells_env = getenv("DECIPHON_HIP_CHUNK_CELLS");
    double const chunk_cells = cells_env ? atof(cells_env) : 0.0;
    double const first_cells = 1.0e10;
    for (int p0 = 0; p0 < nprof;)
    {
      int p1 = p0;
      double cells = 0;
      double const limit = chunk_cells > 0 ? chunk_cells : p0 == 0 ? first_cells : 1.0e300;
      while (p1 < nprof && p1 - p0 < by_pairs && (p1 == p0 || cells < limit))
        cells += read_nt * (double)dcp_hip_profile_core_size(x->eng, p1++);
      chunks.emplace_back(p0, p1);
      p0 = p1;
    }
  }
  struct PairState
  {
    int profile, seq;
    DcpWindow win;
    Chain const *spec;     // the speculated chain, nullptr: nothing speculated
    float const *spec_lrt; // ... and per window of it: its lrt when it passed the filter, -1 otherwise
  };
  struct Work
  {
    size_t pair;
    dcp_hip_window w;
    float lrt;
  };
  std::deque<PairState> st;                // pairs that need more than their speculated scores
  std::deque<std::vector<float>> kept_lrt; // ... and the speculated lrt of their chains' windows (PairState::spec_lrt)
  std::vector<Work> need_cost, need_path;

  // moves a pair to its next window that needs work: a path pass (a speculated window that passed the filter) or a
  // cost pass (a window nobody has scored); nothing when its chain has ended
  auto advance = [&](size_t i) {
    PairState &ps = st[i];
    while (ps.win.next())
    {
      ++nwindows;
      dcp_hip_window const w{ps.profile, ps.seq, ps.win.start, ps.win.stop};
      bool const as_speculated = ps.spec && (size_t)ps.win.idx < ps.spec->size() &&
                                 (*ps.spec)[(size_t)ps.win.idx] == std::make_pair(ps.win.start, ps.win.stop);
      if (!as_speculated)
      {
        if (x->callback) x->callback(x->userdata); // a window nobody has scored yet
        need_cost.push_back(Work{i, w, 0.0f});
        return;
      }
      float const lrt = ps.spec_lrt[ps.win.idx];
      if (lrt >= 0.0f)
      {
        need_path.push_back(Work{i, w, lrt});
        return;
      }
    }
  };

  // c-core/thread.c:123-166 for a batch of windows that passed the filter: viterbi_path, trellis_unzip, the hit span,
  // last_hit_pos; their rows go to the formatter threads; their pairs move on
  auto run_path_batch = [&]() -> int {
    std::vector<Work> batch_p;
    batch_p.swap(need_path);
    std::vector<dcp_hip_window> hits(batch_p.size());
    for (size_t k = 0; k < batch_p.size(); ++k) hits[k] = batch_p[k].w;
    nhits += hits.size();
    {
      // decoder_setup (c-core/decoder.c:21-36) for the profiles of this batch -- reading a profile's distributions out
      // of the database and exponentiating them -- by host threads WHILE the GPU walks the paths (this thread only
      // waits for it meanwhile); the row formatters below find them ready
      auto warm = std::make_shared<std::vector<std::pair<int, std::shared_ptr<dcp_scan::LazyDecoder>>>>();
      for (Work const &wk : batch_p)
      {
        std::shared_ptr<dcp_scan::LazyDecoder> &d = x->decoders[(size_t)wk.w.profile];
        if (d) continue;
        d = std::make_shared<dcp_scan::LazyDecoder>();
        warm->emplace_back(wk.w.profile, d);
      }
      if (!warm->empty())
      {
        dcp_scan const *scan = x;
        formatters.add(std::thread([warm, scan]() {
          std::atomic<size_t> next{0};
          auto work = [&]() {
            for (size_t k = next.fetch_add(1); k < warm->size(); k = next.fetch_add(1))
            {
              dcp_scan::LazyDecoder &ld = *(*warm)[k].second;
              int const profile = (*warm)[k].first;
              std::call_once(ld.once, [&]() { ld.rc = scan->db->read_decoder(scan->index_offset + profile, ld.dec); });
            }
          };
          unsigned const nthreads = std::min<unsigned>({std::max(1u, std::thread::hardware_concurrency()), 16u,
                                                        (unsigned)std::max<size_t>(warm->size() / 4, 1)});
          std::vector<std::thread> pool;
          for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
          work();
          for (std::thread &t : pool) t.join();
        }));
      }
    }
    ph.windows += ph.lap();
    if (steps_copied.valid()) steps_copied.wait(); // the previous batch's formatters still read the engine's step buffers
    steps_copied = std::shared_future<void>();
    int prc = dcp_hip_path(x->eng, (int)hits.size(), hits.data());
    if (prc) return raise(prc, __func__, dcp_hip_strerror(x->eng));
    ph.path += ph.lap();
    // hit spans first (they move the window chains: a few threads, 6 M steps to walk for the headline's 2301 hits); the
    // rows are then formatted by up to 16 host threads (a row is a few thousand short appends) while the GPU goes on
    std::vector<Job> found(batch_p.size());
    std::vector<char> is_hit(batch_p.size(), 0);
    {
      std::atomic<size_t> next_hit{0};
      std::atomic<int> steps_rc{0};
      dcp_hip const *eng = x->eng;
      auto spans = [&]() {
        for (size_t h = next_hit.fetch_add(16); h < found.size(); h = next_hit.fetch_add(16))
          for (size_t i = h, e = std::min(found.size(), h + 16); i < e; ++i)
          {
            Job &j = found[i];
            if (int const src = dcp_hip_path_steps_packed(eng, (int)i, &j.steps, &j.nsteps))
            {
              int expected = 0;
              steps_rc.compare_exchange_strong(expected, src);
              continue;
            }
            is_hit[i] = dcp_find_hit_packed(j.steps, j.nsteps, j.hit) ? 1 : 0;
          }
      };
      unsigned const nthreads = std::min<unsigned>({std::max(1u, std::thread::hardware_concurrency()), 8u,
                                                    (unsigned)std::max<size_t>(found.size() / 64, 1)});
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(spans);
      spans();
      for (std::thread &t : pool) t.join();
      if (steps_rc) return raise(steps_rc, __func__);
    }
    auto jobs = std::make_shared<std::vector<Job>>();
    for (size_t h = 0; h < batch_p.size(); ++h)
    {
      if (!is_hit[h]) continue;
      Job &j = found[h];
      PairState &ps = st[batch_p[h].pair];
      ps.win.last_hit_pos = j.hit.last_hit_pos; // window_set_last_hit_position, c-core/thread.c:162
      j.profile = ps.profile;
      j.seq = ps.seq;
      j.widx = ps.win.idx;
      j.wstart = ps.win.start;
      j.wstop = ps.win.stop;
      j.lrt = batch_p[h].lrt;
      if (!x->decoders[(size_t)ps.profile]) x->decoders[(size_t)ps.profile] = std::make_shared<dcp_scan::LazyDecoder>();
      j.dec = x->decoders[(size_t)ps.profile];
      jobs->push_back(std::move(j));
    }
    if (!jobs->empty())
    {
      formatted.emplace_back(jobs->size());
      std::vector<Row> *out = &formatted.back();
      dcp_scan const *scan = x;
      std::atomic<int> *drc = &decode_rc;
      auto copied = std::make_shared<std::promise<void>>();
      steps_copied = copied->get_future().share();
      std::shared_future<void> all_copied = steps_copied;
      formatters.add(std::thread([jobs, out, scan, batch, drc, copied, all_copied]() {
        std::atomic<size_t> next_copy{0}, ncopied{0}, next_job{0};
        auto work = [&]() {
          // the steps out of the engine's buffers first: the scan's next path pass waits for that, not for the rows
          for (size_t k = next_copy.fetch_add(1); k < jobs->size(); k = next_copy.fetch_add(1))
          {
            Job &j = (*jobs)[k];
            j.owned.assign(j.steps, j.steps + j.nsteps);
            j.steps = j.owned.data();
            if (ncopied.fetch_add(1) + 1 == jobs->size()) copied->set_value();
          }
          all_copied.wait(); // (a job may be formatted by another thread than the one that copied it)
          for (size_t k = next_job.fetch_add(1); k < jobs->size(); k = next_job.fetch_add(1))
          {
            Job const &j = (*jobs)[k];
            dcp_batch::Seq const &seq = batch->seqs[(size_t)j.seq];
            dcp_scan::LazyDecoder &ld = *j.dec; // decoder_setup, c-core/decoder.c:21-36, once per profile
            std::call_once(ld.once, [&]() { ld.rc = scan->db->read_decoder(scan->index_offset + j.profile, ld.dec); });
            if (ld.rc)
            {
              int expected = 0;
              drc->compare_exchange_strong(expected, ld.rc);
              continue;
            }
            (*out)[k] = Row{j.profile, j.seq, j.widx,
                            format_row(seq, j.widx, j.wstart, j.wstop, j.hit,
                                       dcp_hip_profile_accession(scan->eng, j.profile), scan->abc_name.c_str(),
                                       j.lrt, j.steps, ld.dec, drc)};
          }
        };
        unsigned const nthreads = std::min<unsigned>({std::max(1u, std::thread::hardware_concurrency()), 16u,
                                                      (unsigned)std::max<size_t>(jobs->size() / 8, 1)});
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
      }));
    }
    ph.rows += ph.lap();
    for (Work const &wk : batch_p) advance(wk.pair);
    ph.windows += ph.lap();
    return 0;
  };

  // c-core/thread.c:114-121 for windows nobody has scored yet (only with no batch in flight)
  auto run_cost_batch = [&]() -> int {
    std::vector<Work> batch_c;
    batch_c.swap(need_cost);
    std::vector<dcp_hip_window> wins(batch_c.size());
    for (size_t k = 0; k < batch_c.size(); ++k) wins[k] = batch_c[k].w;
    std::vector<int32_t> hit_index(wins.size());
    std::vector<float> lrts(wins.size());
    int nh = 0;
    ++rounds;
    ph.windows += ph.lap();
    int crc = dcp_hip_cost_hits(x->eng, (int)wins.size(), wins.data(), &nh, hit_index.data(), lrts.data());
    if (crc) return raise(crc, __func__, dcp_hip_strerror(x->eng));
    ph.cost += ph.lap();
    std::vector<char> is_hit(wins.size(), 0);
    for (int h = 0; h < nh; ++h)
    {
      size_t const k = (size_t)hit_index[(size_t)h];
      is_hit[k] = 1;
      need_path.push_back(Work{batch_c[k].pair, batch_c[k].w, lrts[(size_t)h]});
    }
    for (size_t k = 0; k < batch_c.size(); ++k)
      if (!is_hit[k]) advance(batch_c[k].pair);
    ph.windows += ph.lap();
    return 0;
  };

  if (speculate)
  {
    struct InFlight
    {
      int chunk;
      std::vector<dcp_hip_window> wins;
      std::vector<size_t> base; // first window of pair (p - p0) * nseq + s
    };
    std::deque<InFlight> flight;
    // whatever happens, no batch stays outstanding on the engine
    struct Drain
    {
      dcp_hip *eng;
      std::deque<InFlight> *flight;
      ~Drain()
      {
        int nh = 0;
        for (InFlight &f : *flight)
        {
          std::vector<int32_t> hw(f.wins.size() + 1);
          std::vector<float> hl(f.wins.size() + 1);
          (void)dcp_hip_cost_hits_end(eng, &nh, hw.data(), hl.data());
        }
      }
    } drain{x->eng, &flight};
    auto begin_chunk = [&](int c) -> int {
      int const p0 = chunks[(size_t)c].first, p1 = chunks[(size_t)c].second;
      InFlight f;
      f.chunk = c;
      f.base.reserve((size_t)(p1 - p0) * (size_t)nseq + 1);
      for (int pass = 0; pass < 2; ++pass) // count, then fill
      {
        size_t n = 0;
        for (int p = p0; p < p1; ++p)
        {
          int const K = dcp_hip_profile_core_size(x->eng, p);
          int last_len = -1;
          Chain const *ch = nullptr; // reads of one length follow each other more often than not
          chains_of_profile.clear();
          for (int s = 0; s < nseq; ++s)
          {
            int const len = (int)batch->seqs[(size_t)s].nt.size();
            if (len != last_len)
            {
              ch = nullptr;
              if (len > 0)
              {
                auto it = chains_of_profile.find(len);
                if (it == chains_of_profile.end())
                {
                  it = chains_of_profile.emplace(len, Chain()).first;
                  make_chain(len, K, it->second);
                }
                ch = &it->second;
              }
              last_len = len;
            }
            if (pass == 1)
            {
              f.base.push_back(n);
              if (ch)
              {
                dcp_hip_window *w = f.wins.data() + n;
                for (std::pair<int, int> const &r : *ch) *w++ = dcp_hip_window{p, s, r.first, r.second};
              }
            }
            n += ch ? ch->size() : 0;
          }
        }
        if (pass == 0) f.wins.resize(n);
      }
      f.base.push_back(f.wins.size());
      ph.windows += ph.lap();
      ++rounds;
      int brc = dcp_hip_cost_hits_begin(x->eng, (int)f.wins.size(), f.wins.data());
      if (brc) return raise(brc, __func__, dcp_hip_strerror(x->eng));
      size_t const nw = f.wins.size();
      flight.push_back(std::move(f));
      // while the GPU scores them: one callback per window scored (c-core/thread.c:74).  (A window of a pair that
      // hit earlier in its chain may turn out not to be the chain's -- it was scored all the same.)
      if (x->callback)
        for (size_t i = 0; i < nw && !x->interrupted; ++i) x->callback(x->userdata);
      ph.callbacks += ph.lap();
      return 0;
    };
    int next = 0;
    int const nchunks = (int)chunks.size();
    while (next < nchunks && flight.size() < 2 && !x->interrupted)
      if ((rc = begin_chunk(next++))) return rc;
    while (!flight.empty())
    {
      InFlight f = std::move(flight.front());
      flight.pop_front();
      int const p0 = chunks[(size_t)f.chunk].first, p1 = chunks[(size_t)f.chunk].second;
      std::vector<int32_t> hit_index(f.wins.size() + 1);
      std::vector<float> lrts(f.wins.size() + 1);
      int nh = 0;
      if ((rc = dcp_hip_cost_hits_end(x->eng, &nh, hit_index.data(), lrts.data())))
        return raise(rc, __func__, dcp_hip_strerror(x->eng));
      ph.cost += ph.lap();
      if (x->interrupted) continue; // (the batches still in flight are ended and dropped)
      if (next < nchunks && (rc = begin_chunk(next++))) return rc;
      size_t speculated_of_hit_pairs = 0, first_new = st.size();
      size_t last_pi = (size_t)-1;
      for (int h = 0; h < nh; ++h) // hit_index ascends: the hits of a pair are neighbours
      {
        size_t const wi = (size_t)hit_index[(size_t)h];
        size_t const pi = (size_t)(std::upper_bound(f.base.begin(), f.base.end(), wi) - f.base.begin()) - 1;
        if (pi != last_pi) // a pair's first hit: its chain and the (so far hit-less) scores of the chain's windows
        {
          last_pi = pi;
          int const p = p0 + (int)(pi / (size_t)nseq), sq = (int)(pi % (size_t)nseq);
          int const len = (int)batch->seqs[(size_t)sq].nt.size(), K = dcp_hip_profile_core_size(x->eng, p);
          kept_chains.emplace_back();
          make_chain(len, K, kept_chains.back());
          kept_lrt.emplace_back(f.base[pi + 1] - f.base[pi], -1.0f);
          st.push_back(PairState{p, sq, DcpWindow(len, K), &kept_chains.back(), kept_lrt.back().data()});
          speculated_of_hit_pairs += f.base[pi + 1] - f.base[pi];
        }
        kept_lrt.back()[wi - f.base[pi]] = lrts[(size_t)h];
      }
      nwindows += f.wins.size() - speculated_of_hit_pairs; // the windows of the pairs without a hit are final
      for (size_t i = first_new; i < st.size(); ++i) advance(i);
      ph.windows += ph.lap();
      // The path passes of the hits so far -- once no batch is in flight: beside a cost pass the path kernels, few
      // wavefronts bound by memory latency, take several times as long and hold the cost kernels up for as long
      // (whichever priority their streams have: profiles/r03_scan_pipeline.txt), so the scan gains nothing from the
      // overlap and a short one loses.  What needs scoring again waits for the end as well.
      while ((flight.empty() || path_beside) && !need_path.empty() && !x->interrupted)
        if ((rc = run_path_batch())) return rc;
      // the decoders of this chunk's profiles go with the chunk once its rows are under way (a memo of (K + 3) * 1364
      // bytes each; the formatter jobs hold their own references): a Pfam-sized database with hits on most profiles
      // would pin gigabytes.  (A pair of this chunk that hits again in the final rounds makes a new one.)
      for (int p = p0; p < p1; ++p) x->decoders[(size_t)p].reset();
      x->done_proteins += p1 - p0;
    }
  }
  else
  {
    for (int p = 0; p < nprof; ++p)
      for (int s = 0; s < nseq; ++s)
        if (!batch->seqs[(size_t)s].nt.empty())
          st.push_back(PairState{p, s, DcpWindow((int)batch->seqs[(size_t)s].nt.size(), dcp_hip_profile_core_size(x->eng, p)),
                                 nullptr, nullptr});
    for (size_t i = 0; i < st.size(); ++i) advance(i);
  }
  // the rounds of what is left: windows to score again (or, with nothing speculated, every window), their path passes
  while ((!need_cost.empty() || !need_path.empty()) && !x->interrupted)
  {
    if (!need_path.empty() && (rc = run_path_batch())) return rc;
    if (!need_cost.empty() && (rc = run_cost_batch())) return rc;
  }
  if (!speculate) x->done_proteins += nprof;
  for (int p = 0; p < nprof; ++p) x->decoders[(size_t)p].reset();

  formatters.join();
  if (decode_rc) return raise(decode_rc, __func__); // c-core/match.c:66-89 fails the scan the same way
  for (std::vector<Row> &part : formatted)
    for (Row &r : part) rows.push_back(std::move(r));
  ph.rows += ph.lap();
  // product_close (c-core/product.c:34-88): rows in profile, read, window order
  std::stable_sort(rows.begin(), rows.end(), [](Row const &a, Row const &b) {
    if (a.profile != b.profile) return a.profile < b.profile;
    if (a.seq != b.seq) return a.seq < b.seq;
    return a.window < b.window;
  });
  std::string const file = dir + "/products.tsv";
  FILE *fp = fopen(file.c_str(), "wb");
  if (!fp) return raise(DCP_EFOPEN, __func__, file.c_str());
  bool ok = fputs("sequence\twindow\twindow_start\twindow_stop\thit\thit_start\thit_stop\tprofile\tabc\tlrt\tevalue\tmatch\n",
                  fp) >= 0;
  x->products.reserve(rows.size());
  for (Row &r : rows)
  {
    ok = ok && fwrite(r.text.data(), 1, r.text.size(), fp) == r.text.size() && fputc('\n', fp) != EOF;
    x->products.push_back(std::move(r.text));
  }
  if (fclose(fp) != 0 || !ok) return raise(DCP_EWRITEPROD, __func__, file.c_str());
  ph.write += ph.lap();
  {
    // (the progress callbacks of a batch are made while the GPU scores it: they count as cost pass)
    double const t[DCP_SCAN_TIMING_VALUES] = {ph.total(), ph.reads, ph.windows, ph.cost + ph.callbacks, ph.path, ph.rows, ph.write,
                                              (double)rounds, (double)nwindows, (double)nhits};
    memcpy(x->timing, t, sizeof t);
  }
  if (getenv("DECIPHON_HIP_TIMING"))
    fprintf(stderr,
            "dcp_scan_run: %d rounds, %zu windows, %zu path passes; windows %.3f s, cost pass %.3f s, path pass %.3f s, "
            "rows %.3f s, products.tsv %.3f s; of the cost pass %.3f s in progress callbacks\n",
            rounds, nwindows, nhits, ph.windows, ph.cost + ph.callbacks, ph.path, ph.rows, ph.write, ph.callbacks);
  return 0;
}

void dcp_scan_interrupt(struct dcp_scan *x)
{
  if (x) x->interrupted = true;
}

int dcp_scan_progress(struct dcp_scan const *x)
{
  if (!x || x->num_proteins <= 0) return 0;
  return (100 * x->done_proteins.load()) / x->num_proteins; // c-core/scan.c:224-227
}

long dcp_scan_num_products(struct dcp_scan const *x) { return x ? (long)x->products.size() : 0; }

int dcp_scan_last_timing(struct dcp_scan const *x, double *out, int n)
{
  if (!x || (n > 0 && !out)) return 0;
  for (int i = 0; i < n && i < DCP_SCAN_TIMING_VALUES; ++i) out[i] = x->timing[i];
  return DCP_SCAN_TIMING_VALUES;
}

char const *dcp_scan_product(struct dcp_scan const *x, long i)
{
  if (!x || i < 0 || i >= (long)x->products.size()) return nullptr;
  return x->products[(size_t)i].c_str();
}

struct dcp_batch *dcp_batch_new(void) { return new (std::nothrow) dcp_batch; }

void dcp_batch_del(struct dcp_batch *x) { delete x; }

int dcp_batch_add(struct dcp_batch *x, long id, char const *name, char const *data)
{
  if (!x || !name || !data) return raise(DCP_EFUNCUSE, __func__);
  dcp_batch::Seq s;
  s.id = id;
  s.name = name;
  size_t const n = strlen(data);
  s.nt.resize(n);
  int rc = dcp_encode_sequence(data, (int64_t)n, s.nt.data());
  if (rc) return raise(rc, __func__);
  s.text.resize(n);
  for (size_t i = 0; i < n; ++i) s.text[i] = "ACGT"[s.nt[i]];
  // the reference keeps U as U in the stored text (c-core/disambiguate.c:14-21)
  for (size_t i = 0; i < n; ++i)
  {
    s.has_u = s.has_u || data[i] == 'U' || data[i] == 'u';
    s.has_t = s.has_t || data[i] == 'T' || data[i] == 't';
  }
  if (s.has_u)
    for (size_t i = 0; i < n; ++i)
      if (s.text[i] == 'T') s.text[i] = 'U';
  x->seqs.push_back(std::move(s));
  return 0;
}

void dcp_batch_reset(struct dcp_batch *x)
{
  if (x) x->seqs.clear();
}

// ---- press: not provided (needs third-party imm + hmmer_reader) ----------------
struct dcp_press *dcp_press_new(void) { return new (std::nothrow) dcp_press; }
int dcp_press_setup(struct dcp_press *, int, float) { return raise(DCP_EFUNCUSE, __func__, "press is not part of this build"); }
int dcp_press_open(struct dcp_press *, char const *, char const *) { return raise(DCP_EFUNCUSE, __func__, "press is not part of this build"); }
long dcp_press_nproteins(struct dcp_press const *) { return 0; }
int dcp_press_next(struct dcp_press *) { return raise(DCP_EFUNCUSE, __func__, "press is not part of this build"); }
bool dcp_press_end(struct dcp_press const *) { return true; }
int dcp_press_close(struct dcp_press *) { return 0; }
void dcp_press_del(struct dcp_press const *x) { delete const_cast<dcp_press *>(x); }

} // extern "C"
