// host_logic.cpp -- see host_logic.h
#include "host_logic.h"
#include "dcp_errors.h"

#include <initializer_list>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <utility>

void dcp_xtrans(int seq_size, bool multi_hits, bool hmmer3_compat, float xt[DCP_NUM_XTRANS])
{
  // The reference evaluates these with double log() on float operands and stores
  // floats (c-core/xtrans.c:26-45); the same expression shapes are kept so the
  // roundings agree.
  float const L = (float)seq_size;
  float q = 0.0f;
  float log_q = -INFINITY;
  if (multi_hits)
  {
    q = 0.5f;
    log_q = (float)log(0.5);
  }
  // C++ would pick the float overload of log() for a float argument; the reference
  // is C, where log() is the double function, so widen explicitly.
  float const denom = L + 2 + q / (1 - q);
  float const two_q = 2 + q / (1 - q);
  float const Lp1 = L + 1;
  float const lp = (float)(log((double)L) - log((double)denom));
  float const l1p = (float)(log((double)two_q) - log((double)denom));
  float const lr = (float)(log((double)L) - log((double)Lp1));

  float NN = lp, CC = lp, JJ = lp;
  float const NB = l1p, CT = l1p, JB = l1p, RR = lr;
  float const EJ = log_q;
  float const one_q = 1 - q;
  float const EC = (float)log((double)one_q);
  if (hmmer3_compat) NN = CC = JJ = logf(1);

  // c-core/xtrans.c:53-68
  xt[DCP_RR] = -RR;
  xt[DCP_SN] = -0 - NN;
  xt[DCP_NN] = -NN;
  xt[DCP_SB] = -0 - NB;
  xt[DCP_NB] = -NB;
  xt[DCP_EB] = -EJ - JB;
  xt[DCP_JB] = -JB;
  xt[DCP_EJ] = -EJ - JJ;
  xt[DCP_JJ] = -JJ;
  xt[DCP_EC] = -EC - CC;
  xt[DCP_CC] = -CC;
  xt[DCP_ET] = -EC - CT;
  xt[DCP_CT] = -CT;
}

void dcp_setup_profile(int K, int Kp, float const *node_trans, float const *node_emission, float const *BMk,
                       float const *null_lprob, float const *bg_lprob, float *trans, float *rows)
{
  for (size_t i = 0; i < (size_t)DCP_NUM_TRANS * Kp; ++i) trans[i] = INFINITY; // viterbi_setup fills +inf
  for (int k = 0; k < K; ++k) trans[DCP_BM * Kp + k] = -BMk[k];
  for (int k = 0; k + 1 < K; ++k)
  {
    float const *t = node_trans + 7 * (size_t)k; // MM MI MD IM II DM DD, c-core/trans.h:8-27
    trans[DCP_MM * Kp + k + 1] = -t[0];
    trans[DCP_MI * Kp + k] = -t[1];
    trans[DCP_MD * Kp + k + 1] = -t[2];
    trans[DCP_IM * Kp + k + 1] = -t[3];
    trans[DCP_II * Kp + k] = -t[4];
    trans[DCP_DM * Kp + k + 1] = -t[5];
    trans[DCP_DD * Kp + k + 1] = -t[6];
  }
  trans[DCP_MI * Kp + K - 1] = INFINITY;
  trans[DCP_II * Kp + K - 1] = INFINITY;
  // node-major [k][code] on disk -> code-major [code][k] rows for the kernels
  size_t const stride = (size_t)Kp + DCP_ROW_HDR;
  for (int c = 0; c < DCP_TABLE_SIZE; ++c)
  {
    float *hdr = rows + (size_t)c * stride;
    hdr[0] = -null_lprob[c];
    hdr[1] = -bg_lprob[c];
    hdr[2] = hdr[3] = 0.0f;
    float *row = hdr + DCP_ROW_HDR;
    for (int k = 0; k < K; ++k) row[k] = -node_emission[(size_t)k * DCP_TABLE_SIZE + c];
    for (int k = K; k < Kp; ++k) row[k] = INFINITY;
  }
}

int dcp_encode_sequence(char const *data, int64_t n, uint8_t *out)
{
  enum { A, C, G, T, U, NSYM };
  int64_t count[NSYM] = {0};
  for (int64_t i = 0; i < n; ++i)
  {
    switch (data[i] & ~0x20) // ASCII letters: clear the lowercase bit
    {
    case 'A': count[A]++; break;
    case 'C': count[C]++; break;
    case 'G': count[G]++; break;
    case 'T': count[T]++; break;
    case 'U': count[U]++; break;
    default: break;
    }
  }
  if (count[T] > 0 && count[U] > 0) return DCP_ENUCLTSEQTU;

  // IUPAC ambiguity codes resolve to the most frequent member base of THIS
  // sequence, the first listed winning ties (c-core/disambiguate.c:23-35,55-83)
  auto pick = [&](std::initializer_list<int> set) {
    int best = *set.begin();
    for (int s : set)
      if (count[s] > count[best]) best = s;
    return best;
  };
  int rc = 0;
  for (int64_t i = 0; i < n; ++i)
  {
    char const ch = data[i];
    bool const letter = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z');
    int sym = -1;
    switch (letter ? (ch & ~0x20) : 0)
    {
    case 'A': sym = A; break;
    case 'C': sym = C; break;
    case 'G': sym = G; break;
    case 'T': sym = T; break;
    case 'U': sym = U; break;
    case 'R': sym = pick({A, G}); break;
    case 'Y': sym = pick({C, T}); break;
    case 'M': sym = pick({A, C}); break;
    case 'K': sym = pick({G, T}); break;
    case 'S': sym = pick({C, G}); break;
    case 'W': sym = pick({A, T}); break;
    case 'H': sym = pick({A, C, T}); break;
    case 'B': sym = pick({C, G, T}); break;
    case 'V': sym = pick({A, C, G}); break;
    case 'D': sym = pick({A, G, T}); break;
    case 'N': sym = pick({A, C, G, T}); break;
    case 'X': sym = pick({A, C, G, T}); break;
    default: break;
    }
    if (sym < 0)
    {
      rc = DCP_ESEQABC;
      out[i] = 0;
    }
    else
      out[i] = (uint8_t)(sym == U ? 3 : sym);
  }
  return rc;
}

namespace
{
enum
{
  ST_M = 0 << 14, ST_I = 1 << 14, ST_D = 2 << 14, ST_X = 3 << 14, // c-core/state.h:9-25
  ST_S = ST_X | 3, ST_N = ST_X | 4, ST_B = ST_X | 5, ST_E = ST_X | 6, ST_J = ST_X | 7, ST_C = ST_X | 8, ST_T = ST_X | 9,
};
inline int msb(int id) { return id & (3 << 14); }
inline bool is_core(int id) { return msb(id) != ST_X; }
inline int core_idx(int id) { return (id & 0x3FFF) - 1; }
} // namespace

int dcp_unzip(int K, int L, uint32_t const *xnodes, uint16_t const *nodes, std::vector<int32_t> &state_ids,
              std::vector<int32_t> &seqsizes)
{
  size_t const first = state_ids.size();
  int state = ST_T; // state_make_end()
  int stage = L;
  // a valid trellis walks at most (L+1)*(K+4) steps; anything longer is corrupt
  int64_t const limit = ((int64_t)L + 1) * ((int64_t)K + 4) + 8;
  int64_t steps = 0;
  while (state != ST_S || stage)
  {
    if (++steps > limit) return DCP_EINVALSTATE;
    int size = 0, prev = 0;
    if (!is_core(state))
    {
      uint32_t const x = xnodes[stage];
      // field offsets/widths: c-core/trellis.h:42-56, c-core/state.h:27-39
      switch (state)
      {
      case ST_N: { unsigned v = x & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_N : ST_S; break; }
      case ST_B: { unsigned v = (x >> 4) & 0x3; static int const from[4] = {ST_S, ST_N, ST_E, ST_J}; prev = from[v]; break; }
      case ST_E: { unsigned v = (x >> 6) & 0x7FFF; prev = (v & 1 ? ST_D : ST_M) | (int)(v / 2 + 1); break; }
      case ST_C: { unsigned v = (x >> 21) & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_C : ST_E; break; }
      case ST_T: { unsigned v = (x >> 25) & 0x1; prev = v ? ST_C : ST_E; break; }
      case ST_J: { unsigned v = (x >> 26) & 0xF; size = (int)(v % 5) + 1; prev = v / 5 ? ST_J : ST_E; break; }
      default: return DCP_EINVALSTATE;
      }
    }
    else
    {
      int const idx = core_idx(state);
      if (idx < 0 || idx >= K) return DCP_EINVALSTATE;
      uint16_t const w = nodes[(size_t)stage * (size_t)K + (size_t)idx];
      if (msb(state) == ST_M)
      {
        unsigned v = w & 0x1F;
        size = (int)(v % 5) + 1;
        unsigned s = v / 5;
        if (s == 0) prev = ST_B;
        else
        {
          if (idx <= 0) return DCP_EINVALSTATE; // BUG_ON(idx <= 0), c-core/trellis.c:72
          prev = (s == 1 ? ST_M : s == 2 ? ST_I : ST_D) | idx;
        }
      }
      else if (msb(state) == ST_D)
      {
        unsigned v = (w >> 5) & 0x1;
        if (idx <= 0) return DCP_EINVALSTATE;
        prev = (v ? ST_D : ST_M) | idx;
      }
      else
      {
        unsigned v = (w >> 6) & 0xF;
        size = (int)(v % 5) + 1;
        prev = (v / 5 ? ST_I : ST_M) | (idx + 1);
      }
    }
    state_ids.push_back(state);
    seqsizes.push_back(size);
    state = prev;
    stage -= size;
    if (stage < 0) return DCP_EINVALSTATE;
  }
  state_ids.push_back(state);
  seqsizes.push_back(0);
  for (size_t i = first, j = state_ids.size() - 1; i < j; ++i, --j) // imm_path_reverse
  {
    std::swap(state_ids[i], state_ids[j]);
    std::swap(seqsizes[i], seqsizes[j]);
  }
  return 0;
}

void dcp_state_name(int id, char name[8])
{
  if (msb(id) == ST_X)
  {
    static char const letters[] = "FRGSNBEJCT";
    int const n = id & 0x3FFF;
    name[0] = n < 10 ? letters[n] : '?';
    name[1] = 0;
    return;
  }
  name[0] = msb(id) == ST_M ? 'M' : msb(id) == ST_I ? 'I' : 'D';
  snprintf(name + 1, 7, "%d", core_idx(id) + 1);
}

bool dcp_state_is_mute(int id)
{
  if (msb(id) == ST_X) return id == ST_S || id == ST_B || id == ST_E || id == ST_T;
  return msb(id) == ST_D;
}

namespace
{
// c-core/thread.c:130-160 (the span between the first B and the last E of the path), over any view of the steps
template <class Id, class Size> bool find_hit(int n, Id id, Size size, DcpHit &hit)
{
  int it = 0, pos = 0;
  while (it < n && id(it) != ST_B) pos += size(it++);
  if (it >= n) return false;
  hit.hit_start = pos;
  hit.begin_step = it;
  int end = it + 1;
  int stop = pos;
  for (;;)
  {
    it = end;
    hit.hit_stop = stop;
    while (it < n && id(it) != ST_E) stop += size(it++);
    if (it >= n) break;
    end = it + 1;
  }
  hit.end_step = end;
  hit.last_hit_pos = hit.hit_stop - 1;
  return true;
}
} // namespace

bool dcp_find_hit(std::vector<int32_t> const &ids, std::vector<int32_t> const &sizes, DcpHit &hit)
{
  return find_hit((int)ids.size(), [&](int i) { return ids[(size_t)i]; }, [&](int i) { return sizes[(size_t)i]; }, hit);
}

bool dcp_find_hit_packed(uint32_t const *steps, int n, DcpHit &hit)
{
  return find_hit(n, [&](int i) { return (int)(steps[i] & 0xffffu); }, [&](int i) { return (int)(steps[i] >> 16); }, hit);
}

bool DcpWindow::next()
{
  if (stop == seq_size) return false;
  int const stop_miss = stop + 1;
  int start_miss = start + 1;
  if (start + last_hit_pos + 1 > start_miss) start_miss = start + last_hit_pos + 1;
  if (stop_miss - core_size * 4 > start_miss) start_miss = stop_miss - core_size * 4;
  start = start_miss;
  int const span = core_size * 50 < 100000 ? core_size * 50 : 100000;
  stop = start_miss + span;
  if (stop > seq_size) stop = seq_size;
  idx += 1;
  return true;
}

// c-core/error.c:10-101
char const *dcp_error_string(int code)
{
  static char const *const msg[] = {
      nullptr,
      "different alphabets", "failed to close file", "invalid file data", "failed to re-open file",
      "failed to read from file", "failed to seek file", "failed to get file position", "invalid function usage",
      "failed to write to file", "failed to get file path", "zero-length sequence", "zero-length model",
      "no partition", "failed to decode into codon", "model is too large", "protein is too large",
      "failed to read hmmer3 profile", "too may partitions", "too many transitions", "not enough memory",
      "failed to open DB file", "failed to open HMM file", "failed to open temporary file", "truncated file path",
      "failed to unpack DP", "failed to pack DP", "failed to unpack nuclt dist", "failed to pack nuclt dist",
      "failed to set transition", "failed to add state", "failed to reset DP", "failed to get file stat",
      "failed to open file", "file is too large", "path is too long", "failed to reset task",
      "failed to create new task", "failed to setup task", "failed to write product", "invalid partition",
      "accession string is too long", "too many threads", "failed to create temporary file", "failed to flush file",
      "failed to create directory", "wrong file format", "failed to remove directory", "failed to remove file",
      "must set gencode first", "invalid gencode id", "dialing to hmmer daemon failed",
      "failed to put a task to the hmmer daemon", "failed to pop a task from the hmmer daemon",
      "failed to pack hmmer result", "reached maximum number of retries on hmmer daemon",
      "failed to warmup hmmer daemon", "invalid sequence letter (neither DNA nor RNA alphabet)",
      "failed to open file descriptor", "failed to make temporary file", "abc string is too long",
      "consensus string is too long", nullptr /* DCP_ENOTDIALED has no message in the reference */,
      "number of core nodes is too long", "invalid state", "invalid size", "unexpected end of file",
      "unexpected end of nodes", "unsupported database version", "not a database file", "invalid state id",
      "unsupported nucleotide (must be either DNA or RNA)", "database is DNA but sequence is RNA",
      "database is RNA but sequence is DNA", "nucleotide sequence cannot have both U and T", "failed to find hit",
      "failed to open file", "failed to close file", "failed to duplicate descriptor", "too many proteins",
      "invalid number of proteins",
  };
  int const n = (int)(sizeof(msg) / sizeof(msg[0]));
  if (code > 0 && code < n) return msg[code];
  static thread_local char unknown[32];
  snprintf(unknown, sizeof unknown, "unknown error #%d", code);
  return unknown;
}

void dcp_partition_bounds(int n, int32_t const *core_sizes, int nparts, bool balanced, int32_t *first)
{
  first[0] = 0;
  if (!balanced || !core_sizes)
  {
    for (int p = 0; p < nparts; ++p)
    {
      int const left = n - p > 0 ? n - p : 0;
      first[p + 1] = first[p] + (left + nparts - 1) / nparts; // ceil((n - p) / nparts), c-core/partition_size.c:13-16
    }
    return;
  }
  std::vector<double> cum((size_t)n + 1, 0.0);
  for (int i = 0; i < n; ++i) cum[(size_t)i + 1] = cum[(size_t)i] + (double)core_sizes[i];
  int j = 0;
  for (int p = 1; p < nparts; ++p)
  {
    double const target = cum[(size_t)n] * (double)p / (double)nparts;
    while (j < n && cum[(size_t)j + 1] <= target) ++j;                                  // cum[j] <= target < cum[j + 1]
    if (j < n && target - cum[(size_t)j] > cum[(size_t)j + 1] - target) ++j; // the nearer of the two
    first[p] = j;
  }
  first[nparts] = n;
}

// ---- quasi-codon decoding (see host_logic.h) ----
namespace
{

// how many single-base deletions of codon x leave the pair (a, b)
inline int del1(uint8_t const x[3], int a, int b) { return (x[1] == a && x[2] == b) + (x[0] == a && x[2] == b) + (x[0] == a && x[1] == b); }
inline int has(uint8_t const x[3], int a) { return (x[0] == a) + (x[1] == a) + (x[2] == a); }

double frag_given_codon(double e, double const p[4], uint8_t const x[3], uint8_t const *z, int n)
{
  double const f = 1.0 - e;
  switch (n)
  {
  case 1: return e * e * f * f / 3.0 * has(x, z[0]);
  case 2:
    return 2.0 * e * f * f * f / 3.0 * del1(x, z[0], z[1]) +
           e * e * e * f / 3.0 * (p[z[0]] * has(x, z[1]) + p[z[1]] * has(x, z[0]));
  case 3:
  {
    double v = f * f * f * f * (x[0] == z[0] && x[1] == z[1] && x[2] == z[2]);
    v += 4.0 * e * e * f * f / 9.0 *
         (p[z[0]] * del1(x, z[1], z[2]) + p[z[1]] * del1(x, z[0], z[2]) + p[z[2]] * del1(x, z[0], z[1]));
    return v + e * e * e * e * p[z[0]] * p[z[1]] * p[z[2]];
  }
  case 4:
  {
    double one = 0, two = 0;
    for (int j = 0; j < 4; ++j)
    {
      uint8_t r[3];
      for (int t = 0, k = 0; t < 4; ++t)
        if (t != j) r[k++] = z[t];
      one += p[z[j]] * (x[0] == r[0] && x[1] == r[1] && x[2] == r[2]);
    }
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j)
      {
        uint8_t r[2];
        for (int t = 0, k = 0; t < 4; ++t)
          if (t != i && t != j) r[k++] = z[t];
        two += p[z[i]] * p[z[j]] * del1(x, r[0], r[1]);
      }
    return e * f * f * f / 2.0 * one + e * e * e * f / 9.0 * two;
  }
  case 5:
  {
    double v = 0;
    for (int i = 0; i < 5; ++i)
      for (int j = i + 1; j < 5; ++j)
      {
        uint8_t r[3];
        for (int t = 0, k = 0; t < 5; ++t)
          if (t != i && t != j) r[k++] = z[t];
        v += p[z[i]] * p[z[j]] * (x[0] == r[0] && x[1] == r[1] && x[2] == r[2]);
      }
    return e * e * f * f / 10.0 * v;
  }
  default: return 0.0;
  }
}

} // namespace

bool dcp_decode_codon_prob(double epsilon, double const p[4], double const prior[64], uint8_t const *z, int n,
                           uint8_t codon[3])
{
  if (n < 1 || n > 5) return false;
  if (n == 3)
  {
    // The common step is an exact codon.  Any other codon x reaches z only through an indel pair or two, so
    // P(x) P(z | x) <= max P * (4 e^2 (1-e)^2 / 9 * 3 (p(z1) + p(z2) + p(z3)) + e^4 p(z1) p(z2) p(z3)): when
    // P(z) (1-e)^4 alone exceeds that bound, z is the strict maximum and the 64-codon search is not needed.
    double const e = epsilon, f = 1.0 - e;
    double pmax = 0.0;
    for (int i = 0; i < 64; ++i) pmax = prior[i] > pmax ? prior[i] : pmax;
    double const self = prior[z[0] * 16 + z[1] * 4 + z[2]] * (f * f * f * f);
    double const others = pmax * (4.0 * e * e * f * f / 9.0 * 3.0 * (p[z[0]] + p[z[1]] + p[z[2]]) + e * e * e * e * p[z[0]] * p[z[1]] * p[z[2]]);
    if (self > others * (1.0 + 1e-9))
    {
      codon[0] = z[0];
      codon[1] = z[1];
      codon[2] = z[2];
      return true;
    }
  }
  double best = 0.0;
  bool found = false;
  for (uint8_t a = 0; a < 4; ++a)
    for (uint8_t b = 0; b < 4; ++b)
      for (uint8_t c = 0; c < 4; ++c)
      {
        double const px = prior[a * 16 + b * 4 + c];
        if (!(px > 0.0)) continue;
        uint8_t const x[3] = {a, b, c};
        double const joint = px * frag_given_codon(epsilon, p, x, z, n);
        if (joint > best)
        {
          best = joint;
          codon[0] = a;
          codon[1] = b;
          codon[2] = c;
          found = true;
        }
      }
  return found;
}

bool dcp_decode_codon(float epsilon, float const nucltp[4], float const codonm[125], uint8_t const *z, int n,
                      uint8_t codon[3])
{
  double p[4], prior[64];
  for (int i = 0; i < 4; ++i) p[i] = exp((double)nucltp[i]);
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      for (int c = 0; c < 4; ++c) prior[a * 16 + b * 4 + c] = exp((double)codonm[a * 25 + b * 5 + c]);
  return dcp_decode_codon_prob((double)epsilon, p, prior, z, n, codon);
}

char dcp_gencode_amino(int id, uint8_t const codon[3])
{
  // NCBI translation tables, amino acids in TCAG order of the three codon positions
  struct Table { int id; char const *aa; };
  static Table const tables[] = {
      {1, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {2, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG"},
      {3, "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {4, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {5, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG"},
      {6, "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {9, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"},
      {10, "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {11, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
      {12, "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
  };
  static int const tcag[4] = {2, 1, 3, 0}; // our indices A, C, G, T -> position in T, C, A, G
  for (Table const &t : tables)
    if (t.id == id) return t.aa[tcag[codon[0]] * 16 + tcag[codon[1]] * 4 + tcag[codon[2]]];
  return 0;
}
