// dcp_db.cpp -- see dcp_db.h
#include "dcp_db.h"
#include <math.h>
#include "dcp_errors.h"
#include "dcp_types.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace
{

// ---- a cursor over MessagePack bytes ------------------------------------------
struct Cur
{
  uint8_t const *p;
  uint8_t const *end;
  bool ok = true;

  bool need(size_t n)
  {
    if ((size_t)(end - p) < n) ok = false;
    return ok;
  }
  uint64_t be(int n)
  {
    if (!need((size_t)n)) return 0;
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
    p += n;
    return v;
  }
  uint8_t peek() { return need(1) ? *p : 0; }
};

enum Kind { K_NIL, K_BOOL, K_INT, K_FLOAT, K_STR, K_BIN, K_EXT, K_ARRAY, K_MAP, K_BAD };

struct Tok
{
  Kind kind = K_BAD;
  int64_t i = 0;        // int / bool value, or element count of array/map
  double f = 0;         // float value
  uint8_t const *data = nullptr; // payload of str/bin/ext
  uint32_t len = 0;
  int ext_type = 0;
};

// reads one token; for str/bin/ext the payload is skipped over (and referenced),
// for array/map only the header is consumed
Tok next(Cur &c)
{
  Tok t;
  if (!c.need(1)) return t;
  uint8_t b = *c.p++;
  auto payload = [&](Kind k, uint32_t n) {
    t.kind = k;
    t.len = n;
    if (c.need(n))
    {
      t.data = c.p;
      c.p += n;
    }
    else
      t.kind = K_BAD;
  };
  if (b <= 0x7f) { t.kind = K_INT; t.i = b; }
  else if (b >= 0xe0) { t.kind = K_INT; t.i = (int8_t)b; }
  else if ((b & 0xf0) == 0x80) { t.kind = K_MAP; t.i = b & 0x0f; }
  else if ((b & 0xf0) == 0x90) { t.kind = K_ARRAY; t.i = b & 0x0f; }
  else if ((b & 0xe0) == 0xa0) payload(K_STR, b & 0x1f);
  else switch (b)
  {
  case 0xc0: t.kind = K_NIL; break;
  case 0xc2: t.kind = K_BOOL; t.i = 0; break;
  case 0xc3: t.kind = K_BOOL; t.i = 1; break;
  case 0xc4: payload(K_BIN, (uint32_t)c.be(1)); break;
  case 0xc5: payload(K_BIN, (uint32_t)c.be(2)); break;
  case 0xc6: payload(K_BIN, (uint32_t)c.be(4)); break;
  case 0xc7: { uint32_t n = (uint32_t)c.be(1); t.ext_type = (int8_t)c.be(1); payload(K_EXT, n); break; }
  case 0xc8: { uint32_t n = (uint32_t)c.be(2); t.ext_type = (int8_t)c.be(1); payload(K_EXT, n); break; }
  case 0xc9: { uint32_t n = (uint32_t)c.be(4); t.ext_type = (int8_t)c.be(1); payload(K_EXT, n); break; }
  case 0xca: { uint32_t u = (uint32_t)c.be(4); float f; memcpy(&f, &u, 4); t.kind = K_FLOAT; t.f = f; break; }
  case 0xcb: { uint64_t u = c.be(8); double d; memcpy(&d, &u, 8); t.kind = K_FLOAT; t.f = d; break; }
  case 0xcc: t.kind = K_INT; t.i = (int64_t)c.be(1); break;
  case 0xcd: t.kind = K_INT; t.i = (int64_t)c.be(2); break;
  case 0xce: t.kind = K_INT; t.i = (int64_t)c.be(4); break;
  case 0xcf: t.kind = K_INT; t.i = (int64_t)c.be(8); break;
  case 0xd0: t.kind = K_INT; t.i = (int8_t)c.be(1); break;
  case 0xd1: t.kind = K_INT; t.i = (int16_t)c.be(2); break;
  case 0xd2: t.kind = K_INT; t.i = (int32_t)c.be(4); break;
  case 0xd3: t.kind = K_INT; t.i = (int64_t)c.be(8); break;
  case 0xd4: t.ext_type = (int8_t)c.be(1); payload(K_EXT, 1); break;
  case 0xd5: t.ext_type = (int8_t)c.be(1); payload(K_EXT, 2); break;
  case 0xd6: t.ext_type = (int8_t)c.be(1); payload(K_EXT, 4); break;
  case 0xd7: t.ext_type = (int8_t)c.be(1); payload(K_EXT, 8); break;
  case 0xd8: t.ext_type = (int8_t)c.be(1); payload(K_EXT, 16); break;
  case 0xd9: payload(K_STR, (uint32_t)c.be(1)); break;
  case 0xda: payload(K_STR, (uint32_t)c.be(2)); break;
  case 0xdb: payload(K_STR, (uint32_t)c.be(4)); break;
  case 0xdc: t.kind = K_ARRAY; t.i = (int64_t)c.be(2); break;
  case 0xdd: t.kind = K_ARRAY; t.i = (int64_t)c.be(4); break;
  case 0xde: t.kind = K_MAP; t.i = (int64_t)c.be(2); break;
  case 0xdf: t.kind = K_MAP; t.i = (int64_t)c.be(4); break;
  default: t.kind = K_BAD; break;
  }
  if (!c.ok) t.kind = K_BAD;
  return t;
}

// skips one complete value.  Iterative, with a count of values still owed: a file of nested
// array headers cannot take the stack, it just runs into the end of the buffer.
bool skip(Cur &c)
{
  uint64_t pending = 1;
  while (pending)
  {
    Tok t = next(c);
    if (t.kind == K_BAD) return false;
    --pending;
    uint64_t const more = t.kind == K_ARRAY ? (uint64_t)t.i : t.kind == K_MAP ? 2 * (uint64_t)t.i : 0;
    if (more > (uint64_t)1 << 40) return false; // no .dcp value holds that many items
    pending += more;
  }
  return true;
}

bool expect_key(Cur &c, char const *key) // c-core/expect.c:8-22
{
  Tok t = next(c);
  return t.kind == K_STR && t.len == strlen(key) && memcmp(t.data, key, t.len) == 0;
}

bool expect_map(Cur &c, int64_t n) // c-core/expect.c:24-30
{
  Tok t = next(c);
  return t.kind == K_MAP && t.i == n;
}

bool read_str(Cur &c, std::string &out, size_t max)
{
  Tok t = next(c);
  if (t.kind != K_STR || t.len >= max) return false;
  out.assign((char const *)t.data, t.len);
  return true;
}

bool read_int(Cur &c, int64_t &v)
{
  Tok t = next(c);
  if (t.kind != K_INT) return false;
  v = t.i;
  return true;
}

// c-core/read.c:118-132 (read_f32array) plus the legacy encoding
bool read_f32array(Cur &c, size_t n, float *out)
{
  Tok t = next(c);
  if (t.len != n * sizeof(float)) return false;
  if (t.kind == K_BIN)
  {
    memcpy(out, t.data, t.len); // native (little-endian) floats
    return true;
  }
  if (t.kind == K_EXT && t.ext_type == 8)
  {
    for (size_t i = 0; i < n; ++i)
    {
      uint8_t const *b = t.data + 4 * i;
      uint32_t u = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
      memcpy(out + i, &u, 4);
    }
    return true;
  }
  return false;
}

// imm_abc_pack (third-party imm): map(4){symbols, idx, any_symbol_id, typeid}
bool read_abc(Cur &c, std::string &symbols, int &typeid_)
{
  Tok m = next(c);
  if (m.kind != K_MAP) return false;
  for (int64_t i = 0; i < m.i; ++i)
  {
    Tok k = next(c);
    if (k.kind != K_STR) return false;
    std::string key((char const *)k.data, k.len);
    if (key == "symbols")
    {
      if (!read_str(c, symbols, 64)) return false;
    }
    else if (key == "typeid")
    {
      int64_t v;
      if (!read_int(c, v)) return false;
      typeid_ = (int)v;
    }
    else if (!skip(c))
      return false;
  }
  return true;
}

} // namespace

DcpDbReader::~DcpDbReader() { close(); }

void DcpDbReader::close()
{
  if (data_) munmap((void *)data_, size_);
  if (fd_ >= 0) ::close(fd_);
  data_ = nullptr;
  size_ = 0;
  fd_ = -1;
  header_ = DcpDbHeader();
  offsets_.clear();
}

int DcpDbReader::open(char const *path)
{
  close();
  fd_ = ::open(path, O_RDONLY);
  if (fd_ < 0) return DCP_EOPENDB;
  struct stat st;
  if (fstat(fd_, &st) != 0 || st.st_size <= 0)
  {
    close();
    return DCP_EFSTAT;
  }
  size_ = (size_t)st.st_size;
  void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
  if (m == MAP_FAILED)
  {
    data_ = nullptr;
    close();
    return DCP_EFREAD;
  }
  data_ = (uint8_t const *)m;

  Cur c{data_, data_ + size_};
  auto fail = [&](int rc) {
    close();
    return rc;
  };
  // c-core/database_reader.c:38-71
  if (!expect_map(c, 2)) return fail(DCP_EFDATA);
  if (!expect_key(c, "header")) return fail(DCP_EFDATA);
  if (!expect_map(c, 8)) return fail(DCP_EFDATA);

  int64_t v = 0;
  if (!expect_key(c, "magic_number") || !read_int(c, v)) return fail(DCP_EFDATA);
  header_.magic_number = (int)v;
  if (header_.magic_number != 0xC6F1) return fail(DCP_ENOTDBFILE); // c-core/magic_number.h:4
  if (!expect_key(c, "version") || !read_int(c, v)) return fail(DCP_EFDATA);
  header_.version = (int)v;
  if (header_.version != 1) return fail(DCP_EDBVERSION); // c-core/database_version.h:4
  if (!expect_key(c, "entry_dist") || !read_int(c, v)) return fail(DCP_EFDATA);
  header_.entry_dist = (int)v;
  if (header_.entry_dist != 1 && header_.entry_dist != 2) return fail(DCP_EFDATA); // c-core/entry_dist.h:6-11
  if (!expect_key(c, "epsilon")) return fail(DCP_EFDATA);
  {
    Tok t = next(c);
    if (t.kind != K_FLOAT) return fail(DCP_EFDATA);
    header_.epsilon = (float)t.f;
    if (header_.epsilon < 0 || header_.epsilon > 1) return fail(DCP_EFDATA);
  }
  if (!expect_key(c, "abc") || !read_abc(c, header_.abc_symbols, header_.abc_typeid))
    return fail(DCP_ENUCLTDUNPACK);
  int amino_typeid = 0;
  if (!expect_key(c, "amino") || !read_abc(c, header_.amino_symbols, amino_typeid))
    return fail(DCP_ENUCLTDUNPACK);
  if (!expect_key(c, "has_ga")) return fail(DCP_EFDATA);
  {
    Tok t = next(c);
    if (t.kind != K_BOOL) return fail(DCP_EFDATA);
    header_.has_ga = t.i != 0;
  }
  if (!expect_key(c, "protein_sizes")) return fail(DCP_EFDATA);
  {
    Tok t = next(c);
    if (t.kind == K_ARRAY) // c-core/database_reader.c:103-130
    {
      if (t.i > INT32_MAX) return fail(DCP_EFDATA);
      header_.protein_sizes.resize((size_t)t.i);
      for (int64_t i = 0; i < t.i; ++i)
      {
        if (!read_int(c, v) || v < 0) return fail(DCP_EFDATA);
        header_.protein_sizes[(size_t)i] = (uint32_t)v;
      }
    }
    else if (t.kind == K_EXT && t.ext_type == 6 && t.len % 4 == 0) // legacy: big-endian u32
    {
      header_.protein_sizes.resize(t.len / 4);
      for (size_t i = 0; i < header_.protein_sizes.size(); ++i)
      {
        uint8_t const *b = t.data + 4 * i;
        header_.protein_sizes[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
      }
    }
    else
      return fail(DCP_EFDATA);
  }
  // c-core/protein_reader.c:42-52
  if (!expect_key(c, "proteins")) return fail(DCP_EFDATA);
  {
    Tok t = next(c);
    if (t.kind != K_ARRAY) return fail(DCP_EFDATA);
    if (t.i > INT32_MAX) return fail(DCP_ETOOMANYPROTEINS);
    if ((size_t)t.i != header_.protein_sizes.size()) return fail(DCP_EINVALNUMPROTEINS);
  }
  if (!c.ok) return fail(DCP_EFDATA);
  offsets_.resize(header_.protein_sizes.size() + 1);
  offsets_[0] = (int64_t)(c.p - data_);
  for (size_t i = 0; i < header_.protein_sizes.size(); ++i)
    offsets_[i + 1] = offsets_[i] + (int64_t)header_.protein_sizes[i];
  if ((size_t)offsets_.back() > size_) return fail(DCP_EFDATA);
  return 0;
}

// c-core/protein.c:283-351
int DcpDbReader::read_protein(int i, DcpProtein &x) const
{
  if (!data_ || i < 0 || i >= num_proteins()) return DCP_EINVALPART;
  Cur c{data_ + offsets_[(size_t)i], data_ + offsets_[(size_t)i + 1]};
  int64_t v = 0;
  if (!expect_map(c, 10)) return DCP_EFDATA;
  if (!expect_key(c, "accession") || !read_str(c, x.accession, 32)) return DCP_EFDATA;
  if (!expect_key(c, "gencode") || !read_int(c, v)) return DCP_EFDATA;
  x.gencode = (int)v;
  if (!expect_key(c, "consensus") || !read_str(c, x.consensus, DCP_MODEL_MAX + 1)) return DCP_EFDATA;
  if (!expect_key(c, "core_size") || !read_int(c, v)) return DCP_EFDATA;
  if (v <= 0 || v > DCP_MODEL_MAX) return DCP_ELARGECORESIZE;
  x.core_size = (int)v;
  size_t const K = (size_t)x.core_size;

  x.null_emission.resize(DCP_TABLE_SIZE);
  x.bg_emission.resize(DCP_TABLE_SIZE);
  if (!expect_key(c, "null_nuclt_dist") || !skip(c)) return DCP_ENUCLTDUNPACK;
  if (!expect_key(c, "null_emission") || !read_f32array(c, DCP_TABLE_SIZE, x.null_emission.data()))
    return DCP_EFDATA;
  if (!expect_key(c, "bg_nuclt_dist") || !skip(c)) return DCP_ENUCLTDUNPACK;
  if (!expect_key(c, "bg_emission") || !read_f32array(c, DCP_TABLE_SIZE, x.bg_emission.data()))
    return DCP_EFDATA;

  x.trans.resize((K + 1) * 7);
  x.emission.resize((K + 1) * DCP_TABLE_SIZE);
  if (!expect_key(c, "nodes") || !expect_map(c, (int64_t)(K + 1) * 3)) return DCP_EFDATA;
  for (size_t n = 0; n <= K; ++n)
  {
    if (!expect_key(c, "nuclt_dist") || !skip(c)) return DCP_ENUCLTDUNPACK;
    if (!expect_key(c, "trans") || !read_f32array(c, 7, x.trans.data() + 7 * n)) return DCP_EFDATA;
    if (!expect_key(c, "emission") || !read_f32array(c, DCP_TABLE_SIZE, x.emission.data() + DCP_TABLE_SIZE * n))
      return DCP_EFDATA;
  }
  x.BMk.resize(K);
  if (!expect_key(c, "BMk") || !read_f32array(c, K, x.BMk.data())) return DCP_EFDATA;
  if (!c.ok || c.p != c.end) return DCP_EFDATA;
  return 0;
}

// nuclt_dist_unpack (c-core/nuclt_dist.c:22-31): array(2){ nuclt lprobs (4 f32), codon marginals (125 f32) }
static bool read_nuclt_dist(Cur &c, float *nucltp, float *codonm)
{
  Tok t = next(c);
  if (t.kind != K_ARRAY || t.i != 2) return false;
  return read_f32array(c, 4, nucltp) && read_f32array(c, 125, codonm);
}

void DcpDecoder::prepare()
{
  size_t const n = nucltp.size() / 4;
  base.resize(n * 4);
  prior.resize(n * 64);
  static_assert(sizeof(std::atomic<uint8_t>) == 1, "the memo is a byte array");
  memo.reset(new std::atomic<uint8_t>[n * DCP_TABLE_SIZE]);
  memset((void *)memo.get(), 0xFF, n * DCP_TABLE_SIZE); // nobody else sees the decoder yet
  for (size_t i = 0; i < n * 4; ++i) base[i] = exp((double)nucltp[i]);
  for (size_t e = 0; e < n; ++e)
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        for (int c = 0; c < 4; ++c) prior[e * 64 + (size_t)(a * 16 + b * 4 + c)] = exp((double)codonm[e * 125 + (size_t)(a * 25 + b * 5 + c)]);
}

int DcpDbReader::read_decoder(int i, DcpDecoder &x) const
{
  if (!data_ || i < 0 || i >= num_proteins()) return DCP_EINVALPART;
  Cur c{data_ + offsets_[(size_t)i], data_ + offsets_[(size_t)i + 1]};
  int64_t v = 0;
  std::string text;
  if (!expect_map(c, 10)) return DCP_EFDATA;
  if (!expect_key(c, "accession") || !read_str(c, text, 32)) return DCP_EFDATA;
  if (!expect_key(c, "gencode") || !read_int(c, v)) return DCP_EFDATA;
  x.gencode = (int)v;
  if (!expect_key(c, "consensus") || !read_str(c, text, DCP_MODEL_MAX + 1)) return DCP_EFDATA;
  if (!expect_key(c, "core_size") || !read_int(c, v)) return DCP_EFDATA;
  if (v <= 0 || v > DCP_MODEL_MAX) return DCP_ELARGECORESIZE;
  x.core_size = (int)v;
  x.epsilon = header_.epsilon;
  size_t const K = (size_t)x.core_size;
  x.nucltp.assign((K + 3) * 4, 0.0f);
  x.codonm.assign((K + 3) * 125, 0.0f);
  if (!expect_key(c, "null_nuclt_dist") || !read_nuclt_dist(c, x.nucltp.data(), x.codonm.data())) return DCP_ENUCLTDUNPACK;
  if (!expect_key(c, "null_emission") || !skip(c)) return DCP_EFDATA;
  if (!expect_key(c, "bg_nuclt_dist") || !read_nuclt_dist(c, x.nucltp.data() + 4, x.codonm.data() + 125))
    return DCP_ENUCLTDUNPACK;
  if (!expect_key(c, "bg_emission") || !skip(c)) return DCP_EFDATA;
  if (!expect_key(c, "nodes") || !expect_map(c, (int64_t)(K + 1) * 3)) return DCP_EFDATA;
  for (size_t n = 0; n <= K; ++n)
  {
    if (!expect_key(c, "nuclt_dist") || !read_nuclt_dist(c, x.nucltp.data() + 4 * (2 + n), x.codonm.data() + 125 * (2 + n)))
      return DCP_ENUCLTDUNPACK;
    if (!expect_key(c, "trans") || !skip(c)) return DCP_EFDATA;
    if (!expect_key(c, "emission") || !skip(c)) return DCP_EFDATA;
  }
  if (!c.ok) return DCP_EFDATA;
  x.prepare();
  return 0;
}

int DcpDbReader::read_protein_head(int i, int &core_size, std::string &accession) const
{
  if (!data_ || i < 0 || i >= num_proteins()) return DCP_EINVALPART;
  Cur c{data_ + offsets_[(size_t)i], data_ + offsets_[(size_t)i + 1]};
  int64_t v = 0;
  std::string consensus;
  if (!expect_map(c, 10)) return DCP_EFDATA;
  if (!expect_key(c, "accession") || !read_str(c, accession, 32)) return DCP_EFDATA;
  if (!expect_key(c, "gencode") || !read_int(c, v)) return DCP_EFDATA;
  if (!expect_key(c, "consensus") || !read_str(c, consensus, DCP_MODEL_MAX + 1)) return DCP_EFDATA;
  if (!expect_key(c, "core_size") || !read_int(c, v)) return DCP_EFDATA;
  if (v <= 0 || v > DCP_MODEL_MAX) return DCP_ELARGECORESIZE;
  core_size = (int)v;
  return 0;
}

long dcp_partition_size(long nelems, long nparts, long idx)
{
  long x = nelems - idx;
  if (x < 0) x = 0;
  return (x + nparts - 1) / nparts;
}
