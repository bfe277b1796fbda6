// dcp_db.h -- reader of pressed Deciphon databases (.dcp).
//
// Replaces, for the scan path, c-core/database_reader.c:26-80 (header),
// c-core/protein_reader.c:29-128 (partition offsets from protein_sizes) and
// c-core/protein.c:283-351 (protein_unpack).  The reference reads through the
// third-party lio/lite_pack MessagePack library; this is an own reader of the
// MessagePack subset the writer emits (c-core/database_writer.c:95-193,
// c-core/protein.c:234-281), accepting both encodings of numeric arrays that
// exist in the wild: `bin` + native-endian floats (current
// writer, c-core/write.c:59-66) and the legacy big-endian `ext` types 8 / 6 of
// the committed fixture control/tests/files/minifam.dcp.
#pragma once
#include <atomic>
#include <memory>
#include <stdint.h>
#include <string>
#include <vector>

struct DcpProtein
{
  std::string accession;
  std::string consensus;
  int gencode = 0;
  int core_size = 0;
  std::vector<float> null_emission; // [1364] log-probs
  std::vector<float> bg_emission;   // [1364]
  std::vector<float> trans;         // [(K+1)*7]  MM MI MD IM II DM DD
  std::vector<float> emission;      // [(K+1)*1364] node-major
  std::vector<float> BMk;           // [K]
};

// What decoder_setup (c-core/decoder.c:21-36) takes from a protein: the nucleotide and codon distributions of
// the null model, the background and every node.  imm_nuclt_lprob is 4 log-probabilities (A, C, G, T);
// imm_codon_marg is the 5 x 5 x 5 table of codon marginals, index 4 = "any nucleotide" (third-party imm).
struct DcpDecoder
{
  int gencode = 0;
  int core_size = 0;
  float epsilon = 0;
  // [K + 3]: 0 = null, 1 = background, 2 + n = node n (n = 0..K)
  std::vector<float> nucltp; // [(K + 3) * 4]
  std::vector<float> codonm; // [(K + 3) * 125]
  // probabilities of the above, made once when the distributions are read (decoding runs once per path step):
  std::vector<double> base;  // [(K + 3) * 4]  exp(nucltp)
  std::vector<double> prior; // [(K + 3) * 64] exp(codonm) of the 64 codons, index a * 16 + b * 4 + c
  // decoded codon (a * 16 + b * 4 + c) by [entry][quasi-codon code 0..1363], filled as steps ask for them:
  // 0xFF = not decoded yet, 0xFE = no codon has positive probability.  A scan decodes the same few codes of
  // the same nodes over and over.
  std::unique_ptr<std::atomic<uint8_t>[]> memo;
  void prepare();
};

struct DcpDbHeader
{
  int magic_number = 0;
  int version = 0;
  int entry_dist = 0;
  float epsilon = 0;
  bool has_ga = false;
  std::string abc_symbols;
  int abc_typeid = 0;
  std::string amino_symbols;
  std::vector<uint32_t> protein_sizes;
};

class DcpDbReader
{
public:
  DcpDbReader() = default;
  ~DcpDbReader();
  DcpDbReader(DcpDbReader const &) = delete;
  DcpDbReader &operator=(DcpDbReader const &) = delete;

  // all return 0 or a DCP_E* code
  int open(char const *path);
  void close();
  DcpDbHeader const &header() const { return header_; }
  int num_proteins() const { return (int)header_.protein_sizes.size(); }
  // byte offset of protein i inside the file (protein_reader's partition offsets)
  int64_t protein_offset(int i) const { return offsets_[(size_t)i]; }
  int read_protein(int i, DcpProtein &out) const;
  // only accession and core size (the first keys of the record): cheap pre-scan
  int read_protein_head(int i, int &core_size, std::string &accession) const;
  // the distributions codon decoding needs (the emission and transition arrays are stepped over)
  int read_decoder(int i, DcpDecoder &out) const;

private:
  uint8_t const *data_ = nullptr;
  size_t size_ = 0;
  int fd_ = -1;
  DcpDbHeader header_;
  std::vector<int64_t> offsets_;
};

// c-core/partition_size.c:13-16
extern "C" long dcp_partition_size(long nelems, long nparts, long idx);
