/* deciphon.h -- the reference's public C API (c-core/deciphon.h:9-32), as exported
 * by libdeciphon_hip.so, so that existing callers (python-core's CFFI binding,
 * python-core/deciphon_core/interface.h:1-40; c-core/test_scan.c:36-41) link
 * against the MI355X path unchanged.
 *
 * What differs from the reference, by design of this build:
 *  - dcp_scan_run scores every (profile x window) on the GPU; `num_threads` is
 *    accepted and ignored (one engine drives one GPU), `cache` likewise (profiles
 *    are always resident in HBM).
 *  - HMMER rescoring (the h3daemon TCP client of c-core/hmmer.c) is out of scope:
 *    `port` is ignored, no row is dropped for lack of a HMMER hit, the `evalue`
 *    column holds `nan` and no hmmer/ directory with .h3r files is written.
 *  - the codon and amino fields of the `match` column come from a restatement of third-party
 *    imm's imm_frame_cond_decode (csrc/host_logic.h), pinned by the reference's committed
 *    products.tsv only.
 *  - dcp_press_* needs the absent third-party imm/hmmer_reader libraries; the
 *    symbols exist and fail with DCP_EFUNCUSE.
 */
#ifndef DECIPHON_AMD_DECIPHON_H
#define DECIPHON_AMD_DECIPHON_H

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

struct dcp_scan;
struct dcp_batch;
struct dcp_press;

/* scan: c-core/scan.c:40-227 */
struct dcp_scan *dcp_scan_new(void);
void dcp_scan_del(struct dcp_scan const *);
int dcp_scan_setup(struct dcp_scan *, char const *dbfile, int port, int num_threads, bool multi_hits,
                   bool hmmer3_compat, bool cache, void (*callback)(void *), void *userdata);
int dcp_scan_run(struct dcp_scan *, struct dcp_batch *, char const *product_dir);
void dcp_scan_interrupt(struct dcp_scan *);
int dcp_scan_progress(struct dcp_scan const *);

/* batch: c-core/batch.c:15-58 */
struct dcp_batch *dcp_batch_new(void);
void dcp_batch_del(struct dcp_batch *);
int dcp_batch_add(struct dcp_batch *, long id, char const *name, char const *data);
void dcp_batch_reset(struct dcp_batch *);

/* press: c-core/press.c:43-204 -- not provided by this build, see above */
struct dcp_press *dcp_press_new(void);
int dcp_press_setup(struct dcp_press *, int gencode_id, float epsilon);
int dcp_press_open(struct dcp_press *, char const *hmm, char const *db);
long dcp_press_nproteins(struct dcp_press const *);
int dcp_press_next(struct dcp_press *);
bool dcp_press_end(struct dcp_press const *);
int dcp_press_close(struct dcp_press *);
void dcp_press_del(struct dcp_press const *);

char const *dcp_error_string(int error_code);

/* The return codes, value for value c-core/deciphon.h:34-116: they are part of the ABI Python sees
 * through CFFI.  GPU failures are mapped onto existing codes (DCP_ENOMEM, DCP_EFUNCUSE). */
enum
{
  DCP_EDIFFABC = 1,
  DCP_EFCLOSE = 2,
  DCP_EFDATA = 3,
  DCP_EREFOPEN = 4,
  DCP_EFREAD = 5,
  DCP_EFSEEK = 6,
  DCP_EFTELL = 7,
  DCP_EFUNCUSE = 8,
  DCP_EFWRITE = 9,
  DCP_EGETPATH = 10,
  DCP_EZEROSEQ = 11,
  DCP_EZEROMODEL = 12,
  DCP_EZEROPART = 13,
  DCP_EDECODON = 14,
  DCP_ELARGEMODEL = 15,
  DCP_ELARGEPROTEIN = 16,
  DCP_EREADHMMER3 = 17,
  DCP_EMANYPARTS = 18,
  DCP_EMANYTRANS = 19,
  DCP_ENOMEM = 20,
  DCP_EOPENDB = 21,
  DCP_EOPENHMM = 22,
  DCP_EOPENTMP = 23,
  DCP_ETRUNCPATH = 24,
  DCP_EDPUNPACK = 25,
  DCP_EDPPACK = 26,
  DCP_ENUCLTDUNPACK = 27,
  DCP_ENUCLTDPACK = 28,
  DCP_ESETTRANS = 29,
  DCP_EADDSTATE = 30,
  DCP_EDPRESET = 31,
  DCP_EFSTAT = 32,
  DCP_EFOPEN = 33,
  DCP_ELARGEFILE = 34,
  DCP_ELONGPATH = 35,
  DCP_EIMMRESETTASK = 36,
  DCP_EIMMNEWTASK = 37,
  DCP_EIMMSETUPTASK = 38,
  DCP_EWRITEPROD = 39,
  DCP_EINVALPART = 40,
  DCP_ELONGACCESSION = 41,
  DCP_EMANYTHREADS = 42,
  DCP_ETMPFILE = 43,
  DCP_EFFLUSH = 44,
  DCP_EMKDIR = 45,
  DCP_EFORMAT = 46,
  DCP_ERMDIR = 47,
  DCP_ERMFILE = 48,
  DCP_ESETGENCODE = 49,
  DCP_EGENCODEID = 50,
  DCP_EH3CDIAL = 51,
  DCP_EH3CPUT = 52,
  DCP_EH3CPOP = 53,
  DCP_EH3CPACK = 54,
  DCP_EH3CMAXRETRY = 55,
  DCP_EH3CWARMUP = 56,
  DCP_ESEQABC = 57,
  DCP_EFDOPEN = 58,
  DCP_EMKSTEMP = 59,
  DCP_ELONGABC = 60,
  DCP_ELONGCONSENSUS = 61,
  DCP_ENOTDIALED = 62,
  DCP_ELARGECORESIZE = 63,
  DCP_EINVALSTATE = 64,
  DCP_EINVALSIZE = 65,
  DCP_EENDOFFILE = 66,
  DCP_EENDOFNODES = 67,
  DCP_EDBVERSION = 68,
  DCP_ENOTDBFILE = 69,
  DCP_EINVALSTATEID = 70,
  DCP_ENUCLTNOSUPPORT = 71,
  DCP_EDBDNASEQRNA = 72,
  DCP_EDBRNASEQDNA = 73,
  DCP_ENUCLTSEQTU = 74,
  DCP_ENOHIT = 75,
  DCP_EOPEN = 76,
  DCP_ECLOSE = 77,
  DCP_EDUP = 78,
  DCP_ETOOMANYPROTEINS = 79,
  DCP_EINVALNUMPROTEINS = 80,
};

/* Not in the reference: a scan that owns only partition `index` of `nparts`
 * contiguous profile partitions (partition_size, c-core/partition_size.c:13-16) on
 * HIP device `device` -- what one rank of a multi-GPU job calls instead of
 * dcp_scan_setup.  Its products.tsv rows are that partition's, in database order. */
int dcp_scan_setup_partition(struct dcp_scan *, char const *dbfile, int device, int index, int nparts,
                             bool multi_hits, bool hmmer3_compat, void (*callback)(void *), void *userdata);
/* The same with partition boundaries that balance the sum of core sizes instead of the number of profiles
 * (DP cells go with K): still contiguous and in database order, so the ranks' rows still
 * concatenate to the whole scan's.  dcp_scan_partition_range reports what a scan owns. */
int dcp_scan_setup_partition_balanced(struct dcp_scan *, char const *dbfile, int device, int index, int nparts,
                                      bool multi_hits, bool hmmer3_compat, void (*callback)(void *), void *userdata);
int dcp_scan_partition_range(struct dcp_scan const *, int *first, int *count);
/* Number of product rows the last dcp_scan_run wrote, and row i (without newline). */
long dcp_scan_num_products(struct dcp_scan const *);
char const *dcp_scan_product(struct dcp_scan const *, long i);
/* Where the wall time of the last dcp_scan_run went (measurement only; SURVEY 8d's wall definition: first H2D of the
 * reads to the last product row on the host).  Fills out[0..n) with, in order: total seconds, reads H2D + encode,
 * window bookkeeping, cost pass + LRT filter, path pass + unzip (the part the cost pass did not cover), row
 * formatting + decoding, products.tsv, then the counts of rounds, windows scored and path passes.  Returns how many
 * values exist (DCP_SCAN_TIMING_VALUES). */
#define DCP_SCAN_TIMING_VALUES 10
int dcp_scan_last_timing(struct dcp_scan const *, double *out, int n);

#ifdef __cplusplus
}
#endif

#endif
