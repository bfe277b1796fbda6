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
 *  - quasi-codon decoding (c-core/decoder.c, third-party imm) is out of scope: the
 *    codon and amino fields of the `match` column are left empty.
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

/* the DCP_E* return codes, c-core/deciphon.h:34-116 */
#include "deciphon_errors.h"

/* Not in the reference: a scan that owns only partition `index` of `nparts`
 * contiguous profile partitions (partition_size, c-core/partition_size.c:13-16) on
 * HIP device `device` -- what one rank of a multi-GPU job calls instead of
 * dcp_scan_setup.  Its products.tsv rows are that partition's, in database order. */
int dcp_scan_setup_partition(struct dcp_scan *, char const *dbfile, int device, int index, int nparts,
                             bool multi_hits, bool hmmer3_compat, void (*callback)(void *), void *userdata);
/* Number of product rows the last dcp_scan_run wrote, and row i (without newline). */
long dcp_scan_num_products(struct dcp_scan const *);
char const *dcp_scan_product(struct dcp_scan const *, long i);

#ifdef __cplusplus
}
#endif

#endif
