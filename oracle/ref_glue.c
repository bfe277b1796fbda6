/* ref_glue.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Flat-array driver around the REFERENCE's own DP engine.  oracle/Makefile
 * compiles c-core/viterbi.c (+ error.c, loglevel.c) unmodified, straight from
 * /root/reference, and links them with this file into oracle/_ref/libdcp_ref.so.
 * The same library carries, equally unmodified, the other files of the path
 * that need only libc and the reference's own headers -- partition_size.c,
 * state.c (+ bug.c), disambiguate.c, uppercase.c -- whose functions the tests
 * call directly (tests/test_reference_pins.py).
 * Nothing of the reference is copied into this repository.
 *
 * c-core/trellis.c is NOT compiled: it includes the third-party imm_path.h,
 * which this image lacks, so it is unbuildable here.  viterbi.c only needs the
 * five trellis *storage* functions (init/setup/cleanup/seek_xnode/seek_node,
 * declared in the reference's own trellis.h); they are restated below.  All
 * back-pointer arithmetic (viterbi.c cost()/after() and trellis.h
 * trellis_set()) is the reference's own code.  trellis_unzip() is not part of
 * this build; the traceback restatement lives in dcp_oracle.c and is pinned by
 * the reference's committed products.tsv paths.
 */
#include "trellis.h"
#include "viterbi.h"

#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- trellis storage, after c-core/trellis.c:14-49,115-123 ----------------- */
void trellis_init(struct trellis *x)
{
  x->core_size = 0;
  x->xnodes = NULL;
  x->nodes = NULL;
  x->xnode = NULL;
  x->node = NULL;
}

int trellis_setup(struct trellis *x, int core_size, int seq_size)
{
  size_t stages = (size_t)seq_size + 1;
  x->core_size = core_size;
  x->xnodes = realloc(x->xnodes, sizeof(*x->xnodes) * stages);
  x->nodes = realloc(x->nodes, sizeof(*x->nodes) * stages * (size_t)core_size);
  if (!x->xnodes || !x->nodes)
  {
    free(x->xnodes);
    free(x->nodes);
    x->xnodes = NULL;
    x->nodes = NULL;
    return 20; /* DCP_ENOMEM */
  }
  return 0;
}

void trellis_cleanup(struct trellis *x)
{
  free(x->xnodes);
  free(x->nodes);
  trellis_init(x);
}

void trellis_seek_xnode(struct trellis *x, int stage) { x->xnode = x->xnodes + stage; }

void trellis_seek_node(struct trellis *x, int stage, int core_idx)
{
  x->node = x->nodes + (size_t)stage * x->core_size + core_idx;
}

/* ---- flat driver ------------------------------------------------------------ */
struct ref
{
  struct viterbi *v;
  int K;
  /* borrowed parameter arrays of the last ref_setup(), for ref_fresh() */
  float const *trans, *match, *null_cost, *bg_cost;
  float xt[13];
};

struct codes
{
  uint16_t const *m; /* [pos][5], like imm_eseq's code matrix */
};

static int code_fn(int pos, int len, void *arg)
{
  struct codes const *c = arg;
  return c->m[(size_t)pos * 5 + (len - 1)];
}

static uint16_t *make_codes(uint8_t const *seq, int L)
{
  static int const off[6] = {0, 0, 4, 20, 84, 340};
  uint16_t *m = malloc(sizeof(uint16_t) * 5 * ((size_t)L + 1));
  for (int pos = 0; pos < L; ++pos)
    for (int len = 1; len <= 5; ++len)
    {
      int idx = 0;
      if (pos + len <= L)
        for (int i = 0; i < len; ++i) idx = idx * 4 + seq[pos + i];
      m[(size_t)pos * 5 + len - 1] = (uint16_t)(off[len] + idx);
    }
  return m;
}

void *ref_new(void)
{
  struct ref *r = malloc(sizeof(*r));
  r->v = viterbi_new();
  r->K = 0;
  r->trans = r->match = r->null_cost = r->bg_cost = NULL;
  return r;
}

void ref_del(void *p)
{
  struct ref *r = p;
  if (!r) return;
  viterbi_del(r->v);
  free(r);
}

/* trans[8][K] in enum core_trans_id order; costs, exactly what
 * protein_setup_viterbi would pass to the setters */
int ref_setup(void *p, int K, float const *trans, float const *match, float const *null_cost,
              float const *bg_cost)
{
  struct ref *r = p;
  int rc = viterbi_setup(r->v, K);
  if (rc) return rc;
  r->K = K;
  r->trans = trans;
  r->match = match;
  r->null_cost = null_cost;
  r->bg_cost = bg_cost;
  for (int id = 0; id < 8; ++id)
    for (int k = 0; k < K; ++k)
      viterbi_set_core_trans(r->v, (enum core_trans_id)id, trans[id * K + k], k);
  int T = viterbi_table_size();
  for (int c = 0; c < T; ++c)
  {
    viterbi_set_null(r->v, null_cost[c], c);
    viterbi_set_background(r->v, bg_cost[c], c);
    for (int k = 0; k < K; ++k) viterbi_set_match(r->v, match[(size_t)c * K + k], k, c);
  }
  return 0;
}

void ref_set_xtrans(void *p, float const xt[13])
{
  struct ref *r = p;
  memcpy(r->xt, xt, sizeof(r->xt));
  for (int id = 0; id < 13; ++id) viterbi_set_extr_trans(r->v, (enum extr_trans_id)id, xt[id]);
}

/* The reference never clears "row 0" of its DP ring between runs: cost()
 * (viterbi.c:471-473) only sets S and B, so N,J,E,C,T and M,D,I of row 0 still
 * hold the LAST row of whatever ran before on the same struct viterbi.  A run is
 * history-free only right after viterbi_setup() (viterbi.c:336-381 fills
 * everything with +inf).  ref_fresh() re-does setup + setters so the next run
 * starts from that defined state; the arrays given to ref_setup() must still
 * be alive. */
int ref_fresh(void *p)
{
  struct ref *r = p;
  float xt[13];
  memcpy(xt, r->xt, sizeof(xt));
  int rc = ref_setup(p, r->K, r->trans, r->match, r->null_cost, r->bg_cost);
  if (rc) return rc;
  ref_set_xtrans(p, xt);
  return 0;
}

float ref_null(void *p, uint8_t const *seq, int L)
{
  struct ref *r = p;
  uint16_t *m = make_codes(seq, L);
  struct codes c = {m};
  float x = viterbi_null(r->v, L, code_fn, &c);
  free(m);
  return x;
}

float ref_cost(void *p, uint8_t const *seq, int L)
{
  struct ref *r = p;
  uint16_t *m = make_codes(seq, L);
  struct codes c = {m};
  float x = viterbi_cost(r->v, L, code_fn, &c);
  free(m);
  return x;
}

int ref_path(void *p, uint8_t const *seq, int L, uint32_t *xnodes, uint16_t *nodes)
{
  struct ref *r = p;
  uint16_t *m = make_codes(seq, L);
  struct codes c = {m};
  int rc = viterbi_path(r->v, L, code_fn, &c);
  free(m);
  if (rc) return rc;
  struct trellis *t = viterbi_trellis(r->v);
  memcpy(xnodes, t->xnodes, sizeof(uint32_t) * ((size_t)L + 1));
  memcpy(nodes, t->nodes, sizeof(uint16_t) * ((size_t)L + 1) * (size_t)r->K);
  return 0;
}

/* CPU baseline: the reference's per-window work (viterbi_null + viterbi_cost,
 * c-core/thread.c:114-117) for `nprob` windows of one profile, spread over
 * `nthreads` OpenMP threads with one struct viterbi each, like the per-thread
 * works of c-core/scan.c:188-208.  Returns seconds; scores go to out[2*nprob]. */
double ref_bench(int K, float const *trans, float const *match, float const *null_cost,
                 float const *bg_cost, float const *xt /*[nprob][13]*/, uint8_t const *seqs,
                 int64_t const *offsets /*[nprob+1]*/, int nprob, int nthreads, int repeat, float *out)
{
  void **refs = malloc(sizeof(void *) * (size_t)nthreads);
  uint16_t **codes = malloc(sizeof(uint16_t *) * (size_t)nprob);
  for (int i = 0; i < nthreads; ++i)
  {
    refs[i] = ref_new();
    ref_setup(refs[i], K, trans, match, null_cost, bg_cost);
  }
  for (int i = 0; i < nprob; ++i)
    codes[i] = make_codes(seqs + offsets[i], (int)(offsets[i + 1] - offsets[i]));

  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int j = 0; j < nprob * repeat; ++j)
  {
    int i = j % nprob;
    struct ref *r = refs[omp_get_thread_num()];
    int L = (int)(offsets[i + 1] - offsets[i]);
    struct codes c = {codes[i]};
    ref_set_xtrans(r, xt + 13 * (size_t)i);
    out[2 * i + 0] = viterbi_null(r->v, L, code_fn, &c);
    out[2 * i + 1] = viterbi_cost(r->v, L, code_fn, &c);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);

  for (int i = 0; i < nprob; ++i) free(codes[i]);
  for (int i = 0; i < nthreads; ++i) ref_del(refs[i]);
  free(codes);
  free(refs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
