// host_sanitize.cpp -- TEST INFRASTRUCTURE: the host-side .dcp reader, windows and partitions under ASan + UBSan,
// on the golden database, on truncated copies and on 200 randomly corrupted copies (must fail cleanly, never fault).
#include "deciphon_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv)
{
  struct dcp_db *db = NULL;
  int rc = dcp_db_open(argv[1], &db);
  if (rc) { printf("open rc %d\n", rc); return 1; }
  int n = dcp_db_num_proteins(db);
  printf("proteins %d\n", n);
  for (int i = 0; i < n; ++i)
  {
    int K = 0;
    rc = dcp_db_protein_core_size(db, i, &K);
    char acc[32];
    float *tr = (float *)malloc((K + 1) * 7 * 4), *em = (float *)malloc((size_t)(K + 1) * 1364 * 4), *bm = (float *)malloc(K * 4);
    float nul[1364], bg[1364];
    char *cons = (char *)malloc(K + 1);
    rc = dcp_db_read_protein(db, i, tr, em, bm, nul, bg, acc, cons);
    printf("  %d K=%d acc=%s rc=%d\n", i, K, acc, rc);
    free(tr); free(em); free(bm); free(cons);
  }
  struct dcp_window w;
  dcp_window_setup(&w, 10000, 173);
  while (dcp_window_next(&w)) printf("  window %d [%d,%d)\n", w.idx, w.start, w.stop);
  for (int N = 0; N < 30; ++N) for (int P = 1; P < 9; ++P) { long s = 0; for (int i = 0; i < P; ++i) s += dcp_partition_size(N, P, i); if (s != N) { printf("partition bug\n"); return 1; } }
  dcp_db_close(db);
  /* truncated and corrupt files must fail cleanly */
  FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  char *buf = (char *)malloc(sz); fread(buf, 1, sz, f); fclose(f);
  long cuts[] = {0, 1, 10, 100, 1000, sz / 2, sz - 1};
  for (unsigned c = 0; c < sizeof cuts / sizeof *cuts; ++c)
  {
    f = fopen(argv[2], "wb"); fwrite(buf, 1, cuts[c], f); fclose(f);
    rc = dcp_db_open(argv[2], &db);
    printf("  truncated at %ld: rc %d\n", cuts[c], rc);
    if (!rc) { int m = dcp_db_num_proteins(db); for (int i = 0; i < m; ++i) { int K; (void)dcp_db_protein_core_size(db, i, &K); (void)dcp_db_read_protein(db, i, 0, 0, 0, 0, 0, 0, 0); } dcp_db_close(db); }
  }
  srand(1);
  for (int it = 0; it < 200; ++it)
  {
    char *c2 = (char *)malloc(sz); memcpy(c2, buf, sz);
    for (int j = 0; j < 8; ++j) c2[rand() % (it < 100 ? 2000 : sz)] = (char)rand();
    f = fopen(argv[2], "wb"); fwrite(c2, 1, sz, f); fclose(f); free(c2);
    rc = dcp_db_open(argv[2], &db);
    if (!rc) { int m = dcp_db_num_proteins(db); for (int i = 0; i < m && i < 3; ++i) { int K; if (!dcp_db_protein_core_size(db, i, &K) && K > 0 && K < 100000) { float *em = (float *)malloc((size_t)(K + 1) * 1364 * 4); float *tr = (float *)malloc((size_t)(K + 1) * 7 * 4); float *bm = (float *)malloc((size_t)K * 4); char *cons = (char *)malloc(K + 1); char acc[32]; (void)dcp_db_read_protein(db, i, tr, em, bm, 0, 0, acc, cons); free(em); free(tr); free(bm); free(cons); } } dcp_db_close(db); }
  }
  {
    /* a value nested a few hundred thousand arrays deep behind the first "idx" key (skipped by the reader):
     * must fail with an error code, not overflow the stack */
    char const key[] = {(char)0xa3, 'i', 'd', 'x'};
    long at = -1;
    for (long i = 0; i + 4 < sz && i < 4096; ++i) if (!memcmp(buf + i, key, 4)) { at = i + 4; break; }
    if (at < 0) { printf("no idx key\n"); return 1; }
    long const deep = 400000;
    char *c2 = (char *)malloc(at + deep); memcpy(c2, buf, at); memset(c2 + at, 0x91, deep);
    f = fopen(argv[2], "wb"); fwrite(c2, 1, at + deep, f); fclose(f); free(c2);
    rc = dcp_db_open(argv[2], &db);
    printf("  nested arrays: rc %d\n", rc);
    if (!rc) { printf("nested arrays accepted\n"); return 1; }
  }
  printf("fuzz done\n");
  free(buf);
  return 0;
}
