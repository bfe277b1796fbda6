/* scan_client.c -- a plain C client of the reference's public API, shaped after
 * c-core/test_scan.c:20-60: new batch, add sequences, new scan, setup, run, progress,
 * delete.  Compiled with gcc against include/deciphon.h and linked to
 * libdeciphon_hip.so by tests/test_gpu_c_client.py; it proves that the shared library is
 * usable from C exactly as libdeciphon is.
 *
 *   scan_client <db.dcp> <reads.fna> <product_dir> [multi_hits hmmer3_compat]
 * prints "rows=<n> progress=<p>" and exits 0, or prints the error string and exits 1.
 */
#include "deciphon.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int windows_seen = 0;
static void on_window(void *userdata)
{
  (void)userdata;
  windows_seen++;
}

int main(int argc, char **argv)
{
  if (argc < 4) return 2;
  bool multi_hits = argc > 4 ? atoi(argv[4]) != 0 : true;
  bool hmmer3_compat = argc > 5 ? atoi(argv[5]) != 0 : false;

  struct dcp_batch *batch = dcp_batch_new();
  if (!batch) return 1;
  FILE *fp = fopen(argv[2], "r");
  if (!fp) return 1;
  static char line[1 << 16], name[256], seq[1 << 20];
  long id = 0;
  int rc = 0;
  name[0] = seq[0] = 0;
  while (fgets(line, sizeof line, fp))
  {
    line[strcspn(line, "\r\n")] = 0;
    if (line[0] == '>')
    {
      if (seq[0] && (rc = dcp_batch_add(batch, id++, name, seq))) break;
      snprintf(name, sizeof name, "%s", line + 1);
      seq[0] = 0;
    }
    else
      strncat(seq, line, sizeof seq - strlen(seq) - 1);
  }
  fclose(fp);
  if (!rc && seq[0]) rc = dcp_batch_add(batch, id++, name, seq);
  if (rc)
  {
    fprintf(stderr, "dcp_batch_add: %s\n", dcp_error_string(rc));
    return 1;
  }

  struct dcp_scan *scan = dcp_scan_new();
  if (!scan) return 1;
  /* the error codes are part of the public header (c-core/deciphon.h:34-116) */
  if (dcp_scan_setup(scan, "/nonexistent/none.dcp", 51300, 1, true, false, false, NULL, NULL) != DCP_EOPENDB)
  {
    fprintf(stderr, "a missing database must fail with DCP_EOPENDB\n");
    return 1;
  }
  if ((rc = dcp_scan_setup(scan, argv[1], 51300, 1, multi_hits, hmmer3_compat, false, on_window, NULL)))
  {
    fprintf(stderr, "dcp_scan_setup: %s\n", dcp_error_string(rc));
    return 1;
  }
  if ((rc = dcp_scan_run(scan, batch, argv[3])))
  {
    fprintf(stderr, "dcp_scan_run: %s\n", dcp_error_string(rc));
    return 1;
  }
  /* a scan object is reusable (c-core/test_scan.c:62-80 "reuse") */
  if ((rc = dcp_scan_run(scan, batch, argv[3])))
  {
    fprintf(stderr, "dcp_scan_run (reuse): %s\n", dcp_error_string(rc));
    return 1;
  }
  printf("rows=%ld progress=%d windows=%d\n", dcp_scan_num_products(scan), dcp_scan_progress(scan), windows_seen);
  dcp_scan_del(scan);
  dcp_batch_del(batch);
  return 0;
}
