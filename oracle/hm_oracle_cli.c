/* TEST INFRASTRUCTURE ONLY: command line front-end of the CPU restatement.
 *   hm_oracle_cli <in.yuv> <w> <h> <bitdepth> <frames> <qp> <wpp> <dump.bin> [trace.txt] [max_ctus]
 * Writes the same "HMD1" dump format as oracle/_ref/hm_dump (see oracle/ref_harness.cpp). */
#include "hm_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char **argv)
{
  if (argc < 9) { fprintf(stderr, "usage: %s in.yuv w h bd frames qp wpp dump.bin [trace] [max_ctus]\n", argv[0]); return 2; }
  const int w = atoi(argv[2]), h = atoi(argv[3]), bd = atoi(argv[4]), frames = atoi(argv[5]), qp = atoi(argv[6]), wpp = atoi(argv[7]);
  if (argc > 9 && argv[9][0] && strcmp(argv[9], "-")) hmo_set_trace(argv[9]);
  const int maxCtus = argc > 10 ? atoi(argv[10]) : 0;
  FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[8], "wb");
  if (!fi || !fo) { perror("open"); return 1; }
  hmo_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.width = w; cfg.height = h; cfg.bit_depth = bd; cfg.wpp = wpp; hmo_cfg_set_qp(&cfg, qp);
  const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
  const size_t ny = (size_t)w * h, nc = ny / 4;
  uint16_t *org[3] = { malloc(ny * 2), malloc(nc * 2), malloc(nc * 2) }, *rec[3] = { calloc(ny, 2), calloc(nc, 2), calloc(nc, 2) };
  hmo_ctu *ctus = calloc(nctu, sizeof(hmo_ctu));
  fwrite("HMD1", 1, 4, fo);
  uint32_t hdr[5] = { (uint32_t)w, (uint32_t)h, (uint32_t)bd, 64, (uint32_t)frames }; fwrite(hdr, 4, 5, fo);
  double total = 0;
  for (int f = 0; f < frames; f++) {
    for (int c = 0; c < 3; c++) {
      const size_t n = c ? nc : ny;
      if (bd == 8) { unsigned char *t = malloc(n); if (fread(t, 1, n, fi) != n) return 3; for (size_t i = 0; i < n; i++) org[c][i] = t[i]; free(t); }
      else if (fread(org[c], 2, n, fi) != n) return 3;
    }
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    const uint16_t *o[3] = { org[0], org[1], org[2] };
    int rc = maxCtus ? hmo_compress_rows(&cfg, o, rec, ctus, maxCtus) : hmo_compress_slice(&cfg, o, rec, ctus);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) { fprintf(stderr, "oracle failed %d\n", rc); return 4; }
    total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    uint32_t u[2] = { (uint32_t)f, (uint32_t)nctu }; fwrite(u, 4, 2, fo);
    for (int a = 0; a < nctu; a++) {
      const hmo_ctu *c = ctus + a;
      fwrite(&c->total_cost, 8, 1, fo); fwrite(&c->total_bits, 4, 1, fo); fwrite(&c->total_dist, 4, 1, fo);
      fwrite(c->depth, 1, 256, fo); fwrite(c->part_size, 1, 256, fo); fwrite(c->pred_mode, 1, 256, fo);
      fwrite(c->intra_dir_luma, 1, 256, fo); fwrite(c->intra_dir_chroma, 1, 256, fo); fwrite(c->tr_idx, 1, 256, fo);
      fwrite(c->cbf, 1, 768, fo); fwrite(c->tskip, 1, 768, fo);
      fwrite(c->coeff_y, 4, 4096, fo); fwrite(c->coeff_cb, 4, 1024, fo); fwrite(c->coeff_cr, 4, 1024, fo);
    }
    fwrite(rec[0], 2, ny, fo); fwrite(rec[1], 2, nc, fo); fwrite(rec[2], 2, nc, fo);
  }
  fprintf(stderr, "oracle: %d frame(s), %.3f s, %.1f CTU/s\n", frames, total, (maxCtus ? maxCtus : nctu) * frames / total);
  fclose(fo); fclose(fi);
  return 0;
}
