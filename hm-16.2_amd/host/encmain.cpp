// Small driver in the shape of the reference's TAppEncoder loop (TAppEncTop.cpp:407-520): reads a planar
// 4:2:0 file, feeds TEncTop::encode picture by picture and dumps what compressSlice left behind in the
// same "HMD1" format the reference harness writes (oracle/ref_harness.cpp), for the parity tests.
//   hm355_encmain <in.yuv> <w> <h> <bitdepth> <frames> <qp> <wpp> <dump.bin> [lf]     lf: run deblocking + SAO, the dump then holds the finished pictures
// The slice data of every picture (TEncSlice::encodeSlice) goes to <dump.bin>.bits: per picture u32 numSubstreams, then per substream u32 size + bytes.
#include "TEncTop.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

int main(int argc, char **argv)
{
  if (argc < 9) { fprintf(stderr, "usage: %s in.yuv w h bd frames qp wpp dump.bin\n", argv[0]); return 2; }
  const int w = atoi(argv[2]), h = atoi(argv[3]), bd = atoi(argv[4]), frames = atoi(argv[5]), qp = atoi(argv[6]), wpp = atoi(argv[7]);
  FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[8], "wb");
  if (!fi || !fo) { perror("open"); return 1; }
  TEncTop enc;
  enc.setSourceWidth(w); enc.setSourceHeight(h); enc.setInternalBitDepth(bd); enc.setQP(qp); enc.setIntraPeriod(1); enc.setGOPSize(1);
  enc.setWaveFrontSynchro(wpp); enc.setFramesToBeEncoded(frames);
  if (argc > 9 && !strcmp(argv[9], "lf")) { enc.setLoopFilterDisable(false); enc.setUseSAO(true); }
  enc.create(); enc.init();
  FILE *fb = fopen((std::string(argv[8]) + ".bits").c_str(), "wb");
  if (!fb) { perror("open"); return 1; }
  fwrite("HMD1", 1, 4, fo);
  uint32_t hdr[5] = { (uint32_t)w, (uint32_t)h, (uint32_t)bd, 64, (uint32_t)frames }; fwrite(hdr, 4, 5, fo);
  TComPicYuv org; org.create(w, h);
  for (int f = 0; f < frames; f++) {
    for (int c = 0; c < 3; c++) {
      const size_t n = (size_t)org.getWidth(ComponentID(c)) * org.getHeight(ComponentID(c)); uint16_t *p = org.getAddr(ComponentID(c));
      if (bd == 8) { for (size_t i = 0; i < n; i++) { int v = fgetc(fi); if (v < 0) return 3; p[i] = (uint16_t)v; } }
      else if (fread(p, 2, n, fi) != n) return 3;
    }
    std::list<TComPic *> out; Int numEncoded = 0;
    enc.encode(f == frames - 1, &org, out, numEncoded);
    if (!numEncoded) continue;
    TComPic *pic = out.back();
    uint32_t u[2] = { (uint32_t)pic->getPOC(), pic->getNumberOfCtusInFrame() }; fwrite(u, 4, 2, fo);
    for (UInt a = 0; a < pic->getNumberOfCtusInFrame(); a++) {
      const hm355_ctu_out *c = pic->getCtu(a);
      fwrite(&c->total_cost, 8, 1, fo); fwrite(&c->total_bits, 4, 1, fo); fwrite(&c->total_dist, 4, 1, fo);
      fwrite(c->depth, 1, 256 * 12, fo);
      fwrite(c->coeff_y, 4, 6144, fo);
    }
    for (int c = 0; c < 3; c++) fwrite(pic->getPicYuvRec()->getAddr(ComponentID(c)), 2, (size_t)org.getWidth(ComponentID(c)) * org.getHeight(ComponentID(c)), fo);
    const uint32_t ns = (uint32_t)pic->getSubstreams().size(); fwrite(&ns, 4, 1, fb);
    for (uint32_t k = 0; k < ns; k++) { const std::vector<uint8_t> &b = pic->getSubstreams()[k].getFIFO(); const uint32_t nb = (uint32_t)b.size(); fwrite(&nb, 4, 1, fb); if (nb) fwrite(b.data(), 1, nb, fb); }
  }
  enc.destroy();
  fclose(fo); fclose(fi); fclose(fb);
  return 0;
}
