// Small driver in the shape of the reference's TAppEncoder loop (TAppEncTop.cpp:407-520): reads a planar
// 4:2:0 file, feeds TEncTop::encode picture by picture and dumps what compressSlice left behind in the
// same "HMD1" format the reference harness writes (oracle/ref_harness.cpp), for the parity tests.
//   hm355_encmain <in.yuv> <w> <h> <bitdepth> <frames> <qp> <wpp> <dump.bin> [lf]     lf: run deblocking + SAO, the dump then holds the finished pictures
//   ... <dump.bin> ldp | ldb | ra [aq[range]] : (aq: --AdaptiveQP=1) the GOP table of cfg/encoder_lowdelay_P_main.cfg / encoder_lowdelay_main.cfg / encoder_randomaccess_main10.cfg (IntraPeriod -1, GOPSize 4, P slices with up to 4 references, loop filters
//   on); the dump is then a "HMD3" stream in coding order: per picture i32 poc, sliceType, qp, depth, cabacInitType, numRefIdx0, numRefIdx1, colFromL0, mvdL1Zero, refPoc[2][16]; f64 lambda;
//   u32 numCtus; per CTU the record of tests/hmd2.py CTU_DT (cost, bits, dist, decision arrays, motion arrays, coefficients); the finished planes.
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
  const bool ldb = argc > 9 && !strcmp(argv[9], "ldb");            // cfg/encoder_lowdelay_main.cfg: the same table with B slices
  const bool ra = argc > 9 && !strcmp(argv[9], "ra");              // cfg/encoder_randomaccess_main*.cfg
  const bool ldp = ldb || ra || (argc > 9 && !strcmp(argv[9], "ldp"));
  if (argc > 9 && (!strcmp(argv[9], "lf") || ldp)) { enc.setLoopFilterDisable(false); enc.setUseSAO(true); }
  if (argc > 10 && !strncmp(argv[10], "aq", 2)) { enc.setUseAdaptiveQP(true); if (argv[10][2]) enc.setQPAdaptationRange(atoi(argv[10] + 2)); }   // --AdaptiveQP=1 [--MaxQPAdaptationRange=n]
  if (ra) { // cfg/encoder_randomaccess_main10.cfg:20-31
    enc.setIntraPeriod(32); enc.setGOPSize(8);
    static const int poc[8] = { 8, 4, 2, 1, 3, 6, 5, 7 }, qpOff[8] = { 1, 2, 3, 4, 4, 3, 4, 4 }, tid[8] = { 0, 0, 0, 1, 1, 0, 1, 1 }, act[8] = { 4, 2, 2, 2, 2, 2, 2, 2 }, num[8] = { 4, 3, 4, 4, 4, 4, 4, 4 };
    static const double qpFac[8] = { 0.442, 0.3536, 0.3536, 0.68, 0.68, 0.3536, 0.68, 0.68 };
    static const int refs[8][4] = { { -8, -10, -12, -16 }, { -4, -6, 4, 0 }, { -2, -4, 2, 6 }, { -1, 1, 3, 7 }, { -1, -3, 1, 5 }, { -2, -4, -6, 2 }, { -1, -5, 1, 3 }, { -1, -3, -7, 1 } };
    for (int i = 0; i < 8; i++) {
      GOPEntry e; e.m_sliceType = 'B'; e.m_POC = poc[i]; e.m_QPOffset = qpOff[i]; e.m_QPFactor = qpFac[i]; e.m_temporalId = tid[i]; e.m_numRefPicsActive = act[i]; e.m_numRefPics = num[i];
      for (int k = 0; k < num[i]; k++) e.m_referencePics[k] = refs[i][k];
      enc.setGOPEntry(i, e);
    }
  } else if (ldp) { // cfg/encoder_lowdelay_P_main.cfg:20-27
    enc.setIntraPeriod(-1); enc.setGOPSize(4);
    static const int qpOff[4] = { 3, 2, 3, 1 }; static const double qpFac[4] = { 0.4624, 0.4624, 0.4624, 0.578 };
    static const int refs[4][4] = { { -1, -5, -9, -13 }, { -1, -2, -6, -10 }, { -1, -3, -7, -11 }, { -1, -4, -8, -12 } };
    for (int i = 0; i < 4; i++) {
      GOPEntry e; e.m_sliceType = ldb ? 'B' : 'P'; e.m_POC = i + 1; e.m_QPOffset = qpOff[i]; e.m_QPFactor = qpFac[i]; e.m_temporalId = 0; e.m_numRefPicsActive = 4; e.m_numRefPics = 4;
      for (int k = 0; k < 4; k++) e.m_referencePics[k] = refs[i][k];
      enc.setGOPEntry(i, e);
    }
  }
  enc.create(); enc.init();
  FILE *fb = fopen((std::string(argv[8]) + ".bits").c_str(), "wb");
  if (!fb) { perror("open"); return 1; }
  fwrite(ldp ? "HMD3" : "HMD1", 1, 4, fo);
  uint32_t hdr[5] = { (uint32_t)w, (uint32_t)h, (uint32_t)bd, 64, (uint32_t)frames }; fwrite(hdr, 4, 5, fo);
  TComPicYuv org; org.create(w, h);
  size_t dumped = 0;
  for (int f = 0; f < frames; f++) {
    for (int c = 0; c < 3; c++) {
      const size_t n = (size_t)org.getWidth(ComponentID(c)) * org.getHeight(ComponentID(c)); uint16_t *p = org.getAddr(ComponentID(c));
      if (bd == 8) { for (size_t i = 0; i < n; i++) { int v = fgetc(fi); if (v < 0) return 3; p[i] = (uint16_t)v; } }
      else if (fread(p, 2, n, fi) != n) return 3;
    }
    std::list<TComPic *> out; Int numEncoded = 0;
    enc.encode(f == frames - 1, &org, out, numEncoded);
    if (!numEncoded) continue;
    if (ldp) { // the pictures this call encoded, in coding order
      const std::vector<TComPic *> &coded = enc.getGOPEncoder()->getCodedPictures();
      for (; dumped < coded.size(); dumped++) {
        TComPic *pic = coded[dumped];
        TComSlice *sl = pic->getSlice(0);
        int32_t h[9 + 32] = { pic->getPOC(), (int32_t)sl->getSliceType(), sl->getSliceQp(), sl->getDepth(), sl->getCabacInitType(), sl->getNumRefIdx(REF_PIC_LIST_0),
                              sl->getNumRefIdx(REF_PIC_LIST_1), (int32_t)sl->getColFromL0Flag(), sl->getMvdL1ZeroFlag() ? 1 : 0 };
        for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) h[9 + 16 * l + i] = i < sl->getNumRefIdx(RefPicList(l)) ? sl->getRefPOC(RefPicList(l), i) : 0;
        fwrite(h, 4, 41, fo);
        const double lambda = sl->getLambda(); fwrite(&lambda, 8, 1, fo);
        const uint32_t n = pic->getNumberOfCtusInFrame(); fwrite(&n, 4, 1, fo);
        for (UInt a = 0; a < n; a++) {
          const hm355_ctu_out *c = pic->getCtu(a); const hm355_ctu_inter_out *m = pic->getCtuInter(a);
          fwrite(&c->total_cost, 8, 1, fo); fwrite(&c->total_bits, 4, 1, fo); fwrite(&c->total_dist, 4, 1, fo);
          fwrite(c->depth, 1, 256 * 12, fo);
          fwrite(m->skip, 1, 256 * 4, fo);
          for (int l = 0; l < 2; l++) { fwrite(m->mv[l], 2, 512, fo); fwrite(m->mvd[l], 2, 512, fo); fwrite(m->ref_idx[l], 1, 256, fo); fwrite(m->mvp_idx[l], 1, 256, fo); fwrite(m->mvp_num[l], 1, 256, fo); }
          fwrite(c->coeff_y, 4, 6144, fo);
        }
        for (int c = 0; c < 3; c++) fwrite(pic->getPicYuvRec()->getAddr(ComponentID(c)), 2, (size_t)org.getWidth(ComponentID(c)) * org.getHeight(ComponentID(c)), fo);
        const uint32_t ns = (uint32_t)pic->getSubstreams().size(); fwrite(&ns, 4, 1, fb);
        for (uint32_t k = 0; k < ns; k++) { const std::vector<uint8_t> &b = pic->getSubstreams()[k].getFIFO(); const uint32_t nb = (uint32_t)b.size(); fwrite(&nb, 4, 1, fb); if (nb) fwrite(b.data(), 1, nb, fb); }
      }
      continue;
    }
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
