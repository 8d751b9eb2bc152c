// DEBUGGING AID, NOT PRODUCT: the inter-slice (P / B) path of hm-16.2_amd/csrc/hm355_core.h compiled for the host with one "lane"
// (see hostsim.cpp), driven by an HMD2 record stream of the real reference (oracle/ref_harness.cpp, `hm_dump enc2`):
// every P or B slice is re-run with the slice parameters and reference pictures of its record and compared in place.
//   hostsim_inter <in.yuv> <dump2.bin> <w> <h> <bitdepth> [wpp]   exit code 0 = every inter slice bit-exact
// 'B' records (with the 'A' record of the same picture in front, tests/hmd2.py write(bits=True)) also replay the bitstream pass
// (hm355_bits_kernel.h) on the CTU data of the preceding 'S' record, I slices included, and compare the substream bytes.
#define HM355_HOSTSIM 1
#include "../../hm-16.2_amd/csrc/hm355_core.h"
#include "../../hm-16.2_amd/csrc/hm355_bits_kernel.h"
#include "../../hm-16.2_amd/csrc/hm355_host_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>

struct FinalPic { int poc, sliceType; std::vector<Pel> buf[3]; int stride[3]; int numRef[2]; int refPoc[2][16], refLT[2][16]; std::vector<uint8_t> pm; std::vector<MvD> mv[2]; std::vector<int8_t> ri[2]; };
static const unsigned char *g_p; static size_t g_n, g_off;
template <class T> static T rd() { T v; memcpy(&v, g_p + g_off, sizeof(T)); g_off += sizeof(T); return v; }
static void rdbuf(void *d, size_t n) { memcpy(d, g_p + g_off, n); g_off += n; }

int main(int argc, char **argv)
{
  if (argc < 6) { fprintf(stderr, "usage: %s in.yuv dump2.bin w h bd\n", argv[0]); return 2; }
  const int w = atoi(argv[3]), h = atoi(argv[4]), bd = atoi(argv[5]);
  FILE *fy = fopen(argv[1], "rb"), *fd = fopen(argv[2], "rb");
  if (!fy || !fd) { perror("open"); return 1; }
  fseek(fd, 0, SEEK_END); g_n = ftell(fd); fseek(fd, 0, SEEK_SET);
  std::vector<unsigned char> data(g_n); if (fread(data.data(), 1, g_n, fd) != g_n) return 1;
  g_p = data.data(); g_off = 4;
  Params P; memset(&P, 0, sizeof(P));
  P.width = w; P.height = h; P.bitDepth = bd; P.wpp = argc > 6 ? atoi(argv[6]) : 0; P.wCtu = (w + 63) / 64; P.hCtu = (h + 63) / 64;
  P.stride[0] = P.wCtu * 64; P.stride[1] = P.stride[2] = P.wCtu * 32;
  const int nctu = P.wCtu * P.hCtu;
  Tables *tab = new Tables; hm355_build_tables(tab); P.tab = tab;
  P.ws = (WorkSpace *)calloc(1, sizeof(WorkSpace));
  std::map<int, FinalPic> finals;
  const size_t frameBytes = (size_t)w * h * 3 / 2 * (bd == 8 ? 1 : 2);
  int bad = 0, nP = 0, nB = 0;
  static Shared sh;
  // the last 'S' record (the reference's own CTU data) and 'A' record, for the bitstream pass
  std::vector<CtuMeta> lastMeta; std::vector<InterMeta> lastIm; std::vector<TCoeff> lastCoef; InterPic lastIp; int lastPoc = -1, lastQp = 0;
  std::vector<int32_t> lastSao; int saoPoc = -1, saoEn[2] = {0, 0};
  memset(&lastIp, 0, sizeof(lastIp));
#ifdef HM355_TRACE
  if (argc > 7) g_hm_trace = fopen(argv[7], "w");
#endif
  while (g_off < g_n) {
    const char tag = (char)rd<unsigned char>();
    if (tag == 'F') {
      FinalPic f; f.poc = rd<int32_t>();
      for (int c = 0; c < 3; c++) {
        const int cw = w >> (c ? 1 : 0), ch = h >> (c ? 1 : 0), mg = HM_REF_MARGIN >> (c ? 1 : 0), st = cw + 2 * mg;
        f.stride[c] = st; f.buf[c].resize((size_t)st * (ch + 2 * mg));
        std::vector<uint16_t> pl((size_t)cw * ch); rdbuf(pl.data(), pl.size() * 2);
        for (int y = -mg; y < ch + mg; y++) for (int x = -mg; x < cw + mg; x++) {   // TComPicYuv::extendPicBorder
          const int sy = y < 0 ? 0 : (y >= ch ? ch - 1 : y), sx = x < 0 ? 0 : (x >= cw ? cw - 1 : x);
          f.buf[c][(size_t)(y + mg) * st + x + mg] = (Pel)pl[(size_t)sy * cw + sx];
        }
      }
      f.sliceType = rd<int32_t>(); f.numRef[0] = rd<int32_t>(); f.numRef[1] = rd<int32_t>();
      rdbuf(f.refPoc, sizeof(f.refPoc)); rdbuf(f.refLT, sizeof(f.refLT));
      const uint32_t n = rd<uint32_t>();
      f.pm.resize((size_t)n * 256); for (int l = 0; l < 2; l++) { f.mv[l].resize((size_t)n * 256); f.ri[l].resize((size_t)n * 256); }
      for (uint32_t a = 0; a < n; a++) {
        rdbuf(&f.pm[(size_t)a * 256], 256);
        for (int l = 0; l < 2; l++) { rdbuf(&f.mv[l][(size_t)a * 256], 1024); rdbuf(&f.ri[l][(size_t)a * 256], 256); }
      }
      finals[f.poc] = f;
      continue;
    }
    if (tag == 'A') { // SAO decisions of a picture: part of its slice data
      saoPoc = rd<int32_t>(); rd<int32_t>(); saoEn[0] = rd<int32_t>(); saoEn[1] = rd<int32_t>();
      const uint32_t n = rd<uint32_t>(); lastSao.resize((size_t)n * 105); rdbuf(lastSao.data(), lastSao.size() * 4);
      continue;
    }
    if (tag == 'B') { // TEncSlice::encodeSlice of the picture of the last 'S' record
      const int poc = rd<int32_t>(); const uint32_t ns = rd<uint32_t>();
      std::vector<std::vector<uint8_t> > want(ns);
      for (uint32_t k = 0; k < ns; k++) { const uint32_t nb = rd<uint32_t>(); want[k].resize(nb); rdbuf(want[k].data(), nb); }
      const int wantNext = rd<int32_t>(); const uint32_t wantBins = rd<uint32_t>();
      if (poc != lastPoc) { fprintf(stderr, "B record of POC %d without its S record\n", poc); return 1; }
      nB++;
      FrameBuf fb; memset(&fb, 0, sizeof(fb));
      fb.meta = lastMeta.data(); fb.coef = lastCoef.data();
      if (lastIp.sliceType != 2) { fb.imeta = lastIm.data(); fb.ip = &lastIp; }
      P.frames = &fb;
      const uint32_t cap = 16384;
      std::vector<uint8_t> raw((size_t)nctu * cap); std::vector<uint32_t> sizes(P.hCtu); std::vector<CabacW> sync(P.hCtu); std::vector<uint32_t> flags(P.hCtu + 1);
      BitsParams bp; memset(&bp, 0, sizeof(bp));
      bp.sliceType = lastIp.sliceType; bp.qp = lastQp; bp.cabacInitType = lastIp.cabacInitType;
      const int en = saoPoc == poc;
      bp.saoEnabled[0] = en ? saoEn[0] : 0; bp.saoEnabled[1] = bp.saoEnabled[2] = en ? saoEn[1] : 0;
      bp.sao = (bp.saoEnabled[0] || bp.saoEnabled[1]) ? lastSao.data() : NULL;
      bp.raw = raw.data(); bp.capPerCtu = cap; bp.subSizes = sizes.data(); bp.sync = sync.data(); bp.syncFlag = flags.data(); bp.epoch = 1;
      static CabacW cw;
      const int numSub = P.wpp ? P.hCtu : 1;
      int bbad = (uint32_t)numSub != ns;
      for (int k = 0; k < numSub && !bbad; k++) {
        bits_encode_substream(&sh, &cw, &P, 0, &bp, k, 0);
        const uint8_t *got = raw.data() + (size_t)(P.wpp ? k * P.wCtu : 0) * cap;
        if (sizes[k] != want[k].size() || memcmp(got, want[k].data(), sizes[k])) { bbad = 1; printf("POC %d substream %d: %u bytes, want %zu\n", poc, k, sizes[k], want[k].size()); }
      }
      if (!bbad && (bp.bins != wantBins || bp.nextInitType != wantNext)) { bbad = 1; printf("POC %d: bins %u/%u next table %d/%d\n", poc, bp.bins, wantBins, bp.nextInitType, wantNext); }
      printf("POC %d bitstream: %s\n", poc, bbad ? "MISMATCH" : "ok");
      bad += bbad;
      continue;
    }
    if (tag != 'S') { fprintf(stderr, "bad tag at %zu\n", g_off - 1); return 1; }
    const int poc = rd<int32_t>(), sliceType = rd<int32_t>(), qp = rd<int32_t>(); rd<int32_t>(); rd<int32_t>();
    const double lambda = rd<double>(); rd<double>(); const double wcb = rd<double>(); rd<double>();
    const uint32_t lmSAD = rd<uint32_t>(), lmSSE = rd<uint32_t>();
    int numRef[2]; numRef[0] = rd<int32_t>(); numRef[1] = rd<int32_t>();
    int refPoc[2][16], refLT[2][16]; rdbuf(refPoc, sizeof(refPoc)); rdbuf(refLT, sizeof(refLT));
    int misc[7]; rdbuf(misc, sizeof(misc)); g_off += 64;
    const uint32_t n = rd<uint32_t>();
    std::vector<CtuStat> wantStat(n); std::vector<CtuMeta> wantMeta(n); std::vector<InterMeta> wantIm(n); std::vector<TCoeff> wantCoef((size_t)n * HM_COEF_CTU);
    for (uint32_t a = 0; a < n; a++) {
      wantStat[a].cost = rd<double>(); wantStat[a].bits = rd<uint32_t>(); wantStat[a].dist = rd<uint32_t>();
      rdbuf(&wantMeta[a], 12 * 256);
      InterMeta &im = wantIm[a];
      rdbuf(im.skip, 256); rdbuf(im.mrg, 256); rdbuf(im.mrgIdx, 256); rdbuf(im.interDir, 256);
      for (int l = 0; l < 2; l++) { rdbuf(im.mv[l], 1024); rdbuf(im.mvd[l], 1024); rdbuf(im.refIdx[l], 256); rdbuf(im.mvpIdx[l], 256); rdbuf(im.mvpNum[l], 256); }
      rdbuf(&wantCoef[(size_t)a * HM_COEF_CTU], HM_COEF_CTU * 4);
    }
    std::vector<uint16_t> wantRec[3];
    for (int c = 0; c < 3; c++) { wantRec[c].resize((size_t)(w >> (c ? 1 : 0)) * (h >> (c ? 1 : 0))); rdbuf(wantRec[c].data(), wantRec[c].size() * 2); }
    lastMeta = wantMeta; lastIm = wantIm; lastCoef = wantCoef; lastPoc = poc; lastQp = qp;
    memset(&lastIp, 0, sizeof(lastIp));
    lastIp.sliceType = sliceType; lastIp.numRefIdx[0] = numRef[0]; lastIp.numRefIdx[1] = numRef[1]; lastIp.mvdL1Zero = misc[3]; lastIp.maxMergeCand = misc[4]; lastIp.cabacInitType = misc[6];
    if (sliceType != HM_P_SLICE && sliceType != HM_B_SLICE) continue;
    nP++;
    FrameBuf fb; memset(&fb, 0, sizeof(fb));
    fseek(fy, (long)(frameBytes * poc), SEEK_SET);
    for (int c = 0; c < 3; c++) {
      const size_t sz = (size_t)P.stride[c] * P.hCtu * (c ? 32 : 64);
      fb.org[c] = (Pel *)calloc(sz, sizeof(Pel)); fb.rec[c] = (Pel *)calloc(sz, sizeof(Pel));
      const int pw = w >> (c ? 1 : 0), ph = h >> (c ? 1 : 0);
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
        unsigned v;
        if (bd == 8) { unsigned char t; if (fread(&t, 1, 1, fy) != 1) return 3; v = t; } else { unsigned short t; if (fread(&t, 2, 1, fy) != 1) return 3; v = t; }
        fb.org[c][y * P.stride[c] + x] = (Pel)v;
      }
    }
    fb.meta = (CtuMeta *)calloc(nctu, sizeof(CtuMeta)); fb.coef = (TCoeff *)calloc((size_t)nctu * HM_COEF_CTU, sizeof(TCoeff));
    fb.stat = (CtuStat *)calloc(nctu, sizeof(CtuStat)); fb.endState = (Cabac *)calloc(nctu, sizeof(Cabac));
    fb.imeta = (InterMeta *)calloc(nctu, sizeof(InterMeta)); fb.intMv = (MvD *)calloc((size_t)nctu * 32, sizeof(MvD));
    InterPic *ip = (InterPic *)calloc(1, sizeof(InterPic)); fb.ip = ip;
    ip->sliceType = sliceType; ip->poc = poc; ip->numRefIdx[0] = numRef[0]; ip->numRefIdx[1] = numRef[1];
    ip->colFromL0 = misc[0]; ip->colRefIdx = misc[1]; ip->tmvp = misc[2]; ip->mvdL1Zero = misc[3]; ip->maxMergeCand = misc[4]; ip->checkLDC = misc[5]; ip->cabacInitType = misc[6];
    ip->lambdaMotionSAD = lmSAD; ip->lambdaMotionSSE = lmSSE;
    for (int l = 0; l < 2; l++) for (int i = 0; i < numRef[l]; i++) {
      FinalPic &f = finals[refPoc[l][i]]; RefPicDev &r = ip->ref[l][i];
      for (int c = 0; c < 3; c++) { const int mg = HM_REF_MARGIN >> (c ? 1 : 0); r.plane[c] = f.buf[c].data() + (size_t)mg * f.stride[c] + mg; r.stride[c] = f.stride[c]; }
      r.poc = f.poc; r.isLongTerm = refLT[l][i]; r.predMode = f.pm.data();
      for (int ll = 0; ll < 2; ll++) { r.mv[ll] = f.mv[ll].data(); r.refIdx[ll] = f.ri[ll].data(); memcpy(r.refPoc[ll], f.refPoc[ll], sizeof(r.refPoc[ll])); memcpy(r.refLT[ll], f.refLT[ll], sizeof(r.refLT[ll])); }
    }
    for (int i1 = 0; i1 < numRef[1]; i1++) { ip->list1ToList0[i1] = -1; for (int i0 = 0; i0 < numRef[0]; i0++) if (refPoc[0][i0] == refPoc[1][i1]) { ip->list1ToList0[i1] = i0; break; } }
    hm355_fill_slice_params(&fb, bd, qp, lambda, wcb);
    P.frames = &fb;
    for (int a = 0; a < nctu; a++) { WorkItem it; it.frame = 0; it.ctuX = a % P.wCtu; it.ctuY = a / P.wCtu; it.pad = 0; process_ctu(&sh, &P, &it, 0); }
    int slcBad = 0;
    for (int a = 0; a < nctu; a++) {
      const char *what = 0;
      if (fb.stat[a].cost != wantStat[a].cost || fb.stat[a].bits != wantStat[a].bits || fb.stat[a].dist != wantStat[a].dist) what = "cost/bits/dist";
      else if (memcmp(&fb.meta[a], &wantMeta[a], sizeof(CtuMeta))) what = "decision arrays";
      else if (memcmp(&fb.imeta[a], &wantIm[a], sizeof(InterMeta))) what = "motion arrays";
      else if (memcmp(fb.coef + (size_t)a * HM_COEF_CTU, &wantCoef[(size_t)a * HM_COEF_CTU], HM_COEF_CTU * 4)) what = "coefficients";
      if (what) { slcBad++; if (slcBad <= 3) printf("POC %d CTU %d: %s differ (got %.1f/%u/%u want %.1f/%u/%u)\n", poc, a, what, fb.stat[a].cost, fb.stat[a].bits, fb.stat[a].dist, wantStat[a].cost, wantStat[a].bits, wantStat[a].dist); }
    }
    for (int c = 0; c < 3; c++) {
      const int pw = w >> (c ? 1 : 0), ph = h >> (c ? 1 : 0);
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) if ((uint16_t)fb.rec[c][y * P.stride[c] + x] != wantRec[c][(size_t)y * pw + x]) { slcBad++; y = ph; break; }
    }
    printf("POC %d: %s\n", poc, slcBad ? "MISMATCH" : "ok");
    bad += slcBad;
  }
  printf("%d P slices, %d bitstream passes, %s\n", nP, nB, bad ? "MISMATCH" : "all bit-exact");
  return bad ? 1 : 0;
}
