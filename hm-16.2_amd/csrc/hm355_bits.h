// hm355 bitstream pass, part 1: the arithmetic-coding engine (TEncBinCABAC, TEncBinCoderCABAC.cpp:69-437, writing into a
// TComOutputBitstream, TComBitStream.cpp:96-182).  SURVEY.md 8f row n2: TEncSlice::encodeSlice.
//
// The CTU syntax functions of the search (encode_ctu and everything below it) are templates over the engine they drive:
// Cabac / CabacR count Q15 bits, CabacW below codes the same bins for real.  One substream is coded by one wavefront in
// wave-uniform control flow (every lane holds the same coder state); the coefficient staging inside code_coeff_nxn stays
// lane-parallel.  Included from hm355_core.h after the estimator engines.
#pragma once

struct CabacW {
  uint8_t s[192];                     // context states, numbering of Cabac; [HM_NUM_CTX] = SAO merge, [HM_NUM_CTX + 1] = SAO type index
  uint8_t used[192];                  // ContextModel::m_binsCoded (read by determineCabacInitIdx)
  uint32_t low, range; int32_t bitsLeft; uint32_t bufferedByte; int32_t numBufferedBytes;   // TEncBinCABAC
  uint32_t held; int32_t numHeld;     // TComOutputBitstream::m_held_bits
  uint8_t *out; uint32_t len, cap;    // byte FIFO of the substream (HBM)
  uint32_t bins;                      // TEncBinCABAC::m_uiBinsCoded
  int32_t tabLps[64], tabNlps[32];    // LDS copies of the LPS range table (4 bytes per state) and of the LPS transitions (4 states per word)
};
enum { C_SAO_MERGE = HM_NUM_CTX, C_SAO_TYPE = HM_NUM_CTX + 1 };
static_assert(HM_NUM_CTX + 2 <= 192, "SAO contexts live behind the estimator's contexts");

HM_CONST uint8_t HM_LPS_TABLE[64][4] __attribute__((aligned(4))) = {   // TComCABACTables::sm_aucLPSTable (H.265 table 9-46)
  {128,176,208,240},{128,167,197,227},{128,158,187,216},{123,150,178,205},{116,142,169,195},{111,135,160,185},{105,128,152,175},{100,122,144,166},
  { 95,116,137,158},{ 90,110,130,150},{ 85,104,123,142},{ 81, 99,117,135},{ 77, 94,111,128},{ 73, 89,105,122},{ 69, 85,100,116},{ 66, 80, 95,110},
  { 62, 76, 90,104},{ 59, 72, 86, 99},{ 56, 69, 81, 94},{ 53, 65, 77, 89},{ 51, 62, 73, 85},{ 48, 59, 69, 80},{ 46, 56, 66, 76},{ 43, 53, 63, 72},
  { 41, 50, 59, 69},{ 39, 48, 56, 65},{ 37, 45, 54, 62},{ 35, 43, 51, 59},{ 33, 41, 48, 56},{ 32, 39, 46, 53},{ 30, 37, 43, 50},{ 29, 35, 41, 48},
  { 27, 33, 39, 45},{ 26, 31, 37, 43},{ 24, 30, 35, 41},{ 23, 28, 33, 39},{ 22, 27, 32, 37},{ 21, 26, 30, 35},{ 20, 24, 29, 33},{ 19, 23, 27, 31},
  { 18, 22, 26, 30},{ 17, 21, 25, 28},{ 16, 20, 23, 27},{ 15, 19, 22, 25},{ 14, 18, 21, 24},{ 14, 17, 20, 23},{ 13, 16, 19, 22},{ 12, 15, 18, 21},
  { 12, 14, 17, 20},{ 11, 14, 16, 19},{ 11, 13, 15, 18},{ 10, 12, 15, 17},{ 10, 12, 14, 16},{  9, 11, 13, 15},{  9, 11, 12, 14},{  8, 10, 12, 14},
  {  8,  9, 11, 13},{  7,  9, 11, 12},{  7,  9, 10, 12},{  7,  8, 10, 11},{  6,  8,  9, 11},{  6,  7,  9, 10},{  6,  7,  8,  9},{  2,  2,  2,  2}
};

// ---- byte FIFO ----
HM_DEV inline void cabw_put_byte(CabacW *w, uint32_t b)
{
  if (w->len < w->cap) w->out[w->len] = (uint8_t)b;      // every lane stores the same byte to the same address
  w->len = w->len + 1;
}
HM_DEV inline void cabw_put_bits(CabacW *w, uint32_t bits, int n)
{ // TComOutputBitstream::write :96-138, MSB first
  uint32_t held = w->held; int nh = w->numHeld;
  for (int i = n - 1; i >= 0; i--) {
    held = (held << 1) | ((bits >> i) & 1); nh++;
    if (nh == 8) { cabw_put_byte(w, held); held = 0; nh = 0; }
  }
  w->held = held; w->numHeld = nh;
}
// ---- arithmetic coder ----
HM_DEV inline void cabw_load_tables(CabacW *w)
{ HM_PAR_FOR(i, 64) { w->tabLps[i] = ((const int32_t *)HM_LPS_TABLE)[i]; if (i < 32) w->tabNlps[i] = ((const int32_t *)HM_NEXT_LPS)[i]; } }
HM_DEV inline void cabw_start(CabacW *w)
{ w->low = 0; w->range = 510; w->bitsLeft = 23; w->numBufferedBytes = 0; w->bufferedByte = 0xff; }       // TEncBinCABAC::start :69
HM_DEV inline void cabw_write_out(CabacW *w)
{ // TEncBinCABAC::writeOut :403-437: the coder only ever emits whole bytes here, and the FIFO is byte aligned while it runs
  const uint32_t leadByte = w->low >> (24 - w->bitsLeft);
  w->bitsLeft += 8; w->low &= 0xffffffffu >> w->bitsLeft;
  if (leadByte == 0xff) w->numBufferedBytes++;
  else if (w->numBufferedBytes > 0) {
    const uint32_t carry = leadByte >> 8;
    cabw_put_byte(w, (w->bufferedByte + carry) & 0xff);
    w->bufferedByte = leadByte & 0xff;
    const uint32_t fill = (0xff + carry) & 0xff;
    while (w->numBufferedBytes > 1) { cabw_put_byte(w, fill); w->numBufferedBytes--; }
  } else { w->numBufferedBytes = 1; w->bufferedByte = leadByte; }
}
HM_DEV inline void enc_bin(const Shared *e, CabacW *w, int ctx, int bin)
{ // TEncBinCABAC::encodeBin :190-246 + ContextModel::updateMPS/updateLPS
  (void)e;
  const int st = w->s[ctx];
  w->bins++; w->used[ctx] = 1;
  const uint32_t lps = HM_LPS_TABLE[st >> 1][(w->range >> 6) & 3];
  uint32_t range = w->range - lps;
  if (bin != (st & 1)) {
    const int numBits = __builtin_clz(lps) - 23;          // sm_aucRenormTable[lps >> 3]: shifts that bring lps (6..240) back to >= 256
    w->low = (w->low + range) << numBits; range = lps << numBits; w->bitsLeft -= numBits;
  } else if (range < 256) { w->low <<= 1; range <<= 1; w->bitsLeft--; }
  w->range = range;
  w->s[ctx] = (uint8_t)hm_next_state(st, bin);
  if (w->bitsLeft < 12) cabw_write_out(w);
}
HM_DEV inline void enc_epv(CabacW *w, uint32_t val, int n)
{ // TEncBinCABAC::encodeBinsEP :277-311 (n bypass bins, first bin = MSB of val; encodeBinEP :253 is n == 1).  With range == 256 the
  // reference's aligned variant :318-351 leaves the same low register
  if (n <= 0) return;
  w->bins += (uint32_t)n;
  while (n > 8) {
    n -= 8;
    const uint32_t pattern = val >> n;
    w->low = (w->low << 8) + w->range * pattern; val -= pattern << n; w->bitsLeft -= 8;
    if (w->bitsLeft < 12) cabw_write_out(w);
  }
  w->low = (w->low << n) + w->range * val; w->bitsLeft -= n;
  if (w->bitsLeft < 12) cabw_write_out(w);
}
HM_DEV inline void enc_trm(const Shared *e, CabacW *w, int bin)
{ // TEncBinCABAC::encodeBinTrm :358-383
  (void)e;
  w->bins++;
  w->range -= 2;
  if (bin) { w->low += w->range; w->low <<= 7; w->range = 2 << 7; w->bitsLeft -= 7; }
  else if (w->range >= 256) return;
  else { w->low <<= 1; w->range <<= 1; w->bitsLeft--; }
  if (w->bitsLeft < 12) cabw_write_out(w);
}
HM_DEV inline void cabw_finish(CabacW *w)
{ // TEncBinCABAC::finish :81-110
  if (w->low >> (32 - w->bitsLeft)) {
    cabw_put_byte(w, (w->bufferedByte + 1) & 0xff);
    while (w->numBufferedBytes > 1) { cabw_put_byte(w, 0x00); w->numBufferedBytes--; }
    w->low -= 1u << (32 - w->bitsLeft);
  } else {
    if (w->numBufferedBytes > 0) cabw_put_byte(w, w->bufferedByte);
    while (w->numBufferedBytes > 1) { cabw_put_byte(w, 0xff); w->numBufferedBytes--; }
  }
  cabw_put_bits(w, w->low >> 8, 24 - w->bitsLeft);
}

// Register-resident form for code_coeff_nxn (the bulk of the bins): context states, the LPS range table and the LPS transitions live in lane
// registers as in CabacR (4 bytes per lane, read with v_readlane), low / range / bitsLeft / the bin count in scalars; only the byte output
// (once per 8 bits) goes through the LDS copy.  A state needs 7 bits: bit 7 of its byte carries the "coded" flag while it is in registers.
struct CabacWR { CabacW *w; HM_LV(int32_t, st); HM_LV(int32_t, lpsRow); HM_LV(int32_t, nlps); uint32_t low, range; int32_t bitsLeft; uint32_t bins; };
HM_DEV inline void cabr_load(const Shared *e, CabacWR &r, CabacW *c)
{
  (void)e;
  r.w = c;
  HM_WAVE_FOR(k) {
    HM_LVK(r.st, k) = k < 48 ? (((const int32_t *)c->s)[k] | ((((const int32_t *)c->used)[k] & 0x01010101) << 7)) : 0;
    HM_LVK(r.lpsRow, k) = c->tabLps[k];
    HM_LVK(r.nlps, k) = c->tabNlps[k & 31];
  }
  r.low = c->low; r.range = c->range; r.bitsLeft = c->bitsLeft; r.bins = c->bins;
}
HM_DEV inline void cabr_store(const CabacWR &r, CabacW *c)
{
  HM_WAVE_FOR(k) { if (k < 48) { ((int32_t *)c->s)[k] = HM_LVK(r.st, k) & 0x7f7f7f7f; ((int32_t *)c->used)[k] = (int32_t)(((uint32_t)HM_LVK(r.st, k) >> 7) & 0x01010101u); } }
  c->low = r.low; c->range = r.range; c->bitsLeft = r.bitsLeft; c->bins = r.bins;
  HM_SYNC();
}
HM_DEV inline void cabwr_write_out(CabacWR *r)
{ // cabw_write_out with low / bitsLeft in registers
  CabacW *w = r->w;
  const uint32_t leadByte = r->low >> (24 - r->bitsLeft);
  r->bitsLeft += 8; r->low &= 0xffffffffu >> r->bitsLeft;
  if (leadByte == 0xff) w->numBufferedBytes++;
  else if (w->numBufferedBytes > 0) {
    const uint32_t carry = leadByte >> 8;
    cabw_put_byte(w, (w->bufferedByte + carry) & 0xff);
    w->bufferedByte = leadByte & 0xff;
    const uint32_t fill = (0xff + carry) & 0xff;
    while (w->numBufferedBytes > 1) { cabw_put_byte(w, fill); w->numBufferedBytes--; }
  } else { w->numBufferedBytes = 1; w->bufferedByte = leadByte; }
}
HM_DEV inline void enc_bin(const Shared *e, CabacWR *r, int ctx, int bin)
{
  (void)e;
  const int wd = HM_LV_GET(r->st, ctx >> 2), sh = (ctx & 3) * 8, s = (int)(((uint32_t)wd >> sh) & 0x7f);
  const uint32_t lps = ((uint32_t)HM_LV_GET(r->lpsRow, s >> 1) >> (((r->range >> 6) & 3) * 8)) & 0xff;
  uint32_t range = r->range - lps;
  if (bin != (s & 1)) {
    const int numBits = __builtin_clz(lps) - 23;
    r->low = (r->low + range) << numBits; range = lps << numBits; r->bitsLeft -= numBits;
  } else if (range < 256) { r->low <<= 1; range <<= 1; r->bitsLeft--; }
  r->range = range; r->bins++;
  const int lw = HM_LV_GET(r->nlps, s >> 2);
  const int ns = (bin == (s & 1)) ? (s < 124 ? s + 2 : s) : ((lw >> ((s & 3) * 8)) & 0xff);
  HM_LV_SET(r->st, ctx >> 2, (int32_t)(((uint32_t)wd & ~(0xffu << sh)) | ((uint32_t)(ns | 0x80) << sh)));
  if (r->bitsLeft < 12) cabwr_write_out(r);
}
HM_DEV inline void enc_epv(CabacWR *r, uint32_t val, int n)
{
  if (n <= 0) return;
  r->bins += (uint32_t)n;
  while (n > 8) {
    n -= 8;
    const uint32_t pattern = val >> n;
    r->low = (r->low << 8) + r->range * pattern; val -= pattern << n; r->bitsLeft -= 8;
    if (r->bitsLeft < 12) cabwr_write_out(r);
  }
  r->low = (r->low << n) + r->range * val; r->bitsLeft -= n;
  if (r->bitsLeft < 12) cabwr_write_out(r);
}
// engine traits: the register-resident form code_coeff_nxn codes on, and whether the values of bypass bins matter
template <class C> struct EngOf { typedef CabacR R; enum { REAL = 0 }; };
template <> struct EngOf<CabacW> { typedef CabacWR R; enum { REAL = 1 }; };
