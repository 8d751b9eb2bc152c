// hm355 -- CU-level mode tests of P / B slices (included by hm355_core.h after the intra CU check)
#pragma once

HM_DEV inline void check_best_mode(Shared *e, CuFrame *f, int cuZ, int cuDepth)
{ // xCheckBestMode :1702
  HM_TRACE(e, 1, e->outBits, e->outDist, e->outCost);
  if (e->outCost < f->bestCost) {
    f->bestCost = e->outCost; f->bestBits = e->outBits; f->bestDist = e->outDist;
    save_best(e, cuZ, cuDepth);
    cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
  }
}
HM_DEV HM_NOINLINE void check_rd_cost_merge_2Nx2N(Shared *e, int cuZ, int cuDepth, int sp)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); sp = HM_UNI(sp);
  CuFrame *f = &e->cuf[sp];
  CtuMeta *m = &e->meta; InterMeta *im = e->im; const int parts = 256 >> (2 * cuDepth);
  const Rect r = pu_rect(cuZ, cuDepth, SIZE_2Nx2N, 0);
  init_est_data(e, cuZ, cuDepth);
  par_set8(m->part + cuZ, SIZE_2Nx2N, parts);
  merge_candidates(e, cuZ, cuDepth, SIZE_2Nx2N, 0, &e->ml);
  auto &ml = e->ws->mrg2N;                                    // copy in HBM: merge_estimation of later modes reuses e->ml
  for (int i = 0; i < 5; i++) { ml.dir[i] = e->ml.dir[i]; for (int l = 0; l < 2; l++) { ml.mv[i][l] = e->ml.f[i][l].mv; ml.ref[i][l] = e->ml.f[i][l].ref; } }
  ml.num = e->ml.num;
  int mergeCandBuffer[5] = {0, 0, 0, 0, 0};
  int bestIsSkip = 0;
  MvD zero; zero.x = zero.y = 0;
  for (int noResidual = 0; noResidual < 2; noResidual++) {
    for (int cand = 0; cand < ml.num; cand++) {
      if (noResidual == 1 && mergeCandBuffer[cand] == 1) continue;
      if (bestIsSkip && noResidual == 0) continue;
      par_set8(m->pred + cuZ, MODE_INTER, parts); par_set8(m->part + cuZ, SIZE_2Nx2N, parts);
      par_set8(im->mrg + cuZ, 1, parts); par_set8(im->mrgIdx + cuZ, cand, parts); par_set8(im->interDir + cuZ, ml.dir[cand], parts);
      pu_set_motion(e, r, 0, ml.mv[cand][0], ml.ref[cand][0]); pu_set_motion(e, r, 1, ml.mv[cand][1], ml.ref[cand][1]);
      motion_compensation_pu(e, cuZ, r, e->ws->pred);
      { HM_PROF_BEGIN(e, PR_IRES); encode_res_and_calc_rd_inter(e, cuZ, cuDepth, noResidual != 0); HM_PROF_END(e, PR_IRES); }
      if (noResidual == 0 && !qt_root_cbf(m, cuZ)) mergeCandBuffer[cand] = 1;
      par_set8(im->skip + cuZ, !qt_root_cbf(m, cuZ), parts);
      check_dqp(e, cuZ, cuDepth);                                 // TEncCu.cpp:1500-1502
      check_best_mode(e, f, cuZ, cuDepth);
      init_est_data(e, cuZ, cuDepth);
      if (!bestIsSkip) bestIsSkip = f->bestCost < HM_MAX_DOUBLE && !qt_root_cbf(&e->ws->best[cuDepth].m, cuZ);   // FDM
    }
  }
}
HM_DEV HM_NOINLINE void check_rd_cost_inter(Shared *e, int cuZ, int cuDepth, int partSize, int useMRG, int sp)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); useMRG = HM_UNI(useMRG); sp = HM_UNI(sp);
  CuFrame *f = &e->cuf[sp];
  CtuMeta *m = &e->meta; const int parts = 256 >> (2 * cuDepth);
  init_est_data(e, cuZ, cuDepth);
  par_set8(m->part + cuZ, partSize, parts); par_set8(m->pred + cuZ, MODE_INTER, parts);
  pred_inter_search(e, cuZ, cuDepth, partSize, useMRG);
  { HM_PROF_BEGIN(e, PR_IRES); encode_res_and_calc_rd_inter(e, cuZ, cuDepth, 0); HM_PROF_END(e, PR_IRES); }
  check_dqp(e, cuZ, cuDepth);                                     // TEncCu.cpp:1563
  check_best_mode(e, f, cuZ, cuDepth);
}
// TEncSearch::m_integerMv2Nx2N as the 2Nx2N motion search of this CU leaves it (xMotionEstimation :3880-3888, :3938): the AMVP predictor and the
// integer search of every (list, reference index) the uni-directional loop of predInterSearch visits, nothing else.  The team search
// (hm355_team.h) runs it on the main wavefront so that the sub-CUs can start while a helper evaluates the CU's candidates.
HM_DEV HM_NOINLINE void me_token_prepass(Shared *e, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth);
  const InterPic *s = e->fb.ip; CtuMeta *m = &e->meta; const int parts = 256 >> (2 * cuDepth);
  init_est_data(e, cuZ, cuDepth);
  par_set8(m->part + cuZ, SIZE_2Nx2N, parts); par_set8(m->pred + cuZ, MODE_INTER, parts);
  const int numPredDir = s->sliceType == HM_P_SLICE ? 1 : 2;
  uint32_t distBiP = 0xffffffffu;
  for (int list = 0; list < numPredDir; list++)
    for (int ri = 0; ri < s->numRefIdx[list]; ri++) {
      if (list == 1 && s->list1ToList0[ri] >= 0) continue;         // GPB_SIMPLE_UNI: no search, the entry keeps its value
      int mvpIdx;
      const MvD mvPred = estimate_mvp_amvp(e, cuZ, cuDepth, SIZE_2Nx2N, 0, list, ri, &e->amvp, &mvpIdx, &distBiP);
      motion_estimation(e, cuZ, cuDepth, SIZE_2Nx2N, 0, mvPred.x, mvPred.y, (list << 4) | ri, 0, 0, 0, 0, 1);
    }
}
// the mode tests of one CU in a P / B slice (TEncCu::xCompressCU :600-857 with ESD/CFM/ECU off)
HM_DEV HM_NOINLINE void compress_cu_inter_modes(Shared *e, int cuZ, int cuDepth, int sp)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); sp = HM_UNI(sp);
  CuFrame *f = &e->cuf[sp];
  f->ampSens = 0;
  { HM_PROF_BEGIN(e, PR_MRG2N); check_rd_cost_merge_2Nx2N(e, cuZ, cuDepth, sp); HM_PROF_END(e, PR_MRG2N); }
  check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2Nx2N, 0, sp);
  check_rd_cost_inter(e, cuZ, cuDepth, SIZE_Nx2N, 0, sp);
  check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxN, 0, sp);
  if (cuDepth < 3) { // deriveTestModeAMP :386-447 on the best mode so far
    const Best *b = &e->ws->best[cuDepth];
    const int ps = b->m.part[cuZ], bmrg = b->im.mrg[cuZ], bskip = b->im.skip[cuZ], parent = f->parentPart;
    int hor = 0, ver = 0, mh = 0, mv = 0;
    if (ps == SIZE_2NxN) hor = 1;
    else if (ps == SIZE_Nx2N) ver = 1;
    else if (ps == SIZE_2Nx2N && !bmrg && !bskip) { hor = 1; ver = 1; }
    if (parent >= SIZE_2NxnU && parent <= SIZE_nRx2N) { mh = 1; mv = 1; }
    if (parent == SIZE_NONE) { if (ps == SIZE_2NxN) mh = 1; else if (ps == SIZE_Nx2N) mv = 1; }
    if (ps == SIZE_2Nx2N && !bskip) { mh = 1; mv = 1; }
    if ((64 >> cuDepth) == 64) { hor = 0; ver = 0; }
    f->ampSens = (int8_t)((!hor && !mh) || (!ver && !mv));    // a parent with an AMP part size would add merge-only AMP candidates here (team search, hm355_team.h)
    if (hor) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnU, 0, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnD, 0, sp); }
    else if (mh) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnU, 1, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnD, 1, sp); }
    if (ver) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nLx2N, 0, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nRx2N, 0, sp); }
    else if (mv) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nLx2N, 1, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nRx2N, 1, sp); }
  }
  { // intra only when the best inter mode left a residual ("avoid very complex intra if it is unlikely", :820)
    const Best *b = &e->ws->best[cuDepth];
    if (b->m.cbf[0][cuZ] != 0 || b->m.cbf[1][cuZ] != 0 || b->m.cbf[2][cuZ] != 0) {
      HM_PROF_BEGIN(e, PR_INTRA_IN_P);
      check_rd_cost_intra(e, cuZ, cuDepth, SIZE_2Nx2N); check_best_mode(e, f, cuZ, cuDepth);
      if (cuDepth == 3) { check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN); check_best_mode(e, f, cuZ, cuDepth); }
      HM_PROF_END(e, PR_INTRA_IN_P);
    }
  }
}
