// TEST INFRASTRUCTURE ONLY -- part of oracle/ref_shim_enc.cpp (included inside its anonymous namespace).
//
// InterSearch::predInterSearch (EncoderLib/InterSearch.cpp:2245-3065) inside the REAL encoder with ONE device call per CU:
//
//   gather   what the translational part of the member reads from the CU recursion, before it runs: the real AMVP lists of every (list, refIdx)
//            (PU::fillMvpCand), m_uniMvList, the block-vector cache hits of the mode control, the ASR search ranges, lambda, the index-bit table
//   device   vtmhip_predInterSearch_batch_dev: xEstimateMvPredAMVP + xMotionEstimation of every row, xCheckBestMVP, the FastMEForGenBLowDelay copies, the best
//            reference per list, the bi refinement, the SMVD block and the uni / bi decision -- one upload, one chain of launches, one download
//   replay   the reference's OWN predInterSearch then runs over the device's results: its member calls (xEstimateMvPredAMVP, xMotionEstimation, xGetSymmetricCost,
//            xSymmetricMotionEstimation, symmvdCheckBestMvp -- the link-level hooks of ref_shim_enc.cpp) are served from the downloaded tables instead of computing.
//            Every served call first checks that the arguments the reference passes in (predictor, bits on entry, start vector, search state) are the ones the
//            DEVICE's glue derived for that call -- so the device's composition of the members is checked against the reference's own code call by call --
//            and all host-side state of the member (m_uniMvList, g_reusedUniMVs, the block-vector cache, cu.refIdxBi, the affine search that follows, the final
//            motion compensation) stays the reference's.
//   compare  (default) the served members ALSO run the reference's code and the two results are compared; VTMREF_REPLACE=1: they do not (replace mode: the
//            translational motion search of the encoder is the device's alone).
//   final    what the member leaves in `pu` (interDir, vectors, vector differences, predictor indices, reference indices, smvdMode) and the best translational cost
//            against the device's own decision record.
//   record   VTMREF_PIS_DUMP=<file> (no device needed): the gathered inputs and what the reference's members returned, as flat records -- the golden file
//            tests/golden/pis_enc.npz is made from such a dump (tests/golden/gen_pis_golden.py).


constexpr int PIS_MAX_REF = VTMHIP_MAX_REF, PIS_ROWS = 2 * VTMHIP_MAX_REF;

// One block of page-locked host memory mirrored by one device block of the same layout: every table of a one-PU vtmhip_pis_level
struct PisSlots
{
  vtmhip_me_job    uniJobs[PIS_ROWS];
  vtmhip_me_out    uniOut[PIS_ROWS];
  vtmhip_pis_row   uniRows[PIS_ROWS];
  uint64_t         distBiP[PIS_ROWS];
  vtmhip_pis_pu    pus[1];
  vtmhip_pis_pu_in puIn[1];
  vtmhip_pred_job  predOther[1];
  vtmhip_me_job    biJobs[PIS_MAX_REF];
  vtmhip_me_out    biOut[PIS_MAX_REF];
  vtmhip_pis_row   biRows[PIS_MAX_REF];
  vtmhip_smvd_job  smvd[1];
  int64_t          pos[1];
  int16_t          org[128 * 128];       // the PU's original block, stride = width (uploaded up to width * height samples)
};

struct PisFinal     // what predInterSearch leaves behind (translational part)
{
  int32_t  ran;               // the translational part ran (checkNonAffine)
  int32_t  affine;            // cu.affine on return: the fields below are then the affine winner's, only hevcCost is comparable
  int32_t  interDir, smvdMode, refIdx[2], mv[2][2], mvd[2][2], mvpIdx[2], mvpNum[2], refIdxBi[2];
  int32_t  biList;            // the list the bi iteration searched, -1: no bi stage
  int32_t  smvdRan;
  uint64_t hevcCost;          // m_affineMotion.hevcCost[cu.imv] = uiHevcCost (:3054-3057)
};

struct PisHeader    // one dump record: PisHeader, PisSlots (inputs), PisSlots (what the reference's members returned), PisFinal
{
  uint32_t magic, bytes;
  int32_t  poc, x, y, w, h, imv, picW, picH, ctuSize, bitDepth;
  int32_t  numRef[2], smvdBit, symRefIdx[2], hasSmvd, biRestricted, mvdL1Zero, fdm, list1FromList0[PIS_MAX_REF];
  uint32_t mbBits[3];
  int32_t  bipredSearchRange, useHadME, fen13, extendedSettings, firstSearchStop, uniMvListSize;
  int32_t  rowPlane[PIS_ROWS];   // index of the row's reference plane among the dumped planes (PisPlaneHeader records, in file order)
  int64_t  rowOff[PIS_ROWS];     // sample offset of the PU position (vector 0,0) inside that plane's dump
  int32_t  rowCached[PIS_ROWS], rowCalls[PIS_ROWS];   // block-vector cache hit; the reference searched the row (1) or copied it from list 0 (0)
  // (appended in round 4; records of round 3 end above and read as zeros here)
  int32_t  refPoc[2][PIS_MAX_REF], curPoc;            // POCs of the reference pictures and of the current picture (BcwFast's same-POC skip :2588-2593)
  int32_t  givenRows;                                 // vtmhip_pis_level::givenRows: rows served by xReadBufferedUniMv (a CU-level BCW weight other than the default)
  int32_t  bcwIdx, bcwNoBi;                           // cu.BcwIdx on entry; the bi stage is switched off by the BcwFast rule of :2462-2464 (folded into biRestricted)
};
constexpr uint64_t PIS_COST_UNKNOWN = ~0ull - 1;      // PisFinal::hevcCost when the member did not store its translational cost (:3054-3057: a non-default weight with both affine models buffered)
struct PisPlaneHeader { uint32_t magic, bytes; int32_t poc, stride, margin, width, height, index; };   // followed by (height + 2 margin) * stride samples
constexpr uint32_t PIS_MAGIC = 0x50495331, PIS_PLANE_MAGIC = 0x50495332;

struct PisReplay
{
  bool      active = false, replace = false, record = false;
  PisSlots *s = nullptr;               // the downloaded tables (compare / replace) or the recorded member results (record)
  int       numRef[2] = { 0, 0 }, w = 0, h = 0;
  AMVPInfo  amvp[2][PIS_MAX_REF];
  bool      cached[PIS_ROWS] = {}, given[PIS_ROWS] = {};
  int       refineList = -1, amvpServed = 0, meServed = 0, smvdCostCalls = 0;
  bool      hasSmvd = false, inMember = false;
  int       row( int list, int ref ) const { return ( list ? numRef[0] : 0 ) + ref; }
};
thread_local PisReplay g_rp;      // (one predInterSearch call per encoder thread at a time)
bool        g_hookPis = false;
uint64_t    g_pisCtr = 0, g_pisDumpCtr = 0, g_pisDumpStride = 1, g_pisDumpBcwCtr = 0, g_pisDumpBcwStride = 0;   // (BcwStride != 0: the calls at a non-default BCW weight are sampled with their own stride)
FILE       *g_pisDump = nullptr;
std::vector<std::pair<const Picture *, int>> g_pisDumpedPlanes;   // (picture buffer, POC) in dump order
decltype( &vtmhip_predInterSearch_batch_dev ) g_apiPis = nullptr;
decltype( &vtmhip_is_uniform_shape )          g_apiUniformShape = nullptr;
decltype( &vtmhip_host_alloc )                g_apiHostAlloc = nullptr;


void pisNote( int what, int a, int b, int c, long long ref, long long dev )   // what: 0 final decision, 1 AMVP, 2 uni ME, 3 bi ME, 4 SMVD, 5 replay argument check
{
  if( ST_INC( pisMismatch[what] ) == 0 && g_st->pisFirstMismatch[0] == 0 )
  {
    const int32_t v[8] = { what + 1, a, b, c, ( int32_t ) ref, ( int32_t ) dev, ( int32_t ) ( ref >> 32 ), ( int32_t ) ( dev >> 32 ) };
    memcpy( g_st->pisFirstMismatch, v, sizeof( v ) );
  }
}

// the calling encoder thread's context and slots (created on its first call: the first thread takes the context ref_encode made, every further thread -- the split-parallel
// build's OpenMP workers -- its own)
HookThread *hookThread()
{
  if( t_hk && t_hk->gen == g_hkGen ) return t_hk->ok ? t_hk : nullptr;      // (a record of an earlier ref_encode call of this process is stale)
  if( g_countOnly || !g_ctx ) return nullptr;
  HookThread *T = new HookThread();
  T->gen = g_hkGen;
  {
    std::lock_guard<std::mutex> lock( g_hkMutex );
    if( g_hkAll.empty() ) T->ctx = g_ctx;
    g_hkAll.push_back( T );
  }
  if( !T->ctx ) { T->ownsCtx = A.create( 0, &T->ctx ) == VTMHIP_OK; if( !T->ownsCtx ) T->ctx = nullptr; }
  t_hk = T;
  constexpr size_t HK_BLK = 128 * 128 * 2;
  T->ok = T->ctx && g_apiHostAlloc && g_apiHostAlloc( T->ctx, sizeof( PisSlots ), &T->pisHost ) == VTMHIP_OK
       && A.dalloc( T->ctx, sizeof( PisSlots ), ( void ** ) &T->d_pis ) == VTMHIP_OK && A.dalloc( T->ctx, HK_BLK, ( void ** ) &T->d_pisOrgBi ) == VTMHIP_OK
       && A.dalloc( T->ctx, HK_BLK, ( void ** ) &T->d_org ) == VTMHIP_OK && A.dalloc( T->ctx, HK_BLK, ( void ** ) &T->d_other ) == VTMHIP_OK
       && A.dalloc( T->ctx, 4096, &T->d_job ) == VTMHIP_OK && A.dalloc( T->ctx, 4096, &T->d_out ) == VTMHIP_OK;
  if( !T->ok ) note_error( T->ctx );
  return T->ok ? T : nullptr;
}
void hookThreadsFree()
{
  std::lock_guard<std::mutex> lock( g_hkMutex );
  for( HookThread *T : g_hkAll )
  {
    if( T->ctx )
    {
      for( void *d : { ( void * ) T->d_pis, ( void * ) T->d_pisOrgBi, ( void * ) T->d_org, ( void * ) T->d_other, T->d_job, T->d_out } ) if( d ) A.dfree( T->ctx, d );
      if( T->ownsCtx ) A.destroy( T->ctx );
    }
    T->ok = false; T->ctx = nullptr;      // (the records stay allocated: a worker thread of the OpenMP pool may outlive this encode with its t_hk pointer)
  }
  g_hkAll.clear();
  g_hkGen++;
}
int hookThreadsCount() { std::lock_guard<std::mutex> lock( g_hkMutex ); int n = 0; for( HookThread *T : g_hkAll ) n += T->ctx != nullptr; return n; }

int pisDumpPlane( const Picture *pic )
{
  for( size_t i = 0; i < g_pisDumpedPlanes.size(); i++ ) if( g_pisDumpedPlanes[i].first == pic && g_pisDumpedPlanes[i].second == pic->getPOC() ) return ( int ) i;
  const CPelBuf y = pic->getRecoBuf( COMPONENT_Y );
  const int     m = pic->margin;
  PisPlaneHeader ph = { PIS_PLANE_MAGIC, 0, pic->getPOC(), ( int32_t ) y.stride, m, ( int32_t ) y.width, ( int32_t ) y.height, ( int32_t ) g_pisDumpedPlanes.size() };
  const size_t samples = size_t( y.height + 2 * m ) * y.stride;
  ph.bytes = ( uint32_t ) ( samples * 2 );
  // rows -margin .. height + margin - 1, each `stride` samples from column -margin (the last row only up to its last sample: pad the tail with zeros)
  std::vector<Pel> buf( samples, 0 );
  const Pel *src = y.buf - ptrdiff_t( m ) * y.stride - m;
  const size_t tail = y.stride > int( y.width ) + 2 * m ? size_t( y.stride - y.width - 2 * m ) : 0;
  memcpy( buf.data(), src, ( samples - tail ) * sizeof( Pel ) );
  fwrite( &ph, sizeof( ph ), 1, g_pisDump );
  fwrite( buf.data(), sizeof( Pel ), samples, g_pisDump );
  g_pisDumpedPlanes.emplace_back( pic, pic->getPOC() );
  return ( int ) g_pisDumpedPlanes.size() - 1;
}

// ---- the members, served from the tables ------------------------------------------------------------------------------------------------------------
// xEstimateMvPredAMVP (:3088-3128): the AMVP stage wrote the chosen candidate into the row's job (mvpIdx, mvPred) and the template cost into distBiP
void pisServeAmvp( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, int iRefIdx, Mv &rcMvPred, AMVPInfo &info, bool bFilled, Distortion *puiDistBiP )
{
  PisSlots &S = *g_rp.s;
  const int row = g_rp.row( eRefPicList, iRefIdx );
  g_rp.amvpServed++;
  if( g_rp.record || !g_rp.replace )
  {
    vtmref_orig_xEstimateMvPredAMVP( is, pu, origBuf, eRefPicList, iRefIdx, rcMvPred, info, bFilled, puiDistBiP );
    const AMVPInfo &mine = g_rp.amvp[eRefPicList][iRefIdx];
    bool same = mine.numCand == info.numCand;
    for( int i = 0; same && i < info.numCand; i++ ) same = mine.mvCand[i] == info.mvCand[i];
    if( !same ) pisNote( 5, 1, row, info.numCand, 0, 0 );      // the list gathered before the member ran is not the list the member derives
    if( g_rp.record )
    {
      vtmhip_me_job &j = S.uniJobs[row];
      j.mvpIdx = ( uint8_t ) pu.mvpIdx[eRefPicList]; j.mvPredHor = rcMvPred.hor; j.mvPredVer = rcMvPred.ver; j.bits += is->m_auiMVPIdxCost[j.mvpIdx][AMVP_MAX_NUM_CANDS];
      S.distBiP[row] = puiDistBiP ? *puiDistBiP : 0;
      return;
    }
    const vtmhip_me_job &j = S.uniJobs[row];
    if( j.mvpIdx != pu.mvpIdx[eRefPicList] || j.mvPredHor != rcMvPred.hor || j.mvPredVer != rcMvPred.ver || ( puiDistBiP && S.distBiP[row] != *puiDistBiP ) )
      pisNote( 1, row, j.mvpIdx - pu.mvpIdx[eRefPicList], j.mvPredHor - rcMvPred.hor, puiDistBiP ? ( long long ) *puiDistBiP : 0, ( long long ) S.distBiP[row] );
    return;
  }
  const vtmhip_me_job &j = S.uniJobs[row];
  info = g_rp.amvp[eRefPicList][iRefIdx];
  rcMvPred.set( j.mvPredHor, j.mvPredVer );
  pu.mvpIdx[eRefPicList] = j.mvpIdx;
  pu.mvpNum[eRefPicList] = info.numCand;
  if( puiDistBiP ) *puiDistBiP = S.distBiP[row];
}

// xMotionEstimation (:3299-3494), uni rows and the rows of the bi iteration
void pisServeMe( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, Mv &rcMvPred, int iRefIdxPred, Mv &rcMv, int &riMVPIdx, uint32_t &ruiBits,
                 Distortion &ruiCost, const AMVPInfo &amvpInfo, bool bBi )
{
  PisSlots &S = *g_rp.s;
  const int row = g_rp.row( eRefPicList, iRefIdxPred );
  g_rp.meServed++;
  vtmhip_me_job &j = bBi ? S.biJobs[iRefIdxPred] : S.uniJobs[row];
  vtmhip_me_out &o = bBi ? S.biOut[iRefIdxPred] : S.uniOut[row];
  if( g_rp.record )
  {
    if( bBi )
    {
      g_rp.refineList = eRefPicList;
      j.mvPredHor = rcMvPred.hor; j.mvPredVer = rcMvPred.ver; j.mvHor = rcMv.hor; j.mvVer = rcMv.ver; j.mvpIdx = ( uint8_t ) riMVPIdx; j.bits = ruiBits;
    }
    vtmref_orig_xMotionEstimation( is, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
    o.mvHor = rcMv.hor; o.mvVer = rcMv.ver; o.mvPredHor = rcMvPred.hor; o.mvPredVer = rcMvPred.ver; o.mvpIdx = riMVPIdx; o.bits = ruiBits; o.cost = ruiCost;
    o.intX = o.intY = 0; o.intDist = 1;      // intDist != 0 marks a recorded row
    return;
  }
  // the arguments of this call against what the device's glue derived for the row
  bool argsOk = j.mvPredHor == rcMvPred.hor && j.mvPredVer == rcMvPred.ver && j.mvpIdx == riMVPIdx && j.bits == ruiBits;
  if( bBi ) argsOk = argsOk && eRefPicList == g_rp.refineList && j.mvHor == rcMv.hor && j.mvVer == rcMv.ver;
  if( !argsOk ) pisNote( 5, bBi ? 3 : 2, row, ( int ) j.bits - ( int ) ruiBits, rcMvPred.hor, j.mvPredHor );
  if( !g_rp.replace || !argsOk )
  {
    vtmref_orig_xMotionEstimation( is, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
    if( !argsOk ) { ST_INC( pisReplayFallback ); return; }
    if( o.mvHor != rcMv.hor || o.mvVer != rcMv.ver || o.mvPredHor != rcMvPred.hor || o.mvPredVer != rcMvPred.ver || o.mvpIdx != riMVPIdx || o.bits != ruiBits || o.cost != ruiCost )
      pisNote( bBi ? 3 : 2, row * 10 + pu.cu->imv, rcMv.hor - o.mvHor, rcMv.ver - o.mvVer, ( long long ) ruiCost, ( long long ) o.cost );
    return;
  }
  rcMv.set( o.mvHor, o.mvVer ); rcMvPred.set( o.mvPredHor, o.mvPredVer ); riMVPIdx = o.mvpIdx; ruiBits = o.bits; ruiCost = o.cost;
  if( !bBi && !g_rp.cached[row] && !g_rp.given[row] )      // the member's own side effect (:3449-3456): the integer vector enters the block-vector cache (a buffered row returns before it)
  {
    auto blkCache = dynamic_cast<CacheBlkInfoCtrl *>( is->m_modeCtrl );
    const Mv intMv( o.intX, o.intY );
    if( blkCache ) blkCache->setMv( pu.cs->area, eRefPicList, iRefIdxPred, intMv ); else is->m_integerMv2Nx2N[eRefPicList][iRefIdxPred] = intMv;
  }
}

// the SMVD block (:2656-2790) from vtmhip_smvd_job::trace
Distortion pisServeSmvdCost( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eCur, const MvField &cur, MvField &tar, int bcwIdx )
{
  const vtmhip_smvd_job &j = g_rp.s->smvd[0];
  if( g_rp.inMember || g_rp.record || !g_rp.hasSmvd ) return vtmref_orig_xGetSymmetricCost( is, pu, origBuf, eCur, cur, tar, bcwIdx );
  // the predictor-pair loop: the device kept the first minimum, pair trace[0].idx at cost trace[0].cost
  const int  i0 = j.trace[0].idx[0], i1 = j.trace[0].idx[1];
  const bool isBest = cur.mv.hor == j.cand[0][i0][0] && cur.mv.ver == j.cand[0][i0][1] && tar.mv.hor == j.cand[1][i1][0] && tar.mv.ver == j.cand[1][i1][1];
  g_rp.smvdCostCalls++;
  if( !g_rp.replace )
  {
    const Distortion ref = vtmref_orig_xGetSymmetricCost( is, pu, origBuf, eCur, cur, tar, bcwIdx );
    if( isBest ? ref != j.trace[0].cost : ref < j.trace[0].cost ) pisNote( 4, 0, isBest, g_rp.smvdCostCalls, ( long long ) ref, ( long long ) j.trace[0].cost );
    return ref;
  }
  return isBest ? j.trace[0].cost : j.trace[0].cost + 1 + g_rp.smvdCostCalls;
}

void pisServeSmvdCheck( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, Mv curMv, RefPicList curList, AMVPInfo amvpInfo[2][33], int32_t bcwIdx, Mv predSym[2], int32_t idxSym[2],
                        Distortion &bestCost, bool skip )
{
  vtmhip_smvd_job &j = g_rp.s->smvd[0];
  if( g_rp.record || !g_rp.hasSmvd || !g_rp.replace )
  {
    g_rp.inMember = true;
    vtmref_orig_symmvdCheckBestMvp( is, pu, origBuf, curMv, curList, amvpInfo, bcwIdx, predSym, idxSym, bestCost, skip );
    g_rp.inMember = false;
    if( g_rp.record )
    {
      if( skip ) { j.trace[3].cost = bestCost; j.trace[3].idx[0] = idxSym[curList]; j.trace[3].idx[1] = idxSym[1 - curList]; j.trace[3].mv[0] = curMv.hor; j.trace[3].mv[1] = curMv.ver; }
      return;
    }
    if( g_rp.hasSmvd && skip && ( bestCost != j.trace[3].cost || idxSym[curList] != j.trace[3].idx[0] || idxSym[1 - curList] != j.trace[3].idx[1] ) )
      pisNote( 4, 3, idxSym[curList] * 2 + idxSym[1 - curList], j.trace[3].idx[0] * 2 + j.trace[3].idx[1], ( long long ) bestCost, ( long long ) j.trace[3].cost );
    return;
  }
  // replace mode.  Start-vector loop (skip == false): only the state after the whole loop matters to the reference's code -- the start vector that holds the final
  // minimum reports it, every other start leaves the state alone.  Final check (skip == true): trace[3].
  const int  k = skip ? 3 : 1;
  const bool mine = skip || ( curMv.hor == j.trace[1].mv[0] && curMv.ver == j.trace[1].mv[1] );
  if( mine && j.trace[k].cost < bestCost )
  {
    const int tarList = 1 - curList, i0 = j.trace[k].idx[0], i1 = j.trace[k].idx[1];
    bestCost = j.trace[k].cost;
    idxSym[curList] = i0; idxSym[tarList] = i1;
    predSym[curList].set( j.cand[0][i0][0], j.cand[0][i0][1] ); predSym[tarList].set( j.cand[1][i1][0], j.cand[1][i1][1] );
  }
}

void pisServeSmvdMe( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, Mv &predCur, Mv &predTar, RefPicList eCur, MvField &cur, MvField &tar, Distortion &cost, int bcwIdx )
{
  vtmhip_smvd_job &j = g_rp.s->smvd[0];
  if( g_rp.record )
  {
    j.trace[1].cost = cost;      // as the member receives it: costStart - mvpCost (the golden test adds the index-pair rate of the device's pair)
    j.trace[1].mv[0] = cur.mv.hor; j.trace[1].mv[1] = cur.mv.ver;
    j.predSym[0][0] = predCur.hor; j.predSym[0][1] = predCur.ver; j.predSym[1][0] = predTar.hor; j.predSym[1][1] = predTar.ver;
    g_rp.inMember = true;
    vtmref_orig_xSymmetricMotionEstimation( is, pu, origBuf, predCur, predTar, eCur, cur, tar, cost, bcwIdx );
    g_rp.inMember = false;
    j.trace[2].cost = cost; j.trace[2].mv[0] = cur.mv.hor; j.trace[2].mv[1] = cur.mv.ver;
    g_rp.smvdCostCalls = -1;      // marks: the SMVD block ran
    return;
  }
  if( !g_rp.hasSmvd ) { vtmref_orig_xSymmetricMotionEstimation( is, pu, origBuf, predCur, predTar, eCur, cur, tar, cost, bcwIdx ); return; }
  // arguments: the state after the start-vector loop
  const int      i0 = j.trace[1].idx[0], i1 = j.trace[1].idx[1];
  const Distortion mvpCost = is->m_pcRdCost->getCost( j.mvpIdxBits[i0] + j.mvpIdxBits[i1] );
  const bool argsOk = cur.mv.hor == j.trace[1].mv[0] && cur.mv.ver == j.trace[1].mv[1] && cost == j.trace[1].cost - mvpCost && predCur.hor == j.cand[0][i0][0] && predCur.ver == j.cand[0][i0][1]
                   && predTar.hor == j.cand[1][i1][0] && predTar.ver == j.cand[1][i1][1];
  if( !argsOk ) pisNote( 5, 4, cur.mv.hor - j.trace[1].mv[0], cur.mv.ver - j.trace[1].mv[1], ( long long ) cost, ( long long ) ( j.trace[1].cost - mvpCost ) );
  if( !g_rp.replace || !argsOk )
  {
    g_rp.inMember = true;
    vtmref_orig_xSymmetricMotionEstimation( is, pu, origBuf, predCur, predTar, eCur, cur, tar, cost, bcwIdx );
    g_rp.inMember = false;
    if( !argsOk ) { ST_INC( pisReplayFallback ); g_rp.hasSmvd = false; return; }      // the rest of the block runs the reference's code
    if( cur.mv.hor != j.trace[2].mv[0] || cur.mv.ver != j.trace[2].mv[1] || cost != j.trace[2].cost ) pisNote( 4, 2, cur.mv.hor - j.trace[2].mv[0], cur.mv.ver - j.trace[2].mv[1], ( long long ) cost, ( long long ) j.trace[2].cost );
    return;
  }
  cur.mv.set( j.trace[2].mv[0], j.trace[2].mv[1] );
  tar.mv = cur.mv.getSymmvdMv( predCur, predTar );
  cost = j.trace[2].cost;
}

// ---- gather + device + replay --------------------------------------------------------------------------------------------------------------------
void pisHook( InterSearch *is, CodingUnit &cu, Partitioner &partitioner )
{
  ST_INC( pisCalls );
  const uint64_t t0 = nowNs();
  PredictionUnit &pu    = *cu.firstPU;
  const Slice    &slice = *cu.cs->slice;
  const SPS      &sps   = *cu.cs->sps;
  const int       w = pu.Y().width, h = pu.Y().height, imv = cu.imv;
  const bool      isB = slice.isInterB();
  const int       numRef[2] = { slice.getNumRefIdx( REF_PIC_LIST_0 ), isB ? slice.getNumRefIdx( REF_PIC_LIST_1 ) : 0 };
  // does the translational part run at all (:2297-2310)?
  bool checkAffine    = ( imv == 0 || sps.getAffineAmvrEnabledFlag() ) && imv != IMV_HPEL;
  bool checkNonAffine = imv == 0 || imv == IMV_HPEL || ( sps.getAMVREnabledFlag() && imv <= ( sps.getAMVREnabledFlag() ? IMV_4PEL : 0 ) );
  CodingUnit *bestCU  = cu.cs->bestCS != nullptr ? cu.cs->bestCS->getCU( CHANNEL_TYPE_LUMA ) : nullptr;
  const bool trySmvd  = ( bestCU != nullptr && imv == 2 && checkAffine ) ? ( !bestCU->firstPU->mergeFlag && !bestCU->affine ) : true;
  if( imv == 2 && checkNonAffine && sps.getAffineAmvrEnabledFlag() ) checkNonAffine = is->m_affineMotion.hevcCost[1] < is->m_affineMotion.hevcCost[0] * 1.06f;
  const auto fsm = is->m_pcEncCfg->getFastInterSearchMode();
  const uint8_t bcwIdx = isB ? cu.BcwIdx : BCW_DEFAULT;      // (:2286)
  const bool    useBcw = sps.getUseBcw();
  bool unsupported = slice.getPPS()->getUseWP() || slice.getPPS()->getWPBiPred() || is->m_pcEncCfg->getMCTSEncConstraint()
                  || is->m_useCompositeRef || is->m_pcEncCfg->getUseHashME() || is->m_pcEncCfg->getClipForBiPredMeEnabled()
                  || ( fsm != FASTINTERSEARCH_MODE1 && fsm != FASTINTERSEARCH_MODE2 )
                  || ( is->m_motionEstimationSearchMethod != MESEARCH_DIAMOND && is->m_motionEstimationSearchMethod != MESEARCH_DIAMOND_ENHANCED )
                  || w > 128 || h > 128 || w < 4 || h < 4 || w * h < 32 || numRef[0] < 1 || numRef[0] > PIS_MAX_REF || numRef[1] > PIS_MAX_REF || ( isB && numRef[0] != numRef[1] )
                  || sps.getBitDepth( CHANNEL_TYPE_LUMA ) > 10 || cu.firstPU->next != nullptr || is->m_uniMvListSize > 15;
  for( int l = 0; l < 2 && !unsupported; l++ )
    for( int r = 0; r < numRef[l]; r++ )
    {
      const Picture *rp = slice.getRefPic( RefPicList( l ), r );
      unsupported = unsupported || rp->isWrapAroundEnabled( cu.cs->pps ) || rp->isRefScaled( cu.cs->pps );
    }
  if( !checkNonAffine ) { ST_INC( pisSkipped ); vtmref_orig_predInterSearch( is, cu, partitioner ); ST_ADD( pisNs[3], nowNs() - t0 ); return; }
  if( unsupported ) ST_INC( pisUnsupported );
  if( unsupported && getenv( "VTMREF_PIS_WHY" ) )
    fprintf( stderr, "PISWHY bcw%d wp%d hash%d clip%d mvdl1z%d fsm%d me%d w%d h%d nr%d,%d bd%d next%d imv%d\n", bcwIdx != BCW_DEFAULT, slice.getPPS()->getUseWP(),
             is->m_pcEncCfg->getUseHashME(), is->m_pcEncCfg->getClipForBiPredMeEnabled(), cu.cs->picHeader->getMvdL1ZeroFlag(), ( int ) fsm, ( int ) is->m_motionEstimationSearchMethod, w, h,
             numRef[0], numRef[1], sps.getBitDepth( CHANNEL_TYPE_LUMA ), cu.firstPU->next != nullptr, imv );
  const bool dumping = g_pisDump != nullptr;
  const bool dumpOwn = dumping && bcwIdx != BCW_DEFAULT && g_pisDumpBcwStride != 0;
  if( unsupported || ( !dumping && !hookSampled( g_pisCtr ) ) || ( dumping && ( dumpOwn ? ( g_pisDumpBcwCtr++ % g_pisDumpBcwStride ) != 0 : ( g_pisDumpCtr++ % g_pisDumpStride ) != 0 ) ) )
  {
    const uint64_t t1 = nowNs();
    vtmref_orig_predInterSearch( is, cu, partitioner );
    ST_ADD( pisNs[3], nowNs() - t1 );
    return;
  }

  // ---- gather ----
  static PisSlots recIn, recOut;
  HookThread *T = dumping ? nullptr : hookThread();
  if( !dumping && !T ) { note_error(); vtmref_orig_predInterSearch( is, cu, partitioner ); return; }
  PisSlots &S = dumping ? recIn : *( PisSlots * ) T->pisHost;
  memset( &S, 0, offsetof( PisSlots, org ) );
  PisHeader hd; memset( &hd, 0, sizeof( hd ) );
  is->m_pcRdCost->selectMotionLambda();      // (:2357; idempotent)
  PU::spanMotionInfo( pu );                  // (:2330; the member does it again)
  const double   lambda = is->m_pcRdCost->m_motionLambda;
  const Position pos = cu.lumaPos();
  // xGetBlkBits (:3158-3163); with BCW enabled every bi cost carries the bits of the weight index (:2594, 2622, 2780: getWeightIdxBits( bcwIdx )) -- of the default weight here,
  // other weights are unsupported above -- which is one more constant on the bi mode's bits
  const uint32_t mbBits[3] = { isB ? 3u : 1u, 3u, 5u };      // (the weight-index bits travel per PU: vtmhip_pis_pu_in::bcwIdxBits)
  const bool     fdm = is->m_pcEncCfg->getFastMEForGenBLowDelayEnabled();
  // no bi stage (and no SMVD block, which sits inside it): 8x4 / 4x8 PUs, and -- BcwFast -- a non-default weight after an affine winner of the default pass (:2460-2464)
  const bool     bcwNoBi = !( slice.getCheckLDC() || bcwIdx == BCW_DEFAULT || !is->m_affineModeSelected || !is->m_pcEncCfg->getUseBcwFast() );
  const bool     biRestricted = PU::isBipredRestriction( pu ) || bcwNoBi;
  const bool     hasSmvd = isB && !biRestricted && slice.getBiDirPred() && trySmvd;
  const CPelBuf  org = cu.cs->getOrgBuf( pu ).Y();
  for( int y = 0; y < h; y++ ) memcpy( S.org + size_t( y ) * w, org.buf + ptrdiff_t( y ) * org.stride, sizeof( Pel ) * w );
  auto blkCache = dynamic_cast<CacheBlkInfoCtrl *>( is->m_modeCtrl );
  const int16_t *devBase = nullptr;
  int  refStride = 0;
  bool ok = true;
  g_rp = PisReplay();
  g_rp.numRef[0] = numRef[0]; g_rp.numRef[1] = numRef[1]; g_rp.w = w; g_rp.h = h;
  vtmhip_pis_level_run R; memset( &R, 0, sizeof( R ) );
  vtmhip_pis_level &L = R.pis;
  for( int l = 0; l < 2 && ok; l++ )
    for( int r = 0; r < numRef[l] && ok; r++ )
    {
      const int      row = g_rp.row( l, r );
      const Picture *refPic = slice.getRefPic( RefPicList( l ), r );
      const CPelBuf  ry = refPic->getRecoBuf( COMPONENT_Y );
      const int      m = refPic->margin;
      const int64_t  inPlane = ( int64_t ) ( m + pu.Y().y ) * ry.stride + m + pu.Y().x;
      vtmhip_me_job &j = S.uniJobs[row];
      if( dumping ) { hd.rowPlane[row] = pisDumpPlane( refPic ); hd.rowOff[row] = inPlane; j.refOff = inPlane; }
      else
      {
        const RefPlaneRef rp = refPlaneOf( T->ctx, refPic );
        if( !rp ) { ok = false; break; }
        if( !devBase ) devBase = rp->dev;
        L.refPlaneOff[l][r] = ( int64_t ) ( rp->dev - devBase ) + ( int64_t ) m * ry.stride + m;      // the plane's sample (0, 0)
        j.refOff = L.refPlaneOff[l][r] + ( int64_t ) pu.Y().y * ry.stride + pu.Y().x;
      }
      refStride = ry.stride;
      j.orgOff = 0; j.orgStride = w; j.refStride = ry.stride;
      j.puX = ( int16_t ) pos.x; j.puY = ( int16_t ) pos.y; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
      j.bi = 0; j.imv = ( uint8_t ) imv;
      AMVPInfo &info = g_rp.amvp[l][r];
      PU::fillMvpCand( pu, RefPicList( l ), r, info );
      j.numAmvpCand = ( uint8_t ) info.numCand;
      for( int i = 0; i < 2; i++ )
      {
        j.amvpCand[i][0] = i < info.numCand ? info.mvCand[i].hor : 0; j.amvpCand[i][1] = i < info.numCand ? info.mvCand[i].ver : 0;
        j.mvpIdxBits[i] = is->m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS];
      }
      j.bits = mbBits[l] + ( numRef[l] > 1 ? ( uint32_t ) ( r + 1 - ( r == numRef[l] - 1 ) ) : 0u );      // (:2368-2376)
      j.searchRange = is->m_aaiAdaptSR[l][r]; j.motionLambda = lambda;
      j.numExtraStart = is->m_uniMvListSize;
      for( int i = 0; i < is->m_uniMvListSize; i++ )
      {
        const BlkUniMvInfo *e = is->m_uniMvList + ( ( is->m_uniMvListIdx - 1 - i + is->m_uniMvListMaxSize ) % is->m_uniMvListMaxSize );
        j.extraStart[i][0] = e->uniMvs[l][r].hor; j.extraStart[i][1] = e->uniMvs[l][r].ver;
      }
      Mv cachedMv;
      if( blkCache && blkCache->getMv( pu, RefPicList( l ), r, cachedMv ) )      // (:3360-3368)
      {
        cachedMv.changePrecision( MV_PRECISION_INT, MV_PRECISION_INTERNAL );
        j.flags |= VTMHIP_MEJ_CACHED_INT_MV; j.mvHor = cachedMv.hor; j.mvVer = cachedMv.ver;
        g_rp.cached[row] = true;
      }
      hd.rowCached[row] = g_rp.cached[row];
      const int from0 = l == 1 ? slice.getList1IdxToList0Idx( r ) : -1;
      if( l == 1 ) L.list1FromList0[r] = from0 >= 0 ? from0 + 1 : 0;
      hd.rowCalls[row] = !( l == 1 && fdm && from0 >= 0 );
      L.refPoc[l][r] = hd.refPoc[l][r] = refPic->getPOC();
      // xReadBufferedUniMv (:3301-3304, 7677-7697): under a CU-level weight other than the default the row is the default-weight pass's vector and distortion (m_uniMotions)
      if( useBcw && cu.BcwIdx != BCW_DEFAULT && hd.rowCalls[row] && is->m_uniMotions.isReadMode( ( uint32_t ) l, ( uint32_t ) r ) )
      {
        Mv mv; Distortion dist = 0;
        is->m_uniMotions.copyTo( mv, dist, ( uint32_t ) l, ( uint32_t ) r );
        j.flags |= VTMHIP_MEJ_GIVEN_UNI;
        S.uniOut[row].mvHor = mv.hor; S.uniOut[row].mvVer = mv.ver; S.uniOut[row].cost = dist;
        g_rp.given[row] = true; g_rp.cached[row] = false;      // (the member returns before it looks at the block-vector cache)
        j.flags &= ~VTMHIP_MEJ_CACHED_INT_MV; hd.rowCached[row] = 0;
        L.givenRows |= 1 << row;
      }
      CHECK( ry.stride != refStride, "reference planes of one picture size share a stride" );
    }
  if( !ok ) { note_error(); vtmref_orig_predInterSearch( is, cu, partitioner ); return; }
  // m_uniMvList after insertUniMvCands (:2451-2459)
  vtmhip_pis_pu_in &pin = S.puIn[0];
  pin.noSmvd = !trySmvd;
  pin.uniMvInsert = imv == 0 && ( !useBcw || bcwIdx == BCW_DEFAULT );      // (:2451)
  pin.bcwWeightL1 = bcwIdx != BCW_DEFAULT ? g_BcwWeights[bcwIdx] : 0;
  pin.bcwIdxBits = ( uint8_t ) ( isB && useBcw ? is->getWeightIdxBits( bcwIdx ) : 0 );
  pin.bcwFastSkipPoc = is->m_pcEncCfg->getUseBcwFast() && slice.getTLayer() > 1;
  if( pin.uniMvInsert )
  {
    int k = 0;
    for( ; k < is->m_uniMvListSize; k++ )
    {
      const BlkUniMvInfo *e = is->m_uniMvList + ( ( is->m_uniMvListIdx - 1 - k + is->m_uniMvListMaxSize ) % is->m_uniMvListMaxSize );
      if( e->x == pu.Y().x && e->y == pu.Y().y && e->w == w && e->h == h ) break;
    }
    pin.uniMvSelfIsNew = k == is->m_uniMvListSize; pin.uniMvSelfPos = pin.uniMvSelfIsNew ? 0 : k;
  }
  L.numPU = 1; L.numRef[0] = numRef[0]; L.numRef[1] = numRef[1]; L.smvdBit = isB && slice.getBiDirPred(); L.refStride = refStride;
  for( int i = 0; i < 3; i++ ) L.mbBits[i] = mbBits[i];
  L.candsGiven = 1; L.biRestricted = biRestricted; L.mvdL1Zero = isB && cu.cs->picHeader->getMvdL1ZeroFlag(); L.fastMEForGenBLowDelay = fdm;
  L.picW = cu.cs->pps->getPicWidthInLumaSamples(); L.picH = cu.cs->pps->getPicHeightInLumaSamples(); L.ctuSize = sps.getMaxCUWidth();
  L.curPoc = hd.curPoc = slice.getPOC(); hd.givenRows = L.givenRows; hd.bcwIdx = bcwIdx; hd.bcwNoBi = bcwNoBi;
  if( hasSmvd ) { L.symRefIdx[0] = slice.getSymRefIdx( 0 ); L.symRefIdx[1] = slice.getSymRefIdx( 1 ); }
  S.pos[0] = ( int64_t ) pu.Y().y * refStride + pu.Y().x;
  vtmhip_pred_job &po = S.predOther[0];
  po.orgOff = 0; po.orgStride = w; po.refStride[0] = po.refStride[1] = refStride; po.predOff = po.outOff = 0; po.predStride = po.outStride = w;
  po.width = ( int16_t ) w; po.height = ( int16_t ) h; po.epilogue = 2; po.bitDepth = ( uint8_t ) sps.getBitDepth( CHANNEL_TYPE_LUMA ); po.useAltHpelIf = imv == IMV_HPEL;
  for( int r = 0; r < numRef[0] && isB; r++ )
  {
    vtmhip_me_job &b = S.biJobs[r];
    b.orgOff = 0; b.orgStride = w; b.otherPredOff = 0; b.otherPredStride = w; b.puX = ( int16_t ) pos.x; b.puY = ( int16_t ) pos.y; b.width = ( int16_t ) w; b.height = ( int16_t ) h;
  }
  vtmhip_me_cfg cfg; memset( &cfg, 0, sizeof( cfg ) );
  cfg.bipredSearchRange = is->m_bipredSearchRange;
  cfg.useHadME = is->m_pcEncCfg->getUseHADME() && !cu.cs->slice->getDisableSATDForRD();
  cfg.fastInterSearchMode13 = fsm == FASTINTERSEARCH_MODE1 || fsm == FASTINTERSEARCH_MODE3;
  cfg.extendedSettings = is->m_motionEstimationSearchMethod == MESEARCH_DIAMOND_ENHANCED;
  cfg.firstSearchStop = is->m_pcEncCfg->getFastMEAssumingSmootherMVEnabled();
  cfg.uniformImv = imv;
  hd.magic = PIS_MAGIC; hd.poc = slice.getPOC(); hd.x = pu.Y().x; hd.y = pu.Y().y; hd.w = w; hd.h = h; hd.imv = imv;
  hd.picW = cu.cs->pps->getPicWidthInLumaSamples(); hd.picH = cu.cs->pps->getPicHeightInLumaSamples(); hd.ctuSize = sps.getMaxCUWidth(); hd.bitDepth = sps.getBitDepth( CHANNEL_TYPE_LUMA );
  hd.numRef[0] = numRef[0]; hd.numRef[1] = numRef[1]; hd.smvdBit = L.smvdBit; hd.symRefIdx[0] = L.symRefIdx[0]; hd.symRefIdx[1] = L.symRefIdx[1]; hd.hasSmvd = hasSmvd; hd.biRestricted = biRestricted; hd.mvdL1Zero = L.mvdL1Zero; hd.fdm = fdm;
  for( int r = 0; r < PIS_MAX_REF; r++ ) hd.list1FromList0[r] = L.list1FromList0[r];
  for( int i = 0; i < 3; i++ ) hd.mbBits[i] = mbBits[i];
  hd.bipredSearchRange = cfg.bipredSearchRange; hd.useHadME = cfg.useHadME; hd.fen13 = cfg.fastInterSearchMode13; hd.extendedSettings = cfg.extendedSettings;
  hd.firstSearchStop = cfg.firstSearchStop; hd.uniMvListSize = is->m_uniMvListSize;
  const uint64_t t1 = nowNs();
  ST_ADD( pisNs[0], t1 - t0 );

  // ---- device ----
  static const bool trace = getenv( "VTMREF_PIS_TRACE" ) != nullptr;      // one line per device call BEFORE it is issued: the last line names the PU of a faulting launch
  if( trace ) { fprintf( stderr, "PIS poc %d pu %d,%d %dx%d imv %d refs %d+%d list %d smvd %d bir %d l1from0 %d,%d cached %d%d%d%d\n", hd.poc, hd.x, hd.y, w, h, imv, numRef[0], numRef[1],
                         is->m_uniMvListSize, hasSmvd, biRestricted, L.list1FromList0[0], L.list1FromList0[1], g_rp.cached[0], g_rp.cached[1], g_rp.cached[2], g_rp.cached[3] ); fflush( stderr ); }
  if( !dumping )
  {
    char *d = T->d_pis;
    auto  dp = [&]( size_t off ) { return ( void * ) ( d + off ); };
    L.uniJobs = ( vtmhip_me_job * ) dp( offsetof( PisSlots, uniJobs ) ); L.uniOut = ( const vtmhip_me_out * ) dp( offsetof( PisSlots, uniOut ) );
    L.uniRows = ( vtmhip_pis_row * ) dp( offsetof( PisSlots, uniRows ) ); L.distBiP = ( uint64_t * ) dp( offsetof( PisSlots, distBiP ) );
    L.pus = ( vtmhip_pis_pu * ) dp( offsetof( PisSlots, pus ) ); L.puIn = ( const vtmhip_pis_pu_in * ) dp( offsetof( PisSlots, puIn ) );
    L.predOther = ( vtmhip_pred_job * ) dp( offsetof( PisSlots, predOther ) ); L.biJobs = ( vtmhip_me_job * ) dp( offsetof( PisSlots, biJobs ) );
    L.biOut = ( const vtmhip_me_out * ) dp( offsetof( PisSlots, biOut ) ); L.biRows = ( vtmhip_pis_row * ) dp( offsetof( PisSlots, biRows ) );
    L.smvdJobs = hasSmvd ? ( vtmhip_smvd_job * ) dp( offsetof( PisSlots, smvd ) ) : nullptr;
    L.pos = ( const int64_t * ) dp( offsetof( PisSlots, pos ) );
    R.uniOut = ( vtmhip_me_out * ) dp( offsetof( PisSlots, uniOut ) ); R.biOut = ( vtmhip_me_out * ) dp( offsetof( PisSlots, biOut ) );
    R.width = w; R.height = h;
    const int big = std::max( w, h );
    R.pic.picW = hd.picW; R.pic.picH = hd.picH; R.pic.ctuSize = hd.ctuSize; R.pic.bitDepth = hd.bitDepth; R.pic.wavesPerJob = big >= 128 ? 8 : big >= 64 ? 2 : 1;
    R.picBi = R.pic; R.picBi.wavesPerJob = big >= 128 ? 16 : big >= 64 ? 8 : big >= 32 ? 4 : 1;
    const int uniformShape = g_apiUniformShape( w, h );
    R.cfgUni = cfg; R.cfgUni.uniformSquare = uniformShape; R.cfgUni.uniformBi = 1; R.cfgUni.noUniMvList = is->m_uniMvListSize == 0;
    R.cfgBi = cfg; R.cfgBi.uniformSquare = uniformShape; R.cfgBi.uniformBi = 2; R.cfgBi.noUniMvList = is->m_uniMvListSize == 0 && !pin.uniMvInsert; R.cfgBi.biPatternGiven = 1;
    vtmhip_pis_buffers B; memset( &B, 0, sizeof( B ) );
    B.org = ( const int16_t * ) dp( offsetof( PisSlots, org ) ); B.dpb = devBase; B.orgBi = T->d_pisOrgBi;
    ok = A.h2d( T->ctx, T->d_pis, &S, offsetof( PisSlots, org ) + size_t( w ) * h * 2 ) == VTMHIP_OK && g_apiPis( T->ctx, &R, &B ) == VTMHIP_OK
      && A.d2h( T->ctx, &S, T->d_pis, offsetof( PisSlots, org ) ) == VTMHIP_OK;
    if( !ok ) { note_error( T->ctx ); vtmref_orig_predInterSearch( is, cu, partitioner ); return; }
    ST_INC( pisDevice );
  }
  const uint64_t t2 = nowNs();
  ST_ADD( pisNs[1], t2 - t1 );

  // ---- the reference's own member over the tables ----
  if( dumping ) { recOut = recIn; g_rp.s = &recOut; g_rp.record = true; }
  else { g_rp.s = &S; g_rp.replace = g_pisReplace; g_rp.refineList = S.pus[0].refineList; }
  g_rp.hasSmvd = hasSmvd; g_rp.active = true;
  if( !dumping && ( !isB || biRestricted ) ) g_rp.refineList = -1;
  vtmref_orig_predInterSearch( is, cu, partitioner );
  g_rp.active = false;
  const uint64_t t3 = nowNs();
  ST_ADD( pisNs[2], t3 - t2 );

  // ---- what the member left behind ----
  PisFinal F; memset( &F, 0, sizeof( F ) );
  F.ran = g_rp.amvpServed > 0; F.affine = cu.affine; F.interDir = pu.interDir; F.smvdMode = cu.smvdMode; F.biList = g_rp.refineList;
  for( int l = 0; l < 2; l++ )
  {
    F.refIdx[l] = pu.refIdx[l]; F.mv[l][0] = pu.mv[l].hor; F.mv[l][1] = pu.mv[l].ver; F.mvd[l][0] = pu.mvd[l].hor; F.mvd[l][1] = pu.mvd[l].ver;
    F.mvpIdx[l] = pu.mvpIdx[l]; F.mvpNum[l] = pu.mvpNum[l]; F.refIdxBi[l] = cu.refIdxBi[l];
  }
  // (:3054-3057) the member stores its translational cost unless the weight is not the default one and both affine models are buffered
  F.hevcCost = ( bcwIdx == BCW_DEFAULT || !is->m_affineMotion.affine4ParaAvail || !is->m_affineMotion.affine6ParaAvail ) ? is->m_affineMotion.hevcCost[imv] : PIS_COST_UNKNOWN;
  F.smvdRan = dumping ? g_rp.smvdCostCalls == -1 : hasSmvd;
  if( dumping )
  {
    hd.bytes = ( uint32_t ) ( sizeof( hd ) + 2 * offsetof( PisSlots, org ) + size_t( w ) * h * 2 + sizeof( F ) );
    fwrite( &hd, sizeof( hd ), 1, g_pisDump ); fwrite( &recIn, offsetof( PisSlots, org ), 1, g_pisDump ); fwrite( &recOut, offsetof( PisSlots, org ), 1, g_pisDump );
    fwrite( recIn.org, 2, size_t( w ) * h, g_pisDump ); fwrite( &F, sizeof( F ), 1, g_pisDump );
    ST_INC( pisDevice );
    return;
  }
  if( !F.ran ) { ST_INC( pisSkipped ); return; }
  // the device's own decision record against the member's result
  const vtmhip_pis_pu &P = S.pus[0];
  const bool bi = P.interDir == 3;
  const uint64_t devCost = bi ? P.costBi : P.cost[P.interDir == 2 ? 1 : 0];
  bool bad = F.hevcCost != PIS_COST_UNKNOWN && devCost != F.hevcCost;
  if( !cu.affine )
  {
    bad = bad || P.interDir != F.interDir || ( bi && ( P.smvdMode != 0 ) != ( F.smvdMode != 0 ) );
    for( int l = 0; l < 2 && !bad; l++ )
    {
      if( !( P.interDir & ( 1 << l ) ) ) continue;
      const int ref = bi ? P.refIdxBi[l] : P.refIdx[l];
      int mvH = bi ? P.mvBi[l][0] : P.mv[l][0], mvV = bi ? P.mvBi[l][1] : P.mv[l][1], predH, predV, idx;
      if( bi && P.smvdMode ) { idx = S.smvd[0].mvpIdxSym[l]; predH = S.smvd[0].predSym[l][0]; predV = S.smvd[0].predSym[l][1]; }
      else if( bi && l == P.refineList ) { const vtmhip_pis_row &r = S.biRows[ref]; idx = r.mvpIdx; predH = r.mvPredHor; predV = r.mvPredVer; }
      else if( bi && L.mvdL1Zero ) { idx = P.mvpIdxL1Zero; predH = mvH; predV = mvV; }      // list 1 at its predictor (:2477-2487)
      else { const vtmhip_pis_row &r = S.uniRows[g_rp.row( l, ref )]; idx = r.mvpIdx; predH = r.mvPredHor; predV = r.mvPredVer; }
      bad = ref != F.refIdx[l] || mvH != F.mv[l][0] || mvV != F.mv[l][1] || mvH - predH != F.mvd[l][0] || mvV - predV != F.mvd[l][1] || idx != F.mvpIdx[l];
    }
  }
  if( bad ) pisNote( 0, w * 1000 + h, imv * 10 + P.interDir, F.interDir * 10 + cu.affine, ( long long ) F.hevcCost, ( long long ) devCost );
  ST_ADD( pisNs[0], nowNs() - t3 );
}
