// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// ref_shim_me.cpp: drives the REAL reference motion search (VTM 9.3 EncoderLib/InterSearch.cpp, compiled in
// place by oracle/Makefile.ref) on caller-supplied sample planes, so that the C restatement in
// oracle/vtm_oracle.c (vo_tz_search / vo_full_search / vo_frac_search) and the HIP search kernels can be
// pinned against the reference's own decisions:
//   InterSearch::xTZSearch             EncoderLib/InterSearch.cpp:3640-3976
//   InterSearch::xSetSearchRange       :3496-3563
//   InterSearch::xPatternSearch        :3566-3608
//   InterSearch::xPatternSearchFracDIF :4284-4339 (+ xExtDIFUpSamplingH/Q :5840-6051, xPatternRefinement :707-761)
// Nothing here is reference source: it builds the minimum object graph those members read
// (SPS/PPS picture size, CodingUnit/PredictionUnit position, EncCfg switches, RdCost lambda/predictor)
// and calls them (compiled with -fno-access-control: they are protected members).
#include "CommonLib/CommonDef.h"
#include "CommonLib/Unit.h"
#include "CommonLib/Slice.h"
#include "CommonLib/CodingStructure.h"
#include "CommonLib/RdCost.h"
#include "CommonLib/Mv.h"
#include "EncoderLib/EncCfg.h"
#include "EncoderLib/InterSearch.h"

#include <cstdint>
#include <cstdlib>
#include <new>
#include <utility>
#include <cstring>

namespace
{
// every rig object starts from zeroed storage: the reference's constructors leave members the encoder normally sets elsewhere (and these rigs set only where the
// member under test reads them) indeterminate, which made a result depend on what the heap block held before
template<class T, class... A> T *zeroNew( A &&... a ) { void *m = calloc( 1, sizeof( T ) ); return new( m ) T( std::forward<A>( a )... ); }

struct MeRig
{
  SPS             sps;
  PPS             pps;
  Slice           slice;
  CUCache         cuCache;
  PUCache         puCache;
  TUCache         tuCache;
  CodingStructure cs;
  CodingUnit      cu;
  PredictionUnit  pu;
  EncCfg          cfg;
  RdCost          rd;
  InterSearch     is;
  BlkUniMvInfo    uniMv[16];

  MeRig() : cs( cuCache, puCache, tuCache )
  {
    clipMv = clipMvInPic;   // what EncLib installs when there are no sub-pictures (EncLib.cpp)
    cs.sps   = &sps;
    cs.pps   = &pps;
    cs.slice = &slice;
    cu.cs    = &cs;
    cu.slice = &slice;
    pu.cs    = &cs;
    pu.cu    = &cu;
    cu.imv   = 0;
    pps.m_numSubPics = 1;
    pps.m_subPics.resize( 1 );
    pps.m_subPics[0].setTreatedAsPicFlag( false );
    pps.setWrapAroundEnabledFlag( false );
    sps.setWrapAroundEnabledFlag( false );
    cfg.setMCTSEncConstraint( false );
    cfg.setUseHashME( false );
    cfg.setUseHADME( true );
    cfg.setFastMEAssumingSmootherMVEnabled( true );
    cfg.setClipForBiPredMeEnabled( false );
    slice.setDisableSATDForRD( false );
    is.m_pcEncCfg      = &cfg;
    is.m_pcRdCost      = &rd;
    is.m_useCompositeRef = false;
    is.m_skipFracME    = false;
    is.m_currChromaFormat = CHROMA_420;
    is.m_uniMvList        = uniMv;
    is.m_uniMvListMaxSize = 15;
    is.m_uniMvListSize    = 0;
    is.m_uniMvListIdx     = 0;
    is.m_currRefPicList   = REF_PIC_LIST_0;
    is.m_currRefPicIndex  = 0;
    is.m_if.initInterpolationFilter( true );
    // m_filteredBlock / m_filteredBlockTmp as InterPrediction::init allocates them (InterPrediction.cpp:150-176)
    const int extW = MAX_CU_SIZE + 16, extH = MAX_CU_SIZE + 1 + 16;
    for( int i = 0; i < 4; i++ )
    {
      is.m_filteredBlockTmp[i][0] = ( Pel * ) xMalloc( Pel, ( extW + 4 ) * ( extH + 7 + 4 ) );
      for( int j = 0; j < 4; j++ ) is.m_filteredBlock[i][j][0] = ( Pel * ) xMalloc( Pel, extW * extH );
    }
  }
};

MeRig *g_rig = nullptr;

void ensureRom()
{
  static bool romInit = false;
  if( !romInit ) { initROM(); romInit = true; }
}

struct MeCtxC   // must match vo_me_ctx_t (oracle/vtm_oracle.h)
{
  const int16_t *org; int orgStride; const int16_t *ref; int refStride; int w, h, subShift, bitDepth; unsigned imvShift;
  struct { double motionLambda; int predHor, predVer, costScale; } mv;   // vo_mvcost_t (nested: keeps its tail padding)
  int picW, picH, puX, puY, ctuSize;
};
struct TzJobC   // vo_tz_job_t
{
  int mvHor, mvVer, searchRange, extendedSettings, fastSettings, firstSearchStop, hasInt, intHor, intVer, numExtra; int extra[16][2];
};
struct MeResC { int mvX, mvY; uint64_t cost, dist, nEval; };
struct FracResC { int halfX, halfY, qterX, qterY; uint64_t costHalf, cost, candHalf[9], candQuarter[9]; };

void setup( MeRig &r, const MeCtxC &c, InterSearch::IntTZSearchStruct &st, CPelBuf &pattern )
{
  r.sps.setMaxCUWidth( c.ctuSize );
  r.sps.setMaxCUHeight( c.ctuSize );
  r.pps.setPicWidthInLumaSamples( c.picW );
  r.pps.setPicHeightInLumaSamples( c.picH );
  const UnitArea ua( CHROMA_420, Area( c.puX, c.puY, c.w, c.h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  r.rd.m_motionLambda = c.mv.motionLambda;
  r.rd.setPredictor( Mv( c.mv.predHor, c.mv.predVer ) );
  r.rd.setCostScale( c.mv.costScale );
  r.is.m_lumaClpRng.min = 0;
  r.is.m_lumaClpRng.max = ( 1 << c.bitDepth ) - 1;
  r.is.m_lumaClpRng.bd  = c.bitDepth;
  r.is.m_lumaClpRng.n   = 0;
  pattern               = CPelBuf( c.org, c.orgStride, c.w, c.h );
  st.pcPatternKey = &pattern;
  st.iRefStride   = c.refStride;
  st.piRefY       = c.ref;
  st.imvShift     = c.imvShift;
  st.useAltHpelIf = false;
  st.inCtuSearch  = false;
  st.zeroMV       = false;
  // subShiftMode: 2 reproduces the CTC FEN=1 rule; we pass the mode that yields c.subShift for this block
  st.subShiftMode = c.subShift ? ( ( c.h > 8 && c.w <= 64 ) ? 2 : 3 ) : 0;
}
}   // namespace

extern "C"
{

void ref_tz_search( const MeCtxC *c, const TzJobC *job, MeResC *res )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  InterSearch::IntTZSearchStruct st;
  CPelBuf pattern;
  setup( r, *c, st, pattern );
  r.cfg.setFastMEAssumingSmootherMVEnabled( job->firstSearchStop != 0 );
  r.is.m_iSearchRange = job->searchRange;
  r.is.m_uniMvListSize = job->numExtra;
  r.is.m_uniMvListIdx  = job->numExtra % 15;
  // candidate i is read from slot (idx - 1 - i) mod max (InterSearch.cpp:3728)
  for( int i = 0; i < job->numExtra; i++ )
  {
    const int slot = ( r.is.m_uniMvListIdx - 1 - i + 15 ) % 15;
    r.uniMv[slot].uniMvs[0][0] = Mv( job->extra[i][0], job->extra[i][1] );
  }
  Mv         rcMv( job->mvHor, job->mvVer );
  Mv         intPred( job->intHor, job->intVer );
  Distortion sad = 0;
  r.is.xTZSearch( r.pu, REF_PIC_LIST_0, 0, st, rcMv, sad, job->hasInt ? &intPred : nullptr, job->extendedSettings != 0, job->fastSettings != 0 );
  res->mvX   = rcMv.hor;
  res->mvY   = rcMv.ver;
  res->cost  = st.uiBestSad;
  res->dist  = sad;
  res->nEval = 0;
  r.is.m_uniMvListSize = 0;
}

void ref_set_search_range( const MeCtxC *c, int predHor, int predVer, int range, int out[4] )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  InterSearch::IntTZSearchStruct st;
  CPelBuf pattern;
  setup( r, *c, st, pattern );
  InterSearch::SearchRange sr;
  r.is.xSetSearchRange( r.pu, Mv( predHor, predVer ), range, sr, st );
  out[0] = sr.left; out[1] = sr.right; out[2] = sr.top; out[3] = sr.bottom;
}

void ref_full_search( const MeCtxC *c, const int range[4], MeResC *res )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  InterSearch::IntTZSearchStruct st;
  CPelBuf pattern;
  setup( r, *c, st, pattern );
  st.searchRange.left = range[0]; st.searchRange.right = range[1]; st.searchRange.top = range[2]; st.searchRange.bottom = range[3];
  Mv         mv;
  Distortion sad = 0;
  r.is.xPatternSearch( st, mv, sad );
  res->mvX = mv.hor; res->mvY = mv.ver; res->cost = st.uiBestSad; res->dist = sad; res->nEval = 0;
}

void ref_frac_search( const MeCtxC *c, int intX, int intY, int useHad, int useAltHpelIf, FracResC *res )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  InterSearch::IntTZSearchStruct st;
  CPelBuf pattern;
  setup( r, *c, st, pattern );
  st.useAltHpelIf = useAltHpelIf != 0;
  r.cfg.setUseHADME( useHad != 0 );
  Mv         mvInt( intX, intY ), mvHalf, mvQter;
  Distortion cost = 0;
  r.is.xPatternSearchFracDIF( r.pu, REF_PIC_LIST_0, 0, st, mvInt, mvHalf, mvQter, cost );
  res->halfX = mvHalf.hor; res->halfY = mvHalf.ver; res->qterX = mvQter.hor; res->qterY = mvQter.ver;
  res->cost  = cost; res->costHalf = 0;
  memset( res->candHalf, 0, sizeof( res->candHalf ) );
  memset( res->candQuarter, 0, sizeof( res->candQuarter ) );
  r.cfg.setUseHADME( true );
}

}   // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Scalar quantisation through the REAL Quant::quant / Quant::dequant (CommonLib/Quant.cpp:955-1038, 357-482): a TransformUnit
// with just the fields those members read (block size, mtsIdx, cu->qp / predMode, SPS bit depth, slice NAL type for
// isIRAP, no sign-bit hiding, no explicit scaling lists).
// ------------------------------------------------------------------------------------------------------------------
#include "CommonLib/Quant.h"
#include "CommonLib/Rom.h"
#include "CommonLib/Contexts.h"

extern "C" int ref_quant_dequant2( const int32_t *coef, int w, int h, int bitDepth, int qp, int isIRAP, int mtsIdx, int32_t *qcoef, int32_t *absSum, int32_t *dqcoef );
extern "C" int ref_quant_dequant( const int32_t *coef, int w, int h, int bitDepth, int qp, int isIRAP, int32_t *qcoef, int32_t *absSum, int32_t *dqcoef )
{
  return ref_quant_dequant2( coef, w, h, bitDepth, qp, isIRAP, MTS_DCT2_DCT2, qcoef, absSum, dqcoef );
}
// mtsIdx = MTS_SKIP (1): the transform-skip forms of both members (useTransformSkip, Quant.cpp:966-997, 357-482)
extern "C" int ref_quant_dequant2( const int32_t *coef, int w, int h, int bitDepth, int qp, int isIRAP, int mtsIdx, int32_t *qcoef, int32_t *absSum, int32_t *dqcoef )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Quant *quant   = nullptr;
  ensureRom();
  if( !quant ) { quant = zeroNew<Quant>( nullptr ); quant->init( 64, false, false, false ); }
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, bitDepth );
  r.sps.setBitDepth( CHANNEL_TYPE_CHROMA, bitDepth );
  r.sps.setQpBDOffset( CHANNEL_TYPE_LUMA, 6 * ( bitDepth - 8 ) );
  r.sps.setQpBDOffset( CHANNEL_TYPE_CHROMA, 6 * ( bitDepth - 8 ) );
  r.sps.setInternalMinusInputBitDepth( CHANNEL_TYPE_LUMA, 0 );
  r.slice.setNalUnitType( isIRAP ? NAL_UNIT_CODED_SLICE_IDR_W_RADL : NAL_UNIT_CODED_SLICE_TRAIL );
  r.slice.setSignDataHidingEnabledFlag( false );
  r.slice.setExplicitScalingListUsed( false );
  r.slice.setSPS( &r.sps );
  static TransformUnit *tu = nullptr;
  if( !tu ) tu = zeroNew<TransformUnit>( UnitArea( CHROMA_400, Area( 0, 0, w, h ) ) );
  tu->UnitArea::operator=( UnitArea( CHROMA_400, Area( 0, 0, w, h ) ) );
  tu->cs = &r.cs;
  tu->cu = &r.cu;
  tu->chromaFormat = CHROMA_400;
  tu->mtsIdx[0]    = ( uint8_t ) mtsIdx;
  tu->noResidual   = false;
  r.cu.qp = qp; r.cu.predMode = MODE_INTER; r.cu.lfnstIdx = 0; r.cu.colorTransform = false; r.cu.bdpcmMode = 0; r.cu.bdpcmModeChroma = 0;
  r.cu.chromaQpAdj = 0; r.cu.treeType = TREE_D; r.cu.modeType = MODE_TYPE_ALL;
  static TCoeff levels[64 * 64];
  tu->m_coeffs[0] = levels;
  const QpParam cQP( *tu, COMPONENT_Y );
  CCoeffBuf src( coef, w, w, h );
  TCoeff    sum = 0;
  Ctx       ctx;
  quant->quant( *tu, COMPONENT_Y, src, sum, cQP, ctx );
  memcpy( qcoef, levels, sizeof( TCoeff ) * w * h );
  *absSum = sum;
  CoeffBuf dst( dqcoef, w, w, h );
  quant->dequant( *tu, dst, COMPONENT_Y, cQP );
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The whole InterSearch::xMotionEstimation (InterSearch.cpp:3299-3494, with xPatternSearchIntRefine :4172-4282 for the
// integer / 4-sample AMVR modes) on caller-supplied planes: a Picture whose reconstruction buffer aliases the caller's
// reference plane is installed as slice reference (list 0, index 0); the other list's prediction goes through
// m_tmpPredStorage[1] exactly as predInterSearch leaves it for the bi-pred iteration.
// ------------------------------------------------------------------------------------------------------------------
#include "CommonLib/Picture.h"
#include "vtm_oracle.h"

extern "C" void ref_motion_estimation( const vo_mest_cfg_t *cfg, const vo_mest_job_t *j, vo_mest_result_t *res )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pic = nullptr;
  static bool     storage = false;
  if( !pic ) pic = zeroNew<Picture>();
  if( !storage )
  {
    const UnitArea lcu( CHROMA_400, Area( 0, 0, MAX_CU_SIZE, MAX_CU_SIZE ) );
    if( r.is.m_tmpStorageLCU.bufs.empty() ) r.is.m_tmpStorageLCU.create( lcu );
    if( r.is.m_tmpPredStorage[0].bufs.empty() ) r.is.m_tmpPredStorage[0].create( lcu );
    if( r.is.m_tmpPredStorage[1].bufs.empty() ) r.is.m_tmpPredStorage[1].create( lcu );
    storage = true;
  }
  r.sps.setMaxCUWidth( j->ctuSize );
  r.sps.setMaxCUHeight( j->ctuSize );
  r.sps.setUseBcw( false );
  r.pps.setPicWidthInLumaSamples( j->picW );
  r.pps.setPicHeightInLumaSamples( j->picH );
  r.pps.setUseWP( false );
  r.pps.setWPBiPred( false );
  r.slice.setPPS( &r.pps );
  r.slice.setSPS( &r.sps );
  r.slice.setSliceType( B_SLICE );
  const UnitArea ua( CHROMA_400, Area( j->puX, j->puY, j->w, j->h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_400;
  r.pu.chromaFormat = CHROMA_400;
  r.cu.imv    = j->imv;
  r.cu.BcwIdx = BCW_DEFAULT;
  if( j->bi && j->bcwWeight != 0 && j->bcwWeight != 4 )      // the rig searches list 0: the CU-level index whose list-0 weight is the job's
    for( int i = 0; i < BCW_NUM; i++ ) if( getBcwWeight( ( uint8_t ) i, REF_PIC_LIST_0 ) == j->bcwWeight ) r.cu.BcwIdx = ( uint8_t ) i;
  r.rd.m_motionLambda = j->motionLambda;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << j->bitDepth ) - 1; clp.bd = j->bitDepth; clp.n = 0;
  r.slice.m_clpRngs.comp[COMPONENT_Y] = clp;
  // reference picture: reconstruction buffer = the caller's plane, picture origin at (0,0)
  Pel *origin = const_cast<Pel *>( j->ref ) - ( ptrdiff_t ) j->puY * j->refStride - j->puX;
  pic->chromaFormat = CHROMA_400;
  pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_400, PelBuf( origin, j->refStride, j->picW, j->picH ) ) );
  r.slice.m_apcRefPicList[REF_PIC_LIST_0][0] = pic;
  // encoder switches
  r.cfg.setUseHADME( cfg->useHadME != 0 );
  r.cfg.setFastInterSearchMode( cfg->fastInterSearchMode13 ? FASTINTERSEARCH_MODE1 : FASTINTERSEARCH_DISABLED );
  r.cfg.setRestrictMESampling( false );
  r.cfg.setMotionEstimationSearchMethod( cfg->extendedSettings ? MESEARCH_DIAMOND_ENHANCED : MESEARCH_DIAMOND );
  r.cfg.setFastMEAssumingSmootherMVEnabled( cfg->firstSearchStop != 0 );
  r.is.m_motionEstimationSearchMethod = cfg->extendedSettings ? MESEARCH_DIAMOND_ENHANCED : MESEARCH_DIAMOND;
  r.is.m_bipredSearchRange            = cfg->bipredSearchRange;
  r.is.m_aaiAdaptSR[0][0]             = j->searchRange;
  r.is.m_modeCtrl                     = nullptr;
  r.is.m_uniMvListSize = j->numExtraStart;
  r.is.m_uniMvListIdx  = j->numExtraStart % 15;
  for( int i = 0; i < j->numExtraStart; i++ )
  {
    const int slot = ( r.is.m_uniMvListIdx - 1 - i + 15 ) % 15;
    r.uniMv[slot].uniMvs[0][0] = Mv( j->extraStart[i][0], j->extraStart[i][1] );
  }
  for( int i = 0; i < 2; i++ ) r.is.m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] = j->mvpIdxBits[i];
  AMVPInfo amvp;
  amvp.numCand = j->numAmvpCand;
  for( int i = 0; i < 2; i++ ) amvp.mvCand[i] = Mv( j->amvpCand[i][0], j->amvpCand[i][1] );
  if( j->bi )
  {
    PelBuf other = r.is.m_tmpPredStorage[1].getBuf( UnitAreaRelative( r.cu, r.pu ) ).Y();
    for( int y = 0; y < j->h; y++ ) memcpy( other.buf + ( ptrdiff_t ) y * other.stride, j->otherPred + ( ptrdiff_t ) y * j->otherStride, sizeof( Pel ) * j->w );
  }
  PelUnitBuf origBuf( CHROMA_400, PelBuf( const_cast<Pel *>( j->org ), j->orgStride, j->w, j->h ) );
  Mv         mvPred( j->mvPredHor, j->mvPredVer ), mv( j->mvHor, j->mvVer );
  int        mvpIdx = j->mvpIdx;
  uint32_t   bits   = j->bits;
  Distortion cost   = std::numeric_limits<Distortion>::max();
  r.is.xMotionEstimation( r.pu, origBuf, REF_PIC_LIST_0, mvPred, 0, mv, mvpIdx, bits, cost, amvp, j->bi != 0 );
  res->mvHor = mv.hor; res->mvVer = mv.ver; res->mvPredHor = mvPred.hor; res->mvPredVer = mvPred.ver; res->mvpIdx = mvpIdx;
  res->bits = bits; res->cost = cost;
  res->intX = r.is.m_integerMv2Nx2N[0][0].hor; res->intY = r.is.m_integerMv2Nx2N[0][0].ver; res->intDist = 0;
  // leave the rig as the other shims expect it
  r.is.m_uniMvListSize = 0;
  r.cu.imv = 0;
  r.cfg.setUseHADME( true );
  r.cfg.setFastMEAssumingSmootherMVEnabled( true );
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
}

// ------------------------------------------------------------------------------------------------------------------
// InterPrediction::xPredInterBlk (CommonLib/InterPrediction.cpp:660-815) for one component of one PU: the three planes of a
// 4:2:0 reference picture alias the caller's buffers (plane origin pointers), the prediction goes to the caller's block.
// comp 0 luma, 1 Cb, 2 Cr.  w, h: LUMA size of the PU; mv in internal 1/16 luma units.
// ------------------------------------------------------------------------------------------------------------------
extern "C" void ref_pred_inter_blk( int comp, const int16_t *planeY, int strideY, const int16_t *planeC, int strideC, int picW, int picH, int puX, int puY,
                                    int w, int h, int mvHor, int mvVer, int bi, int bitDepth, int imv, int16_t *dst, int dstStride )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pic = nullptr;
  if( !pic )
  {
    pic = zeroNew<Picture>();
    for( int c = 1; c < 3; c++ ) r.is.m_filteredBlockTmp[0][c] = ( Pel * ) xMalloc( Pel, ( MAX_CU_SIZE + 16 + 4 ) * ( MAX_CU_SIZE + 1 + 16 + 7 + 4 ) );
  }
  r.pps.setPicWidthInLumaSamples( picW );
  r.pps.setPicHeightInLumaSamples( picH );
  const UnitArea ua( CHROMA_420, Area( puX, puY, w, h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  r.cu.imv = imv;
  pic->chromaFormat = CHROMA_420;
  pic->unscaledPic  = pic;
  Pel *py = const_cast<Pel *>( planeY ), *pc = const_cast<Pel *>( planeC );
  pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf(
    PelUnitBuf( CHROMA_420, PelBuf( py, strideY, picW, picH ), PelBuf( pc, strideC, picW / 2, picH / 2 ), PelBuf( pc, strideC, picW / 2, picH / 2 ) ) );
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  const int  cw = comp ? w / 2 : w, ch = comp ? h / 2 : h;
  PelBuf     d( dst, dstStride, cw, ch );
  PelUnitBuf dstPic( CHROMA_420, comp == 0 ? d : PelBuf(), comp == 1 ? d : PelBuf(), comp == 2 ? d : PelBuf() );
  r.is.xPredInterBlk( ComponentID( comp ), r.pu, pic, Mv( mvHor, mvVer ), dstPic, bi != 0, clp, false, false );
  r.cu.imv = 0;
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
}

// ------------------------------------------------------------------------------------------------------------------
// GEO blending: InterpolationFilter::m_weightedGeoBlk (x86, simd 1) / xWeightedGeoBlk (scalar, simd 0), CommonLib/InterpolationFilter.cpp:902-957.
// ref_geo_walk gives what a device trampoline would pass on: which prestored plane, the first weight and the two steps, derived from the
// reference's own tables (g_GeoParams, g_angle2mask, g_angle2mirror, g_weightOffset) the way those functions do.
// ------------------------------------------------------------------------------------------------------------------
extern "C" void ref_geo_weights( int maskIdx, int16_t *out )
{
  ensureRom();
  memcpy( out, g_globalGeoWeights[maskIdx], sizeof( int16_t ) * GEO_WEIGHT_MASK_SIZE * GEO_WEIGHT_MASK_SIZE );
}

extern "C" void ref_geo_walk( int splitDir, int comp, int lumaW, int lumaH, int out[4] )
{
  ensureRom();
  const int  M = GEO_WEIGHT_MASK_SIZE;
  const int  angle = g_GeoParams[splitDir][0], mirror = g_angle2mirror[angle];
  const int  wIdx = floorLog2( lumaW ) - GEO_MIN_CU_LOG2, hIdx = floorLog2( lumaH ) - GEO_MIN_CU_LOG2;
  const int  ox = g_weightOffset[splitDir][hIdx][wIdx][0], oy = g_weightOffset[splitDir][hIdx][wIdx][1];
  const int  sc = comp ? 1 : 0;   // 4:2:0
  out[0] = g_angle2mask[angle];
  out[1] = mirror == 2 ? ( M - 1 - oy ) * M + ox : mirror == 1 ? oy * M + ( M - 1 - ox ) : oy * M + ox;
  out[2] = ( mirror == 1 ? -1 : 1 ) * ( 1 << sc );
  out[3] = ( mirror == 2 ? -M : M ) * ( 1 << sc );
}

extern "C" void ref_weighted_geo_blk( int simd, int splitDir, int comp, int lumaW, int lumaH, const int16_t *src0, int s0Stride, const int16_t *src1,
                                      int s1Stride, int16_t *dst, int dstStride, int bitDepth )
{
  ensureRom();
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  const UnitArea ua( CHROMA_420, Area( 0, 0, lumaW, lumaH ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  for( int c = 0; c < 3; c++ )
  {
    ClpRng &clp = r.slice.getClpRngs().comp[c];
    clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  }
  const int w = comp ? lumaW / 2 : lumaW, h = comp ? lumaH / 2 : lumaH;
  auto mk = [&]( const int16_t *p, int stride ) {
    PelBuf b( const_cast<Pel *>( p ), stride, w, h );
    return PelUnitBuf( CHROMA_420, comp == 0 ? b : PelBuf(), comp == 1 ? b : PelBuf(), comp == 2 ? b : PelBuf() );
  };
  PelUnitBuf a = mk( src0, s0Stride ), b = mk( src1, s1Stride ), d = mk( dst, dstStride );
  if( simd ) r.is.m_if.m_weightedGeoBlk( r.pu, w, h, ComponentID( comp ), uint8_t( splitDir ), d, a, b );
  else       InterpolationFilter::xWeightedGeoBlk( r.pu, w, h, ComponentID( comp ), uint8_t( splitDir ), d, a, b );
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
}

// ------------------------------------------------------------------------------------------------------------------
// BDOF of one bi-predicted luma PU with the reference's own pieces: per region of at most 16 x 16 (the cut xSubPuBio makes,
// InterPrediction.cpp:414-417) xPredInterBlk(..., bioApplied = true) for both lists (:660-810) and applyBiOptFlow (:1233-1334).
// simd 0: scalar g_pelBufOP (Buffer.cpp:88-200), 1: the x86 entries (initPelBufOpsX86).  plane0 / plane1: origins of the two reference
// luma planes (same stride and size).
// ------------------------------------------------------------------------------------------------------------------
extern "C" void ref_bdof_pu( int simd, const int16_t *plane0, const int16_t *plane1, int stride, int picW, int picH, int puX, int puY, int w, int h,
                             int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver, int bitDepth, int16_t *dst, int dstStride )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pic = nullptr;
  static Pel     *dummy = nullptr;
  if( !pic )
  {
    pic   = zeroNew<Picture>();
    dummy = ( Pel * ) xMalloc( Pel, MAX_CU_SIZE * MAX_CU_SIZE );
  }
  if( !r.is.m_gradX0 )
  {
    r.is.m_gradX0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); r.is.m_gradY0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
    r.is.m_gradX1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); r.is.m_gradY1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
  }
  const PelBufferOps saved = g_pelBufOP;
  g_pelBufOP = PelBufferOps();
  if( simd ) g_pelBufOP.initPelBufOpsX86();
  r.pps.setPicWidthInLumaSamples( picW );
  r.pps.setPicHeightInLumaSamples( picH );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  r.cu.imv = 0;
  pic->chromaFormat = CHROMA_420;
  pic->unscaledPic  = pic;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  for( int c = 0; c < 3; c++ ) r.slice.getClpRngs().comp[c] = clp;
  BitDepths bds; bds.recon[CHANNEL_TYPE_LUMA] = bds.recon[CHANNEL_TYPE_CHROMA] = bitDepth;
  const Mv  mv[2] = { Mv( mv0Hor, mv0Ver ), Mv( mv1Hor, mv1Ver ) };
  const int dx = std::min( ( int ) MAX_BDOF_APPLICATION_REGION, w ), dy = std::min( ( int ) MAX_BDOF_APPLICATION_REGION, h );
  for( int y = 0; y < h; y += dy )
  {
    for( int x = 0; x < w; x += dx )
    {
      const UnitArea ua( CHROMA_420, Area( puX + x, puY + y, dx, dy ) );
      r.cu.UnitArea::operator=( ua );
      r.pu.UnitArea::operator=( ua );
      PelUnitBuf scratch( CHROMA_420, PelBuf( dummy, dx, dx, dy ), PelBuf(), PelBuf() );
      for( int l = 0; l < 2; l++ )
      {
        Pel *py = const_cast<Pel *>( l ? plane1 : plane0 );
        pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_420, PelBuf( py, stride, picW, picH ), PelBuf(), PelBuf() ) );
        r.is.m_iRefListIdx = l;
        r.is.xPredInterBlk( COMPONENT_Y, r.pu, pic, mv[l], scratch, true, clp, true, false );
      }
      PelUnitBuf  out( CHROMA_420, PelBuf( dst + y * dstStride + x, dstStride, dx, dy ), PelBuf(), PelBuf() );
      CPelUnitBuf none0( CHROMA_420, CPelBuf( dummy, dx, dx, dy ), CPelBuf(), CPelBuf() ), none1( CHROMA_420, CPelBuf( dummy, dx, dx, dy ), CPelBuf(), CPelBuf() );
      r.is.applyBiOptFlow( r.pu, none0, none1, 0, 0, out, bds );
    }
  }
  g_pelBufOP = saved;
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
}

// ------------------------------------------------------------------------------------------------------------------
// DMVR of one bi-predicted luma PU through the reference's own InterPrediction::xProcessDMVR (InterPrediction.cpp:1997-2195).  The rig PU is
// 4:0:0, so the member runs its luma part only: xPrefetch, xinitMC (bilinear), xDMVRCost through the DF_SAD table, xBIPMVRefine, the error
// surface, xPad, xFinalPaddedMCForDMVR and xWeightedAverage (with applyBiOptFlow when bioApplied and the cost allows).
// mvdOut: pu.mvdL0SubPu, [num][2].
// ------------------------------------------------------------------------------------------------------------------
extern "C" void ref_dmvr_pu( const int16_t *plane0, const int16_t *plane1, int stride, int picW, int picH, int ctuSize, int puX, int puY, int w, int h,
                             int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver, int bitDepth, int bioApplied, int16_t *dst, int dstStride, int32_t *mvdOut )
{
  ensureRom();
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pics[2] = { nullptr, nullptr };
  InterPrediction &ip = r.is;
  if( !pics[0] ) { pics[0] = zeroNew<Picture>(); pics[1] = zeroNew<Picture>(); }
  const size_t dm = ( MAX_CU_SIZE + ( 2 * DMVR_NUM_ITERATION ) ), dr = dm + NTAPS_LUMA;
  if( !ip.m_cYuvPredTempDMVRL0 ) { ip.m_cYuvPredTempDMVRL0 = ( Pel * ) xMalloc( Pel, dm * dm ); ip.m_cYuvPredTempDMVRL1 = ( Pel * ) xMalloc( Pel, dm * dm ); }
  if( !ip.m_cRefSamplesDMVRL0[0] ) { ip.m_cRefSamplesDMVRL0[0] = ( Pel * ) xMalloc( Pel, dr * dr ); ip.m_cRefSamplesDMVRL1[0] = ( Pel * ) xMalloc( Pel, dr * dr ); }
  if( !ip.m_acYuvPred[0][0] ) { ip.m_acYuvPred[0][0] = ( Pel * ) xMalloc( Pel, MAX_CU_SIZE * MAX_CU_SIZE ); ip.m_acYuvPred[1][0] = ( Pel * ) xMalloc( Pel, MAX_CU_SIZE * MAX_CU_SIZE ); }
  if( !ip.m_gradX0 )
  {
    ip.m_gradX0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); ip.m_gradY0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
    ip.m_gradX1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); ip.m_gradY1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
  }
  ip.m_pcRdCost = &r.rd;
  const PelBufferOps saved = g_pelBufOP;
  g_pelBufOP = PelBufferOps();
  g_pelBufOP.initPelBufOpsX86();
  r.pps.setPicWidthInLumaSamples( picW );
  r.pps.setPicHeightInLumaSamples( picH );
  r.sps.setMaxCUWidth( ctuSize );
  r.sps.setMaxCUHeight( ctuSize );
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, bitDepth );
  r.sps.setBitDepth( CHANNEL_TYPE_CHROMA, bitDepth );
  r.slice.m_pcSPS = &r.sps;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  for( int c = 0; c < 3; c++ ) r.slice.getClpRngs().comp[c] = clp;
  for( int l = 0; l < 2; l++ )
  {
    Picture *pic = pics[l];
    pic->chromaFormat = CHROMA_400;
    pic->unscaledPic  = pic;
    Pel *py = const_cast<Pel *>( l ? plane1 : plane0 );
    pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_400, PelBuf( py, stride, picW, picH ) ) );
    r.slice.m_apcRefPicList[l][0] = pic;
    r.slice.m_scalingRatio[l][0]  = SCALE_1X;
  }
  const UnitArea ua( CHROMA_400, Area( puX, puY, w, h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_400;
  r.pu.chromaFormat = CHROMA_400;
  r.cu.imv = 0;
  r.cu.BcwIdx = BCW_DEFAULT;
  r.cu.affine = false;
  r.cu.geoFlag = false;
  r.pu.ciipFlag = false;
  r.pu.refIdx[0] = r.pu.refIdx[1] = 0;
  r.pu.mv[0] = Mv( mv0Hor, mv0Ver );
  r.pu.mv[1] = Mv( mv1Hor, mv1Ver );
  PelUnitBuf out( CHROMA_400, PelBuf( dst, dstStride, w, h ) );
  ip.xProcessDMVR( r.pu, out, r.slice.clpRngs(), bioApplied != 0 );
  if( mvdOut )
  {
    const int n = ( w / std::min( w, 16 ) ) * ( h / std::min( h, 16 ) );
    for( int i = 0; i < n; i++ ) { mvdOut[2 * i] = r.pu.mvdL0SubPu[i].hor; mvdOut[2 * i + 1] = r.pu.mvdL0SubPu[i].ver; }
  }
  g_pelBufOP = saved;
  r.slice.m_apcRefPicList[0][0] = r.slice.m_apcRefPicList[1][0] = nullptr;
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
}

// The same member on a 4:2:0 PU: luma and both chroma planes (xPrefetch forLuma = 0, xPad and the padded 4-tap prediction of moved sub-PUs).
// planes[l][c]: origin of component c of reference picture l; dst[c]: the three prediction blocks.
extern "C" void ref_dmvr_pu420( const int16_t *const planes[2][3], int strideY, int strideC, int picW, int picH, int ctuSize, int puX, int puY, int w, int h,
                                int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver, int bitDepth, int bioApplied, int16_t *const dst[3], int dstStrideY,
                                int dstStrideC, int32_t *mvdOut )
{
  ensureRom();
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pics[2] = { nullptr, nullptr };
  InterPrediction &ip = r.is;
  if( !pics[0] ) { pics[0] = zeroNew<Picture>(); pics[1] = zeroNew<Picture>(); }
  const size_t dm = ( MAX_CU_SIZE + ( 2 * DMVR_NUM_ITERATION ) ), dr = dm + NTAPS_LUMA;
  if( !ip.m_cYuvPredTempDMVRL0 ) { ip.m_cYuvPredTempDMVRL0 = ( Pel * ) xMalloc( Pel, dm * dm ); ip.m_cYuvPredTempDMVRL1 = ( Pel * ) xMalloc( Pel, dm * dm ); }
  for( int c = 0; c < 3; c++ )
  {
    if( !ip.m_cRefSamplesDMVRL0[c] ) { ip.m_cRefSamplesDMVRL0[c] = ( Pel * ) xMalloc( Pel, dr * dr ); ip.m_cRefSamplesDMVRL1[c] = ( Pel * ) xMalloc( Pel, dr * dr ); }
    if( !ip.m_acYuvPred[0][c] ) { ip.m_acYuvPred[0][c] = ( Pel * ) xMalloc( Pel, MAX_CU_SIZE * MAX_CU_SIZE ); ip.m_acYuvPred[1][c] = ( Pel * ) xMalloc( Pel, MAX_CU_SIZE * MAX_CU_SIZE ); }
    if( !ip.m_filteredBlockTmp[0][c] ) ip.m_filteredBlockTmp[0][c] = ( Pel * ) xMalloc( Pel, ( MAX_CU_SIZE + 16 + 4 ) * ( MAX_CU_SIZE + 1 + 16 + 7 + 4 ) );
  }
  if( !ip.m_gradX0 )
  {
    ip.m_gradX0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); ip.m_gradY0 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
    ip.m_gradX1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE ); ip.m_gradY1 = ( Pel * ) xMalloc( Pel, BIO_TEMP_BUFFER_SIZE );
  }
  ip.m_pcRdCost = &r.rd;
  const PelBufferOps saved = g_pelBufOP;
  g_pelBufOP = PelBufferOps();
  g_pelBufOP.initPelBufOpsX86();
  r.pps.setPicWidthInLumaSamples( picW );
  r.pps.setPicHeightInLumaSamples( picH );
  r.sps.setMaxCUWidth( ctuSize );
  r.sps.setMaxCUHeight( ctuSize );
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, bitDepth );
  r.sps.setBitDepth( CHANNEL_TYPE_CHROMA, bitDepth );
  r.slice.m_pcSPS = &r.sps;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  for( int c = 0; c < 3; c++ ) r.slice.getClpRngs().comp[c] = clp;
  for( int l = 0; l < 2; l++ )
  {
    Picture *pic = pics[l];
    pic->chromaFormat = CHROMA_420;
    pic->unscaledPic  = pic;
    pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_420, PelBuf( const_cast<Pel *>( planes[l][0] ), strideY, picW, picH ),
                                                               PelBuf( const_cast<Pel *>( planes[l][1] ), strideC, picW / 2, picH / 2 ),
                                                               PelBuf( const_cast<Pel *>( planes[l][2] ), strideC, picW / 2, picH / 2 ) ) );
    r.slice.m_apcRefPicList[l][0] = pic;
    r.slice.m_scalingRatio[l][0]  = SCALE_1X;
  }
  const UnitArea ua( CHROMA_420, Area( puX, puY, w, h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
  r.cu.imv = 0;
  r.cu.BcwIdx = BCW_DEFAULT;
  r.cu.affine = false;
  r.cu.geoFlag = false;
  r.pu.ciipFlag = false;
  r.pu.refIdx[0] = r.pu.refIdx[1] = 0;
  r.pu.mv[0] = Mv( mv0Hor, mv0Ver );
  r.pu.mv[1] = Mv( mv1Hor, mv1Ver );
  PelUnitBuf out( CHROMA_420, PelBuf( dst[0], dstStrideY, w, h ), PelBuf( dst[1], dstStrideC, w / 2, h / 2 ), PelBuf( dst[2], dstStrideC, w / 2, h / 2 ) );
  ip.xProcessDMVR( r.pu, out, r.slice.clpRngs(), bioApplied != 0 );
  if( mvdOut )
  {
    const int n = ( w / std::min( w, 16 ) ) * ( h / std::min( h, 16 ) );
    for( int i = 0; i < n; i++ ) { mvdOut[2 * i] = r.pu.mvdL0SubPu[i].hor; mvdOut[2 * i + 1] = r.pu.mvdL0SubPu[i].ver; }
  }
  g_pelBufOP = saved;
  r.slice.m_apcRefPicList[0][0] = r.slice.m_apcRefPicList[1][0] = nullptr;
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
}

// ------------------------------------------------------------------------------------------------------------------
// LFNST kernels: the real TrQuant::fwdLfnstNxN / invLfnstNxN (TrQuant.cpp:233-311) and the reference's trained core matrices (Rom.h:132-133),
// handed out as DATA the way an integration hands them to the device library.
// ------------------------------------------------------------------------------------------------------------------
#include "CommonLib/TrQuant.h"
extern "C" void ref_lfnst_tables( int8_t *out8x8, int8_t *out4x4 )
{
  memcpy( out8x8, g_lfnst8x8, sizeof( g_lfnst8x8 ) );
  memcpy( out4x4, g_lfnst4x4, sizeof( g_lfnst4x4 ) );
}
extern "C" void ref_lfnst( int inverse, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize )
{
  static TrQuant *tq = new TrQuant();
  int             in[48], out[48];
  memset( in, 0, sizeof( in ) );
  memcpy( in, src, sizeof( int ) * ( inverse ? zeroOutSize : ( size > 4 ? 48 : 16 ) ) );
  if( inverse ) tq->invLfnstNxN( in, out, mode, index, size, zeroOutSize, 15 );
  else tq->fwdLfnstNxN( in, out, mode, index, size, zeroOutSize );
  memcpy( dst, out, sizeof( int ) * ( size > 4 ? 48 : 16 ) );
}

// ------------------------------------------------------------------------------------------------------------------
// The 2-D compositions TrQuant::xT / xIT (TrQuant.cpp:776-923: getTrTypes, shift1 / shift2, skipWidth / skipHeight, the Pel truncation of
// xIT's output) and the MTS candidate pre-selection TrQuant::transformNxN( tu, compID, cQP, &trModes, maxCand ) (:950-1019, including the
// transform-skip candidate xTransformSkip + scaleSAD), called as the real protected members on a rig TU of an inter CU with explicit MTS on.
// ------------------------------------------------------------------------------------------------------------------
namespace
{
struct TrRig
{
  TrQuant        tq;
  TransformUnit *tu = nullptr;
  bool           resiCreated = false;
};
TrRig *g_trRig = nullptr;

TransformUnit &trSetup( int w, int h, int bitDepth, int mtsIdx )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  if( !g_trRig ) g_trRig = zeroNew<TrRig>();
  ensureRom();
  MeRig &r = *g_rig;
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, bitDepth );
  r.sps.setBitDepth( CHANNEL_TYPE_CHROMA, bitDepth );
  r.sps.setUseMTS( true );
  r.sps.setUseInterMTS( true );
  r.sps.setUseIntraMTS( true );
  r.sps.setUseLFNST( false );
  r.slice.setSPS( &r.sps );
  const UnitArea ua( CHROMA_400, Area( 0, 0, w, h ) );
  if( !g_trRig->tu ) g_trRig->tu = zeroNew<TransformUnit>( ua );
  TransformUnit &tu = *g_trRig->tu;
  tu.UnitArea::operator=( ua );
  tu.cs = &r.cs;
  tu.cu = &r.cu;
  tu.chromaFormat = CHROMA_400;
  tu.mtsIdx[0]    = ( uint8_t ) mtsIdx;
  tu.noResidual   = false;
  r.cu.predMode = MODE_INTER; r.cu.lfnstIdx = 0; r.cu.sbtInfo = 0; r.cu.ispMode = 0; r.cu.mipFlag = false; r.cu.colorTransform = false; r.cu.bdpcmMode = 0;
  return tu;
}
}   // namespace

extern "C" int ref_xT( const int16_t *resi, int stride, int w, int h, int bitDepth, int mtsIdx, int32_t *coef )
{
  TransformUnit &tu = trSetup( w, h, bitDepth, mtsIdx );
  const CPelBuf src( resi, stride, w, h );
  CoeffBuf      dst( coef, w, w, h );
  g_trRig->tq.xT( tu, COMPONENT_Y, src, dst, w, h );
  return 0;
}

extern "C" int ref_xIT( const int32_t *coef, int w, int h, int bitDepth, int mtsIdx, int16_t *resi, int stride )
{
  TransformUnit &tu = trSetup( w, h, bitDepth, mtsIdx );
  const CCoeffBuf src( coef, w, w, h );
  PelBuf          dst( resi, stride, w, h );
  g_trRig->tq.xIT( tu, COMPONENT_Y, src, dst );
  return 0;
}

// trModes in: mtsIdx[i]; out: test[i] (the .second the real member leaves) -- the residual is copied into the coding structure's own
// residual buffer, where the member reads it (cs.getResiBuf)
extern "C" int ref_transformNxN_select( const int16_t *resi, int stride, int w, int h, int bitDepth, const uint8_t *mtsIdx, int numCand, int maxCand,
                                        uint8_t *test )
{
  TransformUnit &tu = trSetup( w, h, bitDepth, 0 );
  MeRig &r = *g_rig;
  if( !g_trRig->resiCreated )
  {
    r.cs.area = UnitArea( CHROMA_400, Area( 0, 0, MAX_TB_SIZEY, MAX_TB_SIZEY ) );
    r.cs.m_resi.create( r.cs.area );
    r.cs.parent = &r.cs;   // only consulted to decide whether the residual address is folded to the CTU (KEEP_PRED_AND_RESI_SIGNALS off): no folding
    g_trRig->resiCreated = true;
  }
  PelBuf dst = r.cs.getResiBuf( tu.Y() );
  for( int y = 0; y < h; y++ ) memcpy( dst.buf + y * dst.stride, resi + y * stride, sizeof( Pel ) * w );
  std::vector<TrMode> modes;
  for( int i = 0; i < numCand; i++ ) modes.push_back( TrMode( mtsIdx[i], true ) );
  r.cu.qp = 32; r.cu.chromaQpAdj = 0; r.cu.treeType = TREE_D; r.cu.modeType = MODE_TYPE_ALL;
  r.sps.setQpBDOffset( CHANNEL_TYPE_LUMA, 6 * ( bitDepth - 8 ) );
  r.sps.setInternalMinusInputBitDepth( CHANNEL_TYPE_LUMA, 0 );
  const QpParam cQP( tu, COMPONENT_Y );   // not read by this overload
  g_trRig->tq.transformNxN( tu, COMPONENT_Y, cQP, &modes, maxCand );
  for( int i = 0; i < numCand; i++ ) test[i] = modes[i].second;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The AMVP helpers of predInterSearch as the real members: InterSearch::xEstimateMvPredAMVP (:3088-3128, bFilled = true: the
// candidates are the caller's) with xGetTemplateCost (:3235-3270), and xCheckBestMVP (:3185-3232).  Same rig as ref_motion_estimation.
// ------------------------------------------------------------------------------------------------------------------
extern "C" void ref_estimate_mvp_amvp( const vo_mest_job_t *j, int *mvpIdx, int *mvPredHor, int *mvPredVer, uint64_t *distBiP )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static Picture *pic = nullptr;
  static bool     storage = false;
  if( !pic ) pic = zeroNew<Picture>();
  if( !storage )
  {
    const UnitArea lcu( CHROMA_400, Area( 0, 0, MAX_CU_SIZE, MAX_CU_SIZE ) );
    if( r.is.m_tmpStorageLCU.bufs.empty() ) r.is.m_tmpStorageLCU.create( lcu );
    storage = true;
  }
  r.sps.setMaxCUWidth( j->ctuSize );
  r.sps.setMaxCUHeight( j->ctuSize );
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, j->bitDepth );
  r.pps.setPicWidthInLumaSamples( j->picW );
  r.pps.setPicHeightInLumaSamples( j->picH );
  r.pps.setUseWP( false );
  r.pps.setWPBiPred( false );
  r.slice.setPPS( &r.pps );
  r.slice.setSPS( &r.sps );
  r.slice.setSliceType( B_SLICE );
  const UnitArea ua( CHROMA_400, Area( j->puX, j->puY, j->w, j->h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_400;
  r.pu.chromaFormat = CHROMA_400;
  r.cu.imv    = j->imv;
  r.cu.BcwIdx = BCW_DEFAULT;
  r.cu.affine = false;
  r.rd.m_motionLambda = j->motionLambda;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << j->bitDepth ) - 1; clp.bd = j->bitDepth; clp.n = 0;
  r.slice.m_clpRngs.comp[COMPONENT_Y] = clp;
  Pel *origin = const_cast<Pel *>( j->ref ) - ( ptrdiff_t ) j->puY * j->refStride - j->puX;
  pic->chromaFormat = CHROMA_400;
  pic->unscaledPic  = pic;
  pic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_400, PelBuf( origin, j->refStride, j->picW, j->picH ) ) );
  r.slice.m_apcRefPicList[REF_PIC_LIST_0][0] = pic;
  for( int i = 0; i < 2; i++ ) r.is.m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] = j->mvpIdxBits[i];
  AMVPInfo amvp;
  amvp.numCand = j->numAmvpCand;
  for( int i = 0; i < 2; i++ ) amvp.mvCand[i] = Mv( j->amvpCand[i][0], j->amvpCand[i][1] );
  PelUnitBuf origBuf( CHROMA_400, PelBuf( const_cast<Pel *>( j->org ), j->orgStride, j->w, j->h ) );
  Mv         mvPred;
  Distortion dist = 0;
  r.is.xEstimateMvPredAMVP( r.pu, origBuf, REF_PIC_LIST_0, 0, mvPred, amvp, true, &dist );
  *mvpIdx = r.pu.mvpIdx[REF_PIC_LIST_0]; *mvPredHor = mvPred.hor; *mvPredVer = mvPred.ver; *distBiP = dist;
  r.cu.imv = 0;
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
}

extern "C" void ref_check_best_mvp( double motionLambda, int imv, int numCand, const int cands[2][2], const unsigned idxBits[2], int mvHor, int mvVer,
                                    int *mvPredHor, int *mvPredVer, int *mvpIdx, unsigned *bits, uint64_t *cost )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  r.rd.m_motionLambda = motionLambda;
  for( int i = 0; i < 2; i++ ) r.is.m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] = idxBits[i];
  AMVPInfo amvp;
  amvp.numCand = numCand;
  for( int i = 0; i < 2; i++ ) amvp.mvCand[i] = Mv( cands[i][0], cands[i][1] );
  Mv         pred( *mvPredHor, *mvPredVer );
  uint32_t   b = *bits;
  Distortion c = *cost;
  r.is.xCheckBestMVP( REF_PIC_LIST_0, Mv( mvHor, mvVer ), pred, *mvpIdx, amvp, b, c, ( uint8_t ) imv );
  *mvPredHor = pred.hor; *mvPredVer = pred.ver; *bits = b; *cost = c;
}

// ------------------------------------------------------------------------------------------------------------------
// Affine motion estimation: the real InterPrediction::xPredAffineBlk (luma, PROF included; InterPrediction.cpp:856-1232) and the real
// InterSearch::xAffineMotionEstimation (InterSearch.cpp:5340-5775: gradient iterations with solveEqual, control-point refinement) on the rig.
// ------------------------------------------------------------------------------------------------------------------
namespace
{
Picture   *g_affPic = nullptr;
PicHeader *g_affPh  = nullptr;

void affineSetup( MeRig &r, const vo_affine_pred_t *p )
{
  if( !g_affPic ) { g_affPic = zeroNew<Picture>(); g_affPh = zeroNew<PicHeader>(); }
  if( !r.is.m_storedMv ) r.is.m_storedMv = new Mv[( MAX_CU_SIZE / MIN_PU_SIZE ) * ( MAX_CU_SIZE / MIN_PU_SIZE )];
  r.sps.setMaxCUWidth( p->ctuSize ); r.sps.setMaxCUHeight( p->ctuSize );
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, p->bitDepth );
  r.sps.setUsePROF( p->profAllowed != 0 );
  r.pps.setPicWidthInLumaSamples( p->picW ); r.pps.setPicHeightInLumaSamples( p->picH );
  r.pps.setUseWP( false ); r.pps.setWPBiPred( false );
  r.slice.setPPS( &r.pps ); r.slice.setSPS( &r.sps ); r.slice.setSliceType( B_SLICE );
  r.slice.setCheckLDC( !p->profNeedsLargeGrad );
  g_affPh->setDisProfFlag( false );
  r.cs.picHeader = g_affPh;
  r.is.m_skipPROF = false;
  r.is.m_encOnly  = true;
  r.is.m_isBi     = p->profIsBi != 0;
  const UnitArea ua( CHROMA_400, Area( p->puX, p->puY, p->w, p->h ) );
  r.cu.UnitArea::operator=( ua ); r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_400; r.pu.chromaFormat = CHROMA_400;
  r.cu.affine = true; r.cu.affineType = p->sixParam ? AFFINEMODEL_6PARAM : AFFINEMODEL_4PARAM; r.cu.BcwIdx = BCW_DEFAULT;
  r.pu.interDir = ( uint8_t ) p->interDir;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << p->bitDepth ) - 1; clp.bd = p->bitDepth; clp.n = 0;
  r.slice.m_clpRngs.comp[COMPONENT_Y] = clp;
  Pel *origin = const_cast<Pel *>( p->ref ) - ( ptrdiff_t ) p->puY * p->refStride - p->puX;
  g_affPic->chromaFormat = CHROMA_400;
  g_affPic->unscaledPic  = g_affPic;
  g_affPic->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_400, PelBuf( origin, p->refStride, p->picW, p->picH ) ) );
  r.slice.m_apcRefPicList[REF_PIC_LIST_0][0] = g_affPic;
}
void affineRestore( MeRig &r )
{
  r.cu.affine = false; r.cu.imv = 0; r.is.m_encOnly = false; r.is.m_isBi = false; r.sps.setUsePROF( false );
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) ); r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.cu.chromaFormat = CHROMA_420; r.pu.chromaFormat = CHROMA_420;
}
}   // namespace

extern "C" void ref_pred_affine_blk( const vo_affine_pred_t *p, const int mv[3][2], int bi, int16_t *dst, int dstStride )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  affineSetup( r, p );
  Mv m[3] = { Mv( mv[0][0], mv[0][1] ), Mv( mv[1][0], mv[1][1] ), Mv( mv[2][0], mv[2][1] ) };
  PelUnitBuf d( CHROMA_400, PelBuf( dst, dstStride, p->w, p->h ) );
  r.is.m_iRefListIdx = 0;
  r.is.xPredAffineBlk( COMPONENT_Y, r.pu, g_affPic, m, d, bi != 0, r.slice.clpRng( COMPONENT_Y ) );
  affineRestore( r );
}

extern void solveEqual( double dEqualCoeff[7][7], int iOrder, double *dAffinePara );
extern "C" void ref_solve_equal( double eq[7][7], int order, double *para ) { solveEqual( eq, order, para ); }

extern "C" void ref_affine_motion_estimation( const vo_affine_me_job_t *j, vo_affine_me_result_t *res )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  static bool storage = false;
  if( !storage )
  {
    const UnitArea lcu( CHROMA_400, Area( 0, 0, MAX_CU_SIZE, MAX_CU_SIZE ) );
    if( r.is.m_tmpStorageLCU.bufs.empty() ) r.is.m_tmpStorageLCU.create( lcu );
    if( r.is.m_tmpPredStorage[0].bufs.empty() ) r.is.m_tmpPredStorage[0].create( lcu );
    if( r.is.m_tmpPredStorage[1].bufs.empty() ) r.is.m_tmpPredStorage[1].create( lcu );
    r.is.m_tmpAffiStorage.create( lcu );
    r.is.m_tmpAffiError   = new Pel[MAX_CU_SIZE * MAX_CU_SIZE];
    r.is.m_tmpAffiDeri[0] = new int[MAX_CU_SIZE * MAX_CU_SIZE];
    r.is.m_tmpAffiDeri[1] = new int[MAX_CU_SIZE * MAX_CU_SIZE];
    storage = true;
  }
  affineSetup( r, &j->pred );
  r.cu.imv = j->imv;
  r.sps.setUseAffineType( j->useAffineType != 0 );
  r.sps.setUseBcw( false );
  r.slice.setDisableSATDForRD( !j->useSatd );
  r.cfg.setUseAffineAmvrEncOpt( j->amvrEncOpt != 0 );
  r.cfg.setIntraPeriod( j->lowDelayRounds ? -1 : 32 );
  r.cfg.setMCTSEncConstraint( false );
  r.cfg.setClipForBiPredMeEnabled( false );
  r.rd.m_motionLambda = j->lambda;
  r.is.m_hevcCost = j->hevcCost;
  r.is.m_modeCtrl = nullptr;
  AffineAMVPInfo aamvp;
  aamvp.numCand = 2;
  for( int i = 0; i < 2; i++ )
  {
    aamvp.mvCandLT[i] = Mv( j->mvPred[0][0], j->mvPred[0][1] ); aamvp.mvCandRT[i] = Mv( j->mvPred[1][0], j->mvPred[1][1] ); aamvp.mvCandLB[i] = Mv( j->mvPred[2][0], j->mvPred[2][1] );
    r.is.m_auiMVPIdxCost[i][2] = 1;
  }
  if( j->bi )
  {
    PelBuf other = r.is.m_tmpPredStorage[1].getBuf( UnitAreaRelative( r.cu, r.pu ) ).Y();
    for( int y = 0; y < j->pred.h; y++ ) memcpy( other.buf + ( ptrdiff_t ) y * other.stride, j->otherPred + ( ptrdiff_t ) y * j->otherStride, sizeof( Pel ) * j->pred.w );
  }
  PelUnitBuf origBuf( CHROMA_400, PelBuf( const_cast<Pel *>( j->org ), j->orgStride, j->pred.w, j->pred.h ) );
  Mv pred[3], mv[3];
  for( int i = 0; i < 3; i++ ) { pred[i] = Mv( j->mvPred[i][0], j->mvPred[i][1] ); mv[i] = Mv( j->mv[i][0], j->mv[i][1] ); }
  uint32_t   bits = j->bits;
  Distortion cost = std::numeric_limits<Distortion>::max();
  int        mvpIdx = 0;
  r.is.xAffineMotionEstimation( r.pu, origBuf, REF_PIC_LIST_0, pred, 0, mv, bits, cost, mvpIdx, aamvp, j->bi != 0 );
  for( int i = 0; i < 3; i++ ) { res->mv[i][0] = mv[i].hor; res->mv[i][1] = mv[i].ver; }
  res->bits = bits; res->cost = cost; res->iterations = res->refinements = 0;
  r.slice.setDisableSATDForRD( false );
  affineRestore( r );
}

// the reference's own scan tables for the LFNST scatter / gather (TrQuant.cpp:347, 435): g_coefTopLeftDiagScan8x8 for TUs >= 8x8, else the grouped
// diagonal scan of the block
extern "C" void ref_lfnst_scan( int w, int h, int32_t *pos48 )
{
  ensureRom();
  const bool whge3 = w >= 8 && h >= 8;
  const ScanElement *scan = whge3 ? g_coefTopLeftDiagScan8x8[gp_sizeIdxInfo->idxFrom( w )] : g_scanOrder[SCAN_GROUPED_4x4][SCAN_DIAG][gp_sizeIdxInfo->idxFrom( w )][gp_sizeIdxInfo->idxFrom( h )];
  for( int k = 0; k < ( whge3 ? 48 : 16 ); k++ ) pos48[k] = scan[k].idx;
}

// ------------------------------------------------------------------------------------------------------------------
// SMVD: the real InterSearch::xGetSymmetricCost / xSymmetricMotionEstimation / symmvdCheckBestMvp (InterSearch.cpp:4341-4518, 7787-7886) on
// caller-supplied planes.  The searched list is list 0 (as in predInterSearch :2662), its symmetric reference index 0 in both lists.
// ------------------------------------------------------------------------------------------------------------------
namespace
{
void smvdSetup( MeRig &r, const vo_smvd_job_t *j, PelUnitBuf &origBuf )
{
  static Picture *pic[2] = { nullptr, nullptr };
  const UnitArea  lcu( CHROMA_400, Area( 0, 0, MAX_CU_SIZE, MAX_CU_SIZE ) );
  if( r.is.m_tmpStorageLCU.bufs.empty() ) r.is.m_tmpStorageLCU.create( lcu );
  if( r.is.m_tmpPredStorage[0].bufs.empty() ) r.is.m_tmpPredStorage[0].create( lcu );
  if( r.is.m_tmpPredStorage[1].bufs.empty() ) r.is.m_tmpPredStorage[1].create( lcu );
  r.sps.setMaxCUWidth( j->ctuSize );
  r.sps.setMaxCUHeight( j->ctuSize );
  r.sps.setBitDepth( CHANNEL_TYPE_LUMA, j->bitDepth );
  r.sps.setUseBcw( j->bcwWeightTar != 4 );
  r.pps.setPicWidthInLumaSamples( j->picW );
  r.pps.setPicHeightInLumaSamples( j->picH );
  r.pps.setUseWP( false );
  r.pps.setWPBiPred( false );
  r.slice.setPPS( &r.pps );
  r.slice.setSPS( &r.sps );
  r.slice.setSliceType( B_SLICE );
  r.slice.setDisableSATDForRD( !j->useSatd );
  r.slice.m_symRefIdx[0] = r.slice.m_symRefIdx[1] = 0;
  const UnitArea ua( CHROMA_400, Area( j->puX, j->puY, j->w, j->h ) );
  r.cu.UnitArea::operator=( ua );
  r.pu.UnitArea::operator=( ua );
  r.cu.chromaFormat = CHROMA_400;
  r.pu.chromaFormat = CHROMA_400;
  r.cu.imv    = j->imv;
  r.cu.affine = false;
  r.cu.BcwIdx = BCW_DEFAULT;
  for( int i = 0; i < BCW_NUM; i++ ) if( g_BcwWeights[i] == j->bcwWeightTar ) r.cu.BcwIdx = ( uint8_t ) i;
  r.cfg.setClipForBiPredMeEnabled( j->clipBiPred != 0 );
  r.cfg.setMCTSEncConstraint( false );
  r.rd.m_motionLambda = j->motionLambda;
  ClpRng clp; clp.min = 0; clp.max = ( 1 << j->bitDepth ) - 1; clp.bd = j->bitDepth; clp.n = 0;
  r.slice.m_clpRngs.comp[COMPONENT_Y] = clp;
  for( int l = 0; l < 2; l++ )
  {
    if( !pic[l] ) pic[l] = zeroNew<Picture>();
    Pel *origin = const_cast<Pel *>( j->ref[l] ) - ( ptrdiff_t ) j->puY * j->refStride[l] - j->puX;
    pic[l]->chromaFormat = CHROMA_400;
    pic[l]->unscaledPic  = pic[l];
    pic[l]->m_bufs[PIC_RECONSTRUCTION].createFromBuf( PelUnitBuf( CHROMA_400, PelBuf( origin, j->refStride[l], j->picW, j->picH ) ) );
    r.slice.m_apcRefPicList[l][0] = pic[l];
  }
  for( int i = 0; i < 2; i++ ) r.is.m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] = j->mvpIdxBits[i];
  origBuf = PelUnitBuf( CHROMA_400, PelBuf( const_cast<Pel *>( j->org ), j->orgStride, j->w, j->h ) );
}
void smvdRestore( MeRig &r )
{
  r.cu.imv = 0; r.cu.BcwIdx = BCW_DEFAULT;
  r.slice.setDisableSATDForRD( false );
  r.cfg.setClipForBiPredMeEnabled( false );
  r.sps.setUseBcw( false );
  r.cu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.pu.UnitArea::operator=( UnitArea( CHROMA_420, Area( 0, 0, 8, 8 ) ) );
  r.cu.chromaFormat = CHROMA_420;
  r.pu.chromaFormat = CHROMA_420;
}
}   // namespace

extern "C" uint64_t ref_symmetric_cost( const vo_smvd_job_t *j, const int mvCur[2], const int mvTar[2] )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  PelUnitBuf origBuf;
  smvdSetup( r, j, origBuf );
  MvField cur, tar;
  cur.setMvField( Mv( mvCur[0], mvCur[1] ), 0 ); tar.setMvField( Mv( mvTar[0], mvTar[1] ), 0 );
  const Distortion d = r.is.xGetSymmetricCost( r.pu, origBuf, REF_PIC_LIST_0, cur, tar, r.cu.BcwIdx );
  smvdRestore( r );
  return d;
}

extern "C" void ref_symmetric_me( const vo_smvd_job_t *j, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], uint64_t *cost )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  PelUnitBuf origBuf;
  smvdSetup( r, j, origBuf );
  Mv pc( predCur[0], predCur[1] ), pt( predTar[0], predTar[1] );
  MvField cur, tar;
  cur.setMvField( Mv( mvCur[0], mvCur[1] ), 0 ); tar.setMvField( Mv( mvTar[0], mvTar[1] ), 0 );
  Distortion c = *cost;
  r.is.xSymmetricMotionEstimation( r.pu, origBuf, pc, pt, REF_PIC_LIST_0, cur, tar, c, r.cu.BcwIdx );
  mvCur[0] = cur.mv.hor; mvCur[1] = cur.mv.ver; mvTar[0] = tar.mv.hor; mvTar[1] = tar.mv.ver; *cost = c;
  smvdRestore( r );
}

extern "C" void ref_symmvd_check_best_mvp( const vo_smvd_job_t *j, const int curMv[2], int skip, int predSym[2][2], int mvpIdxSym[2], uint64_t *bestCost )
{
  if( !g_rig ) g_rig = zeroNew<MeRig>();
  MeRig &r = *g_rig;
  PelUnitBuf origBuf;
  smvdSetup( r, j, origBuf );
  static AMVPInfo ( *amvp )[33] = nullptr;
  if( !amvp ) amvp = new AMVPInfo[2][33];
  for( int l = 0; l < 2; l++ )
  {
    amvp[l][0].numCand = j->numCand[l];
    for( int i = 0; i < 2; i++ ) amvp[l][0].mvCand[i] = Mv( j->cand[l][i][0], j->cand[l][i][1] );
  }
  Mv      pred[2] = { Mv( predSym[0][0], predSym[0][1] ), Mv( predSym[1][0], predSym[1][1] ) };
  int32_t idx[2]  = { mvpIdxSym[0], mvpIdxSym[1] };
  Distortion c = *bestCost;
  r.is.symmvdCheckBestMvp( r.pu, origBuf, Mv( curMv[0], curMv[1] ), REF_PIC_LIST_0, amvp, r.cu.BcwIdx, pred, idx, c, skip != 0 );
  for( int l = 0; l < 2; l++ ) { predSym[l][0] = pred[l].hor; predSym[l][1] = pred[l].ver; mvpIdxSym[l] = idx[l]; }
  *bestCost = c;
  smvdRestore( r );
}
