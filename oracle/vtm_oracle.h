/* vtm_oracle.h -- TEST INFRASTRUCTURE ONLY (see vtm_oracle.c). */
#ifndef VTM_ORACLE_H
#define VTM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VO_DCT2 = 0, VO_DCT8 = 1, VO_DST7 = 2 };   /* TransType, CommonLib/TypeDef.h */

typedef struct
{
  double motionLambda;   /* RdCost::m_motionLambda = sqrt(lambda) */
  int    predHor, predVer;   /* RdCost::m_mvPredictor (quarter-sample units) */
  int    costScale;          /* RdCost::m_iCostScale: 2 integer, 1 half, 0 quarter */
} vo_mvcost_t;

typedef struct { int left, right, top, bottom; } vo_range_t;   /* SearchRange, EncoderLib/InterSearch.h */

/* Everything xTZSearchHelp / xPatternSearch / xPatternSearchFracDIF read for one PU x one reference picture */
typedef struct
{
  const int16_t *org;   /* pcPatternKey */
  int            orgStride;
  const int16_t *ref;   /* piRefY: reference luma plane at the PU position (MV 0,0) */
  int            refStride;
  int            w, h;
  int            subShift;   /* DistParam::subShift after setDistParam(subShiftMode) */
  int            bitDepth;
  unsigned       imvShift;
  vo_mvcost_t    mv;
  int            picW, picH, puX, puY, ctuSize;   /* for clipMv / xClipMv */
} vo_me_ctx_t;

typedef struct
{
  int mvHor, mvVer;   /* rcMv on entry (internal 1/16 precision) */
  int searchRange;    /* m_iSearchRange */
  int extendedSettings, fastSettings, firstSearchStop;
  int hasIntMv2Nx2NPred, intMv2Nx2NPredHor, intMv2Nx2NPredVer;   /* integer precision */
  int numExtraStart;
  int extraStart[16][2];   /* m_uniMvList candidates after de-duplication, internal precision */
} vo_tz_job_t;

typedef struct
{
  int      mvX, mvY;   /* integer MV */
  uint64_t cost;       /* uiBestSad (distortion + MV rate) */
  uint64_t dist;       /* ruiSAD */
  uint64_t nEval;      /* number of distFunc evaluations (statistics only) */
} vo_me_result_t;

typedef struct
{
  int      halfX, halfY;   /* rcMvHalf in {-1,0,1} */
  int      qterX, qterY;   /* rcMvQter in {-1,0,1} */
  uint64_t costHalf, cost;
  uint64_t candHalf[9], candQuarter[9];
} vo_frac_result_t;

uint64_t vo_sad( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift );
uint64_t vo_sad_mask( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift, const int16_t *mask,
                      int maskStride, int stepX, int maskStride2 );
/* BDOF of one bi-predicted luma PU: xSubPuBio + xPredInterBlk(bioApplied) + applyBiOptFlow, CommonLib/InterPrediction.cpp:352-443, 733-810, 1233-1334 */
void vo_bdof_pu( const int16_t *ref0, int stride0, const int16_t *ref1, int stride1, int w, int h, int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver,
                 int bitDepth, int16_t *dst, int dstStride );
/* DMVR of one bi-predicted luma PU: InterPrediction::xProcessDMVR, CommonLib/InterPrediction.cpp:1997-2195 (luma plane) */
void vo_dmvr_pu( const int16_t *plane0, const int16_t *plane1, int stride, int picW, int picH, int ctuSize, int puX, int puY, int w, int h, int mv0Hor,
                 int mv0Ver, int mv1Hor, int mv1Ver, int bitDepth, int bioApplied, int16_t *dst, int dstStride, int32_t *mvdOut );
/* ... and one 4:2:0 chroma plane of the same PU, given pu.mvdL0SubPu (xPrefetch forLuma = 0, xPad, xFinalPaddedMCForDMVR, addAvg) */
void vo_dmvr_chroma( const int16_t *planeC0, const int16_t *planeC1, int strideC, int picW, int picH, int ctuSize, int puX, int puY, int w, int h,
                     int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver, const int32_t *mvd, int bitDepth, int16_t *dst, int dstStride );
/* BCW: AreaBuf::removeWeightHighFreq (Buffer.h:417-460), AreaBuf::addWeightedAvg (Buffer.cpp:365-397); w1 = the list-1 weight g_BcwWeights[bcwIdx] */
void vo_remove_weight_high_freq( int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h, int bcwWeight );
void vo_add_weighted_avg( const int16_t *a, int aStride, const int16_t *b, int bStride, int16_t *dst, int dstStride, int w, int h, int bitDepth, int w1 );
/* TrQuant::fwdLfnstNxN / invLfnstNxN, CommonLib/TrQuant.cpp:233-311; M = the caller's 16 x (size > 4 ? 48 : 16) int8 core matrix */
void vo_fwd_lfnst( const int32_t *src, int32_t *dst, const int8_t *M, int size, int zeroOutSize );
void vo_inv_lfnst( const int32_t *src, int32_t *dst, const int8_t *M, int size, int zeroOutSize );
/* InterpolationFilter::xWeightedGeoBlk, CommonLib/InterpolationFilter.cpp:902-957 */
void vo_weighted_geo_blk( const int16_t *src0, int src0Stride, const int16_t *src1, int src1Stride, int16_t *dst, int dstStride, int w, int h,
                          const int16_t *weight, int stepX, int weightStride, int bitDepth, int clipMin, int clipMax );
uint64_t vo_sse( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h );
uint64_t vo_satd( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h );
int      vo_satd_tile_shape( int w, int h, int *tw, int *th );
int      vo_subshift_for_mode( int w, int h, int subShiftMode );
void     vo_satd8_grid( const int16_t *org, int orgStride, const int16_t *ref, int refStride, int w, int h, int r, uint64_t *out );
unsigned vo_mv_bits( const vo_mvcost_t *mc, int x, int y, unsigned imvShift );
uint64_t vo_mv_cost( const vo_mvcost_t *mc, int x, int y, unsigned imvShift );

extern const int16_t vo_luma_filter[16][8];
extern const int16_t vo_luma_filter_4x4[16][8];
extern const int16_t vo_luma_alt_hpel[8];
extern const int16_t vo_chroma_filter[32][4];

void vo_if_copy( int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int bitDepth, int clipMin,
                 int clipMax, int biMCForDMVR );
void vo_if_filter( int vertical, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h,
                   const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR );
void vo_if_hor( int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac, int isLast, int bitDepth,
                int nFilterIdx, int biMCForDMVR, int useAltHpelIf );
void vo_if_ver( int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac, int isFirst, int isLast,
                int bitDepth, int nFilterIdx, int biMCForDMVR, int useAltHpelIf );
void vo_mc_luma( const int16_t *ref, int refStride, int w, int h, int mvHor, int mvVer, int bi, int bitDepth, int useAltHpelIf, int16_t *dst,
                 int dstStride );
void vo_mc_block( int comp, const int16_t *ref, int refStride, int w, int h, int mvHor, int mvVer, int bi, int bitDepth, int useAltHpelIf,
                  int16_t *dst, int dstStride );
void vo_interp_qpel( const int16_t *pat, int ps, int w, int h, int bitDepth, int qx, int qy, int16_t *dst, int ds );

int vo_tr_matrix( int type, int n, int16_t *out );
int vo_fwd_trans( int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2 );
int vo_inv_trans( int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2, int clipMin, int clipMax );
int vo_fwd_2d( const int16_t *resi, int stride, int w, int h, int bitDepth, int typeHor, int typeVer, int32_t *coef );
int vo_inv_2d( const int32_t *coef, int w, int h, int bitDepth, int typeHor, int typeVer, int16_t *resi, int stride );
void vo_quant( const int32_t *coef, int w, int h, int bitDepth, int qpPer, int qpRem, int isIRAP, int isTS, int32_t *qcoef, int32_t *deltaU,
               int32_t *absSum );
void vo_dequant( const int32_t *qcoef, int w, int h, int bitDepth, int qpPer, int qpRem, int isTS, int32_t *coef );

void vo_remove_high_freq( int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h );
void vo_add_avg( const int16_t *a, int aStride, const int16_t *b, int bStride, int16_t *dst, int dstStride, int w, int h, int bitDepth );
void vo_sobel( int vertical, const int16_t *p, int ps, int32_t *d, int ds, int w, int h );
void vo_equal_coeff( const int16_t *resi, int rs, const int32_t *gx, const int32_t *gy, int ds, int64_t eq[7][7], int w, int h, int b6Param );

void vo_set_search_range( const vo_me_ctx_t *c, int predHor, int predVer, int range, vo_range_t *sr );
void vo_tz_search( const vo_me_ctx_t *c, const vo_tz_job_t *job, vo_me_result_t *res );
void vo_full_search( const vo_me_ctx_t *c, const vo_range_t *sr, vo_me_result_t *res );
void vo_frac_search( const vo_me_ctx_t *c, int intX, int intY, int useHad, int useAltHpelIf, vo_frac_result_t *res );

/* InterSearch::xMotionEstimation: encoder switches shared by a batch, one (PU, list, refIdx) job, its outputs */
typedef struct
{
  int bipredSearchRange;       /* m_bipredSearchRange (cfg BipredSearchRange, 4) */
  int useHadME;                /* HadamardME && !slice.getDisableSATDForRD() */
  int fastInterSearchMode13;   /* FEN 1 or 3: setDistParam subShiftMode 2, else 0 */
  int extendedSettings;        /* MESEARCH_DIAMOND_ENHANCED */
  int firstSearchStop;         /* FastMEAssumingSmootherMVEnabled */
} vo_mest_cfg_t;

typedef struct
{
  const int16_t *org;  int orgStride;        /* origBuf.Y() */
  const int16_t *ref;  int refStride;        /* reconstructed reference luma at the PU position */
  const int16_t *otherPred; int otherStride; /* bi: prediction from the other list (m_tmpPredStorage[1 - list]) */
  int w, h, puX, puY, picW, picH, ctuSize, bitDepth;
  int bi, imv, mvpIdx, numAmvpCand;          /* bBi, cu.imv (0 off, 1 integer, 2 four-sample, 3 half), riMVPIdx, amvpInfo.numCand */
  int mvPredHor, mvPredVer;                  /* rcMvPred, internal 1/16 precision */
  int mvHor, mvVer;                          /* rcMv on entry (bi: start vector) */
  int amvpCand[2][2];                        /* amvpInfo.mvCand */
  unsigned mvpIdxBits[2];                    /* m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] */
  unsigned bits;                             /* ruiBits on entry */
  int searchRange;                           /* m_aaiAdaptSR[list][refIdx] */
  double motionLambda;
  int numExtraStart;                         /* m_uniMvListSize */
  int extraStart[16][2];                     /* m_uniMvList entries for (list, refIdx), newest first, NOT de-duplicated */
  int cachedIntMv;                           /* uni: the block-vector cache holds a vector for this (block, list, refIdx) (CacheBlkInfoCtrl::getMv, :3360-3368): rcMv = mvHor / mvVer
                                                (that integer vector in internal precision), xTZSearch with bFastSettings (:3434-3441) */
  int bcwWeight;                             /* bi: getBcwWeight( cu.BcwIdx, searched list ) in {-2, 3, 5, 10} under a CU-level BCW weight; 0 or 4: the default pair.  The search target is
                                                then removeWeightHighFreq( org, otherPred, w ) (Buffer.h:417-460) and the distortion weight |w| / 8 (xGetMEDistortionWeight :7666-7676) */
} vo_mest_job_t;

typedef struct
{
  int      mvHor, mvVer, mvPredHor, mvPredVer, mvpIdx;   /* rcMv, rcMvPred, riMVPIdx on return */
  unsigned bits;                                         /* ruiBits */
  uint64_t cost;                                         /* ruiCost */
  int      intX, intY;                                   /* integer-stage vector */
  uint64_t intDist;                                      /* ruiCost after the integer stage (distortion without the vector rate) */
} vo_mest_result_t;

void vo_motion_estimation( const vo_mest_cfg_t *cfg, const vo_mest_job_t *job, vo_mest_result_t *res );

/* InterSearch::xEstimateMvPredAMVP with bFilled = true (InterSearch.cpp:3088-3128) and xGetTemplateCost (:3235-3270): SAD of the
 * uni-directional luma prediction at each (clipped) AMVP candidate + getCost( mvpIdxBits ); first candidate with the smallest cost */
void vo_estimate_mvp_amvp( const vo_mest_job_t *job, int *mvpIdx, int *mvPredHor, int *mvPredVer, uint64_t *distBiP );
/* InterSearch::xCheckBestMVP (:3185-3232): in / out predictor, index, bits, cost */
void vo_check_best_mvp( double motionLambda, int imv, int numCand, const int cands[2][2], const unsigned idxBits[2], int mvHor, int mvVer,
                        int *mvPredHor, int *mvPredVer, int *mvpIdx, unsigned *bits, uint64_t *cost );

/* TrQuant::xFwdLfnst / xInvLfnst (TrQuant.cpp:340-527) in place on a W x H coefficient block (stride W): gather (transposed when `transpose`), core
 * multiply with M ([16][trSize] for this mode / index), scatter along the diagonal coefficient scan in 4x4 groups */
void vo_lfnst_tu( int32_t *coef, int w, int h, const int8_t *M, int transpose, int inverse );
void vo_lfnst_scan( int w, int h, int32_t *pos48 );

/* ---- affine motion estimation: InterPrediction::xPredAffineBlk (luma, incl. PROF) and InterSearch::xAffineMotionEstimation ---- */
typedef struct
{
  const int16_t *ref; int refStride;   /* reconstructed reference luma at the PU position (MV 0,0) */
  int w, h, puX, puY, picW, picH, ctuSize, bitDepth;
  int sixParam;                        /* cu.affineType == AFFINEMODEL_6PARAM */
  int interDir;                        /* pu.interDir as isSubblockVectorSpreadOverLimit sees it (1, 2, 3) */
  int profAllowed;                     /* sps.getUsePROF() && !m_skipPROF && !picHeader.getDisProfFlag() (no reference scaling) */
  int profNeedsLargeGrad;              /* m_encOnly && !slice.getCheckLDC(): PROF only beyond the profThres gradient (InterPrediction.cpp:913-914) */
  int profIsBi;                        /* m_isBi */
} vo_affine_pred_t;
void vo_pred_affine_blk( const vo_affine_pred_t *p, const int mv[3][2], int bi, int16_t *dst, int dstStride );

typedef struct
{
  vo_affine_pred_t pred;
  const int16_t *org; int orgStride;          /* origBuf.Y() */
  const int16_t *otherPred; int otherStride;  /* bi: m_tmpPredStorage[1 - list] */
  int bi, imv, useSatd, useAffineType, amvrEncOpt, lowDelayRounds;   /* lowDelayRounds: IntraPeriod == -1 (with AffineAmvrEncOpt: 2 refinement rounds for imv 0) */
  int mvPred[3][2], mv[3][2];
  unsigned bits;
  double lambda;
  uint64_t hevcCost;                          /* m_hevcCost */
} vo_affine_me_job_t;
typedef struct { int mv[3][2]; unsigned bits; uint64_t cost; int iterations, refinements; } vo_affine_me_result_t;
/* InterSearch.cpp:5340-5775 for cu.imv 0 / 1, and 2 without AffineAmvrEncOpt (no xDetermineBestMvp); default BCW weight, no MCTS */
void vo_affine_motion_estimation( const vo_affine_me_job_t *job, vo_affine_me_result_t *res );
void vo_solve_equal( double eq[7][7], int order, double *para );   /* solveEqual, InterSearch.cpp:5215-5284 */

/* ---- symmetric MVD search (SMVD): InterSearch::xGetSymmetricCost (InterSearch.cpp:4341-4391), xSymmetricMotionEstimation with xSymmeticRefineMvSearch
 * (:4393-4518), symmvdCheckBestMvp (:7787-7886) and the SMVD block of predInterSearch (:2656-2790) ---- */
typedef struct
{
  const int16_t *org; int orgStride;            /* origBuf.Y() */
  const int16_t *ref[2]; int refStride[2];      /* reconstructed luma of the searched list's symmetric reference [0] and the mirrored list's [1], at the PU position */
  int w, h, puX, puY, picW, picH, ctuSize, bitDepth;
  int imv, useSatd, clipBiPred, bcwWeightTar;   /* cu.imv; !slice.getDisableSATDForRD(); cfg ClipForBiPredMEEnabled; getBcwWeight( BcwIdx, tarList ): 4 = default */
  int numCand[2]; int cand[2][2][2];            /* AMVP lists [searched / mirrored][i][hor / ver] */
  unsigned mvpIdxBits[2];                       /* m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] */
  double motionLambda;
} vo_smvd_job_t;
typedef struct { int mvCur[2], mvTar[2], predSym[2][2], mvpIdxSym[2]; uint64_t cost; } vo_smvd_result_t;
uint64_t vo_symmetric_cost( const vo_smvd_job_t *job, const int mvCur[2], const int mvTar[2] );
void vo_symmetric_me( const vo_smvd_job_t *job, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], uint64_t *cost );
void vo_symmvd_check_best_mvp( const vo_smvd_job_t *job, const int curMv[2], int skip, int predSym[2][2], int mvpIdxSym[2], uint64_t *bestCost );
void vo_smvd_search( const vo_smvd_job_t *job, int numFixed, int numStart, const int starts[][2], unsigned modeBits, vo_smvd_result_t *res );

/* TrQuant::transformNxN( tu, compID, cQP, &trModes, maxCand ) (TrQuant.cpp:950-1019): which candidates survive the sum |coef| pre-selection */
void vo_mts_select( const int32_t *sumAbs, const uint8_t *mtsIdx, int numCand, int w, int h, int bitDepth, int maxLog2TrDynamicRange, int maxCand, uint8_t *test );

#ifdef __cplusplus
}
#endif
#endif
