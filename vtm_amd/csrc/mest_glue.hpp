// mest_glue.hpp -- the per-row bookkeeping of InterSearch::xMotionEstimation (reference EncoderLib/InterSearch.cpp:3299-3494) as device functions, shared by
//   * the stand-alone glue kernels of mest.hip (mixed batches: one thread per row turns one stage's results into the next stage's job records), and
//   * the FUSED forms of the search kernels (round 4): tz_search_kernel / full_search_sq_kernel / frac_search_*_kernel derive their job records from the
//     xMotionEstimation rows themselves in a prologue (a record in LDS, built by one lane of the job's workgroup) and frac_search_* writes the row's final result in its
//     epilogue -- the one-thread-per-row launches between the searches (amvp_select, mest_prepare, mest_mid, mest_final: 232-byte records moved at HBM speed, and a launch each on
//     the dependent chain of a level / of a CU-level call) disappear for uniform batches.
// One source for both, so that the fused and the stand-alone forms cannot drift apart.
#pragma once
#include "ctx.hpp"

namespace mg
{

__device__ __forceinline__ unsigned eg_bits( int v )
{
  // xGetExpGolombNumberOfBits (RdCost.h:301-313): its `while( t > 128 ) { len += 14; t >>= 7; }` only splits floorLog2( t ) = 7 + floorLog2( t >> 7 ),
  // so the length is 1 + 2 * floorLog2( t ) for every t >= 1 -- no loop
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}
__device__ __forceinline__ unsigned mv_bits( int x, int y, int predHor, int predVer, int costScale, unsigned imvShift )
{
  return eg_bits( ( ( x << costScale ) - predHor ) >> imvShift ) + eg_bits( ( ( y << costScale ) - predVer ) >> imvShift );
}
__device__ __forceinline__ unsigned long long rate( double lambda, unsigned bits ) { return ( unsigned long long ) ( lambda * bits ); }   // RdCost::getCost
__device__ __forceinline__ int prec_down( int v, int rs ) { const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }   // Mv::changePrecision
__device__ __forceinline__ int amvr_shift( int imv ) { return imv == 0 ? 2 : imv == 1 ? 4 : imv == 2 ? 6 : 3; }   // Mv::m_amvrPrecision vs INTERNAL
__device__ __forceinline__ void clip_mv( const vtmhip_pic_params &pic, const vtmhip_me_job &j, int &hor, int &ver )   // clipMvInPic
{
  const int horMax = ( pic.picW + 8 - j.puX - 1 ) << 4, horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  const int verMax = ( pic.picH + 8 - j.puY - 1 ) << 4, verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  hor = min( horMax, max( horMin, hor ) );
  ver = min( verMax, max( verMin, ver ) );
}
__device__ __forceinline__ int sub_shift( const vtmhip_me_cfg &cfg, int w, int h ) { return cfg.fastInterSearchMode13 && h > 8 && w <= 64 ? 1 : 0; }   // RdCost.cpp:289-323, mode 2
__device__ __forceinline__ unsigned imv_shift( int imv ) { return imv == 3 ? 1u : ( unsigned ) imv << 1; }
// CU-level BCW weight of the searched list of a bi job (0: the default pair); the default weight (4 of 8) is normalised to 0
__device__ __forceinline__ int bcw_weight( const vtmhip_me_job &j ) { const int w = j.bi ? VTMHIP_MEJ_BCW_WEIGHT( j.flags ) : 0; return w == 4 ? 0 : w; }

// m_uniMvList entry i (newest first) is kept only if no earlier entry equals it (:3391-3403, :3728-3746).  No local array: a run-time indexed one would live in scratch
// memory, and a kernel with scratch pays for its set-up at every launch (mest_prepare_kernel: 18 us per launch for a handful of rows in the round-3 traces)
__device__ __forceinline__ bool extra_is_first( const vtmhip_me_job &j, int i )
{
  for( int k = 0; k < i; k++ ) if( j.extraStart[k][0] == j.extraStart[i][0] && j.extraStart[k][1] == j.extraStart[i][1] ) return false;
  return true;
}
__device__ __forceinline__ int num_extra( const vtmhip_me_job &j ) { return min( 15, max( 0, j.numExtraStart ) ); }

// xEstimateMvPredAMVP's selection (:3088-3128): the first candidate with the smallest template cost; in place
__device__ __forceinline__ void amvp_select( vtmhip_me_job &j, const unsigned long long *dout2, int addIdxBits, unsigned long long *distBiP )
{
  unsigned long long best = ~0ull;
  int                bestIdx = 0;
  for( int c = 0; c < j.numAmvpCand && c < 2; c++ )
  {
    const unsigned long long cost = dout2[c] + rate( j.motionLambda, j.mvpIdxBits[c] );
    if( best > cost ) { best = cost; bestIdx = c; }
  }
  j.mvPredHor = bestIdx ? j.amvpCand[1][0] : j.amvpCand[0][0]; j.mvPredVer = bestIdx ? j.amvpCand[1][1] : j.amvpCand[0][1];
  j.mvpIdx    = ( uint8_t ) bestIdx;
  if( addIdxBits ) j.bits += bestIdx ? j.mvpIdxBits[1] : j.mvpIdxBits[0];
  if( distBiP ) *distBiP = best;
}

// the xTZSearch job of a uni row (:3434-3447); `slot` / `slotStride`: where the search pattern (the original block) sits.  _scalars: everything but the start-candidate list
// (predH / predV: the row's predictor -- j.mvPredHor / mvPredVer, or the AMVP selection a fused prologue has just made)
__device__ __forceinline__ void make_tz_job_scalars( const vtmhip_me_cfg &cfg, const vtmhip_me_job &j, long slot, int slotStride, int predH, int predV, vtmhip_tz_job &t )
{
  t.orgOff = slot; t.refOff = j.refOff; t.orgStride = slotStride; t.refStride = j.refStride;
  t.puX = j.puX; t.puY = j.puY; t.width = j.width; t.height = j.height; t.subShift = ( int16_t ) sub_shift( cfg, j.width, j.height ); t.imvShift = ( uint8_t ) imv_shift( j.imv );
  t.signedSamples = 0;
  t.predHor = prec_down( predH, 2 ); t.predVer = prec_down( predV, 2 ); t.motionLambda = j.motionLambda;
  const bool cached = ( j.flags & VTMHIP_MEJ_CACHED_INT_MV ) != 0;   // block-vector cache hit (:3360-3368): rcMv = the cached vector, xTZSearch with bFastSettings (:3434-3441)
  t.mvHor = cached ? j.mvHor : predH; t.mvVer = cached ? j.mvVer : predV;   // else rcMv = rcMvPred (:3446)
  t.searchRange = j.searchRange;
  t.extendedSettings = cfg.extendedSettings; t.fastSettings = cached; t.firstSearchStop = cfg.firstSearchStop; t.hasIntMv2Nx2NPred = 0;
  t.intMv2Nx2NPredHor = t.intMv2Nx2NPredVer = 0;
}
__device__ __forceinline__ void make_tz_job( const vtmhip_me_cfg &cfg, const vtmhip_me_job &j, long slot, int slotStride, vtmhip_tz_job &t )
{
  make_tz_job_scalars( cfg, j, slot, slotStride, j.mvPredHor, j.mvPredVer, t );
  int nex = 0;
  const int m = num_extra( j );
  for( int i = 0; i < m; i++ )
    if( extra_is_first( j, i ) ) { t.extraStart[nex][0] = j.extraStart[i][0]; t.extraStart[nex][1] = j.extraStart[i][1]; nex++; }
  t.numExtraStart = nex;
}

// the exhaustive-search job of a bi row around `centerHor / centerVer` (xSetSearchRange + xPatternSearch, :3421-3432)
__device__ __forceinline__ void make_full_job( const vtmhip_me_cfg &cfg, const vtmhip_me_job &j, long slot, int slotStride, int centerHor, int centerVer, vtmhip_full_job &f )
{
  f.orgOff = slot; f.refOff = j.refOff; f.orgStride = slotStride; f.refStride = j.refStride;
  f.puX = j.puX; f.puY = j.puY; f.width = j.width; f.height = j.height; f.subShift = ( int16_t ) sub_shift( cfg, j.width, j.height );
  f.imvShift = ( uint8_t ) imv_shift( j.imv ); f.signedSamples = 1;
  f.predHor = prec_down( j.mvPredHor, 2 ); f.predVer = prec_down( j.mvPredVer, 2 ); f.motionLambda = j.motionLambda; f.centerHor = centerHor; f.centerVer = centerVer;
  f.searchRange = cfg.bipredSearchRange; f.pad = 0;
}

// the xPatternSearchFracDIF job of a row with cu.imv 0 / IMV_HPEL (:3459-3475)
__device__ __forceinline__ void make_frac_job( int useHadME, int bitDepth, const vtmhip_me_job &j, int intX, int intY, long slot, int slotStride, vtmhip_frac_job &q )
{
  q.orgOff = slot; q.refOff = j.refOff; q.orgStride = slotStride; q.refStride = j.refStride; q.width = j.width; q.height = j.height;
  q.intX = ( int16_t ) intX; q.intY = ( int16_t ) intY;
  q.predHor = prec_down( j.mvPredHor, 2 ); q.predVer = prec_down( j.mvPredVer, 2 ); q.motionLambda = j.motionLambda;
  q.useHad = ( uint8_t ) useHadME; q.useAltHpelIf = j.imv == 3; q.imvShift = j.imv == 3; q.bitDepth = ( uint8_t ) bitDepth; q.wideOrg = bcw_weight( j ) != 0;
}

// the rate re-weighting after the fractional search (:3476-3485): the row's result
__device__ __forceinline__ void make_out_frac( const vtmhip_me_job &j, int intX, int intY, unsigned long long intDist, const vtmhip_frac_result &f, vtmhip_me_out &o )
{
  const int    bcw = bcw_weight( j );
  const double fWeight = j.bi ? ( bcw ? fabs( ( double ) bcw / 8.0 ) : 0.5 ) : 1.0;    // xGetMEDistortionWeight (:7666-7676): |getBcwWeight| / g_BcwWeightBase, 0.5 for BCW_DEFAULT
  const double lam = j.motionLambda;
  o.intX = intX; o.intY = intY; o.intDist = intDist;
  const int      qx = ( intX << 2 ) + ( f.halfX << 1 ) + f.qterX, qy = ( intY << 2 ) + ( f.halfY << 1 ) + f.qterY;
  const unsigned mvBits = mv_bits( qx, qy, prec_down( j.mvPredHor, 2 ), prec_down( j.mvPredVer, 2 ), 0, imv_shift( j.imv ) );
  const unsigned bits = j.bits + mvBits;
  o.cost = ( unsigned long long ) ( floor( fWeight * ( ( double ) f.cost - ( double ) rate( lam, mvBits ) ) ) + ( double ) rate( lam, bits ) );   // :3483
  o.mvHor = qx << 2; o.mvVer = qy << 2; o.mvPredHor = j.mvPredHor; o.mvPredVer = j.mvPredVer; o.mvpIdx = j.mvpIdx; o.bits = bits;
}

}   // namespace mg

// What a fused search kernel gets beside (instead of) its own job table.  me == nullptr: not fused (the kernel reads its job table as before).
struct MeFuse
{
  vtmhip_me_job            *me;          // the xMotionEstimation rows of the batch (written only by the AMVP selection)
  const unsigned long long *amvpDout;    // tz: [2 * row + c] template SADs of the AMVP candidates -> xEstimateMvPredAMVP's selection in the prologue (nullptr: rows already chosen)
  unsigned long long       *distBiP;     //     [row] *puiDistBiP (may be nullptr)
  int                       addIdxBits;  //     the chosen predictor's index bits join the row's bits (predInterSearch :2381)
  const vtmhip_me_result   *ires;        // frac: the integer stage's results
  vtmhip_me_out            *out;         // frac: the row's final record, written in the epilogue
  vtmhip_tz_job            *tzSpill;     // tz, split launches: the LDS-built job of a search that goes to the raster kernel is stored here (that kernel and the resume launch read it)
  const int16_t            *biBase;      // (unused by kernels: host bookkeeping)
  vtmhip_me_cfg             cfg;
  int                       bitDepth;
  int                       patIsOther;  // 0: the search pattern is the row's original block (orgOff / orgStride); 1: the caller-made bi-pred target (otherPredOff / otherPredStride)
};
__device__ __forceinline__ long fuse_pat_off( const MeFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredOff : j.orgOff; }
__device__ __forceinline__ int  fuse_pat_stride( const MeFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredStride : j.orgStride; }

// the slim view a fused fractional-search kernel takes (kernel arguments live in scalar registers for the kernel's whole life: only what it needs)
struct FracFuse
{
  const vtmhip_me_job    *me;      // nullptr: not fused
  const vtmhip_me_result *ires;
  vtmhip_me_out          *out;
  int                     useHadME, bitDepth, patIsOther;
};
__device__ __forceinline__ long fuse_pat_off( const FracFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredOff : j.orgOff; }
__device__ __forceinline__ int  fuse_pat_stride( const FracFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredStride : j.orgStride; }

// the view a fused exhaustive-search kernel takes: the bi rows; noStart: rcMv is the search centre (empty m_uniMvList lists), else the kernel evaluates the start candidates
// (rcMv + the distinct m_uniMvList entries, :3377-3420) itself
struct FullFuse
{
  const vtmhip_me_job *me;      // nullptr: not fused
  int                  subShiftMode13, bipredSearchRange, patIsOther, noStart;
};
__device__ __forceinline__ long fuse_pat_off( const FullFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredOff : j.orgOff; }
__device__ __forceinline__ int  fuse_pat_stride( const FullFuse &F, const vtmhip_me_job &j ) { return F.patIsOther ? j.otherPredStride : j.orgStride; }
namespace mg
{
__device__ __forceinline__ void make_full_job( const FullFuse &F, const vtmhip_me_job &j, int centerHor, int centerVer, vtmhip_full_job &f )
{
  vtmhip_me_cfg c; c.fastInterSearchMode13 = ( uint8_t ) F.subShiftMode13; c.bipredSearchRange = F.bipredSearchRange;
  make_full_job( c, j, fuse_pat_off( F, j ), fuse_pat_stride( F, j ), centerHor, centerVer, f );
}
}

