// TEST INFRASTRUCTURE ONLY -- encoder-level drop-in check.
//
// Runs the REAL reference encoder (EncApp / EncLib of VTM 9.3, compiled in place into oracle/_ref/libvtmref.so) and, when a
// libvtmhip.so path is given, installs trampolines into the reference's own dispatch surface exactly where INTEGRATION.md
// section 2 says a maintainer would:
//   RdCost::m_afpDistortFunc[DF_SAD*, DF_HAD*, DF_SSE*, DF_SAD_WITH_MASK]  (RdCost.h:113, RdCost.cpp:125-217)
//   InterpolationFilter::m_filterHor / m_filterVer / m_filterCopy    (InterpolationFilter.h:93-95) of EncLib's InterSearch
//   InterpolationFilter::m_weightedGeoBlk                            (InterpolationFilter.h:99)
//   fastFwdTrans / fastInvTrans                                      (TrQuant.cpp:69-81)
// A hooked call is routed to the device through the C ABI of include/vtmhip.h; its result is what the encoder continues with
// (replace mode), and it is also compared with the reference's own function on the same arguments.  Only every `stride`-th
// call per table slot goes to the device (the first `head` calls of each slot always do) so that a small encode stays within
// a test's time budget; all other calls run the reference function untouched.
//
// The driver below follows the call order of the reference's own main() (App/EncoderApp/encmain.cpp:120-320):
// initROM, EncApp::create, parseCfg, createLib, { encodePrep, encode } until eos, destroyLib, destroy, destroyROM.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>
#include <mutex>

#include "EncoderLib/EncLibCommon.h"
#include "EncApp.h"
#include "CommonLib/RdCost.h"
#include "CommonLib/TrQuant.h"
#include "CommonLib/TrQuant_EMT.h"
#include "CommonLib/InterpolationFilter.h"
#include "CommonLib/Rom.h"
#include "EncoderLib/EncLib.h"
#include "EncoderLib/EncModeCtrl.h"
#include "CommonLib/UnitTools.h"

#include "../include/vtmhip.h"

extern FwdTrans *fastFwdTrans[NUM_TRANS_TYPE][g_numTransformMatrixSizes];
extern InvTrans *fastInvTrans[NUM_TRANS_TYPE][g_numTransformMatrixSizes];

extern "C" {
struct RefEncStats
{
  uint64_t calls[4];      // dist / interpolation / transform / (buffer ops + affine gradients) calls through a trampoline
  uint64_t device[4];     // ... of which were executed on the device
  uint64_t mismatch[4];   // device result != reference result
  uint64_t errors;        // non-zero vtmhip status
  int32_t  firstMismatch[8];
  char     firstError[160]; // vtmhip_last_error() of the first failed call
  // batched hooks, [0]: InterSearch::xMotionEstimation as ONE vtmhip_xMotionEstimation_batch_dev call (hooks B1-B6),
  //                                       [1]: TrQuant::transformNxN( trModes ) = all MTS candidates' forward transforms + the pre-selection (hook B8)
  uint64_t hookCalls[2], hookDevice[2], hookMismatch[2], hookUnsupported[2];
  int32_t  hookFirstMismatch[8];
  uint64_t affineCalls, affineDevice, affineMismatch, affineUnsupported;
  uint64_t lfnstCalls[2], lfnstDevice[2], lfnstMismatch[2];   // TrQuant::xFwdLfnst / xInvLfnst (gather + core multiply + scatter) as vtmhip_lfnst_tu_batch_dev   // InterSearch::xAffineMotionEstimation as one vtmhip_xAffineMotionEstimation_batch_dev call
  uint64_t amvpCalls, amvpDevice, amvpMismatch, amvpUnsupported;   // InterSearch::xEstimateMvPredAMVP's candidate selection as one vtmhip_xEstimateMvPredAMVP_batch_dev call (hook B7)
  uint64_t smvdCalls[3], smvdDevice[3], smvdMismatch[3], smvdUnsupported;   // xGetSymmetricCost / xSymmetricMotionEstimation / symmvdCheckBestMvp as vtmhip_smvd_batch_dev ops
  // InterSearch::predInterSearch as ONE vtmhip_predInterSearch_batch_dev call per CU (oracle/ref_shim_pis.hpp): calls; on the device (or recorded); unsupported (the member
  // runs as it is); skipped (no translational part: checkNonAffine false); served member calls whose arguments differed from the device's glue (the reference's code ran instead)
  uint64_t pisCalls, pisDevice, pisUnsupported, pisSkipped, pisReplayFallback;
  uint64_t pisMismatch[6];      // [0] final decision, [1] xEstimateMvPredAMVP, [2] uni xMotionEstimation, [3] bi xMotionEstimation, [4] SMVD members, [5] argument checks
  int32_t  pisFirstMismatch[8];
  uint64_t pisNs[4];            // wall time: [0] gather + compare, [1] upload + device + download, [2] the member over the tables, [3] the member where the device was not used
  uint64_t affineNs[2];         // xAffineMotionEstimation hook: [0] the reference's member (0 in replace mode), [1] + gather + device
  // the SATD pre-selection of IntraSearch::estIntraPredLumaQT as one vtmhip_intra_cand_cost_batch_dev call per CU (oracle/ref_shim_intra.hpp): [0] pre-selections seen,
  // [1] batched on the device, [2] (unused), [3] served calls refused because the reference's prediction differed from the batch's
  uint64_t intraBatches[4], intraServed, intraMismatch;
  int32_t  intraFirstMismatch[8];
  uint64_t hookThreads;         // encoder threads that made CU-level device calls, each through its own vtmhip context (1 unless the split-parallel build runs --NumSplitThreads > 1)
};
}

namespace
{
struct Api
{
  void *so = nullptr;
  decltype( &vtmhip_create )       create;
  decltype( &vtmhip_destroy )      destroy;
  decltype( &vtmhip_last_error )   last_error;
  decltype( &vtmhip_xGetSAD )      sad;
  decltype( &vtmhip_xGetHADs )     had;
  decltype( &vtmhip_xGetSSE )      sse;
  decltype( &vtmhip_xGetSADwMask ) sadmask;
  decltype( &vtmhip_filterHor )    fhor;
  decltype( &vtmhip_filterVer )    fver;
  decltype( &vtmhip_filterCopy )   fcopy;
  decltype( &vtmhip_weightedGeoBlk ) geo;
  decltype( &vtmhip_fastFwdTrans ) fwd;
  decltype( &vtmhip_fastInvTrans ) inv;
  decltype( &vtmhip_dev_alloc )    dalloc;
  decltype( &vtmhip_dev_free )     dfree;
  decltype( &vtmhip_h2d )          h2d;
  decltype( &vtmhip_d2h )          d2h;
  decltype( &vtmhip_sync )         sync;
  decltype( &vtmhip_add_avg_batch_dev )            addAvg;
  decltype( &vtmhip_remove_high_freq_batch_dev )   rhf;
  decltype( &vtmhip_affine_sobel_batch_dev )       sobel;
  decltype( &vtmhip_affine_equal_coeff_batch_dev ) eqc;
  decltype( &vtmhip_xMotionEstimation_batch_dev )  me;
  decltype( &vtmhip_xT_batch_dev )                 xT;
  decltype( &vtmhip_tu_ts_chain_batch_dev )        tsChain;
  decltype( &vtmhip_mts_select2 )                  mtsSelect;
  decltype( &vtmhip_xAffineMotionEstimation_batch_dev ) affineMe, affineMeBcw;
  decltype( &vtmhip_lfnst_set_tables )             lfnstTables;
  decltype( &vtmhip_lfnst_tu_batch_dev )           lfnstTu;
  decltype( &vtmhip_xEstimateMvPredAMVP_batch_dev ) amvp;
  decltype( &vtmhip_smvd_batch_dev )               smvd;
} A;

vtmhip_ctx  *g_ctx = nullptr;
RefEncStats *g_st  = nullptr;
uint64_t     g_stride = 1, g_head = 0;
unsigned     g_mask = 7;
bool         g_countOnly = false;   // familyMask bit 3: trampolines installed, nothing sent to the device (call census on a CPU-only box)

template<class F> bool sym( F &f, const char *name ) { f = (F) dlsym( A.so, name ); return f != nullptr; }

inline bool sampled( uint64_t &ctr )
{
  const uint64_t n = ctr++;
  if( g_countOnly ) return false;
  return n < g_head || ( n % g_stride ) == 0;
}
// counters the CU-level hooks bump from several encoder threads (ENABLE_SPLIT_PARALLELISM build)
#define ST_ADD( field, v ) __atomic_fetch_add( &g_st->field, ( uint64_t ) ( v ), __ATOMIC_RELAXED )
#define ST_INC( field ) ST_ADD( field, 1 )
inline void note_error( vtmhip_ctx *ctx = nullptr )
{
  if( ST_INC( errors ) == 0 ) { strncpy( g_st->firstError, A.last_error( ctx ? ctx : g_ctx ), sizeof( g_st->firstError ) - 1 ); }
}
inline void note_mismatch( int family, int a, int b, int c, int d, long long ref, long long dev )
{
  if( g_st->mismatch[0] + g_st->mismatch[1] + g_st->mismatch[2] + g_st->mismatch[3] == 0 )
  {
    const int32_t v[8] = { family, a, b, c, d, (int32_t) ref, (int32_t) dev, 0 };
    memcpy( g_st->firstMismatch, v, sizeof( v ) );
  }
  g_st->mismatch[family]++;
}

// ---- distortion ------------------------------------------------------------------------------------------------------------
bool intraServe( const DistParam &p, int kind, FpDistFunc orig, Distortion &out );      // oracle/ref_shim_intra.hpp
extern bool g_intraLive;
FpDistFunc g_distOrig[DF_TOTAL_FUNCTIONS];
uint64_t   g_distCtr[DF_TOTAL_FUNCTIONS];

template<int IDX> Distortion distTramp( const DistParam &p )
{
  constexpr int kind = ( IDX >= DF_HAD && IDX <= DF_HAD16N ) ? 1 : ( IDX >= DF_SSE && IDX <= DF_SSE16N ) ? 2 : IDX == DF_SAD_WITH_MASK ? 3 : 0;
  g_st->calls[0]++;
  if( g_intraLive && kind <= 1 && !p.applyWeight && !p.useMR && p.step == 1 && p.mask == nullptr )      // the batched intra pre-selection serves its own calls
  {
    Distortion d;
    if( intraServe( p, kind, g_distOrig[IDX], d ) ) return d;
  }
  // same guards as the x86 table entries (x86/RdCostX86.h:213,344,2065,2157) -- what the device path does not cover stays on the host
  if( p.applyWeight || p.useMR || p.step != 1 || ( ( p.mask != nullptr ) != ( kind == 3 ) ) || p.bitDepth > 12 || ( ( kind == 1 || kind == 2 ) && p.subShift != 0 )
      || ( kind == 3 && p.stepX != 1 && p.stepX != -1 ) || !sampled( g_distCtr[IDX] ) )
  {
    return g_distOrig[IDX]( p );
  }
  const Distortion ref = g_distOrig[IDX]( p );
  uint64_t dev = 0;
  const int w = p.org.width, h = p.org.height;
  const int st = kind == 0 ? A.sad( g_ctx, p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, w, h, p.subShift, &dev )
               : kind == 1 ? A.had( g_ctx, p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, w, h, &dev )
               : kind == 2 ? A.sse( g_ctx, p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, w, h, &dev )
                           : A.sadmask( g_ctx, p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, w, h, p.subShift, p.mask, p.maskStride, p.stepX, p.maskStride2, &dev );
  if( st != VTMHIP_OK ) { note_error(); return ref; }
  g_st->device[0]++;
  // the scalar SAD may stop early once it exceeds maximumDistortionForEarlyExit (RdCost.cpp:516-519); callers only compare '<'
  const bool early = ref > p.maximumDistortionForEarlyExit && dev >= ref;
  if( dev != ref && !early ) note_mismatch( 0, IDX, w, h, p.subShift, (long long) ref, (long long) dev );
  return early ? ref : Distortion( dev );
}
template<int... I> void installDist( std::integer_sequence<int, I...> )
{
  static const FpDistFunc t[] = { distTramp<I>... };
  for( int i = 0; i < DF_TOTAL_FUNCTIONS; i++ )
  {
    g_distOrig[i] = RdCost::m_afpDistortFunc[i];
    const bool hook = ( i >= DF_SSE && i <= DF_SSE16N ) || ( i >= DF_SAD && i <= DF_SAD16N ) || ( i >= DF_HAD && i <= DF_HAD16N ) || ( i >= DF_SAD12 && i <= DF_SAD48 ) || i == DF_SAD_WITH_MASK;
    if( hook && ( g_mask & 1 ) ) RdCost::m_afpDistortFunc[i] = t[i];
  }
}
void restoreDist() { for( int i = 0; i < DF_TOTAL_FUNCTIONS; i++ ) if( g_distOrig[i] ) RdCost::m_afpDistortFunc[i] = g_distOrig[i]; }

// ---- interpolation ---------------------------------------------------------------------------------------------------------
typedef void ( *IfFn )( const ClpRng &, Pel const *, int, Pel *, int, int, int, TFilterCoeff const *, bool );
typedef void ( *IfCopyFn )( const ClpRng &, Pel const *, int, Pel *, int, int, int, bool );
IfFn     g_ifOrig[2][3][2][2];
IfCopyFn g_ifCopyOrig[2][2];
uint64_t g_ifCtr[2][3][2][2], g_ifCopyCtr[2][2];
std::vector<Pel> g_ifTmp;

inline void compare_block( int family, int a, int b, const Pel *ref, int rs, const Pel *dev, int ds, int w, int h )
{
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      if( ref[y * rs + x] != dev[y * ds + x] ) { note_mismatch( family, a, b, x, y, ref[y * rs + x], dev[y * ds + x] ); return; }
}
template<int VER, int TI, int FIRST, int LAST>
void ifTramp( const ClpRng &c, Pel const *src, int ss, Pel *dst, int ds, int w, int h, TFilterCoeff const *coeff, bool biMC )
{
  g_st->calls[1]++;
  g_ifOrig[VER][TI][FIRST][LAST]( c, src, ss, dst, ds, w, h, coeff, biMC );
  if( !sampled( g_ifCtr[VER][TI][FIRST][LAST] ) ) return;
  constexpr int taps = TI == 0 ? 8 : TI == 1 ? 4 : 2;
  g_ifTmp.resize( size_t( w ) * h );
  const int st = ( VER ? A.fver : A.fhor )( g_ctx, taps, FIRST, LAST, src, ss, g_ifTmp.data(), w, w, h, coeff, c.bd, c.min, c.max, biMC );
  if( st != VTMHIP_OK ) { note_error(); return; }
  g_st->device[1]++;
  compare_block( 1, VER * 100 + TI * 10 + FIRST * 2 + LAST, w * 1000 + h, dst, ds, g_ifTmp.data(), w, w, h );
  for( int y = 0; y < h; y++ ) memcpy( dst + y * ds, g_ifTmp.data() + y * w, sizeof( Pel ) * w );   // the encoder continues with the device output
}
template<int FIRST, int LAST>
void ifCopyTramp( const ClpRng &c, Pel const *src, int ss, Pel *dst, int ds, int w, int h, bool biMC )
{
  g_st->calls[1]++;
  g_ifCopyOrig[FIRST][LAST]( c, src, ss, dst, ds, w, h, biMC );
  if( !sampled( g_ifCopyCtr[FIRST][LAST] ) ) return;
  g_ifTmp.resize( size_t( w ) * h );
  const int st = A.fcopy( g_ctx, FIRST, LAST, src, ss, g_ifTmp.data(), w, w, h, c.bd, c.min, c.max, biMC );
  if( st != VTMHIP_OK ) { note_error(); return; }
  g_st->device[1]++;
  compare_block( 1, 900 + FIRST * 2 + LAST, w * 1000 + h, dst, ds, g_ifTmp.data(), w, w, h );
  for( int y = 0; y < h; y++ ) memcpy( dst + y * ds, g_ifTmp.data() + y * w, sizeof( Pel ) * w );
}
// m_weightedGeoBlk (InterpolationFilter.h:99): the trampoline derives the weight walk from the reference's GEO tables the way xWeightedGeoBlk does
// (InterpolationFilter.cpp:925-945) and hands plain pointers to the C ABI
typedef void ( *GeoFn )( const PredictionUnit &, const uint32_t, const uint32_t, const ComponentID, const uint8_t, PelUnitBuf &, PelUnitBuf &, PelUnitBuf & );
GeoFn    g_geoOrig = nullptr;
uint64_t g_geoCtr  = 0;
void geoTramp( const PredictionUnit &pu, const uint32_t width, const uint32_t height, const ComponentID compIdx, const uint8_t splitDir, PelUnitBuf &predDst,
               PelUnitBuf &predSrc0, PelUnitBuf &predSrc1 )
{
  g_st->calls[1]++;
  g_geoOrig( pu, width, height, compIdx, splitDir, predDst, predSrc0, predSrc1 );
  if( !sampled( g_geoCtr ) ) return;
  const ClpRng &c = pu.cu->slice->clpRngs().comp[compIdx];
  const int M = GEO_WEIGHT_MASK_SIZE, angle = g_GeoParams[splitDir][0], mirror = g_angle2mirror[angle];
  const int wIdx = floorLog2( pu.lwidth() ) - GEO_MIN_CU_LOG2, hIdx = floorLog2( pu.lheight() ) - GEO_MIN_CU_LOG2;
  const int ox = g_weightOffset[splitDir][hIdx][wIdx][0], oy = g_weightOffset[splitDir][hIdx][wIdx][1];
  const int scX = getComponentScaleX( compIdx, pu.chromaFormat ), scY = getComponentScaleY( compIdx, pu.chromaFormat );
  const int16_t *weight = g_globalGeoWeights[g_angle2mask[angle]] + ( mirror == 2 ? ( M - 1 - oy ) * M + ox : mirror == 1 ? oy * M + ( M - 1 - ox ) : oy * M + ox );
  const int stepX = ( mirror == 1 ? -1 : 1 ) * ( 1 << scX ), wStride = ( mirror == 2 ? -M : M ) * ( 1 << scY );
  const PelBuf &d = predDst.get( compIdx ), &a = predSrc0.get( compIdx ), &b = predSrc1.get( compIdx );
  g_ifTmp.resize( size_t( width ) * height );
  const int st = A.geo( g_ctx, a.buf, a.stride, b.buf, b.stride, g_ifTmp.data(), width, width, height, weight, stepX, wStride, c.bd, c.min, c.max );
  if( st != VTMHIP_OK ) { note_error(); return; }
  g_st->device[1]++;
  compare_block( 1, 800 + compIdx, width * 1000 + height, d.buf, d.stride, g_ifTmp.data(), width, width, height );
  for( uint32_t y = 0; y < height; y++ ) memcpy( d.buf + y * d.stride, g_ifTmp.data() + y * width, sizeof( Pel ) * width );
}
template<int VER, int TI> void installIfSlot( InterpolationFilter &f )
{
  auto &tab = VER ? f.m_filterVer : f.m_filterHor;
  g_ifOrig[VER][TI][0][0] = tab[TI][0][0]; tab[TI][0][0] = ifTramp<VER, TI, 0, 0>;
  g_ifOrig[VER][TI][0][1] = tab[TI][0][1]; tab[TI][0][1] = ifTramp<VER, TI, 0, 1>;
  g_ifOrig[VER][TI][1][0] = tab[TI][1][0]; tab[TI][1][0] = ifTramp<VER, TI, 1, 0>;
  g_ifOrig[VER][TI][1][1] = tab[TI][1][1]; tab[TI][1][1] = ifTramp<VER, TI, 1, 1>;
}
void installIf( InterpolationFilter &f )
{
  installIfSlot<0, 0>( f ); installIfSlot<0, 1>( f ); installIfSlot<0, 2>( f );
  installIfSlot<1, 0>( f ); installIfSlot<1, 1>( f ); installIfSlot<1, 2>( f );
  g_ifCopyOrig[0][0] = f.m_filterCopy[0][0]; f.m_filterCopy[0][0] = ifCopyTramp<0, 0>;
  g_ifCopyOrig[0][1] = f.m_filterCopy[0][1]; f.m_filterCopy[0][1] = ifCopyTramp<0, 1>;
  g_ifCopyOrig[1][0] = f.m_filterCopy[1][0]; f.m_filterCopy[1][0] = ifCopyTramp<1, 0>;
  g_ifCopyOrig[1][1] = f.m_filterCopy[1][1]; f.m_filterCopy[1][1] = ifCopyTramp<1, 1>;
  g_geoOrig = f.m_weightedGeoBlk; f.m_weightedGeoBlk = geoTramp;
}

// ---- transforms ------------------------------------------------------------------------------------------------------------
FwdTrans *g_fwdOrig[NUM_TRANS_TYPE][g_numTransformMatrixSizes];
InvTrans *g_invOrig[NUM_TRANS_TYPE][g_numTransformMatrixSizes];
uint64_t  g_trCtr[2][NUM_TRANS_TYPE][g_numTransformMatrixSizes];
std::vector<TCoeff> g_trTmp;

template<int T, int L> void fwdTramp( const TCoeff *src, TCoeff *dst, int shift, int line, int skip1, int skip2 )
{
  g_st->calls[2]++;
  g_fwdOrig[T][L]( src, dst, shift, line, skip1, skip2 );
  if( !sampled( g_trCtr[0][T][L] ) ) return;
  constexpr int n = 2 << L;
  g_trTmp.assign( size_t( n ) * line, 0 );
  if( A.fwd( g_ctx, T, n, src, g_trTmp.data(), shift, line, skip1, skip2 ) != VTMHIP_OK ) { note_error(); return; }
  g_st->device[2]++;
  // rows [n - skip2, n) of the transposed output are left to the caller's zero-out in some of the reference's butterflies; compare
  // (and hand on) what the reference function defines: coefficient rows k < n - skip2, columns j < line - skip1
  const int rows = n - skip2, cols = line - skip1;
  for( int k = 0; k < rows; k++ )
    for( int j = 0; j < cols; j++ )
      if( dst[k * line + j] != g_trTmp[k * line + j] ) { note_mismatch( 2, T, n, line, k * line + j, dst[k * line + j], g_trTmp[k * line + j] ); k = rows; break; }
  for( int k = 0; k < rows; k++ ) memcpy( dst + k * line, g_trTmp.data() + k * line, sizeof( TCoeff ) * cols );
}
template<int T, int L> void invTramp( const TCoeff *src, TCoeff *dst, int shift, int line, int skip1, int skip2, const TCoeff lo, const TCoeff hi )
{
  g_st->calls[2]++;
  g_invOrig[T][L]( src, dst, shift, line, skip1, skip2, lo, hi );
  if( !sampled( g_trCtr[1][T][L] ) ) return;
  constexpr int n = 2 << L;
  g_trTmp.assign( size_t( n ) * line, 0 );
  if( A.inv( g_ctx, T, n, src, g_trTmp.data(), shift, line, skip1, skip2, lo, hi ) != VTMHIP_OK ) { note_error(); return; }
  g_st->device[2]++;
  const int rows = line - skip1;                               // output rows i < line - skip1, n samples each
  for( int i = 0; i < rows * n; i++ )
    if( dst[i] != g_trTmp[i] ) { note_mismatch( 2, 10 + T, n, line, i, dst[i], g_trTmp[i] ); break; }
  memcpy( dst, g_trTmp.data(), sizeof( TCoeff ) * rows * n );
}
template<int T, int L> void installTrSlot()
{
  g_fwdOrig[T][L] = fastFwdTrans[T][L]; if( fastFwdTrans[T][L] ) fastFwdTrans[T][L] = fwdTramp<T, L>;
  g_invOrig[T][L] = fastInvTrans[T][L]; if( fastInvTrans[T][L] ) fastInvTrans[T][L] = invTramp<T, L>;
}
template<int T> void installTrType() { installTrSlot<T, 0>(); installTrSlot<T, 1>(); installTrSlot<T, 2>(); installTrSlot<T, 3>(); installTrSlot<T, 4>(); installTrSlot<T, 5>(); }
void installTr() { installTrType<DCT2>(); installTrType<DCT8>(); installTrType<DST7>(); }
void restoreTr()
{
  for( int t = 0; t < NUM_TRANS_TYPE; t++ )
    for( int l = 0; l < g_numTransformMatrixSizes; l++ )
    {
      if( g_fwdOrig[t][l] ) fastFwdTrans[t][l] = g_fwdOrig[t][l];
      if( g_invOrig[t][l] ) fastInvTrans[t][l] = g_invOrig[t][l];
    }
}
// ---- buffer ops (g_pelBufOP.addAvg4/8, removeHighFreq4/8: Buffer.h:64-81) and affine gradients (AffineGradientSearch.h:50-54) -----------
// These have no host-pointer entry in the C ABI: the trampolines drive the batched device calls with a batch of one over device
// buffers of the context (vtmhip_dev_alloc / h2d / d2h), which is what hook B5 / the affine hook of INTEGRATION.md section 3 do.
constexpr int AUX_MAX = 128 * 128;
int16_t *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;
int32_t *d_deriv = nullptr;
int64_t *d_eq = nullptr;
void    *d_job = nullptr;
std::vector<Pel> g_auxA, g_auxB, g_auxC;
uint64_t g_auxCtr[8];
decltype( PelBufferOps::addAvg4 )         g_addAvgOrig[2];
decltype( PelBufferOps::removeHighFreq4 ) g_rhfOrig[2];
decltype( AffineGradientSearch::m_HorizontalSobelFilter ) g_sobelOrig[2];
decltype( AffineGradientSearch::m_EqualCoeffComputer )    g_eqOrig;
AffineGradientSearch *g_affine = nullptr;

bool auxAlloc()
{
  return A.dalloc( g_ctx, AUX_MAX * 2, (void **) &d_a ) == VTMHIP_OK && A.dalloc( g_ctx, AUX_MAX * 2, (void **) &d_b ) == VTMHIP_OK
      && A.dalloc( g_ctx, AUX_MAX * 2, (void **) &d_c ) == VTMHIP_OK && A.dalloc( g_ctx, AUX_MAX * 8, (void **) &d_deriv ) == VTMHIP_OK
      && A.dalloc( g_ctx, 49 * 8, (void **) &d_eq ) == VTMHIP_OK && A.dalloc( g_ctx, 256, &d_job ) == VTMHIP_OK;
}
void pack( std::vector<Pel> &v, const Pel *src, int stride, int w, int h )
{
  v.resize( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( v.data() + size_t( y ) * w, src + ptrdiff_t( y ) * stride, sizeof( Pel ) * w );
}
// dst = op( a, b ) for one w x h block, compact strides on the device
bool pelopOnDevice( int op, const Pel *a, int as, const Pel *b, int bs, int w, int h, int bd )
{
  pack( g_auxA, a, as, w, h ); pack( g_auxB, b, bs, w, h );
  vtmhip_pelop_job j; memset( &j, 0, sizeof( j ) );
  j.aStride = j.bStride = j.dstStride = w; j.width = (int16_t) w; j.height = (int16_t) h; j.bitDepth = (uint8_t) bd;
  g_auxC.assign( size_t( w ) * h, 0 );
  bool ok = A.h2d( g_ctx, d_a, g_auxA.data(), g_auxA.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_b, g_auxB.data(), g_auxB.size() * 2 ) == VTMHIP_OK
         && A.h2d( g_ctx, d_job, &j, sizeof( j ) ) == VTMHIP_OK;
  ok = ok && ( op ? A.addAvg( g_ctx, d_a, d_b, d_c, (const vtmhip_pelop_job *) d_job, 1 ) : A.rhf( g_ctx, d_a, d_b, d_c, (const vtmhip_pelop_job *) d_job, 1 ) ) == VTMHIP_OK;
  return ok && A.d2h( g_ctx, g_auxC.data(), d_c, g_auxC.size() * 2 ) == VTMHIP_OK;
}
template<int V> void addAvgTramp( const Pel *s0, int s0s, const Pel *s1, int s1s, Pel *dst, int ds, int w, int h, int shift, int offset, const ClpRng &c )
{
  g_st->calls[3]++;
  g_addAvgOrig[V]( s0, s0s, s1, s1s, dst, ds, w, h, shift, offset, c );
  const int headRoom = std::max( 2, 14 - c.bd );
  // the device op derives shift / offset / clip range from the bit depth (PelBuf::addAvg, Buffer.cpp:467-507); anything else stays on the host
  if( w * h > AUX_MAX || shift != headRoom + 1 || offset != ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS || c.min != 0 || c.max != ( 1 << c.bd ) - 1 || !sampled( g_auxCtr[V] ) ) return;
  if( !pelopOnDevice( 1, s0, s0s, s1, s1s, w, h, c.bd ) ) { note_error(); return; }
  g_st->device[3]++;
  compare_block( 3, V, w * 1000 + h, dst, ds, g_auxC.data(), w, w, h );
  for( int y = 0; y < h; y++ ) memcpy( dst + ptrdiff_t( y ) * ds, g_auxC.data() + size_t( y ) * w, sizeof( Pel ) * w );
}
template<int V> void rhfTramp( Pel *s0, int s0s, const Pel *s1, int s1s, int w, int h )
{
  g_st->calls[3]++;
  if( w * h > AUX_MAX || !sampled( g_auxCtr[2 + V] ) ) { g_rhfOrig[V]( s0, s0s, s1, s1s, w, h ); return; }
  const bool ok = pelopOnDevice( 0, s0, s0s, s1, s1s, w, h, 10 );      // in-place op: run the device first, on the untouched input
  g_rhfOrig[V]( s0, s0s, s1, s1s, w, h );
  if( !ok ) { note_error(); return; }
  g_st->device[3]++;
  compare_block( 3, 2 + V, w * 1000 + h, s0, s0s, g_auxC.data(), w, w, h );
  for( int y = 0; y < h; y++ ) memcpy( s0 + ptrdiff_t( y ) * s0s, g_auxC.data() + size_t( y ) * w, sizeof( Pel ) * w );
}
template<int V> void sobelTramp( Pel *const pred, const int ps, int *const deriv, const int ds, const int w, const int h )
{
  g_st->calls[3]++;
  g_sobelOrig[V]( pred, ps, deriv, ds, w, h );
  if( w * h > AUX_MAX || !sampled( g_auxCtr[4 + V] ) ) return;
  pack( g_auxA, pred, ps, w, h );
  vtmhip_affine_job j; memset( &j, 0, sizeof( j ) );
  j.derivHOff = 0; j.derivVOff = AUX_MAX; j.predStride = w; j.resiStride = w; j.derivStride = w; j.width = (int16_t) w; j.height = (int16_t) h;
  std::vector<int> out( size_t( w ) * h );
  const bool ok = A.h2d( g_ctx, d_a, g_auxA.data(), g_auxA.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_job, &j, sizeof( j ) ) == VTMHIP_OK
               && A.sobel( g_ctx, d_a, d_deriv, (const vtmhip_affine_job *) d_job, 1 ) == VTMHIP_OK
               && A.d2h( g_ctx, out.data(), d_deriv + ( V ? AUX_MAX : 0 ), out.size() * 4 ) == VTMHIP_OK;
  if( !ok ) { note_error(); return; }
  g_st->device[3]++;
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      if( deriv[y * ds + x] != out[y * w + x] ) { note_mismatch( 3, 4 + V, w * 1000 + h, x, y, deriv[y * ds + x], out[y * w + x] ); y = h; break; }
  for( int y = 0; y < h; y++ ) memcpy( deriv + ptrdiff_t( y ) * ds, out.data() + size_t( y ) * w, sizeof( int ) * w );
}
void eqTramp( Pel *resi, int rs, int **ppDeriv, int ds, int64_t ( *eq )[7], int w, int h, bool b6 )
{
  g_st->calls[3]++;
  int64_t before[7][7]; memcpy( before, eq, sizeof( before ) );
  g_eqOrig( resi, rs, ppDeriv, ds, eq, w, h, b6 );
  if( w * h > AUX_MAX || ds != w || !sampled( g_auxCtr[6] ) ) return;
  pack( g_auxA, resi, rs, w, h );
  vtmhip_affine_job j; memset( &j, 0, sizeof( j ) );
  j.derivHOff = 0; j.derivVOff = AUX_MAX; j.predStride = w; j.resiStride = w; j.derivStride = w; j.width = (int16_t) w; j.height = (int16_t) h; j.sixParam = b6;
  int64_t dev[7][7];
  const bool ok = A.h2d( g_ctx, d_a, g_auxA.data(), g_auxA.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_job, &j, sizeof( j ) ) == VTMHIP_OK
               && A.h2d( g_ctx, d_deriv, ppDeriv[0], size_t( w ) * h * 4 ) == VTMHIP_OK && A.h2d( g_ctx, d_deriv + AUX_MAX, ppDeriv[1], size_t( w ) * h * 4 ) == VTMHIP_OK
               && A.h2d( g_ctx, d_eq, before, sizeof( before ) ) == VTMHIP_OK          // the device call accumulates, like the reference
               && A.eqc( g_ctx, d_a, d_deriv, (const vtmhip_affine_job *) d_job, 1, d_eq ) == VTMHIP_OK && A.d2h( g_ctx, dev, d_eq, sizeof( dev ) ) == VTMHIP_OK;
  if( !ok ) { note_error(); return; }
  g_st->device[3]++;
  const int np = b6 ? 6 : 4;
  for( int r = 1; r <= np; r++ )
    for( int c = 0; c <= np; c++ )
      if( dev[r][c] != eq[r][c] ) { note_mismatch( 3, 6, w * 1000 + h, r, c, (long long) eq[r][c], (long long) dev[r][c] ); r = np + 1; break; }
  for( int r = 1; r <= np; r++ ) memcpy( &eq[r][0], &dev[r][0], sizeof( int64_t ) * ( np + 1 ) );
}
void installAux( AffineGradientSearch &ag )
{
  g_addAvgOrig[0] = g_pelBufOP.addAvg4; g_pelBufOP.addAvg4 = addAvgTramp<0>;
  g_addAvgOrig[1] = g_pelBufOP.addAvg8; g_pelBufOP.addAvg8 = addAvgTramp<1>;
  g_rhfOrig[0] = g_pelBufOP.removeHighFreq4; g_pelBufOP.removeHighFreq4 = rhfTramp<0>;
  g_rhfOrig[1] = g_pelBufOP.removeHighFreq8; g_pelBufOP.removeHighFreq8 = rhfTramp<1>;
  g_affine = &ag;
  g_sobelOrig[0] = ag.m_HorizontalSobelFilter; ag.m_HorizontalSobelFilter = sobelTramp<0>;
  g_sobelOrig[1] = ag.m_VerticalSobelFilter;   ag.m_VerticalSobelFilter = sobelTramp<1>;
  g_eqOrig = ag.m_EqualCoeffComputer; ag.m_EqualCoeffComputer = eqTramp;
}
void restoreAux()
{
  if( !g_affine ) return;
  g_pelBufOP.addAvg4 = g_addAvgOrig[0]; g_pelBufOP.addAvg8 = g_addAvgOrig[1];
  g_pelBufOP.removeHighFreq4 = g_rhfOrig[0]; g_pelBufOP.removeHighFreq8 = g_rhfOrig[1];
  g_affine = nullptr;
}

// ---- batched hooks: whole member functions on the device ------------------------------------------------------------------------
// The strong definitions below replace the (weakened) reference symbols at link time (oracle/Makefile.ref); without an installed hook they
// just run the reference's code.
}   // namespace
extern "C" void vtmref_orig_xMotionEstimation( InterSearch *, PredictionUnit &, PelUnitBuf &, RefPicList, Mv &, int, Mv &, int &, uint32_t &, Distortion &, const AMVPInfo &, bool );
extern "C" void vtmref_orig_xAffineMotionEstimation( InterSearch *, PredictionUnit &, PelUnitBuf &, RefPicList, Mv *, int, Mv *, uint32_t &, Distortion &, int &, const AffineAMVPInfo &, bool );
extern "C" void vtmref_orig_xEstimateMvPredAMVP( InterSearch *, PredictionUnit &, PelUnitBuf &, RefPicList, int, Mv &, AMVPInfo &, bool, Distortion * );
extern "C" Distortion vtmref_orig_xGetSymmetricCost( InterSearch *, PredictionUnit &, PelUnitBuf &, RefPicList, const MvField &, MvField &, int );
extern "C" void vtmref_orig_xSymmetricMotionEstimation( InterSearch *, PredictionUnit &, PelUnitBuf &, Mv &, Mv &, RefPicList, MvField &, MvField &, Distortion &, int );
extern "C" void vtmref_orig_symmvdCheckBestMvp( InterSearch *, PredictionUnit &, PelUnitBuf &, Mv, RefPicList, AMVPInfo ( * )[33], int32_t, Mv *, int32_t *, Distortion &, bool );
extern "C" void vtmref_orig_xFwdLfnst( TrQuant *, const TransformUnit &, const ComponentID, const bool );
extern "C" void vtmref_orig_xInvLfnst( TrQuant *, const TransformUnit &, const ComponentID );
extern "C" void vtmref_orig_transformNxN_select( TrQuant *, TransformUnit &, const ComponentID &, const QpParam &, std::vector<TrMode> *, const int );
extern "C" void vtmref_orig_predInterSearch( InterSearch *, CodingUnit &, Partitioner & );
extern "C" void vtmref_orig_initIntraPatternChType( IntraPrediction *, const CodingUnit &, const CompArea &, const bool );
namespace
{
bool     g_hookMe = false, g_hookMts = false, g_hookAffine = false, g_hookLfnst = false, g_hookAmvp = false, g_hookSmvd = false;
uint64_t g_amvpCtr = 0, g_smvdCtr[3] = { 0, 0, 0 };
uint64_t g_lfnstCtr[2] = { 0, 0 };
uint64_t g_affineCtr = 0, g_affineNs[2] = { 0, 0 };   // [0] the reference's member, [1] + the device call
bool     g_pisReplace = false;
inline uint64_t nowNs() { return ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
uint64_t g_hookCtr[2] = { 0, 0 }, g_hookStride = 0;   // VTMREF_HOOK_STRIDE: the hooks' own sampling stride (default: the tables' stride)
inline bool hookSampled( uint64_t &ctr )
{
  const uint64_t n = __atomic_fetch_add( &ctr, 1, __ATOMIC_RELAXED );
  if( g_countOnly ) return false;
  return n < g_head || ( n % ( g_hookStride ? g_hookStride : g_stride ) ) == 0;
}

// ---- per-thread state of the CU-level hooks -------------------------------------------------------------------------------------------------
// The ENABLE_SPLIT_PARALLELISM build (oracle/Makefile.ref SP=1) runs up to six EncCu / InterSearch instances in OpenMP threads (EncCu::xCompressCUParallel): every encoder
// thread gets its own vtmhip context (own stream, own workspaces: "one context per encoder thread", include/vtmhip.h), its own page-locked slot block and device blocks, so
// that the threads' CU-level calls are in flight on the device TOGETHER.  The single-threaded build has exactly one.
struct HookThread
{
  vtmhip_ctx *ctx = nullptr;
  void       *pisHost = nullptr;                 // PisSlots, page-locked (vtmhip_host_alloc)
  char       *d_pis = nullptr;
  int16_t    *d_pisOrgBi = nullptr, *d_org = nullptr, *d_other = nullptr;   // affine hook: the original block, the other list's prediction
  void       *d_job = nullptr, *d_out = nullptr;
  bool        ok = false, ownsCtx = false;
  unsigned    gen = 0;
};
unsigned                  g_hkGen = 1;
std::mutex                g_hkMutex;
std::vector<HookThread *> g_hkAll;
thread_local HookThread  *t_hk = nullptr;
HookThread *hookThread();      // (defined behind ref_shim_pis.hpp: it sizes the slot block)

struct RefPlane { const Picture *pic; int poc; int16_t *dev, *alloc; size_t samples; int stride, margin; };
constexpr size_t PLANE_PAD = VTMHIP_PLANE_SLACK;   // samples of slack before and after an uploaded plane (include/vtmhip.h: "reference planes under d_refBase: what must be readable")
std::vector<RefPlane> g_planes;
int16_t *d_hOrg = nullptr, *d_hOther = nullptr;
void    *d_hJob = nullptr, *d_hOut = nullptr;
int32_t *d_hCoef = nullptr, *d_hSum = nullptr;
constexpr size_t HOOK_BLK = 128 * 128;

bool hookAlloc()
{
  return A.dalloc( g_ctx, HOOK_BLK * 2, (void **) &d_hOrg ) == VTMHIP_OK && A.dalloc( g_ctx, HOOK_BLK * 2, (void **) &d_hOther ) == VTMHIP_OK
      && A.dalloc( g_ctx, 4096, &d_hJob ) == VTMHIP_OK && A.dalloc( g_ctx, 4096, &d_hOut ) == VTMHIP_OK
      && A.dalloc( g_ctx, 8 * 64 * 64 * 4, (void **) &d_hCoef ) == VTMHIP_OK && A.dalloc( g_ctx, 256, (void **) &d_hSum ) == VTMHIP_OK;
}

// the reconstructed reference picture, border included, uploaded once per (picture buffer, POC).  Returns a copy under the table's lock (encoder threads share the table; reference
// pictures only change between pictures, i.e. outside the parallel regions); ok == false: allocation / upload failed
struct RefPlaneRef { RefPlane p; bool ok; const RefPlane *operator->() const { return &p; } explicit operator bool() const { return ok; } };
RefPlaneRef refPlaneOf( vtmhip_ctx *ctx, const Picture *pic )
{
  std::lock_guard<std::mutex> lock( g_hkMutex );
  for( const RefPlane &p : g_planes ) if( p.pic == pic && p.poc == pic->getPOC() ) return { p, true };
  const CPelBuf y = pic->getRecoBuf( COMPONENT_Y );
  const int     m = pic->margin;
  RefPlane      r; r.pic = pic; r.poc = pic->getPOC(); r.stride = y.stride; r.margin = m; r.dev = r.alloc = nullptr;
  r.samples = size_t( y.height + 2 * m ) * y.stride;
  for( RefPlane &p : g_planes ) if( p.pic == pic ) { A.dfree( ctx, p.alloc ); p = g_planes.back(); g_planes.pop_back(); break; }   // the buffer now holds another picture
  if( A.dalloc( ctx, ( r.samples + 2 * PLANE_PAD ) * 2, (void **) &r.alloc ) != VTMHIP_OK ) return { r, false };
  r.dev = r.alloc + PLANE_PAD;
  // rows -margin .. height + margin - 1 of the plane; the last row is copied only up to its last sample (the allocation ends there); complete before any other thread's stream reads it
  if( A.h2d( ctx, r.dev, y.buf - ptrdiff_t( m ) * y.stride - m, ( r.samples - size_t( y.stride - y.width - 2 * m > 0 ? y.stride - y.width - 2 * m : 0 ) ) * 2 ) != VTMHIP_OK
      || A.sync( ctx ) != VTMHIP_OK ) return { r, false };
  g_planes.push_back( r );
  return { r, true };
}
inline RefPlaneRef refPlane( const Picture *pic ) { return refPlaneOf( t_hk ? t_hk->ctx : g_ctx, pic ); }

void hookNoteMismatch( int which, int a, int b, int c, int d, long long ref, long long dev )
{
  if( g_st->hookMismatch[0] + g_st->hookMismatch[1] == 0 )
  {
    const int32_t v[8] = { which, a, b, c, d, (int32_t) ref, (int32_t) dev, 0 };
    memcpy( g_st->hookFirstMismatch, v, sizeof( v ) );
  }
  g_st->hookMismatch[which]++;
}

void meHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, Mv &rcMvPred, int iRefIdxPred, Mv &rcMv, int &riMVPIdx, uint32_t &ruiBits,
             Distortion &ruiCost, const AMVPInfo &amvpInfo, bool bBi )
{
  g_st->hookCalls[0]++;
  const Slice   &slice  = *pu.cu->slice;
  const Picture *refPic = slice.getRefPic( eRefPicList, iRefIdxPred );
  const int      w = pu.Y().width, h = pu.Y().height;
  Mv             cachedMv;
  auto           blkCache = dynamic_cast<CacheBlkInfoCtrl *>( is->m_modeCtrl );
  // what vtmhip_xMotionEstimation_batch_dev does not cover stays with the reference (include/vtmhip.h): BCW / explicit weights, MCTS, composite
  // references, the block-vector cache, full / selective search, wrap-around
  const bool unsupported = pu.cu->BcwIdx != BCW_DEFAULT || slice.getPPS()->getUseWP() || slice.getPPS()->getWPBiPred() || is->m_pcEncCfg->getMCTSEncConstraint()
                        || is->m_useCompositeRef || ( !bBi && blkCache && blkCache->getMv( pu, eRefPicList, iRefIdxPred, cachedMv ) )
                        || ( is->m_motionEstimationSearchMethod != MESEARCH_DIAMOND && is->m_motionEstimationSearchMethod != MESEARCH_DIAMOND_ENHANCED )
                        || is->m_pcEncCfg->getClipForBiPredMeEnabled() || refPic->isWrapAroundEnabled( pu.cs->pps ) || w > 128 || h > 128 || ( w == 4 && h == 4 )
                        || is->m_uniMvListSize > 15 || amvpInfo.numCand > 2 || slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA ) > 10;
  if( unsupported ) g_st->hookUnsupported[0]++;
  if( unsupported || !hookSampled( g_hookCtr[0] ) )
  {
    vtmref_orig_xMotionEstimation( is, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
    return;
  }
  // ---- the job, from the arguments and the members the reference function reads (InterSearch.cpp:3299-3494) ----
  vtmhip_me_job j; memset( &j, 0, sizeof( j ) );
  const RefPlaneRef rp = refPlane( refPic );
  const CPelBuf   org = origBuf.Y();
  std::vector<Pel> blk( size_t( w ) * h ), oth( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], org.buf + ptrdiff_t( y ) * org.stride, sizeof( Pel ) * w );
  if( bBi )
  {
    const CPelBuf o = is->m_tmpPredStorage[1 - (int) eRefPicList].getBuf( UnitAreaRelative( *pu.cu, pu ) ).Y();
    for( int y = 0; y < h; y++ ) memcpy( &oth[size_t( y ) * w], o.buf + ptrdiff_t( y ) * o.stride, sizeof( Pel ) * w );
  }
  const Position pos = pu.cu->lumaPos();
  j.orgOff = 0; j.orgStride = w; j.otherPredOff = 0; j.otherPredStride = w;
  j.refOff = rp ? ( int64_t ) ( rp->margin + pu.Y().y ) * rp->stride + rp->margin + pu.Y().x : 0; j.refStride = rp ? rp->stride : 0;
  j.puX = ( int16_t ) pos.x; j.puY = ( int16_t ) pos.y; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
  j.bi = bBi; j.imv = pu.cu->imv; j.mvpIdx = ( uint8_t ) riMVPIdx; j.numAmvpCand = ( uint8_t ) amvpInfo.numCand;
  j.mvPredHor = rcMvPred.hor; j.mvPredVer = rcMvPred.ver; j.mvHor = rcMv.hor; j.mvVer = rcMv.ver;
  for( int i = 0; i < 2; i++ )
  {
    j.amvpCand[i][0] = amvpInfo.mvCand[i].hor; j.amvpCand[i][1] = amvpInfo.mvCand[i].ver;
    j.mvpIdxBits[i] = is->m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS];
  }
  j.bits = ruiBits; j.searchRange = is->m_aaiAdaptSR[eRefPicList][iRefIdxPred]; j.motionLambda = is->m_pcRdCost->m_motionLambda;
  j.numExtraStart = is->m_uniMvListSize;
  for( int i = 0; i < is->m_uniMvListSize; i++ )
  {
    const BlkUniMvInfo *e = is->m_uniMvList + ( ( is->m_uniMvListIdx - 1 - i + is->m_uniMvListMaxSize ) % is->m_uniMvListMaxSize );
    j.extraStart[i][0] = e->uniMvs[eRefPicList][iRefIdxPred].hor; j.extraStart[i][1] = e->uniMvs[eRefPicList][iRefIdxPred].ver;
  }
  vtmhip_me_cfg cfg; memset( &cfg, 0, sizeof( cfg ) );
  cfg.bipredSearchRange = is->m_bipredSearchRange;
  cfg.useHadME = is->m_pcEncCfg->getUseHADME() && !pu.cu->slice->getDisableSATDForRD();
  cfg.fastInterSearchMode13 = is->m_pcEncCfg->getFastInterSearchMode() == FASTINTERSEARCH_MODE1 || is->m_pcEncCfg->getFastInterSearchMode() == FASTINTERSEARCH_MODE3;
  cfg.extendedSettings = is->m_motionEstimationSearchMethod == MESEARCH_DIAMOND_ENHANCED;
  cfg.firstSearchStop = is->m_pcEncCfg->getFastMEAssumingSmootherMVEnabled();
  cfg.uniformImv = -1;
  vtmhip_pic_params pic; memset( &pic, 0, sizeof( pic ) );
  pic.picW = pu.cs->pps->getPicWidthInLumaSamples(); pic.picH = pu.cs->pps->getPicHeightInLumaSamples(); pic.ctuSize = pu.cs->sps->getMaxCUWidth();
  pic.bitDepth = slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA );
  // ---- the reference's own run (also leaves every member the callers look at in the reference's state) ----
  vtmref_orig_xMotionEstimation( is, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
  vtmhip_me_out o; memset( &o, 0, sizeof( o ) );
  const bool ok = rp && A.h2d( g_ctx, d_hOrg, blk.data(), blk.size() * 2 ) == VTMHIP_OK && ( !bBi || A.h2d( g_ctx, d_hOther, oth.data(), oth.size() * 2 ) == VTMHIP_OK )
               && A.h2d( g_ctx, d_hJob, &j, sizeof( j ) ) == VTMHIP_OK
               && A.me( g_ctx, &pic, &cfg, d_hOrg, rp->dev, d_hOther, ( const vtmhip_me_job * ) d_hJob, 1, w, h, ( vtmhip_me_out * ) d_hOut ) == VTMHIP_OK
               && A.d2h( g_ctx, &o, d_hOut, sizeof( o ) ) == VTMHIP_OK;
  if( !ok ) { note_error(); return; }
  g_st->hookDevice[0]++;
  if( o.mvHor != rcMv.hor || o.mvVer != rcMv.ver || o.mvPredHor != rcMvPred.hor || o.mvPredVer != rcMvPred.ver || o.mvpIdx != riMVPIdx || o.bits != ruiBits || o.cost != ruiCost )
    hookNoteMismatch( 0, w * 1000 + h, bBi * 10 + pu.cu->imv, rcMv.hor - o.mvHor, rcMv.ver - o.mvVer, ( long long ) ruiCost, ( long long ) o.cost );
  // the encoder continues with the device's results
  rcMv.hor = o.mvHor; rcMv.ver = o.mvVer; rcMvPred.hor = o.mvPredHor; rcMvPred.ver = o.mvPredVer; riMVPIdx = o.mvpIdx; ruiBits = o.bits; ruiCost = o.cost;
}


// InterSearch::xEstimateMvPredAMVP (InterSearch.cpp:3088-3128): the candidate list needs the CU context (PU::fillMvpCand) and stays with the reference; the
// template cost of every candidate (xGetTemplateCost: prediction at the clipped candidate + SAD + index rate) and the selection run on the device
void amvpHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, int iRefIdx, Mv &rcMvPred, AMVPInfo &info, bool bFilled, Distortion *puiDistBiP )
{
  g_st->amvpCalls++;
  vtmref_orig_xEstimateMvPredAMVP( is, pu, origBuf, eRefPicList, iRefIdx, rcMvPred, info, bFilled, puiDistBiP );
  const Slice   &slice  = *pu.cu->slice;
  const Picture *refPic = slice.getRefPic( eRefPicList, iRefIdx );
  const int      w = pu.Y().width, h = pu.Y().height;
  const bool unsupported = slice.testWeightPred() || slice.testWeightBiPred() || refPic->isWrapAroundEnabled( pu.cs->pps ) || refPic->isRefScaled( pu.cs->pps ) || w > 128 || h > 128
                        || info.numCand < 1 || info.numCand > 2 || slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA ) > 10 || !puiDistBiP;
  if( unsupported ) g_st->amvpUnsupported++;
  if( unsupported || !hookSampled( g_amvpCtr ) ) return;
  vtmhip_me_job j; memset( &j, 0, sizeof( j ) );
  const RefPlaneRef rp = refPlane( refPic );
  const CPelBuf   org = origBuf.Y();
  std::vector<Pel> blk( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], org.buf + ptrdiff_t( y ) * org.stride, sizeof( Pel ) * w );
  const Position pos = pu.cu->lumaPos();
  j.orgOff = 0; j.orgStride = w;
  j.refOff = rp ? ( int64_t ) ( rp->margin + pu.Y().y ) * rp->stride + rp->margin + pu.Y().x : 0; j.refStride = rp ? rp->stride : 0;
  j.puX = ( int16_t ) pos.x; j.puY = ( int16_t ) pos.y; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
  j.imv = pu.cu->imv; j.numAmvpCand = ( uint8_t ) info.numCand;
  for( int i = 0; i < info.numCand; i++ )
  {
    j.amvpCand[i][0] = info.mvCand[i].hor; j.amvpCand[i][1] = info.mvCand[i].ver;
    j.mvpIdxBits[i] = is->m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS];
  }
  j.motionLambda = is->m_pcRdCost->m_motionLambda;
  vtmhip_pic_params pic; memset( &pic, 0, sizeof( pic ) );
  pic.picW = pu.cs->pps->getPicWidthInLumaSamples(); pic.picH = pu.cs->pps->getPicHeightInLumaSamples(); pic.ctuSize = pu.cs->sps->getMaxCUWidth();
  pic.bitDepth = slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA );
  uint64_t dist = 0;
  const bool ok = rp && A.h2d( g_ctx, d_hOrg, blk.data(), blk.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_hJob, &j, sizeof( j ) ) == VTMHIP_OK
               && A.amvp( g_ctx, &pic, d_hOrg, rp->dev, ( vtmhip_me_job * ) d_hJob, 1, w, h, 0, 0, ( uint64_t * ) d_hOut ) == VTMHIP_OK
               && A.d2h( g_ctx, &j, d_hJob, sizeof( j ) ) == VTMHIP_OK && A.d2h( g_ctx, &dist, d_hOut, sizeof( dist ) ) == VTMHIP_OK;
  if( !ok ) { note_error(); return; }
  g_st->amvpDevice++;
  if( j.mvpIdx != pu.mvpIdx[eRefPicList] || j.mvPredHor != rcMvPred.hor || j.mvPredVer != rcMvPred.ver || dist != *puiDistBiP )
  {
    if( g_st->amvpMismatch++ == 0 && g_st->hookMismatch[0] + g_st->hookMismatch[1] == 0 )
    {
      const int32_t v[8] = { 3, w * 1000 + h, info.numCand * 10 + pu.cu->imv, j.mvpIdx - pu.mvpIdx[eRefPicList], j.mvPredHor - rcMvPred.hor, ( int32_t ) *puiDistBiP, ( int32_t ) dist, 0 };
      memcpy( g_st->hookFirstMismatch, v, sizeof( v ) );
    }
  }
  // the encoder continues with the device's choice
  rcMvPred.hor = j.mvPredHor; rcMvPred.ver = j.mvPredVer; pu.mvpIdx[eRefPicList] = j.mvpIdx; *puiDistBiP = dist;
}

// ---- SMVD (InterSearch.cpp:4341-4518, 7787-7886): one vtmhip_smvd_job per call; the two reference planes are separate device allocations, addressed from the
// searched list's plane ----
struct SmvdCall { vtmhip_smvd_job j; vtmhip_pic_params pic; const int16_t *base; bool ok; };
SmvdCall smvdJob( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eCur, int refIdxCur, int refIdxTar )
{
  SmvdCall c; memset( &c.j, 0, sizeof( c.j ) ); memset( &c.pic, 0, sizeof( c.pic ) ); c.base = nullptr; c.ok = false;
  const Slice   &slice = *pu.cu->slice;
  const RefPicList eTar = RefPicList( 1 - eCur );
  const Picture *picA = slice.getRefPic( eCur, refIdxCur ), *picB = slice.getRefPic( eTar, refIdxTar );
  const int      w = pu.Y().width, h = pu.Y().height;
  if( slice.testWeightPred() || slice.testWeightBiPred() || picA->isWrapAroundEnabled( pu.cs->pps ) || picB->isWrapAroundEnabled( pu.cs->pps ) || picA->isRefScaled( pu.cs->pps )
      || picB->isRefScaled( pu.cs->pps ) || is->m_pcEncCfg->getMCTSEncConstraint() || w > 128 || h > 128 || slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA ) > 12 ) return c;
  const RefPlaneRef pa = refPlane( picA ), pb = refPlane( picB );
  if( !pa || !pb ) return c;
  const RefPlane a = pa.p, b = pb.p;
  const Position  pos = pu.cu->lumaPos();
  vtmhip_smvd_job &j = c.j;
  j.orgOff = 0; j.orgStride = w;
  j.refOff[0] = ( int64_t ) ( a.margin + pu.Y().y ) * a.stride + a.margin + pu.Y().x; j.refStride[0] = a.stride;
  j.refOff[1] = ( int64_t ) ( b.dev - a.dev ) + ( int64_t ) ( b.margin + pu.Y().y ) * b.stride + b.margin + pu.Y().x; j.refStride[1] = b.stride;
  j.puX = ( int16_t ) pos.x; j.puY = ( int16_t ) pos.y; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
  j.imv = pu.cu->imv; j.useSatd = !pu.cu->slice->getDisableSATDForRD(); j.clipBiPred = is->m_pcEncCfg->getClipForBiPredMeEnabled();
  j.bcwWeightTar = getBcwWeight( pu.cu->BcwIdx, eTar );
  for( int i = 0; i < 2; i++ ) j.mvpIdxBits[i] = is->m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS];
  j.motionLambda = is->m_pcRdCost->m_motionLambda;
  c.pic.picW = pu.cs->pps->getPicWidthInLumaSamples(); c.pic.picH = pu.cs->pps->getPicHeightInLumaSamples(); c.pic.ctuSize = pu.cs->sps->getMaxCUWidth();
  c.pic.bitDepth = slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA );
  c.base = a.dev;
  const CPelBuf org = origBuf.Y();
  std::vector<Pel> blk( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], org.buf + ptrdiff_t( y ) * org.stride, sizeof( Pel ) * w );
  c.ok = A.h2d( g_ctx, d_hOrg, blk.data(), blk.size() * 2 ) == VTMHIP_OK;
  return c;
}
bool smvdRun( SmvdCall &c, int op )
{
  const bool ok = A.h2d( g_ctx, d_hJob, &c.j, sizeof( c.j ) ) == VTMHIP_OK
               && A.smvd( g_ctx, &c.pic, d_hOrg, c.base, ( vtmhip_smvd_job * ) d_hJob, 1, c.j.width, c.j.height, op | VTMHIP_SMVD_UNIFORM ) == VTMHIP_OK   // one job: uniform by construction -> the lane-per-tile kernel the level-order driver uses (PUs with a side of 4: the block-wide kernel)
               && A.d2h( g_ctx, &c.j, d_hJob, sizeof( c.j ) ) == VTMHIP_OK;
  if( !ok ) note_error();
  return ok;
}
void smvdMismatch( int op, const vtmhip_smvd_job &j, long long ref, long long dev )
{
  if( g_st->smvdMismatch[op]++ == 0 && g_st->hookMismatch[0] + g_st->hookMismatch[1] == 0 )
  {
    const int32_t v[8] = { 4 + op, j.width * 1000 + j.height, j.imv, j.bcwWeightTar, j.useSatd, ( int32_t ) ref, ( int32_t ) dev, 0 };
    memcpy( g_st->hookFirstMismatch, v, sizeof( v ) );
  }
}

Distortion smvdCostHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eCur, const MvField &cur, MvField &tar, int bcwIdx )
{
  g_st->smvdCalls[0]++;
  const Distortion ref = vtmref_orig_xGetSymmetricCost( is, pu, origBuf, eCur, cur, tar, bcwIdx );
  if( !hookSampled( g_smvdCtr[0] ) ) return ref;
  SmvdCall c = smvdJob( is, pu, origBuf, eCur, cur.refIdx, tar.refIdx );
  if( !c.ok ) { g_st->smvdUnsupported++; return ref; }
  c.j.mvCur[0] = cur.mv.hor; c.j.mvCur[1] = cur.mv.ver; c.j.mvTar[0] = tar.mv.hor; c.j.mvTar[1] = tar.mv.ver;
  if( !smvdRun( c, VTMHIP_SMVD_COST ) ) return ref;
  g_st->smvdDevice[0]++;
  if( c.j.cost != ref ) smvdMismatch( 0, c.j, ( long long ) ref, ( long long ) c.j.cost );
  return c.j.cost;
}

void smvdMeHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, Mv &predCur, Mv &predTar, RefPicList eCur, MvField &cur, MvField &tar, Distortion &cost, int bcwIdx )
{
  g_st->smvdCalls[1]++;
  const MvField    cur0 = cur, tar0 = tar;
  const Distortion cost0 = cost;
  vtmref_orig_xSymmetricMotionEstimation( is, pu, origBuf, predCur, predTar, eCur, cur, tar, cost, bcwIdx );
  if( !hookSampled( g_smvdCtr[1] ) ) return;
  SmvdCall c = smvdJob( is, pu, origBuf, eCur, cur0.refIdx, tar0.refIdx );
  if( !c.ok ) { g_st->smvdUnsupported++; return; }
  c.j.mvCur[0] = cur0.mv.hor; c.j.mvCur[1] = cur0.mv.ver; c.j.mvTar[0] = tar0.mv.hor; c.j.mvTar[1] = tar0.mv.ver;
  c.j.predSym[0][0] = predCur.hor; c.j.predSym[0][1] = predCur.ver; c.j.predSym[1][0] = predTar.hor; c.j.predSym[1][1] = predTar.ver;
  c.j.cost = cost0;
  if( !smvdRun( c, VTMHIP_SMVD_ME ) ) return;
  g_st->smvdDevice[1]++;
  if( c.j.mvCur[0] != cur.mv.hor || c.j.mvCur[1] != cur.mv.ver || c.j.mvTar[0] != tar.mv.hor || c.j.mvTar[1] != tar.mv.ver || c.j.cost != cost )
    smvdMismatch( 1, c.j, ( long long ) cost, ( long long ) c.j.cost );
  cur.mv.set( c.j.mvCur[0], c.j.mvCur[1] ); tar.mv.set( c.j.mvTar[0], c.j.mvTar[1] ); cost = c.j.cost;
}

void smvdCheckHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, Mv curMv, RefPicList curList, AMVPInfo amvpInfo[2][33], int32_t bcwIdx, Mv predSym[2], int32_t idxSym[2],
                    Distortion &bestCost, bool skip )
{
  g_st->smvdCalls[2]++;
  const int        tarList = 1 - curList;
  const Mv         pred0[2] = { predSym[0], predSym[1] };
  const int32_t    idx0[2]  = { idxSym[0], idxSym[1] };
  const Distortion cost0 = bestCost;
  vtmref_orig_symmvdCheckBestMvp( is, pu, origBuf, curMv, curList, amvpInfo, bcwIdx, predSym, idxSym, bestCost, skip );
  if( !hookSampled( g_smvdCtr[2] ) ) return;
  const int refIdxCur = pu.cu->slice->getSymRefIdx( curList ), refIdxTar = pu.cu->slice->getSymRefIdx( tarList );
  SmvdCall  c = smvdJob( is, pu, origBuf, curList, refIdxCur, refIdxTar );
  const AMVPInfo &ac = amvpInfo[curList][refIdxCur], &at = amvpInfo[tarList][refIdxTar];
  if( !c.ok || ac.numCand > 2 || at.numCand > 2 ) { g_st->smvdUnsupported++; return; }
  c.j.numCand[0] = ( uint8_t ) ac.numCand; c.j.numCand[1] = ( uint8_t ) at.numCand;
  for( int i = 0; i < 2; i++ )
  {
    c.j.cand[0][i][0] = ac.mvCand[i].hor; c.j.cand[0][i][1] = ac.mvCand[i].ver; c.j.cand[1][i][0] = at.mvCand[i].hor; c.j.cand[1][i][1] = at.mvCand[i].ver;
  }
  c.j.mvCur[0] = curMv.hor; c.j.mvCur[1] = curMv.ver; c.j.skip = skip;
  c.j.predSym[0][0] = pred0[curList].hor; c.j.predSym[0][1] = pred0[curList].ver; c.j.predSym[1][0] = pred0[tarList].hor; c.j.predSym[1][1] = pred0[tarList].ver;
  c.j.mvpIdxSym[0] = idx0[curList]; c.j.mvpIdxSym[1] = idx0[tarList];
  c.j.cost = cost0;
  if( !smvdRun( c, VTMHIP_SMVD_CHECK_MVP ) ) return;
  g_st->smvdDevice[2]++;
  const bool changed = bestCost != cost0;     // the predictors are outputs only when a pair improved the cost
  bool bad = c.j.cost != bestCost;
  if( changed ) bad |= c.j.predSym[0][0] != predSym[curList].hor || c.j.predSym[0][1] != predSym[curList].ver || c.j.predSym[1][0] != predSym[tarList].hor
                    || c.j.predSym[1][1] != predSym[tarList].ver || c.j.mvpIdxSym[0] != idxSym[curList] || c.j.mvpIdxSym[1] != idxSym[tarList];
  if( bad ) smvdMismatch( 2, c.j, ( long long ) bestCost, ( long long ) c.j.cost );
  bestCost = c.j.cost;
  if( c.j.cost != cost0 )
  {
    predSym[curList].set( c.j.predSym[0][0], c.j.predSym[0][1] ); predSym[tarList].set( c.j.predSym[1][0], c.j.predSym[1][1] );
    idxSym[curList] = c.j.mvpIdxSym[0]; idxSym[tarList] = c.j.mvpIdxSym[1];
  }
}

void affineHook( InterSearch *is, PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, Mv acMvPred[3], int iRefIdxPred, Mv acMv[3], uint32_t &ruiBits, Distortion &ruiCost,
                 int &mvpIdx, const AffineAMVPInfo &aamvpi, bool bBi )
{
  ST_INC( affineCalls );
  HookThread *T = hookThread();
  const Slice   &slice  = *pu.cu->slice;
  const Picture *refPic = slice.getRefPic( eRefPicList, iRefIdxPred );
  const int      w = pu.Y().width, h = pu.Y().height;
  const bool     encOpt = is->m_pcEncCfg->getUseAffineAmvrEncOpt();
  const bool     bcw = pu.cu->cs->sps->getUseBcw() && pu.cu->BcwIdx != BCW_DEFAULT;
  if( bcw && !bBi && is->m_uniMotions.isReadModeAffine( ( uint32_t ) eRefPicList, ( uint32_t ) iRefIdxPred, pu.cu->affineType ) )
  {
    // xReadBufferedAffineUniMv (:5352-5357, 7699-7716): the member only copies the default-weight pass's model out of m_uniMotions and re-prices it -- nothing to search
    ST_ADD( affineCalls, -1 );
    vtmref_orig_xAffineMotionEstimation( is, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi );
    return;
  }
  const bool pickMvp = pu.cu->imv == 2 && encOpt;      // xDetermineBestMvp over the affine AMVP list (:5444-5449, 5629-5634): the list travels in the job
  const bool unsupported = is->m_pcEncCfg->getMCTSEncConstraint() || is->m_pcEncCfg->getClipForBiPredMeEnabled() || ( pickMvp && ( aamvpi.numCand < 1 || aamvpi.numCand > 2 ) )
                        || refPic->isWrapAroundEnabled( pu.cs->pps ) || refPic->isRefScaled( pu.cs->pps ) || w < 16 || h < 16 || w > 128 || h > 128
                        || slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA ) > 10 || slice.getPPS()->getUseWP() || slice.getPPS()->getWPBiPred();
  if( unsupported || !T ) ST_INC( affineUnsupported );
  if( unsupported || !T || !hookSampled( g_affineCtr ) )
  {
    vtmref_orig_xAffineMotionEstimation( is, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi );
    return;
  }
  vtmhip_affine_me_job j; memset( &j, 0, sizeof( j ) );
  const RefPlaneRef rp = refPlane( refPic );
  const CPelBuf   org = origBuf.Y();
  std::vector<Pel> blk( size_t( w ) * h ), oth( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], org.buf + ptrdiff_t( y ) * org.stride, sizeof( Pel ) * w );
  if( bBi )
  {
    const CPelBuf o = is->m_tmpPredStorage[1 - (int) eRefPicList].getBuf( UnitAreaRelative( *pu.cu, pu ) ).Y();
    for( int y = 0; y < h; y++ ) memcpy( &oth[size_t( y ) * w], o.buf + ptrdiff_t( y ) * o.stride, sizeof( Pel ) * w );
  }
  const Position pos = pu.cu->lumaPos();
  j.orgOff = 0; j.orgStride = w; j.otherPredOff = 0; j.otherPredStride = w;
  j.refOff = rp ? ( int64_t ) ( rp->margin + pu.Y().y ) * rp->stride + rp->margin + pu.Y().x : 0; j.refStride = rp ? rp->stride : 0;
  j.puX = ( int16_t ) pos.x; j.puY = ( int16_t ) pos.y; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
  j.sixParam = pu.cu->affineType == AFFINEMODEL_6PARAM; j.interDir = pu.interDir; j.imv = pu.cu->imv; j.bi = bBi;
  j.useSatd = !pu.cs->slice->getDisableSATDForRD(); j.useAffineType = pu.cu->cs->sps->getUseAffineType(); j.amvrEncOpt = encOpt;
  j.lowDelayRounds = is->m_pcEncCfg->getIntraPeriod() == -1;
  j.profAllowed = pu.cs->sps->getUsePROF() && !is->m_skipPROF && !pu.cs->picHeader->getDisProfFlag();
  j.profNeedsLargeGrad = is->m_encOnly && !pu.cu->slice->getCheckLDC(); j.profIsBi = is->m_isBi;
  j.bcwWeight = ( bcw && bBi ) ? getBcwWeight( pu.cu->BcwIdx, eRefPicList ) : 0;      // removeHighFreq( ..., getBcwWeight ) / xGetMEDistortionWeight (:5377-5385)
  for( int i = 0; i < 3; i++ ) { j.mvPred[i][0] = acMvPred[i].hor; j.mvPred[i][1] = acMvPred[i].ver; j.mv[i][0] = acMv[i].hor; j.mv[i][1] = acMv[i].ver; }
  j.bits = ruiBits; j.motionLambda = is->m_pcRdCost->m_motionLambda; j.hevcCost = is->m_hevcCost;
  if( pickMvp )
  {
    j.numAmvpCand = ( uint8_t ) aamvpi.numCand; j.mvpIdx = ( uint8_t ) mvpIdx;
    for( int i = 0; i < aamvpi.numCand; i++ )
    {
      const Mv cp[3] = { aamvpi.mvCandLT[i], aamvpi.mvCandRT[i], aamvpi.mvCandLB[i] };
      for( int v = 0; v < 3; v++ ) { j.amvpCand[i][v][0] = cp[v].hor; j.amvpCand[i][v][1] = cp[v].ver; }
      j.mvpIdxBits[i] = is->m_auiMVPIdxCost[i][aamvpi.numCand];
    }
  }
  const int mvpIdxIn = mvpIdx;
  vtmhip_pic_params pic; memset( &pic, 0, sizeof( pic ) );
  pic.picW = pu.cs->pps->getPicWidthInLumaSamples(); pic.picH = pu.cs->pps->getPicHeightInLumaSamples(); pic.ctuSize = pu.cs->sps->getMaxCUWidth();
  pic.bitDepth = slice.getSPS()->getBitDepth( CHANNEL_TYPE_LUMA );
  const bool replaceOnly = g_pisReplace;      // VTMREF_REPLACE=1: the member's body does not run (its outputs on this path: acMv, ruiBits, ruiCost)
  const uint64_t tA = nowNs();
  if( !replaceOnly ) vtmref_orig_xAffineMotionEstimation( is, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi );
  __atomic_fetch_add( &g_affineNs[0], nowNs() - tA, __ATOMIC_RELAXED );
  vtmhip_affine_me_out o; memset( &o, 0, sizeof( o ) );
  const bool ok = rp && A.h2d( T->ctx, T->d_org, blk.data(), blk.size() * 2 ) == VTMHIP_OK && ( !bBi || A.h2d( T->ctx, T->d_other, oth.data(), oth.size() * 2 ) == VTMHIP_OK )
               && A.h2d( T->ctx, T->d_job, &j, sizeof( j ) ) == VTMHIP_OK
               && ( j.bcwWeight ? A.affineMeBcw : A.affineMe )( T->ctx, &pic, T->d_org, rp->dev, T->d_other, ( const vtmhip_affine_me_job * ) T->d_job, 1, w, h, ( vtmhip_affine_me_out * ) T->d_out ) == VTMHIP_OK
               && A.d2h( T->ctx, &o, T->d_out, sizeof( o ) ) == VTMHIP_OK;
  __atomic_fetch_add( &g_affineNs[1], nowNs() - tA, __ATOMIC_RELAXED );
  if( !ok ) { note_error( T->ctx ); if( replaceOnly ) vtmref_orig_xAffineMotionEstimation( is, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi ); return; }
  ST_INC( affineDevice );
  const int mvNum = j.sixParam ? 3 : 2;
  bool bad = !replaceOnly && ( o.bits != ruiBits || o.cost != ruiCost || ( pickMvp && o.mvpIdx != mvpIdx ) );
  for( int i = 0; i < mvNum && !replaceOnly; i++ ) bad |= o.mv[i][0] != acMv[i].hor || o.mv[i][1] != acMv[i].ver;
  if( bad ) { if( ST_INC( affineMismatch ) == 0 && g_st->hookMismatch[0] + g_st->hookMismatch[1] == 0 ) { const int32_t v[8] = { 2, w * 1000 + h, j.sixParam * 100 + j.bi * 10 + j.imv, o.mv[0][0] - acMv[0].hor, o.mv[1][0] - acMv[1].hor, ( int32_t ) ruiCost, ( int32_t ) o.cost, 0 }; memcpy( g_st->hookFirstMismatch, v, sizeof( v ) ); } }
  for( int i = 0; i < mvNum; i++ ) { acMv[i].hor = o.mv[i][0]; acMv[i].ver = o.mv[i][1]; }
  ruiBits = o.bits; ruiCost = o.cost;
  // the member's last statements (:5768-5770): acMvPred = the AMVP candidate of mvpIdx -- which only moves when xDetermineBestMvp ran
  mvpIdx = pickMvp ? o.mvpIdx : mvpIdxIn;
  acMvPred[0] = aamvpi.mvCandLT[mvpIdx]; acMvPred[1] = aamvpi.mvCandRT[mvpIdx]; acMvPred[2] = aamvpi.mvCandLB[mvpIdx];
}

// what the trampoline of xFwdLfnst / xInvLfnst derives from the TU exactly as the reference does (TrQuant.cpp:346-366): does LFNST apply, and with
// which matrix set / orientation
bool lfnstParams( const TransformUnit &tu, const ComponentID compID, int &mode, bool &transpose )
{
  const CompArea &area = tu.blocks[compID];
  const uint32_t  lfnstIdx = tu.cu->lfnstIdx;
  if( !( lfnstIdx && tu.mtsIdx[compID] != MTS_SKIP && ( tu.cu->isSepTree() ? true : isLuma( compID ) ) ) || lfnstIdx >= 3 ) return false;
  uint32_t intraMode = PU::getFinalIntraMode( *tu.cs->getPU( area.pos(), toChannelType( compID ) ), toChannelType( compID ) );
  if( PU::isLMCMode( tu.cs->getPU( area.pos(), toChannelType( compID ) )->intraDir[toChannelType( compID )] ) ) intraMode = PU::getCoLocatedIntraLumaMode( *tu.cs->getPU( area.pos(), toChannelType( compID ) ) );
  if( PU::isMIP( *tu.cs->getPU( area.pos(), toChannelType( compID ) ), toChannelType( compID ) ) ) intraMode = PLANAR_IDX;
  TrQuant dummy;
  intraMode = dummy.getLFNSTIntraMode( PU::getWideAngle( tu, intraMode, compID ) );
  transpose = dummy.getTransposeFlag( intraMode );
  mode      = g_lfnstLut[intraMode];
  return true;
}

void lfnstHook( TrQuant *tq, const TransformUnit &tu, const ComponentID compID, bool inverse, bool loadTr )
{
  const int which = inverse ? 1 : 0;
  g_st->lfnstCalls[which]++;
  const CompArea &area = tu.blocks[compID];
  const int       w = area.width, h = area.height;
  int  mode = 0;
  bool transpose = false;
  TCoeff *buf = inverse ? tq->m_tempCoeff : ( loadTr ? tq->m_mtsCoeffs[tu.mtsIdx[compID]] : tq->m_tempCoeff );
  const bool applies = lfnstParams( tu, compID, mode, transpose ) && w >= 4 && h >= 4 && w <= 64 && h <= 64;
  std::vector<TCoeff> before;
  // an encode makes ~10^6 LFNST calls: every 16th sampled call is enough
  const bool dev = applies && hookSampled( g_lfnstCtr[which] ) && ( g_lfnstCtr[which] & 15 ) == 1;
  if( dev ) before.assign( buf, buf + size_t( w ) * h );
  if( inverse ) vtmref_orig_xInvLfnst( tq, tu, compID ); else vtmref_orig_xFwdLfnst( tq, tu, compID, loadTr );
  if( !dev ) return;
  vtmhip_lfnst_tu_job j; memset( &j, 0, sizeof( j ) );
  j.coefOff = 0; j.width = ( int16_t ) w; j.height = ( int16_t ) h; j.mode = ( uint8_t ) mode; j.index = ( uint8_t ) ( tu.cu->lfnstIdx - 1 ); j.transpose = transpose; j.inverse = inverse;
  std::vector<TCoeff> out( size_t( w ) * h );
  const bool ok = A.h2d( g_ctx, d_hCoef, before.data(), before.size() * 4 ) == VTMHIP_OK && A.h2d( g_ctx, d_hJob, &j, sizeof( j ) ) == VTMHIP_OK
               && A.lfnstTu( g_ctx, d_hCoef, ( const vtmhip_lfnst_tu_job * ) d_hJob, 1 ) == VTMHIP_OK && A.d2h( g_ctx, out.data(), d_hCoef, out.size() * 4 ) == VTMHIP_OK;
  if( !ok ) { note_error(); return; }
  g_st->lfnstDevice[which]++;
  if( memcmp( out.data(), buf, out.size() * 4 ) != 0 ) g_st->lfnstMismatch[which]++;
  else memcpy( buf, out.data(), out.size() * 4 );
}

void mtsHook( TrQuant *tq, TransformUnit &tu, const ComponentID &compID, const QpParam &cQP, std::vector<TrMode> *trModes, const int maxCand )
{
  g_st->hookCalls[1]++;
  const CompArea &rect = tu.blocks[compID];
  const int       w = rect.width, h = rect.height, n = ( int ) trModes->size();
  // LFNST changes the zero-out of xT; a TU without residual has nothing to transform: both stay with the reference
  const bool skip = tu.noResidual || ( tu.cs->sps->getUseLFNST() && tu.cu->lfnstIdx ) || w < 4 || h < 4 || w > 64 || h > 64 || n > 8 || n < 1;
  vtmref_orig_transformNxN_select( tq, tu, compID, cQP, trModes, maxCand );
  if( skip ) { g_st->hookUnsupported[1]++; return; }
  if( !hookSampled( g_hookCtr[1] ) ) return;
  const CPelBuf    resi = tu.cs->getResiBuf( rect );
  const int        bd = tu.cs->sps->getBitDepth( toChannelType( compID ) );
  std::vector<Pel> blk( size_t( w ) * h );
  for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], resi.buf + ptrdiff_t( y ) * resi.stride, sizeof( Pel ) * w );
  // one batch: the forward transform of every candidate (a transform-skip candidate: the residual itself)
  vtmhip_tr_job jobs[8]; memset( jobs, 0, sizeof( jobs ) );
  uint8_t       mts[8];
  int32_t       sums[8] = { 0 };
  int           nTr = 0, trOf[8];
  for( int i = 0; i < n; i++ )
  {
    mts[i] = ( uint8_t ) trModes->at( i ).first;
    if( mts[i] == MTS_SKIP ) { long long sa = 0; for( Pel v : blk ) sa += abs( v ); sums[i] = ( int32_t ) sa; continue; }
    int th = DCT2, tv = DCT2;
    tu.mtsIdx[compID] = mts[i];
    tq->getTrTypes( tu, compID, th, tv );
    vtmhip_tr_job &t = jobs[nTr];
    t.srcOff = 0; t.dstOff = ( int64_t ) nTr * w * h; t.srcStride = w; t.dstStride = w; t.width = ( int16_t ) w; t.height = ( int16_t ) h;
    t.typeHor = ( uint8_t ) th; t.typeVer = ( uint8_t ) tv; t.bitDepth = ( uint8_t ) bd;
    trOf[nTr++] = i;
  }
  tu.mtsIdx[compID] = trModes->back().first;   // what the reference's loop leaves behind
  std::vector<int32_t> coef( size_t( nTr ) * w * h ), sumsDev( 8 );
  const bool ok = A.h2d( g_ctx, d_hOrg, blk.data(), blk.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_hJob, jobs, sizeof( vtmhip_tr_job ) * ( nTr ? nTr : 1 ) ) == VTMHIP_OK
               && ( nTr == 0 || ( A.xT( g_ctx, d_hOrg, d_hCoef, ( const vtmhip_tr_job * ) d_hJob, nTr, w, h, d_hSum ) == VTMHIP_OK
                                  && A.d2h( g_ctx, coef.data(), d_hCoef, coef.size() * 4 ) == VTMHIP_OK && A.d2h( g_ctx, sumsDev.data(), d_hSum, 4 * nTr ) == VTMHIP_OK ) );
  if( !ok ) { note_error(); return; }
  g_st->hookDevice[1]++;
  bool bad = false;
  for( int k = 0; k < nTr; k++ )
  {
    sums[trOf[k]] = sumsDev[k];
    if( memcmp( &coef[size_t( k ) * w * h], tq->m_mtsCoeffs[mts[trOf[k]]], sizeof( TCoeff ) * w * h ) != 0 ) bad = true;
    else memcpy( tq->m_mtsCoeffs[mts[trOf[k]]], &coef[size_t( k ) * w * h], sizeof( TCoeff ) * w * h );   // the later transformNxN( loadTr ) reads these
  }
  uint8_t test[8];
  if( A.mtsSelect( sums, mts, n, w, h, bd, tu.cs->sps->getMaxLog2TrDynamicRange( toChannelType( compID ) ), maxCand, test ) != VTMHIP_OK ) { note_error(); return; }
  for( int i = 0; i < n; i++ ) if( ( test[i] != 0 ) != trModes->at( i ).second ) bad = true;
  if( bad ) hookNoteMismatch( 1, w * 1000 + h, n, maxCand, nTr, 0, 0 );
  for( int i = 0; i < n; i++ ) trModes->at( i ).second = test[i] != 0;
}
#include "ref_shim_pis.hpp"
#include "ref_shim_intra.hpp"
bool g_intraLive = false;      // the intra hook is installed: the distortion trampolines ask intraServe first
}   // namespace

// the strong definitions that take over the weakened reference symbols
void IntraPrediction::initIntraPatternChType( const CodingUnit &cu, const CompArea &area, const bool forceRefFilterFlag )
{
  vtmref_orig_initIntraPatternChType( this, cu, area, forceRefFilterFlag );
  if( g_hookIntra && forceRefFilterFlag ) intraArm( this, cu, area );
  else if( g_hookIntra ) g_intra.armed = g_intra.serving = false;
}
void InterSearch::predInterSearch( CodingUnit &cu, Partitioner &partitioner )
{
  if( g_hookPis ) pisHook( this, cu, partitioner ); else vtmref_orig_predInterSearch( this, cu, partitioner );
}
void InterSearch::xMotionEstimation( PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, Mv &rcMvPred, int iRefIdxPred, Mv &rcMv, int &riMVPIdx, uint32_t &ruiBits,
                                     Distortion &ruiCost, const AMVPInfo &amvpInfo, bool bBi )
{
  if( g_rp.active ) pisServeMe( this, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
  else if( g_hookMe ) meHook( this, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
  else vtmref_orig_xMotionEstimation( this, pu, origBuf, eRefPicList, rcMvPred, iRefIdxPred, rcMv, riMVPIdx, ruiBits, ruiCost, amvpInfo, bBi );
}
void InterSearch::xAffineMotionEstimation( PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, Mv acMvPred[3], int iRefIdxPred, Mv acMv[3], uint32_t &ruiBits,
                                           Distortion &ruiCost, int &mvpIdx, const AffineAMVPInfo &aamvpi, bool bBi )
{
  if( g_hookAffine ) affineHook( this, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi );
  else vtmref_orig_xAffineMotionEstimation( this, pu, origBuf, eRefPicList, acMvPred, iRefIdxPred, acMv, ruiBits, ruiCost, mvpIdx, aamvpi, bBi );
}
void InterSearch::xEstimateMvPredAMVP( PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eRefPicList, int iRefIdx, Mv &rcMvPred, AMVPInfo &amvpInfo, bool bFilled, Distortion *puiDistBiP )
{
  if( g_rp.active ) pisServeAmvp( this, pu, origBuf, eRefPicList, iRefIdx, rcMvPred, amvpInfo, bFilled, puiDistBiP );
  else if( g_hookAmvp ) amvpHook( this, pu, origBuf, eRefPicList, iRefIdx, rcMvPred, amvpInfo, bFilled, puiDistBiP );
  else vtmref_orig_xEstimateMvPredAMVP( this, pu, origBuf, eRefPicList, iRefIdx, rcMvPred, amvpInfo, bFilled, puiDistBiP );
}
Distortion InterSearch::xGetSymmetricCost( PredictionUnit &pu, PelUnitBuf &origBuf, RefPicList eCurRefPicList, const MvField &cCurMvField, MvField &cTarMvField, int bcwIdx )
{
  if( g_rp.active ) return pisServeSmvdCost( this, pu, origBuf, eCurRefPicList, cCurMvField, cTarMvField, bcwIdx );
  if( g_hookSmvd ) return smvdCostHook( this, pu, origBuf, eCurRefPicList, cCurMvField, cTarMvField, bcwIdx );
  return vtmref_orig_xGetSymmetricCost( this, pu, origBuf, eCurRefPicList, cCurMvField, cTarMvField, bcwIdx );
}
void InterSearch::xSymmetricMotionEstimation( PredictionUnit &pu, PelUnitBuf &origBuf, Mv &rcMvCurPred, Mv &rcMvTarPred, RefPicList eRefPicList, MvField &rCurMvField,
                                              MvField &rTarMvField, Distortion &ruiCost, int bcwIdx )
{
  if( g_rp.active ) pisServeSmvdMe( this, pu, origBuf, rcMvCurPred, rcMvTarPred, eRefPicList, rCurMvField, rTarMvField, ruiCost, bcwIdx );
  else if( g_hookSmvd ) smvdMeHook( this, pu, origBuf, rcMvCurPred, rcMvTarPred, eRefPicList, rCurMvField, rTarMvField, ruiCost, bcwIdx );
  else vtmref_orig_xSymmetricMotionEstimation( this, pu, origBuf, rcMvCurPred, rcMvTarPred, eRefPicList, rCurMvField, rTarMvField, ruiCost, bcwIdx );
}
void InterSearch::symmvdCheckBestMvp( PredictionUnit &pu, PelUnitBuf &origBuf, Mv curMv, RefPicList curRefList, AMVPInfo amvpInfo[2][33], int32_t bcwIdx, Mv cMvPredSym[2],
                                      int32_t mvpIdxSym[2], Distortion &bestCost, bool skip )
{
  if( g_rp.active ) pisServeSmvdCheck( this, pu, origBuf, curMv, curRefList, amvpInfo, bcwIdx, cMvPredSym, mvpIdxSym, bestCost, skip );
  else if( g_hookSmvd ) smvdCheckHook( this, pu, origBuf, curMv, curRefList, amvpInfo, bcwIdx, cMvPredSym, mvpIdxSym, bestCost, skip );
  else vtmref_orig_symmvdCheckBestMvp( this, pu, origBuf, curMv, curRefList, amvpInfo, bcwIdx, cMvPredSym, mvpIdxSym, bestCost, skip );
}
void TrQuant::xFwdLfnst( const TransformUnit &tu, const ComponentID compID, const bool loadTr )
{
  if( g_hookLfnst ) lfnstHook( this, tu, compID, false, loadTr ); else vtmref_orig_xFwdLfnst( this, tu, compID, loadTr );
}
void TrQuant::xInvLfnst( const TransformUnit &tu, const ComponentID compID )
{
  if( g_hookLfnst ) lfnstHook( this, tu, compID, true, false ); else vtmref_orig_xInvLfnst( this, tu, compID );
}
void TrQuant::transformNxN( TransformUnit &tu, const ComponentID &compID, const QpParam &cQP, std::vector<TrMode> *trModes, const int maxCand )
{
  if( g_hookMts ) mtsHook( this, tu, compID, cQP, trModes, maxCand );
  else vtmref_orig_transformNxN_select( this, tu, compID, cQP, trModes, maxCand );
}
namespace
{
}   // namespace

extern "C" int ref_encode( int argc, char **argv, const char *vtmhipPath, unsigned familyMask, uint64_t stride, uint64_t head, RefEncStats *stats )
{
  memset( stats, 0, sizeof( *stats ) );
  g_st = stats; g_stride = stride ? stride : 1; g_head = head; g_mask = familyMask; g_countOnly = ( familyMask & 8 ) != 0;
  memset( g_distCtr, 0, sizeof( g_distCtr ) ); memset( g_ifCtr, 0, sizeof( g_ifCtr ) ); memset( g_ifCopyCtr, 0, sizeof( g_ifCopyCtr ) ); g_geoCtr = 0; memset( g_trCtr, 0, sizeof( g_trCtr ) ); memset( g_auxCtr, 0, sizeof( g_auxCtr ) );
  memset( g_distOrig, 0, sizeof( g_distOrig ) ); memset( g_fwdOrig, 0, sizeof( g_fwdOrig ) ); memset( g_invOrig, 0, sizeof( g_invOrig ) );
  if( vtmhipPath && !g_countOnly )
  {
    A.so = dlopen( vtmhipPath, RTLD_NOW | RTLD_GLOBAL );
    if( !A.so ) { fprintf( stderr, "ref_encode: %s\n", dlerror() ); return -10; }
    const bool ok = sym( A.create, "vtmhip_create" ) && sym( A.destroy, "vtmhip_destroy" ) && sym( A.last_error, "vtmhip_last_error" ) && sym( A.sad, "vtmhip_xGetSAD" )
                 && sym( A.had, "vtmhip_xGetHADs" ) && sym( A.sse, "vtmhip_xGetSSE" ) && sym( A.sadmask, "vtmhip_xGetSADwMask" ) && sym( A.fhor, "vtmhip_filterHor" ) && sym( A.fver, "vtmhip_filterVer" )
                 && sym( A.fcopy, "vtmhip_filterCopy" ) && sym( A.geo, "vtmhip_weightedGeoBlk" ) && sym( A.fwd, "vtmhip_fastFwdTrans" ) && sym( A.inv, "vtmhip_fastInvTrans" )
                 && sym( A.dalloc, "vtmhip_dev_alloc" ) && sym( A.dfree, "vtmhip_dev_free" ) && sym( A.h2d, "vtmhip_h2d" ) && sym( A.d2h, "vtmhip_d2h" ) && sym( A.sync, "vtmhip_sync" )
                 && sym( A.addAvg, "vtmhip_add_avg_batch_dev" ) && sym( A.rhf, "vtmhip_remove_high_freq_batch_dev" )
                 && sym( A.sobel, "vtmhip_affine_sobel_batch_dev" ) && sym( A.eqc, "vtmhip_affine_equal_coeff_batch_dev" )
                 && sym( A.amvp, "vtmhip_xEstimateMvPredAMVP_batch_dev" ) && sym( A.smvd, "vtmhip_smvd_batch_dev" ) && sym( A.me, "vtmhip_xMotionEstimation_batch_dev" ) && sym( A.xT, "vtmhip_xT_batch_dev" ) && sym( A.tsChain, "vtmhip_tu_ts_chain_batch_dev" )
                 && sym( A.mtsSelect, "vtmhip_mts_select2" ) && sym( A.affineMe, "vtmhip_xAffineMotionEstimation_batch_dev" ) && sym( A.affineMeBcw, "vtmhip_xAffineMotionEstimation_bcw_batch_dev" )
                 && sym( A.lfnstTables, "vtmhip_lfnst_set_tables" ) && sym( A.lfnstTu, "vtmhip_lfnst_tu_batch_dev" );
    if( !ok || !sym( g_apiPis, "vtmhip_predInterSearch_batch_dev" ) || !sym( g_apiUniformShape, "vtmhip_is_uniform_shape" ) || !sym( g_apiHostAlloc, "vtmhip_host_alloc" ) )
    { fprintf( stderr, "ref_encode: libvtmhip.so lacks a pointer-surface symbol\n" ); return -11; }
    const int st = A.create( 0, &g_ctx );
    if( st != VTMHIP_OK ) { fprintf( stderr, "ref_encode: vtmhip_create failed (%d) -- no CPU fallback\n", st ); return -12; }
  }

  int rc = 0;
  std::fstream bitstream;
  EncLibCommon common;
  initROM();
  TComHash::initBlockSizeToIndex();
  EncApp *app = new EncApp( bitstream, &common );
  app->create();
  try
  {
    if( !app->parseCfg( argc, argv ) ) { rc = 1; }
    else
    {
      app->createLib( 0 );
      if( g_ctx || g_countOnly )
      {
        installDist( std::make_integer_sequence<int, DF_TOTAL_FUNCTIONS>() );
#if VTMREF_SPLIT_PARALLEL
        // the split-parallel build holds one InterSearch / TrQuant / ... stack per job (EncLib.h:82-131): only the CU-level hooks (masks 128, 2048), which receive the instance
        // they were called on, are thread-aware here
        if( g_mask & ~( 8u | 128u | 2048u ) ) { fprintf( stderr, "ref_encode: the split-parallel build takes the CU-level hooks only (masks 8, 128, 2048)\n" ); rc = 3; }
#else
        if( g_mask & 2 ) installIf( app->m_cEncLib.m_cInterSearch.m_if );
        if( g_mask & 4 ) installTr();
        if( ( g_mask & 16 ) && ( g_countOnly || auxAlloc() ) ) installAux( (AffineGradientSearch &) app->m_cEncLib.m_cInterSearch );   // private base: C-style cast
#endif
        g_pisDump = nullptr; g_pisDumpedPlanes.clear(); g_pisCtr = g_pisDumpCtr = 0;
        if( ( g_mask & 4096 ) && g_ctx && !g_countOnly && sym( g_apiIntra, "vtmhip_intra_cand_cost_batch_dev" ) && intraAlloc() )
        {
          g_hookIntra = g_intraLive = true;
          if( !( g_mask & 2048 ) ) g_pisReplace = getenv( "VTMREF_REPLACE" ) && atoi( getenv( "VTMREF_REPLACE" ) ) != 0;
        }
        if( g_mask & 2048 )
        {
          // VTMREF_PIS_DUMP: record mode (no device); otherwise compare mode, VTMREF_REPLACE=1: replace mode
          if( getenv( "VTMREF_PIS_DUMP" ) ) { g_pisDump = fopen( getenv( "VTMREF_PIS_DUMP" ), "wb" ); g_pisDumpStride = getenv( "VTMREF_PIS_DUMP_STRIDE" ) ? strtoull( getenv( "VTMREF_PIS_DUMP_STRIDE" ), nullptr, 10 ) : 1;
            g_pisDumpBcwStride = getenv( "VTMREF_PIS_DUMP_BCW_STRIDE" ) ? strtoull( getenv( "VTMREF_PIS_DUMP_BCW_STRIDE" ), nullptr, 10 ) : 0; g_pisDumpBcwCtr = 0; }
          g_pisReplace = getenv( "VTMREF_REPLACE" ) && atoi( getenv( "VTMREF_REPLACE" ) ) != 0;
          g_hookPis = g_pisDump != nullptr || g_countOnly || g_ctx != nullptr;      // (count-only: the hook only times the member; the slots are made per encoder thread: hookThread)
        }
        if( ( g_mask & ( 96 | 128 | 256 | 512 | 1024 ) ) && ( g_countOnly || ( g_ctx && hookAlloc() ) ) ) { g_hookAffine = ( g_mask & 128 ) != 0; g_affineCtr = 0;
          g_hookAmvp = ( g_mask & 512 ) != 0 && !g_countOnly; g_amvpCtr = 0;
          g_hookSmvd = ( g_mask & 1024 ) != 0 && !g_countOnly; g_smvdCtr[0] = g_smvdCtr[1] = g_smvdCtr[2] = 0;
          g_hookLfnst = ( g_mask & 256 ) != 0 && ( g_countOnly || A.lfnstTables( g_ctx, &g_lfnst8x8[0][0][0][0], &g_lfnst4x4[0][0][0][0] ) == VTMHIP_OK ); g_lfnstCtr[0] = g_lfnstCtr[1] = 0; g_hookMe = ( g_mask & 32 ) != 0; g_hookMts = ( g_mask & 64 ) != 0; g_hookCtr[0] = g_hookCtr[1] = 0;
          g_hookStride = getenv( "VTMREF_HOOK_STRIDE" ) ? strtoull( getenv( "VTMREF_HOOK_STRIDE" ), nullptr, 10 ) : 0; }
      }
      bool eos = false;
      while( !eos && rc == 0 )
      {
        while( app->encodePrep( eos ) ) {}
        while( app->encode() ) {}
      }
      restoreAux();      // InterSearch dies with the library
      app->destroyLib();
    }
  }
  catch( Exception &e ) { fprintf( stderr, "ref_encode: %s\n", e.what() ); rc = 2; }
  g_hookMe = g_hookMts = g_hookAffine = g_hookLfnst = g_hookAmvp = g_hookSmvd = g_hookPis = g_hookIntra = g_intraLive = false; g_intra = IntraBatch();
  stats->affineNs[0] = g_affineNs[0]; stats->affineNs[1] = g_affineNs[1]; g_affineNs[0] = g_affineNs[1] = 0;
  if( g_pisDump ) { fclose( g_pisDump ); g_pisDump = nullptr; }
  stats->hookThreads = ( uint64_t ) hookThreadsCount();
  for( RefPlane &p : g_planes ) A.dfree( g_ctx, p.alloc );
  g_planes.clear();
  hookThreadsFree();
  if( g_ctx || g_countOnly ) { restoreDist(); restoreTr(); restoreAux(); }
  app->destroy();
  delete app;
  destroyROM();
  if( g_ctx ) { A.destroy( g_ctx ); g_ctx = nullptr; }
  fflush( stdout );
  return rc;
}
