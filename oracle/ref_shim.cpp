// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// ref_shim.cpp: our own thin C-ABI over the REAL reference (VTM 9.3) objects that
// oracle/Makefile.ref compiles from the sources where they lie under /root/reference.
// It exposes the reference's own dispatch surface for the hot path so that
//   (1) the plain-C restatement in oracle/vtm_oracle.c can be pinned against the reference, and
//   (2) tests/golden/gen_golden.py can record golden input/output vectors, and
//   (3) bench.py's cpu_baseline leg can time the reference's x86-SIMD kernels ("kind":"reference").
//
// Nothing here is reference source: it only #includes the reference headers in place and calls
//   DistParam::distFunc                (CommonLib/RdCost.h:60,67-105, RdCost.cpp:238-324)
//   InterpolationFilter::filterHor/Ver (CommonLib/InterpolationFilter.h:93-110, .cpp:749-891)
//   fastFwdTrans / fastInvTrans        (CommonLib/TrQuant.cpp:69-81, TrQuant_EMT.cpp)
//   g_trCore* matrices                 (CommonLib/Rom.h:115-130)
//   AffineGradientSearch::m_*          (CommonLib/AffineGradientSearch.h:50-54)
//   g_pelBufOP                         (CommonLib/Buffer.h:54-88)
// (compiled with -fno-access-control: the scalar kernels are private static members)
#include "CommonLib/CommonDef.h"
#include "CommonLib/Unit.h"
#include "CommonLib/Buffer.h"
#include "CommonLib/RdCost.h"
#include "CommonLib/InterpolationFilter.h"
#include "CommonLib/TrQuant.h"
#include "CommonLib/TrQuant_EMT.h"
#include "CommonLib/Rom.h"
#include "CommonLib/AffineGradientSearch.h"

#include <cstring>
#include <cstdint>

extern FwdTrans *fastFwdTrans[NUM_TRANS_TYPE][g_numTransformMatrixSizes];
extern InvTrans *fastInvTrans[NUM_TRANS_TYPE][g_numTransformMatrixSizes];

namespace
{
RdCost              *g_rd       = nullptr;   // ctor runs RdCost::init() -> scalar table + initRdCostX86()
InterpolationFilter *g_ifSimd   = nullptr;
InterpolationFilter *g_ifScalar = nullptr;
AffineGradientSearch *g_ags     = nullptr;
bool                 g_bufOpsInit = false;

void ensureInit()
{
  if( g_rd ) return;
  g_rd       = new RdCost();
  g_ifScalar = new InterpolationFilter();
  g_ifSimd   = new InterpolationFilter();
  g_ifSimd->initInterpolationFilter( true );
  g_ags      = new AffineGradientSearch();
#if ENABLE_SIMD_OPT_BUFFER && defined( TARGET_SIMD_X86 )
  g_pelBufOP.initPelBufOpsX86();
#endif
  g_bufOpsInit = true;
}
}   // namespace

extern "C"
{

int ref_version() { return 93; }

// kind: 0 = SAD, 1 = SATD (HAD), 2 = SSE.  simd: 1 = whatever RdCost::init() installed on this CPU
// (AVX2 here), 0 = the scalar member functions called directly.
// subShift is written into DistParam after setDistParam (which is what InterSearch does through subShiftMode).
uint64_t ref_dist( int kind, int simd, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h,
                   int bitDepth, int subShift )
{
  ensureInit();
  DistParam dp;
  CPelBuf   o( org, orgStride, w, h );
  CPelBuf   c( cur, curStride, w, h );
  if( kind == 2 )
  {
    dp.org = o; dp.cur = c; dp.step = 1; dp.bitDepth = bitDepth; dp.compID = COMPONENT_Y;
    const int idx = isPowerOf2( w ) ? DF_SSE + floorLog2( w ) : DF_SSE;
    dp.distFunc = simd ? RdCost::m_afpDistortFunc[idx] : RdCost::xGetSSE;
    return dp.distFunc( dp );
  }
  g_rd->setDistParam( dp, o, c, bitDepth, COMPONENT_Y, kind == 1 );
  dp.subShift = subShift;
  if( !simd )
  {
    dp.distFunc = kind == 1 ? RdCost::xGetHADs : RdCost::xGetSAD;
  }
  return dp.distFunc( dp );
}

// The real RdCost::getDistPart (RdCost.cpp:411-455) with the slice's chroma distortion weight installed (setDistortionWeight, RdCost.h:151)
uint64_t ref_get_dist_part( int compID, double weight, int dfunc /* 0 SAD, 1 SATD, 2 SSE */, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w,
                            int h, int bitDepth )
{
  ensureInit();
  if( compID != COMPONENT_Y ) g_rd->setDistortionWeight( ( ComponentID ) compID, weight );
  const CPelBuf o( org, orgStride, w, h ), c( cur, curStride, w, h );
#if WCG_EXT
  return g_rd->getDistPart( o, c, bitDepth, ( ComponentID ) compID, dfunc == 0 ? DF_SAD : ( dfunc == 1 ? DF_HAD : DF_SSE ), nullptr );
#else
  return g_rd->getDistPart( o, c, bitDepth, ( ComponentID ) compID, dfunc == 0 ? DF_SAD : ( dfunc == 1 ? DF_HAD : DF_SSE ) );
#endif
}

// RdCost::xGetSADwMask through the mask overload of setDistParam (RdCost.cpp:3488-3511); simd 1: the table entry (x86), 0: the scalar member
uint64_t ref_sad_mask( int simd, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int bitDepth, const int16_t *mask,
                       int maskStride, int stepX, int maskStride2 )
{
  ensureInit();
  DistParam dp;
  CPelBuf   o( org, orgStride, w, h );
  g_rd->setDistParam( dp, o, cur, curStride, mask, maskStride, stepX, maskStride2, bitDepth, COMPONENT_Y );
  if( !simd ) dp.distFunc = RdCost::xGetSADwMask;
  return dp.distFunc( dp );
}

// subShift as chosen by RdCost::setDistParam( ..., subShiftMode, ... ) (RdCost.cpp:238-324)
int ref_subshift_for_mode( int w, int h, int subShiftMode )
{
  ensureInit();
  static int16_t dummy[128 * 128];
  DistParam dp;
  CPelBuf   o( dummy, 128, w, h );
  g_rd->setDistParam( dp, o, dummy, 128, 10, COMPONENT_Y, subShiftMode, 1, false );
  return dp.subShift;
}

// Batched form for timing the reference's own SIMD kernels on the host (cpu_baseline "reference"):
// n candidate positions (offsets into the cur plane) against one org block.
void ref_dist_batch( int kind, int simd, const int16_t *org, int orgStride, const int16_t *curPlane, int curStride,
                     const int32_t *curOffsets, int n, int w, int h, int bitDepth, int subShift, uint64_t *out )
{
  ensureInit();
  DistParam dp;
  CPelBuf   o( org, orgStride, w, h );
  CPelBuf   c( curPlane, curStride, w, h );
  g_rd->setDistParam( dp, o, c, bitDepth, COMPONENT_Y, kind == 1 );
  dp.subShift = subShift;
  if( !simd ) dp.distFunc = kind == 1 ? RdCost::xGetHADs : RdCost::xGetSAD;
  for( int i = 0; i < n; i++ )
  {
    dp.cur.buf = curPlane + curOffsets[i];
    out[i]     = dp.distFunc( dp );
  }
}

// SATD 8x8 grid micro-benchmark as SURVEY.md 8(d) defines it, through the reference's distFunc:
// every 8-aligned 8x8 block of `org` (w x h) against `ref` displaced by (dx,dy) in [-r,r]^2.
// ref must have >= r samples of margin. out[(by*bw+bx)*(2r+1)^2 + (dy+r)*(2r+1) + (dx+r)].
void ref_satd8_grid( int simd, const int16_t *org, int orgStride, const int16_t *ref, int refStride, int w, int h, int r,
                     int bitDepth, uint64_t *out )
{
  ensureInit();
  DistParam dp;
  const int bw = w / 8, bh = h / 8, nd = 2 * r + 1;
  for( int by = 0; by < bh; by++ )
    for( int bx = 0; bx < bw; bx++ )
    {
      CPelBuf o( org + by * 8 * orgStride + bx * 8, orgStride, 8, 8 );
      CPelBuf c( ref + by * 8 * refStride + bx * 8, refStride, 8, 8 );
      g_rd->setDistParam( dp, o, c, bitDepth, COMPONENT_Y, true );
      if( !simd ) dp.distFunc = RdCost::xGetHADs;
      uint64_t *po = out + ( size_t )( by * bw + bx ) * nd * nd;
      for( int dy = -r; dy <= r; dy++ )
        for( int dx = -r; dx <= r; dx++ )
        {
          dp.cur.buf = c.buf + dy * refStride + dx;
          *po++      = dp.distFunc( dp );
        }
    }
}

// MV rate exactly as RdCost::getCostOfVectorWithPredictor (RdCost.h:301-315) computes it.
uint64_t ref_mv_cost( double motionLambda, int predHor, int predVer, int costScale, int x, int y, unsigned imvShift )
{
  ensureInit();
  g_rd->m_motionLambda = motionLambda;
  g_rd->setPredictor( Mv( predHor, predVer ) );
  g_rd->setCostScale( costScale );
  return g_rd->getCostOfVectorWithPredictor( x, y, imvShift );
}

// Public InterpolationFilter::filterHor/filterVer (tap-table selection included).  compID 0 = luma, 1 = chroma (4:2:0).
void ref_if_hor( int simd, int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac,
                 int isLast, int bitDepth, int nFilterIdx, int biMCForDMVR, int useAltHpelIf )
{
  ensureInit();
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  ( simd ? g_ifSimd : g_ifScalar )->filterHor( ComponentID( compID ), src, srcStride, dst, dstStride, w, h, frac, isLast != 0, CHROMA_420, clp,
                                               nFilterIdx, biMCForDMVR != 0, useAltHpelIf != 0 );
}

void ref_if_ver( int simd, int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac,
                 int isFirst, int isLast, int bitDepth, int nFilterIdx, int biMCForDMVR, int useAltHpelIf )
{
  ensureInit();
  ClpRng clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  ( simd ? g_ifSimd : g_ifScalar )->filterVer( ComponentID( compID ), src, srcStride, dst, dstStride, w, h, frac, isFirst != 0, isLast != 0, CHROMA_420,
                                               clp, nFilterIdx, biMCForDMVR != 0, useAltHpelIf != 0 );
}

// Raw pointer-table entries m_filterHor/Ver[tapIdx][isFirst][isLast] with caller-supplied taps (InterpolationFilter.h:93-95).
void ref_if_raw( int simd, int vertical, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride,
                 int w, int h, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  ensureInit();
  ClpRng clp; clp.min = clipMin; clp.max = clipMax; clp.bd = bitDepth; clp.n = 0;
  InterpolationFilter *f   = simd ? g_ifSimd : g_ifScalar;
  const int            idx = taps == 8 ? 0 : taps == 4 ? 1 : 2;
  ( vertical ? f->m_filterVer : f->m_filterHor )[idx][isFirst != 0][isLast != 0]( clp, src, srcStride, dst, dstStride, w, h, coeff, biMCForDMVR != 0 );
}

void ref_if_copy( int simd, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h,
                  int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  ensureInit();
  ClpRng clp; clp.min = clipMin; clp.max = clipMax; clp.bd = bitDepth; clp.n = 0;
  ( simd ? g_ifSimd : g_ifScalar )->m_filterCopy[isFirst != 0][isLast != 0]( clp, src, srcStride, dst, dstStride, w, h, biMCForDMVR != 0 );
}

// type: 0 DCT2, 1 DCT8, 2 DST7 (TypeDef.h TransType); sizeIdx = log2(N) - 1.  Returns -1 for the nullptr slots.
int ref_fwd_trans( int type, int sizeIdx, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2 )
{
  if( !fastFwdTrans[type][sizeIdx] ) return -1;
  fastFwdTrans[type][sizeIdx]( src, dst, shift, line, skip1, skip2 );
  return 0;
}

int ref_inv_trans( int type, int sizeIdx, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2, int clipMin, int clipMax )
{
  if( !fastInvTrans[type][sizeIdx] ) return -1;
  fastInvTrans[type][sizeIdx]( src, dst, shift, line, skip1, skip2, clipMin, clipMax );
  return 0;
}

// Copies the N x N forward (dir 0) / inverse (dir 1) core matrix into out (row-major int16).
int ref_tr_matrix( int type, int n, int dir, int16_t *out )
{
  const TMatrixCoeff *m = nullptr;
  switch( type * 100 + n )
  {
  case   2: m = g_trCoreDCT2P2 [dir][0]; break;
  case   4: m = g_trCoreDCT2P4 [dir][0]; break;
  case   8: m = g_trCoreDCT2P8 [dir][0]; break;
  case  16: m = g_trCoreDCT2P16[dir][0]; break;
  case  32: m = g_trCoreDCT2P32[dir][0]; break;
  case  64: m = g_trCoreDCT2P64[dir][0]; break;
  case 104: m = g_trCoreDCT8P4 [dir][0]; break;
  case 108: m = g_trCoreDCT8P8 [dir][0]; break;
  case 116: m = g_trCoreDCT8P16[dir][0]; break;
  case 132: m = g_trCoreDCT8P32[dir][0]; break;
  case 204: m = g_trCoreDST7P4 [dir][0]; break;
  case 208: m = g_trCoreDST7P8 [dir][0]; break;
  case 216: m = g_trCoreDST7P16[dir][0]; break;
  case 232: m = g_trCoreDST7P32[dir][0]; break;
  default: return -1;
  }
  memcpy( out, m, sizeof( TMatrixCoeff ) * n * n );
  return 0;
}

// Luma / chroma tap tables (InterpolationFilter.cpp:57-330), for checking our own tables.
void ref_if_taps( int which, int frac, int16_t *out8 )
{
  memset( out8, 0, 8 * sizeof( int16_t ) );
  switch( which )
  {
  case 0: memcpy( out8, InterpolationFilter::m_lumaFilter[frac], 16 ); break;
  case 1: memcpy( out8, InterpolationFilter::m_lumaFilter4x4[frac], 16 ); break;
  case 2: memcpy( out8, InterpolationFilter::m_chromaFilter[frac], 8 ); break;
  case 3: memcpy( out8, InterpolationFilter::m_bilinearFilterPrec4[frac], 4 ); break;
  case 4: memcpy( out8, InterpolationFilter::m_lumaAltHpelIFilter, 16 ); break;
  case 5: memcpy( out8, InterpolationFilter::m_bilinearFilter[frac], 4 ); break;
  }
}

// AffineGradientSearch pointer members (AffineGradientSearch.h:50-54). simd selects the X86 install.
void ref_sobel( int simd, int vertical, const int16_t *pred, int predStride, int32_t *deriv, int derivStride, int w, int h )
{
  ensureInit();
  static AffineGradientSearch *scalar = nullptr;
  if( !scalar )
  {
    scalar = new AffineGradientSearch();
    scalar->m_HorizontalSobelFilter = AffineGradientSearch::xHorizontalSobelFilter;
    scalar->m_VerticalSobelFilter   = AffineGradientSearch::xVerticalSobelFilter;
    scalar->m_EqualCoeffComputer    = AffineGradientSearch::xEqualCoeffComputer;
  }
  AffineGradientSearch *a = simd ? g_ags : scalar;
  ( vertical ? a->m_VerticalSobelFilter : a->m_HorizontalSobelFilter )( const_cast<Pel *>( pred ), predStride, deriv, derivStride, w, h );
}

void ref_equal_coeff( int simd, const int16_t *resi, int resiStride, int32_t **deriv, int derivStride, int64_t ( *eq )[7], int w, int h, int b6Param )
{
  ensureInit();
  static AffineGradientSearch *scalar = nullptr;
  if( !scalar )
  {
    scalar = new AffineGradientSearch();
    scalar->m_EqualCoeffComputer = AffineGradientSearch::xEqualCoeffComputer;
  }
  ( simd ? g_ags : scalar )->m_EqualCoeffComputer( const_cast<Pel *>( resi ), resiStride, deriv, derivStride, eq, w, h, b6Param != 0 );
}

// PelBufferOps used by bi-pred ME (Buffer.h:64-81; call site InterSearch.cpp:3320-3326):
//   removeHighFreq: org = 2*org - pred (unclipped, BCW default weight)
void ref_remove_high_freq( int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h )
{
  ensureInit();
  PelBuf  o( org, orgStride, w, h );
  PelBuf  p( const_cast<int16_t *>( pred ), predStride, w, h );
  ClpRng  clp; clp.min = 0; clp.max = 1023; clp.bd = 10; clp.n = 0;
  o.removeHighFreq( p, false, clp );
}

// addAvg: dst = clip( ( src0 + src1 + offset ) >> shift ) on 14-bit intermediates (Buffer.cpp addAvg)
void ref_add_avg( const int16_t *src0, int s0Stride, const int16_t *src1, int s1Stride, int16_t *dst, int dstStride, int w, int h, int bitDepth )
{
  ensureInit();
  PelBuf  d( dst, dstStride, w, h );
  CPelBuf a( src0, s0Stride, w, h );
  CPelBuf b( src1, s1Stride, w, h );
  ClpRng  clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  d.addAvg( a, b, clp );
}

// BCW: removeWeightHighFreq through g_pelBufOP (simd 1: x86 entries, 0: the scalar defaults of a fresh PelBufferOps) and addWeightedAvg (Buffer.cpp:365-397)
void ref_remove_weight_high_freq( int simd, int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h, int bcwWeight )
{
  ensureInit();
  const PelBufferOps saved = g_pelBufOP;
  if( !simd ) g_pelBufOP = PelBufferOps();
  PelBuf  o( org, orgStride, w, h );
  PelBuf  p( const_cast<int16_t *>( pred ), predStride, w, h );
  ClpRng  clp; clp.min = 0; clp.max = 1023; clp.bd = 10; clp.n = 0;
  o.removeWeightHighFreq( p, false, clp, ( int8_t ) bcwWeight );
  g_pelBufOP = saved;
}

void ref_add_weighted_avg( const int16_t *src0, int s0Stride, const int16_t *src1, int s1Stride, int16_t *dst, int dstStride, int w, int h, int bitDepth, int bcwIdx )
{
  ensureInit();
  PelBuf  d( dst, dstStride, w, h );
  CPelBuf a( src0, s0Stride, w, h );
  CPelBuf b( src1, s1Stride, w, h );
  ClpRng  clp; clp.min = 0; clp.max = ( 1 << bitDepth ) - 1; clp.bd = bitDepth; clp.n = 0;
  d.addWeightedAvg( a, b, clp, ( int8_t ) bcwIdx );
}

int ref_bcw_weight( int bcwIdx ) { return g_BcwWeights[bcwIdx]; }

}   // extern "C"
