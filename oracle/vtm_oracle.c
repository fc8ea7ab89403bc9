/*
 * vtm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the VTM 9.3 inter motion-estimation + transform/quant hot path
 * (SURVEY.md section 8a).  It is the *checker* for the HIP kernels: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (vtm_amd/, libvtmhip.so) never does.
 *
 * Parity status: PINNED.  Every function below is compared against the real reference compiled from
 * /root/reference by oracle/Makefile.ref (oracle/_ref/libvtmref.so, tests/test_oracle_vs_ref.py) and
 * against the golden vectors that tests/golden/gen_golden.py recorded from that build.
 *
 * Each function cites the reference file:line whose arithmetic it restates.  Paths are relative to
 * /root/reference/source/Lib.  Types: Pel = int16_t, TCoeff = int32_t, Distortion = uint64_t
 * (CommonLib/TypeDef.h:259-270, high-bit-depth OFF).
 */
#include "vtm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VO_MAX_TB 64

static inline int vo_abs( int v ) { return v < 0 ? -v : v; }
static inline int vo_clip3( int lo, int hi, int v ) { return v < lo ? lo : ( v > hi ? hi : v ); }
static inline int vo_floor_log2( unsigned v )
{
  int r = -1;
  while( v ) { v >>= 1; r++; }
  return r;
}

/* ------------------------------------------------------------------------------------------------
 * K1  SAD   CommonLib/RdCost.cpp:493-528 (generic), :530-1003 (width-specialised, same arithmetic),
 *           SIMD x86/RdCostX86.h:210-301.  Rows are visited with step 1<<subShift, the sum is shifted
 *           back left; DISTORTION_PRECISION_ADJUSTMENT is 0 (TypeDef.h:228-233).  The scalar early
 *           exit (:516-519) is not restated: the SIMD path never takes it and callers compare with '<'.
 * ------------------------------------------------------------------------------------------------ */
uint64_t vo_sad( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift )
{
  const int step = 1 << subShift;
  uint64_t  sum  = 0;
  for( int y = 0; y < h; y += step )
  {
    const int16_t *o = org + ( ptrdiff_t ) y * orgStride;
    const int16_t *c = cur + ( ptrdiff_t ) y * curStride;
    for( int x = 0; x < w; x++ )
    {
      sum += ( uint64_t ) vo_abs( ( int ) o[x] - ( int ) c[x] );
    }
  }
  return sum << subShift;
}

/* RdCost::xGetSADwMask (CommonLib/RdCost.cpp:3513-3549; GEO merge estimation, EncCu.cpp:2930-2960): |org - cur| weighted by a mask that is
 * walked with a per-sample step (+1 / -1) and two per-row strides.  The x86 version (x86/RdCostX86.h:2065-2152) addresses the mask
 * as row * maskStride (+ a reversed load for stepX == -1); both agree for the callers' maskStride2 == -stepX * width. */
uint64_t vo_sad_mask( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift, const int16_t *mask,
                      int maskStride, int stepX, int maskStride2 )
{
  const int step = 1 << subShift;
  uint64_t  sum  = 0;
  for( int y = 0; y < h; y += step )
  {
    for( int x = 0; x < w; x++ )
    {
      sum += ( uint64_t )( int64_t )( vo_abs( ( int ) org[( ptrdiff_t ) y * orgStride + x] - ( int ) cur[( ptrdiff_t ) y * curStride + x] ) * ( int ) *mask );
      mask += stepX;
    }
    mask += maskStride * step;
    mask += maskStride2;
  }
  return sum << subShift;
}

/* InterpolationFilter::xWeightedGeoBlk (CommonLib/InterpolationFilter.cpp:902-957): blend of the two GEO partitions' 14-bit predictions with the
 * 0..8 weights of one prestored mask plane.  The reference walks the plane with stepX per sample and stepY at the end of a row; here the
 * row advance is given whole: weightStride = width * stepX + stepY (what the x86 version uses, x86/InterpolationFilterX86.h:1366-1389). */
void vo_weighted_geo_blk( const int16_t *src0, int src0Stride, const int16_t *src1, int src1Stride, int16_t *dst, int dstStride, int w, int h,
                          const int16_t *weight, int stepX, int weightStride, int bitDepth, int clipMin, int clipMax )
{
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
  const int shift    = headRoom + 3;
  const int offset   = ( 1 << ( shift - 1 ) ) + ( 8192 << 3 );
  for( int y = 0; y < h; y++ )
  {
    const int16_t *wr = weight + ( ptrdiff_t ) y * weightStride;
    for( int x = 0; x < w; x++ )
    {
      const int wt = wr[( ptrdiff_t ) x * stepX];
      const int v  = ( wt * ( int ) src0[( ptrdiff_t ) y * src0Stride + x] + ( 8 - wt ) * ( int ) src1[( ptrdiff_t ) y * src1Stride + x] + offset ) >> shift;
      dst[( ptrdiff_t ) y * dstStride + x] = ( int16_t ) ( v < clipMin ? clipMin : v > clipMax ? clipMax : v );
    }
  }
}

/* RdCost::setDistParam subShift rule, CommonLib/RdCost.cpp:289-323 */
int vo_subshift_for_mode( int w, int h, int subShiftMode )
{
  if( subShiftMode == 1 )
  {
    if( h > 32 && ( h & 15 ) == 0 ) return 4;
    if( h > 16 && ( h & 7 ) == 0 ) return 3;
    if( h > 8 && ( h & 3 ) == 0 ) return 2;
    if( ( h & 1 ) == 0 ) return 1;
    return 0;
  }
  if( subShiftMode == 2 ) return ( h > 8 && w <= 64 ) ? 1 : 0;
  if( subShiftMode == 3 ) return ( h > 8 ) ? 1 : 0;
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K4  SSE   CommonLib/RdCost.cpp:1783-1814: per-addend (d*d) >> 0 in 32-bit, 64-bit sum.
 * ------------------------------------------------------------------------------------------------ */
uint64_t vo_sse( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h )
{
  uint64_t sum = 0;
  for( int y = 0; y < h; y++ )
  {
    for( int x = 0; x < w; x++ )
    {
      const int32_t d = ( int32_t ) org[( ptrdiff_t ) y * orgStride + x] - ( int32_t ) cur[( ptrdiff_t ) y * curStride + x];
      sum += ( uint64_t )( ( uint32_t ) d * ( uint32_t ) d );   /* |d| > 46340 overflows the reference's int product (UB there); we define it as 32-bit wrap */
    }
  }
  return sum;
}

/* ------------------------------------------------------------------------------------------------
 * K2  SATD  CommonLib/RdCost.cpp:2140-2934.
 *   One tile = 2-D Walsh-Hadamard of (org - cur); t = sum|coef| - |dc| + (|dc| >> 2)
 *   (JVET_R0164_MEAN_SCALED_SATD, TypeDef.h:62); per-tile normalisation:
 *     2x2: t (only the dc term is >>2, :2154-2162)   4x4: (t+1)>>1 (:2258-2262)   8x8: (t+2)>>2 (:2359-2363)
 *     16x8 / 8x16: (int)(t / sqrt(16.0*8) * 2) (:2513,2654)    8x4 / 4x8: (int)(t / sqrt(4.0*8) * 2) (:2731,2814)
 *   Tile selection: xGetHADs :2819-2934.
 *   The butterfly order is free: only the multiset of |coef| and coef[0][0] = sum(diff) matter.
 * ------------------------------------------------------------------------------------------------ */
static int vo_had_tile_abs_sum( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int tw, int th )
{
  int32_t m[16 * 16];
  for( int y = 0; y < th; y++ )
    for( int x = 0; x < tw; x++ )
      m[y * tw + x] = ( int32_t ) org[( ptrdiff_t ) y * orgStride + x] - ( int32_t ) cur[( ptrdiff_t ) y * curStride + x];

  /* in-place WHT along x, then along y (natural/Hadamard order, dc at [0]) */
  for( int y = 0; y < th; y++ )
    for( int len = 1; len < tw; len <<= 1 )
      for( int i = 0; i < tw; i += len << 1 )
        for( int j = i; j < i + len; j++ )
        {
          const int32_t a = m[y * tw + j], b = m[y * tw + j + len];
          m[y * tw + j]       = a + b;
          m[y * tw + j + len] = a - b;
        }
  for( int x = 0; x < tw; x++ )
    for( int len = 1; len < th; len <<= 1 )
      for( int i = 0; i < th; i += len << 1 )
        for( int j = i; j < i + len; j++ )
        {
          const int32_t a = m[j * tw + x], b = m[( j + len ) * tw + x];
          m[j * tw + x]           = a + b;
          m[( j + len ) * tw + x] = a - b;
        }
  int t = 0;
  for( int i = 0; i < tw * th; i++ ) t += vo_abs( m[i] );
  const int dc = vo_abs( m[0] );
  return t - dc + ( dc >> 2 );
}

static uint64_t vo_had_tile( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int tw, int th )
{
  const int t = vo_had_tile_abs_sum( org, orgStride, cur, curStride, tw, th );
  if( tw == 2 && th == 2 ) return ( uint64_t ) t;
  if( tw == 4 && th == 4 ) return ( uint64_t )( ( t + 1 ) >> 1 );
  if( tw == 8 && th == 8 ) return ( uint64_t )( ( t + 2 ) >> 2 );
  if( tw * th == 128 ) return ( uint64_t )( int ) ( t / sqrt( 16.0 * 8 ) * 2 );
  /* 8x4, 4x8 */
  return ( uint64_t )( int ) ( t / sqrt( 4.0 * 8 ) * 2 );
}

/* Tile shape chosen by xGetHADs (RdCost.cpp:2837-2931).  Returns 0 and sets *tw,*th; -1 = "Invalid size". */
int vo_satd_tile_shape( int w, int h, int *tw, int *th )
{
  if( w > h && ( h & 7 ) == 0 && ( w & 15 ) == 0 ) { *tw = 16; *th = 8; }
  else if( w < h && ( w & 7 ) == 0 && ( h & 15 ) == 0 ) { *tw = 8; *th = 16; }
  else if( w > h && ( h & 3 ) == 0 && ( w & 7 ) == 0 ) { *tw = 8; *th = 4; }
  else if( w < h && ( w & 3 ) == 0 && ( h & 7 ) == 0 ) { *tw = 4; *th = 8; }
  else if( ( h % 8 == 0 ) && ( w % 8 == 0 ) ) { *tw = 8; *th = 8; }
  else if( ( h % 4 == 0 ) && ( w % 4 == 0 ) ) { *tw = 4; *th = 4; }
  else if( ( h % 2 == 0 ) && ( w % 2 == 0 ) ) { *tw = 2; *th = 2; }
  else return -1;
  return 0;
}

uint64_t vo_satd( const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h )
{
  int tw, th;
  if( vo_satd_tile_shape( w, h, &tw, &th ) ) return UINT64_MAX;
  uint64_t sum = 0;
  for( int y = 0; y < h; y += th )
    for( int x = 0; x < w; x += tw )
      sum += vo_had_tile( org + ( ptrdiff_t ) y * orgStride + x, orgStride, cur + ( ptrdiff_t ) y * curStride + x, curStride, tw, th );
  return sum;
}

/* SATD 8x8 block grid x (2r+1)^2 displacements: the SURVEY.md 8(d) micro-benchmark, same output layout as
 * ref_satd8_grid / vtmhip_satd8_grid. */
void vo_satd8_grid( const int16_t *org, int orgStride, const int16_t *ref, int refStride, int w, int h, int r, uint64_t *out )
{
  const int bw = w / 8, bh = h / 8, nd = 2 * r + 1;
  for( int by = 0; by < bh; by++ )
    for( int bx = 0; bx < bw; bx++ )
    {
      const int16_t *o  = org + ( ptrdiff_t ) by * 8 * orgStride + bx * 8;
      const int16_t *c  = ref + ( ptrdiff_t ) by * 8 * refStride + bx * 8;
      uint64_t      *po = out + ( size_t )( by * bw + bx ) * nd * nd;
      for( int dy = -r; dy <= r; dy++ )
        for( int dx = -r; dx <= r; dx++ )
          *po++ = vo_had_tile( o, orgStride, c + ( ptrdiff_t ) dy * refStride + dx, refStride, 8, 8 );
    }
}

/* ------------------------------------------------------------------------------------------------
 * MV rate  CommonLib/RdCost.h:301-315.  bits(v): exp-Golomb length of the signed value; the loop
 * form strips MAX_CU_DEPTH(7) bits at a time while temp > MAX_CU_SIZE(128).
 * cost = (uint64)(motionLambda * bits), fp64 multiply then truncation.
 * ------------------------------------------------------------------------------------------------ */
static unsigned vo_eg_bits( int v )
{
  unsigned len = 1;
  unsigned t   = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  while( t > 128 )
  {
    len += 14;
    t >>= 7;
  }
  return len + ( ( unsigned ) vo_floor_log2( t ) << 1 );
}

unsigned vo_mv_bits( const vo_mvcost_t *mc, int x, int y, unsigned imvShift )
{
  return vo_eg_bits( ( ( x << mc->costScale ) - mc->predHor ) >> imvShift ) + vo_eg_bits( ( ( y << mc->costScale ) - mc->predVer ) >> imvShift );
}

uint64_t vo_mv_cost( const vo_mvcost_t *mc, int x, int y, unsigned imvShift )
{
  return ( uint64_t ) ( mc->motionLambda * vo_mv_bits( mc, x, y, imvShift ) );
}

/* ------------------------------------------------------------------------------------------------
 * K7/K8 Interpolation  CommonLib/InterpolationFilter.cpp:398-525 (copy), :548-651 (FIR),
 *       tap tables :57-330 (H.266 8.5.6.3 tables 27/28 constants), dispatch :749-891.
 * ------------------------------------------------------------------------------------------------ */
#define VO_IF_PREC 14
#define VO_IF_OFFS ( 1 << ( VO_IF_PREC - 1 ) )
#define VO_IF_FILTER_PREC 6

/* H.266 luma 1/16-sample 8-tap filter (InterpolationFilter.cpp:77-95) */
const int16_t vo_luma_filter[16][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },       { 0, 1, -3, 63, 4, -2, 1, 0 },     { -1, 2, -5, 62, 8, -3, 1, 0 },    { -1, 3, -8, 60, 13, -4, 1, 0 },
  { -1, 4, -10, 58, 17, -5, 1, 0 },  { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
  { -1, 4, -11, 40, 40, -11, 4, -1 },{ -1, 4, -10, 34, 45, -11, 4, -1 },{ -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
  { 0, 1, -5, 17, 58, -10, 4, -1 },  { 0, 1, -4, 13, 60, -8, 3, -1 },   { 0, 1, -3, 8, 62, -5, 2, -1 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };

/* 6-tap variant used for 4x4 (affine sub-)blocks (InterpolationFilter.cpp:57-75) */
const int16_t vo_luma_filter_4x4[16][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },      { 0, 1, -3, 63, 4, -2, 1, 0 },    { 0, 1, -5, 62, 8, -3, 1, 0 },    { 0, 2, -8, 60, 13, -4, 1, 0 },
  { 0, 3, -10, 58, 17, -5, 1, 0 },  { 0, 3, -11, 52, 26, -8, 2, 0 },  { 0, 2, -9, 47, 31, -10, 3, 0 },  { 0, 3, -11, 45, 34, -10, 3, 0 },
  { 0, 3, -11, 40, 40, -11, 3, 0 }, { 0, 3, -10, 34, 45, -11, 3, 0 }, { 0, 3, -10, 31, 47, -9, 2, 0 },  { 0, 2, -8, 26, 52, -11, 3, 0 },
  { 0, 1, -5, 17, 58, -10, 3, 0 },  { 0, 1, -4, 13, 60, -8, 2, 0 },   { 0, 1, -3, 8, 62, -5, 1, 0 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };

const int16_t vo_luma_alt_hpel[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };   /* InterpolationFilter.cpp:181 */

/* H.266 chroma 1/32-sample 4-tap filter (InterpolationFilter.cpp:182-216) */
const int16_t vo_chroma_filter[32][4] = {
  { 0, 64, 0, 0 },    { -1, 63, 2, 0 },   { -2, 62, 4, 0 },   { -2, 60, 7, -1 },  { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 },
  { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 },
  { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
  { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 }, { -2, 10, 58, -2 }, { -1, 7, 60, -2 },  { 0, 4, 62, -2 },   { 0, 2, 63, -1 } };

/* bilinear, 4-bit precision: { 16 - f, f } (InterpolationFilter.cpp:312-330); DMVR / BDOF only */
static void vo_bilinear_prec4( int frac, int16_t c[2] ) { c[0] = ( int16_t )( 16 - frac ); c[1] = ( int16_t ) frac; }

void vo_if_copy( int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int bitDepth, int clipMin,
                 int clipMax, int biMCForDMVR )
{
  const int headRoom = ( VO_IF_PREC - bitDepth ) > 2 ? ( VO_IF_PREC - bitDepth ) : 2;   /* IF_INTERNAL_FRAC_BITS, InterpolationFilter.h:54 */
  for( int y = 0; y < h; y++, src += srcStride, dst += dstStride )
  {
    for( int x = 0; x < w; x++ )
    {
      const int s = src[x];
      if( isFirst == isLast )
      {
        dst[x] = ( int16_t ) s;
      }
      else if( biMCForDMVR )
      {
        /* both directions identical: :417-447 and :470-500 (IF_INTERNAL_PREC_BILINEAR = 10) */
        if( bitDepth > 10 )
        {
          const int sh = bitDepth - 10;
          dst[x]       = ( int16_t )( ( s + ( 1 << ( sh - 1 ) ) ) >> sh );
        }
        else
        {
          dst[x] = ( int16_t )( s << ( 10 - bitDepth ) );
        }
      }
      else if( isFirst )
      {
        const int16_t v = ( int16_t )( s << headRoom );   /* Pel val = leftShift_round(src, shift) */
        dst[x]          = ( int16_t )( v - ( int16_t ) VO_IF_OFFS );
      }
      else
      {
        const int16_t v = ( int16_t )( ( s + VO_IF_OFFS + ( 1 << ( headRoom - 1 ) ) ) >> headRoom );
        dst[x]          = ( int16_t ) vo_clip3( clipMin, clipMax, v );
      }
    }
  }
}

void vo_if_filter( int vertical, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h,
                   const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  const int cStride  = vertical ? srcStride : 1;
  const int headRoom = ( VO_IF_PREC - bitDepth ) > 2 ? ( VO_IF_PREC - bitDepth ) : 2;
  int       shift    = VO_IF_FILTER_PREC;
  int       offset;
  src -= ( taps / 2 - 1 ) * cStride;
  if( isLast )
  {
    shift += isFirst ? 0 : headRoom;
    offset = 1 << ( shift - 1 );
    offset += isFirst ? 0 : ( VO_IF_OFFS << VO_IF_FILTER_PREC );
  }
  else
  {
    shift -= isFirst ? headRoom : 0;
    offset = isFirst ? -( VO_IF_OFFS << shift ) : 0;
  }
  if( biMCForDMVR )
  {
    shift  = isFirst ? 4 - ( 10 - bitDepth ) : 4;   /* IF_FILTER_PREC_BILINEAR - (IF_INTERNAL_PREC_BILINEAR - bd) */
    offset = 1 << ( shift - 1 );
  }
  for( int y = 0; y < h; y++, src += srcStride, dst += dstStride )
  {
    for( int x = 0; x < w; x++ )
    {
      int sum = 0;
      for( int k = 0; k < taps; k++ ) sum += ( int ) src[x + k * cStride] * ( int ) coeff[k];
      int16_t val = ( int16_t )( ( sum + offset ) >> shift );   /* truncation to Pel happens before the clip (:645) */
      if( isLast ) val = ( int16_t ) vo_clip3( clipMin, clipMax, val );
      dst[x] = val;
    }
  }
}

/* public filterHor (InterpolationFilter.cpp:749-812): isFirst is implicitly true */
void vo_if_hor( int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac, int isLast, int bitDepth,
                int nFilterIdx, int biMCForDMVR, int useAltHpelIf )
{
  const int clipMax = ( 1 << bitDepth ) - 1;
  int16_t   bil[2];
  if( frac == 0 && nFilterIdx < 2 )
  {
    vo_if_copy( 1, isLast, src, srcStride, dst, dstStride, w, h, bitDepth, 0, clipMax, biMCForDMVR );
    return;
  }
  const int16_t *c;
  int            taps = 8;
  if( compID == 0 )
  {
    if( nFilterIdx == 1 ) { vo_bilinear_prec4( frac, bil ); c = bil; taps = 2; }
    else if( nFilterIdx == 2 ) c = vo_luma_filter_4x4[frac];
    else if( frac == 8 && useAltHpelIf ) c = vo_luma_alt_hpel;
    else if( ( w == 4 && h == 4 ) || ( w == 4 && h == 4 + 8 - 1 ) ) c = vo_luma_filter_4x4[frac];
    else c = vo_luma_filter[frac];
  }
  else
  {
    c    = vo_chroma_filter[frac];   /* 4:2:0: frac << (1 - csx) with csx = 1 */
    taps = 4;
  }
  vo_if_filter( 0, taps, 1, isLast, src, srcStride, dst, dstStride, w, h, c, bitDepth, 0, clipMax, biMCForDMVR );
}

/* public filterVer (InterpolationFilter.cpp:832-891) */
void vo_if_ver( int compID, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int frac, int isFirst, int isLast,
                int bitDepth, int nFilterIdx, int biMCForDMVR, int useAltHpelIf )
{
  const int clipMax = ( 1 << bitDepth ) - 1;
  int16_t   bil[2];
  if( frac == 0 && nFilterIdx < 2 )
  {
    vo_if_copy( isFirst, isLast, src, srcStride, dst, dstStride, w, h, bitDepth, 0, clipMax, biMCForDMVR );
    return;
  }
  const int16_t *c;
  int            taps = 8;
  if( compID == 0 )
  {
    if( nFilterIdx == 1 ) { vo_bilinear_prec4( frac, bil ); c = bil; taps = 2; }
    else if( nFilterIdx == 2 ) c = vo_luma_filter_4x4[frac];
    else if( frac == 8 && useAltHpelIf ) c = vo_luma_alt_hpel;
    else if( w == 4 && h == 4 ) c = vo_luma_filter_4x4[frac];
    else c = vo_luma_filter[frac];
  }
  else
  {
    c    = vo_chroma_filter[frac];
    taps = 4;
  }
  vo_if_filter( 1, taps, isFirst, isLast, src, srcStride, dst, dstStride, w, h, c, bitDepth, 0, clipMax, biMCForDMVR );
}

/* ------------------------------------------------------------------------------------------------
 * K9 Transforms.  Core matrices = H.266 8.7.4 transMatrix constants (CommonLib/RomTr.cpp:432-..., 6-bit set,
 * Rom.h:79-84,115-130).  We do not carry the tables: DCT-2 is generated from the 63 distinct magnitudes
 * c[j] ~ 64*sqrt(2)*cos(j*pi/128) of the 64-point matrix (smaller sizes are its row-subsampled top-left
 * corners), DST-7 from the first row of each size (entries are +-row0[fold((2k+1)(n+1) mod (2N+1))]),
 * DCT-8[k][n] = (-1)^k * DST-7[k][N-1-n].  tests/test_oracle_vs_ref.py checks all 14 matrices against
 * g_trCore*.  The reference's "fast" butterflies (TrQuant_EMT.cpp:51-1860) are exact refactorings of the
 * plain matrix product (no intermediate rounding), so the product below is bit-identical.
 * ------------------------------------------------------------------------------------------------ */
static const int8_t vo_dct2_mag[64] = { 0, /* j = 1..63 */
  91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80, 79, 78, 77, 75, 73, 73, 71, 70, 69, 67, 65, 64,
  62, 61, 59, 57, 56, 54, 52, 50, 48, 46, 44, 43, 41, 38, 37, 36, 33, 31, 28, 25, 24, 22, 20, 18, 15, 13, 11, 9,  7,  4,  2 };

static const int8_t vo_dst7_row0_4[4]   = { 29, 55, 74, 84 };
static const int8_t vo_dst7_row0_8[8]   = { 17, 32, 46, 60, 71, 78, 85, 86 };
static const int8_t vo_dst7_row0_16[16] = { 8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88 };
static const int8_t vo_dst7_row0_32[32] = { 4,  9,  13, 17, 21, 26, 30, 34, 38, 42, 46, 50, 53, 56, 60, 63,
                                            66, 68, 72, 74, 77, 78, 80, 82, 84, 85, 86, 87, 88, 89, 90, 90 };

int vo_tr_matrix( int type, int n, int16_t *out )
{
  if( type == VO_DCT2 )
  {
    if( n != 2 && n != 4 && n != 8 && n != 16 && n != 32 && n != 64 ) return -1;
    const int s = 64 / n;
    for( int k = 0; k < n; k++ )
      for( int x = 0; x < n; x++ )
      {
        if( k == 0 ) { out[x] = 64; continue; }
        int j    = ( ( k * s ) * ( 2 * x + 1 ) ) % 256;   /* angle in units of pi/128 */
        int sign = 1;
        if( j > 128 ) j = 256 - j;
        if( j > 64 ) { j = 128 - j; sign = -1; }
        out[k * n + x] = ( int16_t )( sign * vo_dct2_mag[j] );
      }
    return 0;
  }
  const int8_t *row0 = n == 4 ? vo_dst7_row0_4 : n == 8 ? vo_dst7_row0_8 : n == 16 ? vo_dst7_row0_16 : n == 32 ? vo_dst7_row0_32 : NULL;
  if( !row0 || ( type != VO_DST7 && type != VO_DCT8 ) ) return -1;
  const int p = 2 * n + 1;
  for( int k = 0; k < n; k++ )
    for( int x = 0; x < n; x++ )
    {
      int m    = ( ( 2 * k + 1 ) * ( x + 1 ) ) % ( 2 * p );   /* sin(pi*m/p) */
      int sign = 1;
      if( m > p ) { m = 2 * p - m; sign = -1; }
      if( m > n ) m = p - m;
      const int v = m == 0 ? 0 : sign * row0[m - 1];
      if( type == VO_DST7 ) out[k * n + x] = ( int16_t ) v;
      else out[k * n + ( n - 1 - x )] = ( int16_t )( ( k & 1 ) ? -v : v );
    }
  return 0;
}

static const int16_t *vo_matrix( int type, int n )
{
  static int16_t cache[3][7][64 * 64];
  static int     have[3][7];
  const int      li = vo_floor_log2( ( unsigned ) n );
  if( !have[type][li] )
  {
    if( vo_tr_matrix( type, n, cache[type][li] ) ) return NULL;
    have[type][li] = 1;
  }
  return cache[type][li];
}

/* FwdTrans signature (CommonLib/TrQuant.h:53): dst[k*line + j] = (sum_n M[k][n]*src[j*N+n] + rnd) >> shift for
 * j < line-skip1, k < N-skip2; zero elsewhere (_fastForwardMM TrQuant_EMT.cpp:274-323).  int32 wrap-around. */
int vo_fwd_trans( int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2 )
{
  const int16_t *m = vo_matrix( type, n );
  if( !m ) return -1;
  const uint32_t rnd    = shift > 0 ? 1u << ( shift - 1 ) : 0;
  const int      rl     = line - skip1;
  const int      cutoff = n - skip2;
  for( int k = 0; k < n; k++ )
    for( int j = 0; j < line; j++ )
    {
      if( j >= rl || k >= cutoff ) { dst[k * line + j] = 0; continue; }
      uint32_t sum = 0;
      for( int x = 0; x < n; x++ ) sum += ( uint32_t )( ( int32_t ) m[k * n + x] ) * ( uint32_t ) src[j * n + x];
      dst[k * line + j] = ( int32_t )( sum + rnd ) >> shift;
    }
  return 0;
}

/* InvTrans signature (TrQuant.h:54): dst[i*N + j] = clip((sum_{k<N-skip2} src[k*line+i]*M[k][j] + rnd) >> shift),
 * rows i >= line-skip1 zero (_fastInverseMM TrQuant_EMT.cpp:235-271). */
int vo_inv_trans( int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skip1, int skip2, int clipMin, int clipMax )
{
  const int16_t *m = vo_matrix( type, n );
  if( !m ) return -1;
  const uint32_t rnd    = 1u << ( shift - 1 );
  const int      rl     = line - skip1;
  const int      cutoff = n - skip2;
  for( int i = 0; i < line; i++ )
    for( int j = 0; j < n; j++ )
    {
      if( i >= rl ) { dst[i * n + j] = 0; continue; }
      uint32_t sum = 0;
      for( int k = 0; k < cutoff; k++ ) sum += ( uint32_t ) src[k * line + i] * ( uint32_t )( ( int32_t ) m[k * n + j] );
      dst[i * n + j] = vo_clip3( clipMin, clipMax, ( int32_t )( sum + rnd ) >> shift );
    }
  return 0;
}

static int vo_skip( int type, int n ) { return ( type != VO_DCT2 && n == 32 ) ? 16 : ( n > 32 ? n - 32 : 0 ); }

/* TrQuant::xT (CommonLib/TrQuant.cpp:776-851), 2-D and the W==1 / H==1 1-D cases; maxLog2TrDynamicRange = 15,
 * TRANSFORM_MATRIX_SHIFT = 6, COM16_C806_TRANS_PREC = 0.  No LFNST (out of scope). */
int vo_fwd_2d( const int16_t *resi, int stride, int w, int h, int bitDepth, int typeHor, int typeVer, int32_t *coef )
{
  int32_t   block[VO_MAX_TB * VO_MAX_TB], tmp[VO_MAX_TB * VO_MAX_TB];
  const int skipW = vo_skip( typeHor, w ), skipH = vo_skip( typeVer, h );
  if( w > VO_MAX_TB || h > VO_MAX_TB ) return -1;
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ ) block[y * w + x] = resi[( ptrdiff_t ) y * stride + x];
  if( w > 1 && h > 1 )
  {
    const int s1 = vo_floor_log2( w ) + bitDepth + 6 - 15;
    const int s2 = vo_floor_log2( h ) + 6;
    if( vo_fwd_trans( typeHor, w, block, tmp, s1, h, 0, skipW ) ) return -1;
    if( vo_fwd_trans( typeVer, h, tmp, coef, s2, w, skipW, skipH ) ) return -1;
  }
  else if( h == 1 )
  {
    if( vo_fwd_trans( typeHor, w, block, coef, vo_floor_log2( w ) + bitDepth + 6 - 15, 1, 0, skipW ) ) return -1;
  }
  else
  {
    if( vo_fwd_trans( typeVer, h, block, coef, vo_floor_log2( h ) + bitDepth + 6 - 15, 1, 0, skipH ) ) return -1;
  }
  return 0;
}

/* TrQuant::xIT (TrQuant.cpp:853-923) */
int vo_inv_2d( const int32_t *coef, int w, int h, int bitDepth, int typeHor, int typeVer, int16_t *resi, int stride )
{
  int32_t   block[VO_MAX_TB * VO_MAX_TB], tmp[VO_MAX_TB * VO_MAX_TB];
  const int skipW = vo_skip( typeHor, w ), skipH = vo_skip( typeVer, h );
  const int cmin = -( 1 << 15 ), cmax = ( 1 << 15 ) - 1;
  if( w > VO_MAX_TB || h > VO_MAX_TB ) return -1;
  if( w > 1 && h > 1 )
  {
    const int s1 = 6 + 1;
    const int s2 = ( 6 + 15 - 1 ) - bitDepth;
    if( vo_inv_trans( typeVer, h, coef, tmp, s1, w, skipW, skipH, cmin, cmax ) ) return -1;
    if( vo_inv_trans( typeHor, w, tmp, block, s2, h, 0, skipW, cmin, cmax ) ) return -1;
  }
  else if( w == 1 )
  {
    if( vo_inv_trans( typeVer, h, coef, block, ( 6 + 15 - 1 ) - bitDepth + 1, 1, 0, skipH, cmin, cmax ) ) return -1;
  }
  else
  {
    if( vo_inv_trans( typeHor, w, coef, block, ( 6 + 15 - 1 ) - bitDepth + 1, 1, 0, skipW, cmin, cmax ) ) return -1;
  }
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ ) resi[( ptrdiff_t ) y * stride + x] = ( int16_t ) block[y * w + x];
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K10 Scalar quant / dequant, flat scaling list.  CommonLib/Quant.cpp:955-1038 (quant), :357-482 (dequant),
 * scales CommonLib/Rom.cpp:463-473, getTransformShift ChromaFormat.h:111-114,
 * TU::needsBlockSizeTrafoScale = (log2W + log2H) odd.  Sign-bit hiding is not restated (SBH needs the scan).
 * Pinned against the real Quant::quant / Quant::dequant through oracle/ref_shim_me.cpp:ref_quant_dequant.
 * ------------------------------------------------------------------------------------------------ */
static const int vo_quant_scales[2][6]     = { { 26214, 23302, 20560, 18396, 16384, 14564 }, { 18396, 16384, 14564, 13107, 11651, 10280 } };
static const int vo_inv_quant_scales[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };

void vo_quant( const int32_t *coef, int w, int h, int bitDepth, int qpPer, int qpRem, int isIRAP, int isTS, int32_t *qcoef, int32_t *deltaU,
               int32_t *absSum )
{
  const int lw = vo_floor_log2( w ), lh = vo_floor_log2( h );
  const int needSqrt = ( ( lw + lh ) & 1 ) && !isTS;   /* TU::needsBlockSizeTrafoScale: not for transform skip */
  const int scale    = vo_quant_scales[needSqrt][qpRem];
  const int trShift  = 15 - bitDepth - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
  const int qBits    = 14 + qpPer + ( isTS ? 0 : trShift );
  const int64_t add  = ( int64_t )( isIRAP ? 171 : 85 ) << ( qBits - 9 );
  const int qBits8   = qBits - 8;
  int32_t   sum      = 0;
  for( int i = 0; i < w * h; i++ )
  {
    /* the coefficient scan of blocks wider/taller than 32 only covers the 32x32 zero-out region (g_scanOrder is built for
     * min(32, W) x min(32, H), Rom.cpp initROM): positions outside are never visited and keep level 0 (memset :1004) */
    if( ( i % w ) >= 32 || ( i / w ) >= 32 ) { qcoef[i] = 0; if( deltaU ) deltaU[i] = 0; continue; }
    const int32_t c   = coef[i];
    const int64_t t   = ( int64_t ) vo_abs( c ) * scale;
    const int32_t mag = ( int32_t )( ( t + add ) >> qBits );
    if( deltaU ) deltaU[i] = ( int32_t )( ( t - ( ( int64_t ) mag << qBits ) ) >> qBits8 );
    sum += mag;
    qcoef[i] = vo_clip3( -32768, 32767, c < 0 ? -mag : mag );
  }
  *absSum = sum;
}

void vo_dequant( const int32_t *qcoef, int w, int h, int bitDepth, int qpPer, int qpRem, int isTS, int32_t *coef )
{
  const int lw = vo_floor_log2( w ), lh = vo_floor_log2( h );
  const int needSqrt   = ( ( lw + lh ) & 1 ) && !isTS;
  const int trShift    = 15 - bitDepth - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
  const int rightShift = 6 - ( ( isTS ? 0 : trShift ) + qpPer );
  const int scale      = vo_inv_quant_scales[needSqrt][qpRem];
  int       inBits     = 32 + rightShift - 7;
  if( inBits > 16 ) inBits = 16;
  const int32_t inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
  for( int i = 0; i < w * h; i++ )
  {
    const int32_t q = vo_clip3( inMin, inMax, qcoef[i] );
    int32_t       v;
    if( rightShift > 0 ) v = ( int32_t )( ( uint32_t )( q * scale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
    else v = ( int32_t )( ( uint32_t )( q * scale ) << ( -rightShift ) );
    coef[i] = vo_clip3( -32768, 32767, v );
  }
}

/* ------------------------------------------------------------------------------------------------
 * K12 PelBufferOps used by bi-pred ME.
 *   removeHighFreq (CommonLib/Buffer.h:475-520 / Buffer.cpp removeHighFreq, call InterSearch.cpp:3320-3326):
 *     org = 2*org - pred, no clip (ClipForBiPredMEEnabled = 0), BCW default.
 *   addAvg (Buffer.cpp:467-507): dst = clip((a + b + offset) >> shift), shift = headRoom + 1,
 *     offset = (1 << (shift-1)) + 2*IF_INTERNAL_OFFS.
 * ------------------------------------------------------------------------------------------------ */
void vo_remove_high_freq( int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h )
{
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      org[( ptrdiff_t ) y * orgStride + x] = ( int16_t )( 2 * org[( ptrdiff_t ) y * orgStride + x] - pred[( ptrdiff_t ) y * predStride + x] );
}

void vo_add_avg( const int16_t *a, int aStride, const int16_t *b, int bStride, int16_t *dst, int dstStride, int w, int h, int bitDepth )
{
  const int headRoom = ( VO_IF_PREC - bitDepth ) > 2 ? ( VO_IF_PREC - bitDepth ) : 2;
  const int shift    = headRoom + 1;
  const int offset   = ( 1 << ( shift - 1 ) ) + 2 * VO_IF_OFFS;
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      dst[( ptrdiff_t ) y * dstStride + x] =
        ( int16_t ) vo_clip3( 0, ( 1 << bitDepth ) - 1, ( a[( ptrdiff_t ) y * aStride + x] + b[( ptrdiff_t ) y * bStride + x] + offset ) >> shift );
}

/* BCW variants (CommonLib/Buffer.h:417-460, Buffer.cpp:365-397; weights g_BcwWeights = {-2, 3, 4, 5, 10} of 8, Rom.cpp:188-190).
 * removeWeightHighFreq: the bi-pred ME target when the searched list carries weight bcwWeight: (org * 8 - pred * (8 - w)) / w in 16.16 fixed point,
 * unclipped.  addWeightedAvg: src0 * (8 - w1) + src1 * w1 on the 14-bit intermediates. */
void vo_remove_weight_high_freq( int16_t *org, int orgStride, const int16_t *pred, int predStride, int w, int h, int bcwWeight )
{
  const int normalizer = ( ( 1 << 16 ) + ( bcwWeight > 0 ? ( bcwWeight >> 1 ) : -( bcwWeight >> 1 ) ) ) / bcwWeight;
  const int weight0 = normalizer * 8, weight1 = ( 8 - bcwWeight ) * normalizer;
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      org[( ptrdiff_t ) y * orgStride + x] =
        ( int16_t )( ( org[( ptrdiff_t ) y * orgStride + x] * weight0 - pred[( ptrdiff_t ) y * predStride + x] * weight1 + ( 1 << 15 ) ) >> 16 );
}

void vo_add_weighted_avg( const int16_t *a, int aStride, const int16_t *b, int bStride, int16_t *dst, int dstStride, int w, int h, int bitDepth, int w1 )
{
  const int headRoom = ( VO_IF_PREC - bitDepth ) > 2 ? ( VO_IF_PREC - bitDepth ) : 2;
  const int shift    = headRoom + 3;
  const int offset   = ( 1 << ( shift - 1 ) ) + ( VO_IF_OFFS << 3 );
  const int w0       = 8 - w1;
  for( int y = 0; y < h; y++ )
    for( int x = 0; x < w; x++ )
      dst[( ptrdiff_t ) y * dstStride + x] = ( int16_t ) vo_clip3( 0, ( 1 << bitDepth ) - 1,
                                                                  ( a[( ptrdiff_t ) y * aStride + x] * w0 + b[( ptrdiff_t ) y * bStride + x] * w1 + offset ) >> shift );
}

/* LFNST kernels: TrQuant::fwdLfnstNxN / invLfnstNxN (CommonLib/TrQuant.cpp:233-311).  M: the 16 x trSize int8 core matrix of (mode, index) -- data of
 * the standard that the caller supplies (g_lfnst8x8 / g_lfnst4x4, Rom.h:132-133); trSize = 48 for size > 4, else 16. */
void vo_fwd_lfnst( const int32_t *src, int32_t *dst, const int8_t *M, int size, int zeroOutSize )
{
  const int trSize = size > 4 ? 48 : 16;
  for( int j = 0; j < trSize; j++ )
  {
    int coef = 0;
    if( j < zeroOutSize )
    {
      for( int i = 0; i < trSize; i++ ) coef += src[i] * ( int ) M[j * trSize + i];
      coef = ( coef + 64 ) >> 7;
    }
    dst[j] = coef;
  }
}

void vo_inv_lfnst( const int32_t *src, int32_t *dst, const int8_t *M, int size, int zeroOutSize )
{
  const int trSize = size > 4 ? 48 : 16;
  for( int j = 0; j < trSize; j++ )
  {
    int resi = 0;
    for( int i = 0; i < zeroOutSize; i++ ) resi += src[i] * ( int ) M[i * trSize + j];
    dst[j] = vo_clip3( -32768, 32767, ( resi + 64 ) >> 7 );
  }
}

/* ------------------------------------------------------------------------------------------------
 * K11 Affine gradient  CommonLib/AffineGradientSearch.cpp:62-170.
 * ------------------------------------------------------------------------------------------------ */
void vo_sobel( int vertical, const int16_t *p, int ps, int32_t *d, int ds, int w, int h )
{
  for( int j = 1; j < h - 1; j++ )
    for( int k = 1; k < w - 1; k++ )
    {
      const int16_t *c = p + ( ptrdiff_t ) j * ps + k;
      d[j * ds + k]    = vertical ? ( c[ps - 1] - c[-ps - 1] + ( c[ps] << 1 ) - ( c[-ps] << 1 ) + c[ps + 1] - c[-ps + 1] )
                                  : ( c[1 - ps] - c[-1 - ps] + ( c[1] << 1 ) - ( c[-1] << 1 ) + c[1 + ps] - c[-1 + ps] );
    }
  /* border replication; the end state is order-independent: edges copy the adjacent interior sample,
   * corners copy the diagonal interior sample (:77-92, :111-126) */
  for( int j = 1; j < h - 1; j++ )
  {
    d[j * ds]         = d[j * ds + 1];
    d[j * ds + w - 1] = d[j * ds + w - 2];
  }
  for( int k = 1; k < w - 1; k++ )
  {
    d[k]                  = d[ds + k];
    d[( h - 1 ) * ds + k] = d[( h - 2 ) * ds + k];
  }
  d[0]                      = d[ds + 1];
  d[w - 1]                  = d[ds + w - 2];
  d[( h - 1 ) * ds]         = d[( h - 2 ) * ds + 1];
  d[( h - 1 ) * ds + w - 1] = d[( h - 2 ) * ds + w - 2];
}

void vo_equal_coeff( const int16_t *resi, int rs, const int32_t *gx, const int32_t *gy, int ds, int64_t eq[7][7], int w, int h, int b6Param )
{
  const int np = b6Param ? 6 : 4;
  for( int j = 0; j < h; j++ )
  {
    const int cy = ( ( j >> 2 ) << 2 ) + 2;
    for( int k = 0; k < w; k++ )
    {
      const int cx = ( ( k >> 2 ) << 2 ) + 2;
      const int x = gx[j * ds + k], y = gy[j * ds + k];
      int       c[6];
      if( !b6Param ) { c[0] = x; c[1] = cx * x + cy * y; c[2] = y; c[3] = cy * x - cx * y; }
      else { c[0] = x; c[1] = cx * x; c[2] = y; c[3] = cx * y; c[4] = cy * x; c[5] = cy * y; }
      for( int col = 0; col < np; col++ )
      {
        for( int row = 0; row < np; row++ ) eq[col + 1][row] += ( int64_t ) c[col] * c[row];
        eq[col + 1][np] += ( ( int64_t ) c[col] * resi[( ptrdiff_t ) j * rs + k] ) << 3;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Integer motion search  EncoderLib/InterSearch.cpp.
 *   clipMv            CommonLib/Mv.cpp:56-74 (no wrap-around, no sub-pictures -- CTC)
 *   xClipMv           InterSearch.cpp:7735-7764 (identical arithmetic under the same conditions)
 *   xSetSearchRange   :3496-3563        xTZSearchHelp :330-419 (subShiftMode != 1 branch)
 *   xTZ2PointSearch   :422-447          xTZ8PointDiamondSearch :504-705
 *   xTZSearch         :3640-3976        xPatternSearch :3566-3608
 * Structure here: every round first *generates* its ordered candidate list, then evaluates it, then
 * replays the accept rule in list order -- the same three steps the HIP kernel runs (evaluation in parallel).
 * The accept rule "sad < best, then sad + cost < best" equals "sad + cost < best" because cost >= 0.
 * ------------------------------------------------------------------------------------------------ */
static void vo_clip_mv( int *hor, int *ver, const vo_me_ctx_t *c )
{
  const int horMax = ( c->picW + 8 - c->puX - 1 ) << 4;
  const int horMin = ( -c->ctuSize - 8 - c->puX + 1 ) << 4;
  const int verMax = ( c->picH + 8 - c->puY - 1 ) << 4;
  const int verMin = ( -c->ctuSize - 8 - c->puY + 1 ) << 4;
  *hor = *hor < horMin ? horMin : ( *hor > horMax ? horMax : *hor );
  *ver = *ver < verMin ? verMin : ( *ver > verMax ? verMax : *ver );
}

static inline int vo_div_pow2( int v, int i ) { return ( v + ( 1 << ( i - 1 ) ) - ( v >= 0 ) ) >> i; }          /* Mv::divideByPowerOf2 Mv.h:128 */
static inline int vo_prec_down( int v, int rs ) { const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; } /* Mv.h:183-197 */

void vo_set_search_range( const vo_me_ctx_t *c, int predHor, int predVer, int range, vo_range_t *sr )
{
  vo_clip_mv( &predHor, &predVer, c );
  int l = predHor - ( range << 4 ), t = predVer - ( range << 4 );
  int r = predHor + ( range << 4 ), b = predVer + ( range << 4 );
  vo_clip_mv( &l, &t, c );
  vo_clip_mv( &r, &b, c );
  sr->left   = vo_div_pow2( l, 4 );
  sr->top    = vo_div_pow2( t, 4 );
  sr->right  = vo_div_pow2( r, 4 );
  sr->bottom = vo_div_pow2( b, 4 );
}

typedef struct
{
  const vo_me_ctx_t *c;
  vo_range_t         sr;
  uint64_t           bestSad;
  int                bestX, bestY;
  unsigned           bestDist, bestRound;
  int                pointNr;
  uint64_t           nEval;
} vo_tz_t;

typedef struct { int x, y, nr, dist; } vo_pt_t;

static uint64_t vo_me_sad( const vo_me_ctx_t *c, int x, int y )
{
  return vo_sad( c->org, c->orgStride, c->ref + ( ptrdiff_t ) y * c->refStride + x, c->refStride, c->w, c->h, c->subShift );
}

static void vo_tz_check( vo_tz_t *s, const vo_pt_t *p )
{
  const uint64_t cost = vo_me_sad( s->c, p->x, p->y ) + vo_mv_cost( &s->c->mv, p->x, p->y, s->c->imvShift );
  s->nEval++;
  if( cost < s->bestSad )
  {
    s->bestSad   = cost;
    s->bestX     = p->x;
    s->bestY     = p->y;
    s->bestDist  = ( unsigned ) p->dist;
    s->bestRound = 0;
    s->pointNr   = p->nr;
  }
}

#define VO_PUSH( X, Y, NR, D ) do { pts[n].x = ( X ); pts[n].y = ( Y ); pts[n].nr = ( NR ); pts[n].dist = ( D ); n++; } while( 0 )

/* ordered candidate list of one diamond round; at most 16 points */
static int vo_diamond_points( const vo_range_t *sr, int sx, int sy, int d, int cornersAtDist1, vo_pt_t *pts )
{
  int       n = 0;
  const int top = sy - d, bot = sy + d, left = sx - d, right = sx + d;
  if( d == 1 )
  {
    if( top >= sr->top )
    {
      if( cornersAtDist1 && left >= sr->left ) VO_PUSH( left, top, 1, d );
      VO_PUSH( sx, top, 2, d );
      if( cornersAtDist1 && right <= sr->right ) VO_PUSH( right, top, 3, d );
    }
    if( left >= sr->left ) VO_PUSH( left, sy, 4, d );
    if( right <= sr->right ) VO_PUSH( right, sy, 5, d );
    if( bot <= sr->bottom )
    {
      if( cornersAtDist1 && left >= sr->left ) VO_PUSH( left, bot, 6, d );
      VO_PUSH( sx, bot, 7, d );
      if( cornersAtDist1 && right <= sr->right ) VO_PUSH( right, bot, 8, d );
    }
  }
  else if( d <= 8 )
  {
    const int h2 = d >> 1, top2 = sy - h2, bot2 = sy + h2, left2 = sx - h2, right2 = sx + h2;
    if( top >= sr->top && left >= sr->left && right <= sr->right && bot <= sr->bottom )
    {
      VO_PUSH( sx, top, 2, d );
      VO_PUSH( left2, top2, 1, h2 );
      VO_PUSH( right2, top2, 3, h2 );
      VO_PUSH( left, sy, 4, d );
      VO_PUSH( right, sy, 5, d );
      VO_PUSH( left2, bot2, 6, h2 );
      VO_PUSH( right2, bot2, 8, h2 );
      VO_PUSH( sx, bot, 7, d );
    }
    else
    {
      if( top >= sr->top ) VO_PUSH( sx, top, 2, d );
      if( top2 >= sr->top )
      {
        if( left2 >= sr->left ) VO_PUSH( left2, top2, 1, h2 );
        if( right2 <= sr->right ) VO_PUSH( right2, top2, 3, h2 );
      }
      if( left >= sr->left ) VO_PUSH( left, sy, 4, d );
      if( right <= sr->right ) VO_PUSH( right, sy, 5, d );
      if( bot2 <= sr->bottom )
      {
        if( left2 >= sr->left ) VO_PUSH( left2, bot2, 6, h2 );
        if( right2 <= sr->right ) VO_PUSH( right2, bot2, 8, h2 );
      }
      if( bot <= sr->bottom ) VO_PUSH( sx, bot, 7, d );
    }
  }
  else
  {
    const int q      = d >> 2;
    const int inside = top >= sr->top && left >= sr->left && right <= sr->right && bot <= sr->bottom;
    if( inside || top >= sr->top ) VO_PUSH( sx, top, 0, d );
    if( inside || left >= sr->left ) VO_PUSH( left, sy, 0, d );
    if( inside || right <= sr->right ) VO_PUSH( right, sy, 0, d );
    if( inside || bot <= sr->bottom ) VO_PUSH( sx, bot, 0, d );
    for( int i = 1; i < 4; i++ )
    {
      const int yt = top + q * i, yb = bot - q * i, xl = sx - q * i, xr = sx + q * i;
      if( inside || yt >= sr->top )
      {
        if( inside || xl >= sr->left ) VO_PUSH( xl, yt, 0, d );
        if( inside || xr <= sr->right ) VO_PUSH( xr, yt, 0, d );
      }
      if( inside || yb <= sr->bottom )
      {
        if( inside || xl >= sr->left ) VO_PUSH( xl, yb, 0, d );
        if( inside || xr <= sr->right ) VO_PUSH( xr, yb, 0, d );
      }
    }
  }
  return n;
}

static void vo_tz_diamond( vo_tz_t *s, int sx, int sy, int d, int cornersAtDist1 )
{
  vo_pt_t   pts[16];
  const int n = vo_diamond_points( &s->sr, sx, sy, d, cornersAtDist1, pts );
  s->bestRound += 1;
  for( int i = 0; i < n; i++ ) vo_tz_check( s, &pts[i] );
}

static void vo_tz_two_point( vo_tz_t *s )
{
  /* untested neighbours of the best point, indexed by the point number 1..8 of the dist-1 round (:426-446) */
  static const int xo[2][9] = { { 0, -1, -1, 0, -1, +1, -1, -1, +1 }, { 0, 0, +1, +1, -1, +1, 0, +1, 0 } };
  static const int yo[2][9] = { { 0, 0, -1, -1, +1, -1, 0, +1, 0 }, { 0, -1, -1, 0, -1, +1, +1, +1, +1 } };
  const int        nr = s->pointNr;
  vo_pt_t          p[2];
  for( int i = 0; i < 2; i++ ) { p[i].x = s->bestX + xo[i][nr]; p[i].y = s->bestY + yo[i][nr]; p[i].nr = 0; p[i].dist = 2; }
  for( int i = 0; i < 2; i++ )
    if( p[i].x >= s->sr.left && p[i].x <= s->sr.right && p[i].y >= s->sr.top && p[i].y <= s->sr.bottom ) vo_tz_check( s, &p[i] );
}

void vo_tz_search( const vo_me_ctx_t *c, const vo_tz_job_t *job, vo_me_result_t *res )
{
  vo_tz_t s;
  memset( &s, 0, sizeof( s ) );
  s.c       = c;
  s.bestSad = UINT64_MAX;

  const int ext = job->extendedSettings, fast = job->fastSettings;
  const int iRaster            = fast ? 8 : 5;
  const int testZeroVector     = !fast;
  const int firstSearchRounds  = 3;
  const int starRounds         = 2;
  const int searchRange        = job->searchRange;

  /* start vector: clip, internal(1/16) -> quarter with Mv::changePrecision, then divideByPowerOf2(2) (:3686-3687) */
  int mx = job->mvHor, my = job->mvVer;
  vo_clip_mv( &mx, &my, c );
  mx = vo_div_pow2( vo_prec_down( mx, 2 ), 2 );
  my = vo_div_pow2( vo_prec_down( my, 2 ), 2 );

  vo_pt_t p = { mx, my, 0, 0 };
  vo_tz_check( &s, &p );
  if( testZeroVector && ( mx != 0 || my != 0 ) && ( s.bestX != 0 || s.bestY != 0 ) )
  {
    vo_pt_t z = { 0, 0, 0, 0 };
    vo_tz_check( &s, &z );
  }
  if( job->hasIntMv2Nx2NPred )
  {
    int ix = job->intMv2Nx2NPredHor << 4, iy = job->intMv2Nx2NPredVer << 4;   /* INT -> INTERNAL */
    vo_clip_mv( &ix, &iy, c );
    ix = vo_div_pow2( vo_prec_down( ix, 2 ), 2 );
    iy = vo_div_pow2( vo_prec_down( iy, 2 ), 2 );
    if( ( mx != ix || my != iy ) && ( ix != s.bestX || iy != s.bestY ) )
    {
      vo_pt_t q = { ix, iy, 0, 0 };
      vo_tz_check( &s, &q );
    }
  }
  /* m_uniMvList candidates, already de-duplicated by the caller as :3730-3746 does; these do not touch
   * bestDistance / bestRound / pointNr (:3754-3761), which are all still 0 here anyway. */
  for( int i = 0; i < job->numExtraStart; i++ )
  {
    int ex = job->extraStart[i][0], ey = job->extraStart[i][1];
    vo_clip_mv( &ex, &ey, c );
    ex = vo_prec_down( ex, 4 );
    ey = vo_prec_down( ey, 4 );
    const uint64_t cost = vo_me_sad( c, ex, ey ) + vo_mv_cost( &c->mv, ex, ey, c->imvShift );
    s.nEval++;
    if( cost < s.bestSad ) { s.bestSad = cost; s.bestX = ex; s.bestY = ey; }
  }

  vo_set_search_range( c, s.bestX << 4, s.bestY << 4, searchRange >> ( fast ? 1 : 0 ), &s.sr );

  int       startX = s.bestX, startY = s.bestY;
  const int bestCandidateZero = ( s.bestX == 0 && s.bestY == 0 );

  for( int d = 1; d <= searchRange; d *= 2 )
  {
    vo_tz_diamond( &s, startX, startY, d, ext );
    if( job->firstSearchStop && s.bestRound >= ( unsigned ) firstSearchRounds ) break;
  }

  if( ext && !bestCandidateZero )   /* bNewZeroNeighbourhoodTest branch (:3861-3873) */
  {
    for( int d = 1; d <= ( searchRange >> 1 ); d *= 2 ) vo_tz_diamond( &s, 0, 0, d, 0 );
  }

  if( s.bestDist == 1 )
  {
    s.bestDist = 0;
    vo_tz_two_point( &s );
  }

  if( ext )   /* bUseAdaptiveRaster (:3883-3903) */
  {
    int        win = iRaster;
    vo_range_t lsr = s.sr;
    if( !( ( int ) s.bestDist >= iRaster ) )
    {
      win++;
      lsr.left /= 2; lsr.right /= 2; lsr.top /= 2; lsr.bottom /= 2;
    }
    s.bestDist = ( unsigned ) win;
    for( int y = lsr.top; y <= lsr.bottom; y += win )
      for( int x = lsr.left; x <= lsr.right; x += win )
      {
        vo_pt_t q = { x, y, 0, win };
        vo_tz_check( &s, &q );
      }
  }
  else if( ( int ) s.bestDist >= iRaster )
  {
    s.bestDist = ( unsigned ) iRaster;
    for( int y = s.sr.top; y <= s.sr.bottom; y += iRaster )
      for( int x = s.sr.left; x <= s.sr.right; x += iRaster )
      {
        vo_pt_t q = { x, y, 0, iRaster };
        vo_tz_check( &s, &q );
      }
  }

  /* star refinement (:3937-3971) */
  while( s.bestDist > 0 )
  {
    startX     = s.bestX;
    startY     = s.bestY;
    s.bestDist = 0;
    s.pointNr  = 0;
    for( int d = 1; d < searchRange + 1; d *= 2 )
    {
      vo_tz_diamond( &s, startX, startY, d, ext );
      if( fast && s.bestRound >= ( unsigned ) starRounds ) break;
    }
    if( s.bestDist == 1 )
    {
      s.bestDist = 0;
      if( s.pointNr != 0 ) vo_tz_two_point( &s );
    }
  }

  res->mvX   = s.bestX;
  res->mvY   = s.bestY;
  res->cost  = s.bestSad;
  res->dist  = s.bestSad - vo_mv_cost( &c->mv, s.bestX, s.bestY, c->imvShift );
  res->nEval = s.nEval;
}

/* xPatternSearch (:3566-3608): exhaustive raster over the range, first strict minimum */
void vo_full_search( const vo_me_ctx_t *c, const vo_range_t *sr, vo_me_result_t *res )
{
  uint64_t best = UINT64_MAX;
  int      bx = 0, by = 0;
  uint64_t n  = 0;
  for( int y = sr->top; y <= sr->bottom; y++ )
    for( int x = sr->left; x <= sr->right; x++ )
    {
      const uint64_t cost = vo_me_sad( c, x, y ) + vo_mv_cost( &c->mv, x, y, c->imvShift );
      n++;
      if( cost < best ) { best = cost; bx = x; by = y; }
    }
  res->mvX   = bx;
  res->mvY   = by;
  res->cost  = best;
  res->dist  = best - vo_mv_cost( &c->mv, bx, by, c->imvShift );
  res->nEval = n;
}

/* ------------------------------------------------------------------------------------------------
 * Fractional motion search  xPatternSearchFracDIF :4284-4339, xExtDIFUpSamplingH :5840-5889,
 * xExtDIFUpSamplingQ :5895-6051, xPatternRefinement :707-761 (tables s_acMvRefineH/Q :60-85).
 * The plane buffers mirror m_filteredBlockTmp[4] / m_filteredBlock[4][4]; strides are W+1 as in the reference.
 * ------------------------------------------------------------------------------------------------ */
#define VO_FB_STRIDE ( 128 + 1 )
#define VO_FB_ROWS ( 128 + 8 + 1 )
typedef struct
{
  int16_t tmp[4][VO_FB_STRIDE * VO_FB_ROWS];
  int16_t blk[4][4][VO_FB_STRIDE * VO_FB_ROWS];
} vo_fb_t;

static const int8_t vo_refine_h[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
static const int8_t vo_refine_q[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };

static void vo_upsample_h( vo_fb_t *fb, const int16_t *pat, int ps, int w, int h, int bd, int altHpel )
{
  const int      is = w + 1, dsd = w + 1;
  const int16_t *src = pat - 4 * ps - 1;
  vo_if_hor( 0, src, ps, fb->tmp[0], is, w + 1, h + 8, 0 << 2, 0, bd, 0, 0, altHpel );
  vo_if_hor( 0, src, ps, fb->tmp[2], is, w + 1, h + 8, 2 << 2, 0, bd, 0, 0, altHpel );
  vo_if_ver( 0, fb->tmp[0] + 4 * is + 1, is, fb->blk[0][0], dsd, w, h, 0 << 2, 0, 1, bd, 0, 0, altHpel );
  vo_if_ver( 0, fb->tmp[0] + 3 * is + 1, is, fb->blk[2][0], dsd, w, h + 1, 2 << 2, 0, 1, bd, 0, 0, altHpel );
  vo_if_ver( 0, fb->tmp[2] + 4 * is, is, fb->blk[0][2], dsd, w + 1, h, 0 << 2, 0, 1, bd, 0, 0, altHpel );
  vo_if_ver( 0, fb->tmp[2] + 3 * is, is, fb->blk[2][2], dsd, w + 1, h + 1, 2 << 2, 0, 1, bd, 0, 0, altHpel );
}

static void vo_upsample_q( vo_fb_t *fb, const int16_t *pat, int ps, int w, int h, int bd, int halfHor, int halfVer )
{
  const int      is = w + 1, dsd = w + 1;
  const int      extH = ( halfVer == 0 ) ? h + 8 : h + 7;
  const int16_t *src;
  int16_t       *ip;

  src = pat - 4 * ps - 1;
  if( halfVer > 0 ) src += ps;
  if( halfHor >= 0 ) src += 1;
  vo_if_hor( 0, src, ps, fb->tmp[1], is, w, extH, 1 << 2, 0, bd, 0, 0, 0 );

  src = pat - 4 * ps - 1;
  if( halfVer > 0 ) src += ps;
  if( halfHor > 0 ) src += 1;
  vo_if_hor( 0, src, ps, fb->tmp[3], is, w, extH, 3 << 2, 0, bd, 0, 0, 0 );

  ip = fb->tmp[1] + 3 * is;                                   /* @1,1 */
  if( halfVer == 0 ) ip += is;
  vo_if_ver( 0, ip, is, fb->blk[1][1], dsd, w, h, 1 << 2, 0, 1, bd, 0, 0, 0 );
  ip = fb->tmp[1] + 3 * is;                                   /* @3,1 */
  vo_if_ver( 0, ip, is, fb->blk[3][1], dsd, w, h, 3 << 2, 0, 1, bd, 0, 0, 0 );

  if( halfVer != 0 )
  {
    vo_if_ver( 0, fb->tmp[1] + 3 * is, is, fb->blk[2][1], dsd, w, h, 2 << 2, 0, 1, bd, 0, 0, 0 );   /* @2,1 */
    vo_if_ver( 0, fb->tmp[3] + 3 * is, is, fb->blk[2][3], dsd, w, h, 2 << 2, 0, 1, bd, 0, 0, 0 );   /* @2,3 */
  }
  else
  {
    vo_if_ver( 0, fb->tmp[1] + 4 * is, is, fb->blk[0][1], dsd, w, h, 0 << 2, 0, 1, bd, 0, 0, 0 );   /* @0,1 */
    vo_if_ver( 0, fb->tmp[3] + 4 * is, is, fb->blk[0][3], dsd, w, h, 0 << 2, 0, 1, bd, 0, 0, 0 );   /* @0,3 */
  }

  if( halfHor != 0 )
  {
    ip = fb->tmp[2] + 3 * is;                                 /* @1,2 */
    if( halfHor > 0 ) ip += 1;
    if( halfVer >= 0 ) ip += is;
    vo_if_ver( 0, ip, is, fb->blk[1][2], dsd, w, h, 1 << 2, 0, 1, bd, 0, 0, 0 );
    ip = fb->tmp[2] + 3 * is;                                 /* @3,2 */
    if( halfHor > 0 ) ip += 1;
    if( halfVer > 0 ) ip += is;
    vo_if_ver( 0, ip, is, fb->blk[3][2], dsd, w, h, 3 << 2, 0, 1, bd, 0, 0, 0 );
  }
  else
  {
    ip = fb->tmp[0] + 3 * is + 1;                             /* @1,0 */
    if( halfVer >= 0 ) ip += is;
    vo_if_ver( 0, ip, is, fb->blk[1][0], dsd, w, h, 1 << 2, 0, 1, bd, 0, 0, 0 );
    ip = fb->tmp[0] + 3 * is + 1;                             /* @3,0 */
    if( halfVer > 0 ) ip += is;
    vo_if_ver( 0, ip, is, fb->blk[3][0], dsd, w, h, 3 << 2, 0, 1, bd, 0, 0, 0 );
  }

  ip = fb->tmp[3] + 3 * is;                                   /* @1,3 */
  if( halfVer == 0 ) ip += is;
  vo_if_ver( 0, ip, is, fb->blk[1][3], dsd, w, h, 1 << 2, 0, 1, bd, 0, 0, 0 );
  vo_if_ver( 0, fb->tmp[3] + 3 * is, is, fb->blk[3][3], dsd, w, h, 3 << 2, 0, 1, bd, 0, 0, 0 );     /* @3,3 */
}

static uint64_t vo_pattern_refinement( const vo_me_ctx_t *c, const vo_mvcost_t *mc, vo_fb_t *fb, int baseHor, int baseVer, int frac, int *mvHor,
                                       int *mvVer, int useHad, uint64_t cand[9] )
{
  const int     rs = c->w + 1;
  const int8_t( *tab )[2] = frac == 2 ? vo_refine_h : vo_refine_q;
  uint64_t best = UINT64_MAX;
  int      bi   = 0;
  for( int i = 0; i < 9; i++ )
  {
    const int      hv = ( tab[i][0] + baseHor ) * frac, vv = ( tab[i][1] + baseVer ) * frac;
    const int16_t *p  = fb->blk[vv & 3][hv & 3];
    if( hv == 2 && ( vv & 1 ) == 0 ) p += 1;
    if( ( hv & 1 ) == 0 && vv == 2 ) p += rs;
    const int tx = tab[i][0] + *mvHor, ty = tab[i][1] + *mvVer;
    uint64_t  d  = useHad ? vo_satd( c->org, c->orgStride, p, rs, c->w, c->h ) : vo_sad( c->org, c->orgStride, p, rs, c->w, c->h, 0 );
    d += vo_mv_cost( mc, tx, ty, 0 );
    if( cand ) cand[i] = d;
    if( d < best ) { best = d; bi = i; }
  }
  *mvHor = tab[bi][0];
  *mvVer = tab[bi][1];
  return best;
}

/* xPatternSearchFracDIF for cu.imv == 0 (c->imvShift 0: half + quarter) and IMV_HPEL (c->imvShift 1: half only, :4322);
 * intX/intY = integer MV from the integer search. */
void vo_frac_search( const vo_me_ctx_t *c, int intX, int intY, int useHad, int useAltHpelIf, vo_frac_result_t *res )
{
  vo_fb_t       *fb  = ( vo_fb_t * ) malloc( sizeof( vo_fb_t ) );
  const int16_t *pat = c->ref + ( ptrdiff_t ) intY * c->refStride + intX;
  vo_mvcost_t    mc  = c->mv;

  mc.costScale = 1;
  vo_upsample_h( fb, pat, c->refStride, c->w, c->h, c->bitDepth, useAltHpelIf );
  int hx = intX << 1, hy = intY << 1;
  res->costHalf = vo_pattern_refinement( c, &mc, fb, 0, 0, 2, &hx, &hy, useHad, res->candHalf );
  res->halfX    = hx;
  res->halfY    = hy;
  if( c->imvShift != 0 )
  {
    res->cost  = res->costHalf;
    res->qterX = res->qterY = 0;
    memset( res->candQuarter, 0, sizeof( res->candQuarter ) );
    free( fb );
    return;
  }

  mc.costScale = 0;
  vo_upsample_q( fb, pat, c->refStride, c->w, c->h, c->bitDepth, hx, hy );
  int qx = ( ( intX << 1 ) + hx ) << 1, qy = ( ( intY << 1 ) + hy ) << 1;
  res->cost = vo_pattern_refinement( c, &mc, fb, hx << 1, hy << 1, 1, &qx, &qy, useHad, res->candQuarter );
  res->qterX = qx;
  res->qterY = qy;
  free( fb );
}

/* Direct 2-D separable luma interpolation at quarter-sample offset (qx,qy) from `pat`: what each
 * fractional candidate's block is, independent of the plane bookkeeping above (used to cross-check the
 * fused HIP interp+SATD kernel). */
void vo_interp_qpel( const int16_t *pat, int ps, int w, int h, int bitDepth, int qx, int qy, int16_t *dst, int ds )
{
  int16_t        tmp[( 128 + 8 ) * 128];
  const int      fx = qx & 3, fy = qy & 3, cmax = ( 1 << bitDepth ) - 1;
  const int16_t *src = pat + ( ptrdiff_t )( ( qy >> 2 ) - 3 ) * ps + ( qx >> 2 );
  if( fx == 0 ) vo_if_copy( 1, 0, src, ps, tmp, w, w, h + 7, bitDepth, 0, cmax, 0 );
  else vo_if_filter( 0, 8, 1, 0, src, ps, tmp, w, w, h + 7, vo_luma_filter[fx << 2], bitDepth, 0, cmax, 0 );
  if( fy == 0 ) vo_if_copy( 0, 1, tmp + 3 * w, w, dst, ds, w, h, bitDepth, 0, cmax, 0 );
  else vo_if_filter( 1, 8, 0, 1, tmp + 3 * w, w, dst, ds, w, h, vo_luma_filter[fy << 2], bitDepth, 0, cmax, 0 );
}

/* ------------------------------------------------------------------------------------------------
 * Luma motion compensation of one block, InterPrediction::xPredInterBlk (CommonLib/InterPrediction.cpp:660-815)
 * without BDOF / DMVR / RPR / wrap-around: mv in internal 1/16 precision, bi = 0 -> rounded + clipped samples
 * (rndRes), bi = 1 -> 14-bit intermediates for addAvg.  `ref` points at the block position with MV (0,0).
 * (Composition of the pinned filterHor / filterVer; the composition itself is restated from the lines cited.)
 * ------------------------------------------------------------------------------------------------ */
void vo_mc_luma( const int16_t *ref, int refStride, int w, int h, int mvHor, int mvVer, int bi, int bitDepth, int useAltHpelIf, int16_t *dst,
                 int dstStride )
{
  const int      xFrac = mvHor & 15, yFrac = mvVer & 15, rndRes = !bi;
  const int16_t *src   = ref + ( ptrdiff_t )( mvVer >> 4 ) * refStride + ( mvHor >> 4 );
  if( yFrac == 0 )
  {
    vo_if_hor( 0, src, refStride, dst, dstStride, w, h, xFrac, rndRes, bitDepth, 0, 0, useAltHpelIf );
  }
  else if( xFrac == 0 )
  {
    vo_if_ver( 0, src, refStride, dst, dstStride, w, h, yFrac, 1, rndRes, bitDepth, 0, 0, useAltHpelIf );
  }
  else
  {
    int16_t tmp[128 * ( 128 + 7 )];
    vo_if_hor( 0, src - 3 * refStride, refStride, tmp, w, w, h + 7, xFrac, 0, bitDepth, 0, 0, useAltHpelIf );
    vo_if_ver( 0, tmp + 3 * w, w, dst, dstStride, w, h, yFrac, 0, rndRes, bitDepth, 0, 0, useAltHpelIf );
  }
}

/* xPredInterBlk for luma (comp 0) or a 4:2:0 chroma plane (comp 1 / 2): the vector stays in luma 1/16 units, so the chroma phase
 * has 5 bits (1/32 sample) and the 4-tap chroma filter applies (InterPrediction.cpp:675-676, 693-694, 766-785).
 * `ref` points at the block position with MV (0,0) in THAT plane; w, h in samples of that plane. */
void vo_mc_block( int comp, const int16_t *ref, int refStride, int w, int h, int mvHor, int mvVer, int bi, int bitDepth, int useAltHpelIf,
                  int16_t *dst, int dstStride )
{
  if( comp == 0 )
  {
    vo_mc_luma( ref, refStride, w, h, mvHor, mvVer, bi, bitDepth, useAltHpelIf, dst, dstStride );
    return;
  }
  const int      xFrac = mvHor & 31, yFrac = mvVer & 31, rndRes = !bi;
  const int16_t *src   = ref + ( ptrdiff_t )( mvVer >> 5 ) * refStride + ( mvHor >> 5 );
  if( yFrac == 0 ) vo_if_hor( comp, src, refStride, dst, dstStride, w, h, xFrac, rndRes, bitDepth, 0, 0, useAltHpelIf );
  else if( xFrac == 0 ) vo_if_ver( comp, src, refStride, dst, dstStride, w, h, yFrac, 1, rndRes, bitDepth, 0, 0, useAltHpelIf );
  else
  {
    int16_t tmp[64 * ( 64 + 3 )];
    vo_if_hor( comp, src - refStride, refStride, tmp, w, w, h + 3, xFrac, 0, bitDepth, 0, 0, useAltHpelIf );
    vo_if_ver( comp, tmp + w, w, dst, dstStride, w, h, yFrac, 0, rndRes, bitDepth, 0, 0, useAltHpelIf );
  }
}

/* ------------------------------------------------------------------------------------------------
 * BDOF (bi-directional optical flow) of one bi-predicted luma PU -- InterPrediction::xPredInterBi with bioApplied
 * (CommonLib/InterPrediction.cpp:527-660): xSubPuBio (:352-443) cuts the PU into regions of at most 16 x 16, every region is predicted
 * from both lists by xPredInterBlk(..., bioApplied) (:733-810: the 14-bit prediction plus a one-sample ring taken from the NEAREST
 * INTEGER reference samples), and xWeightedAverage -> applyBiOptFlow (:1233-1334) refines each 4 x 4 unit with
 * gradFilterCore / calcBIOSumsCore / addBIOAvgCore (CommonLib/Buffer.cpp:88-200).  ref0 / ref1 point at the PU position with MV (0,0).
 * ------------------------------------------------------------------------------------------------ */
static void vo_bdof_region( const int16_t *ref0, int stride0, const int16_t *ref1, int stride1, int w, int h, const int mv[2][2], int bitDepth,
                            int16_t *dst, int dstStride )
{
  enum { S = 16 + 4, G = 16 + 2 };
  int16_t   pred[2][S * S], gx[2][G * G], gy[2][G * G];
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
  for( int l = 0; l < 2; l++ )
  {
    const int16_t *ref = l ? ref1 : ref0;
    const int      rs = l ? stride1 : stride0;
    int16_t       *P = pred[l];
    /* interior at (2,2), stride S: xPredInterBlk redirects its output there (:733-741) */
    vo_mc_block( 0, ref, rs, w, h, mv[l][0], mv[l][1], 1, bitDepth, 0, P + 2 * S + 2, S );
    /* ring (:768-803): integer sample nearest to the fractional position, as 14-bit intermediate */
    const int      xo = ( mv[l][0] & 15 ) < 8 ? 1 : 0, yo = ( mv[l][1] & 15 ) < 8 ? 1 : 0;
    const int16_t *src = ref + ( ptrdiff_t )( mv[l][1] >> 4 ) * rs + ( mv[l][0] >> 4 );
    for( int r = -1; r <= h; r++ )
      for( int c = -1; c <= w; c++ )
        if( r == -1 || r == h || c == -1 || c == w )
          P[( r + 2 ) * S + c + 2] = ( int16_t )( ( src[( ptrdiff_t )( r + 1 - yo ) * rs + c + 1 - xo ] << headRoom ) - 8192 );
    /* gradFilterCore<true> (Buffer.cpp:130-170) on the (w+2) x (h+2) window whose origin is the ring corner */
    int16_t *X = gx[l], *Y = gy[l];
    for( int y = 0; y < h; y++ )
      for( int x = 0; x < w; x++ )
      {
        const int16_t *q = P + ( y + 2 ) * S + x + 2;
        Y[( y + 1 ) * G + x + 1] = ( int16_t )( ( q[S] >> 6 ) - ( q[-S] >> 6 ) );
        X[( y + 1 ) * G + x + 1] = ( int16_t )( ( q[1] >> 6 ) - ( q[-1] >> 6 ) );
      }
    for( int y = 1; y <= h; y++ )
    {
      X[y * G] = X[y * G + 1]; X[y * G + w + 1] = X[y * G + w];
      Y[y * G] = Y[y * G + 1]; Y[y * G + w + 1] = Y[y * G + w];
    }
    for( int x = 0; x < w + 2; x++ )
    {
      X[x] = X[G + x]; X[( h + 1 ) * G + x] = X[h * G + x];
      Y[x] = Y[G + x]; Y[( h + 1 ) * G + x] = Y[h * G + x];
    }
    /* applyBiOptFlow then replaces the ring of the prediction by its replicated border (:1266-1276) */
    for( int y = 0; y < h; y++ )
    {
      P[( y + 2 ) * S + 1]     = P[( y + 2 ) * S + 2];
      P[( y + 2 ) * S + w + 2] = P[( y + 2 ) * S + w + 1];
    }
    for( int x = 1; x < w + 3; x++ )
    {
      P[S + x]             = P[2 * S + x];
      P[( h + 2 ) * S + x] = P[( h + 1 ) * S + x];
    }
  }
  const int shiftNum = headRoom + 1, offset = ( 1 << ( shiftNum - 1 ) ) + 2 * 8192, limit = 15, cmax = ( 1 << bitDepth ) - 1;
  for( int yu = 0; yu < ( h >> 2 ); yu++ )
    for( int xu = 0; xu < ( w >> 2 ); xu++ )
    {
      int sumAbsGX = 0, sumAbsGY = 0, sumDIX = 0, sumDIY = 0, sumSignGYGX = 0;
      for( int y = 0; y < 6; y++ )   /* calcBIOSumsCore (Buffer.cpp:173-200): 6 x 6 window around the unit */
        for( int x = 0; x < 6; x++ )
        {
          const int gi = ( yu * 4 + y ) * G + xu * 4 + x, pi = ( yu * 4 + y + 1 ) * S + xu * 4 + x + 1;
          const int tGX = ( gx[0][gi] + gx[1][gi] ) >> 1, tGY = ( gy[0][gi] + gy[1][gi] ) >> 1;
          const int tDI = ( pred[1][pi] >> 4 ) - ( pred[0][pi] >> 4 );
          sumAbsGX += tGX < 0 ? -tGX : tGX;
          sumAbsGY += tGY < 0 ? -tGY : tGY;
          sumDIX += tGX < 0 ? -tDI : tGX == 0 ? 0 : tDI;
          sumDIY += tGY < 0 ? -tDI : tGY == 0 ? 0 : tDI;
          sumSignGYGX += tGY < 0 ? -tGX : tGY == 0 ? 0 : tGX;
        }
      int tmpx = sumAbsGX == 0 ? 0 : ( sumDIX * 4 ) >> vo_floor_log2( ( unsigned ) sumAbsGX );   /* rightShiftMSB (:1606-1609) */
      tmpx = tmpx < -limit ? -limit : tmpx > limit ? limit : tmpx;
      const int mains = sumSignGYGX >> 12, secs = sumSignGYGX & 4095;
      int       tmpData = tmpx * mains;
      tmpData = ( tmpData * 4096 + tmpx * secs ) >> 1;
      int tmpy = sumAbsGY == 0 ? 0 : ( sumDIY * 4 - tmpData ) >> vo_floor_log2( ( unsigned ) sumAbsGY );
      tmpy = tmpy < -limit ? -limit : tmpy > limit ? limit : tmpy;
      for( int y = 0; y < 4; y++ )   /* addBIOAvgCore (Buffer.cpp:88-127) */
        for( int x = 0; x < 4; x++ )
        {
          const int gi = ( yu * 4 + y + 1 ) * G + xu * 4 + x + 1, pi = ( yu * 4 + y + 2 ) * S + xu * 4 + x + 2;
          const int b  = tmpx * ( gx[0][gi] - gx[1][gi] ) + tmpy * ( gy[0][gi] - gy[1][gi] );
          const int v  = ( int16_t )( ( pred[0][pi] + pred[1][pi] + b + offset ) >> shiftNum );
          dst[( ptrdiff_t )( yu * 4 + y ) * dstStride + xu * 4 + x] = ( int16_t )( v < 0 ? 0 : v > cmax ? cmax : v );
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * DMVR of one bi-predicted LUMA PU -- InterPrediction::xProcessDMVR (CommonLib/InterPrediction.cpp:1997-2195) for the luma plane:
 * per sub-PU of at most 16 x 16: xPrefetch (:1666-1708: the (dx+7) x (dy+7) integer window of each list into a private buffer), xinitMC
 * (:1941-1995: bilinear (dx+4) x (dy+4) predictions, 10-bit), xDMVRCost (:1919-1927: SAD of every other row), the 25-point integer
 * refinement xBIPMVRefine (:1819-1843) around the mirrored displacement, the parametric error surface (:1733-1817, 1929-1947), xPad (:1709-1731,
 * paddingCore Buffer.cpp:340-364: the window replicated by 2 samples, only when the sub-PU moved), xFinalPaddedMCForDMVR (:1845-1917: 8-tap
 * prediction out of the padded window) and xWeightedAverage (:1354-1435) with BDOF unless the matching cost was below 2*dx*dy.
 * plane0 / plane1: origins of the two reference luma planes; the picture geometry is needed for clipMv (Mv.cpp:56-74).
 * mvdOut (optional): pu.mvdL0SubPu of every sub-PU, [num][2].
 * ------------------------------------------------------------------------------------------------ */
static void vo_clip_mv_pic( int *hor, int *ver, int picW, int picH, int ctuSize, int x, int y )
{
  const int horMax = ( picW + 8 - x - 1 ) << 4, horMin = ( -ctuSize - 8 - x + 1 ) << 4;
  const int verMax = ( picH + 8 - y - 1 ) << 4, verMin = ( -ctuSize - 8 - y + 1 ) << 4;
  *hor = *hor < horMin ? horMin : *hor > horMax ? horMax : *hor;
  *ver = *ver < verMin ? verMin : *ver > verMax ? verMax : *ver;
}

static int vo_div_for_maxq7( int64_t N, int64_t D )   /* :1733-1767 */
{
  int sign = 0, q = 0;
  if( N < 0 ) { sign = 1; N = -N; }
  D = D * 8;
  if( N >= D ) { N -= D; q++; }
  q = q * 2;
  D = D >> 1;
  if( N >= D ) { N -= D; q++; }
  q = q * 2;
  if( N >= ( D >> 1 ) ) q++;
  return sign ? -q : q;
}

void vo_dmvr_pu( const int16_t *plane0, const int16_t *plane1, int stride, int picW, int picH, int ctuSize, int puX, int puY, int w, int h, int mv0Hor,
                 int mv0Ver, int mv1Hor, int mv1Ver, int bitDepth, int bioApplied, int16_t *dst, int dstStride, int32_t *mvdOut )
{
  enum { MAXS = 16, PS = MAXS + 12, BS = MAXS + 4 };
  const int mergeMv[2][2] = { { mv0Hor, mv0Ver }, { mv1Hor, mv1Ver } };
  const int dx = w < 16 ? w : 16, dy = h < 16 ? h : 16;
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2, cmax = ( 1 << bitDepth ) - 1;
  int num = 0;
  for( int sy = 0; sy < h; sy += dy )
    for( int sx = 0; sx < w; sx += dx, num++ )
    {
      const int x = puX + sx, y = puY + sy;
      int16_t   pad[2][PS * PS], bil[2][BS * BS], tmp[BS * ( BS + 1 )];
      for( int l = 0; l < 2; l++ )
      {
        const int16_t *plane = l ? plane1 : plane0;
        /* xPrefetch, luma: the vector moved by the filter reach, clipped, whole samples */
        int ph = mergeMv[l][0] - ( 3 << 4 ), pv = mergeMv[l][1] - ( 3 << 4 );
        vo_clip_mv_pic( &ph, &pv, picW, picH, ctuSize, x, y );
        const int16_t *src = plane + ( ptrdiff_t )( y + ( pv >> 4 ) ) * stride + x + ( ph >> 4 );
        for( int r = 0; r < dy + 7; r++ ) memcpy( pad[l] + ( r + 2 ) * PS + 2, src + ( ptrdiff_t ) r * stride, sizeof( int16_t ) * ( dx + 7 ) );
        /* xinitMC: bilinear prediction of the (dx+4) x (dy+4) block that starts 2 samples up-left of the merge position */
        int mh = mergeMv[l][0], mvv = mergeMv[l][1];
        vo_clip_mv_pic( &mh, &mvv, picW, picH, ctuSize, x, y );
        const int      xFrac = mh & 15, yFrac = mvv & 15, bw = dx + 4, bh = dy + 4;
        const int16_t *b0 = pad[l] + 3 * ( PS + 1 );
        if( yFrac == 0 ) vo_if_hor( 0, b0, PS, bil[l], BS, bw, bh, xFrac, 0, bitDepth, 1, 1, 0 );
        else if( xFrac == 0 ) vo_if_ver( 0, b0, PS, bil[l], BS, bw, bh, yFrac, 1, 0, bitDepth, 1, 1, 0 );
        else
        {
          vo_if_hor( 0, b0, PS, tmp, bw, bw, bh + 1, xFrac, 0, bitDepth, 1, 1, 0 );
          vo_if_ver( 0, tmp, bw, bil[l], BS, bw, bh, yFrac, 0, 0, bitDepth, 1, 1, 0 );
        }
      }
      /* matching cost: SAD over every other row (setDistParam(..., subShiftMode = 1) then >> 1, RdCost.cpp:368-408) */
#define VO_DMVR_COST( ox, oy, out )                                                                                         \
  {                                                                                                                         \
    const int16_t *a = bil[0] + ( 2 + ( oy ) ) * BS + 2 + ( ox ), *b = bil[1] + ( 2 - ( oy ) ) * BS + 2 - ( ox );             \
    uint64_t       acc = 0;                                                                                                 \
    for( int r = 0; r < dy; r += 2 )                                                                                        \
      for( int c = 0; c < dx; c++ ) acc += ( uint64_t ) vo_abs( ( int ) a[r * BS + c] - ( int ) b[r * BS + c] );             \
    ( out ) = ( ( acc << 1 ) >> 1 );                                                                                        \
  }
      uint64_t sads[25], minCost;
      int      notZero = 1, total[2] = { 0, 0 }, best = 12;
      for( int i = 0; i < 25; i++ ) sads[i] = UINT64_MAX;
      VO_DMVR_COST( 0, 0, minCost );
      minCost -= minCost >> 2;
      if( minCost < ( uint64_t )( dx * dy ) ) notZero = 0;
      else
      {
        sads[12] = minCost;
        for( int i = 0; i < 25; i++ )   /* xBIPMVRefine: raster order over [-2,2]^2, strict '<' */
        {
          if( sads[i] == UINT64_MAX ) VO_DMVR_COST( i % 5 - 2, i / 5 - 2, sads[i] );
          if( sads[i] < minCost ) { minCost = sads[i]; best = i; }
        }
        total[0] = best % 5 - 2; total[1] = best / 5 - 2;
      }
#undef VO_DMVR_COST
      const int bio = minCost < ( uint64_t )( 2 * dx * dy ) ? 0 : bioApplied;
      total[0] *= 16; total[1] *= 16;
      if( notZero && total[0] != 32 && total[0] != -32 && total[1] != 32 && total[1] != -32 )   /* xDMVRSubPixelErrorSurface */
      {
        const uint64_t sb[5] = { sads[best], sads[best - 1], sads[best - 5], sads[best + 1], sads[best + 5] };
        for( int d = 0; d < 2; d++ )
        {
          const uint64_t s1 = sb[1 + d], s3 = sb[3 + d];
          const int64_t  numer = ( int64_t )( ( s1 - s3 ) << 4 ), denom = ( int64_t )( s1 + s3 - ( sb[0] << 1 ) );
          if( denom != 0 )
          {
            if( s1 != sb[0] && s3 != sb[0] ) total[d] += vo_div_for_maxq7( numer, denom );
            else total[d] += s1 == sb[0] ? -8 : 8;
          }
        }
      }
      total[0] = ( int16_t ) total[0]; total[1] = ( int16_t ) total[1];
      if( mvdOut ) { mvdOut[2 * num] = total[0]; mvdOut[2 * num + 1] = total[1]; }
      const int moved = total[0] != 0 || total[1] != 0;
      if( moved )   /* xPad: the prefetched window replicated by DMVR_NUM_ITERATION samples on every side */
        for( int l = 0; l < 2; l++ )
        {
          int16_t *p = pad[l] + 2 * ( PS + 1 );
          const int pw = dx + 7, ph2 = dy + 7;
          for( int r = 0; r < ph2; r++ )
            for( int j = 1; j <= 2; j++ ) { p[r * PS - j] = p[r * PS]; p[r * PS + pw - 1 + j] = p[r * PS + pw - 1]; }
          for( int i = 1; i <= 2; i++ )
          {
            memcpy( p - 2 - i * PS, p - 2, sizeof( int16_t ) * ( pw + 4 ) );
            memcpy( p - 2 + ( ph2 - 1 + i ) * PS, p - 2 + ( ph2 - 1 ) * PS, sizeof( int16_t ) * ( pw + 4 ) );
          }
        }
      /* xFinalPaddedMCForDMVR + xWeightedAverage */
      const int16_t *fsrc[2];
      int            frac[2][2];
      for( int l = 0; l < 2; l++ )
      {
        int rh = mergeMv[l][0] + ( l ? -total[0] : total[0] ), rv = mergeMv[l][1] + ( l ? -total[1] : total[1] );
        rh = rh < -( 1 << 17 ) ? -( 1 << 17 ) : rh > ( 1 << 17 ) - 1 ? ( 1 << 17 ) - 1 : rh;   /* clipToStorageBitDepth */
        rv = rv < -( 1 << 17 ) ? -( 1 << 17 ) : rv > ( 1 << 17 ) - 1 ? ( 1 << 17 ) - 1 : rv;
        int ch = rh, cv = rv;
        vo_clip_mv_pic( &ch, &cv, picW, picH, ctuSize, x, y );
        frac[l][0] = ch & 15; frac[l][1] = cv & 15;
        const int dX = ( rh >> 4 ) - ( mergeMv[l][0] >> 4 ), dY = ( rv >> 4 ) - ( mergeMv[l][1] >> 4 );
        fsrc[l] = pad[l] + 5 * ( PS + 1 ) + dY * PS + dX;
      }
      int16_t *out = dst + ( ptrdiff_t ) sy * dstStride + sx;
      if( bio ) vo_bdof_region( fsrc[0], PS, fsrc[1], PS, dx, dy, ( const int( * )[2] ) frac, bitDepth, out, dstStride );
      else
      {
        int16_t p0[MAXS * MAXS], p1[MAXS * MAXS];
        vo_mc_block( 0, fsrc[0], PS, dx, dy, frac[0][0], frac[0][1], 1, bitDepth, 0, p0, dx );
        vo_mc_block( 0, fsrc[1], PS, dx, dy, frac[1][0], frac[1][1], 1, bitDepth, 0, p1, dx );
        const int shift = headRoom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * 8192;
        for( int r = 0; r < dy; r++ )
          for( int c = 0; c < dx; c++ )
          {
            const int v = ( p0[r * dx + c] + p1[r * dx + c] + offset ) >> shift;
            out[( ptrdiff_t ) r * dstStride + c] = ( int16_t )( v < 0 ? 0 : v > cmax ? cmax : v );
          }
      }
    }
}

/* The 4:2:0 chroma planes of the same PU (xProcessDMVR with chroma enabled): a sub-PU that did not move is predicted straight from the reference
 * pictures (xFinalPaddedMCForDMVR passes no window, :1879-1905), a moved one out of its own (w/2+3) x (h/2+3) window, prefetched with the MERGE vector
 * (xPrefetch forLuma = 0, :1666-1708), replicated by ONE sample (xPad: padsize = 2 >> scaleY, :1709-1731) and addressed with the whole-sample part of
 * the refined vector; then the plain addAvg.  planeC0 / planeC1: origins of that chroma plane in the two reference pictures; puX, puY, w, h and the
 * vectors stay in luma units; mvd: pu.mvdL0SubPu as vo_dmvr_pu returned it. */
void vo_dmvr_chroma( const int16_t *planeC0, const int16_t *planeC1, int strideC, int picW, int picH, int ctuSize, int puX, int puY, int w, int h,
                     int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver, const int32_t *mvd, int bitDepth, int16_t *dst, int dstStride )
{
  enum { MAXC = 8, PSC = MAXC + 12 };
  const int mergeMv[2][2] = { { mv0Hor, mv0Ver }, { mv1Hor, mv1Ver } };
  const int dx = w < 16 ? w : 16, dy = h < 16 ? h : 16, cw = dx >> 1, chh = dy >> 1;
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2, cmax = ( 1 << bitDepth ) - 1;
  int num = 0;
  for( int sy = 0; sy < h; sy += dy )
    for( int sx = 0; sx < w; sx += dx, num++ )
    {
      const int x = puX + sx, y = puY + sy, xc = x >> 1, yc = y >> 1;
      const int moved = mvd[2 * num] != 0 || mvd[2 * num + 1] != 0;
      int16_t   p[2][MAXC * MAXC];
      for( int l = 0; l < 2; l++ )
      {
        const int16_t *plane = l ? planeC1 : planeC0;
        int rh = mergeMv[l][0] + ( l ? -mvd[2 * num] : mvd[2 * num] ), rv = mergeMv[l][1] + ( l ? -mvd[2 * num + 1] : mvd[2 * num + 1] );
        rh = rh < -( 1 << 17 ) ? -( 1 << 17 ) : rh > ( 1 << 17 ) - 1 ? ( 1 << 17 ) - 1 : rh;
        rv = rv < -( 1 << 17 ) ? -( 1 << 17 ) : rv > ( 1 << 17 ) - 1 ? ( 1 << 17 ) - 1 : rv;
        int ch = rh, cv = rv;
        vo_clip_mv_pic( &ch, &cv, picW, picH, ctuSize, x, y );
        if( !moved )
        {
          vo_mc_block( 1, plane + ( ptrdiff_t ) yc * strideC + xc, strideC, cw, chh, ch, cv, 1, bitDepth, 0, p[l], cw );
          continue;
        }
        int16_t pad[PSC * PSC];
        int     ph = mergeMv[l][0] - ( 1 << 5 ), pv = mergeMv[l][1] - ( 1 << 5 );
        vo_clip_mv_pic( &ph, &pv, picW, picH, ctuSize, x, y );
        const int16_t *src = plane + ( ptrdiff_t )( yc + ( pv >> 5 ) ) * strideC + xc + ( ph >> 5 );
        const int      pw = cw + 3, phh = chh + 3;
        for( int r = 0; r < phh; r++ ) memcpy( pad + ( r + 2 ) * PSC + 2, src + ( ptrdiff_t ) r * strideC, sizeof( int16_t ) * pw );
        int16_t *q = pad + 2 * ( PSC + 1 );
        for( int r = 0; r < phh; r++ ) { q[r * PSC - 1] = q[r * PSC]; q[r * PSC + pw] = q[r * PSC + pw - 1]; }
        memcpy( q - 1 - PSC, q - 1, sizeof( int16_t ) * ( pw + 2 ) );
        memcpy( q - 1 + phh * PSC, q - 1 + ( phh - 1 ) * PSC, sizeof( int16_t ) * ( pw + 2 ) );
        const int dX = ( rh >> 5 ) - ( mergeMv[l][0] >> 5 ), dY = ( rv >> 5 ) - ( mergeMv[l][1] >> 5 );
        vo_mc_block( 1, pad + 3 * ( PSC + 1 ) + dY * PSC + dX, PSC, cw, chh, ch & 31, cv & 31, 1, bitDepth, 0, p[l], cw );
      }
      const int shift = headRoom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * 8192;
      int16_t  *out = dst + ( ptrdiff_t )( sy >> 1 ) * dstStride + ( sx >> 1 );
      for( int r = 0; r < chh; r++ )
        for( int c = 0; c < cw; c++ )
        {
          const int v = ( p[0][r * cw + c] + p[1][r * cw + c] + offset ) >> shift;
          out[( ptrdiff_t ) r * dstStride + c] = ( int16_t )( v < 0 ? 0 : v > cmax ? cmax : v );
        }
    }
}

void vo_bdof_pu( const int16_t *ref0, int stride0, const int16_t *ref1, int stride1, int w, int h, int mv0Hor, int mv0Ver, int mv1Hor, int mv1Ver,
                 int bitDepth, int16_t *dst, int dstStride )
{
  const int mv[2][2] = { { mv0Hor, mv0Ver }, { mv1Hor, mv1Ver } };
  const int dx = w < 16 ? w : 16, dy = h < 16 ? h : 16;   /* MAX_BDOF_APPLICATION_REGION, xSubPuBio :414-417 */
  for( int y = 0; y < h; y += dy )
    for( int x = 0; x < w; x += dx )
      vo_bdof_region( ref0 + ( ptrdiff_t ) y * stride0 + x, stride0, ref1 + ( ptrdiff_t ) y * stride1 + x, stride1, dx, dy, mv, bitDepth,
                      dst + ( ptrdiff_t ) y * dstStride + x, dstStride );
}

/* ------------------------------------------------------------------------------------------------
 * InterSearch::xMotionEstimation (EncoderLib/InterSearch.cpp:3299-3494) for one (PU, list, refIdx), without BCW,
 * weighted prediction, MCTS, composite references and the block-MV cache (none is on in the CTC), with
 * xPatternSearchIntRefine (:4172-4282) for the integer / 4-sample AMVR modes.  Composition of the pinned stages
 * above; the composition itself (start candidates of the bi-pred search :3384-3425, final rate re-weighting
 * :3478-3484, AMVR refinement) is restated from the lines cited and pinned through ref_motion_estimation().
 * ------------------------------------------------------------------------------------------------ */
static void vo_to_amvr( int *v, int imv )   /* Mv::changeTransPrecInternal2Amvr: INTERNAL(6) -> {QUARTER 4, INT 2, 4PEL 0, HALF 3} */
{
  static const int rs[4] = { 2, 4, 6, 3 };
  *v = vo_prec_down( *v, rs[imv] );
}
static int vo_amvr_shift( int imv ) { static const int rs[4] = { 2, 4, 6, 3 }; return rs[imv]; }

void vo_motion_estimation( const vo_mest_cfg_t *cfg, const vo_mest_job_t *j, vo_mest_result_t *res )
{
  const int      w = j->w, h = j->h;
  int16_t       *tmp = NULL;
  const int16_t *pat = j->org;
  int            ps  = j->orgStride;
  double         fWeight = 1.0;
  if( j->bi )
  {
    tmp = ( int16_t * ) malloc( sizeof( int16_t ) * w * h );
    for( int y = 0; y < h; y++ ) memcpy( tmp + y * w, j->org + ( ptrdiff_t ) y * j->orgStride, sizeof( int16_t ) * w );
    if( j->bcwWeight != 0 && j->bcwWeight != 4 )      /* a CU-level BCW weight (:3320-3328): weighted target, distortion weight |w| / 8 */
    {
      vo_remove_weight_high_freq( tmp, w, j->otherPred, j->otherStride, w, h, j->bcwWeight );
      fWeight = fabs( ( double ) j->bcwWeight / 8.0 );
    }
    else
    {
      vo_remove_high_freq( tmp, w, j->otherPred, j->otherStride, w, h );
      fWeight = 0.5;
    }
    pat = tmp; ps = w;
  }
  vo_me_ctx_t c;
  memset( &c, 0, sizeof( c ) );
  c.org = pat; c.orgStride = ps; c.ref = j->ref; c.refStride = j->refStride; c.w = w; c.h = h;
  c.bitDepth = j->bitDepth; c.imvShift = j->imv == 3 ? 1u : ( unsigned ) j->imv << 1;
  c.picW = j->picW; c.picH = j->picH; c.puX = j->puX; c.puY = j->puY; c.ctuSize = j->ctuSize;
  c.mv.motionLambda = j->motionLambda;
  c.mv.predHor = vo_prec_down( j->mvPredHor, 2 );      /* predQuarter */
  c.mv.predVer = vo_prec_down( j->mvPredVer, 2 );
  c.mv.costScale = 2;

  /* m_uniMvList candidates, newest first, de-duplicated against the earlier ones (:3391-3403 and :3728-3746) */
  int ex[16][2], nex = 0;
  for( int i = 0; i < j->numExtraStart; i++ )
  {
    int k = 0;
    for( ; k < i; k++ ) if( j->extraStart[k][0] == j->extraStart[i][0] && j->extraStart[k][1] == j->extraStart[i][1] ) break;
    if( k < i ) continue;
    ex[nex][0] = j->extraStart[i][0]; ex[nex][1] = j->extraStart[i][1]; nex++;
  }

  vo_me_result_t ir;
  if( j->bi )
  {
    c.subShift = vo_subshift_for_mode( w, h, cfg->fastInterSearchMode13 ? 2 : 0 );
    int bestH = j->mvHor, bestV = j->mvVer;
    int th = bestH, tv = bestV;
    vo_clip_mv( &th, &tv, &c );
    th = vo_prec_down( th, 4 ); tv = vo_prec_down( tv, 4 );
    uint64_t best = vo_me_sad( &c, th, tv ) + vo_mv_cost( &c.mv, th, tv, c.imvShift );
    for( int i = 0; i < nex; i++ )
    {
      th = ex[i][0]; tv = ex[i][1];
      vo_clip_mv( &th, &tv, &c );
      th = vo_prec_down( th, 4 ); tv = vo_prec_down( tv, 4 );
      const uint64_t sad = vo_me_sad( &c, th, tv ) + vo_mv_cost( &c.mv, th, tv, c.imvShift );
      if( sad < best ) { best = sad; bestH = ex[i][0]; bestV = ex[i][1]; }
    }
    vo_range_t sr;
    vo_set_search_range( &c, bestH, bestV, cfg->bipredSearchRange, &sr );
    vo_full_search( &c, &sr, &ir );
  }
  else
  {
    c.subShift = vo_subshift_for_mode( w, h, cfg->fastInterSearchMode13 ? 2 : 0 );
    vo_tz_job_t t;
    memset( &t, 0, sizeof( t ) );
    t.mvHor = j->cachedIntMv ? j->mvHor : j->mvPredHor; t.mvVer = j->cachedIntMv ? j->mvVer : j->mvPredVer; t.searchRange = j->searchRange;   /* bQTBTMV2 (:3434-3441) or rcMv = rcMvPred (:3446) */
    t.extendedSettings = cfg->extendedSettings; t.fastSettings = j->cachedIntMv != 0; t.firstSearchStop = cfg->firstSearchStop;
    t.numExtraStart = nex;
    memcpy( t.extraStart, ex, sizeof( int ) * 2 * nex );
    vo_tz_search( &c, &t, &ir );
  }
  res->intX = ir.mvX; res->intY = ir.mvY; res->intDist = ir.dist;

  unsigned bits = j->bits;
  if( j->imv == 0 || j->imv == 3 )
  {
    vo_frac_result_t fr;
    c.subShift = 0;
    vo_frac_search( &c, ir.mvX, ir.mvY, cfg->useHadME, j->imv == 3, &fr );
    const int qx = ( ir.mvX << 2 ) + ( fr.halfX << 1 ) + fr.qterX, qy = ( ir.mvY << 2 ) + ( fr.halfY << 1 ) + fr.qterY;
    c.mv.costScale = 0;
    const unsigned mvBits = vo_mv_bits( &c.mv, qx, qy, c.imvShift );
    bits += mvBits;
    const uint64_t costMvBits = ( uint64_t )( j->motionLambda * mvBits ), costBits = ( uint64_t )( j->motionLambda * bits );
    res->cost      = ( uint64_t )( floor( fWeight * ( ( double ) fr.cost - ( double ) costMvBits ) ) + ( double ) costBits );
    res->mvHor     = qx << 2;
    res->mvVer     = qy << 2;
    res->mvPredHor = j->mvPredHor; res->mvPredVer = j->mvPredVer; res->mvpIdx = j->mvpIdx; res->bits = bits;
  }
  else
  {
    static const int testPos[9][2] = { { 0, 0 }, { -1, -1 }, { -1, 0 }, { -1, 1 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { 1, 0 }, { 1, 1 } };
    const int  sh = vo_amvr_shift( j->imv );
    const int  rcH = ir.mvX << 4, rcV = ir.mvY << 4;
    bits -= j->mvpIdxBits[j->mvpIdx];
    int base[2][2];
    for( int i = 0; i < 2; i++ )
    {
      base[i][0] = rcH - j->amvpCand[i][0]; base[i][1] = rcV - j->amvpCand[i][1];
      for( int k = 0; k < 2; k++ ) { vo_to_amvr( &base[i][k], j->imv ); base[i][k] <<= sh; }   /* roundTransPrecInternal2Amvr */
    }
    uint64_t bestDist = UINT64_MAX, satd = 0;
    int      bestH = rcH, bestV = rcV, bestBits = 0, bestIdx = j->mvpIdx;
    for( int pos = 0; pos < 9; pos++ )
    {
      int test[2][2] = { { 0, 0 }, { 0, 0 } };
      for( int i = 0; i < j->numAmvpCand; i++ )
      {
        test[i][0] = ( testPos[pos][0] << sh ) + base[i][0] + j->amvpCand[i][0];
        test[i][1] = ( testPos[pos][1] << sh ) + base[i][1] + j->amvpCand[i][1];
        uint64_t dist;
        if( i == 0 || test[0][0] != test[1][0] || test[0][1] != test[1][1] )
        {
          int th = test[i][0], tv = test[i][1];
          vo_clip_mv( &th, &tv, &c );
          const int16_t *cur = j->ref + ( ptrdiff_t )( tv >> 4 ) * j->refStride + ( th >> 4 );
          const uint64_t d   = cfg->useHadME ? vo_satd( pat, ps, cur, j->refStride, w, h ) : vo_sad( pat, ps, cur, j->refStride, w, h, 0 );
          dist = satd = ( uint64_t )( ( double ) d * fWeight );
        }
        else dist = satd;
        int mvBits = ( int ) j->mvpIdxBits[i];
        int ph = j->amvpCand[i][0], pv = j->amvpCand[i][1], mh = test[i][0], mv = test[i][1];
        vo_to_amvr( &ph, j->imv ); vo_to_amvr( &pv, j->imv ); vo_to_amvr( &mh, j->imv ); vo_to_amvr( &mv, j->imv );
        vo_mvcost_t mc = c.mv;
        mc.predHor = ph; mc.predVer = pv; mc.costScale = 0;
        mvBits += ( int ) vo_mv_bits( &mc, mh, mv, 0 );
        dist += ( uint64_t )( j->motionLambda * ( unsigned ) mvBits );
        if( dist < bestDist ) { bestDist = dist; bestH = test[i][0]; bestV = test[i][1]; bestIdx = i; bestBits = mvBits; }
      }
    }
    res->mvHor = bestH; res->mvVer = bestV; res->mvpIdx = bestIdx;
    res->mvPredHor = j->amvpCand[bestIdx][0]; res->mvPredVer = j->amvpCand[bestIdx][1];
    bits += ( unsigned ) bestBits;
    res->bits = bits;
    res->cost = bestDist - ( uint64_t )( j->motionLambda * ( unsigned ) bestBits ) + ( uint64_t )( j->motionLambda * bits );
  }
  free( tmp );
}


/* ---- AMVP helpers of predInterSearch ---------------------------------------------------------------------------------------------- */

void vo_estimate_mvp_amvp( const vo_mest_job_t *j, int *mvpIdx, int *mvPredHor, int *mvPredVer, uint64_t *distBiP )
{
  /* InterSearch.cpp:3088-3128: iBestIdx = 0, cBestMv = mvCand[0]; for every candidate xGetTemplateCost; `uiBestCost > uiTmpCost` keeps the first minimum */
  uint64_t best = UINT64_MAX;
  int      bestIdx = 0;
  static int16_t pred[128 * 128];
  for( int i = 0; i < j->numAmvpCand; i++ )
  {
    /* xGetTemplateCost (:3235-3270): clipMv (clipMvInPic, Mv.cpp:56-74), xPredInterBlk uni-directional (rounded, clipped), DF_SAD + getCost( m_auiMVPIdxCost ) */
    int       h = j->amvpCand[i][0], v = j->amvpCand[i][1];
    const int horMax = ( j->picW + 8 - j->puX - 1 ) << 4, horMin = ( -j->ctuSize - 8 - j->puX + 1 ) << 4;
    const int verMax = ( j->picH + 8 - j->puY - 1 ) << 4, verMin = ( -j->ctuSize - 8 - j->puY + 1 ) << 4;
    h = h > horMax ? horMax : ( h < horMin ? horMin : h );
    v = v > verMax ? verMax : ( v < verMin ? verMin : v );
    vo_mc_luma( j->ref, j->refStride, j->w, j->h, h, v, 0, j->bitDepth, j->imv == 3, pred, j->w );
    const uint64_t cost = vo_sad( j->org, j->orgStride, pred, j->w, j->w, j->h, 0 ) + ( uint64_t ) ( j->motionLambda * j->mvpIdxBits[i] );
    if( best > cost ) { best = cost; bestIdx = i; }
  }
  *mvpIdx = bestIdx; *mvPredHor = j->amvpCand[bestIdx][0]; *mvPredVer = j->amvpCand[bestIdx][1]; *distBiP = best;
}

void vo_check_best_mvp( double motionLambda, int imv, int numCand, const int cands[2][2], const unsigned idxBits[2], int mvHor, int mvVer,
                        int *mvPredHor, int *mvPredVer, int *mvpIdx, unsigned *bits, uint64_t *cost )
{
  /* InterSearch.cpp:3185-3232 */
  if( imv > 0 && imv < 3 ) return;
  if( numCand < 2 ) return;
  const int sh = vo_amvr_shift( imv );
  const int mh = vo_prec_down( mvHor, sh ), mv = vo_prec_down( mvVer, sh );
  const int orgBits = ( int ) ( vo_eg_bits( mh - vo_prec_down( *mvPredHor, sh ) ) + vo_eg_bits( mv - vo_prec_down( *mvPredVer, sh ) ) + idxBits[*mvpIdx] );
  int bestBits = orgBits, bestIdx = *mvpIdx;
  for( int i = 0; i < numCand; i++ )
  {
    if( i == *mvpIdx ) continue;
    const int b = ( int ) ( vo_eg_bits( mh - vo_prec_down( cands[i][0], sh ) ) + vo_eg_bits( mv - vo_prec_down( cands[i][1], sh ) ) + idxBits[i] );
    if( b < bestBits ) { bestBits = b; bestIdx = i; }
  }
  if( bestIdx != *mvpIdx )
  {
    *mvPredHor = cands[bestIdx][0]; *mvPredVer = cands[bestIdx][1];
    *mvpIdx = bestIdx;
    const unsigned org = *bits;
    *bits = org - ( unsigned ) orgBits + ( unsigned ) bestBits;
    *cost = ( *cost - ( uint64_t ) ( motionLambda * org ) ) + ( uint64_t ) ( motionLambda * *bits );
  }
}


/* ---- affine prediction and affine motion estimation ------------------------------------------------------------------------------- */
static void vo_round_affine_mv( int *x, int *y, int shift ) { const int o = 1 << ( shift - 1 ); *x = ( *x + o - ( *x >= 0 ) ) >> shift; *y = ( *y + o - ( *y >= 0 ) ) >> shift; }   /* Mv.cpp:47-52 */
static int vo_ilog2( int v ) { int r = 0; while( ( 2 << r ) <= v ) r++; return r; }

/* isSubblockVectorSpreadOverLimit (InterPrediction.cpp:816-854) */
static int vo_spread_over_limit( int a, int b, int c, int d, int predType )
{
#define MX( x, y ) ( ( x ) > ( y ) ? ( x ) : ( y ) )
#define MN( x, y ) ( ( x ) < ( y ) ? ( x ) : ( y ) )
  const int s4 = 4 << 11, tap = 6;
  if( predType == 3 )
  {
    int rw = MX( MX( 0, 4 * a + s4 ), MX( 4 * c, 4 * a + 4 * c + s4 ) ) - MN( MN( 0, 4 * a + s4 ), MN( 4 * c, 4 * a + 4 * c + s4 ) );
    int rh = MX( MX( 0, 4 * b ), MX( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - MN( MN( 0, 4 * b ), MN( 4 * d + s4, 4 * b + 4 * d + s4 ) );
    rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
    return rw * rh > ( tap + 9 ) * ( tap + 9 );
  }
  int rw = MX( 0, 4 * a + s4 ) - MN( 0, 4 * a + s4 ), rh = MX( 0, 4 * b ) - MN( 0, 4 * b );
  rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
  if( rw * rh > ( tap + 9 ) * ( tap + 5 ) ) return 1;
  rw = MX( 0, 4 * c ) - MN( 0, 4 * c ); rh = MX( 0, 4 * d + s4 ) - MN( 0, 4 * d + s4 );
  rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
  return rw * rh > ( tap + 5 ) * ( tap + 9 );
#undef MX
#undef MN
}

/* InterPrediction::xPredAffineBlk, luma (InterPrediction.cpp:856-1232): one vector per 4x4 sub-block from the control-point vectors, the 6-tap
 * sub-block interpolation (m_lumaFilter4x4), optionally PROF (gradients of the 14-bit prediction x per-sample vector offsets, Buffer.cpp:45-70, 130-147) */
void vo_pred_affine_blk( const vo_affine_pred_t *p, const int mv[3][2], int bi, int16_t *dst, int dstStride )
{
  const int iBit = 7, w = p->w, h = p->h;
  int dHX = ( mv[1][0] - mv[0][0] ) << ( iBit - vo_ilog2( w ) ), dHY = ( mv[1][1] - mv[0][1] ) << ( iBit - vo_ilog2( w ) ), dVX, dVY;
  if( p->sixParam ) { dVX = ( mv[2][0] - mv[0][0] ) << ( iBit - vo_ilog2( h ) ); dVY = ( mv[2][1] - mv[0][1] ) << ( iBit - vo_ilog2( h ) ); }
  else { dVX = -dHY; dVY = dHX; }
  const int baseH = mv[0][0] << iBit, baseV = mv[0][1] << iBit, shift = iBit - 4 + 4;
  const int over = vo_spread_over_limit( dHX, dHY, dVX, dVY, p->interDir );
  int prof = p->profAllowed;
  prof &= !( ( p->sixParam && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1] && mv[0][0] == mv[2][0] && mv[0][1] == mv[2][1] ) || ( !p->sixParam && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1] ) );
  prof &= !over;
  const int thr = 1 << ( iBit + ( p->profIsBi ? 1 : 0 ) );
  prof &= !p->profNeedsLargeGrad || dHX > thr || dHY > thr || dVX > thr || dVY > thr || dHX < -thr || dHY < -thr || dVX < -thr || dVY < -thr;
  const int isLast = prof ? 0 : !bi;
  int dMvH[16], dMvV[16];
  if( prof )
  {
    const int qHX = dHX << 2, qHY = dHY << 2, qVX = dVX << 2, qVY = dVY << 2;
    dMvH[0] = ( ( dHX + dVX ) << 1 ) - ( ( qHX + qVX ) << 1 );
    dMvV[0] = ( ( dHY + dVY ) << 1 ) - ( ( qHY + qVY ) << 1 );
    for( int x = 1; x < 4; x++ ) { dMvH[x] = dMvH[x - 1] + qHX; dMvV[x] = dMvV[x - 1] + qHY; }
    for( int y = 1; y < 4; y++ )
      for( int x = 0; x < 4; x++ ) { dMvH[y * 4 + x] = dMvH[( y - 1 ) * 4 + x] + qVX; dMvV[y * 4 + x] = dMvV[( y - 1 ) * 4 + x] + qVY; }
    for( int i = 0; i < 16; i++ )
    {
      vo_round_affine_mv( &dMvH[i], &dMvV[i], 8 );
      dMvH[i] = vo_clip3( -31, 31, dMvH[i] ); dMvV[i] = vo_clip3( -31, 31, dMvV[i] );
    }
  }
  const int horMax = ( p->picW + 8 - p->puX - 1 ) << 4, horMin = ( -p->ctuSize - 8 - p->puX + 1 ) << 4;
  const int verMax = ( p->picH + 8 - p->puY - 1 ) << 4, verMin = ( -p->ctuSize - 8 - p->puY + 1 ) << 4;
  const int ifShift = 14 - p->bitDepth > 2 ? 14 - p->bitDepth : 2;
  for( int y = 0; y < h; y += 4 )
    for( int x = 0; x < w; x += 4 )
    {
      int mh, mvv;
      if( !over ) { mh = baseH + dHX * ( 2 + x ) + dVX * ( 2 + y ); mvv = baseV + dHY * ( 2 + x ) + dVY * ( 2 + y ); }
      else { mh = baseH + dHX * ( w >> 1 ) + dVX * ( h >> 1 ); mvv = baseV + dHY * ( w >> 1 ) + dVY * ( h >> 1 ); }
      vo_round_affine_mv( &mh, &mvv, shift );
      mh = vo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mh ); mvv = vo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mvv );   /* clipToStorageBitDepth */
      mh = vo_clip3( horMin, horMax, mh ); mvv = vo_clip3( verMin, verMax, mvv );                                 /* clipMv (pu position and size) */
      const int16_t *r0 = p->ref + ( ptrdiff_t ) y * p->refStride + x;
      if( !prof )
      {
        vo_mc_luma( r0, p->refStride, 4, 4, mh, mvv, !isLast, p->bitDepth, 0, dst + ( ptrdiff_t ) y * dstStride + x, dstStride );
        continue;
      }
      /* PROF: 14-bit prediction of the sub-block inside a one-sample ring of integer reference samples */
      int16_t ext[6 * 8], gx[6 * 6], gy[6 * 6];
      const int es = 8, xInt = mh >> 4, yInt = mvv >> 4, xFrac = mh & 15, yFrac = mvv & 15;
      vo_mc_luma( r0, p->refStride, 4, 4, mh, mvv, 1, p->bitDepth, 0, ext + es + 1, es );
      const int16_t *rb = r0 + ( ptrdiff_t ) yInt * p->refStride + xInt;
      const int      xo = xFrac >> 3, yo = yFrac >> 3;
      const int16_t *rp = rb - ( 1 - yo ) * p->refStride + xo - 1;
      for( int i = 0; i < 6; i++ )
      {
        ext[i]          = ( int16_t ) ( ( rp[i] << ifShift ) - 8192 );
        ext[5 * es + i] = ( int16_t ) ( ( rp[i + 5 * p->refStride] << ifShift ) - 8192 );
      }
      rp = rb + yo * p->refStride + xo;
      for( int j = 0; j < 4; j++ )
      {
        ext[( j + 1 ) * es]     = ( int16_t ) ( ( rp[j * p->refStride - 1] << ifShift ) - 8192 );
        ext[( j + 1 ) * es + 5] = ( int16_t ) ( ( rp[j * p->refStride + 4] << ifShift ) - 8192 );
      }
      for( int j = 0; j < 4; j++ )
        for( int i = 0; i < 4; i++ )
        {
          const int16_t *c = ext + ( j + 1 ) * es + i + 1;
          gy[( j + 1 ) * 6 + i + 1] = ( int16_t ) ( ( c[es] >> 6 ) - ( c[-es] >> 6 ) );
          gx[( j + 1 ) * 6 + i + 1] = ( int16_t ) ( ( c[1] >> 6 ) - ( c[-1] >> 6 ) );
        }
      const int dILimit = 1 << ( p->bitDepth + 1 > 13 ? p->bitDepth + 1 : 13 );
      const int offset = ( 1 << ( ifShift - 1 ) ) + 8192;
      for( int j = 0; j < 4; j++ )
        for( int i = 0; i < 4; i++ )
        {
          int dI = dMvH[j * 4 + i] * gx[( j + 1 ) * 6 + i + 1] + dMvV[j * 4 + i] * gy[( j + 1 ) * 6 + i + 1];
          dI = vo_clip3( -dILimit, dILimit - 1, dI );
          int16_t v = ( int16_t ) ( ext[( j + 1 ) * es + i + 1] + dI );
          if( !bi ) { int t = ( v + offset ) >> ifShift; v = ( int16_t ) vo_clip3( 0, ( 1 << p->bitDepth ) - 1, ( int16_t ) t ); }
          dst[( ptrdiff_t ) ( y + j ) * dstStride + x + i] = v;
        }
    }
}

void vo_solve_equal( double eq[7][7], int order, double *para )
{
  for( int k = 0; k < order; k++ ) para[k] = 0.;
  for( int i = 1; i < order; i++ )
  {
    double temp = fabs( eq[i][i - 1] );
    int    idx = i;
    for( int j = i + 1; j < order + 1; j++ ) if( fabs( eq[j][i - 1] ) > temp ) { temp = fabs( eq[j][i - 1] ); idx = j; }
    if( idx != i )
      for( int j = 0; j < order + 1; j++ ) { eq[0][j] = eq[i][j]; eq[i][j] = eq[idx][j]; eq[idx][j] = eq[0][j]; }
    if( eq[i][i - 1] == 0. ) return;
    for( int j = i + 1; j < order + 1; j++ )
      for( int k = i; k < order + 1; k++ ) eq[j][k] = eq[j][k] - eq[i][k] * eq[j][i - 1] / eq[i][i - 1];
  }
  if( eq[order][order - 1] == 0. ) return;
  para[order - 1] = eq[order][order] / eq[order][order - 1];
  for( int i = order - 2; i >= 0; i-- )
  {
    if( eq[i + 1][i] == 0. ) { for( int k = 0; k < order; k++ ) para[k] = 0.; return; }
    double temp = 0;
    for( int j = i + 1; j < order; j++ ) temp += eq[i + 1][j] * para[j];
    para[i] = ( eq[i + 1][order] - temp ) / eq[i + 1][i];
  }
}

static const int vo_affine_prec_shift[3] = { 2, 0, 4 };   /* INTERNAL -> m_amvrPrecAffine[imv] = QUARTER, SIXTEENTH, INT */
static int vo_prec_dn( int v, int rs ) { if( rs == 0 ) return v; const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }
static void vo_round_affine_prec( int *mv, int imv ) { const int rs = vo_affine_prec_shift[imv]; mv[0] = vo_prec_dn( mv[0], rs ) << rs; mv[1] = vo_prec_dn( mv[1], rs ) << rs; }   /* roundAffinePrecInternal2Amvr */

/* xCalcAffineMVBits (InterSearch.cpp:3067-3085) */
static unsigned vo_affine_mv_bits( int sixParam, int imv, const int mv[3][2], const int pred[3][2] )
{
  const int n = sixParam ? 3 : 2, rs = vo_affine_prec_shift[imv];
  unsigned  bits = 0;
  for( int v = 0; v < n; v++ )
  {
    int ph = v == 0 ? pred[0][0] : pred[v][0] + mv[0][0] - pred[0][0], pv = v == 0 ? pred[0][1] : pred[v][1] + mv[0][1] - pred[0][1];
    ph = vo_prec_dn( ph, rs ); pv = vo_prec_dn( pv, rs );
    bits += vo_eg_bits( vo_prec_dn( mv[v][0], rs ) - ph ) + vo_eg_bits( vo_prec_dn( mv[v][1], rs ) - pv );
  }
  return bits;
}

void vo_affine_motion_estimation( const vo_affine_me_job_t *j, vo_affine_me_result_t *res )
{
  const vo_affine_pred_t *p = &j->pred;
  const int w = p->w, h = p->h, six = p->sixParam, mvNum = six ? 3 : 2, paraNum = six ? 7 : 5;
  static int16_t pat[128 * 128], pred[128 * 128], err[128 * 128];
  static int32_t deri[2][128 * 128];
  const double fWeight = j->bi ? 0.5 : 1.0;
  for( int y = 0; y < h; y++ ) memcpy( pat + y * w, j->org + ( ptrdiff_t ) y * j->orgStride, sizeof( int16_t ) * w );
  if( j->bi ) vo_remove_high_freq( pat, w, j->otherPred, j->otherStride, w, h );
  const int horMax = ( p->picW + 8 - p->puX - 1 ) << 4, horMin = ( -p->ctuSize - 8 - p->puX + 1 ) << 4;
  const int verMax = ( p->picH + 8 - p->puY - 1 ) << 4, verMin = ( -p->ctuSize - 8 - p->puY + 1 ) << 4;
#define CLIPMV( m ) do { ( m )[0] = vo_clip3( horMin, horMax, ( m )[0] ); ( m )[1] = vo_clip3( verMin, verMax, ( m )[1] ); } while( 0 )
#define DIST() ( j->useSatd ? vo_satd( pred, w, pat, w, w, h ) : vo_sad( pred, w, pat, w, w, h, 0 ) )
#define RATE( b ) ( ( uint64_t ) ( j->lambda * ( b ) ) )
  int tmp[3][2], best[3][2];
  memcpy( tmp, j->mv, sizeof( tmp ) );
  for( int i = 0; i < mvNum; i++ ) { CLIPMV( tmp[i] ); vo_round_affine_prec( tmp[i], j->imv ); }
  vo_pred_affine_blk( p, tmp, 0, pred, w );
  uint64_t costBest = DIST();
  unsigned bitsBest = j->bits + vo_affine_mv_bits( six, j->imv, tmp, j->mvPred );
  costBest = ( uint64_t ) ( floor( fWeight * ( double ) costBest ) + ( double ) RATE( bitsBest ) );
  memcpy( best, tmp, sizeof( best ) );
  int iterTime = six ? ( j->bi ? 3 : 4 ) : ( j->bi ? 3 : 5 );
  if( !j->useAffineType ) iterTime = j->bi ? 5 : 7;
  int prev[7][3][2];
  res->iterations = 0; res->refinements = 0;
  for( int iter = 0; iter < iterTime; iter++ )
  {
    memcpy( prev[iter], tmp, sizeof( tmp ) );
    for( int i = 0; i < w * h; i++ ) err[i] = ( int16_t ) ( pat[i] - pred[i] );
    vo_sobel( 0, pred, w, deri[0], w, w, h );
    vo_sobel( 1, pred, w, deri[1], w, w, h );
    int64_t eq[7][7];
    memset( eq, 0, sizeof( eq ) );
    vo_equal_coeff( err, w, deri[0], deri[1], w, eq, w, h, six );
    double deq[7][7], para[6], dmv[6] = { 0, 0, 0, 0, 0, 0 };
    for( int r = 0; r < paraNum; r++ ) for( int c = 0; c < paraNum; c++ ) deq[r][c] = ( double ) eq[r][c];
    vo_solve_equal( deq, paraNum - 1, para );
    dmv[0] = para[0]; dmv[2] = para[2];
    if( six ) { dmv[1] = para[1] * w + para[0]; dmv[3] = para[3] * w + para[2]; dmv[4] = para[4] * h + para[0]; dmv[5] = para[5] * h + para[2]; }
    else { dmv[1] = para[1] * w + para[0]; dmv[3] = -para[3] * w + para[2]; }
    static const int normShift[3] = { 2, 4, 2 }, stepShift[3] = { 2, 0, 2 };
    const int mult = 1 << normShift[j->imv], ms = stepShift[j->imv];
#define SGN( x ) ( ( x ) >= 0 ? 1 : -1 )
    int delta[3][2];
    delta[0][0] = ( int ) ( dmv[0] * mult + SGN( dmv[0] ) * 0.5 ) << ms; delta[0][1] = ( int ) ( dmv[2] * mult + SGN( dmv[2] ) * 0.5 ) << ms;
    delta[1][0] = ( int ) ( dmv[1] * mult + SGN( dmv[1] ) * 0.5 ) << ms; delta[1][1] = ( int ) ( dmv[3] * mult + SGN( dmv[3] ) * 0.5 ) << ms;
    delta[2][0] = delta[2][1] = 0;
    if( six ) { delta[2][0] = ( int ) ( dmv[4] * mult + SGN( dmv[4] ) * 0.5 ) << ms; delta[2][1] = ( int ) ( dmv[5] * mult + SGN( dmv[5] ) * 0.5 ) << ms; }
#undef SGN
    if( !j->amvrEncOpt )
    {
      int allZero = 0;
      for( int i = 0; i < mvNum; i++ )
      {
        int d[2] = { delta[i][0], delta[i][1] };
        if( j->imv == 2 ) { d[0] = vo_prec_dn( d[0], 3 ) << 3; d[1] = vo_prec_dn( d[1], 3 ) << 3; }   /* roundToPrecision( INTERNAL, HALF ) */
        if( d[0] != 0 || d[1] != 0 ) { allZero = 0; break; }
        allZero = 1;
      }
      if( allZero ) break;
    }
    for( int i = 0; i < mvNum; i++ )
    {
      tmp[i][0] = vo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, tmp[i][0] + delta[i][0] );
      tmp[i][1] = vo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, tmp[i][1] + delta[i][1] );
      vo_round_affine_prec( tmp[i], j->imv );
      CLIPMV( tmp[i] );
    }
    if( j->amvrEncOpt )
    {
      int identical = 0;
      for( int k = iter; k >= 0; k-- )
        if( tmp[0][0] == prev[k][0][0] && tmp[0][1] == prev[k][0][1] && tmp[1][0] == prev[k][1][0] && tmp[1][1] == prev[k][1][1] )
        {
          identical = six ? ( tmp[2][0] == prev[k][2][0] && tmp[2][1] == prev[k][2][1] ) : 1;
          if( identical ) break;
        }
      if( identical ) break;
    }
    vo_pred_affine_blk( p, tmp, 0, pred, w );
    res->iterations++;
    uint64_t c = DIST();
    const unsigned b = j->bits + vo_affine_mv_bits( six, j->imv, tmp, j->mvPred );
    c = ( uint64_t ) ( floor( fWeight * ( double ) c ) + ( double ) RATE( b ) );
    if( c < costBest ) { costBest = c; bitsBest = b; memcpy( best, tmp, sizeof( best ) ); }
  }
#define CHECK_CPMV( m ) do { vo_pred_affine_blk( p, m, 0, pred, w ); res->refinements++; uint64_t c_ = DIST(); const unsigned b_ = j->bits + vo_affine_mv_bits( six, j->imv, m, j->mvPred ); \
    c_ = ( uint64_t ) ( floor( fWeight * ( double ) c_ ) + ( double ) RATE( b_ ) ); if( c_ < costBest ) { costBest = c_; bitsBest = b_; memcpy( best, m, sizeof( best ) ); changed = 1; } } while( 0 )
  static const int mvShiftTab[3] = { 2, 0, 4 };
  const int mvShift = mvShiftTab[j->imv];
  if( ( double ) costBest <= 1.0 * ( double ) j->hevcCost )
  {
    int me[3][2], dMv[2], changed = 0;
    memcpy( me, best, sizeof( me ) );
    dMv[0] = me[0][0] - j->mvPred[0][0]; dMv[1] = me[0][1] - j->mvPred[0][1];
    for( int k = 0; k < mvNum; k++ )
    {
      const int ph = j->mvPred[k][0] + ( k ? dMv[0] : 0 ), pv = j->mvPred[k][1] + ( k ? dMv[1] : 0 );
      if( me[k][0] != ph || me[k][1] != pv )
      {
        memcpy( tmp, me, sizeof( tmp ) );
        tmp[k][0] = ph; tmp[k][1] = pv;
        CHECK_CPMV( tmp );
      }
    }
    if( me[0][0] != j->mvPred[0][0] || me[0][1] != j->mvPred[0][1] )
    {
      memcpy( tmp, me, sizeof( tmp ) );
      for( int i = 1; i < mvNum; i++ ) { tmp[i][0] -= dMv[0]; tmp[i][1] -= dMv[1]; }
      tmp[0][0] = j->mvPred[0][0]; tmp[0][1] = j->mvPred[0][1];
      CHECK_CPMV( tmp );
    }
    if( six && ( me[1][0] != j->mvPred[1][0] + dMv[0] || me[1][1] != j->mvPred[1][1] + dMv[1] ) && ( me[2][0] != j->mvPred[2][0] + dMv[0] || me[2][1] != j->mvPred[2][1] + dMv[1] ) )
    {
      memcpy( tmp, me, sizeof( tmp ) );
      tmp[1][0] = j->mvPred[1][0] + dMv[0]; tmp[1][1] = j->mvPred[1][1] + dMv[1];
      tmp[2][0] = j->mvPred[2][0] + dMv[0]; tmp[2][1] = j->mvPred[2][1] + dMv[1];
      CHECK_CPMV( tmp );
    }
    static const int testPos[8][2] = { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, 1 }, { 1, -1 } };
    const int maxRound = j->imv ? 3 : ( ( j->amvrEncOpt && j->lowDelayRounds ) ? 2 : 3 );   /* :5712 */
    for( int rnd = 0; rnd < maxRound; rnd++ )
    {
      int modelChange = 0;
      for( int k = 0; k < mvNum; k++ )
      {
        int loopChange = 0;
        for( int it = 0; it < 2; it++ )
        {
          if( it == 1 && !loopChange ) break;
          int center[3][2];
          memcpy( center, best, sizeof( center ) );
          memcpy( tmp, best, sizeof( tmp ) );
          for( int i = it == 0 ? 0 : 4; i < ( it == 0 ? 4 : 8 ); i++ )
          {
            tmp[k][0] = center[k][0] + ( testPos[i][0] << mvShift ); tmp[k][1] = center[k][1] + ( testPos[i][1] << mvShift );
            CLIPMV( tmp[k] );
            changed = 0;
            CHECK_CPMV( tmp );
            if( changed ) { modelChange = 1; loopChange = 1; }
          }
        }
      }
      if( !modelChange ) break;
    }
    ( void ) changed;
  }
#undef CHECK_CPMV
#undef CLIPMV
#undef DIST
#undef RATE
  memcpy( res->mv, best, sizeof( best ) );
  res->bits = bitsBest; res->cost = costBest;
}


/* ---- LFNST at TU level ------------------------------------------------------------------------------------------------------------ */
void vo_lfnst_scan( int w, int h, int32_t *pos )
{
  /* the first 16 (TUs 4 wide or high) or 48 positions of g_scanOrder[SCAN_GROUPED_4x4][SCAN_DIAG] / g_coefTopLeftDiagScan8x8 (Rom.cpp: up-right
   * diagonal scan inside 4x4 coefficient groups, the groups of the 8x8 region in the same diagonal order) */
  const int sb8 = w >= 8 && h >= 8;
  int       gxs[4], gys[4], ng = 0;
  for( int d = 0; d < 3 && sb8; d++ )
    for( int gy = d < 1 ? d : 1; gy >= 0 && d - gy <= 1; gy-- ) { gxs[ng] = d - gy; gys[ng] = gy; ng++; }
  if( !sb8 ) { gxs[0] = gys[0] = 0; ng = 1; }
  int k = 0;
  for( int g = 0; g < ng && k < ( sb8 ? 48 : 16 ); g++ )
    for( int d = 0; d < 7; d++ )
      for( int y = d < 3 ? d : 3; y >= 0 && d - y <= 3; y-- ) pos[k++] = ( gxs[g] * 4 + d - y ) + ( gys[g] * 4 + y ) * w;
}

void vo_lfnst_tu( int32_t *coef, int w, int h, const int8_t *M, int transpose, int inverse )
{
  const int sb8 = w >= 8 && h >= 8, sb = sb8 ? 8 : 4, trSize = sb8 ? 48 : 16, zo = ( ( w == 4 && h == 4 ) || ( w == 8 && h == 8 ) ) ? 8 : 16;
  int32_t   scan[48], in[48], out[48];
  vo_lfnst_scan( w, h, scan );
  memset( in, 0, sizeof( in ) );
  if( !inverse )
  {
    /* TrQuant.cpp:456-497: rows 0..3 are sb wide, rows 4..7 (8x8 region) 4 wide; transposed: vector index 8 x + y (x < 4), 32 + 4 (x - 4) + y */
    for( int y = 0; y < sb; y++ )
      for( int x = 0; x < ( y < 4 ? sb : 4 ); x++ )
      {
        const int v = y < 4 ? y * sb + x : 32 + ( y - 4 ) * 4 + x;
        in[v] = transpose ? coef[x * w + y] : coef[y * w + x];
      }
    vo_fwd_lfnst( in, out, M, sb, zo );
    for( int k = 0; k < trSize; k++ ) coef[scan[k]] = out[k];
  }
  else
  {
    for( int k = 0; k < 16; k++ ) in[k] = coef[scan[k]];
    vo_inv_lfnst( in, out, M, sb, zo );
    for( int y = 0; y < sb; y++ )
      for( int x = 0; x < ( y < 4 ? sb : 4 ); x++ )
      {
        const int v = y < 4 ? y * sb + x : 32 + ( y - 4 ) * 4 + x;
        if( transpose ) coef[x * w + y] = out[v]; else coef[y * w + x] = out[v];
      }
  }
}

/* ---- symmetric MVD (SMVD) search of predInterSearch ----------------------------------------------------------------------------------
 * InterSearch::xGetSymmetricCost (InterSearch.cpp:4341-4391), xSymmeticRefineMvSearch (:4393-4503), xSymmetricMotionEstimation (:4506-4518),
 * symmvdCheckBestMvp (:7787-7886) and the SMVD block of predInterSearch that composes them (:2656-2790).  No MCTS constraint, no weighted
 * prediction.  "cur" is the searched list (list 0 in predInterSearch), "tar" the mirrored one. */
static void vo_smvd_pred( const vo_smvd_job_t *j, int l, const int mv[2], int16_t *dst )
{
  int h = mv[0], v = mv[1];
  vo_clip_mv_pic( &h, &v, j->picW, j->picH, j->ctuSize, j->puX, j->puY );
  /* integer vectors alias the reconstruction (:4351-4358), others go through xPredInterBlk (uni-directional: rounded and clipped) -- the copy is the same samples */
  vo_mc_luma( j->ref[l], j->refStride[l], j->w, j->h, h, v, 0, j->bitDepth, j->imv == 3, dst, j->w );
}

static void vo_smvd_pattern( const vo_smvd_job_t *j, const int16_t *predA, int16_t *pat )
{
  /* bufTmp.copyFrom( origBuf ); bufTmp.removeHighFreq( predBufA, clip, clpRngs, getBcwWeight( BcwIdx, tarList ) ) (Buffer.h:417-520, 946-957) */
  const int w = j->w, h = j->h, maxV = ( 1 << j->bitDepth ) - 1;
  for( int y = 0; y < h; y++ ) memcpy( pat + y * w, j->org + ( ptrdiff_t ) y * j->orgStride, sizeof( int16_t ) * w );
  if( j->bcwWeightTar != 4 ) vo_remove_weight_high_freq( pat, w, predA, w, w, h, j->bcwWeightTar );
  else vo_remove_high_freq( pat, w, predA, w, w, h );
  if( j->clipBiPred )
  {
    /* the clipped forms clip the int result before the store */
    const int bw = j->bcwWeightTar;
    const int normalizer = bw != 4 ? ( ( 1 << 16 ) + ( bw > 0 ? ( bw >> 1 ) : -( bw >> 1 ) ) ) / bw : 0;
    for( int y = 0; y < h; y++ )
      for( int x = 0; x < w; x++ )
      {
        const int o = j->org[( ptrdiff_t ) y * j->orgStride + x], p = predA[y * w + x];
        const int v = bw != 4 ? ( o * ( normalizer * 8 ) - p * ( ( 8 - bw ) * normalizer ) + ( 1 << 15 ) ) >> 16 : 2 * o - p;
        pat[y * w + x] = ( int16_t ) vo_clip3( 0, maxV, v );
      }
  }
}

static uint64_t vo_smvd_dist( const vo_smvd_job_t *j, const int16_t *pat, const int16_t *predB )
{
  const double   fWeight = j->bcwWeightTar != 4 ? fabs( ( double ) j->bcwWeightTar / 8.0 ) : 0.5;   /* xGetMEDistortionWeight :7666-7676 */
  const uint64_t d = j->useSatd ? vo_satd( pat, j->w, predB, j->w, j->w, j->h ) : vo_sad( pat, j->w, predB, j->w, j->w, j->h, 0 );
  return ( uint64_t ) floor( fWeight * ( double ) d );
}

uint64_t vo_symmetric_cost( const vo_smvd_job_t *j, const int mvCur[2], const int mvTar[2] )
{
  int16_t *buf = ( int16_t * ) malloc( sizeof( int16_t ) * 3 * j->w * j->h );
  int16_t *a = buf, *b = buf + j->w * j->h, *pat = b + j->w * j->h;
  vo_smvd_pred( j, 0, mvCur, a );
  vo_smvd_pred( j, 1, mvTar, b );
  vo_smvd_pattern( j, a, pat );
  const uint64_t c = vo_smvd_dist( j, pat, b );
  free( buf );
  return c;
}

static unsigned vo_smvd_mv_bits( const vo_smvd_job_t *j, const int mv[2], const int pred[2] )
{
  /* pred / mv .changeTransPrecInternal2Amvr( imv ); getBitsOfVectorWithPredictor( hor, ver, 0 ) at cost scale 0 */
  const int sh = vo_amvr_shift( j->imv );
  return vo_eg_bits( vo_prec_down( mv[0], sh ) - vo_prec_down( pred[0], sh ) ) + vo_eg_bits( vo_prec_down( mv[1], sh ) - vo_prec_down( pred[1], sh ) );
}

static uint64_t vo_smvd_refine( const vo_smvd_job_t *j, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], uint64_t minCost, int pattern,
                                int stepShift, unsigned maxRounds )
{
  static const int cross[4][2]   = { { 0, 1 }, { 1, 0 }, { 0, -1 }, { -1, 0 } };
  static const int diamond[8][2] = { { 0, 2 }, { 1, 1 }, { 2, 0 }, { 1, -1 }, { 0, -2 }, { -1, -1 }, { -2, 0 }, { -1, 1 } };
  const int ( *off )[2] = pattern == 0 ? cross : diamond;
  const int rounding = pattern == 0 ? 4 : 8, mask = rounding - 1;
  int       start = 0, end = pattern == 0 ? 3 : 7;
  for( unsigned round = 0; round < maxRounds; round++ )
  {
    int       bestDirect = -1;
    const int centre[2]  = { mvCur[0], mvCur[1] };
    for( int idx = start; idx <= end; idx++ )
    {
      const int direct  = ( idx + rounding ) & mask;
      const int cand[2] = { centre[0] + ( off[direct][0] << stepShift ), centre[1] + ( off[direct][1] << stepShift ) };
      const int pair[2] = { predTar[0] - ( cand[0] - predCur[0] ), predTar[1] - ( cand[1] - predCur[1] ) };
      uint64_t  cost    = ( uint64_t ) ( j->motionLambda * vo_smvd_mv_bits( j, cand, predCur ) );
      cost += vo_symmetric_cost( j, cand, pair );
      if( cost < minCost ) { minCost = cost; mvCur[0] = cand[0]; mvCur[1] = cand[1]; mvTar[0] = pair[0]; mvTar[1] = pair[1]; bestDirect = direct; }
    }
    if( bestDirect == -1 ) break;
    const int step = pattern == 2 ? 2 - ( bestDirect & 1 ) : 1;
    start = bestDirect - step; end = bestDirect + step;
  }
  return minCost;
}

void vo_symmetric_me( const vo_smvd_job_t *j, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], uint64_t *cost )
{
  /* :4506-4518: step = one AMVR unit, 8 >> imv diamond rounds (pattern 2), then one cross round (pattern 0) */
  const int stepShift = 2 + ( j->imv == 3 ? 1 : ( j->imv << 1 ) );
  *cost = vo_smvd_refine( j, predCur, predTar, mvCur, mvTar, *cost, 2, stepShift, 8u >> j->imv );
  *cost = vo_smvd_refine( j, predCur, predTar, mvCur, mvTar, *cost, 0, stepShift, 1 );
}

void vo_symmvd_check_best_mvp( const vo_smvd_job_t *j, const int curMv[2], int skip, int predSym[2][2], int mvpIdxSym[2], uint64_t *bestCost )
{
  const int w = j->w, h = j->h;
  int16_t  *buf = ( int16_t * ) malloc( sizeof( int16_t ) * 3 * w * h );
  int16_t  *a = buf, *b = buf + w * h, *pat = b + w * h;
  vo_smvd_pred( j, 0, curMv, a );
  vo_smvd_pattern( j, a, pat );
  const int skip0 = skip ? mvpIdxSym[0] : -1, skip1 = skip ? mvpIdxSym[1] : -1;
  for( int i = 0; i < j->numCand[0]; i++ )
    for( int k = 0; k < j->numCand[1]; k++ )
    {
      if( skip0 == i && skip1 == k ) continue;
      const int tar[2] = { j->cand[1][k][0] - curMv[0] + j->cand[0][i][0], j->cand[1][k][1] - curMv[1] + j->cand[0][i][1] };   /* Mv::getSymmvdMv */
      vo_smvd_pred( j, 1, tar, b );
      uint64_t       cost = vo_smvd_dist( j, pat, b );
      const unsigned bits = vo_smvd_mv_bits( j, curMv, j->cand[0][i] ) + j->mvpIdxBits[i] + j->mvpIdxBits[k];
      cost += ( uint64_t ) ( j->motionLambda * bits );
      if( cost < *bestCost )
      {
        *bestCost = cost;
        predSym[0][0] = j->cand[0][i][0]; predSym[0][1] = j->cand[0][i][1]; predSym[1][0] = j->cand[1][k][0]; predSym[1][1] = j->cand[1][k][1];
        mvpIdxSym[0] = i; mvpIdxSym[1] = k;
      }
    }
  free( buf );
}

void vo_smvd_search( const vo_smvd_job_t *job, int numFixed, int numStart, const int starts[][2], unsigned modeBits, vo_smvd_result_t *res )
{
  /* predInterSearch :2656-2790 (curRefList = list 0).  starts: cMvHevcTemp, cMvTemp, [cMvBi] (numFixed, taken as they are), then the m_uniMvList
   * entries newest first (rounded to the AMVR precision, while fewer than 5 distinct candidates are collected).  modeBits: uiMbBits[2] + 1 + BCW index bits */
  vo_smvd_job_t jj = *job;
  vo_smvd_job_t *j = &jj;
  for( int l = 0; l < 2; l++ )
    if( j->numCand[l] > 1 && j->cand[l][0][0] == j->cand[l][1][0] && j->cand[l][0][1] == j->cand[l][1][1] ) j->numCand[l] = 1;   /* :2668-2671 */
  int      predSym[2][2] = { { 0, 0 }, { 0, 0 } }, mvpIdxSym[2] = { 0, 0 };
  uint64_t costStart = UINT64_MAX;
  for( int i = 0; i < j->numCand[0]; i++ )
    for( int k = 0; k < j->numCand[1]; k++ )
    {
      const uint64_t c = vo_symmetric_cost( j, j->cand[0][i], j->cand[1][k] );
      if( c < costStart )
      {
        costStart = c; mvpIdxSym[0] = i; mvpIdxSym[1] = k;
        predSym[0][0] = j->cand[0][i][0]; predSym[0][1] = j->cand[0][i][1]; predSym[1][0] = j->cand[1][k][0]; predSym[1][1] = j->cand[1][k][1];
      }
    }
  int mvCur[2] = { predSym[0][0], predSym[0][1] }, mvTar[2] = { predSym[1][0], predSym[1][1] };
  costStart += ( uint64_t ) ( j->motionLambda * ( vo_smvd_mv_bits( j, mvCur, predSym[0] ) + j->mvpIdxBits[mvpIdxSym[0]] + j->mvpIdxBits[mvpIdxSym[1]] ) );
  /* distinct start candidates (smmvdCandsGen) */
  int cands[16][2], nc = 0;
  for( int s = 0; s < numStart; s++ )
  {
    int c[2] = { starts[s][0], starts[s][1] };
    if( s >= numFixed )
    {
      if( nc >= 5 ) break;
      if( j->imv ) { const int sh = vo_amvr_shift( j->imv ); c[0] = vo_prec_down( c[0], sh ) * ( 1 << sh ); c[1] = vo_prec_down( c[1], sh ) * ( 1 << sh ); }   /* roundTransPrecInternal2Amvr */
    }
    int dup = 0;
    for( int q = 0; q < nc; q++ ) dup |= cands[q][0] == c[0] && cands[q][1] == c[1];
    if( !dup ) { cands[nc][0] = c[0]; cands[nc][1] = c[1]; nc++; }
  }
  for( int s = 0; s < nc; s++ )
  {
    int checked = 0;
    for( int i = 0; i < j->numCand[0] && !checked; i++ ) checked |= cands[s][0] == j->cand[0][i][0] && cands[s][1] == j->cand[0][i][1];
    if( checked ) continue;
    const uint64_t before = costStart;
    vo_symmvd_check_best_mvp( j, cands[s], 0, predSym, mvpIdxSym, &costStart );
    if( costStart < before )
    {
      mvCur[0] = cands[s][0]; mvCur[1] = cands[s][1];
      mvTar[0] = predSym[1][0] - mvCur[0] + predSym[0][0]; mvTar[1] = predSym[1][1] - mvCur[1] + predSym[0][1];
    }
  }
  const int      startPt[2] = { mvCur[0], mvCur[1] };
  const uint64_t mvpCost    = ( uint64_t ) ( j->motionLambda * ( j->mvpIdxBits[mvpIdxSym[0]] + j->mvpIdxBits[mvpIdxSym[1]] ) );
  uint64_t       symCost    = costStart - mvpCost;
  vo_symmetric_me( j, predSym[0], predSym[1], mvCur, mvTar, &symCost );
  symCost += mvpCost;
  if( startPt[0] != mvCur[0] || startPt[1] != mvCur[1] ) vo_symmvd_check_best_mvp( j, mvCur, 1, predSym, mvpIdxSym, &symCost );
  symCost += ( uint64_t ) ( j->motionLambda * modeBits );
  mvTar[0] = predSym[1][0] - mvCur[0] + predSym[0][0]; mvTar[1] = predSym[1][1] - mvCur[1] + predSym[0][1];
  res->mvCur[0] = mvCur[0]; res->mvCur[1] = mvCur[1]; res->mvTar[0] = mvTar[0]; res->mvTar[1] = mvTar[1];
  for( int l = 0; l < 2; l++ ) { res->predSym[l][0] = predSym[l][0]; res->predSym[l][1] = predSym[l][1]; res->mvpIdxSym[l] = mvpIdxSym[l]; }
  res->cost = symCost;
}


/* ---- MTS candidate pre-selection of TrQuant::transformNxN( tu, compID, cQP, &trModes, maxCand ) (TrQuant.cpp:950-1019) ------------------------------
 * sumAbs[i]: sum |coef| of candidate i after xT (sum |residual| for a transform-skip candidate, scaled here as :992-1001 scale it); the list is in trModes
 * order (DCT2 first).  A candidate survives when its cost is at most facBB[log2 size] x the first candidate's cost -- the candidate at list position 1
 * against the unscaled cost (:1012) -- and no more than maxCand + 1 have survived before it. */
void vo_mts_select( const int32_t *sumAbs, const uint8_t *mtsIdx, int numCand, int w, int h, int bitDepth, int maxLog2TrDynamicRange, int maxCand, uint8_t *test )
{
  static const double facBB[] = { 1.2, 1.3, 1.3, 1.4, 1.5 };
  int lw = 0, lh = 0;
  while( ( 2 << lw ) <= w ) lw++;
  while( ( 2 << lh ) <= h ) lh++;
  int32_t cost[16];
  for( int i = 0; i < numCand; i++ )
  {
    cost[i] = sumAbs[i];
    if( mtsIdx[i] == 1 )
    {
      double scaleSAD = 1.0;
      if( ( lw + lh ) & 1 ) scaleSAD = 1.0 / 1.414213562;
      scaleSAD *= pow( 2, maxLog2TrDynamicRange - bitDepth - ( ( lw + lh ) >> 1 ) );
      cost[i] = ( int ) ( sumAbs[i] * scaleSAD );
    }
  }
  const int    lg  = lw > lh ? lw : lh;
  const int    fi  = lg - 2 > 0 ? ( lg - 2 > 4 ? 4 : lg - 2 ) : 0;
  const double thr = facBB[fi] * cost[0], thrTS = cost[0];
  int numTests = 0;
  for( int i = 0; i < numCand; i++ )
  {
    const int t = cost[i] <= ( i == 1 ? thrTS : thr ) && numTests <= maxCand;
    test[i] = ( uint8_t ) t;
    numTests += t;
  }
}
