// affine.hip -- affine motion estimation on the device.
//
// Reference: InterPrediction::xPredAffineBlk (CommonLib/InterPrediction.cpp:856-1232: one vector per 4x4 sub-block from the control-point
// vectors, isSubblockVectorSpreadOverLimit :816-854, the 6-tap sub-block interpolation InterpolationFilter.cpp:57-75 / 786-789, PROF
// Buffer.cpp:45-70, 130-147) and InterSearch::xAffineMotionEstimation (EncoderLib/InterSearch.cpp:5340-5775: gradient iterations =
// prediction -> error -> Sobel -> normal equations (AffineGradientSearch.cpp:62-170) -> solveEqual :5215-5284 -> vector update, then the
// control-point refinement around the best model; xCalcAffineMVBits :3067-3085).
//
// One workgroup = one job, from the first prediction to the last refinement step -- nothing returns to the host in between: the pattern
// (org, or 2*org - otherPred) and the current prediction live in LDS, the normal equations are accumulated in exact 64-bit integers, and the
// reference's fp64 arithmetic (the solver, the delta-vector rounding, the cost weighting) runs on the device in IEEE double (division and
// multiplication correctly rounded, no contraction: the build uses -fno-fast-math -ffp-contract=off), so every decision is bit-identical.
#include "ctx.hpp"
#include "had.hpp"

#include <cmath>

namespace
{

__constant__ int16_t c_affTaps4x4[16][8] = {   // m_lumaFilter4x4 (InterpolationFilter.cpp:57-75): the taps every 4x4 (sub-)block uses
  { 0, 0, 0, 64, 0, 0, 0, 0 },      { 0, 1, -3, 63, 4, -2, 1, 0 },    { 0, 1, -5, 62, 8, -3, 1, 0 },    { 0, 2, -8, 60, 13, -4, 1, 0 },
  { 0, 3, -10, 58, 17, -5, 1, 0 },  { 0, 3, -11, 52, 26, -8, 2, 0 },  { 0, 2, -9, 47, 31, -10, 3, 0 },  { 0, 3, -11, 45, 34, -10, 3, 0 },
  { 0, 3, -11, 40, 40, -11, 3, 0 }, { 0, 3, -10, 34, 45, -11, 3, 0 }, { 0, 3, -10, 31, 47, -9, 2, 0 },  { 0, 2, -8, 26, 52, -11, 3, 0 },
  { 0, 1, -5, 17, 58, -10, 3, 0 },  { 0, 1, -4, 13, 60, -8, 2, 0 },   { 0, 1, -3, 8, 62, -5, 1, 0 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };

__device__ __forceinline__ int clip3( int lo, int hi, int v ) { return min( hi, max( lo, v ) ); }
__device__ __forceinline__ void round_affine_mv( int &x, int &y, int shift ) { const int o = 1 << ( shift - 1 ); x = ( x + o - ( x >= 0 ) ) >> shift; y = ( y + o - ( y >= 0 ) ) >> shift; }
__device__ __forceinline__ int ilog2i( int v ) { return 31 - __clz( v ); }
__device__ __forceinline__ int prec_dn( int v, int rs ) { if( rs == 0 ) return v; const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }
__device__ __forceinline__ unsigned eg_bits( int v )
{
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}

struct Mv3 { int v[3][2]; };

struct AffCtx   // block-uniform view of one job
{
  const int16_t *ref;
  int            refStride, w, h, bd, six, interDir, imv;
  int            horMin, horMax, verMin, verMax;
  bool           profAllowed, profLarge, profIsBi;
};

__device__ __forceinline__ bool spread_over_limit( int a, int b, int c, int d, int predType )
{
  const int s4 = 4 << 11, tap = 6;
  if( predType == 3 )
  {
    int rw = max( max( 0, 4 * a + s4 ), max( 4 * c, 4 * a + 4 * c + s4 ) ) - min( min( 0, 4 * a + s4 ), min( 4 * c, 4 * a + 4 * c + s4 ) );
    int rh = max( max( 0, 4 * b ), max( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - min( min( 0, 4 * b ), min( 4 * d + s4, 4 * b + 4 * d + s4 ) );
    rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
    return rw * rh > ( tap + 9 ) * ( tap + 9 );
  }
  int rw = max( 0, 4 * a + s4 ) - min( 0, 4 * a + s4 ), rh = max( 0, 4 * b ) - min( 0, 4 * b );
  rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
  if( rw * rh > ( tap + 9 ) * ( tap + 5 ) ) return true;
  rw = max( 0, 4 * c ) - min( 0, 4 * c ); rh = max( 0, 4 * d + s4 ) - min( 0, 4 * d + s4 );
  rw = ( rw >> 11 ) + tap + 3; rh = ( rh >> 11 ) + tap + 3;
  return rw * rh > ( tap + 5 ) * ( tap + 9 );
}

// 4x4 sub-block at vector (mh, mv): out[16], rounded + clipped (last) or 14-bit intermediates (!last); InterpolationFilter::filter rules (:577-602)
__device__ __forceinline__ void subblock_4x4( const int16_t *r0, int rs, int mh, int mv, bool last, int bd, int out[16] )
{
  const int      xFrac = mh & 15, yFrac = mv & 15, headRoom = max( 2, 14 - bd ), cmax = ( 1 << bd ) - 1;
  const int16_t *src = r0 + ( long ) ( mv >> 4 ) * rs + ( mh >> 4 );
  if( yFrac == 0 || xFrac == 0 )
  {
    if( xFrac == 0 && yFrac == 0 && last )
    {
#pragma unroll
      for( int i = 0; i < 16; i++ ) out[i] = src[( long ) ( i >> 2 ) * rs + ( i & 3 )];
      return;
    }
    const bool     ver = yFrac != 0;
    const int16_t *c   = c_affTaps4x4[ver ? yFrac : xFrac];
    const int      shift = last ? 6 : 6 - headRoom, offset = last ? 32 : -( 8192 << shift ), step = ver ? rs : 1;
#pragma unroll
    for( int i = 0; i < 16; i++ )
    {
      const int16_t *s = src + ( long ) ( i >> 2 ) * rs + ( i & 3 );
      int            sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) s[( long ) ( k - 3 ) * step] * ( int ) c[k];
      int v = ( int ) ( int16_t ) ( ( sum + offset ) >> shift );
      if( last ) v = clip3( 0, cmax, v );
      out[i] = v;
    }
    return;
  }
  const int16_t *ch = c_affTaps4x4[xFrac], *cv = c_affTaps4x4[yFrac];
  const int      sh1 = 6 - headRoom, of1 = -( 8192 << sh1 );
  int            tmp[11][4];
#pragma unroll
  for( int r = 0; r < 11; r++ )
#pragma unroll
    for( int x = 0; x < 4; x++ )
    {
      const int16_t *s = src + ( long ) ( r - 3 ) * rs + x;
      int            sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) s[k - 3] * ( int ) ch[k];
      tmp[r][x] = ( int ) ( int16_t ) ( ( sum + of1 ) >> sh1 );
    }
  const int sh2 = last ? 6 + headRoom : 6, of2 = last ? ( 1 << ( sh2 - 1 ) ) + ( 8192 << 6 ) : 0;
#pragma unroll
  for( int i = 0; i < 16; i++ )
  {
    int sum = 0;
#pragma unroll
    for( int k = 0; k < 8; k++ ) sum += tmp[( i >> 2 ) + k][i & 3] * ( int ) cv[k];
    int v = ( int ) ( int16_t ) ( ( sum + of2 ) >> sh2 );
    if( last ) v = clip3( 0, cmax, v );
    out[i] = v;
  }
}

struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8u { unsigned v[4]; };   // 8 samples from a 2-byte aligned address

// One COLUMN of a 4x4 sub-block at vector (mh, mv): o[4] = rows 0..3 of column q, rounded + clipped (last) or 14-bit intermediates (!last); the same
// arithmetic as subblock_4x4, split so that the four lanes of a quad share one sub-block without talking to each other: a lane filters the 11 rows of
// its column horizontally (ONE 16-byte load per row holds its eight taps' samples), then its own column vertically.
__device__ __forceinline__ void subblock_column( const int16_t *r0, int rs, int mh, int mv, int q, bool last, int bd, int o[4] )
{
  const int      xFrac = mh & 15, yFrac = mv & 15, headRoom = max( 2, 14 - bd ), cmax = ( 1 << bd ) - 1;
  const int16_t *src = r0 + ( long ) ( mv >> 4 ) * rs + ( mh >> 4 ) + q;
  if( xFrac == 0 && yFrac == 0 && last )
  {
#pragma unroll
    for( int y = 0; y < 4; y++ ) o[y] = src[( long ) y * rs];
    return;
  }
  if( yFrac == 0 )   // horizontal only (also the 14-bit copy: taps {0,0,0,64,0,0,0,0})
  {
    const int16_t *c = c_affTaps4x4[xFrac];
    const int      shift = last ? 6 : 6 - headRoom, offset = last ? 32 : -( 8192 << shift );
#pragma unroll
    for( int y = 0; y < 4; y++ )
    {
      const Pel8u v = *reinterpret_cast<const Pel8u *>( src + ( long ) y * rs - 3 );
      int         sum = 0;
#pragma unroll
      for( int k = 0; k < 4; k++ ) sum += ( int ) ( short ) v.v[k] * ( int ) c[2 * k] + ( ( int ) v.v[k] >> 16 ) * ( int ) c[2 * k + 1];
      int r = ( int ) ( int16_t ) ( ( sum + offset ) >> shift );
      if( last ) r = clip3( 0, cmax, r );
      o[y] = r;
    }
    return;
  }
  if( xFrac == 0 )   // vertical only
  {
    const int16_t *c = c_affTaps4x4[yFrac];
    const int      shift = last ? 6 : 6 - headRoom, offset = last ? 32 : -( 8192 << shift );
    int            col[11];
#pragma unroll
    for( int r = 0; r < 11; r++ ) col[r] = src[( long ) ( r - 3 ) * rs];
#pragma unroll
    for( int y = 0; y < 4; y++ )
    {
      int sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += col[y + k] * ( int ) c[k];
      int r = ( int ) ( int16_t ) ( ( sum + offset ) >> shift );
      if( last ) r = clip3( 0, cmax, r );
      o[y] = r;
    }
    return;
  }
  const int16_t *ch = c_affTaps4x4[xFrac], *cv = c_affTaps4x4[yFrac];
  const int      sh1 = 6 - headRoom, of1 = -( 8192 << sh1 );
  int            col[11];
#pragma unroll
  for( int r = 0; r < 11; r++ )
  {
    const Pel8u v = *reinterpret_cast<const Pel8u *>( src + ( long ) ( r - 3 ) * rs - 3 );
    int         sum = 0;
#pragma unroll
    for( int k = 0; k < 4; k++ ) sum += ( int ) ( short ) v.v[k] * ( int ) ch[2 * k] + ( ( int ) v.v[k] >> 16 ) * ( int ) ch[2 * k + 1];
    col[r] = ( int ) ( int16_t ) ( ( sum + of1 ) >> sh1 );
  }
  const int sh2 = last ? 6 + headRoom : 6, of2 = last ? ( 1 << ( sh2 - 1 ) ) + ( 8192 << 6 ) : 0;
#pragma unroll
  for( int y = 0; y < 4; y++ )
  {
    int sum = 0;
#pragma unroll
    for( int k = 0; k < 8; k++ ) sum += col[y + k] * ( int ) cv[k];
    int r = ( int ) ( int16_t ) ( ( sum + of2 ) >> sh2 );
    if( last ) r = clip3( 0, cmax, r );
    o[y] = r;
  }
}

// xPredAffineBlk (luma, uni-directional use of the estimation: bi = false) into sPred[h][w]: a QUAD of lanes per 4x4 sub-block, lane q = column q.
// PROF needs the horizontal neighbours of a sample: the neighbouring lanes of the quad (DPP quad_perm) or, at the sub-block's edge, the ring of
// integer reference samples; the vertical neighbours are the lane's own column and the ring.
__device__ void affine_pred( const AffCtx &c, const Mv3 &m, int16_t *sPred )
{
  const int iBit = 7, w = c.w, h = c.h;
  int dHX = ( m.v[1][0] - m.v[0][0] ) << ( iBit - ilog2i( w ) ), dHY = ( m.v[1][1] - m.v[0][1] ) << ( iBit - ilog2i( w ) ), dVX, dVY;
  if( c.six ) { dVX = ( m.v[2][0] - m.v[0][0] ) << ( iBit - ilog2i( h ) ); dVY = ( m.v[2][1] - m.v[0][1] ) << ( iBit - ilog2i( h ) ); }
  else { dVX = -dHY; dVY = dHX; }
  const int  baseH = m.v[0][0] << iBit, baseV = m.v[0][1] << iBit;
  const bool over = spread_over_limit( dHX, dHY, dVX, dVY, c.interDir );
  bool       prof = c.profAllowed;
  prof = prof && !( ( c.six && m.v[0][0] == m.v[1][0] && m.v[0][1] == m.v[1][1] && m.v[0][0] == m.v[2][0] && m.v[0][1] == m.v[2][1] )
                    || ( !c.six && m.v[0][0] == m.v[1][0] && m.v[0][1] == m.v[1][1] ) );
  prof = prof && !over;
  const int thr = 1 << ( iBit + ( c.profIsBi ? 1 : 0 ) );
  prof = prof && ( !c.profLarge || dHX > thr || dHY > thr || dVX > thr || dVY > thr || dHX < -thr || dHY < -thr || dVX < -thr || dVY < -thr );
  const int ifShift = max( 2, 14 - c.bd );
  const int sbw = w >> 2, nsb = sbw * ( h >> 2 );
  const int q = threadIdx.x & 3;
  for( int sb = threadIdx.x >> 2; sb < nsb; sb += blockDim.x >> 2 )   // whole quads enter and leave together
  {
    const int y = ( sb / sbw ) << 2, x = ( sb - ( sb / sbw ) * sbw ) << 2;
    int       mh, mv;
    if( !over ) { mh = baseH + dHX * ( 2 + x ) + dVX * ( 2 + y ); mv = baseV + dHY * ( 2 + x ) + dVY * ( 2 + y ); }
    else { mh = baseH + dHX * ( w >> 1 ) + dVX * ( h >> 1 ); mv = baseV + dHY * ( w >> 1 ) + dVY * ( h >> 1 ); }
    round_affine_mv( mh, mv, iBit );   // shift = iBit - 4 + MV_FRACTIONAL_BITS_INTERNAL
    mh = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mh ); mv = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mv );
    mh = clip3( c.horMin, c.horMax, mh ); mv = clip3( c.verMin, c.verMax, mv );
    const int16_t *r0 = c.ref + ( long ) y * c.refStride + x;
    int            o[4];
    subblock_column( r0, c.refStride, mh, mv, q, !prof, c.bd, o );
    if( prof )
    {
      // PROF: gradients of the 14-bit prediction inside a ring of integer reference samples, times the per-sample vector offsets
      const int16_t *rb = r0 + ( long ) ( mv >> 4 ) * c.refStride + ( mh >> 4 );
      const int      xo = ( mh & 15 ) >> 3, yo = ( mv & 15 ) >> 3;
      const int16_t *rp = rb - ( long ) ( 1 - yo ) * c.refStride + xo - 1;
      const int      top = ( int ) ( int16_t ) ( ( rp[q + 1] << ifShift ) - 8192 ), bot = ( int ) ( int16_t ) ( ( rp[q + 1 + 5l * c.refStride] << ifShift ) - 8192 );
      const int16_t *rq = rb + ( long ) yo * c.refStride + xo;
      int            lf[4], rt[4];
#pragma unroll
      for( int j = 0; j < 4; j++ )
      {
        lf[j] = __builtin_amdgcn_mov_dpp( o[j], 0x90, 0xF, 0xF, false );   // quad_perm [0, 0, 1, 2]: the lane to the left
        rt[j] = __builtin_amdgcn_mov_dpp( o[j], 0xF9, 0xF, 0xF, false );   // quad_perm [1, 2, 3, 3]: the lane to the right
      }
      if( q == 0 )
      {
#pragma unroll
        for( int j = 0; j < 4; j++ ) lf[j] = ( int ) ( int16_t ) ( ( rq[( long ) j * c.refStride - 1] << ifShift ) - 8192 );
      }
      if( q == 3 )
      {
#pragma unroll
        for( int j = 0; j < 4; j++ ) rt[j] = ( int ) ( int16_t ) ( ( rq[( long ) j * c.refStride + 4] << ifShift ) - 8192 );
      }
      const int qHX = dHX << 2, qHY = dHY << 2, qVX = dVX << 2, qVY = dVY << 2;
      const int d0H = ( ( dHX + dVX ) << 1 ) - ( ( qHX + qVX ) << 1 ), d0V = ( ( dHY + dVY ) << 1 ) - ( ( qHY + qVY ) << 1 );
      const int dILimit = 1 << max( c.bd + 1, 13 ), offset = ( 1 << ( ifShift - 1 ) ) + 8192, cmax = ( 1 << c.bd ) - 1;
      int       res[4];
#pragma unroll
      for( int j = 0; j < 4; j++ )
      {
        int dh = d0H + q * qHX + j * qVX, dv = d0V + q * qHY + j * qVY;
        round_affine_mv( dh, dv, 8 );
        dh = clip3( -31, 31, dh ); dv = clip3( -31, 31, dv );
        const int up = j == 0 ? top : o[j > 0 ? j - 1 : 0], dn = j == 3 ? bot : o[j < 3 ? j + 1 : 3];
        const int gx = ( int ) ( int16_t ) ( ( rt[j] >> 6 ) - ( lf[j] >> 6 ) );
        const int gy = ( int ) ( int16_t ) ( ( dn >> 6 ) - ( up >> 6 ) );
        const int dI = clip3( -dILimit, dILimit - 1, dh * gx + dv * gy );
        const int v  = ( int ) ( int16_t ) ( o[j] + dI );
        res[j] = clip3( 0, cmax, ( int ) ( int16_t ) ( ( v + offset ) >> ifShift ) );
      }
#pragma unroll
      for( int j = 0; j < 4; j++ ) o[j] = res[j];
    }
#pragma unroll
    for( int j = 0; j < 4; j++ ) sPred[( y + j ) * w + x + q] = ( int16_t ) o[j];
  }
}

// getDistPart( DF_HAD / DF_SAD ) of prediction vs pattern over the block (tile shapes of xGetHADs: RdCost.cpp:2837-2931), block-wide sum.
// packed (bitDepth <= 10: |pattern - prediction| <= 3 * 1023): a lane takes one 8x8 unit with the packed 16-bit Hadamard of had.hpp; the two halves of a
// 16x8 / 8x16 tile sit in neighbouring lanes (units in pair order), each returns the finished tile value and the even lane counts it.
template<bool packed>
__device__ __forceinline__ unsigned long long block_dist( const int16_t *sPred, const int16_t *sPat, int w, int h, bool satd, unsigned long long *sRed )
{
  unsigned long long acc = 0;
  if( satd && packed )
  {
    const int nu = ( w >> 3 ) * ( h >> 3 );   // 8x8 units; w, h >= 16 here, so nu is even and pair lanes enter the loop together
    for( int u = threadIdx.x; u < nu; u += blockDim.x )
    {
      int x, y;
      if( w == h ) { const int ux = w >> 3; y = ( u / ux ) << 3; x = ( u - ( u / ux ) * ux ) << 3; }
      else if( w > h ) { const int px = w >> 4, p = u >> 1; y = ( p / px ) << 3; x = ( ( p - ( p / px ) * px ) << 4 ) + ( ( u & 1 ) << 3 ); }
      else { const int px = w >> 3, p = u >> 1; y = ( ( p / px ) << 4 ) + ( ( u & 1 ) << 3 ); x = ( p - ( p / px ) * px ) << 3; }
      v2s D[8][4];
#pragma unroll
      for( int r = 0; r < 8; r++ )
      {
        const uint4 a = *reinterpret_cast<const uint4 *>( sPred + ( y + r ) * w + x ), b = *reinterpret_cast<const uint4 *>( sPat + ( y + r ) * w + x );
        const unsigned aw[4] = { a.x, a.y, a.z, a.w }, bw[4] = { b.x, b.y, b.z, b.w };
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          v2s p, q;
          __builtin_memcpy( &p, &aw[k], 4 ); __builtin_memcpy( &q, &bw[k], 4 );
          D[r][k] = p - q;
        }
      }
      if( w == h ) acc += satd8_packed( D );
      else { const unsigned tv = satd8_pair_packed( D ); acc += ( u & 1 ) ? 0u : tv; }
    }
  }
  else if( satd && !packed )
  {
    const int tw = w > h ? 16 : 8, th = w < h ? 16 : 8, tx = w / tw, nt = tx * ( h / th );
    for( int t = threadIdx.x; t < nt; t += blockDim.x )
    {
      const int      y = ( t / tx ) * th, x = ( t - ( t / tx ) * tx ) * tw;
      const int16_t *o = sPred + y * w + x, *cc = sPat + y * w + x;
      acc += tw == 16 ? had_tile<16, 8>( o, w, cc, w ) : th == 16 ? had_tile<8, 16>( o, w, cc, w ) : had_tile<8, 8>( o, w, cc, w );
    }
  }
  else
  {
    for( int i = threadIdx.x; i < w * h; i += blockDim.x ) acc += ( unsigned ) abs( ( int ) sPred[i] - ( int ) sPat[i] );
  }
  acc = wave_reduce_add_u64( acc );
  __syncthreads();
  if( ( threadIdx.x & 63 ) == 0 ) sRed[threadIdx.x >> 6] = acc;
  __syncthreads();
  unsigned long long t = 0;
  for( int wv = 0; wv < ( int ) ( blockDim.x >> 6 ); wv++ ) t += sRed[wv];   // one or four waves per job
  __syncthreads();
  return t;
}

__device__ __forceinline__ unsigned affine_mv_bits( int six, int imv, const Mv3 &m, const int pred[3][2] )
{
  const int rsTab[3] = { 2, 0, 4 };
  const int n = six ? 3 : 2, rs = rsTab[imv];
  unsigned  bits = 0;
  for( int v = 0; v < n; v++ )
  {
    const int ph = prec_dn( v == 0 ? pred[0][0] : pred[v][0] + m.v[0][0] - pred[0][0], rs ), pv = prec_dn( v == 0 ? pred[0][1] : pred[v][1] + m.v[0][1] - pred[0][1], rs );
    bits += eg_bits( prec_dn( m.v[v][0], rs ) - ph ) + eg_bits( prec_dn( m.v[v][1], rs ) - pv );
  }
  return bits;
}

// solveEqual (InterSearch.cpp:5215-5284), operation for operation; ORDER is a compile-time constant so that the system lives in registers (rows 1 .. ORDER, columns 0 .. ORDER)
template<int ORDER>
__device__ __forceinline__ void solve_equal( double eq[ORDER + 1][ORDER + 1], double *para )
{
  constexpr int order = ORDER;
#pragma unroll
  for( int k = 0; k < order; k++ ) para[k] = 0.;
#pragma unroll
  for( int i = 1; i < order; i++ )
  {
    double temp = fabs( eq[i][i - 1] );
    int    idx = i;
#pragma unroll
    for( int j = i + 1; j < order + 1; j++ ) if( fabs( eq[j][i - 1] ) > temp ) { temp = fabs( eq[j][i - 1] ); idx = j; }
    if( idx != i )
    {
      // the reference swaps rows i and idx through row 0 as scratch; the row index is data dependent, the columns are not
#pragma unroll
      for( int j = 0; j < order + 1; j++ )
      {
        double other = 0.;
#pragma unroll
        for( int r = 2; r < order + 1; r++ ) if( r == idx ) other = eq[r][j];
        eq[0][j] = eq[i][j]; eq[i][j] = other;
#pragma unroll
        for( int r = 2; r < order + 1; r++ ) if( r == idx ) eq[r][j] = eq[0][j];
      }
    }
    if( eq[i][i - 1] == 0. ) return;
#pragma unroll
    for( int j = i + 1; j < order + 1; j++ )
#pragma unroll
      for( int k = i; k < order + 1; k++ ) eq[j][k] = eq[j][k] - eq[i][k] * eq[j][i - 1] / eq[i][i - 1];
  }
  if( eq[order][order - 1] == 0. ) return;
  para[order - 1] = eq[order][order] / eq[order][order - 1];
#pragma unroll
  for( int i = order - 2; i >= 0; i-- )
  {
    if( eq[i + 1][i] == 0. )
    {
#pragma unroll
      for( int k = 0; k < order; k++ ) para[k] = 0.;
      return;
    }
    double temp = 0;
#pragma unroll
    for( int j = i + 1; j < order; j++ ) temp += eq[i + 1][j] * para[j];
    para[i] = ( eq[i + 1][order] - temp ) / eq[i + 1][i];
  }
}

__device__ __forceinline__ int uni( int v ) { return __builtin_amdgcn_readfirstlane( v ); }      // a block-uniform value: keep it on the scalar side
__device__ __forceinline__ unsigned long long uni64( unsigned long long v )
{
  return ( ( unsigned long long ) ( unsigned ) __builtin_amdgcn_readfirstlane( ( int ) ( v >> 32 ) ) << 32 ) | ( unsigned ) __builtin_amdgcn_readfirstlane( ( int ) v );
}
__device__ __forceinline__ void uni_mv3( Mv3 &m ) { for( int i = 0; i < 3; i++ ) { m.v[i][0] = uni( m.v[i][0] ); m.v[i][1] = uni( m.v[i][1] ); } }

// One job = one xAffineMotionEstimation (:5340-5775).  The member evaluates a SEQUENCE of models -- the start model, one per gradient iteration, then the control-point
// refinement around the best -- and every evaluation is the same work: prediction (xPredAffineBlk), distortion, vector bits, strict comparison with the best so far.
// The kernel is therefore ONE loop with one evaluation site; what differs per step is how the next model is produced (a small block-uniform state machine).  That keeps the
// code a quarter of the size of the member written out call by call (the evaluation inlined at seven sites: 60 KB of code, more than the instruction cache two CUs share)
// and the search state in scalar registers.  SIX: the 6-parameter model (3 control points, 6 x 6 normal equations); the 4-parameter kernel carries 4 x 4.
// The normal equations are symmetric in their first NP columns ((int64) a * b commutes): the upper triangle is accumulated.
template<bool SIX, bool PACKED>      // PACKED: bitDepth <= 10 (the distortion runs on packed 16-bit words; deeper samples take the 32-bit Hadamard tiles, a kernel of its own)
__global__ __launch_bounds__( 256 ) __attribute__( ( amdgpu_waves_per_eu( PACKED ? ( SIX ? 2 : 4 ) : 1 ) ) ) void affine_me_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                          const int16_t *__restrict__ otherBase, const vtmhip_affine_me_job *__restrict__ jobs,
                                                          vtmhip_affine_me_out *__restrict__ results )
{
  constexpr int NP = SIX ? 6 : 4, NTRI = NP * ( NP + 1 ) / 2, NACC = NTRI + NP, MVNUM = SIX ? 3 : 2;
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sMem[];
  __shared__ unsigned long long sRed[4];
  __shared__ long long          sAcc[4][NACC];
  __shared__ Mv3                sMv;      // the model of the next gradient step (written by thread 0)
  __shared__ Mv3                sPrev[7]; // models of the earlier gradient steps (thread 0; the AMVR encoder option compares against them)
  __shared__ int                sCtl;     // loop control of the gradient iterations: 0 continue, 1 stop
  // XCD-aware job order: the hardware deals workgroups round-robin over the 8 XCDs; workgroup b takes the job at position xcd_order( b ), so that the jobs one XCD works on at
  // any moment are NEIGHBOURS in the table (PUs in raster order: overlapping reference windows) and share that XCD's L2 (the 16x16 level read 628 MB from HBM per launch without)
  const int jobIdx = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x );
  const vtmhip_affine_me_job &j = jobs[jobIdx];
  if( ( j.sixParam != 0 ) != SIX ) return;      // a mixed batch is launched once per model; every job belongs to exactly one of the two launches
  // A CU-level BCW weight of -2 makes the bi-pred target -4 org + 5 pred: differences up to 6138 leave the packed 16-bit Hadamard levels.  Such jobs belong to the 32-bit
  // variant (launched beside the packed one by vtmhip_xAffineMotionEstimation_bcw_batch_dev); every other job of a <= 10-bit picture to the packed variant.
  const int  bcw = ( j.bi && j.bcwWeight != 4 ) ? j.bcwWeight : 0;
  const bool wideJob = bcw < 0;
  if( PACKED ? wideJob : ( pic.bitDepth <= 10 && !wideJob ) ) return;
  const int w = j.width, h = j.height;
  int16_t  *sPat = sMem, *sPred = sMem + w * h;
  AffCtx c;
  c.ref = refBase + j.refOff; c.refStride = j.refStride; c.w = w; c.h = h; c.bd = pic.bitDepth; c.six = SIX; c.interDir = j.interDir; c.imv = j.imv;
  c.horMax = ( pic.picW + 8 - j.puX - 1 ) << 4; c.horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  c.verMax = ( pic.picH + 8 - j.puY - 1 ) << 4; c.verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  c.profAllowed = j.profAllowed != 0; c.profLarge = j.profNeedsLargeGrad != 0; c.profIsBi = j.profIsBi != 0;
  const bool   bi = j.bi != 0, satd = j.useSatd != 0;
  const double fWeight = bi ? ( bcw ? fabs( ( double ) bcw / 8.0 ) : 0.5 ) : 1.0, lam = j.motionLambda;      // xGetMEDistortionWeight (InterSearch.cpp:7666-7676)
  const int    imv = j.imv, rs = imv == 0 ? 2 : imv == 1 ? 0 : 4;      // (rsTab of the reference: MV_PRECISION of the AMVR mode)
  // pattern: org, or 2*org - otherPred (removeHighFreq, unclipped), or the weighted form ( org * w0 - otherPred * w1 + 2^15 ) >> 16 under a CU-level BCW weight (Buffer.h:417-460)
  {
    const int16_t *o = orgBase + j.orgOff, *p = bi ? otherBase + j.otherPredOff : nullptr;
    const int      nrm = bcw ? ( ( 1 << 16 ) + ( bcw > 0 ? ( bcw >> 1 ) : -( bcw >> 1 ) ) ) / bcw : 0, bw0 = nrm << 3, bw1 = ( 8 - bcw ) * nrm;
    for( int i = threadIdx.x; i < w * h; i += blockDim.x )
    {
      const int y = i / w, x = i - y * w;
      const int v = o[( long ) y * j.orgStride + x];
      sPat[i] = ( int16_t ) ( !bi ? v : bcw ? ( v * bw0 - ( int ) p[( long ) y * j.otherPredStride + x] * bw1 + ( 1 << 15 ) ) >> 16 : 2 * v - p[( long ) y * j.otherPredStride + x] );
    }
  }
  int pred[3][2];
  for( int i = 0; i < 3; i++ ) { pred[i][0] = j.mvPred[i][0]; pred[i][1] = j.mvPred[i][1]; }

  Mv3 tmp, best, cand, me, base;
  int center[2] = { 0, 0 }, dMv[2] = { 0, 0 };
  for( int i = 0; i < 3; i++ ) { tmp.v[i][0] = j.mv[i][0]; tmp.v[i][1] = j.mv[i][1]; }
  for( int i = 0; i < MVNUM; i++ )
  {
    tmp.v[i][0] = clip3( c.horMin, c.horMax, tmp.v[i][0] ); tmp.v[i][1] = clip3( c.verMin, c.verMax, tmp.v[i][1] );
    tmp.v[i][0] = prec_dn( tmp.v[i][0], rs ) << rs; tmp.v[i][1] = prec_dn( tmp.v[i][1], rs ) << rs;   // roundAffinePrecInternal2Amvr
  }
  best = me = base = cand = tmp;
  unsigned long long costBest = ~0ull;
  unsigned           bitsBest = 0;
  int iterTime = SIX ? ( bi ? 3 : 4 ) : ( bi ? 3 : 5 );
  if( !j.useAffineType ) iterTime = bi ? 5 : 7;
  const int maxRound = imv ? 3 : ( ( j.amvrEncOpt && j.lowDelayRounds ) ? 2 : 3 );
  int iterations = 0, refinements = 0;
  // cu.imv == 2 with AffineAmvrEncOpt: xDetermineBestMvp over the job's AMVP list at the start model and at every gradient evaluation (InterSearch.cpp:5444-5449, 5629-5634)
  const bool     pickMvp = imv == 2 && j.amvrEncOpt && j.numAmvpCand > 0;
  const unsigned dirBits = pickMvp ? j.bits - j.mvpIdxBits[j.mvpIdx & 1] : 0u;      // (:5359)
  int            mvpIdx = j.mvpIdx, bestMvpIdx = j.mvpIdx;
  enum { P_INIT, P_ITER, P_REF_START, P_C1, P_C2, P_C3, P_RND, P_DONE };
  int  phase = P_INIT, iter = 0, k = 0, it = 0, pos = 0, rnd = 0;
  bool modelChange = false, loopChange = false;
  __syncthreads();

#pragma unroll 1
  while( phase != P_DONE )
  {
    bool have = false;
    const int cur = phase;
    if( phase == P_INIT ) { cand = tmp; have = true; }
    else if( phase == P_ITER )
    {
      if( iter >= iterTime ) { phase = P_REF_START; continue; }
      // ---- normal equations from the error and the Sobel gradients of the current prediction (exact 64-bit sums) ----
      long long acc[NACC];
#pragma unroll
      for( int i = 0; i < NACC; i++ ) acc[i] = 0;
      for( int i = threadIdx.x; i < w * h; i += blockDim.x )
      {
        const int y = i / w, x = i - y * w;
        const int yc = min( h - 2, max( 1, y ) ), xc = min( w - 2, max( 1, x ) );
        const int16_t *q = sPred + yc * w + xc;
        const int a = q[1 - w] - q[-1 - w] + ( q[1] << 1 ) - ( q[-1] << 1 ) + q[1 + w] - q[-1 + w];
        const int b = q[w - 1] - q[-w - 1] + ( q[w] << 1 ) - ( q[-w] << 1 ) + q[w + 1] - q[-w + 1];
        const int e = ( int ) ( int16_t ) ( sPat[i] - sPred[i] );
        const int cy = ( ( y >> 2 ) << 2 ) + 2, cx = ( ( x >> 2 ) << 2 ) + 2;
        int       cc[NP];
        if( !SIX ) { cc[0] = a; cc[1] = cx * a + cy * b; cc[2] = b; cc[3] = cy * a - cx * b; }
        else { cc[0] = a; cc[1] = cx * a; cc[2] = b; cc[3] = cx * b; cc[4] = cy * a; cc[5] = cy * b; }
        int t = 0;
#pragma unroll
        for( int col = 0; col < NP; col++ )
        {
#pragma unroll
          for( int row = col; row < NP; row++ ) acc[t++] += ( long long ) cc[col] * cc[row];
          acc[NTRI + col] += ( ( long long ) cc[col] * e ) << 3;
        }
      }
#pragma unroll
      for( int i = 0; i < NACC; i++ ) acc[i] = ( long long ) wave_reduce_add_u64( ( unsigned long long ) acc[i] );
      if( ( threadIdx.x & 63 ) == 0 )
      {
#pragma unroll
        for( int i = 0; i < NACC; i++ ) sAcc[threadIdx.x >> 6][i] = acc[i];
      }
      __syncthreads();
      if( threadIdx.x == 0 )
      {
        sPrev[iter] = tmp;
        double deq[NP + 1][NP + 1];
#pragma unroll
        for( int r = 0; r <= NP; r++ )
#pragma unroll
          for( int q = 0; q <= NP; q++ ) deq[r][q] = 0.;
        {
          int t = 0;
#pragma unroll
          for( int col = 0; col < NP; col++ )
          {
#pragma unroll
            for( int row = col; row < NP; row++ )
            {
              long long v = sAcc[0][t];
              for( int wv = 1; wv < ( int ) ( blockDim.x >> 6 ); wv++ ) v += sAcc[wv][t];
              deq[col + 1][row] = ( double ) v; deq[row + 1][col] = ( double ) v;
              t++;
            }
            long long v = sAcc[0][NTRI + col];
            for( int wv = 1; wv < ( int ) ( blockDim.x >> 6 ); wv++ ) v += sAcc[wv][NTRI + col];
            deq[col + 1][NP] = ( double ) v;
          }
        }
        double para[NP], dmv[6] = { 0, 0, 0, 0, 0, 0 };
        solve_equal<NP>( deq, para );
        dmv[0] = para[0]; dmv[2] = para[2];
        if( SIX ) { dmv[1] = para[1] * w + para[0]; dmv[3] = para[3] * w + para[2]; dmv[4] = para[NP - 2] * h + para[0]; dmv[5] = para[NP - 1] * h + para[2]; }
        else { dmv[1] = para[1] * w + para[0]; dmv[3] = -para[3] * w + para[2]; }
        const int mult = 1 << ( imv == 1 ? 4 : 2 ), ms = imv == 1 ? 0 : 2;      // normShift / stepShift of the reference
#define SGN( x ) ( ( x ) >= 0 ? 1 : -1 )
        int delta[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
        delta[0][0] = ( int ) ( dmv[0] * mult + SGN( dmv[0] ) * 0.5 ) << ms; delta[0][1] = ( int ) ( dmv[2] * mult + SGN( dmv[2] ) * 0.5 ) << ms;
        delta[1][0] = ( int ) ( dmv[1] * mult + SGN( dmv[1] ) * 0.5 ) << ms; delta[1][1] = ( int ) ( dmv[3] * mult + SGN( dmv[3] ) * 0.5 ) << ms;
        if( SIX ) { delta[2][0] = ( int ) ( dmv[4] * mult + SGN( dmv[4] ) * 0.5 ) << ms; delta[2][1] = ( int ) ( dmv[5] * mult + SGN( dmv[5] ) * 0.5 ) << ms; }
#undef SGN
        int stop = 0;
        if( !j.amvrEncOpt )
        {
          bool allZero = false;
          for( int i = 0; i < MVNUM; i++ )
          {
            int d0 = delta[i][0], d1 = delta[i][1];
            if( imv == 2 ) { d0 = prec_dn( d0, 3 ) << 3; d1 = prec_dn( d1, 3 ) << 3; }
            if( d0 != 0 || d1 != 0 ) { allZero = false; break; }
            allZero = true;
          }
          if( allZero ) stop = 1;
        }
        Mv3 nt = tmp;
        if( !stop )
        {
          for( int i = 0; i < MVNUM; i++ )
          {
            nt.v[i][0] = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, nt.v[i][0] + delta[i][0] );
            nt.v[i][1] = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, nt.v[i][1] + delta[i][1] );
            nt.v[i][0] = prec_dn( nt.v[i][0], rs ) << rs; nt.v[i][1] = prec_dn( nt.v[i][1], rs ) << rs;
            nt.v[i][0] = clip3( c.horMin, c.horMax, nt.v[i][0] ); nt.v[i][1] = clip3( c.verMin, c.verMax, nt.v[i][1] );
          }
          if( j.amvrEncOpt )
            for( int q = iter; q >= 0; q-- )
              if( nt.v[0][0] == sPrev[q].v[0][0] && nt.v[0][1] == sPrev[q].v[0][1] && nt.v[1][0] == sPrev[q].v[1][0] && nt.v[1][1] == sPrev[q].v[1][1] )
              {
                const bool same = SIX ? ( nt.v[2][0] == sPrev[q].v[2][0] && nt.v[2][1] == sPrev[q].v[2][1] ) : true;
                if( same ) { stop = 1; break; }
              }
        }
        sMv = nt; sCtl = stop;
      }
      __syncthreads();
      const int stop = uni( sCtl );
      tmp = sMv;
      uni_mv3( tmp );
      __syncthreads();
      iter++;
      if( stop ) { phase = P_REF_START; continue; }
      cand = tmp; have = true;
    }
    else if( phase == P_REF_START )
    {
      if( !( ( double ) costBest <= 1.0 * ( double ) j.hevcCost ) ) break;      // AFFINE_ME_LIST_MVP_TH * m_hevcCost
      me = best; dMv[0] = me.v[0][0] - pred[0][0]; dMv[1] = me.v[0][1] - pred[0][1];
      k = 0; phase = P_C1;
      continue;
    }
    else if( phase == P_C1 )      // each control point at its predictor (:5655-5671)
    {
      if( k >= MVNUM ) { phase = P_C2; continue; }
      const int kk = k++;
      // (control-point arrays are indexed through constant loops: a run-time index would move the whole search state to scratch memory)
      int ph = 0, pv = 0, mh = 0, mv = 0;
#pragma unroll
      for( int i = 0; i < MVNUM; i++ )
        if( i == kk ) { ph = pred[i][0] + ( i ? dMv[0] : 0 ); pv = pred[i][1] + ( i ? dMv[1] : 0 ); mh = me.v[i][0]; mv = me.v[i][1]; }
      if( mh != ph || mv != pv )
      {
        cand = me;
#pragma unroll
        for( int i = 0; i < MVNUM; i++ )
          if( i == kk ) { cand.v[i][0] = ph; cand.v[i][1] = pv; }
        have = true;
      }
    }
    else if( phase == P_C2 )
    {
      phase = P_C3;
      if( me.v[0][0] != pred[0][0] || me.v[0][1] != pred[0][1] )
      {
        cand = me;
        for( int i = 1; i < MVNUM; i++ ) { cand.v[i][0] -= dMv[0]; cand.v[i][1] -= dMv[1]; }
        cand.v[0][0] = pred[0][0]; cand.v[0][1] = pred[0][1];
        have = true;
      }
    }
    else if( phase == P_C3 )
    {
      phase = P_RND; rnd = 0; k = 0; it = 0; pos = 0; modelChange = false; loopChange = false;
      if( SIX && ( me.v[1][0] != pred[1][0] + dMv[0] || me.v[1][1] != pred[1][1] + dMv[1] ) && ( me.v[2][0] != pred[2][0] + dMv[0] || me.v[2][1] != pred[2][1] + dMv[1] ) )
      {
        cand = me;
        cand.v[1][0] = pred[1][0] + dMv[0]; cand.v[1][1] = pred[1][1] + dMv[1]; cand.v[2][0] = pred[2][0] + dMv[0]; cand.v[2][1] = pred[2][1] + dMv[1];
        have = true;
      }
    }
    else      // P_RND: rounds x control points x { the four direct, then (if one of them won) the four diagonal neighbours } (:5696-5765)
    {
      if( pos == 0 )
      {
        base = best;
#pragma unroll
        for( int i = 0; i < MVNUM; i++ )
          if( i == k ) { center[0] = best.v[i][0]; center[1] = best.v[i][1]; }
      }
      const int idx = it * 4 + pos;
      const int tx = idx == 0 || idx == 4 || idx == 5 ? -1 : idx == 3 || idx == 6 || idx == 7 ? 1 : 0;           // testPos { -1,0 } { 0,-1 } { 0,1 } { 1,0 } { -1,-1 } { -1,1 } { 1,1 } { 1,-1 }
      const int ty = idx == 1 || idx == 4 || idx == 7 ? -1 : idx == 2 || idx == 5 || idx == 6 ? 1 : 0;
      cand = base;
#pragma unroll
      for( int i = 0; i < MVNUM; i++ )
        if( i == k ) { cand.v[i][0] = clip3( c.horMin, c.horMax, center[0] + ( tx << rs ) ); cand.v[i][1] = clip3( c.verMin, c.verMax, center[1] + ( ty << rs ) ); }
      have = true;
    }
    if( !have ) continue;

    // ---- the one evaluation site: prediction, distortion, bits, comparison ----
    affine_pred( c, cand, sPred );
    __syncthreads();
    unsigned long long cost = uni64( block_dist<PACKED>( sPred, sPat, w, h, satd, sRed ) );
    unsigned           bits;
    const bool         picking = pickMvp && ( cur == P_INIT || cur == P_ITER );
    if( picking )
    {
      // xDetermineBestMvp (:7766-7785): the candidate with the fewest bits for this model, first minimum; acMvPred follows it whether or not the model wins
      unsigned minBits = ~0u;
      for( int ci = 0; ci < j.numAmvpCand && ci < 2; ci++ )
      {
        int pc[3][2];
#pragma unroll
        for( int v = 0; v < 3; v++ ) { pc[v][0] = ci ? j.amvpCand[1][v][0] : j.amvpCand[0][v][0]; pc[v][1] = ci ? j.amvpCand[1][v][1] : j.amvpCand[0][v][1]; }
        const unsigned cb = ( ci ? j.mvpIdxBits[1] : j.mvpIdxBits[0] ) + affine_mv_bits( SIX, imv, cand, pc );
        if( cb < minBits ) { minBits = cb; bestMvpIdx = ci; }
      }
#pragma unroll
      for( int v = 0; v < 3; v++ ) { pred[v][0] = bestMvpIdx ? j.amvpCand[1][v][0] : j.amvpCand[0][v][0]; pred[v][1] = bestMvpIdx ? j.amvpCand[1][v][1] : j.amvpCand[0][v][1]; }
      bits = dirBits + minBits;
    }
    else bits = j.bits + affine_mv_bits( SIX, imv, cand, pred );
    cost = ( unsigned long long ) ( floor( fWeight * ( double ) cost ) + ( double ) ( unsigned long long ) ( lam * bits ) );
    const bool better = cur == P_INIT || cost < costBest;
    if( better ) { costBest = cost; bitsBest = bits; best = cand; if( picking ) mvpIdx = bestMvpIdx; }
    if( cur == P_INIT ) phase = P_ITER;
    else if( cur == P_ITER ) iterations++;
    else
    {
      refinements++;
      if( cur == P_RND )
      {
        if( better ) { modelChange = true; loopChange = true; }
        if( ++pos == 4 )
        {
          pos = 0;
          if( it == 0 && loopChange ) it = 1;
          else
          {
            it = 0; loopChange = false;
            if( ++k == MVNUM )
            {
              k = 0;
              if( !modelChange || ++rnd == maxRound ) phase = P_DONE;
              modelChange = false;
            }
          }
        }
      }
    }
  }
  if( threadIdx.x == 0 )
  {
    vtmhip_affine_me_out o;
    for( int i = 0; i < 3; i++ ) { o.mv[i][0] = best.v[i][0]; o.mv[i][1] = best.v[i][1]; }
    o.bits = bitsBest; o.cost = costBest; o.iterations = iterations; o.refinements = refinements; o.mvpIdx = mvpIdx;
    results[jobIdx] = o;
  }
}

__global__ __launch_bounds__( 256 ) void affine_pred_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ refBase, int16_t *__restrict__ dstBase,
                                                            const vtmhip_affine_me_job *__restrict__ jobs )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sMem[];
  const vtmhip_affine_me_job &j = jobs[blockIdx.x];
  AffCtx c;
  c.ref = refBase + j.refOff; c.refStride = j.refStride; c.w = j.width; c.h = j.height; c.bd = pic.bitDepth; c.six = j.sixParam; c.interDir = j.interDir; c.imv = j.imv;
  c.horMax = ( pic.picW + 8 - j.puX - 1 ) << 4; c.horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  c.verMax = ( pic.picH + 8 - j.puY - 1 ) << 4; c.verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  c.profAllowed = j.profAllowed != 0; c.profLarge = j.profNeedsLargeGrad != 0; c.profIsBi = j.profIsBi != 0;
  Mv3 m;
  for( int i = 0; i < 3; i++ ) { m.v[i][0] = j.mv[i][0]; m.v[i][1] = j.mv[i][1]; }
  affine_pred( c, m, sMem );
  __syncthreads();
  int16_t *d = dstBase + j.predOff;
  for( int i = threadIdx.x; i < c.w * c.h; i += 256 ) d[( long ) ( i / c.w ) * j.predStride + ( i % c.w )] = sMem[i];
}

}   // namespace

extern "C"
{

static int check_affine_args( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, int n, int maxWidth, int maxHeight )
{
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "pic / n" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 16 && maxWidth <= 128 && maxHeight >= 16 && maxHeight <= 128, "affine blocks are 16..128 wide and high" );
  VTMHIP_REQUIRE( ctx, pic->bitDepth >= 8 && pic->bitDepth <= 12, "bit depth" );
  return VTMHIP_OK;
}

}   // extern "C"

// models: 0 = the caller knows every job is 4-parameter, 1 = every job 6-parameter, -1 = unknown / mixed (both launches; a job runs in the launch of its own model)
int vtmhip_internal_affine_me_launch( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const int16_t *d_otherPredBase,
                                      const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_affine_me_out *d_results, int models )
{
  VTMHIP_CHECK_CTX( ctx );
  int st = check_affine_args( ctx, pic, n, maxWidth, maxHeight );
  if( st ) return st;
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && d_jobs && d_results, "null pointer" );
  const size_t lds = 2 * ( size_t ) maxWidth * maxHeight * sizeof( int16_t );
  const bool packed = pic->bitDepth <= 10;
  const void *k4 = packed ? reinterpret_cast<const void *>( affine_me_kernel<false, true> ) : reinterpret_cast<const void *>( affine_me_kernel<false, false> );
  const void *k6 = packed ? reinterpret_cast<const void *>( affine_me_kernel<true, true> ) : reinterpret_cast<const void *>( affine_me_kernel<true, false> );
  if( lds > 48 * 1024 )
  {
    VTMHIP_HIP( ctx, hipFuncSetAttribute( k4, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
    VTMHIP_HIP( ctx, hipFuncSetAttribute( k6, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
  }
  VTMHIP_TIME_KERNEL( ctx, "affine_me_kernel" );
  // one wave per job up to 32x32 (64 4x4 sub-blocks: a lane each), four waves above: the model iterations are a serial chain per job, so small blocks gain
  // from four times as many jobs in flight, not from idle lanes
  const int threads = maxWidth * maxHeight <= 1024 ? 64 : 256;
  const int sel = models < 0 ? -1 : ( models & 3 );      // 0: 4-parameter jobs only, 1: 6-parameter only, -1 / 2: both
  const bool wideToo = models >= 0 && ( models & VTMHIP_AFFINE_LAUNCH_WIDE ) != 0;      // the batch may hold BCW weight -2 jobs: the 32-bit variants run beside the packed ones
  const int16_t *oth = d_otherPredBase ? d_otherPredBase : d_orgBase;
  if( packed )
  {
    if( sel != 1 ) hipLaunchKernelGGL( ( affine_me_kernel<false, true> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
    if( sel != 0 ) hipLaunchKernelGGL( ( affine_me_kernel<true, true> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
    if( wideToo )
    {
      if( lds > 48 * 1024 )
      {
        VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( affine_me_kernel<false, false> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
        VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( affine_me_kernel<true, false> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
      }
      if( sel != 1 ) hipLaunchKernelGGL( ( affine_me_kernel<false, false> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
      if( sel != 0 ) hipLaunchKernelGGL( ( affine_me_kernel<true, false> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
    }
  }
  else
  {
    if( sel != 1 ) hipLaunchKernelGGL( ( affine_me_kernel<false, false> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
    if( sel != 0 ) hipLaunchKernelGGL( ( affine_me_kernel<true, false> ), dim3( n ), dim3( threads ), lds, ctx->stream, *pic, d_orgBase, d_refBase, oth, d_jobs, d_results );
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

extern "C"
{

int vtmhip_xAffineMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                              const int16_t *d_otherPredBase, const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight,
                                              vtmhip_affine_me_out *d_results )
{
  return vtmhip_internal_affine_me_launch( ctx, pic, d_orgBase, d_refBase, d_otherPredBase, d_jobs, n, maxWidth, maxHeight, d_results, -1 );
}

int vtmhip_xAffineMotionEstimation_bcw_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                  const int16_t *d_otherPredBase, const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight,
                                                  vtmhip_affine_me_out *d_results )
{
  return vtmhip_internal_affine_me_launch( ctx, pic, d_orgBase, d_refBase, d_otherPredBase, d_jobs, n, maxWidth, maxHeight, d_results, 2 | VTMHIP_AFFINE_LAUNCH_WIDE );
}

int vtmhip_xPredAffineBlk_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_affine_me_job *d_jobs, int n,
                                     int maxWidth, int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  int st = check_affine_args( ctx, pic, n, maxWidth, maxHeight );
  if( st ) return st;
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_refBase && d_dstBase && d_jobs, "null pointer" );
  const size_t lds = ( size_t ) maxWidth * maxHeight * sizeof( int16_t );
  hipLaunchKernelGGL( affine_pred_kernel, dim3( n ), dim3( 256 ), lds, ctx->stream, *pic, d_refBase, d_dstBase, d_jobs );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"
