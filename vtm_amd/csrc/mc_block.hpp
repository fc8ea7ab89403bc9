// mc_block.hpp -- one block of InterPrediction::xPredInterBlk (CommonLib/InterPrediction.cpp:660-815) as device functions: the luma 8-tap / 4:2:0 chroma
// 4-tap separable filter of a W x H block with a caller-supplied sink for the output samples.  Shared by mc.hip (motion compensation, BDOF, DMVR)
// and smvd.hip (the symmetric-MVD search predicts two blocks per candidate).
#pragma once
#include "ctx.hpp"

namespace
{

__constant__ int16_t c_lumaFilterMc[16][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },       { 0, 1, -3, 63, 4, -2, 1, 0 },     { -1, 2, -5, 62, 8, -3, 1, 0 },    { -1, 3, -8, 60, 13, -4, 1, 0 },
  { -1, 4, -10, 58, 17, -5, 1, 0 },  { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
  { -1, 4, -11, 40, 40, -11, 4, -1 },{ -1, 4, -10, 34, 45, -11, 4, -1 },{ -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
  { 0, 1, -5, 17, 58, -10, 4, -1 },  { 0, 1, -4, 13, 60, -8, 3, -1 },   { 0, 1, -3, 8, 62, -5, 2, -1 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };
__constant__ int16_t c_lumaFilter4x4Mc[16][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },      { 0, 1, -3, 63, 4, -2, 1, 0 },    { 0, 1, -5, 62, 8, -3, 1, 0 },    { 0, 2, -8, 60, 13, -4, 1, 0 },
  { 0, 3, -10, 58, 17, -5, 1, 0 },  { 0, 3, -11, 52, 26, -8, 2, 0 },  { 0, 2, -9, 47, 31, -10, 3, 0 },  { 0, 3, -11, 45, 34, -10, 3, 0 },
  { 0, 3, -11, 40, 40, -11, 3, 0 }, { 0, 3, -10, 34, 45, -11, 3, 0 }, { 0, 3, -10, 31, 47, -9, 2, 0 },  { 0, 2, -8, 26, 52, -11, 3, 0 },
  { 0, 1, -5, 17, 58, -10, 3, 0 },  { 0, 1, -4, 13, 60, -8, 2, 0 },   { 0, 1, -3, 8, 62, -5, 1, 0 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };
__constant__ int16_t c_altHpelMc[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };

struct Fir { int shift, offset, clip, cmax; };

// p / d for 0 <= p with p * d < 2^32: multiply-high by floor(2^32 / d) and one correction step (4 instructions) instead of the generic ~25-instruction
// integer division sequence; the constructor's division is wave-uniform, once per block
struct FastDiv
{
  unsigned magic; int d;
  __device__ __forceinline__ explicit FastDiv( int dd ) : magic( 0xffffffffu / ( unsigned ) dd ), d( dd ) {}
  __device__ __forceinline__ int operator()( int p ) const
  {
    const int q = ( int ) __umulhi( ( unsigned ) p, magic );
    return q + ( ( q + 1 ) * d <= p ? 1 : 0 );
  }
};

// InterpolationFilter::filter shift/offset rules (:577-602); integer phases use taps {0,0,0,64,0,0,0,0}, which is
// arithmetically identical to filterCopy (:398-525) for every (isFirst, isLast) pair that occurs here except
// (first && last), handled as a plain copy.
__device__ __forceinline__ Fir fir_params( int isFirst, int isLast, int bd )
{
  Fir       f;
  const int headRoom = max( 2, 14 - bd );
  int       shift    = 6, offset;
  if( isLast ) { shift += isFirst ? 0 : headRoom; offset = ( 1 << ( shift - 1 ) ) + ( isFirst ? 0 : ( 8192 << 6 ) ); }
  else { shift -= isFirst ? headRoom : 0; offset = isFirst ? -( 8192 << shift ) : 0; }
  f.shift = shift; f.offset = offset; f.clip = isLast; f.cmax = ( 1 << bd ) - 1;
  return f;
}
__device__ __forceinline__ int16_t fir_out( int sum, const Fir &f )
{
  int16_t v = ( int16_t ) ( ( sum + f.offset ) >> f.shift );
  if( f.clip ) v = ( int16_t ) min( f.cmax, max( 0, ( int ) v ) );
  return v;
}

__device__ __forceinline__ const int16_t *luma_taps( int frac, int w, int h, int hForRule, bool altHpel )
{
  if( frac == 8 && altHpel ) return c_altHpelMc;
  if( w == 4 && ( hForRule == 4 ) ) return c_lumaFilter4x4Mc[frac];   // InterpolationFilter.cpp:786-789, 869-872
  return c_lumaFilterMc[frac];
}

// H.266 table 28 (chroma 4-tap filter, 1/32 sample phases), InterpolationFilter.cpp:132-166
__constant__ int16_t c_chromaFilterMc[32][4] = {
  { 0, 64, 0, 0 },    { -1, 63, 2, 0 },   { -2, 62, 4, 0 },   { -2, 60, 7, -1 },  { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 },
  { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 },
  { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
  { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 }, { -2, 10, 58, -2 }, { -1, 7, 60, -2 },  { 0, 4, 62, -2 },   { 0, 2, 63, -1 } };

// One block, one wave.  NT = 8: luma (phase = 4 fraction bits); NT = 4: a 4:2:0 chroma plane (the vector stays in luma 1/16 units, so
// the phase has 5 bits; InterPrediction.cpp:675-676).  lds holds the (h + NT - 1) x w horizontal-pass intermediates.
// `out( y, x, v )` receives every output sample (store to HBM, keep in LDS, or a fused epilogue).
template<int THREADS>
__device__ __forceinline__ void block_sync()
{
  if( THREADS <= 64 )
  {
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
  }
  else __syncthreads();
}

template<int NT, int THREADS, class Out>
__device__ __forceinline__ void mc_block( const vtmhip_mc_job &j, const int16_t *__restrict__ refBase, int16_t *lds, int lane, Out out )
{
  constexpr int FB = NT == 8 ? 4 : 5, HALO = NT / 2 - 1;
  const int     w = j.width, h = j.height, bd = j.bitDepth;
  const FastDiv divW( w );
  const int     xFrac = j.mvHor & ( ( 1 << FB ) - 1 ), yFrac = j.mvVer & ( ( 1 << FB ) - 1 ), rnd = !j.bi;
  const bool    alt = j.useAltHpelIf != 0;
  const int16_t *src = refBase + j.refOff + ( long ) ( j.mvVer >> FB ) * j.refStride + ( j.mvHor >> FB );
  if( yFrac == 0 )
  {
    if( xFrac == 0 && rnd )   // filterCopy<true,true>: plain copy
    {
      for( int i = lane; i < w * h; i += THREADS ) { const int y = divW( i ), x = i - y * w; out( y, x, src[( long ) y * j.refStride + x] ); }
      return;
    }
    const Fir      f = fir_params( 1, rnd, bd );
    const int16_t *c = NT == 8 ? luma_taps( xFrac, w, h, h, alt ) : c_chromaFilterMc[xFrac];
    for( int i = lane; i < w * h; i += THREADS )
    {
      const int y = divW( i ), x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < NT; k++ ) sum += ( int ) src[( long ) y * j.refStride + x + k - HALO] * ( int ) c[k];
      out( y, x, fir_out( sum, f ) );
    }
  }
  else if( xFrac == 0 )
  {
    const Fir      f = fir_params( 1, rnd, bd );
    const int16_t *c = NT == 8 ? luma_taps( yFrac, w, h, h, alt ) : c_chromaFilterMc[yFrac];
    for( int i = lane; i < w * h; i += THREADS )
    {
      const int y = divW( i ), x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < NT; k++ ) sum += ( int ) src[( long ) ( y + k - HALO ) * j.refStride + x] * ( int ) c[k];
      out( y, x, fir_out( sum, f ) );
    }
  }
  else
  {
    const Fir      fh = fir_params( 1, 0, bd ), fv = fir_params( 0, rnd, bd );
    const int16_t *ch = NT == 8 ? luma_taps( xFrac, w, h, h + 7 == 11 ? 4 : -1, alt ) : c_chromaFilterMc[xFrac];   // luma H pass sees W x (H+7): 4 x 11 takes the 4x4 taps
    const int16_t *cv = NT == 8 ? luma_taps( yFrac, w, h, h, alt ) : c_chromaFilterMc[yFrac];
    for( int i = lane; i < w * ( h + NT - 1 ); i += THREADS )
    {
      const int r = divW( i ), x = i - r * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < NT; k++ ) sum += ( int ) src[( long ) ( r - HALO ) * j.refStride + x + k - HALO] * ( int ) ch[k];
      lds[i] = fir_out( sum, fh );
    }
    block_sync<THREADS>();
    for( int i = lane; i < w * h; i += THREADS )
    {
      const int y = divW( i ), x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < NT; k++ ) sum += ( int ) lds[( y + k ) * w + x] * ( int ) cv[k];
      out( y, x, fir_out( sum, fv ) );
    }
  }
}


// ---- 8 outputs per lane (blocks whose width is a multiple of 8): 16-byte loads / stores, each input sample fetched once per lane --------
struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8u { unsigned v[4]; };   // 8 samples at a 2-byte aligned address (global memory)

__device__ __forceinline__ void unpack8( const unsigned u[4], int a[8] )
{
#pragma unroll
  for( int k = 0; k < 4; k++ ) { a[2 * k] = ( int ) ( short ) ( u[k] & 0xffffu ); a[2 * k + 1] = ( int ) u[k] >> 16; }
}
__device__ __forceinline__ uint4 pack8( const int v[8] )
{
  uint4 u;
  u.x = ( ( unsigned ) v[0] & 0xffffu ) | ( ( unsigned ) v[1] << 16 ); u.y = ( ( unsigned ) v[2] & 0xffffu ) | ( ( unsigned ) v[3] << 16 );
  u.z = ( ( unsigned ) v[4] & 0xffffu ) | ( ( unsigned ) v[5] << 16 ); u.w = ( ( unsigned ) v[6] & 0xffffu ) | ( ( unsigned ) v[7] << 16 );
  return u;
}
__device__ __forceinline__ void load8g( const int16_t *p, int a[8] )    // global, 2-byte aligned
{
  const Pel8u t = *reinterpret_cast<const Pel8u *>( p );
  unpack8( t.v, a );
}
__device__ __forceinline__ void store8g( int16_t *p, const int v[8] )
{
  const uint4 u = pack8( v );
  Pel8u       t;
  t.v[0] = u.x; t.v[1] = u.y; t.v[2] = u.z; t.v[3] = u.w;
  *reinterpret_cast<Pel8u *>( p ) = t;
}
__device__ __forceinline__ void load8s( const int16_t *p, int a[8] )    // LDS, 16-byte aligned
{
  const uint4    u = *reinterpret_cast<const uint4 *>( p );
  const unsigned t[4] = { u.x, u.y, u.z, u.w };
  unpack8( t, a );
}
__device__ __forceinline__ void load16( const int16_t *p, int a[16] )
{
  load8g( p, a );
  load8g( p + 8, a + 8 );
}

// `out.vec( y, x0, v )` receives 8 horizontally adjacent output samples.
template<int NT, int THREADS, class Out>
__device__ __forceinline__ void mc_block_vec( const vtmhip_mc_job &j, const int16_t *__restrict__ refBase, int16_t *lds, int lane, Out out )
{
  constexpr int FB = NT == 8 ? 4 : 5, HALO = NT / 2 - 1;
  const int     w = j.width, h = j.height, bd = j.bitDepth, segs = w >> 3;
  const FastDiv divS( segs );
  const int     xFrac = j.mvHor & ( ( 1 << FB ) - 1 ), yFrac = j.mvVer & ( ( 1 << FB ) - 1 ), rnd = !j.bi;
  const bool    alt = j.useAltHpelIf != 0;
  const int16_t *src = refBase + j.refOff + ( long ) ( j.mvVer >> FB ) * j.refStride + ( j.mvHor >> FB );
  if( yFrac == 0 && xFrac == 0 && rnd )   // filterCopy<true,true>: plain copy
  {
    for( int i = lane; i < segs * h; i += THREADS )
    {
      const int   y = divS( i ), x0 = ( i - y * segs ) << 3;
      int a[8];
      load8g( src + ( long ) y * j.refStride + x0, a );
      out.vec( y, x0, a );
    }
    return;
  }
  if( yFrac == 0 )   // horizontal only (phase 0 of a bi prediction runs the {0,0,0,64,..} taps = filterCopy<first,!last> arithmetic)
  {
    const Fir      f = fir_params( 1, rnd, bd );
    const int16_t *c = NT == 8 ? luma_taps( xFrac, w, h, h, alt ) : c_chromaFilterMc[xFrac];
    int            cc[NT];
#pragma unroll
    for( int t = 0; t < NT; t++ ) cc[t] = c[t];
    for( int i = lane; i < segs * h; i += THREADS )
    {
      const int y = divS( i ), x0 = ( i - y * segs ) << 3;
      int       a[16];
      load16( src + ( long ) y * j.refStride + x0 - HALO, a );
      int v[8];
#pragma unroll
      for( int k = 0; k < 8; k++ )
      {
        int sum = 0;
#pragma unroll
        for( int t = 0; t < NT; t++ ) sum += a[k + t] * cc[t];
        v[k] = fir_out( sum, f );
      }
      out.vec( y, x0, v );
    }
    return;
  }
  const bool     twoPass = xFrac != 0;
  const int16_t *cvp = NT == 8 ? luma_taps( yFrac, w, h, h, alt ) : c_chromaFilterMc[yFrac];
  int            cv[NT];
#pragma unroll
  for( int t = 0; t < NT; t++ ) cv[t] = cvp[t];
  if( twoPass )
  {
    const Fir      fh = fir_params( 1, 0, bd );
    const int16_t *chp = NT == 8 ? luma_taps( xFrac, w, h, h + 7 == 11 ? 4 : -1, alt ) : c_chromaFilterMc[xFrac];
    int            ch[NT];
#pragma unroll
    for( int t = 0; t < NT; t++ ) ch[t] = chp[t];
    for( int i = lane; i < segs * ( h + NT - 1 ); i += THREADS )
    {
      const int r = divS( i ), x0 = ( i - r * segs ) << 3;
      int       a[16];
      load16( src + ( long ) ( r - HALO ) * j.refStride + x0 - HALO, a );
      int v[8];
#pragma unroll
      for( int k = 0; k < 8; k++ )
      {
        int sum = 0;
#pragma unroll
        for( int t = 0; t < NT; t++ ) sum += a[k + t] * ch[t];
        v[k] = fir_out( sum, fh );
      }
      *reinterpret_cast<uint4 *>( lds + r * w + x0 ) = pack8( v );   // rows of w = 8n samples: 16-byte aligned
    }
    block_sync<THREADS>();
  }
  const Fir fv = fir_params( twoPass ? 0 : 1, rnd, bd );
  for( int i = lane; i < segs * h; i += THREADS )
  {
    const int y = divS( i ), x0 = ( i - y * segs ) << 3;
    int       sum[8];
#pragma unroll
    for( int k = 0; k < 8; k++ ) sum[k] = 0;
#pragma unroll
    for( int t = 0; t < NT; t++ )
    {
      int row[8];
      if( twoPass ) load8s( lds + ( y + t ) * w + x0, row );
      else load8g( src + ( long ) ( y + t - HALO ) * j.refStride + x0, row );
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum[k] += row[k] * cv[t];
    }
    int v[8];
#pragma unroll
    for( int k = 0; k < 8; k++ ) v[k] = fir_out( sum[k], fv );
    out.vec( y, x0, v );
  }
}

// luma / chroma, 8-outputs-per-lane path for widths that are multiples of 8, sample-per-lane path otherwise (4-wide luma, small chroma)
template<int THREADS, class Out>
__device__ __forceinline__ void mc_any( const vtmhip_mc_job &j, const int16_t *__restrict__ refBase, int16_t *lds, int lane, Out out )
{
  if( ( j.width & 7 ) == 0 )
  {
    if( j.chroma ) mc_block_vec<4, THREADS>( j, refBase, lds, lane, out );
    else mc_block_vec<8, THREADS>( j, refBase, lds, lane, out );
  }
  else
  {
    if( j.chroma ) mc_block<4, THREADS>( j, refBase, lds, lane, out );
    else mc_block<8, THREADS>( j, refBase, lds, lane, out );
  }
}

}   // namespace
