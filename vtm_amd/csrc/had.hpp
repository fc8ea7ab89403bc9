// had.hpp -- 8x8 Hadamard SATD on packed 16-bit differences (shared by dist.hip and interp.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace
{

// 8x8 SATD on PACKED 16-bit differences: valid when every |org - ref| <= 4095 (all samples in [0, 4095]), so that three butterfly levels stay
// inside int16.  D[y][k] = (d[y][2k], d[y][2k+1]); the vertical transform runs on the packed words (v_pk_add_i16 / v_pk_sub_i16: half the
// instructions), the horizontal one in 32 bits after unpacking, its last level folded into the absolute sum.
typedef short v2s __attribute__( ( ext_vector_type( 2 ) ) );

__device__ __forceinline__ unsigned satd8_packed( v2s D[8][4] )
{
#pragma unroll
  for( int len = 1; len < 8; len <<= 1 )
#pragma unroll
    for( int i = 0; i < 8; i += len << 1 )
#pragma unroll
      for( int j = i; j < i + len; j++ )
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const v2s a = D[j][k], b = D[j + len][k];
          D[j][k]       = a + b;
          D[j + len][k] = a - b;
        }
  int t = 0, dc = 0;
#pragma unroll
  for( int y = 0; y < 8; y++ )
  {
    int m[8];
#pragma unroll
    for( int k = 0; k < 4; k++ ) { m[2 * k] = D[y][k].x; m[2 * k + 1] = D[y][k].y; }
    // horizontal: levels 1 and 2, then |a + b| + |a - b| = 2 max(|a|, |b|) for the pairs (x, x + 4)
#pragma unroll
    for( int len = 1; len < 4; len <<= 1 )
#pragma unroll
      for( int i = 0; i < 8; i += len << 1 )
#pragma unroll
        for( int j = i; j < i + len; j++ )
        {
          const int a = m[j], b = m[j + len];
          m[j] = a + b; m[j + len] = a - b;
        }
#pragma unroll
    for( int x = 0; x < 4; x++ ) t += max( abs( m[x] ), abs( m[x + 4] ) );
    if( y == 0 ) dc = abs( m[0] + m[4] );
  }
  t <<= 1;
  t = t - dc + ( dc >> 2 );
  return ( unsigned ) ( ( t + 2 ) >> 2 );
}

// The same with five of the six butterfly levels in packed 16-bit arithmetic: valid when every |org - ref| <= 1023 (all samples in [0, 1023],
// 10-bit pictures), so that 32 * 1023 still fits int16.  The sixth level pairs the two halves of a word: |lo + hi| + |lo - hi| =
// 2 max(|lo|, |hi|): packed |.|, then one max of the two halves per word.
__device__ __forceinline__ unsigned satd8_packed10( v2s D[8][4] )
{
  // vertical: three levels between rows
#pragma unroll
  for( int len = 1; len < 8; len <<= 1 )
#pragma unroll
    for( int i = 0; i < 8; i += len << 1 )
#pragma unroll
      for( int j = i; j < i + len; j++ )
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const v2s a = D[j][k], b = D[j + len][k];
          D[j][k]       = a + b;
          D[j + len][k] = a - b;
        }
  // horizontal: the two levels between words (columns x vs x + 2, x vs x + 4)
#pragma unroll
  for( int y = 0; y < 8; y++ )
  {
#pragma unroll
    for( int len = 1; len < 4; len <<= 1 )
#pragma unroll
      for( int i = 0; i < 4; i += len << 1 )
#pragma unroll
        for( int j = i; j < i + len; j++ )
        {
          const v2s a = D[y][j], b = D[y][j + len];
          D[y][j]       = a + b;
          D[y][j + len] = a - b;
        }
  }
  const int dc = abs( ( int ) D[0][0].x + ( int ) D[0][0].y );
  unsigned  t  = 0;
  const v2s zero = { 0, 0 };
#pragma unroll
  for( int y = 0; y < 8; y++ )
#pragma unroll
    for( int k = 0; k < 4; k++ )
    {
      const v2s neg = zero - D[y][k];
      v2s       av;
      av.x = D[y][k].x > neg.x ? D[y][k].x : neg.x;      // packed |.| (v_pk_max_i16)
      av.y = D[y][k].y > neg.y ? D[y][k].y : neg.y;
      unsigned aw;
      __builtin_memcpy( &aw, &av, 4 );
      t += max( aw & 0xffffu, aw >> 16 );   // one v_max_u32 with sub-dword operand selects
    }
  const int tt = ( int ) ( t << 1 ) - dc + ( dc >> 2 );
  return ( unsigned ) ( ( tt + 2 ) >> 2 );
}

// ---- Hadamard building blocks (32-bit, order-free: only sum|coef| and coef[0] matter) ----------
template<int N, int STRIDE, bool LAST = true>
__device__ __forceinline__ void wht1d( int *m )   // LAST = false: every butterfly level but the last (len = N / 2)
{
#pragma unroll
  for( int len = 1; len < ( LAST ? N : N / 2 ); len <<= 1 )
  {
#pragma unroll
    for( int i = 0; i < N; i += len << 1 )
    {
#pragma unroll
      for( int j = i; j < i + len; j++ )
      {
        const int a = m[j * STRIDE], b = m[( j + len ) * STRIDE];
        m[j * STRIDE]           = a + b;
        m[( j + len ) * STRIDE] = a - b;
      }
    }
  }
}

// m[] holds the TW x TH differences (row-major) on entry.  Returns the per-tile SATD with the reference's
// mean-scaled dc (JVET_R0164) and per-shape normalisation.
template<int TW, int TH>
__device__ __forceinline__ unsigned had_finish( int *m )
{
#pragma unroll
  for( int y = 0; y < TH; y++ ) wht1d<TW, 1>( m + y * TW );
#pragma unroll
  for( int x = 0; x < TW; x++ ) wht1d<TH, TW, false>( m + x );
  // last butterfly level + |.| + sum in one: |a + b| + |a - b| = 2 max(|a|, |b|) for the pair (row j, row j + TH/2) of a column
  int t = 0;
#pragma unroll
  for( int i = 0; i < TW * TH / 2; i++ ) t += max( abs( m[i] ), abs( m[i + TW * TH / 2] ) );
  t <<= 1;
  const int dc = abs( m[0] + m[TW * TH / 2] );
  t            = t - dc + ( dc >> 2 );
  if( TW == 2 && TH == 2 ) return ( unsigned ) t;
  if( TW == 4 && TH == 4 ) return ( unsigned ) ( ( t + 1 ) >> 1 );
  if( TW == 8 && TH == 8 ) return ( unsigned ) ( ( t + 2 ) >> 2 );
  if( TW * TH == 128 ) return ( unsigned ) ( int ) ( ( double ) t / 11.313708498984761 * 2.0 );   // sqrt(16.0*8)
  return ( unsigned ) ( int ) ( ( double ) t / 5.656854249492381 * 2.0 );                        // sqrt(4.0*8)
}

template<int TW, int TH>
__device__ __forceinline__ unsigned had_tile( const int16_t *o, int os, const int16_t *c, int cs )
{
  int m[TW * TH];
#pragma unroll
  for( int y = 0; y < TH; y++ )
  {
#pragma unroll
    for( int x = 0; x < TW; x++ ) m[y * TW + x] = ( int ) o[y * os + x] - ( int ) c[y * cs + x];
  }
  return had_finish<TW, TH>( m );
}

}   // namespace
