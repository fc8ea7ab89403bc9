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

__device__ __forceinline__ void wht1d_inl8( int *m )   // all three levels of an 8-point transform, in place
{
#pragma unroll
  for( int len = 1; len < 8; len <<= 1 )
#pragma unroll
    for( int i = 0; i < 8; i += len << 1 )
#pragma unroll
      for( int j = i; j < i + len; j++ )
      {
        const int a = m[j], b = m[j + len];
        m[j] = a + b; m[j + len] = a - b;
      }
}

// ---- 16x8 / 8x16 Hadamard tiles from two 8x8 halves in neighbouring lanes ------------------------------------------------------------------------
// A 16x8 (8x16) Hadamard is the 8x8 transforms A, B of its left / right (upper / lower) halves plus one butterfly level between them:
// coefficients A_i + B_i and A_i - B_i, so sum |coef| = 2 sum_i max(|A_i|, |B_i|) and the dc term is |A_0 + B_0| (RdCost.cpp xCalcHADs16x8 / 8x16:
// mean-scaled dc, then (int)(sad / sqrt(16.0 * 8) * 2)).  Lane 2k holds A, lane 2k + 1 holds B; both return the finished tile value.
__device__ __forceinline__ int dpp_swap1( int v ) { return __builtin_amdgcn_mov_dpp( v, 0xB1, 0xF, 0xF, false ); }   // quad_perm [1, 0, 3, 2]: the value of lane ^ 1

__device__ __forceinline__ unsigned satd_pair_norm( int sumAbs, int dcAbs )
{
  const int t = sumAbs - dcAbs + ( dcAbs >> 2 );
  return ( unsigned ) ( int ) ( ( double ) t / 11.313708498984761 * 2.0 );   // sqrt(16.0 * 8)
}

// c[64]: ALL six butterfly levels of this lane's 8x8 half done (32-bit)
__device__ __forceinline__ unsigned satd8_pair_finish32( const int *c )
{
  int t = 0;
#pragma unroll
  for( int i = 0; i < 64; i++ )
  {
    const int a = abs( c[i] );
    t += max( a, dpp_swap1( a ) );
  }
  const int dc = abs( c[0] + dpp_swap1( c[0] ) );
  return satd_pair_norm( t << 1, dc );
}

// packed differences with |d| <= 4095: three vertical levels packed, the horizontal ones in 32 bits, row by row
__device__ __forceinline__ unsigned satd8_pair_packed( v2s D[8][4] )
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
    wht1d_inl8( m );
    if( y == 0 ) dc = abs( m[0] + dpp_swap1( m[0] ) );
#pragma unroll
    for( int x = 0; x < 8; x++ )
    {
      const int a = abs( m[x] );
      t += max( a, dpp_swap1( a ) );
    }
  }
  return satd_pair_norm( t << 1, dc );
}

// packed differences with |d| <= 1023: the three vertical levels, one level between words and the level BETWEEN THE TWO LANES run packed (32 * 1023 fits
// int16); the last level between words and the one between the halves of a word in 32 bits, folded: |p + q| + |p - q| = 2 max(|p|, |q|).
__device__ __forceinline__ unsigned satd8_pair_packed10( v2s D[8][4], bool odd )
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
  const v2s sgn = odd ? v2s{ -1, -1 } : v2s{ 1, 1 };
  int       t = 0, dcl = 0;
#pragma unroll
  for( int y = 0; y < 8; y++ )
  {
#pragma unroll
    for( int k = 0; k < 4; k += 2 )
    {
      const v2s a = D[y][k], b = D[y][k + 1];
      D[y][k]     = a + b;
      D[y][k + 1] = a - b;
    }
#pragma unroll
    for( int k = 0; k < 4; k++ )   // between the lanes: even lane A + B, odd lane A - B
    {
      int own;
      __builtin_memcpy( &own, &D[y][k], 4 );
      const int oth = dpp_swap1( own );
      v2s       P;
      __builtin_memcpy( &P, &oth, 4 );
      D[y][k] = D[y][k] * sgn + P;
    }
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      const int p = ( int ) D[y][k].x + ( int ) D[y][k + 2].x, q = ( int ) D[y][k].y + ( int ) D[y][k + 2].y;
      const int r = ( int ) D[y][k].x - ( int ) D[y][k + 2].x, u = ( int ) D[y][k].y - ( int ) D[y][k + 2].y;
      t += max( abs( p ), abs( q ) ) + max( abs( r ), abs( u ) );
      if( y == 0 && k == 0 ) dcl = abs( p + q );
    }
  }
  const int tot = t + dpp_swap1( t );
  const int dco = dpp_swap1( dcl );
  return satd_pair_norm( tot << 1, odd ? dco : dcl );
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
