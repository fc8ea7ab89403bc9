// tr_tables.hpp -- host-side generator of the VVC core transform matrices (H.266 8.7.4.2 transMatrix, 6-bit precision;
// the reference keeps them as literal tables in CommonLib/RomTr.cpp:432-...).
//
//  * DCT-2: the 64-point matrix has 63 distinct magnitudes c[j] ~ 64*sqrt(2)*cos(j*pi/128), j = 1..63 (hand-tuned in the
//    standard, so they are listed); M64[k][n] = +-c[fold(k*(2n+1) mod 256)], row 0 = 64.  The N-point matrix is the
//    top-left N columns of every (64/N)-th row.
//  * DST-7: M[k][n] = +-r[fold((2k+1)(n+1) mod (4N+2))] with r = first row of the N-point matrix (listed per size).
//  * DCT-8: M[k][n] = (-1)^k * DST7[k][N-1-n].
// tests/test_oracle_vs_ref.py checks all 14 matrices against the reference's g_trCore* arrays.
#pragma once
#include <cstdint>
#include <vector>

inline int vtmhip_tr_matrix( int type, int n, int16_t *out )
{
  static const int8_t dct2Mag[64] = { 0,
    91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80, 79, 78, 77, 75, 73, 73, 71, 70, 69, 67, 65, 64,
    62, 61, 59, 57, 56, 54, 52, 50, 48, 46, 44, 43, 41, 38, 37, 36, 33, 31, 28, 25, 24, 22, 20, 18, 15, 13, 11, 9,  7,  4,  2 };
  static const int8_t dst7r4[4]   = { 29, 55, 74, 84 };
  static const int8_t dst7r8[8]   = { 17, 32, 46, 60, 71, 78, 85, 86 };
  static const int8_t dst7r16[16] = { 8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88 };
  static const int8_t dst7r32[32] = { 4,  9,  13, 17, 21, 26, 30, 34, 38, 42, 46, 50, 53, 56, 60, 63,
                                      66, 68, 72, 74, 77, 78, 80, 82, 84, 85, 86, 87, 88, 89, 90, 90 };
  if( type == 0 )
  {
    if( n != 2 && n != 4 && n != 8 && n != 16 && n != 32 && n != 64 ) return -1;
    const int s = 64 / n;
    for( int k = 0; k < n; k++ )
      for( int x = 0; x < n; x++ )
      {
        if( k == 0 ) { out[x] = 64; continue; }
        int j = ( ( k * s ) * ( 2 * x + 1 ) ) % 256, sign = 1;
        if( j > 128 ) j = 256 - j;
        if( j > 64 ) { j = 128 - j; sign = -1; }
        out[k * n + x] = ( int16_t ) ( sign * dct2Mag[j] );
      }
    return 0;
  }
  const int8_t *r = n == 4 ? dst7r4 : n == 8 ? dst7r8 : n == 16 ? dst7r16 : n == 32 ? dst7r32 : nullptr;
  if( !r || ( type != 1 && type != 2 ) ) return -1;
  const int p = 2 * n + 1;
  for( int k = 0; k < n; k++ )
    for( int x = 0; x < n; x++ )
    {
      int m = ( ( 2 * k + 1 ) * ( x + 1 ) ) % ( 2 * p ), sign = 1;
      if( m > p ) { m = 2 * p - m; sign = -1; }
      if( m > n ) m = p - m;
      const int v = m == 0 ? 0 : sign * r[m - 1];
      if( type == 2 ) out[k * n + x] = ( int16_t ) v;
      else out[k * n + ( n - 1 - x )] = ( int16_t ) ( ( k & 1 ) ? -v : v );
    }
  return 0;
}
