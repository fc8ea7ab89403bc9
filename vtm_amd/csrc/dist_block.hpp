// dist_block.hpp -- the distortion of ONE (original block, candidate block) pair by one wave: SAD with row sub-sampling (RdCost.cpp:493-1003), SSE (:1783-2133) or SATD by
// the tile rules of xGetHADs (:2837-2931).  Shared by dist_batch_kernel (dist.hip) and the fused AMVR refinement of xMotionEstimation (mest.hip).  The result is in every lane.
#pragma once
#include "ctx.hpp"
#include "had.hpp"

__device__ __forceinline__ unsigned long long wave_block_dist( int kind, const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, int subShift, int lane )
{
  unsigned long long acc = 0;

  if( kind == VTMHIP_DIST_SAD )
  {
    // rows y = 0, step, 2*step ...; work items = (row, 4-sample segment), or single samples for widths like 2 / 6 (chroma)
    const int ss = subShift, rows = ( h + ( 1 << ss ) - 1 ) >> ss, segs = w >> 2;
    unsigned  s = 0;
    if( ( w & 3 ) == 0 )
    {
      for( int it = lane; it < rows * segs; it += 64 )
      {
        const int      r = it / segs, x = ( it - r * segs ) << 2;
        const int16_t *o = org + ( long ) ( r << ss ) * os + x;
        const int16_t *c = cur + ( long ) ( r << ss ) * cs + x;
#pragma unroll
        for( int k = 0; k < 4; k++ ) s += ( unsigned ) abs( ( int ) o[k] - ( int ) c[k] );
      }
    }
    else
    {
      for( int it = lane; it < rows * w; it += 64 )
      {
        const int r = it / w, x = it - r * w;
        s += ( unsigned ) abs( ( int ) org[( long ) ( r << ss ) * os + x] - ( int ) cur[( long ) ( r << ss ) * cs + x] );
      }
    }
    acc = ( unsigned long long ) s << ss;   // per-lane partial (W*H*65535 < 2^32 for W,H <= 128 needs care: 128*128*65535 = 2^30)
  }
  else if( kind == VTMHIP_DIST_SSE )
  {
    const int segs = w >> 2;
    if( ( w & 3 ) == 0 )
    {
      for( int it = lane; it < h * segs; it += 64 )
      {
        const int      r = it / segs, x = ( it - r * segs ) << 2;
        const int16_t *o = org + ( long ) r * os + x;
        const int16_t *c = cur + ( long ) r * cs + x;
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const int d = ( int ) o[k] - ( int ) c[k];
          acc += ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );   // per-addend 32-bit product as RdCost.cpp:1783-1814
        }
      }
    }
    else
    {
      for( int it = lane; it < h * w; it += 64 )
      {
        const int r = it / w, x = it - r * w;
        const int d = ( int ) org[( long ) r * os + x] - ( int ) cur[( long ) r * cs + x];
        acc += ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
      }
    }
  }
  else   // SATD: tile shape by the rules of xGetHADs (RdCost.cpp:2837-2931); one tile per lane
  {
    int tw, th;
    if( w > h && ( h & 7 ) == 0 && ( w & 15 ) == 0 ) { tw = 16; th = 8; }
    else if( w < h && ( w & 7 ) == 0 && ( h & 15 ) == 0 ) { tw = 8; th = 16; }
    else if( w > h && ( h & 3 ) == 0 && ( w & 7 ) == 0 ) { tw = 8; th = 4; }
    else if( w < h && ( w & 3 ) == 0 && ( h & 7 ) == 0 ) { tw = 4; th = 8; }
    else if( ( h & 7 ) == 0 && ( w & 7 ) == 0 ) { tw = 8; th = 8; }
    else if( ( h & 3 ) == 0 && ( w & 3 ) == 0 ) { tw = 4; th = 4; }
    else { tw = 2; th = 2; }
    const int tx = w / tw, ty = h / th;
    for( int it = lane; it < tx * ty; it += 64 )
    {
      const int      y = ( it / tx ) * th, x = ( it % tx ) * tw;
      const int16_t *o = org + ( long ) y * os + x;
      const int16_t *c = cur + ( long ) y * cs + x;
      unsigned       v;
      if( tw == 16 ) v = had_tile<16, 8>( o, os, c, cs );
      else if( th == 16 ) v = had_tile<8, 16>( o, os, c, cs );
      else if( tw == 8 && th == 4 ) v = had_tile<8, 4>( o, os, c, cs );
      else if( tw == 4 && th == 8 ) v = had_tile<4, 8>( o, os, c, cs );
      else if( tw == 8 ) v = had_tile<8, 8>( o, os, c, cs );
      else if( tw == 4 ) v = had_tile<4, 4>( o, os, c, cs );
      else v = had_tile<2, 2>( o, os, c, cs );
      acc += v;
    }
  }
  return wave_reduce_add_u64( acc );
}
