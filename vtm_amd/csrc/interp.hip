// interp.hip -- sub-sample interpolation (8-tap luma / 4-tap chroma / 2-tap bilinear / copy) and the fused
// fractional motion search (interpolate -> SATD -> first-strict-minimum) of one PU.
//
// Reference: CommonLib/InterpolationFilter.cpp filter<> :548-651, filterCopy<> :398-525, tap tables :57-330;
// EncoderLib/InterSearch.cpp xPatternSearchFracDIF :4284-4339, xExtDIFUpSamplingH/Q :5840-6051,
// xPatternRefinement :707-761 (s_acMvRefineH/Q :60-85).
//
// Arithmetic (bit-exact): val = (int16)((sum_k src[.]*c[k] + offset) >> shift), clip only when isLast
// (the int16 truncation precedes the clip, InterpolationFilter.cpp:645-650).
//
// Fractional search: the reference builds 4 + 10 shifted planes (m_filteredBlock[v][h]) and indexes them with pointer
// nudges; every candidate block is nothing but the separable interpolation of the reference picture at
// (intMv*4 + q) quarter samples: H pass (first, !last) on rows -3..H+3, V pass (!first, last).  Integer phases go
// through the same FIR with taps {0,0,0,64,0,0,0,0}, which is arithmetically identical to filterCopy.
// One wave per PU: window -> LDS once, 3 H passes per round (one per horizontal phase), 9 V passes + SATDs.
#include "ctx.hpp"
#include "mest_glue.hpp"
#include "had.hpp"

#include <type_traits>

namespace
{

__constant__ int16_t c_lumaFilter[16][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },       { 0, 1, -3, 63, 4, -2, 1, 0 },     { -1, 2, -5, 62, 8, -3, 1, 0 },    { -1, 3, -8, 60, 13, -4, 1, 0 },
  { -1, 4, -10, 58, 17, -5, 1, 0 },  { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
  { -1, 4, -11, 40, 40, -11, 4, -1 },{ -1, 4, -10, 34, 45, -11, 4, -1 },{ -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
  { 0, 1, -5, 17, 58, -10, 4, -1 },  { 0, 1, -4, 13, 60, -8, 3, -1 },   { 0, 1, -3, 8, 62, -5, 2, -1 },    { 0, 1, -2, 4, 63, -3, 1, 0 } };
__constant__ int16_t c_lumaAltHpel[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };

struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8u { unsigned v[4]; };   // 8 samples from a 2-byte aligned address

struct IfParams { int shift, offset, clip, cmin, cmax; };

// shift / offset rules of InterpolationFilter::filter (:577-614)
__device__ __forceinline__ IfParams if_params( int isFirst, int isLast, int bitDepth, int clipMin, int clipMax, int biMC )
{
  IfParams  p;
  const int headRoom = max( 2, 14 - bitDepth );
  int       shift    = 6, offset;
  if( isLast )
  {
    shift += isFirst ? 0 : headRoom;
    offset = 1 << ( shift - 1 );
    offset += isFirst ? 0 : ( 8192 << 6 );
  }
  else
  {
    shift -= isFirst ? headRoom : 0;
    offset = isFirst ? -( 8192 << shift ) : 0;
  }
  if( biMC )
  {
    shift  = isFirst ? 4 - ( 10 - bitDepth ) : 4;
    offset = 1 << ( shift - 1 );
  }
  p.shift = shift; p.offset = offset; p.clip = isLast; p.cmin = clipMin; p.cmax = clipMax;
  return p;
}

__device__ __forceinline__ int16_t if_finish( int sum, const IfParams &p )
{
  int16_t v = ( int16_t ) ( ( sum + p.offset ) >> p.shift );
  if( p.clip ) v = ( int16_t ) min( p.cmax, max( p.cmin, ( int ) v ) );
  return v;
}

// ---- generic batched filter: one WAVE per job (four jobs per workgroup); 8 outputs per lane with 16-byte loads / stores when the width is
//      a multiple of 8, one output per lane otherwise ------------------------------------------------------------------------------------
__device__ __forceinline__ void if_unpack8( const Pel8u &t, int a[8] )
{
#pragma unroll
  for( int k = 0; k < 4; k++ ) { a[2 * k] = ( int ) ( short ) ( t.v[k] & 0xffffu ); a[2 * k + 1] = ( int ) t.v[k] >> 16; }
}

__global__ __launch_bounds__( 256 ) void if_batch_kernel( const int16_t *__restrict__ srcBase, int16_t *__restrict__ dstBase,
                                                         const vtmhip_if_job *__restrict__ jobs, int n )
{
  const int lane = threadIdx.x & 63, job = blockIdx.x * 4 + ( threadIdx.x >> 6 );
  if( job >= n ) return;
  const vtmhip_if_job j   = jobs[job];
  const int16_t      *src = srcBase + j.srcOff;
  int16_t            *dst = dstBase + j.dstOff;
  const int           w = j.width, h = j.height, taps = j.taps;
  if( taps == 0 )   // filterCopy<isFirst,isLast> (:398-525)
  {
    const int headRoom = max( 2, 14 - ( int ) j.bitDepth );
    for( int i = lane; i < w * h; i += 64 )
    {
      const int y = i / w, x = i - y * w;
      const int s = src[( long ) y * j.srcStride + x];
      int16_t   v;
      if( j.isFirst == j.isLast ) v = ( int16_t ) s;
      else if( j.biMCForDMVR )
      {
        if( j.bitDepth > 10 ) { const int sh = j.bitDepth - 10; v = ( int16_t ) ( ( s + ( 1 << ( sh - 1 ) ) ) >> sh ); }
        else v = ( int16_t ) ( s << ( 10 - j.bitDepth ) );
      }
      else if( j.isFirst ) v = ( int16_t ) ( ( int16_t ) ( s << headRoom ) - ( int16_t ) 8192 );
      else
      {
        const int16_t t = ( int16_t ) ( ( s + 8192 + ( 1 << ( headRoom - 1 ) ) ) >> headRoom );
        v               = ( int16_t ) min( ( int ) j.clipMax, max( ( int ) j.clipMin, ( int ) t ) );
      }
      dst[( long ) y * j.dstStride + x] = v;
    }
    return;
  }
  const IfParams p       = if_params( j.isFirst, j.isLast, j.bitDepth, j.clipMin, j.clipMax, j.biMCForDMVR );
  const long     cStride = j.vertical ? j.srcStride : 1;
  const int      halo    = taps / 2 - 1;
  if( ( w & 7 ) == 0 )
  {
    int c[8];
#pragma unroll
    for( int k = 0; k < 8; k++ ) c[k] = k < taps ? ( int ) j.coeff[k] : 0;
    const int segs = w >> 3;
    for( int i = lane; i < segs * h; i += 64 )
    {
      const int y = i / segs, x0 = ( i - y * segs ) << 3;
      int       sum[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum[k] = 0;
      if( j.vertical )
      {
        for( int t = 0; t < taps; t++ )
        {
          int row[8];
          if_unpack8( *reinterpret_cast<const Pel8u *>( src + ( long ) ( y + t - halo ) * j.srcStride + x0 ), row );
#pragma unroll
          for( int k = 0; k < 8; k++ ) sum[k] += row[k] * c[t];
        }
      }
      else
      {
        int            a[16];
        const int16_t *s0 = src + ( long ) y * j.srcStride + x0 - halo;
        if_unpack8( *reinterpret_cast<const Pel8u *>( s0 ), a );
        if_unpack8( *reinterpret_cast<const Pel8u *>( s0 + 8 ), a + 8 );
#pragma unroll
        for( int k = 0; k < 8; k++ )
        {
#pragma unroll
          for( int t = 0; t < 8; t++ ) sum[k] += a[k + t] * c[t];   // taps < 8: the remaining coefficients are zero
        }
      }
      Pel8u o;
#pragma unroll
      for( int k = 0; k < 4; k++ )
        o.v[k] = ( ( unsigned ) ( unsigned short ) if_finish( sum[2 * k], p ) ) | ( ( unsigned ) ( unsigned short ) if_finish( sum[2 * k + 1], p ) << 16 );
      *reinterpret_cast<Pel8u *>( dst + ( long ) y * j.dstStride + x0 ) = o;
    }
    return;
  }
  src -= halo * cStride;
  for( int i = lane; i < w * h; i += 64 )
  {
    const int      y = i / w, x = i - y * w;
    const int16_t *s = src + ( long ) y * j.srcStride + x;
    int            sum = 0;
    for( int k = 0; k < taps; k++ ) sum += ( int ) s[k * cStride] * ( int ) j.coeff[k];
    dst[( long ) y * j.dstStride + x] = if_finish( sum, p );
  }
}

// (Hadamard tiles: had.hpp)

// distortion of the W x H block: org from global memory, candidate block in LDS (stride W); whole wave, result uniform
__device__ __forceinline__ unsigned long long block_dist( const int16_t *org, int os, const int16_t *pred, int w, int h, bool useHad, int lane )
{
  unsigned long long acc = 0;
  if( !useHad )
  {
    unsigned s = 0;
    for( int i = lane; i < w * h; i += 64 )
    {
      const int y = i / w, x = i - y * w;
      s += ( unsigned ) abs( ( int ) org[( long ) y * os + x] - ( int ) pred[y * w + x] );
    }
    acc = s;
  }
  else
  {
    int tw, th;   // xGetHADs tile rules (RdCost.cpp:2837-2931)
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
      const int16_t *c = pred + y * w + x;
      unsigned       v;
      if( tw == 16 ) v = had_tile<16, 8>( o, os, c, w );
      else if( th == 16 ) v = had_tile<8, 16>( o, os, c, w );
      else if( tw == 8 && th == 4 ) v = had_tile<8, 4>( o, os, c, w );
      else if( tw == 4 && th == 8 ) v = had_tile<4, 8>( o, os, c, w );
      else if( tw == 8 ) v = had_tile<8, 8>( o, os, c, w );
      else if( tw == 4 ) v = had_tile<4, 4>( o, os, c, w );
      else v = had_tile<2, 2>( o, os, c, w );
      acc += v;
    }
  }
  return wave_reduce_add_u64( acc );
}

__device__ __forceinline__ int floor_log2_u( unsigned v ) { return 31 - __clz( ( int ) v ); }
__device__ __forceinline__ unsigned eg_bits( int v )
{
  // xGetExpGolombNumberOfBits (RdCost.h:301-313): its `while( t > 128 ) { len += 14; t >>= 7; }` only splits floorLog2( t ) = 7 + floorLog2( t >> 7 ),
  // so the length is 1 + 2 * floorLog2( t ) for every t >= 1 -- no loop
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}
// getCostOfVectorWithPredictor( x, y, 0 ) with m_iCostScale = costScale (RdCost.h:314-315)
__device__ __forceinline__ unsigned long long mv_cost( double lambda, int predHor, int predVer, int costScale, int x, int y )
{
  const unsigned bits = eg_bits( ( x << costScale ) - predHor ) + eg_bits( ( y << costScale ) - predVer );
  return ( unsigned long long ) ( lambda * ( double ) bits );
}

__constant__ int8_t c_refineH[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
__constant__ int8_t c_refineQ[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };

// One refinement round (xPatternRefinement :707-761).  (cx, cy): round centre in quarter samples relative to the
// integer vector; step: 2 (half) or 1 (quarter).  Returns the best table index; cost[] gets the 9 costs.
__device__ int refine_round( const vtmhip_frac_job &j, const int16_t *org, const int16_t *win, int winLd, int16_t *tmp, int16_t *pred, int cx,
                             int cy, int step, int costScale, int mvBaseX, int mvBaseY, bool altHpel, unsigned long long *bestCost, int lane )
{
  const int w = j.width, h = j.height;
  const int8_t( *tab )[2] = step == 2 ? c_refineH : c_refineQ;
  const IfParams pH = if_params( 1, 0, j.bitDepth, 0, ( 1 << j.bitDepth ) - 1, 0 );
  const IfParams pV = if_params( 0, 1, j.bitDepth, 0, ( 1 << j.bitDepth ) - 1, 0 );
  unsigned long long cost[9];
#pragma unroll
  for( int i = 0; i < 9; i++ ) cost[i] = ~0ull;

  for( int dx = -1; dx <= 1; dx++ )
  {
    const int      qx = cx + dx * step, ix = qx >> 2, fx = qx & 3;
    const int16_t *cH = ( altHpel && fx == 2 ) ? c_lumaAltHpel : c_lumaFilter[fx << 2];
    int            ch[8];
#pragma unroll
    for( int k = 0; k < 8; k++ ) ch[k] = cH[k];
    // H pass (first, !last): tmp[r][x], r = 0..h+7 <-> picture rows -4..h+3 (window rows r), columns x + ix (+4 in the window)
    for( int i = lane; i < ( h + 8 ) * w; i += 64 )
    {
      const int      r = i / w, x = i - r * w;
      const int16_t *s = win + r * winLd + ( x + ix + 4 - 3 );
      int            sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) s[k] * ch[k];
      tmp[i] = if_finish( sum, pH );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
    for( int dy = -1; dy <= 1; dy++ )
    {
      const int      qy = cy + dy * step, iy = qy >> 2, fy = qy & 3;
      const int16_t *cV = ( altHpel && fy == 2 ) ? c_lumaAltHpel : c_lumaFilter[fy << 2];
      int            cv[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) cv[k] = cV[k];
      // V pass (!first, last): output row y uses tmp rows (y + iy + 4 - 3) .. +7
      for( int i = lane; i < h * w; i += 64 )
      {
        const int      y = i / w, x = i - y * w;
        const int16_t *s = tmp + ( y + iy + 1 ) * w + x;
        int            sum = 0;
#pragma unroll
        for( int k = 0; k < 8; k++ ) sum += ( int ) s[k * w] * cv[k];
        pred[i] = if_finish( sum, pV );
      }
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      __builtin_amdgcn_wave_barrier();
      unsigned long long d = block_dist( org, j.orgStride, pred, w, h, j.useHad != 0, lane );
      d += mv_cost( j.motionLambda, j.predHor, j.predVer, costScale, mvBaseX + dx, mvBaseY + dy );
#pragma unroll
      for( int i = 0; i < 9; i++ )
        if( tab[i][0] == dx && tab[i][1] == dy ) cost[i] = d;
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      __builtin_amdgcn_wave_barrier();
    }
  }
  unsigned long long best = ~0ull;
  int                bi   = 0;
#pragma unroll
  for( int i = 0; i < 9; i++ )
    if( cost[i] < best ) { best = cost[i]; bi = i; }   // first strict minimum in table order
  *bestCost = best;
  return bi;
}

__global__ __launch_bounds__( 64 ) void frac_search_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                           const vtmhip_frac_job *__restrict__ jobs, vtmhip_frac_result *__restrict__ results,
                                                           int maxW, int maxH, FracFuse fu )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];
  const int             lane = threadIdx.x;
  vtmhip_frac_job       j;
  if( fu.me )      // FUSED (mest_glue.hpp): the job from the xMotionEstimation row and its integer result (what mest_mid_kernel would have written)
  {
    const vtmhip_me_job   &mj = fu.me[blockIdx.x];
    const vtmhip_me_result r  = fu.ires[blockIdx.x];
    if( mj.imv != 0 && mj.imv != 3 ) return;      // an AMVR row of a mixed batch: integer refinement, not this kernel's (the stand-alone chain handles mixed batches)
    mg::make_frac_job( fu.useHadME, fu.bitDepth, mj, r.mvX, r.mvY, fuse_pat_off( fu, mj ), fuse_pat_stride( fu, mj ), j );
  }
  else j = jobs[blockIdx.x];
  const int             w = j.width, h = j.height;
  if( w == 0 ) return;   // empty slot of a multi-stage call
  const int             winLd = w + 8;
  int16_t              *win  = lds;                                   // [(h+8)][w+8]: picture rows/cols -4 .. +3 around the block
  int16_t              *tmp  = win + ( maxH + 8 ) * ( maxW + 8 );     // [(h+8)][w]
  int16_t              *pred = tmp + ( maxH + 8 ) * maxW;             // [h][w]
  const int16_t        *org  = orgBase + j.orgOff;
  const int16_t        *ref  = refBase + j.refOff + ( long ) j.intY * j.refStride + j.intX;   // cPatternRoi (:4298-4299)

  for( int i = lane; i < ( h + 8 ) * winLd; i += 64 )
  {
    const int r = i / winLd, c = i - r * winLd;
    win[i]      = ref[( long ) ( r - 4 ) * j.refStride + ( c - 4 )];
  }
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  __builtin_amdgcn_wave_barrier();

  vtmhip_frac_result res;
  res.halfX = res.halfY = res.qterX = res.qterY = 0;
  unsigned long long cost = 0;
  const bool         altHpel = j.useAltHpelIf != 0;

  // half-sample refinement: setCostScale(1), rcMvHalf = rcMvInt << 1 (:4320-4325)
  int bi    = refine_round( j, org, win, winLd, tmp, pred, 0, 0, 2, 1, j.intX << 1, j.intY << 1, altHpel, &cost, lane );
  res.halfX = c_refineH[bi][0];
  res.halfY = c_refineH[bi][1];
  if( j.imvShift == 0 )
  {
    // quarter-sample refinement: setCostScale(0), rcMvQter = ((rcMvInt << 1) + rcMvHalf) << 1 (:4328-4337)
    const int bx = ( ( j.intX << 1 ) + res.halfX ) << 1, by = ( ( j.intY << 1 ) + res.halfY ) << 1;
    bi        = refine_round( j, org, win, winLd, tmp, pred, res.halfX * 2, res.halfY * 2, 1, 0, bx, by, false, &cost, lane );
    res.qterX = c_refineQ[bi][0];
    res.qterY = c_refineQ[bi][1];
  }
  res.cost = cost;
  if( lane == 0 && results ) results[blockIdx.x] = res;
  if( lane == 0 && fu.me )      // the row's final record: the rate re-weighting mest_final_kernel would have done
  {
    const vtmhip_me_result r = fu.ires[blockIdx.x];
    vtmhip_me_out          o;
    mg::make_out_frac( fu.me[blockIdx.x], r.mvX, r.mvY, r.dist, res, o );
    fu.out[blockIdx.x] = o;
  }
}


// =====================================================================================================================
// Tiled fractional search for uniform W x H PUs (squares 8 .. 128 and the binary / ternary split shapes 16x8 .. 64x32, both orientations): the fast path.
//
// The generic kernel above gives a whole wave to one PU and its SATD phase keeps one lane per 8x8 tile busy -- one lane
// of 64 for an 8x8 PU.  Here a workgroup takes JPW PUs and every lane owns one (PU, candidate, 8x8 tile) ITEM:
//   phase H   3 horizontal passes per PU (one per horizontal quarter position of the round) -> LDS planes [(H+8)][W]
//   phase V   per item: 15 plane rows (16-byte LDS reads) -> 8x8 vertical FIR in registers (64 accumulators) ->
//             round/clip -> difference to the original tile -> 8x8 Hadamard in registers -> LDS atomic add into cost[PU][cand]
//   select    one lane per PU: + MV rate, first strict minimum in the reference's table order (xPatternRefinement :707-761)
// The fractional search has no data-dependent control flow (2 rounds x 9 candidates), so PUs batch perfectly.
// Rectangles: the reference's SATD tiles are 16x8 (W > H) or 8x16 (W < H) Hadamards (RdCost.cpp:2837-2931).  Such a transform is the 8x8 transforms A, B of
// its two halves plus one more butterfly level, sum |A_i + B_i| + |A_i - B_i| = 2 sum max(|A_i|, |B_i|): the two 8x8 items of a tile sit in
// neighbouring lanes (tile order: the long direction fastest) and finish through one DPP exchange (had.hpp satd8_pair_*).
// =====================================================================================================================
// Plane slots of the tiled search (three per PU): the half-sample round stores plane dx in slot dx + 1; the quarter-sample round keeps the slot of its centre plane
// (the half-sample winner's plane, centreX = 2 * halfX) and puts dx = -1 / +1 into the two others.
__device__ __forceinline__ int plane_slot( int round, int centreX, int dx )
{
  if( round == 0 ) return dx + 1;
  const int c = ( centreX >> 1 ) + 1, s = c + ( dx == 0 ? 0 : dx < 0 ? 1 : 2 );
  return s >= 3 ? s - 3 : s;
}

template<int W, int H>
struct FracSq
{
  static constexpr int TX = W / 8, TY = H / 8;
  static constexpr int TILES = TX * TY;
  static constexpr bool PAIR = W != H;                          // 16x8 / 8x16 Hadamard tiles: two neighbouring 8x8 items per tile
  static constexpr int ITEMS = 9 * TILES;                       // per PU per round
  static constexpr int BLOCK = TILES == 16 ? 192 : TILES == 256 ? 512 : 256;   // 128x128: 72 KB of LDS = two workgroups per CU, so eight waves each (768 items per plane pass = 1.5 trips): 3.13 -> 3.03 ms per picture; 16 waves 3.09; 64x64 with six waves 3.23
  static constexpr int JPW   = TILES == 16 ? 2 : ITEMS >= BLOCK ? 1 : BLOCK / ITEMS;   // 8x8: 28, 16x8: 14, 16x16 / 32x8: 7, 32x16: 3, larger: 1
#ifndef VTMHIP_FRAC_RECT_MINW
#define VTMHIP_FRAC_RECT_MINW 4
#endif
  static constexpr int MINW  = ( W != H ) ? VTMHIP_FRAC_RECT_MINW : 4;   // waves per SIMD the register budget is sized for (128 VGPRs; the dot-product V pass fits every square size; the rectangles' Hadamard pairs spill 11 - 21 registers there)
  static constexpr int WLD   = W + 8;                           // window stride
  static constexpr int WIN   = ( H + 8 ) * WLD;                 // window samples per PU
  static constexpr int PLANE = ( H + 8 ) * W;                   // one H-pass plane
  static constexpr bool SEQ  = W == 128 && H == 128;            // one plane buffer, the three planes one after the other: 72 KB instead of 141 KB of LDS -> two workgroups per CU
  static constexpr int NPL   = SEQ ? 1 : 3;
  static constexpr int PERJOB = ( ( WIN + NPL * PLANE ) + 7 ) & ~7;   // samples, keeps every plane 16-byte aligned
  static constexpr size_t LDS = ( size_t ) JPW * PERJOB * sizeof( int16_t ) + ( size_t ) JPW * 16 * sizeof( unsigned );
};

template<int W, int H>
__global__ __launch_bounds__( ( FracSq<W, H>::BLOCK ), ( FracSq<W, H>::MINW ) ) void frac_search_sq_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                                            const vtmhip_frac_job *__restrict__ jobs, int numJobs,
                                                                            vtmhip_frac_result *__restrict__ results, FracFuse fu )
{
  using C = FracSq<W, H>;
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];
  unsigned *sCost = reinterpret_cast<unsigned *>( lds + C::JPW * C::PERJOB );   // [JPW][16]: 9 candidate distortions (+ scratch)
  __shared__ int sCentre[C::JPW][2];                                            // half-sample winner per PU (round 2 centre)
  __shared__ unsigned sKeep[C::JPW];                                            // its distortion = candidate 0 of round 2 (same samples)
  const int tid = threadIdx.x;
  const int job0 = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x ) * C::JPW;   // neighbouring PUs share most of their windows: keep them on one XCD's L2
  const int nj = min( C::JPW, numJobs - job0 );
  // FUSED (fu.me != nullptr, mest_glue.hpp): the workgroup's job records from the xMotionEstimation rows and their integer results, into LDS (what mest_mid_kernel wrote to a table)
  // (the records sit in LDS either way -- copied from the job table when not fused -- so that every later read is an LDS read at a 32-bit address)
  __shared__ vtmhip_frac_job sFJ[C::JPW];
  const vtmhip_frac_job     *jb = sFJ;
  if( tid < nj )
  {
    if( fu.me )
    {
      const vtmhip_me_job   &mj = fu.me[job0 + tid];
      const vtmhip_me_result r  = fu.ires[job0 + tid];
      mg::make_frac_job( fu.useHadME, fu.bitDepth, mj, r.mvX, r.mvY, fuse_pat_off( fu, mj ), fuse_pat_stride( fu, mj ), sFJ[tid] );
    }
    else sFJ[tid] = jobs[job0 + tid];
  }
  __syncthreads();

  // ---- windows: rows -4 .. H+3, columns -4 .. W+3 around the integer vector; 8 samples (16 bytes) per thread and step ----------------------
  {
    constexpr int CH = C::WLD / 8;   // 16-byte chunks per window row
    for( int i = tid; i < nj * ( H + 8 ) * CH; i += C::BLOCK )
    {
      const int               jl = i / ( ( H + 8 ) * CH ), rem = i - jl * ( H + 8 ) * CH, r = rem / CH, c = ( rem - r * CH ) * 8;
      const vtmhip_frac_job &j  = jb[jl];
      const int16_t         *ref = refBase + j.refOff + ( long ) ( j.intY + r - 4 ) * j.refStride + ( j.intX + c - 4 );
      const Pel8u            v   = *reinterpret_cast<const Pel8u *>( ref );
      *reinterpret_cast<int4 *>( lds + jl * C::PERJOB + r * C::WLD + c ) = make_int4( ( int ) v.v[0], ( int ) v.v[1], ( int ) v.v[2], ( int ) v.v[3] );
    }
  }
  for( int i = tid; i < C::JPW * 2; i += C::BLOCK ) sCentre[i >> 1][i & 1] = 0;
  // the packed 16-bit evaluation (phase V) needs every PU of the workgroup to ask for the Hadamard cost at bitDepth <= 10
  const bool pkAll = !__syncthreads_or( tid < nj && !( jb[tid].useHad && jb[tid].bitDepth <= 10 && !jb[tid].wideOrg ) );   // wideOrg: a BCW-weighted target

#pragma unroll 1
  for( int round = 0; round < 2; round++ )
  {
    const int step = round == 0 ? 2 : 1;
    if( round == 1 && jb[0].imvShift != 0 ) break;   // IMV_HPEL: half-sample refinement only (uniform per batch, checked on the host side)
    // round 2: candidate 0 is the half-sample winner itself -- the block round 1 already measured -- so only 8 candidates are formed
    for( int i = tid; i < C::JPW * 16; i += C::BLOCK ) sCost[i] = ( round == 1 && ( i & 15 ) == 0 ) ? sKeep[i >> 4] : 0u;
#pragma unroll 1
    for( int seqPl = 0; seqPl < ( C::SEQ ? 3 : 1 ); seqPl++ )   // SEQ: plane by plane through ONE buffer; otherwise a single trip over all three
    {
    // ---- phase H: plane p (dx = p - 1) of PU jl: (first, !last) 8-tap FIR of window rows 0..H+7; 8 outputs per thread and step ----
    {
      constexpr int CH = W / 8, NP = C::NPL, PER = ( H + 8 ) * CH;
      // item order: plane slowest, so that whole waves work on the centre plane, whose horizontal phase is 0 in the half-sample round (and in the quarter-sample
      // round of the PUs whose half-sample winner has no horizontal part): the identity filter (InterpolationFilter.cpp:77-79, m_lumaFilter[0] = {0,0,0,64,0,0,0,0})
      // Half-sample round: the planes dx = -1 and dx = +1 are the SAME filter one integer column apart (qx = -2 -> column offset -1, phase 2; qx = +2 -> offset 0,
      // phase 2): plane(+1)[x] = plane(-1)[x + 1], so one item forms 9 sums from its 16 window samples and stores both planes' 8 outputs (36 dot products
      // instead of 64).  SEQ has one plane buffer and keeps the two passes.
      // Quarter-sample round: its centre plane (qx = 2 * halfX) IS the half-sample round's plane dx = halfX, still in its slot; only dx = -1 / +1 are formed, into the
      // two other slots (plane_slot).  So both rounds run two plane items per (PU, row, chunk).  SEQ has one plane buffer and keeps the three passes.
      const bool pairH = !C::SEQ && round == 0;
      for( int i = tid; i < C::JPW * ( C::SEQ ? NP : 2 ) * PER; i += C::BLOCK )
      {
        const int pp = i / ( C::JPW * PER ), rem = i - pp * C::JPW * PER;
        const int jl = rem / PER, o = rem - jl * PER, r = o / CH, x0 = ( o - r * CH ) * 8;
        if( jl >= nj ) continue;
        const int p = C::SEQ ? seqPl : round == 0 ? pp : 2 * pp;                                  // dx + 1
        const int ps = C::SEQ ? 0 : plane_slot( round, sCentre[jl][0], p - 1 );                    // the slot the plane is stored in
        const vtmhip_frac_job &j = jb[jl];
        const int qx = sCentre[jl][0] + ( p - 1 ) * step, ix = qx >> 2, fx = qx & 3;
        const IfParams pH = if_params( 1, 0, j.bitDepth, 0, ( 1 << j.bitDepth ) - 1, 0 );
        if( fx == 0 )
        {
          // (64 v + offset) >> shift with offset = -(8192 << shift), shift <= 6: (v << (6 - shift)) - 8192, on packed 16-bit words (the truncation to Pel is the
          // arithmetic modulo 2^16 either way); fx == 0 implies ix == 0: output x0 + i is window column x0 + 4 + i
          typedef short v2s __attribute__( ( ext_vector_type( 2 ) ) );
          const int16_t *wr = lds + jl * C::PERJOB + r * C::WLD + x0 + 4;
          const int2     lo = *reinterpret_cast<const int2 *>( wr ), hi = *reinterpret_cast<const int2 *>( wr + 4 );
          const int      in[4] = { lo.x, lo.y, hi.x, hi.y };
          int            ow[4];
          const short    up6 = ( short ) ( 6 - pH.shift );
#pragma unroll
          for( int m = 0; m < 4; m++ )
          {
            v2s v;
            __builtin_memcpy( &v, &in[m], 4 );
            v = ( v << up6 ) - ( short ) 8192;
            __builtin_memcpy( &ow[m], &v, 4 );
          }
          *reinterpret_cast<int4 *>( lds + jl * C::PERJOB + C::WIN + ps * C::PLANE + r * W + x0 ) = make_int4( ow[0], ow[1], ow[2], ow[3] );
          continue;
        }
        const int16_t *cH = ( round == 0 && j.useAltHpelIf && fx == 2 ) ? c_lumaAltHpel : c_lumaFilter[fx << 2];
        int ch[8];
#pragma unroll
        for( int k = 0; k < 8; k++ ) ch[k] = cH[k];
        // output x0+i needs window columns x0 + ix + 1 + i + (0..7): 16 samples from the 16-byte aligned column x0, shifted by ix + 1 in {0, 1}
        const int4 *w4 = reinterpret_cast<const int4 *>( lds + jl * C::PERJOB + r * C::WLD + x0 );
        const int4  a = w4[0], b = w4[1];
        unsigned    d[8] = { ( unsigned ) a.x, ( unsigned ) a.y, ( unsigned ) a.z, ( unsigned ) a.w, ( unsigned ) b.x, ( unsigned ) b.y, ( unsigned ) b.z, ( unsigned ) b.w };
        const unsigned sh = ( unsigned ) ( ix + 1 ) << 4;   // 0 or 16 bits
#pragma unroll
        for( int m = 0; m < 7; m++ ) d[m] = __builtin_amdgcn_alignbit( d[m + 1], d[m], sh );
        d[7] = d[7] >> sh;
        // 8 outputs with v_dot2c_i32_i16: d[m] = (v[2m], v[2m+1]) serves the even outputs, e[m] = (v[2m+1], v[2m+2]) the odd ones
        typedef short v2s __attribute__( ( ext_vector_type( 2 ) ) );
        unsigned e[7];
#pragma unroll
        for( int m = 0; m < 7; m++ ) e[m] = __builtin_amdgcn_alignbit( d[m + 1], d[m], 16 );
        v2s cpk[4];
#pragma unroll
        for( int m = 0; m < 4; m++ ) { cpk[m].x = ( short ) ch[2 * m]; cpk[m].y = ( short ) ch[2 * m + 1]; }
        auto hsum = [&]( int q ) -> unsigned
        {
          int sum = 0;
#pragma unroll
          for( int m = 0; m < 4; m++ )
          {
            const unsigned pw = ( q & 1 ) ? e[( q >> 1 ) + m] : d[( q >> 1 ) + m];
            v2s pv;
            __builtin_memcpy( &pv, &pw, 4 );
            sum = __builtin_amdgcn_sdot2( pv, cpk[m], sum, false );
          }
          return ( unsigned ) ( unsigned short ) if_finish( sum, pH );
        };
        const unsigned h0 = hsum( 0 ), h1 = hsum( 1 ), h2 = hsum( 2 ), h3 = hsum( 3 ), h4 = hsum( 4 ), h5 = hsum( 5 ), h6 = hsum( 6 ), h7 = hsum( 7 );
        int16_t *out = lds + jl * C::PERJOB + C::WIN + r * W + x0;
        *reinterpret_cast<int4 *>( out + ps * C::PLANE ) = make_int4( ( int ) ( h0 | h1 << 16 ), ( int ) ( h2 | h3 << 16 ), ( int ) ( h4 | h5 << 16 ), ( int ) ( h6 | h7 << 16 ) );
        if( pairH )
        {
          const unsigned h8 = hsum( 8 );
          *reinterpret_cast<int4 *>( out + 2 * C::PLANE ) = make_int4( ( int ) ( h1 | h2 << 16 ), ( int ) ( h3 | h4 << 16 ), ( int ) ( h5 | h6 << 16 ), ( int ) ( h7 | h8 << 16 ) );
        }
      }
    }
    __syncthreads();

    // ---- phase V: one (PU, candidate, tile) item per lane ------------------------------------------------------------------
    // SEQ: the candidates whose plane is in the buffer -- dx = -1: table entries 3, 5, 7; dx = 0: 0, 1, 2 (entry 0 only in the half-sample round); dx = +1: 4, 6, 8
    // Item order otherwise: candidate slowest, and the candidates without a vertical offset (table entries 0, 3, 4) first: their vertical phase is 0 in the
    // half-sample round (and in the quarter-sample round of the PUs whose half-sample winner has no vertical part) = the identity filter, so whole waves skip the FIR.
    // PK (every PU of the workgroup: Hadamard cost, bitDepth <= 10): the packed path with the identity shortcut; otherwise the general path.
    auto phaseV = [&]( auto pkTag )
    {
    constexpr bool PK = decltype( pkTag )::value;
    const int seqSkip = ( C::SEQ && round == 1 && seqPl == 1 ) ? 1 : 0, seqN = 3 - seqSkip;
    const int ncand = C::SEQ ? seqN : ( round == 0 ? 9 : 8 );
    constexpr unsigned long long ORDER = 0x876521430ull;   // nibble k: the k-th candidate in evaluation order
#pragma unroll 1
    for( int it = tid; it < ncand * C::JPW * C::TILES; it += C::BLOCK )
    {
      const int ci = it / ( C::JPW * C::TILES ), rem = it - ci * C::JPW * C::TILES, jl = rem / C::TILES, tile = rem - jl * C::TILES;
      if( jl >= nj ) continue;
      const int cand = C::SEQ ? ( seqPl == 1 ? ci + seqSkip : 3 + 2 * ci + ( seqPl >> 1 ) ) : ( int ) ( ( ORDER >> ( 4 * ( ci + round ) ) ) & 15 );
      const vtmhip_frac_job &j = jb[jl];
      const int8_t( *tab )[2] = round == 0 ? c_refineH : c_refineQ;
      const int dx = tab[cand][0], dy = tab[cand][1];
      const int qy = sCentre[jl][1] + dy * step, iy = qy >> 2, fy = qy & 3;
      const int16_t *cV = ( round == 0 && j.useAltHpelIf && fy == 2 ) ? c_lumaAltHpel : c_lumaFilter[fy << 2];
      int cv[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) cv[k] = cV[k];
      // tile order: the long direction of the PU fastest, so that the two 8x8 halves of a 16x8 / 8x16 Hadamard tile are the items of lanes 2k, 2k + 1
      const int      ty = W >= H ? tile / C::TX : tile % C::TY, tx = W >= H ? tile - ty * C::TX : tile / C::TY;
      const int16_t *pl = lds + jl * C::PERJOB + C::WIN + ( C::SEQ ? 0 : plane_slot( round, sCentre[jl][0], dx ) ) * C::PLANE + ( ty * 8 + iy + 1 ) * W + tx * 8;
      const IfParams pV = if_params( 0, 1, j.bitDepth, 0, ( 1 << j.bitDepth ) - 1, 0 );
      // Vertical FIR with v_dot2c_i32_i16: rows r and r + 1 are interleaved column-wise (two v_perm per dword pair), so one instruction
      // applies two taps: output row y takes the row pairs (y, y+1), (y+2, y+3), (y+4, y+5), (y+6, y+7) with the tap pairs
      // (c0,c1) .. (c6,c7) -- 256 dot products instead of 512 multiply-adds, and no 16-bit unpacking.
      typedef short v2s __attribute__( ( ext_vector_type( 2 ) ) );
      // Packed-SATD path (bitDepth <= 10): the taps and the rounding offset are pre-scaled by 2^(16 - shift), so that the UPPER half of an accumulator
      // is the shifted sum ((acc << up) >> 16 == acc >> shift, arithmetic) and one v_perm packs two of them -- no shifts, and the clip runs on
      // packed words.  |taps| <= 58 and 16 - shift <= 8 keep the scaled taps in 16 bits; 112 * 2^up * 32768 + offset * 2^up < 2^31.
      const int  up   = PK ? 16 - pV.shift : 0, accInit = pV.offset << up;
      v2s cpk[4];
#pragma unroll
      for( int m = 0; m < 4; m++ ) { cpk[m].x = ( short ) ( cv[2 * m] << up ); cpk[m].y = ( short ) ( cv[2 * m + 1] << up ); }
      int  acc[64];
      auto fir = [&]()
      {
        int4 prev = *reinterpret_cast<const int4 *>( pl );
#pragma unroll
        for( int r = 1; r < 15; r++ )
        {
          const int4     cur = *reinterpret_cast<const int4 *>( pl + r * W );   // 8 samples of plane row r
          const unsigned pw[4] = { ( unsigned ) prev.x, ( unsigned ) prev.y, ( unsigned ) prev.z, ( unsigned ) prev.w };
          const unsigned cw[4] = { ( unsigned ) cur.x, ( unsigned ) cur.y, ( unsigned ) cur.z, ( unsigned ) cur.w };
          v2s pr[8];   // column x: (row r-1, row r)
#pragma unroll
          for( int k = 0; k < 4; k++ )
          {
            const unsigned lo = __builtin_amdgcn_perm( cw[k], pw[k], 0x05040100u ), hi = __builtin_amdgcn_perm( cw[k], pw[k], 0x07060302u );
            __builtin_memcpy( &pr[2 * k], &lo, 4 );
            __builtin_memcpy( &pr[2 * k + 1], &hi, 4 );
          }
          const int q = r - 1;   // the pair (q, q+1)
#pragma unroll
          for( int m = 0; m < 4; m++ )
          {
            const int y = q - 2 * m;
            if( y >= 0 && y < 8 )
            {
#pragma unroll
              for( int x = 0; x < 8; x++ ) acc[y * 8 + x] = __builtin_amdgcn_sdot2( pr[x], cpk[m], m == 0 ? accInit : acc[y * 8 + x], false );   // m == 0 is a row's first pair
            }
          }
          prev = cur;
          __builtin_amdgcn_sched_barrier( 0 );   // keep the row loads from being hoisted together (register pressure -> occupancy)
        }
      };
      // |sum of taps| <= 112 and |plane sample| <= 32768: (acc >> shift) fits 16 bits, so the reference's Pel truncation is the identity here
      const int16_t *org = orgBase + j.orgOff + ( long ) ( ty * 8 ) * j.orgStride + tx * 8;
      unsigned       d;
      if constexpr( PK )
      {
        // 10-bit prediction and |org| <= 3071 (picture samples or the bi-pred target 2*org - pred): |diff| <= 4095, so the first three
        // butterfly levels run on packed 16-bit words (the reference's own SIMD SATD is 16-bit for bitDepth <= 10, x86/RdCostX86.h:2157);
        // when every lane of the wave sees original samples inside [0, 1023] (|diff| <= 1023), five levels do.
        v2s P[8][4];   // the shifted sums before the clip, packed
        if( fy == 0 )
        {
          // m_lumaFilter[0] = {0,0,0,64,0,0,0,0}: output row y is plane row y + 3 of the item's 15.  (64 t + offset) >> shift with offset = 2^(shift - 1) + (8192 << 6),
          // shift >= 8: (t + offset / 64) >> (shift - 6); |t| <= 13 762 for 10-bit samples (first pass minus 8192), so t + offset / 64 stays inside 16 bits
          const v2s addv = { ( short ) ( pV.offset >> 6 ), ( short ) ( pV.offset >> 6 ) }, shv = { ( short ) ( pV.shift - 6 ), ( short ) ( pV.shift - 6 ) };
#pragma unroll
          for( int y = 0; y < 8; y++ )
          {
            const int4 t4 = *reinterpret_cast<const int4 *>( pl + ( y + 3 ) * W );
            const int  tw[4] = { t4.x, t4.y, t4.z, t4.w };
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
              v2s t;
              __builtin_memcpy( &t, &tw[k], 4 );
              P[y][k] = ( t + addv ) >> shv;
            }
          }
        }
        else
        {
          fir();
#pragma unroll
          for( int y = 0; y < 8; y++ )
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
              const unsigned pw = __builtin_amdgcn_perm( ( unsigned ) acc[y * 8 + 2 * k + 1], ( unsigned ) acc[y * 8 + 2 * k], 0x07060302u );   // the two upper halves
              __builtin_memcpy( &P[y][k], &pw, 4 );
            }
        }
        v2s       D[8][4];
        const v2s zero = { 0, 0 }, cmaxv = { ( short ) pV.cmax, ( short ) pV.cmax };
        unsigned  wide = 0;
#pragma unroll
        for( int y = 0; y < 8; y++ )
        {
          const Pel8u o = *reinterpret_cast<const Pel8u *>( org + ( long ) y * j.orgStride );
#pragma unroll
          for( int k = 0; k < 4; k++ )
          {
            v2s ov;
            __builtin_memcpy( &ov, &o.v[k], 4 );
            D[y][k] = ov - __builtin_elementwise_min( __builtin_elementwise_max( P[y][k], zero ), cmaxv );
            wide |= o.v[k];
          }
        }
        if( C::PAIR ) d = __all( ( wide & 0xfc00fc00u ) == 0 ) ? satd8_pair_packed10( D, ( tile & 1 ) != 0 ) : satd8_pair_packed( D );
        else          d = __all( ( wide & 0xfc00fc00u ) == 0 ) ? satd8_packed10( D ) : satd8_packed( D );
      }
      else if( j.useHad )
      {
        fir();
#pragma unroll
        for( int y = 0; y < 8; y++ )
        {
          const Pel8u o = *reinterpret_cast<const Pel8u *>( org + ( long ) y * j.orgStride );
#pragma unroll
          for( int x = 0; x < 8; x++ )
          {
            const int ov   = ( x & 1 ) ? ( int ) o.v[x >> 1] >> 16 : ( int ) ( short ) o.v[x >> 1];
            acc[y * 8 + x] = ov - min( pV.cmax, max( 0, acc[y * 8 + x] >> pV.shift ) );
          }
        }
#pragma unroll
        for( int y = 0; y < 8; y++ ) wht1d<8, 1>( acc + y * 8 );
        if( C::PAIR )
        {
#pragma unroll
          for( int x = 0; x < 8; x++ ) wht1d<8, 8>( acc + x );
          d = satd8_pair_finish32( acc );
        }
        else
        {
#pragma unroll
          for( int x = 0; x < 8; x++ ) wht1d<8, 8, false>( acc + x );
          int t = 0;   // last level + |.| + sum: |a + b| + |a - b| = 2 max(|a|, |b|)
#pragma unroll
          for( int i = 0; i < 32; i++ ) t += max( abs( acc[i] ), abs( acc[i + 32] ) );
          t <<= 1;
          const int dc = abs( acc[0] + acc[32] );
          d            = ( unsigned ) ( ( t - dc + ( dc >> 2 ) + 2 ) >> 2 );
        }
      }
      else
      {
        fir();
        unsigned t = 0;
#pragma unroll
        for( int y = 0; y < 8; y++ )
        {
          const Pel8u o = *reinterpret_cast<const Pel8u *>( org + ( long ) y * j.orgStride );
#pragma unroll
          for( int x = 0; x < 8; x++ )
          {
            const int ov = ( x & 1 ) ? ( int ) o.v[x >> 1] >> 16 : ( int ) ( short ) o.v[x >> 1];
            t += ( unsigned ) abs( ov - min( pV.cmax, max( 0, acc[y * 8 + x] >> pV.shift ) ) );
          }
        }
        d = t;
      }
      if( !( C::PAIR && j.useHad ) || ( tile & 1 ) == 0 ) atomicAdd( &sCost[jl * 16 + cand], d );   // a Hadamard pair's value is the same in both lanes: the even one adds it
    }
    };
    if( pkAll ) phaseV( std::true_type{} ); else phaseV( std::false_type{} );
    __syncthreads();
    }   // planes

    // ---- select: first strict minimum of distortion + MV rate in table order ------------------------------------------------
    if( tid < nj )
    {
      const vtmhip_frac_job &j = jb[tid];
      const int8_t( *tab )[2] = round == 0 ? c_refineH : c_refineQ;
      const int costScale = round == 0 ? 1 : 0;
      const int bx = round == 0 ? ( j.intX << 1 ) : ( ( ( j.intX << 1 ) + ( sCentre[tid][0] >> 1 ) ) << 1 );
      const int by = round == 0 ? ( j.intY << 1 ) : ( ( ( j.intY << 1 ) + ( sCentre[tid][1] >> 1 ) ) << 1 );
      unsigned long long best = ~0ull;
      int                bi   = 0;
      for( int i = 0; i < 9; i++ )
      {
        const unsigned long long c = ( unsigned long long ) sCost[tid * 16 + i] + mv_cost( j.motionLambda, j.predHor, j.predVer, costScale, bx + tab[i][0], by + tab[i][1] );
        if( c < best ) { best = c; bi = i; }
      }
      vtmhip_frac_result *res = results + job0 + tid;
      if( round == 0 )
      {
        res->halfX = tab[bi][0]; res->halfY = tab[bi][1]; res->qterX = 0; res->qterY = 0; res->cost = best;
        sCentre[tid][0] = tab[bi][0] * 2;   // quarter-sample units
        sCentre[tid][1] = tab[bi][1] * 2;
        sKeep[tid]      = sCost[tid * 16 + bi];
      }
      else
      {
        res->qterX = tab[bi][0]; res->qterY = tab[bi][1]; res->cost = best;
      }
      if( fu.me && ( round == 1 || j.imvShift != 0 ) )      // the last round: the row's final record (the rate re-weighting mest_final_kernel would have done)
      {
        vtmhip_frac_result f;
        f.halfX = ( int16_t ) ( round == 0 ? tab[bi][0] : sCentre[tid][0] >> 1 ); f.halfY = ( int16_t ) ( round == 0 ? tab[bi][1] : sCentre[tid][1] >> 1 );
        f.qterX = ( int16_t ) ( round == 0 ? 0 : tab[bi][0] ); f.qterY = ( int16_t ) ( round == 0 ? 0 : tab[bi][1] ); f.cost = best;
        const vtmhip_me_result r = fu.ires[job0 + tid];
        vtmhip_me_out          o;
        mg::make_out_frac( fu.me[job0 + tid], r.mvX, r.mvY, r.dist, f, o );
        fu.out[job0 + tid] = o;
      }
    }
    __syncthreads();
  }
}

size_t frac_lds_bytes( int maxW, int maxH ) { return ( size_t ) ( ( maxH + 8 ) * ( maxW + 8 ) + ( maxH + 8 ) * maxW + maxH * maxW ) * sizeof( int16_t ); }

int if_single( vtmhip_ctx *ctx, int vertical, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w,
               int h, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMC )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, src && dst && ( taps == 0 || coeff ), "null pointer" );
  VTMHIP_REQUIRE( ctx, w >= 1 && h >= 1 && w <= 256 && h <= 256, "block size" );
  VTMHIP_REQUIRE( ctx, taps == 0 || taps == 2 || taps == 4 || taps == 8, "taps must be 8, 4, 2 or 0 (copy)" );
  VTMHIP_REQUIRE( ctx, bitDepth >= 8 && bitDepth <= 14, "bitDepth" );
  // stage the source region the filter touches: (taps/2 - 1) samples before, taps/2 after, along the filter direction
  const int before = taps ? taps / 2 - 1 : 0, after = taps ? taps / 2 : 0;
  const int sw = w + ( vertical ? 0 : before + after ), sh = h + ( vertical ? before + after : 0 );
  const size_t srcBytes = ( size_t ) sw * sh * 2, dstBytes = ( size_t ) w * h * 2;
  const size_t dstOffB = ( srcBytes + 63 ) & ~( size_t ) 63, jobOffB = ( dstOffB + dstBytes + 63 ) & ~( size_t ) 63;
  int st = vtmhip_internal_scratch( ctx, jobOffB + sizeof( vtmhip_if_job ) );
  if( st ) return st;
  char          *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  const int16_t *s0 = src - ( vertical ? ( ptrdiff_t ) before * srcStride : before );
  for( int y = 0; y < sh; y++ ) memcpy( hp + ( size_t ) y * sw * 2, s0 + ( ptrdiff_t ) y * srcStride, ( size_t ) sw * 2 );
  vtmhip_if_job j;
  memset( &j, 0, sizeof( j ) );
  j.srcOff = vertical ? ( int64_t ) before * sw : before;
  j.dstOff = 0; j.srcStride = sw; j.dstStride = w; j.width = ( int16_t ) w; j.height = ( int16_t ) h;
  j.vertical = ( uint8_t ) vertical; j.taps = ( uint8_t ) taps; j.isFirst = ( uint8_t ) isFirst; j.isLast = ( uint8_t ) isLast;
  for( int k = 0; k < taps; k++ ) j.coeff[k] = coeff[k];
  j.clipMin = ( int16_t ) clipMin; j.clipMax = ( int16_t ) clipMax; j.bitDepth = ( uint8_t ) bitDepth; j.biMCForDMVR = ( uint8_t ) biMC;
  memcpy( hp + jobOffB, &j, sizeof( j ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, srcBytes, hipMemcpyHostToDevice, ctx->stream ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp + jobOffB, hp + jobOffB, sizeof( j ), hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( if_batch_kernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ( const int16_t * ) dp, ( int16_t * ) ( dp + dstOffB ),
                      ( const vtmhip_if_job * ) ( dp + jobOffB ), 1 );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + dstOffB, dp + dstOffB, dstBytes, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  for( int y = 0; y < h; y++ ) memcpy( dst + ( ptrdiff_t ) y * dstStride, hp + dstOffB + ( size_t ) y * w * 2, ( size_t ) w * 2 );
  return VTMHIP_OK;
}

template<int W, int H>
int launch_frac_sq( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_frac_job *d_jobs, int n, vtmhip_frac_result *d_results, const FracFuse &fu )
{
  using C = FracSq<W, H>;
  if( C::LDS > 64 * 1024 )
    VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( frac_search_sq_kernel<W, H> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) C::LDS ) );
  VTMHIP_TIME_KERNEL( ctx, "frac_search_sq_kernel" );
  hipLaunchKernelGGL( ( frac_search_sq_kernel<W, H> ), dim3( ( n + C::JPW - 1 ) / C::JPW ), dim3( C::BLOCK ), C::LDS, ctx->stream, d_orgBase, d_refBase, d_jobs, n,
                      d_results, fu );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // namespace

extern "C"
{

int vtmhip_filterHor( vtmhip_ctx *ctx, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width,
                      int height, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  return if_single( ctx, 0, taps, isFirst, isLast, src, srcStride, dst, dstStride, width, height, coeff, bitDepth, clipMin, clipMax, biMCForDMVR );
}

int vtmhip_filterVer( vtmhip_ctx *ctx, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width,
                      int height, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  return if_single( ctx, 1, taps, isFirst, isLast, src, srcStride, dst, dstStride, width, height, coeff, bitDepth, clipMin, clipMax, biMCForDMVR );
}

int vtmhip_filterCopy( vtmhip_ctx *ctx, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height,
                       int bitDepth, int clipMin, int clipMax, int biMCForDMVR )
{
  return if_single( ctx, 0, 0, isFirst, isLast, src, srcStride, dst, dstStride, width, height, nullptr, bitDepth, clipMin, clipMax, biMCForDMVR );
}

int vtmhip_if_batch_dev( vtmhip_ctx *ctx, const int16_t *d_srcBase, int16_t *d_dstBase, const vtmhip_if_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_srcBase && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( if_batch_kernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_srcBase, d_dstBase, d_jobs, n );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_frac_search_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_frac_job *d_jobs, int n,
                                  int maxWidth, int maxHeight, int uniformSquare, vtmhip_frac_result *d_results )
{
  return vtmhip_internal_frac_search( ctx, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, uniformSquare, d_results, nullptr );
}

}   // extern "C"

// fuse != nullptr: the fractional searches of the xMotionEstimation rows fuse->me after their integer stage fuse->ires; the rows' final records go to fuse->out (d_jobs unused)
int vtmhip_internal_frac_search( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_frac_job *d_jobs, int n, int maxWidth, int maxHeight,
                                 int uniformSquare, vtmhip_frac_result *d_results, const MeFuse *fuse )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && ( d_jobs || fuse ) && d_results, "null pointer" );
  FracFuse fu; memset( &fu, 0, sizeof( fu ) );
  if( fuse ) { fu.me = fuse->me; fu.ires = fuse->ires; fu.out = fuse->out; fu.useHadME = fuse->cfg.useHadME; fu.bitDepth = fuse->bitDepth; fu.patIsOther = fuse->patIsOther; }
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxWidth <= 128 && maxHeight >= 4 && maxHeight <= 128, "maxWidth / maxHeight" );
  if( uniformSquare )
  {
    // caller's promise: every job is exactly maxWidth x maxHeight and all jobs share imvShift -> tiled fast path (squares and the split shapes)
#define VTMHIP_FRAC_CASE( WW, HH ) case ( WW ) * 256 + ( HH ): return launch_frac_sq<WW, HH>( ctx, d_orgBase, d_refBase, d_jobs, n, d_results, fu );
    switch( maxWidth * 256 + maxHeight )
    {
      VTMHIP_FRAC_CASE( 8, 8 ) VTMHIP_FRAC_CASE( 16, 16 ) VTMHIP_FRAC_CASE( 32, 32 ) VTMHIP_FRAC_CASE( 64, 64 ) VTMHIP_FRAC_CASE( 128, 128 )
      VTMHIP_FRAC_CASE( 16, 8 ) VTMHIP_FRAC_CASE( 8, 16 ) VTMHIP_FRAC_CASE( 32, 8 ) VTMHIP_FRAC_CASE( 8, 32 ) VTMHIP_FRAC_CASE( 32, 16 ) VTMHIP_FRAC_CASE( 16, 32 )
      VTMHIP_FRAC_CASE( 64, 16 ) VTMHIP_FRAC_CASE( 16, 64 ) VTMHIP_FRAC_CASE( 64, 32 ) VTMHIP_FRAC_CASE( 32, 64 )
    default: break;   // other shapes: generic kernel below
    }
#undef VTMHIP_FRAC_CASE
  }
  const size_t lds = frac_lds_bytes( maxWidth, maxHeight );
  if( lds > 64 * 1024 )
    VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( frac_search_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
  VTMHIP_TIME_KERNEL( ctx, "frac_search_kernel" );
  hipLaunchKernelGGL( frac_search_kernel, dim3( n ), dim3( 64 ), lds, ctx->stream, d_orgBase, d_refBase, d_jobs, d_results, maxWidth, maxHeight, fu );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}
