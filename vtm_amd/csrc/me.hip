// me.hip -- integer motion search on the device: one wave (64 lanes) runs one complete InterSearch::xTZSearch.
//
// Reference: EncoderLib/InterSearch.cpp xTZSearch :3640-3976, xTZSearchHelp :330-419, xTZ2PointSearch :422-447,
// xTZ8PointDiamondSearch :504-705, xSetSearchRange :3496-3563, clipMv CommonLib/Mv.cpp:56-74,
// MV rate CommonLib/RdCost.h:301-315.
//
// The reference evaluates one candidate per distFunc call and updates the best point after each.  Its accept rule
// (strict '<' in evaluation order, cost = SAD + MV rate) makes every search round equivalent to
//     (cmin, imin) = lexicographic min over the round's ordered candidate list of (cost, index);
//     if cmin < best: best = cmin, bestX/Y/dist/pointNr = those of candidate imin, bestRound = 0
// so a round is: generate the list (wave-uniform), evaluate all candidates in parallel, one wave arg-min.
// The sequential dependence that remains is round -> round, which is why a job owns a wave and a launch carries
// thousands of jobs (all PUs x reference pictures of a picture).
#include "ctx.hpp"

namespace
{

struct Range { int left, right, top, bottom; };

struct MeJob   // wave-uniform view of one vtmhip_tz_job
{
  const int16_t *org;
  const int16_t *ref;
  int            orgStride, refStride, w, h, ss;
  unsigned       imvShift;
  int            predHor, predVer, costScale;
  double         lambda;
  int            horMin, horMax, verMin, verMax;   // clipMv limits (internal 1/16 precision)
};

__device__ __forceinline__ int floor_log2_u( unsigned v ) { return 31 - __clz( ( int ) v ); }

__device__ __forceinline__ unsigned eg_bits( int v )
{
  unsigned len = 1;
  unsigned t   = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  while( t > 128 ) { len += 14; t >>= 7; }
  return len + ( ( unsigned ) floor_log2_u( t ) << 1 );
}

__device__ __forceinline__ unsigned long long mv_cost( const MeJob &j, int x, int y )
{
  const unsigned bits = eg_bits( ( ( x << j.costScale ) - j.predHor ) >> j.imvShift ) + eg_bits( ( ( y << j.costScale ) - j.predVer ) >> j.imvShift );
  return ( unsigned long long ) ( j.lambda * ( double ) bits );   // fp64 multiply, truncation (RdCost.h:314)
}

__device__ __forceinline__ void clip_mv( const MeJob &j, int &hor, int &ver )
{
  hor = min( j.horMax, max( j.horMin, hor ) );
  ver = min( j.verMax, max( j.verMin, ver ) );
}
__device__ __forceinline__ int div_pow2( int v, int i ) { return ( v + ( 1 << ( i - 1 ) ) - ( v >= 0 ? 1 : 0 ) ) >> i; }
__device__ __forceinline__ int prec_down( int v, int rs ) { const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }

__device__ __forceinline__ Range search_range( const MeJob &j, int predHor, int predVer, int range )
{
  clip_mv( j, predHor, predVer );
  int l = predHor - ( range << 4 ), t = predVer - ( range << 4 ), r = predHor + ( range << 4 ), b = predVer + ( range << 4 );
  clip_mv( j, l, t );
  clip_mv( j, r, b );
  Range sr;
  sr.left = div_pow2( l, 4 ); sr.top = div_pow2( t, 4 ); sr.right = div_pow2( r, 4 ); sr.bottom = div_pow2( b, 4 );
  return sr;
}

// 8 consecutive samples from a 2-byte aligned address (gfx950 global memory handles the misalignment in hardware)
struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8 { int16_t v[8]; };
struct __attribute__( ( packed, aligned( 2 ) ) ) Pel4 { int16_t v[4]; };

// full SAD of one candidate by ONE lane (rows stepped by 1 << ss, result shifted back) -- RdCost.cpp:493-528
__device__ __forceinline__ unsigned sad_lane( const MeJob &j, int cx, int cy )
{
  const int16_t *o    = j.org;
  const int16_t *c    = j.ref + ( long ) cy * j.refStride + cx;
  const int      step = 1 << j.ss;
  unsigned       s    = 0;
  if( ( j.w & 7 ) == 0 )
  {
    for( int y = 0; y < j.h; y += step )
    {
      for( int x = 0; x < j.w; x += 8 )
      {
        const Pel8 a = *reinterpret_cast<const Pel8 *>( o + x );
        const Pel8 b = *reinterpret_cast<const Pel8 *>( c + x );
#pragma unroll
        for( int k = 0; k < 8; k++ ) s += ( unsigned ) abs( ( int ) a.v[k] - ( int ) b.v[k] );
      }
      o += ( long ) j.orgStride * step;
      c += ( long ) j.refStride * step;
    }
  }
  else
  {
    for( int y = 0; y < j.h; y += step )
    {
      for( int x = 0; x < j.w; x += 4 )
      {
        const Pel4 a = *reinterpret_cast<const Pel4 *>( o + x );
        const Pel4 b = *reinterpret_cast<const Pel4 *>( c + x );
#pragma unroll
        for( int k = 0; k < 4; k++ ) s += ( unsigned ) abs( ( int ) a.v[k] - ( int ) b.v[k] );
      }
      o += ( long ) j.orgStride * step;
      c += ( long ) j.refStride * step;
    }
  }
  return s << j.ss;
}

// lexicographic (cost, index) minimum over the wave
__device__ __forceinline__ void wave_argmin( unsigned long long &cost, unsigned &idx )
{
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 )
  {
    const unsigned long long oc = __shfl_xor( cost, o, 64 );
    const unsigned           oi = __shfl_xor( idx, o, 64 );
    if( oc < cost || ( oc == cost && oi < idx ) ) { cost = oc; idx = oi; }
  }
}

struct TzState
{
  Range              sr;
  unsigned long long bestSad;
  int                bestX, bestY, pointNr;
  unsigned           bestDist, bestRound, nEval;
};

// evaluate the n (<= 64) candidates whose (x, y, nr, dist) sit in pts[] and replay the accept rule
__device__ __forceinline__ void tz_round( const MeJob &j, TzState &s, const int4 *pts, int n, int lane, bool touchMeta )
{
  unsigned long long cost = ~0ull;
  unsigned           idx  = 0xffffffffu;
  // pts[] was written by lane 0 of this wave: DS operations of one wave execute in order; the fence keeps the
  // compiler from moving the reads above the writes
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  __builtin_amdgcn_wave_barrier();
  if( lane < n )
  {
    const int4 p = pts[lane];
    cost         = ( unsigned long long ) sad_lane( j, p.x, p.y ) + mv_cost( j, p.x, p.y );
    idx          = ( unsigned ) lane;
  }
  wave_argmin( cost, idx );
  s.nEval += ( unsigned ) n;
  if( n > 0 && cost < s.bestSad )
  {
    const int4 p = pts[idx];
    s.bestSad    = cost;
    s.bestX      = p.x;
    s.bestY      = p.y;
    if( touchMeta )
    {
      s.bestDist  = ( unsigned ) p.w;
      s.bestRound = 0;
      s.pointNr   = p.z;
    }
  }
}

#define PUSH( X, Y, NR, D ) do { if( lane == 0 ) pts[n] = make_int4( ( X ), ( Y ), ( NR ), ( D ) ); n++; } while( 0 )

// ordered candidate list of one diamond round (xTZ8PointDiamondSearch :504-705); returns the count (<= 16)
__device__ int diamond_points( const Range &sr, int sx, int sy, int d, bool corners, int4 *pts, int lane )
{
  int       n = 0;
  const int top = sy - d, bot = sy + d, left = sx - d, right = sx + d;
  if( d == 1 )
  {
    if( top >= sr.top )
    {
      if( corners && left >= sr.left ) PUSH( left, top, 1, d );
      PUSH( sx, top, 2, d );
      if( corners && right <= sr.right ) PUSH( right, top, 3, d );
    }
    if( left >= sr.left ) PUSH( left, sy, 4, d );
    if( right <= sr.right ) PUSH( right, sy, 5, d );
    if( bot <= sr.bottom )
    {
      if( corners && left >= sr.left ) PUSH( left, bot, 6, d );
      PUSH( sx, bot, 7, d );
      if( corners && right <= sr.right ) PUSH( right, bot, 8, d );
    }
  }
  else if( d <= 8 )
  {
    const int h2 = d >> 1, top2 = sy - h2, bot2 = sy + h2, left2 = sx - h2, right2 = sx + h2;
    if( top >= sr.top && left >= sr.left && right <= sr.right && bot <= sr.bottom )
    {
      PUSH( sx, top, 2, d );
      PUSH( left2, top2, 1, h2 );
      PUSH( right2, top2, 3, h2 );
      PUSH( left, sy, 4, d );
      PUSH( right, sy, 5, d );
      PUSH( left2, bot2, 6, h2 );
      PUSH( right2, bot2, 8, h2 );
      PUSH( sx, bot, 7, d );
    }
    else
    {
      if( top >= sr.top ) PUSH( sx, top, 2, d );
      if( top2 >= sr.top )
      {
        if( left2 >= sr.left ) PUSH( left2, top2, 1, h2 );
        if( right2 <= sr.right ) PUSH( right2, top2, 3, h2 );
      }
      if( left >= sr.left ) PUSH( left, sy, 4, d );
      if( right <= sr.right ) PUSH( right, sy, 5, d );
      if( bot2 <= sr.bottom )
      {
        if( left2 >= sr.left ) PUSH( left2, bot2, 6, h2 );
        if( right2 <= sr.right ) PUSH( right2, bot2, 8, h2 );
      }
      if( bot <= sr.bottom ) PUSH( sx, bot, 7, d );
    }
  }
  else
  {
    const int  q      = d >> 2;
    const bool inside = top >= sr.top && left >= sr.left && right <= sr.right && bot <= sr.bottom;
    if( inside || top >= sr.top ) PUSH( sx, top, 0, d );
    if( inside || left >= sr.left ) PUSH( left, sy, 0, d );
    if( inside || right <= sr.right ) PUSH( right, sy, 0, d );
    if( inside || bot <= sr.bottom ) PUSH( sx, bot, 0, d );
    for( int i = 1; i < 4; i++ )
    {
      const int yt = top + q * i, yb = bot - q * i, xl = sx - q * i, xr = sx + q * i;
      if( inside || yt >= sr.top )
      {
        if( inside || xl >= sr.left ) PUSH( xl, yt, 0, d );
        if( inside || xr <= sr.right ) PUSH( xr, yt, 0, d );
      }
      if( inside || yb <= sr.bottom )
      {
        if( inside || xl >= sr.left ) PUSH( xl, yb, 0, d );
        if( inside || xr <= sr.right ) PUSH( xr, yb, 0, d );
      }
    }
  }
  return n;
}

__device__ __forceinline__ void tz_diamond( const MeJob &j, TzState &s, int sx, int sy, int d, bool corners, int4 *pts, int lane )
{
  const int n = diamond_points( s.sr, sx, sy, d, corners, pts, lane );
  s.bestRound += 1;
  tz_round( j, s, pts, n, lane, true );
}

__device__ __forceinline__ void tz_two_point( const MeJob &j, TzState &s, int4 *pts, int lane )
{
  // untested neighbours of the best point, by the point number of the dist-1 round (xTZ2PointSearch :426-446);
  // packed as 2-bit fields (value + 1) per point number 0..8
  const int xo0[9] = { 0, -1, -1, 0, -1, +1, -1, -1, +1 }, xo1[9] = { 0, 0, +1, +1, -1, +1, 0, +1, 0 };
  const int yo0[9] = { 0, 0, -1, -1, +1, -1, 0, +1, 0 }, yo1[9] = { 0, -1, -1, 0, -1, +1, +1, +1, +1 };
  int       ax = 0, ay = 0, bx = 0, by = 0;
#pragma unroll
  for( int k = 0; k < 9; k++ )
    if( k == s.pointNr ) { ax = xo0[k]; ay = yo0[k]; bx = xo1[k]; by = yo1[k]; }
  const int x1 = s.bestX + ax, y1 = s.bestY + ay, x2 = s.bestX + bx, y2 = s.bestY + by;
  int       n  = 0;
  if( x1 >= s.sr.left && x1 <= s.sr.right && y1 >= s.sr.top && y1 <= s.sr.bottom ) PUSH( x1, y1, 0, 2 );
  if( x2 >= s.sr.left && x2 <= s.sr.right && y2 >= s.sr.top && y2 <= s.sr.bottom ) PUSH( x2, y2, 0, 2 );
  tz_round( j, s, pts, n, lane, true );
}

// raster scan (xTZSearch :3888-3899 / adaptive :3883-3903): candidate k = (row k / nx, column k % nx), one per lane,
// 64 at a time; per-lane running minimum keeps the earliest index, the wave arg-min the earliest lane.
__device__ __forceinline__ void tz_raster( const MeJob &j, TzState &s, const Range &r, int stepXY, int lane )
{
  const int nx = r.right >= r.left ? ( r.right - r.left ) / stepXY + 1 : 0;
  const int ny = r.bottom >= r.top ? ( r.bottom - r.top ) / stepXY + 1 : 0;
  const int total = nx * ny;
  unsigned long long cost = ~0ull;
  unsigned           idx  = 0xffffffffu;
  for( int k = lane; k < total; k += 64 )
  {
    const int                ry = k / nx, rx = k - ry * nx;
    const int                x = r.left + rx * stepXY, y = r.top + ry * stepXY;
    const unsigned long long c = ( unsigned long long ) sad_lane( j, x, y ) + mv_cost( j, x, y );
    if( c < cost ) { cost = c; idx = ( unsigned ) k; }
  }
  wave_argmin( cost, idx );
  s.nEval += ( unsigned ) total;
  if( total > 0 && cost < s.bestSad )
  {
    const int ry = ( int ) idx / nx, rx = ( int ) idx - ry * nx;
    s.bestSad   = cost;
    s.bestX     = r.left + rx * stepXY;
    s.bestY     = r.top + ry * stepXY;
    s.bestDist  = ( unsigned ) stepXY;
    s.bestRound = 0;
    s.pointNr   = 0;
  }
}

constexpr int TZ_WAVES = 4;

__global__ __launch_bounds__( 64 * TZ_WAVES ) void tz_search_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase,
                                                                    const int16_t *__restrict__ refBase, const vtmhip_tz_job *__restrict__ jobs, int numJobs,
                                                                    vtmhip_me_result *__restrict__ results )
{
  __shared__ int4 sPts[TZ_WAVES][16];
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int       jobIdx = blockIdx.x * TZ_WAVES + wave;
  if( jobIdx >= numJobs ) return;
  const vtmhip_tz_job *jp  = jobs + jobIdx;
  int4                *pts = sPts[wave];

  MeJob j;
  j.org       = orgBase + jp->orgOff;
  j.ref       = refBase + jp->refOff;
  j.orgStride = jp->orgStride;
  j.refStride = jp->refStride;
  j.w         = jp->width;
  j.h         = jp->height;
  j.ss        = jp->subShift;
  j.imvShift  = ( unsigned ) jp->imvShift;
  j.predHor   = jp->predHor;
  j.predVer   = jp->predVer;
  j.costScale = 2;   // m_pcRdCost->setCostScale(2) for the integer search (InterSearch.cpp:3374)
  j.lambda    = jp->motionLambda;
  j.horMax    = ( pic.picW + 8 - jp->puX - 1 ) << 4;
  j.horMin    = ( -pic.ctuSize - 8 - jp->puX + 1 ) << 4;
  j.verMax    = ( pic.picH + 8 - jp->puY - 1 ) << 4;
  j.verMin    = ( -pic.ctuSize - 8 - jp->puY + 1 ) << 4;

  const bool ext = jp->extendedSettings != 0, fast = jp->fastSettings != 0, firstStop = jp->firstSearchStop != 0;
  const int  iRaster = fast ? 8 : 5, searchRange = jp->searchRange;

  TzState s;
  s.bestSad = ~0ull; s.bestX = 0; s.bestY = 0; s.pointNr = 0; s.bestDist = 0; s.bestRound = 0; s.nEval = 0;
  s.sr.left = s.sr.right = s.sr.top = s.sr.bottom = 0;

  // start vector (:3675-3687)
  int mx = jp->mvHor, my = jp->mvVer;
  clip_mv( j, mx, my );
  mx = div_pow2( prec_down( mx, 2 ), 2 );
  my = div_pow2( prec_down( my, 2 ), 2 );
  int n0 = 0;
  {
    int n = 0;
    PUSH( mx, my, 0, 0 );
    n0 = n;
  }
  tz_round( j, s, pts, n0, lane, true );
  if( !fast && ( mx != 0 || my != 0 ) && ( s.bestX != 0 || s.bestY != 0 ) )
  {
    int n = 0;
    PUSH( 0, 0, 0, 0 );
    tz_round( j, s, pts, n, lane, true );
  }
  if( jp->hasIntMv2Nx2NPred )
  {
    int ix = jp->intMv2Nx2NPredHor << 4, iy = jp->intMv2Nx2NPredVer << 4;
    clip_mv( j, ix, iy );
    ix = div_pow2( prec_down( ix, 2 ), 2 );
    iy = div_pow2( prec_down( iy, 2 ), 2 );
    if( ( mx != ix || my != iy ) && ( ix != s.bestX || iy != s.bestY ) )
    {
      int n = 0;
      PUSH( ix, iy, 0, 0 );
      tz_round( j, s, pts, n, lane, true );
    }
  }
  {
    // m_uniMvList start candidates (:3725-3762): one parallel round, same first-strict-minimum semantics
    int       n  = 0;
    const int ne = min( jp->numExtraStart, 15 );
    for( int i = 0; i < ne; i++ )
    {
      int ex = jp->extraStart[i][0], ey = jp->extraStart[i][1];
      clip_mv( j, ex, ey );
      PUSH( prec_down( ex, 4 ), prec_down( ey, 4 ), 0, 0 );
    }
    tz_round( j, s, pts, n, lane, false );
  }

  s.sr = search_range( j, s.bestX << 4, s.bestY << 4, searchRange >> ( fast ? 1 : 0 ) );

  int        startX = s.bestX, startY = s.bestY;
  const bool bestCandidateZero = ( s.bestX == 0 && s.bestY == 0 );

  for( int d = 1; d <= searchRange; d *= 2 )
  {
    tz_diamond( j, s, startX, startY, d, ext, pts, lane );
    if( firstStop && s.bestRound >= 3 ) break;
  }
  if( ext && !bestCandidateZero )
  {
    for( int d = 1; d <= ( searchRange >> 1 ); d *= 2 ) tz_diamond( j, s, 0, 0, d, false, pts, lane );
  }
  if( s.bestDist == 1 )
  {
    s.bestDist = 0;
    tz_two_point( j, s, pts, lane );
  }
  if( ext )
  {
    int   win = iRaster;
    Range lsr = s.sr;
    if( !( ( int ) s.bestDist >= iRaster ) )
    {
      win++;
      lsr.left /= 2; lsr.right /= 2; lsr.top /= 2; lsr.bottom /= 2;
    }
    s.bestDist = ( unsigned ) win;
    tz_raster( j, s, lsr, win, lane );
  }
  else if( ( int ) s.bestDist >= iRaster )
  {
    s.bestDist = ( unsigned ) iRaster;
    tz_raster( j, s, s.sr, iRaster, lane );
  }
  // star refinement (:3937-3971)
  while( s.bestDist > 0 )
  {
    startX     = s.bestX;
    startY     = s.bestY;
    s.bestDist = 0;
    s.pointNr  = 0;
    for( int d = 1; d < searchRange + 1; d *= 2 )
    {
      tz_diamond( j, s, startX, startY, d, ext, pts, lane );
      if( fast && s.bestRound >= 2 ) break;
    }
    if( s.bestDist == 1 )
    {
      s.bestDist = 0;
      if( s.pointNr != 0 ) tz_two_point( j, s, pts, lane );
    }
  }

  if( lane == 0 )
  {
    vtmhip_me_result r;
    r.mvX = s.bestX; r.mvY = s.bestY; r.nEval = s.nEval; r.reserved = 0;
    r.cost = s.bestSad;
    r.dist = s.bestSad - mv_cost( j, s.bestX, s.bestY );
    results[jobIdx] = r;
  }
}

}   // namespace

extern "C" int vtmhip_tz_search_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                           const vtmhip_tz_job *d_jobs, int n, vtmhip_me_result *d_results )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "pic / n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && d_jobs && d_results, "null pointer" );
  VTMHIP_REQUIRE( ctx, pic->picW > 0 && pic->picH > 0 && pic->ctuSize > 0, "picture parameters" );
  hipLaunchKernelGGL( tz_search_kernel, dim3( ( n + TZ_WAVES - 1 ) / TZ_WAVES ), dim3( 64 * TZ_WAVES ), 0, ctx->stream, *pic, d_orgBase, d_refBase,
                      d_jobs, n, d_results );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}
