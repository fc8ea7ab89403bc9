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
#include "mest_glue.hpp"

#include <cstdlib>

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
  // cooperative SAD decomposition: a candidate = `items` segments of `seg` samples (rows stepped by 1 << ss);
  // `lpc` lanes (power of two) share one candidate, 64 / lpc candidates are evaluated at a time
  int            seg, segsPerRow, items, lpc;
  int            lpcShift, sprShift;   // log2(lpc); log2(segsPerRow) or -1 when it is not a power of two (widths 12, 24, 48)
  bool           narrow;               // lambda * 126 < 2^31: distortion + MV rate fits 32 bits -> packed (cost, index) keys
  bool           tiny;                 // distortion + MV rate < 2^26 whatever the candidate: (cost << 6 | index) is ONE 32-bit key for rounds of at most 64 candidates
  unsigned       bias;   // 0x80008000 when samples may be negative (v_sad_u16 is unsigned), else 0
  // items == lpc (blocks up to 32x32 with row sub-sampling): every lane owns ONE segment of the original block for the whole
  // search, so it is loaded once (per-lane data, already XORed with bias) instead of once per candidate
  const int16_t *orgLds;   // != nullptr: the (row sub-sampled, bias-XORed) original block staged in LDS, row length w (multi-wave jobs)
  bool           orgResident;
  unsigned       orgSeg[4];
  int            resOff;   // r * (refStride << ss) + x of that segment
  int            totCap;   // raster column kernel: entries of the scan's totals in LDS (the dummy slot of idle lanes sits behind them)
};

__device__ __forceinline__ int floor_log2_u( unsigned v ) { return 31 - __clz( ( int ) v ); }

// wave-uniform values: tell the compiler (SGPRs, scalar branches instead of exec-mask control flow)
__device__ __forceinline__ int uni( int v ) { return __builtin_amdgcn_readfirstlane( v ); }
__device__ __forceinline__ unsigned uni( unsigned v ) { return ( unsigned ) __builtin_amdgcn_readfirstlane( ( int ) v ); }
__device__ __forceinline__ unsigned long long uni( unsigned long long v )
{
  const unsigned lo = uni( ( unsigned ) v ), hi = uni( ( unsigned ) ( v >> 32 ) );
  return ( ( unsigned long long ) hi << 32 ) | lo;
}

__device__ __forceinline__ unsigned eg_bits( int v )
{
  // xGetExpGolombNumberOfBits (RdCost.h:301-313): its `while( t > 128 ) { len += 14; t >>= 7; }` only splits floorLog2( t ) = 7 + floorLog2( t >> 7 ),
  // so the length is 1 + 2 * floorLog2( t ) for every t >= 1 -- no loop
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}

__device__ __forceinline__ unsigned long long mv_cost( const MeJob &j, int x, int y )
{
  const unsigned bits = eg_bits( ( ( x << j.costScale ) - j.predHor ) >> j.imvShift ) + eg_bits( ( ( y << j.costScale ) - j.predVer ) >> j.imvShift );
  const double   c    = j.lambda * ( double ) bits;   // fp64 multiply, truncation (RdCost.h:314)
  return j.narrow ? ( unsigned long long ) ( unsigned ) c : ( unsigned long long ) c;   // bits <= 126: c < 2^31 when narrow
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

// 8 / 4 consecutive samples from a 2-byte aligned address (gfx950 global memory handles the misalignment in hardware)
struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8 { unsigned v[4]; };
struct __attribute__( ( packed, aligned( 2 ) ) ) Pel4 { unsigned v[2]; };

// 8 (4) reference samples at a candidate's sample offset.  A multi-dword load whose address is only 2-byte aligned runs at 0.28x the rate of a 4-byte aligned one
// on gfx950 (scripts/unaligned_load.hip: 9.2 vs 33 TB/s out of L2 / TCP), and half of all candidates sit at odd sample offsets -- so the segment is always
// fetched from the dword below (one 16- / 8-byte load + the next dword, both 4-byte aligned) and funnel-shifted by 0 or 16 bits: no branch, the loads of an
// unrolled trip still go out together.  Reads at most 2 bytes before and 4 bytes after the segment (inside the plane's margin rows / the next segment).
struct __attribute__( ( packed, aligned( 4 ) ) ) Dw4 { unsigned v[4]; };
struct __attribute__( ( packed, aligned( 4 ) ) ) Dw2 { unsigned v[2]; };
__device__ __forceinline__ Pel8 ld8( const int16_t *p )
{
  const unsigned  sh = ( ( unsigned ) reinterpret_cast<uintptr_t>( p ) & 2u ) << 3;
  const unsigned *q  = reinterpret_cast<const unsigned *>( reinterpret_cast<uintptr_t>( p ) & ~( uintptr_t ) 3 );
  const Dw4       a  = *reinterpret_cast<const Dw4 *>( q );
  const unsigned  e  = q[4];
  Pel8 r;
  r.v[0] = __builtin_amdgcn_alignbit( a.v[1], a.v[0], sh ); r.v[1] = __builtin_amdgcn_alignbit( a.v[2], a.v[1], sh );
  r.v[2] = __builtin_amdgcn_alignbit( a.v[3], a.v[2], sh ); r.v[3] = __builtin_amdgcn_alignbit( e, a.v[3], sh );
  return r;
}
__device__ __forceinline__ Pel4 ld4( const int16_t *p )
{
  const unsigned  sh = ( ( unsigned ) reinterpret_cast<uintptr_t>( p ) & 2u ) << 3;
  const unsigned *q  = reinterpret_cast<const unsigned *>( reinterpret_cast<uintptr_t>( p ) & ~( uintptr_t ) 3 );
  const Dw2       a  = *reinterpret_cast<const Dw2 *>( q );
  const unsigned  e  = q[2];
  Pel4 r;
  r.v[0] = __builtin_amdgcn_alignbit( a.v[1], a.v[0], sh ); r.v[1] = __builtin_amdgcn_alignbit( e, a.v[1], sh );
  return r;
}

// |a.lo - b.lo| + |a.hi - b.hi| + acc on unsigned 16-bit halves: one VALU instruction per two samples.
// Signed samples are made unsigned by flipping the sign bits of both operands (bias), which leaves |a - b| unchanged.
__device__ __forceinline__ unsigned sad2( unsigned a, unsigned b, unsigned acc ) { return __builtin_amdgcn_sad_u16( a, b, acc ); }

// partial SAD of candidate (cx, cy) over the items sub, sub + lpc, ... (RdCost.cpp:493-528 arithmetic)
// The sign-bias XOR (samples that may be negative: the bi-prediction target 2 org - pred) is needed on a minority of jobs; unsigned jobs -- every uni
// search -- take the instantiation without it (one vector instruction less per v_sad_u16 in the inner loops)
template<bool SGN> __device__ __forceinline__ unsigned bx( unsigned v, unsigned bias ) { return SGN ? v ^ bias : v; }

template<int WPJ, bool SGN>
__device__ __forceinline__ unsigned sad_partial_impl( const MeJob &j, int cx, int cy, int sub )
{
  const int16_t *c0 = j.ref + ( long ) cy * j.refStride + cx;
  const long     os = ( long ) j.orgStride << j.ss, cs = ( long ) j.refStride << j.ss;
  unsigned       s = 0;
  if( j.orgResident )
  {
    if( j.seg == 8 )
    {
      const Pel8 b = *reinterpret_cast<const Pel8 *>( c0 + j.resOff );
#pragma unroll
      for( int k = 0; k < 4; k++ ) s = sad2( j.orgSeg[k], bx<SGN>( b.v[k], j.bias ), s );
    }
    else
    {
      const Pel4 b = *reinterpret_cast<const Pel4 *>( c0 + j.resOff );
#pragma unroll
      for( int k = 0; k < 2; k++ ) s = sad2( j.orgSeg[k], bx<SGN>( b.v[k], j.bias ), s );
    }
    return s;
  }
  if( WPJ >= 2 && j.seg == 8 && j.orgLds )   // (single-wave kernels never stage the block: keep this path out of their register budget)
  {
    // items are 16-byte aligned in the LDS copy (w is a multiple of 8): one ds_read_b128 per item instead of a second vector-memory load
    if( j.lpc == 64 && j.sprShift >= 0 && j.sprShift <= 6 )
    {
      // a whole wave on one candidate, power-of-two row length: lane `sub` keeps its column and walks down 64 >> sprShift rows per step,
      // so both addresses advance by constants
      const int      rowsPerStep = 64 >> j.sprShift;
      const int16_t *pr = c0 + ( long ) ( sub >> j.sprShift ) * cs + ( ( sub & ( j.segsPerRow - 1 ) ) << 3 );
      const int16_t *po = j.orgLds + ( sub << 3 );
      const long     dr = cs * rowsPerStep;
      int            it = sub;
      // four steps per trip: the four reference loads of a lane are in flight together (a wave walks ONE candidate here, so without this every
      // step pays a full vector-memory latency: 100+ cycles per vector instruction on the 128x128 level)
      const bool odd = ( reinterpret_cast<uintptr_t>( pr ) & 2 ) != 0 && !( cs & 1 );   // wave-uniform: every lane's address has the candidate's parity (even strides)
      for( ; it + 192 < j.items; it += 256 )
      {
        Pel8 b0, b1, b2, b3;
        if( odd ) { b0 = ld8( pr ); b1 = ld8( pr + dr ); b2 = ld8( pr + 2 * dr ); b3 = ld8( pr + 3 * dr ); }
        else
        {
          b0 = *reinterpret_cast<const Pel8 *>( pr ); b1 = *reinterpret_cast<const Pel8 *>( pr + dr );
          b2 = *reinterpret_cast<const Pel8 *>( pr + 2 * dr ); b3 = *reinterpret_cast<const Pel8 *>( pr + 3 * dr );
        }
        const uint4 a0 = *reinterpret_cast<const uint4 *>( po ), a1 = *reinterpret_cast<const uint4 *>( po + 512 );
        const uint4 a2 = *reinterpret_cast<const uint4 *>( po + 1024 ), a3 = *reinterpret_cast<const uint4 *>( po + 1536 );
        s = sad2( a0.x, bx<SGN>( b0.v[0], j.bias ), s ); s = sad2( a0.y, bx<SGN>( b0.v[1], j.bias ), s ); s = sad2( a0.z, bx<SGN>( b0.v[2], j.bias ), s ); s = sad2( a0.w, bx<SGN>( b0.v[3], j.bias ), s );
        s = sad2( a1.x, bx<SGN>( b1.v[0], j.bias ), s ); s = sad2( a1.y, bx<SGN>( b1.v[1], j.bias ), s ); s = sad2( a1.z, bx<SGN>( b1.v[2], j.bias ), s ); s = sad2( a1.w, bx<SGN>( b1.v[3], j.bias ), s );
        s = sad2( a2.x, bx<SGN>( b2.v[0], j.bias ), s ); s = sad2( a2.y, bx<SGN>( b2.v[1], j.bias ), s ); s = sad2( a2.z, bx<SGN>( b2.v[2], j.bias ), s ); s = sad2( a2.w, bx<SGN>( b2.v[3], j.bias ), s );
        s = sad2( a3.x, bx<SGN>( b3.v[0], j.bias ), s ); s = sad2( a3.y, bx<SGN>( b3.v[1], j.bias ), s ); s = sad2( a3.z, bx<SGN>( b3.v[2], j.bias ), s ); s = sad2( a3.w, bx<SGN>( b3.v[3], j.bias ), s );
        pr += 4 * dr;
        po += 2048;
      }
      for( ; it < j.items; it += 64 )
      {
        const uint4 a = *reinterpret_cast<const uint4 *>( po );
        const Pel8  b = *reinterpret_cast<const Pel8 *>( pr );
        s = sad2( a.x, bx<SGN>( b.v[0], j.bias ), s ); s = sad2( a.y, bx<SGN>( b.v[1], j.bias ), s );
        s = sad2( a.z, bx<SGN>( b.v[2], j.bias ), s ); s = sad2( a.w, bx<SGN>( b.v[3], j.bias ), s );
        pr += dr;
        po += 512;
      }
      return s;
    }
    for( int it = sub; it < j.items; it += j.lpc )
    {
      const int   r = j.sprShift >= 0 ? it >> j.sprShift : it / j.segsPerRow, x = ( it - r * j.segsPerRow ) << 3;
      const uint4 a = *reinterpret_cast<const uint4 *>( j.orgLds + ( it << 3 ) );
      const Pel8  b = *reinterpret_cast<const Pel8 *>( c0 + r * cs + x );
      s = sad2( a.x, bx<SGN>( b.v[0], j.bias ), s ); s = sad2( a.y, bx<SGN>( b.v[1], j.bias ), s );
      s = sad2( a.z, bx<SGN>( b.v[2], j.bias ), s ); s = sad2( a.w, bx<SGN>( b.v[3], j.bias ), s );
    }
  }
  else if( j.seg == 8 )
  {
    for( int it = sub; it < j.items; it += j.lpc )
    {
      const int  r = j.sprShift >= 0 ? it >> j.sprShift : it / j.segsPerRow, x = ( it - r * j.segsPerRow ) << 3;
      const Pel8 a = *reinterpret_cast<const Pel8 *>( j.org + r * os + x );
      const Pel8 b = *reinterpret_cast<const Pel8 *>( c0 + r * cs + x );
#pragma unroll
      for( int k = 0; k < 4; k++ ) s = sad2( bx<SGN>( a.v[k], j.bias ), bx<SGN>( b.v[k], j.bias ), s );
    }
  }
  else
  {
    for( int it = sub; it < j.items; it += j.lpc )
    {
      const int  r = j.sprShift >= 0 ? it >> j.sprShift : it / j.segsPerRow, x = ( it - r * j.segsPerRow ) << 2;
      const Pel4 a = *reinterpret_cast<const Pel4 *>( j.org + r * os + x );
      const Pel4 b = *reinterpret_cast<const Pel4 *>( c0 + r * cs + x );
#pragma unroll
      for( int k = 0; k < 2; k++ ) s = sad2( bx<SGN>( a.v[k], j.bias ), bx<SGN>( b.v[k], j.bias ), s );
    }
  }
  return s;
}

template<int WPJ>
__device__ __forceinline__ unsigned sad_partial( const MeJob &j, int cx, int cy, int sub )
{
  return j.bias ? sad_partial_impl<WPJ, true>( j, cx, cy, sub ) : sad_partial_impl<WPJ, false>( j, cx, cy, sub );
}

// Multi-wave jobs (large blocks): the original block -- read again for EVERY candidate -- is staged once in LDS, item-major
// (item it = 8 samples at sLds[it * 8]), already XORed with the sign bias; candidates then fetch only the reference through the vector
// memory path and the original through the LDS port.  Blocks that do not fit (or 4-sample segments) keep the global path.
template<int WPJ, int CAP>
__device__ __forceinline__ void stage_org( MeJob &j, int16_t *sLds, int tid )
{
  j.orgLds = nullptr;
  if( WPJ < 2 || j.seg != 8 || j.items * 8 > CAP ) return;
  const long os = ( long ) j.orgStride << j.ss;
  for( int it = tid; it < j.items; it += 64 * WPJ )
  {
    const int  r = j.sprShift >= 0 ? it >> j.sprShift : it / j.segsPerRow, x = ( it - r * j.segsPerRow ) << 3;
    const Pel8 a = *reinterpret_cast<const Pel8 *>( j.org + r * os + x );
    *reinterpret_cast<uint4 *>( sLds + ( it << 3 ) ) = make_uint4( a.v[0] ^ j.bias, a.v[1] ^ j.bias, a.v[2] ^ j.bias, a.v[3] ^ j.bias );
  }
  __syncthreads();
  j.orgLds = sLds;
}

// lexicographic (cost, index) minimum over the wave
// Cross-lane steps on the DPP path of the vector ALU (one instruction each, no LDS round trip as with ds_bpermute): xor 1 / xor 2 inside a quad,
// then the mirrors inside 8 and 16 lanes -- after each step the lanes of the growing group all hold the group's result.
template<int CTRL>
__device__ __forceinline__ unsigned dpp_u32( unsigned v )
{
  return ( unsigned ) __builtin_amdgcn_update_dpp( 0, ( int ) v, CTRL, 0xf, 0xf, true );
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror

__device__ __forceinline__ unsigned group_sum( unsigned s, int lpc )   // sum over aligned groups of lpc (power of two) lanes; every lane gets it
{
  if( lpc >= 2 ) s += dpp_u32<DPP_XOR1>( s );
  if( lpc >= 4 ) s += dpp_u32<DPP_XOR2>( s );
  if( lpc >= 8 ) s += dpp_u32<DPP_HALF_MIRROR>( s );
  if( lpc >= 16 ) s += dpp_u32<DPP_MIRROR>( s );
  if( lpc >= 32 ) s += __shfl_xor( s, 16, 64 );
  if( lpc >= 64 ) s += __shfl_xor( s, 32, 64 );
  return s;
}

__device__ __forceinline__ unsigned wave_min_u32( unsigned v )   // all 64 lanes active; result wave-uniform (SGPR)
{
  v = min( v, dpp_u32<DPP_XOR1>( v ) );
  v = min( v, dpp_u32<DPP_XOR2>( v ) );
  v = min( v, dpp_u32<DPP_HALF_MIRROR>( v ) );
  v = min( v, dpp_u32<DPP_MIRROR>( v ) );   // the four rows of 16 are uniform now
  const unsigned a = ( unsigned ) __builtin_amdgcn_readlane( ( int ) v, 0 ), b = ( unsigned ) __builtin_amdgcn_readlane( ( int ) v, 16 );
  const unsigned c = ( unsigned ) __builtin_amdgcn_readlane( ( int ) v, 32 ), d = ( unsigned ) __builtin_amdgcn_readlane( ( int ) v, 48 );
  return min( min( a, b ), min( c, d ) );
}

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

constexpr int ORG_LDS_CAP  = 128 * 128;   // samples of the LDS copy of the original block (kernels with 4+ waves per job)
constexpr int ORG_LDS_CAP2 = 64 * 64;     // two waves per job (blocks up to 64x64)

struct TzState
{
  Range              sr;
  unsigned long long bestSad;
  int                bestX, bestY, pointNr;
  unsigned           bestDist, bestRound, nEval;
};

// The lanes that run one job: WPJ waves (WPJ = 1: one wave, four independent jobs per workgroup; WPJ > 1: the whole
// workgroup is one job and the waves split every candidate list).  All waves carry identical search state.
struct Coop
{
  int                 lane, wave, wpj;
  bool                leader;    // the single thread that writes the candidate list
  unsigned long long *redCost;   // [wpj] cross-wave arg-min exchange (LDS)
  unsigned           *redIdx;
};

template<int WPJ>
__device__ __forceinline__ void job_sync()
{
  if( WPJ == 1 )
  {
    // DS operations of one wave execute in order; the fence keeps the compiler from reordering across it
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
  }
  else
  {
    __syncthreads();
  }
}

constexpr int TZ_UNR = 2;   // passes whose reference loads are issued together when the block sits in registers

// Evaluates `total` candidates, 64 / lpc at a time, and returns the lexicographic (cost, index) minimum (wave-uniform).
// RASTER: candidate k = grid point (k % nx, k / nx) of the raster; otherwise pts[k].
template<bool RASTER, int WPJ>
__device__ __forceinline__ void eval_candidates( const MeJob &j, const int4 *pts, int total, int rLeft, int rTop, int nx, int stepXY, const Coop &co,
                                                 unsigned long long &bestCost, unsigned &bestIdx )
{
  const int lane = co.lane;
  const int lpc = j.lpc, cpw = 64 >> j.lpcShift;
  const int grp = lane >> j.lpcShift, sub = lane & ( lpc - 1 );
  bestCost = ~0ull;
  bestIdx  = 0xffffffffu;
  int first = co.wave * cpw;
  if( !RASTER && WPJ == 1 && j.orgResident && j.seg == 8 && lpc == 64 )   // (32x32 blocks: single-wave jobs, one candidate per pass)
  {
    // block in registers (one 16-byte segment per lane): a pass is ONE reference load per lane, so the loads of TZ_UNR passes go out together --
    // a 16-point round of a 32x32 block (one candidate per pass) otherwise waits out 16 vector-memory latencies in a row
    for( ; first < total; first += TZ_UNR * cpw * WPJ )
    {
      Pel8 b[TZ_UNR];
      int  xs[TZ_UNR], ys[TZ_UNR];
#pragma unroll
      for( int q = 0; q < TZ_UNR; q++ )
      {
        const int  k = first + q * cpw * WPJ + grp;
        const int4 p = pts[min( k, total - 1 )];
        xs[q] = p.x; ys[q] = p.y;
        b[q]  = *reinterpret_cast<const Pel8 *>( j.ref + ( long ) p.y * j.refStride + p.x + j.resOff );
      }
#pragma unroll
      for( int q = 0; q < TZ_UNR; q++ )
      {
        const int k = first + q * cpw * WPJ + grp;
        unsigned  s = 0;
        if( j.bias )
        {
#pragma unroll
          for( int i = 0; i < 4; i++ ) s = sad2( j.orgSeg[i], b[q].v[i] ^ j.bias, s );
        }
        else
        {
#pragma unroll
          for( int i = 0; i < 4; i++ ) s = sad2( j.orgSeg[i], b[q].v[i], s );
        }
        s = group_sum( s, lpc );
        if( k < total )
        {
          const unsigned long long c = ( ( unsigned long long ) s << j.ss ) + mv_cost( j, xs[q], ys[q] );
          if( c < bestCost ) { bestCost = c; bestIdx = ( unsigned ) k; }
        }
      }
    }
  }
  for( int base = first; base < total; base += cpw * WPJ )
  {
    const int k = base + grp;
    unsigned  s = 0;
    int       x = 0, y = 0;
    if( k < total )
    {
      if( RASTER )
      {
        const int ry = k / nx, rx = k - ry * nx;
        x = rLeft + rx * stepXY;
        y = rTop + ry * stepXY;
      }
      else
      {
        const int4 p = pts[k];
        x = p.x;
        y = p.y;
      }
      s = sad_partial<WPJ>( j, x, y, sub );
    }
    s = group_sum( s, lpc );   // every lane of the group ends with the total
    if( k < total )
    {
      const unsigned long long c = ( ( unsigned long long ) s << j.ss ) + mv_cost( j, x, y );
      if( c < bestCost ) { bestCost = c; bestIdx = ( unsigned ) k; }   // k increases: keeps the earliest index per lane
    }
  }
  if( j.narrow )
  {
    // cost < 2^32: lexicographic minimum of (cost, index) as two 32-bit wave minima -- the cost, then the index among the lanes that hold it;
    // cost < 2^26 and at most 64 candidates (every diamond / start round of a block up to 128x128 of 10-bit samples): one minimum over (cost << 6 | index)
    unsigned cm, km;
    if( j.tiny && total <= 64 )
    {
      const unsigned m = wave_min_u32( bestIdx == 0xffffffffu ? 0xffffffffu : ( ( unsigned ) bestCost << 6 ) | bestIdx );
      cm = m == 0xffffffffu ? 0xffffffffu : m >> 6;
      km = m == 0xffffffffu ? 0xffffffffu : m & 63u;
    }
    else
    {
      const unsigned c32 = bestIdx == 0xffffffffu ? 0xffffffffu : ( unsigned ) bestCost;
      cm = wave_min_u32( c32 );
      km = wave_min_u32( c32 == cm ? bestIdx : 0xffffffffu );
    }
    unsigned long long key = km == 0xffffffffu ? ~0ull : ( ( unsigned long long ) cm << 32 ) | km;
    if( WPJ > 1 )
    {
      if( lane == 0 ) co.redCost[co.wave] = key;
      __syncthreads();
#pragma unroll
      for( int w = 0; w < WPJ; w++ )
      {
        const unsigned long long ok = co.redCost[w];
        key = ok < key ? ok : key;
      }
      __syncthreads();
    }
    key      = uni( key );
    bestCost = key == ~0ull ? ~0ull : key >> 32;
    bestIdx  = ( unsigned ) key;
    return;
  }
  wave_argmin( bestCost, bestIdx );
  if( WPJ > 1 )
  {
    if( lane == 0 ) { co.redCost[co.wave] = bestCost; co.redIdx[co.wave] = bestIdx; }
    __syncthreads();
#pragma unroll
    for( int w = 0; w < WPJ; w++ )
    {
      const unsigned long long oc = co.redCost[w];
      const unsigned           oi = co.redIdx[w];
      if( oc < bestCost || ( oc == bestCost && oi < bestIdx ) ) { bestCost = oc; bestIdx = oi; }
    }
    __syncthreads();
  }
  bestCost = uni( bestCost );
  bestIdx  = uni( bestIdx );
}

// evaluate the n (<= 16) candidates whose (x, y, nr, dist) sit in pts[] and replay the accept rule
template<int WPJ>
__device__ __forceinline__ void tz_round( const MeJob &j, TzState &s, const int4 *pts, int n, const Coop &co, bool touchMeta )
{
  job_sync<WPJ>();   // pts[] was written by the job's leader thread
  unsigned long long cost;
  unsigned           idx;
  eval_candidates<false, WPJ>( j, pts, n, 0, 0, 1, 1, co, cost, idx );
  s.nEval += ( unsigned ) n;
  if( n > 0 && cost < s.bestSad )
  {
    const int4 p = pts[idx];
    s.bestSad    = cost;
    s.bestX      = uni( p.x );
    s.bestY      = uni( p.y );
    if( touchMeta )
    {
      s.bestDist  = ( unsigned ) uni( p.w );
      s.bestRound = 0;
      s.pointNr   = uni( p.z );
    }
  }
  job_sync<WPJ>();   // everyone has read pts[] before the next round overwrites it
}

#define PUSH( X, Y, NR, D ) do { if( co.leader ) pts[n] = make_int4( ( X ), ( Y ), ( NR ), ( D ) ); n++; } while( 0 )

// ordered candidate list of one diamond round (xTZ8PointDiamondSearch :504-705); returns the count (<= 16)
__device__ int diamond_points( const Range &sr, int sx, int sy, int d, bool corners, int4 *pts, const Coop &co )
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

template<int WPJ>
__device__ __forceinline__ void tz_diamond( const MeJob &j, TzState &s, int sx, int sy, int d, bool corners, int4 *pts, const Coop &co )
{
  const int n = diamond_points( s.sr, sx, sy, d, corners, pts, co );
  s.bestRound += 1;
  tz_round<WPJ>( j, s, pts, n, co, true );
}

template<int WPJ>
__device__ __forceinline__ void tz_two_point( const MeJob &j, TzState &s, int4 *pts, const Coop &co )
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
  tz_round<WPJ>( j, s, pts, n, co, true );
}

// Raster scan when a whole wave shares one candidate (lpc == 64, 8-sample segments, power-of-two row length): the lane's IPL
// original segments stay in registers for the whole scan (they are the same for every candidate) and only the reference
// segments are fetched per candidate, which halves the vector-memory traffic of the scan.  Candidate order and the
// (cost, index) minimum are those of eval_candidates<true, WPJ>.
template<int IPL, int WPJ>
__device__ __forceinline__ void raster_resident_org( const MeJob &j, int total, int rLeft, int rTop, int nx, int stepXY, const Coop &co,
                                                     unsigned long long &bestCost, unsigned &bestIdx )
{
  const int lane = co.lane;
  const long os = ( long ) j.orgStride << j.ss, cs = ( long ) j.refStride << j.ss;
  Pel8 o[IPL];
  int  off[IPL];
#pragma unroll
  for( int m = 0; m < IPL; m++ )
  {
    const int it = lane + 64 * m, r = it >> j.sprShift, x = ( it - ( r << j.sprShift ) ) << 3;
    o[m] = *reinterpret_cast<const Pel8 *>( j.org + r * os + x );
#pragma unroll
    for( int k = 0; k < 4; k++ ) o[m].v[k] ^= j.bias;
    off[m] = ( int ) ( r * cs + x );
  }
  bestCost = ~0ull;
  bestIdx  = 0xffffffffu;
  int ry = co.wave / nx, rx = co.wave - ry * nx;          // candidate k = wave, wave + WPJ, ...
  for( int k = co.wave; k < total; k += WPJ )
  {
    const int      x = rLeft + rx * stepXY, y = rTop + ry * stepXY;
    const int16_t *c0 = j.ref + ( long ) y * j.refStride + x;
    unsigned       s = 0;
#pragma unroll
    for( int m = 0; m < IPL; m++ )
    {
      const Pel8 b = *reinterpret_cast<const Pel8 *>( c0 + off[m] );
#pragma unroll
      for( int q = 0; q < 4; q++ ) s = sad2( o[m].v[q], b.v[q] ^ j.bias, s );
    }
#pragma unroll
    for( int sh = 32; sh > 0; sh >>= 1 ) s += __shfl_xor( s, sh, 64 );
    const unsigned long long c = ( ( unsigned long long ) uni( s ) << j.ss ) + mv_cost( j, x, y );
    if( c < bestCost ) { bestCost = c; bestIdx = ( unsigned ) k; }
    rx += WPJ;
    while( rx >= nx ) { rx -= nx; ry++; }
  }
  // wave-uniform (cost, index); merge across the job's waves
  if( WPJ > 1 )
  {
    if( lane == 0 ) { co.redCost[co.wave] = bestCost; co.redIdx[co.wave] = bestIdx; }
    __syncthreads();
#pragma unroll
    for( int w = 0; w < WPJ; w++ )
    {
      const unsigned long long oc = co.redCost[w];
      const unsigned           oi = co.redIdx[w];
      if( oc < bestCost || ( oc == bestCost && oi < bestIdx ) ) { bestCost = oc; bestIdx = oi; }
    }
    __syncthreads();
  }
  bestCost = uni( bestCost );
  bestIdx  = uni( bestIdx );
}

// raster scan (xTZSearch :3888-3899 / adaptive :3883-3903): candidate k = (row k / nx, column k % nx) in row-major order
template<int WPJ>
__device__ __forceinline__ void tz_raster( const MeJob &j, TzState &s, const Range &r, int stepXY, const Coop &co )
{
  const int nx = r.right >= r.left ? ( r.right - r.left ) / stepXY + 1 : 0;
  const int ny = r.bottom >= r.top ? ( r.bottom - r.top ) / stepXY + 1 : 0;
  const int total = nx * ny;
  unsigned long long cost;
  unsigned           idx;
  // register-resident original (raster_resident_org) for 1 and 4 segments per lane (32x32 / 64x64 with row sub-sampling); the
  // 16-segment case (128x128) costs ~110 VGPRs and measured slower than the generic path at the occupancy that leaves
  const int  ipl  = j.items >> 6;
  const bool fits = total > 0 && j.lpc == 64 && j.seg == 8 && j.sprShift >= 0 && ( j.items & 63 ) == 0;
  if( false ) {}
  else if( WPJ >= 4 && fits && ipl == 4 ) raster_resident_org<4, WPJ>( j, total, r.left, r.top, nx, stepXY, co, cost, idx );
  else if( WPJ >= 2 && fits && ipl == 1 ) raster_resident_org<1, WPJ>( j, total, r.left, r.top, nx, stepXY, co, cost, idx );
  else eval_candidates<true, WPJ>( j, nullptr, total, r.left, r.top, nx > 0 ? nx : 1, stepXY, co, cost, idx );
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

// ---- split search: the raster scan as its own kernel ---------------------------------------------------------------------------------
// A raster scan (:3888-3899) evaluates (2 SR / 5 + 1)^2 = 1521 positions of a 128x128 block: 25 M sample differences, each reference sample
// re-read by ~26 candidates.  One candidate at a time per workgroup (eval_candidates) makes that an L2-bandwidth problem (50 MB through L1 per
// job).  When the batch is launched in split mode the search kernel stops at the raster decision, the jobs that need a scan are collected in a
// list, tz_raster_cols_kernel scans them with register reuse, and the search kernel resumes (mode 2) with the scan's (cost, index) minimum --
// the same accept rule, the same result.
struct TzSaved
{
  Range              sr;
  unsigned long long bestSad;
  int                bestX, bestY, pointNr;
  unsigned           bestDist, bestRound, nEval;
  unsigned long long rasterCost;
  unsigned           rasterIdx;
  int                pad;
};
constexpr int RASTER_TOT_CAP = 40 * 40;   // candidates of a scan the column kernel takes by default (SR 96: 39 x 39); vtmhip_pic_params::maxSearchRange raises it per launch
constexpr int RASTER_CHUNK   = 13;        // block rows (slots) of one task at most

// One lane = one raster COLUMN (dx = left + 5 col) of one 8-sample column segment k.  The block rows the SAD visits (every (1 << ss)-th) and the scan's dy values
// (top + 5 m) pair up by residue: with rowStep = 5 << ss, block row r = rho + rowStep * s (rho = c << ss, c = 0..4) and candidate
// m = p + (1 << ss) * m' meet in reference row top + 5 p + rho + rowStep * (m' + s).  A task = (8-sample column segment k, class rho, parity p,
// chunk of <= 13 rows of the class): the lane walks down those reference rows; row u feeds candidate m' = u - slot for every slot at once, so
// ONE 16-byte reference load serves up to 13 candidates.  The chunk's block-row segments stay in registers for the whole task, the running
// sums rotate by one slot per step (a candidate enters slot 0 and leaves slot NS - 1 complete for this task) and are added to the candidate's
// total in LDS.  Reference traffic of a 128x128 scan: ~5 MB instead of 50 MB through L1 / L2; no per-candidate cross-lane reduction at all.
// A chunk of exactly N block rows: N steps per trip with the accumulators addressed by candidate modulo N -- at step u = u0 + d slot sl works for
// candidate u - sl, whose accumulator is A[(d - sl) mod N]: a compile-time index, so nothing rotates; the candidate entering slot 0 starts from zero,
// the one in slot N - 1 is complete for this task and goes to its total (from step N - 1 on, every step completes one).  Scalar instructions
// issue at one per 4 cycles per SIMD like the vector ones (profiles/r02_valu_issue.jsonl, s_add_u32), so the steady-state trips carry no
// per-step range checks, branches or exec-mask edits: lanes without a column add into a dummy slot, the reference rows arrive two steps ahead.
#define RASTER_STEP( D, FLUSH, GUARDED )                                                                                            \
  {                                                                                                                                 \
    const Pel8 b = q0;                                                                                                              \
    q0 = q1;                                                                                                                        \
    if( !( GUARDED ) || u0 + ( D ) + 2 <= uEnd ) q1 = *reinterpret_cast<const Pel8 *>( pr );                                        \
    pr += dr;                                                                                                                       \
    const unsigned b0 = bx<SGN>( b.v[0], j.bias ), b1 = bx<SGN>( b.v[1], j.bias ), b2 = bx<SGN>( b.v[2], j.bias ), b3 = bx<SGN>( b.v[3], j.bias );  \
    _Pragma( "unroll" ) for( int sl = 0; sl < N; sl++ )                                                                             \
    {                                                                                                                               \
      const int ai = ( ( D ) - sl + N ) % N;                                                                                        \
      unsigned  a  = sl == 0 ? 0u : A[ai];                                                                                          \
      a = sad2( O[sl][0], b0, a ); a = sad2( O[sl][1], b1, a ); a = sad2( O[sl][2], b2, a ); a = sad2( O[sl][3], b3, a );           \
      A[ai] = a;                                                                                                                    \
    }                                                                                                                               \
    if( FLUSH ) { atomicAdd( &sTot[fa], A[( ( D ) + 1 ) % N] ); fa += fstep; }                                                      \
  }

template<int N, bool SGN>
__device__ __forceinline__ void raster_run( const MeJob &j, int nx, int nyp, int p, int col, bool live, const int16_t *po, long orgStep, const int16_t *pr, long dr,
                                            unsigned *sTot )
{
  unsigned O[N][4], A[N];
#pragma unroll
  for( int sl = 0; sl < N; sl++ )
  {
    A[sl] = 0;
    const Pel8 a = *reinterpret_cast<const Pel8 *>( po + sl * orgStep );
    O[sl][0] = bx<SGN>( a.v[0], j.bias ); O[sl][1] = bx<SGN>( a.v[1], j.bias ); O[sl][2] = bx<SGN>( a.v[2], j.bias ); O[sl][3] = bx<SGN>( a.v[3], j.bias );
  }
  const int uEnd = nyp - 1 + N - 1;
  int       fa = live ? p * nx + col : j.totCap;   // total of candidate (p + (mp << ss)) of this lane's column; lanes without a column: the dummy slot
  const int fstep = live ? nx << j.ss : 0;
  Pel8 q0 = *reinterpret_cast<const Pel8 *>( pr ), q1 = q0;
  if( uEnd >= 1 ) q1 = *reinterpret_cast<const Pel8 *>( pr + dr );
  pr += 2 * dr;
  int u0 = 0;
#pragma unroll
  for( int d = 0; d < N; d++ ) RASTER_STEP( d, d == N - 1, true )      // trip 0: the first candidate completes at its last step
  for( u0 = N; u0 + N + 1 <= uEnd; u0 += N )
  {
#pragma unroll
    for( int d = 0; d < N; d++ ) RASTER_STEP( d, true, false )         // steady state
  }
  for( ; u0 <= uEnd; u0 += N )
  {
#pragma unroll
    for( int d = 0; d < N; d++ )
    {
      if( u0 + d > uEnd ) break;
      RASTER_STEP( d, true, true )
    }
  }
}
#undef RASTER_STEP

template<int NS, bool SGN>
__device__ __forceinline__ void raster_task( const MeJob &j, const Range &r, int nx, int ny, int col, bool live, int myX, int k, int rho, int p, int chunk, int rowStep,
                                             unsigned *sTot )
{
  const int par = 1 << j.ss;
  const int nyp = ( ny - p + par - 1 ) >> j.ss;                       // candidates m = p, p + par, ...
  if( nyp <= 0 ) return;
  const int rowsOfClass = ( j.h - rho + rowStep - 1 ) / rowStep;       // block rows rho, rho + rowStep, ... < h
  const int needed = min( NS, rowsOfClass - chunk * RASTER_CHUNK );
  if( needed <= 0 ) return;
  // reference row of step u: top + 5 p + rho + rowStep * (u - pad + chunk * RASTER_CHUNK)
  const int16_t *pr = j.ref + ( long ) ( r.top + 5 * p + rho + rowStep * chunk * RASTER_CHUNK ) * j.refStride + myX + ( k << 3 );
  const long     dr = ( long ) rowStep * j.refStride;
  {
    // row counts of the power-of-two block heights (128: 13 / 12 per chunk, 64 with row sub-sampling: 7 / 6, 32: 4 / 3, 16 and 8: 2 / 1) take the
    // branch-free form; any other count the rotating form below
    const int16_t *po = j.org + ( long ) ( rho + rowStep * chunk * RASTER_CHUNK ) * j.orgStride + ( k << 3 );
    const long     os = ( long ) rowStep * j.orgStride;
    switch( needed )
    {
    case 13: if( NS >= 13 ) { raster_run<13, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 12: if( NS >= 12 ) { raster_run<12, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 7: if( NS >= 7 ) { raster_run<7, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 6: if( NS >= 6 ) { raster_run<6, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 4: if( NS >= 4 ) { raster_run<4, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 3: if( NS >= 3 ) { raster_run<3, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return; } break;
    case 2: raster_run<2, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return;
    case 1: raster_run<1, SGN>( j, nx, nyp, p, col, live, po, os, pr, dr, sTot ); return;
    default: break;
    }
  }
  const int pad = NS - needed;                                         // the chunk's rows sit in slots pad .. NS - 1
  unsigned  O[NS][4], A[NS];
#pragma unroll
  for( int sl = 0; sl < NS; sl++ )
  {
    A[sl] = 0;
    const int  rr = rho + rowStep * ( chunk * RASTER_CHUNK + max( sl - pad, 0 ) );
    const Pel8 a  = *reinterpret_cast<const Pel8 *>( j.org + ( long ) rr * j.orgStride + ( k << 3 ) );
    O[sl][0] = bx<SGN>( a.v[0], j.bias ); O[sl][1] = bx<SGN>( a.v[1], j.bias ); O[sl][2] = bx<SGN>( a.v[2], j.bias ); O[sl][3] = bx<SGN>( a.v[3], j.bias );
  }
  const int uEnd = nyp - 1 + NS - 1;
  for( int u = pad; u <= uEnd; u++ )   // slots below pad stay empty, the sums rotate by one slot per step
  {
    const Pel8     b  = *reinterpret_cast<const Pel8 *>( pr );
    const unsigned b0 = bx<SGN>( b.v[0], j.bias ), b1 = bx<SGN>( b.v[1], j.bias ), b2 = bx<SGN>( b.v[2], j.bias ), b3 = bx<SGN>( b.v[3], j.bias );
    pr += dr;
#pragma unroll
    for( int sl = 0; sl < NS; sl++ )
      if( sl >= pad )
      {
        unsigned a = A[sl];
        a = sad2( O[sl][0], b0, a ); a = sad2( O[sl][1], b1, a ); a = sad2( O[sl][2], b2, a ); a = sad2( O[sl][3], b3, a );
        A[sl] = a;
      }
    const int mp = u - ( NS - 1 );   // the candidate in the last slot has met every row of the chunk
    if( mp >= 0 && mp < nyp && live ) atomicAdd( &sTot[( p + ( mp << j.ss ) ) * nx + col], A[NS - 1] );
#pragma unroll
    for( int sl = NS - 1; sl > 0; sl-- ) A[sl] = A[sl - 1];
    A[0] = 0;
  }
}

__global__ __launch_bounds__( 256 ) void tz_raster_cols_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                               const vtmhip_tz_job *__restrict__ jobs, TzSaved *__restrict__ saved, const int *__restrict__ list,
                                                               unsigned *__restrict__ gTot, int parts, int totCap )
{
  // parts > 1 (few searches in the batch: one band of a picture sharded over several GPUs, the 128x128 level): `parts` workgroups share one scan -- each takes
  // every parts-th task, adds its partial totals into the scan's global totals (gTot[e][RASTER_TOT_CAP + 1], zeroed by the host; the last entry counts the
  // workgroups that are done) and the one that arrives last picks the minimum.  Integer sums: the result does not depend on the split.
  extern __shared__ unsigned    sTot[];                     // totCap totals + the dummy slot of lanes without a column
  __shared__ unsigned long long sRedCost[4];
  __shared__ unsigned           sRedIdx[4];
  __shared__ int                sLast;
  const int lane = threadIdx.x & 63, wv = uni( ( int ) ( threadIdx.x >> 6 ) );
  const int count = list[0];
  const int part = parts > 1 ? ( int ) blockIdx.x % parts : 0, eStride = parts > 1 ? ( int ) gridDim.x / parts : ( int ) gridDim.x;
  for( int e = parts > 1 ? ( int ) blockIdx.x / parts : ( int ) blockIdx.x; e < count; e += eStride )
  {
    const int            jobIdx = list[1 + e];
    const vtmhip_tz_job *jp = jobs + jobIdx;
    TzSaved             &sv = saved[jobIdx];
    MeJob j;
    j.org = orgBase + jp->orgOff; j.ref = refBase + jp->refOff; j.orgStride = jp->orgStride; j.refStride = jp->refStride;
    j.w = jp->width; j.h = jp->height; j.ss = jp->subShift; j.imvShift = ( unsigned ) jp->imvShift;
    j.predHor = jp->predHor; j.predVer = jp->predVer; j.costScale = 2; j.lambda = jp->motionLambda;
    j.bias = jp->signedSamples ? 0x80008000u : 0u;
    j.narrow = false; j.tiny = false; j.totCap = totCap;
    const Range r = sv.sr;
    const int   nx = ( r.right - r.left ) / 5 + 1, ny = ( r.bottom - r.top ) / 5 + 1, total = nx * ny;
    for( int i = threadIdx.x; i < total; i += 256 ) sTot[i] = 0;
    __syncthreads();
    const int segs = j.w >> 3, par = 1 << j.ss, rowStep = 5 << j.ss;
    const int rowsMax = ( j.h + rowStep - 1 ) / rowStep, chunks = ( rowsMax + RASTER_CHUNK - 1 ) / RASTER_CHUNK;
    // a lane = one (8-sample segment k, raster column) pair: the pairs of a (class, parity, chunk) combination are dealt to whole waves, so a 39-column scan
    // of a 16-segment block fills 624 of 640 lanes instead of 39 of every 64
    const int pairs = segs * nx, wslots = ( pairs + 63 ) >> 6;
    const int tasks = wslots * 5 * par * chunks;
    for( int t = wv + 4 * part; t < tasks; t += 4 * parts )
    {
      int q = t;
      const int ws = q % wslots; q /= wslots;
      const int c = q % 5; q /= 5;
      const int p = q % par; q /= par;
      const int chunk = q, rho = c << j.ss;
      const int  pi = ws * 64 + lane;
      const bool live = pi < pairs;
      const int  pc = min( pi, pairs - 1 );             // lanes beyond the last pair repeat it (their sums are never stored)
      const int  k = pc / nx, col = pc - k * nx, myX = r.left + 5 * col;
      // unsigned samples (every uni search): the instantiation without the sign-bias XOR of the reference segments (4 of 60 vector instructions per step)
#define RASTER_DISPATCH( SGN )                                                                                  \
      if( rowsMax > 7 ) raster_task<13, SGN>( j, r, nx, ny, col, live, myX, k, rho, p, chunk, rowStep, sTot );         \
      else if( rowsMax > 4 ) raster_task<7, SGN>( j, r, nx, ny, col, live, myX, k, rho, p, chunk, rowStep, sTot );     \
      else if( rowsMax > 2 ) raster_task<4, SGN>( j, r, nx, ny, col, live, myX, k, rho, p, chunk, rowStep, sTot );     \
      else raster_task<2, SGN>( j, r, nx, ny, col, live, myX, k, rho, p, chunk, rowStep, sTot );
      if( j.bias ) { RASTER_DISPATCH( true ) } else { RASTER_DISPATCH( false ) }
#undef RASTER_DISPATCH
    }
    __syncthreads();
    if( parts > 1 )
    {
      unsigned *g = gTot + ( size_t ) e * ( totCap + 1 );
      for( int i = threadIdx.x; i < total; i += 256 )
        if( sTot[i] ) atomicAdd( &g[i], sTot[i] );
      __threadfence();
      __syncthreads();
      if( threadIdx.x == 0 ) sLast = atomicAdd( &g[totCap], 1u ) == ( unsigned ) ( parts - 1 );
      __syncthreads();
      if( !sLast ) continue;   // block-uniform
      __threadfence();
      for( int i = threadIdx.x; i < total; i += 256 ) sTot[i] = __hip_atomic_load( &g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      __syncthreads();
    }
    // first strict minimum of SAD + MV rate in raster order = lexicographic (cost, index) minimum
    unsigned long long bc = ~0ull;
    unsigned           bi = 0xffffffffu;
    for( int i = threadIdx.x; i < total; i += 256 )
    {
      const int ry = i / nx, rx = i - ry * nx;
      const unsigned long long c = ( ( unsigned long long ) sTot[i] << j.ss ) + mv_cost( j, r.left + 5 * rx, r.top + 5 * ry );
      if( c < bc ) { bc = c; bi = ( unsigned ) i; }
    }
    wave_argmin( bc, bi );
    if( lane == 0 ) { sRedCost[wv] = bc; sRedIdx[wv] = bi; }
    __syncthreads();
    if( threadIdx.x == 0 )
    {
      for( int w = 1; w < 4; w++ )
        if( sRedCost[w] < bc || ( sRedCost[w] == bc && sRedIdx[w] < bi ) ) { bc = sRedCost[w]; bi = sRedIdx[w]; }
      sv.rasterCost = bc; sv.rasterIdx = bi;
    }
    __syncthreads();
  }
}

// WPJ = 1: 256 threads = 4 independent jobs.  WPJ > 1: 64 * WPJ threads = 1 job.
template<int WPJ>
__device__ __forceinline__ void tz_search_one( const vtmhip_pic_params &pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                               const vtmhip_tz_job *__restrict__ jobs, int numJobs, vtmhip_me_result *__restrict__ results, int mode, TzSaved *__restrict__ saved,
                                               int *__restrict__ list, int totCap, const MeFuse &fu, int blockIdxX, int gridDimX )
{
  // mode 0: the whole search.  Split launches: mode 1 stops at the raster decision of jobs tz_raster_cols_kernel can take (state -> saved[], job
  // index -> list[]; every other job runs to the end here); mode 2 resumes the listed jobs after the scan.
  constexpr int JOBS_PER_BLOCK = WPJ == 1 ? 4 : 1;
  __shared__ int4               sPts[JOBS_PER_BLOCK][16];   // 15 m_uniMvList candidates / 16 diamond points at most
  __shared__ unsigned long long sRedCost[WPJ];
  __shared__ unsigned           sRedIdx[WPJ];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sOrgLds[WPJ >= 4 ? ORG_LDS_CAP : WPJ == 2 ? ORG_LDS_CAP2 : 8];
  const int lane = threadIdx.x & 63, wv = uni( ( int ) ( threadIdx.x >> 6 ) );
  const int blk    = mode == 2 ? blockIdxX : xcd_order( blockIdxX, gridDimX );
  int       jobIdx = WPJ == 1 ? blk * 4 + wv : blk;
  if( mode == 2 )
  {
    if( jobIdx >= list[0] ) return;   // only the listed jobs resume
    jobIdx = uni( list[1 + jobIdx] );
  }
  if( jobIdx >= numJobs ) return;   // WPJ == 1: whole waves leave; WPJ > 1: never true (grid = numJobs)
  // The job's scalar fields live in registers (wave-uniform: the scalar unit keeps them), its start-candidate list behind a pointer.
  // FUSED (fu.me != nullptr, mest_glue.hpp): both are derived from the xMotionEstimation row itself -- xEstimateMvPredAMVP's selection when the template SADs are given, then
  // what mest_prepare_kernel would have written to a job table; the distinct m_uniMvList entries go to LDS (no job table, no launch in between)
  __shared__ int sExtra[JOBS_PER_BLOCK][15][2];
  vtmhip_tz_job  tj;      // (extraStart / numExtraStart of this copy are never touched: `extra` / `numExtra`)
  const int( *extra )[2];
  int       numExtra;
  if( fu.me )
  {
    vtmhip_me_job &mj = fu.me[jobIdx];
    int mvpIdx = mj.mvpIdx, predH = mj.mvPredHor, predV = mj.mvPredVer;
    unsigned bits = mj.bits;
    if( fu.amvpDout )      // xEstimateMvPredAMVP (:3088-3128): the first candidate with the smallest template cost; every lane derives it, the job's first lane stores it
    {
      const unsigned long long c0 = fu.amvpDout[2 * ( long ) jobIdx] + mg::rate( mj.motionLambda, mj.mvpIdxBits[0] );
      const unsigned long long c1 = mj.numAmvpCand > 1 ? fu.amvpDout[2 * ( long ) jobIdx + 1] + mg::rate( mj.motionLambda, mj.mvpIdxBits[1] ) : ~0ull;
      mvpIdx = c0 > c1 ? 1 : 0;
      predH = mvpIdx ? mj.amvpCand[1][0] : mj.amvpCand[0][0]; predV = mvpIdx ? mj.amvpCand[1][1] : mj.amvpCand[0][1];
      if( fu.addIdxBits ) bits += mvpIdx ? mj.mvpIdxBits[1] : mj.mvpIdxBits[0];
      if( lane == 0 && ( WPJ == 1 || wv == 0 ) )
      {
        mj.mvPredHor = predH; mj.mvPredVer = predV; mj.mvpIdx = ( uint8_t ) mvpIdx; mj.bits = bits;
        if( fu.distBiP ) fu.distBiP[jobIdx] = c0 > c1 ? c1 : c0;
      }
    }
    mg::make_tz_job_scalars( fu.cfg, mj, fuse_pat_off( fu, mj ), fuse_pat_stride( fu, mj ), predH, predV, tj );
    // the distinct m_uniMvList entries, newest first: lane i owns entry i, compares it with every earlier one and the survivors close up
    const int m = mg::num_extra( mj );
    int( *se )[2] = sExtra[WPJ == 1 ? wv : 0];
    unsigned long long keep = 0;
    if( m > 0 )      // (uniform)
    {
      int eh = 0, ev = 0;
      if( lane < m ) { eh = mj.extraStart[lane][0]; ev = mj.extraStart[lane][1]; }
      bool first = lane < m;
#pragma unroll
      for( int k = 0; k < 14; k++ )
      {
        const int oh = __shfl( eh, k, 64 ), ov = __shfl( ev, k, 64 );
        if( k < lane && oh == eh && ov == ev ) first = false;
      }
      keep = __ballot( first );
      if( first && ( WPJ == 1 || wv == 0 ) ) { const int pos = __popcll( keep & ( ( 1ull << lane ) - 1ull ) ); se[pos][0] = eh; se[pos][1] = ev; }
      job_sync<WPJ>();
    }
    numExtra = __popcll( keep );
    extra = se;
  }
  else
  {
    const vtmhip_tz_job *jp = jobs + jobIdx;
    if( uni( ( int ) jp->width ) == 0 ) return;   // empty slot of a multi-stage call (job handled by another stage); uniform per wave / per block
    tj.orgOff = jp->orgOff; tj.refOff = jp->refOff; tj.orgStride = jp->orgStride; tj.refStride = jp->refStride; tj.puX = jp->puX; tj.puY = jp->puY;
    tj.width = jp->width; tj.height = jp->height; tj.subShift = jp->subShift; tj.imvShift = jp->imvShift; tj.signedSamples = jp->signedSamples;
    tj.predHor = jp->predHor; tj.predVer = jp->predVer; tj.motionLambda = jp->motionLambda; tj.mvHor = jp->mvHor; tj.mvVer = jp->mvVer; tj.searchRange = jp->searchRange;
    tj.extendedSettings = jp->extendedSettings; tj.fastSettings = jp->fastSettings; tj.firstSearchStop = jp->firstSearchStop; tj.hasIntMv2Nx2NPred = jp->hasIntMv2Nx2NPred;
    tj.intMv2Nx2NPredHor = jp->intMv2Nx2NPredHor; tj.intMv2Nx2NPredVer = jp->intMv2Nx2NPredVer;
    numExtra = jp->numExtraStart;
    extra = jp->extraStart;
  }
  const vtmhip_tz_job *jp = &tj;      // (scalar fields only below)
  int4                *pts = sPts[WPJ == 1 ? wv : 0];
  Coop                 co;
  co.lane = lane; co.wave = WPJ == 1 ? 0 : wv; co.wpj = WPJ; co.leader = ( lane == 0 && co.wave == 0 );
  co.redCost = sRedCost; co.redIdx = sRedIdx;

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
  j.seg        = ( j.w & 7 ) == 0 ? 8 : 4;
  j.segsPerRow = j.seg == 8 ? j.w >> 3 : j.w >> 2;
  j.items      = j.segsPerRow * ( ( j.h + ( 1 << j.ss ) - 1 ) >> j.ss );
  j.lpc        = 1;
  while( j.lpc < 64 && ( j.lpc << 1 ) <= j.items ) j.lpc <<= 1;
  j.bias       = jp->signedSamples ? 0x80008000u : 0u;
  j.lpcShift   = floor_log2_u( ( unsigned ) j.lpc );
  j.sprShift   = ( j.segsPerRow & ( j.segsPerRow - 1 ) ) == 0 ? floor_log2_u( ( unsigned ) j.segsPerRow ) : -1;
  j.narrow     = j.lambda >= 0.0 && j.lambda * 126.0 < 2147483648.0;
  j.tiny       = j.narrow && ( double ) ( j.w * j.h ) * ( j.bias ? 65535.0 : ( double ) ( ( 1 << pic.bitDepth ) - 1 ) ) + j.lambda * 126.0 < 67108864.0;

  j.orgResident = j.items == j.lpc;
  j.resOff      = 0;
  j.orgSeg[0] = j.orgSeg[1] = j.orgSeg[2] = j.orgSeg[3] = 0;
  if( j.orgResident )
  {
    const int  sub = lane & ( j.lpc - 1 );
    const int  r = j.sprShift >= 0 ? sub >> j.sprShift : sub / j.segsPerRow, x = ( sub - r * j.segsPerRow ) * j.seg;
    const long os = ( long ) j.orgStride << j.ss;
    if( j.seg == 8 )
    {
      const Pel8 a = *reinterpret_cast<const Pel8 *>( j.org + r * os + x );
#pragma unroll
      for( int k = 0; k < 4; k++ ) j.orgSeg[k] = a.v[k] ^ j.bias;
    }
    else
    {
      const Pel4 a = *reinterpret_cast<const Pel4 *>( j.org + r * os + x );
      j.orgSeg[0] = a.v[0] ^ j.bias; j.orgSeg[1] = a.v[1] ^ j.bias;
    }
    j.resOff = ( int ) ( r * ( ( long ) j.refStride << j.ss ) + x );
  }
  stage_org<WPJ, ( WPJ >= 4 ? ORG_LDS_CAP : ORG_LDS_CAP2 )>( j, sOrgLds, ( int ) threadIdx.x );

  const bool ext = jp->extendedSettings != 0, fast = jp->fastSettings != 0, firstStop = jp->firstSearchStop != 0;
  const int  iRaster = fast ? 8 : 5, searchRange = jp->searchRange;

  TzState s;
  s.bestSad = ~0ull; s.bestX = 0; s.bestY = 0; s.pointNr = 0; s.bestDist = 0; s.bestRound = 0; s.nEval = 0;
  s.sr.left = s.sr.right = s.sr.top = s.sr.bottom = 0;

  int startX = 0, startY = 0;
  if( mode != 2 )
  {
  // start vector (:3675-3687)
  int mx = jp->mvHor, my = jp->mvVer;
  clip_mv( j, mx, my );
  mx = div_pow2( prec_down( mx, 2 ), 2 );
  my = div_pow2( prec_down( my, 2 ), 2 );
  if( !jp->hasIntMv2Nx2NPred )
  {
    // Start candidates in ONE round: rcMv is always accepted first (best = max), so the zero-vector test condition
    // (:3696-3704) is known up front, and the m_uniMvList candidates (:3725-3762) use the same strict-minimum rule
    // with dist = pointNr = 0 like the start vector.
    int n = 0;
    PUSH( mx, my, 0, 0 );
    if( !fast && ( mx != 0 || my != 0 ) ) PUSH( 0, 0, 0, 0 );
    const int ne = min( numExtra, 14 );
    for( int i = 0; i < ne; i++ )
    {
      int ex = extra[i][0], ey = extra[i][1];
      clip_mv( j, ex, ey );
      PUSH( prec_down( ex, 4 ), prec_down( ey, 4 ), 0, 0 );
    }
    tz_round<WPJ>( j, s, pts, n, co, true );
    if( numExtra > 14 )   // 15th candidate: the list holds 16 points
    {
      int m = 0;
      {
        int n = 0;
        int ex = extra[14][0], ey = extra[14][1];
        clip_mv( j, ex, ey );
        PUSH( prec_down( ex, 4 ), prec_down( ey, 4 ), 0, 0 );
        m = n;
      }
      tz_round<WPJ>( j, s, pts, m, co, false );
    }
  }
  else
  {
    {
      int n = 0;
      PUSH( mx, my, 0, 0 );
      tz_round<WPJ>( j, s, pts, n, co, true );
    }
    if( !fast && ( mx != 0 || my != 0 ) && ( s.bestX != 0 || s.bestY != 0 ) )
    {
      int n = 0;
      PUSH( 0, 0, 0, 0 );
      tz_round<WPJ>( j, s, pts, n, co, true );
    }
    int ix = jp->intMv2Nx2NPredHor << 4, iy = jp->intMv2Nx2NPredVer << 4;
    clip_mv( j, ix, iy );
    ix = div_pow2( prec_down( ix, 2 ), 2 );
    iy = div_pow2( prec_down( iy, 2 ), 2 );
    if( ( mx != ix || my != iy ) && ( ix != s.bestX || iy != s.bestY ) )
    {
      int n = 0;
      PUSH( ix, iy, 0, 0 );
      tz_round<WPJ>( j, s, pts, n, co, true );
    }
    // m_uniMvList start candidates (:3725-3762): one parallel round, same first-strict-minimum semantics
    int       n  = 0;
    const int ne = min( numExtra, 15 );
    for( int i = 0; i < ne; i++ )
    {
      int ex = extra[i][0], ey = extra[i][1];
      clip_mv( j, ex, ey );
      PUSH( prec_down( ex, 4 ), prec_down( ey, 4 ), 0, 0 );
    }
    tz_round<WPJ>( j, s, pts, n, co, false );
  }

  s.sr = search_range( j, s.bestX << 4, s.bestY << 4, searchRange >> ( fast ? 1 : 0 ) );

  startX = s.bestX; startY = s.bestY;
  const bool bestCandidateZero = ( s.bestX == 0 && s.bestY == 0 );

  for( int d = 1; d <= searchRange; d *= 2 )
  {
    tz_diamond<WPJ>( j, s, startX, startY, d, ext, pts, co );
    if( firstStop && s.bestRound >= 3 ) break;
  }
  if( ext && !bestCandidateZero )
  {
    for( int d = 1; d <= ( searchRange >> 1 ); d *= 2 ) tz_diamond<WPJ>( j, s, 0, 0, d, false, pts, co );
  }
  if( s.bestDist == 1 )
  {
    s.bestDist = 0;
    tz_two_point<WPJ>( j, s, pts, co );
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
    tz_raster<WPJ>( j, s, lsr, win, co );
  }
  else if( ( int ) s.bestDist >= iRaster )
  {
    s.bestDist = ( unsigned ) iRaster;
    const int nx = s.sr.right >= s.sr.left ? ( s.sr.right - s.sr.left ) / iRaster + 1 : 0, ny = s.sr.bottom >= s.sr.top ? ( s.sr.bottom - s.sr.top ) / iRaster + 1 : 0;
    if( mode == 1 && iRaster == 5 && j.seg == 8 && j.h <= 128 && nx >= 1 && ny >= 1 && nx * ny <= totCap )
    {
      if( co.leader )
      {
        TzSaved sv;
        sv.sr = s.sr; sv.bestSad = s.bestSad; sv.bestX = s.bestX; sv.bestY = s.bestY; sv.pointNr = s.pointNr; sv.bestDist = s.bestDist;
        sv.bestRound = s.bestRound; sv.nEval = s.nEval; sv.rasterCost = ~0ull; sv.rasterIdx = 0; sv.pad = 0;
        saved[jobIdx] = sv;
        list[1 + atomicAdd( &list[0], 1 )] = jobIdx;
      }
      if( fu.me && co.leader )      // the raster kernel and the resume launch read the record from the job table: store the scalar fields (neither reads the start candidates)
      {
        vtmhip_tz_job *g = fu.tzSpill + jobIdx;
        g->orgOff = tj.orgOff; g->refOff = tj.refOff; g->orgStride = tj.orgStride; g->refStride = tj.refStride; g->puX = tj.puX; g->puY = tj.puY;
        g->width = tj.width; g->height = tj.height; g->subShift = tj.subShift; g->imvShift = tj.imvShift; g->signedSamples = tj.signedSamples;
        g->predHor = tj.predHor; g->predVer = tj.predVer; g->motionLambda = tj.motionLambda; g->mvHor = tj.mvHor; g->mvVer = tj.mvVer; g->searchRange = tj.searchRange;
        g->extendedSettings = tj.extendedSettings; g->fastSettings = tj.fastSettings; g->firstSearchStop = tj.firstSearchStop; g->hasIntMv2Nx2NPred = tj.hasIntMv2Nx2NPred;
        g->intMv2Nx2NPredHor = tj.intMv2Nx2NPredHor; g->intMv2Nx2NPredVer = tj.intMv2Nx2NPredVer; g->numExtraStart = 0;
      }
      return;   // wave-uniform (WPJ == 1) / block-uniform: the search continues in the mode-2 launch
    }
    tz_raster<WPJ>( j, s, s.sr, iRaster, co );
  }
  }   // mode != 2
  else
  {
    // resume after tz_raster_cols_kernel: the scan's first strict minimum against the best point so far (xTZSearchHelp's rule)
    const TzSaved sv = saved[jobIdx];
    s.sr = sv.sr; s.bestSad = uni( sv.bestSad ); s.bestX = uni( sv.bestX ); s.bestY = uni( sv.bestY ); s.pointNr = uni( sv.pointNr );
    s.bestDist = uni( sv.bestDist ); s.bestRound = uni( sv.bestRound ); s.nEval = uni( sv.nEval );
    s.sr.left = uni( s.sr.left ); s.sr.right = uni( s.sr.right ); s.sr.top = uni( s.sr.top ); s.sr.bottom = uni( s.sr.bottom );
    const int nx = ( s.sr.right - s.sr.left ) / 5 + 1, ny = ( s.sr.bottom - s.sr.top ) / 5 + 1;
    s.nEval += ( unsigned ) ( nx * ny );
    const unsigned long long rc = uni( sv.rasterCost );
    if( rc < s.bestSad )
    {
      const int idx = ( int ) uni( sv.rasterIdx ), ry = idx / nx, rx = idx - ry * nx;
      s.bestSad = rc; s.bestX = s.sr.left + rx * 5; s.bestY = s.sr.top + ry * 5; s.bestDist = 5; s.bestRound = 0; s.pointNr = 0;
    }
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
      tz_diamond<WPJ>( j, s, startX, startY, d, ext, pts, co );
      if( fast && s.bestRound >= 2 ) break;
    }
    if( s.bestDist == 1 )
    {
      s.bestDist = 0;
      if( s.pointNr != 0 ) tz_two_point<WPJ>( j, s, pts, co );
    }
  }

  if( co.leader )
  {
    vtmhip_me_result r;
    r.mvX = s.bestX; r.mvY = s.bestY; r.nEval = s.nEval; r.reserved = 0;
    r.cost = s.bestSad;
    r.dist = s.bestSad - mv_cost( j, s.bestX, s.bestY );
    results[jobIdx] = r;
  }
}

template<int WPJ>
__global__ __launch_bounds__( WPJ == 1 ? 256 : 64 * WPJ ) __attribute__( ( amdgpu_waves_per_eu( WPJ == 2 || WPJ == 4 || WPJ == 8 ? 4 : 1 ) ) )      // two / four / eight waves per search sit at 129 .. 131 VGPRs: a register or two over the four-waves-per-SIMD budget
void tz_search_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase, const vtmhip_tz_job *__restrict__ jobs, int numJobs,
                       vtmhip_me_result *__restrict__ results, int mode, TzSaved *__restrict__ saved, int *__restrict__ list, int totCap, MeFuse fu )
{
  if( WPJ == 1 && mode == 2 )
  {
    // the resume launch of a level of small blocks: only the listed searches run (a few per cent of the level), and a grid of one workgroup per four searches of the LEVEL would be
    // 130 000 workgroups that leave at once -- 50 us of a launch for nothing.  A bounded grid walks the list instead (the waves of a workgroup are independent searches here)
    const int listed = uni( list[0] );
    for( int b = ( int ) blockIdx.x; b * 4 < listed; b += ( int ) gridDim.x )
      tz_search_one<WPJ>( pic, orgBase, refBase, jobs, numJobs, results, mode, saved, list, totCap, fu, b, ( int ) gridDim.x );
    return;
  }
  tz_search_one<WPJ>( pic, orgBase, refBase, jobs, numJobs, results, mode, saved, list, totCap, fu, ( int ) blockIdx.x, ( int ) gridDim.x );
}


// ---- small blocks: FOUR searches per wave ---------------------------------------------------------------------------------------------------------
// tz_group_kernel: a search owns one DPP row (16 lanes) and a lane is one CANDIDATE of the round -- a round of xTZ8PointDiamondSearch has at most 16 points, and lane `slot` takes the
// slot-th point of the round's list in the reference's evaluation order (:504-705; points outside the search range are skipped there and idle here, which keeps the order).  The lane
// walks its candidate's whole block (8 .. 64 segments of 8 samples: uniform batches of 8x8 .. 32x32 PUs; the original block sits in LDS, item-major) and the round's first strict
// minimum is ONE row-wide DPP minimum over (cost << 4 | slot).  No per-candidate cross-lane sum, no candidate list in LDS, no scalar control: the search state is per-lane (equal in
// the 16 lanes of a row) and the four searches of a wave step through ONE loop -- a state machine (start round, first diamond loop, two-point step, raster decision, star refinement)
// whose single evaluation site serves whatever phase each row is in, so a wave runs max-over-its-rows rounds.  Against one wave per search (tz_search_kernel<1>: 700 vector + 600 scalar
// instructions per 8x8 search, vector port 60-78 % busy) a search costs about a third of the instructions.
// Takes the FUSED uni rows only (job record from the xMotionEstimation row, mest_glue.hpp), modes 0 / 1; the few searches that go to the raster kernel resume in tz_search_kernel<1>.
#ifndef VTMHIP_TZG_SPEC
#define VTMHIP_TZG_SPEC 1
#endif
#ifndef VTMHIP_TZG_WAVES
#define VTMHIP_TZG_WAVES 4      // (128 VGPRs with 7 spilled in cold paths: measured faster than 3 waves without spills, 0.68 against 0.73 ms)
#endif
enum { GP_START = 0, GP_START15, GP_DIA1, GP_TWO1, GP_DIA2, GP_TWO2 };

template<int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64( unsigned long long v )
{
  return ( ( unsigned long long ) dpp_u32<CTRL>( ( unsigned ) ( v >> 32 ) ) << 32 ) | dpp_u32<CTRL>( ( unsigned ) v );
}
__device__ __forceinline__ unsigned row_min_u32( unsigned v )      // minimum over the 16 lanes of a DPP row; every lane of the row gets it
{
  v = min( v, dpp_u32<DPP_XOR1>( v ) );
  v = min( v, dpp_u32<DPP_XOR2>( v ) );
  v = min( v, dpp_u32<DPP_HALF_MIRROR>( v ) );
  v = min( v, dpp_u32<DPP_MIRROR>( v ) );
  return v;
}
__device__ __forceinline__ unsigned long long row_min_u64( unsigned long long v )
{
  unsigned long long o;
  o = dpp_u64<DPP_XOR1>( v ); v = o < v ? o : v;
  o = dpp_u64<DPP_XOR2>( v ); v = o < v ? o : v;
  o = dpp_u64<DPP_HALF_MIRROR>( v ); v = o < v ? o : v;
  o = dpp_u64<DPP_MIRROR>( v ); v = o < v ? o : v;
  return v;
}
__device__ __forceinline__ int row_read( int v, int lane, int slot )      // v of lane `slot` of this lane's row
{
  return __builtin_amdgcn_ds_bpermute( ( ( lane & 48 ) + slot ) << 2, v );
}

// SAD of the whole block at candidate (x, y): SPR segments per row, 8 segments (= 8 / SPR rows) per trip with their loads in flight together
template<int SPR, int SS>
__device__ __forceinline__ unsigned grp_sad( const MeJob &j, const uint4 *so, int x, int y )
{
  const int16_t *p  = j.ref + ( long ) y * j.refStride + x;
  const long     cs = ( long ) j.refStride << SS;
  unsigned       s  = 0;
  for( int it0 = 0; it0 < j.items; it0 += 8 )
  {
    Pel8 b[8];
#pragma unroll
    for( int q = 0; q < 8; q++ ) b[q] = *reinterpret_cast<const Pel8 *>( p + ( q / SPR ) * cs + ( q % SPR ) * 8 );
#pragma unroll
    for( int q = 0; q < 8; q++ )
    {
      const uint4 a = so[it0 + q];
      s = sad2( a.x, b[q].v[0], s ); s = sad2( a.y, b[q].v[1], s ); s = sad2( a.z, b[q].v[2], s ); s = sad2( a.w, b[q].v[3], s );
    }
    p += ( 8 / SPR ) * cs;
  }
  return s;
}

// first strict minimum of the row's (cost, slot) pairs: cost (~0: no candidate) and the winner's slot
// (SMALLKEY: keys below 16 -- the slots of a round; the branch must be the same for the 16 lanes of a row, the DPP steps read their neighbours)
template<bool SMALLKEY>
__device__ __forceinline__ void row_argmin( const MeJob &j, unsigned long long c, bool valid, unsigned key, unsigned long long &minCost, unsigned &minKey )
{
  if( SMALLKEY && j.tiny )      // cost < 2^26 (see MeJob::tiny): one 32-bit key
  {
    const unsigned m = row_min_u32( valid ? ( ( unsigned ) c << 4 ) | key : 0xffffffffu );
    minCost = m == 0xffffffffu ? ~0ull : ( unsigned long long ) ( m >> 4 );
    minKey  = m & 15u;
  }
  else
  {
    const unsigned long long cc = valid ? c : ~0ull;
    minCost = row_min_u64( cc );
    minKey  = row_min_u32( valid && cc == minCost ? key : 0xffffffffu );
  }
}

// the idx-th point of a diamond round at distance 2 .. 8 (:537-608), offsets in units of d / 2: (sx, top, 2) (left2, top2, 1) (right2, top2, 3) (left, sy, 4) (right, sy, 5)
// (left2, bot2, 6) (right2, bot2, 8) (sx, bot, 7); half: the point reports distance d / 2
__device__ __forceinline__ void tzg_mid_point( int idx, int &ux, int &uy, int &nr, bool &half )
{
  half = idx == 1 || idx == 2 || idx == 5 || idx == 6;
  ux   = idx == 0 || idx == 7 ? 0 : ( half ? ( idx & 1 ? -1 : 1 ) : ( idx == 3 ? -2 : 2 ) );
  uy   = idx == 3 || idx == 4 ? 0 : ( half ? ( idx < 3 ? -1 : 1 ) : ( idx == 0 ? -2 : 2 ) );
  nr   = idx == 0 ? 2 : idx == 1 ? 1 : idx == 2 ? 3 : idx == 3 ? 4 : idx == 4 ? 5 : idx == 5 ? 6 : idx == 6 ? 8 : 7;
}

template<int SPR, int SS>
__global__ __launch_bounds__( 256 ) __attribute__( ( amdgpu_waves_per_eu( VTMHIP_TZG_WAVES ) ) ) void tz_group_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase, int numJobs,
                                                         vtmhip_me_result *__restrict__ results, int mode, TzSaved *__restrict__ saved, int *__restrict__ list, int totCap, int itemsMax,
                                                         MeFuse fu )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char sDyn[];      // [16 searches][itemsMax] segments of the original blocks
  const int lane = threadIdx.x & 63, wv = ( int ) ( threadIdx.x >> 6 ), row = lane >> 4, slot = lane & 15;
  const int blk = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x );
  const int jobIdx = blk * 16 + wv * 4 + row;
  if( jobIdx >= numJobs ) return;      // whole rows leave
  uint4 *so = reinterpret_cast<uint4 *>( sDyn ) + ( size_t ) ( wv * 4 + row ) * itemsMax;

  // the job from the xMotionEstimation row (every lane of the row derives the same record)
  vtmhip_me_job &mj = fu.me[jobIdx];
  int mvpIdx = mj.mvpIdx, predH = mj.mvPredHor, predV = mj.mvPredVer;
  if( fu.amvpDout )      // xEstimateMvPredAMVP (:3088-3128): the first candidate with the smallest template cost
  {
    unsigned bits = mj.bits;
    const unsigned long long c0 = fu.amvpDout[2 * ( long ) jobIdx] + mg::rate( mj.motionLambda, mj.mvpIdxBits[0] );
    const unsigned long long c1 = mj.numAmvpCand > 1 ? fu.amvpDout[2 * ( long ) jobIdx + 1] + mg::rate( mj.motionLambda, mj.mvpIdxBits[1] ) : ~0ull;
    mvpIdx = c0 > c1 ? 1 : 0;
    predH = mvpIdx ? mj.amvpCand[1][0] : mj.amvpCand[0][0]; predV = mvpIdx ? mj.amvpCand[1][1] : mj.amvpCand[0][1];
    if( fu.addIdxBits ) bits += mvpIdx ? mj.mvpIdxBits[1] : mj.mvpIdxBits[0];
    // (every lane has read the row's fields above before lane 0 of the row rewrites them: the loads' results are consumed -- c0 / c1 / predH -- before the stores issue in program order)
    if( slot == 0 )
    {
      mj.mvPredHor = predH; mj.mvPredVer = predV; mj.mvpIdx = ( uint8_t ) mvpIdx; mj.bits = bits;
      if( fu.distBiP ) fu.distBiP[jobIdx] = c0 > c1 ? c1 : c0;
    }
  }
  vtmhip_tz_job tj;
  mg::make_tz_job_scalars( fu.cfg, mj, fuse_pat_off( fu, mj ), fuse_pat_stride( fu, mj ), predH, predV, tj );

  MeJob j;
  j.org = orgBase + tj.orgOff; j.ref = refBase + tj.refOff; j.orgStride = tj.orgStride; j.refStride = tj.refStride;
  j.w = tj.width; j.h = tj.height; j.ss = tj.subShift; j.imvShift = ( unsigned ) tj.imvShift; j.predHor = tj.predHor; j.predVer = tj.predVer; j.costScale = 2; j.lambda = tj.motionLambda;
  j.horMax = ( pic.picW + 8 - tj.puX - 1 ) << 4; j.horMin = ( -pic.ctuSize - 8 - tj.puX + 1 ) << 4;
  j.verMax = ( pic.picH + 8 - tj.puY - 1 ) << 4; j.verMin = ( -pic.ctuSize - 8 - tj.puY + 1 ) << 4;
  j.seg = 8; j.segsPerRow = SPR; j.items = SPR * ( ( j.h + ( 1 << SS ) - 1 ) >> SS ); j.bias = 0;
  j.narrow = j.lambda >= 0.0 && j.lambda * 126.0 < 2147483648.0;
  j.tiny   = j.narrow && ( double ) ( j.w * j.h ) * ( double ) ( ( 1 << pic.bitDepth ) - 1 ) + j.lambda * 126.0 < 67108864.0;
  if( j.w != SPR * 8 || j.ss != SS || j.items > itemsMax || ( j.items & 7 ) ) return;      // (the caller promised a uniform batch of this shape: never true)

  // original block -> LDS, item-major
  for( int it = slot; it < j.items; it += 16 )
  {
    const Pel8 a = *reinterpret_cast<const Pel8 *>( j.org + ( long ) ( ( it / SPR ) << SS ) * j.orgStride + ( it % SPR ) * 8 );
    so[it] = make_uint4( a.v[0], a.v[1], a.v[2], a.v[3] );
  }

  const bool fast = tj.fastSettings != 0, firstStop = tj.firstSearchStop != 0;
  const int  iRaster = fast ? 8 : 5, searchRange = tj.searchRange;

  // start candidates (:3675-3762) in the reference's order: slot 0 rcMv, slot 1 the zero vector (when tested), slots 2 .. 15 the m_uniMvList entries 0 .. 13 (an entry equal to an earlier
  // one is skipped); entry 14 follows in a round of its own (it is the last of the list either way, and a round's minimum is the first STRICT one, so the split changes nothing)
  int mx = tj.mvHor, my = tj.mvVer;
  clip_mv( j, mx, my );
  mx = div_pow2( prec_down( mx, 2 ), 2 );
  my = div_pow2( prec_down( my, 2 ), 2 );
  const int m = mg::num_extra( mj );
  int  cx = slot == 0 ? mx : 0, cy = slot == 0 ? my : 0, cnr = 0, cdist = 0;
  bool cv = slot == 0 || ( slot == 1 && !fast && ( mx != 0 || my != 0 ) );
  bool e15 = false;
  if( m > 0 )      // (row-uniform)
  {
    int  eh = 0, ev = 0;
    bool efirst = slot < m;
    if( slot < m ) { eh = mj.extraStart[slot][0]; ev = mj.extraStart[slot][1]; }
    for( int k = 0; k + 1 < m; k++ )      // (row-uniform trip count)
    {
      const int oh = row_read( eh, lane, k ), ov = row_read( ev, lane, k );
      if( k < slot && oh == eh && ov == ev ) efirst = false;
    }
    clip_mv( j, eh, ev );
    eh = prec_down( eh, 4 ); ev = prec_down( ev, 4 );
    e15 = m > 14 && row_read( efirst ? 1 : 0, lane, 14 ) != 0;      // entry 14 (lane 14 of the row): the second round
    // entries 0 .. 13 move two lanes up
    const int  sh = row_read( eh, lane, ( slot + 14 ) & 15 ), sv = row_read( ev, lane, ( slot + 14 ) & 15 );
    const bool sf = row_read( efirst ? 1 : 0, lane, ( slot + 14 ) & 15 ) != 0;
    if( slot >= 2 ) { cx = sh; cy = sv; cv = sf; }
  }

  // lane constants of the diamond rounds (xTZ8PointDiamondSearch :504-705): the slot-th point of a round's list, as 4-bit fields
  //   bits  0 ..  7  distance 1:      (x, y) offsets of (left, top) (sx, top) (right, top) (left, sy) (right, sy) (left, bot) (sx, bot) (right, bot); the corners only with the
  //                                   extended settings -- never here; point number slot + 1
  //   bits  8 .. 15  distance 2 .. 8: in units of d / 2: (sx, top, 2) (left2, top2, 1) (right2, top2, 3) (left, sy, 4) (right, sy, 5) (left2, bot2, 6) (right2, bot2, 8) (sx, bot, 7)
  //   bits 16 .. 23  distance > 8:    in units of d / 4: (sx, top) (left, sy) (right, sy) (sx, bot), then for i = 1 .. 3: (xl, yt) (xr, yt) (xl, yb) (xr, yb) along the diamond's edges
  //   bits 24 .. 27  point number of the distance 2 .. 8 class; bit 28 / 29: the slot has a point at distance 1 / 2 .. 8; bit 30: that point reports distance d / 2
  unsigned lc;
  {
    const int  u1x = slot == 1 || slot == 6 ? 0 : ( slot == 0 || slot == 3 || slot == 5 ? -1 : 1 ), u1y = slot < 3 ? -1 : slot < 5 ? 0 : 1;
    int  u2x, u2y, n2;
    bool half;
    tzg_mid_point( slot, u2x, u2y, n2, half );
    const int  i4 = ( ( slot - 4 ) >> 2 ) + 1, m4 = ( slot - 4 ) & 3;
    const int  u4x = slot < 4 ? ( slot == 1 ? -4 : slot == 2 ? 4 : 0 ) : ( ( m4 & 1 ) ? i4 : -i4 );
    const int  u4y = slot < 4 ? ( slot == 0 ? -4 : slot == 3 ? 4 : 0 ) : ( ( m4 & 2 ) ? 4 - i4 : -( 4 - i4 ) );
    lc = ( unsigned ) ( u1x & 15 ) | ( unsigned ) ( u1y & 15 ) << 4 | ( unsigned ) ( u2x & 15 ) << 8 | ( unsigned ) ( u2y & 15 ) << 12 | ( unsigned ) ( u4x & 15 ) << 16 | ( unsigned ) ( u4y & 15 ) << 20
         | ( unsigned ) n2 << 24 | ( slot < 8 && ( u1x == 0 || u1y == 0 ) ? 1u << 28 : 0u ) | ( slot < 8 ? 1u << 29 : 0u ) | ( half ? 1u << 30 : 0u );
  }

  // SPECULATIVE first round.  The first diamond loop's rounds at distance 1 and 2 are always evaluated (xTZSearch stops it after three rounds without a new best point at the
  // earliest) and their points depend only on the start round's winner -- nearly always rcMv, the predictor.  So when the start round has at most four candidates the row's other
  // twelve lanes evaluate the distance-1 points (lanes 4 .. 7) and the distance-2 points (lanes 8 .. 15) around rcMv in the SAME round; if rcMv wins the start round, the two
  // diamond rounds are replayed from their minima with the reference's accept rule and the search goes on at distance 4: two dependent round trips less.  Otherwise the
  // speculative costs are dropped (and not counted) and the search proceeds as usual.
  TzState s;
  s.bestSad = ~0ull; s.bestX = 0; s.bestY = 0; s.pointNr = 0; s.bestDist = 0; s.bestRound = 0; s.nEval = 0;
  s.sr.left = s.sr.right = s.sr.top = s.sr.bottom = 0;
  bool spec = VTMHIP_TZG_SPEC && m <= 2 && searchRange >= 2 && j.tiny;      // (row-uniform)
  if( spec )
  {
    s.sr = search_range( j, mx << 4, my << 4, searchRange >> ( fast ? 1 : 0 ) );      // what the loop's entry computes when rcMv wins
    if( slot >= 4 )
    {
      int dx, dy;
      if( slot < 8 )      // (sx, top, 2) (left, sy, 4) (right, sy, 5) (sx, bot, 7)
      {
        const int t = slot - 4;
        dx = t == 1 ? -1 : t == 2 ? 1 : 0; dy = t == 0 ? -1 : t == 3 ? 1 : 0;
        cnr = t == 0 ? 2 : t == 1 ? 4 : t == 2 ? 5 : 7; cdist = 1;
      }
      else
      {
        bool half;
        tzg_mid_point( slot - 8, dx, dy, cnr, half );
        cdist = half ? 1 : 2;
      }
      cx = mx + dx; cy = my + dy;
      cv = ( dx >= 0 || cx >= s.sr.left ) && ( dx <= 0 || cx <= s.sr.right ) && ( dy >= 0 || cy >= s.sr.top ) && ( dy <= 0 || cy <= s.sr.bottom );
    }
  }

  job_sync<1>();      // the row's original block is in LDS (DS operations of a wave execute in order)

  int  phase = GP_START, d = 0, startX = 0, startY = 0;
  bool touch = true, listed = false;

  for( ;; )
  {
    // ---- the round: every lane its candidate, the row's first strict minimum, xTZSearchHelp's accept rule (:397-417)
    {
      const unsigned           sad = grp_sad<SPR, SS>( j, so, cv ? cx : s.bestX, cv ? cy : s.bestY );      // idle lanes re-read the best point (a valid address)
      const unsigned long long c   = ( ( unsigned long long ) sad << SS ) + mv_cost( j, cx, cy );
      const unsigned vrow = ( unsigned ) ( __ballot( cv ) >> ( lane & 48 ) ) & 0xffffu;      // the row's lanes with a candidate
      const int      myXy = ( cx & 0xffff ) | ( cy << 16 ), myNd = cnr | ( cdist << 8 );
      if( spec )      // (row-uniform; the first round only)
      {
        spec = false;
        unsigned k = cv ? ( ( unsigned ) c << 4 ) | ( unsigned ) slot : 0xffffffffu;      // (j.tiny: cost < 2^26)
        k = min( k, dpp_u32<DPP_XOR1>( k ) );
        k = min( k, dpp_u32<DPP_XOR2>( k ) );
        const unsigned kq = k;                                                              // minimum of the lane's quad
        const unsigned k8 = min( k, dpp_u32<DPP_HALF_MIRROR>( k ) );                        // ... of its eight lanes
        const unsigned kS = ( unsigned ) row_read( ( int ) kq, lane, 0 ), k1 = ( unsigned ) row_read( ( int ) kq, lane, 4 ), k2 = ( unsigned ) row_read( ( int ) k8, lane, 8 );
        const int xyS = row_read( myXy, lane, ( int ) ( kS & 15u ) );
        const int xy1 = row_read( myXy, lane, ( int ) ( k1 & 15u ) ), nd1 = row_read( myNd, lane, ( int ) ( k1 & 15u ) );
        const int xy2 = row_read( myXy, lane, ( int ) ( k2 & 15u ) ), nd2 = row_read( myNd, lane, ( int ) ( k2 & 15u ) );
        // the start round (slot 0 always has a candidate): best = its first strict minimum, distance / point number / round counter 0
        s.bestSad = kS >> 4; s.bestX = ( int ) ( short ) ( xyS & 0xffff ); s.bestY = xyS >> 16;
        s.nEval   = ( unsigned ) __popc( vrow & 0xfu );
        if( ( kS & 15u ) == 0 )
        {
          // rcMv won: the rounds at distance 1 and 2 around it, as the loop would have run them (round counter + 1, then the accept rule)
          startX = s.bestX; startY = s.bestY;
          s.nEval += ( unsigned ) __popc( vrow & 0xfff0u );
          s.bestRound = 1;
          if( k1 != 0xffffffffu && ( unsigned long long ) ( k1 >> 4 ) < s.bestSad )
          {
            s.bestSad = k1 >> 4; s.bestX = ( int ) ( short ) ( xy1 & 0xffff ); s.bestY = xy1 >> 16; s.bestDist = ( unsigned ) ( nd1 >> 8 ); s.bestRound = 0; s.pointNr = nd1 & 0xff;
          }
          s.bestRound += 1;
          if( k2 != 0xffffffffu && ( unsigned long long ) ( k2 >> 4 ) < s.bestSad )
          {
            s.bestSad = k2 >> 4; s.bestX = ( int ) ( short ) ( xy2 & 0xffff ); s.bestY = xy2 >> 16; s.bestDist = ( unsigned ) ( nd2 >> 8 ); s.bestRound = 0; s.pointNr = nd2 & 0xff;
          }
          phase = GP_DIA1; d = 2;      // (the round at distance 2 of the first loop is done)
        }
      }
      else
      {
      unsigned long long minCost;
      unsigned           minKey;
      row_argmin<true>( j, c, cv, ( unsigned ) slot, minCost, minKey );
      s.nEval += ( unsigned ) __popc( vrow );
      const int pxy = row_read( myXy, lane, ( int ) ( minKey & 15u ) );
      const int pnd = row_read( myNd, lane, ( int ) ( minKey & 15u ) );
      const bool acc = minCost < s.bestSad;      // (minCost == ~0: no candidate in the round)
      s.bestSad   = acc ? minCost : s.bestSad;
      s.bestX     = acc ? ( int ) ( short ) ( pxy & 0xffff ) : s.bestX;
      s.bestY     = acc ? pxy >> 16 : s.bestY;
      s.bestDist  = acc && touch ? ( unsigned ) ( pnd >> 8 ) : s.bestDist;
      s.bestRound = acc && touch ? 0u : s.bestRound;
      s.pointNr   = acc && touch ? ( pnd & 0xff ) : s.pointNr;
      }
    }
    // ---- what comes next (xTZSearch :3765-3971 without the extended settings), as conditions rather than branches: the four rows of a wave are in whatever phase each is in
    const bool isS = phase == GP_START, isS15 = phase == GP_START15, isD1 = phase == GP_DIA1, isT1 = phase == GP_TWO1, isD2 = phase == GP_DIA2, isT2 = phase == GP_TWO2;
    const bool more   = 2 * d <= searchRange;                                        // the diamond loops' `d *= 2` stays inside the range
    const bool end1   = isD1 && ( ( firstStop && s.bestRound >= 3 ) || !more );      // first loop (:3765-3772)
    const bool end2   = isD2 && ( ( fast && s.bestRound >= 2 ) || !more );           // a star refinement's loop (:3944-3951)
    const bool toS15  = isS && e15;
    const bool enter1 = ( isS && !e15 ) || isS15;
    if( enter1 )
    {
      s.sr   = search_range( j, s.bestX << 4, s.bestY << 4, searchRange >> ( fast ? 1 : 0 ) );
      startX = s.bestX; startY = s.bestY;
    }
    const bool after1 = end1 || ( enter1 && searchRange < 1 );
    const bool toT1   = after1 && s.bestDist == 1;                                   // :3790-3794
    const bool rdec   = ( after1 && !toT1 ) || isT1;                                 // the raster decision (:3872)
    const bool toT2   = end2 && s.bestDist == 1 && s.pointNr != 0;                   // :3953-3960
    if( toT1 || ( end2 && s.bestDist == 1 ) ) s.bestDist = 0;
    bool star = ( end2 && !toT2 ) || isT2;                                           // back at the star loop's condition (:3937)
    if( rdec )
    {
      if( ( int ) s.bestDist >= iRaster )
      {
        s.bestDist = ( unsigned ) iRaster;
        const int nx = s.sr.right >= s.sr.left ? ( s.sr.right - s.sr.left ) / iRaster + 1 : 0, ny = s.sr.bottom >= s.sr.top ? ( s.sr.bottom - s.sr.top ) / iRaster + 1 : 0;
        if( mode == 1 && iRaster == 5 && j.h <= 128 && nx >= 1 && ny >= 1 && nx * ny <= totCap )
        {
          // the scan runs in tz_raster_cols_kernel, the search resumes in tz_search_kernel<1> (mode 2): state and job record as that kernel stores them
          if( slot == 0 )
          {
            TzSaved sv;
            sv.sr = s.sr; sv.bestSad = s.bestSad; sv.bestX = s.bestX; sv.bestY = s.bestY; sv.pointNr = s.pointNr; sv.bestDist = s.bestDist;
            sv.bestRound = s.bestRound; sv.nEval = s.nEval; sv.rasterCost = ~0ull; sv.rasterIdx = 0; sv.pad = 0;
            saved[jobIdx] = sv;
            list[1 + atomicAdd( &list[0], 1 )] = jobIdx;
            // (the record again from the row, not kept in registers through the search)
            vtmhip_tz_job t2;
            mg::make_tz_job_scalars( fu.cfg, mj, fuse_pat_off( fu, mj ), fuse_pat_stride( fu, mj ), predH, predV, t2 );
            vtmhip_tz_job *g = fu.tzSpill + jobIdx;
            g->orgOff = t2.orgOff; g->refOff = t2.refOff; g->orgStride = t2.orgStride; g->refStride = t2.refStride; g->puX = t2.puX; g->puY = t2.puY;
            g->width = t2.width; g->height = t2.height; g->subShift = t2.subShift; g->imvShift = t2.imvShift; g->signedSamples = t2.signedSamples;
            g->predHor = t2.predHor; g->predVer = t2.predVer; g->motionLambda = t2.motionLambda; g->mvHor = t2.mvHor; g->mvVer = t2.mvVer; g->searchRange = t2.searchRange;
            g->extendedSettings = t2.extendedSettings; g->fastSettings = t2.fastSettings; g->firstSearchStop = t2.firstSearchStop; g->hasIntMv2Nx2NPred = t2.hasIntMv2Nx2NPred;
            g->intMv2Nx2NPredHor = t2.intMv2Nx2NPredHor; g->intMv2Nx2NPredVer = t2.intMv2Nx2NPredVer; g->numExtraStart = 0;
          }
          listed = true;
        }
        else
        {
          // the scan here (:3888-3899): lane `slot` takes the candidates slot, slot + 16, ... of the row-major raster; (cost, index) first strict minimum
          const int          total = nx * ny;
          unsigned long long bc = ~0ull;
          unsigned           bk = 0xffffffffu;
          for( int k = slot; k < total; k += 16 )
          {
            const int ry = k / nx, rx = k - ry * nx, x = s.sr.left + rx * iRaster, y = s.sr.top + ry * iRaster;
            const unsigned long long c = ( ( unsigned long long ) grp_sad<SPR, SS>( j, so, x, y ) << SS ) + mv_cost( j, x, y );
            if( c < bc ) { bc = c; bk = ( unsigned ) k; }
          }
          unsigned long long rc;
          unsigned           rk;
          row_argmin<false>( j, bc, bk != 0xffffffffu, bk, rc, rk );
          s.nEval += ( unsigned ) total;
          if( total > 0 && rc < s.bestSad )
          {
            const int ry = ( int ) rk / nx, rx = ( int ) rk - ry * nx;
            s.bestSad = rc; s.bestX = s.sr.left + rx * iRaster; s.bestY = s.sr.top + ry * iRaster; s.bestDist = ( unsigned ) iRaster; s.bestRound = 0; s.pointNr = 0;
          }
        }
      }
      star = true;
    }
    const bool go2 = star && !listed && s.bestDist > 0 && searchRange >= 1;      // another star refinement (:3937-3971)
    if( listed || ( star && !go2 ) ) break;
    startX     = go2 ? s.bestX : startX;
    startY     = go2 ? s.bestY : startY;
    s.bestDist = go2 ? 0u : s.bestDist;
    s.pointNr  = go2 ? 0 : s.pointNr;
    d          = ( enter1 || go2 ) ? 1 : ( ( isD1 && !end1 ) || ( isD2 && !end2 ) ) ? 2 * d : d;
    phase      = toS15 ? GP_START15 : toT1 ? GP_TWO1 : toT2 ? GP_TWO2 : go2 ? GP_DIA2 : enter1 ? GP_DIA1 : phase;
    touch      = !toS15;

    // ---- the next round's candidate of this lane
    if( phase == GP_DIA1 || phase == GP_DIA2 )
    {
      s.bestRound += 1;
      const int      cls = d == 1 ? 0 : d <= 8 ? 1 : 2;                         // the three forms of the pattern; their offsets count in units of 1, d / 2, d / 4
      const int      unit = d >> cls;
      const unsigned f = lc >> ( cls * 8 );
      const int      dx = ( ( int ) ( f << 28 ) >> 28 ) * unit, dy = ( ( int ) ( f << 24 ) >> 28 ) * unit;
      cx    = startX + dx; cy = startY + dy;
      cnr   = cls == 0 ? slot + 1 : cls == 1 ? ( int ) ( ( lc >> 24 ) & 15u ) : 0;
      cdist = cls == 1 && ( lc & ( 1u << 30 ) ) ? d >> 1 : d;
      const bool on = cls == 0 ? ( lc & ( 1u << 28 ) ) != 0 : cls == 1 ? ( lc & ( 1u << 29 ) ) != 0 : true;
      // a point is tested when it lies inside the search range on the side(s) it moved to (the checks of :510-700, which test exactly those sides)
      cv = on && ( dx >= 0 || cx >= s.sr.left ) && ( dx <= 0 || cx <= s.sr.right ) && ( dy >= 0 || cy >= s.sr.top ) && ( dy <= 0 || cy <= s.sr.bottom );
    }
    else if( phase == GP_START15 )
    {
      // m_uniMvList entry 14, on the row's first lane
      int ex = mj.extraStart[14][0], ey = mj.extraStart[14][1];
      clip_mv( j, ex, ey );
      cx = prec_down( ex, 4 ); cy = prec_down( ey, 4 ); cnr = 0; cdist = 0; cv = slot == 0;
    }
    else
    {
      // xTZ2PointSearch (:426-446): the two untested neighbours of the best point, by the point number of the distance-1 round; 2-bit fields (offset + 1) per point number 0 .. 8
      constexpr unsigned XO0 = 1u | 0u << 2 | 0u << 4 | 1u << 6 | 0u << 8 | 2u << 10 | 0u << 12 | 0u << 14 | 2u << 16;
      constexpr unsigned XO1 = 1u | 1u << 2 | 2u << 4 | 2u << 6 | 0u << 8 | 2u << 10 | 1u << 12 | 2u << 14 | 1u << 16;
      constexpr unsigned YO0 = 1u | 1u << 2 | 0u << 4 | 0u << 6 | 2u << 8 | 0u << 10 | 1u << 12 | 2u << 14 | 1u << 16;
      constexpr unsigned YO1 = 1u | 0u << 2 | 0u << 4 | 1u << 6 | 0u << 8 | 2u << 10 | 2u << 12 | 2u << 14 | 2u << 16;
      const int      sh2 = 2 * s.pointNr;
      const unsigned xo = slot == 0 ? XO0 : XO1, yo = slot == 0 ? YO0 : YO1;
      cx = s.bestX + ( int ) ( ( xo >> sh2 ) & 3u ) - 1;
      cy = s.bestY + ( int ) ( ( yo >> sh2 ) & 3u ) - 1;
      cnr = 0; cdist = 2;
      cv = slot < 2 && cx >= s.sr.left && cx <= s.sr.right && cy >= s.sr.top && cy <= s.sr.bottom;
    }
  }

  if( !listed && slot == 0 )
  {
    vtmhip_me_result r;
    r.mvX = s.bestX; r.mvY = s.bestY; r.nEval = s.nEval; r.reserved = 0;
    r.cost = s.bestSad;
    r.dist = s.bestSad - mv_cost( j, s.bestX, s.bestY );
    results[jobIdx] = r;
  }
}

// ---- exhaustive search (InterSearch::xPatternSearch :3566-3608 after xSetSearchRange :3496-3563): the bi-predictive
// refinement of xMotionEstimation (:3385-3440, +-BipredSearchRange around the current vector).  One wave per job.
template<int WPJ>
__global__ __launch_bounds__( WPJ == 1 ? 256 : 64 * WPJ ) void full_search_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase,
                                                                                  const int16_t *__restrict__ refBase,
                                                                                  const vtmhip_full_job *__restrict__ jobs, int numJobs,
                                                                                  vtmhip_me_result *__restrict__ results, FullFuse fu )
{
  constexpr int JOBS_PER_BLOCK = WPJ == 1 ? 4 : 1;
  __shared__ unsigned long long sRedCost[WPJ];
  __shared__ unsigned           sRedIdx[WPJ];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sOrgLds[WPJ >= 4 ? ORG_LDS_CAP : WPJ == 2 ? ORG_LDS_CAP2 : 8];
  __shared__ int4               sPts[JOBS_PER_BLOCK][16];   // fused: the start candidates (integer positions) ...
  __shared__ int                sStart[JOBS_PER_BLOCK][16][2];   // ... and the vectors they came from (the winner's UNCLIPPED vector is the search centre)
  const int lane = threadIdx.x & 63, wv = uni( ( int ) ( threadIdx.x >> 6 ) );
  const int blk    = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x );
  const int jobIdx = WPJ == 1 ? blk * 4 + wv : blk;
  if( jobIdx >= numJobs ) return;
  // the job's fields in registers (wave-uniform).  FUSED (fu.me != nullptr, mest_glue.hpp): from the bi row itself -- what mest_prepare_kernel / mest_bi_start_kernel wrote to a table
  vtmhip_full_job fj;
  if( fu.me ) mg::make_full_job( fu, fu.me[jobIdx], fu.me[jobIdx].mvHor, fu.me[jobIdx].mvVer, fj );
  else
  {
    const vtmhip_full_job *gp = jobs + jobIdx;
    if( uni( ( int ) gp->width ) == 0 ) return;   // empty slot (see tz_search_kernel)
    fj = *gp;
  }
  const vtmhip_full_job *jp = &fj;
  Coop                   co;
  co.lane = lane; co.wave = WPJ == 1 ? 0 : wv; co.wpj = WPJ; co.leader = ( lane == 0 && co.wave == 0 ); co.redCost = sRedCost; co.redIdx = sRedIdx;
  MeJob j;
  j.org = orgBase + jp->orgOff; j.ref = refBase + jp->refOff;
  j.orgStride = jp->orgStride; j.refStride = jp->refStride;
  j.w = jp->width; j.h = jp->height; j.ss = jp->subShift; j.imvShift = jp->imvShift;
  j.predHor = jp->predHor; j.predVer = jp->predVer; j.costScale = 2; j.lambda = jp->motionLambda;
  j.horMax = ( pic.picW + 8 - jp->puX - 1 ) << 4;
  j.horMin = ( -pic.ctuSize - 8 - jp->puX + 1 ) << 4;
  j.verMax = ( pic.picH + 8 - jp->puY - 1 ) << 4;
  j.verMin = ( -pic.ctuSize - 8 - jp->puY + 1 ) << 4;
  j.seg        = ( j.w & 7 ) == 0 ? 8 : 4;
  j.segsPerRow = j.seg == 8 ? j.w >> 3 : j.w >> 2;
  j.items      = j.segsPerRow * ( ( j.h + ( 1 << j.ss ) - 1 ) >> j.ss );
  j.lpc        = 1;
  while( j.lpc < 64 && ( j.lpc << 1 ) <= j.items ) j.lpc <<= 1;
  j.bias       = jp->signedSamples ? 0x80008000u : 0u;
  j.lpcShift   = floor_log2_u( ( unsigned ) j.lpc );
  j.sprShift   = ( j.segsPerRow & ( j.segsPerRow - 1 ) ) == 0 ? floor_log2_u( ( unsigned ) j.segsPerRow ) : -1;
  j.narrow     = j.lambda >= 0.0 && j.lambda * 126.0 < 2147483648.0;
  j.tiny       = false;

  j.orgResident = j.items == j.lpc;
  j.resOff      = 0;
  j.orgSeg[0] = j.orgSeg[1] = j.orgSeg[2] = j.orgSeg[3] = 0;
  if( j.orgResident )
  {
    const int  sub = lane & ( j.lpc - 1 );
    const int  r = j.sprShift >= 0 ? sub >> j.sprShift : sub / j.segsPerRow, x = ( sub - r * j.segsPerRow ) * j.seg;
    const long os = ( long ) j.orgStride << j.ss;
    if( j.seg == 8 )
    {
      const Pel8 a = *reinterpret_cast<const Pel8 *>( j.org + r * os + x );
#pragma unroll
      for( int k = 0; k < 4; k++ ) j.orgSeg[k] = a.v[k] ^ j.bias;
    }
    else
    {
      const Pel4 a = *reinterpret_cast<const Pel4 *>( j.org + r * os + x );
      j.orgSeg[0] = a.v[0] ^ j.bias; j.orgSeg[1] = a.v[1] ^ j.bias;
    }
    j.resOff = ( int ) ( r * ( ( long ) j.refStride << j.ss ) + x );
  }
  stage_org<WPJ, ( WPJ >= 4 ? ORG_LDS_CAP : ORG_LDS_CAP2 )>( j, sOrgLds, ( int ) threadIdx.x );

  int centerHor = jp->centerHor, centerVer = jp->centerVer;
  if( fu.me && !fu.noStart )
  {
    // the best start (:3377-3420): rcMv, then the distinct m_uniMvList entries, newest first -- clipped, at integer precision, SAD + vector rate, first strict minimum.
    // Lane i owns list entry i - 1 (lane 0: rcMv); the survivors close up in LDS and one candidate round picks the lexicographic (cost, index) minimum
    const vtmhip_me_job &mj = fu.me[jobIdx];
    const int m = mg::num_extra( mj );
    int4 *pts = sPts[WPJ == 1 ? wv : 0];
    int( *st )[2] = sStart[WPJ == 1 ? wv : 0];
    int eh = mj.mvHor, ev = mj.mvVer;
    if( lane >= 1 && lane <= m ) { eh = mj.extraStart[lane - 1][0]; ev = mj.extraStart[lane - 1][1]; }
    bool first = lane <= m;
#pragma unroll
    for( int k = 1; k < 15; k++ )      // (entry k - 1 against the later entries; rcMv itself is always a candidate and never removes one)
    {
      const int oh = __shfl( eh, k, 64 ), ov = __shfl( ev, k, 64 );
      if( k < lane && oh == eh && ov == ev ) first = false;
    }
    const unsigned long long keep = __ballot( first );
    if( first && ( WPJ == 1 || wv == 0 ) )
    {
      const int pos = __popcll( keep & ( ( 1ull << lane ) - 1ull ) );
      int ch = eh, cv = ev;
      clip_mv( j, ch, cv );
      pts[pos] = make_int4( mg::prec_down( ch, 4 ), mg::prec_down( cv, 4 ), 0, 0 );
      st[pos][0] = eh; st[pos][1] = ev;
    }
    job_sync<WPJ>();
    unsigned long long c0;
    unsigned           i0;
    eval_candidates<false, WPJ>( j, pts, __popcll( keep ), 0, 0, 1, 1, co, c0, i0 );
    centerHor = st[i0 & 15][0]; centerVer = st[i0 & 15][1];
  }
  const Range sr = search_range( j, centerHor, centerVer, jp->searchRange );
  const int   nx = sr.right >= sr.left ? sr.right - sr.left + 1 : 0, ny = sr.bottom >= sr.top ? sr.bottom - sr.top + 1 : 0;
  unsigned long long cost;
  unsigned           idx;
  eval_candidates<true, WPJ>( j, nullptr, nx * ny, sr.left, sr.top, nx > 0 ? nx : 1, 1, co, cost, idx );
  if( co.leader )
  {
    vtmhip_me_result r;
    r.mvX = 0; r.mvY = 0; r.nEval = ( unsigned ) ( nx * ny ); r.reserved = 0; r.cost = ~0ull; r.dist = ~0ull;
    if( nx * ny > 0 )
    {
      const int ry = ( int ) idx / nx, rx = ( int ) idx - ry * nx;
      r.mvX  = sr.left + rx;
      r.mvY  = sr.top + ry;
      r.cost = cost;
      r.dist = cost - mv_cost( j, r.mvX, r.mvY );
    }
    results[jobIdx] = r;
  }
}

// ---- exhaustive search of uniform W x H blocks (squares 8 .. 64 and the binary / ternary split shapes between them), +-4: one LANE per candidate ----
// The cooperative kernel above spends a wave-step (8 or 4 candidates) on ~110 instructions; here the (W+8) x (H+8) reference window and
// the original block of every job of the workgroup are staged in LDS once and each lane owns one of the <= 81 candidates: W*H/2
// v_sad_u16 over dword LDS reads (odd columns through v_alignbit), then the MV rate and an LDS arg-min per job.
template<int W, int H> struct FullSq
{
  static constexpr int AREA    = W * H;
  static constexpr int JPB     = AREA <= 64 ? 12 : AREA <= 128 ? 9 : AREA <= 256 ? 6 : AREA <= 512 ? 4 : AREA <= 1024 ? 3 : 1;   // jobs per workgroup (LDS: windows + blocks <= 64 KB)
  static constexpr int THREADS = AREA <= 1024 ? 256 : 128;                     // 81 candidates per job
};

template<int W, int H>
__global__ __launch_bounds__( ( FullSq<W, H>::THREADS ) ) void full_search_sq_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                               const vtmhip_full_job *__restrict__ jobs, int numJobs, vtmhip_me_result *__restrict__ results, FullFuse fu )
{
  constexpr int JPB = FullSq<W, H>::JPB, THREADS = FullSq<W, H>::THREADS, WLD = W + 8, WLR = H + 8, WD = WLD / 2 + 1, MAXC = 81;
  __shared__ vtmhip_full_job    sJob[JPB];              // the workgroup's job records: copied from the table, or (FUSED, fu.me != nullptr) built from the bi rows with rcMv as the centre
  __shared__ unsigned           sWin[JPB][WLR][WD];     // reference window, two samples per dword, one spare dword per row
  __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned sOrg[JPB][H][W / 2];
  __shared__ int                sRange[JPB][5];          // left, top, nx, ny, floor((2^32 - 1) / nx)
  __shared__ unsigned long long sBest[JPB];
  __shared__ unsigned           sIdx[JPB];
  const int tid  = threadIdx.x;
  const int job0 = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x ) * JPB;
  const int nj   = min( JPB, numJobs - job0 );
  if( nj <= 0 ) return;

  MeJob j;   // per-lane view of job `jl` (set below); only the fields mv_cost / search_range read
  auto  view = [&]( const vtmhip_full_job &q ) {
    j.ss = q.subShift; j.imvShift = q.imvShift; j.predHor = q.predHor; j.predVer = q.predVer; j.costScale = 2; j.lambda = q.motionLambda;
    j.horMax = ( pic.picW + 8 - q.puX - 1 ) << 4; j.horMin = ( -pic.ctuSize - 8 - q.puX + 1 ) << 4;
    j.verMax = ( pic.picH + 8 - q.puY - 1 ) << 4; j.verMin = ( -pic.ctuSize - 8 - q.puY + 1 ) << 4;
    j.narrow = j.lambda >= 0.0 && j.lambda * 126.0 < 2147483648.0; j.tiny = false;
  };
  if( tid < nj )
  {
    if( fu.me ) mg::make_full_job( fu, fu.me[job0 + tid], fu.me[job0 + tid].mvHor, fu.me[job0 + tid].mvVer, sJob[tid] );
    else sJob[tid] = jobs[job0 + tid];
    const vtmhip_full_job &q = sJob[tid];
    view( q );
    const Range sr = search_range( j, q.centerHor, q.centerVer, q.searchRange );
    sRange[tid][0] = sr.left; sRange[tid][1] = sr.top;
    int nx = sr.right >= sr.left ? sr.right - sr.left + 1 : 0, ny = sr.bottom >= sr.top ? sr.bottom - sr.top + 1 : 0;
    if( nx > 9 || ny > 9 || q.width != W || q.height != H ) nx = ny = 0;   // the caller's promise is broken: no candidates, cost stays ~0 (never touch memory outside the window)
    sRange[tid][2] = nx;
    sRange[tid][3] = ny;
    sRange[tid][4] = ( int ) ( 0xffffffffu / ( unsigned ) max( nx, 1 ) );   // one division per job instead of one per candidate
    sBest[tid] = ~0ull; sIdx[tid] = 0xffffffffu;
  }
  __syncthreads();
  // stage windows (rows top .. top + ny + H - 2, columns left .. left + nx + W - 2) and original blocks, two samples per thread and step
  for( int i = tid; i < nj * WLR * ( WLD / 2 ); i += THREADS )
  {
    const int jl = i / ( WLR * ( WLD / 2 ) ), rem = i - jl * ( WLR * ( WLD / 2 ) ), r = rem / ( WLD / 2 ), c2 = ( rem - r * ( WLD / 2 ) ) * 2;
    const int nx = sRange[jl][2], ny = sRange[jl][3];
    unsigned  v = 0;
    if( r < ny + H - 1 && c2 < nx + W - 1 )
    {
      const vtmhip_full_job &q = sJob[jl];
      const int16_t *p = refBase + q.refOff + ( long ) ( sRange[jl][1] + r ) * q.refStride + ( sRange[jl][0] + c2 );
      const unsigned lo = ( unsigned short ) p[0], hi = c2 + 1 < nx + W - 1 ? ( unsigned short ) p[1] : 0u;
      v = ( lo | ( hi << 16 ) ) ^ ( q.signedSamples ? 0x80008000u : 0u );
    }
    sWin[jl][r][c2 >> 1] = v;
  }
  for( int i = tid; i < nj * WLR; i += THREADS ) sWin[i / WLR][i % WLR][WD - 1] = 0;
  for( int i = tid; i < nj * H * ( W / 2 ); i += THREADS )
  {
    const int jl = i / ( H * ( W / 2 ) ), rem = i - jl * ( H * ( W / 2 ) ), r = rem / ( W / 2 ), c2 = ( rem - r * ( W / 2 ) ) * 2;
    const vtmhip_full_job &q = sJob[jl];
    const int16_t *p = orgBase + q.orgOff + ( long ) r * q.orgStride + c2;
    sOrg[jl][r][c2 >> 1] = ( ( unsigned ) ( unsigned short ) p[0] | ( ( unsigned ) ( unsigned short ) p[1] << 16 ) ) ^ ( q.signedSamples ? 0x80008000u : 0u );
  }
  __syncthreads();

  // one candidate per lane; two passes over the arg-min: cost first, then the smallest index among equal costs (raster order = the
  // reference's first strict minimum)
  unsigned long long myCost[( JPB * MAXC + THREADS - 1 ) / THREADS];
#pragma unroll
  for( int pass = 0; pass < ( JPB * MAXC + THREADS - 1 ) / THREADS; pass++ )
  {
    const int it = tid + pass * THREADS;
    myCost[pass] = ~0ull;
    if( it < nj * MAXC )
    {
      const int jl = it / MAXC, k = it - jl * MAXC;
      const int nx = sRange[jl][2], ny = sRange[jl][3];
      if( k < nx * ny )
      {
        const vtmhip_full_job &q = sJob[jl];
        view( q );
        int cy = ( int ) __umulhi( ( unsigned ) k, ( unsigned ) sRange[jl][4] );   // k / nx: the multiply-high is the quotient or one less (k < 81)
        cy += ( cy + 1 ) * nx <= k ? 1 : 0;
        const int      cx = k - cy * nx;
        const unsigned sh = ( unsigned ) ( cx & 1 ) << 4;
        unsigned       s = 0;
        const int      step = 1 << j.ss;
#pragma unroll 2
        for( int r = 0; r < H; r += step )
        {
          const unsigned *w = &sWin[jl][cy + r][cx >> 1];
          const unsigned *o = &sOrg[jl][r][0];
          unsigned        d[W / 2 + 1];
#pragma unroll
          for( int m = 0; m <= W / 2; m++ ) d[m] = w[m];
#pragma unroll
          for( int m = 0; m < W / 2; m++ ) s = sad2( o[m], __builtin_amdgcn_alignbit( d[m + 1], d[m], sh ), s );
        }
        const int x = sRange[jl][0] + cx, y = sRange[jl][1] + cy;
        myCost[pass] = ( ( unsigned long long ) s << j.ss ) + mv_cost( j, x, y );
        atomicMin( &sBest[jl], myCost[pass] );
      }
    }
  }
  __syncthreads();
#pragma unroll
  for( int pass = 0; pass < ( JPB * MAXC + THREADS - 1 ) / THREADS; pass++ )
  {
    const int it = tid + pass * THREADS;
    if( it < nj * MAXC )
    {
      const int jl = it / MAXC, k = it - jl * MAXC;
      if( myCost[pass] == sBest[jl] ) atomicMin( &sIdx[jl], ( unsigned ) k );
    }
  }
  __syncthreads();
  if( tid < nj )
  {
    const vtmhip_full_job &q = sJob[tid];
    view( q );
    const int nx = sRange[tid][2], ny = sRange[tid][3];
    vtmhip_me_result r;
    r.mvX = 0; r.mvY = 0; r.nEval = ( unsigned ) ( nx * ny ); r.reserved = 0; r.cost = ~0ull; r.dist = ~0ull;
    if( nx * ny > 0 )
    {
      const int idx = ( int ) sIdx[tid], ry = idx / nx, rx = idx - ry * nx;
      r.mvX  = sRange[tid][0] + rx;
      r.mvY  = sRange[tid][1] + ry;
      r.cost = sBest[tid];
      r.dist = r.cost - mv_cost( j, r.mvX, r.mvY );
    }
    results[job0 + tid] = r;
  }
}

}   // namespace

extern "C" int vtmhip_tz_search_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                           const vtmhip_tz_job *d_jobs, int n, vtmhip_me_result *d_results )
{
  return vtmhip_internal_tz_search( ctx, pic, d_orgBase, d_refBase, d_jobs, n, d_results, nullptr, 0, 0 );
}

// fuse != nullptr: the searches of the xMotionEstimation rows fuse->me (job records built in the kernel's prologue, mest_glue.hpp); d_jobs is then the table the searches that
// go to the raster kernel store their record in (fuse->tzSpill == d_jobs)
int vtmhip_internal_tz_search( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_tz_job *d_jobs, int n,
                               vtmhip_me_result *d_results, const MeFuse *fuse, int uniformW, int uniformH )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "pic / n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && d_jobs && d_results, "null pointer" );
  MeFuse fuNone; memset( &fuNone, 0, sizeof( fuNone ) );
  const MeFuse fuFirst = fuse ? *fuse : fuNone;      // the first launch (mode 0 / 1) builds the records; the resume launch (mode 2) reads the stored ones
  VTMHIP_REQUIRE( ctx, pic->picW > 0 && pic->picH > 0 && pic->ctuSize > 0, "picture parameters" );
  // wavesPerJob: tuning hint for the whole batch (0 / 1: a wave per search -- small blocks, short candidate lists;
  // 2..16: that many waves split every candidate list -- large blocks, raster scans)
  const int wpj = pic->wavesPerJob;
  VTMHIP_REQUIRE( ctx, wpj == 0 || wpj == 1 || wpj == 2 || wpj == 4 || wpj == 8 || wpj == 16, "wavesPerJob must be 0, 1, 2, 4, 8 or 16" );
  // split search (default; VTMHIP_TZ_SPLIT=0: one launch): the raster scans of the batch run in tz_raster_cols_kernel between two launches of the search kernel
  static const bool split = !( getenv( "VTMHIP_TZ_SPLIT" ) && atoi( getenv( "VTMHIP_TZ_SPLIT" ) ) == 0 );
  // the raster column kernel keeps a scan's totals in LDS: 39 x 39 points (SearchRange 96) by default, ((2 * range) / 5 + 1)^2 for the caller's maxSearchRange hint (384: 154 x 154 = 93 KB)
  // The hint never makes the call fail: the totals of one scan must fit the CU's LDS beside the kernel's static arrays (160 KB on gfx950: ( 160 KB - 1 KB ) / 4 - 1 = 40 703
  // points = a 201 x 201 scan = search range 500); scans with more points than totCap are not listed for the column kernel and run inside the search kernel (mode 1 checks totCap).
  constexpr int RASTER_TOT_MAX = ( 160 * 1024 - 1024 ) / ( int ) sizeof( unsigned ) - 1;
  const int rasterSide = pic->maxSearchRange > 96 ? ( 2 * ( pic->maxSearchRange < 512 ? pic->maxSearchRange : 512 ) ) / 5 + 1 : 0;
  const int totWant = rasterSide * rasterSide > RASTER_TOT_CAP ? rasterSide * rasterSide : RASTER_TOT_CAP;
  const int totCap = totWant < RASTER_TOT_MAX ? totWant : RASTER_TOT_MAX;
  TzSaved  *d_saved = nullptr;
  int      *d_list  = nullptr;
  unsigned *d_tot   = nullptr;
  int       rasterParts = 1;
  if( split )
  {
    // few searches (one band of a sharded picture): several workgroups per raster scan, so that the scans of a 128x128 level still fill the GPU
    static const bool splitScan = !( getenv( "VTMHIP_RASTER_PARTS" ) && atoi( getenv( "VTMHIP_RASTER_PARTS" ) ) == 0 );
    rasterParts = ( splitScan && n <= 640 ) ? ( 1280 / n < 8 ? 1280 / n : 8 ) : 1;
    if( totCap > RASTER_TOT_CAP && splitScan && n <= 4096 && rasterParts < 4 ) rasterParts = 4;      // big scans (one workgroup per CU by LDS): a few workgroups per scan balance the tail   // aim at the 1280 resident workgroups (5 per CU); twice that measured slower
    const size_t oList = ( ( size_t ) n * sizeof( TzSaved ) + 255 ) & ~( size_t ) 255;
    const size_t oTot  = ( oList + ( ( size_t ) n + 1 ) * sizeof( int ) + 255 ) & ~( size_t ) 255;
    // global totals of the split scans (rasterParts > 1): n x (totCap + 1) words, zeroed per call -- bounded to 64 MB (big scans of many jobs keep one workgroup per scan)
    if( rasterParts > 1 && ( size_t ) n * ( totCap + 1 ) * sizeof( unsigned ) > ( ( size_t ) 64 << 20 ) ) rasterParts = 1;
    const size_t totBytes = rasterParts > 1 ? ( size_t ) n * ( totCap + 1 ) * sizeof( unsigned ) : 0;
    void        *arena = nullptr;
    int          st    = vtmhip_internal_workspace( ctx, oTot + totBytes, &arena, 1 );
    if( st ) return st;
    d_saved = ( TzSaved * ) arena;
    d_list  = ( int * ) ( ( char * ) arena + oList );
    d_tot   = ( unsigned * ) ( ( char * ) arena + oTot );
    // the list counter and (split scans) the global totals in ONE fill: they sit next to each other in the arena (a fill is a launch on the dependent chain of the call)
    if( totBytes ) VTMHIP_HIP( ctx, hipMemsetAsync( d_list, 0, ( size_t ) ( ( char * ) d_tot - ( char * ) d_list ) + totBytes, ctx->stream ) );
    else VTMHIP_HIP( ctx, hipMemsetAsync( d_list, 0, sizeof( int ), ctx->stream ) );
  }
#define VTMHIP_TZ_LAUNCH( W, GRID, MODE ) \
  hipLaunchKernelGGL( tz_search_kernel<W>, dim3( GRID ), dim3( W == 1 ? 256 : 64 * W ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, d_results, MODE, d_saved, d_list, totCap, \
                      ( MODE ) == 2 ? fuNone : fuFirst )
#define VTMHIP_TZ_SWITCH( MODE )                                    \
  switch( wpj )                                                     \
  {                                                                 \
  case 2: VTMHIP_TZ_LAUNCH( 2, n, MODE ); break;                    \
  case 4: VTMHIP_TZ_LAUNCH( 4, n, MODE ); break;                    \
  case 8: VTMHIP_TZ_LAUNCH( 8, n, MODE ); break;                    \
  case 16: VTMHIP_TZ_LAUNCH( 16, n, MODE ); break;                  \
  default: VTMHIP_TZ_LAUNCH( 1, ( ( MODE ) == 2 && ( n + 3 ) / 4 > 2048 ? 2048 : ( n + 3 ) / 4 ), MODE ); break; \
  }
  // Uniform batches of small blocks (fused uni rows, no extended settings): four searches per wave, a lane per candidate (tz_group_kernel).  VTMHIP_TZ_GROUP=0: off;
  // VTMHIP_TZ_GROUP_ITEMS: the largest block in 8-sample segments after row sub-sampling (default 32: 8x8 .. 16x16 and the 32x16 / 16x32 split shapes -- `--partition btt` 17.57 -> 17.49 ms;
  // 32x32, 64 segments, is faster with a wave per search: 0.21 against 0.26 ms)
  static const bool groupOn = !( getenv( "VTMHIP_TZ_GROUP" ) && atoi( getenv( "VTMHIP_TZ_GROUP" ) ) == 0 );
  static const int  groupItems = getenv( "VTMHIP_TZ_GROUP_ITEMS" ) ? atoi( getenv( "VTMHIP_TZ_GROUP_ITEMS" ) ) : 32;
  int grpItems = 0, grpSs = 0;
  if( groupOn && fuse && fuse->me && !fuse->cfg.extendedSettings && ( wpj == 0 || wpj == 1 ) && ( uniformW == 8 || uniformW == 16 || uniformW == 32 ) && uniformH >= 8 )
  {
    const int ss = fuse->cfg.fastInterSearchMode13 && uniformH > 8 && uniformW <= 64 ? 1 : 0;      // mg::sub_shift
    const int items = ( uniformW >> 3 ) * ( ( uniformH + ( 1 << ss ) - 1 ) >> ss );
    if( ( items & 7 ) == 0 && items <= groupItems && items <= 128 ) { grpItems = items; grpSs = ss; }
  }
  if( grpItems )
  {
    VTMHIP_TIME_KERNEL( ctx, "tz_group_kernel" );
    const int    spr = uniformW >> 3;
    const size_t lds = ( size_t ) 16 * grpItems * 16;
    const dim3   grid( ( n + 15 ) / 16 );
    const int    md = split ? 1 : 0;
#define VTMHIP_TZG_LAUNCH( SPR, SS )                                                                                                                                        \
  {                                                                                                                                                                         \
    if( lds > 48 * 1024 ) VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( tz_group_kernel<SPR, SS> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) ); \
    hipLaunchKernelGGL( ( tz_group_kernel<SPR, SS> ), grid, dim3( 256 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, n, d_results, md, d_saved, d_list, totCap, grpItems, fuFirst ); \
  }
    if( spr == 1 && grpSs == 0 ) VTMHIP_TZG_LAUNCH( 1, 0 )
    else if( spr == 1 ) VTMHIP_TZG_LAUNCH( 1, 1 )
    else if( spr == 2 && grpSs == 0 ) VTMHIP_TZG_LAUNCH( 2, 0 )
    else if( spr == 2 ) VTMHIP_TZG_LAUNCH( 2, 1 )
    else if( grpSs == 0 ) VTMHIP_TZG_LAUNCH( 4, 0 )
    else VTMHIP_TZG_LAUNCH( 4, 1 )
#undef VTMHIP_TZG_LAUNCH
  }
  else
  { VTMHIP_TIME_KERNEL( ctx, "tz_search_kernel" );
    VTMHIP_TZ_SWITCH( split ? 1 : 0 )
  }
  if( split )
  {
    { VTMHIP_TIME_KERNEL( ctx, "tz_raster_cols_kernel" );
      const size_t totLds = ( size_t ) ( totCap + 1 ) * sizeof( unsigned );
      if( totLds > 32 * 1024 ) VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( tz_raster_cols_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) totLds ) );
      hipLaunchKernelGGL( tz_raster_cols_kernel, dim3( rasterParts > 1 ? n * rasterParts : ( n < 3072 ? n : 3072 ) ), dim3( 256 ), totLds, ctx->stream, *pic, d_orgBase,
                          d_refBase, d_jobs, d_saved, d_list, d_tot, rasterParts, totCap );
    }
    { VTMHIP_TIME_KERNEL( ctx, "tz_search_kernel" );
      VTMHIP_TZ_SWITCH( 2 )
    }
  }
#undef VTMHIP_TZ_SWITCH
#undef VTMHIP_TZ_LAUNCH
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}


static int full_search_uniform( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_full_job *d_jobs, int n,
                                int width, int height, vtmhip_me_result *d_results, const FullFuse &fu );

extern "C" int vtmhip_full_search_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                             const vtmhip_full_job *d_jobs, int n, vtmhip_me_result *d_results )
{
  return vtmhip_internal_full_search( ctx, pic, d_orgBase, d_refBase, d_jobs, n, 0, 0, d_results, nullptr );
}

// fuse != nullptr: the exhaustive searches of the bi rows fuse->me (job records -- and, unless fuse->noStart, the choice of the start vector -- in the kernel's prologue; d_jobs
// unused).  width / height != 0: the caller's promise of a uniform batch (the lane-per-candidate kernel when the shape has one, the batch has no start candidates to weigh and the
// range is +-4)
int vtmhip_internal_full_search( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_full_job *d_jobs, int n,
                                 int width, int height, vtmhip_me_result *d_results, const FullFuse *fuse )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "pic / n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && ( d_jobs || fuse ) && d_results, "null pointer" );
  FullFuse fu; memset( &fu, 0, sizeof( fu ) );
  if( fuse ) fu = *fuse;
  if( width && height && ( !fuse || ( fuse->noStart && fuse->bipredSearchRange <= 4 ) ) )
  {
    int st = full_search_uniform( ctx, pic, d_orgBase, d_refBase, d_jobs, n, width, height, d_results, fu );
    if( st != VTMHIP_E_INVALID + 1000 ) return st;      // (no lane-per-candidate kernel for this shape: the cooperative kernel below)
  }
  VTMHIP_REQUIRE( ctx, pic->picW > 0 && pic->picH > 0 && pic->ctuSize > 0, "picture parameters" );
  const int wpj = pic->wavesPerJob;
  VTMHIP_REQUIRE( ctx, wpj == 0 || wpj == 1 || wpj == 2 || wpj == 4 || wpj == 8 || wpj == 16, "wavesPerJob must be 0, 1, 2, 4, 8 or 16" );
#define VTMHIP_FS_LAUNCH( W, GRID ) \
  hipLaunchKernelGGL( full_search_kernel<W>, dim3( GRID ), dim3( W == 1 ? 256 : 64 * W ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, d_results, fu )
  { VTMHIP_TIME_KERNEL( ctx, "full_search_kernel" );
  switch( wpj )
  {
  case 2: VTMHIP_FS_LAUNCH( 2, n ); break;
  case 4: VTMHIP_FS_LAUNCH( 4, n ); break;
  case 8: VTMHIP_FS_LAUNCH( 8, n ); break;
  case 16: VTMHIP_FS_LAUNCH( 16, n ); break;
  default: VTMHIP_FS_LAUNCH( 1, ( n + 3 ) / 4 ); break;
  }
  }
#undef VTMHIP_FS_LAUNCH
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}


extern "C" int vtmhip_full_search_uniform_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                     const vtmhip_full_job *d_jobs, int n, int width, int height, vtmhip_me_result *d_results )
{
  return vtmhip_internal_full_search( ctx, pic, d_orgBase, d_refBase, d_jobs, n, width, height, d_results, nullptr );
}

static int full_search_uniform( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_full_job *d_jobs, int n,
                                int width, int height, vtmhip_me_result *d_results, const FullFuse &fu )
{
  VTMHIP_REQUIRE( ctx, pic->picW > 0 && pic->picH > 0 && pic->ctuSize > 0, "picture parameters" );
#define VTMHIP_FSQ_LAUNCH( WW, HH )                                                                                                                                     \
  case ( WW ) * 256 + ( HH ):                                                                                                                                           \
    hipLaunchKernelGGL( ( full_search_sq_kernel<WW, HH> ), dim3( ( n + FullSq<WW, HH>::JPB - 1 ) / FullSq<WW, HH>::JPB ), dim3( FullSq<WW, HH>::THREADS ), 0, ctx->stream, \
                        *pic, d_orgBase, d_refBase, d_jobs, n, d_results, fu );                                                                                         \
    break;
  {
    VTMHIP_TIME_KERNEL( ctx, "full_search_sq_kernel" );
    switch( width * 256 + height )
    {
      VTMHIP_FSQ_LAUNCH( 8, 8 ) VTMHIP_FSQ_LAUNCH( 16, 16 ) VTMHIP_FSQ_LAUNCH( 32, 32 ) VTMHIP_FSQ_LAUNCH( 64, 64 )
      VTMHIP_FSQ_LAUNCH( 16, 8 ) VTMHIP_FSQ_LAUNCH( 8, 16 ) VTMHIP_FSQ_LAUNCH( 32, 8 ) VTMHIP_FSQ_LAUNCH( 8, 32 ) VTMHIP_FSQ_LAUNCH( 32, 16 ) VTMHIP_FSQ_LAUNCH( 16, 32 )
      VTMHIP_FSQ_LAUNCH( 64, 16 ) VTMHIP_FSQ_LAUNCH( 16, 64 ) VTMHIP_FSQ_LAUNCH( 64, 32 ) VTMHIP_FSQ_LAUNCH( 32, 64 )
    default: return VTMHIP_E_INVALID + 1000;   // other shapes: the caller takes the cooperative kernel
    }
  }
#undef VTMHIP_FSQ_LAUNCH
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

extern "C" int vtmhip_full_search_square_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                    const vtmhip_full_job *d_jobs, int n, int size, vtmhip_me_result *d_results )
{
  return vtmhip_full_search_uniform_batch_dev( ctx, pic, d_orgBase, d_refBase, d_jobs, n, size, size, d_results );
}
