// dist.hip -- distortion kernels: SAD / SATD (Hadamard) / SSE.
//
// Replaces RdCost::m_afpDistortFunc[DF_SAD*|DF_HAD*|DF_SSE*] (reference CommonLib/RdCost.cpp:493-1003, 2140-2934,
// 1783-2133; x86 versions CommonLib/x86/RdCostX86.h).  Integer results are bit-exact with the reference for any
// int16 input (diffs are formed in 32-bit; the rectangular-tile SATD normalisation is done in fp64 exactly as
// RdCost.cpp:2513,2654,2731,2814 does).
//
// Kernels (wave = 64 lanes, gfx950):
//   dist_batch_kernel   one wave per (org block, candidate block) job, arbitrary W x H -- hooks B1-B7/B10.
//   satd8_grid_kernel   8x8 SATD of every aligned 8x8 block x (2r+1)^2 displacements: the reference tile of a
//                       workgroup is staged ONCE in LDS with coalesced 16-byte loads and re-used by all displacements
//                       (HBM traffic ~ one read of each picture; the kernel is integer-VALU bound).
#include <cstdlib>
#include "ctx.hpp"
#include "had.hpp"
#include "dist_block.hpp"

namespace
{

// ---- general batch: one wave per job ------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void dist_batch_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ curBase,
                                                           const vtmhip_dist_job *__restrict__ jobs, int n, unsigned long long *__restrict__ out )
{
  const int lane = threadIdx.x & 63;
  const int job  = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
  if( job >= n ) return;   // whole wave leaves together
  const vtmhip_dist_job j   = jobs[job];
  const int16_t        *org = orgBase + j.orgOff;
  const int16_t        *cur = curBase + j.curOff;
  const int             w = j.width, h = j.height, os = j.orgStride, cs = j.curStride;
  if( w == 0 ) return;   // empty slot of a multi-stage call: nothing read, nothing written
  const unsigned long long acc = wave_block_dist( j.kind, org, os, cur, cs, w, h, j.subShift, lane );
  if( lane == 0 ) out[job] = acc;
}

// ---- SATD 8x8 grid ---------------------------------------------------------------------------------------------------
// Workgroup = 256 threads = TBX x TBY org blocks; LDS holds the org tile and the reference tile (+r halo).
constexpr int GRID_TBX = 8;

template<int THREADS, int GRID_TBY>
__global__ __launch_bounds__( THREADS ) void satd8_grid_kernel( const int16_t *__restrict__ org, int orgStride, const int16_t *__restrict__ ref,
                                                           int refStride, int bw, int bh, int r, unsigned *__restrict__ out )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];
  const int nd = 2 * r + 1, nd2 = nd * nd;
  const unsigned magic1 = 0xffffffffu / ( unsigned ) nd, magic2 = 0xffffffffu / ( unsigned ) nd2;   // uniform: one division each per wave
  const int refW = GRID_TBX * 8 + 2 * r, refH = GRID_TBY * 8 + 2 * r;
  const int refLd = ( refW + 7 ) & ~7;   // keep rows 16-byte aligned
  int16_t  *sOrg = lds;                                   // [TBY*8][TBX*8]
  int16_t  *sRef = lds + GRID_TBY * 8 * GRID_TBX * 8;     // [refH][refLd]
  const int bx0 = blockIdx.x * GRID_TBX, by0 = blockIdx.y * GRID_TBY;

  unsigned wide = 0, wide10 = 0;   // any sample outside [0, 4095] in this workgroup's tiles -> 32-bit path; outside [0, 1023] -> three packed levels only
  // stage org tile: 32 rows x 64 samples = 256 x 16-byte vectors, one per thread (coalesced 128-byte rows)
  for( int i = threadIdx.x; i < GRID_TBY * 8 * 8; i += THREADS )
  {
    const int row = i >> 3, seg = i & 7;
    const int gy = by0 * 8 + row, gx = bx0 * 8 + seg * 8;
    int4      v  = make_int4( 0, 0, 0, 0 );
    if( gy < bh * 8 && gx < bw * 8 ) v = *reinterpret_cast<const int4 *>( org + ( long ) gy * orgStride + gx );
    *reinterpret_cast<int4 *>( sOrg + row * 64 + seg * 8 ) = v;
    wide |= ( unsigned ) ( v.x | v.y | v.z | v.w ) & 0xf000f000u;
    wide10 |= ( unsigned ) ( v.x | v.y | v.z | v.w ) & 0xfc00fc00u;
  }
  // stage reference tile: refH rows x refW samples starting at (bx0*8 - r, by0*8 - r); 2-byte granularity on the
  // global side (the halo start is not 16-byte aligned), dword stores on the LDS side
  const int needW = min( GRID_TBX, bw - bx0 ) * 8 + 2 * r, needH = min( GRID_TBY, bh - by0 ) * 8 + 2 * r;   // never read past the last block's halo
  for( int i = threadIdx.x; i < refH * ( refLd >> 1 ); i += THREADS )
  {
    const int      row = i / ( refLd >> 1 ), c2 = ( i - row * ( refLd >> 1 ) ) << 1;
    const int16_t *p   = ref + ( long ) ( by0 * 8 - r + row ) * refStride + ( bx0 * 8 - r + c2 );
    int16_t        a = 0, b = 0;
    if( row < needH && c2 < needW ) a = p[0];
    if( row < needH && c2 + 1 < needW ) b = p[1];
    const unsigned pk = ( unsigned ) ( unsigned short ) a | ( ( unsigned ) ( unsigned short ) b << 16 );
    *reinterpret_cast<unsigned *>( sRef + row * refLd + c2 ) = pk;
    wide |= pk & 0xf000f000u;
    wide10 |= pk & 0xfc00fc00u;
  }
  const bool packed = __syncthreads_or( ( int ) wide ) == 0, packed10 = __syncthreads_or( ( int ) wide10 ) == 0;

  const int pairs = GRID_TBX * GRID_TBY * nd2;
  for( int p = threadIdx.x; p < pairs; p += THREADS )
  {
    // p / nd2 and d / nd without the ~25-instruction division sequence: multiply-high by floor((2^32 - 1) / divisor) is the quotient or one less
    // for these small operands (p * nd2 < 2^32); one correction step makes it exact (also for a divisor of 1)
    int b = ( int ) __umulhi( ( unsigned ) p, magic2 );
    b += ( b + 1 ) * nd2 <= p ? 1 : 0;
    const int d = p - b * nd2;
    const int lby = b / GRID_TBX, lbx = b - lby * GRID_TBX;
    int       dy = ( int ) __umulhi( ( unsigned ) d, magic1 );
    dy += ( dy + 1 ) * nd <= d ? 1 : 0;
    const int dx = d - dy * nd;   // 0..2r
    if( bx0 + lbx >= bw || by0 + lby >= bh ) continue;
    const int16_t *o = sOrg + lby * 8 * 64 + lbx * 8;
    const int16_t *c = sRef + ( lby * 8 + dy ) * refLd + lbx * 8 + dx;
    if( packed )
    {
      // dword reads: the reference row starts at an even or odd sample -> 5 dwords and v_alignbit by 0 / 16 bits
      const unsigned sh = ( unsigned ) ( dx & 1 ) << 4;
      v2s            D[8][4];
#pragma unroll
      for( int y = 0; y < 8; y++ )
      {
        const uint4     ov = *reinterpret_cast<const uint4 *>( o + y * 64 );
        const unsigned *cr = reinterpret_cast<const unsigned *>( c + y * refLd - ( dx & 1 ) );
        const unsigned  r0 = cr[0], r1 = cr[1], r2 = cr[2], r3 = cr[3], r4 = cr[4];
        const unsigned  ow[4] = { ov.x, ov.y, ov.z, ov.w };
        const unsigned  cw[4] = { __builtin_amdgcn_alignbit( r1, r0, sh ), __builtin_amdgcn_alignbit( r2, r1, sh ), __builtin_amdgcn_alignbit( r3, r2, sh ),
                                  __builtin_amdgcn_alignbit( r4, r3, sh ) };
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          v2s ovv, cvv;
          __builtin_memcpy( &ovv, &ow[k], 4 );
          __builtin_memcpy( &cvv, &cw[k], 4 );
          D[y][k] = ovv - cvv;
        }
      }
      out[( ( long ) ( by0 + lby ) * bw + ( bx0 + lbx ) ) * nd2 + d] = packed10 ? satd8_packed10( D ) : satd8_packed( D );
      continue;
    }
    int            m[64];
#pragma unroll
    for( int y = 0; y < 8; y++ )
    {
#pragma unroll
      for( int x = 0; x < 8; x++ ) m[y * 8 + x] = ( int ) o[y * 64 + x] - ( int ) c[y * refLd + x];
    }
    out[( ( long ) ( by0 + lby ) * bw + ( bx0 + lbx ) ) * nd2 + d] = had_finish<8, 8>( m );
  }
}

// ---- uniform batch: every job is W x H of one kind -> several jobs per wave -----------------------------------------------------------
// dist_batch_kernel gives a wave to every job, so an 8x8 SATD (one tile) keeps 1 lane of 64 busy and an 8x8 SAD 16.  When the
// caller can promise one block size per launch (the merge / AMVP candidates of one CU size, hooks B7 / B10), LPJ = the power of two
// >= the job's work items (8-sample row segments, Hadamard tiles) lanes form a job and 64 / LPJ jobs share a wave.
struct __attribute__( ( packed, aligned( 2 ) ) ) DPel8 { unsigned v[4]; };

__global__ __launch_bounds__( 256 ) void dist_uniform_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ curBase,
                                                             const vtmhip_dist_job *__restrict__ jobs, int n, int kind, int w, int h, int ss, int lpjShift,
                                                             unsigned long long *__restrict__ out )
{
  const int lpj = 1 << lpjShift, g = 64 >> lpjShift;
  const int lane = threadIdx.x & 63, wave = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x ) * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
  const int job = wave * g + ( lane >> lpjShift ), sub = lane & ( lpj - 1 );
  const bool live = job < n;
  const vtmhip_dist_job j = jobs[live ? job : 0];
  const int16_t *org = orgBase + j.orgOff, *cur = curBase + j.curOff;
  const int os = j.orgStride, cs = j.curStride;
  unsigned long long acc = 0;
  if( kind == VTMHIP_DIST_SATD )
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
    for( int it = sub; live && it < tx * ty; it += lpj )
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
  else
  {
    // 8-sample segments (host side guarantees w % 8 == 0 for this kernel); rows stepped by 1 << ss for SAD
    const int segs = w >> 3, rows = kind == VTMHIP_DIST_SAD ? h >> ss : h, rs = kind == VTMHIP_DIST_SAD ? ss : 0;
    unsigned  s32 = 0;
    for( int it = sub; live && it < rows * segs; it += lpj )
    {
      const int   r = it / segs, x = ( it - r * segs ) << 3;
      const DPel8 a = *reinterpret_cast<const DPel8 *>( org + ( long ) ( r << rs ) * os + x );
      const DPel8 b = *reinterpret_cast<const DPel8 *>( cur + ( long ) ( r << rs ) * cs + x );
      if( kind == VTMHIP_DIST_SAD )
      {
#pragma unroll
        for( int k = 0; k < 4; k++ ) s32 = __builtin_amdgcn_sad_u16( a.v[k] ^ 0x80008000u, b.v[k] ^ 0x80008000u, s32 );   // sign bias: any int16 pair
      }
      else
      {
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const int d0 = ( int ) ( short ) ( a.v[k] & 0xffffu ) - ( int ) ( short ) ( b.v[k] & 0xffffu ), d1 = ( ( int ) a.v[k] >> 16 ) - ( ( int ) b.v[k] >> 16 );
          acc += ( unsigned long long ) ( ( unsigned ) d0 * ( unsigned ) d0 );
          acc += ( unsigned long long ) ( ( unsigned ) d1 * ( unsigned ) d1 );
        }
      }
    }
    if( kind == VTMHIP_DIST_SAD ) acc = ( unsigned long long ) s32 << ss;
  }
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 )
    if( o < lpj ) acc += __shfl_xor( acc, o, 64 );
  if( live && sub == 0 ) out[job] = acc;
}

// ---- masked SAD (GEO merge estimation): one wave per job ---------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void sad_mask_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ curBase, const int16_t *__restrict__ maskBase,
                                                         const vtmhip_masked_sad_job *__restrict__ jobs, int n, unsigned long long *__restrict__ out )
{
  const int lane = threadIdx.x & 63;
  const int job  = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
  if( job >= n ) return;
  const vtmhip_masked_sad_job j = jobs[job];
  const int16_t *org = orgBase + j.orgOff, *cur = curBase + j.curOff, *mask = maskBase + j.maskOff;
  const int  w = j.width, ss = j.subShift, rows = j.height >> ss;
  const long rowStep = ( long ) w * j.stepX + ( ( long ) j.maskStride << ss ) + j.maskStride2;   // mask pointer advance per (sub-sampled) row
  unsigned long long acc = 0;
  for( int it = lane; it < rows * w; it += 64 )
  {
    const int r = it / w, x = it - r * w;
    const int d = abs( ( int ) org[( long ) ( r << ss ) * j.orgStride + x] - ( int ) cur[( long ) ( r << ss ) * j.curStride + x] );
    acc += ( unsigned long long ) ( long long ) ( d * ( int ) mask[r * rowStep + ( long ) x * j.stepX] );
  }
  acc = wave_reduce_add_u64( acc );
  if( lane == 0 ) out[job] = acc << ss;
}

int check_dist_args( vtmhip_ctx *ctx, int w, int h, int subShift, int kind )
{
  VTMHIP_REQUIRE( ctx, w >= 1 && h >= 1 && w <= 128 && h <= 128, "block size must be 1..128" );
  VTMHIP_REQUIRE( ctx, kind != VTMHIP_DIST_SATD || ( ( ( w | h ) & 1 ) == 0 ), "SATD needs even width and height (RdCost.cpp:2925-2931: \"Invalid size\")" );
  VTMHIP_REQUIRE( ctx, subShift >= 0 && subShift <= 4 && ( kind == VTMHIP_DIST_SAD || subShift == 0 ), "subShift" );
  VTMHIP_REQUIRE( ctx, ( h & ( ( 1 << subShift ) - 1 ) ) == 0 || subShift == 0, "height must be a multiple of the row step" );
  return VTMHIP_OK;
}

// pointer-surface helper: stage both blocks compactly (stride = width) and run a batch of one
int dist_single( vtmhip_ctx *ctx, int kind, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift,
                 uint64_t *dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, org && cur && dist, "null pointer" );
  int st = check_dist_args( ctx, w, h, subShift, kind );
  if( st ) return st;
  const size_t blk = ( size_t ) w * h * sizeof( int16_t );
  const size_t jobOff = ( 2 * blk + 63 ) & ~( size_t ) 63, outOff = jobOff + 64;
  st = vtmhip_internal_scratch( ctx, outOff + 64 );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  for( int y = 0; y < h; y++ )
  {
    memcpy( hp + ( size_t ) y * w * 2, org + ( ptrdiff_t ) y * orgStride, ( size_t ) w * 2 );
    memcpy( hp + blk + ( size_t ) y * w * 2, cur + ( ptrdiff_t ) y * curStride, ( size_t ) w * 2 );
  }
  vtmhip_dist_job j;
  j.orgOff = 0; j.curOff = ( int64_t ) w * h; j.orgStride = w; j.curStride = w;
  j.width = ( int16_t ) w; j.height = ( int16_t ) h; j.subShift = ( int16_t ) subShift; j.kind = ( int16_t ) kind;
  memcpy( hp + jobOff, &j, sizeof( j ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, outOff, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( dist_batch_kernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ( const int16_t * ) dp, ( const int16_t * ) dp,
                      ( const vtmhip_dist_job * ) ( dp + jobOff ), 1, ( unsigned long long * ) ( dp + outOff ) );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + outOff, dp + outOff, 8, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  memcpy( dist, hp + outOff, 8 );
  return VTMHIP_OK;
}

}   // namespace

extern "C"
{

int vtmhip_xGetSAD( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, int subShift,
                    uint64_t *dist )
{
  return dist_single( ctx, VTMHIP_DIST_SAD, org, orgStride, cur, curStride, width, height, subShift, dist );
}

int vtmhip_xGetHADs( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, uint64_t *dist )
{
  return dist_single( ctx, VTMHIP_DIST_SATD, org, orgStride, cur, curStride, width, height, 0, dist );
}

int vtmhip_xGetSSE( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, uint64_t *dist )
{
  return dist_single( ctx, VTMHIP_DIST_SSE, org, orgStride, cur, curStride, width, height, 0, dist );
}

int vtmhip_dist_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const vtmhip_dist_job *d_jobs, int n,
                           uint64_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_curBase && d_jobs && d_dist, "null pointer" );
  const int wavesPerBlock = 4;
  hipLaunchKernelGGL( dist_batch_kernel, dim3( ( n + wavesPerBlock - 1 ) / wavesPerBlock ), dim3( 64 * wavesPerBlock ), 0, ctx->stream,
                      d_orgBase, d_curBase, d_jobs, n, ( unsigned long long * ) d_dist );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

// ---- intra mode pre-selection (IntraSearch::estIntraPredLumaQT, EncoderLib/IntraSearch.cpp:555-592): SAD and SATD of N predictor blocks against ONE original block ----
}   // extern "C"
namespace
{
__global__ __launch_bounds__( 256 ) void intra_cand_jobs_kernel( vtmhip_dist_job *__restrict__ jobs, int n, long orgOff, int orgStride, long predOff, int w, int h )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= 2 * n ) return;
  const int cand = i < n ? i : i - n;
  vtmhip_dist_job d;
  d.orgOff = orgOff; d.curOff = predOff + ( long ) cand * w * h; d.orgStride = orgStride; d.curStride = w; d.width = ( int16_t ) w; d.height = ( int16_t ) h;
  d.subShift = 0; d.kind = ( int16_t ) ( i < n ? VTMHIP_DIST_SAD : VTMHIP_DIST_SATD );
  jobs[i] = d;
}
}   // namespace
extern "C"
{

int vtmhip_intra_cand_cost_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, int64_t orgOff, int orgStride, const int16_t *d_predBase, int64_t predOff, int n, int width,
                                      int height, uint64_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0 && n <= 4096, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_predBase && d_dist, "null pointer" );
  VTMHIP_REQUIRE( ctx, width >= 4 && width <= 128 && height >= 4 && height <= 128 && ( width & 3 ) == 0 && ( height & 3 ) == 0, "block size" );
  void *arena = nullptr;
  int   st    = vtmhip_internal_workspace( ctx, 2 * ( size_t ) n * sizeof( vtmhip_dist_job ), &arena );
  if( st ) return st;
  vtmhip_dist_job *jobs = ( vtmhip_dist_job * ) arena;
  hipLaunchKernelGGL( intra_cand_jobs_kernel, dim3( ( 2 * n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, jobs, n, ( long ) orgOff, orgStride, ( long ) predOff, width, height );
  VTMHIP_LAUNCHED( ctx );
  return vtmhip_dist_batch_dev( ctx, d_orgBase, d_predBase, jobs, 2 * n, d_dist );
}

int vtmhip_satd8_grid_dev( vtmhip_ctx *ctx, const int16_t *d_org, int orgStride, const int16_t *d_ref, int refStride, int width, int height, int r,
                           uint32_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, d_org && d_ref && d_dist, "null pointer" );
  VTMHIP_REQUIRE( ctx, width >= 8 && height >= 8 && r >= 0 && r <= 16, "size / range" );
  VTMHIP_REQUIRE( ctx, ( orgStride & 7 ) == 0 && ( ( ( uintptr_t ) d_org ) & 15 ) == 0, "org plane must be 16-byte aligned with a stride multiple of 8" );
  const int bw = width / 8, bh = height / 8;
  // workgroup shape: 8 x TBY blocks, THREADS lanes over the TBY * 8 * (2r+1)^2 pairs.  Tuned on 3840x2160, r = 4 (VTMHIP_SATD_VARIANT overrides it)
  static const int variant = getenv( "VTMHIP_SATD_VARIANT" ) ? atoi( getenv( "VTMHIP_SATD_VARIANT" ) ) : 0;
  auto launch = [&]( auto kern, int threads, int tby ) {
    const int    refW = GRID_TBX * 8 + 2 * r, refH = tby * 8 + 2 * r, refLd = ( refW + 7 ) & ~7;
    const size_t lds = ( size_t ) ( tby * 8 * GRID_TBX * 8 + refH * refLd ) * sizeof( int16_t ) + 16;   // + one spare vector: the packed path reads a fifth dword per row
    dim3         grid( ( bw + GRID_TBX - 1 ) / GRID_TBX, ( bh + tby - 1 ) / tby );
    VTMHIP_TIME_KERNEL( ctx, "satd8_grid_kernel" );
    hipLaunchKernelGGL( kern, grid, dim3( threads ), lds, ctx->stream, d_org, orgStride, d_ref, refStride, bw, bh, r, d_dist );
  };
  switch( variant )
  {
  case 1: launch( satd8_grid_kernel<192, 4>, 192, 4 ); break;
  case 2: launch( satd8_grid_kernel<128, 4>, 128, 4 ); break;
  case 3: launch( satd8_grid_kernel<256, 2>, 256, 2 ); break;
  case 4: launch( satd8_grid_kernel<128, 2>, 128, 2 ); break;
  case 5: launch( satd8_grid_kernel<64, 2>, 64, 2 ); break;
  case 6: launch( satd8_grid_kernel<192, 2>, 192, 2 ); break;
  case 7: launch( satd8_grid_kernel<64, 1>, 64, 1 ); break;
  default: launch( satd8_grid_kernel<256, 4>, 256, 4 ); break;
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"


extern "C" int vtmhip_masked_sad_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const int16_t *d_maskBase,
                                            const vtmhip_masked_sad_job *d_jobs, int n, uint64_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_curBase && d_maskBase && d_jobs && d_dist, "null pointer" );
  hipLaunchKernelGGL( sad_mask_kernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_orgBase, d_curBase, d_maskBase, d_jobs, n, ( unsigned long long * ) d_dist );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

extern "C" int vtmhip_xGetSADwMask( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, int subShift,
                                    const int16_t *mask, int maskStride, int stepX, int maskStride2, uint64_t *dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, org && cur && mask && dist, "null pointer" );
  VTMHIP_REQUIRE( ctx, width >= 1 && height >= 1 && width <= 128 && height <= 128, "block size must be 1..128" );
  VTMHIP_REQUIRE( ctx, subShift >= 0 && subShift <= 4 && ( height & ( ( 1 << subShift ) - 1 ) ) == 0, "subShift" );
  VTMHIP_REQUIRE( ctx, stepX == 1 || stepX == -1, "stepX must be +1 or -1" );
  // span of the mask the walk touches: offsets i * rowStep + x * stepX over the corners
  const int  rows = height >> subShift;
  const long rowStep = ( long ) width * stepX + ( long ) maskStride * ( 1 << subShift ) + maskStride2;
  long lo = 0, hi = 0;
  const long corners[4] = { 0, ( long ) ( width - 1 ) * stepX, ( long ) ( rows - 1 ) * rowStep, ( long ) ( rows - 1 ) * rowStep + ( long ) ( width - 1 ) * stepX };
  for( long c : corners ) { lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
  const size_t maskN = ( size_t ) ( hi - lo + 1 );
  VTMHIP_REQUIRE( ctx, maskN <= ( size_t ) 1 << 22, "mask walk spans more than 4 M samples" );
  const size_t blk = ( size_t ) width * height * sizeof( int16_t );
  const size_t maskOff = ( 2 * blk + 63 ) & ~( size_t ) 63, jobOff = ( maskOff + maskN * 2 + 63 ) & ~( size_t ) 63, outOff = jobOff + 64;
  int st = vtmhip_internal_scratch( ctx, outOff + 64 );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  for( int y = 0; y < height; y++ )
  {
    memcpy( hp + ( size_t ) y * width * 2, org + ( ptrdiff_t ) y * orgStride, ( size_t ) width * 2 );
    memcpy( hp + blk + ( size_t ) y * width * 2, cur + ( ptrdiff_t ) y * curStride, ( size_t ) width * 2 );
  }
  memcpy( hp + maskOff, mask + lo, maskN * 2 );
  vtmhip_masked_sad_job j;
  memset( &j, 0, sizeof( j ) );
  j.orgOff = 0; j.curOff = ( int64_t ) width * height; j.maskOff = ( int64_t ) ( maskOff / 2 ) - lo;
  j.orgStride = width; j.curStride = width; j.maskStride = maskStride; j.maskStride2 = maskStride2;
  j.width = ( int16_t ) width; j.height = ( int16_t ) height; j.subShift = ( int16_t ) subShift; j.stepX = ( int16_t ) stepX;
  memcpy( hp + jobOff, &j, sizeof( j ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, outOff, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( sad_mask_kernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ( const int16_t * ) dp, ( const int16_t * ) dp, ( const int16_t * ) dp,
                      ( const vtmhip_masked_sad_job * ) ( dp + jobOff ), 1, ( unsigned long long * ) ( dp + outOff ) );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + outOff, dp + outOff, 8, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  memcpy( dist, hp + outOff, 8 );
  return VTMHIP_OK;
}


extern "C" int vtmhip_dist_uniform_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const vtmhip_dist_job *d_jobs, int n,
                                              int kind, int width, int height, int subShift, uint64_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_curBase && d_jobs && d_dist, "null pointer" );
  int st = check_dist_args( ctx, width, height, subShift, kind );
  if( st ) return st;
  if( kind != VTMHIP_DIST_SATD && ( width & 7 ) != 0 )   // narrow / odd widths: the wave-per-job kernel (reads width / height / kind from the jobs)
    return vtmhip_dist_batch_dev( ctx, d_orgBase, d_curBase, d_jobs, n, d_dist );
  // work items of one job -> lanes per job
  int items;
  if( kind == VTMHIP_DIST_SATD )
  {
    const int w = width, h = height;
    int tw, th;
    if( w > h && ( h & 7 ) == 0 && ( w & 15 ) == 0 ) { tw = 16; th = 8; }
    else if( w < h && ( w & 7 ) == 0 && ( h & 15 ) == 0 ) { tw = 8; th = 16; }
    else if( w > h && ( h & 3 ) == 0 && ( w & 7 ) == 0 ) { tw = 8; th = 4; }
    else if( w < h && ( w & 3 ) == 0 && ( h & 7 ) == 0 ) { tw = 4; th = 8; }
    else if( ( h & 7 ) == 0 && ( w & 7 ) == 0 ) { tw = 8; th = 8; }
    else if( ( h & 3 ) == 0 && ( w & 3 ) == 0 ) { tw = 4; th = 4; }
    else { tw = 2; th = 2; }
    items = ( w / tw ) * ( h / th );
  }
  else items = ( width >> 3 ) * ( kind == VTMHIP_DIST_SAD ? height >> subShift : height );
  // lanes per job: SATD one tile per lane; SAD / SSE up to four 8-sample segments per lane (fewer waves, shorter reductions)
  const int perLane = kind == VTMHIP_DIST_SATD ? 1 : 4;
  int       lpjShift = 0;
  while( ( perLane << lpjShift ) < items && lpjShift < 6 ) lpjShift++;
  const int jobsPerWave = 64 >> lpjShift, waves = ( n + jobsPerWave - 1 ) / jobsPerWave;
  VTMHIP_TIME_KERNEL( ctx, "dist_uniform_kernel" );
  hipLaunchKernelGGL( dist_uniform_kernel, dim3( ( waves + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_orgBase, d_curBase, d_jobs, n, kind, width, height, subShift, lpjShift,
                      ( unsigned long long * ) d_dist );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}
