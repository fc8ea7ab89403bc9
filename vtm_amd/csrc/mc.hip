// mc.hip -- luma motion compensation of whole blocks and the two PelBufferOps of bi-predictive ME.
//
// Reference: CommonLib/InterPrediction.cpp xPredInterBlk :660-815 (luma, no BDOF/DMVR/RPR/wrap-around):
//   yFrac == 0 -> filterHor(isLast = rndRes); xFrac == 0 -> filterVer(first, isLast = rndRes);
//   else filterHor(first, !last) on rows -3..H+3 then filterVer(!first, isLast = rndRes);   rndRes = !bi.
// CommonLib/Buffer.cpp removeHighFreq :475-520 (org = 2*org - pred, unclipped: ClipForBiPredMEEnabled = 0),
// addAvg :467-507 (dst = clip((a + b + offset) >> shift) on 14-bit intermediates).
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

__global__ __launch_bounds__( 64 ) void mc_luma_kernel( const int16_t *__restrict__ refBase, int16_t *__restrict__ dstBase,
                                                       const vtmhip_mc_job *__restrict__ jobs, int maxW, int maxH )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];   // [(h+7)][w] H-pass intermediates
  const vtmhip_mc_job j    = jobs[blockIdx.x];
  const int           lane = threadIdx.x, w = j.width, h = j.height, bd = j.bitDepth;
  const int           xFrac = j.mvHor & 15, yFrac = j.mvVer & 15, rnd = !j.bi;
  const bool          alt = j.useAltHpelIf != 0;
  const int16_t      *src = refBase + j.refOff + ( long ) ( j.mvVer >> 4 ) * j.refStride + ( j.mvHor >> 4 );
  int16_t            *dst = dstBase + j.dstOff;
  if( yFrac == 0 )
  {
    if( xFrac == 0 && rnd )   // filterCopy<true,true>: plain copy
    {
      for( int i = lane; i < w * h; i += 64 ) { const int y = i / w, x = i - y * w; dst[( long ) y * j.dstStride + x] = src[( long ) y * j.refStride + x]; }
      return;
    }
    const Fir      f = fir_params( 1, rnd, bd );
    const int16_t *c = luma_taps( xFrac, w, h, h, alt );
    for( int i = lane; i < w * h; i += 64 )
    {
      const int y = i / w, x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) src[( long ) y * j.refStride + x + k - 3] * ( int ) c[k];
      dst[( long ) y * j.dstStride + x] = fir_out( sum, f );
    }
  }
  else if( xFrac == 0 )
  {
    const Fir      f = fir_params( 1, rnd, bd );
    const int16_t *c = luma_taps( yFrac, w, h, h, alt );
    for( int i = lane; i < w * h; i += 64 )
    {
      const int y = i / w, x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) src[( long ) ( y + k - 3 ) * j.refStride + x] * ( int ) c[k];
      dst[( long ) y * j.dstStride + x] = fir_out( sum, f );
    }
  }
  else
  {
    const Fir      fh = fir_params( 1, 0, bd ), fv = fir_params( 0, rnd, bd );
    const int16_t *ch = luma_taps( xFrac, w, h, h + 7 == 11 ? 4 : -1, alt );   // the H pass sees a W x (H+7) block: 4 x 11 takes the 4x4 taps
    const int16_t *cv = luma_taps( yFrac, w, h, h, alt );
    for( int i = lane; i < w * ( h + 7 ); i += 64 )
    {
      const int r = i / w, x = i - r * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) src[( long ) ( r - 3 ) * j.refStride + x + k - 3] * ( int ) ch[k];
      lds[i] = fir_out( sum, fh );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
    for( int i = lane; i < w * h; i += 64 )
    {
      const int y = i / w, x = i - y * w;
      int       sum = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) sum += ( int ) lds[( y + k ) * w + x] * ( int ) cv[k];
      dst[( long ) y * j.dstStride + x] = fir_out( sum, fv );
    }
  }
}

__global__ __launch_bounds__( 256 ) void pelop_kernel( const int16_t *__restrict__ aBase, const int16_t *__restrict__ bBase, int16_t *__restrict__ dstBase,
                                                      const vtmhip_pelop_job *__restrict__ jobs, int op )
{
  const vtmhip_pelop_job j = jobs[blockIdx.x];
  const int16_t         *a = aBase + j.aOff, *b = bBase + j.bOff;
  int16_t               *d = dstBase + j.dstOff;
  const int              w = j.width, h = j.height;
  const int              headRoom = max( 2, 14 - ( int ) j.bitDepth ), shift = headRoom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * 8192;
  const int              cmax = ( 1 << j.bitDepth ) - 1;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int y = i / w, x = i - y * w;
    const int av = a[( long ) y * j.aStride + x], bv = b[( long ) y * j.bStride + x];
    int       v;
    if( op == 0 ) v = ( int16_t ) ( 2 * av - bv );                        // removeHighFreq
    else if( op == 2 ) v = ( int16_t ) ( av - bv );                        // subtract (residual = org - pred, Buffer.cpp AreaBuf::subtract)
    else v = min( cmax, max( 0, ( av + bv + offset ) >> shift ) );         // addAvg
    d[( long ) y * j.dstStride + x] = ( int16_t ) v;
  }
}

}   // namespace

extern "C"
{

int vtmhip_mc_luma_batch_dev( vtmhip_ctx *ctx, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_mc_job *d_jobs, int n, int maxWidth,
                              int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_refBase && d_dstBase && d_jobs, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxWidth <= 128 && maxHeight >= 4 && maxHeight <= 128, "maxWidth / maxHeight" );
  const size_t lds = ( size_t ) maxWidth * ( maxHeight + 7 ) * sizeof( int16_t );
  hipLaunchKernelGGL( mc_luma_kernel, dim3( n ), dim3( 64 ), lds, ctx->stream, d_refBase, d_dstBase, d_jobs, maxWidth, maxHeight );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_remove_high_freq_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_predBase, int16_t *d_dstBase,
                                       const vtmhip_pelop_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_predBase && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( pelop_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_orgBase, d_predBase, d_dstBase, d_jobs, 0 );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_subtract_batch_dev( vtmhip_ctx *ctx, const int16_t *d_aBase, const int16_t *d_bBase, int16_t *d_dstBase, const vtmhip_pelop_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_aBase && d_bBase && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( pelop_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_aBase, d_bBase, d_dstBase, d_jobs, 2 );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_add_avg_batch_dev( vtmhip_ctx *ctx, const int16_t *d_src0Base, const int16_t *d_src1Base, int16_t *d_dstBase,
                              const vtmhip_pelop_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_src0Base && d_src1Base && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( pelop_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_src0Base, d_src1Base, d_dstBase, d_jobs, 1 );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"
