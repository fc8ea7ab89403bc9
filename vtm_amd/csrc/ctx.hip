// ctx.hip -- context management, memory helpers and stream timing of the C ABI (include/vtmhip.h).
#include "ctx.hpp"

extern "C"
{

int vtmhip_abi_version( void ) { return VTMHIP_ABI_VERSION; }

int vtmhip_struct_size( int which )
{
  switch( which )
  {
  case 0: return ( int ) sizeof( vtmhip_dist_job );
  case 1: return ( int ) sizeof( vtmhip_tz_job );
  case 2: return ( int ) sizeof( vtmhip_me_result );
  case 3: return ( int ) sizeof( vtmhip_pic_params );
  case 4: return ( int ) sizeof( vtmhip_if_job );
  case 5: return ( int ) sizeof( vtmhip_frac_job );
  case 6: return ( int ) sizeof( vtmhip_frac_result );
  case 7: return ( int ) sizeof( vtmhip_tr_job );
  case 8: return ( int ) sizeof( vtmhip_quant_job );
  case 9: return ( int ) sizeof( vtmhip_full_job );
  case 10: return ( int ) sizeof( vtmhip_mc_job );
  case 11: return ( int ) sizeof( vtmhip_pelop_job );
  case 12: return ( int ) sizeof( vtmhip_tu_job );
  case 13: return ( int ) sizeof( vtmhip_tu_result );
  case 14: return ( int ) sizeof( vtmhip_affine_job );
  case 15: return ( int ) sizeof( vtmhip_me_cfg );
  case 16: return ( int ) sizeof( vtmhip_me_job );
  case 17: return ( int ) sizeof( vtmhip_me_out );
  case 18: return ( int ) sizeof( vtmhip_pred_job );
  case 19: return ( int ) sizeof( vtmhip_masked_sad_job );
  case 20: return ( int ) sizeof( vtmhip_geo_blend_job );
  case 21: return ( int ) sizeof( vtmhip_dmvr_job );
  case 22: return ( int ) sizeof( vtmhip_lfnst_job );
  case 23: return ( int ) sizeof( vtmhip_pis_row );
  case 24: return ( int ) sizeof( vtmhip_pis_pu );
  case 25: return ( int ) sizeof( vtmhip_pis_level );
  case 26: return ( int ) sizeof( vtmhip_affine_me_job );
  case 27: return ( int ) sizeof( vtmhip_affine_me_out );
  case 28: return ( int ) sizeof( vtmhip_lfnst_tu_job );
  case 29: return ( int ) sizeof( vtmhip_pis_level_run );
  case 30: return ( int ) sizeof( vtmhip_pis_buffers );
  case 31: return ( int ) sizeof( vtmhip_smvd_job );
  case 32: return ( int ) sizeof( vtmhip_pis_pu_in );
  default: return -1;
  }
}

const char *vtmhip_status_string( int status )
{
  switch( status )
  {
  case VTMHIP_OK: return "ok";
  case VTMHIP_E_INVALID: return "invalid argument";
  case VTMHIP_E_NODEVICE: return "no such HIP device";
  case VTMHIP_E_HIP: return "HIP runtime error";
  case VTMHIP_E_NOMEM: return "out of memory";
  case VTMHIP_E_UNSUPPORTED: return "unsupported on the device path (keep the CPU function)";
  default: return "unknown status";
  }
}

int vtmhip_device_count( int *count )
{
  if( !count ) return VTMHIP_E_INVALID;
  int n = 0;
  if( hipGetDeviceCount( &n ) != hipSuccess ) n = 0;
  *count = n;
  return VTMHIP_OK;
}

int vtmhip_create( int device, vtmhip_ctx **out )
{
  if( !out ) return VTMHIP_E_INVALID;
  *out  = nullptr;
  int n = 0;
  if( hipGetDeviceCount( &n ) != hipSuccess || n <= 0 || device < 0 || device >= n ) return VTMHIP_E_NODEVICE;
  vtmhip_ctx *ctx = new( std::nothrow ) vtmhip_ctx();
  if( !ctx ) return VTMHIP_E_NOMEM;
  ctx->device = device;
  if( hipSetDevice( device ) != hipSuccess || hipStreamCreateWithFlags( &ctx->ownStream, hipStreamNonBlocking ) != hipSuccess ||
      hipEventCreate( &ctx->evStart ) != hipSuccess || hipEventCreate( &ctx->evStop ) != hipSuccess )
  {
    delete ctx;
    return VTMHIP_E_HIP;
  }
  hipDeviceProp_t prop;
  if( hipGetDeviceProperties( &prop, device ) == hipSuccess ) ctx->numCUs = prop.multiProcessorCount;
  ctx->stream = ctx->ownStream;
  *out        = ctx;
  return VTMHIP_OK;
}

int vtmhip_destroy( vtmhip_ctx *ctx )
{
  if( !ctx ) return VTMHIP_OK;
  ( void ) hipSetDevice( ctx->device );
  ( void ) hipStreamSynchronize( ctx->stream );
  if( ctx->scratch ) ( void ) hipFree( ctx->scratch );
  for( auto &kv : ctx->work ) if( kv.second.ptr ) ( void ) hipFree( kv.second.ptr );
  if( ctx->lfnstTab ) ( void ) hipFree( ctx->lfnstTab );
  if( ctx->trTabBuf ) ( void ) hipFree( ctx->trTabBuf );
  if( ctx->pinned ) ( void ) hipHostFree( ctx->pinned );
  for( auto &t : ctx->timed ) { ( void ) hipEventDestroy( t.start ); ( void ) hipEventDestroy( t.stop ); }
  for( hipEvent_t e : ctx->forkEvents ) ( void ) hipEventDestroy( e );
  if( ctx->evStart ) ( void ) hipEventDestroy( ctx->evStart );
  if( ctx->evStop ) ( void ) hipEventDestroy( ctx->evStop );
  if( ctx->ownStream ) ( void ) hipStreamDestroy( ctx->ownStream );
  delete ctx;
  return VTMHIP_OK;
}

int vtmhip_set_stream( vtmhip_ctx *ctx, void *hipStream )
{
  VTMHIP_CHECK_CTX( ctx );
  ctx->stream = ( hipStream_t ) hipStream;   // NULL is HIP's default (null) stream, e.g. torch's default stream
  return VTMHIP_OK;
}

int vtmhip_use_own_stream( vtmhip_ctx *ctx )
{
  VTMHIP_CHECK_CTX( ctx );
  ctx->stream = ctx->ownStream;
  return VTMHIP_OK;
}

int vtmhip_sync( vtmhip_ctx *ctx )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  return VTMHIP_OK;
}

const char *vtmhip_last_error( vtmhip_ctx *ctx ) { return ctx ? ctx->lastError.c_str() : "null context"; }

int vtmhip_dev_alloc( vtmhip_ctx *ctx, size_t bytes, void **devPtr )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, devPtr != nullptr, "devPtr" );
  VTMHIP_HIP( ctx, hipSetDevice( ctx->device ) );
  VTMHIP_HIP( ctx, hipMalloc( devPtr, bytes ? bytes : 1 ) );
  return VTMHIP_OK;
}

int vtmhip_dev_free( vtmhip_ctx *ctx, void *devPtr )
{
  VTMHIP_CHECK_CTX( ctx );
  if( devPtr ) VTMHIP_HIP( ctx, hipFree( devPtr ) );
  return VTMHIP_OK;
}

int vtmhip_host_alloc( vtmhip_ctx *ctx, size_t bytes, void **hostPtr )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, hostPtr != nullptr, "hostPtr" );
  VTMHIP_HIP( ctx, hipHostMalloc( hostPtr, bytes ? bytes : 1 ) );
  return VTMHIP_OK;
}

int vtmhip_host_free( vtmhip_ctx *ctx, void *hostPtr )
{
  VTMHIP_CHECK_CTX( ctx );
  if( hostPtr ) VTMHIP_HIP( ctx, hipHostFree( hostPtr ) );
  return VTMHIP_OK;
}

int vtmhip_h2d( vtmhip_ctx *ctx, void *dev, const void *host, size_t bytes )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dev, host, bytes, hipMemcpyHostToDevice, ctx->stream ) );
  return VTMHIP_OK;
}

int vtmhip_d2h( vtmhip_ctx *ctx, void *host, const void *dev, size_t bytes )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  return VTMHIP_OK;
}

int vtmhip_timer_start( vtmhip_ctx *ctx )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_HIP( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );
  return VTMHIP_OK;
}

int vtmhip_timer_stop_ms( vtmhip_ctx *ctx, float *ms )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, ms != nullptr, "ms" );
  VTMHIP_HIP( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );
  VTMHIP_HIP( ctx, hipEventSynchronize( ctx->evStop ) );
  VTMHIP_HIP( ctx, hipEventElapsedTime( ms, ctx->evStart, ctx->evStop ) );
  return VTMHIP_OK;
}

static void clear_timed( vtmhip_ctx *ctx )
{
  for( auto &t : ctx->timed ) { ( void ) hipEventDestroy( t.start ); ( void ) hipEventDestroy( t.stop ); }
  ctx->timed.clear();   // the fork / join events of the picture loop (ctx->forkEvents) are reused by every picture and live until vtmhip_destroy
}

int vtmhip_kernel_timing( vtmhip_ctx *ctx, int enable )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_HIP( ctx, hipDeviceSynchronize() );
  clear_timed( ctx );
  ctx->timing = enable != 0;
  return VTMHIP_OK;
}

int vtmhip_kernel_timing_read( vtmhip_ctx *ctx, const char *kernel, double *totalMs, int *launches )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, kernel && totalMs && launches, "null pointer" );
  VTMHIP_HIP( ctx, hipDeviceSynchronize() );
  double sum = 0;
  int    n   = 0;
  for( auto &t : ctx->timed )
    if( std::strcmp( t.kernel, kernel ) == 0 )
    {
      float ms = 0;
      VTMHIP_HIP( ctx, hipEventElapsedTime( &ms, t.start, t.stop ) );
      sum += ms;
      n++;
    }
  *totalMs = sum; *launches = n;
  return VTMHIP_OK;
}

}   // extern "C"

int vtmhip_internal_scratch( vtmhip_ctx *ctx, size_t bytes )
{
  if( bytes <= ctx->scratchSize ) return VTMHIP_OK;
  size_t want = ( bytes + ( 1u << 20 ) - 1 ) & ~( size_t )( ( 1u << 20 ) - 1 );
  VTMHIP_HIP( ctx, hipSetDevice( ctx->device ) );
  VTMHIP_HIP( ctx, hipDeviceSynchronize() );   // vtmhip_set_stream may have moved the context between streams: launches queued on ANY of them may still use the old block
  if( ctx->scratch ) VTMHIP_HIP( ctx, hipFree( ctx->scratch ) );
  if( ctx->pinned ) VTMHIP_HIP( ctx, hipHostFree( ctx->pinned ) );
  ctx->scratch = nullptr; ctx->pinned = nullptr; ctx->scratchSize = 0; ctx->pinnedSize = 0;
  VTMHIP_HIP( ctx, hipMalloc( &ctx->scratch, want ) );
  VTMHIP_HIP( ctx, hipHostMalloc( &ctx->pinned, want ) );
  ctx->scratchSize = ctx->pinnedSize = want;
  return VTMHIP_OK;
}

int vtmhip_internal_workspace( vtmhip_ctx *ctx, size_t bytes, void **out, int slot )
{
  std::lock_guard<std::mutex> lock( ctx->initMutex );
  vtmhip_ctx::WorkArena &a = ctx->work[std::make_pair( ctx->stream, slot )];
  if( bytes > a.size )
  {
    const size_t want = ( bytes + ( 1u << 20 ) - 1 ) & ~( size_t )( ( 1u << 20 ) - 1 );
    VTMHIP_HIP( ctx, hipSetDevice( ctx->device ) );
    VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );   // earlier launches of THIS stream may still use the old block (no other stream ever sees it)
    if( a.ptr ) VTMHIP_HIP( ctx, hipFree( a.ptr ) );
    a.ptr = nullptr; a.size = 0;
    if( hipMalloc( &a.ptr, want ) != hipSuccess )
    {
      ( void ) hipGetLastError();
      a.ptr = nullptr;
      ctx->lastError = "out of device memory for the call's workspace";
      return VTMHIP_E_NOMEM;
    }
    a.size = want;
  }
  *out = a.ptr;
  return VTMHIP_OK;
}
