// ctx.hpp -- context object and small host/device helpers shared by the libvtmhip.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <utility>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vtmhip.h"

struct vtmhip_ctx
{
  int         device      = 0;
  hipStream_t ownStream   = nullptr;
  hipStream_t stream      = nullptr;   // the stream every launch goes to (own or caller-supplied)
  hipEvent_t  evStart     = nullptr;
  hipEvent_t  evStop      = nullptr;
  void       *scratch     = nullptr;   // staging area for the pointer-surface (host pointer) calls
  size_t      scratchSize = 0;
  void       *pinned      = nullptr;   // pinned host mirror of the staging area
  size_t      pinnedSize  = 0;
  // device workspaces of the multi-stage calls (vtmhip_xMotionEstimation_batch_dev ...): ONE ARENA PER STREAM the context has been pointed at, so that
  // calls queued on different streams (a level-order driver runs the levels' chains concurrently) never share scratch memory; calls on one stream are
  // ordered and reuse their arena
  struct WorkArena { void *ptr = nullptr; size_t size = 0; };
  std::map<std::pair<hipStream_t, int>, WorkArena> work;   // key: (stream, slot) -- slot 0: the multi-stage calls, slot 1: calls they nest (the split TZ search)
  int8_t     *lfnstTab    = nullptr;   // the caller's LFNST core matrices: g_lfnst8x8 [4][2][16][48] then g_lfnst4x4 [4][2][16][16] (vtmhip_lfnst_set_tables)
  int16_t    *trTabBuf    = nullptr;   // the transform core matrices of THIS context's device (transform.hip ensure_tables; freed by vtmhip_destroy)
  const int16_t *trTab[3][7] = {};     // [type][log2 N] -> N x N forward matrix inside trTabBuf
  std::mutex  initMutex;               // guards the lazy per-context initialisations (tables, staging / workspace growth)
  int         numCUs      = 256;
  std::string lastError;
  // per-kernel launch timing (vtmhip_kernel_timing): HIP events recorded on the launch stream around every launch of the main kernels
  bool        timing      = false;
  struct TimedLaunch { const char *kernel; hipEvent_t start, stop; };
  std::vector<TimedLaunch> timed;
  std::vector<hipEvent_t>  forkEvents;   // fork / join events of vtmhip_pis_run_picture (driver.hip): created on demand, reused by every picture, freed by vtmhip_destroy
};

// scope guard around a kernel launch: records start / stop events on ctx->stream when timing is on (bench.py's roofline: the dominant kernel's
// average launch duration, measured live on the stream the kernel is launched on)
struct vtmhip_launch_timer
{
  vtmhip_ctx *c;
  hipEvent_t  stop = nullptr;
  vtmhip_launch_timer( vtmhip_ctx *ctx, const char *kernel ) : c( ctx )
  {
    if( !c->timing ) return;
    hipEvent_t start = nullptr;
    if( hipEventCreate( &start ) != hipSuccess || hipEventCreate( &stop ) != hipSuccess ) { stop = nullptr; return; }
    ( void ) hipEventRecord( start, c->stream );
    c->timed.push_back( { kernel, start, stop } );
  }
  ~vtmhip_launch_timer() { if( stop ) ( void ) hipEventRecord( stop, c->stream ); }
};
#define VTMHIP_TIME_KERNEL( ctx, name ) vtmhip_launch_timer vtmhip_timer_guard_( ctx, name )

// every entry point starts here: a null context is an error, and the calling thread is pointed at the context's device (a host with
// several contexts / GPUs in one process, or encoder threads that never called hipSetDevice, would otherwise allocate and launch on
// whatever device happens to be current)
#define VTMHIP_CHECK_CTX( ctx )                                                                    \
  do {                                                                                             \
    if( !( ctx ) ) return VTMHIP_E_INVALID;                                                        \
    if( hipSetDevice( ( ctx )->device ) != hipSuccess )                                            \
    {                                                                                              \
      ( ctx )->lastError = "hipSetDevice failed";                                                  \
      return VTMHIP_E_HIP;                                                                         \
    }                                                                                              \
  } while( 0 )

#define VTMHIP_HIP( ctx, call )                                                                        \
  do {                                                                                                 \
    hipError_t e_ = ( call );                                                                          \
    if( e_ != hipSuccess )                                                                             \
    {                                                                                                  \
      ( ctx )->lastError = std::string( #call ) + ": " + hipGetErrorString( e_ );                      \
      return VTMHIP_E_HIP;                                                                             \
    }                                                                                                  \
  } while( 0 )

#define VTMHIP_REQUIRE( ctx, cond, msg )                   \
  do {                                                     \
    if( !( cond ) )                                        \
    {                                                      \
      ( ctx )->lastError = std::string( "invalid argument: " ) + ( msg ); \
      return VTMHIP_E_INVALID;                             \
    }                                                      \
  } while( 0 )

// launch-error check after a kernel launch (asynchronous errors surface at the next sync)
#define VTMHIP_LAUNCHED( ctx ) VTMHIP_HIP( ctx, hipGetLastError() )

int vtmhip_internal_scratch( vtmhip_ctx *ctx, size_t bytes );   // grows ctx->scratch / ctx->pinned
int vtmhip_internal_mc_launch( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase, const vtmhip_pred_job *d_jobs,
                               int n, int maxWidth, int maxHeight, unsigned long long *d_sadOut );   // mc.hip: motionCompensation, optionally reduced to the SAD against the original
int vtmhip_internal_mc_amvp_launch( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_me_job *d_rows, int n,
                                    int maxWidth, int maxHeight, unsigned long long *d_sadOut );   // mc.hip: the AMVP candidates' predictions reduced to their SADs, jobs from the ME rows
#define VTMHIP_AFFINE_LAUNCH_WIDE 8      // `models` of vtmhip_internal_affine_me_launch: also launch the 32-bit variants (jobs with a BCW weight of -2)
int vtmhip_internal_affine_me_launch( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const int16_t *d_otherPredBase,
                                      const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_affine_me_out *d_results, int models );   // affine.hip
struct MeFuse;      // mest_glue.hpp
int vtmhip_internal_tz_search( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_tz_job *d_jobs, int n,
                               vtmhip_me_result *d_results, const MeFuse *fuse, int uniformW, int uniformH );   // me.hip: vtmhip_tz_search_batch_dev (uniformW / H != 0: every job has this shape), optionally fused with the row bookkeeping around it
struct FullFuse;    // mest_glue.hpp
int vtmhip_internal_full_search( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_full_job *d_jobs, int n,
                                 int width, int height, vtmhip_me_result *d_results, const FullFuse *fuse );   // me.hip: the exhaustive searches, optionally fused
int vtmhip_internal_frac_search( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_frac_job *d_jobs, int n, int maxWidth, int maxHeight,
                                 int uniformSquare, vtmhip_frac_result *d_results, const MeFuse *fuse );   // interp.hip: vtmhip_frac_search_batch_dev, optionally fused
int vtmhip_internal_amvp_sads( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_me_job *d_jobs, int n,
                               int maxWidth, int maxHeight, unsigned long long **d_dout );   // pis.hip
int vtmhip_internal_mest_with_amvp( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase, const int16_t *d_refBase,
                                    vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_me_out *d_results, const unsigned long long *d_amvpDout,
                                    unsigned long long *d_distBiP, int addIdxBits );   // mest.hip
int vtmhip_internal_mest_fusable( const vtmhip_me_cfg *cfg );
int vtmhip_internal_workspace( vtmhip_ctx *ctx, size_t bytes, void **out, int slot = 0 ); // the arena (slot) of ctx->stream, grown to `bytes` (device only)

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_reduce_add( int v )
{
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 ) v += __shfl_xor( v, o, 64 );
  return v;
}

__device__ __forceinline__ unsigned long long wave_reduce_add_u64( unsigned long long v )
{
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 ) v += __shfl_xor( v, o, 64 );
  return v;
}

// XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so block b takes
// the work item at position (b % 8) * (n / 8) + b / 8 (bijective for any n): the workgroups running on one XCD at any moment
// then hold NEIGHBOURING jobs of the table (PUs in raster order), whose search windows overlap and stay in that XCD's 4 MiB L2.
__device__ __forceinline__ int xcd_order( int b, int n )
{
  const int q = n >> 3, r = n & 7, xcd = b & 7;
  return ( xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q ) + ( b >> 3 );
}
