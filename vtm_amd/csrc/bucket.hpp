// bucket.hpp -- device-side bucketing of a mixed-shape job batch by block shape (the batched hooks of a real encoder hand over binary / ternary split
// blocks of every size in one call; the fast kernels want uniform batches).
//
//   count    one pass over the job table: histogram of the shape classes                                (bucket_count_kernel)
//   offsets  exclusive prefix sums on the device; the class counts go to the host (80 bytes: the only synchronisation)  (bucket_offsets_kernel)
//   scatter  job i -> slot offsets[class] + atomic cursor; perm[slot] = i                                (bucket_scatter_kernel)
//   ...      one uniform launch chain per non-empty class over its contiguous slice of the permuted table
//   gather   results[perm[slot]] = permuted results[slot]                                                 (bucket_gather_kernel)
// The order inside a class is whatever the atomics give; every job is independent and its result returns to its own index, so the output is
// deterministic.
#pragma once
#include <hip/hip_runtime.h>

#include "ctx.hpp"

namespace
{

constexpr int BUCKET_CLASSES = 20;   // fast shapes (15 PU shapes / 16 TU shapes) + "everything else" (the last class)
constexpr int BUCKET_OTHER   = BUCKET_CLASSES - 1;

// squares 8 .. 128, then the split shapes in both orientations; BUCKET_OTHER = any other shape
__host__ __device__ inline int shape_class( int w, int h )
{
  if( w == h ) return w == 8 ? 0 : w == 16 ? 1 : w == 32 ? 2 : w == 64 ? 3 : w == 128 ? 4 : BUCKET_OTHER;
  const int a = w > h ? w : h, b = w > h ? h : w, o = w > h ? 0 : 1;
  if( a == 16 && b == 8 ) return 5 + o;
  if( a == 32 && b == 8 ) return 7 + o;
  if( a == 32 && b == 16 ) return 9 + o;
  if( a == 64 && b == 16 ) return 11 + o;
  if( a == 64 && b == 32 ) return 13 + o;
  return BUCKET_OTHER;
}

inline void class_shape( int c, int &w, int &h )
{
  static const int sq[5] = { 8, 16, 32, 64, 128 };
  static const int ra[5] = { 16, 32, 32, 64, 64 }, rb[5] = { 8, 8, 16, 16, 32 };
  if( c < 5 ) { w = h = sq[c]; return; }
  const int k = ( c - 5 ) >> 1, o = ( c - 5 ) & 1;
  w = o ? rb[k] : ra[k];
  h = o ? ra[k] : rb[k];
}

template<typename Job, typename ClassOf>
__global__ __launch_bounds__( 256 ) void bucket_count_kernel( const Job *__restrict__ jobs, int n, int *__restrict__ counts, ClassOf classOf )
{
  __shared__ int sCnt[BUCKET_CLASSES];
  if( threadIdx.x < BUCKET_CLASSES ) sCnt[threadIdx.x] = 0;
  __syncthreads();
  for( int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x ) atomicAdd( &sCnt[classOf( jobs[i] )], 1 );
  __syncthreads();
  if( threadIdx.x < BUCKET_CLASSES && sCnt[threadIdx.x] ) atomicAdd( &counts[threadIdx.x], sCnt[threadIdx.x] );
}

template<typename Job, typename ClassOf>
__global__ __launch_bounds__( 256 ) void bucket_scatter_kernel( const Job *__restrict__ jobs, int n, const int *__restrict__ offsets, int *__restrict__ cursor,
                                                               Job *__restrict__ out, int *__restrict__ perm, ClassOf classOf )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if( i >= n ) return;
  const Job j    = jobs[i];
  const int c    = classOf( j );
  const int slot = offsets[c] + atomicAdd( &cursor[c], 1 );
  out[slot]  = j;
  perm[slot] = i;
}

__global__ void bucket_offsets_kernel( const int *__restrict__ counts, int *__restrict__ offsets )
{
  int acc = 0;
  for( int c = 0; c < BUCKET_CLASSES; c++ ) { offsets[c] = acc; acc += counts[c]; }
}

template<typename R>
__global__ __launch_bounds__( 256 ) void bucket_gather_kernel( const R *__restrict__ in, const int *__restrict__ perm, int n, R *__restrict__ out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if( i < n ) out[perm[i]] = in[i];
}

// The bucketed path reads the class counts back (one hipStreamSynchronize): it is NOT usable while the stream is being captured into a hipGraph -- callers
// check this and keep their non-bucketed chain then (same results, one generic launch chain for the whole batch).
inline bool bucket_allowed( vtmhip_ctx *ctx )
{
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing( ctx->stream, &s ) == hipSuccess && s == hipStreamCaptureStatusNone;
}

struct BucketPlan
{
  int count[BUCKET_CLASSES], offset[BUCKET_CLASSES];
  int *d_perm;
  void *d_jobs;      // permuted job table
  void *d_results;   // permuted results
};

// Buckets `n` jobs: on return plan.d_jobs holds the permuted table (class slices at plan.offset[], sizes plan.count[]); the caller runs its chains over
// the slices writing plan.d_results, then calls bucket_finish.  Workspace: arena slot 2 of the context's stream.
template<typename Job, typename R, typename ClassOf>
int bucket_begin( vtmhip_ctx *ctx, const Job *d_jobs, int n, ClassOf classOf, BucketPlan &plan )
{
  const size_t oJobs = 256, oRes = ( oJobs + ( size_t ) n * sizeof( Job ) + 255 ) & ~( size_t ) 255, oPerm = ( oRes + ( size_t ) n * sizeof( R ) + 255 ) & ~( size_t ) 255;
  void *arena = nullptr;
  int   st    = vtmhip_internal_workspace( ctx, oPerm + ( size_t ) n * sizeof( int ), &arena, 2 );
  if( st ) return st;
  char *base = ( char * ) arena;
  int  *d_cnt = ( int * ) base, *d_off = d_cnt + BUCKET_CLASSES, *d_cur = d_off + BUCKET_CLASSES;   // three small tables at the head of the arena
  plan.d_jobs = base + oJobs; plan.d_results = base + oRes; plan.d_perm = ( int * ) ( base + oPerm );
  VTMHIP_HIP( ctx, hipMemsetAsync( d_cnt, 0, 3 * BUCKET_CLASSES * sizeof( int ), ctx->stream ) );
  const int blocks = ( n + 255 ) / 256;
  hipLaunchKernelGGL( ( bucket_count_kernel<Job, ClassOf> ), dim3( blocks < 1024 ? blocks : 1024 ), dim3( 256 ), 0, ctx->stream, d_jobs, n, d_cnt, classOf );
  hipLaunchKernelGGL( bucket_offsets_kernel, dim3( 1 ), dim3( 1 ), 0, ctx->stream, d_cnt, d_off );
  VTMHIP_HIP( ctx, hipMemcpyAsync( plan.count, d_cnt, sizeof( plan.count ), hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );   // the class sizes decide the launches that follow
  int acc = 0;
  for( int c = 0; c < BUCKET_CLASSES; c++ ) { plan.offset[c] = acc; acc += plan.count[c]; }
  VTMHIP_REQUIRE( ctx, acc == n, "bucket counts" );
  hipLaunchKernelGGL( ( bucket_scatter_kernel<Job, ClassOf> ), dim3( blocks ), dim3( 256 ), 0, ctx->stream, d_jobs, n, d_off, d_cur, ( Job * ) plan.d_jobs, plan.d_perm, classOf );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

template<typename R>
int bucket_finish( vtmhip_ctx *ctx, const BucketPlan &plan, int n, R *d_results )
{
  hipLaunchKernelGGL( bucket_gather_kernel<R>, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, ( const R * ) plan.d_results, plan.d_perm, n, d_results );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // namespace
