// frame.hip -- device-side chaining of the search stages of one picture: each helper derives the NEXT stage's job table from the
// previous stage's results with one thread per PU, so a level-order driver (vtm_amd/pipeline.py, bench.py) never brings a decision
// back to the host between stages.  The decisions mirror the orchestration of InterSearch::predInterSearch for the FEN operating point
// (EncoderLib/InterSearch.cpp:2531-2680): refine the list with the LARGER uni-prediction cost in the bi-predictive iteration, take
// the bi-prediction when its (halved, :3483) cost is below both uni costs.
#include "ctx.hpp"

namespace
{

__global__ __launch_bounds__( 256 ) void child_start_kernel( vtmhip_tz_job *__restrict__ child, int n, const int *__restrict__ parentIdx,
                                                            const vtmhip_me_result *__restrict__ parentRes )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  const int p  = parentIdx[i];
  const int mx = p >= 0 ? parentRes[p].mvX : 0, my = p >= 0 ? parentRes[p].mvY : 0;
  child[i].mvHor = mx << 4; child[i].mvVer = my << 4;        // start vector, internal 1/16 precision
  child[i].predHor = mx << 2; child[i].predVer = my << 2;    // MV predictor, quarter-sample units
}

__global__ __launch_bounds__( 256 ) void frac_jobs_kernel( vtmhip_frac_job *__restrict__ fj, const vtmhip_tz_job *__restrict__ tz,
                                                          const vtmhip_me_result *__restrict__ res, int n )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  fj[i].intX = ( int16_t ) res[i].mvX; fj[i].intY = ( int16_t ) res[i].mvY;
  fj[i].predHor = tz[i].predHor; fj[i].predVer = tz[i].predVer;
}

__device__ __forceinline__ int quarter( int integer, int half, int qter ) { return ( integer << 2 ) + ( half << 1 ) + qter; }

__global__ __launch_bounds__( 256 ) void bi_jobs_kernel( vtmhip_frame_tabs t )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= t.numPU ) return;
  const int r0 = t.row0[i], r1 = t.row1[i];
  const vtmhip_frac_result f0 = t.fracRes[r0], f1 = t.fracRes[r1];
  const int q0x = quarter( t.tzRes[r0].mvX, f0.halfX, f0.qterX ), q0y = quarter( t.tzRes[r0].mvY, f0.halfY, f0.qterY );
  const int q1x = quarter( t.tzRes[r1].mvX, f1.halfX, f1.qterX ), q1y = quarter( t.tzRes[r1].mvY, f1.halfY, f1.qterY );
  t.mvq[2 * r0] = q0x; t.mvq[2 * r0 + 1] = q0y; t.mvq[2 * r1] = q1x; t.mvq[2 * r1 + 1] = q1y;
  const bool refine1 = f0.cost <= f1.cost;                    // refine the list with the larger uni cost (:2544-2556)
  t.refineList[i]    = refine1;
  const int rr = refine1 ? r1 : r0;
  vtmhip_pred_job &po = t.predOther[i];
  po.mode = refine1 ? 0 : 1;                                  // prediction of the OTHER list
  po.mv[0][0] = q0x << 2; po.mv[0][1] = q0y << 2; po.mv[1][0] = q1x << 2; po.mv[1][1] = q1y << 2;
  const int64_t off = t.refBase[refine1 ? 1 : 0] + t.pos[i];
  vtmhip_full_job &fu = t.full[i];
  fu.refOff = off; fu.predHor = t.tz[rr].predHor; fu.predVer = t.tz[rr].predVer;
  fu.centerHor = ( refine1 ? q1x : q0x ) << 2; fu.centerVer = ( refine1 ? q1y : q0y ) << 2;
  vtmhip_frac_job &fb = t.fracBi[i];
  fb.refOff = off; fb.predHor = fu.predHor; fb.predVer = fu.predVer;
}

__global__ __launch_bounds__( 256 ) void bi_frac_jobs_kernel( vtmhip_frame_tabs t )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= t.numPU ) return;
  t.fracBi[i].intX = ( int16_t ) t.fullRes[i].mvX; t.fracBi[i].intY = ( int16_t ) t.fullRes[i].mvY;
}

__global__ __launch_bounds__( 256 ) void final_jobs_kernel( vtmhip_frame_tabs t )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= t.numPU ) return;
  const int r0 = t.row0[i], r1 = t.row1[i];
  const uint64_t c0 = t.fracRes[r0].cost, c1 = t.fracRes[r1].cost;
  const vtmhip_frac_result fb = t.fracBiRes[i];
  const int bx = quarter( t.fullRes[i].mvX, fb.halfX, fb.qterX ), by = quarter( t.fullRes[i].mvY, fb.halfY, fb.qterY );
  const int64_t costBi = ( int64_t ) ( fb.cost >> 1 );       // fWeight 0.5; the rate terms of :3483 are the host's mode decision
  const bool refine1 = t.refineList[i] != 0;
  const bool useBi   = ( uint64_t ) costBi < ( c0 < c1 ? c0 : c1 );
  t.biMv[2 * i] = bx; t.biMv[2 * i + 1] = by; t.costBi[i] = costBi; t.useBi[i] = useBi;
  vtmhip_pred_job &pf = t.predFinal[i];
  pf.mode = useBi ? 2 : ( c1 < c0 ? 1 : 0 );
  const bool b0 = useBi && !refine1, b1 = useBi && refine1;  // in a bi-predicted PU the refined list takes the bi vector
  pf.mv[0][0] = ( b0 ? bx : t.mvq[2 * r0] ) << 2; pf.mv[0][1] = ( b0 ? by : t.mvq[2 * r0 + 1] ) << 2;
  pf.mv[1][0] = ( b1 ? bx : t.mvq[2 * r1] ) << 2; pf.mv[1][1] = ( b1 ? by : t.mvq[2 * r1 + 1] ) << 2;
}

}   // namespace

extern "C"
{

int vtmhip_frame_child_start( vtmhip_ctx *ctx, vtmhip_tz_job *d_childJobs, int n, const int32_t *d_parentIdx, const vtmhip_me_result *d_parentRes )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_childJobs && d_parentIdx && d_parentRes, "null pointer" );
  hipLaunchKernelGGL( child_start_kernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_childJobs, n, d_parentIdx, d_parentRes );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_frame_frac_jobs( vtmhip_ctx *ctx, vtmhip_frac_job *d_fracJobs, const vtmhip_tz_job *d_tzJobs, const vtmhip_me_result *d_tzRes, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_fracJobs && d_tzJobs && d_tzRes, "null pointer" );
  hipLaunchKernelGGL( frac_jobs_kernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_fracJobs, d_tzJobs, d_tzRes, n );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_frame_stage( vtmhip_ctx *ctx, const vtmhip_frame_tabs *tabs, int stage )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, tabs && tabs->numPU >= 0 && stage >= 0 && stage <= 2, "tabs / stage" );
  if( tabs->numPU == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, tabs->tz && tabs->tzRes && tabs->fracRes && tabs->row0 && tabs->row1 && tabs->pos && tabs->predOther && tabs->full && tabs->fracBi
                         && tabs->fullRes && tabs->fracBiRes && tabs->predFinal && tabs->mvq && tabs->refineList && tabs->biMv && tabs->costBi && tabs->useBi,
                  "null pointer in the table set" );
  const dim3 grid( ( tabs->numPU + 255 ) / 256 ), blk( 256 );
  if( stage == 0 ) hipLaunchKernelGGL( bi_jobs_kernel, grid, blk, 0, ctx->stream, *tabs );
  else if( stage == 1 ) hipLaunchKernelGGL( bi_frac_jobs_kernel, grid, blk, 0, ctx->stream, *tabs );
  else hipLaunchKernelGGL( final_jobs_kernel, grid, blk, 0, ctx->stream, *tabs );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"
