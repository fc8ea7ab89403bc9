// smvd.hip -- the symmetric MVD (SMVD) search of predInterSearch: InterSearch::xGetSymmetricCost (EncoderLib/InterSearch.cpp:4341-4391),
// xSymmeticRefineMvSearch / xSymmetricMotionEstimation (:4393-4518), symmvdCheckBestMvp (:7787-7886) and the block that composes them (:2656-2790).
//
// One workgroup per PU (one wave up to 32x32 samples, four above).  A candidate = two luma predictions (mc_block.hpp) + one distortion:
//   pattern  = clip?( 2 * org - predA )      written by the sink of prediction A straight into LDS (removeHighFreq fused, BCW form included)
//   predB                                   into LDS
//   cost     = floor( fWeight * HAD-or-SAD( pattern, predB ) )    (tile shapes of RdCost::xGetHADs)
// The search control (rounds, directions, best-so-far) is workgroup-uniform: every lane derives it from the same reduced cost, so no state is shared
// but the three LDS blocks.  HBM traffic per candidate is the two (w + 7) x (h + 7) windows, which stay in L2 across the ~40 candidates of a PU.
#include "ctx.hpp"
#include "had.hpp"
#include "mc_block.hpp"

namespace
{

__constant__ int8_t c_diamond[8][2] = { { 0, 2 }, { 1, 1 }, { 2, 0 }, { 1, -1 }, { 0, -2 }, { -1, -1 }, { -2, 0 }, { -1, 1 } };
__constant__ int8_t c_cross[4][2]   = { { 0, 1 }, { 1, 0 }, { 0, -1 }, { -1, 0 } };

__device__ __forceinline__ int      prec_dn( int v, int rs ) { const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }   // Mv.h:183-197
__device__ __forceinline__ unsigned eg_bits( int v )   // RdCost::xGetExpGolombNumberOfBits (RdCost.h:301-315)
{
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}

struct SmvdCtx   // workgroup-uniform view of one job
{
  const int16_t *org, *refBase;
  long           refOff[2];
  int            orgStride, refStride[2];
  int            w, h, bd, imv, amvrShift, bcw;
  int            horMin, horMax, verMin, verMax;
  bool           satd, clip;
  double         lam, fWeight;
  unsigned       idxBits[2];
  int16_t       *sPat, *sB, *sTmp;
  unsigned long long *sRed;
};

// sink of prediction A: bufTmp = org; bufTmp.removeHighFreq( predA, bClip, clpRng, bcwWeight ) (Buffer.h:417-520, 946-957)
struct PatternOut
{
  int16_t *p; int w; const int16_t *org; int os, cmax, w0, w1; bool clip, bcw;
  __device__ __forceinline__ int f( int o, int v ) const
  {
    int r = bcw ? ( o * w0 - v * w1 + ( 1 << 15 ) ) >> 16 : 2 * o - v;
    if( clip ) r = min( cmax, max( 0, r ) );
    return ( int ) ( int16_t ) r;
  }
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const { p[y * w + x] = ( int16_t ) f( org[( long ) y * os + x], v ); }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const
  {
    int o[8], r[8];
    load8g( org + ( long ) y * os + x0, o );
#pragma unroll
    for( int k = 0; k < 8; k++ ) r[k] = f( o[k], v[k] );
    *reinterpret_cast<uint4 *>( p + y * w + x0 ) = pack8( r );
  }
};
struct LdsOut
{
  int16_t *p; int w;
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const { p[y * w + x] = v; }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const { *reinterpret_cast<uint4 *>( p + y * w + x0 ) = pack8( v ); }
};

template<int THREADS, class Out>
__device__ __forceinline__ void predict( const SmvdCtx &c, int l, int mvHor, int mvVer, Out out )
{
  vtmhip_mc_job m;
  m.refOff = c.refOff[l]; m.dstOff = 0; m.refStride = c.refStride[l]; m.dstStride = c.w; m.width = ( int16_t ) c.w; m.height = ( int16_t ) c.h;
  m.mvHor = min( c.horMax, max( c.horMin, mvHor ) ); m.mvVer = min( c.verMax, max( c.verMin, mvVer ) );   // clipMv (clipMvInPic, Mv.cpp:56-74)
  m.bi = 0; m.bitDepth = ( uint8_t ) c.bd; m.useAltHpelIf = c.imv == 3; m.chroma = 0;
  mc_any<THREADS>( m, c.refBase, c.sTmp, ( int ) threadIdx.x, out );
  block_sync<THREADS>();
}

template<int THREADS>
__device__ __forceinline__ void make_pattern( const SmvdCtx &c, int mvHor, int mvVer )
{
  PatternOut po;
  po.p = c.sPat; po.w = c.w; po.org = c.org; po.os = c.orgStride; po.cmax = ( 1 << c.bd ) - 1; po.clip = c.clip; po.bcw = c.bcw != 4;
  const int normalizer = po.bcw ? ( ( 1 << 16 ) + ( c.bcw > 0 ? ( c.bcw >> 1 ) : -( c.bcw >> 1 ) ) ) / c.bcw : 0;
  po.w0 = normalizer * 8; po.w1 = ( 8 - c.bcw ) * normalizer;
  predict<THREADS>( c, 0, mvHor, mvVer, po );
}

// getDistPart( DF_HAD / DF_SAD ) of pattern vs prediction B (tile shapes of xGetHADs, RdCost.cpp:2837-2931), workgroup-wide sum, same value in every lane
template<int THREADS>
__device__ __forceinline__ unsigned long long block_dist( const SmvdCtx &c )
{
  const int          w = c.w, h = c.h;
  unsigned long long acc = 0;
  if( c.satd )
  {
    int tw, th;
    if( w > h && ( h & 7 ) == 0 && ( w & 15 ) == 0 ) { tw = 16; th = 8; }
    else if( w < h && ( w & 7 ) == 0 && ( h & 15 ) == 0 ) { tw = 8; th = 16; }
    else if( w > h && ( h & 3 ) == 0 && ( w & 7 ) == 0 ) { tw = 8; th = 4; }
    else if( w < h && ( w & 3 ) == 0 && ( h & 7 ) == 0 ) { tw = 4; th = 8; }
    else if( ( ( w | h ) & 7 ) == 0 ) { tw = 8; th = 8; }
    else { tw = 4; th = 4; }
    const int tx = w / tw, nt = tx * ( h / th );
    for( int t = threadIdx.x; t < nt; t += THREADS )
    {
      const int      y = ( t / tx ) * th, x = ( t - ( t / tx ) * tx ) * tw;
      const int16_t *o = c.sPat + y * w + x, *q = c.sB + y * w + x;
      if( tw == 16 ) acc += had_tile<16, 8>( o, w, q, w );
      else if( th == 16 ) acc += had_tile<8, 16>( o, w, q, w );
      else if( tw == 8 && th == 8 ) acc += had_tile<8, 8>( o, w, q, w );
      else if( tw == 8 ) acc += had_tile<8, 4>( o, w, q, w );
      else if( th == 8 ) acc += had_tile<4, 8>( o, w, q, w );
      else acc += had_tile<4, 4>( o, w, q, w );
    }
  }
  else
  {
    for( int i = threadIdx.x; i < w * h; i += THREADS ) acc += ( unsigned ) abs( ( int ) c.sPat[i] - ( int ) c.sB[i] );
  }
  acc = wave_reduce_add_u64( acc );
  if( THREADS == 64 ) return acc;
  if( ( threadIdx.x & 63 ) == 0 ) c.sRed[threadIdx.x >> 6] = acc;
  __syncthreads();
  unsigned long long t = 0;
#pragma unroll
  for( int wv = 0; wv < THREADS / 64; wv++ ) t += c.sRed[wv];
  __syncthreads();
  return t;
}

template<int THREADS>
__device__ __forceinline__ unsigned long long cost_b( const SmvdCtx &c, int mvHor, int mvVer )   // pattern in place
{
  predict<THREADS>( c, 1, mvHor, mvVer, LdsOut{ c.sB, c.w } );
  const unsigned long long d = block_dist<THREADS>( c );
  block_sync<THREADS>();
  return ( unsigned long long ) floor( c.fWeight * ( double ) d );
}

template<int THREADS>
__device__ __forceinline__ unsigned long long symmetric_cost( const SmvdCtx &c, const int mvCur[2], const int mvTar[2] )
{
  make_pattern<THREADS>( c, mvCur[0], mvCur[1] );
  return cost_b<THREADS>( c, mvTar[0], mvTar[1] );
}

__device__ __forceinline__ unsigned mv_bits( const SmvdCtx &c, const int mv[2], const int pred[2] )
{
  return eg_bits( prec_dn( mv[0], c.amvrShift ) - prec_dn( pred[0], c.amvrShift ) ) + eg_bits( prec_dn( mv[1], c.amvrShift ) - prec_dn( pred[1], c.amvrShift ) );
}
__device__ __forceinline__ unsigned long long rate( const SmvdCtx &c, unsigned bits ) { return ( unsigned long long ) ( c.lam * bits ); }   // RdCost::getCost

template<int THREADS>
__device__ unsigned long long refine( const SmvdCtx &c, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], unsigned long long minCost, int pattern,
                                      unsigned maxRounds )
{
  const int stepShift = 2 + ( c.imv == 3 ? 1 : ( c.imv << 1 ) );
  const int rounding = pattern == 0 ? 4 : 8, mask = rounding - 1;
  int       start = 0, end = pattern == 0 ? 3 : 7;
  for( unsigned round = 0; round < maxRounds; round++ )
  {
    int       bestDirect = -1;
    const int cx = mvCur[0], cy = mvCur[1];
    for( int idx = start; idx <= end; idx++ )
    {
      const int direct = ( idx + rounding ) & mask;
      const int ox = pattern == 0 ? c_cross[direct][0] : c_diamond[direct][0], oy = pattern == 0 ? c_cross[direct][1] : c_diamond[direct][1];
      const int cand[2] = { cx + ( ox << stepShift ), cy + ( oy << stepShift ) };
      const int pair[2] = { predTar[0] - ( cand[0] - predCur[0] ), predTar[1] - ( cand[1] - predCur[1] ) };
      const unsigned long long cost = rate( c, mv_bits( c, cand, predCur ) ) + symmetric_cost<THREADS>( c, cand, pair );
      if( cost < minCost ) { minCost = cost; mvCur[0] = cand[0]; mvCur[1] = cand[1]; mvTar[0] = pair[0]; mvTar[1] = pair[1]; bestDirect = direct; }
    }
    if( bestDirect == -1 ) break;
    const int step = pattern == 2 ? 2 - ( bestDirect & 1 ) : 1;
    start = bestDirect - step; end = bestDirect + step;
  }
  return minCost;
}

template<int THREADS>
__device__ void symmetric_me( const SmvdCtx &c, const int predCur[2], const int predTar[2], int mvCur[2], int mvTar[2], unsigned long long &cost )
{
  cost = refine<THREADS>( c, predCur, predTar, mvCur, mvTar, cost, 2, 8u >> c.imv );
  cost = refine<THREADS>( c, predCur, predTar, mvCur, mvTar, cost, 0, 1 );
}

struct AmvpLists { int num[2]; int cand[2][2][2]; };

template<int THREADS>
__device__ void check_best_mvp( const SmvdCtx &c, const AmvpLists &a, const int curMv[2], bool skip, int predSym[2][2], int mvpIdxSym[2], unsigned long long &bestCost )
{
  make_pattern<THREADS>( c, curMv[0], curMv[1] );
  const int skip0 = skip ? mvpIdxSym[0] : -1, skip1 = skip ? mvpIdxSym[1] : -1;
  for( int i = 0; i < a.num[0]; i++ )
    for( int k = 0; k < a.num[1]; k++ )
    {
      if( skip0 == i && skip1 == k ) continue;
      const int tx = a.cand[1][k][0] - curMv[0] + a.cand[0][i][0], ty = a.cand[1][k][1] - curMv[1] + a.cand[0][i][1];   // Mv::getSymmvdMv
      unsigned long long cost = cost_b<THREADS>( c, tx, ty );
      cost += rate( c, mv_bits( c, curMv, a.cand[0][i] ) + c.idxBits[i] + c.idxBits[k] );
      if( cost < bestCost )
      {
        bestCost = cost;
        predSym[0][0] = a.cand[0][i][0]; predSym[0][1] = a.cand[0][i][1]; predSym[1][0] = a.cand[1][k][0]; predSym[1][1] = a.cand[1][k][1];
        mvpIdxSym[0] = i; mvpIdxSym[1] = k;
      }
    }
}

template<int THREADS>
__global__ __launch_bounds__( THREADS ) void smvd_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                         vtmhip_smvd_job *__restrict__ jobs, int n, int op )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sMem[];
  __shared__ unsigned long long sRed[4];
  vtmhip_smvd_job &j = jobs[xcd_order( blockIdx.x, n )];
  SmvdCtx c;
  c.w = j.width; c.h = j.height;
  c.org = orgBase + j.orgOff; c.orgStride = j.orgStride; c.refBase = refBase;
  c.refOff[0] = j.refOff[0]; c.refOff[1] = j.refOff[1]; c.refStride[0] = j.refStride[0]; c.refStride[1] = j.refStride[1];
  c.bd = pic.bitDepth; c.imv = j.imv; c.bcw = j.bcwWeightTar ? j.bcwWeightTar : 4;
  c.amvrShift = c.imv == 0 ? 2 : c.imv == 1 ? 4 : c.imv == 2 ? 6 : 3;
  c.horMax = ( pic.picW + 8 - j.puX - 1 ) << 4; c.horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  c.verMax = ( pic.picH + 8 - j.puY - 1 ) << 4; c.verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  c.satd = j.useSatd != 0; c.clip = j.clipBiPred != 0;
  c.lam = j.motionLambda; c.fWeight = c.bcw != 4 ? fabs( ( double ) c.bcw / 8.0 ) : 0.5;   // xGetMEDistortionWeight
  c.idxBits[0] = j.mvpIdxBits[0]; c.idxBits[1] = j.mvpIdxBits[1];
  c.sPat = sMem; c.sB = sMem + c.w * c.h; c.sTmp = c.sB + c.w * c.h; c.sRed = sRed;

  AmvpLists a;
  for( int l = 0; l < 2; l++ )
  {
    a.num[l] = j.numCand[l];
    for( int i = 0; i < 2; i++ ) { a.cand[l][i][0] = j.cand[l][i][0]; a.cand[l][i][1] = j.cand[l][i][1]; }
  }
  int mvCur[2] = { j.mvCur[0], j.mvCur[1] }, mvTar[2] = { j.mvTar[0], j.mvTar[1] };
  int predSym[2][2] = { { j.predSym[0][0], j.predSym[0][1] }, { j.predSym[1][0], j.predSym[1][1] } }, mvpIdxSym[2] = { j.mvpIdxSym[0], j.mvpIdxSym[1] };
  unsigned long long cost = j.cost;

  if( op == VTMHIP_SMVD_COST ) cost = symmetric_cost<THREADS>( c, mvCur, mvTar );
  else if( op == VTMHIP_SMVD_ME ) symmetric_me<THREADS>( c, predSym[0], predSym[1], mvCur, mvTar, cost );
  else if( op == VTMHIP_SMVD_CHECK_MVP ) check_best_mvp<THREADS>( c, a, mvCur, j.skip != 0, predSym, mvpIdxSym, cost );
  else
  {
    for( int l = 0; l < 2; l++ )
      if( a.num[l] > 1 && a.cand[l][0][0] == a.cand[l][1][0] && a.cand[l][0][1] == a.cand[l][1][1] ) a.num[l] = 1;   // :2668-2671
    unsigned long long costStart = ~0ull;
    mvpIdxSym[0] = mvpIdxSym[1] = 0;
    for( int i = 0; i < a.num[0]; i++ )
      for( int k = 0; k < a.num[1]; k++ )
      {
        const unsigned long long d = symmetric_cost<THREADS>( c, a.cand[0][i], a.cand[1][k] );
        if( d < costStart ) { costStart = d; mvpIdxSym[0] = i; mvpIdxSym[1] = k; }
      }
    for( int l = 0; l < 2; l++ ) { predSym[l][0] = a.cand[l][mvpIdxSym[l]][0]; predSym[l][1] = a.cand[l][mvpIdxSym[l]][1]; }
    mvCur[0] = predSym[0][0]; mvCur[1] = predSym[0][1]; mvTar[0] = predSym[1][0]; mvTar[1] = predSym[1][1];
    costStart += rate( c, mv_bits( c, mvCur, predSym[0] ) + c.idxBits[mvpIdxSym[0]] + c.idxBits[mvpIdxSym[1]] );
    // distinct start vectors (smmvdCandsGen :2709-2744), evaluated as they are collected: a later duplicate is skipped exactly as the list would have dropped it
    int seen[VTMHIP_SMVD_MAX_START][2], nc = 0;
    const int numStart = min( ( int ) j.numStart, VTMHIP_SMVD_MAX_START );
    for( int s = 0; s < numStart; s++ )
    {
      int v[2] = { j.starts[s][0], j.starts[s][1] };
      if( s >= j.numFixed )
      {
        if( nc >= 5 ) break;
        if( c.imv ) { v[0] = prec_dn( v[0], c.amvrShift ) * ( 1 << c.amvrShift ); v[1] = prec_dn( v[1], c.amvrShift ) * ( 1 << c.amvrShift ); }   // roundTransPrecInternal2Amvr
      }
      bool dup = false;
      for( int q = 0; q < nc; q++ ) dup |= seen[q][0] == v[0] && seen[q][1] == v[1];
      if( dup ) continue;
      seen[nc][0] = v[0]; seen[nc][1] = v[1]; nc++;
    }
    for( int s = 0; s < nc; s++ )
    {
      const int v[2] = { seen[s][0], seen[s][1] };
      bool checked = false;
      for( int i = 0; i < a.num[0]; i++ ) checked |= v[0] == a.cand[0][i][0] && v[1] == a.cand[0][i][1];
      if( checked ) continue;
      const unsigned long long before = costStart;
      check_best_mvp<THREADS>( c, a, v, false, predSym, mvpIdxSym, costStart );
      if( costStart < before )
      {
        mvCur[0] = v[0]; mvCur[1] = v[1];
        mvTar[0] = predSym[1][0] - v[0] + predSym[0][0]; mvTar[1] = predSym[1][1] - v[1] + predSym[0][1];
      }
    }
    const int                startX = mvCur[0], startY = mvCur[1];
    const unsigned long long mvpCost = rate( c, c.idxBits[mvpIdxSym[0]] + c.idxBits[mvpIdxSym[1]] );
    cost = costStart - mvpCost;
    symmetric_me<THREADS>( c, predSym[0], predSym[1], mvCur, mvTar, cost );
    cost += mvpCost;
    if( startX != mvCur[0] || startY != mvCur[1] ) check_best_mvp<THREADS>( c, a, mvCur, true, predSym, mvpIdxSym, cost );
    cost += rate( c, j.modeBits );
    mvTar[0] = predSym[1][0] - mvCur[0] + predSym[0][0]; mvTar[1] = predSym[1][1] - mvCur[1] + predSym[0][1];
  }
  if( threadIdx.x == 0 )
  {
    j.mvCur[0] = mvCur[0]; j.mvCur[1] = mvCur[1]; j.mvTar[0] = mvTar[0]; j.mvTar[1] = mvTar[1];
    for( int l = 0; l < 2; l++ ) { j.predSym[l][0] = predSym[l][0]; j.predSym[l][1] = predSym[l][1]; j.mvpIdxSym[l] = mvpIdxSym[l]; }
    j.cost = cost;
  }
}

}   // namespace

extern "C" {

int vtmhip_smvd_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs, int n,
                           int maxWidth, int maxHeight, int op )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "bad arguments" );
  VTMHIP_REQUIRE( ctx, op >= VTMHIP_SMVD_COST && op <= VTMHIP_SMVD_SEARCH, "unknown SMVD op" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxHeight >= 4 && maxWidth <= 128 && maxHeight <= 128 && ( maxWidth & 3 ) == 0 && ( maxHeight & 3 ) == 0, "block size out of range" );
  VTMHIP_REQUIRE( ctx, pic->bitDepth >= 8 && pic->bitDepth <= 12, "bit depth out of range" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && d_jobs, "null pointer" );
  // pattern + prediction B + the (h + 7) x w intermediates of the separable filter
  const size_t lds = ( 2 * ( size_t ) maxWidth * maxHeight + ( size_t ) maxWidth * ( maxHeight + 7 ) ) * sizeof( int16_t );
  VTMHIP_TIME_KERNEL( ctx, "smvd_kernel" );
  if( maxWidth * maxHeight <= 1024 )
  {
    hipLaunchKernelGGL( smvd_kernel<64>, dim3( n ), dim3( 64 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op );
  }
  else
  {
    if( lds > 48 * 1024 ) VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( smvd_kernel<256> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
    hipLaunchKernelGGL( smvd_kernel<256>, dim3( n ), dim3( 256 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op );
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_xGetSymmetricCost_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs,
                                        int n, int maxWidth, int maxHeight )
{
  return vtmhip_smvd_batch_dev( ctx, pic, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, VTMHIP_SMVD_COST );
}
int vtmhip_xSymmetricMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                 vtmhip_smvd_job *d_jobs, int n, int maxWidth, int maxHeight )
{
  return vtmhip_smvd_batch_dev( ctx, pic, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, VTMHIP_SMVD_ME );
}
int vtmhip_symmvdCheckBestMvp_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs,
                                         int n, int maxWidth, int maxHeight )
{
  return vtmhip_smvd_batch_dev( ctx, pic, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, VTMHIP_SMVD_CHECK_MVP );
}

}   // extern "C"
