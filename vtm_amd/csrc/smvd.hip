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
#include <cstdlib>

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
                                                         vtmhip_smvd_job *__restrict__ jobs, int n, int op, int maxWidth, int maxHeight )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sMem[];
  __shared__ unsigned long long sRed[4];
  vtmhip_smvd_job &j = jobs[xcd_order( blockIdx.x, n )];
  if( j.width > maxWidth || j.height > maxHeight || j.width < 4 || j.height < 4 )   // the launch sized its LDS for maxWidth x maxHeight: a larger job is the caller's error
  {
    if( threadIdx.x == 0 ) j.cost = ~0ull;
    return;
  }
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
    a.num[l] = min( 2, max( 1, ( int ) j.numCand[l] ) );   // AMVP lists hold one or two candidates
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
    auto snap = [&]( int k, unsigned long long cst ) {      // vtmhip_smvd_job::trace: the block's state between two member calls of the reference
      if( threadIdx.x == 0 ) { j.trace[k].cost = cst; j.trace[k].mv[0] = mvCur[0]; j.trace[k].mv[1] = mvCur[1]; j.trace[k].idx[0] = mvpIdxSym[0]; j.trace[k].idx[1] = mvpIdxSym[1]; }
    };
    snap( 0, costStart );
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
    snap( 1, costStart );
    cost = costStart - mvpCost;
    symmetric_me<THREADS>( c, predSym[0], predSym[1], mvCur, mvTar, cost );
    snap( 2, cost );
    cost += mvpCost;
    if( startX != mvCur[0] || startY != mvCur[1] ) check_best_mvp<THREADS>( c, a, mvCur, true, predSym, mvpIdxSym, cost );
    snap( 3, cost );
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


// =====================================================================================================================================
// Small PUs (8x8, 16x8, 8x16, 16x16; bitDepth <= 10): one LANE per (candidate slot, 8x8 tile).  A pass of the search evaluates up to eight candidates of a PU
// at once -- the eight directions of a diamond round, the four of the cross round, the (i, k) predictor pairs -- so a wave holds 64 / (8 * tiles) PUs, each
// with its own search state (kept redundantly in every lane of the PU's group).  A lane predicts its tile of both lists straight from the reference planes
// (L2 / TCP resident windows), all in registers: horizontal FIR of input row r with v_dot2_i32_i16 on sample pairs, the row pair (r - 1, r) interleaved
// column-wise (v_perm) feeds the vertical v_dot2 of the up to four output rows it belongs to; the target 2 * org - predA and the difference to predB are
// packed 16-bit words, the Hadamard is the packed 8x8 form of had.hpp (two lanes = one 16x8 / 8x16 tile).  No LDS, no barriers: the only cross-lane traffic
// is the tile sum and the (cost, slot) minimum of a group.  The sequential "first candidate below the best so far" rule of the reference equals the
// lexicographic (cost, evaluation order) minimum of a pass compared with the cost on entry.
// The integer phases run the same two passes with the taps {0,0,0,64,0,0,0,0}: for rounded uni-prediction that is exactly filterCopy / the single-pass forms
// ((sum >> s) + 2^(h-1)) >> h == (sum + 32) >> 6 for sum = 2^s q + r.
// =====================================================================================================================================
#ifndef VTMHIP_SMVD_PREFETCH
#define VTMHIP_SMVD_PREFETCH 1
#endif
struct TileFir { int shH, offH, shV, offV, cmax; };

__device__ __forceinline__ void load_taps( int frac, bool alt, v2s c[4] )
{
  const int16_t *t = ( alt && frac == 8 ) ? c_altHpelMc : c_lumaFilterMc[frac];
#pragma unroll
  for( int m = 0; m < 4; m++ ) { c[m].x = t[2 * m]; c[m].y = t[2 * m + 1]; }
}

// the rounded, clipped uni-directional prediction of one 8x8 tile: out( y, packed words of row y ) in row order
template<class RowOut>
__device__ __forceinline__ void pred_tile( const int16_t *__restrict__ ref, int stride, int mvHor, int mvVer, bool alt, const TileFir &f, RowOut out )
{
  v2s ch[4], cv[4];
  load_taps( mvHor & 15, alt, ch );
  load_taps( mvVer & 15, alt, cv );
  const int16_t *src = ref + ( long ) ( ( mvVer >> 4 ) - 3 ) * stride + ( mvHor >> 4 ) - 3;
  int      acc[64];
  unsigned prevH[4] = { 0, 0, 0, 0 };
#if VTMHIP_SMVD_PREFETCH
  // the next input row's two loads are issued BEFORE this row's filter work (the scheduling barrier at the end of a row keeps the compiler from hoisting all fifteen rows' loads --
  // registers -- but it also kept every row's loads behind the previous row's arithmetic: a full memory latency per row at two waves per SIMD)
  Pel8u na = *reinterpret_cast<const Pel8u *>( src ), nb = *reinterpret_cast<const Pel8u *>( src + 8 );
#endif
#pragma unroll
  for( int r = 0; r < 15; r++ )
  {
#if VTMHIP_SMVD_PREFETCH
    const Pel8u a = na, b = nb;
    if( r < 14 ) { na = *reinterpret_cast<const Pel8u *>( src + ( long ) ( r + 1 ) * stride ); nb = *reinterpret_cast<const Pel8u *>( src + ( long ) ( r + 1 ) * stride + 8 ); }
#else
    const Pel8u a = *reinterpret_cast<const Pel8u *>( src + ( long ) r * stride ), b = *reinterpret_cast<const Pel8u *>( src + ( long ) r * stride + 8 );
#endif
    const unsigned d[8] = { a.v[0], a.v[1], a.v[2], a.v[3], b.v[0], b.v[1], b.v[2], b.v[3] };   // d[m] = samples (2m, 2m + 1)
    unsigned e[7];                                                                             // e[m] = samples (2m + 1, 2m + 2)
#pragma unroll
    for( int m = 0; m < 7; m++ ) e[m] = __builtin_amdgcn_alignbit( d[m + 1], d[m], 16 );
    unsigned curH[4];
#pragma unroll
    for( int q = 0; q < 8; q++ )
    {
      int sum = f.offH;
#pragma unroll
      for( int m = 0; m < 4; m++ )
      {
        const unsigned pw = ( q & 1 ) ? e[( q >> 1 ) + m] : d[( q >> 1 ) + m];
        v2s pv;
        __builtin_memcpy( &pv, &pw, 4 );
        sum = __builtin_amdgcn_sdot2( pv, ch[m], sum, false );
      }
      const unsigned hv = ( unsigned ) ( sum >> f.shH ) & 0xffffu;
      if( q & 1 ) curH[q >> 1] |= hv << 16; else curH[q >> 1] = hv;
    }
    if( r >= 1 )
    {
      v2s pr[8];   // column x: (row r - 1, row r)
#pragma unroll
      for( int k = 0; k < 4; k++ )
      {
        const unsigned lo = __builtin_amdgcn_perm( curH[k], prevH[k], 0x05040100u ), hi = __builtin_amdgcn_perm( curH[k], prevH[k], 0x07060302u );
        __builtin_memcpy( &pr[2 * k], &lo, 4 );
        __builtin_memcpy( &pr[2 * k + 1], &hi, 4 );
      }
      const int q = r - 1;
#pragma unroll
      for( int m = 0; m < 4; m++ )
      {
        const int y = q - 2 * m;
        if( y >= 0 && y < 8 )
        {
#pragma unroll
          for( int x = 0; x < 8; x++ ) acc[y * 8 + x] = __builtin_amdgcn_sdot2( pr[x], cv[m], m == 0 ? f.offV : acc[y * 8 + x], false );
        }
      }
      if( r >= 7 )   // output row r - 7 is complete
      {
        const int y = r - 7;
        unsigned  w[4];
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const int lo = min( f.cmax, max( 0, acc[y * 8 + 2 * k] >> f.shV ) ), hi = min( f.cmax, max( 0, acc[y * 8 + 2 * k + 1] >> f.shV ) );
          w[k] = ( unsigned ) lo | ( ( unsigned ) hi << 16 );
        }
        out( y, w );
      }
    }
#pragma unroll
    for( int k = 0; k < 4; k++ ) prevH[k] = curH[k];
    __builtin_amdgcn_sched_barrier( 0 );   // keep the row loads from being hoisted together (register pressure -> occupancy)
  }
}

// The packed Hadamard of had.hpp runs three butterfly levels in 16 bits: |difference| <= 4095.  The BCW weight -2 makes the target -4 org + 5 pred (up to 5115
// for 10-bit samples) and the difference to predB up to 6138: two packed levels (x4 fits 16 bits), the third vertical one and the horizontal ones in 32 bits.
__device__ __forceinline__ void rows_unpack( const v2s r[4], int m[8] )
{
#pragma unroll
  for( int k = 0; k < 4; k++ ) { m[2 * k] = r[k].x; m[2 * k + 1] = r[k].y; }
}
template<bool PAIR>
__device__ __forceinline__ unsigned satd8_wide( v2s D[8][4] )
{
#pragma unroll
  for( int len = 1; len < 4; len <<= 1 )
#pragma unroll
    for( int i = 0; i < 8; i += len << 1 )
#pragma unroll
      for( int jj = i; jj < i + len; jj++ )
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
          const v2s a = D[jj][k], b = D[jj + len][k];
          D[jj][k]       = a + b;
          D[jj + len][k] = a - b;
        }
  int t = 0, dc = 0;
#pragma unroll
  for( int y = 0; y < 4; y++ )
  {
    int m0[8], m1[8], a[8], b[8];
    rows_unpack( D[y], m0 );
    rows_unpack( D[y + 4], m1 );
#pragma unroll
    for( int x = 0; x < 8; x++ ) { a[x] = m0[x] + m1[x]; b[x] = m0[x] - m1[x]; }
    wht1d_inl8( a );
    wht1d_inl8( b );
    if( y == 0 ) dc = PAIR ? abs( a[0] + dpp_swap1( a[0] ) ) : abs( a[0] );
#pragma unroll
    for( int x = 0; x < 8; x++ )
    {
      const int ua = abs( a[x] ), ub = abs( b[x] );
      t += PAIR ? max( ua, dpp_swap1( ua ) ) + max( ub, dpp_swap1( ub ) ) : ua + ub;
    }
  }
  if( PAIR ) return satd_pair_norm( t << 1, dc );
  t = t - dc + ( dc >> 2 );
  return ( unsigned ) ( ( t + 2 ) >> 2 );
}

struct TileJob   // what a lane needs of its PU for the evaluation (workgroup- / group-uniform)
{
  const int16_t *org, *refA, *refB;   // PU origins (MV 0,0)
  int orgStride, strideA, strideB;
  int horMin, horMax, verMin, verMax;
  int w0, w1;                         // BCW form of removeHighFreq (w0 == 0: the default 2 * org - pred)
  bool clip, alt, satd, wide;         // wide: differences beyond the packed Hadamard's range (BCW weight -2)
  TileFir f;
};

// distortion of the tile at (tx, ty) (in tiles) for the vector pair (A, B); PAIR: lanes 2k, 2k + 1 hold the halves of one 16x8 / 8x16 Hadamard tile (both return
// its value)
template<bool PAIR>
__device__ __forceinline__ unsigned tile_eval( const TileJob &t, int tx, int ty, int ax, int ay, int bx, int by )
{
  ax = min( t.horMax, max( t.horMin, ax ) ); ay = min( t.verMax, max( t.verMin, ay ) );   // clipMv
  bx = min( t.horMax, max( t.horMin, bx ) ); by = min( t.verMax, max( t.verMin, by ) );
  const int16_t *org = t.org + ( long ) ( ty * 8 ) * t.orgStride + tx * 8;
  v2s D[8][4];
#if VTMHIP_SMVD_PREFETCH
  Pel8u onx = *reinterpret_cast<const Pel8u *>( org );      // the original block's rows arrive one output row ahead, like the reference rows in pred_tile
#endif
  pred_tile( t.refA + ( long ) ( ty * 8 ) * t.strideA + tx * 8, t.strideA, ax, ay, t.alt, t.f, [&]( int y, const unsigned w[4] )
  {
#if VTMHIP_SMVD_PREFETCH
    const Pel8u o = onx;
    if( y < 7 ) onx = *reinterpret_cast<const Pel8u *>( org + ( long ) ( y + 1 ) * t.orgStride );
#else
    const Pel8u o = *reinterpret_cast<const Pel8u *>( org + ( long ) y * t.orgStride );
#endif
#pragma unroll
    for( int k = 0; k < 4; k++ )
    {
      const int o0 = ( int ) ( short ) ( o.v[k] & 0xffffu ), o1 = ( int ) o.v[k] >> 16, p0 = ( int ) ( w[k] & 0xffffu ), p1 = ( int ) ( w[k] >> 16 );
      int r0 = t.w0 ? ( o0 * t.w0 - p0 * t.w1 + ( 1 << 15 ) ) >> 16 : 2 * o0 - p0, r1 = t.w0 ? ( o1 * t.w0 - p1 * t.w1 + ( 1 << 15 ) ) >> 16 : 2 * o1 - p1;
      if( t.clip ) { r0 = min( t.f.cmax, max( 0, r0 ) ); r1 = min( t.f.cmax, max( 0, r1 ) ); }
      D[y][k].x = ( short ) r0; D[y][k].y = ( short ) r1;
    }
  } );
  pred_tile( t.refB + ( long ) ( ty * 8 ) * t.strideB + tx * 8, t.strideB, bx, by, t.alt, t.f, [&]( int y, const unsigned w[4] )
  {
#pragma unroll
    for( int k = 0; k < 4; k++ )
    {
      v2s b;
      __builtin_memcpy( &b, &w[k], 4 );
      D[y][k] = D[y][k] - b;     // |pattern - predB| <= 2 * 1023 + 1023; BCW weights 3, 5, 10 stay below that, -2 reaches 6138 (t.wide)
    }
  } );
  if( !t.satd )
  {
    int s = 0;
#pragma unroll
    for( int y = 0; y < 8; y++ )
#pragma unroll
      for( int k = 0; k < 4; k++ ) s += abs( ( int ) D[y][k].x ) + abs( ( int ) D[y][k].y );
    return ( unsigned ) s;
  }
  if( t.wide ) return satd8_wide<PAIR>( D );
  return PAIR ? satd8_pair_packed( D ) : satd8_packed( D );
}

__device__ __forceinline__ int      shfl_x( int v, int m ) { return __shfl_xor( v, m, 64 ); }
__device__ __forceinline__ unsigned long long shfl_x64( unsigned long long v, int m )
{
  return ( ( unsigned long long ) ( unsigned ) __shfl_xor( ( int ) ( v >> 32 ), m, 64 ) << 32 ) | ( unsigned ) __shfl_xor( ( int ) v, m, 64 );
}

enum { PH_INIT = 0, PH_STARTS, PH_DIAMOND, PH_CROSS, PH_FINAL, PH_COST, PH_DONE };

// NW == 0: "group" form for PUs of at most eight tiles -- a lane is (PU of the wave, candidate slot, tile), 64 / (NSLOT * tiles) PUs per wave.
// NW >= 1: one PU per workgroup of NW waves -- the (slot, tile) items of a pass are dealt to the lanes, the per-slot sums meet in LDS.
#ifndef VTMHIP_SMVD_G8
#define VTMHIP_SMVD_G8 0    // waves per PU of the 8-tile shapes (0: group form, two PUs per wave)
#endif
#ifndef VTMHIP_SMVD_G16
#define VTMHIP_SMVD_G16 1   // ... of the 16-tile shapes (measured: one wave 2.58 ms for the picture's SMVD stage, group form 2.63, two waves 2.71)
#endif
template<int TX, int TY, int NW, int NSLOT>
__global__ __launch_bounds__( NW ? 64 * NW : 64 ) void smvd_tile_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                                        vtmhip_smvd_job *__restrict__ jobs, int n, int op )
{
  // group form: NSLOT candidate slots per PU and pass step; a pass of the search (up to 8 candidates: the diamond's first round, two start vectors x four predictor
  // pairs) runs as ceil( candidates / NSLOT ) steps.  Four slots instead of eight: the passes with <= 4 candidates (predictor pairs, later diamond rounds, cross, final
  // check) no longer carry four idle slots, and a wave holds twice as many PUs.
  constexpr int  T = TX * TY, LPP = ( NW ? 8 : NSLOT ) * T, PPW = NW ? 1 : 64 / LPP, NT = NW ? 64 * NW : 64;
  constexpr bool PAIR = TX != TY, GROUP = NW == 0;
  static_assert( NW != 0 || NSLOT * T <= 64, "the group form keeps a PU inside one wave" );
  __shared__ unsigned sDist[8];
  const int lane = threadIdx.x, g = GROUP ? lane / LPP : 0, l = GROUP ? lane - g * LPP : lane, slot = GROUP ? l / T : 0, tile = GROUP ? l - slot * T : 0;
  auto tile_xy = [&]( int tl, int &tx, int &ty ) { if( TX >= TY ) { ty = tl / TX; tx = tl - ty * TX; } else { tx = tl / TY; ty = tl - tx * TY; } };   // pair halves adjacent
  // XCD-aware order of the WORKGROUPS (the hardware deals them round-robin over the 8 XCDs): workgroup b takes the PPW consecutive PUs of position xcd_order( b ) -- neighbouring
  // PUs (raster order: overlapping reference windows) then run on one XCD and share its L2.  (Round 3 permuted the PU index instead: with PPW > 1 that scattered a wave's PUs over
  // eight distant regions of the picture and sent neighbours to different XCDs -- 1.2 GB of HBM-side traffic for the 8x8 launch.)
  const int puRaw = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x ) * PPW + g;
  const bool live = puRaw < n;
  vtmhip_smvd_job &j = jobs[live ? puRaw : 0];

  TileJob t;
  t.orgStride = j.orgStride; t.strideA = j.refStride[0]; t.strideB = j.refStride[1];
  t.org = orgBase + j.orgOff; t.refA = refBase + j.refOff[0]; t.refB = refBase + j.refOff[1];
  t.horMax = ( pic.picW + 8 - j.puX - 1 ) << 4; t.horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  t.verMax = ( pic.picH + 8 - j.puY - 1 ) << 4; t.verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  const int bcw = j.bcwWeightTar ? j.bcwWeightTar : 4;
  const int normalizer = bcw != 4 ? ( ( 1 << 16 ) + ( bcw > 0 ? ( bcw >> 1 ) : -( bcw >> 1 ) ) ) / bcw : 0;
  t.w0 = normalizer * 8; t.w1 = ( 8 - bcw ) * normalizer;
  t.clip = j.clipBiPred != 0; t.alt = j.imv == 3; t.satd = j.useSatd != 0; t.wide = bcw < 0 && !t.clip;
  {
    const int bd = pic.bitDepth, headRoom = max( 2, 14 - bd );
    t.f.shH = 6 - headRoom; t.f.offH = -( 8192 << t.f.shH ); t.f.shV = 6 + headRoom; t.f.offV = ( 1 << ( t.f.shV - 1 ) ) + ( 8192 << 6 ); t.f.cmax = ( 1 << bd ) - 1;
  }
  const int    imv = j.imv, amvrShift = imv == 0 ? 2 : imv == 1 ? 4 : imv == 2 ? 6 : 3, stepShift = 2 + ( imv == 3 ? 1 : ( imv << 1 ) );
  const double lam = j.motionLambda, fWeight = bcw != 4 ? fabs( ( double ) bcw / 8.0 ) : 0.5;
  const unsigned idxBits0 = j.mvpIdxBits[0], idxBits1 = j.mvpIdxBits[1];
  auto idx_bits = [&]( int i ) { return i ? idxBits1 : idxBits0; };
  auto rate = [&]( unsigned bits ) { return ( unsigned long long ) ( lam * bits ); };
  auto mvbits = [&]( int mx, int my, int px, int py )
  { return eg_bits( prec_dn( mx, amvrShift ) - prec_dn( px, amvrShift ) ) + eg_bits( prec_dn( my, amvrShift ) - prec_dn( py, amvrShift ) ); };

  // AMVP lists
  int cnd[2][2][2], num0 = min( 2, max( 1, ( int ) j.numCand[0] ) ), num1 = min( 2, max( 1, ( int ) j.numCand[1] ) );   // AMVP lists hold one or two candidates
#pragma unroll
  for( int a = 0; a < 2; a++ )
#pragma unroll
    for( int i = 0; i < 2; i++ ) { cnd[a][i][0] = j.cand[a][i][0]; cnd[a][i][1] = j.cand[a][i][1]; }
  auto cand = [&]( int a, int i, int c ) { return i ? cnd[a][1][c] : cnd[a][0][c]; };

  // search state of this lane's PU
  int mvCur[2] = { j.mvCur[0], j.mvCur[1] }, mvTar[2] = { j.mvTar[0], j.mvTar[1] };
  int pred[2][2] = { { j.predSym[0][0], j.predSym[0][1] }, { j.predSym[1][0], j.predSym[1][1] } }, idxSym[2] = { j.mvpIdxSym[0], j.mvpIdxSym[1] };
  unsigned long long cost = j.cost, mvpCost = 0;
  int phase = !live ? PH_DONE : op == VTMHIP_SMVD_COST ? PH_COST : op == VTMHIP_SMVD_ME ? PH_DIAMOND : op == VTMHIP_SMVD_CHECK_MVP ? PH_FINAL : PH_INIT;
  int round = 0, dStart = 0, dEnd = 7, startX = 0, startY = 0, si = 0, sj = -1;   // si, sj: the (up to) two start vectors of a STARTS pass
  const int maxRounds = 8 >> imv;
  const bool skipPair = op == VTMHIP_SMVD_CHECK_MVP ? j.skip != 0 : true;
  unsigned startMask = 0;
  const int numFixed = j.numFixed;
  auto start_vec = [&]( int s, int c )
  {
    int v = j.starts[s][c];
    if( s >= numFixed && imv ) v = prec_dn( v, amvrShift ) * ( 1 << amvrShift );   // roundTransPrecInternal2Amvr
    return v;
  };
  if( phase == PH_INIT )
  {
    if( num0 > 1 && cnd[0][0][0] == cnd[0][1][0] && cnd[0][0][1] == cnd[0][1][1] ) num0 = 1;   // :2668-2671
    if( num1 > 1 && cnd[1][0][0] == cnd[1][1][0] && cnd[1][0][1] == cnd[1][1][1] ) num1 = 1;
    // distinct start vectors (smmvdCandsGen): bit s = raw entry s is evaluated
    const int numStart = min( ( int ) j.numStart, VTMHIP_SMVD_MAX_START );
    int nc = 0;
    for( int s = 0; s < numStart; s++ )
    {
      if( s >= numFixed && nc >= 5 ) break;
      const int vx = start_vec( s, 0 ), vy = start_vec( s, 1 );
      bool dup = false;
      for( int q = 0; q < s; q++ ) dup |= ( ( startMask >> q ) & 1 ) && start_vec( q, 0 ) == vx && start_vec( q, 1 ) == vy;
      if( !dup ) { startMask |= 1u << s; nc++; }
    }
    cost = ~0ull;
  }
  // the next start vector that is not one of the searched list's predictors (those were covered by the predictor pairs), or -1
  auto next_start = [&]( int from )
  {
    for( int s = from; s < VTMHIP_SMVD_MAX_START; s++ )
    {
      if( !( ( startMask >> s ) & 1 ) ) continue;
      const int vx = start_vec( s, 0 ), vy = start_vec( s, 1 );
      bool checked = false;
      for( int i = 0; i < num0; i++ ) checked |= vx == cand( 0, i, 0 ) && vy == cand( 0, i, 1 );
      if( !checked ) return s;
    }
    return -1;
  };
  const bool tracing = live && l == 0 && ( op & 0xff ) == VTMHIP_SMVD_SEARCH;
  auto snap = [&]( int k, unsigned long long cst ) {      // vtmhip_smvd_job::trace: the block's state between two member calls of the reference
    if( tracing ) { j.trace[k].cost = cst; j.trace[k].mv[0] = mvCur[0]; j.trace[k].mv[1] = mvCur[1]; j.trace[k].idx[0] = idxSym[0]; j.trace[k].idx[1] = idxSym[1]; }
  };
  auto to_me = [&]()   // ME preparation (:2765-2770)
  {
    snap( 1, cost );
    startX = mvCur[0]; startY = mvCur[1];
    mvpCost = rate( idx_bits( idxSym[0] ) + idx_bits( idxSym[1] ) );
    cost -= mvpCost;
    phase = PH_DIAMOND; round = 0; dStart = 0; dEnd = 7;
  };
  auto finish = [&]()   // :2781-2786
  {
    snap( 3, cost );
    cost += rate( j.modeBits );
    mvTar[0] = pred[1][0] - mvCur[0] + pred[0][0]; mvTar[1] = pred[1][1] - mvCur[1] + pred[0][1];
    phase = PH_DONE;
  };
  // candidate of slot s in the current pass: vectors, rate bits; false: the slot is empty
  auto slot_params = [&]( int s, int &ax, int &ay, int &bx, int &by, unsigned &bits )
  {
    bool valid = false;
    ax = ay = bx = by = 0; bits = 0;
    if( phase == PH_INIT || phase == PH_STARTS || phase == PH_FINAL )
    {
      // STARTS: two start vectors per pass, slots 0..3 the pairs of the first, 4..7 of the second (the order the reference visits them in)
      const int second = phase == PH_STARTS ? s >> 2 : 0, ps = phase == PH_STARTS ? s & 3 : s;
      int pi = ps / num1, pk = ps - pi * num1;
      valid = ps < num0 * num1 && ( !second || sj >= 0 );
      if( !valid ) pi = pk = 0;
      if( phase == PH_INIT ) { ax = cand( 0, pi, 0 ); ay = cand( 0, pi, 1 ); bx = cand( 1, pk, 0 ); by = cand( 1, pk, 1 ); }
      else
      {
        const int sv = second && sj >= 0 ? sj : si;
        ax = phase == PH_STARTS ? start_vec( sv, 0 ) : mvCur[0]; ay = phase == PH_STARTS ? start_vec( sv, 1 ) : mvCur[1];
        bx = cand( 1, pk, 0 ) - ax + cand( 0, pi, 0 ); by = cand( 1, pk, 1 ) - ay + cand( 0, pi, 1 );   // Mv::getSymmvdMv
        bits = mvbits( ax, ay, cand( 0, pi, 0 ), cand( 0, pi, 1 ) ) + idx_bits( pi ) + idx_bits( pk );
        if( phase == PH_FINAL && skipPair && pi == idxSym[0] && pk == idxSym[1] ) valid = false;
      }
    }
    else if( phase == PH_DIAMOND || phase == PH_CROSS )
    {
      const int idx = dStart + s;
      valid = idx <= dEnd;
      const int direct = phase == PH_CROSS ? ( idx + 4 ) & 3 : ( idx + 8 ) & 7;
      const int ox = phase == PH_CROSS ? c_cross[direct][0] : c_diamond[direct][0], oy = phase == PH_CROSS ? c_cross[direct][1] : c_diamond[direct][1];
      ax = mvCur[0] + ( ox << stepShift ); ay = mvCur[1] + ( oy << stepShift );
      bx = pred[1][0] - ( ax - pred[0][0] ); by = pred[1][1] - ( ay - pred[0][1] );
      bits = mvbits( ax, ay, pred[0][0], pred[0][1] );
    }
    else if( phase == PH_COST ) { valid = s == 0; ax = mvCur[0]; ay = mvCur[1]; bx = mvTar[0]; by = mvTar[1]; }
    return valid;
  };
  auto slot_cost = [&]( bool valid, unsigned d, unsigned bits )
  { return valid ? ( unsigned long long ) floor( fWeight * ( double ) d ) + ( phase == PH_INIT || phase == PH_COST ? 0ull : rate( bits ) ) : ~0ull; };

  while( GROUP ? __any( phase != PH_DONE ) : phase != PH_DONE )
  {
    unsigned long long c;
    int bs;
    if( GROUP )
    {
      // candidates of this PU's pass (slot_params marks the empty ones inside that range)
      const int nSl = phase == PH_DONE ? 0 : ( phase == PH_DIAMOND || phase == PH_CROSS ) ? dEnd - dStart + 1 : phase == PH_COST ? 1 : phase == PH_STARTS && sj >= 0 ? 8 : num0 * num1;
      int tx, ty;
      tile_xy( tile, tx, ty );
      c = ~0ull; bs = 0;
#pragma unroll 1
      for( int s0 = 0; __any( s0 < nSl ); s0 += NSLOT )
      {
        const int s = s0 + slot;
        int ax, ay, bx, by;
        unsigned bits;
        const bool valid = slot_params( s, ax, ay, bx, by, bits ) && s < nSl;
        unsigned d = tile_eval<PAIR>( t, tx, ty, ax, ay, bx, by );
        // PU sum over the tile lanes of the slot; a Hadamard pair already holds its tile's value in both lanes
#pragma unroll
        for( int off = 1; off < T; off <<= 1 )
          if( off > 1 || !PAIR || !t.satd ) d += ( unsigned ) shfl_x( ( int ) d, off );
        unsigned long long cs = slot_cost( valid, d, bits );
        int ss = s;
        // first minimum of the step over the slots of the group
#pragma unroll
        for( int off = T; off < LPP; off <<= 1 )
        {
          const unsigned long long oc = shfl_x64( cs, off );
          const int                os = shfl_x( ss, off );
          if( oc < cs || ( oc == cs && os < ss ) ) { cs = oc; ss = os; }
        }
        if( cs < c ) { c = cs; bs = ss; }   // steps run in candidate order: a tie keeps the earlier step's candidate
      }
    }
    else
    {
      if( lane < 8 ) sDist[lane] = 0;
      __syncthreads();
      const int nSlots = ( phase == PH_DIAMOND || phase == PH_CROSS ) ? dEnd - dStart + 1 : phase == PH_COST ? 1 : phase == PH_STARTS && sj >= 0 ? 8 : num0 * num1;
#pragma unroll 1
      for( int it = lane; it < nSlots * T; it += NT )
      {
        const int s = it / T, tl = it - s * T;
        int ax, ay, bx, by, tx, ty;
        unsigned bits;
        if( !slot_params( s, ax, ay, bx, by, bits ) ) continue;    // the skipped pair of the final check (whole tiles: pair lanes stay together)
        tile_xy( tl, tx, ty );
        const unsigned d = tile_eval<PAIR>( t, tx, ty, ax, ay, bx, by );
        if( !PAIR || !t.satd || !( tl & 1 ) ) atomicAdd( &sDist[s], d );
      }
      __syncthreads();
      c = ~0ull; bs = 0;
      for( int s = 0; s < nSlots; s++ )
      {
        int ax, ay, bx, by;
        unsigned bits;
        const bool valid = slot_params( s, ax, ay, bx, by, bits );
        const unsigned long long cs = slot_cost( valid, sDist[s], bits );
        if( cs < c ) { c = cs; bs = s; }
      }
      __syncthreads();
    }
    const bool found = c < cost;
    // ---- state update (group- / workgroup-uniform) ----
    if( phase == PH_INIT )
    {
      const int bi = bs / num1, bk = bs - bi * num1;
      idxSym[0] = bi; idxSym[1] = bk;
      pred[0][0] = cand( 0, bi, 0 ); pred[0][1] = cand( 0, bi, 1 ); pred[1][0] = cand( 1, bk, 0 ); pred[1][1] = cand( 1, bk, 1 );
      mvCur[0] = pred[0][0]; mvCur[1] = pred[0][1]; mvTar[0] = pred[1][0]; mvTar[1] = pred[1][1];
      snap( 0, c );
      cost = c + rate( mvbits( mvCur[0], mvCur[1], pred[0][0], pred[0][1] ) + idx_bits( bi ) + idx_bits( bk ) );
      si = next_start( 0 );
      sj = si >= 0 ? next_start( si + 1 ) : -1;
      if( si >= 0 ) phase = PH_STARTS; else to_me();
    }
    else if( phase == PH_STARTS )
    {
      if( found )   // the first minimum over (start vector, i, k) in the reference's order = what its one-by-one strict comparisons keep
      {
        const int sv = ( bs >> 2 ) ? sj : si, ps = bs & 3, bi = ps / num1, bk = ps - bi * num1;
        cost = c; idxSym[0] = bi; idxSym[1] = bk;
        pred[0][0] = cand( 0, bi, 0 ); pred[0][1] = cand( 0, bi, 1 ); pred[1][0] = cand( 1, bk, 0 ); pred[1][1] = cand( 1, bk, 1 );
        mvCur[0] = start_vec( sv, 0 ); mvCur[1] = start_vec( sv, 1 );
        mvTar[0] = pred[1][0] - mvCur[0] + pred[0][0]; mvTar[1] = pred[1][1] - mvCur[1] + pred[0][1];
      }
      si = next_start( ( sj >= 0 ? sj : si ) + 1 );
      sj = si >= 0 ? next_start( si + 1 ) : -1;
      if( si < 0 ) to_me();
    }
    else if( phase == PH_DIAMOND || phase == PH_CROSS )
    {
      int direct = 0;
      if( found )
      {
        const int idx = dStart + bs;
        direct = phase == PH_CROSS ? ( idx + 4 ) & 3 : ( idx + 8 ) & 7;
        const int ox = phase == PH_CROSS ? c_cross[direct][0] : c_diamond[direct][0], oy = phase == PH_CROSS ? c_cross[direct][1] : c_diamond[direct][1];
        cost = c;
        mvCur[0] += ox << stepShift; mvCur[1] += oy << stepShift;
        mvTar[0] = pred[1][0] - ( mvCur[0] - pred[0][0] ); mvTar[1] = pred[1][1] - ( mvCur[1] - pred[0][1] );
      }
      if( phase == PH_DIAMOND )
      {
        round++;
        if( found && round < maxRounds ) { const int step = 2 - ( direct & 1 ); dStart = direct - step; dEnd = direct + step; }
        else { phase = PH_CROSS; dStart = 0; dEnd = 3; }
      }
      else if( op == VTMHIP_SMVD_ME ) phase = PH_DONE;
      else
      {
        snap( 2, cost );
        cost += mvpCost;
        if( startX != mvCur[0] || startY != mvCur[1] ) phase = PH_FINAL; else finish();
      }
    }
    else if( phase == PH_FINAL )
    {
      if( found )
      {
        const int bi = bs / num1, bk = bs - bi * num1;
        cost = c; idxSym[0] = bi; idxSym[1] = bk;
        pred[0][0] = cand( 0, bi, 0 ); pred[0][1] = cand( 0, bi, 1 ); pred[1][0] = cand( 1, bk, 0 ); pred[1][1] = cand( 1, bk, 1 );
      }
      if( op == VTMHIP_SMVD_CHECK_MVP ) phase = PH_DONE; else finish();
    }
    else if( phase == PH_COST ) { cost = c; phase = PH_DONE; }
  }
  if( live && l == 0 )
  {
    j.mvCur[0] = mvCur[0]; j.mvCur[1] = mvCur[1]; j.mvTar[0] = mvTar[0]; j.mvTar[1] = mvTar[1];
    for( int a = 0; a < 2; a++ ) { j.predSym[a][0] = pred[a][0]; j.predSym[a][1] = pred[a][1]; j.mvpIdxSym[a] = idxSym[a]; }
    j.cost = cost;
  }
}

template<int TX, int TY, int NW>
void launch_tile( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs, int n, int op )
{
  if constexpr( NW != 0 )
  {
    hipLaunchKernelGGL( ( smvd_tile_kernel<TX, TY, NW, 8> ), dim3( n ), dim3( 64 * NW ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op );
  }
  else
  {
  // group form: four candidate slots per PU fill the lanes best (the passes of a search hold 3 .. 8 candidates) and are the choice when the batch is larger than
  // the machine; a batch of at most two waves per SIMD (a rank's share of a picture on eight GPUs) is bound by the latency of ONE wave's
  // search, and eight slots halve the pass steps of that wave (1/8 share of a 4K picture: 2.60 -> 2.52 ms)
  constexpr int T = TX * TY, WIDE = 8 * T <= 64 ? 8 : 4, PPW_WIDE = 64 / ( WIDE * T ), PPW4 = 64 / ( 4 * T );
  const long wavesWide = ( ( long ) n + PPW_WIDE - 1 ) / PPW_WIDE;
  if( WIDE == 8 && wavesWide <= 2L * 4 * ctx->numCUs )
    hipLaunchKernelGGL( ( smvd_tile_kernel<TX, TY, 0, WIDE> ), dim3( ( n + PPW_WIDE - 1 ) / PPW_WIDE ), dim3( 64 ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op );
  else
    hipLaunchKernelGGL( ( smvd_tile_kernel<TX, TY, 0, 4> ), dim3( ( n + PPW4 - 1 ) / PPW4 ), dim3( 64 ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op );
  }
}

}   // namespace

extern "C" {

int vtmhip_smvd_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs, int n,
                           int maxWidth, int maxHeight, int op )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, pic && n >= 0, "bad arguments" );
  const bool uniform = ( op & VTMHIP_SMVD_UNIFORM ) != 0;   // every job is exactly maxWidth x maxHeight
  op &= ~VTMHIP_SMVD_UNIFORM;
  VTMHIP_REQUIRE( ctx, op >= VTMHIP_SMVD_COST && op <= VTMHIP_SMVD_SEARCH, "unknown SMVD op" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxHeight >= 4 && maxWidth <= 128 && maxHeight <= 128 && ( maxWidth & 3 ) == 0 && ( maxHeight & 3 ) == 0, "block size out of range" );
  VTMHIP_REQUIRE( ctx, pic->bitDepth >= 8 && pic->bitDepth <= 12, "bit depth out of range" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_refBase && d_jobs, "null pointer" );
  // pattern + prediction B + the (h + 7) x w intermediates of the separable filter
  const size_t lds = ( 2 * ( size_t ) maxWidth * maxHeight + ( size_t ) maxWidth * ( maxHeight + 7 ) ) * sizeof( int16_t );
  const bool tileForm = uniform && pic->bitDepth <= 10 && maxWidth >= 8 && maxHeight >= 8 && !getenv( "VTMHIP_SMVD_NO_TILE" );
  VTMHIP_TIME_KERNEL( ctx, tileForm ? "smvd_tile_kernel" : "smvd_kernel" );
  bool tiled = true;
  if( tileForm )
  {
    // group form up to four tiles; above: one PU per workgroup, waves by the number of (candidate, tile) items of a pass (8 candidates x tiles)
    switch( maxWidth * 256 + maxHeight )
    {
    case 8 * 256 + 8:     launch_tile<1, 1, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 16 * 256 + 8:    launch_tile<2, 1, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 8 * 256 + 16:    launch_tile<1, 2, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 16 * 256 + 16:   launch_tile<2, 2, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 32 * 256 + 8:    launch_tile<4, 1, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 8 * 256 + 32:    launch_tile<1, 4, 0>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 32 * 256 + 16:   launch_tile<4, 2, VTMHIP_SMVD_G8>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 16 * 256 + 32:   launch_tile<2, 4, VTMHIP_SMVD_G8>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 32 * 256 + 32:   launch_tile<4, 4, VTMHIP_SMVD_G16>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 64 * 256 + 16:   launch_tile<8, 2, VTMHIP_SMVD_G16>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 16 * 256 + 64:   launch_tile<2, 8, VTMHIP_SMVD_G16>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 64 * 256 + 32:   launch_tile<8, 4, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 32 * 256 + 64:   launch_tile<4, 8, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 64 * 256 + 64:   launch_tile<8, 8, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 128 * 256 + 64:  launch_tile<16, 8, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 64 * 256 + 128:  launch_tile<8, 16, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    case 128 * 256 + 128: launch_tile<16, 16, 4>( ctx, pic, d_orgBase, d_refBase, d_jobs, n, op ); break;
    default: tiled = false;
    }
  }
  else tiled = false;
  if( tiled ) {}
  else if( maxWidth * maxHeight <= 1024 )
  {
    hipLaunchKernelGGL( smvd_kernel<64>, dim3( n ), dim3( 64 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op, maxWidth, maxHeight );
  }
  else
  {
    if( lds > 48 * 1024 ) VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( smvd_kernel<256> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
    hipLaunchKernelGGL( smvd_kernel<256>, dim3( n ), dim3( 256 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_jobs, n, op, maxWidth, maxHeight );
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
