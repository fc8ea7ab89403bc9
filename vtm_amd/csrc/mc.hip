// mc.hip -- motion compensation of whole blocks (luma 8-tap, 4:2:0 chroma 4-tap) and the two PelBufferOps of bi-predictive ME.
//
// Reference: CommonLib/InterPrediction.cpp xPredInterBlk :660-815 (no BDOF/DMVR/RPR/wrap-around):
//   yFrac == 0 -> filterHor(isLast = rndRes); xFrac == 0 -> filterVer(first, isLast = rndRes);
//   else filterHor(first, !last) on rows -3..H+3 then filterVer(!first, isLast = rndRes);   rndRes = !bi.
// CommonLib/Buffer.cpp removeHighFreq :475-520 (org = 2*org - pred, unclipped: ClipForBiPredMEEnabled = 0),
// addAvg :467-507 (dst = clip((a + b + offset) >> shift) on 14-bit intermediates).
#include "ctx.hpp"
#include "mc_block.hpp"

namespace
{

struct StoreGlobal
{
  int16_t *dst; int stride;
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const { dst[( long ) y * stride + x] = v; }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const { store8g( dst + ( long ) y * stride + x0, v ); }
};

__global__ __launch_bounds__( 64 ) void mc_kernel( const int16_t *__restrict__ refBase, int16_t *__restrict__ dstBase,
                                                  const vtmhip_mc_job *__restrict__ jobs, int maxW, int maxH )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];   // [(h+7)][w] H-pass intermediates
  const vtmhip_mc_job j = jobs[blockIdx.x];
  const StoreGlobal   st{ dstBase + j.dstOff, j.dstStride };
  mc_any<64>( j, refBase, lds, ( int ) threadIdx.x, st );
}

// ---- InterPrediction::motionCompensation for one PU and one plane: xPredInterUni (:445-520) or xPredInterBi + xWeightedAverage
// (:527-660, 1354-1435; default weights -> PelBuf::addAvg) with the consumer of the prediction fused in: the residual
// org - pred (CodingStructure resi buffer, InterSearch.cpp:7260-7262) or the bi-pred ME target 2*org - pred (removeHighFreq). ----------
struct StoreLds
{
  int16_t *p; int w;
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const { p[y * w + x] = v; }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const { *reinterpret_cast<uint4 *>( p + y * w + x0 ) = pack8( v ); }
};
struct Epilogue
{
  const int16_t *org; int orgStride; int16_t *pred; int predStride; int16_t *out; int outStride; int mode;
  unsigned *sad;   // mode 3: this lane's running SAD of the prediction against the original (the AMVP template cost: nothing is stored)
  int w0, w1;      // mode 4: removeWeightHighFreq (Buffer.h:417-460): out = ( org * w0 - pred * w1 + 2^15 ) >> 16 -- the bi-pred search target under a CU-level BCW weight
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const
  {
    if( pred ) pred[( long ) y * predStride + x] = v;
    if( mode == 1 ) out[( long ) y * outStride + x] = ( int16_t ) ( org[( long ) y * orgStride + x] - v );
    else if( mode == 2 ) out[( long ) y * outStride + x] = ( int16_t ) ( 2 * org[( long ) y * orgStride + x] - v );
    else if( mode == 3 ) *sad += ( unsigned ) abs( ( int ) org[( long ) y * orgStride + x] - ( int ) v );
    else if( mode == 4 ) out[( long ) y * outStride + x] = ( int16_t ) ( ( ( int ) org[( long ) y * orgStride + x] * w0 - ( int ) v * w1 + ( 1 << 15 ) ) >> 16 );
  }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const
  {
    if( pred ) store8g( pred + ( long ) y * predStride + x0, v );
    if( mode == 3 )
    {
      int o[8];
      load8g( org + ( long ) y * orgStride + x0, o );
      unsigned t = 0;
#pragma unroll
      for( int k = 0; k < 8; k++ ) t += ( unsigned ) abs( o[k] - v[k] );
      *sad += t;
    }
    else if( mode )
    {
      int o[8], r[8];
      load8g( org + ( long ) y * orgStride + x0, o );
#pragma unroll
      for( int k = 0; k < 8; k++ ) r[k] = mode == 4 ? ( int ) ( int16_t ) ( ( o[k] * w0 - v[k] * w1 + ( 1 << 15 ) ) >> 16 ) : ( mode == 1 ? o[k] : 2 * o[k] ) - v[k];
      store8g( out + ( long ) y * outStride + x0, r );
    }
  }
};
struct AvgThen
{
  const int16_t *p0; int w; int shift, offset, cmax; Epilogue ep;
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const
  {
    ep( y, x, ( int16_t ) min( cmax, max( 0, ( ( int ) p0[y * w + x] + ( int ) v + offset ) >> shift ) ) );   // addAvg (Buffer.cpp:467-507)
  }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const
  {
    int a[8], r[8];
    load8s( p0 + y * w + x0, a );
#pragma unroll
    for( int k = 0; k < 8; k++ ) r[k] = min( cmax, max( 0, ( a[k] + v[k] + offset ) >> shift ) );
    ep.vec( y, x0, r );
  }
};

// THREADS lanes work on one block: 256 = a workgroup of four waves, 64 = a wave, 16 = a quarter of a wave (blocks of up to 64 samples: an 8x8 block has eight
// 8-sample segments per row pass, so a whole wave per block leaves 49 of 64 lanes idle and makes the launch a queue of 129 600 one-block waves whose
// time is the latency chain job -> window -> LDS -> store; four blocks per wave quarter the number of waves).  JPB blocks per workgroup (lanes of a block
// never span waves, so the H -> V hand-over needs no more than the wave-level fence).
// the block of one job; JobFn: jobIdx -> vtmhip_pred_job (a table entry, or built in registers from another table)
template<int THREADS, int JPB, class JobFn>
__device__ __forceinline__ void motion_comp_body( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase, int16_t *__restrict__ predBase,
                                                  int16_t *__restrict__ outBase, JobFn job, int n, int maxW, int maxH, unsigned long long *__restrict__ sadOut )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t lds[];
  __shared__ unsigned sSad;   // THREADS == 256 with sadOut: the block's SAD
  const int             sub = JPB > 1 ? ( int ) threadIdx.x / THREADS : 0;
  int16_t              *tmp = lds + sub * ( ( maxW * ( maxH + 7 ) + maxW * maxH + 7 ) & ~7 );   // [(h+7)][w] H-pass intermediates (16-byte aligned per block)
  int16_t              *p0  = tmp + maxW * ( maxH + 7 );                                         // [h][w] list-0 prediction of a bi-predicted block (14-bit)
  const int             jobIdx = xcd_order( ( int ) blockIdx.x, ( int ) gridDim.x ) * JPB + sub;   // neighbouring PUs (overlapping reference windows) on one XCD's L2
  if( jobIdx >= n ) return;
  const vtmhip_pred_job j    = job( jobIdx );
  const int             lane = JPB > 1 ? ( int ) threadIdx.x % THREADS : ( int ) threadIdx.x;
  if( j.route == 1 ) return;   // routed to vtmhip_bdof_batch_dev (a table both calls are launched over)
  Epilogue ep;
  ep.org = orgBase ? orgBase + j.orgOff : nullptr; ep.orgStride = j.orgStride;
  ep.pred = predBase ? predBase + j.predOff : nullptr; ep.predStride = j.predStride;
  ep.out = outBase ? outBase + j.outOff : nullptr; ep.outStride = j.outStride;
  ep.mode = sadOut ? 3 : ( outBase && orgBase ) ? j.epilogue : 0;
  ep.w0 = ep.w1 = 0;
  if( ep.mode == 2 && j.bcwWeight != 0 && j.bcwWeight != 4 )
  {
    const int bcw = j.bcwWeight, nrm = ( ( 1 << 16 ) + ( bcw > 0 ? ( bcw >> 1 ) : -( bcw >> 1 ) ) ) / bcw;
    ep.w0 = nrm << 3; ep.w1 = ( 8 - bcw ) * nrm; ep.mode = 4;
  }
  unsigned sadAcc = 0;
  ep.sad = &sadAcc;
  if( THREADS == 256 && sadOut ) { if( threadIdx.x == 0 ) sSad = 0; __syncthreads(); }
  vtmhip_mc_job m;
  m.width = j.width; m.height = j.height; m.bitDepth = j.bitDepth; m.useAltHpelIf = j.useAltHpelIf; m.chroma = j.chroma;
  m.dstOff = 0; m.dstStride = 0;
  if( j.mode != 2 )
  {
    const int l = j.mode;
    m.refOff = l ? j.refOff[1] : j.refOff[0]; m.refStride = l ? j.refStride[1] : j.refStride[0];   // (no dynamic indexing: keeps the job in registers)
    m.mvHor = l ? j.mv[1][0] : j.mv[0][0]; m.mvVer = l ? j.mv[1][1] : j.mv[0][1]; m.bi = 0;
    mc_any<THREADS>( m, refBase, tmp, lane, ep );
  }
  else
  {
    m.bi = 1;
    m.refOff = j.refOff[0]; m.refStride = j.refStride[0]; m.mvHor = j.mv[0][0]; m.mvVer = j.mv[0][1];
    const StoreLds s0{ p0, j.width };
    mc_any<THREADS>( m, refBase, tmp, lane, s0 );
    block_sync<THREADS>();   // p0 complete, tmp free again
    const int headRoom = max( 2, 14 - ( int ) j.bitDepth ), shift = headRoom + 1;
    const AvgThen av{ p0, j.width, shift, ( 1 << ( shift - 1 ) ) + 2 * 8192, ( 1 << j.bitDepth ) - 1, ep };
    m.refOff = j.refOff[1]; m.refStride = j.refStride[1]; m.mvHor = j.mv[1][0]; m.mvVer = j.mv[1][1];
    mc_any<THREADS>( m, refBase, tmp, lane, av );
  }
  if( sadOut )   // SAD of the whole block: over the block's lanes (a sub-wave group, a wave, or the workgroup)
  {
#pragma unroll
    for( int o = ( THREADS < 64 ? THREADS : 64 ) >> 1; o > 0; o >>= 1 ) sadAcc += __shfl_xor( sadAcc, o, 64 );
    if( THREADS == 256 )
    {
      if( ( threadIdx.x & 63 ) == 0 ) atomicAdd( &sSad, sadAcc );
      __syncthreads();
      if( threadIdx.x == 0 ) sadOut[jobIdx] = sSad;
    }
    else if( lane == 0 ) sadOut[jobIdx] = sadAcc;
  }
}

template<int THREADS, int JPB>
__global__ __launch_bounds__( THREADS * JPB ) void motion_comp_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase, int16_t *__restrict__ predBase,
                                                                 int16_t *__restrict__ outBase, const vtmhip_pred_job *__restrict__ jobs, int n, int maxW, int maxH,
                                                                 unsigned long long *__restrict__ sadOut )
{
  motion_comp_body<THREADS, JPB>( orgBase, refBase, predBase, outBase, [=]( int i ) { return jobs[i]; }, n, maxW, maxH, sadOut );
}

// xGetTemplateCost of the AMVP candidates (InterSearch.cpp:3235-3270): prediction job 2 * row + c = candidate c of ME job `row`, built in registers (no job table
// between the ME rows and the prediction: 112 bytes per candidate less to write and read); the block is reduced to its SAD against the original
template<int THREADS, int JPB>
__global__ __launch_bounds__( THREADS * JPB ) void motion_comp_amvp_kernel( vtmhip_pic_params pic, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                                      const vtmhip_me_job *__restrict__ rows, int n2, int maxW, int maxH,
                                                                      unsigned long long *__restrict__ sadOut )
{
  motion_comp_body<THREADS, JPB>( orgBase, refBase, nullptr, nullptr, [=]( int i )
  {
    const vtmhip_me_job &r = rows[i >> 1];
    const int c = ( i & 1 ) < r.numAmvpCand ? ( i & 1 ) : 0;   // a missing second candidate repeats the first (its cost is not looked at)
    int th = r.amvpCand[c][0], tv = r.amvpCand[c][1];
    th = min( ( pic.picW + 8 - r.puX - 1 ) << 4, max( ( -pic.ctuSize - 8 - r.puX + 1 ) << 4, th ) );   // clipMvInPic (Mv.cpp:56-74)
    tv = min( ( pic.picH + 8 - r.puY - 1 ) << 4, max( ( -pic.ctuSize - 8 - r.puY + 1 ) << 4, tv ) );
    vtmhip_pred_job p;
    p.orgOff = r.orgOff; p.refOff[0] = r.refOff; p.refOff[1] = r.refOff; p.predOff = 0; p.outOff = 0;
    p.orgStride = r.orgStride; p.refStride[0] = p.refStride[1] = r.refStride; p.predStride = r.width; p.outStride = r.width;
    p.mv[0][0] = th; p.mv[0][1] = tv; p.mv[1][0] = p.mv[1][1] = 0;
    p.width = r.width; p.height = r.height; p.mode = 0; p.epilogue = 0; p.bitDepth = ( uint8_t ) pic.bitDepth; p.useAltHpelIf = r.imv == 3; p.chroma = 0; p.route = 0; p.bcwWeight = 0;
    return p;
  }, n2, maxW, maxH, sadOut );
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
  const int              bcw = j.bcwWeight ? j.bcwWeight : 4;   // BCW ops: weight of 8 (4 = the default pair, which these ops are not called with)
  const int              nrm = ( ( 1 << 16 ) + ( bcw > 0 ? ( bcw >> 1 ) : -( bcw >> 1 ) ) ) / bcw, bw0 = nrm << 3, bw1 = ( 8 - bcw ) * nrm;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int y = i / w, x = i - y * w;
    const int av = a[( long ) y * j.aStride + x], bv = b[( long ) y * j.bStride + x];
    int       v;
    if( op == 0 ) v = ( int16_t ) ( 2 * av - bv );                        // removeHighFreq
    else if( op == 2 ) v = ( int16_t ) ( av - bv );                        // subtract (residual = org - pred, Buffer.cpp AreaBuf::subtract)
    else if( op == 3 ) v = ( int16_t ) ( ( av * bw0 - bv * bw1 + ( 1 << 15 ) ) >> 16 );   // removeWeightHighFreq (Buffer.h:417-460)
    else if( op == 4 ) v = min( cmax, max( 0, ( av * ( 8 - bcw ) + bv * bcw + ( ( 1 << ( shift + 1 ) ) + ( 8192 << 3 ) ) ) >> ( shift + 2 ) ) );   // addWeightedAvg
    else v = min( cmax, max( 0, ( av + bv + offset ) >> shift ) );         // addAvg
    d[( long ) y * j.dstStride + x] = ( int16_t ) v;
  }
}


// ---- BDOF: xPredInterBi with bioApplied (InterPrediction.cpp:527-660) for a bi-predicted luma PU.  One wave per region of at most 16 x 16 (the cut
// xSubPuBio makes, :414-417): both 14-bit predictions go to LDS inside a one-sample ring of nearest-integer reference samples (xPredInterBlk
// :733-810), gradients as gradFilterCore (Buffer.cpp:130-170), then per 4 x 4 unit the 6 x 6 sums of calcBIOSumsCore (:173-200), the clipped
// refinement (applyBiOptFlow :1296-1318) and addBIOAvgCore (:88-127).  Four lanes share a unit (9 window samples each, then one output row each).
struct StorePad   // prediction interior at (2,2) of a [h+4][BDOF_S] tile
{
  int16_t *p; int stride;
  __device__ __forceinline__ void operator()( int y, int x, int16_t v ) const { p[( y + 2 ) * stride + x + 2] = v; }
  __device__ __forceinline__ void vec( int y, int x0, const int v[8] ) const
  {
    unsigned *q = reinterpret_cast<unsigned *>( p + ( y + 2 ) * stride + x0 + 2 );   // even sample index: 4-byte aligned
    const uint4 u = pack8( v );
    q[0] = u.x; q[1] = u.y; q[2] = u.z; q[3] = u.w;
  }
};

constexpr int BDOF_S = 20, BDOF_G = 18;

struct BdofLds
{
  __attribute__( ( aligned( 16 ) ) ) int16_t tmp[16 * ( 16 + 7 )];   // H-pass intermediates of the 8-tap prediction
  __attribute__( ( aligned( 16 ) ) ) int16_t pred[2][BDOF_S * BDOF_S];   // 14-bit predictions, interior at (2,2), ring at (1,1)
  int16_t gx[2][BDOF_G * BDOF_G], gy[2][BDOF_G * BDOF_G];
};

// one list of a region: 8-tap prediction into the padded tile; with `bio` also the ring, the gradients and the border replication.
// m.refOff / refStride / mvHor / mvVer describe the list (refBase may be global memory or an LDS window).
__device__ __forceinline__ void bdof_list( BdofLds &L, int l, const vtmhip_mc_job &m, const int16_t *refBase, int lane, bool bio )
{
  const int dx = m.width, dy = m.height, headRoom = max( 2, 14 - ( int ) m.bitDepth );
  int16_t  *P = L.pred[l], *X = L.gx[l], *Y = L.gy[l];
  mc_any<64>( m, refBase, L.tmp, lane, StorePad{ P, BDOF_S } );
  if( !bio ) { block_sync<64>(); return; }
  // ring: the integer sample nearest to the fractional position, as a 14-bit intermediate (:768-803)
  const int16_t *src = refBase + m.refOff + ( long ) ( ( m.mvVer >> 4 ) + ( ( m.mvVer & 15 ) < 8 ? 0 : 1 ) ) * m.refStride + ( m.mvHor >> 4 ) + ( ( m.mvHor & 15 ) < 8 ? 0 : 1 );
  for( int i = lane; i < 2 * ( dx + 2 ) + 2 * dy; i += 64 )
  {
    int r, c;
    if( i < 2 * ( dx + 2 ) ) { r = i < dx + 2 ? -1 : dy; c = ( i < dx + 2 ? i : i - ( dx + 2 ) ) - 1; }
    else { const int t = i - 2 * ( dx + 2 ); r = t >> 1; c = ( t & 1 ) ? dx : -1; }
    P[( r + 2 ) * BDOF_S + c + 2] = ( int16_t ) ( ( ( int ) src[( long ) r * m.refStride + c] << headRoom ) - 8192 );
  }
  block_sync<64>();
  for( int i = lane; i < dx * dy; i += 64 )
  {
    const int      y = i / dx, x = i - y * dx;
    const int16_t *q = P + ( y + 2 ) * BDOF_S + x + 2;
    Y[( y + 1 ) * BDOF_G + x + 1] = ( int16_t ) ( ( q[BDOF_S] >> 6 ) - ( q[-BDOF_S] >> 6 ) );
    X[( y + 1 ) * BDOF_G + x + 1] = ( int16_t ) ( ( q[1] >> 6 ) - ( q[-1] >> 6 ) );
  }
  block_sync<64>();
  // replicate the borders: gradients (gradFilterCore PAD part), then the prediction's ring (applyBiOptFlow :1266-1276) -- columns first, rows after
  for( int y = lane; y < dy; y += 64 )
  {
    X[( y + 1 ) * BDOF_G] = X[( y + 1 ) * BDOF_G + 1]; X[( y + 1 ) * BDOF_G + dx + 1] = X[( y + 1 ) * BDOF_G + dx];
    Y[( y + 1 ) * BDOF_G] = Y[( y + 1 ) * BDOF_G + 1]; Y[( y + 1 ) * BDOF_G + dx + 1] = Y[( y + 1 ) * BDOF_G + dx];
    P[( y + 2 ) * BDOF_S + 1] = P[( y + 2 ) * BDOF_S + 2]; P[( y + 2 ) * BDOF_S + dx + 2] = P[( y + 2 ) * BDOF_S + dx + 1];
  }
  block_sync<64>();
  for( int x = lane; x < dx + 2; x += 64 )
  {
    X[x] = X[BDOF_G + x]; X[( dy + 1 ) * BDOF_G + x] = X[dy * BDOF_G + x];
    Y[x] = Y[BDOF_G + x]; Y[( dy + 1 ) * BDOF_G + x] = Y[dy * BDOF_G + x];
    P[BDOF_S + x + 1] = P[2 * BDOF_S + x + 1]; P[( dy + 2 ) * BDOF_S + x + 1] = P[( dy + 1 ) * BDOF_S + x + 1];
  }
  block_sync<64>();
}

// where the final samples of a region go: the prediction and / or the fused consumer (residual, 2*org - pred)
struct RegionOut
{
  const int16_t *org; int orgStride; int16_t *pred; int predStride; int16_t *out; int outStride; int mode;
  __device__ __forceinline__ void row4( int y, int x0, const int v[4] ) const
  {
    if( pred )
    {
      int16_t *d = pred + ( long ) y * predStride + x0;
#pragma unroll
      for( int x = 0; x < 4; x++ ) d[x] = ( int16_t ) v[x];
    }
    if( mode )
    {
      const int16_t *o = org + ( long ) y * orgStride + x0;
      int16_t       *d = out + ( long ) y * outStride + x0;
#pragma unroll
      for( int x = 0; x < 4; x++ ) d[x] = ( int16_t ) ( ( mode == 1 ? ( int ) o[x] : 2 * ( int ) o[x] ) - v[x] );
    }
  }
};

// both predictions are in L.pred: per 4 x 4 unit the 6 x 6 sums, the clipped refinement and the refined average (bio), or the plain addAvg.
// Four lanes share a unit; lane q writes row q.  (y, x) passed to `ro` are relative to the region.
__device__ __forceinline__ void bdof_units( const BdofLds &L, int dx, int dy, int bd, int lane, bool bio, const RegionOut &ro )
{
  const int  unitsX = dx >> 2, units = unitsX * ( dy >> 2 );
  const int  u = lane >> 2, q = lane & 3;
  const bool live = u < units;
  const int  yu = live ? u / unitsX : 0, xu = live ? u - yu * unitsX : 0;
  const int  headRoom = max( 2, 14 - bd ), shiftNum = headRoom + 1, offset = ( 1 << ( shiftNum - 1 ) ) + 2 * 8192, cmax = ( 1 << bd ) - 1;
  int        tmpx = 0, tmpy = 0;
  if( bio )
  {
    int sAbsGX = 0, sAbsGY = 0, sDIX = 0, sDIY = 0, sSign = 0;
    if( live )
    {
#pragma unroll
      for( int t = 0; t < 9; t++ )
      {
        const int k = q + 4 * t, wy = k / 6, wx = k - wy * 6;
        const int gi = ( yu * 4 + wy ) * BDOF_G + xu * 4 + wx, pi = ( yu * 4 + wy + 1 ) * BDOF_S + xu * 4 + wx + 1;
        const int tGX = ( ( int ) L.gx[0][gi] + ( int ) L.gx[1][gi] ) >> 1, tGY = ( ( int ) L.gy[0][gi] + ( int ) L.gy[1][gi] ) >> 1;
        const int tDI = ( ( int ) L.pred[1][pi] >> 4 ) - ( ( int ) L.pred[0][pi] >> 4 );
        sAbsGX += abs( tGX ); sAbsGY += abs( tGY );
        sDIX += tGX < 0 ? -tDI : tGX == 0 ? 0 : tDI;
        sDIY += tGY < 0 ? -tDI : tGY == 0 ? 0 : tDI;
        sSign += tGY < 0 ? -tGX : tGY == 0 ? 0 : tGX;
      }
    }
#pragma unroll
    for( int o = 1; o <= 2; o <<= 1 )
    {
      sAbsGX += __shfl_xor( sAbsGX, o, 64 ); sAbsGY += __shfl_xor( sAbsGY, o, 64 ); sDIX += __shfl_xor( sDIX, o, 64 );
      sDIY += __shfl_xor( sDIY, o, 64 ); sSign += __shfl_xor( sSign, o, 64 );
    }
    tmpx = sAbsGX == 0 ? 0 : ( sDIX * 4 ) >> ( 31 - __clz( sAbsGX ) );   // rightShiftMSB (:1606-1609)
    tmpx = min( 15, max( -15, tmpx ) );
    const int mains = sSign >> 12, secs = sSign & 4095;
    const int tmpData = ( tmpx * mains * 4096 + tmpx * secs ) >> 1;
    tmpy = sAbsGY == 0 ? 0 : ( sDIY * 4 - tmpData ) >> ( 31 - __clz( sAbsGY ) );
    tmpy = min( 15, max( -15, tmpy ) );
  }
  if( !live ) return;
  const int y = yu * 4 + q;
  int v[4];
#pragma unroll
  for( int x = 0; x < 4; x++ )
  {
    const int gi = ( y + 1 ) * BDOF_G + xu * 4 + x + 1, pi = ( y + 2 ) * BDOF_S + xu * 4 + x + 2;
    const int b  = bio ? tmpx * ( ( int ) L.gx[0][gi] - ( int ) L.gx[1][gi] ) + tmpy * ( ( int ) L.gy[0][gi] - ( int ) L.gy[1][gi] ) : 0;
    v[x] = min( cmax, max( 0, ( int ) ( int16_t ) ( ( ( int ) L.pred[0][pi] + ( int ) L.pred[1][pi] + b + offset ) >> shiftNum ) ) );
  }
  ro.row4( y, xu * 4, v );
}

__global__ __launch_bounds__( 64 ) void bdof_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase, int16_t *__restrict__ predBase,
                                                    int16_t *__restrict__ outBase, const vtmhip_pred_job *__restrict__ jobs )
{
  __shared__ BdofLds L;
  const vtmhip_pred_job j = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  if( j.route == 2 ) return;   // routed to vtmhip_motion_compensation_batch_dev
  const int dx = min( 16, ( int ) j.width ), dy = min( 16, ( int ) j.height ), perRow = j.width / dx;
  const int region = blockIdx.x;
  if( region >= perRow * ( j.height / dy ) ) return;
  const int ry = ( region / perRow ) * dy, rx = ( region % perRow ) * dx;
  vtmhip_mc_job m;
  m.width = ( int16_t ) dx; m.height = ( int16_t ) dy; m.bitDepth = j.bitDepth; m.useAltHpelIf = 0; m.chroma = 0; m.bi = 1;
  m.dstOff = 0; m.dstStride = 0;
#pragma unroll
  for( int l = 0; l < 2; l++ )
  {
    m.refOff = ( l ? j.refOff[1] : j.refOff[0] ) + ( long ) ry * ( l ? j.refStride[1] : j.refStride[0] ) + rx;
    m.refStride = l ? j.refStride[1] : j.refStride[0];
    m.mvHor = l ? j.mv[1][0] : j.mv[0][0]; m.mvVer = l ? j.mv[1][1] : j.mv[0][1];
    bdof_list( L, l, m, refBase, lane, true );
  }
  const int mode = ( outBase && orgBase ) ? j.epilogue : 0;
  const RegionOut ro{ orgBase ? orgBase + j.orgOff + ( long ) ry * j.orgStride + rx : nullptr, j.orgStride,
                      predBase ? predBase + j.predOff + ( long ) ry * j.predStride + rx : nullptr, j.predStride,
                      outBase ? outBase + j.outOff + ( long ) ry * j.outStride + rx : nullptr, j.outStride, mode };
  bdof_units( L, dx, dy, j.bitDepth, lane, true, ro );
}

// ---- DMVR: InterPrediction::xProcessDMVR (InterPrediction.cpp:1997-2195) for the luma plane of a bi-predicted PU.  One wave per sub-PU of at most
// 16 x 16: the (dx+7) x (dy+7) integer window of each list is fetched ONCE into LDS (xPrefetch :1666-1708) and everything else works from there:
// bilinear (dx+4) x (dy+4) predictions (xinitMC :1941-1995), the 25 row-sub-sampled SADs of the mirrored displacements (xDMVRCost, xBIPMVRefine
// :1819-1843; 2 lanes per displacement), the error surface (:1733-1817), the 2-sample replication of the window when the sub-PU moved (xPad
// :1709-1731), the 8-tap prediction out of the padded window (xFinalPaddedMCForDMVR :1845-1917) and the average, with BDOF when the caller's
// bioApplied holds and the matching cost is not below 2*dx*dy (:2139).
constexpr int DMVR_PS = 28, DMVR_BS = 20;

__device__ __forceinline__ void clip_mv_pic( int &hor, int &ver, const vtmhip_pic_params &pp, int x, int y )   // clipMvInPic, Mv.cpp:56-74
{
  hor = min( ( pp.picW + 8 - x - 1 ) << 4, max( ( -pp.ctuSize - 8 - x + 1 ) << 4, hor ) );
  ver = min( ( pp.picH + 8 - y - 1 ) << 4, max( ( -pp.ctuSize - 8 - y + 1 ) << 4, ver ) );
}

__device__ __forceinline__ int div_for_maxq7( long long N, long long D )   // :1733-1767
{
  int sign = 0, q = 0;
  if( N < 0 ) { sign = 1; N = -N; }
  D = D * 8;
  if( N >= D ) { N -= D; q++; }
  q = q * 2;
  D = D >> 1;
  if( N >= D ) { N -= D; q++; }
  q = q * 2;
  if( N >= ( D >> 1 ) ) q++;
  return sign ? -q : q;
}

__global__ __launch_bounds__( 64 ) void dmvr_kernel( vtmhip_pic_params pp, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                    int16_t *__restrict__ predBase, int16_t *__restrict__ outBase, const vtmhip_dmvr_job *__restrict__ jobs,
                                                    int32_t *__restrict__ mvdOut )
{
  __shared__ BdofLds L;
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sPad[2][DMVR_PS * DMVR_PS];
  __shared__ int16_t  sBil[2][DMVR_BS * DMVR_BS];
  __shared__ unsigned sSad[32];
  const vtmhip_dmvr_job j = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const int dx = min( 16, ( int ) j.width ), dy = min( 16, ( int ) j.height ), perRow = j.width / dx;
  const int region = blockIdx.x;
  if( region >= perRow * ( j.height / dy ) ) return;
  const int ry = ( region / perRow ) * dy, rx = ( region % perRow ) * dx;
  const int x = j.puX + rx, y = j.puY + ry, bd = j.bitDepth;
  // xPrefetch + xinitMC
#pragma unroll
  for( int l = 0; l < 2; l++ )
  {
    const int  mvH = l ? j.mv[1][0] : j.mv[0][0], mvV = l ? j.mv[1][1] : j.mv[0][1];
    const long rs  = l ? j.refStride[1] : j.refStride[0];
    int ph = mvH - ( 3 << 4 ), pv = mvV - ( 3 << 4 );
    clip_mv_pic( ph, pv, pp, x, y );
    const int16_t *src = refBase + ( l ? j.refOff[1] : j.refOff[0] ) + ( long ) ( ry + ( pv >> 4 ) ) * rs + rx + ( ph >> 4 );
    for( int i = lane; i < ( dx + 7 ) * ( dy + 7 ); i += 64 )
    {
      const int r = i / ( dx + 7 ), c = i - r * ( dx + 7 );
      sPad[l][( r + 2 ) * DMVR_PS + c + 2] = src[r * rs + c];
    }
  }
  block_sync<64>();
#pragma unroll
  for( int l = 0; l < 2; l++ )
  {
    int mh = l ? j.mv[1][0] : j.mv[0][0], mv = l ? j.mv[1][1] : j.mv[0][1];
    clip_mv_pic( mh, mv, pp, x, y );
    const int xFrac = mh & 15, yFrac = mv & 15, bw = dx + 4, bh = dy + 4;
    const int sh1 = 4 - ( 10 - bd ), off1 = 1 << ( sh1 - 1 );   // biMCForDMVR, first pass (InterpolationFilter.cpp:603-614)
    const int16_t *b0 = sPad[l] + 3 * ( DMVR_PS + 1 );
    for( int i = lane; i < bw * bh; i += 64 )
    {
      const int      r = i / bw, c = i - r * bw;
      const int16_t *p = b0 + r * DMVR_PS + c;
      int v;
      if( xFrac == 0 && yFrac == 0 ) v = bd > 10 ? ( ( int ) p[0] + ( 1 << ( bd - 11 ) ) ) >> ( bd - 10 ) : ( int ) p[0] << ( 10 - bd );   // filterCopy, :417-447
      else if( yFrac == 0 ) v = ( ( 16 - xFrac ) * ( int ) p[0] + xFrac * ( int ) p[1] + off1 ) >> sh1;
      else if( xFrac == 0 ) v = ( ( 16 - yFrac ) * ( int ) p[0] + yFrac * ( int ) p[DMVR_PS] + off1 ) >> sh1;
      else
      {
        const int t0 = ( int16_t ) ( ( ( 16 - xFrac ) * ( int ) p[0] + xFrac * ( int ) p[1] + off1 ) >> sh1 );
        const int t1 = ( int16_t ) ( ( ( 16 - xFrac ) * ( int ) p[DMVR_PS] + xFrac * ( int ) p[DMVR_PS + 1] + off1 ) >> sh1 );
        v = ( ( 16 - yFrac ) * t0 + yFrac * t1 + 8 ) >> 4;
      }
      sBil[l][r * DMVR_BS + c] = ( int16_t ) v;
    }
  }
  block_sync<64>();
  // the 25 matching costs: lane = 2 * displacement + half; every other row (setDistParam subShift 1, then >> 1: RdCost.cpp:368-408)
  {
    const int o = lane >> 1, half = lane & 1;
    unsigned  acc = 0;
    if( o < 25 )
    {
      const int      ox = o % 5 - 2, oy = o / 5 - 2, rows = dy >> 2;   // even rows of this half
      const int16_t *a = sBil[0] + ( 2 + oy ) * DMVR_BS + 2 + ox, *b = sBil[1] + ( 2 - oy ) * DMVR_BS + 2 - ox;
      for( int t = 0; t < rows; t++ )
      {
        const int r = 2 * ( half * rows + t );
        for( int c = 0; c < dx; c++ ) acc += ( unsigned ) abs( ( int ) a[r * DMVR_BS + c] - ( int ) b[r * DMVR_BS + c] );
      }
    }
    acc += __shfl_xor( acc, 1, 64 );
    if( o < 25 && half == 0 ) sSad[o] = acc;
  }
  block_sync<64>();
  // decisions, computed by every lane alike (wave-uniform): xProcessDMVR :2110-2150
  unsigned minCost = sSad[12] - ( sSad[12] >> 2 );
  int      best = 12, notZero = 1, total0 = 0, total1 = 0;
  if( minCost < ( unsigned ) ( dx * dy ) ) notZero = 0;
  else
  {
    const unsigned centre = minCost;
    for( int i = 0; i < 25; i++ )
    {
      const unsigned s = i == 12 ? centre : sSad[i];
      if( s < minCost ) { minCost = s; best = i; }
    }
    total0 = ( best % 5 - 2 ) * 16; total1 = ( best / 5 - 2 ) * 16;
  }
  const bool bio = minCost < ( unsigned ) ( 2 * dx * dy ) ? false : j.bioApplied != 0;
  if( notZero && abs( total0 ) != 32 && abs( total1 ) != 32 )   // xDMVRSubPixelErrorSurface :1929-1947
  {
    const unsigned centre = sSad[12] - ( sSad[12] >> 2 );
    auto sad = [&]( int i ) -> long long { return ( long long ) ( i == 12 ? centre : sSad[i] ); };
    const long long s0 = sad( best ), sl = sad( best - 1 ), st = sad( best - 5 ), sr = sad( best + 1 ), sb = sad( best + 5 );
    const long long dh = sl + sr - 2 * s0, dv = st + sb - 2 * s0;
    if( dh != 0 ) total0 += ( sl != s0 && sr != s0 ) ? div_for_maxq7( ( sl - sr ) * 16, dh ) : ( sl == s0 ? -8 : 8 );
    if( dv != 0 ) total1 += ( st != s0 && sb != s0 ) ? div_for_maxq7( ( st - sb ) * 16, dv ) : ( st == s0 ? -8 : 8 );
  }
  if( mvdOut && lane == 0 )
  {
    int32_t *m = mvdOut + ( ( long ) blockIdx.y * gridDim.x + region ) * 2;
    m[0] = total0; m[1] = total1;
  }
  if( total0 != 0 || total1 != 0 )   // xPad (paddingCore, Buffer.cpp:340-364): columns, then rows
  {
    const int pw = dx + 7, ph = dy + 7;
    for( int i = lane; i < 2 * ph; i += 64 )
    {
      int16_t *p = sPad[i & 1] + 2 * ( DMVR_PS + 1 ) + ( i >> 1 ) * DMVR_PS;
      p[-1] = p[-2] = p[0];
      p[pw] = p[pw + 1] = p[pw - 1];
    }
    block_sync<64>();
    for( int i = lane; i < 2 * ( pw + 4 ); i += 64 )
    {
      int16_t *p = sPad[i & 1] + 2 * DMVR_PS + ( i >> 1 );
      p[-DMVR_PS] = p[-2 * DMVR_PS] = p[0];
      p[ph * DMVR_PS] = p[( ph + 1 ) * DMVR_PS] = p[( ph - 1 ) * DMVR_PS];
    }
    block_sync<64>();
  }
  // xFinalPaddedMCForDMVR + xWeightedAverage
  vtmhip_mc_job m;
  m.width = ( int16_t ) dx; m.height = ( int16_t ) dy; m.bitDepth = j.bitDepth; m.useAltHpelIf = 0; m.chroma = 0; m.bi = 1;
  m.dstOff = 0; m.dstStride = 0; m.refStride = DMVR_PS;
#pragma unroll
  for( int l = 0; l < 2; l++ )
  {
    const int mvH = l ? j.mv[1][0] : j.mv[0][0], mvV = l ? j.mv[1][1] : j.mv[0][1];
    int rh = min( ( 1 << 17 ) - 1, max( -( 1 << 17 ), mvH + ( l ? -total0 : total0 ) ) );   // clipToStorageBitDepth
    int rv = min( ( 1 << 17 ) - 1, max( -( 1 << 17 ), mvV + ( l ? -total1 : total1 ) ) );
    const int dX = ( rh >> 4 ) - ( mvH >> 4 ), dY = ( rv >> 4 ) - ( mvV >> 4 );
    clip_mv_pic( rh, rv, pp, x, y );
    m.refOff = 5 * ( DMVR_PS + 1 ) + dY * DMVR_PS + dX;
    m.mvHor = rh & 15; m.mvVer = rv & 15;
    bdof_list( L, l, m, l ? sPad[1] : sPad[0], lane, bio );
  }
  const int mode = ( outBase && orgBase ) ? j.epilogue : 0;
  const RegionOut ro{ orgBase ? orgBase + j.orgOff + ( long ) ry * j.orgStride + rx : nullptr, j.orgStride,
                      predBase ? predBase + j.predOff + ( long ) ry * j.predStride + rx : nullptr, j.predStride,
                      outBase ? outBase + j.outOff + ( long ) ry * j.outStride + rx : nullptr, j.outStride, mode };
  bdof_units( L, dx, dy, bd, lane, bio, ro );
}

// One 4:2:0 chroma plane of the same PUs (xProcessDMVR with chroma): a sub-PU that did not move is predicted from the reference pictures, a moved one
// out of its (w/2+3) x (h/2+3) window -- fetched with the MERGE vector (xPrefetch forLuma = 0), replicated by one sample (xPad) -- then addAvg.
constexpr int DMVR_PSC = 20;

__global__ __launch_bounds__( 64 ) void dmvr_chroma_kernel( vtmhip_pic_params pp, const int16_t *__restrict__ orgBase, const int16_t *__restrict__ refBase,
                                                           int16_t *__restrict__ predBase, int16_t *__restrict__ outBase,
                                                           const vtmhip_dmvr_job *__restrict__ jobs, const int32_t *__restrict__ mvdIn )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t tmp[8 * ( 8 + 3 )];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sP[2][64];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sPad[2][DMVR_PSC * DMVR_PSC];
  const vtmhip_dmvr_job j = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const int dx = min( 16, ( int ) j.width ), dy = min( 16, ( int ) j.height ), perRow = j.width / dx;
  const int region = blockIdx.x;
  if( region >= perRow * ( j.height / dy ) ) return;
  const int ry = ( region / perRow ) * dy, rx = ( region % perRow ) * dx;
  const int x = j.puX + rx, y = j.puY + ry, bd = j.bitDepth, cw = dx >> 1, ch = dy >> 1, ryc = ry >> 1, rxc = rx >> 1;
  const int32_t *mvd = mvdIn + ( ( long ) j.mvdRow * gridDim.x + region ) * 2;
  const int  md0 = mvd[0], md1 = mvd[1];
  const bool moved = md0 != 0 || md1 != 0;
  vtmhip_mc_job m;
  m.width = ( int16_t ) cw; m.height = ( int16_t ) ch; m.bitDepth = j.bitDepth; m.useAltHpelIf = 0; m.chroma = 1; m.bi = 1;
  m.dstOff = 0; m.dstStride = 0;
#pragma unroll
  for( int l = 0; l < 2; l++ )
  {
    const int  mvH = l ? j.mv[1][0] : j.mv[0][0], mvV = l ? j.mv[1][1] : j.mv[0][1];
    const long rs  = l ? j.refStride[1] : j.refStride[0];
    const long blk = ( l ? j.refOff[1] : j.refOff[0] ) + ( long ) ryc * rs + rxc;
    int rh = min( ( 1 << 17 ) - 1, max( -( 1 << 17 ), mvH + ( l ? -md0 : md0 ) ) );
    int rv = min( ( 1 << 17 ) - 1, max( -( 1 << 17 ), mvV + ( l ? -md1 : md1 ) ) );
    const int dX = ( rh >> 5 ) - ( mvH >> 5 ), dY = ( rv >> 5 ) - ( mvV >> 5 );
    clip_mv_pic( rh, rv, pp, x, y );
    const StoreLds st{ sP[l], cw };
    if( !moved )
    {
      m.refOff = blk; m.refStride = ( int ) rs; m.mvHor = rh; m.mvVer = rv;
      mc_any<64>( m, refBase, tmp, lane, st );
    }
    else
    {
      int ph = mvH - ( 1 << 5 ), pv = mvV - ( 1 << 5 );
      clip_mv_pic( ph, pv, pp, x, y );
      const int16_t *src = refBase + blk + ( long ) ( pv >> 5 ) * rs + ( ph >> 5 );
      const int      pw = cw + 3, phh = ch + 3;
      int16_t       *W = sPad[l];
      for( int i = lane; i < pw * phh; i += 64 )
      {
        const int r = i / pw, c = i - r * pw;
        W[( r + 2 ) * DMVR_PSC + c + 2] = src[r * rs + c];
      }
      block_sync<64>();
      for( int r = lane; r < phh; r += 64 )
      {
        int16_t *q = W + ( r + 2 ) * DMVR_PSC + 2;
        q[-1] = q[0]; q[pw] = q[pw - 1];
      }
      block_sync<64>();
      for( int c = lane; c < pw + 2; c += 64 )
      {
        int16_t *q = W + 2 * DMVR_PSC + 1 + c;
        q[-DMVR_PSC] = q[0]; q[phh * DMVR_PSC] = q[( phh - 1 ) * DMVR_PSC];
      }
      block_sync<64>();
      m.refOff = 3 * ( DMVR_PSC + 1 ) + dY * DMVR_PSC + dX; m.refStride = DMVR_PSC; m.mvHor = rh & 31; m.mvVer = rv & 31;
      mc_any<64>( m, W, tmp, lane, st );
    }
    block_sync<64>();
  }
  if( lane >= cw * ch ) return;
  const int r = lane / cw, c = lane - r * cw;
  const int headRoom = max( 2, 14 - bd ), shift = headRoom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * 8192;
  const int v = min( ( 1 << bd ) - 1, max( 0, ( ( int ) sP[0][lane] + ( int ) sP[1][lane] + offset ) >> shift ) );
  if( predBase ) predBase[j.predOff + ( long ) ( ryc + r ) * j.predStride + rxc + c] = ( int16_t ) v;
  const int mode = ( outBase && orgBase ) ? j.epilogue : 0;
  if( mode )
  {
    const int o = orgBase[j.orgOff + ( long ) ( ryc + r ) * j.orgStride + rxc + c];
    outBase[j.outOff + ( long ) ( ryc + r ) * j.outStride + rxc + c] = ( int16_t ) ( ( mode == 1 ? o : 2 * o ) - v );
  }
}

// InterpolationFilter::xWeightedGeoBlk (InterpolationFilter.cpp:902-957): one wave per blend, four output samples per lane and step when the width
// allows (8-byte loads of both predictions, the weights gathered one by one -- their walk may be mirrored or 2:1 sub-sampled)
__global__ __launch_bounds__( 256 ) void geo_blend_kernel( const int16_t *__restrict__ srcBase, int16_t *__restrict__ dstBase, const int16_t *__restrict__ wBase,
                                                          const vtmhip_geo_blend_job *__restrict__ jobs, int n, int shift, int offset, int cmin, int cmax )
{
  const int lane = threadIdx.x & 63;
  const int job  = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
  if( job >= n ) return;
  const vtmhip_geo_blend_job j = jobs[job];
  const int16_t *s0 = srcBase + j.src0Off, *s1 = srcBase + j.src1Off, *wt = wBase + j.weightOff;
  int16_t       *d  = dstBase + j.dstOff;
  const int      w = j.width, h = j.height, sx = j.stepX;
  const bool     vec = ( w & 3 ) == 0 && ( ( j.src0Off | j.src1Off | j.dstOff | j.src0Stride | j.src1Stride | j.dstStride ) & 3 ) == 0;
  if( vec )
  {
    const int qw = w >> 2;
    for( int it = lane; it < qw * h; it += 64 )
    {
      const int   y = it / qw, x = ( it - y * qw ) << 2;
      const uint2 a = *( const uint2 * ) ( s0 + ( long ) y * j.src0Stride + x ), b = *( const uint2 * ) ( s1 + ( long ) y * j.src1Stride + x );
      const int16_t *wp = wt + ( long ) y * j.weightStride + ( long ) x * sx;
      const int   av[4] = { ( int16_t ) a.x, ( int ) a.x >> 16, ( int16_t ) a.y, ( int ) a.y >> 16 };
      const int   bv[4] = { ( int16_t ) b.x, ( int ) b.x >> 16, ( int16_t ) b.y, ( int ) b.y >> 16 };
      int         o[4];
#pragma unroll
      for( int k = 0; k < 4; k++ )
      {
        const int wk = wp[k * sx];
        o[k] = min( cmax, max( cmin, ( wk * av[k] + ( 8 - wk ) * bv[k] + offset ) >> shift ) );
      }
      uint2 r;
      r.x = ( unsigned ) ( o[0] & 0xffff ) | ( ( unsigned ) o[1] << 16 );
      r.y = ( unsigned ) ( o[2] & 0xffff ) | ( ( unsigned ) o[3] << 16 );
      *( uint2 * ) ( d + ( long ) y * j.dstStride + x ) = r;
    }
  }
  else
  {
    for( int it = lane; it < w * h; it += 64 )
    {
      const int y = it / w, x = it - y * w;
      const int wk = wt[( long ) y * j.weightStride + ( long ) x * sx];
      const int v  = wk * ( int ) s0[( long ) y * j.src0Stride + x] + ( 8 - wk ) * ( int ) s1[( long ) y * j.src1Stride + x] + offset;
      d[( long ) y * j.dstStride + x] = ( int16_t ) min( cmax, max( cmin, v >> shift ) );
    }
  }
}

}   // namespace

// motionCompensation of a job table; sadOut != nullptr: nothing is stored, the SAD of every prediction against its original block goes to sadOut[job]
// (xEstimateMvPredAMVP's template cost: prediction and distortion in one pass)
int vtmhip_internal_mc_launch( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase, const vtmhip_pred_job *d_jobs,
                               int n, int maxWidth, int maxHeight, unsigned long long *d_sadOut )
{
  const size_t lds = ( ( ( size_t ) maxWidth * ( maxHeight + 7 ) + ( size_t ) maxWidth * maxHeight + 7 ) & ~( size_t ) 7 ) * sizeof( int16_t );
  // lanes per block by its number of 8-sample items (a lane filters 8 samples per step; the samples of a block are independent, only the H -> V hand-over syncs):
  // 8 lanes up to 64 samples (8 blocks per wave), 16 up to 16x16 (4 per wave), one wave up to 32x32, four waves above.  Measured on the bench picture (AMVP stage +
  // all predictions): 16 / 32 / 256 / 256 lanes 0.90 ms, 16 / 32 / 64 / 256: 0.83, 8 / 32 / 64 / 256: 0.72, 8 / 16 / 64 / 256: 0.70, 8 / 16 / 32 / 256: 0.71
  VTMHIP_TIME_KERNEL( ctx, "motion_comp_kernel" );
  if( maxWidth * maxHeight > 1024 )
  {
    if( lds > 64 * 1024 )
      VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( motion_comp_kernel<256, 1> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
    hipLaunchKernelGGL( ( motion_comp_kernel<256, 1> ), dim3( n ), dim3( 256 ), lds, ctx->stream, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, n, maxWidth, maxHeight,
                        d_sadOut );
  }
  else if( maxWidth * maxHeight > 256 )   // (see the threshold above) up to 32x32 samples: one wave per block -- the H -> V hand-over is a wave barrier, not a workgroup one
    hipLaunchKernelGGL( ( motion_comp_kernel<64, 1> ), dim3( n ), dim3( 64 ), lds, ctx->stream, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, n, maxWidth, maxHeight,
                        d_sadOut );
  else if( maxWidth * maxHeight > 64 )
    hipLaunchKernelGGL( ( motion_comp_kernel<16, 4> ), dim3( ( n + 3 ) / 4 ), dim3( 64 ), 4 * lds, ctx->stream, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, n, maxWidth,
                        maxHeight, d_sadOut );
  else
    hipLaunchKernelGGL( ( motion_comp_kernel<8, 8> ), dim3( ( n + 7 ) / 8 ), dim3( 64 ), 8 * lds, ctx->stream, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, n, maxWidth,
                        maxHeight, d_sadOut );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

// the AMVP form of the launch above: the prediction jobs are derived from the ME rows inside the kernel
int vtmhip_internal_mc_amvp_launch( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_me_job *d_rows, int n,
                                    int maxWidth, int maxHeight, unsigned long long *d_sadOut )
{
  const size_t lds = ( ( ( size_t ) maxWidth * ( maxHeight + 7 ) + ( size_t ) maxWidth * maxHeight + 7 ) & ~( size_t ) 7 ) * sizeof( int16_t );
  const int    n2  = 2 * n;
  VTMHIP_TIME_KERNEL( ctx, "motion_comp_kernel" );
  if( maxWidth * maxHeight > 1024 )
  {
    if( lds > 64 * 1024 )
      VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( motion_comp_amvp_kernel<256, 1> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
    hipLaunchKernelGGL( ( motion_comp_amvp_kernel<256, 1> ), dim3( n2 ), dim3( 256 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_rows, n2, maxWidth, maxHeight, d_sadOut );
  }
  else if( maxWidth * maxHeight > 256 )
    hipLaunchKernelGGL( ( motion_comp_amvp_kernel<64, 1> ), dim3( n2 ), dim3( 64 ), lds, ctx->stream, *pic, d_orgBase, d_refBase, d_rows, n2, maxWidth, maxHeight, d_sadOut );
  else if( maxWidth * maxHeight > 64 )
    hipLaunchKernelGGL( ( motion_comp_amvp_kernel<16, 4> ), dim3( ( n2 + 3 ) / 4 ), dim3( 64 ), 4 * lds, ctx->stream, *pic, d_orgBase, d_refBase, d_rows, n2, maxWidth, maxHeight,
                        d_sadOut );
  else
    hipLaunchKernelGGL( ( motion_comp_amvp_kernel<8, 8> ), dim3( ( n2 + 7 ) / 8 ), dim3( 64 ), 8 * lds, ctx->stream, *pic, d_orgBase, d_refBase, d_rows, n2, maxWidth, maxHeight,
                        d_sadOut );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

extern "C"
{

int vtmhip_mc_luma_batch_dev( vtmhip_ctx *ctx, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_mc_job *d_jobs, int n, int maxWidth,
                              int maxHeight )
{
  return vtmhip_mc_batch_dev( ctx, d_refBase, d_dstBase, d_jobs, n, maxWidth, maxHeight );
}

int vtmhip_mc_batch_dev( vtmhip_ctx *ctx, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_mc_job *d_jobs, int n, int maxWidth, int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_refBase && d_dstBase && d_jobs, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 2 && maxWidth <= 128 && maxHeight >= 2 && maxHeight <= 128, "maxWidth / maxHeight" );
  const size_t lds = ( size_t ) maxWidth * ( maxHeight + 7 ) * sizeof( int16_t );
  hipLaunchKernelGGL( mc_kernel, dim3( n ), dim3( 64 ), lds, ctx->stream, d_refBase, d_dstBase, d_jobs, maxWidth, maxHeight );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_motion_compensation_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase,
                                          const vtmhip_pred_job *d_jobs, int n, int maxWidth, int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_refBase && d_jobs && ( d_predBase || d_outBase ), "null pointer" );
  VTMHIP_REQUIRE( ctx, !d_outBase || d_orgBase, "an epilogue output needs the original plane" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 2 && maxWidth <= 128 && maxHeight >= 2 && maxHeight <= 128, "maxWidth / maxHeight" );
  return vtmhip_internal_mc_launch( ctx, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, n, maxWidth, maxHeight, nullptr );
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

int vtmhip_remove_weight_high_freq_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_predBase, int16_t *d_dstBase,
                                              const vtmhip_pelop_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_orgBase && d_predBase && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( pelop_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_orgBase, d_predBase, d_dstBase, d_jobs, 3 );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_add_weighted_avg_batch_dev( vtmhip_ctx *ctx, const int16_t *d_src0Base, const int16_t *d_src1Base, int16_t *d_dstBase,
                                       const vtmhip_pelop_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_src0Base && d_src1Base && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( pelop_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_src0Base, d_src1Base, d_dstBase, d_jobs, 4 );
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

int vtmhip_bdof_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase,
                           const vtmhip_pred_job *d_jobs, int n, int maxWidth, int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_refBase && d_jobs && ( d_predBase || d_outBase ), "null pointer" );
  VTMHIP_REQUIRE( ctx, !d_outBase || d_orgBase, "an epilogue output needs the original plane" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 8 && maxWidth <= 128 && maxHeight >= 8 && maxHeight <= 128, "maxWidth / maxHeight (BDOF needs 8 <= w, h <= 128)" );
  const int regions = ( ( maxWidth + 15 ) / 16 ) * ( ( maxHeight + 15 ) / 16 );
  VTMHIP_TIME_KERNEL( ctx, "bdof_kernel" );
  for( int at = 0; at < n; at += 65535 )   // grid.y carries the PU index: 65535 per launch
  {
    const int m = n - at < 65535 ? n - at : 65535;
    hipLaunchKernelGGL( bdof_kernel, dim3( regions, m ), dim3( 64 ), 0, ctx->stream, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs + at );
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_dmvr_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                           int16_t *d_outBase, const vtmhip_dmvr_job *d_jobs, int n, int maxWidth, int maxHeight, int32_t *d_mvd )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && d_refBase && d_jobs && ( d_predBase || d_outBase || d_mvd ), "null pointer" );
  VTMHIP_REQUIRE( ctx, !d_outBase || d_orgBase, "an epilogue output needs the original plane" );
  VTMHIP_REQUIRE( ctx, pic->bitDepth >= 8 && pic->bitDepth <= 12 && pic->picW > 0 && pic->picH > 0 && pic->ctuSize >= 16, "picture parameters" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 8 && maxWidth <= 128 && maxHeight >= 8 && maxHeight <= 128, "maxWidth / maxHeight (DMVR needs 8 <= w, h <= 128)" );
  VTMHIP_REQUIRE( ctx, n <= 65535, "at most 65535 PUs per launch" );
  const int regions = ( ( maxWidth + 15 ) / 16 ) * ( ( maxHeight + 15 ) / 16 );
  hipLaunchKernelGGL( dmvr_kernel, dim3( regions, n ), dim3( 64 ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, d_mvd );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_dmvr_chroma_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                                  int16_t *d_outBase, const vtmhip_dmvr_job *d_jobs, int n, int maxWidth, int maxHeight, const int32_t *d_mvd )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && d_refBase && d_jobs && d_mvd && ( d_predBase || d_outBase ), "null pointer" );
  VTMHIP_REQUIRE( ctx, !d_outBase || d_orgBase, "an epilogue output needs the original plane" );
  VTMHIP_REQUIRE( ctx, pic->bitDepth >= 8 && pic->bitDepth <= 12 && pic->picW > 0 && pic->picH > 0 && pic->ctuSize >= 16, "picture parameters" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 8 && maxWidth <= 128 && maxHeight >= 8 && maxHeight <= 128, "maxWidth / maxHeight (luma size of the PUs, 8..128)" );
  VTMHIP_REQUIRE( ctx, n <= 65535, "at most 65535 PUs per launch" );
  const int regions = ( ( maxWidth + 15 ) / 16 ) * ( ( maxHeight + 15 ) / 16 );
  hipLaunchKernelGGL( dmvr_chroma_kernel, dim3( regions, n ), dim3( 64 ), 0, ctx->stream, *pic, d_orgBase, d_refBase, d_predBase, d_outBase, d_jobs, d_mvd );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

static int geo_blend_params( vtmhip_ctx *ctx, int bitDepth, int clipMin, int clipMax, int *shift, int *offset )
{
  VTMHIP_REQUIRE( ctx, bitDepth >= 8 && bitDepth <= 12, "bitDepth" );
  VTMHIP_REQUIRE( ctx, clipMin >= 0 && clipMin <= clipMax && clipMax < ( 1 << bitDepth ), "clip range" );
  *shift  = ( 14 - bitDepth > 2 ? 14 - bitDepth : 2 ) + 3;
  *offset = ( 1 << ( *shift - 1 ) ) + ( 8192 << 3 );
  return VTMHIP_OK;
}

int vtmhip_weightedGeoBlk_batch_dev( vtmhip_ctx *ctx, const int16_t *d_srcBase, int16_t *d_dstBase, const int16_t *d_weightBase,
                                     const vtmhip_geo_blend_job *d_jobs, int n, int bitDepth, int clipMin, int clipMax )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_srcBase && d_dstBase && d_weightBase && d_jobs, "null pointer" );
  int shift, offset;
  int st = geo_blend_params( ctx, bitDepth, clipMin, clipMax, &shift, &offset );
  if( st ) return st;
  hipLaunchKernelGGL( geo_blend_kernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_srcBase, d_dstBase, d_weightBase, d_jobs, n, shift, offset, clipMin, clipMax );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_weightedGeoBlk( vtmhip_ctx *ctx, const int16_t *src0, int src0Stride, const int16_t *src1, int src1Stride, int16_t *dst, int dstStride, int width,
                           int height, const int16_t *weight, int stepX, int weightStride, int bitDepth, int clipMin, int clipMax )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, src0 && src1 && dst && weight, "null pointer" );
  VTMHIP_REQUIRE( ctx, width >= 1 && height >= 1 && width <= 128 && height <= 128, "block size must be 1..128" );
  VTMHIP_REQUIRE( ctx, stepX != 0 && stepX >= -2 && stepX <= 2, "stepX must be +-1 or +-2" );
  int shift, offset;
  int st = geo_blend_params( ctx, bitDepth, clipMin, clipMax, &shift, &offset );
  if( st ) return st;
  // span of the weight plane the walk touches
  long       lo = 0, hi = 0;
  const long corners[4] = { 0, ( long ) ( width - 1 ) * stepX, ( long ) ( height - 1 ) * weightStride, ( long ) ( height - 1 ) * weightStride + ( long ) ( width - 1 ) * stepX };
  for( long c : corners ) { lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
  const size_t wN = ( size_t ) ( hi - lo + 1 );
  VTMHIP_REQUIRE( ctx, wN <= ( size_t ) 1 << 22, "weight walk spans more than 4 M samples" );
  const size_t blk = ( size_t ) width * height * sizeof( int16_t );
  const size_t dstOff = ( 2 * blk + 63 ) & ~( size_t ) 63, wOff = ( dstOff + blk + 63 ) & ~( size_t ) 63, jobOff = ( wOff + wN * 2 + 63 ) & ~( size_t ) 63, total = jobOff + 64;
  st = vtmhip_internal_scratch( ctx, total );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  for( int y = 0; y < height; y++ )
  {
    memcpy( hp + ( size_t ) y * width * 2, src0 + ( ptrdiff_t ) y * src0Stride, ( size_t ) width * 2 );
    memcpy( hp + blk + ( size_t ) y * width * 2, src1 + ( ptrdiff_t ) y * src1Stride, ( size_t ) width * 2 );
  }
  memcpy( hp + wOff, weight + lo, wN * 2 );
  vtmhip_geo_blend_job j;
  memset( &j, 0, sizeof( j ) );
  j.src0Off = 0; j.src1Off = ( int64_t ) width * height; j.dstOff = ( int64_t ) ( dstOff / 2 ); j.weightOff = ( int64_t ) ( wOff / 2 ) - lo;
  j.src0Stride = j.src1Stride = j.dstStride = width; j.weightStride = weightStride;
  j.width = ( int16_t ) width; j.height = ( int16_t ) height; j.stepX = ( int16_t ) stepX;
  memcpy( hp + jobOff, &j, sizeof( j ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, total, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( geo_blend_kernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ( const int16_t * ) dp, ( int16_t * ) dp, ( const int16_t * ) dp,
                      ( const vtmhip_geo_blend_job * ) ( dp + jobOff ), 1, shift, offset, clipMin, clipMax );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + dstOff, dp + dstOff, blk, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  for( int y = 0; y < height; y++ ) memcpy( dst + ( ptrdiff_t ) y * dstStride, hp + dstOff + ( size_t ) y * width * 2, ( size_t ) width * 2 );
  return VTMHIP_OK;
}

}   // extern "C"

// =====================================================================================================================
// Affine motion estimation gradients (reference CommonLib/AffineGradientSearch.cpp:62-170; callers
// EncoderLib/InterSearch.cpp xAffineMotionEstimation :5340-5775): 3x3 Sobel derivatives with border replication and the
// normal-equation accumulation of the 4- / 6-parameter model (int64 sums; the fp64 Gaussian solve stays on the host).
// =====================================================================================================================
namespace
{

__global__ __launch_bounds__( 256 ) void sobel_kernel( const int16_t *__restrict__ predBase, int *__restrict__ derivBase,
                                                      const vtmhip_affine_job *__restrict__ jobs, int vertical )
{
  const vtmhip_affine_job j = jobs[blockIdx.x];
  const int16_t          *p = predBase + j.predOff;
  int                    *d = derivBase + ( vertical ? j.derivVOff : j.derivHOff );
  const int               w = j.width, h = j.height, ps = j.predStride, ds = j.derivStride;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int y = i / w, x = i - y * w;
    // border samples take the value of the nearest interior position (corners: the diagonal neighbour), :77-92 / :111-126
    const int yc = min( h - 2, max( 1, y ) ), xc = min( w - 2, max( 1, x ) );
    const int16_t *c = p + ( long ) yc * ps + xc;
    const int v = vertical ? ( c[ps - 1] - c[-ps - 1] + ( c[ps] << 1 ) - ( c[-ps] << 1 ) + c[ps + 1] - c[-ps + 1] )
                           : ( c[1 - ps] - c[-1 - ps] + ( c[1] << 1 ) - ( c[-1] << 1 ) + c[1 + ps] - c[-1 + ps] );
    d[( long ) y * ds + x] = v;
  }
}

__global__ __launch_bounds__( 256 ) void equal_coeff_kernel( const int16_t *__restrict__ resiBase, const int *__restrict__ derivBase,
                                                            const vtmhip_affine_job *__restrict__ jobs, long long *__restrict__ eqBase )
{
  __shared__ long long     sAcc[4][42];
  const vtmhip_affine_job j = jobs[blockIdx.x];
  const int16_t           *r  = resiBase + j.resiOff;
  const int               *gx = derivBase + j.derivHOff, *gy = derivBase + j.derivVOff;
  const int                w = j.width, h = j.height, np = j.sixParam ? 6 : 4;
  long long                acc[42];   // [col][row 0..np] flattened with stride 7: col * 7 + row
#pragma unroll
  for( int i = 0; i < 42; i++ ) acc[i] = 0;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int y = i / w, x = i - y * w;
    const int cy = ( ( y >> 2 ) << 2 ) + 2, cx = ( ( x >> 2 ) << 2 ) + 2;
    const int a = gx[( long ) y * j.derivStride + x], b = gy[( long ) y * j.derivStride + x];
    const int e = r[( long ) y * j.resiStride + x];
    int       c[6];
    if( !j.sixParam ) { c[0] = a; c[1] = cx * a + cy * b; c[2] = b; c[3] = cy * a - cx * b; c[4] = 0; c[5] = 0; }
    else { c[0] = a; c[1] = cx * a; c[2] = b; c[3] = cx * b; c[4] = cy * a; c[5] = cy * b; }
#pragma unroll
    for( int col = 0; col < 6; col++ )
    {
      if( col < np )
      {
#pragma unroll
        for( int row = 0; row < 6; row++ )
          if( row < np ) acc[col * 7 + row] += ( long long ) c[col] * c[row];
        acc[col * 7 + 6] += ( ( long long ) c[col] * e ) << 3;   // slot 6 holds the right-hand side (written to column np below)
      }
    }
  }
#pragma unroll
  for( int i = 0; i < 42; i++ ) acc[i] = ( long long ) wave_reduce_add_u64( ( unsigned long long ) acc[i] );
  if( ( threadIdx.x & 63 ) == 0 )
  {
#pragma unroll
    for( int i = 0; i < 42; i++ ) sAcc[threadIdx.x >> 6][i] = acc[i];
  }
  __syncthreads();
  if( threadIdx.x < 42 )
  {
    const int       col = threadIdx.x / 7, row = threadIdx.x % 7;
    const long long v   = sAcc[0][threadIdx.x] + sAcc[1][threadIdx.x] + sAcc[2][threadIdx.x] + sAcc[3][threadIdx.x];
    long long      *eq  = eqBase + ( long ) blockIdx.x * 49;   // pEqualCoeff[7][7]; row 0 is unused by the reference
    if( col < np )
    {
      if( row < np ) eq[( col + 1 ) * 7 + row] += v;
      else if( row == 6 ) eq[( col + 1 ) * 7 + np] += v;
    }
  }
}

}   // namespace

extern "C"
{

int vtmhip_affine_sobel_batch_dev( vtmhip_ctx *ctx, const int16_t *d_predBase, int32_t *d_derivBase, const vtmhip_affine_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_predBase && d_derivBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( sobel_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_predBase, d_derivBase, d_jobs, 0 );
  hipLaunchKernelGGL( sobel_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_predBase, d_derivBase, d_jobs, 1 );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_affine_equal_coeff_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const int32_t *d_derivBase, const vtmhip_affine_job *d_jobs, int n,
                                         int64_t *d_equalCoeff )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_derivBase && d_jobs && d_equalCoeff, "null pointer" );
  hipLaunchKernelGGL( equal_coeff_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_resiBase, d_derivBase, d_jobs, ( long long * ) d_equalCoeff );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"
