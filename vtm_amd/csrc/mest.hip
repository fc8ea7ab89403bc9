// mest.hip -- InterSearch::xMotionEstimation as one batched call (reference EncoderLib/InterSearch.cpp:3299-3494, with
// xPatternSearchIntRefine :4172-4282).  The heavy stages are the library's own search kernels (me.hip, interp.hip, dist.hip);
// this file is the device-side bookkeeping between them -- one thread per job builds the next stage's job table from the
// previous stage's results, so the whole function is a fixed sequence of launches on one stream with no host round trip:
//
//   pattern   copy of the PU's original block (uni) or 2*org - otherPred (bi, removeHighFreq :3320-3326) into a compact slot
//   prepare   tz job (uni) / start-candidate SAD jobs (bi: rcMv + de-duplicated m_uniMvList entries, :3377-3420)
//   [dist]    start-candidate SADs            -> bi_start: best start, exhaustive-search job (xSetSearchRange + xPatternSearch)
//   tz / full integer search
//   mid       fractional job (cu.imv 0 / HPEL) or the 9 x numCand SATD jobs of the AMVR refinement (cu.imv 1 / 2)
//   frac / [dist]
//   final     rate re-weighting (:3478-3484) or the AMVR selection loop (:4208-4262), fp64 exactly as the reference
#include "ctx.hpp"
#include "mest_glue.hpp"
#include "dist_block.hpp"
#include "bucket.hpp"

namespace
{

constexpr int REFINE_SLOTS = 18;   // 9 positions x 2 AMVP candidates

using namespace mg;      // the per-row arithmetic (mest_glue.hpp): shared with the fused prologues / epilogues of the search kernels

struct Work
{
  int16_t            *pattern;   // n slots of slotSamples
  vtmhip_tz_job      *tz;
  vtmhip_full_job    *full;
  vtmhip_me_result   *ires;
  vtmhip_frac_job    *frac;
  vtmhip_frac_result *fres;
  vtmhip_dist_job    *dist;      // n x REFINE_SLOTS (the start candidates use the first START_SLOTS of each row)
  unsigned long long *dout;
  long                slotSamples;
  int                 direct;    // 1: no pattern copies -- the searches read the job's own block (uni: orgOff / orgStride in the original plane; bi: the
                                 // caller-made 2*org - pred block at otherPredOff / otherPredStride); 0: compact slots filled by mest_pattern_kernel
  int                 hasTz, hasFull;   // 0: the batch has no uni / no bi job and that job table is not allocated
  int                 needDist;  // 0: no job of the batch uses the per-job distortion slots (their job records are not even initialised)
};

__device__ __forceinline__ long pat_off( const Work &wk, const vtmhip_me_job &j, int i ) { return wk.direct ? ( j.bi ? j.otherPredOff : j.orgOff ) : ( long ) i * wk.slotSamples; }
__device__ __forceinline__ int  pat_stride( const Work &wk, const vtmhip_me_job &j ) { return wk.direct ? ( j.bi ? j.otherPredStride : j.orgStride ) : ( int ) j.width; }

// one workgroup per job: the search pattern in a compact slot (stride = width)
__global__ __launch_bounds__( 256 ) void mest_pattern_kernel( const int16_t *__restrict__ orgBase, const int16_t *__restrict__ otherBase,
                                                             const vtmhip_me_job *__restrict__ jobs, Work wk )
{
  const vtmhip_me_job &j = jobs[blockIdx.x];
  const int16_t *o = orgBase + j.orgOff;
  int16_t       *d = wk.pattern + ( long ) blockIdx.x * wk.slotSamples;
  const int      w = j.width, h = j.height;
  if( j.bi )
  {
    const int16_t *p = otherBase + j.otherPredOff;
    const int      bcw = bcw_weight( j );
    if( bcw )      // removeWeightHighFreq (Buffer.h:417-460)
    {
      const int nrm = ( ( 1 << 16 ) + ( bcw > 0 ? ( bcw >> 1 ) : -( bcw >> 1 ) ) ) / bcw, w0 = nrm << 3, w1 = ( 8 - bcw ) * nrm;
      for( int i = threadIdx.x; i < w * h; i += 256 )
      {
        const int y = i / w, x = i - y * w;
        d[i] = ( int16_t ) ( ( ( int ) o[( long ) y * j.orgStride + x] * w0 - ( int ) p[( long ) y * j.otherPredStride + x] * w1 + ( 1 << 15 ) ) >> 16 );
      }
    }
    else
    for( int i = threadIdx.x; i < w * h; i += 256 )
    {
      const int y = i / w, x = i - y * w;
      d[i] = ( int16_t ) ( 2 * o[( long ) y * j.orgStride + x] - p[( long ) y * j.otherPredStride + x] );   // removeHighFreq, unclipped (Buffer.h:475-520)
    }
  }
  else
  {
    for( int i = threadIdx.x; i < w * h; i += 256 )
    {
      const int y = i / w, x = i - y * w;
      d[i] = o[( long ) y * j.orgStride + x];
    }
  }
}

__device__ __forceinline__ void sad_job( vtmhip_dist_job &d, long slot, int slotStride, const vtmhip_me_job &j, int intX, int intY, int ss, int kind )
{
  d.orgOff = slot; d.curOff = j.refOff + ( long ) intY * j.refStride + intX;
  d.orgStride = slotStride; d.curStride = j.refStride; d.width = j.width; d.height = j.height; d.subShift = ( int16_t ) ss; d.kind = ( int16_t ) kind;
}

__global__ __launch_bounds__( 256 ) void mest_prepare_kernel( vtmhip_pic_params pic, vtmhip_me_cfg cfg, const vtmhip_me_job *__restrict__ jobs, int n, Work wk )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  const vtmhip_me_job &j = jobs[i];
  if( j.bi ? !wk.hasFull : !wk.hasTz ) return;   // a job that breaks the caller's uniformBi promise is left out (its result is undefined), never written out of bounds
  const long     slot = pat_off( wk, j, i );
  const int      sst  = pat_stride( wk, j );
  const int      ss   = sub_shift( cfg, j.width, j.height );
  vtmhip_dist_job *d = wk.dist + ( long ) i * REFINE_SLOTS;
  if( wk.needDist ) for( int k = 0; k < REFINE_SLOTS; k++ ) d[k].width = 0;
  if( !j.bi )
  {
    if( wk.hasFull ) wk.full[i].width = 0;
    make_tz_job( cfg, j, slot, sst, wk.tz[i] );
  }
  else
  {
    if( wk.hasTz ) wk.tz[i].width = 0;
    wk.full[i].width = 0;   // filled by mest_bi_start_kernel
    if( wk.needDist )
    {
      // start candidates (:3377-3420): rcMv, then the distinct m_uniMvList entries, newest first -- slot k of the row's distortion jobs
      int k = 0;
      for( int e = -1; e < num_extra( j ); e++ )
      {
        if( e >= 0 && !extra_is_first( j, e ) ) continue;
        int th = e < 0 ? j.mvHor : j.extraStart[e][0], tv = e < 0 ? j.mvVer : j.extraStart[e][1];
        clip_mv( pic, j, th, tv );
        sad_job( d[k], slot, sst, j, prec_down( th, 4 ), prec_down( tv, 4 ), ss, VTMHIP_DIST_SAD );
        k++;
      }
    }
  }
}

__global__ __launch_bounds__( 256 ) void mest_bi_start_kernel( vtmhip_pic_params pic, vtmhip_me_cfg cfg, const vtmhip_me_job *__restrict__ jobs, int n, Work wk )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  const vtmhip_me_job &j = jobs[i];
  if( !j.bi ) return;
  const unsigned is = imv_shift( j.imv );
  const int      ph = prec_down( j.mvPredHor, 2 ), pv = prec_down( j.mvPredVer, 2 );
  unsigned long long best = 0;
  int                bestH = j.mvHor, bestV = j.mvVer;
  if( wk.needDist )      // needDist == 0: the caller promised empty m_uniMvList lists -- rcMv is the only candidate and wins whatever its cost
  {
    int k = 0;
    for( int e = -1; e < num_extra( j ); e++ )
    {
      if( e >= 0 && !extra_is_first( j, e ) ) continue;
      int th = e < 0 ? j.mvHor : j.extraStart[e][0], tv = e < 0 ? j.mvVer : j.extraStart[e][1];
      clip_mv( pic, j, th, tv );
      th = prec_down( th, 4 ); tv = prec_down( tv, 4 );
      const unsigned long long c = wk.dout[( long ) i * REFINE_SLOTS + k] + rate( j.motionLambda, mv_bits( th, tv, ph, pv, 2, is ) );
      if( k == 0 || c < best ) { best = c; if( e >= 0 ) { bestH = j.extraStart[e][0]; bestV = j.extraStart[e][1]; } }
      k++;
    }
  }
  make_full_job( cfg, j, pat_off( wk, j, i ), pat_stride( wk, j ), bestH, bestV, wk.full[i] );
}

// test vector of the AMVR refinement: position `pos`, AMVP candidate c (:4208-4215)
__device__ __forceinline__ void refine_test_mv( const vtmhip_me_job &j, int intX, int intY, int pos, int c, int &th, int &tv )
{
  const int px = pos == 0 ? 0 : ( pos <= 3 ? -1 : pos <= 5 ? 0 : 1 );
  const int py = pos == 0 ? 0 : ( pos <= 3 ? pos - 2 : pos == 4 ? -1 : pos == 5 ? 1 : pos - 7 );
  const int sh = amvr_shift( j.imv );
  const int bh = prec_down( ( intX << 4 ) - j.amvpCand[c][0], sh ) << sh;   // cBaseMvd, roundTransPrecInternal2Amvr
  const int bv = prec_down( ( intY << 4 ) - j.amvpCand[c][1], sh ) << sh;
  th = ( px << sh ) + bh + j.amvpCand[c][0];
  tv = ( py << sh ) + bv + j.amvpCand[c][1];
}

__global__ __launch_bounds__( 256 ) void mest_mid_kernel( vtmhip_pic_params pic, vtmhip_me_cfg cfg, const vtmhip_me_job *__restrict__ jobs, int n, Work wk )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  const vtmhip_me_job   &j = jobs[i];
  const vtmhip_me_result r = wk.ires[i];
  const long             slot = pat_off( wk, j, i );
  const int              sst  = pat_stride( wk, j );
  vtmhip_frac_job       &q = wk.frac[i];
  vtmhip_dist_job       *d = wk.dist + ( long ) i * REFINE_SLOTS;
  if( wk.needDist ) for( int k = 0; k < REFINE_SLOTS; k++ ) d[k].width = 0;
  if( j.imv == 0 || j.imv == 3 ) make_frac_job( cfg.useHadME, pic.bitDepth, j, r.mvX, r.mvY, slot, sst, q );
  else
  {
    q.width = 0;
    for( int pos = 0; pos < 9; pos++ )
      for( int c = 0; c < j.numAmvpCand && c < 2; c++ )
      {
        int th, tv;
        refine_test_mv( j, r.mvX, r.mvY, pos, c, th, tv );
        clip_mv( pic, j, th, tv );
        sad_job( d[pos * 2 + c], slot, sst, j, th >> 4, tv >> 4, 0, cfg.useHadME ? VTMHIP_DIST_SATD : VTMHIP_DIST_SAD );
      }
  }
}

__global__ __launch_bounds__( 256 ) void mest_final_kernel( vtmhip_me_cfg cfg, const vtmhip_me_job *__restrict__ jobs, int n, Work wk, vtmhip_me_out *__restrict__ out )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  const vtmhip_me_job   &j = jobs[i];
  const vtmhip_me_result r = wk.ires[i];
  const int              bcw = bcw_weight( j );
  const double           fWeight = j.bi ? ( bcw ? fabs( ( double ) bcw / 8.0 ) : 0.5 ) : 1.0;    // xGetMEDistortionWeight (:7666-7676): |getBcwWeight| / g_BcwWeightBase, 0.5 for BCW_DEFAULT
  const double           lam = j.motionLambda;
  vtmhip_me_out o;
  o.intX = r.mvX; o.intY = r.mvY; o.intDist = r.dist;
  unsigned bits = j.bits;
  if( j.imv == 0 || j.imv == 3 ) make_out_frac( j, r.mvX, r.mvY, r.dist, wk.fres[i], o );
  else
  {
    const int sh = amvr_shift( j.imv );
    bits -= j.mvpIdxBits[j.mvpIdx & 1];
    unsigned long long bestDist = ~0ull, satd = 0;
    int                bestH = r.mvX << 4, bestV = r.mvY << 4, bestBits = 0, bestIdx = j.mvpIdx;
    for( int pos = 0; pos < 9; pos++ )
    {
      int t0h = 0, t0v = 0;
      for( int c = 0; c < j.numAmvpCand && c < 2; c++ )
      {
        int th, tv;
        refine_test_mv( j, r.mvX, r.mvY, pos, c, th, tv );
        if( c == 0 ) { t0h = th; t0v = tv; }
        unsigned long long dist;
        if( c == 0 || th != t0h || tv != t0v ) dist = satd = ( unsigned long long ) ( ( double ) wk.dout[( long ) i * REFINE_SLOTS + pos * 2 + c] * fWeight );
        else dist = satd;
        const int mvBits = ( int ) j.mvpIdxBits[c]
                         + ( int ) mv_bits( prec_down( th, sh ), prec_down( tv, sh ), prec_down( j.amvpCand[c][0], sh ), prec_down( j.amvpCand[c][1], sh ), 0, 0 );
        dist += rate( lam, ( unsigned ) mvBits );
        if( dist < bestDist ) { bestDist = dist; bestH = th; bestV = tv; bestIdx = c; bestBits = mvBits; }
      }
    }
    o.mvHor = bestH; o.mvVer = bestV; o.mvpIdx = bestIdx; o.mvPredHor = j.amvpCand[bestIdx][0]; o.mvPredVer = j.amvpCand[bestIdx][1];
    bits += ( unsigned ) bestBits;
    o.bits = bits;
    o.cost = bestDist - rate( lam, ( unsigned ) bestBits ) + rate( lam, bits );
  }
  out[i] = o;
}

// FUSED integer refinement of the AMVR modes (xPatternSearchIntRefine :4172-4282; cu.imv 1 / 2): one wave per row forms the nine test vectors per AMVP candidate around the
// integer result, measures each (SATD or SAD of the pattern against the reference block at the clipped vector; a second candidate whose vector equals the first's re-uses its
// value, :4226-4239), adds the vector + index rate and keeps the first strict minimum -- what mest_mid_kernel (18 job records) + dist_batch_kernel + mest_final_kernel did in
// three launches.  The loop IS mest_final_kernel's, with the distortion computed where that kernel read it.
__global__ __launch_bounds__( 576 ) void mest_amvr_kernel( vtmhip_pic_params pic, vtmhip_me_cfg cfg, const int16_t *__restrict__ patBase, const int16_t *__restrict__ refBase,
                                                          const vtmhip_me_job *__restrict__ jobs, int n, const vtmhip_me_result *__restrict__ ires, int patIsOther,
                                                          vtmhip_me_out *__restrict__ out )
{
  // one workgroup of nine waves per row: wave `pos` measures the test vectors of refinement position `pos` (both AMVP candidates), then the row's first lane runs the
  // selection loop over the 18 values in the reference's order
  __shared__ unsigned long long sDist[18];
  const int lane = threadIdx.x & 63, pos = ( int ) ( threadIdx.x >> 6 );
  const int i    = blockIdx.x;
  const vtmhip_me_job   &j = jobs[i];
  const vtmhip_me_result r = ires[i];
  const int16_t *pat = patBase + ( patIsOther ? j.otherPredOff : j.orgOff );
  const int      ps  = patIsOther ? j.otherPredStride : j.orgStride;
  const int      kind = cfg.useHadME ? VTMHIP_DIST_SATD : VTMHIP_DIST_SAD;
  {
    int t0h = 0, t0v = 0;
    for( int c = 0; c < j.numAmvpCand && c < 2; c++ )
    {
      int th, tv;
      refine_test_mv( j, r.mvX, r.mvY, pos, c, th, tv );
      if( c == 0 ) { t0h = th; t0v = tv; }
      if( c == 0 || th != t0h || tv != t0v )      // (a second candidate with the first one's vector re-uses its value, :4226-4239)
      {
        int ch = th, cv = tv;
        clip_mv( pic, j, ch, cv );
        const unsigned long long d = wave_block_dist( kind, pat, ps, refBase + j.refOff + ( long ) ( cv >> 4 ) * j.refStride + ( ch >> 4 ), j.refStride, j.width, j.height, 0, lane );
        if( lane == 0 ) sDist[pos * 2 + c] = d;
      }
    }
  }
  __syncthreads();
  if( pos != 0 ) return;
  // the selection (:4208-4262): lane k = (position k / 2, candidate k & 1) prices its test vector; the first strict minimum in the reference's evaluation order = the
  // lexicographic minimum of (cost, k)
  const int      bcw = bcw_weight( j );
  const double   fWeight = j.bi ? ( bcw ? fabs( ( double ) bcw / 8.0 ) : 0.5 ) : 1.0, lam = j.motionLambda;
  const int      sh = amvr_shift( j.imv ), nc = min( ( int ) j.numAmvpCand, 2 );
  const int      p = lane >> 1, c = lane & 1;
  unsigned long long cost = ~0ull;
  int                th = 0, tv = 0, mvBits = 0;
  if( lane < 18 && c < nc )
  {
    int t0h, t0v;
    refine_test_mv( j, r.mvX, r.mvY, p, 0, t0h, t0v );
    refine_test_mv( j, r.mvX, r.mvY, p, c, th, tv );
    const bool own = c == 0 || th != t0h || tv != t0v;
    const unsigned long long d = ( unsigned long long ) ( ( double ) sDist[p * 2 + ( own ? c : 0 )] * fWeight );
    mvBits = ( int ) j.mvpIdxBits[c] + ( int ) mv_bits( prec_down( th, sh ), prec_down( tv, sh ), prec_down( j.amvpCand[c][0], sh ), prec_down( j.amvpCand[c][1], sh ), 0, 0 );
    cost = d + rate( lam, ( unsigned ) mvBits );
  }
  unsigned long long bc = cost;
  unsigned           bk = ( unsigned ) lane;
#pragma unroll
  for( int o = 16; o > 0; o >>= 1 )      // lanes 0 .. 31 hold the 18 candidates
  {
    const unsigned long long oc = __shfl_xor( bc, o, 64 );
    const unsigned           ok = __shfl_xor( bk, o, 64 );
    if( oc < bc || ( oc == bc && ok < bk ) ) { bc = oc; bk = ok; }
  }
  if( lane >= 32 || ( unsigned ) lane != bk ) return;      // the winner (among the first 32 lanes: the upper half reduced its own idle lanes) writes the row's record
  vtmhip_me_out o;
  o.intX = r.mvX; o.intY = r.mvY; o.intDist = r.dist;
  o.mvHor = th; o.mvVer = tv; o.mvpIdx = c; o.mvPredHor = j.amvpCand[c][0]; o.mvPredVer = j.amvpCand[c][1];
  const unsigned bits = j.bits - j.mvpIdxBits[j.mvpIdx & 1] + ( unsigned ) mvBits;
  o.bits = bits;
  o.cost = cost - rate( lam, ( unsigned ) mvBits ) + rate( lam, bits );
  out[i] = o;
}

size_t align_up( size_t v ) { return ( v + 255 ) & ~( size_t ) 255; }

struct MestClassOf
{
  __device__ int operator()( const vtmhip_me_job &j ) const { return shape_class( j.width, j.height ); }
};

// AMVP selection folded into the fused integer search (vtmhip_internal_mest): the template SADs of the rows' candidates, written by the launch before this call
struct MestAmvp { const unsigned long long *dout; unsigned long long *distBiP; int addIdxBits; };
int mest_run( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase, const int16_t *d_refBase,
              const int16_t *d_otherPredBase, const vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_me_out *d_results, const MestAmvp *amvp = nullptr );

// a mixed-shape batch: bucket the jobs by shape on the device and run the uniform chain of every non-empty shape class over its slice
int mest_bucketed( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase, const int16_t *d_refBase,
                   const int16_t *d_otherPredBase, const vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_me_out *d_results )
{
  BucketPlan plan;
  int st = bucket_begin<vtmhip_me_job, vtmhip_me_out>( ctx, d_jobs, n, MestClassOf(), plan );
  if( st ) return st;
  for( int c = 0; c < BUCKET_CLASSES; c++ )
  {
    if( !plan.count[c] ) continue;
    vtmhip_me_cfg sub = *cfg;
    int w = maxWidth, h = maxHeight;
    sub.uniformSquare = c < 15;
    if( sub.uniformSquare ) class_shape( c, w, h );
    st = mest_run( ctx, pic, &sub, d_orgBase, d_refBase, d_otherPredBase, ( const vtmhip_me_job * ) plan.d_jobs + plan.offset[c], plan.count[c], w, h,
                   ( vtmhip_me_out * ) plan.d_results + plan.offset[c] );
    if( st ) return st;
  }
  return bucket_finish<vtmhip_me_out>( ctx, plan, n, d_results );
}

}   // namespace

extern "C" int vtmhip_xMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase,
                                                   const int16_t *d_refBase, const int16_t *d_otherPredBase, const vtmhip_me_job *d_jobs, int n,
                                                   int maxWidth, int maxHeight, vtmhip_me_out *d_results )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && cfg && d_orgBase && d_refBase && d_jobs && d_results, "null pointer" );
  // mixed shapes, enough jobs to fill launches per class: bucket by shape (VTMHIP_MEST_BUCKET=0 keeps the one-wave-per-PU chain for the whole batch)
  static const bool bucket = !( getenv( "VTMHIP_MEST_BUCKET" ) && atoi( getenv( "VTMHIP_MEST_BUCKET" ) ) == 0 );
  if( bucket && !cfg->uniformSquare && n >= 64 && bucket_allowed( ctx ) )   // (not under hipGraph capture: the bucketing synchronises the stream once)
    return mest_bucketed( ctx, pic, cfg, d_orgBase, d_refBase, d_otherPredBase, d_jobs, n, maxWidth, maxHeight, d_results );
  return mest_run( ctx, pic, cfg, d_orgBase, d_refBase, d_otherPredBase, d_jobs, n, maxWidth, maxHeight, d_results );
}

namespace
{

int mest_run( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase, const int16_t *d_refBase,
              const int16_t *d_otherPredBase, const vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_me_out *d_results, const MestAmvp *amvp )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && cfg && d_orgBase && d_refBase && d_jobs && d_results, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxWidth <= 128 && maxHeight >= 4 && maxHeight <= 128, "maxWidth / maxHeight" );
  VTMHIP_REQUIRE( ctx, cfg->uniformImv >= -1 && cfg->uniformImv <= 3, "uniformImv" );
  VTMHIP_REQUIRE( ctx, cfg->bipredSearchRange >= 0 && cfg->bipredSearchRange <= 64, "bipredSearchRange" );
  VTMHIP_REQUIRE( ctx, cfg->uniformBi >= 0 && cfg->uniformBi <= 2, "uniformBi" );

  const int  uimv = cfg->uniformImv, ubi = cfg->uniformBi;
  const bool fracOnly = uimv == 0 || uimv == 3;                       // no AMVR refinement in the batch: its distortion slots are never used
  const bool allUni = ubi == 1, allBi = ubi == 2;
  const bool noStart = allBi && cfg->noUniMvList;                    // bi jobs without m_uniMvList candidates: the start is rcMv, no start SADs
  VTMHIP_REQUIRE( ctx, !cfg->biPatternGiven || d_otherPredBase, "biPatternGiven needs d_otherPredBase (the 2*org - pred blocks)" );

  // workspace layout (regions a uniform batch never touches are not allocated)
  Work   wk;
  // direct addressing: uniform uni batches read the original plane, uniform bi batches with a caller-made pattern read that; mixed batches copy
  wk.direct   = allUni || ( allBi && cfg->biPatternGiven );
  wk.needDist = !( fracOnly && ( allUni || noStart ) );
  wk.hasTz = !allBi; wk.hasFull = !allUni;
  size_t off = 0;
  const size_t slotSamples = ( size_t ) maxWidth * maxHeight;
  const size_t oPattern = off; off = align_up( off + ( wk.direct ? 0 : ( size_t ) n * slotSamples * sizeof( int16_t ) ) );
  const size_t oTz      = off; off = align_up( off + ( allBi ? 0 : ( size_t ) n * sizeof( vtmhip_tz_job ) ) );
  const size_t oFull    = off; off = align_up( off + ( allUni ? 0 : ( size_t ) n * sizeof( vtmhip_full_job ) ) );
  const size_t oIres    = off; off = align_up( off + ( size_t ) n * sizeof( vtmhip_me_result ) );
  const size_t oFrac    = off; off = align_up( off + ( size_t ) n * sizeof( vtmhip_frac_job ) );
  const size_t oFres    = off; off = align_up( off + ( size_t ) n * sizeof( vtmhip_frac_result ) );
  const size_t oDist    = off; off = align_up( off + ( wk.needDist ? ( size_t ) n * REFINE_SLOTS * sizeof( vtmhip_dist_job ) : 0 ) );
  const size_t oDout    = off; off = align_up( off + ( wk.needDist ? ( size_t ) n * REFINE_SLOTS * sizeof( unsigned long long ) : 0 ) );
  void *arena = nullptr;
  int st = vtmhip_internal_workspace( ctx, off + 256, &arena );
  if( st ) return st;
  char *base = ( char * ) arena;
  wk.pattern = ( int16_t * ) ( base + oPattern ); wk.tz = ( vtmhip_tz_job * ) ( base + oTz ); wk.full = ( vtmhip_full_job * ) ( base + oFull );
  wk.ires = ( vtmhip_me_result * ) ( base + oIres ); wk.frac = ( vtmhip_frac_job * ) ( base + oFrac ); wk.fres = ( vtmhip_frac_result * ) ( base + oFres );
  wk.dist = ( vtmhip_dist_job * ) ( base + oDist ); wk.dout = ( unsigned long long * ) ( base + oDout ); wk.slotSamples = ( long ) slotSamples;

  const dim3 perJob( ( n + 255 ) / 256 ), tpb( 256 );
  const int  big = maxWidth > maxHeight ? maxWidth : maxHeight;
  vtmhip_pic_params pTz = *pic, pFull = *pic;
  if( pic->wavesPerJob == 0 )   // tuning defaults by block size (waves that share one search; measured on 3840x2160 level batches)
  {
    pTz.wavesPerJob   = big > 64 ? 8 : big > 32 ? 2 : 1;
    pFull.wavesPerJob = big > 64 ? 16 : big > 32 ? 8 : big > 16 ? 4 : 1;
  }
  const int16_t *patBase = wk.direct ? ( allUni ? d_orgBase : d_otherPredBase ) : wk.pattern;

  // FUSED forms (round 4; VTMHIP_MEST_FUSE=0 keeps the stand-alone glue launches): uniform batches whose pattern is addressed directly and whose rows all take the fractional
  // refinement -- the integer search builds its job records from the rows in its prologue (with xEstimateMvPredAMVP's selection when `amvp` is given), the fractional search
  // builds its jobs from the rows + the integer results and writes the rows' final records: no mest_prepare / mest_mid / mest_final launch, no job tables in between
  static const bool fuseOn = !( getenv( "VTMHIP_MEST_FUSE" ) && atoi( getenv( "VTMHIP_MEST_FUSE" ) ) == 0 );
  const bool fuseAny  = fuseOn && wk.direct && uimv != -1;      // a uniform AMVR mode: the integer stage fuses either way, then the fractional or the integer refinement
  const bool fuseFrac = fuseAny && fracOnly;
  const bool fuseAmvr = fuseAny && !fracOnly;
  const bool fuseTz   = fuseAny && allUni;
  VTMHIP_REQUIRE( ctx, !amvp || fuseTz, "the folded AMVP selection needs the fused uni chain" );
  MeFuse fu; memset( &fu, 0, sizeof( fu ) );
  fu.me = const_cast<vtmhip_me_job *>( d_jobs ); fu.cfg = *cfg; fu.bitDepth = pic->bitDepth; fu.patIsOther = allUni ? 0 : 1;
  fu.ires = wk.ires; fu.out = d_results; fu.tzSpill = wk.tz;
  if( amvp ) { fu.amvpDout = amvp->dout; fu.distBiP = amvp->distBiP; fu.addIdxBits = amvp->addIdxBits; }

  if( !wk.direct ) hipLaunchKernelGGL( mest_pattern_kernel, dim3( n ), tpb, 0, ctx->stream, d_orgBase, d_otherPredBase ? d_otherPredBase : d_orgBase, d_jobs, wk );
  const bool fuseFull = fuseAny && allBi;      // bi rows whose target the caller made: the start choice and the job record ride in the exhaustive search's prologue
  if( !fuseTz && !fuseFull )
  {
    hipLaunchKernelGGL( mest_prepare_kernel, perJob, tpb, 0, ctx->stream, *pic, *cfg, d_jobs, n, wk );
    VTMHIP_LAUNCHED( ctx );
  }
  if( fuseFull )
  {
    FullFuse ff; ff.me = d_jobs; ff.subShiftMode13 = cfg->fastInterSearchMode13; ff.bipredSearchRange = cfg->bipredSearchRange; ff.patIsOther = 1; ff.noStart = noStart;
    st = vtmhip_internal_full_search( ctx, &pFull, patBase, d_refBase, nullptr, n, cfg->uniformSquare ? maxWidth : 0, cfg->uniformSquare ? maxHeight : 0, wk.ires, &ff );
    if( st ) return st;
  }
  else if( !allUni )
  {
    // bi-pred: start candidates -> best start -> exhaustive search
    if( !noStart )
    {
      st = vtmhip_dist_batch_dev( ctx, patBase, d_refBase, wk.dist, n * REFINE_SLOTS, ( uint64_t * ) wk.dout );
      if( st ) return st;
    }
    hipLaunchKernelGGL( mest_bi_start_kernel, perJob, tpb, 0, ctx->stream, *pic, *cfg, d_jobs, n, wk );
    VTMHIP_LAUNCHED( ctx );
    // the lane-per-candidate kernel needs every slot to be a real maxWidth x maxHeight job: uniform batches of bi jobs only (in a mixed batch the
    // slots of uni jobs are empty; the cooperative kernel skips them)
    if( allBi && cfg->uniformSquare && cfg->bipredSearchRange <= 4 )
      st = vtmhip_full_search_uniform_batch_dev( ctx, &pFull, patBase, d_refBase, wk.full, n, maxWidth, maxHeight, wk.ires );
    else
      st = vtmhip_full_search_batch_dev( ctx, &pFull, patBase, d_refBase, wk.full, n, wk.ires );
    if( st ) return st;
  }
  if( !allBi )
  {
    // uni: TZ search
    st = vtmhip_internal_tz_search( ctx, &pTz, patBase, d_refBase, wk.tz, n, wk.ires, fuseTz ? &fu : nullptr, cfg->uniformSquare ? maxWidth : 0, cfg->uniformSquare ? maxHeight : 0 );
    if( st ) return st;
  }
  if( fuseFrac )      // integer results -> fractional search -> the rows' final records, in one launch
    return vtmhip_internal_frac_search( ctx, patBase, d_refBase, nullptr, n, maxWidth, maxHeight, cfg->uniformSquare, wk.fres, &fu );
  if( fuseAmvr )      // integer results -> the nine-position integer refinement of the AMVR modes -> the rows' final records, in one launch
  {
    hipLaunchKernelGGL( mest_amvr_kernel, dim3( n ), dim3( 576 ), 0, ctx->stream, *pic, *cfg, patBase, d_refBase, d_jobs, n, wk.ires, allUni ? 0 : 1, d_results );
    VTMHIP_LAUNCHED( ctx );
    return VTMHIP_OK;
  }
  hipLaunchKernelGGL( mest_mid_kernel, perJob, tpb, 0, ctx->stream, *pic, *cfg, d_jobs, n, wk );
  VTMHIP_LAUNCHED( ctx );
  if( uimv == -1 || uimv == 0 || uimv == 3 )
  {
    st = vtmhip_frac_search_batch_dev( ctx, patBase, d_refBase, wk.frac, n, maxWidth, maxHeight, cfg->uniformSquare && uimv != -1, wk.fres );
    if( st ) return st;
  }
  if( uimv == -1 || uimv == 1 || uimv == 2 )
  {
    st = vtmhip_dist_batch_dev( ctx, patBase, d_refBase, wk.dist, n * REFINE_SLOTS, ( uint64_t * ) wk.dout );
    if( st ) return st;
  }
  hipLaunchKernelGGL( mest_final_kernel, perJob, tpb, 0, ctx->stream, *cfg, d_jobs, n, wk, d_results );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // namespace

// vtmhip_xMotionEstimation_batch_dev for the level / CU-level driver (driver.hip): uniform uni rows whose AMVP candidates' template SADs sit in d_amvpDout -- xEstimateMvPredAMVP's
// selection happens in the prologue of the fused integer search (the rows are written in place).  Falls back to the selection kernel + the ordinary call when the batch does not
// take the fused chain (AMVR rows, mixed batches, VTMHIP_MEST_FUSE=0).
int vtmhip_internal_mest_with_amvp( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase, const int16_t *d_refBase,
                                    vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, vtmhip_me_out *d_results, const unsigned long long *d_amvpDout,
                                    unsigned long long *d_distBiP, int addIdxBits )
{
  MestAmvp a = { d_amvpDout, d_distBiP, addIdxBits };
  return mest_run( ctx, pic, cfg, d_orgBase, d_refBase, nullptr, d_jobs, n, maxWidth, maxHeight, d_results, &a );
}
int vtmhip_internal_mest_fusable( const vtmhip_me_cfg *cfg )
{
  static const bool fuseOn = !( getenv( "VTMHIP_MEST_FUSE" ) && atoi( getenv( "VTMHIP_MEST_FUSE" ) ) == 0 );
  return fuseOn && cfg->uniformBi == 1 && cfg->uniformImv >= 0 && cfg->uniformImv <= 3;
}

