// driver.hip -- the level-order picture loop of predInterSearch + residual coding as ONE library call (vtmhip_pis_run_picture): the sequence of batched
// calls per partition level and the fork / join between the uni-search chain and the levels' remaining stages, issued natively (no interpreter between the
// ~130 launches of a picture).  Every step is a public entry point of this library; nothing here computes.
#include "ctx.hpp"

#include <vector>
#include <cstring>
#include <cstdlib>

namespace
{

struct EventPool   // the context's fork / join events: created on demand (on the context's device), reused by every picture
{
  std::vector<hipEvent_t> &ev;
  size_t                   used = 0;
  explicit EventPool( std::vector<hipEvent_t> &v ) : ev( v ) {}
  hipEvent_t get()
  {
    if( used == ev.size() )
    {
      hipEvent_t e = nullptr;
      if( hipEventCreateWithFlags( &e, hipEventDisableTiming ) != hipSuccess ) return nullptr;
      ev.push_back( e );
    }
    return ev[used++];
  }
};

int run_uni( vtmhip_ctx *ctx, const vtmhip_pis_level_run &L, const vtmhip_pis_buffers &b )
{
  const int n = L.pis.numPU, rows = ( L.pis.numRef[0] + L.pis.numRef[1] ) * n;
  int st = L.pis.candsGiven ? VTMHIP_OK : vtmhip_pis_stage( ctx, &L.pis, 0 );   // candsGiven: the rows carry the caller's real AMVP lists
  if( st ) return st;
  // every row group searched (no FastMEForGenBLowDelay copy, no given row) and every row a fractional-refinement uni search: xEstimateMvPredAMVP's selection rides in the prologue
  // of the fused integer search (mest.hip), so the level's uni chain is: template SADs -> TZ (-> raster -> resume) -> fractional search + final records -> stage 1
  bool allSearched = L.pis.givenRows == 0;
  for( int r = 0; r < L.pis.numRef[1] && allSearched; r++ )
    allSearched = !( L.pis.fastMEForGenBLowDelay && L.pis.list1FromList0[r] > 0 && L.pis.list1FromList0[r] <= L.pis.numRef[0] );
  if( allSearched && vtmhip_internal_mest_fusable( &L.cfgUni ) )
  {
    unsigned long long *dout = nullptr;
    st = vtmhip_internal_amvp_sads( ctx, &L.pic, b.org, b.dpb, L.pis.uniJobs, rows, L.width, L.height, &dout );
    if( st ) return st;
    st = vtmhip_internal_mest_with_amvp( ctx, &L.pic, &L.cfgUni, b.org, b.dpb, L.pis.uniJobs, rows, L.width, L.height, L.uniOut, dout, ( unsigned long long * ) L.pis.distBiP, 1 );
    if( st ) return st;
    return vtmhip_pis_stage( ctx, &L.pis, 1 );
  }
  st = vtmhip_xEstimateMvPredAMVP_batch_dev( ctx, &L.pic, b.org, b.dpb, L.pis.uniJobs, rows, L.width, L.height, 1, 1, L.pis.distBiP );   // (the index bits of the chosen predictor join the row's bits)
  if( st ) return st;
  // the searched rows: every (list, refIdx) group but the list-1 pictures that are list-0 pictures too (FastMEForGenBLowDelay copies those in stage 1) and the groups whose rows
  // are GIVEN (pis.givenRows: buffered uni vectors of a CU-level BCW weight, re-priced in stage 1); rows of one group are contiguous, so the searched rows are a few contiguous runs
  const int groups = L.pis.numRef[0] + L.pis.numRef[1];
  int first = 0, count = 0;
  for( int g = 0; g <= groups; g++ )
  {
    const int  r1 = g - L.pis.numRef[0];      // refIdx of a list-1 group
    const bool copied = g < groups && r1 >= 0 && L.pis.fastMEForGenBLowDelay && L.pis.list1FromList0[r1] > 0 && L.pis.list1FromList0[r1] <= L.pis.numRef[0];
    const bool searched = g < groups && !copied && !( ( L.pis.givenRows >> g ) & 1 );
    if( searched ) { if( !count ) first = g; count++; continue; }
    if( count )
    {
      st = vtmhip_xMotionEstimation_batch_dev( ctx, &L.pic, &L.cfgUni, b.org, b.dpb, nullptr, L.pis.uniJobs + ( size_t ) first * n, count * n, L.width, L.height, L.uniOut + ( size_t ) first * n );
      if( st ) return st;
    }
    count = 0;
  }
  return vtmhip_pis_stage( ctx, &L.pis, 1 );
}

// B slices: the other list's prediction, the bi refinement, (the SMVD block,) the uni / bi decision
int run_bi( vtmhip_ctx *ctx, const vtmhip_pis_level_run &L, const vtmhip_pis_buffers &b )
{
  const int n = L.pis.numPU, w = L.width, h = L.height;
  int       st = VTMHIP_OK;
  if( L.pis.numRef[1] > 0 && !L.pis.biRestricted )
  {
    st = vtmhip_pis_stage( ctx, &L.pis, 2 );
    if( st ) return st;
    st = vtmhip_motion_compensation_batch_dev( ctx, b.org, b.dpb, nullptr, b.orgBi, L.pis.predOther, n, w, h );
    if( st ) return st;
    st = vtmhip_xMotionEstimation_batch_dev( ctx, &L.picBi, &L.cfgBi, b.org, b.dpb, b.orgBi, L.pis.biJobs, L.pis.numRef[0] * n, w, h, L.biOut );
    if( st ) return st;
    st = vtmhip_pis_stage( ctx, &L.pis, 3 );
    if( st ) return st;
    if( L.pis.smvdJobs )   // the SMVD block sits between the bi refinement and the uni / bi decision
    {
      st = vtmhip_smvd_batch_dev( ctx, &L.pic, b.org, b.dpb, L.pis.smvdJobs, n, w, h, VTMHIP_SMVD_SEARCH | ( L.cfgUni.uniformSquare ? VTMHIP_SMVD_UNIFORM : 0 ) );
      if( st ) return st;
      st = vtmhip_pis_stage( ctx, &L.pis, 5 );
      if( st ) return st;
    }
  }
  return st;
}

int run_rest( vtmhip_ctx *ctx, const vtmhip_pis_level_run &L, const vtmhip_pis_buffers &b )
{
  const int n = L.pis.numPU, w = L.width, h = L.height;
  int       st = VTMHIP_OK;
  st = run_bi( ctx, L, b );
  if( st ) return st;
  st = vtmhip_motion_compensation_batch_dev( ctx, b.org, b.dpb, b.pred, b.resi, L.pis.predFinal, n, w, h );
  if( st ) return st;
  if( L.bdof )
  {
    st = vtmhip_bdof_batch_dev( ctx, b.org, b.dpb, b.pred, b.resi, L.pis.predFinal, n, w, h );
    if( st ) return st;
  }
  if( L.pis.predFinalC )
  {
    st = vtmhip_motion_compensation_batch_dev( ctx, b.org, b.dpb, b.predC, b.resiC, L.pis.predFinalC, 2 * n, w / 2, h / 2 );
    if( st ) return st;
  }
  if( L.pis.affJobs )
  {
    const int rows = ( L.pis.numRef[0] + L.pis.numRef[1] ) * n;
    st = vtmhip_pis_stage( ctx, &L.pis, 4 );
    if( st ) return st;
    st = vtmhip_internal_affine_me_launch( ctx, &L.pic, b.org, b.dpb, nullptr, L.pis.affJobs, rows, w, h, L.affOut, 0 );      // stage 4 writes 4-parameter jobs
    if( st ) return st;
  }
  // luma TU chains: runs of consecutive candidates with a real transform go to the uniform kernel in one launch, transform skip to its own kernel
  for( int k = 0; k < L.numCands; )
  {
    int run = 1;
    if( L.cand[k] != 1 )
      while( k + run < L.numCands && L.cand[k + run] != 1 ) run++;
    const long a = ( long ) k * L.numTU, m = ( long ) run * L.numTU;
    if( L.cand[k] == 1 ) st = vtmhip_tu_ts_chain_batch_dev( ctx, b.resi, L.tu + a, ( int ) m, L.tuW, L.tuH, L.qcoef, nullptr, L.tuRes + a );
    else st = vtmhip_tu_chain_batch_dev( ctx, b.resi, L.tu + a, ( int ) m, L.tuW, L.tuH, 1, L.qcoef, nullptr, L.tuRes + a );
    if( st ) return st;
    k += run;
  }
  if( L.mtsTest && !st )
    st = vtmhip_mts_select_batch_dev( ctx, L.tuRes, L.numTU, L.numCands, L.cand, L.tuW, L.tuH, L.pic.bitDepth, 15, L.mtsMaxCand, L.mtsTest );
  if( st ) return st;
  if( L.tuC ) st = vtmhip_tu_chain_batch_dev( ctx, b.resiC, L.tuC, 2 * L.numTUC, L.tuWC, L.tuHC, 1, L.qcoefC, nullptr, L.tuResC );
  return st;
}

}   // namespace

extern "C" int vtmhip_is_uniform_shape( int w, int h )
{
  if( w == h ) return w == 8 || w == 16 || w == 32 || w == 64 || w == 128;
  const int a = w > h ? w : h, b = w > h ? h : w;
  return ( a == 16 && b == 8 ) || ( a == 32 && ( b == 8 || b == 16 ) ) || ( a == 64 && ( b == 16 || b == 32 ) );
}

extern "C" int vtmhip_predInterSearch_batch_dev( vtmhip_ctx *ctx, const vtmhip_pis_level_run *L, const vtmhip_pis_buffers *buf )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, L && buf && buf->org && buf->dpb, "level / buffers" );
  VTMHIP_REQUIRE( ctx, L->pis.candsGiven, "vtmhip_predInterSearch_batch_dev takes the caller's AMVP lists (pis.candsGiven)" );
  VTMHIP_REQUIRE( ctx, L->pis.numRef[1] == 0 || L->pis.biRestricted || ( buf->orgBi && L->pis.numRef[0] == L->pis.numRef[1] ), "B slices: orgBi scratch and equal list sizes" );
  VTMHIP_REQUIRE( ctx, L->uniOut == L->pis.uniOut && ( L->pis.numRef[1] == 0 || L->pis.biRestricted || L->biOut == L->pis.biOut ), "uniOut / biOut of the run and of the level differ" );
  if( L->pis.numPU == 0 ) return VTMHIP_OK;
  int st = run_uni( ctx, *L, *buf );
  if( st ) return st;
  return run_bi( ctx, *L, *buf );
}

extern "C" int vtmhip_pis_run_picture( vtmhip_ctx *ctx, const vtmhip_pis_level_run *levels, int numLevels, const vtmhip_pis_buffers *buf, void *mainStream,
                                       void *const *sideStreams, int numSide )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, levels && buf && numLevels >= 0 && numSide >= 0 && ( numSide == 0 || sideStreams ), "levels / buffers / streams" );
  VTMHIP_REQUIRE( ctx, buf->org && buf->dpb && buf->pred && buf->resi, "null picture buffer" );
  EventPool pool( ctx->forkEvents );
  hipStream_t main = ( hipStream_t ) mainStream;
  int         st = VTMHIP_OK;
  ctx->stream = main;
  if( numSide == 0 )
  {
    for( int i = 0; i < numLevels && !st; i++ ) st = run_uni( ctx, levels[i], *buf );
    for( int i = 0; i < numLevels && !st; i++ ) st = run_rest( ctx, levels[i], *buf );
    return st;
  }
  // Issue order: the uni chains of ALL levels first (one dependent chain on `main`, an event behind every level), then every level's remaining stages on its side stream
  // behind that event, so that the host never holds back the dependent chain while it enqueues a level's ~20 side launches.  Measured neutral on one GPU (whole picture and a
  // 1/8 share alike): the chain's idle time between its 61 launches (profiles/r03_trace_sim8_timeline.txt: busy 1.76 ms of a 2.62 ms span) is dispatch latency of dependent
  // kernels that share the CUs with the side streams' kernels, not host enqueue time.
  // VTMHIP_ISSUE_ORDER=interleave (round 4 experiment): level i's remaining stages are enqueued right after level i + 1's uni chain instead of after
  // ALL uni chains, so that a side chain's first launch is in its queue as soon as the event it waits for can fire
  static const bool interleave = getenv( "VTMHIP_ISSUE_ORDER" ) && !strcmp( getenv( "VTMHIP_ISSUE_ORDER" ), "interleave" );
  std::vector<hipEvent_t> done( ( size_t ) numLevels, nullptr );
  auto rest = [&]( int i ) -> int
  {
    hipStream_t side = ( hipStream_t ) sideStreams[i % numSide];
    VTMHIP_HIP( ctx, hipStreamWaitEvent( side, done[i], 0 ) );
    ctx->stream = side;
    const int r = run_rest( ctx, levels[i], *buf );
    ctx->stream = main;
    return r;
  };
  ctx->stream = main;
  for( int i = 0; i < numLevels && !st; i++ )
  {
    st = run_uni( ctx, levels[i], *buf );
    if( st ) break;
    done[i] = pool.get();
    VTMHIP_REQUIRE( ctx, done[i], "hipEventCreate" );
    VTMHIP_HIP( ctx, hipEventRecord( done[i], main ) );
    if( interleave && i > 0 ) st = rest( i - 1 );
  }
  if( interleave ) { if( !st && numLevels > 0 ) st = rest( numLevels - 1 ); }
  else for( int i = 0; i < numLevels && !st; i++ ) st = rest( i );
  ctx->stream = main;
  for( int s = 0; s < numSide && s < numLevels; s++ )   // join
  {
    hipEvent_t e = pool.get();
    VTMHIP_REQUIRE( ctx, e, "hipEventCreate" );
    VTMHIP_HIP( ctx, hipEventRecord( e, ( hipStream_t ) sideStreams[s] ) );
    VTMHIP_HIP( ctx, hipStreamWaitEvent( main, e, 0 ) );
  }
  return st;
}
