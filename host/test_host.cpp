// test_host.cpp -- reads like a reference-side unit test: builds DistParams with RdCost::setDistParam, calls dp.distFunc(dp),
// InterpolationFilter::m_filterHor/Ver and the fastFwdTrans/fastInvTrans tables through the C++ host mirror, and checks
// every result against the CPU oracle (TEST INFRASTRUCTURE: links oracle/libvtmoracle.so as the checker only).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../oracle/vtm_oracle.h"
#include "vtmhip_host.hpp"

using namespace vtmhip;

int main()
{
  std::mt19937 rng( 12345 );
  int          fails = 0, checks = 0;
  try
  {
    RdCost rd;
    const int ws[] = { 4, 8, 12, 16, 24, 32, 48, 64, 128 }, hs[] = { 4, 8, 16, 32, 64, 128 };
    for( int w : ws )
      for( int h : hs )
      {
        std::vector<Pel> org( ( size_t ) ( w + 5 ) * h ), cur( ( size_t ) ( w + 9 ) * h );
        for( auto &v : org ) v = ( Pel ) ( ( int ) ( rng() % 3070 ) - 1023 );   // bi-pred target range
        for( auto &v : cur ) v = ( Pel ) ( rng() % 1024 );
        CPelBuf   o( org.data(), w + 5, w, h );
        DistParam dp;
        for( int mode = 0; mode <= 3; mode++ )
        {
          rd.setDistParam( dp, o, cur.data(), w + 9, 10, COMPONENT_Y, mode, 1, false );
          if( dp.subShift != vo_subshift_for_mode( w, h, mode ) ) { fails++; printf( "subShift %dx%d mode %d\n", w, h, mode ); }
          const Distortion d = dp.distFunc( dp );
          checks++;
          if( d != vo_sad( org.data(), w + 5, cur.data(), w + 9, w, h, dp.subShift ) ) { fails++; printf( "SAD %dx%d mode %d\n", w, h, mode ); }
        }
        rd.setDistParam( dp, o, cur.data(), w + 9, 10, COMPONENT_Y, 0, 1, true );
        checks++;
        if( dp.distFunc( dp ) != vo_satd( org.data(), w + 5, cur.data(), w + 9, w, h ) ) { fails++; printf( "HAD %dx%d\n", w, h ); }
        checks++;
        if( rd.getDistPart( o, CPelBuf( cur.data(), w + 9, w, h ), 10, COMPONENT_Y, DF_SSE ) != vo_sse( org.data(), w + 5, cur.data(), w + 9, w, h ) ) { fails++; printf( "SSE %dx%d\n", w, h ); }
        // chroma: the fp64 distortion weight of the slice (RdCost.cpp:448-451; set per slice from the chroma QP offset, EncSlice.cpp:580-600)
        const double cw = 0.5 + ( rng() % 4000 ) / 1000.0;
        rd.setDistortionWeight( COMPONENT_Cb, cw );
        checks++;
        if( rd.getDistPart( o, CPelBuf( cur.data(), w + 9, w, h ), 10, COMPONENT_Cb, DF_SSE ) != ( Distortion ) ( cw * vo_sse( org.data(), w + 5, cur.data(), w + 9, w, h ) ) ) { fails++; printf( "chroma SSE %dx%d\n", w, h ); }
      }
    // guards behave like the reference's trampoline contract
    {
      DistParam dp; Pel z[64] = {};
      rd.setDistParam( dp, CPelBuf( z, 8, 8, 8 ), z, 8, 10, COMPONENT_Y );
      dp.useMR = true;
      bool threw = false;
      try { dp.distFunc( dp ); } catch( const Exception & ) { threw = true; }
      checks++; if( !threw ) { fails++; printf( "useMR guard\n" ); }
    }
    // GEO masked SAD through the mask overload of setDistParam
    {
      const int M = 112;
      std::vector<Pel> plane( M * M ), org( 64 * 64 ), cur( 64 * 64 );
      for( auto &v : plane ) v = ( Pel ) ( rng() % 9 );
      for( auto &v : org ) v = ( Pel ) ( rng() % 1024 );
      for( auto &v : cur ) v = ( Pel ) ( rng() % 1024 );
      for( int k = 0; k < 24; k++ )
      {
        const int w = 8 << ( k % 4 ), h = 8 << ( ( k / 4 ) % 4 ), sx = ( k % 3 ) ? 1 : -1, rd2 = ( k % 2 ) ? 1 : -1;
        const int x0 = ( int ) ( rng() % ( M - w ) ) + ( sx < 0 ? w - 1 : 0 ), y0 = ( int ) ( rng() % ( M - h ) ) + ( rd2 < 0 ? h - 1 : 0 );
        DistParam dp;
        rd.setDistParam( dp, CPelBuf( org.data(), 64, w, h ), cur.data(), 64, plane.data() + y0 * M + x0, rd2 * M, sx, -sx * w, 10, COMPONENT_Y );
        checks++;
        if( dp.distFunc( dp ) != vo_sad_mask( org.data(), 64, cur.data(), 64, w, h, 0, plane.data() + y0 * M + x0, rd2 * M, sx, -sx * w ) ) { fails++; printf( "masked SAD %dx%d\n", w, h ); }
      }
    }
    // motion cost
    vo_mvcost_t mc = { 7.25, -13, 22, 2 };
    rd.setMotionLambda( 7.25 ); rd.setPredictor( -13, 22 ); rd.setCostScale( 2 );
    for( int i = 0; i < 2000; i++ )
    {
      const int x = ( int ) ( rng() % 800 ) - 400, y = ( int ) ( rng() % 800 ) - 400; const unsigned s = rng() % 3;
      checks++;
      if( rd.getCostOfVectorWithPredictor( x, y, s ) != vo_mv_cost( &mc, x, y, s ) ) { fails++; printf( "mvcost\n" ); }
    }
    // interpolation tables
    InterpolationFilter f;
    std::vector<Pel> src( 80 * 80 ), a( 64 * 64 ), b( 64 * 64 );
    for( auto &v : src ) v = ( Pel ) ( rng() % 1024 );
    const ClpRng clp = { 0, 1023, 10, 0 };
    for( int frac = 1; frac < 16; frac += 3 )
      for( int first = 0; first < 2; first++ )
        for( int last = 0; last < 2; last++ )
          for( int ver = 0; ver < 2; ver++ )
          {
            ( ver ? f.m_filterVer : f.m_filterHor )[0][first][last]( clp, src.data() + 8 * 80 + 8, 80, a.data(), 64, 64, 64, vo_luma_filter[frac], false );
            vo_if_filter( ver, 8, first, last, src.data() + 8 * 80 + 8, 80, b.data(), 64, 64, 64, vo_luma_filter[frac], 10, 0, 1023, 0 );
            checks++; if( a != b ) { fails++; printf( "filter frac %d %d%d ver %d\n", frac, first, last, ver ); }
          }
    f.m_filterCopy[1][0]( clp, src.data(), 80, a.data(), 64, 64, 64, false );
    vo_if_copy( 1, 0, src.data(), 80, b.data(), 64, 64, 64, 10, 0, 1023, 0 );
    checks++; if( a != b ) { fails++; printf( "copy\n" ); }
    // transform tables: nullptr slots and one call per slot
    for( int t = 0; t < 3; t++ )
      for( int si = 0; si < 6; si++ )
      {
        const int n = 2 << si;
        std::vector<TCoeff> s( n * 8 ), d1( n * 8 ), d2( n * 8 );
        for( auto &v : s ) v = ( int ) ( rng() % 2048 ) - 1024;
        const bool have = vo_fwd_trans( t, n, s.data(), d2.data(), 7, 8, 0, 0 ) == 0;
        checks++;
        if( ( fastFwdTrans( t, si ) != nullptr ) != have ) { fails++; printf( "table slot %d %d\n", t, si ); continue; }
        if( !have ) continue;
        fastFwdTrans( t, si )( s.data(), d1.data(), 7, 8, 0, 0 );
        if( d1 != d2 ) { fails++; printf( "fwd %d %d\n", t, n ); }
        fastInvTrans( t, si )( s.data(), d1.data(), 7, 8, 0, 0, -32768, 32767 );
        vo_inv_trans( t, n, s.data(), d2.data(), 7, 8, 0, 0, -32768, 32767 );
        checks++; if( d1 != d2 ) { fails++; printf( "inv %d %d\n", t, n ); }
      }
  }
  catch( const Exception &e )
  {
    printf( "EXCEPTION: %s\n", e.what() );
    return 2;
  }
  printf( "host mirror: %d checks, %d failures\n", checks, fails );
  return fails ? 1 : 0;
}
