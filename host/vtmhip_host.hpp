// vtmhip_host.hpp -- C++ host-side mirror of the reference's dispatch surface for the hot path, on top of the C ABI
// (include/vtmhip.h).  Same names, argument meaning and error behaviour as the reference classes it mirrors, so that the
// VTM-side patch of INTEGRATION.md is a mechanical substitution and host code reads like the reference's:
//
//   vtmhip::DistParam / FpDistFunc / RdCost        CommonLib/RdCost.h:60-105, 107-424; RdCost.cpp:125-455
//   vtmhip::InterpolationFilter                     CommonLib/InterpolationFilter.h:59-116; InterpolationFilter.cpp:749-891
//   vtmhip::fastFwdTrans / fastInvTrans tables      CommonLib/TrQuant.cpp:69-81, TrQuant.h:53-54
//
// Errors: the reference THROWs an Exception (TypeDef.h:1065-1081); here vtmhip::Exception (std::runtime_error) carries
// the C ABI status text.  What the device path does not cover (applyWeight, useMR, step != 1 -- the same guards as
// x86/RdCostX86.h:213,344,2157) throws "unsupported": inside VTM the trampoline falls back to the scalar function instead.
#pragma once
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>

#include "../include/vtmhip.h"

namespace vtmhip
{
typedef int16_t  Pel;
typedef int32_t  TCoeff;
typedef int16_t  TFilterCoeff;
typedef uint64_t Distortion;

struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };

inline vtmhip_ctx *&context() { static vtmhip_ctx *c = nullptr; return c; }
inline void         check( int st, const char *what )
{
  if( st != VTMHIP_OK ) throw Exception( std::string( what ) + ": " + vtmhip_status_string( st ) + " (" + ( context() ? vtmhip_last_error( context() ) : "no context" ) + ")" );
}
inline void initHIP( int device = 0 )
{
  if( !context() ) check( vtmhip_create( device, &context() ), "vtmhip_create" );
}

struct CPelBuf
{
  const Pel *buf = nullptr; int stride = 0; int width = 0, height = 0;
  CPelBuf() {}
  CPelBuf( const Pel *b, int s, int w, int h ) : buf( b ), stride( s ), width( w ), height( h ) {}
};

enum ComponentID { COMPONENT_Y = 0, COMPONENT_Cb = 1, COMPONENT_Cr = 2, MAX_NUM_COMPONENT = 3 };

// DFunc numbering of the slots this path uses (TypeDef.h:476-554): base + floorLog2(width), dedicated slots for 12/24/48
enum DFunc
{
  DF_SSE = 0, DF_SSE2, DF_SSE4, DF_SSE8, DF_SSE16, DF_SSE32, DF_SSE64, DF_SSE16N,
  DF_SAD, DF_SAD2, DF_SAD4, DF_SAD8, DF_SAD16, DF_SAD32, DF_SAD64, DF_SAD16N,
  DF_HAD, DF_HAD2, DF_HAD4, DF_HAD8, DF_HAD16, DF_HAD32, DF_HAD64, DF_HAD16N,
  DF_SAD12, DF_SAD24, DF_SAD48,
  DF_SAD_WITH_MASK,
  DF_TOTAL_FUNCTIONS
};

class DistParam;
typedef Distortion ( *FpDistFunc )( const DistParam & );

class DistParam   // RdCost.h:67-105
{
public:
  CPelBuf     org, cur;
  int         step = 1;
  FpDistFunc  distFunc = nullptr;
  int         bitDepth = 0;
  bool        useMR = false, applyWeight = false, isBiPred = false;
  ComponentID compID = MAX_NUM_COMPONENT;
  Distortion  maximumDistortionForEarlyExit = std::numeric_limits<Distortion>::max();
  int         subShift = 0;
  const Pel  *mask = nullptr;   // GEO merge estimation (RdCost.h:97-100)
  int         maskStride = 0, stepX = 0, maskStride2 = 0;
};

inline int floorLog2( unsigned v ) { int r = -1; while( v ) { v >>= 1; r++; } return r; }

class RdCost
{
  static FpDistFunc *table() { static FpDistFunc t[DF_TOTAL_FUNCTIONS] = {}; return t; }
  double m_motionLambda = 0;
  double m_distortionWeight[MAX_NUM_COMPONENT] = { 1.0, 1.0, 1.0 };   // RdCost.h:117 (only chroma is weighted)
  int    m_iCostScale   = 0;
  int    m_predHor = 0, m_predVer = 0;

  static void guard( const DistParam &p )
  {
    if( p.applyWeight || p.useMR || p.step != 1 ) throw Exception( "unsupported on the device path: applyWeight / useMR / step != 1 (keep the scalar function)" );
  }
  static Distortion xGetSAD( const DistParam &p )
  {
    guard( p ); uint64_t d = 0;
    check( vtmhip_xGetSAD( context(), p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, p.org.width, p.org.height, p.subShift, &d ), "xGetSAD" );
    return d;
  }
  static Distortion xGetHADs( const DistParam &p )
  {
    guard( p ); uint64_t d = 0;
    check( vtmhip_xGetHADs( context(), p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, p.org.width, p.org.height, &d ), "xGetHADs" );
    return d;
  }
  static Distortion xGetSSE( const DistParam &p )
  {
    guard( p ); uint64_t d = 0;
    check( vtmhip_xGetSSE( context(), p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, p.org.width, p.org.height, &d ), "xGetSSE" );
    return d;
  }

  static Distortion xGetSADwMask( const DistParam &p )   // RdCost.cpp:3513-3549
  {
    guard( p ); uint64_t d = 0;
    check( vtmhip_xGetSADwMask( context(), p.org.buf, p.org.stride, p.cur.buf, p.cur.stride, p.org.width, p.org.height, p.subShift, p.mask, p.maskStride, p.stepX,
                                p.maskStride2, &d ), "xGetSADwMask" );
    return d;
  }

public:
  RdCost() { init(); }
  static FpDistFunc distFuncAt( int i ) { return table()[i]; }

  void init()   // RdCost::init (RdCost.cpp:125-217) + the "initRdCostHIP" installer of INTEGRATION.md
  {
    initHIP();
    for( int i = DF_SSE; i <= DF_SSE16N; i++ ) table()[i] = xGetSSE;
    for( int i = DF_SAD; i <= DF_SAD16N; i++ ) table()[i] = xGetSAD;
    for( int i = DF_HAD; i <= DF_HAD16N; i++ ) table()[i] = xGetHADs;
    table()[DF_SAD12] = table()[DF_SAD24] = table()[DF_SAD48] = xGetSAD;
    table()[DF_SAD_WITH_MASK] = xGetSADwMask;
  }

  // RdCost::setDistParam( rcDP, org, piRefY, iRefStride, bitDepth, compID, subShiftMode, step, useHadamard ) (RdCost.cpp:238-324)
  void setDistParam( DistParam &rcDP, const CPelBuf &org, const Pel *piRefY, int iRefStride, int bitDepth, ComponentID compID, int subShiftMode = 0,
                     int step = 1, bool useHadamard = false )
  {
    rcDP.bitDepth = bitDepth; rcDP.compID = compID; rcDP.org = org;
    rcDP.cur = CPelBuf( piRefY, iRefStride, org.width, org.height );
    rcDP.step = step; rcDP.maximumDistortionForEarlyExit = std::numeric_limits<Distortion>::max();
    const bool p2 = ( org.width & ( org.width - 1 ) ) == 0;
    if( !useHadamard )
    {
      if( org.width == 12 ) rcDP.distFunc = table()[DF_SAD12];
      else if( org.width == 24 ) rcDP.distFunc = table()[DF_SAD24];
      else if( org.width == 48 ) rcDP.distFunc = table()[DF_SAD48];
      else rcDP.distFunc = table()[DF_SAD + ( p2 ? floorLog2( org.width ) : 0 )];
    }
    else rcDP.distFunc = table()[DF_HAD + ( p2 ? floorLog2( org.width ) : 0 )];
    rcDP.subShift = 0;
    const int h = org.height, w = org.width;
    if( subShiftMode == 1 )
    {
      if( h > 32 && ( h & 15 ) == 0 ) rcDP.subShift = 4;
      else if( h > 16 && ( h & 7 ) == 0 ) rcDP.subShift = 3;
      else if( h > 8 && ( h & 3 ) == 0 ) rcDP.subShift = 2;
      else if( ( h & 1 ) == 0 ) rcDP.subShift = 1;
    }
    else if( subShiftMode == 2 ) { if( h > 8 && w <= 64 ) rcDP.subShift = 1; }
    else if( subShiftMode == 3 ) { if( h > 8 ) rcDP.subShift = 1; }
  }
  // RdCost::setDistParam( rcDP, org, cur, bitDepth, compID, useHadamard ) (RdCost.cpp:326-366)
  void setDistParam( DistParam &rcDP, const CPelBuf &org, const CPelBuf &cur, int bitDepth, ComponentID compID, bool useHadamard = false )
  {
    setDistParam( rcDP, org, cur.buf, cur.stride, bitDepth, compID, 0, 1, useHadamard );
  }
  // RdCost::setDistParam( rcDP, org, piRefY, iRefStride, mask, maskStride, stepX, maskStride2, bitDepth, compID ) (RdCost.cpp:3488-3511)
  void setDistParam( DistParam &rcDP, const CPelBuf &org, const Pel *piRefY, int iRefStride, const Pel *mask, int maskStride, int stepX, int maskStride2,
                     int bitDepth, ComponentID compID )
  {
    rcDP.bitDepth = bitDepth; rcDP.compID = compID; rcDP.org = org;
    rcDP.cur = CPelBuf( piRefY, iRefStride, org.width, org.height );
    rcDP.mask = mask; rcDP.maskStride = maskStride; rcDP.stepX = stepX; rcDP.maskStride2 = maskStride2;
    rcDP.step = 1; rcDP.subShift = 0; rcDP.maximumDistortionForEarlyExit = std::numeric_limits<Distortion>::max();
    rcDP.distFunc = table()[DF_SAD_WITH_MASK];
  }
  // RdCost::getDistPart (RdCost.cpp:411-455): a chroma distortion is scaled by m_distortionWeight[compID] in fp64 and truncated (:448-451)
  void setDistortionWeight( ComponentID compID, double w ) { m_distortionWeight[compID] = w; }   // RdCost.h:151
  Distortion getDistPart( const CPelBuf &org, const CPelBuf &cur, int bitDepth, ComponentID compID, DFunc eDFunc )
  {
    DistParam dp;
    dp.org = org; dp.cur = cur; dp.step = 1; dp.bitDepth = bitDepth; dp.compID = compID;
    const bool p2 = ( org.width & ( org.width - 1 ) ) == 0;
    dp.distFunc = table()[eDFunc + ( p2 ? floorLog2( org.width ) : 0 )];
    if( compID != COMPONENT_Y ) return ( Distortion ) ( m_distortionWeight[compID] * dp.distFunc( dp ) );
    return dp.distFunc( dp );
  }
  // motion cost (RdCost.h:186-190, 301-315)
  void     setPredictor( int hor, int ver ) { m_predHor = hor; m_predVer = ver; }
  void     setCostScale( int s ) { m_iCostScale = s; }
  void     setMotionLambda( double l ) { m_motionLambda = l; }
  static unsigned xGetExpGolombNumberOfBits( int iVal )
  {
    unsigned len = 1, t = ( iVal <= 0 ) ? ( unsigned( -iVal ) << 1 ) + 1 : unsigned( iVal << 1 );
    while( t > 128 ) { len += 14; t >>= 7; }
    return len + ( unsigned( floorLog2( t ) ) << 1 );
  }
  unsigned getBitsOfVectorWithPredictor( int x, int y, unsigned imvShift ) const
  {
    return xGetExpGolombNumberOfBits( ( ( x << m_iCostScale ) - m_predHor ) >> imvShift ) + xGetExpGolombNumberOfBits( ( ( y << m_iCostScale ) - m_predVer ) >> imvShift );
  }
  Distortion getCostOfVectorWithPredictor( int x, int y, unsigned imvShift ) const { return Distortion( m_motionLambda * getBitsOfVectorWithPredictor( x, y, imvShift ) ); }
};

struct ClpRng { int min, max, bd, n; };

class InterpolationFilter   // pointer tables m_filterHor / m_filterVer [tapIdx][isFirst][isLast], m_filterCopy[isFirst][isLast]
{
  template<int N, bool VER, bool FIRST, bool LAST>
  static void xFilter( const ClpRng &c, Pel const *src, int ss, Pel *dst, int ds, int w, int h, TFilterCoeff const *coeff, bool biMC )
  {
    check( ( VER ? vtmhip_filterVer : vtmhip_filterHor )( context(), N, FIRST, LAST, src, ss, dst, ds, w, h, coeff, c.bd, c.min, c.max, biMC ), "filter" );
  }
  template<bool FIRST, bool LAST>
  static void xCopy( const ClpRng &c, Pel const *src, int ss, Pel *dst, int ds, int w, int h, bool biMC )
  {
    check( vtmhip_filterCopy( context(), FIRST, LAST, src, ss, dst, ds, w, h, c.bd, c.min, c.max, biMC ), "filterCopy" );
  }
  template<int N, bool VER> void fill( void ( *t[2][2] )( const ClpRng &, Pel const *, int, Pel *, int, int, int, TFilterCoeff const *, bool ) )
  {
    t[0][0] = xFilter<N, VER, false, false>; t[0][1] = xFilter<N, VER, false, true>; t[1][0] = xFilter<N, VER, true, false>; t[1][1] = xFilter<N, VER, true, true>;
  }

public:
  void ( *m_filterHor[3][2][2] )( const ClpRng &, Pel const *, int, Pel *, int, int, int, TFilterCoeff const *, bool );
  void ( *m_filterVer[3][2][2] )( const ClpRng &, Pel const *, int, Pel *, int, int, int, TFilterCoeff const *, bool );
  void ( *m_filterCopy[2][2] )( const ClpRng &, Pel const *, int, Pel *, int, int, int, bool );
  InterpolationFilter()
  {
    initHIP();
    fill<8, false>( m_filterHor[0] ); fill<4, false>( m_filterHor[1] ); fill<2, false>( m_filterHor[2] );
    fill<8, true>( m_filterVer[0] ); fill<4, true>( m_filterVer[1] ); fill<2, true>( m_filterVer[2] );
    m_filterCopy[0][0] = xCopy<false, false>; m_filterCopy[0][1] = xCopy<false, true>; m_filterCopy[1][0] = xCopy<true, false>; m_filterCopy[1][1] = xCopy<true, true>;
  }
};

// FwdTrans / InvTrans signatures (TrQuant.h:53-54) and the [type][log2(N) - 1] tables (TrQuant.cpp:69-81)
typedef void FwdTrans( const TCoeff *, TCoeff *, int, int, int, int );
typedef void InvTrans( const TCoeff *, TCoeff *, int, int, int, int, const TCoeff, const TCoeff );
template<int TYPE, int N> void xFwd( const TCoeff *s, TCoeff *d, int shift, int line, int skip1, int skip2 ) { check( vtmhip_fastFwdTrans( context(), TYPE, N, s, d, shift, line, skip1, skip2 ), "fastFwdTrans" ); }
template<int TYPE, int N> void xInv( const TCoeff *s, TCoeff *d, int shift, int line, int skip1, int skip2, const TCoeff lo, const TCoeff hi ) { check( vtmhip_fastInvTrans( context(), TYPE, N, s, d, shift, line, skip1, skip2, lo, hi ), "fastInvTrans" ); }
inline FwdTrans **fastFwdTransTable()
{
  static FwdTrans *t[3 * 6] = { xFwd<0, 2>, xFwd<0, 4>, xFwd<0, 8>, xFwd<0, 16>, xFwd<0, 32>, xFwd<0, 64>,
                                nullptr,    xFwd<1, 4>, xFwd<1, 8>, xFwd<1, 16>, xFwd<1, 32>, nullptr,
                                nullptr,    xFwd<2, 4>, xFwd<2, 8>, xFwd<2, 16>, xFwd<2, 32>, nullptr };
  return t;
}
inline InvTrans **fastInvTransTable()
{
  static InvTrans *t[3 * 6] = { xInv<0, 2>, xInv<0, 4>, xInv<0, 8>, xInv<0, 16>, xInv<0, 32>, xInv<0, 64>,
                                nullptr,    xInv<1, 4>, xInv<1, 8>, xInv<1, 16>, xInv<1, 32>, nullptr,
                                nullptr,    xInv<2, 4>, xInv<2, 8>, xInv<2, 16>, xInv<2, 32>, nullptr };
  return t;
}
inline FwdTrans *fastFwdTrans( int type, int sizeIdx ) { initHIP(); return fastFwdTransTable()[type * 6 + sizeIdx]; }
inline InvTrans *fastInvTrans( int type, int sizeIdx ) { initHIP(); return fastInvTransTable()[type * 6 + sizeIdx]; }

}   // namespace vtmhip
