// TEST INFRASTRUCTURE ONLY -- part of oracle/ref_shim_enc.cpp (included inside its anonymous namespace).
//
// The SATD pre-selection of IntraSearch::estIntraPredLumaQT (EncoderLib/IntraSearch.cpp:549-592) as ONE batched device call per CU.
// The member tests its 35 first-round modes one by one: predIntraAng, then min( 2 * SAD, SATD ) through two DistParam::distFunc slots.  The loop lives in the middle of a
// 700-line member, so the hook works through the two surfaces that ARE replaceable:
//   arm     IntraPrediction::initIntraPatternChType( cu, area, forceRefFilterFlag = true ) (link-level hook) marks the start of a pre-selection
//   batch   the first distortion call after it reveals the original block; the hook then forms the predictors of ALL first-round modes itself (the reference's own
//           initPredIntraParams / predIntraAng -- the predictors stay host work, as the C ABI says), uploads them and gets the 35 SADs + 35 SATDs from ONE
//           vtmhip_intra_cand_cost_batch_dev call
//   serve   this and the following distFunc calls of the loop return the batch's values (the prediction the reference formed for the mode must equal the hook's, byte
//           for byte); in compare mode the reference's function runs too and the values are compared.
// What the batch does not hold (the second-round neighbours, MRL lines, MIP) falls through to the ordinary trampoline.
struct IntraBatch
{
  bool                  armed = false, serving = false;
  IntraPrediction      *ip = nullptr;
  const CodingUnit     *cu = nullptr;
  const Pel            *org = nullptr;
  int                   w = 0, h = 0;
  std::vector<Pel>      preds;           // 67 slots of w * h
  uint64_t              sad[NUM_LUMA_MODE], satd[NUM_LUMA_MODE];
  bool                  have[NUM_LUMA_MODE];
};
IntraBatch g_intra;
bool       g_hookIntra = false;
int16_t   *d_intraOrg = nullptr, *d_intraPred = nullptr;
uint64_t  *d_intraOut = nullptr;
decltype( &vtmhip_intra_cand_cost_batch_dev ) g_apiIntra = nullptr;

bool intraAlloc()
{
  return A.dalloc( g_ctx, 64 * 64 * 2, ( void ** ) &d_intraOrg ) == VTMHIP_OK && A.dalloc( g_ctx, size_t( NUM_LUMA_MODE ) * 64 * 64 * 2, ( void ** ) &d_intraPred ) == VTMHIP_OK
      && A.dalloc( g_ctx, 2 * NUM_LUMA_MODE * 8, ( void ** ) &d_intraOut ) == VTMHIP_OK;
}

void intraArm( IntraPrediction *ip, const CodingUnit &cu, const CompArea &area )
{
  g_intra.armed = isLuma( area.compID ) && cu.firstPU != nullptr && area.width <= 64 && area.height <= 64;
  g_intra.serving = false;
  g_intra.ip = ip; g_intra.cu = &cu;
}

// kind 0 SAD, 1 SATD; true: `out` is the value to return
bool intraServe( const DistParam &p, int kind, FpDistFunc orig, Distortion &out )
{
  IntraBatch &I = g_intra;
  PredictionUnit &pu = *const_cast<PredictionUnit *>( static_cast<const PredictionUnit *>( I.cu->firstPU ) );
  if( I.armed )
  {
    I.armed = false;
    const int w = p.org.width, h = p.org.height;
    if( pu.multiRefIdx != 0 || I.cu->mipFlag || I.cu->ispMode || p.subShift != 0 || w != ( int ) pu.Y().width || h != ( int ) pu.Y().height || p.bitDepth > 10 ) return false;
    g_st->intraBatches[0]++;
    I.org = p.org.buf; I.w = w; I.h = h;
    I.preds.resize( size_t( NUM_LUMA_MODE ) * w * h );
    memset( I.have, 0, sizeof( I.have ) );
    const uint8_t keepDir = pu.intraDir[0];
    std::vector<int> modes;
    for( int m = 0; m < NUM_LUMA_MODE; m++ ) if( m <= DC_IDX || !( m & 1 ) ) modes.push_back( m );      // the first round of the pre-selection (:555-565)
    std::vector<Pel> blk( size_t( w ) * h ), cand( modes.size() * size_t( w ) * h );
    for( int y = 0; y < h; y++ ) memcpy( &blk[size_t( y ) * w], p.org.buf + ptrdiff_t( y ) * p.org.stride, sizeof( Pel ) * w );
    for( size_t k = 0; k < modes.size(); k++ )
    {
      pu.intraDir[0] = ( uint8_t ) modes[k];
      I.ip->initPredIntraParams( pu, pu.Y(), *pu.cs->sps );
      PelBuf dst( &I.preds[size_t( modes[k] ) * w * h], w, w, h );
      I.ip->predIntraAng( COMPONENT_Y, dst, pu );
      memcpy( &cand[k * size_t( w ) * h], dst.buf, sizeof( Pel ) * w * h );
    }
    pu.intraDir[0] = keepDir;
    I.ip->initPredIntraParams( pu, pu.Y(), *pu.cs->sps );
    std::vector<uint64_t> res( 2 * modes.size() );
    const int n = ( int ) modes.size();
    const bool ok = A.h2d( g_ctx, d_intraOrg, blk.data(), blk.size() * 2 ) == VTMHIP_OK && A.h2d( g_ctx, d_intraPred, cand.data(), cand.size() * 2 ) == VTMHIP_OK
                 && g_apiIntra( g_ctx, d_intraOrg, 0, w, d_intraPred, 0, n, w, h, d_intraOut ) == VTMHIP_OK && A.d2h( g_ctx, res.data(), d_intraOut, res.size() * 8 ) == VTMHIP_OK;
    if( !ok ) { note_error(); return false; }
    g_st->intraBatches[1]++;
    for( int k = 0; k < n; k++ ) { I.sad[modes[k]] = res[k]; I.satd[modes[k]] = res[n + k]; I.have[modes[k]] = true; }
    I.serving = true;
  }
  if( !I.serving ) return false;
  const int mode = pu.intraDir[0];
  if( p.org.buf != I.org || ( int ) p.org.width != I.w || ( int ) p.org.height != I.h || pu.multiRefIdx != 0 || I.cu->mipFlag ) { I.serving = false; return false; }
  if( mode >= NUM_LUMA_MODE || !I.have[mode] ) return false;
  const Pel *mine = &I.preds[size_t( mode ) * I.w * I.h];
  for( int y = 0; y < I.h; y++ )
    if( memcmp( mine + size_t( y ) * I.w, p.cur.buf + ptrdiff_t( y ) * p.cur.stride, sizeof( Pel ) * I.w ) != 0 ) { g_st->intraBatches[3]++; return false; }   // not the prediction the batch was made for
  out = kind ? I.satd[mode] : I.sad[mode];
  g_st->intraServed++;
  if( !g_pisReplace )
  {
    const Distortion ref = orig( p );
    if( ref != out ) { if( g_st->intraMismatch++ == 0 ) { g_st->intraFirstMismatch[0] = mode; g_st->intraFirstMismatch[1] = I.w * 1000 + I.h; g_st->intraFirstMismatch[2] = kind; g_st->intraFirstMismatch[3] = ( int32_t ) ref; g_st->intraFirstMismatch[4] = ( int32_t ) out; } }
  }
  return true;
}
