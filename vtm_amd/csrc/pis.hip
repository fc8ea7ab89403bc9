// pis.hip -- the per-PU arithmetic of InterSearch::predInterSearch around the batched searches (reference EncoderLib/InterSearch.cpp):
//   xEstimateMvPredAMVP :3088-3128 + xGetTemplateCost :3235-3270   (vtmhip_xEstimateMvPredAMVP_batch_dev, hook B7)
//   xCheckBestMVP :3185-3232, the reference selection of the uni loop :2354-2450, the bi iteration :2452-2640 (one iteration: FEN), the
//   uni / bi decision :2846-2893                                   (vtmhip_pis_stage)
// Heavy work stays in the library's kernels (motion_comp_kernel, dist_uniform_kernel, the searches behind vtmhip_xMotionEstimation_batch_dev);
// here one thread per row / PU turns one stage's results into the next stage's job records, fp64 exactly where the reference uses it.
#include "ctx.hpp"

namespace
{

__device__ __forceinline__ unsigned eg_bits( int v )   // RdCost::xGetExpGolombNumberOfBits (RdCost.h:301-313), closed form (see mest.hip)
{
  const unsigned t = ( v <= 0 ) ? ( ( unsigned ) ( -v ) << 1 ) + 1 : ( unsigned ) ( v << 1 );
  return 1u + ( ( unsigned ) ( 31 - __clz( ( int ) t ) ) << 1 );
}
__device__ __forceinline__ unsigned long long rate( double lambda, unsigned bits ) { return ( unsigned long long ) ( lambda * bits ); }   // RdCost::getCost
__device__ __forceinline__ int prec_down( int v, int rs ) { const int o = 1 << ( rs - 1 ); return v >= 0 ? ( v + o - 1 ) >> rs : ( v + o ) >> rs; }   // Mv::changePrecision
__device__ __forceinline__ int amvr_shift( int imv ) { return imv == 0 ? 2 : imv == 1 ? 4 : imv == 2 ? 6 : 3; }
__device__ __forceinline__ void clip_mv( const vtmhip_pic_params &pic, const vtmhip_me_job &j, int &hor, int &ver )   // clipMvInPic (Mv.cpp:56-74)
{
  const int horMax = ( pic.picW + 8 - j.puX - 1 ) << 4, horMin = ( -pic.ctuSize - 8 - j.puX + 1 ) << 4;
  const int verMax = ( pic.picH + 8 - j.puY - 1 ) << 4, verMin = ( -pic.ctuSize - 8 - j.puY + 1 ) << 4;
  hor = min( horMax, max( horMin, hor ) );
  ver = min( verMax, max( verMin, ver ) );
}

// ---- xEstimateMvPredAMVP ---------------------------------------------------------------------------------------------------------
struct AmvpWork
{
  unsigned long long *dout;   // [2n] SAD of the candidates' predictions (motion_comp_amvp_kernel, mc.hip: job 2 * row + c = candidate c of ME row `row`)
};

__global__ __launch_bounds__( 256 ) void amvp_select_kernel( vtmhip_me_job *__restrict__ jobs, int n, AmvpWork wk, int addIdxBits, unsigned long long *__restrict__ distBiP )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= n ) return;
  vtmhip_me_job &j = jobs[i];
  unsigned long long best = ~0ull;
  int                bestIdx = 0;
  for( int c = 0; c < j.numAmvpCand && c < 2; c++ )
  {
    const unsigned long long cost = wk.dout[2 * i + c] + rate( j.motionLambda, j.mvpIdxBits[c] );
    if( best > cost ) { best = cost; bestIdx = c; }
  }
  j.mvPredHor = j.amvpCand[bestIdx][0]; j.mvPredVer = j.amvpCand[bestIdx][1];
  j.mvpIdx    = ( uint8_t ) bestIdx;
  if( addIdxBits ) j.bits += j.mvpIdxBits[bestIdx];
  if( distBiP ) distBiP[i] = best;
}

// ---- predInterSearch glue ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uni_row( const vtmhip_pis_level &L, int list, int ref, int pu ) { return ( ( list ? L.numRef[0] : 0 ) + ref ) * L.numPU + pu; }
__device__ __forceinline__ unsigned ref_idx_bits( int numRef, int r ) { return numRef > 1 ? ( unsigned ) ( r + 1 - ( r == numRef - 1 ) ) : 0u; }   // :2357-2364

// xCheckBestMVP (:3185-3232): the vector is re-priced against the other AMVP candidate; a cheaper predictor replaces the chosen one
__device__ __forceinline__ void check_best_mvp( const vtmhip_me_job &j, vtmhip_pis_row &r )
{
  if( j.imv > 0 && j.imv < 3 ) return;
  if( j.numAmvpCand < 2 ) return;
  const int sh = amvr_shift( j.imv );
  const int mh = prec_down( r.mvHor, sh ), mv = prec_down( r.mvVer, sh );
  int       bestIdx  = r.mvpIdx;
  const int orgBits  = ( int ) ( eg_bits( mh - prec_down( r.mvPredHor, sh ) ) + eg_bits( mv - prec_down( r.mvPredVer, sh ) ) + j.mvpIdxBits[r.mvpIdx & 1] );
  int       bestBits = orgBits;
  for( int c = 0; c < 2; c++ )
  {
    if( c == r.mvpIdx ) continue;
    const int b = ( int ) ( eg_bits( mh - prec_down( j.amvpCand[c][0], sh ) ) + eg_bits( mv - prec_down( j.amvpCand[c][1], sh ) ) + j.mvpIdxBits[c] );
    if( b < bestBits ) { bestBits = b; bestIdx = c; }
  }
  if( bestIdx != r.mvpIdx )
  {
    r.mvPredHor = j.amvpCand[bestIdx][0]; r.mvPredVer = j.amvpCand[bestIdx][1];
    r.mvpIdx    = bestIdx;
    const unsigned orgAll = r.bits;
    r.bits = orgAll - ( unsigned ) orgBits + ( unsigned ) bestBits;
    r.cost = ( r.cost - rate( j.motionLambda, orgAll ) ) + rate( j.motionLambda, r.bits );
  }
}

__global__ __launch_bounds__( 256 ) void pis_cands_kernel( vtmhip_pis_level L )
{
  const int row = blockIdx.x * 256 + threadIdx.x, rows = ( L.numRef[0] + L.numRef[1] ) * L.numPU;
  if( row >= rows ) return;
  const int lr = row / L.numPU, pu = row - lr * L.numPU;
  const int list = lr >= L.numRef[0], ref = lr - ( list ? L.numRef[0] : 0 );
  vtmhip_me_job &j = L.uniJobs[row];
  int ph = 0, pv = 0;
  const int par = L.parentIdx ? L.parentIdx[pu] : -1;
  if( par >= 0 ) { const vtmhip_pis_row &p = L.parentRows[lr * L.parentNumPU + par]; ph = p.mvHor; pv = p.mvVer; }
  j.amvpCand[0][0] = ph; j.amvpCand[0][1] = pv; j.amvpCand[1][0] = 0; j.amvpCand[1][1] = 0;
  j.numAmvpCand = 2; j.mvpIdxBits[0] = 1; j.mvpIdxBits[1] = 1;   // m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] (xGetMvpIdxBits :3137-3162)
  j.bi = 0; j.numExtraStart = 0;
  j.bits = L.mbBits[list] + ref_idx_bits( L.numRef[list], ref );
}

// m_uniMvList as the bi stage and the SMVD block see it for row (list, refIdx) of a PU: the caller's list (extraStart of the uni row: the state BEFORE the uni loop) after
// insertUniMvCands( pu.Y(), cMvTemp ) (InterSearch.cpp:2451-2459, InterSearch.h:247-275) -- the block's own uni vector (selfH, selfV) replaces its existing entry in place,
// or becomes the newest entry (of 15 the oldest drops out).  Entry k of that list (newest first) -> (h, v); the list has uni_mv_list_size() entries.
__device__ __forceinline__ int uni_mv_list_size( const vtmhip_pis_level &L, int pu, const vtmhip_me_job &u )
{
  const int n0 = min( 15, max( 0, u.numExtraStart ) );
  const vtmhip_pis_pu_in *pi = L.puIn ? &L.puIn[pu] : nullptr;
  return ( pi && pi->uniMvInsert && pi->uniMvSelfIsNew ) ? min( 15, n0 + 1 ) : n0;
}
__device__ __forceinline__ void uni_mv_list_entry( const vtmhip_pis_level &L, int pu, const vtmhip_me_job &u, int selfH, int selfV, int k, int &h, int &v )
{
  const vtmhip_pis_pu_in *pi = L.puIn ? &L.puIn[pu] : nullptr;
  const bool ins = pi && pi->uniMvInsert;
  const int  src = ( ins && pi->uniMvSelfIsNew ) ? k - 1 : k;
  const bool self = ins && ( pi->uniMvSelfIsNew ? k == 0 : k == pi->uniMvSelfPos );
  h = self ? selfH : u.extraStart[max( src, 0 )][0];
  v = self ? selfV : u.extraStart[max( src, 0 )][1];
}

// CU-level BCW weight of the PU (puIn.bcwWeightL1): list 1's weight of 8, 0 when it is the default pair (4 / 4) or no per-PU record exists
__device__ __forceinline__ int bcw_weight_l1( const vtmhip_pis_level &L, int pu ) { const int w = L.puIn ? L.puIn[pu].bcwWeightL1 : 0; return w == 4 ? 0 : w; }
__device__ __forceinline__ unsigned bcw_idx_bits( const vtmhip_pis_level &L, int pu ) { return L.puIn ? L.puIn[pu].bcwIdxBits : 0u; }
// BcwFast (:2588-2593): under a non-default weight the refined list skips its pictures that carry the POC of the other list's chosen picture (cu.imv 0, temporal layer > 1)
__device__ __forceinline__ bool bcw_fast_skip( const vtmhip_pis_level &L, int pu, int imv, int rl, int ref, int otherRef )
{
  if( !L.puIn || !L.puIn[pu].bcwFastSkipPoc || !bcw_weight_l1( L, pu ) || imv != 0 ) return false;
  const int a = rl ? L.refPoc[1][ref] : L.refPoc[0][ref], b = rl ? L.refPoc[0][otherRef] : L.refPoc[1][otherRef];
  return a == b;
}

__device__ __forceinline__ void final_pred( const vtmhip_pis_level &L, int pu, const vtmhip_pis_pu &P )
{
  if( !L.predFinal ) return;      // vtmhip_predInterSearch_batch_dev: decisions only
  vtmhip_pred_job &pf = L.predFinal[pu];
  const bool bi = P.interDir == 3;
  const int  r0 = bi ? P.refIdxBi[0] : P.refIdx[0], r1 = bi ? P.refIdxBi[1] : P.refIdx[1];
  pf.mode = bi ? 2 : ( P.interDir == 2 ? 1 : 0 );
  if( P.interDir & 1 ) { pf.refOff[0] = L.refPlaneOff[0][r0] + L.pos[pu]; pf.mv[0][0] = bi ? P.mvBi[0][0] : P.mv[0][0]; pf.mv[0][1] = bi ? P.mvBi[0][1] : P.mv[0][1]; }
  if( P.interDir & 2 ) { pf.refOff[1] = L.refPlaneOff[1][r1] + L.pos[pu]; pf.mv[1][0] = bi ? P.mvBi[1][0] : P.mv[1][0]; pf.mv[1][1] = bi ? P.mvBi[1][1] : P.mv[1][1]; }
  // BDOF as InterPrediction::xPredInterBi decides it (:527-572; no weighted prediction / BCW / SMVD / CIIP / affine in this driver): true bi-prediction from
  // opposite directions at equal POC distance (PU::isBiPredFromDifferentDirEqDistPoc, UnitTools.cpp:3239-3260), w, h >= 8, w * h >= 128
  bool bio = false;
  if( bi && L.bdofEnabled )
  {
    const int d0 = L.curPoc - L.refPoc[0][r0], d1 = L.curPoc - L.refPoc[1][r1];
    bio = d0 * d1 < 0 && abs( d0 ) == abs( d1 ) && pf.width >= 8 && pf.height >= 8 && pf.width * pf.height >= 128 && !P.smvdMode;
  }
  pf.route = bio ? 1 : 2;
  if( L.predFinalC )
    for( int c = 0; c < 2; c++ )
    {
      vtmhip_pred_job &pc = L.predFinalC[c * L.numPU + pu];
      pc.mode = pf.mode; pc.route = 0;
      if( P.interDir & 1 ) { pc.refOff[0] = L.refPlaneOffC[c][0][r0] + L.posC[pu]; pc.mv[0][0] = pf.mv[0][0]; pc.mv[0][1] = pf.mv[0][1]; }
      if( P.interDir & 2 ) { pc.refOff[1] = L.refPlaneOffC[c][1][r1] + L.posC[pu]; pc.mv[1][0] = pf.mv[1][0]; pc.mv[1][1] = pf.mv[1][1]; }
    }
}

// Before the uni / bi decision list 1 is re-read from its "valid" rows only (:2438-2446, 2826-2829): a list-1 picture that is a list-0 picture too competes as list 0, not as
// list 1 (the unrestricted best of list 1 served the bi stage until here)
__device__ __forceinline__ void take_valid_list1( const vtmhip_pis_level &L, int pu, vtmhip_pis_pu &P )
{
  bool any = false;
  for( int ref = 0; ref < L.numRef[1]; ref++ ) any |= L.list1FromList0[ref] > 0;
  if( !any ) return;
  P.cost[1] = ~0ull; P.bits[1] = ~0u; P.refIdx[1] = 0; P.mv[1][0] = P.mv[1][1] = 0;
  for( int ref = 0; ref < L.numRef[1]; ref++ )
  {
    if( L.list1FromList0[ref] > 0 ) continue;
    const vtmhip_pis_row &r = L.uniRows[uni_row( L, 1, ref, pu )];
    if( r.cost < P.cost[1] ) { P.cost[1] = r.cost; P.bits[1] = r.bits; P.mv[1][0] = r.mvHor; P.mv[1][1] = r.mvVer; P.refIdx[1] = ref; }
  }
}

__global__ __launch_bounds__( 256 ) void pis_uni_select_kernel( vtmhip_pis_level L )
{
  const int pu = blockIdx.x * 256 + threadIdx.x;
  if( pu >= L.numPU ) return;
  vtmhip_pis_pu P;
  P.cost[0] = P.cost[1] = P.costBi = ~0ull;
  P.bits[0] = P.bits[1] = P.bits[2] = 0;
  P.refIdx[0] = P.refIdx[1] = P.refIdxBi[0] = P.refIdxBi[1] = -1;
  P.mv[0][0] = P.mv[0][1] = P.mv[1][0] = P.mv[1][1] = 0;
  P.mvBi[0][0] = P.mvBi[0][1] = P.mvBi[1][0] = P.mvBi[1][1] = 0;
  P.refineList = 0; P.interDir = 1; P.smvdMode = 0; P.mvpIdxL1Zero = 0; P.pad = 0;
  for( int list = 0; list < 2; list++ )
    for( int ref = 0; ref < L.numRef[list]; ref++ )
    {
      const int            row = uni_row( L, list, ref, pu );
      const vtmhip_me_job &j   = L.uniJobs[row];
      vtmhip_pis_row r;
      const int from0 = ( list == 1 && L.fastMEForGenBLowDelay ) ? L.list1FromList0[ref] - 1 : -1;
      if( from0 >= 0 && from0 < L.numRef[0] )
      {
        // FastMEForGenBLowDelay (:2391-2404): the same picture sits in list 0 -- its vector, and its cost with the rate part re-priced against this row's predictor
        // (getBitsOfVectorWithPredictor at cost scale 0: the raw difference shifted by imvShift + MV_FRACTIONAL_BITS_DIFF)
        const vtmhip_pis_row &r0 = L.uniRows[uni_row( L, 0, from0, pu )];      // written by this thread (after xCheckBestMVP: uiCostTempL0 / uiBitsTempL0 of :2423-2427)
        const int sh = ( j.imv == 3 ? 1 : ( int ) j.imv << 1 ) + 2;
        r.mvHor = r0.mvHor; r.mvVer = r0.mvVer; r.mvPredHor = j.mvPredHor; r.mvPredVer = j.mvPredVer; r.mvpIdx = j.mvpIdx;
        r.bits = j.bits + eg_bits( ( r.mvHor - j.mvPredHor ) >> sh ) + eg_bits( ( r.mvVer - j.mvPredVer ) >> sh );
        r.cost = r0.cost - rate( j.motionLambda, r0.bits ) + rate( j.motionLambda, r.bits );
        vtmhip_me_out c;       // the row's "search result" for the caller (m_uniMotions stores vector and cost before xCheckBestMVP)
        c.mvHor = r.mvHor; c.mvVer = r.mvVer; c.mvPredHor = r.mvPredHor; c.mvPredVer = r.mvPredVer; c.mvpIdx = r.mvpIdx; c.bits = r.bits; c.cost = r.cost;
        c.intX = c.intY = 0; c.intDist = 0;
        const_cast<vtmhip_me_out *>( L.uniOut )[row] = c;
      }
      else if( j.flags & VTMHIP_MEJ_GIVEN_UNI )
      {
        // xReadBufferedUniMv (:7677-7697): vector and distortion of the default-weight pass (m_uniMotions), the rate re-priced against this call's predictor -- both in the
        // AMVR precision (changeTransPrecInternal2Amvr), cost scale 0
        const vtmhip_me_out g = L.uniOut[row];
        const int sh = amvr_shift( j.imv );
        r.mvHor = g.mvHor; r.mvVer = g.mvVer; r.mvPredHor = j.mvPredHor; r.mvPredVer = j.mvPredVer; r.mvpIdx = j.mvpIdx;
        r.bits = j.bits + eg_bits( prec_down( r.mvHor, sh ) - prec_down( j.mvPredHor, sh ) ) + eg_bits( prec_down( r.mvVer, sh ) - prec_down( j.mvPredVer, sh ) );
        r.cost = g.cost + rate( j.motionLambda, r.bits );
        vtmhip_me_out c;
        c.mvHor = r.mvHor; c.mvVer = r.mvVer; c.mvPredHor = r.mvPredHor; c.mvPredVer = r.mvPredVer; c.mvpIdx = r.mvpIdx; c.bits = r.bits; c.cost = r.cost;
        c.intX = c.intY = 0; c.intDist = 0;
        const_cast<vtmhip_me_out *>( L.uniOut )[row] = c;
      }
      else
      {
        const vtmhip_me_out o = L.uniOut[row];
        r.mvHor = o.mvHor; r.mvVer = o.mvVer; r.mvPredHor = o.mvPredHor; r.mvPredVer = o.mvPredVer; r.mvpIdx = o.mvpIdx; r.bits = o.bits; r.cost = o.cost;
      }
      check_best_mvp( j, r );
      L.uniRows[row] = r;
      if( r.cost < P.cost[list] ) { P.cost[list] = r.cost; P.bits[list] = r.bits; P.mv[list][0] = r.mvHor; P.mv[list][1] = r.mvVer; P.refIdx[list] = ref; }
    }
  if( L.numRef[1] == 0 || L.biRestricted )      // P slice: list 0 it is; 8x4 / 4x8 PUs of a B slice (PU::isBipredRestriction): the cheaper list (:2866-2885)
  {
    if( L.numRef[1] ) take_valid_list1( L, pu, P );
    P.interDir = ( L.numRef[1] == 0 || P.cost[0] <= P.cost[1] ) ? 1 : 2;
    final_pred( L, pu, P );
  }
  L.pus[pu] = P;
}

__global__ __launch_bounds__( 256 ) void pis_bi_jobs_kernel( vtmhip_pis_level L )
{
  // VTMHIP_MAX_REF threads per PU: every thread derives the PU's refined list and the other list's bits (a few reads), thread 0 of the PU writes the PU's record and the
  // other list's prediction job, thread `ref` the bi row of reference picture `ref` (a 232-byte record each: side by side instead of one after the other)
  const int t = blockIdx.x * 256 + threadIdx.x, pu = t / VTMHIP_MAX_REF, ref = t % VTMHIP_MAX_REF;
  if( pu >= L.numPU ) return;
  const bool head = ref == 0;
  const vtmhip_pis_pu &P = L.pus[pu];
  // FASTINTERSEARCH_MODE1: refine the list with the larger cost (:2544-2556); a CU-level BCW weight: the list with the smaller |weight| (:2556-2559); MvdL1Zero: list 0 (:2576-2580)
  const int wL1 = bcw_weight_l1( L, pu );
  const int rl = L.mvdL1Zero ? 0 : wL1 ? ( abs( 8 - wL1 ) > abs( wL1 ) ? 1 : 0 ) : ( P.cost[0] <= P.cost[1] ? 1 : 0 ), ot = 1 - rl;
  const int wRefined = wL1 ? ( rl ? wL1 : 8 - wL1 ) : 0;      // getBcwWeight( bcwIdx, refined list )
  unsigned motOther;
  int      otherRef, otherMvH, otherMvV, mvpL1 = 0;
  if( L.mvdL1Zero )
  {
    // list 1 enters the bi mode AT its best AMVP predictor: the row with the smallest template cost (first minimum, :2382-2387), vector = that predictor, no vector difference
    unsigned long long bestDist = ~0ull;
    int                bestRef = 0, bestMvp = 0;
    for( int r = 0; r < L.numRef[1]; r++ )
    {
      const int row = uni_row( L, 1, r, pu );
      if( L.distBiP[row] < bestDist ) { bestDist = L.distBiP[row]; bestRef = r; bestMvp = L.uniJobs[row].mvpIdx; }
    }
    const vtmhip_me_job &u1 = L.uniJobs[uni_row( L, 1, bestRef, pu )];
    otherRef = bestRef; mvpL1 = bestMvp;
    otherMvH = bestMvp & 1 ? u1.amvpCand[1][0] : u1.amvpCand[0][0]; otherMvV = bestMvp & 1 ? u1.amvpCand[1][1] : u1.amvpCand[0][1];
    motOther = L.mbBits[1] + ref_idx_bits( L.numRef[1], bestRef ) + ( bestMvp & 1 ? u1.mvpIdxBits[1] : u1.mvpIdxBits[0] );      // uiMotBits[1] (:2506-2517)
  }
  else
  {
    otherRef = ot ? P.refIdx[1] : P.refIdx[0];
    otherMvH = ot ? P.mv[1][0] : P.mv[0][0]; otherMvV = ot ? P.mv[1][1] : P.mv[0][1];
    motOther = ( ot ? P.bits[1] : P.bits[0] ) - ( ot ? L.mbBits[1] : L.mbBits[0] );           // uiMotBits[1 - iRefList] (:2525-2527)
  }
  if( head )
  {
    vtmhip_pis_pu   &Pw = L.pus[pu];
    vtmhip_pred_job &po = L.predOther[pu];
    Pw.refineList = rl;
    po.mode = ( uint8_t ) ot;
    po.bcwWeight = ( int16_t ) wRefined;      // epilogue 2: the weighted target of removeHighFreq (:3320-3326)
    if( L.mvdL1Zero ) { Pw.mvBi[1][0] = otherMvH; Pw.mvBi[1][1] = otherMvV; Pw.refIdxBi[1] = otherRef; Pw.mvpIdxL1Zero = mvpL1; }
    int ph = otherMvH, pv = otherMvV;
    if( L.picW )      // clipMv of motionCompensation (the PU's own record keeps the unclipped vector)
    {
      const vtmhip_me_job &u0 = L.uniJobs[uni_row( L, 0, 0, pu )];
      ph = min( ( L.picW + 8 - u0.puX - 1 ) << 4, max( ( -L.ctuSize - 8 - u0.puX + 1 ) << 4, ph ) );
      pv = min( ( L.picH + 8 - u0.puY - 1 ) << 4, max( ( -L.ctuSize - 8 - u0.puY + 1 ) << 4, pv ) );
    }
    const int64_t off = ( ot ? L.refPlaneOff[1][otherRef] : L.refPlaneOff[0][otherRef] ) + L.pos[pu];
    if( ot ) { po.refOff[1] = off; po.mv[1][0] = ph; po.mv[1][1] = pv; }
    else     { po.refOff[0] = off; po.mv[0][0] = ph; po.mv[0][1] = pv; }
  }
  if( ref < ( rl ? L.numRef[1] : L.numRef[0] ) )
  {
    const int             row = uni_row( L, rl, ref, pu );
    const vtmhip_me_job  &u   = L.uniJobs[row];
    const vtmhip_pis_row &r   = L.uniRows[row];
    vtmhip_me_job        &b   = L.biJobs[ref * L.numPU + pu];
    b.refOff = u.refOff; b.refStride = u.refStride;
    b.bi = 1; b.imv = u.imv; b.mvpIdx = ( uint8_t ) r.mvpIdx; b.numAmvpCand = u.numAmvpCand;
    b.mvPredHor = r.mvPredHor; b.mvPredVer = r.mvPredVer;       // cMvPredBi (= cMvPred after xCheckBestMVP)
    b.mvHor = r.mvHor; b.mvVer = r.mvVer;                       // cMvTemp[iRefList][iRefIdxTemp]: the uni result is the start of the bi search
    b.amvpCand[0][0] = u.amvpCand[0][0]; b.amvpCand[0][1] = u.amvpCand[0][1]; b.amvpCand[1][0] = u.amvpCand[1][0]; b.amvpCand[1][1] = u.amvpCand[1][1];
    b.mvpIdxBits[0] = u.mvpIdxBits[0]; b.mvpIdxBits[1] = u.mvpIdxBits[1];
    b.bits = L.mbBits[2] + motOther + bcw_idx_bits( L, pu ) + ref_idx_bits( L.numRef[rl], ref ) + u.mvpIdxBits[r.mvpIdx & 1] + ( L.smvdBit ? 1u : 0u );   // :2594-2608
    b.searchRange = u.searchRange; b.motionLambda = u.motionLambda; b.flags = VTMHIP_MEJ_BCW_FLAGS( wRefined );
    const int nl = uni_mv_list_size( L, pu, u );   // the start candidates of the bi search (:3397-3426): none in the level-order driver
    b.numExtraStart = nl;
    for( int k = 0; k < nl; k++ ) { int eh, ev; uni_mv_list_entry( L, pu, u, r.mvHor, r.mvVer, k, eh, ev ); b.extraStart[k][0] = eh; b.extraStart[k][1] = ev; }
  }
}

__global__ __launch_bounds__( 256 ) void pis_final_kernel( vtmhip_pis_level L )
{
  const int pu = blockIdx.x * 256 + threadIdx.x;
  if( pu >= L.numPU ) return;
  vtmhip_pis_pu P  = L.pus[pu];
  const int     rl = P.refineList;
  // (constant indices only: a run-time index into P or into a local job record would put them into scratch memory)
  auto set_bi = [&]( int list, int h, int v, int ref )
  {
    if( list == 0 ) { P.mvBi[0][0] = h; P.mvBi[0][1] = v; P.refIdxBi[0] = ref; }
    else            { P.mvBi[1][0] = h; P.mvBi[1][1] = v; P.refIdxBi[1] = ref; }
  };
  if( !L.mvdL1Zero || rl == 0 ) set_bi( 0, P.mv[0][0], P.mv[0][1], P.refIdx[0] );      // (MvdL1Zero: stage 2 put list 1 at its predictor, and list 0 is the refined list)
  if( !L.mvdL1Zero || rl == 1 ) set_bi( 1, P.mv[1][0], P.mv[1][1], P.refIdx[1] );
  const int wL1 = bcw_weight_l1( L, pu );
  const int otherRef = L.mvdL1Zero ? P.refIdxBi[1] : ( rl ? P.refIdx[0] : P.refIdx[1] );      // pu.refIdx[1 - iRefList] during the iteration
  bool symRowSkipped = false;
  for( int ref = 0; ref < L.numRef[rl]; ref++ )
  {
    const vtmhip_me_job &j = L.biJobs[ref * L.numPU + pu];
    const vtmhip_me_out  o = L.biOut[ref * L.numPU + pu];
    vtmhip_pis_row r;
    r.mvHor = o.mvHor; r.mvVer = o.mvVer; r.mvPredHor = o.mvPredHor; r.mvPredVer = o.mvPredVer; r.mvpIdx = o.mvpIdx; r.bits = o.bits; r.cost = o.cost;
    if( bcw_fast_skip( L, pu, j.imv, rl, ref, otherRef ) )      // the reference `continue`s before the search (:2588-2593): the row's search result is not used
    {
      if( L.biRows ) { r.cost = ~0ull; L.biRows[ref * L.numPU + pu] = r; }
      if( rl == 0 && ref == L.symRefIdx[0] ) symRowSkipped = true;
      continue;
    }
    check_best_mvp( j, r );
    if( L.biRows ) L.biRows[ref * L.numPU + pu] = r;
    if( r.cost < P.costBi ) { P.costBi = r.cost; P.bits[2] = r.bits; set_bi( rl, r.mvHor, r.mvVer, ref ); }
  }
  if( L.smvdJobs )
  {
    // the SMVD block (:2656-2790) for list 0 and the slice's symmetric reference pair: AMVP lists of the two rows, start vectors cMvHevcTemp (the uni
    // result), cMvTemp (overwritten by the bi search when list 0 was the refined one) and cMvBi when it points at the symmetric reference
    const int s0 = L.symRefIdx[0], s1 = L.symRefIdx[1];
    const vtmhip_me_job &u0 = L.uniJobs[uni_row( L, 0, s0, pu )], &u1 = L.uniJobs[uni_row( L, 1, s1, pu )];
    const vtmhip_pis_row &r0 = L.uniRows[uni_row( L, 0, s0, pu )];
    vtmhip_smvd_job &j = L.smvdJobs[pu];   // written in place
    j.orgOff = u0.orgOff; j.refOff[0] = u0.refOff; j.refOff[1] = u1.refOff; j.orgStride = u0.orgStride; j.refStride[0] = u0.refStride; j.refStride[1] = u1.refStride;
    j.puX = u0.puX; j.puY = u0.puY; j.width = u0.width; j.height = u0.height;
    j.imv = u0.imv; j.useSatd = 1; j.clipBiPred = 0; j.bcwWeightTar = ( int8_t ) ( wL1 ? wL1 : 4 );      // the mirrored list is list 1
    j.numCand[0] = u0.numAmvpCand; j.numCand[1] = u1.numAmvpCand; j.skip = 0; j.pad_[0] = j.pad_[1] = j.pad_[2] = 0;
    for( int c = 0; c < 2; c++ )
    {
      j.cand[0][c][0] = u0.amvpCand[c][0]; j.cand[0][c][1] = u0.amvpCand[c][1]; j.cand[1][c][0] = u1.amvpCand[c][0]; j.cand[1][c][1] = u1.amvpCand[c][1];
      j.mvpIdxBits[c] = u0.mvpIdxBits[c];
    }
    j.modeBits = L.mbBits[2] + 1 + bcw_idx_bits( L, pu ); j.motionLambda = u0.motionLambda;      // :2779-2782
    int ns = 0;
    j.starts[ns][0] = r0.mvHor; j.starts[ns][1] = r0.mvVer; ns++;
    if( rl == 0 && !symRowSkipped ) { const vtmhip_me_out o = L.biOut[s0 * L.numPU + pu]; j.starts[ns][0] = o.mvHor; j.starts[ns][1] = o.mvVer; }
    else { j.starts[ns][0] = r0.mvHor; j.starts[ns][1] = r0.mvVer; }
    ns++;
    if( P.refIdxBi[0] == s0 ) { j.starts[ns][0] = P.mvBi[0][0]; j.starts[ns][1] = P.mvBi[0][1]; ns++; }
    j.numFixed = ( uint8_t ) ns;
    {
      // then the m_uniMvList entries of (list 0, symmetric reference), newest first (:2736-2742; the search rounds them to the AMVR precision and stops at five distinct vectors)
      const int nl = uni_mv_list_size( L, pu, u0 );
      for( int k = 0; k < nl && ns < VTMHIP_SMVD_MAX_START; k++ ) { int eh, ev; uni_mv_list_entry( L, pu, u0, r0.mvHor, r0.mvVer, k, eh, ev ); j.starts[ns][0] = eh; j.starts[ns][1] = ev; ns++; }
    }
    j.numStart = ( uint8_t ) ns;      // (entries beyond numStart are never read)
    for( int k = 0; k < 4; k++ ) { j.trace[k].cost = ~0ull; j.trace[k].mv[0] = j.trace[k].mv[1] = 0; j.trace[k].idx[0] = j.trace[k].idx[1] = 0; }
    j.mvCur[0] = j.mvCur[1] = j.mvTar[0] = j.mvTar[1] = 0;
    for( int l = 0; l < 2; l++ ) { j.predSym[l][0] = j.predSym[l][1] = 0; j.mvpIdxSym[l] = 0; }
    j.cost = ~0ull;
    L.pus[pu] = P;
    return;
  }
  take_valid_list1( L, pu, P );
  if( wL1 ) P.cost[0] = P.cost[1] = 0xffffffffull;      // enforceBcwPred (:2543, 2843-2847): uiCost[0] = uiCost[1] = MAX_UINT
  P.interDir = ( P.costBi <= P.cost[0] && P.costBi <= P.cost[1] ) ? 3 : ( P.cost[0] <= P.cost[1] ? 1 : 2 );   // :2846-2893
  final_pred( L, pu, P );
  L.pus[pu] = P;
}

__global__ __launch_bounds__( 256 ) void pis_smvd_merge_kernel( vtmhip_pis_level L )
{
  const int pu = blockIdx.x * 256 + threadIdx.x;
  if( pu >= L.numPU ) return;
  vtmhip_pis_pu          P = L.pus[pu];
  const vtmhip_smvd_job &j = L.smvdJobs[pu];
  const bool tried = !( L.puIn && L.puIn[pu].noSmvd );      // trySmvd (:2301)
  if( tried && j.cost < P.costBi )   // :2787-2803
  {
    P.costBi = j.cost; P.smvdMode = 1;
    P.mvBi[0][0] = j.mvCur[0]; P.mvBi[0][1] = j.mvCur[1]; P.refIdxBi[0] = L.symRefIdx[0];
    P.mvBi[1][0] = j.mvTar[0]; P.mvBi[1][1] = j.mvTar[1]; P.refIdxBi[1] = L.symRefIdx[1];
  }
  take_valid_list1( L, pu, P );
  if( bcw_weight_l1( L, pu ) ) P.cost[0] = P.cost[1] = 0xffffffffull;      // enforceBcwPred
  P.interDir = ( P.costBi <= P.cost[0] && P.costBi <= P.cost[1] ) ? 3 : ( P.cost[0] <= P.cost[1] ? 1 : 2 );
  final_pred( L, pu, P );
  L.pus[pu] = P;
}

// ---- affine uni jobs (stage 4) ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void pis_affine_jobs_kernel( vtmhip_pis_level L )
{
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int rows = ( L.numRef[0] + L.numRef[1] ) * L.numPU;
  if( row >= rows ) return;
  const int                  pu = row % L.numPU, list = row / L.numPU >= L.numRef[0] ? 1 : 0;
  const vtmhip_me_job       &u = L.uniJobs[row];
  const vtmhip_pis_row      &r = L.uniRows[row];
  const vtmhip_pis_pu       &P = L.pus[pu];
  vtmhip_affine_me_job       a;
  a.orgOff = u.orgOff; a.refOff = u.refOff; a.otherPredOff = 0; a.predOff = 0;
  a.orgStride = u.orgStride; a.refStride = u.refStride; a.otherPredStride = 0; a.predStride = 0;
  a.puX = u.puX; a.puY = u.puY; a.width = u.width; a.height = u.height;
  a.sixParam = 0; a.interDir = ( uint8_t ) ( 1 + list ); a.imv = 0; a.bi = 0; a.useSatd = 1; a.useAffineType = 1; a.amvrEncOpt = 0;
  a.lowDelayRounds = ( uint8_t ) L.affLowDelay; a.profAllowed = 1; a.profNeedsLargeGrad = ( uint8_t ) !L.affCheckLDC; a.profIsBi = 0; a.bcwWeight = 0;
  for( int c = 0; c < 3; c++ )
  {
    a.mvPred[c][0] = r.mvPredHor; a.mvPred[c][1] = r.mvPredVer;
    a.mv[c][0] = r.mvHor; a.mv[c][1] = r.mvVer;
  }
  a.bits = u.bits; a.pad1 = 0; a.motionLambda = u.motionLambda;
  unsigned long long hc = P.cost[0] < P.cost[1] ? P.cost[0] : P.cost[1];
  if( L.numRef[1] > 0 && P.costBi < hc ) hc = P.costBi;
  a.hevcCost = hc;
  a.numAmvpCand = 0; a.mvpIdx = 0;      // no AMVP list: the predictor stays (the other list fields are never read)
  L.affJobs[row] = a;
}

// ---- merge-candidate SATD (EncCu::xCheckRDCostMerge2Nx2N, EncCu.cpp:2399-2440): the distortion job of every candidate prediction ----
__global__ __launch_bounds__( 256 ) void merge_dist_jobs_kernel( const vtmhip_pred_job *__restrict__ plain, int nPlain, const vtmhip_pred_job *__restrict__ bdof, int nBdof,
                                                                const vtmhip_dmvr_job *__restrict__ dmvr, int nDmvr, int useSatd, vtmhip_dist_job *__restrict__ out )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= nPlain + nBdof + nDmvr ) return;
  vtmhip_dist_job d;
  if( i < nPlain + nBdof )
  {
    const vtmhip_pred_job &j = i < nPlain ? plain[i] : bdof[i - nPlain];
    d.orgOff = j.orgOff; d.curOff = j.predOff; d.orgStride = j.orgStride; d.curStride = j.predStride; d.width = j.width; d.height = j.height;
  }
  else
  {
    const vtmhip_dmvr_job &j = dmvr[i - nPlain - nBdof];
    d.orgOff = j.orgOff; d.curOff = j.predOff; d.orgStride = j.orgStride; d.curStride = j.predStride; d.width = j.width; d.height = j.height;
  }
  d.subShift = 0; d.kind = ( int16_t ) ( useSatd ? VTMHIP_DIST_SATD : VTMHIP_DIST_SAD );
  out[i] = d;
}

size_t align_up( size_t v ) { return ( v + 255 ) & ~( size_t ) 255; }

}   // namespace

extern "C"
{

int vtmhip_xEstimateMvPredAMVP_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                          vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, int uniformSize, int addIdxBits, uint64_t *d_distBiP )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && d_orgBase && d_refBase && d_jobs, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 4 && maxWidth <= 128 && maxHeight >= 4 && maxHeight <= 128, "maxWidth / maxHeight" );
  AmvpWork     wk;
  size_t       off = 0;
  const size_t oDout = off; off = align_up( off + 2 * ( size_t ) n * sizeof( unsigned long long ) );
  void *arena = nullptr;
  int   st    = vtmhip_internal_workspace( ctx, off, &arena );
  if( st ) return st;
  char *base = ( char * ) arena;
  wk.dout = ( unsigned long long * ) ( base + oDout );
  // xGetTemplateCost (:3162-3183): the candidate's prediction and its SAD against the original in ONE pass (no prediction buffer, no distortion jobs, and no
  // prediction-job table either: the kernel derives job 2 * row + c from the ME row)
  ( void ) uniformSize;
  st = vtmhip_internal_mc_amvp_launch( ctx, pic, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, wk.dout );
  if( st ) return st;
  hipLaunchKernelGGL( amvp_select_kernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_jobs, n, wk, addIdxBits, ( unsigned long long * ) d_distBiP );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"

// the first half of vtmhip_xEstimateMvPredAMVP_batch_dev: the template SADs of the rows' AMVP candidates ([2 * row + c]) into a workspace of the context's stream (slot 2: it must
// outlive the start of the fused integer search, whose prologue makes the selection -- mest.hip: vtmhip_internal_mest_with_amvp)
int vtmhip_internal_amvp_sads( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_me_job *d_jobs, int n,
                               int maxWidth, int maxHeight, unsigned long long **d_dout )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n > 0 && pic && d_orgBase && d_refBase && d_jobs && d_dout, "amvp sads: arguments" );
  void *arena = nullptr;
  int   st    = vtmhip_internal_workspace( ctx, align_up( 2 * ( size_t ) n * sizeof( unsigned long long ) ), &arena, 2 );
  if( st ) return st;
  *d_dout = ( unsigned long long * ) arena;
  return vtmhip_internal_mc_amvp_launch( ctx, pic, d_orgBase, d_refBase, d_jobs, n, maxWidth, maxHeight, *d_dout );
}

extern "C"
{

int vtmhip_merge_cand_satd_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                                      const vtmhip_pred_job *d_plain, int nPlain, const vtmhip_pred_job *d_bdof, int nBdof, const vtmhip_dmvr_job *d_dmvr, int nDmvr,
                                      int32_t *d_mvd, int maxWidth, int maxHeight, int uniformSize, int useSatd, uint64_t *d_dist )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, nPlain >= 0 && nBdof >= 0 && nDmvr >= 0, "n" );
  const int n = nPlain + nBdof + nDmvr;
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, pic && d_orgBase && d_refBase && d_predBase && d_dist, "null pointer" );
  VTMHIP_REQUIRE( ctx, ( nPlain == 0 || d_plain ) && ( nBdof == 0 || d_bdof ) && ( nDmvr == 0 || d_dmvr ), "null job table" );
  // the candidates' predictions (kept in d_predBase: the encoder's acMergeBuffer), by the kind of motion compensation the reference runs for them
  int st = vtmhip_motion_compensation_batch_dev( ctx, nullptr, d_refBase, d_predBase, nullptr, d_plain, nPlain, maxWidth, maxHeight );
  if( st ) return st;
  st = vtmhip_bdof_batch_dev( ctx, nullptr, d_refBase, d_predBase, nullptr, d_bdof, nBdof, maxWidth, maxHeight );
  if( st ) return st;
  st = vtmhip_dmvr_batch_dev( ctx, pic, nullptr, d_refBase, d_predBase, nullptr, d_dmvr, nDmvr, maxWidth, maxHeight, d_mvd );
  if( st ) return st;
  void *arena = nullptr;
  st = vtmhip_internal_workspace( ctx, ( size_t ) n * sizeof( vtmhip_dist_job ), &arena );
  if( st ) return st;
  vtmhip_dist_job *dj = ( vtmhip_dist_job * ) arena;
  hipLaunchKernelGGL( merge_dist_jobs_kernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_plain, nPlain, d_bdof, nBdof, d_dmvr, nDmvr, useSatd, dj );
  VTMHIP_LAUNCHED( ctx );
  if( uniformSize ) return vtmhip_dist_uniform_batch_dev( ctx, d_orgBase, d_predBase, dj, n, useSatd ? VTMHIP_DIST_SATD : VTMHIP_DIST_SAD, maxWidth, maxHeight, 0, d_dist );
  return vtmhip_dist_batch_dev( ctx, d_orgBase, d_predBase, dj, n, d_dist );
}

int vtmhip_pis_stage( vtmhip_ctx *ctx, const vtmhip_pis_level *lvl, int stage )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, lvl && stage >= 0 && stage <= 5, "level / stage" );
  VTMHIP_REQUIRE( ctx, lvl->numPU >= 0 && lvl->numRef[0] >= 1 && lvl->numRef[0] <= VTMHIP_MAX_REF && lvl->numRef[1] >= 0 && lvl->numRef[1] <= VTMHIP_MAX_REF, "numPU / numRef" );
  if( lvl->numPU == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, lvl->uniJobs && lvl->uniOut && lvl->uniRows && lvl->pus && lvl->pos && ( lvl->predFinal || lvl->candsGiven ), "null pointer in the level" );
  VTMHIP_REQUIRE( ctx, !lvl->parentIdx || lvl->parentRows, "parentIdx without parentRows" );
  VTMHIP_REQUIRE( ctx, !lvl->smvdJobs || ( lvl->symRefIdx[0] >= 0 && lvl->symRefIdx[0] < lvl->numRef[0] && lvl->symRefIdx[1] >= 0 && lvl->symRefIdx[1] < lvl->numRef[1] ),
                  "symRefIdx outside the reference lists" );
  const int  rows = ( lvl->numRef[0] + lvl->numRef[1] ) * lvl->numPU;
  const dim3 perPU( ( lvl->numPU + 255 ) / 256 ), perRow( ( rows + 255 ) / 256 ), tpb( 256 );
  if( stage == 0 ) hipLaunchKernelGGL( pis_cands_kernel, perRow, tpb, 0, ctx->stream, *lvl );
  else if( stage == 1 ) hipLaunchKernelGGL( pis_uni_select_kernel, perPU, tpb, 0, ctx->stream, *lvl );
  else if( stage == 5 )
  {
    VTMHIP_REQUIRE( ctx, lvl->smvdJobs && lvl->numRef[1] > 0, "stage 5 needs the SMVD job table of a B slice" );
    hipLaunchKernelGGL( pis_smvd_merge_kernel, perPU, tpb, 0, ctx->stream, *lvl );
  }
  else if( stage == 4 )
  {
    VTMHIP_REQUIRE( ctx, lvl->affJobs, "stage 4 needs the affine job table" );
    hipLaunchKernelGGL( pis_affine_jobs_kernel, perRow, tpb, 0, ctx->stream, *lvl );
  }
  else
  {
    VTMHIP_REQUIRE( ctx, lvl->numRef[1] >= 1 && lvl->predOther && lvl->biJobs && lvl->biOut, "the bi stages need list 1 and the bi tables" );
    VTMHIP_REQUIRE( ctx, !lvl->mvdL1Zero || ( lvl->distBiP && !lvl->smvdJobs ), "MvdL1Zero needs the template costs (distBiP) and has no SMVD block" );
    if( stage == 2 ) hipLaunchKernelGGL( pis_bi_jobs_kernel, dim3( ( lvl->numPU * VTMHIP_MAX_REF + 255 ) / 256 ), tpb, 0, ctx->stream, *lvl );
    else hipLaunchKernelGGL( pis_final_kernel, perPU, tpb, 0, ctx->stream, *lvl );
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

}   // extern "C"
