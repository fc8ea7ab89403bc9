// transform.hip -- separable DCT-2 / DST-7 / DCT-8 (2..64 points), scalar quantisation / dequantisation.
//
// Reference: CommonLib/TrQuant.cpp xT :776-851, xIT :853-923, tables fastFwdTrans/fastInvTrans :69-81;
// CommonLib/TrQuant_EMT.cpp (the "fast" butterflies are exact refactorings of the matrix product
// _fastForwardMM :274-323 / _fastInverseMM :235-271 -- no intermediate rounding -- so a plain 32-bit integer
// matrix product is bit-identical, wrap-around included); core matrices CommonLib/RomTr.cpp (H.266 8.7.4.2
// transMatrix constants, regenerated procedurally in tr_tables.hpp); Quant::quant CommonLib/Quant.cpp:955-1038,
// Quant::dequant :357-482 (flat scaling list), scales CommonLib/Rom.cpp:463-473.
//
// One workgroup per transform unit.  The residual block, the intermediate (transposed) block and the transposed core
// matrix live in LDS; lane l of a wave produces output frequency k = l (matrix columns are contiguous in LDS, the data
// row is a broadcast), so LDS reads are conflict-free.
#include "ctx.hpp"
#include "bucket.hpp"
#include "tr_tables.hpp"

#include <cmath>
#include <vector>

namespace
{

struct TrTables { const int16_t *m[3][7]; };   // [type][log2 N] -> device pointer to the N x N forward matrix (row-major)

// The core matrices live in the CONTEXT (one copy per context, on the context's device, freed by vtmhip_destroy): no process-global
// state, so two contexts on two GPUs -- or two encoder threads with a context each -- cannot race or see the other device's pointers.
int ensure_tables( vtmhip_ctx *ctx )
{
  std::lock_guard<std::mutex> lock( ctx->initMutex );
  if( ctx->trTabBuf ) return VTMHIP_OK;
  VTMHIP_HIP( ctx, hipSetDevice( ctx->device ) );
  std::vector<int16_t> host;
  size_t               offs[3][7];
  for( int t = 0; t < 3; t++ )
    for( int l = 0; l < 7; l++ )
    {
      offs[t][l] = ( size_t ) -1;
      const int n = 1 << l;
      std::vector<int16_t> m( ( size_t ) n * n );
      if( n >= 2 && vtmhip_tr_matrix( t, n, m.data() ) == 0 )
      {
        offs[t][l] = host.size();
        host.insert( host.end(), m.begin(), m.end() );
      }
    }
  int16_t *dbuf = nullptr;
  VTMHIP_HIP( ctx, hipMalloc( ( void ** ) &dbuf, host.size() * sizeof( int16_t ) ) );
  if( hipMemcpy( dbuf, host.data(), host.size() * sizeof( int16_t ), hipMemcpyHostToDevice ) != hipSuccess )
  {
    ( void ) hipFree( dbuf );
    ctx->lastError = "hipMemcpy of the transform core matrices failed";
    return VTMHIP_E_HIP;
  }
  for( int t = 0; t < 3; t++ )
    for( int l = 0; l < 7; l++ ) ctx->trTab[t][l] = offs[t][l] == ( size_t ) -1 ? nullptr : dbuf + offs[t][l];
  ctx->trTabBuf = dbuf;
  return VTMHIP_OK;
}

TrTables tabs_of( const vtmhip_ctx *ctx )
{
  TrTables t;
  for( int a = 0; a < 3; a++ )
    for( int l = 0; l < 7; l++ ) t.m[a][l] = ctx->trTab[a][l];
  return t;
}

__device__ __forceinline__ int ilog2( int v ) { return 31 - __clz( v ); }
__device__ __forceinline__ int tr_skip( int type, int n ) { return ( type != VTMHIP_DCT2 && n == 32 ) ? 16 : ( n > 32 ? n - 32 : 0 ); }

// sMT[n * N + k] = M[k][n]
__device__ __forceinline__ void load_matrix_T( const int16_t *__restrict__ m, int N, int16_t *sMT )
{
  for( int i = threadIdx.x; i < N * N; i += blockDim.x )
  {
    const int k = i / N, n = i - k * N;
    sMT[n * N + k] = m[i];
  }
}

// dst[k * dstLd + j] = (sum_n M[k][n] * src[j * srcLd + n] + rnd) >> shift   for j < lines, k < kEff; zero for kEff <= k < N
__device__ __forceinline__ void fwd_pass( const int *src, int srcLd, int *dst, int dstLd, const int16_t *sMT, int N, int lines, int kEff, int shift )
{
  const int rnd = shift > 0 ? 1 << ( shift - 1 ) : 0;
  for( int o = threadIdx.x; o < lines * N; o += blockDim.x )
  {
    const int j = o / N, k = o - j * N;
    int       v = 0;
    if( k < kEff )
    {
      unsigned sum = 0;
      for( int n = 0; n < N; n++ ) sum += ( unsigned ) src[j * srcLd + n] * ( unsigned ) ( int ) sMT[n * N + k];
      v = ( int ) ( sum + ( unsigned ) rnd ) >> shift;
    }
    dst[k * dstLd + j] = v;
  }
}

// dst[i * dstLd + j] = clip((sum_{k<cut} src[k * srcLd + i] * M[k][j] + rnd) >> shift)   for i < lines; M[k][j] = sMT[j * N + k]... we need
// M row-major here: sM[k * N + j]
__device__ __forceinline__ void inv_pass( const int *src, int srcLd, int *dst, int dstLd, const int16_t *sM, int N, int lines, int linesEff, int cut,
                                          int shift, int cmin, int cmax )
{
  const unsigned rnd = 1u << ( shift - 1 );
  for( int o = threadIdx.x; o < lines * N; o += blockDim.x )
  {
    const int i = o / N, j = o - i * N;
    int       v = 0;
    if( i < linesEff )
    {
      unsigned sum = 0;
      for( int k = 0; k < cut; k++ ) sum += ( unsigned ) src[k * srcLd + i] * ( unsigned ) ( int ) sM[k * N + j];
      v = min( cmax, max( cmin, ( int ) ( sum + rnd ) >> shift ) );
    }
    dst[i * dstLd + j] = v;
  }
}

constexpr int TB = 64;   // MAX_TB_SIZEY

__global__ __launch_bounds__( 256 ) void xT_kernel( const int16_t *__restrict__ resiBase, int *__restrict__ coefBase,
                                                   const vtmhip_tr_job *__restrict__ jobs, TrTables tabs, int *__restrict__ sumAbsOut )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int ldsw[];
  const vtmhip_tr_job j = jobs[blockIdx.x];
  const int           w = j.width, h = j.height, bd = j.bitDepth;
  int                *blk = ldsw;                    // [h][w]
  int                *tmp = blk + w * h;             // [w][h+1]
  int16_t            *sMT = ( int16_t * ) ( tmp + w * ( h + 1 ) );
  const int16_t      *resi = resiBase + j.srcOff;
  int                *coef = coefBase + j.dstOff;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int y = i / w, x = i - y * w;
    blk[i]      = resi[( long ) y * j.srcStride + x];
  }
  const int skipW = tr_skip( j.typeHor, w ), skipH = tr_skip( j.typeVer, h );
  int       sumAbs = 0;
  if( w > 1 && h > 1 )
  {
    const int s1 = ilog2( w ) + bd + 6 - 15, s2 = ilog2( h ) + 6;
    load_matrix_T( tabs.m[j.typeHor][ilog2( w )], w, sMT );
    __syncthreads();
    fwd_pass( blk, w, tmp, h + 1, sMT, w, h, w - skipW, s1 );   // tmp[k][y]
    __syncthreads();
    load_matrix_T( tabs.m[j.typeVer][ilog2( h )], h, sMT );
    __syncthreads();
    // second pass straight to global: coef[k2 * w + j2], zero where j2 >= w - skipW or k2 >= h - skipH
    const int rnd = 1 << ( s2 - 1 ), kEff = h - skipH, jEff = w - skipW;
    for( int o = threadIdx.x; o < w * h; o += blockDim.x )
    {
      const int j2 = o / h, k2 = o - j2 * h;
      int       v  = 0;
      if( j2 < jEff && k2 < kEff )
      {
        unsigned sum = 0;
        for( int n = 0; n < h; n++ ) sum += ( unsigned ) tmp[j2 * ( h + 1 ) + n] * ( unsigned ) ( int ) sMT[n * h + k2];
        v = ( int ) ( sum + ( unsigned ) rnd ) >> s2;
      }
      coef[k2 * w + j2] = v;
      sumAbs += abs( v );
    }
  }
  else
  {
    // 1-D cases (W == 1 or H == 1, TrQuant.cpp:836-850)
    const int n = h == 1 ? w : h, type = h == 1 ? j.typeHor : j.typeVer, skip = h == 1 ? skipW : skipH;
    const int s = ilog2( n ) + bd + 6 - 15;
    load_matrix_T( tabs.m[type][ilog2( n )], n, sMT );
    __syncthreads();
    fwd_pass( blk, n, tmp, 1, sMT, n, 1, n - skip, s );
    __syncthreads();
    for( int i = threadIdx.x; i < n; i += blockDim.x ) { coef[i] = tmp[i]; sumAbs += abs( tmp[i] ); }
  }
  if( sumAbsOut )
  {
    // sum |coef| of the block (MTS candidate pre-selection, TrQuant.cpp:986-990); block-wide reduction through LDS
    __syncthreads();
    sumAbs = wave_reduce_add( sumAbs );
    if( ( threadIdx.x & 63 ) == 0 ) blk[threadIdx.x >> 6] = sumAbs;
    __syncthreads();
    if( threadIdx.x == 0 )
    {
      int t = 0;
      for( int k = 0; k < ( int ) ( blockDim.x >> 6 ); k++ ) t += blk[k];
      sumAbsOut[blockIdx.x] = t;
    }
  }
}

__global__ __launch_bounds__( 256 ) void xIT_kernel( const int *__restrict__ coefBase, int16_t *__restrict__ resiBase,
                                                    const vtmhip_tr_job *__restrict__ jobs, TrTables tabs )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int ldsw[];
  const vtmhip_tr_job j = jobs[blockIdx.x];
  const int           w = j.width, h = j.height, bd = j.bitDepth;
  int                *blk = ldsw;                  // coefficients [h][w]
  int                *tmp = blk + w * h;           // [w][h]
  int16_t            *sM  = ( int16_t * ) ( tmp + w * h );
  const int          *coef = coefBase + j.srcOff;
  int16_t            *resi = resiBase + j.dstOff;
  const int           cmin = -32768, cmax = 32767;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x ) blk[i] = coef[i];
  const int skipW = tr_skip( j.typeHor, w ), skipH = tr_skip( j.typeVer, h );
  if( w > 1 && h > 1 )
  {
    const int s1 = 7, s2 = 20 - bd;
    {
      const int16_t *m = tabs.m[j.typeVer][ilog2( h )];
      for( int i = threadIdx.x; i < h * h; i += blockDim.x ) sM[i] = m[i];
    }
    __syncthreads();
    inv_pass( blk, w, tmp, h, sM, h, w, w - skipW, h - skipH, s1, cmin, cmax );   // tmp[i (x-frequency)][y]
    __syncthreads();
    {
      const int16_t *m = tabs.m[j.typeHor][ilog2( w )];
      for( int i = threadIdx.x; i < w * w; i += blockDim.x ) sM[i] = m[i];
    }
    __syncthreads();
    const unsigned rnd = 1u << ( s2 - 1 );
    const int      cut = w - skipW;
    for( int o = threadIdx.x; o < w * h; o += blockDim.x )
    {
      const int y = o / w, x = o - y * w;
      unsigned  sum = 0;
      for( int k = 0; k < cut; k++ ) sum += ( unsigned ) tmp[k * h + y] * ( unsigned ) ( int ) sM[k * w + x];
      const int v = min( cmax, max( cmin, ( int ) ( sum + rnd ) >> s2 ) );
      resi[( long ) y * j.dstStride + x] = ( int16_t ) v;
    }
  }
  else
  {
    const int n = w == 1 ? h : w, type = w == 1 ? j.typeVer : j.typeHor, skip = w == 1 ? skipH : skipW;
    const int s = 20 - bd + 1;
    const int16_t *m = tabs.m[type][ilog2( n )];
    for( int i = threadIdx.x; i < n * n; i += blockDim.x ) sM[i] = m[i];
    __syncthreads();
    inv_pass( blk, 1, tmp, n, sM, n, 1, 1, n - skip, s, cmin, cmax );
    __syncthreads();
    for( int i = threadIdx.x; i < n; i += blockDim.x )
    {
      if( w == 1 ) resi[( long ) i * j.dstStride] = ( int16_t ) tmp[i];
      else resi[i] = ( int16_t ) tmp[i];
    }
  }
}

// 1-D pointer-surface kernels (FwdTrans / InvTrans signatures, arbitrary int32 input)
__global__ __launch_bounds__( 256 ) void fwd1d_kernel( const int *__restrict__ src, int *__restrict__ dst, const int16_t *__restrict__ m, int N, int shift,
                                                      int line, int skip1, int skip2 )
{
  const int      rl = line - skip1, cut = N - skip2;
  const unsigned rnd = shift > 0 ? 1u << ( shift - 1 ) : 0;
  for( int o = blockIdx.x * blockDim.x + threadIdx.x; o < N * line; o += gridDim.x * blockDim.x )
  {
    const int k = o / line, j = o - k * line;
    int       v = 0;
    if( j < rl && k < cut )
    {
      unsigned sum = 0;
      for( int n = 0; n < N; n++ ) sum += ( unsigned ) src[j * N + n] * ( unsigned ) ( int ) m[k * N + n];
      v = ( int ) ( sum + rnd ) >> shift;
    }
    dst[o] = v;
  }
}

__global__ __launch_bounds__( 256 ) void inv1d_kernel( const int *__restrict__ src, int *__restrict__ dst, const int16_t *__restrict__ m, int N, int shift,
                                                      int line, int skip1, int skip2, int cmin, int cmax )
{
  const int      rl = line - skip1, cut = N - skip2;
  const unsigned rnd = 1u << ( shift - 1 );
  for( int o = blockIdx.x * blockDim.x + threadIdx.x; o < N * line; o += gridDim.x * blockDim.x )
  {
    const int i = o / N, jj = o - i * N;
    int       v = 0;
    if( i < rl )
    {
      unsigned sum = 0;
      for( int k = 0; k < cut; k++ ) sum += ( unsigned ) src[k * line + i] * ( unsigned ) ( int ) m[k * N + jj];
      v = min( cmax, max( cmin, ( int ) ( sum + rnd ) >> shift ) );
    }
    dst[o] = v;
  }
}

// ---- scalar quantisation (flat scaling list) -----------------------------------------------------------------------------
__constant__ int c_quantScales[2][6]    = { { 26214, 23302, 20560, 18396, 16384, 14564 }, { 18396, 16384, 14564, 13107, 11651, 10280 } };
__constant__ int c_invQuantScales[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };

__global__ __launch_bounds__( 256 ) void quant_kernel( const int *__restrict__ coefBase, int *__restrict__ qBase, int *__restrict__ deltaUBase,
                                                      const vtmhip_quant_job *__restrict__ jobs, int *__restrict__ absSumOut )
{
  __shared__ int            sRed[4];
  const vtmhip_quant_job j  = jobs[blockIdx.x];
  const int              w = j.width, h = j.height, lw = ilog2( w ), lh = ilog2( h );
  const int              needSqrt = ( ( lw + lh ) & 1 ) && !j.isTransformSkip;
  const int              scale    = c_quantScales[needSqrt][j.qpRem];
  const int              trShift  = 15 - j.bitDepth - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
  const int              qBits    = 14 + j.qpPer + ( j.isTransformSkip ? 0 : trShift );
  const long long        add      = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
  const int             *coef     = coefBase + j.srcOff;
  int                   *q        = qBase + j.dstOff;
  int                    sum      = 0;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    // the reference's scan covers only the 32x32 zero-out region of larger blocks: positions outside keep level 0 (Quant.cpp:1004-1008)
    if( ( i % w ) >= 32 || ( i / w ) >= 32 ) { q[i] = 0; if( deltaUBase ) deltaUBase[j.dstOff + i] = 0; continue; }
    const int       c   = coef[i];
    const long long t   = ( long long ) abs( c ) * scale;
    const int       mag = ( int ) ( ( t + add ) >> qBits );
    if( deltaUBase ) deltaUBase[j.dstOff + i] = ( int ) ( ( t - ( ( long long ) mag << qBits ) ) >> ( qBits - 8 ) );
    sum += mag;
    q[i] = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
  }
  sum = wave_reduce_add( sum );
  if( ( threadIdx.x & 63 ) == 0 ) sRed[threadIdx.x >> 6] = sum;
  __syncthreads();
  if( threadIdx.x == 0 ) absSumOut[blockIdx.x] = sRed[0] + sRed[1] + sRed[2] + sRed[3];
}

__global__ __launch_bounds__( 256 ) void dequant_kernel( const int *__restrict__ qBase, int *__restrict__ coefBase,
                                                        const vtmhip_quant_job *__restrict__ jobs )
{
  const vtmhip_quant_job j = jobs[blockIdx.x];
  const int              w = j.width, h = j.height, lw = ilog2( w ), lh = ilog2( h );
  const int              needSqrt   = ( ( lw + lh ) & 1 ) && !j.isTransformSkip;
  const int              trShift    = 15 - j.bitDepth - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
  const int              rightShift = 6 - ( ( j.isTransformSkip ? 0 : trShift ) + j.qpPer );
  const int              scale      = c_invQuantScales[needSqrt][j.qpRem];
  const int              inBits     = min( 16, 32 + rightShift - 7 );
  const int              inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
  const int             *q    = qBase + j.srcOff;
  int                   *coef = coefBase + j.dstOff;
  for( int i = threadIdx.x; i < w * h; i += blockDim.x )
  {
    const int qq = min( inMax, max( inMin, q[i] ) );
    int       v;
    if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * scale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
    else v = ( int ) ( ( unsigned ) ( qq * scale ) << ( -rightShift ) );
    coef[i] = min( 32767, max( -32768, v ) );
  }
}


// ---- fused residual-coding chain of one TU: xT -> Quant::quant -> Quant::dequant -> xIT -> SSE(residual, reconstructed residual) ----
// = transformNxN + invTransformNxN + getDistPart( DF_SSE ) of xEstimateInterResidualQT (EncoderLib/InterSearch.cpp:6637-6733) without the
// CABAC bit estimate in between (host).  Everything stays in LDS; TPT threads per TU (64: four independent TUs per workgroup, wave-level
// synchronisation only; 256: one TU per workgroup).
template<int TPT>
__device__ __forceinline__ void tu_sync()
{
  if( TPT == 64 ) { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();
}

template<int TPT>
__global__ __launch_bounds__( 256 ) void tu_chain_kernel( const int16_t *__restrict__ resiBase, const vtmhip_tu_job *__restrict__ jobs, int numJobs,
                                                         TrTables tabs, int *__restrict__ levelsBase, int16_t *__restrict__ recBase,
                                                         vtmhip_tu_result *__restrict__ results, int maxW, int maxH )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int ldsw[];
  __shared__ long long sRed[4][3];
  constexpr int TUS = 256 / TPT;
  const int     sub = TPT == 64 ? ( int ) ( threadIdx.x >> 6 ) : 0, t = TPT == 64 ? ( int ) ( threadIdx.x & 63 ) : ( int ) threadIdx.x;
  const int     jobIdx = blockIdx.x * TUS + sub;
  if( jobIdx >= numJobs ) return;   // TPT == 64: whole waves leave; TPT == 256: grid == numJobs
  const vtmhip_tu_job j = jobs[jobIdx];
  const int           w = j.width, h = j.height, bd = j.bitDepth;
  const int           mx = maxW > maxH ? maxW : maxH;
  const int           perTu = maxW * maxH + maxW * ( maxH + 1 ) + ( ( mx * mx + 1 ) >> 1 ) + ( ( maxW * maxH + 1 ) >> 1 );   // ints
  int                *blk = ldsw + sub * perTu;            // [h][w] residual -> coefficients -> dequantised coefficients
  int                *tmp = blk + maxW * maxH;             // [w][h+1] / [w][h]
  int16_t            *sM  = ( int16_t * ) ( tmp + maxW * ( maxH + 1 ) );
  int16_t            *sR  = sM + ( ( mx * mx + 1 ) & ~1 );   // residual copy for the SSE
  const int16_t      *resi = resiBase + j.resiOff;
  for( int i = t; i < w * h; i += TPT )
  {
    const int y = i / w, x = i - y * w;
    const int16_t v = resi[( long ) y * j.resiStride + x];
    blk[i] = v;
    sR[i]  = v;
  }
  const int lw = ilog2( w ), lh = ilog2( h );
  long long sumAbs = 0, absSum = 0, sse = 0;
  const bool ts = j.typeHor == VTMHIP_TRSKIP;   // MTS_SKIP candidate: xTransformSkip / xITransformSkip are plain copies (TrQuant.cpp:1200-1213, 925-941)
  const int skipW = ts ? 0 : tr_skip( j.typeHor, w ), skipH = ts ? 0 : tr_skip( j.typeVer, h );
  tu_sync<TPT>();
  if( ts )
  {
    // coefficients = residual samples; Quant::quant / dequant with useTransformSkip: no transform shift, no sqrt(2) compensation
    // (Quant.cpp:966-997, 357-482); the caller puts QpParam::per( true ) / rem( true ) into the job
    const int       qBits = 14 + j.qpPer;
    const long long add   = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
    const int       scale = c_quantScales[0][j.qpRem], iscale = c_invQuantScales[0][j.qpRem];
    const int       rightShift = 6 - j.qpPer;
    const int       inBits = min( 16, 32 + rightShift - 7 );
    const int       inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
    int            *levels = levelsBase ? levelsBase + j.outOff : nullptr;
    int16_t        *rec    = recBase ? recBase + j.outOff : nullptr;
    for( int i = t; i < w * h; i += TPT )
    {
      const int c = blk[i];
      sumAbs += abs( c );
      const long long tt  = ( long long ) abs( c ) * scale;
      const int       mag = ( int ) ( ( tt + add ) >> qBits );
      absSum += mag;
      const int q = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
      if( levels ) levels[i] = q;
      const int qq = min( inMax, max( inMin, q ) );
      int       v;
      if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
      else v = ( int ) ( ( unsigned ) ( qq * iscale ) << ( -rightShift ) );
      v = ( int ) ( int16_t ) min( 32767, max( -32768, v ) );
      if( rec ) rec[i] = ( int16_t ) v;
      const int d = c - v;
      sse += ( long long ) ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
    }
  }
  else
  {
  // ---- forward: TrQuant::xT ----------------------------------------------------------------------------------------------------
  {
    const int s1 = lw + bd + 6 - 15, s2 = lh + 6;
    const int16_t *m = tabs.m[j.typeHor][lw];
    for( int i = t; i < w * w; i += TPT ) { const int k = i / w, n = i - k * w; sM[n * w + k] = m[i]; }
    tu_sync<TPT>();
    {
      const int rnd = s1 > 0 ? 1 << ( s1 - 1 ) : 0, kEff = w - skipW;
      for( int o = t; o < h * w; o += TPT )
      {
        const int jj = o / w, k = o - jj * w;
        int       v  = 0;
        if( k < kEff )
        {
          unsigned sum = 0;
          for( int n = 0; n < w; n++ ) sum += ( unsigned ) blk[jj * w + n] * ( unsigned ) ( int ) sM[n * w + k];
          v = ( int ) ( sum + ( unsigned ) rnd ) >> s1;
        }
        tmp[k * ( h + 1 ) + jj] = v;
      }
    }
    tu_sync<TPT>();
    m = tabs.m[j.typeVer][lh];
    for( int i = t; i < h * h; i += TPT ) { const int k = i / h, n = i - k * h; sM[n * h + k] = m[i]; }
    tu_sync<TPT>();
    {
      const int rnd = 1 << ( s2 - 1 ), kEff = h - skipH, jEff = w - skipW;
      for( int o = t; o < w * h; o += TPT )
      {
        const int j2 = o / h, k2 = o - j2 * h;
        int       v  = 0;
        if( j2 < jEff && k2 < kEff )
        {
          unsigned sum = 0;
          for( int n = 0; n < h; n++ ) sum += ( unsigned ) tmp[j2 * ( h + 1 ) + n] * ( unsigned ) ( int ) sM[n * h + k2];
          v = ( int ) ( sum + ( unsigned ) rnd ) >> s2;
        }
        blk[k2 * w + j2] = v;
        sumAbs += abs( v );
      }
    }
    tu_sync<TPT>();
  }
  // ---- Quant::quant + Quant::dequant (flat scaling list), in place ------------------------------------------------------------
  {
    const int       needSqrt = ( lw + lh ) & 1;
    const int       trShift  = 15 - bd - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
    const int       qBits    = 14 + j.qpPer + trShift;
    const long long add      = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
    const int       scale    = c_quantScales[needSqrt][j.qpRem], iscale = c_invQuantScales[needSqrt][j.qpRem];
    const int       rightShift = 6 - ( trShift + j.qpPer );
    const int       inBits   = min( 16, 32 + rightShift - 7 );
    const int       inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
    int            *levels   = levelsBase ? levelsBase + j.outOff : nullptr;
    for( int i = t; i < w * h; i += TPT )
    {
      const int       c   = blk[i];
      const long long tt  = ( long long ) abs( c ) * scale;
      const int       mag = ( int ) ( ( tt + add ) >> qBits );
      absSum += mag;
      const int q = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
      if( levels ) levels[i] = q;
      const int qq = min( inMax, max( inMin, q ) );
      int       v;
      if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
      else v = ( int ) ( ( unsigned ) ( qq * iscale ) << ( -rightShift ) );
      blk[i] = min( 32767, max( -32768, v ) );
    }
    tu_sync<TPT>();
  }
  // ---- inverse: TrQuant::xIT, then SSE against the residual ------------------------------------------------------------------
  {
    const int s1 = 7, s2 = 20 - bd;
    const int16_t *m = tabs.m[j.typeVer][lh];
    for( int i = t; i < h * h; i += TPT ) sM[i] = m[i];
    tu_sync<TPT>();
    {
      const unsigned rnd = 1u << ( s1 - 1 );
      const int      linesEff = w - skipW, cut = h - skipH;
      for( int o = t; o < w * h; o += TPT )
      {
        const int i = o / h, jj = o - i * h;
        int       v = 0;
        if( i < linesEff )
        {
          unsigned sum = 0;
          for( int k = 0; k < cut; k++ ) sum += ( unsigned ) blk[k * w + i] * ( unsigned ) ( int ) sM[k * h + jj];
          v = min( 32767, max( -32768, ( int ) ( sum + rnd ) >> s1 ) );
        }
        tmp[i * h + jj] = v;
      }
    }
    tu_sync<TPT>();
    m = tabs.m[j.typeHor][lw];
    for( int i = t; i < w * w; i += TPT ) sM[i] = m[i];
    tu_sync<TPT>();
    {
      const unsigned rnd = 1u << ( s2 - 1 );
      const int      cut = w - skipW;
      int16_t       *rec = recBase ? recBase + j.outOff : nullptr;
      for( int o = t; o < w * h; o += TPT )
      {
        const int y = o / w, x = o - y * w;
        unsigned  sum = 0;
        for( int k = 0; k < cut; k++ ) sum += ( unsigned ) tmp[k * h + y] * ( unsigned ) ( int ) sM[k * w + x];
        const int v = min( 32767, max( -32768, ( int ) ( sum + rnd ) >> s2 ) );
        if( rec ) rec[o] = ( int16_t ) v;
        const int d = ( int ) sR[o] - ( int ) ( int16_t ) v;
        sse += ( long long ) ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
      }
    }
  }
  }   // !ts
  // ---- reduce the three sums over the TU's threads ------------------------------------------------------------------------------
  sumAbs = ( long long ) wave_reduce_add_u64( ( unsigned long long ) sumAbs );
  absSum = ( long long ) wave_reduce_add_u64( ( unsigned long long ) absSum );
  sse    = ( long long ) wave_reduce_add_u64( ( unsigned long long ) sse );
  if( TPT == 64 )
  {
    if( t == 0 ) { vtmhip_tu_result r; r.sse = ( uint64_t ) sse; r.sumAbs = ( int32_t ) sumAbs; r.absSum = ( int32_t ) absSum; results[jobIdx] = r; }
  }
  else
  {
    __syncthreads();
    if( ( threadIdx.x & 63 ) == 0 ) { sRed[threadIdx.x >> 6][0] = sumAbs; sRed[threadIdx.x >> 6][1] = absSum; sRed[threadIdx.x >> 6][2] = sse; }
    __syncthreads();
    if( threadIdx.x == 0 )
    {
      vtmhip_tu_result r;
      r.sumAbs = ( int32_t ) ( sRed[0][0] + sRed[1][0] + sRed[2][0] + sRed[3][0] );
      r.absSum = ( int32_t ) ( sRed[0][1] + sRed[1][1] + sRed[2][1] + sRed[3][1] );
      r.sse    = ( uint64_t ) ( sRed[0][2] + sRed[1][2] + sRed[2][2] + sRed[3][2] );
      results[jobIdx] = r;
    }
  }
}


// ---- the smallest TUs (4x4, 8x4, 4x8: the chroma TUs of 8x8 / 16x8 / 8x16 PUs; 4x4 luma) of a uniform batch: ONE LANE per TU ---------------------------
// The whole block lives in registers (W * H <= 32 values), the 4- / 8-point core matrices of the three types sit in LDS (staged once per workgroup);
// the generic kernel above spends 64 threads and a dozen LDS round trips on such a block.  Same arithmetic, same order of the integer operations.
template<int W, int H>
__global__ __launch_bounds__( 256 ) void tu_chain_lane_kernel( const int16_t *__restrict__ resiBase, const vtmhip_tu_job *__restrict__ jobs, int numJobs, TrTables tabs,
                                                              int *__restrict__ levelsBase, int16_t *__restrict__ recBase, vtmhip_tu_result *__restrict__ results )
{
  constexpr int LW = W == 4 ? 2 : 3, LH = H == 4 ? 2 : 3, N = W * H;
  __shared__ int16_t sMH[3][W * W], sMV[3][H * H];
  for( int i = threadIdx.x; i < 3 * W * W; i += 256 ) sMH[i / ( W * W )][i % ( W * W )] = tabs.m[i / ( W * W )][LW][i % ( W * W )];
  for( int i = threadIdx.x; i < 3 * H * H; i += 256 ) sMV[i / ( H * H )][i % ( H * H )] = tabs.m[i / ( H * H )][LH][i % ( H * H )];
  __syncthreads();
  const int jobIdx = blockIdx.x * 256 + threadIdx.x;
  if( jobIdx >= numJobs ) return;
  const vtmhip_tu_job j = jobs[jobIdx];
  if( j.width != W || j.height != H || j.typeHor > 2 || j.typeVer > 2 ) return;   // the caller's promise is broken: leave the result untouched
  const int      bd = j.bitDepth;
  const int16_t *mh = sMH[j.typeHor], *mv = sMV[j.typeVer];
  int            r[N], b[N], t[N];
  const int16_t *resi = resiBase + j.resiOff;
  if( ( ( j.resiOff | j.resiStride ) & 3 ) == 0 )   // 8-byte aligned rows (offsets and strides of 4-sample blocks normally are): 4 samples per load
  {
#pragma unroll
    for( int y = 0; y < H; y++ )
#pragma unroll
      for( int x = 0; x < W; x += 4 )
      {
        const int2 v = *reinterpret_cast<const int2 *>( resi + ( long ) y * j.resiStride + x );
        r[y * W + x] = ( int ) ( short ) v.x; r[y * W + x + 1] = v.x >> 16; r[y * W + x + 2] = ( int ) ( short ) v.y; r[y * W + x + 3] = v.y >> 16;
      }
  }
  else
  {
#pragma unroll
    for( int y = 0; y < H; y++ )
#pragma unroll
      for( int x = 0; x < W; x++ ) r[y * W + x] = resi[( long ) y * j.resiStride + x];
  }
  long long sumAbs = 0, absSum = 0, sse = 0;
  // forward: rows with the horizontal matrix, then columns with the vertical one (TrQuant::xT; no zero-out at these sizes)
  {
    const int s1 = LW + bd + 6 - 15, s2 = LH + 6;
    const int rnd1 = s1 > 0 ? 1 << ( s1 - 1 ) : 0, rnd2 = 1 << ( s2 - 1 );
#pragma unroll
    for( int y = 0; y < H; y++ )
#pragma unroll
      for( int k = 0; k < W; k++ )
      {
        unsigned sum = 0;
#pragma unroll
        for( int n = 0; n < W; n++ ) sum += ( unsigned ) r[y * W + n] * ( unsigned ) ( int ) mh[k * W + n];
        t[k * H + y] = ( int ) ( sum + ( unsigned ) rnd1 ) >> s1;
      }
#pragma unroll
    for( int x = 0; x < W; x++ )
#pragma unroll
      for( int k = 0; k < H; k++ )
      {
        unsigned sum = 0;
#pragma unroll
        for( int n = 0; n < H; n++ ) sum += ( unsigned ) t[x * H + n] * ( unsigned ) ( int ) mv[k * H + n];
        const int v = ( int ) ( sum + ( unsigned ) rnd2 ) >> s2;
        b[k * W + x] = v;
        sumAbs += abs( v );
      }
  }
  // Quant::quant + Quant::dequant, flat scaling list
  {
    constexpr int   needSqrt = ( LW + LH ) & 1;
    const int       trShift  = 15 - bd - ( ( LW + LH ) >> 1 ) + ( needSqrt ? -1 : 0 );
    const int       qBits    = 14 + j.qpPer + trShift;
    const long long add      = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
    const int       scale    = c_quantScales[needSqrt][j.qpRem], iscale = c_invQuantScales[needSqrt][j.qpRem];
    const int       rightShift = 6 - ( trShift + j.qpPer );
    const int       inBits   = min( 16, 32 + rightShift - 7 );
    const int       inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
    int            *levels   = levelsBase ? levelsBase + j.outOff : nullptr;
#pragma unroll
    for( int i = 0; i < N; i++ )
    {
      const int       c   = b[i];
      const long long tt  = ( long long ) abs( c ) * scale;
      const int       mag = ( int ) ( ( tt + add ) >> qBits );
      absSum += mag;
      const int q = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
      if( levels ) levels[i] = q;
      const int qq = min( inMax, max( inMin, q ) );
      int       v;
      if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
      else v = ( int ) ( ( unsigned ) ( qq * iscale ) << ( -rightShift ) );
      b[i] = min( 32767, max( -32768, v ) );
    }
  }
  // inverse: columns, then rows (TrQuant::xIT), SSE against the residual
  {
    const int      s2 = 20 - bd;
    const unsigned rnd1 = 1u << 6, rnd2 = 1u << ( s2 - 1 );
    int16_t       *rec = recBase ? recBase + j.outOff : nullptr;
#pragma unroll
    for( int x = 0; x < W; x++ )
#pragma unroll
      for( int y = 0; y < H; y++ )
      {
        unsigned sum = 0;
#pragma unroll
        for( int k = 0; k < H; k++ ) sum += ( unsigned ) b[k * W + x] * ( unsigned ) ( int ) mv[k * H + y];
        t[x * H + y] = min( 32767, max( -32768, ( int ) ( sum + rnd1 ) >> 7 ) );
      }
#pragma unroll
    for( int y = 0; y < H; y++ )
#pragma unroll
      for( int x = 0; x < W; x++ )
      {
        unsigned sum = 0;
#pragma unroll
        for( int k = 0; k < W; k++ ) sum += ( unsigned ) t[k * H + y] * ( unsigned ) ( int ) mh[k * W + x];
        const int v = min( 32767, max( -32768, ( int ) ( sum + rnd2 ) >> s2 ) );
        if( rec ) rec[y * W + x] = ( int16_t ) v;
        const int d = r[y * W + x] - v;
        sse += ( long long ) ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
      }
  }
  vtmhip_tu_result res;
  res.sse = ( uint64_t ) sse; res.sumAbs = ( int32_t ) sumAbs; res.absSum = ( int32_t ) absSum;
  results[jobIdx] = res;
}

template<int W, int H>
int launch_tu_lane( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int32_t *d_levelsBase, int16_t *d_recBase, vtmhip_tu_result *d_results,
                    const TrTables &tb )
{
  VTMHIP_TIME_KERNEL( ctx, "tu_chain_lane_kernel" );
  hipLaunchKernelGGL( ( tu_chain_lane_kernel<W, H> ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_resiBase, d_jobs, n, tb, d_levelsBase, d_recBase, d_results );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}


// ---- register-blocked fast path of the fused chain for batches of ONE TU size (W, H >= 8) --------------------------------------
// Every pass is out[r][c] = sum_n A(r, n) * B[n][c] with B = the core matrix in the orientation that makes B[n][c .. c+7]
// contiguous (forward: transposed, inverse: plain).  A lane owns 8 consecutive outputs of one row: per inner step it needs ONE
// 32-bit LDS read of A and ONE 128-bit LDS read of B for 8 multiply-adds (the simple kernel above needs two LDS reads per
// multiply-add and is LDS-issue bound).  LPT lanes share one TU; 256 / LPT TUs per workgroup; matrices staged once per workgroup.
//
// 24-bit multiplies are exact here: every B entry is a matrix coefficient (|m| <= 91) and every A value is bounded by 2^23 --
// residuals / clipped coefficients are 16-bit and the first forward pass of 16-bit input stays below (sum|m| * 32768) >> shift1
// < 2^23 for every size and bit depth >= 8.  v_mad_i32_i24 issues at full rate, v_mul_lo_u32 at a quarter of it.
template<int LPT>
__device__ __forceinline__ void tuq_sync()
{
  if( LPT <= 64 ) { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();
}

// rows x cols outputs (cols multiple of 8); rEff / cEff: outputs beyond them are zero (zero-out); inner: summation length
template<int LPT, bool CLIP>
__device__ __forceinline__ void tuq_pass( const int *A, int aRowStride, int aColStride, const int16_t *B, int ldb, int inner, int rows, int cols,
                                          int rEff, int cEff, int *out, int oRowStride, int oColStride, int shift, int t, long long *sumAbs )
{
  // one lane: a 2 x 8 block of outputs (rows r, r + 1): the 16-byte matrix read of a summation step feeds 16 multiply-adds
  const int cb = cols >> 3, lcb = 31 - __clz( cb ), rnd = shift > 0 ? 1 << ( shift - 1 ) : 0;
  for( int it = t; it < ( rows >> 1 ) * cb; it += LPT )
  {
    const int r = ( it >> lcb ) << 1, c0 = ( it & ( cb - 1 ) ) << 3;   // cb is a power of two (TU sizes are)
    int       acc[2][8];
#pragma unroll
    for( int i = 0; i < 8; i++ ) acc[0][i] = acc[1][i] = rnd;
    if( r < rEff && c0 < cEff )
    {
      const int *a0 = A + r * aRowStride, *a1 = a0 + aRowStride;
      for( int n = 0; n < inner; n++ )
      {
        const int  av0 = a0[n * aColStride], av1 = a1[n * aColStride];
        const int4 bv = *reinterpret_cast<const int4 *>( B + n * ldb + c0 );
        const int  b[8] = { ( int ) ( short ) bv.x, bv.x >> 16, ( int ) ( short ) bv.y, bv.y >> 16, ( int ) ( short ) bv.z, bv.z >> 16, ( int ) ( short ) bv.w, bv.w >> 16 };
#pragma unroll
        for( int i = 0; i < 8; i++ ) { acc[0][i] += __mul24( av0, b[i] ); acc[1][i] += __mul24( av1, b[i] ); }
      }
    }
#pragma unroll
    for( int q = 0; q < 2; q++ )
#pragma unroll
      for( int i = 0; i < 8; i++ )
      {
        int v = ( r + q < rEff && c0 + i < cEff ) ? acc[q][i] >> shift : 0;
        if( CLIP ) v = min( 32767, max( -32768, v ) );
        out[( r + q ) * oRowStride + ( c0 + i ) * oColStride] = v;
        if( sumAbs ) *sumAbs += abs( v );
      }
  }
}

// tuq_pass with the matrix in the pair-interleaved layout of tuq_pass16 (Bp[(n >> 1) * cols + c] = (B[n][c], B[n+1][c])): square TUs take the second
// forward pass's M^T from the slot the first one uses, so no plain copy is staged.  Two summation steps per trip; same loads per step as tuq_pass.
template<int LPT, bool CLIP>
__device__ __forceinline__ void tuq_pass_il( const int *A, int aRowStride, int aColStride, const unsigned *Bp, int inner, int rows, int cols, int rEff, int cEff,
                                             int *out, int oRowStride, int oColStride, int shift, int t, long long *sumAbs )
{
  const int cb = cols >> 3, lcb = 31 - __clz( cb ), rnd = shift > 0 ? 1 << ( shift - 1 ) : 0;
  for( int it = t; it < ( rows >> 1 ) * cb; it += LPT )
  {
    const int r = ( it >> lcb ) << 1, c0 = ( it & ( cb - 1 ) ) << 3;
    int       acc[2][8];
#pragma unroll
    for( int i = 0; i < 8; i++ ) acc[0][i] = acc[1][i] = rnd;
    if( r < rEff && c0 < cEff )
    {
      const int *a0 = A + r * aRowStride, *a1 = a0 + aRowStride;
      for( int n2 = 0; n2 < ( inner >> 1 ); n2++ )
      {
        const int   n = n2 << 1;
        const int   e0 = a0[n * aColStride], o0 = a0[( n + 1 ) * aColStride], e1 = a1[n * aColStride], o1 = a1[( n + 1 ) * aColStride];
        const uint4 b0 = *reinterpret_cast<const uint4 *>( Bp + n2 * cols + c0 ), b1 = *reinterpret_cast<const uint4 *>( Bp + n2 * cols + c0 + 4 );
        const unsigned bw[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
        for( int i = 0; i < 8; i++ )
        {
          const int be = ( int ) ( short ) bw[i], bo = ( int ) bw[i] >> 16;
          acc[0][i] += __mul24( e0, be ) + __mul24( o0, bo );
          acc[1][i] += __mul24( e1, be ) + __mul24( o1, bo );
        }
      }
    }
#pragma unroll
    for( int q = 0; q < 2; q++ )
#pragma unroll
      for( int i = 0; i < 8; i++ )
      {
        int v = ( r + q < rEff && c0 + i < cEff ) ? acc[q][i] >> shift : 0;
        if( CLIP ) v = min( 32767, max( -32768, v ) );
        out[( r + q ) * oRowStride + ( c0 + i ) * oColStride] = v;
        if( sumAbs ) *sumAbs += abs( v );
      }
  }
}

// The same product when every A value fits 16 bits (residuals; dequantised coefficients and the first inverse pass are clipped to
// 16 bits): v_dot2c_i32_i16 takes two summation steps per instruction.  A: int16, the summation index contiguous (rows of
// aRowStride samples, even); Bp: the matrix with rows n, n + 1 interleaved per column -- Bp[(n >> 1) * cols + c] = (B[n][c], B[n+1][c]).
template<int LPT, bool CLIP, class OutT>
__device__ __forceinline__ void tuq_pass16( const int16_t *A, int aRowStride, const unsigned *Bp, int inner, int rows, int cols, int rEff, int cEff, OutT *out,
                                            int oRowStride, int oColStride, int shift, int t )
{
  typedef short v2s __attribute__( ( ext_vector_type( 2 ) ) );
  const int cb = cols >> 3, lcb = 31 - __clz( cb ), rnd = shift > 0 ? 1 << ( shift - 1 ) : 0;
  for( int it = t; it < ( rows >> 1 ) * cb; it += LPT )   // a 2 x 8 block of outputs per lane
  {
    const int r = ( it >> lcb ) << 1, c0 = ( it & ( cb - 1 ) ) << 3;   // cb is a power of two (TU sizes are)
    int       acc[2][8];
#pragma unroll
    for( int i = 0; i < 8; i++ ) acc[0][i] = acc[1][i] = rnd;
    if( r < rEff && c0 < cEff )
    {
      const unsigned *a0 = reinterpret_cast<const unsigned *>( A + r * aRowStride ), *a1 = reinterpret_cast<const unsigned *>( A + ( r + 1 ) * aRowStride );
      for( int n2 = 0; n2 < ( inner >> 1 ); n2++ )
      {
        const unsigned av0 = a0[n2], av1 = a1[n2];
        const uint4    b0 = *reinterpret_cast<const uint4 *>( Bp + n2 * cols + c0 ), b1 = *reinterpret_cast<const uint4 *>( Bp + n2 * cols + c0 + 4 );
        const unsigned bw[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
        v2s va0, va1;
        __builtin_memcpy( &va0, &av0, 4 );
        __builtin_memcpy( &va1, &av1, 4 );
#pragma unroll
        for( int i = 0; i < 8; i++ )
        {
          v2s vb;
          __builtin_memcpy( &vb, &bw[i], 4 );
          acc[0][i] = __builtin_amdgcn_sdot2( va0, vb, acc[0][i], false );
          acc[1][i] = __builtin_amdgcn_sdot2( va1, vb, acc[1][i], false );
        }
      }
    }
#pragma unroll
    for( int q = 0; q < 2; q++ )
#pragma unroll
      for( int i = 0; i < 8; i++ )
      {
        int v = ( r + q < rEff && c0 + i < cEff ) ? acc[q][i] >> shift : 0;
        if( CLIP ) v = min( 32767, max( -32768, v ) );
        out[( r + q ) * oRowStride + ( c0 + i ) * oColStride] = ( OutT ) v;
      }
  }
}

template<int LPT>
__global__ __launch_bounds__( 256 ) void tu_chain_uni_kernel( const int16_t *__restrict__ resiBase, const vtmhip_tu_job *__restrict__ jobs, int numJobs,
                                                             TrTables tabs, int *__restrict__ levelsBase, int16_t *__restrict__ recBase,
                                                             vtmhip_tu_result *__restrict__ results, int w, int h, int *__restrict__ fwdCoefBase )
{
  // fwdCoefBase != nullptr: forward transform only (TrQuant::xT): coefficients to fwdCoefBase + outOff, sum|coef| to results
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int ldsw[];
  __shared__ long long sRed[4][3];
  constexpr int TUS = 256 / LPT;
  const int     sub = threadIdx.x / LPT, t = threadIdx.x - sub * LPT;
  const int     perTu = w * h + w * ( h + 1 );   // ints: blk, tmp (the 16-bit residual copy lives in blk until the second forward pass overwrites it)
  int16_t      *sMat = ( int16_t * ) ( ldsw + TUS * perTu );              // [dim][type][orientation][N*N]
  const int     lw = ilog2( w ), lh = ilog2( h );
  const int     hBase = ( w > 32 ? 2 : 6 ) * w * w;   // only DCT-2 exists above 32: one matrix pair instead of three
  const bool    sq    = w == h;                       // square TUs: both height matrices ARE the width's slots (tuq_pass_il reads the interleaved transpose)
  for( int ty = 0; ty < 3; ty++ )
  {
    const int16_t *mw = ( ty == 0 || w <= 32 ) ? tabs.m[ty][lw] : nullptr, *mh = ( ty == 0 || h <= 32 ) ? tabs.m[ty][lh] : nullptr;   // DST-7 / DCT-8 exist up to 32
    // width:  slot 0 = M_W with rows k, k+1 interleaved (second inverse pass), slot 1 = M_W^T with rows n, n+1 interleaved (first forward pass)
    // height: slot 0 = M_H with rows k, k+1 interleaved (first inverse pass),   slot 1 = M_H^T plain (second forward pass: 32-bit input)
    if( mw )
      for( int i = threadIdx.x; i < w * w; i += 256 )
      {
        const int k = i >> lw, n = i & ( w - 1 );   // w, h: powers of two
        sMat[( ty * 2 + 0 ) * w * w + ( ( k >> 1 ) * w + n ) * 2 + ( k & 1 )] = mw[i];   // (M[k][n], M[k+1][n]) at pair-row k >> 1, column n
        sMat[( ty * 2 + 1 ) * w * w + ( ( n >> 1 ) * w + k ) * 2 + ( n & 1 )] = mw[i];   // (M[k][n], M[k][n+1]) at pair-row n >> 1, column k
      }
    if( mh && !sq )
      for( int i = threadIdx.x; i < h * h; i += 256 )
      {
        const int k = i >> lh, n = i & ( h - 1 );
        sMat[hBase + ( ty * 2 + 0 ) * h * h + ( ( k >> 1 ) * h + n ) * 2 + ( k & 1 )] = mh[i];
        sMat[hBase + ( ty * 2 + 1 ) * h * h + n * h + k]                              = mh[i];
      }
  }
  __syncthreads();
  const int  jobIdx = blockIdx.x * TUS + sub;
  const bool live   = jobIdx < numJobs;   // LPT <= 64: dead groups are whole waves or idle lane groups that only meet at the barriers
  vtmhip_tu_job j;
  if( live ) j = jobs[jobIdx];
  else { j = jobs[numJobs - 1]; }
  const int      bd = j.bitDepth;
  int           *blk = ldsw + sub * perTu, *tmp = blk + w * h;
  int16_t       *sR  = ( int16_t * ) blk;
  const int16_t *mW = sMat + ( j.typeHor * 2 ) * w * w;
  const int16_t *mH0 = sq ? sMat + ( j.typeVer * 2 ) * w * w : sMat + hBase + ( j.typeVer * 2 ) * h * h;          // M_H, rows k, k+1 interleaved
  const int16_t *mH1 = sq ? sMat + ( j.typeVer * 2 + 1 ) * w * w : sMat + hBase + ( j.typeVer * 2 + 1 ) * h * h;   // M_H^T: pair-interleaved (square) / plain
  const int16_t *resi = resiBase + j.resiOff;
  for( int i = t; i < w * h; i += LPT )
  {
    const int y = i >> lw, x = i & ( w - 1 );
    sR[i] = resi[( long ) y * j.resiStride + x];
  }
  int16_t *dq16 = reinterpret_cast<int16_t *>( tmp );   // dequantised coefficients [k][k2] (after the second forward pass has consumed tmp)
  int16_t *t16  = reinterpret_cast<int16_t *>( blk );   // first inverse pass output [y][i] (after quantisation has consumed blk)
  int     *rec32 = tmp;                                  // reconstructed residual [y][x]
  const int skipW = tr_skip( j.typeHor, w ), skipH = tr_skip( j.typeVer, h );
  long long sumAbs = 0, absSum = 0, sse = 0;
  tuq_sync<LPT>();
  // forward (TrQuant::xT): tmp[k][y] = sum_n blk[y][n] * MT_hor[n][k];  blk[k2][j2] = sum_n tmp[j2][n] * MT_ver[n][k2]
  tuq_pass16<LPT, false>( sR, w, reinterpret_cast<const unsigned *>( mW + w * w ), w, h, w, h, w - skipW, tmp, 1, h + 1, lw + bd + 6 - 15, t );
  tuq_sync<LPT>();
  if( sq ) tuq_pass_il<LPT, false>( tmp, h + 1, 1, reinterpret_cast<const unsigned *>( mH1 ), h, w, h, w - skipW, h - skipH, blk, 1, w, lh + 6, t, &sumAbs );
  else tuq_pass<LPT, false>( tmp, h + 1, 1, mH1, h, h, w, h, w - skipW, h - skipH, blk, 1, w, lh + 6, t, &sumAbs );
  tuq_sync<LPT>();
  if( fwdCoefBase )
  {
    if( live )
      for( int i = t; i < w * h; i += LPT ) fwdCoefBase[j.outOff + i] = blk[i];
  }
  else
  {
    // Quant::quant + Quant::dequant (flat scaling list), in place
    {
      const int       needSqrt = ( lw + lh ) & 1;
      const int       trShift  = 15 - bd - ( ( lw + lh ) >> 1 ) + ( needSqrt ? -1 : 0 );
      const int       qBits    = 14 + j.qpPer + trShift;
      const long long add      = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
      const int       scale    = c_quantScales[needSqrt][j.qpRem], iscale = c_invQuantScales[needSqrt][j.qpRem];
      const int       rightShift = 6 - ( trShift + j.qpPer );
      const int       inBits   = min( 16, 32 + rightShift - 7 );
      const int       inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
      int            *levels   = ( levelsBase && live ) ? levelsBase + j.outOff : nullptr;
      for( int i = t; i < w * h; i += LPT )
      {
        const int       c   = blk[i];
        const long long tt  = ( long long ) abs( c ) * scale;
        const int       mag = ( int ) ( ( tt + add ) >> qBits );
        absSum += mag;
        const int q = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
        if( levels ) levels[i] = q;
        const int qq = min( inMax, max( inMin, q ) );
        int       v;
        if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
        else v = ( int ) ( ( unsigned ) ( qq * iscale ) << ( -rightShift ) );
        dq16[( ( i & ( w - 1 ) ) << lh ) + ( i >> lw )] = ( int16_t ) min( 32767, max( -32768, v ) );   // transposed: the vertical index contiguous
      }
    }
    tuq_sync<LPT>();
    // inverse (TrQuant::xIT): tmp[i][y] = clip( sum_k blk[k][i] * M_ver[k][y] );  rec[y][x] = clip( sum_k tmp[k][y] * M_hor[k][x] )
    tuq_pass16<LPT, true>( dq16, h, reinterpret_cast<const unsigned *>( mH0 ), h - skipH, w, h, w - skipW, h, t16, 1, w, 7, t );
    tuq_sync<LPT>();
    tuq_pass16<LPT, true>( t16, w, reinterpret_cast<const unsigned *>( mW ), w - skipW, h, w, h, w, rec32, w, 1, 20 - bd, t );
    tuq_sync<LPT>();
    {
      int16_t *rec = ( recBase && live ) ? recBase + j.outOff : nullptr;
      for( int i = t; i < w * h; i += LPT )
      {
        const int v = rec32[i];
        if( rec ) rec[i] = ( int16_t ) v;
        const int d = ( int ) resi[( long ) ( i >> lw ) * j.resiStride + ( i & ( w - 1 ) )] - v;   // the residual again, from L2: its LDS copy made room for a third workgroup per CU
        sse += ( long long ) ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
      }
    }
  }
  // reduce the three sums over the LPT lanes of the TU
  if( LPT <= 64 )
  {
#pragma unroll
    for( int o = 32; o > 0; o >>= 1 )
      if( o < LPT )
      {
        sumAbs += __shfl_xor( sumAbs, o, 64 );
        absSum += __shfl_xor( absSum, o, 64 );
        sse += __shfl_xor( sse, o, 64 );
      }
    if( t == 0 && live ) { vtmhip_tu_result r; r.sse = ( uint64_t ) sse; r.sumAbs = ( int32_t ) sumAbs; r.absSum = ( int32_t ) absSum; results[jobIdx] = r; }
  }
  else
  {
    sumAbs = ( long long ) wave_reduce_add_u64( ( unsigned long long ) sumAbs );
    absSum = ( long long ) wave_reduce_add_u64( ( unsigned long long ) absSum );
    sse    = ( long long ) wave_reduce_add_u64( ( unsigned long long ) sse );
    __syncthreads();
    if( ( threadIdx.x & 63 ) == 0 ) { sRed[threadIdx.x >> 6][0] = sumAbs; sRed[threadIdx.x >> 6][1] = absSum; sRed[threadIdx.x >> 6][2] = sse; }
    __syncthreads();
    constexpr int WPT = LPT / 64;   // waves per TU
    if( t == 0 && live )
    {
      vtmhip_tu_result r;
      long long a0 = 0, a1 = 0, a2 = 0;
      for( int k = 0; k < WPT; k++ ) { a0 += sRed[sub * WPT + k][0]; a1 += sRed[sub * WPT + k][1]; a2 += sRed[sub * WPT + k][2]; }
      r.sumAbs = ( int32_t ) a0; r.absSum = ( int32_t ) a1; r.sse = ( uint64_t ) a2;
      results[jobIdx] = r;
    }
  }
}

template<int LPT>
int launch_tu_uni( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int w, int h, int32_t *d_levelsBase, int16_t *d_recBase,
                   vtmhip_tu_result *d_results, const TrTables &tabs, int32_t *d_fwdCoefBase = nullptr )
{
  constexpr int TUS   = 256 / LPT;
  const size_t  perTu = ( size_t ) w * h + ( size_t ) w * ( h + 1 );
  const size_t  lds   = TUS * perTu * sizeof( int ) + ( ( size_t ) ( w > 32 ? 2 : 6 ) * w * w + ( w == h ? 0 : ( size_t ) ( h > 32 ? 2 : 6 ) * h * h ) ) * sizeof( int16_t );
  if( lds > 64 * 1024 )
    VTMHIP_HIP( ctx, hipFuncSetAttribute( reinterpret_cast<const void *>( tu_chain_uni_kernel<LPT> ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds ) );
  VTMHIP_TIME_KERNEL( ctx, "tu_chain_uni_kernel" );
  hipLaunchKernelGGL( tu_chain_uni_kernel<LPT>, dim3( ( n + TUS - 1 ) / TUS ), dim3( 256 ), lds, ctx->stream, d_resiBase, d_jobs, n, tabs, d_levelsBase, d_recBase,
                      d_results, w, h, d_fwdCoefBase );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

// LPT by block size: one lane = 2 x 8 outputs of a transform pass
int launch_tu_uni_sized( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int w, int h, int32_t *d_levelsBase, int16_t *d_recBase,
                         vtmhip_tu_result *d_results, const TrTables &tb, int32_t *d_fwdCoefBase )
{
  const int items = w * h / 16;
  if( items <= 4 ) return launch_tu_uni<4>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  if( items <= 8 ) return launch_tu_uni<8>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  if( items <= 16 ) return launch_tu_uni<16>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  if( items <= 32 ) return launch_tu_uni<32>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  if( items <= 64 ) return launch_tu_uni<64>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  if( items <= 128 ) return launch_tu_uni<128>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
  return launch_tu_uni<256>( ctx, d_resiBase, d_jobs, n, w, h, d_levelsBase, d_recBase, d_results, tb, d_fwdCoefBase );
}

// Transform-skip candidates of a uniform batch (every job typeHor == VTMHIP_TRSKIP): elementwise, LPT = 8 lanes per TU (8 TUs per wave)
__global__ __launch_bounds__( 256 ) void tu_ts_kernel( const int16_t *__restrict__ resiBase, const vtmhip_tu_job *__restrict__ jobs, int numJobs,
                                                      int *__restrict__ levelsBase, int16_t *__restrict__ recBase, vtmhip_tu_result *__restrict__ results, int w, int h )
{
  const int jobIdx = ( blockIdx.x * 256 + threadIdx.x ) >> 3, t = threadIdx.x & 7;
  const bool live  = jobIdx < numJobs;
  const vtmhip_tu_job j = jobs[live ? jobIdx : numJobs - 1];
  const int       lw = ilog2( w );
  const int       qBits = 14 + j.qpPer;
  const long long add   = ( long long ) ( j.isIRAP ? 171 : 85 ) << ( qBits - 9 );
  const int       scale = c_quantScales[0][j.qpRem], iscale = c_invQuantScales[0][j.qpRem];
  const int       rightShift = 6 - j.qpPer;
  const int       inBits = min( 16, 32 + rightShift - 7 );
  const int       inMin = -( 1 << ( inBits - 1 ) ), inMax = ( 1 << ( inBits - 1 ) ) - 1;
  int            *levels = ( levelsBase && live ) ? levelsBase + j.outOff : nullptr;
  int16_t        *rec    = ( recBase && live ) ? recBase + j.outOff : nullptr;
  const int16_t  *resi   = resiBase + j.resiOff;
  long long sumAbs = 0, absSum = 0, sse = 0;
  for( int i = t; i < w * h; i += 8 )
  {
    const int c = resi[( long ) ( i >> lw ) * j.resiStride + ( i & ( w - 1 ) )];
    sumAbs += abs( c );
    const long long tt  = ( long long ) abs( c ) * scale;
    const int       mag = ( int ) ( ( tt + add ) >> qBits );
    absSum += mag;
    const int q = min( 32767, max( -32768, c < 0 ? -mag : mag ) );
    if( levels ) levels[i] = q;
    const int qq = min( inMax, max( inMin, q ) );
    int       v;
    if( rightShift > 0 ) v = ( int ) ( ( unsigned ) ( qq * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
    else v = ( int ) ( ( unsigned ) ( qq * iscale ) << ( -rightShift ) );
    v = ( int ) ( int16_t ) min( 32767, max( -32768, v ) );
    if( rec ) rec[i] = ( int16_t ) v;
    const int d = c - v;
    sse += ( long long ) ( unsigned long long ) ( ( unsigned ) d * ( unsigned ) d );
  }
#pragma unroll
  for( int o = 4; o > 0; o >>= 1 )
  {
    sumAbs += __shfl_xor( sumAbs, o, 64 );
    absSum += __shfl_xor( absSum, o, 64 );
    sse += __shfl_xor( sse, o, 64 );
  }
  if( t == 0 && live ) { vtmhip_tu_result r; r.sse = ( uint64_t ) sse; r.sumAbs = ( int32_t ) sumAbs; r.absSum = ( int32_t ) absSum; results[jobIdx] = r; }
}

bool pow2( int v ) { return v > 0 && ( v & ( v - 1 ) ) == 0; }
int  hlog2( int v ) { int r = 0; while( ( 1 << r ) < v ) r++; return r; }

struct TuClassOf   // {8, 16, 32, 64} x {8, 16, 32, 64} with a real transform: class 4 * log2(w / 8) + log2(h / 8); everything else (4-sample sides, transform skip): the last class
{
  __device__ int operator()( const vtmhip_tu_job &j ) const
  {
    const int w = j.width, h = j.height;
    if( w < 8 || h < 8 || w > 64 || h > 64 || ( w & ( w - 1 ) ) || ( h & ( h - 1 ) ) || j.typeHor == VTMHIP_TRSKIP ) return BUCKET_OTHER;
    return 4 * ( __ffs( w ) - 4 ) + ( __ffs( h ) - 4 );
  }
};

int tu_chain_generic( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int maxWidth, int maxHeight, int32_t *d_levelsBase,
                      int16_t *d_recBase, vtmhip_tu_result *d_results )
{
  const int    mx    = maxWidth > maxHeight ? maxWidth : maxHeight;
  const size_t perTu = ( size_t ) maxWidth * maxHeight + ( size_t ) maxWidth * ( maxHeight + 1 ) + ( ( mx * mx + 1 ) >> 1 ) + ( ( maxWidth * maxHeight + 1 ) >> 1 );
  if( maxWidth * maxHeight <= 256 )
  {
    hipLaunchKernelGGL( tu_chain_kernel<64>, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 4 * perTu * sizeof( int ), ctx->stream, d_resiBase, d_jobs, n,
                        tabs_of( ctx ), d_levelsBase, d_recBase, d_results, maxWidth, maxHeight );
  }
  else
  {
    hipLaunchKernelGGL( tu_chain_kernel<256>, dim3( n ), dim3( 256 ), perTu * sizeof( int ), ctx->stream, d_resiBase, d_jobs, n,
                        tabs_of( ctx ), d_levelsBase, d_recBase, d_results, maxWidth, maxHeight );
  }
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

struct MtsSelectArgs { double scaleSAD, fac; int numTU, numCand, maxCand; bool skip[8]; };

// TrQuant::transformNxN( tu, compID, cQP, &trModes, maxCand ) (TrQuant.cpp:950-1019) per TU: results[c * numTU + t].sumAbs -> test[c * numTU + t]
__global__ __launch_bounds__( 256 ) void mts_select_kernel( const vtmhip_tu_result *__restrict__ results, MtsSelectArgs a, uint8_t *__restrict__ test )
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  if( t >= a.numTU ) return;
  int cost0 = 0, numTests = 0;
  double thr = 0.0;
  for( int c = 0; c < a.numCand; c++ )
  {
    int cost = results[( long ) c * a.numTU + t].sumAbs;
    if( a.skip[c] ) cost = ( int ) ( cost * a.scaleSAD );      // scaleSAD of the transform-skip candidate (:992-1001)
    if( c == 0 ) { cost0 = cost; thr = a.fac * cost0; }
    const bool keep = ( double ) cost <= ( c == 1 ? ( double ) cost0 : thr ) && numTests <= a.maxCand;
    test[( long ) c * a.numTU + t] = keep;
    numTests += keep;
  }
}

}   // namespace

extern "C"
{

int vtmhip_tr_matrix_host( int type, int n, int16_t *out ) { return vtmhip_tr_matrix( type, n, out ) ? VTMHIP_E_INVALID : VTMHIP_OK; }

int vtmhip_fastFwdTrans( vtmhip_ctx *ctx, int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2 )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, src && dst, "null pointer" );
  VTMHIP_REQUIRE( ctx, type >= 0 && type < 3 && pow2( n ) && n >= 2 && n <= 64 && ( type == VTMHIP_DCT2 || ( n >= 4 && n <= 32 ) ),
                  "no such transform (fastFwdTrans slot is nullptr)" );
  VTMHIP_REQUIRE( ctx, line >= 1 && line <= 64 && shift >= 0 && shift < 32 && skipLine >= 0 && skipLine <= line && skipLine2 >= 0 && skipLine2 <= n, "shape" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  const size_t bytes = ( size_t ) n * line * 4;
  st = vtmhip_internal_scratch( ctx, 2 * bytes + 128 );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  const size_t dOff = ( bytes + 63 ) & ~( size_t ) 63;
  memcpy( hp, src, bytes );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, bytes, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( fwd1d_kernel, dim3( ( n * line + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, ( const int * ) dp, ( int * ) ( dp + dOff ),
                      ctx->trTab[type][hlog2( n )], n, shift, line, skipLine, skipLine2 );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + dOff, dp + dOff, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  memcpy( dst, hp + dOff, bytes );
  return VTMHIP_OK;
}

int vtmhip_fastInvTrans( vtmhip_ctx *ctx, int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2,
                         int32_t outputMinimum, int32_t outputMaximum )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, src && dst, "null pointer" );
  VTMHIP_REQUIRE( ctx, type >= 0 && type < 3 && pow2( n ) && n >= 2 && n <= 64 && ( type == VTMHIP_DCT2 || ( n >= 4 && n <= 32 ) ),
                  "no such transform (fastInvTrans slot is nullptr)" );
  VTMHIP_REQUIRE( ctx, line >= 1 && line <= 64 && shift >= 1 && shift < 32 && skipLine >= 0 && skipLine <= line && skipLine2 >= 0 && skipLine2 <= n, "shape" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  const size_t bytes = ( size_t ) n * line * 4;
  st = vtmhip_internal_scratch( ctx, 2 * bytes + 128 );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  const size_t dOff = ( bytes + 63 ) & ~( size_t ) 63;
  memcpy( hp, src, bytes );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, bytes, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( inv1d_kernel, dim3( ( n * line + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, ( const int * ) dp, ( int * ) ( dp + dOff ),
                      ctx->trTab[type][hlog2( n )], n, shift, line, skipLine, skipLine2, outputMinimum, outputMaximum );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + dOff, dp + dOff, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  memcpy( dst, hp + dOff, bytes );
  return VTMHIP_OK;
}

int vtmhip_xT_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, int32_t *d_coefBase, const vtmhip_tr_job *d_jobs, int n, int maxWidth,
                         int maxHeight, int32_t *d_sumAbs )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_coefBase && d_jobs, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 1 && maxWidth <= TB && maxHeight >= 1 && maxHeight <= TB, "maxWidth / maxHeight (max transform size 64)" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  const int    mx  = maxWidth > maxHeight ? maxWidth : maxHeight;
  const size_t lds = ( size_t ) ( maxWidth * maxHeight + maxWidth * ( maxHeight + 1 ) ) * 4 + ( size_t ) mx * mx * 2 + 16;
  hipLaunchKernelGGL( xT_kernel, dim3( n ), dim3( 256 ), lds, ctx->stream, d_resiBase, d_coefBase, d_jobs, tabs_of( ctx ), d_sumAbs );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_xIT_batch_dev( vtmhip_ctx *ctx, const int32_t *d_coefBase, int16_t *d_resiBase, const vtmhip_tr_job *d_jobs, int n, int maxWidth,
                          int maxHeight )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_coefBase && d_jobs, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 1 && maxWidth <= TB && maxHeight >= 1 && maxHeight <= TB, "maxWidth / maxHeight (max transform size 64)" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  const int    mx  = maxWidth > maxHeight ? maxWidth : maxHeight;
  const size_t lds = ( size_t ) ( 2 * maxWidth * maxHeight ) * 4 + ( size_t ) mx * mx * 2 + 16;
  hipLaunchKernelGGL( xIT_kernel, dim3( n ), dim3( 256 ), lds, ctx->stream, d_coefBase, d_resiBase, d_jobs, tabs_of( ctx ) );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_quant_batch_dev( vtmhip_ctx *ctx, const int32_t *d_coefBase, int32_t *d_qBase, int32_t *d_deltaUBase, const vtmhip_quant_job *d_jobs, int n,
                            int32_t *d_absSum )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_coefBase && d_qBase && d_jobs && d_absSum, "null pointer" );
  hipLaunchKernelGGL( quant_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_coefBase, d_qBase, d_deltaUBase, d_jobs, d_absSum );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_dequant_batch_dev( vtmhip_ctx *ctx, const int32_t *d_qBase, int32_t *d_coefBase, const vtmhip_quant_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_coefBase && d_qBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( dequant_kernel, dim3( n ), dim3( 256 ), 0, ctx->stream, d_qBase, d_coefBase, d_jobs );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_tu_chain_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int maxWidth, int maxHeight,
                               int uniformSize, int32_t *d_levelsBase, int16_t *d_recBase, vtmhip_tu_result *d_results )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_jobs && d_results, "null pointer" );
  VTMHIP_REQUIRE( ctx, maxWidth >= 2 && maxWidth <= TB && maxHeight >= 2 && maxHeight <= TB, "maxWidth / maxHeight: 2..64 (2-D transforms)" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  // a mixed batch with enough TUs: bucket by shape on the device (bucket.hpp), the register-blocked kernel per {8,16,32,64} x {8,16,32,64} class, the
  // generic kernel for the rest (4-sample sides, transform skip).  VTMHIP_TU_BUCKET=0 keeps the generic kernel for the whole batch.
  static const bool bucket = !( getenv( "VTMHIP_TU_BUCKET" ) && atoi( getenv( "VTMHIP_TU_BUCKET" ) ) == 0 );
  if( bucket && !uniformSize && n >= 256 && maxWidth >= 8 && maxHeight >= 8 && bucket_allowed( ctx ) )
  {
    BucketPlan plan;
    st = bucket_begin<vtmhip_tu_job, vtmhip_tu_result>( ctx, d_jobs, n, TuClassOf(), plan );
    if( st ) return st;
    for( int c = 0; c < BUCKET_CLASSES; c++ )
    {
      if( !plan.count[c] ) continue;
      const vtmhip_tu_job *jobs = ( const vtmhip_tu_job * ) plan.d_jobs + plan.offset[c];
      vtmhip_tu_result    *res  = ( vtmhip_tu_result * ) plan.d_results + plan.offset[c];
      if( c < 16 ) st = launch_tu_uni_sized( ctx, d_resiBase, jobs, plan.count[c], 8 << ( c >> 2 ), 8 << ( c & 3 ), d_levelsBase, d_recBase, res, tabs_of( ctx ), nullptr );
      else         st = tu_chain_generic( ctx, d_resiBase, jobs, plan.count[c], maxWidth, maxHeight, d_levelsBase, d_recBase, res );
      if( st ) return st;
    }
    return bucket_finish<vtmhip_tu_result>( ctx, plan, n, d_results );
  }
  if( uniformSize && maxWidth * maxHeight <= 32 && maxWidth >= 4 && maxHeight >= 4 )
  {
    // caller's promise: every TU is exactly maxWidth x maxHeight (4x4, 8x4 or 4x8) with a real transform -> one lane per TU
    if( maxWidth == 4 && maxHeight == 4 ) return launch_tu_lane<4, 4>( ctx, d_resiBase, d_jobs, n, d_levelsBase, d_recBase, d_results, tabs_of( ctx ) );
    if( maxWidth == 8 ) return launch_tu_lane<8, 4>( ctx, d_resiBase, d_jobs, n, d_levelsBase, d_recBase, d_results, tabs_of( ctx ) );
    return launch_tu_lane<4, 8>( ctx, d_resiBase, d_jobs, n, d_levelsBase, d_recBase, d_results, tabs_of( ctx ) );
  }
  if( uniformSize && maxWidth >= 8 && maxHeight >= 8 )
  {
    VTMHIP_REQUIRE( ctx, ( maxWidth & ( maxWidth - 1 ) ) == 0 && ( maxHeight & ( maxHeight - 1 ) ) == 0, "uniformSize: width / height must be powers of two (TU sizes are)" );
    // caller's promise: every TU is exactly maxWidth x maxHeight -> register-blocked kernel, LPT lanes per TU
    return launch_tu_uni_sized( ctx, d_resiBase, d_jobs, n, maxWidth, maxHeight, d_levelsBase, d_recBase, d_results, tabs_of( ctx ), nullptr );
  }
  return tu_chain_generic( ctx, d_resiBase, d_jobs, n, maxWidth, maxHeight, d_levelsBase, d_recBase, d_results );
}

int vtmhip_tu_ts_chain_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int width, int height,
                                  int32_t *d_levelsBase, int16_t *d_recBase, vtmhip_tu_result *d_results )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_jobs && d_results, "null pointer" );
  VTMHIP_REQUIRE( ctx, pow2( width ) && pow2( height ) && width >= 4 && height >= 4 && width <= 32 && height <= 32, "transform skip: 4..32 (log2MaxTransformSkipBlockSize)" );
  VTMHIP_TIME_KERNEL( ctx, "tu_ts_kernel" );
  hipLaunchKernelGGL( tu_ts_kernel, dim3( ( n + 31 ) / 32 ), dim3( 256 ), 0, ctx->stream, d_resiBase, d_jobs, n, d_levelsBase, d_recBase, d_results, width, height );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

// The same selection from RAW sums: mtsIdx[i] is the candidate's tu.mtsIdx (MTS_DCT2_DCT2 = 0, MTS_SKIP = 1, MTS_DST7_DST7 = 2 ...); the sum of a
// transform-skip candidate (sum |residual|) is scaled here as the reference scales it (:992-1001): int( sumAbs * scaleSAD ) with scaleSAD =
// 2^trShift, times 1/1.414213562 when log2(width) + log2(height) is odd.
int vtmhip_mts_select2( const int32_t *sumAbs, const uint8_t *mtsIdx, int numCand, int width, int height, int bitDepth, int maxLog2TrDynamicRange, int maxCand,
                        uint8_t *test )
{
  if( !sumAbs || !mtsIdx || !test || numCand < 1 || numCand > 16 || width < 1 || height < 1 ) return VTMHIP_E_INVALID;
  int lw = 0, lh = 0;
  while( ( 2 << lw ) <= width ) lw++;
  while( ( 2 << lh ) <= height ) lh++;
  int32_t scaled[16];
  for( int i = 0; i < numCand; i++ )
  {
    scaled[i] = sumAbs[i];
    if( mtsIdx[i] == 1 )
    {
      double scaleSAD = 1.0;
      if( ( lw + lh ) & 1 ) scaleSAD = 1.0 / 1.414213562;
      const int trShift = maxLog2TrDynamicRange - bitDepth - ( ( lw + lh ) >> 1 );
      scaleSAD *= pow( 2, trShift );
      scaled[i] = ( int ) ( sumAbs[i] * scaleSAD );
    }
  }
  return vtmhip_mts_select( scaled, numCand, width, height, maxCand, test );
}

// the same rule for every TU of a level, one thread per TU (fp64 compare as the reference: facBB[] x the first candidate's cost)
int vtmhip_mts_select_batch_dev( vtmhip_ctx *ctx, const vtmhip_tu_result *d_results, int numTU, int numCand, const uint8_t *mtsIdx, int width, int height,
                                 int bitDepth, int maxLog2TrDynamicRange, int maxCand, uint8_t *d_test )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, numTU >= 0 && numCand >= 1 && numCand <= 8 && width >= 4 && height >= 4 && mtsIdx, "numTU / numCand / size" );
  if( numTU == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_results && d_test, "null pointer" );
  int lw = 0, lh = 0;
  while( ( 2 << lw ) <= width ) lw++;
  while( ( 2 << lh ) <= height ) lh++;
  MtsSelectArgs a;
  for( int i = 0; i < 8; i++ ) a.skip[i] = i < numCand && mtsIdx[i] == 1;
  a.scaleSAD = ( ( lw + lh ) & 1 ) ? 1.0 / 1.414213562 : 1.0;
  a.scaleSAD *= pow( 2, maxLog2TrDynamicRange - bitDepth - ( ( lw + lh ) >> 1 ) );
  static const double facBB[] = { 1.2, 1.3, 1.3, 1.4, 1.5 };
  const int lg = lw > lh ? lw : lh;
  a.fac = facBB[lg - 2 > 0 ? ( lg - 2 > 4 ? 4 : lg - 2 ) : 0];
  a.numTU = numTU; a.numCand = numCand; a.maxCand = maxCand;
  hipLaunchKernelGGL( mts_select_kernel, dim3( ( numTU + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_results, a, d_test );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

// MTS candidate pre-selection thresholds (TrQuant::transformNxN( ..., trModes, maxCand ), TrQuant.cpp:950-1019): host arithmetic (fp64).
// sumAbs[i]: sum |coef| of candidate i in trModes order, as vtmhip_xT_batch_dev returns it (a transform-skip candidate already scaled by
// the caller, :992-1001).  The reference compares the candidate at LIST POSITION 1 against the unscaled threshold (:1012).
// test[i] = 1 when the candidate survives.
int vtmhip_mts_select( const int32_t *sumAbs, int numCand, int width, int height, int maxCand, uint8_t *test )
{
  if( !sumAbs || !test || numCand < 1 || width < 1 || height < 1 ) return VTMHIP_E_INVALID;
  static const double facBB[] = { 1.2, 1.3, 1.3, 1.4, 1.5 };
  const int    mx  = width > height ? width : height;
  int          lg  = 0;
  while( ( 2 << lg ) <= mx ) lg++;
  const int    fi  = lg - 2 > 0 ? ( lg - 2 > 4 ? 4 : lg - 2 ) : 0;
  const double thr = facBB[fi] * sumAbs[0], thrTS = sumAbs[0];
  int          numTests = 0;
  for( int i = 0; i < numCand; i++ )
  {
    const bool t = sumAbs[i] <= ( i == 1 ? thrTS : thr ) && numTests <= maxCand;
    test[i]      = t;
    numTests += t;
  }
  return VTMHIP_OK;
}

}   // extern "C"


extern "C" int vtmhip_xT_uniform_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int width, int height, int32_t *d_coefBase,
                                            vtmhip_tu_result *d_results )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, d_resiBase && d_jobs && d_coefBase && d_results, "null pointer" );
  VTMHIP_REQUIRE( ctx, width >= 8 && width <= TB && height >= 8 && height <= TB && pow2( width ) && pow2( height ), "width / height: powers of two 8..64" );
  int st = ensure_tables( ctx );
  if( st ) return st;
  return launch_tu_uni_sized( ctx, d_resiBase, d_jobs, n, width, height, nullptr, nullptr, d_results, tabs_of( ctx ), d_coefBase );
}


// =====================================================================================================================
// LFNST kernels (TrQuant::fwdLfnstNxN / invLfnstNxN, TrQuant.cpp:233-311): a 16 x 16 or 16 x 48 int8 matrix-vector product per TU.
// One wave per job.  Forward: lane = part * 16 + j, four lanes share output j (12 or 4 of the 48 / 16 products each).
// Inverse: lane j < trSize sums its column over the first zeroOutSize rows.  The matrices come from the caller (vtmhip_lfnst_set_tables).
// =====================================================================================================================
namespace
{
__global__ __launch_bounds__( 256 ) void lfnst_kernel( const int8_t *__restrict__ tab, const int *__restrict__ srcBase, int *__restrict__ dstBase,
                                                      const vtmhip_lfnst_job *__restrict__ jobs, int n )
{
  const int lane = threadIdx.x & 63;
  const int job  = blockIdx.x * 4 + ( threadIdx.x >> 6 );
  if( job >= n ) return;
  const vtmhip_lfnst_job j = jobs[job];
  const int     trSize = j.size > 4 ? 48 : 16, zo = j.zeroOutSize;
  const int8_t *M   = j.size > 4 ? tab + ( ( j.mode * 2 + j.index ) * 16 ) * 48 : tab + 4 * 2 * 16 * 48 + ( ( j.mode * 2 + j.index ) * 16 ) * 16;
  const int    *src = srcBase + j.srcOff;
  int          *dst = dstBase + j.dstOff;
  if( !j.inverse )
  {
    const int o = lane & 15, part = lane >> 4, per = trSize >> 2;
    int       acc = 0;
    if( o < zo )
      for( int i = part * per; i < ( part + 1 ) * per; i++ ) acc += src[i] * ( int ) M[o * trSize + i];
    acc += __shfl_xor( acc, 16, 64 );
    acc += __shfl_xor( acc, 32, 64 );
    if( lane < trSize ) dst[lane] = lane < zo ? ( acc + 64 ) >> 7 : 0;   // lanes 16..47 of the 8x8 case write the zeroed tail
    return;
  }
  if( lane < trSize )
  {
    int acc = 0;
    for( int i = 0; i < zo; i++ ) acc += src[i] * ( int ) M[i * trSize + lane];
    dst[lane] = min( 32767, max( -32768, ( acc + 64 ) >> 7 ) );
  }
}

// TrQuant::xFwdLfnst / xInvLfnst (TrQuant.cpp:340-420, 422-527) on a TU's coefficient block IN PLACE: the low-frequency region is gathered
// (row-major, or transposed for the intra modes past the diagonal), multiplied with the 16 x 16 / 16 x 48 core matrix and written back along the
// coefficient scan (forward) -- or the reverse.  The scan positions are the first 16 / 48 of the diagonal scan in 4x4 coefficient groups
// (g_scanOrder[SCAN_GROUPED_4x4][SCAN_DIAG] / g_coefTopLeftDiagScan8x8): generated here, checked against the reference's tables in the tests.
__device__ __forceinline__ int lfnst_scan_pos( int k, int width, bool sb8 )   // k-th scan position -> index x + y * width
{
  // inside a 4x4 group: up-right diagonals, each from bottom-left to top-right; groups of the 8x8 region in the same diagonal order: (0,0), (0,1), (1,0), (1,1)
  const int g = k >> 4, i = k & 15;
  const int dx[16] = { 0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 }, dy[16] = { 0, 1, 0, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 3, 2, 3 };
  const int gx = sb8 ? ( g == 2 || g == 3 ) : 0, gy = sb8 ? ( g == 1 || g == 3 ) : 0;
  return ( gx * 4 + dx[i] ) + ( gy * 4 + dy[i] ) * width;
}

__global__ __launch_bounds__( 256 ) void lfnst_tu_kernel( const int8_t *__restrict__ tab, int *__restrict__ coefBase, const vtmhip_lfnst_tu_job *__restrict__ jobs, int n )
{
  __shared__ int sIn[4][48], sOut[4][48];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int job  = blockIdx.x * 4 + wv;
  if( job >= n ) return;
  const vtmhip_lfnst_tu_job j = jobs[job];
  const int  w = j.width, h = j.height;
  const bool sb8 = w >= 8 && h >= 8;
  const int  sbSize = sb8 ? 8 : 4, trSize = sb8 ? 48 : 16, zo = ( ( w == 4 && h == 4 ) || ( w == 8 && h == 8 ) ) ? 8 : 16;
  const int8_t *M = sb8 ? tab + ( ( j.mode * 2 + j.index ) * 16 ) * 48 : tab + 4 * 2 * 16 * 48 + ( ( j.mode * 2 + j.index ) * 16 ) * 16;
  int *coef = coefBase + j.coefOff;
  // region position r (0 .. trSize-1) of the LFNST input / output vector -> (x, y) in the block
  auto region_xy = [&]( int r, int &x, int &y ) {
    int row, col;
    if( sbSize == 4 ) { row = r >> 2; col = r & 3; }
    else if( r < 32 ) { row = r >> 3; col = r & 7; }
    else { row = 4 + ( ( r - 32 ) >> 2 ); col = ( r - 32 ) & 3; }
    // transposed gather: lfnstTemp[col' * sbSize(or 4) ...]: the vector index of block position (x, y) is that of (y, x) in the untransposed layout
    if( j.transpose ) { x = row; y = col; } else { x = col; y = row; }
  };
  if( !j.inverse )
  {
    if( lane < trSize ) { int x, y; region_xy( lane, x, y ); sIn[wv][lane] = coef[y * w + x]; }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
    if( lane < trSize )
    {
      int acc = 0;
      if( lane < zo ) { for( int i = 0; i < trSize; i++ ) acc += sIn[wv][i] * ( int ) M[lane * trSize + i]; acc = ( acc + 64 ) >> 7; }
      sOut[wv][lane] = lane < zo ? acc : 0;
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
    if( lane < trSize ) coef[lfnst_scan_pos( lane, w, sb8 )] = sOut[wv][lane];
  }
  else
  {
    if( lane < 16 ) sIn[wv][lane] = coef[lfnst_scan_pos( lane, w, sb8 )];
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
    if( lane < trSize )
    {
      int acc = 0;
      for( int i = 0; i < zo; i++ ) acc += sIn[wv][i] * ( int ) M[i * trSize + lane];
      int x, y; region_xy( lane, x, y );
      coef[y * w + x] = min( 32767, max( -32768, ( acc + 64 ) >> 7 ) );
    }
  }
}
}   // namespace

extern "C"
{

int vtmhip_lfnst_tu_batch_dev( vtmhip_ctx *ctx, int32_t *d_coefBase, const vtmhip_lfnst_tu_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, ctx->lfnstTab, "vtmhip_lfnst_set_tables has not been called" );
  VTMHIP_REQUIRE( ctx, d_coefBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( lfnst_tu_kernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, ctx->lfnstTab, d_coefBase, d_jobs, n );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

int vtmhip_lfnst_scan_host( int width, int height, int32_t *pos48 )
{
  if( !pos48 || width < 4 || height < 4 ) return VTMHIP_E_INVALID;
  const bool sb8 = width >= 8 && height >= 8;
  static const int dx[16] = { 0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 }, dy[16] = { 0, 1, 0, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 3, 2, 3 };
  for( int k = 0; k < ( sb8 ? 48 : 16 ); k++ )
  {
    const int g = k >> 4, i = k & 15, gx = sb8 ? ( g == 2 || g == 3 ) : 0, gy = sb8 ? ( g == 1 || g == 3 ) : 0;
    pos48[k] = ( gx * 4 + dx[i] ) + ( gy * 4 + dy[i] ) * width;
  }
  return VTMHIP_OK;
}

int vtmhip_lfnst_set_tables( vtmhip_ctx *ctx, const int8_t *lfnst8x8, const int8_t *lfnst4x4 )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, lfnst8x8 && lfnst4x4, "null pointer" );
  const size_t n8 = 4 * 2 * 16 * 48, n4 = 4 * 2 * 16 * 16;
  if( !ctx->lfnstTab ) VTMHIP_HIP( ctx, hipMalloc( ( void ** ) &ctx->lfnstTab, n8 + n4 ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( ctx->lfnstTab, lfnst8x8, n8, hipMemcpyHostToDevice, ctx->stream ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( ctx->lfnstTab + n8, lfnst4x4, n4, hipMemcpyHostToDevice, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );   // the host arrays may go away
  return VTMHIP_OK;
}

int vtmhip_lfnst_batch_dev( vtmhip_ctx *ctx, const int32_t *d_srcBase, int32_t *d_dstBase, const vtmhip_lfnst_job *d_jobs, int n )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, n >= 0, "n" );
  if( n == 0 ) return VTMHIP_OK;
  VTMHIP_REQUIRE( ctx, ctx->lfnstTab, "vtmhip_lfnst_set_tables has not been called" );
  VTMHIP_REQUIRE( ctx, d_srcBase && d_dstBase && d_jobs, "null pointer" );
  hipLaunchKernelGGL( lfnst_kernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, ctx->lfnstTab, d_srcBase, d_dstBase, d_jobs, n );
  VTMHIP_LAUNCHED( ctx );
  return VTMHIP_OK;
}

static int lfnst_single( vtmhip_ctx *ctx, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize, int inverse )
{
  VTMHIP_CHECK_CTX( ctx );
  VTMHIP_REQUIRE( ctx, src && dst, "null pointer" );
  VTMHIP_REQUIRE( ctx, ctx->lfnstTab, "vtmhip_lfnst_set_tables has not been called" );
  VTMHIP_REQUIRE( ctx, mode >= 0 && mode < 4 && index >= 0 && index < 2 && ( size == 4 || size == 8 ) && ( zeroOutSize == 8 || zeroOutSize == 16 ), "mode / index / size / zeroOutSize" );
  const int trSize = size > 4 ? 48 : 16, nIn = inverse ? zeroOutSize : trSize;
  int st = vtmhip_internal_scratch( ctx, 1024 );
  if( st ) return st;
  char *hp = ( char * ) ctx->pinned, *dp = ( char * ) ctx->scratch;
  memcpy( hp, src, sizeof( int32_t ) * nIn );
  vtmhip_lfnst_job j;
  memset( &j, 0, sizeof( j ) );
  j.srcOff = 0; j.dstOff = 64; j.mode = ( uint8_t ) mode; j.index = ( uint8_t ) index; j.size = ( uint8_t ) size; j.zeroOutSize = ( uint8_t ) zeroOutSize; j.inverse = ( uint8_t ) inverse;
  memcpy( hp + 512, &j, sizeof( j ) );
  VTMHIP_HIP( ctx, hipMemcpyAsync( dp, hp, 576, hipMemcpyHostToDevice, ctx->stream ) );
  hipLaunchKernelGGL( lfnst_kernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ctx->lfnstTab, ( const int * ) dp, ( int * ) dp, ( const vtmhip_lfnst_job * ) ( dp + 512 ), 1 );
  VTMHIP_LAUNCHED( ctx );
  VTMHIP_HIP( ctx, hipMemcpyAsync( hp + 256, dp + 256, sizeof( int32_t ) * trSize, hipMemcpyDeviceToHost, ctx->stream ) );
  VTMHIP_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
  memcpy( dst, hp + 256, sizeof( int32_t ) * trSize );
  return VTMHIP_OK;
}

int vtmhip_fwdLfnstNxN( vtmhip_ctx *ctx, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize )
{
  return lfnst_single( ctx, src, dst, mode, index, size, zeroOutSize, 0 );
}
int vtmhip_invLfnstNxN( vtmhip_ctx *ctx, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize )
{
  return lfnst_single( ctx, src, dst, mode, index, size, zeroOutSize, 1 );
}

}   // extern "C"
