/*
 * vtmhip.h -- C ABI of libvtmhip.so: MI355X (gfx950) implementation of the VTM 9.3 inter motion-estimation +
 * transform/quantisation hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Two layers:
 *   (1) POINTER-SURFACE calls (host pointers in, host results out): one entry point per slot of the reference's
 *       function-pointer dispatch surface, same argument meaning, so a `initRdCostHIP()`-style installer can
 *       trampoline into them.  They stage the operands to the device, run the SAME kernels as
 *       layer 2 with a batch of one and copy the result back: correct drop-ins, not fast ones (a launch costs more
 *       than the whole CPU call, SURVEY.md section 7 hard part 1).
 *   (2) BATCHED DEVICE calls (`*_dev`): operands already resident in HBM (device pointers), many jobs per launch,
 *       asynchronous on the context's stream.  This is the product path; the hooks B1..B10 of SURVEY.md Appendix B
 *       feed it.
 *
 * Reference interfaces replaced (paths relative to /root/reference/source/Lib):
 *   DistParam::distFunc / RdCost::m_afpDistortFunc[DF_SAD*|DF_HAD*|DF_SSE*]   CommonLib/RdCost.h:60,67-105,113; RdCost.cpp:125-217
 *   RdCost::getCostOfVectorWithPredictor                                       CommonLib/RdCost.h:301-315
 *   InterSearch::xTZSearch / xPatternSearch / xPatternSearchFracDIF            EncoderLib/InterSearch.cpp:3640-3976, 3566-3608, 4284-4339
 *   InterpolationFilter::m_filterHor/m_filterVer/m_filterCopy                  CommonLib/InterpolationFilter.h:93-96
 *   fastFwdTrans / fastInvTrans, TrQuant::xT / xIT                             CommonLib/TrQuant.cpp:69-81, 776-923
 *   Quant::quant / Quant::dequant (flat scaling list)                          CommonLib/Quant.cpp:955-1038, 357-482
 *   AffineGradientSearch::m_HorizontalSobelFilter/m_VerticalSobelFilter/m_EqualCoeffComputer  CommonLib/AffineGradientSearch.h:50-54
 *   PelBufferOps::removeHighFreq / addAvg                                      CommonLib/Buffer.h:64-81
 *
 * Types: Pel = int16_t, TCoeff = int32_t, Distortion = uint64_t (CommonLib/TypeDef.h:259-270).
 * Every function returns VTMHIP_OK (0) or a negative VTMHIP_E_* code and never throws; a C++ trampoline turns a
 * non-zero status into the reference's THROW (TypeDef.h:1065-1081).
 *
 * Threading: a context is NOT re-entrant.  It owns one staging area (pointer surface), one workspace and one "current stream"; use one
 * context per host thread (per encoder stack, as the reference keeps one RdCost / InterSearch / TrQuant per thread) and, in a multi-GPU
 * process, one per device.  Every entry point makes the context's device the calling thread's current HIP device.
 */
#ifndef VTMHIP_H
#define VTMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTMHIP_ABI_VERSION 6

enum
{
  VTMHIP_OK            = 0,
  VTMHIP_E_INVALID     = -1,   /* bad argument (the reference would THROW "Unsupported size" / CHECK) */
  VTMHIP_E_NODEVICE    = -2,   /* no HIP device / device index out of range */
  VTMHIP_E_HIP         = -3,   /* a HIP runtime call failed; see vtmhip_last_error() */
  VTMHIP_E_NOMEM       = -4,
  VTMHIP_E_UNSUPPORTED = -5    /* caller must keep its CPU path (applyWeight, useMR, step != 1, explicit scaling lists, ...) */
};

enum { VTMHIP_DIST_SAD = 0, VTMHIP_DIST_SATD = 1, VTMHIP_DIST_SSE = 2 };
enum { VTMHIP_DCT2 = 0, VTMHIP_DCT8 = 1, VTMHIP_DST7 = 2,      /* TransType, CommonLib/TypeDef.h */
       VTMHIP_TRSKIP = 3 };                                    /* fused chain only: the MTS_SKIP candidate (xTransformSkip / xITransformSkip, no transform) */

typedef struct vtmhip_ctx vtmhip_ctx;

/* ---- context ------------------------------------------------------------------------------------------------ */
int         vtmhip_abi_version( void );
int         vtmhip_struct_size( int which );   /* sizeof() of the job/result structs below, in declaration order (0 = vtmhip_dist_job ...): lets a foreign-language binding verify its layout */
int         vtmhip_device_count( int *count );
int         vtmhip_create( int device, vtmhip_ctx **ctx );              /* one context per encoder stack / per rank */
int         vtmhip_destroy( vtmhip_ctx *ctx );
int         vtmhip_set_stream( vtmhip_ctx *ctx, void *hipStream );     /* launch on the caller's hipStream_t; NULL = HIP's default stream (torch's default) */
int         vtmhip_use_own_stream( vtmhip_ctx *ctx );                  /* back to the non-blocking stream the context created (the initial state) */
int         vtmhip_sync( vtmhip_ctx *ctx );                            /* waits for the context's stream */
const char *vtmhip_last_error( vtmhip_ctx *ctx );
const char *vtmhip_status_string( int status );

/* device memory helpers for hosts without their own allocator (the C++ encoder); torch hosts pass data_ptr() */
int vtmhip_dev_alloc( vtmhip_ctx *ctx, size_t bytes, void **devPtr );
int vtmhip_dev_free( vtmhip_ctx *ctx, void *devPtr );
int vtmhip_host_alloc( vtmhip_ctx *ctx, size_t bytes, void **hostPtr );         /* page-locked host memory: job / result slots a hook re-uses for every call (h2d / d2h from it do not stage) */
int vtmhip_host_free( vtmhip_ctx *ctx, void *hostPtr );
int vtmhip_h2d( vtmhip_ctx *ctx, void *dev, const void *host, size_t bytes );   /* asynchronous on the stream */
int vtmhip_d2h( vtmhip_ctx *ctx, void *host, const void *dev, size_t bytes );   /* synchronises before returning */
/* stream timing with HIP events (bench.py uses this around the timed region) */
int vtmhip_timer_start( vtmhip_ctx *ctx );
int vtmhip_timer_stop_ms( vtmhip_ctx *ctx, float *ms );   /* synchronises */

/* per-kernel launch timing: while enabled, every launch of the library's main kernels (tz_search_kernel, full_search_kernel, full_search_sq_kernel,
 * frac_search_sq_kernel, frac_search_kernel, motion_comp_kernel, tu_chain_uni_kernel, tu_ts_kernel, dist_uniform_kernel, satd8_grid_kernel) is
 * bracketed by HIP events on the stream it is launched on.  vtmhip_kernel_timing( ctx, 1 ) clears the record and starts, ( ctx, 0 ) stops;
 * _read synchronises the device and returns the summed duration and the number of launches of one kernel since the start. */
int vtmhip_kernel_timing( vtmhip_ctx *ctx, int enable );
int vtmhip_kernel_timing_read( vtmhip_ctx *ctx, const char *kernel, double *totalMs, int *launches );

/* ================================================================================================================
 * (1) POINTER-SURFACE CALLS -- host pointers
 * ============================================================================================================== */

/* DistParam::distFunc for DF_SAD* / DF_HAD* / DF_SSE* (RdCost.cpp:493-1003, 2819-2934, 1783-2133).
 * org/cur: top-left sample of the W x H blocks, strides in samples.  subShift as DistParam::subShift (SAD only).
 * Any int16 sample values are accepted (bi-pred ME passes 2*org - pred).
 * The reference guards (applyWeight, useMR, step != 1) stay in the trampoline: fall back to the scalar function. */
int vtmhip_xGetSAD( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, int subShift,
                    uint64_t *dist );
int vtmhip_xGetHADs( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height,
                     uint64_t *dist );
int vtmhip_xGetSSE( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height,
                    uint64_t *dist );
/* DistParam::distFunc for DF_SAD_WITH_MASK (RdCost::xGetSADwMask, RdCost.cpp:3513-3549; set up by the mask overload of setDistParam :3488-3511;
 * GEO merge estimation EncCu.cpp:2930-2960): sum |org - cur| * mask, the mask walked with stepX (+1 / -1) per sample and
 * maskStride * (1 << subShift) + maskStride2 per row, as the scalar reference does. */
int vtmhip_xGetSADwMask( vtmhip_ctx *ctx, const int16_t *org, int orgStride, const int16_t *cur, int curStride, int width, int height, int subShift,
                         const int16_t *mask, int maskStride, int stepX, int maskStride2, uint64_t *dist );


/* InterpolationFilter::m_filterHor / m_filterVer [tapIdx][isFirst][isLast] and m_filterCopy[isFirst][isLast]
 * (InterpolationFilter.h:93-95; filter<> InterpolationFilter.cpp:548-651, filterCopy<> :398-525).  `src` points at the
 * OUTPUT-aligned sample, as in the reference: the callee reads (taps/2 - 1) samples before it along the filter direction.
 * taps: 8, 4 or 2; coeff: `taps` filter taps; clipMin/clipMax/bitDepth: ClpRng::min/max/bd. */
int vtmhip_filterHor( vtmhip_ctx *ctx, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width,
                      int height, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR );
int vtmhip_filterVer( vtmhip_ctx *ctx, int taps, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width,
                      int height, const int16_t *coeff, int bitDepth, int clipMin, int clipMax, int biMCForDMVR );
int vtmhip_filterCopy( vtmhip_ctx *ctx, int isFirst, int isLast, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height,
                       int bitDepth, int clipMin, int clipMax, int biMCForDMVR );

/* fastFwdTrans[type][log2(n)-1] / fastInvTrans[type][log2(n)-1] (TrQuant.cpp:69-81; typedefs TrQuant.h:53-54): 1-D transform of `line`
 * rows, TRANSPOSED output dst[k*line + j] (forward) / dst[i*n + j] (inverse), zero-out through skipLine / skipLine2.
 * (type, n) pairs whose table slot is nullptr in the reference return VTMHIP_E_INVALID. */
int vtmhip_fastFwdTrans( vtmhip_ctx *ctx, int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2 );
int vtmhip_fastInvTrans( vtmhip_ctx *ctx, int type, int n, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2,
                         int32_t outputMinimum, int32_t outputMaximum );
/* LFNST kernels: TrQuant::fwdLfnstNxN / invLfnstNxN (TrQuant.cpp:233-311; called by xFwdLfnst / xInvLfnst :313-527, which gather / scatter the
 * low-frequency region on the host).  The trained core matrices g_lfnst8x8[4][2][16][48] and g_lfnst4x4[4][2][16][16] (Rom.h:132-133, int8) are the
 * CALLER's data: the integration hands the reference's own arrays over once per context; nothing of them is compiled into this library.
 * forward: dst[j] = ( sum_i src[i] * M[j][i] + 64 ) >> 7 for j < zeroOutSize, zeros up to trSize (16 / 48);
 * inverse: dst[j] = clip( ( sum_{i < zeroOutSize} src[i] * M[i][j] + 64 ) >> 7, +-2^15 ) for j < trSize.   size: 4 or 8 (the `size > 4` switch). */
int vtmhip_lfnst_set_tables( vtmhip_ctx *ctx, const int8_t *lfnst8x8, const int8_t *lfnst4x4 );
int vtmhip_fwdLfnstNxN( vtmhip_ctx *ctx, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize );
int vtmhip_invLfnstNxN( vtmhip_ctx *ctx, const int32_t *src, int32_t *dst, int mode, int index, int size, int zeroOutSize );
typedef struct
{
  int64_t srcOff, dstOff;       /* coefficients inside d_srcBase / d_dstBase: forward reads trSize and writes trSize, inverse reads zeroOutSize and writes trSize */
  uint8_t mode, index;          /* g_lfnstLut[intraMode] (0..3), lfnstIdx - 1 (0..1) */
  uint8_t size;                 /* 4: the 16 x 16 matrices, 8: the 16 x 48 ones */
  uint8_t zeroOutSize;          /* 8 or 16 */
  uint8_t inverse, pad0, pad1, pad2;
} vtmhip_lfnst_job;
int vtmhip_lfnst_batch_dev( vtmhip_ctx *ctx, const int32_t *d_srcBase, int32_t *d_dstBase, const vtmhip_lfnst_job *d_jobs, int n );

/* The whole secondary transform of a TU, TrQuant::xFwdLfnst / xInvLfnst (TrQuant.cpp:340-527), IN PLACE on its W x H coefficient block (stride W):
 * gather of the low-frequency region (transposed for the intra modes past the diagonal, getTransposeFlag :334-338), the core multiply above, scatter along
 * the coefficient scan (forward: 16 / 48 positions; inverse: gather 16 scan positions, write the 16 / 48 region samples).  The trampoline derives
 * mode = g_lfnstLut[ getLFNSTIntraMode( PU::getWideAngle( ... ) ) ] and transpose from the PU exactly as the reference does (:350-366, 438-454). */
typedef struct
{
  int64_t coefOff;          /* the TU's coefficients inside d_coefBase */
  int16_t width, height;    /* 4..64 */
  uint8_t mode, index;      /* g_lfnstLut[intraMode] (0..3), lfnstIdx - 1 (0..1) */
  uint8_t transpose, inverse;
} vtmhip_lfnst_tu_job;
int vtmhip_lfnst_tu_batch_dev( vtmhip_ctx *ctx, int32_t *d_coefBase, const vtmhip_lfnst_tu_job *d_jobs, int n );
/* host-only: the scan positions (x + y * width) the calls above use, 16 (4-wide / 4-high TUs) or 48 entries */
int vtmhip_lfnst_scan_host( int width, int height, int32_t *pos48 );

/* n x n forward core matrix g_trCore<type>P<n>[TRANSFORM_FORWARD] (Rom.h:115-130), row-major int16; host-only helper */
int vtmhip_tr_matrix_host( int type, int n, int16_t *out );
/* MTS candidate pre-selection thresholds of TrQuant::transformNxN( tu, compID, cQP, &trModes, maxCand ) (TrQuant.cpp:950-1019); host-only */
int vtmhip_mts_select( const int32_t *sumAbs, int numCand, int width, int height, int maxCand, uint8_t *test );
/* the same from RAW sums with the transform-skip scaling of :992-1001 applied here: mtsIdx[i] = the candidate's tu.mtsIdx (MTS_SKIP = 1 marks
 * sum |residual| of a transform-skip candidate); maxLog2TrDynamicRange: sps.getMaxLog2TrDynamicRange() (15).  numCand <= 16. */
int vtmhip_mts_select2( const int32_t *sumAbs, const uint8_t *mtsIdx, int numCand, int width, int height, int bitDepth, int maxLog2TrDynamicRange,
                        int maxCand, uint8_t *test );

/* ================================================================================================================
 * (2) BATCHED DEVICE CALLS -- device pointers, asynchronous on the context's stream
 *     One exception to "asynchronous": a MIXED-shape batch (no uniform promise) of >= 64 jobs to vtmhip_xMotionEstimation_batch_dev or >= 256 jobs to
 *     vtmhip_tu_chain_batch_dev is bucketed by block shape on the device, and the class counts (80 bytes) are read back with ONE
 *     hipStreamSynchronize inside the call.  While the stream is being captured into a hipGraph the bucketing is skipped (the generic chain runs the
 *     whole batch: same results); uniform batches never synchronise.
 * ============================================================================================================== */

/* One distortion evaluation: org block at orgBase + orgOff, candidate block at curBase + curOff (offsets in samples). */
typedef struct
{
  int64_t orgOff;
  int64_t curOff;
  int32_t orgStride;
  int32_t curStride;
  int16_t width;
  int16_t height;
  int16_t subShift;   /* SAD only */
  int16_t kind;       /* VTMHIP_DIST_* */
} vtmhip_dist_job;

/* n independent distFunc evaluations (hooks B1-B7, B10).  d_jobs and d_dist are device arrays of n entries. */
int vtmhip_dist_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const vtmhip_dist_job *d_jobs, int n,
                           uint64_t *d_dist );

typedef struct
{
  int64_t orgOff, curOff, maskOff;   /* maskOff: first mask sample (DistParam::mask) inside d_maskBase */
  int32_t orgStride, curStride, maskStride, maskStride2;
  int16_t width, height, subShift, stepX;
} vtmhip_masked_sad_job;

/* n masked SADs (DF_SAD_WITH_MASK), e.g. every (GEO split, merge candidate) pair of a CU in one launch */
int vtmhip_masked_sad_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const int16_t *d_maskBase,
                                 const vtmhip_masked_sad_job *d_jobs, int n, uint64_t *d_dist );

/* The same distortions for a batch the caller promises to be uniform: EVERY job is width x height of one `kind` (and subShift for SAD); the jobs'
 * own width / height / kind / subShift fields are ignored.  Small blocks then share a wave (an 8x8 SATD is one Hadamard tile = one lane): the form
 * hooks B7 / B10 use for the merge / AMVP candidates of one CU size. */
int vtmhip_dist_uniform_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_curBase, const vtmhip_dist_job *d_jobs, int n,
                                   int kind, int width, int height, int subShift, uint64_t *d_dist );

/* Intra mode pre-selection (IntraSearch::estIntraPredLumaQT, EncoderLib/IntraSearch.cpp:555-592: per tested mode `min( 2 * SAD, SATD )` of the mode's prediction against
 * the original block -- the same DistParam::distFunc slots as the inter path): the N candidate predictions of ONE block in one call.  The host forms the predictors
 * (predIntraAng: angular / planar / DC with PDPC, reference smoothing -- not part of this library) as N consecutive width x height blocks (stride = width) at d_predBase +
 * predOff; the original block sits at d_orgBase + orgOff.  d_dist[0 .. N) = the SADs, d_dist[N .. 2N) = the SATDs (xGetHADs tile rules), in candidate order. */
int vtmhip_intra_cand_cost_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, int64_t orgOff, int orgStride, const int16_t *d_predBase, int64_t predOff, int n, int width,
                                      int height, uint64_t *d_dist );

/* SATD 8x8 block-grid micro-benchmark: every 8-aligned 8x8 block of the W x H org picture against the
 * reference picture displaced by (dx,dy) in [-r,r]^2.  d_ref must carry >= r samples of valid margin on every side.
 * d_dist[(by*(W/8)+bx)*(2r+1)^2 + (dy+r)*(2r+1) + (dx+r)], 32-bit (an 8x8 SATD of int16 samples fits). */
int vtmhip_satd8_grid_dev( vtmhip_ctx *ctx, const int16_t *d_org, int orgStride, const int16_t *d_ref, int refStride, int width, int height,
                           int r, uint32_t *d_dist );

/* ---- reference planes under d_refBase: what must be readable ---------------------------------------------------------
 * A reference plane handed to any *_dev call is the reconstructed picture WITH its border: at least ctuSize + 16 samples on every side (the reference's own
 * Picture::margin, 144 for CTU 128: clipMv keeps a block's origin inside [-ctuSize - 7, pic + 7], the 8-tap filter adds 3 / 4 samples, the block its size).
 * The kernels fetch reference samples in whole 16-byte groups and whole window rows (search windows, the column-walking raster scan), so a fetch may start
 * before / end after the samples it uses: the library may READ (never write, never use) up to VTMHIP_PLANE_SLACK samples before the first and after the last
 * sample of a plane (first = row -margin, column -margin; last = the last sample of row height + margin - 1).  Planes that sit inside one allocation (the DPB
 * layout of DESIGN.md section 2) give each other that slack; the first and the last plane of an allocation need VTMHIP_PLANE_SLACK readable samples in front
 * of / behind them -- allocate ( samples + 2 * VTMHIP_PLANE_SLACK ) and hand over the pointer VTMHIP_PLANE_SLACK samples in (oracle/ref_shim_enc.cpp:refPlane
 * does exactly that for the encoder's pictures; tests/pis_golden.py:build_dpb for the recorded planes). */
#define VTMHIP_PLANE_SLACK 2048

/* ---- integer motion search ----------------------------------------------------------------------------------- */
typedef struct
{
  int32_t picW, picH;   /* pps.getPicWidth/HeightInLumaSamples: clipMv / xClipMv limits */
  int32_t ctuSize;      /* sps.getMaxCUWidth() */
  int32_t bitDepth;
  int32_t wavesPerJob;  /* tuning hint for this batch: 0/1 = one wave per search; 2, 4, 8, 16 = waves that split each candidate list (large PUs) */
  int32_t maxSearchRange; /* tuning hint: the largest searchRange of the batch's TZ jobs (m_aaiAdaptSR: up to 384 with ASR).  The raster scan of xTZSearch (:3888-3899) is run by a
                             column-walking kernel whose per-scan totals live in LDS: 0 (or <= 96) sizes it for the 39 x 39 points of SearchRange 96; a larger value for
                             ((2 * range) / 5 + 1)^2 points (384: 154 x 154), capped at what one CU's LDS holds (201 x 201 points = range 500).  A scan that does not fit runs
                             inside the search kernel: same result, much slower.  Any value is accepted (a hint never makes a call fail). */
} vtmhip_pic_params;

/* One (PU, reference picture) integer search = one call of InterSearch::xTZSearch. */
typedef struct
{
  int64_t orgOff;       /* sample offset of the PU's top-left in the original plane (pcPatternKey)         */
  int64_t refOff;       /* sample offset of the SAME position (MV 0,0) in the reference plane (piRefY)      */
  int32_t orgStride, refStride;
  int16_t puX, puY;     /* luma position of the PU in the picture                                          */
  int16_t width, height;
  int16_t subShift;     /* DistParam::subShift after setDistParam(subShiftMode) (RdCost.cpp:289-323)       */
  uint8_t imvShift;
  uint8_t signedSamples;   /* 0: every org/ref sample is >= 0 (uni-pred ME on pictures); 1: full int16 range (2*org - pred targets) */
  int32_t predHor, predVer;     /* RdCost::setPredictor, quarter-sample units                              */
  double  motionLambda;         /* RdCost::m_motionLambda                                                  */
  int32_t mvHor, mvVer;         /* rcMv on entry (internal 1/16 precision)                                 */
  int32_t searchRange;          /* m_iSearchRange                                                          */
  uint8_t extendedSettings, fastSettings, firstSearchStop, hasIntMv2Nx2NPred;
  int32_t intMv2Nx2NPredHor, intMv2Nx2NPredVer;
  int32_t numExtraStart;        /* m_uniMvList candidates, de-duplicated by the caller (InterSearch.cpp:3728-3746) */
  int32_t extraStart[15][2];    /* internal precision */
} vtmhip_tz_job;

typedef struct
{
  int32_t  mvX, mvY;   /* integer MV (rcMv on return) */
  uint32_t nEval;      /* distFunc evaluations performed (statistics; the reference does not return it) */
  uint32_t reserved;
  uint64_t cost;       /* cStruct.uiBestSad: distortion + MV rate */
  uint64_t dist;       /* ruiSAD */
} vtmhip_me_result;

int vtmhip_tz_search_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                const vtmhip_tz_job *d_jobs, int n, vtmhip_me_result *d_results );


/* One exhaustive search = xSetSearchRange + xPatternSearch (InterSearch.cpp:3496-3608): every integer position of the
 * clipped window around `center`, first strict minimum in raster order (hook B5, the +-BipredSearchRange refinement). */
typedef struct
{
  int64_t orgOff, refOff;
  int32_t orgStride, refStride;
  int16_t puX, puY, width, height;
  int16_t subShift;
  uint8_t imvShift;
  uint8_t signedSamples;        /* 1 for the bi-pred target 2*org - pred */
  int32_t predHor, predVer;     /* quarter-sample units */
  double  motionLambda;
  int32_t centerHor, centerVer; /* bestInitMv, internal 1/16 precision */
  int32_t searchRange;          /* iSrchRng (BipredSearchRange = 4) */
  int32_t pad;
} vtmhip_full_job;

int vtmhip_full_search_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                  const vtmhip_full_job *d_jobs, int n, vtmhip_me_result *d_results );
/* Same searches, same results, for a batch the caller promises to be uniform: EVERY job is width x height with searchRange <= 4
 * (the bi-pred refinement of one quadtree level or of one split shape).  Squares 8 .. 64 and the binary / ternary split shapes 16x8, 32x8, 32x16, 64x16,
 * 64x32 in both orientations: one lane per candidate over an LDS-resident window; any other shape forwards to vtmhip_full_search_batch_dev.
 * The reference plane must be readable 7 samples beyond the search window's right edge. */
int vtmhip_full_search_uniform_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                          const vtmhip_full_job *d_jobs, int n, int width, int height, vtmhip_me_result *d_results );
/* width == height == size */
int vtmhip_full_search_square_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                         const vtmhip_full_job *d_jobs, int n, int size, vtmhip_me_result *d_results );

/* ---- whole motion estimation of one (PU, list, refIdx): InterSearch::xMotionEstimation -------------------------------------
 * (InterSearch.cpp:3299-3494; AMVR integer refinement xPatternSearchIntRefine :4172-4282).  One call runs, for n jobs:
 *   bi:  pattern = 2*org - otherPred (removeHighFreq :3320-3326), best start among rcMv and the m_uniMvList entries (:3377-3420),
 *        xSetSearchRange + xPatternSearch over +-bipredSearchRange;        uni: xTZSearch from rcMvPred (:3440-3446)
 *   cu.imv 0 / IMV_HPEL: xPatternSearchFracDIF and the rate re-weighting of :3478-3484
 *   cu.imv 1 / 2:        xPatternSearchIntRefine over 9 positions x the AMVP candidates
 * as a fixed sequence of launches on the context's stream (no host synchronisation).  Not covered (the trampoline keeps the host
 * path): BCW weights, weighted prediction, MCTS, composite references, the block-MV cache (xReadBufferedUniMv / CacheBlkInfoCtrl). */
typedef struct
{
  int32_t bipredSearchRange;       /* m_bipredSearchRange */
  uint8_t useHadME;                /* HadamardME && !slice.getDisableSATDForRD() */
  uint8_t fastInterSearchMode13;   /* FEN mode 1 or 3: setDistParam subShiftMode 2 (RdCost.cpp:289-323); else 0 */
  uint8_t extendedSettings;        /* MESEARCH_DIAMOND_ENHANCED */
  uint8_t firstSearchStop;         /* FastMEAssumingSmootherMVEnabled */
  int32_t uniformImv;              /* -1: jobs mix cu.imv values; 0..3: every job of the batch has this cu.imv (lets whole stages be skipped) */
  int32_t uniformSquare;           /* != 0: every job is exactly maxWidth x maxHeight (squares 8 .. 128 or a split shape 16x8 .. 64x32, either orientation:
                                      tiled fractional kernel when uniformImv is 0 or 3); the name predates the rectangular fast paths */
  int32_t uniformBi;               /* 0: jobs mix bBi values; 1: every job is a uni search (no pattern copies: the searches read the original plane);
                                      2: every job is a bi search (no TZ stage; lane-per-candidate exhaustive kernel when uniformSquare) */
  uint8_t noUniMvList;             /* caller's promise: numExtraStart == 0 in every job (with uniformBi 2 the start-candidate SADs are skipped: rcMv is the start) */
  uint8_t biPatternGiven;          /* with uniformBi 2: d_otherPredBase + otherPredOff already holds the search pattern 2*org - otherPred (a fused
                                      vtmhip_motion_compensation_batch_dev epilogue wrote it), not the other list's prediction: no removeHighFreq pass here */
  uint8_t pad0, pad1;
} vtmhip_me_cfg;

typedef struct
{
  int64_t  orgOff, refOff;          /* PU top-left in the original plane / the same position (MV 0,0) in the reference plane */
  int64_t  otherPredOff;            /* bi: block of the other list's prediction inside d_otherPredBase (m_tmpPredStorage[1 - list]) */
  int32_t  orgStride, refStride, otherPredStride;
  int16_t  puX, puY, width, height;
  uint8_t  bi, imv, mvpIdx, numAmvpCand;   /* bBi, cu.imv (0 quarter, 1 integer, 2 four-sample, 3 half), riMVPIdx, amvpInfo.numCand */
  int32_t  mvPredHor, mvPredVer;    /* rcMvPred, internal 1/16 precision */
  int32_t  mvHor, mvVer;            /* rcMv on entry (start of the bi-pred search) */
  int32_t  amvpCand[2][2];          /* amvpInfo.mvCand */
  uint32_t mvpIdxBits[2];           /* m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] */
  uint32_t bits;                    /* ruiBits on entry */
  int32_t  searchRange;             /* m_aaiAdaptSR[list][refIdx] */
  double   motionLambda;
  int32_t  numExtraStart;           /* m_uniMvListSize */
  int32_t  extraStart[15][2];       /* uniMvs[list][refIdx] of the m_uniMvList entries, newest first, duplicates allowed */
  uint32_t flags;                   /* VTMHIP_MEJ_* */
} vtmhip_me_job;
/* uni job: the block-vector cache of the mode control holds an integer vector for this (block, list, refIdx) (CacheBlkInfoCtrl::getMv, InterSearch.cpp:3360-3368):
 * rcMv = mvHor / mvVer (that vector in internal precision) and xTZSearch runs with bFastSettings (:3434-3441) instead of starting at rcMvPred */
#define VTMHIP_MEJ_CACHED_INT_MV 1u
/* uni row of vtmhip_predInterSearch_batch_dev that is GIVEN, not searched (InterSearch::xReadBufferedUniMv :7677-7697: a CU-level BCW weight other than the default re-uses the
 * vectors the default-weight pass left in m_uniMotions): on entry uniOut[row].mvHor / mvVer = the buffered vector and uniOut[row].cost = its distortion WITHOUT any rate; the
 * row's AMVP estimation runs as usual, its bits become the entry bits + the vector's bits against the chosen predictor (both in the AMVR precision) and its cost
 * distortion + getCost( bits ); uniOut[row] then holds that result.  The level lists the (list, refIdx) groups whose rows are all given in vtmhip_pis_level::givenRows. */
#define VTMHIP_MEJ_GIVEN_UNI 2u
/* bi job: flags bits 8..15 = getBcwWeight( cu.BcwIdx, searched list ) as int8 (0 or 4: the default weight): the search target is removeWeightHighFreq( org, otherPred, w )
 * = ( org * w0 - otherPred * w1 + 2^15 ) >> 16 (Buffer.h:417-460) instead of 2 * org - otherPred, and the distortion weight of :3483 / xPatternSearchIntRefine is |w| / 8
 * (xGetMEDistortionWeight :7666-7676) instead of 0.5 */
#define VTMHIP_MEJ_BCW_WEIGHT( flags ) ( ( int ) ( int8_t ) ( ( ( flags ) >> 8 ) & 0xffu ) )
#define VTMHIP_MEJ_BCW_FLAGS( weight ) ( ( ( uint32_t ) ( uint8_t ) ( int8_t ) ( weight ) ) << 8 )

typedef struct
{
  int32_t  mvHor, mvVer;            /* rcMv, internal precision */
  int32_t  mvPredHor, mvPredVer;    /* rcMvPred (changes only in the AMVR integer refinement) */
  int32_t  mvpIdx;                  /* riMVPIdx */
  uint32_t bits;                    /* ruiBits */
  uint64_t cost;                    /* ruiCost */
  int32_t  intX, intY;              /* integer-stage vector (m_integerMv2Nx2N for uni searches) */
  uint64_t intDist;                 /* distortion of the integer stage without the vector rate (the D_ME trace value MECostFPel) */
} vtmhip_me_out;

int vtmhip_xMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const vtmhip_me_cfg *cfg, const int16_t *d_orgBase,
                                        const int16_t *d_refBase, const int16_t *d_otherPredBase, const vtmhip_me_job *d_jobs, int n,
                                        int maxWidth, int maxHeight, vtmhip_me_out *d_results );

/* ---- AMVP predictor estimation: InterSearch::xEstimateMvPredAMVP with xGetTemplateCost (hook B7) ----------------------------------
 * (InterSearch.cpp:3088-3128, 3235-3270; caller predInterSearch :2367).  For each of the n (PU, list, refIdx) rows and each of its
 * numAmvpCand candidates (amvpCand, already filled -- bFilled): clipMv, uni-directional luma prediction at the candidate (xPredInterBlk, rounded
 * and clipped), SAD against the original block + getCost( mvpIdxBits[i] ); the FIRST candidate with the smallest cost wins (`uiBestCost >
 * uiTmpCost`).  In place: mvPredHor / mvPredVer (the unclipped candidate), mvpIdx; with addIdxBits the winner's index bits are added to `bits`
 * as predInterSearch does right after the call (:2381).  d_distBiP (may be NULL): the winning template cost per row (*puiDistBiP).
 * uniformSize != 0: every row is maxWidth x maxHeight (uniform SAD kernel). */
int vtmhip_xEstimateMvPredAMVP_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                          vtmhip_me_job *d_jobs, int n, int maxWidth, int maxHeight, int uniformSize, int addIdxBits, uint64_t *d_distBiP );

/* ---- motion compensation / bi-pred buffer ops ------------------------------------------------------------------------ */
typedef struct
{
  int64_t refOff, dstOff;       /* refOff: block position with MV (0,0) in the reference plane */
  int32_t refStride, dstStride;
  int16_t width, height;
  int32_t mvHor, mvVer;         /* internal 1/16 precision */
  uint8_t bi;                   /* 0: rounded + clipped samples (uni-pred); 1: 14-bit intermediates for addAvg */
  uint8_t bitDepth, useAltHpelIf;
  uint8_t chroma;               /* 0: luma plane, 8-tap, width/height/offsets in luma samples; 1: a 4:2:0 chroma plane, 4-tap at 1/32 phase --
                                   width/height/refOff/dstOff in samples of THAT plane, the vector still in luma 1/16 units */
} vtmhip_mc_job;

/* InterPrediction::xPredInterBlk without BDOF/DMVR/RPR/wrap-around (InterPrediction.cpp:660-815), luma and 4:2:0 chroma blocks mixed
 * freely in one batch (vtmhip_mc_job::chroma); maxWidth/maxHeight bound the block sizes of the batch (2..128). */
int vtmhip_mc_batch_dev( vtmhip_ctx *ctx, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_mc_job *d_jobs, int n, int maxWidth,
                         int maxHeight );
/* the same call under its first name (luma-only callers) */
int vtmhip_mc_luma_batch_dev( vtmhip_ctx *ctx, const int16_t *d_refBase, int16_t *d_dstBase, const vtmhip_mc_job *d_jobs, int n, int maxWidth,
                              int maxHeight );

/* InterPrediction::motionCompensation for one PU and one plane (InterPrediction.cpp:445-660: xPredInterUni, or xPredInterBi with the
 * default-weight xWeightedAverage = PelBuf::addAvg :1354-1435), with the consumer of the prediction optionally fused in. */
typedef struct
{
  int64_t orgOff;               /* block in the original plane (epilogue 1 / 2) */
  int64_t refOff[2];            /* block position with MV (0,0) in the list-0 / list-1 reference plane (both inside d_refBase) */
  int64_t predOff, outOff;      /* prediction block inside d_predBase, epilogue output inside d_outBase */
  int32_t orgStride, refStride[2], predStride, outStride;
  int32_t mv[2][2];             /* [list][hor, ver], internal 1/16 luma precision */
  int16_t width, height;
  uint8_t mode;                 /* 0: uni-prediction from list 0, 1: from list 1, 2: bi-prediction (two 14-bit predictions, addAvg) */
  uint8_t epilogue;             /* 0: none; 1: out = org - pred (residual, InterSearch.cpp:7260-7262); 2: out = 2*org - pred (removeHighFreq) */
  uint8_t bitDepth, useAltHpelIf, chroma;
  uint8_t route;                /* 0: every call processes the job; a driver that launches BOTH vtmhip_motion_compensation_batch_dev and vtmhip_bdof_batch_dev over
                                   one table marks each job on the device: 1 = BDOF's (the plain call skips it), 2 = the plain call's (BDOF skips it) */
  int16_t bcwWeight;            /* epilogue 2 only: 0 or 4 = out = 2*org - pred; another getBcwWeight( cu.BcwIdx, searched list ) in {-2, 3, 5, 10}: out = removeWeightHighFreq
                                   = ( org * w0 - pred * w1 + 2^15 ) >> 16 with normalizer = ( 2^16 + |w| / 2 ) / w, w0 = normalizer * 8, w1 = ( 8 - w ) * normalizer (Buffer.h:417-460) */
} vtmhip_pred_job;

/* d_predBase and d_outBase may each be NULL (that output is skipped); d_orgBase is needed when d_outBase is given. */
int vtmhip_motion_compensation_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase,
                                          const vtmhip_pred_job *d_jobs, int n, int maxWidth, int maxHeight );

typedef struct
{
  int64_t aOff, bOff, dstOff;
  int32_t aStride, bStride, dstStride;
  int16_t width, height;
  uint8_t bitDepth;
  int8_t  bcwWeight;            /* the BCW calls only: g_BcwWeights[bcwIdx] in {-2, 3, 4, 5, 10} (of 8), see there */
  uint8_t pad1, pad2;
  int32_t pad3;
} vtmhip_pelop_job;

/* PelBuf::removeHighFreq (Buffer.cpp:475-520; bi-pred ME target, InterSearch.cpp:3320-3326): dst = 2*org - pred, unclipped */
int vtmhip_remove_high_freq_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_predBase, int16_t *d_dstBase,
                                       const vtmhip_pelop_job *d_jobs, int n );
/* PelBuf::subtract (residual = org - pred; CodingStructure resi buffer, InterSearch.cpp:7260-7262) */
int vtmhip_subtract_batch_dev( vtmhip_ctx *ctx, const int16_t *d_aBase, const int16_t *d_bBase, int16_t *d_dstBase, const vtmhip_pelop_job *d_jobs, int n );
/* PelBuf::addAvg (Buffer.cpp:467-507): dst = clip((src0 + src1 + offset) >> shift) on 14-bit intermediates */
int vtmhip_add_avg_batch_dev( vtmhip_ctx *ctx, const int16_t *d_src0Base, const int16_t *d_src1Base, int16_t *d_dstBase,
                              const vtmhip_pelop_job *d_jobs, int n );

/* BCW (Buffer.h:417-460, Buffer.cpp:365-397).  removeWeightHighFreq: the bi-pred ME target when the searched list carries weight job.bcwWeight,
 * dst = ( org * (n << 3) - pred * ((8 - w) * n) + 2^15 ) >> 16 with n = (2^16 + |w >> 1|) / w, unclipped (g_pelBufOP.removeWeightHighFreq4/8).
 * addWeightedAvg: dst = clip( ( src0 * (8 - w) + src1 * w + offset ) >> shift ), job.bcwWeight = w = the LIST-1 weight g_BcwWeights[bcwIdx]. */
int vtmhip_remove_weight_high_freq_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_predBase, int16_t *d_dstBase,
                                              const vtmhip_pelop_job *d_jobs, int n );
int vtmhip_add_weighted_avg_batch_dev( vtmhip_ctx *ctx, const int16_t *d_src0Base, const int16_t *d_src1Base, int16_t *d_dstBase,
                                       const vtmhip_pelop_job *d_jobs, int n );

/* BDOF: InterPrediction::xPredInterBi with bioApplied for bi-predicted LUMA PUs (InterPrediction.cpp:527-660: xSubPuBio :352-443 cuts the PU into
 * regions of at most 16 x 16, xPredInterBlk(..., bioApplied) :733-810 predicts each from both lists inside a ring of integer samples, and
 * xWeightedAverage -> applyBiOptFlow :1233-1334 refines every 4 x 4 unit with g_pelBufOP.bioGradFilter / calcBIOSums / addBIOAvg4, Buffer.cpp:88-200).
 * The job table is the one of vtmhip_motion_compensation_batch_dev: the caller sends here the PUs for which the reference sets bioApplied (:527-572:
 * true bi-prediction with equal and opposite POC distances, w, h >= 8, w*h >= 128, no affine / SMVD / CIIP / BCW / weighted prediction); mode must be
 * 2, chroma 0 (the chroma planes of such a PU take the plain addAvg path).  The epilogue and the NULL rules are those of that call. */
int vtmhip_bdof_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase, int16_t *d_outBase,
                           const vtmhip_pred_job *d_jobs, int n, int maxWidth, int maxHeight );

/* DMVR: InterPrediction::xProcessDMVR (InterPrediction.cpp:1997-2195) for the LUMA plane of the bi-predicted PUs for which the reference's
 * PU::checkDMVRCondition holds (merge mode, equal and opposite POC distances, w, h >= 8, w*h >= 128, default weights ...; xPredInterBi :576-583).
 * Per sub-PU of at most 16 x 16: xPrefetch, bilinear xinitMC, the 25-point mirrored integer refinement (xDMVRCost / xBIPMVRefine :1819-1927), the
 * parametric error surface (:1733-1817, 1929-1947), xPad + xFinalPaddedMCForDMVR (:1709-1731, 1845-1917) and xWeightedAverage, with BDOF
 * (vtmhip_bdof_batch_dev's arithmetic) where bioApplied is set and the sub-PU's matching cost is not below 2*dx*dy (:2139). */
typedef struct
{
  int64_t orgOff;               /* block in the original plane (epilogue 1 / 2) */
  int64_t refOff[2];            /* PU position with MV (0,0) in the list-0 / list-1 reference plane (both inside d_refBase) */
  int64_t predOff, outOff;      /* prediction block inside d_predBase, epilogue output inside d_outBase */
  int32_t orgStride, refStride[2], predStride, outStride;
  int32_t mv[2][2];             /* the merge vectors, [list][hor, ver], internal 1/16 precision */
  int32_t puX, puY;             /* luma position of the PU in the picture: clipMv (Mv.cpp:56-74) */
  int16_t width, height;
  uint8_t bioApplied;           /* what xPredInterBi decided for the PU (:527-572) */
  uint8_t epilogue;             /* as vtmhip_pred_job */
  uint8_t bitDepth, pad0;
  int32_t mvdRow;               /* chroma call only: the d_mvd row (= job index of the luma call) that holds this PU's vector differences */
} vtmhip_dmvr_job;
/* pic: picW / picH / ctuSize / bitDepth.  d_mvd (may be NULL): pu.mvdL0SubPu, int32 [n][regions][2] with regions = ceil(maxWidth/16) * ceil(maxHeight/16),
 * sub-PUs in the reference's raster order. */
int vtmhip_dmvr_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                           int16_t *d_outBase, const vtmhip_dmvr_job *d_jobs, int n, int maxWidth, int maxHeight, int32_t *d_mvd );

/* One 4:2:0 chroma plane of the same PUs, after the luma call: a sub-PU whose difference is zero is predicted from the reference pictures, a moved one
 * out of its (w/2+3) x (h/2+3) window -- prefetched with the MERGE vector (xPrefetch forLuma = 0, :1666-1708), replicated by one sample (xPad: padsize =
 * 2 >> scaleY, :1709-1731), addressed with the whole-sample part of the refined vector (xFinalPaddedMCForDMVR :1879-1905) -- then the plain addAvg.
 * The jobs keep width / height / puX / puY / mv in LUMA units; refOff / strides / orgOff / predOff / outOff address THAT chroma plane (refOff: the PU's
 * position in it); d_mvd, maxWidth and maxHeight are those of the luma call (same `regions`), vtmhip_dmvr_job::mvdRow picks the row. */
int vtmhip_dmvr_chroma_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                                  int16_t *d_outBase, const vtmhip_dmvr_job *d_jobs, int n, int maxWidth, int maxHeight, const int32_t *d_mvd );

/* InterpolationFilter::m_weightedGeoBlk (InterpolationFilter.h:99; xWeightedGeoBlk InterpolationFilter.cpp:902-957, x86/InterpolationFilterX86.h:1343-1470;
 * callers InterPrediction::weightedGeoBlk InterPrediction.cpp:1642-1661, EncCu.cpp:3004,3030): blend of the two GEO partitions' 14-bit predictions,
 *   dst = clip( ( w * src0 + (8 - w) * src1 + offset ) >> shift ),  shift = max(2, 14 - bitDepth) + 3,  offset = (1 << (shift-1)) + (8192 << 3),
 * with w = weight[y * weightStride + x * stepX] in 0..8.  The caller (the trampoline that receives pu / splitDir) derives weight, stepX and
 * weightStride from g_GeoParams / g_angle2mirror / g_weightOffset exactly as the reference does: stepX = +-1 (luma) or +-2 (4:2:0 chroma),
 * weightStride = +-GEO_WEIGHT_MASK_SIZE << scaleY (the scalar version's `width * stepX + stepY`). */
int vtmhip_weightedGeoBlk( vtmhip_ctx *ctx, const int16_t *src0, int src0Stride, const int16_t *src1, int src1Stride, int16_t *dst, int dstStride, int width,
                           int height, const int16_t *weight, int stepX, int weightStride, int bitDepth, int clipMin, int clipMax );
typedef struct
{
  int64_t src0Off, src1Off, dstOff;   /* samples inside d_srcBase / d_srcBase / d_dstBase */
  int64_t weightOff;                  /* first weight (top-left output sample) inside d_weightBase */
  int32_t src0Stride, src1Stride, dstStride, weightStride;
  int16_t width, height, stepX, pad;
} vtmhip_geo_blend_job;
/* n GEO blends in one launch (e.g. every tested (split, candidate pair) of a CU, EncCu.cpp:2990-3035); d_weightBase: the g_globalGeoWeights planes,
 * uploaded once */
int vtmhip_weightedGeoBlk_batch_dev( vtmhip_ctx *ctx, const int16_t *d_srcBase, int16_t *d_dstBase, const int16_t *d_weightBase,
                                     const vtmhip_geo_blend_job *d_jobs, int n, int bitDepth, int clipMin, int clipMax );

/* ---- interpolation ----------------------------------------------------------------------------------------------- */
typedef struct
{
  int64_t srcOff, dstOff;   /* samples; srcOff addresses the output-aligned sample */
  int32_t srcStride, dstStride;
  int16_t width, height;
  uint8_t vertical, taps /* 8, 4, 2; 0 = filterCopy */, isFirst, isLast;
  int16_t coeff[8];
  int16_t clipMin, clipMax;
  uint8_t bitDepth, biMCForDMVR, pad0, pad1;
} vtmhip_if_job;

/* n filter / copy passes, one workgroup each (hook B6 plane generation, motion compensation) */
int vtmhip_if_batch_dev( vtmhip_ctx *ctx, const int16_t *d_srcBase, int16_t *d_dstBase, const vtmhip_if_job *d_jobs, int n );

/* ---- fractional motion search: one InterSearch::xPatternSearchFracDIF per job (hook B6) ------------------------------ */
typedef struct
{
  int64_t orgOff, refOff;   /* as vtmhip_tz_job: PU top-left in the original plane / same position (MV 0,0) in the reference plane */
  int32_t orgStride, refStride;
  int16_t width, height;    /* not 4x4 (no 4x4 inter PU exists; the reference switches tap tables there, InterpolationFilter.cpp:786-789) */
  int16_t intX, intY;       /* rcMvInt: result of the integer search */
  int32_t predHor, predVer; /* RdCost::setPredictor, quarter-sample units */
  double  motionLambda;
  uint8_t useHad;           /* HadamardME && !DisableSATDForRD: SATD, else SAD.  For bitDepth <= 10 the tiled kernel forms the SATD differences in 16 bits (as the
                               reference's SIMD does): the original samples must lie in [-3072, 3071] -- picture samples or the bi-pred target 2*org - pred */
  uint8_t useAltHpelIf;     /* cu.imv == IMV_HPEL */
  uint8_t imvShift;         /* 0: half + quarter refinement; 1 (IMV_HPEL): half only */
  uint8_t bitDepth;
  int32_t wideOrg;          /* != 0: the original samples may leave [-3072, 3071] (a BCW-weighted bi-pred target: up to +-5115): the tiled kernel takes its 32-bit SATD path */
} vtmhip_frac_job;

typedef struct
{
  int16_t  halfX, halfY;   /* rcMvHalf in {-1,0,1}^2 */
  int16_t  qterX, qterY;   /* rcMvQter in {-1,0,1}^2; final MV (quarter units) = (int << 2) + (half << 1) + qter */
  uint64_t cost;           /* ruiCost */
} vtmhip_frac_result;

/* maxWidth/maxHeight: upper bounds of the job sizes in this batch (they size the per-workgroup LDS window).
 * uniformSquare != 0: the caller guarantees that EVERY job is exactly maxWidth x maxHeight -- a square in {8,16,32,64,128} or one of the split
 * shapes 16x8, 32x8, 32x16, 64x16, 64x32 in either orientation (their SATD tiles are the reference's 16x8 / 8x16 Hadamards) -- and that all
 * jobs share imvShift; the library then uses the tiled kernel (one lane per (PU, candidate, 8x8 tile)); other uniform shapes and
 * 0 = any mix of sizes: one wave per PU. */
int vtmhip_frac_search_batch_dev( vtmhip_ctx *ctx, const int16_t *d_orgBase, const int16_t *d_refBase, const vtmhip_frac_job *d_jobs, int n,
                                  int maxWidth, int maxHeight, int uniformSquare, vtmhip_frac_result *d_results );

/* ---- transform / quantisation: one TU per job --------------------------------------------------------------------- */
typedef struct
{
  int64_t srcOff, dstOff;   /* xT: residual samples -> coefficients; xIT: coefficients -> residual samples (offsets in elements) */
  int32_t srcStride;        /* xT: residual stride (CS resi stride); coefficients are always W x H contiguous */
  int32_t dstStride;        /* xIT: residual stride */
  int16_t width, height;    /* 1..64 (MAX_TB_SIZEY) */
  uint8_t typeHor, typeVer; /* VTMHIP_DCT2 / DCT8 / DST7 as TrQuant::getTrTypes chose them */
  uint8_t bitDepth, pad;
} vtmhip_tr_job;

/* TrQuant::xT (TrQuant.cpp:776-851) for n TUs.  d_sumAbs (may be NULL): sum |coef| per TU (MTS pre-selection, :986-990). */
int vtmhip_xT_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, int32_t *d_coefBase, const vtmhip_tr_job *d_jobs, int n, int maxWidth,
                         int maxHeight, int32_t *d_sumAbs );
/* TrQuant::xIT (TrQuant.cpp:853-923) */
int vtmhip_xIT_batch_dev( vtmhip_ctx *ctx, const int32_t *d_coefBase, int16_t *d_resiBase, const vtmhip_tr_job *d_jobs, int n, int maxWidth,
                          int maxHeight );

typedef struct
{
  int64_t srcOff, dstOff;   /* W x H contiguous TCoeff blocks */
  int16_t width, height;
  int16_t qpPer, qpRem;     /* QpParam::per / rem (Quant.cpp:65-104) */
  uint8_t bitDepth, isIRAP, isTransformSkip, pad;
  int32_t pad2;
} vtmhip_quant_job;

/* Quant::quant, flat scaling list, no sign-bit hiding (Quant.cpp:955-1038): levels, optional deltaU, absSum per TU */
int vtmhip_quant_batch_dev( vtmhip_ctx *ctx, const int32_t *d_coefBase, int32_t *d_qBase, int32_t *d_deltaUBase, const vtmhip_quant_job *d_jobs, int n,
                            int32_t *d_absSum );
/* Quant::dequant, flat scaling list (Quant.cpp:357-482) */
int vtmhip_dequant_batch_dev( vtmhip_ctx *ctx, const int32_t *d_qBase, int32_t *d_coefBase, const vtmhip_quant_job *d_jobs, int n );

/* ---- fused residual-coding chain: hooks B8 + B9 in one launch -------------------------------------------------------------
 * per TU and transform candidate: TrQuant::xT -> Quant::quant -> Quant::dequant -> TrQuant::xIT -> SSE(residual, reconstruction)
 * (xEstimateInterResidualQT, InterSearch.cpp:6637-6733, minus the CABAC bit estimate between quant and dequant).  2-D TUs only. */
typedef struct
{
  int64_t resiOff;          /* residual samples (int16) */
  int64_t outOff;           /* W x H contiguous block inside d_levelsBase / d_recBase (when those are given) */
  int32_t resiStride;
  int16_t width, height;    /* 2..64 */
  int16_t qpPer, qpRem;
  uint8_t typeHor, typeVer, bitDepth, isIRAP;
  int32_t pad;
} vtmhip_tu_job;

typedef struct
{
  uint64_t sse;             /* getDistPart( DF_SSE ) of residual vs reconstructed residual */
  int32_t  sumAbs;          /* sum |coef| after xT (MTS pre-selection cost, TrQuant.cpp:986-990) */
  int32_t  absSum;          /* uiAbsSum of Quant::quant (cbf = absSum > 0) */
} vtmhip_tu_result;

/* d_levelsBase (quantised levels for the host's CABAC estimate) and d_recBase (reconstructed residual) may be NULL.
 * uniformSize != 0: the caller guarantees every TU is exactly maxWidth x maxHeight with a real transform (no VTMHIP_TRSKIP) -> register-blocked kernel
 * (both sides >= 8) or one lane per TU (4x4, 8x4, 4x8); other uniform shapes (16x4 ...) take the generic kernel. */
int vtmhip_tu_chain_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int maxWidth, int maxHeight,
                               int uniformSize, int32_t *d_levelsBase, int16_t *d_recBase, vtmhip_tu_result *d_results );

/* The transform-skip candidate (tu.mtsIdx == MTS_SKIP; TrQuant.cpp:976-980 xTransformSkip, :925-941 xITransformSkip; Quant.cpp:966-997 with
 * useTransformSkip) of the same chain for a uniform batch: every job width x height (powers of two, 4..32) with typeHor == VTMHIP_TRSKIP and
 * qpPer / qpRem = QpParam::per( true ) / rem( true ).  results[i].sumAbs = sum |residual| (unscaled: vtmhip_mts_select2 applies scaleSAD).
 * vtmhip_tu_chain_batch_dev with uniformSize == 0 takes VTMHIP_TRSKIP jobs mixed with the others. */
int vtmhip_tu_ts_chain_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int width, int height,
                                  int32_t *d_levelsBase, int16_t *d_recBase, vtmhip_tu_result *d_results );

/* The MTS pre-selection of a whole level on the device (vtmhip_mts_select2 per TU): d_results holds numCand runs of numTU results one after the other (the
 * layout vtmhip_pis_level_run uses: results[c * numTU + t]), mtsIdx[c] = the candidate's tu.mtsIdx (1 = transform skip: sumAbs is sum |residual|);
 * d_test[c * numTU + t] = 1 when candidate c of TU t survives.  numCand <= 8. */
int vtmhip_mts_select_batch_dev( vtmhip_ctx *ctx, const vtmhip_tu_result *d_results, int numTU, int numCand, const uint8_t *mtsIdx, int width, int height,
                                 int bitDepth, int maxLog2TrDynamicRange, int maxCand, uint8_t *d_test );

/* TrQuant::xT only, for a batch the caller promises to be uniform (every TU width x height, powers of two 8..64) -- the forward transforms of all
 * MTS candidates of TrQuant::transformNxN( ..., trModes, maxCand ) (TrQuant.cpp:950-1019): same job table as the fused chain (qp fields unused);
 * coefficients (H x W contiguous) go to d_coefBase + outOff, results[i].sumAbs = sum |coef| for vtmhip_mts_select(). */
int vtmhip_xT_uniform_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const vtmhip_tu_job *d_jobs, int n, int width, int height, int32_t *d_coefBase,
                                 vtmhip_tu_result *d_results );

/* ---- affine ME gradients: AffineGradientSearch::m_HorizontalSobelFilter / m_VerticalSobelFilter / m_EqualCoeffComputer -----------
 * (AffineGradientSearch.h:50-54, AffineGradientSearch.cpp:62-170; caller xAffineMotionEstimation, InterSearch.cpp:5340-5775) */
typedef struct
{
  int64_t predOff;      /* prediction block (int16) */
  int64_t resiOff;      /* error block org - pred (int16) */
  int64_t derivHOff;    /* horizontal derivative plane (int32) inside d_derivBase */
  int64_t derivVOff;    /* vertical derivative plane */
  int32_t predStride, resiStride, derivStride;
  int16_t width, height;   /* >= 16 in the reference (affine CUs) */
  uint8_t sixParam;     /* b6Param */
  uint8_t pad[7];
} vtmhip_affine_job;

/* both Sobel planes of n blocks (interior 3x3 Sobel, border samples replicated as the reference does) */
int vtmhip_affine_sobel_batch_dev( vtmhip_ctx *ctx, const int16_t *d_predBase, int32_t *d_derivBase, const vtmhip_affine_job *d_jobs, int n );
/* d_equalCoeff: n x [7][7] int64, ACCUMULATED into (the reference zeroes pEqualCoeff before the call); rows 1..np, columns 0..np are written */
int vtmhip_affine_equal_coeff_batch_dev( vtmhip_ctx *ctx, const int16_t *d_resiBase, const int32_t *d_derivBase, const vtmhip_affine_job *d_jobs, int n,
                                         int64_t *d_equalCoeff );

/* ---- affine motion estimation: InterSearch::xAffineMotionEstimation for one (PU, list, refIdx) per job ----------------------------------
 * (InterSearch.cpp:5340-5775) = the first prediction at the (clipped, AMVR-rounded) start model, up to 7 gradient iterations (error -> Sobel ->
 * normal equations -> solveEqual :5215-5284 -> control-point update -> InterPrediction::xPredAffineBlk :856-1232 -> SATD / SAD + xCalcAffineMVBits
 * :3067-3085) and the control-point refinement (:5655-5765), one workgroup per job from start to end: the fp64 steps run on the device in IEEE
 * double.  Covered: cu.imv 0 / 1 / 2, with AffineAmvrEncOpt too (imv 2: xDetermineBestMvp over the job's AMVP list); BCW weights (bcwWeight); PROF as the
 * caller's flags say.  Without an AMVP list in the job the predictor acMvPred and the AMVP index do not change, so the result is the model, its bits and cost. */
typedef struct
{
  int64_t  orgOff, refOff;          /* PU top-left in the original plane / the same position (MV 0,0) in the reference plane */
  int64_t  otherPredOff;            /* bi: the other list's prediction inside d_otherPredBase */
  int64_t  predOff;                 /* vtmhip_xPredAffineBlk_batch_dev only: where the prediction block goes inside d_dstBase */
  int32_t  orgStride, refStride, otherPredStride, predStride;
  int16_t  puX, puY, width, height; /* 16..128 */
  uint8_t  sixParam;                /* cu.affineType == AFFINEMODEL_6PARAM */
  uint8_t  interDir;                /* pu.interDir as isSubblockVectorSpreadOverLimit sees it */
  uint8_t  imv, bi;
  uint8_t  useSatd;                 /* !slice.getDisableSATDForRD() */
  uint8_t  useAffineType;           /* sps.getUseAffineType(): iteration counts :5470-5483 */
  uint8_t  amvrEncOpt;              /* m_pcEncCfg->getUseAffineAmvrEncOpt() */
  uint8_t  lowDelayRounds;          /* m_pcEncCfg->getIntraPeriod() == -1 */
  uint8_t  profAllowed;             /* sps.getUsePROF() && !m_skipPROF && !picHeader.getDisProfFlag() */
  uint8_t  profNeedsLargeGrad;      /* m_encOnly && !slice.getCheckLDC() */
  uint8_t  profIsBi;                /* m_isBi */
  int8_t   bcwWeight;               /* bi: getBcwWeight( cu.BcwIdx, searched list ); 0 or 4 = default (target 2*org - otherPred, distortion weight 0.5), else the weighted target
                                       of Buffer.h:417-460 and the distortion weight |w| / 8 */
  int32_t  mvPred[3][2];            /* acMvPred */
  int32_t  mv[3][2];                /* acMv on entry */
  uint32_t bits;                    /* ruiBits on entry */
  uint32_t pad1;
  double   motionLambda;
  uint64_t hevcCost;                /* m_hevcCost: the refinement stage runs when the best cost so far <= AFFINE_ME_LIST_MVP_TH * hevcCost */
  /* cu.imv == 2 with AffineAmvrEncOpt (ABI 6): xDetermineBestMvp (:7766-7785) re-picks the affine AMVP candidate at the start model and at every evaluation of the
   * gradient stage (:5444-5449, 5629-5634: acMvPred follows the pick even when the evaluation does not improve the cost; the refinement stage prices against the
   * predictor the LAST gradient evaluation left).  numAmvpCand == 0: no list given, the predictor never changes (every other mode). */
  int32_t  amvpCand[2][3][2];       /* aamvpi.mvCandLT / mvCandRT / mvCandLB of candidate i */
  uint32_t mvpIdxBits[2];           /* m_auiMVPIdxCost[i][aamvpi.numCand] */
  uint8_t  numAmvpCand;             /* aamvpi.numCand (1 or 2), 0: not given */
  uint8_t  mvpIdx;                  /* mvpIdx on entry: `bits` holds m_auiMVPIdxCost[mvpIdx][numCand] (dirBits = bits - that, :5359) */
  uint8_t  pad2[6];
} vtmhip_affine_me_job;

typedef struct
{
  int32_t  mv[3][2];                /* acMv */
  uint32_t bits;                    /* ruiBits */
  int32_t  iterations, refinements; /* predictions evaluated in the gradient stage / the refinement stage (statistics) */
  int32_t  mvpIdx;                  /* mvpIdx on return (changes only with an AMVP list in the job; acMvPred = that candidate, :5768-5770) */
  uint64_t cost;                    /* ruiCost */
} vtmhip_affine_me_out;

int vtmhip_xAffineMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                              const int16_t *d_otherPredBase, const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight,
                                              vtmhip_affine_me_out *d_results );
/* The same call for a batch that may hold bi jobs under a CU-level BCW weight (vtmhip_affine_me_job::bcwWeight): a weight of -2 makes the search target -4 org + 5 pred, whose
 * differences leave the packed 16-bit Hadamard levels of the <= 10-bit kernel, so those jobs run in the 32-bit variant, launched beside the packed one (each job belongs to
 * exactly one of the two).  Results as above. */
int vtmhip_xAffineMotionEstimation_bcw_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                  const int16_t *d_otherPredBase, const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight,
                                                  vtmhip_affine_me_out *d_results );
/* InterPrediction::xPredAffineBlk, luma, uni-directional (rounded and clipped; PROF as flagged) at the jobs' `mv` models */
int vtmhip_xPredAffineBlk_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_refBase, int16_t *d_dstBase,
                                     const vtmhip_affine_me_job *d_jobs, int n, int maxWidth, int maxHeight );

/* ---- symmetric MVD search (SMVD) of predInterSearch (InterSearch.cpp:2656-2790) -----------------------------------------------------
 * One job = one PU with its two symmetric reference pictures: [0] the searched list (list 0 in predInterSearch :2662), [1] the mirrored list.
 *   op VTMHIP_SMVD_COST      InterSearch::xGetSymmetricCost (:4341-4391): cost = floor( fWeight * HAD-or-SAD( clip?( 2 * org - predA( mvCur ) ), predB( mvTar ) ) ),
 *                            each vector clipped (clipMv), integer vectors = the reconstruction, others xPredInterBlk (uni-directional, rounded, clipped);
 *                            BCW weights other than the default use removeWeightHighFreq and fWeight = |w| / 8 (xGetMEDistortionWeight :7666-7676)
 *   op VTMHIP_SMVD_ME        InterSearch::xSymmetricMotionEstimation (:4506-4518) with xSymmeticRefineMvSearch (:4393-4503): 8 >> imv diamond rounds and one
 *                            cross round of one AMVR step around mvCur, the mirrored vector predSym[1] - ( mv - predSym[0] ); in / out mvCur, mvTar, cost
 *   op VTMHIP_SMVD_CHECK_MVP InterSearch::symmvdCheckBestMvp (:7787-7886) for curMv = mvCur: every (i, j) of the two AMVP lists (skip != 0: but the current
 *                            pair); in / out predSym, mvpIdxSym, cost
 *   op VTMHIP_SMVD_SEARCH    the whole block :2656-2790: shortened AMVP lists (:2668-2671), best predictor pair, the distinct start vectors (starts[0..numFixed)
 *                            as they are = cMvHevcTemp, cMvTemp[, cMvBi]; the rest = m_uniMvList entries newest first, rounded to the AMVR precision, while
 *                            fewer than 5 are collected), symmvdCheckBestMvp per start, the search, the final predictor check, + getCost( modeBits );
 *                            out mvCur, mvTar, predSym, mvpIdxSym, cost (= symCost, to be compared with uiCostBi by the caller)
 * No MCTS constraint, no weighted prediction / RPR / wrap-around (the host keeps those PUs). */
#define VTMHIP_SMVD_COST 0
#define VTMHIP_SMVD_ME 1
#define VTMHIP_SMVD_CHECK_MVP 2
#define VTMHIP_SMVD_SEARCH 3
#define VTMHIP_SMVD_UNIFORM 0x100  /* or-ed into op: every job is exactly maxWidth x maxHeight (8x8 .. 16x16, bitDepth <= 10: the lane-per-tile kernel) */
#define VTMHIP_SMVD_MAX_START 18   /* cMvHevcTemp, cMvTemp, cMvBi + the 15 entries of m_uniMvList */
typedef struct
{
  int64_t  orgOff;                 /* origBuf.Y() inside d_orgBase */
  int64_t  refOff[2];              /* PU position with MV (0,0) inside d_refBase: [0] searched list, [1] mirrored list */
  int32_t  orgStride, refStride[2];
  int16_t  puX, puY, width, height;
  uint8_t  imv;                    /* cu.imv: 0 quarter, 1 integer, 2 four-sample, 3 half (alternative half-sample filter) */
  uint8_t  useSatd;                /* !slice.getDisableSATDForRD() */
  uint8_t  clipBiPred;             /* cfg ClipForBiPredMEEnabled */
  int8_t   bcwWeightTar;           /* getBcwWeight( cu.BcwIdx, mirrored list ): 0 or 4 = default */
  uint8_t  numCand[2];             /* AMVP lists of the two symmetric references */
  uint8_t  numStart, numFixed;     /* SEARCH: start vectors */
  uint8_t  skip;                   /* CHECK_MVP */
  uint8_t  pad_[3];
  int32_t  cand[2][2][2];          /* [list][i][hor / ver] */
  uint32_t mvpIdxBits[2];          /* m_auiMVPIdxCost[i][AMVP_MAX_NUM_CANDS] */
  uint32_t modeBits;               /* SEARCH: uiMbBits[2] + 1 + BCW index bits (:2782-2785) */
  double   motionLambda;
  int32_t  starts[VTMHIP_SMVD_MAX_START][2];
  /* in / out */
  int32_t  mvCur[2], mvTar[2];
  int32_t  predSym[2][2];          /* cMvPredSym */
  int32_t  mvpIdxSym[2];
  uint64_t cost;
  /* op VTMHIP_SMVD_SEARCH only: the state of the block at the points where the reference's code sits between two member calls (a host that replays predInterSearch's
   * own glue over device results -- INTEGRATION.md section 3.1 -- serves the members from these):
   *   [0] after the predictor-pair loop (:2675-2691): cost = the winning pair's xGetSymmetricCost (no rate), idx = the pair
   *   [1] after the start-vector loop (:2744-2763):   mv = cCurMvField.mv, idx = mvpIdxSym, cost = costStart
   *   [2] after xSymmetricMotionEstimation (:2770):   mv = cCurMvField.mv, cost = symCost as the member returns it (without mvpCost)
   *   [3] after the final predictor check (:2774-2777): idx = mvpIdxSym, cost = symCost (with mvpCost, before the mode bits) */
  struct { uint64_t cost; int32_t mv[2]; int32_t idx[2]; } trace[4];
} vtmhip_smvd_job;
int vtmhip_smvd_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs, int n,
                           int maxWidth, int maxHeight, int op );
/* the reference's names for the four ops */
int vtmhip_xGetSymmetricCost_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs,
                                        int n, int maxWidth, int maxHeight );
int vtmhip_xSymmetricMotionEstimation_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase,
                                                 vtmhip_smvd_job *d_jobs, int n, int maxWidth, int maxHeight );
int vtmhip_symmvdCheckBestMvp_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, vtmhip_smvd_job *d_jobs,
                                         int n, int maxWidth, int maxHeight );

/* ================================================================================================================
 * (3) LEVEL-ORDER DRIVER SUPPORT -- whole functions per batch and the glue between them, decisions kept on the device
 * ============================================================================================================== */

/* ---- merge-candidate estimation: the SATD pre-selection of EncCu::xCheckRDCostMerge2Nx2N (hook B10) ---------------------------------------
 * (EncCu.cpp:2399-2440; the MMVD and CIIP candidate loops :2470-2549 use the same two steps).  For every candidate the reference runs
 * motionCompensation( pu, ..., luma ) -- plain uni / bi prediction, BDOF or DMVR (+ BDOF) by the candidate's motion -- and the Hadamard (or SAD)
 * distortion against the original block; the candidate's bits x sqrt(lambda) and the sorted insertion (updateCandList) stay with the host.
 * One call: the three job tables (candidates grouped by the kind of prediction; any may be empty) -> predictions into d_predBase + predOff (kept:
 * they are the encoder's acMergeBuffer) -> d_dist[nPlain + nBdof + nDmvr] in table order.  Jobs: epilogue 0; d_mvd as vtmhip_dmvr_batch_dev. */
int vtmhip_merge_cand_satd_batch_dev( vtmhip_ctx *ctx, const vtmhip_pic_params *pic, const int16_t *d_orgBase, const int16_t *d_refBase, int16_t *d_predBase,
                                      const vtmhip_pred_job *d_plain, int nPlain, const vtmhip_pred_job *d_bdof, int nBdof, const vtmhip_dmvr_job *d_dmvr, int nDmvr,
                                      int32_t *d_mvd, int maxWidth, int maxHeight, int uniformSize, int useSatd, uint64_t *d_dist );

/* ================================================================================================================
 * (4) LEVEL-ORDER predInterSearch -- InterSearch::predInterSearch (InterSearch.cpp:2245-3065, translational part: cu.imv 0, default
 *     BCW weight, no SMVD / affine) for ALL PUs of one block size at once, on top of the batched calls above
 * ==============================================================================================================
 * One PU of the reference runs: for every (list, refIdx): xEstimateMvPredAMVP, xMotionEstimation, xCheckBestMVP, best reference per list
 * (:2354-2450); B slices: the list with the larger cost is refined against the other list's prediction for every refIdx (one iteration:
 * FASTINTERSEARCH_MODE1, :2544-2556), xCheckBestMVP, then the uni / bi decision (:2846-2893).  A level-order driver issues the same steps as
 * batched launches over all PUs of a level; these helpers are the per-PU arithmetic in between (one thread per PU / row), so that nothing
 * returns to the host.  Rows of the uni tables are ordered (list, refIdx)-major, PU-minor: row = (list ? numRef[0] : 0) + refIdx) * numPU + pu;
 * rows of the bi tables: refIdx * numPU + pu.  What stands in for the CU recursion: the two AMVP candidates of a row are the enclosing
 * parent block's vector for the same (list, refIdx) and the zero vector (the reference derives them from neighbouring CUs). */
#define VTMHIP_MAX_REF 4

typedef struct
{
  int32_t  mvHor, mvVer;           /* cMvTemp[list][refIdx] */
  int32_t  mvPredHor, mvPredVer;   /* cMvPred[list][refIdx] after xCheckBestMVP */
  int32_t  mvpIdx;                 /* aaiMvpIdx */
  uint32_t bits;                   /* uiBitsTemp */
  uint64_t cost;                   /* uiCostTemp */
} vtmhip_pis_row;

typedef struct
{
  uint64_t cost[2];                /* uiCost[list] (max when the list is empty) */
  uint64_t costBi;                 /* uiCostBi */
  uint32_t bits[3];                /* uiBits */
  int32_t  refIdx[2];              /* iRefIdx */
  int32_t  mv[2][2];               /* cMv */
  int32_t  refIdxBi[2];            /* iRefIdxBi */
  int32_t  mvBi[2][2];             /* cMvBi */
  int32_t  refineList;             /* the list the bi stage searched */
  int32_t  interDir;               /* pu.interDir: 1 list 0, 2 list 1, 3 bi-prediction */
  int32_t  smvdMode;               /* symMode (:2789): 1 when the symmetric-MVD pair replaced the bi vectors, else 0 */
  int32_t  mvpIdxL1Zero;           /* MvdL1Zero pictures: bestBiPMvpL1 (:2382-2387), the list-1 predictor index of the bi mode (its vector IS the predictor: mvBi[1], refIdxBi[1]) */
  int32_t  pad;
} vtmhip_pis_pu;

/* per-PU state of the CU recursion that predInterSearch reads */
typedef struct
{
  uint8_t  noSmvd;                 /* !trySmvd (:2301) */
  uint8_t  uniMvInsert;            /* insertUniMvCands runs between the uni loop and the bi stage (:2451-2459: cu.imv == 0, default BCW): the bi searches and the SMVD
                                      start list see the PU's own uni vectors in m_uniMvList */
  uint8_t  uniMvSelfIsNew;         /* with uniMvInsert: the block has no entry yet -- its vectors become the newest entry (of 15 entries the oldest drops out) */
  int8_t   bcwWeightL1;            /* g_BcwWeights[cu.BcwIdx] = getBcwWeight( cu.BcwIdx, REF_PIC_LIST_1 ) in {-2, 3, 4, 5, 10}; 0 or 4: the default weight.  Another weight (list 0 then
                                      weighs 8 - w): the bi stage refines the list with the SMALLER |weight| (:2556-2559) against the weighted target, prices with |w| / 8, skips the refined
                                      list's pictures that have the other list's POC (bcwFastSkipPoc, :2588-2593) and ENFORCES the bi mode (enforceBcwPred :2543, 2843-2847) */
  int16_t  uniMvSelfPos;           /* with uniMvInsert && !uniMvSelfIsNew: position (newest first, 0 .. 14) of the block's entry, overwritten in place (InterSearch.h:247-275) */
  uint8_t  bcwIdxBits;             /* getWeightIdxBits( cu.BcwIdx ) when sps.getUseBcw(), else 0: joins every bi row's bits and the SMVD mode bits (:2595, 2781) */
  uint8_t  bcwFastSkipPoc;         /* m_pcEncCfg->getUseBcwFast() && slice.getTLayer() > 1 (the device adds: weight != default && cu.imv == 0) */
} vtmhip_pis_pu_in;                /* 8 bytes as in ABI 5 (uniMvSelfPos was an int32 holding 0 .. 14: records of ABI 5 read unchanged, with the BCW fields 0) */

typedef struct
{
  int32_t  numPU;
  int32_t  numRef[2];              /* slice.getNumRefIdx(); numRef[1] == 0: P slice */
  int32_t  smvdBit;                /* slice.getBiDirPred(): one more bit on the bi rows (:2590-2593) */
  uint32_t mbBits[3];              /* xGetBlkBits (:3164-3169) */
  int32_t  refStride;
  int64_t  refPlaneOff[2][VTMHIP_MAX_REF];   /* sample offset of each reference plane's (0,0) inside d_refBase */
  vtmhip_me_job        *uniJobs;   /* [(numRef[0] + numRef[1]) * numPU]: static fields by the host, candidates / bits by stage 0 */
  const vtmhip_me_out  *uniOut;
  vtmhip_pis_row       *uniRows;
  vtmhip_pis_pu        *pus;       /* [numPU] */
  vtmhip_pred_job      *predOther; /* [numPU]: static fields by the host; mode / refOff / mv by stage 2 */
  vtmhip_me_job        *biJobs;    /* [numRef[refined list] * numPU] (numRef[0] == numRef[1] in B slices here) */
  const vtmhip_me_out  *biOut;
  vtmhip_pred_job      *predFinal; /* [numPU] */
  const int32_t        *parentIdx; /* [numPU] PU index in the parent level, or -1 (NULL: no parent level) */
  const vtmhip_pis_row *parentRows;/* the parent level's uniRows */
  int32_t  parentNumPU, pad;
  const int64_t        *pos;       /* [numPU] y * refStride + x */
  /* final prediction of the chosen mode (InterPrediction::motionCompensation with luma and chroma): BDOF where xPredInterBi applies it (:527-572) and
   * the two 4:2:0 chroma planes */
  int32_t  bdofEnabled;            /* sps.getBDOFEnabledFlag() && !picHeader.getDisBdofFlag() */
  int32_t  curPoc;
  int32_t  refPoc[2][VTMHIP_MAX_REF];
  vtmhip_pred_job      *predFinalC;/* [2 * numPU] or NULL: the Cb jobs, then the Cr jobs (static fields by the host; mode / refOff / mv by the final stage) */
  const int64_t        *posC;      /* [numPU] (y / 2) * refStrideC + x / 2 */
  int64_t  refPlaneOffC[2][2][VTMHIP_MAX_REF];   /* [Cb / Cr][list][refIdx] sample offset of the chroma plane's (0,0) inside d_refBase */
  /* affine uni stage (predAffineInterSearch's uni loop :4480-4660, 4-parameter model) for PUs of at least 16x16: one xAffineMotionEstimation job per
   * (PU, list, refIdx) row, starting from / predicted by the row's translational result (the affine AMVP list needs the CU recursion: stand-in as for AMVP) */
  vtmhip_affine_me_job *affJobs;   /* [(numRef[0] + numRef[1]) * numPU] or NULL; filled by stage 4 */
  int32_t  affLowDelay;            /* m_pcEncCfg->getIntraPeriod() == -1 (refinement rounds) */
  int32_t  affCheckLDC;            /* slice.getCheckLDC(): PROF's large-gradient rule */
  /* SMVD block (:2656-2790) between the bi refinement and the uni / bi decision, when the slice has a symmetric reference pair (smvdBit != 0): one
   * vtmhip_smvd_job per PU, searched list = list 0 */
  vtmhip_smvd_job      *smvdJobs;  /* [numPU] or NULL; filled by stage 3, searched by vtmhip_smvd_batch_dev( VTMHIP_SMVD_SEARCH ), merged by stage 5 */
  int32_t  symRefIdx[2];           /* slice.getSymRefIdx( list ) */
  /* ---- the caller's CU context instead of the level-order stand-ins (vtmhip_predInterSearch_batch_dev: one call = predInterSearch of n real PUs) ---- */
  int32_t  candsGiven;             /* != 0: uniJobs already hold the real AMVP lists (PU::fillMvpCand: amvpCand, numAmvpCand 1 or 2), mvpIdxBits, the m_uniMvList start vectors
                                      (numExtraStart / extraStart), imv, flags and the entry bits (mbBits + reference-index bits): stage 0 is skipped */
  int32_t  biRestricted;           /* PU::isBipredRestriction (8x4 / 4x8): no bi stage, no SMVD */
  int32_t  list1FromList0[VTMHIP_MAX_REF]; /* slice.getList1IdxToList0Idx( refIdx ) + 1: v > 0: list-1 picture refIdx is list-0 picture v - 1.  Such a row never becomes the uni-directional
                                      list-1 result (mvValidList1 / costValidList1, :2438-2446, 2826-2829); with fastMEForGenBLowDelay it is not searched either.  0: a picture of its own */
  const vtmhip_pis_pu_in *puIn;    /* [numPU] or NULL */
  vtmhip_pis_row       *biRows;    /* [numRef[refined list] * numPU] or NULL: the bi rows after xCheckBestMVP (cMvTemp, cMvPredBi, aaiMvpIdxBi, bits, cost) */
  uint64_t             *distBiP;   /* [(numRef[0] + numRef[1]) * numPU] or NULL: *puiDistBiP of xEstimateMvPredAMVP per row */
  int32_t  mvdL1Zero;              /* picHeader.getMvdL1ZeroFlag() (both lists hold the same pictures): the bi mode takes list 1 AT its best AMVP predictor (smallest template cost over
                                      the list-1 rows, :2382-2387, 2477-2522: needs distBiP) and refines list 0 only (:2576-2580) */
  int32_t  fastMEForGenBLowDelay;  /* cfg FDM (:2391-2404): the rows of list-1 pictures that are list-0 pictures too take the list-0 vector and a re-priced cost instead of a search */
  int32_t  givenRows;              /* bit ( list ? numRef[0] : 0 ) + refIdx: every row of that (list, refIdx) carries VTMHIP_MEJ_GIVEN_UNI -- the group is not searched */
  int32_t  picW, picH, ctuSize;    /* picW != 0: the other list's vector is clipped as motionCompensation does it (clipMv, InterPrediction.cpp:445-470) before the prediction for the bi
                                      refinement is formed -- search results are inside the clip range by construction, an AMVP predictor (MvdL1Zero) need not be */
} vtmhip_pis_level;


/* stage 0: AMVP candidates and entry bits of the uni rows (before vtmhip_xEstimateMvPredAMVP_batch_dev)
 * stage 1: after the uni searches: xCheckBestMVP per row, best reference per list; P slices: interDir and predFinal
 * stage 2: B slices: refined list, predOther (the other list's prediction, epilogue as the host set it), the bi rows
 * stage 3: after the bi searches: xCheckBestMVP, best bi row, decision, predFinal; with smvdJobs: the SMVD job of every PU instead of the decision (start
 *          vectors cMvHevcTemp / cMvTemp / cMvBi of list 0's symmetric reference; the m_uniMvList history is CU-recursion state and stays empty)
 * stage 5: (smvdJobs != NULL) after the SMVD search: symCost < uiCostBi replaces the bi vectors (:2787-2803), then the decision and predFinal (no BDOF for an
 *          SMVD pair, InterPrediction.cpp:552-555)
 * stage 4: (affJobs != NULL) the affine uni jobs of every row: start vector and predictor = the row's translational result at all control points,
 *          bits = the row's bits before the vector rate, hevcCost = the PU's best translational cost -> vtmhip_xAffineMotionEstimation_batch_dev */
int vtmhip_pis_stage( vtmhip_ctx *ctx, const vtmhip_pis_level *lvl, int stage );

/* ---- the level-order picture loop itself, natively -------------------------------------------------------------------------------------------
 * One call runs the whole chain of a picture over the levels' tables (what vtm_amd/pipeline.py:FrameHotPath.run does call by call from Python:
 * ~130 library calls, 2 ms of interpreter time per picture -- more than a GPU needs for its share of a picture sharded over eight GPUs).
 * Per level: stage 0, xEstimateMvPredAMVP, the uni searches, stage 1 on `mainStream` (a level's AMVP candidates are its parent's vectors: one dependent
 * chain); everything after that -- stage 2, the other list's prediction, the bi refinement, stage 3, (the SMVD search, stage 5,) the final prediction (plain / BDOF / chroma),
 * the affine uni stage, the TU chains -- on sideStreams[level % numSide] behind an event, beside the next levels' searches; the side streams join
 * `mainStream` at the end.  numSide == 0: everything on mainStream in level order.  The context's stream is mainStream on return. */
typedef struct
{
  vtmhip_pis_level  pis;
  vtmhip_pic_params pic, picBi;        /* wavesPerJob tuned per level for the uni / bi searches */
  vtmhip_me_cfg     cfgUni, cfgBi;
  int32_t  width, height;              /* the level's PU shape */
  int32_t  bdof;                       /* launch vtmhip_bdof_batch_dev over predFinal after the plain prediction */
  int32_t  pad0;
  vtmhip_me_out        *uniOut;        /* = pis.uniOut, writable */
  vtmhip_me_out        *biOut;
  /* luma TU chains: `numCands` candidate runs of `numTU` jobs each, stored one after the other in tu[] / tuRes[]; cand[i] = 1: transform skip */
  vtmhip_tu_job        *tu;
  vtmhip_tu_result     *tuRes;
  int32_t              *qcoef;
  int32_t  numTU, numCands, tuW, tuH;
  uint8_t  cand[8];
  /* chroma TU chains (NULL: none): 2 * numTUC jobs (Cb then Cr) */
  vtmhip_tu_job        *tuC;
  vtmhip_tu_result     *tuResC;
  int32_t              *qcoefC;
  int32_t  numTUC, tuWC, tuHC, pad1;
  vtmhip_affine_me_out *affOut;        /* affine uni stage when pis.affJobs != NULL */
  uint8_t *mtsTest;                    /* [numCands * numTU] or NULL: the MTS pre-selection flags of the luma candidates (vtmhip_mts_select_batch_dev) */
  int32_t  mtsMaxCand, pad2;           /* cfg MTSInterMaxCand */
} vtmhip_pis_level_run;

typedef struct
{
  const int16_t *org;                  /* original picture (Y | Cb | Cr) */
  const int16_t *dpb;                  /* reference pictures */
  int16_t *pred, *resi;                /* level-wide luma sample buffers (compact per-PU slots, offsets in the job tables) */
  int16_t *orgBi;                      /* 2 * org - pred of the other list (B slices) */
  int16_t *predC, *resiC;              /* chroma (NULL: luma only) */
} vtmhip_pis_buffers;

int vtmhip_pis_run_picture( vtmhip_ctx *ctx, const vtmhip_pis_level_run *levels, int numLevels, const vtmhip_pis_buffers *buf, void *mainStream,
                            void *const *sideStreams, int numSide );

/* ---- InterSearch::predInterSearch (InterSearch.cpp:2245-2893, the translational part incl. the SMVD block) of n real PUs of one slice and one block shape in ONE call ------
 * The batched hook a CU-level integration uses (oracle/ref_shim_enc.cpp drives it from inside the real encoder): L->pis.candsGiven = 1, the
 * uni rows carry the PUs' real AMVP lists, m_uniMvList start vectors, cu.imv (0 .. 3: fractional or AMVR integer refinement) and block-vector cache hits.  Runs on the
 * context's stream, no host synchronisation:
 *   xEstimateMvPredAMVP per row -> xMotionEstimation per searched row -> xCheckBestMVP, list-1 rows copied from list 0 (FastMEForGenBLowDelay), best reference per list
 *   -> B slices without bi-prediction restriction: the other list's prediction, the bi refinement of the costlier list (one iteration: FASTINTERSEARCH_MODE1 / 2), xCheckBestMVP
 *   -> the SMVD block (L->pis.smvdJobs) -> uni / bi decision.
 * Results: L->pis.uniRows, L->uniOut (every row's vector / bits / cost BEFORE xCheckBestMVP: what m_uniMotions and the block-vector cache store), L->pis.biRows, L->biOut,
 * L->pis.smvdJobs (with the trace), L->pis.pus.  No prediction, no residual coding (predFinal may be NULL); buf->orgBi: numPU * width * height samples of scratch.
 * Covered: P and B slices, cu.imv 0 .. 3, block-vector cache hits, m_uniMvList, FastMEForGenBLowDelay copies, MvdL1Zero pictures (mvdL1Zero, distBiP, mvpIdxL1Zero),
 * bi-prediction restriction, the SMVD block, BCW (ABI 6): the weight-index bits (puIn.bcwIdxBits) and CU-level weights other than the default (puIn.bcwWeightL1, given uni rows).
 * Not covered (the caller keeps the reference path): explicit weighted prediction, four bi iterations (FEN off), MCTS, composite references, hash ME, IBC.
 * Synchronisation: none for the shapes vtmhip_is_uniform_shape() accepts and for every call with fewer than 64 rows (a CU-level hook); a batch of >= 64 rows of another shape
 * (64x8, 128x64, ...) is bucketed by shape inside vtmhip_xMotionEstimation_batch_dev, which synchronises the stream once to read the class counts (never under stream capture). */
int vtmhip_predInterSearch_batch_dev( vtmhip_ctx *ctx, const vtmhip_pis_level_run *L, const vtmhip_pis_buffers *buf );
/* 1 when a batch of width x height blocks may promise cfg.uniformSquare (the tiled kernels know the shape) */
int vtmhip_is_uniform_shape( int width, int height );

#ifdef __cplusplus
}
#endif
#endif /* VTMHIP_H */
