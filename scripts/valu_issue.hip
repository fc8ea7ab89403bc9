// valu_issue.hip -- gfx950 micro-benchmark: issue cost of the integer / packed-16-bit vector opcodes the hot-path kernels are made of.
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/valu_issue scripts/valu_issue.hip && scripts/valu_issue > profiles/rNN_valu_issue.jsonl
//
// MEASUREMENT TOOL (not part of libvtmhip.so).  For every opcode: each wave runs CH independent dependency chains of the SAME instruction
// (CH = 8: throughput; CH = 1: dependent-issue latency), 32 instructions per loop trip, TRIPS trips; W waves per SIMD run concurrently on
// every CU (one or two workgroups per CU, pinned there by their LDS allocation).  Each wave brackets its loop with s_memtime (shader-clock
// ticks, MI355X_MICROARCH.md constants table); reported: cycles per wave-instruction per SIMD = mean(ticks) / (W * instructions) and max/mean (the scheduler favours
// older waves, so with 4+ waves per SIMD they finish at different times: the steady-state issue cost is max(ticks) / (W * instructions)).  A value of
// 2 is the full wave64 rate of a SIMD-32 (32 lanes per cycle), 4 = half rate, 8 = quarter rate.  The DESIGN.md roofline for the VALU-bound
// kernels (`bound: "valu_issue"`) prices each kernel's instruction mix with these numbers.  The "imix" rows answer how MIXED streams issue: a stream
// with any sizeable share of half-rate opcodes runs at 3.7 .. 4.0 cycles per instruction whatever the order (it is NOT the share-weighted sum
// of the two rates), so scripts/isa_mix.py prices a kernel by interpolating these rows at the kernel's half-rate share.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK( x )                                                                                   \
  do {                                                                                               \
    hipError_t e_ = ( x );                                                                           \
    if( e_ != hipSuccess ) { fprintf( stderr, "%s: %s\n", #x, hipGetErrorString( e_ ) ); exit( 1 ); } \
  } while( 0 )

static constexpr int TRIPS = 1500;

// one asm statement = 32 instructions over 8 (or 1) chains; operands: 8 accumulators "+v", two sources "v"
#define REP8( I0, I1, I2, I3, I4, I5, I6, I7 ) I0 I1 I2 I3 I4 I5 I6 I7
#define BODY8( OP )                                                                                                                        \
  asm volatile( REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) )                                             \
                REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) )                                             \
                REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) )                                             \
                REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) )                                             \
                : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] )           \
                : "v"( s0 ), "v"( s1 ) )
#define BODY1( OP )                                                                                                                        \
  asm volatile( REP8( OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ) )                                             \
                REP8( OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ) )                                             \
                REP8( OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ) )                                             \
                REP8( OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ), OP( 0 ) )                                             \
                : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] )           \
                : "v"( s0 ), "v"( s1 ) )

// %0..%7 accumulators, %8 / %9 sources
#define OP_ADD_U32( i )      "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_SUB_U32( i )      "v_sub_u32 %" #i ", %" #i ", %8\n"
#define OP_MAX_I32( i )      "v_max_i32 %" #i ", %" #i ", %8\n"
#define OP_AND_B32( i )      "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_LSHL( i )         "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_ASHR( i )         "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define OP_ADD3( i )         "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_LSHL_ADD( i )     "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define OP_MAX3( i )         "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define OP_FMA_F32( i )      "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PK_ADD_I16( i )   "v_pk_add_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_SUB_I16( i )   "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_MAX_I16( i )   "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_MIN_I16( i )   "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_ADD_U16( i )   "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define OP_PK_MUL_LO( i )    "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define OP_PK_MAD_I16( i )   "v_pk_mad_i16 %" #i ", %8, %9, %" #i "\n"
#define OP_PK_ASHR( i )      "v_pk_ashrrev_i16 %" #i ", 1, %" #i "\n"
#define OP_PK_LSHL( i )      "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define OP_PK_SUB_NEG( i )   "v_pk_sub_i16 %" #i ", 0, %" #i "\n"
#define OP_SAD_U16( i )      "v_sad_u16 %" #i ", %8, %9, %" #i "\n"
#define OP_SAD_U32( i )      "v_sad_u32 %" #i ", %8, %9, %" #i "\n"
#define OP_SAD_U8( i )       "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define OP_DOT2C_I16( i )    "v_dot2c_i32_i16 %" #i ", %8, %9\n"
#define OP_DOT2_I16( i )     "v_dot2_i32_i16 %" #i ", %8, %9, %" #i "\n"
#define OP_DOT4_I8( i )      "v_dot4_i32_i8 %" #i ", %8, %9, %" #i "\n"
#define OP_MAD_I24( i )      "v_mad_i32_i24 %" #i ", %8, %9, %" #i "\n"
#define OP_MAD_U24( i )      "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define OP_MUL_I24( i )      "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define OP_MUL_LO_U32( i )   "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL_HI_U32( i )   "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_MAD_U64_U32( i )  "v_mad_u64_u32 v[100:101], vcc, %" #i ", %8, v[100:101]\n"
#define OP_PERM( i )         "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ALIGNBIT( i )     "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"
#define OP_BFE_I32( i )      "v_bfe_i32 %" #i ", %" #i ", 0, 16\n"
#define OP_BFI( i )          "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define OP_CNDMASK( i )      "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMP_LT( i )       "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define OP_MOV( i )          "v_mov_b32 %" #i ", %8\n"
#define OP_MOV_DPP_QP( i )   "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_DPP_ROWSHR( i ) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_DPP_MIRROR( i ) "v_add_u32_dpp %" #i ", %" #i ", %8 row_mirror row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_SDWA( i )     "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
#define OP_READLANE( i )     "v_readlane_b32 s20, %" #i ", 3\n"
#define OP_CVT_F64( i )      "v_cvt_f64_i32 v[100:101], %" #i "\n"
#define OP_MUL_F64( i )      "v_mul_f64 v[100:101], v[100:101], v[102:103]\n"
#define OP_SNOP( i )         "s_nop 0\n"
#define OP_SADD( i )         "s_add_u32 s20, s20, 1\n"
#define OP_XOR( i )          "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_OR( i )           "v_or_b32 %" #i ", %" #i ", %8\n"
#define OP_LSHR( i )         "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define OP_MAX_U32( i )      "v_max_u32 %" #i ", %" #i ", %8\n"
#define OP_MIN_I32( i )      "v_min_i32 %" #i ", %" #i ", %8\n"
#define OP_SUBREV( i )       "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL_I24_SDWA( i ) "v_mul_i32_i24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define OP_SUB_SDWA( i )     "v_sub_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define OP_LSHL_ADD_U64( i ) "v_lshl_add_u64 v[100:101], v[100:101], 1, v[102:103]\n"
#define OP_MOV_B64( i )      "v_mov_b64 v[100:101], v[102:103]\n"
#define OP_LSHL_OR( i )      "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define OP_MED3( i )         "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN_U32_DPP( i )  "v_min_u32_dpp %" #i ", %" #i ", %8 row_shr:2 row_mask:0xf bank_mask:0xf\n"
#define OP_CNDMASK_S( i )    "v_cndmask_b32 %" #i ", %" #i ", %8, s[22:23]\n"
#define OP_CMP_E64( i )      "v_cmp_lt_i32 s[22:23], %" #i ", %8\n"
#define OP_CMP_CND( i )      "v_cmp_lt_i32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_ADDC( i )         "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define OP_BFE_U32( i )      "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define OP_ABS_PAIR( i )     "v_pk_sub_i16 %8, 0, %" #i "\nv_pk_max_i16 %" #i ", %" #i ", %8\n"
// mixes of the kernels' inner loops
#define OP_MIX_PERM_DOT( i ) "v_perm_b32 %" #i ", %" #i ", %8, %9\nv_dot2c_i32_i16 %" #i ", %8, %9\n"   /* dependent perm -> dot2c pair, 8 chains */
#define OP_MIX_ADDSUB( i )   "v_pk_add_i16 %" #i ", %" #i ", %8\nv_pk_sub_i16 %" #i ", %" #i ", %9\n"
#define OP_MIX_VS( i )       "v_pk_add_i16 %" #i ", %" #i ", %8\ns_add_u32 s20, s20, 1\n"               /* VALU + SALU pairs: do they co-issue? */

#define OP_MIX_FH( i )       "v_add_u32 %" #i ", %" #i ", %8\nv_pk_add_i16 %" #i ", %" #i ", %9\n"             /* full-rate + half-rate alternating */
#define OP_MIX_FFFH( i )     "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %9\nv_sub_u32 %" #i ", %" #i ", %8\nv_pk_add_i16 %" #i ", %" #i ", %9\n"
#define OP_MIX_FHHH( i )     "v_add_u32 %" #i ", %" #i ", %8\nv_pk_sub_i16 %" #i ", %" #i ", %9\nv_pk_max_i16 %" #i ", %" #i ", %8\nv_pk_add_i16 %" #i ", %" #i ", %9\n"

template<int OPID, int CH>
__global__ void __launch_bounds__( 1024 ) bench_kernel( uint32_t *sink, uint64_t *ticks, uint32_t seed )
{
  extern __shared__ uint32_t lds[];   // only pins the number of workgroups per CU
  uint32_t a[8];
#pragma unroll
  for( int i = 0; i < 8; i++ ) a[i] = seed * ( threadIdx.x + 1 ) + i * 0x10001u;
  uint32_t s0 = seed ^ 0x00030005u, s1 = seed + 0x00010002u;
  if( seed == 0xdeadbeefu ) lds[threadIdx.x] = a[0];
  __syncthreads();
  uint64_t t0 = __builtin_readcyclecounter();   // s_memtime
  asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
  for( int t = 0; t < TRIPS; t++ )
  {
#define CASE( ID, OP ) if constexpr( OPID == ID ) { if constexpr( CH == 8 ) BODY8( OP ); else BODY1( OP ); }
    CASE( 0, OP_ADD_U32 ) CASE( 1, OP_SUB_U32 ) CASE( 2, OP_MAX_I32 ) CASE( 3, OP_AND_B32 ) CASE( 4, OP_LSHL ) CASE( 5, OP_ASHR )
    CASE( 6, OP_ADD3 ) CASE( 7, OP_LSHL_ADD ) CASE( 8, OP_MAX3 ) CASE( 9, OP_FMA_F32 )
    CASE( 10, OP_PK_ADD_I16 ) CASE( 11, OP_PK_SUB_I16 ) CASE( 12, OP_PK_MAX_I16 ) CASE( 13, OP_PK_MIN_I16 ) CASE( 14, OP_PK_ADD_U16 )
    CASE( 15, OP_PK_MUL_LO ) CASE( 16, OP_PK_MAD_I16 ) CASE( 17, OP_PK_ASHR ) CASE( 18, OP_PK_LSHL ) CASE( 19, OP_PK_SUB_NEG )
    CASE( 20, OP_SAD_U16 ) CASE( 21, OP_SAD_U32 ) CASE( 22, OP_SAD_U8 ) CASE( 23, OP_DOT2C_I16 ) CASE( 24, OP_DOT2_I16 ) CASE( 25, OP_DOT4_I8 )
    CASE( 26, OP_MAD_I24 ) CASE( 27, OP_MAD_U24 ) CASE( 28, OP_MUL_I24 ) CASE( 29, OP_MUL_LO_U32 ) CASE( 30, OP_MUL_HI_U32 )
    CASE( 32, OP_PERM ) CASE( 33, OP_ALIGNBIT ) CASE( 34, OP_BFE_I32 ) CASE( 35, OP_BFI ) CASE( 36, OP_CNDMASK ) CASE( 37, OP_CMP_LT )
    CASE( 38, OP_MOV ) CASE( 39, OP_MOV_DPP_QP ) CASE( 40, OP_ADD_DPP_ROWSHR ) CASE( 41, OP_ADD_DPP_MIRROR ) CASE( 42, OP_ADD_SDWA )
    CASE( 46, OP_SNOP ) CASE( 48, OP_MIX_PERM_DOT ) CASE( 49, OP_MIX_ADDSUB )
    CASE( 51, OP_XOR ) CASE( 52, OP_OR ) CASE( 53, OP_LSHR ) CASE( 54, OP_MAX_U32 ) CASE( 55, OP_MIN_I32 ) CASE( 56, OP_SUBREV ) CASE( 57, OP_MUL_I24_SDWA )
    CASE( 58, OP_SUB_SDWA ) CASE( 61, OP_LSHL_OR ) CASE( 62, OP_MED3 ) CASE( 63, OP_MIN_U32_DPP ) CASE( 67, OP_ADDC ) CASE( 68, OP_BFE_U32 )
#undef CASE
    if constexpr( OPID == 43 ) { asm volatile( REP8( OP_READLANE( 0 ), OP_READLANE( 1 ), OP_READLANE( 2 ), OP_READLANE( 3 ), OP_READLANE( 4 ), OP_READLANE( 5 ), OP_READLANE( 6 ), OP_READLANE( 7 ) ) REP8( OP_READLANE( 0 ), OP_READLANE( 1 ), OP_READLANE( 2 ), OP_READLANE( 3 ), OP_READLANE( 4 ), OP_READLANE( 5 ), OP_READLANE( 6 ), OP_READLANE( 7 ) ) REP8( OP_READLANE( 0 ), OP_READLANE( 1 ), OP_READLANE( 2 ), OP_READLANE( 3 ), OP_READLANE( 4 ), OP_READLANE( 5 ), OP_READLANE( 6 ), OP_READLANE( 7 ) ) REP8( OP_READLANE( 0 ), OP_READLANE( 1 ), OP_READLANE( 2 ), OP_READLANE( 3 ), OP_READLANE( 4 ), OP_READLANE( 5 ), OP_READLANE( 6 ), OP_READLANE( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) : "s20" ); }
    if constexpr( OPID == 47 ) { asm volatile( REP8( OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ) ) REP8( OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ) ) REP8( OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ) ) REP8( OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ), OP_SADD( 0 ) ) ::: "s20", "scc" ); }
    if constexpr( OPID == 50 ) { asm volatile( REP8( OP_MIX_VS( 0 ), OP_MIX_VS( 1 ), OP_MIX_VS( 2 ), OP_MIX_VS( 3 ), OP_MIX_VS( 4 ), OP_MIX_VS( 5 ), OP_MIX_VS( 6 ), OP_MIX_VS( 7 ) ) REP8( OP_MIX_VS( 0 ), OP_MIX_VS( 1 ), OP_MIX_VS( 2 ), OP_MIX_VS( 3 ), OP_MIX_VS( 4 ), OP_MIX_VS( 5 ), OP_MIX_VS( 6 ), OP_MIX_VS( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) : "s20", "scc" ); }
#define SPECIAL( ID, OP, ... ) if constexpr( OPID == ID ) { asm volatile( REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) ) REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) ) REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) ) REP8( OP( 0 ), OP( 1 ), OP( 2 ), OP( 3 ), OP( 4 ), OP( 5 ), OP( 6 ), OP( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ), "+v"( s1 ) : "v"( s0 ) : __VA_ARGS__ ); }
    SPECIAL( 59, OP_LSHL_ADD_U64, "v100", "v101", "v102", "v103" )
    SPECIAL( 60, OP_MOV_B64, "v100", "v101", "v102", "v103" )
    SPECIAL( 64, OP_CNDMASK_S, "s22", "s23" )
    SPECIAL( 65, OP_CMP_E64, "s22", "s23" )
    SPECIAL( 66, OP_CMP_CND, "vcc" )
    SPECIAL( 69, OP_ABS_PAIR, "vcc" )
    SPECIAL( 70, OP_MIX_FH, "vcc" )
    SPECIAL( 71, OP_MIX_FFFH, "vcc" )
    SPECIAL( 72, OP_MIX_FHHH, "vcc" )
    // independent mixes: every chain keeps ONE opcode; neighbouring instructions of a wave never depend on each other
#define MIXI( ID, O0, O1, O2, O3, O4, O5, O6, O7 ) if constexpr( OPID == ID ) { asm volatile( REP8( O0( 0 ), O1( 1 ), O2( 2 ), O3( 3 ), O4( 4 ), O5( 5 ), O6( 6 ), O7( 7 ) ) REP8( O0( 0 ), O1( 1 ), O2( 2 ), O3( 3 ), O4( 4 ), O5( 5 ), O6( 6 ), O7( 7 ) ) REP8( O0( 0 ), O1( 1 ), O2( 2 ), O3( 3 ), O4( 4 ), O5( 5 ), O6( 6 ), O7( 7 ) ) REP8( O0( 0 ), O1( 1 ), O2( 2 ), O3( 3 ), O4( 4 ), O5( 5 ), O6( 6 ), O7( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) ); }
    MIXI( 73, OP_ADD_U32, OP_PK_ADD_I16, OP_ADD_U32, OP_PK_ADD_I16, OP_ADD_U32, OP_PK_ADD_I16, OP_ADD_U32, OP_PK_ADD_I16 )
    MIXI( 74, OP_ADD_U32, OP_SUB_U32, OP_XOR, OP_PK_ADD_I16, OP_ADD_U32, OP_SUB_U32, OP_XOR, OP_PK_SUB_I16 )
    MIXI( 75, OP_ADD_U32, OP_PK_ADD_I16, OP_PK_SUB_I16, OP_PK_MAX_I16, OP_SUB_U32, OP_PK_ADD_I16, OP_PK_SUB_I16, OP_PK_MAX_I16 )
    MIXI( 76, OP_ADD_U32, OP_MAX_I32, OP_SUB_U32, OP_MAX_I32, OP_ADD_U32, OP_MAX_I32, OP_SUB_U32, OP_MAX_I32 )
    MIXI( 77, OP_ADD_U32, OP_ADD_U32, OP_ADD_U32, OP_ADD_U32, OP_PK_ADD_I16, OP_PK_ADD_I16, OP_PK_ADD_I16, OP_PK_ADD_I16 )
    MIXI( 80, OP_ADD_U32, OP_SUB_U32, OP_XOR, OP_AND_B32, OP_ADD_U32, OP_SUB_U32, OP_OR, OP_PK_ADD_I16 )
    MIXI( 78, OP_ADD_U32, OP_SUB_U32, OP_ADD_U32, OP_SUB_U32, OP_XOR, OP_AND_B32, OP_ASHR, OP_OR )
    MIXI( 79, OP_PK_ADD_I16, OP_MAX_I32, OP_PK_SUB_I16, OP_SAD_U16, OP_PERM, OP_MAD_I24, OP_LSHL, OP_BFE_I32 )
    if constexpr( OPID == 31 ) { asm volatile( REP8( OP_MAD_U64_U32( 0 ), OP_MAD_U64_U32( 1 ), OP_MAD_U64_U32( 2 ), OP_MAD_U64_U32( 3 ), OP_MAD_U64_U32( 4 ), OP_MAD_U64_U32( 5 ), OP_MAD_U64_U32( 6 ), OP_MAD_U64_U32( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) : "v100", "v101", "vcc" ); }
    if constexpr( OPID == 44 ) { asm volatile( REP8( OP_CVT_F64( 0 ), OP_CVT_F64( 1 ), OP_CVT_F64( 2 ), OP_CVT_F64( 3 ), OP_CVT_F64( 4 ), OP_CVT_F64( 5 ), OP_CVT_F64( 6 ), OP_CVT_F64( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) : "v100", "v101" ); }
    if constexpr( OPID == 45 ) { asm volatile( REP8( OP_MUL_F64( 0 ), OP_MUL_F64( 1 ), OP_MUL_F64( 2 ), OP_MUL_F64( 3 ), OP_MUL_F64( 4 ), OP_MUL_F64( 5 ), OP_MUL_F64( 6 ), OP_MUL_F64( 7 ) ) : "+v"( a[0] ), "+v"( a[1] ), "+v"( a[2] ), "+v"( a[3] ), "+v"( a[4] ), "+v"( a[5] ), "+v"( a[6] ), "+v"( a[7] ) : "v"( s0 ), "v"( s1 ) : "v100", "v101", "v102", "v103" ); }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
  uint32_t acc = 0;
#pragma unroll
  for( int i = 0; i < 8; i++ ) acc ^= a[i];
  if( acc == 0x12345678u ) sink[threadIdx.x] = acc;
  if( ( threadIdx.x & 63 ) == 0 ) ticks[( blockIdx.x * blockDim.x + threadIdx.x ) >> 6] = t1 - t0;
}

struct Op { int id; const char *name; int instPerTrip; };
static const Op OPS[] = {
  { 0, "v_add_u32", 32 }, { 1, "v_sub_u32", 32 }, { 2, "v_max_i32", 32 }, { 3, "v_and_b32", 32 }, { 4, "v_lshlrev_b32", 32 }, { 5, "v_ashrrev_i32", 32 },
  { 6, "v_add3_u32", 32 }, { 7, "v_lshl_add_u32", 32 }, { 8, "v_max3_i32", 32 }, { 9, "v_fma_f32", 32 },
  { 10, "v_pk_add_i16", 32 }, { 11, "v_pk_sub_i16", 32 }, { 12, "v_pk_max_i16", 32 }, { 13, "v_pk_min_i16", 32 }, { 14, "v_pk_add_u16", 32 },
  { 15, "v_pk_mul_lo_u16", 32 }, { 16, "v_pk_mad_i16", 32 }, { 17, "v_pk_ashrrev_i16", 32 }, { 18, "v_pk_lshlrev_b16", 32 }, { 19, "v_pk_sub_i16(0-x)", 32 },
  { 20, "v_sad_u16", 32 }, { 21, "v_sad_u32", 32 }, { 22, "v_sad_u8", 32 }, { 23, "v_dot2c_i32_i16", 32 }, { 24, "v_dot2_i32_i16", 32 }, { 25, "v_dot4_i32_i8", 32 },
  { 26, "v_mad_i32_i24", 32 }, { 27, "v_mad_u32_u24", 32 }, { 28, "v_mul_i32_i24", 32 }, { 29, "v_mul_lo_u32", 32 }, { 30, "v_mul_hi_u32", 32 }, { 31, "v_mad_u64_u32", 8 },
  { 32, "v_perm_b32", 32 }, { 33, "v_alignbit_b32", 32 }, { 34, "v_bfe_i32", 32 }, { 35, "v_bfi_b32", 32 }, { 36, "v_cndmask_b32", 32 }, { 37, "v_cmp_lt_i32", 32 },
  { 38, "v_mov_b32", 32 }, { 39, "v_mov_b32_dpp quad_perm", 32 }, { 40, "v_add_u32_dpp row_shr", 32 }, { 41, "v_add_u32_dpp row_mirror", 32 }, { 42, "v_add_u32_sdwa", 32 },
  { 43, "v_readlane_b32", 32 }, { 44, "v_cvt_f64_i32", 8 }, { 45, "v_mul_f64", 8 }, { 46, "s_nop 0", 32 }, { 47, "s_add_u32", 32 },
  { 51, "v_xor_b32", 32 }, { 52, "v_or_b32", 32 }, { 53, "v_lshrrev_b32", 32 }, { 54, "v_max_u32", 32 }, { 55, "v_min_i32", 32 }, { 56, "v_subrev_u32", 32 },
  { 57, "v_mul_i32_i24_sdwa", 32 }, { 58, "v_sub_u32_sdwa", 32 }, { 59, "v_lshl_add_u64", 32 }, { 60, "v_mov_b64", 32 }, { 61, "v_lshl_or_b32", 32 }, { 62, "v_med3_i32", 32 },
  { 63, "v_min_u32_dpp row_shr", 32 }, { 64, "v_cndmask_b32 (sgpr-pair mask)", 32 }, { 65, "v_cmp_lt_i32 e64 -> sgpr pair", 32 },
  { 66, "pair: v_cmp_lt_i32 vcc + v_cndmask_b32 vcc", 64 }, { 67, "v_addc_co_u32", 32 }, { 68, "v_bfe_u32", 32 }, { 69, "pair: |x| packed = v_pk_sub_i16(0-x) + v_pk_max_i16", 64 },
  { 70, "mix: v_add_u32 + v_pk_add_i16 (1 full : 1 half)", 64 }, { 71, "mix: 3 full-rate + 1 half-rate", 128 }, { 72, "mix: 1 full-rate + 3 half-rate", 128 },
  { 73, "imix: add_u32 / pk_add_i16 alternating, independent chains (1 full : 1 half)", 32 }, { 74, "imix: 3 full : 1 half, independent chains", 32 },
  { 75, "imix: 1 full : 3 half, independent chains", 32 }, { 76, "imix: add_u32 / max_i32 alternating (1 full : 1 half)", 32 },
  { 77, "imix: 4 add_u32 then 4 pk_add_i16 (runs of four)", 32 }, { 78, "imix: eight different full-rate opcodes", 32 }, { 79, "imix: eight different half-rate opcodes", 32 },
  { 80, "imix: 7 full : 1 half, independent chains", 32 },
  { 48, "mix: v_perm_b32 -> v_dot2c_i32_i16 (dependent pair)", 64 }, { 49, "mix: v_pk_add_i16 + v_pk_sub_i16", 64 }, { 50, "mix: v_pk_add_i16 + s_add_u32 (pairs)", 32 },
};

template<int OPID> static void launch( int ch, int grid, int block, size_t lds, uint32_t *sink, uint64_t *ticks )
{
  if( ch == 8 ) hipLaunchKernelGGL( ( bench_kernel<OPID, 8> ), dim3( grid ), dim3( block ), lds, 0, sink, ticks, 12345u );
  else          hipLaunchKernelGGL( ( bench_kernel<OPID, 1> ), dim3( grid ), dim3( block ), lds, 0, sink, ticks, 12345u );
}

template<int... IDS> struct Seq {};
template<int ID> static bool tryLaunch( int id, int ch, int grid, int block, size_t lds, uint32_t *sink, uint64_t *ticks )
{
  if( id != ID ) return false;
  launch<ID>( ch, grid, block, lds, sink, ticks );
  return true;
}
template<int... IDS> static void dispatch( Seq<IDS...>, int id, int ch, int grid, int block, size_t lds, uint32_t *sink, uint64_t *ticks )
{
  bool done = ( tryLaunch<IDS>( id, ch, grid, block, lds, sink, ticks ) || ... );
  if( !done ) { fprintf( stderr, "no such op %d\n", id ); exit( 1 ); }
}
template<int N, int... IDS> struct MakeSeq : MakeSeq<N - 1, N - 1, IDS...> {};
template<int... IDS> struct MakeSeq<0, IDS...> { using type = Seq<IDS...>; };

int main( int argc, char **argv )   // optional arguments: the op ids to run (default: all)
{
  hipDeviceProp_t prop;
  CHECK( hipGetDeviceProperties( &prop, 0 ) );
  const int nCU = prop.multiProcessorCount;
  uint32_t *sink; uint64_t *ticks;
  CHECK( hipMalloc( &sink, 4096 ) );
  CHECK( hipMalloc( &ticks, sizeof( uint64_t ) * nCU * 64 ) );
  std::vector<uint64_t> h( nCU * 64 );
  const int wavesPerSimd[] = { 1, 2, 4, 8 };
  for( const Op &op : OPS )
    for( int ch : { 8, 1 } )
    {
      bool wanted = argc <= 1;
      for( int k = 1; k < argc; k++ ) wanted |= atoi( argv[k] ) == op.id;
      if( !wanted ) continue;
      if( ch == 1 && ( op.id == 31 || ( op.id >= 43 && op.id <= 50 ) || op.id == 59 || op.id == 60 || op.id >= 64 ) ) continue;
      for( int w : wavesPerSimd )
      {
        // w waves per SIMD = 4 w waves per CU: one workgroup of 256 w threads (w <= 4) or two of 1024 (w = 8); LDS pins them one / two per CU
        const int wgPerCU = w == 8 ? 2 : 1;
        const int block   = 64 * 4 * w / wgPerCU;
        const size_t lds  = wgPerCU == 1 ? 96 * 1024 : 64 * 1024;
        const int grid    = nCU * wgPerCU;
        hipEvent_t e0, e1;
        CHECK( hipEventCreate( &e0 ) ); CHECK( hipEventCreate( &e1 ) );
        dispatch( MakeSeq<81>::type(), op.id, ch, grid, block, lds, sink, ticks );   // warm-up (clocks, code fetch)
        CHECK( hipEventRecord( e0 ) );
        dispatch( MakeSeq<81>::type(), op.id, ch, grid, block, lds, sink, ticks );
        CHECK( hipEventRecord( e1 ) );
        CHECK( hipDeviceSynchronize() );
        float ms = 0;
        CHECK( hipEventElapsedTime( &ms, e0, e1 ) );
        const int nw = grid * block / 64;
        CHECK( hipMemcpy( h.data(), ticks, sizeof( uint64_t ) * nw, hipMemcpyDeviceToHost ) );
        double sum = 0; uint64_t mx = 0;
        for( int i = 0; i < nw; i++ ) { sum += ( double ) h[i]; if( h[i] > mx ) mx = h[i]; }
        const double inst = ( double ) op.instPerTrip * TRIPS;
        printf( "{\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.3f, \"cycles_per_inst_one_wave\": %.3f, "
                "\"max_over_mean\": %.3f, \"kernel_ms\": %.4f, \"implied_MHz\": %.0f}\n",
                op.name, ch, w, sum / nw / ( w * inst ), sum / nw / inst, mx / ( sum / nw ), ms, ( double ) mx / ( ms * 1e3 ) );
        fflush( stdout );
        CHECK( hipEventDestroy( e0 ) ); CHECK( hipEventDestroy( e1 ) );
      }
    }
  return 0;
}
