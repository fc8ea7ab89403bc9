// Throughput of 16-byte global loads per lane at 2-byte, 4-byte, 8-byte and 16-byte alignment (the motion-search kernels read 8-sample segments of a reference
// block at the candidate's sample offset: 2-byte aligned).  Each wave walks rows of a 128-sample-wide block (16 lanes per row, 4 rows per load instruction) of an
// L2-resident plane; prints GB/s per alignment.  hipcc --offload-arch=gfx950 -O2 -o unaligned_load unaligned_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct __attribute__( ( packed, aligned( 2 ) ) ) Pel8 { unsigned v[4]; };

__global__ __launch_bounds__( 256 ) void walk( const int16_t *plane, int stride, int offSamples, int rows, int iters, unsigned *out )
{
  const int lane = threadIdx.x & 63, wave = ( blockIdx.x * 256 + threadIdx.x ) >> 6;
  const int16_t *p = plane + ( long ) ( ( wave * 7 ) & 255 ) * stride + ( ( wave * 40 ) & 1023 ) + offSamples + ( lane >> 4 ) * stride + ( lane & 15 ) * 8;
  unsigned s = 0;
  for( int it = 0; it < iters; it++ )
  {
    const int16_t *q = p + ( long ) ( it & 7 ) * 5 * stride;
    for( int r = 0; r < rows; r += 16 )
    {
      const Pel8 a = *reinterpret_cast<const Pel8 *>( q + ( long ) r * stride ), b = *reinterpret_cast<const Pel8 *>( q + ( long ) ( r + 4 ) * stride );
      const Pel8 c = *reinterpret_cast<const Pel8 *>( q + ( long ) ( r + 8 ) * stride ), d = *reinterpret_cast<const Pel8 *>( q + ( long ) ( r + 12 ) * stride );
#pragma unroll
      for( int k = 0; k < 4; k++ ) s += __builtin_amdgcn_sad_u16( a.v[k], b.v[k], 0 ) + __builtin_amdgcn_sad_u16( c.v[k], d.v[k], 0 );
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
  const int stride = 4096, H = 1024;
  std::vector<int16_t> h( ( size_t ) stride * H );
  for( size_t i = 0; i < h.size(); i++ ) h[i] = ( int16_t ) ( i * 2654435761u >> 22 );
  int16_t *d; unsigned *o;
  hipMalloc( &d, h.size() * 2 ); hipMalloc( &o, 4096 * 256 * 4 );
  hipMemcpy( d, h.data(), h.size() * 2, hipMemcpyHostToDevice );
  hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
  const int blocks = 2048, rows = 64, iters = 64;
  for( int off : { 0, 1, 2, 4, 3, 7 } )
  {
    walk<<<blocks, 256>>>( d, stride, off, rows, 4, o );
    hipDeviceSynchronize();
    hipEventRecord( e0 );
    walk<<<blocks, 256>>>( d, stride, off, rows, iters, o );
    hipEventRecord( e1 ); hipEventSynchronize( e1 );
    float ms; hipEventElapsedTime( &ms, e0, e1 );
    const double bytes = ( double ) blocks * 4 * 64 * 16 * ( rows / 4 ) * iters;
    printf( "{\"offset_samples\": %d, \"alignment_bytes\": %d, \"ms\": %.3f, \"GBps\": %.1f}\n", off, off == 0 ? 16 : ( off * 2 ) & -( off * 2 ), ms, bytes / ms / 1e6 );
  }
  return 0;
}
