// lanc2.hip -- LANCIR exact 2x RGBA float (cfg5 of BASELINE.json), vertical +
// horizontal Lanczos passes fused in ONE launch.
//
// CLancIR at k = 0.5, la = 3 (lancir.h:430-483, 882-1351): KernelLen 6, two
// filter phases (frac 0.75 for even outputs, 0.25 for odd ones); per axis
//   o[2q]   = dot6( fa, in[q-3 .. q+2] ),   o[2q+1] = dot6( fb, in[q-2 .. q+3] )
// with indices clamped to the image (copyScanline*v / padScanline*h edge
// replication), and the x86 summation order of resize4 (lancir.h:2466-2544):
//   dot6 = ((f0*p0 + f2*p2) + f4*p4) + ((f1*p1 + f3*p3) + f5*p5).
// LANCIR is vertical-first: the vertical pass produces a float intermediate of
// SrcWidth x NewHeight which the horizontal pass then resizes.
//
// Structure: like up2.hip, a workgroup (256 threads, half-pixel = 8-byte units)
// owns a strip of 128 output columns and a chunk of source rows and marches
// down 8 source rows per step:
//   V  one thread per source half-column (70 px incl. halo): the 7-row window
//      lives in registers (ring of 8, statically unrolled); source rows are
//      read straight from HBM, coalesced, prefetched one step ahead; each row
//      yields two intermediate rows -> LDS (16 rows x 70 px = 17.9 KB)
//   H  one thread per output half-column: 6 LDS reads + dot6 per output row,
//      16 rows per step, written to HBM coalesced (2 KiB per row)
// The SrcWidth x NewHeight intermediate never exists in HBM. Compute is light
// (~72 flop per output pixel): the kernel is HBM-bound.

#include "plan.h"
#include <algorithm>
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#ifndef L2_PRIO
#define L2_PRIO 2 // (wave priority by phase: 1 vertical high, 2 horizontal high)
#endif

namespace avirhip {

typedef float f2 __attribute__(( ext_vector_type( 2 )));
typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef float f8 __attribute__(( ext_vector_type( 8 )));

#define L2_TW 128                 // output columns per strip
#define L2_NT ( L2_TW * 2 )       // threads: one per output half pixel
#define L2_RB 8                   // source rows per marching step
#define L2_SW ( L2_TW / 2 + 6 )   // source / intermediate columns incl. halo

struct Lanc2Params
{
	const float* src; long src_ss; int sw, sh;
	int rmin, rmax; // source rows that exist behind `src` (a window: plan.h)
	float* dst; long dst_ss; int dst_row0; int nw, nh;
	int srow_lo, srow_hi;
	int nstrips, chunk0, cq;
	const float* coef; // device: [va 6 | vb 6 | pad 4 | ha 6 | hb 6 | pad 4]
	// IO != 0: the OWNER plan's output stage (outputScanline, lancir.h:
	// 1772-2056: gain, clamps, rounding by the element's position in its
	// scanline) run by the horizontal phase, into the caller's image
	// SRC != 0: the owner's uint8 / uint16 RGBA image read as it is (the
	// (float) cast of copyScanline*v, lancir.h:1541-1594, in the loader)
	const void* raw; long raw_ss; int rch; // elements per row; channels (3, 4)
	void* ibase; long istride; // the band's first row; elements per row
	int ich, l4, unity;        // channels; elements rounded to nearest even
	float out_mul, clampv;
};

struct Taps12
{
	f8 a; f4 b; // fa[0..5] fb[0..1] | fb[2..5]
	__device__ __forceinline__ float fa( int i ) const { return( a[ i ]); }
	__device__ __forceinline__ float fb( int i ) const
		{ return( i < 2 ? a[ 6 + i ] : b[ i - 2 ]); }
};

__device__ __forceinline__ Taps12 load_taps12( const float* p )
{
	Taps12 t;
	asm volatile( "s_load_dwordx8 %0, %2, 0x0\n\t"
		"s_load_dwordx4 %1, %2, 0x20\n\t"
		AVIRHIP_WAITCNT_LGKM( 0 )
		: "=&s"( t.a ), "=&s"( t.b ) : "s"( p ) : "memory" );
	return( t );
}

// resize4's x86 order: even and odd taps accumulate separately.
#define L2_DOT6( F, p0, p1, p2, p3, p4, p5 ) \
	((( F( 0 ) * ( p0 ) + F( 2 ) * ( p2 )) + F( 4 ) * ( p4 )) + \
	(( F( 1 ) * ( p1 ) + F( 3 ) * ( p3 )) + F( 5 ) * ( p5 )))

// LORD: 4 the order above; 3 RGB images' (resize3, lancir.h:2284-2464, as
// gp_dot of gpass_dev.h spells it: sums over taps 4g + j, the two tail taps of
// a 6-tap kernel entering differently in channel 0).
template< int LORD, class FA >
__device__ __forceinline__ f2 l2_dot6( FA F, const bool ch0, const f2 p0,
	const f2 p1, const f2 p2, const f2 p3, const f2 p4, const f2 p5 )
{
	if constexpr( LORD == 4 )
	{
		return((( F( 0 ) * p0 + F( 2 ) * p2 ) + F( 4 ) * p4 ) +
			(( F( 1 ) * p1 + F( 3 ) * p3 ) + F( 5 ) * p5 ));
	}
	else
	{
		f2 a0 = F( 0 ) * p0;
		const f2 a1 = F( 1 ) * p1, a2 = F( 2 ) * p2, a3 = F( 3 ) * p3;
		const f2 t0 = F( 4 ) * p4, t1 = F( 5 ) * p5;
		a0 = a0 + t0;
		f2 r = (( a0 + a1 ) + ( a2 + a3 )) + t1;
		const f2 r0 = ( a0 + ( a1 + t1 )) + ( a2 + a3 );

		if( ch0 )
		{
			r.x = r0.x;
		}

		return( r );
	}
}

// IO: 0 the float RGBA result; 1 uint8, 2 uint16, 3 float (gain and / or
// fewer than four channels) elements of the owner plan's image.
// SRC: 0 float RGBA; 1 uint8, 2 uint16 pixels (RGBA or RGB: P.rch) of the
// owner's image.
template< int IO, int SRC, int LORD >
__global__ void __launch_bounds__( L2_NT ) k_lanc2( const Lanc2Params P )
{
	// SRC 1 / 2: uint8 / uint16 RGBA; 3 / 4: uint8 / uint16 RGB (compile-time:
	// as a run-time channel count the RGBA form lost a third of its speed)
	constexpr int RCH = ( SRC == 3 || SRC == 4 ? 3 : 4 );
	constexpr bool S8 = ( SRC == 1 || SRC == 3 );
	constexpr bool S16 = ( SRC == 2 || SRC == 4 );
	// intermediate rows of this step, half-pixel units: [16][L2_SW * 2]
	__shared__ __attribute__(( aligned( 16 ))) f2 sT[ 2 * L2_RB * L2_SW * 2 ];

	const int nwg = gridDim.x;
	const int b = blockIdx.x;
	const int xcd = b & 7;
	const int qd = nwg >> 3;
	const int rm = nwg & 7;
	const int item = ( xcd < rm ? xcd * ( qd + 1 ) :
		rm * ( qd + 1 ) + ( xcd - rm ) * qd ) + ( b >> 3 );

	const int strip = item % P.nstrips;
	const int chunk = P.chunk0 + item / P.nstrips;
	const int tid = threadIdx.x;
	const int hf = tid & 1;

	const int qx0 = strip * ( L2_TW / 2 );
	const int qy0 = chunk * P.cq;
	const int qy1 = min( qy0 + P.cq, P.nh >> 1 );
	const int u0 = qy0 - 3;            // first source row index consumed
	const int nsteps = qy1 - qy0 + 6;  // 6 warm-up rows + chunk rows

	// vertical phase: this thread's source half-column (threads >= 2*L2_SW idle)
	const bool vact = ( tid < L2_SW * 2 );
	const int sx = min( max( qx0 - 3 + ( tid >> 1 ), 0 ), P.sw - 1 );
	const unsigned scol = (unsigned) ( sx * 4 + hf * 2 );
	const unsigned rcol = (unsigned) ( sx * RCH + hf * 2 ); // (SRC != 0)

	// horizontal phase: this thread's output half-column
	const int xo = qx0 * 2 + ( tid >> 1 );
	const bool xok = ( xo < P.nw );
	const unsigned dlane = (unsigned) ( xo * 4 + hf * 2 );
	const int hq = ( tid >> 2 );       // pair index within the strip
	const int hodd = ( tid >> 1 ) & 1; // odd output column?
	const int ylo = max( qy0 * 2, P.srow_lo );
	const int yhi = min( qy1 * 2, P.srow_hi );

	f2 ring[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ ) ring[ i ] = (f2) 0.0f;

	f2 pre[ L2_RB ];

	auto prefetch = [&]( const int ub )
	{
		if( vact )
		{
#pragma unroll
			for( int r = 0; r < L2_RB; r++ )
			{
				const int sy = min( max( ub + r, P.rmin ), P.rmax );

				if constexpr( S8 )
				{
					const unsigned char* const q = (const unsigned char*) P.raw +
						( (unsigned) sy * (unsigned) P.raw_ss + rcol );

					if constexpr( RCH == 4 )
					{
						const unsigned short w = *(const unsigned short*) q;
						pre[ r ].x = (float) ( w & 0xffu );
						pre[ r ].y = (float) ( w >> 8 );
					}
					else
					{
						// RGB: channels 0, 1 | channel 2 and the zero padding
						pre[ r ].x = (float) q[ 0 ];
						pre[ r ].y = ( hf == 0 ? (float) q[ 1 ] : 0.0f );
					}
				}
				else
				if constexpr( S16 )
				{
					const unsigned short* const q = (const unsigned short*) P.raw +
						( (unsigned) sy * (unsigned) P.raw_ss + rcol );

					if constexpr( RCH == 4 )
					{
						const unsigned int w = *(const unsigned int*) q;
						pre[ r ].x = (float) ( w & 0xffffu );
						pre[ r ].y = (float) ( w >> 16 );
					}
					else
					{
						pre[ r ].x = (float) q[ 0 ];
						pre[ r ].y = ( hf == 0 ? (float) q[ 1 ] : 0.0f );
					}
				}
				else
				{
					pre[ r ] = *(const f2*) ( P.src + ( (unsigned) sy *
						(unsigned) P.src_ss + scol ));
				}
			}
		}
	};

	prefetch( u0 );

	for( int sb = 0; sb < nsteps; sb += L2_RB )
	{
		const int ub = u0 + sb;

		// ---- V: 8 source rows -> 16 intermediate rows (ring phase == row & 7)
		__builtin_amdgcn_s_setprio( L2_PRIO == 1 ? 1 : 0 );
		if( vact )
		{
			const Taps12 V = load_taps12( P.coef );
#define VA( i ) V.fa( i )
#define VB( i ) V.fb( i )
#pragma unroll
			for( int rr = 0; rr < L2_RB; rr++ )
			{
				ring[ rr ] = pre[ rr ];
				// rows u-6 .. u of the ring, u = ub + rr
				const f2 m6 = ring[ ( rr - 6 ) & 7 ], m5 = ring[ ( rr - 5 ) & 7 ];
				const f2 m4 = ring[ ( rr - 4 ) & 7 ], m3 = ring[ ( rr - 3 ) & 7 ];
				const f2 m2 = ring[ ( rr - 2 ) & 7 ], m1 = ring[ ( rr - 1 ) & 7 ];
				const f2 m0 = ring[ rr ];
				sT[ ( 2 * rr ) * ( L2_SW * 2 ) + tid ] = l2_dot6< LORD >(
					[&]( const int i ) { return( V.fa( i )); }, hf == 0,
					m6, m5, m4, m3, m2, m1 );
				sT[ ( 2 * rr + 1 ) * ( L2_SW * 2 ) + tid ] = l2_dot6< LORD >(
					[&]( const int i ) { return( V.fb( i )); }, hf == 0,
					m5, m4, m3, m2, m1, m0 );
			}
#undef VA
#undef VB
		}

		AVIRHIP_BARRIER_DRAIN();
		__syncthreads();
		// (wave priority by phase, as in k_up2: the phase that carries the
		// step's arithmetic and its stores -- here the horizontal one -- runs
		// one level above the other; same-box A/B of both orders against none:
		// 1080p -> 4K RGBA uint8 0.0402 / 0.0390 / 0.0368 ms (none / vertical
		// high / horizontal high), RGB uint8 0.0540 / 0.0522 / 0.0506, cfg5
		// 0.1088 / 0.1083 / 0.1072; profiles/r06_up2_prio)
		__builtin_amdgcn_s_setprio( L2_PRIO == 1 ? 0 : 1 );

		// next step's source rows start their trip from HBM now
		if( sb + L2_RB < nsteps )
		{
			prefetch( ub + L2_RB );
		}

		// ---- H: 16 intermediate rows -> 16 output rows of this strip
		{
			const Taps12 H = load_taps12( P.coef + 16 );
#define HA( i ) H.fa( i )
#define HB( i ) H.fb( i )
			// even output 2q reads columns q-3..q+2 -> local q..q+5;
			// odd output 2q+1 reads q-2..q+3 -> local q+1..q+6
			const f2* base = &sT[ ( hq + hodd ) * 2 + hf ];
#pragma unroll 4
			for( int r = 0; r < 2 * L2_RB; r++ )
			{
				const int y = ( ub + ( r >> 1 ) - 3 ) * 2 + ( r & 1 );
				const f2* t = base + r * ( L2_SW * 2 );
				const f2 p0 = t[ 0 ], p1 = t[ 2 ], p2 = t[ 4 ];
				const f2 p3 = t[ 6 ], p4 = t[ 8 ], p5 = t[ 10 ];
				f2 o;

				if( hodd )
				{
					o = l2_dot6< LORD >( [&]( const int i ) { return( H.fb( i )); },
						hf == 0, p0, p1, p2, p3, p4, p5 );
				}
				else
				{
					o = l2_dot6< LORD >( [&]( const int i ) { return( H.fa( i )); },
						hf == 0, p0, p1, p2, p3, p4, p5 );
				}

				if( xok && y >= ylo && y < yhi )
				{
					if constexpr( IO == 0 )
					{
						// streaming store (never re-read): nt, measured -10 %
						__builtin_nontemporal_store( o, (f2*) ( P.dst +
							(long) ( y - P.dst_row0 ) * P.dst_ss + dlane ));
					}
					else
					if constexpr( SRC != 0 && ( IO == 1 || IO == 2 ))
					{
					if constexpr( RCH == 4 )
					{
						// An integer RGBA image in, an integer RGBA image out:
						// every value is finite (either spelling of the clamps
						// gives the same number), a scanline is whole groups of
						// four elements (nearest even throughout), and the
						// lane's two channels are adjacent elements: one store.
						float v0 = o.x, v1 = o.y;

						if( !P.unity )
						{
							v0 = v0 * P.out_mul;
							v1 = v1 * P.out_mul;
						}

						const unsigned int q0 = (unsigned int) (int) rintf(
							fminf( fmaxf( v0, 0.0f ), P.clampv ));
						const unsigned int q1 = (unsigned int) (int) rintf(
							fminf( fmaxf( v1, 0.0f ), P.clampv ));
						const long ei = (long) ( y - P.dst_row0 ) * P.istride +
							xo * 4 + hf * 2;

						if constexpr( IO == 1 )
						{
							*(unsigned short*) ( (unsigned char*) P.ibase + ei ) =
								(unsigned short) ( q0 | ( q1 << 8 ));
						}
						else
						{
							*(unsigned int*) ( (unsigned short*) P.ibase + ei ) =
								q0 | ( q1 << 16 );
						}
					}
					else
					{
						// RGB: finite values again, but the last 1-3 elements of
						// a scanline round by + 0.5 and a lane's channels are
						// stored one by one (channel 3 is padding)
						const int e0 = xo * 3 + hf * 2;
						const long ro = (long) ( y - P.dst_row0 ) * P.istride;
						const float vv[ 2 ] = { o.x, o.y };
#pragma unroll
						for( int c = 0; c < 2; c++ )
						{
							if( hf * 2 + c < 3 )
							{
								float v = vv[ c ];

								if( !P.unity )
								{
									v = v * P.out_mul;
								}

								v = fminf( fmaxf( v, 0.0f ), P.clampv );
								const unsigned int q = (unsigned int) (int) (
									e0 + c < P.l4 ? rintf( v ) : v + 0.5f );

								if constexpr( IO == 1 )
								{
									((unsigned char*) P.ibase )[ ro + e0 + c ] =
										(unsigned char) q;
								}
								else
								{
									((unsigned short*) P.ibase )[ ro + e0 + c ] =
										(unsigned short) q;
								}
							}
						}
					}
					}
					else
					{
						// this lane's two channels: elements e0, e0 + 1 of the
						// scanline (as gp_store_lancir of gpass_dev.h spells the
						// stage out: the SSE2 form rounds whole groups of four
						// to nearest even, the 1-3 element tail by + 0.5)
						const int e0 = xo * P.ich + hf * 2;
						const long ro = (long) ( y - P.dst_row0 ) * P.istride;
						const float vv[ 2 ] = { o.x, o.y };
#pragma unroll
						for( int c = 0; c < 2; c++ )
						{
							if( hf * 2 + c < P.ich )
							{
								float v = vv[ c ];

								if( !P.unity )
								{
									v = v * P.out_mul;
								}

								if constexpr( IO == 3 )
								{
									((float*) P.ibase )[ ro + e0 + c ] = v;
								}
								else
								{
									unsigned int q;

									if( e0 + c < P.l4 )
									{
										v = ( v < P.clampv ? v : P.clampv );
										v = ( v > 0.0f ? v : 0.0f );
										q = (unsigned int) (int) rintf( v );
									}
									else
									{
										q = (unsigned int) (int) (( v > P.clampv ?
											P.clampv : ( v < 0.0f ? 0.0f : v )) + 0.5f );
									}

									if constexpr( IO == 1 )
									{
										((unsigned char*) P.ibase )[ ro + e0 + c ] =
											(unsigned char) q;
									}
									else
									{
										((unsigned short*) P.ibase )[ ro + e0 + c ] =
											(unsigned short) q;
									}
								}
							}
						}
					}
				}
			}
#undef HA
#undef HB
		}

		AVIRHIP_BARRIER_DRAIN();
		__syncthreads();
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct Lanc2Data
{
	float* d_coef;
};

static bool match_lanc_axis( const LancirAxisDev& A, float* fa, float* fb )
{
	if( A.kernel_len != 6 || A.dst_len != A.src_len * 2 || A.n_filters != 2 )
	{
		return( false );
	}

	const int ia = A.h_fidx[ 0 ];
	const int ib = ( A.dst_len > 1 ? A.h_fidx[ 1 ] : 1 - ia );

	if( ia == ib )
	{
		return( false );
	}

	for( int j = 0; j < A.dst_len; j++ )
	{
		const int q = j >> 1;

		if( A.h_start[ j ] != ( j & 1 ? q - 2 : q - 3 ) ||
			A.h_fidx[ j ] != ( j & 1 ? ib : ia ))
		{
			return( false );
		}
	}

	memcpy( fa, &A.h_flt[ (size_t) ia * 6 ], 6 * sizeof( float ));
	memcpy( fb, &A.h_flt[ (size_t) ib * 6 ], 6 * sizeof( float ));
	return( true );
}

int lanc2_prepare( avirhip_plan* p )
{
	p -> lanc2 = nullptr;

	if( !p -> is_lancir || p -> ch != 4 ||
		( p -> l_order != 4 && p -> l_order != 3 ) ||
		p -> in_type != AVIRHIP_F32 ||
		p -> out_type != AVIRHIP_F32 || !p -> l_unity ||
		p -> new_stride != p -> new_w * 4 )
	{
		return( AVIRHIP_OK );
	}

	float hc[ 32 ] = { 0 };

	if( !match_lanc_axis( p -> lv, hc, hc + 6 ) ||
		!match_lanc_axis( p -> lh, hc + 16, hc + 22 ))
	{
		return( AVIRHIP_OK );
	}

	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, sizeof( hc )));
	p -> allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, hc, sizeof( hc ), hipMemcpyHostToDevice ));
	Lanc2Data* D = new Lanc2Data();
	D -> d_coef = (float*) q;
	p -> lanc2 = D;
	p -> fused_ok |= 4;
	p -> auto_path = 4;
	return( AVIRHIP_OK );
}

void lanc2_release( avirhip_plan* p )
{
	delete (Lanc2Data*) p -> lanc2;
	p -> lanc2 = nullptr;
}

// Whether the exact-2x kernel of an inner plan reads its owner's image itself
// (uint8 / uint16 RGBA; the conditions lanc2_run checks again).
bool lanc2_takes_raw( const avirhip_plan* q, const void* raw, int type, int ch,
	long stride )
{
	const int lpath = ( q -> path != 0 ? q -> path : q -> auto_path );

	return( q -> lanc2 != nullptr && q -> is_lancir && lpath == 4 &&
		( ch == 4 || ch == 3 ) && ch == q -> l_order &&
		( type == AVIRHIP_U8 || type == AVIRHIP_U16 ) &&
		( ch == 3 || ( stride & 1 ) == 0 ) &&
		( (uintptr_t) raw & ( ch == 3 ? ( type == AVIRHIP_U8 ? 0 : 1 ) :
		( type == AVIRHIP_U8 ? 1 : 3 ))) == 0 &&
		(long) q -> src_h * stride < ( 1L << 31 ) &&
		(long) q -> new_w * 4 < ( 1L << 30 ));
}

// The plain float RGBA call lanc2_run cannot refuse (its own checks with io ==
// 0, sk == 0): a source window may then be handed over as a virtual frame base.
bool lanc2_takes_window( const avirhip_plan* p, const void* src, const void* dst )
{
	return( p -> lanc2 != nullptr && p -> is_lancir && p -> inner == nullptr &&
		!p -> lfuse.on && !p -> lraw.on && p -> ch == 4 && p -> io_ch == 4 &&
		p -> in_type == AVIRHIP_F32 && p -> out_type == AVIRHIP_F32 &&
		p -> l_unity && dst != nullptr &&
		( (uintptr_t) src & 7 ) == 0 && ( (uintptr_t) dst & 7 ) == 0 &&
		( p -> src_stride & 1 ) == 0 &&
		(long) p -> src_h * p -> src_stride < ( 1L << 31 ) &&
		(long) p -> new_w * 4 < ( 1L << 30 ));
}

// p -> lfuse.on (an inner plan, set by its owner around the call): the
// horizontal phase runs the owner's output stage and stores into its image;
// dst is not used then. Sets p -> lfuse.done.
int lanc2_run( avirhip_plan* p, const float* src, float* dst, int row0,
	int row1, hipStream_t st )
{
	const Lanc2Data* D = (const Lanc2Data*) p -> lanc2;
	const int io = ( !p -> lfuse.on ? 0 :
		( p -> lfuse.type == AVIRHIP_U8 ? 1 :
		( p -> lfuse.type == AVIRHIP_U16 ? 2 :
		( p -> lfuse.type == AVIRHIP_F32 ? 3 : -1 ))));

	if( io < 0 || ( io == 0 && dst == nullptr ))
	{
		return( 1 );
	}

	// (an inner plan whose owner asked for its image to be read as it is)
	const int sk = ( !p -> lraw.on ? 0 : ( p -> lraw.type == AVIRHIP_U8 ? 1 :
		( p -> lraw.type == AVIRHIP_U16 ? 2 : -1 )));

	const bool r4 = ( p -> lraw.ch == 4 );

	if( sk < 0 || ( sk != 0 && (( p -> lraw.ch != 4 && p -> lraw.ch != 3 ) ||
		p -> lraw.ch != p -> l_order ||
		( r4 && sk == 1 && ( (uintptr_t) p -> lraw.ptr & 1 )) ||
		( r4 && ( p -> lraw.stride & 1 )) ||
		( r4 && sk == 2 && ( (uintptr_t) p -> lraw.ptr & 3 )) ||
		( !r4 && sk == 2 && ( (uintptr_t) p -> lraw.ptr & 1 )) ||
		(long) p -> src_h * p -> lraw.stride >= ( 1L << 31 ) ||
		( io != 0 && ( p -> lfuse.ch != p -> lraw.ch ||
		( r4 && ( p -> lfuse.stride & 1 )) ||
		( r4 && io == 1 && ( (uintptr_t) p -> lfuse.dst & 1 )) ||
		( io == 2 && ( (uintptr_t) p -> lfuse.dst & ( r4 ? 3 : 1 ))))))))
	{
		return( 1 );
	}

	if( D == nullptr || ( sk == 0 && ( (uintptr_t) src & 7 )) ||
		( io == 0 && ( (uintptr_t) dst & 7 )) ||
		( io == 2 && ( (uintptr_t) p -> lfuse.dst & 1 )) ||
		( io == 3 && ( (uintptr_t) p -> lfuse.dst & 3 )) ||
		// (the float image's pitch only matters when it is the one read: an
		// inner plan reading its owner's integer image was checked on that
		// image's pitch above, as lanc2_takes_raw() did)
		( sk == 0 && (( p -> src_stride & 1 ) ||
		(long) p -> src_h * p -> src_stride >= ( 1L << 31 ))) ||
		(long) p -> new_w * 4 >= ( 1L << 30 ))
	{
		return( 1 );
	}

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	Lanc2Params P;
	P.src = src; P.src_ss = p -> src_stride; P.sw = p -> src_w;
	P.sh = p -> src_h;
	P.rmin = ( p -> win_rows > 0 ? p -> win_first : 0 );
	P.rmax = ( p -> win_rows > 0 ? p -> win_first + p -> win_rows : p -> src_h ) - 1;
	P.dst = dst; P.dst_ss = (long) p -> new_w * 4; P.dst_row0 = row0;
	P.nw = p -> new_w; P.nh = p -> new_h;
	P.srow_lo = row0; P.srow_hi = row1;
	P.nstrips = ( p -> new_w + L2_TW - 1 ) / L2_TW;
	P.coef = D -> d_coef;
	P.raw = p -> lraw.ptr; P.raw_ss = p -> lraw.stride; P.rch = p -> lraw.ch;
	P.ibase = p -> lfuse.dst; P.istride = p -> lfuse.stride;
	P.ich = p -> lfuse.ch; P.l4 = ( p -> new_w * p -> lfuse.ch ) & ~3;
	P.unity = p -> lfuse.unity; P.out_mul = p -> lfuse.out_mul;
	P.clampv = p -> lfuse.clampv;

	// chunk = 8k - 6 source rows (6 warm-up rows per chunk); fill whole rounds
	// of 256 CUs x 8 resident workgroups with chunks of >= 58 rows
	const int slots = 256 * 8;
	int cq = 0;

	for( int rounds = 1; rounds <= 8 && cq == 0; rounds++ )
	{
		const int nch = std::max( 1, rounds * slots / P.nstrips );
		int c = ( p -> src_h + nch - 1 ) / nch;
		c = (( c + 6 + L2_RB - 1 ) / L2_RB ) * L2_RB - 6;

		if( c >= 58 || rounds == 8 )
		{
			cq = std::max( c, 58 );
		}
	}

	const char* ecq = getenv( "AVIRHIP_LANC2_CQ" );

	if( ecq != nullptr && atoi( ecq ) >= 2 )
	{
		cq = (( atoi( ecq ) + 6 + L2_RB - 1 ) / L2_RB ) * L2_RB - 6;
	}

	P.cq = cq;
	const int cr = cq * 2;
	P.chunk0 = row0 / cr;
	const int chunk1 = ( row1 - 1 ) / cr;
	const int items = P.nstrips * ( chunk1 - P.chunk0 + 1 );

#define L2_LAUNCH( IOK, SK ) { if( p -> l_order == 3 ) \
		hipLaunchKernelGGL(( k_lanc2< IOK, SK, 3 > ), dim3( items ), \
		dim3( L2_NT ), 0, st, P ); else \
		hipLaunchKernelGGL(( k_lanc2< IOK, SK, 4 > ), dim3( items ), \
		dim3( L2_NT ), 0, st, P ); }
#define L2_LAUNCH_S( IOK ) switch( sk == 0 ? 0 : ( r4 ? sk : sk + 2 )) { \
		case 0: L2_LAUNCH( IOK, 0 ); break; case 1: L2_LAUNCH( IOK, 1 ); break; \
		case 2: L2_LAUNCH( IOK, 2 ); break; case 3: L2_LAUNCH( IOK, 3 ); break; \
		default: L2_LAUNCH( IOK, 4 ); break; }

	switch( io )
	{
		case 0: L2_LAUNCH_S( 0 ) break;
		case 1: L2_LAUNCH_S( 1 ) break;
		case 2: L2_LAUNCH_S( 2 ) break;
		default: L2_LAUNCH_S( 3 ) break;
	}
#undef L2_LAUNCH_S
#undef L2_LAUNCH

	AVIRHIP_HIPCHECK( hipGetLastError() );
	p -> lfuse.done = ( io != 0 );
	return( AVIRHIP_OK );
}

} // namespace avirhip
