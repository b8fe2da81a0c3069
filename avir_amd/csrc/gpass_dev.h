// gpass_dev.h -- what the pass kernels of gpass.hip and sacc.hip share: the
// lowered axis as the device sees it, the LDS-DMA and wait-count primitives.
#ifndef AVIRHIP_GPASS_DEV_H
#define AVIRHIP_GPASS_DEV_H

#include "plan.h"
#include <vector>

namespace avirhip {

typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef float f2 __attribute__(( ext_vector_type( 2 )));
typedef __attribute__(( address_space( 3 ))) char lds_char;
// Wave-uniform table reads (positions, coefficient rows) go through the
// constant address space: they become scalar loads (lgkmcnt), so they never
// wait on the vector-memory counter the LDS-DMA prefetch keeps busy.
typedef const __attribute__(( address_space( 4 ))) float gp_cfloat;
typedef const __attribute__(( address_space( 4 ))) int gp_cint;

enum { GP_NONE = 0, GP_PRE = 1, GP_POST = 2 };

struct GPAxis
{
	int mode;          // GP_*
	int lanc;          // LANCIR summation order
	int nt;            // taps of every gather output
	int in_len;        // source length along the axis (clamp range)
	int c_len;         // PRE: length of the FIR's output array
	int view_mmax;     // gather input index above which the view reads 0
	int g_len;         // outputs of the gather
	int out_len;       // outputs of the axis
	int e;             // FIR edge pixel count (3)
	float f[ 4 ];      // FIR taps f0..f3
	const int* start;  // [g_len] first gather input index (unclamped)
	const float* coef; // AVIR: [g_len][nt]; LANCIR: [phases][nt]
	const int* fidx;   // LANCIR: [g_len] phase of each output
};

__device__ __forceinline__ int gp_clamp( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes land at lds + l * 16. Inline
// assembly on purpose: the compiler must not know that these loads write LDS,
// or it would drain them (vmcnt 0) before every LDS read that follows --
// completion is awaited explicitly, one block / row later.
__device__ __forceinline__ void gp_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( "s_mov_b32 m0, %0\n\t"
		"buffer_load_dwordx4 %1, %2, 0 offen lds"
		:: "s"( __builtin_amdgcn_readfirstlane( lds )), "v"( voff ), "s"( rs )
		: "memory" );
}

#define GP_WAIT_VM( n ) asm volatile( "s_waitcnt vmcnt(" #n ")" ::: "memory" )

// s_waitcnt takes an immediate: a wave-uniform count picks its instruction.
// Waiting for MORE than `n` to retire is always safe (loads retire in order).
__device__ __forceinline__ void gp_wait_vm( const int n )
{
	switch( n )
	{
		case 0: GP_WAIT_VM( 0 ); break;
		case 1: GP_WAIT_VM( 1 ); break;
		case 2: GP_WAIT_VM( 2 ); break;
		case 3: GP_WAIT_VM( 3 ); break;
		case 4: GP_WAIT_VM( 4 ); break;
		case 5: GP_WAIT_VM( 5 ); break;
		case 6: GP_WAIT_VM( 6 ); break;
		case 7: GP_WAIT_VM( 7 ); break;
		case 8: GP_WAIT_VM( 8 ); break;
		case 9: GP_WAIT_VM( 9 ); break;
		case 10: GP_WAIT_VM( 10 ); break;
		case 11: GP_WAIT_VM( 11 ); break;
		case 12: GP_WAIT_VM( 12 ); break;
		default: GP_WAIT_VM( 12 ); break;
	}
}

// The integer output stage fused into the last pass' store
// (CImageResizerDithererDefINL::dither, avir.h:4392-4419 + unpackScanline,
// avir.h:3155-3215, without gamma): round, optional bit-depth truncation,
// clamp, C cast -- the arithmetic of k_epilogue_px (generic.hip).
struct GPOut
{
	int on;          // 0: float RGBA rows (no output stage)
	int type;        // AVIRHIP_U8 / AVIRHIP_U16; AVIRHIP_F32 with 1-3 channels
	int ch;          // channels of the caller's pixels (1..4)
	int use_tr;
	float tr_mul, tr_muli, pk_out;
	void* base;      // the caller's rows (row 0 = the band's first row)
	long stride;     // elements per row
};

__device__ __forceinline__ float gp_round( const float d )
{
	return( d < 0.0f ? -(float) (int) ( 0.5f - d ) : (float) (int) ( d + 0.5f ));
}

// pixel x of row `row` (both relative to O.base)
__device__ __forceinline__ void gp_store_int( const GPOut& O, const long row,
	const int x, const f4 o )
{
	const float v[ 4 ] = { o.x, o.y, o.z, o.w };

	if( O.type == AVIRHIP_F32 )
	{
		// float pixels of 1-3 channels: the padding is dropped, nothing else
		float* const p = (float*) O.base + row * O.stride + (long) x * O.ch;
		p[ 0 ] = v[ 0 ];
		if( O.ch > 1 ) p[ 1 ] = v[ 1 ];
		if( O.ch > 2 ) p[ 2 ] = v[ 2 ];
		return;
	}

	unsigned int q[ 4 ];
#pragma unroll
	for( int c = 0; c < 4; c++ )
	{
		// avir::round's branch for negative values is dropped: whatever it
		// returns for them is <= 0 and the clamp that follows makes it 0 --
		// as does the truncation of d + 0.5 (<= 0 for every d < 0)
		float t = v[ c ];

		if( O.use_tr )
		{
			t = (float) (int) ( t * O.tr_muli + 0.5f ) * O.tr_mul;
		}
		else
		{
			t = (float) (int) ( t + 0.5f );
		}

		t = fminf( fmaxf( t, 0.0f ), O.pk_out );
		q[ c ] = (unsigned int) t;
	}

	if( O.type == AVIRHIP_U8 )
	{
		unsigned char* const p = (unsigned char*) O.base + row * O.stride +
			(long) x * O.ch;

		if( O.ch == 4 )
		{
			*(unsigned int*) p = q[ 0 ] | ( q[ 1 ] << 8 ) | ( q[ 2 ] << 16 ) |
				( q[ 3 ] << 24 );
		}
		else
		{
			p[ 0 ] = (unsigned char) q[ 0 ];
			if( O.ch > 1 ) p[ 1 ] = (unsigned char) q[ 1 ];
			if( O.ch > 2 ) p[ 2 ] = (unsigned char) q[ 2 ];
		}
	}
	else
	{
		unsigned short* const p = (unsigned short*) O.base + row * O.stride +
			(long) x * O.ch;

		p[ 0 ] = (unsigned short) q[ 0 ];
		if( O.ch > 1 ) p[ 1 ] = (unsigned short) q[ 1 ];
		if( O.ch > 2 ) p[ 2 ] = (unsigned short) q[ 2 ];
		if( O.ch > 3 ) p[ 3 ] = (unsigned short) q[ 3 ];
	}
}

// LANCIR's output stage (outputScanline, lancir.h:1772-2056, as k_lancir_out /
// k_lancir_out_pad of generic.hip spell it out) fused into the horizontal
// pass' store: gain, then for integer elements the clamps and -- by the
// ELEMENT's position in its scanline of l = width * ch elements -- nearest-even
// rounding (whole groups of four) or + 0.5 truncation (the 1-3 element tail).
struct GPLOut
{
	int on;
	int type;        // AVIRHIP_U8 / AVIRHIP_U16 / AVIRHIP_F32
	int ch;          // channels of the caller's pixels (1..4)
	int unity;
	float out_mul, clampv;
	int l4;          // ( width * ch ) & ~3
	void* base;      // the caller's rows (row 0 = the band's first row)
	long stride;     // elements per row
};

__device__ __forceinline__ void gp_store_lancir( const GPLOut& O,
	const long row, const int x, const f4 o )
{
	const float vv[ 4 ] = { o.x, o.y, o.z, o.w };
	const long e0 = row * O.stride + (long) x * O.ch;
#pragma unroll
	for( int c = 0; c < 4; c++ )
	{
		if( c < O.ch )
		{
			float v = vv[ c ];

			if( !O.unity )
			{
				v = v * O.out_mul;
			}

			if( O.type == AVIRHIP_F32 )
			{
				((float*) O.base )[ e0 + c ] = v;
			}
			else
			{
				unsigned int q;

				if( x * O.ch + c < O.l4 )
				{
					v = ( v < O.clampv ? v : O.clampv );
					v = ( v > 0.0f ? v : 0.0f );
					q = (unsigned int) (int) rintf( v );
				}
				else
				{
					q = (unsigned int) (int) (( v > O.clampv ? O.clampv :
						( v < 0.0f ? 0.0f : v )) + 0.5f );
				}

				if( O.type == AVIRHIP_U8 )
				{
					((unsigned char*) O.base )[ e0 + c ] = (unsigned char) q;
				}
				else
				{
					((unsigned short*) O.base )[ e0 + c ] = (unsigned short) q;
				}
			}
		}
	}
}

// Host side of a recognised axis.
struct GPAxisHost
{
	bool ok;
	GPAxis a;       // device pointers filled in
	double kmax;    // largest source advance per output (ring sizing)
	std::vector< int > start; // host copy of the positions
	const float* h_coef; // host copy of the coefficient rows (a.coef's layout)
	const int* h_fidx;   // LANCIR: host copy of a.fidx
};

} // namespace avirhip

#endif
