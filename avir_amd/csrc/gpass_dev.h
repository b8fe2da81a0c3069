// gpass_dev.h -- what the pass kernels of gpass.hip and sacc.hip share: the
// lowered axis as the device sees it, the LDS-DMA and wait-count primitives.
#ifndef AVIRHIP_GPASS_DEV_H
#define AVIRHIP_GPASS_DEV_H

#include "plan.h"
#include <vector>
#include <algorithm>
#include <math.h>
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

// Wave priority inside a pass kernel's step (round 6, NOTEBOOK 10.11): low while
// a wave issues its DMA and waits for a row, one level up from the arithmetic
// on -- of the waves of a SIMD that are ready, the one that holds its row goes
// first. Small here (these kernels wait more than they issue): 1080p x3 on the
// fused kernel -2.0 %, 1080p -> 2500x1400 RGB uint8 -1.3 %, the others within
// +-0.5 % (profiles/r06_up2_prio/pass_kernels_ab.txt). -DGP_NO_PRIO: without.
#ifndef GP_NO_PRIO
#define GP_PRIO_LOW() __builtin_amdgcn_s_setprio( 0 )
#define GP_PRIO_HIGH() __builtin_amdgcn_s_setprio( 1 )
#else
#define GP_PRIO_LOW() do {} while( 0 )
#define GP_PRIO_HIGH() do {} while( 0 )
#endif

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
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"buffer_load_dwordx4 %1, %2, 0 offen lds"
		:: "s"( __builtin_amdgcn_readfirstlane( lds )), "v"( voff ), "s"( rs )
		: "memory" );
}

// ... of 64 x 4 bytes: lane l's dword lands at lds + l * 4 (raw integer rows,
// coefficient tables)
__device__ __forceinline__ void gp_dma4( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"buffer_load_dword %1, %2, 0 offen lds"
		:: "s"( __builtin_amdgcn_readfirstlane( lds )), "v"( voff ), "s"( rs )
		: "memory" );
}

// (waits.h: a counted wait in the product build, a full drain in `make paranoid`)
#define GP_WAIT_VM( n ) AVIRHIP_WAIT_VM( n )

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
		case 13: GP_WAIT_VM( 13 ); break;
		case 14: GP_WAIT_VM( 14 ); break;
		case 15: GP_WAIT_VM( 15 ); break;
		case 16: GP_WAIT_VM( 16 ); break;
		case 17: GP_WAIT_VM( 17 ); break;
		case 18: GP_WAIT_VM( 18 ); break;
		case 19: GP_WAIT_VM( 19 ); break;
		case 20: GP_WAIT_VM( 20 ); break;
		case 21: GP_WAIT_VM( 21 ); break;
		case 22: GP_WAIT_VM( 22 ); break;
		case 23: GP_WAIT_VM( 23 ); break;
		case 24: GP_WAIT_VM( 24 ); break;
		default: GP_WAIT_VM( 24 ); break;
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
	// an INTEGER source without bit-depth truncation: every result is finite
	// and within a few hundred of [0, pk_out], so the stage is add, convert,
	// integer clamp -- the same number as (float) (int) ( v + 0.5f ) clamped
	// in float and cast, for every |v| < 2^31 (three instructions a channel
	// instead of ten)
	int fin, pk_i;
	float tr_mul, tr_muli, pk_out;
	void* base;      // the caller's rows (row 0 = the band's first row)
	long stride;     // elements per row
};

// the plan's output stage, for rows stored from `iout` on
static inline void gp_make_out( const avirhip_plan* p, void* iout, GPOut& O )
{
	memset( &O, 0, sizeof( O ));
	O.on = 1; O.type = p -> out_type; O.ch = p -> io_ch;
	O.use_tr = ( p -> tr_mul != 1.0 );
	O.tr_mul = (float) p -> tr_mul;
	O.tr_muli = (float) ( 1.0 / p -> tr_mul );
	O.pk_out = (float) p -> pk_out;
	O.fin = ( !O.use_tr && !p -> gamma && ( p -> in_type == AVIRHIP_U8 ||
		p -> in_type == AVIRHIP_U16 ));
	O.pk_i = (int) p -> pk_out;
	O.base = iout;
	O.stride = (long) p -> new_w * p -> io_ch;
}

__device__ __forceinline__ float gp_round( const float d )
{
	return( d < 0.0f ? -(float) avirhip_x86_cvtt( 0.5f - d ) :
		(float) avirhip_x86_cvtt( d + 0.5f ));
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

	if( O.fin )
	{
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			q[ c ] = (unsigned int) min( max( (int) ( v[ c ] + 0.5f ), 0 ), O.pk_i );
		}
	}
	else
#pragma unroll
	for( int c = 0; c < 4; c++ )
	{
		// avir::round's branch for negative values is dropped: whatever it
		// returns for them is <= 0 and the clamp that follows makes it 0 --
		// as does the truncation of d + 0.5 (<= 0 for every d < 0)
		float t = v[ c ];
		float a = t;

		if( O.use_tr )
		{
			a = t * O.tr_muli;
			t = (float) (int) ( a + 0.5f ) * O.tr_mul;
		}
		else
		{
			t = (float) (int) ( t + 0.5f );
		}

		t = fminf( fmaxf( t, 0.0f ), O.pk_out );
		t = avirhip_x86_round_fix( a, t, O.pk_out );
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

// The same, called by EVERY lane of the wave (`ok`: this lane's pixel exists):
// lanes are consecutive pixels x of one row. RGB uint8 pixels -- three byte
// stores per lane in gp_store_int, three store instructions per output row and
// 64 x 3 scattered bytes each -- leave as whole dwords instead: a quad of lanes
// holds 12 contiguous bytes, lane q of it builds dword q from its own packed
// pixel and its right neighbour's (one DPP move) and lanes 0-2 store: one store
// instruction per row, 192 contiguous bytes per wave. Quads cut by the row's
// end keep the byte stores. (1920x1080 -> 2500x1400 RGB uint8: the vertical
// pass with the integer output stage took 33.0 us against 25.5 for floats.)
__device__ __forceinline__ void gp_store_int_row( const GPOut& O, const long row,
	const int x, const f4 o, const bool ok )
{
	if( !( O.type == AVIRHIP_U8 && O.ch == 3 ))
	{
		if( ok )
		{
			gp_store_int( O, row, x, o );
		}

		return;
	}

	const float v[ 3 ] = { o.x, o.y, o.z };
	unsigned int pk = 0;

	if( O.fin )
	{
#pragma unroll
		for( int c = 0; c < 3; c++ )
		{
			pk |= (unsigned int) min( max( (int) ( v[ c ] + 0.5f ), 0 ), 255 ) <<
				( 8 * c );
		}
	}
	else
#pragma unroll
	for( int c = 0; c < 3; c++ )
	{
		float t = v[ c ];
		float a = t;

		if( O.use_tr )
		{
			a = t * O.tr_muli;
			t = (float) (int) ( a + 0.5f ) * O.tr_mul;
		}
		else
		{
			t = (float) (int) ( t + 0.5f );
		}

		t = fminf( fmaxf( t, 0.0f ), O.pk_out );
		t = avirhip_x86_round_fix( a, t, O.pk_out );
		pk |= ( (unsigned int) t & 0xffu ) << ( 8 * c );
	}

	const int lane = (int) __builtin_amdgcn_mbcnt_hi( ~0u,
		__builtin_amdgcn_mbcnt_lo( ~0u, 0u ));
	const int q = lane & 3;
	// the right neighbour's packed pixel (quad_perm [1, 2, 3, 3])
	const unsigned int nx = (unsigned int) __builtin_amdgcn_mov_dpp( (int) pk,
		0xF9, 0xF, 0xF, true );
	const unsigned long long m = __builtin_amdgcn_ballot_w64( ok );
	const bool full = ((( m >> ( lane & ~3 )) & 0xFull ) == 0xFull );
	unsigned char* const p = (unsigned char*) O.base + row * O.stride +
		(long) x * 3;

	if( full )
	{
		if( q < 3 )
		{
			// (dword q of the quad's 12 bytes: at the quad's first byte + 4 q
			// = this lane's pixel + q bytes)
			typedef unsigned int u32u __attribute__(( aligned( 1 )));
			*(u32u*) ( p + q ) = ( pk >> ( 8 * q )) | ( nx << ( 24 - 8 * q ));
		}
	}
	else
	if( ok )
	{
		p[ 0 ] = (unsigned char) pk;
		p[ 1 ] = (unsigned char) ( pk >> 8 );
		p[ 2 ] = (unsigned char) ( pk >> 16 );
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

// The same, called by EVERY lane of the wave (`ok`: this lane's pixel exists;
// lanes are consecutive pixels x of one row). uint8 pixels of 3 or 4 channels
// leave as whole dwords -- RGBA: the lane's own four bytes; RGB: a quad of lanes
// holds 12 contiguous bytes, lane q builds dword q from its packed pixel and its
// right neighbour's (one DPP move), lanes 0-2 store -- one store instruction per
// row instead of three or four with 64 scattered bytes each. uint16 RGBA: one
// 8-byte store. The rounding rule is a property of the ELEMENT's position, and
// the 1-3 element tail of a scanline lies in its last pixel: a wave none of
// whose pixels reaches it (one ballot) rounds to nearest-even in straight-line
// code; the first form branched per element and channel -- 25 branches per
// output row, more cycles than the two dot products before them.
// KIND: 0 = whatever O says; 1 / 2 = uint8 RGB / RGBA known at compile time.
template< int KIND >
__device__ __forceinline__ void gp_store_lancir_row( const GPLOut& O,
	const long row, const int x, const f4 o, const bool ok )
{
	const int type = ( KIND != 0 ? AVIRHIP_U8 : O.type );
	const int ch = ( KIND == 1 ? 3 : ( KIND == 2 ? 4 : O.ch ));
	const bool b8 = ( type == AVIRHIP_U8 && ch >= 3 );
	const bool b16 = ( type == AVIRHIP_U16 && ch == 4 );

	if( !b8 && !b16 )
	{
		if( ok )
		{
			gp_store_lancir( O, row, x, o );
		}

		return;
	}

	const float vv[ 4 ] = { o.x, o.y, o.z, o.w };
	// (x * 1.0f is x: no branch on O.unity)
	const float mulv = ( O.unity ? 1.0f : O.out_mul );
	unsigned int q[ 4 ] = { 0, 0, 0, 0 };
	const bool tail = ( __builtin_amdgcn_ballot_w64( ok &&
		x * ch + ch > O.l4 ) != 0 );

	if( !tail )
	{
#pragma unroll
		for( int c = 0; c < ( KIND == 1 ? 3 : 4 ); c++ )
		{
			float v = vv[ c ] * mulv;
			v = ( v < O.clampv ? v : O.clampv );
			v = ( v > 0.0f ? v : 0.0f );
			q[ c ] = (unsigned int) (int) rintf( v );
		}
	}
	else
	{
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			float v = vv[ c ] * mulv;

			if( x * ch + c < O.l4 )
			{
				v = ( v < O.clampv ? v : O.clampv );
				v = ( v > 0.0f ? v : 0.0f );
				q[ c ] = (unsigned int) (int) rintf( v );
			}
			else
			{
				q[ c ] = (unsigned int) (int) (( v > O.clampv ? O.clampv :
					( v < 0.0f ? 0.0f : v )) + 0.5f );
			}
		}
	}

	typedef unsigned int u32u __attribute__(( aligned( 1 )));

	if( b16 )
	{
		if( ok )
		{
			unsigned short* const p = (unsigned short*) O.base +
				row * O.stride + (long) x * 4;
			typedef unsigned int u32h __attribute__(( aligned( 2 )));
			((u32h*) p )[ 0 ] = ( q[ 0 ] & 0xffffu ) | ( q[ 1 ] << 16 );
			((u32h*) p )[ 1 ] = ( q[ 2 ] & 0xffffu ) | ( q[ 3 ] << 16 );
		}

		return;
	}

	const unsigned int pk = ( q[ 0 ] & 0xffu ) | (( q[ 1 ] & 0xffu ) << 8 ) |
		(( q[ 2 ] & 0xffu ) << 16 ) | ( q[ 3 ] << 24 );

	if( ch == 4 )
	{
		if( ok )
		{
			*(u32u*) ((unsigned char*) O.base + row * O.stride +
				(long) x * 4 ) = pk;
		}

		return;
	}

	const int lane = (int) __builtin_amdgcn_mbcnt_hi( ~0u,
		__builtin_amdgcn_mbcnt_lo( ~0u, 0u ));
	const int qd = lane & 3;
	// the right neighbour's packed pixel (quad_perm [1, 2, 3, 3])
	const unsigned int nx = (unsigned int) __builtin_amdgcn_mov_dpp( (int) pk,
		0xF9, 0xF, 0xF, true );
	const unsigned long long m = __builtin_amdgcn_ballot_w64( ok );
	const bool full = ((( m >> ( lane & ~3 )) & 0xFull ) == 0xFull );
	unsigned char* const p = (unsigned char*) O.base + row * O.stride +
		(long) x * 3;

	// (dword qd of the quad's 12 bytes: at the quad's first byte + 4 qd = this
	// lane's pixel + qd bytes)
	const unsigned int dw = (( pk & 0xffffffu ) >> ( 8 * qd )) |
		( nx << ( 24 - 8 * qd ));

	if( m == ~0ull )
	{
		if( qd < 3 )
		{
			*(u32u*) ( p + qd ) = dw;
		}

		return;
	}

	if( full )
	{
		if( qd < 3 )
		{
			*(u32u*) ( p + qd ) = dw;
		}
	}
	else
	if( ok )
	{
		p[ 0 ] = (unsigned char) pk;
		p[ 1 ] = (unsigned char) ( pk >> 8 );
		p[ 2 ] = (unsigned char) ( pk >> 16 );
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

// ---- shared by gpass.hip (horizontal pass, host side) and gpassv.hip
// (vertical pass): launch parameters, the gather's dot product, the raw-pixel
// loader, the work-item sizing

struct GVParams
{
	const float* src; long src_ss; int width; // px per row, floats per row
	float* dst; long dst_ss; int dst_row0;
	GPAxis ax;
	int row_lo, row_hi;  // output rows to produce
	int dbg;             // timing ablations (GP_DBG builds only)
	int nstrips, chunk;  // 32-px strips, output rows per work item
	int blk;             // output rows per block
	int rs, rc;          // ring sizes (rows, powers of two): source, C / G
	GPOut out;           // on != 0 (whole-pixel lanes): integer output stage
	const void* raw;     // != nullptr (whole-pixel lanes): the caller's image
	long raw_ss;         // (integer / narrower pixels), elements per row,
	int raw_ch, raw_kind; // channels, 1 uint8 / 2 uint16 / 3 float
	long raw_elems;      // elements of that image (the loader's bound)
	int raw_dma;         // the rows travel as bytes by LDS-DMA (dword-aligned
	                     // base and row pitch, < 2 GiB): bytes per pixel, else 0
	int raw_bytes;       // bytes of that image (the DMA's range check)
};

struct GHParams
{
	const float* src; long src_ss; int src_w;
	float* dst; long dst_ss; int dst_row0; int dst_w;
	GPAxis ax;
	int row_lo, row_hi;  // rows (of both images) to process
	int nstrips, ow, chunk; // outputs per strip, rows per work item
	const int* seg;      // [nstrips][4]: source lo, count; mid lo, count
	int nseg;            // DMA instructions per row (uniform)
	int sbuf, mbuf;      // LDS pixels per source row buffer / mid buffer
	const void* raw;     // SRC != 0: the caller's image (integer / narrower
	long raw_ss;         // pixels), elements per row, channels
	int raw_ch;
	long raw_elems;      // elements of that image (the loader's bound)
	GPLOut lout;         // on != 0: LANCIR's output stage in the store
	int dbg;             // timing ablations (GP_DBG builds only): 1 no stores
};

// gfuse.hip: both passes of an upsizing plan in one launch (k_gf)
struct GFParams
{
	const float* src; long src_ss;  // float RGBA source, floats per row
	float* dst; long dst_ss; int dst_row0;
	GPAxis hx, vx;       // both GP_PRE, 12 taps
	int row_lo, row_hi;  // output rows to produce
	int nstrips, ow, chunk; // k_gh's strips; output rows per work item
	const int* seg;      // [nstrips][4]: source lo, count; FIR output lo, count
	int nseg;            // DMA instructions per source row
	int sbuf, mbuf;      // LDS pixels per source row buffer / FIR row
	GPOut out;           // on != 0: integer output stage in the store
};

int gfuse_launch( GFParams& P, double k_v, hipStream_t st );

// lfuse.hip: both passes of a LANCIR upsizing plan in one launch (k_lf)
struct LFParams
{
	const float* src; long src_ss;  // float RGBA source, floats per row
	float* dst; long dst_ss; int dst_row0;
	GPAxis hx, vx;       // both LANCIR (the same summation order), 6 taps
	int row_lo, row_hi;  // output rows to produce
	int nstrips, ow, chunk; // strips of `ow` outputs (segment <= 64 source px)
	const int* seg;      // [nstrips][2]: first source column, count
	const float* vtab;   // [vx.g_len][8]: the vertical coefficient row of every
	                     // output row (its phase's, zero padded)
	GPLOut lout;         // on != 0: the owner's output stage in the store
	int dbg;
	// raw != nullptr: the owner's image as it is (integer / narrower pixels)
	// -- rows travel as bytes, lfuse.hip
	const void* raw; long raw_ss;   // elements per row
	int raw_kind, raw_ch;           // 1 uint8, 2 uint16, 3 float; 1..4 channels
	int raw_bpp, raw_bytes, raw_tdn; // bytes per pixel, of the image; DMA
	                                // instructions per row segment
};

int lfuse_launch( LFParams& P, hipStream_t st );
// gpass_h2.hip: k_gh2 (two gather outputs per lane); 1: no variant for the tap count
int launch_gh2( const GHParams& P, int items, size_t lds, hipStream_t st );
bool lfuse_takes_raw( const void* raw, int type, int ch, long stride,
	int in_len_v, int width );

// One gather output: tap t reads ld( t ); cf4( b ) returns the coefficients
// of taps 4b .. 4b + 3 (rows are padded to a multiple of four; padded taps are
// never multiplied: x * 0 is not a no-op for non-finite x). NTC > 0: the tap
// count is a compile-time constant and everything unrolls.
//
// LANCIR's summation order depends on the image's channel count (the lane
// structure of resize1..4, lancir.h:2102-2544; generic.hip spells it out):
// `lord` = 4: even and odd taps; 1, 2, 3: four sums over taps 4g + j, folded
// (s0 + s2) + (s1 + s3) for 1 and 2 channels, (s0 + s1) + (s2 + s3) for 3, with
// the two tail taps of a length 4n + 2 entering where that code adds them --
// for RGB differently in channel 0 (`ch0`: this lane's .x is channel 0).
template< bool LANC, typename T, int NTC, typename LD, typename CF >
__device__ __forceinline__ T gp_dot( const int nt_, LD ld, CF cf4,
	const int lord = 4, const bool ch0 = true )
{
	const int nt = ( NTC > 0 ? NTC : nt_ );
	const int nb = nt >> 2;

	if( LANC && lord != 4 )
	{
		T a0, a1, a2, a3;
		{
			const f4 c = cf4( 0 );
			a0 = c.x * ld( 0 );
			a1 = c.y * ld( 1 );
			a2 = c.z * ld( 2 );
			a3 = c.w * ld( 3 );
		}

		if( NTC > 0 )
		{
#pragma unroll
			for( int b = 1; b < ( NTC > 0 ? NTC / 4 : 1 ); b++ )
			{
				const f4 c = cf4( b );
				a0 = a0 + c.x * ld( 4 * b );
				a1 = a1 + c.y * ld( 4 * b + 1 );
				a2 = a2 + c.z * ld( 4 * b + 2 );
				a3 = a3 + c.w * ld( 4 * b + 3 );
			}
		}
		else
		{
			for( int b = 1; b < nb; b++ )
			{
				const f4 c = cf4( b );
				a0 = a0 + c.x * ld( 4 * b );
				a1 = a1 + c.y * ld( 4 * b + 1 );
				a2 = a2 + c.z * ld( 4 * b + 2 );
				a3 = a3 + c.w * ld( 4 * b + 3 );
			}
		}

		if( nt & 2 )
		{
			const f4 c = cf4( nb );
			const T t0 = c.x * ld( 4 * nb );
			const T t1 = c.y * ld( 4 * nb + 1 );

			if( lord == 3 )
			{
				a0 = a0 + t0;
				T r = (( a0 + a1 ) + ( a2 + a3 )) + t1;     // channels 1, 2
				const T r0 = ( a0 + ( a1 + t1 )) + ( a2 + a3 ); // channel 0

				if( ch0 )
				{
					r.x = r0.x;
				}

				return( r );
			}

			return((( a0 + a2 ) + t0 ) + (( a1 + a3 ) + t1 ));
		}

		if( lord == 3 )
		{
			return(( a0 + a1 ) + ( a2 + a3 ));
		}

		return(( a0 + a2 ) + ( a1 + a3 ));
	}

	if( LANC )
	{
		// (t0*p0 + t2*p2 + ...) + (t1*p1 + t3*p3 + ...), bare first products
		// (lancir.h:2482-2515); nt is even
		T ea, oa;
		int b = 0;

		if( nt >= 4 )
		{
			const f4 c = cf4( 0 );
			ea = c.x * ld( 0 );
			oa = c.y * ld( 1 );
			ea = ea + c.z * ld( 2 );
			oa = oa + c.w * ld( 3 );
			b = 1;
		}
		else
		{
			const f4 c = cf4( 0 );
			ea = c.x * ld( 0 );
			oa = c.y * ld( 1 );
			return( ea + oa );
		}
		if( NTC > 0 )
		{
#pragma unroll
			for( int bb = 1; bb < ( NTC > 0 ? NTC / 4 : 1 ); bb++ )
			{
				const f4 c = cf4( bb );
				ea = ea + c.x * ld( 4 * bb );
				oa = oa + c.y * ld( 4 * bb + 1 );
				ea = ea + c.z * ld( 4 * bb + 2 );
				oa = oa + c.w * ld( 4 * bb + 3 );
			}
		}
		else
		{
#pragma unroll 2
			for( ; b < nb; b++ )
			{
				const f4 c = cf4( b );
				ea = ea + c.x * ld( 4 * b );
				oa = oa + c.y * ld( 4 * b + 1 );
				ea = ea + c.z * ld( 4 * b + 2 );
				oa = oa + c.w * ld( 4 * b + 3 );
			}
		}

		if( nt & 2 )
		{
			const f4 c = cf4( nb );
			ea = ea + c.x * ld( 4 * nb );
			oa = oa + c.y * ld( 4 * nb + 1 );
		}

		return( ea + oa );
	}

	T acc = (T) 0.0f;

	if( NTC < 0 )
	{
		// any count up to 28: the caller keeps the coefficient row in
		// registers (cf4( b ), b compile-time); whole blocks of four taps and
		// the last partial block behind wave-uniform branches
#pragma unroll
		for( int b = 0; b < 7; b++ )
		{
			if( 4 * b + 4 <= nt )
			{
				const f4 c = cf4( b );
				acc = acc + c.x * ld( 4 * b );
				acc = acc + c.y * ld( 4 * b + 1 );
				acc = acc + c.z * ld( 4 * b + 2 );
				acc = acc + c.w * ld( 4 * b + 3 );
			}
			else
			if( 4 * b < nt )
			{
				const f4 c = cf4( b );
				acc = acc + c.x * ld( 4 * b );

				if(( nt & 3 ) > 1 )
				{
					acc = acc + c.y * ld( 4 * b + 1 );
				}

				if(( nt & 3 ) > 2 )
				{
					acc = acc + c.z * ld( 4 * b + 2 );
				}
			}
		}

		return( acc );
	}

	if( NTC > 0 )
	{
#pragma unroll
		for( int b = 0; b < ( NTC > 0 ? NTC / 4 : 1 ); b++ )
		{
			const f4 c = cf4( b );
			acc = acc + c.x * ld( 4 * b );
			acc = acc + c.y * ld( 4 * b + 1 );
			acc = acc + c.z * ld( 4 * b + 2 );
			acc = acc + c.w * ld( 4 * b + 3 );
		}
	}
	else
	{
#pragma unroll 2
		for( int b = 0; b < nb; b++ )
		{
			const f4 c = cf4( b );
			acc = acc + c.x * ld( 4 * b );
			acc = acc + c.y * ld( 4 * b + 1 );
			acc = acc + c.z * ld( 4 * b + 2 );
			acc = acc + c.w * ld( 4 * b + 3 );
		}
	}

	if( nt & 3 )
	{
		const f4 c = cf4( nb );
		acc = acc + c.x * ld( 4 * nb );

		if(( nt & 3 ) > 1 )
		{
			acc = acc + c.y * ld( 4 * nb + 1 );
		}

		if(( nt & 3 ) > 2 )
		{
			acc = acc + c.z * ld( 4 * nb + 2 );
		}
	}

	return( acc );
}

// One raw pixel (1 uint8, 2 uint16, 3 float elements; `ch` of them) as float
// RGBA with zero padding: ONE load of four elements at the pixel's (possibly
// unaligned) address -- what it reads past the pixel is dropped; only where
// those four elements would cross the end of the image (its last pixel) are
// the channels loaded one by one. Per-element loads everywhere cost three or
// four vector-memory instructions per pixel and as much time as the pack pass
// they replace. (Buffer loads force dword alignment: global loads it is.)
__device__ __forceinline__ f4 gp_load_raw( const void* const base,
	const int kind, const int ch, const long eoff, const long total )
{
	float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
	const bool fast = ( eoff + 4 <= total );
	// integer elements, branch-free: the load starts at most at total - 4 and
	// the value is shifted down by the elements it started early (non-zero
	// for the image's last pixel only; the image holds >= 4 elements)
	const long ea = min( eoff, total - 4 );
	const int sh = (int) ( eoff - ea );

	if( kind == 1 )
	{
		typedef unsigned int u32u __attribute__(( aligned( 1 )));
		const unsigned int w = *(const u32u*) ( (const unsigned char*) base +
			ea ) >> ( 8 * sh );
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			v[ c ] = ( c < ch ? (float) (( w >> ( 8 * c )) & 0xffu ) : 0.0f );
		}
	}
	else
	if( kind == 2 )
	{
		typedef unsigned long long u64u __attribute__(( aligned( 2 )));
		const unsigned long long w = *(const u64u*) (
			(const unsigned short*) base + ea ) >> ( 16 * sh );
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			v[ c ] = ( c < ch ? (float) (unsigned int) (( w >> ( 16 * c )) &
				0xffffull ) : 0.0f );
		}
	}
	else
	{
		const float* const p = (const float*) base + eoff;

		if( fast )
		{
			typedef float f4u __attribute__(( ext_vector_type( 4 ),
				aligned( 4 )));
			const f4u w = *(const f4u*) p;
			const float ww[ 4 ] = { w.x, w.y, w.z, w.w };
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				v[ c ] = ( c < ch ? ww[ c ] : 0.0f );
			}
		}
		else
		{
			for( int c = 0; c < ch; c++ )
			{
				v[ c ] = p[ c ];
			}
		}
	}

	f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
	return( o );
}

#ifndef GV_PF
#define GV_PF 5 // k_gv: DMA groups in flight (one of them is read a step ahead)
#endif
#define GV_QB (( GV_PF + 2 ) * 1024 ) // landing queue: GV_PF + 1 slots, a dump

// Rows per work item. The vertical pass kernel is bound by vector-ALU issue,
// not by memory (measured: removing the DMA, its waits or the stores changes a
// launch by < 10 %): a launch takes as long as its busiest SIMD, that is
// ceil( waves / 1024 SIMDs ) waves of ( chunk + warm ) rows each, where `warm`
// = the rows of work a chunk spends before its first output. A single wave
// per SIMD has nobody to hide its latencies behind (x 1.6: a lone wave issues
// its dependent multiply-adds at half the rate of two, tools/pk_bench.hip), and
// more waves than the chip holds (`wpc` per CU) run in residency rounds.
// Fits the k_gv sweep profiles/r02_gpass/chunk_sweep.txt within 10 %.
static inline int balanced_chunk( int rows, int nstrips, int min_chunk, int max_chunk,
	int warm, int wpc, bool valu_bound )
{
	const double cap = 256.0 * wpc;
	double best = -1.0;
	int best_chunk = std::max( 1, std::min( rows, max_chunk ));
	static const double solo = ( getenv( "AVIRHIP_GP_SOLO" ) != nullptr ?
		atof( getenv( "AVIRHIP_GP_SOLO" )) : 1.6 ); // tuning aids
	static const int minc = ( getenv( "AVIRHIP_GP_MINCHUNK" ) != nullptr ?
		atoi( getenv( "AVIRHIP_GP_MINCHUNK" )) : 0 );

	if( minc > 0 )
	{
		min_chunk = minc;
	}

	min_chunk = std::max( 1, std::min( min_chunk, best_chunk ));

	for( int nch = 1; nch <= rows; nch++ )
	{
		const int chunk = ( rows + nch - 1 ) / nch;

		if( chunk > max_chunk )
		{
			continue;
		}

		if( chunk < min_chunk )
		{
			break;
		}

		const int n = ( rows + chunk - 1 ) / chunk;
		const double items = (double) n * nstrips;
		const double res = ceil( items / cap ); // residency rounds
		const double per = ceil( items / res / 1024.0 ); // waves per SIMD
		// (k_gh is not ALU-bound: there whole rounds of resident waves is
		// the model that fits -- 4661 waves on 4608 slots ran as long as 9216)
		const double cost = ( valu_bound ?
			res * per * ( chunk + (double) warm ) * ( per < 2.0 ? solo : 1.0 ) :
			res * cap / items * ( chunk + (double) warm ) / chunk );

		if( best < 0.0 || cost < best )
		{
			best = cost;
			best_chunk = chunk;
		}
	}

	return( best_chunk );
}

// gpassv.hip
int gpass_run_v( const GPAxis& A, int v_blk, int v_rs, int v_rc,
	const float* src, long src_ss, int width, float* dst, long dst_ss,
	int dst_row0, int row_lo, int row_hi, hipStream_t st,
	const GPOut* out = nullptr, const void* raw = nullptr, int raw_type = 0,
	int raw_ch = 0, long raw_stride = 0 );

} // namespace avirhip

#endif
