// lfuse.hip -- k_lf: BOTH passes of a general-ratio LANCIR upsizing plan in one
// launch (CLancIR::resizeImage, lancir.h:386-713: the vertical pass first --
// copyScanline*v + resize* down the columns, lancir.h:601-646 -- then the
// horizontal one -- padScanline*h + resize* along the rows, lancir.h:648-710).
//
// LANCIR runs its passes the other way round than AVIR, and that makes its
// fusion free of recomputation: the vertical window of an output row holds RAW
// source rows (a chunk's run-in is six row loads, no arithmetic), and the
// horizontal taps of an output pixel read the vertically resized row of that
// very output row -- nothing a neighbouring chunk or strip computes is needed
// twice. (AVIR's fused kernel, gfuse.hip, repeats the horizontal arithmetic of
// its 18-row vertical run-in and only pays from 12 Mpixels on.)
//
//   one wavefront per (strip of `ow` output columns whose source segment is at
//   most 64 pixels) x (chunk of output rows), marching down the source rows:
//     a source row's segment arrives by LDS-DMA (three rows ahead); each lane
//     reads ITS source column's pixel into the 6-row window it keeps in
//     registers (compile-time slots: the step loop is unrolled 8 times);
//     for every output row whose last source row this was:
//       V  lane = source column: the 6-tap Lanczos dot product down its window,
//          x86 even / odd summation order (lancir.h:2482-2515), wave-uniform
//          coefficient row of the row's fractional phase -> LDS;
//       H  lane = output column: its 6 taps from LDS, its coefficient row in
//          registers -> the pixel, stored (or through the owner's output stage:
//          gain, clamps, integer rounding -- gp_store_lancir).
//
// The intermediate image [new_h][src_w] never exists, one launch instead of two.
// Same expressions in the same order as k_gv< NONE, LANC > + k_gh< NONE, LANC >
// => the same bits.

#include "gpass_dev.h"
#include <algorithm>
#include <type_traits>

namespace avirhip {

#define LF_NB 8   // source row buffers: rows u .. u + 7 are in LDS or in flight
#define LF_TE 24  // 6 KiB of vertical coefficient rows per chunk at most

// timing ablations (make gpdbg; AVIRHIP_LF_DBG bits: 1 no stores, 2 no
// horizontal stage, 4 no vertical dot product, 8 no DMA, 16 no row loop at all)
#ifdef GP_DBG
#define LF_ON( bit ) ( !( P.dbg & ( bit )))
#else
#define LF_ON( bit ) true
#endif

// TDN = 0: float RGBA rows, a pixel per lane and DMA instruction. TDN > 0: raw
// rows (uint8 / uint16 / 1-3 channel float pixels) as BYTES -- a row's segment
// is TDN dword instructions of 256 bytes from the dword at or below its first
// pixel, and a lane converts its pixel when it reads the row buffer
// (copyScanline*v's cast and zero padding, lancir.h:1406-1594); the host only
// picks these for dword-aligned bases and pitches (lfuse_takes_raw).
// SRC = 0: float RGBA; 1: uint8 pixels (always one DMA instruction); 2 .. 4: any
// raw kind with SRC - 1 instructions. OUTK: the output stage's compile-time
// kind (gp_store_lancir_row; 0 = whatever P.lout says).
// LORD: the summation order at compile time (3, 4), or 0 = the axis' own.
template< int SRC, int OUTK, int LORD >
__global__ void __launch_bounds__( 64 ) k_lf( const LFParams P )
{
	constexpr bool RAW = ( SRC > 0 );
	constexpr int TDN = ( SRC <= 1 ? SRC : SRC - 1 );
	const int raw_kind = ( SRC == 1 ? 1 : P.raw_kind );
	constexpr int NT = 6;   // taps of both kernels (la = 3, upsizing)
	constexpr int NTP = 8;  // ... padded to whole f4 rows
	constexpr int RW = 8;   // vertical window (ring slots, >= NT)
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& HA = P.hx;
	const GPAxis& VA = P.vx;
	f4* const sbuf = (f4*) smem;           // [LF_NB][64 px]
	f4* const mbuf = sbuf + LF_NB * 64;    // [64 px]: the vertically resized row
	float* const ctab = (float*) ( mbuf + 64 ); // [chunk rows][NTP]
	const unsigned sbuf_lds = (unsigned) (unsigned long) (lds_char*) smem;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int y0 = P.row_lo + chunk * P.chunk;
	const int y1 = min( y0 + P.chunk, P.row_hi );

	if( y0 >= y1 )
	{
		return;
	}

	// ---- horizontal side: this lane's output column
	const int slo = P.seg[ strip * 2 ]; // first source column of the segment
	const int j = strip * P.ow + lane;
	const bool jok = ( lane < P.ow && j < HA.out_len );
	const int gh = gp_clamp( j, HA.g_len - 1 );
	const int gs = HA.start[ gh ] - slo;
	const int hslast = HA.in_len - 1;
	float cf[ NTP ];
	{
		const float* const cr = HA.coef + (long) HA.fidx[ gh ] * NT;
#pragma unroll
		for( int t = 0; t < NTP; t++ )
		{
			cf[ t ] = ( t < NT ? cr[ t < NT ? t : 0 ] : 0.0f );
		}
	}

	// ---- vertical side: this lane's source column (clamped: the padded view)
	gp_cint* const a_start = (gp_cint*) (unsigned long) VA.start;
	const int vslast = VA.in_len - 1;
	// the image's channel count picks the order
	const int lord = ( LORD != 0 ? LORD : VA.lanc );
	const int ga = y0, gb = y1 - 1;
	const int u_a = a_start[ ga ];
	const int u_b = a_start[ gb ] + NT - 1;

	// ---- the feed: source row clamp( u ) of step u, 64 pixels from slo on
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) ( RAW ? P.raw : (const void*) P.src ), 0,
		( RAW ? P.raw_bytes : 0x7fffffff ), 0x00020000 );
	const int srow_b = ( RAW ? (int) P.raw_ss * ( raw_kind == 1 ? 1 :
		( raw_kind == 2 ? 2 : 4 )) : (int) P.src_ss * 4 );
	// (raw: the segment's first byte, its dword, this lane's pixel behind it;
	// bytes past the image's end read as zeros -- the buffer's range check)
	const int sa = gp_clamp( slo, hslast );
	const int b0 = sa * P.raw_bpp;
	const int poff = ( b0 & 3 ) + ( gp_clamp( slo + lane, hslast ) - sa ) *
		P.raw_bpp;
	const int xoff = ( RAW ? ( b0 & ~3 ) + lane * 4 :
		gp_clamp( slo + lane, hslast ) * 16 );

	auto issue_row = [&]( const int u, const int buf )
	{
		const int boff = gp_clamp( u, vslast ) * srow_b + xoff;

		if( RAW )
		{
#pragma unroll
			for( int k = 0; k < ( RAW ? TDN : 1 ); k++ )
			{
				gp_dma4( rsrc, sbuf_lds + ( buf * 64 ) * 16 + k * 256,
					boff + k * 256 );
			}
		}
		else
		{
			gp_dma( rsrc, sbuf_lds + ( buf * 64 ) * 16, boff );
		}
	};

	// this lane's pixel of a row buffer: four elements whatever the channel
	// count, the padding selected to zero (no branch per channel)
	auto row_px = [&]( const int buf ) -> f4
	{
		if( !RAW )
		{
			return( sbuf[ buf * 64 + lane ]);
		}

		const char* const rp = smem + buf * 1024 + poff;
		float v[ 4 ];

		if( raw_kind == 1 )
		{
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				v[ c ] = (float) ((const unsigned char*) rp )[ c ];
			}
		}
		else
		if( raw_kind == 2 )
		{
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				v[ c ] = (float) ((const unsigned short*) rp )[ c ];
			}
		}
		else
		{
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				v[ c ] = ((const float*) rp )[ c ];
			}
		}
#pragma unroll
		for( int c = 1; c < 4; c++ )
		{
			v[ c ] = ( c < P.raw_ch ? v[ c ] : 0.0f );
		}

		f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
		return( o );
	};

	for( int k = 0; k < LF_NB - 1 && u_a + k <= u_b; k++ )
	{
		issue_row( u_a + k, k );
	}

	// (the first rows are on their way: the tables are fetched under them)
	// This chunk's coefficient rows -> LDS. P.vtab is the vertical axis' table
	// expanded per output row on the host ([g_len][NTP], zero padded): one
	// contiguous run, fetched four independent loads at a time with clamped
	// (always valid) addresses. Through the per-phase table and its index the
	// same staging compiled to 24 dependent load pairs, each behind its own
	// branch and vmcnt(0) -- half of a 24-us launch.
	int* const stab = (int*) ( ctab + ( gb - ga + 1 ) * NTP );
	{
		const int total = ( gb - ga + 1 ) * NTP;
		const float* const vt = P.vtab + (long) ga * NTP;

		for( int k0 = 0; k0 * 64 < total; k0 += 4 )
		{
			float t4[ 4 ];
#pragma unroll
			for( int q = 0; q < 4; q++ )
			{
				t4[ q ] = vt[ min(( k0 + q ) * 64 + lane, total - 1 )];
			}
#pragma unroll
			for( int q = 0; q < 4; q++ )
			{
				const int i = ( k0 + q ) * 64 + lane;

				if( i < total )
				{
					ctab[ i ] = t4[ q ];
				}
			}
		}

		for( int i = lane; i <= gb - ga; i += 64 )
		{
			stab[ i ] = VA.start[ ga + i ] + NT - 1;
		}
	}


	int g = ga;                          // next output row
	int g_need = a_start[ g ] + NT - 1;  // its last source row
	float* dp = P.dst + (long) ( y0 - P.dst_row0 ) * P.dst_ss + (long) j * 4;
	long orow = y0 - P.dst_row0;

	f4 cpre[ 2 ];
	int need1_v;

	auto prefetch_tabs = [&]()
	{
		const int gi = min( g + 1, gb ) - ga;
		need1_v = ( g + 1 <= gb ? stab[ gi ] : 0x7fffffff );
		const float* const cr = ctab + ( min( g, gb ) - ga ) * NTP;
		cpre[ 0 ] = *(const f4*) cr;
		cpre[ 1 ] = *(const f4*) ( cr + 4 );
	};

	f4 win[ RW ];
#pragma unroll
	for( int i = 0; i < RW; i++ )
	{
		win[ i ] = (f4) 0.0f;
	}

	auto drain = [&]( const int u, auto IC )
	{
		constexpr int I = decltype( IC )::value;

		// (after the chunk's last output g_need is prefetch_tabs' 0x7fffffff)
		while( g_need <= u )
		{
			// V: this lane's source column, the last NT rows
			f4 c = win[ I & ( RW - 1 )];

			if( LF_ON( 4 ))
			c = gp_dot< true, f4, NT >( NT,
				[&]( const int t ) { return( win[ ( I - ( NT - 1 ) + t + 4 * RW ) &
					( RW - 1 )]); },
				[&]( const int b ) { return( cpre[ b ]); }, lord );

			if( LF_ON( 2 ))
			mbuf[ lane ] = c;

			// H: this lane's output column
			f4 o = c;

			if( jok && LF_ON( 2 ))
			{
				const f4* const in = mbuf + gs;
				f4 w[ NT ];
#pragma unroll
				for( int t = 0; t < NT; t++ )
				{
					w[ t ] = in[ t ];
				}

				o = gp_dot< true, f4, NT >( NT,
					[&]( const int t ) { return( w[ t % NT ]); },
					[&]( const int b ) -> f4
					{
						f4 q;
						q.x = cf[ ( 4 * b ) % NTP ]; q.y = cf[ ( 4 * b + 1 ) % NTP ];
						q.z = cf[ ( 4 * b + 2 ) % NTP ]; q.w = cf[ ( 4 * b + 3 ) % NTP ];
						return( q );
					}, lord );
			}

			if( P.lout.on != 0 )
			{
				gp_store_lancir_row< OUTK >( P.lout, orow, j, o, jok );

				orow++;
			}
			else
			{
				if( jok && ( LF_ON( 1 ) || o.x == 1234.5f ))
				{
					*(f4*) dp = o;
				}

				dp += P.dst_ss;
			}

			g++;
			g_need = __builtin_amdgcn_readfirstlane( need1_v );
			prefetch_tabs();
		}
	};

	prefetch_tabs();

	auto step = [&]( const int u, auto IC )
	{
		constexpr int i = decltype( IC )::value;

		if( u > u_b )
		{
			return;
		}

		const int buf = i & ( LF_NB - 1 );

		if( u + LF_NB - 1 <= u_b && LF_ON( 8 ))
		{
			// The row goes into the buffer of row u - 1, whose pixels the step
			// before read into the window -- and nothing has waited for that
			// read yet when the step had no output (a chunk's first five
			// steps). Under a full chip the LDS queue can hold a read longer
			// than a row that sits in L2 takes to arrive: the read then
			// returned the NEW row (found by the round-5 fuzz campaign at
			// 834x546 -> 2030x1112, chunks of ten and more output rows, nine
			// runs in ten; the kernel had carried the race since round 4).
			// (-DLF_OLD_RACE restores the round-4 kernel WITHOUT this wait: the
			// known-bad build the differential harness must catch, `make lfrace`)
#ifndef LF_OLD_RACE
			AVIRHIP_WAIT_LGKM( 0 );
#endif
			issue_row( u + LF_NB - 1, ( buf + LF_NB - 1 ) & ( LF_NB - 1 ));
		}

		// Row u has landed when at most the LF_NB - 1 rows after it are
		// outstanding. The stores in between count on vmcnt too, so the fixed
		// count also waits for some of them and for rows nearer than u + 7 --
		// which the deep queue can afford; an exact count (rows + stores since)
		// needs a 24-way branch ladder per step, which measured slower than
		// what it saves.
		if( u_b - u >= LF_NB - 1 )
		{
			gp_wait_vm(( LF_NB - 1 ) * ( RAW ? TDN : 1 ));
		}
		else
		{
			gp_wait_vm(( u_b - u ) * ( RAW ? TDN : 1 ));
		}

		win[ i & ( RW - 1 )] = row_px( buf );
		drain( u, IC );
	};

	if( !LF_ON( 16 )) { GP_WAIT_VM( 0 ); return; }
	for( int ub = u_a; ub <= u_b; ub += 8 )
	{
		step( ub + 0, std::integral_constant< int, 0 >() );
		step( ub + 1, std::integral_constant< int, 1 >() );
		step( ub + 2, std::integral_constant< int, 2 >() );
		step( ub + 3, std::integral_constant< int, 3 >() );
		step( ub + 4, std::integral_constant< int, 4 >() );
		step( ub + 5, std::integral_constant< int, 5 >() );
		step( ub + 6, std::integral_constant< int, 6 >() );
		step( ub + 7, std::integral_constant< int, 7 >() );
	}

	GP_WAIT_VM( 0 ); // no LDS-DMA may outlive the workgroup's LDS
}

// Output rows [row_lo, row_hi); P comes filled in by gpass.hip but for the
// launch geometry.
int lfuse_launch( LFParams& P, hipStream_t st )
{
	const int rows = P.row_hi - P.row_lo;
	const size_t lds_fix = (size_t) ( LF_NB * 64 + 64 ) * 16;
	const int max_chunk = ( LF_TE * 64 ) / 8; // 6 KiB of table: 192 rows
	const size_t lds_max = lds_fix + (size_t) max_chunk * ( 8 + 1 ) * 4 + 64;
	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_max + 512 ))));

	// (a chunk's run-in: six row loads and the table staging, about three
	// output rows' worth)
	// (read per call: tools/stress_geom.py varies it case by case)
	const int force = ( getenv( "AVIRHIP_LF_CHUNK" ) != nullptr ?
		atoi( getenv( "AVIRHIP_LF_CHUNK" )) : 0 );

	P.chunk = ( force > 0 ? std::min( force, max_chunk ) :
		balanced_chunk( rows, P.nstrips, 4, max_chunk, 3, wpc, true ));

	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const size_t lds = lds_fix + (size_t) P.chunk * ( 8 + 1 ) * 4 + 64;
	static const int dbg = ( getenv( "AVIRHIP_LF_DBG" ) != nullptr ?
		atoi( getenv( "AVIRHIP_LF_DBG" )) : 0 );
	P.dbg = dbg;
	const dim3 grid( P.nstrips * nch );

	const int src_kind = ( P.raw == nullptr ? 0 : ( P.raw_kind == 1 ? 1 :
		P.raw_tdn + 1 ));
	const int out_kind = ( P.lout.on != 0 && P.lout.type == AVIRHIP_U8 &&
		P.lout.ch >= 3 && src_kind <= 1 ? P.lout.ch - 2 : 0 );
	// (the order at compile time where the kinds above leave one variant)
	const int lord = P.vx.lanc;
	const int lk = (( out_kind == 1 && lord == 3 ) || ( out_kind != 1 &&
		src_kind <= 1 && lord == 4 ) ? lord : 0 );

#define LF_GO( S, O, L ) hipLaunchKernelGGL(( k_lf< S, O, L > ), grid, dim3( 64 ), lds, st, P )
	switch( src_kind * 100 + out_kind * 10 + lk )
	{
		case 0: LF_GO( 0, 0, 0 ); break;
		case 4: LF_GO( 0, 0, 4 ); break;
		case 10: LF_GO( 0, 1, 0 ); break;
		case 13: LF_GO( 0, 1, 3 ); break;
		case 20: LF_GO( 0, 2, 0 ); break;
		case 24: LF_GO( 0, 2, 4 ); break;
		case 100: LF_GO( 1, 0, 0 ); break;
		case 104: LF_GO( 1, 0, 4 ); break;
		case 110: LF_GO( 1, 1, 0 ); break;
		case 113: LF_GO( 1, 1, 3 ); break;
		case 120: LF_GO( 1, 2, 0 ); break;
		case 124: LF_GO( 1, 2, 4 ); break;
		case 200: LF_GO( 2, 0, 0 ); break;
		case 300: LF_GO( 3, 0, 0 ); break;
		case 400: LF_GO( 4, 0, 0 ); break;
		default: return( 1 );
	}
#undef LF_GO

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// Whether k_lf can read the owner's image as it is: dword-aligned base and row
// pitch (a segment starts at the dword at or below its first pixel), byte
// offsets in 31 bits.
bool lfuse_takes_raw( const void* raw, int type, int ch, long stride,
	int in_len_v, int width )
{
	const long esz = ( type == AVIRHIP_U8 ? 1 : ( type == AVIRHIP_U16 ? 2 : 4 ));
	const long bytes = ( (long) ( in_len_v - 1 ) * stride + (long) width * ch ) * esz;

	return( raw != nullptr && ( (uintptr_t) raw & 3 ) == 0 &&
		(( stride * esz ) & 3 ) == 0 && bytes >= 4 && bytes <= 0x7ffffffcL &&
		ch >= 1 && ch <= 4 && esz * ch < 16 );
}

} // namespace avirhip
