// sacc.hip -- the downsizing pass kernel: STREAMING ACCUMULATION.
//
// A downsizing axis (AVIR k >= 2: doResize's NT-tap dot products, then the
// correction FIR, avir.h:3884-4110 + 3748-3866; LANCIR k > 1, lancir.h:
// 2466-2544) has long filters -- NT grows with k, 24 .. 64 taps -- but few
// outputs alive at a time: an input sample feeds about NT / k = 12 .. 14
// outputs whatever k is. The gather kernels of gpass.hip keep an NT-deep
// window per lane (LDS rings, 33 ds_read_b128 per output, one or two waves
// per SIMD); here the window does not exist at all:
//
//   * a lane owns one line of samples ALONG the axis (vertical pass: a pixel
//     column of a 64-pixel strip; horizontal pass: an image ROW of a 64-row
//     strip) and walks it sample by sample;
//   * sample u is multiplied into the accumulators of the outputs alive at u.
//     Samples arrive in ascending order, so every accumulator sees its taps in
//     the reference's order, starting from +0.0f: bit-exact;
//   * the coefficients of the (outputs alive) x (sample) pairs are wave-uniform:
//     the host lays them out by SAMPLE (16 floats per sample, first alive output
//     first, their count last); table rows travel through an LDS FIFO by
//     LDS-DMA, a sample's row is one ds_read_b32 (lane i: entry i), entry i
//     becomes the scalar operand of the multiply by v_readlane;
//   * accumulator slots are compile-time: the loop over outputs is unrolled 16
//     times (output g lives in slot g % 16), the loop over the samples that
//     complete an output is dynamic.
//
// Both passes are the same kernel: only the lane and step strides differ.
// Samples reach the lanes through a small LDS FIFO: float RGBA sources by
// LDS-DMA in groups of four samples (column lanes: 4 rows x 64 pixels; row
// lanes: 16 rows x 4 consecutive samples per instruction), integer / narrower
// sources (the caller's image, first pass only) by per-lane wide loads of
// whole 4-, 8-, 12- or 16-byte groups converted in registers -- packScanline's
// (float) cast and zero padding (avir.h:2777-2840) without a float copy of
// the source. Row lanes collect four outputs in LDS and store 64 contiguous
// bytes per row.

#include "dnf_dev.h"
#include <algorithm>
#include <type_traits>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <map>
#include <array>
#include <vector>

namespace avirhip {

#define SA_NS 16 // accumulator slots (at most 15 outputs alive at a sample)
#define SA_TQ 4  // coefficient-table FIFO: groups of 4 table rows (256 B)

struct SAAxis
{
	int nt, in_len, g_len, out_len, e, lanc;
	float f[ 4 ];
	int mmax;           // zero-stuffed view (the 2x upsample before a 1 < k < 2
	                    // resize, folded into the gather): samples above read 0
	int u_min;          // view index of table row 0
	const int* start;   // [g_len] first input of each output (unclamped)
	const int* first;   // [rows] first output alive at the sample
	const float* tab;   // [rows][16]: coefficients of the outputs alive at the
	                    // sample, first alive output first; [15] = their count
	const float* tab2;  // k_sacc2 (nullptr: not available): rows of 16, entry s
	                    // = the coefficient of the alive output g with g % 16 == s,
	                    // 0 elsewhere; addressed through ridx
	int tbase;          // (a multiple of 4, <= u_min)
	const int* ridx;    // k_sacc2: byte offset into tab2 of sample ( tbase + i )'s
	                    // row -- equal rows are stored once (a ratio like 27/10
	                    // has 216 distinct ones: the scalar cache holds them)
};

struct SAParams
{
	const void* src; long s_lane, s_step; // bytes between lanes / samples
	float* dst; long d_lane, d_step;      // floats between lanes / outputs
	int lane_lo, lane_hi;                 // lanes [lo, hi): rows or pixels
	int out_lo, out_hi;                   // outputs [lo, hi)
	int nstrips, chunk;
	GPOut out;       // on != 0 (column lanes): integer output stage
	SAAxis ax;
	// Float sources on the branch-free kernels (k_sacc2 / k_sacc2v) run
	// OPTIMISTICALLY: those kernels are only exact for finite samples, so the
	// last pass raises *flag_set when any value it emits is not finite -- a
	// non-finite sample inside the call's window always reaches an output as
	// NaN or Inf (c * Inf and 0 * Inf are never finite) -- and the exact kernels
	// (k_sacc) follow in the stream behind `guard`: they return at once while
	// *guard == 0 and recompute both passes otherwise.
	unsigned int* flag_set;
	unsigned int* flag_clear; // (the first optimistic pass lowers the alarm)
	const unsigned int* guard;
	int nitems;      // work items (k_sacc: walked by gridDim.x workgroups)
};

typedef unsigned int sa_u32u __attribute__(( aligned( 1 )));

template< int SRC > struct SAElem;
template<> struct SAElem< 1 > { typedef uint8_t T; };
template<> struct SAElem< 2 > { typedef uint16_t T; };
template<> struct SAElem< 3 > { typedef float T; };
template<> struct SAElem< 0 > { typedef float T; };

// A pixel's NC live channels. Packed float operations take as long as two
// scalar ones on this hardware (profiles/r01_microbench_valu_lds_copy.txt), so
// a channel that is not there costs real time: RGB runs on one packed pair and
// one scalar (3/4 of the RGBA arithmetic), padded channels are never computed
// -- they are written as the +0.0f the reference's zero padding would yield.
template< int NC >
struct SAV
{
	f2 a;    // channels 0, 1
	f2 b;    // channels 2, 3 (NC == 3: only b.x is computed)

	__device__ __forceinline__ static SAV zero()
	{
		SAV r; r.a = (f2) 0.0f; r.b = (f2) 0.0f; return( r );
	}

	__device__ __forceinline__ static SAV from( const f4 v )
	{
		SAV r; r.a = v.xy; r.b = v.zw; return( r );
	}

	__device__ __forceinline__ f4 to4() const
	{
		f4 r; r.xy = a;
		r.z = ( NC >= 3 ? b.x : 0.0f );
		r.w = ( NC >= 4 ? b.y : 0.0f );

		if( NC < 2 )
		{
			r.y = 0.0f;
		}

		return( r );
	}

	// this + c * x (mul, then add: no contraction)
	__device__ __forceinline__ SAV madd( const float c, const SAV& x ) const
	{
		SAV r = *this;

		if( NC >= 2 )
		{
			r.a = a + c * x.a;
		}
		else
		{
			r.a.x = a.x + c * x.a.x;
		}

		if( NC >= 4 )
		{
			r.b = b + c * x.b;
		}
		else
		if( NC == 3 )
		{
			r.b.x = b.x + c * x.b.x;
		}

		return( r );
	}

	// this + x
	__device__ __forceinline__ SAV madd1( const SAV& x ) const
	{
		SAV r = *this;

		if( NC >= 2 ) r.a = a + x.a; else r.a.x = a.x + x.a.x;
		if( NC >= 4 ) r.b = b + x.b; else if( NC == 3 ) r.b.x = b.x + x.b.x;
		return( r );
	}

	// c * this
	__device__ __forceinline__ SAV scale( const float c ) const
	{
		SAV r = *this;

		if( NC >= 2 ) r.a = c * a; else r.a.x = c * a.x;
		if( NC >= 4 ) r.b = c * b; else if( NC == 3 ) r.b.x = c * b.x;
		return( r );
	}
};

// LDS-DMA of 64 x 4 bytes (a group of four coefficient-table rows)
__device__ __forceinline__ void sa_dma4( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"buffer_load_dword %1, %2, 0 offen lds"
		:: "s"( __builtin_amdgcn_readfirstlane( lds )), "v"( voff ), "s"( rs )
		: "memory" );
}

// SRC: 0 float RGBA by LDS-DMA; 1 uint8, 2 uint16, 3 float of CH channels
// (lane = image row only). HOR (SRC 0): lanes are image rows -- a DMA
// instruction then fetches 16 rows x 4 consecutive samples (64 contiguous
// bytes per row) instead of 64 rows x 1 sample: four times fewer cache lines
// per instruction.
template< int SRC, int CH, bool POST, bool HOR, int NC >
__global__ void __launch_bounds__( 64 ) k_sacc( const SAParams P )
{
	typedef SAV< NC > PV;
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const SAAxis& A = P.ax;
	// SRC 0: three groups of four samples (4 KiB each); typed: 8 samples
	constexpr int DQ_BYTES = ( SRC == 0 ? 3 * 4096 : 8 * 1024 );
	f4* const fifo = (f4*) smem;
	float* const tfifo = (float*) ( smem + DQ_BYTES ); // [SA_TQ][4][16] + pad

	if( P.guard != nullptr &&
		*(const __attribute__(( address_space( 4 ))) unsigned int*)
		(unsigned long) P.guard == 0 )
	{
		return; // (the optimistic passes' results stand)
	}

	// (a launch behind a guard is a small grid of workgroups that walk the work
	// items: 130,000 workgroups that return at once still cost 5 us to
	// dispatch; every other launch has one item per workgroup)
	for( int item = blockIdx.x; item < P.nitems; item += gridDim.x )
	{
	const int lane = threadIdx.x;
	const int strip = item % P.nstrips;
	const int chunk = item / P.nstrips;
	const int l0 = P.lane_lo + strip * 64;
	const int li = l0 + lane;
	const bool lok = ( li < P.lane_hi );
	const int llast = P.lane_hi - 1;
	const int lc = min( li, llast );               // lanes past the end reload
	const int n0 = P.out_lo + chunk * P.chunk;
	const int n1 = min( n0 + P.chunk, P.out_hi );

	if( n0 >= n1 )
	{
		continue;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	const int nt = A.nt;
	const int glast = A.g_len - 1;
	const int slast = A.in_len - 1;

	// accumulated outputs this chunk needs: [ga, gb]; POST: the correction
	// FIR reads +-3 around every output, beyond the ends the edge value --
	// gbv counts those virtual repeats too
	const int ga = ( POST ? gp_clamp( n0 - A.e - 3, glast ) : n0 );
	const int gb = ( POST ? gp_clamp( n1 - 1 - A.e + 3, glast ) : n1 - 1 );
	const int gbv = ( POST ? n1 - 1 - A.e + 3 : gb );
	const int u_a = a_start[ ga ];
	const int u_b = a_start[ gb ] + nt - 1;
	(void) u_b;

	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) smem;

	// ---- the coefficient-table feed: groups of four rows by LDS-DMA, three
	// groups ahead of the one being read
	const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
		(void*) A.tab, 0, 0x7fffffff, 0x00020000 );

	auto t_issue = [&]( const int q ) // group q = table rows 4q .. 4q + 3
	{
		sa_dma4( trs, lds0 + DQ_BYTES + (unsigned) ( q & ( SA_TQ - 1 )) * 256,
			q * 256 + lane * 4 );
	};

	// this lane's element of sample u's row, `lo` entries in
	auto t_read = [&]( const int u, const int lo ) -> float
	{
		const int r = u - A.u_min;
		return( tfifo[ (( r >> 2 ) & ( SA_TQ - 1 )) * 64 + ( r & 3 ) * 16 + lo +
			( lane & 15 )]);
	};

	// ---- the sample feed
	constexpr int G = ( SRC == 1 ? 4 : ( SRC == 2 ? 2 : ( SRC == 0 ? 4 : 1 )));
	typedef typename SAElem< SRC > :: T Tin;
	// groups are aligned in absolute sample coordinates
	const int ub0 = ( u_a >= 0 ? u_a / G : -(( -u_a + G - 1 ) / G )) * G;
	const int q_a = ub0 / G;

	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int sstep = (int) P.s_step;
	const int slane = (int) P.s_lane;

	// SRC 0: group q = samples 4q .. 4q + 3, four DMA instructions, in FIFO
	// third ( q - q_a ) % 3
	int hrow[ 4 ]; // HOR: byte offsets of the four rows this lane fetches
#pragma unroll
	for( int k = 0; k < 4; k++ )
	{
		hrow[ k ] = min( l0 + k * 16 + ( lane >> 2 ), llast ) * slane;
	}

	int ith = 0; // FIFO third of the next group issued (groups go out in order)
	int rth = 0; // FIFO third of the group holding the sample read next

	auto d_issue = [&]( const int q )
	{
		const unsigned base = lds0 + (unsigned) ith * 4096;
		ith = ( ith == 2 ? 0 : ith + 1 );

		// (groups past the last sample are fetched like any other -- clamped
		// addresses, nobody reads them: the count of operations per group
		// stays uniform and there is one copy of this code)
		if( HOR )
		{
			const int x = gp_clamp( q * 4 + ( lane & 3 ), slast ) * sstep;
#pragma unroll
			for( int k = 0; k < 4; k++ )
			{
				gp_dma( rsrc, base + k * 1024, hrow[ k ] + x );
			}
		}
		else
		{
#pragma unroll
			for( int k = 0; k < 4; k++ )
			{
				gp_dma( rsrc, base + k * 1024, lc * slane +
					gp_clamp( q * 4 + k, slast ) * sstep );
			}
		}
	};

	auto d_read = [&]( const int u ) -> f4 // SRC 0; samples in order
	{
		const int r = u - ub0;

		if( u != u_a && ( r & 3 ) == 0 )
		{
			rth = ( rth == 2 ? 0 : rth + 1 );
		}

		const int th = rth;

		if( HOR )
		{
			return( fifo[ th * 256 + ( lane >> 4 ) * 64 + ( lane & 15 ) * 4 +
				( r & 3 )]);
		}

		return( fifo[ th * 256 + ( r & 3 ) * 64 + lane ]);
	};

	const unsigned char* const lrow = (const unsigned char*) P.src +
		(long) lc * P.s_lane;

	unsigned int gw[ CH ]; // the group in flight (typed sources)
#pragma unroll
	for( int k = 0; k < CH; k++ )
	{
		gw[ k ] = 0;
	}

	auto g_interior = [&]( const int q ) -> bool
	{
		return( q * G >= 0 && q * G + G - 1 <= slast );
	};

	// typed: whole group, one wide load -- UNCONDITIONAL (groups on the edge
	// or past the data read a clamped address and are not used: a load under
	// a condition made the compiler wait for it on the spot, every group at
	// full memory latency); the image is at least one group wide (host check)
	auto g_load = [&]( const int q )
	{
		const int gs = max( 0, min( q * G, slast + 1 - G ));
		const sa_u32u* const gp = (const sa_u32u*) ( lrow + (long) gs * CH *
			sizeof( Tin ));
#pragma unroll
		for( int k = 0; k < CH; k++ )
		{
			gw[ k ] = gp[ k ];
		}
	};

	auto g_store = [&]( const int q ) // typed: convert, write the FIFO slots
	{
		const int sl0 = ( q * G - ub0 ) & 7;

		if( g_interior( q ))
		{
#pragma unroll
			for( int j = 0; j < G; j++ )
			{
				float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
				for( int c = 0; c < CH; c++ )
				{
					const int ei = j * CH + c;

					if( SRC == 1 )
					{
						v[ c ] = (float) (( gw[ ei >> 2 ] >> (( ei & 3 ) * 8 )) &
							0xffu );
					}
					else
					if( SRC == 2 )
					{
						v[ c ] = (float) (( gw[ ei >> 1 ] >> (( ei & 1 ) * 16 )) &
							0xffffu );
					}
					else
					{
						v[ c ] = __uint_as_float( gw[ ei ]);
					}
				}

				f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
				fifo[ ( sl0 + j ) * 64 + lane ] = o;
			}
		}
		else
		{
			// a group on the image's edge: the view repeats the edge sample
			for( int j = 0; j < G; j++ )
			{
				const int x = gp_clamp( q * G + j, slast );
				const Tin* const px = (const Tin*) lrow + (long) x * CH;
				float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
				for( int c = 0; c < CH; c++ )
				{
					v[ c ] = (float) px[ c ];
				}

				f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
				fifo[ ( sl0 + j ) * 64 + lane ] = o;
			}
		}
	};

	auto x_read = [&]( const int u ) -> f4
	{
		// (d_read counts the samples: called for every u)
		const f4 v = ( SRC == 0 ? d_read( u ) :
			fifo[ (( u - ub0 ) & 7 ) * 64 + lane ]);

		// a zero-stuffed view reads 0 above its last replicated sample
		return( u > A.mmax ? (f4) 0.0f : v );
	};

	// ---- prologue: the first output's samples may be alive for outputs before
	// ga too (their row entries come first): `first` of the chunk's first 64
	// samples, one per lane
	const int tq_a = ( u_a - A.u_min ) >> 2;
	t_issue( tq_a );
	t_issue( tq_a + 1 );
	t_issue( tq_a + 2 );
	int tq_next = tq_a + 3; // typed sources: the next table group to go out

	if( SRC != 0 || (( u_a - A.u_min ) & 3 ) != 0 )
	{
		t_issue( tq_a + 3 ); // (as if group tq_a had just been entered)
		tq_next = tq_a + 4;
	}

	const int fv = A.first[ u_a - A.u_min + lane ]; // (table padded by 64 rows)

	if( SRC == 0 )
	{
		d_issue( q_a );
		d_issue( q_a + 1 );

		if( u_a != ub0 )
		{
			d_issue( q_a + 2 );
		}
	}
	else
	{
		// (as if group q_a had just been entered: the next one is in the FIFO
		// too, the one after it in flight)
		g_load( q_a );
		g_store( q_a );
		g_load( q_a + 1 );
		g_store( q_a + 1 );
		g_load( q_a + 2 );
	}

	// everything issued so far has to land before the first reads (the
	// compiler's own wait for `fv` does not count the LDS-DMA operations)
	GP_WAIT_VM( 0 );

	auto lo_of = [&]( const int u ) -> int
	{
		const int i = u - u_a;
		return( i < 64 ? max( 0, ga - __builtin_amdgcn_readlane( fv, i )) : 0 );
	};

	PV xnext = PV::from( x_read( u_a ));
	int lon = lo_of( u_a );
	float cnext = t_read( u_a, lon );

	// ---- the accumulators and (POST) the correction FIR's window
	PV acc[ SA_NS ];
#pragma unroll
	for( int i = 0; i < SA_NS; i++ )
	{
		acc[ i ] = PV::zero();
	}

	PV w[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		w[ i ] = PV::zero();
	}

	int g = ga;                    // the output being completed
	int sg = u_a;                  // its first input
	int sgn = a_start[ min( g + 1, gb )];
	int u = u_a;                   // the next sample

	const float f0 = A.f[ 0 ], f1c = A.f[ 1 ], f2c = A.f[ 2 ], f3c = A.f[ 3 ];

	float* dp = P.dst + (long) li * P.d_lane + (long) n0 * P.d_step;

	// Row lanes: a lane's outputs are consecutive in memory, a store of one
	// output per lane would write 64 x 16 bytes to 64 different rows -- the
	// first version did, and the counters showed 290 MB written for the
	// 106 MB intermediate of the README photo case (partial lines evicted and
	// re-merged) and 400 MB fetched for a 54 MB source. Four outputs are
	// collected in LDS and leave as 64 contiguous bytes per row, 16 rows per
	// store instruction.
	constexpr bool ROWS = ( HOR || SRC != 0 );
	f4* const ofifo = (f4*) ( smem + DQ_BYTES + SA_TQ * 256 + 64 ); // [4][64]
	int n_out = n0;
	const int orow = l0 + ( lane >> 2 ); // + 16 k: the rows this lane stores

	auto store_out = [&]( const PV& o )
	{
		if( !ROWS )
		{
			if( P.out.on != 0 )
			{
				gp_store_int_row( P.out, n_out - P.out_lo, li, o.to4(), lok );
				n_out++;
				return;
			}

			if( lok )
			{
				*(f4*) dp = o.to4();
			}

			dp += P.d_step;
			return;
		}

		ofifo[ ( n_out & 3 ) * 64 + lane ] = o.to4();

		if(( n_out & 3 ) == 3 || n_out == n1 - 1 )
		{
			const int nb = n_out & ~3;
			const int n = nb + ( lane & 3 );
			const bool nok = ( n >= n0 && n <= n_out );
#pragma unroll
			for( int k = 0; k < 4; k++ )
			{
				const f4 v = ofifo[ ( lane & 3 ) * 64 + k * 16 + ( lane >> 2 )];
				const int r = orow + k * 16;

				if( nok && r < P.lane_hi )
				{
					*(f4*) ( P.dst + (long) r * P.d_lane + (long) n * 4 ) = v;
				}
			}
		}

		n_out++;
	};

	// One output: the samples up to its last input, then the accumulator
	// leaves. SC: its slot (compile-time); FC: the chunk's first output (its
	// samples' table rows may start before ga). Returns true after the last.
	auto body = [&]( auto SC, auto FC ) -> bool
	{
		constexpr int S = decltype( SC )::value;
		constexpr bool FIRST = decltype( FC )::value;
		const int end = ( g <= gb ? sg + nt - 1 : u - 1 );

#pragma clang loop unroll( disable )
		while( u <= end )
		{
			// -- the feeds
			if( SRC == 0 )
			{
				if((( u - A.u_min ) & 3 ) == 0 )
				{
					t_issue((( u - A.u_min ) >> 2 ) + 3 );
				}

				if((( u - ub0 ) & 3 ) == 0 )
				{
					// entering a group: the one after the next goes out; the
					// next one (read from the last sample of this one on) was
					// issued a group ago -- only the four (five, with a table
					// group) youngest operations may still be in flight
					d_issue( q_a + (( u - ub0 ) >> 2 ) + 2 );
					GP_WAIT_VM( 4 );
				}
			}
			else
			if((( u - ub0 ) & ( G - 1 )) == 0 && u != u_a )
			{
				const int q = q_a + (( u - ub0 ) / G );
				g_store( q + 1 );
				g_load( q + 2 );

				// Table groups go out HERE, after this trigger's loads: the
				// compiler's wait for those loads at the next trigger counts
				// operations it does not know of -- an LDS-DMA issued just
				// before a wait would be waited for at full memory latency
				// (44 % of the wave-cycles of the first version were waits).
				// Three groups of lookahead: a group is read two triggers
				// after it went out at the earliest.
				while( tq_next <= (( u - A.u_min ) >> 2 ) + 3 )
				{
					t_issue( tq_next );
					tq_next++;
				}
			}

			const PV x = xnext;
			const float cv = cnext;
			const int lo = lon;
			xnext = PV::from( x_read( u + 1 ));
			lon = ( FIRST ? lo_of( u + 1 ) : 0 );
			cnext = t_read( u + 1, lon );

			// -- sample u into the outputs alive: g .. g + nl - 1
			const int cnt = __builtin_amdgcn_readlane( __float_as_int( cv ),
				15 - lo );

			const int nl = min( cnt - lo, gb - g + 1 );

			// One wave-uniform branch per alive output, nested: the first
			// output that is not alive leaves the chain. (A switch with
			// fall-through entries was structurised into a decision tree with
			// thousands of register copies; un-nested ifs were if-converted
			// into all 16 products and four selects each -- the empty asm
			// statement keeps every block a real branch.)
#define SA_T( i, inner ) if( i < nl ) { asm volatile( "" ::: "memory" ); \
	acc[ ( S + i ) & ( SA_NS - 1 )] = acc[ ( S + i ) & ( SA_NS - 1 )].madd( \
	__int_as_float( __builtin_amdgcn_readlane( __float_as_int( cv ), i )), x ); \
	inner }

			SA_T( 0, SA_T( 1, SA_T( 2, SA_T( 3, SA_T( 4, SA_T( 5, SA_T( 6,
			SA_T( 7, SA_T( 8, SA_T( 9, SA_T( 10, SA_T( 11, SA_T( 12,
			SA_T( 13, SA_T( 14, ; )))))))))))))))

#undef SA_T

			u++;
		}

		if( POST )
		{
			// the window of the last eight accumulated outputs; past the last
			// one the edge value repeats, before the first one it does too
			// (not a ?: on the two array elements: that selects an ADDRESS and
			// keeps both arrays in scratch memory)
			PV val = acc[ S ];

			if( g > gb )
			{
				val = w[ ( S + 7 ) & 7 ];
			}
			w[ S & 7 ] = val;

			if( FIRST && n0 - A.e - 3 < 0 )
			{
#pragma unroll
				for( int i = 0; i < 8; i++ )
				{
					w[ i ] = val;
				}
			}

			// output n = g + e - 3: f0*G[c] + sum f_i*( G[c+i] + G[c-i] ), c = g - 3
			if( g >= n0 - A.e + 3 )
			{
				PV r = w[ ( S + 5 ) & 7 ].scale( f0 );
				r = r.madd( f1c, w[ ( S + 6 ) & 7 ].madd1( w[ ( S + 4 ) & 7 ]));
				r = r.madd( f2c, w[ ( S + 7 ) & 7 ].madd1( w[ ( S + 3 ) & 7 ]));
				r = r.madd( f3c, w[ S & 7 ].madd1( w[ ( S + 2 ) & 7 ]));
				store_out( r );
			}
		}
		else
		{
			store_out( acc[ S ]);
		}

		acc[ S ] = PV::zero();
		g++;
		sg = sgn;
		sgn = a_start[ min( g + 1, gb )];
		return( g > gbv );
	};

	typedef std::integral_constant< bool, true > TC;
	typedef std::integral_constant< bool, false > FC;

	if( !body( std::integral_constant< int, 0 >(), TC() ))
	{
		for( ; ; )
		{
			if( body( std::integral_constant< int, 1 >(), FC() )) break;
			if( body( std::integral_constant< int, 2 >(), FC() )) break;
			if( body( std::integral_constant< int, 3 >(), FC() )) break;
			if( body( std::integral_constant< int, 4 >(), FC() )) break;
			if( body( std::integral_constant< int, 5 >(), FC() )) break;
			if( body( std::integral_constant< int, 6 >(), FC() )) break;
			if( body( std::integral_constant< int, 7 >(), FC() )) break;
			if( body( std::integral_constant< int, 8 >(), FC() )) break;
			if( body( std::integral_constant< int, 9 >(), FC() )) break;
			if( body( std::integral_constant< int, 10 >(), FC() )) break;
			if( body( std::integral_constant< int, 11 >(), FC() )) break;
			if( body( std::integral_constant< int, 12 >(), FC() )) break;
			if( body( std::integral_constant< int, 13 >(), FC() )) break;
			if( body( std::integral_constant< int, 14 >(), FC() )) break;
			if( body( std::integral_constant< int, 15 >(), FC() )) break;
			if( body( std::integral_constant< int, 0 >(), FC() )) break;
		}
	}

	GP_WAIT_VM( 0 ); // no LDS-DMA may outlive the workgroup's LDS (or the item)
	}
}

// ---------------------------------------------------------------------
// k_sacc2 -- the same pass, branch-free per sample, for sources known to be
// FINITE (integer images, and the FltBuf made from one).
//
// k_sacc spends ~1000 cycles per sample on ~210 cycles of arithmetic: a
// compare ladder over the outputs alive (one wave-uniform branch per output),
// a v_readlane per coefficient, samples and table rows through LDS FIFOs.
// Here a table row holds one coefficient per accumulator SLOT (output g lives
// in slot g % 16) and zeros for the slots whose output is not alive, so a
// sample is multiplied into ALL 16 slots unconditionally: x * 0 = +-0 leaves a
// finite accumulator as it is (an accumulator is never -0: it starts from +0
// and (+0) + (-0) = +0), which is why the sources must be finite -- a 0 * inf
// would put a NaN into a slot between two of its outputs. What this buys:
//   * no branch and no LDS access per sample: the row is ONE s_load_dwordx16
//     per two samples, issued two samples ahead (lgkmcnt counts nothing else),
//     its entries are the scalar operands of the packed multiplies (op_sel
//     picks a half of an SGPR pair);
//   * RGB: the third channel of TWO slots shares one packed multiply-add (the
//     coefficient pair as it lies x the channel broadcast): 3 instead of 4
//     instructions per slot;
//   * samples arrive by per-lane loads of eight pixels of the lane's row (RGB
//     uint8: two dwordx3), one group in flight while the other is used; the
//     four samples up next always sit in the same registers, so every position
//     is compile-time and nothing goes through LDS;
//   * an output leaves its accumulator AFTER the block of four samples in
//     which its last tap fell (its later coefficients are zeros), outside the
//     sample code: the loop over outputs is unrolled 16 times with the slot
//     compile-time (a 16-way switch in one copy of the sample loop copied all
//     48 accumulator registers in and out of every case). The host checks
//     that the slot's next output starts after that block (sacc_prepare_axis).
// Same order of operations per output as k_sacc => the same bits.
// ---------------------------------------------------------------------

typedef float f16v __attribute__(( ext_vector_type( 16 )));

#define SA2_LGKM_WAIT() AVIRHIP_WAIT_LGKM( 0 )

// W dwords as one register tuple
template< int W > struct SA2Vec
{
	typedef unsigned int T __attribute__(( ext_vector_type( W )));
	__device__ __forceinline__ static unsigned int get( const T& v, const int i )
		{ return( v[ i ]); }
	__device__ __forceinline__ static void orr( T& v, const int i,
		const unsigned int x ) { v[ i ] |= x; }
};

// A per-lane global load whose destination the compiler does not treat as a
// load result ("+v": the group in flight lives in ONE register tuple across
// the loop; as a plain load its loop-carried copies were made right after the
// load went out, at full memory latency). sa2_landed() is the wait: every use
// of the group depends on it.
template< int W, int OFF, class V >
__device__ __forceinline__ void sa2_gload( V& d, const unsigned char* const p )
{
	if constexpr( W == 1 )
		asm volatile( "global_load_dword %0, %1, off offset:%2"
			: "+v"( d ) : "v"( p ), "n"( OFF ) : "memory" );
	else if constexpr( W == 2 )
		asm volatile( "global_load_dwordx2 %0, %1, off offset:%2"
			: "+v"( d ) : "v"( p ), "n"( OFF ) : "memory" );
	else if constexpr( W == 3 )
		asm volatile( "global_load_dwordx3 %0, %1, off offset:%2"
			: "+v"( d ) : "v"( p ), "n"( OFF ) : "memory" );
	else
		asm volatile( "global_load_dwordx4 %0, %1, off offset:%2"
			: "+v"( d ) : "v"( p ), "n"( OFF ) : "memory" );
}

template< class V >
__device__ __forceinline__ void sa2_landed( V& a, V& b )
{
	asm volatile( AVIRHIP_WAITCNT_VM( 0 ) : "+v"( a ), "+v"( b ) :: "memory" );
}

template<> struct SA2Vec< 1 >
{
	typedef unsigned int T;
	__device__ __forceinline__ static unsigned int get( const T& v, const int )
		{ return( v ); }
	__device__ __forceinline__ static void orr( T& v, const int,
		const unsigned int x ) { v |= x; }
};

template< int NC >
struct SA2Acc
{
	f2 A[ 16 ];  // channels 0, 1 of slot s            (NC >= 2)
	f2 B[ 16 ];  // channels 2, 3 of slot s            (NC == 4)
	f2 BB[ 8 ];  // the odd channel of slots 2i, 2i + 1 (NC == 3: 2; NC == 1: 0)
};

// four slots 4q .. 4q + 3 of one sample; c0 = { c[4q], c[4q+1] }, c1 the next two
template< int NC >
__device__ __forceinline__ void sa2_quad( f2* const A, f2* const B,
	f2* const BB, const f2 xa, const f2 xb, const f2 c0, const f2 c1 )
{
	f2 p0, p1, p2, p3, p4, p5, p6, p7;

	if constexpr( NC == 4 )
	{
		asm volatile(
			"v_pk_mul_f32 %[p0], %[xa], %[c0]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p1], %[xb], %[c0]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p2], %[xa], %[c0]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p3], %[xb], %[c0]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p4], %[xa], %[c1]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p5], %[xb], %[c1]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p6], %[xa], %[c1]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p7], %[xb], %[c1]" DF_SEL1 "\n\t"
			"v_pk_add_f32 %[a0], %[a0], %[p0]\n\t"
			"v_pk_add_f32 %[b0], %[b0], %[p1]\n\t"
			"v_pk_add_f32 %[a1], %[a1], %[p2]\n\t"
			"v_pk_add_f32 %[b1], %[b1], %[p3]\n\t"
			"v_pk_add_f32 %[a2], %[a2], %[p4]\n\t"
			"v_pk_add_f32 %[b2], %[b2], %[p5]\n\t"
			"v_pk_add_f32 %[a3], %[a3], %[p6]\n\t"
			"v_pk_add_f32 %[b3], %[b3], %[p7]"
			: [a0] "+v"( A[ 0 ]), [a1] "+v"( A[ 1 ]), [a2] "+v"( A[ 2 ]),
			[a3] "+v"( A[ 3 ]), [b0] "+v"( B[ 0 ]), [b1] "+v"( B[ 1 ]),
			[b2] "+v"( B[ 2 ]), [b3] "+v"( B[ 3 ]),
			[p0] "=&v"( p0 ), [p1] "=&v"( p1 ), [p2] "=&v"( p2 ),
			[p3] "=&v"( p3 ), [p4] "=&v"( p4 ), [p5] "=&v"( p5 ),
			[p6] "=&v"( p6 ), [p7] "=&v"( p7 )
			: [xa] "v"( xa ), [xb] "v"( xb ), [c0] "s"( c0 ), [c1] "s"( c1 ));
	}
	else
	if constexpr( NC == 3 )
	{
		asm volatile(
			"v_pk_mul_f32 %[p0], %[xa], %[c0]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p1], %[xa], %[c0]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p2], %[c0], %[xb] op_sel_hi:[1,0]\n\t"
			"v_pk_mul_f32 %[p3], %[xa], %[c1]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p4], %[xa], %[c1]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p5], %[c1], %[xb] op_sel_hi:[1,0]\n\t"
			"v_pk_add_f32 %[a0], %[a0], %[p0]\n\t"
			"v_pk_add_f32 %[a1], %[a1], %[p1]\n\t"
			"v_pk_add_f32 %[q0], %[q0], %[p2]\n\t"
			"v_pk_add_f32 %[a2], %[a2], %[p3]\n\t"
			"v_pk_add_f32 %[a3], %[a3], %[p4]\n\t"
			"v_pk_add_f32 %[q1], %[q1], %[p5]"
			: [a0] "+v"( A[ 0 ]), [a1] "+v"( A[ 1 ]), [a2] "+v"( A[ 2 ]),
			[a3] "+v"( A[ 3 ]), [q0] "+v"( BB[ 0 ]), [q1] "+v"( BB[ 1 ]),
			[p0] "=&v"( p0 ), [p1] "=&v"( p1 ), [p2] "=&v"( p2 ),
			[p3] "=&v"( p3 ), [p4] "=&v"( p4 ), [p5] "=&v"( p5 )
			: [xa] "v"( xa ), [xb] "v"( xb ), [c0] "s"( c0 ), [c1] "s"( c1 ));
	}
	else
	if constexpr( NC == 2 )
	{
		asm volatile(
			"v_pk_mul_f32 %[p0], %[xa], %[c0]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p1], %[xa], %[c0]" DF_SEL1 "\n\t"
			"v_pk_mul_f32 %[p3], %[xa], %[c1]" DF_SEL0 "\n\t"
			"v_pk_mul_f32 %[p4], %[xa], %[c1]" DF_SEL1 "\n\t"
			"v_pk_add_f32 %[a0], %[a0], %[p0]\n\t"
			"v_pk_add_f32 %[a1], %[a1], %[p1]\n\t"
			"v_pk_add_f32 %[a2], %[a2], %[p3]\n\t"
			"v_pk_add_f32 %[a3], %[a3], %[p4]"
			: [a0] "+v"( A[ 0 ]), [a1] "+v"( A[ 1 ]), [a2] "+v"( A[ 2 ]),
			[a3] "+v"( A[ 3 ]),
			[p0] "=&v"( p0 ), [p1] "=&v"( p1 ), [p3] "=&v"( p3 ), [p4] "=&v"( p4 )
			: [xa] "v"( xa ), [c0] "s"( c0 ), [c1] "s"( c1 ));
	}
	else
	{
		asm volatile(
			"v_pk_mul_f32 %[p2], %[c0], %[xa] op_sel_hi:[1,0]\n\t"
			"v_pk_mul_f32 %[p5], %[c1], %[xa] op_sel_hi:[1,0]\n\t"
			"v_pk_add_f32 %[q0], %[q0], %[p2]\n\t"
			"v_pk_add_f32 %[q1], %[q1], %[p5]"
			: [q0] "+v"( BB[ 0 ]), [q1] "+v"( BB[ 1 ]),
			[p2] "=&v"( p2 ), [p5] "=&v"( p5 )
			: [xa] "v"( xa ), [c0] "s"( c0 ), [c1] "s"( c1 ));
	}
}

// The running state of a chunk's outputs
struct SA2State
{
	int g;      // the next output to complete
	int n_out;  // the next final output
	int ga, gb; // accumulated outputs the chunk's correction FIR reads
	int n0, n1; // the chunk's final outputs
	int e, li;
	bool lok;
	float f0, f1, f2, f3;
	float* dp;  // column lanes: the next output's pixel; row lanes: the lane's row
	unsigned long long bad; // lanes that emitted a NaN / Inf (column lanes)
};

// Output T.g leaves slot S (compile-time): through the correction FIR's
// window into the destination. ROWS: four consecutive outputs are collected
// and leave as 64 contiguous bytes of the lane's row (see k_sacc); column
// lanes store a pixel per output, or go through the integer output stage.
template< int NC, int S, bool ROWS >
__device__ __forceinline__ void sa2_emit( SA2Acc< NC >& R, SAV< NC > ( &w )[ 8 ],
	f4 ( &pend )[ 4 ], SA2State& T, const SAParams& P0 )
{
	typedef SAV< NC > PV;
	// (column lanes: the parameters are read from the kernel argument segment
	// HERE, through a pointer the compiler cannot see through: hoisted out of
	// the sample loop they sat in scalar registers the coefficient rows need
	// and came back from their spill slots by dozens of v_readlanes per
	// output. Row lanes use few of them: there the reloads cost more.)
	(void) P0;
	typedef const __attribute__(( address_space( 4 ))) SAParams CP;
	CP* Pq = (CP*) __builtin_amdgcn_kernarg_segment_ptr(); // (the only argument)

	if( !ROWS )
	{
		asm volatile( "" : "+s"( Pq ));
	}

	CP& P = *Pq;
	PV val = PV::zero();

	if( NC >= 2 ) val.a = R.A[ S ];
	else val.a.x = R.BB[ S >> 1 ][ S & 1 ];

	if( NC == 4 ) val.b = R.B[ S ];
	else if( NC == 3 ) val.b.x = R.BB[ S >> 1 ][ S & 1 ];

	R.A[ S ] = (f2) 0.0f;
	R.B[ S ] = (f2) 0.0f;
	R.BB[ S >> 1 ][ S & 1 ] = 0.0f;

	if( T.g < T.ga )
	{
		return; // alive at the chunk's first sample, not the chunk's own
	}

	if( T.g > T.gb )
	{
		val = w[ ( S + 7 ) & 7 ]; // past the last one the edge value repeats
	}

	w[ S & 7 ] = val;

	if( T.g == T.ga && T.n0 - T.e - 3 < 0 )
	{
#pragma unroll
		for( int i = 0; i < 8; i++ )
		{
			w[ i ] = val;
		}
	}

	// output n = g + e - 3: f0*G[c] + sum f_i*( G[c+i] + G[c-i] ), c = g - 3
	if( T.g < T.n0 - T.e + 3 )
	{
		return;
	}

	PV r = w[ ( S + 5 ) & 7 ].scale( T.f0 );
	r = r.madd( T.f1, w[ ( S + 6 ) & 7 ].madd1( w[ ( S + 4 ) & 7 ]));
	r = r.madd( T.f2, w[ ( S + 7 ) & 7 ].madd1( w[ ( S + 3 ) & 7 ]));
	r = r.madd( T.f3, w[ S & 7 ].madd1( w[ ( S + 2 ) & 7 ]));

	if( ROWS )
	{
		// pend[ k ] holds the output n with n % 4 == ( k + dsh ) % 4
		pend[ S & 3 ] = r.to4();

		if(( T.n_out & 3 ) == 3 || T.n_out == T.n1 - 1 )
		{
			const int dsh = ( T.n_out - S ) & 3;
			const int nb = T.n_out & ~3;
#pragma unroll
			for( int k = 0; k < 4; k++ )
			{
				const int n = nb + (( k + dsh ) & 3 );

				if( n >= T.n0 && n <= T.n_out && T.lok )
				{
					*(f4*) ( T.dp + (long) n * 4 ) = pend[ k ]; // (dp: the lane's row)
				}
			}
		}
		T.n_out++;
		return;
	}

	// column lanes -- the plan's last pass. NaN or Inf in what leaves
	// (v_cmp_class: signalling / quiet NaN, -inf, +inf) raises the optimistic
	// float path's alarm (SAParams::flag_set; gathered per wave, see the
	// kernel's end)
	{
		const f4 r4 = r.to4();
		const bool nf = ( __builtin_amdgcn_classf( r4.x, 0x207 ) ||
			__builtin_amdgcn_classf( r4.y, 0x207 ) ||
			__builtin_amdgcn_classf( r4.z, 0x207 ) ||
			__builtin_amdgcn_classf( r4.w, 0x207 ));

		T.bad |= __builtin_amdgcn_ballot_w64( nf );
	}

	if( P.out.on != 0 )
	{
		GPOut O;
		O.on = 1; O.type = P.out.type; O.ch = P.out.ch;
		O.use_tr = P.out.use_tr; O.tr_mul = P.out.tr_mul;
		O.fin = P.out.fin; O.pk_i = P.out.pk_i;
		O.tr_muli = P.out.tr_muli; O.pk_out = P.out.pk_out;
		O.base = P.out.base; O.stride = P.out.stride;
		gp_store_int_row( O, T.n_out - P.out_lo, T.li, r.to4(), T.lok );
	}
	else
	{
		if( T.lok )
		{
			*(f4*) T.dp = r.to4();
		}

		T.dp += P.d_step;
	}

	T.n_out++;
}

// SRC 1 / 2: uint8 / uint16 pixels of CH channels; lanes are image ROWS (the
// caller's image: the first pass of a plan with an integer source)
template< int SRC, int CH, int NC >
__global__ void __launch_bounds__( 64 ) k_sacc2( const SAParams P )
{
	typedef SAV< NC > PV;
	typedef typename SAElem< SRC > :: T Tin;
	const SAAxis& A = P.ax;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int l0 = P.lane_lo + strip * 64;
	const int li = l0 + lane;
	const bool lok = ( li < P.lane_hi );
	const int lc = min( li, P.lane_hi - 1 );
	const int n0 = P.out_lo + chunk * P.chunk;
	const int n1 = min( n0 + P.chunk, P.out_hi );

	if( n0 >= n1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	gp_cint* const a_first = (gp_cint*) (unsigned long) A.first;
	const int nt = A.nt;
	const int glast = A.g_len - 1;
	const int slast = A.in_len - 1;

	// accumulated outputs the chunk's correction FIR reads: [ga, gb], virtual
	// repeats of the last one up to gbv (see k_sacc)
	const int ga = gp_clamp( n0 - A.e - 3, glast );
	const int gb = gp_clamp( n1 - 1 - A.e + 3, glast );
	const int gbv = n1 - 1 - A.e + 3;
	const int u_a = a_start[ ga ];
	const int q_a = u_a >> 2; // blocks of four samples, absolute coordinates

	// ---- the sample feed. A lane loads EIGHT samples of its row at a time
	// (2 * NDW dwords, one group ahead: `nxt` is in flight while `cur` is
	// used); the four samples up next always sit in cur[ 0 .. NDW ), so their
	// positions are compile-time: after a block the upper half moves down or
	// the next group takes over.
	constexpr int NDW = CH * (int) sizeof( Tin ); // dwords per four samples
	constexpr int EPD = 4 / (int) sizeof( Tin );  // elements per dword
	// (vector-typed so that a half group is one register tuple from the load
	// on: with single dwords the loop-carried copies of the group in flight
	// were made right after its load went out -- at full memory latency)
	constexpr int NV = ( NDW + 3 ) / 4;
	constexpr int VW = NDW / NV;
	static_assert( NV * VW == NDW, "a half group is NV equal register tuples" );
	typedef typename SA2Vec< VW > :: T HV;
	struct Half { HV v[ NV ]; };
	Half curA, curB, nxtA, nxtB; // cur: A = the block up next; nxt: in flight
	const unsigned char* const lrow = (const unsigned char*) P.src +
		(long) lc * P.s_lane;

	auto g_load = [&]( const int q8 ) // unconditional: a group on the image's
	{                                 // edge reads a clamped address, g_fix
		const int gs = max( 0, min( q8 * 8, slast + 1 - 8 )); // replaces it
		const unsigned char* const gp = lrow + (long) gs * CH * sizeof( Tin );
		sa2_gload< VW, 0 >( nxtA.v[ 0 ], gp );
		sa2_gload< VW, NDW * 4 >( nxtB.v[ 0 ], gp );

		if constexpr( NV >= 2 )
		{
			sa2_gload< VW, VW * 4 >( nxtA.v[ NV >= 2 ? 1 : 0 ], gp );
			sa2_gload< VW, ( NDW + VW ) * 4 >( nxtB.v[ NV >= 2 ? 1 : 0 ], gp );
		}

		if constexpr( NV >= 3 )
		{
			sa2_gload< VW, 2 * VW * 4 >( nxtA.v[ NV >= 3 ? 2 : 0 ], gp );
			sa2_gload< VW, ( NDW + 2 * VW ) * 4 >( nxtB.v[ NV >= 3 ? 2 : 0 ], gp );
		}

		if constexpr( NV >= 4 )
		{
			sa2_gload< VW, 3 * VW * 4 >( nxtA.v[ NV >= 4 ? 3 : 0 ], gp );
			sa2_gload< VW, ( NDW + 3 * VW ) * 4 >( nxtB.v[ NV >= 4 ? 3 : 0 ], gp );
		}
	};

	auto g_take = [&]() // the group in flight becomes the current one
	{
		sa2_landed( nxtA.v[ 0 ], nxtB.v[ 0 ]);
#pragma unroll
		for( int i = 1; i < NV; i++ )
		{
			sa2_landed( nxtA.v[ i ], nxtB.v[ i ]);
		}

		curA = nxtA;
		curB = nxtB;
	};

	auto h_get = [&]( const Half& h, const int k ) -> unsigned int
	{
		return( SA2Vec< VW > :: get( h.v[ k / VW ], k % VW ));
	};

	auto h_or = [&]( Half& h, const int k, const unsigned int x )
	{
		SA2Vec< VW > :: orr( h.v[ k / VW ], k % VW, x );
	};

	// A group that is not the eight samples as they lie in memory: the view
	// repeats the edge sample beyond the image; the samples before the chunk's
	// first one count as zeros (their table rows belong to earlier outputs,
	// whose slots must stay clean -- 0 * c changes nothing). Rare: element loads.
	auto g_fix = [&]( const int q8 )
	{
		if( q8 * 8 >= 0 && q8 * 8 + 7 <= slast && q8 * 8 >= u_a )
		{
			return;
		}

#pragma unroll
		for( int i = 0; i < NV; i++ )
		{
			curA.v[ i ] = (HV) 0u;
			curB.v[ i ] = (HV) 0u;
		}

#pragma unroll
		for( int j = 0; j < 8; j++ )
		{
			const int u = q8 * 8 + j;
			const Tin* const px = (const Tin*) lrow + (long) gp_clamp( u, slast ) *
				CH;
#pragma unroll
			for( int c = 0; c < CH; c++ )
			{
				const int ei = ( j & 3 ) * CH + c;
				unsigned int v = 0u;

				if( u >= u_a && u <= A.mmax ) // (zero-stuffed view: 0 above)
				{
					v = (unsigned int) px[ c ];
				}

				if( j < 4 ) h_or( curA, ei / EPD, v << (( ei % EPD ) * ( 32 / EPD )));
				else h_or( curB, ei / EPD, v << (( ei % EPD ) * ( 32 / EPD )));
			}
		}
	};

	auto x_of = [&]( const int j ) -> PV // sample j of the block up next
	{
		float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
		for( int c = 0; c < CH; c++ )
		{
			const int ei = j * CH + c;

			v[ c ] = (float) (( h_get( curA, ei / EPD ) >> (( ei % EPD ) *
				( 32 / EPD ))) & ( SRC == 1 ? 0xffu : 0xffffu ));
		}

		f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
		return( PV::from( o ));
	};

	// ---- the accumulators, the correction FIR's window, pending outputs
	SA2Acc< NC > R;
#pragma unroll
	for( int i = 0; i < 16; i++ )
	{
		R.A[ i ] = (f2) 0.0f; R.B[ i ] = (f2) 0.0f; R.BB[ i >> 1 ] = (f2) 0.0f;
	}

	PV w[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		w[ i ] = PV::zero();
	}

	f4 pend[ 4 ]; // four consecutive outputs leave as 64 bytes per row
#pragma unroll
	for( int i = 0; i < 4; i++ )
	{
		pend[ i ] = (f4) 0.0f;
	}

	SA2State T;
	T.g = a_first[ u_a - A.u_min ]; // (<= ga)
	T.n_out = n0; T.ga = ga; T.gb = gb; T.n0 = n0; T.n1 = n1; T.e = A.e;
	T.li = li; T.lok = lok; T.dp = P.dst + (long) li * P.d_lane;
	T.f0 = A.f[ 0 ]; T.f1 = A.f[ 1 ]; T.f2 = A.f[ 2 ]; T.f3 = A.f[ 3 ];
	T.bad = 0;
	int endg = a_start[ T.g ] + nt - 1; // the last sample of output T.g

	// the first group: loaded, fixed up, the block of u_a at the bottom
#pragma unroll
	for( int i = 0; i < NV; i++ )
	{
		nxtA.v[ i ] = (HV) 0u;
		nxtB.v[ i ] = (HV) 0u;
	}

	g_load( q_a >> 1 );
	g_take();
	g_fix( q_a >> 1 );

	if( q_a & 1 )
	{
		curA = curB;
	}

	g_load(( q_a >> 1 ) + 1 );

	// ---- the table feed: the rows of TWO samples per scalar load batch, one
	// batch ahead. Scalar loads return out of order, so a wait is a wait for
	// all of them: a batch is waited for BEFORE the next one goes out (the
	// compiler would wait at the first use, after it), and nothing else in the
	// loop counts on lgkmcnt.
	gp_cfloat* const tabu = (gp_cfloat*) (unsigned long) A.tab2;
	gp_cint* ip = (gp_cint*) (unsigned long) A.ridx + ( q_a * 4 - A.tbase );

	// (the loads are asm so that the compiler's own counter bookkeeping does
	// not see them: it would add a second lgkmcnt(0) at the first use of a
	// batch -- after the next batch has gone out)
	auto rows2 = [&]( f16v& lo, f16v& hi, const int o0, const int o1 )
	{
		asm volatile( "s_load_dwordx16 %0, %2, %3\n\t"
			"s_load_dwordx16 %1, %2, %4"
			: "=&s"( lo ), "=&s"( hi ) : "s"( tabu ), "s"( o0 ), "s"( o1 )
			: "memory" );
	};

	typedef int i4v __attribute__(( ext_vector_type( 4 )));
	auto idx4 = [&]( i4v& d, gp_cint* const q )
	{
		asm volatile( "s_load_dwordx4 %0, %1, 0x0" : "=&s"( d ) : "s"( q )
			: "memory" );
	};

	// row offsets of this block's samples (ia), the next block's (ib), and the
	// one after it (ic, in flight)
	i4v ia, ib, ic;
	idx4( ia, ip );
	idx4( ib, ip + 4 );
	ic = ib;
	SA2_LGKM_WAIT();
	f16v ca0, ca1, cb0, cb1;
	rows2( ca0, ca1, ia[ 0 ], ia[ 1 ]);
	cb0 = ca0; cb1 = ca1;
	ip += 8;

	int qn = q_a;              // the next block of samples
	int ulast = q_a * 4 - 1;   // the last sample multiplied in
	const int s0 = T.g & 15;     // slot of the first output to complete
	bool started = false;

	auto step = [&]( const int j, const f16v& c )
	{
		const PV x = x_of( j );
#pragma unroll
		for( int k = 0; k < 4; k++ )
		{
			f2 c0, c1;
			c0.x = c[ k * 4 ]; c0.y = c[ k * 4 + 1 ];
			c1.x = c[ k * 4 + 2 ]; c1.y = c[ k * 4 + 3 ];
			sa2_quad< NC >( &R.A[ k * 4 ], &R.B[ k * 4 ], &R.BB[ k * 2 ], x.a,
				x.b, c0, c1 );
		}
	};

	// One output: the blocks of samples up to the one its last tap falls into,
	// then it leaves its accumulator. SC = its slot g % 16, compile-time: the
	// loop over the outputs is unrolled 16 times, the trip into it starts at
	// slot s0 (the copies before it are skipped once). Returns true after the
	// last output.
	auto body = [&]( auto SC ) -> bool
	{
		constexpr int S = decltype( SC )::value;

		if( !started )
		{
			if( S != s0 )
			{
				return( false );
			}

			started = true;
		}

		// (the next output's first sample: loaded here, used after this one left)
		const int nst = a_start[ min( T.g + 1, gb )];

#pragma clang loop unroll( disable )
		while( T.g <= gb && endg > ulast )
		{
			SA2_LGKM_WAIT();
			rows2( cb0, cb1, ia[ 2 ], ia[ 3 ]);
			idx4( ic, ip );
			step( 0, ca0 );
			step( 1, ca1 );
			SA2_LGKM_WAIT();
			rows2( ca0, ca1, ib[ 0 ], ib[ 1 ]);
			step( 2, cb0 );
			step( 3, cb1 );
			ia = ib;
			ib = ic;
			ip += 4;
			qn++;
			ulast += 4;

			if( qn & 1 )
			{
				curA = curB;
			}
			else
			{
				g_take();
				g_fix( qn >> 1 );
				g_load(( qn >> 1 ) + 1 );
			}
		}

		sa2_emit< NC, S, true >( R, w, pend, T, P );
		T.g++;

		if( T.g <= gb )
		{
			endg = nst + nt - 1;
		}

		return( T.g > gbv );
	};

	for( ; ; )
	{
		if( body( std::integral_constant< int, 0 >() )) break;
		if( body( std::integral_constant< int, 1 >() )) break;
		if( body( std::integral_constant< int, 2 >() )) break;
		if( body( std::integral_constant< int, 3 >() )) break;
		if( body( std::integral_constant< int, 4 >() )) break;
		if( body( std::integral_constant< int, 5 >() )) break;
		if( body( std::integral_constant< int, 6 >() )) break;
		if( body( std::integral_constant< int, 7 >() )) break;
		if( body( std::integral_constant< int, 8 >() )) break;
		if( body( std::integral_constant< int, 9 >() )) break;
		if( body( std::integral_constant< int, 10 >() )) break;
		if( body( std::integral_constant< int, 11 >() )) break;
		if( body( std::integral_constant< int, 12 >() )) break;
		if( body( std::integral_constant< int, 13 >() )) break;
		if( body( std::integral_constant< int, 14 >() )) break;
		if( body( std::integral_constant< int, 15 >() )) break;
	}
}

// The same for the SECOND pass of such a plan: lanes are pixel COLUMNS of the
// float RGBA intermediate (FltBuf), a sample is one of its rows. The rows
// travel HBM -> LDS by LDS-DMA (1 KiB per row and instruction, no registers),
// three blocks of four rows ahead of the block being multiplied in; a block's
// four pixels are read from LDS while the block before it runs. Rows outside
// [ulo, uhi] -- not written by this call's first pass, possibly not finite --
// are never touched: the nearest row inside stands in (its taps are zeros or
// belong to outputs of other chunks).
// (A lane per CHANNEL of a pixel column -- three times the strips, so three
// times longer chunks and less of the 36-sample run-in per chunk -- was built
// and measured: 59.9 us against 51.0 us for the README photo case. A sample
// is then only 128 cycles of arithmetic, too little to hide a scalar load
// behind, and every output costs its own trip through the correction filter
// and the output stage three times.)
#define SA2V_PF 3    // blocks of four rows in flight or landed ahead of use
#define SA2V_RB 4    // blocks in the LDS ring (a power of two > SA2V_PF)
#define SA2V_RING ( SA2V_RB * 4 ) // rows in the ring
#define SA2V_WAIT() GP_WAIT_VM( 8 ) // ( SA2V_PF - 1 ) * 4 younger operations

// HCH > 0: the FIRST pass of a plan with a FLOAT source of HCH channels (run
// optimistically, SAParams::flag_set): lanes are image ROWS, a sample is a
// pixel of the lane's row. A block of four samples is 16 * HCH contiguous
// bytes of each row = HCH 16-byte pieces: one LDS-DMA instruction fetches them
// for 16 rows (lane l: row l / HCH, piece l % HCH -- whole 64-byte runs of a
// row, not 16 bytes of 64 rows each), four instructions a block; a lane reads
// its HCH pieces back (4-way bank conflicts on 1 % of the block's time).
// Blocks that reach beyond the row are re-read sample by sample (the view's
// replicated edge). Outputs leave as in k_sacc2: four at a time, 64 bytes.
template< int NC, int HCH >
__global__ void __launch_bounds__( 64 ) k_sacc2v( const SAParams P )
{
	typedef SAV< NC > PV;
	constexpr bool HOR = ( HCH > 0 );
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const f4* const ring = (const f4*) smem;
	const SAAxis& A = P.ax;

	if( P.flag_clear != nullptr && blockIdx.x == 0 && threadIdx.x == 0 )
	{
		*P.flag_clear = 0u; // (set, if at all, by the pass after this one)
	}

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int l0 = P.lane_lo + strip * 64;
	const int li = l0 + lane;
	const bool lok = ( li < P.lane_hi );
	const int lc = min( li, P.lane_hi - 1 );
	const int n0 = P.out_lo + chunk * P.chunk;
	const int n1 = min( n0 + P.chunk, P.out_hi );

	if( n0 >= n1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	gp_cint* const a_first = (gp_cint*) (unsigned long) A.first;
	const int nt = A.nt;
	const int glast = A.g_len - 1;
	const int slast = A.in_len - 1;
	const int ga = gp_clamp( n0 - A.e - 3, glast );
	const int gb = gp_clamp( n1 - 1 - A.e + 3, glast );
	const int gbv = n1 - 1 - A.e + 3;
	const int u_a = a_start[ ga ];
	const int u_e = a_start[ gb ] + nt - 1;
	const int ulo = max( u_a, 0 );
	const int uhi = min( u_e, slast );
	const int q_a = u_a >> 2;

	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) smem;
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int voff = lc * (int) P.s_lane;
	const unsigned sstep = (unsigned) P.s_step;

	// Row lanes fetch SUPERBLOCKS of eight samples: 32 * HC contiguous bytes of
	// each row = PPR 16-byte pieces, one DMA instruction for RPI rows (lane l:
	// row l / PPR, piece l % PPR) -- whole 128-byte lines of a row per request
	// (float RGBA); as blocks of four samples (64 bytes of 16 rows) the pass ran
	// at 2.1 TB/s with 45 % of its wave-cycles in s_waitcnt. Two superblocks
	// in the ring: the one being multiplied in and the next one in flight.
	constexpr int HC = ( HOR ? HCH : 1 );
	constexpr int PPR = 2 * HC;         // pieces per row and superblock
	constexpr int RPI = 64 / PPR;       // rows per instruction
	constexpr int NI = ( 64 + RPI - 1 ) / RPI; // instructions per superblock
	static_assert( !HOR || 2 * NI <= SA2V_RING, "two superblocks fit the ring" );
	int hvoff[ HOR ? NI : 1 ];

	if( HOR )
	{
		const int hr = min( lane / PPR, RPI - 1 );
		const int hp = lane - ( lane / PPR ) * PPR;
#pragma unroll
		for( int i = 0; i < ( HOR ? NI : 1 ); i++ )
		{
			hvoff[ i ] = min( l0 + RPI * i + hr, P.lane_hi - 1 ) * (int) P.s_lane +
				hp * 16;
		}
	}

	auto d_issue_sb = [&]( const int sb ) // the eight samples of superblock sb
	{
		// its first sample, inside the row (a superblock that reaches beyond
		// the row is re-read pixel by pixel when it is used)
		const int us = min( max( sb * 8, 0 ), slast - 7 );
#pragma unroll
		for( int i = 0; i < ( HOR ? NI : 1 ); i++ )
		{
			df_dma( rsrc, lds0 + (unsigned) ((( sb & 1 ) * NI + i ) * 1024 ),
				hvoff[ i ], (unsigned) ( us * HC * 4 ));
		}
	};

	auto d_issue = [&]( const int qb ) // column lanes: the four rows of block qb
	{
#pragma unroll
		for( int j = 0; j < 4; j++ )
		{
			const int uu = min( max( qb * 4 + j, ulo ), uhi );
			df_dma( rsrc, lds0 + (unsigned) ((( qb & ( SA2V_RB - 1 )) * 4 + j ) *
				1024 ), voff, (unsigned) uu * sstep );
		}
	};

	PV x[ 4 ];
	auto x_read = [&]( const int qb )
	{
		if( HOR )
		{
			// HC pieces = 4 * HC floats = four pixels of HC channels
			float fl[ 4 * HC ];
			const int sb = qb >> 1;
			const f4* const rp = ring + (( sb & 1 ) * NI + lane / RPI ) * 64 +
				( lane - ( lane / RPI ) * RPI ) * PPR + ( qb & 1 ) * HC;
#pragma unroll
			for( int k = 0; k < HC; k++ )
			{
				const f4 t = rp[ k ];
				fl[ 4 * k ] = t.x; fl[ 4 * k + 1 ] = t.y;
				fl[ 4 * k + 2 ] = t.z; fl[ 4 * k + 3 ] = t.w;
			}

			if( sb * 8 < 0 || sb * 8 + 7 > slast )
			{
				// (the view repeats the edge pixels beyond the row; the
				// superblock was fetched from a shifted position: rare)
				const float* const row = (const float*) ( (const char*) P.src +
					(long) lc * P.s_lane );
#pragma unroll
				for( int j = 0; j < 4; j++ )
				{
					const float* const px = row + (long) gp_clamp( qb * 4 + j,
						slast ) * HC;
#pragma unroll
					for( int c = 0; c < HC; c++ )
					{
						fl[ j * HC + c ] = ( qb * 4 + j > A.mmax ? 0.0f : px[ c ]);
					}
				}
			}
#pragma unroll
			for( int j = 0; j < 4; j++ )
			{
				f4 o;
				o.x = fl[ j * HC ];
				o.y = ( HC > 1 ? fl[ j * HC + ( HC > 1 ? 1 : 0 )] : 0.0f );
				o.z = ( HC > 2 ? fl[ j * HC + ( HC > 2 ? 2 : 0 )] : 0.0f );
				o.w = ( HC > 3 ? fl[ j * HC + ( HC > 3 ? 3 : 0 )] : 0.0f );
				x[ j ] = PV::from( o );
			}
		}
		else
		{
#pragma unroll
		for( int j = 0; j < 4; j++ )
		{
			x[ j ] = PV::from( ring[ (( qb & ( SA2V_RB - 1 )) * 4 + j ) * 64 +
				lane ]);
		}
		}

		if( qb == q_a )
		{
			// the samples before the chunk's first one count as zeros (their
			// table rows belong to earlier outputs, whose slots stay clean)
#pragma unroll
			for( int j = 0; j < 3; j++ )
			{
				if( qb * 4 + j < u_a )
				{
					x[ j ] = PV::zero();
				}
			}
		}

		if( !HOR && qb * 4 + 3 > A.mmax )
		{
			// (column lanes: a zero-stuffed view reads 0 above its last
			// replicated row; row lanes did it in their re-read above)
#pragma unroll
			for( int j = 0; j < 4; j++ )
			{
				if( qb * 4 + j > A.mmax )
				{
					x[ j ] = PV::zero();
				}
			}
		}
	};

	SA2Acc< NC > R;
#pragma unroll
	for( int i = 0; i < 16; i++ )
	{
		R.A[ i ] = (f2) 0.0f; R.B[ i ] = (f2) 0.0f; R.BB[ i >> 1 ] = (f2) 0.0f;
	}

	PV w[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		w[ i ] = PV::zero();
	}

	f4 pend[ 4 ]; // (row lanes only)
#pragma unroll
	for( int i = 0; i < 4; i++ )
	{
		pend[ i ] = (f4) 0.0f;
	}

	SA2State T;
	T.g = a_first[ u_a - A.u_min ]; // (<= ga)
	T.n_out = n0; T.ga = ga; T.gb = gb; T.n0 = n0; T.n1 = n1; T.e = A.e;
	T.li = li; T.lok = lok;
	T.dp = ( HOR ? P.dst + (long) li * P.d_lane :
		P.dst + (long) li * P.d_lane + (long) n0 * P.d_step );
	T.f0 = A.f[ 0 ]; T.f1 = A.f[ 1 ]; T.f2 = A.f[ 2 ]; T.f3 = A.f[ 3 ];
	T.bad = 0;
	int endg = a_start[ T.g ] + nt - 1; // the last sample of output T.g

	if( HOR )
	{
		d_issue_sb( q_a >> 1 );
		GP_WAIT_VM( 0 );
	}
	else
	{
#pragma unroll
		for( int i = 0; i < SA2V_PF; i++ )
		{
			d_issue( q_a + i );
		}

		SA2V_WAIT(); // loads retire in order: block q_a has landed
	}

	x_read( q_a );

	if( HOR )
	{
		d_issue_sb(( q_a >> 1 ) + 1 ); // (into the other half of the ring)
	}

	// ---- the table feed (see k_sacc2)
	gp_cfloat* const tabu = (gp_cfloat*) (unsigned long) A.tab2;
	gp_cint* ip = (gp_cint*) (unsigned long) A.ridx + ( q_a * 4 - A.tbase );

	auto rows2 = [&]( f16v& lo, f16v& hi, const int o0, const int o1 )
	{
		asm volatile( "s_load_dwordx16 %0, %2, %3\n\t"
			"s_load_dwordx16 %1, %2, %4"
			: "=&s"( lo ), "=&s"( hi ) : "s"( tabu ), "s"( o0 ), "s"( o1 )
			: "memory" );
	};

	typedef int i4v __attribute__(( ext_vector_type( 4 )));
	auto idx4 = [&]( i4v& d, gp_cint* const q )
	{
		asm volatile( "s_load_dwordx4 %0, %1, 0x0" : "=&s"( d ) : "s"( q )
			: "memory" );
	};

	// row offsets of this block's samples (ia), the next block's (ib), and the
	// one after it (ic, in flight)
	i4v ia, ib, ic;
	idx4( ia, ip );
	idx4( ib, ip + 4 );
	ic = ib;
	SA2_LGKM_WAIT();
	f16v ca0, ca1, cb0, cb1;
	rows2( ca0, ca1, ia[ 0 ], ia[ 1 ]);
	cb0 = ca0; cb1 = ca1;
	ip += 8;

	int qn = q_a;
	int ulast = q_a * 4 - 1;
	const int s0 = T.g & 15;
	bool started = false;

	auto step = [&]( const int j, const f16v& c )
	{
#pragma unroll
		for( int k = 0; k < 4; k++ )
		{
			f2 c0, c1;
			c0.x = c[ k * 4 ]; c0.y = c[ k * 4 + 1 ];
			c1.x = c[ k * 4 + 2 ]; c1.y = c[ k * 4 + 3 ];
			sa2_quad< NC >( &R.A[ k * 4 ], &R.B[ k * 4 ], &R.BB[ k * 2 ],
				x[ j ].a, x[ j ].b, c0, c1 );
		}
	};

	auto body = [&]( auto SC ) -> bool
	{
		constexpr int S = decltype( SC )::value;

		if( !started )
		{
			if( S != s0 )
			{
				return( false );
			}

			started = true;
		}

		// (the next output's first sample: loaded here, used after this one left)
		const int nst = a_start[ min( T.g + 1, gb )];

#pragma clang loop unroll( disable )
		while( T.g <= gb && endg > ulast )
		{
			SA2_LGKM_WAIT();
			// (the block's pixels, read from LDS a block ago: used HERE as far
			// as the compiler is concerned, so that its own wait for them
			// comes before the next scalar loads go out, not after)
			asm volatile( "" : "+v"( x[ 0 ].a ), "+v"( x[ 1 ].a ), "+v"( x[ 2 ].a ),
				"+v"( x[ 3 ].a ), "+v"( x[ 0 ].b ), "+v"( x[ 1 ].b ),
				"+v"( x[ 2 ].b ), "+v"( x[ 3 ].b ));
			rows2( cb0, cb1, ia[ 2 ], ia[ 3 ]);
			idx4( ic, ip );
			step( 0, ca0 );
			step( 1, ca1 );
			SA2_LGKM_WAIT();
			rows2( ca0, ca1, ib[ 0 ], ib[ 1 ]);
			step( 2, cb0 );
			step( 3, cb1 );
			ia = ib;
			ib = ic;
			ip += 4;
			qn++;
			ulast += 4;
			// the next block goes out into slots read long ago; with only
			// the operations of the SA2V_PF - 1 blocks after qn (or fewer:
			// stores) outstanding, block qn has landed
			if( HOR )
			{
				if(( qn & 1 ) == 0 )
				{
					// a new superblock: its pieces went out two blocks ago, its
					// half of the ring was read out before that; the one after
					// it goes out into the half just used up
					GP_WAIT_VM( 0 );
					x_read( qn );
					d_issue_sb(( qn >> 1 ) + 1 );
				}
				else
				{
					x_read( qn );
				}
			}
			else
			{
				d_issue( qn + SA2V_PF - 1 );
				SA2V_WAIT();
				x_read( qn );
			}
		}

		sa2_emit< NC, S, HOR >( R, w, pend, T, P );
		T.g++;

		if( T.g <= gb )
		{
			endg = nst + nt - 1;
		}

		return( T.g > gbv );
	};

	for( ; ; )
	{
		if( body( std::integral_constant< int, 0 >() )) break;
		if( body( std::integral_constant< int, 1 >() )) break;
		if( body( std::integral_constant< int, 2 >() )) break;
		if( body( std::integral_constant< int, 3 >() )) break;
		if( body( std::integral_constant< int, 4 >() )) break;
		if( body( std::integral_constant< int, 5 >() )) break;
		if( body( std::integral_constant< int, 6 >() )) break;
		if( body( std::integral_constant< int, 7 >() )) break;
		if( body( std::integral_constant< int, 8 >() )) break;
		if( body( std::integral_constant< int, 9 >() )) break;
		if( body( std::integral_constant< int, 10 >() )) break;
		if( body( std::integral_constant< int, 11 >() )) break;
		if( body( std::integral_constant< int, 12 >() )) break;
		if( body( std::integral_constant< int, 13 >() )) break;
		if( body( std::integral_constant< int, 14 >() )) break;
		if( body( std::integral_constant< int, 15 >() )) break;
	}

	if( P.flag_set != nullptr && T.bad != 0 && lane == 0 )
	{
		atomicOr( P.flag_set, 1u );
	}

	GP_WAIT_VM( 0 ); // no LDS-DMA may outlive the workgroup's LDS
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct SAData
{
	SAAxis a;
	double k; // average input advance per output
};

static int sa_upload( std::vector< void* >& allocs, const void* h, size_t bytes,
	void** out )
{
	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, bytes ));
	allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, h, bytes, hipMemcpyHostToDevice ));
	*out = q;
	return( AVIRHIP_OK );
}

// Builds the by-sample tables of a recognised axis; *out stays nullptr if the
// axis is not a streaming-accumulation case (more than 16 outputs alive at a
// sample -- upsizing --, zero-stuffed views, LANCIR's split sums).
int sacc_prepare_axis( const GPAxisHost& H, std::vector< void* >& allocs,
	SAData** out )
{
	*out = nullptr;
	const GPAxis& A = H.a;

	if( !H.ok || A.mode != GP_POST || A.lanc || A.g_len < 1 ||
		A.view_mmax < A.in_len - 1 || getenv( "AVIRHIP_NO_SACC" ) != nullptr ||
		( A.view_mmax != 0x7fffffff && getenv( "AVIRHIP_NO_SACC_ZS" ) != nullptr ))
	{
		return( AVIRHIP_OK );
	}

	const std::vector< int >& st = H.start;
	const int nt = A.nt;
	const int u_min = st[ 0 ];
	const int u_max = st[ A.g_len - 1 ] + nt - 1;
	const int rows = u_max - u_min + 1;
	const int pad = 96; // table groups are fetched ahead, `first` 64 at a time
	std::vector< int > meta( (size_t) ( rows + pad ), 0 );
	std::vector< float > tab( (size_t) ( rows + pad ) * 16, 0.0f );
	int first = 0, last = 0;

	for( int r = 0; r < rows; r++ )
	{
		const int u = u_min + r;

		while( first < A.g_len - 1 && st[ first ] + nt - 1 < u )
		{
			first++;
		}

		while( last < A.g_len - 1 && st[ last + 1 ] <= u )
		{
			last++;
		}

		if( last - first + 1 > SA_NS - 1 || last < first )
		{
			return( AVIRHIP_OK );
		}

		meta[ (size_t) r ] = first;
		const int cnt = last - first + 1;
		memcpy( &tab[ (size_t) r * 16 + 15 ], &cnt, sizeof( cnt ));

		for( int g = first; g <= last; g++ )
		{
			const float* const cr = H.h_coef + (size_t) ( H.h_fidx != nullptr ?
				H.h_fidx[ g ] : g ) * nt;

			tab[ (size_t) r * 16 + ( g - first )] = cr[ u - st[ g ]];
		}
	}

	for( int r = rows; r < rows + pad; r++ )
	{
		meta[ (size_t) r ] = A.g_len - 1;
	}

	SAData* D = new SAData();
	memset( &D -> a, 0, sizeof( D -> a ));
	D -> a.nt = nt; D -> a.in_len = A.in_len; D -> a.g_len = A.g_len;
	D -> a.out_len = A.out_len; D -> a.e = A.e; D -> a.lanc = 0;
	memcpy( D -> a.f, A.f, sizeof( A.f ));
	D -> a.u_min = u_min;
	D -> a.mmax = A.view_mmax;
	D -> a.start = A.start;
	D -> k = (double) rows / A.g_len;

	void* q = nullptr;
	int rc = sa_upload( allocs, meta.data(), meta.size() * sizeof( int ), &q );
	D -> a.first = (const int*) q;

	if( rc == 0 )
	{
		rc = sa_upload( allocs, tab.data(), tab.size() * sizeof( float ), &q );
		D -> a.tab = (const float*) q;
	}

	if( rc != 0 )
	{
		delete D;
		return( rc );
	}

	// k_sacc2's table: one coefficient per accumulator slot. Usable when an
	// output can leave its slot at the end of the 4-sample group its last tap
	// falls into, i.e. the slot's next output (16 later) starts after it.
	{
		const int tbase = (int) ( floor( u_min / 4.0 ) * 4.0 );
		const int rows2 = u_max - tbase + 1;
		bool ok2 = ( getenv( "AVIRHIP_NO_SACC2" ) == nullptr );

		for( int g = 0; ok2 && g + 1 < A.g_len; g++ )
		{
			ok2 = ( st[ g + 1 ] > st[ g ]);
		}

		for( int g = 0; ok2 && g + 16 < A.g_len; g++ )
		{
			const int e = st[ g ] + nt - 1;
			const int gend = (int) ( floor( e / 4.0 ) * 4.0 ) + 3;
			ok2 = ( st[ g + 16 ] > gend );
		}

		if( ok2 )
		{
			std::vector< float > t2( (size_t) ( rows2 + 16 ) * 16, 0.0f );
			int fi = 0, la = 0;

			for( int r = 0; r < rows; r++ )
			{
				const int u = u_min + r;

				while( fi < A.g_len - 1 && st[ fi ] + nt - 1 < u ) fi++;
				while( la < A.g_len - 1 && st[ la + 1 ] <= u ) la++;

				for( int g = fi; g <= la; g++ )
				{
					const float* const cr = H.h_coef + (size_t) (
						H.h_fidx != nullptr ? H.h_fidx[ g ] : g ) * nt;

					t2[ (size_t) ( u - tbase ) * 16 + ( g & 15 )] = cr[ u - st[ g ]];
				}
			}

			// equal rows once (bit patterns), samples address them by offset
			const size_t nr = (size_t) rows2 + 16;
			std::vector< float > tu;
			std::vector< int > ri( nr + 16, 0 );
			std::map< std::array< uint32_t, 16 >, int > seen;

			for( size_t r = 0; r < nr; r++ )
			{
				std::array< uint32_t, 16 > key;
				memcpy( key.data(), &t2[ r * 16 ], 64 );
				auto it = seen.find( key );

				if( it == seen.end() )
				{
					it = seen.insert( std::make_pair( key,
						(int) ( tu.size() * sizeof( float )))).first;

					tu.insert( tu.end(), &t2[ r * 16 ], &t2[ r * 16 ] + 16 );
				}

				ri[ r ] = it -> second;
			}

			for( size_t r = nr; r < ri.size(); r++ )
			{
				ri[ r ] = ri[ nr - 1 ]; // (an all-zero row)
			}

			if( getenv( "AVIRHIP_SA2_SAMEROW" ) != nullptr ) // timing experiment
			{
				for( size_t r = 0; r < ri.size(); r++ )
				{
					ri[ r ] = ( atoi( getenv( "AVIRHIP_SA2_SAMEROW" )) == 2 ?
						(int) ( r % 8 ) * 64 : 0 );
				}
			}

			rc = sa_upload( allocs, tu.data(), tu.size() * sizeof( float ), &q );

			if( rc == 0 )
			{
				D -> a.tab2 = (const float*) q;
				rc = sa_upload( allocs, ri.data(), ri.size() * sizeof( int ), &q );
			}

			if( rc != 0 )
			{
				delete D;
				return( rc );
			}

			D -> a.ridx = (const int*) q;
			D -> a.tbase = tbase;
		}
	}

	*out = D;
	return( AVIRHIP_OK );
}

void sacc_release_axis( SAData* D )
{
	delete D;
}

// whether the axis has the per-slot table of the branch-free kernels
bool sacc_has_two( const SAData* D )
{
	return( D != nullptr && D -> a.tab2 != nullptr );
}

// the axis reads a zero-stuffed view (1 < k < 2: a 2x upsample folded into
// the resize); its average input advance per output
bool sacc_is_zs( const SAData* D )
{
	return( D != nullptr && D -> a.mmax != 0x7fffffff );
}

double sacc_k( const SAData* D )
{
	return( D != nullptr ? D -> k : 0.0 );
}

template< int SRC, int CH, int NC >
static void sa2_launch( const SAParams& P, int items, hipStream_t st )
{
	hipLaunchKernelGGL(( k_sacc2< SRC, CH, NC > ), dim3( items ), dim3( 64 ), 0,
		st, P );
}

template< int SRC >
static void sa2_launch1( const SAParams& P, int ch, int items, hipStream_t st )
{
	switch( ch )
	{
		case 1: sa2_launch< SRC, 1, 1 >( P, items, st ); break;
		case 2: sa2_launch< SRC, 2, 2 >( P, items, st ); break;
		case 3: sa2_launch< SRC, 3, 3 >( P, items, st ); break;
		default: sa2_launch< SRC, 4, 4 >( P, items, st ); break;
	}
}

template< int SRC, int CH, bool HOR, int NC >
static void sa_launch2( const SAParams& P, int items, size_t lds,
	hipStream_t st )
{
	hipLaunchKernelGGL(( k_sacc< SRC, CH, true, HOR, NC > ), dim3( items ),
		dim3( 64 ), lds, st, P );
}

// typed sources: the channels of the image are the live ones
template< int SRC >
static void sa_launch1( const SAParams& P, int ch, int items, size_t lds,
	hipStream_t st )
{
	switch( ch )
	{
		case 1: sa_launch2< SRC, 1, false, 1 >( P, items, lds, st ); break;
		case 2: sa_launch2< SRC, 2, false, 2 >( P, items, lds, st ); break;
		case 3: sa_launch2< SRC, 3, false, 3 >( P, items, lds, st ); break;
		default: sa_launch2< SRC, 4, false, 4 >( P, items, lds, st ); break;
	}
}

// One pass. src_type / src_ch: AVIRHIP_F32 with 4 channels goes by LDS-DMA
// (16-byte aligned base and strides required: returns 1 otherwise); U8, U16
// and narrower floats are read as they are. Strides in bytes (source) and
// floats (destination); `rows`: the lanes are image rows (horizontal pass) --
// integer sources only come so; `live_ch`: channels of the float RGBA source
// that carry data (the others are zero padding and come out as zeros).
int sacc_run_axis( const SAData* D, bool rows, const void* src, int src_type,
	int src_ch, int live_ch, long s_lane, long s_step, float* dst, long d_lane,
	long d_step, int lane_lo, int lane_hi, int out_lo, int out_hi,
	hipStream_t st, const GPOut* out, bool finite, bool ladder,
	bool must_two, unsigned int* flag_set, const unsigned int* guard )
{
	if( lane_hi <= lane_lo || out_hi <= out_lo )
	{
		return( AVIRHIP_OK );
	}

	SAParams P;
	P.src = src; P.s_lane = s_lane; P.s_step = s_step;
	P.dst = dst; P.d_lane = d_lane; P.d_step = d_step;
	P.lane_lo = lane_lo; P.lane_hi = lane_hi;
	P.out_lo = out_lo; P.out_hi = out_hi;
	P.ax = D -> a;
	memset( &P.out, 0, sizeof( P.out ));

	if( out != nullptr )
	{
		if( rows || lane_lo != 0 )
		{
			return( 1 );
		}

		P.out = *out;
	}

	const bool dma = ( src_type == AVIRHIP_F32 && src_ch == 4 );

	if( dma && (( (uintptr_t) src & 15 ) || ( s_lane & 15 ) || ( s_step & 15 )))
	{
		return( 1 );
	}

	if(( out == nullptr && ( (uintptr_t) dst & 15 )) || ( d_lane & 3 ) ||
		( d_step & 3 ) ||
		( rows && d_step != 4 ))
	{
		return( 1 );
	}

	const int nout = out_hi - out_lo;

	// Outputs per work item: a chunk's only overhead is feeding the NT
	// samples before its first output completes (they are multiplied into
	// this chunk's outputs only: no arithmetic is repeated). The kernel is
	// ALU-bound: whole rounds of the 1024 SIMDs, two or more waves on each.
	// k_sacc2: sources known to be finite -- the caller's integer image (row
	// lanes) or the FltBuf made from one (column lanes)
	// (float sources come here optimistically, see SAParams::flag_set: row
	// lanes read pixels of src_ch floats as they lie, s_step == src_ch * 4)
	const bool two = ( finite && !ladder && D -> a.tab2 != nullptr &&
		P.ax.in_len >= 8 && (( rows && out == nullptr &&
		(( !dma && ( src_type == AVIRHIP_U8 || src_type == AVIRHIP_U16 )) ||
		( src_type == AVIRHIP_F32 && s_step == (long) src_ch * 4 &&
		( (uintptr_t) src & 3 ) == 0 && ( s_lane & 3 ) == 0 &&
		(double) s_lane * lane_hi < 2147483648.0 ))) ||
		( !rows && dma && s_lane == 16 &&
		(double) s_step * P.ax.in_len < 2147483648.0 )));

	if( must_two && !two )
	{
		return( 1 );
	}

	P.flag_set = ( two && !rows ? flag_set : nullptr );
	P.flag_clear = ( two && rows ? flag_set : nullptr );
	P.guard = ( two ? nullptr : guard );

	const size_t lds = ( two ? ( rows && src_type != AVIRHIP_F32 ? 0 :
		SA2V_RING * 1024 ) :
		(size_t) ( dma ? 3 * 4096 : 8 * 1024 ) + SA_TQ * 256 + 64 +
		( rows ? 4096 : 0 ));

	const int lanes = P.lane_hi - P.lane_lo;
	P.nstrips = ( lanes + 63 ) / 64;

	const double cap = 256.0 * std::max( 1, std::min( 16,
		(int) ( 160 * 1024 / ( lds + 512 ))));

	const double warm = std::max( 1.0, P.ax.nt / std::max( 1.0, D -> k ) *
		0.35 ) + 2.0;

	double best = -1.0;
	int best_chunk = nout;

	for( int nch = 1; nch <= nout; nch++ )
	{
		const int chunk = ( nout + nch - 1 ) / nch;

		if( chunk < 4 && nch > 1 )
		{
			break;
		}

		const double items = (double) (( nout + chunk - 1 ) / chunk ) *
			P.nstrips;

		const double res = ceil( items / cap );
		const double per = ceil( items / res / 1024.0 );
		const double cost = res * per * ( chunk + warm ) *
			( per < 2.0 ? 1.26 : 1.0 );

		if( best < 0.0 || cost < best )
		{
			best = cost;
			best_chunk = chunk;
		}
	}

	P.chunk = best_chunk;

	if( getenv( "AVIRHIP_SA_CHUNK" ) != nullptr ) // tuning aid
	{
		P.chunk = std::max( 1, atoi( getenv( "AVIRHIP_SA_CHUNK" )));
	}

	if( !rows && getenv( "AVIRHIP_SA_CHUNK_V" ) != nullptr )
	{
		P.chunk = std::max( 1, atoi( getenv( "AVIRHIP_SA_CHUNK_V" )));
	}

	int items = P.nstrips * (( nout + P.chunk - 1 ) / P.chunk );
	P.nitems = items;
	// the branch-free kernel of float row lanes (k_sacc2v< NC, CH >)
	const bool two_hf = ( two && rows && src_type == AVIRHIP_F32 );

	if( !dma && ( !rows || P.ax.in_len < 4 ))
	{
		return( 1 );
	}

	if( two )
	{
		if( !rows )
		{
			if( live_ch == 3 )
			{
				hipLaunchKernelGGL(( k_sacc2v< 3, 0 > ), dim3( items ), dim3( 64 ),
					lds, st, P );
			}
			else
			{
				hipLaunchKernelGGL(( k_sacc2v< 4, 0 > ), dim3( items ), dim3( 64 ),
					lds, st, P );
			}
		}
		else
		if( two_hf )
		{
			// float pixels of src_ch channels, lanes = image rows
#define SA2H( NC, CH ) hipLaunchKernelGGL(( k_sacc2v< NC, CH > ), dim3( items ), \
				dim3( 64 ), lds, st, P )
			switch( src_ch )
			{
				case 1: SA2H( 1, 1 ); break;
				case 2: SA2H( 2, 2 ); break;
				case 3: SA2H( 3, 3 ); break;
				default: if( live_ch == 3 ) SA2H( 3, 4 ); else SA2H( 4, 4 ); break;
			}
#undef SA2H
		}
		else
		if( src_type == AVIRHIP_U8 )
		{
			sa2_launch1< 1 >( P, src_ch, items, st );
		}
		else
		{
			sa2_launch1< 2 >( P, src_ch, items, st );
		}

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( P.guard != nullptr )
	{
		items = std::min( items, 2048 ); // (workgroups: they walk P.nitems)
	}

	if( dma )
	{
		if( rows )
		{
			if( live_ch == 3 ) sa_launch2< 0, 4, true, 3 >( P, items, lds, st );
			else sa_launch2< 0, 4, true, 4 >( P, items, lds, st );
		}
		else
		{
			if( live_ch == 3 ) sa_launch2< 0, 4, false, 3 >( P, items, lds, st );
			else sa_launch2< 0, 4, false, 4 >( P, items, lds, st );
		}
	}
	else
	if( src_type == AVIRHIP_U8 )
	{
		sa_launch1< 1 >( P, src_ch, items, lds, st );
	}
	else
	if( src_type == AVIRHIP_U16 )
	{
		sa_launch1< 2 >( P, src_ch, items, lds, st );
	}
	else
	if( src_type == AVIRHIP_F32 )
	{
		sa_launch1< 3 >( P, src_ch, items, lds, st );
	}
	else
	{
		return( 1 );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
