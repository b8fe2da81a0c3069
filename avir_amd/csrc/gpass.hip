// gpass.hip -- pass kernels for general (non-2x) ratios: ONE AXIS PER LAUNCH,
// autonomous wavefronts, every dynamically indexed window in LDS rings.
//
// A lowered axis (plan.h) of the common plans is one of
//   PRE   FIR7 then GATHER     upsizing: correction filter, then the 12-tap
//                              interpolation over the zero-stuffed view
//                              (doFilter 3748 + doUpsample 3260 + doResize2 4114)
//   POST  GATHER then FIR7     downsizing: NT-tap resize, then the correction
//                              filter (doResize 3884 / doResize2 + doFilter)
//   NONE  GATHER               LANCIR's N-tap dot products (lancir.h:2466-2544)
// with a position table (first input index per output) and one coefficient row
// per output (AVIR) or per fractional phase (LANCIR). The tiled interpreter of
// fused.hip runs any chain but keeps whole 2-D tiles in LDS: two workgroups
// per CU, phases separated by barriers, half of its wave-cycles parked
// (profiles/r02_tile). Here a wavefront owns a strip and never synchronises
// with another wave:
//
//   k_gv  vertical pass. Lane = whole pixel of a 64-pixel strip (half pixel /
//         32-pixel strips for images narrower than 49 pixels): a row is one
//         contiguous 1-KiB segment. Source rows arrive by LDS-DMA in a landing
//         queue, read a step ahead of their use; the FIR7 window and -- for
//         the tap counts that occur -- the gather window live in registers
//         (the step loop is unrolled to the window length, ring slots are
//         compile-time); coefficient rows are staged in LDS per chunk and read
//         as broadcasts one output ahead.
//   k_gh  horizontal pass. Lane = whole output pixel (16 B) of a strip of up
//         to 64 outputs; its coefficient row stays in registers while the wave
//         walks down its rows; the row's source segment arrives by LDS-DMA
//         three rows ahead.
//   k_sacc (sacc.hip) takes the downsizing axes (k >= 2) of integer / narrow
//         sources: streaming accumulation instead of a gather window.
//
// Arithmetic contract (bit-exact): mul and add separate (-ffp-contract=off),
// AVIR sums start from +0.0f and take their taps in ascending order, the FIR
// is f0*c + sum f_i*(right_i + left_i); LANCIR accumulates even and odd taps
// separately from bare first products and adds the two sums at the end.

#include "gpass_dev.h"
#include <algorithm>
#include <math.h>
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace avirhip {

// ---------------------------------------------------------------------
// horizontal pass
// ---------------------------------------------------------------------

#ifndef GH_NB
#define GH_NB 4 // source row buffers: rows r .. r + 3 are in LDS or in flight
#endif

// SRC: 0 float RGBA by LDS-DMA; 1 uint8, 2 uint16, 3 float pixels of raw_ch
// channels, read as they are (packScanline's cast and zero padding,
// avir.h:2777-2840, in the loader: no float copy of the source): a lane loads
// its pixel of each 64-pixel piece a row ahead, converts and writes it into
// the row buffer after the row in hand has been computed.
template< int SRC > struct GHElem { typedef float T; };
template<> struct GHElem< 1 > { typedef unsigned char T; };
template<> struct GHElem< 2 > { typedef unsigned short T; };

// (LANCIR plans have float RGBA sources only: there the fourth parameter is the
// compile-time kind of the output stage, gp_store_lancir_row)
template< int MODE, bool LANC, int NTC, int SX >
__global__ void __launch_bounds__( 64 ) k_gh( const GHParams P )
{
	constexpr int SRC = ( LANC ? 0 : SX );
	constexpr int OUTK = ( LANC ? SX : 0 );
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	f4* const sbuf = (f4*) smem;                    // [GH_NB][sbuf px]
	f4* const mbuf = sbuf + GH_NB * (size_t) P.sbuf; // [mbuf px]
	float* const cfl = (float*) ( mbuf + P.mbuf );  // NTC == 0: [nt][64 lanes]
	const unsigned sbuf_lds = (unsigned) (unsigned long) (lds_char*) smem;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int r0 = P.row_lo + chunk * P.chunk;
	const int r1 = min( r0 + P.chunk, P.row_hi );

	if( r0 >= r1 )
	{
		return;
	}

	const int slo = P.seg[ strip * 4 ], scnt = P.seg[ strip * 4 + 1 ];
	const int mlo = P.seg[ strip * 4 + 2 ], mcnt = P.seg[ strip * 4 + 3 ];
	const int j0 = strip * P.ow;
	const int j = j0 + lane;                        // this lane's output
	const bool jok = ( lane < P.ow && j < A.out_len );
	const int nt = ( NTC > 0 ? NTC : A.nt );
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int srow_b = (int) P.src_ss * 4;

	// per-lane source pixels of the DMA pieces (clamped: the extended view)
	const int slast = A.in_len - 1;

	// (the clamped columns once, not per row: four registers -- which the
	// any-tap-count variant, NTC == -1, does not have: 168 -> 172 VGPRs is three
	// waves per SIMD -> two)
	constexpr bool XB = ( NTC != -1 );
	int xb[ 4 ];
#pragma unroll
	for( int k = 0; k < 4; k++ )
	{
		xb[ k ] = ( XB ? gp_clamp( slo + k * 64 + lane, slast ) * 16 : 0 );
	}

	auto issue_row = [&]( const int r, const int buf )
	{
		const int rb = r * srow_b;
		const unsigned lb = sbuf_lds + ( buf * P.sbuf ) * 16;

		if( XB && P.nseg <= 4 )
		{
#pragma unroll
			for( int k = 0; k < 4; k++ )
			{
				if( k < P.nseg )
				{
					gp_dma( rsrc, lb + k * 1024, rb + xb[ k ]);
				}
			}
		}
		else
		for( int k = 0; k < P.nseg; k++ )
		{
			const int x = gp_clamp( slo + k * 64 + lane, slast );
			gp_dma( rsrc, lb + k * 1024, rb + x * 16 );
		}
	};

	// The gather this lane evaluates, its first input (relative to the
	// buffer it reads) and its coefficient row -- fixed for every row.
	//   PRE / NONE: gather output == final output j
	//   POST: lane l evaluates G[ mlo + l ] (the strip's outputs +- 3)
	const int g = ( MODE == GP_POST ? gp_clamp( mlo + lane, A.g_len - 1 ) :
		gp_clamp( j, A.g_len - 1 ));

	const bool gok = ( MODE == GP_POST ? lane < mcnt : jok );
	const int gs = A.start[ g ] - ( MODE == GP_PRE ? mlo : slo );
	// a compile-time tap count keeps the row in registers; any other count
	// parks it in LDS, tap-major (lane-consecutive: conflict-free reads)
	// (NTC == -1: any count up to 28, in registers too)
	float cf[ NTC > 0 ? NTC : ( NTC < 0 ? 28 : 1 )];
	{
		const float* cr = A.coef + (long) ( LANC ? A.fidx[ g ] : g ) * nt;

		if( NTC < 0 )
		{
#pragma unroll
			for( int t = 0; t < ( NTC < 0 ? 28 : 1 ); t++ )
			{
				cf[ t ] = ( t < nt ? cr[ t ] : 0.0f );
			}
		}
		else
		if( NTC > 0 )
		{
#pragma unroll
			for( int t = 0; t < ( NTC > 0 ? NTC : 1 ); t++ )
			{
				cf[ t ] = cr[ t ];
			}
		}
		else
		{
			for( int t = 0; t < nt; t++ )
			{
				cfl[ t * 64 + lane ] = cr[ t ];
			}
		}
	}

	// typed sources: the row in flight (up to four pieces, four channels)
	f4 rw[ SRC != 0 ? 4 : 1 ];

	auto t_load = [&]( const int r )
	{
#pragma unroll
		for( int k = 0; k < ( SRC != 0 ? 4 : 1 ); k++ )
		{
			if( k < P.nseg )
			{
				rw[ k ] = gp_load_raw( P.raw, SRC, P.raw_ch, (long) r * P.raw_ss +
					(long) gp_clamp( slo + k * 64 + lane, slast ) * P.raw_ch,
					P.raw_elems );
			}
		}
	};

	auto t_store = [&]( const int buf )
	{
#pragma unroll
		for( int k = 0; k < ( SRC != 0 ? 4 : 1 ); k++ )
		{
			if( k < P.nseg )
			{
				sbuf[ buf * P.sbuf + k * 64 + lane ] = rw[ k ];
			}
		}
	};

	if( SRC != 0 )
	{
		t_load( r0 );
		t_store( 0 );

		if( r0 + 1 < r1 )
		{
			t_load( r0 + 1 );
		}
	}
	else
	{
		for( int k = 0; k < GH_NB - 1 && r0 + k < r1; k++ )
		{
			issue_row( r0 + k, k );
		}
	}

	for( int r = r0; r < r1; r++ )
	{
		const int buf = ( r - r0 ) & ( GH_NB - 1 );
		const f4* const s = sbuf + buf * P.sbuf;
		GP_PRIO_LOW();

		if( SRC == 0 )
		{
			// row r + 3 goes out (into the buffer row r - 1 has left), then
			// wait for row r: only the pieces of the rows after it may be in
			// flight
			if( r + GH_NB - 1 < r1 )
			{
				issue_row( r + GH_NB - 1, ( buf + GH_NB - 1 ) & ( GH_NB - 1 ));
			}

			// (the count stands for LOADS only. The stores of the rows that left
			// meanwhile count on vmcnt too, but they do not retire in order with
			// the loads -- a store's acknowledgement can overtake an older load
			// (measured in sacc.hip's feed at the end of round 4: a count that
			// included "the stores since" let a group be used before it landed)
			// -- so they may only make this wait longer, never shorter. Rounds
			// 3-4 added them to the count here; no mismatch was ever seen, but
			// it was a race.)
			// (steady state: the count per segment count spelled out -- the
			// general count walks gp_wait_vm's 25-way ladder, ten branches a row)
			if( r1 - 1 - r >= GH_NB - 1 && P.nseg <= 4 )
			{
				if( P.nseg <= 2 )
				{
					if( P.nseg == 1 ) gp_wait_vm(( GH_NB - 1 ) * 1 );
					else gp_wait_vm(( GH_NB - 1 ) * 2 );
				}
				else
				{
					if( P.nseg == 3 ) gp_wait_vm(( GH_NB - 1 ) * 3 );
					else gp_wait_vm(( GH_NB - 1 ) * 4 );
				}
			}
			else
			{
				gp_wait_vm( min( GH_NB - 1, r1 - 1 - r ) * P.nseg );
			}
		}

		if( !LANC && MODE != GP_PRE && slo + scnt - 1 > A.view_mmax )
		{
			// zero-stuffed view of the source: its tail reads 0
			for( int x = lane; x < scnt; x += 64 )
			{
				if( slo + x > A.view_mmax )
				{
					((f4*) s )[ x ] = (f4) 0.0f;
				}
			}
		}

		if( MODE == GP_PRE )
		{
			// C[ mlo .. mlo + mcnt ) = FIR7 along x of the segment
			for( int i = lane; i < mcnt; i += 64 )
			{
				const int m = mlo + i;
				f4 c;

				if( m > A.view_mmax )
				{
					c = (f4) 0.0f;
				}
				else
				{
					const f4* q = s + ( gp_clamp( m, A.c_len - 1 ) - A.e - slo );
					c = A.f[ 0 ] * q[ 0 ];
					c = c + A.f[ 1 ] * ( q[ 1 ] + q[ -1 ]);
					c = c + A.f[ 2 ] * ( q[ 2 ] + q[ -2 ]);
					c = c + A.f[ 3 ] * ( q[ 3 ] + q[ -3 ]);
				}

				mbuf[ i ] = c;
			}
		}

		f4 o = (f4) 0.0f;

		if( gok )
		{
			const f4* const in = ( MODE == GP_PRE ? mbuf : s ) + gs;
			// a compile-time tap count: the whole window is read first (all
			// its ds_read_b128 in flight at once), then the arithmetic runs
			// -- left alone the compiler keeps two reads in flight and pays
			// the LDS latency six times per pixel
			f4 w[ NTC > 0 ? NTC : 1 ];

			if( NTC > 0 )
			{
#pragma unroll
				for( int t = 0; t < ( NTC > 0 ? NTC : 1 ); t++ )
				{
					w[ t ] = in[ t ];
				}

				__builtin_amdgcn_sched_barrier( 0 );
			}

			GP_PRIO_HIGH();
			o = gp_dot< LANC, f4, NTC >( nt,
				[&]( const int t ) { return( NTC > 0 ?
					w[ NTC > 0 ? t % ( NTC > 0 ? NTC : 1 ) : 0 ] : in[ t ]); },
				[&]( const int b ) -> f4
				{
					f4 c;

					if( NTC < 0 )
					{
						c.x = cf[ NTC < 0 ? ( 4 * b ) % 28 : 0 ];
						c.y = cf[ NTC < 0 ? ( 4 * b + 1 ) % 28 : 0 ];
						c.z = cf[ NTC < 0 ? ( 4 * b + 2 ) % 28 : 0 ];
						c.w = cf[ NTC < 0 ? ( 4 * b + 3 ) % 28 : 0 ];
					}
					else
					if( NTC > 0 )
					{
						c.x = cf[ ( 4 * b ) % ( NTC > 0 ? NTC : 1 )];
						c.y = cf[ ( 4 * b + 1 ) % ( NTC > 0 ? NTC : 1 )];
						c.z = cf[ ( 4 * b + 2 ) % ( NTC > 0 ? NTC : 1 )];
						c.w = cf[ ( 4 * b + 3 ) % ( NTC > 0 ? NTC : 1 )];
					}
					else
					{
						// (rows beyond nt hold whatever: never multiplied)
						c.x = cfl[ ( 4 * b ) * 64 + lane ];
						c.y = cfl[ ( 4 * b + 1 ) * 64 + lane ];
						c.z = cfl[ ( 4 * b + 2 ) * 64 + lane ];
						c.w = cfl[ ( 4 * b + 3 ) * 64 + lane ];
					}

					return( c );
				}, A.lanc, true );
		}

		if( MODE == GP_POST )
		{
			if( gok )
			{
				mbuf[ lane ] = o;
			}

			if( jok )
			{
				// out[j] = FIR7 over G[ clamp( j - e +- i ) ]
				const int glast = A.g_len - 1;
				const int c = j - A.e;
				o = A.f[ 0 ] * mbuf[ gp_clamp( c, glast ) - mlo ];
				o = o + A.f[ 1 ] * ( mbuf[ gp_clamp( c + 1, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 1, glast ) - mlo ]);
				o = o + A.f[ 2 ] * ( mbuf[ gp_clamp( c + 2, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 2, glast ) - mlo ]);
				o = o + A.f[ 3 ] * ( mbuf[ gp_clamp( c + 3, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 3, glast ) - mlo ]);
			}
		}

		if( LANC && P.lout.on != 0 )
		{
			gp_store_lancir_row< OUTK >( P.lout, r - P.dst_row0, j, o, jok );
		}
		else
		if( jok )
		{
#ifdef GP_DBG
			if( !( P.dbg & 1 ) || o.x == 1234.5f )
#endif
			*(f4*) ( P.dst + (long) ( r - P.dst_row0 ) * P.dst_ss +
				(long) j * 4 ) = o;
		}

		if( SRC != 0 && r + 1 < r1 )
		{
			// row r + 1 (loaded a row ago) into its buffer, row r + 2 out
			t_store(( buf + 1 ) & ( GH_NB - 1 ));

			if( r + 2 < r1 )
			{
				t_load( r + 2 );
			}
		}
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

// sacc.hip: the streaming-accumulation kernel of downsizing axes
struct SAData;
int sacc_prepare_axis( const GPAxisHost& H, std::vector< void* >& allocs,
	SAData** out );
void sacc_release_axis( SAData* D );
int sacc_run_axis( const SAData* D, bool rows, const void* src, int src_type,
	int src_ch, int live_ch, long s_lane, long s_step, float* dst, long d_lane,
	long d_step, int lane_lo, int lane_hi, int out_lo, int out_hi,
	hipStream_t st, const GPOut* out = nullptr, bool finite = false,
	bool ladder = false, bool must_two = false,
	unsigned int* flag_set = nullptr, const unsigned int* guard = nullptr );
bool sacc_has_two( const SAData* D );
bool sacc_is_zs( const SAData* D );
double sacc_k( const SAData* D );

// Whether a pass takes the accumulation kernel where the axis has one. Axes
// that downsize by 1 < k < 2 (13 .. 25 taps over the zero-stuffed view) have had
// one since round 4; measured against the gather kernels on one box
// (3840x2160 sources): RGB uint8 k = 1.5 0.109 vs 0.151 ms, k = 1.92 0.099 vs
// 0.141; float RGBA on the EXACT kernels k = 1.5 0.118 vs 0.122, k = 1.75 0.109
// vs 0.113, k = 1.92 0.111 vs 0.116 -- `always`: integer and float RGBA
// sources. Float pixels of 1-3 channels run optimistically (gpass_run: two
// more launches) and take it from k = 1.7 on (RGBA, optimistic: k = 1.5 0.139
// vs 0.126, k = 1.75 0.113 vs 0.115, k = 1.92 0.111 vs 0.118).
// Round 6: FLOAT RGBA sources on 1 < k < 2 axes go back to the gather kernels
// (`frgba`). k_gh's gather-then-FIR form ran its 13 .. 25 taps through the
// any-count variant -- a wave-uniform branch and an LDS round trip per block of
// four taps; with the tap count a compile-time constant (the whole window read at
// once, as the LANCIR variants have had it since round 5) the horizontal pass of
// 3840x2160 -> 2560x1440 takes 51 instead of 70 us and the plan 0.097 instead of
// 0.118 ms on the accumulation kernels (same box; -> 2000x1125 0.092 vs 0.111,
// -> 2200x1238 0.091 vs 0.107: profiles/r06_gh_post/). Integer sources keep the
// branch-free accumulation kernels, which read them raw (RGB uint8 k = 1.5:
// 0.104 vs 0.133). AVIRHIP_SA_ZS_FRGBA=1: the round-5 routing (A/B).
static bool sa_wanted( const SAData* sa, const bool always,
	const bool frgba = false )
{
	static const double kmin = ( getenv( "AVIRHIP_SA_ZS_KMIN" ) != nullptr ?
		atof( getenv( "AVIRHIP_SA_ZS_KMIN" )) : 1.7 ); // (tuning aid)
	static const bool zs_frgba = ( getenv( "AVIRHIP_SA_ZS_FRGBA" ) != nullptr &&
		atoi( getenv( "AVIRHIP_SA_ZS_FRGBA" )) != 0 );

	if( sa != nullptr && frgba && !zs_frgba && sacc_is_zs( sa ))
	{
		return( false );
	}

	return( sa != nullptr && ( always || !sacc_is_zs( sa ) ||
		sacc_k( sa ) >= kmin ));
}

struct GPData
{
	GPAxisHost h, v;     // AVIR: H pass first; LANCIR: V pass first
	SAData* sa_h;        // downsizing axes: by-sample tables (sacc.hip), or
	SAData* sa_v;        // nullptr where the gather kernels run the axis
	bool h_geom, v_geom; // the gather kernels' launch geometry exists
	int* d_hseg;
	int h_nstrips, h_ow, h_nseg, h_sbuf, h_mbuf;
	// k_gh2 (gpass_h2.hip: two gather outputs per lane on 1 < k < 2 axes):
	// its own strips of up to 122 outputs; h2_ow == 0: no such geometry
	int* d_h2seg;
	int h2_nstrips, h2_ow, h2_nseg, h2_sbuf, h2_mbuf;
	int v_blk, v_rs, v_rc;
	float* mid;          // intermediate image between the passes
	size_t mid_bytes;
	unsigned int* nf_flag; // float sources on the branch-free kernels: the alarm
	// LANCIR upsizing in one launch (lfuse.hip): strips whose source segment is
	// at most 64 pixels
	int lf_ow, lf_nstrips;
	int* d_lfseg;
	float* d_lfvtab;     // [v.g_len][8]: vertical coefficient rows per output
	std::vector< void* > allocs;
};

static int next_pow2( int v )
{
	int p = 2;
	while( p < v ) p <<= 1;
	return( p );
}

// Recognises PRE / POST / NONE in a lowered AVIR axis.
static bool match_avir_axis( const LAxis& ax, GPAxisHost& H )
{
	H.ok = false;
	const LOp* fir = nullptr;
	const LOp* gat = nullptr;
	int mode;

	if( ax.ops.size() == 2 && ax.ops[ 0 ].type == OP_FIR &&
		ax.ops[ 1 ].type == OP_GATHER )
	{
		mode = GP_PRE; fir = &ax.ops[ 0 ]; gat = &ax.ops[ 1 ];
	}
	else
	if( ax.ops.size() == 2 && ax.ops[ 0 ].type == OP_GATHER &&
		ax.ops[ 1 ].type == OP_FIR )
	{
		mode = GP_POST; gat = &ax.ops[ 0 ]; fir = &ax.ops[ 1 ];
	}
	else
	if( ax.ops.size() == 1 && ax.ops[ 0 ].type == OP_GATHER )
	{
		mode = GP_NONE; gat = &ax.ops[ 0 ];
	}
	else
	{
		return( false );
	}

	if( fir != nullptr && ( fir -> rf != 1 || fir -> lat != 3 ||
		fir -> view != VIEW_CLAMP || fir -> e < 0 || fir -> e > 3 ))
	{
		return( false );
	}

	if( gat -> view == VIEW_RAW || gat -> maxtaps < 2 || gat -> maxtaps > 64 )
	{
		return( false );
	}

	for( int j = 0; j < gat -> out_len; j++ )
	{
		if( gat -> h_ntaps[ j ] != gat -> maxtaps )
		{
			return( false ); // (a shorter row would multiply padded taps)
		}

		if( j > 0 && gat -> h_start[ j ] < gat -> h_start[ j - 1 ])
		{
			return( false );
		}
	}

	GPAxis& A = H.a;
	memset( &A, 0, sizeof( A ));
	A.mode = mode; A.lanc = 0; A.nt = gat -> maxtaps;
	A.in_len = ax.src_len;
	A.view_mmax = ( gat -> view == VIEW_ZS ? gat -> zs_mmax : 0x7fffffff );
	A.g_len = gat -> out_len;
	A.out_len = ax.dst_len;
	A.e = ( fir != nullptr ? fir -> e : 0 );
	A.c_len = 0;

	if( mode == GP_PRE )
	{
		// (the window algebra below is written for e == lat == 3: C[m] reads
		// S[clamp(m) - 6 .. clamp(m)])
		if( fir -> e != 3 || fir -> in_len != ax.src_len ||
			gat -> in_len != fir -> out_len ||
			fir -> out_len != ax.src_len + 2 * fir -> e )
		{
			return( false );
		}

		A.c_len = fir -> out_len;
	}
	else
	{
		if( gat -> in_len != ax.src_len )
		{
			return( false );
		}

		if( mode == GP_POST && ( fir -> in_len != gat -> out_len ||
			fir -> out_len != ax.dst_len ))
		{
			return( false );
		}
	}

	if( fir != nullptr )
	{
		for( int i = 0; i < 4; i++ )
		{
			A.f[ i ] = fir -> h_flt[ i ];
		}
	}

	A.start = gat -> d_start;
	A.coef = gat -> d_coef;
	A.fidx = nullptr;
	H.start = gat -> h_start;
	H.h_coef = gat -> h_coef.data();
	H.h_fidx = nullptr;
	H.ok = true;
	return( true );
}

static void match_lancir_axis( const LancirAxisDev& L, int order,
	GPAxisHost& H )
{
	GPAxis& A = H.a;
	memset( &A, 0, sizeof( A ));
	H.ok = ( L.kernel_len >= 2 && L.kernel_len <= 64 &&
		( L.kernel_len & 1 ) == 0 );

	A.mode = GP_NONE; A.lanc = ( order >= 1 && order <= 3 ? order : 4 );
	A.nt = L.kernel_len;
	A.in_len = L.src_len; A.view_mmax = 0x7fffffff;
	A.g_len = L.dst_len; A.out_len = L.dst_len; A.e = 0;
	A.start = L.d_start; A.coef = L.d_flt; A.fidx = L.d_fidx;
	H.start = L.h_start;
	H.h_coef = L.h_flt.data();
	H.h_fidx = L.h_fidx.data();

	for( size_t j = 1; j < H.start.size(); j++ )
	{
		H.ok &= ( H.start[ j ] >= H.start[ j - 1 ]);
	}
}

bool fused_dn_both( const avirhip_plan* p ); // fused.hip

int gpass_prepare( avirhip_plan* p )
{
	p -> gpass = nullptr;

	if( p -> ch != 4 || getenv( "AVIRHIP_NO_GPASS" ) != nullptr )
	{
		return( AVIRHIP_OK );
	}

	GPData* D = new GPData();
	D -> mid = nullptr; D -> mid_bytes = 0; D -> d_hseg = nullptr;
	D -> d_h2seg = nullptr; D -> h2_ow = 0; D -> h2_nstrips = 0;
	D -> h2_nseg = 0; D -> h2_sbuf = 0; D -> h2_mbuf = 0;
	D -> nf_flag = nullptr;
	D -> lf_ow = 0; D -> lf_nstrips = 0; D -> d_lfseg = nullptr;
	D -> d_lfvtab = nullptr;
	D -> sa_h = nullptr; D -> sa_v = nullptr;
	D -> h_geom = false; D -> v_geom = false;

	if( p -> is_lancir )
	{
		// (l_order: the channel count whose summation order applies -- the
		// padded RGBA inner plan of a 1-3 channel image keeps the image's)
		match_lancir_axis( p -> lv, p -> l_order, D -> v );
		match_lancir_axis( p -> lh, p -> l_order, D -> h );

		if( p -> l_order != 4 && ( p -> lv.kernel_len < 4 ||
			p -> lh.kernel_len < 4 ))
		{
			D -> h.ok = false;
		}

		// the float pipeline only: integer I/O and the OutMul / clamp output
		// stage stay with the generic kernels
		if( p -> in_type != AVIRHIP_F32 || p -> out_type != AVIRHIP_F32 ||
			!p -> l_unity )
		{
			D -> h.ok = false;
		}
	}
	else
	{
		match_avir_axis( p -> h, D -> h );
		match_avir_axis( p -> v, D -> v );
	}

	if( !D -> h.ok || !D -> v.ok )
	{
		delete D;
		return( AVIRHIP_OK );
	}

	{
		int rc = sacc_prepare_axis( D -> h, D -> allocs, &D -> sa_h );
		if( rc == 0 ) rc = sacc_prepare_axis( D -> v, D -> allocs, &D -> sa_v );

		if( rc != 0 )
		{
			p -> gpass = D;
			gpass_release( p );
			return( rc );
		}
	}

	// ---- horizontal pass geometry: strips of `ow` outputs whose gather
	// inputs (and, PRE, their FIR inputs) fit 64-lane pieces
	{
		const GPAxis& A = D -> h.a;
		const std::vector< int >& st = D -> h.start;
		// Strip width: every candidate that fits is priced by the pixels a
		// wave moves and the lanes it keeps busy per output -- LDS-DMA pieces
		// are 64 pixels whatever the segment needs (a 67-pixel segment costs
		// two; the H pass of 1920 -> 2500 moved 88 MB for a 33 MB source with
		// 64-output strips), and a strip narrower than the wave idles lanes.
		int ow = 0;
		std::vector< int > seg;
		int nstrips = 0, maxs = 0, maxm = 0;
		double best_cost = 0.0;

		for( int cw = ( A.mode == GP_POST ? 58 : 64 ); cw >= 8; cw-- )
		{
			const int cn = ( A.out_len + cw - 1 ) / cw;
			std::vector< int > cseg( (size_t) cn * 4, 0 );
			int cs = 0, cm = 0;

			for( int s = 0; s < cn; s++ )
			{
				const int ja = s * cw;
				const int jb = std::min( A.out_len, ja + cw ) - 1;
				int ga = ja, gb = jb;

				if( A.mode == GP_POST )
				{
					ga = std::max( 0, std::min( ja - A.e - 3, A.g_len - 1 ));
					gb = std::max( 0, std::min( jb - A.e + 3, A.g_len - 1 ));
				}

				// gather input range
				int ia = st[ ga ], ib = st[ gb ] + A.nt - 1;
				int sa = ia, sb = ib, ma = ga, mb = gb;

				if( A.mode == GP_PRE )
				{
					// C[m] reads S[clamp(m) - 6 .. clamp(m)]
					ma = ia; mb = ib;
					sa = std::max( 0, std::min( ia, A.c_len - 1 )) - 2 * A.e;
					sb = std::max( 0, std::min( ib, A.c_len - 1 ));
				}

				cseg[ s * 4 ] = sa; cseg[ s * 4 + 1 ] = sb - sa + 1;
				cseg[ s * 4 + 2 ] = ma; cseg[ s * 4 + 3 ] = mb - ma + 1;
				cs = std::max( cs, sb - sa + 1 );
				cm = std::max( cm, mb - ma + 1 );
			}

			// the FIR and POST stages run one buffer entry per lane and pass
			if( cs > 256 || ( A.mode != GP_NONE && cm > 64 ))
			{
				continue;
			}

			const double cost = ((( cs + 63 ) / 64 ) * 64 + 64 ) / (double) cw;

			if( ow == 0 || cost < best_cost )
			{
				best_cost = cost;
				ow = cw; nstrips = cn; maxs = cs; maxm = cm;
				seg.swap( cseg );
			}
		}

		if( ow == 0 && D -> sa_h == nullptr )
		{
			p -> gpass = D;
			gpass_release( p );
			return( AVIRHIP_OK );
		}

		D -> h_geom = ( ow != 0 );
		D -> h_ow = ow;
		D -> h_nstrips = nstrips;
		D -> h_nseg = ( maxs + 63 ) / 64;
		D -> h_sbuf = D -> h_nseg * 64;
		D -> h_mbuf = std::max( 64, (( maxm + 63 ) / 64 ) * 64 );
		void* q = nullptr;
		if( ow != 0 )
		{
		AVIRHIP_HIPCHECK( hipMalloc( &q, seg.size() * sizeof( int )));
		D -> allocs.push_back( q );
		AVIRHIP_HIPCHECK( hipMemcpy( q, seg.data(), seg.size() * sizeof( int ),
			hipMemcpyHostToDevice ));
		D -> d_hseg = (int*) q;
		}
	}

	// ---- ... and the strips of k_gh2: gather-then-FIR axes over the zero-
	// stuffed view (1 < k < 2) with 13 .. 25 taps whose consecutive outputs
	// start 1 or 2 source pixels apart -- a lane evaluates two of them from one
	// window of NT + 2 pixels
	{
		const GPAxis& A = D -> h.a;
		const std::vector< int >& st = D -> h.start;
		bool ok = ( A.mode == GP_POST && !A.lanc && A.nt >= 13 && A.nt <= 25 &&
			A.g_len >= 2 && A.out_len >= 2 );

		for( int g = 0; ok && g + 1 < A.g_len; g++ )
		{
			const int d = st[ g + 1 ] - st[ g ];
			ok = ( d == 1 || d == 2 );
		}

		int ow = 0, nstrips = 0, maxs = 0, maxm = 0;
		std::vector< int > seg;
		double best_cost = 0.0;

		for( int cw = 122; ok && cw >= 16; cw-- )
		{
			const int cn = ( A.out_len + cw - 1 ) / cw;
			std::vector< int > cseg( (size_t) cn * 4, 0 );
			int cs = 0, cm = 0;

			for( int sx = 0; sx < cn; sx++ )
			{
				const int ja = sx * cw;
				const int jb = std::min( A.out_len, ja + cw ) - 1;
				const int ga = std::max( 0, std::min( ja - A.e - 3, A.g_len - 1 ));
				const int gb = std::max( 0, std::min( jb - A.e + 3, A.g_len - 1 ));
				const int sa = st[ ga ], sb = st[ gb ] + A.nt - 1;

				cseg[ sx * 4 ] = sa; cseg[ sx * 4 + 1 ] = sb - sa + 1;
				cseg[ sx * 4 + 2 ] = ga; cseg[ sx * 4 + 3 ] = gb - ga + 1;
				cs = std::max( cs, sb - sa + 1 );
				cm = std::max( cm, gb - ga + 1 );
			}

			if( cs > 256 || cm > 128 )
			{
				continue;
			}

			// (pixels a wave moves by DMA, lanes it keeps busy, per output)
			const double cost = ((( cs + 63 ) / 64 ) * 64 + 64 ) / (double) cw;

			if( ow == 0 || cost < best_cost )
			{
				best_cost = cost;
				ow = cw; nstrips = cn; maxs = cs; maxm = cm;
				seg.swap( cseg );
			}
		}

		if( ow != 0 )
		{
			void* q = nullptr;
			AVIRHIP_HIPCHECK( hipMalloc( &q, seg.size() * sizeof( int )));
			D -> allocs.push_back( q );
			AVIRHIP_HIPCHECK( hipMemcpy( q, seg.data(), seg.size() * sizeof( int ),
				hipMemcpyHostToDevice ));
			D -> d_h2seg = (int*) q;
			D -> h2_ow = ow; D -> h2_nstrips = nstrips;
			D -> h2_nseg = ( maxs + 63 ) / 64;
			D -> h2_sbuf = D -> h2_nseg * 64;
			D -> h2_mbuf = 128;
			(void) maxm;
		}
	}

	// ---- vertical pass geometry: LDS rings hold what the gathers address
	// dynamically, one window plus the row being written
	{
		const GPAxis& A = D -> v.a;
		int rs = 2, rc = 2;

		// tap counts the register-window kernels exist for (launch_gv): no
		// window in LDS at all
		bool rw;

		if( A.lanc )
		{
			rw = ( A.nt >= 6 && A.nt <= 24 && ( A.nt & 1 ) == 0 );
		}
		else
		if( A.mode == GP_PRE )
		{
			rw = ( A.nt == 12 );
		}
		else
		{
			rw = ( A.mode == GP_POST && A.nt >= 13 && A.nt <= 25 );
		}

		if( rw )
		{
			rc = ( A.mode == GP_POST ? 8 : 2 );
		}
		else
		if( A.mode == GP_PRE )
		{
			rc = next_pow2( A.nt + 1 );
		}
		else
		{
			// the window, the row pair being consumed and GV_PF pairs in flight
			rs = next_pow2( A.nt + 2 * GV_PF + 3 );
			rc = ( A.mode == GP_POST ? 8 : 2 );
		}

		// (+ the PRE landing queue: 5 pair slots and one dump slot)
		D -> v_geom = (( rs + rc ) * 512 + GV_QB + 6 * 1024 <= 64 * 1024 );

		if( !D -> v_geom && D -> sa_v == nullptr )
		{
			p -> gpass = D;
			gpass_release( p );
			return( AVIRHIP_OK );
		}

		D -> v_blk = 1; D -> v_rs = rs; D -> v_rc = rc;
	}

	// ---- LANCIR upsizing, both kernels 6 taps: the fused launch's strips --
	// the widest whose source segments fit one 64-pixel piece
	if( p -> is_lancir && D -> h.a.lanc != 0 && D -> v.a.lanc == D -> h.a.lanc &&
		D -> h.a.nt == 6 && D -> v.a.nt == 6 &&
		getenv( "AVIRHIP_NO_LFUSE" ) == nullptr )
	{
		const GPAxis& A = D -> h.a;
		const std::vector< int >& st = D -> h.start;

		for( int cw = 64; cw >= 16 && D -> lf_ow == 0; cw-- )
		{
			const int cn = ( A.out_len + cw - 1 ) / cw;
			std::vector< int > seg( (size_t) cn * 2, 0 );
			bool fits = true;

			for( int s = 0; s < cn && fits; s++ )
			{
				const int ja = s * cw;
				const int jb = std::min( A.out_len, ja + cw ) - 1;
				const int sa = st[ ja ], sb = st[ jb ] + A.nt - 1;
				seg[ s * 2 ] = sa; seg[ s * 2 + 1 ] = sb - sa + 1;
				fits = ( sb - sa + 1 <= 64 );
			}

			if( fits )
			{
				void* q = nullptr;
				AVIRHIP_HIPCHECK( hipMalloc( &q, seg.size() * sizeof( int )));
				D -> allocs.push_back( q );
				AVIRHIP_HIPCHECK( hipMemcpy( q, seg.data(), seg.size() *
					sizeof( int ), hipMemcpyHostToDevice ));
				D -> d_lfseg = (int*) q;

				// the vertical axis' coefficient rows, one per OUTPUT row
				const GPAxis& V = D -> v.a;
				std::vector< float > vt( (size_t) V.g_len * 8, 0.0f );

				for( int g = 0; g < V.g_len; g++ )
				{
					memcpy( &vt[ (size_t) g * 8 ], D -> v.h_coef +
						(size_t) D -> v.h_fidx[ g ] * V.nt, V.nt * sizeof( float ));
				}

				AVIRHIP_HIPCHECK( hipMalloc( &q, vt.size() * sizeof( float )));
				D -> allocs.push_back( q );
				AVIRHIP_HIPCHECK( hipMemcpy( q, vt.data(), vt.size() *
					sizeof( float ), hipMemcpyHostToDevice ));
				D -> d_lfvtab = (float*) q;
				D -> lf_ow = cw; D -> lf_nstrips = cn;
			}
		}
	}

	// row offsets travel as 32-bit byte offsets
	const long mid_w = ( p -> is_lancir ? p -> src_w : p -> new_w );
	const long mid_h = ( p -> is_lancir ? p -> new_h : p -> src_h );

	if( (long) p -> src_h * p -> src_stride * 4 >= ( 1L << 31 ) ||
		(long) p -> src_h * p -> src_w * 16 >= ( 1L << 31 ) ||
		mid_h * mid_w * 16 >= ( 1L << 31 ))
	{
		p -> gpass = D;
		gpass_release( p );
		return( AVIRHIP_OK );
	}

	D -> mid_bytes = (size_t) mid_w * mid_h * 16;
	p -> gpass = D;
	return( AVIRHIP_OK );
}

void gpass_release( avirhip_plan* p )
{
	GPData* D = (GPData*) p -> gpass;

	if( D != nullptr )
	{
		for( size_t i = 0; i < D -> allocs.size(); i++ )
		{
			(void) hipFree( D -> allocs[ i ]);
		}

		sacc_release_axis( D -> sa_h );
		sacc_release_axis( D -> sa_v );
		delete D;
	}

	p -> gpass = nullptr;
}

bool gpass_ok( const avirhip_plan* p )
{
	return( p -> gpass != nullptr );
}

// Whether the pass kernels are the fastest path for this plan (measured on
// MI355X, profiles/r02_gpass): LANCIR always; AVIR when both axes upsize (FIR7
// then 12-tap gather) from a float RGBA source -- integer sources keep the
// tiled kernels, whose loader converts them in place of a pack pass, and the
// long-tap downsizing chains are still faster there too.
bool gpass_preferred( const avirhip_plan* p )
{
	const GPData* D = (const GPData*) p -> gpass;

	if( D == nullptr )
	{
		return( false );
	}

	if( p -> is_lancir )
	{
		return( true );
	}

	// both axes downsize by 2 or more: the streaming-accumulation kernel
	// (any source type: its first pass converts)
	// (measured, profiles/r02_sacc: it wins where its first pass replaces the
	// pack pass or a long-tap tile loader -- integer or narrower sources; for
	// float RGBA the gather / whole-ratio kernels stay ahead)
	// (1 < k < 2, round 4: large frames only -- 1920x1080 -> 1280x720 RGB uint8
	// ties with the tiles at 0.061 ms, 1280x720 -> 854x480 loses 0.048 vs 0.037)
	const bool zs_axes = ( sacc_is_zs( D -> sa_h ) || sacc_is_zs( D -> sa_v ));

	if( D -> sa_h != nullptr && D -> sa_v != nullptr && !p -> gamma &&
		( p -> in_type == AVIRHIP_U8 || p -> in_type == AVIRHIP_U16 ||
		( p -> in_type == AVIRHIP_F32 && p -> io_ch != 4 && !zs_axes )) &&
		( !zs_axes || (long) p -> new_w * p -> new_h >= 2000000L ))
	{
		return( true );
	}

	// ... and float sources (any channel count) of large frames: the same
	// kernels run optimistically (gpass_run), 3840x2160 -> 1500x844 RGBA 0.103
	// against 0.125 ms on the tiles, 5184x3456 -> 1920x1280 0.19 against 0.265;
	// at 1920x1080 -> 700x394 the four launches lose (0.065 against 0.045).
	// Whole-number ratios keep their own kernels (dnf.hip: cfg4 0.046 ms).
	if( D -> sa_h != nullptr && D -> sa_v != nullptr && !p -> gamma &&
		p -> in_type == AVIRHIP_F32 && sacc_has_two( D -> sa_h ) &&
		sacc_has_two( D -> sa_v ) && !fused_dn_both( p ) &&
		sa_wanted( D -> sa_h, p -> io_ch == 4, p -> io_ch == 4 ) &&
		sa_wanted( D -> sa_v, p -> io_ch == 4, p -> io_ch == 4 ) &&
		(long) p -> src_w * p -> src_h >= 6000000L )
	{
		return( true );
	}

	// both axes upsize: float RGBA sources always; sources that need the pack
	// pass (integer, 1-3 channels, gamma) once the frame is large enough for
	// two more launches not to matter -- 1920x1080 -> 2500x1400 RGB u8: 0.068
	// against 0.094 ms on the tiles (whose loader converts in place), but
	// 640x480 -> 1024x768: 0.036 against 0.032
	// both axes downsize by 1 < k < 2 (zero-stuffed 13 .. 25-tap gathers, then
	// the correction FIR): ahead of the tiles on large frames (3840x2160 ->
	// 2560x1440 RGBA float 0.137 against 0.179 ms), even with them at 1080p
	if( D -> h.a.mode == GP_POST && D -> v.a.mode == GP_POST &&
		(( D -> sa_h == nullptr && D -> sa_v == nullptr ) || zs_axes ))
	{
		return( (long) p -> new_w * p -> new_h >= 2000000L );
	}

	if( D -> h.a.mode != GP_PRE || D -> v.a.mode != GP_PRE )
	{
		return( false );
	}

	if( p -> in_type == AVIRHIP_F32 && p -> io_ch == 4 && !p -> gamma )
	{
		return( true );
	}

	// integer / narrow pixels on both sides without gamma: the two pass
	// kernels read and write the caller's images themselves -- two launches,
	// ahead of the tiles at every size (640x480 -> 1024x768 RGB u8: 0.027
	// against 0.032 ms)
	const bool raw_in = ( !p -> gamma && ( p -> in_type == AVIRHIP_U8 ||
		p -> in_type == AVIRHIP_U16 || p -> in_type == AVIRHIP_F32 ) &&
		D -> h_geom && D -> h_nseg <= 4 );

	const bool fused_out = ( !p -> gamma && (( p -> dither == AVIRHIP_DITHER_DEF &&
		( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )) ||
		p -> out_type == AVIRHIP_F32 ));

	return(( raw_in && fused_out ) ||
		(long) p -> new_w * p -> new_h >= 2000000L );
}

template< int MODE, bool LANC, int SX >
static void launch_gh2( const GHParams& P, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;
	// (SX: AVIR plans -- the raw source's element type; LANCIR plans -- the
	// output stage's kind, the source is float RGBA)
	constexpr bool FSRC = ( LANC || SX == 0 );

	// (register-resident coefficient rows for the tap counts that occur: 6
	// LANCIR, 12 AVIR upsizing; raw sources only come with AVIR plans)
	if constexpr( FSRC )
	{
		if( nt == 6 )
		{
			hipLaunchKernelGGL(( k_gh< MODE, LANC, 6, SX > ), dim3( items ),
				dim3( 64 ), lds, st, P );
			return;
		}
	}

	if constexpr( FSRC || MODE == GP_PRE )
	{
		if( nt == 12 )
		{
			hipLaunchKernelGGL(( k_gh< MODE, LANC, 12, SX > ), dim3( items ),
				dim3( 64 ), lds, st, P );
			return;
		}
	}

	// (LANCIR downsizing: the even counts up to 24, as gpassv_lanc.hip has them.
	// Through the run-time tap loop a row was 18 dependent LDS round trips: the
	// horizontal pass of CLancIR 5184x3456 -> 1920x1280 took 33 us for 3 us of
	// arithmetic, and ran slower the longer its chunks were)
	if constexpr( LANC && MODE == GP_NONE )
	{
#define GH_L( N ) case N: hipLaunchKernelGGL(( k_gh< MODE, LANC, N, SX > ), \
	dim3( items ), dim3( 64 ), lds, st, P ); return
		switch( nt )
		{
			GH_L( 8 ); GH_L( 10 ); GH_L( 14 ); GH_L( 16 ); GH_L( 18 );
			GH_L( 20 ); GH_L( 22 ); GH_L( 24 );
			default: break;
		}
#undef GH_L
	}

	// (gather-then-FIR7 axes, 1 < k < 2: 13 .. 25 taps. Float RGBA sources get a
	// compile-time tap count -- the whole window read at once, one LDS round
	// trip per row instead of one per block of four taps in the any-count
	// variant below: round 6, NOTEBOOK 10)
	if constexpr( MODE == GP_POST && !LANC && FSRC )
	{
		if( getenv( "AVIRHIP_GH_POST_ANY" ) == nullptr ) // (A/B: the old variant)
		{
#define GH_L( N ) case N: hipLaunchKernelGGL(( k_gh< MODE, LANC, N, SX > ), \
	dim3( items ), dim3( 64 ), lds, st, P ); return
			switch( nt )
			{
				GH_L( 13 ); GH_L( 14 ); GH_L( 15 ); GH_L( 16 ); GH_L( 17 );
				GH_L( 18 ); GH_L( 19 ); GH_L( 20 ); GH_L( 21 ); GH_L( 22 );
				GH_L( 23 ); GH_L( 24 ); GH_L( 25 );
				default: break;
			}
#undef GH_L
		}
	}

	// (... any other count up to 28, and raw sources: one variant with the
	// coefficient row in registers)
	if constexpr( MODE == GP_POST && !LANC )
	{
		if( nt <= 28 )
		{
			hipLaunchKernelGGL(( k_gh< MODE, LANC, -1, SX > ), dim3( items ),
				dim3( 64 ), lds, st, P );
			return;
		}
	}

	hipLaunchKernelGGL(( k_gh< MODE, LANC, 0, SX > ), dim3( items ),
		dim3( 64 ), lds, st, P );
}

// src_kind: 0 float RGBA (LDS-DMA); 1 uint8, 2 uint16, 3 float pixels read raw
// (LANCIR plans: 1 / 2 = the output stage stores uint8 RGB / RGBA pixels)
template< int MODE, bool LANC >
static void launch_gh( const GHParams& P, int src_kind, int items, size_t lds,
	hipStream_t st )
{
	switch( src_kind )
	{
		case 1: launch_gh2< MODE, LANC, 1 >( P, items, lds, st ); break;
		case 2: launch_gh2< MODE, LANC, 2 >( P, items, lds, st ); break;
		case 3: launch_gh2< MODE, LANC, ( LANC ? 0 : 3 ) >( P, items, lds, st );
			break;
		default: launch_gh2< MODE, LANC, 0 >( P, items, lds, st ); break;
	}
}

static int run_h( const GPData* D, const float* src, long src_ss, int src_w,
	float* dst, long dst_ss, int dst_row0, int dst_w, int row_lo, int row_hi,
	hipStream_t st, const void* raw = nullptr, int raw_type = 0,
	int raw_ch = 0, long raw_stride = 0, const GPLOut* lout = nullptr )
{
	GHParams P;
	memset( &P.lout, 0, sizeof( P.lout ));
	P.dbg = 0;
#ifdef GP_DBG
	P.dbg = ( getenv( "AVIRHIP_GP_DBG" ) != nullptr ?
		atoi( getenv( "AVIRHIP_GP_DBG" )) : 0 );
#endif

	if( lout != nullptr )
	{
		P.lout = *lout;
	}

	P.raw = raw; P.raw_ss = raw_stride; P.raw_ch = raw_ch;
	const int src_kind = ( raw == nullptr ? 0 : ( raw_type == AVIRHIP_U8 ? 1 :
		( raw_type == AVIRHIP_U16 ? 2 : 3 )));

	if( raw != nullptr && ( D -> h_nseg > 4 || D -> h.a.lanc ))
	{
		return( 1 );
	}

	P.raw_elems = (long) ( row_hi - 1 ) * raw_stride + (long) src_w * raw_ch;

	if( raw != nullptr && P.raw_elems < 4 )
	{
		return( 1 );
	}

	P.src = src; P.src_ss = src_ss; P.src_w = src_w;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0; P.dst_w = dst_w;
	P.ax = D -> h.a;
	P.row_lo = row_lo; P.row_hi = row_hi;

	// float RGBA rows of a 1 < k < 2 axis with 22 .. 25 taps (k from about 1.7):
	// two outputs per lane (gpass_h2.hip). Same box, three alternations
	// (profiles/r06_gh_post/): 3840x2160 -> 2200x1238 (22 taps) 0.0913 -> 0.0865 ms,
	// -> 2000x1125 (25) 0.0924 -> 0.0861; -> 2560x1440 (19 taps, k = 1.5) LOSES,
	// 0.0976 -> 0.1136: both forms issue the same number of vector instructions
	// per output and run at the same issue rate -- k_gh2 only removes LDS work
	// (reads 2.6 M -> 1.6 M, bank conflicts 7.2 M -> 0.2 M cycles per launch),
	// which binds at k = 1.9 (lanes 30 bytes apart) and not at k = 1.5, where
	// its half as many, twice as long waves cost more than they save.
	// AVIRHIP_NO_GH2 / AVIRHIP_GH2_MIN_NT: A/B.
	// (read per call: the GPU suite runs every form on every tap count)
	const int gh2_min_nt = ( getenv( "AVIRHIP_GH2_MIN_NT" ) != nullptr ?
		atoi( getenv( "AVIRHIP_GH2_MIN_NT" )) : 22 );

	if( D -> h2_ow != 0 && raw == nullptr && lout == nullptr &&
		P.ax.nt >= gh2_min_nt && getenv( "AVIRHIP_NO_GH2" ) == nullptr )
	{
		P.nstrips = D -> h2_nstrips; P.ow = D -> h2_ow;
		P.seg = D -> d_h2seg; P.nseg = D -> h2_nseg;
		P.sbuf = D -> h2_sbuf; P.mbuf = D -> h2_mbuf;
		const int rows2 = row_hi - row_lo;
		const size_t lds2 = (size_t) ( GH_NB * P.sbuf + P.mbuf ) * 16;
		const int wpc2 = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
			( lds2 + 512 ))));

		P.chunk = balanced_chunk( rows2, P.nstrips, 4, rows2, 2, wpc2, false );

		if( getenv( "AVIRHIP_GH_CHUNK" ) != nullptr ) // tuning aid
		{
			P.chunk = std::max( 1, std::min( atoi( getenv( "AVIRHIP_GH_CHUNK" )),
				rows2 ));
		}

		const int items2 = P.nstrips * (( rows2 + P.chunk - 1 ) / P.chunk );

		if( launch_gh2( P, items2, lds2, st ) == 0 )
		{
			AVIRHIP_HIPCHECK( hipGetLastError() );
			return( AVIRHIP_OK );
		}
	}

	P.nstrips = D -> h_nstrips; P.ow = D -> h_ow;
	P.seg = D -> d_hseg; P.nseg = D -> h_nseg;
	P.sbuf = D -> h_sbuf; P.mbuf = D -> h_mbuf;

	const int rows = row_hi - row_lo;
	// (tap counts 6 and 12 keep the coefficient row in registers: no LDS table)
	const bool regcoef = ( P.ax.nt == 6 || P.ax.nt == 12 ||
		( P.ax.lanc && raw == nullptr && P.ax.nt >= 6 && P.ax.nt <= 24 &&
		( P.ax.nt & 1 ) == 0 ) ||
		( P.ax.mode == GP_POST && !P.ax.lanc && P.ax.nt <= 28 ));
	const size_t lds_h = (size_t) ( GH_NB * P.sbuf + P.mbuf ) * 16 +
		( regcoef ? 0 : (size_t) (( P.ax.nt + 3 ) & ~3 ) * 64 * 4 );

	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_h + 512 ))));

	// (a chunk's fixed cost: its coefficient row load and three rows of DMA
	// lead, about two rows of work)
	P.chunk = balanced_chunk( rows, P.nstrips, 4, rows, 2, wpc, false );

	if( getenv( "AVIRHIP_GH_CHUNK" ) != nullptr ) // tuning aid
	{
		P.chunk = std::max( 1, std::min( atoi( getenv( "AVIRHIP_GH_CHUNK" )), rows ));
	}

	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = lds_h;
	const GPAxis& A = P.ax;

	if( A.lanc )
	{
		launch_gh< GP_NONE, true >( P, ( P.lout.on != 0 &&
			P.lout.type == AVIRHIP_U8 && P.lout.ch >= 3 ? P.lout.ch - 2 : 0 ),
			items, lds, st );
	}
	else
	if( A.mode == GP_PRE )
	{
		launch_gh< GP_PRE, false >( P, src_kind, items, lds, st );
	}
	else
	if( A.mode == GP_POST )
	{
		launch_gh< GP_POST, false >( P, src_kind, items, lds, st );
	}
	else
	{
		launch_gh< GP_NONE, false >( P, src_kind, items, lds, st );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// Source rows (AVIR: of the H pass' output; LANCIR: unused) the vertical pass
// reads for output rows [row0, row1).
static void v_source_rows( const GPData* D, int row0, int row1, int& a, int& b )
{
	const GPAxis& A = D -> v.a;
	const std::vector< int >& st = D -> v.start;
	int ga = row0, gb = row1 - 1;

	if( A.mode == GP_POST )
	{
		ga = std::max( 0, std::min( row0 - A.e - 3, A.g_len - 1 ));
		gb = std::max( 0, std::min( row1 - 1 - A.e + 3, A.g_len - 1 ));
	}

	a = st[ ga ]; b = st[ gb ] + A.nt - 1;

	if( A.mode == GP_PRE )
	{
		a = std::max( 0, std::min( a, A.c_len - 1 )) - 2 * A.e;
		b = std::max( 0, std::min( b, A.c_len - 1 ));
	}

	a = std::max( 0, std::min( a, A.in_len - 1 ));
	b = std::max( 0, std::min( b, A.in_len - 1 ));
}

// Runs both passes for output rows [row0, row1) of a float RGBA plan. Returns
// 1 when the call cannot take this path (alignment), so that the caller falls
// back.
// LANCIR (an inner plan): whether its vertical pass -- the first one -- can
// read the owner's integer / narrower image itself (whole-pixel lanes).
// Whether the vertical pass of a LANCIR plan reads the owner's image as it is
// (`raw`: element type, channels, elements per row): its rows travel as bytes
// by LDS-DMA, so the base and the row pitch have to be dword-aligned, the image
// under 2 GiB, and the tap count one of the register-window variants of k_gv
// (the LDS-window form keeps a ring of float pixels).
bool gpass_lancir_takes_raw( const avirhip_plan* p, const void* raw, int type,
	int ch, long stride )
{
	const GPData* D = (const GPData*) p -> gpass;
	const int lpath = ( p -> path != 0 ? p -> path : p -> auto_path );
	const long esz = ( type == AVIRHIP_U8 ? 1 : ( type == AVIRHIP_U16 ? 2 : 4 ));
	static const bool off = ( getenv( "AVIRHIP_GV_RAWDMA" ) != nullptr &&
		atoi( getenv( "AVIRHIP_GV_RAWDMA" )) == 0 ); // (A/B: the pack pass instead)

	if( D == nullptr || off )
	{
		return( false );
	}

	const int nt = D -> v.a.nt;

	return( p -> is_lancir && lpath == 5 && D -> v_geom &&
		D -> h_geom && p -> src_w > 48 &&
		( D -> v_rs + D -> v_rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 &&
		nt >= 6 && nt <= 24 && ( nt & 1 ) == 0 &&
		( (uintptr_t) raw & 3 ) == 0 && (( stride * esz ) & 3 ) == 0 &&
		// (the DMA's num_records is this rounded up to a dword, in an int)
		( (long) ( p -> src_h - 1 ) * stride + (long) p -> src_w * ch ) * esz <=
		0x7ffffffcL );
}

// Whether the first pass reads the caller's image as it is (integer types,
// 1-4 channels): AVIR plans whose horizontal axis streams (sacc.hip).
bool gpass_takes_raw( const avirhip_plan* p )
{
	const GPData* D = (const GPData*) p -> gpass;
	const bool int_src = ( p -> in_type == AVIRHIP_U8 ||
		p -> in_type == AVIRHIP_U16 );

	return( D != nullptr && !p -> is_lancir &&
		( sa_wanted( D -> sa_h, int_src || ( p -> in_type == AVIRHIP_F32 &&
		p -> io_ch == 4 ), p -> in_type == AVIRHIP_F32 && p -> io_ch == 4 ) ||
		( D -> h_geom && D -> h_nseg <= 4 )));
}

// `iout` != nullptr: the caller's integer image (the band's first row) -- the
// last pass converts and stores there (no float result, no epilogue pass);
// returns 1 if this plan's last pass cannot (the caller then runs unfused).
int gpass_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st, const void* raw, int raw_type,
	int raw_ch, long raw_stride, void* iout )
{
	GPOut O;
	memset( &O, 0, sizeof( O ));

	if( iout != nullptr )
	{
		if( p -> is_lancir || p -> gamma ||
			( p -> out_type != AVIRHIP_F32 &&
			p -> dither != AVIRHIP_DITHER_DEF ) ||
			( p -> out_type != AVIRHIP_U8 && p -> out_type != AVIRHIP_U16 &&
			!( p -> out_type == AVIRHIP_F32 && p -> io_ch < 4 )) ||
			( p -> out_type == AVIRHIP_U8 && p -> io_ch == 4 &&
			(( (uintptr_t) iout & 3 ) != 0 )))
		{
			return( 1 );
		}

		gp_make_out( p, iout, O );
		dst = (float*) iout; // (alignment checks below: not written as floats)
	}

	const GPOut* const out = ( iout != nullptr ? &O : nullptr );

	GPData* D = (GPData*) p -> gpass;

	if( D == nullptr || ( iout == nullptr && ( (uintptr_t) dst & 15 )) ||
		( p -> is_lancir && ( p -> new_stride & 3 )) ||
		( raw == nullptr && (( (uintptr_t) src & 15 ) || ( src_stride & 3 ))) ||
		( raw != nullptr && !gpass_takes_raw( p )))
	{
		return( 1 );
	}

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	// Large upsizing plans of float RGBA sources run both passes in one launch
	// (gfuse.hip): no intermediate image. "Large": a fused chunk repeats the
	// horizontal arithmetic of its 18-row vertical run-in, which the saved
	// FltBuf round trip only pays for once the frame is memory-heavy -- measured
	// (tools/gf_sweep.py, fused / two-pass, profiles/r04_upg/gf_sweep.txt):
	// 1920x1080 x1.3 1.06, x1.6 0.99, x1.9 0.94, x2.3 0.90, x3.0 0.78;
	// 3840x2160 x1.15 0.92, x1.6 0.71; 1280x720 x2.3 1.01, x3.0 0.97 -- the
	// crossing is at about 7 Mpixels of output.
	// AVIRHIP_GF_MINPIX moves it (0: always), AVIRHIP_VARIANT_UPG_TWO_PASS
	// keeps the two pass kernels.
	static const long gf_minpix = ( getenv( "AVIRHIP_GF_MINPIX" ) != nullptr ?
		atol( getenv( "AVIRHIP_GF_MINPIX" )) : 7000000L );
	// (a forced variant bit 4 = "fused whatever the size": tests)
	const bool use_gf = ( !p -> is_lancir && D -> h.a.mode == GP_PRE &&
		D -> v.a.mode == GP_PRE && D -> h.a.nt == 12 && D -> v.a.nt == 12 &&
		D -> h_geom && raw == nullptr &&
		( p -> variant & AVIRHIP_VARIANT_UPG_TWO_PASS ) == 0 &&
		((long) p -> new_w * p -> new_h >= gf_minpix ||
		( p -> variant & AVIRHIP_VARIANT_UPG_FUSED ) != 0 ) &&
		getenv( "AVIRHIP_NO_GFUSE" ) == nullptr );

	// ... and so do LANCIR's upsizing plans of float RGBA pixels (lfuse.hip)
	// (with the owner's image as the source where its rows can travel as bytes
	// -- what gpass_lancir_takes_raw promised covers it). With an INTEGER result
	// the fusion pays up to about 2.7x horizontally (RGBA: 3.2x): the vertical
	// stage works on the 64 / ratio + 6 source columns of a strip's 64 lanes,
	// and the two pass kernels' intermediate image is no longer small against a
	// uint8 result (1920x1080 RGB uint8, same box, fused / two passes: x1.5
	// 0.028 / 0.034 ms, x2.1 0.047 / 0.051, x2.5 0.064 / 0.065, x3 0.092 /
	// 0.086, x4 0.166 / 0.136; RGBA uint8 x3 0.086 / 0.088, x4 0.152 / 0.140;
	// float results: fused wins at every ratio, x3 0.082 / 0.126;
	// profiles/r05_lanc, tools/rounds/r05_lf2.sh)
	static const double lf_ratio = ( getenv( "AVIRHIP_LF_INT_RATIO" ) != nullptr ?
		atof( getenv( "AVIRHIP_LF_INT_RATIO" )) : -1.0 ); // (tuning aid)
	const bool lf_pays = ( !p -> lfuse.on || p -> lfuse.type == AVIRHIP_F32 ||
		(double) p -> new_w <= (double) p -> src_w * ( lf_ratio >= 0.0 ?
		lf_ratio : ( p -> lfuse.ch == 4 ? 3.2 : 2.7 )) ||
		( p -> variant & AVIRHIP_VARIANT_UPG_FUSED ) != 0 );
	const bool use_lf = ( p -> is_lancir && D -> lf_ow != 0 && lf_pays &&
		( !p -> lraw.on || lfuse_takes_raw( p -> lraw.ptr, p -> lraw.type,
		p -> lraw.ch, p -> lraw.stride, p -> src_h, p -> src_w )) &&
		( p -> variant & AVIRHIP_VARIANT_UPG_TWO_PASS ) == 0 );

	if( D -> mid == nullptr && !use_gf && !use_lf )
	{
		void* q = nullptr;
		AVIRHIP_HIPCHECK( hipMalloc( &q, D -> mid_bytes ));
		D -> allocs.push_back( q );
		p -> alloc_bytes += D -> mid_bytes;
		D -> mid = (float*) q;
	}

	const long dst_ss = ( p -> is_lancir ? p -> new_stride : (long) p -> new_w * 4 );
	int rc;

	if( p -> is_lancir )
	{
		if( !D -> h_geom || !D -> v_geom )
		{
			return( 1 );
		}

		if( dst == nullptr && !p -> lfuse.on )
		{
			return( AVIRHIP_NEED_DST );
		}

		// Upsizing from float RGBA pixels: both passes in ONE launch
		// (lfuse.hip) -- LANCIR's vertical-first order makes the fusion free of
		// recomputation. AVIRHIP_VARIANT_UPG_TWO_PASS keeps the two kernels.
		if( use_lf )
		{
			LFParams G;
			memset( &G, 0, sizeof( G ));
			G.src = src; G.src_ss = src_stride;
			G.dst = dst; G.dst_ss = dst_ss; G.dst_row0 = row0;
			G.hx = D -> h.a; G.vx = D -> v.a;
			G.row_lo = row0; G.row_hi = row1;
			G.nstrips = D -> lf_nstrips; G.ow = D -> lf_ow;
			G.seg = D -> d_lfseg; G.vtab = D -> d_lfvtab;

			if( p -> lraw.on )
			{
				// (an inner plan: the owner's image, read as it is)
				const int esz = ( p -> lraw.type == AVIRHIP_U8 ? 1 :
					( p -> lraw.type == AVIRHIP_U16 ? 2 : 4 ));

				G.raw = p -> lraw.ptr; G.raw_ss = p -> lraw.stride;
				G.raw_kind = ( esz == 1 ? 1 : ( esz == 2 ? 2 : 3 ));
				G.raw_ch = p -> lraw.ch; G.raw_bpp = esz * p -> lraw.ch;
				// (rounded up to whole dwords, as in gpass_run_v)
				G.raw_bytes = (int) (((( (long) ( p -> src_h - 1 ) * p -> lraw.stride +
					(long) p -> src_w * p -> lraw.ch ) * esz ) + 3 ) & ~3L );
				G.raw_tdn = ( 64 * G.raw_bpp + ( G.raw_bpp & 3 ? 3 : 0 ) + 255 ) >> 8;
			}

			if( p -> lfuse.on )
			{
				G.lout.on = 1; G.lout.type = p -> lfuse.type;
				G.lout.ch = p -> lfuse.ch; G.lout.unity = p -> lfuse.unity;
				G.lout.out_mul = p -> lfuse.out_mul;
				G.lout.clampv = p -> lfuse.clampv;
				G.lout.l4 = ( p -> new_w * p -> lfuse.ch ) & ~3;
				G.lout.base = p -> lfuse.dst; G.lout.stride = p -> lfuse.stride;
			}

			rc = lfuse_launch( G, st );
			p -> lfuse.done = ( rc == 0 && p -> lfuse.on );
			return( rc );
		}

		// vertical first (lancir.h:601-646): mid = [new_h][src_w], only the
		// rows of the band
		const long mid_ss = (long) p -> src_w * 4;

		if( p -> lraw.on )
		{
			// (an inner plan: the owner's image, read as it is)
			rc = gpass_run_v( D -> v.a, D -> v_blk, D -> v_rs, D -> v_rc, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
				row0, row1, st, nullptr, p -> lraw.ptr, p -> lraw.type,
				p -> lraw.ch, p -> lraw.stride );

			// (the kernel refused the image after all: the owner makes the
			// float copy and calls again)
			if( rc == 1 ) return( AVIRHIP_NEED_SRC );
			if( rc != 0 ) return( rc );
		}
		else
		if(( rc = gpass_run_v( D -> v.a, D -> v_blk, D -> v_rs, D -> v_rc, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
			row0, row1, st )) != 0 ) return( rc );

		if( p -> lfuse.on )
		{
			// (an inner plan: the outer plan's output stage goes into the
			// store of this pass, its result rows are never written)
			GPLOut L;
			memset( &L, 0, sizeof( L ));
			L.on = 1; L.type = p -> lfuse.type; L.ch = p -> lfuse.ch;
			L.unity = p -> lfuse.unity; L.out_mul = p -> lfuse.out_mul;
			L.clampv = p -> lfuse.clampv;
			L.l4 = ( p -> new_w * p -> lfuse.ch ) & ~3;
			L.base = p -> lfuse.dst; L.stride = p -> lfuse.stride;
			rc = run_h( D, D -> mid, mid_ss, p -> src_w, dst, dst_ss, row0,
				p -> new_w, row0, row1, st, nullptr, 0, 0, 0, &L );

			p -> lfuse.done = ( rc == 0 );
			return( rc );
		}

		if( dst == nullptr )
		{
			return( AVIRHIP_NEED_DST ); // (an inner plan run without a result buffer)
		}

		return( run_h( D, D -> mid, mid_ss, p -> src_w, dst, dst_ss, row0,
			p -> new_w, row0, row1, st ));
	}

	// Upsizing on both axes from a float RGBA source: ONE launch runs both
	// passes (gfuse.hip), FltBuf never exists. AVIRHIP_VARIANT_UPG_TWO_PASS
	// keeps the two pass kernels (tests, A/B timing).
	if( use_gf )
	{
		GFParams G;
		memset( &G, 0, sizeof( G ));
		G.src = src; G.src_ss = src_stride;
		G.dst = dst; G.dst_ss = dst_ss; G.dst_row0 = row0;
		G.hx = D -> h.a; G.vx = D -> v.a;
		G.row_lo = row0; G.row_hi = row1;
		G.nstrips = D -> h_nstrips; G.ow = D -> h_ow;
		G.seg = D -> d_hseg; G.nseg = D -> h_nseg;
		G.sbuf = D -> h_sbuf; G.mbuf = D -> h_mbuf;

		if( out != nullptr )
		{
			G.out = *out;
		}

		return( gfuse_launch( G, (double) p -> src_h / p -> new_h, st ));
	}

	// AVIR: horizontal first into mid = FltBuf [src_h][new_w], only the source
	// rows the band's vertical windows read
	const long mid_ss = (long) p -> new_w * 4;
	int a, b;
	v_source_rows( D, row0, row1, a, b );

	rc = 1;
	// an integer image is finite, and so is the FltBuf made from it: both
	// passes may take the branch-free form of the accumulation kernel
	const bool int_src = ( raw != nullptr && ( raw_type == AVIRHIP_U8 ||
		raw_type == AVIRHIP_U16 ));
	const bool ladder = (( p -> variant & AVIRHIP_VARIANT_SACC_LADDER ) != 0 );

	// Float sources, both axes streaming: OPTIMISTIC execution. The branch-free
	// kernels are exact for finite samples only; they run first, the last pass
	// raises the plan's alarm word when it emits a NaN or Inf (any non-finite
	// sample inside the call's windows ends up in one: c * Inf and 0 * Inf are
	// never finite; so does an overflow), and the exact kernels follow in the
	// stream behind the alarm -- empty launches for finite images, a complete
	// recomputation of both passes otherwise. (5184x3456 -> 1920x1280 float RGBA:
	// 0.265 ms on the tiles, 0.15 ms this way.)
	// Float RGBA sources (raw == nullptr: they travel by LDS-DMA) do NOT run
	// optimistically unless asked to (AVIRHIP_VARIANT_SACC_OPTIMISTIC): their
	// exact kernels are as fast as the branch-free ones (5184x3456 -> 1920x1280:
	// 127 + 59 us against 131 + 59, profiles/r04_sacc/exact_vs_optimistic.txt)
	// and the two launches behind the alarm cost 9 us a frame. Float pixels of
	// 1-3 channels gain: 0.273 -> 0.198 ms for the same frame as float RGB.
	static const bool no_opt = ( getenv( "AVIRHIP_NO_SACC_OPT" ) != nullptr );
	const bool fsrc = ( raw != nullptr ? raw_type == AVIRHIP_F32 : true );
	static const bool opt_rgba = ( getenv( "AVIRHIP_SACC_OPT_RGBA" ) != nullptr &&
		atoi( getenv( "AVIRHIP_SACC_OPT_RGBA" )) != 0 ); // (tuning aid)
	const bool opt_wanted = ( raw != nullptr || opt_rgba ||
		( p -> variant & AVIRHIP_VARIANT_SACC_OPTIMISTIC ) != 0 );

	// (raw == nullptr: float RGBA -- the exact kernels unless the variant asks)
	const bool sa_always = ( int_src || ( raw == nullptr && !opt_rgba &&
		( p -> variant & AVIRHIP_VARIANT_SACC_OPTIMISTIC ) == 0 ));
	// (raw == nullptr: a float RGBA source)
	const bool use_sa_h = sa_wanted( D -> sa_h, sa_always, raw == nullptr );
	const bool use_sa_v = sa_wanted( D -> sa_v, sa_always, raw == nullptr );

	if( !int_src && fsrc && !ladder && !no_opt && opt_wanted && use_sa_h &&
		use_sa_v &&
		sacc_has_two( D -> sa_h ) && sacc_has_two( D -> sa_v ))
	{
		if( D -> nf_flag == nullptr )
		{
			void* q = nullptr;
			AVIRHIP_HIPCHECK( hipMalloc( &q, 64 ));
			D -> allocs.push_back( q );
			D -> nf_flag = (unsigned int*) q;
		}

		// (the first optimistic pass lowers the alarm, the second may raise it)

		auto h_pass = [&]( const bool fast ) -> int
		{
			if( raw != nullptr )
			{
				return( sacc_run_axis( D -> sa_h, true, raw, raw_type, raw_ch,
					raw_ch, raw_stride * 4, raw_ch * 4, D -> mid, mid_ss, 4, a,
					b + 1, 0, p -> new_w, st, nullptr, fast, false, fast,
					fast ? D -> nf_flag : nullptr, fast ? nullptr : D -> nf_flag ));
			}

			return( sacc_run_axis( D -> sa_h, true, src, AVIRHIP_F32, 4,
				p -> io_ch, src_stride * 4, 16, D -> mid, mid_ss, 4, a, b + 1, 0,
				p -> new_w, st, nullptr, fast, false, fast,
				fast ? D -> nf_flag : nullptr, fast ? nullptr : D -> nf_flag ));
		};

		auto v_pass = [&]( const bool fast ) -> int
		{
			return( sacc_run_axis( D -> sa_v, false, D -> mid, AVIRHIP_F32, 4,
				p -> io_ch, 16, mid_ss * 4, dst - (long) row0 * dst_ss, 4, dst_ss,
				0, p -> new_w, row0, row1, st, out, fast, false, fast,
				fast ? D -> nf_flag : nullptr, fast ? nullptr : D -> nf_flag ));
		};

		rc = h_pass( true );

		if( rc == 0 )
		{
			rc = v_pass( true );

			if( rc == 0 )
			{
				// (the exact kernels behind the alarm)
				rc = h_pass( false );
				if( rc == 0 ) rc = v_pass( false );
				return( rc );
			}
		}

		if( rc != 1 )
		{
			return( rc );
		}
		// (a pass cannot take the branch-free form: the exact kernels below)
	}

	rc = 1;

	if( use_sa_h )
	{
		if( raw != nullptr )
		{
			const long es = ( raw_type == AVIRHIP_U8 ? 1 :
				( raw_type == AVIRHIP_U16 ? 2 : 4 ));

			rc = sacc_run_axis( D -> sa_h, true, raw, raw_type, raw_ch, raw_ch,
				raw_stride * es, raw_ch * es, D -> mid, mid_ss, 4, a, b + 1, 0,
				p -> new_w, st, nullptr, int_src, ladder );
		}
		else
		{
			rc = sacc_run_axis( D -> sa_h, true, src, AVIRHIP_F32, 4, p -> io_ch,
				src_stride * 4, 16, D -> mid, mid_ss, 4, a, b + 1, 0,
				p -> new_w, st );
		}
	}

	if( rc == 1 )
	{
		if( !D -> h_geom )
		{
			return( 1 );
		}

		rc = run_h( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
			p -> new_w, a, b + 1, st, raw, raw_type, raw_ch, raw_stride );
	}

	if( rc != 0 ) return( rc );

	rc = 1;

	if( use_sa_v )
	{
		rc = sacc_run_axis( D -> sa_v, false, D -> mid, AVIRHIP_F32, 4,
			p -> io_ch, 16, mid_ss * 4, dst - (long) row0 * dst_ss, 4, dst_ss,
			0, p -> new_w, row0, row1, st, out, int_src, ladder );
	}

	if( rc == 1 )
	{
		if( !D -> v_geom )
		{
			return( 1 );
		}

		rc = gpass_run_v( D -> v.a, D -> v_blk, D -> v_rs, D -> v_rc, D -> mid, mid_ss, p -> new_w, dst, dst_ss, row0, row0,
			row1, st, out );
	}

	return( rc );
}

} // namespace avirhip
