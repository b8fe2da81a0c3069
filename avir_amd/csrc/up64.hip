// up64.hip -- the double pipeline (avir::CImageResizer< fpclass_def< double > >,
// avir.h:4553-4560) on UPSIZING axes: marching kernels for the chain every
// upsizing axis lowers to,
//
//   C[n] = f0*S[c] + sum_{i=1..3} f_i*(S[c+i] + S[c-i]),  c = n - 3, S clamped
//          (the correction filter at the source rate, doFilter avir.h:3748-3866;
//          three outputs beyond either end of the source)
//   O[j] = 0 + sum_{t<12} coef[j][t] * C[clamp(start[j] + t)]
//          (doUpsample without a filter + doResize2, avir.h:3260-3402, 4114-4328:
//          the zero-stuffed view folded into the gather, plan.h VIEW_ZS)
//
// in place of tile64.hip's interpreter (LDS tiles, run-time op types and tap
// counts: 12 LDS reads and a coefficient read per multiply, cfg3 1.38 ms).
//
//   k_uh64  horizontal pass, caller's pixels -> FltBuf (double). A workgroup owns
//           256 consecutive ELEMENTS of an output row (channels are independent:
//           a strip may start and end inside a pixel) and marches down its rows
//           eight at a time (the next step's source rows travel in registers
//           meanwhile); a thread keeps the 12 coefficients of its output
//           column in registers for the whole march; the source segment is cast
//           to double into LDS (packScanline, avir.h:2777-2840), the correction
//           filter writes the C segment, the gather reads it.
//   k_uv64  vertical pass, FltBuf -> the caller's image through the output stage.
//           No LDS, no barriers: a lane owns one or two double elements of a row
//           and marches down the rows of its chunk with the last 7 F rows and the
//           last 12 C rows in REGISTERS (ring slots are compile-time: the loop is
//           unrolled over the ring's period), rows prefetched 5 ahead straight
//           into the ring; an output row is produced in the step its last C row
//           arrives, its 12 coefficients are wave-uniform scalar operands.
//
// Out-of-range C indices: C_ext[i] = C[clamp(i)]. Because the correction filter's
// extension (e = 3) equals its latency, the filter evaluated at the VIRTUAL index
// i over the clamped source gives exactly that value (every tap reads the edge
// sample, in the same expression), so both kernels march over virtual indices
// and clamp only source addresses -- up64_axis_ok() checks the conditions.
//
// Arithmetic contract (-ffp-contract=off): v_mul_f64 / v_add_f64 apart, taps in
// the reference's order, sums from +0.0: the per-output expressions of
// generic64.hip / tile64.hip, the same bits (tools/fuzz_f64.py, oracle/_ref
// variant 4).

#include "plan.h"
#include "dn_dev.h"
#include "f64_dev.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

namespace avirhip {

namespace {

#define U64_NT 12   // gather taps
#define U64_PF 5    // rows in flight in the vertical kernel

// (constant address space: wave-uniform tables through scalar loads -- a plain
// pointer cannot be proven unclobbered by the kernel's own stores and would be
// read by every lane through the vector memory path)
typedef const __attribute__(( address_space( 4 ))) double u64_cdouble;
typedef const __attribute__(( address_space( 4 ))) int u64_cint;

__device__ __forceinline__ int u64_cl( const int i, const int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

struct U64HParams
{
	const void* src; long src_ss; // elements of the source type per row
	int src_w;                    // pixels (clamp range)
	const double* fir;            // f0 .. f3
	const int* start;             // [new_w]
	const double* coef;           // [new_w][12]
	double* flt; long fss;        // FltBuf: row y at flt + y * fss
	int nel;                      // elements per FltBuf row (new_w * ch)
	int ya, yb;                   // source rows [ya, yb]
	int crows;                    // rows per chunk (a whole number of steps)
	int nstrips;
	int maxSe, maxCe;             // LDS row pitches (elements)
};

template< int CH, typename Tin, int RH >
__global__ void __launch_bounds__( 256 ) k_uh64( const U64HParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) double u64_smem[];
	double* const S = u64_smem;
	double* const C = u64_smem + RH * P.maxSe;

	const int tid = threadIdx.x;
	const int chunk = blockIdx.x / P.nstrips;
	const int strip = blockIdx.x - chunk * P.nstrips;
	const int eo0 = strip * 256;
	const bool ok = ( eo0 + tid < P.nel );
	const int eo = ( ok ? eo0 + tid : P.nel - 1 );
	const int j = eo / CH;
	const int k = eo - j * CH;
	const int j_lo = eo0 / CH;
	const int j_hi = min( eo0 + 255, P.nel - 1 ) / CH;
	const int c_lo = P.start[ j_lo ];
	const int c_hi = P.start[ j_hi ] + U64_NT - 1;
	const int nCe = ( c_hi - c_lo + 1 ) * CH;
	const int nSe = nCe + 6 * CH; // <= 512 (up64_run_h)

	double cf[ U64_NT ];
#pragma unroll
	for( int t = 0; t < U64_NT; t++ )
	{
		cf[ t ] = P.coef[ (long) j * U64_NT + t ];
	}

	const int off = ( P.start[ j ] - c_lo ) * CH + k;
	const double f0 = P.fir[ 0 ], f1 = P.fir[ 1 ], f2 = P.fir[ 2 ],
		f3 = P.fir[ 3 ];

	const int y0 = P.ya + chunk * P.crows;
	const int y1 = min( y0 + P.crows, P.yb + 1 );
	const int xlast = P.src_w - 1;

	// the source elements this thread moves into LDS each step: segment
	// [c_lo - 6, c_hi] as virtual indices, addresses clamped to the row
	int soff[ 2 ];
	bool sok[ 2 ];
#pragma unroll
	for( int q = 0; q < 2; q++ )
	{
		const int x = q * 256 + tid;
		const int px = x / CH;
		const int kk = x - px * CH;
		sok[ q ] = ( x < nSe );
		soff[ q ] = u64_cl( c_lo - 6 + px, xlast ) * CH + kk;
	}

	const bool two = ( nSe > 256 );
	Tin sv[ RH ][ 2 ];

	auto fetch = [&]( const int y )
	{
#pragma unroll
		for( int r = 0; r < RH; r++ )
		{
			const int row = min( y + r, P.yb );
			const Tin* const sp = (const Tin*) P.src + (long) row * P.src_ss;
			sv[ r ][ 0 ] = ( sok[ 0 ] ? sp[ soff[ 0 ]] : (Tin) 0 );
			sv[ r ][ 1 ] = ( two && sok[ 1 ] ? sp[ soff[ 1 ]] : (Tin) 0 );
		}
	};

	fetch( y0 );

	for( int y = y0; y < y1; y += RH )
	{
		// ---- the rows fetched during the previous step, cast to double
		// (packScanline, avir.h:2777-2840)
#pragma unroll
		for( int r = 0; r < RH; r++ )
		{
			if( sok[ 0 ])
			{
				S[ r * P.maxSe + tid ] = (double) sv[ r ][ 0 ];
			}

			if( two && sok[ 1 ])
			{
				S[ r * P.maxSe + 256 + tid ] = (double) sv[ r ][ 1 ];
			}
		}

		__syncthreads();

		// (the next step's rows travel while this one computes)
		if( y + RH < y1 )
		{
			fetch( y + RH );
		}

		// ---- correction filter: C_v[i], i = c_lo + x / CH, centre S_v[i - 3]
#pragma unroll
		for( int r = 0; r < RH; r++ )
		{
			for( int x = tid; x < nCe; x += 256 )
			{
				const double* const s = S + r * P.maxSe + x;
				double a = f0 * s[ 3 * CH ];
				a = a + f1 * ( s[ 4 * CH ] + s[ 2 * CH ]);
				a = a + f2 * ( s[ 5 * CH ] + s[ 1 * CH ]);
				a = a + f3 * ( s[ 6 * CH ] + s[ 0 ]);
				C[ r * P.maxCe + x ] = a;
			}
		}

		__syncthreads();

		// ---- the gather: one element of each of the rows per thread
#pragma unroll
		for( int r = 0; r < RH; r++ )
		{
			const double* const c = C + r * P.maxCe + off;
			double sum = 0.0;
#pragma unroll
			for( int t = 0; t < U64_NT; t++ )
			{
				sum = sum + cf[ t ] * c[ t * CH ];
			}

			if( ok && y + r < y1 )
			{
				P.flt[ (long) ( y + r ) * P.fss + eo ] = sum;
			}
		}

		// (two barriers per step are enough: the next step writes S before its
		// first barrier -- nothing reads S after this step's second one -- and C
		// after it, when every thread has finished this gather)
	}
}

// the vertical chain's tables and the destination behind the output stage
struct U64Out
{
	const double* fir;            // f0 .. f3
	const int* start;             // [new_h]
	const double* coef;           // [new_h][12]
	void* dst; long dst_ss;       // elements of the output type per row
	int dst_row0;
	int row_lo, row_hi;           // output rows [lo, hi)
	int crows;                    // output rows per chunk
	int vec_st;                   // EPL == 2: rows are aligned for paired stores
	int use_tr; double tr_mul, tr_muli, pk_out;
};

struct U64VParams
{
	const double* flt; long fss;  // FltBuf rows
	int nel;                      // elements per row
	int in_len;                   // F rows (clamp range)
	int ngroups;                  // groups of 64 * EPL elements
	U64Out O;
};

template< int EPL >
__device__ __forceinline__ void u64_ld( const double* const p,
	double ( &v )[ EPL ])
{
	if constexpr( EPL == 2 )
	{
		typedef double d2 __attribute__(( ext_vector_type( 2 )));
		const d2 t = *(const d2*) p;
		v[ 0 ] = t.x;
		v[ 1 ] = t.y;
	}
	else
	{
		v[ 0 ] = p[ 0 ];
	}
}

// The vertical chain on the ring, phase PH: row u of the march (C / F index i)
// is in F[ PH ]; C_v[ i ] from rows u - 6 .. u, then the output rows whose last
// tap it is (tap t reads C_v[ i - 11 + t ], slot ( PH + 1 + t ) % 12).
template< int PH, int EPL, typename Tout >
__device__ __forceinline__ void u64_vfir_emit( const U64Out& O,
	double ( &F )[ 12 ][ EPL ], double ( &C )[ 12 ][ EPL ], const int i,
	const double f0, const double f1, const double f2, const double f3,
	u64_cint* const cstart, u64_cdouble* const ccoef, int& jn, int& nlast,
	const int j1, const bool ok, const int e0 )
{
	// (centre F_v[ i - 3 ]; the first six steps of a chunk produce values
	// nothing reads)
#pragma unroll
	for( int e = 0; e < EPL; e++ )
	{
		double a = f0 * F[ ( PH + 9 ) % 12 ][ e ];
		a = a + f1 * ( F[ ( PH + 10 ) % 12 ][ e ] + F[ ( PH + 8 ) % 12 ][ e ]);
		a = a + f2 * ( F[ ( PH + 11 ) % 12 ][ e ] + F[ ( PH + 7 ) % 12 ][ e ]);
		a = a + f3 * ( F[ PH ][ e ] + F[ ( PH + 6 ) % 12 ][ e ]);
		C[ PH ][ e ] = a;
	}

	while( nlast == i )
	{
		u64_cdouble* const cf = ccoef + (long) jn * U64_NT;
		double sum[ EPL ];
#pragma unroll
		for( int e = 0; e < EPL; e++ )
		{
			sum[ e ] = 0.0;
		}

		dn_static_for< U64_NT >( [&]( auto tc )
		{
			constexpr int t = decltype( tc )::value;
			const double c = cf[ t ];
#pragma unroll
			for( int e = 0; e < EPL; e++ )
			{
				sum[ e ] = sum[ e ] + c * C[ ( PH + 1 + t ) % 12 ][ e ];
			}
		} );

		Tout o[ EPL ];
#pragma unroll
		for( int e = 0; e < EPL; e++ )
		{
			o[ e ] = out_stage64< Tout >( O.use_tr, O.tr_mul, O.tr_muli,
				O.pk_out, sum[ e ]);
		}

		if( ok )
		{
			Tout* const q = (Tout*) O.dst + (long) ( jn - O.dst_row0 ) *
				O.dst_ss + e0;

			if constexpr( EPL == 2 )
			{
				if( O.vec_st )
				{
					typedef Tout t2 __attribute__(( ext_vector_type( 2 )));
					t2 ov;
					ov.x = o[ 0 ];
					ov.y = o[ 1 ];

					// (non-temporal: the result is not read again, FltBuf is --
					// a 1080p source's FltBuf then stays in the last-level
					// cache: cfg2 -8 %, profiles/r06_up64/)
					__builtin_nontemporal_store( ov, (t2*) q );
				}
				else
				{
					q[ 0 ] = o[ 0 ];
					q[ 1 ] = o[ 1 ];
				}
			}
			else
			{
				__builtin_nontemporal_store( o[ 0 ], q );
			}
		}

		jn++;
		nlast = ( jn < j1 ? cstart[ jn ] + U64_NT - 1 : 0x7fffffff );
	}
}

// One step of the vertical march, ring phase PH: virtual row u = ub + PH, C / F
// index i = ibase + u.
template< int PH, int EPL, typename Tout >
__device__ __forceinline__ void u64_vstep( const U64VParams& P,
	double ( &F )[ 12 ][ EPL ], double ( &C )[ 12 ][ EPL ],
	const double* const col, const int flast, const int i,
	const double f0, const double f1, const double f2, const double f3,
	u64_cint* const cstart, u64_cdouble* const ccoef, int& jn, int& nlast,
	const int j1, const bool ok, const int e0 )
{
	// row u + PF travels into the slot of row u + PF - 12 (last read in step
	// u + PF - 6); past the chunk's end the clamped address is still a row
	u64_ld< EPL >( col + (long) u64_cl( i + U64_PF, flast ) * P.fss,
		F[ ( PH + U64_PF ) % 12 ]);

	u64_vfir_emit< PH, EPL, Tout >( P.O, F, C, i, f0, f1, f2, f3, cstart,
		ccoef, jn, nlast, j1, ok, e0 );
}

template< int EPL, typename Tout >
__global__ void __launch_bounds__( 256 ) k_uv64( const U64VParams P )
{
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int gb = ( P.ngroups + 3 ) >> 2;
	const int chunk = blockIdx.x / gb;
	const int group = ( blockIdx.x - chunk * gb ) * 4 + wave;
	const int j0 = P.O.row_lo + chunk * P.O.crows;
	const int j1 = min( j0 + P.O.crows, P.O.row_hi );

	if( group >= P.ngroups || j0 >= j1 )
	{
		return;
	}

	int e0 = ( group * 64 + lane ) * EPL;
	const bool ok = ( e0 < P.nel );

	if( !ok )
	{
		e0 = P.nel - EPL;
	}

	u64_cint* const cstart = (u64_cint*) P.O.start;
	u64_cdouble* const ccoef = (u64_cdouble*) P.O.coef;
	u64_cdouble* const cfir = (u64_cdouble*) P.O.fir;
	const double* const col = P.flt + e0;
	const int flast = P.in_len - 1;
	// virtual row u of the march is F_v / C_v index ibase + u; the first C row
	// needed is start[ j0 ], its filter reads the six F rows before it
	const int ibase = cstart[ j0 ] - 6;
	const int total = cstart[ j1 - 1 ] + U64_NT - ibase;
	const double f0 = cfir[ 0 ], f1 = cfir[ 1 ], f2 = cfir[ 2 ], f3 = cfir[ 3 ];

	double F[ 12 ][ EPL ], C[ 12 ][ EPL ];

	dn_static_for< 12 >( [&]( auto pc )
	{
		constexpr int p = decltype( pc )::value;
#pragma unroll
		for( int e = 0; e < EPL; e++ )
		{
			F[ p ][ e ] = 0.0;
			C[ p ][ e ] = 0.0;
		}
	} );

	dn_static_for< U64_PF >( [&]( auto uc )
	{
		constexpr int u = decltype( uc )::value;
		u64_ld< EPL >( col + (long) u64_cl( ibase + u, flast ) * P.fss, F[ u ]);
	} );

	int jn = j0;
	int nlast = cstart[ jn ] + U64_NT - 1; // the C row that completes output jn

	// (a step is entered only after all steps before it: the compiler's count
	// of the loads in flight at a step is the same on every path to it)
#define U64_VSTEP( PH ) u64_vstep< PH, EPL, Tout >( P, F, C, col, flast, \
	ibase + ub + PH, f0, f1, f2, f3, cstart, ccoef, jn, nlast, j1, ok, e0 ); \
	if( ub + PH + 1 >= total ) break;

	for( int ub = 0; ; ub += 12 )
	{
		U64_VSTEP( 0 ) U64_VSTEP( 1 ) U64_VSTEP( 2 ) U64_VSTEP( 3 )
		U64_VSTEP( 4 ) U64_VSTEP( 5 ) U64_VSTEP( 6 ) U64_VSTEP( 7 )
		U64_VSTEP( 8 ) U64_VSTEP( 9 ) U64_VSTEP( 10 ) U64_VSTEP( 11 )
	}

#undef U64_VSTEP
}

} // namespace

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

// tuning knobs (A/B sweeps, tools/rounds/r06_up64*.sh), read per call
static int u64_env( const char* const name, const int dflt )
{
	const char* const e = getenv( name );
	return( e != nullptr && atoi( e ) > 0 ? atoi( e ) : dflt );
}

// Whether an axis is the chain these kernels are written for (see the header).
bool up64_axis_ok( const LAxis& ax )
{
	if( ax.ops.size() != 2 )
	{
		return( false );
	}

	const LOp& f = ax.ops[ 0 ];
	const LOp& g = ax.ops[ 1 ];

	if( f.type != OP_FIR || f.view != VIEW_CLAMP || f.rf != 1 || f.lat != 3 ||
		f.e != 3 || f.out_len != f.in_len + 6 || f.in_len < 1 ||
		f.d_flt64 == nullptr )
	{
		return( false );
	}

	if( g.type != OP_GATHER || g.view != VIEW_ZS || g.in_len != f.out_len ||
		g.maxtaps != U64_NT || g.d_coef64 == nullptr || g.d_start == nullptr ||
		g.out_len < 1 || (int) g.h_start.size() != g.out_len ||
		(int) g.h_ntaps.size() != g.out_len )
	{
		return( false );
	}

	for( int j = 0; j < g.out_len; j++ )
	{
		// whole windows, ascending, below the zero tail of the stuffed view
		if( g.h_ntaps[ j ] != U64_NT ||
			( j > 0 && g.h_start[ j ] < g.h_start[ j - 1 ]) ||
			g.h_start[ j ] + U64_NT - 1 > g.zs_mmax )
		{
			return( false );
		}
	}

	return( true );
}

// LDS row pitches of the horizontal kernel for strips of 256 elements
static void up64_h_pitch( const LOp& g, const int ch, const int nel,
	int& maxSe, int& maxCe )
{
	maxCe = 0;

	for( int e0 = 0; e0 < nel; e0 += 256 )
	{
		const int jl = e0 / ch;
		const int jh = std::min( e0 + 255, nel - 1 ) / ch;
		maxCe = std::max( maxCe, ( g.h_start[ jh ] + U64_NT - g.h_start[ jl ]) *
			ch );
	}

	maxSe = maxCe + 6 * ch;
	// (odd pitches in units of 16 bytes keep the four rows in different banks)
	maxCe = ( maxCe + 1 ) | 1;
	maxSe = ( maxSe + 1 ) | 1;
}

// H pass over source rows [ya, yb] into FltBuf. 1: not taken.
int up64_run_h( const avirhip_plan* p, const void* src, const int src_type,
	const long src_ss, double* fltbuf, const int ya, const int yb,
	hipStream_t st )
{
	const LOp& f = p -> h.ops[ 0 ];
	const LOp& g = p -> h.ops[ 1 ];
	const int ch = p -> ch;

	U64HParams P;
	memset( &P, 0, sizeof( P ));
	P.src = src; P.src_ss = src_ss; P.src_w = p -> src_w;
	P.fir = f.d_flt64; P.start = g.d_start; P.coef = g.d_coef64;
	P.flt = fltbuf; P.fss = (long) p -> new_w * ch;
	P.nel = p -> new_w * ch;
	P.ya = ya; P.yb = yb;
	P.nstrips = ( P.nel + 255 ) / 256;
	up64_h_pitch( g, ch, P.nel, P.maxSe, P.maxCe );

	const int rh = ( u64_env( "AVIRHIP_UP64_RH", 8 ) == 4 ? 4 : 8 );
	const size_t lds = (size_t) rh * ( P.maxSe + P.maxCe ) * 8;

	if( lds > 64 * 1024 || P.maxSe > 512 )
	{
		return( 1 );
	}

	// ~3000 workgroups; a chunk is a whole number of steps
	const int rows = yb - ya + 1;
	const int hb = u64_env( "AVIRHIP_UP64_HB", 4096 );
	const int want = std::max( 1, hb / P.nstrips );
	int crows = ( rows + want - 1 ) / want;
	crows = std::max( 4 * rh, ( crows + rh - 1 ) / rh * rh );
	P.crows = crows;
	const int nchunks = ( rows + crows - 1 ) / crows;
	const dim3 grid( (unsigned) ( P.nstrips * nchunks ));

#define UHR( CH, T, R ) do { (void) AVIRHIP_DYN_LDS(( k_uh64< CH, T, R > ), lds ); \
	hipLaunchKernelGGL(( k_uh64< CH, T, R > ), grid, dim3( 256 ), lds, st, P ); \
	} while( 0 )
#define UH( CH, T ) do { if( rh == 4 ) UHR( CH, T, 4 ); else UHR( CH, T, 8 ); \
	} while( 0 )
#define UHT( CH ) switch( src_type ) { \
	case AVIRHIP_U8: UH( CH, uint8_t ); break; \
	case AVIRHIP_U16: UH( CH, uint16_t ); break; \
	case AVIRHIP_F32: UH( CH, float ); break; \
	default: UH( CH, double ); break; }

	switch( ch )
	{
		case 1: UHT( 1 ); break;
		case 2: UHT( 2 ); break;
		case 3: UHT( 3 ); break;
		default: UHT( 4 ); break;
	}

#undef UHT
#undef UH
#undef UHR
	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// the vertical chain's tables, the destination and its output stage
static void u64_out( const avirhip_plan* p, void* dst, const long dst_ss,
	const int row0, const int row1, U64Out& O )
{
	const LOp& f = p -> v.ops[ 0 ];
	const LOp& g = p -> v.ops[ 1 ];
	O.fir = f.d_flt64; O.start = g.d_start; O.coef = g.d_coef64;
	O.dst = dst; O.dst_ss = dst_ss; O.dst_row0 = row0;
	O.row_lo = row0; O.row_hi = row1;
	O.crows = row1 - row0; O.vec_st = 0;
	O.use_tr = ( p -> tr_mul != 1.0 ); O.tr_mul = p -> tr_mul;
	O.tr_muli = 1.0 / p -> tr_mul; O.pk_out = p -> pk_out;
}

// V pass: output rows [row0, row1) from FltBuf through the output stage into
// `dst` (row `row0` there; `dst_ss` elements of the output type per row).
int up64_run_v( const avirhip_plan* p, const double* fltbuf, void* dst,
	const int dst_type, const long dst_ss, const int row0, const int row1,
	hipStream_t st )
{
	const int ch = p -> ch;

	U64VParams P;
	memset( &P, 0, sizeof( P ));
	P.flt = fltbuf; P.fss = (long) p -> new_w * ch;
	P.nel = p -> new_w * ch;
	P.in_len = p -> src_h;
	u64_out( p, dst, dst_ss, row0, row1, P.O );

	// two elements per lane (16-byte loads) when rows hold an even number
	const int epl = (( P.nel & 1 ) == 0 &&
		u64_env( "AVIRHIP_UP64_EPL", 2 ) == 2 ? 2 : 1 );
	const size_t osz = dtype_size( dst_type );
	P.O.vec_st = ( epl == 2 && ( dst_ss & 1 ) == 0 &&
		( (uintptr_t) dst % ( 2 * osz )) == 0 ? 1 : 0 );
	P.ngroups = ( P.nel + 64 * epl - 1 ) / ( 64 * epl );

	// ~4000 waves; a chunk's run-in is 18 source rows
	const int rows = row1 - row0;
	const int vw = u64_env( "AVIRHIP_UP64_VW", 4096 );
	const int want = std::max( 1, vw / P.ngroups );
	P.O.crows = std::max( 48, ( rows + want - 1 ) / want );
	const int nchunks = ( rows + P.O.crows - 1 ) / P.O.crows;
	const dim3 grid( (unsigned) ((( P.ngroups + 3 ) / 4 ) * nchunks ));

#define UV( E, T ) hipLaunchKernelGGL(( k_uv64< E, T > ), grid, dim3( 256 ), 0, \
	st, P )
#define UVT( E ) switch( dst_type ) { \
	case AVIRHIP_U8: UV( E, uint8_t ); break; \
	case AVIRHIP_U16: UV( E, uint16_t ); break; \
	case AVIRHIP_F32: UV( E, float ); break; \
	default: UV( E, double ); break; }

	if( epl == 2 )
	{
		UVT( 2 );
	}
	else
	{
		UVT( 1 );
	}

#undef UVT
#undef UV
	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
