// tile64.hip -- LDS-tiled two-pass executor of the DOUBLE pipeline
// (avir::CImageResizer< fpclass_def< double > >, avir.h:4553-4560).
//
// generic64.hip runs such a plan one launch per lowered op with every
// intermediate in HBM (pack -> FIR -> gather -> ... -> output stage: cfg2 moved
// 1.2 GB for 0.17 GB of images, 1.28 ms). Here an axis is ONE launch:
//
//   k_h64  a workgroup owns TW outputs x TR source rows: the source segment
//          (the caller's uint8 / uint16 / float / double pixels, cast to double
//          as packScanline does, avir.h:2777-2840 -- no double copy of the
//          source) -> LDS -> the axis' chain of up to three ops, every
//          intermediate in LDS -> FltBuf (double) in HBM;
//   k_v64  a workgroup owns TW columns x TH output rows of FltBuf, the chain
//          runs down the columns; the last op stores through the output stage
//          (unpackScanline's cast, or the default ditherer's round / clamp for
//          integer pixels, avir.h:4392-4419, 3155-3215) into the caller's image.
//
// As in the float tiles (fused.hip) every LDS buffer holds the EXTENDED range
// its consumer reads: the producer fills out-of-array slots with what the
// reference's replicated prefix / suffix or the zero tail of a zero-stuffed
// buffer holds there (host-side backward range propagation per tile), so the
// tap loops have no clamps. Channels stay as they are (1-4 doubles per pixel,
// no RGBA padding). Gamma plans keep their pack pass (its double result is the
// source here) and their separate output pass.
//
// Arithmetic contract (-ffp-contract=off): v_mul_f64 / v_add_f64 apart, taps in
// the reference's order, sums from +0.0 -- the per-output expressions of
// generic64.hip, so the same bits (tests: oracle/_ref variant 4, tools/
// fuzz_f64.py compares both executors).

#include "plan.h"
#include "f64_dev.h"
#include <algorithm>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace avirhip {

namespace {

#define T64_MAXCHAIN 3

struct T64Op
{
	int type;
	int out_len;
	int zero_above; // consumer reads a zero-stuffed view: outputs above -> 0
	int rf, lat, e;
	const double* flt;
	const int* start;
	const int* ntaps;
	const double* coef;
	int maxtaps;
};

struct T64Params
{
	int nops;
	T64Op op[ T64_MAXCHAIN ];
	const int* xrng;  // [ntx][nops + 1][2]: (lo, count) of each op's input
	                  // along the chain's axis; [.][nops] = the tile's outputs
	int ntx, nty;
	int t0;           // first tile launched along the OTHER axis (k_v64: rows
	                  // are the chain's axis, see below)
	int tsz_o;        // tile size along the other axis
	int o_lo, o_hi;   // range along the other axis [lo, hi)
	const void* src; long src_ss; int src_len_o; // clamp range, other axis
	int src_len_a;    // clamp range of the first op's input along the axis
	int zero_above0;  // zero-stuffed first view
	void* dst; long dst_ss; int dst_row0;
	int a_lo, a_hi;   // k_v64: store mask on output rows [lo, hi)
	int maxA, maxB, maxC; // LDS carve: pixels, pixels, doubles
	// output stage (k_v64 storing integer / float pixels)
	int use_tr; double tr_mul, tr_muli, pk_out;
};

__device__ __forceinline__ int cl( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// (the output stage: f64_dev.h)
template< typename Tout >
__device__ __forceinline__ Tout out_stage( const T64Params& P, const double v )
{
	return( out_stage64< Tout >( P.use_tr, P.tr_mul, P.tr_muli, P.pk_out, v ));
}

// One op over the tile in LDS. The chain's axis is "a", the other one "o".
// `in`: [n_o][ipitch] pixels of CH doubles when ALONG_X (a = x, rows = o), or
// [n_a_in][pitch] with a along rows (ALONG_X false: a = y, o = x, pitch shared
// by input and output). Logical input indices from `ilo` on are present along
// a; outputs [olo, olo + on).
template< int CH, bool ALONG_X, int STORE, typename Tout >
__device__ __forceinline__ void run_op64( const T64Op& op, const double* in,
	const int ipitch, const int ilo, double* out, const int opitch,
	const int olo, const int on, const int n_o, double* cfl, int* sti,
	const T64Params& P, const int g_o_lo )
{
	const int tid = threadIdx.x;
	const int mt = op.maxtaps;
	const int olast = op.out_len - 1;

	if( op.type == OP_GATHER )
	{
		for( int i = tid; i < on; i += 256 )
		{
			const int ge = cl( olo + i, olast );
			sti[ i * 2 ] = op.start[ ge ] - ilo;
			sti[ i * 2 + 1 ] = op.ntaps[ ge ];
		}

		const int tot = on * mt;

		for( int i = tid; i < tot; i += 256 )
		{
			const int j = i / mt;
			const int t = i - j * mt;
			cfl[ i ] = op.coef[ (long) cl( olo + j, olast ) * mt + t ];
		}
	}
	else
	{
		for( int i = tid; i <= op.lat; i += 256 )
		{
			cfl[ i ] = op.flt[ i ];
		}
	}

	__syncthreads();

	// A thread computes ONE double at a time: channel k of one output pixel.
	// Lanes run over the flat elements of an image row (x * CH + k), so
	// neighbouring lanes read neighbouring doubles of LDS (conflict-free
	// ds_read_b64; a lane per whole pixel strode 32 bytes: 4-way conflicts,
	// k_v64 took 349 us for cfg2) and store contiguous runs; the four waves of
	// the workgroup take rows, TWO rows each per trip (y and y + 4): two
	// independent accumulation chains per thread hide the LDS latency that two
	// workgroups per CU cannot. LDS offsets are 32-bit.
	const int nx = ( ALONG_X ? on : n_o ) * CH;
	const int ny = ( ALONG_X ? n_o : on );
	const int istep = ( ALONG_X ? CH : ipitch * CH );
	const int tx = tid & 63, ty = tid >> 6;

	auto put = [&]( const int n, const int o, const int k, const int xe,
		const double acc )
	{
		const int g = olo + n;

		if( STORE == 1 )
		{
			// H pass -> FltBuf (double), row = o (source row), column = g
			((double*) P.dst )[ (long) ( g_o_lo + o ) * P.dst_ss +
				(long) g * CH + k ] = acc;
		}
		else
		if( STORE == 2 )
		{
			// V pass -> the caller's image: row = g, columns from g_o_lo on
			if( g >= P.a_lo && g < P.a_hi )
			{
				((Tout*) P.dst )[ (long) ( g - P.dst_row0 ) * P.dst_ss +
					(long) g_o_lo * CH + xe ] = out_stage< Tout >( P, acc );
			}
		}
		else
		{
			out[ ( ALONG_X ? o * opitch + n : n * opitch + o ) * CH + k ] =
				( g > op.zero_above ? 0.0 : acc );
		}
	};

	for( int y = ty; y < ny; y += 8 )
	{
		const int y2 = y + 4;
		const bool two = ( y2 < ny );

		for( int xe = tx; xe < nx; xe += 64 )
		{
			const int xp = xe / CH;       // pixel along x
			const int k = xe - xp * CH;   // its channel
			const int n0 = ( ALONG_X ? xp : y ), n1 = ( ALONG_X ? xp : y2 );
			const int o0 = ( ALONG_X ? y : xp ), o1 = ( ALONG_X ? y2 : xp );
			const double* const b0 = in + ( ALONG_X ? o0 * ipitch * CH + k : xe );
			const double* const b1 = in + ( ALONG_X ? o1 * ipitch * CH + k : xe );
			double a0, a1 = 0.0;

			if( op.type == OP_FIR )
			{
				const double* c0 = b0 + ( op.rf * ( cl( olo + n0, olast ) -
					op.e ) - ilo ) * istep;
				const double* c1 = b1 + ( op.rf * ( cl( olo + ( two ? n1 : n0 ),
					olast ) - op.e ) - ilo ) * istep;

				if( !two )
				{
					c1 = c0;
				}

				a0 = cfl[ 0 ] * c0[ 0 ];
				a1 = cfl[ 0 ] * c1[ 0 ];

				for( int i = 1; i <= op.lat; i++ )
				{
					const double f = cfl[ i ];
					a0 = a0 + f * ( c0[ i * istep ] + c0[ -i * istep ]);
					a1 = a1 + f * ( c1[ i * istep ] + c1[ -i * istep ]);
				}
			}
			else
			{
				const int m1 = ( two ? n1 : n0 );
				const double* c0 = b0 + sti[ n0 * 2 ] * istep;
				const double* c1 = ( two ? b1 : b0 ) + sti[ m1 * 2 ] * istep;
				const int nt0 = sti[ n0 * 2 + 1 ], nt1 = sti[ m1 * 2 + 1 ];
				const double* cf0 = cfl + n0 * mt;
				const double* cf1 = cfl + m1 * mt;
				const int ntc = min( nt0, nt1 );
				a0 = 0.0;
				int t = 0;

				for( ; t < ntc; t++ )
				{
					a0 = a0 + cf0[ t ] * c0[ 0 ];
					a1 = a1 + cf1[ t ] * c1[ 0 ];
					c0 += istep;
					c1 += istep;
				}

				for( int u = t; u < nt0; u++ )
				{
					a0 = a0 + cf0[ u ] * c0[ 0 ];
					c0 += istep;
				}

				for( int u = t; u < nt1; u++ )
				{
					a1 = a1 + cf1[ u ] * c1[ 0 ];
					c1 += istep;
				}
			}

			put( n0, o0, k, xe, a0 );

			if( two )
			{
				put( n1, o1, k, xe, a1 );
			}
		}
	}
}

// Horizontal chain. Grid: ntx tiles of outputs x row tiles of tsz_o rows.
template< int CH, typename Tin >
__global__ void __launch_bounds__( 256 ) k_h64( const T64Params P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	double* bufA = (double*) smem;
	double* bufB = bufA + (long) P.maxA * CH;
	double* cfl = bufB + (long) P.maxB * CH;
	int* sti = (int*) ( cfl + P.maxC );

	const int tile = blockIdx.x;
	const int txi = tile % P.ntx;
	const int tyi = P.t0 + tile / P.ntx;
	const int* xr = P.xrng + txi * ( P.nops + 1 ) * 2;
	const int col_lo = xr[ 0 ], cols = xr[ 1 ];
	const int row_lo = P.o_lo + tyi * P.tsz_o;
	const int rows = min( P.tsz_o, P.o_hi - row_lo );
	const int cp = cols | 1;
	const int tid = threadIdx.x;

	// the source segment, cast to double; coordinates clamp to the image, a
	// zero-stuffed first view reads 0 above its last replicated sample
	for( int i = tid; i < rows * cols; i += 256 )
	{
		const int y = i / cols;
		const int x = i - y * cols;
		const int gx = col_lo + x;
		const Tin* const q = (const Tin*) P.src + (long) ( row_lo + y ) *
			P.src_ss + (long) cl( gx, P.src_len_a - 1 ) * CH;
		const bool z = ( gx > P.zero_above0 );
#pragma unroll
		for( int k = 0; k < CH; k++ )
		{
			bufA[ ( (long) y * cp + x ) * CH + k ] = ( z ? 0.0 : (double) q[ k ]);
		}
	}

	__syncthreads();

	double* cur = bufA;
	double* oth = bufB;
	int ilo = col_lo, ipitch = cp;

	for( int s = 0; s < P.nops; s++ )
	{
		const int olo = xr[ ( s + 1 ) * 2 ];
		const int on = xr[ ( s + 1 ) * 2 + 1 ];
		const int op_ = on | 1;

		if( s == P.nops - 1 )
		{
			run_op64< CH, true, 1, double >( P.op[ s ], cur, ipitch, ilo, oth,
				op_, olo, on, rows, cfl, sti, P, row_lo );
		}
		else
		{
			run_op64< CH, true, 0, double >( P.op[ s ], cur, ipitch, ilo, oth,
				op_, olo, on, rows, cfl, sti, P, row_lo );
		}

		__syncthreads();
		double* t = cur; cur = oth; oth = t;
		ilo = olo;
		ipitch = op_;
	}
}

// Vertical chain over FltBuf (double). Grid: column tiles x row tiles.
template< int CH, typename Tout >
__global__ void __launch_bounds__( 256 ) k_v64( const T64Params P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	double* bufA = (double*) smem;
	double* bufB = bufA + (long) P.maxA * CH;
	double* cfl = bufB + (long) P.maxB * CH;
	int* sti = (int*) ( cfl + P.maxC );

	// (the chain's axis is y: xrng holds ROW ranges per row tile; the other
	// axis, x, is cut into tiles of tsz_o columns)
	const int ncx = ( P.o_hi - P.o_lo + P.tsz_o - 1 ) / P.tsz_o;
	const int tile = blockIdx.x;
	const int cxi = tile % ncx;
	const int tyi = P.t0 + tile / ncx;
	const int* yr = P.xrng + tyi * ( P.nops + 1 ) * 2;
	const int row_lo = yr[ 0 ], rows = yr[ 1 ];
	const int col_lo = P.o_lo + cxi * P.tsz_o;
	const int cols = min( P.tsz_o, P.o_hi - col_lo );
	const int pitch = cols | 1;
	const int tid = threadIdx.x;

	for( int i = tid; i < rows * cols; i += 256 )
	{
		const int y = i / cols;
		const int x = i - y * cols;
		const int gy = row_lo + y;
		const double* const q = (const double*) P.src + (long) cl( gy,
			P.src_len_a - 1 ) * P.src_ss + (long) ( col_lo + x ) * CH;
		const bool z = ( gy > P.zero_above0 );
#pragma unroll
		for( int k = 0; k < CH; k++ )
		{
			bufA[ ( (long) y * pitch + x ) * CH + k ] = ( z ? 0.0 : q[ k ]);
		}
	}

	__syncthreads();

	double* cur = bufA;
	double* oth = bufB;
	int ilo = row_lo;

	for( int s = 0; s < P.nops; s++ )
	{
		const int olo = yr[ ( s + 1 ) * 2 ];
		const int on = yr[ ( s + 1 ) * 2 + 1 ];

		if( s == P.nops - 1 )
		{
			run_op64< CH, false, 2, Tout >( P.op[ s ], cur, pitch, ilo, oth,
				pitch, olo, on, cols, cfl, sti, P, col_lo );
		}
		else
		{
			run_op64< CH, false, 0, Tout >( P.op[ s ], cur, pitch, ilo, oth,
				pitch, olo, on, cols, cfl, sti, P, col_lo );
		}

		__syncthreads();
		double* t = cur; cur = oth; oth = t;
		ilo = olo;
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct T64Geom
{
	int ta;      // tile size along the chain's axis (outputs)
	int to;      // tile size along the other axis
	int nta;     // tiles along the chain's axis
	int* d_rng;
	int maxA, maxB, maxC, maxS;
	size_t lds;
	bool ok;
};

struct T64Data
{
	T64Geom h, v;
	double* fltbuf; // [src_h][new_w] pixels of ch doubles
	bool uh, uv;    // the axis is an upsizing chain up64.hip's kernels run
	std::vector< void* > allocs;
};

// the input range outputs [a, b] of `op` touch, NOT clamped to the input array
static void need64( const LOp& op, int a, int b, int& ia, int& ib )
{
	a = std::max( 0, std::min( a, op.out_len - 1 ));
	b = std::max( 0, std::min( b, op.out_len - 1 ));

	if( op.type == OP_FIR )
	{
		ia = op.rf * ( a - op.e ) - op.lat;
		ib = op.rf * ( b - op.e ) + op.lat;
		return;
	}

	ia = 0x7fffffff;
	ib = -0x7fffffff;

	for( int j = a; j <= b; j++ )
	{
		ia = std::min( ia, op.h_start[ j ]);
		ib = std::max( ib, op.h_start[ j ] + op.h_ntaps[ j ] - 1 );
	}
}

static bool chain64_ok( const LAxis& ax )
{
	if( ax.ops.empty() || ax.ops.size() > T64_MAXCHAIN )
	{
		return( false );
	}

	for( size_t i = 0; i < ax.ops.size(); i++ )
	{
		if( ax.ops[ i ].type == OP_UPF || ax.ops[ i ].view == VIEW_RAW )
		{
			return( false );
		}
	}

	return( true );
}

// Range tables for tiles of `ta` outputs; picks nothing, only measures.
static void ranges64( const LAxis& ax, int out_len, int ta,
	std::vector< int >& rng, int& ntiles, std::vector< int >& maxn )
{
	const int n = (int) ax.ops.size();
	ntiles = ( out_len + ta - 1 ) / ta;
	rng.assign( (size_t) ntiles * ( n + 1 ) * 2, 0 );
	maxn.assign( n + 1, 0 );

	for( int t = 0; t < ntiles; t++ )
	{
		int a = t * ta;
		int b = std::min( out_len, a + ta ) - 1;
		int* r = &rng[ (size_t) t * ( n + 1 ) * 2 ];
		r[ n * 2 ] = a;
		r[ n * 2 + 1 ] = b - a + 1;

		for( int s = n - 1; s >= 0; s-- )
		{
			int ia, ib;
			need64( ax.ops[ s ], a, b, ia, ib );
			r[ s * 2 ] = ia;
			r[ s * 2 + 1 ] = ib - ia + 1;
			a = ia;
			b = ib;
		}

		for( int s = 0; s <= n; s++ )
		{
			maxn[ s ] = std::max( maxn[ s ], r[ s * 2 + 1 ]);
		}
	}
}

static int geom64( T64Data* D, const LAxis& ax, int out_len, int ch,
	const int ( *cand )[ 2 ], int ncand, size_t cap, T64Geom& G, bool along_x )
{
	G.ok = false;
	const int n = (int) ax.ops.size();

	for( int c = 0; c < ncand && !G.ok; c++ )
	{
		std::vector< int > rng, mx;
		int nt;
		ranges64( ax, out_len, cand[ c ][ 0 ], rng, nt, mx );
		const int to = cand[ c ][ 1 ];
		// stage regions alternate A, B, A ...: source | op outputs (the last
		// one leaves for global memory)
		long mA = 0, mB = 1;

		for( int s = 0; s < n; s++ )
		{
			// H: `to` rows of ( count | 1 ) pixels; V: count rows of ( to | 1 )
			const long px = ( along_x ? (long) ( mx[ s ] | 1 ) * to :
				(long) mx[ s ] * ( to | 1 ));
			if( s & 1 ) mB = std::max( mB, px ); else mA = std::max( mA, px );
		}

		long mC = 16, mS = 2;

		for( int s = 0; s < n; s++ )
		{
			const LOp& op = ax.ops[ s ];

			if( op.type == OP_FIR )
			{
				mC = std::max( mC, (long) op.lat + 1 );
			}
			else
			{
				mC = std::max( mC, (long) mx[ s + 1 ] * op.maxtaps );
				mS = std::max( mS, (long) mx[ s + 1 ] * 2 );
			}
		}

		const size_t lds = (size_t) ( mA + mB ) * ch * 8 + (size_t) mC * 8 +
			(size_t) mS * 4 + 64;

		if( lds > cap )
		{
			continue;
		}

		void* q = nullptr;

		if( hipMalloc( &q, rng.size() * sizeof( int )) != hipSuccess )
		{
			(void) hipGetLastError();
			return( AVIRHIP_EHIP );
		}

		D -> allocs.push_back( q );

		if( hipMemcpy( q, rng.data(), rng.size() * sizeof( int ),
			hipMemcpyHostToDevice ) != hipSuccess )
		{
			(void) hipGetLastError();
			return( AVIRHIP_EHIP );
		}

		G.ta = cand[ c ][ 0 ]; G.to = to; G.nta = nt; G.d_rng = (int*) q;
		G.maxA = (int) mA; G.maxB = (int) mB; G.maxC = (int) mC;
		G.maxS = (int) mS; G.lds = lds; G.ok = true;
	}

	return( AVIRHIP_OK );
}

static void fill64( const LAxis& ax, T64Params& P )
{
	const int n = (int) ax.ops.size();
	P.nops = n;

	for( int i = 0; i < n; i++ )
	{
		const LOp& o = ax.ops[ i ];
		T64Op& t = P.op[ i ];
		t.type = o.type; t.out_len = o.out_len;
		t.zero_above = ( i + 1 < n && ax.ops[ i + 1 ].view == VIEW_ZS ?
			ax.ops[ i + 1 ].zs_mmax : 0x7fffffff );
		t.rf = o.rf; t.lat = o.lat; t.e = o.e;
		t.flt = o.d_flt64; t.start = o.d_start; t.ntaps = o.d_ntaps;
		t.coef = o.d_coef64; t.maxtaps = o.maxtaps;
	}

	P.zero_above0 = ( ax.ops[ 0 ].view == VIEW_ZS ? ax.ops[ 0 ].zs_mmax :
		0x7fffffff );
}

} // namespace

int tile64_prepare( avirhip_plan* p )
{
	p -> tile64 = nullptr;

	if( !p -> f64 || p -> is_lancir || getenv( "AVIRHIP_NO_TILE64" ) != nullptr ||
		!chain64_ok( p -> h ) || !chain64_ok( p -> v ))
	{
		return( AVIRHIP_OK );
	}

	T64Data* D = new T64Data();
	D -> fltbuf = nullptr;
	D -> h.ok = D -> v.ok = false;
	D -> uh = up64_axis_ok( p -> h );
	D -> uv = up64_axis_ok( p -> v );
	p -> tile64 = D;
	// (tile along the chain's axis, tile along the other one)
	static const int ch_[][ 2 ] = { { 64, 8 }, { 128, 4 }, { 64, 4 }, { 32, 4 },
		{ 32, 2 }, { 16, 2 }, { 8, 1 } };
	static const int cv_[][ 2 ] = { { 64, 16 }, { 48, 16 }, { 32, 32 }, { 32, 16 },
		{ 16, 32 }, { 16, 16 }, { 8, 16 }, { 8, 8 }, { 4, 8 }, { 2, 8 } };
	const size_t cap = 64 * 1024;
	int rc = geom64( D, p -> h, p -> new_w, p -> ch, ch_, 7, cap, D -> h, true );
	if( rc == 0 ) rc = geom64( D, p -> v, p -> new_h, p -> ch, cv_, 10, cap,
		D -> v, false );

	if( rc != 0 || !D -> h.ok || !D -> v.ok )
	{
		tile64_release( p );
		return( rc );
	}

	return( AVIRHIP_OK );
}

void tile64_release( avirhip_plan* p )
{
	T64Data* D = (T64Data*) p -> tile64;

	if( D != nullptr )
	{
		for( size_t i = 0; i < D -> allocs.size(); i++ )
		{
			(void) hipFree( D -> allocs[ i ]);
		}

		delete D;
	}

	p -> tile64 = nullptr;
}

bool tile64_ok( const avirhip_plan* p )
{
	return( p -> tile64 != nullptr );
}

template< int CH >
static void launch_h64( const T64Params& P, int in_type, int ntiles, size_t lds,
	hipStream_t st )
{
#define TH( T ) do { (void) AVIRHIP_DYN_LDS(( k_h64< CH, T > ), lds ); \
	hipLaunchKernelGGL(( k_h64< CH, T > ), dim3( ntiles ), dim3( 256 ), lds, st, \
	P ); } while( 0 )

	switch( in_type )
	{
		case AVIRHIP_U8: TH( uint8_t ); break;
		case AVIRHIP_U16: TH( uint16_t ); break;
		case AVIRHIP_F32: TH( float ); break;
		default: TH( double ); break;
	}
#undef TH
}

template< int CH >
static void launch_v64( const T64Params& P, int out_type, int ntiles,
	size_t lds, hipStream_t st )
{
#define TV( T ) do { (void) AVIRHIP_DYN_LDS(( k_v64< CH, T > ), lds ); \
	hipLaunchKernelGGL(( k_v64< CH, T > ), dim3( ntiles ), dim3( 256 ), lds, st, \
	P ); } while( 0 )

	switch( out_type )
	{
		case AVIRHIP_U8: TV( uint8_t ); break;
		case AVIRHIP_U16: TV( uint16_t ); break;
		case AVIRHIP_F32: TV( float ); break;
		default: TV( double ); break;
	}
#undef TV
}

// Output rows [row0, row1). `src` / `src_type` / `src_ss`: the image the H pass
// reads (the caller's, or the plan's double copy after a gamma pack pass);
// `dst` / `dst_type` / `dst_ss`: where the V pass stores row `row0` (the
// caller's band, through the output stage; or the plan's double result, type
// AVIRHIP_F64, when a gamma output pass follows). Source rows [ya, yb] are the
// ones the band's vertical windows read.
int tile64_run( avirhip_plan* p, const void* src, int src_type, long src_ss,
	void* dst, int dst_type, long dst_ss, int row0, int row1, int ya, int yb,
	hipStream_t st )
{
	T64Data* D = (T64Data*) p -> tile64;

	if( D == nullptr )
	{
		return( 1 );
	}

	const int ch = p -> ch;

	if( D -> fltbuf == nullptr )
	{
		void* q = nullptr;
		const size_t bytes = (size_t) p -> src_h * p -> new_w * ch * 8;
		AVIRHIP_HIPCHECK( hipMalloc( &q, bytes ));
		D -> allocs.push_back( q );
		p -> alloc_bytes += bytes;
		D -> fltbuf = (double*) q;
	}

	T64Params P;
	memset( &P, 0, sizeof( P ));

	// upsizing chains: the marching kernels of up64.hip (AVIRHIP_NO_UP64: the
	// tiles below, for A/B timing and the differential tests)
	const bool no_up64 = ( getenv( "AVIRHIP_NO_UP64" ) != nullptr );
	bool hdone = false, vdone = false;

	if( D -> uh && !no_up64 )
	{
		const int rc = up64_run_h( p, src, src_type, src_ss, D -> fltbuf, ya, yb,
			st );

		if( rc < 0 )
		{
			return( rc );
		}

		hdone = ( rc == 0 );
	}

	// ---- H chain over source rows [ya, yb]
	fill64( p -> h, P );
	P.xrng = D -> h.d_rng; P.ntx = D -> h.nta;
	P.tsz_o = D -> h.to; P.t0 = 0;
	P.o_lo = ya; P.o_hi = yb + 1;
	P.src = src; P.src_ss = src_ss;
	P.src_len_a = p -> src_w; P.src_len_o = p -> src_h;
	P.dst = D -> fltbuf; P.dst_ss = (long) p -> new_w * ch; P.dst_row0 = 0;
	P.maxA = D -> h.maxA; P.maxB = D -> h.maxB; P.maxC = D -> h.maxC;

	if( !hdone )
	{
		const int nty = ( yb + 1 - ya + P.tsz_o - 1 ) / P.tsz_o;
		const int nt = P.ntx * nty;

		switch( ch )
		{
			case 1: launch_h64< 1 >( P, src_type, nt, D -> h.lds, st ); break;
			case 2: launch_h64< 2 >( P, src_type, nt, D -> h.lds, st ); break;
			case 3: launch_h64< 3 >( P, src_type, nt, D -> h.lds, st ); break;
			default: launch_h64< 4 >( P, src_type, nt, D -> h.lds, st ); break;
		}
	}

	if( D -> uv && !no_up64 )
	{
		const int rc = up64_run_v( p, D -> fltbuf, dst, dst_type, dst_ss, row0,
			row1, st );

		if( rc < 0 )
		{
			return( rc );
		}

		vdone = ( rc == 0 );
	}

	if( vdone )
	{
		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	// ---- V chain over FltBuf columns, output rows [row0, row1)
	fill64( p -> v, P );
	P.xrng = D -> v.d_rng; P.nty = D -> v.nta;
	P.tsz_o = D -> v.to;
	P.t0 = row0 / D -> v.ta;
	P.o_lo = 0; P.o_hi = p -> new_w;
	P.src = D -> fltbuf; P.src_ss = (long) p -> new_w * ch;
	P.src_len_a = p -> src_h; P.src_len_o = p -> new_w;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = row0;
	P.a_lo = row0; P.a_hi = row1;
	P.maxA = D -> v.maxA; P.maxB = D -> v.maxB; P.maxC = D -> v.maxC;
	P.use_tr = ( p -> tr_mul != 1.0 ); P.tr_mul = p -> tr_mul;
	P.tr_muli = 1.0 / p -> tr_mul; P.pk_out = p -> pk_out;
	{
		const int t1 = ( row1 - 1 ) / D -> v.ta;
		const int ncx = ( p -> new_w + P.tsz_o - 1 ) / P.tsz_o;
		const int nt = ncx * ( t1 - P.t0 + 1 );

		switch( ch )
		{
			case 1: launch_v64< 1 >( P, dst_type, nt, D -> v.lds, st ); break;
			case 2: launch_v64< 2 >( P, dst_type, nt, D -> v.lds, st ); break;
			case 3: launch_v64< 3 >( P, dst_type, nt, D -> v.lds, st ); break;
			default: launch_v64< 4 >( P, dst_type, nt, D -> v.lds, st ); break;
		}
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
