// fused.hip -- LDS-tiled executor for RGBA (4-channel) float plans.
//
// One kernel, k_tile, runs a chain of lowered ops over a 2-D output tile with
// every intermediate kept in LDS:
//
//   path 3 "fused"    H chain + V chain in one launch: source tile (with its
//                     halo) -> LDS -> ... -> destination tile. The horizontally
//                     resized intermediate image (the reference's FltBuf,
//                     avir.h:4881) never exists in HBM; traffic ~= the
//                     algorithmic bytes. Chosen for upsizing, where FltBuf is
//                     large and the vertical support is short.
//   path 2 "two-pass" the same kernel launched twice (H chain only into
//                     FltBuf, then V chain only). Chosen for downsizing, where
//                     FltBuf is small and a fused tile's halo would be huge.
//
// gfx950 mapping: a workgroup is 256 threads = 4 wave64. The source tile and
// the tile's block of coefficient rows (tables re-laid with 16-byte rows) are
// fetched by LDS-DMA (buffer_load_dwordx4 ... lds), every piece in flight at
// once; uint8 / uint16 sources are converted and padded to RGBA by the loader
// itself. Every LDS buffer holds the *extended* range its consumer reads (the
// producer fills the replicated / zero-stuffed slots), so the tap loops have
// no clamps. Lanes run along the image x axis -- contiguous 16-byte global
// accesses, conflict-free ds_read_b128 -- except in a downsizing gather along
// x, where the 16 lanes of an LDS access group take 16 rows at one x (see
// run_op). Tiles are dealt to workgroups so that consecutive tiles (which
// share halo rows/columns) land on the same XCD and hit in its private L2.
//
// Arithmetic contract: -ffp-contract=off, separate mul/add in the reference's
// tap order, sums started from +0.0f (see generic.hip) => bit-identical.

#include "gpass_dev.h"
#include <algorithm>
#include <string.h>
#include <stdio.h>

namespace avirhip {

typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef __attribute__(( address_space( 3 ))) void* tile_ldsptr;

#define MAXCHAIN 3

struct TOp
{
	int type;
	int rowfast;    // ALONG_X gather with a lane stride > 1 px (see run_op)
	int out_len;    // logical output length (producer-side clamp range)
	int zero_above; // consumer is a zero-stuffed view: outputs with a logical
	                // index above this are stored as 0 (INT_MAX: none)
	int rf, lat, e;
	const float* flt;
	const int* start;
	const int* ntaps;
	const float* coef;
	int maxtaps;
};

struct TileParams
{
	int nh, nv;
	TOp hop[ MAXCHAIN ];
	TOp vop[ MAXCHAIN ];
	const int* hrng; // [ntx][nh+1][2] (lo, count); [.][nh] = tile columns
	const int* vrng; // [nty][nv+1][2]
	int ntx, nty, ty0; // ty0: first tile row launched (bands)
	const void* src; long src_ss;  // elements per source row
	int src_type, src_ch;          // AVIRHIP_F32/4: float RGBA; U8/U16: raw image
	int src_w, src_h;              // extent of `src` (clamp range of the load)
	int xzero_above, yzero_above;  // zero-stuffed first views (INT_MAX: none)
	float* dst; long dst_ss;       // floats per destination row
	int dst_row0;                  // row index stored at dst
	int srow_lo, srow_hi;          // store mask on the final-axis index
	int maxA, maxB, maxC;          // LDS carve (pixels, pixels, floats)
#ifdef U2_DBG
	unsigned long long* clk;       // [blocks][8] phase timestamps (debug build)
#endif
};

// Every buffer in LDS holds the *extended* index range its consumer touches:
// slots whose logical index lies outside the producing array are filled by
// the producer with the value the reference's replicated prefix / suffix (or
// the zero tail of a zero-stuffed buffer) would hold there. The consumer's
// tap loops therefore carry no clamps, no zero tests and no view logic at
// all -- index = start + tap.
__device__ __forceinline__ int clampi( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// One op over a tile held in LDS. `in` is [irows][icols] pixels; the op runs
// along x (ALONG_X) or along y. Logical input indices from `ilo` on are
// present along the op's axis. Output logical indices [olo, olo+on).
template< bool ALONG_X, bool TO_GLOBAL >
__device__ __forceinline__ void run_op( const TOp& op, const f4* in,
	const int ipitch, const int ilo, f4* out, const int opitch,
	const int olo, const int on, const int other_n, float* cfl, int* sti,
	const TileParams& P, const int g_other_lo )
{
	const int tx = threadIdx.x; // 0..31
	const int ty = threadIdx.y; // 0..7
	const int mt = ( op.maxtaps + 3 ) & ~3;
	const int olast = op.out_len - 1;

	if( op.type == OP_GATHER )
	{
		// stage (start - ilo, ntaps) and the coefficient rows of this tile's
		// outputs; an output outside the array is its clamped neighbour
		const int tid = ty * 32 + tx;

		for( int i = tid; i < on; i += 256 )
		{
			const int ge = clampi( olo + i, olast );
			sti[ i * 2 ] = op.start[ ge ] - ilo;
			sti[ i * 2 + 1 ] = op.ntaps[ ge ];
		}

		const int tot = on * mt;

		if( olo >= 0 && olo + on <= op.out_len )
		{
			// interior tile: its coefficient rows are one contiguous block of
			// the (16-byte-row) table -> LDS-DMA, 1 KiB per wave-instruction,
			// all in flight at once
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int lane = tid & 63;
			const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
				(void*) ( op.coef + (long) olo * mt ), 0, 0x7fffffff,
				0x00020000 );

			for( int q0 = wave * 64; q0 * 4 < tot; q0 += 256 )
			{
				if(( q0 + lane ) * 4 < tot )
				{
					__builtin_amdgcn_raw_ptr_buffer_load_lds( rc,
						(tile_ldsptr) ( cfl + q0 * 4 ), 16, ( q0 + lane ) * 16,
						0, 0, 0 );
				}
			}

			AVIRHIP_WAIT_VM( 0 );
		}
		else
		{
			// edge tile: rows clamp individually
			for( int i = tid; i < tot; i += 256 )
			{
				const int j = i / mt;
				const int t = i - j * mt;
				const int ge = clampi( olo + j, olast );
				cfl[ i ] = op.coef[ (long) ge * mt + t ];
			}
		}

		__syncthreads();
	}
	else
	{
		// FIR taps -> LDS once (read from global memory inside the tap loop
		// they cost a dependent vector load per tap and output)
		const int tid = ty * 32 + tx;

		// (deep decimation steps have hundreds of taps: 204 -> 1 pixels is a
		// 771-tap filter, lat = 385 -- more than one tap per thread)
		for( int i = tid; i <= op.lat; i += 256 )
		{
			cfl[ i ] = op.flt[ i ];
		}

		__syncthreads();
	}

	// loops: x over lanes, y over thread rows
	const int nx = ( ALONG_X ? on : other_n );
	const int ny = ( ALONG_X ? other_n : on );
	const int istep = ( ALONG_X ? 1 : ipitch );

	// Thread -> output mapping. Default: lanes run along x (32 wide), which
	// is conflict-free when neighbouring outputs read neighbouring inputs.
	// A downsizing gather along x reads with a stride of k > 1 pixels between
	// lanes and its 128-bit LDS reads collide k-fold; there (op.rowfast) the
	// 16 lanes of an LDS access group take 16 ROWS at the same x instead:
	// with odd row pitches every group hits 16 distinct bank quads, and the
	// coefficient read of a group is a broadcast.
	const bool rowfast = ( ALONG_X && !TO_GLOBAL && op.rowfast != 0 );
	const int tid_ = ty * 32 + tx;
	const int ybeg = ( rowfast ? ( tid_ & 15 ) : ty );
	const int ystep = ( rowfast ? 16 : 8 );
	const int xbeg = ( rowfast ? ( tid_ >> 4 ) : tx );
	const int xstep = ( rowfast ? 16 : 32 );

	for( int y = ybeg; y < ny; y += ystep )
	{
		for( int x = xbeg; x < nx; x += xstep )
		{
			const int n = ( ALONG_X ? x : y );     // output index in tile
			const int o = ( ALONG_X ? y : x );     // position on other axis
			const int g = olo + n;                 // logical output index
			const f4* base = ( ALONG_X ? in + o * ipitch : in + o );
			f4 acc;

			if( op.type == OP_FIR )
			{
				const int cp = op.rf * ( clampi( g, olast ) - op.e ) - ilo;
				const f4* c = base + cp * istep;
				acc = cfl[ 0 ] * c[ 0 ];

				for( int i = 1; i <= op.lat; i++ )
				{
					const f4 s = c[ i * istep ] + c[ -i * istep ];
					acc = acc + cfl[ i ] * s;
				}
			}
			else
			{
				const f4* c = base + sti[ n * 2 ] * istep;
				const int nt = sti[ n * 2 + 1 ];
				const float* cf = cfl + n * mt;
				acc = (f4) 0.0f;
				int t = 0;

				for( ; t + 4 <= nt; t += 4 )
				{
					const f4 c4 = *(const f4*) ( cf + t );
					const f4 v0 = c[ 0 ];
					const f4 v1 = c[ istep ];
					const f4 v2 = c[ 2 * istep ];
					const f4 v3 = c[ 3 * istep ];
					acc = acc + c4.x * v0;
					acc = acc + c4.y * v1;
					acc = acc + c4.z * v2;
					acc = acc + c4.w * v3;
					c += 4 * istep;
				}

				if( t < nt )
				{
					// 1..3 remaining taps (the padded coefficients are never
					// multiplied: x * 0 is not a no-op for non-finite x)
					const f4 c4 = *(const f4*) ( cf + t );
					acc = acc + c4.x * c[ 0 ];

					if( t + 1 < nt )
					{
						acc = acc + c4.y * c[ istep ];

						if( t + 2 < nt )
						{
							acc = acc + c4.z * c[ 2 * istep ];
						}
					}
				}
			}

			if( TO_GLOBAL )
			{
				// final op of the launch: x is always image x
				const int gx = ( ALONG_X ? g : g_other_lo + o );
				const int gy = ( ALONG_X ? g_other_lo + o : g );
				const int gm = ( ALONG_X ? gy : g ); // masked (row) index

				if( P.nv == 0 || ( gm >= P.srow_lo && gm < P.srow_hi ))
				{
					*(f4*) ( P.dst + (long) ( gy - P.dst_row0 ) * P.dst_ss +
						(long) gx * 4 ) = acc;
				}
			}
			else
			{
				if( g > op.zero_above )
				{
					acc = (f4) 0.0f;
				}

				out[ ( ALONG_X ? o * opitch + n : n * opitch + o ) ] = acc;
			}
		}
	}
}

// Tile load straight from the caller's integer image: the (float) cast and
// the zero padding to 4 channels of the pack stage (packScanline,
// avir.h:2777-2840) happen here, so no float copy of the source ever exists.
// Four 64-pixel pieces per trip: their loads are in flight together.
template< typename Tin >
__device__ __forceinline__ void load_raw_tile( const TileParams& P, f4* bufA,
	const int cp, const int row_lo, const int rows, const int col_lo,
	const int cols, const int wave, const int lane )
{
	const int ch = P.src_ch;
	// A pixel is ONE unaligned load of four elements, shifted down where it
	// had to start early (the image's last pixel): elements [0, total)
	const long total = (long) ( P.src_h - 1 ) * P.src_ss + (long) P.src_w * ch;

	for( int y = wave; y < rows; y += 4 )
	{
		const int gy = clampi( row_lo + y, P.src_h - 1 );
		const long row_e = (long) gy * P.src_ss;

		for( int x0 = 0; x0 < cols; x0 += 256 )
		{
			unsigned long long v[ 4 ];
#pragma unroll
			for( int u = 0; u < 4; u++ )
			{
				const int x = x0 + u * 64 + lane;
				const long e = row_e + (long) clampi( col_lo + x, P.src_w - 1 ) *
					ch;
				const long ea = ( e < total - 4 ? e : total - 4 );
				const int sh = (int) ( e - ea );

				if( sizeof( Tin ) == 1 )
				{
					typedef unsigned int u32u __attribute__(( aligned( 1 )));
					v[ u ] = *(const u32u*) ( (const Tin*) P.src + ea ) >>
						( 8 * sh );
				}
				else
				{
					typedef unsigned long long u64u __attribute__(( aligned( 2 )));
					v[ u ] = *(const u64u*) ( (const Tin*) P.src + ea ) >>
						( 16 * sh );
				}
			}
#pragma unroll
			for( int u = 0; u < 4; u++ )
			{
				const int x = x0 + u * 64 + lane;

				if( x < cols )
				{
					float w[ 4 ];
#pragma unroll
					for( int c = 0; c < 4; c++ )
					{
						const unsigned int q = (unsigned int) ( sizeof( Tin ) == 1 ?
							( v[ u ] >> ( 8 * c )) & 0xffull :
							( v[ u ] >> ( 16 * c )) & 0xffffull );

						w[ c ] = ( c < ch ? (float) q : 0.0f );
					}

					f4 o;
					o.x = w[ 0 ]; o.y = w[ 1 ]; o.z = w[ 2 ]; o.w = w[ 3 ];
					bufA[ y * cp + x ] = o;
				}
			}
		}
	}
}

__global__ void __launch_bounds__( 256 ) k_tile( const TileParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	f4* bufA = (f4*) smem;
	f4* bufB = bufA + P.maxA;
	float* cfl = (float*) ( bufB + P.maxB );
	int* sti = (int*) ( cfl + P.maxC );

	// XCD-aware tile dealing: workgroup b runs on XCD b % 8 (observed);
	// give each XCD a contiguous run of tiles so halo re-reads hit its L2.
	const int nwg = gridDim.x;
	const int b = blockIdx.x;
	const int xcd = b & 7;
	const int qd = nwg >> 3;
	const int rm = nwg & 7;
	const int tile = ( xcd < rm ? xcd * ( qd + 1 ) :
		rm * ( qd + 1 ) + ( xcd - rm ) * qd ) + ( b >> 3 );

	const int txi = tile % P.ntx;
	const int tyi = P.ty0 + tile / P.ntx;
	const int* hr = P.hrng + txi * ( P.nh + 1 ) * 2;
	const int* vr = P.vrng + tyi * ( P.nv + 1 ) * 2;

	const int col_lo = hr[ 0 ], cols = hr[ 1 ];
	const int row_lo = vr[ 0 ], rows = vr[ 1 ];
	const int tx = threadIdx.x;
	const int ty = threadIdx.y;
	const int cp = cols | 1; // row pitches in LDS are odd (see run_op)
#ifdef U2_DBG
	int dbg_n = 0;
#define TILE_STAMP() do { __syncthreads(); if( tx == 0 && ty == 0 && \
	P.clk != nullptr && dbg_n < 8 ) P.clk[ blockIdx.x * 8 + dbg_n++ ] = \
	wall_clock64(); } while( 0 )
#else
#define TILE_STAMP() do {} while( 0 )
#endif
	TILE_STAMP();

	// source tile -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
	// registers, every piece of the tile in flight at once), extended range:
	// coordinates clamp to the image. One instruction moves 64 consecutive
	// pixels of a row; rows are dealt to the four waves.
	{
		const int tid = ty * 32 + tx;
		const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
		const int lane = tid & 63;

		if( P.src_type == AVIRHIP_F32 )
		{
			for( int y = wave; y < rows; y += 4 )
			{
				const int gy = clampi( row_lo + y, P.src_h - 1 );
				const __amdgpu_buffer_rsrc_t rs =
					__builtin_amdgcn_make_buffer_rsrc( (void*) (
					(const float*) P.src + (long) gy * P.src_ss ), 0,
					0x7fffffff, 0x00020000 );

				for( int x0 = 0; x0 < cols; x0 += 64 )
				{
					if( x0 + lane < cols )
					{
						const int gx = clampi( col_lo + x0 + lane,
							P.src_w - 1 );
						__builtin_amdgcn_raw_ptr_buffer_load_lds( rs,
							(tile_ldsptr) ( bufA + y * cp + x0 ), 16, gx * 16,
							0, 0, 0 );
					}
				}
			}
		}
		else
		if( P.src_type == AVIRHIP_U8 )
		{
			load_raw_tile< uint8_t >( P, bufA, cp, row_lo, rows, col_lo, cols,
				wave, lane );
		}
		else
		{
			load_raw_tile< uint16_t >( P, bufA, cp, row_lo, rows, col_lo,
				cols, wave, lane );
		}

		AVIRHIP_WAIT_VM( 0 );
		__syncthreads();

		// zero-stuffed first views: their tail reads 0 (rare: the tile must
		// reach beyond the last replicated sample)
		if( col_lo + cols - 1 > P.xzero_above ||
			row_lo + rows - 1 > P.yzero_above )
		{
			for( int y = ty; y < rows; y += 8 )
			{
				for( int x = tx; x < cols; x += 32 )
				{
					if( col_lo + x > P.xzero_above ||
						row_lo + y > P.yzero_above )
					{
						bufA[ y * cp + x ] = (f4) 0.0f;
					}
				}
			}
		}
	}

	__syncthreads();
	TILE_STAMP();

	f4* cur = bufA;
	f4* oth = bufB;
	int ilo = col_lo;
	int ipitch = cp;

	for( int s = 0; s < P.nh; s++ )
	{
		const int olo = hr[ ( s + 1 ) * 2 ];
		const int on = hr[ ( s + 1 ) * 2 + 1 ];
		const int op_ = on | 1;

		if( s == P.nh - 1 && P.nv == 0 )
		{
			run_op< true, true >( P.hop[ s ], cur, ipitch, ilo, oth, op_, olo,
				on, rows, cfl, sti, P, row_lo );
		}
		else
		{
			run_op< true, false >( P.hop[ s ], cur, ipitch, ilo, oth, op_, olo,
				on, rows, cfl, sti, P, row_lo );
		}

		__syncthreads();
		TILE_STAMP();
		f4* t = cur; cur = oth; oth = t;
		ilo = olo;
		ipitch = op_;
	}

	// after the H chain the tile is [rows][tcols], row pitch ipitch
	const int tcol_lo = ( P.nh > 0 ? hr[ P.nh * 2 ] : col_lo );
	const int tcols = ( P.nh > 0 ? hr[ P.nh * 2 + 1 ] : cols );
	ilo = row_lo;

	for( int s = 0; s < P.nv; s++ )
	{
		const int olo = vr[ ( s + 1 ) * 2 ];
		const int on = vr[ ( s + 1 ) * 2 + 1 ];

		if( s == P.nv - 1 )
		{
			run_op< false, true >( P.vop[ s ], cur, ipitch, ilo, oth, ipitch,
				olo, on, tcols, cfl, sti, P, tcol_lo );
		}
		else
		{
			run_op< false, false >( P.vop[ s ], cur, ipitch, ilo, oth, ipitch,
				olo, on, tcols, cfl, sti, P, tcol_lo );
		}

		__syncthreads();
		TILE_STAMP();
		f4* t = cur; cur = oth; oth = t;
		ilo = olo;
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct TileGeom
{
	int tw, th, ntx, nty;
	int* d_hrng;
	int* d_vrng;
	int maxA, maxB, maxC;
	size_t lds;
	bool ok;
};

struct FusedData
{
	TileGeom fused; // path 3
	TileGeom hpass; // path 2, pass 1 (H chain only)
	TileGeom vpass; // path 2, pass 2 (V chain only)
	float* fltbuf;  // path 2 intermediate [src_h][new_w] pixels
	void* dn;       // dn.hip: specialised passes of path 2 (integer downsizing)
	// coefficient tables of the GATHER ops re-laid with a row stride of
	// (maxtaps+3)&~3 floats: 16-byte rows, so a tile's block of rows is one
	// contiguous LDS-DMA transfer
	const float* hpad[ MAXCHAIN ];
	const float* vpad[ MAXCHAIN ];
	std::vector< void* > allocs;
};

// Input index range touched by outputs [a, b] of `op` (a, b may lie outside
// the array: such outputs are copies of the clamped edge output). The result
// is NOT clamped to the input array: the producer of that buffer fills the
// out-of-array slots (see run_op).
static void op_need( const LOp& op, int a, int b, int& ia, int& ib )
{
	a = std::max( 0, std::min( a, op.out_len - 1 ));
	b = std::max( 0, std::min( b, op.out_len - 1 ));

	if( op.type == OP_FIR )
	{
		ia = op.rf * ( a - op.e ) - op.lat;
		ib = op.rf * ( b - op.e ) + op.lat;
	}
	else
	{
		ia = 0x7fffffff;
		ib = -0x7fffffff;

		for( int j = a; j <= b; j++ )
		{
			ia = std::min( ia, op.h_start[ j ]);
			ib = std::max( ib, op.h_start[ j ] + op.h_ntaps[ j ] - 1 );
		}
	}
}

// Range table of one axis for tiles of `tsz` outputs: rng[t][s] = (lo, count)
// of the input of op s; rng[t][n] = the tile's outputs.
static void build_ranges( const LAxis& ax, int nops, int out_len, int tsz,
	std::vector< int >& rng, int& ntiles, std::vector< int >& maxn )
{
	ntiles = ( out_len + tsz - 1 ) / tsz;
	rng.assign( (size_t) ntiles * ( nops + 1 ) * 2, 0 );
	maxn.assign( nops + 1, 0 );

	for( int t = 0; t < ntiles; t++ )
	{
		int a = t * tsz;
		int b = std::min( out_len, a + tsz ) - 1;
		int* r = &rng[ (size_t) t * ( nops + 1 ) * 2 ];
		r[ nops * 2 ] = a;
		r[ nops * 2 + 1 ] = b - a + 1;

		for( int s = nops - 1; s >= 0; s-- )
		{
			int ia, ib;
			op_need( ax.ops[ s ], a, b, ia, ib );
			r[ s * 2 ] = ia;
			r[ s * 2 + 1 ] = ib - ia + 1;
			a = ia;
			b = ib;
		}

		for( int s = 0; s <= nops; s++ )
		{
			maxn[ s ] = std::max( maxn[ s ], r[ s * 2 + 1 ]);
		}
	}
}

static bool chain_ok( const LAxis& ax )
{
	if( ax.ops.empty() || ax.ops.size() > MAXCHAIN )
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

// Builds geometry for running `useh` / `usev` chains with tile (tw, th).
static int make_geom( avirhip_plan* p, FusedData* F, bool useh, bool usev,
	int tw, int th, size_t lds_cap, TileGeom& G )
{
	G.ok = false;
	G.tw = tw;
	G.th = th;
	const int nh = ( useh ? (int) p -> h.ops.size() : 0 );
	const int nv = ( usev ? (int) p -> v.ops.size() : 0 );
	std::vector< int > hr, vr, hm, vm;

	// x extent: outputs of the H chain (new_w) when it runs, or when only the
	// V chain runs (its columns are FltBuf columns = new_w as well)
	build_ranges( p -> h, nh, p -> new_w, tw, hr, G.ntx, hm );
	// y extent: V chain outputs (new_h), or source rows for an H-only pass
	build_ranges( p -> v, nv, ( usev ? p -> new_h : p -> src_h ), th, vr,
		G.nty, vm );

	// stage regions alternate A, B, A, ...: S | H ops | V ops (last -> global)
	// (row pitches in LDS are made odd: | 1)
	std::vector< long > reg;
	reg.push_back( (long) ( hm[ 0 ] | 1 ) * vm[ 0 ]);

	for( int s = 1; s <= nh; s++ )
	{
		reg.push_back( (long) ( hm[ s ] | 1 ) * vm[ 0 ]);
	}

	for( int s = 1; s <= nv; s++ )
	{
		reg.push_back( (long) ( hm[ nh ] | 1 ) * vm[ s ]);
	}

	reg.pop_back(); // the last stage writes to global memory
	long mA = 0, mB = 0;

	for( size_t i = 0; i < reg.size(); i++ )
	{
		if( i & 1 ) mB = std::max( mB, reg[ i ]);
		else mA = std::max( mA, reg[ i ]);
	}

	long mC = 16, mS = 2; // mC: FIR taps of the longest (decimating) filter ...

	for( int s = 0; s < nh + nv; s++ )
	{
		const LOp& op = ( s < nh ? p -> h.ops[ s ] : p -> v.ops[ s - nh ]);

		if( op.type == OP_FIR )
		{
			mC = std::max( mC, (long) op.lat + 1 );
		}
	}

	// ... or the coefficient rows of a gather's outputs

	for( int s = 0; s < nh; s++ )
	{
		const LOp& op = p -> h.ops[ s ];

		if( op.type == OP_GATHER )
		{
			mC = std::max( mC, (long) hm[ s + 1 ] * (( op.maxtaps + 3 ) & ~3 ));
			mS = std::max( mS, (long) hm[ s + 1 ] * 2 );
		}
	}

	for( int s = 0; s < nv; s++ )
	{
		const LOp& op = p -> v.ops[ s ];

		if( op.type == OP_GATHER )
		{
			mC = std::max( mC, (long) vm[ s + 1 ] * (( op.maxtaps + 3 ) & ~3 ));
			mS = std::max( mS, (long) vm[ s + 1 ] * 2 );
		}
	}

	if( mB < 1 ) mB = 1;
	G.maxA = (int) mA;
	G.maxB = (int) mB;
	G.maxC = (int) mC;
	G.lds = (size_t) ( mA + mB ) * 16 + (size_t) mC * 4 + (size_t) mS * 4;

	if( G.lds > lds_cap )
	{
		return( AVIRHIP_OK ); // not usable with this tile; caller tries next
	}

	void* q;
	AVIRHIP_HIPCHECK( hipMalloc( &q, hr.size() * sizeof( int )));
	F -> allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, hr.data(), hr.size() * sizeof( int ),
		hipMemcpyHostToDevice ));
	G.d_hrng = (int*) q;
	AVIRHIP_HIPCHECK( hipMalloc( &q, vr.size() * sizeof( int )));
	F -> allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, vr.data(), vr.size() * sizeof( int ),
		hipMemcpyHostToDevice ));
	G.d_vrng = (int*) q;
	G.ok = true;
	return( AVIRHIP_OK );
}

static int pick_geom( avirhip_plan* p, FusedData* F, bool useh, bool usev,
	const int ( *cand )[ 2 ], int ncand, TileGeom& G )
{
	// first try to fit >= 2 workgroups per CU (64 KiB), then anything <= 150 KiB
	const size_t caps[ 2 ] = { 64 * 1024, 150 * 1024 };

	for( int c = 0; c < 2; c++ )
	{
		for( int i = 0; i < ncand; i++ )
		{
			int rc = make_geom( p, F, useh, usev, cand[ i ][ 0 ],
				cand[ i ][ 1 ], caps[ c ], G );

			if( rc != 0 ) return( rc );
			if( G.ok ) return( AVIRHIP_OK );
		}
	}

	return( AVIRHIP_OK );
}

int fused_prepare( avirhip_plan* p )
{
	p -> fused_ok = 0;
	p -> auto_path = 1;
	p -> fused = nullptr;

	if( p -> is_lancir || p -> ch != 4 || !chain_ok( p -> h ) ||
		!chain_ok( p -> v ))
	{
		return( AVIRHIP_OK );
	}

	FusedData* F = new FusedData();
	F -> fltbuf = nullptr;
	F -> dn = nullptr;
	F -> fused.ok = F -> hpass.ok = F -> vpass.ok = false;
	p -> fused = F;

	for( int a = 0; a < 2; a++ )
	{
		const LAxis& ax = ( a == 0 ? p -> h : p -> v );
		const float** pad = ( a == 0 ? F -> hpad : F -> vpad );

		for( size_t i = 0; i < ax.ops.size(); i++ )
		{
			const LOp& o = ax.ops[ i ];
			pad[ i ] = nullptr;

			if( o.type != OP_GATHER )
			{
				continue;
			}

			const int mt = ( o.maxtaps + 3 ) & ~3;
			std::vector< float > t( (size_t) o.out_len * mt, 0.0f );

			for( int j = 0; j < o.out_len; j++ )
			{
				memcpy( &t[ (size_t) j * mt ], &o.h_coef[ (size_t) j *
					o.maxtaps ], o.maxtaps * sizeof( float ));
			}

			void* q = nullptr;
			hipError_t e = hipMalloc( &q, t.size() * sizeof( float ));

			if( e == hipSuccess )
			{
				F -> allocs.push_back( q );
				e = hipMemcpy( q, t.data(), t.size() * sizeof( float ),
					hipMemcpyHostToDevice );
			}

			if( e != hipSuccess )
			{
				set_error( "tiled path: coefficient upload failed: %s",
					hipGetErrorString( e ));
				fused_release( p );
				return( AVIRHIP_EHIP );
			}

			pad[ i ] = (const float*) q;
		}
	}

	static const int cf[][ 2 ] = { { 32, 64 }, { 32, 32 }, { 64, 32 },
		{ 32, 16 }, { 64, 16 }, { 16, 16 }, { 16, 8 }, { 8, 8 } };
	static const int ch[][ 2 ] = { { 64, 8 }, { 128, 4 }, { 64, 4 }, { 32, 4 },
		{ 32, 2 }, { 16, 2 }, { 8, 1 } };
	// downsizing H pass: 16-row tiles (run_op's row-fast lane mapping)
	static const int chd[][ 2 ] = { { 32, 16 }, { 16, 16 }, { 64, 8 }, { 32, 8 },
		{ 32, 4 }, { 16, 2 }, { 8, 1 } };
	static const int cv[][ 2 ] = { { 32, 32 }, { 64, 16 }, { 32, 16 },
		{ 32, 8 }, { 16, 8 }, { 16, 4 }, { 8, 4 } };

	int rc = pick_geom( p, F, true, true, cf, 8, F -> fused );
	if( rc == 0 ) rc = pick_geom( p, F, true, false,
		( p -> src_w > p -> new_w ? chd : ch ), 7, F -> hpass );
	if( rc == 0 ) rc = pick_geom( p, F, false, true, cv, 7, F -> vpass );

	if( rc != 0 )
	{
		fused_release( p );
		return( rc );
	}

	if( rc == 0 ) rc = dn_prepare( p, &F -> dn );

	if( rc != 0 )
	{
		fused_release( p );
		return( rc );
	}

	const bool two = (( F -> hpass.ok || dn_has_h( F -> dn )) &&
		( F -> vpass.ok || dn_has_v( F -> dn )));
	p -> fused_ok = ( F -> fused.ok ? 2 : 0 ) | ( two ? 1 : 0 );

	// Upsizing (both axes): the intermediate is the big image -> keep it in
	// LDS (fused). Downsizing: the intermediate is small and a fused tile's
	// vertical halo is k times taller -> two passes.
	const bool up = ( p -> new_w >= p -> src_w && p -> new_h >= p -> src_h );

	// Measured on MI355X (bench.py --workload): the two-pass launch beats the
	// fused tile for every ratio class (cfg3 0.71 vs 0.75 ms, 1.3x non-integer
	// upsizing 0.10 vs 0.17 ms, 3x downsizing 0.18 vs 2.3 ms) -- the fused
	// tile's halo recomputation costs more than FltBuf's round trip. Path 3
	// stays selectable; exact-2x plans take the marching kernel (path 4).
	(void) up;
	p -> auto_path = ( two ? 2 : ( F -> fused.ok ? 3 : 1 ));

	return( AVIRHIP_OK );
}

// Both axes of the two-pass path run dn.hip's integer-ratio kernels.
bool fused_dn_both( const avirhip_plan* p )
{
	const FusedData* F = (const FusedData*) p -> fused;
	return( F != nullptr && dn_has_h( F -> dn ) && dn_has_v( F -> dn ));
}

void fused_release( avirhip_plan* p )
{
	FusedData* F = (FusedData*) p -> fused;

	if( F == nullptr )
	{
		return;
	}

	for( size_t i = 0; i < F -> allocs.size(); i++ )
	{
		(void) hipFree( F -> allocs[ i ]);
	}

	dn_release( F -> dn );
	delete F;
	p -> fused = nullptr;
}

static int zero_above_of( const LOp& consumer )
{
	return( consumer.view == VIEW_ZS ? consumer.zs_mmax : 0x7fffffff );
}

static void fill_ops( const LAxis& ax, int n, TOp* out,
	const float* const* pad )
{
	for( int i = 0; i < n; i++ )
	{
		const LOp& o = ax.ops[ i ];
		out[ i ].type = o.type;
		out[ i ].rowfast = ( o.type == OP_GATHER && o.in_len > o.out_len );
		out[ i ].out_len = o.out_len;
		out[ i ].zero_above = ( i + 1 < n ? zero_above_of( ax.ops[ i + 1 ]) :
			0x7fffffff );
		out[ i ].rf = o.rf; out[ i ].lat = o.lat; out[ i ].e = o.e;
		out[ i ].flt = o.d_flt; out[ i ].start = o.d_start;
		out[ i ].ntaps = o.d_ntaps; out[ i ].coef = pad[ i ];
		out[ i ].maxtaps = o.maxtaps;
	}
}

static int launch_tiles( avirhip_plan* p, const TileGeom& G, bool useh,
	bool usev, const void* src, int src_type, int src_ch, long src_ss,
	float* dst, long dst_ss, int dst_row0, int a, int b, hipStream_t st )
{
	// [a, b) = range on the y extent (output rows, or source rows for an
	// H-only pass)
	if( b <= a )
	{
		return( AVIRHIP_OK );
	}

	TileParams P;
	P.nh = ( useh ? (int) p -> h.ops.size() : 0 );
	P.nv = ( usev ? (int) p -> v.ops.size() : 0 );
	const FusedData* FD = (const FusedData*) p -> fused;
	fill_ops( p -> h, P.nh, P.hop, FD -> hpad );
	fill_ops( p -> v, P.nv, P.vop, FD -> vpad );
	P.hrng = G.d_hrng; P.vrng = G.d_vrng;
	P.ntx = G.ntx; P.nty = G.nty;
	P.ty0 = a / G.th;
	const int ty1 = ( b - 1 ) / G.th;
	P.src = src; P.src_ss = src_ss;
	P.src_type = src_type; P.src_ch = src_ch;
	// extent of the array at `src`: the source image, or FltBuf [src_h][new_w]
	P.src_w = ( useh ? p -> src_w : p -> new_w );
	P.src_h = p -> src_h;
	P.xzero_above = ( useh ? zero_above_of( p -> h.ops[ 0 ]) : 0x7fffffff );
	P.yzero_above = ( usev ? zero_above_of( p -> v.ops[ 0 ]) : 0x7fffffff );
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0;
	P.srow_lo = a; P.srow_hi = b;
	P.maxA = G.maxA; P.maxB = G.maxB; P.maxC = G.maxC;

	const int ntiles = G.ntx * ( ty1 - P.ty0 + 1 );
	AVIRHIP_HIPCHECK( AVIRHIP_DYN_LDS( k_tile, G.lds ));

#ifdef U2_DBG
	static unsigned long long* dclk = nullptr;
	static int calls = 0;

	if( dclk == nullptr )
	{
		(void) hipMalloc( &dclk, 65536 * 64 );
		(void) hipMemset( dclk, 0, 65536 * 64 );
	}

	P.clk = ( ntiles <= 65536 ? dclk : nullptr );
#endif
	hipLaunchKernelGGL( k_tile, dim3( ntiles ), dim3( 32, 8 ), G.lds, st, P );
	AVIRHIP_HIPCHECK( hipGetLastError() );
#ifdef U2_DBG
	if( ++calls == 40 || calls == 41 )
	{
		std::vector< unsigned long long > h( (size_t) ntiles * 8 );
		(void) hipStreamSynchronize( st );
		(void) hipMemcpy( h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost );
		double ph[ 8 ] = { 0 };
		unsigned long long t0 = ~0ULL, t1 = 0;
		const int np = 1 + 1 + P.nh + P.nv;

		for( int b = 0; b < ntiles; b++ )
		{
			for( int k = 1; k < np; k++ )
			{
				ph[ k ] += (double) ( h[ b * 8 + k ] - h[ b * 8 + k - 1 ]);
			}

			t0 = std::min( t0, h[ b * 8 ]);
			t1 = std::max( t1, h[ b * 8 + np - 1 ]);
		}

		fprintf( stderr, "k_tile nh=%d nv=%d tiles=%d lds=%zu span %.1f us; "
			"mean per tile (us): load %.2f", P.nh, P.nv, ntiles, G.lds,
			(double) ( t1 - t0 ) / 100.0, ph[ 1 ] / ntiles / 100.0 );

		for( int k = 2; k < np; k++ )
		{
			fprintf( stderr, " | op%d %.2f", k - 2, ph[ k ] / ntiles / 100.0 );
		}

		fprintf( stderr, "\n" );
	}
#endif
	return( AVIRHIP_OK );
}

// AVIRHIP_DNF=0 keeps integer-ratio downsizing on the two pass kernels (A/B)
static bool dnf_enabled()
{
	static const int on = []() { const char* e = getenv( "AVIRHIP_DNF" );
		return( e == nullptr || atoi( e ) != 0 ? 1 : 0 ); }();
	return( on != 0 );
}

// mode 3: fused; mode 2: two-pass.
bool fused_takes_raw( const avirhip_plan* p, int mode )
{
	// every loader of the tiled paths converts integer sources itself
	return( p -> fused != nullptr && ( mode == 2 || mode == 3 ));
}

// Whether the last pass of this mode stores integer / narrow pixels itself
// (fused_run's `iout`): the whole-ratio vertical kernel of the two-pass path.
bool fused_stores_int( const avirhip_plan* p, int mode )
{
	const FusedData* F = (const FusedData*) p -> fused;
	return( F != nullptr && mode == 2 && ( p -> fused_ok & 1 ) &&
		dn_has_v( F -> dn ));
}

int fused_run( avirhip_plan* p, int mode, const void* src, int src_type,
	int src_ch, long src_stride, float* dst, int row0, int row1,
	hipStream_t st, void* iout )
{
	FusedData* F = (FusedData*) p -> fused;

	if( iout != nullptr && !fused_stores_int( p, mode ))
	{
		return( 1 );
	}

	if( F == nullptr || ( mode == 3 && !( p -> fused_ok & 2 )) ||
		( mode == 2 && !( p -> fused_ok & 1 )) || ( mode != 2 && mode != 3 ))
	{
		return( 1 );
	}

	if( src_type == AVIRHIP_F32 )
	{
		if( src_ch != 4 || ( (uintptr_t) src & 15 ) || ( src_stride & 3 ))
		{
			return( 1 ); // DMA tile loads need 16-byte aligned float rows
		}
	}
	else
	if(( src_type != AVIRHIP_U8 && src_type != AVIRHIP_U16 ) ||
		!fused_takes_raw( p, mode ) ||
		( src_type == AVIRHIP_U16 && ( (uintptr_t) src & 1 )))
	{
		return( 1 );
	}

	// (the integer loaders read four elements per pixel)
	if( src_type != AVIRHIP_F32 && (long) ( p -> src_h - 1 ) * src_stride +
		(long) p -> src_w * src_ch < 4 )
	{
		return( 1 );
	}

	if( (uintptr_t) dst & 15 )
	{
		return( 1 );
	}

	const long dss = (long) p -> new_w * 4;

	if( mode == 3 )
	{
		return( launch_tiles( p, F -> fused, true, true, src, src_type, src_ch,
			src_stride, dst, dss, row0, row0, row1, st ));
	}

	// integer-ratio downsizing on both axes: one marching launch, no FltBuf
	if( src_type == AVIRHIP_F32 && dn_has_h( F -> dn ) && dn_has_v( F -> dn ) &&
		dnf_enabled() && !( p -> variant & AVIRHIP_VARIANT_DN_TWO_PASS ))
	{
		GPOut O;

		if( iout != nullptr )
		{
			gp_make_out( p, iout, O );
		}

		const int rc = dn_run_hv( F -> dn, (const float*) src, src_stride, dst,
			row0, row1, st, ( iout != nullptr ? &O : nullptr ));

		if( rc != 1 )
		{
			return( rc );
		}
	}

	// two-pass: which FltBuf rows does the band need?
	int a = row0, b = row1 - 1;

	for( int i = (int) p -> v.ops.size() - 1; i >= 0; i-- )
	{
		int ia, ib;
		op_need( p -> v.ops[ i ], a, b, ia, ib );
		a = ia;
		b = ib;
	}

	a = std::max( a, 0 );
	b = std::min( b, p -> src_h - 1 );

	if( F -> fltbuf == nullptr )
	{
		void* q;
		AVIRHIP_HIPCHECK( hipMalloc( &q, (size_t) p -> src_h * p -> new_w *
			4 * sizeof( float )));
		F -> allocs.push_back( q );
		F -> fltbuf = (float*) q;
		p -> alloc_bytes += (size_t) p -> src_h * p -> new_w * 4 *
			sizeof( float );
	}

	int rc = ( dn_has_h( F -> dn ) ?
		dn_run_h( F -> dn, src, src_type, src_ch, src_stride, F -> fltbuf,
		dss, a, b + 1, st ) :
		launch_tiles( p, F -> hpass, true, false, src, src_type, src_ch,
		src_stride, F -> fltbuf, dss, 0, a, b + 1, st ));

	if( rc != 0 )
	{
		return( rc );
	}

	if( dn_has_v( F -> dn ))
	{
		if( iout != nullptr )
		{
			GPOut O;
			gp_make_out( p, iout, O );
			return( dn_run_v( F -> dn, F -> fltbuf, dss, p -> new_w, dst, row0,
				row1, st, &O ));
		}

		return( dn_run_v( F -> dn, F -> fltbuf, dss, p -> new_w, dst, row0,
			row1, st ));
	}

	return( launch_tiles( p, F -> vpass, false, true, F -> fltbuf,
		AVIRHIP_F32, 4, dss, dst, dss, row0, row0, row1, st ));
}

} // namespace avirhip
