// NOT IN THE BUILD -- a round-5 experiment kept for the record (profiles/r05_bacc/README.md: bit-exact in all three of
// its forms, slower than k_sacc / k_sacc2v in each). To try it again: move it to avir_amd/csrc/, add build/bacc.o to the
// Makefile's OBJS and restore the three hooks in gpass.hip (prepare after sacc_prepare_axis, release, and
// bacc_run_col( D -> ba_v, D -> mid, mid_ss, p -> new_w, dst - row0 * dst_ss, dst_ss, row0, row1, st, out ) before the
// vertical sacc_run_axis of gpass_run's AVIR branch).
//
// bacc.hip -- BLOCK ACCUMULATION: the downsizing axes of a plan (AVIR k > 1:
// doResize / doResize2 over the zero-stuffed view, then the correction filter,
// avir.h:3884-4328 + 3748-3866) without a window and without a run-in.
//
// What the other kernels of path 5 pay for on such an axis (NOTEBOOK.md 4.2d/e):
// the gather kernels keep an NT-deep window per lane (NT = 13 .. 64 rows of
// 16 bytes: registers or a 42-KiB LDS ring, one or two waves per SIMD); the
// streaming-accumulation kernels keep 16 accumulator slots instead, but a work
// item has to multiply NT samples in before its first output is complete -- and
// since an axis only has a few dozen 64-lane strips, the chip is filled by
// cutting the axis into short chunks: x 1.26 (horizontal) to x 1.7 (vertical)
// of repeated or wasted multiplies, plus a compare ladder (k_sacc) or zero
// products that are only exact for finite samples (k_sacc2).
//
// Here an axis is cut into BLOCKS of four consecutive gather outputs. A lane
// owns a line of samples along the axis, exactly as in sacc.hip; a block
// streams the samples [ s0, s3 + NT ) its four windows cover ONCE and multiplies
// each into the outputs alive at it. The windows of neighbouring outputs start
// d1 <= d2 <= d3 < NT samples apart, so the alive set is a contiguous range of
// slots that only changes at six points: seven straight loops -- {0}, {0,1},
// {0,1,2}, {0..3}, {1,2,3}, {2,3}, {3} -- whose trip counts come from the block's
// header. Every product the reference makes is made once, in its order
// (ascending taps from +0.0f), and no other: nothing is multiplied by a padding
// zero, so non-finite samples behave as in the reference, and a block needs
// nothing from the block before it but the rows they share. Accumulators are
// 16 registers. The samples travel HBM -> LDS by LDS-DMA, each row ONCE per work
// item, into a ring of Q rows per wave (Q >= NT + 7 k: the rows of a block and
// of the block after it, which are sent while this one is multiplied); a block
// reads its rows from the ring as often as its windows overlap. (The first
// version kept an 8-row FIFO and fetched every block's rows again "from L2":
// with a few thousand waves in flight the reuse distance is the whole
// intermediate image, the re-reads went to HBM -- 3.9 x the traffic -- and the
// pass was slower than the kernel it replaces, profiles/r05_bacc/.)
//
// The correction filter (GP_POST) runs in registers on the stream of gather
// outputs: a 7-value window + the block's 4 give 4 filtered outputs per block
// with compile-time indices. A work item is a strip of 64 lanes x a chunk of
// outputs; its only overhead are the 6 (+ 2 of alignment) gather outputs the
// filter reads beyond the chunk.
//
// Coefficients: one table row of four floats per (block, sample) -- the taps of
// the four slots at that sample, laid out by the host (zeros where a slot is
// not alive; those entries are never used as factors) -- read as ONE scalar
// load per sample; rows are shared by every strip of the launch.

#include "gpass_dev.h"

namespace avirhip {

#define BA_R 4   // gather outputs per block

struct BAAxis
{
	int mode;        // GP_POST or GP_NONE
	int nt, in_len, g_len, out_len, e;
	int mmax;        // zero-stuffed view: samples above read 0
	float f[ 4 ];    // correction filter f0..f3
	const int* hdr;  // [nblocks][8]: s0, d1, d2, d3, first table row, flags
	const float* tab; // rows of 4 coefficients
	int nblocks;
	int shift;       // block b's slot r is gather output 4 b + r - shift
	int ring;        // rows of the sample ring (32, 64 or 128): every [ s0 of
	                 // a block, last row of the next block ] + a group fits
};

struct BAParams
{
	const float* src; long src_ss; // float RGBA rows, floats per row
	int width;                     // pixels per row (lanes)
	float* dst; long dst_ss;       // row 0 of the destination image
	int row_lo, row_hi;            // output rows to produce
	int nstrips, chunk;            // chunk: a multiple of 4
	int chunk0;                    // first chunk index of the launch
	BAAxis ax;
	GPOut out;                     // on != 0: integer output stage
};

__device__ __forceinline__ void ba_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff, const unsigned soff )
{
	asm volatile( "s_mov_b32 m0, %0\n\t"
		"buffer_load_dwordx4 %1, %2, %3 offen lds"
		:: "s"( lds ), "v"( voff ), "s"( rs ), "s"( soff ) : "memory" );
}

// One channel's output stage: GPOut's arithmetic (gp_store_int) for the element
// this lane holds.
__device__ __forceinline__ void ba_store_elem( const GPOut& O, const long row,
	const int x, const int ch, const float v )
{
	if( ch >= O.ch )
	{
		return;
	}

	const long e = row * O.stride + (long) x * O.ch + ch;

	if( O.type == AVIRHIP_F32 )
	{
		((float*) O.base )[ e ] = v; // (1-3 channels: the padding is dropped)
		return;
	}

	unsigned int q;

	if( O.fin )
	{
		q = (unsigned int) min( max( (int) ( v + 0.5f ), 0 ), O.pk_i );
	}
	else
	{
		float t = v;
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
		q = (unsigned int) t;
	}

	if( O.type == AVIRHIP_U8 )
	{
		((unsigned char*) O.base )[ e ] = (unsigned char) q;
	}
	else
	{
		((unsigned short*) O.base )[ e ] = (unsigned short) q;
	}
}

// One block: the samples [ s0, s3 + nt ) of the ring multiplied into the
// accumulators of the slots alive at each. Before that, what the NEXT block
// needs and is not on its way yet is sent: its coefficient rows (ONE LDS-DMA
// instruction: lane i carries row i into one of two 1-KiB table buffers behind
// the ring) and its sample rows -- groups of four view rows per instruction
// (`gnext`: the first group not sent; a wave's row is 16 pixels = 256 bytes;
// view row u = source row clamp( u ), ring slot u & ( Q - 1 )). Everything THIS
// block reads was sent before those, so the wait is "at most the loads just
// issued may be outstanding" (loads retire in order; the count stands for loads
// only, stores in the queue can only make it wait longer).
//
// Coefficients come from LDS as broadcast reads, not as scalar loads: scalar
// loads return out of order and share lgkmcnt with the LDS reads, so every wait
// for a sample would also drain the coefficient load issued a moment before it
// (the first version did exactly that: a full scalar-memory latency per trip).
// The reads stay ordinary loads: the compiler tracks them, counts its waits
// (LDS returns in order) and may copy their registers between the loops -- reads
// hidden in inline assembly were copied while still in flight.
template< bool ZS, int Q >
__device__ __forceinline__ void ba_block( const BAAxis& A, gp_cint* const hd,
	const int b, const __amdgpu_buffer_rsrc_t rs,
	const __amdgpu_buffer_rsrc_t trs, const unsigned lds0, const int vcol,
	const int vrow4, const int pitch, const int lane, int& gnext, float* const acc )
{
	extern __shared__ float ba_smem[];
	const int s0 = hd[ 0 ];
	const int d1 = hd[ 1 ];
	const int d2 = hd[ 2 ];
	const int d3 = hd[ 3 ];
	const int unx = hd[ 8 ] + hd[ 11 ] + A.nt; // the next block's last row + 1
	const int slast = A.in_len - 1;

	// the next block's table, then its rows
	ba_dma( trs, lds0 + (unsigned) ( Q * 256 + (( b + 1 ) & 1 ) * 1024 ),
		lane * 16, (unsigned) hd[ 12 ] * 16u );
	int nsent = 1;

	for( ; gnext * 4 < unx; gnext++ )
	{
		// lane l carries 16 bytes of view row 4 g + ( l >> 4 )
		const int r = min( max( gnext * 4 + vrow4, 0 ), slast );
		ba_dma( rs, lds0 + (unsigned) ((( gnext * 4 ) & ( Q - 1 )) * 256 ),
			vcol + r * pitch, 0u );
		nsent++;
	}

	gp_wait_vm( nsent );

	const f4* const ct = (const f4*) ( ba_smem + Q * 64 + ( b & 1 ) * 256 ) - s0;

	auto fetch = [&]( const int u ) -> float
	{
		float v = ba_smem[ ( u & ( Q - 1 )) * 64 + lane ];

		// a zero-stuffed view reads 0 above its last replicated sample
		if( ZS && u > A.mmax )
		{
			v = 0.0f;
		}

		return( v );
	};

#pragma unroll
	for( int r = 0; r < BA_R; r++ )
	{
		acc[ r ] = 0.0f; // (sums start from +0.0f, avir.h:3938 / 4170)
	}

	int u = s0;
	float xa = fetch( u ), xb;
	f4 ca = ct[ u ], cb;

	// sample ( X, C ) into the slots LO .. HI: a product, then a SEPARATE add
	// (no contraction: the reference rounds twice)
#define BA_MAC( LO, HI, X, C ) \
	if( LO <= 0 && HI >= 0 ) acc[ 0 ] = acc[ 0 ] + C.x * X; \
	if( LO <= 1 && HI >= 1 ) acc[ 1 ] = acc[ 1 ] + C.y * X; \
	if( LO <= 2 && HI >= 2 ) acc[ 2 ] = acc[ 2 ] + C.z * X; \
	if( LO <= 3 && HI >= 3 ) acc[ 3 ] = acc[ 3 ] + C.w * X;

	// one phase: CNT samples into the slots LO .. HI; ( xa, ca ) is the sample
	// in hand, the next one's value and coefficient row are read while it is
	// multiplied. Two samples per trip so that no register changes its role.
	// (Reading one sample past the block's last: a slot of the ring, any value.)
#define BA_PHASE( LO, HI, CNT ) \
	{ \
		int i_ = ( CNT ); \
		for( ; i_ >= 2; i_ -= 2 ) \
		{ \
			cb = ct[ u + 1 ]; \
			xb = fetch( u + 1 ); \
			BA_MAC( LO, HI, xa, ca ) \
			ca = ct[ u + 2 ]; \
			xa = fetch( u + 2 ); \
			BA_MAC( LO, HI, xb, cb ) \
			u += 2; \
		} \
		if( i_ > 0 ) \
		{ \
			cb = ct[ u + 1 ]; \
			xb = fetch( u + 1 ); \
			BA_MAC( LO, HI, xa, ca ) \
			xa = xb; ca = cb; \
			u++; \
		} \
	}

	BA_PHASE( 0, 0, d1 )
	BA_PHASE( 0, 1, d2 - d1 )
	BA_PHASE( 0, 2, d3 - d2 )
	BA_PHASE( 0, 3, A.nt - d3 )
	BA_PHASE( 1, 3, d1 )
	BA_PHASE( 2, 3, d2 - d1 )
	BA_PHASE( 3, 3, d3 - d2 )
#undef BA_PHASE
#undef BA_MAC
}

// Lanes: ONE CHANNEL of one pixel each -- lane l = channel l & 3 of pixel l >> 2
// of the strip's 16 pixels. The arithmetic per pixel is what whole-pixel lanes
// would issue (a scalar multiply and add take half the time of the packed pair
// they replace), but a strip is 256 bytes wide: four times the work items for
// the same image (an axis has few 64-pixel strips: 40 for a 2560-pixel row), a
// ring of Q rows is Q / 4 KiB instead of Q KiB, so that a CU holds sixteen waves
// instead of four and their latencies hide each other. (Whole-pixel lanes were
// built first: one wave per SIMD, every sample waiting out its LDS read --
// slower than the kernels they were to replace, profiles/r05_bacc/.)
template< bool POST, int Q >
__global__ void __launch_bounds__( 64 ) k_bacc_col( const BAParams P )
{
	extern __shared__ float ba_smem[];
	const BAAxis& A = P.ax;
	const int lane = (int) threadIdx.x;
	const int item = (int) blockIdx.x;
	const int strip = item % P.nstrips;
	const int chunk = P.chunk0 + item / P.nstrips;
	const int x = strip * 16 + ( lane >> 2 );
	const int ch = lane & 3;
	const bool lok = ( x < P.width );

	// outputs of this item
	const int n0 = max( P.row_lo, chunk * P.chunk );
	const int n1 = min( min( P.row_hi, ( chunk + 1 ) * P.chunk ), A.out_len );

	if( n1 <= n0 )
	{
		return;
	}

	// blocks: slot r of block b is the (virtual) gather output m = 4 b + r;
	// output n reads m in [ n, n + 6 ] (POST) or m = n
	const int b0 = n0 >> 2;
	const int b1 = ( n1 - 1 + ( POST ? 6 : 0 )) >> 2;

	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) ba_smem;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
		(void*) A.tab, 0, 0x7fffffff, 0x00020000 );
	// the row feed: lane l carries piece l & 15 (a pixel) of row 4 g + ( l >> 4 )
	const int vcol = min( strip * 16 + ( lane & 15 ), P.width - 1 ) * 16;
	const int vrow4 = lane >> 4;
	const int pitch = (int) ( P.src_ss * 4 );
	gp_cint* const hdr = (gp_cint*) (unsigned long) A.hdr;

	const float f0 = A.f[ 0 ], f1 = A.f[ 1 ], f2 = A.f[ 2 ], f3 = A.f[ 3 ];
	float w[ 7 ]; // the correction filter's window: the 7 gather outputs before the block
#pragma unroll
	for( int i = 0; i < 7; i++ )
	{
		w[ i ] = 0.0f;
	}

	// the first block's table and rows (later ones are sent a block ahead)
	int gnext = hdr[ b0 * 8 ] >> 2; // (floor: view rows may be negative)
	ba_dma( trs, lds0 + (unsigned) ( Q * 256 + ( b0 & 1 ) * 1024 ), lane * 16,
		(unsigned) hdr[ b0 * 8 + 4 ] * 16u );
	{
		const int ux = hdr[ b0 * 8 ] + hdr[ b0 * 8 + 3 ] + A.nt;

		for( ; gnext * 4 < ux; gnext++ )
		{
			const int r = min( max( gnext * 4 + vrow4, 0 ), A.in_len - 1 );
			ba_dma( rs, lds0 + (unsigned) ((( gnext * 4 ) & ( Q - 1 )) * 256 ),
				vcol + r * pitch, 0u );
		}
	}

	for( int b = b0; b <= b1; b++ )
	{
		float acc[ BA_R ];

		// (a block whose samples reach above a zero-stuffed view's last
		// replicated sample -- the last blocks of a 1 < k < 2 axis -- takes the
		// copy of the loops that checks for it; the others pay nothing)
		if( hdr[ b * 8 + 5 ] != 0 )
		{
			ba_block< true, Q >( A, hdr + b * 8, b, rs, trs, lds0, vcol, vrow4,
				pitch, lane, gnext, acc );
		}
		else
		{
			ba_block< false, Q >( A, hdr + b * 8, b, rs, trs, lds0, vcol, vrow4,
				pitch, lane, gnext, acc );
		}

		const int m0 = b * 4;

		if( POST )
		{
			// v[ 0 .. 10 ] = gather outputs m0 - 7 .. m0 + 3; output n = m0 - 6 + j
			// = f0 G[c] + sum f_i ( G[c+i] + G[c-i] ), c at v[ 4 + j ]
			// (avir.h:3787-3802: right + left, times the tap, accumulated)
			const float v[ 11 ] = { w[ 0 ], w[ 1 ], w[ 2 ], w[ 3 ], w[ 4 ], w[ 5 ],
				w[ 6 ], acc[ 0 ], acc[ 1 ], acc[ 2 ], acc[ 3 ]};
#pragma unroll
			for( int j = 0; j < 4; j++ )
			{
				const int n = m0 - 6 + j;

				if( n >= n0 && n < n1 )
				{
					float r = f0 * v[ 4 + j ];
					r = r + f1 * ( v[ 5 + j ] + v[ 3 + j ]);
					r = r + f2 * ( v[ 6 + j ] + v[ 2 + j ]);
					r = r + f3 * ( v[ 7 + j ] + v[ 1 + j ]);

					if( lok )
					{
						if( P.out.on != 0 )
						{
							ba_store_elem( P.out, (long) ( n - P.row_lo ), x, ch, r );
						}
						else
						{
							P.dst[ (long) n * P.dst_ss + (long) x * 4 + ch ] = r;
						}
					}
				}
			}
#pragma unroll
			for( int i = 0; i < 7; i++ )
			{
				w[ i ] = v[ 4 + i ];
			}
		}
		else
		{
#pragma unroll
			for( int j = 0; j < 4; j++ )
			{
				const int n = m0 + j;

				if( n >= n0 && n < n1 && lok )
				{
					if( P.out.on != 0 )
					{
						ba_store_elem( P.out, (long) ( n - P.row_lo ), x, ch,
							acc[ j ]);
					}
					else
					{
						P.dst[ (long) n * P.dst_ss + (long) x * 4 + ch ] = acc[ j ];
					}
				}
			}
		}
	}

	// no LDS-DMA may outlive the workgroup's LDS
	asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct BAData
{
	BAAxis a;
	double k; // average source advance per gather output
};

// Block tables of a downsizing AVIR axis. Returns 0 with *out == nullptr when
// the axis is not one this kernel runs.
int bacc_prepare_axis( const GPAxisHost& H, std::vector< void* >& allocs,
	BAData** out )
{
	*out = nullptr;
	const GPAxis& A = H.a;

	if( !H.ok || A.lanc != 0 || ( A.mode != GP_POST && A.mode != GP_NONE ) ||
		A.nt < 4 || A.g_len < 1 || H.h_coef == nullptr ||
		getenv( "AVIRHIP_NO_BACC" ) != nullptr )
	{
		return( AVIRHIP_OK );
	}

	// output n = f0 G[c] + sum f_i ( G[c+i] + G[c-i] ), c = n - e: it reads the
	// virtual gather outputs m = n .. n + 6 when m = index + e + 3
	const int shift = ( A.mode == GP_POST ? A.e + 3 : 0 );
	const int nblocks = ( A.out_len + 6 + 3 ) / 4 + 1;
	// (+ 1: every block looks at the header after its own)
	std::vector< int > hdr( (size_t) ( nblocks + 1 ) * 8, 0 );
	int span = 0; // rows from a block's first to the next block's last
	std::vector< float > tab;
	const std::vector< int >& st = H.start;

	for( int b = 0; b < nblocks; b++ )
	{
		int gi[ BA_R ], s[ BA_R ];

		for( int r = 0; r < BA_R; r++ )
		{
			gi[ r ] = std::max( 0, std::min( 4 * b + r - shift, A.g_len - 1 ));
			s[ r ] = st[ gi[ r ]];

			if( r > 0 && s[ r ] < s[ r - 1 ])
			{
				return( AVIRHIP_OK ); // (positions must not go back)
			}
		}

		if( s[ 3 ] - s[ 0 ] >= A.nt || A.nt + ( s[ 3 ] - s[ 0 ]) + 2 > 64 )
		{
			// (the four windows must overlap; a block's table rows travel in
			// one LDS-DMA instruction of 64 lanes)
			return( AVIRHIP_OK );
		}

		const int rows = A.nt + ( s[ 3 ] - s[ 0 ]);

		if( b > 0 )
		{
			span = std::max( span, s[ 3 ] + A.nt - hdr[ (size_t) ( b - 1 ) * 8 ]);
		}

		hdr[ (size_t) b * 8 ] = s[ 0 ];
		hdr[ (size_t) b * 8 + 1 ] = s[ 1 ] - s[ 0 ];
		hdr[ (size_t) b * 8 + 2 ] = s[ 2 ] - s[ 0 ];
		hdr[ (size_t) b * 8 + 3 ] = s[ 3 ] - s[ 0 ];
		hdr[ (size_t) b * 8 + 4 ] = (int) ( tab.size() / 4 );
		hdr[ (size_t) b * 8 + 5 ] = ( s[ 3 ] + A.nt - 1 > A.view_mmax ? 1 : 0 );

		// (+ 2: the rows read ahead of the block's last sample)
		for( int i = 0; i <= rows + 1; i++ )
		{
			for( int r = 0; r < BA_R; r++ )
			{
				const int t = i - ( s[ r ] - s[ 0 ]);
				tab.push_back( t >= 0 && t < A.nt && i < rows ?
					H.h_coef[ (size_t) gi[ r ] * A.nt + t ] : 0.0f );
			}
		}
	}

	// (a block's table is fetched as 64 rows whatever it holds)
	tab.resize( tab.size() + 64 * 4, 0.0f );

	// the header after the last block: "nothing more to send"
	for( int i = 0; i < 8; i++ )
	{
		hdr[ (size_t) nblocks * 8 + i ] = hdr[ (size_t) ( nblocks - 1 ) * 8 + i ];
	}

	// (rows travel in aligned groups of four: up to 3 beyond what was asked)
	if( span + 4 > 128 )
	{
		return( AVIRHIP_OK );
	}

	BAData* D = new BAData();
	memset( &D -> a, 0, sizeof( D -> a ));
	D -> a.ring = ( span + 4 <= 32 ? 32 : ( span + 4 <= 64 ? 64 : 128 ));
	D -> a.mode = A.mode; D -> a.nt = A.nt; D -> a.in_len = A.in_len;
	D -> a.g_len = A.g_len; D -> a.out_len = A.out_len; D -> a.e = A.e;
	D -> a.mmax = A.view_mmax;
	memcpy( D -> a.f, A.f, sizeof( A.f ));
	D -> a.nblocks = nblocks; D -> a.shift = shift;
	D -> k = H.kmax;

	void* q = nullptr;
	hipError_t e = hipMalloc( &q, hdr.size() * sizeof( int ));

	if( e == hipSuccess )
	{
		allocs.push_back( q );
		D -> a.hdr = (const int*) q;
		e = hipMemcpy( q, hdr.data(), hdr.size() * sizeof( int ),
			hipMemcpyHostToDevice );
	}

	if( e == hipSuccess )
	{
		e = hipMalloc( &q, tab.size() * sizeof( float ));
	}

	if( e == hipSuccess )
	{
		allocs.push_back( q );
		D -> a.tab = (const float*) q;
		e = hipMemcpy( q, tab.data(), tab.size() * sizeof( float ),
			hipMemcpyHostToDevice );
	}

	if( e != hipSuccess )
	{
		delete D;
		set_error( "bacc_prepare_axis: %s", hipGetErrorString( e ));
		return( e == hipErrorOutOfMemory ? AVIRHIP_ENOMEM : AVIRHIP_EHIP );
	}

	*out = D;
	return( AVIRHIP_OK );
}

void bacc_release_axis( BAData* D )
{
	delete D;
}

// The vertical pass: `src` = float RGBA rows (FltBuf), lanes = its pixels.
// Returns 1 when the call cannot take this kernel.
int bacc_run_col( const BAData* D, const float* src, long src_ss, int width,
	float* dst, long dst_ss, int row_lo, int row_hi, hipStream_t st,
	const GPOut* out )
{
	if( D == nullptr || ( (uintptr_t) src & 15 ) || ( src_ss & 3 ) ||
		( out == nullptr && (( (uintptr_t) dst & 15 ) || ( dst_ss & 3 ))) ||
		(double) src_ss * 4.0 * D -> a.in_len >= 2147483648.0 ||
		(double) width * 16.0 >= 2147483648.0 )
	{
		return( 1 );
	}

	if( row_hi <= row_lo )
	{
		return( AVIRHIP_OK );
	}

	BAParams P;
	memset( &P, 0, sizeof( P ));
	P.src = src; P.src_ss = src_ss; P.width = width;
	P.dst = dst; P.dst_ss = dst_ss;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.nstrips = ( width + 15 ) / 16;
	P.ax = D -> a;

	if( out != nullptr )
	{
		P.out = *out;
	}

	// Outputs per work item: the correction filter reads 6 gather outputs
	// beyond the chunk (+ 2 of block alignment): ( chunk + 8 ) / chunk of the
	// gather work; the chip wants two or more waves per SIMD.
	static const int force = ( getenv( "AVIRHIP_BA_CHUNK" ) != nullptr ?
		atoi( getenv( "AVIRHIP_BA_CHUNK" )) : 0 );
	const int nout = row_hi - row_lo;
	const int halo = ( D -> a.mode == GP_POST ? 8 : 0 );
	double best = -1.0;
	int chunk = 56;

	// (resident waves: 160 KiB of LDS per CU over ring + tables, at most 32)
	const size_t lds = (size_t) D -> a.ring * 256 + 2048;
	const double cap = 256.0 * std::max( 1, std::min( 32,
		(int) ( 160 * 1024 / ( lds + 256 ))));

	for( int c = 8; c <= 512; c += 4 )
	{
		const double items = (double) (( nout + c - 1 ) / c ) * P.nstrips;
		const double rounds = ceil( items / cap );
		const double per = ceil( items / rounds / 1024.0 );
		const double cost = rounds * per * ( c + halo ) *
			( per < 2.0 ? 1.3 : 1.0 );

		if( best < 0.0 || cost < best )
		{
			best = cost;
			chunk = c;
		}
	}

	if( force > 0 )
	{
		chunk = std::max( 4, force & ~3 );
	}

	P.chunk = chunk;
	P.chunk0 = row_lo / chunk;
	const int nch = ( row_hi - 1 ) / chunk - P.chunk0 + 1;
	const int items = nch * P.nstrips;
	const bool post = ( D -> a.mode == GP_POST );

#define BA_GO( POST, Q ) { \
		AVIRHIP_HIPCHECK( AVIRHIP_DYN_LDS(( k_bacc_col< POST, Q > ), lds )); \
		hipLaunchKernelGGL(( k_bacc_col< POST, Q > ), dim3( items ), dim3( 64 ), \
			lds, st, P ); }

	if( D -> a.ring == 32 )
	{
		if( post ) BA_GO( true, 32 ) else BA_GO( false, 32 )
	}
	else
	if( D -> a.ring == 64 )
	{
		if( post ) BA_GO( true, 64 ) else BA_GO( false, 64 )
	}
	else
	{
		if( post ) BA_GO( true, 128 ) else BA_GO( false, 128 )
	}
#undef BA_GO

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
