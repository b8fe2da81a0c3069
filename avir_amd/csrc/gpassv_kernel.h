// gpassv_kernel.h -- k_gv, the vertical pass kernel (template); included by
// the translation units that instantiate its variants (gpassv_lanc.hip,
// gpassv_post.hip, gpassv.hip): one file would take five minutes to compile.
#ifndef AVIRHIP_GPASSV_KERNEL_H
#define AVIRHIP_GPASSV_KERNEL_H

#include "gpass_dev.h"

namespace avirhip {

// ---------------------------------------------------------------------
// vertical pass
// ---------------------------------------------------------------------

// One source row per step: the row's pixels travel global memory -> register
// queue (GV_PF rows in flight, statically indexed: the step loop is unrolled
// GV_PF times) -> whatever the mode keeps of them:
//   PRE   a 7-row FIR window in registers (statically indexed too); each
//         step yields one row of C, stored in the LDS ring the gathers read. The clamped source view makes FIR( Sext )
//         the clamped view of C for EVERY index (rows before the first / after
//         the last are constant), so there are no edge cases, only clamped
//         row loads; rows of C above the zero-stuffed view's end are zero.
//   NONE  the row goes into the LDS source ring.
//   POST  the same, and gather outputs go through a second LDS ring that the
//         correction FIR reads.
// After every step the outputs whose last input has arrived are produced (a
// wave-uniform while loop). LDS holds only what is addressed dynamically:
// rs rows >= NT + 1 of source (NONE / POST), rc rows of C (>= NT + 1) or of
// gather outputs (8).

// RW > 0 (needs NTC > 0): the gather window lives in REGISTERS. An output is
// produced in the very step its last input row arrives, so its window is
// always "the last NTC rows": with the step loop unrolled RW times (RW a power
// of two >= NTC) the window's ring slots are compile-time constants -- no LDS
// reads for the taps at all, LDS only holds the landing queue and the tables.
// WP: lane width. 1 = half pixel (8 B): a wave row is 32 pixels, one DMA
// instruction lands two rows. 2 = whole pixel (16 B): 64 pixels per wave row,
// one row per DMA instruction -- twice the independent arithmetic per
// instruction stream: the kernel is bound by the serial latency of a wave
// (queue read -> FIR chain -> 12-deep accumulation chain), not by issue slots,
// so the wide form is the default wherever the image is wide enough.
template< int WP > struct GVLane;
template<> struct GVLane< 1 > { typedef f2 T; };
template<> struct GVLane< 2 > { typedef f4 T; };

// LVAR (LANCIR variants only -- compiled in, they cost the float RGBA kernels
// 35 .. 50 registers): bit 0 = the four-lane summation orders of 1-3 channel
// images (gp_dot), bit 1 = the raw-source loader, bit 2 = ... of uint8 pixels
// (the element type known at compile time: no dispatch in the step).
template< int MODE, bool LANC, int NTC, int RW, int WP, int LVAR >
__global__ void __launch_bounds__( 64 ) k_gv( const GVParams P )
{
	// (a value the compiler can see is never 4 / always 4)
	const int lord = (( LVAR & 1 ) ? ( P.ax.lanc & 3 ) : 4 );
	typedef typename GVLane< WP > :: T VT;
	constexpr int ROWB = WP * 512;   // LDS bytes of one wave row
	constexpr int RPG = 2 / WP;      // rows per DMA group (instruction)
	constexpr int PXW = 32 * WP;     // pixels per wave row
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	VT* const sring = (VT*) smem;                       // [rs][64 lanes]
	VT* const mring = sring + (size_t) P.rs * 64;       // [rc][64 lanes]
	// behind the rings and the 6 KiB landing queue: this chunk's coefficient
	// rows (padded to four taps) and positions -- in LDS, not behind scalar
	// loads: a scalar load in the tap loop costs its whole latency per use
	// (every wait on lgkmcnt with one outstanding is a full drain)
	float* const ctab = (float*) ( smem + ( P.rs + P.rc ) * ROWB + GV_QB );

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int x0 = strip * PXW;
	const int y0 = P.row_lo + chunk * P.chunk;
	const int y1 = min( y0 + P.chunk, P.row_hi );

	if( y0 >= y1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	const int smask = P.rs - 1;
	const int mmask = P.rc - 1;
	const int slast = A.in_len - 1;
	const int nt = ( NTC > 0 ? NTC : A.nt );
	const int ntp = ( nt + 3 ) & ~3;

	// this lane's (half) pixel
	const int hx = x0 + ( WP == 2 ? lane : lane >> 1 );
	const bool xok = ( hx < P.width );
	const long dcol = (long) hx * 4 + ( WP == 2 ? 0 : ( lane & 1 ) * 2 );
	// (LANCIR, RGB: channel 0 sums its tail taps differently -- gp_dot)
	const bool ach = ( WP == 2 || ( lane & 1 ) == 0 );

	auto mread = [&]( const int m ) -> VT
	{
		return( mring[ ( m & mmask ) * 64 + lane ]);
	};

	auto sread = [&]( const int u ) -> VT
	{
		return( sring[ ( u & smask ) * 64 + lane ]);
	};

	const VT f0 = (VT) A.f[ 0 ], f1 = (VT) A.f[ 1 ], f2_ = (VT) A.f[ 2 ],
		f3 = (VT) A.f[ 3 ];

	// gather outputs this chunk evaluates: [ga, gb]
	const int glast = A.g_len - 1;
	const int ga = ( MODE == GP_POST ? gp_clamp( y0 - A.e - 3, glast ) : y0 );
	const int gb = ( MODE == GP_POST ? gp_clamp( y1 - 1 - A.e + 3, glast ) :
		y1 - 1 );

	// source rows fed: u_a .. u_b (PRE: six rows of FIR warm-up first)
	const int u_a = a_start[ ga ] - ( MODE == GP_PRE ? 6 : 0 );
	const int u_b = a_start[ gb ] + nt - 1;

	// stage the tables (plain loads: they retire before the first DMA goes
	// out). All of a lane's loads are issued before any is written to LDS --
	// a load-store loop pays the memory latency once per 64 elements, which
	// measured as a third of this kernel's time.
	int* const stab = (int*) ( ctab + ( gb - ga + 1 ) * ntp );
	{
		const int total = ( gb - ga + 1 ) * ntp;
		constexpr int GV_TE = 24; // 6 KiB of table / 4 bytes / 64 lanes
		float tmp[ GV_TE ];
#pragma unroll
		for( int k = 0; k < GV_TE; k++ )
		{
			const int i = lane + 64 * k;
			tmp[ k ] = 0.0f;

#ifdef GP_DBG
			if( P.dbg & 16 ) continue;
#endif
			if( i < total )
			{
				const int gi = i / ntp;
				const int t = i - gi * ntp;
				const long row = ( LANC ? A.fidx[ ga + gi ] : ga + gi );

				if( t < nt )
				{
					tmp[ k ] = A.coef[ row * nt + t ];
				}
			}
		}
#pragma unroll
		for( int k = 0; k < GV_TE; k++ )
		{
			const int i = lane + 64 * k;

			if( i < total )
			{
				ctab[ i ] = tmp[ k ];
			}
		}

		for( int i = lane; i <= gb - ga; i += 64 )
		{
			stab[ i ] = A.start[ ga + i ] + nt - 1;
		}
	}

	int g = ga;                 // next gather output
	int g_need = a_start[ g ] + nt - 1; // its last input row
	int n = y0;                 // POST: next final output

	// output rows leave strictly in order, from y0 on: a running pointer
	float* dp = P.dst + (long) ( y0 - P.dst_row0 ) * P.dst_ss + dcol;
	long orow = y0 - P.dst_row0; // (integer output stage: row of P.out.base)

	auto store_row = [&]( const VT o )
	{
#ifdef GP_DBG
		if(( P.dbg & 1 ) && o.x != 1234.5f ) return;
#endif
		// (LANCIR's output stage lives in its LAST pass, the horizontal one:
		// GPLOut in k_gh; its vertical pass never stores integer pixels)
		if( !LANC && WP == 2 && P.out.on != 0 )
		{
			{
				f4 o4;
				o4.x = o.x; o4.y = o.y;
				o4.z = ( WP == 2 ? o[ WP == 2 ? 2 : 0 ] : 0.0f );
				o4.w = ( WP == 2 ? o[ WP == 2 ? 3 : 0 ] : 0.0f );
				gp_store_int_row( P.out, orow, hx, o4, xok );
			}

			orow++;
			return;
		}

		if( xok )
		{
			*(VT*) dp = o;
		}

		dp += P.dst_ss;
	};

	// the next output's coefficient row and the position after it are read
	// from LDS an output ahead, so that their latency hides behind the
	// arithmetic in between (a wave runs alone on most of its issue slots:
	// exposed LDS round trips were a third of the step time)
	constexpr int CNB = ( NTC > 0 && NTC <= 16 ? ( NTC + 3 ) / 4 : 0 );
	f4 cpre[ CNB > 0 ? CNB : 1 ];
	int need1_v; // g_need1, still in its vector register

	auto prefetch_tabs = [&]()
	{
		// (rows past gb are read -- clamped -- and never used)
		const int gi = min( g + 1, gb ) - ga;
		need1_v = ( g + 1 <= gb ? stab[ gi ] : 0x7fffffff );

		if( CNB > 0 )
		{
			const float* const cr = ctab + ( min( g, gb ) - ga ) * ntp;
#pragma unroll
			for( int b = 0; b < CNB; b++ )
			{
				cpre[ b ] = *(const f4*) ( cr + 4 * b );
			}
		}
	};

	VT win[ RW > 0 ? RW : 1 ]; // RW > 0: the last RW rows the gathers read
#pragma unroll
	for( int i = 0; i < ( RW > 0 ? RW : 1 ); i++ )
	{
		win[ i ] = (VT) 0.0f;
	}

	// gather output g is done: store it, or (POST) run the correction FIR
	auto finish = [&]( const VT o )
	{
		if( MODE == GP_POST )
		{
			mring[ ( g & mmask ) * 64 + lane ] = o;

			// final outputs whose FIR window [c - 3, c + 3] (clamped) is
			// complete: c + 3 <= g, or g is the last gather output
			while( n < y1 && ( gp_clamp( n - A.e + 3, glast ) <= g ))
			{
				const int c = n - A.e;
				VT r = f0 * mread( gp_clamp( c, glast ));
				r = r + f1 * ( mread( gp_clamp( c + 1, glast )) +
					mread( gp_clamp( c - 1, glast )));
				r = r + f2_ * ( mread( gp_clamp( c + 2, glast )) +
					mread( gp_clamp( c - 2, glast )));
				r = r + f3 * ( mread( gp_clamp( c + 3, glast )) +
					mread( gp_clamp( c - 3, glast )));
				store_row( r );
				n++;
			}
		}
		else
		{
			store_row( o );
		}

		g++;
		g_need = __builtin_amdgcn_readfirstlane( need1_v );
		prefetch_tabs();
	};

	// everything whose inputs are complete once row `u` is in its ring.
	// IC: the step's phase within the unrolled loop (window slot of row u).
	auto drain = [&]( const int u, auto IC )
	{
		constexpr int I = decltype( IC )::value;

		// (after the chunk's last output g_need is 0x7fffffff -- prefetch_tabs'
		// sentinel --, so one compare decides a step without an output: with
		// `g <= gb` in front the addresses of the loop body were computed
		// before the second test, ~14 scalar instructions in every step)
		while( g_need <= u )
		{
			const int s0 = g_need - nt + 1;
			const float* const cr = ctab + ( g - ga ) * ntp;
			VT o;

#ifdef GP_DBG
			if( P.dbg & 2 )
			{
				o = win[ 0 ];
			}
			else
#endif
			if( RW > 0 )
			{
				// g_need == u: outputs are drained the step they complete
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( win[ ( I - ( NTC - 1 ) + t +
						4 * ( RW > 0 ? RW : 1 )) & (( RW > 0 ? RW : 1 ) - 1 )]); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, lord, ach );
			}
			else
			if( MODE == GP_PRE )
			{
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( mread( s0 + t )); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, lord, ach );
			}
			else
			{
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( sread( s0 + t )); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, lord, ach );
			}

			finish( o );
		}
	};

	prefetch_tabs();

	// ---- the feed: source rows by LDS-DMA, one group per instruction (WP 1:
	// a row pair, lanes 0-31 row 2p, lanes 32-63 row 2p + 1; WP 2: one row),
	// GV_PF groups in flight. Waiting for "at most GV_PF - 1 vector-memory
	// operations outstanding" right after issuing group p + GV_PF guarantees
	// group p + 1 has landed (loads retire in order; stores in between only
	// make the wait stricter): rows are read from the queue a step ahead.
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) smem;
	const int dpx = gp_clamp( x0 + ( lane & ( PXW - 1 )), P.width - 1 );
	const int srow_b = (int) P.src_ss * 4;
	// PRE: pairs land in a queue of GV_PF + 1 slots behind the C ring;
	// NONE / POST: straight in the source ring (rs >= nt + 2 * GV_PF + 3)
	const unsigned qbase = lds0 + (unsigned) ( P.rs + P.rc ) * ROWB;
	const int pa = ( RPG == 2 ? u_a >> 1 : u_a ); // first group (floor)
	const int pb = ( RPG == 2 ? u_b >> 1 : u_b );
	const int pbs = min( pb, slast );

	// (Slot reuse: the DMA of group p + GV_PF takes the slot group p - 1 held.
	// That slot was read into `qnext` two steps earlier; the step waits
	// lgkmcnt(0) right before its DMA: an LDS-DMA may only take a slot whose
	// reads have been WAITED for, not merely issued. Until round 6
	// this relied on `qnext` having been "consumed" a step earlier, which for the
	// register window is a rename, not a wait: see step(). k_lf: NOTEBOOK 9.13.)
	int qi = 0; // landing-queue slot of the next pair issued ...
	int qr = 0; // ... and of the pair being read (both count modulo GV_PF + 1)

	auto issue_pair = [&]( const int p )
	{
		// past the last group: the count of operations per step stays
		// uniform -- the last group again, into the slot the next group would
		// have taken (nobody reads it; no branch, no dump slot)
		if( MODE == GP_PRE || RW > 0 )
		{
			const int r = ( RPG == 2 ? gp_clamp( min( p, pb ) * 2 + ( lane >> 5 ),
				slast ) : max( min( p, pbs ), 0 ));

			gp_dma( rsrc, qbase + (unsigned) qi * 1024, r * srow_b + dpx * 16 );
			qi = ( qi == GV_PF ? 0 : qi + 1 );
		}
		else
		{
			// (straight into the source ring: rows past the last one must not
			// overwrite ring rows still in use -- the dump slot)
			const bool live = ( p <= pb );
			const int r = gp_clamp( min( p, pb ) * RPG +
				( RPG == 2 ? lane >> 5 : 0 ), slast );
			const unsigned slot = ( !live ? qbase + (unsigned) ( GV_PF + 1 ) * 1024 :
				lds0 + (unsigned) (( p * RPG ) & smask ) * ROWB );

			gp_dma( rsrc, slot, r * srow_b + dpx * 16 );
		}
	};

	// Raw sources (integer / narrower pixels, whole-pixel lanes; LVAR bit 1):
	// the rows travel as BYTES by LDS-DMA into the landing queue, GV_PF rows
	// ahead like float rows, and a lane converts its pixel when it reads the
	// queue -- the (float) cast and zero padding of the pack pass, in the loader.
	// The host only picks these variants for images whose base and row pitch
	// are dword-aligned (gpass_lancir_takes_raw; a 64-pixel segment starts at a
	// multiple of 64 pixels). A row is ONE 16-byte-per-lane instruction issued
	// by the first 4 x (bytes per pixel) lanes: 64 pixels x bytes per pixel,
	// contiguous. (As 1-4 dword instructions of 256 bytes the count was a
	// run-time loop and picked the s_waitcnt immediate through a branch chain in
	// every step; with the element-type dispatch of the conversion a step was 68
	// scalar instructions for 46 vector ones -- the pass was bound by the CU's
	// scalar unit: rocprofv3 SQ_INSTS_SALU 21.5 M against SQ_INSTS_VALU 14.5 M per
	// launch, profiles/r05_lanc.)
	// (The first form had every lane load its own pixel one step ahead and carried
	// three loaders in one kernel: 92-96 KB of code for the 18-tap variants
	// against the float form's 38 -- more than the instruction cache holds -- and
	// a step as long as a global-memory round trip: CLancIR 5184x3456 ->
	// 1920x1280 RGB uint8 spent 118 of its 153 us in this pass. NOTEBOOK 9.8.)
	constexpr bool TD = ( WP == 2 && ( LVAR & 2 ) != 0 );
	const int tbpp = P.raw_dma;
	const int raw_kind = (( LVAR & 4 ) ? 1 : P.raw_kind );
	const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.raw, 0, P.raw_bytes, 0x00020000 );
	const int trow_b = (int) P.raw_ss * ( raw_kind == 1 ? 1 :
		( raw_kind == 2 ? 2 : 4 ));
	const bool tlane = ( lane < 4 * tbpp );

	auto issue_raw = [&]( const int p )
	{
		// (past the last row: the last row again, into the next slot)
		const int r = max( min( p, pbs ), 0 );
		const unsigned slot = qbase + (unsigned) qi * 1024;

		qi = ( qi == GV_PF ? 0 : qi + 1 );

		// (bytes past the image's end read as zeros: the buffer's range check;
		// bytes past the row's end are the next row's -- lanes beyond the width,
		// never stored)
		const int boff = r * trow_b + x0 * tbpp + lane * 16;

		if( tlane )
		{
			gp_dma( rraw, slot, boff );
		}
	};

	// this lane's pixel of the row in queue slot `q`: packScanline's (float)
	// cast and zero padding (avir.h:2777-2840; copyScanline*v, lancir.h:1406)
	auto raw_px = [&]( const int q ) -> f4
	{
		const char* const rp = smem + ( P.rs + P.rc ) * ROWB + q * 1024 +
			lane * tbpp;
		float v[ 4 ];

		// four elements whatever the channel count (what lies behind a pixel is
		// the next pixel, inside the slot), the padding selected to zero after:
		// no branch per channel in a loop body that is unrolled 32 times
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

	for( int k = 0; k < GV_PF; k++ )
	{
		if( TD ) issue_raw( pa + k ); else issue_pair( pa + k );
	}

	VT qnext = (VT) 0.0f; // the next step's row of the queue

	gp_wait_vm( GV_PF - 1 ); // the first group has landed

	if( TD )
	{
		const f4 o = raw_px( 0 );
		qnext[ 0 ] = o.x; qnext[ 1 ] = o.y;
		qnext[ WP == 2 ? 2 : 0 ] = o.z; qnext[ WP == 2 ? 3 : 0 ] = o.w;
	}
	else
	if( MODE == GP_PRE || RW > 0 )
	{
		qnext = *(const VT*) ( smem + ( P.rs + P.rc ) * ROWB +
			lane * sizeof( VT ));
	}

	VT tw[ 8 ]; // PRE: the FIR window (rows u - 6 .. u in slots ( u - u_a ) & 7)
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		tw[ i ] = (VT) 0.0f;
	}

	// steps run over whole pairs; rows before u_a / after u_b of the first /
	// last pair are fed like any other (clamped loads), outputs never need
	// them
	const int u0 = pa * RPG;

	constexpr int UN = ( RW > 8 ? RW : 8 );

	auto step = [&]( const int u, auto IC )
	{
		constexpr int i = decltype( IC )::value;

		if( u > ( pb * RPG + RPG - 1 ))
		{
			return;
		}

		// The DMA below takes the queue slot that was read into `qnext` two
		// steps ago. "Consumed one step ago" is not a wait when the consumer is
		// the register window (a rename; the first real use is the output's dot
		// product, steps later): in a chunk's run-in, steps without an output
		// are a few dozen instructions, and with many short-chunk waves on a CU
		// the sub-dword queue reads of 1-3 channel integer rows can sit in the
		// LDS queue longer than a row that hits in L2 takes to arrive -- the
		// read then returns the NEW row (found in round 6 by the differential
		// against the paranoid build: CLancIR 3555x2951 -> 3135x2672 RGB uint16,
		// chunks of 2-4 rows, a tenth of the frames; NOTEBOOK 10).
		// One s_waitcnt lgkmcnt(0) before the DMA goes out -- as in k_lf: every
		// LDS read this wave has issued, the queue row of two steps ago among
		// them, has RETURNED (the rule of waits.h: a slot's reads are waited
		// for, not merely issued). Same box, three alternations: no workload
		// that runs k_gv changes (profiles/r06_paranoid/gv_fix_variants_ab.txt;
		// pinning the queue row in registers instead cost CLancIR 3840x2160 ->
		// 1280x720 RGBA uint8 6 %). PRE runs the row through the FIR in the step
		// that takes it over -- arithmetic, i.e. a wait -- and needs none.
		GP_PRIO_LOW();

		if( MODE != GP_PRE )
		{
			AVIRHIP_WAIT_LGKM( 0 );
		}

		if( TD )
		{
			issue_raw( u + GV_PF );
			// (row u + 1 opens the oldest group in flight)
			gp_wait_vm( GV_PF - 1 );
			qr = ( qr == GV_PF ? 0 : qr + 1 );
		}
		else
		{
			if( RPG == 1 || ( i & 1 ) == 0 )
			{
#ifdef GP_DBG
				if( !( P.dbg & 8 ))
#endif
				issue_pair(( RPG == 2 ? u >> 1 : u ) + GV_PF );
			}

			if( RPG == 1 || ( i & 1 ))
			{
				// row u + 1 opens a group: it is the oldest one in flight
#ifdef GP_DBG
				if( !( P.dbg & 4 ))
#endif
				gp_wait_vm( GV_PF - 1 );

				qr = ( qr == GV_PF ? 0 : qr + 1 );
			}
		}

		// this step's row was read from the queue during the last one
		const VT qcur = qnext;

		if( TD )
		{
			const f4 o = raw_px( qr );
			qnext[ 0 ] = o.x; qnext[ 1 ] = o.y;
			qnext[ WP == 2 ? 2 : 0 ] = o.z; qnext[ WP == 2 ? 3 : 0 ] = o.w;
		}
		else
		if( MODE == GP_PRE || RW > 0 )
		{
			qnext = *(const VT*) ( smem + ( P.rs + P.rc ) * ROWB + qr * 1024 +
				( RPG == 2 ? (( i + 1 ) & 1 ) * 512 : 0 ) + lane * sizeof( VT ));
		}

		GP_PRIO_HIGH();

		if( MODE == GP_PRE )
		{
			tw[ i & 7 ] = qcur;
			VT c = f0 * tw[ ( i - 3 ) & 7 ];
			c = c + f1 * ( tw[ ( i - 2 ) & 7 ] + tw[ ( i - 4 ) & 7 ]);
			c = c + f2_ * ( tw[ ( i - 1 ) & 7 ] + tw[ ( i - 5 ) & 7 ]);
			c = c + f3 * ( tw[ i & 7 ] + tw[ ( i - 6 ) & 7 ]);

			if( u > A.view_mmax )
			{
				c = (VT) 0.0f;
			}

			if( RW > 0 )
			{
				win[ i & (( RW > 0 ? RW : 1 ) - 1 )] = c;
			}
			else
			{
				mring[ ( u & mmask ) * 64 + lane ] = c;
			}

			if( u - u0 >= 6 )
			{
				drain( u, IC );
			}
		}
		else
		{
			if( RW > 0 )
			{
				VT v = qcur;

				// (the zero-stuffed view's tail: AVIR plans only)
				if( !LANC && u > A.view_mmax )
				{
					v = (VT) 0.0f;
				}

				win[ i & (( RW > 0 ? RW : 1 ) - 1 )] = v;
			}
			else
			if( u > A.view_mmax )
			{
				sring[ ( u & smask ) * 64 + lane ] = (VT) 0.0f;
			}

			drain( u, IC );
		}
	};

#ifdef GP_DBG
	if( P.dbg & 32 ) return;
#endif
	for( int ub = u0; ub <= u_b; ub += UN )
	{
		step( ub + 0, std::integral_constant< int, 0 >() );
		step( ub + 1, std::integral_constant< int, 1 >() );
		step( ub + 2, std::integral_constant< int, 2 >() );
		step( ub + 3, std::integral_constant< int, 3 >() );
		step( ub + 4, std::integral_constant< int, 4 >() );
		step( ub + 5, std::integral_constant< int, 5 >() );
		step( ub + 6, std::integral_constant< int, 6 >() );
		step( ub + 7, std::integral_constant< int, 7 >() );

		if( UN > 8 )
		{
			step( ub + 8, std::integral_constant< int, 8 >() );
			step( ub + 9, std::integral_constant< int, 9 >() );
			step( ub + 10, std::integral_constant< int, 10 >() );
			step( ub + 11, std::integral_constant< int, 11 >() );
			step( ub + 12, std::integral_constant< int, 12 >() );
			step( ub + 13, std::integral_constant< int, 13 >() );
			step( ub + 14, std::integral_constant< int, 14 >() );
			step( ub + 15, std::integral_constant< int, 15 >() );
		}

		if( UN > 16 )
		{
			step( ub + 16, std::integral_constant< int, 16 >() );
			step( ub + 17, std::integral_constant< int, 17 >() );
			step( ub + 18, std::integral_constant< int, 18 >() );
			step( ub + 19, std::integral_constant< int, 19 >() );
			step( ub + 20, std::integral_constant< int, 20 >() );
			step( ub + 21, std::integral_constant< int, 21 >() );
			step( ub + 22, std::integral_constant< int, 22 >() );
			step( ub + 23, std::integral_constant< int, 23 >() );
			step( ub + 24, std::integral_constant< int, 24 >() );
			step( ub + 25, std::integral_constant< int, 25 >() );
			step( ub + 26, std::integral_constant< int, 26 >() );
			step( ub + 27, std::integral_constant< int, 27 >() );
			step( ub + 28, std::integral_constant< int, 28 >() );
			step( ub + 29, std::integral_constant< int, 29 >() );
			step( ub + 30, std::integral_constant< int, 30 >() );
			step( ub + 31, std::integral_constant< int, 31 >() );
		}
	}

	GP_WAIT_VM( 0 );
}

// (half-pixel lanes: the LDS-window kernel only, see gpass_run_v)
#define GV_LAUNCH( MODE, LANC, NT, RW ) do { if( wp == 2 ) \
	hipLaunchKernelGGL(( k_gv< MODE, LANC, NT, RW, 2, 0 > ), dim3( items ), \
	dim3( 64 ), lds, st, P ); else \
	hipLaunchKernelGGL(( k_gv< MODE, LANC, 0, 0, 1, 0 > ), dim3( items ), \
	dim3( 64 ), lds, st, P ); } while( 0 )

// LANCIR: + the summation order of the image's channel count, + raw sources
#define GV_LAUNCH_L1( NT, RW, WPV, LV ) hipLaunchKernelGGL(( k_gv< GP_NONE, \
	true, NT, RW, WPV, LV > ), dim3( items ), dim3( 64 ), lds, st, P )

// (float sources: gpassv_lanc.hip; raw sources -- whole-pixel lanes only --:
// gpassv_lancraw.hip)
#define GV_LAUNCH_L( NT, RW ) do { \
	const bool l3_ = ( P.ax.lanc != 4 ); \
	if( wp != 2 ) { if( l3_ ) GV_LAUNCH_L1( 0, 0, 1, 1 ); \
		else GV_LAUNCH_L1( 0, 0, 1, 0 ); } \
	else { if( l3_ ) GV_LAUNCH_L1( NT, RW, 2, 1 ); \
		else GV_LAUNCH_L1( NT, RW, 2, 0 ); } } while( 0 )

#define GV_LAUNCH_LR( NT, RW ) do { \
	if( P.ax.lanc != 4 ) GV_LAUNCH_L1( NT, RW, 2, 3 ); \
	else GV_LAUNCH_L1( NT, RW, 2, 2 ); } while( 0 )

// (uint8 pixels: gpassv_lancraw8.hip)
#define GV_LAUNCH_LR8( NT, RW ) do { \
	if( P.ax.lanc != 4 ) GV_LAUNCH_L1( NT, RW, 2, 7 ); \
	else GV_LAUNCH_L1( NT, RW, 2, 6 ); } while( 0 )

// the launchers of the variant groups (one translation unit each)
// (0, or 1 when the variant group has no kernel for the plan: nothing was
// launched)
int launch_gv_lanc( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st );
int launch_gv_lanc_raw( const GVParams& P, int items, size_t lds,
	hipStream_t st );
int launch_gv_lanc_raw8( const GVParams& P, int items, size_t lds,
	hipStream_t st );
void launch_gv_post( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st );

} // namespace avirhip

#endif
