// gpassv.hip -- the vertical pass of the general-ratio pass kernels (see
// gpass.hip for the family's description): launch geometry, the AVIR upsizing
// variants of k_gv and the dispatch to the other variant groups.

#include "gpassv_kernel.h"

namespace avirhip {

// Register-window variants exist for the tap counts that occur: 12 for AVIR
// upsizing (FIR7 then gather), 13-25 for AVIR 1 < k < 2 downsizing (gather
// then FIR7, gpassv_post.hip), the even counts 6-24 for LANCIR
// (gpassv_lanc.hip); the LDS-window kernel takes everything else.
static int launch_gv( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	if( P.ax.lanc )
	{
		return( launch_gv_lanc( P, wp, items, lds, st ));
	}
	else
	if( P.ax.mode == GP_PRE )
	{
		if( nt == 12 ) GV_LAUNCH( GP_PRE, false, 12, 16 );
		else GV_LAUNCH( GP_PRE, false, 0, 0 );
	}
	else
	if( P.ax.mode == GP_POST )
	{
		launch_gv_post( P, wp, items, lds, st );
	}
	else
	{
		GV_LAUNCH( GP_NONE, false, 0, 0 );
	}

	return( 0 );
}

int gpass_run_v( const GPAxis& A_, int v_blk, int v_rs, int v_rc,
	const float* src, long src_ss, int width, float* dst, long dst_ss,
	int dst_row0, int row_lo, int row_hi, hipStream_t st, const GPOut* out,
	const void* raw, int raw_type, int raw_ch, long raw_stride )
{
	GVParams P;
	memset( &P.out, 0, sizeof( P.out ));
	P.raw = raw; P.raw_ss = raw_stride; P.raw_ch = raw_ch;
	P.raw_kind = ( raw_type == AVIRHIP_U8 ? 1 : ( raw_type == AVIRHIP_U16 ? 2 : 3 ));
	// (rows are raw_stride elements apart; the last one ends with its pixels)
	P.raw_elems = (long) ( A_.in_len - 1 ) * raw_stride +
		(long) width * raw_ch;

	if( raw != nullptr && P.raw_elems < 4 )
	{
		return( 1 );
	}

	// raw rows travel as bytes by LDS-DMA (k_gv, LVAR bit 1): dword-aligned base
	// and row pitch, under 2 GiB -- what gpass_lancir_takes_raw promised
	P.raw_dma = 0; P.raw_bytes = 0;

	if( raw != nullptr )
	{
		const long esz = ( raw_type == AVIRHIP_U8 ? 1 :
			( raw_type == AVIRHIP_U16 ? 2 : 4 ));
		const long bytes = P.raw_elems * esz;
		const int nt = A_.nt;

		if(( (uintptr_t) raw & 3 ) != 0 || (( raw_stride * esz ) & 3 ) != 0 ||
			bytes > 0x7ffffffcL || !A_.lanc || nt < 6 || nt > 24 || ( nt & 1 ))
		{
			return( 1 );
		}

		P.raw_dma = (int) ( esz * raw_ch );
		// (whole dwords: base and pitch are dword-aligned and device allocations
		// dword-granular, so the row's last partial dword is fetched whichever
		// way the range check treats a dword that straddles num_records)
		P.raw_bytes = (int) (( bytes + 3 ) & ~3L );
	}

	if( out != nullptr )
	{
		P.out = *out;
	}

	P.src = src; P.src_ss = src_ss; P.width = width;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0;
	P.ax = A_;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.dbg = ( getenv( "AVIRHIP_GP_DBG" ) != nullptr ? atoi( getenv( "AVIRHIP_GP_DBG" )) : 0 );
	P.blk = v_blk; P.rs = v_rs; P.rc = v_rc;

	// whole-pixel lanes (64-pixel strips) wherever the image is wide enough
	// to fill them and the rings stay within 64 KiB
	int wp = ( width > 48 && ( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 ?
		2 : 1 );

	if( getenv( "AVIRHIP_GV_WP" ) != nullptr ) // tuning aid
	{
		wp = ( atoi( getenv( "AVIRHIP_GV_WP" )) == 2 &&
			( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 ? 2 : 1 );
	}

	if( out != nullptr || raw != nullptr )
	{
		// the integer output stage and the raw-source loader handle whole
		// pixels
		if(( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 > 64 * 1024 )
		{
			return( 1 );
		}

		wp = 2;
	}

	if( wp == 1 )
	{
		// half-pixel lanes (images narrower than 49 pixels, or windows too
		// long for whole-pixel rings) only exist as the LDS-window kernel:
		// its rings hold the window, whatever the plan sized them for
		int rs = 2, rc = 2;

		if( P.ax.mode == GP_PRE )
		{
			rc = 2;
			while( rc < P.ax.nt + 1 ) rc <<= 1;
		}
		else
		{
			while( rs < P.ax.nt + 2 * GV_PF + 3 ) rs <<= 1;
			rc = ( P.ax.mode == GP_POST ? 8 : 2 );
		}

		if(( rs + rc ) * 512 + GV_QB + 6 * 1024 > 64 * 1024 )
		{
			return( 1 );
		}

		P.rs = rs; P.rc = rc;
	}

	const int rowb = wp * 512;
	P.nstrips = ( width + 32 * wp - 1 ) / ( 32 * wp );

	// chunk height: a chunk's first window (NT rows, + 6 of FIR warm-up) is
	// fed before its first output; its coefficient rows and positions live
	// in LDS (<= 6 KiB)
	const int rows = row_hi - row_lo;
	const int ntp = ( P.ax.nt + 3 ) & ~3;
	const int gextra = ( P.ax.mode == GP_POST ? 7 : 0 );
	const int maxg = std::max( gextra + 1, 6144 / ( ntp * 4 + 4 ));
	// warm-up in units of output rows (a fed row costs about a third of one)
	const int warm_div = ( getenv( "AVIRHIP_GV_WARMDIV" ) != nullptr ?
		std::max( 1, atoi( getenv( "AVIRHIP_GV_WARMDIV" ))) : 3 );

	const int warm = std::max( 1, (int) ( (long) ( P.ax.nt + 6 ) *
		std::max( 1, P.ax.out_len ) / std::max( 1, P.ax.in_len ) / warm_div ));

	const size_t lds_guess = (size_t) ( P.rs + P.rc ) * rowb + GV_QB + 3072;
	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_guess + 512 ))));

	// (chunks down to 4 rows: a small frame -- 640x480 -> 1024x768 -- runs
	// faster as two short waves per SIMD than as one of 12 rows, 0.0231 ->
	// 0.0193 ms; tools/gp_solo_sweep.sh)
	P.chunk = balanced_chunk( rows, P.nstrips, 4, maxg - gextra, warm, wpc,
		true );

	if( getenv( "AVIRHIP_GV_CHUNK" ) != nullptr ) // tuning aid
	{
		P.chunk = std::max( 1, std::min( atoi( getenv( "AVIRHIP_GV_CHUNK" )),
			maxg - gextra ));
	}

	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = (size_t) ( P.rs + P.rc ) * rowb + GV_QB +
		(size_t) ( P.chunk + gextra ) * ( ntp * 4 + 4 ) + 64;

#ifdef GP_DBG
	{
		static int once = 0;
		if( !once++ ) fprintf( stderr, "k_gv: wp %d strips %d chunk %d items %d "
			"lds %zu nt %d mode %d\n", wp, P.nstrips, P.chunk, items, lds,
			P.ax.nt, P.ax.mode );
	}
#endif
	// (1: no kernel of the variant group takes this plan -- the caller falls
	// back, api.cpp turns an unexpected 1 into AVIRHIP_NEED_SRC / the generic path)
	if( launch_gv( P, wp, items, lds, st ) != 0 )
	{
		return( 1 );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
