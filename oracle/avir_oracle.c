/*
 * oracle/avir_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of avir::CImageResizer< fpclass_def< float > >::
 * resizeImage() (avaneev/avir v3.1): filter design, build-mode selection,
 * step bookkeeping and the per-scanline executor, written scanline-at-a-time
 * with replicated prefix/suffix pixels in flip-flop buffers exactly as the
 * reference does (and unlike the product, which uses index algebra on the
 * GPU). Each function cites the reference lines it follows. Must be compiled
 * with -ffp-contract=off.
 *
 * Parity: PINNED against oracle/_ref (the real reference) and the golden
 * hashes of SURVEY.md 8(c) by tests/test_oracle.py. The sRGB gamma stages are
 * restated too, and so is the error-diffusion ditherer
 * (CImageResizerDithererErrdINL, avir.h:4441-4537; oracle_set_dither).
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.1415926535897932   /* avir.h:101 */
#define PID2 1.5707963267948966 /* avir.h:104 */

static void* xalloc( size_t n )
{
	void* p = calloc( n > 0 ? n : 1, 1 );

	if( p == NULL )
	{
		abort();
	}

	return( p );
}

/* ---- CSineGen, avir.h:1004-1040 ---- */

typedef struct { double v1, v2, incr; } sinegen;

static void sg_init( sinegen* g, double si, double ph )
{
	g -> v1 = sin( ph );
	g -> v2 = sin( ph - si );
	g -> incr = 2.0 * cos( si );
}

static double sg_next( sinegen* g )
{
	const double res = g -> v1;
	g -> v1 = g -> incr * res - g -> v2;
	g -> v2 = res;
	return( res );
}

/* ---- CDSPWindowGenPeakedCosine, avir.h:1054-1093 ---- */

typedef struct { double alpha, len2i, wn; sinegen w1; } pcwin;

static void pw_init( pcwin* w, double alpha, double len2 )
{
	w -> alpha = alpha;
	w -> len2i = 1.0 / len2;
	w -> wn = 0.0;
	sg_init( &w -> w1, PID2 / len2, PI * 0.5 );
}

static double pw_next( pcwin* w )
{
	const double h = pow( w -> wn * w -> len2i, w -> alpha );
	w -> wn += 1.0;
	return( sg_next( &w -> w1 ) * ( 1.0 - h ));
}

/* ---- CDSPPeakedCosineLPF, avir.h:1490-1589 ---- */

static int lpf_fl2( double len2 )
{
	return( (int) ceil( len2 ) - 1 );
}

static void lpf_generate( double* op, double len2, double freq2, double alpha,
	double dcgain )
{
	const int fl2 = lpf_fl2( len2 );
	const int flen = fl2 + fl2 + 1;
	pcwin wf;
	sinegen f2;
	double* op2;
	int t;

	pw_init( &wf, alpha, len2 );
	sg_init( &f2, freq2, 0.0 );
	op += fl2;
	op2 = op;
	sg_next( &f2 );

	if( dcgain > 0.0 )
	{
		double s;
		*op = freq2 * pw_next( &wf );
		s = *op;

		for( t = 1; t <= fl2; t++ )
		{
			const double v = sg_next( &f2 ) * pw_next( &wf ) / t;
			op++;
			op2--;
			*op = v;
			*op2 = v;
			s += v + v;
		}

		s = dcgain / s;

		for( t = 0; t < flen; t++ )
		{
			*op2 = *op2 * s;
			op2++;
		}
	}
	else
	{
		*op = freq2 * pw_next( &wf );

		for( t = 1; t <= fl2; t++ )
		{
			const double v = sg_next( &f2 ) * pw_next( &wf ) / t;
			op++;
			op2--;
			*op = v;
			*op2 = v;
		}
	}
}

/* ---- calcFIRFilterResponse<float>, avir.h:460-503 (fltlat = 0) ---- */

static void fir_response( const float* flt, int fltlen, double th,
	double* re0, double* im0 )
{
	const double sincr = 2.0 * cos( th );
	double cvalue1 = 1.0;
	double svalue1 = 0.0;
	double cvalue2 = cos( -th );
	double svalue2 = sin( -th );
	double re = 0.0;
	double im = 0.0;

	while( fltlen > 0 )
	{
		double tmp;
		re += cvalue1 * (double) flt[ 0 ];
		im += svalue1 * (double) flt[ 0 ];
		flt++;
		fltlen--;

		tmp = cvalue1;
		cvalue1 = sincr * cvalue1 - cvalue2;
		cvalue2 = tmp;

		tmp = svalue1;
		svalue1 = sincr * svalue1 - svalue2;
		svalue2 = tmp;
	}

	*re0 = re;
	*im0 = im;
}

/* ---- normalizeFIRFilter<double>, avir.h:516-541 ---- */

static void normalize_fir( double* p, int l, double dcgain )
{
	double s = 0.0;
	int i;

	for( i = 0; i < l; i++ )
	{
		s += p[ i ];
	}

	s = dcgain / s;

	for( i = 0; i < l; i++ )
	{
		p[ i ] = p[ i ] * s;
	}
}

/* ---- CDSPFIREQ, avir.h:1116-1480 (linear bands, MinFreq == 0) ---- */

typedef struct
{
	int z, zi, z2, bandcount, uselast;
	double* center;
	double* k1;
	double* k2;
} fireq;

static void eq_fill_band( const fireq* q, double x1, double x2, double* kb1,
	double* kb2, double* oscbuf, const double* winbuf )
{
	const int z = q -> z;
	const double s2_incr = PI * x2;
	const double s2_coeff = 2.0 * cos( s2_incr );
	double s2_value1 = sin( s2_incr * ( -z + 1 ));
	double c2_value1 = sin( s2_incr * ( -z + 1 ) + PI * 0.5 );
	int ks;

	oscbuf[ 0 ] = sin( s2_incr * -z );
	oscbuf[ 1 ] = sin( s2_incr * -z + PI * 0.5 );

	for( ks = 1; ks < z; ks++ )
	{
		const int ks2 = ks * 2;
		const double s1_value1 = oscbuf[ ks2 ];
		const double c1_value1 = oscbuf[ ks2 + 1 ];
		const double x = PI * ( ks - z );
		const double v0 = winbuf[ ks - 1 ] / (( x1 - x2 ) * x );

		oscbuf[ ks2 ] = s2_value1;
		oscbuf[ ks2 + 1 ] = c2_value1;

		kb1[ ks - 1 ] = ( x2 * s2_value1 - x1 * s1_value1 +
			( c2_value1 - c1_value1 ) / x ) * v0;

		kb2[ ks - 1 ] = ( s2_value1 - s1_value1 ) * v0;

		s2_value1 = s2_coeff * s2_value1 - oscbuf[ ks2 - 2 ];
		c2_value1 = s2_coeff * c2_value1 - oscbuf[ ks2 - 1 ];
	}

	kb1[ z - 1 ] = ( x2 * x2 - x1 * x1 ) / ( x1 - x2 ) * 0.5;
	kb2[ z - 1 ] = -1.0;
}

static void eq_init( fireq* q, double samplerate, double filterlength,
	int bandcount, double maxfreq, double wfalpha )
{
	double* oscbuf;
	double* winbuf;
	pcwin wf;
	double m, mo, f, x1, x2;
	double* kb1;
	double* kb2;
	int i;

	q -> bandcount = bandcount;
	q -> z = (int) ceil( filterlength * 0.5 );
	q -> zi = q -> z + ( q -> z & 1 );
	q -> z2 = q -> z * 2;
	q -> center = (double*) xalloc( sizeof( double ) * bandcount );
	oscbuf = (double*) xalloc( sizeof( double ) * q -> z2 );

	for( i = 0; i < q -> z; i++ )
	{
		oscbuf[ i * 2 ] = 0.0;
		oscbuf[ i * 2 + 1 ] = 1.0;
	}

	winbuf = (double*) xalloc( sizeof( double ) * q -> z );
	pw_init( &wf, wfalpha, filterlength * 0.5 );

	for( i = 1; i <= q -> z; i++ )
	{
		winbuf[ q -> z - i ] = pw_next( &wf );
	}

	q -> k1 = (double*) xalloc( sizeof( double ) * q -> zi * bandcount );
	q -> k2 = (double*) xalloc( sizeof( double ) * q -> zi * bandcount );

	m = 1.0;
	mo = ( maxfreq - 0.0 ) / ( bandcount - 1 );
	f = 0.0;
	x1 = 0.0;
	q -> center[ 0 ] = 0.0;
	f = f * m + mo;
	kb1 = q -> k1;
	kb2 = q -> k2;

	for( i = 1; i < bandcount; i++ )
	{
		x2 = f * 2.0 / samplerate;
		q -> center[ i ] = x2;
		eq_fill_band( q, x1, x2, kb1, kb2, oscbuf, winbuf );
		kb1 += q -> zi;
		kb2 += q -> zi;
		x1 = x2;
		f = f * m + mo;
	}

	if( x1 < 1.0 )
	{
		q -> uselast = 1;
		eq_fill_band( q, x1, 1.0, kb1, kb2, oscbuf, winbuf );
	}
	else
	{
		q -> uselast = 0;
	}

	free( oscbuf );
	free( winbuf );
}

static void eq_free( fireq* q )
{
	free( q -> center );
	free( q -> k1 );
	free( q -> k2 );
}

/* buildFilter, avir.h:1247-1304. */
static void eq_build( const fireq* q, const double* gains, double* filter )
{
	const int z = q -> z;
	const double* kb1 = q -> k1;
	const double* kb2 = q -> k2;
	double x1 = 0.0;
	double y1 = gains[ 0 ];
	double x2 = q -> center[ 1 ];
	double y2 = gains[ 1 ];
	double c = y1 - y2;
	double d = x1 * y2 - x2 * y1;
	int i, ks;

	for( ks = 0; ks < z; ks++ )
	{
		filter[ ks ] = c * kb1[ ks ] + d * kb2[ ks ];
	}

	kb1 += q -> zi;
	kb2 += q -> zi;
	x1 = x2;
	y1 = y2;

	for( i = 2; i < q -> bandcount; i++ )
	{
		x2 = q -> center[ i ];
		y2 = gains[ i ];
		c = y1 - y2;
		d = x1 * y2 - x2 * y1;

		for( ks = 0; ks < z; ks++ )
		{
			filter[ ks ] += c * kb1[ ks ] + d * kb2[ ks ];
		}

		kb1 += q -> zi;
		kb2 += q -> zi;
		x1 = x2;
		y1 = y2;
	}

	if( q -> uselast )
	{
		c = y1 - y2;
		d = x1 * y2 - y1;

		for( ks = 0; ks < z; ks++ )
		{
			filter[ ks ] += c * kb1[ ks ] + d * kb2[ ks ];
		}
	}

	for( i = 0; i < z - 1; i++ )
	{
		filter[ z + i ] = filter[ z - 2 - i ];
	}
}

/* ---- CFltBuffer, avir.h:1600-1629 ---- */

typedef struct
{
	double len2, freq, alpha, dcgain;
	double* p;
	int cap;
} fltbuf;

static void fb_copy( fltbuf* d, const fltbuf* s )
{
	free( d -> p );
	*d = *s;
	d -> p = NULL;

	if( s -> cap > 0 )
	{
		d -> p = (double*) xalloc( sizeof( double ) * s -> cap );
		memcpy( d -> p, s -> p, sizeof( double ) * s -> cap );
	}
}

static int fb_eq( const fltbuf* a, const fltbuf* b )
{
	return( a -> len2 == b -> len2 && a -> freq == b -> freq &&
		a -> alpha == b -> alpha && a -> dcgain == b -> dcgain );
}

/* ---- CDSPFracFilterBankLin<float>, avir.h:1647-2100 ---- */

typedef struct
{
	double wflen2, wffreq, wfalpha;
	int fraccount, order, srcfilterlen, filterlen, filtersize;
	int isinitrequired, issrctablebuilt;
	fltbuf ext;
	double* srctable;
	float* table;
	char* flags;
	int flagscap;
} bank;

static void bank_ctor( bank* b )
{
	memset( b, 0, sizeof( *b ));
	b -> order = -1;
}

static void bank_dtor( bank* b )
{
	free( b -> ext.p );
	free( b -> srctable );
	free( b -> table );
	free( b -> flags );
}

/* copyInitParams, avir.h:1668-1691. */
static void bank_copy_init_params( bank* b, const bank* s )
{
	int i;
	b -> wflen2 = s -> wflen2;
	b -> wffreq = s -> wffreq;
	b -> wfalpha = s -> wfalpha;
	b -> fraccount = s -> fraccount;
	b -> order = s -> order;
	b -> srcfilterlen = s -> srcfilterlen;
	b -> filterlen = s -> filterlen;
	b -> filtersize = s -> filtersize;
	b -> issrctablebuilt = 0;
	fb_copy( &b -> ext, &s -> ext );
	free( b -> flags );
	b -> flagscap = s -> flagscap;
	b -> flags = (char*) xalloc( (size_t) s -> flagscap );

	for( i = 0; i < s -> flagscap; i++ )
	{
		b -> flags[ i ] = (char) ( s -> flags[ i ] << 2 );
	}
}

/* operator ==, avir.h:1702-1707. */
static int bank_eq( const bank* a, const bank* s )
{
	return( a -> order == s -> order && a -> wflen2 == s -> wflen2 &&
		a -> wffreq == s -> wffreq && a -> wfalpha == s -> wfalpha &&
		a -> fraccount == s -> fraccount && fb_eq( &a -> ext, &s -> ext ));
}

/* init, avir.h:1732-1772. */
static void bank_init( bank* b, int reqfraccount, int reqorder,
	double baselen, double cutoff, double wfalpha, const fltbuf* ext )
{
	const double newlen2 = 0.5 * baselen * reqfraccount;
	const double newfreq = PI * cutoff / reqfraccount;
	const double newalpha = wfalpha;

	if( reqorder == b -> order && newlen2 == b -> wflen2 &&
		newfreq == b -> wffreq && newalpha == b -> wfalpha &&
		reqfraccount == b -> fraccount && fb_eq( ext, &b -> ext ))
	{
		b -> isinitrequired = 0;
		return;
	}

	b -> wflen2 = newlen2;
	b -> wffreq = newfreq;
	b -> wfalpha = newalpha;
	b -> fraccount = reqfraccount;
	b -> order = reqorder;
	fb_copy( &b -> ext, ext );

	b -> srcfilterlen = ( lpf_fl2( b -> wflen2 ) / reqfraccount + 1 ) * 2;
	b -> filterlen = b -> srcfilterlen;

	if( b -> ext.cap > 0 )
	{
		b -> filterlen += b -> ext.cap - 1;
	}

	b -> filtersize = b -> filterlen * ( reqorder + 1 );
	b -> issrctablebuilt = 0;
	b -> isinitrequired = 1;
}

/* buildSrcTable, avir.h:1970-2009. */
static void bank_build_src_table( bank* b )
{
	const int fl2 = lpf_fl2( b -> wflen2 );
	const int buflen = b -> srcfilterlen * b -> fraccount + 1;
	const int bufcenter = b -> srcfilterlen * b -> fraccount / 2;
	double* buf = (double*) xalloc( sizeof( double ) * buflen );
	double* op0;
	int i, j;

	b -> issrctablebuilt = 1;
	b -> isinitrequired = 0;

	lpf_generate( &buf[ bufcenter - fl2 ], b -> wflen2, b -> wffreq,
		b -> wfalpha, 0.0 );

	free( b -> srctable );
	b -> srctable = (double*) xalloc( sizeof( double ) *
		( b -> fraccount + 1 ) * b -> srcfilterlen );

	free( b -> flags );
	b -> flagscap = b -> fraccount + 1;
	b -> flags = (char*) xalloc( (size_t) b -> flagscap + 1 );
	op0 = b -> srctable;

	for( i = b -> fraccount; i >= 0; i-- )
	{
		const double* ip = buf + i;

		for( j = 0; j < b -> srcfilterlen; j++ )
		{
			op0[ j ] = ip[ 0 ];
			ip += b -> fraccount;
		}

		normalize_fir( op0, b -> srcfilterlen, 1.0 );
		op0 += b -> srcfilterlen;
	}

	free( b -> table );
	b -> table = (float*) xalloc( sizeof( float ) *
		( b -> fraccount + 2 ) * b -> filtersize );

	free( buf );
}

/* createFilter, avir.h:2021-2099. */
static void bank_create_filter( bank* b, int n )
{
	const int extcap = b -> ext.cap;
	const int reslatency = extcap / 2 + b -> srcfilterlen / 2;
	int reslen = b -> srcfilterlen;
	int resoffs, i, j;
	float* op;
	const double* srcflt;

	if( b -> flags[ n ] != 0 )
	{
		return;
	}

	b -> flags[ n ] |= 1;

	if( extcap > 0 )
	{
		reslen += extcap - 1;
	}

	resoffs = b -> filterlen / 2 - reslatency;
	op = &b -> table[ (size_t) n * b -> filtersize ];

	for( i = 0; i < resoffs; i++ )
	{
		op[ i ] = 0;
	}

	for( i = resoffs + reslen; i < b -> filterlen; i++ )
	{
		op[ i ] = 0;
	}

	op += resoffs;
	srcflt = &b -> srctable[ (size_t) n * b -> srcfilterlen ];

	if( extcap == 0 )
	{
		for( i = 0; i < reslen; i++ )
		{
			op[ i ] = (float) srcflt[ i ];
		}

		return;
	}

	for( j = 0; j < reslen; j++ )
	{
		int k = 0;
		int l = j - extcap + 1;
		int r = l + extcap;
		const double* extfltb;
		const double* srcfltb;
		double s = 0.0;

		if( l < 0 )
		{
			k -= l;
			l = 0;
		}

		if( r > b -> srcfilterlen )
		{
			r = b -> srcfilterlen;
		}

		extfltb = b -> ext.p + k;
		srcfltb = srcflt + l;
		l = r - l;

		for( i = 0; i < l; i++ )
		{
			s += extfltb[ i ] * srcfltb[ i ];
		}

		op[ j ] = (float) s;
	}
}

/* getFilter, avir.h:1814-1846. */
static const float* bank_get_filter( bank* b, int i )
{
	float* res;

	if( !b -> issrctablebuilt )
	{
		bank_build_src_table( b );
	}

	res = &b -> table[ (size_t) i * b -> filtersize ];

	if(( b -> flags[ i ] & 2 ) == 0 )
	{
		bank_create_filter( b, i );
		b -> flags[ i ] |= 2;

		if( b -> order > 0 )
		{
			const float* res2 = res + b -> filtersize;
			float* op = res + b -> filterlen;
			int j;

			bank_create_filter( b, i + 1 );

			for( j = 0; j < b -> filterlen; j++ )
			{
				op[ j ] = res2[ j ] - res[ j ];
			}
		}
	}

	return( res );
}

/* calcInitComplexity, avir.h:1895-1929. */
static int bank_init_complexity( const bank* b, const char* fracusemap )
{
	const int fltinitcost = 65;
	const int fltusecost = b -> filterlen * b -> order +
		b -> srcfilterlen * b -> ext.cap;
	int ucb[ 2 ];
	int ic, i;

	ucb[ 0 ] = 0;
	ucb[ 1 ] = fltusecost;

	if( b -> isinitrequired )
	{
		ic = b -> fraccount * b -> srcfilterlen * fltinitcost;

		for( i = 0; i < b -> fraccount; i++ )
		{
			ic += ucb[ (size_t) fracusemap[ i ]];
		}
	}
	else
	{
		ic = 0;

		for( i = 0; i < b -> fraccount; i++ )
		{
			if( fracusemap[ i ] != 0 )
			{
				ic += ucb[ b -> flags[ i ] == 0 ? 1 : 0 ];
			}
		}
	}

	return( ic );
}

/* ---- filtering steps, avir.h:2568-2728 ---- */

typedef struct
{
	int srcposint, fti, srcoffs, fl;
	const float* ftp;
	float x;
} rpos;

typedef struct
{
	int isupsample, resamplefactor;
	float* flt;
	int fltcap;
	fltbuf fltorig;
	double dcgain;
	int fltlatency;
	int inlen, inbuf, inprefix, insuffix;
	int outlen, outbuf, outprefix, outsuffix;
	float* prefixdc;
	int prefixdccap;
	float* suffixdc;
	int suffixdccap;
	int edgepixelcount;
	rpos* rp;
	bank* fltbank;
	bank* fltbankdyn;
} step;

#define MAXSTEPS 16

typedef struct
{
	step s[ MAXSTEPS ];
	int n;
} steps;

static void steps_clear( steps* S )
{
	int i;

	for( i = 0; i < S -> n; i++ )
	{
		free( S -> s[ i ].flt );
		free( S -> s[ i ].fltorig.p );
		free( S -> s[ i ].prefixdc );
		free( S -> s[ i ].suffixdc );
		free( S -> s[ i ].rp );
	}

	memset( S, 0, sizeof( *S ));
}

static step* steps_add( steps* S )
{
	step* fs;

	if( S -> n >= MAXSTEPS )
	{
		abort();
	}

	fs = &S -> s[ S -> n++ ];
	memset( fs, 0, sizeof( *fs ));
	return( fs );
}

typedef struct
{
	/* CImageResizerVarsBase, avir.h:2473-2506 (the parts in use) */
	int elcount, elcountio;
	int buflen[ 2 ], bufoffs[ 2 ];
	double k, o;
	int resizestep, isresize2;
} vars;

typedef struct
{
	double p[ 8 ]; /* CorrFltAlpha, CorrFltLen, IntFltAlpha, IntFltCutoff,
		IntFltLen, LPFltAlpha, LPFltBaseLen, LPFltCutoffMult */
	double hbalpha, hbcutoff, hblen;
	int srcbitdepth, resbitdepth;
	bank fixed;
} resizer;

#define P_CORRALPHA 0
#define P_CORRLEN 1
#define P_INTALPHA 2
#define P_INTCUTOFF 3
#define P_INTLEN 4
#define P_LPALPHA 5
#define P_LPBASELEN 6
#define P_LPCUTOFFMULT 7

/* initFilterBank, avir.h:5128-5164. */
static void init_filter_bank( const resizer* R, bank* b, double cutoffmult,
	int forcehiorder, const fltbuf* ext )
{
	const int intbitdepth = ( R -> resbitdepth > R -> srcbitdepth ?
		R -> resbitdepth : R -> srcbitdepth );

	const double snr = -6.02 * ( intbitdepth + 3 );
	int useorder, fraccount;

	if( forcehiorder || intbitdepth > 8 )
	{
		useorder = 1;
		fraccount = (int) ceil( 0.23134052 * exp( -0.058062929 * snr ));
	}
	else
	{
		useorder = 0;
		fraccount = (int) ceil( 0.33287686 * exp( -0.11334583 * snr ));
	}

	if( fraccount < 2 )
	{
		fraccount = 2;
	}

	bank_init( b, fraccount, useorder, R -> p[ P_INTLEN ] / cutoffmult,
		R -> p[ P_INTCUTOFF ] * cutoffmult, R -> p[ P_INTALPHA ], ext );
}

/* assignFilterParams, avir.h:5231-5360. */
static void assign_filter_params( const resizer* R, step* fs, int isupsample,
	int resamplefactor, double fltcutoff, double dcgain, int usefltorig,
	int ismodel )
{
	double fltalpha, len2, freq;
	int fl2, flen, i;

	if( fltcutoff == 0.0 )
	{
		const double m = 2.0 / resamplefactor;
		fltalpha = R -> hbalpha;
		len2 = 0.5 * R -> hblen / m;
		freq = PI * R -> hbcutoff * m;
	}
	else
	{
		fltalpha = R -> p[ P_LPALPHA ];
		len2 = 0.25 * R -> p[ P_LPBASELEN ] / fltcutoff;
		freq = PI * R -> p[ P_LPCUTOFFMULT ] * fltcutoff;
	}

	if( isupsample )
	{
		len2 *= resamplefactor;
		freq /= resamplefactor;
		fs -> dcgain = dcgain * resamplefactor;
	}
	else
	{
		fs -> dcgain = dcgain;
	}

	fs -> fltorig.len2 = len2;
	fs -> fltorig.freq = freq;
	fs -> fltorig.alpha = fltalpha;
	fs -> fltorig.dcgain = fs -> dcgain;

	fl2 = lpf_fl2( len2 );
	flen = fl2 + fl2 + 1;
	fs -> isupsample = isupsample;
	fs -> resamplefactor = resamplefactor;
	fs -> fltlatency = fl2;
	fs -> fltcap = flen;

	if( ismodel )
	{
		if( usefltorig )
		{
			fs -> fltorig.p = (double*) xalloc( sizeof( double ) * flen );
			fs -> fltorig.cap = flen;
		}
	}
	else
	{
		fs -> fltorig.p = (double*) xalloc( sizeof( double ) * flen );
		fs -> fltorig.cap = flen;
		lpf_generate( fs -> fltorig.p, len2, freq, fltalpha, fs -> dcgain );

		fs -> flt = (float*) xalloc( sizeof( float ) * flen );

		for( i = 0; i < flen; i++ )
		{
			fs -> flt[ i ] = (float) fs -> fltorig.p[ i ];
		}

		if( !usefltorig )
		{
			free( fs -> fltorig.p );
			fs -> fltorig.p = NULL;
			fs -> fltorig.cap = 0;
		}
	}

	if( isupsample )
	{
		int l = fs -> fltcap - fs -> fltlatency - resamplefactor;
		const float* ip;
		float* op;

		fs -> prefixdccap = l;
		fs -> suffixdccap = fs -> fltlatency;

		if( ismodel )
		{
			return;
		}

		fs -> prefixdc = (float*) xalloc( sizeof( float ) * ( l > 0 ? l : 1 ));
		fs -> suffixdc = (float*) xalloc( sizeof( float ) *
			( fs -> fltlatency > 0 ? fs -> fltlatency : 1 ));

		ip = &fs -> flt[ fs -> fltlatency + resamplefactor ];

		for( i = 0; i < l; i++ )
		{
			fs -> prefixdc[ i ] = ip[ i ];
		}

		while( 1 )
		{
			ip += resamplefactor;
			l -= resamplefactor;

			if( l <= 0 )
			{
				break;
			}

			for( i = 0; i < l; i++ )
			{
				fs -> prefixdc[ i ] += ip[ i ];
			}
		}

		l = fs -> fltlatency;
		op = fs -> suffixdc;

		for( i = 0; i < l; i++ )
		{
			op[ i ] = fs -> flt[ i ];
		}

		while( 1 )
		{
			op += resamplefactor;
			l -= resamplefactor;

			if( l <= 0 )
			{
				break;
			}

			for( i = 0; i < l; i++ )
			{
				op[ i ] += fs -> flt[ i ];
			}
		}
	}
	else
	if( !usefltorig )
	{
		fs -> edgepixelcount = 3; /* EdgePixelCountDef, avir.h:2629 */
	}
}

/* addCorrectionFilter, avir.h:5384-5506. */
static void add_correction_filter( const resizer* R, steps* S, double bw,
	int isprecorrection, int ismodel )
{
	step* nfs = ( isprecorrection ? &S -> s[ 0 ] : steps_add( S ));
	const int bincount = 65;
	const int bincount1 = bincount - 1;
	double curbw = 1.0;
	double bins[ 65 ];
	double re, im;
	double* filter;
	fireq eq;
	int i, j, si, flen;

	nfs -> isupsample = 0;
	nfs -> resamplefactor = 1;
	nfs -> dcgain = 1.0;
	nfs -> edgepixelcount = ( isprecorrection ? 3 : 0 );

	if( ismodel )
	{
		/* CDSPFIREQ::calcFilterLength, avir.h:1316-1322 */
		const int l = (int) ceil( R -> p[ P_CORRLEN ] * 0.5 );
		nfs -> fltlatency = l - 1;
		nfs -> fltcap = l * 2 - 1;
		return;
	}

	for( j = 0; j < bincount; j++ )
	{
		bins[ j ] = 1.0;
	}

	si = ( isprecorrection ? 1 : 0 );

	for( i = si; i < S -> n - ( si ^ 1 ); i++ )
	{
		step* fs = &S -> s[ i ];
		const float* flt;
		int fltlen;
		double thm;

		if( fs -> isupsample )
		{
			curbw *= fs -> resamplefactor;

			if( fs -> fltorig.cap > 0 )
			{
				continue;
			}
		}

		if( fs -> resamplefactor == 0 )
		{
			if( fs -> fltbankdyn == NULL )
			{
				flt = &fs -> fltbank -> table[ 0 ];
				fltlen = fs -> fltbank -> filterlen;
			}
			else
			{
				flt = bank_get_filter( fs -> fltbankdyn, 0 );
				fltlen = fs -> fltbankdyn -> filterlen;
			}
		}
		else
		{
			flt = fs -> flt;
			fltlen = fs -> fltcap;
		}

		thm = PI * bw / ( curbw * bincount1 );

		for( j = 0; j < bincount; j++ )
		{
			fir_response( flt, fltlen, j * thm, &re, &im );
			bins[ j ] *= fs -> dcgain / sqrt( re * re + im * im );
		}

		if( !fs -> isupsample && fs -> resamplefactor > 1 )
		{
			curbw /= fs -> resamplefactor;
		}
	}

	eq_init( &eq, bw * 2.0, R -> p[ P_CORRLEN ], bincount, bw,
		R -> p[ P_CORRALPHA ]);

	nfs -> fltlatency = eq.z - 1;
	flen = eq.z2 - 1;
	filter = (double*) xalloc( sizeof( double ) * flen );
	eq_build( &eq, bins, filter );
	normalize_fir( filter, flen, 1.0 );

	nfs -> fltcap = flen;
	nfs -> flt = (float*) xalloc( sizeof( float ) * flen );

	for( i = 0; i < flen; i++ )
	{
		nfs -> flt[ i ] = (float) filter[ i ];
	}

	free( filter );
	eq_free( &eq );
}

/* buildFilterSteps, avir.h:5616-5739. */
static void build_filter_steps( resizer* R, steps* S, vars* V, bank* fltbank,
	double dcgain, int modeflags, int ismodel )
{
	const int dofltandintcombo = (( modeflags & 1 ) != 0 );
	const int forcehiorderint = (( modeflags & 2 ) != 0 );
	const int usehalfband = (( modeflags & 4 ) != 0 );
	const double bw = 1.0 / V -> k;
	const int upsamplefactor = ( (int) floor( V -> k ) < 2 ? 2 : 1 );
	double intcutoffmult;
	step* reusestep;
	step* extfltstep;
	int isprecorrection;
	double fltcutoff, corrbw;
	step* fs;
	fltbuf emptyext;

	steps_clear( S );
	memset( &emptyext, 0, sizeof( emptyext ));

	if( V -> k <= 1.0 )
	{
		isprecorrection = 1;
		fltcutoff = 1.0;
		corrbw = 1.0;
		steps_add( S );
	}
	else
	{
		isprecorrection = 0;
		fltcutoff = bw;
		corrbw = bw;
	}

	if( upsamplefactor > 1 )
	{
		fs = steps_add( S );
		assign_filter_params( R, fs, 1, upsamplefactor, fltcutoff, dcgain,
			dofltandintcombo, ismodel );

		intcutoffmult = fltcutoff * 2.0 / upsamplefactor;
		reusestep = NULL;
		extfltstep = ( dofltandintcombo ? fs : NULL );
	}
	else
	{
		int downsamplefactor;

		while( 1 )
		{
			downsamplefactor = (int) floor( 0.5 / fltcutoff );

			if( usehalfband && downsamplefactor > 1 )
			{
				assign_filter_params( R, steps_add( S ), 0, downsamplefactor,
					0.0, 1.0, 0, ismodel );

				fltcutoff *= downsamplefactor;
			}
			else
			{
				if( downsamplefactor < 1 )
				{
					downsamplefactor = 1;
				}

				break;
			}
		}

		fs = steps_add( S );
		assign_filter_params( R, fs, 0, downsamplefactor, fltcutoff, dcgain,
			dofltandintcombo, ismodel );

		intcutoffmult = fltcutoff / 0.5;

		if( dofltandintcombo )
		{
			reusestep = fs;
			extfltstep = fs;
		}
		else
		{
			intcutoffmult *= downsamplefactor;
			reusestep = NULL;
			extfltstep = NULL;
		}
	}

	fs = ( reusestep == NULL ? steps_add( S ) : reusestep );
	V -> resizestep = S -> n - 1;
	fs -> isupsample = 0;
	fs -> resamplefactor = 0;
	fs -> dcgain = ( extfltstep == NULL ? 1.0 : extfltstep -> dcgain );

	init_filter_bank( R, fltbank, intcutoffmult, forcehiorderint,
		( extfltstep == NULL ? &emptyext : &extfltstep -> fltorig ));

	if( bank_eq( fltbank, &R -> fixed ))
	{
		fs -> fltbank = &R -> fixed;
		fs -> fltbankdyn = NULL;
	}
	else
	{
		fs -> fltbank = fltbank;
		fs -> fltbankdyn = fltbank;
	}

	add_correction_filter( R, S, corrbw, isprecorrection, ismodel );
}

/* updateFilterStepBuffers + fillRPosBuf + extendUpsample,
 * avir.h:5753-5937. */
static void update_filter_step_buffers( steps* S, vars* V, int srclen,
	int newlen )
{
	int upstep = -1;
	int inbuf = 0;
	int i, j;

	for( i = 0; i < S -> n; i++ )
	{
		step* fs = &S -> s[ i ];
		fs -> inlen = srclen;
		fs -> inbuf = inbuf;
		fs -> outbuf = ( inbuf + 1 ) & 1;

		if( fs -> isupsample )
		{
			int l0, l;
			upstep = i;
			V -> k *= fs -> resamplefactor;
			V -> o *= fs -> resamplefactor;
			fs -> inprefix = 0;
			fs -> insuffix = 0;
			fs -> outlen = fs -> inlen * fs -> resamplefactor;
			fs -> outprefix = fs -> fltlatency;
			fs -> outsuffix = fs -> fltcap - fs -> fltlatency -
				fs -> resamplefactor;

			l0 = fs -> outprefix + fs -> outlen + fs -> outsuffix;
			l = fs -> inlen * fs -> resamplefactor + fs -> suffixdccap;

			if( l > l0 )
			{
				fs -> outsuffix += l - l0;
			}

			l0 = fs -> outlen + fs -> outsuffix;

			if( fs -> prefixdccap > l0 )
			{
				fs -> outsuffix += fs -> prefixdccap - l0;
			}
		}
		else
		if( fs -> resamplefactor == 0 )
		{
			const int filterlend2 = fs -> fltbank -> filterlen / 2;
			const int filterlend21 = filterlend2 - 1;
			const int resizelpix = (int) floor( V -> o ) - filterlend21;
			const int resizerpix = (int) floor( V -> o +
				( newlen - 1 ) * V -> k ) + filterlend2 + 1;

			const int fraccount = fs -> fltbank -> fraccount;

			fs -> inprefix = ( resizelpix < 0 ? -resizelpix : 0 );
			fs -> insuffix = ( resizerpix > fs -> inlen ?
				resizerpix - fs -> inlen : 0 );

			fs -> outlen = newlen;
			free( fs -> rp );
			fs -> rp = (rpos*) xalloc( sizeof( rpos ) * newlen );

			for( j = 0; j < newlen; j++ )
			{
				const double srcpos = V -> o + V -> k * j;
				const int srcposint = (int) floor( srcpos );
				const double x = ( srcpos - srcposint ) * fraccount;
				const int fti = (int) x;
				fs -> rp[ j ].x = (float) ( x - fti );
				fs -> rp[ j ].fti = fti;
				fs -> rp[ j ].srcposint = srcposint;
			}
		}
		else
		{
			V -> k /= fs -> resamplefactor;
			V -> o /= fs -> resamplefactor;
			V -> o += fs -> edgepixelcount;

			fs -> inprefix = fs -> fltlatency;
			fs -> insuffix = fs -> fltcap - fs -> fltlatency - 1;

			fs -> outlen = ( fs -> inlen + fs -> resamplefactor - 1 ) /
				fs -> resamplefactor + fs -> edgepixelcount;

			fs -> insuffix += ( fs -> outlen - 1 ) * fs -> resamplefactor +
				1 - fs -> inlen;

			fs -> inprefix += fs -> edgepixelcount * fs -> resamplefactor;
			fs -> outlen += fs -> edgepixelcount;
		}

		inbuf = fs -> outbuf;
		srclen = fs -> outlen;
	}

	S -> s[ S -> n - 1 ].outbuf = 2;
	V -> isresize2 = 0;

	if( upstep != -1 )
	{
		step* fs = &S -> s[ upstep ];
		step* nx = &S -> s[ upstep + 1 ];

		fs -> inprefix = ( nx -> inprefix + fs -> resamplefactor - 1 ) /
			fs -> resamplefactor;

		fs -> outprefix += fs -> inprefix * fs -> resamplefactor;
		nx -> inprefix = 0;

		fs -> insuffix = ( nx -> insuffix + fs -> resamplefactor - 1 ) /
			fs -> resamplefactor;

		fs -> outsuffix += fs -> insuffix * fs -> resamplefactor;
		nx -> insuffix = 0;

		if( fs -> resamplefactor == 2 && V -> resizestep == upstep + 1 &&
			fs -> fltorig.cap > 0 )
		{
			V -> isresize2 = 1;
		}
	}
}

/* updateBufLenAndRPosPtrs, avir.h:5957-6127 (interleaved mode). */
static void update_buflen_and_rpos_ptrs( steps* S, vars* V )
{
	int maxprefix[ 2 ] = { 0, 0 };
	int maxlen[ 2 ] = { 0, 0 };
	step* fs;
	int i, fl, filterlend21;

	for( i = 0; i < S -> n; i++ )
	{
		int l, ib, ob;
		fs = &S -> s[ i ];
		ib = fs -> inbuf;

		if( fs -> inprefix > maxprefix[ ib ])
		{
			maxprefix[ ib ] = fs -> inprefix;
		}

		l = fs -> inlen + fs -> insuffix;

		if( l > maxlen[ ib ])
		{
			maxlen[ ib ] = l;
		}

		if( fs -> outbuf == 2 )
		{
			break;
		}

		ob = fs -> outbuf;

		if( fs -> isupsample )
		{
			if( fs -> outprefix > maxprefix[ ob ])
			{
				maxprefix[ ob ] = fs -> outprefix;
			}

			l = fs -> outlen + fs -> outsuffix;

			if( l > maxlen[ ob ])
			{
				maxlen[ ob ] = l;
			}
		}
		else
		{
			if( fs -> outlen > maxlen[ ob ])
			{
				maxlen[ ob ] = fs -> outlen;
			}
		}
	}

	for( i = 0; i < 2; i++ )
	{
		V -> buflen[ i ] = ( maxprefix[ i ] + maxlen[ i ]) * V -> elcount;
		V -> bufoffs[ i ] = maxprefix[ i ] * V -> elcount;
	}

	fs = &S -> s[ V -> resizestep ];
	fl = fs -> fltbank -> filterlen;
	filterlend21 = fl / 2 - 1;

	for( i = 0; i < fs -> outlen; i++ )
	{
		rpos* rp = &fs -> rp[ i ];
		const float* flt = ( fs -> fltbankdyn == NULL ?
			&fs -> fltbank -> table[ (size_t) rp -> fti *
			fs -> fltbank -> filtersize ] :
			bank_get_filter( fs -> fltbankdyn, rp -> fti ));

		if( V -> isresize2 )
		{
			const int p = rp -> srcposint - filterlend21;
			const int fo = p & 1;
			rp -> srcoffs = ( p + fo ) * V -> elcount;
			rp -> ftp = flt + fo;
			rp -> fl = fl - fo;
		}
		else
		{
			rp -> srcoffs = ( rp -> srcposint - filterlend21 ) * V -> elcount;
			rp -> ftp = flt;
			rp -> fl = fl;
		}
	}
}

/* modifyCorrFilterDCGain, avir.h:6137-6157. */
static void modify_corr_filter_dc_gain( steps* S, double m )
{
	const int z = S -> n - 1;
	step* fs;
	int i;

	if( !S -> s[ z ].isupsample && S -> s[ z ].resamplefactor == 1 )
	{
		fs = &S -> s[ z ];
	}
	else
	{
		fs = &S -> s[ 0 ];
	}

	for( i = 0; i < fs -> fltcap; i++ )
	{
		fs -> flt[ i ] = (float) ( (double) fs -> flt[ i ] * m );
	}
}

/* fillUsedFracMap + calcComplexity, avir.h:6167-6270. */
static int calc_complexity( const steps* S, const vars* V, int scanlinecount )
{
	const step* rs = &S -> s[ V -> resizestep ];
	const int fraccount = rs -> fltbank -> fraccount;
	char* usedfracmap = (char*) xalloc( (size_t) fraccount + 1 );
	const int fcnum = 3;
	const int fcdenom = 4;
	int s = 0;
	int s2 = 0;
	int i;

	for( i = 0; i < rs -> outlen; i++ )
	{
		usedfracmap[ rs -> rp[ i ].fti ] |= 1;
	}

	for( i = 0; i < S -> n; i++ )
	{
		const step* fs = &S -> s[ i ];
		s2 += 65 * fs -> fltcap;

		if( fs -> isupsample )
		{
			if( fs -> fltorig.cap > 0 )
			{
				continue;
			}

			s += ( fs -> fltcap *
				( fs -> inprefix + fs -> inlen + fs -> insuffix ) +
				fs -> suffixdccap + fs -> prefixdccap ) * V -> elcount;
		}
		else
		if( fs -> resamplefactor == 0 )
		{
			s += fs -> fltbank -> filterlen *
				( fs -> fltbank -> order + V -> elcount ) * fs -> outlen;

			if( i == V -> resizestep && V -> isresize2 )
			{
				s >>= 1;
			}

			s2 += bank_init_complexity( fs -> fltbank, usedfracmap );
		}
		else
		{
			s += fs -> fltcap * V -> elcount * fs -> outlen * fcnum / fcdenom;
		}
	}

	free( usedfracmap );
	return( s + s2 / scanlinecount );
}

/* ---- scanline executor (CImageResizerFilterStepINL, avir.h:2741-4328),
 * generic in ElCount ---- */

/* prepareInBuf, avir.h:3227-3239. */
static void prepare_in_buf( const step* fs, float* src, int el )
{
	int i, c;
	float* e;

	if( fs -> isupsample || fs -> inprefix + fs -> insuffix == 0 )
	{
		return;
	}

	for( i = 1; i <= fs -> inprefix; i++ )
	{
		for( c = 0; c < el; c++ )
		{
			src[ -i * el + c ] = src[ c ];
		}
	}

	e = src + ( fs -> inlen - 1 ) * el;

	for( i = 1; i <= fs -> insuffix; i++ )
	{
		for( c = 0; c < el; c++ )
		{
			e[ i * el + c ] = e[ c ];
		}
	}
}

/* doUpsample, avir.h:3249-3734. */
static void do_upsample( const step* fs, const float* src, float* dst, int el )
{
	float* op0 = &dst[ -fs -> outprefix * el ];
	const float* ip = src;
	const int opstep = el * fs -> resamplefactor;
	int l, i, c;
	float* op;
	const float* dc;

	memset( op0, 0, (size_t) ( fs -> outprefix + fs -> outlen +
		fs -> outsuffix ) * (size_t) el * sizeof( float ));

	if( fs -> fltorig.cap > 0 )
	{
		/* filterless: place samples, zero elsewhere (3260-3402) */
		op0 += ( fs -> outprefix % fs -> resamplefactor ) * el;
		l = fs -> outprefix / fs -> resamplefactor;

		while( l > 0 )
		{
			for( c = 0; c < el; c++ ) op0[ c ] = ip[ c ];
			op0 += opstep;
			l--;
		}

		l = fs -> inlen - 1;

		while( l > 0 )
		{
			for( c = 0; c < el; c++ ) op0[ c ] = ip[ c ];
			op0 += opstep;
			ip += el;
			l--;
		}

		l = fs -> outsuffix / fs -> resamplefactor;

		while( l >= 0 )
		{
			for( c = 0; c < el; c++ ) op0[ c ] = ip[ c ];
			op0 += opstep;
			l--;
		}

		return;
	}

	l = fs -> inprefix;

	while( l > 0 )
	{
		op = op0;

		for( i = 0; i < fs -> fltcap; i++ )
		{
			for( c = 0; c < el; c++ ) op[ c ] += fs -> flt[ i ] * ip[ c ];
			op += el;
		}

		op0 += opstep;
		l--;
	}

	l = fs -> inlen - 1;

	while( l > 0 )
	{
		op = op0;

		for( i = 0; i < fs -> fltcap; i++ )
		{
			for( c = 0; c < el; c++ ) op[ c ] += fs -> flt[ i ] * ip[ c ];
			op += el;
		}

		ip += el;
		op0 += opstep;
		l--;
	}

	l = fs -> insuffix;

	while( l >= 0 )
	{
		op = op0;

		for( i = 0; i < fs -> fltcap; i++ )
		{
			for( c = 0; c < el; c++ ) op[ c ] += fs -> flt[ i ] * ip[ c ];
			op += el;
		}

		op0 += opstep;
		l--;
	}

	op = op0;
	dc = fs -> suffixdc;

	for( i = 0; i < fs -> suffixdccap; i++ )
	{
		for( c = 0; c < el; c++ ) op[ c ] += ip[ c ] * dc[ i ];
		op += el;
	}

	ip = src;
	op = dst - fs -> inprefix * opstep;
	dc = fs -> prefixdc;

	for( i = 0; i < fs -> prefixdccap; i++ )
	{
		for( c = 0; c < el; c++ ) op[ c ] += ip[ c ] * dc[ i ];
		op += el;
	}
}

/* doFilter, avir.h:3748-3866. */
static void do_filter( const step* fs, const float* src, float* dst,
	int dstincr, int el )
{
	const float* f = &fs -> flt[ fs -> fltlatency ];
	const int flen = fs -> fltlatency + 1;
	const int ipstep = el * fs -> resamplefactor;
	const float* ip = src - fs -> edgepixelcount * ipstep;
	int l = fs -> outlen;
	int i, c;

	while( l > 0 )
	{
		for( c = 0; c < el; c++ )
		{
			float s = f[ 0 ] * ip[ c ];

			for( i = 1; i < flen; i++ )
			{
				s += f[ i ] * ( ip[ c + i * el ] + ip[ c - i * el ]);
			}

			dst[ c ] = s;
		}

		dst += dstincr;
		ip += ipstep;
		l--;
	}
}

/* doResize / doResize2, avir.h:3884-4328. `tapstep` is 1 or 2. */
static void do_resize( const step* fs, const float* srcline, float* dstline,
	int dstincr, int el, int tapstep )
{
	const int intfltlen0 = fs -> fltbank -> filterlen;
	const int order = fs -> fltbank -> order;
	int j, i, c;

	for( j = 0; j < fs -> outlen; j++ )
	{
		const rpos* rp = &fs -> rp[ j ];
		const float x = rp -> x;
		const float* ftp = rp -> ftp;
		const float* ftp2 = ftp + intfltlen0;
		const float* src = srcline + rp -> srcoffs;
		const int fl = rp -> fl;
		float sum[ 4 ] = { 0, 0, 0, 0 };

		for( i = 0; i < fl; i += tapstep )
		{
			float xx;

			if( order == 1 )
			{
				const float t = ftp2[ i ] * x;
				xx = ftp[ i ] + t;
			}
			else
			{
				xx = ftp[ i ];
			}

			for( c = 0; c < el; c++ )
			{
				sum[ c ] += xx * src[ c ];
			}

			src += el * tapstep;
		}

		for( c = 0; c < el; c++ )
		{
			dstline[ c ] = sum[ c ];
		}

		dstline += dstincr;
	}
}

/* CThreadData::resizeScanlineH/V step loop, avir.h:6522-6619. */
static void run_steps( const steps* S, const vars* V, float* bufptrs[ 3 ],
	int lastincr )
{
	const int el = V -> elcount;
	int j;

	for( j = 0; j < S -> n; j++ )
	{
		const step* fs = &S -> s[ j ];
		const int dstincr = ( fs -> outbuf == 2 ? lastincr : el );
		prepare_in_buf( fs, bufptrs[ fs -> inbuf ], el );

		if( fs -> resamplefactor != 0 )
		{
			if( fs -> isupsample )
			{
				do_upsample( fs, bufptrs[ fs -> inbuf ],
					bufptrs[ fs -> outbuf ], el );
			}
			else
			{
				do_filter( fs, bufptrs[ fs -> inbuf ],
					bufptrs[ fs -> outbuf ], dstincr, el );
			}
		}
		else
		{
			do_resize( fs, bufptrs[ fs -> inbuf ], bufptrs[ fs -> outbuf ],
				dstincr, el, ( V -> isresize2 ? 2 : 1 ));
		}
	}
}

/* avir::round, avir.h:130-135. */
static float avir_round( float d )
{
	return( d < 0.0f ? -(float) (int) ( 0.5f - d ) :
		(float) (int) ( d + 0.5f ));
}

static float load_in( const void* p, size_t i, int t )
{
	switch( t )
	{
		case ORACLE_U8: return( (float) ((const uint8_t*) p )[ i ]);
		case ORACLE_U16: return( (float) ((const uint16_t*) p )[ i ]);
		case ORACLE_F32: return( ((const float*) p )[ i ]);
		default: return( (float) ((const double*) p )[ i ]);
	}
}

static void store_out( void* p, size_t i, int t, float v )
{
	switch( t )
	{
		case ORACLE_U8: ((uint8_t*) p )[ i ] = (uint8_t) v; break;
		case ORACLE_U16: ((uint16_t*) p )[ i ] = (uint16_t) v; break;
		case ORACLE_F32: ((float*) p )[ i ] = v; break;
		default: ((double*) p )[ i ] = (double) v; break;
	}
}

static size_t tsize( int t )
{
	return( t == ORACLE_U8 ? 1 : t == ORACLE_U16 ? 2 : t == ORACLE_F32 ? 4 :
		8 );
}

/* resizeImage, avir.h:4680-5092. */
/* ---- sRGB gamma stages (CImageResizerVars::UseSRGBGamma) ------------- */

static int g_gamma = 0;
static int g_alpha = -1;

/* UseSRGBGamma / AlphaIndex of the following oracle_avir_resize() calls. */
void oracle_set_gamma( int use, int alpha_index )
{
	g_gamma = use;
	g_alpha = alpha_index;
}

static int g_dither = 0;

/* Ditherer class of the following oracle_avir_resize() calls: 0 =
 * CImageResizerDithererDefINL (rounding only), 1 =
 * CImageResizerDithererErrdINL (error diffusion), avir.h:4351-4537. */
void oracle_set_dither( int kind )
{
	g_dither = kind;
}

/* pow24_sRGB, avir.h:162-175: double arithmetic from a float argument. */
static double srgb_pow24( double x )
{
	const double x2 = x * x;
	const double x3 = x2 * x;
	const double x4 = x2 * x2;

	return( 0.0985766365536824 + 0.839474952656502 * x2 +
		0.363287814061725 * x3 - 0.0125559718896615 /
		( 0.12758338921578 + 0.290283465468235 * x ) -
		0.231757513261358 * x - 0.0395365717969074 * x4 );
}

/* pow24i_sRGB, avir.h:187-198. */
static double srgb_pow24i( double x )
{
	const double sx = sqrt( x );
	const double ssx = sqrt( sx );
	const double sssx = sqrt( ssx );

	return( 0.000213364515060263 + 0.0149409239419218 * x +
		0.433973412731747 * sx + ssx * ( 0.659628181609715 * sssx -
		0.0380957908841466 - 0.0706476137208521 * sx ));
}

/* convertSRGB2Lin<float,Tin>, avir.h:206-219. */
static float srgb_to_lin( float s0, float m )
{
	const float s = s0 * m;
	const float a = 0.055f;

	if( s <= 0.04045f )
	{
		return( s / 12.92f );
	}

	return( (float) srgb_pow24( (double) (( s + a ) / ( 1.0f + a ))));
}

/* convertLin2SRGB<float>, avir.h:301-312. */
static float lin_to_srgb( float s )
{
	const float a = 0.055f;

	if( s <= 0.0031308f )
	{
		return( 12.92f * s );
	}

	return(( 1.0f + a ) * (float) srgb_pow24i( (double) s ) - a );
}

/* The uint8_t specialisation of convertSRGB2Lin is a table of 256 literals
 * (avir.h:229-291): the double-precision evaluation of the approximation
 * written with 7 significant digits. Regenerated here; test_oracle.py checks
 * every entry against the reference's table. */
void oracle_srgb_u8_table( float* tbl )
{
	int i;

	for( i = 0; i < 256; i++ )
	{
		const double s = (double) i * ( 1.0 / 255.0 );
		const double a = 0.055;
		const double v = ( s <= 0.04045 ? s / 12.92 :
			srgb_pow24(( s + a ) / ( 1.0 + a )));

		char buf[ 64 ];
		snprintf( buf, sizeof( buf ), "%.7g", v );
		tbl[ i ] = strtof( buf, NULL );
	}
}

int oracle_avir_resize( const void* src, int sw, int sh, int sstride,
	void* dst, int nw, int nh, int ch, double k, int in_type, int out_type,
	int resbits, int srcbits, const double* params, double ox0, double oy0,
	int build_mode, int* mode_h, int* mode_v )
{
	static const double defp[ 8 ] = { 0.97946, 6.4262, 6.41341, 0.7372, 18,
		4.76449, 7.55999999999998, 0.79285 }; /* avir.h:2328-2341 */

	resizer R;
	vars V;
	bank fltbank;
	steps S;
	steps T;
	fltbuf emptyext;
	double kx, ky, ox, oy, outmul;
	int isinfloat, isoutfloat, usebuildmode, prevusebuildmode;
	int buildmodecount, m, i, j, c;
	float* fltbuf_;
	float* bufs;
	float* bp[ 3 ];
	float* resbuf = NULL;
	int neww_e;
	float ingm = 1.0f, outgm = 1.0f;
	float srgbtbl[ 256 ];
	const int alpha = ( ch == 4 ? g_alpha : -1 );

	if( ch < 1 || ch > 4 || in_type < 0 || in_type > 3 || out_type < 0 ||
		out_type > 3 )
	{
		return( -1 );
	}

	if( sw == 0 || sh == 0 )
	{
		memset( dst, 0, (size_t) nw * (size_t) nh * tsize( out_type ));
		return( 0 );
	}
	else
	if( nw == 0 || nh == 0 )
	{
		return( 0 );
	}

	/* constructor, avir.h:4630-4639 */
	memset( &R, 0, sizeof( R ));
	memcpy( R.p, ( params != NULL ? params : defp ), sizeof( R.p ));
	R.hbalpha = 1.94609;
	R.hbcutoff = 0.46437;
	R.hblen = 24;
	R.resbitdepth = resbits;
	R.srcbitdepth = ( srcbits == 0 ? resbits : srcbits );
	memset( &emptyext, 0, sizeof( emptyext ));
	bank_ctor( &R.fixed );
	init_filter_bank( &R, &R.fixed, 1.0, 0, &emptyext );

	for( i = 0; i < R.fixed.fraccount; i++ )
	{
		bank_get_filter( &R.fixed, i );
	}

	ox = ox0;
	oy = oy0;

	if( k == 0.0 )
	{
		kx = (double) sw / nw;
		ox += ( kx - 1.0 ) * 0.5;
		ky = (double) sh / nh;
		oy += ( ky - 1.0 ) * 0.5;
	}
	else
	if( k > 0.0 )
	{
		const double ko = ( k - 1.0 ) * 0.5;
		kx = k;
		ky = k;
		ox += ko;
		oy += ko;
	}
	else
	{
		kx = -k;
		ky = -k;
	}

	isinfloat = ( in_type >= ORACLE_F32 );
	isoutfloat = ( out_type >= ORACLE_F32 );

	/* avir.h:4740-4782 */
	if( g_gamma )
	{
		ingm = (float) ( isinfloat ? 1.0 :
			1.0 / ( in_type == ORACLE_U8 ? 255.0 : 65535.0 ));

		outgm = (float) ( isoutfloat ? 1.0 :
			( out_type == ORACLE_U8 ? 255.0 : 65535.0 ));

		outmul = 1.0;
		oracle_srgb_u8_table( srgbtbl );
	}
	else
	{
		if( isoutfloat )
		{
			outmul = 1.0;
		}
		else
		{
			outmul = ( out_type == ORACLE_U8 ? 255.0 : 65535.0 );
		}

		if( !isinfloat )
		{
			outmul /= ( in_type == ORACLE_U8 ? 255.0 : 65535.0 );
		}
	}

	neww_e = nw * ch;

	if( sstride < 1 )
	{
		sstride = sw * ch;
	}

	memset( &V, 0, sizeof( V ));
	V.elcount = ch;
	V.elcountio = ch;
	bank_ctor( &fltbank );
	memset( &S, 0, sizeof( S ));
	memset( &T, 0, sizeof( T ));

	/* horizontal: model all build modes, avir.h:4813-4847 */
	usebuildmode = 1;
	buildmodecount = ( R.fixed.order == 0 ? 4 : 2 );

	if( build_mode >= 0 )
	{
		usebuildmode = build_mode;
	}
	else
	{
		int bestscore = 0x7FFFFFFF;

		for( m = 0; m < buildmodecount; m++ )
		{
			bank tmpbank;
			int cc;
			bank_ctor( &tmpbank );
			V.k = kx;
			V.o = ox;
			build_filter_steps( &R, &T, &V, &tmpbank, outmul, m, 1 );
			update_filter_step_buffers( &T, &V, sw, nw );
			cc = calc_complexity( &T, &V, sh );

			if( cc < bestscore )
			{
				usebuildmode = m;
				bestscore = cc;
			}

			steps_clear( &T );
			bank_dtor( &tmpbank );
		}
	}

	V.k = kx;
	V.o = ox;
	build_filter_steps( &R, &S, &V, &fltbank, outmul, usebuildmode, 0 );
	update_filter_step_buffers( &S, &V, sw, nw );
	update_buflen_and_rpos_ptrs( &S, &V );

	if( mode_h != NULL )
	{
		*mode_h = usebuildmode;
	}

	/* horizontal pass into FltBuf, avir.h:4881-4894 */
	fltbuf_ = (float*) xalloc( sizeof( float ) * (size_t) neww_e * sh );
	bufs = (float*) xalloc( sizeof( float ) *
		( (size_t) V.buflen[ 0 ] + V.buflen[ 1 ]));

	bp[ 0 ] = bufs + V.bufoffs[ 0 ];
	bp[ 1 ] = bufs + V.buflen[ 0 ] + V.bufoffs[ 1 ];

	for( i = 0; i < sh; i++ )
	{
		/* packScanline, avir.h:2777-2840; gamma branch 2841-2930 */
		for( j = 0; j < sw * ch; j++ )
		{
			const float v = load_in( src, (size_t) i * sstride + j, in_type );

			if( !g_gamma )
			{
				bp[ 0 ][ j ] = v;
			}
			else
			if( j % ch == alpha )
			{
				bp[ 0 ][ j ] = v * ingm;
			}
			else
			if( in_type == ORACLE_U8 )
			{
				bp[ 0 ][ j ] = srgbtbl[ (int) v ];
			}
			else
			{
				bp[ 0 ][ j ] = srgb_to_lin( v, ingm );
			}
		}

		bp[ 2 ] = &fltbuf_[ (size_t) i * neww_e ];
		run_steps( &S, &V, bp, ch );
	}

	free( bufs );

	/* vertical: re-model, reuse steps if possible, avir.h:4899-4954 */
	prevusebuildmode = usebuildmode;

	if( build_mode >= 0 )
	{
		usebuildmode = build_mode;
	}
	else
	{
		vars TV = V;
		int bestscore = 0x7FFFFFFF;

		for( m = 0; m < buildmodecount; m++ )
		{
			bank tmpbank;
			int cc;
			bank_ctor( &tmpbank );
			bank_copy_init_params( &tmpbank, &fltbank );
			TV.k = ky;
			TV.o = oy;
			build_filter_steps( &R, &T, &TV, &tmpbank, 1.0, m, 1 );
			update_filter_step_buffers( &T, &TV, sh, nh );
			cc = calc_complexity( &T, &TV, nw );

			if( cc < bestscore )
			{
				usebuildmode = m;
				bestscore = cc;
			}

			steps_clear( &T );
			bank_dtor( &tmpbank );
		}
	}

	V.k = ky;
	V.o = oy;

	if( usebuildmode == prevusebuildmode && ky == kx )
	{
		if( outmul != 1.0 )
		{
			modify_corr_filter_dc_gain( &S, 1.0 / outmul );
		}
	}
	else
	{
		build_filter_steps( &R, &S, &V, &fltbank, 1.0, usebuildmode, 0 );
	}

	update_filter_step_buffers( &S, &V, sh, nh );
	update_buflen_and_rpos_ptrs( &S, &V );

	if( mode_v != NULL )
	{
		*mode_v = usebuildmode;
	}

	bufs = (float*) xalloc( sizeof( float ) *
		( (size_t) V.buflen[ 0 ] + V.buflen[ 1 ]));

	bp[ 0 ] = bufs + V.bufoffs[ 0 ];
	bp[ 1 ] = bufs + V.buflen[ 0 ] + V.bufoffs[ 1 ];

	/* float output is written in place by the vertical pass and returned as
	 * is, avir.h:4956-4979 -- with gamma it therefore stays LINEAR (only the
	 * other output types pass through applySRGBGamma) */
	if( out_type != ORACLE_F32 )
	{
		resbuf = (float*) xalloc( sizeof( float ) * (size_t) neww_e * nh );
	}

	/* vertical pass, avir.h:4956-5000 (in-place output for float) */
	for( i = 0; i < nw; i++ )
	{
		/* convertVtoH, avir.h:3085-3136 */
		const float* ip = &fltbuf_[ i * ch ];

		for( j = 0; j < sh; j++ )
		{
			for( c = 0; c < ch; c++ )
			{
				bp[ 0 ][ j * ch + c ] = ip[ c ];
			}

			ip += neww_e;
		}

		bp[ 2 ] = ( resbuf != NULL ? &resbuf[ i * ch ] :
			(float*) dst + i * ch );

		run_steps( &S, &V, bp, neww_e );
	}

	if( resbuf != NULL && g_gamma )
	{
		/* applySRGBGamma, avir.h:2982-3068 */
		for( j = 0; j < nh * neww_e; j++ )
		{
			if( j % ch == alpha )
			{
				resbuf[ j ] = resbuf[ j ] * outgm;
			}
			else
			{
				resbuf[ j ] = lin_to_srgb( resbuf[ j ]) * outgm;
			}
		}
	}

	if( resbuf != NULL )
	{
		if( isoutfloat )
		{
			/* unpackScanline to double, avir.h:3155-3215 */
			for( j = 0; j < nh * neww_e; j++ )
			{
				store_out( dst, (size_t) j, out_type, resbuf[ j ]);
			}
		}
		else
		{
			/* dither + unpack, avir.h:5029-5088, 4392-4419 */
			int truncbits, outrange;
			double pkout0, trmul0;
			float pkout;

			if( out_type == ORACLE_U8 )
			{
				truncbits = 8 - resbits;
				outrange = 255;
			}
			else
			{
				truncbits = 16 - resbits;
				outrange = 65535;
			}

			pkout0 = outrange;
			trmul0 = ( truncbits > 0 ?
				pkout0 / ( outrange >> truncbits ) : 1.0 );

			pkout = (float) pkout0;

			if( g_dither == 1 )
			{
				/* CImageResizerDithererErrdINL::dither, avir.h:4485-4525:
				 * rows in order, one diffusion row `dith` of LenE + ElCount
				 * floats (pointer past the first ElCount) kept between them */
				const float trmul = (float) trmul0;
				const float trmuli = (float) ( 1.0 / trmul0 );
				const int lene = neww_e;
				float* const dith0 = (float*) calloc( (size_t) lene + ch,
					sizeof( float ));
				float* const dith = dith0 + ch;
				int y;

				for( y = 0; y < nh; y++ )
				{
					float* const rs = &resbuf[ (size_t) y * lene ];

					for( j = 0; j < lene; j++ )
					{
						rs[ j ] += dith[ j ];
						dith[ j ] = 0.0f;
					}

					for( j = 0; j < lene - ch; j++ )
					{
						const float z0 = avir_round( rs[ j ] * trmuli ) * trmul;
						const float noise = rs[ j ] - z0;
						const float noisem1 = noise * (float) 0.364842;
						rs[ j ] = ( z0 < 0.0f ? 0.0f :
							( z0 > pkout ? pkout : z0 ));

						rs[ j + ch ] += noisem1;
						dith[ j - ch ] += noise * (float) 0.207305;
						dith[ j ] += noisem1;
						dith[ j + ch ] += noise * (float) 0.063011;
					}

					while( j < lene )
					{
						const float z0 = avir_round( rs[ j ] * trmuli ) * trmul;
						const float noise = rs[ j ] - z0;
						rs[ j ] = ( z0 < 0.0f ? 0.0f :
							( z0 > pkout ? pkout : z0 ));

						dith[ j - ch ] += noise * (float) 0.207305;
						dith[ j ] += noise * (float) 0.364842;
						j++;
					}

					for( j = 0; j < lene; j++ )
					{
						store_out( dst, (size_t) y * lene + j, out_type,
							rs[ j ]);
					}
				}

				free( dith0 );
			}
			else
			if( trmul0 == 1.0 )
			{
				for( j = 0; j < nh * neww_e; j++ )
				{
					float v = avir_round( resbuf[ j ]);
					v = ( v < 0.0f ? 0.0f : ( v > pkout ? pkout : v ));
					store_out( dst, (size_t) j, out_type, v );
				}
			}
			else
			{
				const float trmul = (float) trmul0;
				const float trmuli = (float) ( 1.0 / trmul0 );

				for( j = 0; j < nh * neww_e; j++ )
				{
					float v = avir_round( resbuf[ j ] * trmuli ) * trmul;
					v = ( v < 0.0f ? 0.0f : ( v > pkout ? pkout : v ));
					store_out( dst, (size_t) j, out_type, v );
				}
			}
		}

		free( resbuf );
	}

	free( bufs );
	free( fltbuf_ );
	steps_clear( &S );
	bank_dtor( &fltbank );
	bank_dtor( &R.fixed );
	return( 0 );
}
