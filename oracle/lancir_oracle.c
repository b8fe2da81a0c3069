/*
 * oracle/lancir_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of avir::CLancIR::resizeImage() (lancir.h:386-713) as
 * built for x86-64 with SSE2 (the canonical LANCIR oracle): the
 * 1000-phase normalised Lanczos filter cache (lancir.h:882-1218), scanline
 * positions (1290-1351), the vertical-first two-pass structure, and the SSE2
 * lane-wise summation orders of resize1..4 (2102-2544) written out as scalar
 * expressions, plus outputScanline's scaling / clamp / rounding (1772-2056:
 * groups of four elements round to nearest-even via cvtps, the 1-3 element
 * tail uses roundclamp's +0.5 truncation). Row batching (lancir.h:485-509) is
 * a cache optimisation with no effect on results and is not restated. Must be
 * compiled with -ffp-contract=off.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

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

/* CResizeFilters, lancir.h:840-1219 (taps stored un-replicated). */
typedef struct
{
	double freq, freqa, len2, k;
	int fl2, kernellen, fraccount;
	float* filters[ 1001 ];
} rfilters;

static void rf_update( rfilters* rf, double la, double k0 )
{
	const double normfreq = ( k0 <= 1.0 ? 1.0 : 1.0 / k0 );
	memset( rf, 0, sizeof( *rf ));
	rf -> freq = 3.1415926535897932 * normfreq;
	rf -> freqa = rf -> freq / la;
	rf -> len2 = la / normfreq;
	rf -> fl2 = (int) ceil( rf -> len2 );
	rf -> kernellen = rf -> fl2 + rf -> fl2;
	rf -> fraccount = 1000;
	rf -> k = k0;
}

static void rf_free( rfilters* rf )
{
	int i;

	for( i = 0; i <= 1000; i++ )
	{
		free( rf -> filters[ i ]);
	}
}

/* makeFilterNorm, lancir.h:1076-1156. */
static void rf_make( const rfilters* rf, float* op, double fracdelay )
{
	const int fl2 = rf -> fl2;
	sinegen f, fw;
	float* op0 = op;
	double s = 0.0;
	double ut;
	int t = -fl2;
	int iszerox, mt;

	sg_init( &f, rf -> freq, rf -> freq * ( fracdelay - fl2 ));
	sg_init( &fw, rf -> freqa, rf -> freqa * ( fracdelay - fl2 ));

	if( t + fracdelay < -rf -> len2 )
	{
		sg_next( &f );
		sg_next( &fw );
		*op = 0;
		op++;
		t++;
	}

	iszerox = ( fabs( fracdelay - 1.0 ) < 2.3e-13 );
	mt = 0 - iszerox;
	iszerox |= ( fabs( fracdelay ) < 2.3e-13 );

	while( t < mt )
	{
		ut = t + fracdelay;
		*op = (float) ( sg_next( &f ) * sg_next( &fw ) / ( ut * ut ));
		s += *op;
		op++;
		t++;
	}

	if( iszerox )
	{
		*op = (float) ( rf -> freq * rf -> freqa );
		s += *op;
		sg_next( &f );
		sg_next( &fw );
	}
	else
	{
		ut = fracdelay;
		*op = (float) ( sg_next( &f ) * sg_next( &fw ) / ( ut * ut ));
		s += *op;
	}

	mt = fl2 - 2;

	while( t < mt )
	{
		op++;
		t++;
		ut = t + fracdelay;
		*op = (float) ( sg_next( &f ) * sg_next( &fw ) / ( ut * ut ));
		s += *op;
	}

	op++;
	ut = t + 1 + fracdelay;

	if( ut > rf -> len2 )
	{
		*op = 0;
	}
	else
	{
		*op = (float) ( sg_next( &f ) * sg_next( &fw ) / ( ut * ut ));
		s += *op;
	}

	s = 1.0 / s;
	t = (int) ( op - op0 + 1 );

	while( t != 0 )
	{
		*op0 = (float) ( *op0 * s );
		op0++;
		t--;
	}
}

/* getFilter, lancir.h:940-967. */
static const float* rf_get( rfilters* rf, double x )
{
	const int frac = (int) ( x * rf -> fraccount + 0.5 );

	if( rf -> filters[ frac ] == NULL )
	{
		rf -> filters[ frac ] = (float*) calloc( (size_t) rf -> kernellen + 4,
			sizeof( float ));

		rf_make( rf, rf -> filters[ frac ],
			1.0 - (double) frac / rf -> fraccount );
	}

	return( rf -> filters[ frac ]);
}

/* CResizeScanline::update, lancir.h:1290-1351. */
typedef struct
{
	int padl, padr;
	const float** flt;
	int* so;
} rscan;

static void rs_update( rscan* rs, int srclen, int dstlen, double o0,
	rfilters* rf )
{
	const int fl2m1 = rf -> fl2 - 1;
	const double k = rf -> k;
	const int dstlen_m1 = dstlen - 1;
	const double oe = o0 + k * dstlen_m1;
	const int ie = (int) floor( oe );
	int so, i;

	rs -> padl = fl2m1 - (int) floor( o0 );

	if( rs -> padl < 0 )
	{
		rs -> padl = 0;
	}

	rs -> padr = ie + rf -> fl2 + 1 - srclen;

	if( rs -> padr < 0 )
	{
		rs -> padr = 0;
	}

	rs -> flt = (const float**) calloc( (size_t) dstlen, sizeof( float* ));
	rs -> so = (int*) calloc( (size_t) dstlen, sizeof( int ));
	so = rs -> padl - fl2m1;

	for( i = 0; i < dstlen_m1; i++ )
	{
		const double ox = o0 + k * i;
		const int ix = (int) floor( ox );
		rs -> flt[ i ] = rf_get( rf, ox - ix );
		rs -> so[ i ] = so + ix;
	}

	rs -> flt[ dstlen_m1 ] = rf_get( rf, oe - ie );
	rs -> so[ dstlen_m1 ] = so + ie;
}

static void rs_free( rscan* rs )
{
	free( (void*) rs -> flt );
	free( rs -> so );
}

/* resize1..4 (lancir.h:2102-2544), SSE2 lane order, one output pixel.
 * `ip` points at the first source pixel (el floats per pixel). */
static void dot_pixel( const float* flt, const float* ip, int kl, int el,
	float* op )
{
	int g, t;

	if( el == 4 )
	{
		/* sumA: even taps, sumB: odd taps; first term is the bare product */
		float a[ 4 ], b[ 4 ];
		int c;

		for( c = 0; c < 4; c++ )
		{
			a[ c ] = flt[ 0 ] * ip[ c ];
			b[ c ] = flt[ 1 ] * ip[ 4 + c ];
		}

		for( t = 2; t < kl; t += 2 )
		{
			for( c = 0; c < 4; c++ )
			{
				a[ c ] = a[ c ] + flt[ t ] * ip[ t * 4 + c ];
				b[ c ] = b[ c ] + flt[ t + 1 ] * ip[ ( t + 1 ) * 4 + c ];
			}
		}

		for( c = 0; c < 4; c++ )
		{
			op[ c ] = a[ c ] + b[ c ];
		}
	}
	else
	if( el == 1 )
	{
		/* lane L accumulates taps 4g+L; then (s0+s2)+(s1+s3)
		 * (lancvec_store32_hadd); kl%4==2 adds the two extra taps to the two
		 * partial sums first (lancir.h:2186-2194) */
		const int ci = kl >> 2;
		float s[ 4 ];

		for( t = 0; t < 4; t++ )
		{
			s[ t ] = flt[ t ] * ip[ t ];
		}

		for( g = 1; g < ci; g++ )
		{
			for( t = 0; t < 4; t++ )
			{
				s[ t ] = s[ t ] + flt[ g * 4 + t ] * ip[ g * 4 + t ];
			}
		}

		if(( kl & 3 ) == 0 )
		{
			op[ 0 ] = ( s[ 0 ] + s[ 2 ]) + ( s[ 1 ] + s[ 3 ]);
		}
		else
		{
			const float e0 = flt[ kl - 2 ] * ip[ kl - 2 ];
			const float e1 = flt[ kl - 1 ] * ip[ kl - 1 ];
			op[ 0 ] = (( s[ 0 ] + s[ 2 ]) + e0 ) + (( s[ 1 ] + s[ 3 ]) + e1 );
		}
	}
	else
	if( el == 2 )
	{
		/* per channel: sumA lanes {0,2} = taps 4g, 4g+1; sumB lanes = taps
		 * 4g+2, 4g+3; sumA += sumB; (+ taps 4ci, 4ci+1 if kl%4==2); then
		 * lane0 + lane2 (lancir.h:2266-2291) */
		const int ci = kl >> 2;
		int c;

		for( c = 0; c < 2; c++ )
		{
			float a0 = flt[ 0 ] * ip[ 0 * 2 + c ];
			float a2 = flt[ 1 ] * ip[ 1 * 2 + c ];
			float b0 = flt[ 2 ] * ip[ 2 * 2 + c ];
			float b2 = flt[ 3 ] * ip[ 3 * 2 + c ];

			for( g = 1; g < ci; g++ )
			{
				const int q = g * 4;
				a0 = a0 + flt[ q ] * ip[ q * 2 + c ];
				a2 = a2 + flt[ q + 1 ] * ip[ ( q + 1 ) * 2 + c ];
				b0 = b0 + flt[ q + 2 ] * ip[ ( q + 2 ) * 2 + c ];
				b2 = b2 + flt[ q + 3 ] * ip[ ( q + 3 ) * 2 + c ];
			}

			a0 = a0 + b0;
			a2 = a2 + b2;

			if(( kl & 3 ) == 2 )
			{
				const int q = ci * 4;
				a0 = a0 + flt[ q ] * ip[ q * 2 + c ];
				a2 = a2 + flt[ q + 1 ] * ip[ ( q + 1 ) * 2 + c ];
			}

			op[ c ] = a0 + a2;
		}
	}
	else
	{
		/* el == 3: products e[i], i = 3*tap + channel, summed in three
		 * 4-lane vectors per group of 4 taps (lancir.h:2372-2424) */
		const int ci = kl >> 2;
		const int cir = kl & 3;
		float r[ 12 ];
		float o0, o1, o2;
		int i;

		for( i = 0; i < 12; i++ )
		{
			r[ i ] = flt[ i / 3 ] * ip[ i ];
		}

		for( g = 1; g < ci; g++ )
		{
			for( i = 0; i < 12; i++ )
			{
				const int q = g * 12 + i;
				r[ i ] = r[ i ] + flt[ q / 3 ] * ip[ q ];
			}
		}

		if( cir == 2 )
		{
			for( i = 0; i < 4; i++ )
			{
				const int q = ci * 12 + i;
				r[ i ] = r[ i ] + flt[ q / 3 ] * ip[ q ];
			}
		}

		o0 = r[ 0 ] + r[ 3 ];
		o1 = r[ 1 ] + r[ 4 ];
		o2 = r[ 2 ] + r[ 5 ];
		o0 += r[ 6 ] + r[ 9 ];
		o1 += r[ 7 ] + r[ 10 ];
		o2 += r[ 8 ] + r[ 11 ];

		if( cir == 2 )
		{
			const int q = ci * 12 + 4;
			o1 += flt[ q / 3 ] * ip[ q ];
			o2 += flt[ ( q + 1 ) / 3 ] * ip[ q + 1 ];
		}

		op[ 0 ] = o0;
		op[ 1 ] = o1;
		op[ 2 ] = o2;
	}
}

/* roundclamp, lancir.h:1746-1750. */
static int roundclamp( float v, float clampv )
{
	return( (int) (( v > clampv ? clampv : ( v < 0.0f ? 0.0f : v )) + 0.5f ));
}

/* outputScanline, lancir.h:1772-2056 (SSE2 build). */
static void output_scanline( const float* ip, void* op, size_t oo, int l,
	int out_type, int isunitymul, float clampv, float outmul )
{
	int i;

	if( out_type == ORACLE_F32 )
	{
		float* o = (float*) op + oo;

		for( i = 0; i < l; i++ )
		{
			o[ i ] = ( isunitymul ? ip[ i ] : ip[ i ] * outmul );
		}

		return;
	}

	if( out_type == ORACLE_F64 )
	{
		/* (T) ip[ i ] or (T) ( ip[ i ] * OutMul ): the product is a float
		 * product, widened afterwards (lancir.h:1786-1856) */
		double* o = (double*) op + oo;

		for( i = 0; i < l; i++ )
		{
			const float v = ( isunitymul ? ip[ i ] : ip[ i ] * outmul );
			o[ i ] = (double) v;
		}

		return;
	}

	{
		const int l4 = ( l >> 2 ) << 2;

		for( i = 0; i < l; i++ )
		{
			int r;

			if( i < l4 )
			{
				/* max( min( v*om, maxv ), 0 ) then cvtps (nearest-even) */
				float v = ( isunitymul ? ip[ i ] : ip[ i ] * outmul );
				v = ( v < clampv ? v : clampv );
				v = ( v > 0.0f ? v : 0.0f );
				r = (int) lrintf( v );
			}
			else
			{
				r = roundclamp( isunitymul ? ip[ i ] : ip[ i ] * outmul,
					clampv );
			}

			if( out_type == ORACLE_U8 )
			{
				((uint8_t*) op )[ oo + i ] = (uint8_t) r;
			}
			else
			if( out_type == ORACLE_U32 )
			{
				/* sizeof( op[ 0 ]) == 4: the four cvtps results stored whole
				 * (lancir.h:1869-1887), the tail as (T) roundclamp() */
				((uint32_t*) op )[ oo + i ] = (uint32_t) r;
			}
			else
			{
				((uint16_t*) op )[ oo + i ] = (uint16_t) r;
			}
		}
	}
}

static float load_in( const void* p, size_t i, int t )
{
	switch( t )
	{
		case ORACLE_U8: return( (float) ((const uint8_t*) p )[ i ]);
		case ORACLE_U16: return( (float) ((const uint16_t*) p )[ i ]);
		case ORACLE_U32: return( (float) ((const uint32_t*) p )[ i ]);
		case ORACLE_F64: return( (float) ((const double*) p )[ i ]);
		default: return( ((const float*) p )[ i ]);
	}
}

int oracle_lancir_resize( const void* src, int sw, int sh, void* dst, int nw,
	int nh, int ch, int in_type, int out_type, int src_ssize, int new_ssize,
	double pkx, double pky, double pox, double poy, double la )
{
	rfilters rfv, rfh0;
	rfilters* rfh;
	rscan rsv, rsh;
	double ox = pox, oy = poy, kx, ky;
	size_t srcss, newss, fltw;
	float* fltbuf;
	float* spv;
	float* sph;
	int isinfloat, isoutfloat, isunitymul, szin, szout;
	float clampv, outmul;
	int x, y, t, c;

	if( sw < 0 || sh < 0 || nw <= 0 || nh <= 0 || src == NULL ||
		dst == NULL || src == dst || la < 2.0 || in_type > ORACLE_U32 ||
		out_type > ORACLE_U32 || in_type < 0 || out_type < 0 || ch < 1 ||
		ch > 4 )
	{
		return( 0 );
	}

	/* sizeof( Tin ) / sizeof( Tout ): uint8_t, uint16_t, float, double, uint32_t */
	{
		static const int tsz[ 5 ] = { 1, 2, 4, 8, 4 };
		szin = tsz[ in_type ];
		szout = tsz[ out_type ];
	}
	newss = (size_t) ( new_ssize < 1 ? nw * ch : new_ssize );

	if( sw == 0 || sh == 0 )
	{
		for( y = 0; y < nh; y++ )
		{
			memset( (char*) dst + (size_t) y * newss * szout, 0,
				(size_t) nw * ch * szout );
		}

		return( nh );
	}

	srcss = (size_t) ( src_ssize < 1 ? sw * ch : src_ssize );

	/* lancir.h:430-457 */
	if( pkx >= 0.0 )
	{
		kx = ( pkx == 0.0 ? (double) sw / nw : pkx );
		ox += ( kx - 1.0 ) * 0.5;
	}
	else
	{
		kx = -pkx;
	}

	if( pky >= 0.0 )
	{
		ky = ( pky == 0.0 ? (double) sh / nh : pky );
		oy += ( ky - 1.0 ) * 0.5;
	}
	else
	{
		ky = -pky;
	}

	rf_update( &rfv, la, ky );
	rfh = &rfv;

	if( kx != ky )
	{
		rf_update( &rfh0, la, kx );
		rfh = &rfh0;
	}

	rs_update( &rsv, sh, nh, oy, &rfv );
	rs_update( &rsh, sw, nw, ox, rfh );

	/* lancir.h:526-533 */
	isinfloat = ( in_type == ORACLE_F32 || in_type == ORACLE_F64 );
	isoutfloat = ( out_type == ORACLE_F32 || out_type == ORACLE_F64 );
	isunitymul = ( isinfloat && isoutfloat ) ||
		( isinfloat == isoutfloat && szin == szout );

	clampv = ( szout == 1 ? 255.0f : 65535.0f );
	outmul = ( isoutfloat ? 1.0f : clampv ) /
		( isinfloat ? 1.0f : ( szin == 1 ? 255.0f : 65535.0f ));

	/* Vertical pass: every source column -> NewHeight values, stored at
	 * column offset padl in rows of FltWidthE floats (lancir.h:601-646). */
	fltw = (size_t) ( rsh.padl + sw + rsh.padr ) * ch;
	fltbuf = (float*) calloc( fltw * nh + 16, sizeof( float ));
	spv = (float*) calloc( (size_t) ( rsv.padl + sh + rsv.padr ) * ch + 16,
		sizeof( float ));

	for( x = 0; x < sw; x++ )
	{
		/* copyScanline*v with full-column edge replication (1406-1594) */
		for( t = 0; t < rsv.padl + sh + rsv.padr; t++ )
		{
			int sy = t - rsv.padl;
			sy = ( sy < 0 ? 0 : ( sy >= sh ? sh - 1 : sy ));

			for( c = 0; c < ch; c++ )
			{
				spv[ t * ch + c ] = load_in( src,
					(size_t) sy * srcss + (size_t) x * ch + c, in_type );
			}
		}

		for( y = 0; y < nh; y++ )
		{
			dot_pixel( rsv.flt[ y ], spv + (size_t) rsv.so[ y ] * ch,
				rfv.kernellen, ch,
				fltbuf + (size_t) y * fltw + (size_t) ( rsh.padl + x ) * ch );
		}
	}

	/* Horizontal pass + output (lancir.h:650-706). */
	sph = (float*) calloc( (size_t) nw * ch + 16, sizeof( float ));

	for( y = 0; y < nh; y++ )
	{
		float* row = fltbuf + (size_t) y * fltw;

		/* padScanline*h, lancir.h:1611-1734 */
		for( t = 0; t < rsh.padl; t++ )
		{
			for( c = 0; c < ch; c++ )
			{
				row[ t * ch + c ] = row[ rsh.padl * ch + c ];
			}
		}

		for( t = 0; t < rsh.padr; t++ )
		{
			for( c = 0; c < ch; c++ )
			{
				row[ ( rsh.padl + sw + t ) * ch + c ] =
					row[ ( rsh.padl + sw - 1 ) * ch + c ];
			}
		}

		for( x = 0; x < nw; x++ )
		{
			dot_pixel( rsh.flt[ x ], row + (size_t) rsh.so[ x ] * ch,
				rfh -> kernellen, ch, sph + (size_t) x * ch );
		}

		output_scanline( sph, dst, (size_t) y * newss, nw * ch, out_type,
			isunitymul, clampv, outmul );
	}

	free( sph );
	free( spv );
	free( fltbuf );
	rs_free( &rsv );
	rs_free( &rsh );
	rf_free( &rfv );

	if( rfh != &rfv )
	{
		rf_free( &rfh0 );
	}

	return( nh );
}
