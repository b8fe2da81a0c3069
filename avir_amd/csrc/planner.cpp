// planner.cpp -- host-side planner of the front-end mirror (see planner.h).
//
// The device executes plans; it never designs filters. Every float table a plan
// carries is produced here, on the host, in double precision with the same
// operation order as the reference so that the tables are bit-identical:
// windowed-sinc low-pass design (avir.h:1490-1589), the 65-band correction
// equaliser (avir.h:1116-1480, 5384-5506), the polyphase fractional-delay bank
// with optional external-filter convolution (avir.h:1647-2100), the step list
// and its buffer bookkeeping (avir.h:5616-6127), the build-mode cost model
// (avir.h:6206-6270), and LANCIR's normalised Lanczos phases (lancir.h:882-
// 1351). Compiled with g++ -O2 -ffp-contract=off (no FMA contraction).
//
// DERIVED WORK NOTICE. Because bit-identical tables are the contract, the
// numerical parts of this file follow the reference's algorithms statement by
// statement (same formulas, same operation order): it is a derived work of
// avir.h / lancir.h of AVIR v3.1 (https://github.com/avaneev/avir), which are
// distributed under the MIT License:
//
//   Copyright (c) 2015-2025 Aleksey Vaneev
//
//   Permission is hereby granted, free of charge, to any person obtaining a
//   copy of this software and associated documentation files (the
//   "Software"), to deal in the Software without restriction, including
//   without limitation the rights to use, copy, modify, merge, publish,
//   distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the
//   following conditions:
//
//   The above copyright notice and this permission notice shall be included
//   in all copies or substantial portions of the Software.
//
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS
//   OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF
//   MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN
//   NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM,
//   DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE
//   USE OR OTHER DEALINGS IN THE SOFTWARE.

#include "planner.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>

namespace avirhip {

namespace plc { // (shared by the float and the double planner)

const double kPi = 3.1415926535897932;   // avir.h:101
const double kPiD2 = 1.5707963267948966; // avir.h:104

// Recursive sine oscillator (CSineGen, avir.h:1004-1040).
struct SineOsc
{
	double cur, prev, incr;

	SineOsc( const double si, const double ph )
		: cur( sin( ph ))
		, prev( sin( ph - si ))
		, incr( 2.0 * cos( si ))
	{
	}

	double next()
	{
		const double r = cur;
		cur = incr * r - prev;
		prev = r;
		return( r );
	}
};

// Right half of the Peaked Cosine window (avir.h:1054-1093).
struct PeakedCosine
{
	double alpha, len2i, pos;
	SineOsc osc;

	PeakedCosine( const double aAlpha, const double aLen2 )
		: alpha( aAlpha )
		, len2i( 1.0 / aLen2 )
		, pos( 0.0 )
		, osc( kPiD2 / aLen2, kPi * 0.5 )
	{
	}

	double next()
	{
		const double h = pow( pos * len2i, alpha );
		pos += 1.0;
		return( osc.next() * ( 1.0 - h ));
	}
};

// Symmetric-odd windowed-sinc low-pass (CDSPPeakedCosineLPF, avir.h:1490).
struct LowPass
{
	int half;  // fl2
	int len;   // FilterLen
	double len2, freq, alpha;

	LowPass( const double aLen2, const double aFreq, const double aAlpha )
		: half( (int) ceil( aLen2 ) - 1 )
		, len( half + half + 1 )
		, len2( aLen2 )
		, freq( aFreq )
		, alpha( aAlpha )
	{
	}

	// generateLPF<double>, avir.h:1528-1582.
	void generate( double* op, const double dcgain ) const
	{
		PeakedCosine wf( alpha, len2 );
		SineOsc f2( freq, 0.0 );
		op += half;
		double* lo = op;
		f2.next();
		*op = freq * wf.next();

		if( dcgain > 0.0 )
		{
			double s = *op;

			for( int t = 1; t <= half; t++ )
			{
				const double v = f2.next() * wf.next() / t;
				op++;
				lo--;
				*op = v;
				*lo = v;
				s += v + v;
			}

			s = dcgain / s;

			for( int t = 0; t < len; t++ )
			{
				lo[ t ] = lo[ t ] * s;
			}
		}
		else
		{
			for( int t = 1; t <= half; t++ )
			{
				const double v = f2.next() * wf.next() / t;
				op++;
				lo--;
				*op = v;
				*lo = v;
			}
		}
	}
};

// A designed low-pass kept for embedding into the bank (CFltBuffer,
// avir.h:1600-1629): compared by its parameters only.
struct FirDesign
{
	double len2, freq, alpha, dcgain;
	std::vector< double > taps;

	FirDesign() : len2( 0.0 ), freq( 0.0 ), alpha( 0.0 ), dcgain( 0.0 ) { }

	bool sameParams( const FirDesign& b ) const
	{
		return( len2 == b.len2 && freq == b.freq && alpha == b.alpha &&
			dcgain == b.dcgain );
	}
};

// normalizeFIRFilter<double>, avir.h:516-541.
void normalizeDC( double* p, const int l, const double dcgain )
{
	double s = 0.0;

	for( int i = 0; i < l; i++ )
	{
		s += p[ i ];
	}

	s = dcgain / s;

	for( int i = 0; i < l; i++ )
	{
		p[ i ] = p[ i ] * s;
	}
}

// Paragraphic FIR equaliser used for the correction filter (CDSPFIREQ,
// avir.h:1116-1480), specialised to the way addCorrectionFilter() calls it:
// linear bands, MinFreq = 0.
class Equalizer
{
public:
	int z, zi, z2, bands;
	std::vector< double > center, k1, k2;
	bool lastVirt;

	void init( const double rate, const double flen, const int nbands,
		const double maxfreq, const double alpha )
	{
		bands = nbands;
		z = (int) ceil( flen * 0.5 );
		zi = z + ( z & 1 );
		z2 = z * 2;
		center.assign( bands, 0.0 );

		std::vector< double > osc( z2 );

		for( int i = 0; i < z; i++ )
		{
			osc[ 2 * i ] = 0.0;
			osc[ 2 * i + 1 ] = 1.0;
		}

		std::vector< double > win( z );
		PeakedCosine wf( alpha, flen * 0.5 );

		for( int i = 1; i <= z; i++ )
		{
			win[ z - i ] = wf.next();
		}

		k1.assign( (size_t) zi * bands, 0.0 );
		k2.assign( (size_t) zi * bands, 0.0 );

		const double m = 1.0;
		const double mo = ( maxfreq - 0.0 ) / ( bands - 1 );
		double f = 0.0;
		double x1 = 0.0;
		double x2;
		center[ 0 ] = 0.0;
		f = f * m + mo;
		double* b1 = k1.data();
		double* b2 = k2.data();

		for( int i = 1; i < bands; i++ )
		{
			x2 = f * 2.0 / rate;
			center[ i ] = x2;
			bandKernel( x1, x2, b1, b2, osc.data(), win.data() );
			b1 += zi;
			b2 += zi;
			x1 = x2;
			f = f * m + mo;
		}

		if( x1 < 1.0 )
		{
			lastVirt = true;
			bandKernel( x1, 1.0, b1, b2, osc.data(), win.data() );
		}
		else
		{
			lastVirt = false;
		}
	}

	int length() const { return( z2 - 1 ); }
	int latency() const { return( z - 1 ); }

	// buildFilter, avir.h:1247-1304 (no first virtual band).
	void build( const double* gains, double* flt ) const
	{
		const double* b1 = k1.data();
		const double* b2 = k2.data();
		double x1 = 0.0;
		double y1 = gains[ 0 ];
		double x2 = center[ 1 ];
		double y2 = gains[ 1 ];

		mix( flt, b1, b2, y1 - y2, x1 * y2 - x2 * y1, false );
		b1 += zi;
		b2 += zi;
		x1 = x2;
		y1 = y2;

		for( int i = 2; i < bands; i++ )
		{
			x2 = center[ i ];
			y2 = gains[ i ];
			mix( flt, b1, b2, y1 - y2, x1 * y2 - x2 * y1, true );
			b1 += zi;
			b2 += zi;
			x1 = x2;
			y1 = y2;
		}

		if( lastVirt )
		{
			mix( flt, b1, b2, y1 - y2, x1 * y2 - y1, true );
		}

		for( int i = 0; i < z - 1; i++ )
		{
			flt[ z + i ] = flt[ z - 2 - i ];
		}
	}

	static int calcLength( const double flen, int& lat )
	{
		const int l = (int) ceil( flen * 0.5 );
		lat = l - 1;
		return( l * 2 - 1 );
	}

private:
	// fillBandKernel, avir.h:1402-1437.
	void bandKernel( const double x1, const double x2, double* b1, double* b2,
		double* osc, const double* win ) const
	{
		const double s2_incr = kPi * x2;
		const double s2_coeff = 2.0 * cos( s2_incr );
		double s2v = sin( s2_incr * ( -z + 1 ));
		double c2v = sin( s2_incr * ( -z + 1 ) + kPi * 0.5 );
		osc[ 0 ] = sin( s2_incr * -z );
		osc[ 1 ] = sin( s2_incr * -z + kPi * 0.5 );

		for( int ks = 1; ks < z; ks++ )
		{
			const int ks2 = ks * 2;
			const double s1v = osc[ ks2 ];
			const double c1v = osc[ ks2 + 1 ];
			osc[ ks2 ] = s2v;
			osc[ ks2 + 1 ] = c2v;

			const double x = kPi * ( ks - z );
			const double v0 = win[ ks - 1 ] / (( x1 - x2 ) * x );

			b1[ ks - 1 ] = ( x2 * s2v - x1 * s1v + ( c2v - c1v ) / x ) * v0;
			b2[ ks - 1 ] = ( s2v - s1v ) * v0;

			s2v = s2_coeff * s2v - osc[ ks2 - 2 ];
			c2v = s2_coeff * c2v - osc[ ks2 - 1 ];
		}

		b1[ z - 1 ] = ( x2 * x2 - x1 * x1 ) / ( x1 - x2 ) * 0.5;
		b2[ z - 1 ] = -1.0;
	}

	void mix( double* out, const double* b1, const double* b2,
		const double c, const double d, const bool add ) const
	{
		if( add )
		{
			for( int ks = 0; ks < z; ks++ )
			{
				out[ ks ] += c * b1[ ks ] + d * b2[ ks ];
			}
		}
		else
		{
			for( int ks = 0; ks < z; ks++ )
			{
				out[ ks ] = c * b1[ ks ] + d * b2[ ks ];
			}
		}
	}
};

// the typed halves of a description: float tables in the fields the float
// pipeline reads, double tables in their ...64 twins
inline std::vector< float >& desc_pool( DescStore& S, float* )
{
	S.f.emplace_back( new std::vector< float >() );
	return( *S.f.back() );
}

inline std::vector< double >& desc_pool( DescStore& S, double* )
{
	S.fd.emplace_back( new std::vector< double >() );
	return( *S.fd.back() );
}

inline void desc_set( const float*& f32, const double*&, const float* p )
{
	f32 = p;
}

inline void desc_set( const float*&, const double*& f64, const double* p )
{
	f64 = p;
}

inline void desc_set_x( avirhip_rpos& r, const float x ) { r.x = x; }
inline void desc_set_x( avirhip_rpos& r, const double x ) { r.x64 = x; }

} // namespace plc

using namespace plc;

} // namespace avirhip

#define PL_NS pl32
#define PL_FT 0
#define FT float
#include "planner_avir.inl"
#undef PL_NS
#undef PL_FT
#undef FT
#define PL_NS pl64
#define PL_FT 1
#define FT double
#include "planner_avir.inl"
#undef PL_NS
#undef PL_FT
#undef FT

namespace avirhip {

AvirPlanner::AvirPlanner( int res_bit_depth, int src_bit_depth,
	const avirhip_params& params )
{
	// (built aside: a throw half-way must not leave the object with a
	// half-built bank)
	std::unique_ptr< pl32::AvirPlannerImpl > q( new pl32::AvirPlannerImpl() );
	impl = nullptr;
	impl64 = nullptr;
	q -> P = params;
	q -> resBits = res_bit_depth;
	q -> srcBits = ( src_bit_depth == 0 ? res_bit_depth : src_bit_depth );

	// FixedFilterBank, avir.h:4637-4638.
	q -> setupBank( q -> fixedBank, 1.0, false, FirDesign() );
	q -> fixedBank.buildAll();
	impl = q.release();
}

AvirPlanner::~AvirPlanner()
{
	delete impl;
	delete impl64;
}

DescStore* AvirPlanner::build( int sw, int sh, int sstride, int nw, int nh,
	int ch, double k, const avirhip_vars& vars, int in_type, int out_type,
	int fppack, bool f64 )
{
	if( f64 )
	{
		if( impl64 == nullptr )
		{
			std::unique_ptr< pl64::AvirPlannerImpl > q(
				new pl64::AvirPlannerImpl() );
			q -> P = impl -> P;
			q -> resBits = impl -> resBits;
			q -> srcBits = impl -> srcBits;
			q -> setupBank( q -> fixedBank, 1.0, false, FirDesign() );
			q -> fixedBank.buildAll();
			impl64 = q.release();
		}

		return( pl64::build_impl( *impl64, sw, sh, sstride, nw, nh, ch, k, vars,
			in_type, out_type, fppack ));
	}

	return( pl32::build_impl( *impl, sw, sh, sstride, nw, nh, ch, k, vars,
		in_type, out_type, fppack ));
}

// ---------------------------------------------------------------------
// LANCIR planning (lancir.h:430-483, 882-1351).
// ---------------------------------------------------------------------

namespace {

struct LancFilters // CResizeFilters, lancir.h:840-1219 (no SIMD replication)
{
	double freq, freqA, len2, k;
	int fl2, kernelLen, fracCount;
	std::map< int, std::vector< float > > phases;

	void setup( const double la, const double k0 )
	{
		const double norm = ( k0 <= 1.0 ? 1.0 : 1.0 / k0 );
		freq = 3.1415926535897932 * norm;
		freqA = freq / la;
		len2 = la / norm;
		fl2 = (int) ceil( len2 );
		kernelLen = fl2 + fl2;
		fracCount = 1000;
		k = k0;
		phases.clear();
	}

	int phaseIndex( const double x )
	{
		const int frac = (int) ( x * fracCount + 0.5 );

		if( phases.find( frac ) == phases.end() )
		{
			std::vector< float >& f = phases[ frac ];
			f.assign( kernelLen, 0.0f );
			make( f.data(), 1.0 - (double) frac / fracCount );
		}

		return( frac );
	}

	// makeFilterNorm, lancir.h:1076-1156.
	void make( float* op, const double fd ) const
	{
		SineOsc f( freq, freq * ( fd - fl2 ));
		SineOsc fw( freqA, freqA * ( fd - fl2 ));
		float* op0 = op;
		double s = 0.0;
		double ut;
		int t = -fl2;

		if( t + fd < -len2 )
		{
			f.next();
			fw.next();
			*op = 0;
			op++;
			t++;
		}

		int zeroX = ( fabs( fd - 1.0 ) < 2.3e-13 );
		int mt = 0 - zeroX;
		zeroX |= ( fabs( fd ) < 2.3e-13 );

		while( t < mt )
		{
			ut = t + fd;
			*op = (float) ( f.next() * fw.next() / ( ut * ut ));
			s += *op;
			op++;
			t++;
		}

		if( zeroX )
		{
			*op = (float) ( freq * freqA );
			s += *op;
			f.next();
			fw.next();
		}
		else
		{
			ut = fd;
			*op = (float) ( f.next() * fw.next() / ( ut * ut ));
			s += *op;
		}

		mt = fl2 - 2;

		while( t < mt )
		{
			op++;
			t++;
			ut = t + fd;
			*op = (float) ( f.next() * fw.next() / ( ut * ut ));
			s += *op;
		}

		op++;
		ut = t + 1 + fd;

		if( ut > len2 )
		{
			*op = 0;
		}
		else
		{
			*op = (float) ( f.next() * fw.next() / ( ut * ut ));
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
};

// CResizeScanline::update, lancir.h:1290-1351.
void lancAxis( DescStore& S, LancFilters& rf, const int srcLen,
	const int dstLen, const double o0, avirhip_lancir_axis& ax )
{
	const int fl2m1 = rf.fl2 - 1;
	int padl = fl2m1 - (int) floor( o0 );

	if( padl < 0 )
	{
		padl = 0;
	}

	const double k = rf.k;
	const int dm1 = dstLen - 1;
	const double oe = o0 + k * dm1;
	const int ie = (int) floor( oe );
	int padr = ie + rf.fl2 + 1 - srcLen;

	if( padr < 0 )
	{
		padr = 0;
	}

	const int so = padl - fl2m1;
	std::vector< int > frac( dstLen ), pos( dstLen );

	for( int i = 0; i < dm1; i++ )
	{
		const double ox = o0 + k * i;
		const int ix = (int) floor( ox );
		frac[ i ] = rf.phaseIndex( ox - ix );
		pos[ i ] = so + ix;
	}

	frac[ dm1 ] = rf.phaseIndex( oe - ie );
	pos[ dm1 ] = so + ie;

	std::map< int, int > used;

	for( int i = 0; i < dstLen; i++ )
	{
		used[ frac[ i ]] = 0;
	}

	S.i.emplace_back( new std::vector< int32_t >() );
	std::vector< int32_t >& ff = *S.i.back();
	S.f.emplace_back( new std::vector< float >() );
	std::vector< float >& ft = *S.f.back();
	int ci = 0;

	for( auto& u : used )
	{
		u.second = ci++;
		ff.push_back( u.first );
		const std::vector< float >& t = rf.phases[ u.first ];
		ft.insert( ft.end(), t.begin(), t.end() );
	}

	S.lp.emplace_back( new std::vector< avirhip_lancir_pos >( dstLen ));
	std::vector< avirhip_lancir_pos >& pp = *S.lp.back();

	for( int i = 0; i < dstLen; i++ )
	{
		pp[ i ].flt_index = used[ frac[ i ]];
		pp[ i ].so = pos[ i ];
	}

	ax.kernel_len = rf.kernelLen;
	ax.padl = padl; ax.padr = padr;
	ax.src_len = srcLen; ax.dst_len = dstLen;
	ax.n_filters = ci;
	ax.filter_frac = ff.data();
	ax.filters = ft.data();
	ax.pos = pp.data();
}

} // namespace

DescStore* lancir_build( int sw, int sh, int nw, int nh, int ch,
	const avirhip_lancir_params& P, int in_type, int out_type )
{
	if( sw < 1 || sh < 1 || nw < 1 || nh < 1 || ch < 1 || ch > 4 ||
		P.la < 2.0 || in_type < 0 || in_type > AVIRHIP_U32 || out_type < 0 ||
		out_type > AVIRHIP_U32 )
	{
		set_error( "lancir planner: bad arguments" );
		return( nullptr );
	}

	double ox = P.ox, oy = P.oy, kx, ky;

	if( P.kx >= 0.0 )
	{
		kx = ( P.kx == 0.0 ? (double) sw / nw : P.kx );
		ox += ( kx - 1.0 ) * 0.5;
	}
	else
	{
		kx = -P.kx;
	}

	if( P.ky >= 0.0 )
	{
		ky = ( P.ky == 0.0 ? (double) sh / nh : P.ky );
		oy += ( ky - 1.0 ) * 0.5;
	}
	else
	{
		ky = -P.ky;
	}

	std::unique_ptr< DescStore > S( new DescStore() );
	memset( &S -> d, 0, sizeof( S -> d ));
	memset( &S -> ld, 0, sizeof( S -> ld ));
	avirhip_lancir_desc& d = S -> ld;

	LancFilters rfv, rfh;
	rfv.setup( P.la, ky );
	lancAxis( *S, rfv, sh, nh, oy, d.v );

	if( kx == ky )
	{
		lancAxis( *S, rfv, sw, nw, ox, d.h );
	}
	else
	{
		rfh.setup( P.la, kx );
		lancAxis( *S, rfh, sw, nw, ox, d.h );
	}

	d.src_w = sw; d.src_h = sh;
	d.src_stride_elems = ( P.SrcSSize < 1 ? sw * ch : P.SrcSSize );
	d.new_w = nw; d.new_h = nh;
	d.new_stride_elems = ( P.NewSSize < 1 ? nw * ch : P.NewSSize );
	d.channels = ch;
	d.in_type = in_type; d.out_type = out_type;

	// Output constants, lancir.h:526-533.
	// Tin / Tout: uint8_t, uint16_t, float, double, and uint32_t "treated as
	// uint16_t" (lancir.h:373-381) -- what the reference derives from the types
	// is IsInFloat / IsOutFloat and sizeof() (a uint32_t is a non-float of size
	// 4: range 65535, but never "the same size" as a uint16_t)
	const bool inFloat = ( in_type == AVIRHIP_F32 || in_type == AVIRHIP_F64 );
	const bool outFloat = ( out_type == AVIRHIP_F32 || out_type == AVIRHIP_F64 );
	static const int tsz[ 5 ] = { 1, 2, 4, 8, 4 };
	const int szin = tsz[ in_type ];
	const int szout = tsz[ out_type ];
	d.is_unity_mul = (( inFloat && outFloat ) ||
		( inFloat == outFloat && szin == szout )) ? 1 : 0;
	d.clamp = ( szout == 1 ? 255.0f : 65535.0f );
	d.out_mul = ( outFloat ? 1.0f : d.clamp ) /
		( inFloat ? 1.0f : ( szin == 1 ? 255.0f : 65535.0f ));

	return( S.release() );
}

} // namespace avirhip
