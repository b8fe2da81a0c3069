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

namespace {

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

// calcFIRFilterResponse<float>, avir.h:460-503 (fltlat = 0).
void firResponse( const float* flt, int n, const double th, double& re0,
	double& im0 )
{
	const double sincr = 2.0 * cos( th );
	double c1 = 1.0, s1 = 0.0;
	double c2 = cos( -th );
	double s2 = sin( -th );
	double re = 0.0, im = 0.0;

	while( n > 0 )
	{
		re += c1 * (double) flt[ 0 ];
		im += s1 * (double) flt[ 0 ];
		flt++;
		n--;

		double t = c1;
		c1 = sincr * c1 - c2;
		c2 = t;
		t = s1;
		s1 = sincr * s1 - s2;
		s2 = t;
	}

	re0 = re;
	im0 = im;
}

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

// Polyphase fractional-delay bank with order-0/1 interpolation and optional
// embedded external low-pass (CDSPFracFilterBankLin<float>, avir.h:1647-2100).
class FracBank
{
public:
	double wfLen2, wfFreq, wfAlpha;
	int fracCount, order, srcLen, fltLen, fltSize;
	bool initRequired, srcBuilt;
	FirDesign ext;
	std::vector< double > src; // (fracCount+1) blocks of srcLen
	std::vector< float > table; // (fracCount+1) * fltSize
	std::vector< char > flags;  // fracCount+1

	FracBank()
		: wfLen2( 0.0 ), wfFreq( 0.0 ), wfAlpha( 0.0 ), fracCount( 0 )
		, order( -1 ), srcLen( 0 ), fltLen( 0 ), fltSize( 0 )
		, initRequired( false ), srcBuilt( false )
	{
	}

	// copyInitParams, avir.h:1668-1691.
	void inheritParams( const FracBank& s )
	{
		wfLen2 = s.wfLen2; wfFreq = s.wfFreq; wfAlpha = s.wfAlpha;
		fracCount = s.fracCount; order = s.order;
		srcLen = s.srcLen; fltLen = s.fltLen; fltSize = s.fltSize;
		srcBuilt = false;
		ext = s.ext;
		flags.resize( s.flags.size() );

		for( size_t i = 0; i < flags.size(); i++ )
		{
			flags[ i ] = (char) ( s.flags[ i ] << 2 );
		}
	}

	// operator ==, avir.h:1702-1707.
	bool sameAs( const FracBank& s ) const
	{
		return( order == s.order && wfLen2 == s.wfLen2 &&
			wfFreq == s.wfFreq && wfAlpha == s.wfAlpha &&
			fracCount == s.fracCount && ext.sameParams( s.ext ));
	}

	// init, avir.h:1732-1772 (FltLenAlign = 1).
	void configure( const int reqFrac, const int reqOrder,
		const double baseLen, const double cutoff, const double alpha,
		const FirDesign& aExt )
	{
		const double nLen2 = 0.5 * baseLen * reqFrac;
		const double nFreq = kPi * cutoff / reqFrac;

		if( reqOrder == order && nLen2 == wfLen2 && nFreq == wfFreq &&
			alpha == wfAlpha && reqFrac == fracCount &&
			aExt.sameParams( ext ))
		{
			initRequired = false;
			return;
		}

		wfLen2 = nLen2; wfFreq = nFreq; wfAlpha = alpha;
		fracCount = reqFrac; order = reqOrder;
		ext = aExt;

		const LowPass p( wfLen2, wfFreq, wfAlpha );
		srcLen = ( p.half / reqFrac + 1 ) * 2;
		fltLen = srcLen;

		if( !ext.taps.empty() )
		{
			fltLen += (int) ext.taps.size() - 1;
		}

		fltSize = fltLen * ( reqOrder + 1 );
		srcBuilt = false;
		initRequired = true;
	}

	// getFilter, avir.h:1814-1846.
	const float* phase( const int i )
	{
		if( !srcBuilt )
		{
			buildSource();
		}

		float* const res = &table[ (size_t) i * fltSize ];

		if(( flags[ i ] & 2 ) == 0 )
		{
			makePhase( i );
			flags[ i ] |= 2;

			if( order > 0 )
			{
				makePhase( i + 1 );
				const float* const nxt = res + fltSize;
				float* const dif = res + fltLen;

				for( int j = 0; j < fltLen; j++ )
				{
					dif[ j ] = nxt[ j ] - res[ j ];
				}
			}
		}

		return( res );
	}

	void buildAll()
	{
		for( int i = 0; i < fracCount; i++ )
		{
			phase( i );
		}
	}

	// calcInitComplexity, avir.h:1895-1929.
	int initCost( const std::vector< char >& useMap ) const
	{
		const int fltInit = 65;
		const int fltUse = fltLen * order + srcLen * (int) ext.taps.size();
		const int ucb[ 2 ] = { 0, fltUse };
		int ic;

		if( initRequired )
		{
			ic = fracCount * srcLen * fltInit;

			for( int i = 0; i < fracCount; i++ )
			{
				ic += ucb[ (size_t) useMap[ i ]];
			}
		}
		else
		{
			ic = 0;

			for( int i = 0; i < fracCount; i++ )
			{
				if( useMap[ i ] != 0 )
				{
					ic += ucb[ flags[ i ] == 0 ? 1 : 0 ];
				}
			}
		}

		return( ic );
	}

private:
	// buildSrcTable, avir.h:1970-2009.
	void buildSource()
	{
		srcBuilt = true;
		initRequired = false;

		const LowPass p( wfLen2, wfFreq, wfAlpha );
		const int bufLen = srcLen * fracCount + 1;
		const int center = srcLen * fracCount / 2;
		std::vector< double > buf( bufLen, 0.0 );
		p.generate( &buf[ center - p.half ], 0.0 );

		src.assign( (size_t) ( fracCount + 1 ) * srcLen, 0.0 );
		flags.assign( fracCount + 1, 0 );
		double* op = src.data();

		for( int i = fracCount; i >= 0; i-- )
		{
			const double* ip = buf.data() + i;

			for( int j = 0; j < srcLen; j++ )
			{
				op[ j ] = ip[ 0 ];
				ip += fracCount;
			}

			normalizeDC( op, srcLen, 1.0 );
			op += srcLen;
		}

		table.assign( (size_t) ( fracCount + 1 ) * fltSize, 0.0f );
	}

	// createFilter, avir.h:2021-2099.
	void makePhase( const int n )
	{
		if( flags[ n ] != 0 )
		{
			return;
		}

		flags[ n ] |= 1;
		const int extLen = (int) ext.taps.size();
		const int resLat = extLen / 2 + srcLen / 2;
		int resLen = srcLen;

		if( extLen > 0 )
		{
			resLen += extLen - 1;
		}

		const int resOffs = fltLen / 2 - resLat;
		float* op = &table[ (size_t) n * fltSize ];

		for( int i = 0; i < resOffs; i++ )
		{
			op[ i ] = 0;
		}

		for( int i = resOffs + resLen; i < fltLen; i++ )
		{
			op[ i ] = 0;
		}

		op += resOffs;
		const double* const sf = &src[ (size_t) n * srcLen ];

		if( extLen == 0 )
		{
			for( int i = 0; i < resLen; i++ )
			{
				op[ i ] = (float) sf[ i ];
			}

			return;
		}

		const double* const ef = ext.taps.data();

		for( int j = 0; j < resLen; j++ )
		{
			int k = 0;
			int l = j - extLen + 1;
			int r = l + extLen;

			if( l < 0 )
			{
				k -= l;
				l = 0;
			}

			if( r > srcLen )
			{
				r = srcLen;
			}

			const double* const eb = ef + k;
			const double* const sb = sf + l;
			double s = 0.0;
			l = r - l;

			for( int i = 0; i < l; i++ )
			{
				s += eb[ i ] * sb[ i ];
			}

			op[ j ] = (float) s;
		}
	}
};

struct RPos
{
	int srcPosInt, fti, srcOffs, fl, ftpOff;
	float x;
};

// One filtering step (CImageResizerFilterStep, avir.h:2568-2728). `fltCap`,
// `pdcCap`, `sdcCap` track CBuffer capacities, which exist in modelling mode
// where the contents are not built.
struct PStep
{
	bool up;
	int rf;
	std::vector< float > flt;
	int fltCap;
	FirDesign orig;
	double dcgain;
	int lat;
	int inLen, inPrefix, inSuffix, outLen, outPrefix, outSuffix;
	std::vector< float > pdc, sdc;
	int pdcCap, sdcCap;
	int edge;
	FracBank* bank;
	bool bankDyn;
	std::vector< RPos > rpos;

	PStep()
		: up( false ), rf( 0 ), fltCap( 0 ), dcgain( 0.0 ), lat( 0 )
		, inLen( 0 ), inPrefix( 0 ), inSuffix( 0 ), outLen( 0 )
		, outPrefix( 0 ), outSuffix( 0 ), pdcCap( 0 ), sdcCap( 0 ), edge( 0 )
		, bank( nullptr ), bankDyn( false )
	{
	}
};

typedef std::vector< std::unique_ptr< PStep > > PSteps;

struct AxisVars // the per-axis part of CImageResizerVarsBase
{
	double k, o;
	int resizeStep;
	bool isResize2;
	int elCount;
};

} // namespace

class AvirPlannerImpl
{
public:
	avirhip_params P;
	int srcBits, resBits;
	FracBank fixedBank;

	// initFilterBank, avir.h:5128-5164.
	void setupBank( FracBank& b, const double cutoffMult, const bool hiOrder,
		const FirDesign& ext ) const
	{
		const int ibd = ( resBits > srcBits ? resBits : srcBits );
		const double snr = -6.02 * ( ibd + 3 );
		int useOrder, fracCount;

		if( hiOrder || ibd > 8 )
		{
			useOrder = 1;
			fracCount = (int) ceil( 0.23134052 * exp( -0.058062929 * snr ));
		}
		else
		{
			useOrder = 0;
			fracCount = (int) ceil( 0.33287686 * exp( -0.11334583 * snr ));
		}

		if( fracCount < 2 )
		{
			fracCount = 2;
		}

		b.configure( fracCount, useOrder, P.IntFltLen / cutoffMult,
			P.IntFltCutoff * cutoffMult, P.IntFltAlpha, ext );
	}

	// assignFilterParams, avir.h:5231-5360.
	bool designLowPass( PStep& fs, const bool isUp, const int rf,
		const double cutoff, const double dcgain, const bool keepOrig,
		const bool model ) const
	{
		double alpha, len2, freq;

		if( cutoff == 0.0 )
		{
			const double m = 2.0 / rf;
			alpha = P.HBFltAlpha;
			len2 = 0.5 * P.HBFltLen / m;
			freq = kPi * P.HBFltCutoff * m;
		}
		else
		{
			alpha = P.LPFltAlpha;
			len2 = 0.25 * P.LPFltBaseLen / cutoff;
			freq = kPi * P.LPFltCutoffMult * cutoff;
		}

		if( isUp )
		{
			len2 *= rf;
			freq /= rf;
			fs.dcgain = dcgain * rf;
		}
		else
		{
			fs.dcgain = dcgain;
		}

		fs.orig.len2 = len2;
		fs.orig.freq = freq;
		fs.orig.alpha = alpha;
		fs.orig.dcgain = fs.dcgain;

		const LowPass w( len2, freq, alpha );
		fs.up = isUp;
		fs.rf = rf;
		fs.lat = w.half;

		if( w.len < 1 )
		{
			return( false );
		}

		if( model )
		{
			fs.fltCap = w.len;
			fs.flt.clear();

			if( keepOrig )
			{
				fs.orig.taps.assign( w.len, 0.0 );
			}
		}
		else
		{
			fs.orig.taps.assign( w.len, 0.0 );
			w.generate( fs.orig.taps.data(), fs.dcgain );
			fs.fltCap = w.len;
			fs.flt.resize( w.len );

			for( int i = 0; i < w.len; i++ )
			{
				fs.flt[ i ] = (float) fs.orig.taps[ i ];
			}

			if( !keepOrig )
			{
				fs.orig.taps.clear();
			}
		}

		if( isUp )
		{
			int l = fs.fltCap - fs.lat - rf;

			if( l < 0 )
			{
				return( false );
			}

			fs.pdcCap = l;
			fs.sdcCap = fs.lat;

			if( model )
			{
				return( true );
			}

			// DC "tails" added at the scanline ends, accumulated in float
			// (copyArray/addArray, avir.h:5320-5353).
			fs.pdc.assign( l, 0.0f );
			fs.sdc.assign( fs.lat, 0.0f );
			const float* ip = &fs.flt[ fs.lat + rf ];

			for( int i = 0; i < l; i++ )
			{
				fs.pdc[ i ] = ip[ i ];
			}

			while( true )
			{
				ip += rf;
				l -= rf;

				if( l <= 0 )
				{
					break;
				}

				for( int i = 0; i < l; i++ )
				{
					fs.pdc[ i ] += ip[ i ];
				}
			}

			l = fs.lat;
			float* op = fs.sdc.data();

			for( int i = 0; i < l; i++ )
			{
				op[ i ] = fs.flt[ i ];
			}

			while( true )
			{
				op += rf;
				l -= rf;

				if( l <= 0 )
				{
					break;
				}

				for( int i = 0; i < l; i++ )
				{
					op[ i ] += fs.flt[ i ];
				}
			}
		}
		else
		if( !keepOrig )
		{
			fs.edge = 3; // EdgePixelCountDef, avir.h:2629
		}

		return( true );
	}

	// addCorrectionFilter, avir.h:5384-5506.
	void addCorrection( PSteps& steps, const double bw, const bool pre,
		const bool model ) const
	{
		if( !pre )
		{
			steps.emplace_back( new PStep() );
		}

		PStep& nfs = ( pre ? *steps[ 0 ] : *steps.back() );
		nfs.up = false;
		nfs.rf = 1;
		nfs.dcgain = 1.0;
		nfs.edge = ( pre ? 3 : 0 );

		if( model )
		{
			nfs.fltCap = Equalizer::calcLength( P.CorrFltLen, nfs.lat );
			return;
		}

		const int binCount = 65;
		const int binCount1 = binCount - 1;
		double curbw = 1.0;
		double re, im;
		std::vector< double > bins( binCount, 1.0 );
		const int si = ( pre ? 1 : 0 );
		const int cnt = (int) steps.size();

		for( int i = si; i < cnt - ( si ^ 1 ); i++ )
		{
			PStep& fs = *steps[ i ];

			if( fs.up )
			{
				curbw *= fs.rf;

				if( !fs.orig.taps.empty() )
				{
					continue;
				}
			}

			const float* flt;
			int fltLen;

			if( fs.rf == 0 )
			{
				// phase 0 of the bank; a private bank builds it on demand
				flt = ( fs.bankDyn ? fs.bank -> phase( 0 ) :
					&fs.bank -> table[ 0 ]);
				fltLen = fs.bank -> fltLen;
			}
			else
			{
				flt = fs.flt.data();
				fltLen = fs.fltCap;
			}

			const double thm = kPi * bw / ( curbw * binCount1 );

			for( int j = 0; j < binCount; j++ )
			{
				firResponse( flt, fltLen, j * thm, re, im );
				bins[ j ] *= fs.dcgain / sqrt( re * re + im * im );
			}

			if( !fs.up && fs.rf > 1 )
			{
				curbw /= fs.rf;
			}
		}

		Equalizer eq;
		eq.init( bw * 2.0, P.CorrFltLen, binCount, bw, P.CorrFltAlpha );
		nfs.lat = eq.latency();

		std::vector< double > filter( eq.length(), 0.0 );
		eq.build( bins.data(), filter.data() );
		normalizeDC( filter.data(), (int) filter.size(), 1.0 );

		nfs.fltCap = (int) filter.size();
		nfs.flt.resize( filter.size() );

		for( size_t i = 0; i < filter.size(); i++ )
		{
			nfs.flt[ i ] = (float) filter[ i ];
		}
	}

	// buildFilterSteps, avir.h:5616-5739. Half-band mode (flag 4) is
	// unreachable in the reference (BuildModeCount <= 4, avir.h:4814) but is
	// carried for forced build modes.
	bool buildSteps( PSteps& steps, AxisVars& V, FracBank& bank,
		const double dcgain, const int mode, const bool model )
	{
		steps.clear();
		const bool combo = (( mode & 1 ) != 0 );
		const bool hiOrder = (( mode & 2 ) != 0 );
		const bool halfband = (( mode & 4 ) != 0 );
		const double bw = 1.0 / V.k;
		const int upFactor = ( (int) floor( V.k ) < 2 ? 2 : 1 );
		double intCutoffMult;
		PStep* reuse;
		PStep* extStep;
		bool pre;
		double cutoff, corrbw;

		if( V.k <= 1.0 )
		{
			pre = true;
			cutoff = 1.0;
			corrbw = 1.0;
			steps.emplace_back( new PStep() );
		}
		else
		{
			pre = false;
			cutoff = bw;
			corrbw = bw;
		}

		if( upFactor > 1 )
		{
			steps.emplace_back( new PStep() );
			PStep& fs = *steps.back();

			if( !designLowPass( fs, true, upFactor, cutoff, dcgain, combo,
				model ))
			{
				return( false );
			}

			intCutoffMult = cutoff * 2.0 / upFactor;
			reuse = nullptr;
			extStep = ( combo ? &fs : nullptr );
		}
		else
		{
			int down;

			while( true )
			{
				down = (int) floor( 0.5 / cutoff );
				const bool addHB = ( halfband && down > 1 );

				if( addHB )
				{
					steps.emplace_back( new PStep() );

					if( !designLowPass( *steps.back(), false, down, 0.0, 1.0,
						false, model ))
					{
						return( false );
					}

					cutoff *= down;
				}
				else
				{
					if( down < 1 )
					{
						down = 1;
					}

					break;
				}
			}

			steps.emplace_back( new PStep() );
			PStep& fs = *steps.back();

			if( !designLowPass( fs, false, down, cutoff, dcgain, combo,
				model ))
			{
				return( false );
			}

			intCutoffMult = cutoff / 0.5;

			if( combo )
			{
				reuse = &fs;
				extStep = &fs;
			}
			else
			{
				intCutoffMult *= down;
				reuse = nullptr;
				extStep = nullptr;
			}
		}

		if( reuse == nullptr )
		{
			steps.emplace_back( new PStep() );
		}

		PStep& fs = ( reuse == nullptr ? *steps.back() : *reuse );
		V.resizeStep = (int) steps.size() - 1;
		fs.up = false;
		fs.rf = 0;
		fs.dcgain = ( extStep == nullptr ? 1.0 : extStep -> dcgain );

		setupBank( bank, intCutoffMult, hiOrder,
			( extStep == nullptr ? fs.orig : extStep -> orig ));

		if( bank.sameAs( fixedBank ))
		{
			fs.bank = &fixedBank;
			fs.bankDyn = false;
		}
		else
		{
			fs.bank = &bank;
			fs.bankDyn = true;
		}

		addCorrection( steps, corrbw, pre, model );
		return( true );
	}

	// ---- step geometry, in the executor's own terms -----------------------
	// The kernels read every step's input through an index view (plan.h), so
	// what a step needs to know is the RANGE of input indices its outputs
	// touch: [lo, hi] relative to an input array [0, n). The part of the range
	// below 0 is the step's prefix, the part above n - 1 its suffix -- the
	// quantities the reference calls InPrefix / InSuffix (and materialises as
	// replicated pixels, avir.h:5827-5937; here they only parameterise the
	// views and are exported for plan-parity checks). The coordinate pair
	// (k, o) that places the resize positions walks the same list: an
	// upsampler scales it, a decimating filter divides it and shifts it by
	// its edge pixels.
	struct Span
	{
		int lo, hi;
		int below() const { return( lo < 0 ? -lo : 0 ); }
		int above( const int n ) const { return( hi > n - 1 ? hi - ( n - 1 ) : 0 ); }
	};

	// input indices read by outputs [0, outs) of a symmetric FIR of `cap`
	// taps and latency `lat` that decimates by `rf` and starts `edge`
	// outputs before the array
	static Span firSpan( const int rf, const int lat, const int cap,
		const int edge, const int outs )
	{
		Span r;
		r.lo = -edge * rf - lat;
		r.hi = rf * ( outs - 1 - edge ) + ( cap - lat - 1 );
		return( r );
	}

	// input indices read by a bank filter of length fl around positions
	// o + k*j, j in [0, outs)
	static Span bankSpan( const double o, const double k, const int fl,
		const int outs )
	{
		Span r;
		r.lo = (int) floor( o ) - ( fl / 2 - 1 );
		r.hi = (int) floor( o + ( outs - 1 ) * k ) + fl / 2;
		return( r );
	}

	static void updateBuffers( PSteps& steps, AxisVars& V, int srcLen,
		const int newLen )
	{
		const int cnt = (int) steps.size();
		int zs = -1; // the upsampling step, if any

		for( int i = 0; i < cnt; i++ )
		{
			PStep& fs = *steps[ i ];
			const int n = srcLen;
			fs.inLen = n;

			if( fs.up )
			{
				// n inputs land on slots 0, rf, 2rf ...; the filter (taps
				// scattered from each slot) spills lat slots before and the
				// rest of its length after, and the DC tails must fit too
				zs = i;
				V.k *= fs.rf;
				V.o *= fs.rf;
				fs.inPrefix = fs.inSuffix = 0;
				fs.outLen = n * fs.rf;
				fs.outPrefix = fs.lat;
				const int tail = std::max( fs.fltCap - fs.lat - fs.rf,
					fs.sdcCap - fs.lat );

				fs.outSuffix = std::max( tail, fs.pdcCap - fs.outLen );
			}
			else
			if( fs.rf == 0 )
			{
				const int fl = fs.bank -> fltLen;
				const Span need = bankSpan( V.o, V.k, fl, newLen );
				fs.inPrefix = need.below();
				fs.inSuffix = ( need.hi + 1 > n ? need.hi + 1 - n : 0 );
				fs.outLen = newLen;

				// the position table: double arithmetic in the reference's
				// order (fillRPosBuf, avir.h:5782-5808) -- numerical contract
				const int fracCount = fs.bank -> fracCount;
				fs.rpos.resize( newLen );

				for( int j = 0; j < newLen; j++ )
				{
					const double srcPos = V.o + V.k * j;
					const int spi = (int) floor( srcPos );
					const double x = ( srcPos - spi ) * fracCount;
					const int fti = (int) x;
					fs.rpos[ j ].x = (float) ( x - fti );
					fs.rpos[ j ].fti = fti;
					fs.rpos[ j ].srcPosInt = spi;
				}
			}
			else
			{
				V.k /= fs.rf;
				V.o /= fs.rf;
				V.o += fs.edge;
				fs.outLen = ( n + fs.rf - 1 ) / fs.rf + 2 * fs.edge;
				const Span need = firSpan( fs.rf, fs.lat, fs.fltCap, fs.edge,
					fs.outLen );

				fs.inPrefix = need.below();
				fs.inSuffix = need.hi - ( n - 1 );
			}

			srcLen = fs.outLen;
		}

		V.isResize2 = false;

		if( zs >= 0 )
		{
			// the step after an upsampler reads the upsampler's OUTPUT buffer
			// directly: its overhang becomes extra replicated input samples
			// of the upsampler (whole samples: rounded up to the factor)
			PStep& up = *steps[ zs ];
			PStep& nx = *steps[ zs + 1 ];
			const int pre = ( nx.inPrefix + up.rf - 1 ) / up.rf;
			const int suf = ( nx.inSuffix + up.rf - 1 ) / up.rf;
			up.inPrefix = pre; up.outPrefix += pre * up.rf;
			up.inSuffix = suf; up.outSuffix += suf * up.rf;
			nx.inPrefix = nx.inSuffix = 0;

			V.isResize2 = ( up.rf == 2 && V.resizeStep == zs + 1 &&
				!up.orig.taps.empty() );
		}
	}

	// First tap, tap count and phase offset of every output of the resize
	// step. After a zero-stuffing 2x upsampler only the even slots hold
	// samples, so a window that would start on an odd slot starts one slot
	// later, one tap into the phase (avir.h:6077-6083).
	static void bindPhases( PSteps& steps, const AxisVars& V )
	{
		PStep& fs = *steps[ V.resizeStep ];
		const int fl = fs.bank -> fltLen;
		const int back = fl / 2 - 1;

		for( int j = 0; j < fs.outLen; j++ )
		{
			RPos& rp = fs.rpos[ j ];

			if( fs.bankDyn )
			{
				fs.bank -> phase( rp.fti ); // build on demand
			}

			const int first = rp.srcPosInt - back;
			const int skip = ( V.isResize2 ? first & 1 : 0 );
			rp.srcOffs = first + skip;
			rp.ftpOff = skip;
			rp.fl = fl - skip;
		}
	}

	// calcComplexity, avir.h:6206-6270 (interleaved mode: 3/4 multiplier).
	static int cost( const PSteps& steps, const AxisVars& V,
		const std::vector< char >& useMap, const int lines )
	{
		const int fcnum = 3, fcdenom = 4;
		int s = 0, s2 = 0;

		for( int i = 0; i < (int) steps.size(); i++ )
		{
			const PStep& fs = *steps[ i ];
			s2 += 65 * fs.fltCap;

			if( fs.up )
			{
				if( !fs.orig.taps.empty() )
				{
					continue;
				}

				s += ( fs.fltCap * ( fs.inPrefix + fs.inLen + fs.inSuffix ) +
					fs.sdcCap + fs.pdcCap ) * V.elCount;
			}
			else
			if( fs.rf == 0 )
			{
				s += fs.bank -> fltLen * ( fs.bank -> order + V.elCount ) *
					fs.outLen;

				if( i == V.resizeStep && V.isResize2 )
				{
					s >>= 1;
				}

				s2 += fs.bank -> initCost( useMap );
			}
			else
			{
				s += fs.fltCap * V.elCount * fs.outLen * fcnum / fcdenom;
			}
		}

		return( s + s2 / lines );
	}

	// fillUsedFracMap, avir.h:6167-6183.
	static void usedPhases( const PStep& fs, std::vector< char >& m )
	{
		m.assign( fs.bank -> fracCount, 0 );

		for( int j = 0; j < fs.outLen; j++ )
		{
			m[ fs.rpos[ j ].fti ] |= 1;
		}
	}

	// Flattens one planned axis into the C ABI description.
	static void emitAxis( DescStore& S, const PSteps& steps, const AxisVars& V,
		const int mode, avirhip_axis& ax )
	{
		const int n = (int) steps.size();
		S.s.emplace_back( new std::vector< avirhip_step >( n ));
		std::vector< avirhip_step >& out = *S.s.back();

		for( int i = 0; i < n; i++ )
		{
			const PStep& fs = *steps[ i ];
			avirhip_step& o = out[ i ];
			memset( &o, 0, sizeof( o ));
			o.resample_factor = fs.rf;
			o.in_len = fs.inLen; o.in_prefix = fs.inPrefix;
			o.in_suffix = fs.inSuffix; o.out_len = fs.outLen;

			if( fs.rf == 0 )
			{
				const FracBank& B = *fs.bank;
				o.kind = ( V.isResize2 ? AVIRHIP_STEP_RESIZE2 :
					AVIRHIP_STEP_RESIZE );
				o.bank_filter_len = B.fltLen;
				o.bank_order = B.order;
				o.bank_frac_count = B.fracCount;

				std::map< int, int > used;

				for( int j = 0; j < fs.outLen; j++ )
				{
					used[ fs.rpos[ j ].fti ] = 0;
				}

				S.i.emplace_back( new std::vector< int32_t >() );
				std::vector< int32_t >& pf = *S.i.back();
				S.f.emplace_back( new std::vector< float >() );
				std::vector< float >& pt = *S.f.back();
				int pi = 0;

				for( auto& u : used )
				{
					u.second = pi++;
					pf.push_back( u.first );
					const float* t = &B.table[ (size_t) u.first * B.fltSize ];
					pt.insert( pt.end(), t, t + B.fltSize );
				}

				o.n_phases = pi;
				o.phase_fti = pf.data();
				o.phase_taps = pt.data();

				S.r.emplace_back( new std::vector< avirhip_rpos >(
					fs.outLen ));
				std::vector< avirhip_rpos >& rp = *S.r.back();

				for( int j = 0; j < fs.outLen; j++ )
				{
					const RPos& p = fs.rpos[ j ];
					rp[ j ].src_pos_int = p.srcPosInt;
					rp[ j ].fti = p.fti;
					rp[ j ].phase = used[ p.fti ];
					rp[ j ].x = p.x;
					rp[ j ].src_offs_px = p.srcOffs;
					rp[ j ].fl = p.fl;
					rp[ j ].ftp_off = p.ftpOff;
				}

				o.rpos = rp.data();
			}
			else
			if( fs.up )
			{
				o.out_prefix = fs.outPrefix;
				o.out_suffix = fs.outSuffix;

				if( !fs.orig.taps.empty() )
				{
					o.kind = AVIRHIP_STEP_UP_ZEROSTUFF;
				}
				else
				{
					o.kind = AVIRHIP_STEP_UP_FILTERED;
					o.flt_len = fs.fltCap;
					o.flt_latency = fs.lat;
					o.flt = keep( S, fs.flt );
					o.prefix_dc_len = fs.pdcCap;
					o.prefix_dc = keep( S, fs.pdc );
					o.suffix_dc_len = fs.sdcCap;
					o.suffix_dc = keep( S, fs.sdc );
				}
			}
			else
			{
				o.kind = AVIRHIP_STEP_FIR;
				o.edge_pixel_count = fs.edge;
				o.flt_len = fs.fltCap;
				o.flt_latency = fs.lat;
				o.flt = keep( S, fs.flt );
			}
		}

		ax.n_steps = n;
		ax.steps = out.data();
		ax.build_mode = mode;
	}

	static const float* keep( DescStore& S, const std::vector< float >& v )
	{
		S.f.emplace_back( new std::vector< float >( v ));
		return( S.f.back() -> data() );
	}
};

AvirPlanner::AvirPlanner( int res_bit_depth, int src_bit_depth,
	const avirhip_params& params )
{
	impl = new AvirPlannerImpl();
	impl -> P = params;
	impl -> resBits = res_bit_depth;
	impl -> srcBits = ( src_bit_depth == 0 ? res_bit_depth : src_bit_depth );

	// FixedFilterBank, avir.h:4637-4638.
	impl -> setupBank( impl -> fixedBank, 1.0, false, FirDesign() );
	impl -> fixedBank.buildAll();
}

AvirPlanner::~AvirPlanner()
{
	delete impl;
}

DescStore* AvirPlanner::build( int sw, int sh, int sstride, int nw, int nh,
	int ch, double k, const avirhip_vars& vars, int in_type, int out_type,
	int fppack )
{
	if( sw < 1 || sh < 1 || nw < 1 || nh < 1 || ch < 1 || ch > 4 ||
		in_type < 0 || in_type > 3 || out_type < 0 || out_type > 3 )
	{
		set_error( "planner: bad geometry / types" );
		return( nullptr );
	}

	AvirPlannerImpl& I = *impl;

	// Steps and offsets, avir.h:4709-4736.
	double kx, ky;
	double ox = vars.ox;
	double oy = vars.oy;

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
		kx = k;
		ky = k;
		const double ko = ( k - 1.0 ) * 0.5;
		ox += ko;
		oy += ko;
	}
	else
	{
		kx = -k;
		ky = -k;
	}

	// Output multiplier, avir.h:4740-4782. With sRGB gamma the range
	// conversion happens in the gamma stages (InGammaMult / OutGammaMult) and
	// the filters keep unity gain.
	const bool inFloat = ( in_type >= AVIRHIP_F32 );
	const bool outFloat = ( out_type >= AVIRHIP_F32 );
	double outMul;

	if( vars.UseSRGBGamma )
	{
		outMul = 1.0;
	}
	else
	{
		if( outFloat )
		{
			outMul = 1.0;
		}
		else
		{
			outMul = ( out_type == AVIRHIP_U8 ? 255.0 : 65535.0 );
		}

		if( !inFloat )
		{
			outMul /= ( in_type == AVIRHIP_U8 ? 255.0 : 65535.0 );
		}
	}

	if( sstride < 1 )
	{
		sstride = sw * ch;
	}

	AxisVars V;
	// ElCount counts `fptype` values per pixel (avir.h:4786-4787): with a
	// 4-float SIMD fptype (fpclass_float4) a pixel of 1-4 channels is ONE
	// element, and the build-mode cost model weighs the steps differently
	V.elCount = ( ch + fppack - 1 ) / fppack;
	V.k = 0.0; V.o = 0.0; V.resizeStep = 0; V.isResize2 = false;

	FracBank bank;
	PSteps steps;
	std::vector< char > useMap;
	int useMode = 1;
	const int modeCount = ( I.fixedBank.order == 0 ? 4 : 2 );

	// Horizontal axis: model every build mode, keep the cheapest
	// (avir.h:4813-4847).
	if( vars.BuildMode >= 0 )
	{
		useMode = vars.BuildMode;
	}
	else
	{
		int best = 0x7FFFFFFF;

		for( int m = 0; m < modeCount; m++ )
		{
			FracBank tb;
			PSteps ts;
			V.k = kx;
			V.o = ox;

			if( !I.buildSteps( ts, V, tb, outMul, m, true ))
			{
				set_error( "planner: degenerate filter parameters" );
				return( nullptr );
			}

			AvirPlannerImpl::updateBuffers( ts, V, sw, nw );
			AvirPlannerImpl::usedPhases( *ts[ V.resizeStep ], useMap );
			const int c = AvirPlannerImpl::cost( ts, V, useMap, sh );

			if( c < best )
			{
				useMode = m;
				best = c;
			}
		}
	}

	V.k = kx;
	V.o = ox;

	if( !I.buildSteps( steps, V, bank, outMul, useMode, false ))
	{
		set_error( "planner: degenerate filter parameters" );
		return( nullptr );
	}

	AvirPlannerImpl::updateBuffers( steps, V, sw, nw );
	AvirPlannerImpl::bindPhases( steps, V );

	std::unique_ptr< DescStore > S( new DescStore() );
	memset( &S -> d, 0, sizeof( S -> d ));
	memset( &S -> ld, 0, sizeof( S -> ld ));
	AvirPlannerImpl::emitAxis( *S, steps, V, useMode, S -> d.h );

	// Vertical axis, avir.h:4899-4954: re-model (phases already built for
	// the horizontal axis count as free), reuse the steps when possible.
	const int prevMode = useMode;

	if( vars.BuildMode >= 0 )
	{
		useMode = vars.BuildMode;
	}
	else
	{
		AxisVars TV = V;
		int best = 0x7FFFFFFF;

		for( int m = 0; m < modeCount; m++ )
		{
			FracBank tb;
			tb.inheritParams( bank );
			PSteps ts;
			TV.k = ky;
			TV.o = oy;

			if( !I.buildSteps( ts, TV, tb, 1.0, m, true ))
			{
				set_error( "planner: degenerate filter parameters" );
				return( nullptr );
			}

			AvirPlannerImpl::updateBuffers( ts, TV, sh, nh );
			AvirPlannerImpl::usedPhases( *ts[ TV.resizeStep ], useMap );
			const int c = AvirPlannerImpl::cost( ts, TV, useMap, nw );

			if( c < best )
			{
				useMode = m;
				best = c;
			}
		}
	}

	V.k = ky;
	V.o = oy;

	if( useMode == prevMode && ky == kx )
	{
		if( outMul != 1.0 )
		{
			// modifyCorrFilterDCGain, avir.h:6137-6157: rescaled in float.
			const double m = 1.0 / outMul;
			PStep& last = *steps.back();
			PStep& cf = ( !last.up && last.rf == 1 ? last : *steps[ 0 ]);

			for( size_t i = 0; i < cf.flt.size(); i++ )
			{
				cf.flt[ i ] = (float) ( (double) cf.flt[ i ] * m );
			}
		}
	}
	else
	{
		if( !I.buildSteps( steps, V, bank, 1.0, useMode, false ))
		{
			set_error( "planner: degenerate filter parameters" );
			return( nullptr );
		}
	}

	AvirPlannerImpl::updateBuffers( steps, V, sh, nh );
	AvirPlannerImpl::bindPhases( steps, V );
	AvirPlannerImpl::emitAxis( *S, steps, V, useMode, S -> d.v );

	avirhip_plan_desc& d = S -> d;
	d.src_w = sw; d.src_h = sh; d.src_stride_elems = sstride;
	d.new_w = nw; d.new_h = nh; d.channels = ch;
	d.in_type = in_type; d.out_type = out_type;
	d.tr_mul = 1.0;
	d.pk_out = 0.0;
	d.use_srgb_gamma = ( vars.UseSRGBGamma ? 1 : 0 );
	d.alpha_index = vars.AlphaIndex;
	d.dither = AVIRHIP_DITHER_DEF; // the front end fills in its ditherer

	if( !outFloat )
	{
		// avir.h:5029-5045.
		int truncBits, outRange;

		if( out_type == AVIRHIP_U8 )
		{
			truncBits = 8 - I.resBits;
			outRange = 255;
		}
		else
		{
			truncBits = 16 - I.resBits;
			outRange = 65535;
		}

		d.pk_out = outRange;
		d.tr_mul = ( truncBits > 0 ?
			d.pk_out / ( outRange >> truncBits ) : 1.0 );
	}

	return( S.release() );
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
		P.la < 2.0 || in_type < 0 || in_type > 2 || out_type < 0 ||
		out_type > 2 )
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
	const bool inFloat = ( in_type >= AVIRHIP_F32 );
	const bool outFloat = ( out_type >= AVIRHIP_F32 );
	const int szin = ( in_type == AVIRHIP_U8 ? 1 : in_type == AVIRHIP_U16 ? 2 : 4 );
	const int szout = ( out_type == AVIRHIP_U8 ? 1 : out_type == AVIRHIP_U16 ? 2 : 4 );
	d.is_unity_mul = (( inFloat && outFloat ) ||
		( inFloat == outFloat && szin == szout )) ? 1 : 0;
	d.clamp = ( szout == 1 ? 255.0f : 65535.0f );
	d.out_mul = ( outFloat ? 1.0f : d.clamp ) /
		( inFloat ? 1.0f : ( szin == 1 ? 255.0f : 65535.0f ));

	return( S.release() );
}

} // namespace avirhip
