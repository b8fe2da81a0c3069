// planner_avir.inl -- the part of the AVIR planner that depends on the
// pipeline's floating-point type (the reference's `fptype`: tables are stored,
// accumulated and read back in it, avir.h:5128-6270). Included twice by
// planner.cpp: FT = float in namespace pl32, FT = double in namespace pl64.
//
// A declared derived work: this file restates avir.h:460-2100, 5128-6270 of
// AVIR v3.1 (Copyright (c) 2015-2025 Aleksey Vaneev, MIT License -- the licence
// text is at the top of planner.cpp, which is the only file that includes this
// one; NOTICE.md) statement by statement, because the parity contract is
// bit-identical filter tables and that fixes every expression and its order.

namespace avirhip {

namespace PL_NS {

// calcFIRFilterResponse<FT>, avir.h:460-503 (fltlat = 0).
void firResponse( const FT* flt, int n, const double th, double& re0,
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

// Polyphase fractional-delay bank with order-0/1 interpolation and optional
// embedded external low-pass (CDSPFracFilterBankLin<FT>, avir.h:1647-2100).
class FracBank
{
public:
	double wfLen2, wfFreq, wfAlpha;
	int fracCount, order, srcLen, fltLen, fltSize;
	bool initRequired, srcBuilt;
	FirDesign ext;
	std::vector< double > src; // (fracCount+1) blocks of srcLen
	std::vector< FT > table; // (fracCount+1) * fltSize
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
	const FT* phase( const int i )
	{
		if( !srcBuilt )
		{
			buildSource();
		}

		FT* const res = &table[ (size_t) i * fltSize ];

		if(( flags[ i ] & 2 ) == 0 )
		{
			makePhase( i );
			flags[ i ] |= 2;

			if( order > 0 )
			{
				makePhase( i + 1 );
				const FT* const nxt = res + fltSize;
				FT* const dif = res + fltLen;

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
		FT* op = &table[ (size_t) n * fltSize ];

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
				op[ i ] = (FT) sf[ i ];
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

			op[ j ] = (FT) s;
		}
	}
};

struct RPos
{
	int srcPosInt, fti, srcOffs, fl, ftpOff;
	FT x;
};

// One filtering step (CImageResizerFilterStep, avir.h:2568-2728). `fltCap`,
// `pdcCap`, `sdcCap` track CBuffer capacities, which exist in modelling mode
// where the contents are not built.
struct PStep
{
	bool up;
	int rf;
	std::vector< FT > flt;
	int fltCap;
	FirDesign orig;
	double dcgain;
	int lat;
	int inLen, inPrefix, inSuffix, outLen, outPrefix, outSuffix;
	std::vector< FT > pdc, sdc;
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
				fs.flt[ i ] = (FT) fs.orig.taps[ i ];
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

			// DC "tails" added at the scanline ends, accumulated in FT
			// (copyArray/addArray, avir.h:5320-5353).
			fs.pdc.assign( l, 0.0f );
			fs.sdc.assign( fs.lat, 0.0f );
			const FT* ip = &fs.flt[ fs.lat + rf ];

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
			FT* op = fs.sdc.data();

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

			const FT* flt;
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
			nfs.flt[ i ] = (FT) filter[ i ];
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
					fs.rpos[ j ].x = (FT) ( x - fti );
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
				std::vector< FT >& pt = desc_pool( S, (FT*) nullptr );
				int pi = 0;

				for( auto& u : used )
				{
					u.second = pi++;
					pf.push_back( u.first );
					const FT* t = &B.table[ (size_t) u.first * B.fltSize ];
					pt.insert( pt.end(), t, t + B.fltSize );
				}

				o.n_phases = pi;
				o.phase_fti = pf.data();
				desc_set( o.phase_taps, o.phase_taps64, pt.data() );

				S.r.emplace_back( new std::vector< avirhip_rpos >(
					fs.outLen ));
				std::vector< avirhip_rpos >& rp = *S.r.back();

				for( int j = 0; j < fs.outLen; j++ )
				{
					const RPos& p = fs.rpos[ j ];
					rp[ j ].src_pos_int = p.srcPosInt;
					rp[ j ].fti = p.fti;
					rp[ j ].phase = used[ p.fti ];
					desc_set_x( rp[ j ], p.x );
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
					desc_set( o.flt, o.flt64, keep( S, fs.flt ));
					o.prefix_dc_len = fs.pdcCap;
					desc_set( o.prefix_dc, o.prefix_dc64, keep( S, fs.pdc ));
					o.suffix_dc_len = fs.sdcCap;
					desc_set( o.suffix_dc, o.suffix_dc64, keep( S, fs.sdc ));
				}
			}
			else
			{
				o.kind = AVIRHIP_STEP_FIR;
				o.edge_pixel_count = fs.edge;
				o.flt_len = fs.fltCap;
				o.flt_latency = fs.lat;
				desc_set( o.flt, o.flt64, keep( S, fs.flt ));
			}
		}

		ax.n_steps = n;
		ax.steps = out.data();
		ax.build_mode = mode;
	}

	static const FT* keep( DescStore& S, const std::vector< FT >& v )
	{
		std::vector< FT >& q = desc_pool( S, (FT*) nullptr );
		q = v;
		return( q.data() );
	}
};

DescStore* build_impl( AvirPlannerImpl& I, int sw, int sh, int sstride, int nw,
	int nh, int ch, double k, const avirhip_vars& vars, int in_type,
	int out_type, int fppack )
{
	if( sw < 1 || sh < 1 || nw < 1 || nh < 1 || ch < 1 || ch > 4 ||
		in_type < 0 || in_type > 3 || out_type < 0 || out_type > 3 )
	{
		set_error( "planner: bad geometry / types" );
		return( nullptr );
	}


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
	// 4-FT SIMD fptype (fpclass_float4) a pixel of 1-4 channels is ONE
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
			// modifyCorrFilterDCGain, avir.h:6137-6157: rescaled in FT.
			const double m = 1.0 / outMul;
			PStep& last = *steps.back();
			PStep& cf = ( !last.up && last.rf == 1 ? last : *steps[ 0 ]);

			for( size_t i = 0; i < cf.flt.size(); i++ )
			{
				cf.flt[ i ] = (FT) ( (double) cf.flt[ i ] * m );
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
	d.work_f64 = PL_FT;

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

} // namespace PL_NS

} // namespace avirhip
