// examples/resize_example.cpp -- a program written against the REFERENCE's API
// (avir::CImageResizer<>, avir::CLancIR; reference README.md:83-150), compiled
// against the drop-in headers of this repository instead:
//
//   g++ -O2 -Iinclude/avir_hip examples/resize_example.cpp \
//       -Lavir_amd/lib -lavirhip -Wl,-rpath,$PWD/avir_amd/lib \
//       -Wl,-rpath-link,/opt/rocm/lib -o resize_example
//
// It resizes synthetic images (the LCG generators of SURVEY.md 8c) and prints
// the FNV-1a-64 hash of each result, which tests/test_cpp_frontend.py compares
// with the golden hashes of the reference's own output.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "avir.h"
#include "avir_float4_sse.h"
#include "lancir.h"

static uint64_t fnv1a64( const void* p, size_t n )
{
	const uint8_t* b = (const uint8_t*) p;
	uint64_t h = 1469598103934665603ULL;

	for( size_t i = 0; i < n; i++ )
	{
		h ^= b[ i ];
		h *= 1099511628211ULL;
	}

	return( h );
}

int main()
{
	try
	{
		// cfg1 of BASELINE.json: 640x480 -> 1024x768 RGB uint8
		std :: vector< uint8_t > In8( 640 * 480 * 3 );
		uint32_t s = 1;

		for( size_t i = 0; i < In8.size(); i++ )
		{
			s = s * 1664525u + 1013904223u;
			In8[ i ] = (uint8_t) ( s >> 24 );
		}

		std :: vector< uint8_t > Out8( 1024 * 768 * 3 );
		avir :: CImageResizer<> ImageResizer( 8 );
		ImageResizer.resizeImage( In8.data(), 640, 480, 0, Out8.data(), 1024,
			768, 3, 0 );

		printf( "cfg1 %016llx\n", (unsigned long long) fnv1a64( Out8.data(),
			Out8.size() ));

		// RGBA float, non-default parameters and variables
		std :: vector< float > InF( 640 * 480 * 4 );
		s = 12345;

		for( size_t i = 0; i < InF.size(); i++ )
		{
			s = s * 1664525u + 1013904223u;
			InF[ i ] = (float) ( s >> 8 ) * ( 1.0f / 16777216.0f );
		}

		std :: vector< float > OutF( 1024 * 768 * 4 );
		avir :: CImageResizer< avir :: fpclass_def< float > > Resizer16( 16 );
		avir :: CImageResizerVars Vars;
		Resizer16.resizeImage( InF.data(), 640, 480, 0, OutF.data(), 1024, 768,
			4, 0.0, &Vars );

		printf( "f32x4_16 %016llx\n", (unsigned long long) fnv1a64(
			OutF.data(), OutF.size() * sizeof( float )));

		avir :: CImageResizer<> ResizerUltra( 8, 0,
			avir :: CImageResizerParamsUltra() );

		std :: vector< float > OutU( 400 * 300 * 4 );
		ResizerUltra.resizeImage( InF.data(), 640, 480, 0, OutU.data(), 400,
			300, 4, 0.0 );

		printf( "ultra %016llx\n", (unsigned long long) fnv1a64( OutU.data(),
			OutU.size() * sizeof( float )));

		// LANCIR, 4-channel float
		avir :: CLancIR LancIR;
		const int rc = LancIR.resizeImage( InF.data(), 640, 480, OutF.data(),
			1024, 768, 4 );

		printf( "lancir rc=%d %016llx\n", rc, (unsigned long long) fnv1a64(
			OutF.data(), OutF.size() * sizeof( float )));

		// error-diffusion dithering, exactly as the reference's README enables
		// it (README.md:135-142)
		typedef avir :: fpclass_def< float, float,
			avir :: CImageResizerDithererErrdINL< float > > fpclass_dith;
		avir :: CImageResizer< fpclass_dith > ImageResizerDith( 8 );
		ImageResizerDith.resizeImage( In8.data(), 640, 480, 0, Out8.data(),
			1024, 768, 3, 0 );

		printf( "errd %016llx\n", (unsigned long long) fnv1a64( Out8.data(),
			Out8.size() ));

		// the SIMD value type, exactly as the reference's README enables it
		// (README.md:174-186): 8-bit RGB up, 16-bit RGBA-float down, and the
		// variables it writes back
		avir :: CImageResizer< avir :: fpclass_float4 > ImageResizerF4( 8 );
		avir :: CImageResizerVars VarsF4;
		ImageResizerF4.resizeImage( In8.data(), 640, 480, 0, Out8.data(),
			1024, 768, 3, 0, &VarsF4 );

		printf( "float4_u8 %016llx\n", (unsigned long long) fnv1a64(
			Out8.data(), Out8.size() ));

		printf( "float4_vars %d %d %d %d %d %d %d\n", VarsF4.ElCount,
			VarsF4.fppack, VarsF4.fpalign, VarsF4.BufLen[ 0 ],
			VarsF4.BufLen[ 1 ], VarsF4.BufOffs[ 0 ], VarsF4.BufOffs[ 1 ]);

		std :: vector< uint16_t > Out16( 400 * 300 * 4 );
		avir :: CImageResizer< avir :: fpclass_float4 > ImageResizerF4b( 12 );
		ImageResizerF4b.resizeImage( InF.data(), 640, 480, 0, Out16.data(),
			400, 300, 4, 0 );

		printf( "float4_u16 %016llx\n", (unsigned long long) fnv1a64(
			Out16.data(), Out16.size() * 2 ));

		// the double pipeline (avir.h:4553-4560): double in, double out (the
		// vertical pass' in-place result), and 8-bit RGB through it
		std :: vector< double > InD( InF.begin(), InF.end() );
		std :: vector< double > OutD( 400 * 300 * 4 );
		avir :: CImageResizer< avir :: fpclass_def< double > > ImageResizerD( 16 );
		ImageResizerD.resizeImage( InD.data(), 640, 480, 0, OutD.data(), 400,
			300, 4, 0 );

		printf( "double_f64 %016llx\n", (unsigned long long) fnv1a64(
			OutD.data(), OutD.size() * 8 ));

		avir :: CImageResizer< avir :: fpclass_def< double > > ImageResizerD8( 8 );
		ImageResizerD8.resizeImage( In8.data(), 640, 480, 0, Out8.data(), 1024,
			768, 3, 0 );

		printf( "double_u8 %016llx\n", (unsigned long long) fnv1a64(
			Out8.data(), Out8.size() ));

		// CImageResizerVarsBase is written back (avir.h:4663-4669)
		avir :: CImageResizerVars Vars2;
		Vars2.UseSRGBGamma = true;
		std :: vector< uint8_t > OutS( 300 * 200 * 3 );
		ImageResizer.resizeImage( In8.data(), 640, 480, 0, OutS.data(), 300,
			200, 3, 0, &Vars2 );

		printf( "vars %d %d %d %d %d %d %.17g %.17g %d %d %.17g %.17g\n",
			Vars2.ElCount, Vars2.BufLen[ 0 ], Vars2.BufLen[ 1 ],
			Vars2.BufOffs[ 0 ], Vars2.BufOffs[ 1 ], Vars2.ResizeStep, Vars2.k,
			Vars2.o, (int) Vars2.IsResize2, Vars2.packmode, Vars2.InGammaMult,
			Vars2.OutGammaMult );

		// LANCIR's legacy 13-argument overload with padded scanlines
		// (lancir.h:744-755)
		std :: vector< float > InP( 480 * ( 640 * 4 + 8 ), 7.0f );
		std :: vector< float > OutP( 300 * ( 400 * 4 + 12 ), 5.0f );

		for( int y = 0; y < 480; y++ )
		{
			for( int x = 0; x < 640 * 4; x++ )
			{
				InP[ (size_t) y * ( 640 * 4 + 8 ) + x ] =
					InF[ (size_t) y * 640 * 4 + x ];
			}
		}

		const int rcl = LancIR.resizeImage( InP.data(), 640, 480, 640 * 4 + 8,
			OutP.data(), 400, 300, 400 * 4 + 12, 4, 0.0, 0.0, 0.25, -0.5 );

		printf( "lancir_legacy rc=%d %016llx\n", rcl, (unsigned long long)
			fnv1a64( OutP.data(), OutP.size() * sizeof( float )));

		// double and uint32_t elements (lancir.h:373-381: "`uint32_t` type is
		// treated as `uint16_t`")
		{
			std :: vector< double > InD( InF.begin(), InF.end() );
			std :: vector< double > OutD( 400 * 300 * 4 );
			const int rcd = LancIR.resizeImage< double, double >( InD.data(),
				640, 480, OutD.data(), 400, 300, 4 );

			printf( "lancir_f64 rc=%d %016llx\n", rcd, (unsigned long long)
				fnv1a64( OutD.data(), OutD.size() * sizeof( double )));

			std :: vector< uint32_t > Out32( 1024 * 768 * 3 );
			const int rcu = LancIR.resizeImage( In8.data(), 640, 480,
				Out32.data(), 1024, 768, 3 );

			printf( "lancir_u8_u32 rc=%d %016llx\n", rcu, (unsigned long long)
				fnv1a64( Out32.data(), Out32.size() * sizeof( uint32_t )));

			std :: vector< uint8_t > OutD8( 1024 * 768 * 4 );
			const int rcb = LancIR.resizeImage( InD.data(), 640, 480,
				OutD8.data(), 1024, 768, 4 );

			printf( "lancir_f64_u8 rc=%d %016llx\n", rcb, (unsigned long long)
				fnv1a64( OutD8.data(), OutD8.size() ));
		}

		// parameter error -> 0, like the reference
		printf( "lancir_err rc=%d\n", LancIR.resizeImage( InF.data(), 640, 480,
			OutF.data(), 0, 768, 4 ));
	}
	catch( const std :: exception& e )
	{
		fprintf( stderr, "error: %s\n", e.what() );
		return( 1 );
	}

	return( 0 );
}
