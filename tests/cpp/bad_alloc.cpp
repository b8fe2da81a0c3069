// tests/cpp/bad_alloc.cpp -- the C++ drop-in front end under host-memory
// exhaustion: the library returns AVIRHIP_ENOMEM across its C boundary (no
// exception crosses extern "C"), and the headers rethrow it as what the
// reference itself throws, std::bad_alloc (avir.h:564-827 CBuffer::alloc,
// lancir.h:769-815 reallocBuf). Needs no GPU: the failing allocation is in the
// host planner, before any device call. Prints one line per case.
#include <cstdio>
#include <cstdint>
#include <new>
#include <vector>
#include "avir.h"
#include "lancir.h"

int main()
{
	std :: vector< uint8_t > In( 64 * 48 * 3, 7 ), Out( 100 * 77 * 3 );
	int bad = 0;

	{
		avir :: CImageResizer<> R( 8 );

		for( long n = 1; n <= 40; n += 13 )
		{
			avirhip_debug_fail_alloc( n );

			try
			{
				R.resizeImage( In.data(), 64, 48, 0, Out.data(), 100, 77, 3, 0.0 );
				printf( "avir n=%ld no-throw\n", n );
				bad++;
			}
			catch( const std :: bad_alloc& )
			{
				printf( "avir n=%ld bad_alloc\n", n );
			}
			catch( const std :: exception& e )
			{
				printf( "avir n=%ld other: %s\n", n, e.what() );
				bad++;
			}

			avirhip_debug_fail_alloc( 0 );
		}
	}

	{
		avir :: CLancIR L;

		for( long n = 1; n <= 9; n += 4 )
		{
			avirhip_debug_fail_alloc( n );

			try
			{
				L.resizeImage( In.data(), 64, 48, Out.data(), 100, 77, 3 );
				printf( "lancir n=%ld no-throw\n", n );
				bad++;
			}
			catch( const std :: bad_alloc& )
			{
				printf( "lancir n=%ld bad_alloc\n", n );
			}
			catch( const std :: exception& e )
			{
				printf( "lancir n=%ld other: %s\n", n, e.what() );
				bad++;
			}

			avirhip_debug_fail_alloc( 0 );
		}
	}

	// constructors
	avirhip_debug_fail_alloc( 1 );

	try
	{
		avir :: CImageResizer<> R2( 8 );
		printf( "ctor no-throw\n" );
		bad++;
	}
	catch( const std :: bad_alloc& )
	{
		printf( "ctor bad_alloc\n" );
	}

	avirhip_debug_fail_alloc( 0 );
	printf( bad == 0 ? "OK\n" : "FAILED\n" );
	return( bad );
}
