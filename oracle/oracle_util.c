/* oracle/oracle_util.c -- TEST INFRASTRUCTURE ONLY: hashing and the input
 * generators of SURVEY.md 8(c). */
#include "oracle.h"

uint64_t oracle_fnv1a64( const void* p, size_t n )
{
	const uint8_t* b = (const uint8_t*) p;
	uint64_t h = 1469598103934665603ULL;
	size_t i;

	for( i = 0; i < n; i++ )
	{
		h ^= b[ i ];
		h *= 1099511628211ULL;
	}

	return( h );
}

void oracle_fill_u8( uint8_t* p, size_t n, uint32_t seed )
{
	uint32_t s = seed;
	size_t i;

	for( i = 0; i < n; i++ )
	{
		s = s * 1664525u + 1013904223u;
		p[ i ] = (uint8_t) ( s >> 24 );
	}
}

void oracle_fill_f32( float* p, size_t n, uint32_t seed )
{
	uint32_t s = seed;
	size_t i;

	for( i = 0; i < n; i++ )
	{
		s = s * 1664525u + 1013904223u;
		p[ i ] = (float) ( s >> 8 ) * ( 1.0f / 16777216.0f );
	}
}
