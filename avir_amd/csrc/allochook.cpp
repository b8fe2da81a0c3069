// allochook.cpp -- the library's own operator new / delete, so that the test
// hook avirhip_debug_fail_alloc() (include/avirhip.h) can make any host
// allocation of the planning / plan-creation code fail on demand. The linker's
// version script (avirhip.map) keeps these definitions LOCAL to libavirhip.so:
// they replace nothing in the host application, and only allocations made by
// code inside the library -- its std::vector, std::map, std::thread ... --
// go through them. Memory comes from malloc / free (what libstdc++'s own
// operators use), so blocks may cross to libstdc++'s operator delete.

#include <stdlib.h>
#include <stddef.h>
#include <new>

namespace {

thread_local long g_countdown = 0; // > 0: that many allocations to the failure
thread_local long g_count = 0;     // allocations seen since the last hook call

inline void* hooked_alloc( size_t n, size_t al )
{
	g_count++;

	if( g_countdown > 0 && --g_countdown == 0 )
	{
		return( nullptr );
	}

	if( n == 0 )
	{
		n = 1;
	}

	if( al <= alignof( ::max_align_t ))
	{
		return( malloc( n ));
	}

	void* p = nullptr;
	return( posix_memalign( &p, al, n ) == 0 ? p : nullptr );
}

inline void* alloc_or_throw( size_t n, size_t al )
{
	void* p = hooked_alloc( n, al );

	if( p == nullptr )
	{
		throw std::bad_alloc();
	}

	return( p );
}

} // namespace

extern "C" long avirhip_debug_fail_alloc( long nth )
{
	const long seen = g_count;
	g_count = 0;
	g_countdown = ( nth > 0 ? nth : 0 );
	return( seen );
}

void* operator new( size_t n ) { return( alloc_or_throw( n, 1 )); }
void* operator new[]( size_t n ) { return( alloc_or_throw( n, 1 )); }
void* operator new( size_t n, const std::nothrow_t& ) noexcept
	{ return( hooked_alloc( n, 1 )); }
void* operator new[]( size_t n, const std::nothrow_t& ) noexcept
	{ return( hooked_alloc( n, 1 )); }
void* operator new( size_t n, std::align_val_t a )
	{ return( alloc_or_throw( n, (size_t) a )); }
void* operator new[]( size_t n, std::align_val_t a )
	{ return( alloc_or_throw( n, (size_t) a )); }
void* operator new( size_t n, std::align_val_t a, const std::nothrow_t& ) noexcept
	{ return( hooked_alloc( n, (size_t) a )); }
void* operator new[]( size_t n, std::align_val_t a, const std::nothrow_t& ) noexcept
	{ return( hooked_alloc( n, (size_t) a )); }

void operator delete( void* p ) noexcept { free( p ); }
void operator delete[]( void* p ) noexcept { free( p ); }
void operator delete( void* p, size_t ) noexcept { free( p ); }
void operator delete[]( void* p, size_t ) noexcept { free( p ); }
void operator delete( void* p, const std::nothrow_t& ) noexcept { free( p ); }
void operator delete[]( void* p, const std::nothrow_t& ) noexcept { free( p ); }
void operator delete( void* p, std::align_val_t ) noexcept { free( p ); }
void operator delete[]( void* p, std::align_val_t ) noexcept { free( p ); }
void operator delete( void* p, size_t, std::align_val_t ) noexcept { free( p ); }
void operator delete[]( void* p, size_t, std::align_val_t ) noexcept { free( p ); }
void operator delete( void* p, std::align_val_t, const std::nothrow_t& ) noexcept
	{ free( p ); }
void operator delete[]( void* p, std::align_val_t, const std::nothrow_t& ) noexcept
	{ free( p ); }
