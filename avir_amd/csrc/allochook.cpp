// allochook.cpp -- failure injection for the host allocations of the planning /
// plan-creation code: the test hook avirhip_debug_fail_alloc() (include/avirhip.h)
// makes the calling thread's n-th allocation inside the library throw.
//
// The library defines operator new (LOCAL to libavirhip.so: the linker's
// version script avirhip.map exports avirhip_* only, so nothing in the host
// application is replaced) but NOT operator delete, and its operator new hands
// every request to the process-global operator new, looked up once with
// dlsym( RTLD_DEFAULT ): libstdc++'s, or whatever the host application or a
// sanitizer runtime replaced it with. So every block -- whether it was
// requested by code compiled into this library or by libstdc++'s out-of-line
// code on its behalf (std::string growth and the like) -- comes from the global
// allocator and goes back to the global operator delete: no block ever crosses
// between two allocators, whatever the host uses (the previous form
// paired its own malloc-based new with free(), which a host with a non-malloc
// operator new, or ASan's alloc-dealloc-mismatch check, does not survive).

#include <dlfcn.h>
#include <stdlib.h>
#include <stddef.h>
#include <new>

namespace {

thread_local long g_countdown = 0; // > 0: that many allocations to the failure
thread_local long g_count = 0;     // allocations seen since the last hook call

// true: this allocation is the one that has to fail
inline bool hook_fails()
{
	g_count++;
	return( g_countdown > 0 && --g_countdown == 0 );
}

typedef void* ( *fn_new )( size_t );
typedef void* ( *fn_new_nt )( size_t, const std::nothrow_t& );
typedef void* ( *fn_new_al )( size_t, std::align_val_t );
typedef void* ( *fn_new_al_nt )( size_t, std::align_val_t, const std::nothrow_t& );

// the process-global operators, by their Itanium-ABI names (this library's own
// definitions are local symbols: the global scope cannot return them)
template< typename F >
inline F global_op( const char* const mangled )
{
	void* const f = dlsym( RTLD_DEFAULT, mangled );

	if( f == nullptr )
	{
		abort(); // no C++ runtime in the process: cannot happen
	}

	return( (F) f );
}

} // namespace

extern "C" long avirhip_debug_fail_alloc( long nth )
{
	const long seen = g_count;
	g_count = 0;
	g_countdown = ( nth > 0 ? nth : 0 );
	return( seen );
}

void* operator new( size_t n )
{
	static const fn_new f = global_op< fn_new >( "_Znwm" );
	if( hook_fails() ) throw std::bad_alloc();
	return( f( n ));
}

void* operator new[]( size_t n )
{
	static const fn_new f = global_op< fn_new >( "_Znam" );
	if( hook_fails() ) throw std::bad_alloc();
	return( f( n ));
}

void* operator new( size_t n, const std::nothrow_t& t ) noexcept
{
	static const fn_new_nt f = global_op< fn_new_nt >( "_ZnwmRKSt9nothrow_t" );
	if( hook_fails() ) return( nullptr );
	return( f( n, t ));
}

void* operator new[]( size_t n, const std::nothrow_t& t ) noexcept
{
	static const fn_new_nt f = global_op< fn_new_nt >( "_ZnamRKSt9nothrow_t" );
	if( hook_fails() ) return( nullptr );
	return( f( n, t ));
}

void* operator new( size_t n, std::align_val_t a )
{
	static const fn_new_al f = global_op< fn_new_al >( "_ZnwmSt11align_val_t" );
	if( hook_fails() ) throw std::bad_alloc();
	return( f( n, a ));
}

void* operator new[]( size_t n, std::align_val_t a )
{
	static const fn_new_al f = global_op< fn_new_al >( "_ZnamSt11align_val_t" );
	if( hook_fails() ) throw std::bad_alloc();
	return( f( n, a ));
}

void* operator new( size_t n, std::align_val_t a, const std::nothrow_t& t ) noexcept
{
	static const fn_new_al_nt f = global_op< fn_new_al_nt >(
		"_ZnwmSt11align_val_tRKSt9nothrow_t" );
	if( hook_fails() ) return( nullptr );
	return( f( n, a, t ));
}

void* operator new[]( size_t n, std::align_val_t a, const std::nothrow_t& t ) noexcept
{
	static const fn_new_al_nt f = global_op< fn_new_al_nt >(
		"_ZnamSt11align_val_tRKSt9nothrow_t" );
	if( hook_fails() ) return( nullptr );
	return( f( n, a, t ));
}
