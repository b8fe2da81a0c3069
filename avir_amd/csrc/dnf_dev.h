// dnf_dev.h -- the building blocks of dnf.hip's row waves: LDS-DMA with a
// scalar row offset, the packed multiply-add statements, batched ds_read_b64.
// DF_CC: the asm constraint of a tap pair ("s": uniform taps in SGPR pairs;
// an includer with per-lane taps says "v" -- tried for non-integer ratios,
// profiles/README.md round 3: the lane stride of ~8 pixels makes the window
// reads collide in the LDS banks).
#ifndef AVIRHIP_DNF_DEV_H
#define AVIRHIP_DNF_DEV_H

#include "gpass_dev.h"
#include "dn_dev.h"

#ifndef DF_CC
#define DF_CC "s"
#endif

namespace avirhip {

// LDS-DMA of 64 x 16 bytes with a scalar row offset (gp_dma + soffset)
__device__ __forceinline__ void df_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff, const unsigned soff )
{
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"s_nop 0\n\t"
		"buffer_load_dwordx4 %1, %2, %3 offen lds"
		:: "s"( lds ), "v"( voff ), "s"( rs ), "s"( soff ) : "memory" );
}

// slot (16-byte unit) of relative input pixel i inside a row buffer of NP * 64
// slots: K = 3 as it comes; K = 2 even pixels first, then the odd ones
template< int K, int NP >
__device__ __forceinline__ constexpr int df_slot( const int i )
{
	return( K == 2 ? ( i & 1 ) * ( NP * 32 ) + ( i >> 1 ) : i );
}

// Packed multiply-adds of a row wave: one input half pixel (two channels)
// times the taps it has in 1-3 of the lane's outputs, then the adds -- one asm
// statement per input (hipcc pads every statement boundary with a wait state
// and never moves a load across one; products first, sums second: no
// instruction waits for its predecessor). A tap is one half of an SGPR pair
// {c[2m], c[2m+1]}: op_sel picks it for both result halves.
#define DF_SEL0 " op_sel_hi:[1,0]"
#define DF_SEL1 " op_sel:[0,1] op_sel_hi:[1,1]"
#define DF_PK1_BODY( S0 ) asm volatile( \
	"v_pk_mul_f32 %1, %2, %3" S0 "\n\t" \
	"v_pk_add_f32 %0, %0, %1" \
	: "+v"( a0 ), "=&v"( p0 ) : "v"( v ), DF_CC( c0 ))
#define DF_PK2_BODY( S0, S1 ) asm volatile( \
	"v_pk_mul_f32 %2, %4, %5" S0 "\n\t" \
	"v_pk_mul_f32 %3, %4, %6" S1 "\n\t" \
	"v_pk_add_f32 %0, %0, %2\n\t" \
	"v_pk_add_f32 %1, %1, %3" \
	: "+v"( a0 ), "+v"( a1 ), "=&v"( p0 ), "=&v"( p1 ) \
	: "v"( v ), DF_CC( c0 ), DF_CC( c1 ))
#define DF_PK3_BODY( S0, S1, S2 ) asm volatile( \
	"v_pk_mul_f32 %3, %6, %7" S0 "\n\t" \
	"v_pk_mul_f32 %4, %6, %8" S1 "\n\t" \
	"v_pk_mul_f32 %5, %6, %9" S2 "\n\t" \
	"v_pk_add_f32 %0, %0, %3\n\t" \
	"v_pk_add_f32 %1, %1, %4\n\t" \
	"v_pk_add_f32 %2, %2, %5" \
	: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "=&v"( p0 ), "=&v"( p1 ), \
	"=&v"( p2 ) : "v"( v ), DF_CC( c0 ), DF_CC( c1 ), DF_CC( c2 ))

// PAR: bit k = parity of the tap of accumulator k
template< int PAR >
__device__ __forceinline__ void df_pk1( f2& a0, const f2 v, const f2 c0 )
{
	f2 p0;
	if constexpr( PAR == 0 ) DF_PK1_BODY( DF_SEL0 );
	else DF_PK1_BODY( DF_SEL1 );
}

template< int PAR >
__device__ __forceinline__ void df_pk2( f2& a0, f2& a1, const f2 v,
	const f2 c0, const f2 c1 )
{
	f2 p0, p1;
	if constexpr( PAR == 0 ) DF_PK2_BODY( DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 1 ) DF_PK2_BODY( DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 2 ) DF_PK2_BODY( DF_SEL0, DF_SEL1 );
	else DF_PK2_BODY( DF_SEL1, DF_SEL1 );
}

template< int PAR >
__device__ __forceinline__ void df_pk3( f2& a0, f2& a1, f2& a2, const f2 v,
	const f2 c0, const f2 c1, const f2 c2 )
{
	f2 p0, p1, p2;
	if constexpr( PAR == 0 ) DF_PK3_BODY( DF_SEL0, DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 1 ) DF_PK3_BODY( DF_SEL1, DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 2 ) DF_PK3_BODY( DF_SEL0, DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 3 ) DF_PK3_BODY( DF_SEL1, DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 4 ) DF_PK3_BODY( DF_SEL0, DF_SEL0, DF_SEL1 );
	else if constexpr( PAR == 5 ) DF_PK3_BODY( DF_SEL1, DF_SEL0, DF_SEL1 );
	else if constexpr( PAR == 6 ) DF_PK3_BODY( DF_SEL0, DF_SEL1, DF_SEL1 );
	else DF_PK3_BODY( DF_SEL1, DF_SEL1, DF_SEL1 );
}

// ---- row waves: the resize inputs of a lane, ds_read_b64 in batches that
// are issued one batch ahead of their use (asm: hipcc would merge them into
// ds_read2_b64, half the LDS rate, and waits for every load it issues between
// two asm statements on the spot)
template< int B >
struct DfBatch
{
	f2 v[ B ];
};

#define DF_LD( i ) "ds_read_b64 %" #i ", %[ad] offset:%[o" #i "]\n\t"
#define DF_LDO( i ) "=v"( d.v[ i ])
#define DF_LDI( i ) [o##i] "i"( O##i )

template< int B, int O0, int O1, int O2, int O3, int O4, int O5, int O6,
	int O7 = 0, int O8 = 0, int O9 = 0, int O10 = 0 >
__device__ __forceinline__ void df_load( DfBatch< B >& d, const unsigned ad )
{
	if constexpr( B == 11 )
	{
		asm volatile( DF_LD( 0 ) DF_LD( 1 ) DF_LD( 2 ) DF_LD( 3 ) DF_LD( 4 )
			DF_LD( 5 ) DF_LD( 6 ) DF_LD( 7 ) DF_LD( 8 ) DF_LD( 9 ) DF_LD( 10 )
			: DF_LDO( 0 ), DF_LDO( 1 ), DF_LDO( 2 ), DF_LDO( 3 ), DF_LDO( 4 ),
			DF_LDO( 5 ), DF_LDO( 6 ), DF_LDO( 7 ), DF_LDO( 8 ), DF_LDO( 9 ),
			DF_LDO( 10 )
			: [ad] "v"( ad ), DF_LDI( 0 ), DF_LDI( 1 ), DF_LDI( 2 ),
			DF_LDI( 3 ), DF_LDI( 4 ), DF_LDI( 5 ), DF_LDI( 6 ), DF_LDI( 7 ),
			DF_LDI( 8 ), DF_LDI( 9 ), DF_LDI( 10 ));
	}
	else
	{
		static_assert( B == 7, "batch of 11 (K = 3) or 7 (K = 2) inputs" );
		asm volatile( DF_LD( 0 ) DF_LD( 1 ) DF_LD( 2 ) DF_LD( 3 ) DF_LD( 4 )
			DF_LD( 5 ) DF_LD( 6 )
			: DF_LDO( 0 ), DF_LDO( 1 ), DF_LDO( 2 ), DF_LDO( 3 ), DF_LDO( 4 ),
			DF_LDO( 5 ), DF_LDO( 6 )
			: [ad] "v"( ad ), DF_LDI( 0 ), DF_LDI( 1 ), DF_LDI( 2 ),
			DF_LDI( 3 ), DF_LDI( 4 ), DF_LDI( 5 ), DF_LDI( 6 ));
	}
}

// batch starting at input I0: byte offsets of its inputs
template< int K, int NP, int B, int I0 >
__device__ __forceinline__ void df_load_at( DfBatch< B >& d, const unsigned ad )
{
#define DF_O( j ) ( 16 * df_slot< K, NP >( I0 + ( j )))
	if constexpr( B == 11 )
	{
		df_load< B, DF_O( 0 ), DF_O( 1 ), DF_O( 2 ), DF_O( 3 ), DF_O( 4 ),
			DF_O( 5 ), DF_O( 6 ), DF_O( 7 ), DF_O( 8 ), DF_O( 9 ),
			DF_O( 10 ) >( d, ad );
	}
	else
	{
		df_load< B, DF_O( 0 ), DF_O( 1 ), DF_O( 2 ), DF_O( 3 ), DF_O( 4 ),
			DF_O( 5 ), DF_O( 6 ) >( d, ad );
	}
#undef DF_O
}

// the batch's data has landed once at most `LEFT` younger LDS reads are
// outstanding (LDS returns in order); naming the registers keeps hipcc from
// touching them before
template< int B, int LEFT >
__device__ __forceinline__ void df_wait( DfBatch< B >& d )
{
	static_assert( LEFT == 0 || LEFT == 7 || LEFT == 11, "wait count" );

	if constexpr( B == 11 )
	{
		if constexpr( LEFT == 0 )
		{
			asm volatile( AVIRHIP_WAITCNT_LGKM( 0 )
				: "+v"( d.v[ 0 ]), "+v"( d.v[ 1 ]), "+v"( d.v[ 2 ]),
				"+v"( d.v[ 3 ]), "+v"( d.v[ 4 ]), "+v"( d.v[ 5 ]),
				"+v"( d.v[ 6 ]), "+v"( d.v[ 7 ]), "+v"( d.v[ 8 ]),
				"+v"( d.v[ 9 ]), "+v"( d.v[ 10 ]));
		}
		else
		{
			asm volatile( AVIRHIP_WAITCNT_LGKM( 11 )
				: "+v"( d.v[ 0 ]), "+v"( d.v[ 1 ]), "+v"( d.v[ 2 ]),
				"+v"( d.v[ 3 ]), "+v"( d.v[ 4 ]), "+v"( d.v[ 5 ]),
				"+v"( d.v[ 6 ]), "+v"( d.v[ 7 ]), "+v"( d.v[ 8 ]),
				"+v"( d.v[ 9 ]), "+v"( d.v[ 10 ]));
		}
	}
	else
	{
		if constexpr( LEFT == 0 )
		{
			asm volatile( AVIRHIP_WAITCNT_LGKM( 0 )
				: "+v"( d.v[ 0 ]), "+v"( d.v[ 1 ]), "+v"( d.v[ 2 ]),
				"+v"( d.v[ 3 ]), "+v"( d.v[ 4 ]), "+v"( d.v[ 5 ]),
				"+v"( d.v[ 6 ]));
		}
		else
		{
			asm volatile( AVIRHIP_WAITCNT_LGKM( 7 )
				: "+v"( d.v[ 0 ]), "+v"( d.v[ 1 ]), "+v"( d.v[ 2 ]),
				"+v"( d.v[ 3 ]), "+v"( d.v[ 4 ]), "+v"( d.v[ 5 ]),
				"+v"( d.v[ 6 ]));
		}
	}
}

} // namespace avirhip

#endif
