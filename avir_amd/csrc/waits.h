// waits.h -- the ONE place every hand-counted wait of the kernels goes through.
//
// The marching / pass kernels overlap their LDS-DMA feeds with arithmetic by
// COUNTING: "row r has landed when at most N younger vector-memory loads are
// outstanding" (s_waitcnt vmcnt(N); loads retire in issue order), "batch b of
// LDS reads has returned when at most N younger ones are" (lgkmcnt(N)), and a
// ring slot may be handed to the next LDS-DMA only after the reads of the row
// it held have been WAITED for (NOTEBOOK 9.13: k_lf carried that race for a
// round). A human derived each literal; DESIGN.md section 4a lists them with
// the loads each one counts and why the slot it guards is safe.
//
// `make paranoid` (-DAVIRHIP_PARANOID_WAITS) builds ../lib/libavirhip_paranoid.so
// in which
//   * every counted wait is a full drain of both counters,
//   * every LDS-DMA instruction is preceded by a full drain (the wave's own LDS
//     reads of the slot it takes have returned, every older DMA has landed) --
//     AVIRHIP_HANDOVER, inside the DMA primitives themselves, so no handover
//     point can be forgotten,
//   * the workgroup-synchronous kernels drain before each s_barrier.
// Same arithmetic, same order, no overlap: slower, and race-free by
// construction as far as a wave's own counters go (rings shared BETWEEN waves
// change hands at s_barriers in both builds). tools/stress_geom.py --against
// loads both libraries in one process and diffs them per kernel family on
// random large geometries; tests/test_gpu_parity.py runs a slice of it.
#ifndef AVIRHIP_WAITS_H
#define AVIRHIP_WAITS_H

#if defined( AVIRHIP_PARANOID_ONLY_WAITS )
// (bisection aids: only the counted waits / only the handover drains)
#define AVIRHIP_WAITCNT_VM( n ) "s_waitcnt vmcnt(0) lgkmcnt(0)"
#define AVIRHIP_WAITCNT_LGKM( n ) "s_waitcnt vmcnt(0) lgkmcnt(0)"
#define AVIRHIP_HANDOVER ""
#define AVIRHIP_PARANOID 0
#elif defined( AVIRHIP_PARANOID_ONLY_HANDOVER )
#define AVIRHIP_WAITCNT_VM( n ) "s_waitcnt vmcnt(" #n ")"
#define AVIRHIP_WAITCNT_LGKM( n ) "s_waitcnt lgkmcnt(" #n ")"
#define AVIRHIP_HANDOVER "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define AVIRHIP_PARANOID 0
#elif defined( AVIRHIP_PARANOID_ONLY_HANDOVER_LGKM )
#define AVIRHIP_WAITCNT_VM( n ) "s_waitcnt vmcnt(" #n ")"
#define AVIRHIP_WAITCNT_LGKM( n ) "s_waitcnt lgkmcnt(" #n ")"
#define AVIRHIP_HANDOVER "s_waitcnt lgkmcnt(0)\n\t"
#define AVIRHIP_PARANOID 0
#elif defined( AVIRHIP_PARANOID_WAITS )
#define AVIRHIP_WAITCNT_VM( n ) "s_waitcnt vmcnt(0) lgkmcnt(0)"
#define AVIRHIP_WAITCNT_LGKM( n ) "s_waitcnt vmcnt(0) lgkmcnt(0)"
#define AVIRHIP_HANDOVER "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define AVIRHIP_PARANOID 1
#else
// (string forms: for waits that sit inside a larger asm statement or carry
// register operands)
#define AVIRHIP_WAITCNT_VM( n ) "s_waitcnt vmcnt(" #n ")"
#define AVIRHIP_WAITCNT_LGKM( n ) "s_waitcnt lgkmcnt(" #n ")"
#define AVIRHIP_HANDOVER ""
#define AVIRHIP_PARANOID 0
#endif

// at most `n` (a literal) vector-memory LOADS younger than the awaited one
#define AVIRHIP_WAIT_VM( n ) asm volatile( AVIRHIP_WAITCNT_VM( n ) ::: "memory" )
// at most `n` younger LDS / scalar-memory operations
#define AVIRHIP_WAIT_LGKM( n ) asm volatile( AVIRHIP_WAITCNT_LGKM( n ) ::: "memory" )

// Before an s_barrier of a workgroup-synchronous kernel whose DMA is a compiler
// builtin (the compiler's own wait-count insertion covers it): nothing in the
// product build, a full drain in the paranoid one.
#if AVIRHIP_PARANOID
#define AVIRHIP_BARRIER_DRAIN() asm volatile( "s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory" )
#else
#define AVIRHIP_BARRIER_DRAIN() do {} while( 0 )
#endif

#endif
