// plan.h -- internal (not part of the C ABI): the lowered, device-resident form
// of a plan and the launch entry points of the kernel families.
//
// Lowering turns the reference-shaped step list of include/avirhip.h into a
// chain of "ops" per axis. Every op produces a materialised array of pixels
// along the axis; it reads its input through a *view* that reproduces, by
// index algebra, what the reference obtains by writing replicated/zeroed
// prefix and suffix pixels into its flip-flop buffers:
//
//   VIEW_CLAMP  in[clamp(i, 0, in_len-1)]          (prepareInBuf, avir.h:3227)
//   VIEW_ZS     zero-stuffed 2x view of a clamped input; the consumer reads
//               only the even (non-zero) slots 2m -> in[clamp(m)], and slots
//               past the last replicated pixel read 0 (doUpsample no-filter
//               branch, avir.h:3260-3402)
//   VIEW_RAW    in[i + prefix] of a materialised filtered-upsample buffer
//
#pragma once
#include "waits.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include "../../include/avirhip.h"

// The (int) cast of the reference's x86-64 build (cvttss2si), which avir::round
// (avir.h:130-135) is made of: a value the 32-bit integer cannot hold -- NaN
// included -- converts to INT_MIN, the "integer indefinite" (C++ leaves it
// undefined; the parity target is that build). gfx950's v_cvt_i32_f32
// saturates instead, so float images with HDR-sized or non-finite samples would
// give other integer pixels: +1e10 -> 0 and -1e10 -> the maximum in the
// reference (round() returns -2^31 / +2^31, then the clamp).
__host__ __device__ __forceinline__ int avirhip_x86_cvtt( const float f )
{
	return( fabsf( f ) < 2147483648.0f ? (int) f : ( -2147483647 - 1 ));
}

// ... for the output stages that drop avir::round's negative branch (whatever it
// returns for -2^31 < d < 0 the clamp turns into 0): `t` is the clamped result
// computed with the saturating cast from round()'s argument `a`; beyond the
// int range the reference's round() is -2^31 for positive and +2^31 for
// negative arguments, i.e. 0 and the maximum after the clamp.
__device__ __forceinline__ float avirhip_x86_round_fix( const float a,
	const float t, const float pk_out )
{
	return( fabsf( a ) >= 2147483648.0f ? ( a < 0.0f ? pk_out : 0.0f ) : t );
}

// internal: an inner plan was run without a float result buffer and its path
// would have written one (exec_device allocates it and runs again)
#define AVIRHIP_NEED_DST 1001
// internal: an inner plan was told to read its owner's image as it is
// (lraw.on) and the kernel that would have done so refused the call: the owner
// runs the pack pass and calls again with the float copy
#define AVIRHIP_NEED_SRC 1002

namespace avirhip {

enum { OP_FIR = 0, OP_GATHER = 1, OP_UPF = 2 };
enum { VIEW_CLAMP = 0, VIEW_ZS = 1, VIEW_RAW = 2 };

// One lowered op. All pointers are device pointers owned by the plan.
struct LOp
{
	int type;
	int view;
	int in_len;     // logical input length (clamp range)
	int in_prefix;  // VIEW_RAW: pixels stored before logical index 0
	int zs_mmax;    // VIEW_ZS: source index m > zs_mmax reads 0.0f
	int out_len;    // logical output length
	int out_prefix; // OP_UPF: OutPrefix pixels stored before index 0
	int out_total;  // out_prefix + out_len + suffix (materialised length)

	// OP_FIR: out[n] = f[0]*in[c] + sum_i f[i]*(in[c+i] + in[c-i]),
	// c = rf*(n-e)  (doFilter, avir.h:3748-3866)
	int rf, lat, e;
	float* d_flt; // FIR: &Flt[FltLatency] (lat+1 taps); UPF: whole Flt
	std::vector< float > h_flt;
	// (double pipeline, plan -> f64: the same tables in double; the float ones
	// stay empty)
	double* d_flt64;
	std::vector< double > h_flt64;

	// OP_GATHER: out[j] = 0 + sum_{t<ntaps[j]} coef[j][t]*in[start[j]+t]
	// (doResize / doResize2, avir.h:3884-4328). coef is pre-expanded per
	// output position: order-0 taps, or ftp[i] + ftp2[i]*x in float.
	int maxtaps;
	int* d_start;
	int* d_ntaps;
	float* d_coef; // [out_len][maxtaps]
	std::vector< int > h_start, h_ntaps;
	std::vector< float > h_coef;
	double* d_coef64;
	std::vector< double > h_coef64;

	// OP_UPF (filtered upsample, avir.h:3404-3733)
	int flen, up_inprefix, up_R, sdc_len, pdc_len, pdc_d0;
	float* d_sdc;
	float* d_pdc;
	double* d_sdc64;
	double* d_pdc64;
};

struct LAxis
{
	std::vector< LOp > ops;
	int src_len, dst_len;
};

// Image element access: address(scan, idx, c) =
//   base + scan*scan_stride + (idx + prefix)*idx_stride + c
struct Surf
{
	float* base;
	long scan_stride;
	long idx_stride;
	int prefix;
};

struct LancirAxisDev
{
	int kernel_len, src_len, dst_len, n_filters;
	int* d_start;   // [dst_len]: first source pixel, un-padded coordinates
	int* d_fidx;    // [dst_len]
	float* d_flt;   // [n_filters][kernel_len]
	std::vector< int > h_start, h_fidx;
	std::vector< float > h_flt;
};

} // namespace avirhip

struct avirhip_plan
{
	int is_lancir;
	int device;
	int src_w, src_h, src_stride, new_w, new_h, new_stride;
	int ch;    // channels the kernels execute with (4 when 1-3 channel pixels
	           // are padded so that the RGBA fast paths apply)
	int io_ch; // channels of the caller's buffers (ElCountIO)
	int in_type, out_type;
	double tr_mul, pk_out;
	int gamma, alpha_index; // sRGB gamma stages (avir.h:2841-2930, 2982-3068)
	int dither;             // AVIRHIP_DITHER_* (integer outputs only)
	int fp4;                // plan of an fpclass_float4 object (no in-place float output)
	int f64;                // double pipeline (fpclass_def<double>): generic64.hip runs it
	double* packed64;       // its copy of the source, its result, its intermediates
	double* resbuf64;
	std::vector< double* > hbuf64, vbuf64;
	float* errd_line;       // error-diffusion rows handed between row blocks
	float* d_srgb_tbl;      // 256-entry uint8 linearisation table (gamma plans)
	float* d_gthr;          // uint8 gamma output stage as 2 x 256 thresholds
	avirhip::LAxis h, v;
	// LANCIR
	avirhip::LancirAxisDev lv, lh;
	float l_out_mul, l_clamp;
	int l_unity;
	// LANCIR with integer / scaled RGBA I/O: a float RGBA, unity-gain plan of
	// the same geometry whose fast kernels run between a pack pass and the
	// output stage (nullptr: this plan runs its own kernels)
	avirhip_plan* inner;
	int l_order; // LANCIR: channel count whose summation order the kernels use
	// (set on an INNER plan by its owner around a call: the owner's output
	// stage, fused into the last pass when that is a pass kernel)
	struct { int on, done, type, ch, unity; float out_mul, clampv; void* dst;
		long stride; } lfuse;
	struct { int on, type, ch; const void* ptr; long stride; } lraw;

	int path;       // forced path (0 = auto)
	int variant;    // AVIRHIP_VARIANT_* bits (0 = automatic kernel forms)
	int fused_ok;   // tiled kernels: bit 0 = two-pass (path 2), bit 1 = fused (3)
	int auto_path;  // path taken when `path` == 0
	void* fused;    // tiled-kernel private data
	void* up2;      // exact-2x marching kernel private data (fused_ok bit 2)
	void* lanc2;    // LANCIR exact-2x kernel private data (fused_ok bit 2)
	void* gpass;    // general-ratio pass kernels, path 5 (fused_ok bit 3)
	void* tile64;   // double pipeline: LDS-tiled two-pass executor (tile64.hip)

	// scratch (device), lazily sized
	std::vector< void* > allocs;
	size_t alloc_bytes; // device bytes this plan holds (tables + scratch)
	float* packed;  // source converted to float (non-f32 input)
	float* resbuf;  // float result before the integer / f64 epilogue
	float* lres;    // LANCIR: float result rows before the output stage
	std::vector< float* > hbuf, vbuf; // per-op outputs
	// avirhip_resize_sharded: replicas of this plan on other devices, and the
	// per-replica band / source buffers
	std::vector< avirhip_plan* > replicas;
	// same-device copies used when a call finds this plan's scratch busy
	// (concurrent resizeImage() calls on one object); `is_spare` marks them
	std::vector< avirhip_plan* > spares;
	int is_spare;
	void* shard_band; size_t shard_band_bytes;
	int shard_ldev; // replica: the device id its owner's caller named
	void* shard_src; size_t shard_src_bytes;
	// A plan owns scratch buffers (packed source, float result, FltBuf, op
	// outputs, staging): calls that use them are serialised -- host side by
	// the mutex, device side by making every call's stream wait for the
	// previous call's completion event.
	std::mutex exec_mtx;
	std::mutex shard_mtx; // avirhip_resize_sharded: replicas, band buffers
	std::mutex spare_mtx; // the list of spares
	hipEvent_t last_done;
	void* last_stream; // stream of the last call that used the scratch buffers
	bool last_used;    // (there was one)
	std::thread::id last_tid; // its thread (hipStreamPerThread: one handle, a stream per thread)
	bool last_recorded; // last_done was recorded when that call ended (per-thread streams)
	// host-pointer calls: copy streams and events of the band pipeline
	void* pipe_in; void* pipe_out;
	std::vector< hipEvent_t > pipe_ev;
	void* stage_src; // host-pointer staging
	void* stage_dst;
	size_t stage_src_bytes, stage_dst_bytes;
	// avirhip_resize_window on the marching kernels (k_up2, k_lanc2): the source
	// pointer handed to them is a VIRTUAL frame base -- window - win_first rows
	// -- and only rows [win_first, win_first + win_rows) exist behind it: the
	// kernels clamp their row indices to that range instead of [0, src_h)
	// (set by exec_any under the plan's lock for the duration of one call;
	// win_rows == 0: the whole frame)
	int win_first, win_rows;
};

namespace avirhip {

void set_error( const char* fmt, ... );
void clear_error();
#define AVIRHIP_HIPCHECK( expr ) do { hipError_t e_ = ( expr ); \
	if( e_ != hipSuccess ) { avirhip::set_error( "%s: %s (%s:%d)", #expr, \
		hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
		return( e_ == hipErrorOutOfMemory ? AVIRHIP_ENOMEM : AVIRHIP_EHIP ); } } \
	while( 0 )

// The boundary is exception-tight (include/avirhip.h, "Conventions"): every
// `extern "C"` entry point is a function-try-block that ends in AVIRHIP_CATCH.
// guard_fail() runs inside the handler, classifies the exception in flight
// (std::bad_alloc -> AVIRHIP_ENOMEM, anything else -> AVIRHIP_EINTERNAL) and
// leaves the message for avirhip_last_error(). The reference's own contract is
// "no exceptions besides bad_alloc" (avir.h:564-827 CBuffer::alloc); the C++
// front end (include/avir_hip) turns AVIRHIP_ENOMEM back into std::bad_alloc.
int guard_fail( const char* fn ) noexcept;
#define AVIRHIP_CATCH( fn ) catch( ... ) { return( avirhip::guard_fail( #fn )); }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a loaded function on a
// device, not to a launch: it is raised when a launch needs more dynamic LDS
// than any earlier one asked for, not once per frame (a driver call on the hot
// path otherwise). `have` is the call site's own record, one slot per device.
inline hipError_t ensure_dyn_lds( const void* fn, const size_t bytes,
	std::atomic< int >* have )
{
	int dev = 0;
	(void) hipGetDevice( &dev );
	std::atomic< int >& h = have[ dev & 31 ];

	if( h.load( std::memory_order_relaxed ) >= (int) bytes )
	{
		return( hipSuccess );
	}

	const hipError_t e = hipFuncSetAttribute( fn,
		hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes );

	if( e == hipSuccess )
	{
		h.store( (int) bytes, std::memory_order_relaxed );
	}

	return( e );
}
#define AVIRHIP_DYN_LDS( fn, bytes ) ( [&]() -> hipError_t { \
	static std::atomic< int > have_[ 32 ]; \
	return( avirhip::ensure_dyn_lds( (const void*) ( fn ), ( bytes ), have_ )); }() )

// a * b * c * d without wrapping; false when the product does not fit size_t
bool mul_fits( size_t a, size_t b, size_t c, size_t d, size_t* out = nullptr );
// Geometry every entry point checks before it allocates: row lengths in
// elements fit `int` (the reference computes them in int, avir.h:4786-4794,
// lancir.h:409-411), image sizes in bytes fit size_t. Sets the error string.
bool geometry_ok( const char* fn, int src_w, int src_h, long src_stride,
	int new_w, int new_h, long new_stride, int ch, int in_type, int out_type );

// An owning handle for a plan under construction: a throw between new_plan()
// and the hand-over destroys it (device tables included).
struct PlanHold
{
	avirhip_plan* p;
	explicit PlanHold( avirhip_plan* q ) : p( q ) {}
	~PlanHold();
	avirhip_plan* release() { avirhip_plan* q = p; p = nullptr; return( q ); }
};

size_t dtype_size( int t );
int finalize_avir_plan( avirhip_plan* p ); // api.cpp
// device bytes a plan holds, with its inner plan, same-device spares and
// other-device replicas (the plan caches' byte bound)
size_t plan_device_bytes( avirhip_plan* p );

// generic.hip
// `ch` = channels in `src`, `ech` >= ch = channels written (zero padded)
int launch_pack( const void* src, int in_type, float* dst, int w, int h,
	int ch, int ech, long src_stride, hipStream_t st );
int launch_pack_gamma( const void* src, int in_type, float* dst, int w, int h,
	int ch, int ech, long src_stride, int alpha_index, const float* tbl,
	hipStream_t st );
void srgb_u8_table( float* tbl );
int launch_op( const LOp& op, int ch, bool x_is_idx, const Surf& in,
	const Surf& out, int scan0, int scan1, int idx0, int idx1,
	hipStream_t st );
int launch_epilogue( const float* res, void* dst, int out_type, long n,
	double tr_mul, double pk_out, int gamma, int ch, int ech,
	int alpha_index, hipStream_t st, const float* gthr = nullptr,
	bool rne = false );
bool gamma_u8_thresholds( float ogm, int use_tr, float trm, float trmi,
	float pk, float* thr );
// the input range [ia, ib] an op reads for outputs [a, b] (api.cpp)
void need_range( const LOp& op, int a, int b, int& ia, int& ib );
// the source rows [*first, *last] the outputs [row0, row1) of a lowered axis
// read; the same from a plan description's vertical axis (host only)
void axis_src_range( const LAxis& ax, int row0, int row1, int src_len,
	int* first, int* last );
int desc_band_src_rows( const avirhip_plan_desc* d, int row0, int row1,
	int* first, int* last );

// generic64.hip: the double pipeline (plan -> f64), output rows [row0, row1)
int exec_f64( avirhip_plan* p, const void* src, void* dst, int row0, int row1,
	hipStream_t st );

// tile64.hip: the double pipeline's tiled two-pass executor. tile64_run
// returns 1 when the plan has none (filtered upsamples: generic64.hip runs it).
int tile64_prepare( avirhip_plan* p );
void tile64_release( avirhip_plan* p );
bool tile64_ok( const avirhip_plan* p );
int tile64_run( avirhip_plan* p, const void* src, int src_type, long src_ss,
	void* dst, int dst_type, long dst_ss, int row0, int row1, int ya, int yb,
	hipStream_t st );

// up64.hip: marching kernels of the double pipeline's upsizing chains
// (FIR7 -> 12-tap gather over the zero-stuffed view), used by tile64_run for
// the axes that have them; 1: the call was not taken
bool up64_axis_ok( const LAxis& ax );
int up64_run_h( const avirhip_plan* p, const void* src, int src_type,
	long src_ss, double* fltbuf, int ya, int yb, hipStream_t st );
int up64_run_v( const avirhip_plan* p, const double* fltbuf, void* dst,
	int dst_type, long dst_ss, int row0, int row1, hipStream_t st );

int launch_errd( const float* res, void* dst, int out_type, int w, int h,
	int ch, int ech, double tr_mul, double pk_out, int gamma, int alpha_index,
	float* line, hipStream_t st );
int launch_lancir_out( const avirhip_plan* p, const float* res, long rstride,
	void* dst, int nrows, hipStream_t st );
int launch_lancir_out_pad( const avirhip_plan* p, const float* res, void* dst,
	int nrows, hipStream_t st );
int launch_lancir_generic( const avirhip_plan* p, const void* src, void* dst,
	float* tmp, int row0, int row1, hipStream_t st );

// fused.hip
int fused_prepare( avirhip_plan* p );
void fused_release( avirhip_plan* p );
// Runs path `mode` (2 = two-pass tiled, 3 = fused). Returns 1 when the call
// cannot take a tiled path (e.g. unaligned rows) and the generic path should.
// `src_type` / `src_ch`: element type and channel count of the array at `src`
// (AVIRHIP_F32 and 4: a float RGBA image; AVIRHIP_U8 / AVIRHIP_U16 with 1-4
// channels: the caller's integer image, converted and padded by the tile
// loader itself -- no pack pass). `src_stride` in elements of that type.
int fused_run( avirhip_plan* p, int mode, const void* src, int src_type,
	int src_ch, long src_stride, float* dst, int row0, int row1,
	hipStream_t st, void* iout = nullptr );
bool fused_stores_int( const avirhip_plan* p, int mode );
bool fused_takes_raw( const avirhip_plan* p, int mode );

// dn.hip: integer-ratio downsizing passes used by the two-pass tiled path
int dn_prepare( avirhip_plan* p, void** out );
void dn_release( void* d );
bool dn_has_h( const void* d );
bool dn_has_v( const void* d );
int dn_run_h( void* d, const void* src, int src_type, int src_ch, long src_ss,
	float* flt, long flt_ss, int a, int b, hipStream_t st );
struct GPOut; // gpass_dev.h: the integer output stage fused into a store
int dn_run_v( void* d, const float* flt, long flt_ss, int width, float* dst,
	int row0, int row1, hipStream_t st, const GPOut* out = nullptr );
// dnf.hip: both axes of such a plan in one marching launch (no FltBuf);
// returns 1 when the call cannot take it
int dn_run_hv( void* d, const float* src, long src_ss, float* dst, int row0,
	int row1, hipStream_t st, const GPOut* out = nullptr );

// gpass.hip: general-ratio pass kernels (path 5), AVIR and LANCIR RGBA float
int gpass_prepare( avirhip_plan* p );
void gpass_release( avirhip_plan* p );
bool gpass_ok( const avirhip_plan* p );
bool gpass_preferred( const avirhip_plan* p );
bool gpass_takes_raw( const avirhip_plan* p );
bool gpass_lancir_takes_raw( const avirhip_plan* p, const void* raw, int type,
	int ch, long stride );
int gpass_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st, const void* raw = nullptr,
	int raw_type = 0, int raw_ch = 0, long raw_stride = 0,
	void* iout = nullptr );
bool fused_dn_both( const avirhip_plan* p );

// up2.hip: specialised exact-2x RGBA kernel (path 4)
int up2_prepare( avirhip_plan* p );
void up2_release( avirhip_plan* p );
int up2_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st, void* iout = nullptr,
	const void* raw = nullptr, long raw_stride = 0 );
bool up2_stores_io( const avirhip_plan* p );
// whether up2_run / lanc2_run take this float RGBA call for certain (no refusal,
// no fall-back to kernels that know nothing of a source window)
bool up2_takes_window( const avirhip_plan* p, const void* src, const void* dst );
bool lanc2_takes_window( const avirhip_plan* p, const void* src, const void* dst );

// lanc2.hip: LANCIR exact-2x RGBA float kernel (path 4 of LANCIR plans)
int lanc2_prepare( avirhip_plan* p );
void lanc2_release( avirhip_plan* p );
int lanc2_run( avirhip_plan* p, const float* src, float* dst, int row0,
	int row1, hipStream_t st );
bool lanc2_takes_raw( const avirhip_plan* q, const void* raw, int type, int ch,
	long stride );

} // namespace avirhip
