// planner.h -- internal: host-side planner of the AVIR / LANCIR front-end
// mirror. Produces plan descriptions (include/avirhip.h) whose float tables are
// bit-identical to the ones the reference's planner builds (avir.h:5128-6270,
// lancir.h:840-1381). Pure host code: double precision, libm, compiled by g++
// with -ffp-contract=off. No HIP dependency.
#pragma once
#include <stdint.h>
#include <memory>
#include <vector>
#include "../../include/avirhip.h"

namespace avirhip {

// Owns every array a description points to.
struct DescStore
{
	std::vector< std::unique_ptr< std::vector< float > > > f;
	std::vector< std::unique_ptr< std::vector< double > > > fd;
	std::vector< std::unique_ptr< std::vector< int32_t > > > i;
	std::vector< std::unique_ptr< std::vector< avirhip_rpos > > > r;
	std::vector< std::unique_ptr< std::vector< avirhip_step > > > s;
	std::vector< std::unique_ptr< std::vector< avirhip_lancir_pos > > > lp;
	avirhip_plan_desc d;
	avirhip_lancir_desc ld;
};

namespace pl32 { class AvirPlannerImpl; } // float pipeline (fpclass_def<float>, fpclass_float4)
namespace pl64 { class AvirPlannerImpl; } // double pipeline (fpclass_def<double>)

// Mirror of avir::CImageResizer's planning half (ctor avir.h:4630-4639).
class AvirPlanner
{
public:
	AvirPlanner( int res_bit_depth, int src_bit_depth,
		const avirhip_params& params );
	~AvirPlanner();

	// Runs the planning half of resizeImage() (avir.h:4699-4954, 5029-5045).
	// Returns nullptr (and sets the error string) on an unsupported request.
	DescStore* build( int src_w, int src_h, int src_scanline_size, int new_w,
		int new_h, int el_count_io, double k, const avirhip_vars& vars,
		int in_type, int out_type, int fppack = 1, bool f64 = false );

private:
	pl32::AvirPlannerImpl* impl;
	pl64::AvirPlannerImpl* impl64; // (made when the first double plan is asked for)
};

// Mirror of avir::CLancIR's planning half (lancir.h:430-483).
DescStore* lancir_build( int src_w, int src_h, int new_w, int new_h,
	int el_count, const avirhip_lancir_params& params, int in_type,
	int out_type );

void set_error( const char* fmt, ... );

} // namespace avirhip
