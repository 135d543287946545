// aggregate_half.hip -- the aggregation kernels for fp16 storage (fp32 accumulation).  See aggregate_more.hip.
// (bf16 lives in aggregate_bf16.hip: with the wire-mirror variants the two types in one translation unit took 3.7 minutes to compile,
//  the longest pole of the parallel build.)
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
