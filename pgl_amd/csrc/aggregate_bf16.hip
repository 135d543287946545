// aggregate_bf16.hip -- the aggregation kernels for bf16 storage (fp32 accumulation).  See aggregate_more.hip / aggregate_half.hip.
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<__hip_bfloat16>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
