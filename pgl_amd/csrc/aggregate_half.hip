// aggregate_half.hip -- the aggregation kernels for 16-bit storage (fp16, bf16; fp32 accumulation).  See aggregate_more.hip.
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<__hip_bfloat16>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
