// aggregate_half.hip -- the aggregation kernels for fp16 storage (fp32 accumulation).  See aggregate_more.hip.
// (bf16 lives in aggregate_bf16.hip: one storage type per translation unit keeps the parallel build's longest pole short.)
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
