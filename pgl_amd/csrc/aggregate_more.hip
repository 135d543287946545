// aggregate_more.hip -- the aggregation kernels for the integer storage types (int32, int64).  One translation unit per pair
// of types (aggregate.hip fp32 + the C ABI, aggregate_f64.hip, this one, aggregate_half.hip) keeps the build's critical path
// at one type's worth of instantiations.  Same templates: aggregate_flat.hpp.
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<int32_t>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<int64_t>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
