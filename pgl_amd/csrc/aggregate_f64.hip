// aggregate_f64.hip -- the aggregation kernels for fp64.  See aggregate_more.hip.
#include "aggregate_flat.hpp"

namespace pglamd {

template int32_t aggregate_typed<double>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
