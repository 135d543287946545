// aggregate_f64.hip -- the aggregation kernels for fp64.  See aggregate_more.hip.
#include "aggregate_flat.hpp"

namespace pglamd {

#define PGLAMD_AGG_ARGS const void*, int64_t, const void*, int64_t, const int32_t*, const int32_t*, const int32_t*, const int64_t*, \
                        int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, const float*, const float*, int, void*, void*, size_t, hipStream_t
template int32_t aggregate_typed<double>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
