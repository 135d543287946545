// aggregate_more.hip -- the flat aggregation kernels for the integer and 16-bit storage types (int32, int64, fp16, bf16),
// compiled next to aggregate.hip (fp32, fp64) to halve the build's critical path.  Same templates: aggregate_flat.hpp.
#include "aggregate_flat.hpp"

namespace pglamd {

#define PGLAMD_AGG_ARGS const void*, int64_t, const void*, int64_t, const int32_t*, const int32_t*, const int32_t*, const int64_t*, \
                        int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, const float*, const float*, int, void*, void*, size_t, hipStream_t
template int32_t aggregate_typed<int32_t>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<int64_t>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<__hip_bfloat16>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
