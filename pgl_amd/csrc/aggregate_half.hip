// aggregate_half.hip -- the aggregation kernels for 16-bit storage (fp16, bf16; fp32 accumulation).  See aggregate_more.hip.
#include "aggregate_flat.hpp"

namespace pglamd {

#define PGLAMD_AGG_ARGS const void*, int64_t, const void*, int64_t, const int32_t*, const int32_t*, const int32_t*, const int64_t*, \
                        int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, const float*, const float*, int, void*, void*, size_t, hipStream_t
template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);
template int32_t aggregate_typed<__hip_bfloat16>(PGLAMD_AGG_ARGS);

}  // namespace pglamd
