// Instantiations of the one-query-per-lane search kernel (lanes_kernel.hpp) for the sparse table with 32-bit tags, depths 25..29 (kSparse = 3).
#include "lanes_kernel.hpp"

namespace msbwt {

MSBWT_DEFINE_SPARSE_LAUNCH(launch_lanes_sparse_wide, 3)

}  // namespace msbwt
