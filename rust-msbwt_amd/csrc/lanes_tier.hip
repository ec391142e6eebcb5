// Instantiations of the one-query-per-lane search kernel (lanes_kernel.hpp) for the TWO-TIER form of the sparse table (kSparse = 2).
#include "lanes_kernel.hpp"

namespace msbwt {

MSBWT_DEFINE_SPARSE_LAUNCH(launch_lanes_sparse_tier, 2)

}  // namespace msbwt
