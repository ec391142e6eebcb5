// Instantiations of the one-query-per-lane search kernel (lanes_kernel.hpp) for the sparse table with 40-bit tags, depths 30..31 (kSparse = 4).
#include "lanes_kernel.hpp"

namespace msbwt {

MSBWT_DEFINE_SPARSE_LAUNCH(launch_lanes_sparse_xwide, 4)

}  // namespace msbwt
