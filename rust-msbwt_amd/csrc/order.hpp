// Batch order keys (order.hip): the order in which a batch of k-mers walks the index most cheaply.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace msbwt {
void order_keys_host(const uint8_t *kmers, uint32_t k, uint64_t n, uint64_t *keys);
hipError_t launch_order_keys(const uint8_t *d_kmers, uint32_t k, uint64_t n, uint64_t *d_keys, hipStream_t stream);
}  // namespace msbwt
