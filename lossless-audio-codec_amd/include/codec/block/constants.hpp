// Format constants of the LAC block layer under the names the reference exports (src/codec/block/constants.hpp:6-15);
// the values are fixed by the .lac format (docs/format.md) and restated here from lacx_types.h / lacx.h.
#pragma once
#include <cstdint>

#include "lacx.h"

namespace Block {

// block geometry: 2^14 frames per block, 2^8 the smallest canonical non-final block, 2^5 the smallest partition
inline constexpr std::uint32_t MAX_BLOCK_SIZE = LACX_MAX_BLOCK, MIN_CANONICAL_NON_FINAL_BLOCK_SIZE = 1u << 8,
                               MIN_PARTITION_SIZE = 1u << 5;
// zero-run tokens: runs of at least four zeros, run length minus four Rice-coded with k = 2
inline constexpr std::uint32_t ZERO_RUN_MIN_LENGTH = 4u, ZERO_RUN_LENGTH_K = 2u;
// residual control byte: bit 7 = partitioned, bit 4 reserved, low nibble = partition order (at most 8)
inline constexpr std::uint8_t PARTITION_FLAG = 1u << 7, RESIDUAL_RESERVED_MASK = 1u << 4, PARTITION_ORDER_MASK = 0xFu,
                              PARTITION_ORDER_SHIFT = 0u, MAX_PARTITION_ORDER = 8u;

static_assert(MAX_BLOCK_SIZE == 16384u && (MAX_BLOCK_SIZE >> MAX_PARTITION_ORDER) >= MIN_PARTITION_SIZE, "format constants");

}  // namespace Block
