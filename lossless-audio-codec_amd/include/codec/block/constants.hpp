// Mirror of the reference's src/codec/block/constants.hpp:6-15 (values restated; they are part of the
// .lac format, docs/format.md).
#pragma once
#include <cstdint>

namespace Block {
constexpr uint32_t MAX_BLOCK_SIZE = 16384;
constexpr uint32_t MIN_CANONICAL_NON_FINAL_BLOCK_SIZE = 256;
constexpr uint32_t ZERO_RUN_MIN_LENGTH = 4;
constexpr uint32_t ZERO_RUN_LENGTH_K = 2;
constexpr uint32_t MIN_PARTITION_SIZE = 32;
constexpr uint8_t MAX_PARTITION_ORDER = 8;
constexpr uint8_t PARTITION_FLAG = 0x80;
constexpr uint8_t RESIDUAL_RESERVED_MASK = 0x10;
constexpr uint8_t PARTITION_ORDER_SHIFT = 0;
constexpr uint8_t PARTITION_ORDER_MASK = 0x0F;
}  // namespace Block
