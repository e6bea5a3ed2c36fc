// block_scalers.hpp -- tables of the 8x8 -> NxN spatial block scalers (see block_scalers.cpp)
#pragma once
#include <cstdint>

namespace ifhip {

struct BlockScaler {          // one size N: output i takes source rows/cols first[i]..last[i] with int8 weights
    uint32_t n;
    int8_t w[7][8];
    uint8_t first[7], last[7];
    uint8_t log2_div[7];      // the weights of output i sum to 1 << log2_div[i]
};

struct BlockScalerTables {
    BlockScaler scaler[8];    // index 1..7
    uint16_t srgb_to_linear[256];   // 12-bit linear
    uint8_t linear_to_srgb[4096];
};

const BlockScalerTables* block_scaler_tables();   // nullptr if the generator failed (never expected)

}  // namespace ifhip
