// stages_host.hpp -- host-only stages that are compiled by g++ (no HIP headers): declarations shared with stages.hpp
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace tdc {
// LZ78 parse (compressors/LZ78Compressor.hpp:97-131): ids[k] = id of the longest dictionary phrase at the start of factor k (0: none),
// chars[k] = the byte behind it; returns the number of factors.  *leftover_is_high: the text ended inside a phrase whose last byte is >= 0x80.
size_t lz78_parse_host(const uint8_t* in, size_t n, std::vector<uint32_t>& ids, std::vector<uint8_t>& chars, bool* leftover_is_high);
}
