// arith.hpp -- ArithmeticCoder (coders/ArithmeticCoder.hpp:35-177) as literal coder of the lzss token stream
#pragma once
#include "common.hpp"
#include "huffman_host.hpp"

namespace tdc {

struct ArithModel {
    u32 C[256];            // normalised cumulative counts
    u64 min_range;         // C[254]
    u32 literal_count;     // cumulative (un-normalised) count up to byte 254
    u32 tot;               // C[255]
};
// builds the model from the literal histogram and appends the code book to `hw`; false if the reference would divide by 0
bool arith_build_model(const u32 hist[256], ArithModel* m, HostBitWriter& hw);

struct ArithPlan {
    const u32* litidx = nullptr;   // per text position: ordinal among the literals (valid at literal positions)
    const u8* amark = nullptr;     // per literal: 1 = a 64-bit word is flushed in front of it
    const u64* fval = nullptr;     // per literal: that word
    u32 lc_index = 0;              // literal after which `lower` + all-ones are written
    u64 pp_lb = 0;                 // that `lower`
    size_t nlit = 0;
    bool sequential_fallback = false;
};
// device arrays come from the arena (the caller brackets the call with mark / release)
void arith_prepare(Ctx& c, const u8* text, size_t n, const u32* owner, const ArithModel& m, ArithPlan* plan);

}  // namespace tdc
