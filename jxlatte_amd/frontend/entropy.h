// Entropy decoding of JPEG XL codestreams: clustered context map, hybrid-uint configuration, ANS (alias table) and
// prefix (Brotli-style canonical Huffman) symbol codes, LZ77 window. Host-side front-end (row f2); counterpart of
// J/entropy/{EntropyStream,ANSSymbolDistribution,PrefixSymbolDistribution,VLCTable,HybridIntegerConfig}.java.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "bits.h"

namespace jxf {

struct HybridUint {
    int split_exp = 0, msb = 0, lsb = 0;
    void read(BitReader& br, int log_alpha);
};

// one clustered distribution: either an ANS alias table or a prefix code
struct SymbolCode {
    HybridUint cfg;
    // ANS
    int log_bucket = 0;
    std::vector<uint16_t> freq;                    // per symbol, sum 4096
    std::vector<uint16_t> cutoff, alias_sym;       // per bucket
    std::vector<int32_t> offset;                   // per bucket
    // prefix: flat lookup of `bits` peeked bits -> (symbol, length)
    int pbits = 0;
    std::vector<uint32_t> ptable;                  // (length << 16) | symbol
    int single_symbol = -1;                        // prefix code with one symbol: consumes no bits
};

// the part of an entropy stream shared by every reader of the same histograms (immutable once parsed)
struct EntropyCode {
    bool lz77 = false;
    uint32_t lz_min_symbol = 0, lz_min_length = 0;
    HybridUint lz_len_cfg;
    std::vector<uint8_t> cluster;  // context -> distribution (last entry = LZ77 distance context when lz77)
    std::vector<SymbolCode> codes;
    bool prefix = false;
    int log_alpha = 0;
    void read(BitReader& br, int num_ctx, bool allow_lz77 = true);
};

// reads the context map of an entropy stream (also used on its own by HFBlockContext): returns the cluster count
int read_cluster_map(BitReader& br, std::vector<uint8_t>& map, int max_clusters);

// per-reader state: ANS state + LZ77 window
class EntropyDecoder {
  public:
    EntropyDecoder() = default;
    explicit EntropyDecoder(std::shared_ptr<const EntropyCode> code) { reset(std::move(code)); }
    void reset(std::shared_ptr<const EntropyCode> code);
    uint32_t read(BitReader& br, int ctx, uint32_t dist_multiplier = 0);
    // ANS streams must end in the initial state 0x130000 (EntropyStream.validateFinalState)
    bool final_state_ok() const { return !has_state_ || state_ == 0x130000u; }
    void check_final(const char* what) const {
        if (!final_state_ok()) throw BitstreamError(std::string("Illegal final ANS state: ") + what);
    }
    const EntropyCode& code() const { return *code_; }

  private:
    uint32_t symbol(BitReader& br, const SymbolCode& c);
    uint32_t hybrid(BitReader& br, const HybridUint& h, uint32_t token);
    std::shared_ptr<const EntropyCode> code_;
    uint32_t state_ = 0;
    bool has_state_ = false;
    std::vector<uint32_t> window_;
    uint32_t num_to_copy_ = 0, copy_pos_ = 0, num_decoded_ = 0;
};

}  // namespace jxf
