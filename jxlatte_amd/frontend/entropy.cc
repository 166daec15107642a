#include "entropy.h"

#include <algorithm>
#include <cstring>

namespace jxf {

void HybridUint::read(BitReader& br, int log_alpha) {  // HybridIntegerConfig.java:18-30
    split_exp = (int)br.bits(ceil_log1p(log_alpha));
    msb = lsb = 0;
    if (split_exp == log_alpha) return;
    msb = (int)br.bits(ceil_log1p(split_exp));
    if (msb > split_exp) throw BitstreamError("msbInToken is too large");
    lsb = (int)br.bits(ceil_log1p(split_exp - msb));
    if (msb + lsb > split_exp) throw BitstreamError("msbInToken + lsbInToken is too large");
}

namespace {

// canonical prefix code -> flat table indexed by `bits` LSB-first peeked bits. lengths[i] is the code length of
// symbols[i] in the order the code words are assigned (already sorted by (length, symbol) by the callers).
void build_prefix_table(SymbolCode& c, int bits, const std::vector<int>& lengths, const std::vector<int>& symbols) {
    c.pbits = bits;
    c.ptable.assign((size_t)1 << bits, 0);
    uint64_t code = 0;  // left-aligned 32-bit code word accumulator
    for (size_t i = 0; i < lengths.size(); i++) {
        const int len = lengths[i];
        if (len <= 0) continue;
        if (len > bits) throw BitstreamError("Table size too small");
        if (code >= (1ull << 32)) throw BitstreamError("Too many VLC codes");
        // bit-reverse the len-bit code word: the stream is read LSB first
        uint32_t word = (uint32_t)(code >> (32 - len)), rev = 0;
        for (int b = 0; b < len; b++) rev |= ((word >> b) & 1u) << (len - 1 - b);
        for (uint32_t idx = rev; idx < (1u << bits); idx += (1u << len)) c.ptable[idx] = ((uint32_t)len << 16) | (uint32_t)symbols[i];
        code += 1ull << (32 - len);
    }
    if (code != (1ull << 32)) throw BitstreamError("Not enough VLC codes");
}

// fixed code of the ANS log-count alphabet (ANSSymbolDistribution.java:14-33): 7-bit lookup, (symbol, length)
const uint8_t kLogCountLut[128][2] = {
    {10, 3}, {12, 7}, {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {0, 5},  {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {11, 6}, {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {0, 5},  {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {13, 7}, {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {0, 5},  {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {11, 6}, {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
    {10, 3}, {0, 5},  {7, 3}, {3, 4}, {6, 3}, {8, 3}, {9, 3}, {5, 4}, {10, 3}, {4, 4},  {7, 3}, {1, 4}, {6, 3}, {8, 3}, {9, 3}, {2, 4},
};

void read_ans_code(BitReader& br, SymbolCode& c, int log_alpha) {  // ANSSymbolDistribution.java:39-133
    const int table_size = 1 << log_alpha;
    std::vector<int> freq;
    int uniq = -1;
    auto check_size = [&](int n) { if (n > table_size) throw BitstreamError("Illegal Alphabet Size"); };
    if (br.flag()) {
        if (br.flag()) {  // two symbols
            const int v1 = (int)br.u8(), v2 = (int)br.u8();
            if (v1 == v2) throw BitstreamError("Overlapping dual peak distribution");
            check_size(1 + std::max(v1, v2));
            freq.assign(1 + std::max(v1, v2), 0);
            freq[v1] = (int)br.bits(12);
            freq[v2] = 4096 - freq[v1];
            if (freq[v1] == 0) uniq = v2;
        } else {  // one symbol
            const int x = (int)br.u8();
            freq.assign(1 + x, 0);
            freq[x] = 4096;
            uniq = x;
        }
    } else if (br.flag()) {  // flat
        const int n = 1 + (int)br.u8();
        check_size(n);
        if (n == 1) uniq = 0;
        freq.assign(n, 4096 / n);
        for (int i = 0; i < 4096 % n; i++) freq[i]++;
    } else {
        int len = 0;
        while (len < 3 && br.flag()) len++;
        const int shift = (int)((br.bits(len) | (1u << len)) - 1);
        if (shift > 13) throw BitstreamError("Shift > 13");
        const int n = 3 + (int)br.u8();
        check_size(n);
        freq.assign(n, 0);
        std::vector<int> logc(n, 0), same(n, 0);
        int omit_log = -1, omit_pos = -1;
        for (int i = 0; i < n; i++) {
            const uint32_t pk = br.peek(7) & 127;
            br.skip(kLogCountLut[pk][1]);
            logc[i] = kLogCountLut[pk][0];
            if (logc[i] == 13) {
                const int rle = (int)br.u8();
                same[i] = rle + 5;
                i += rle + 3;
                continue;
            }
            if (logc[i] > omit_log) {
                omit_log = logc[i];
                omit_pos = i;
            }
        }
        if (omit_pos < 0 || (omit_pos + 1 < n && logc[omit_pos + 1] == 13)) throw BitstreamError("Invalid OmitPos");
        int total = 0, num_same = 0, prev = 0;
        for (int i = 0; i < n; i++) {
            if (same[i]) {
                num_same = same[i] - 1;
                prev = i > 0 ? freq[i - 1] : 0;
            }
            if (num_same) {
                freq[i] = prev;
                num_same--;
            } else {
                if (i == omit_pos || logc[i] == 0) continue;
                if (logc[i] == 1) {
                    freq[i] = 1;
                } else {
                    int bc = shift - ((12 - logc[i] + 1) >> 1);
                    bc = std::max(0, std::min(bc, logc[i] - 1));
                    freq[i] = (1 << (logc[i] - 1)) + ((int)br.bits(bc) << (logc[i] - 1 - bc));
                }
            }
            total += freq[i];
        }
        freq[omit_pos] = 4096 - total;
        if (freq[omit_pos] < 0) throw BitstreamError("ANS frequencies exceed 4096");
    }
    // alias table (Vose-style split into 2^log_alpha buckets of 2^(12 - log_alpha) slots)
    const int n = (int)freq.size();
    c.log_bucket = 12 - log_alpha;
    const int bucket = 1 << c.log_bucket;
    c.freq.assign(std::max(n, table_size), 0);
    for (int i = 0; i < n; i++) c.freq[i] = (uint16_t)freq[i];
    c.cutoff.assign(table_size, 0);
    c.alias_sym.assign(table_size, 0);
    c.offset.assign(table_size, 0);
    if (uniq >= 0) {
        for (int i = 0; i < table_size; i++) {
            c.alias_sym[i] = (uint16_t)uniq;
            c.offset[i] = i * bucket;
            c.cutoff[i] = 0;
        }
        return;
    }
    std::vector<int> cut(table_size, 0), over, under;
    for (int i = 0; i < n; i++) {
        cut[i] = freq[i];
        c.alias_sym[i] = (uint16_t)i;
        if (cut[i] > bucket) over.push_back(i);
        else if (cut[i] < bucket) under.push_back(i);
    }
    for (int i = n; i < table_size; i++) under.push_back(i);
    while (!over.empty()) {
        if (under.empty()) throw BitstreamError("ANS alias table: inconsistent frequencies");
        const int u = under.back(), o = over.back();
        under.pop_back();
        over.pop_back();
        const int by = bucket - cut[u];
        cut[o] -= by;
        c.alias_sym[u] = (uint16_t)o;
        c.offset[u] = cut[o];
        if (cut[o] < bucket) under.push_back(o);
        else if (cut[o] > bucket) over.push_back(o);
    }
    for (int i = 0; i < table_size; i++) {
        if (cut[i] == bucket) {
            c.alias_sym[i] = (uint16_t)i;
            c.offset[i] = 0;
            c.cutoff[i] = 0;
        } else {
            c.offset[i] -= cut[i];
            c.cutoff[i] = (uint16_t)cut[i];
        }
    }
}

// sort symbols by (code length, symbol value): the canonical code assignment order
void canonical_order(const std::vector<int>& len_of_symbol, std::vector<int>& lengths, std::vector<int>& symbols) {
    const int n = (int)len_of_symbol.size();
    symbols.resize(n);
    for (int i = 0; i < n; i++) symbols[i] = i;
    std::stable_sort(symbols.begin(), symbols.end(), [&](int a, int b) { return len_of_symbol[a] < len_of_symbol[b]; });
    lengths.resize(n);
    for (int i = 0; i < n; i++) lengths[i] = len_of_symbol[symbols[i]];
}

void read_prefix_code(BitReader& br, SymbolCode& c, int alphabet_size) {  // PrefixSymbolDistribution.java
    if (alphabet_size == 1) {
        c.single_symbol = 0;
        return;
    }
    const int log_alpha = ceil_log1p((uint64_t)alphabet_size - 1);
    const int hskip = (int)br.bits(2);
    if (hskip == 1) {  // simple code: 1..4 symbols given explicitly
        const int nsym = 1 + (int)br.bits(2);
        int sym[4] = {0, 0, 0, 0};
        for (int i = 0; i < nsym; i++) sym[i] = (int)br.bits(log_alpha);
        const bool tree_select = nsym == 4 ? br.flag() : false;
        std::vector<int> lengths, symbols;
        int bits = 0;
        switch (nsym) {
            case 1:
                c.single_symbol = sym[0];
                return;
            case 2:
                bits = 1;
                if (sym[0] > sym[1]) std::swap(sym[0], sym[1]);
                lengths = {1, 1};
                break;
            case 3:
                bits = 2;
                if (sym[1] > sym[2]) std::swap(sym[1], sym[2]);
                lengths = {1, 2, 2};
                break;
            default:
                if (tree_select) {
                    bits = 3;
                    if (sym[2] > sym[3]) std::swap(sym[2], sym[3]);
                    lengths = {1, 2, 3, 3};
                } else {
                    bits = 2;
                    std::sort(sym, sym + 4);
                    lengths = {2, 2, 2, 2};
                }
        }
        symbols.assign(sym, sym + nsym);
        build_prefix_table(c, bits, lengths, symbols);
        return;
    }
    // complex code: code lengths themselves are prefix coded (RFC 7932 section 3.5)
    static const uint8_t kL0[16][2] = {{0, 2}, {4, 2}, {3, 2}, {2, 3}, {0, 2}, {4, 2}, {3, 2}, {1, 4},
                                       {0, 2}, {4, 2}, {3, 2}, {2, 3}, {0, 2}, {4, 2}, {3, 2}, {5, 4}};
    static const int kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    std::vector<int> l1(18, 0);
    int total = 0, num_codes = 0;
    for (int i = hskip; i < 18; i++) {
        const uint32_t pk = br.peek(4) & 15;
        br.skip(kL0[pk][1]);
        const int code = kL0[pk][0];
        l1[kOrder[i]] = code;
        if (code) {
            total += 32 >> code;
            num_codes++;
        }
        if (total >= 32) break;
    }
    if ((total != 32 && num_codes >= 2) || num_codes < 1) throw BitstreamError("Invalid Level 1 Prefix codes");
    SymbolCode l1code;
    if (num_codes == 1) {
        for (int i = 0; i < 18; i++)
            if (l1[i]) l1code.single_symbol = i;
    } else {
        std::vector<int> lengths, symbols;
        canonical_order(l1, lengths, symbols);
        build_prefix_table(l1code, 5, lengths, symbols);
    }
    auto read_l1 = [&]() -> int {
        if (l1code.single_symbol >= 0) return l1code.single_symbol;
        const uint32_t e = l1code.ptable[br.peek(5) & 31];
        br.skip((int)(e >> 16));
        return (int)(e & 0xffff);
    };
    std::vector<int> l2(alphabet_size, 0);
    int prev = 8, prev_repeat = 0, prev_zero = 0, nonzero = 0;
    total = 0;
    for (int i = 0; i < alphabet_size; i++) {
        const int code = read_l1();
        if (code == 16) {
            int extra = 3 + (int)br.bits(2);
            if (prev_repeat > 0) extra = 4 * (prev_repeat - 2) - prev_repeat + extra;
            if (i + extra > alphabet_size) throw BitstreamError("Prefix code repeat past the alphabet");
            for (int j = 0; j < extra; j++) l2[i + j] = prev;
            total += (32768 >> prev) * extra;
            nonzero += extra;
            i += extra - 1;
            prev_repeat += extra;
            prev_zero = 0;
        } else if (code == 17) {
            int extra = 3 + (int)br.bits(3);
            if (prev_zero > 0) extra = 8 * (prev_zero - 2) - prev_zero + extra;
            if (i + extra > alphabet_size) throw BitstreamError("Prefix code zero run past the alphabet");
            i += extra - 1;
            prev_repeat = 0;
            prev_zero += extra;
        } else {
            l2[i] = code;
            prev_repeat = prev_zero = 0;
            if (code) {
                total += 32768 >> code;
                prev = code;
                nonzero++;
            }
        }
        if (total >= 32768) break;
    }
    if (total != 32768 && nonzero > 1) throw BitstreamError("Invalid Level 2 Prefix Codes");
    if (nonzero == 0) throw BitstreamError("Invalid Level 2 Prefix Codes");
    if (nonzero == 1) {  // a single used symbol: zero-length code
        for (int i = 0; i < alphabet_size; i++)
            if (l2[i]) c.single_symbol = i;
        return;
    }
    std::vector<int> lengths, symbols;
    canonical_order(l2, lengths, symbols);
    build_prefix_table(c, 15, lengths, symbols);
}

}  // namespace

int read_cluster_map(BitReader& br, std::vector<uint8_t>& map, int max_clusters) {  // EntropyStream.java:54-98
    const int n = (int)map.size();
    if (n == 1) {
        map[0] = 0;
    } else if (br.flag()) {  // simple: fixed-width entries
        const int nbits = (int)br.bits(2);
        for (int i = 0; i < n; i++) map[i] = (uint8_t)br.bits(nbits);
    } else {
        const bool mtf = br.flag();
        auto nested = std::make_shared<EntropyCode>();
        nested->read(br, 1, n > 2);
        EntropyDecoder dec(nested);
        for (int i = 0; i < n; i++) {
            const uint32_t v = dec.read(br, 0);
            if (v > 255) throw BitstreamError("Cluster index too large");
            map[i] = (uint8_t)v;
        }
        dec.check_final("Nested distribution");
        if (mtf) {  // inverse move-to-front
            uint8_t tab[256];
            for (int i = 0; i < 256; i++) tab[i] = (uint8_t)i;
            for (int i = 0; i < n; i++) {
                const int idx = map[i];
                const uint8_t v = tab[idx];
                map[i] = v;
                if (idx) {
                    memmove(tab + 1, tab, (size_t)idx);
                    tab[0] = v;
                }
            }
        }
    }
    int clusters = 0;
    for (int i = 0; i < n; i++) clusters = std::max(clusters, map[i] + 1);
    if (clusters > max_clusters) throw BitstreamError("Too many clusters");
    return clusters;
}

void EntropyCode::read(BitReader& br, int num_ctx, bool allow_lz77) {  // EntropyStream.java:118-163
    if (num_ctx <= 0) throw std::invalid_argument("Num Dists must be positive");
    lz77 = br.flag();
    if (lz77) {
        if (!allow_lz77) throw BitstreamError("Nested distributions cannot use LZ77");
        lz_min_symbol = br.u32(224, 0, 512, 0, 4096, 0, 8, 15);
        lz_min_length = br.u32(3, 0, 4, 0, 5, 2, 9, 8);
        num_ctx++;
        lz_len_cfg.read(br, 8);
    }
    cluster.assign(num_ctx, 0);
    const int n_codes = read_cluster_map(br, cluster, num_ctx);
    codes.assign(n_codes, SymbolCode());
    prefix = br.flag();
    log_alpha = prefix ? 15 : 5 + (int)br.bits(2);
    for (auto& c : codes) c.cfg.read(br, log_alpha);
    if (prefix) {
        std::vector<int> sizes(n_codes, 1);
        for (int i = 0; i < n_codes; i++)
            if (br.flag()) {
                const int n = (int)br.bits(4);
                sizes[i] = 1 + (1 << n) + (int)br.bits(n);
            }
        for (int i = 0; i < n_codes; i++) read_prefix_code(br, codes[i], sizes[i]);
    } else {
        for (auto& c : codes) read_ans_code(br, c, log_alpha);
    }
}

void EntropyDecoder::reset(std::shared_ptr<const EntropyCode> code) {
    code_ = std::move(code);
    has_state_ = false;
    state_ = 0;
    num_to_copy_ = copy_pos_ = num_decoded_ = 0;
    if (code_ && code_->lz77 && window_.empty()) window_.assign(1u << 20, 0);
}

uint32_t EntropyDecoder::symbol(BitReader& br, const SymbolCode& c) {
    if (code_->prefix) {
        if (c.single_symbol >= 0) return (uint32_t)c.single_symbol;
        const uint32_t e = c.ptable[br.peek(c.pbits) & ((1u << c.pbits) - 1)];
        if ((e >> 16) == 0) throw BitstreamError("Illegal VLC codes");
        br.skip((int)(e >> 16));
        return e & 0xffff;
    }
    if (!has_state_) {  // ANSSymbolDistribution.readSymbol: the state is read lazily
        state_ = br.bits(32);
        has_state_ = true;
    }
    const uint32_t index = state_ & 0xfff;
    const uint32_t i = index >> c.log_bucket, pos = index & ((1u << c.log_bucket) - 1);
    const bool greater = pos >= c.cutoff[i];
    const uint32_t sym = greater ? c.alias_sym[i] : i;
    const uint32_t off = greater ? (uint32_t)(c.offset[i] + (int32_t)pos) : pos;
    state_ = c.freq[sym] * (state_ >> 12) + off;
    if (state_ < (1u << 16)) state_ = (state_ << 16) | br.bits(16);
    return sym;
}

uint32_t EntropyDecoder::hybrid(BitReader& br, const HybridUint& h, uint32_t token) {  // EntropyStream.readHybridInteger
    const uint32_t split = 1u << h.split_exp;
    if (token < split) return token;
    const int n = h.split_exp - h.lsb - h.msb + (int)((token - split) >> (h.msb + h.lsb));
    if (n > 32) throw BitstreamError("n is too large");
    const uint32_t low = token & ((1u << h.lsb) - 1);
    token >>= h.lsb;
    token &= (1u << h.msb) - 1;
    token |= 1u << h.msb;
    const uint64_t v = (((uint64_t)token << n) | br.bits(n)) << h.lsb;
    return (uint32_t)v | low;
}

namespace {
const int8_t kSpecialDistances[120][2] = {  // EntropyStream.java:14-28
    {0, 1}, {1, 0}, {1, 1}, {-1, 1}, {0, 2}, {2, 0}, {1, 2}, {-1, 2}, {2, 1}, {-2, 1}, {2, 2}, {-2, 2}, {0, 3}, {3, 0}, {1, 3},
    {-1, 3}, {3, 1}, {-3, 1}, {2, 3}, {-2, 3}, {3, 2}, {-3, 2}, {0, 4}, {4, 0}, {1, 4}, {-1, 4}, {4, 1}, {-4, 1}, {3, 3}, {-3, 3},
    {2, 4}, {-2, 4}, {4, 2}, {-4, 2}, {0, 5}, {3, 4}, {-3, 4}, {4, 3}, {-4, 3}, {5, 0}, {1, 5}, {-1, 5}, {5, 1}, {-5, 1}, {2, 5},
    {-2, 5}, {5, 2}, {-5, 2}, {4, 4}, {-4, 4}, {3, 5}, {-3, 5}, {5, 3}, {-5, 3}, {0, 6}, {6, 0}, {1, 6}, {-1, 6}, {6, 1}, {-6, 1},
    {2, 6}, {-2, 6}, {6, 2}, {-6, 2}, {4, 5}, {-4, 5}, {5, 4}, {-5, 4}, {3, 6}, {-3, 6}, {6, 3}, {-6, 3}, {0, 7}, {7, 0}, {1, 7},
    {-1, 7}, {5, 5}, {-5, 5}, {7, 1}, {-7, 1}, {4, 6}, {-4, 6}, {6, 4}, {-6, 4}, {2, 7}, {-2, 7}, {7, 2}, {-7, 2}, {3, 7}, {-3, 7},
    {7, 3}, {-7, 3}, {5, 6}, {-5, 6}, {6, 5}, {-6, 5}, {8, 0}, {4, 7}, {-4, 7}, {7, 4}, {-7, 4}, {8, 1}, {8, 2}, {6, 6}, {-6, 6},
    {8, 3}, {5, 7}, {-5, 7}, {7, 5}, {-7, 5}, {8, 4}, {6, 7}, {-6, 7}, {7, 6}, {-7, 6}, {8, 5}, {7, 7}, {-7, 7}, {8, 6}, {8, 7},
};
}

uint32_t EntropyDecoder::read(BitReader& br, int ctx, uint32_t dist_multiplier) {  // EntropyStream.readSymbol
    const EntropyCode& ec = *code_;
    for (;;) {
        if (num_to_copy_ > 0) {
            const uint32_t v = window_[copy_pos_++ & 0xfffff];
            num_to_copy_--;
            window_[num_decoded_++ & 0xfffff] = v;
            return v;
        }
        if (ctx < 0 || ctx >= (int)ec.cluster.size()) throw std::invalid_argument("Context cannot be bigger than bundle length");
        const SymbolCode& c = ec.codes[ec.cluster[ctx]];
        uint32_t token = symbol(br, c);
        if (ec.lz77 && token >= ec.lz_min_symbol) {
            const SymbolCode& dc = ec.codes[ec.cluster.back()];
            num_to_copy_ = ec.lz_min_length + hybrid(br, ec.lz_len_cfg, token - ec.lz_min_symbol);
            token = symbol(br, dc);
            int64_t distance = hybrid(br, dc.cfg, token);
            if (dist_multiplier == 0) {
                distance++;
            } else if (distance < 120) {
                distance = kSpecialDistances[distance][0] + (int64_t)dist_multiplier * kSpecialDistances[distance][1];
                if (distance < 1) distance = 1;
            } else {
                distance -= 119;
            }
            distance = std::min<int64_t>(distance, 1 << 20);
            distance = std::min<int64_t>(distance, num_decoded_);
            copy_pos_ = num_decoded_ - (uint32_t)distance;
            if (num_to_copy_ == 0) continue;
            continue;  // the copy loop above produces the value
        }
        const uint32_t v = hybrid(br, c.cfg, token);
        if (ec.lz77) window_[num_decoded_++ & 0xfffff] = v;
        return v;
    }
}

}  // namespace jxf
