// Bit reader over a byte buffer (JPEG XL codestream order: LSB first). Host-side front-end:
// the counterpart of J/io/Bitreader.java, written for random access into TOC sections instead of a pull stream.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace jxf {

struct BitstreamError : std::runtime_error {  // J/io/InvalidBitstreamException
    explicit BitstreamError(const std::string& m) : std::runtime_error(m) {}
};
struct UnsupportedError : std::runtime_error {  // java.lang.UnsupportedOperationException
    explicit UnsupportedError(const std::string& m) : std::runtime_error(m) {}
};

class BitReader {
  public:
    BitReader() = default;
    BitReader(const uint8_t* data, size_t size) : p_(data), size_(size) {}

    uint32_t bits(int n) {  // 0..32 bits
        if (n == 0) return 0;
        refill();
        if (n > avail_) throw BitstreamError("unexpected end of codestream section");
        const uint32_t v = (uint32_t)(buf_ & ((1ull << n) - 1));
        buf_ >>= n;
        avail_ -= n;
        consumed_ += n;
        return v;
    }
    bool flag() { return bits(1) != 0; }
    // peek up to 32 bits without consuming; bits past the end read as zero (prefix-code lookups near the end)
    uint32_t peek(int n) {
        refill();
        return (uint32_t)(buf_ & ((n >= 64) ? ~0ull : ((1ull << n) - 1)));
    }
    void skip(int n) {
        refill();
        if (n > avail_) throw BitstreamError("unexpected end of codestream section");
        buf_ >>= n;
        avail_ -= n;
        consumed_ += n;
    }
    uint32_t u32(uint32_t c0, int u0, uint32_t c1, int u1, uint32_t c2, int u2, uint32_t c3, int u3) {  // Bitreader.readU32
        switch (bits(2)) {
            case 0: return c0 + bits(u0);
            case 1: return c1 + bits(u1);
            case 2: return c2 + bits(u2);
            default: return c3 + bits(u3);
        }
    }
    uint64_t u64() {  // Bitreader.readU64
        const uint32_t sel = bits(2);
        if (sel == 0) return 0;
        if (sel == 1) return 1 + bits(4);
        if (sel == 2) return 17 + bits(8);
        uint64_t v = bits(12);
        int shift = 12;
        while (flag()) {
            if (shift == 60) {
                v |= (uint64_t)bits(4) << shift;
                break;
            }
            v |= (uint64_t)bits(8) << shift;
            shift += 8;
        }
        return v;
    }
    uint32_t u8() {  // Bitreader.readU8 (ANS alphabet sizes)
        if (!flag()) return 0;
        const int n = (int)bits(3);
        return n == 0 ? 1 : bits(n) + (1u << n);
    }
    uint32_t enum_() {  // Bitreader.readEnum
        const uint32_t v = u32(0, 0, 1, 0, 2, 4, 18, 6);
        if (v > 63) throw BitstreamError("Enum constant > 63");
        return v;
    }
    float f16() {  // Bitreader.readF16 + MathHelper.floatFromF16
        const uint32_t b = bits(16);
        const uint32_t sign = b >> 15, exp = (b >> 10) & 31, mant = b & 1023;
        if (exp == 31) throw BitstreamError("Illegal infinite/NaN float16");
        float v;
        if (exp == 0) {
            v = (float)mant * (1.0f / 16777216.0f);  // 2^-24
        } else {
            union { uint32_t u; float f; } c;
            c.u = ((exp + 112) << 23) | (mant << 13);
            v = c.f;
        }
        return sign ? -v : v;
    }
    void align_to_byte(bool must_be_zero = true) {  // Bitreader.zeroPadToByte
        const int rem = (int)(consumed_ & 7);
        if (rem) {
            const uint32_t pad = bits(8 - rem);
            if (must_be_zero && pad != 0) throw BitstreamError("Nonzero zero-padding-to-byte");
        }
    }
    size_t bit_pos() const { return consumed_; }
    size_t byte_pos() const { return (consumed_ + 7) >> 3; }
    size_t size_bytes() const { return size_; }
    const uint8_t* data() const { return p_; }
    bool at_end() const { return consumed_ >= size_ * 8; }

  private:
    void refill() {
        while (avail_ <= 56 && next_ < size_) {
            buf_ |= (uint64_t)p_[next_++] << avail_;
            avail_ += 8;
        }
    }
    const uint8_t* p_ = nullptr;
    size_t size_ = 0, next_ = 0, consumed_ = 0;
    uint64_t buf_ = 0;
    int avail_ = 0;
};

inline int32_t unpack_signed(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }  // MathHelper.unpackSigned
inline int ceil_log2(uint64_t x) {  // MathHelper.ceilLog2 for x >= 1
    int r = 0;
    while ((1ull << r) < x) r++;
    return r;
}
inline int ceil_log1p(uint64_t x) { return x == 0 ? 0 : 64 - __builtin_clzll(x); }  // ceil(log2(x + 1))
inline int floor_log1p(uint64_t x) { return ceil_log1p((x + 1) >> 1); }             // floor(log2(x + 1)), x + 1 <= 2^63
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace jxf
