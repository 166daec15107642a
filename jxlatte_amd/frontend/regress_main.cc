// Sanitizer regression driver for the front-end's bounds checks on untrusted geometry (not part of the library).
//   make -C jxlatte_amd/frontend regress && ./regress        (run by tests/test_frontend_asan.py)
// Each case hands the parser a hand-built input that used to corrupt the heap or divide by zero, and expects a
// BitstreamError / UnsupportedError -- under AddressSanitizer + UBSan, so a missing check fails the run.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "frame.h"

using namespace jxf;

namespace {
struct BitWriter {  // LSB-first, as BitReader reads
    std::vector<uint8_t> bytes;
    int n = 0;
    void put(uint32_t v, int bits) {
        for (int i = 0; i < bits; i++, n++) {
            if ((n & 7) == 0) bytes.push_back(0);
            bytes.back() |= ((v >> i) & 1u) << (n & 7);
        }
    }
};

int failures = 0;
template <typename E>
void expect_throw(const char* what, const std::function<void()>& f) {
    try {
        f();
    } catch (const E& e) {
        printf("ok   %-58s -> %s\n", what, e.what());
        return;
    } catch (const std::exception& e) {
        printf("FAIL %-58s -> wrong exception: %s\n", what, e.what());
        failures++;
        return;
    }
    printf("FAIL %-58s -> no exception\n", what);
    failures++;
}
}  // namespace

int main() {
    // 1. ModularStream::init: Palette over channels of unequal size (the inverse would recreate the removed channels with
    //    the index channel's shape: a restored sub-channel larger than its destination)
    expect_throw<BitstreamError>("palette over channels of unequal size", [] {
        BitWriter w;
        w.put(0, 1);              // use_global_tree = false
        w.put(1, 1);              // WPParams: default
        w.put(1, 2);              // nb_transforms: selector 1 -> 1
        w.put(1, 2);              // transform = Palette
        w.put(0, 2); w.put(0, 3); // begin_c: selector 0, 3 bits -> 0
        w.put(1, 2);              // num_c: selector 1 -> 3
        w.put(0, 2); w.put(4, 8); // nb_colors: selector 0, 8 bits -> 4
        w.put(0, 2);              // nb_deltas: selector 0 -> 0
        w.put(0, 4);              // d_pred
        w.put(0, 32);
        BitReader br(w.bytes.data(), w.bytes.size());
        std::vector<Channel> ch = {Channel(8, 8, 0, 0), Channel(4, 4, 1, 1), Channel(8, 8, 0, 0)};
        ModularStream ms;
        ms.init(br, std::move(ch), 0, nullptr, 8);
    });
    // 2. copy_back: a sub-stream hands back a channel of another size / origin than requested
    expect_throw<BitstreamError>("sub-stream channel larger than requested", [] {
        Channel dst(16, 16, 0, 0);
        Channel want = sub_channel(dst, 8, 3);  // the 8x8 part at (8, 8)
        Channel src(16, 16, 0, 0);              // what a palette inverse over unequal channels would produce
        src.oy = want.oy; src.ox = want.ox;
        src.allocate();
        copy_back(dst, src, want);
    });
    expect_throw<BitstreamError>("sub-stream channel with a moved origin", [] {
        Channel dst(16, 16, 0, 0);
        Channel want = sub_channel(dst, 8, 0);
        Channel src(8, 8, 0, 0);
        src.oy = 12; src.ox = 12;
        src.allocate();
        copy_back(dst, src, want);
    });
    {   // the good case still copies
        Channel dst(16, 16, 0, 0);
        Channel want = sub_channel(dst, 8, 3);
        Channel src = want;
        src.allocate();
        for (auto& v : src.buf) v = 7;
        copy_back(dst, src, want);
        const bool ok = dst.row(8)[8] == 7 && dst.row(15)[15] == 7 && dst.row(7)[7] == 0;
        printf("%s valid sub-channel is copied to its place\n", ok ? "ok  " : "FAIL");
        failures += !ok;
    }
    // 3. sub_channel: shifts that empty the group size (8 or more horizontal squeezes: gw = dim >> hshift = 0 -> division by
    //    zero), or that are not shifts at all
    expect_throw<BitstreamError>("hshift larger than log2(group size)", [] { sub_channel(Channel(4, 4, 0, 9), 256, 0); });
    expect_throw<BitstreamError>("shift of 32 or more", [] { sub_channel(Channel(4, 4, 40, 0), 256, 0); });
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
