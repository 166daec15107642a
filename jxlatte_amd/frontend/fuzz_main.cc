// Robustness driver for the front-end (not part of the library): decodes every file given on the command line, then N
// mutated copies of each (byte flips, truncations, splices) and reports crashes through the sanitizers. Frame-level
// modular transforms are answered by trivial hooks (zero-filled outputs): only the parser is under test.
//   g++ -O1 -g -std=c++17 -fwrapv -fsanitize=address,undefined fuzz_main.cc entropy.cc headers.cc modular.cc frame.cc api.cc -o fuzz
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/jxlatte_frontend.h"

static int32_t sq(void*, const jxf_chan*, int32_t, const jxf_squeeze_step*, int32_t, jxf_chan* out, int32_t n_out) {
    for (int i = 0; i < n_out; i++) memset(out[i].data, 0, sizeof(int32_t) * (size_t)out[i].w * out[i].h);
    return 0;
}
static int32_t rct(void*, int32_t*, int32_t*, int32_t*, int64_t, int32_t) { return 0; }

static int run(const std::vector<uint8_t>& d) {
    char err[256];
    jxf_dec* dec = jxf_open(d.data(), d.size(), err, sizeof err);
    if (!dec) return -1;
    jxf_hooks h{nullptr, sq, rct};
    int frames = 0, st;
    while ((st = jxf_next_frame(dec, &h)) == JXF_OK && frames < 200) {
        jxf_frame_info fi;
        jxf_get_frame_info(dec, &fi);
        for (int i = 0; i < fi.num_lf_groups; i++) {
            jxf_lfgroup_view v;
            jxf_get_lfgroup(dec, i, &v);
        }
        frames++;
    }
    jxf_close(dec);
    return st < 0 ? st : frames;
}

int main(int argc, char** argv) {
    int n_mut = 300;
    std::mt19937_64 rng(12345);
    for (int a = 1; a < argc; a++) {
        if (!strncmp(argv[a], "-n", 2)) { n_mut = atoi(argv[a] + 2); continue; }
        FILE* f = fopen(argv[a], "rb");
        if (!f) continue;
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        const int base = run(d);
        int ok = 0, rejected = 0;
        for (int m = 0; m < n_mut; m++) {
            std::vector<uint8_t> e = d;
            const int kind = (int)(rng() % 4);
            if (kind == 0) {  // flip a few bits, biased towards the headers
                const int flips = 1 + (int)(rng() % 4);
                for (int k = 0; k < flips; k++) {
                    const size_t lim = (rng() & 1) ? std::min<size_t>(e.size(), 200) : e.size();
                    e[rng() % lim] ^= (uint8_t)(1u << (rng() % 8));
                }
            } else if (kind == 1) {  // truncate
                e.resize(1 + rng() % e.size());
            } else if (kind == 2) {  // overwrite a run with random bytes
                const size_t pos = rng() % e.size(), len = 1 + rng() % 16;
                for (size_t k = pos; k < std::min(e.size(), pos + len); k++) e[k] = (uint8_t)rng();
            } else {  // splice: copy one region over another
                const size_t len = 1 + rng() % std::min<size_t>(64, e.size());
                const size_t src = rng() % (e.size() - len + 1), dst = rng() % (e.size() - len + 1);
                memmove(e.data() + dst, e.data() + src, len);
            }
            const int r = run(e);
            if (r >= 0) ok++; else rejected++;
        }
        printf("%s: %d frames; %d mutants decoded, %d rejected cleanly\n", argv[a], base, ok, rejected);
    }
    return 0;
}
