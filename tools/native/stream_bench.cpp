// Native host threads for the streaming boundary leg of bench.py (measurement harness, not product code).
//
// bench.streaming_leg() drives the C-ABI from Python threads: every return from a ctypes call re-takes the interpreter lock, and
// with 8 threads the calls of one frame queue behind each other for milliseconds. jxlatte's host is a JVM whose threads hold no
// such lock (one JXLDecoder per thread), so the same call sequence is issued here from std::threads:
//   begin_frame + set_weights + set_lfgroup... + prepare + map (no fill) + the decoder's coefficient stores + commit + run +
//   read_output_begin per frame, read_output_wait one frame later -- n_ctx contexts, one thread each.
// The library is reached through dlopen of the path bench.py has loaded (no link-time dependency: JXL_AMD_LIB builds work too).
//
//   g++ -O2 -std=c++17 -fPIC -shared -pthread -I include tools/native/stream_bench.cpp -o jxlatte_amd/libjxl_stream_bench.so -ldl
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "jxlatte_amd.h"

namespace {

struct Api {
    void* h = nullptr;
    decltype(&jxl_ctx_create) ctx_create;
    decltype(&jxl_ctx_destroy) ctx_destroy;
    decltype(&jxl_last_error) last_error;
    decltype(&jxl_vardct_begin_frame) begin_frame;
    decltype(&jxl_vardct_set_weights) set_weights;
    decltype(&jxl_vardct_set_lfgroup) set_lfgroup;
    decltype(&jxl_vardct_prepare) prepare;
    decltype(&jxl_vardct_map_coeffs_i16_ex) map_ex;
    decltype(&jxl_vardct_commit_coeffs_i16_groups) commit_groups;
    decltype(&jxl_vardct_run) run;
    decltype(&jxl_vardct_read_output_begin) read_begin;
    decltype(&jxl_vardct_read_output_wait) read_wait;
};

template <class F>
bool sym(void* h, const char* name, F& f) {
    f = reinterpret_cast<F>(dlsym(h, name));
    return f != nullptr;
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Gate {  // all workers + the caller meet here once
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, need = 0;
    bool open = false, broken = false;
    bool wait() {
        std::unique_lock<std::mutex> l(m);
        if (++waiting >= need) {
            open = true;
            cv.notify_all();
        }
        cv.wait(l, [&] { return open || broken; });
        return !broken;
    }
    void abort() {
        std::lock_guard<std::mutex> l(m);
        broken = true;
        cv.notify_all();
    }
};

}  // namespace

extern "C" {

struct jxl_stream_bench_args {
    const char* lib_path;
    int32_t device, n_ctx, frames_per_ctx;
    const jxl_vardct_params* params;
    const float* weights;
    size_t n_weights;
    const int32_t* woffs;
    const jxl_lfgroup_desc* const* lfgroups;
    int32_t n_lfgroups;
    const int16_t* coeff[3];  // the decoder's output, [rows[c]][cols[c]] contiguous
    int32_t rows[3], cols[3];
    const uint8_t* group_written;
    int32_t n_groups;
    void* const* outs;  // n_ctx page-locked result buffers
    int64_t out_stride;
    // results
    double wall_s;
    double phase_s[8];  // summed over all timed frames of all contexts
    char err[256];
};

// the binding's layout check (bench.py's ctypes mirror)
size_t jxl_stream_bench_args_size(void) { return sizeof(jxl_stream_bench_args); }

int jxl_stream_bench(jxl_stream_bench_args* a) {
    Api api;
    a->err[0] = 0;
    api.h = dlopen(a->lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!api.h) {
        snprintf(a->err, sizeof a->err, "dlopen: %s", dlerror());
        return 1;
    }
    if (!(sym(api.h, "jxl_ctx_create", api.ctx_create) && sym(api.h, "jxl_ctx_destroy", api.ctx_destroy) &&
          sym(api.h, "jxl_last_error", api.last_error) && sym(api.h, "jxl_vardct_begin_frame", api.begin_frame) &&
          sym(api.h, "jxl_vardct_set_weights", api.set_weights) && sym(api.h, "jxl_vardct_set_lfgroup", api.set_lfgroup) &&
          sym(api.h, "jxl_vardct_prepare", api.prepare) && sym(api.h, "jxl_vardct_map_coeffs_i16_ex", api.map_ex) &&
          sym(api.h, "jxl_vardct_commit_coeffs_i16_groups", api.commit_groups) && sym(api.h, "jxl_vardct_run", api.run) &&
          sym(api.h, "jxl_vardct_read_output_begin", api.read_begin) && sym(api.h, "jxl_vardct_read_output_wait", api.read_wait))) {
        snprintf(a->err, sizeof a->err, "dlsym: an entry of include/jxlatte_amd.h is missing from %s", a->lib_path);
        return 1;
    }
    const int n = a->n_ctx;
    std::vector<jxl_ctx*> ctx(n, nullptr);
    for (int i = 0; i < n; i++)
        if (api.ctx_create(a->device, &ctx[i]) != JXL_OK) {
            snprintf(a->err, sizeof a->err, "jxl_ctx_create: %s", api.last_error(nullptr));
            for (int j = 0; j < i; j++) api.ctx_destroy(ctx[j]);
            return 1;
        }
    Gate gate;
    gate.need = n + 1;
    std::mutex em;
    std::string err;
    std::vector<double> t_end(n, 0.0);
    std::vector<std::vector<double>> ph(n, std::vector<double>(8, 0.0));

    auto worker = [&](int i) {
        jxl_ctx* c = ctx[i];
        bool pending = false;
        auto fail = [&](const char* what) {
            std::lock_guard<std::mutex> l(em);
            if (err.empty()) err = std::string(what) + ": " + api.last_error(c);
            gate.abort();
        };
#define CK(call, what)            \
    if ((call) != JXL_OK) {       \
        fail(what);               \
        return false;             \
    }
        auto one_frame = [&](double* p) -> bool {
            double t[9];
            t[0] = now();
            CK(api.begin_frame(c, a->params), "begin_frame");
            CK(api.set_weights(c, a->weights, a->n_weights, a->woffs), "set_weights");
            t[1] = now();
            for (int g = 0; g < a->n_lfgroups; g++) CK(api.set_lfgroup(c, a->lfgroups[g]), "set_lfgroup");
            t[2] = now();
            CK(api.prepare(c), "prepare");
            t[3] = now();
            int16_t* planes[3];
            int32_t strides[3];
            CK(api.map_ex(c, planes, strides, JXL_MAP_NO_FILL), "map_coeffs_i16_ex");
            t[4] = now();
            for (int ch = 0; ch < 3; ch++) {  // stands for the entropy decoder's stores (every group, zeros included)
                const size_t row = (size_t)a->cols[ch] * sizeof(int16_t);
                if (strides[ch] == a->cols[ch])
                    memcpy(planes[ch], a->coeff[ch], row * a->rows[ch]);
                else
                    for (int y = 0; y < a->rows[ch]; y++)
                        memcpy(planes[ch] + (size_t)y * strides[ch], a->coeff[ch] + (size_t)y * a->cols[ch], row);
            }
            t[5] = now();
            CK(api.commit_groups(c, a->group_written, a->n_groups), "commit_coeffs_i16_groups");
            t[6] = now();
            if (pending) CK(api.read_wait(c), "read_output_wait");  // frame k's pixels have landed: the buffer is free again
            CK(api.run(c), "run");
            t[7] = now();
            void* out[3] = {a->outs[i], nullptr, nullptr};
            CK(api.read_begin(c, out, a->out_stride), "read_output_begin");
            t[8] = now();
            pending = true;
            if (p)
                for (int k = 0; k < 8; k++) p[k] += t[k + 1] - t[k];
            return true;
        };
        if (!one_frame(nullptr)) return;  // allocations, page-locking
        if (api.read_wait(c) != JXL_OK) return fail("read_output_wait");
        pending = false;
        if (!gate.wait()) return;
        for (int f = 0; f < a->frames_per_ctx; f++)
            if (!one_frame(ph[i].data())) return;
        if (api.read_wait(c) != JXL_OK) return fail("read_output_wait");
        t_end[i] = now();
#undef CK
    };

    std::vector<std::thread> th;
    for (int i = 0; i < n; i++) th.emplace_back(worker, i);
    const bool ok = gate.wait();
    const double t0 = now();
    for (auto& t : th) t.join();
    for (int i = 0; i < n; i++) api.ctx_destroy(ctx[i]);
    if (!ok || !err.empty()) {
        snprintf(a->err, sizeof a->err, "%s", err.empty() ? "a worker failed before the start" : err.c_str());
        return 1;
    }
    double last = 0.0;
    for (int i = 0; i < n; i++) last = t_end[i] > last ? t_end[i] : last;
    a->wall_s = last - t0;
    for (int k = 0; k < 8; k++) {
        a->phase_s[k] = 0.0;
        for (int i = 0; i < n; i++) a->phase_s[k] += ph[i][k];
    }
    return 0;
}

}  // extern "C"
