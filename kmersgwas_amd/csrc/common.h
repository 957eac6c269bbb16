// common.h — error plumbing shared by the host side of libkgwas.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

#include "../../include/kgwas.h"
#include "env.h"

namespace kgwas {

// Thread-local last error (kgwas_last_error()).
void set_error(const std::string& msg);

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define KGWAS_HIP(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            throw ::kgwas::Error(KGWAS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// heap_guard.cpp: throws KGWAS_ERR_STATE unless csrc/heap.h reproduces this process's std::priority_queue (checked once).
void require_heap_emulation();

// Run body, translate exceptions into a status code + message. No exception crosses the C ABI.
template <class F>
int guarded(F&& body) {
    try {
        body();
        return KGWAS_OK;
    } catch (const Error& e) {
        set_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return KGWAS_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return KGWAS_ERR_ARG;
    }
}

}  // namespace kgwas

// Names the calling thread (comm, <= 15 characters): a hang report (tools/fuzz_parity.py's watchdog reads /proc/self/task) or
// a profiler then says WHICH of the library's helper threads it is looking at.
#include <pthread.h>
inline void kgwas_name_this_thread(const char* name) { (void)pthread_setname_np(pthread_self(), name); }

// Runs `work` on the calling thread and on up to n - 1 others (fewer if the system has no more threads to give), joins them all
// whatever happens and rethrows the first exception any of them left: `work` takes its items from a shared counter.
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>
template <class F>
void kgwas_run_on_threads(unsigned n, const char* name, F&& work) {
    std::exception_ptr err;
    std::mutex emu;
    auto guarded_work = [&](bool named) {
        try {
            if (named) kgwas_name_this_thread(name);
            work();
        } catch (...) {
            std::lock_guard<std::mutex> lk(emu);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n; t++) {
        try {
            th.emplace_back(guarded_work, true);
        } catch (const std::system_error&) {
            break;
        }
    }
    guarded_work(false);
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}
