// launch.h - events bound to a kernel launch instead of recorded behind it.
// A HIP event recorded into a stream is a packet of its own between two kernels: 2.9 us of the stream's time each
// (tools/probe_events.hip: a chain of 200 launches, 68.5 us per kernel plain, 71.4 with a hipEventRecord behind each - timing
// disabled or not), and a sparse chunk carries four (start, filter done, kernels done, counts ready). An event BOUND to a launch
// (hipExtLaunchKernel's stopEvent) takes the launch's own completion signal: its time is the kernel's end, hipEventQuery /
// hipEventSynchronize / hipEventElapsedTime work on it as on a recorded one, and it costs the stream nothing (68.6 us in the
// same chain). The launchers are called through several layers of templates, so the event travels beside them: the caller
// names it (bind_stop), the launcher's LAST kernel launch (launch_last) takes it; a launcher that returns without launching
// leaves it, and the caller records it the ordinary way.
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <utility>

// (hip/hip_ext.h's one C entry point, declared here: its templates do not compile warning-free as plain C++17)
extern "C" hipError_t hipExtLaunchKernel(const void* function_address, dim3 numBlocks, dim3 dimBlocks, void** args, size_t sharedMemBytes,
                                         hipStream_t stream, hipEvent_t startEvent, hipEvent_t stopEvent, int flags);

namespace kgwas {

extern thread_local hipEvent_t g_bound_stop;  // scan_host.cpp

template <class... KA, size_t... I>
inline hipError_t launch_bound_impl(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t stop, std::tuple<KA...>& vals,
                                    std::index_sequence<I...>) {
    void* ptrs[] = {const_cast<void*>(static_cast<const void*>(&std::get<I>(vals)))..., nullptr};
    return hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, st, nullptr, stop, 0);
}

// the last kernel launch of a launcher: with the caller's event bound to it, if there is one
template <class... KA, class... A>
inline void launch_last(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... args) {
    static_assert(sizeof...(KA) == sizeof...(A), "one argument per kernel parameter");
    const hipEvent_t stop = g_bound_stop;
    g_bound_stop = nullptr;
    if (!stop) {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, static_cast<KA>(args)...);
        return;
    }
    std::tuple<KA...> vals(static_cast<KA>(args)...);
    (void)launch_bound_impl(kernel, grid, block, lds, st, stop, vals, std::index_sequence_for<KA...>{});  // (the launcher reads hipGetLastError)
}

// run `launcher` (which ends in a launch_last) with `ev` bound to its last kernel; recorded behind it if nothing took it
template <class F>
inline hipError_t bind_stop(hipEvent_t ev, hipStream_t st, F&& launcher) {
    g_bound_stop = ev;
    const hipError_t e = launcher();
    if (g_bound_stop) {
        g_bound_stop = nullptr;
        if (e == hipSuccess) return hipEventRecord(ev, st);
    }
    return e;
}

}  // namespace kgwas
