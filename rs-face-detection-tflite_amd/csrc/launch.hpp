// launch.hpp — one way to launch a kernel, two destinations: a HIP stream (eager) or a hipGraph under construction.
//
// A model's launch plan is replayed as a hipGraph.  The graph is built EXPLICITLY (hipGraphAddKernelNode with the
// dependencies the plan knows), not by stream capture: capture is process-wide state in the HIP runtime — on this ROCm a
// hipDeviceSynchronize on ANY other host thread (torch.cuda.synchronize() in a worker, another handle's blocking copy)
// invalidates an open capture in every capture mode, and fails itself with hipErrorStreamCaptureUnsupported.  A library
// whose `infer(&self)` may be called from several threads (face_detection.rs:205) cannot own process-wide state like that.
//
// While a GraphRecorder is installed on the calling thread (thread-local), launch_kernel() adds a kernel node instead of
// launching, and record_event()/wait_event() translate the plan's fork/join between its streams into node dependencies:
// every stream has a "tail" (the nodes the next node on that stream must follow); recording an event snapshots the tail,
// waiting for it merges the snapshot into the waiting stream's tail.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <utility>
#include <vector>

namespace mi {

struct GraphRecorder {
    hipGraph_t graph = nullptr;
    std::map<hipStream_t, std::vector<hipGraphNode_t>> tail;
    std::map<hipEvent_t, std::vector<hipGraphNode_t>> events;
    int kernels = 0;

    hipError_t add_kernel(hipStream_t s, const void* func, dim3 grid, dim3 block, size_t shmem, void** params) {
        hipKernelNodeParams p{};
        p.func = const_cast<void*>(func);
        p.gridDim = grid;
        p.blockDim = block;
        p.sharedMemBytes = static_cast<unsigned>(shmem);
        p.kernelParams = params;
        p.extra = nullptr;
        std::vector<hipGraphNode_t>& deps = tail[s];
        hipGraphNode_t node = nullptr;
        hipError_t e = hipGraphAddKernelNode(&node, graph, deps.empty() ? nullptr : deps.data(), deps.size(), &p);
        if (e != hipSuccess) return e;
        deps.assign(1, node);
        kernels++;
        return hipSuccess;
    }
    void record(hipEvent_t ev, hipStream_t s) { events[ev] = tail[s]; }
    void wait(hipStream_t s, hipEvent_t ev) {
        std::vector<hipGraphNode_t>& deps = tail[s];
        for (hipGraphNode_t n : events[ev])
            if (std::find(deps.begin(), deps.end(), n) == deps.end()) deps.push_back(n);
    }
};

// the recorder of the calling thread (null: launches go to their stream)
inline GraphRecorder*& current_recorder() {
    static thread_local GraphRecorder* r = nullptr;
    return r;
}

namespace detail {
template <typename Tuple, size_t... I>
inline hipError_t record_kernel(GraphRecorder* r, hipStream_t s, const void* func, dim3 grid, dim3 block, size_t shmem, Tuple& args,
                                std::index_sequence<I...>) {
    void* ptrs[] = {static_cast<void*>(&std::get<I>(args))...};
    return r->add_kernel(s, func, grid, block, shmem, ptrs);
}
}  // namespace detail

// hipLaunchKernelGGL(kern, grid, block, shmem, stream, args...) — or the same launch as a graph node.  Returns the launch status.
template <typename... KArgs, typename... Args>
inline hipError_t launch_kernel(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count does not match the kernel's parameters");
    if (GraphRecorder* r = current_recorder()) {
        std::tuple<KArgs...> copy(static_cast<KArgs>(args)...);  // the kernel's own parameter types, by value
        return detail::record_kernel(r, s, reinterpret_cast<const void*>(kern), grid, block, shmem, copy, std::index_sequence_for<KArgs...>{});
    }
    hipLaunchKernelGGL(kern, grid, block, shmem, s, std::forward<Args>(args)...);
    return hipGetLastError();
}

// Per-device launcher state.  A process may hold handles on several devices (every mi_*_create takes a device ordinal) and
// call them from several threads, so nothing here is a plain function-local flag.

// compute units of the CURRENT device (MI355X: 256); 256 when no device answers (host-only planning: mi_plan_describe)
inline int device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if ((v = cache[dev].load(std::memory_order_relaxed)) > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize = the CU's 160 KB) once per (kernel, device)
inline hipError_t allow_full_lds(const void* func) {
    static std::mutex m;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(m);
    if (done.count({func, dev})) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({func, dev});
    return e;
}

inline hipError_t record_event(hipEvent_t ev, hipStream_t s) {
    if (GraphRecorder* r = current_recorder()) {
        r->record(ev, s);
        return hipSuccess;
    }
    return hipEventRecord(ev, s);
}

inline hipError_t wait_event(hipStream_t s, hipEvent_t ev) {
    if (GraphRecorder* r = current_recorder()) {
        r->wait(s, ev);
        return hipSuccess;
    }
    return hipStreamWaitEvent(s, ev, 0);
}

}  // namespace mi
