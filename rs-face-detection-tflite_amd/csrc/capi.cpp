// capi.cpp — extern "C" boundary (include/mi_face.h).  Translates exceptions into status codes + mi_last_error().
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi_face.h"
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <thread>

#include "engine.hpp"
#include "launch.hpp"
#include "host_glue.hpp"
#include "kernels.hpp"
#include "jpeg.hpp"
#include "preproc.hpp"

namespace {

thread_local std::string g_error;

struct ApiError : std::runtime_error {
    int code;
    ApiError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return MI_OK;
    } catch (const ApiError& e) {
        g_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_error = "out of memory";
        return MI_ENOMEM;
    } catch (const std::exception& e) {
        g_error = e.what();
        std::string m = e.what();
        if (m.rfind("tflite:", 0) == 0 || m.rfind("plan:", 0) == 0) return MI_EMODEL;
        if (m.find("hip") != std::string::npos || m.find("HIP") != std::string::npos || m.find("device") != std::string::npos ||
            m.find("kernel") != std::string::npos)
            return MI_EDEVICE;
        return MI_EINVAL;
    } catch (...) {
        g_error = "unknown error";
        return MI_EINVAL;
    }
}

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ApiError(MI_EIO, "cannot open model file '" + path + "'");
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (data.empty()) throw ApiError(MI_EIO, "model file '" + path + "' is empty");
    return data;
}

// small device scratch that grows on demand
struct DeviceBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* get(size_t bytes) {
        if (bytes > cap) {
            if (p) hipFree(p);
            p = nullptr;
            mi::hip_check(hipMalloc(&p, bytes), "hipMalloc scratch");
            cap = bytes;
        }
        return p;
    }
    ~DeviceBuf() {
        if (p) hipFree(p);
    }
};

// The single-image entries (one picture per call: the reference's only operating point, face_detection.rs:205) keep every small
// operand and every result of a call in ONE pinned host block that is mapped into the device's address space: the kernels read the
// padding / ROI / flags from it and write their results into it.  A call is then: one copy of the picture, the pre-processing launch,
// the network's graph, the post-processing launch and ONE synchronisation, with no small copies either way.
struct OneShot {
    void* host = nullptr;
    void* dev = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (host) hipHostFree(host);
        host = dev = nullptr;
        cap = 0;
        // (coherent: the kernels' stores must be visible to the host once the stream has been synchronised, without a copy)
        mi::hip_check(hipHostMalloc(&host, bytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
        mi::hip_check(hipHostGetDevicePointer(&dev, host, 0), "hipHostGetDevicePointer");
        cap = bytes;
    }
    template <class T> T* h(size_t off) const { return reinterpret_cast<T*>(static_cast<char*>(host) + off); }
    template <class T> T* d(size_t off) const { return reinterpret_cast<T*>(static_cast<char*>(dev) + off); }
    ~OneShot() {
        if (host) hipHostFree(host);
    }
};
// A single-launch run (bandnet_kernels.hip) needs every one of its workgroups resident, one per CU, until it has finished: the handles of
// a process share the device's CUs through this count.  A call that does not get its CUs runs on the batched plan instead — it never waits.
std::atomic<int> g_band_cus[64];
struct BandClaim {
    int dev = 0, n = 0;
    bool ok = false;
    BandClaim(mi::Model& m, int batch) : BandClaim(m.device(), m.band_workgroups(batch)) {}
    // (launches that follow one another on one stream share a claim of the largest of them)
    BandClaim(int device, int workgroups) {
        dev = device;
        n = workgroups;
        if (n <= 0 || dev < 0 || dev >= 64) return;
        // A CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) takes CUs away that hipDeviceAttributeMultiprocessorCount still counts: the launch's
        // workgroups would not all be resident, every call would wait out the kernel's bounded spin and then repeat itself.  Under a
        // mask the single-image entries stay on the batched plan (MI_BAND_WITH_CU_MASK=1: the mask leaves enough CUs, use the plan anyway).
        static const bool masked = (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) && !getenv("MI_BAND_WITH_CU_MASK");
        if (masked) return;
        if (g_band_cus[dev].fetch_add(n) + n <= mi::device_cu_count()) ok = true;
        else g_band_cus[dev].fetch_sub(n);
    }
    ~BandClaim() {
        if (ok) g_band_cus[dev].fetch_sub(n);
    }
    BandClaim(const BandClaim&) = delete;
    BandClaim& operator=(const BandClaim&) = delete;
};
// offsets of the small operands inside the block; results start at kOneResults
constexpr size_t kOnePad = 0, kOneRoi = 64, kOneSize = 128, kOneFlip = 136, kOneCount = 192, kOneResults = 256;

void require(bool ok, const char* msg) {
    if (!ok) throw ApiError(MI_EINVAL, msg);
}

}  // namespace

// Concurrency (face_detection.rs:205: `infer(&self)` may be called from several threads): a handle's interpreter state —
// activation arena, captured hipGraphs, staging buffers — is shared by its callers, so calls on one handle are SERIALISED
// by `mu`, and because a call on a caller-supplied stream returns before its kernels finish, the next call on a
// DIFFERENT stream first waits (on the device, hipStreamWaitEvent) for the event the previous call recorded.  Results
// are therefore the same as with one caller; for parallel execution use one handle per worker thread / stream.
struct mi_model {
    std::unique_ptr<mi::Model> m;
    std::mutex mu;
    hipEvent_t done = nullptr;     // recorded after the last enqueued call
    hipStream_t last = nullptr;    // stream of that call
    bool busy = false;
    ~mi_model() {
        if (done) hipEventDestroy(done);
    }
};

namespace {
// One call's claim on a handle: holds the mutex, orders this call's stream after the previous call's work.
// Construct after hipSetDevice; the destructor (also on the error paths) records the new tail of the handle's work.
struct Use {
    mi_model& h;
    hipStream_t s;
    std::unique_lock<std::mutex> lock;
    Use(mi_model& handle, hipStream_t stream) : h(handle), s(stream), lock(handle.mu) {
        if (h.busy && h.last != s) mi::hip_check(hipStreamWaitEvent(s, h.done, 0), "hipStreamWaitEvent");
    }
    ~Use() {
        if (!h.done && hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess) h.done = nullptr;
        if (h.done && hipEventRecord(h.done, s) == hipSuccess) {
            h.last = s;
            h.busy = true;
        }
    }
};
}  // namespace

// One of the two slots of the host-fed batch form (mi_fd_submit_images / mi_fd_collect): its own copy stream, device frames
// and results, pinned host results.  The network itself runs on the handle's stream (one arena per handle), so the copies
// of one slot overlap the kernels of the other.
struct FdSlot {
    hipStream_t copy = nullptr;
    hipEvent_t copied = nullptr, done = nullptr;
    DeviceBuf d_frames, d_pad;
    void* h_out = nullptr;      // pinned + mapped: [batch][cap] detections, then [batch] counts (written by the post-processing kernel)
    size_t h_cap = 0;
    int batch = 0, cap = 0;
    bool pending = false;
    ~FdSlot() {
        if (pending && done) hipEventSynchronize(done);  // a batch was submitted and never collected: its kernels still write into h_out
        if (copy) hipStreamDestroy(copy);
        if (copied) hipEventDestroy(copied);
        if (done) hipEventDestroy(done);
        if (h_out) hipHostFree(h_out);
    }
};

// One of the two slots of the streamed JPEG form (mi_fd_submit_jpeg / mi_fd_collect_jpeg; utils.rs:8-21 + face_detection.rs:205 for a stream of
// encoded pictures): the entropy decoder writes the coefficients into the slot's pinned block, their copy runs on the slot's own stream, the
// sample arithmetic / image_to_tensor / network / post-processing on the handle's stream, the results land in the slot's pinned mapped block.
struct JpegSlot {
    hipStream_t copy = nullptr;
    hipEvent_t copied = nullptr, done = nullptr;
    void* h_coef = nullptr;        // pinned: coefficients, then the quantisation tables
    size_t h_coef_cap = 0;
    DeviceBuf d_coef, d_qt, d_planes, d_rgb;
    OneShot one;                   // padding in; detections + count out
    mi::JpegFrame f;               // the picture in flight (its coefficients live in h_coef)
    int cap = 0;
    bool pending = false, band = false, suspect = false, claimed = false;
    ~JpegSlot() {
        if (pending && done) hipEventSynchronize(done);
        if (copy) hipStreamDestroy(copy);
        if (copied) hipEventDestroy(copied);
        if (done) hipEventDestroy(done);
        if (h_coef) hipHostFree(h_coef);
    }
};

struct mi_fd {
    mi_model model;
    int kind = 0, in_w = 0, in_h = 0, n_anchors = 0;
    std::vector<float> anchors;
    float* d_anchors = nullptr;
    float* d_lut = nullptr;  // u8 -> f32 in (-1, 1), 256 entries (frames of the network's own size feed the first convolution as bytes)
    DeviceBuf d_in, d_pad, d_out, d_counts, d_img, d_geom, d_roi;
    OneShot one;
    FdSlot slot[2];
    JpegSlot jslot[2];
    std::unique_ptr<BandClaim> jclaim;   // the CUs of the single-launch plan, held while a streamed picture is in flight (both slots run on one stream: one claim)
    int jclaim_users = 0;
    ~mi_fd() {
        for (JpegSlot& sl : jslot)
            if (sl.pending && sl.done) hipEventSynchronize(sl.done);
        //  a batch that was submitted and never collected still reads the anchors and the table: wait for it before
        // anything is freed (hipFree happens to synchronise the device; this does not rely on it)
        for (FdSlot& sl : slot)
            if (sl.pending && sl.done) hipEventSynchronize(sl.done);
        if (d_anchors) hipFree(d_anchors);
        if (d_lut) hipFree(d_lut);
    }
};

// One of the two slots of the mesh's host-fed batch form (mi_fl_submit_images / mi_fl_collect), like FdSlot: its own copy stream, device
// frames and ROIs, and a pinned + mapped host block that the projection kernel writes the results into.
struct FlSlot {
    hipStream_t copy = nullptr;
    hipEvent_t copied = nullptr, done = nullptr;
    DeviceBuf d_frames, d_rois;
    void* h_out = nullptr;      // [N][468][3] landmarks, [N] present, [N] raw flags
    size_t h_cap = 0;
    int N = 0;
    bool pending = false;
    ~FlSlot() {
        if (pending && done) hipEventSynchronize(done);
        if (copy) hipStreamDestroy(copy);
        if (copied) hipEventDestroy(copied);
        if (done) hipEventDestroy(done);
        if (h_out) hipHostFree(h_out);
    }
};

struct mi_fl {
    mi_model model;
    int in_w = 0, in_h = 0;
    DeviceBuf d_in, d_roi, d_size, d_lm, d_present, d_flag, d_img, d_geom, d_sizes_b;
    int sizes_N = 0, sizes_w = 0, sizes_h = 0;  // what d_sizes_b holds (mi_fl_infer_images: uploaded when the batch geometry changes)
    OneShot one;
    FlSlot slot[2];
    ~mi_fl() {
        for (FlSlot& sl : slot)
            if (sl.pending && sl.done) hipEventSynchronize(sl.done);   // its kernels still read the handle's buffers
    }
};

struct mi_pipeline {
    std::unique_ptr<mi_fd> fd;
    std::unique_ptr<mi_fl> fl;
    std::unique_ptr<mi_iris> iris;
    DeviceBuf frames, geom, pad_det, pad_eye, in_det, dets, roi_face, valid_face, in_lm, roi_eye, valid_eye, flip_eye, in_eye, sizes;
    // results of a call from host memory: ONE device block (faces | counts | landmarks | present | eyes) and its pinned host image — one
    // asynchronous copy and one synchronisation per call (five copies into the caller's pageable arrays were five synchronous round trips:
    // 115 us of the 0.66 ms of a one-picture call)
    DeviceBuf results;
    OneShot out;
    DeviceBuf geom_det;                              // the detector's letterbox geometry of `geom_B` pictures of geom_w x geom_h (no ROI: uploaded once)
    int geom_B = 0, geom_w = 0, geom_h = 0;
    int sizes_B = 0, sizes_w = 0, sizes_h = 0;  // what `sizes` holds (uploaded once per batch geometry, not per call)
    // the streamed form from encoded pictures (mi_pipeline_submit_jpeg / mi_pipeline_collect_jpeg): per slot the decoder's state and picture, the
    // results' device block and its pinned image
    struct PipeJpegSlot {
        JpegSlot jpeg;
        DeviceBuf results;
        OneShot out;
        bool pending = false, band = false, suspect = false, claimed = false;
    };
    PipeJpegSlot jslot[2];
    std::unique_ptr<BandClaim> jclaim;
    int jclaim_users = 0;
    ~mi_pipeline() {
        for (PipeJpegSlot& sl : jslot)
            if (sl.pending && sl.jpeg.done) hipEventSynchronize(sl.jpeg.done);
    }
};
using PipeJpegSlot = mi_pipeline::PipeJpegSlot;

struct mi_iris {
    mi_model model;
    int in_w = 0, in_h = 0;
    DeviceBuf d_in, d_roi, d_size, d_pad, d_flip, d_contour, d_iris, d_img, d_geom, d_sizes_b;
    int sizes_N = 0, sizes_w = 0, sizes_h = 0;
    OneShot one;
};

extern "C" {

const char* mi_last_error(void) { return g_error.c_str(); }
const char* mi_version(void) { return "mi_face 0.1 (gfx950)"; }

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ L0
int mi_model_load_bytes(const uint8_t* tflite, size_t nbytes, int device, mi_model** out) {
    return guarded([&] {
        require(tflite && nbytes && out, "null argument");
        auto h = std::make_unique<mi_model>();
        h->m = std::make_unique<mi::Model>(tflite, nbytes, device);
        *out = h.release();
    });
}

int mi_model_load_file(const char* path, int device, mi_model** out) {
    return guarded([&] {
        require(path && out, "null argument");
        auto bytes = read_file(path);
        auto h = std::make_unique<mi_model>();
        h->m = std::make_unique<mi::Model>(bytes.data(), bytes.size(), device);
        *out = h.release();
    });
}

void mi_model_free(mi_model* m) { delete m; }

int mi_model_input_dims(const mi_model* m, int dims[4]) {
    return guarded([&] {
        require(m && dims, "null argument");
        auto d = m->m->input_dims();
        for (int i = 0; i < 4; i++) dims[i] = d[i];
    });
}

int mi_model_num_outputs(const mi_model* m) { return m ? m->m->num_outputs() : MI_EINVAL; }

int mi_model_output_dims(const mi_model* m, int index, int dims[4], int* rank) {
    return guarded([&] {
        require(m && dims, "null argument");
        require(index >= 0 && index < m->m->num_outputs(), "output index out of range");
        const auto& d = m->m->output_dims(index);
        for (int i = 0; i < 4; i++) dims[i] = i < static_cast<int>(d.size()) ? d[i] : 1;
        if (rank) *rank = static_cast<int>(d.size());
    });
}

size_t mi_model_output_elems(const mi_model* m, int index) {
    if (!m || index < 0 || index >= m->m->num_outputs()) return 0;
    return m->m->output_elems(index);
}

int mi_model_run(mi_model* m, const float* in, int batch, float* const* outs, int mem, void* stream) {
    return guarded([&] {
        require(m && in && outs, "null argument");
        require(batch > 0, "batch must be positive");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::hip_check(hipSetDevice(m->m->device()), "hipSetDevice");
        Use use(*m, stream ? static_cast<hipStream_t>(stream) : m->m->stream());
        m->m->run(in, batch, outs, mem, static_cast<hipStream_t>(stream));
    });
}

int mi_model_debug_tensor(mi_model* m, int tensor_index, int frame, float* dst, size_t cap, size_t* n) {
    return guarded([&] {
        require(m && dst, "null argument");
        std::lock_guard<std::mutex> lock(m->mu);
        size_t got = m->m->debug_tensor(tensor_index, frame, dst, cap);
        if (n) *n = got;
    });
}

size_t mi_model_describe(const mi_model* m, char* buf, size_t cap) {
    if (!m) return 0;
    std::string s = m->m->describe();
    if (buf && cap) {
        size_t k = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return s.size() + 1;
}

size_t mi_plan_describe(const uint8_t* tflite, size_t nbytes, int fuse_level, char* buf, size_t cap) {
    size_t need = 0;
    int rc = guarded([&] {
        require(tflite && nbytes, "null argument");
        std::string s = mi::build_plan(mi::parse_tflite(tflite, nbytes), fuse_level).describe();
        if (buf && cap) {
            size_t k = std::min(cap - 1, s.size());
            std::memcpy(buf, s.data(), k);
            buf[k] = 0;
        }
        need = s.size() + 1;
    });
    return rc == MI_OK ? need : 0;
}

int mi_model_set_option(mi_model* m, const char* key, int value) {
    return guarded([&] {
        require(m && key, "null argument");
        std::lock_guard<std::mutex> lock(m->mu);
        m->m->set_option(key, value);
    });
}

int mi_model_get_option(mi_model* m, const char* key, int* value) {
    return guarded([&] {
        require(m && key && value, "null argument");
        std::lock_guard<std::mutex> lock(m->mu);
        *value = m->m->get_option(key);
    });
}

size_t mi_model_profile(mi_model* m, const float* in_device, int batch, int reps, char* buf, size_t cap) {
    size_t need = 0;
    int rc = guarded([&] {
        require(m && in_device, "null argument");
        require(batch > 0 && reps > 0, "batch and reps must be positive");
        std::lock_guard<std::mutex> lock(m->mu);
        auto stats = m->m->profile(in_device, batch, reps, nullptr);
        std::string s = "[";
        char line[512];
        for (size_t i = 0; i < stats.size(); i++) {
            std::snprintf(line, sizeof line, "%s{\"kernel\": \"%s\", \"shape\": \"%s\", \"ms\": %.6f, \"bytes\": %.0f, \"macs\": %.0f}", i ? ", " : "",
                          stats[i].kernel.c_str(), stats[i].detail.c_str(), stats[i].ms, stats[i].bytes, stats[i].macs);
            s += line;
        }
        s += "]";
        if (buf && cap) {
            size_t k = std::min(cap - 1, s.size());
            std::memcpy(buf, s.data(), k);
            buf[k] = 0;
        }
        need = s.size() + 1;
    });
    return rc == MI_OK ? need : 0;
}

int mi_model_single_launch_workgroups(mi_model* m, int batch) {
    int n = 0;
    int rc = guarded([&] {
        require(m && m->m, "null argument");
        require(batch > 0, "batch must be positive");
        std::lock_guard<std::mutex> g(m->mu);
        mi::hip_check(hipSetDevice(m->m->device()), "hipSetDevice");
        n = m->m->band_workgroups(batch);
    });
    return rc == MI_OK ? n : rc;
}

int mi_model_plan_stats(const mi_model* m, double* bytes_per_frame, double* macs_per_frame, int* launches) {
    return guarded([&] {
        require(m, "null argument");
        if (bytes_per_frame) *bytes_per_frame = m->m->plan().bytes_per_frame;
        if (macs_per_frame) *macs_per_frame = m->m->plan().macs_per_frame;
        if (launches) {
            int n = 0;
            for (const auto& nd : m->m->plan().nodes) n += !(nd.kind == mi::Node::Reshape || nd.kind == mi::Node::Concat);
            *launches = n;
        }
    });
}

// ------------------------------------------------------------------------------------------------ FaceDetection
static void fd_finish_create(mi_fd* h, int kind) {
    mi::SsdOptions opts;
    if (!mi::ssd_options_for(kind, &opts)) throw ApiError(MI_EINVAL, "unsupported model type");
    h->kind = kind;
    auto d = h->model.m->input_dims();
    h->in_h = d[1];
    h->in_w = d[2];
    if (d[3] != 3) throw ApiError(MI_EMODEL, "detector input must have 3 channels");
    h->anchors = mi::ssd_generate_anchors(opts);
    h->n_anchors = static_cast<int>(h->anchors.size() / 2);
    if (h->model.m->num_outputs() < 2) throw ApiError(MI_EMODEL, "detector must have two outputs");
    const auto& rb = h->model.m->output_dims(0);
    if (rb.size() != 3 || rb[1] != h->n_anchors || rb[2] != 16 || static_cast<int>(h->model.m->output_elems(1)) != h->n_anchors)
        throw ApiError(MI_EMODEL, "detector outputs do not match the SSD anchor layout");
    if (h->in_h != opts.input_h || h->in_w != opts.input_w) throw ApiError(MI_EMODEL, "detector input size does not match SSD options");
    mi::hip_check(hipMalloc(reinterpret_cast<void**>(&h->d_anchors), h->anchors.size() * sizeof(float)), "hipMalloc anchors");
    mi::hip_check(hipMemcpy(h->d_anchors, h->anchors.data(), h->anchors.size() * sizeof(float), hipMemcpyHostToDevice), "upload anchors");
}

int mi_fd_create_from_bytes(int kind, const uint8_t* tflite, size_t nbytes, int device, mi_fd** out) {
    return guarded([&] {
        require(tflite && nbytes && out, "null argument");
        auto h = std::make_unique<mi_fd>();
        h->model.m = std::make_unique<mi::Model>(tflite, nbytes, device);
        fd_finish_create(h.get(), kind);
        *out = h.release();
    });
}

int mi_fd_create(int kind, const char* model_dir, int device, mi_fd** out) {
    return guarded([&] {
        require(out, "null argument");
        const char* file = mi::model_file_for(kind);
        if (!file) throw ApiError(MI_EINVAL, "unsupported model type");
        std::string dir = model_dir ? model_dir : "./models";  // face_detection.rs:157-161
        auto bytes = read_file(dir + "/" + file);
        auto h = std::make_unique<mi_fd>();
        h->model.m = std::make_unique<mi::Model>(bytes.data(), bytes.size(), device);
        fd_finish_create(h.get(), kind);
        *out = h.release();
    });
}

void mi_fd_free(mi_fd* h) { delete h; }
mi_model* mi_fd_model(mi_fd* h) { return h ? &h->model : nullptr; }

int mi_fd_input_size(const mi_fd* h, int* width, int* height) {
    if (!h) return MI_EINVAL;
    if (width) *width = h->in_w;
    if (height) *height = h->in_h;
    return MI_OK;
}
int mi_fd_num_anchors(const mi_fd* h) { return h ? h->n_anchors : MI_EINVAL; }
int mi_fd_anchors(const mi_fd* h, float* out_xy, int cap) {
    if (!h || !out_xy) return MI_EINVAL;
    int n = std::min(cap, h->n_anchors);
    std::memcpy(out_xy, h->anchors.data(), static_cast<size_t>(n) * 2 * sizeof(float));
    return n;
}

// shared tail: raw device outputs -> detections
static void fd_post(mi_fd* h, const float* d_boxes, const float* d_scores, int batch, const double* padding, mi_detection* out,
                    int cap, int* counts, int mem, hipStream_t s, mi::RectD* d_face_rois = nullptr, int* d_face_valid = nullptr, int image_w = 0,
                    int image_h = 0) {
    mi::PostArgs a;
    if (d_face_rois) {   // (device results only: the batched pipeline) zeros behind the last detection and faces[0]'s ROI from the same launch
        a.zero_rest = 1;
        a.face_rois = d_face_rois; a.face_valid = d_face_valid; a.image_w = image_w; a.image_h = image_h;
    }
    a.raw_boxes = d_boxes;
    a.raw_scores = d_scores;
    a.anchors = h->d_anchors;
    a.B = batch;
    a.N = h->n_anchors;
    a.cap = cap;
    a.scale = static_cast<float>(h->in_h);  // decode_boxes(raw_boxes, input_shape[1] as f32) — face_detection.rs:259
    if (mem == MI_MEM_DEVICE) {
        a.padding = padding;
        a.out = reinterpret_cast<float*>(out);
        a.counts = counts;
        int rc = mi::launch_postprocess(a, s);
        if (rc) throw std::runtime_error(std::string("postprocess kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        return;
    }
    const double* d_pad = nullptr;
    if (padding) {
        d_pad = static_cast<const double*>(h->d_pad.get(sizeof(double) * 4 * batch));
        mi::hip_check(hipMemcpyAsync(const_cast<double*>(d_pad), padding, sizeof(double) * 4 * batch, hipMemcpyHostToDevice, s), "H2D padding");
    }
    a.padding = d_pad;
    a.out = static_cast<float*>(h->d_out.get(sizeof(mi_detection) * static_cast<size_t>(cap) * batch));
    a.counts = static_cast<int*>(h->d_counts.get(sizeof(int) * batch));
    // the kernel stores min(count, cap) detections per frame: the rest of the caller's host buffer reads as zeros, not as
    // whatever the staging buffer held before
    mi::hip_check(hipMemsetAsync(a.out, 0, sizeof(mi_detection) * static_cast<size_t>(cap) * batch, s), "hipMemsetAsync");
    int rc = mi::launch_postprocess(a, s);
    if (rc) throw std::runtime_error(std::string("postprocess kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    mi::hip_check(hipMemcpyAsync(out, a.out, sizeof(mi_detection) * static_cast<size_t>(cap) * batch, hipMemcpyDeviceToHost, s), "D2H detections");
    mi::hip_check(hipMemcpyAsync(counts, a.counts, sizeof(int) * batch, hipMemcpyDeviceToHost, s), "D2H counts");
    mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    for (int b = 0; b < batch; b++)
        if (counts[b] < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
}

int mi_fd_infer_tensor(mi_fd* h, const float* in, int batch, const double* padding, mi_detection* out, int cap_per_frame,
                       int* counts, int mem, void* stream) {
    return guarded([&] {
        require(h && in && out && counts, "null argument");
        require(batch > 0 && cap_per_frame > 0, "batch and cap_per_frame must be positive");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const float* din = in;
        if (mem == MI_MEM_HOST) {
            size_t bytes = m.input_elems() * sizeof(float) * batch;
            din = static_cast<const float*>(h->d_in.get(bytes));
            mi::hip_check(hipMemcpyAsync(const_cast<float*>(din), in, bytes, hipMemcpyHostToDevice, s), "H2D input");
        }
        m.run_device(din, batch, s);
        fd_post(h, m.output_device(0), m.output_device(1), batch, padding, out, cap_per_frame, counts, mem, s);
        if (mem == MI_MEM_DEVICE && !stream) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}

int mi_fd_postprocess(mi_fd* h, const float* raw_boxes, const float* raw_scores, int batch, const double* padding,
                      mi_detection* out, int cap_per_frame, int* counts, int mem, void* stream) {
    return guarded([&] {
        require(h && raw_boxes && raw_scores && out && counts, "null argument");
        require(batch > 0 && cap_per_frame > 0, "batch and cap_per_frame must be positive");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const float *db = raw_boxes, *ds = raw_scores;
        if (mem == MI_MEM_HOST) {
            size_t nb = sizeof(float) * 16 * h->n_anchors * batch, ns = sizeof(float) * h->n_anchors * batch;
            char* p = static_cast<char*>(h->d_in.get(nb + ns));
            mi::hip_check(hipMemcpyAsync(p, raw_boxes, nb, hipMemcpyHostToDevice, s), "H2D boxes");
            mi::hip_check(hipMemcpyAsync(p + nb, raw_scores, ns, hipMemcpyHostToDevice, s), "H2D scores");
            db = reinterpret_cast<const float*>(p);
            ds = reinterpret_cast<const float*>(p + nb);
        }
        fd_post(h, db, ds, batch, padding, out, cap_per_frame, counts, mem, s);
        if (mem == MI_MEM_DEVICE && !stream) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}

int mi_fd_infer_image(mi_fd* h, const uint8_t* rgb, int width, int height, int stride, const mi_rect* roi, mi_detection* out,
                      int cap, int* count) {
    return guarded([&] {
        require(h && rgb && out && count, "null argument");
        require(width > 0 && height > 0 && stride >= 3 * width && cap > 0, "bad image geometry");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        float* d_t = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float)));
        const size_t nout = sizeof(mi_detection) * static_cast<size_t>(cap);
        OneShot& o = h->one;
        o.reserve(kOneResults + nout);
        // image_to_tensor(image, roi, (w,h), keep_aspect_ratio = true, (-1,1), flip = false) — face_detection.rs:219
        mi::image_to_tensor_enqueue(rgb, width, height, stride, roi, h->in_w, h->in_h, true, -1.0, 1.0, false, d_t, o.h<double>(kOnePad),
                                    static_cast<uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height)), s);
        BandClaim claim(m, 1);
        auto network = [&](bool one_shot) {
            std::memset(o.h<char>(kOneResults), 0, nout);  // slots beyond the count read as zeros
            m.run_device(d_t, 1, s, one_shot);
            fd_post(h, m.output_device(0), m.output_device(1), 1, o.d<double>(kOnePad), o.d<mi_detection>(kOneResults), cap, o.d<int>(kOneCount),
                    MI_MEM_DEVICE, s);
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        };
        network(claim.ok);
        if (claim.ok && m.band_failed()) {  // the single launch did not get its CUs (someone else's kernels hold them): the batched plan
            for (JpegSlot& js : h->jslot)       // (the flag is the handle's: a streamed picture in flight may have raised it)
                if (js.pending && js.band) js.suspect = true;
            network(false);
        }
        *count = *o.h<int>(kOneCount);
        std::memcpy(out, o.h<char>(kOneResults), nout);
        if (*count < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
    });
}

// ---- convert_image_to_mat + FaceDetection::infer for a stream of encoded pictures (utils.rs:8-21, lib.rs:20-24), two slots.
// submit: Huffman decoding on the calling thread, straight into the slot's pinned block — the GPU is still busy with the previous picture's
// kernels — then, all asynchronous: coefficients to the device (slot's copy stream), dequantise + IDCT, up-sampling + colour conversion
// (the RGB picture stays in HBM), image_to_tensor, the network (single-launch plan when its CUs are free), post-processing into the slot's
// pinned mapped block.  collect: waits for the slot, repeats the network on the batched plan if its single launch gave up, hands the results out.
static void fd_jpeg_network(mi_fd* h, JpegSlot& sl, bool one_shot, hipStream_t s) {
    mi::Model& m = *h->model.m;
    OneShot& o = sl.one;
    float* d_t = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float)));
    mi::image_to_tensor_enqueue_device(static_cast<const uint8_t*>(sl.d_rgb.p), sl.f.width, sl.f.height, 3 * sl.f.width, nullptr, h->in_w, h->in_h, true, -1.0, 1.0, false,
                                       d_t, o.h<double>(kOnePad), s);
    std::memset(o.h<char>(kOneResults), 0, sizeof(mi_detection) * static_cast<size_t>(sl.cap));
    m.run_device(d_t, 1, s, one_shot);
    fd_post(h, m.output_device(0), m.output_device(1), 1, o.d<double>(kOnePad), o.d<mi_detection>(kOneResults), sl.cap, o.d<int>(kOneCount), MI_MEM_DEVICE, s);
}

int mi_fd_submit_jpeg(mi_fd* h, int slot, const uint8_t* bytes, size_t nbytes, int cap) {
    return guarded([&] {
        require(h && bytes, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        require(cap > 0, "cap must be positive");
        JpegSlot& sl = h->jslot[slot];
        if (sl.pending) throw ApiError(MI_EINVAL, "slot holds a picture that has not been collected");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        // (1) the serial part, on this thread, while the device works on the other slot's picture: the coefficients go straight into pinned memory
        sl.f.coef.ctx = &sl;
        sl.f.coef.provide = [](size_t count, void* ctx) -> int16_t* {
            JpegSlot& q = *static_cast<JpegSlot*>(ctx);
            const size_t need = count * sizeof(int16_t) + sizeof q.f.qt;
            if (need > q.h_coef_cap) {
                if (q.h_coef) hipHostFree(q.h_coef);
                q.h_coef = nullptr;
                q.h_coef_cap = 0;
                mi::hip_check(hipHostMalloc(&q.h_coef, need + need / 4, hipHostMallocDefault), "hipHostMalloc coefficients");
                q.h_coef_cap = need + need / 4;
            }
            return static_cast<int16_t*>(q.h_coef);
        };
        try {
            mi::jpeg_entropy_decode(bytes, nbytes, &sl.f);
        } catch (const std::runtime_error& e) {
            throw ApiError(MI_EINVAL, e.what());
        }
        const mi::JpegFrame& f = sl.f;
        const size_t coef_bytes = f.coef.size() * sizeof(int16_t);
        char* h_qt = static_cast<char*>(sl.h_coef) + coef_bytes;
        std::memcpy(h_qt, f.qt, sizeof f.qt);
        // (2) everything else is queued
        if (!sl.copy) {
            mi::hip_check(hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking), "hipStreamCreate");
            mi::hip_check(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming), "hipEventCreate");
            mi::hip_check(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming), "hipEventCreate");
        }
        auto* d_coef = static_cast<int16_t*>(sl.d_coef.get(coef_bytes + sizeof f.qt));
        auto* d_qt = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_coef) + coef_bytes);
        auto* d_planes = static_cast<uint8_t*>(sl.d_planes.get(mi::jpeg_plane_bytes(f)));
        auto* d_rgb = static_cast<uint8_t*>(sl.d_rgb.get(static_cast<size_t>(3) * f.width * f.height));
        const size_t nout = sizeof(mi_detection) * static_cast<size_t>(cap);
        sl.one.reserve(kOneResults + nout);
        sl.cap = cap;
        mi::hip_check(hipMemcpyAsync(d_coef, sl.h_coef, coef_bytes + sizeof f.qt, hipMemcpyHostToDevice, sl.copy), "H2D coefficients");
        mi::hip_check(hipEventRecord(sl.copied, sl.copy), "hipEventRecord");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        mi::hip_check(hipStreamWaitEvent(s, sl.copied, 0), "hipStreamWaitEvent");
        int rc = mi::launch_jpeg_idct(f, d_coef, d_qt, d_planes, s);
        if (rc == 0) rc = mi::launch_jpeg_color(f, d_planes, d_rgb, s);
        if (rc) throw std::runtime_error(std::string("jpeg kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        // the single-launch plan's CUs: one claim for the handle while any of its streamed pictures is in flight
        if (h->jclaim_users == 0) h->jclaim.reset(new BandClaim(m, 1));
        h->jclaim_users++;
        sl.claimed = true;
        sl.band = h->jclaim && h->jclaim->ok;
        sl.suspect = false;
        fd_jpeg_network(h, sl, sl.band, s);
        mi::hip_check(hipEventRecord(sl.done, s), "hipEventRecord");
        sl.pending = true;
    });
}

int mi_fd_collect_jpeg(mi_fd* h, int slot, mi_detection* out, int cap, int* count, int* width, int* height) {
    return guarded([&] {
        require(h && out && count, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        JpegSlot& sl = h->jslot[slot];
        if (!sl.pending) throw ApiError(MI_EINVAL, "nothing was submitted to this slot");
        require(cap >= sl.cap, "cap is smaller than the one the picture was submitted with");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        mi::hip_check(hipEventSynchronize(sl.done), "hipEventSynchronize");
        {
            hipStream_t s = m.stream();
            Use use(h->model, s);
            auto release = [&] {
                if (sl.claimed && --h->jclaim_users == 0) h->jclaim.reset();
                sl.claimed = false;
                sl.pending = false;
            };
            try {
                if (sl.band) {
                    // a single launch that gave up leaves void results; the flag is the handle's: with the other slot's picture in flight behind
                    // this one it may be that launch's — both are repeated on the batched plan (from the RGB pictures, which are still in HBM)
                    if (m.band_failed()) {
                        sl.suspect = true;
                        JpegSlot& other = h->jslot[1 - slot];
                        if (other.pending && other.band) other.suspect = true;
                    }
                    if (sl.suspect) {
                        fd_jpeg_network(h, sl, false, s);
                        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
                    }
                }
            } catch (...) {
                release();
                throw;
            }
            release();
        }
        const OneShot& o = sl.one;
        *count = *o.h<int>(kOneCount);
        std::memcpy(out, o.h<char>(kOneResults), sizeof(mi_detection) * static_cast<size_t>(sl.cap));
        if (cap > sl.cap) std::memset(out + sl.cap, 0, sizeof(mi_detection) * static_cast<size_t>(cap - sl.cap));
        if (width) *width = sl.f.width;
        if (height) *height = sl.f.height;
        if (*count < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
    });
}

// image_to_tensor(frame, roi, (w,h), keep_aspect_ratio = true, (-1,1)) for a batch of frames already in device memory, then the
// network and the post-processing into device buffers (face_detection.rs:219-265).  d_rois: device [batch] or null.
static void fd_images_device(mi_fd* h, const uint8_t* d_frames, int batch, int width, int height, int stride, const mi::RectD* d_rois,
                             double* d_pad, mi_detection* d_out, int cap, int* d_counts, hipStream_t s) {
    mi::Model& m = *h->model.m;
    static const bool no_u8_stem = getenv("MI_NO_U8_STEM") != nullptr;  // tuning aid: always through the f32 tensor
    if (!d_rois && width == h->in_w && height == h->in_h && !no_u8_stem && m.takes_u8_input()) {
        // Frames of exactly the network's input size: image_to_tensor is the u8 -> f32 loop alone (transform.rs:292-301, no warp, no
        // letterbox, padding 0), and that loop is a 256-entry table inside the first convolution's tile load: no f32 copy of the frames
        if (!h->d_lut) {
            float lut[256];
            const double k = 1.0 - (-1.0);
            for (int v = 0; v < 256; v++) lut[v] = static_cast<float>(static_cast<double>(v) * k / 255.0 + (-1.0));
            mi::hip_check(hipMalloc(reinterpret_cast<void**>(&h->d_lut), sizeof lut), "hipMalloc lut");
            mi::hip_check(hipMemcpy(h->d_lut, lut, sizeof lut, hipMemcpyHostToDevice), "H2D lut");
        }
        m.run_device_u8(d_frames, static_cast<long>(stride) * height, stride, h->d_lut, batch, s);
        fd_post(h, m.output_device(0), m.output_device(1), batch, nullptr, d_out, cap, d_counts, MI_MEM_DEVICE, s);
        return;
    }
    mi::PreItems it{};
    it.frames = d_frames; it.frame_bytes = static_cast<long>(stride) * height; it.width = width; it.height = height; it.stride = stride;
    it.rois = d_rois; it.items_per_frame = 1; it.N = batch; it.out_w = h->in_w; it.out_h = h->in_h; it.keep_aspect = 1;
    it.range_min = -1.0; it.range_max = 1.0;
    auto* d_geom = static_cast<mi::PreGeom*>(h->d_geom.get(sizeof(mi::PreGeom) * batch));
    float* d_in = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float) * batch));
    mi::launch_pre_geom(it, d_geom, d_pad, s);
    mi::launch_pre_tensor(it, d_geom, d_in, s);
    m.run_device(d_in, batch, s);
    fd_post(h, m.output_device(0), m.output_device(1), batch, d_pad, d_out, cap, d_counts, MI_MEM_DEVICE, s);
}

static size_t frames_bytes(int batch, int width, int height, int stride) {
    // the last frame's last row owns 3 * width bytes, not a whole stride
    return static_cast<size_t>(stride) * height * (batch - 1) + static_cast<size_t>(stride) * (height - 1) + static_cast<size_t>(3) * width;
}

int mi_fd_infer_images(mi_fd* h, const uint8_t* frames, int batch, int width, int height, int stride, const mi_rect* rois, mi_detection* out,
                       int cap_per_frame, int* counts, int mem, void* stream) {
    return guarded([&] {
        require(h && frames && out && counts, "null argument");
        require(batch > 0 && cap_per_frame > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const uint8_t* d_frames = frames;
        const mi::RectD* d_rois = nullptr;
        static_assert(sizeof(mi::RectD) == sizeof(mi_rect), "mi_rect and RectD must have one layout");
        if (mem == MI_MEM_HOST) {
            d_frames = static_cast<const uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<uint8_t*>(d_frames), frames, frames_bytes(batch, width, height, stride), hipMemcpyHostToDevice, s), "H2D frames");
            if (rois) {
                d_rois = static_cast<const mi::RectD*>(h->d_roi.get(sizeof(mi_rect) * batch));
                mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(d_rois), rois, sizeof(mi_rect) * batch, hipMemcpyHostToDevice, s), "H2D rois");
            }
        } else {
            d_rois = reinterpret_cast<const mi::RectD*>(rois);
        }
        double* d_pad = static_cast<double*>(h->d_pad.get(sizeof(double) * 4 * batch));
        if (mem == MI_MEM_DEVICE) {
            fd_images_device(h, d_frames, batch, width, height, stride, d_rois, d_pad, out, cap_per_frame, counts, s);
            if (!stream) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
            return;
        }
        const size_t nout = sizeof(mi_detection) * static_cast<size_t>(cap_per_frame) * batch;
        auto* d_out = static_cast<mi_detection*>(h->d_out.get(nout));
        int* d_counts = static_cast<int*>(h->d_counts.get(sizeof(int) * batch));
        mi::hip_check(hipMemsetAsync(d_out, 0, nout, s), "hipMemsetAsync");  // unused slots of the caller's buffer read as zeros
        fd_images_device(h, d_frames, batch, width, height, stride, d_rois, d_pad, d_out, cap_per_frame, d_counts, s);
        mi::hip_check(hipMemcpyAsync(out, d_out, nout, hipMemcpyDeviceToHost, s), "D2H detections");
        mi::hip_check(hipMemcpyAsync(counts, d_counts, sizeof(int) * batch, hipMemcpyDeviceToHost, s), "D2H counts");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        for (int b = 0; b < batch; b++)
            if (counts[b] < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
    });
}

int mi_fd_submit_images(mi_fd* h, int slot, const uint8_t* frames, int batch, int width, int height, int stride, int cap_per_frame) {
    return guarded([&] {
        require(h && frames, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        require(batch > 0 && cap_per_frame > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        FdSlot& sl = h->slot[slot];
        if (sl.pending) throw ApiError(MI_EINVAL, "slot still holds an uncollected batch (call mi_fd_collect first)");
        if (!sl.copy) mi::hip_check(hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking), "hipStreamCreate");
        if (!sl.copied) mi::hip_check(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming), "hipEventCreate");
        if (!sl.done) mi::hip_check(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming), "hipEventCreate");
        const size_t nout = sizeof(mi_detection) * static_cast<size_t>(cap_per_frame) * batch, ncnt = sizeof(int) * batch;
        if (nout + ncnt > sl.h_cap) {
            if (sl.h_out) hipHostFree(sl.h_out);
            sl.h_out = nullptr; sl.h_cap = 0;
            // (coherent: the kernel's stores must be visible to the host once `done` has fired, without a copy)
            mi::hip_check(hipHostMalloc(&sl.h_out, nout + ncnt, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
            sl.h_cap = nout + ncnt;
        }
        // frames: host -> device on the slot's copy stream (asynchronous when `frames` is pinned memory: mi_host_alloc);
        // everything else on the handle's stream, behind the copy.  The post-processing kernel writes detections and counts
        // straight into the slot's pinned host block (mapped into the device's address space): a hipMemcpyAsync of the results
        // blocked the host for whole batches on this runtime (measured: every third call, 7 ms).
        auto* d_frames = static_cast<uint8_t*>(sl.d_frames.get(static_cast<size_t>(stride) * height * batch));
        mi::hip_check(hipMemcpyAsync(d_frames, frames, frames_bytes(batch, width, height, stride), hipMemcpyHostToDevice, sl.copy), "H2D frames");
        mi::hip_check(hipEventRecord(sl.copied, sl.copy), "hipEventRecord");
        mi::hip_check(hipStreamWaitEvent(s, sl.copied, 0), "hipStreamWaitEvent");
        std::memset(sl.h_out, 0, nout + ncnt);  // slots beyond a frame's count read as zeros (the slot was collected: nothing in flight writes here)
        void* mapped = nullptr;
        mi::hip_check(hipHostGetDevicePointer(&mapped, sl.h_out, 0), "hipHostGetDevicePointer");
        double* d_pad = static_cast<double*>(sl.d_pad.get(sizeof(double) * 4 * batch));
        try {
            fd_images_device(h, d_frames, batch, width, height, stride, nullptr, d_pad, static_cast<mi_detection*>(mapped), cap_per_frame,
                             reinterpret_cast<int*>(static_cast<char*>(mapped) + nout), s);
        } catch (...) {
            (void)hipStreamSynchronize(sl.copy);   //  the queued copy may still be reading the caller's frames: not after we return
            throw;
        }
        mi::hip_check(hipEventRecord(sl.done, s), "hipEventRecord");
        sl.batch = batch; sl.cap = cap_per_frame; sl.pending = true;
    });
}

int mi_fd_collect(mi_fd* h, int slot, mi_detection* out, int* counts) {
    return guarded([&] {
        require(h && out && counts, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        FdSlot& sl = h->slot[slot];
        {
            std::lock_guard<std::mutex> g(h->model.mu);
            if (!sl.pending) throw ApiError(MI_EINVAL, "nothing was submitted to this slot");
        }
        mi::hip_check(hipSetDevice(h->model.m->device()), "hipSetDevice");
        mi::hip_check(hipEventSynchronize(sl.done), "hipEventSynchronize");
        std::lock_guard<std::mutex> g(h->model.mu);
        const size_t nout = sizeof(mi_detection) * static_cast<size_t>(sl.cap) * sl.batch;
        std::memcpy(out, sl.h_out, nout);
        std::memcpy(counts, static_cast<char*>(sl.h_out) + nout, sizeof(int) * sl.batch);
        sl.pending = false;
        for (int b = 0; b < sl.batch; b++)
            if (counts[b] < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
    });
}

int mi_host_alloc(size_t bytes, void** out) {
    return guarded([&] {
        require(out && bytes > 0, "null argument");
        *out = nullptr;
        mi::hip_check(hipHostMalloc(out, bytes, hipHostMallocDefault), "hipHostMalloc");
    });
}
void mi_host_free(void* p) {
    if (p) hipHostFree(p);
}

// ------------------------------------------------------------------------------------------------ FaceLandmark
static void fl_finish_create(mi_fl* h) {
    auto d = h->model.m->input_dims();
    h->in_h = d[1];
    h->in_w = d[2];
    // face_landmark.rs:241-247: last dim of outputs()[0] must hold NUM_DIMS * NUM_LANDMARKS = 1404 values
    if (h->model.m->num_outputs() < 2 || h->model.m->output_dims(0).back() < 3 * MI_NUM_FACE_LANDMARKS)
        throw ApiError(MI_EMODEL, "incompatible model: mesh output narrower than 1404");
}

int mi_fl_create_from_bytes(const uint8_t* tflite, size_t nbytes, int device, mi_fl** out) {
    return guarded([&] {
        require(tflite && nbytes && out, "null argument");
        auto h = std::make_unique<mi_fl>();
        h->model.m = std::make_unique<mi::Model>(tflite, nbytes, device);
        fl_finish_create(h.get());
        *out = h.release();
    });
}

int mi_fl_create(const char* model_path, int device, mi_fl** out) {
    return guarded([&] {
        require(out, "null argument");
        auto bytes = read_file(model_path ? model_path : "./models/face_landmark.tflite");  // face_landmark.rs:211-215
        auto h = std::make_unique<mi_fl>();
        h->model.m = std::make_unique<mi::Model>(bytes.data(), bytes.size(), device);
        fl_finish_create(h.get());
        *out = h.release();
    });
}

void mi_fl_free(mi_fl* h) { delete h; }
mi_model* mi_fl_model(mi_fl* h) { return h ? &h->model : nullptr; }

int mi_fl_infer_tensor(mi_fl* h, const float* in, int batch, const mi_rect* rois, const int* image_sizes, float* landmarks,
                       int* present, float* raw_flags, int mem, void* stream) {
    return guarded([&] {
        require(h && in && landmarks && present, "null argument");
        require(batch > 0, "batch must be positive");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        require(!rois || image_sizes, "image_sizes is required when rois are given");
        static_assert(sizeof(mi_rect) == sizeof(mi::RectD), "mi_rect layout");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const size_t lm_bytes = sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * batch;
        mi::ProjArgs a;
        a.B = batch;
        a.n = MI_NUM_FACE_LANDMARKS;
        a.tensor_w = h->in_w;
        a.tensor_h = h->in_h;
        const float* din = in;
        if (mem == MI_MEM_HOST) {
            size_t bytes = m.input_elems() * sizeof(float) * batch;
            din = static_cast<const float*>(h->d_in.get(bytes));
            mi::hip_check(hipMemcpyAsync(const_cast<float*>(din), in, bytes, hipMemcpyHostToDevice, s), "H2D input");
            if (rois) {
                a.roi = static_cast<const mi::RectD*>(h->d_roi.get(sizeof(mi_rect) * batch));
                mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(a.roi), rois, sizeof(mi_rect) * batch, hipMemcpyHostToDevice, s), "H2D rois");
                a.image_size = static_cast<const int*>(h->d_size.get(sizeof(int) * 2 * batch));
                mi::hip_check(hipMemcpyAsync(const_cast<int*>(a.image_size), image_sizes, sizeof(int) * 2 * batch, hipMemcpyHostToDevice, s), "H2D sizes");
            }
            a.out = static_cast<float*>(h->d_lm.get(lm_bytes));
            a.present = static_cast<int*>(h->d_present.get(sizeof(int) * batch));
            a.raw_flag_out = static_cast<float*>(h->d_flag.get(sizeof(float) * batch));
        } else {
            a.roi = reinterpret_cast<const mi::RectD*>(rois);
            a.image_size = image_sizes;
            a.out = landmarks;
            a.present = present;
            a.raw_flag_out = raw_flags;
        }
        m.run_device(din, batch, s);
        a.raw = m.output_device(0);
        a.raw_fs = static_cast<long>(m.output_elems(0));
        a.flag = m.output_device(1) + (m.output_elems(1) - 1);  // `flatten[flatten.len() - 1]` — face_landmark.rs:292-293
        a.flag_fs = static_cast<long>(m.output_elems(1));
        int rc = mi::launch_project(a, s);
        if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        if (mem == MI_MEM_HOST) {
            mi::hip_check(hipMemcpyAsync(landmarks, a.out, lm_bytes, hipMemcpyDeviceToHost, s), "D2H landmarks");
            mi::hip_check(hipMemcpyAsync(present, a.present, sizeof(int) * batch, hipMemcpyDeviceToHost, s), "D2H present");
            if (raw_flags) mi::hip_check(hipMemcpyAsync(raw_flags, a.raw_flag_out, sizeof(float) * batch, hipMemcpyDeviceToHost, s), "D2H flags");
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        } else if (!stream) {
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
    });
}

// (w, h) of the source image for every item of a batch of equally sized frames, on the device; re-uploaded only when the batch geometry changes
static int* batch_image_sizes(DeviceBuf& buf, int& have_N, int& have_w, int& have_h, int N, int width, int height, hipStream_t s) {
    int* d = static_cast<int*>(buf.get(sizeof(int) * 2 * N));
    if (have_N != N || have_w != width || have_h != height) {
        std::vector<int> hs(2 * static_cast<size_t>(N));
        for (int i = 0; i < N; i++) { hs[2 * i] = width; hs[2 * i + 1] = height; }
        have_N = 0;
        mi::hip_check(hipMemcpyAsync(d, hs.data(), hs.size() * sizeof(int), hipMemcpyHostToDevice, s), "H2D sizes");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");  // hs is a host temporary
        have_N = N; have_w = width; have_h = height;
    }
    return d;
}

// FaceLandmark::infer(&Mat, Option<Rect>) for a batch of (frame, ROI) items (face_landmark.rs:232-306): image_to_tensor(frame, roi,
// (192,192), keep_aspect_ratio = false, (0,1)) on the device, network, face flag, project_landmarks.  Item i reads frame
// i / items_per_frame.  Only u8 frames and ROIs cross the bus (the f32 crops of mi_fl_infer_tensor are 442 KB per ROI).
int mi_fl_infer_images(mi_fl* h, const uint8_t* frames, int batch, int width, int height, int stride, const mi_rect* rois, int items_per_frame,
                       float* landmarks, int* present, float* raw_flags, int mem, void* stream) {
    return guarded([&] {
        require(h && frames && landmarks && present, "null argument");
        require(batch > 0 && items_per_frame > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        require(rois || items_per_frame == 1, "several items per frame need their ROIs");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        require(static_cast<long long>(batch) * items_per_frame <= (1 << 24), "too many items");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const int N = batch * items_per_frame;
        const uint8_t* d_frames = frames;
        const mi::RectD* d_rois = reinterpret_cast<const mi::RectD*>(rois);
        if (mem == MI_MEM_HOST) {
            d_frames = static_cast<const uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<uint8_t*>(d_frames), frames, frames_bytes(batch, width, height, stride), hipMemcpyHostToDevice, s), "H2D frames");
            if (rois) {
                d_rois = static_cast<const mi::RectD*>(h->d_roi.get(sizeof(mi_rect) * N));
                mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(d_rois), rois, sizeof(mi_rect) * N, hipMemcpyHostToDevice, s), "H2D rois");
            }
        }
        mi::PreItems it{};
        it.frames = d_frames; it.frame_bytes = static_cast<long>(stride) * height; it.width = width; it.height = height; it.stride = stride;
        it.rois = d_rois; it.items_per_frame = items_per_frame; it.N = N; it.out_w = h->in_w; it.out_h = h->in_h; it.keep_aspect = 0;
        it.range_min = 0.0; it.range_max = 1.0;
        auto* d_geom = static_cast<mi::PreGeom*>(h->d_geom.get(sizeof(mi::PreGeom) * N));
        float* d_in = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float) * N));
        mi::launch_pre_geom(it, d_geom, nullptr, s);
        mi::launch_pre_tensor(it, d_geom, d_in, s);
        const size_t lm_bytes = sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * N;
        mi::ProjArgs a;
        a.B = N; a.n = MI_NUM_FACE_LANDMARKS; a.tensor_w = h->in_w; a.tensor_h = h->in_h;
        a.roi = d_rois;
        a.image_size = d_rois ? batch_image_sizes(h->d_sizes_b, h->sizes_N, h->sizes_w, h->sizes_h, N, width, height, s) : nullptr;
        if (mem == MI_MEM_HOST) {
            a.out = static_cast<float*>(h->d_lm.get(lm_bytes));
            a.present = static_cast<int*>(h->d_present.get(sizeof(int) * N));
            a.raw_flag_out = static_cast<float*>(h->d_flag.get(sizeof(float) * N));
        } else {
            a.out = landmarks; a.present = present; a.raw_flag_out = raw_flags;
        }
        m.run_device(d_in, N, s);
        a.raw = m.output_device(0);
        a.raw_fs = static_cast<long>(m.output_elems(0));
        a.flag = m.output_device(1) + (m.output_elems(1) - 1);
        a.flag_fs = static_cast<long>(m.output_elems(1));
        int rc = mi::launch_project(a, s);
        if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        if (mem == MI_MEM_HOST) {
            mi::hip_check(hipMemcpyAsync(landmarks, a.out, lm_bytes, hipMemcpyDeviceToHost, s), "D2H landmarks");
            mi::hip_check(hipMemcpyAsync(present, a.present, sizeof(int) * N, hipMemcpyDeviceToHost, s), "D2H present");
            if (raw_flags) mi::hip_check(hipMemcpyAsync(raw_flags, a.raw_flag_out, sizeof(float) * N, hipMemcpyDeviceToHost, s), "D2H flags");
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        } else if (!stream) {
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
    });
}

// The same for a continuous host feed, split in two like the detector's (mi_fd_submit_images / mi_fd_collect): submit queues the H2D copy
// of the frames on the slot's own stream, then warp + network + projection on the handle's stream behind it, and returns; the projection
// kernel writes into the slot's pinned, mapped host block; collect waits for the slot and hands the results out.
int mi_fl_submit_images(mi_fl* h, int slot, const uint8_t* frames, int batch, int width, int height, int stride, const mi_rect* rois, int items_per_frame) {
    return guarded([&] {
        require(h && frames, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        require(batch > 0 && items_per_frame > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        require(rois || items_per_frame == 1, "several items per frame need their ROIs");
        require(static_cast<long long>(batch) * items_per_frame <= (1 << 24), "too many items");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        FlSlot& sl = h->slot[slot];
        if (sl.pending) throw ApiError(MI_EINVAL, "slot still holds an uncollected batch (call mi_fl_collect first)");
        if (!sl.copy) mi::hip_check(hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking), "hipStreamCreate");
        if (!sl.copied) mi::hip_check(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming), "hipEventCreate");
        if (!sl.done) mi::hip_check(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming), "hipEventCreate");
        const int N = batch * items_per_frame;
        const size_t lm_bytes = sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * N, total = lm_bytes + (sizeof(int) + sizeof(float)) * N;
        if (total > sl.h_cap) {
            if (sl.h_out) hipHostFree(sl.h_out);
            sl.h_out = nullptr; sl.h_cap = 0;
            mi::hip_check(hipHostMalloc(&sl.h_out, total, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
            sl.h_cap = total;
        }
        const size_t frame_bytes = static_cast<size_t>(stride) * height;
        auto* d_frames = static_cast<uint8_t*>(sl.d_frames.get(frame_bytes * batch));
        mi::hip_check(hipMemcpyAsync(d_frames, frames, frames_bytes(batch, width, height, stride), hipMemcpyHostToDevice, sl.copy), "H2D frames");
        const mi::RectD* d_rois = nullptr;
        if (rois) {   // (the ROIs are small and usually pageable: the runtime stages them before the call returns)
            d_rois = static_cast<const mi::RectD*>(sl.d_rois.get(sizeof(mi_rect) * N));
            mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(d_rois), rois, sizeof(mi_rect) * N, hipMemcpyHostToDevice, sl.copy), "H2D rois");
        }
        mi::hip_check(hipEventRecord(sl.copied, sl.copy), "hipEventRecord");
        void* mapped = nullptr;
        mi::hip_check(hipHostGetDevicePointer(&mapped, sl.h_out, 0), "hipHostGetDevicePointer");
        try {
            const int* d_sizes = d_rois ? batch_image_sizes(h->d_sizes_b, h->sizes_N, h->sizes_w, h->sizes_h, N, width, height, s) : nullptr;
            mi::hip_check(hipStreamWaitEvent(s, sl.copied, 0), "hipStreamWaitEvent");
            mi::PreItems it{};
            it.frames = d_frames; it.frame_bytes = static_cast<long>(frame_bytes); it.width = width; it.height = height; it.stride = stride;
            it.rois = d_rois; it.items_per_frame = items_per_frame; it.N = N; it.out_w = h->in_w; it.out_h = h->in_h; it.keep_aspect = 0;
            it.range_min = 0.0; it.range_max = 1.0;
            auto* d_geom = static_cast<mi::PreGeom*>(h->d_geom.get(sizeof(mi::PreGeom) * N));
            float* d_in = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float) * N));
            mi::launch_pre_geom(it, d_geom, nullptr, s);
            mi::launch_pre_tensor(it, d_geom, d_in, s);
            mi::ProjArgs a;
            a.B = N; a.n = MI_NUM_FACE_LANDMARKS; a.tensor_w = h->in_w; a.tensor_h = h->in_h;
            a.roi = d_rois; a.image_size = d_sizes;
            a.out = static_cast<float*>(mapped);
            a.present = reinterpret_cast<int*>(static_cast<char*>(mapped) + lm_bytes);
            a.raw_flag_out = reinterpret_cast<float*>(static_cast<char*>(mapped) + lm_bytes + sizeof(int) * N);
            m.run_device(d_in, N, s);
            a.raw = m.output_device(0);
            a.raw_fs = static_cast<long>(m.output_elems(0));
            a.flag = m.output_device(1) + (m.output_elems(1) - 1);
            a.flag_fs = static_cast<long>(m.output_elems(1));
            int rc = mi::launch_project(a, s);
            if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        } catch (...) {
            (void)hipStreamSynchronize(sl.copy);   // the queued copy may still be reading the caller's frames
            throw;
        }
        mi::hip_check(hipEventRecord(sl.done, s), "hipEventRecord");
        sl.N = N; sl.pending = true;
    });
}

int mi_fl_collect(mi_fl* h, int slot, float* landmarks, int* present, float* raw_flags) {
    return guarded([&] {
        require(h && landmarks && present, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        FlSlot& sl = h->slot[slot];
        {
            std::lock_guard<std::mutex> g(h->model.mu);
            if (!sl.pending) throw ApiError(MI_EINVAL, "nothing was submitted to this slot");
        }
        mi::hip_check(hipSetDevice(h->model.m->device()), "hipSetDevice");
        mi::hip_check(hipEventSynchronize(sl.done), "hipEventSynchronize");
        std::lock_guard<std::mutex> g(h->model.mu);
        const size_t lm_bytes = sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * sl.N;
        std::memcpy(landmarks, sl.h_out, lm_bytes);
        std::memcpy(present, static_cast<char*>(sl.h_out) + lm_bytes, sizeof(int) * sl.N);
        if (raw_flags) std::memcpy(raw_flags, static_cast<char*>(sl.h_out) + lm_bytes + sizeof(int) * sl.N, sizeof(float) * sl.N);
        sl.pending = false;
    });
}

int mi_fl_infer_image(mi_fl* h, const uint8_t* rgb, int width, int height, int stride, const mi_rect* roi, mi_landmark* out,
                      int cap, int* count) {
    return guarded([&] {
        require(h && rgb && out && count, "null argument");
        require(width > 0 && height > 0 && stride >= 3 * width, "bad image geometry");
        require(cap >= MI_NUM_FACE_LANDMARKS, "output capacity must be at least 468");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        float* d_t = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float)));
        double pad[4];
        OneShot& o = h->one;
        const size_t lm_bytes = sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS;
        o.reserve(kOneResults + lm_bytes);
        // image_to_tensor(image, roi, (192,192), keep_aspect_ratio = false, (0,1), false) — face_landmark.rs:250
        mi::image_to_tensor_enqueue(rgb, width, height, stride, roi, h->in_w, h->in_h, false, 0.0, 1.0, false, d_t, pad,
                                    static_cast<uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height)), s);
        mi::ProjArgs a;
        a.B = 1; a.n = MI_NUM_FACE_LANDMARKS; a.tensor_w = h->in_w; a.tensor_h = h->in_h;
        if (roi) {
            std::memcpy(o.h<char>(kOneRoi), roi, sizeof(mi_rect));
            o.h<int>(kOneSize)[0] = width;
            o.h<int>(kOneSize)[1] = height;
            a.roi = o.d<mi::RectD>(kOneRoi);
            a.image_size = o.d<int>(kOneSize);
        }
        a.out = o.d<float>(kOneResults);
        a.present = o.d<int>(kOneCount);
        BandClaim claim(m, 1);
        auto network = [&](bool one_shot) {
            m.run_device(d_t, 1, s, one_shot);
            a.raw = m.output_device(0);
            a.raw_fs = static_cast<long>(m.output_elems(0));
            a.flag = m.output_device(1) + (m.output_elems(1) - 1);
            a.flag_fs = static_cast<long>(m.output_elems(1));
            int rc = mi::launch_project(a, s);
            if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        };
        network(claim.ok);
        if (claim.ok && m.band_failed()) network(false);
        const float* lm = o.h<float>(kOneResults);
        const int present = *o.h<int>(kOneCount);
        *count = 0;
        if (present) {
            for (int i = 0; i < MI_NUM_FACE_LANDMARKS; i++) out[i] = mi_landmark{lm[3 * i], lm[3 * i + 1], lm[3 * i + 2]};
            *count = MI_NUM_FACE_LANDMARKS;
        }
    });
}

// ------------------------------------------------------------------------------------------------ IrisLandmark
static void iris_finish_create(mi_iris* h) {
    auto d = h->model.m->input_dims();
    h->in_h = d[1];
    h->in_w = d[2];
    // iris_landmark.rs:172-184
    if (h->model.m->num_outputs() < 2 || h->model.m->output_dims(0).back() != 3 * MI_NUM_EYE_LANDMARKS ||
        h->model.m->output_dims(1).back() != 3 * MI_NUM_IRIS_LANDMARKS)
        throw ApiError(MI_EMODEL, "unexpected number of eye landmarks");
}

int mi_iris_create_from_bytes(const uint8_t* tflite, size_t nbytes, int device, mi_iris** out) {
    return guarded([&] {
        require(tflite && nbytes && out, "null argument");
        auto h = std::make_unique<mi_iris>();
        h->model.m = std::make_unique<mi::Model>(tflite, nbytes, device);
        iris_finish_create(h.get());
        *out = h.release();
    });
}

int mi_iris_create(const char* model_path, int device, mi_iris** out) {
    return guarded([&] {
        require(out, "null argument");
        auto bytes = read_file(model_path ? model_path : "./models/iris_landmark.tflite");  // iris_landmark.rs:145-149
        auto h = std::make_unique<mi_iris>();
        h->model.m = std::make_unique<mi::Model>(bytes.data(), bytes.size(), device);
        iris_finish_create(h.get());
        *out = h.release();
    });
}

void mi_iris_free(mi_iris* h) { delete h; }
mi_model* mi_iris_model(mi_iris* h) { return h ? &h->model : nullptr; }

static void iris_project(mi_iris* h, int batch, const mi::RectD* d_roi, const int* d_size, const double* d_pad, const int* d_flip,
                         float* d_contour, float* d_iris, hipStream_t s) {
    mi::Model& m = *h->model.m;
    // both outputs (71 contour + 5 iris landmarks per eye) in one launch
    mi::ProjArgs a;
    a.B = batch;
    a.n = MI_NUM_EYE_LANDMARKS;
    a.n2 = MI_NUM_IRIS_LANDMARKS;
    a.tensor_w = h->in_w;
    a.tensor_h = h->in_h;
    a.roi = d_roi;
    a.image_size = d_size;
    a.padding = d_pad;
    a.flip = d_flip;
    a.raw = m.output_device(0);
    a.raw_fs = static_cast<long>(m.output_elems(0));
    a.out = d_contour;
    a.raw2 = m.output_device(1);
    a.raw2_fs = static_cast<long>(m.output_elems(1));
    a.out2 = d_iris;
    int rc = mi::launch_project(a, s);
    if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
}

int mi_iris_infer_tensor(mi_iris* h, const float* in, int batch, const mi_rect* rois, const int* image_sizes, const double* padding,
                         const int* is_right_eye, float* contour, float* iris, int mem, void* stream) {
    return guarded([&] {
        require(h && in && contour && iris, "null argument");
        require(batch > 0, "batch must be positive");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        require(!rois || image_sizes, "image_sizes is required when rois are given");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const size_t cb = sizeof(float) * 3 * MI_NUM_EYE_LANDMARKS * batch, ib = sizeof(float) * 3 * MI_NUM_IRIS_LANDMARKS * batch;
        if (mem == MI_MEM_DEVICE) {
            m.run_device(in, batch, s);
            iris_project(h, batch, reinterpret_cast<const mi::RectD*>(rois), image_sizes, padding, is_right_eye, contour, iris, s);
            if (!stream) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
            return;
        }
        size_t bytes = m.input_elems() * sizeof(float) * batch;
        float* din = static_cast<float*>(h->d_in.get(bytes));
        mi::hip_check(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, s), "H2D input");
        const mi::RectD* d_roi = nullptr;
        const int *d_size = nullptr, *d_flip = nullptr;
        const double* d_pad = nullptr;
        if (rois) {
            d_roi = static_cast<const mi::RectD*>(h->d_roi.get(sizeof(mi_rect) * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(d_roi), rois, sizeof(mi_rect) * batch, hipMemcpyHostToDevice, s), "H2D rois");
            d_size = static_cast<const int*>(h->d_size.get(sizeof(int) * 2 * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<int*>(d_size), image_sizes, sizeof(int) * 2 * batch, hipMemcpyHostToDevice, s), "H2D sizes");
        }
        if (padding) {
            d_pad = static_cast<const double*>(h->d_pad.get(sizeof(double) * 4 * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<double*>(d_pad), padding, sizeof(double) * 4 * batch, hipMemcpyHostToDevice, s), "H2D padding");
        }
        if (is_right_eye) {
            d_flip = static_cast<const int*>(h->d_flip.get(sizeof(int) * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<int*>(d_flip), is_right_eye, sizeof(int) * batch, hipMemcpyHostToDevice, s), "H2D flip");
        }
        float* dc = static_cast<float*>(h->d_contour.get(cb));
        float* di = static_cast<float*>(h->d_iris.get(ib));
        m.run_device(din, batch, s);
        iris_project(h, batch, d_roi, d_size, d_pad, d_flip, dc, di, s);
        mi::hip_check(hipMemcpyAsync(contour, dc, cb, hipMemcpyDeviceToHost, s), "D2H contour");
        mi::hip_check(hipMemcpyAsync(iris, di, ib, hipMemcpyDeviceToHost, s), "D2H iris");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}

// IrisLandmark::infer(&Mat, Option<Rect>, Option<bool>) for a batch of (frame, eye ROI) items (iris_landmark.rs:158-248):
// image_to_tensor(frame, roi, (64,64), keep_aspect_ratio = true, (0,1), flip = is_right_eye) on the device, network, both
// project_landmarks calls.  Item i reads frame i / items_per_frame (2 = the two eyes of a face).
int mi_iris_infer_images(mi_iris* h, const uint8_t* frames, int batch, int width, int height, int stride, const mi_rect* rois, const int* is_right_eye,
                         int items_per_frame, float* contour, float* iris, int mem, void* stream) {
    return guarded([&] {
        require(h && frames && contour && iris, "null argument");
        require(batch > 0 && items_per_frame > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        require(rois || items_per_frame == 1, "several items per frame need their ROIs");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        require(static_cast<long long>(batch) * items_per_frame <= (1 << 24), "too many items");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m.stream();
        Use use(h->model, s);
        const int N = batch * items_per_frame;
        const uint8_t* d_frames = frames;
        const mi::RectD* d_rois = reinterpret_cast<const mi::RectD*>(rois);
        const int* d_flip = is_right_eye;
        if (mem == MI_MEM_HOST) {
            d_frames = static_cast<const uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height * batch));
            mi::hip_check(hipMemcpyAsync(const_cast<uint8_t*>(d_frames), frames, frames_bytes(batch, width, height, stride), hipMemcpyHostToDevice, s), "H2D frames");
            if (rois) {
                d_rois = static_cast<const mi::RectD*>(h->d_roi.get(sizeof(mi_rect) * N));
                mi::hip_check(hipMemcpyAsync(const_cast<mi::RectD*>(d_rois), rois, sizeof(mi_rect) * N, hipMemcpyHostToDevice, s), "H2D rois");
            }
            if (is_right_eye) {
                d_flip = static_cast<const int*>(h->d_flip.get(sizeof(int) * N));
                mi::hip_check(hipMemcpyAsync(const_cast<int*>(d_flip), is_right_eye, sizeof(int) * N, hipMemcpyHostToDevice, s), "H2D flip");
            }
        }
        mi::PreItems it{};
        it.frames = d_frames; it.frame_bytes = static_cast<long>(stride) * height; it.width = width; it.height = height; it.stride = stride;
        it.rois = d_rois; it.flip = d_flip; it.items_per_frame = items_per_frame; it.N = N; it.out_w = h->in_w; it.out_h = h->in_h; it.keep_aspect = 1;
        it.range_min = 0.0; it.range_max = 1.0;
        auto* d_geom = static_cast<mi::PreGeom*>(h->d_geom.get(sizeof(mi::PreGeom) * N));
        double* d_pad = static_cast<double*>(h->d_pad.get(sizeof(double) * 4 * N));
        float* d_in = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float) * N));
        mi::launch_pre_geom(it, d_geom, d_pad, s);
        mi::launch_pre_tensor(it, d_geom, d_in, s);
        const int* d_size = d_rois ? batch_image_sizes(h->d_sizes_b, h->sizes_N, h->sizes_w, h->sizes_h, N, width, height, s) : nullptr;
        const size_t cb = sizeof(float) * 3 * MI_NUM_EYE_LANDMARKS * N, ib = sizeof(float) * 3 * MI_NUM_IRIS_LANDMARKS * N;
        float* dc = mem == MI_MEM_DEVICE ? contour : static_cast<float*>(h->d_contour.get(cb));
        float* di = mem == MI_MEM_DEVICE ? iris : static_cast<float*>(h->d_iris.get(ib));
        m.run_device(d_in, N, s);
        iris_project(h, N, d_rois, d_size, d_pad, d_flip, dc, di, s);
        if (mem == MI_MEM_HOST) {
            mi::hip_check(hipMemcpyAsync(contour, dc, cb, hipMemcpyDeviceToHost, s), "D2H contour");
            mi::hip_check(hipMemcpyAsync(iris, di, ib, hipMemcpyDeviceToHost, s), "D2H iris");
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        } else if (!stream) {
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
    });
}

int mi_iris_infer_image(mi_iris* h, const uint8_t* rgb, int width, int height, int stride, const mi_rect* roi, int is_right_eye,
                        mi_landmark* contour71, mi_landmark* iris5) {
    return guarded([&] {
        require(h && rgb && contour71 && iris5, "null argument");
        require(width > 0 && height > 0 && stride >= 3 * width, "bad image geometry");
        mi::Model& m = *h->model.m;
        mi::hip_check(hipSetDevice(m.device()), "hipSetDevice");
        hipStream_t s = m.stream();
        Use use(h->model, s);
        float* d_t = static_cast<float*>(h->d_in.get(m.input_elems() * sizeof(float)));
        OneShot& o = h->one;
        constexpr size_t kContour = kOneResults, kIris = kOneResults + sizeof(float) * 3 * MI_NUM_EYE_LANDMARKS;
        o.reserve(kIris + sizeof(float) * 3 * MI_NUM_IRIS_LANDMARKS);
        // image_to_tensor(image, roi, (64,64), keep_aspect_ratio = true, (0,1), is_right_eye) — iris_landmark.rs:188-189
        mi::image_to_tensor_enqueue(rgb, width, height, stride, roi, h->in_w, h->in_h, true, 0.0, 1.0, is_right_eye != 0, d_t,
                                    o.h<double>(kOnePad), static_cast<uint8_t*>(h->d_img.get(static_cast<size_t>(stride) * height)), s);
        const mi::RectD* d_roi = nullptr;
        const int* d_size = nullptr;
        if (roi) {
            std::memcpy(o.h<char>(kOneRoi), roi, sizeof(mi_rect));
            o.h<int>(kOneSize)[0] = width;
            o.h<int>(kOneSize)[1] = height;
            d_roi = o.d<mi::RectD>(kOneRoi);
            d_size = o.d<int>(kOneSize);
        }
        *o.h<int>(kOneFlip) = is_right_eye != 0;
        BandClaim claim(m, 1);
        auto network = [&](bool one_shot) {
            m.run_device(d_t, 1, s, one_shot);
            iris_project(h, 1, d_roi, d_size, o.d<double>(kOnePad), o.d<int>(kOneFlip), o.d<float>(kContour), o.d<float>(kIris), s);
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        };
        network(claim.ok);
        if (claim.ok && m.band_failed()) network(false);
        const float *c = o.h<float>(kContour), *ir = o.h<float>(kIris);
        for (int i = 0; i < MI_NUM_EYE_LANDMARKS; i++) contour71[i] = mi_landmark{c[3 * i], c[3 * i + 1], c[3 * i + 2]};
        for (int i = 0; i < MI_NUM_IRIS_LANDMARKS; i++) iris5[i] = mi_landmark{ir[3 * i], ir[3 * i + 1], ir[3 * i + 2]};
    });
}

// ------------------------------------------------------------------------------------------------ batched pipeline
int mi_pipeline_create(int fd_kind, const char* model_dir, int device, mi_pipeline** out) {
    return guarded([&] {
        require(out, "null argument");
        std::string dir = model_dir ? model_dir : "./models";
        auto p = std::make_unique<mi_pipeline>();
        mi_fd* fd = nullptr;
        mi_fl* fl = nullptr;
        mi_iris* ir = nullptr;
        if (int rc = mi_fd_create(fd_kind, dir.c_str(), device, &fd)) throw ApiError(rc, g_error);
        p->fd.reset(fd);
        if (int rc = mi_fl_create((dir + "/face_landmark.tflite").c_str(), device, &fl)) throw ApiError(rc, g_error);
        p->fl.reset(fl);
        if (int rc = mi_iris_create((dir + "/iris_landmark.tflite").c_str(), device, &ir)) throw ApiError(rc, g_error);
        p->iris.reset(ir);
        *out = p.release();
    });
}

int mi_pipeline_create_from_bytes(int fd_kind, const uint8_t* fd_tflite, size_t fd_nbytes, const uint8_t* fl_tflite, size_t fl_nbytes,
                                  const uint8_t* iris_tflite, size_t iris_nbytes, int device, mi_pipeline** out) {
    return guarded([&] {
        require(out && fd_tflite && fl_tflite && iris_tflite && fd_nbytes && fl_nbytes && iris_nbytes, "null argument");
        auto p = std::make_unique<mi_pipeline>();
        mi_fd* fd = nullptr;
        mi_fl* fl = nullptr;
        mi_iris* ir = nullptr;
        if (int rc = mi_fd_create_from_bytes(fd_kind, fd_tflite, fd_nbytes, device, &fd)) throw ApiError(rc, g_error);
        p->fd.reset(fd);
        if (int rc = mi_fl_create_from_bytes(fl_tflite, fl_nbytes, device, &fl)) throw ApiError(rc, g_error);
        p->fl.reset(fl);
        if (int rc = mi_iris_create_from_bytes(iris_tflite, iris_nbytes, device, &ir)) throw ApiError(rc, g_error);
        p->iris.reset(ir);
        *out = p.release();
    });
}

void mi_pipeline_free(mi_pipeline* p) { delete p; }
mi_model* mi_pipeline_model(mi_pipeline* p, int which) {
    if (!p) return nullptr;
    return which == 0 ? &p->fd->model : which == 1 ? &p->fl->model : which == 2 ? &p->iris->model : nullptr;
}

int mi_pipeline_set_option(mi_pipeline* p, const char* key, int value) {
    return guarded([&] {
        require(p && key, "null argument");
        std::lock_guard<std::mutex> l0(p->fd->model.mu), l1(p->fl->model.mu), l2(p->iris->model.mu);
        p->fd->model.m->set_option(key, value);
        p->fl->model.m->set_option(key, value);
        p->iris->model.m->set_option(key, value);
    });
}

namespace {
struct PipeOut {   // DEVICE pointers the stages of one pipeline pass write their results to
    mi_detection* faces;   // [B] top-1 detection
    int* counts;           // [B]
    float* lm;             // [B][468][3]
    int* present;          // [B]
    float* eyes;           // [B][2][76][3]
};
constexpr int kPipeCap = 4;
constexpr long kEyeFs = 3L * (MI_NUM_EYE_LANDMARKS + MI_NUM_IRIS_LANDMARKS);
struct PipeLayout {        // one block: faces | counts | landmarks | present | eyes
    size_t faces, counts, lm, present, eyes, bytes;
    explicit PipeLayout(int B) {
        auto up16 = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
        faces = 0;
        counts = up16(faces + sizeof(mi_detection) * B);
        lm = up16(counts + sizeof(int) * B);
        present = up16(lm + sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * B);
        eyes = up16(present + sizeof(int) * B);
        bytes = up16(eyes + sizeof(float) * kEyeFs * 2 * B);
    }
    PipeOut in(char* base) const {
        return PipeOut{reinterpret_cast<mi_detection*>(base + faces), reinterpret_cast<int*>(base + counts), reinterpret_cast<float*>(base + lm),
                       reinterpret_cast<int*>(base + present), reinterpret_cast<float*>(base + eyes)};
    }
};
}  // namespace

// One pass of the flow of lib.rs:24-40 over B frames that are in device memory: everything is queued on `s`, nothing is waited for (but the
// upload of the per-item image sizes when the batch geometry changes).  The caller holds the three handles (Use) and, for one_shot, the CUs.
static void pipeline_enqueue(mi_pipeline* p, const uint8_t* d_frames, int B, int width, int height, int stride, const PipeOut& o, bool one_shot, hipStream_t s) {
    mi::Model& fdm = *p->fd->model.m;
    mi::Model& flm = *p->fl->model.m;
    mi::Model& irm = *p->iris->model.m;
    const int cap = kPipeCap;
    const long eye_fs = kEyeFs;
    const size_t frame_bytes = static_cast<size_t>(stride) * height;
    auto* d_geom = static_cast<mi::PreGeom*>(p->geom.get(sizeof(mi::PreGeom) * 2 * B));
    // (w, h) of the source image of every ROI, for Rect::scaled in project_landmarks
    int* d_sizes = static_cast<int*>(p->sizes.get(sizeof(int) * 4 * B));
    if (p->sizes_B != B || p->sizes_w != width || p->sizes_h != height) {  // a host round trip only when the batch geometry changes
        std::vector<int> hs(4 * static_cast<size_t>(B));
        for (int i = 0; i < 2 * B; i++) { hs[2 * i] = width; hs[2 * i + 1] = height; }
        p->sizes_B = 0;
        mi::hip_check(hipMemcpyAsync(d_sizes, hs.data(), hs.size() * sizeof(int), hipMemcpyHostToDevice, s), "H2D sizes");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");  // hs is a host temporary
        p->sizes_B = B; p->sizes_w = width; p->sizes_h = height;
    }
    static const bool ptrace = getenv("MI_PIPE_TRACE") != nullptr;
    auto t_start = std::chrono::steady_clock::now();
    auto tr = [&](const char* what) { if (ptrace) std::fprintf(stderr, "  %-22s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count()); };
    // ---- 1. detector: image_to_tensor(frame, None, (w,h), keep_aspect = true, (-1,1)) -> net -> decode + NMS
    mi::PreItems it{};
    it.frames = d_frames; it.frame_bytes = static_cast<long>(frame_bytes); it.width = width; it.height = height; it.stride = stride;
    it.items_per_frame = 1; it.N = B; it.out_w = p->fd->in_w; it.out_h = p->fd->in_h; it.keep_aspect = 1;
    it.range_min = -1.0; it.range_max = 1.0;
    double* d_pad_det = static_cast<double*>(p->pad_det.get(sizeof(double) * 4 * B));
    float* d_in_det = static_cast<float*>(p->in_det.get(fdm.input_elems() * sizeof(float) * B));
    // (no ROI: the geometry is the same for every call on pictures of this size — no pre_geom launch, 9 us of a one-picture call)
    auto* d_geom_det = static_cast<mi::PreGeom*>(p->geom_det.get(sizeof(mi::PreGeom) * B));
    if (p->geom_B != B || p->geom_w != width || p->geom_h != height) {
        p->geom_B = 0;
        mi::upload_whole_image_geom(width, height, it.out_w, it.out_h, true, B, d_geom_det, d_pad_det, s);
        p->geom_B = B; p->geom_w = width; p->geom_h = height;
    }
    mi::launch_pre_tensor(it, d_geom_det, d_in_det, s);
    tr("pre det");
    fdm.run_device(d_in_det, B, s, one_shot);
    tr("det run_device");
    float* d_dets = static_cast<float*>(p->dets.get(sizeof(mi_detection) * cap * B));
    // ---- 2. faces[0] -> face_detection_to_roi -> image_to_tensor(frame, roi, (192,192), false, (0,1)) -> mesh net
    // (the post-processing launch writes zeros behind a frame's last detection — frames without a face report zeros — and faces[0]'s ROI)
    auto* d_roi_face = static_cast<mi::RectD*>(p->roi_face.get(sizeof(mi::RectD) * B));
    int* d_valid_face = static_cast<int*>(p->valid_face.get(sizeof(int) * B));
    fd_post(p->fd.get(), fdm.output_device(0), fdm.output_device(1), B, d_pad_det, reinterpret_cast<mi_detection*>(d_dets), cap, o.counts,
            MI_MEM_DEVICE, s, d_roi_face, d_valid_face, width, height);
    tr("fd_post");
    it.rois = d_roi_face; it.roi_valid = d_valid_face; it.out_w = p->fl->in_w; it.out_h = p->fl->in_h; it.keep_aspect = 0;
    it.range_min = 0.0; it.range_max = 1.0;
    float* d_in_lm = static_cast<float*>(p->in_lm.get(flm.input_elems() * sizeof(float) * B));
    mi::launch_pre_geom(it, d_geom, nullptr, s);
    mi::launch_pre_tensor(it, d_geom, d_in_lm, s);
    tr("pre mesh");
    flm.run_device(d_in_lm, B, s, one_shot);
    tr("mesh run_device");
    {
        mi::ProjArgs a;
        a.B = B; a.n = MI_NUM_FACE_LANDMARKS; a.tensor_w = p->fl->in_w; a.tensor_h = p->fl->in_h;
        a.roi = d_roi_face; a.image_size = d_sizes; a.gate = d_valid_face;
        a.raw = flm.output_device(0); a.raw_fs = static_cast<long>(flm.output_elems(0));
        a.flag = flm.output_device(1) + (flm.output_elems(1) - 1); a.flag_fs = static_cast<long>(flm.output_elems(1));
        a.out = o.lm; a.present = o.present;
        int rc = mi::launch_project(a, s);
        if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    }
    // ---- 3. iris_roi_from_face_landmarks -> image_to_tensor(frame, eye roi, (64,64), true, (0,1), flip = right eye) -> iris net
    auto* d_roi_eye = static_cast<mi::RectD*>(p->roi_eye.get(sizeof(mi::RectD) * 2 * B));
    int* d_valid_eye = static_cast<int*>(p->valid_eye.get(sizeof(int) * 2 * B));
    int* d_flip_eye = static_cast<int*>(p->flip_eye.get(sizeof(int) * 2 * B));
    mi::launch_iris_rois(o.lm, o.present, B, width, height, d_roi_eye, d_valid_eye, d_flip_eye, s);
    it.rois = d_roi_eye; it.roi_valid = d_valid_eye; it.flip = d_flip_eye; it.items_per_frame = 2; it.N = 2 * B;
    it.out_w = p->iris->in_w; it.out_h = p->iris->in_h; it.keep_aspect = 1;
    double* d_pad_eye = static_cast<double*>(p->pad_eye.get(sizeof(double) * 8 * B));
    float* d_in_eye = static_cast<float*>(p->in_eye.get(irm.input_elems() * sizeof(float) * 2 * B));
    mi::launch_pre_geom(it, d_geom, d_pad_eye, s);
    mi::launch_pre_tensor(it, d_geom, d_in_eye, s);
    tr("pre iris");
    irm.run_device(d_in_eye, 2 * B, s, one_shot);
    tr("iris run_device");
    {   // contour and iris landmarks of both eyes in one launch
        mi::ProjArgs a;
        a.B = 2 * B; a.n = MI_NUM_EYE_LANDMARKS; a.n2 = MI_NUM_IRIS_LANDMARKS; a.tensor_w = p->iris->in_w; a.tensor_h = p->iris->in_h;
        a.roi = d_roi_eye; a.image_size = d_sizes; a.padding = d_pad_eye; a.flip = d_flip_eye; a.gate = d_valid_eye;
        a.raw = irm.output_device(0); a.raw_fs = static_cast<long>(irm.output_elems(0));
        a.raw2 = irm.output_device(1); a.raw2_fs = static_cast<long>(irm.output_elems(1));
        a.out = o.eyes; a.out_fs = eye_fs;
        a.out2 = o.eyes + 3 * MI_NUM_EYE_LANDMARKS; a.out2_fs = eye_fs;
        int rc = mi::launch_project(a, s);
        if (rc) throw std::runtime_error(std::string("projection kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    }
    // ---- top-1 faces out (strided gather [B][cap][17] -> [B][17])
    tr("projections");
    mi::hip_check(hipMemcpy2DAsync(o.faces, sizeof(mi_detection), d_dets, sizeof(mi_detection) * cap, sizeof(mi_detection), B,
                                   hipMemcpyDeviceToDevice, s), "gather faces");
    tr("gather");
}

// the results of one pass, from the pinned image of its block into the caller's arrays
static void pipeline_hand_out(const OneShot& out, const PipeLayout& L, int B, mi_detection* faces, int* face_counts, float* landmarks, int* present, float* eyes) {
    std::memcpy(faces, out.h<char>(L.faces), sizeof(mi_detection) * B);
    std::memcpy(face_counts, out.h<char>(L.counts), sizeof(int) * B);
    std::memcpy(landmarks, out.h<char>(L.lm), sizeof(float) * 3 * MI_NUM_FACE_LANDMARKS * B);
    std::memcpy(present, out.h<char>(L.present), sizeof(int) * B);
    std::memcpy(eyes, out.h<char>(L.eyes), sizeof(float) * kEyeFs * 2 * B);
    for (int b = 0; b < B; b++) {
        if (face_counts[b] < 0) throw ApiError(MI_ERANGE, "letterbox scale is too small (reference asserts at transform.rs:121-122)");
        if (face_counts[b] == 0) std::memset(&faces[b], 0, sizeof(mi_detection));
    }
}

int mi_pipeline_run(mi_pipeline* p, const uint8_t* frames, int batch, int width, int height, int stride, mi_detection* faces,
                    int* face_counts, float* landmarks, int* present, float* eyes, int mem, void* stream) {
    return guarded([&] {
        require(p && frames && faces && face_counts && landmarks && present && eyes, "null argument");
        require(batch > 0 && width > 0 && height > 0 && stride >= 3 * width, "bad frame geometry");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::Model& fdm = *p->fd->model.m;
        mi::Model& flm = *p->fl->model.m;
        mi::Model& irm = *p->iris->model.m;
        mi::hip_check(hipSetDevice(fdm.device()), "hipSetDevice");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : fdm.stream();
        Use use_fd(p->fd->model, s), use_fl(p->fl->model, s), use_ir(p->iris->model, s);  // the pipeline's own three handles, fixed order
        const int B = batch;
        const PipeLayout L(B);
        char* d_results = nullptr;
        if (mem == MI_MEM_HOST) {
            d_results = static_cast<char*>(p->results.get(L.bytes));
            p->out.reserve(L.bytes);
        }
        const size_t frame_bytes = static_cast<size_t>(stride) * height;
        const uint8_t* d_frames = frames;
        if (mem == MI_MEM_HOST) {
            d_frames = static_cast<const uint8_t*>(p->frames.get(frame_bytes * B));
            // the last frame's last row owns 3 * width bytes, not a whole stride
            const size_t host_bytes = frame_bytes * (B - 1) + static_cast<size_t>(stride) * (height - 1) + static_cast<size_t>(3) * width;
            mi::hip_check(hipMemcpyAsync(const_cast<uint8_t*>(d_frames), frames, host_bytes, hipMemcpyHostToDevice, s), "H2D frames");
        }
        // A picture or two from host memory (lib.rs:24-40 on one image): the detector and the face mesh may take their single-launch plans —
        // this call holds the CUs until its synchronisation below, and repeats itself on the batched plan should such a launch have given up.
        std::unique_ptr<BandClaim> claim;
        if (mem == MI_MEM_HOST) claim = std::make_unique<BandClaim>(fdm.device(), std::max(std::max(fdm.band_workgroups(B), flm.band_workgroups(B)), irm.band_workgroups(2 * B)));
        const PipeOut o = mem == MI_MEM_DEVICE ? PipeOut{faces, face_counts, landmarks, present, eyes} : L.in(d_results);
        for (int attempt = 0; attempt < 2; attempt++) {
            const bool one_shot = claim && claim->ok && attempt == 0;
            pipeline_enqueue(p, d_frames, B, width, height, stride, o, one_shot, s);
            if (mem == MI_MEM_HOST) {
                mi::hip_check(hipMemcpyAsync(p->out.host, d_results, L.bytes, hipMemcpyDeviceToHost, s), "D2H results");
                mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
                if (one_shot && (static_cast<int>(fdm.band_failed()) | static_cast<int>(flm.band_failed()) | static_cast<int>(irm.band_failed()))) continue;
                pipeline_hand_out(p->out, L, B, faces, face_counts, landmarks, present, eyes);
            } else if (!stream) {
                mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
            }
            break;
        }
    });
}

// ---- the flow of lib.rs:18-40 for a STREAM of encoded pictures: convert_image_to_mat + FaceDetection::infer + FaceLandmark::infer + 2 x
// IrisLandmark::infer per picture, two slots (see mi_fd_submit_jpeg): the entropy decoder of picture n + 1 runs on the calling thread while the
// device works through picture n's four networks.
int mi_pipeline_submit_jpeg(mi_pipeline* p, int slot, const uint8_t* bytes, size_t nbytes) {
    return guarded([&] {
        require(p && bytes, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        PipeJpegSlot& sl = p->jslot[slot];
        if (sl.pending) throw ApiError(MI_EINVAL, "slot holds a picture that has not been collected");
        mi::Model& fdm = *p->fd->model.m;
        mi::Model& flm = *p->fl->model.m;
        mi::Model& irm = *p->iris->model.m;
        mi::hip_check(hipSetDevice(fdm.device()), "hipSetDevice");
        JpegSlot& js = sl.jpeg;
        js.f.coef.ctx = &js;
        js.f.coef.provide = [](size_t count, void* ctx) -> int16_t* {
            JpegSlot& q = *static_cast<JpegSlot*>(ctx);
            const size_t need = count * sizeof(int16_t) + sizeof q.f.qt;
            if (need > q.h_coef_cap) {
                if (q.h_coef) hipHostFree(q.h_coef);
                q.h_coef = nullptr;
                q.h_coef_cap = 0;
                mi::hip_check(hipHostMalloc(&q.h_coef, need + need / 4, hipHostMallocDefault), "hipHostMalloc coefficients");
                q.h_coef_cap = need + need / 4;
            }
            return static_cast<int16_t*>(q.h_coef);
        };
        try {
            mi::jpeg_entropy_decode(bytes, nbytes, &js.f);
        } catch (const std::runtime_error& e) {
            throw ApiError(MI_EINVAL, e.what());
        }
        const mi::JpegFrame& f = js.f;
        const size_t coef_bytes = f.coef.size() * sizeof(int16_t);
        std::memcpy(static_cast<char*>(js.h_coef) + coef_bytes, f.qt, sizeof f.qt);
        if (!js.copy) {
            mi::hip_check(hipStreamCreateWithFlags(&js.copy, hipStreamNonBlocking), "hipStreamCreate");
            mi::hip_check(hipEventCreateWithFlags(&js.copied, hipEventDisableTiming), "hipEventCreate");
            mi::hip_check(hipEventCreateWithFlags(&js.done, hipEventDisableTiming), "hipEventCreate");
        }
        auto* d_coef = static_cast<int16_t*>(js.d_coef.get(coef_bytes + sizeof f.qt));
        auto* d_qt = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_coef) + coef_bytes);
        auto* d_planes = static_cast<uint8_t*>(js.d_planes.get(mi::jpeg_plane_bytes(f)));
        auto* d_rgb = static_cast<uint8_t*>(js.d_rgb.get(static_cast<size_t>(3) * f.width * f.height));
        const PipeLayout L(1);
        char* d_results = static_cast<char*>(sl.results.get(L.bytes));
        sl.out.reserve(L.bytes);
        mi::hip_check(hipMemcpyAsync(d_coef, js.h_coef, coef_bytes + sizeof f.qt, hipMemcpyHostToDevice, js.copy), "H2D coefficients");
        mi::hip_check(hipEventRecord(js.copied, js.copy), "hipEventRecord");
        hipStream_t s = fdm.stream();
        Use use_fd(p->fd->model, s), use_fl(p->fl->model, s), use_ir(p->iris->model, s);
        mi::hip_check(hipStreamWaitEvent(s, js.copied, 0), "hipStreamWaitEvent");
        int rc = mi::launch_jpeg_idct(f, d_coef, d_qt, d_planes, s);
        if (rc == 0) rc = mi::launch_jpeg_color(f, d_planes, d_rgb, s);
        if (rc) throw std::runtime_error(std::string("jpeg kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        if (p->jclaim_users == 0)
            p->jclaim.reset(new BandClaim(fdm.device(), std::max(std::max(fdm.band_workgroups(1), flm.band_workgroups(1)), irm.band_workgroups(2))));
        p->jclaim_users++;
        sl.claimed = true;
        sl.band = p->jclaim && p->jclaim->ok;
        sl.suspect = false;
        pipeline_enqueue(p, d_rgb, 1, f.width, f.height, 3 * f.width, L.in(d_results), sl.band, s);
        mi::hip_check(hipMemcpyAsync(sl.out.host, d_results, L.bytes, hipMemcpyDeviceToHost, s), "D2H results");
        mi::hip_check(hipEventRecord(js.done, s), "hipEventRecord");
        sl.pending = true;
    });
}

int mi_pipeline_collect_jpeg(mi_pipeline* p, int slot, mi_detection* face, int* face_count, float* landmarks, int* present, float* eyes, int* width, int* height) {
    return guarded([&] {
        require(p && face && face_count && landmarks && present && eyes, "null argument");
        require(slot == 0 || slot == 1, "slot must be 0 or 1");
        PipeJpegSlot& sl = p->jslot[slot];
        if (!sl.pending) throw ApiError(MI_EINVAL, "nothing was submitted to this slot");
        mi::Model& fdm = *p->fd->model.m;
        mi::Model& flm = *p->fl->model.m;
        mi::Model& irm = *p->iris->model.m;
        mi::hip_check(hipSetDevice(fdm.device()), "hipSetDevice");
        JpegSlot& js = sl.jpeg;
        mi::hip_check(hipEventSynchronize(js.done), "hipEventSynchronize");
        const PipeLayout L(1);
        {
            hipStream_t s = fdm.stream();
            Use use_fd(p->fd->model, s), use_fl(p->fl->model, s), use_ir(p->iris->model, s);
            auto release = [&] {
                if (sl.claimed && --p->jclaim_users == 0) p->jclaim.reset();
                sl.claimed = false;
                sl.pending = false;
            };
            try {
                if (sl.band) {
                    // (the flags are the handles': with the other slot's picture in flight behind this one a raised flag may be that picture's — both are repeated)
                    if (static_cast<int>(fdm.band_failed()) | static_cast<int>(flm.band_failed()) | static_cast<int>(irm.band_failed())) {
                        sl.suspect = true;
                        PipeJpegSlot& other = p->jslot[1 - slot];
                        if (other.pending && other.band) other.suspect = true;
                    }
                    if (sl.suspect) {
                        char* d_results = static_cast<char*>(sl.results.p);
                        pipeline_enqueue(p, static_cast<const uint8_t*>(js.d_rgb.p), 1, js.f.width, js.f.height, 3 * js.f.width, L.in(d_results), false, s);
                        mi::hip_check(hipMemcpyAsync(sl.out.host, d_results, L.bytes, hipMemcpyDeviceToHost, s), "D2H results");
                        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
                    }
                }
            } catch (...) {
                release();
                throw;
            }
            release();
        }
        pipeline_hand_out(sl.out, L, 1, face, face_count, landmarks, present, eyes);
        if (width) *width = js.f.width;
        if (height) *height = js.f.height;
    });
}

// ------------------------------------------------------------------------------------------------ multi-GPU: weight broadcast
// The one exchange of the sharded path: the frozen .tflite bytes travel once from `root` to every rank over RCCL, then
// each rank builds its handles with mi_*_create_from_bytes.  A Rust host (north_star) has no torch.distributed: this entry speaks to
// librccl directly (loaded at call time, so the library itself does not depend on it).  Rendezvous = the ncclUniqueId in a file every
// rank of the node can read: `root` writes it (temporary name + rename), the others wait for it.
namespace {
struct UniqueId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value like the original
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace

// ------------------------------------------------------------------------------------------------ streams on distinct hardware queues
// HIP streams share the device's hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in the order they are first used; work on two streams of
// one queue runs strictly in order.  A host that keeps two batches in flight on two handles needs two streams that do NOT share a queue, and
// HIP has no call that tells: so the streams are created one by one and each candidate is TESTED — an idle wave of 40 us goes to every stream
// accepted so far and to the candidate at once; side by side the set takes 40 us, one after the other a multiple.
int mi_streams_create_distinct(int device, int n, void** streams) {
    return guarded([&] {
        require(streams != nullptr, "null argument");
        require(n >= 1 && n <= 4, "n must be 1 .. 4");
        mi::hip_check(hipSetDevice(device), "hipSetDevice");
        for (int k = 0; k < n; k++) streams[k] = nullptr;
        int* sink = nullptr;
        mi::hip_check(hipMalloc(reinterpret_cast<void**>(&sink), sizeof(int)), "hipMalloc");
        std::vector<hipStream_t> accepted, rejected;
        auto cleanup = [&](bool all) {
            for (hipStream_t s : rejected) hipStreamDestroy(s);
            if (all) for (hipStream_t s : accepted) hipStreamDestroy(s);
            hipFree(sink);
        };
        const long long ticks = 40 * 100;   // 40 us of the 100 MHz wall clock
        auto timed_together = [&](hipStream_t cand) {
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                for (hipStream_t s : accepted) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
                mi::hip_check(hipStreamSynchronize(cand), "hipStreamSynchronize");
                const auto t0 = std::chrono::steady_clock::now();
                for (hipStream_t s : accepted)
                    if (mi::launch_spin(ticks, 1 << 20, sink, s) != 0) throw std::runtime_error("spin kernel launch failed");
                if (mi::launch_spin(ticks, 1 << 20, sink, cand) != 0) throw std::runtime_error("spin kernel launch failed");
                for (hipStream_t s : accepted) mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
                mi::hip_check(hipStreamSynchronize(cand), "hipStreamSynchronize");
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            return best;
        };
        try {
            for (int tries = 0; tries < 16 && static_cast<int>(accepted.size()) < n; tries++) {
                hipStream_t c = nullptr;
                mi::hip_check(hipStreamCreateWithFlags(&c, hipStreamNonBlocking), "hipStreamCreate");
                if (accepted.empty()) {
                    if (mi::launch_spin(ticks, 1 << 20, sink, c) != 0) throw std::runtime_error("spin kernel launch failed");   // (first use: the stream gets its queue)
                    mi::hip_check(hipStreamSynchronize(c), "hipStreamSynchronize");
                    accepted.push_back(c);
                    continue;
                }
                // side by side: about one idle wave (40 us + launch and synchronisation cost); sharing a queue with any accepted stream: two or more
                if (timed_together(c) < 1.6 * 40.0 + 30.0 * static_cast<double>(accepted.size())) accepted.push_back(c);
                else rejected.push_back(c);
            }
        } catch (...) {
            cleanup(true);
            throw;
        }
        if (static_cast<int>(accepted.size()) < n) {
            cleanup(true);
            throw ApiError(MI_EDEVICE, "could not find that many streams on distinct hardware queues (GPU_MAX_HW_QUEUES?)");
        }
        for (int k = 0; k < n; k++) streams[k] = accepted[static_cast<size_t>(k)];
        cleanup(false);
    });
}

int mi_streams_destroy(int device, int n, void** streams) {
    return guarded([&] {
        require(streams != nullptr && n >= 0, "bad argument");
        mi::hip_check(hipSetDevice(device), "hipSetDevice");
        for (int k = 0; k < n; k++)
            if (streams[k]) {
                mi::hip_check(hipStreamSynchronize(static_cast<hipStream_t>(streams[k])), "hipStreamSynchronize");
                mi::hip_check(hipStreamDestroy(static_cast<hipStream_t>(streams[k])), "hipStreamDestroy");
                streams[k] = nullptr;
            }
    });
}

int mi_dist_broadcast_bytes(const char* id_path, int rank, int world, int root, int device, uint8_t* buf, size_t nbytes, int timeout_ms) {
    return guarded([&] {
        require(id_path && buf, "null argument");
        require(world >= 1 && rank >= 0 && rank < world && root >= 0 && root < world, "rank / root outside [0, world)");
        require(nbytes > 0 && nbytes <= (static_cast<size_t>(1) << 31), "nbytes must be in (0, 2 GiB]");
        mi::hip_check(hipSetDevice(device), "hipSetDevice");
        static std::mutex mu;
        static Rccl R;
        {
            std::lock_guard<std::mutex> g(mu);
            if (!R.lib) {
                void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
                if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
                if (!lib) throw ApiError(MI_EDEVICE, std::string("librccl is not loadable: ") + dlerror());
                auto sym = [&](const char* n) {
                    void* p = dlsym(lib, n);
                    if (!p) throw ApiError(MI_EDEVICE, std::string("librccl lacks ") + n);
                    return p;
                };
                R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
                R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
                R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(sym("ncclBroadcast"));
                R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
                R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
                R.lib = lib;
            }
        }
        auto nccl = [&](int rc, const char* what) {
            if (rc != 0) throw ApiError(MI_EDEVICE, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "rccl error"));
        };
        UniqueId id{};
        const std::string path = id_path, tmp = path + ".tmp";
        // rendezvous file: {"MIRV", world, root} + the 128-byte id.  A file of another job shape is never taken for this one's; a stale file of the
        // SAME shape (a crashed earlier run under a reused name) cannot be told from a root that was simply early — for that case ncclCommInitRank
        // runs under the caller's timeout below instead of blocking for ever.
        const uint32_t head[3] = {0x5652494du, static_cast<uint32_t>(world), static_cast<uint32_t>(root)};
        if (rank == root) {
            std::remove(path.c_str());   // whatever an earlier run left under this name
            nccl(R.GetUniqueId(&id), "ncclGetUniqueId");
            {
                std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
                if (!f || !f.write(reinterpret_cast<const char*>(head), sizeof head) || !f.write(id.internal, sizeof id.internal))
                    throw ApiError(MI_EIO, "cannot write the rendezvous file '" + tmp + "'");
            }
            if (std::rename(tmp.c_str(), path.c_str()) != 0) throw ApiError(MI_EIO, "cannot publish the rendezvous file '" + path + "'");
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                std::ifstream f(path, std::ios::binary);
                uint32_t got[3] = {0, 0, 0};
                if (f && f.read(reinterpret_cast<char*>(got), sizeof got) && got[0] == head[0] && got[1] == head[1] && got[2] == head[2] &&
                    f.read(id.internal, sizeof id.internal) && f.gcount() == static_cast<std::streamsize>(sizeof id.internal))
                    break;
                if (timeout_ms >= 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms))
                    throw ApiError(MI_EIO, "no rendezvous file '" + path + "' of this job (world " + std::to_string(world) + ", root " + std::to_string(root) + ") from the root rank within the timeout");
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        void* comm = nullptr;
        uint8_t* d = nullptr;
        hipStream_t s = nullptr;
        try {
            if (timeout_ms >= 0 && world > 1) {
                // bounded: a communicator that does not form (a stale id, a rank that died) must not hang the caller.  The call cannot be cancelled; on
                // a timeout its thread is left behind and the caller gets MI_EIO (and is expected to exit)
                struct Init { std::mutex m; std::condition_variable cv; bool done = false; int rc = 0; void* comm = nullptr; };
                auto st = std::make_shared<Init>();
                auto fn = R.CommInitRank;
                std::thread([st, fn, world, id, rank, device] {
                    hipSetDevice(device);
                    void* c = nullptr;
                    const int rc = fn(&c, world, id, rank);
                    std::lock_guard<std::mutex> g(st->m);
                    st->rc = rc; st->comm = c; st->done = true;
                    st->cv.notify_all();
                }).detach();
                std::unique_lock<std::mutex> lk(st->m);
                // (the id exchange itself is quick; the bound is the caller's timeout, at least 10 s for the bootstrap of a whole node)
                if (!st->cv.wait_for(lk, std::chrono::milliseconds(std::max(timeout_ms, 10000)), [&] { return st->done; }))
                    throw ApiError(MI_EIO, "ncclCommInitRank did not complete within the timeout (a stale rendezvous file '" + path + "', or a rank that never arrived)");
                comm = st->comm;
                nccl(st->rc, "ncclCommInitRank");
            } else {
                nccl(R.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
            }
            if (rank == root) std::remove(path.c_str());   // every rank has read it: CommInitRank is collective
            mi::hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
            mi::hip_check(hipMalloc(reinterpret_cast<void**>(&d), nbytes), "hipMalloc");
            if (rank == root) mi::hip_check(hipMemcpyAsync(d, buf, nbytes, hipMemcpyHostToDevice, s), "H2D model bytes");
            nccl(R.Broadcast(d, d, nbytes, /* ncclUint8 */ 1, root, comm, s), "ncclBroadcast");
            if (rank != root) mi::hip_check(hipMemcpyAsync(buf, d, nbytes, hipMemcpyDeviceToHost, s), "D2H model bytes");
            mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        } catch (...) {
            if (d) hipFree(d);
            if (s) hipStreamDestroy(s);
            if (comm) R.CommDestroy(comm);
            throw;
        }
        hipFree(d);
        hipStreamDestroy(s);
        nccl(R.CommDestroy(comm), "ncclCommDestroy");
    });
}

// ------------------------------------------------------------------------------------------------ host helpers
int mi_bbox_to_roi(const double bbox[4], int image_w, int image_h, const double* rotation_keypoints, double scale_x, double scale_y, int size_mode,
                   mi_rect* out) {
    return guarded([&] {
        require(bbox && out, "null argument");
        require(size_mode >= 0 && size_mode <= 2, "size_mode must be 0 (Default), 1 (SquareLong) or 2 (SquareShort)");
        if (!mi::bbox_to_roi(bbox, image_w, image_h, rotation_keypoints, scale_x, scale_y, size_mode, out)) throw ApiError(MI_EINVAL, "bbox must be normalized");
    });
}

int mi_bbox_from_landmarks(const mi_landmark* lm, int count, double bbox_out[4]) {
    return guarded([&] {
        require(lm && bbox_out, "null argument");
        if (count < 2) throw ApiError(MI_EINVAL, "landmarks must contain at least 2 items");  // transform.rs:147-149
        double xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
        for (int i = 0; i < count; i++) {  // f64::min / max: a NaN operand is ignored, like fmin / fmax
            xmin = std::fmin(xmin, lm[i].x); ymin = std::fmin(ymin, lm[i].y);
            xmax = std::fmax(xmax, lm[i].x); ymax = std::fmax(ymax, lm[i].y);
        }
        bbox_out[0] = xmin; bbox_out[1] = ymin; bbox_out[2] = xmax; bbox_out[3] = ymax;
    });
}

int mi_face_detection_to_roi(const mi_detection* det, int image_w, int image_h, mi_rect* out) {
    return guarded([&] {
        require(det && out, "null argument");
        // Detection::scaled_by_image_size multiplies in f32 (types.rs:237-245); keypoints 0/1 = eyes (face_detection.rs:91-98)
        const float w = static_cast<float>(image_w), hgt = static_cast<float>(image_h);
        const float lx = det->data[4] * w, ly = det->data[5] * hgt, rx = det->data[6] * w, ry = det->data[7] * hgt;
        const double kp[4] = {lx, ly, rx, ry};
        const double bbox[4] = {det->data[0], det->data[1], det->data[2], det->data[3]};
        if (!mi::bbox_to_roi(bbox, image_w, image_h, kp, 1.5, 1.5, 1, out)) throw ApiError(MI_EINVAL, "bbox must be normalized");
    });
}

int mi_iris_roi_from_face_landmarks(const mi_landmark* lm, int image_w, int image_h, mi_rect* left_eye, mi_rect* right_eye) {
    return guarded([&] {
        require(lm && left_eye && right_eye, "null argument");
        const int idx[2][2] = {{33, 133}, {362, 263}};  // iris_landmark.rs:29-35
        mi_rect* outs[2] = {left_eye, right_eye};
        for (int e = 0; e < 2; e++) {
            const mi_landmark &a = lm[idx[e][0]], &b = lm[idx[e][1]];
            const double bbox[4] = {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmax(a.x, b.x), std::fmax(a.y, b.y)};
            const double kp[4] = {a.x, a.y, b.x, b.y};
            if (!mi::bbox_to_roi(bbox, image_w, image_h, kp, 2.3, 2.3, 1, outs[e])) throw ApiError(MI_EINVAL, "bbox must be normalized");
        }
    });
}

int mi_jpeg_info(const uint8_t* bytes, size_t nbytes, int* width, int* height) {
    return guarded([&] {
        require(bytes && width && height, "null argument");
        try {
            mi::jpeg_parse_size(bytes, nbytes, width, height);
        } catch (const std::runtime_error& e) {
            throw ApiError(MI_EINVAL, e.what());
        }
    });
}

int mi_jpeg_decode_rgb(int device, const uint8_t* bytes, size_t nbytes, uint8_t* rgb, size_t cap_bytes, int* width, int* height, int mem, void* stream) {
    return guarded([&] {
        require(bytes && rgb && width && height, "null argument");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        mi::JpegFrame f;
        try {
            mi::jpeg_entropy_decode(bytes, nbytes, &f);  // host: the serial part
        } catch (const std::runtime_error& e) {
            throw ApiError(MI_EINVAL, e.what());
        }
        const size_t out_bytes = static_cast<size_t>(3) * f.width * f.height;
        if (cap_bytes < out_bytes) throw ApiError(MI_EINVAL, "rgb buffer too small for the decoded picture");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) throw ApiError(MI_EDEVICE, "no such HIP device (the JPEG sample arithmetic runs on the GPU; no CPU fallback exists)");
        mi::hip_check(hipSetDevice(device), "hipSetDevice");
        hipStream_t s = static_cast<hipStream_t>(stream);
        DeviceBuf coef, qt, planes, out;
        const size_t coef_bytes = f.coef.size() * sizeof(int16_t);
        auto* d_coef = static_cast<int16_t*>(coef.get(coef_bytes));
        auto* d_qt = static_cast<uint16_t*>(qt.get(sizeof f.qt));
        auto* d_planes = static_cast<uint8_t*>(planes.get(mi::jpeg_plane_bytes(f)));
        uint8_t* d_rgb = mem == MI_MEM_DEVICE ? rgb : static_cast<uint8_t*>(out.get(out_bytes));
        mi::hip_check(hipMemcpyAsync(d_coef, f.coef.data(), coef_bytes, hipMemcpyHostToDevice, s), "H2D coefficients");
        mi::hip_check(hipMemcpyAsync(d_qt, f.qt, sizeof f.qt, hipMemcpyHostToDevice, s), "H2D quantisation tables");
        int rc = mi::launch_jpeg_idct(f, d_coef, d_qt, d_planes, s);
        if (rc == 0) rc = mi::launch_jpeg_color(f, d_planes, d_rgb, s);
        if (rc) throw std::runtime_error(std::string("jpeg kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        if (mem == MI_MEM_HOST) mi::hip_check(hipMemcpyAsync(rgb, d_rgb, out_bytes, hipMemcpyDeviceToHost, s), "D2H rgb");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");  // the scratch buffers and the host coefficients die with this call
        *width = f.width;
        *height = f.height;
    });
}

namespace {
// eye-contour landmark n of the iris model refines face-mesh landmark kEyeToFace[eye][n] (iris_landmark.rs:64-95)
const int kEyeToFace[2][MI_NUM_EYE_LANDMARKS] = {
    {
        33, 7, 163, 144, 145, 153, 154, 155, 133, 246, 161, 160, 159, 158, 157, 173, 130, 25, 110, 24, 23,
        22, 26, 112, 243, 247, 30, 29, 27, 28, 56, 190, 226, 31, 228, 229, 230, 231, 232, 233, 244, 113, 225,
        224, 223, 222, 221, 189, 35, 124, 46, 53, 52, 65, 143, 111, 117, 118, 119, 120, 121, 128, 245, 156,
        70, 63, 105, 66, 107, 55, 193,
    },
    {
        263, 249, 390, 373, 374, 380, 381, 382, 362, 466, 388, 387, 386, 385, 384, 398, 359, 255, 339, 254,
        253, 252, 256, 341, 463, 467, 260, 259, 257, 258, 286, 414, 446, 261, 448, 449, 450, 451, 452, 453,
        464, 342, 445, 444, 443, 442, 441, 413, 265, 353, 276, 283, 282, 295, 372, 340, 346, 347, 348, 349,
        350, 357, 465, 383, 300, 293, 334, 296, 336, 285, 417,
    },
};
}  // namespace

int mi_update_face_landmarks_with_iris_results(const mi_landmark* face, const mi_landmark* left, const mi_landmark* right, mi_landmark* out) {
    return guarded([&] {
        require(face && left && right && out, "null argument");
        if (out != face) std::memmove(out, face, sizeof(mi_landmark) * MI_NUM_FACE_LANDMARKS);
        const mi_landmark* eyes[2] = {left, right};  // left first, then right, as the reference's two loops
        for (int e = 0; e < 2; e++)
            for (int n = 0; n < MI_NUM_EYE_LANDMARKS; n++) out[kEyeToFace[e][n]] = eyes[e][n];
    });
}

int mi_image_to_tensor(int device, const uint8_t* rgb, int width, int height, int stride, const mi_rect* roi, int out_w, int out_h,
                       int keep_aspect_ratio, double range_min, double range_max, int flip_horizontal, float* out,
                       double padding_out[4], int mem, void* stream) {
    return guarded([&] {
        require(rgb && out && padding_out, "null argument");
        require(width > 0 && height > 0 && stride >= 3 * width && out_w > 0 && out_h > 0, "bad image geometry");
        require(mem == MI_MEM_HOST || mem == MI_MEM_DEVICE, "mem must be MI_MEM_HOST or MI_MEM_DEVICE");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) throw ApiError(MI_EDEVICE, "no such HIP device");
        mi::hip_check(hipSetDevice(device), "hipSetDevice");
        hipStream_t s = static_cast<hipStream_t>(stream);
        DeviceBuf img, outbuf;
        float* d_out = out;
        size_t ob = sizeof(float) * 3 * out_w * out_h;
        if (mem == MI_MEM_HOST) d_out = static_cast<float*>(outbuf.get(ob));
        mi::image_to_tensor_device(rgb, width, height, stride, roi, out_w, out_h, keep_aspect_ratio != 0, range_min, range_max,
                                   flip_horizontal != 0, d_out, padding_out, img.get(mi::image_to_tensor_scratch_bytes(width, height, stride, roi, out_w, out_h, keep_aspect_ratio != 0)), s);
        if (mem == MI_MEM_HOST) mi::hip_check(hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, s), "D2H tensor");
        mi::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}

}  // extern "C"
