// Host-pointer batches at PCIe speed: a three-slot pipeline of pinned staging buffers in which
// chunk i's host->pinned copy (CPU threads), chunk i-1's H2D, chunk i-2's kernel and D2H and
// chunk i-3's pinned->host copy all overlap.  Copies and kernels sit on three HIP streams chained
// by events; nothing synchronises the device per chunk.  Used by the host-pointer entry points
// of capi.cpp (the reference's trait surface takes host slices, msbwt_core.rs:124).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace msbwt {

// A few persistent worker threads that split large memcpys (pageable <-> pinned) between them.
class CopyPool {
  public:
    explicit CopyPool(int threads) {
        for (int t = 0; t < threads; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &w : workers_) w.join();
    }
    int threads() const { return int(workers_.size()); }

    // dst[0..bytes) = src[0..bytes), split over the workers; returns when done
    void copy(void *dst, const void *src, size_t bytes) {
        if (bytes < (size_t(1) << 20) || workers_.empty()) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::unique_lock<std::mutex> lock(mu_);
        dst_ = static_cast<char *>(dst);
        src_ = static_cast<const char *>(src);
        bytes_ = bytes;
        pending_ = int(workers_.size());
        ++generation_;
        wake_.notify_all();
        done_.wait(lock, [this] { return pending_ == 0; });
    }

  private:
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lock(mu_);
            wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            const size_t n = workers_.size(), piece = (bytes_ / n + 63) / 64 * 64;
            const size_t lo = std::min(bytes_, piece * size_t(t)), hi = (size_t(t) + 1 == n) ? bytes_ : std::min(bytes_, lo + piece);
            char *d = dst_;
            const char *s = src_;
            lock.unlock();
            if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
            lock.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    bool stop_ = false;
    uint64_t generation_ = 0;
    int pending_ = 0;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0;
};

// One input or output array of a batch: `item_bytes` bytes per item, contiguous on the host.
struct HostArray {
    const void *in = nullptr;  // inputs
    void *out = nullptr;       // outputs
    size_t item_bytes = 0;
};

class HostPipeline {
  public:
    static constexpr int kSlots = 3;
    ~HostPipeline() { release(); }

    // launch(first_item, items, d_in[], d_out[], stream): enqueue the kernel(s) of one chunk.
    using Launch = std::function<hipError_t(size_t, size_t, void *const *, void *const *, hipStream_t)>;

    // Runs n items through the pipeline in chunks of `chunk` items.  `compute` is the stream the
    // kernels go to.  Returns the first HIP error.
    hipError_t run(size_t n, size_t chunk, const std::vector<HostArray> &ins, const std::vector<HostArray> &outs,
                   hipStream_t compute, const Launch &launch) {
        if (n == 0) return hipSuccess;
        chunk = std::max<size_t>(1, std::min(chunk, n));
        size_t in_bytes = 0, out_bytes = 0;
        std::vector<size_t> in_off, out_off;
        for (const HostArray &a : ins) { in_off.push_back(in_bytes); in_bytes += (chunk * a.item_bytes + 255) / 256 * 256; }
        for (const HostArray &a : outs) { out_off.push_back(out_bytes); out_bytes += (chunk * a.item_bytes + 255) / 256 * 256; }
        hipError_t e = ensure(in_bytes, out_bytes);
        if (e != hipSuccess) return e;
        if (!pool_) {
            const unsigned hw = std::thread::hardware_concurrency();
            pool_.reset(new CopyPool(int(std::max(2u, std::min(hw ? hw / 2 : 4u, 12u)))));
        }
        const size_t nchunks = (n + chunk - 1) / chunk;
        std::vector<void *> d_in(ins.size()), d_out(outs.size());
        auto finish = [&](size_t c) -> hipError_t {  // chunk c's counts are in its slot's pinned buffer
            Slot &s = slots_[c % kSlots];
            hipError_t err = hipEventSynchronize(s.d2h);
            if (err != hipSuccess) return err;
            const size_t first = c * chunk, m = std::min(chunk, n - first);
            for (size_t a = 0; a < outs.size(); ++a)
                pool_->copy(static_cast<char *>(outs[a].out) + first * outs[a].item_bytes, s.h_out + out_off[a], m * outs[a].item_bytes);
            return hipSuccess;
        };
        // iteration c: stage and enqueue chunk c (its slot was freed one iteration ago), then hand chunk
        // c-2 back to the caller -- by then the device is busy with chunks c-1 and c
        for (size_t c = 0; c < nchunks + kSlots - 1; ++c) {
            if (c < nchunks) {
                Slot &s = slots_[c % kSlots];
                const size_t first = c * chunk, m = std::min(chunk, n - first);
                for (size_t a = 0; a < ins.size(); ++a)
                    pool_->copy(s.h_in + in_off[a], static_cast<const char *>(ins[a].in) + first * ins[a].item_bytes, m * ins[a].item_bytes);
                for (size_t a = 0; a < ins.size(); ++a) {
                    e = hipMemcpyAsync(s.d_in + in_off[a], s.h_in + in_off[a], m * ins[a].item_bytes, hipMemcpyHostToDevice, h2d_);
                    if (e != hipSuccess) return e;
                    d_in[a] = s.d_in + in_off[a];
                }
                for (size_t a = 0; a < outs.size(); ++a) d_out[a] = s.d_out + out_off[a];
                if ((e = hipEventRecord(s.up, h2d_)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(compute, s.up, 0)) != hipSuccess) return e;
                if ((e = launch(first, m, d_in.data(), d_out.data(), compute)) != hipSuccess) return e;
                if ((e = hipEventRecord(s.done, compute)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(d2h_, s.done, 0)) != hipSuccess) return e;
                for (size_t a = 0; a < outs.size(); ++a) {
                    e = hipMemcpyAsync(s.h_out + out_off[a], s.d_out + out_off[a], m * outs[a].item_bytes, hipMemcpyDeviceToHost, d2h_);
                    if (e != hipSuccess) return e;
                }
                if ((e = hipEventRecord(s.d2h, d2h_)) != hipSuccess) return e;
            }
            if (c >= size_t(kSlots - 1) && (e = finish(c - (kSlots - 1))) != hipSuccess) return e;
        }
        return hipSuccess;
    }

    void release() {
        for (Slot &s : slots_) {
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.h_out) (void)hipHostFree(s.h_out);
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.up) (void)hipEventDestroy(s.up);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.d2h) (void)hipEventDestroy(s.d2h);
            s = Slot{};
        }
        if (h2d_) (void)hipStreamDestroy(h2d_);
        if (d2h_) (void)hipStreamDestroy(d2h_);
        h2d_ = d2h_ = nullptr;
        in_cap_ = out_cap_ = 0;
    }

  private:
    struct Slot {
        char *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        hipEvent_t up = nullptr, done = nullptr, d2h = nullptr;
    };

    hipError_t ensure(size_t in_bytes, size_t out_bytes) {
        hipError_t e;
        if (!h2d_ && (e = hipStreamCreateWithFlags(&h2d_, hipStreamNonBlocking)) != hipSuccess) return e;
        if (!d2h_ && (e = hipStreamCreateWithFlags(&d2h_, hipStreamNonBlocking)) != hipSuccess) return e;
        for (Slot &s : slots_) {
            if (!s.up && (e = hipEventCreateWithFlags(&s.up, hipEventDisableTiming)) != hipSuccess) return e;
            if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
            if (!s.d2h && (e = hipEventCreateWithFlags(&s.d2h, hipEventDisableTiming)) != hipSuccess) return e;
        }
        if (in_bytes > in_cap_) {
            for (Slot &s : slots_) {
                if (s.h_in) (void)hipHostFree(s.h_in);
                if (s.d_in) (void)hipFree(s.d_in);
                s.h_in = s.d_in = nullptr;
            }
            in_cap_ = 0;
            for (Slot &s : slots_) {
                if ((e = hipHostMalloc(reinterpret_cast<void **>(&s.h_in), in_bytes, hipHostMallocDefault)) != hipSuccess) return e;
                if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_in), in_bytes)) != hipSuccess) return e;
            }
            in_cap_ = in_bytes;
        }
        if (out_bytes > out_cap_) {
            for (Slot &s : slots_) {
                if (s.h_out) (void)hipHostFree(s.h_out);
                if (s.d_out) (void)hipFree(s.d_out);
                s.h_out = s.d_out = nullptr;
            }
            out_cap_ = 0;
            for (Slot &s : slots_) {
                if ((e = hipHostMalloc(reinterpret_cast<void **>(&s.h_out), out_bytes, hipHostMallocDefault)) != hipSuccess) return e;
                if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_out), out_bytes)) != hipSuccess) return e;
            }
            out_cap_ = out_bytes;
        }
        return hipSuccess;
    }

    Slot slots_[kSlots];
    hipStream_t h2d_ = nullptr, d2h_ = nullptr;
    size_t in_cap_ = 0, out_cap_ = 0;
    std::unique_ptr<CopyPool> pool_;
};

}  // namespace msbwt
