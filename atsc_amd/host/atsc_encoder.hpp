// C++ host adapter with the shape of the reference's encoder interface
// (reference: src/atsc/atsc.h:7-16 -- atsc::atsc_encoder::create() / process(pkt, count, fn)), implemented
// over the C-ABI of include/atsc_hip.h.  A driver written against the reference (util/atsc_encoder_null.cpp,
// util/atsc_encoder.cpp) compiles against this header unchanged and links libatsc_hip.so instead of libatsc.a.
//
// Differences a caller can observe: create() throws std::runtime_error when no MI355X/HIP device is
// available (the reference never fails, atsc.cpp:93-95); the buffer handed to fn is pinned host memory
// owned by the library, valid only during the callback exactly like the reference's (atsc.cpp:89).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "atsc_hip.h"

namespace atsc {

class atsc_encoder {
public:
    virtual ~atsc_encoder() = default;

    // N 188-byte TS packets in; fn(iq, 260416*8) once per completed 312-packet field, synchronously, in order.
    virtual void process(uint8_t* pkt, unsigned count, std::function<void(void*, unsigned)> fn) = 0;

    // SDR sink adapter (reference caller: util/atsc_encoder.cpp:75,84-100).  The sample format of the
    // buffers handed to fn: cf32 (the reference's), cs16 or cs8 (converted inside the FIR kernel); and the transmitter
    // loop's chunking: write(samples, n) for every piece of at most mtu samples of every completed field.
    enum class iq_format { cf32 = ATSC_HIP_IQ_CF32, cs16 = ATSC_HIP_IQ_CS16, cs8 = ATSC_HIP_IQ_CS8 };
    virtual void set_iq_format(iq_format f) = 0;
    // 2: the interpolating form of the filter (atsc_hip_set_upsample): two samples per symbol, for a radio run at twice the
    // symbol rate (setSampleRate(2 * 4500000.0 / 286 * 684) instead of util/atsc_encoder.cpp:70); fn / write then see
    // 2 x 260416 samples per field, the even ones the reference's
    virtual void set_upsample(int factor) = 0;
    virtual void process_chunked(uint8_t* pkt, unsigned count, size_t mtu_samples,
                                 std::function<void(const void*, size_t)> write) = 0;

    // device: HIP ordinal; max_fields: how many complete fields one process() call may push through the GPU
    // in a single batch (1 reproduces the reference's field-at-a-time latency; larger amortises launches).
    static std::unique_ptr<atsc_encoder> create(int device = 0, int max_fields = 8);

protected:
    atsc_encoder() = default;
};

namespace detail {

class hip_encoder final : public atsc_encoder {
public:
    hip_encoder(int device, int max_fields) {
        atsc_hip_config cfg{device, 1, max_fields, 0};
        const int rc = atsc_hip_create(&cfg, &h_);
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string("atsc_hip_create: ") + atsc_hip_error_string(rc));
    }
    ~hip_encoder() override { atsc_hip_destroy(h_); }
    hip_encoder(const hip_encoder&) = delete;
    hip_encoder& operator=(const hip_encoder&) = delete;

    void process(uint8_t* pkt, unsigned count, std::function<void(void*, unsigned)> fn) override {
        const int rc = atsc_hip_process(h_, pkt, count, &hip_encoder::trampoline, &fn);
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string("atsc_hip_process: ") + atsc_hip_error_string(rc));
    }

    void set_iq_format(iq_format f) override {
        const int rc = atsc_hip_set_iq_format(h_, static_cast<int>(f));
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string("atsc_hip_set_iq_format: ") + atsc_hip_error_string(rc));
    }
    void set_upsample(int factor) override {
        const int rc = atsc_hip_set_upsample(h_, factor);
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string("atsc_hip_set_upsample: ") + atsc_hip_error_string(rc));
    }
    void process_chunked(uint8_t* pkt, unsigned count, size_t mtu_samples,
                         std::function<void(const void*, size_t)> write) override {
        const int rc = atsc_hip_process_chunked(
            h_, pkt, count, mtu_samples,
            [](void* user, const void* samples, size_t n, int) {
                (*static_cast<std::function<void(const void*, size_t)>*>(user))(samples, n);
            },
            &write);
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string("atsc_hip_process_chunked: ") + atsc_hip_error_string(rc));
    }

private:
    static void trampoline(void* user, void* iq, unsigned bytes) {
        auto& fn = *static_cast<std::function<void(void*, unsigned)>*>(user);
        if (fn) fn(iq, bytes);
    }
    atsc_hip_encoder* h_ = nullptr;
};

}  // namespace detail

inline std::unique_ptr<atsc_encoder> atsc_encoder::create(int device, int max_fields) {
    return std::make_unique<detail::hip_encoder>(device, max_fields);
}

// K streams, K atsc_encoder-shaped objects, ONE K-channel handle (include/atsc_hip.h: atsc_hip_process_multi).  A caller
// that runs one reference encoder per TS (util/atsc_encoder.cpp:83-100 per stream) keeps its code per stream -- process(pkt,
// count, fn) on its member -- and the fields of all streams go through the GPU together, on the many-channel kernels.
//   auto group = atsc::atsc_encoder_group::create(32, 0, 4);
//   group->member(7).process(pkt, n, fn);     // buffers; the call that completes a field on every member runs the batch
// What differs from K independent encoders: the members advance in lockstep (a member that stops being fed stalls the
// others: feed null packets); a member's fn is KEPT (copied) and may be invoked during ANOTHER member's process() call, on
// that call's thread -- what it captures by reference must outlive the group or the member's next call; set_iq_format /
// set_upsample on any member apply to the whole group (one handle) and are allowed only while nothing is buffered.
class atsc_encoder_group {
public:
    static std::unique_ptr<atsc_encoder_group> create(int streams, int device = 0, int max_fields = 4) {
        return std::unique_ptr<atsc_encoder_group>(new atsc_encoder_group(streams, device, max_fields));
    }
    ~atsc_encoder_group() { atsc_hip_destroy(h_); }
    atsc_encoder_group(const atsc_encoder_group&) = delete;
    atsc_encoder_group& operator=(const atsc_encoder_group&) = delete;

    int streams() const { return static_cast<int>(members_.size()); }
    atsc_encoder& member(int i) { return *members_.at(static_cast<size_t>(i)); }
    atsc_hip_encoder* handle() { return h_; }   // for atsc_hip_get_fir_form and the like

private:
    class member_t final : public atsc_encoder {
    public:
        member_t(atsc_encoder_group* g, int channel) : g_(g), channel_(channel) {}
        void process(uint8_t* pkt, unsigned count, std::function<void(void*, unsigned)> fn) override {
            fn_ = std::move(fn);
            mtu_ = 0;
            feed(pkt, count);
        }
        void process_chunked(uint8_t* pkt, unsigned count, size_t mtu_samples,
                             std::function<void(const void*, size_t)> write) override {
            if (!mtu_samples || !write) throw std::runtime_error("process_chunked: invalid argument");
            write_ = std::move(write);
            mtu_ = mtu_samples;
            feed(pkt, count);
        }
        void set_iq_format(iq_format f) override { g_->check(atsc_hip_set_iq_format(g_->h_, static_cast<int>(f)), "atsc_hip_set_iq_format"); g_->sample_bytes_ = atsc_hip_iq_sample_bytes(static_cast<int>(f)); }
        void set_upsample(int factor) override { g_->check(atsc_hip_set_upsample(g_->h_, factor), "atsc_hip_set_upsample"); }

    private:
        void feed(uint8_t* pkt, unsigned count) {
            g_->check(atsc_hip_process_multi(g_->h_, channel_, pkt, count, &member_t::trampoline, this), "atsc_hip_process_multi");
        }
        static void trampoline(void* user, void* iq, unsigned bytes) {
            auto* m = static_cast<member_t*>(user);
            if (m->mtu_ == 0) {
                if (m->fn_) m->fn_(iq, bytes);
                return;
            }
            // the transmitter loop's chunking (util/atsc_encoder.cpp:84-100)
            const size_t ss = m->g_->sample_bytes_;
            size_t remaining = bytes / ss;
            const uint8_t* p = static_cast<const uint8_t*>(iq);
            while (remaining > 0) {
                const size_t n = remaining < m->mtu_ ? remaining : m->mtu_;
                m->write_(p, n);
                p += n * ss;
                remaining -= n;
            }
        }
        atsc_encoder_group* g_;
        int channel_;
        size_t mtu_ = 0;
        std::function<void(void*, unsigned)> fn_;
        std::function<void(const void*, size_t)> write_;
    };

    atsc_encoder_group(int streams, int device, int max_fields) {
        atsc_hip_config cfg{device, streams, max_fields, 0};
        check(atsc_hip_create(&cfg, &h_), "atsc_hip_create");
        for (int i = 0; i < streams; ++i) members_.push_back(std::make_unique<member_t>(this, i));
    }
    static void check(int rc, const char* what) {
        if (rc != ATSC_HIP_OK) throw std::runtime_error(std::string(what) + ": " + atsc_hip_error_string(rc));
    }
    atsc_hip_encoder* h_ = nullptr;
    size_t sample_bytes_ = 8;
    std::vector<std::unique_ptr<member_t>> members_;
};

}  // namespace atsc
