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

}  // namespace atsc
