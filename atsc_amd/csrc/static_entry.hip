// atsc_hip_mix_fir_static: the mixer + FIR of atsc_hip_mix_fir (offset.process_field + filter.process_field,
// reference: src/atsc/atsc.cpp:85-86) for callers that cannot allocate -- code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc.  No handle: everything the kernels need beyond the caller's input and output
// (the oscillator table T_0, the oscillator state entering the batch, and the per-field table T_f of the batch that
// k_mix_fir_fft reads) lives in ONE file-scope device array, `scratch`, of fixed size; the entry point calls no
// hipMalloc / hipFree / allocating routine, on the device or on the host, after its one-time set-up per device.
//
// Each GPU holds its own copy of `scratch` at its own address: in host code the array's NAME is only a stand-in, so
// the set-up asks the runtime for the address on the current device (hipGetSymbolAddress) once per device and every use
// below -- kernel arguments, copies -- goes through that pointer.
//
// All calls on one GPU share that GPU's array: calls that overlap on one GPU (two host threads, or two streams
// without an order between them) race on it.  A call leaves T_(field_index + fields) and the batch table behind; the
// next call recomputes both from T_0, nothing has to start at zero.
#include "atsc_hip.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <new>

#include "kernels.h"
#include "tables.h"

using namespace atsc_k;

namespace {

constexpr int kStaticMaxFields = ATSC_HIP_STATIC_MAX_FIELDS;
constexpr int kMaxDevices = 64;

// layout of `scratch`, in complex64 entries
constexpr size_t kOffT0 = 0;                                   // T_0: written once per device at set-up, then read-only
constexpr size_t kOffStateA = kOffT0 + kPaddedSyms;            // oscillator state entering the batch (T_field_index)
constexpr size_t kOffStateB = kOffStateA + kPaddedSyms;        // oscillator state leaving the batch
constexpr size_t kOffTab = kOffStateB + kPaddedSyms;           // T_f of every field of the batch + the slack rows
constexpr size_t kScratchEntries = kOffTab + (size_t)kStaticMaxFields * kPaddedSyms + kOscTabSlack;

// (3 + 4) x 261248 + 1024 complex64 = 14.6 MB per GPU, held until the process ends
static __device__ __align__(256) float2 scratch[kScratchEntries];

struct DeviceSetup {
    std::once_flag once;     // two host threads' first calls on a device run the set-up once
    int rc = ATSC_HIP_ERR_STATE;
    float2* ptr = nullptr;   // this device's address of `scratch`
    float2 scale{};
};
DeviceSetup g_setup[kMaxDevices];

// puts the caller's current device back on every way out of the entry point
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Once per device, with that device current.  This is the only place that builds host tables and copies them over;
// it must run outside a graph capture (call the entry point once with fields = 0 to force it).  It also launches one
// kernel of every code object the entry point uses (kernels.hip: k_osc_advance; fir_fft.hip: k_osc_table, k_mix_fir_fft)
// on scratch memory, so that the runtime's lazy module loading -- which allocates -- happens here and not in a later call
// that may be under capture.
int setup_device_once(DeviceSetup& d) {
    float2* p = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return static_cast<int>(e);
    if (!p) return static_cast<int>(hipErrorInvalidSymbol);
    try {
        atsc_tables::Tables tab;
        atsc_tables::build(tab);
        e = hipMemcpy(p + kOffT0, tab.osc0.data(), sizeof(float2) * kPaddedSyms, hipMemcpyHostToDevice);
        if (e != hipSuccess) return static_cast<int>(e);
        e = upload_fft_tables(tab.taps);   // the FFT form's twiddle / spectrum constants of THIS device
        if (e != hipSuccess) return static_cast<int>(e);
        d.scale = make_float2(tab.scale[0], tab.scale[1]);
    } catch (const std::bad_alloc&) {
        return static_cast<int>(hipErrorOutOfMemory);
    }
    // module warm-up: one field of one channel, symbols read from the (initialised) T_0 bytes, IQ written into the slot of
    // the batch table's third field; every result is overwritten by the first real call
    launch_osc_advance(nullptr, p + kOffT0, p + kOffStateA, d.scale, 1);
    launch_osc_table(nullptr, p + kOffStateA, p + kOffStateB, p + kOffTab, d.scale, 1, kOscComplex);
    for (int fmt : {ATSC_HIP_IQ_CF32, ATSC_HIP_IQ_CS16, ATSC_HIP_IQ_CS8})
        launch_mix_fir_fft(nullptr, reinterpret_cast<const uint8_t*>(p + kOffT0), kFieldSyms, p + kOffTab + 2 * (size_t)kPaddedSyms,
                           kFieldSyms, p + kOffTab, 1, 1, fmt);
    e = hipGetLastError();
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return static_cast<int>(e);
    d.ptr = p;
    return ATSC_HIP_OK;
}

int setup_device(int device) {
    DeviceSetup& d = g_setup[device];
    std::call_once(d.once, [&] { d.rc = setup_device_once(d); });
    return d.rc;   // a failed set-up stays failed for the process (the error code says why)
}

}  // namespace

extern "C" int atsc_hip_mix_fir_static(int device, const uint8_t* d_sym, int channels, int fields, uint64_t field_index,
                                       void* d_iq, int iq_format, void* stream) {
    if (device < 0 || device >= kMaxDevices || channels < 1 || fields < 0) return ATSC_HIP_ERR_ARG;
    if (fields > kStaticMaxFields) return ATSC_HIP_ERR_TOO_LARGE;   // the batch table in `scratch` holds 4 fields
    if (!atsc_hip_iq_sample_bytes(iq_format)) return ATSC_HIP_ERR_ARG;
    // T_f is re-derived by field_index sequential steps per call and lane: 2^24 of them are about a second of GPU time
    if (field_index > ATSC_HIP_MAX_SEEK_STEPS) return ATSC_HIP_ERR_TOO_LARGE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ATSC_HIP_ERR_NO_DEVICE;
    if (device >= ndev) return ATSC_HIP_ERR_ARG;
    DeviceRestore restore;   // the caller's current device is put back when the call returns
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return static_cast<int>(e);
    const int rc = setup_device(device);
    if (rc) return rc;
    if (fields == 0) return ATSC_HIP_OK;   // set-up only
    if (!d_sym || !d_iq) return ATSC_HIP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_sym) & 15) || (reinterpret_cast<uintptr_t>(d_iq) & 15)) return ATSC_HIP_ERR_ARG;

    const DeviceSetup& d = g_setup[device];
    auto s = static_cast<hipStream_t>(stream);
    float2* t0 = d.ptr + kOffT0;
    float2* state_in = d.ptr + kOffStateA;
    float2* state_out = d.ptr + kOffStateB;
    float2* tab = d.ptr + kOffTab;
    // T_field_index = T_0 (x) scale^field_index by the reference's own sequential fp32 recursion (offset.h:35-41)
    const float2* osc_in = t0;
    if (field_index) {
        launch_osc_advance(s, t0, state_in, d.scale, (unsigned)field_index);
        osc_in = state_in;
    }
    launch_osc_table(s, osc_in, state_out, tab, d.scale, fields, kOscComplex);
    launch_mix_fir_fft(s, d_sym, (size_t)fields * kFieldSyms, d_iq, (size_t)fields * kFieldSyms, tab, channels, fields,
                       iq_format);
    return static_cast<int>(hipGetLastError());
}
