// Null-sink throughput driver for the MI355X encoder: the counterpart of the reference's
// util/atsc_encoder_null.cpp (reads whole 312-packet fields from a file or stdin, pushes them through
// atsc::atsc_encoder::process, discards the IQ).  Adds what the reference's tool lacks: a field-batch
// option and a throughput report.
//   usage: atsc_encoder_null <file|-> [fields_per_call=8] [device=0]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "atsc_encoder.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <ts-file|-> [fields_per_call] [device]\n", argv[0]);
        return 2;
    }
    const int batch = argc > 2 ? std::max(1, std::atoi(argv[2])) : 8;
    const int device = argc > 3 ? std::atoi(argv[3]) : 0;
    FILE* in = std::strcmp(argv[1], "-") == 0 ? stdin : std::fopen(argv[1], "rb");
    if (!in) {
        std::perror("open");
        return 1;
    }
    std::unique_ptr<atsc::atsc_encoder> enc;
    try {
        enc = atsc::atsc_encoder::create(device, batch);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    std::vector<uint8_t> buf(static_cast<size_t>(batch) * ATSC_HIP_FIELD_TS_BYTES);
    unsigned long long fields = 0, bytes_out = 0;
    const auto t0 = std::chrono::steady_clock::now();
    size_t got;
    while ((got = std::fread(buf.data(), ATSC_HIP_FIELD_TS_BYTES, batch, in)) > 0) {
        enc->process(buf.data(), static_cast<unsigned>(got) * ATSC_HIP_PACKETS_PER_FIELD, [&](void*, unsigned n) {
            ++fields;
            bytes_out += n;
        });
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (in != stdin) std::fclose(in);
    const double msym = fields * double(ATSC_HIP_FIELD_SYMBOLS) / s / 1e6;
    std::printf("{\"fields\": %llu, \"seconds\": %.4f, \"msym_s\": %.2f, \"realtime_channels\": %.2f, \"iq_bytes\": %llu}\n",
                fields, s, msym, msym / 10.762238, bytes_out);
    return 0;
}
