// Null-sink throughput driver for the MI355X encoder: the counterpart of the reference's
// util/atsc_encoder_null.cpp (reads whole 312-packet fields from a file or stdin, pushes them through
// atsc::atsc_encoder::process, discards the IQ).  Adds what the reference's tool lacks: a field-batch
// option, a throughput report, the SDR sample formats, and the TS ingest front-end (include/atsc_ts_ingest.h): the
// input may be an unaligned / damaged byte stream; with "pad" the last partial field is completed with null packets
// instead of being dropped (the reference drops it, util/atsc_encoder_null.cpp:40).
//   usage: atsc_encoder_null <file|-> [fields_per_batch=8] [device=0] [cf32|cs16|cs8] [pad] [--streams K]
// --streams K: K copies of the input as K independent streams through atsc::atsc_encoder_group (one K-channel handle, the
// many-channel kernels) instead of one atsc::atsc_encoder; the report counts the fields of all streams.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "atsc_encoder.hpp"
#include "atsc_ts_ingest.h"

int main(int argc, char** argv) {
    int streams = 1;
    for (int i = 1; i + 1 < argc; ++i)
        if (!std::strcmp(argv[i], "--streams")) {   // taken out of the positional arguments
            streams = std::max(1, std::atoi(argv[i + 1]));
            for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2];
            argc -= 2;
            break;
        }
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <ts-file|-> [fields_per_batch] [device] [cf32|cs16|cs8] [pad] [--streams K]\n", argv[0]);
        return 2;
    }
    const int batch = argc > 2 ? std::max(1, std::atoi(argv[2])) : 8;
    const int device = argc > 3 ? std::atoi(argv[3]) : 0;
    auto fmt = atsc::atsc_encoder::iq_format::cf32;
    if (argc > 4 && !std::strcmp(argv[4], "cs16")) fmt = atsc::atsc_encoder::iq_format::cs16;
    if (argc > 4 && !std::strcmp(argv[4], "cs8")) fmt = atsc::atsc_encoder::iq_format::cs8;
    FILE* in = std::strcmp(argv[1], "-") == 0 ? stdin : std::fopen(argv[1], "rb");
    if (!in) {
        std::perror("open");
        return 1;
    }
    std::unique_ptr<atsc::atsc_encoder> enc;
    std::unique_ptr<atsc::atsc_encoder_group> group;
    try {
        if (streams > 1) {
            group = atsc::atsc_encoder_group::create(streams, device, batch);
            group->member(0).set_iq_format(fmt);
        } else {
            enc = atsc::atsc_encoder::create(device, batch);
            enc->set_iq_format(fmt);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    const bool pad = argc > 5 && !std::strcmp(argv[5], "pad");
    atsc_ts_ingest* ing = atsc_ts_ingest_create(0, 0);
    // four GPU batches per process() call: inside a call the upload and kernels of batch k+1 overlap the download and the
    // callbacks of batch k (include/atsc_hip.h); a call with a single batch cannot overlap anything
    const size_t call_packets = static_cast<size_t>(4 * batch) * ATSC_HIP_PACKETS_PER_FIELD;
    std::vector<uint8_t> chunk(1 << 20), buf(call_packets * ATSC_HIP_TS_PACKET_BYTES);
    unsigned long long fields = 0, bytes_out = 0;
    auto sink = [&](void*, unsigned n) {
        ++fields;
        bytes_out += n;
    };
    // one call's worth of packets to the encoder, or to every member of the group in turn (the last member's call runs the
    // batch and fires every member's callbacks)
    auto push = [&](uint8_t* p, size_t packets) {
        if (!group) {
            enc->process(p, static_cast<unsigned>(packets), sink);
            return;
        }
        for (int c = 0; c < streams; ++c) group->member(c).process(p, static_cast<unsigned>(packets), sink);
    };
    const auto t0 = std::chrono::steady_clock::now();
    size_t got;
    while ((got = std::fread(chunk.data(), 1, chunk.size(), in)) > 0) {
        atsc_ts_ingest_push(ing, chunk.data(), got);
        while (atsc_ts_ingest_available(ing) >= call_packets) {
            atsc_ts_ingest_pop(ing, buf.data(), call_packets);
            push(buf.data(), call_packets);
        }
    }
    // end of stream: whole fields that are left, then (with "pad") the last partial field completed with null packets
    size_t left = atsc_ts_ingest_available(ing);
    if (pad && left % ATSC_HIP_PACKETS_PER_FIELD) left += ATSC_HIP_PACKETS_PER_FIELD - left % ATSC_HIP_PACKETS_PER_FIELD;
    left -= left % ATSC_HIP_PACKETS_PER_FIELD;
    if (left) {
        atsc_ts_ingest_pop_padded(ing, buf.data(), left);
        push(buf.data(), left);
    }
    atsc_ts_ingest_stats ist{};
    atsc_ts_ingest_get_stats(ing, &ist);
    atsc_ts_ingest_destroy(ing);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (in != stdin) std::fclose(in);
    const double msym = fields * double(ATSC_HIP_FIELD_SYMBOLS) / s / 1e6;
    std::printf("{\"streams\": %d, \"fir_form\": %d, \"fields\": %llu, \"seconds\": %.4f, \"msym_s\": %.2f, \"realtime_channels\": %.2f, \"iq_bytes\": %llu, "
                "\"bytes_dropped\": %llu, \"sync_losses\": %llu, \"nulls_stuffed\": %llu}\n",
                streams, group ? atsc_hip_get_fir_form(group->handle()) : -1, fields, s, msym, msym / 10.762238, bytes_out, (unsigned long long)ist.bytes_dropped,
                (unsigned long long)ist.sync_losses, (unsigned long long)ist.nulls_stuffed);
    return 0;
}
