// TS ingest front-end (include/atsc_ts_ingest.h): sync acquisition / loss / re-acquisition on a raw byte stream and
// null-packet stuffing.  The step before atsc_hip_process; the reference's callers assume aligned input
// (util/atsc_encoder_null.cpp:38-43) and the randomizer drops the sync byte unchecked (randomize.h:48).
#include "atsc_ts_ingest.h"

#include <cstring>
#include <deque>
#include <new>
#include <vector>

namespace {
constexpr size_t kPkt = 188;
constexpr uint8_t kSync = 0x47;
}  // namespace

struct atsc_ts_ingest {
    int lock_packets = 5;
    int loss_packets = 3;
    bool locked = false;
    int bad_run = 0;               // consecutive bad sync bytes while locked
    std::vector<uint8_t> raw;      // unconsumed raw bytes; raw[0] is a packet boundary while locked
    std::vector<uint8_t> ready;    // aligned packets not yet popped
    size_t ready_head = 0;         // bytes of `ready` already popped
    atsc_ts_ingest_stats st{};

    // packets emitted in lock whose sync was bad but which may still turn out to be the start of a loss run: they are
    // held back until the run ends (good sync -> they were bit errors, pass them on) or reaches loss_packets (drop them)
    size_t held = 0;               // bytes at the start of `raw` that are such packets

    void compact_ready() {
        if (ready_head && ready_head == ready.size()) {
            ready.clear();
            ready_head = 0;
        } else if (ready_head > (1u << 20)) {
            ready.erase(ready.begin(), ready.begin() + ready_head);
            ready_head = 0;
        }
    }

    void emit(const uint8_t* p, bool bad_sync) {
        const size_t at = ready.size();
        ready.insert(ready.end(), p, p + kPkt);
        if (bad_sync) {
            ready[at] = kSync;
            st.sync_errors++;
        }
    }

    // HUNT from `pos`: smallest offset o >= pos with sync bytes at o, o+188, ... (lock_packets of them).  Everything
    // before it is dropped; at most (lock_packets-1)*188 undecided bytes stay buffered.
    bool hunt(size_t& pos) {
        const size_t need = (size_t)(lock_packets - 1) * kPkt + 1;
        const size_t start = pos;
        bool found = false;
        while (raw.size() >= pos + need) {
            bool ok = true;
            for (int k = 0; k < lock_packets; ++k)
                if (raw[pos + (size_t)k * kPkt] != kSync) { ok = false; break; }
            if (ok) { found = true; break; }
            ++pos;
        }
        st.bytes_dropped += pos - start;
        if (found) {
            locked = true;
            bad_run = 0;
            held = 0;
        }
        return found;
    }

    // consumes what can be decided of `raw` (one erase per call, so a push costs O(bytes))
    void run() {
        size_t pos = 0;
        for (;;) {
            if (!locked && !hunt(pos)) break;
            // LOCKED: raw[pos + held] is a packet boundary; raw[pos, pos+held) are held-back bad-sync packets
            while (raw.size() - pos - held >= kPkt) {
                if (raw[pos + held] == kSync) {
                    // a good sync ends a bad run: the held packets were bit errors in the sync byte
                    for (size_t q = 0; q < held; q += kPkt) emit(raw.data() + pos + q, true);
                    emit(raw.data() + pos + held, false);
                    pos += held + kPkt;
                    held = 0;
                    bad_run = 0;
                } else {
                    held += kPkt;
                    if (++bad_run >= loss_packets) {
                        // sync lost: hunt again from the byte after the first bad sync position
                        locked = false;
                        st.sync_losses++;
                        st.bytes_dropped += 1;
                        pos += 1;
                        held = 0;
                        bad_run = 0;
                        break;
                    }
                }
            }
            if (locked) break;
        }
        raw.erase(raw.begin(), raw.begin() + pos);
    }
};

extern "C" void atsc_ts_null_packet(uint8_t* out) {
    out[0] = kSync; out[1] = 0x1f; out[2] = 0xff; out[3] = 0x10;
    std::memset(out + 4, 0xff, kPkt - 4);
}

extern "C" atsc_ts_ingest* atsc_ts_ingest_create(int lock_packets, int loss_packets) {
    if (lock_packets == 0) lock_packets = 5;
    if (loss_packets == 0) loss_packets = 3;
    if (lock_packets < 2 || loss_packets < 1 || lock_packets > 4096 || loss_packets > 4096) return nullptr;
    auto* h = new (std::nothrow) atsc_ts_ingest();
    if (!h) return nullptr;
    h->lock_packets = lock_packets;
    h->loss_packets = loss_packets;
    return h;
}

extern "C" void atsc_ts_ingest_destroy(atsc_ts_ingest* h) { delete h; }

extern "C" size_t atsc_ts_ingest_available(const atsc_ts_ingest* h) {
    return h ? (h->ready.size() - h->ready_head) / kPkt : 0;
}

extern "C" size_t atsc_ts_ingest_push(atsc_ts_ingest* h, const uint8_t* data, size_t n) {
    if (!h) return 0;
    if (data && n) {
        h->st.bytes_in += n;
        try {   // no exception may cross the C boundary: on out-of-memory the chunk is dropped and counted as such
            h->raw.insert(h->raw.end(), data, data + n);
            h->run();
        } catch (const std::bad_alloc&) {
            h->st.bytes_dropped += n;
        }
    }
    return atsc_ts_ingest_available(h);
}

extern "C" size_t atsc_ts_ingest_pop(atsc_ts_ingest* h, uint8_t* out, size_t max_packets) {
    if (!h || !out) return 0;
    const size_t n = std::min(max_packets, atsc_ts_ingest_available(h));
    std::memcpy(out, h->ready.data() + h->ready_head, n * kPkt);
    h->ready_head += n * kPkt;
    h->st.packets_out += n;
    h->compact_ready();
    return n;
}

extern "C" size_t atsc_ts_ingest_pop_padded(atsc_ts_ingest* h, uint8_t* out, size_t packets) {
    if (!h || !out) return 0;
    const size_t got = atsc_ts_ingest_pop(h, out, packets);
    for (size_t i = got; i < packets; ++i) atsc_ts_null_packet(out + i * kPkt);
    h->st.nulls_stuffed += packets - got;
    h->st.packets_out += packets - got;
    return packets - got;
}

extern "C" void atsc_ts_ingest_get_stats(const atsc_ts_ingest* h, atsc_ts_ingest_stats* out) {
    if (!h || !out) return;
    *out = h->st;
    out->locked = h->locked ? 1 : 0;
}
