// Host rows -> device passes: the host half of the ndarray-in / ndarray-out call (hg_flow_execute; the call the reference
// makes at FaceDetectUpdated.py:699 hands over a host float64 ndarray, face_analysis.py:786).  Plain C++ (no HIP): the device
// side is behind PipeSink, so the whole machinery runs under ThreadSanitizer with a memcpy sink (tests/tsan_pipe_driver.cpp).
//
// Round 3 packed a chunk of 256 rows with all threads, THEN enqueued its copy and eight kernels with the pool idle, chunk by
// chunk: sixteen 105 us latency floors per 4096-row call and a packing rate of half the memory's.  Now three things run side by
// side and meet only through counters:
//   * workers (HostPool, on the memory node of the caller's rows) take row TICKETS in order and write wire rows — uint8 when
//     every value of the row is an integer 0..255 (exact narrowing, hg_hostpack.cpp), the caller's type otherwise — into a
//     pinned ring; a ticket may start once the ring rows it writes have left the host (`limit`);
//   * the driving thread (the caller's) follows the contiguous prefix of finished tickets and hands it to the copy queue in
//     PIECES (as soon as a few rows are ready: the DMA trails the packers by one piece, not by one chunk);
//   * rows are executed in PASSES whose sizes come from a small dynamic programme over a cost model (plan_passes): rows arrive
//     at a steady rate, a pass costs c0 + c1 rows, the last pass should be short because nothing hides it, and a batch of a few
//     hundred rows is one or two passes, not three chunks.
// A row that cannot be narrowed stops the call's narrow mode: passes launched so far stand, the rest of the call (from the start
// of the pass in work) is packed again in the caller's type (run() reports the row; hg_capi.cpp calls again with narrow off).
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <vector>

#include "hg_hostpool.hpp"

namespace hg {

bool narrow_row_f64(const double* src, uint8_t* dst, int64_t n);      // hg_hostpack.cpp
bool narrow_row_f32(const float* src, uint8_t* dst, int64_t n);
void stream_copy(void* dst, const void* src, size_t bytes);           // memcpy with non-temporal stores (write-combined destinations)
void store_fence();

// ---- pass sizes --------------------------------------------------------------------------------------------------------------
// Rows [0, r) have reached the device at  lat + a r  microseconds; a pass of m rows takes  c0 + c1 m  and passes run one after
// the other.  finish(r) = min over m of  max(finish(r - m), arrive(r)) + cost(m), plus a price per pass (see per_pass_us) that
// makes the programme prefer few long passes where short ones would only keep the device busy.  Unit: 16-row tiles, coarser for
// long calls so that the table stays a few hundred entries.
struct PassModel {
    double lat_us = 30, arrive_us_per_row = 0.45, c0_us = 64, c1_us_per_row = 0.17;
    double per_pass_us = 50;      // what one more pass is allowed to buy: a pass that ends the call less than this earlier is not worth its
                                  // launches (host time of the driving thread, a GPU kept busy with short grids)
    int64_t max_pass_rows = 1024;
};

inline std::vector<int64_t> plan_passes(int64_t n, const PassModel& m) {
    std::vector<int64_t> out;
    if (n <= 0) return out;
    // No pass may exceed max_pass_rows: the caller sizes its device and pinned buffers for exactly that many rows.  The unit grows
    // with n only while it stays below that bound (round 4 let it grow past it: a call of more wire bytes than ~8 GiB planned
    // passes wider than their buffers — ADVICE r4); for long calls with a small bound the table simply has more entries.
    const int64_t cap = std::max<int64_t>(1, m.max_pass_rows);
    int64_t unit = 16 * std::max<int64_t>(1, (n + 16 * 512 - 1) / (16 * 512));
    if (unit > cap) unit = cap >= 16 ? cap / 16 * 16 : cap;
    const int64_t U = (n + unit - 1) / unit;
    const int64_t span = std::max<int64_t>(1, std::min<int64_t>(64, cap / unit));
    auto rows = [&](int64_t i) { return std::min(i * unit, n); };
    std::vector<double> fin((size_t)U + 1, 0.0);
    std::vector<int32_t> cnt((size_t)U + 1, 0);
    std::vector<int64_t> from((size_t)U + 1, 0);
    for (int64_t i = 1; i <= U; ++i) {
        const double arr = m.lat_us + m.arrive_us_per_row * (double)rows(i);
        double best = std::numeric_limits<double>::infinity();
        int64_t bj = i - 1;
        for (int64_t j = std::max<int64_t>(0, i - span); j < i; ++j) {
            const double t = std::max(fin[(size_t)j], arr) + m.c0_us + m.c1_us_per_row * (double)(rows(i) - rows(j)) + m.per_pass_us * (cnt[(size_t)j] + 1);
            if (t < best - 1e-9) {      // ties: the earlier j (the longer pass) wins
                best = t;
                bj = j;
            }
        }
        from[(size_t)i] = bj;
        cnt[(size_t)i] = cnt[(size_t)bj] + 1;
        fin[(size_t)i] = best - m.per_pass_us * cnt[(size_t)i];
    }
    for (int64_t i = U; i > 0; i = from[(size_t)i]) out.push_back(rows(i) - rows(from[(size_t)i]));
    std::reverse(out.begin(), out.end());
    for (int64_t p : out)
        if (p <= 0 || p > cap) throw std::logic_error("plan_passes: a pass outside 1..max_pass_rows");
    return out;
}

// ---- the device side, as the pipe sees it ------------------------------------------------------------------------------------
// All calls come from the driving thread, in this order per pass: copy()* then launch().  The sink rotates its own device /
// feature buffers and waits (on its queues, or on the host where it must) before it reuses one.
struct PipeSink {
    virtual ~PipeSink() {}
    virtual void copy(int pass, int64_t dst_row, const void* src, int64_t rows) = 0;   // wire rows -> row dst_row of the pass's input buffer
    virtual void launch(int pass, int64_t r0, int64_t rows) = 0;                       // every row of the pass has been copied: kernels, features back
    virtual uint64_t mark() = 0;                                                       // a point in the copy queue behind everything enqueued so far
    virtual bool reached(uint64_t mark) = 0;                                           // has the copy queue passed it? (polled)
    virtual bool finished(int pass) { return true; }                                   // direct mode: has a LAUNCHED pass completed? (polled)
};

// Optional timeline of one run_pipe invocation (HIGSFA_HOST_TRACE=1; microseconds since `t0`)
struct PipeTrace {
    std::chrono::steady_clock::time_point t0;
    std::vector<float> ticket_done;                                  // per ticket, written by the worker that packed it
    struct Piece { float t_begin, t_end; int pass; int64_t rows; };      // host time around sink.copy
    struct Pass { float t_begin, t_end; int64_t rows; };                 // host time around sink.launch
    std::vector<Piece> pieces;
    std::vector<Pass> passes;
    float now() const { return std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

struct PipeJob {
    const void* x = nullptr;      // caller rows
    int elem = 8;                 // bytes per value of the caller's type (1, 4, 8)
    bool is_float = true;         // 4 / 8-byte values are float32 / float64
    int64_t n = 0, ldx = 0, in_dim = 0;
    bool narrow = false;          // try to send uint8 (only for 4 / 8-byte input)
    uint8_t* ring = nullptr;      // pinned staging
    size_t ring_bytes = 0;
    std::vector<int64_t> passes;  // rows per pass (plan_passes), sum == n
    int64_t piece_min = 64, piece_max = 256;      // rows handed to the copy queue at a time: at least / at most
    double copy_us_per_row = 0.29, copy_fixed_us = 7;      // what the pipe assumes a copy costs (only to decide whether the link is busy)
    // Direct mode (large-BAR devices: the host can store into device memory): pass p's rows are written straight to pass_dst[p] —
    // the pass's input buffer in HBM, mapped write-combining — and there is no ring and no copy queue; pass_dst[p] may be
    // written once pass p - direct_slots has finished (PipeSink::finished), the buffers rotating over direct_slots slots.
    std::vector<uint8_t*> pass_dst;
    int direct_slots = 0;
    size_t ticket_bytes = 512u << 10;      // caller bytes per ticket.  Rows that are only copied are bound by the link, which the packers share:
                                           // k of them finish their tickets together after k tickets' worth of link time — small tickets
                                           // (128 KiB) let the first pass start early
    int max_workers = 1 << 30;             // packers that take tickets (HostPool::begin)
    PipeTrace* trace = nullptr;
};

struct PipeResult {
    int64_t rows_done = 0;        // rows of fully launched passes (== n unless narrowing failed)
    int passes_done = 0;
    bool narrow_failed = false;
};

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// Runs the pipeline for one mode (narrow or as given).  `pool` may be null or `inline_pack` set: the driving thread packs every
// ticket itself first (tiny calls: waking workers costs more than the rows).
inline PipeResult run_pipe(const PipeJob& J, PipeSink& sink, HostPool* pool, bool inline_pack) {
    PipeResult res;
    if (J.n <= 0) return res;
    const size_t wire = J.narrow ? 1 : (size_t)J.elem;
    const size_t row_src = (size_t)J.ldx * J.elem, row_wire = (size_t)J.in_dim * wire;
    const int64_t ring_rows = J.pass_dst.empty() ? (int64_t)(J.ring_bytes / row_wire) : std::numeric_limits<int64_t>::max() / 4;
    // a ticket: ticket_bytes of caller bytes, at least one row, never across a pass boundary (in direct mode a ticket then needs only
    // its own pass's buffer, whatever the pass sizes are)
    const int64_t ticket_rows = std::max<int64_t>(1, std::min<int64_t>((int64_t)(J.ticket_bytes / std::max<size_t>(1, (size_t)J.in_dim * J.elem)), ring_rows / 4));
    std::vector<int64_t> tk_first;      // first row of every ticket, and n at the end
    {
        int64_t r = 0;
        for (size_t p = 0; p <= J.passes.size() && r < J.n; ++p) {
            const int64_t pe = p < J.passes.size() ? std::min(J.n, r + J.passes[p]) : J.n;
            for (; r < pe; r = std::min(pe, r + ticket_rows)) tk_first.push_back(r);
        }
        tk_first.push_back(J.n);
    }
    const int64_t n_tickets = (int64_t)tk_first.size() - 1;
    if (n_tickets > (int64_t)std::numeric_limits<int32_t>::max()) throw std::runtime_error("host pipe: too many tickets");
    std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[(size_t)n_tickets]);
    for (int64_t t = 0; t < n_tickets; ++t) done[(size_t)t].store(0, std::memory_order_relaxed);
    std::atomic<int64_t> limit{std::min(J.n, ring_rows)};      // rows below it may be written into the ring
    std::atomic<int64_t> fail_row{J.n};
    std::atomic<bool> stop{false};
    if (J.trace) J.trace->ticket_done.assign((size_t)n_tickets, -1.f);

    const bool direct = !J.pass_dst.empty();
    std::vector<int64_t> pass_first;      // direct mode: first row of every pass (and n at the end)
    if (direct) {
        if (J.pass_dst.size() != J.passes.size() || J.direct_slots < 1) throw std::runtime_error("host pipe: direct mode needs one destination per pass");
        pass_first.push_back(0);
        for (int64_t m : J.passes) pass_first.push_back(pass_first.back() + m);
        limit.store(pass_first[std::min<size_t>(J.passes.size(), (size_t)J.direct_slots)], std::memory_order_relaxed);
    }

    const std::function<void(int)> ticket = [&](int t) {
        const int64_t a = tk_first[(size_t)t], e = tk_first[(size_t)t + 1];
        for (unsigned spins = 0; e > limit.load(std::memory_order_acquire); ++spins) {      // the rows' destination is still in use
            if (stop.load(std::memory_order_relaxed)) return;
            if (spins < 64) cpu_relax();
            else sched_yield();
        }
        if (stop.load(std::memory_order_relaxed)) return;
        const char* src = (const char*)J.x + (size_t)a * row_src;
        size_t p = direct ? (size_t)(std::upper_bound(pass_first.begin(), pass_first.end(), a) - pass_first.begin()) - 1 : 0;
        auto dst_of = [&](int64_t r) -> uint8_t* {
            if (!direct) return J.ring + (size_t)(r % ring_rows) * row_wire;
            while (r >= pass_first[p + 1]) ++p;
            return J.pass_dst[p] + (size_t)(r - pass_first[p]) * row_wire;
        };
        if (J.narrow) {
            for (int64_t r = a; r < e; ++r, src += row_src) {
                uint8_t* dst = dst_of(r);
                const bool ok = J.elem == 8 ? narrow_row_f64((const double*)src, dst, J.in_dim) : narrow_row_f32((const float*)src, dst, J.in_dim);
                if (!ok) {
                    int64_t cur = fail_row.load(std::memory_order_relaxed);
                    while (a < cur && !fail_row.compare_exchange_weak(cur, a, std::memory_order_release, std::memory_order_relaxed)) {}
                    return;      // done[t] stays 0: the driving thread never goes past this ticket
                }
            }
        } else if (direct) {
            for (int64_t r = a; r < e; ++r, src += row_src) stream_copy(dst_of(r), src, row_wire);
        } else {
            const int64_t ra = a % ring_rows;
            if (J.ldx == J.in_dim && ra + (e - a) <= ring_rows) memcpy(J.ring + (size_t)ra * row_wire, src, (size_t)(e - a) * row_wire);
            else for (int64_t r = a; r < e; ++r, src += row_src) memcpy(J.ring + (size_t)(r % ring_rows) * row_wire, src, row_wire);
        }
        if (direct) store_fence();      // write-combining buffers drained before anybody is told the rows are there
        if (J.trace) J.trace->ticket_done[(size_t)t] = J.trace->now();
        done[(size_t)t].store(1, std::memory_order_release);
    };

    const bool use_pool = pool && pool->size() > 0 && !inline_pack;
    struct Region {      // the workers must have left `ticket` before anything it captures goes away, whatever happens
        HostPool* p;
        std::atomic<bool>& stop;
        bool open;
        void close() {
            if (!open) return;
            open = false;
            p->end();
        }
        ~Region() {
            if (!open) return;
            stop.store(true);
            try { p->end(); } catch (...) {}
        }
    } region{pool, stop, false};
    if (use_pool) {
        pool->begin((int)n_tickets, ticket, J.max_workers);
        region.open = true;
    }

    struct Piece { uint64_t mark; int64_t upto; };
    std::vector<Piece> flying;      // copies whose ring rows are still needed, oldest first (only when the ring is shorter than the call)
    size_t fly_head = 0;
    const bool ring_wraps = !direct && ring_rows < J.n;
    int64_t next_ticket = 0, packed = 0, sent = 0;
    int pass = 0, limit_pass = direct ? (int)std::min<size_t>(J.passes.size(), (size_t)J.direct_slots) : 0;
    int64_t pass_start = 0, pass_end = J.passes.empty() ? J.n : J.passes[0];
    const int64_t piece_min = std::max<int64_t>(1, std::min(J.piece_min, ring_rows / 4));
    const int64_t piece_max = std::max<int64_t>(piece_min, std::min(J.piece_max, ring_rows / 2));
    // While the link is still busy with earlier pieces (by the cost model: the pipe never asks the device), rows that become ready
    // wait and leave as ONE larger copy — a call bound by PCIe sends few large copies at the link's best rate, a call bound by
    // packing sends each piece as soon as it is ready and the device trails the packers by one small copy.
    const auto clk0 = std::chrono::steady_clock::now();
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - clk0).count(); };
    double link_free_us = 0;
    while (sent < J.n) {
        if (!use_pool && next_ticket < n_tickets && !res.narrow_failed &&
            tk_first[(size_t)next_ticket + 1] <= limit.load(std::memory_order_relaxed))
            ticket((int)next_ticket);      // inline: pack the next ticket ourselves (never while it would have to wait for ring rows)
        while (next_ticket < n_tickets && done[(size_t)next_ticket].load(std::memory_order_acquire)) ++next_ticket;
        packed = tk_first[(size_t)next_ticket];
        // narrowing failed somewhere, and every ticket before the failed one is finished (the failed one never will be)
        if (fail_row.load(std::memory_order_acquire) < J.n && packed >= fail_row.load(std::memory_order_relaxed)) res.narrow_failed = true;
        if (ring_wraps) {      // copies that have left the host free ring rows
            bool moved = false;
            while (fly_head < flying.size() && sink.reached(flying[fly_head].mark)) {
                ++fly_head;
                moved = true;
            }
            if (moved) limit.store(std::min(J.n, flying[fly_head - 1].upto + ring_rows), std::memory_order_release);
        }
        if (direct) {
            // buffers of finished passes take the rows of later ones; a pass is launched the moment its last ticket is in
            while (limit_pass < (int)J.passes.size() && limit_pass - J.direct_slots < pass && sink.finished(limit_pass - J.direct_slots)) {
                ++limit_pass;
                limit.store(pass_first[(size_t)limit_pass], std::memory_order_release);
            }
            if (packed >= pass_end) {
                const float lb = J.trace ? J.trace->now() : 0.f;
                sink.launch(pass, pass_start, pass_end - pass_start);
                if (J.trace) J.trace->passes.push_back(PipeTrace::Pass{lb, J.trace->now(), pass_end - pass_start});
                sent = pass_end;
                res.rows_done = pass_end;
                res.passes_done = ++pass;
                pass_start = pass_end;
                pass_end = pass < (int)J.passes.size() ? pass_start + J.passes[(size_t)pass] : J.n;
                continue;
            }
            if (res.narrow_failed) break;
            if (use_pool) cpu_relax();
            continue;
        }
        const int64_t avail = std::min(packed, pass_end);
        // send at the end of a pass, when a full piece is ready, when the packers have filled the ring and wait for us, or when a
        // small piece is ready and the link is (about to be) idle
        if (avail > sent && (avail == pass_end || avail - sent >= piece_max || packed >= limit.load(std::memory_order_relaxed) ||
                             (avail - sent >= piece_min && now_us() >= link_free_us - 20.0))) {
            int64_t m = std::min(avail - sent, piece_max);
            m = std::min(m, ring_rows - sent % ring_rows);      // a piece does not wrap around the ring
            const float tb = J.trace ? J.trace->now() : 0.f;
            sink.copy(pass, sent - pass_start, J.ring + (size_t)(sent % ring_rows) * row_wire, m);
            if (J.trace) J.trace->pieces.push_back(PipeTrace::Piece{tb, J.trace->now(), pass, m});
            link_free_us = std::max(link_free_us, now_us()) + J.copy_fixed_us + J.copy_us_per_row * (double)m;
            sent += m;
            if (ring_wraps) flying.push_back(Piece{sink.mark(), sent});
            if (sent == pass_end) {
                const float lb = J.trace ? J.trace->now() : 0.f;
                sink.launch(pass, pass_start, pass_end - pass_start);
                if (J.trace) J.trace->passes.push_back(PipeTrace::Pass{lb, J.trace->now(), pass_end - pass_start});
                res.rows_done = pass_end;
                res.passes_done = ++pass;
                pass_start = pass_end;
                pass_end = pass < (int)J.passes.size() ? pass_start + J.passes[(size_t)pass] : J.n;
            }
            continue;
        }
        if (res.narrow_failed) break;      // nothing more can be sent in this mode
        if (use_pool) cpu_relax();
    }
    stop.store(true, std::memory_order_relaxed);
    region.close();
    return res;
}

}  // namespace hg
