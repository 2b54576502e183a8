// ThreadSanitizer / plain driver for the host pipeline (pyfaceanalysis_amd/csrc/hg_hostpipe.hpp) with a memcpy sink in the
// place of the GPU: packers, the driving thread and a copy queue that lags behind meet only through the pipe's counters.
// Every row's "feature" is the sum of its values computed from the WIRE rows the sink received; a row packed twice into a
// live ring slot, a piece sent before its tickets finished, a pass launched with rows missing or a wrong fall-through
// boundary shows up as a wrong sum (or as a data race report under TSAN).
#include <cmath>
#include <cstdio>
#include <deque>
#include <random>
#include <vector>

#include "hg_hostpipe.hpp"

struct FakeSink final : hg::PipeSink {
    size_t row_wire = 0;
    int wire_elem = 1;
    int64_t in_dim = 0, row_base = 0;
    int lag;                                   // a mark is reached only after this many polls: the copy queue trails the host
    std::vector<std::vector<uint8_t>> dev;     // per pass: the rows copied so far
    std::vector<double>& y;
    std::vector<int64_t> pass_sizes;
    std::deque<std::pair<uint64_t, int>> marks;
    uint64_t seq = 0;
    int launched = 0;
    // the ring rows of a copy stay "in flight" until its mark is reached: the sink keeps the pointer and re-reads it then
    struct Pending { uint64_t mark; int pass; int64_t dst_row; const uint8_t* src; int64_t rows; };
    std::deque<Pending> pending;
    bool deferred;                             // true: bytes are read from the ring only when the mark is reached (as a DMA would)
    // direct mode: the packers write into slot buffers themselves; a launched pass "runs" for a few polls, during which its
    // slot must not be written — the rows are summed when the pass FINISHES, so a packer that got in early shows as wrong sums
    std::vector<std::vector<uint8_t>> slots;
    struct Running { int pass; int64_t r0, rows; int polls; };
    std::deque<Running> running;
    int finished_upto = 0;
    size_t slot_off = 0;                       // the passes' rows start this far into a slot (unaligned destinations)

    FakeSink(std::vector<double>& y_, int lag_, bool deferred_) : lag(lag_), y(y_), deferred(deferred_) {}
    void do_copy(int pass, int64_t dst_row, const uint8_t* src, int64_t rows) {
        if ((int)dev.size() <= pass) dev.resize((size_t)pass + 1);
        auto& d = dev[(size_t)pass];
        if (d.size() < (size_t)(dst_row + rows) * row_wire) d.resize((size_t)(dst_row + rows) * row_wire);
        memcpy(d.data() + (size_t)dst_row * row_wire, src, (size_t)rows * row_wire);
    }
    void copy(int pass, int64_t dst_row, const void* src, int64_t rows) override {
        if (deferred) pending.push_back(Pending{seq + 1, pass, dst_row, (const uint8_t*)src, rows});
        else do_copy(pass, dst_row, (const uint8_t*)src, rows);
    }
    void flush(uint64_t upto) {
        while (!pending.empty() && pending.front().mark <= upto) {
            const Pending& p = pending.front();
            do_copy(p.pass, p.dst_row, p.src, p.rows);
            pending.pop_front();
        }
    }
    void sum_rows(const uint8_t* d, int64_t r0, int64_t rows) {
        for (int64_t r = 0; r < rows; ++r) {
            double s = 0;
            const uint8_t* p = d + (size_t)r * row_wire;
            for (int64_t c = 0; c < in_dim; ++c)
                s += wire_elem == 1 ? (double)p[c] : wire_elem == 4 ? (double)((const float*)p)[c] : ((const double*)p)[c];
            y[(size_t)(row_base + r0 + r)] = s;
        }
    }
    void retire(bool all) {
        while (!running.empty() && (all || --running.front().polls <= 0)) {
            const Running& q = running.front();
            sum_rows(slots[(size_t)q.pass % slots.size()].data() + slot_off, q.r0, q.rows);
            finished_upto = q.pass + 1;
            running.pop_front();
            if (!all) break;
        }
    }
    bool finished(int pass) override {
        if (pass < finished_upto) return true;
        retire(false);
        return pass < finished_upto;
    }
    void launch(int pass, int64_t r0, int64_t rows) override {
        if (!slots.empty()) {
            running.push_back(Running{pass, r0, rows, lag});
            pass_sizes.push_back(rows);
            ++launched;
            return;
        }
        flush(~0ull);      // the compute queue waits for the copy queue
        const auto& d = dev[(size_t)pass];
        if ((int64_t)(d.size() / row_wire) != rows) throw std::runtime_error("pass launched with the wrong number of rows");
        for (int64_t r = 0; r < rows; ++r) {
            double s = 0;
            const uint8_t* p = d.data() + (size_t)r * row_wire;
            for (int64_t c = 0; c < in_dim; ++c)
                s += wire_elem == 1 ? (double)p[c] : wire_elem == 4 ? (double)((const float*)p)[c] : ((const double*)p)[c];
            y[(size_t)(row_base + r0 + r)] = s;
        }
        pass_sizes.push_back(rows);
        ++launched;
    }
    uint64_t mark() override {
        marks.emplace_back(++seq, lag);
        return seq;
    }
    bool reached(uint64_t m) override {
        while (!marks.empty() && marks.front().first < m) marks.pop_front();
        if (marks.empty() || marks.front().first != m) return true;
        if (--marks.front().second > 0) return false;
        flush(m);
        marks.pop_front();
        return true;
    }
};

template <typename T>
static int run_case(hg::HostPool* pool, bool inline_pack, int64_t n, int64_t in_dim, int64_t ldx, size_t ring_bytes, int64_t bad_row, bool narrow,
                    int lag, unsigned seed, const char* what, int direct_slots = 0) {
    if (getenv("PIPE_DRIVER_VERBOSE")) fprintf(stderr, "%s seed %u n %lld in_dim %lld ldx %lld ring %zu bad %lld inline %d lag %d slots %d\n", what, seed, (long long)n, (long long)in_dim, (long long)ldx, ring_bytes, (long long)bad_row, (int)inline_pack, lag, direct_slots);
    std::mt19937 rng(seed);
    std::vector<T> x((size_t)n * ldx);
    for (auto& v : x) v = (T)(rng() & 255);
    if (bad_row >= 0) x[(size_t)bad_row * ldx + (size_t)(rng() % in_dim)] = (T)17.5;
    std::vector<double> want((size_t)n), got((size_t)n, -1.0);
    for (int64_t r = 0; r < n; ++r) {
        double s = 0;
        for (int64_t c = 0; c < in_dim; ++c) s += (double)x[(size_t)r * ldx + (size_t)c];
        want[(size_t)r] = s;
    }
    std::vector<uint8_t> ring(ring_bytes);
    FakeSink sink(got, lag, true);
    sink.in_dim = in_dim;
    int invocations = 0;
    bool nar = narrow && sizeof(T) > 1;
    for (int64_t r0 = 0; r0 < n; ++invocations) {
        hg::PipeJob J;
        J.x = x.data() + (size_t)r0 * ldx;
        J.elem = (int)sizeof(T);
        J.n = n - r0;
        J.ldx = ldx;
        J.in_dim = in_dim;
        J.narrow = nar;
        J.ring = ring.data();
        J.ring_bytes = ring.size();
        hg::PassModel M;
        M.max_pass_rows = 16 * (1 + (int64_t)(rng() % 40));
        M.arrive_us_per_row = 0.1 + (rng() % 100) * 0.01;
        J.passes = hg::plan_passes(J.n, M);
        int64_t sum = 0;
        for (auto p : J.passes) {
            sum += p;
            if (p <= 0 || p > M.max_pass_rows) { printf("%s: bad pass size %lld\n", what, (long long)p); return 1; }
        }
        if (sum != J.n) { printf("%s: plan covers %lld of %lld rows\n", what, (long long)sum, (long long)J.n); return 1; }
        J.piece_min = 1 + (int64_t)(rng() % 40);
        J.piece_max = J.piece_min + (int64_t)(rng() % 100);
        sink.wire_elem = nar ? 1 : (int)sizeof(T);
        sink.row_wire = (size_t)in_dim * sink.wire_elem;
        sink.row_base = r0;
        sink.dev.clear();
        sink.pending.clear();
        sink.marks.clear();
        if (direct_slots) {
            sink.slots.assign((size_t)direct_slots, std::vector<uint8_t>((size_t)M.max_pass_rows * in_dim * sizeof(T) + 64));
            sink.finished_upto = 0;
            sink.slot_off = (seed % 3) * 16;
            J.direct_slots = direct_slots;
            for (size_t k = 0; k < J.passes.size(); ++k) J.pass_dst.push_back(sink.slots[k % (size_t)direct_slots].data() + (seed % 3) * 16);
        }
        const hg::PipeResult res = hg::run_pipe(J, sink, pool, inline_pack);
        if (direct_slots) sink.retire(true);
        r0 += res.rows_done;
        if (!res.narrow_failed) {
            if (r0 != n) { printf("%s: pipe stopped at row %lld of %lld\n", what, (long long)r0, (long long)n); return 1; }
            break;
        }
        if (!nar) { printf("%s: failure reported in wide mode\n", what); return 1; }
        if (bad_row < 0 || r0 > bad_row) { printf("%s: narrow failure handling went past the bad row (%lld > %lld)\n", what, (long long)r0, (long long)bad_row); return 1; }
        nar = false;
    }
    int bad = 0;
    for (int64_t r = 0; r < n; ++r) bad += want[(size_t)r] != got[(size_t)r];
    if (bad) printf("%s: %d of %lld rows wrong\n", what, bad, (long long)n);
    if (bad_row >= 0 && narrow && sizeof(T) > 1 && invocations != 1) { printf("%s: expected one fall-through, saw %d\n", what, invocations); return 1; }
    return bad != 0;
}

int main() {
    hg::HostPool pool(6);
    int bad = 0, cases = 0;
    for (unsigned seed = 1; seed <= 40; ++seed) {
        std::mt19937 rng(seed * 7919u);
        const int64_t in_dim = 16 * (1 + rng() % 24), ldx = in_dim + (rng() % 3 == 0 ? rng() % 9 : 0);
        const int64_t n = 1 + rng() % 3000;
        // rings from "much shorter than the call" (wraps many times) to "holds the call"
        const size_t ring = (size_t)in_dim * 8 * (64 + rng() % (seed % 4 == 0 ? 6000 : 300));
        const int lag = 1 + rng() % 6;
        const bool inl = seed % 5 == 0;
        const int64_t bad_row = seed % 3 == 0 ? (int64_t)(rng() % n) : -1;
        bad += run_case<double>(&pool, inl, n, in_dim, ldx, ring, bad_row, true, lag, seed, "float64 narrow");
        bad += run_case<float>(&pool, inl, n, in_dim, ldx, ring, bad_row, true, lag, seed + 100, "float32 narrow");
        bad += run_case<double>(&pool, inl, n, in_dim, ldx, ring, -1, false, lag, seed + 200, "float64 as given");
        bad += run_case<uint8_t>(&pool, inl, n, in_dim, ldx, ring, -1, false, lag, seed + 300, "uint8");
        const int slots = 1 + (int)(rng() % 4);
        bad += run_case<double>(&pool, inl, n, in_dim, ldx, ring, bad_row, true, lag, seed + 400, "float64 narrow, direct", slots);
        bad += run_case<float>(&pool, inl, n, in_dim, ldx, ring, -1, false, lag, seed + 500, "float32 as given, direct", slots);
        bad += run_case<uint8_t>(&pool, inl, n, in_dim, ldx, ring, -1, false, lag, seed + 600, "uint8, direct", slots);
        cases += 7;
    }
    // no pool at all
    bad += run_case<double>(nullptr, false, 777, 64, 64, 64 * 8 * 80, 400, true, 2, 99, "no pool");
    // planner: the preset's cases are one or two passes for a frame's largest batch and long passes in the middle of a big batch
    hg::PassModel M;
    const auto p728 = hg::plan_passes(728, M);
    const auto p4096 = hg::plan_passes(4096, M);
    int64_t longest = 0;
    for (auto p : p4096) longest = std::max(longest, p);
    if (p728.size() > 3 || p4096.size() > 12 || longest < 512 || p4096.back() > 512) {
        printf("planner: unexpected shape (%zu passes for 728 rows, %zu for 4096, longest %lld, last %lld)\n", p728.size(), p4096.size(),
               (long long)longest, (long long)p4096.back());
        ++bad;
    }
    // planner bound: whatever the call's length, no pass is wider than max_pass_rows — the device and pinned buffers
    // of a call are sized for exactly that many rows — and the passes cover the call
    {
        const int64_t ns[] = {1, 15, 16, 17, 4096, 8191, 8192, 8193, 9000, 65536, 70000, 140000, 1000003, 5000000};
        const int64_t caps[] = {1, 7, 16, 17, 100, 128, 144, 256, 1024, 65536};
        for (int64_t n : ns)
            for (int64_t cap : caps) {
                hg::PassModel Q;
                Q.max_pass_rows = cap;
                const auto ps = hg::plan_passes(n, Q);
                int64_t sum = 0, widest = 0;
                for (auto p : ps) {
                    sum += p;
                    widest = std::max(widest, p);
                    if (p <= 0) widest = cap + 1;
                }
                if (sum != n || widest > cap) {
                    printf("planner: n %lld max_pass_rows %lld -> %zu passes, sum %lld, widest %lld\n", (long long)n, (long long)cap, ps.size(), (long long)sum, (long long)widest);
                    ++bad;
                }
            }
    }
    // topology helpers of the pool: whatever this machine exposes, the groups partition the node's CPUs and binding does not break a region
    {
        const std::vector<int> cpus = hg::cpus_of_node(0);
        const auto groups = hg::group_by_llc(cpus);
        size_t covered = 0;
        for (const auto& g : groups) covered += g.size();
        if (!cpus.empty() && (groups.empty() || covered != cpus.size())) {
            printf("topology: %zu cpus on node 0 but the cache groups cover %zu\n", cpus.size(), covered);
            ++bad;
        }
        if (hg::usable_cpus() < 1) ++bad;
        for (int node : {0, -1, 7, 0}) {      // an existing node, "anywhere", a node that (probably) does not exist, back
            pool.bind_to_node(node);
            bad += run_case<float>(&pool, false, 500, 64, 64, 64 * 4 * 100, -1, true, 2, 1000 + node, "after bind_to_node");
        }
    }
    printf("cases %d bad %d\n", cases + 5, bad);
    return bad != 0;
}
