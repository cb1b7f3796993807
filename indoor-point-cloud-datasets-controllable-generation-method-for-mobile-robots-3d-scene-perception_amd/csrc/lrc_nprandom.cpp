// lrc_nprandom.cpp -- numpy's legacy seeded stream (RandomState: MT19937 + random_sample + polar-method normals),
// restated so that a whole trajectory's draws of the dual-axis sensor are produced at memory speed (host code, product).
//
// The reference's BLK2GO generator draws, per pose and from the GLOBAL numpy stream, two normals per ray (phi noise,
// theta noise) and then one uniform per ray (dropout): lidar/indoor_lidar.py:257-296 (128 000
// normals + 64 000 uniforms per pose).  "Noise seeded identically" (BASELINE north_star) therefore means: the same
// doubles numpy would hand out, in the same order, and the same generator state afterwards.  What numpy computes:
//   word      MT19937 (32-bit), regenerated in blocks of 624                      numpy/random/src/mt19937/mt19937.c
//   uniform   ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two consecutive words      mt19937_next_double / legacy_double
//   normal    polar method: x1 = 2u - 1, x2 = 2u' - 1, r2 = x1^2 + x2^2, rejected while r2 >= 1 or r2 == 0;
//             f = sqrt(-2 log(r2) / r2); returns f * x2 and caches f * x1 for the next call
//             (legacy-distributions.c: legacy_gauss); normal(loc, scale) = loc + scale * gauss
// Only the word stream is sequential.  An attempt of the polar method consumes exactly two uniforms = four words, and with
// an even number of normals and of uniforms per pose every attempt of every pose starts on a multiple of four words from
// the call's first word.  The streaming path (scan_streaming) therefore splits the work three ways:
//   * ONE generator thread produces tempered words into 1 MiB chunks and does nothing else (0.4 ns per word);
//   * worker threads compute, per chunk, for EVERY aligned group of four words whether it would be an accepted attempt
//     (a quarter of the groups turn out to be uniforms: their flags are never looked at) and how many a chunk holds;
//   * the calling thread walks the counts to find where each pose's normals end (whole chunks by their count, the last
//     one by its flags), and hands the pose -- log, sqrt, division per accepted attempt, with THIS process's libm, as
//     numpy calls it, then the uniforms -- to the workers.
// Odd counts or a cached normal at entry take the sequential path (scan_sequential: the acceptance pass on the producer).
// Pinned by tests/test_nprandom.py (-m "not gpu") against numpy itself, and through the frames by the G3 goldens.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int code, const char* msg);

namespace {

// Thread starts can fail (EAGAIN at a pid / thread limit -> std::system_error).  Every start goes through try_start(): a
// failed start returns false and leaves the vector as it was, so that the caller can go on with fewer threads -- never an
// exception that unwinds past joinable threads (std::terminate).  g_thread_budget >= 0 is a test hook
// (lrc_internal_set_thread_budget): that many further starts succeed, the next ones fail as the system would make them.
std::atomic<long> g_thread_budget{-1};

template <class F>
bool try_start(std::vector<std::thread>& pool, F&& f) {
    long b = g_thread_budget.load();
    while (b >= 0) {
        if (b == 0) return false;
        if (g_thread_budget.compare_exchange_weak(b, b - 1)) break;
    }
    try {
        pool.emplace_back(std::forward<F>(f));
        return true;
    } catch (const std::system_error&) {
        return false;
    } catch (const std::bad_alloc&) {
        return false;
    }
}

// joins on every exit: `wind_down` makes the threads return (called first), then they are joined
template <class W>
struct JoinGuard {
    std::vector<std::thread>& pool;
    W wind_down;
    ~JoinGuard() {
        wind_down();
        for (auto& t : pool) if (t.joinable()) t.join();
    }
};
template <class W> JoinGuard<W> make_join_guard(std::vector<std::thread>& pool, W w) { return JoinGuard<W>{pool, std::move(w)}; }

struct NoGeneratorThread {};      // the streaming path could not start its generator thread: take the sequential path

constexpr int kN = 624, kM = 397;
constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;

// next block of 624 raw words from the previous one (out of place: the loops carry no dependence shorter than 227)
__attribute__((target_clones("avx2", "default")))
void mt_next_block(const uint32_t* __restrict__ old, uint32_t* __restrict__ nw) {
    for (int i = 0; i < kN - kM; ++i) {
        const uint32_t y = (old[i] & kUpper) | (old[i + 1] & kLower);
        nw[i] = old[i + kM] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
    }
    for (int i = kN - kM; i < kN - 1; ++i) {
        const uint32_t y = (old[i] & kUpper) | (old[i + 1] & kLower);
        nw[i] = nw[i - (kN - kM)] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
    }
    const uint32_t y = (old[kN - 1] & kUpper) | (nw[0] & kLower);
    nw[kN - 1] = nw[kM - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
}

inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__attribute__((target_clones("avx2", "default")))
void temper_block(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = temper(raw[i]);
}

inline double to_double(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// accepted attempts among the first `attempts` attempts of w (4 tempered words each); stops early once `need` are found:
// returns the number of attempts scanned up to and including the need-th accepted one (or `attempts` if not reached)
__attribute__((target_clones("avx2", "default")))
size_t count_block(const uint32_t* __restrict__ w, size_t attempts, uint32_t* __restrict__ flags) {
    size_t acc = 0;
    for (size_t k = 0; k < attempts; ++k) {
        const double x1 = 2.0 * to_double(w[4 * k], w[4 * k + 1]) - 1.0;
        const double x2 = 2.0 * to_double(w[4 * k + 2], w[4 * k + 3]) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        const uint32_t ok = (r2 >= 1.0 || r2 == 0.0) ? 0u : 1u;
        flags[k] = ok;
        acc += ok;
    }
    return acc;
}

// The numpy-compatible generator: `key` is the current block (raw words), `pos` the next word in it.
struct Stream {
    uint32_t key[kN];
    int pos;
    // tempered words [0, avail) not yet handed out, in stream order; block bookkeeping so that the state after any
    // number of consumed words can be reconstructed: raw blocks generated beyond `key` are kept until consumed
    std::vector<uint32_t> words;        // tempered, from the consumption point on
    size_t head = 0;                    // consumption point inside `words`
    std::deque<std::vector<uint32_t>> ahead;   // raw blocks generated after `key`, oldest first

    void init(const uint32_t* k, int p) {
        std::memcpy(key, k, sizeof(key));
        pos = p;
        words.clear();
        head = 0;
        ahead.clear();
        // the rest of the current block is available without generating anything
        if (pos < kN) {
            words.resize((size_t)(kN - pos));
            temper_block(key + pos, words.data(), (size_t)(kN - pos));
        }
    }
    const uint32_t* last_raw() const { return ahead.empty() ? key : ahead.back().data(); }
    // make at least n words available from the consumption point
    void ensure(size_t n) {
        if (words.size() - head >= n) return;
        if (head > 0 && head >= words.size() / 2) {          // drop what was consumed
            words.erase(words.begin(), words.begin() + (ptrdiff_t)head);
            head = 0;
        }
        while (words.size() - head < n) {
            std::vector<uint32_t> blk((size_t)kN);
            mt_next_block(last_raw(), blk.data());
            const size_t at = words.size();
            words.resize(at + (size_t)kN);
            temper_block(blk.data(), words.data() + at, (size_t)kN);
            ahead.push_back(std::move(blk));
        }
    }
    const uint32_t* peek() const { return words.data() + head; }
    // n words have been consumed: advance, and move `key` / `pos` to the block the consumption point is in
    void consume(size_t n) {
        head += n;
        size_t p = (size_t)pos + n;
        while (p > (size_t)kN) {           // numpy regenerates lazily: pos == 624 stays on the old block
            std::memcpy(key, ahead.front().data(), sizeof(key));
            ahead.pop_front();
            p -= (size_t)kN;
        }
        pos = (int)p;
    }
};

struct PoseJob {
    std::vector<uint32_t> w;            // tempered words of this pose's normal attempts (4 per attempt), then its uniforms
    size_t attempts = 0;                // attempts scanned (the last one is accepted, unless no attempt is needed)
    size_t n_norm = 0, n_unif = 0;
    bool lead = false;                  // the first normal is the value cached by the previous draw
    double lead_gauss = 0.0;
    double* out_norm = nullptr;
    double* out_unif = nullptr;
    double loc = 0.0, scale = 1.0;
};

// The transform of one pose (worker thread): normals in stream order, then the uniforms.
void run_job(const PoseJob& j) {
    size_t o = 0;
    if (j.lead && j.n_norm) j.out_norm[o++] = j.loc + j.scale * j.lead_gauss;
    const uint32_t* w = j.w.data();
    for (size_t k = 0; k < j.attempts && o < j.n_norm; ++k) {
        const double x1 = 2.0 * to_double(w[4 * k], w[4 * k + 1]) - 1.0;
        const double x2 = 2.0 * to_double(w[4 * k + 2], w[4 * k + 3]) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        if (r2 >= 1.0 || r2 == 0.0) continue;
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        j.out_norm[o++] = j.loc + j.scale * (f * x2);
        if (o < j.n_norm) j.out_norm[o++] = j.loc + j.scale * (f * x1);
    }
    const uint32_t* u = w + 4 * j.attempts;
    for (size_t k = 0; k < j.n_unif; ++k) j.out_unif[k] = to_double(u[2 * k], u[2 * k + 1]);
}

struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<PoseJob>> q;
    bool done = false;
    void push(std::unique_ptr<PoseJob> j) {
        { std::lock_guard<std::mutex> l(m); q.push_back(std::move(j)); }
        cv.notify_one();
    }
    std::unique_ptr<PoseJob> pop() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return done || !q.empty(); });
        if (q.empty()) return nullptr;
        auto j = std::move(q.front());
        q.pop_front();
        return j;
    }
    void finish() {
        { std::lock_guard<std::mutex> l(m); done = true; }
        cv.notify_all();
    }
};

}  // namespace


// ---- the streaming path ---------------------------------------------------------------------------------------------------
namespace {

constexpr size_t kChunkWords = (size_t)1 << 18;          // 1 MiB of tempered words
constexpr size_t kChunkGroups = kChunkWords / 4;         // aligned groups of four words = candidate attempts
constexpr size_t kMaxAhead = 96;                         // chunks the generator may be ahead of the walk (memory bound)

struct Chunk {
    std::unique_ptr<uint32_t[]> buf;                     // kN words of history (raw), then the chunk's kChunkWords words:
                                                         // raw as generated, tempered in place by the flag task
    uint32_t* w = nullptr;                               // buf + kN
    std::unique_ptr<uint8_t[]> flag;                     // per group: it is an accepted attempt (if it is an attempt at all)
    uint32_t count = 0;                                  // accepted groups of the chunk
    uint32_t history[kN];                                // the kN raw words before the chunk (chunk 0: the caller's key)
    std::atomic<int> state{0};                           // 1 = raw words there, 2 = tempered, flags and count there
    std::atomic<int> refs{1};                            // poses still to be transformed that read it (+1: the walk)
};

// MT19937 as a stream: word j from the words 624, 623 and 227 places before it (the block form's mt[i], mt[i + 1],
// mt[i + 397] seen from one contiguous array).  No dependence shorter than 227: runs of 224 vectorise.
__attribute__((target_clones("avx2", "default")))
void mt_stream(uint32_t* __restrict__ b, size_t j0, size_t j1) {
    for (size_t j = j0; j < j1;) {
        const size_t m = std::min<size_t>(224, j1 - j);
#pragma GCC ivdep
        for (size_t q = j; q < j + m; ++q) {
            const uint32_t y = (b[q - kN] & kUpper) | (b[q - kN + 1] & kLower);
            b[q] = b[q - (kN - kM)] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
        }
        j += m;
    }
}

// tempering in place and the acceptance flags of a chunk in one pass (worker threads)
__attribute__((target_clones("avx2", "default")))
uint32_t temper_and_flag(uint32_t* __restrict__ w, size_t groups, uint8_t* __restrict__ flags) {
    uint32_t acc = 0;
    for (size_t k = 0; k < groups; ++k) {
        const uint32_t a = temper(w[4 * k]), b = temper(w[4 * k + 1]), c = temper(w[4 * k + 2]), d = temper(w[4 * k + 3]);
        w[4 * k] = a; w[4 * k + 1] = b; w[4 * k + 2] = c; w[4 * k + 3] = d;
        const double x1 = 2.0 * to_double(a, b) - 1.0;
        const double x2 = 2.0 * to_double(c, d) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        const uint32_t ok = (r2 >= 1.0 || r2 == 0.0) ? 0u : 1u;
        flags[k] = (uint8_t)ok;
        acc += ok;
    }
    return acc;
}

struct StreamJob {
    uint64_t pose = 0;
    uint64_t g0 = 0, attempts = 0;                       // first group, groups scanned (normals), then the uniforms' words
};

struct Streaming {
    // fixed for the call
    uint64_t P, nn, nu;
    double loc, scale;
    double* out_norm;
    double* out_unif;
    // chunks, created by the generator in order
    std::mutex cm;                                       // guards `chunks` (the vector, not the chunks)
    std::vector<std::unique_ptr<Chunk>> chunks;
    std::atomic<size_t> generated{0};                    // chunks whose words are there
    std::atomic<size_t> walk_chunk{0};                   // the chunk the walk is in (the generator stays < walk_chunk + kMaxAhead)
    std::atomic<bool> stop{false};
    std::atomic<bool> failed{false};                     // a thread ran out of memory: everybody winds down
    // tasks
    std::mutex qm;
    std::condition_variable qcv;                         // tasks, and progress the walk waits for
    std::deque<std::pair<int, uint64_t>> tasks;          // (0, chunk): flag it; (1, job index): transform it
    std::vector<StreamJob> jobs;
    bool closed = false;
    std::atomic<uint64_t> poses_done{0};

    Chunk* chunk(size_t c) {
        std::lock_guard<std::mutex> l(cm);
        return c < chunks.size() ? chunks[c].get() : nullptr;
    }
    // buffers of chunks nobody reads any more go back to the generator / the flag tasks (fresh memory costs page faults:
    // a third of the generator's time)
    std::mutex fm;
    std::vector<std::unique_ptr<uint32_t[]>> free_words;
    std::vector<std::unique_ptr<uint8_t[]>> free_flags;
    std::unique_ptr<uint32_t[]> take_words() {
        { std::lock_guard<std::mutex> l(fm); if (!free_words.empty()) { auto b = std::move(free_words.back()); free_words.pop_back(); return b; } }
        return std::unique_ptr<uint32_t[]>(new uint32_t[kN + kChunkWords]);
    }
    std::unique_ptr<uint8_t[]> take_flags() {
        { std::lock_guard<std::mutex> l(fm); if (!free_flags.empty()) { auto b = std::move(free_flags.back()); free_flags.pop_back(); return b; } }
        return std::unique_ptr<uint8_t[]>(new uint8_t[kChunkGroups]);
    }
    void release(size_t c) {
        Chunk* k = chunk(c);
        if (k && k->refs.fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> l(fm);
            if (k->buf) free_words.push_back(std::move(k->buf));
            if (k->flag) free_flags.push_back(std::move(k->flag));
            k->w = nullptr;
        }
    }
    void push(int kind, uint64_t v) {
        { std::lock_guard<std::mutex> l(qm); tasks.emplace_back(kind, v); }
        qcv.notify_all();
    }
    void run(std::pair<int, uint64_t> t) {
        if (t.first == 0) {
            Chunk* k = chunk((size_t)t.second);
            k->flag = take_flags();
            k->count = temper_and_flag(k->w, kChunkGroups, k->flag.get());
            k->state.store(2, std::memory_order_release);
            { std::lock_guard<std::mutex> l(qm); }
            qcv.notify_all();
        } else {
            transform(jobs[(size_t)t.second]);
            poses_done.fetch_add(1);
            { std::lock_guard<std::mutex> l(qm); }
            qcv.notify_all();
        }
    }
    // one task if there is one; false when the queue is empty
    bool help() {
        std::pair<int, uint64_t> t;
        {
            std::lock_guard<std::mutex> l(qm);
            if (tasks.empty()) return false;
            t = tasks.front();
            tasks.pop_front();
        }
        run(t);
        return true;
    }
    // the calling thread sleeps until there is a task to help with, `ready` holds, or something failed.  Every state change
    // it can wait for is followed by a lock / unlock of qm and a notify_all (run, push), so no wake-up is lost; the time-out
    // is a safety net only (and absent from the ThreadSanitizer build, whose runtime does not know pthread_cond_clockwait).
    template <class Ready>
    void wait_until(Ready ready) {
        std::unique_lock<std::mutex> l(qm);
        auto pred = [&] { return failed.load() || !tasks.empty() || ready(); };
#ifdef LRC_TSAN
        qcv.wait(l, pred);
#else
        qcv.wait_for(l, std::chrono::milliseconds(2), pred);
#endif
    }
    void worker() {
        for (;;) {
            std::pair<int, uint64_t> t;
            {
                std::unique_lock<std::mutex> l(qm);
                qcv.wait(l, [&] { return closed || !tasks.empty(); });
                if (tasks.empty()) return;
                t = tasks.front();
                tasks.pop_front();
            }
            try { run(t); } catch (...) { failed.store(true); stop.store(true); qcv.notify_all(); return; }
        }
    }
    // the pose's normals (accepted groups in stream order: f * x2, then f * x1), then its uniforms
    void transform(const StreamJob& j) {
        double* on = out_norm ? out_norm + j.pose * nn : nullptr;
        size_t o = 0;
        uint64_t g = j.g0;
        const uint64_t gend = j.g0 + j.attempts;
        while (g < gend) {
            const size_t c = (size_t)(g / kChunkGroups);
            Chunk* k = chunk(c);
            const size_t a = (size_t)(g % kChunkGroups), b = (size_t)std::min<uint64_t>(kChunkGroups, a + (gend - g));
            const uint32_t* w = k->w;
            const uint8_t* f = k->flag.get();
            for (size_t q = a; q < b; ++q) {
                if (!f[q]) continue;
                const double x1 = 2.0 * to_double(w[4 * q], w[4 * q + 1]) - 1.0;
                const double x2 = 2.0 * to_double(w[4 * q + 2], w[4 * q + 3]) - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                const double fac = std::sqrt(-2.0 * std::log(r2) / r2);
                on[o++] = loc + scale * (fac * x2);
                on[o++] = loc + scale * (fac * x1);
            }
            g += b - a;
        }
        double* ou = out_unif ? out_unif + j.pose * nu : nullptr;
        uint64_t wpos = 4 * gend;                         // word offset of the first uniform
        size_t left = (size_t)nu, u = 0;
        while (left) {
            const size_t c = (size_t)(wpos / kChunkWords);
            Chunk* k = chunk(c);
            const size_t a = (size_t)(wpos % kChunkWords);
            const size_t m = std::min(left, (kChunkWords - a) / 2);
            const uint32_t* w = k->w + a;
            for (size_t q = 0; q < m; ++q) ou[u + q] = to_double(w[2 * q], w[2 * q + 1]);
            u += m; left -= m; wpos += 2 * m;
        }
        // this pose no longer needs its chunks
        const size_t c0 = (size_t)(j.g0 / kChunkGroups);
        const uint64_t last_word = 4 * gend + 2 * nu;     // one past
        const size_t c1 = last_word ? (size_t)((last_word - 1) / kChunkWords) : c0;
        for (size_t c = c0; c <= c1; ++c) release(c);
    }
};

// raw words into chunk after chunk until told to stop: the only sequential part of the whole draw, and nothing but the
// recurrence (0.25 ns per word; tempering is the flag task's)
void generate(Streaming* S, const uint32_t* key0, int pos0) try {
    std::unique_ptr<Chunk> k = std::make_unique<Chunk>();
    k->buf = S->take_words();
    k->w = k->buf.get() + kN;
    // chunk 0: the caller's block lies so that key0[pos0] is the chunk's first word; what it still holds needs no generating
    std::memcpy(k->history, key0, sizeof(k->history));
    std::memcpy(k->buf.get() + (kN - pos0), key0, sizeof(uint32_t) * kN);
    size_t j0 = (size_t)(2 * kN - pos0);
    for (size_t c = 0; !S->stop.load(std::memory_order_acquire); ++c) {
        while (c >= S->walk_chunk.load(std::memory_order_acquire) + kMaxAhead && !S->stop.load(std::memory_order_acquire))
            std::this_thread::sleep_for(std::chrono::microseconds(50));      // far ahead of the walk: rare, and no hurry
        if (S->stop.load(std::memory_order_acquire)) break;
        mt_stream(k->buf.get(), j0, kN + kChunkWords);
        // the next chunk's history is this chunk's raw tail: taken before the flag task tempers it
        std::unique_ptr<Chunk> nx = std::make_unique<Chunk>();
        nx->buf = S->take_words();
        nx->w = nx->buf.get() + kN;
        std::memcpy(nx->buf.get(), k->buf.get() + kChunkWords, sizeof(uint32_t) * kN);
        std::memcpy(nx->history, nx->buf.get(), sizeof(nx->history));
        k->state.store(1, std::memory_order_release);
        { std::lock_guard<std::mutex> l(S->cm); S->chunks.push_back(std::move(k)); }
        S->generated.store(c + 1, std::memory_order_release);
        S->push(0, c);
        k = std::move(nx);
        j0 = kN;
    }
} catch (...) {
    S->failed.store(true);
    S->stop.store(true);
    S->qcv.notify_all();
}

int scan_streaming(lrc_mt19937_state* st, uint64_t num_poses, uint64_t normals_per_pose, uint64_t uniforms_per_pose,
                   double loc, double scale, double* out_normals, double* out_uniforms, int nthreads) {
    Streaming S;
    S.P = num_poses; S.nn = normals_per_pose; S.nu = uniforms_per_pose;
    S.loc = loc; S.scale = scale; S.out_norm = out_normals; S.out_unif = out_uniforms;
    S.jobs.resize((size_t)num_poses);
    // threads[0] = the generator, the rest = workers.  A start that fails leaves fewer workers (the walk helps with their
    // tasks anyway); without the generator there is no streaming path.  The guard winds everything down and joins on every
    // exit, exceptions included.
    std::vector<std::thread> threads;
    threads.reserve((size_t)std::max(nthreads, 2));
    auto guard = make_join_guard(threads, [&S] {
        S.stop.store(true, std::memory_order_release);
        { std::lock_guard<std::mutex> l(S.qm); S.closed = true; S.tasks.clear(); }
        S.qcv.notify_all();
    });
    {
        const uint32_t* key0 = (const uint32_t*)st->key;
        const int pos0 = st->pos;
        if (!try_start(threads, [&S, key0, pos0] { generate(&S, key0, pos0); })) throw NoGeneratorThread{};
    }
    for (int t = 2; t < nthreads; ++t)
        if (!try_start(threads, [&S] { S.worker(); })) break;
    // the walk: wait for chunk c's flags (helping with tasks meanwhile)
    auto flagged = [&](size_t c) -> Chunk* {
        for (;;) {
            if (S.failed.load()) throw std::bad_alloc();
            if (c < S.generated.load(std::memory_order_acquire)) {
                Chunk* k = S.chunk(c);
                if (k->state.load(std::memory_order_acquire) == 2) return k;
            }
            if (!S.help()) S.wait_until([&] { return c < S.generated.load(std::memory_order_acquire) && S.chunk(c)->state.load(std::memory_order_acquire) == 2; });
        }
    };
    uint64_t G = 0;                                      // group cursor
    bool bad = false;
    try {
    size_t held = 0;                                     // chunks below this have been released by the walk
    const uint64_t need_per_pose = normals_per_pose / 2, ugroups = uniforms_per_pose / 2;
    for (uint64_t p = 0; p < num_poses; ++p) {
        StreamJob& j = S.jobs[(size_t)p];
        j.pose = p; j.g0 = G;
        uint64_t need = need_per_pose, g = G;
        while (need) {
            const size_t c = (size_t)(g / kChunkGroups), a = (size_t)(g % kChunkGroups);
            S.walk_chunk.store(std::max(S.walk_chunk.load(std::memory_order_relaxed), c), std::memory_order_release);
            Chunk* k = flagged(c);
            if (a == 0 && k->count < need) { need -= k->count; g += kChunkGroups; continue; }
            const uint8_t* f = k->flag.get();
            size_t q = a;
            for (; q < kChunkGroups && need; ++q) need -= f[q];
            g += q - a;
        }
        j.attempts = g - G;
        G = g + ugroups;
        // the pose's chunks stay until its transform is done; the walk lets go of the chunks it has left behind
        const uint64_t last_word = 4 * g + 2 * uniforms_per_pose;
        const size_t c0 = (size_t)(j.g0 / kChunkGroups), c1 = last_word ? (size_t)((last_word - 1) / kChunkWords) : c0;
        for (size_t c = c0; c <= c1; ++c) {
            S.walk_chunk.store(std::max(S.walk_chunk.load(std::memory_order_relaxed), c), std::memory_order_release);
            Chunk* k = flagged(c);
            k->refs.fetch_add(1);
        }
        S.push(1, p);
        const size_t now = (size_t)(G / kChunkGroups);
        S.walk_chunk.store(std::max(S.walk_chunk.load(std::memory_order_relaxed), now), std::memory_order_release);
        for (; held < now; ++held) S.release(held);
    }
    S.stop.store(true, std::memory_order_release);
    while (S.poses_done.load() < num_poses) {
        if (S.failed.load()) throw std::bad_alloc();
        if (!S.help()) S.wait_until([&] { return S.poses_done.load() >= num_poses; });
    }
    } catch (...) { bad = true; }
    guard.wind_down();
    for (auto& t : threads) t.join();
    if (bad || S.failed.load()) throw std::bad_alloc();
    // the generator state numpy would be left with: key = numpy's block of the last word consumed, pos = one past it.  numpy's
    // block b holds the words [624 b - pos0, 624 (b + 1) - pos0) of this call's stream; block 0 is the caller's key.
    const uint64_t W = 4 * G;
    if (W) {
        const uint64_t pos0 = (uint64_t)st->pos, L = W - 1, blk = (L + pos0) / kN;
        if (blk) {
            const uint64_t bs = blk * kN - pos0;                   // stream offset of the block's first word
            const size_t c = (size_t)(bs / kChunkWords);
            Chunk* k = S.chunk(c);
            const size_t need = (size_t)(bs - (uint64_t)c * kChunkWords) + 2 * kN;      // words of scratch incl. the history
            std::vector<uint32_t> scratch(need + kN);
            size_t j0 = kN;
            if (c == 0) {
                std::memcpy(scratch.data() + (kN - pos0), k->history, sizeof(uint32_t) * kN);
                j0 = (size_t)(2 * kN - pos0);
            } else {
                std::memcpy(scratch.data(), k->history, sizeof(uint32_t) * kN);
            }
            if (need > j0) mt_stream(scratch.data(), j0, need);
            std::memcpy(st->key, scratch.data() + (need - kN), sizeof(uint32_t) * kN);
        }
        st->pos = (int)((L + pos0) % kN) + 1;
    }
    st->has_gauss = 0;
    st->gauss = 0.0;
    return LRC_OK;
}

}  // namespace

static int scan_sequential(lrc_mt19937_state* st, uint64_t num_poses, uint64_t normals_per_pose,
                           uint64_t uniforms_per_pose, double loc, double scale, double* out_normals,
                           double* out_uniforms, int threads) {
    try {
        Stream s;
        s.init(st->key, st->pos);
        bool has_gauss = st->has_gauss != 0;
        double gauss = st->gauss;
        unsigned hw = std::thread::hardware_concurrency();
        int nthreads = threads > 0 ? threads : (int)std::min<unsigned>(hw ? hw : 1u, 16u);
        if ((uint64_t)nthreads > num_poses) nthreads = (int)num_poses;
        Queue queue;
        std::vector<std::thread> pool;
        pool.reserve((size_t)nthreads);
        auto guard = make_join_guard(pool, [&queue] { queue.finish(); });      // joins on every exit, exceptions included
        for (int t = 1; t < nthreads; ++t)
            if (!try_start(pool, [&queue] { while (auto j = queue.pop()) run_job(*j); })) break;    // fewer threads, or none
        std::vector<uint32_t> flags;
        for (uint64_t p = 0; p < num_poses; ++p) {
            auto job = std::make_unique<PoseJob>();
            job->n_norm = normals_per_pose; job->n_unif = uniforms_per_pose;
            job->loc = loc; job->scale = scale;
            job->out_norm = out_normals ? out_normals + p * normals_per_pose : nullptr;
            job->out_unif = out_uniforms ? out_uniforms + p * uniforms_per_pose : nullptr;
            size_t left = normals_per_pose;
            if (left && has_gauss) {           // the cached second value of an earlier attempt comes first
                job->lead = true; job->lead_gauss = gauss;
                has_gauss = false; gauss = 0.0;
                --left;
            }
            const size_t need = (left + 1) / 2;        // accepted attempts this pose consumes
            size_t attempts = 0, found = 0;
            if (need) {
                // scan in growing batches until `need` accepted attempts are in the window; acceptance is pi / 4
                size_t batch = need + need / 3 + 64;
                while (found < need) {
                    s.ensure(4 * (attempts + batch));
                    flags.resize(batch);
                    const uint32_t* w = s.peek() + 4 * attempts;
                    const size_t got = count_block(w, batch, flags.data());
                    if (found + got < need) { found += got; attempts += batch; }
                    else {                     // the need-th acceptance is inside this batch
                        size_t k = 0;
                        for (; k < batch; ++k) { found += flags[k]; if (found == need) break; }
                        attempts += k + 1;
                    }
                    batch = std::max<size_t>(256, (need - std::min(found, need)) * 2 + 64);
                }
                if (left & 1u) {               // odd count: the last attempt's second value is cached for the next draw
                    const uint32_t* w = s.peek() + 4 * (attempts - 1);
                    const double x1 = 2.0 * to_double(w[0], w[1]) - 1.0, x2 = 2.0 * to_double(w[2], w[3]) - 1.0;
                    const double r2 = x1 * x1 + x2 * x2;
                    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
                    (void)x2;
                    has_gauss = true; gauss = f * x1;
                }
            }
            const size_t nwords = 4 * attempts + 2 * (size_t)uniforms_per_pose;
            s.ensure(nwords);
            job->attempts = attempts;
            job->w.assign(s.peek(), s.peek() + nwords);
            s.consume(nwords);
            if (pool.empty()) { run_job(*job); continue; }
            queue.push(std::move(job));
            for (;;) {                         // back-pressure: the producer works too while the queue is long
                std::unique_ptr<PoseJob> j;
                {
                    std::lock_guard<std::mutex> l(queue.m);
                    if (queue.q.size() <= (size_t)(2 * nthreads)) break;
                    j = std::move(queue.q.front());
                    queue.q.pop_front();
                }
                run_job(*j);
            }
        }
        // the producer helps to drain what is left
        if (!pool.empty()) {
            for (;;) {
                std::unique_ptr<PoseJob> j;
                {
                    std::lock_guard<std::mutex> l(queue.m);
                    if (queue.q.empty()) break;
                    j = std::move(queue.q.front());
                    queue.q.pop_front();
                }
                run_job(*j);
            }
            queue.finish();
            for (auto& t : pool) t.join();      // the transforms must be complete before the state is handed back
        }
        std::memcpy(st->key, s.key, sizeof(s.key));
        st->pos = s.pos;
        st->has_gauss = has_gauss ? 1 : 0;
        st->gauss = has_gauss ? gauss : 0.0;
    } catch (const std::bad_alloc&) {
        return lrc_internal_fail(LRC_ERR_OOM, "lrc_rng_scan_draws: out of host memory");
    } catch (...) {
        return lrc_internal_fail(LRC_ERR_INTERNAL, "lrc_rng_scan_draws: failed");
    }
    return LRC_OK;
}

// test hook (tests/test_nprandom.py): the next `budget` thread starts succeed, later ones fail like EAGAIN; < 0 = no limit
extern "C" void lrc_internal_set_thread_budget(long budget) { g_thread_budget.store(budget); }

extern "C" int lrc_rng_scan_draws(lrc_mt19937_state* st, uint64_t num_poses, uint64_t normals_per_pose,
                                  uint64_t uniforms_per_pose, double loc, double scale, double* out_normals,
                                  double* out_uniforms, int threads) {
    if (!st) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: state is NULL");
    if (st->pos < 0 || st->pos > kN) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: pos outside [0, 624]");
    if ((normals_per_pose && num_poses && !out_normals) || (uniforms_per_pose && num_poses && !out_uniforms))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: NULL output");
    if (num_poses == 0 || (normals_per_pose == 0 && uniforms_per_pose == 0)) return LRC_OK;
    const unsigned hw = std::thread::hardware_concurrency();
    int nthreads = threads > 0 ? threads : (int)std::min<unsigned>(hw ? hw : 1u, 16u);
    // threads < 0: the sequential path with |threads| threads (tests compare the two paths)
    const bool aligned = normals_per_pose % 2 == 0 && uniforms_per_pose % 2 == 0 && !st->has_gauss;
    // the streaming path pays when there are poses to hand out and normals to find the ends of: a trajectory of sensor-sized
    // poses; few huge poses, uniforms only or thousands of tiny poses are quicker on the sequential path
    const bool shaped = num_poses >= 4 && normals_per_pose >= 2 && normals_per_pose + uniforms_per_pose >= 4096 &&
                        num_poses * (normals_per_pose + uniforms_per_pose) >= (1u << 16);
    if (threads >= 0 && aligned && nthreads >= 3 && shaped) {
        try {
            return scan_streaming(st, num_poses, normals_per_pose, uniforms_per_pose, loc, scale, out_normals, out_uniforms, nthreads);
        } catch (const NoGeneratorThread&) {
            // no thread could be started at all: nothing was consumed yet, the sequential path does the same draws
        } catch (const std::bad_alloc&) {
            return lrc_internal_fail(LRC_ERR_OOM, "lrc_rng_scan_draws: out of host memory");
        } catch (...) {
            return lrc_internal_fail(LRC_ERR_INTERNAL, "lrc_rng_scan_draws: failed");
        }
    }
    return scan_sequential(st, num_poses, normals_per_pose, uniforms_per_pose, loc, scale, out_normals, out_uniforms,
                           threads < 0 ? -threads : threads);
}

// ---- the rays of a dual-axis pose from numpy's sines and cosines (include/lidarcast.h) --------------------------------------
__attribute__((target_clones("avx2", "default")))
static void rays_from_trig(const double* __restrict__ ct, const double* __restrict__ st, const double* __restrict__ cp,
                           const double* __restrict__ sp, uint64_t n, const double* __restrict__ M, float* __restrict__ out) {
    const float ox = (float)M[3], oy = (float)M[7], oz = (float)M[11];
    const double r00 = M[0], r01 = M[1], r02 = M[2], r10 = M[4], r11 = M[5], r12 = M[6], r20 = M[8], r21 = M[9], r22 = M[10];
    for (uint64_t i = 0; i < n; ++i) {
        const double d0 = ct[i] * cp[i], d1 = ct[i] * sp[i], d2 = st[i];
        float* o = out + 6 * i;
        o[0] = ox; o[1] = oy; o[2] = oz;
        o[3] = (float)((d0 * r00 + d1 * r01) + d2 * r02);
        o[4] = (float)((d0 * r10 + d1 * r11) + d2 * r12);
        o[5] = (float)((d0 * r20 + d1 * r21) + d2 * r22);
    }
}

extern "C" int lrc_rays_from_trig(const double* cos_theta, const double* sin_theta, const double* cos_phi, const double* sin_phi,
                                  uint64_t n, const double* pose16, float* out_rays6) {
    if (n && (!cos_theta || !sin_theta || !cos_phi || !sin_phi || !pose16 || !out_rays6))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rays_from_trig: NULL argument");
    if (n == 0) return LRC_OK;
    rays_from_trig(cos_theta, sin_theta, cos_phi, sin_phi, n, pose16, out_rays6);
    return LRC_OK;
}
