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
// Only the word stream is sequential.  An attempt of the polar method consumes exactly two uniforms, so where a pose's
// normals end is known after a cheap, vectorisable pass over r2 alone (the producer thread); the expensive part -- log,
// sqrt, division per accepted attempt, with THIS process's libm, as numpy calls it -- is done pose by pose on worker
// threads while the producer is already generating the next pose's words.
// Pinned by tests/test_nprandom.py (-m "not gpu") against numpy itself, and through the frames by the G3 goldens.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int code, const char* msg);

namespace {

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

extern "C" int lrc_rng_scan_draws(lrc_mt19937_state* st, uint64_t num_poses, uint64_t normals_per_pose,
                                  uint64_t uniforms_per_pose, double loc, double scale, double* out_normals,
                                  double* out_uniforms, int threads) {
    if (!st) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: state is NULL");
    if (st->pos < 0 || st->pos > kN) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: pos outside [0, 624]");
    if ((normals_per_pose && num_poses && !out_normals) || (uniforms_per_pose && num_poses && !out_uniforms))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rng_scan_draws: NULL output");
    if (num_poses == 0 || (normals_per_pose == 0 && uniforms_per_pose == 0)) return LRC_OK;
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
        for (int t = 1; t < nthreads; ++t)
            pool.emplace_back([&queue] { while (auto j = queue.pop()) run_job(*j); });
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
            for (auto& t : pool) t.join();
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
