// hostpack.cpp - host side of the drop-in boundary: the reference hands over float64 NumPy piano-roll windows
// (one-hot rows, reference import_midi.py:245-286; packed per song by vae_definition.py:880-1045), the device engine
// consumes one byte per row in time-major order.  These functions do that conversion - validation, argmax, transpose and
// padding in ONE pass over the caller's array - on a small persistent thread pool, straight into the (pinned) staging block
// the engine uploads with a single asynchronous copy.  No device access here: every pointer is a HOST pointer.
//
// 64 MB of float64 one-hot rows per 256-window minibatch (T=512) become 128 KB: the conversion is a memory-bound read of
// the caller's array, so it is split over threads by windows; each thread writes its own columns of the time-major output.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <sched.h>
#include <cstdlib>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/midivae_hip.h"

namespace {

class Pool {
public:
    explicit Pool(int n) : pid_(getpid()) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { run(i); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    pid_t pid() const { return pid_; }
    // fn(part, parts) for part in [0, parts): the calling thread takes part 0
    void parallel(int parts, const std::function<void(int, int)>& fn) {
        if (parts <= 1 || workers_.empty()) {
            fn(0, 1);
            return;
        }
        if (parts > size() + 1) parts = size() + 1;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            parts_ = parts;
            next_ = 1;
            pending_ = parts - 1;
            ++gen_;
        }
        cv_.notify_all();
        fn(0, parts);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run(int) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn = nullptr;
            int part = -1, parts = 0;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || (gen_ != seen && next_ < parts_); });
                if (stop_) return;
                part = next_++;
                parts = parts_;
                fn = fn_;
                if (next_ >= parts_) seen = gen_;
            }
            (*fn)(part, parts);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    int parts_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    pid_t pid_;
};

std::mutex g_pool_mutex;
// One packer call at a time: ctypes releases the GIL around these calls, so two Python threads (two models, two Stagers) can arrive
// together - Pool::parallel holds ONE job (fn_, parts_, pending_), and mvae_host_threads may replace the pool.  Every entry point
// takes this lock for its whole duration; a second caller waits (the conversion is memory-bound: it would not run faster beside the first).
std::mutex g_call_mutex;
Pool* g_pool = nullptr;
int g_threads = 0;          // 0 = default

int default_threads() {
    // The hardware threads THIS process may use: its affinity mask (a launcher that pins ranks to cores is honoured), shared between
    // the processes of the node - one per GPU under data parallelism (LOCAL_WORLD_SIZE, set by torch.distributed.run): eight ranks
    // each sizing their pool for the whole host were 512 threads on a 256-thread box.
    int hw = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = CPU_COUNT(&set);
    if (hw <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        hw = hc ? (int)hc : 4;
    }
    int local = 1;
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int v = std::atoi(e);
        if (v > 1) local = v;
    }
    const int share = hw / local > 1 ? hw / local : 1;
    // Memory-bound, and called once per train step: FEW threads.  Round 5 on the 256-thread two-socket host of an MI355X box: warm, 8
    // threads convert a 256 x 512 x 61 float64 minibatch (64 MB) in 0.71 ms = 90 GB/s, 64 threads in 0.5-0.8 ms - but a pool that has
    // slept through a 6 ms train step wakes slowly and unevenly: inside fit the same conversion took 0.8-3.7 ms with 64 threads
    // (5.85-9.09 ms per step over six processes), 0.5-1.1 ms with 8 (5.66-5.85 ms, all six).  At most 8, within the share.
    const int n = share < 8 ? share : 8;
    return n < 1 ? 1 : n;
}

Pool& pool() {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    const int want = (g_threads > 0 ? g_threads : default_threads()) - 1;
    if (g_pool && g_pool->pid() != getpid()) g_pool = nullptr;        // forked child: the parent's threads do not exist here
    if (!g_pool || g_pool->size() != want) {
        if (g_pool && g_pool->pid() == getpid()) delete g_pool;
        g_pool = new Pool(want < 0 ? 0 : want);
    }
    return *g_pool;
}

template <typename T>
inline int64_t onehot_rows(const T* x, int64_t n_rows, int K, uint8_t* out, int64_t out_stride, int64_t ldx = 0) {
    // returns -1, or the first row (relative) that is not one-hot: entries must be exactly 0 or 1, exactly one 1
    // (ldx: elements between rows when the K columns are a block of wider rows)
    if (ldx == 0) ldx = K;
    for (int64_t r = 0; r < n_rows; ++r) {
        const T* row = x + r * ldx;
        int idx = 0, ones = 0, bad = 0;            // branch-free (vectorises): with exactly one non-zero, idx = its position
        for (int k = 0; k < K; ++k) {
            const T v = row[k];
            const int nz = v != (T)0;
            bad |= nz & (v != (T)1);
            ones += nz;
            idx += nz ? k : 0;
        }
        if (bad || ones != 1) return r;
        out[r * out_stride] = (uint8_t)idx;
    }
    return -1;
}

template <typename T>
int onehot_to_index_tm(const T* x, int64_t n, int T_, int K, int64_t lo, int64_t hi, uint8_t* out, int Bp, uint8_t fill,
                       int64_t* bad_row) {
    const int64_t B = hi - lo;
    std::atomic<int64_t> bad(-1);
    const int64_t bytes = B * (int64_t)T_ * K * (int64_t)sizeof(T);
    int parts = (int)(bytes / (1 << 20)) + 1;                         // ~1 MiB of input per part at least
    Pool& p = pool();
    if (parts > p.size() + 1) parts = p.size() + 1;
    if (parts > B) parts = B > 0 ? (int)B : 1;
    p.parallel(parts, [&](int part, int nparts) {
        const int64_t b0 = B * part / nparts, b1 = B * (part + 1) / nparts;
        for (int64_t b = b0; b < b1; ++b) {
            // window b: rows (lo+b)*T .. +T of x -> column b of the (T, Bp) output
            const int64_t r = onehot_rows<T>(x + (lo + b) * (int64_t)T_ * K, T_, K, out + b, Bp);
            if (r >= 0) {       // the LOWEST offending row over all parts (each part meets its own lowest first)
                const int64_t mine = (lo + b) * (int64_t)T_ + r;
                int64_t cur = bad.load();
                while ((cur < 0 || mine < cur) && !bad.compare_exchange_weak(cur, mine)) {}
                return;
            }
        }
    });
    for (int t = 0; t < T_; ++t)
        for (int64_t b = B; b < Bp; ++b) out[(int64_t)t * Bp + b] = fill;
    if (bad.load() >= 0) {
        if (bad_row) *bad_row = bad.load();
        return MVAE_E_FORMAT;
    }
    return MVAE_OK;
}

template <typename T>
int twohot_to_index_tm(const T* x, int64_t n, int T_, int K, int K1, int64_t lo, int64_t hi, uint8_t* out1, uint8_t* out2, int Bp,
                       uint8_t fill, int64_t* bad_row) {
    const int64_t B = hi - lo;
    std::atomic<int64_t> bad(-1);
    const int64_t bytes = B * (int64_t)T_ * K * (int64_t)sizeof(T);
    int parts = (int)(bytes / (1 << 20)) + 1;
    Pool& p = pool();
    if (parts > p.size() + 1) parts = p.size() + 1;
    if (parts > B) parts = B > 0 ? (int)B : 1;
    p.parallel(parts, [&](int part, int nparts) {
        const int64_t b0 = B * part / nparts, b1 = B * (part + 1) / nparts;
        for (int64_t b = b0; b < b1; ++b) {
            const T* win = x + (lo + b) * (int64_t)T_ * K;
            int64_t r = onehot_rows<T>(win, T_, K1, out1 + b, Bp, K);
            const int64_t r2 = onehot_rows<T>(win + K1, T_, K - K1, out2 + b, Bp, K);
            if (r2 >= 0 && (r < 0 || r2 < r)) r = r2;
            if (r >= 0) {
                const int64_t mine = (lo + b) * (int64_t)T_ + r;
                int64_t cur = bad.load();
                while ((cur < 0 || mine < cur) && !bad.compare_exchange_weak(cur, mine)) {}
                return;
            }
        }
    });
    for (int t = 0; t < T_; ++t)
        for (int64_t b = B; b < Bp; ++b) out1[(int64_t)t * Bp + b] = out2[(int64_t)t * Bp + b] = fill;
    if (bad.load() >= 0) {
        if (bad_row) *bad_row = bad.load();
        return MVAE_E_FORMAT;
    }
    return MVAE_OK;
}

template <typename T>
void rows_to_tm(const T* v, int T_, int64_t lo, int64_t hi, float scale, float* out, int Bp) {
    const int64_t B = hi - lo;
    int parts = (int)(B * (int64_t)T_ / (1 << 17)) + 1;
    Pool& p = pool();
    p.parallel(parts, [&](int part, int nparts) {
        const int64_t b0 = B * part / nparts, b1 = B * (part + 1) / nparts;
        for (int64_t b = b0; b < b1; ++b) {
            const T* src = v + (lo + b) * (int64_t)T_;
            for (int t = 0; t < T_; ++t) out[(int64_t)t * Bp + b] = scale * (float)src[t];
        }
    });
    for (int t = 0; t < T_; ++t)
        for (int64_t b = B; b < Bp; ++b) out[(int64_t)t * Bp + b] = 0.0f;
}

}  // namespace

extern "C" int mvae_host_threads(int32_t n) {
    std::lock_guard<std::mutex> call(g_call_mutex);
    {
        std::lock_guard<std::mutex> g(g_pool_mutex);
        if (n >= 0) g_threads = n;
    }
    return pool().size() + 1;
}

extern "C" int mvae_host_onehot_to_index_tm(const void* x, int32_t xkind, int64_t n, int32_t T, int32_t K, int64_t lo, int64_t hi,
                                            uint8_t* out, int32_t Bp, uint8_t fill, int64_t* bad_row) {
    if (!x || !out || n < 0 || T <= 0 || K <= 0 || K > 255 || lo < 0 || hi < lo || hi > n || Bp < hi - lo) return MVAE_E_ARG;
    std::lock_guard<std::mutex> call(g_call_mutex);
    switch (xkind) {
        case MVAE_HOST_F64: return onehot_to_index_tm<double>(static_cast<const double*>(x), n, T, K, lo, hi, out, Bp, fill, bad_row);
        case MVAE_HOST_F32: return onehot_to_index_tm<float>(static_cast<const float*>(x), n, T, K, lo, hi, out, Bp, fill, bad_row);
        case MVAE_HOST_U8: return onehot_to_index_tm<uint8_t>(static_cast<const uint8_t*>(x), n, T, K, lo, hi, out, Bp, fill, bad_row);
    }
    return MVAE_E_ARG;
}

extern "C" int mvae_host_twohot_to_index_tm(const void* x, int32_t xkind, int64_t n, int32_t T, int32_t K, int32_t K1, int64_t lo,
                                            int64_t hi, uint8_t* out1, uint8_t* out2, int32_t Bp, uint8_t fill, int64_t* bad_row) {
    if (!x || !out1 || !out2 || n < 0 || T <= 0 || K1 <= 0 || K <= K1 || K1 > 255 || K - K1 > 255 || lo < 0 || hi < lo || hi > n ||
        Bp < hi - lo)
        return MVAE_E_ARG;
    std::lock_guard<std::mutex> call(g_call_mutex);
    switch (xkind) {
        case MVAE_HOST_F64: return twohot_to_index_tm<double>(static_cast<const double*>(x), n, T, K, K1, lo, hi, out1, out2, Bp, fill, bad_row);
        case MVAE_HOST_F32: return twohot_to_index_tm<float>(static_cast<const float*>(x), n, T, K, K1, lo, hi, out1, out2, Bp, fill, bad_row);
        case MVAE_HOST_U8: return twohot_to_index_tm<uint8_t>(static_cast<const uint8_t*>(x), n, T, K, K1, lo, hi, out1, out2, Bp, fill, bad_row);
    }
    return MVAE_E_ARG;
}

extern "C" int mvae_host_index_to_tm(const uint8_t* idx, int64_t n, int32_t T, int64_t lo, int64_t hi, uint8_t* out, int32_t Bp,
                                     uint8_t fill) {
    if (!idx || !out || n < 0 || T <= 0 || lo < 0 || hi < lo || hi > n || Bp < hi - lo) return MVAE_E_ARG;
    const int64_t B = hi - lo;
    for (int64_t b = 0; b < B; ++b) {
        const uint8_t* src = idx + (lo + b) * (int64_t)T;
        for (int t = 0; t < T; ++t) out[(int64_t)t * Bp + b] = src[t];
    }
    for (int t = 0; t < T; ++t)
        for (int64_t b = B; b < Bp; ++b) out[(int64_t)t * Bp + b] = fill;
    return MVAE_OK;
}

extern "C" int mvae_host_rows_to_tm_f32(const void* v, int32_t vkind, int64_t n, int32_t T, int64_t lo, int64_t hi, float scale,
                                        float* out, int32_t Bp) {
    if (!v || !out || n < 0 || T <= 0 || lo < 0 || hi < lo || hi > n || Bp < hi - lo) return MVAE_E_ARG;
    std::lock_guard<std::mutex> call(g_call_mutex);
    switch (vkind) {
        case MVAE_HOST_F64: rows_to_tm<double>(static_cast<const double*>(v), T, lo, hi, scale, out, Bp); return MVAE_OK;
        case MVAE_HOST_F32: rows_to_tm<float>(static_cast<const float*>(v), T, lo, hi, scale, out, Bp); return MVAE_OK;
        case MVAE_HOST_U8: rows_to_tm<uint8_t>(static_cast<const uint8_t*>(v), T, lo, hi, scale, out, Bp); return MVAE_OK;
    }
    return MVAE_E_ARG;
}
