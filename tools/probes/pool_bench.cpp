// microbenchmark: fork-join latency of the sampler's chain pool (A: one epoch line + sequential gather, as in
// tamcmc_sampler.cpp) against B: a mailbox line per worker + a sweeping gather
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdint>

class PoolA {
public:
    explicit PoolA(int nthreads) : P_(nthreads), slots_((size_t)nthreads) { for (int t = 1; t < P_; t++) workers_.emplace_back([this, t] { loop(t); }); }
    ~PoolA() { { std::lock_guard<std::mutex> g(mx_); stop_ = true; hdr_.epoch.fetch_add(1, std::memory_order_release); } cv_.notify_all(); for (auto &t : workers_) t.join(); }
    template <class F> void run(int n, F &&fn)
    {
        std::function<void(int)> job = std::ref(fn);
        hdr_.job.store(&job, std::memory_order_relaxed); hdr_.n.store(n, std::memory_order_relaxed);
        uint64_t e;
        { std::lock_guard<std::mutex> g(mx_); e = hdr_.epoch.fetch_add(1, std::memory_order_release) + 1; }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        for (int i = 0, hi = block_end(0, n); i < hi; i++) fn(i);
        for (int p = 1; p < P_; p++) while (slots_[(size_t)p].done.load(std::memory_order_acquire) != e) __builtin_ia32_pause();
    }
private:
    struct alignas(64) Header { std::atomic<uint64_t> epoch{0}; std::atomic<int> n{0}; std::atomic<const std::function<void(int)> *> job{nullptr}; };
    struct alignas(64) Slot { std::atomic<uint64_t> done{0}; };
    int block_end(int p, int n) const { return (int)(((long long)(p + 1) * n) / P_); }
    void loop(int p)
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t e = hdr_.epoch.load(std::memory_order_acquire);
            for (int spin = 0; e == seen && spin < 40000; spin++) { __builtin_ia32_pause(); e = hdr_.epoch.load(std::memory_order_acquire); }
            if (e == seen) {
                std::unique_lock<std::mutex> g(mx_);
                sleepers_.fetch_add(1, std::memory_order_release);
                cv_.wait(g, [&] { return hdr_.epoch.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_release);
                e = hdr_.epoch.load(std::memory_order_acquire);
            }
            seen = e;
            if (stop_) return;
            const int n = hdr_.n.load(std::memory_order_relaxed);
            const std::function<void(int)> &job = *hdr_.job.load(std::memory_order_relaxed);
            for (int i = block_end(p - 1, n), hi = block_end(p, n); i < hi; i++) job(i);
            slots_[(size_t)p].done.store(e, std::memory_order_release);
        }
    }
    int P_; Header hdr_; std::vector<Slot> slots_; std::vector<std::thread> workers_; std::mutex mx_; std::condition_variable cv_; std::atomic<int> sleepers_{0}; bool stop_ = false;
};

class PoolB {
public:
    explicit PoolB(int nthreads) : P_(nthreads), box_((size_t)nthreads) { for (int t = 1; t < P_; t++) workers_.emplace_back([this, t] { loop(t); }); }
    ~PoolB() { stop_.store(true); { std::lock_guard<std::mutex> g(mx_); epoch_++; for (int p = 1; p < P_; p++) box_[(size_t)p].go.store(epoch_, std::memory_order_release); } cv_.notify_all(); for (auto &t : workers_) t.join(); }
    template <class F> void run(int n, F &&fn)
    {
        std::function<void(int)> job = std::ref(fn);
        hdr_.job.store(&job, std::memory_order_relaxed); hdr_.n.store(n, std::memory_order_relaxed);
        const uint64_t e = ++epoch_;
        for (int p = 1; p < P_; p++) box_[(size_t)p].go.store(e, std::memory_order_release);      // one line per worker
        if (sleepers_.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> g(mx_); cv_.notify_all(); }
        for (int i = 0, hi = block_end(0, n); i < hi; i++) fn(i);
        for (;;) {                         // sweep: the loads of one pass are independent of each other
            bool all = true;
            for (int p = 1; p < P_; p++) all &= (box_[(size_t)p].done.load(std::memory_order_acquire) == e);
            if (all) break;
            __builtin_ia32_pause();
        }
    }
private:
    struct alignas(64) Header { std::atomic<int> n{0}; std::atomic<const std::function<void(int)> *> job{nullptr}; };
    struct alignas(64) Go { std::atomic<uint64_t> go{0}; };
    struct alignas(128) Box { std::atomic<uint64_t> go{0}; char pad[56]; std::atomic<uint64_t> done{0}; char pad2[56]; };
    int block_end(int p, int n) const { return (int)(((long long)(p + 1) * n) / P_); }
    void loop(int p)
    {
        uint64_t seen = 0;
        Box &b = box_[(size_t)p];
        for (;;) {
            uint64_t e = b.go.load(std::memory_order_acquire);
            for (int spin = 0; e == seen && spin < 40000; spin++) { __builtin_ia32_pause(); e = b.go.load(std::memory_order_acquire); }
            if (e == seen) {
                std::unique_lock<std::mutex> g(mx_);
                sleepers_.fetch_add(1, std::memory_order_seq_cst);
                cv_.wait(g, [&] { return b.go.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_seq_cst);
                e = b.go.load(std::memory_order_acquire);
            }
            seen = e;
            if (stop_.load()) return;
            const int n = hdr_.n.load(std::memory_order_relaxed);
            const std::function<void(int)> &job = *hdr_.job.load(std::memory_order_relaxed);
            for (int i = block_end(p - 1, n), hi = block_end(p, n); i < hi; i++) job(i);
            b.done.store(e, std::memory_order_release);
        }
    }
    int P_; Header hdr_; std::vector<Box> box_; std::vector<std::thread> workers_; std::mutex mx_; std::condition_variable cv_; std::atomic<int> sleepers_{0}; std::atomic<bool> stop_{false}; uint64_t epoch_ = 0;
};

template <class Pool> double bench(int nt, int reps, int gap_us)
{
    Pool pool(nt);
    std::vector<double> sink(64 * 8, 0.0);
    auto job = [&](int m) { sink[(size_t)m * 8] += 1.0; };
    for (int i = 0; i < 2000; i++) pool.run(64, job);
    double total = 0;
    for (int i = 0; i < reps; i++) {
        if (gap_us) { auto t = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count() < gap_us) __builtin_ia32_pause(); }
        auto t0 = std::chrono::steady_clock::now();
        pool.run(64, job);
        total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return total / reps;
}

int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 16;
    for (int gap : {0, 10, 40})
        for (int r = 0; r < 2; r++)
            printf("threads %d, %2d us between forks: A (epoch line, sequential gather) %.2f us   B (mailboxes, sweep) %.2f us\n", nt, gap,
                   bench<PoolA>(nt, 20000, gap), bench<PoolB>(nt, 20000, gap));
    return 0;
}
