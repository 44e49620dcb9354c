// Host model of the instance FIFO of k_cl_loop (robust-nonlinear-mpc_amd/csrc/slsqp_api.hip: clq_pop / clq_push), test infrastructure only: the same
// protocol on std::atomic with one thread per "wave", so that ThreadSanitizer and plain repetition on the CPU can look for lost, duplicated or stranded
// instances and for waves that never exit.  Protocol: slots[cap] hold 0 (empty) or instance + 1; `tail` / `head` hand out push / pop tickets; `avail`
// counts published items that no popper has claimed yet.  pop: claim (avail--; give the claim back and report "empty" if none was left), take a head
// ticket, wait for that slot to be published, take the item and clear the slot.  push: take a tail ticket, wait for the slot to be clear, publish, avail++.
// A worker that finds the queue empty exits: an item is always either in the queue or held by a running worker, who will push it and pop again.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

struct Queue {
    std::vector<std::atomic<int>> slots;
    unsigned mask;
    std::atomic<unsigned> head{0}, tail{0};
    std::atomic<int> avail{0}, err{0};
    explicit Queue(unsigned cap) : slots(cap), mask(cap - 1) { for (auto &s : slots) s.store(0); }
    int pop() {
        const int av = avail.fetch_sub(1);
        if (av <= 0) { avail.fetch_add(1); return -1; }
        const unsigned t = head.fetch_add(1);
        std::atomic<int> &slot = slots[t & mask];
        for (long spin = 0; spin < (1L << 28); spin++) {
            const int got = slot.exchange(0, std::memory_order_acquire);
            if (got) return got - 1;
            std::this_thread::yield();
        }
        err.store(1);
        return -1;
    }
    void push(int b) {
        const unsigned p = tail.fetch_add(1);
        std::atomic<int> &slot = slots[p & mask];
        for (long spin = 0; spin < (1L << 28); spin++) {
            int expected = 0;
            if (slot.compare_exchange_strong(expected, b + 1, std::memory_order_release, std::memory_order_relaxed)) { avail.fetch_add(1); return; }
            std::this_thread::yield();
        }
        err.store(2);
    }
};

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 97, steps = argc > 2 ? atoi(argv[2]) : 50, workers = argc > 3 ? atoi(argv[3]) : 13, keep = argc > 4 ? atoi(argv[4]) : 1;
    unsigned cap = 1;
    while (cap < 4u * (unsigned)B) cap <<= 1;
    Queue q(cap);
    for (int b = 0; b < B; b++) q.slots[b].store(b + 1);
    q.tail.store(B); q.avail.store(B);
    std::vector<int> stepno(B, 0);                  // written only by the worker that holds the instance; handed over through the queue's release / acquire
    std::vector<long long> state(B, 0);             // "instance data": must see every previous step's write
    std::atomic<long long> done_steps{0};
    std::atomic<int> bad{0};
    auto worker = [&](int w) {
        int b = -1;
        for (;;) {
            if (b < 0) { b = q.pop(); if (b < 0) break; }
            const int s = stepno[b];
            if (state[b] != (long long)s * (s + 1) / 2) bad.store(1);      // the sum of the steps run so far: a lost hand-over shows here
            state[b] += s + 1;
            for (volatile int spin = 0; spin < 50 * ((b * 7 + s * 3 + w) % 11); spin++) {}      // uneven step times
            stepno[b] = s + 1;
            const long long d = done_steps.fetch_add(1) + 1;
            if (s + 1 >= steps) { b = -1; continue; }
            if (keep && (long long)(s + 1) * B < d) continue;      // an instance behind the mean keeps its worker (k_cl_loop's rule)
            q.push(b);
            b = -1;
        }
    };
    std::vector<std::thread> th;
    for (int w = 0; w < workers; w++) th.emplace_back(worker, w);
    for (auto &t : th) t.join();
    int unfinished = 0;
    for (int b = 0; b < B; b++) unfinished += (stepno[b] != steps) || (state[b] != (long long)steps * (steps + 1) / 2);
    if (q.err.load() || bad.load() || unfinished || done_steps.load() != (long long)B * steps || q.avail.load() != 0) {
        printf("FAIL err %d bad %d unfinished %d done %lld avail %d\n", q.err.load(), bad.load(), unfinished, done_steps.load(), q.avail.load());
        return 1;
    }
    printf("ok B %d steps %d workers %d keep %d\n", B, steps, workers, keep);
    return 0;
}
