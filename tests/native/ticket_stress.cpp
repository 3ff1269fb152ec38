// Host-thread stress of the block_ticket_arrive protocol of csrc/device_intrinsics.h (TEST INFRASTRUCTURE):
// the same word protocol on std::atomic, T threads standing in for the blocks of a launch, the words starting
// from zeros, from garbage and from the leftovers of a "dead" launch.  Checks: every round terminates, exactly one
// arrival per round is told it is the last one, and it is told so only after every other arrival has arrived.
//   g++ -O2 -std=c++17 -pthread ticket_stress.cpp -o ticket_stress && ./ticket_stress
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

static const int kGroups = 16, kStride = 16, kSingleMax = 32;

static unsigned ticket_count(std::atomic<uint64_t>* word, unsigned nonce) {
    const uint64_t old = word->fetch_add(1, std::memory_order_relaxed);
    if ((unsigned)(old >> 32) == nonce) return (unsigned)old + 1u;
    uint64_t cur = word->load(std::memory_order_relaxed);
    for (long spin = 0;; ++spin) {
        if ((unsigned)(cur >> 32) == nonce) return (unsigned)word->fetch_add(1, std::memory_order_relaxed) + 1u;
        if (word->compare_exchange_strong(cur, ((uint64_t)nonce << 32) | 1ull, std::memory_order_relaxed)) return 1u;
        if (spin > (1L << 24)) { fprintf(stderr, "ticket_count: no progress\n"); abort(); }
    }
}

// device_intrinsics.h::ticket_prepare_word: the early self-preparation by one block, racing with the arrivals
static void prepare_word(std::atomic<uint64_t>* word, unsigned nonce) {
    uint64_t cur = word->load(std::memory_order_relaxed);
    for (long spin = 0; (unsigned)(cur >> 32) != nonce; ++spin) {
        if (word->compare_exchange_strong(cur, (uint64_t)nonce << 32, std::memory_order_relaxed)) break;
        if (spin > (1L << 24)) { fprintf(stderr, "prepare_word: no progress\n"); abort(); }
    }
}

static bool arrive(std::atomic<uint64_t>* words, unsigned nonce, unsigned total, unsigned id) {
    if (total <= (unsigned)kSingleMax) {
        const bool last = ticket_count(words, nonce) == total;
        if (last) words->store(0, std::memory_order_relaxed);
        return last;
    }
    const unsigned g = id % kGroups, members = (total - g + kGroups - 1) / kGroups;
    std::atomic<uint64_t>* gw = words + (1 + g) * kStride;
    if (ticket_count(gw, nonce) != members) return false;
    gw->store(0, std::memory_order_relaxed);
    const bool last = ticket_count(words, nonce) == (unsigned)kGroups;
    if (last) words->store(0, std::memory_order_relaxed);
    return last;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937_64 rng(12345);
    std::vector<std::atomic<uint64_t>> words((1 + kGroups) * kStride);
    unsigned nonce = 1;
    for (int r = 0; r < rounds; ++r) {
        const unsigned total = (r % 3 == 0) ? 1 + rng() % 32 : 33 + rng() % 96;      // single- and two-level rounds
        const int mode = r % 4;                                                      // 0/1: clean (left by the last round), 2: garbage, 3: dead launch
        if (mode == 2) for (auto& w : words) w.store(rng(), std::memory_order_relaxed);
        if (mode == 3) for (auto& w : words) w.store(((uint64_t)(nonce - 1) << 32) | (rng() % 7), std::memory_order_relaxed);
        ++nonce;
        std::atomic<unsigned> arrived{0}, lasts{0}, arrived_when_last{0};
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        // in half of the rounds "block 0" prepares the words before its own arrival (program order on the device: the
        // preparation is the first thing block 0 does), at a random moment relative to the OTHER blocks' arrivals
        const bool with_prepare = (r / 4) % 2 == 1;
        const unsigned delay = rng() % 4000;
        for (unsigned t = 0; t < total; ++t)
            th.emplace_back([&, t] {
                while (!go.load(std::memory_order_acquire)) {}
                if (with_prepare && t == 0) {
                    for (volatile unsigned d = 0; d < delay; ++d) {}
                    const int n = total <= (unsigned)kSingleMax ? 1 : 1 + kGroups;
                    for (int w = 0; w < n; ++w) prepare_word(words.data() + w * kStride, nonce);
                }
                arrived.fetch_add(1, std::memory_order_seq_cst);
                if (arrive(words.data(), nonce, total, t)) {
                    lasts.fetch_add(1);
                    arrived_when_last.store(arrived.load(std::memory_order_seq_cst));
                }
            });
        go.store(1, std::memory_order_release);
        for (auto& x : th) x.join();
        if (lasts.load() != 1 || arrived_when_last.load() != total) {
            printf("FAIL round %d total %u mode %d: lasts %u, arrived at that moment %u\n", r, total, mode, lasts.load(), arrived_when_last.load());
            return 1;
        }
        // every word the round used is left clean (the other words may still hold the garbage of an earlier round)
        // (a group word may hold (nonce, 0 arrivals) when block 0's preparation reached it after the group had finished:
        // an empty state of this launch, foreign -- hence empty -- to every later one)
        const uint64_t empty_of_this_launch = (uint64_t)nonce << 32;
        if (words[0].load() != 0) { printf("FAIL round %d: top word left dirty\n", r); return 1; }
        if (total > (unsigned)kSingleMax)
            for (int g = 0; g < kGroups; ++g) {
                const uint64_t w = words[(1 + g) * kStride].load();
                if (w != 0 && !(with_prepare && w == empty_of_this_launch)) { printf("FAIL round %d: group word %d left dirty\n", r, g); return 1; }
            }
    }
    printf("OK %d rounds\n", rounds);
    return 0;
}
