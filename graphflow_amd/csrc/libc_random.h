// libc_random.h -- glibc's rand() stream stepped inline (host code; used by smp_model.hip for the slice masks of
// RisiContraction_18_dropout, GraphFlow/RisiContraction_18_dropout.h:113-125, and by tests/cpp/test_libc_random.cpp).
#ifndef GF_LIBC_RANDOM_H_INCLUDED
#define GF_LIBC_RANDOM_H_INCLUDED
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace gf {
// The slice masks of RisiContraction_18_dropout are drawn with rand() -- a million calls per 1024-sample step of SMP_sigma_pairgraphs,
// 6 - 12 ms of locked libc calls, more than the device step they feed.  glibc's rand() is random()'s TYPE_3 additive-feedback
// generator (31 words, r[i] += r[i - 3], result >> 1); setstate() hands out the live state array, with the rear index encoded in the
// word in front of it (glibc random_r.c: __setstate_r).  LibcRandom borrows that array for a run of draws, steps the recurrence inline,
// writes the indices back and reinstalls the array: the process's rand() stream continues exactly where the reference's would.
// A one-time self-check compares eight borrowed draws with rand() itself (and rewinds); on any doubt every draw is a plain rand().
// Threads: libc's state is process-global, so the borrow is too -- ONE process-wide mutex is held from borrow() to give_back()
// (a second model's forward on another thread waits for the first one's mask loop, exactly as its rand() calls would queue on libc's
// own lock), and the array libc is parked on meanwhile is a static buffer, never a stack frame that could be gone when a
// setstate() names it (SMP_sigma_pairgraphs::Threaded_ComputeGradient runs one model per worker thread).
class LibcRandom {
    static constexpr int kDeg = 31, kSep = 3, kType = 3, kMaxTypes = 5;
    static std::mutex &borrow_lock() {
        static std::mutex m;
        return m;
    }
    static int32_t *parking() {   // where libc's generator lives while its real array is borrowed (guarded by borrow_lock)
        static int32_t buf[34] = {0};
        return buf;
    }
    int32_t *live_ = nullptr;   // the borrowed array's type / rear word; the state words follow
    int f_ = 0, r_ = 0;
    bool borrowed_ = false;
    bool borrow() {
        borrow_lock().lock();
        char *prev = initstate(1u, reinterpret_cast<char *>(parking()), 128);   // libc now runs on the parking array; prev = its own, indices saved
        if (!prev) {
            borrow_lock().unlock();
            return false;
        }
        live_ = reinterpret_cast<int32_t *>(prev);
        const int word = live_[0];
        if (word % kMaxTypes != kType || word / kMaxTypes < 0 || word / kMaxTypes >= kDeg) {   // not the default generator: hands off
            (void)setstate(prev);
            borrow_lock().unlock();
            return false;
        }
        r_ = word / kMaxTypes;
        f_ = (r_ + kSep) % kDeg;
        borrowed_ = true;
        return true;
    }
    void give_back() {
        live_[0] = kMaxTypes * r_ + kType;
        (void)setstate(reinterpret_cast<char *>(live_));
        borrowed_ = false;
        borrow_lock().unlock();
    }
    inline int step() {
        int32_t *st = live_ + 1;
        const uint32_t val = (uint32_t)st[f_] + (uint32_t)st[r_];
        st[f_] = (int32_t)val;
        if (++f_ >= kDeg) f_ = 0;
        if (++r_ >= kDeg) r_ = 0;
        return (int)(val >> 1);
    }
    static bool self_check() {
        LibcRandom g;
        if (!g.borrow()) return false;
        int32_t saved[32];
        std::memcpy(saved, g.live_, sizeof saved);
        const int r0 = g.r_;
        int mine[8];
        for (int i = 0; i < 8; ++i) mine[i] = g.step();
        std::memcpy(g.live_, saved, sizeof saved);   // rewind, hand back untouched
        g.r_ = r0;
        g.give_back();
        bool same = true;
        for (int i = 0; i < 8; ++i) same = (rand() == mine[i]) && same;
        if (!g.borrow()) return false;               // rewind the eight rand() calls as well
        std::memcpy(g.live_, saved, sizeof saved);
        g.r_ = r0;
        g.give_back();
        return same;
    }

  public:
    bool fast = false;
    LibcRandom() {}
    LibcRandom(const LibcRandom &) = delete;
    LibcRandom &operator=(const LibcRandom &) = delete;
    explicit LibcRandom(bool want) {
        static const bool ok = self_check() && !(std::getenv("GF_FAST_RAND") && std::getenv("GF_FAST_RAND")[0] == '0');
        fast = want && ok && borrow();
    }
    ~LibcRandom() {
        if (borrowed_) give_back();
    }
    inline int next() { return fast ? step() : rand(); }
};
}  // namespace gf
#endif
