// TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): the execution engine of the source-level HIP emulation and a synchronous,
// one-device runtime API.  Linked with the transformed kernel sources into tests/emu/_build/libpyrovi_emu.so by build_emu.py.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- context switch (x86-64 System V: callee-saved registers only) ------------------------------------------------------------
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

// AddressSanitizer build (build_emu.py --asan): every stack switch is announced, so that the sanitizer knows which stack it is on
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/common_interface_defs.h>
#define EMU_ASAN 1
#endif
#endif
#ifndef EMU_ASAN
#define EMU_ASAN 0
#endif

namespace emu {
thread_local FiberPub* g_me = nullptr;
thread_local BlockCtx* g_blk = nullptr;

enum State { ST_RUN, ST_WAVE, ST_BAR, ST_DONE };
struct Fiber {
    FiberPub pub;
    void* sp = nullptr;
    char* stack = nullptr;
    State state = ST_DONE;
    Op op = OP_SHFL;
    const void* pcs[24] = {};   // return addresses of the call stack at the parked collective, OUTERMOST first ("program position")
    int npcs = 0;
    unsigned long long val = 0, aux = 0, res = 0;
    unsigned p[4] = {0, 0, 0, 0};
    void* asan_fake = nullptr;
};
static constexpr uintptr_t COOP_BASE = 0x900000, COOP_END = 0xFF0000;   // LDS arenas of cooperative launches
static constexpr size_t STACK_BYTES = 256 * 1024, MAX_THREADS = 1024, LDS_BYTES = 256 * 1024, GUARD = 128 * 1024;   // (a workgroup's LDS is at most 160 KiB)
struct Worker {                 // per host thread: fiber stacks, LDS arena, the workgroup being run
    char* stacks = nullptr;
    char* lds = nullptr;
    std::vector<Fiber> f;
    void* sched_sp = nullptr;
    const std::function<void()>* body = nullptr;
    BlockCtx ctx;
    int cur = -1;
    const char* kname = "";
    const void* sched_stack = nullptr;   // (ASan: the scheduler's stack, learnt by the first fiber that starts)
    size_t sched_size = 0;
    std::vector<std::vector<unsigned>> lds_trace;   // (--ldstrace builds) per work-item: the LDS byte addresses it read, in order
};
static thread_local Worker* g_w = nullptr;
static std::atomic<int> g_arena_slot{0};
std::atomic<unsigned long long> g_inactive_reads{0};

static Worker* worker() {
    if (g_w) return g_w;
    Worker* w = new Worker;
    w->stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w->stacks == MAP_FAILED) {
        fprintf(stderr, "emu: cannot map fiber stacks\n");
        abort();
    }
    // The dynamic LDS of the workgroup lives below 16 MiB: the kernels keep LDS byte addresses in 32-bit integers AND in float32
    // registers (exact up to 2^24; real LDS addresses are below 160 KiB).
    // On the GPU an LDS read beyond the workgroup's allocation returns zeros and a write is dropped; the 4-D window sweep relies
    // on that for cells whose value it discards (a masked cell's address can lie a few slots outside the window).  Here the arena
    // sits between two read-only zero pages ranges of GUARD bytes: such reads see zeros, an out-of-range WRITE faults (a bug).
    void* p = MAP_FAILED;
    for (int tries = 0; tries < 16 && p == MAP_FAILED; ++tries) {
        const int slot = g_arena_slot.fetch_add(1);
        char* want = (char*)(uintptr_t)(0x100000ull + (unsigned long long)slot * (LDS_BYTES + 2 * GUARD));
        if ((uintptr_t)want + LDS_BYTES + 2 * GUARD > COOP_BASE) break;
        p = mmap(want, LDS_BYTES + 2 * GUARD, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
        if (p != MAP_FAILED && p != (void*)want) {
            munmap(p, LDS_BYTES + 2 * GUARD);
            p = MAP_FAILED;
        }
        if (p != MAP_FAILED) {
            mprotect(want + GUARD, LDS_BYTES, PROT_READ | PROT_WRITE);
            p = want + GUARD;
        }
    }
    if (p == MAP_FAILED) {
        fprintf(stderr, "emu: cannot map an LDS arena below 16 MiB (vm.mmap_min_addr? more than 16 host threads?)\n");
        abort();
    }
    w->lds = (char*)p;
    w->f.resize(MAX_THREADS);
    g_w = w;
    return w;
}

static void fiber_entry() {
    Worker* w = g_w;
    Fiber& me = w->f[(size_t)w->cur];
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &w->sched_stack, &w->sched_size);
#endif
    (*w->body)();
    me.state = ST_DONE;
    void* dummy;
#if EMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, w->sched_stack, w->sched_size);   // (nullptr: this fiber's stack is done with)
#endif
    emu_switch(&dummy, w->sched_sp);
    abort();  // a finished fiber is never resumed
}

static inline void park(State s) {
    Worker* w = g_w;
    Fiber& me = w->f[(size_t)w->cur];
    me.state = s;
#if EMU_ASAN
    __sanitizer_start_switch_fiber(&me.asan_fake, w->sched_stack, w->sched_size);
#endif
    emu_switch(&me.sp, w->sched_sp);
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(me.asan_fake, &w->sched_stack, &w->sched_size);
#endif
}

// Program position of a parked lane: the chain of return addresses from the kernel's outermost frame down to the collective
// (frame pointers: the emulated build is compiled with -fno-omit-frame-pointer; a fiber's first frame has a null saved rbp).
// Compared lexicographically it orders "earlier in the kernel" before "later in the kernel" for code the host compiler laid out in
// source order -- the same assumption as the minimum-PC reconvergence heuristic of SIMT simulators.
static inline void record_position(Fiber& me) {
    const void* tmp[24];
    int n = 0;
    void** fp = (void**)__builtin_frame_address(0);
    while (fp && n < 24) {
        tmp[n++] = fp[1];
        void** next = (void**)fp[0];
        if (next <= fp) break;
        fp = next;
    }
    me.npcs = n;
    for (int i = 0; i < n; ++i) me.pcs[i] = tmp[n - 1 - i];
}
static inline bool position_less(const Fiber& a, const Fiber& b) {
    const int n = std::min(a.npcs, b.npcs);
    for (int i = 0; i < n; ++i)
        if (a.pcs[i] != b.pcs[i]) return (uintptr_t)a.pcs[i] < (uintptr_t)b.pcs[i];
    return a.npcs < b.npcs;
}

__attribute__((noinline)) unsigned long long collective(Op op, unsigned long long val, unsigned long long aux, unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
    Worker* w = g_w;
    Fiber& me = w->f[(size_t)w->cur];
    me.op = op;
    me.val = val;
    me.aux = aux;
    me.p[0] = p0;
    me.p[1] = p1;
    me.p[2] = p2;
    me.p[3] = p3;
    record_position(me);
    park(ST_WAVE);
    return me.res;
}
void barrier() { park(ST_BAR); }

// ---- LDS trace --------------------------------------------------------------------------------------------------------------------
static std::atomic<unsigned long long> g_lds_reads{0}, g_lds_cycles{0}, g_lds_ideal{0};
void lds_note(unsigned byte_adr) {
    Worker* w = g_w;
    if ((int)w->lds_trace.size() <= w->cur) w->lds_trace.resize(MAX_THREADS);
    w->lds_trace[(size_t)w->cur].push_back(byte_adr);
}
// The k-th traced read of every lane of a wave is one ds_read_b64 of the wave (the lanes of a wave run the same action loop).
// gfx950 serves it in two groups of 32 lanes; a group takes one LDS cycle when no two DISTINCT 8-byte slots fall on the same pair
// of banks (slot mod 32), N cycles for N of them (tools/lds_conflict_model4.py, calibrated on tools/ldsbank.hip in round 4).
static void lds_account(Worker* w, int T) {
    if (w->lds_trace.empty()) return;
    unsigned long long reads = 0, cycles = 0, ideal = 0;
    for (int base = 0; base < T; base += 64) {
        const int nl = std::min(64, T - base);
        size_t len = 0;
        for (int l = 0; l < nl; ++l) len = std::max(len, w->lds_trace[(size_t)(base + l)].size());
        static const int dump = getenv("PVI_EMU_LDSDUMP") ? atoi(getenv("PVI_EMU_LDSDUMP")) : 0;
        static std::atomic<int> dumped{0};
        for (size_t k = 0; k < len; ++k) {
            ++reads;
            if (dump && base == 0 && k >= 16 && k < 20 && dumped.fetch_add(1) < dump) {
                fprintf(stderr, "LDSDUMP block (%u) read %zu:", w->ctx.bid.x, k);
                for (int l = 0; l < nl; ++l) {
                    const auto& t = w->lds_trace[(size_t)(base + l)];
                    fprintf(stderr, " %d", k < t.size() ? (int)(t[k] >> 3) - (int)((uintptr_t)w->lds >> 3) : -1);
                }
                fprintf(stderr, "\n");
            }
            for (int g = 0; g < 64; g += 32) {
                unsigned slots[32];
                int n = 0;
                for (int l = g; l < std::min(g + 32, nl); ++l) {
                    const auto& t = w->lds_trace[(size_t)(base + l)];
                    if (k >= t.size()) continue;
                    const unsigned s = t[k] >> 3;
                    bool dup = false;
                    for (int i = 0; i < n && !dup; ++i) dup = slots[i] == s;
                    if (!dup) slots[n++] = s;
                }
                if (!n) continue;
                int cnt[32] = {0}, mx = 0;
                for (int i = 0; i < n; ++i) mx = std::max(mx, ++cnt[slots[i] & 31u]);
                cycles += (unsigned long long)mx;
                ++ideal;
            }
        }
    }
    for (int i = 0; i < T; ++i) w->lds_trace[(size_t)i].clear();
    g_lds_reads.fetch_add(reads);
    g_lds_cycles.fetch_add(cycles);
    g_lds_ideal.fetch_add(ideal);
}
void yield_host() { sched_yield(); }

static inline void resume(Worker* w, int i) {
    w->cur = i;
    g_me = &w->f[(size_t)i].pub;
#if EMU_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, w->f[(size_t)i].stack, STACK_BYTES);
#endif
    emu_switch(&w->sched_sp, w->f[(size_t)i].sp);
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

// one group of lanes of a wave (the same operation at the same call site) completes its collective operation
static void resolve(Worker* w, int base, int nl, unsigned long long active) {
    Fiber* f = &w->f[(size_t)base];
    auto act = [&](int l) { return l >= 0 && l < nl && ((active >> l) & 1ull); };
    int first = 0;
    while (!act(first)) ++first;
    const Op op = f[first].op;
    unsigned long long ballot = 0;
    if (op == OP_BALLOT)
        for (int l = 0; l < nl; ++l)
            if (act(l) && (f[l].val & 1ull)) ballot |= 1ull << l;
    for (int l = 0; l < nl; ++l) {
        if (!act(l)) continue;
        Fiber& me = f[l];
        if (me.op != op) {
            fprintf(stderr, "emu: %s: lanes of one wave at different collective operations (%d vs %d)\n", w->kname, (int)op, (int)me.op);
            abort();
        }
        auto from = [&](int src, unsigned long long fallback) {
            if (act(src)) return f[src].val;
            g_inactive_reads.fetch_add(1, std::memory_order_relaxed);
            return fallback;
        };
        switch (op) {
            case OP_BALLOT: me.res = ballot; break;
            case OP_SYNC: me.res = 0; break;
            case OP_SHFL: {
                const int width = me.p[1] ? (int)me.p[1] : 64;
                me.res = from((l / width) * width + (int)(me.p[0] % (unsigned)width), me.val);
                break;
            }
            case OP_SHFL_XOR: {
                const int width = me.p[1] ? (int)me.p[1] : 64;
                const int src = l ^ (int)me.p[0];
                me.res = (src / width == l / width) ? from(src, me.val) : me.val;
                break;
            }
            case OP_READLANE: me.res = from((int)me.p[0], 0ull); break;
            case OP_READFIRST: me.res = f[first].val; break;
            case OP_DPP: {
                const unsigned ctrl = me.p[0], row_mask = me.p[1], bank_mask = me.p[2];
                const bool bound = me.p[3] != 0;
                const int row = l >> 4, inrow = l & 15;
                int src = -1;  // -1: no valid source lane
                if (ctrl <= 0xff)
                    src = (l & ~3) | (int)((ctrl >> (2 * (l & 3))) & 3u);
                else if (ctrl >= 0x101 && ctrl <= 0x10f)
                    src = inrow + (int)(ctrl & 15u) <= 15 ? l + (int)(ctrl & 15u) : -1;      // row_shl
                else if (ctrl >= 0x111 && ctrl <= 0x11f)
                    src = inrow - (int)(ctrl & 15u) >= 0 ? l - (int)(ctrl & 15u) : -1;       // row_shr
                else if (ctrl >= 0x121 && ctrl <= 0x12f)
                    src = row * 16 + ((inrow - (int)(ctrl & 15u)) & 15);                      // row_ror
                else if (ctrl == 0x130)
                    src = l + 1 < 64 ? l + 1 : -1;  // wave_shl:1
                else if (ctrl == 0x134)
                    src = (l + 1) & 63;             // wave_rol:1
                else if (ctrl == 0x138)
                    src = l - 1;                    // wave_shr:1
                else if (ctrl == 0x13c)
                    src = (l - 1) & 63;             // wave_ror:1
                else if (ctrl == 0x140)
                    src = row * 16 + 15 - inrow;    // row_mirror
                else if (ctrl == 0x141)
                    src = (l & ~7) + 7 - (l & 7);   // row_half_mirror
                else if (ctrl == 0x142)
                    src = row > 0 ? row * 16 - 1 : -1;  // row_bcast:15
                else if (ctrl == 0x143)
                    src = l >= 32 ? 31 : -1;            // row_bcast:31
                else {
                    fprintf(stderr, "emu: dpp_ctrl 0x%x is not modelled\n", ctrl);
                    abort();
                }
                const bool enabled = ((row_mask >> row) & 1u) && ((bank_mask >> (inrow >> 2)) & 1u);
                if (!enabled)
                    me.res = me.aux;
                else if (src >= 0 && act(src))
                    me.res = f[src].val;
                else
                    me.res = bound ? 0ull : me.aux;
                break;
            }
        }
    }
    for (int l = 0; l < nl; ++l)
        if (act(l)) f[l].state = ST_RUN;
}

static void run_block(Worker* w, Idx3 bid, Idx3 bdim, Idx3 gdim, size_t lds, const std::function<void()>& body, const char* name) {
    const int T = (int)(bdim.x * bdim.y * bdim.z);
    if (T < 1 || T > (int)MAX_THREADS || lds > LDS_BYTES) {
        fprintf(stderr, "emu: %s: workgroup of %d work-items / %zu LDS bytes is not modelled\n", name, T, lds);
        abort();
    }
    w->ctx.bid = bid;
    w->ctx.bdim = bdim;
    w->ctx.gdim = gdim;
    w->ctx.dyn_lds = w->lds;
    w->ctx.dyn_bytes = (unsigned)lds;
    w->body = &body;
    w->kname = name;
    g_blk = &w->ctx;
    for (int i = 0; i < T; ++i) {
        Fiber& f = w->f[(size_t)i];
        f.pub.flat = (unsigned)i;
        f.pub.lane = (unsigned)i & 63u;
        f.pub.wave = (unsigned)i >> 6;
        f.pub.tid.x = (unsigned)i % bdim.x;
        f.pub.tid.y = ((unsigned)i / bdim.x) % bdim.y;
        f.pub.tid.z = (unsigned)i / (bdim.x * bdim.y);
        f.stack = w->stacks + (size_t)i * STACK_BYTES;
        void** top = (void**)(f.stack + STACK_BYTES);  // 16-byte aligned
        top[-2] = (void*)&fiber_entry;                   // the `ret` of emu_switch: rsp = top - 8 on entry, as after a call
        for (int k = 3; k <= 8; ++k) top[-k] = nullptr;  // rbp, rbx, r12 .. r15
        f.sp = (void*)(top - 8);
        f.state = ST_RUN;
    }
    const int nwaves = (T + 63) / 64;
    for (;;) {
        for (int wv = 0; wv < nwaves; ++wv) {
            const int base = wv * 64, nl = std::min(64, T - base);
            for (;;) {
                for (int l = 0; l < nl; ++l)
                    while (w->f[(size_t)(base + l)].state == ST_RUN) resume(w, base + l);   // (until it parks or ends)
                // Lanes at wave collectives.  Lanes of one wave can be parked at DIFFERENT operations: a lane whose loop body took
                // a branch with one more ballot is behind the lanes that already wait at the reduction behind the loop.  The lane
                // that is EARLIEST in the program leads (nobody is behind it), and the lanes at the same operation with the same
                // immediate operands complete it together -- "the same operation", not "the same call site": the host compiler
                // duplicates code (the statistics tail of a kernel exists once per branch that reaches it), and lanes in different
                // copies of one source-level operation still execute it together on the GPU.
                int lead = -1;
                for (int l = 0; l < nl; ++l) {
                    const Fiber& f = w->f[(size_t)(base + l)];
                    if (f.state == ST_WAVE && (lead < 0 || position_less(f, w->f[(size_t)(base + lead)]))) lead = l;
                }
                if (lead < 0) break;
                unsigned long long active = 0;
                const Fiber& ld = w->f[(size_t)(base + lead)];
                for (int l = 0; l < nl; ++l) {
                    const Fiber& f = w->f[(size_t)(base + l)];
                    const bool same = f.op == ld.op && (f.op == OP_BALLOT || f.op == OP_READFIRST || f.op == OP_SYNC ||
                                                        (f.p[0] == ld.p[0] && f.p[1] == ld.p[1] && f.p[2] == ld.p[2] && f.p[3] == ld.p[3]));
                    if (f.state == ST_WAVE && same) active |= 1ull << l;
                }
                resolve(w, base, nl, active);
            }
        }
        bool any = false;
        for (int i = 0; i < T; ++i)
            if (w->f[(size_t)i].state == ST_BAR) {
                w->f[(size_t)i].state = ST_RUN;
                any = true;
            }
        if (!any) break;  // every work-item has ended
    }
    lds_account(w, T);
    g_me = nullptr;
}

// ---- host threads -----------------------------------------------------------------------------------------------------------------
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    unsigned long long epoch = 0;
    int pending = 0;
    std::function<void()> job;
    bool stop = false;
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i)
            th.emplace_back([this] {
                unsigned long long seen = 0;
                for (;;) {
                    std::function<void()> j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || epoch != seen; });
                        if (stop) return;
                        seen = epoch;
                        j = job;
                    }
                    j();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) done_cv.notify_all();
                    }
                }
            });
    }
    void start(const std::function<void()>& j) {
        std::lock_guard<std::mutex> lk(mu);
        job = j;
        pending = (int)th.size();
        ++epoch;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
};
static Pool* pool() {
    static Pool* p = [] {
        const char* e = getenv("PVI_EMU_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        n = std::max(1, std::min(n, 16));
        return new Pool(n - 1);     // (the launching thread works too)
    }();
    return p;
}

static std::atomic<unsigned long long> g_launches{0};
void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body, const char* name) {
    const unsigned long long nb = (unsigned long long)grid.x * grid.y * grid.z;
    const Idx3 bdim{block.x, block.y, block.z}, gdim{grid.x, grid.y, grid.z};
    g_launches.fetch_add(1);
    std::atomic<unsigned long long> next{0};
    auto work = [&]() {
        Worker* w = worker();
        for (;;) {
            const unsigned long long b = next.fetch_add(1);
            if (b >= nb) break;
            const Idx3 bid{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y))};
            run_block(w, bid, bdim, gdim, lds, body, name);
        }
    };
    Pool* p = pool();
    if (nb <= 1 || p->th.empty()) {
        work();
    } else {  // the pool's threads and this one drain the same counter
        p->start(work);
        work();
        p->wait();
    }
}
// ---- cooperative launches: every workgroup resident at once, one host thread each (grid barriers spin on atomics) -------------------
static std::mutex g_reg_mu;
static std::vector<std::pair<const void*, Invoker>> g_reg;
void register_kernel(const void* fn, Invoker inv) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (auto& e : g_reg)
        if (e.first == fn) return;
    g_reg.emplace_back(fn, inv);
}
static Invoker find_kernel(const void* fn) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (auto& e : g_reg)
        if (e.first == fn) return e.second;
    return nullptr;
}
static size_t coop_stride(size_t lds) {
    size_t stride = 16 * 1024;
    while (stride < lds + 8192) stride *= 2;       // (the arena + a zero page on either side)
    return stride;
}
int coop_capacity(size_t lds) {   // workgroups of a cooperative launch the LDS region below 16 MiB has room for
    const size_t n = (COOP_END - COOP_BASE) / coop_stride(lds);
    return (int)(n > 512 ? 512 : n);
}
static int launch_coop(const void* fn, dim3 grid, dim3 block, void** args, size_t lds) {
    Invoker inv = find_kernel(fn);
    if (!inv) return 1;
    const unsigned long long nb = (unsigned long long)grid.x * grid.y * grid.z;
    const size_t T = (size_t)block.x * block.y * block.z;
    const size_t stride = coop_stride(lds);
    if (nb == 0 || nb > (unsigned long long)coop_capacity(lds) || T > MAX_THREADS) {
        fprintf(stderr, "emu: cooperative launch of %llu workgroups x %zu work-items x %zu LDS bytes is beyond the emulation\n", nb, T, lds);
        return 1;
    }
    void* region = mmap((void*)COOP_BASE, nb * stride, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
    if (region != (void*)COOP_BASE) {
        fprintf(stderr, "emu: cannot map the LDS region of a cooperative launch\n");
        return 1;
    }
    g_launches.fetch_add(1);
    const std::function<void()> body = [=]() { inv(fn, args); };
    const Idx3 bdim{block.x, block.y, block.z}, gdim{grid.x, grid.y, grid.z};
    std::vector<std::thread> th;
    for (unsigned long long b = 0; b < nb; ++b)
        th.emplace_back([&, b]() {
            Worker w;     // a worker of its own: `static thread_local` (= __shared__) objects are per workgroup this way
            w.stacks = (char*)mmap(nullptr, STACK_BYTES * T, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (w.stacks == MAP_FAILED) abort();
            w.lds = (char*)COOP_BASE + b * stride + 4096;
            mprotect(w.lds, stride - 8192, PROT_READ | PROT_WRITE);
            w.f.resize(T);
            Worker* keep = g_w;
            g_w = &w;
            const Idx3 bid{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y))};
            run_block(&w, bid, bdim, gdim, lds, body, "cooperative kernel");
            g_w = keep;
            munmap(w.stacks, STACK_BYTES * T);
        });
    for (auto& t : th) t.join();
    munmap(region, nb * stride);
    return 0;
}
}  // namespace emu

extern "C" void emu_lds_stats(unsigned long long* out3, int reset) {   // {wave-level reads, LDS cycles, cycles without any conflict}
    out3[0] = emu::g_lds_reads.load();
    out3[1] = emu::g_lds_cycles.load();
    out3[2] = emu::g_lds_ideal.load();
    if (reset) {
        emu::g_lds_reads.store(0);
        emu::g_lds_cycles.store(0);
        emu::g_lds_ideal.store(0);
    }
}
extern "C" unsigned long long emu_inactive_lane_reads() { return emu::g_inactive_reads.load(); }
extern "C" unsigned long long emu_launch_count() { return emu::g_launches.load(); }

// ---- runtime API ------------------------------------------------------------------------------------------------------------------
struct emu_stream { int unused; };
struct emu_event { std::chrono::steady_clock::time_point t; };
static int emu_devices() {   // PVI_EMU_DEVICES: how many "devices" a process sees (all of them are this host; default 1)
    static const int n = [] {
        const char* e = getenv("PVI_EMU_DEVICES");
        const int v = e ? atoi(e) : 1;
        return v < 1 ? 1 : v > 64 ? 64 : v;
    }();
    return n;
}
hipError_t hipGetDeviceCount(int* n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d) { return d >= 0 && d < emu_devices() ? hipSuccess : hipErrorInvalidValue; }
// live "device" allocations, for tests that inspect or poison device memory (emu_alloc_count / emu_alloc_get)
static std::mutex g_alloc_mu;
static std::vector<std::pair<void*, size_t>> g_allocs;
hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 256)) return hipErrorOutOfMemory;
    memset(q, 0xA5, n ? n : 256);   // device memory is not zeroed: make a read of uninitialised memory visible
    *p = q;
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_allocs.emplace_back(q, n);
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        for (size_t i = 0; i < g_allocs.size(); ++i)
            if (g_allocs[i].first == p) {
                g_allocs.erase(g_allocs.begin() + (long)i);
                break;
            }
    }
    free(p);
    return hipSuccess;
}
extern "C" long emu_alloc_count() {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    return (long)g_allocs.size();
}
extern "C" int emu_alloc_get(long i, void** p, size_t* n) {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    if (i < 0 || i >= (long)g_allocs.size()) return -1;
    *p = g_allocs[(size_t)i].first;
    *n = g_allocs[(size_t)i].second;
    return 0;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu_stream; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "invalid value"; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
    *v = 1;   // ONE "compute unit" (hipOccupancyMaxActiveBlocksPerMultiprocessor then is the whole device's capacity); cooperative launches: yes
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
namespace emu { int coop_capacity(size_t lds); }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t lds) { *n = emu::coop_capacity(lds); return hipSuccess; }
hipError_t hipLaunchCooperativeKernel(const void* fn, dim3 grid, dim3 block, void** args, unsigned lds, hipStream_t) {
    return emu::launch_coop(fn, grid, block, args, lds) ? hipErrorInvalidValue : hipSuccess;
}
