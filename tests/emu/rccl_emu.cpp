// TEST INFRASTRUCTURE: a stand-in for librccl.so for the emulated build of libpyrovi (PVI_RCCL_LIB points here; tests only).
// The dozen entry points pyro_amd/csrc/shard.inc resolves, over a POSIX shared-memory segment between the rank PROCESSES of one
// machine.  "Device" pointers are host pointers in the emulation and streams are synchronous, so an operation completes before
// the call (or ncclGroupEnd) returns.  Inside a group, sends are posted first and receives completed afterwards, as RCCL does
// -- a rank that sends to and receives from the same neighbour in one group does not deadlock.
//
// What this can show: that the library's exchange schedule (who sends which rows to whom, in which group, on which buffer; the
// all-gather by broadcasts; the three-double all-reduce) moves the right bytes.  Nothing about xGMI, RCCL's protocols or timing.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>

typedef struct emu_stream* hipStream_t;

namespace {
constexpr int MAXW = 8;
constexpr size_t MAXMSG = 8u << 20;    // bytes per message
struct Chan {                            // one direction of one pair (or one broadcast root)
    std::atomic<unsigned long long> posted, consumed;
    size_t bytes;
};
struct Shared {
    std::atomic<int> attached;
    std::atomic<unsigned long long> bar_count, bar_gen;
    Chan p2p[MAXW][MAXW];                // [src][dst]
    Chan bc[MAXW];                       // [root]; consumed counts readers
    double red[MAXW][64];
    // followed by the data areas: p2p then bc
};
size_t seg_bytes() { return sizeof(Shared) + (size_t)(MAXW * MAXW + MAXW) * MAXMSG; }
struct Comm {
    Shared* sh;
    char* data;
    int rank, world;
    unsigned long long bc_seen[MAXW];
    char name[64];
};
char* p2p_buf(Comm* c, int src, int dst) { return c->data + (size_t)(src * MAXW + dst) * MAXMSG; }
char* bc_buf(Comm* c, int root) { return c->data + (size_t)(MAXW * MAXW + root) * MAXMSG; }

struct Pending {
    int kind;  // 0 send, 1 recv, 2 broadcast
    const void* src;
    void* dst;
    size_t bytes;
    int peer;
    Comm* comm;
};
thread_local int g_group = 0;
thread_local std::vector<Pending> g_pending;

size_t tsize(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
template <class F>
ncclResult_t spin(F ok, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned n = 0;
    while (!ok()) {
        sched_yield();
        if ((++n & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
            fprintf(stderr, "rccl_emu: timed out waiting for %s\n", what);
            return ncclSystemError;
        }
    }
    return ncclSuccess;
}
ncclResult_t do_send(const Pending& p) {
    Comm* c = p.comm;
    if (p.bytes > MAXMSG) return ncclInvalidArgument;
    Chan& ch = c->sh->p2p[c->rank][p.peer];
    ncclResult_t r = spin([&] { return ch.posted.load() == ch.consumed.load(); }, "a free send slot");
    if (r) return r;
    memcpy(p2p_buf(c, c->rank, p.peer), p.src, p.bytes);
    ch.bytes = p.bytes;
    ch.posted.fetch_add(1);
    return ncclSuccess;
}
ncclResult_t do_recv(const Pending& p) {
    Comm* c = p.comm;
    Chan& ch = c->sh->p2p[p.peer][c->rank];
    ncclResult_t r = spin([&] { return ch.posted.load() > ch.consumed.load(); }, "a message");
    if (r) return r;
    if (ch.bytes != p.bytes) {
        fprintf(stderr, "rccl_emu: rank %d expects %zu bytes from %d, the message has %zu\n", c->rank, p.bytes, p.peer, ch.bytes);
        return ncclInvalidArgument;
    }
    memcpy(p.dst, p2p_buf(c, p.peer, c->rank), p.bytes);
    ch.consumed.fetch_add(1);
    return ncclSuccess;
}
ncclResult_t do_bcast_root(const Pending& p) {
    Comm* c = p.comm;
    if (p.bytes > MAXMSG) return ncclInvalidArgument;
    Chan& ch = c->sh->bc[c->rank];
    // every reader of the previous broadcast of this root has taken it
    ncclResult_t r = spin([&] { return ch.consumed.load() == ch.posted.load() * (unsigned long long)(c->world - 1); }, "broadcast readers");
    if (r) return r;
    memcpy(bc_buf(c, c->rank), p.src, p.bytes);
    ch.bytes = p.bytes;
    ch.posted.fetch_add(1);
    if (p.dst != p.src) memcpy(p.dst, p.src, p.bytes);
    return ncclSuccess;
}
ncclResult_t do_bcast_read(const Pending& p) {
    Comm* c = p.comm;
    Chan& ch = c->sh->bc[p.peer];
    ncclResult_t r = spin([&] { return ch.posted.load() > c->bc_seen[p.peer]; }, "a broadcast");
    if (r) return r;
    if (ch.bytes != p.bytes) return ncclInvalidArgument;
    memcpy(p.dst, bc_buf(c, p.peer), p.bytes);
    c->bc_seen[p.peer]++;
    ch.consumed.fetch_add(1);
    return ncclSuccess;
}
ncclResult_t flush() {
    ncclResult_t r = ncclSuccess;
    for (const Pending& p : g_pending)      // everything that only WRITES shared slots first
        if (r == ncclSuccess && p.kind == 0) r = do_send(p);
    for (const Pending& p : g_pending)
        if (r == ncclSuccess && p.kind == 2 && p.peer == p.comm->rank) r = do_bcast_root(p);
    for (const Pending& p : g_pending)
        if (r == ncclSuccess && p.kind == 1) r = do_recv(p);
    for (const Pending& p : g_pending)
        if (r == ncclSuccess && p.kind == 2 && p.peer != p.comm->rank) r = do_bcast_read(p);
    g_pending.clear();
    return r;
}
ncclResult_t submit(const Pending& p) {
    g_pending.push_back(p);
    return g_group ? ncclSuccess : flush();
}
ncclResult_t barrier(Comm* c) {
    Shared* s = c->sh;
    const unsigned long long gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == (unsigned long long)c->world) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return ncclSuccess;
    }
    return spin([&] { return s->bar_gen.load() != gen; }, "a barrier");
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/pvi_rccl_emu_%d_%llx", (int)getpid(),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)seg_bytes())) return ncclSystemError;   // zero-filled: every counter starts at 0
    close(fd);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > MAXW || rank < 0 || rank >= world) return ncclInvalidArgument;
    int fd = -1;
    ncclResult_t r = spin([&] { return (fd = shm_open(id.internal, O_RDWR, 0600)) >= 0; }, "the communicator's segment");
    if (r) return r;
    void* p = mmap(nullptr, seg_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm;
    c->sh = (Shared*)p;
    c->data = (char*)p + sizeof(Shared);
    c->rank = rank;
    c->world = world;
    memset(c->bc_seen, 0, sizeof(c->bc_seen));
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->sh->attached.fetch_add(1);
    r = spin([&] { return c->sh->attached.load() >= world; }, "every rank to attach");
    if (r) return r;
    *out = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    if (c->sh->attached.fetch_sub(1) == 1) shm_unlink(c->name);    // the last rank out removes the segment
    munmap((void*)c->sh, seg_bytes());
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) {
    *n = ((Comm*)comm)->world;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) {
    return submit(Pending{0, buf, nullptr, count * tsize(t), peer, (Comm*)comm});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) {
    return submit(Pending{1, nullptr, buf, count * tsize(t), peer, (Comm*)comm});
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t) {
    Comm* c = (Comm*)comm;
    if (c->world == 1) {
        if (recv != send) memcpy(recv, send, count * tsize(t));
        return ncclSuccess;
    }
    return submit(Pending{2, send, recv, count * tsize(t), root, c});
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t) {
    Comm* c = (Comm*)comm;
    if (t != ncclDouble || count > 64 || (op != ncclMax && op != ncclSum && op != ncclMin)) return ncclInvalidArgument;
    memcpy(c->sh->red[c->rank], send, count * 8);
    ncclResult_t r = barrier(c);
    if (r) return r;
    double out[64];
    for (size_t i = 0; i < count; ++i) {
        double v = c->sh->red[0][i];
        for (int k = 1; k < c->world; ++k) {
            const double x = c->sh->red[k][i];
            v = op == ncclMax ? (x > v ? x : v) : op == ncclMin ? (x < v ? x : v) : v + x;
        }
        out[i] = v;
    }
    r = barrier(c);     // (nobody overwrites its slot before everyone has read)
    memcpy(recv, out, count * 8);
    return r;
}
ncclResult_t ncclGroupStart() {
    ++g_group;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (g_group > 0 && --g_group == 0) return flush();
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (rccl_emu)" : "system error (rccl_emu)";
}
}
