// dist.cpp -- RCCL collective layer and host-side sharding for the row-partitioned solve.
//
// Partition: rank p owns rows [p*ceil(m/P), ...) of A together with
// y, AL, AU for those rows, and rows [p*ceil(n/P), ...) of A^T (= columns of A) together with
// x, c, l, u.  Both stored matrices keep GLOBAL column indices: the x-half gathers from the full y,
// the y-half from the full x_hat, so the only data-path exchange is one in-place all-gather of the
// freshly written slice after each half-step (RCCL ncclAllGather over xGMI), plus one tiny
// all-reduce of the reduction scalars per residual evaluation.  The reference has no multi-GPU
// path at all: this file replaces nothing and is new design.
#include "dist.h"
#include "env.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>
#include <cstring>
#include <iostream>
#include <mutex>
#include <vector>

#include "hprlp_amd.h"
#include "solver.h"

namespace hprlp {

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl() {
    static RcclApi api;
    if (api.handle) return api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) throw std::runtime_error(std::string("cannot load librccl.so: ") + dlerror());
    auto sym = [&](const char *s) {
        void *p = dlsym(api.handle, s);
        if (!p) throw std::runtime_error(std::string("librccl.so lacks symbol ") + s);
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.CommCuDevice = reinterpret_cast<decltype(api.CommCuDevice)>(sym("ncclCommCuDevice"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return api;
}

void check(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL ") + what + ": " + rccl().GetErrorString(r));
}

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    ~RcclComm() override {
        if (comm) (void)rccl().CommDestroy(comm);
    }
    int reported_ranks() const override {
        int v = -1;
        return rccl().CommCount(comm, &v) == ncclSuccess ? v : -1;
    }
    int reported_rank() const override {
        int v = -1;
        return rccl().CommUserRank(comm, &v) == ncclSuccess ? v : -1;
    }
    int reported_device() const override {
        int v = -1;
        return rccl().CommCuDevice(comm, &v) == ncclSuccess ? v : -1;
    }
    void allgather_inplace(double *buf, size_t chunk, hipStream_t s) override {
        check(rccl().AllGather(buf + static_cast<size_t>(rank) * chunk, buf, chunk, ncclDouble, comm, s), "allgather");
    }
    void allreduce_sum(double *buf, int count, hipStream_t s) override {
        check(rccl().AllReduce(buf, buf, static_cast<size_t>(count), ncclDouble, ncclSum, comm, s), "allreduce");
    }
    void exchange(const P2P *ops, int nops, hipStream_t s) override {
        // one group: every send/recv of the list is posted before any has to complete (no ordering deadlock)
        check(rccl().GroupStart(), "group start");
        for (int i = 0; i < nops; ++i) {
            const P2P &o = ops[i];
            if (o.send_bytes) check(rccl().Send(o.send, o.send_bytes, ncclInt8, o.peer, comm, s), "send");
            if (o.recv_bytes) check(rccl().Recv(o.recv, o.recv_bytes, ncclInt8, o.peer, comm, s), "recv");
        }
        check(rccl().GroupEnd(), "group end");
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// in-process group (see dist.h): host barriers + device-to-device copies on the caller's stream
// ------------------------------------------------------------------------------------------------
struct LocalGroup {
    int size = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    long generation = 0;
    bool broken = false;
    std::vector<double *> bufs;
    std::vector<const P2P *> ops;
    std::vector<int> nops;
    std::vector<std::vector<double>> host;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) throw std::runtime_error("local group: another rank failed");
        const long gen = generation;
        if (++waiting == size) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; })) {
            broken = true;
            cv.notify_all();
            throw std::runtime_error("local group: barrier timed out (a rank is missing)");
        }
        if (broken) throw std::runtime_error("local group: another rank failed");
    }
};

namespace {
struct LocalComm : Comm {
    LocalGroup *g = nullptr;
    bool host_blocking() const override { return true; }
    void allgather_inplace(double *buf, size_t chunk, hipStream_t s) override {
        HIP_CHECK(hipStreamSynchronize(s));
        g->bufs[rank] = buf;
        g->barrier();
        for (int p = 0; p < size; ++p)
            if (p != rank)
                HIP_CHECK(hipMemcpyAsync(buf + p * chunk, g->bufs[p] + p * chunk, chunk * sizeof(double),
                                         hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        g->barrier();
    }
    void allreduce_sum(double *buf, int count, hipStream_t s) override {
        std::vector<double> &mine = g->host[rank];
        mine.resize(count);
        HIP_CHECK(hipMemcpyAsync(mine.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        g->barrier();
        std::vector<double> sum(count, 0.0);
        for (int p = 0; p < size; ++p)  // rank order: every rank gets the same bits
            for (int i = 0; i < count; ++i) sum[i] += g->host[p][i];
        g->barrier();
        HIP_CHECK(hipMemcpyAsync(buf, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    void exchange(const P2P *ops, int nops, hipStream_t s) override {
        HIP_CHECK(hipStreamSynchronize(s));
        g->ops[rank] = ops;
        g->nops[rank] = nops;
        g->barrier();
        for (int i = 0; i < nops; ++i) {
            const P2P &o = ops[i];
            if (!o.recv_bytes) continue;
            const P2P *match = nullptr;
            for (int k = 0; k < g->nops[o.peer]; ++k)
                if (g->ops[o.peer][k].peer == rank && g->ops[o.peer][k].send_bytes) match = &g->ops[o.peer][k];
            if (!match || match->send_bytes != o.recv_bytes) {
                {
                    std::lock_guard<std::mutex> lk(g->mu);
                    g->broken = true;
                }
                g->cv.notify_all();
                throw std::runtime_error("local group: unmatched receive");
            }
            HIP_CHECK(hipMemcpyAsync(o.recv, match->send, o.recv_bytes, hipMemcpyDeviceToDevice, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        g->barrier();
    }
};
}  // namespace

LocalGroup *make_local_group(int size) {
    if (size < 1) throw std::runtime_error("local group size must be positive");
    auto *g = new LocalGroup();
    g->size = size;
    g->bufs.assign(size, nullptr);
    g->ops.assign(size, nullptr);
    g->nops.assign(size, 0);
    g->host.resize(size);
    return g;
}
void free_local_group(LocalGroup *g) { delete g; }
Comm *make_local_comm(LocalGroup *g, int rank) {
    if (!g || rank < 0 || rank >= g->size) throw std::runtime_error("bad local group / rank");
    auto *c = new LocalComm();
    c->g = g;
    c->rank = rank;
    c->size = g->size;
    return c;
}

// ------------------------------------------------------------------------------------------------
// host-staged group of PROCESSES on one node (see dist.h): a POSIX shared-memory segment carries a barrier, a table of posted
// sends and one staging area per rank; data moves device -> own area -> peer's device.  No RCCL, no device IPC handle.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr char kShmMagic[8] = {'H', 'P', 'R', 'L', 'P', 'S', 'H', 'M'};
constexpr int kShmMaxRanks = 64;

struct ShmPost {  // a send posted by rank a for rank b: where it lies in a's area
    long off, bytes;
};

struct ShmHeader {
    std::atomic<int> attached;   // ranks that mapped the segment
    std::atomic<int> arrived;    // barrier: arrivals of the current generation
    std::atomic<long> generation;
    std::atomic<int> broken;     // a rank failed or timed out: everybody leaves with an error
    int size;
    long area_bytes;
    double reduce[kShmMaxRanks][32];
    ShmPost post[kShmMaxRanks][kShmMaxRanks];  // [sender][receiver]
};

double shm_timeout_s() {
    const char *e = env_get("HPRLP_DIST_TIMEOUT_S");
    const double v = e ? std::atof(e) : 120.0;
    return v > 0.0 ? v : 120.0;
}

struct ShmComm : Comm {
    std::string name;
    ShmHeader *hd = nullptr;
    char *base = nullptr;   // start of the staging areas
    size_t map_bytes = 0;
    bool host_buffers = false;  // the protocol's own test: "device" pointers are host memory (no HIP call at all)
    bool registered = false;
    bool owner = false;
    int device = -1;
    bool host_blocking() const override { return true; }
    int reported_device() const override { return device; }

    ~ShmComm() override {
        if (registered) (void)hipHostUnregister(hd);
        if (hd) munmap(hd, map_bytes);
        if (owner) shm_unlink(name.c_str());
    }
    char *area(int r) const { return base + static_cast<size_t>(r) * static_cast<size_t>(hd->area_bytes); }
    [[noreturn]] void fail(const std::string &what) {
        hd->broken.store(1);
        throw std::runtime_error("shared-memory group: " + what);
    }
    void barrier() {
        if (hd->broken.load()) throw std::runtime_error("shared-memory group: another rank failed");
        const long gen = hd->generation.load();
        if (hd->arrived.fetch_add(1) + 1 == size) {
            hd->arrived.store(0);
            hd->generation.fetch_add(1);
            return;
        }
        const auto t0 = time_now();
        const double limit = shm_timeout_s();
        for (long spin = 0; hd->generation.load() == gen; ++spin) {
            if (hd->broken.load()) throw std::runtime_error("shared-memory group: another rank failed");
            if (spin > 2000) {  // (the first microseconds by spinning: an exchange per half-step is the iteration's latency)
                std::this_thread::sleep_for(std::chrono::microseconds(spin > 20000 ? 200 : 5));
                if ((spin & 1023) == 0 && time_since(t0) > limit) fail("barrier timed out (a rank is missing)");
            }
        }
    }
    void to_host(void *dst, const void *src, size_t bytes, hipStream_t s) {
        if (!bytes) return;
        if (host_buffers) std::memcpy(dst, src, bytes);
        else HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    }
    void to_device(void *dst, const void *src, size_t bytes, hipStream_t s) {
        if (!bytes) return;
        if (host_buffers) std::memcpy(dst, src, bytes);
        else HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    }
    void sync(hipStream_t s) {
        if (!host_buffers) HIP_CHECK(hipStreamSynchronize(s));
    }
    void allgather_inplace(double *buf, size_t chunk, hipStream_t s) override {
        const size_t bytes = chunk * sizeof(double);
        if (bytes > static_cast<size_t>(hd->area_bytes)) fail("all-gather chunk exceeds the staging area");
        to_host(area(rank), buf + static_cast<size_t>(rank) * chunk, bytes, s);
        sync(s);
        barrier();
        for (int p = 0; p < size; ++p)
            if (p != rank) to_device(buf + static_cast<size_t>(p) * chunk, area(p), bytes, s);
        sync(s);
        barrier();  // nobody overwrites its area before every peer has read it
    }
    void allreduce_sum(double *buf, int count, hipStream_t s) override {
        if (count > 32) fail("all-reduce of more than 32 scalars");
        double mine[32];
        to_host(mine, buf, sizeof(double) * static_cast<size_t>(count), s);
        sync(s);
        for (int i = 0; i < count; ++i) hd->reduce[rank][i] = mine[i];
        barrier();
        double sum[32];
        for (int i = 0; i < count; ++i) sum[i] = 0.0;
        for (int p = 0; p < size; ++p)  // rank order: every rank gets the same bits
            for (int i = 0; i < count; ++i) sum[i] += hd->reduce[p][i];
        barrier();
        to_device(buf, sum, sizeof(double) * static_cast<size_t>(count), s);
        sync(s);
    }
    void exchange(const P2P *ops, int nops, hipStream_t s) override {
        for (int p = 0; p < size; ++p) hd->post[rank][p] = ShmPost{0, 0};
        size_t off = 0;
        for (int i = 0; i < nops; ++i) {
            const P2P &o = ops[i];
            if (o.peer < 0 || o.peer >= size || o.peer == rank) fail("bad peer in an exchange");
            if (!o.send_bytes) continue;
            if (off + o.send_bytes > static_cast<size_t>(hd->area_bytes)) fail("sends exceed the staging area");
            to_host(area(rank) + off, o.send, o.send_bytes, s);
            hd->post[rank][o.peer] = ShmPost{static_cast<long>(off), static_cast<long>(o.send_bytes)};
            off += (o.send_bytes + 63) & ~static_cast<size_t>(63);
        }
        sync(s);
        barrier();
        for (int i = 0; i < nops; ++i) {
            const P2P &o = ops[i];
            if (!o.recv_bytes) continue;
            const ShmPost &q = hd->post[o.peer][rank];
            if (static_cast<size_t>(q.bytes) != o.recv_bytes) fail("unmatched receive");
            to_device(o.recv, area(o.peer) + q.off, o.recv_bytes, s);
        }
        sync(s);
        barrier();
    }
};

}  // namespace

bool is_shm_unique_id(const void *unique_id, size_t id_bytes) {
    return unique_id && id_bytes >= 128 && std::memcmp(unique_id, kShmMagic, sizeof(kShmMagic)) == 0;
}

// A fresh segment name in the 128-byte id: magic, then "/hprlp-<pid>-<clock>".
void shm_make_unique_id(void *out, size_t bytes) {
    if (bytes < 128) throw std::runtime_error("unique-id buffer too small");
    std::memset(out, 0, bytes);
    std::memcpy(out, kShmMagic, sizeof(kShmMagic));
    const long long t = std::chrono::duration_cast<std::chrono::nanoseconds>(clock_type::now().time_since_epoch()).count();
    std::snprintf(static_cast<char *>(out) + sizeof(kShmMagic), 100, "/hprlp-%ld-%llx", static_cast<long>(getpid()), static_cast<unsigned long long>(t));
}

// Rank 0 creates the segment, the others wait for it; area_bytes: staging room per rank (pages are touched only where used).
// device < 0: the buffers handed to the collectives are HOST memory (the transport's own test, no HIP call).
Comm *make_shm_comm(int rank, int size, const void *unique_id, size_t id_bytes, size_t area_bytes, int device) {
    if (!is_shm_unique_id(unique_id, id_bytes)) throw std::runtime_error("not a shared-memory group id");
    if (size < 1 || size > kShmMaxRanks || rank < 0 || rank >= size) throw std::runtime_error("shared-memory group: bad rank / size");
    char nm[101];
    std::memcpy(nm, static_cast<const char *>(unique_id) + sizeof(kShmMagic), 100);
    nm[100] = 0;
    area_bytes = (area_bytes + 4095) & ~static_cast<size_t>(4095);
    const size_t head = (sizeof(ShmHeader) + 4095) & ~static_cast<size_t>(4095);
    const size_t total = head + area_bytes * static_cast<size_t>(size);
    auto c = std::unique_ptr<ShmComm>(new ShmComm());
    c->rank = rank;
    c->size = size;
    c->name = nm;
    c->host_buffers = device < 0;
    c->device = device;
    int fd = -1;
    const auto t0 = time_now();
    if (rank == 0) {
        fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) throw std::runtime_error(std::string("shm_open(create) failed: ") + std::strerror(errno));
        c->owner = true;
        if (ftruncate(fd, static_cast<off_t>(total)) != 0) {
            close(fd);
            throw std::runtime_error(std::string("ftruncate of the shared segment failed: ") + std::strerror(errno));
        }
    } else {
        for (;;) {  // until rank 0 has created and sized it
            fd = shm_open(nm, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= total) break;
                close(fd);
                fd = -1;
            }
            if (time_since(t0) > shm_timeout_s()) throw std::runtime_error("shared-memory group: rank 0's segment did not appear");
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void *mp = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mp == MAP_FAILED) throw std::runtime_error(std::string("mmap of the shared segment failed: ") + std::strerror(errno));
    c->hd = static_cast<ShmHeader *>(mp);
    c->map_bytes = total;
    c->base = static_cast<char *>(mp) + head;
    if (rank == 0) {  // (a fresh segment is zero-filled: counters start at 0)
        c->hd->area_bytes = static_cast<long>(area_bytes);
        c->hd->size = size;
    }
    c->hd->attached.fetch_add(1);
    while (c->hd->attached.load() < size || c->hd->size != size) {  // everybody has it mapped (rank 0 may unlink the name at exit)
        if (c->hd->broken.load()) throw std::runtime_error("shared-memory group: another rank failed");
        if (time_since(t0) > shm_timeout_s()) {
            c->hd->broken.store(1);
            throw std::runtime_error("shared-memory group: not every rank attached");
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (static_cast<size_t>(c->hd->area_bytes) != area_bytes) throw std::runtime_error("shared-memory group: ranks disagree on the staging size");
    if (!c->host_buffers) {
        HIP_CHECK(hipSetDevice(device));
        // pinned staging: the copies run as DMA at link rate; a refusal only makes them slower
        if (hipHostRegister(mp, total, hipHostRegisterDefault) == hipSuccess) c->registered = true;
        else (void)hipGetLastError();
    }
    return c.release();
}

// One id per 128 bytes of the buffer, at most two: the second one is for the exchange stream's own communicator
// (Solver::xcomm) -- no communicator is driven from two streams.
void rccl_get_unique_id(void *out, size_t bytes) {
    static_assert(sizeof(ncclUniqueId) == 128, "the launcher broadcasts 128 bytes per id");
    if (bytes < sizeof(ncclUniqueId)) throw std::runtime_error("unique-id buffer too small");
    const size_t ids = std::min<size_t>(bytes / sizeof(ncclUniqueId), 2);
    for (size_t k = 0; k < ids; ++k) {
        ncclUniqueId id;
        check(rccl().GetUniqueId(&id), "get unique id");
        std::memcpy(static_cast<char *>(out) + k * sizeof(id), &id, sizeof(id));
    }
}

Comm *make_rccl_comm(int rank, int size, const void *unique_id, size_t id_bytes, int device) {
    if (id_bytes < sizeof(ncclUniqueId)) throw std::runtime_error("unique id too short");
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    auto *c = new RcclComm();
    c->rank = rank;
    c->size = size;
    try {
        check(rccl().CommInitRank(&c->comm, size, id, rank), "comm init");
    } catch (...) {
        delete c;
        throw;
    }
    return c;
}

}  // namespace hprlp

using namespace hprlp;

// ------------------------------------------------------------------------------------------------
// host-only sharding helpers (no GPU needed; covered by the gloo tests)
// ------------------------------------------------------------------------------------------------
extern "C" int hprlp_partition(int total, int parts, int rank, int *offset, int *count) {
    if (total < 0 || parts <= 0 || rank < 0 || rank >= parts || !offset || !count) return -1;
    const int chunk = (total + parts - 1) / parts;
    const int off = std::min(total, rank * chunk);
    *offset = rank * chunk;
    *count = std::max(0, std::min(total, rank * chunk + chunk) - off);
    return chunk;
}

extern "C" void hprlp_free_shard(hprlp_shard *s) {
    if (!s) return;
    std::free(s->A_rowptr); std::free(s->A_col); std::free(s->A_val);
    std::free(s->AT_rowptr); std::free(s->AT_col); std::free(s->AT_val);
    std::free(s->AL); std::free(s->AU); std::free(s->l); std::free(s->u); std::free(s->c);
    std::memset(s, 0, sizeof(*s));
}

template <class T>
static T *copy_out(const T *src, size_t n) {
    T *d = static_cast<T *>(std::malloc((n ? n : 1) * sizeof(T)));
    if (!d) throw std::runtime_error("out of host memory");
    if (n) std::memcpy(d, src, n * sizeof(T));
    return d;
}

static void slice_rows(int r0, int cnt, const int *rp, const int *ci, const double *v, int **orp, int **oci,
                       double **ov) {
    const int k0 = rp[r0], k1 = rp[r0 + cnt];
    int *p = static_cast<int *>(std::malloc((static_cast<size_t>(cnt) + 1) * sizeof(int)));
    if (!p) throw std::runtime_error("out of host memory");
    for (int i = 0; i <= cnt; ++i) p[i] = rp[r0 + i] - k0;
    *orp = p;
    *oci = copy_out(ci + k0, static_cast<size_t>(k1 - k0));
    *ov = copy_out(v + k0, static_cast<size_t>(k1 - k0));
}

extern "C" int hprlp_extract_shard(const LP_info_cpu *model, int rank, int size, hprlp_shard *out) {
    try {
        if (!model || !model->A || !out || size <= 0 || rank < 0 || rank >= size)
            throw std::runtime_error("hprlp_extract_shard: bad arguments");
        std::memset(out, 0, sizeof(*out));
        const int m = model->m, n = model->n;
        const sparseMatrix *A = model->A;
        out->m = m;
        out->n = n;
        hprlp_partition(m, size, rank, &out->row_off, &out->m_loc);
        hprlp_partition(n, size, rank, &out->col_off, &out->n_loc);
        const int r0 = std::min(out->row_off, m), c0 = std::min(out->col_off, n);
        slice_rows(r0, out->m_loc, A->rowPtr, A->colIndex, A->value, &out->A_rowptr, &out->A_col, &out->A_val);
        // this rank's rows of A^T = the entries of A whose column lies in its range: no full transpose
        std::vector<int> trp, tci;
        std::vector<double> tv;
        csr_transpose_range_host(m, c0, c0 + out->n_loc, A->rowPtr, A->colIndex, A->value, trp, tci, tv);
        slice_rows(0, out->n_loc, trp.data(), tci.data(), tv.data(), &out->AT_rowptr, &out->AT_col, &out->AT_val);
        out->AL = copy_out(model->AL + r0, out->m_loc);
        out->AU = copy_out(model->AU + r0, out->m_loc);
        out->l = copy_out(model->l + c0, out->n_loc);
        out->u = copy_out(model->u + c0, out->n_loc);
        out->c = copy_out(model->c + c0, out->n_loc);
        out->obj_constant = model->obj_constant;
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        if (out) hprlp_free_shard(out);
        return -1;
    }
}

extern "C" int hprlp_dist_unique_id(void *out, int bytes) {
    try {
        // HPRLP_DIST_TRANSPORT=shm: the id names a shared-memory segment (host-staged group of processes on one node, no RCCL)
        const char *tr = env_get("HPRLP_DIST_TRANSPORT");
        if (tr && std::strcmp(tr, "shm") == 0) {
            if (!out) throw std::runtime_error("null id buffer");
            shm_make_unique_id(out, static_cast<size_t>(bytes));
            return 0;
        }
        rccl_get_unique_id(out, static_cast<size_t>(bytes));
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return -1;
    }
}
