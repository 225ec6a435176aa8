// loop_transport.cpp -- TEST transport for the fan-out entries of include/m17gpu.h (m17gpu_shard_set_library): RCCL's
// ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv between "ranks" that are THREADS of one process on one GPU, so that
// the N > 1 protocol of m17gpu_shard_scatter_iq / m17gpu_shard_gather_packed -- including its error paths -- runs on a
// one-GPU box.  A send is matched with the peer's receive of the same (source, destination) pair in posting order; a
// size mismatch or an operation that finds no partner within the time limit fails its group.  loop_pending() counts the
// operations still waiting for a partner: after a correct exchange, and after a refused one, it must be zero.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Comm { int rank, world; };
struct Post {                       // a send waiting for its receive
    const void *buf; size_t bytes; hipEvent_t ready; hipStream_t stream;
    bool taken = false, failed = false; hipEvent_t done = nullptr;
};
struct Op { bool send; void *buf; size_t bytes; int peer; Comm *comm; hipStream_t stream; };
std::mutex mu;
std::condition_variable cv;
std::map<std::pair<int, int>, std::deque<Post *>> box;      // (source, destination) -> sends in posting order
thread_local std::vector<Op> group;
thread_local int depth = 0;
double limit_s = 20.0;
size_t tsize(int t) { return t == 0 ? 1 : 4; }              // ncclChar / ncclInt32, the two the library uses

int flush()
{
    std::vector<Post *> mine;
    int rc = 0;
    {   // post every send first: two ranks that send to each other and then receive cannot wait for one another
        std::lock_guard<std::mutex> lk(mu);
        for (const Op &o : group) {
            if (!o.send) continue;
            Post *p = new Post{o.buf, o.bytes, nullptr, o.stream};
            (void)hipEventCreateWithFlags(&p->ready, hipEventDisableTiming);
            (void)hipEventRecord(p->ready, o.stream);
            box[{o.comm->rank, o.peer}].push_back(p);
            mine.push_back(p);
        }
        cv.notify_all();
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(limit_s);
    for (const Op &o : group) {
        if (o.send) continue;
        std::unique_lock<std::mutex> lk(mu);
        auto &q = box[{o.peer, o.comm->rank}];
        auto it = q.begin();
        if (!cv.wait_until(lk, deadline, [&] { for (it = q.begin(); it != q.end(); ++it) if (!(*it)->taken) return true; return false; })) { rc = 6; continue; }
        Post *p = *it;
        p->taken = true;
        if (p->bytes != o.bytes) { p->failed = true; rc = 5; cv.notify_all(); continue; }
        (void)hipStreamWaitEvent(o.stream, p->ready, 0);
        (void)hipMemcpyAsync(o.buf, p->buf, o.bytes, hipMemcpyDeviceToDevice, o.stream);
        (void)hipEventCreateWithFlags(&p->done, hipEventDisableTiming);
        (void)hipEventRecord(p->done, o.stream);
        cv.notify_all();
    }
    for (Post *p : mine) {          // a send is complete, for its stream, when the receiver's copy is
        std::unique_lock<std::mutex> lk(mu);
        const bool ok = cv.wait_until(lk, deadline, [&] { return p->taken && (p->done || p->failed); });
        if (!ok || p->failed) { rc = rc ? rc : 6; if (!p->taken) continue; }      // never taken: it stays in the box (loop_pending)
        if (p->done) (void)hipStreamWaitEvent(p->stream, p->done, 0);
        for (auto &kv : box) for (auto i = kv.second.begin(); i != kv.second.end(); ++i) if (*i == p) { kv.second.erase(i); break; }
    }
    group.clear();
    return rc;
}
} // namespace

extern "C" {
int ncclGroupStart() { ++depth; return 0; }
int ncclGroupEnd() { if (--depth > 0) return 0; depth = 0; return flush(); }
int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t s)
{
    group.push_back(Op{true, const_cast<void *>(buf), count * tsize(type), peer, static_cast<Comm *>(comm), s});
    return depth ? 0 : flush();
}
int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t s)
{
    group.push_back(Op{false, buf, count * tsize(type), peer, static_cast<Comm *>(comm), s});
    return depth ? 0 : flush();
}
const char *ncclGetErrorString(int r) { return r == 5 ? "loop transport: sizes of a send and its receive differ" : r == 6 ? "loop transport: no partner within the time limit" : "loop transport error"; }
void *loop_comm(int rank, int world) { return new Comm{rank, world}; }
void loop_set_limit(double seconds) { limit_s = seconds; }
int loop_pending() { std::lock_guard<std::mutex> lk(mu); int n = 0; for (auto &kv : box) n += (int)kv.second.size(); return n; }
}
