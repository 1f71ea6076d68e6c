// node_comm.hpp -- the ranks of `smoothMesh -parallel` on one node: one PROCESS per sub-domain / GPU, as the reference runs
// under `mpirun -np N smoothMesh -parallel` (testcase/run_parallel:19).  No MPI in this image, so the front-end starts its
// ranks itself: the parent forks N children BEFORE anything touches HIP (a process that has initialised the GPU must not
// fork) and only waits for them; the children share an anonymous mapping for the small host-side collectives of the
// set-up (gathering the processor-patch point lists, returnReduce of the mesh statistics, the syncPointList calls of the
// layer / boundary set-up, the ncclUniqueId).  Per-iteration data never goes through here on a multi-GPU node: that is
// RCCL on device buffers (smoothMesh_main.cpp).  A host-staged exchange through the same mapping exists as the debug
// transport for several ranks on ONE GPU (RCCL refuses two ranks on the same device).
#pragma once
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace smhost {

// what the front-end's signal handler needs: the ranks' pids and the blackboard's abort word (async-signal-safe access only)
struct NodeSignalState {
    pid_t pids[64];
    volatile sig_atomic_t nPids;
    std::atomic<uint32_t>* abortFlag;
    volatile sig_atomic_t got;
};
inline NodeSignalState& nodeSignalState() { static NodeSignalState st{}; return st; }
// SIGTERM / SIGINT / SIGHUP at the front-end (a scheduler, `timeout`, ^C): the ranks must not outlive it holding their GPUs
// and RCCL communicators.  Raise the abort word (ranks spinning in a barrier leave) and pass the signal on to every rank.
inline void nodeForwardSignal(int sig) {
    NodeSignalState& st = nodeSignalState();
    st.got = sig;
    if (st.abortFlag) st.abortFlag->store(1, std::memory_order_relaxed);
    for (int i = 0; i < st.nPids; ++i) ::kill(st.pids[i], sig == SIGINT ? SIGINT : SIGTERM);
}

struct NodeShm {
    std::atomic<uint32_t> arrived;     // barrier: arrivals of the current generation
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> abortFlag;   // set when a rank dies: everybody leaves
    uint32_t nRanks;
    uint64_t slotBytes;                // capacity of one rank's slot
    uint64_t used[64];                 // bytes a rank has published in its slot
};

class NodeComm {
public:
    int rank = 0, size = 1;

    // parent side: map the blackboard, fork `n` ranks; returns in the children with rank set; the parent never returns
    // (it waits, forwards the worst exit code and kills the survivors when a rank fails)
    void launch(int n, size_t slotBytes = (size_t)1 << 32) {
        if (n > 64) { std::fprintf(stderr, "smoothMesh: at most 64 ranks per node\n"); std::exit(1); }
        size = n;
        slot_ = slotBytes;
        const size_t total = headerBytes() + (size_t)n * slot_;
        void* p = ::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { std::perror("smoothMesh: mmap"); std::exit(1); }
        shm_ = new (p) NodeShm();
        shm_->arrived = 0; shm_->generation = 0; shm_->abortFlag = 0; shm_->nRanks = (uint32_t)n; shm_->slotBytes = slot_;
        std::fflush(stdout);
        std::vector<pid_t> pids;
        const pid_t parent = ::getpid();
        NodeSignalState& sg = nodeSignalState();
        sg.nPids = 0; sg.abortFlag = &shm_->abortFlag; sg.got = 0;
        for (int r = 0; r < n; ++r) {
            const pid_t pid = ::fork();
            if (pid < 0) { std::perror("smoothMesh: fork"); for (pid_t q : pids) ::kill(q, SIGTERM); std::exit(1); }
            if (pid == 0) {
                // a rank dies with the front-end, however that one is killed (kill -9 included); the re-check closes the
                // window in which the parent died between fork() and prctl()
                ::prctl(PR_SET_PDEATHSIG, SIGTERM);
                if (::getppid() != parent) std::_Exit(1);
                rank = r;
                return;
            }
            pids.push_back(pid);
            sg.pids[r] = pid;
            sg.nPids = r + 1;
        }
        {
            struct sigaction sa;
            std::memset(&sa, 0, sizeof(sa));
            sa.sa_handler = nodeForwardSignal;
            sigemptyset(&sa.sa_mask);
            for (int sig : {SIGTERM, SIGINT, SIGHUP}) ::sigaction(sig, &sa, nullptr);
        }
        int worst = 0, left = n;
        while (left > 0) {
            int st = 0;
            const pid_t pid = ::wait(&st);
            if (pid < 0) { if (errno == EINTR) continue; break; }   // (a forwarded signal interrupts the wait)
            --left;
            const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            if (code != 0) {
                if (worst == 0) worst = code;
                shm_->abortFlag = 1;                       // ranks waiting in a barrier leave
                for (pid_t q : pids) if (q != pid) ::kill(q, SIGTERM);
            }
        }
        if (worst == 0 && sg.got) worst = 128 + (int)sg.got;
        std::exit(worst);
    }

    bool master() const { return rank == 0; }

    void barrier() {
        if (size == 1) return;
        const uint32_t gen = shm_->generation.load(std::memory_order_acquire);
        if (shm_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)size) {
            shm_->arrived.store(0, std::memory_order_relaxed);
            shm_->generation.fetch_add(1, std::memory_order_release);
            return;
        }
        unsigned spins = 0;
        while (shm_->generation.load(std::memory_order_acquire) == gen) {
            if (shm_->abortFlag.load(std::memory_order_relaxed)) std::_Exit(1);
            if (++spins > 2000) ::sched_yield();
        }
    }

    char* slot(int r) const { return reinterpret_cast<char*>(shm_) + headerBytes() + (size_t)r * slot_; }
    size_t slotBytes() const { return slot_; }

    // every rank's blob, in rank order
    std::vector<std::vector<char>> allgatherv(const void* data, size_t bytes) {
        std::vector<std::vector<char>> out((size_t)size);
        if (size == 1) { out[0].assign((const char*)data, (const char*)data + bytes); return out; }
        if (bytes > slot_) { std::fprintf(stderr, "smoothMesh: node blackboard slot too small\n"); std::_Exit(1); }
        std::memcpy(slot(rank), data, bytes);
        shm_->used[rank] = bytes;
        barrier();
        for (int r = 0; r < size; ++r) out[(size_t)r].assign(slot(r), slot(r) + shm_->used[r]);
        barrier();
        return out;
    }
    template <typename T>
    std::vector<std::vector<T>> allgatherVec(const std::vector<T>& v) {
        const auto raw = allgatherv(v.data(), v.size() * sizeof(T));
        std::vector<std::vector<T>> out(raw.size());
        for (size_t r = 0; r < raw.size(); ++r) {
            out[r].resize(raw[r].size() / sizeof(T));
            std::memcpy(out[r].data(), raw[r].data(), raw[r].size());
        }
        return out;
    }
    template <typename T>
    std::vector<T> allgather(const T& v) {
        const auto raw = allgatherv(&v, sizeof(T));
        std::vector<T> out(raw.size());
        for (size_t r = 0; r < raw.size(); ++r) std::memcpy(&out[r], raw[r].data(), sizeof(T));
        return out;
    }
    double reduceMin(double v) { double m = v; for (double x : allgather(v)) m = x < m ? x : m; return m; }
    double reduceMax(double v) { double m = v; for (double x : allgather(v)) m = x > m ? x : m; return m; }
    long reduceSum(long v) { long s = 0; for (long x : allgather(v)) s += x; return s; }
    bool reduceOr(bool v) { bool s = false; for (int x : allgather<int>(v ? 1 : 0)) s = s || x; return s; }
    bool reduceAnd(bool v) { bool s = true; for (int x : allgather<int>(v ? 1 : 0)) s = s && x; return s; }
    template <typename T>
    T broadcast(const T& v, int root = 0) { return allgather(v)[(size_t)root]; }

private:
    static size_t headerBytes() { return (sizeof(NodeShm) + 4095) & ~(size_t)4095; }
    NodeShm* shm_ = nullptr;
    size_t slot_ = 0;
};

}  // namespace smhost
