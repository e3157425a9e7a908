// The rumination queue on the GPUs of one node from ONE process (include/rumi_queue.h): contiguous blocks of the time-ordered queue per device,
// one extractor handle per shard, ONE all-gather of fixed-capacity records.  RCCL is bound at run time (dlopen): the library has no link-time
// dependency on it, and inside a process that already carries RCCL (PyTorch) the loaded copy is the one used.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "rumi_common.h"
#include "rumi_queue.h"

namespace {

// the five RCCL entry points of the exchange (rccl.h: ncclResult_t = int, 0 = success; ncclComm_t = opaque pointer; ncclUint8 / ncclChar = 1 / 0)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    }
};
constexpr int kNcclUint8 = 1;

struct Shard {
    int device = 0;
    RumiOrb *orb = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *dOwn = nullptr;       // [per][rb] this shard's block
    uint8_t *dAll = nullptr;       // [n][per][rb] the gathered queue
    void *comm = nullptr;
    int rc = RUMI_OK;
    std::string err;
    float ms = 0;
};

}  // namespace

struct RumiQueue {
    RumiOrbConfig cfg{};
    int cap = 0, per = 0;
    int64_t rb = 0;
    std::vector<Shard> shards;
    bool rccl = false;
    Rccl api;
    float lastMs[4] = {0, 0, 0, 0};
};

using rumi::g_lastError;

extern "C" void rumi_queue_destroy(RumiQueue *q) {
    if (!q) return;
    for (Shard &s : q->shards) {
        (void)hipSetDevice(s.device);
        if (s.comm && q->api.CommDestroy) (void)q->api.CommDestroy(s.comm);
        if (s.orb) rumi_orb_destroy(s.orb);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.dOwn) (void)hipFree(s.dOwn);
        if (s.dAll) (void)hipFree(s.dAll);
    }
    delete q;
}

extern "C" int rumi_queue_create(const RumiOrbConfig *cfg, const int32_t *devices, int32_t n_devices, int32_t cap, RumiQueue **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (!cfg || !devices || n_devices < 1 || n_devices > 64 || cap < 1 || cfg->max_batch < 1) { g_lastError = "rumi_queue_create: bad argument"; return RUMI_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_lastError = "no HIP device visible: librumi_hip has no CPU fallback"; return RUMI_E_NO_DEVICE; }
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) { g_lastError = "rumi_queue_create: device ordinal out of range"; return RUMI_E_INVALID; }
        for (int j = 0; j < i; j++) distinct &= devices[i] != devices[j];
    }
    RumiQueue *q = new RumiQueue();
    q->cfg = *cfg; q->cap = cap; q->per = cfg->max_batch; q->rb = 8 + 60ll * cap;
    q->shards.resize((size_t)n_devices);
    const size_t blockBytes = (size_t)q->per * (size_t)q->rb;
    for (int i = 0; i < n_devices; i++) {
        Shard &s = q->shards[i];
        s.device = devices[i];
        if (hipSetDevice(s.device) != hipSuccess) { rumi_queue_destroy(q); return RUMI_E_NO_DEVICE; }
        RumiOrbConfig c = *cfg;
        c.device = s.device;
        int rc = rumi_orb_create(&c, &s.orb);
        if (rc != RUMI_OK) { rumi_queue_destroy(q); return rc; }
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&s.dOwn, blockBytes) != hipSuccess ||
            hipMalloc((void **)&s.dAll, blockBytes * n_devices) != hipSuccess || hipMemset(s.dOwn, 0, blockBytes) != hipSuccess) {
            g_lastError = "rumi_queue_create: device allocation failed";
            rumi_queue_destroy(q);
            return RUMI_E_NO_DEVICE;
        }
    }
    // the exchange: RCCL when every shard has a device of its own (one shard included: its all-gather is a copy, and it exercises the binding)
    static const bool noRccl = std::getenv("RUMI_QUEUE_NO_RCCL") != nullptr;
    if (distinct && !noRccl) {
        if (!q->api.load()) { g_lastError = "rumi_queue_create: librccl could not be loaded"; rumi_queue_destroy(q); return RUMI_E_NO_DEVICE; }
        std::vector<void *> comms((size_t)n_devices, nullptr);
        std::vector<int> devs(devices, devices + n_devices);
        const int r = q->api.CommInitAll(comms.data(), n_devices, devs.data());
        if (r != 0) {
            g_lastError = std::string("rumi_queue_create: ncclCommInitAll failed: ") + (q->api.GetErrorString ? q->api.GetErrorString(r) : "?");
            rumi_queue_destroy(q);
            return RUMI_E_NO_DEVICE;
        }
        for (int i = 0; i < n_devices; i++) q->shards[i].comm = comms[i];
        q->rccl = true;
    }
    *out = q;
    return RUMI_OK;
}

extern "C" int32_t rumi_queue_shards(const RumiQueue *q) { return q ? (int32_t)q->shards.size() : 0; }
extern "C" int64_t rumi_queue_record_bytes(const RumiQueue *q) { return q ? q->rb : 0; }
extern "C" int32_t rumi_queue_block_capacity(const RumiQueue *q) { return q ? q->per : 0; }
extern "C" int32_t rumi_queue_uses_rccl(const RumiQueue *q) { return q && q->rccl ? 1 : 0; }

static inline int block_begin(int F, int g, int n) { return (int)(((long long)g * F) / n); }

extern "C" int32_t rumi_queue_row(const RumiQueue *q, int32_t n_frames, int32_t frame) {
    if (!q || n_frames < 1 || frame < 0 || frame >= n_frames) return -1;
    const int n = (int)q->shards.size();
    for (int g = 0; g < n; g++)
        if (frame < block_begin(n_frames, g + 1, n)) return g * q->per + (frame - block_begin(n_frames, g, n));
    return -1;
}

extern "C" int rumi_queue_last_ms(const RumiQueue *q, float ms[4]) {
    if (!q || !ms) return RUMI_E_INVALID;
    for (int i = 0; i < 4; i++) ms[i] = q->lastMs[i];
    return RUMI_OK;
}

extern "C" int rumi_queue_extract(RumiQueue *q, const uint8_t *const *imgs, int32_t n_frames, int32_t w, int32_t hgt, int32_t stride, int32_t lap0,
                                  int32_t lap1, void **d_gathered, uint8_t *h_records) {
    if (!q || !imgs || n_frames < 1) { g_lastError = "rumi_queue_extract: bad argument"; return RUMI_E_INVALID; }
    const int n = (int)q->shards.size();
    for (int g = 0; g < n; g++)
        if (block_begin(n_frames, g + 1, n) - block_begin(n_frames, g, n) > q->per) { g_lastError = "rumi_queue_extract: a shard's block exceeds cfg.max_batch"; return RUMI_E_CAPACITY; }
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    const size_t blockBytes = (size_t)q->per * (size_t)q->rb;
    // ---- extraction: every shard its block, one feeding thread per shard (the extractor's host entry overlaps its transfers with its kernels)
    auto work = [&](int g) {
        Shard &s = q->shards[g];
        const double ts = now();
        const int b0 = block_begin(n_frames, g, n), nb = block_begin(n_frames, g + 1, n) - b0;
        s.rc = RUMI_OK;
        if (hipSetDevice(s.device) != hipSuccess) { s.rc = RUMI_E_NO_DEVICE; s.err = "hipSetDevice failed"; return; }
        // rows past the block are empty records (n = 0): only their 8-byte heads have to be cleared
        if (nb < q->per && hipMemset2DAsync(s.dOwn + (size_t)nb * q->rb, (size_t)q->rb, 0, 8, (size_t)(q->per - nb), s.stream) != hipSuccess) { s.rc = RUMI_E_NO_DEVICE; s.err = "memset failed"; return; }
        if (nb > 0) {
            // the shard's records go back to the caller FROM THE SHARD'S OWN DEVICE (every device's link carries its share; a pinned destination
            // receives them sub-chunk by sub-chunk under the kernels): nothing is copied back after the exchange
            s.rc = rumi_orb_extract_batch_host_records(s.orb, imgs + b0, nb, w, hgt, stride, lap0, lap1, s.dOwn, q->rb, q->cap,
                                                       h_records ? h_records + (size_t)b0 * q->rb : nullptr, s.stream);
            if (s.rc != RUMI_OK) s.err = rumi_last_error();
        }
        if (s.rc == RUMI_OK && hipStreamSynchronize(s.stream) != hipSuccess) { s.rc = RUMI_E_NO_DEVICE; s.err = "stream synchronisation failed"; }
        s.ms = (float)(now() - ts);
    };
    {
        std::vector<std::thread> th;
        for (int g = 1; g < n; g++) th.emplace_back(work, g);
        work(0);
        for (auto &t : th) t.join();
    }
    float exMs = 0;
    for (Shard &s : q->shards) {
        exMs = std::max(exMs, s.ms);
        if (s.rc != RUMI_OK) { g_lastError = "rumi_queue_extract: shard on device " + std::to_string(s.device) + ": " + s.err; return s.rc; }
    }
    const double t1 = now();
    // ---- THE exchange step
    if (q->rccl) {
        int r = q->api.GroupStart();
        for (int g = 0; g < n && r == 0; g++) {
            Shard &s = q->shards[g];
            HIP_TRY(hipSetDevice(s.device));
            r = q->api.AllGather(s.dOwn, s.dAll, blockBytes, kNcclUint8, s.comm, s.stream);
        }
        const int r2 = q->api.GroupEnd();
        if (r != 0 || r2 != 0) {
            g_lastError = std::string("rumi_queue_extract: ncclAllGather failed: ") + (q->api.GetErrorString ? q->api.GetErrorString(r != 0 ? r : r2) : "?");
            return RUMI_E_NO_DEVICE;
        }
    } else {
        // logical shards on shared devices: every shard copies every block (what the all-gather would deliver)
        for (int g = 0; g < n; g++) {
            Shard &s = q->shards[g];
            HIP_TRY(hipSetDevice(s.device));
            for (int j = 0; j < n; j++)
                HIP_TRY(hipMemcpyAsync(s.dAll + (size_t)j * blockBytes, q->shards[j].dOwn, blockBytes, hipMemcpyDeviceToDevice, s.stream));
        }
    }
    for (Shard &s : q->shards) { HIP_TRY(hipSetDevice(s.device)); HIP_TRY(hipStreamSynchronize(s.stream)); }
    const double t2 = now();
    const double t3 = now();                                // (the records reached h_records during the extraction: see work())
    if (d_gathered) for (int g = 0; g < n; g++) d_gathered[g] = q->shards[g].dAll;
    q->lastMs[0] = exMs; q->lastMs[1] = (float)(t2 - t1); q->lastMs[2] = (float)(t3 - t2); q->lastMs[3] = (float)(t3 - t0);
    return RUMI_OK;
}
