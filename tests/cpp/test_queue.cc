// The rumination queue through include/rumi_queue.h alone, as a C++ host of the reference would call it (one process, host frames in):
//   test_queue <F> <w> <h> <frames.bin>   -- F frames of w x h bytes back to back
// For 1, 2 and 3 logical shards aliased to device 0 (device-to-device copies stand in for the collective when ordinals repeat; ONE shard with its
// own device goes through RCCL's ncclAllGather: the binding is exercised on a one-GPU box) the gathered records must equal, byte for byte in
// their live part, what ONE rumi_orb_extract_batch_host_records call over the whole queue gives.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rumi_queue.h"

#define CHECK(x) do { const int rc_ = (x); if (rc_ != RUMI_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc_, rumi_last_error()); return 1; } } while (0)

extern "C" int hipMalloc(void **, size_t);
extern "C" int hipFree(void *);
extern "C" int hipMemcpy(void *, const void *, size_t, int);

static bool same_record(const uint8_t *a, const uint8_t *b, int cap) {
    int32_t na, nb;
    std::memcpy(&na, a, 4); std::memcpy(&nb, b, 4);
    if (std::memcmp(a, b, 8) != 0 || na != nb || na < 0 || na > cap) return false;
    return std::memcmp(a + 8, b + 8, (size_t)na * 28) == 0 && std::memcmp(a + 8 + (size_t)cap * 28, b + 8 + (size_t)cap * 28, (size_t)na * 32) == 0;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const int F = std::atoi(argv[1]), w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> buf((size_t)F * w * h);
    FILE *f = std::fopen(argv[4], "rb");
    if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::printf("FAIL reading frames\n"); return 2; }
    std::fclose(f);
    std::vector<const uint8_t *> imgs((size_t)F);
    for (int i = 0; i < F; i++) imgs[i] = buf.data() + (size_t)i * w * h;
    const int cap = 1096;
    const int64_t rb = 8 + 60ll * cap;
    // reference: one call over the whole queue
    RumiOrbConfig cfg{1000, 1.2f, 8, 20, 7, w, h, F, 0, 0, 0};
    RumiOrb *orb = nullptr;
    CHECK(rumi_orb_create(&cfg, &orb));
    void *dRef = nullptr;
    if (hipMalloc(&dRef, (size_t)F * rb) != 0) { std::printf("FAIL hipMalloc\n"); return 1; }
    std::vector<uint8_t> ref((size_t)F * rb);
    CHECK(rumi_orb_extract_batch_host_records(orb, imgs.data(), F, w, h, w, 0, 1000, dRef, rb, cap, ref.data(), nullptr));
    rumi_orb_destroy(orb);
    (void)hipFree(dRef);
    for (int shards = 1; shards <= 3; shards++) {
        for (int lenCase = 0; lenCase < 2; lenCase++) {
            const int Fq = lenCase == 0 ? F : F - 1;                  // an uneven split too
            std::vector<int32_t> dev((size_t)shards, 0);
            RumiOrbConfig qc = cfg;
            qc.max_batch = (F + shards - 1) / shards;
            RumiQueue *q = nullptr;
            CHECK(rumi_queue_create(&qc, dev.data(), shards, cap, &q));
            if (rumi_queue_record_bytes(q) != rb || rumi_queue_shards(q) != shards) { std::printf("FAIL queue geometry\n"); return 1; }
            const int expectRccl = shards == 1 ? 1 : 0;
            if (rumi_queue_uses_rccl(q) != expectRccl) { std::printf("FAIL exchange kind: shards %d uses_rccl %d\n", shards, rumi_queue_uses_rccl(q)); return 1; }
            std::vector<void *> dg((size_t)shards, nullptr);
            std::vector<uint8_t> got((size_t)Fq * rb);
            for (int rep = 0; rep < 2; rep++) {                       // twice: the second call reuses every buffer
                CHECK(rumi_queue_extract(q, imgs.data(), Fq, w, h, w, 0, 1000, dg.data(), got.data()));
                for (int i = 0; i < Fq; i++)
                    if (!same_record(got.data() + (size_t)i * rb, ref.data() + (size_t)i * rb, cap)) { std::printf("FAIL shards %d frames %d: host record %d differs\n", shards, Fq, i); return 1; }
                // every shard's device copy of the gathered queue, through the row map
                const int per = rumi_queue_block_capacity(q);
                std::vector<uint8_t> all((size_t)shards * per * rb);
                for (int g = 0; g < shards; g++) {
                    if (hipMemcpy(all.data(), dg[g], all.size(), 2 /* hipMemcpyDeviceToHost */) != 0) { std::printf("FAIL hipMemcpy\n"); return 1; }
                    for (int i = 0; i < Fq; i++) {
                        const int row = rumi_queue_row(q, Fq, i);
                        if (row < 0 || !same_record(all.data() + (size_t)row * rb, ref.data() + (size_t)i * rb, cap)) { std::printf("FAIL shards %d shard %d frame %d (row %d)\n", shards, g, i, row); return 1; }
                    }
                    int live = 0;
                    for (int r = 0; r < shards * per; r++) { int32_t n; std::memcpy(&n, all.data() + (size_t)r * rb, 4); live += n > 0; }
                    if (live != Fq) { std::printf("FAIL shards %d: %d live rows in the gathered layout, expected %d\n", shards, live, Fq); return 1; }
                }
            }
            float ms[4];
            CHECK(rumi_queue_last_ms(q, ms));
            std::printf("shards %d (exchange: %s) frames %d: extraction %.2f ms, exchange %.3f ms, total %.2f ms\n", shards, expectRccl ? "RCCL all-gather" : "device-to-device copies",
                        Fq, ms[0], ms[1], ms[3]);
            rumi_queue_destroy(q);
        }
    }
    std::printf("queue OK\n");
    return 0;
}
