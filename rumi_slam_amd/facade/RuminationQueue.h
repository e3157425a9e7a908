// Drop-in helper for the rumination queue on the GPUs of one node (include/rumi_queue.h), in the reference's own types.
//
// Where it goes in the reference (R/ = /root/reference/src/rumi-slam/): CloudImageSampler collects and time-sorts the images tracking could not use
// (mvCurrentCloudProcessImages, R/lib_src/CloudImageSampler.cc:148-170); KFDSample (R/lib_src/KFDSample.cc:113,153) runs
// `mpORBextractor->operator()(im, cv::Mat(), mvKeys, mDescriptors, vLapping)` on them one image per call.  With this class the whole vector goes out in one call:
//
//     static ORB_SLAM3::RuminationQueue queue(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST, {0, 1, 2, 3, 4, 5, 6, 7}, /*maxBlock*/ 128, 640, 480);
//     std::vector<std::vector<cv::KeyPoint>> keys; std::vector<cv::Mat> descs;
//     queue.Extract(vImages, keys, descs, {0, 1000});      // keys[i] / descs[i] are what operator() gives for vImages[i]
//
// and every device of the node holds the gathered records afterwards (GatheredDevicePointer(g)): any frame pair can be matched where it is
// (rumi_match_bruteforce_batch_device_strided reads descriptors in place, record stride = RecordBytes()).
#pragma once
#include <cstring>
#include <vector>

#include "cv_shim.h"
#include "rumi_queue.h"
#include "rumi_status.h"

namespace ORB_SLAM3 {

class RuminationQueue {
public:
    RuminationQueue(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, const std::vector<int> &devices, int maxBlock, int maxWidth,
                    int maxHeight)
        : mCap(nfeatures + 96) {
        RumiOrbConfig cfg{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxWidth, maxHeight, maxBlock, -1, 0, 0};
        std::vector<int32_t> dev(devices.begin(), devices.end());
        const int rc = rumi_queue_create(&cfg, dev.data(), (int32_t)dev.size(), mCap, &mQ);
        if (rc != RUMI_OK) { rumi_facade::report("RuminationQueue", rc); mQ = nullptr; }
    }
    ~RuminationQueue() { if (mQ) rumi_queue_destroy(mQ); }
    RuminationQueue(const RuminationQueue &) = delete;
    RuminationQueue &operator=(const RuminationQueue &) = delete;

    // images: 8-bit grey, one size, time order.  Returns the number of frames extracted, -1 on failure (reported through the status hook).
    int Extract(const std::vector<cv::Mat> &images, std::vector<std::vector<cv::KeyPoint>> &keys, std::vector<cv::Mat> &descriptors, const std::vector<int> &vLappingArea) {
        keys.clear(); descriptors.clear();
        if (!mQ || images.empty()) return images.empty() ? 0 : -1;
        const int F = (int)images.size(), w = images[0].cols, h = images[0].rows;
        std::vector<const uint8_t *> ptr((size_t)F);
        for (int i = 0; i < F; i++) {
            if (images[i].cols != w || images[i].rows != h || images[i].step != images[0].step) { rumi_facade::report("RuminationQueue::Extract", RUMI_E_INVALID, "frames of one size and pitch are required"); return -1; }
            ptr[i] = images[i].data;
        }
        const int64_t rb = rumi_queue_record_bytes(mQ);
        mHost.resize((size_t)F * (size_t)rb);
        mGathered.assign((size_t)rumi_queue_shards(mQ), nullptr);
        const int rc = rumi_queue_extract(mQ, ptr.data(), F, w, h, (int)images[0].step, vLappingArea.size() > 0 ? vLappingArea[0] : 0, vLappingArea.size() > 1 ? vLappingArea[1] : 0,
                                          mGathered.data(), mHost.data());
        if (rc != RUMI_OK) { rumi_facade::report("RuminationQueue::Extract", rc); return -1; }
        keys.resize((size_t)F); descriptors.resize((size_t)F);
        for (int i = 0; i < F; i++) {
            const uint8_t *r = mHost.data() + (size_t)i * rb;
            int32_t n; std::memcpy(&n, r, 4);
            keys[i].resize((size_t)n);
            static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint is the 28-byte POD RumiKeyPoint mirrors");
            if (n > 0) std::memcpy((void *)keys[i].data(), r + 8, (size_t)n * 28);
            descriptors[i].create(n, 32, CV_8U);
            if (n > 0) std::memcpy(descriptors[i].data, r + 8 + (size_t)mCap * 28, (size_t)n * 32);
        }
        return F;
    }
    void *GatheredDevicePointer(int shard) const { return shard >= 0 && shard < (int)mGathered.size() ? mGathered[shard] : nullptr; }
    long long RecordBytes() const { return mQ ? rumi_queue_record_bytes(mQ) : 0; }
    int Row(int nFrames, int frame) const { return mQ ? rumi_queue_row(mQ, nFrames, frame) : -1; }

private:
    RumiQueue *mQ = nullptr;
    int mCap;
    std::vector<uint8_t> mHost;
    std::vector<void *> mGathered;
};

}  // namespace ORB_SLAM3
