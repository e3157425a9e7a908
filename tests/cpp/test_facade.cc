// GPU test of the C++ facade classes (reference signatures) against the CPU oracle, with a mock data model that has the
// reference's member names.  Built and run by tests/test_facade_gpu.py:  g++ ... librumi_hip.so liboracle.so
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <tuple>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "rumi_status.h"
#include "Optimizer.h"
#include "FrameFrustum.h"
#include "TrackingStep.h"
#include "RuminationQueue.h"
#include "orb_oracle.h"

extern "C" {
int orc_search_by_projection_frame(const orc::KeyPoint *, const uint8_t *, int, float, float, float, float, const float *, const float *,
                                   const float *, const orc::KeyPoint *, int, const int32_t *, const uint8_t *, const float *, const uint8_t *,
                                   const int32_t *, float, int, int32_t *);
int orc_pose_optimization(const float *, const float *, const float *, int, const float *, float *, uint8_t *);
int orc_local_ba(int, float *, const uint8_t *, int, float *, int, const int32_t *, const int32_t *, const float *, const float *, const float *,
                 const volatile uint8_t *, uint8_t *);
int orc_merge_ba(int, float *, const uint8_t *, int, float *, int, const int32_t *, const int32_t *, const float *, const float *, const float *,
                 const volatile uint8_t *, uint8_t *, int32_t *);
}

// ---- mock data model: only the members the facades touch, named as in the reference ----
struct V3f { float v[3]; float operator()(int i) const { return v[i]; } };
struct Q4f { float q[4]; float x() const { return q[0]; } float y() const { return q[1]; } float z() const { return q[2]; } float w() const { return q[3]; } };
struct SE3f { float T[7]; Q4f unit_quaternion() const { return Q4f{{T[0], T[1], T[2], T[3]}}; } V3f translation() const { return V3f{{T[4], T[5], T[6]}}; } };
struct Map { std::mutex mMutexMapUpdate; long GetInitKFid() { return 0; } int changes = 0; void IncreaseChangeIndex() { changes++; } };
struct KeyFrame;
struct MapPoint {
    static std::mutex mGlobalMutex;
    V3f pos; cv::Mat desc; int nObs = 1; bool bad = false; Map *map = nullptr; long mnBALocalForKF = -1, mnBALocalForMerge = -1;
    std::map<KeyFrame *, std::tuple<int, int>> obs;
    V3f GetWorldPos() { return pos; }
    cv::Mat GetDescriptor() { return desc; }
    int Observations() { return nObs; }
    bool isBad() { return bad; }
    Map *GetMap() { return map; }
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { return obs; }
    void EraseObservation(KeyFrame *k) { obs.erase(k); }
    void SetWorldPosXYZ(float x, float y, float z) { pos = V3f{{x, y, z}}; }
    float minD = 0.5f, maxD = 60.f;
    float GetMinDistance() { return minD; }
    float GetMaxDistance() { return maxD; }
    void UpdateNormalAndDepth() {}
    V3f normal{{0, 0, 1}};
    V3f GetNormal() { return normal; }
    bool IsInKeyFrame(KeyFrame *k) { return obs.count(k) > 0; }
    void AddObservation(KeyFrame *k, int idx) { if (!obs.count(k)) nObs++; obs[k] = std::make_tuple(idx, -1); }
    void Replace(MapPoint *other);                     // defined after KeyFrame
    bool mbTrackInView = false, mbTrackInViewR = false; float mTrackProjX = 0, mTrackProjY = 0, mTrackViewCos = 1, mTrackDepth = 1; int mnTrackScaleLevel = 0;
    long mnLastFrameSeen = -1; int nVisible = 0, nFound = 0;
    void IncreaseVisible() { nVisible++; }
    void IncreaseFound() { nFound++; }
};
std::mutex MapPoint::mGlobalMutex;
struct Frame {
    int N = 0; long mnId = 7;
    std::vector<cv::KeyPoint> mvKeysUn; cv::Mat mDescriptors; std::vector<MapPoint *> mvpMapPoints; std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2; std::vector<float> mvuRight;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 640, mnMaxY = 480, fx = 535.4f, fy = 539.2f, cx = 320.1f, cy = 247.6f;
    SE3f pose;
    float mfLogScaleFactor = 0.1823216f; int mnScaleLevels = 8;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    SE3f GetPose() const { return pose; }
    void SetPoseFromQuatTrans(const float *T7) { std::memcpy(pose.T, T7, 28); }
    void PoseMatrices(float *R, float *t, float *Ow) const {          // Frame::UpdatePoseMatrices in Eigen's / Sophus' float arithmetic (as the device does it)
        const float x = pose.T[0], y = pose.T[1], z = pose.T[2], w = pose.T[3];
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        const float r[9] = {1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy)};
        for (int i = 0; i < 9; i++) R[i] = r[i];
        for (int i = 0; i < 3; i++) t[i] = pose.T[4 + i];
        const float qx = -x, qy = -y, qz = -z, v0 = t[0] * -1.f, v1 = t[1] * -1.f, v2 = t[2] * -1.f;     // Ow = conj(q) * (-tcw)
        float u0 = qy * v2 - qz * v1, u1 = qz * v0 - qx * v2, u2 = qx * v1 - qy * v0;
        u0 += u0; u1 += u1; u2 += u2;
        const float c0 = qy * u2 - qz * u1, c1 = qz * u0 - qx * u2, c2 = qx * u1 - qy * u0;
        Ow[0] = (v0 + w * u0) + c0; Ow[1] = (v1 + w * u1) + c1; Ow[2] = (v2 + w * u2) + c2;
    }
};
struct KeyFrame : Frame {
    long mnId = 0, mnBALocalForKF = -1, mnBAFixedForKF = -1, mnBALocalForMerge = -1; Map *map = nullptr; bool bad = false;
    std::vector<KeyFrame *> covis;
    std::set<MapPoint *> GetMapPoints() { std::set<MapPoint *> s; for (auto *p : mvpMapPoints) if (p) s.insert(p); return s; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return covis; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    bool isBad() { return bad; }
    Map *GetMap() { return map; }
    void EraseMapPointMatch(MapPoint *p) { for (auto &q : mvpMapPoints) if (q == p) q = nullptr; }
    MapPoint *GetMapPoint(size_t idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *p, size_t idx) { mvpMapPoints[idx] = p; }
    // stands in for the Sophus/Eigen expressions of the real tree: a fixed rank-2 F12 (pure x-translation) and a far epipole
    void EpipolarGeometryTo(const KeyFrame *, float *F, float *e) const {
        const float f[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        for (int i = 0; i < 9; i++) F[i] = f[i];
        e[0] = 5000.f; e[1] = 240.f;
    }
};

// the part of MapPoint::Replace the Fuse loop can observe: the replaced point turns bad and hands over its key-frame slots
void MapPoint::Replace(MapPoint *other) {
    if (other == this) return;
    bad = true;
    for (auto &o : obs) {
        KeyFrame *k = o.first;
        if (!other->IsInKeyFrame(k)) { k->mvpMapPoints[std::get<0>(o.second)] = other; other->AddObservation(k, std::get<0>(o.second)); }
        else k->mvpMapPoints[std::get<0>(o.second)] = nullptr;
    }
    obs.clear(); nObs = 0;
}

static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { std::printf("FAIL: %s (%s:%d)\n", msg, __FILE__, __LINE__); fails++; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { std::printf("usage: test_facade frame0.bin frame1.bin (640x480 u8)\n"); return 2; }
    std::vector<uint8_t> im[2];
    for (int k = 0; k < 2; k++) {
        im[k].resize(640 * 480);
        FILE *f = std::fopen(argv[1 + k], "rb");
        if (!f || std::fread(im[k].data(), 1, im[k].size(), f) != im[k].size()) { std::printf("cannot read %s\n", argv[1 + k]); return 2; }
        std::fclose(f);
    }
    // ---- ORBextractor facade == oracle, bit for bit ----
    ORB_SLAM3::ORBextractor ext(1000, 1.2f, 8, 20, 7);
    ext.keepPyramid = true;
    orc::OrbExtractor ref(1000, 1.2f, 8, 20, 7);
    std::vector<int> lap = {0, 1000};
    Frame fr[2];
    for (int k = 0; k < 2; k++) {
        cv::Mat image(480, 640, CV_8U, im[k].data(), 640), mask;
        const int mono = ext(image, mask, fr[k].mvKeysUn, fr[k].mDescriptors, lap);
        std::vector<orc::KeyPoint> rk; std::vector<uint8_t> rd;
        const int rmono = ref.extract(im[k].data(), 640, 480, 640, 0, 1000, rk, rd);
        CHECK(mono == rmono && fr[k].mvKeysUn.size() == rk.size(), "extractor count / monoIndex");
        CHECK(std::memcmp(fr[k].mvKeysUn.data(), rk.data(), rk.size() * 28) == 0, "extractor key-points");
        bool dsame = true;
        for (size_t i = 0; i < rk.size(); i++) dsame &= std::memcmp(fr[k].mDescriptors.ptr((int)i), &rd[i * 32], 32) == 0;
        CHECK(dsame, "extractor descriptors");
        fr[k].N = (int)rk.size();
        fr[k].mvScaleFactors = ext.GetScaleFactors(); fr[k].mvInvLevelSigma2 = ext.GetInverseScaleSigmaSquares();
        fr[k].mvpMapPoints.assign(fr[k].N, nullptr); fr[k].mvbOutlier.assign(fr[k].N, false); fr[k].mvuRight.assign(fr[k].N, -1.f);
    }
    CHECK(ext.GetLevels() == 8 && std::fabs(ext.GetScaleFactor() - 1.2f) < 1e-6, "getters");
    CHECK(ext.mvImagePyramid[1].cols == 533 && ext.mvImagePyramid[1].rows == 400, "mvImagePyramid level size");
    CHECK(ext.mvImagePyramid[0].ptr(0)[-19 * (int)ext.mvImagePyramid[0].step - 19] == im[1][19 * 640 + 19], "REFLECT_101 border of level 0");
    cv::Mat empty, mask;
    std::vector<cv::KeyPoint> k0; cv::Mat d0;
    CHECK(ext(empty, mask, k0, d0, lap) == -1, "empty image returns -1");
    {   // ---- the rumination queue helper: three frames over two logical shards of device 0 == operator() frame by frame ----
        ORB_SLAM3::RuminationQueue queue(1000, 1.2f, 8, 20, 7, {0, 0}, 2, 640, 480);
        std::vector<cv::Mat> vImages = {cv::Mat(480, 640, CV_8U, im[0].data(), 640), cv::Mat(480, 640, CV_8U, im[1].data(), 640), cv::Mat(480, 640, CV_8U, im[0].data(), 640)};
        std::vector<std::vector<cv::KeyPoint>> qk; std::vector<cv::Mat> qd;
        CHECK(queue.Extract(vImages, qk, qd, lap) == 3 && qk.size() == 3, "RuminationQueue::Extract");
        bool qsame = qk.size() == 3;
        for (int i = 0; qsame && i < 3; i++) {
            const Frame &f = fr[i == 1 ? 1 : 0];
            qsame = qk[i].size() == f.mvKeysUn.size() && std::memcmp(qk[i].data(), f.mvKeysUn.data(), f.mvKeysUn.size() * 28) == 0 && qd[i].rows == f.mDescriptors.rows;
            for (int r = 0; qsame && r < qd[i].rows; r++) qsame = std::memcmp(qd[i].ptr(r), f.mDescriptors.ptr(r), 32) == 0;
        }
        CHECK(qsame, "RuminationQueue: key-points and descriptors of every queued frame = ORBextractor::operator()");
        CHECK(queue.GatheredDevicePointer(0) && queue.GatheredDevicePointer(1) && queue.Row(3, 2) == 3 && queue.RecordBytes() == 8 + 60 * 1096, "RuminationQueue: gathered layout");
    }

    // ---- SearchByProjection(Cur, Last) facade == oracle ----
    std::mt19937 rng(5);
    std::vector<MapPoint> mps(fr[0].N);
    std::vector<float> pos((size_t)fr[0].N * 3);
    std::vector<int32_t> lastMp(fr[0].N), obs(fr[0].N);
    std::vector<uint8_t> outl(fr[0].N, 0), mpDesc((size_t)fr[0].N * 32);
    const float T7[7] = {0.004f, -0.003f, 0.002f, 0.99998f, 0.02f, -0.01f, 0.03f};
    std::memcpy(fr[1].pose.T, T7, 28);
    for (int i = 0; i < fr[0].N; i++) {
        const float z = 2.f + (rng() % 600) / 100.f;
        const float u = fr[0].mvKeysUn[i].pt.x + 3.f, v = fr[0].mvKeysUn[i].pt.y - 2.f;
        mps[i].pos = V3f{{(u - 320.1f) / 535.4f * z, (v - 247.6f) / 539.2f * z, z}};
        mps[i].desc = cv::Mat(1, 32, CV_8U, fr[0].mDescriptors.ptr(i), 32);
        mps[i].nObs = (rng() % 20) ? 2 : 0;
        std::memcpy(&pos[3 * i], mps[i].pos.v, 12); std::memcpy(&mpDesc[(size_t)i * 32], fr[0].mDescriptors.ptr(i), 32);
        const bool has = rng() % 10 != 0;
        fr[0].mvpMapPoints[i] = has ? &mps[i] : nullptr;
        lastMp[i] = has ? i : -1; obs[i] = mps[i].nObs;
        outl[i] = rng() % 25 == 0; fr[0].mvbOutlier[i] = outl[i];
    }
    std::vector<int32_t> curRef(fr[1].N, -1);
    std::vector<uint8_t> curDesc((size_t)fr[1].N * 32);
    for (int i = 0; i < fr[1].N; i++) std::memcpy(&curDesc[(size_t)i * 32], fr[1].mDescriptors.ptr(i), 32);
    const float K4[4] = {535.4f, 539.2f, 320.1f, 247.6f};
    const int nref = orc_search_by_projection_frame((const orc::KeyPoint *)fr[1].mvKeysUn.data(), curDesc.data(), fr[1].N, 0, 0, 640, 480,
                                                    fr[1].mvScaleFactors.data(), T7, K4, (const orc::KeyPoint *)fr[0].mvKeysUn.data(), fr[0].N,
                                                    lastMp.data(), outl.data(), pos.data(), mpDesc.data(), obs.data(), 15.f, 1, curRef.data());
    ORB_SLAM3::ORBmatcher matcher(0.9f, true);
    const int ngpu = matcher.SearchByProjection(fr[1], fr[0], 15.f, true);
    CHECK(ngpu == nref && nref > 50, "SearchByProjection(Cur, Last) count");
    bool msame = true;
    for (int i = 0; i < fr[1].N; i++) msame &= (fr[1].mvpMapPoints[i] ? (int)(fr[1].mvpMapPoints[i] - mps.data()) : -1) == curRef[i];
    CHECK(msame, "SearchByProjection(Cur, Last) assignments");

    // ---- the other overloads instantiate and run (their C-ABI parity is covered by tests/test_matcher_gpu.py) ----
    {
        KeyFrame ka, kb;
        static_cast<Frame &>(ka) = fr[0]; static_cast<Frame &>(kb) = fr[1];
        for (int i = 0; i < ka.N; i++) ka.mFeatVec[(unsigned)(i % 97)].push_back((unsigned)i);
        for (int i = 0; i < kb.N; i++) kb.mFeatVec[(unsigned)(i % 97)].push_back((unsigned)i);
        std::vector<MapPoint *> m12;
        const int nb = matcher.SearchByBoW(&ka, &kb, m12);
        CHECK(nb >= 0 && (int)m12.size() == ka.N, "SearchByBoW(KF,KF) runs");
        // the relocalisation walk as one call: three candidates (one of them bad) against the frame == the single calls
        {
            Frame fq = fr[1];
            fq.mFeatVec = kb.mFeatVec;
            KeyFrame bad = ka; bad.bad = true;
            std::vector<KeyFrame *> cands = {&ka, &bad, &ka};
            std::vector<std::vector<MapPoint *>> vv;
            const std::vector<int> nn = matcher.SearchByBoW(cands, fq, vv);
            std::vector<MapPoint *> one;
            const int n1 = matcher.SearchByBoW(&ka, fq, one);
            CHECK(nn.size() == 3 && nn[0] == n1 && nn[2] == n1 && nn[1] == -1 && vv[0] == one && vv[2] == one && vv[1].empty(), "SearchByBoW over candidate key-frames == single calls");
        }
        Frame cur = fr[1];
        cur.mvpMapPoints.assign(cur.N, nullptr);
        std::set<MapPoint *> found;
        const int nr = matcher.SearchByProjection(cur, &ka, found, 10.f, 100);
        CHECK(nr >= 0, "SearchByProjection(Frame, KF, found, th, ORBdist) runs");
        std::vector<std::pair<size_t, size_t>> pairs;
        for (auto &p : ka.mvpMapPoints) p = nullptr;
        for (auto &p : kb.mvpMapPoints) p = nullptr;
        const int nt = matcher.SearchForTriangulation(&ka, &kb, pairs, false, true);
        CHECK(nt >= 0 && (int)pairs.size() == nt, "SearchForTriangulation pairs == count");
        // Fuse(KF, points, th): empty slots get an observation, occupied slots trigger Replace; replay keeps the map consistent
        {
            KeyFrame kf; static_cast<Frame &>(kf) = fr[1];
            std::memcpy(kf.pose.T, T7, 28);
            kf.mvpMapPoints.assign(kf.N, nullptr);
            std::vector<MapPoint> resident(kf.N);
            for (int i = 0; i < kf.N; i += 2) { resident[i].pos = V3f{{0, 0, 5}}; resident[i].nObs = 1 + i % 5; resident[i].obs[&kf] = std::make_tuple(i, -1); kf.mvpMapPoints[i] = &resident[i]; }
            std::vector<MapPoint> cand(mps.size());
            std::vector<MapPoint *> vp;
            for (size_t i = 0; i < mps.size(); i++) { cand[i].pos = mps[i].pos; cand[i].desc = mps[i].desc; cand[i].nObs = 3; cand[i].obs.clear(); vp.push_back(i % 17 == 0 ? nullptr : &cand[i]); }
            // error policy (facade/rumi_status.h): the second-camera branch is not built -> reported through the installed handler, 0 fused, nothing touched
            {
                static int hooked = 0; static int hookedStatus = 0;
                rumi_facade::set_error_handler([](const char *, int status, const char *) { hooked++; hookedStatus = status; });
                rumi_facade::clear_status();
                const int nr = matcher.Fuse(&kf, vp, 15.f, true);
                CHECK(nr == 0 && hooked == 1 && hookedStatus == RUMI_E_INVALID && rumi_facade::last_status() == RUMI_E_INVALID, "Fuse(bRight = true) is reported, not silently run as mono");
                bool untouched = true;
                for (auto &c : cand) untouched &= !c.bad && c.obs.empty();
                CHECK(untouched, "Fuse(bRight = true) leaves the map alone");
                rumi_facade::set_error_handler(nullptr);
                rumi_facade::clear_status();
            }
            const int nf = matcher.Fuse(&kf, vp, 15.f);
            int added = 0, replacedCand = 0, replacedResident = 0;
            for (auto &c : cand) { if (c.bad) replacedCand++; else if (c.IsInKeyFrame(&kf)) added++; }
            for (auto &r : resident) if (r.bad) replacedResident++;
            bool consistent = true;
            for (int i = 0; i < kf.N; i++) if (kf.mvpMapPoints[i] && kf.mvpMapPoints[i]->bad) consistent = false;
            CHECK(nf > 20 && replacedCand + replacedResident > 0 && added > 0, "Fuse adds observations and replaces duplicates");
            CHECK(consistent, "Fuse leaves no bad point in the key-frame");
            CHECK(nf >= replacedCand + replacedResident, "Fuse count covers every replacement");
        }
        // batched isInFrustum -> SearchByProjection(F, local points): the fields written on the points drive the search
        {
            Frame cur2 = fr[1];
            cur2.mvpMapPoints.assign(cur2.N, nullptr);
            std::vector<MapPoint *> local;
            for (auto &p : mps) local.push_back(&p);
            const std::vector<uint8_t> in = rumi_facade::IsInFrustum(cur2, local, 0.5f);
            int nin = 0; bool consistent = true;
            for (size_t i = 0; i < local.size(); i++) {
                nin += in[i];
                if ((in[i] != 0) != local[i]->mbTrackInView) consistent = false;
                if (in[i] && !(local[i]->mTrackProjX >= 0 && local[i]->mTrackProjX <= 640 && local[i]->mnTrackScaleLevel >= 0 && local[i]->mnTrackScaleLevel < 8)) consistent = false;
            }
            CHECK(nin > 300 && consistent, "IsInFrustum fills the tracking fields");
            const int nl = matcher.SearchByProjection(cur2, local, 3.f, false, 50.f);
            CHECK(nl > 100, "SearchByProjection(F, local points) after the batched isInFrustum");
            // the fused call must leave the same fields on the points and the same matches in the frame
            std::vector<float> px0, lv0;
            for (auto *p : local) { px0.push_back(p->mbTrackInView ? p->mTrackProjX : -2.f); lv0.push_back((float)p->mnTrackScaleLevel); p->mbTrackInView = false; p->mTrackProjX = -3.f; p->nVisible = 0; }
            Frame cur3 = fr[1];
            cur3.mvpMapPoints.assign(cur3.N, nullptr);
            int nTo = 0;
            const int nf2 = rumi_facade::SearchLocalPoints(cur3, local, 3.f, false, 50.f, 0.9f, &nTo);   // the ratio of `matcher` above
            bool same = nf2 == nl && nTo == nin;
            for (size_t i = 0; i < local.size(); i++) {
                if ((px0[i] != -2.f) != local[i]->mbTrackInView) same = false;
                if (local[i]->mbTrackInView && (local[i]->mTrackProjX != px0[i] || (float)local[i]->mnTrackScaleLevel != lv0[i] || local[i]->nVisible != 1)) same = false;
            }
            for (int f = 0; f < cur3.N; f++) if (cur3.mvpMapPoints[f] != cur2.mvpMapPoints[f]) same = false;
            if (!same) {
                int dv = 0, dx = 0, dm = 0;
                for (size_t i = 0; i < local.size(); i++) { if ((px0[i] != -2.f) != local[i]->mbTrackInView) dv++; else if (local[i]->mbTrackInView && (local[i]->mTrackProjX != px0[i] || (float)local[i]->mnTrackScaleLevel != lv0[i] || local[i]->nVisible != 1)) dx++; }
                for (int f = 0; f < cur3.N; f++) if (cur3.mvpMapPoints[f] != cur2.mvpMapPoints[f]) dm++;
                std::printf("SearchLocalPoints: matches %d vs %d, nToMatch %d vs %d, in-view flags differ %d, fields differ %d, frame entries differ %d\n", nf2, nl, nTo, nin, dv, dx, dm);
            }
            CHECK(same, "SearchLocalPoints (fused frustum + projection search) == IsInFrustum then SearchByProjection");
        }
        std::vector<cv::Point2f> prev(fr[0].mvKeysUn.size());
        for (size_t i = 0; i < prev.size(); i++) prev[i] = fr[0].mvKeysUn[i].pt;
        std::vector<int> ini;
        const int ni = matcher.SearchForInitialization(fr[0], fr[1], prev, ini, 100);
        int cnt = 0;
        for (int v : ini) cnt += v >= 0;
        CHECK(ni > 20 && ni == cnt && ini.size() == fr[0].mvKeysUn.size(), "SearchForInitialization count == surviving vnMatches12 entries");
    }

    // ---- PoseOptimization facade ~ oracle ----
    std::vector<float> Xw, ob, w;
    for (int i = 0; i < fr[1].N; i++) if (fr[1].mvpMapPoints[i]) {
        for (int c = 0; c < 3; c++) Xw.push_back(fr[1].mvpMapPoints[i]->pos.v[c]);
        ob.push_back(fr[1].mvKeysUn[i].pt.x); ob.push_back(fr[1].mvKeysUn[i].pt.y);
        w.push_back(fr[1].mvInvLevelSigma2[fr[1].mvKeysUn[i].octave]);
    }
    float Tref[7]; std::memcpy(Tref, T7, 28);
    std::vector<uint8_t> oref(w.size());
    const int goodRef = orc_pose_optimization(Xw.data(), ob.data(), w.data(), (int)w.size(), K4, Tref, oref.data());
    const int good = ORB_SLAM3::Optimizer::PoseOptimization(&fr[1]);
    CHECK(good == goodRef, "PoseOptimization inlier count");
    double dT = 0; for (int c = 0; c < 7; c++) dT = std::fmax(dT, std::fabs(fr[1].pose.T[c] - Tref[c]));
    CHECK(dT < 1e-4, "PoseOptimization pose");

    // ---- LocalBundleAdjustment facade ~ oracle on a small synthetic map ----
    Map map;
    const int nKF = 6, nMP = 300;
    std::vector<KeyFrame> kfs(nKF);
    std::vector<MapPoint> pts(nMP);
    std::normal_distribution<float> g(0.f, 1.f);
    for (int p = 0; p < nMP; p++) { pts[p].map = &map; pts[p].pos = V3f{{(rng() % 600) / 100.f - 3.f, (rng() % 400) / 100.f - 2.f, 3.f + (rng() % 500) / 100.f}}; }
    std::vector<float> kfPose, mpPos, eObs, eW; std::vector<uint8_t> kfFixed; std::vector<int32_t> eMp, eKf;
    for (int k = 0; k < nKF; k++) {
        KeyFrame &K = kfs[k];
        K.mnId = k; K.map = &map; K.mvInvLevelSigma2 = fr[0].mvInvLevelSigma2;
        const float a = 0.03f * (k - 3);
        const float T[7] = {0, std::sin(a / 2), 0, std::cos(a / 2), -0.25f * k + 0.6f, 0.01f * k, 0.02f * k};
        std::memcpy(K.pose.T, T, 28);
    }
    for (int k = 1; k < nKF - 1; k++) kfs[nKF - 1].covis.push_back(&kfs[k]);     // kf 5 is "current"; kf 0 sees points but is not covisible -> fixed
    // observations (truth projection + noise), then perturb the local key-frames and the points
    for (int p = 0; p < nMP; p++)
        for (int k = 0; k < nKF; k++) {
            KeyFrame &K = kfs[k];
            const float *T = K.pose.T;
            const float qy = T[1], qw = T[3];
            const float X = pts[p].pos.v[0], Y = pts[p].pos.v[1], Z = pts[p].pos.v[2];
            const float xc = (1 - 2 * qy * qy) * X + 2 * qy * qw * Z + T[4], yc = Y + T[5], zc = -2 * qy * qw * X + (1 - 2 * qy * qy) * Z + T[6];
            const float u = 535.4f * xc / zc + 320.1f, v = 539.2f * yc / zc + 247.6f;
            if (zc < 0.5f || u < 0 || u >= 640 || v < 0 || v >= 480) continue;
            cv::KeyPoint kp; kp.pt.x = u + 0.7f * g(rng); kp.pt.y = v + 0.7f * g(rng); kp.octave = rng() % 3;
            if (rng() % 30 == 0) kp.pt.x += 25.f;
            const int idx = (int)K.mvKeysUn.size();
            K.mvKeysUn.push_back(kp); K.mvuRight.push_back(-1.f); K.mvpMapPoints.push_back(&pts[p]); K.N++;
            pts[p].obs[&K] = std::make_tuple(idx, -1);
        }
    for (int k = 1; k < nKF; k++) { kfs[k].pose.T[4] += 0.01f * g(rng); kfs[k].pose.T[6] += 0.01f * g(rng); }
    for (int p = 0; p < nMP; p++) for (int c = 0; c < 3; c++) pts[p].pos.v[c] += 0.02f * g(rng);
    // the oracle problem in the facade's construction order: local KFs (5,1,2,3,4), fixed (0); points in first-seen order
    std::vector<int> order = {5, 1, 2, 3, 4, 0};
    std::map<KeyFrame *, int> kid;
    for (size_t i = 0; i < order.size(); i++) { kid[&kfs[order[i]]] = (int)i; kfPose.insert(kfPose.end(), kfs[order[i]].pose.T, kfs[order[i]].pose.T + 7); kfFixed.push_back(order[i] == 0); }
    std::vector<MapPoint *> plist; std::map<MapPoint *, bool> seen;
    for (int oi = 0; oi < 5; oi++) for (MapPoint *p : kfs[order[oi]].mvpMapPoints) if (p && !seen[p]) { seen[p] = true; plist.push_back(p); }
    for (size_t p = 0; p < plist.size(); p++) {
        mpPos.insert(mpPos.end(), plist[p]->pos.v, plist[p]->pos.v + 3);
        for (auto &ob2 : plist[p]->obs) {
            eMp.push_back((int)p); eKf.push_back(kid[ob2.first]);
            const cv::KeyPoint &kp = ob2.first->mvKeysUn[std::get<0>(ob2.second)];
            eObs.push_back(kp.pt.x); eObs.push_back(kp.pt.y); eW.push_back(ob2.first->mvInvLevelSigma2[kp.octave]);
        }
    }
    std::vector<uint8_t> er(eMp.size());
    std::vector<float> kpRef(kfPose), mpRef(mpPos);
    orc_local_ba((int)order.size(), kpRef.data(), kfFixed.data(), (int)plist.size(), mpRef.data(), (int)eMp.size(), eMp.data(), eKf.data(), eObs.data(),
                 eW.data(), K4, nullptr, er.data());
    int nFixed, nOpt, nMPs, nEdges; bool stop = false;
    ORB_SLAM3::Optimizer::LocalBundleAdjustment(&kfs[5], &stop, &map, nFixed, nOpt, nMPs, nEdges);
    CHECK(nFixed == 1 && nOpt == 5 && nMPs == (int)plist.size() && nEdges == (int)eMp.size(), "LBA graph sizes");
    double dK = 0, dP = 0;
    for (size_t i = 0; i < order.size(); i++) for (int c = 0; c < 7; c++) dK = std::fmax(dK, std::fabs(kfs[order[i]].pose.T[c] - kpRef[i * 7 + c]));
    for (size_t p = 0; p < plist.size(); p++) for (int c = 0; c < 3; c++) dP = std::fmax(dP, std::fabs(plist[p]->pos.v[c] - mpRef[p * 3 + c]));
    CHECK(dK < 2e-4 && dP < 1e-3, "LBA poses / points vs oracle");
    int erased = 0, erasedRef = 0;
    for (size_t e = 0; e < er.size(); e++) erasedRef += er[e];
    for (size_t p = 0; p < plist.size(); p++) erased += (int)(0);
    size_t remaining = 0; for (auto *p : plist) remaining += p->obs.size();
    CHECK(remaining + erasedRef == eMp.size(), "LBA erased observations");
    CHECK(map.changes == 1, "IncreaseChangeIndex");
    // ---- merge-window LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag) facade ~ oracle on the same map ----
    {
        std::vector<KeyFrame *> adjust = {&kfs[5], &kfs[4], &kfs[3]}, fixedK = {&kfs[0], &kfs[1], &kfs[2]};
        for (int k = 3; k < nKF; k++) { kfs[k].pose.T[4] += 0.02f * g(rng); kfs[k].pose.T[5] += 0.01f * g(rng); }
        std::vector<KeyFrame *> ord(fixedK); ord.insert(ord.end(), adjust.begin(), adjust.end());
        std::map<KeyFrame *, int> kid2;
        std::vector<float> kp2, mp2, ob2v, w2; std::vector<uint8_t> fx2; std::vector<int32_t> em2, ek2;
        for (size_t i = 0; i < ord.size(); i++) { kid2[ord[i]] = (int)i; kp2.insert(kp2.end(), ord[i]->pose.T, ord[i]->pose.T + 7); fx2.push_back(i < fixedK.size()); }
        std::vector<MapPoint *> pl2; std::set<MapPoint *> seen2;
        for (KeyFrame *k : ord) for (MapPoint *p : k->GetMapPoints()) if (seen2.insert(p).second) pl2.push_back(p);
        for (size_t p = 0; p < pl2.size(); p++) {
            mp2.insert(mp2.end(), pl2[p]->pos.v, pl2[p]->pos.v + 3);
            for (auto &o2 : pl2[p]->obs) {
                if (!o2.first->GetMapPoint(std::get<0>(o2.second))) continue;
                em2.push_back((int)p); ek2.push_back(kid2[o2.first]);
                const cv::KeyPoint &kp = o2.first->mvKeysUn[std::get<0>(o2.second)];
                ob2v.push_back(kp.pt.x); ob2v.push_back(kp.pt.y); w2.push_back(o2.first->mvInvLevelSigma2[kp.octave]);
            }
        }
        std::vector<uint8_t> er2(em2.size() + 1);
        std::vector<float> kpR(kp2), mpR(mp2);
        int32_t its2[2];
        orc_merge_ba((int)ord.size(), kpR.data(), fx2.data(), (int)pl2.size(), mpR.data(), (int)em2.size(), em2.data(), ek2.data(), ob2v.data(), w2.data(), K4,
                     nullptr, er2.data(), its2);
        bool stop2 = false;
        ORB_SLAM3::Optimizer::LocalBundleAdjustment(&kfs[5], adjust, fixedK, &stop2);
        double dK2 = 0, dP2 = 0;
        for (size_t i = 0; i < ord.size(); i++) for (int c = 0; c < 7; c++) dK2 = std::fmax(dK2, std::fabs(ord[i]->pose.T[c] - kpR[i * 7 + c]));
        for (size_t p = 0; p < pl2.size(); p++) for (int c = 0; c < 3; c++) dP2 = std::fmax(dP2, std::fabs(pl2[p]->pos.v[c] - mpR[p * 3 + c]));
        CHECK(its2[0] > 0 && its2[1] > 0 && dK2 < 2e-4 && dP2 < 1e-3, "merge BA poses / points vs oracle");
        int erasedRef2 = 0;
        for (size_t e = 0; e < em2.size(); e++) erasedRef2 += er2[e];
        size_t remaining2 = 0;
        for (auto *p : pl2) for (auto &o2 : p->obs) remaining2 += kid2.count(o2.first) && o2.first->GetMapPoint(std::get<0>(o2.second)) ? 1 : 0;
        CHECK(remaining2 + erasedRef2 == em2.size(), "merge BA erased observations");
        std::printf("merge BA: edges %zu, iterations %d + %d, erased %d, dK %.2e dP %.2e\n", em2.size(), its2[0], its2[1], erasedRef2, dK2, dP2);
    }
    // ---- arena growth: 40 000 map points exceed the matcher arena's 32 768 queries; the facade re-creates the arena and repeats the call ----
    {
        const int before = ORB_SLAM3::ORBmatcher::arena_scale();
        Frame F = fr[1];
        F.mvpMapPoints.assign(F.N, nullptr);
        std::vector<MapPoint> many(40000);
        std::vector<MapPoint *> vpMany;
        for (size_t i = 0; i < many.size(); i++) {
            MapPoint &m = many[i];
            m.desc = cv::Mat(1, 32, CV_8U); std::memset(m.desc.ptr(0), (int)(i & 255), 32);
            const size_t src = i % (size_t)fr[1].N;
            m.mbTrackInView = i < (size_t)fr[1].N;                      // the first N project onto the frame's own key-points
            m.mTrackProjX = fr[1].mvKeysUn[src].pt.x; m.mTrackProjY = fr[1].mvKeysUn[src].pt.y; m.mnTrackScaleLevel = fr[1].mvKeysUn[src].octave;
            if (m.mbTrackInView) std::memcpy(m.desc.ptr(0), fr[1].mDescriptors.ptr((int)src), 32);
            vpMany.push_back(&m);
        }
        rumi_facade::clear_status();
        ORB_SLAM3::ORBmatcher big(0.8f, true);
        const int nm = big.SearchByProjection(F, vpMany, 3.f);
        CHECK(nm > fr[1].N / 2 && ORB_SLAM3::ORBmatcher::arena_scale() > before && rumi_facade::last_status() == RUMI_OK, "matcher arena grows past 32768 queries instead of failing");
        std::printf("arena growth: %d matches of %d in-view points, arena scale %d -> %d\n", nm, fr[1].N, before, ORB_SLAM3::ORBmatcher::arena_scale());
    }
    // ---- rumi_facade::TrackFrame (one device-resident call) == the same Tracking step through the separate facade members ----
    {
        for (auto &m : mps) { m.bad = false; m.mnLastFrameSeen = -1; m.mbTrackInView = false; m.normal = V3f{{0, 0, 1}}; }
        for (size_t i = 0; i < mps.size(); i += 41) mps[i].bad = true;
        std::vector<MapPoint *> localPts;
        for (auto &m : mps) localPts.push_back(&m);
        cv::Mat image1(480, 640, CV_8U, im[1].data(), 640), mask1;
        Frame A;
        A.mnId = 21;
        ext(image1, mask1, A.mvKeysUn, A.mDescriptors, lap);
        A.N = (int)A.mvKeysUn.size();
        A.mvScaleFactors = ext.GetScaleFactors(); A.mvInvLevelSigma2 = ext.GetInverseScaleSigmaSquares();
        A.mvpMapPoints.assign(A.N, nullptr); A.mvbOutlier.assign(A.N, false); A.mvuRight.assign(A.N, -1.f);
        std::memcpy(A.pose.T, T7, 28);
        ORB_SLAM3::ORBmatcher m9(0.9f, true);
        int thUsed = 15;
        int nmA = m9.SearchByProjection(A, fr[0], 15.f, true);
        if (nmA < 20) { A.mvpMapPoints.assign(A.N, nullptr); nmA = m9.SearchByProjection(A, fr[0], 30.f, true); thUsed = 30; }
        const int g1 = ORB_SLAM3::Optimizer::PoseOptimization(&A);
        float TmotionA[7]; std::memcpy(TmotionA, A.pose.T, 28);
        int nmatchesMap = 0;
        for (int i = 0; i < A.N; i++) if (A.mvpMapPoints[i]) {                     // Tracking.cc:2489-2508
            if (A.mvbOutlier[i]) { MapPoint *p = A.mvpMapPoints[i]; A.mvpMapPoints[i] = nullptr; A.mvbOutlier[i] = false; p->mbTrackInView = false; p->mnLastFrameSeen = A.mnId; }
            else if (A.mvpMapPoints[i]->Observations() > 0) nmatchesMap++;
        }
        for (auto &p : A.mvpMapPoints) if (p) {                                     // :2998-3010
            if (p->isBad()) p = nullptr; else { p->IncreaseVisible(); p->mnLastFrameSeen = A.mnId; p->mbTrackInView = false; }
        }
        int nto = 0;
        const int nl = rumi_facade::SearchLocalPoints(A, localPts, 1.f, false, 50.f, 0.8f, &nto);
        const int g2 = ORB_SLAM3::Optimizer::PoseOptimization(&A);
        int inliersA = 0;
        for (int i = 0; i < A.N; i++) if (A.mvpMapPoints[i] && !A.mvbOutlier[i] && A.mvpMapPoints[i]->Observations() > 0) inliersA++;
        for (auto &m : mps) { m.nVisible = 0; m.nFound = 0; }
        Frame B;
        B.mnId = 22;
        rumi_facade::TrackStep st;
        const int monoB = rumi_facade::TrackFrame(B, image1, ext, T7, fr[0], localPts, 15.f, 1.f, false, 50.f, &st);
        CHECK(monoB >= 0 && B.N == A.N && std::memcmp(B.mvKeysUn.data(), A.mvKeysUn.data(), (size_t)A.N * 28) == 0, "TrackFrame: the frame's features");
        CHECK(st.thMotion == thUsed && st.nmatches == nmA && st.ngoodMotion == g1 && st.nmatchesMap == nmatchesMap && st.nToMatch == nto && st.nmatchesLocal == nl &&
              st.ngoodLocal == g2 && st.mnMatchesInliers == inliersA, "TrackFrame: the numbers TrackWithMotionModel / TrackLocalMap decide on");
        bool same = B.mvpMapPoints.size() == A.mvpMapPoints.size();
        for (int i = 0; same && i < A.N; i++) same = B.mvpMapPoints[i] == A.mvpMapPoints[i] && (!A.mvpMapPoints[i] || B.mvbOutlier[i] == A.mvbOutlier[i]);
        CHECK(same, "TrackFrame: mvpMapPoints / mvbOutlier");
        double dM = 0, dF = 0;
        for (int c = 0; c < 7; c++) { dM = std::fmax(dM, std::fabs(st.TcwMotion[c] - TmotionA[c])); dF = std::fmax(dF, std::fabs(B.pose.T[c] - A.pose.T[c])); }
        CHECK(dM < 1e-5 && dF < 1e-5, "TrackFrame: poses");
        if (!same || !(dM < 1e-5 && dF < 1e-5)) {
            int dmp = 0, dout = 0;
            for (int i = 0; i < A.N; i++) { dmp += B.mvpMapPoints[i] != A.mvpMapPoints[i]; dout += A.mvpMapPoints[i] && B.mvbOutlier[i] != A.mvbOutlier[i]; }
            std::printf("TrackFrame diff: map points %d, outlier flags %d, dM %.3e dF %.3e\n", dmp, dout, dM, dF);
            for (int i = 0, shown = 0; i < A.N && shown < 8; i++) if (B.mvpMapPoints[i] != A.mvpMapPoints[i]) { shown++; std::printf("  feature %d: A %ld B %ld\n", i, A.mvpMapPoints[i] ? (long)(A.mvpMapPoints[i] - mps.data()) : -1L, B.mvpMapPoints[i] ? (long)(B.mvpMapPoints[i] - mps.data()) : -1L); }
        }
        int vis = 0, found = 0;
        for (auto &m : mps) { vis += m.nVisible; found += m.nFound; }
        CHECK(nl > 20 && nmA >= 20 && vis > 0 && found == g2, "TrackFrame: the scene tracks; IncreaseFound once per inlier");
        std::printf("TrackFrame: motion matches %d (th %d), inliers %d, nmatchesMap %d, in view %d, local matches %d, inliers %d / %d\n", st.nmatches, st.thMotion, st.ngoodMotion,
                    st.nmatchesMap, st.nToMatch, st.nmatchesLocal, st.ngoodLocal, st.mnMatchesInliers);
    }
    // ---- the step-wise members (one Tracking member function per call, the frame resident in between) ----
    {
        for (auto &m : mps) { m.bad = false; m.mnLastFrameSeen = -1; m.mbTrackInView = false; m.normal = V3f{{0, 0, 1}}; m.nVisible = 0; m.nFound = 0; }
        for (size_t i = 0; i < mps.size(); i += 41) mps[i].bad = true;
        std::vector<MapPoint *> localPts;
        for (auto &m : mps) localPts.push_back(&m);
        cv::Mat image1(480, 640, CV_8U, im[1].data(), 640), mask1;
        // (a) the same frame through ExtractFrame -> TrackWithMotionModel -> [host: UpdateLocalMap] -> TrackLocalMap == the fused TrackFrame
        Frame F1; F1.mnId = 31;
        rumi_facade::TrackStep sf;
        const int monoF = rumi_facade::TrackFrame(F1, image1, ext, T7, fr[0], localPts, 15.f, 1.f, false, 50.f, &sf);
        std::vector<int> visF(mps.size()), foundF(mps.size());
        for (size_t i = 0; i < mps.size(); i++) { visF[i] = mps[i].nVisible; foundF[i] = mps[i].nFound; mps[i].nVisible = 0; mps[i].nFound = 0; mps[i].mnLastFrameSeen = -1; mps[i].mbTrackInView = false; }
        Frame F2; F2.mnId = 32;
        rumi_facade::TrackStep ss;
        // (the step-wise path reads the frame from the tracker's pinned capture buffer: no staging copy, same features)
        cv::Mat capture = rumi_facade::CaptureBuffer(ext, 640, 480, (int)mps.size());
        CHECK(!capture.empty() && capture.rows == 480 && capture.cols == 640 && capture.step == 640, "CaptureBuffer: a 640 x 480 Mat on the tracker's pinned memory");
        for (int y = 0; y < 480; y++) std::memcpy(capture.data + (size_t)y * capture.step, image1.data + (size_t)y * image1.step, 640);
        const int monoS = rumi_facade::ExtractFrame(F2, capture, ext, (int)mps.size());
        const bool okM = rumi_facade::TrackWithMotionModel(F2, fr[0], T7, 15.f, &ss);
        int heldAfterMotion = 0, seenAfterMotion = 0;
        for (int i = 0; i < F2.N; i++) heldAfterMotion += F2.mvpMapPoints[i] != nullptr;
        for (auto &m : mps) seenAfterMotion += m.mnLastFrameSeen == F2.mnId;
        int visAfterMotion = 0;
        for (auto &m : mps) visAfterMotion += m.nVisible;
        const int inl = rumi_facade::TrackLocalMap(F2, localPts, 1.f, false, 50.f, &ss);
        CHECK(monoF >= 0 && monoS == monoF && F2.N == F1.N && sf.okMotion && okM && ss.okMotion, "step-wise: extraction and the motion model's verdict");
        CHECK(ss.nmatches == sf.nmatches && ss.thMotion == sf.thMotion && ss.ngoodMotion == sf.ngoodMotion && ss.nmatchesMap == sf.nmatchesMap, "step-wise: TrackWithMotionModel's numbers");
        // (nmatches counts assignments, also those a later query overwrote -- the reference's own count -- so it bounds the discarded points from above)
        CHECK(visAfterMotion == 0 && seenAfterMotion > 0 && seenAfterMotion <= ss.nmatches - heldAfterMotion, "step-wise: TrackWithMotionModel touches MapPoints only in its discard loop");
        CHECK(inl == sf.mnMatchesInliers && ss.nToMatch == sf.nToMatch && ss.nmatchesLocal == sf.nmatchesLocal && ss.ngoodLocal == sf.ngoodLocal, "step-wise: TrackLocalMap's numbers");
        bool sameS = true;
        for (int i = 0; sameS && i < F1.N; i++) sameS = F1.mvpMapPoints[i] == F2.mvpMapPoints[i] && F1.mvbOutlier[i] == F2.mvbOutlier[i];
        bool sameStats = true;
        for (size_t i = 0; sameStats && i < mps.size(); i++) sameStats = mps[i].nVisible == visF[i] && mps[i].nFound == foundF[i];
        CHECK(sameS && std::memcmp(F1.pose.T, F2.pose.T, 28) == 0, "step-wise: the frame after TrackLocalMap equals the fused call's");
        CHECK(sameStats, "step-wise: IncreaseVisible / IncreaseFound per point as in the fused call");
        {   // (a2) isInFrustum's writes reach the MapPoints, and a discarded outlier that still carries the PREVIOUS frame's mbTrackInView is searched at
            // its old projection (monocular: the discard loop clears mbTrackInViewR only, Tracking.cc:2489-2508 with Nleft = -1), in both forms alike
            int stored = 0;
            for (auto &m : mps) if (m.mbTrackInView) stored += m.mTrackProjX >= 0 && m.mTrackProjX <= 640 && m.mTrackProjY >= 0 && m.mTrackProjY <= 480 && m.mnTrackScaleLevel >= 0 && m.mnTrackScaleLevel < 8 && m.mTrackDepth > 0;
            int inViewNow = 0;
            for (auto &m : mps) inViewNow += m.mbTrackInView;
            CHECK(inViewNow == ss.nToMatch && stored == inViewNow && inViewNow > 0, "TrackLocalMap stores mTrackProjX/Y, mnTrackScaleLevel, mTrackDepth of the points in view");
            // every point "was in view in the previous frame" at the feature frame 0 saw it at
            struct Saved { bool v; float x, y, c, d; int l; };
            std::vector<Saved> init(mps.size());
            for (size_t i = 0; i < mps.size(); i++) {
                const cv::KeyPoint &k = fr[0].mvKeysUn[i];
                init[i] = Saved{true, k.pt.x, k.pt.y, 1.f, 1.f, k.octave};
            }
            auto restore = [&]() {
                for (size_t i = 0; i < mps.size(); i++) {
                    MapPoint &m = mps[i];
                    m.mbTrackInView = init[i].v; m.mTrackProjX = init[i].x; m.mTrackProjY = init[i].y; m.mTrackViewCos = init[i].c; m.mTrackDepth = init[i].d; m.mnTrackScaleLevel = init[i].l;
                    m.mnLastFrameSeen = -1; m.nVisible = 0; m.nFound = 0;
                }
            };
            restore();
            Frame G1; G1.mnId = 41;
            rumi_facade::TrackStep g1;
            rumi_facade::TrackFrame(G1, image1, ext, T7, fr[0], localPts, 15.f, 1.f, false, 50.f, &g1);
            int trueAfterFused = 0;                                  // (bad points are skipped by both loops: their flag stays whatever it was)
            for (auto &m : mps) trueAfterFused += m.mbTrackInView && !m.bad;
            std::vector<char> flagsFused(mps.size());
            for (size_t i = 0; i < mps.size(); i++) flagsFused[i] = mps[i].mbTrackInView;
            restore();
            Frame G2; G2.mnId = 42;
            rumi_facade::TrackStep g2;
            rumi_facade::ExtractFrame(G2, capture, ext, (int)mps.size());
            rumi_facade::TrackWithMotionModel(G2, fr[0], T7, 15.f, &g2);
            int staleKept = 0;
            for (auto &m : mps) staleKept += m.mnLastFrameSeen == G2.mnId && m.mbTrackInView && !m.bad;      // discarded, flag untouched
            rumi_facade::TrackLocalMap(G2, localPts, 1.f, false, 50.f, &g2);
            bool sameG = G1.N == G2.N, sameFlags = true;
            for (int i = 0; sameG && i < G1.N; i++) sameG = G1.mvpMapPoints[i] == G2.mvpMapPoints[i] && G1.mvbOutlier[i] == G2.mvbOutlier[i];
            for (size_t i = 0; i < mps.size(); i++) sameFlags = sameFlags && flagsFused[i] == (char)mps[i].mbTrackInView;
            const int nStale = trueAfterFused - g1.nToMatch;
            CHECK(staleKept > 0 && nStale == staleKept, "discard loop (monocular): mbTrackInView of a discarded outlier is left as the previous frame set it");
            CHECK(sameG && sameFlags && g1.nmatchesLocal == g2.nmatchesLocal && g1.mnMatchesInliers == g2.mnMatchesInliers && g1.nToMatch == g2.nToMatch,
                  "stale in-view flags: fused and step-wise forms agree");
            CHECK(g1.nmatchesLocal != sf.nmatchesLocal || g1.mnMatchesInliers != sf.mnMatchesInliers || nStale > 0, "stale in-view flags take part in the local search");
            std::printf("stale mbTrackInView: %d discarded outliers searched at their old projection (local matches %d, %d without the flags)\n", nStale, g1.nmatchesLocal, sf.nmatchesLocal);
            for (auto &m : mps) { m.mbTrackInView = false; m.mnLastFrameSeen = -1; m.nVisible = 0; m.nFound = 0; }
        }
        {   // (a3) a camera with lens distortion (euroc_ori.yaml's coefficients): ExtractFrame fills mvKeysUn with Frame::UndistortKeyPoints' result,
            // mvKeys stay the extractor's, ImageBounds gives ComputeImageBounds' hull of the undistorted corners; switched off again afterwards
            const float Kd[4] = {F1.fx, F1.fy, F1.cx, F1.cy}, dist4[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
            rumi_facade::SetDistortion(Kd, dist4, 4);
            Frame D1; D1.mnId = 51;
            const int monoD = rumi_facade::ExtractFrame(D1, capture, ext, (int)mps.size());
            float bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
            const bool okB = rumi_facade::ImageBounds(bx0, by0, bx1, by1);
            double maxShift = 0;
            bool levelsKept = D1.N == F2.N;
            for (int i = 0; levelsKept && i < D1.N; i++) {
                maxShift = std::fmax(maxShift, std::hypot(D1.mvKeysUn[i].pt.x - F2.mvKeysUn[i].pt.x, D1.mvKeysUn[i].pt.y - F2.mvKeysUn[i].pt.y));
                levelsKept = D1.mvKeysUn[i].octave == F2.mvKeysUn[i].octave && D1.mvKeysUn[i].angle == F2.mvKeysUn[i].angle;
            }
            CHECK(monoD == monoS && levelsKept && maxShift > 5.0 && okB && bx0 < -5.f && by0 < -5.f && bx1 > 645.f && by1 > 485.f, "SetDistortion: mvKeysUn undistorted on the device, bounds = the undistorted corners");
            rumi_facade::SetDistortion(Kd, nullptr, 0);
            Frame D2; D2.mnId = 52;
            rumi_facade::ExtractFrame(D2, capture, ext, (int)mps.size());
            bool backSame = D2.N == F2.N;
            for (int i = 0; backSame && i < D2.N; i++) backSame = D2.mvKeysUn[i].pt.x == F2.mvKeysUn[i].pt.x && D2.mvKeysUn[i].pt.y == F2.mvKeysUn[i].pt.y;
            CHECK(backSame && rumi_facade::ImageBounds(bx0, by0, bx1, by1) && bx0 == 0.f && bx1 == 640.f && by1 == 480.f, "SetDistortion off: mvKeysUn = mvKeys, the image rectangle");
        }
        // (b) a hopeless prediction: the fused call reports okMotion = false, has replayed NOTHING of TrackLocalMap, and the frame is in
        // TrackWithMotionModel's failure state; TrackReferenceKeyFrame takes over on the resident frame
        for (auto &m : mps) { m.nVisible = 0; m.nFound = 0; m.mnLastFrameSeen = -1; m.mbTrackInView = false; }
        const float Tbad[7] = {0, 0, 0, 1, 9.f, 0, 0};
        Frame F3; F3.mnId = 33;
        rumi_facade::TrackStep sb;
        rumi_facade::TrackFrame(F3, image1, ext, Tbad, fr[0], localPts, 15.f, 1.f, false, 50.f, &sb);
        int touched = 0;
        for (auto &m : mps) touched += m.nVisible + m.nFound + (m.mnLastFrameSeen == F3.mnId);
        CHECK(!sb.okMotion && !sb.ranLocal && sb.nmatches < 20 && sb.thMotion == 30 && touched == 0 && F3.N == F1.N, "fused call, motion model fails: nothing of TrackLocalMap replayed");
        // the reference key-frame: frame 0 with its map points, FeatureVectors from a vocabulary whose words are the scene's descriptors
        const char *vocPath = "/tmp/rumi_facade_voc.txt";
        {
            FILE *vf = std::fopen(vocPath, "w");
            const int k = 12, L = 2;
            std::fprintf(vf, "%d %d 0 0\n", k, L);
            std::mt19937 vr(5);
            std::vector<int> inner;
            int id = 0;
            auto put = [&](int parent, int leaf, const uint8_t *d) { std::fprintf(vf, "%d %d", parent, leaf); for (int b = 0; b < 32; b++) std::fprintf(vf, " %d", (int)d[b]); std::fprintf(vf, " %.17g\n", leaf ? 1.0 + (id % 7) * 0.25 : 0.0); id++; };
            for (int a = 0; a < k; a++) { put(0, 0, fr[0].mDescriptors.ptr((int)(vr() % fr[0].N))); inner.push_back(id); }
            for (int a = 0; a < k; a++) for (int b = 0; b < k; b++) put(inner[a], 1, fr[0].mDescriptors.ptr((int)(vr() % fr[0].N)));
            std::fclose(vf);
        }
        rumi_facade::ORBVocabulary voc;
        CHECK(voc.loadFromTextFile(vocPath) && voc.size() == 144, "vocabulary text file");
        KeyFrame kref; static_cast<Frame &>(kref) = fr[0];
        {
            std::vector<cv::Mat> dd;
            for (int i = 0; i < kref.N; i++) dd.push_back(cv::Mat(1, 32, CV_8U, kref.mDescriptors.ptr(i), 32));
            std::map<unsigned, double> bow;
            voc.transform(dd, bow, kref.mFeatVec, 1);
        }
        Frame lastF = fr[0];
        const float Tid[7] = {0, 0, 0, 1, 0, 0, 0};
        std::memcpy(lastF.pose.T, Tid, 28);
        const bool okR = rumi_facade::TrackReferenceKeyFrame(F3, &kref, lastF, voc, &sb, 1);
        // the same through the separate facade members: transform -> SearchByBoW -> PoseOptimization -> discard
        Frame G = F3;
        G.mvpMapPoints.assign(G.N, nullptr); G.mvbOutlier.assign(G.N, false);
        G.mvScaleFactors = ext.GetScaleFactors(); G.mvInvLevelSigma2 = ext.GetInverseScaleSigmaSquares(); G.mvuRight.assign(G.N, -1.f);
        {
            std::vector<cv::Mat> dd;
            for (int i = 0; i < G.N; i++) dd.push_back(cv::Mat(1, 32, CV_8U, G.mDescriptors.ptr(i), 32));
            std::map<unsigned, double> bow;
            G.mFeatVec.clear();
            voc.transform(dd, bow, G.mFeatVec, 1);
        }
        CHECK(G.mFeatVec == F3.mFeatVec && !F3.mFeatVec.empty(), "TrackReferenceKeyFrame: mFeatVec of the frame = ORBVocabulary::transform");
        std::vector<MapPoint *> vpm;
        ORB_SLAM3::ORBmatcher m7(0.7f, true);
        const int nbow = m7.SearchByBoW(&kref, G, vpm);
        CHECK(nbow == sb.nmatchesBoW && nbow >= 15, "TrackReferenceKeyFrame: SearchByBoW's count");
        G.mvpMapPoints = vpm;
        std::memcpy(G.pose.T, Tid, 28);
        const int gR = ORB_SLAM3::Optimizer::PoseOptimization(&G);
        int mapR = 0;
        for (int i = 0; i < G.N; i++) if (G.mvpMapPoints[i]) { if (G.mvbOutlier[i]) { G.mvpMapPoints[i] = nullptr; G.mvbOutlier[i] = false; } else if (G.mvpMapPoints[i]->Observations() > 0) mapR++; }
        bool sameR = true;
        for (int i = 0; sameR && i < G.N; i++) sameR = G.mvpMapPoints[i] == F3.mvpMapPoints[i];
        double dR = 0;
        for (int c = 0; c < 7; c++) dR = std::fmax(dR, std::fabs(G.pose.T[c] - F3.pose.T[c]));
        CHECK(sb.ngoodMotion == gR && sb.nmatchesMap == mapR && okR == (mapR >= 10) && sameR && dR < 1e-5, "TrackReferenceKeyFrame == transform + SearchByBoW + PoseOptimization + discard");
        std::printf("step-wise: motion %d matches (%d map), local inliers %d; fall-back: motion %d matches at th %d -> BoW %d matches, %d inliers, %d map matches\n", ss.nmatches, ss.nmatchesMap, inl,
                    sb.nmatches, sb.thMotion, sb.nmatchesBoW, sb.ngoodMotion, sb.nmatchesMap);
    }
    std::printf("facade test: %d failure(s); matches %d, pose inliers %d, LBA edges %d (erased %d) dK %.2e dP %.2e\n", fails, ngpu, good, nEdges, erasedRef, dK, dP);
    (void)erased;
    return fails ? 1 : 0;
}
