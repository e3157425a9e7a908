// Mock data model with the reference's member names over tests/cpp/mock_sophus.h (Eigen / Sophus / g2o stand-ins): only what the facades touch.
// Shared by test_facade_sophus.cc (facade templates instantiated directly) and test_shells.cc (the non-template shells of facade/shells/).
#pragma once
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>

#include "mock_sophus.h"
#include "cv_shim.h"

struct KeyFrame;
struct MapPoint;
struct Map {
    std::mutex mMutexMapUpdate; long GetInitKFid() { return 0; } int changes = 0; void IncreaseChangeIndex() { changes++; }
    std::vector<KeyFrame *> allKFs; std::vector<MapPoint *> allMPs;
    KeyFrame *GetOriginKF() { return allKFs[0]; }
    std::vector<KeyFrame *> GetAllKeyFrames() { return allKFs; }
    std::vector<MapPoint *> GetAllMapPoints() { return allMPs; }
};
struct Camera {
    float fx = 535.4f, fy = 539.2f, cx = 320.1f, cy = 247.6f;
    float getParameter(int i) const { return i == 0 ? fx : i == 1 ? fy : i == 2 ? cx : cy; }     // GeometricCamera::getParameter
    Eigen::Vector2f project(const Eigen::Vector3f &p) const { return Eigen::Vector2f{{fx * p(0) / p(2) + cx, fy * p(1) / p(2) + cy}}; }
    Eigen::Matrix3f toK_() const { Eigen::Matrix3f K; K(0, 0) = fx; K(0, 2) = cx; K(1, 1) = fy; K(1, 2) = cy; return K; }
};
struct MapPoint {
    static std::mutex mGlobalMutex;
    Eigen::Vector3f pos, normal{0, 0, 1}; cv::Mat desc; int nObs = 1; bool bad = false, isEdge = false; Map *map = nullptr; long mnBALocalForKF = -1;
    std::map<KeyFrame *, std::tuple<int, int>> obs;
    float minD = 0.5f, maxD = 60.f; int mnTrackScaleLevel = 0; Eigen::Vector3f mPosGBA; unsigned long mnBAGlobalForKF = 0;
    bool mbTrackInView = false; float mTrackProjX = 0, mTrackProjY = 0, mTrackViewCos = 1, mTrackDepth = 1; long mnBALocalForMerge = -1;
    Eigen::Vector3f GetWorldPos() { return pos; }
    Eigen::Vector3f GetNormal() { return normal; }
    void SetWorldPos(const Eigen::Vector3f &p) { pos = p; }
    cv::Mat GetDescriptor() { return desc; }
    int Observations() { return nObs; }
    bool isBad() { return bad; }
    Map *GetMap() { return map; }
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { return obs; }
    void EraseObservation(KeyFrame *k) { obs.erase(k); }
    float GetMinDistance() { return minD; }
    float GetMaxDistance() { return maxD; }
    void UpdateNormalAndDepth() {}
    bool IsInKeyFrame(KeyFrame *k) { return obs.count(k) > 0; }
    void AddObservation(KeyFrame *k, int idx) { if (!obs.count(k)) nObs++; obs[k] = std::make_tuple(idx, -1); }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame *k) { auto it = obs.find(k); return it == obs.end() ? std::make_tuple(-1, -1) : it->second; }
    void Replace(MapPoint *) { bad = true; }
};
std::mutex MapPoint::mGlobalMutex;
struct Frame {
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn; cv::Mat mDescriptors; std::vector<MapPoint *> mvpMapPoints; std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2, mvLevelSigma2;
    std::vector<cv::KeyPoint> mvKeys;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 640, mnMaxY = 480, fx = 535.4f, fy = 539.2f, cx = 320.1f, cy = 247.6f;
    float mfLogScaleFactor = 0.1823216f; int mnScaleLevels = 8;
    Sophus::SE3f pose;
    Camera cam, *mpCamera = &cam;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    Sophus::SE3f GetPose() const { return pose; }
    void SetPose(const Sophus::SE3f &T) { pose = T; }
};
struct KeyFrame : Frame {
    long mnId = 0, mnBALocalForKF = -1, mnBAFixedForKF = -1, mnBALocalForMerge = -1; Map *map = nullptr; bool bad = false;
    std::vector<float> mvuRight; Sophus::SE3f mTcwGBA; unsigned long mnBAGlobalForKF = 0;
    std::vector<KeyFrame *> covis;
    KeyFrame() {}
    KeyFrame(const KeyFrame &o) : Frame(o), mnId(o.mnId), map(o.map) { mpCamera = &cam; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return covis; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint *> GetMapPoints() { std::set<MapPoint *> s; for (auto *p : mvpMapPoints) if (p) s.insert(p); return s; }
    MapPoint *GetMapPoint(size_t i) { return mvpMapPoints[i]; }
    void AddMapPoint(MapPoint *p, size_t i) { mvpMapPoints[i] = p; }
    void EraseMapPointMatch(MapPoint *p) { for (auto &q : mvpMapPoints) if (q == p) q = nullptr; }
    Sophus::SE3f GetPoseInverse() const { return pose.inverse(); }
    Eigen::Matrix3f GetRotation() const { return pose.rotationMatrix(); }
    Eigen::Vector3f GetTranslation() const { return pose.translation(); }
    Eigen::Vector3f GetCameraCenter() const { return pose.inverse().translation(); }
    bool isBad() { return bad; }
    Map *GetMap() { return map; }
};

