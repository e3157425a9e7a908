// The class declaration a maintainer's tree has in include/cloud_edge_slam_lib/Sim3Solver.h:28-137 (members and signatures only, over the mock data
// model), so that rumi-slam_amd/facade/shells/Sim3Solver.cc -- which defines these members -- can be compiled and run here.
#ifndef SIM3SOLVER_H
#define SIM3SOLVER_H
#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"

#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace ORB_SLAM3 {
using std::vector;
using GeometricCamera = ::Camera;

class Sim3Solver {
public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Sim3Solver();
    Sim3Solver(KeyFrame *pKF1, KeyFrame *pKF2, const std::vector<MapPoint *> &vpMatched12, const bool bFixScale = true,
               const vector<KeyFrame *> vpKeyFrameMatchedMP = vector<KeyFrame *>());
    void SetRansacParameters(double probability = 0.99, int minInliers = 6, int maxIterations = 300);
    Eigen::Matrix4f find(std::vector<bool> &vbInliers12, int &nInliers);
    Eigen::Matrix4f iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers);
    Eigen::Matrix4f iterate(int nIterations, bool &bNoMore, vector<bool> &vbInliers, int &nInliers, bool &bConverge);
    Eigen::Matrix4f iterate(int nIterations, bool &bNoMore, vector<bool> &vbInliers, int &nInliers, bool &bConverge,
                            const std::vector<KeyFrame *> &map1KFs, const std::vector<KeyFrame *> &map2KFs,
                            const std::vector<std::vector<std::pair<int, int>>> &avpValidKPMatches, float &bestRatio, Eigen::Matrix3f &bestRotation,
                            Eigen::Vector3f &bestTranslation, float &bestScale);
    static Eigen::Matrix4d umeyamaSolve(const vector<Eigen::Vector3d> &srcMatchPoints, const vector<Eigen::Vector3d> &dstMatchPoints);
    static float ComputeInliersNum(const std::vector<KeyFrame *> &map1KFs, const std::vector<KeyFrame *> &map2KFs,
                                   const std::vector<std::vector<std::pair<int, int>>> &avpValidKPMatches, g2o::Sim3 &gSw1w2);
    Eigen::Matrix4f GetEstimatedTransformation();
    Eigen::Matrix3f GetEstimatedRotation();
    Eigen::Vector3f GetEstimatedTranslation();
    float GetEstimatedScale();

protected:
    void ComputeCentroid(Eigen::Matrix3f &P, Eigen::Matrix3f &Pr, Eigen::Vector3f &C);
    void ComputeSim3(Eigen::Matrix3f &P1, Eigen::Matrix3f &P2);
    void CheckInliers();
    void Project(const std::vector<Eigen::Vector3f> &vP3Dw, std::vector<Eigen::Vector2f> &vP2D, Eigen::Matrix4f Tcw, GeometricCamera *pCamera);
    void FromCameraToImage(const std::vector<Eigen::Vector3f> &vP3Dc, std::vector<Eigen::Vector2f> &vP2D, GeometricCamera *pCamera);

protected:
    KeyFrame *mpKF1;
    KeyFrame *mpKF2;
    std::vector<Eigen::Vector3f> mvX3Dc1;
    std::vector<Eigen::Vector3f> mvX3Dc2;
    std::vector<MapPoint *> mvpMapPoints1;
    std::vector<MapPoint *> mvpMapPoints2;
    std::vector<MapPoint *> mvpMatches12;
    std::vector<size_t> mvnIndices1;
    std::vector<size_t> mvSigmaSquare1;
    std::vector<size_t> mvSigmaSquare2;
    std::vector<size_t> mvnMaxError1;
    std::vector<size_t> mvnMaxError2;
    int N;
    int mN1;
    Eigen::Matrix3f mR12i;
    Eigen::Vector3f mt12i;
    float ms12i;
    Eigen::Matrix4f mT12i;
    Eigen::Matrix4f mT21i;
    std::vector<bool> mvbInliersi;
    int mnInliersi;
    int mnIterations;
    std::vector<bool> mvbBestInliers;
    int mnBestInliers;
    Eigen::Matrix4f mBestT12;
    Eigen::Matrix3f mBestRotation;
    Eigen::Vector3f mBestTranslation;
    float mBestScale;
    bool mbFixScale;
    std::vector<size_t> mvAllIndices;
    std::vector<Eigen::Vector2f> mvP1im1;
    std::vector<Eigen::Vector2f> mvP2im2;
    double mRansacProb;
    int mRansacMinInliers;
    int mRansacMaxIts;
    float mTh;
    float mSigma2;
    GeometricCamera *pCamera1, *pCamera2;
};
}  // namespace ORB_SLAM3
#endif
