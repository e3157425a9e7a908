// Parity-pinning kit, reference side.  NOT built in this repository (it needs the reference tree, OpenCV 3.4 and Eigen, none of which are in
// the build image): a maintainer compiles it INSIDE the reference workspace, e.g.
//
//   g++ -O2 -std=c++14 tools/dump_reference_vectors.cc -I<R>/include -I<R>/include/cloud_edge_slam_lib -I<R>/Thirdparty/g2o \
//       `pkg-config --cflags --libs opencv` -I/usr/include/eigen3 -L<R>/lib -lcloud_edge_slam_lib -lg2o -o dump_reference_vectors
//   python tests/golden/make_ref_inputs.py /tmp/ref_in          # the seeded frames / optimiser problems this repository tests with
//   ./dump_reference_vectors /tmp/ref_in tests/golden/ref       # what the REAL OpenCV / Eigen / g2o / ORBextractor produce for them
//
// and commits tests/golden/ref/*.npy.  tests/test_reference_vectors.py then compares the oracle (and through it the HIP kernels, which are held
// bit-exact to the oracle) with these files and says, per primitive, which documented variant the linked OpenCV build implements:
//   level_<f>_<l>.npy        mvImagePyramid[l] without its border (cv::resize chain, ORBextractor.cc:1093-1112)
//   blur_<f>_<l>.npy         cv::GaussianBlur(level, 7x7, 2, 2, BORDER_REFLECT_101) (ORBextractor.cc:1058)   -> RumiOrbConfig.blur_variant
//   fast20_<f>_<l>.npy / fast7_  cv::FAST(level, threshold, true) as rows (x, y, response)                   (ORBextractor.cc:767,784)
//   kp_<f>.npy, desc_<f>.npy, mono_<f>.npy   ORBextractor::operator() outputs (28-byte cv::KeyPoint records as 7 float32 columns, octave / class_id
//                            bit-cast)                                                                         (ORBextractor.cc:1014-1091)
//   atan2.npy                cv::fastAtan2 on a fixed grid                                                     (ORBextractor.cc:96)
//   pose_<p>.npy             Optimizer-style PoseOptimization through g2o on problem p: (n_good, qx qy qz qw tx ty tz)   (Optimizer.cc:723-1001)
// Inputs (written by make_ref_inputs.py): frame_<f>.npy (uint8 HxW), pose_in_<p>.npy (float64 rows: Xw(3) obs(2) invSigma2(1), first row = K4, T0(7)).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "ORBextractor.h"
#include "OptimizableTypes.h"
#include "Thirdparty/g2o/g2o/core/block_solver.h"
#include "Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_dense.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"

// ---- minimal .npy (version 1.0) reader / writer for C-order arrays of u1 / f4 / f8 / i4 ----
static bool npy_write(const std::string &path, const char *descr, const std::vector<size_t> &shape, const void *data, size_t bytes) {
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); i++) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write((const char *)magic, 8);
    const uint16_t hl = (uint16_t)hdr.size();
    f.write((const char *)&hl, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char *)data, bytes);
    return (bool)f;
}
static bool npy_read(const std::string &path, std::vector<size_t> &shape, std::vector<unsigned char> &data, std::string &descr) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    char magic[8]; uint16_t hl;
    f.read(magic, 8); f.read((char *)&hl, 2);
    std::string hdr(hl, ' ');
    f.read(&hdr[0], hl);
    const size_t d0 = hdr.find("'descr': '") + 10;
    descr = hdr.substr(d0, hdr.find("'", d0) - d0);
    const size_t s0 = hdr.find("'shape': (") + 10, s1 = hdr.find(")", s0);
    shape.clear();
    size_t n = 1;
    for (size_t p = s0; p < s1;) {
        while (p < s1 && (hdr[p] == ' ' || hdr[p] == ',')) p++;
        if (p >= s1) break;
        shape.push_back(std::stoul(hdr.substr(p)));
        n *= shape.back();
        while (p < s1 && hdr[p] != ',') p++;
    }
    const size_t item = descr.back() == '1' ? 1 : descr.back() == '4' ? 4 : 8;
    data.resize(n * item);
    f.read((char *)data.data(), data.size());
    return (bool)f;
}

static void dump_mat_u8(const std::string &path, const cv::Mat &m) {
    cv::Mat c = m.clone();                                   // dense rows
    npy_write(path, "|u1", {(size_t)c.rows, (size_t)c.cols}, c.data, (size_t)c.rows * c.cols);
}

static void dump_fast(const std::string &path, const cv::Mat &img, int thr) {
    std::vector<cv::KeyPoint> k;
    cv::FAST(img, k, thr, true);
    std::vector<float> rows;
    for (auto &p : k) { rows.push_back(p.pt.x); rows.push_back(p.pt.y); rows.push_back(p.response); }
    npy_write(path, "<f4", {k.size(), 3}, rows.data(), rows.size() * 4);
}

// Optimizer::PoseOptimization's graph (Optimizer.cc:723-1001) on flat arrays: one VertexSE3Expmap, EdgeSE3ProjectXYZOnlyPose edges with Huber
// sqrt(5.991), 4 rounds x optimize(10) with the reference's outlier re-classification
static void pose_problem(const std::string &in, const std::string &out) {
    std::vector<size_t> sh; std::vector<unsigned char> raw; std::string d;
    if (!npy_read(in, sh, raw, d) || d != "<f8" || sh.size() != 2 || sh[1] != 11) return;
    const double *r = (const double *)raw.data();
    const double fx = r[0], fy = r[1], cx = r[2], cy = r[3];
    Eigen::Quaterniond q(r[7], r[4], r[5], r[6]);            // row 0: K4, then T0 = (qx qy qz qw tx ty tz)
    g2o::SE3Quat T0(q, Eigen::Vector3d(r[8], r[9], r[10]));
    const int n = (int)sh[0] - 1;
    g2o::SparseOptimizer optimizer;
    auto *linearSolver = new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>();
    auto *solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
    optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(solver_ptr));
    auto *vSE3 = new g2o::VertexSE3Expmap();
    vSE3->setEstimate(T0); vSE3->setId(0); vSE3->setFixed(false);
    optimizer.addVertex(vSE3);
    // the reference's edge type carries a GeometricCamera*; a pinhole with K4 is what Frame::mpCamera is in the mono configuration
    ORB_SLAM3::Pinhole cam(std::vector<float>{(float)fx, (float)fy, (float)cx, (float)cy});
    std::vector<ORB_SLAM3::EdgeSE3ProjectXYZOnlyPose *> edges;
    std::vector<char> outlier(n, 0);
    const float deltaMono = std::sqrt(5.991f);
    for (int i = 0; i < n; i++) {
        const double *e = r + 11 * (i + 1);
        auto *edge = new ORB_SLAM3::EdgeSE3ProjectXYZOnlyPose();
        edge->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex *>(optimizer.vertex(0)));
        edge->setMeasurement(Eigen::Vector2d(e[3], e[4]));
        edge->setInformation(Eigen::Matrix2d::Identity() * e[5]);
        auto *rk = new g2o::RobustKernelHuber; edge->setRobustKernel(rk); rk->setDelta(deltaMono);
        edge->pCamera = &cam;
        edge->Xw = Eigen::Vector3d((float)e[0], (float)e[1], (float)e[2]);      // cv::Mat(float) -> double, as Optimizer.cc:792
        optimizer.addEdge(edge);
        edges.push_back(edge);
    }
    const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f};
    int nBad = 0;
    for (size_t it = 0; it < 4; it++) {
        vSE3->setEstimate(T0);                                  // the reference restarts every round from the frame's pose (Optimizer.cc:913)
        optimizer.initializeOptimization(0);
        optimizer.optimize(10);
        nBad = 0;
        for (int i = 0; i < n; i++) {
            auto *e = edges[i];
            if (outlier[i]) e->computeError();
            const float chi2 = e->chi2();
            if (chi2 > chi2Mono[it]) { outlier[i] = 1; e->setLevel(1); nBad++; } else { outlier[i] = 0; e->setLevel(0); }
            if (it == 2) e->setRobustKernel(0);
        }
        if (optimizer.edges().size() < 10) break;
    }
    const g2o::SE3Quat T = vSE3->estimate();
    const Eigen::Quaterniond qo = T.rotation();
    const double res[8] = {(double)(n - nBad), qo.x(), qo.y(), qo.z(), qo.w(), T.translation()[0], T.translation()[1], T.translation()[2]};
    npy_write(out, "<f8", {8}, res, sizeof res);
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <input-dir> <output-dir>\n", argv[0]); return 2; }
    const std::string in = argv[1], out = argv[2];
    for (int f = 0;; f++) {
        std::vector<size_t> sh; std::vector<unsigned char> raw; std::string d;
        if (!npy_read(in + "/frame_" + std::to_string(f) + ".npy", sh, raw, d) || d != "|u1" || sh.size() != 2) break;
        cv::Mat img((int)sh[0], (int)sh[1], CV_8UC1, raw.data());
        ORB_SLAM3::ORBextractor ext(1000, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kps; cv::Mat desc; std::vector<int> lap = {0, 1000};
        const int mono = ext(img, cv::Mat(), kps, desc, lap);
        std::vector<float> rows;
        for (auto &k : kps) {
            float oc, ci; std::memcpy(&oc, &k.octave, 4); std::memcpy(&ci, &k.class_id, 4);
            const float r7[7] = {k.pt.x, k.pt.y, k.size, k.angle, k.response, oc, ci};
            rows.insert(rows.end(), r7, r7 + 7);
        }
        npy_write(out + "/kp_" + std::to_string(f) + ".npy", "<f4", {kps.size(), 7}, rows.data(), rows.size() * 4);
        dump_mat_u8(out + "/desc_" + std::to_string(f) + ".npy", desc);
        const int32_t m32 = mono;
        npy_write(out + "/mono_" + std::to_string(f) + ".npy", "<i4", {1}, &m32, 4);
        for (int l = 0; l < 8; l++) {
            const cv::Mat &lv = ext.mvImagePyramid[l];           // ROI view into the bordered temp image: clone() drops the border
            const std::string tag = std::to_string(f) + "_" + std::to_string(l);
            dump_mat_u8(out + "/level_" + tag + ".npy", lv);
            cv::Mat b = lv.clone();
            cv::GaussianBlur(b, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
            dump_mat_u8(out + "/blur_" + tag + ".npy", b);
            dump_fast(out + "/fast20_" + tag + ".npy", lv.clone(), 20);
            dump_fast(out + "/fast7_" + tag + ".npy", lv.clone(), 7);
        }
        std::printf("frame %d: %zu key-points, monoIndex %d, OpenCV %s\n", f, kps.size(), mono, CV_VERSION);
    }
    std::vector<float> at;
    for (int y = -40; y <= 40; y++) for (int x = -40; x <= 40; x++) at.push_back(cv::fastAtan2((float)y * 37.f, (float)x * 53.f));
    npy_write(out + "/atan2.npy", "<f4", {81, 81}, at.data(), at.size() * 4);
    for (int p = 0;; p++) {
        std::ifstream probe(in + "/pose_in_" + std::to_string(p) + ".npy");
        if (!probe) break;
        pose_problem(in + "/pose_in_" + std::to_string(p) + ".npy", out + "/pose_" + std::to_string(p) + ".npy");
    }
    return 0;
}
