// dump_reference_vectors — runs the REAL PanoVLM functions on the inputs of this repo's golden fixtures and writes
// their outputs, so that the CPU oracle (and through it the HIP path) can be re-pinned against the true reference
// (SURVEY.md §8c).  NOT built by default and not buildable in the development image: it needs PanoVLM's own
// dependencies (Eigen 3.4, PCL 1.10, Ceres 2.0, OpenCV 3.4, glog, Boost).  It contains no PanoVLM code — it only
// calls PanoVLM's public API.
//
// Build, from a PanoVLM checkout that has been compiled (PANOVLM = its root, BUILD = its build dir):
//   g++ -std=c++17 -O2 -fopenmp -I$PANOVLM -I/usr/include/eigen3 -I/usr/include/pcl-1.10 \
//       tools/dump_reference_vectors.cpp $BUILD/libPanoVLM_lib.a (or the object files of base/ sensors/ lidar_mapping/
//       joint_optimization/ util/) `pkg-config --libs opencv` -lceres -lglog -lpcl_common -lpcl_kdtree -lpcl_search
//       -lpcl_filters -lpcl_segmentation -lpcl_io -lboost_filesystem -lboost_system -o dump_reference_vectors
// Use:  python tools/refvec.py export DIR ; ./dump_reference_vectors DIR ; python tools/refvec.py compare DIR
//
// Every block below names the reference entry point it exercises (file:line in PanoVLM).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "base/CostFunction.h"                              // the six functors (X::Create)
#include "base/Math.h"                                      // FastAtan2
#include "joint_optimization/CameraLidarLineAssociate.h"    // AssociateByAngle
#include "lidar_mapping/LidarFeatureAssociate.h"            // FindNeighbors, AssociatePoint2Plane, AssociateLine2Line
#include "sensors/Equirectangular.h"
#include "sensors/Velodyne.h"
#include "util/Visualization.h"                             // ProjectLidar2PanoramaDepth

namespace {

// ---- .pvv container (tools/refvec.py) -------------------------------------------------------------
struct Array {
  int code = 1;                       // 0 f32, 1 f64, 2 i32, 3 i64
  std::vector<uint64_t> dims;
  std::vector<char> bytes;
  size_t count() const { size_t n = 1; for (uint64_t d : dims) n *= d; return n; }
  const float* f32() const { return reinterpret_cast<const float*>(bytes.data()); }
  const double* f64() const { return reinterpret_cast<const double*>(bytes.data()); }
  const int32_t* i32() const { return reinterpret_cast<const int32_t*>(bytes.data()); }
  double as_double(size_t i) const {
    switch (code) { case 0: return f32()[i]; case 1: return f64()[i]; case 2: return i32()[i];
                    default: return (double)reinterpret_cast<const int64_t*>(bytes.data())[i]; }
  }
};
using Bundle = std::map<std::string, Array>;
const size_t kItem[4] = {4, 8, 4, 8};

Bundle ReadBundle(const std::string& path) {
  Bundle b;
  std::ifstream f(path, std::ios::binary);
  char magic[4]; uint32_t count = 0;
  if (!f.read(magic, 4) || std::memcmp(magic, "PVV1", 4) != 0) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); return b; }
  f.read(reinterpret_cast<char*>(&count), 4);
  for (uint32_t k = 0; k < count; ++k) {
    uint32_t nl = 0, nd = 0; uint8_t code = 0;
    f.read(reinterpret_cast<char*>(&nl), 4);
    std::string name(nl, '\0'); f.read(&name[0], nl);
    f.read(reinterpret_cast<char*>(&code), 1); f.read(reinterpret_cast<char*>(&nd), 4);
    Array a; a.code = code; a.dims.resize(nd);
    f.read(reinterpret_cast<char*>(a.dims.data()), 8 * nd);
    a.bytes.resize(a.count() * kItem[code]);
    f.read(a.bytes.data(), a.bytes.size());
    b[name] = std::move(a);
  }
  return b;
}

template <typename T> Array MakeArray(int code, std::vector<uint64_t> dims, const std::vector<T>& v) {
  Array a; a.code = code; a.dims = std::move(dims);
  a.bytes.resize(v.size() * sizeof(T));
  if (!v.empty()) std::memcpy(a.bytes.data(), v.data(), a.bytes.size());
  return a;
}
void WriteBundle(const std::string& path, const Bundle& b) {
  std::ofstream f(path, std::ios::binary);
  uint32_t count = (uint32_t)b.size();
  f.write("PVV1", 4); f.write(reinterpret_cast<const char*>(&count), 4);
  for (const auto& kv : b) {
    uint32_t nl = (uint32_t)kv.first.size(), nd = (uint32_t)kv.second.dims.size(); uint8_t code = (uint8_t)kv.second.code;
    f.write(reinterpret_cast<const char*>(&nl), 4); f.write(kv.first.data(), nl);
    f.write(reinterpret_cast<const char*>(&code), 1); f.write(reinterpret_cast<const char*>(&nd), 4);
    f.write(reinterpret_cast<const char*>(kv.second.dims.data()), 8 * nd);
    f.write(kv.second.bytes.data(), kv.second.bytes.size());
  }
  std::printf("wrote %s (%u arrays)\n", path.c_str(), count);
}

Eigen::Vector3d V3(const double* p) { return Eigen::Vector3d(p[0], p[1], p[2]); }

// ---- the six residual functors: X::Create(...)->Evaluate (base/CostFunction.h:350-507, :567-934) ---------------
void DumpFunctors(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/functors.in.pvv");
  if (in.empty()) return;
  Bundle out;
  const double weight = in.at("weight").as_double(0);
  const int variants[8][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}, {3, 0}, {3, 1}, {4, 0}, {5, 0}};
  for (const auto& var : variants) {
    const int kind = var[0]; const bool normalize = var[1] != 0;
    const std::string k = "k" + std::to_string(kind) + "_n" + std::to_string(var[1]) + "_";
    const Array &aa = in.at(k + "aa"), &t = in.at(k + "t"), &ref = in.at(k + "ref"), &nei = in.at(k + "nei"),
                &off = in.at(k + "off"), &rows = in.at(k + "rows");
    const size_t stride = rows.dims[1], n = rows.dims[0], pairs = ref.count();
    std::vector<double> r(n), J(n * 12);
    for (size_t p = 0; p < pairs; ++p) {
      const int ri = (int)ref.as_double(p), ni = (int)nei.as_double(p);
      for (size_t i = (size_t)off.as_double(p); i < (size_t)off.as_double(p + 1); ++i) {
        const double* row = rows.f64() + i * stride;
        ceres::CostFunction* cost = nullptr;
        switch (kind) {                                       // the call sites of util/Optimization.cpp:547-555, :410-434, :583-602
          case 0: cost = Point2Plane_Meter::Create(V3(row), Eigen::Vector4d(row[3], row[4], row[5], row[6]), weight); break;
          case 1: cost = Point2Plane_Angle::Create(V3(row), Eigen::Vector4d(row[3], row[4], row[5], row[6]), normalize, weight); break;
          case 2: cost = Point2Line_Meter::Create(V3(row), V3(row + 3), V3(row + 6), weight); break;
          case 3: cost = Point2Line_Angle::Create(V3(row), V3(row + 3), V3(row + 6), normalize, weight); break;
          case 4: cost = Plane2Plane_Global::Create(V3(row), V3(row + 3), V3(row + 6), row[9]); break;
          default: cost = PlaneIOUResidual::Create(Eigen::Vector4d(row[0], row[1], row[2], row[3]), V3(row + 4), V3(row + 7), row[10], row[11]); break;
        }
        const double* params[4] = {aa.f64() + 3 * ri, t.f64() + 3 * ri, aa.f64() + 3 * ni, t.f64() + 3 * ni};
        double jb[4][3]; double* jac[4] = {jb[0], jb[1], jb[2], jb[3]};
        cost->Evaluate(params, &r[i], jac);                   // ceres::AutoDiffCostFunction<X,1,3,3,3,3>
        for (int b = 0; b < 4; ++b) for (int c = 0; c < 3; ++c) J[i * 12 + 3 * b + c] = jb[b][c];
        delete cost;
      }
    }
    out[k + "r"] = MakeArray(1, {n}, r);
    out[k + "J"] = MakeArray(1, {n, 12}, J);
  }
  WriteBundle(dir + "/functors.ref.pvv", out);
}

// ---- scans ------------------------------------------------------------------------------------------------
// The fixtures hold the feature clouds already in the world frame as float32 (what Transform2LidarWorld produces,
// sensors/Velodyne.cpp:1773-1808).  `world` is private, so: identity pose -> Transform2LidarWorld() (a numerical
// no-op that sets the flag) -> SetPose(real pose).
void FillCloud(const Array& xyz, const Array* tag, pcl::PointCloud<pcl::PointXYZI>& c) {
  const size_t n = xyz.dims.empty() ? 0 : xyz.dims[0];
  c.clear();
  for (size_t i = 0; i < n; ++i) {
    pcl::PointXYZI p; p.x = xyz.f32()[3 * i]; p.y = xyz.f32()[3 * i + 1]; p.z = xyz.f32()[3 * i + 2];
    p.intensity = tag ? tag->f32()[i] : (float)i;
    c.push_back(p);
  }
}
void FinishPose(Velodyne& v, const Array& R, const Array& t) {
  v.SetPose(Eigen::Matrix3d::Identity(), Eigen::Vector3d::Zero());
  v.Transform2LidarWorld();
  Eigen::Matrix3d Rm;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm(i, j) = R.f64()[3 * i + j];
  v.SetPose(Rm, V3(t.f64()));
}

// ---- AssociatePoint2Plane (lidar_mapping/LidarFeatureAssociate.cpp:550-630) ------------------------------------
void DumpPoint2Plane(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/assoc_point2plane.in.pvv");
  if (in.empty()) return;
  std::vector<Velodyne> scans;
  for (int s = 0; s < 3; ++s) {
    const std::string k = "s" + std::to_string(s) + "_";
    Velodyne v(16, s);
    FillCloud(in.at(k + "flat_xyz"), &in.at(k + "flat_tag"), v.surfFlat);
    FillCloud(in.at(k + "less_xyz"), &in.at(k + "less_tag"), v.surfLessFlat);
    FinishPose(v, in.at(k + "R_wl"), in.at(k + "t_wl"));
    scans.push_back(v);
  }
  Bundle out;
  const Array& cases = in.at("cases");
  for (size_t c = 0; c < cases.dims[0]; ++c) {
    const int r = (int)cases.as_double(4 * c), n = (int)cases.as_double(4 * c + 1);
    const double tol = cases.as_double(4 * c + 2); const float thr = (float)cases.as_double(4 * c + 3);
    const std::vector<Point2Plane> a = AssociatePoint2Plane(scans[r], scans[n], tol, thr);
    std::vector<double> point, plane;
    for (const Point2Plane& m : a) {
      for (int i = 0; i < 3; ++i) point.push_back(m.point(i));
      for (int i = 0; i < 4; ++i) plane.push_back(m.plane_coeff(i));
    }
    out["c" + std::to_string(c) + "_point"] = MakeArray(1, {a.size(), 3}, point);
    out["c" + std::to_string(c) + "_plane"] = MakeArray(1, {a.size(), 4}, plane);
  }
  WriteBundle(dir + "/assoc_point2plane.ref.pvv", out);
}

// ---- line features of a scan ------------------------------------------------------------------------------
struct LineScan {
  std::vector<std::set<int>> p2s; eigen_vector<Vector6d> coeffs; eigen_vector<Eigen::Vector3d> ends;
  std::vector<pcl::PointCloud<pcl::PointXYZI>> segmented;
};
LineScan ReadLineScan(const Bundle& in, const std::string& k, const Array& corner_xyz) {
  LineScan l;
  const Array &off = in.at(k + "p2s_off"), &ids = in.at(k + "p2s_ids"), &co = in.at(k + "seg_coeffs"), &ep = in.at(k + "end_points");
  const size_t n = off.count() - 1, S = co.dims[0];
  l.p2s.resize(n); l.segmented.resize(S);
  for (size_t i = 0; i < n; ++i)
    for (int j = off.i32()[i]; j < off.i32()[i + 1]; ++j) {
      l.p2s[i].insert(ids.i32()[j]);
      pcl::PointXYZI p; p.x = corner_xyz.f32()[3 * i]; p.y = corner_xyz.f32()[3 * i + 1]; p.z = corner_xyz.f32()[3 * i + 2]; p.intensity = (float)i;
      l.segmented[ids.i32()[j]].push_back(p);
    }
  for (size_t s = 0; s < S; ++s) {
    Vector6d c; for (int i = 0; i < 6; ++i) c(i) = co.f64()[6 * s + i];
    l.coeffs.push_back(c);
  }
  for (size_t e = 0; e < ep.count() / 3; ++e) l.ends.push_back(V3(ep.f64() + 3 * e));
  return l;
}

// ---- AssociateLine2Line (:442-476) + FindAssociations (:120-197);  AssociateByAngle (CameraLidarLineAssociate.cpp:340-475)
void DumpLines(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/lines.in.pvv");
  if (in.empty()) return;
  Bundle out;
  Velodyne scan[2] = {Velodyne(16, 3), Velodyne(16, 4)};
  const char* names[2] = {"a_", "b_"};
  for (int s = 0; s < 2; ++s) {
    const std::string k = names[s];
    FillCloud(in.at(k + "corner_xyz"), nullptr, scan[s].cornerLessSharp);
    const LineScan l = ReadLineScan(in, k, in.at(k + "corner_xyz"));
    scan[s].point_to_segment = l.p2s; scan[s].segment_coeffs = l.coeffs; scan[s].end_points = l.ends; scan[s].edge_segmented = l.segmented;
    FinishPose(scan[s], in.at(k + "R_wl"), in.at(k + "t_wl"));
  }
  const float thrs[2] = {0.3f, 0.4f}; const char* tags[2] = {"t03_", "t04_"};
  for (int c = 0; c < 2; ++c) {
    const std::vector<Line2Line> a = AssociateLine2Line(scan[0], scan[1], thrs[c]);       // ref = a, nei = b
    std::vector<int32_t> ni, ri; std::vector<double> p1, p2;
    for (const Line2Line& m : a) {
      ni.push_back(m.neighbor_line_idx); ri.push_back(m.ref_line_idx);
      for (int i = 0; i < 3; ++i) { p1.push_back(m.line_point1(i)); p2.push_back(m.line_point2(i)); }
    }
    out[std::string(tags[c]) + "nei_idx"] = MakeArray(2, {a.size()}, ni);
    out[std::string(tags[c]) + "ref_idx"] = MakeArray(2, {a.size()}, ri);
    out[std::string(tags[c]) + "p1"] = MakeArray(1, {a.size(), 3}, p1);
    out[std::string(tags[c]) + "p2"] = MakeArray(1, {a.size(), 3}, p2);
  }
  // camera <-> LiDAR: the scan is LiDAR-local here (CameraLidarOptimizer.cpp:345-377 passes local clouds + T_cl)
  {
    pcl::PointCloud<pcl::PointXYZI> corner;
    FillCloud(in.at("c_corner_local"), nullptr, corner);
    const LineScan l = ReadLineScan(in, "c_", in.at("c_corner_local"));
    Eigen::Matrix4d T;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T(i, j) = in.at("c_T_cl").f64()[4 * i + j];
    std::vector<cv::Vec4f> lines;
    const Array& li = in.at("c_lines");
    for (size_t i = 0; i < li.dims[0]; ++i) lines.emplace_back(li.f32()[4 * i], li.f32()[4 * i + 1], li.f32()[4 * i + 2], li.f32()[4 * i + 3]);
    for (int mult = 0; mult < 2; ++mult) {
      CameraLidarLineAssociate assoc(2880, 5760);
      assoc.AssociateByAngle(lines, l.segmented, l.coeffs, corner, l.p2s, l.ends, T, mult != 0);
      const std::vector<CameraLidarLinePair> pairs = assoc.GetAssociatedPairs();
      std::vector<int32_t> il, ll; std::vector<float> score; std::vector<double> st, en;
      for (const CameraLidarLinePair& p : pairs) {
        il.push_back(p.image_line_id); ll.push_back(p.lidar_line_id); score.push_back(p.angle);
        for (int i = 0; i < 3; ++i) { st.push_back(p.lidar_line_start(i)); en.push_back(p.lidar_line_end(i)); }
      }
      const std::string m = "m" + std::to_string(mult) + "_";
      out[m + "image_line_id"] = MakeArray(2, {pairs.size()}, il);
      out[m + "lidar_line_id"] = MakeArray(2, {pairs.size()}, ll);
      out[m + "score"] = MakeArray(0, {pairs.size()}, score);
      out[m + "start"] = MakeArray(1, {pairs.size(), 3}, st);
      out[m + "end"] = MakeArray(1, {pairs.size(), 3}, en);
    }
  }
  WriteBundle(dir + "/lines.ref.pvv", out);
}

// ---- Equirectangular (sensors/Equirectangular.h:42-204, .cpp:20-65) ----------------------------------------
void DumpEquirect(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/equirect.in.pvv");
  if (in.empty()) return;
  Bundle out;
  const Array& cam = in.at("cam");
  const int sizes[2][2] = {{2880, 5760}, {720, 1440}};
  for (const auto& sz : sizes) {
    const int rows = sz[0], cols = sz[1];
    const Equirectangular eq(rows, cols);
    const size_t n = cam.dims[0];
    std::vector<float> pf(2 * n); std::vector<double> pd(2 * n);
    for (size_t i = 0; i < n; ++i) {
      const double* c = cam.f64() + 3 * i;
      const cv::Point2f a = eq.CamToImage(cv::Point3f((float)c[0], (float)c[1], (float)c[2]));
      const cv::Point2d b = eq.CamToImage(cv::Point3d(c[0], c[1], c[2]));
      pf[2 * i] = a.x; pf[2 * i + 1] = a.y; pd[2 * i] = b.x; pd[2 * i + 1] = b.y;
    }
    out["px_f32_" + std::to_string(rows)] = MakeArray(0, {n, 2}, pf);
    out["px_f64_" + std::to_string(rows)] = MakeArray(1, {n, 2}, pd);
    const Array& pix = in.at("pix_" + std::to_string(rows));
    std::vector<double> cd(3 * pix.dims[0]);
    for (size_t i = 0; i < pix.dims[0]; ++i) {
      const cv::Point3d c = eq.ImageToCam(cv::Point2d(pix.f64()[2 * i], pix.f64()[2 * i + 1]), 1.0);
      cd[3 * i] = c.x; cd[3 * i + 1] = c.y; cd[3 * i + 2] = c.z;
    }
    out["cam_f64_" + std::to_string(rows)] = MakeArray(1, {pix.dims[0], 3}, cd);
    // the fixture's polyline: (100,200) -> (cols-150, rows-300), chord length 100 (tests/golden/make_golden.py)
    Equirectangular eq2(rows, cols);
    const std::vector<cv::Point2f> seg = eq2.BreakToSegments(cv::Point2f(100.f, 200.f), cv::Point2f(cols - 150.f, rows - 300.f), 100.f);
    std::vector<float> sv;
    for (const cv::Point2f& p : seg) { sv.push_back(p.x); sv.push_back(p.y); }
    out["seg_" + std::to_string(rows)] = MakeArray(0, {seg.size(), 2}, sv);
  }
  WriteBundle(dir + "/equirect.ref.pvv", out);
}

// ---- FindNeighbors (lidar_mapping/LidarFeatureAssociate.cpp:19-111) -------------------------------------------
void DumpNeighbors(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/neighbors.in.pvv");
  if (in.empty()) return;
  const Array &poses = in.at("poses"), &valid = in.at("valid");
  std::vector<Velodyne> lidars;
  for (size_t i = 0; i < poses.dims[0]; ++i) {
    Velodyne v(16, (int)i);
    const double* p = poses.f64() + 12 * i;
    if (valid.as_double(i) != 0) {
      Eigen::Matrix3d R; for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R(a, b) = p[3 * a + b];
      v.SetPose(R, V3(p + 9));
    }
    v.valid = valid.as_double(i) != 0;
    lidars.push_back(v);
  }
  const std::vector<std::vector<int>> nb = FindNeighbors(lidars, 6);
  std::vector<int32_t> off(1, 0), ids;
  for (const std::vector<int>& l : nb) { for (int v : l) ids.push_back(v); off.push_back((int32_t)ids.size()); }
  Bundle out;
  out["off"] = MakeArray(2, {off.size()}, off);
  out["ids"] = MakeArray(2, {ids.size()}, ids);
  WriteBundle(dir + "/neighbors.ref.pvv", out);
}

// ---- FastAtan2 (base/Math.h:15-29) ----------------------------------------------------------------------
void DumpFastAtan2(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/fast_atan2.in.pvv");
  if (in.empty()) return;
  const Array &y = in.at("y"), &x = in.at("x");
  std::vector<float> of(y.count()); std::vector<double> od(y.count());
  for (size_t i = 0; i < y.count(); ++i) {
    of[i] = FastAtan2((float)y.f64()[i], (float)x.f64()[i]);
    od[i] = FastAtan2(y.f64()[i], x.f64()[i]);
  }
  Bundle out;
  out["out_f32"] = MakeArray(0, {y.count()}, of);
  out["out_f64"] = MakeArray(1, {y.count()}, od);
  WriteBundle(dir + "/fast_atan2.ref.pvv", out);
}

// ---- PanoramaReprojResidual_1Angle (base/CostFunction.h:218-247) ---------------------------------------------
void DumpReproj(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/reproj.in.pvv");
  if (in.empty()) return;
  const Array &aa = in.at("aa"), &t = in.at("t"), &X = in.at("X"), &cam = in.at("cam"), &pt = in.at("pt"), &bearing = in.at("bearing");
  const double weight = in.at("weight").as_double(0);
  const size_t n = cam.count();
  std::vector<double> r(n), J(n * 9);
  for (size_t i = 0; i < n; ++i) {
    ceres::CostFunction* cost = PanoramaReprojResidual_1Angle::Create(V3(bearing.f64() + 3 * i), weight);
    const int c = (int)cam.as_double(i), p = (int)pt.as_double(i);
    const double* params[3] = {aa.f64() + 3 * c, t.f64() + 3 * c, X.f64() + 3 * p};
    double jb[3][3]; double* jac[3] = {jb[0], jb[1], jb[2]};
    cost->Evaluate(params, &r[i], jac);
    for (int b = 0; b < 3; ++b) for (int k = 0; k < 3; ++k) J[i * 9 + 3 * b + k] = jb[b][k];
    delete cost;
  }
  Bundle out;
  out["r"] = MakeArray(1, {n}, r);
  out["J"] = MakeArray(1, {n, 9}, J);
  WriteBundle(dir + "/reproj.ref.pvv", out);
}

// ---- ProjectLidar2PanoramaDepth (util/Visualization.h:407-441) -----------------------------------------------
void DumpDepth(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/depth.in.pvv");
  if (in.empty()) return;
  pcl::PointCloud<pcl::PointXYZI> cloud;
  FillCloud(in.at("xyz"), nullptr, cloud);
  Eigen::Matrix4d T;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T(i, j) = in.at("T_cl").f64()[4 * i + j];
  const int rows = (int)in.at("rows").as_double(0), cols = (int)in.at("cols").as_double(0);
  Bundle out;
  for (size_t size : {size_t(3), size_t(2)}) {
    const cv::Mat img = ProjectLidar2PanoramaDepth(cloud, rows, cols, T, size);
    std::vector<int32_t> v((size_t)rows * cols);
    for (int u = 0; u < rows; ++u) for (int w = 0; w < cols; ++w) v[(size_t)u * cols + w] = img.at<uint16_t>(u, w);
    out["depth_size" + std::to_string(size)] = MakeArray(2, {(uint64_t)rows, (uint64_t)cols}, v);
  }
  WriteBundle(dir + "/depth.ref.pvv", out);
}

// ---- Velodyne::ReOrderVLP + ExtractFeatures, ADAPTIVE (sensors/Velodyne.cpp:371-526, :531-760) ----------------------
// The fixture's `raw` is the cloud LoadLidar leaves in Velodyne::cloud (camera-style axes).  Public members only: the
// ring-ordered cloud and the four feature clouds; cornerBeforeFilter is the edge cloud before EdgeToLine (what this
// repo calls cornerLessSharp).  The per-point working arrays are private upstream and stay unpinned.
void DumpFeatures(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/features.in.pvv");
  if (in.empty()) return;
  const Array& raw = in.at("raw");
  const int horizon = (int)in.at("horizon").as_double(0);
  Velodyne v(16, 0, horizon);
  for (size_t i = 0; i < raw.dims[0]; ++i) {
    pcl::PointXYZI p; p.x = raw.f32()[4 * i]; p.y = raw.f32()[4 * i + 1]; p.z = raw.f32()[4 * i + 2]; p.intensity = raw.f32()[4 * i + 3];
    v.cloud.push_back(p);
  }
  v.ReOrderVLP();
  v.ExtractFeatures(1000.f, 5.f, ADAPTIVE, true);
  auto flat = [](const pcl::PointCloud<pcl::PointXYZI>& c) {
    std::vector<float> o;
    for (const auto& p : c.points) { o.push_back(p.x); o.push_back(p.y); o.push_back(p.z); o.push_back(p.intensity); }
    return o;
  };
  Bundle out;
  out["cloud_scan"] = MakeArray(0, {v.cloud_scan.size(), 4}, flat(v.cloud_scan));
  out["cornerSharp"] = MakeArray(0, {v.cornerSharp.size(), 4}, flat(v.cornerSharp));       // NB: upstream re-filters cornerSharp in EdgeToLine (:1312-1323)
  out["cornerLessSharp"] = MakeArray(0, {v.cornerBeforeFilter.size(), 4}, flat(v.cornerBeforeFilter));
  out["surfFlat"] = MakeArray(0, {v.surfFlat.size(), 4}, flat(v.surfFlat));
  out["surfLessFlat"] = MakeArray(0, {v.surfLessFlat.size(), 4}, flat(v.surfLessFlat));
  WriteBundle(dir + "/features.ref.pvv", out);
}

// ---- the LINE branch: Velodyne::EdgeToLine (sensors/Velodyne.cpp:1269-1324) behind ExtractFeatures(ADAPTIVE) (:746-752) ----
// ExtractLineFeatures (sensors/LidarLineExtraction.cpp:296-389) -> edge_segmented / segment_coeffs, FurthestPoints +
// ProjectPoint2Line3D -> end_points, the de-duplicated cornerLessSharp with point_to_segment, the re-filtered cornerSharp.
// All public members (sensors/Velodyne.h:88-91).  NB: FuseLines fits with pcl's SAC_RANSAC (:148-175), whose draws depend
// on pcl's generator; this repo's oracle takes the exhaustive 2-point maximum consensus, so segment_coeffs / end_points
// agree only where the consensus line is unique (tools/refvec.py compares them by direction and distance, not bit for bit).
void DumpLineExtraction(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/line_extraction.in.pvv");
  if (in.empty()) return;
  const Array& raw = in.at("raw");
  const int horizon = (int)in.at("horizon").as_double(0);
  Velodyne v(16, 0, horizon);
  for (size_t i = 0; i < raw.dims[0]; ++i) {
    pcl::PointXYZI p; p.x = raw.f32()[4 * i]; p.y = raw.f32()[4 * i + 1]; p.z = raw.f32()[4 * i + 2]; p.intensity = raw.f32()[4 * i + 3];
    v.cloud.push_back(p);
  }
  v.ReOrderVLP();
  v.ExtractFeatures(1000.f, 5.f, ADAPTIVE, true);
  auto flat = [](const pcl::PointCloud<pcl::PointXYZI>& c) {
    std::vector<float> o;
    for (const auto& p : c.points) { o.push_back(p.x); o.push_back(p.y); o.push_back(p.z); o.push_back(p.intensity); }
    return o;
  };
  Bundle out;
  out["cornerBeforeFilter"] = MakeArray(0, {v.cornerBeforeFilter.size(), 4}, flat(v.cornerBeforeFilter));
  out["cornerLessSharp"] = MakeArray(0, {v.cornerLessSharp.size(), 4}, flat(v.cornerLessSharp));
  out["cornerSharp"] = MakeArray(0, {v.cornerSharp.size(), 4}, flat(v.cornerSharp));
  out["surfFlat"] = MakeArray(0, {v.surfFlat.size(), 4}, flat(v.surfFlat));
  out["surfLessFlat"] = MakeArray(0, {v.surfLessFlat.size(), 4}, flat(v.surfLessFlat));
  std::vector<int32_t> seg_off = {0}, p2s_off = {0}, p2s_ids;
  std::vector<float> seg_pts;
  for (const auto& c : v.edge_segmented) {
    const std::vector<float> f = flat(c);
    seg_pts.insert(seg_pts.end(), f.begin(), f.end());
    seg_off.push_back((int32_t)(seg_pts.size() / 4));
  }
  for (const std::set<int>& sset : v.point_to_segment) {
    for (int id : sset) p2s_ids.push_back(id);                      // std::set: ascending, as the golden lists them
    p2s_off.push_back((int32_t)p2s_ids.size());
  }
  std::vector<double> coeffs, ends;
  for (const auto& c : v.segment_coeffs) for (int k = 0; k < 6; ++k) coeffs.push_back(c[k]);
  for (const auto& e : v.end_points) for (int k = 0; k < 3; ++k) ends.push_back(e[k]);
  out["seg_offsets"] = MakeArray(2, {seg_off.size()}, seg_off);
  out["seg_points"] = MakeArray(0, {seg_pts.size() / 4, 4}, seg_pts);
  out["segment_coeffs"] = MakeArray(1, {v.segment_coeffs.size(), 6}, coeffs);
  out["end_points"] = MakeArray(1, {v.end_points.size() / 2, 2, 3}, ends);
  out["p2s_offsets"] = MakeArray(2, {p2s_off.size()}, p2s_off);
  out["p2s_ids"] = MakeArray(2, {p2s_ids.size()}, p2s_ids);
  WriteBundle(dir + "/line_extraction.ref.pvv", out);
}

}  // namespace

// ---- Velodyne::UndistortCloud (sensors/Velodyne.cpp:1642-1674), SlerpPose (base/Geometry.hpp:572-583) --------------------------
void DumpUndistort(const std::string& dir) {
  const Bundle in = ReadBundle(dir + "/undistort.in.pvv");
  if (in.empty()) return;
  Bundle out;
  const int cases = (int)in.at("cases").as_double(0);
  auto mat3 = [&](const std::string& k) { Eigen::Matrix3d m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = in.at(k).f64()[3 * i + j]; return m; };
  auto vec3 = [&](const std::string& k) { return Eigen::Vector3d(in.at(k).f64()[0], in.at(k).f64()[1], in.at(k).f64()[2]); };
  for (int c = 0; c < cases; ++c) {
    const std::string id = std::to_string(c);
    const Array& raw = in.at("cloud" + id);
    Velodyne v(16, 0, 1800);
    for (size_t i = 0; i < raw.dims[0]; ++i) {
      pcl::PointXYZI p; p.x = raw.f32()[4 * i]; p.y = raw.f32()[4 * i + 1]; p.z = raw.f32()[4 * i + 2]; p.intensity = raw.f32()[4 * i + 3];
      v.cloud.push_back(p);
    }
    v.SetRotation(mat3("R_wl" + id)); v.SetTranslation(vec3("t_wl" + id));
    v.UndistortCloud(mat3("R_we" + id), vec3("t_we" + id));
    std::vector<float> o;
    for (const auto& p : v.cloud.points) { o.push_back(p.x); o.push_back(p.y); o.push_back(p.z); o.push_back(p.intensity); }
    out["out" + id] = MakeArray(0, {v.cloud.size(), 4}, o);
  }
  Eigen::Matrix4d T1, T2;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { T1(i, j) = in.at("pose_w1").f64()[4 * i + j]; T2(i, j) = in.at("pose_w2").f64()[4 * i + j]; }
  const Array& ratios = in.at("ratios");
  std::vector<double> poses;
  for (size_t r = 0; r < ratios.dims[0]; ++r) {
    const Eigen::Matrix4d T = SlerpPose<double>(T1, T2, ratios.f64()[r]);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) poses.push_back(T(i, j));
  }
  out["slerp"] = MakeArray(1, {ratios.dims[0], 4, 4}, poses);
  WriteBundle(dir + "/undistort.ref.pvv", out);
}

int main(int argc, char** argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: %s DIR   (DIR holds the *.in.pvv files of `python tools/refvec.py export DIR`)\n", argv[0]); return 2; }
  google::InitGoogleLogging(argv[0]);
  const std::string dir = argv[1];
  DumpFastAtan2(dir);
  DumpFunctors(dir);
  DumpEquirect(dir);
  DumpNeighbors(dir);
  DumpPoint2Plane(dir);
  DumpLines(dir);
  DumpReproj(dir);
  DumpDepth(dir);
  DumpFeatures(dir);
  DumpLineExtraction(dir);
  DumpUndistort(dir);
  return 0;
}
